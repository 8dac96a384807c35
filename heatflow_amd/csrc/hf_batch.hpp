// Part of libheatflow_hip.so (see heatflow_hip.hip): batched time loop - NV sweep points advance together.
//
// Sweep points on one mesh share the sparsity pattern, the mass matrix, the Dirichlet set and (with a frozen
// hierarchy) every coarse operator of the multigrid preconditioner; they differ in the boundary values (fwhm
// sweeps: even the fine operator A is shared) and / or in the values of A (kappa sweeps).  Reference: the grid of
// parameter_sweep.py:195-235 and the kappa list of sweep_test.py:47-52, which the reference farms out as
// independent runs.  Here NV points (2, 4, 8 or 16) are the columns of one multi-vector PCG:
//   * every vector is stored interleaved, x[i * NV + j] = entry i of column j, so a gather of x[col] fetches the
//     NV columns as one contiguous 8*NV-byte segment and every index / shared value is read once for NV products;
//   * the fine operator is either shared (one value per nonzero) or per column (values interleaved like vectors);
//   * each column keeps its own PCG scalars (alpha, beta, tolerance, iteration count, done flag) on the device; a
//     converged column freezes (its x and r are no longer touched) while the others finish;
//   * at stock mesh sizes, where a single run is launch-bound, NV columns per launch cost about one.
// The algorithm per column is exactly hf_step's (same kernels' arithmetic, same stopping rule); the start vector
// is, per column, the A-norm projection of the new solution on the span of that column's last solutions
// (hf_set_start_vector kind 3 without the boundary responses; kb_proj_*).
#pragma once
#include "hf_solver.hpp"

namespace {

// Sum over the threads of a 256-thread block that serve the same column j = threadIdx.x % NV, fixed order.
template <int NV, int NW = 4>
__device__ __forceinline__ double block_colsum(double v, double* sw /* [NW * NV], NW = wavefronts of the block */) {
#pragma unroll
  for (int o = 32; o >= NV; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l < NV) sw[w * NV + l] = v;
  __syncthreads();
  const int j = threadIdx.x % NV;
  double t = sw[j];
#pragma unroll
  for (int k = 1; k < NW; ++k) t += sw[k * NV + j];
  __syncthreads();
  return t;
}

// Sum of a column's P per-workgroup partials (P <= MAXP): TPB / NV threads serve the column, each adds its entries
// in a fixed order; the loads are issued together (predicated), not one round trip after the other.
template <int NV>
__device__ __forceinline__ double col_partials(const double* __restrict__ part /* column's MAXP slots */, int P, double* sw) {
  constexpr int STRIDE = TPB / NV, NL = (MAXP + STRIDE - 1) / STRIDE;
  double v = 0.0;
#pragma unroll
  for (int u0 = 0; u0 < NL; u0 += 8) {
    double e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = threadIdx.x / NV + (u0 + u) * STRIDE;
      e[u] = (u0 + u < NL && k < P) ? part[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v += e[u];
  }
  return block_colsum<NV>(v, sw);
}

// The per-column scalars (BRed) are reduced ONCE per producer by kb_reduce (one small workgroup) instead of by every
// workgroup of every consumer: with NV columns the partial arrays are NV times larger than in the single-column
// loop, and re-reading them in each of ~1000 consumer workgroups would cost as much as the matrix.

// one workgroup per (array, column): 256 threads add the column's P partials in a fixed order
__global__ __launch_bounds__(TPB) void kb_reduce(int P, int nv, const double* __restrict__ part_a, double* __restrict__ out_a,
                                                 const double* __restrict__ part_b, double* __restrict__ out_b,
                                                 Scal* __restrict__ test /* non-null: out_a is (D^-1 r)^2 - mark converged columns done */,
                                                 ScalMirror* __restrict__ mirror /* with test: publish each column's outcome to the host */) {
  __shared__ double s4[4];
  const int j = blockIdx.x % nv;
  const bool second = blockIdx.x >= nv;
  const double* part = (second ? part_b : part_a) + static_cast<size_t>(j) * MAXP;
  double e[MAXP / TPB];
#pragma unroll
  for (int u = 0; u < MAXP / TPB; ++u) {
    const int k = threadIdx.x + u * TPB;
    e[u] = k < P ? part[k] : 0.0;
  }
  double v = 0.0;
#pragma unroll
  for (int u = 0; u < MAXP / TPB; ++u) v += e[u];
  const double t = block_sum(v, s4);
  if (threadIdx.x == 0) {
    (second ? out_b : out_a)[j] = t;
    // A column that had converged before this launch publishes nothing more: the launches of a blind first burst that
    // run past the convergence of every column must not write into the mirrors the host may already be reading for the
    // next solve (what they would write carries the old epoch besides, so even a late store cannot be taken for progress)
    if (test != nullptr && !second && test[j].done != 1) {
      if (test[j].done == 0 && t <= test[j].tol2) { test[j].zz = t; test[j].done = 1; }
      if (mirror != nullptr) mirror_publish(mirror + j, t, test[j].iters, test[j].done, test[j].epoch);
    }
  }
}

// The fine operator of column j of a batch:
//   OP_SHARED  A_j = A            one value per nonzero (points that differ in their boundary values only)
//   OP_PERCOL  A_j arbitrary      values interleaved like the vectors, v[k * NV + j]
//   OP_AFFINE  A_j = A + d_j A1   two shared value arrays and one scalar per column: a sweep over the conductivity of
//                                 one material (or of materials that move together), A1 = dt K restricted to it
enum { OP_SHARED = 0, OP_PERCOL = 1, OP_AFFINE = 2 };
struct BOp { const double* v0; const double* v1; double delta[NV_MAX]; };

template <int OPK, int NV>
__device__ __forceinline__ double op_value(const BOp& op, size_t k, int j) {
  if (OPK == OP_PERCOL) return op.v0[k * NV + j];
  if (OPK == OP_AFFINE) return op.v0[k] + op.delta[j] * op.v1[k];
  return op.v0[k];
}

// Fine-pattern SpMV on NV interleaved columns, thread = (row, column).  Modes as k_spmv (0, 2, 3, 4, 5, 8, 9).
#ifndef HF_KB_U
#define HF_KB_U 4
#endif
constexpr int KB_U = HF_KB_U;  // products per batch of predicated loads in kb_spmv
constexpr int KB_BT = 512;   // threads per workgroup of kb_spmv: 32 wavefronts per CU with the <= 1024 workgroups of a launch
template <int MODE, int NV, int OPK>
__global__ __launch_bounds__(KB_BT) void kb_spmv(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                               const BOp op, const double* __restrict__ x, double* __restrict__ y,
                                               Scal* __restrict__ scal, double* __restrict__ part0, const double* __restrict__ bvec,
                                               const double* __restrict__ dinv, double* __restrict__ pvec, double* __restrict__ part1,
                                               double* __restrict__ part2, double w, const BRed* __restrict__ red, int parity) {
  __shared__ double sw[(KB_BT / 64) * NV];
  constexpr int RPB = KB_BT / NV;
  const int j = threadIdx.x % NV, rl = threadIdx.x / NV;
  Scal* sc = scal + j;
  bool active = true;
  if (MODE == 3 || MODE == 4 || MODE == 9) active = sc->done == 0;
  double beta = 0.0;
  bool first9 = false;
  if (MODE == 9) {
    first9 = sc->first != 0;
    const double rz_new = red->rz[parity][j], rz_old = red->rz[parity ^ 1][j], zz = red->zz[j];
    if (active && !first9) {
      const bool conv = zz <= sc->tol2;
      if (blockIdx.x == 0 && rl == 0) {
        sc->zz = zz;
        if (conv) sc->done = 1;
      }
      if (conv) active = false;
      beta = rz_new / rz_old;
    }
  }
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  const int nrb = (n + RPB - 1) / RPB;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int row = rb * RPB + rl;
    if (row >= n || !active) continue;
    const int k0 = rowptr[row], k1 = rowptr[row + 1];
    const size_t o = static_cast<size_t>(row) * NV + j;
    double e_b = 0.0, e_d = 0.0, e_y = 0.0, e_p = 0.0, e_x = 0.0;
    if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 8) e_b = bvec[o];
    if (MODE == 2 || MODE == 4 || MODE == 5) e_d = OPK != OP_SHARED ? dinv[o] : dinv[row];
    if (MODE == 9 && !first9) { e_y = y[o]; e_p = pvec[o]; }
    if (MODE == 4 || MODE == 8 || MODE == 9) e_x = x[o];
    double s = 0.0;
    for (int k = k0; k < k1; k += KB_U) {
      int c[KB_U];
      double v[KB_U], xv[KB_U];
#pragma unroll
      for (int u = 0; u < KB_U; ++u) {
        const bool in = k + u < k1;
        c[u] = in ? colidx[k + u] : 0;
        v[u] = in ? op_value<OPK, NV>(op, static_cast<size_t>(k + u), j) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < KB_U; ++u) xv[u] = (k + u < k1) ? x[static_cast<size_t>(c[u]) * NV + j] : 0.0;
#pragma unroll
      for (int u = 0; u < KB_U; ++u) s += v[u] * xv[u];
    }
    if (MODE == 0) {
      y[o] = s;
    } else if (MODE == 2) {
      const double ri = e_b - s, zi = e_d * ri;
      y[o] = ri;
      pvec[o] = zi;
      acc0 += ri * zi;
      acc1 += zi * zi;
      acc2 += (e_d * e_b) * (e_d * e_b);
    } else if (MODE == 3) {
      y[o] = e_b - s;
    } else if (MODE == 4) {
      const double yi = e_x + w * e_d * (e_b - s);
      y[o] = yi;
      acc0 += e_b * yi;
    } else if (MODE == 5) {
      const double ri = e_b - s;
      y[o] = ri;
      pvec[o] = w * e_d * ri;
      acc1 += (e_d * ri) * (e_d * ri);
      acc2 += (e_d * e_b) * (e_d * e_b);
    } else if (MODE == 8) {
      y[o] = s;
      pvec[o] = 2.0 * e_x - e_b;
    } else {
      const double api = first9 ? s : s + beta * e_y;
      const double pi = first9 ? e_x : e_x + beta * e_p;
      y[o] = api;
      pvec[o] = pi;
      acc0 += pi * api;
    }
  }
  if (MODE == 2 || MODE == 9 || (MODE == 4 && part0 != nullptr)) {
    const double t0 = block_colsum<NV, KB_BT / 64>(acc0, sw);
    if (rl == 0) part0[j * MAXP + blockIdx.x] = t0;
  }
  if (MODE == 2 || MODE == 5) {
    const double t1 = block_colsum<NV, KB_BT / 64>(acc1, sw);
    const double t2 = block_colsum<NV, KB_BT / 64>(acc2, sw);
    if (rl == 0) { part1[j * MAXP + blockIdx.x] = t1; part2[j * MAXP + blockIdx.x] = t2; }
  }
}


// The same operation with the chunk's operands staged in LDS (what k_spmv's C16 path does for one column): a workgroup
// takes chunks of `rpc` consecutive rows (a multiple of KB_BT / NV, chosen so that one launch is about one round of
// resident workgroups); all lanes stream the chunk's matrix values (one array, or two for the affine family) and the
// 16-bit column positions in nnz order - coalesced 8-byte lanes instead of eight rows' worth of scattered words per
// wavefront - and gather the chunk's slice of x through its sorted column list ONCE: NV contiguous doubles per list
// entry, own rows plus a halo, 1.3-1.5 entries per row.  After one barrier thread (row, column) walks its row in LDS:
// value(s), position, operand.  The row's own operand x[row] of modes 4 / 8 / 9 comes from the staged slice (a chunk's
// rows are a contiguous run of its sorted list).  Per column the products are added in CSR order, as in kb_spmv.
// Shared and affine operators only: per-column values (NV doubles per nonzero) have no shared stream to stage.
struct BComp {
  const int32_t* cptr;    // chunk -> first entry of its column list (nchunks + 1)
  const int32_t* dict;    // the lists (global column numbers, sorted per chunk)
  const int32_t* own;     // position of the chunk's first row in its list
  const uint16_t* id;     // per nonzero: position of its column in the chunk's list
  int rpc, nchunks, cap_nnz /* even */, cap_dict;
};

// measurement builds (-DHF_PHASE_CLOCK=<any> -DHF_PHASE_CLOCK_B=<mode>, PHASE_BATCH_NV=<nv> scripts/phase_clock.py): phase stamps of
// kb_spmv_lds<mode> as in k_spmv
#ifndef HF_PHASE_CLOCK_B
#define HF_PHASE_CLOCK_B -1
#endif
#if HF_PHASE_CLOCK_B >= 0 && HF_PHASE_CLOCK >= 0
#define HFB_STAMP(slot)                                                                                          \
  do {                                                                                                           \
    if (MODE == HF_PHASE_CLOCK_B && threadIdx.x == 0 && (slot) < 16) g_phase[blockIdx.x * 16 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define HFB_STAMP(slot) do { } while (0)
#endif

template <int MODE, int NV, int OPK>
__global__ __launch_bounds__(KB_BT) void kb_spmv_lds(int n, const int32_t* __restrict__ rowptr, const BOp op,
                                                   const double* __restrict__ x, double* __restrict__ y, Scal* __restrict__ scal,
                                                   double* __restrict__ part0, const double* __restrict__ bvec,
                                                   const double* __restrict__ dinv, double* __restrict__ pvec,
                                                   double* __restrict__ part1, double* __restrict__ part2, double w,
                                                   const BRed* __restrict__ red, int parity, const BComp comp, int npart) {
  static_assert(OPK != OP_PERCOL, "per-column values are not staged");
  extern __shared__ double sdyn[];
  __shared__ double sw[(KB_BT / 64) * NV];
  double* sv0 = sdyn;                                                       // [cap_nnz]
  double* sv1 = sdyn + comp.cap_nnz;                                        // [cap_nnz] (affine)
  double* sx = sdyn + (OPK == OP_AFFINE ? 2 : 1) * comp.cap_nnz;            // [cap_dict * NV]
  uint16_t* sid = reinterpret_cast<uint16_t*>(sx + static_cast<size_t>(comp.cap_dict) * NV);   // [cap_nnz]
  constexpr int RPP = KB_BT / NV;                                           // rows per pass
  const int j = threadIdx.x % NV, rl = threadIdx.x / NV;
  Scal* sc = scal + j;
  bool active = true;
  if (MODE == 3 || MODE == 4 || MODE == 9) active = sc->done == 0;
  double beta = 0.0;
  bool first9 = false;
  if (MODE == 9) {
    first9 = sc->first != 0;
    const double rz_new = red->rz[parity][j], rz_old = red->rz[parity ^ 1][j], zz = red->zz[j];
    if (active && !first9) {
      const bool conv = zz <= sc->tol2;
      if (blockIdx.x == 0 && rl == 0) {
        sc->zz = zz;
        if (conv) sc->done = 1;
      }
      if (conv) active = false;
      beta = rz_new / rz_old;
    }
  }
  // a workgroup whose columns have all ended has nothing to stage (the kernels of a blind burst after convergence)
  const bool any_active = __syncthreads_or(active ? 1 : 0) != 0;
  const double dj = OPK == OP_AFFINE ? op.delta[j] : 0.0;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  const ChunkIter sched(comp.nchunks);
  HFB_STAMP(0);
  int stamp_at = 1;
  (void)stamp_at;
  for (int chunk = sched.chunk; any_active && chunk < sched.end; chunk += sched.step) {
    const int r0 = chunk * comp.rpc, r1 = min(n, r0 + comp.rpc);
    const int k0 = rowptr[r0], nk = rowptr[r1] - k0;
    const int d0 = comp.cptr[chunk], nd = comp.cptr[chunk + 1] - d0;
    // ---- stage: matrix stream (values + positions), then the operand slice through the column list
    for (int k = threadIdx.x; k < nk; k += 2 * KB_BT) {
      const bool two = k + KB_BT < nk;
      const double a0 = op.v0[k0 + k], b0 = two ? op.v0[k0 + k + KB_BT] : 0.0;
      double a1 = 0.0, b1 = 0.0;
      if (OPK == OP_AFFINE) { a1 = op.v1[k0 + k]; b1 = two ? op.v1[k0 + k + KB_BT] : 0.0; }
      const uint16_t ia = comp.id[k0 + k], ib = two ? comp.id[k0 + k + KB_BT] : static_cast<uint16_t>(0);
      sv0[k] = a0; sid[k] = ia;
      if (OPK == OP_AFFINE) sv1[k] = a1;
      if (two) {
        sv0[k + KB_BT] = b0; sid[k + KB_BT] = ib;
        if (OPK == OP_AFFINE) sv1[k + KB_BT] = b1;
      }
    }
    HFB_STAMP(stamp_at); ++stamp_at;      // matrix stream requested and parked (lane 0's share)
    const int nx = nd * NV;
    for (int i = threadIdx.x; i < nx; i += 2 * KB_BT) {
      const bool two = i + KB_BT < nx;
      const int c0 = comp.dict[d0 + i / NV], c1 = two ? comp.dict[d0 + (i + KB_BT) / NV] : 0;     // KB_BT is a multiple of NV: i % NV == j
      const double x0 = x[static_cast<size_t>(c0) * NV + j], x1 = two ? x[static_cast<size_t>(c1) * NV + j] : 0.0;
      sx[i] = x0;
      if (two) sx[i + KB_BT] = x1;
    }
    const int own = comp.own[chunk];
    __syncthreads();
    HFB_STAMP(stamp_at); ++stamp_at;      // chunk staged
    // ---- products: thread (row, column), RPP rows per pass
    for (int rb = r0; rb < r1; rb += RPP) {
      const int row = rb + rl;
      if (row >= r1 || !active) continue;
      const int pa = rowptr[row] - k0, pb = rowptr[row + 1] - k0;
      const size_t o = static_cast<size_t>(row) * NV + j;
      double e_b = 0.0, e_d = 0.0, e_y = 0.0, e_p = 0.0, e_x = 0.0;
      if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 8) e_b = bvec[o];
      if (MODE == 2 || MODE == 4 || MODE == 5) e_d = OPK != OP_SHARED ? dinv[o] : dinv[row];
      if (MODE == 9 && !first9) { e_y = y[o]; e_p = pvec[o]; }
      if (MODE == 4 || MODE == 8 || MODE == 9) e_x = sx[static_cast<size_t>(own + (row - r0)) * NV + j];
      double s = 0.0;
      for (int k = pa; k < pb; ++k) {
        const double v = OPK == OP_AFFINE ? sv0[k] + dj * sv1[k] : sv0[k];
        s += v * sx[static_cast<int>(sid[k]) * NV + j];
      }
      if (MODE == 0) {
        y[o] = s;
      } else if (MODE == 2) {
        const double ri = e_b - s, zi = e_d * ri;
        y[o] = ri;
        pvec[o] = zi;
        acc0 += ri * zi;
        acc1 += zi * zi;
        acc2 += (e_d * e_b) * (e_d * e_b);
      } else if (MODE == 3) {
        y[o] = e_b - s;
      } else if (MODE == 4) {
        const double yi = e_x + w * e_d * (e_b - s);
        y[o] = yi;
        acc0 += e_b * yi;
      } else if (MODE == 5) {
        const double ri = e_b - s;
        y[o] = ri;
        pvec[o] = w * e_d * ri;
        acc1 += (e_d * ri) * (e_d * ri);
        acc2 += (e_d * e_b) * (e_d * e_b);
      } else if (MODE == 8) {
        y[o] = s;
        pvec[o] = 2.0 * e_x - e_b;
      } else {
        const double api = first9 ? s : s + beta * e_y;
        const double pi = first9 ? e_x : e_x + beta * e_p;
        y[o] = api;
        pvec[o] = pi;
        acc0 += pi * api;
      }
    }
    __syncthreads();
    HFB_STAMP(stamp_at); ++stamp_at;      // rows done
  }
  // the consumers (kb_reduce) add `npart` slots per column: this launch may have fewer workgroups, the rest are zeros
  if (MODE == 2 || MODE == 9 || (MODE == 4 && part0 != nullptr)) {
    const double t0 = block_colsum<NV, KB_BT / 64>(acc0, sw);
    if (rl == 0) {
      part0[j * MAXP + blockIdx.x] = t0;
      for (int q = blockIdx.x + gridDim.x; q < npart; q += gridDim.x) part0[j * MAXP + q] = 0.0;
    }
  }
  if (MODE == 2 || MODE == 5) {
    const double t1 = block_colsum<NV, KB_BT / 64>(acc1, sw);
    const double t2 = block_colsum<NV, KB_BT / 64>(acc2, sw);
    if (rl == 0) {
      part1[j * MAXP + blockIdx.x] = t1; part2[j * MAXP + blockIdx.x] = t2;
      for (int q = blockIdx.x + gridDim.x; q < npart; q += gridDim.x) { part1[j * MAXP + q] = 0.0; part2[j * MAXP + q] = 0.0; }
    }
  }
}

// Generic CSR operator (shared values) times NV interleaved columns, thread = (entry lane, column): EL * NV threads
// share a row, thread (e, j) adds the row's entries k = e (mod EL) for column j - one accumulator per thread, the index
// and value loads are broadcasts among the NV column lanes, the operand and result accesses NV contiguous doubles -
// and the EL partial sums are combined by log2(EL) shuffles.  For operators of many rows (the level-1 up leg of the stock
// mesh: 30 k rows of 23 entries, 16 -> 11 us): kb_csr below spends NV * log2(lanes) shuffles per row, this kernel
// log2(EL); for the few long rows of the coarser levels kb_csr's shorter dependent-load chain wins (measured).
// VMODE 0: y = A x, 1: y += A x.
template <int NV, int CPL /* columns per lane */, int EL /* entry lanes */, int VMODE, typename VT>
__global__ __launch_bounds__(TPB) void kb_csr_rc(int nrow, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                              const VT* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  constexpr int NVL = NV / CPL;                // lanes that share an entry, each with CPL adjacent columns
  constexpr int TPR = EL * NVL;                // threads per row: a power of two <= 64
  static_assert(NV % CPL == 0 && TPR <= 64 && (TPR & (TPR - 1)) == 0, "threads per row");
  const int jl = threadIdx.x % NVL, e = (threadIdx.x / NVL) % EL;
  const int rows_per_pass = (gridDim.x * TPB) / TPR;
  for (int row = (blockIdx.x * TPB + threadIdx.x) / TPR; row < nrow; row += rows_per_pass) {
    const int k1 = ptr[row + 1];
    double* yo = y + static_cast<size_t>(row) * NV + jl * CPL;
    double s[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) s[q] = 0.0;
    for (int k = ptr[row] + e; k < k1; k += 4 * EL) {
      int c[4];
      double v[4], xv[4][CPL];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = k + u * EL < k1;
        c[u] = in ? idx[k + u * EL] : 0;
        v[u] = in ? static_cast<double>(val[k + u * EL]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* xs = x + static_cast<size_t>(c[u]) * NV + jl * CPL;
#pragma unroll
        for (int q = 0; q < CPL; ++q) xv[u][q] = (k + u * EL < k1) ? xs[q] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPL; ++q) s[q] += v[u] * xv[u][q];
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
      for (int o = EL / 2; o > 0; o >>= 1) s[q] += __shfl_down(s[q], o * NVL, TPR);
    if (e == 0) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) yo[q] = (VMODE == 1) ? yo[q] + s[q] : s[q];
    }
  }
}

// x = Ainv b with the dense inverse of the coarsest operator in the same mapping: a wavefront per row, lane = (column lane of
// 64 * CPL / NV, column group of CPL columns).
template <int NV, int CPL, typename VT>
__global__ __launch_bounds__(TPB) void kb_dense_rc(int n, int ld, const VT* __restrict__ Ainv, const double* __restrict__ b,
                                                   double* __restrict__ x) {
  constexpr int NVL = NV / CPL, CL = 64 / NVL;
  const int lane = threadIdx.x & 63;
  const int jl = lane % NVL, cl = lane / NVL;
  const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * TPB) >> 6;
  for (int row = wave; row < n; row += nwaves) {
    const VT* arow = Ainv + static_cast<size_t>(row) * ld;
    double s[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) s[q] = 0.0;
    for (int c = cl; c < n; c += 4 * CL) {
      double a[4], bv[4][CPL];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = c + u * CL < n;
        a[u] = in ? static_cast<double>(arow[c + u * CL]) : 0.0;
        const double* bs = b + static_cast<size_t>(in ? c + u * CL : 0) * NV + jl * CPL;
#pragma unroll
        for (int q = 0; q < CPL; ++q) bv[u][q] = in ? bs[q] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPL; ++q) s[q] += a[u] * bv[u][q];
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
      for (int o = CL / 2; o > 0; o >>= 1) s[q] += __shfl_down(s[q], o * NVL, 64);
    if (cl == 0) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) x[static_cast<size_t>(row) * NV + jl * CPL + q] = s[q];
    }
  }
}

// Generic CSR operator (shared values) times NV interleaved columns: LANES lanes share a row, each keeps NV
// accumulators, so every index and value is read once.  VMODE 0: y = A x, 1: y += A x.
template <int NV, int LANES, int VMODE, typename VT>
__global__ __launch_bounds__(TPB) void kb_csr(int nrow, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                              const VT* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x % LANES;
  const int rows_per_pass = (gridDim.x * TPB) / LANES;
  for (int row = (blockIdx.x * TPB + threadIdx.x) / LANES; row < nrow; row += rows_per_pass) {
    const int k1 = ptr[row + 1];
    double acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0;
    for (int k = ptr[row] + lane; k < k1; k += 2 * LANES) {
      const bool in2 = k + LANES < k1;
      const int c0 = idx[k], c1 = in2 ? idx[k + LANES] : 0;
      const double v0 = static_cast<double>(val[k]), v1 = in2 ? static_cast<double>(val[k + LANES]) : 0.0;
      const double* x0 = x + static_cast<size_t>(c0) * NV;
      const double* x1 = x + static_cast<size_t>(c1) * NV;
      double a0[NV], a1[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) { a0[j] = x0[j]; a1[j] = x1[j]; }
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[j] += v0 * a0[j] + v1 * a1[j];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int o = LANES / 2; o > 0; o >>= 1) acc[j] += __shfl_down(acc[j], o, LANES);
    if (lane == 0) {
      double* yo = y + static_cast<size_t>(row) * NV;
#pragma unroll
      for (int j = 0; j < NV; ++j) yo[j] = (VMODE == 1) ? yo[j] + acc[j] : acc[j];
    }
  }
}

// x = Ainv b, dense inverse of the coarsest operator (row-major, leading dimension ld), NV interleaved columns:
// one wavefront per row, every matrix entry read once.
template <int NV, typename VT>
__global__ __launch_bounds__(TPB) void kb_dense(int n, int ld, const VT* __restrict__ Ainv, const double* __restrict__ b,
                                                double* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * TPB) >> 6;
  for (int row = wave; row < n; row += nwaves) {
    const VT* arow = Ainv + static_cast<size_t>(row) * ld;
    double acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0;
    for (int c = lane; c < n; c += 64) {
      const double a = static_cast<double>(arow[c]);
      const double* bc = b + static_cast<size_t>(c) * NV;
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[j] += a * bc[j];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_down(acc[j], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) x[static_cast<size_t>(row) * NV + j] = acc[j];
    }
  }
}

template <int NV>
__global__ __launch_bounds__(TPB) void kb_scale(int n, double w, const double* __restrict__ dinv, const double* __restrict__ b,
                                                double* __restrict__ x) {
  for (size_t q = static_cast<size_t>(blockIdx.x) * TPB + threadIdx.x; q < static_cast<size_t>(n) * NV; q += static_cast<size_t>(gridDim.x) * TPB)
    x[q] = w * dinv[q / NV] * b[q];
}

// PCG start per column: tolerance and convergence of the initial iterate (one workgroup); also reduces the start
// kernel's partial sums (Jacobi: r.z into slot 0).
template <int NV>
__global__ __launch_bounds__(TPB) void kb_begin(int P, double rtol, double atol, const double* __restrict__ part_zz,
                                                const double* __restrict__ part_bn, const double* __restrict__ part_rz0,
                                                Scal* __restrict__ scal, BRed* __restrict__ red, ScalMirror* __restrict__ mirror,
                                                unsigned epoch) {
  __shared__ double sw[4 * NV];
  const int j = threadIdx.x % NV;
  const double zz = col_partials<NV>(part_zz + j * MAXP, P, sw);
  const double bn2 = col_partials<NV>(part_bn + j * MAXP, P, sw);
  double rz0 = 0.0;
  if (part_rz0 != nullptr) rz0 = col_partials<NV>(part_rz0 + j * MAXP, P, sw);
  if (threadIdx.x < NV) {
    const double tol = fmax(rtol * sqrt(bn2 > 0.0 ? bn2 : zz), atol);   // zero right-hand side: relative to the start residual (k_pcg_begin)
    Scal* sc = scal + j;
    sc->tol2 = tol * tol;
    sc->bn2 = bn2;
    sc->zz = zz;
    sc->iters = 0;
    sc->first = 1;
    sc->done = (zz <= tol * tol) ? 1 : 0;
    sc->epoch = epoch;
    red->zz[j] = zz;
    red->bn[j] = bn2;
    if (part_rz0 != nullptr) red->rz[0][j] = rz0;
    if (mirror != nullptr) {
      mirror[j].bn2 = bn2;
      mirror_publish(mirror + j, zz, 0, sc->done, epoch);
    }
  }
}

// x += alpha p; r -= alpha Ap; z = D^-1 r (Jacobi: + r.z partials) or z0 = w D^-1 r (multigrid); (D^-1 r)^2 partials
template <int NV, bool AMG, bool DINV_PERCOL>
__global__ __launch_bounds__(TPB) void kb_update(int n, const BRed* __restrict__ red, int parity, Scal* __restrict__ scal,
                                                 double* __restrict__ part_rz, double* __restrict__ part_zz, double* __restrict__ x,
                                                 double* __restrict__ r, const double* __restrict__ p, const double* __restrict__ Ap,
                                                 const double* __restrict__ dinv, double w, double* __restrict__ z) {
  __shared__ double sw[4 * NV];
  constexpr int RPB = TPB / NV;
  const int j = threadIdx.x % NV, rl = threadIdx.x / NV;
  Scal* sc = scal + j;
  const double pAp = red->pAp[j], rz = red->rz[parity][j];
  bool active = sc->done == 0;
  if (active && !(pAp > 0.0)) {                          // breakdown (only on NaN / a non-SPD preconditioner)
    if (blockIdx.x == 0 && rl == 0) sc->done = 2;
    active = false;
  }
  if (active && blockIdx.x == 0 && rl == 0) { sc->iters += 1; sc->first = 0; }
  const double alpha = rz / pAp;
  double a_rz = 0.0, a_zz = 0.0;
  const int nrb = (n + RPB - 1) / RPB;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int row = rb * RPB + rl;
    if (row >= n || !active) continue;
    const size_t o = static_cast<size_t>(row) * NV + j;
    const double ri = r[o] - alpha * Ap[o];
    const double zi = (DINV_PERCOL ? dinv[o] : dinv[row]) * ri;
    x[o] += alpha * p[o];
    r[o] = ri;
    z[o] = AMG ? w * zi : zi;
    a_rz += ri * zi;
    a_zz += zi * zi;
  }
  if (!AMG) {
    const double t0 = block_colsum<NV>(a_rz, sw);
    if (rl == 0) part_rz[((parity ^ 1) * NV + j) * MAXP + blockIdx.x] = t0;
  }
  const double t1 = block_colsum<NV>(a_zz, sw);
  if (rl == 0) part_zz[j * MAXP + blockIdx.x] = t1;
}

// b[row, j] -= sum_q lift_val[q, j] * g[lift_bc[q], j]; thread = (lifted row, column)
template <int NV, int OPK>
__global__ void kb_lift(int nrows, const int32_t* __restrict__ rows, const int32_t* __restrict__ ptr, const int32_t* __restrict__ bc,
                        const BOp val, const double* __restrict__ g, double* __restrict__ b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / NV, j = t % NV;
  if (q >= nrows) return;
  double s = 0.0;
  for (int k = ptr[q]; k < ptr[q + 1]; ++k) s += op_value<OPK, NV>(val, static_cast<size_t>(k), j) * g[static_cast<size_t>(bc[k]) * NV + j];
  b[static_cast<size_t>(rows[q]) * NV + j] -= s;
}

// affine operators: D^-1 of every column, dinv[i, j] = 1 / (A_ii + d_j A1_ii)
template <int NV>
__global__ void kb_affine_dinv(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const BOp op,
                               double* __restrict__ dinv) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t / NV, j = t % NV;
  if (row >= n) return;
  double d = 0.0;
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
    if (colidx[k] == row) d = op_value<OP_AFFINE, NV>(op, static_cast<size_t>(k), j);
  dinv[static_cast<size_t>(row) * NV + j] = 1.0 / d;
}

// the part of the operator that scales with the swept conductivity, Dirichlet rows and columns removed
__global__ void k_zero_slots(int nq, const int32_t* __restrict__ slot, double* __restrict__ A, double* __restrict__ keep) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  if (keep) keep[q] = A[slot[q]];
  A[slot[q]] = 0.0;
}
__global__ void k_zero_rows(int nbc, const int32_t* __restrict__ dofs, const int32_t* __restrict__ rowptr, double* __restrict__ A) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nbc) return;
  for (int k = rowptr[dofs[q]]; k < rowptr[dofs[q] + 1]; ++k) A[k] = 0.0;
}

template <int NV>
__global__ void kb_set_bc(int nbc, const int32_t* __restrict__ dofs, const double* __restrict__ g, double* __restrict__ b,
                          double* __restrict__ u) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / NV, j = t % NV;
  if (q >= nbc) return;
  const size_t o = static_cast<size_t>(dofs[q]) * NV + j;
  b[o] = g[static_cast<size_t>(q) * NV + j];
  u[o] = g[static_cast<size_t>(q) * NV + j];
}

// Projection start vector (k_proj_* in hf_kernels.hpp) for NV interleaved columns: per column the partial sums of
// h_k = V_k . f and of the Gram column g_k = V_k . Fnew; layout part[(j * 2 PROJ_MT + 2k + s) * MAXP + workgroup]
template <int NV>
__global__ __launch_bounds__(TPB) void kb_proj_dots(int n, ProjVecs a, const double* __restrict__ f, const double* __restrict__ Fnew,
                                                    double* __restrict__ part) {
  __shared__ double sw[4 * NV];
  constexpr int RPB = TPB / NV;
  const int j = threadIdx.x % NV, rl = threadIdx.x / NV;
  double ah[PROJ_MH], ag[PROJ_MH];
#pragma unroll
  for (int k = 0; k < PROJ_MH; ++k) { ah[k] = 0.0; ag[k] = 0.0; }
  const int nrb = (n + RPB - 1) / RPB;
  // two row blocks per pass: the loads of both are in flight together (the stored vectors come from HBM)
  for (int rb = blockIdx.x; rb < nrb; rb += 2 * gridDim.x) {
    const int row = rb * RPB + rl, row2 = (rb + static_cast<int>(gridDim.x)) * RPB + rl;
    const bool one = row < n, two = rb + static_cast<int>(gridDim.x) < nrb && row2 < n;
    const size_t o = static_cast<size_t>(one ? row : 0) * NV + j, o2 = static_cast<size_t>(two ? row2 : 0) * NV + j;
    const double fi = one ? f[o] : 0.0, gi = (one && Fnew) ? Fnew[o] : 0.0;
    const double fj = two ? f[o2] : 0.0, gj = (two && Fnew) ? Fnew[o2] : 0.0;
    double v0[PROJ_MH], v1[PROJ_MH];
#pragma unroll
    for (int k = 0; k < PROJ_MH; ++k) {
      v0[k] = (k < a.m && one) ? a.V[k][o] : 0.0;
      v1[k] = (k < a.m && two) ? a.V[k][o2] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PROJ_MH; ++k) {
      ah[k] += v0[k] * fi;
      ag[k] += v0[k] * gi;
      ah[k] += v1[k] * fj;
      ag[k] += v1[k] * gj;
    }
  }
#pragma unroll
  for (int k = 0; k < PROJ_MH; ++k)
    if (k < a.m) {
      const double th = block_colsum<NV>(ah[k], sw), tg = block_colsum<NV>(ag[k], sw);
      if (rl == 0) {
        part[(static_cast<size_t>(j) * 2 * PROJ_MT + 2 * k) * MAXP + blockIdx.x] = th;
        part[(static_cast<size_t>(j) * 2 * PROJ_MT + 2 * k + 1) * MAXP + blockIdx.x] = tg;
      }
    }
}

template <int NV>
__global__ __launch_bounds__(TPB) void kb_proj_combine(int n, ProjVecs a, const double* __restrict__ alpha /* [NV][PROJ_MT + 1] */,
                                                       double* __restrict__ u) {
  const int j = threadIdx.x % NV;
  double c[PROJ_MH];
#pragma unroll
  for (int k = 0; k < PROJ_MH; ++k) c[k] = k < a.m ? alpha[j * (PROJ_MT + 1) + a.slot[k]] : 0.0;
  const size_t total = static_cast<size_t>(n) * NV, stride = static_cast<size_t>(gridDim.x) * TPB;
  for (size_t q = static_cast<size_t>(blockIdx.x) * TPB + threadIdx.x; q < total; q += 2 * stride) {   // two entries per pass
    const size_t q2 = q + stride;            // TPB is a multiple of NV: a thread stays with its column
    const bool two = q2 < total;
    double v0[PROJ_MH], v1[PROJ_MH];
#pragma unroll
    for (int k = 0; k < PROJ_MH; ++k) {
      v0[k] = k < a.m ? a.V[k][q] : 0.0;
      v1[k] = (k < a.m && two) ? a.V[k][q2] : 0.0;
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < PROJ_MH; ++k) { s0 += c[k] * v0[k]; s1 += c[k] * v1[k]; }
    u[q] = s0;
    if (two) u[q2] = s1;
  }
}

template <int NV>
__global__ void kb_zero_bc(int nbc, const int32_t* __restrict__ dofs, double* __restrict__ v) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nbc * NV) v[static_cast<size_t>(dofs[t / NV]) * NV + t % NV] = 0.0;
}

template <int NV>
__global__ void kb_gather(int ns, const int32_t* __restrict__ idx, const double* __restrict__ u, double* __restrict__ out /* [j][ns] */) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / NV, j = t % NV;
  if (q < ns) out[static_cast<size_t>(j) * ns + q] = u[static_cast<size_t>(idx[q]) * NV + j];
}

// column j of an interleaved array <- / -> a plain array
__global__ void kb_put_column(size_t n, int nv, int j, const double* __restrict__ src, double* __restrict__ dst) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i * nv + j] = src[i];
}
__global__ void kb_get_column(size_t n, int nv, int j, const double* __restrict__ src, double* __restrict__ dst) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] = src[i * nv + j];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
void free_batch_state(hf_ctx::Batch& B) {
  dev_free(&B.A); dev_free(&B.A1); dev_free(&B.dinv); dev_free(&B.lift_val); dev_free(&B.lift1); dev_free(&B.g);
  dev_free(&B.u); dev_free(&B.uprev); dev_free(&B.ustart); dev_free(&B.b); dev_free(&B.r); dev_free(&B.p); dev_free(&B.Ap);
  dev_free(&B.z); dev_free(&B.z2); dev_free(&B.tmp);
  dev_free(&B.part_pAp); dev_free(&B.part_rz); dev_free(&B.part_zz); dev_free(&B.part_bn); dev_free(&B.scal); dev_free(&B.red);
  for (auto& v : B.pV) dev_free(&v);
  for (auto& v : B.pF) dev_free(&v);
  dev_free(&B.pG); dev_free(&B.palpha); dev_free(&B.ppart);
  for (auto& L : B.lev) { dev_free(&L.x); dev_free(&L.cat); if (L.own_b) dev_free(&L.b); }
  B.lev.clear();
  if (B.h_scal) { (void)hipHostFree(B.h_scal); B.h_scal = nullptr; }
  if (B.h_mirror) { (void)hipHostFree(B.h_mirror); B.h_mirror = B.d_mirror = nullptr; }
  B.nv = 0;
  B.sysA = B.sysDinv = nullptr;
}

void free_batch(hf_ctx* ctx) { free_batch_state(ctx->batch); free_batch_state(ctx->fluxnb); }

void free_batch_cols(hf_ctx* ctx) {
  hf_ctx::BatchCols& T = ctx->bcols;
  dev_free(&T.ptr); dev_free(&T.dict); dev_free(&T.own); dev_free(&T.id);
  T = hf_ctx::BatchCols();
}

size_t batch_lds_bytes(int cap_nnz, int cap_dict, int nv, bool affine) {
  return (static_cast<size_t>(affine ? 2 : 1) * cap_nnz + static_cast<size_t>(cap_dict) * nv) * 8 + ((static_cast<size_t>(cap_nnz) * 2 + 7) & ~static_cast<size_t>(7));
}

// Tables of kb_spmv_lds for `nv` columns.  Rows per chunk: two passes of the workgroup (2 * KB_BT / nv rows: 128 for 8
// columns), one on meshes too small to give every workgroup a chunk that way; fewer if the chunk's operands would not
// fit the 64-KB LDS window of a launch.  Measured on the stock mesh (191 k rows, 8 columns, one batched multigrid-PCG
// iteration): 64 / 128 / 192 / 256 rows per chunk -> 140 / 141 / 152 / 157 us - small chunks keep four workgroups per
// CU resident (30 KB of LDS each) and their staging overlaps other workgroups' products.  HEATFLOW_BATCH_LDS=0 keeps
// the gather kernel (A/B); HEATFLOW_BATCH_RPC overrides the rows per chunk.
int ensure_batch_cols(hf_ctx* ctx, int nv) {
  static const bool enabled = !(std::getenv("HEATFLOW_BATCH_LDS") && std::getenv("HEATFLOW_BATCH_LDS")[0] == '0');
  hf_ctx::BatchCols& T = ctx->bcols;
  if (!enabled) { free_batch_cols(ctx); return HF_OK; }
  if (T.nv == nv) return HF_OK;
  free_batch_cols(ctx);
  const int rpp = KB_BT / nv;
  int m = static_cast<long long>(ctx->n) >= static_cast<long long>(rpp) * MAXP ? 2 : 1;
  if (const char* e = std::getenv("HEATFLOW_BATCH_RPC")) m = std::max(1, std::atoi(e) / rpp);
  for (; m >= 1; --m) {
    const int rpc = m * rpp;
    ColDict D;
    if (!build_coldict(ctx->h_rowptr, ctx->h_colidx, ctx->n, rpc, D)) continue;
    const int nch = (ctx->n + rpc - 1) / rpc;
    int cap = 0;
    for (int c = 0; c < nch; ++c) cap = std::max(cap, ctx->h_rowptr[std::min<int64_t>(ctx->n, (c + 1LL) * rpc)] - ctx->h_rowptr[static_cast<size_t>(c) * rpc]);
    cap = (cap + 1) & ~1;
    if (batch_lds_bytes(cap, D.max_dict, nv, true) > 64 * 1024) continue;
    // position of each chunk's first row in its list; the rows of a chunk must be a contiguous run of it (every row
    // stores its diagonal: true for the P1 pattern) - otherwise the staged kernel is not used
    std::vector<int32_t> own(nch);
    bool ok = true;
    for (int c = 0; c < nch && ok; ++c) {
      const int32_t r0 = c * rpc, r1 = std::min<int32_t>(ctx->n, r0 + rpc);
      const int32_t* lo = D.dict.data() + D.ptr[c];
      const int32_t* hi = D.dict.data() + D.ptr[c + 1];
      const int32_t* at = std::lower_bound(lo, hi, r0);
      ok = hi - at >= r1 - r0 && at[0] == r0 && at[r1 - r0 - 1] == r1 - 1;
      own[c] = static_cast<int32_t>(at - lo);
    }
    if (!ok) return HF_OK;
    HF_TRY(dev_alloc(ctx, &T.ptr, D.ptr.size()));
    HF_TRY(dev_alloc(ctx, &T.dict, D.dict.size()));
    HF_TRY(dev_alloc(ctx, &T.own, own.size()));
    HF_TRY(dev_alloc(ctx, &T.id, D.id.size()));
    HF_HIP(copy_sync(ctx, T.ptr, D.ptr.data(), sizeof(int32_t) * D.ptr.size(), hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, T.dict, D.dict.data(), sizeof(int32_t) * D.dict.size(), hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, T.own, own.data(), sizeof(int32_t) * own.size(), hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, T.id, D.id.data(), sizeof(uint16_t) * D.id.size(), hipMemcpyHostToDevice));
    T.nv = nv; T.rpc = rpc; T.nchunks = nch; T.cap_nnz = cap; T.cap_dict = D.max_dict;
    if (std::getenv("HEATFLOW_DEBUG"))
      std::fprintf(stderr, "[batch] nv %d: %d chunks of %d rows, <= %d nonzeros and %d list entries per chunk (%.2f per row), LDS %zu B (affine)\n", nv, nch, rpc, cap,
                   D.max_dict, static_cast<double>(D.dict.size()) / ctx->n, batch_lds_bytes(cap, D.max_dict, nv, true));
    return HF_OK;
  }
  return HF_OK;     // no chunk size fits: the gather kernel stays
}

template <int NV, int VMODE, typename VT>
void blaunch_csr_t(hf_ctx* c, const DevCsr& m, const VT* val, const double* x, double* y) {
  const double avg = m.nrow ? static_cast<double>(m.nnz) / m.nrow : 1.0;
  // Mapping lane = (entry lane, group of CPL columns).  CPL = NV is the first kernel of this file (a lane per entry, NV
  // accumulators, NV * log2(lanes) shuffles per row); CPL < NV trades shuffles for entries per lane.  Defaults from the traces
  // of the stock hierarchy (profiles/r03_batch_csr_mappings.txt); HEATFLOW_BATCH_CPL / HEATFLOW_BATCH_PER_LANE override.
  //   8 / 16 columns: four columns per lane, except where rows are few and long (< 1024 rows: the chain of dependent loads
  //   decides, a lane per entry is shorter) and, for 8 columns, the small operators of middling row length; entries per lane:
  //   8 (16 columns) or 4 (8 columns) where rows are many (>= 16384: fewer, longer-working lanes), 2.5 where they are few;
  //   2 / 4 columns: a lane per entry.
  static const int cpl_env = std::getenv("HEATFLOW_BATCH_CPL") ? std::atoi(std::getenv("HEATFLOW_BATCH_CPL")) : 0;
  static const double per_lane_env = std::getenv("HEATFLOW_BATCH_PER_LANE") ? std::atof(std::getenv("HEATFLOW_BATCH_PER_LANE")) : 0.0;
  int cpl = NV;
  if (NV >= 8 && m.nrow >= 1024 && !(NV == 8 && m.nrow < 4096 && avg < 64.0)) cpl = 4;
  if (cpl_env > 0) cpl = cpl_env;
  double per_lane = cpl == NV ? 1.25 : (m.nrow >= 16384 ? (NV >= 16 ? 8.0 : 4.0) : 2.5);
  if (per_lane_env > 0.0) per_lane = per_lane_env;
  if (cpl > NV) cpl = NV;
  while (NV % cpl) --cpl;
  const int nvl = NV / cpl, elmax = 64 / nvl;
  int el = cpl == NV ? 2 : 1;
  while (el < elmax && avg > per_lane * el) el *= 2;
  const long long threads = static_cast<long long>(m.nrow) * el * nvl;
  const int grid = static_cast<int>(std::max(1LL, std::min<long long>((threads + TPB - 1) / TPB, 4096)));
#define HF_RC(C, L) hipLaunchKernelGGL((kb_csr_rc<NV, (C <= NV ? C : NV), ((L) * (NV / (C <= NV ? C : NV)) <= 64 ? (L) : 64 / (NV / (C <= NV ? C : NV))), VMODE, VT>), \
                                       dim3(grid), dim3(TPB), 0, c->stream, m.nrow, m.ptr, m.idx, val, x, y)
#define HF_RC_EL(C)                                                                      \
  switch (el) {                                                                          \
    case 1: HF_RC(C, 1); break;                                                          \
    case 2: HF_RC(C, 2); break;                                                          \
    case 4: HF_RC(C, 4); break;                                                          \
    case 8: HF_RC(C, 8); break;                                                          \
    case 16: HF_RC(C, 16); break;                                                        \
    case 32: HF_RC(C, 32); break;                                                        \
    default: HF_RC(C, 64); break;                                                        \
  }
  if (cpl == NV) {
    // a lane per entry with NV accumulators: the original kernel (its two-entries-per-pass loop suits short dependent chains)
#define HF_BCSR(L) hipLaunchKernelGGL((kb_csr<NV, L, VMODE, VT>), dim3(grid), dim3(TPB), 0, c->stream, m.nrow, m.ptr, m.idx, val, x, y)
    switch (el) {
      case 2: HF_BCSR(2); break;
      case 4: HF_BCSR(4); break;
      case 8: HF_BCSR(8); break;
      case 16: HF_BCSR(16); break;
      case 32: HF_BCSR(32); break;
      default: HF_BCSR(64); break;
    }
#undef HF_BCSR
    return;
  }
  switch (cpl) {
    case 1: HF_RC_EL(1); break;
    case 2: HF_RC_EL(2); break;
    case 4: HF_RC_EL(4); break;
    default: HF_RC_EL(8); break;
  }
#undef HF_RC_EL
#undef HF_RC
}

template <int NV, int VMODE>
void blaunch_csr(hf_ctx* c, const DevCsr& m, const double* x, double* y) {
  if (m.valf != nullptr) blaunch_csr_t<NV, VMODE, float>(c, m, m.valf, x, y);
  else blaunch_csr_t<NV, VMODE, double>(c, m, m.val, x, y);
}

template <int NV, int OPK>
struct BatchOps {
  static constexpr bool DPC = OPK != OP_SHARED;       // D^-1 stored per column
  static BOp Aop(hf_ctx* c) {
    const hf_ctx::Batch& B = c->batch;
    BOp o{};
    o.v0 = B.sysA ? B.sysA : (OPK == OP_PERCOL ? B.A : c->d_A);
    o.v1 = B.A1;
    for (int j = 0; j < NV_MAX; ++j) o.delta[j] = B.delta[j];
    return o;
  }
  static BOp Liftop(hf_ctx* c) {
    const hf_ctx::Batch& B = c->batch;
    BOp o{};
    o.v0 = OPK == OP_PERCOL ? B.lift_val : c->d_lift_val;
    o.v1 = B.lift1;
    for (int j = 0; j < NV_MAX; ++j) o.delta[j] = B.delta[j];
    return o;
  }
  static const double* Avals(hf_ctx* c) { return c->d_A; }   // tag: "the system operator" (any pointer but d_M)
  static const double* Dinv(hf_ctx* c) { return c->batch.sysDinv ? c->batch.sysDinv : (DPC ? c->batch.dinv : c->d_dinv); }

  template <int MODE>
  static void spmv(hf_ctx* c, const double* vals, const double* x, double* y, double* part0 = nullptr, const double* bvec = nullptr,
                   double* pvec = nullptr, double* part1 = nullptr, double* part2 = nullptr, double w = 0.0, int parity = 0,
                   hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr /* hf_set_profile: the launch carries its own start / stop events */) {
    hf_ctx::Batch& B = c->batch;
    if (B.lds && OPK != OP_PERCOL) {    // chunk operands staged in LDS, 16-bit column positions
      const hf_ctx::BatchCols& T = c->bcols;
      const BComp comp{T.ptr, T.dict, T.own, T.id, T.rpc, T.nchunks, T.cap_nnz, T.cap_dict};
      int grid = std::min(T.nchunks, B.Pb);
      if (grid >= 64) grid &= ~7;
      if (vals == c->d_M) {
        BOp m{};
        m.v0 = c->d_M;
        hipLaunchKernelGGL((kb_spmv_lds<MODE, NV, OP_SHARED>), dim3(grid), dim3(KB_BT), batch_lds_bytes(T.cap_nnz, T.cap_dict, NV, false), c->stream,
                           c->n, c->d_rowptr, m, x, y, B.scal, part0, bvec, c->d_dinv, pvec, part1, part2, w, B.red, parity, comp, B.Pb);
      } else {
        constexpr int OPL = OPK == OP_PERCOL ? OP_SHARED : OPK;     // (never instantiated for per-column values)
        if (ev0 != nullptr)
          hipExtLaunchKernelGGL((kb_spmv_lds<MODE, NV, OPL>), dim3(grid), dim3(KB_BT), static_cast<std::uint32_t>(batch_lds_bytes(T.cap_nnz, T.cap_dict, NV, OPL == OP_AFFINE)),
                                c->stream, ev0, ev1, 0u, c->n, c->d_rowptr, Aop(c), x, y, B.scal, part0, bvec, Dinv(c), pvec, part1, part2, w, B.red, parity, comp, B.Pb);
        else
          hipLaunchKernelGGL((kb_spmv_lds<MODE, NV, OPL>), dim3(grid), dim3(KB_BT), batch_lds_bytes(T.cap_nnz, T.cap_dict, NV, OPL == OP_AFFINE), c->stream,
                             c->n, c->d_rowptr, Aop(c), x, y, B.scal, part0, bvec, Dinv(c), pvec, part1, part2, w, B.red, parity, comp, B.Pb);
      }
      return;
    }
    // M is always shared (rho_c does not change inside a batch): MODE 0 / 8 on M use the shared-value kernel
    if (vals == c->d_M) {
      BOp m{};
      m.v0 = c->d_M;
      hipLaunchKernelGGL((kb_spmv<MODE, NV, OP_SHARED>), dim3(B.Pb), dim3(KB_BT), 0, c->stream, c->n, c->d_rowptr, c->d_colidx, m, x, y,
                         B.scal, part0, bvec, c->d_dinv, pvec, part1, part2, w, B.red, parity);
    } else if (ev0 != nullptr) {
      hipExtLaunchKernelGGL((kb_spmv<MODE, NV, OPK>), dim3(B.Pb), dim3(KB_BT), 0u, c->stream, ev0, ev1, 0u, c->n, c->d_rowptr, c->d_colidx, Aop(c), x, y,
                            B.scal, part0, bvec, Dinv(c), pvec, part1, part2, w, B.red, parity);
    } else {
      hipLaunchKernelGGL((kb_spmv<MODE, NV, OPK>), dim3(B.Pb), dim3(KB_BT), 0, c->stream, c->n, c->d_rowptr, c->d_colidx, Aop(c), x, y,
                         B.scal, part0, bvec, Dinv(c), pvec, part1, part2, w, B.red, parity);
    }
  }

  static void reduce(hf_ctx* c, const double* part_a, double* out_a, const double* part_b = nullptr, double* out_b = nullptr,
                     Scal* test = nullptr) {
    hipLaunchKernelGGL(kb_reduce, dim3(part_b ? 2 * NV : NV), dim3(TPB), 0, c->stream, c->batch.Pb, NV, part_a, out_a, part_b, out_b, test,
                       test ? c->batch.d_mirror : static_cast<ScalMirror*>(nullptr));
  }

  // z = B r for every column: the V(1,1) cycle of hf_solver.hpp's vcycle() on interleaved vectors
  static void vcycle(hf_ctx* c, int out_slot) {
    hf_ctx::Batch& B = c->batch;
    const int nl = static_cast<int>(c->amg.size());
    const double w0 = c->amg[0].omega;
    double* rz_out = B.part_rz + static_cast<size_t>(out_slot) * NV * MAXP;
    if (nl == 1) {
      spmv<4>(c, Avals(c), B.z, B.z2, rz_out, B.r, nullptr, nullptr, nullptr, w0);
      reduce(c, rz_out, B.red->rz[out_slot], B.part_zz, B.red->zz);
      return;
    }
    spmv<3>(c, Avals(c), B.z, B.tmp, nullptr, B.r);
    blaunch_csr<NV, 0>(c, c->amg[0].R, B.tmp, B.lev[1].b);
    for (int l = 1; l + 1 < nl; ++l) blaunch_csr<NV, 0>(c, c->amg[l].Rt, B.lev[l].b, B.lev[l + 1].b);
    {
      const DevLevel& Lc = c->amg[nl - 1];
      if (c->coarse_n > 0) {
        const int g = std::max(1, std::min((Lc.n + 3) / 4, 1024));
        static const int dense_cpl = std::getenv("HEATFLOW_BATCH_DENSE_CPL") ? std::atoi(std::getenv("HEATFLOW_BATCH_DENSE_CPL")) : NV;   // measured: a lane per column wins (few rows)
        constexpr int DC = NV >= 4 ? 4 : NV;
        if (dense_cpl < NV && c->d_coarse_inv_f != nullptr)
          hipLaunchKernelGGL((kb_dense_rc<NV, DC, float>), dim3(g), dim3(TPB), 0, c->stream, Lc.n, c->coarse_ld, c->d_coarse_inv_f,
                             B.lev[nl - 1].b, B.lev[nl - 1].res);
        else if (c->d_coarse_inv_f != nullptr)
          hipLaunchKernelGGL((kb_dense<NV, float>), dim3(g), dim3(TPB), 0, c->stream, Lc.n, c->coarse_ld, c->d_coarse_inv_f,
                             B.lev[nl - 1].b, B.lev[nl - 1].res);
        else
          hipLaunchKernelGGL((kb_dense<NV, double>), dim3(g), dim3(TPB), 0, c->stream, Lc.n, c->coarse_ld, c->d_coarse_inv,
                             B.lev[nl - 1].b, B.lev[nl - 1].res);
      } else {
        const int g = std::max(1, std::min((Lc.n * NV + TPB - 1) / TPB, 1024));
        hipLaunchKernelGGL((kb_scale<NV>), dim3(g), dim3(TPB), 0, c->stream, Lc.n, Lc.omega, Lc.dinv, B.lev[nl - 1].b, B.lev[nl - 1].res);
      }
    }
    for (int l = nl - 2; l >= 1; --l) blaunch_csr<NV, 0>(c, c->amg[l].GP, B.lev[l].cat, B.lev[l].res);
    blaunch_csr<NV, 1>(c, c->amg[0].P, B.lev[1].res, B.z);
    spmv<4>(c, Avals(c), B.z, B.z2, rz_out, B.r, nullptr, nullptr, nullptr, w0);
    reduce(c, rz_out, B.red->rz[out_slot], B.part_zz, B.red->zz);   // r.z of this cycle and (D^-1 r)^2 of the last update
  }

  // `part` (multigrid only): the whole iteration, up to the reduction that tests and publishes, or the V-cycle after it
  static void iteration(hf_ctx* c, bool use_amg, int parity, CyclePart part = CYCLE_ALL) {
    if (part == CYCLE_REST) { vcycle(c, parity ^ 1); return; }
    hf_ctx::Batch& B = c->batch;
    // hf_set_profile: the iteration heads of a solve (its first PROF_PAIRS) carry event pairs, harvested when the solve ends
    const bool timed = c->prof && c->prof_used < PROF_PAIRS;
    hipEvent_t e0 = timed ? c->prof_ev[2 * c->prof_used] : nullptr, e1 = timed ? c->prof_ev[2 * c->prof_used + 1] : nullptr;
    if (timed) c->prof_used++;
    if (use_amg) {
      spmv<9>(c, Avals(c), B.z2, B.Ap, B.part_pAp, nullptr, B.p, B.part_rz, B.part_zz, 0.0, parity, e0, e1);
      reduce(c, B.part_pAp, B.red->pAp);
      hipLaunchKernelGGL((kb_update<NV, true, DPC>), dim3(B.Pb), dim3(TPB), 0, c->stream, c->n, B.red, parity, B.scal,
                         B.part_rz, B.part_zz, B.u, B.r, B.p, B.Ap, Dinv(c), c->amg[0].omega, B.z);
      // (D^-1 r)^2 of the new iterates: columns that have converged are marked done here, so that the V-cycle below
      // does no work for them (it used to be found out by the next iteration head, one cycle too late)
      reduce(c, B.part_zz, B.red->zz, nullptr, nullptr, B.scal);
      if (part == CYCLE_ALL) vcycle(c, parity ^ 1);
    } else {
      spmv<9>(c, Avals(c), B.z, B.Ap, B.part_pAp, nullptr, B.p, B.part_rz, B.part_zz, 0.0, parity, e0, e1);
      reduce(c, B.part_pAp, B.red->pAp);
      hipLaunchKernelGGL((kb_update<NV, false, DPC>), dim3(B.Pb), dim3(TPB), 0, c->stream, c->n, B.red, parity, B.scal,
                         B.part_rz, B.part_zz, B.u, B.r, B.p, B.Ap, Dinv(c), 0.0, B.z);
      reduce(c, B.part_rz + static_cast<size_t>(parity ^ 1) * NV * MAXP, B.red->rz[parity ^ 1], B.part_zz, B.red->zz);
    }
  }

  static int read_scal(hf_ctx* ctx, bool* all_done, int* max_iters, bool* breakdown) {
    hf_ctx::Batch& B = ctx->batch;
    HF_HIP(hipMemcpyAsync(B.h_scal, B.scal, sizeof(Scal) * NV, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    *all_done = true; *max_iters = 0; *breakdown = false;
    for (int j = 0; j < NV; ++j) {
      if (!B.h_scal[j].done) *all_done = false;
      if (B.h_scal[j].done == 2) *breakdown = true;
      *max_iters = std::max(*max_iters, B.h_scal[j].iters);
    }
    return HF_OK;
  }

  // Wait on host memory until every column's iterate after update k has been tested or the column has ended
  // (wait_tested of the single-column loop, per column); fills B.h_scal from the mirrors
  static int wait_tested(hf_ctx* ctx, int k, bool* all_done, int* max_iters, bool* breakdown) {
    static const double limit_s = std::getenv("HEATFLOW_POLL_TIMEOUT_S") ? std::atof(std::getenv("HEATFLOW_POLL_TIMEOUT_S")) : 60.0;
    hf_ctx::Batch& B = ctx->batch;
    const unsigned ep = B.epoch;
    const auto t0 = std::chrono::steady_clock::now();
    auto last_query = t0;
    auto reached = [&]() {
      for (int j = 0; j < NV; ++j)
        if (mirror_tested(B.h_mirror + j, ep) < k && mirror_done(B.h_mirror + j, ep) == 0) return false;
      return true;
    };
    for (unsigned spin = 1; !reached(); ++spin) {
      cpu_relax();
      if ((spin & 0x3ff) != 0) continue;
      const auto now = std::chrono::steady_clock::now();
      if (now - last_query < std::chrono::milliseconds(2)) continue;   // the runtime lock hipStreamQuery takes is the one other sessions' launch threads need
      last_query = now;
      if (hipStreamQuery(ctx->stream) == hipSuccess) {
        if (reached()) break;
        return fail(ctx, HF_ERR_HIP, "batched PCG progress: stream drained before the test of iteration %d ran (%s)", k, hipGetErrorString(hipGetLastError()));
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
        return fail(ctx, HF_ERR_HIP, "batched PCG progress: no convergence test within %.0f s (waiting for iteration %d)", limit_s, k);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *all_done = true; *max_iters = 0; *breakdown = false;
    for (int j = 0; j < NV; ++j) {
      const ScalMirror& m = B.h_mirror[j];
      B.h_scal[j].iters = m.iters; B.h_scal[j].zz = m.zz; B.h_scal[j].bn2 = m.bn2;
      B.h_scal[j].done = mirror_done(B.h_mirror + j, ep);
      if (!B.h_scal[j].done) *all_done = false;
      if (B.h_scal[j].done == 2) *breakdown = true;
      *max_iters = std::max(*max_iters, B.h_scal[j].iters);
    }
    return HF_OK;
  }

  // PCG on all columns, started from B.u; iteration counts / residuals are left in B.h_scal
  static int pcg(hf_ctx* ctx, bool use_amg, double rtol, double atol, int max_it) {
    if (!ctx->prof) return pcg_run(ctx, use_amg, rtol, atol, max_it);
    ctx->prof_used = 0;
    const int rc = pcg_run(ctx, use_amg, rtol, atol, max_it);
    (void)hipStreamSynchronize(ctx->stream);
    int most = 0;                                     // iteration heads that really ran: those of the slowest column
    for (int j = 0; j < NV; ++j) most = std::max(most, ctx->batch.h_scal[j].iters);
    for (int k = 0; k < ctx->prof_used && k < most; ++k) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->prof_ev[2 * k], ctx->prof_ev[2 * k + 1]) == hipSuccess) { ctx->prof_spmv_ms += ms; ctx->prof_spmv_n += 1; }
    }
    ctx->prof_used = 0;
    return rc;
  }

  static int pcg_run(hf_ctx* ctx, bool use_amg, double rtol, double atol, int max_it) {
    hf_ctx::Batch& B = ctx->batch;
    // a new epoch instead of a reset of the mirrors: launches of the previous solve's blind burst may still be queued
    // (they publish nothing once their column has converged, and whatever reaches the mirrors late carries the old epoch)
    B.epoch += 1;
    HF_HIP(hipMemsetAsync(B.scal, 0, sizeof(Scal) * NV, ctx->stream));
    if (!use_amg) {
      spmv<2>(ctx, Avals(ctx), B.u, B.r, B.part_rz, B.b, B.z, B.part_zz, B.part_bn, 0.0);
      hipLaunchKernelGGL((kb_begin<NV>), dim3(1), dim3(TPB), 0, ctx->stream, B.Pb, rtol, atol, B.part_zz, B.part_bn, B.part_rz, B.scal, B.red, B.d_mirror, B.epoch);
    } else {
      spmv<5>(ctx, Avals(ctx), B.u, B.r, nullptr, B.b, B.z, B.part_zz, B.part_bn, ctx->amg[0].omega);
      hipLaunchKernelGGL((kb_begin<NV>), dim3(1), dim3(TPB), 0, ctx->stream, B.Pb, rtol, atol, B.part_zz, B.part_bn,
                         static_cast<const double*>(nullptr), B.scal, B.red, B.d_mirror, B.epoch);
      vcycle(ctx, 0);
    }
    HF_HIP(hipGetLastError());
    bool all_done = false, breakdown = false;
    int launched = 0, iters = 0;
    if (use_amg && B.h_mirror != nullptr && poll_enabled()) {
      // multigrid iterations queued one test ahead (see pcg_solve in hf_solver.hpp): the reduction that marks converged
      // columns publishes every column's outcome, and the V-cycle that follows it gives the host the time to queue
      // the next iteration
      if (B.pred_iters <= 0) {
        HF_TRY(wait_tested(ctx, 0, &all_done, &iters, &breakdown));
        if (all_done) return HF_OK;
      }
      // after the blind burst the V-cycle of an iteration is held back until its test says that a column still needs it
      // (pcg_solve in hf_solver.hpp; HEATFLOW_HOLD_BACK=0: whole iterations)
      static const bool hold_back = !(std::getenv("HEATFLOW_HOLD_BACK") && std::getenv("HEATFLOW_HOLD_BACK")[0] == '0');
      int burst = std::max(1, std::min(max_it, B.pred_iters - 2));
      bool rest_pending = false;
      while (true) {
        if (rest_pending) iteration(ctx, true, (launched - 1) & 1, CYCLE_REST);
        const bool split = hold_back && burst == 1 && launched > 0;
        for (int k = 0; k < burst; ++k) iteration(ctx, true, (launched + k) & 1, split ? CYCLE_FIRST : CYCLE_ALL);
        rest_pending = split;
        launched += burst;
        HF_HIP(hipGetLastError());
        HF_TRY(wait_tested(ctx, launched, &all_done, &iters, &breakdown));
        if (all_done || breakdown || launched >= max_it) break;
        burst = 1;
      }
      B.pred_iters = iters;
      if (breakdown) return fail(ctx, HF_ERR_NOCONV, "batched PCG breakdown (p.Ap <= 0) in at least one column");
      if (!all_done) return fail(ctx, HF_ERR_NOCONV, "batched PCG not converged in %d iterations", launched);
      return HF_OK;
    }
    if (B.pred_iters <= 0) {
      HF_TRY(read_scal(ctx, &all_done, &iters, &breakdown));
      if (all_done) return HF_OK;
    }
    int burst = std::max(1, std::min(max_it, B.pred_iters > 0 ? B.pred_iters : (use_amg ? 8 : 32)));
    while (true) {
      for (int k = 0; k < burst; ++k) iteration(ctx, use_amg, (launched + k) & 1);
      launched += burst;
      HF_HIP(hipGetLastError());
      HF_TRY(read_scal(ctx, &all_done, &iters, &breakdown));
      if (all_done || launched >= max_it) break;
      burst = std::max(1, std::min(std::max(use_amg ? 1 : 8, launched / 8), max_it - launched));
    }
    B.pred_iters = iters;
    if (breakdown) return fail(ctx, HF_ERR_NOCONV, "batched PCG breakdown (p.Ap <= 0) in at least one column");
    if (!all_done) return fail(ctx, HF_ERR_NOCONV, "batched PCG not converged in %d iterations", launched);
    return HF_OK;
  }

  static ProjVecs proj_active(const hf_ctx::Batch& B) {
    ProjVecs a{};
    a.m = 0;
    for (int k = 0; k < PROJ_MH; ++k)
      if (B.pused[k]) { a.V[a.m] = B.pV[k]; a.slot[a.m] = k; ++a.m; }
    return a;
  }

  // one time step of all columns to the boundary values g_dev (n_bc x NV, interleaved, on the device).  Start vector:
  // per column the A-norm projection of the new solution on the span of its last solutions (kind 3 of the
  // single-column loop, without the boundary responses)
  static int step(hf_ctx* ctx, const double* g_dev, double rtol, double atol, int max_it) {
    hf_ctx::Batch& B = ctx->batch;
    const int nb = ctx->nbc;
    spmv<0>(ctx, ctx->d_M, B.u, B.b);
    if (nb > 0) {
      if (ctx->nlift_rows > 0) {
        const int thr = ctx->nlift_rows * NV;
        hipLaunchKernelGGL((kb_lift<NV, OPK>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ctx->nlift_rows, ctx->d_lift_rows,
                           ctx->d_lift_ptr, ctx->d_lift_bc, Liftop(ctx), g_dev, B.b);
      }
    }
    const ProjVecs act = proj_active(B);
    // set_bc once: after the combination when there is one (the basis vectors are zero on the Dirichlet rows)
    if (nb > 0 && act.m == 0)
      hipLaunchKernelGGL((kb_set_bc<NV>), dim3((nb * NV + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, g_dev, B.b, B.u);
    if (act.m > 0) {
      hipLaunchKernelGGL((kb_proj_dots<NV>), dim3(B.Pb), dim3(TPB), 0, ctx->stream, ctx->n, act, B.b,
                         B.ppending >= 0 ? B.pF[B.ppending] : static_cast<const double*>(nullptr), B.ppart);
      hipLaunchKernelGGL(k_proj_solve, dim3(NV), dim3(PROJ_SOLVE_T), 0, ctx->stream, B.Pb, act, B.ppending, 1, B.ppart, B.pG, B.palpha);
      B.ppending = -1;
      hipLaunchKernelGGL((kb_proj_combine<NV>), dim3(B.Pb), dim3(TPB), 0, ctx->stream, ctx->n, act, B.palpha, B.u);
      if (nb > 0)
        hipLaunchKernelGGL((kb_set_bc<NV>), dim3((nb * NV + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, g_dev, B.b, B.u);
    }
    const bool use_amg = ctx->precond == 1 && ctx->amg_ready;
    const int rc = pcg(ctx, use_amg, rtol, atol, max_it);
    if (rc == HF_OK) {   // (solution with zeroed Dirichlet entries, right-hand side) joins every column's basis
      const int slot = B.pnext;
      hipLaunchKernelGGL(k_copy2, dim3(1024), dim3(TPB), 0, ctx->stream, ctx->n * NV, B.u, B.pV[slot], B.b, B.pF[slot]);
      if (nb > 0)
        hipLaunchKernelGGL((kb_zero_bc<NV>), dim3((nb * NV + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, B.pV[slot]);
      B.pused[slot] = true;
      B.ppending = slot;
      B.pnext = (slot + 1) % PROJ_MH;
    }
    return rc;
  }
};

// dispatch over (nv, per-column operator)
template <typename F>
int batch_dispatch(hf_ctx* ctx, F&& f) {
  const hf_ctx::Batch& B = ctx->batch;
  switch (B.nv * 4 + B.opk) {
    case 8: return f(BatchOps<2, OP_SHARED>());
    case 9: return f(BatchOps<2, OP_PERCOL>());
    case 10: return f(BatchOps<2, OP_AFFINE>());
    case 16: return f(BatchOps<4, OP_SHARED>());
    case 17: return f(BatchOps<4, OP_PERCOL>());
    case 18: return f(BatchOps<4, OP_AFFINE>());
    case 32: return f(BatchOps<8, OP_SHARED>());
    case 33: return f(BatchOps<8, OP_PERCOL>());
    case 34: return f(BatchOps<8, OP_AFFINE>());
    case 64: return f(BatchOps<16, OP_SHARED>());
    case 65: return f(BatchOps<16, OP_PERCOL>());
    case 66: return f(BatchOps<16, OP_AFFINE>());
    default: return fail(ctx, HF_ERR_STATE, "no batch is open (hf_batch_begin)");
  }
}

}  // namespace
