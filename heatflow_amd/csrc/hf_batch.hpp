// Part of libheatflow_hip.so (see heatflow_hip.hip): batched time loop - NV sweep points advance together.
//
// Sweep points on one mesh share the sparsity pattern, the mass matrix, the Dirichlet set and (with a frozen
// hierarchy) every coarse operator of the multigrid preconditioner; they differ in the boundary values (fwhm
// sweeps: even the fine operator A is shared) and / or in the values of A (kappa sweeps).  Reference: the grid of
// parameter_sweep.py:195-235 and the kappa list of sweep_test.py:47-52, which the reference farms out as
// independent runs.  Here NV points (2, 4 or 8) are the columns of one multi-vector PCG:
//   * every vector is stored interleaved, x[i * NV + j] = entry i of column j, so a gather of x[col] fetches the
//     NV columns as one contiguous 8*NV-byte segment and every index / shared value is read once for NV products;
//   * the fine operator is either shared (one value per nonzero) or per column (values interleaved like vectors);
//   * each column keeps its own PCG scalars (alpha, beta, tolerance, iteration count, done flag) on the device; a
//     converged column freezes (its x and r are no longer touched) while the others finish;
//   * at stock mesh sizes, where a single run is launch-bound, NV columns per launch cost about one.
// The algorithm per column is exactly hf_step's (same kernels' arithmetic, same stopping rule); the start vector
// is the plain extrapolation 2 u^n - u^{n-1} (hf_set_start_vector kind 1).
#pragma once
#include "hf_solver.hpp"

namespace {

// Sum over the threads of a 256-thread block that serve the same column j = threadIdx.x % NV, fixed order.
template <int NV>
__device__ __forceinline__ double block_colsum(double v, double* sw /* [4 * NV] */) {
#pragma unroll
  for (int o = 32; o >= NV; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l < NV) sw[w * NV + l] = v;
  __syncthreads();
  const int j = threadIdx.x % NV;
  double t = sw[j];
#pragma unroll
  for (int k = 1; k < 4; ++k) t += sw[k * NV + j];
  __syncthreads();
  return t;
}

template <int NV>
__device__ __forceinline__ double col_partials(const double* __restrict__ part /* column's MAXP slots */, int P, double* sw) {
  double v = 0.0;
  for (int k = threadIdx.x / NV; k < P; k += TPB / NV) v += part[k];
  return block_colsum<NV>(v, sw);
}

// Fine-pattern SpMV on NV interleaved columns, thread = (row, column).  Modes as k_spmv (0, 2, 3, 4, 5, 8, 9).
template <int MODE, int NV, bool PERCOL>
__global__ __launch_bounds__(TPB) void kb_spmv(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                               const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y,
                                               Scal* __restrict__ scal, double* __restrict__ part0, const double* __restrict__ bvec,
                                               const double* __restrict__ dinv, double* __restrict__ pvec, double* __restrict__ part1,
                                               double* __restrict__ part2, double w, int P, int parity) {
  __shared__ double sw[4 * NV];
  constexpr int RPB = TPB / NV;
  const int j = threadIdx.x % NV, rl = threadIdx.x / NV;
  Scal* sc = scal + j;
  bool active = true;
  if (MODE == 3 || MODE == 4 || MODE == 9) active = sc->done == 0;
  double beta = 0.0;
  bool first9 = false;
  if (MODE == 9) {
    first9 = sc->first != 0;
    const double rz_new = col_partials<NV>(part1 + (parity * NV + j) * MAXP, P, sw);
    const double rz_old = col_partials<NV>(part1 + ((parity ^ 1) * NV + j) * MAXP, P, sw);
    const double zz = col_partials<NV>(part2 + j * MAXP, P, sw);
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
    if (MODE == 2 || MODE == 4 || MODE == 5) e_d = PERCOL ? dinv[o] : dinv[row];
    if (MODE == 9 && !first9) { e_y = y[o]; e_p = pvec[o]; }
    if (MODE == 4 || MODE == 8 || MODE == 9) e_x = x[o];
    double s = 0.0;
    for (int k = k0; k < k1; k += 4) {
      int c[4];
      double v[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = k + u < k1;
        c[u] = in ? colidx[k + u] : 0;
        v[u] = in ? (PERCOL ? vals[static_cast<size_t>(k + u) * NV + j] : vals[k + u]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = (k + u < k1) ? x[static_cast<size_t>(c[u]) * NV + j] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u] * xv[u];
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
    const double t0 = block_colsum<NV>(acc0, sw);
    if (rl == 0) part0[j * MAXP + blockIdx.x] = t0;
  }
  if (MODE == 2 || MODE == 5) {
    const double t1 = block_colsum<NV>(acc1, sw);
    const double t2 = block_colsum<NV>(acc2, sw);
    if (rl == 0) { part1[j * MAXP + blockIdx.x] = t1; part2[j * MAXP + blockIdx.x] = t2; }
  }
}

// Generic CSR operator (shared values) times NV interleaved columns: LANES lanes share a row, each keeps NV
// accumulators, so every index and value is read once.  VMODE 0: y = A x, 1: y += A x.
template <int NV, int LANES, int VMODE>
__global__ __launch_bounds__(TPB) void kb_csr(int nrow, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                              const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
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
      const double v0 = val[k], v1 = in2 ? val[k + LANES] : 0.0;
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
template <int NV>
__global__ __launch_bounds__(TPB) void kb_dense(int n, int ld, const double* __restrict__ Ainv, const double* __restrict__ b,
                                                double* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * TPB) >> 6;
  for (int row = wave; row < n; row += nwaves) {
    const double* arow = Ainv + static_cast<size_t>(row) * ld;
    double acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0;
    for (int c = lane; c < n; c += 64) {
      const double a = arow[c];
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

// PCG start per column: tolerance and convergence of the initial iterate (one workgroup).
template <int NV>
__global__ __launch_bounds__(TPB) void kb_begin(int P, double rtol, double atol, const double* __restrict__ part_zz,
                                                const double* __restrict__ part_bn, Scal* __restrict__ scal) {
  __shared__ double sw[4 * NV];
  const int j = threadIdx.x % NV;
  const double zz = col_partials<NV>(part_zz + j * MAXP, P, sw);
  const double bn2 = col_partials<NV>(part_bn + j * MAXP, P, sw);
  if (threadIdx.x < NV) {
    const double tol = fmax(rtol * sqrt(bn2), atol);
    Scal* sc = scal + j;
    sc->tol2 = tol * tol;
    sc->bn2 = bn2;
    sc->zz = zz;
    sc->iters = 0;
    sc->first = 1;
    sc->done = (zz <= tol * tol) ? 1 : 0;
  }
}

// x += alpha p; r -= alpha Ap; z = D^-1 r (Jacobi: + r.z partials) or z0 = w D^-1 r (multigrid); (D^-1 r)^2 partials
template <int NV, bool AMG, bool PERCOL>
__global__ __launch_bounds__(TPB) void kb_update(int n, int P, int parity, Scal* __restrict__ scal, const double* __restrict__ part_pAp,
                                                 double* __restrict__ part_rz, double* __restrict__ part_zz, double* __restrict__ x,
                                                 double* __restrict__ r, const double* __restrict__ p, const double* __restrict__ Ap,
                                                 const double* __restrict__ dinv, double w, double* __restrict__ z) {
  __shared__ double sw[4 * NV];
  constexpr int RPB = TPB / NV;
  const int j = threadIdx.x % NV, rl = threadIdx.x / NV;
  Scal* sc = scal + j;
  const double pAp = col_partials<NV>(part_pAp + j * MAXP, P, sw);
  const double rz = col_partials<NV>(part_rz + (parity * NV + j) * MAXP, P, sw);
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
    const double zi = (PERCOL ? dinv[o] : dinv[row]) * ri;
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
template <int NV, bool PERCOL>
__global__ void kb_lift(int nrows, const int32_t* __restrict__ rows, const int32_t* __restrict__ ptr, const int32_t* __restrict__ bc,
                        const double* __restrict__ val, const double* __restrict__ g, double* __restrict__ b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / NV, j = t % NV;
  if (q >= nrows) return;
  double s = 0.0;
  for (int k = ptr[q]; k < ptr[q + 1]; ++k) s += (PERCOL ? val[static_cast<size_t>(k) * NV + j] : val[k]) * g[static_cast<size_t>(bc[k]) * NV + j];
  b[static_cast<size_t>(rows[q]) * NV + j] -= s;
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
void free_batch(hf_ctx* ctx) {
  hf_ctx::Batch& B = ctx->batch;
  dev_free(&B.A); dev_free(&B.dinv); dev_free(&B.lift_val); dev_free(&B.g);
  dev_free(&B.u); dev_free(&B.uprev); dev_free(&B.ustart); dev_free(&B.b); dev_free(&B.r); dev_free(&B.p); dev_free(&B.Ap);
  dev_free(&B.z); dev_free(&B.z2); dev_free(&B.tmp);
  dev_free(&B.part_pAp); dev_free(&B.part_rz); dev_free(&B.part_zz); dev_free(&B.part_bn); dev_free(&B.scal);
  for (auto& L : B.lev) { dev_free(&L.x); dev_free(&L.cat); if (L.own_b) dev_free(&L.b); }
  B.lev.clear();
  if (B.h_scal) { (void)hipHostFree(B.h_scal); B.h_scal = nullptr; }
  B.nv = 0;
}

template <int NV, int VMODE>
void blaunch_csr(hf_ctx* c, const DevCsr& m, const double* x, double* y) {
  const double avg = m.nrow ? static_cast<double>(m.nnz) / m.nrow : 1.0;
  const int lanes = avg <= 2.5 ? 2 : avg <= 5.0 ? 4 : avg <= 10.0 ? 8 : avg <= 20.0 ? 16 : avg <= 40.0 ? 32 : 64;
  const long long threads = static_cast<long long>(m.nrow) * lanes;
  const int grid = static_cast<int>(std::max(1LL, std::min<long long>((threads + TPB - 1) / TPB, 4096)));
#define HF_BCSR(L) hipLaunchKernelGGL((kb_csr<NV, L, VMODE>), dim3(grid), dim3(TPB), 0, c->stream, m.nrow, m.ptr, m.idx, m.val, x, y)
  switch (lanes) {
    case 2: HF_BCSR(2); break;
    case 4: HF_BCSR(4); break;
    case 8: HF_BCSR(8); break;
    case 16: HF_BCSR(16); break;
    case 32: HF_BCSR(32); break;
    default: HF_BCSR(64); break;
  }
#undef HF_BCSR
}

template <int NV, bool PERCOL>
struct BatchOps {
  static const double* Avals(hf_ctx* c) { return PERCOL ? c->batch.A : c->d_A; }
  static const double* Dinv(hf_ctx* c) { return PERCOL ? c->batch.dinv : c->d_dinv; }

  template <int MODE>
  static void spmv(hf_ctx* c, const double* vals, const double* x, double* y, double* part0 = nullptr, const double* bvec = nullptr,
                   double* pvec = nullptr, double* part1 = nullptr, double* part2 = nullptr, double w = 0.0, int parity = 0) {
    hf_ctx::Batch& B = c->batch;
    // M is always shared (rho_c does not change inside a batch): MODE 0 / 8 on M use the shared-value kernel
    if (vals == c->d_M)
      hipLaunchKernelGGL((kb_spmv<MODE, NV, false>), dim3(B.Pb), dim3(TPB), 0, c->stream, c->n, c->d_rowptr, c->d_colidx, vals, x, y,
                         B.scal, part0, bvec, c->d_dinv, pvec, part1, part2, w, B.Pb, parity);
    else
      hipLaunchKernelGGL((kb_spmv<MODE, NV, PERCOL>), dim3(B.Pb), dim3(TPB), 0, c->stream, c->n, c->d_rowptr, c->d_colidx, vals, x, y,
                         B.scal, part0, bvec, Dinv(c), pvec, part1, part2, w, B.Pb, parity);
  }

  // z = B r for every column: the V(1,1) cycle of hf_solver.hpp's vcycle() on interleaved vectors
  static void vcycle(hf_ctx* c, int out_slot) {
    hf_ctx::Batch& B = c->batch;
    const int nl = static_cast<int>(c->amg.size());
    const double w0 = c->amg[0].omega;
    double* rz_out = B.part_rz + static_cast<size_t>(out_slot) * NV * MAXP;
    if (nl == 1) {
      spmv<4>(c, Avals(c), B.z, B.z2, rz_out, B.r, nullptr, nullptr, nullptr, w0);
      return;
    }
    spmv<3>(c, Avals(c), B.z, B.tmp, nullptr, B.r);
    blaunch_csr<NV, 0>(c, c->amg[0].R, B.tmp, B.lev[1].b);
    for (int l = 1; l + 1 < nl; ++l) blaunch_csr<NV, 0>(c, c->amg[l].Rt, B.lev[l].b, B.lev[l + 1].b);
    {
      const DevLevel& Lc = c->amg[nl - 1];
      if (c->coarse_n > 0) {
        const int g = std::max(1, std::min((Lc.n + 3) / 4, 1024));
        hipLaunchKernelGGL((kb_dense<NV>), dim3(g), dim3(TPB), 0, c->stream, Lc.n, c->coarse_ld, c->d_coarse_inv, B.lev[nl - 1].b,
                           B.lev[nl - 1].res);
      } else {
        const int g = std::max(1, std::min((Lc.n * NV + TPB - 1) / TPB, 1024));
        hipLaunchKernelGGL((kb_scale<NV>), dim3(g), dim3(TPB), 0, c->stream, Lc.n, Lc.omega, Lc.dinv, B.lev[nl - 1].b, B.lev[nl - 1].res);
      }
    }
    for (int l = nl - 2; l >= 1; --l) blaunch_csr<NV, 0>(c, c->amg[l].GP, B.lev[l].cat, B.lev[l].res);
    blaunch_csr<NV, 1>(c, c->amg[0].P, B.lev[1].res, B.z);
    spmv<4>(c, Avals(c), B.z, B.z2, rz_out, B.r, nullptr, nullptr, nullptr, w0);
  }

  static void iteration(hf_ctx* c, bool use_amg, int parity) {
    hf_ctx::Batch& B = c->batch;
    if (use_amg) {
      spmv<9>(c, Avals(c), B.z2, B.Ap, B.part_pAp, nullptr, B.p, B.part_rz, B.part_zz, 0.0, parity);
      hipLaunchKernelGGL((kb_update<NV, true, PERCOL>), dim3(B.Pb), dim3(TPB), 0, c->stream, c->n, B.Pb, parity, B.scal, B.part_pAp,
                         B.part_rz, B.part_zz, B.u, B.r, B.p, B.Ap, Dinv(c), c->amg[0].omega, B.z);
      vcycle(c, parity ^ 1);
    } else {
      spmv<9>(c, Avals(c), B.z, B.Ap, B.part_pAp, nullptr, B.p, B.part_rz, B.part_zz, 0.0, parity);
      hipLaunchKernelGGL((kb_update<NV, false, PERCOL>), dim3(B.Pb), dim3(TPB), 0, c->stream, c->n, B.Pb, parity, B.scal, B.part_pAp,
                         B.part_rz, B.part_zz, B.u, B.r, B.p, B.Ap, Dinv(c), 0.0, B.z);
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

  // PCG on all columns, started from B.u; iteration counts / residuals are left in B.h_scal
  static int pcg(hf_ctx* ctx, bool use_amg, double rtol, double atol, int max_it) {
    hf_ctx::Batch& B = ctx->batch;
    HF_HIP(hipMemsetAsync(B.scal, 0, sizeof(Scal) * NV, ctx->stream));
    if (!use_amg) {
      spmv<2>(ctx, Avals(ctx), B.u, B.r, B.part_rz, B.b, B.z, B.part_zz, B.part_bn, 0.0);
      hipLaunchKernelGGL((kb_begin<NV>), dim3(1), dim3(TPB), 0, ctx->stream, B.Pb, rtol, atol, B.part_zz, B.part_bn, B.scal);
    } else {
      spmv<5>(ctx, Avals(ctx), B.u, B.r, nullptr, B.b, B.z, B.part_zz, B.part_bn, ctx->amg[0].omega);
      hipLaunchKernelGGL((kb_begin<NV>), dim3(1), dim3(TPB), 0, ctx->stream, B.Pb, rtol, atol, B.part_zz, B.part_bn, B.scal);
      vcycle(ctx, 0);
    }
    HF_HIP(hipGetLastError());
    bool all_done = false, breakdown = false;
    int launched = 0, iters = 0;
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

  // one time step of all columns to the boundary values g_dev (n_bc x NV, interleaved, on the device)
  static int step(hf_ctx* ctx, const double* g_dev, double rtol, double atol, int max_it) {
    hf_ctx::Batch& B = ctx->batch;
    const int nb = ctx->nbc;
    const size_t vec = sizeof(double) * static_cast<size_t>(ctx->n) * NV;
    if (B.have_prev) {
      spmv<8>(ctx, ctx->d_M, B.u, B.b, nullptr, B.uprev, B.ustart);
      HF_HIP(hipMemcpyAsync(B.uprev, B.u, vec, hipMemcpyDeviceToDevice, ctx->stream));
      HF_HIP(hipMemcpyAsync(B.u, B.ustart, vec, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      spmv<0>(ctx, ctx->d_M, B.u, B.b);
      HF_HIP(hipMemcpyAsync(B.uprev, B.u, vec, hipMemcpyDeviceToDevice, ctx->stream));
      B.have_prev = true;
    }
    if (nb > 0) {
      if (ctx->nlift_rows > 0) {
        const int thr = ctx->nlift_rows * NV;
        hipLaunchKernelGGL((kb_lift<NV, PERCOL>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ctx->nlift_rows, ctx->d_lift_rows,
                           ctx->d_lift_ptr, ctx->d_lift_bc, PERCOL ? B.lift_val : ctx->d_lift_val, g_dev, B.b);
      }
      hipLaunchKernelGGL((kb_set_bc<NV>), dim3((nb * NV + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, g_dev, B.b, B.u);
    }
    const bool use_amg = ctx->precond == 1 && ctx->amg_ready;
    return pcg(ctx, use_amg, rtol, atol, max_it);
  }
};

// dispatch over (nv, per-column operator)
template <typename F>
int batch_dispatch(hf_ctx* ctx, F&& f) {
  const hf_ctx::Batch& B = ctx->batch;
  switch (B.nv * 2 + (B.percol ? 1 : 0)) {
    case 4: return f(BatchOps<2, false>());
    case 5: return f(BatchOps<2, true>());
    case 8: return f(BatchOps<4, false>());
    case 9: return f(BatchOps<4, true>());
    case 16: return f(BatchOps<8, false>());
    case 17: return f(BatchOps<8, true>());
    default: return fail(ctx, HF_ERR_STATE, "no batch is open (hf_batch_begin)");
  }
}

}  // namespace
