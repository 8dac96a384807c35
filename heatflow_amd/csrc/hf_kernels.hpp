// Part of libheatflow_hip.so (see heatflow_hip.hip): device code - element assembly, CSR SpMV (LDS-staged and sub-wave), PCG and multigrid kernels
#pragma once
#include "hf_context.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------

// Sum over the 256 threads of a workgroup, identical order every run: 64-lane shuffle tree
// per wavefront, then the four wave sums added in wave order.  Every thread gets the sum.
template <int NW = 4>
__device__ __forceinline__ double block_sum(double v, double* sw) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sw[w] = v;
  __syncthreads();
  double t = sw[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) t += sw[k];
  __syncthreads();
  return t;
}

// Fixed-order sum of the P per-workgroup partials written by the previous kernel.
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int P, double* s4) {
  double v = 0.0;
  for (int k = threadIdx.x; k < P; k += TPB) v += part[k];
  return block_sum(v, s4);
}

// Chunk schedule of the row-chunked kernels.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 says which blocks share an XCD and its L2; speed only, never correctness), so with
// HF_XCD_MAP each XCD group walks one contiguous eighth of the chunk range: spatially adjacent chunks
// (Morton order) then share an L2, which keeps the SpMV's neighbour gathers and a chunk's vector
// slices from kernel to kernel on the same XCD.  Every kernel uses the same schedule.
#ifndef HF_XCD_MAP
#define HF_XCD_MAP 1
#endif
struct ChunkIter {
  int chunk, step, end;
  __device__ __forceinline__ ChunkIter(int nchunks) {
    if (HF_XCD_MAP && (gridDim.x & 7) == 0) {
      const int per = (nchunks + 7) >> 3;
      const int xcd = blockIdx.x & 7;
      chunk = xcd * per + (blockIdx.x >> 3);
      step = gridDim.x >> 3;
      end = min(nchunks, (xcd + 1) * per);
    } else {
      chunk = blockIdx.x;
      step = gridDim.x;
      end = nchunks;
    }
  }
};

// r-weighted P1 element matrices (reference forms run_with_diamond.py:328-331).
//   M_ii = rho_c |K| (3 r_i + r_j + r_k)/30,  M_ij = rho_c |K| (2 r_i + 2 r_j + r_k)/60
//   K_ij = kappa |K| rbar (b_i b_j + c_i c_j)/d^2, d = 2*signed area, rbar = mean r
// m[] / k[] hold the symmetric 3x3 as {00, 11, 22, 01, 02, 12}.
__device__ __forceinline__ void element_local(const double2 p0, const double2 p1, const double2 p2, double rho_c,
                                              double kappa, double m[6], double k[6]) {
  // No FMA contraction here: the same element is evaluated by different workgroups (and by
  // different unrolled copies of the caller); every evaluation must give the same bits so that the
  // assembled matrices stay exactly symmetric.
#pragma clang fp contract(off)
  const double d = (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
  const double area = 0.5 * fabs(d);
  const double b0 = p1.y - p2.y, b1 = p2.y - p0.y, b2 = p0.y - p1.y;
  const double c0 = p2.x - p1.x, c1 = p0.x - p2.x, c2 = p1.x - p0.x;
  const double rsum = (p0.y + p1.y) + p2.y;
  const double ks = kappa * area * (rsum / 3.0) / (d * d);
  k[0] = ks * (b0 * b0 + c0 * c0);
  k[1] = ks * (b1 * b1 + c1 * c1);
  k[2] = ks * (b2 * b2 + c2 * c2);
  k[3] = ks * (b0 * b1 + c0 * c1);
  k[4] = ks * (b0 * b2 + c0 * c2);
  k[5] = ks * (b1 * b2 + c1 * c2);
  const double ms = rho_c * area;
  m[0] = ms * ((2.0 * p0.y + rsum) / 30.0);
  m[1] = ms * ((2.0 * p1.y + rsum) / 30.0);
  m[2] = ms * ((2.0 * p2.y + rsum) / 30.0);
  m[3] = ms * ((rsum + p0.y + p1.y) / 60.0);
  m[4] = ms * ((rsum + p0.y + p2.y) / 60.0);
  m[5] = ms * ((rsum + p1.y + p2.y) / 60.0);
}

__device__ __forceinline__ int sym_index(int a, int b) {  // (a,b) -> slot in {00,11,22,01,02,12}
  return a == b ? a : (a + b + 2);                        // 01->3, 02->4, 12->5
}

// ------------------------------------------------------------------------------------------
// Assembly, LDS-staged owner-computes.  Workgroup `blk` owns rows [blk*RBA, blk*RBA+RBA): it
// stages that slab of M and A (values) plus its column indices in LDS, walks the elements
// incident to its rows (precomputed list; an element on a block boundary is visited by each
// owning block, which adds only the rows it owns), and writes the slab out coalesced.
//   COLORED = false: LDS f64 atomics (ds_add_f64), any order
//   COLORED = true : elements grouped by colour (no two share an owned row), plain RMW,
//                    barrier between colours -> bitwise reproducible
// ------------------------------------------------------------------------------------------
// One list entry = 24 bytes = three int2: (n0, n1) (n2, tag<<11 | owned<<8 | off8) (off0..3, off4..7):
// the element record and the offsets of its nine contributions (a,b) = (0,0) (0,1) ... (2,2) inside
// the CSR rows of its nodes; `owned` marks the nodes whose rows this workgroup owns.
struct AsmEntry { int n0, n1, n2; unsigned w3, off03, off47; };

__device__ __forceinline__ AsmEntry load_entry(const int2* __restrict__ ent, int q) {
  const int2 a = ent[3 * q], b = ent[3 * q + 1], c = ent[3 * q + 2];
  return AsmEntry{a.x, a.y, b.x, static_cast<unsigned>(b.y), static_cast<unsigned>(c.x), static_cast<unsigned>(c.y)};
}

template <bool COLORED>
__global__ __launch_bounds__(RBA) void k_assemble_lds(int n, int cap, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ blk_eptr,
                                                      const int32_t* __restrict__ blk_cptr,
                                                      const int2* __restrict__ blk_ent,
                                                      const double2* __restrict__ zr,
                                                      const double* __restrict__ kappa_tab,
                                                      const double* __restrict__ rhoc_tab, double dt,
                                                      double* __restrict__ Mv, double* __restrict__ Av) {
  extern __shared__ double smem[];
  // slab layout: the coloured variant keeps (M, A) of a slot side by side (one 16-byte read-modify-write per entry:
  // 80 -> 74 us at C3); the atomic variant keeps two arrays (atomics on neighbouring words collide: 50 vs 56 us)
  double2* sMA = reinterpret_cast<double2*>(smem);
  double* sM = smem;
  double* sA = smem + cap;
  int* sR = reinterpret_cast<int*>(smem + 2 * cap);

  const int blk = blockIdx.x;
  const int r0 = blk * RBA;
  const int r1 = min(n, r0 + RBA);
  const int k0 = rowptr[r0];
  const int nk = rowptr[r1] - k0;
  for (int k = threadIdx.x; k < nk; k += RBA) {
    if (COLORED) {
      sMA[k] = make_double2(0.0, 0.0);
    } else {
      sM[k] = 0.0;
      sA[k] = 0.0;
    }
  }
  for (int k = threadIdx.x; k <= r1 - r0; k += RBA) sR[k] = rowptr[r0 + k] - k0;
  __syncthreads();

  auto scatter = [&](const AsmEntry e, const double2 p0, const double2 p1, const double2 p2) {
    double m[6], kk[6], av6[6];
    const int tag = static_cast<int>(e.w3 >> 11);
    element_local(p0, p1, p2, rhoc_tab[tag], kappa_tab[tag], m, kk);
#pragma unroll
    for (int q = 0; q < 6; ++q) av6[q] = fma(dt, kk[q], m[q]);  // once per unique entry, explicit FMA: same bits everywhere
    const int nd[3] = {e.n0, e.n1, e.n2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!((e.w3 >> (8 + a)) & 1u)) continue;
      const int base = sR[nd[a] - r0];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int q9 = a * 3 + b;
        const unsigned off = q9 < 4 ? (e.off03 >> (8 * q9)) & 0xFFu : q9 < 8 ? (e.off47 >> (8 * (q9 - 4))) & 0xFFu : e.w3 & 0xFFu;
        const int slot = base + static_cast<int>(off);
        const int q = sym_index(a, b);
        if (COLORED) {
          double2 t = sMA[slot];
          t.x += m[q];
          t.y += av6[q];
          sMA[slot] = t;
        } else {
          atomicAdd(&sM[slot], m[q]);
          atomicAdd(&sA[slot], av6[q]);
        }
      }
    }
  };
  // The list is laid out per workgroup and streamed coalesced; only the coordinates are gathered.
  // Three elements in flight per lane: all loads are issued before the first scatter.
  constexpr int NF = 3;
  auto run_range = [&](int e0, int e1) {
    for (int k = e0 + threadIdx.x; k < e1; k += NF * RBA) {
      AsmEntry e[NF];
      double2 pa[NF], pb[NF], pc[NF];
#pragma unroll
      for (int u = 0; u < NF; ++u) e[u] = load_entry(blk_ent, min(k + u * RBA, e1 - 1));
#pragma unroll
      for (int u = 0; u < NF; ++u) { pa[u] = zr[e[u].n0]; pb[u] = zr[e[u].n1]; pc[u] = zr[e[u].n2]; }
#pragma unroll
      for (int u = 0; u < NF; ++u)
        if (k + u * RBA < e1) scatter(e[u], pa[u], pb[u], pc[u]);
    }
  };

  if (COLORED) {
    const int32_t* cp = blk_cptr + static_cast<size_t>(blk) * (NCOL + 1);
    for (int c = 0; c < NCOL; ++c) {
      const int e0 = cp[c], e1 = cp[c + 1];
      if (e0 == e1) { if (e1 == cp[NCOL]) break; else continue; }
      run_range(e0, e1);
      __syncthreads();
    }
  } else {
    run_range(blk_eptr[blk], blk_eptr[blk + 1]);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nk; k += RBA) {
    if (COLORED) {
      const double2 t = sMA[k];
      Mv[k0 + k] = t.x;
      Av[k0 + k] = t.y;
    } else {
      Mv[k0 + k] = sM[k];
      Av[k0 + k] = sA[k];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Assembly, row gather (HF_ASM_ROW_GATHER): lane t of a workgroup owns CSR row r0 + t and visits the
// triangles at its node one after another (ELL slab of 16-bit entries, hf_pattern.hpp RowGather).  A
// triangle is described by the positions of its two other vertices inside the row's column list, so a
// contribution's slot is known without any search or offset table, and the element array is never read.
// Per block of RBA rows the kernel stages, all as coalesced 16-byte streams: the rows' 16-bit column
// positions, the coordinates of the block's column list (own rows + halo, a per-block copy made at
// hf_set_mesh) and its ELL entries; the (M, A) value slab lives in LDS and is streamed out once.  The
// visit loop touches LDS only.  Rows are private to their lane: no atomics, no colours, no barriers
// inside the loop, and the order of additions is the list order - bitwise reproducible.
// Workgroups are persistent: while block b is computed the loads of the workgroup's next block are in
// flight into registers and the stores of its previous block drain, so the three phases overlap.
// Every triangle is evaluated once per vertex ("me", next, previous in the triangle's own cyclic
// order).  The formulas are symmetric by construction, so the three evaluations agree bit for bit
// where they must (A_ij == A_ji) without agreeing on a vertex order:
//   edges   e_v = P_next(v) - P_prev(v): the same subtraction from every vertex's point of view
//   K_ab  = ks * (e_a . e_b)             (products commute)
//   d     = max |e_a x e_b| over the three edge pairs (three roundings of twice the area)
//   rsum  = (lo + mid) + hi of the three radii sorted by value
//   M_aa  = ms/30 * (2 r_a + rsum),  M_ab = ms/60 * (rsum + (r_a + r_b))
// One IEEE division per evaluation (1/d^2); constant divisors are multiplications by rounded reciprocals.
// ------------------------------------------------------------------------------------------
struct ElemRow { double m0, m1, m2, k0, k1, k2; };   // row "me" of the element matrices: (me,me) (me,next) (me,prev)

__device__ __forceinline__ ElemRow element_row(const double2 Pi, const double2 Pj, const double2 Pk, double rho_c, double kappa) {
#pragma clang fp contract(off)
  const double eix = Pk.x - Pj.x, eiy = Pk.y - Pj.y;     // edge opposite "me"
  const double ejx = Pi.x - Pk.x, ejy = Pi.y - Pk.y;     // opposite next
  const double ekx = Pj.x - Pi.x, eky = Pj.y - Pi.y;     // opposite previous
  const double c1 = fabs(ejx * eky - ejy * ekx), c2 = fabs(ekx * eiy - eky * eix), c3 = fabs(eix * ejy - eiy * ejx);
  const double d = fmax(fmax(c1, c2), c3);
  const double ab_lo = fmin(Pi.y, Pj.y), ab_hi = fmax(Pi.y, Pj.y);
  const double lo = fmin(ab_lo, Pk.y), hi = fmax(ab_hi, Pk.y), mid = fmax(ab_lo, fmin(ab_hi, Pk.y));
  const double rsum = (lo + mid) + hi;
  const double area = 0.5 * d;
  const double ks = kappa * area * (rsum * (1.0 / 3.0)) * (1.0 / (d * d));
  const double ms30 = rho_c * area * (1.0 / 30.0), ms60 = rho_c * area * (1.0 / 60.0);
  ElemRow o;
  o.k0 = ks * (eix * eix + eiy * eiy);
  o.k1 = ks * (eix * ejx + eiy * ejy);
  o.k2 = ks * (eix * ekx + eiy * eky);
  o.m0 = ms30 * (2.0 * Pi.y + rsum);
  o.m1 = ms60 * (rsum + (Pi.y + Pj.y));
  o.m2 = ms60 * (rsum + (Pi.y + Pk.y));
  return o;
}

constexpr int RG_NC = 4;   // 16-byte vectors of column positions a lane can prefetch (RBA * 8 * RG_NC >= slab slots: rows hold <= 32 entries)
constexpr int RG_NX = 5;   // coordinate pairs a lane can prefetch (RBA * RG_NX >= column-list length, checked on the host)

__global__ __launch_bounds__(RBA) void k_assemble_rows(int nblk, int cap /* slab slots, even */, int capd /* column-list slots */,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int4* __restrict__ hdr /* 2 per block */,
                                                       const uint4* __restrict__ ell, const uint4* __restrict__ cid16,
                                                       const double2* __restrict__ zrb,
                                                       const double* __restrict__ kappa_idx, const double* __restrict__ rhoc_idx,
                                                       double dt, double* __restrict__ Mv, double* __restrict__ Av) {
  extern __shared__ double smem[];
  double2* sMA = reinterpret_cast<double2*>(smem);                   // (M, A) per slot
  double2* sXd = sMA + cap;                                          // coordinates of the block's column list
  int* sR = reinterpret_cast<int*>(sXd + capd);                      // row starts inside the slab
  uint4* sC4 = reinterpret_cast<uint4*>(sR + RBA + 4);               // column-list position per slot, from the 8-aligned start
  const uint16_t* sC = reinterpret_cast<const uint16_t*>(sC4);

  const int t = threadIdx.x;
  for (int k = t; k < cap; k += RBA) sMA[k] = make_double2(0.0, 0.0);

  // prefetch registers of the next block
  int4 hA, hB;
  uint4 pe, pc[RG_NC];
  double2 px[RG_NX];
  int pr = 0;
  auto prefetch = [&](int blk) {
    hA = hdr[2 * blk];                                               // (k0, nk, d0, nd)
    hB = hdr[2 * blk + 1];                                           // (ell offset in 16-byte units, groups of 8 visits, local id of row r0, rows)
    pe = ell[hB.x + t];
    const int c0 = hA.x >> 3, nc = ((hA.x + hA.y + 7) >> 3) - c0;
#pragma unroll
    for (int u = 0; u < RG_NC; ++u) pc[u] = (t + u * RBA < nc) ? cid16[c0 + t + u * RBA] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < RG_NX; ++u) px[u] = (t + u * RBA < hA.w) ? zrb[hA.z + t + u * RBA] : make_double2(0.0, 0.0);
    pr = (t < hB.w) ? rowptr[blk * RBA + t] - hA.x : 0;
  };
  int blk = blockIdx.x;
  if (blk < nblk) prefetch(blk);
  while (blk < nblk) {
    // stage the prefetched block
    const int4 cA = hA, cB = hB;
    const uint4 ce = pe;
    const int nc = ((cA.x + cA.y + 7) >> 3) - (cA.x >> 3);
#pragma unroll
    for (int u = 0; u < RG_NC; ++u) if (t + u * RBA < nc) sC4[t + u * RBA] = pc[u];
#pragma unroll
    for (int u = 0; u < RG_NX; ++u) if (t + u * RBA < cA.w) sXd[t + u * RBA] = px[u];
    sR[t] = pr;
    __syncthreads();
    const int nxt = blk + gridDim.x;
    if (nxt < nblk) prefetch(nxt);                                   // in flight during the visit loop

    if (t < cB.w) {
      const int base = sR[t];
      const int sbase = base + (cA.x & 7);                           // the position array starts at the 8-aligned slot
      const int ci = cB.z + t;                                       // my own position in the column list
      const double2 Pi = sXd[ci];
      double dM = 0.0, dA = 0.0;
      int pd = 0;                                                    // diagonal: the slot whose column is my own row
      while (pd < 31 && sC[sbase + pd] < ci) ++pd;
      for (int g = 0; g < cB.y; ++g) {
        const uint4 ev = g == 0 ? ce : ell[cB.x + g * RBA + t];
        const unsigned w[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned e = (w[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;
          if (e == 0xFFFFu) continue;
          const int pj = e & 31u, pk = (e >> 5) & 31u, tg = e >> 10;
          const double2 Pj = sXd[sC[sbase + pj]], Pk = sXd[sC[sbase + pk]];
          const ElemRow r = element_row(Pi, Pj, Pk, rhoc_idx[tg], kappa_idx[tg]);
          dM += r.m0;
          dA += fma(dt, r.k0, r.m0);
          double2 v = sMA[base + pj];
          v.x += r.m1;
          v.y += fma(dt, r.k1, r.m1);
          sMA[base + pj] = v;
          v = sMA[base + pk];
          v.x += r.m2;
          v.y += fma(dt, r.k2, r.m2);
          sMA[base + pk] = v;
        }
      }
      sMA[base + pd] = make_double2(dM, dA);
    }
    __syncthreads();
    // stream the slab out and leave it zeroed for the next block (slot k stays with the lane that reads it here)
    for (int k = t; k < cA.y; k += RBA) {
      const double2 v = sMA[k];
      Mv[cA.x + k] = v.x;
      Av[cA.x + k] = v.y;
      sMA[k] = make_double2(0.0, 0.0);
    }
    blk = nxt;
    // no barrier needed here: the staging stores above touch sC / sXd / sR only after every lane has passed the
    // barrier that ended the visit loop, and the slab is re-read only behind the next staging barrier
  }
}

// Read-flux projection right-hand sides by row gather (same lists and staging as k_assemble_rows, plus the block's
// slice of the state u): b_c[i] = sum over the triangles at node i of (d_c T)_e * |K| (2 r_i + r_j + r_k)/12, summed
// in list order in a register - no LDS accumulation, no atomics, bitwise reproducible.
__global__ __launch_bounds__(RBA) void k_grad_rows(int nblk, int capd, const int4* __restrict__ hdr, const uint4* __restrict__ ell,
                                                   const uint4* __restrict__ cid16, const double2* __restrict__ zrb,
                                                   const int32_t* __restrict__ dict, const int32_t* __restrict__ rowptr,
                                                   const double* __restrict__ u, double* __restrict__ bz, double* __restrict__ br,
                                                   int stride /* of the outputs - 1: plain arrays; 2: bz = out, br = out + 1 interleaved; nv: column j of an interleaved batch vector */,
                                                   int ustride, int uoff /* the state is u[node * ustride + uoff] (batched loop: column uoff of nv) */) {
  extern __shared__ double smem[];
  double2* sXd = reinterpret_cast<double2*>(smem);
  double* sU = smem + 2 * capd;
  int* sR = reinterpret_cast<int*>(sU + capd + (capd & 1));
  uint4* sC4 = reinterpret_cast<uint4*>(sR + RBA + 4);
  const uint16_t* sC = reinterpret_cast<const uint16_t*>(sC4);
  const int t = threadIdx.x;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int4 cA = hdr[2 * blk], cB = hdr[2 * blk + 1];
    const int c0 = cA.x >> 3, nc = ((cA.x + cA.y + 7) >> 3) - c0;
    for (int i = t; i < nc; i += RBA) sC4[i] = cid16[c0 + i];
    for (int i = t; i < cA.w; i += RBA) { sXd[i] = zrb[cA.z + i]; sU[i] = u[static_cast<size_t>(dict[cA.z + i]) * ustride + uoff]; }
    if (t < cB.w) sR[t] = rowptr[blk * RBA + t] - cA.x;
    __syncthreads();
    if (t < cB.w) {
#pragma clang fp contract(off)
      const int sbase = sR[t] + (cA.x & 7);
      const int ci = cB.z + t;
      const double2 Pi = sXd[ci];
      const double ui = sU[ci];
      double az = 0.0, ar = 0.0;
      for (int g = 0; g < cB.y; ++g) {
        const uint4 ev = ell[cB.x + g * RBA + t];
        const unsigned w[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned e = (w[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
          if (e == 0xFFFFu) continue;
          const int cj = sC[sbase + (e & 31u)], ck = sC[sbase + ((e >> 5) & 31u)];
          const double2 Pj = sXd[cj], Pk = sXd[ck];
          const double uj = sU[cj], uk = sU[ck];
          const double d = (Pj.x - Pi.x) * (Pk.y - Pi.y) - (Pk.x - Pi.x) * (Pj.y - Pi.y);
          const double gz = (ui * (Pj.y - Pk.y) + uj * (Pk.y - Pi.y) + uk * (Pi.y - Pj.y)) / d;
          const double gr = (ui * (Pk.x - Pj.x) + uj * (Pi.x - Pk.x) + uk * (Pj.x - Pi.x)) / d;
          const double wgt = 0.5 * fabs(d) * (Pi.y + ((Pi.y + Pj.y) + Pk.y)) / 12.0;
          az += gz * wgt;
          ar += gr * wgt;
        }
      }
      if (bz != nullptr) bz[static_cast<size_t>(blk * RBA + t) * stride] = az;
      if (br != nullptr) br[static_cast<size_t>(blk * RBA + t) * stride] = ar;
    }
    __syncthreads();
  }
}

// per-block copy of the coordinates of each block's column list (hf_set_mesh, once)
__global__ void k_gather_coords(int64_t total, const int32_t* __restrict__ dict, const double2* __restrict__ zr,
                                double2* __restrict__ zrb) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < total) zrb[i] = zr[dict[i]];
}

// Baseline: one thread per element, f64 atomics into global CSR (values must be zeroed).
__global__ __launch_bounds__(TPB) void k_assemble_global(int ne, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colidx,
                                                         const int4* __restrict__ elem,
                                                         const double2* __restrict__ zr,
                                                         const double* __restrict__ kappa_tab,
                                                         const double* __restrict__ rhoc_tab, double dt,
                                                         double* __restrict__ Mv, double* __restrict__ Av) {
  const int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= ne) return;
  const int4 el = elem[e];
  const int nd[3] = {el.x, el.y, el.z};
  double m[6], kk[6], av6[6];
  element_local(zr[el.x], zr[el.y], zr[el.z], rhoc_tab[el.w], kappa_tab[el.w], m, kk);
#pragma unroll
  for (int q = 0; q < 6; ++q) av6[q] = fma(dt, kk[q], m[q]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int s0 = rowptr[nd[a]], s1 = rowptr[nd[a] + 1];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      int s = s0;
      while (s < s1 && colidx[s] != nd[b]) ++s;
      const int q = sym_index(a, b);
      atomicAdd(&Mv[s], m[q]);
      atomicAdd(&Av[s], av6[q]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Read-flux projection (reference run_no_diamond.py:479-489, 544-550): L2 projection of grad T
// onto vector P1 with the r-weighted mass matrix.  The reference solves one 2n x 2n system; the
// components decouple into two scalar solves with M_r(1).  This kernel forms both right-hand
// sides  b_c[i] = sum_e (d_c T)_e * int_e phi_i r dx,  int_e phi_i r = |K| (2 r_i + r_j + r_k)/12,
// owner-computes like the assembly: a workgroup owns RBA rows and adds the incident elements.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RBA) void k_grad_rhs(int n, const int32_t* __restrict__ blk_eptr,
                                                  const int2* __restrict__ blk_ent, const double2* __restrict__ zr,
                                                  const double* __restrict__ u, double* __restrict__ bz,
                                                  double* __restrict__ br) {
  __shared__ double sB[2 * RBA];
  const int blk = blockIdx.x;
  const int r0 = blk * RBA;
  const int r1 = min(n, r0 + RBA);
  for (int k = threadIdx.x; k < 2 * RBA; k += RBA) sB[k] = 0.0;
  __syncthreads();
  for (int q = blk_eptr[blk] + threadIdx.x; q < blk_eptr[blk + 1]; q += RBA) {
#pragma clang fp contract(off)
    const AsmEntry e = load_entry(blk_ent, q);
    const double2 p0 = zr[e.n0], p1 = zr[e.n1], p2 = zr[e.n2];
    const double u0 = u[e.n0], u1 = u[e.n1], u2 = u[e.n2];
    const double d = (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
    const double area = 0.5 * fabs(d);
    // grad phi_i = (b_i, c_i)/d
    const double gz = (u0 * (p1.y - p2.y) + u1 * (p2.y - p0.y) + u2 * (p0.y - p1.y)) / d;
    const double gr = (u0 * (p2.x - p1.x) + u1 * (p0.x - p2.x) + u2 * (p1.x - p0.x)) / d;
    const double rsum = (p0.y + p1.y) + p2.y;
    const double wgt[3] = {area * (p0.y + rsum) / 12.0, area * (p1.y + rsum) / 12.0, area * (p2.y + rsum) / 12.0};
    const int nd[3] = {e.n0, e.n1, e.n2};
    const unsigned owned = (e.w3 >> 8) & 7u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!((owned >> a) & 1u)) continue;
      atomicAdd(&sB[2 * (nd[a] - r0)], gz * wgt[a]);
      atomicAdd(&sB[2 * (nd[a] - r0) + 1], gr * wgt[a]);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < r1 - r0; k += RBA) {
    bz[r0 + k] = sB[2 * k];
    br[r0 + k] = sB[2 * k + 1];
  }
}

// ------------------------------------------------------------------------------------------
// Dirichlet elimination (what dolfinx assemble_matrix(form, bcs) leaves): BC rows and
// columns zeroed, unit diagonal.  The column entries A[i, j in B] of free rows i are saved
// first - they are the lifting operator of apply_lifting (run_with_diamond.py:477).
// ------------------------------------------------------------------------------------------
__global__ void k_take_lift(int nlift, const int32_t* __restrict__ slot, double* __restrict__ A,
                            double* __restrict__ val) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nlift) return;
  val[q] = A[slot[q]];
  A[slot[q]] = 0.0;
}

__global__ void k_bc_rows(int nbc, const int32_t* __restrict__ dofs, const int32_t* __restrict__ rowptr,
                          const int32_t* __restrict__ colidx, double* __restrict__ A) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nbc) return;
  const int row = dofs[q];
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) A[k] = (colidx[k] == row) ? 1.0 : 0.0;
}

__global__ void k_dinv(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                       const double* __restrict__ A, double* __restrict__ dinv) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double d = 0.0;
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
    if (colidx[k] == row) d = A[k];
  dinv[row] = 1.0 / d;
}

// b[row] -= sum_q lift_val[q] * g[lift_bc[q]]   (fixed order -> reproducible)
__global__ void k_lift(int nrows, const int32_t* __restrict__ rows, const int32_t* __restrict__ ptr,
                       const int32_t* __restrict__ bc, const double* __restrict__ val,
                       const double* __restrict__ g, double* __restrict__ b) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nrows) return;
  double s = 0.0;
  for (int k = ptr[q]; k < ptr[q + 1]; ++k) s += val[k] * g[bc[k]];
  b[rows[q]] -= s;
}

// set_bc on the right-hand side and on the PCG start vector (u_B = g)
__global__ void k_set_bc(int nbc, const int32_t* __restrict__ dofs, const double* __restrict__ g,
                         double* __restrict__ b, double* __restrict__ u) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nbc) return;
  b[dofs[q]] = g[q];
  u[dofs[q]] = g[q];
}

__global__ void k_gather(int ns, const int32_t* __restrict__ idx, const double* __restrict__ u,
                         double* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < ns) out[q] = u[idx[q]];
}

// ------------------------------------------------------------------------------------------
// CSR SpMV, LDS-staged ("CSR-stream"): a workgroup takes chunks of RB consecutive rows; all
// 256 lanes stream the chunk's values and column indices in nnz order (fully coalesced) and
// park val*x[col] in LDS; then lane t sums the products of row t in column order.  The
// summation order per row is the CSR order -> bitwise reproducible, no atomics.
//   MODE 0: y = A x
//   MODE 2: r = b - A x; p = D^-1 r; partials r.p, p.p, (D^-1 b)^2   (PCG start)
//   MODE 3: y = b - A x                                              (multigrid residual)
//   MODE 4: y = x + w D^-1 (b - A x), partials b.y                   (damped-Jacobi sweep, fused r.z)
//   MODE 5: y = b - A x; p = w D^-1 y; partials (D^-1 y)^2, (D^-1 b)^2   (AMG-PCG start)
//   MODE 6: y += A x                                                 (multigrid prolongation)
//   MODE 7: y = A x, partials b.y                                    (fused up leg of the finest level: z = GP [r; e], r.z)
//   MODE 8: y = A x; p = 2 x - b        (RHS b = M u^n fused with the extrapolated start
//           2 u^n - u^{n-1} of the next solve; `b` carries u^{n-1})
//   MODE 9: PCG iteration head (x = z): convergence test, beta, Ap <- A z + beta Ap, p <- z + beta p,
//           p.Ap partials - SpMV and direction update in one pass
// The chunk is `rpc` rows (512 for the fine operator; fewer for long-row transfer operators so
// that a chunk's products fit the 64-KB LDS window).
// ------------------------------------------------------------------------------------------
// Compressed column indices of the fine operator (C16): per 512-row chunk the sorted list of the columns it
// touches (`dict`, ~1.3 entries per row: the chunk's own rows plus a halo) and a 16-bit position in that list
// per nonzero instead of the 32-bit column.  The chunk's slice of x is staged in LDS once (an almost contiguous
// gather) and the products look it up there: 10 instead of 12 bytes per nonzero, 5x fewer global gathers.
// progress of the solve for the host (ScalMirror): payload first, then the counter the host polls
__device__ __forceinline__ unsigned long long mirror_stamp(unsigned epoch, int v) {
  return (static_cast<unsigned long long>(epoch) << 32) | static_cast<unsigned>(v);
}
__device__ __forceinline__ void mirror_publish(ScalMirror* m, double zz, int iters, int done, unsigned epoch) {
  if (m == nullptr) return;
  m->zz = zz;
  m->iters = iters;
  __hip_atomic_store(&m->done_st, mirror_stamp(epoch, done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(&m->tested_st, mirror_stamp(epoch, iters + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void mirror_breakdown(ScalMirror* m, unsigned epoch) {
  if (m != nullptr) __hip_atomic_store(&m->done_st, mirror_stamp(epoch, 2), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// own != 0 (square operators whose rows all store their diagonal): the chunk's own rows are a contiguous run of its sorted
// column list, so the row-wise operand x[row] of MODE 4 / 8 / 9 is taken from the staged slice instead of read again.
struct ColComp { const int32_t* dptr; const int32_t* dict; const uint16_t* id; int xd_off; int own; };

// The matrix stream (values, 16-bit positions) is read exactly once per launch: HF_NT_STREAM=1 marks those loads non-temporal
// (A/B builds; 0 = plain loads)
#ifndef HF_NT_STREAM
#define HF_NT_STREAM 0
#endif
#ifndef HF_PIPE
#define HF_PIPE 1      // chunk pipeline: 1 = single-precision operators (the transfer operators), 2 = all, 0 = none (A/B builds)
#endif
// HF_PHASE_CLOCK=<mode> (measurement builds only, scripts/phase_clock.py): lane 0 of every workgroup of k_spmv<mode, C16>
// stamps the 100 MHz clock at its phase boundaries (entry, then per chunk: operand slice staged, products parked, rows
// summed) into g_phase[workgroup][16]; hf_debug_phases copies the table out
#ifndef HF_PHASE_CLOCK
#define HF_PHASE_CLOCK -1
#endif
#if HF_PHASE_CLOCK >= 0
#ifndef HF_PHASE_CONV
#define HF_PHASE_CONV 0   // MODE 0 only: 1 = the launch that carries the convergence test (fused restriction of the finest level)
#endif
__device__ unsigned long long g_phase[MAXP * 16];
#define HF_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (MODE == HF_PHASE_CLOCK && C16 && (MODE != 0 || (part2 != nullptr) == (HF_PHASE_CONV != 0)) && threadIdx.x == 0 && (slot) < 16) \
      g_phase[blockIdx.x * 16 + (slot)] = wall_clock64();                                                \
  } while (0)
#else
#define HF_STAMP(slot) do { } while (0)
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if HF_NT_STREAM
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

template <int MODE, bool C16 = false, typename VT = double, int UN = 8>
__global__ __launch_bounds__(TS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_spmv(int n, int nchunks, int rpc /* rows per chunk, <= TS */,
                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                              const VT* __restrict__ vals, const double* __restrict__ x,
                                              double* __restrict__ y, Scal* __restrict__ scal,
                                              double* __restrict__ part0, const double* __restrict__ bvec,
                                              const double* __restrict__ dinv, double* __restrict__ pvec,
                                              double* __restrict__ part1, double* __restrict__ part2, double w,
                                              int npart /* partial slots the consumers sum (>= gridDim.x) */,
                                              int parity, ColComp comp) {
  extern __shared__ double sprod[];
  __shared__ double s4[TS / 64];
  __shared__ int s_own;   // position of the chunk's first row in its column list (ColComp::own)
  // launches inside the PCG loop return at once after convergence; MODE 0 / 7 are also used outside it (right-hand
  // side, debug products), where the launcher passes no `scal`
  if ((MODE == 3 || MODE == 4 || MODE == 6 || MODE == 9) && scal->done) return;
  if ((MODE == 0 || MODE == 7) && scal != nullptr && scal->done) return;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  double beta = 0.0;
  bool first9 = false;
  if (MODE == 9) {
    first9 = scal->first != 0;
    // PCG iteration head: x = z (preconditioned residual).  Convergence test on the (D^-1 r)^2 partials
    // of the last update, beta = r.z(new)/r.z(old) from the two parity slots (part1), then in the row
    // loop  Ap <- A z + beta Ap,  p <- z + beta p  (direction update by recurrence) and p.Ap partials.
    if (!first9) {
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      for (int k = threadIdx.x; k < npart; k += TS) {
        v0 += part1[parity * MAXP + k];
        v1 += part1[(parity ^ 1) * MAXP + k];
        v2 += part2[k];
      }
      const double rz_new = block_sum<TS / 64>(v0, s4);
      const double rz_old = block_sum<TS / 64>(v1, s4);
      const double zz = block_sum<TS / 64>(v2, s4);
      const bool conv = zz <= scal->tol2;
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->zz = zz;
        if (conv) scal->done = 1;
      }
      if (conv) return;
      beta = rz_new / rz_old;
    }
  }
  // first kernel of a V-cycle inside the PCG loop: the update that precedes it left the (D^-1 r)^2 partials of the new
  // iterate - if that iterate has converged, the cycle (and every later launch of this burst) is skipped here instead of
  // after the cycle, by the next iteration head.  True: converged, the caller returns.
  auto cycle_gate = [&]() {
    if (!((MODE == 3 || MODE == 0) && part2 != nullptr)) return false;
    double v2 = 0.0;
    for (int k = threadIdx.x; k < npart; k += TS) v2 += part2[k];
    const double zz = block_sum<TS / 64>(v2, s4);
    const bool conv = zz <= scal->tol2;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (conv) { scal->zz = zz; scal->done = 1; }
      mirror_publish(scal->mirror, zz, scal->iters, conv ? 1 : 0, scal->epoch);
    }
    return conv;
  };
  constexpr bool PIPELINED = C16 && (HF_PIPE == 2 || (HF_PIPE == 1 && sizeof(VT) == 4));
  if (!PIPELINED && cycle_gate()) return;     // (the pipelined path requests its first chunk before it adds up the partials)
  const ChunkIter sched(nchunks);
  HF_STAMP(0);
  int stamp_at = 1;
  (void)stamp_at;
  // Chunks of fewer rows than threads (transfer operators with long rows, MODE 0 / 6 / 7 only): tpr = TS / rpc lanes
  // share a row's sum, so the row-sum phase uses every lane
  const int tpr = (MODE == 0 || MODE == 6 || MODE == 7) ? TS / rpc : 1;
  const int prel = static_cast<int>(threadIdx.x) / tpr;
  const int psub = static_cast<int>(threadIdx.x) % tpr;
  struct RowOps { double b, d, y, p, x; };
  // row-wise epilogue operands, requested early so that their latency hides under the chunk's stream.  Unconditional loads
  // at a clamped row: a load under a divergent branch makes the compiler's wait counters inexact - every later wait
  // would then cover all loads in flight, the prefetched ones of the next chunk included
  auto row_operands = [&](int prc, bool from_slice) {
    RowOps e{0.0, 0.0, 0.0, 0.0, 0.0};
    if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8) e.b = bvec[prc];
    if (MODE == 2 || MODE == 4 || MODE == 5) e.d = dinv[prc];
    if (MODE == 6 || (MODE == 9 && !first9)) e.y = y[prc];
    if (MODE == 9 && !first9) e.p = pvec[prc];
    if ((MODE == 4 || MODE == 8 || MODE == 9) && !from_slice) e.x = x[prc];
    return e;
  };
  // the chunk's products are parked in sprod: row sums in CSR order and the mode's epilogue
  auto rows_phase = [&](bool pin, int prow, int pa, int pb, const RowOps& e) {
    if ((MODE == 0 || MODE == 6 || MODE == 7) && tpr > 1) {
      // tpr is a power of two <= 16 (rpc >= 32): the lanes of a row are neighbours inside one wavefront
      double s = 0.0;
      if (pin)
        for (int j = pa + psub; j < pb; j += tpr) s += sprod[j];
      for (int o = tpr >> 1; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
      if (pin && psub == 0) {
        y[prow] = (MODE == 6) ? e.y + s : s;
        if (MODE == 7) acc0 += e.b * s;
      }
    } else if (pin) {
      const int row = prow;
      double s = 0.0;
      for (int j = pa; j < pb; ++j) s += sprod[j];
      if (MODE == 0) {
        y[row] = s;
      } else if (MODE == 2) {
        const double ri = e.b - s;
        const double zi = e.d * ri;
        y[row] = ri;
        pvec[row] = zi;
        acc0 += ri * zi;
        acc1 += zi * zi;
        acc2 += (e.d * e.b) * (e.d * e.b);
      } else if (MODE == 3) {
        y[row] = e.b - s;
      } else if (MODE == 4) {
        const double yi = e.x + w * e.d * (e.b - s);
        y[row] = yi;
        acc0 += e.b * yi;
      } else if (MODE == 5) {
        const double ri = e.b - s;
        y[row] = ri;
        pvec[row] = w * e.d * ri;
        acc1 += (e.d * ri) * (e.d * ri);
        acc2 += (e.d * e.b) * (e.d * e.b);
      } else if (MODE == 7) {
        y[row] = s;
        acc0 += e.b * s;
      } else if (MODE == 6) {
        y[row] = e.y + s;
      } else if (MODE == 8) {
        y[row] = s;
        pvec[row] = 2.0 * e.x - e.b;
      } else {
        const double api = first9 ? s : s + beta * e.y;          // first iteration: p = z, Ap = A z
        const double pi = first9 ? e.x : e.x + beta * e.p;
        y[row] = api;
        pvec[row] = pi;
        acc0 += pi * api;
      }
    }
  };
  if (PIPELINED) {
    // Software pipeline over the workgroup's chunks.  A chunk needs, in dependent order: its bounds -> the head of its
    // column list and its matrix stream (values, 16-bit positions) -> the gathered operand slice.  While chunk c is
    // worked on, the bounds, the list head and the first UN * TS entries of the stream of chunk c+1 are already in flight
    // (registers), so c+1 starts with its operand gathers and finds its stream landed.  Every load of the pipeline is
    // unconditional (clamped or parked index) - see row_operands.
    struct Bounds { int k0, k1, d0, d1; };
    auto bounds = [&](int ch) {
      Bounds q;
      const int a0 = ch * rpc, a1 = min(n, a0 + rpc);
      q.k0 = rowptr[a0]; q.k1 = rowptr[a1];
      q.d0 = comp.dptr[ch]; q.d1 = comp.dptr[ch + 1];
      return q;
    };
    int lhead[HF_STAGE_U];        // list head of the chunk about to start
    double v[UN];                 // its first UN * TS stream entries
    int id[UN];
    auto heads = [&](const Bounds& q, bool live) {
      // !live (no further chunk; q = the current one): every wavefront re-reads the chunk's first 32 entries, results
      // unused.  The offset stays a per-lane value on purpose: with a uniform parked index the compiler turns the loads
      // into scalar ones behind a branch and waits for every load in flight there.
      const int dl = max(q.d1 - 1, 0), kl = max(q.k1 - 1, 0);   // a chunk of empty rows has an empty list
      const int t = static_cast<int>(threadIdx.x);
#pragma unroll
      for (int u = 0; u < HF_STAGE_U; ++u) lhead[u] = comp.dict[min(q.d0 + (live ? t + u * TS : (t & 31)), dl)];
      __builtin_amdgcn_sched_barrier(0);   // list head first: the next chunk's gathers wait for it and for nothing younger
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int kk = min(q.k0 + (live ? t + u * TS : (t & 31)), kl);
        v[u] = static_cast<double>(stream_load(&vals[kk]));
        id[u] = static_cast<int>(stream_load(&comp.id[kk]));
      }
    };
    double* xd = sprod + comp.xd_off;
    Bounds cur = bounds(min(sched.chunk, nchunks - 1));
    heads(cur, sched.chunk < sched.end);
    if (cycle_gate()) return;       // the first chunk's requests are out: the test's reduction runs under their latency
    for (int chunk = sched.chunk; chunk < sched.end; chunk += sched.step) {
      const bool more = chunk + sched.step < sched.end;
      const int r0 = chunk * rpc, r1 = min(n, r0 + rpc);
      const int k0 = cur.k0, k1 = cur.k1, d0 = cur.d0, nd = cur.d1 - cur.d0;
      const int prow = r0 + prel;
      const bool pin = prow < r1;
      const int prc = min(prow, r1 - 1);
      // (1) operand gathers of the list head, then the next chunk's bounds, this lane's row bounds and row operands
      int c0[HF_STAGE_U];
      double x0[HF_STAGE_U];
#pragma unroll
      for (int u = 0; u < HF_STAGE_U; ++u) { c0[u] = lhead[u]; x0[u] = x[c0[u]]; }
      const Bounds nxt = bounds(more ? chunk + sched.step : chunk);
      const int pa = rowptr[prc] - k0, pb = rowptr[prc + 1] - k0;
      RowOps e = row_operands(prc, comp.own != 0);
      // (2) operand slice into LDS
#pragma unroll
      for (int u = 0; u < HF_STAGE_U; ++u) {
        const int i = static_cast<int>(threadIdx.x) + u * TS;
        if (i < nd) {
          xd[i] = x0[u];
          if ((MODE == 4 || MODE == 8 || MODE == 9) && c0[u] == r0) s_own = i;
        }
      }
      for (int i = threadIdx.x + HF_STAGE_U * TS; i < nd; i += HF_STAGE_U * TS) {   // lists longer than the head (rare)
        int c[HF_STAGE_U];
        double xv[HF_STAGE_U];
#pragma unroll
        for (int u = 0; u < HF_STAGE_U; ++u) c[u] = comp.dict[d0 + min(i + u * TS, nd - 1)];
#pragma unroll
        for (int u = 0; u < HF_STAGE_U; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < HF_STAGE_U; ++u)
          if (i + u * TS < nd) {
            xd[i + u * TS] = xv[u];
            if ((MODE == 4 || MODE == 8 || MODE == 9) && c[u] == r0) s_own = i + u * TS;
          }
      }
      __syncthreads();
      HF_STAMP(stamp_at); ++stamp_at;
      if ((MODE == 4 || MODE == 8 || MODE == 9) && comp.own && pin) e.x = xd[s_own + (prow - r0)];
      // (3) products in stream order
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int kk = k0 + static_cast<int>(threadIdx.x) + u * TS;
        if (kk < k1) sprod[kk - k0] = v[u] * xd[id[u]];
      }
      for (int kb = k0 + UN * TS; kb < k1; kb += UN * TS) {     // chunks longer than UN * TS entries (rare)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int kk = min(kb + static_cast<int>(threadIdx.x) + u * TS, k1 - 1);
          v[u] = static_cast<double>(stream_load(&vals[kk]));
          id[u] = static_cast<int>(stream_load(&comp.id[kk]));
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int kk = kb + static_cast<int>(threadIdx.x) + u * TS;
          if (kk < k1) sprod[kk - k0] = v[u] * xd[id[u]];
        }
      }
      // (4) the next chunk's list head and stream head go out; they land while the rows are summed and stored
      heads(nxt, more);
      __syncthreads();
      HF_STAMP(stamp_at); ++stamp_at;
      rows_phase(pin, prow, pa, pb, e);
      __syncthreads();
      HF_STAMP(stamp_at); ++stamp_at;
      cur = nxt;
    }
  } else {
  const ChunkIter& sched_ = sched;
  for (int chunk = sched_.chunk; chunk < sched_.end; chunk += sched_.step) {
    const int r0 = chunk * rpc;
    const int r1 = min(n, r0 + rpc);
    const int k0 = rowptr[r0];
    const int k1 = rowptr[r1];
    const int prow = r0 + prel;
    const bool pin = prow < r1;
    int pa = 0, pb = 0;
    if (pin) {
      pa = rowptr[prow] - k0;
      pb = rowptr[prow + 1] - k0;
    }
    RowOps e = row_operands(min(prow, r1 - 1), C16 && comp.own != 0);
    if (C16) {
      // first batch of values / 16-bit ids is requested before the operand slice is staged, so both latencies overlap
      double* xd = sprod + comp.xd_off;
      const int d0 = comp.dptr[chunk], nd = comp.dptr[chunk + 1] - d0;
      int k = k0 + threadIdx.x;
      double v[HF_UNROLL];
      int id[HF_UNROLL];
#pragma unroll
      for (int u = 0; u < HF_UNROLL; ++u) {
        const bool in = k + u * TS < k1;
        v[u] = in ? static_cast<double>(stream_load(&vals[k + u * TS])) : 0.0;
        id[u] = in ? static_cast<int>(stream_load(&comp.id[k + u * TS])) : 0;
      }
      // HF_STAGE_U entries of the column list per lane and pass (the fine operator's lists hold 1.3 TS entries, those of the
      // transfer operators up to 3.5 TS): the list loads go out together and the gathers after them - dependent round
      // trips per chunk: two instead of two per TS entries
      for (int i = threadIdx.x; i < nd; i += HF_STAGE_U * TS) {
        int c[HF_STAGE_U];
        double xv[HF_STAGE_U];
#pragma unroll
        for (int u = 0; u < HF_STAGE_U; ++u) c[u] = (i + u * TS < nd) ? comp.dict[d0 + i + u * TS] : 0;
#pragma unroll
        for (int u = 0; u < HF_STAGE_U; ++u) xv[u] = (i + u * TS < nd) ? x[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < HF_STAGE_U; ++u)
          if (i + u * TS < nd) {
            xd[i + u * TS] = xv[u];
            if ((MODE == 4 || MODE == 8 || MODE == 9) && c[u] == r0) s_own = i + u * TS;
          }
      }
      __syncthreads();
      HF_STAMP(stamp_at); ++stamp_at;
      if ((MODE == 4 || MODE == 8 || MODE == 9) && comp.own && pin) e.x = xd[s_own + (prow - r0)];
      while (k < k1) {
        const int kn = k + HF_UNROLL * TS;
        double vn[HF_UNROLL];
        int idn[HF_UNROLL];
#pragma unroll
        for (int u = 0; u < HF_UNROLL; ++u) {
          const bool in = kn + u * TS < k1;
          vn[u] = in ? static_cast<double>(stream_load(&vals[kn + u * TS])) : 0.0;
          idn[u] = in ? static_cast<int>(stream_load(&comp.id[kn + u * TS])) : 0;
        }
#pragma unroll
        for (int u = 0; u < HF_UNROLL; ++u)
          if (k + u * TS < k1) sprod[k - k0 + u * TS] = v[u] * xd[id[u]];
#pragma unroll
        for (int u = 0; u < HF_UNROLL; ++u) { v[u] = vn[u]; id[u] = idn[u]; }
        k = kn;
      }
    } else {
      // products in nnz order: HF_UNROLL predicated value/index loads, then the gathers, in flight per lane
      for (int k = k0 + threadIdx.x; k < k1; k += HF_UNROLL * TS) {
        int c[HF_UNROLL];
        double v[HF_UNROLL], xv[HF_UNROLL];
#pragma unroll
        for (int u = 0; u < HF_UNROLL; ++u) {
          const bool in = k + u * TS < k1;
          c[u] = in ? colidx[k + u * TS] : 0;
          v[u] = in ? static_cast<double>(vals[k + u * TS]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < HF_UNROLL; ++u) xv[u] = (k + u * TS < k1) ? x[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < HF_UNROLL; ++u)
          if (k + u * TS < k1) sprod[k - k0 + u * TS] = v[u] * xv[u];
      }
    }
    __syncthreads();
    HF_STAMP(stamp_at); ++stamp_at;
    rows_phase(pin, prow, pa, pb, e);
    __syncthreads();
    HF_STAMP(stamp_at); ++stamp_at;
  }
  }
  // consumers sum `npart` slots in a fixed order; this launch has fewer workgroups, the rest are zeros
  if (MODE == 2 || MODE == 7 || MODE == 9 || (MODE == 4 && part0 != nullptr)) {
    const double t0 = block_sum<TS / 64>(acc0, s4);
    if (threadIdx.x == 0) {
      part0[blockIdx.x] = t0;
      for (int q = blockIdx.x + gridDim.x; q < npart; q += gridDim.x) part0[q] = 0.0;
    }
  }
  if (MODE == 2 || MODE == 5) {
    const double t1 = block_sum<TS / 64>(acc1, s4);
    const double t2 = block_sum<TS / 64>(acc2, s4);
    if (threadIdx.x == 0) {
      part1[blockIdx.x] = t1;
      part2[blockIdx.x] = t2;
      for (int q = blockIdx.x + gridDim.x; q < npart; q += gridDim.x) { part1[q] = 0.0; part2[q] = 0.0; }
    }
  }
}

// PCG start: tolerance and convergence of the initial iterate (one workgroup).
__global__ __launch_bounds__(TPB) void k_pcg_begin(int P, double rtol, double atol, const double* __restrict__ part_zz,
                                                   const double* __restrict__ part_bn, Scal* __restrict__ scal, unsigned epoch) {
  __shared__ double s4[4];
  const double zz = sum_partials(part_zz, P, s4);
  const double bn2 = sum_partials(part_bn, P, s4);
  if (threadIdx.x == 0) {
    // relative to the right-hand side; a zero right-hand side (the exact answer is then zero) with a non-zero start vector
    // - a warm-started projection of a field that has gone flat - is measured against the start residual instead
    const double tol = fmax(rtol * sqrt(bn2 > 0.0 ? bn2 : zz), atol);
    scal->tol2 = tol * tol;
    scal->bn2 = bn2;
    scal->zz = zz;
    scal->iters = 0;
    scal->first = 1;
    scal->done = (zz <= tol * tol) ? 1 : 0;
    scal->epoch = epoch;
    if (scal->mirror != nullptr) scal->mirror->bn2 = bn2;
    mirror_publish(scal->mirror, zz, 0, scal->done, epoch);
  }
}

// x += alpha p; r -= alpha Ap; z = D^-1 r; partials r.z (into the other parity slot), z.z
__global__ __launch_bounds__(TPB) void k_pcg_update(int n, int nchunks, int P, int parity, Scal* __restrict__ scal,
                                                    const double* __restrict__ part_pAp, double* __restrict__ part_rz,
                                                    double* __restrict__ part_zz, double* __restrict__ x,
                                                    double* __restrict__ r, const double* __restrict__ p,
                                                    const double* __restrict__ Ap, const double* __restrict__ dinv,
                                                    double* __restrict__ z) {
  __shared__ double s4[4];
  if (scal->done) return;
  const double pAp = sum_partials(part_pAp, P, s4);
  const double rz = sum_partials(part_rz + parity * MAXP, P, s4);
  if (!(pAp > 0.0)) {                                   // breakdown (A_hat is SPD, so only on NaN/garbage)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      scal->done = 2;
      mirror_breakdown(scal->mirror, scal->epoch);
    }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal->iters += 1; scal->first = 0; }
  const double alpha = rz / pAp;
  double a_rz = 0.0, a_zz = 0.0;
  const ChunkIter sched(nchunks);
  for (int chunk = sched.chunk; chunk < sched.end; chunk += sched.step) {
    const int i = chunk * RB + threadIdx.x;
    if (i < n) {
      const double ri = r[i] - alpha * Ap[i];
      const double zi = dinv[i] * ri;
      x[i] += alpha * p[i];
      r[i] = ri;
      z[i] = zi;
      a_rz += ri * zi;
      a_zz += zi * zi;
    }
  }
  const double t0 = block_sum(a_rz, s4);
  const double t1 = block_sum(a_zz, s4);
  if (threadIdx.x == 0) {
    part_rz[(parity ^ 1) * MAXP + blockIdx.x] = t0;
    part_zz[blockIdx.x] = t1;
  }
}

// ------------------------------------------------------------------------------------------
// Generic CSR SpMV for the multigrid transfer operators and coarse levels: LANES lanes of a
// wavefront share one row (4..64 by the average row length), four predicated loads + gathers in
// flight per lane (the fused legs have rows of hundreds of entries), fixed-order shuffle reduction.
//   VMODE 0: y = A x      1: y += A x
// ------------------------------------------------------------------------------------------
template <int LANES, int VMODE, typename VT = double>
__global__ __launch_bounds__(TPB) void k_spmv_vec(int nrow, const int32_t* __restrict__ ptr,
                                                  const int32_t* __restrict__ idx, const VT* __restrict__ val,
                                                  const double* __restrict__ x, double* __restrict__ y,
                                                  const Scal* __restrict__ scal) {
  if (scal->done) return;
  const int lane = threadIdx.x % LANES;
  const int rows_per_pass = (gridDim.x * TPB) / LANES;
  for (int row = (blockIdx.x * TPB + threadIdx.x) / LANES; row < nrow; row += rows_per_pass) {
    const int k1 = ptr[row + 1];
    const double y0 = (VMODE == 1 && lane == 0) ? y[row] : 0.0;
    double s = 0.0;
    for (int k = ptr[row] + lane; k < k1; k += 4 * LANES) {
      int c[4];
      double v[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = k + u * LANES < k1;
        c[u] = in ? idx[k + u * LANES] : 0;
        v[u] = in ? static_cast<double>(val[k + u * LANES]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = (k + u * LANES < k1) ? x[c[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u] * xv[u];
    }
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) s += __shfl_down(s, o, LANES);
    if (lane == 0) y[row] = (VMODE == 1) ? y0 + s : s;
  }
}

// Same operation for operators with very long rows (the fused down leg onto a small level): a whole
// workgroup per row, four predicated loads + gathers in flight per lane, fixed-order block reduction.
template <int VMODE, typename VT = double>
__global__ __launch_bounds__(TPB) void k_spmv_row(int nrow, const int32_t* __restrict__ ptr,
                                                  const int32_t* __restrict__ idx, const VT* __restrict__ val,
                                                  const double* __restrict__ x, double* __restrict__ y,
                                                  const Scal* __restrict__ scal) {
  __shared__ double s4[TPB / 64];
  if (scal->done) return;
  for (int row = blockIdx.x; row < nrow; row += gridDim.x) {
    const int k1 = ptr[row + 1];
    const double y0 = (VMODE == 1 && threadIdx.x == 0) ? y[row] : 0.0;
    double s = 0.0;
    for (int k = ptr[row] + threadIdx.x; k < k1; k += 4 * TPB) {
      int c[4];
      double v[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = k + u * TPB < k1;
        c[u] = in ? idx[k + u * TPB] : 0;
        v[u] = in ? static_cast<double>(val[k + u * TPB]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = (k + u * TPB < k1) ? x[c[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u] * xv[u];
    }
    const double t = block_sum<TPB / 64>(s, s4);
    if (threadIdx.x == 0) y[row] = (VMODE == 1) ? y0 + t : t;
  }
}

// start vector: u = base + sum_k c_k w_k   (base = 2 u^n - u^{n-1}; w_k = boundary responses)
struct RespArgs { const double* w[MAXRESP]; double c[MAXRESP]; int k; };
__global__ __launch_bounds__(TPB) void k_start_vector(int n, const double* __restrict__ base, RespArgs a,
                                                      double* __restrict__ u) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    double s = base[i];
    for (int k = 0; k < a.k; ++k) s += a.c[k] * a.w[k][i];
    u[i] = s;
  }
}

// ------------------------------------------------------------------------------------------
// Start vector by Galerkin projection (hf_set_start_vector kind 3; Fischer 1998, "Projection techniques for
// iterative solution of Ax = b with successive right-hand sides").  The free part of each of the last few
// solutions u^k solves A_ff v = f^k with a known right-hand side, and so do the boundary responses w = R d.  The
// start vector of the next solve is the combination of those vectors that is closest to the new solution in the
// A-norm: G alpha = h with G_kl = V_k . F_l, h_k = V_k . f.  V_k holds u^k with its Dirichlet entries zeroed (so
// every dot product runs over the free rows only), F_k the right-hand side b^k as it was.
// ------------------------------------------------------------------------------------------
// (PROJ_MH solutions kept + MAXRESP boundary responses = PROJ_MT vectors: hf_context.hpp; HF_PROJ_MH for A/B builds)
struct ProjVecs { const double* V[PROJ_MT]; int slot[PROJ_MT]; int m; };

// partial sums of h_k = V_k . f and (Fnew != null) of the Gram column g_k = V_k . Fnew, one pass over all vectors
__global__ __launch_bounds__(TPB) void k_proj_dots(int n, ProjVecs a, const double* __restrict__ f, const double* __restrict__ Fnew,
                                                   double* __restrict__ part /* [2 * PROJ_MT][MAXP] */) {
  __shared__ double sw[TPB / 64][2 * PROJ_MT];
  double ah[PROJ_MT], ag[PROJ_MT];
#pragma unroll
  for (int k = 0; k < PROJ_MT; ++k) { ah[k] = 0.0; ag[k] = 0.0; }
  // two rows per pass: the loads of both (up to 2 m + 4) are in flight together - the stored vectors come from HBM
  const int stride = gridDim.x * TPB;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += 2 * stride) {
    const int i2 = i + stride;
    const bool two = i2 < n;
    const double fi = f ? f[i] : 0.0, gi = Fnew ? Fnew[i] : 0.0;
    const double fj = (f && two) ? f[i2] : 0.0, gj = (Fnew && two) ? Fnew[i2] : 0.0;
    double v0[PROJ_MT], v1[PROJ_MT];
#pragma unroll
    for (int k = 0; k < PROJ_MT; ++k) {
      v0[k] = k < a.m ? a.V[k][i] : 0.0;
      v1[k] = (k < a.m && two) ? a.V[k][i2] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PROJ_MT; ++k) {
      ah[k] += v0[k] * fi;
      ag[k] += v0[k] * gi;
      ah[k] += v1[k] * fj;
      ag[k] += v1[k] * gj;
    }
  }
  // all 2m sums through one barrier: shuffle tree per wavefront, then the four wave sums in wave order (block_sum's order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < PROJ_MT; ++k)
    if (k < a.m) {
      double th = ah[k], tg = ag[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { th += __shfl_down(th, o, 64); tg += __shfl_down(tg, o, 64); }
      if (lane == 0) { sw[wave][2 * k] = th; sw[wave][2 * k + 1] = tg; }
    }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < 2 * a.m) {
    double t = sw[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < TPB / 64; ++w) t += sw[w][threadIdx.x];
    part[threadIdx.x * MAXP + blockIdx.x] = t;
  }
}

// One workgroup: finish the sums (a wavefront per sum, fixed order), store the Gram column of slot `jnew` (if >= 0),
// and (do_solve) solve the scaled normal equations by symmetric elimination with diagonal pivoting (the matrix is
// positive semi-definite, so its largest remaining entry sits on the diagonal); directions whose pivot falls below
// 1e-12 of the first are left out (nearly dependent solutions).  alpha[slot] receives the coefficients (0 for
// slots left out), alpha[PROJ_MT] the rank.
static_assert(PROJ_MT * PROJ_MT <= 1024, "k_proj_solve holds one matrix entry per thread");
constexpr int PROJ_SOLVE_T = 1024;   // 16 wavefronts: one or two partial arrays each (the sums are the long part of the kernel)
__global__ __launch_bounds__(PROJ_SOLVE_T) void k_proj_solve(int P, ProjVecs a, int jnew, int do_solve, double* __restrict__ part,
                                                    double* __restrict__ G /* [PROJ_MT][PROJ_MT] by slot */, double* __restrict__ alpha) {
  __shared__ double sh[PROJ_MT], sg[PROJ_MT];
  __shared__ double A[PROJ_MT][PROJ_MT + 1], bb[PROJ_MT], dd[PROJ_MT], xx[PROJ_MT];
  __shared__ int perm[PROJ_MT], piv, stop, rank_s;
  __shared__ double pmax;
  // batched loop: one workgroup per column, each with its own partial sums, Gram matrix and coefficients
  part += static_cast<size_t>(blockIdx.x) * 2 * PROJ_MT * MAXP;
  G += blockIdx.x * PROJ_MT * PROJ_MT;
  alpha += blockIdx.x * (PROJ_MT + 1);
  const int m = a.m, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int q = wave; q < 2 * m; q += PROJ_SOLVE_T / 64) {  // sum q: partial array q of `part`
    double v = 0.0;
    for (int k0 = 0; k0 < P; k0 += 256) {
      double e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int k = k0 + lane + 64 * u; e[u] = k < P ? part[q * MAXP + k] : 0.0; }
      v += (e[0] + e[1]) + (e[2] + e[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) { if (q & 1) sg[q >> 1] = v; else sh[q >> 1] = v; }
  }
  __syncthreads();
  if (jnew >= 0 && t < m) { G[a.slot[t] * PROJ_MT + jnew] = sg[t]; G[jnew * PROJ_MT + a.slot[t]] = sg[t]; }
  if (!do_solve) return;
  __syncthreads();                                          // the column just written is read below (same workgroup)
  if (t < PROJ_MT + 1) alpha[t] = 0.0;
  if (t < m) {
    double gii = (jnew >= 0 && a.slot[t] == jnew) ? sg[t] : G[a.slot[t] * PROJ_MT + a.slot[t]];
    const bool ok = gii > 0.0 && gii < 1e300;
    dd[t] = ok ? 1.0 / sqrt(gii) : 0.0;
    perm[t] = t;
    xx[t] = 0.0;
  }
  __syncthreads();
  const int i = t / PROJ_MT, j = t % PROJ_MT;
  if (i < m && j < m) {
    // entries of the new column come from shared memory (global writes of this launch are not re-read)
    double gij;
    if (jnew >= 0 && a.slot[j] == jnew) gij = sg[i];
    else if (jnew >= 0 && a.slot[i] == jnew) gij = sg[j];
    else gij = G[a.slot[i] * PROJ_MT + a.slot[j]];
    const bool ok = dd[i] > 0.0 && dd[j] > 0.0;
    A[i][j] = ok ? gij * dd[i] * dd[j] : (i == j ? 0.0 : 0.0);
  }
  if (t < m) bb[t] = sh[t] * dd[t];
  if (t == 0) { rank_s = 0; stop = 0; }
  __syncthreads();
  for (int c = 0; c < m; ++c) {
    if (t == 0) {
      int pi = c;
      double best = A[c][c];
      for (int q = c + 1; q < m; ++q)
        if (A[q][q] > best) { best = A[q][q]; pi = q; }
      if (c == 0) pmax = best;
      piv = pi;
      stop = !(best > 1e-12 * pmax) || !(best > 0.0);
    }
    __syncthreads();
    if (stop) break;
    const int pv = piv;
    if (pv != c) {                                          // symmetric interchange c <-> pv
      if (t < m) { const double x0 = A[c][t]; A[c][t] = A[pv][t]; A[pv][t] = x0; }
      __syncthreads();
      if (t < m) { const double x0 = A[t][c]; A[t][c] = A[t][pv]; A[t][pv] = x0; }
      if (t == 0) {
        const double x0 = bb[c]; bb[c] = bb[pv]; bb[pv] = x0;
        const int p0 = perm[c]; perm[c] = perm[pv]; perm[pv] = p0;
      }
      __syncthreads();
    }
    if (i > c && i < m && j > c && j < m) A[i][j] -= A[i][c] / A[c][c] * A[c][j];
    if (i > c && i < m && j == c) bb[i] -= A[i][c] / A[c][c] * bb[c];
    if (t == 0) rank_s = c + 1;
    __syncthreads();
  }
  if (t != 0) return;
  const int rank = rank_s;
  for (int c = rank - 1; c >= 0; --c) {
    double v = bb[c];
    for (int q = c + 1; q < rank; ++q) v -= A[c][q] * xx[q];
    xx[c] = v / A[c][c];
  }
  bool finite = true;
  for (int c = 0; c < rank; ++c) finite = finite && (fabs(xx[c]) < 1e300);
  if (!finite) return;                      // alpha = 0: the start vector falls back to the boundary values alone
  for (int c = 0; c < rank; ++c) alpha[a.slot[perm[c]]] = xx[c] * dd[perm[c]];
  alpha[PROJ_MT] = static_cast<double>(rank);
}

// u = sum_k alpha[slot_k] V_k   (Dirichlet entries come out zero: set_bc writes them afterwards)
__global__ __launch_bounds__(TPB) void k_proj_combine(int n, ProjVecs a, const double* __restrict__ alpha, double* __restrict__ u) {
  double c[PROJ_MT];
#pragma unroll
  for (int k = 0; k < PROJ_MT; ++k) c[k] = k < a.m ? alpha[a.slot[k]] : 0.0;
  const int stride = gridDim.x * TPB;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += 2 * stride) {      // two rows per pass, as k_proj_dots
    const int i2 = i + stride;
    const bool two = i2 < n;
    double v0[PROJ_MT], v1[PROJ_MT];
#pragma unroll
    for (int k = 0; k < PROJ_MT; ++k) {
      v0[k] = k < a.m ? a.V[k][i] : 0.0;
      v1[k] = (k < a.m && two) ? a.V[k][i2] : 0.0;
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < PROJ_MT; ++k) { s0 += c[k] * v0[k]; s1 += c[k] * v1[k]; }
    u[i] = s0;
    if (two) u[i2] = s1;
  }
}

// two vector copies in one launch (projection ring: solution and right-hand side)
__global__ __launch_bounds__(TPB) void k_copy2(int n, const double* __restrict__ a, double* __restrict__ a_out,
                                               const double* __restrict__ b, double* __restrict__ b_out) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    a_out[i] = a[i];
    b_out[i] = b[i];
  }
}

__global__ void k_zero_entries(int nq, const int32_t* __restrict__ idx, double* __restrict__ v) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) v[idx[q]] = 0.0;
}

// Streaming read of two arrays with 16-byte loads (HF_K_STREAM_READ: the bandwidth ceiling SpMV is measured against)
__global__ __launch_bounds__(TS) void k_stream_read(size_t n16a, const double2* __restrict__ a, size_t n16b,
                                                    const double2* __restrict__ b, double* __restrict__ sink) {
  double s = 0.0;
  const size_t stride = static_cast<size_t>(gridDim.x) * TS;
  for (int pass = 0; pass < 2; ++pass) {
    const double2* __restrict__ p = pass ? b : a;
    const size_t n16 = pass ? n16b : n16a;
    size_t i = static_cast<size_t>(blockIdx.x) * TS + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
      const double2 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
      s += (v0.x + v0.y) + (v1.x + v1.y) + (v2.x + v2.y) + (v3.x + v3.y);
    }
    for (; i < n16; i += stride) { const double2 v = p[i]; s += v.x + v.y; }
  }
  if (s == 1.2345678e301) sink[0] = s;   // never true for real data; keeps the loads alive
}

// x = w D^-1 b  (first damped-Jacobi sweep from a zero guess)
__global__ __launch_bounds__(TPB) void k_scale(int n, double w, const double* __restrict__ dinv,
                                               const double* __restrict__ b, double* __restrict__ x,
                                               const Scal* __restrict__ scal) {
  if (scal->done) return;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) x[i] = w * dinv[i] * b[i];
}

// x = Ainv b with the dense inverse of the coarsest operator (row-major, leading dimension ld, even):
// two wavefronts per row, 16-byte loads, halves combined through LDS.
__global__ __launch_bounds__(TPB) void k_dense_mv(int n, int ld, const double* __restrict__ Ainv,
                                                  const double* __restrict__ b, double* __restrict__ x,
                                                  const Scal* __restrict__ scal) {
  __shared__ double half_sum[4];
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;            // 0..3: waves 0,1 -> row 2*blk, waves 2,3 -> row 2*blk + 1
  const int npair = (n + 1) >> 1;
  for (int pr = blockIdx.x; pr < npair; pr += gridDim.x) {
    const int row = 2 * pr + (wave >> 1);
    double s = 0.0;
    if (row < n) {
      const double2* arow = reinterpret_cast<const double2*>(Ainv + static_cast<size_t>(row) * ld);
      const double2* bv = reinterpret_cast<const double2*>(b);
      const int nv = ld >> 1;
      for (int j = (wave & 1) * 64 + lane; j < nv; j += 128) {
        const double2 a = arow[j];
        const double2 v = bv[j];
        s += a.x * v.x + a.y * v.y;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) half_sum[wave] = s;
    __syncthreads();
    if (lane == 0 && (wave & 1) == 0 && row < n) x[row] = half_sum[wave] + half_sum[wave + 1];
    __syncthreads();
  }
}

// The same with the inverse stored in float (the preconditioner's operators are kept in single precision, the
// vectors and every accumulation stay double): 16-byte loads of four entries.
__global__ __launch_bounds__(TPB) void k_dense_mv_f32(int n, int ld /* multiple of 4 */, const float* __restrict__ Ainv,
                                                      const double* __restrict__ b, double* __restrict__ x,
                                                      const Scal* __restrict__ scal) {
  __shared__ double half_sum[4];
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int npair = (n + 1) >> 1;
  for (int pr = blockIdx.x; pr < npair; pr += gridDim.x) {
    const int row = 2 * pr + (wave >> 1);
    double s = 0.0;
    if (row < n) {
      const float4* arow = reinterpret_cast<const float4*>(Ainv + static_cast<size_t>(row) * ld);
      const double2* bv = reinterpret_cast<const double2*>(b);
      const int nv = ld >> 2;
      for (int j = (wave & 1) * 64 + lane; j < nv; j += 128) {
        const float4 a = arow[j];
        const double2 v0 = bv[2 * j], v1 = bv[2 * j + 1];
        s += static_cast<double>(a.x) * v0.x + static_cast<double>(a.y) * v0.y + static_cast<double>(a.z) * v1.x + static_cast<double>(a.w) * v1.y;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) half_sum[wave] = s;
    __syncthreads();
    if (lane == 0 && (wave & 1) == 0 && row < n) x[row] = half_sum[wave] + half_sum[wave + 1];
    __syncthreads();
  }
}

__global__ void k_to_float(size_t n, const double* __restrict__ src, float* __restrict__ dst) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] = static_cast<float>(src[i]);
}

// Dense inverse of the coarsest operator on the GPU: blocked Gauss-Jordan without pivoting (the operator is SPD, its pivots
// stay positive).  W = [A | I] (n x 2n, row pitch 2n) is reduced to [I | A^-1], GJ_B pivots per step.  At the step of pivot block
// [c0, c0 + b) only the n columns [c0 + b, n + c0 + b) can still change (those to the left of the block are finished unit columns
// and never read again, those to the right of the window still hold the identity's zeros), so a step is
//   k_gjb_rows:   R = D^-1 W(block rows, window), D = W(block, block) inverted in LDS by every workgroup for itself
//   k_gjb_update: W(i, window) -= W(i, block) R for the rows outside the block, W(block rows, window) = R
// - n^2 entries read and written per GJ_B pivots instead of 2 n^2 per pivot (1634 rows: 85 MB per pivot before, 30 ms in all).
constexpr int GJ_B = 32;
__global__ __launch_bounds__(TPB) void k_gjb_rows(int n, int c0, int b, const double* __restrict__ W, double* __restrict__ R /* [GJ_B][n] */) {
  __shared__ double D[GJ_B][GJ_B + 1], Di[GJ_B][GJ_B + 1];
  const int t = threadIdx.x, ld = 2 * n;
  for (int q = t; q < GJ_B * GJ_B; q += TPB) {
    const int r = q / GJ_B, c = q % GJ_B;
    D[r][c] = (r < b && c < b) ? W[static_cast<size_t>(c0 + r) * ld + c0 + c] : (r == c ? 1.0 : 0.0);
    Di[r][c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int p = 0; p < b; ++p) {                    // Gauss-Jordan on [D | Di], one pivot per round
    const double piv = D[p][p];
    __syncthreads();
    for (int q = t; q < 2 * GJ_B; q += TPB) {
      if (q < GJ_B) D[p][q] /= piv; else Di[p][q - GJ_B] /= piv;
    }
    __syncthreads();
    for (int q = t; q < GJ_B * 2 * GJ_B; q += TPB) {
      const int r = q / (2 * GJ_B), c = q % (2 * GJ_B);
      if (r == p) continue;
      const double f = D[r][p];
      if (c < GJ_B) { if (c != p) D[r][c] -= f * D[p][c]; } else Di[r][c - GJ_B] -= f * Di[p][c - GJ_B];
    }
    __syncthreads();
    for (int r = t; r < GJ_B; r += TPB)
      if (r != p) D[r][p] = 0.0;
    __syncthreads();
  }
  // this workgroup's columns of the window
  for (int j = blockIdx.x * TPB + t; j < n; j += gridDim.x * TPB) {
    const size_t col = static_cast<size_t>(c0 + b + j);
    double w[GJ_B];
#pragma unroll
    for (int r = 0; r < GJ_B; ++r) w[r] = r < b ? W[static_cast<size_t>(c0 + r) * ld + col] : 0.0;
    for (int r = 0; r < b; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < GJ_B; ++c) s += Di[r][c] * w[c];
      R[static_cast<size_t>(r) * n + j] = s;
    }
  }
}

// W = [A | I] from the coarsest operator in CSR form (W zeroed before): a thread per row
__global__ void k_gjb_fill(int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ val,
                           double* __restrict__ W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double* w = W + static_cast<size_t>(i) * 2 * n;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k) w[idx[k]] = val[k];
  w[n + i] = 1.0;
}

// 64 x 64 tile of the window per workgroup, 4 x 4 entries per thread; the tile's slices of the pivot columns and of R in LDS
__global__ __launch_bounds__(TPB) void k_gjb_update(int n, int c0, int b, double* __restrict__ W, const double* __restrict__ R) {
  __shared__ double sC[64][GJ_B + 1], sR[GJ_B][64 + 1];
  const int ld = 2 * n, t = threadIdx.x;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  for (int q = t; q < 64 * GJ_B; q += TPB) {
    const int r = q / GJ_B, c = q % GJ_B;
    sC[r][c] = (i0 + r < n && c < b) ? W[static_cast<size_t>(i0 + r) * ld + c0 + c] : 0.0;
  }
  for (int q = t; q < GJ_B * 64; q += TPB) {
    const int r = q / 64, c = q % 64;
    sR[r][c] = (r < b && j0 + c < n) ? R[static_cast<size_t>(r) * n + j0 + c] : 0.0;
  }
  __syncthreads();
  const int ti = (t / 16) * 4, tj = (t % 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int k = 0; k < GJ_B; ++k) {
    double cu[4], rv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) cu[u] = sC[ti + u][k];
#pragma unroll
    for (int v = 0; v < 4; ++v) rv[v] = sR[k][tj + v];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += cu[u] * rv[v];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + ti + u;
    if (i >= n) continue;
    const bool pivot_row = i >= c0 && i < c0 + b;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = j0 + tj + v;
      if (j >= n) continue;
      double* w = W + static_cast<size_t>(i) * ld + c0 + b + j;
      *w = pivot_row ? sR[i - c0][tj + v] : *w - acc[u][v];
    }
  }
}

// AMG-PCG: x += alpha p; r -= alpha Ap; z0 = w D^-1 r (pre-smoothed start of the V-cycle);
// partial (D^-1 r)^2 for the convergence test.  r.z comes from the V-cycle's last kernel.
__global__ __launch_bounds__(TPB) void k_pcg_update_amg(int n, int nchunks, int P, int parity, Scal* __restrict__ scal,
                                                        const double* __restrict__ part_pAp,
                                                        const double* __restrict__ part_rz, double* __restrict__ part_zz,
                                                        double* __restrict__ x, double* __restrict__ r,
                                                        const double* __restrict__ p, const double* __restrict__ Ap,
                                                        const double* __restrict__ dinv, double w, double* __restrict__ z0) {
  __shared__ double s4[4];
  if (scal->done) return;
  const double pAp = sum_partials(part_pAp, P, s4);
  const double rz = sum_partials(part_rz + parity * MAXP, P, s4);
  if (!(pAp > 0.0)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      scal->done = 2;
      mirror_breakdown(scal->mirror, scal->epoch);
    }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal->iters += 1; scal->first = 0; }
  const double alpha = rz / pAp;
  double a_zz = 0.0;
  const ChunkIter sched(nchunks);
  for (int chunk = sched.chunk; chunk < sched.end; chunk += sched.step) {
    const int i = chunk * RB + threadIdx.x;
    if (i < n) {
      const double ri = r[i] - alpha * Ap[i];
      const double zi = dinv[i] * ri;
      x[i] += alpha * p[i];
      r[i] = ri;
      if (z0 != nullptr) z0[i] = w * zi;      // not needed when the finest level runs through its fused legs
      a_zz += zi * zi;
    }
  }
  const double t1 = block_sum(a_zz, s4);
  if (threadIdx.x == 0) part_zz[blockIdx.x] = t1;
}

}  // namespace
