// Part of libheatflow_hip.so (see heatflow_hip.hip): the algebraic part of the multigrid set-up on the GPU.
//
// amg_host.hpp builds the smoothed-aggregation hierarchy on one host thread; its sparse products (A P, R (A P)), the
// transposes and the fused legs are 0.36 s of the 0.47 s the set-up of the 1M-DOF mesh takes and 10 s at 16M DOF - more
// than the time loop they prepare.  Here the same operators are formed on the device, level by level, for every level
// whose operator has rows of at most GA_MAXROW entries (the big ones); what is left for the host is the greedy
// aggregation (sequential by nature, 40 ms at 1M rows), the diagonal / Gershgorin pass and the small coarse levels.
// Every product is evaluated in the order the host routines use (per output entry: contributions in ascending order of the
// left operand's column, no FMA contraction), so the hierarchy is bit-identical to the host-built one
// (tests: hf_amg_export of both builds compared byte for byte).
//   k_amg_prolong      P = (I - w D^-1 A) T                     thread per row          (amg::smoothed_prolongator)
//   gpu_transpose      R = P^T, Rt = Pt^T                       stable radix sort by column (hipCUB) + gather
//   k_spgemm           C = A B                                  wavefront per row: candidate columns sorted in LDS,
//                                                               one lane per output entry  (amg::spgemm)
//   k_amg_smooth_p     Pt = P - w D^-1 (A P)                    thread per row          (amg::smoothed_by_product)
//   k_amg_up_leg       GP = [2wD^-1 - w^2 D^-1 A D^-1 | Pt]     thread per row          (amg::fused_up_leg)
//   k_coldict          compressed column stream of the LDS-staged SpMV kernel: per chunk a sorted column list and a
//                      16-bit position per nonzero              workgroup per chunk     (build_coldict)
// The reference has no counterpart (it factors with MUMPS on the host, run_with_diamond.py:389-394).
#pragma once
#include <hipcub/hipcub.hpp>

#include "hf_solver.hpp"

namespace {

constexpr int GA_MAXROW = 32;      // longest row of a level operator the thread-per-row kernels take
constexpr int GA_CAP = 4096;       // candidate columns of one output row of a product (a wavefront's LDS buffer)

struct GCsr {                      // device CSR, double values, sorted columns
  int nrow = 0, ncol = 0;
  int64_t nnz = 0;
  int32_t *ptr = nullptr, *idx = nullptr;
  double* val = nullptr;
  bool owned = true;               // false: aliases the context's fine operator
};

void gfree(GCsr& m) {
  if (m.owned) { dev_free(&m.ptr); dev_free(&m.idx); dev_free(&m.val); }
  m = GCsr();
}

struct GpuScratch {                // temporary storage of the hipCUB calls, grown on demand
  void* p = nullptr;
  size_t bytes = 0;
  ~GpuScratch() { if (p) (void)hipFree(p); }
  int need(hf_ctx* ctx, size_t b) {
    if (b <= bytes) return HF_OK;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    if (hipMalloc(&p, b) != hipSuccess) return fail(ctx, HF_ERR_ALLOC, "hipMalloc(%zu bytes) failed (multigrid set-up scratch)", b);
    bytes = b;
    return HF_OK;
  }
};

// counts[0..n) -> exclusive prefix sums in place, counts[n] = total; returns the total on the host
int gpu_scan(hf_ctx* ctx, GpuScratch& S, int32_t* counts, int n, int64_t* total) {
  size_t b = 0;
  HF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b, counts, counts, n + 1, ctx->stream));
  HF_TRY(S.need(ctx, b));
  HF_HIP(hipcub::DeviceScan::ExclusiveSum(S.p, b, counts, counts, n + 1, ctx->stream));
  int32_t t = 0;
  HF_HIP(copy_sync(ctx, &t, counts + n, sizeof(int32_t), hipMemcpyDeviceToHost));
  *total = t;
  return HF_OK;
}

// ------------------------------------------------------------------------------------------
// P = (I - w D^-1 A) T, T = aggregate indicator (amg::smoothed_prolongator): thread per row, rows of <= GA_MAXROW entries
// ------------------------------------------------------------------------------------------
template <bool FILL>
__global__ void k_amg_prolong(int n, const int32_t* __restrict__ aptr, const int32_t* __restrict__ aidx, const double* __restrict__ aval,
                              const int32_t* __restrict__ agg, const double* __restrict__ diag, double w, int32_t* __restrict__ pptr,
                              int32_t* __restrict__ pidx, double* __restrict__ pval) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ai = agg[i];
  if (ai < 0) {                                   // rows without an aggregate stay empty
    if (!FILL) pptr[i] = 0;
    return;
  }
  int cols[GA_MAXROW + 1];
  double vals[GA_MAXROW + 1];
  int m = 1;
  cols[0] = ai;
  vals[0] = 1.0;
  const double s = -w / diag[i];
  for (int k = aptr[i]; k < aptr[i + 1]; ++k) {
    const int c = agg[aidx[k]];
    if (c < 0) continue;
    int q = 0;
    while (q < m && cols[q] != c) ++q;
    if (q == m) { cols[m] = c; vals[m] = 0.0; ++m; }
    vals[q] += s * aval[k];
  }
  if (!FILL) { pptr[i] = m; return; }
  for (int a = 1; a < m; ++a) {                   // insertion sort by column (m <= GA_MAXROW + 1)
    const int c = cols[a];
    const double v = vals[a];
    int b = a - 1;
    while (b >= 0 && cols[b] > c) { cols[b + 1] = cols[b]; vals[b + 1] = vals[b]; --b; }
    cols[b + 1] = c; vals[b + 1] = v;
  }
  const int o = pptr[i];
  for (int a = 0; a < m; ++a) { pidx[o + a] = cols[a]; pval[o + a] = vals[a]; }
}

// ------------------------------------------------------------------------------------------
// Transpose: stable sort of (column, position) pairs, then gather (rows of the result come out with ascending columns)
// ------------------------------------------------------------------------------------------
__global__ void k_iota_rows(int nrow, const int32_t* __restrict__ ptr, int32_t* __restrict__ pos, int32_t* __restrict__ rowof) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrow) return;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k) { pos[k] = k; rowof[k] = i; }
}
__global__ void k_count_cols(int64_t nnz, const int32_t* __restrict__ idx, int32_t* __restrict__ cnt) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k < nnz) atomicAdd(&cnt[idx[k]], 1);          // integer counts: order does not matter
}
__global__ void k_transpose_gather(int64_t nnz, const int32_t* __restrict__ pos_sorted, const int32_t* __restrict__ rowof,
                                   const double* __restrict__ val, int32_t* __restrict__ tidx, double* __restrict__ tval) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int p = pos_sorted[k];
  tidx[k] = rowof[p];
  tval[k] = val[p];
}

int gpu_transpose(hf_ctx* ctx, GpuScratch& S, const GCsr& A, GCsr& T) {
  T = GCsr();
  T.nrow = A.ncol; T.ncol = A.nrow; T.nnz = A.nnz;
  HF_TRY(dev_alloc(ctx, &T.ptr, static_cast<size_t>(T.nrow) + 1));
  HF_TRY(dev_alloc(ctx, &T.idx, static_cast<size_t>(A.nnz)));
  HF_TRY(dev_alloc(ctx, &T.val, static_cast<size_t>(A.nnz)));
  HF_HIP(hipMemsetAsync(T.ptr, 0, sizeof(int32_t) * (static_cast<size_t>(T.nrow) + 1), ctx->stream));
  if (A.nnz == 0) return HF_OK;
  DevTemp<int32_t> pos, rowof, keys_out, pos_out;
  HF_TRY(dev_alloc(ctx, &pos.p, static_cast<size_t>(A.nnz)));
  HF_TRY(dev_alloc(ctx, &rowof.p, static_cast<size_t>(A.nnz)));
  HF_TRY(dev_alloc(ctx, &keys_out.p, static_cast<size_t>(A.nnz)));
  HF_TRY(dev_alloc(ctx, &pos_out.p, static_cast<size_t>(A.nnz)));
  const unsigned gr = static_cast<unsigned>((A.nrow + 255) / 256), gk = static_cast<unsigned>((A.nnz + 255) / 256);
  hipLaunchKernelGGL(k_iota_rows, dim3(gr), dim3(256), 0, ctx->stream, A.nrow, A.ptr, pos.p, rowof.p);
  hipLaunchKernelGGL(k_count_cols, dim3(gk), dim3(256), 0, ctx->stream, A.nnz, A.idx, T.ptr);
  int64_t total = 0;
  HF_TRY(gpu_scan(ctx, S, T.ptr, T.nrow, &total));
  int bits = 1;
  while ((1LL << bits) < A.ncol) ++bits;
  size_t b = 0;
  HF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, b, A.idx, keys_out.p, pos.p, pos_out.p, static_cast<int>(A.nnz), 0, bits, ctx->stream));
  HF_TRY(S.need(ctx, b));
  HF_HIP(hipcub::DeviceRadixSort::SortPairs(S.p, b, A.idx, keys_out.p, pos.p, pos_out.p, static_cast<int>(A.nnz), 0, bits, ctx->stream));
  hipLaunchKernelGGL(k_transpose_gather, dim3(gk), dim3(256), 0, ctx->stream, A.nnz, pos_out.p, rowof.p, A.val, T.idx, T.val);
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

// ------------------------------------------------------------------------------------------
// C = A B (amg::spgemm): one wavefront per output row.  The candidate columns (the rows of B the row of A points at) are
// collected in the wavefront's LDS buffer, sorted (bitonic) and made unique; then lane t owns the t-th output entry and adds
// A_ik B_kj over the row of A in ascending k (the host's order), finding j in B's sorted row k by bisection.
// FILL = false counts the row's entries, FILL = true writes them.  A row with more than GA_CAP candidates raises `overflow`.
// ------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(64) void k_spgemm(int nrow, const int32_t* __restrict__ aptr, const int32_t* __restrict__ aidx,
                                               const double* __restrict__ aval, const int32_t* __restrict__ bptr,
                                               const int32_t* __restrict__ bidx, const double* __restrict__ bval,
                                               int32_t* __restrict__ cptr, int32_t* __restrict__ cidx, double* __restrict__ cval,
                                               int32_t* __restrict__ overflow) {
#pragma clang fp contract(off)
  __shared__ int32_t cand[GA_CAP];
  const int lane = threadIdx.x;
  for (int i = blockIdx.x; i < nrow; i += gridDim.x) {
    const int a0 = aptr[i], a1 = aptr[i + 1];
    int nc = 0;
    bool over = false;
    for (int kb = a0; kb < a1; kb += 64) {             // a lane per entry of A's row: its row of B goes to the lane's slice of the buffer
      const int k = kb + lane;
      int b0 = 0, len = 0;
      if (k < a1) { const int r = aidx[k]; b0 = bptr[r]; len = bptr[r + 1] - b0; }
      int incl = len;                                  // inclusive scan of the lengths over the wavefront
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      const int total = __shfl(incl, 63, 64);
      if (nc + total > GA_CAP) { over = true; break; }
      const int off = nc + incl - len;
      for (int q = 0; q < len; ++q) cand[off + q] = bidx[b0 + q];
      nc += total;
    }
    if (over) {
      if (lane == 0) { atomicExch(overflow, 1); if (!FILL) cptr[i] = 0; }
      continue;
    }
    int m = 1;
    while (m < nc) m <<= 1;
    for (int q = nc + lane; q < m; q += 64) cand[q] = INT32_MAX;
    __syncthreads();
    for (int size = 2; size <= m; size <<= 1)          // bitonic sort, ascending
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = lane; t < (m >> 1); t += 64) {
          const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const int x = cand[lo], y = cand[hi];
          if ((x > y) == up) { cand[lo] = y; cand[hi] = x; }
        }
        __syncthreads();
      }
    // unique: entry q survives if it differs from its predecessor; its output position = number of survivors before it
    int nout = 0;
    for (int base = 0; base < nc; base += 64) {
      const int q = base + lane;
      const bool keep = q < nc && (q == 0 || cand[q] != cand[q - 1]);
      const unsigned long long mask = __ballot(keep);
      const int before = __popcll(mask & ((1ull << lane) - 1ull));
      const int c = q < nc ? cand[q] : 0;
      __syncthreads();                                  // every lane has read cand[q - 1] of this block before anyone compacts into it
      if (keep) cand[nout + before] = c;                // compaction in place: nout + before <= q
      nout += __popcll(mask);
      __syncthreads();
    }
    if (!FILL) {
      if (lane == 0) cptr[i] = nout;
      __syncthreads();
      continue;
    }
    const int o = cptr[i];
    for (int t = lane; t < nout; t += 64) {
      const int j = cand[t];
      double v = 0.0;
      for (int k = a0; k < a1; ++k) {
        const int r = aidx[k];
        int lo = bptr[r], hi = bptr[r + 1];
        while (lo < hi) {                               // first entry of B's row r with column >= j
          const int mid = (lo + hi) >> 1;
          if (bidx[mid] < j) lo = mid + 1; else hi = mid;
        }
        if (lo < bptr[r + 1] && bidx[lo] == j) v += aval[k] * bval[lo];
      }
      cidx[o + t] = j;
      cval[o + t] = v;
    }
    __syncthreads();
  }
}

// C = A B; *fell_back = true (and C empty) when a row of the product has more candidate columns than the kernel takes
int gpu_spgemm(hf_ctx* ctx, GpuScratch& S, const GCsr& A, const GCsr& B, GCsr& C, bool* fell_back) {
  C = GCsr();
  *fell_back = false;
  C.nrow = A.nrow; C.ncol = B.ncol;
  HF_TRY(dev_alloc(ctx, &C.ptr, static_cast<size_t>(A.nrow) + 1));
  DevTemp<int32_t> over;
  HF_TRY(dev_alloc(ctx, &over.p, 1));
  HF_HIP(hipMemsetAsync(over.p, 0, sizeof(int32_t), ctx->stream));
  HF_HIP(hipMemsetAsync(C.ptr, 0, sizeof(int32_t) * (static_cast<size_t>(A.nrow) + 1), ctx->stream));
  const int grid = std::max(1, std::min(A.nrow, 256 * 40));
  hipLaunchKernelGGL(k_spgemm<false>, dim3(grid), dim3(64), 0, ctx->stream, A.nrow, A.ptr, A.idx, A.val, B.ptr, B.idx, B.val, C.ptr,
                     static_cast<int32_t*>(nullptr), static_cast<double*>(nullptr), over.p);
  HF_HIP(hipGetLastError());
  int32_t ov = 0;
  HF_HIP(copy_sync(ctx, &ov, over.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (ov) { gfree(C); *fell_back = true; return HF_OK; }
  HF_TRY(gpu_scan(ctx, S, C.ptr, A.nrow, &C.nnz));
  HF_TRY(dev_alloc(ctx, &C.idx, static_cast<size_t>(C.nnz)));
  HF_TRY(dev_alloc(ctx, &C.val, static_cast<size_t>(C.nnz)));
  hipLaunchKernelGGL(k_spgemm<true>, dim3(grid), dim3(64), 0, ctx->stream, A.nrow, A.ptr, A.idx, A.val, B.ptr, B.idx, B.val, C.ptr, C.idx,
                     C.val, over.p);
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

// ------------------------------------------------------------------------------------------
// Pt = P - w D^-1 (A P) on the pattern of A P (amg::smoothed_by_product; the pattern of P is contained in it because every
// row of A stores its diagonal); rows merge two sorted lists.  Thread per row.
// ------------------------------------------------------------------------------------------
__global__ void k_amg_smooth_p(int n, const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx, const double* __restrict__ pval,
                               const int32_t* __restrict__ qptr, const int32_t* __restrict__ qidx, const double* __restrict__ qval,
                               const double* __restrict__ dinv, double w, double* __restrict__ out /* on the pattern of A P */,
                               int32_t* __restrict__ bad) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int a = pptr[i];
  const int a1 = pptr[i + 1];
  const double s = -w * dinv[i];
  for (int b = qptr[i]; b < qptr[i + 1]; ++b) {
    const int cb = qidx[b];
    if (a < a1 && pidx[a] < cb) { atomicExch(bad, 1); return; }     // an entry of P outside the pattern of A P: not this kernel's case
    if (a < a1 && pidx[a] == cb) { out[b] = pval[a] + s * qval[b]; ++a; }
    else out[b] = s * qval[b];
  }
  if (a < a1) atomicExch(bad, 1);
}

// GP = [G | Pt], G = 2 w D^-1 - w^2 D^-1 A D^-1 on the pattern of A, the columns of Pt shifted by A.ncol (amg::fused_up_leg)
template <bool FILL>
__global__ void k_amg_up_leg(int n, int ncolA, const int32_t* __restrict__ aptr, const int32_t* __restrict__ aidx, const double* __restrict__ aval,
                             const int32_t* __restrict__ tptr, const int32_t* __restrict__ tidx, const double* __restrict__ tval,
                             const double* __restrict__ dinv, double w, int32_t* __restrict__ gptr, int32_t* __restrict__ gidx,
                             double* __restrict__ gval) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!FILL) { gptr[i] = (aptr[i + 1] - aptr[i]) + (tptr[i + 1] - tptr[i]); return; }
  int o = gptr[i];
  for (int k = aptr[i]; k < aptr[i + 1]; ++k, ++o) {
    const int j = aidx[k];
    gidx[o] = j;
    gval[o] = (j == i ? 2.0 * w * dinv[i] : 0.0) - w * w * dinv[i] * aval[k] * dinv[j];
  }
  for (int k = tptr[i]; k < tptr[i + 1]; ++k, ++o) { gidx[o] = ncolA + tidx[k]; gval[o] = tval[k]; }
}

// ------------------------------------------------------------------------------------------
// Compressed column stream of the LDS-staged SpMV kernel (build_coldict): a workgroup per chunk of `rpc` rows sorts the
// chunk's column indices in LDS, keeps the distinct ones (the chunk's column list) and gives every nonzero the 16-bit
// position of its column in that list.  Chunks hold at most GA_CAP nonzeros (the launcher chooses rpc that way).
// ------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void k_coldict(int nrow, int rpc, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                  int32_t* __restrict__ dptr, int32_t* __restrict__ dict, uint16_t* __restrict__ cid) {
  __shared__ int32_t buf[GA_CAP];
  __shared__ int s_cnt[4];
  const int chunk = blockIdx.x;
  const int r0 = chunk * rpc, r1 = min(nrow, r0 + rpc);
  const int k0 = ptr[r0], nk = ptr[r1] - k0;
  int m = 1;
  while (m < nk) m <<= 1;
  for (int q = threadIdx.x; q < m; q += 256) buf[q] = q < nk ? idx[k0 + q] : INT32_MAX;
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (m >> 1); t += 256) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int x = buf[lo], y = buf[hi];
        if ((x > y) == up) { buf[lo] = y; buf[hi] = x; }
      }
      __syncthreads();
    }
  // unique, in place: blocks of 256 entries, four wavefronts' ballots combined through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nout = 0;
  for (int base = 0; base < nk; base += 256) {
    const int q = base + threadIdx.x;
    const bool keep = q < nk && (q == 0 || buf[q] != buf[q - 1]);
    const int c = q < nk ? buf[q] : 0;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_cnt[wave] = __popcll(mask);
    __syncthreads();                                    // also: every thread has read buf[q - 1] before the compaction below
    int before = __popcll(mask & ((1ull << lane) - 1ull));
    for (int v = 0; v < wave; ++v) before += s_cnt[v];
    const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (keep) buf[nout + before] = c;
    nout += total;
    __syncthreads();
  }
  if (!FILL) {
    if (threadIdx.x == 0) dptr[chunk] = nout;
    return;
  }
  const int d0 = dptr[chunk];
  for (int q = threadIdx.x; q < nout; q += 256) dict[d0 + q] = buf[q];
  for (int q = threadIdx.x; q < nk; q += 256) {
    const int c = idx[k0 + q];
    int lo = 0, hi = nout;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (buf[mid] < c) lo = mid + 1; else hi = mid;
    }
    cid[k0 + q] = static_cast<uint16_t>(lo);
  }
}

__global__ void k_max_int(int n, const int32_t* __restrict__ a, int32_t* __restrict__ out) {
  int v = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v = max(v, a[i + 1] - a[i]);
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, v);
}

__global__ void k_double_to_float(int64_t n, const double* __restrict__ src, float* __restrict__ dst) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = static_cast<float>(src[i]);
}

// A device operator of the set-up becomes a DevCsr of the hierarchy (what upload_csr does for a host operator): values kept in
// double or converted to float, kernel geometry, and - for operators the LDS-staged kernel runs in float - the compressed
// column stream, built on the device.  `g`'s arrays move into `d` (idx, ptr; val unless converted).
int adopt_csr(hf_ctx* ctx, GpuScratch& S, GCsr& g, DevCsr& d, bool f32, bool stream = true) {
  static const int min_rows = std::getenv("HEATFLOW_STREAM_MIN_ROWS") ? std::atoi(std::getenv("HEATFLOW_STREAM_MIN_ROWS")) : 20000;
  static const int max_nnz = std::getenv("HEATFLOW_STREAM_NNZ") ? std::atoi(std::getenv("HEATFLOW_STREAM_NNZ")) : 4096;
  d = DevCsr();
  d.nrow = g.nrow; d.ncol = g.ncol; d.nnz = g.nnz;
  const double avg = g.nrow ? static_cast<double>(g.nnz) / g.nrow : 1.0;
  d.lanes = lanes_for_avg(avg);
  std::vector<int32_t> hptr(static_cast<size_t>(g.nrow) + 1);
  HF_HIP(copy_sync(ctx, hptr.data(), g.ptr, sizeof(int32_t) * hptr.size(), hipMemcpyDeviceToHost));
  for (int i = 0; i < g.nrow; ++i) d.max_row = std::max(d.max_row, hptr[i + 1] - hptr[i]);
  if (stream && g.nrow >= min_rows) {
    for (int rpc = TS; rpc >= 32; rpc /= 2) {
      int mx = 0;
      for (int r0 = 0; r0 < g.nrow; r0 += rpc) mx = std::max(mx, hptr[std::min(g.nrow, r0 + rpc)] - hptr[r0]);
      if (mx <= max_nnz) { d.rpc = rpc; d.nchunks = (g.nrow + rpc - 1) / rpc; d.chunk_nnz = mx; break; }
    }
  }
  d.ptr = g.ptr; d.idx = g.idx;
  g.ptr = nullptr; g.idx = nullptr;
  if (f32) {
    HF_TRY(dev_alloc(ctx, &d.valf, static_cast<size_t>(g.nnz)));
    if (g.nnz) hipLaunchKernelGGL(k_double_to_float, dim3(static_cast<unsigned>((g.nnz + 255) / 256)), dim3(256), 0, ctx->stream, g.nnz, g.val, d.valf);
    HF_HIP(hipGetLastError());
    HF_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(&g.val);
  } else {
    d.val = g.val;
    g.val = nullptr;
  }
  g = GCsr();
  if (f32 && d.rpc > 0 && d.chunk_nnz <= GA_CAP) {
    DevTemp<int32_t> dptr;
    HF_TRY(dev_alloc(ctx, &dptr.p, static_cast<size_t>(d.nchunks) + 1));
    HF_HIP(hipMemsetAsync(dptr.p, 0, sizeof(int32_t) * (static_cast<size_t>(d.nchunks) + 1), ctx->stream));
    hipLaunchKernelGGL(k_coldict<false>, dim3(d.nchunks), dim3(256), 0, ctx->stream, d.nrow, d.rpc, d.ptr, d.idx, dptr.p,
                       static_cast<int32_t*>(nullptr), static_cast<uint16_t*>(nullptr));
    HF_HIP(hipGetLastError());
    DevTemp<int32_t> mx;
    HF_TRY(dev_alloc(ctx, &mx.p, 1));
    HF_HIP(hipMemsetAsync(mx.p, 0, sizeof(int32_t), ctx->stream));
    int64_t ndict = 0;
    HF_TRY(gpu_scan(ctx, S, dptr.p, d.nchunks, &ndict));
    hipLaunchKernelGGL(k_max_int, dim3(64), dim3(256), 0, ctx->stream, d.nchunks, dptr.p, mx.p);
    int32_t max_dict = 0;
    HF_HIP(copy_sync(ctx, &max_dict, mx.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (max_dict <= 65535 && static_cast<size_t>(d.chunk_nnz + max_dict) * 8 <= 64 * 1024) {
      d.max_dict = max_dict;
      d.ndict = ndict;
      HF_TRY(dev_alloc(ctx, &d.dict, static_cast<size_t>(ndict)));
      HF_TRY(dev_alloc(ctx, &d.cid, static_cast<size_t>(d.nnz)));
      d.dptr = dptr.p;
      dptr.p = nullptr;
      hipLaunchKernelGGL(k_coldict<true>, dim3(d.nchunks), dim3(256), 0, ctx->stream, d.nrow, d.rpc, d.ptr, d.idx, d.dptr, d.dict, d.cid);
      HF_HIP(hipGetLastError());
      HF_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  return HF_OK;
}

GCsr gclone_shallow(const GCsr& a) { GCsr b = a; b.owned = false; return b; }

int gdownload(hf_ctx* ctx, const GCsr& g, amg::Csr& h) {
  h.nrow = g.nrow; h.ncol = g.ncol;
  h.ptr.resize(static_cast<size_t>(g.nrow) + 1);
  h.idx.resize(static_cast<size_t>(g.nnz));
  h.val.resize(static_cast<size_t>(g.nnz));
  HF_HIP(hipMemcpyAsync(h.ptr.data(), g.ptr, sizeof(int32_t) * h.ptr.size(), hipMemcpyDeviceToHost, ctx->stream));
  if (g.nnz) {
    HF_HIP(hipMemcpyAsync(h.idx.data(), g.idx, sizeof(int32_t) * h.idx.size(), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipMemcpyAsync(h.val.data(), g.val, sizeof(double) * h.val.size(), hipMemcpyDeviceToHost, ctx->stream));
  }
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}


// The hierarchy with its big levels formed on the device.  Level by level, as amg::build does: diagonal, Gershgorin bound and
// the greedy aggregation on the host (from a host copy of the level's operator), everything else on the device; the first
// level that does not fit the device kernels (a row longer than GA_MAXROW, a product row with more than GA_CAP candidate
// columns), the coarsest level and a stalled coarsening go to amg::build, which continues from that level's operator.
// HEATFLOW_AMG_SETUP=host takes build_amg (the host-only set-up) instead.
int build_amg_device(hf_ctx* ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  free_amg(ctx);
  amg::Params prm;
  amg_params(ctx, prm);
  const bool f32 = ctx->amg_f32;
  auto lap = [&](const char* what) {
    if (prm.verbose) std::fprintf(stderr, "[amg setup] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  };
  GpuScratch S;
  amg::Csr Ah;                                     // host copy of the current level's operator (aggregation, diagonal)
  Ah.nrow = Ah.ncol = ctx->n;
  Ah.ptr.assign(ctx->h_rowptr.begin(), ctx->h_rowptr.end());
  Ah.idx.assign(ctx->h_colidx.begin(), ctx->h_colidx.end());
  Ah.val.resize(ctx->nnz);
  HF_HIP(copy_sync(ctx, Ah.val.data(), ctx->d_A, sizeof(double) * ctx->nnz, hipMemcpyDeviceToHost));
  lap("operator downloaded");
  GCsr Ag;                                         // the same operator on the device
  Ag.nrow = Ag.ncol = ctx->n; Ag.nnz = ctx->nnz; Ag.ptr = ctx->d_rowptr; Ag.idx = ctx->d_colidx; Ag.val = ctx->d_A; Ag.owned = false;
  const double nnz0 = static_cast<double>(ctx->nnz);
  double nnz_sum = nnz0;
  struct Cleanup { GCsr* g; ~Cleanup() { gfree(*g); } } cleanup_ag{&Ag};
  int lev = 0;
  for (;; ++lev) {
    int max_row = 0;
    for (int i = 0; i < Ah.nrow; ++i) max_row = std::max(max_row, Ah.ptr[i + 1] - Ah.ptr[i]);
    const bool last = Ah.nrow <= prm.coarse_size || lev + 1 >= prm.max_levels;
    if (last || max_row > GA_MAXROW) break;
    std::vector<double> d = amg::diagonal(Ah);
    for (double v : d)
      if (!(v > 0.0)) return fail(ctx, HF_ERR_STATE, "AMG set-up failed (non-positive diagonal or singular coarse operator)");
    std::vector<double> dinv(d.size());
    for (size_t i = 0; i < d.size(); ++i) dinv[i] = 1.0 / d[i];
    const double rho = amg::gershgorin_rho(Ah, d);
    const double omega = prm.smooth_scale * 4.0 / (3.0 * rho);
    std::vector<int> agg;
    const double theta_l = amg::level_theta(prm, lev);
    const int na = amg::aggregate(Ah, d, theta_l, agg, prm.attach_weak);
    if (prm.verbose) {
      int64_t none = 0;
      for (int i = 0; i < Ah.nrow; ++i) none += agg[i] < 0;
      std::fprintf(stderr, "[amg setup] level %d rows %d aggregates %d (ratio %.2f) without aggregate %lld theta %.4f (device products)\n", lev, Ah.nrow, na,
                   static_cast<double>(Ah.nrow) / std::max(na, 1), static_cast<long long>(none), theta_l);
    }
    if (na == 0 || na > 0.8 * Ah.nrow) break;      // coarsening stalled: amg::build ends the hierarchy here the same way
    lap("  aggregated");
    const int n = Ah.nrow;
    DevTemp<int32_t> d_agg;
    DevTemp<double> d_diag, d_dinv_tmp;
    double* d_dinv = nullptr;
    HF_TRY(dev_alloc(ctx, &d_agg.p, n));
    HF_TRY(dev_alloc(ctx, &d_diag.p, n));
    HF_HIP(copy_sync(ctx, d_agg.p, agg.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, d_diag.p, d.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    if (lev == 0) {
      d_dinv = ctx->d_dinv;                        // k_dinv formed 1 / A_ii from the same values
    } else {
      HF_TRY(dev_alloc(ctx, &d_dinv_tmp.p, n));
      HF_HIP(copy_sync(ctx, d_dinv_tmp.p, dinv.data(), sizeof(double) * n, hipMemcpyHostToDevice));
      d_dinv = d_dinv_tmp.p;
    }
    const unsigned gr = static_cast<unsigned>((n + 255) / 256);
    // P
    GCsr P, R, AP, Ac, Rt, GP;
    struct Guard { GCsr *a, *b, *c, *d, *e, *f; ~Guard() { gfree(*a); gfree(*b); gfree(*c); gfree(*d); gfree(*e); gfree(*f); } } guard{&P, &R, &AP, &Ac, &Rt, &GP};
    P.nrow = n; P.ncol = na;
    HF_TRY(dev_alloc(ctx, &P.ptr, static_cast<size_t>(n) + 1));
    HF_HIP(hipMemsetAsync(P.ptr, 0, sizeof(int32_t) * (static_cast<size_t>(n) + 1), ctx->stream));
    const double wp = prm.prolong_scale * 4.0 / (3.0 * rho);
    hipLaunchKernelGGL(k_amg_prolong<false>, dim3(gr), dim3(256), 0, ctx->stream, n, Ag.ptr, Ag.idx, Ag.val, d_agg.p, d_diag.p, wp, P.ptr,
                       static_cast<int32_t*>(nullptr), static_cast<double*>(nullptr));
    HF_HIP(hipGetLastError());
    HF_TRY(gpu_scan(ctx, S, P.ptr, n, &P.nnz));
    HF_TRY(dev_alloc(ctx, &P.idx, static_cast<size_t>(P.nnz)));
    HF_TRY(dev_alloc(ctx, &P.val, static_cast<size_t>(P.nnz)));
    hipLaunchKernelGGL(k_amg_prolong<true>, dim3(gr), dim3(256), 0, ctx->stream, n, Ag.ptr, Ag.idx, Ag.val, d_agg.p, d_diag.p, wp, P.ptr, P.idx, P.val);
    HF_HIP(hipGetLastError());
    HF_TRY(gpu_transpose(ctx, S, P, R));
    lap("  P, R");
    bool fb = false;
    HF_TRY(gpu_spgemm(ctx, S, Ag, P, AP, &fb));
    if (fb) break;
    lap("  A P");
    HF_TRY(gpu_spgemm(ctx, S, R, AP, Ac, &fb));
    if (fb) break;
    lap("  R (A P)");
    const bool legs = lev > 0 || prm.fuse_fine;
    const bool up_leg = lev > 0 || !prm.fuse_fine_down_only;
    if (legs) {
      GCsr Pt = gclone_shallow(AP);                // pattern of A P, values of its own
      Pt.val = nullptr;
      DevTemp<double> ptval;
      DevTemp<int32_t> bad;
      HF_TRY(dev_alloc(ctx, &ptval.p, static_cast<size_t>(AP.nnz)));
      HF_TRY(dev_alloc(ctx, &bad.p, 1));
      HF_HIP(hipMemsetAsync(bad.p, 0, sizeof(int32_t), ctx->stream));
      hipLaunchKernelGGL(k_amg_smooth_p, dim3(gr), dim3(256), 0, ctx->stream, n, P.ptr, P.idx, P.val, AP.ptr, AP.idx, AP.val, d_dinv, omega, ptval.p, bad.p);
      HF_HIP(hipGetLastError());
      int32_t hb = 0;
      HF_HIP(copy_sync(ctx, &hb, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost));
      if (hb) break;                               // an entry of P outside the pattern of A P: the host merges general lists
      Pt.val = ptval.p;
      HF_TRY(gpu_transpose(ctx, S, Pt, Rt));
      if (up_leg) {
        GP.nrow = n; GP.ncol = n + na;
        HF_TRY(dev_alloc(ctx, &GP.ptr, static_cast<size_t>(n) + 1));
        HF_HIP(hipMemsetAsync(GP.ptr, 0, sizeof(int32_t) * (static_cast<size_t>(n) + 1), ctx->stream));
        hipLaunchKernelGGL(k_amg_up_leg<false>, dim3(gr), dim3(256), 0, ctx->stream, n, n, Ag.ptr, Ag.idx, Ag.val, Pt.ptr, Pt.idx, Pt.val, d_dinv, omega,
                           GP.ptr, static_cast<int32_t*>(nullptr), static_cast<double*>(nullptr));
        HF_HIP(hipGetLastError());
        HF_TRY(gpu_scan(ctx, S, GP.ptr, n, &GP.nnz));
        HF_TRY(dev_alloc(ctx, &GP.idx, static_cast<size_t>(GP.nnz)));
        HF_TRY(dev_alloc(ctx, &GP.val, static_cast<size_t>(GP.nnz)));
        hipLaunchKernelGGL(k_amg_up_leg<true>, dim3(gr), dim3(256), 0, ctx->stream, n, n, Ag.ptr, Ag.idx, Ag.val, Pt.ptr, Pt.idx, Pt.val, d_dinv, omega,
                           GP.ptr, GP.idx, GP.val);
        HF_HIP(hipGetLastError());
        HF_HIP(hipStreamSynchronize(ctx->stream));
      }
      lap("  fused legs");
    }
    // the level joins the hierarchy
    ctx->amg.resize(static_cast<size_t>(lev) + 1);
    DevLevel& L = ctx->amg[lev];
    L.n = n;
    L.omega = omega;
    if (lev > 0) {
      HF_TRY(adopt_csr(ctx, S, Ag, L.A, false));   // the level's own operator stays in double
      L.dinv = d_dinv_tmp.p;
      d_dinv_tmp.p = nullptr;
    }
    GCsr next = Ac;                                // keep the coarse operator: it is the next level's Ag (and its host copy Ah)
    Ac = GCsr();
    HF_TRY(gdownload(ctx, next, Ah));
    HF_TRY(adopt_csr(ctx, S, P, L.P, f32));
    HF_TRY(adopt_csr(ctx, S, R, L.R, f32, !(lev == 0 && legs) || ctx->amg_reuse != 0));
    if (legs) HF_TRY(adopt_csr(ctx, S, Rt, L.Rt, f32));
    if (legs && up_leg) HF_TRY(adopt_csr(ctx, S, GP, L.GP, f32));
    gfree(Ag);
    Ag = next;
    nnz_sum += static_cast<double>(Ag.nnz);
    lap("  level on the device");
  }
  // the remaining levels (at least the coarsest) on the host, continuing from the operator of level `lev`
  amg::Hierarchy H;
  prm.level_offset = lev;
  if (!amg::build(std::move(Ah), prm, H)) return fail(ctx, HF_ERR_STATE, "AMG set-up failed (non-positive diagonal or singular coarse operator)");
  lap("+ coarse levels on the host");
  const size_t base = static_cast<size_t>(lev);
  ctx->amg.resize(base + H.levels.size());
  HF_TRY(upload_host_levels(ctx, H, base));
  for (size_t k = 1; k < H.levels.size(); ++k) nnz_sum += static_cast<double>(H.levels[k].A.nnz());     // operator complexity: sum of nnz(A_l) / nnz(A_0)
  const double opc = nnz_sum / nnz0;
  const int coarse_n = ctx->amg.size() == 1 ? 0 : H.levels.back().A.nrow;
  return finish_amg(ctx, H.levels.back().A, coarse_n, opc, t0);
}

int build_amg_auto(hf_ctx* ctx) {
  static const bool host_only = std::getenv("HEATFLOW_AMG_SETUP") && std::string(std::getenv("HEATFLOW_AMG_SETUP")) == "host";
  static const bool study = std::getenv("HEATFLOW_AMG_PROLONG_STEPS") != nullptr;      // more than one prolongator smoothing step exists on the host only
  return (host_only || study) ? build_amg(ctx) : build_amg_device(ctx);
}

}  // namespace
