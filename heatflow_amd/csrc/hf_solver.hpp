// Part of libheatflow_hip.so (see heatflow_hip.hip): host code - multigrid hierarchy upload, V-cycle, PCG drivers, time step, Dirichlet lifting
#pragma once
#include "hf_pattern.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// multigrid: hierarchy upload, V-cycle, AMG-PCG step
// ------------------------------------------------------------------------------------------
using DevCsr = hf_ctx::DevCsr;
using DevLevel = hf_ctx::DevLevel;

void free_dev_csr(DevCsr& m) { dev_free(&m.ptr); dev_free(&m.idx); dev_free(&m.val); dev_free(&m.valf); dev_free(&m.dptr); dev_free(&m.dict); dev_free(&m.cid); m = DevCsr(); }

void free_amg(hf_ctx* ctx) {
  for (size_t l = 0; l < ctx->amg.size(); ++l) {
    DevLevel& L = ctx->amg[l];
    if (l > 0) { free_dev_csr(L.A); dev_free(&L.dinv); dev_free(&L.x); dev_free(&L.cat); }
    if (L.own_b) dev_free(&L.b);
    free_dev_csr(L.P); free_dev_csr(L.R); free_dev_csr(L.Rt); free_dev_csr(L.GP);
  }
  ctx->amg.clear();
  dev_free(&ctx->d_coarse_inv);
  dev_free(&ctx->d_coarse_inv_f);
  ctx->coarse_n = 0;
  ctx->amg_ready = false;
}

// lanes of a wavefront that share a row in the sub-wave kernels: about one entry per lane (HEATFLOW_VEC_PER_LANE: A/B)
int lanes_for_avg(double avg) {
  static const double scale = std::getenv("HEATFLOW_VEC_PER_LANE") ? std::atof(std::getenv("HEATFLOW_VEC_PER_LANE")) / 1.125 : 1.0;
  const double a = avg / scale;
  return a <= 4.5 ? 4 : a <= 9.0 ? 8 : a <= 18.0 ? 16 : a <= 36.0 ? 32 : (a <= 128.0 || scale != 1.0) ? 64 : 256;
}
int lanes_for(const amg::Csr& m) { return lanes_for_avg(m.nrow ? static_cast<double>(m.nnz()) / m.nrow : 1.0); }

// f32: values stored in single precision (operators that only act inside the preconditioner).  Operators big enough
// for the LDS-staged kernel also get its compressed column stream (a sorted column list per chunk + a 16-bit position
// per nonzero): 6 instead of 12 bytes per nonzero together.
int upload_csr(hf_ctx* ctx, const amg::Csr& h, DevCsr& d, bool f32 = false, bool stream = true /* false: never run by the LDS-staged kernel, skip its tables */) {
  d.nrow = h.nrow; d.ncol = h.ncol; d.nnz = h.nnz(); d.lanes = lanes_for(h);
  for (int i = 0; i < h.nrow; ++i) d.max_row = std::max(d.max_row, h.ptr[i + 1] - h.ptr[i]);
  d.rpc = 0;
  static const int min_rows = std::getenv("HEATFLOW_STREAM_MIN_ROWS") ? std::atoi(std::getenv("HEATFLOW_STREAM_MIN_ROWS")) : 20000;
  static const int max_nnz = std::getenv("HEATFLOW_STREAM_NNZ") ? std::atoi(std::getenv("HEATFLOW_STREAM_NNZ")) : 4096;
  if (stream && h.nrow >= min_rows) {  // enough 512-row chunks to fill the chip: LDS-staged kernel, chunk products within 64 KB
    for (int rpc = TS; rpc >= 32; rpc /= 2) {
      int mx = 0;
      for (int r0 = 0; r0 < h.nrow; r0 += rpc) mx = std::max(mx, h.ptr[std::min(h.nrow, r0 + rpc)] - h.ptr[r0]);
      if (mx <= max_nnz) { d.rpc = rpc; d.nchunks = (h.nrow + rpc - 1) / rpc; d.chunk_nnz = mx; break; }
    }
  }
  HF_TRY(dev_alloc(ctx, &d.ptr, h.ptr.size()));
  HF_TRY(dev_alloc(ctx, &d.idx, h.idx.size()));
  HF_HIP(copy_sync(ctx, d.ptr, h.ptr.data(), sizeof(int32_t) * h.ptr.size(), hipMemcpyHostToDevice));
  if (!h.idx.empty()) HF_HIP(copy_sync(ctx, d.idx, h.idx.data(), sizeof(int32_t) * h.idx.size(), hipMemcpyHostToDevice));
  if (f32) {
    std::vector<float> vf(h.val.begin(), h.val.end());
    HF_TRY(dev_alloc(ctx, &d.valf, vf.size()));
    if (!vf.empty()) HF_HIP(copy_sync(ctx, d.valf, vf.data(), sizeof(float) * vf.size(), hipMemcpyHostToDevice));
  } else {
    HF_TRY(dev_alloc(ctx, &d.val, h.val.size()));
    if (!h.val.empty()) HF_HIP(copy_sync(ctx, d.val, h.val.data(), sizeof(double) * h.val.size(), hipMemcpyHostToDevice));
  }
  if (f32 && d.rpc > 0) {   // compressed column stream for the LDS-staged kernel
    ColDict cd;
    if (build_coldict(h.ptr, h.idx, h.nrow, d.rpc, cd, h.ncol) && static_cast<size_t>(d.chunk_nnz + cd.max_dict) * 8 <= 64 * 1024) {
      d.max_dict = cd.max_dict;
      d.ndict = static_cast<int64_t>(cd.dict.size());
      HF_TRY(dev_alloc(ctx, &d.dptr, cd.ptr.size()));
      HF_TRY(dev_alloc(ctx, &d.dict, cd.dict.size()));
      HF_TRY(dev_alloc(ctx, &d.cid, cd.id.size()));
      HF_HIP(copy_sync(ctx, d.dptr, cd.ptr.data(), sizeof(int32_t) * cd.ptr.size(), hipMemcpyHostToDevice));
      HF_HIP(copy_sync(ctx, d.dict, cd.dict.data(), sizeof(int32_t) * cd.dict.size(), hipMemcpyHostToDevice));
      HF_HIP(copy_sync(ctx, d.cid, cd.id.data(), sizeof(uint16_t) * cd.id.size(), hipMemcpyHostToDevice));
    }
  }
  return HF_OK;
}

// Level vectors and aliases once every level's operators are on the device (build_amg, hf_amg_install): level 0 is the
// context's own operator; an intermediate level's right-hand side b is the head of `cat` = [b_l ; result of level l + 1],
// the operand of its fused up leg; a level's result goes to `res` (the tail of the finer level's cat, or - level 1 - its
// own x, or behind d_r when the finest level's up leg is fused); the coarsest level owns a zero-padded b.
int wire_levels(hf_ctx* ctx) {
  const size_t nl = ctx->amg.size();
  for (size_t l = 0; l < nl; ++l) {
    DevLevel& L = ctx->amg[l];
    if (l == 0) {
      L.A.nrow = L.A.ncol = ctx->n; L.A.nnz = ctx->nnz; L.A.ptr = ctx->d_rowptr; L.A.idx = ctx->d_colidx; L.A.val = ctx->d_A;
      L.dinv = ctx->d_dinv;
      continue;
    }
    if (l + 1 < nl) {
      const size_t len = static_cast<size_t>(L.n) + ctx->amg[l + 1].n + 2;
      HF_TRY(dev_alloc(ctx, &L.cat, len));
      HF_HIP(hipMemsetAsync(L.cat, 0, sizeof(double) * len, ctx->stream));
      L.b = L.cat;
    } else {
      HF_TRY(dev_alloc(ctx, &L.b, L.n + 4));
      L.own_b = true;
      HF_HIP(hipMemsetAsync(L.b, 0, sizeof(double) * (L.n + 4), ctx->stream));
    }
    if (l == 1) {
      HF_TRY(dev_alloc(ctx, &L.x, L.n + 2));
      L.res = ctx->amg[0].GP.nrow > 0 ? ctx->d_r + ctx->n : L.x;
    } else {
      L.res = ctx->amg[l - 1].cat + ctx->amg[l - 1].n;
    }
  }
  return HF_OK;
}

// Set-up parameters: defaults of amg_host.hpp, the study knobs of the environment, and the form of the finest level.
void amg_params(hf_ctx* ctx, amg::Params& prm) {
  if (const char* e = std::getenv("HEATFLOW_AMG_F32")) ctx->amg_f32 = (e[0] != '0');
  if (const char* e = std::getenv("HEATFLOW_AMG_THETA")) prm.theta = std::atof(e);          // tuning knobs
  if (const char* e = std::getenv("HEATFLOW_AMG_COARSE")) prm.coarse_size = std::atoi(e);
  if (const char* e = std::getenv("HEATFLOW_AMG_THETA_COARSE")) prm.theta_coarse = std::atof(e);
  if (const char* e = std::getenv("HEATFLOW_AMG_THETA_DECAY")) prm.theta_decay = std::atof(e);
  if (const char* e = std::getenv("HEATFLOW_AMG_ATTACH_WEAK")) prm.attach_weak = (e[0] != '0');
  prm.verbose = std::getenv("HEATFLOW_DEBUG") != nullptr;
  if (const char* e = std::getenv("HEATFLOW_AMG_SMOOTH_SCALE")) prm.smooth_scale = std::atof(e);
  // Finest level of the V-cycle.  Its explicit form passes over A twice (residual after pre-smoothing, post-smoothing
  // sweep) besides R_0 and P_0.  The fused down leg Rt_0 = R_0 (I - w A D^-1) replaces the residual pass and R_0 by one
  // operator of ~45 entries per coarse row in single precision: faster at every size measured (1M DOF -6 %, 4M -10 %,
  // 16M -10 % per step against the explicit form).  The fused up leg GP_0 (~16 entries per fine row against A's 7) only
  // pays once A streams from HBM far beyond the Infinity Cache: 1M +7 %, 4M +12 %, 16M -5 % against the down leg alone.
  // Same iteration counts in all three forms (the same preconditioner in exact arithmetic).
  ctx->amg_fuse0 = static_cast<size_t>(ctx->nnz) * 12 > (static_cast<size_t>(1) << 30) ? 1 : 2;
  if (const char* e = std::getenv("HEATFLOW_AMG_FUSE0")) ctx->amg_fuse0 = std::atoi(e);
  prm.fuse_fine = ctx->amg_fuse0 != 0;
  prm.fuse_fine_down_only = ctx->amg_fuse0 == 2;
}

// Levels built on the host go to the device: H.levels[k] becomes level base + k of ctx->amg (already sized).
int upload_host_levels(hf_ctx* ctx, const amg::Hierarchy& H, size_t base) {
  const bool f32 = ctx->amg_f32;
  const size_t nl = ctx->amg.size();
  for (size_t k = 0; k < H.levels.size(); ++k) {
    const size_t l = base + k;
    DevLevel& L = ctx->amg[l];
    const amg::Level& hl = H.levels[k];
    L.n = static_cast<int>(hl.dinv.size());
    L.omega = hl.omega;
    if (l == 0) {
      if (nl > 1 && hl.Rt.nrow > 0) {            // fused finest level: GP's operand is [r; x_1] = d_r with the level-1 result behind it
        HF_TRY(upload_csr(ctx, hl.Rt, L.Rt, f32));
        if (hl.GP.nrow > 0) HF_TRY(upload_csr(ctx, hl.GP, L.GP, f32));
      }
    } else {
      HF_TRY(upload_csr(ctx, hl.A, L.A));
      HF_TRY(dev_alloc(ctx, &L.dinv, L.n));
      HF_HIP(copy_sync(ctx, L.dinv, hl.dinv.data(), sizeof(double) * L.n, hipMemcpyHostToDevice));
      if (l + 1 < nl) {                       // intermediate level: the cycle runs through its fused legs
        HF_TRY(upload_csr(ctx, hl.Rt, L.Rt, f32));
        HF_TRY(upload_csr(ctx, hl.GP, L.GP, f32));
      }
    }
    if (l + 1 < nl) {
      HF_TRY(upload_csr(ctx, hl.P, L.P, f32));
      // with a fused down leg the single-column cycle applies R_0 itself only after the fine operator has been re-valued
      // under a frozen hierarchy (vcycle); the batched loop always does, through kb_csr, which needs no stream tables
      HF_TRY(upload_csr(ctx, hl.R, L.R, f32, !(l == 0 && hl.Rt.nrow > 0) || ctx->amg_reuse != 0));
    }
  }
  return HF_OK;
}

// Last steps of a set-up once every level's operators are on the device: the finest level's fused legs are dropped when they
// are too small for the only kernel that carries their epilogues, the level vectors are wired, the dense inverse of the
// coarsest operator (host copy Ac) is formed on the device by Gauss-Jordan.
int finish_amg(hf_ctx* ctx, const amg::Csr& Ac, int coarse_n, double op_complexity, const std::chrono::steady_clock::time_point t0) {
  const bool f32 = ctx->amg_f32;
  const size_t nl = ctx->amg.size();
  auto lap = [&](const char* what) {
    if (std::getenv("HEATFLOW_DEBUG")) std::fprintf(stderr, "[amg setup] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  };
  {
    DevLevel& L = ctx->amg[0];
    if (L.Rt.nrow > 0 && (L.Rt.rpc == 0 || (L.GP.nrow > 0 && L.GP.rpc == 0))) {    // too small for the LDS-staged kernel (the only one with the convergence test / r.z epilogue): explicit sweeps
      free_dev_csr(L.Rt);
      free_dev_csr(L.GP);
    }
  }
  if (std::getenv("HEATFLOW_DEBUG")) {
    auto show = [](const char* nm, size_t l, const DevCsr& m) {
      if (m.nrow) std::fprintf(stderr, "[amg] level %zu %-2s %8d x %8d nnz %9lld (%.1f/row, max %d) %s rpc %d lanes %d %s%s\n", l, nm, m.nrow, m.ncol,
                               static_cast<long long>(m.nnz), static_cast<double>(m.nnz) / m.nrow, m.max_row, m.rpc ? "stream" : "vec", m.rpc, m.lanes,
                               m.valf ? "f32" : "f64", m.cid ? " c16" : "");
    };
    for (size_t l = 0; l < nl; ++l) {
      show("A", l, ctx->amg[l].A); show("P", l, ctx->amg[l].P); show("R", l, ctx->amg[l].R);
      show("Rt", l, ctx->amg[l].Rt); show("GP", l, ctx->amg[l].GP);
    }
  }
  HF_TRY(wire_levels(ctx));
  lap("+ operators on the device");
  // coarsest level: dense inverse by Gauss-Jordan on the device
  ctx->coarse_n = 0;
  if (nl > 1 && coarse_n > 0 && coarse_n <= 4096) {
    const int nc = coarse_n;
    const int ld = (nc + 3) & ~3;
    for (int i = 0; i < nc; ++i)
      for (int k = Ac.ptr[i]; k < Ac.ptr[i + 1]; ++k)
        if (Ac.idx[k] < 0 || Ac.idx[k] >= nc) return fail(ctx, HF_ERR_ARG, "coarsest operator: column %d outside [0,%d)", Ac.idx[k], nc);
    // W = [A | I] on the device (the operator travels in CSR form), blocked Gauss-Jordan (hf_kernels.hpp), A^-1 = the right half
    DevTemp<double> t_w, t_r, t_val;
    DevTemp<int32_t> t_ptr, t_idx;
    HF_TRY(dev_alloc(ctx, &t_w.p, static_cast<size_t>(nc) * 2 * nc));
    HF_TRY(dev_alloc(ctx, &t_r.p, static_cast<size_t>(GJ_B) * nc));
    HF_TRY(dev_alloc(ctx, &t_ptr.p, Ac.ptr.size()));
    HF_TRY(dev_alloc(ctx, &t_idx.p, std::max<size_t>(Ac.idx.size(), 1)));
    HF_TRY(dev_alloc(ctx, &t_val.p, std::max<size_t>(Ac.val.size(), 1)));
    HF_TRY(dev_alloc(ctx, &ctx->d_coarse_inv, static_cast<size_t>(nc) * ld));
    HF_HIP(copy_sync(ctx, t_ptr.p, Ac.ptr.data(), sizeof(int32_t) * Ac.ptr.size(), hipMemcpyHostToDevice));
    if (!Ac.idx.empty()) {
      HF_HIP(copy_sync(ctx, t_idx.p, Ac.idx.data(), sizeof(int32_t) * Ac.idx.size(), hipMemcpyHostToDevice));
      HF_HIP(copy_sync(ctx, t_val.p, Ac.val.data(), sizeof(double) * Ac.val.size(), hipMemcpyHostToDevice));
    }
    HF_HIP(hipMemsetAsync(t_w.p, 0, sizeof(double) * nc * 2 * nc, ctx->stream));
    hipLaunchKernelGGL(k_gjb_fill, dim3((nc + 255) / 256), dim3(256), 0, ctx->stream, nc, t_ptr.p, t_idx.p, t_val.p, t_w.p);
    const int tiles = (nc + 63) / 64;
    for (int c0 = 0; c0 < nc; c0 += GJ_B) {
      const int bsz = std::min(GJ_B, nc - c0);
      hipLaunchKernelGGL(k_gjb_rows, dim3(std::max(1, std::min((nc + TPB - 1) / TPB, 64))), dim3(TPB), 0, ctx->stream, nc, c0, bsz, t_w.p, t_r.p);
      hipLaunchKernelGGL(k_gjb_update, dim3(tiles, tiles), dim3(TPB), 0, ctx->stream, nc, c0, bsz, t_w.p, t_r.p);
    }
    HF_HIP(hipGetLastError());
    // rows re-pitched to the even leading dimension (zero pad column)
    HF_HIP(hipMemsetAsync(ctx->d_coarse_inv, 0, sizeof(double) * nc * ld, ctx->stream));
    HF_HIP(hipMemcpy2DAsync(ctx->d_coarse_inv, sizeof(double) * ld, t_w.p + nc, sizeof(double) * 2 * nc, sizeof(double) * nc, nc,
                            hipMemcpyDeviceToDevice, ctx->stream));
    if (f32) {
      HF_TRY(dev_alloc(ctx, &ctx->d_coarse_inv_f, static_cast<size_t>(nc) * ld));
      hipLaunchKernelGGL(k_to_float, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(nc) * ld, ctx->d_coarse_inv, ctx->d_coarse_inv_f);
      HF_HIP(hipGetLastError());
    }
    HF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->coarse_ld = ld;
    ctx->coarse_n = nc;
  }
  lap("+ dense inverse");
  ctx->amg_opc = op_complexity;
  ctx->amg_fine_stale = false;
  ctx->amg_ready = true;
  ctx->amg_setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return HF_OK;
}

// Build the hierarchy from the assembled, eliminated fine operator, everything on the host (download -> host set-up ->
// upload): the reference implementation of the set-up; hf_amg_gpu.hpp forms the same operators on the device.
int build_amg(hf_ctx* ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  free_amg(ctx);
  amg::Csr A0;
  A0.nrow = A0.ncol = ctx->n;
  A0.ptr.assign(ctx->h_rowptr.begin(), ctx->h_rowptr.end());
  A0.idx.assign(ctx->h_colidx.begin(), ctx->h_colidx.end());
  A0.val.resize(ctx->nnz);
  HF_HIP(copy_sync(ctx, A0.val.data(), ctx->d_A, sizeof(double) * ctx->nnz, hipMemcpyDeviceToHost));
  amg::Hierarchy H;
  amg::Params prm;
  amg_params(ctx, prm);
  auto lap = [&](const char* what) {
    if (prm.verbose) std::fprintf(stderr, "[amg setup] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  };
  lap("operator downloaded");
  if (!amg::build(std::move(A0), prm, H)) return fail(ctx, HF_ERR_STATE, "AMG set-up failed (non-positive diagonal or singular coarse operator)");
  lap("+ host hierarchy");
  ctx->amg.resize(H.levels.size());
  HF_TRY(upload_host_levels(ctx, H, 0));
  return finish_amg(ctx, H.levels.back().A, H.coarse_n, H.op_complexity, t0);
}

// VMODE 0: y = A x, 1: y += A x; LDS-staged kernel when the matrix is big enough to fill the chip,
// sub-wave kernel otherwise.
// LDS-staged kernel on an operator of the hierarchy in mode SM (0: y = A x, 7: y = A x with b.y partials), optional
// early exit on the convergence partials `conv_part`.  The operator must be one the stream kernel runs (rpc > 0).
// UN = stream entries per lane the chunk pipeline of k_spmv keeps in flight (single-precision operators): 4 where the
// average chunk holds at most 3.5 per lane (a longer chunk takes its remainder in a second, exposed pass), else 8
inline bool short_chunks(const DevCsr& m) { return m.nchunks > 0 && m.nnz <= static_cast<int64_t>(m.nchunks) * (7 * TS / 2); }

template <int SM, typename VT>
void launch_stream_t(hf_ctx* c, const DevCsr& m, const VT* val, const double* x, double* y, double* part0, const double* bvec,
                     double* conv_part) {
  const int npart = c->P;                  // the consumers of part0 sum c->P slots: never more workgroups than that
  // without partial sums to write (SM 0) the grid may exceed the partial slots: HEATFLOW_STREAM_GRID (A/B; default MAXP)
  static const int grid_cap = std::getenv("HEATFLOW_STREAM_GRID") ? std::atoi(std::getenv("HEATFLOW_STREAM_GRID")) : MAXP;
  int grid = std::min(m.nchunks, (SM == 0 && part0 == nullptr) ? std::max(grid_cap, 1) : npart);
  if (grid >= 64) grid &= ~7;
#define HF_STREAM_ARGS2 m.nrow, m.nchunks, m.rpc, m.ptr, m.idx, val, x, y, c->d_scal, part0, bvec, static_cast<const double*>(nullptr), \
                        static_cast<double*>(nullptr), static_cast<double*>(nullptr), conv_part, 0.0, npart, 0
  if (m.cid != nullptr && sizeof(VT) == 4 && short_chunks(m))
    hipLaunchKernelGGL((k_spmv<SM, true, VT, 4>), dim3(grid), dim3(TS), static_cast<size_t>(m.chunk_nnz + m.max_dict) * 8, c->stream,
                       HF_STREAM_ARGS2, ColComp{m.dptr, m.dict, m.cid, m.chunk_nnz, 0});
  else if (m.cid != nullptr)
    hipLaunchKernelGGL((k_spmv<SM, true, VT>), dim3(grid), dim3(TS), static_cast<size_t>(m.chunk_nnz + m.max_dict) * 8, c->stream,
                       HF_STREAM_ARGS2, ColComp{m.dptr, m.dict, m.cid, m.chunk_nnz, 0});
  else
    hipLaunchKernelGGL((k_spmv<SM, false, VT>), dim3(grid), dim3(TS), static_cast<size_t>(m.chunk_nnz) * 8, c->stream,
                       HF_STREAM_ARGS2, ColComp{nullptr, nullptr, nullptr, 0, 0});
#undef HF_STREAM_ARGS2
}

template <int SM>
void launch_stream(hf_ctx* c, const DevCsr& m, const double* x, double* y, double* part0, const double* bvec, double* conv_part) {
  if (m.valf != nullptr) launch_stream_t<SM, float>(c, m, m.valf, x, y, part0, bvec, conv_part);
  else launch_stream_t<SM, double>(c, m, m.val, x, y, part0, bvec, conv_part);
}

template <int VMODE, typename VT>
void launch_vec_t(hf_ctx* c, const DevCsr& m, const VT* val, const double* x, double* y) {
  if (m.rpc > 0) {
    constexpr int SM = VMODE == 0 ? 0 : 6;
    static const int grid_cap = std::getenv("HEATFLOW_STREAM_GRID") ? std::atoi(std::getenv("HEATFLOW_STREAM_GRID")) : MAXP;
    int grid = std::min(m.nchunks, std::max(grid_cap, 1));
    if (grid >= 64) grid &= ~7;
#define HF_STREAM_ARGS m.nrow, m.nchunks, m.rpc, m.ptr, m.idx, val, x, y, c->d_scal, static_cast<double*>(nullptr),                       \
                       static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), static_cast<double*>(nullptr),   \
                       static_cast<double*>(nullptr), static_cast<double*>(nullptr), 0.0, 0, 0
    if (m.cid != nullptr && sizeof(VT) == 4 && short_chunks(m))
      hipLaunchKernelGGL((k_spmv<SM, true, VT, 4>), dim3(grid), dim3(TS), static_cast<size_t>(m.chunk_nnz + m.max_dict) * 8, c->stream,
                         HF_STREAM_ARGS, ColComp{m.dptr, m.dict, m.cid, m.chunk_nnz, 0});
    else if (m.cid != nullptr)
      hipLaunchKernelGGL((k_spmv<SM, true, VT>), dim3(grid), dim3(TS), static_cast<size_t>(m.chunk_nnz + m.max_dict) * 8, c->stream,
                         HF_STREAM_ARGS, ColComp{m.dptr, m.dict, m.cid, m.chunk_nnz, 0});
    else
      hipLaunchKernelGGL((k_spmv<SM, false, VT>), dim3(grid), dim3(TS), static_cast<size_t>(m.chunk_nnz) * 8, c->stream,
                         HF_STREAM_ARGS, ColComp{nullptr, nullptr, nullptr, 0, 0});
#undef HF_STREAM_ARGS
    return;
  }
  if (m.lanes > 64) {  // very long rows: a workgroup per row
    hipLaunchKernelGGL((k_spmv_row<VMODE, VT>), dim3(std::max(1, std::min(m.nrow, 4096))), dim3(TPB), 0, c->stream, m.nrow, m.ptr,
                       m.idx, val, x, y, c->d_scal);
    return;
  }
  const int lanes = m.lanes;
  const long long threads = static_cast<long long>(m.nrow) * lanes;
  const int grid = static_cast<int>(std::max(1LL, std::min<long long>((threads + TPB - 1) / TPB, 2048)));
#define HF_VEC(L) hipLaunchKernelGGL((k_spmv_vec<L, VMODE, VT>), dim3(grid), dim3(TPB), 0, c->stream, m.nrow, m.ptr, m.idx, val, x, y, c->d_scal)
  switch (lanes) {
    case 4: HF_VEC(4); break;
    case 8: HF_VEC(8); break;
    case 16: HF_VEC(16); break;
    case 32: HF_VEC(32); break;
    default: HF_VEC(64); break;
  }
#undef HF_VEC
}

// VMODE 0: y = A x, 1: y += A x; LDS-staged kernel when the matrix is big enough to fill the chip,
// sub-wave kernel otherwise; double or single precision values.
template <int VMODE>
void launch_vec(hf_ctx* c, const DevCsr& m, const double* x, double* y) {
  if (m.valf != nullptr) launch_vec_t<VMODE, float>(c, m, m.valf, x, y);
  else launch_vec_t<VMODE, double>(c, m, m.val, x, y);
}

// z = B r: one V(1,1) cycle.  Fixed buffer roles (no pointer swaps): on entry d_z holds w0 D^-1 r (written by the update /
// start kernel; not needed when both legs of the finest level are fused); on exit d_z2 holds z and part_rz[out_slot] the
// partials of r.z.  The finest level comes in three forms (build_amg chooses by size, HEATFLOW_AMG_FUSE0): explicit
// residual + R_0 down and P_0 + sweep up, fused down leg Rt_0 with the explicit up leg, or both legs fused; every
// intermediate level is two launches, the fused down leg Rt and the fused up leg GP (amg_host.hpp), the coarsest level
// a dense mat-vec.  With `test_convergence` the first kernel of the cycle tests the iterate the cycle starts from.
// `part`: the whole cycle, only its first kernel (the one that tests), or everything after it - the polled loop holds the
// rest of a cycle back while the test it is waiting for may end the solve (pcg_solve).
enum CyclePart { CYCLE_ALL = 0, CYCLE_FIRST = 1, CYCLE_REST = 2 };
void vcycle(hf_ctx* c, int out_slot, bool test_convergence = false, CyclePart part = CYCLE_ALL) {
  const int nl = static_cast<int>(c->amg.size());
  DevLevel& L0 = c->amg[0];
  if (nl == 1) {  // no coarse level: one more Jacobi sweep keeps the operator symmetric
    if (part != CYCLE_REST)
      launch_spmv<4>(c, c->d_A, c->d_z, c->d_z2, c->d_part_rz + out_slot * MAXP, c->d_r, nullptr, nullptr, nullptr, L0.omega);
    return;
  }
  const bool fused0 = L0.GP.nrow > 0;     // both legs of the finest level fused
  // A fused down leg alone must match the operator the explicit up leg sweeps over: after a re-valuation under a frozen
  // hierarchy it does not (Rt_0 holds the old A), the cycle would no longer be symmetric and PCG breaks down - the
  // explicit down leg takes over.  With both legs fused the cycle is a fixed SPD operator of the old A: weaker, but sound.
  if (L0.Rt.nrow > 0 && (fused0 || !c->amg_fine_stale)) {
    // finest level through its fused legs: b_1 = Rt_0 r (pre-smoothing from zero, residual and restriction in one
    // operator; early exit if the update before it has converged)
    if (part != CYCLE_REST) launch_stream<0>(c, L0.Rt, c->d_r, c->amg[1].b, nullptr, nullptr, test_convergence ? c->d_part_zz : nullptr);
  } else {
    if (part != CYCLE_REST)
      launch_spmv<3>(c, c->d_A, c->d_z, c->d_tmp, nullptr, c->d_r, nullptr, nullptr,   // t = r - A z (+ early exit on convergence)
                     test_convergence ? c->d_part_zz : nullptr);
    if (part != CYCLE_FIRST) launch_vec<0>(c, L0.R, c->d_tmp, c->amg[1].b);      // b_1 = R_0 t
  }
  if (part == CYCLE_FIRST) return;
  for (int l = 1; l + 1 < nl; ++l) launch_vec<0>(c, c->amg[l].Rt, c->amg[l].b, c->amg[l + 1].b);   // b_{l+1} = Rt_l b_l
  {
    DevLevel& Lc = c->amg[nl - 1];
    if (c->coarse_n > 0) {
      const int g = std::max(1, std::min((Lc.n + 1) / 2, 2048));
      if (c->d_coarse_inv_f != nullptr)
        hipLaunchKernelGGL(k_dense_mv_f32, dim3(g), dim3(TPB), 0, c->stream, Lc.n, c->coarse_ld, c->d_coarse_inv_f, Lc.b, Lc.res, c->d_scal);
      else
        hipLaunchKernelGGL(k_dense_mv, dim3(g), dim3(TPB), 0, c->stream, Lc.n, c->coarse_ld, c->d_coarse_inv, Lc.b, Lc.res, c->d_scal);
    } else {
      const int g = std::max(1, std::min((Lc.n + TPB - 1) / TPB, 1024));
      hipLaunchKernelGGL(k_scale, dim3(g), dim3(TPB), 0, c->stream, Lc.n, Lc.omega, Lc.dinv, Lc.b, Lc.res, c->d_scal);
    }
  }
  for (int l = nl - 2; l >= 1; --l) launch_vec<0>(c, c->amg[l].GP, c->amg[l].cat, c->amg[l].res);  // x_l = GP_l [b_l; x_{l+1}]
  if (fused0) {
    // z = GP_0 [r; x_1] (prolongation and post-smoothing in one operator) with the r.z partials
    launch_stream<7>(c, L0.GP, c->d_r, c->d_z2, c->d_part_rz + out_slot * MAXP, c->d_r, nullptr);
    return;
  }
  launch_vec<1>(c, L0.P, c->amg[1].res, c->d_z);                                // z += P_0 x_1
  launch_spmv<4>(c, c->d_A, c->d_z, c->d_z2, c->d_part_rz + out_slot * MAXP, c->d_r, nullptr, nullptr, nullptr, L0.omega);
}

// One multigrid-PCG iteration: iteration head (as above), update (alpha, x, r, z0 = w D^-1 r), V-cycle (z, r.z)
void launch_amg_iteration(hf_ctx* c, double* x, int parity, CyclePart part = CYCLE_ALL) {
  if (part == CYCLE_REST) { vcycle(c, parity ^ 1, true, CYCLE_REST); return; }
  const bool timed = c->prof && c->prof_used < PROF_PAIRS;
  hipEvent_t e0 = timed ? c->prof_ev[2 * c->prof_used] : nullptr, e1 = timed ? c->prof_ev[2 * c->prof_used + 1] : nullptr;
  launch_spmv<9>(c, c->d_A, c->d_z2, c->d_Ap, c->d_part_pAp, nullptr, c->d_p, c->d_part_rz, c->d_part_zz, 0.0, nullptr, e0,
                 e1, parity);
  if (timed) c->prof_used++;
  hipLaunchKernelGGL(k_pcg_update_amg, dim3(c->P), dim3(TPB), 0, c->stream, c->n, c->nchunks, c->P, parity, c->d_scal,
                     c->d_part_pAp, c->d_part_rz, c->d_part_zz, x, c->d_r, c->d_p, c->d_Ap, c->d_dinv,
                     c->amg[0].omega, c->amg[0].GP.nrow > 0 ? static_cast<double*>(nullptr) : c->d_z);
  vcycle(c, parity ^ 1, true, part);
}

// HEATFLOW_POLL=0: bursts + copy-back of the scalars instead of the polled, one-test-ahead loops (A/B and diagnosis)
inline bool poll_enabled() {
  static const bool on = !(std::getenv("HEATFLOW_POLL") && std::getenv("HEATFLOW_POLL")[0] == '0');
  return on;
}

// device scalars = zero, with the address of the host mirror (complete on return)
int reset_scal(hf_ctx* ctx) {
  *ctx->h_scal = Scal{};
  ctx->h_scal->mirror = ctx->d_mirror;
  HF_HIP(hipMemcpyAsync(ctx->d_scal, ctx->h_scal, sizeof(Scal), hipMemcpyHostToDevice, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

void harvest_profile(hf_ctx* ctx);

int read_scal(hf_ctx* ctx) {
  HF_HIP(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, sizeof(Scal), hipMemcpyDeviceToHost, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  harvest_profile(ctx);
  return HF_OK;
}

// Wait - reading host memory only - until the iterate after update `k` has been tested (k = 0: the start kernel has run)
// or the solve has ended; h_scal receives what the mirror holds.  A device that makes no progress for
// HEATFLOW_POLL_TIMEOUT_S seconds (default 60) is an error, not a hang of the caller.
int wait_tested(hf_ctx* ctx, int k) {
  static const double limit_s = std::getenv("HEATFLOW_POLL_TIMEOUT_S") ? std::atof(std::getenv("HEATFLOW_POLL_TIMEOUT_S")) : 60.0;
  ScalMirror* m = ctx->h_mirror;
  const unsigned ep = ctx->epoch;
  const auto t0 = std::chrono::steady_clock::now();
  auto last_query = t0;
  for (unsigned spin = 1;; ++spin) {
    const int tested = mirror_tested(m, ep);
    const int done = mirror_done(m, ep);
    if (tested >= k || done != 0) break;
    cpu_relax();
    if ((spin & 0x3ff) == 0) {
      const auto now = std::chrono::steady_clock::now();
      if (now - last_query < std::chrono::milliseconds(2)) continue;   // hipStreamQuery takes the runtime lock other sessions' launch threads need
      last_query = now;
      if (hipStreamQuery(ctx->stream) == hipSuccess) {     // everything queued has run: the test we wait for was never launched, or a launch failed
        if (mirror_tested(m, ep) >= k || mirror_done(m, ep) != 0) break;
        return fail(ctx, HF_ERR_HIP, "PCG progress: stream drained before the test of iteration %d ran (%s)", k, hipGetErrorString(hipGetLastError()));
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
        return fail(ctx, HF_ERR_HIP, "PCG progress: no convergence test within %.0f s (waiting for iteration %d, at %d)", limit_s, k, tested);
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  ctx->h_scal->iters = m->iters;
  ctx->h_scal->zz = m->zz;
  ctx->h_scal->bn2 = m->bn2;
  ctx->h_scal->done = mirror_done(m, ep);
  return HF_OK;
}

void harvest_profile(hf_ctx* ctx) {
  if (ctx->prof) {  // harvest the event pairs of this burst (only launches that really ran count)
    for (int k = 0; k < ctx->prof_used; ++k) {
      float ms = 0.f;
      const bool ran = !ctx->h_scal->done || (ctx->prof_base + k) < ctx->h_scal->iters;
      if (ran && hipEventElapsedTime(&ms, ctx->prof_ev[2 * k], ctx->prof_ev[2 * k + 1]) == hipSuccess) {
        ctx->prof_spmv_ms += ms;
        ctx->prof_spmv_n += 1;
      }
    }
    ctx->prof_used = 0;
  }
}

// PCG on `sys` started from sys.x.  Jacobi: any system on the pattern; AMG: the main system only.
// Iteration count / residual are left in h_scal; *pred carries the burst-size hint between calls.
int pcg_solve(hf_ctx* ctx, const LinSys& sys, bool use_amg, double rtol, double atol, int max_it, int* pred) {
  static const bool trace_res = std::getenv("HEATFLOW_TRACE_RES") != nullptr;   // diagnostics: bursts of 2, residual printed after each
  // a new epoch: whatever an earlier solve's queued launches might still write to the mirror is not this solve's progress
  ctx->epoch += 1;
  if (!use_amg) {
    // r = b - A x, z = D^-1 r, r.z
    launch_spmv<2>(ctx, sys.A, sys.x, ctx->d_r, ctx->d_part_rz, sys.b, ctx->d_z, ctx->d_part_zz, ctx->d_part_bn, 0.0,
                   sys.dinv);
    hipLaunchKernelGGL(k_pcg_begin, dim3(1), dim3(TPB), 0, ctx->stream, ctx->P, rtol, atol, ctx->d_part_zz,
                       ctx->d_part_bn, ctx->d_scal, ctx->epoch);
  } else {
    // r = b - A x, z0 = w D^-1 r; tolerance; z = B r (V-cycle, r.z into slot 0)
    launch_spmv<5>(ctx, sys.A, sys.x, ctx->d_r, nullptr, sys.b, ctx->d_z, ctx->d_part_zz, ctx->d_part_bn,
                   ctx->amg[0].omega);
    hipLaunchKernelGGL(k_pcg_begin, dim3(1), dim3(TPB), 0, ctx->stream, ctx->P, rtol, atol, ctx->d_part_zz,
                       ctx->d_part_bn, ctx->d_scal, ctx->epoch);
    vcycle(ctx, 0);
  }
  HF_HIP(hipGetLastError());

  int launched = 0;
  if (use_amg && !trace_res && ctx->amg.size() > 1 && poll_enabled()) {   // (a one-level hierarchy has no V-cycle kernel that tests: bursts below)
    // Multigrid iterations are queued one test ahead: the first kernel of a V-cycle publishes the convergence test of
    // the iterate it starts from (ScalMirror), so the host learns the outcome while that cycle still has its other ten
    // kernels to run and has the next iteration queued before they finish - no copy, no synchronisation, no idle
    // device between bursts, and no iteration launched beyond the converged one except inside the first burst
    // (previous count - 2: the counts drift slowly).
    ctx->prof_base = 0;
    auto finish = [&](int rc) {
      if (ctx->prof) { (void)hipStreamSynchronize(ctx->stream); harvest_profile(ctx); }
      return rc;
    };
    if (*pred <= 0) {  // previous solve needed no iteration (e.g. constant field): look before launching
      HF_TRY(wait_tested(ctx, 0));
      if (ctx->h_scal->done == 1) return finish(HF_OK);
    }
    // After the blind burst the solve is expected to end within an iteration or two.  From there an iteration is queued up
    // to the kernel that tests (head, update, first kernel of the cycle) and the rest of its cycle only once the test
    // says "not yet": the device idles for the few microseconds the host needs to see the test and launch, instead of
    // running the seven remaining launches of a cycle nobody needs (4.6 us each; HEATFLOW_HOLD_BACK=0: whole iterations).
    static const bool hold_back = !(std::getenv("HEATFLOW_HOLD_BACK") && std::getenv("HEATFLOW_HOLD_BACK")[0] == '0');
    int burst = std::max(1, std::min(max_it, *pred - 2));
    bool rest_pending = false;      // the last launched iteration stops after its testing kernel
    while (true) {
      if (rest_pending) launch_amg_iteration(ctx, sys.x, (launched - 1) & 1, CYCLE_REST);
      const bool split = hold_back && burst == 1 && launched > 0;
      for (int k = 0; k < burst; ++k) launch_amg_iteration(ctx, sys.x, (launched + k) & 1, split ? CYCLE_FIRST : CYCLE_ALL);
      rest_pending = split;
      launched += burst;
      HF_HIP(hipGetLastError());
      const int rc = wait_tested(ctx, launched);
      if (rc != HF_OK) return finish(rc);
      if (ctx->h_scal->done || launched >= max_it) break;
      burst = 1;
    }
    *pred = ctx->h_scal->iters;
    if (ctx->h_scal->done == 2) return finish(fail(ctx, HF_ERR_NOCONV, "PCG breakdown (p.Ap <= 0) after %d iterations", ctx->h_scal->iters));
    if (!ctx->h_scal->done)
      return finish(fail(ctx, HF_ERR_NOCONV, "PCG not converged in %d iterations (rel. residual %.3e)", ctx->h_scal->iters,
                         std::sqrt(ctx->h_scal->zz / std::max(ctx->h_scal->bn2, 1e-300))));
    return finish(HF_OK);
  }
  if (*pred <= 0) {  // previous solve needed no iteration (e.g. constant field): look before launching
    HF_TRY(read_scal(ctx));
    if (ctx->h_scal->done == 1) return HF_OK;
  }
  // first burst: what the previous solve needed (the counts drift slowly), then check in small bursts
  int burst = std::max(1, std::min(max_it, *pred > 0 ? *pred : (use_amg ? 8 : 32)));
  if (trace_res) burst = 2;
  while (true) {
    ctx->prof_base = launched;
    for (int k = 0; k < burst; ++k) {
      if (use_amg) launch_amg_iteration(ctx, sys.x, (launched + k) & 1);
      else launch_pcg_iteration(ctx, sys, (launched + k) & 1);
    }
    launched += burst;
    HF_HIP(hipGetLastError());
    HF_TRY(read_scal(ctx));
    if (trace_res) std::fprintf(stderr, "[res] it %d rel %.3e\n", ctx->h_scal->iters, std::sqrt(ctx->h_scal->zz / std::max(ctx->h_scal->bn2, 1e-300)));
    if (ctx->h_scal->done) break;
    if (launched >= max_it) break;
    // a multigrid iteration that is not needed costs more (11 early-exit launches) than the look that avoids it
    burst = std::max(1, std::min(std::max(use_amg ? 1 : 8, launched / 8), max_it - launched));
    if (trace_res) burst = 2;
  }
  *pred = ctx->h_scal->iters;
  if (ctx->h_scal->done == 2) return fail(ctx, HF_ERR_NOCONV, "PCG breakdown (p.Ap <= 0) after %d iterations", ctx->h_scal->iters);
  if (!ctx->h_scal->done)
    return fail(ctx, HF_ERR_NOCONV, "PCG not converged in %d iterations (rel. residual %.3e)", ctx->h_scal->iters,
                std::sqrt(ctx->h_scal->zz / std::max(ctx->h_scal->bn2, 1e-300)));
  return HF_OK;
}

// forget the projection basis (new operator or new Dirichlet set); buffers are kept.  ring_only: a new state on the
// same operator - the boundary responses stay valid, the solutions of the old trajectory are dropped.
void proj_clear(hf_ctx* ctx, bool ring_only = false) {
  for (int k = 0; k < (ring_only ? PROJ_MH : PROJ_MT); ++k) ctx->proj.used[k] = false;
  ctx->proj.next = 0;
  ctx->proj.pending = -1;
}

void proj_free(hf_ctx* ctx) {
  proj_clear(ctx);
  for (auto& v : ctx->proj.V) dev_free(&v);
  for (auto& v : ctx->proj.F) dev_free(&v);
  dev_free(&ctx->proj.G); dev_free(&ctx->proj.alpha); dev_free(&ctx->proj.part);
  ctx->proj.ready = false;
}

void free_responses(hf_ctx* ctx) {
  for (auto& r : ctx->resp) dev_free(&r.w);
  ctx->resp.clear();
  ctx->g_hist = 0;
  proj_clear(ctx);
}

int proj_ensure(hf_ctx* ctx) {
  hf_ctx::Proj& Q = ctx->proj;
  if (Q.ready) return HF_OK;
  for (int k = 0; k < PROJ_MT; ++k) {
    HF_TRY(dev_alloc(ctx, &Q.V[k], ctx->n));
    HF_TRY(dev_alloc(ctx, &Q.F[k], ctx->n));
  }
  HF_TRY(dev_alloc(ctx, &Q.G, PROJ_MT * PROJ_MT));
  HF_TRY(dev_alloc(ctx, &Q.alpha, PROJ_MT + 1));
  HF_TRY(dev_alloc(ctx, &Q.part, static_cast<size_t>(2) * PROJ_MT * MAXP));
  HF_HIP(hipMemsetAsync(Q.G, 0, sizeof(double) * PROJ_MT * PROJ_MT, ctx->stream));
  Q.ready = true;
  proj_clear(ctx);
  return HF_OK;
}

ProjVecs proj_active(const hf_ctx* ctx) {
  ProjVecs a{};
  a.m = 0;
  for (int k = 0; k < PROJ_MT; ++k)
    if (ctx->proj.used[k]) { a.V[a.m] = ctx->proj.V[k]; a.slot[a.m] = k; ++a.m; }
  return a;
}

// store (u with zeroed Dirichlet entries, right-hand side) in `slot`; its Gram column is computed by the next proj_column
int proj_store(hf_ctx* ctx, int slot, const double* u, const double* rhs) {
  hf_ctx::Proj& Q = ctx->proj;
  // one kernel for both copies: a device-to-device hipMemcpyAsync costs several times a launch on the host side
  hipLaunchKernelGGL(k_copy2, dim3(ctx->P), dim3(TPB), 0, ctx->stream, ctx->n, u, Q.V[slot], rhs, Q.F[slot]);
  if (ctx->nbc > 0)
    hipLaunchKernelGGL(k_zero_entries, dim3((ctx->nbc + 255) / 256), dim3(256), 0, ctx->stream, ctx->nbc, ctx->d_bc_dofs, Q.V[slot]);
  Q.used[slot] = true;
  return HF_OK;
}

// Gram column of `slot` (jnew) and, with `f`, the right-hand side h and the coefficients alpha
void proj_column(hf_ctx* ctx, int slot, const double* f, bool solve) {
  hf_ctx::Proj& Q = ctx->proj;
  const ProjVecs a = proj_active(ctx);
  hipLaunchKernelGGL(k_proj_dots, dim3(ctx->P), dim3(TPB), 0, ctx->stream, ctx->n, a, f, slot >= 0 ? Q.F[slot] : static_cast<const double*>(nullptr),
                     Q.part);
  hipLaunchKernelGGL(k_proj_solve, dim3(1), dim3(PROJ_SOLVE_T), 0, ctx->stream, ctx->P, a, slot, solve ? 1 : 0, Q.part, Q.G, Q.alpha);
}

// w = R d for a new boundary direction d:  A_hat w = -lift(d) on the free rows, w_B = d  (one extra solve).
int solve_response(hf_ctx* ctx, const std::vector<double>& dir, int max_it, double** w_out) {
  double* w = nullptr;
  HF_TRY(dev_alloc(ctx, &w, ctx->n));
  const int nb = ctx->nbc;
  HF_HIP(hipMemsetAsync(w, 0, sizeof(double) * ctx->n, ctx->stream));
  HF_HIP(hipMemsetAsync(ctx->d_b, 0, sizeof(double) * ctx->n, ctx->stream));
  HF_HIP(hipMemcpyAsync(ctx->d_g, dir.data(), sizeof(double) * nb, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->nlift_rows > 0)
    hipLaunchKernelGGL(k_lift, dim3((ctx->nlift_rows + 255) / 256), dim3(256), 0, ctx->stream, ctx->nlift_rows,
                       ctx->d_lift_rows, ctx->d_lift_ptr, ctx->d_lift_bc, ctx->d_lift_val, ctx->d_g, ctx->d_b);
  hipLaunchKernelGGL(k_set_bc, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, ctx->d_g, ctx->d_b, w);
  const LinSys sys{ctx->d_A, ctx->d_dinv, w, ctx->d_b};
  const bool use_amg = ctx->precond == 1 && ctx->amg_ready;
  int pred = ctx->pred_iters;
  int rc = pcg_solve(ctx, sys, use_amg, 1e-8, 0.0, max_it, &pred);
  if (rc == HF_ERR_NOCONV && use_amg && ctx->h_scal->done == 2) rc = pcg_solve(ctx, sys, false, 1e-8, 0.0, max_it, &pred);
  if (rc != HF_OK) { dev_free(&w); return rc; }
  ctx->resp_solves += 1;
  *w_out = w;
  if (ctx->start_kind == 3 && ctx->proj.ready) {   // the response joins the projection basis: (w, its right-hand side)
    const int slot = PROJ_MH + static_cast<int>(ctx->resp.size());
    HF_TRY(proj_store(ctx, slot, w, ctx->d_b));
    proj_column(ctx, slot, nullptr, false);
  }
  return HF_OK;
}

// Coefficients of the boundary-response correction for the step to g_new (host vector): the second difference
// of the boundary values is expanded in the known directions; a remainder that is not rounding noise becomes a
// new direction (one extra solve, at most MAXRESP per operator).  Never fatal: on failure the correction is off.
int prepare_response(hf_ctx* ctx, const double* g_new, int max_it, RespArgs* ra) {
  const int nb = ctx->nbc;
  ra->k = 0;
  std::vector<double> rem(nb);
  double nrm2 = 0.0, gn2 = 0.0;
  for (int q = 0; q < nb; ++q) {
    rem[q] = (g_new[q] - ctx->h_g0[q]) - (ctx->h_g0[q] - ctx->h_g1[q]);
    nrm2 += rem[q] * rem[q];
    gn2 += g_new[q] * g_new[q];
  }
  if (!(nrm2 > 0.0)) return HF_OK;
  for (size_t k = 0; k < ctx->resp.size(); ++k) {       // modified Gram-Schmidt against the orthonormal set
    const std::vector<double>& d = ctx->resp[k].dir;
    double c = 0.0;
    for (int q = 0; q < nb; ++q) c += d[q] * rem[q];
    for (int q = 0; q < nb; ++q) rem[q] -= c * d[q];
    ra->c[k] = c;
    ra->w[k] = ctx->resp[k].w;
  }
  double rn2 = 0.0;
  for (int q = 0; q < nb; ++q) rn2 += rem[q] * rem[q];
  if (rn2 > 1e-12 * nrm2 && nrm2 > 1e-18 * gn2 && ctx->resp.size() < static_cast<size_t>(MAXRESP)) {
    const double rn = std::sqrt(rn2);
    hf_ctx::BcResponse nr;
    nr.dir.resize(nb);
    for (int q = 0; q < nb; ++q) nr.dir[q] = rem[q] / rn;
    const int rc = solve_response(ctx, nr.dir, max_it, &nr.w);
    if (rc == HF_ERR_HIP) return rc;
    if (rc == HF_OK) {
      ra->c[ctx->resp.size()] = rn;
      ra->w[ctx->resp.size()] = nr.w;
      ctx->resp.push_back(std::move(nr));
    } else {
      ctx->start_kind = 1;   // the extra solve did not converge: keep plain extrapolation from here on
    }
  }
  ra->k = static_cast<int>(ctx->resp.size());
  return HF_OK;
}

// One time step to the boundary values g_host (n_bc doubles on the host; g_dev = the same values already on
// the device, or null).  Leaves iteration count / residual in h_scal.
int step_device(hf_ctx* ctx, const double* g_host, const double* g_dev, double rtol, double atol, int max_it) {
  const int nb = ctx->nbc;
  RespArgs ra{};
  ra.k = 0;
  const bool projected = ctx->start_kind == 3;
  if (projected) HF_TRY(proj_ensure(ctx));
  const bool hist_ok = nb > 0 && ctx->extrapolate && ctx->have_prev && ctx->g_hist >= 2;
  if (ctx->start_kind >= 2 && hist_ok) HF_TRY(prepare_response(ctx, g_host, max_it, &ra));
  const double* g = g_dev ? g_dev : ctx->d_g;   // hf_run has every step's boundary values on the device already
  if (nb > 0 && !g_dev) HF_HIP(hipMemcpyAsync(ctx->d_g, g_host, sizeof(double) * nb, hipMemcpyHostToDevice, ctx->stream));
  if (projected) {
    // b = M u^n, lifting, set_bc; then the start vector = A-norm projection of the new solution on the span of the last
    // solutions and the boundary responses (kind 3)
    launch_spmv<0>(ctx, ctx->d_M, ctx->d_u, ctx->d_b);
    ctx->have_prev = true;
  } else if (ctx->extrapolate && ctx->have_prev) {
    // b = M u^n   (assemble_vector, run_with_diamond.py:476); with a previous step available the same
    // pass writes the extrapolated start vector 2 u^n - u^{n-1}, and the three state buffers rotate
    launch_spmv<8>(ctx, ctx->d_M, ctx->d_u, ctx->d_b, nullptr, ctx->d_uprev, ctx->d_ustart);
    // u^{n-1} <- u^n, iterate <- start vector
    HF_HIP(hipMemcpyAsync(ctx->d_uprev, ctx->d_u, sizeof(double) * ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
    if (ra.k > 0)
      hipLaunchKernelGGL(k_start_vector, dim3(ctx->P), dim3(TPB), 0, ctx->stream, ctx->n, ctx->d_ustart, ra, ctx->d_u);
    else
      HF_HIP(hipMemcpyAsync(ctx->d_u, ctx->d_ustart, sizeof(double) * ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    launch_spmv<0>(ctx, ctx->d_M, ctx->d_u, ctx->d_b);
    if (ctx->extrapolate) {         // keep u^n for the next step
      HF_HIP(hipMemcpyAsync(ctx->d_uprev, ctx->d_u, sizeof(double) * ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
      ctx->have_prev = true;
    }
  }
  const bool combine = projected && proj_active(ctx).m > 0;
  if (nb > 0) {
    if (ctx->nlift_rows > 0)  // apply_lifting (:477)
      hipLaunchKernelGGL(k_lift, dim3((ctx->nlift_rows + 255) / 256), dim3(256), 0, ctx->stream, ctx->nlift_rows,
                         ctx->d_lift_rows, ctx->d_lift_ptr, ctx->d_lift_bc, ctx->d_lift_val, g, ctx->d_b);
    // set_bc (:479); the same values seed the iterate.  With a projected start vector it runs once, after the combination
    // below: the basis vectors are zero on the Dirichlet rows, so the dot products do not see what b holds there
    if (!combine)
      hipLaunchKernelGGL(k_set_bc, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, g,
                         ctx->d_b, ctx->d_u);
  }
  if (combine) {
    hf_ctx::Proj& Q = ctx->proj;
    proj_column(ctx, Q.pending, ctx->d_b, true);     // Gram column of the pair stored after the last step + h + alpha
    Q.pending = -1;
    hipLaunchKernelGGL(k_proj_combine, dim3(ctx->P), dim3(TPB), 0, ctx->stream, ctx->n, proj_active(ctx), Q.alpha, ctx->d_u);
    if (nb > 0)
      hipLaunchKernelGGL(k_set_bc, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, nb, ctx->d_bc_dofs, g, ctx->d_b, ctx->d_u);
  }
  const LinSys sys{ctx->d_A, ctx->d_dinv, ctx->d_u, ctx->d_b};
  const bool use_amg = ctx->precond == 1 && ctx->amg_ready;
  int rc = pcg_solve(ctx, sys, use_amg, rtol, atol, max_it, &ctx->pred_iters);
  if (rc == HF_ERR_NOCONV && use_amg && ctx->h_scal->done == 2) {
    // breakdown inside the multigrid-preconditioned loop (p.Ap <= 0: the preconditioner was not SPD for
    // this operator): finish the step with the Jacobi preconditioner from the current iterate - still on
    // the GPU - and count the event
    ctx->amg_fallbacks += 1;
    int pred = 0;
    rc = pcg_solve(ctx, sys, false, rtol, atol, max_it, &pred);
  }
  if (rc == HF_OK && projected) {   // the new solution and its right-hand side join the projection basis
    hf_ctx::Proj& Q = ctx->proj;
    HF_TRY(proj_store(ctx, Q.next, ctx->d_u, ctx->d_b));
    Q.pending = Q.next;
    Q.next = (Q.next + 1) % PROJ_MH;
  }
  if (rc == HF_OK && nb > 0) {   // boundary history for the next step's second difference
    ctx->h_g1.swap(ctx->h_g0);
    ctx->h_g0.assign(g_host, g_host + nb);
    if (ctx->g_hist < 2) ctx->g_hist += 1;
  }
  return rc;
}

int ensure_samples(hf_ctx* ctx, int ns) {
  if (ns <= ctx->samp_cap) return HF_OK;
  HF_TRY(dev_alloc(ctx, &ctx->d_samp_idx, ns));
  HF_TRY(dev_alloc(ctx, &ctx->d_samp, ns));
  ctx->samp_cap = ns;
  return HF_OK;
}

int build_lift(hf_ctx* ctx) {
  // Host: for every free row i and BC column j with A_ij in the pattern -> (row i, bc index of j, slot)
  const int32_t n = ctx->n, nbc = ctx->nbc;
  std::vector<int32_t> dofs(nbc);
  HF_HIP(copy_sync(ctx, dofs.data(), ctx->d_bc_dofs, sizeof(int32_t) * nbc, hipMemcpyDeviceToHost));
  std::vector<int32_t> bc_index(n, -1);
  for (int32_t q = 0; q < nbc; ++q) bc_index[dofs[q]] = q;
  // free rows adjacent to a BC dof = columns of the BC rows (pattern is symmetric)
  std::vector<int32_t> rows;
  for (int32_t q = 0; q < nbc; ++q) {
    const int32_t j = dofs[q];
    for (int32_t k = ctx->h_rowptr[j]; k < ctx->h_rowptr[j + 1]; ++k) {
      const int32_t i = ctx->h_colidx[k];
      if (bc_index[i] < 0) rows.push_back(i);
    }
  }
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  std::vector<int32_t> ptr(rows.size() + 1, 0), lbc, lslot;
  for (size_t r = 0; r < rows.size(); ++r) {
    const int32_t i = rows[r];
    for (int32_t k = ctx->h_rowptr[i]; k < ctx->h_rowptr[i + 1]; ++k) {
      const int32_t q = bc_index[ctx->h_colidx[k]];
      if (q >= 0) { lbc.push_back(q); lslot.push_back(k); }
    }
    ptr[r + 1] = static_cast<int32_t>(lbc.size());
  }
  ctx->nlift_rows = static_cast<int32_t>(rows.size());
  ctx->nlift = static_cast<int32_t>(lbc.size());
  HF_TRY(dev_alloc(ctx, &ctx->d_lift_rows, rows.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_lift_ptr, ptr.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_lift_bc, lbc.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_lift_slot, lslot.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_lift_val, lbc.size()));
  if (!rows.empty()) HF_HIP(copy_sync(ctx, ctx->d_lift_rows, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_lift_ptr, ptr.data(), sizeof(int32_t) * ptr.size(), hipMemcpyHostToDevice));
  if (!lbc.empty()) {
    HF_HIP(copy_sync(ctx, ctx->d_lift_bc, lbc.data(), sizeof(int32_t) * lbc.size(), hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, ctx->d_lift_slot, lslot.data(), sizeof(int32_t) * lslot.size(), hipMemcpyHostToDevice));
  }
  return HF_OK;
}

}  // namespace
