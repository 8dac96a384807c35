// Part of libheatflow_hip.so (see heatflow_hip.hip): the multigrid hierarchy of a context as one blob (hf_amg_export /
// hf_amg_install).  A sweep builds the hierarchy ONCE - the first session of rank 0 - and hands it to every other solver
// session of every rank together with the mesh and the connectivity tables (reference: every pool worker of
// parameter_sweep.py:401-446 re-reads the mesh and factorises for itself, run_with_diamond.py:389-394).  The blob holds
// what build_amg uploads: per level the smoother's D^-1 and damping, the operators A_l, P_l, R_l and the fused legs with
// their kernel geometry and compressed column streams, and the dense inverse of the coarsest level; plus a fingerprint
// of the fine operator it was built from (time step, coefficient tables, Dirichlet set), so that the installing context
// knows whether its own operator is that one (fused finest-level legs usable) or another point of the sweep (frozen
// hierarchy: explicit legs on the finest level until the next rebuild).
#pragma once
#include "hf_amg_gpu.hpp"
#include "hf_batch.hpp"

namespace {

constexpr char AMG_MAGIC[8] = {'H', 'F', 'A', 'M', 'G', '0', '1', 0};

struct AmgBlobHeader {
  char magic[8];
  int64_t total_bytes, nnz;
  int32_t n, nl, fuse0, f32, coarse_n, coarse_ld, nbc, tab_len;
  double opc, dt;
  uint64_t bc_hash;
};

struct CsrRecord {      // one operator: scalars, then ptr / idx / values / (dptr, dict, cid)
  int32_t present, nrow, ncol, lanes, max_row, rpc, nchunks, chunk_nnz, max_dict, val_kind /* 0 none, 1 f64, 2 f32 */, has_c16, pad_;
  int64_t nnz, ndict;
};

uint64_t fnv1a(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

using OperatorPrint = hf_ctx::OperatorPrint;

int operator_print(hf_ctx* ctx, OperatorPrint& f) {
  f.dt = ctx->dt;
  f.kappa.resize(ctx->tab_len);
  f.rhoc.resize(ctx->tab_len);
  HF_HIP(copy_sync(ctx, f.kappa.data(), ctx->d_kappa, sizeof(double) * ctx->tab_len, hipMemcpyDeviceToHost));
  HF_HIP(copy_sync(ctx, f.rhoc.data(), ctx->d_rhoc, sizeof(double) * ctx->tab_len, hipMemcpyDeviceToHost));
  for (int t = 0; t < ctx->tab_len; ++t)
    if (!ctx->h_tag_used[t]) f.kappa[t] = f.rhoc[t] = 0.0;        // tags the mesh does not use carry NaN: not part of the operator
  f.nbc = ctx->nbc;
  std::vector<int32_t> dofs(ctx->nbc);
  if (ctx->nbc > 0) HF_HIP(copy_sync(ctx, dofs.data(), ctx->d_bc_dofs, sizeof(int32_t) * ctx->nbc, hipMemcpyDeviceToHost));
  f.bc_hash = fnv1a(dofs.data(), sizeof(int32_t) * dofs.size());
  return HF_OK;
}

bool same_print(const OperatorPrint& a, const OperatorPrint& b) {
  return a.dt == b.dt && a.nbc == b.nbc && a.bc_hash == b.bc_hash && a.kappa == b.kappa && a.rhoc == b.rhoc;
}

struct BlobOut {          // sizes first (dst == nullptr), then the same walk writes
  unsigned char* dst = nullptr;
  size_t at = 0;
  hf_ctx* ctx = nullptr;
  int rc = HF_OK;
  void host(const void* p, size_t bytes) {
    if (dst && bytes) std::memcpy(dst + at, p, bytes);
    at += pad16(bytes);
  }
  void dev(const void* p, size_t bytes) {
    if (dst && bytes && rc == HF_OK && hipMemcpyAsync(dst + at, p, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = fail(ctx, HF_ERR_HIP, "hf_amg_export: device-to-host copy failed");
    at += pad16(bytes);
  }
};

void put_csr(BlobOut& o, const DevCsr& m, bool alias_fine /* level 0's A: the context's own arrays, not part of the blob */) {
  CsrRecord r{};
  r.present = (m.nrow > 0 && !alias_fine) ? 1 : 0;
  if (r.present) {
    r.nrow = m.nrow; r.ncol = m.ncol; r.lanes = m.lanes; r.max_row = m.max_row; r.rpc = m.rpc; r.nchunks = m.nchunks; r.chunk_nnz = m.chunk_nnz;
    r.max_dict = m.max_dict; r.val_kind = m.valf ? 2 : (m.val ? 1 : 0); r.has_c16 = m.cid ? 1 : 0; r.nnz = m.nnz; r.ndict = m.cid ? m.ndict : 0;
  }
  o.host(&r, sizeof r);
  if (!r.present) return;
  o.dev(m.ptr, sizeof(int32_t) * (static_cast<size_t>(m.nrow) + 1));
  o.dev(m.idx, sizeof(int32_t) * static_cast<size_t>(m.nnz));
  if (r.val_kind == 2) o.dev(m.valf, sizeof(float) * static_cast<size_t>(m.nnz));
  else if (r.val_kind == 1) o.dev(m.val, sizeof(double) * static_cast<size_t>(m.nnz));
  if (r.has_c16) {
    o.dev(m.dptr, sizeof(int32_t) * (static_cast<size_t>(m.nchunks) + 1));
    o.dev(m.dict, sizeof(int32_t) * static_cast<size_t>(m.ndict));
    o.dev(m.cid, sizeof(uint16_t) * static_cast<size_t>(m.nnz));
  }
}

int walk_hierarchy(hf_ctx* ctx, BlobOut& o, const OperatorPrint& f) {
  AmgBlobHeader h{};
  std::memcpy(h.magic, AMG_MAGIC, 8);
  h.nnz = ctx->nnz; h.n = ctx->n; h.nl = static_cast<int32_t>(ctx->amg.size()); h.fuse0 = ctx->amg_fuse0; h.f32 = ctx->amg_f32 ? 1 : 0;
  h.coarse_n = ctx->coarse_n; h.coarse_ld = ctx->coarse_ld; h.nbc = f.nbc; h.tab_len = ctx->tab_len; h.opc = ctx->amg_opc; h.dt = f.dt;
  h.bc_hash = f.bc_hash;
  const size_t head_at = o.at;
  o.host(&h, sizeof h);
  o.host(f.kappa.data(), sizeof(double) * f.kappa.size());
  o.host(f.rhoc.data(), sizeof(double) * f.rhoc.size());
  for (size_t l = 0; l < ctx->amg.size(); ++l) {
    const DevLevel& L = ctx->amg[l];
    struct { int32_t n, pad_; double omega; } lv{L.n, 0, L.omega};
    o.host(&lv, sizeof lv);
    if (l > 0) o.dev(L.dinv, sizeof(double) * static_cast<size_t>(L.n));
    put_csr(o, L.A, l == 0);
    put_csr(o, L.P, false); put_csr(o, L.R, false); put_csr(o, L.Rt, false); put_csr(o, L.GP, false);
  }
  if (ctx->coarse_n > 0) o.dev(ctx->d_coarse_inv, sizeof(double) * static_cast<size_t>(ctx->coarse_n) * ctx->coarse_ld);
  if (o.dst) {                                         // the total is known now: patch the header
    h.total_bytes = static_cast<int64_t>(o.at);
    std::memcpy(o.dst + head_at, &h, sizeof h);
  }
  return o.rc;
}

struct BlobIn {
  const unsigned char* src;
  size_t bytes, at = 0;
  bool ok = true;
  const void* take(size_t need) {
    if (!ok || at + need > bytes) { ok = false; return nullptr; }
    const void* p = src + at;
    at += pad16(need);
    return p;
  }
};

// one operator out of the blob onto the device; every index a kernel follows without a bounds check is verified here
int get_csr(hf_ctx* ctx, BlobIn& in, DevCsr& d, int want_rows /* -1: any */, int want_cols) {
  const CsrRecord* rp = static_cast<const CsrRecord*>(in.take(sizeof(CsrRecord)));
  if (!rp) return fail(ctx, HF_ERR_ARG, "hf_amg_install: blob truncated");
  const CsrRecord r = *rp;
  d = DevCsr();
  if (!r.present) return HF_OK;
  if (r.nrow <= 0 || r.ncol <= 0 || r.nnz < 0 || r.nnz > INT32_MAX || (want_rows >= 0 && r.nrow != want_rows) || (want_cols >= 0 && r.ncol != want_cols) ||
      r.val_kind < 1 || r.val_kind > 2 || r.rpc < 0 || r.rpc > TS || (r.rpc > 0 && (r.nchunks != (r.nrow + r.rpc - 1) / r.rpc || (TS % r.rpc) != 0)) ||
      (r.has_c16 && (r.rpc == 0 || r.ndict <= 0)) || r.lanes < 1)
    return fail(ctx, HF_ERR_ARG, "hf_amg_install: operator record does not fit its level");
  const int32_t* ptr = static_cast<const int32_t*>(in.take(sizeof(int32_t) * (static_cast<size_t>(r.nrow) + 1)));
  const int32_t* idx = static_cast<const int32_t*>(in.take(sizeof(int32_t) * static_cast<size_t>(r.nnz)));
  const void* val = in.take((r.val_kind == 2 ? sizeof(float) : sizeof(double)) * static_cast<size_t>(r.nnz));
  const int32_t *dptr = nullptr, *dict = nullptr;
  const uint16_t* cid = nullptr;
  if (r.has_c16) {
    dptr = static_cast<const int32_t*>(in.take(sizeof(int32_t) * (static_cast<size_t>(r.nchunks) + 1)));
    dict = static_cast<const int32_t*>(in.take(sizeof(int32_t) * static_cast<size_t>(r.ndict)));
    cid = static_cast<const uint16_t*>(in.take(sizeof(uint16_t) * static_cast<size_t>(r.nnz)));
  }
  if (!in.ok) return fail(ctx, HF_ERR_ARG, "hf_amg_install: blob truncated");
  bool ok = ptr[0] == 0 && ptr[r.nrow] == r.nnz;
  int max_row = 0;
  for (int i = 0; i < r.nrow && ok; ++i) { ok = ptr[i + 1] >= ptr[i]; max_row = std::max(max_row, ptr[i + 1] - ptr[i]); }
  if (ok) {                                              // the O(nnz) checks on a few host threads (hf_pattern.hpp)
    std::vector<char> good(host_threads(), 1);
    parallel_ranges(r.nnz, 1 << 16, [&](int64_t k0, int64_t k1, int t) {
      bool g = true;
      for (int64_t k = k0; k < k1 && g; ++k) g = idx[k] >= 0 && idx[k] < r.ncol;
      good[t] = g ? 1 : 0;
    });
    for (char g : good) ok = ok && g;
  }
  int chunk_nnz = 0, max_dict = 0;
  if (ok && r.rpc > 0) {
    for (int c = 0; c < r.nchunks; ++c) chunk_nnz = std::max(chunk_nnz, ptr[std::min<int64_t>(r.nrow, (c + 1LL) * r.rpc)] - ptr[static_cast<size_t>(c) * r.rpc]);
    ok = chunk_nnz <= r.chunk_nnz && r.rpc >= 32;
  }
  if (ok && r.has_c16) {
    ok = dptr[0] == 0 && dptr[r.nchunks] == r.ndict;
    for (int c = 0; c < r.nchunks && ok; ++c) {
      const int nd = dptr[c + 1] - dptr[c];
      ok = nd > 0 && nd <= r.max_dict && dptr[c] >= 0 && dptr[c + 1] <= r.ndict;
      max_dict = std::max(max_dict, nd);
    }
    if (ok) {
      std::vector<char> good(host_threads(), 1);
      parallel_ranges(r.nchunks, 64, [&](int64_t c0, int64_t c1, int t) {
        bool g = true;
        for (int64_t c = c0; c < c1 && g; ++c) {
          const int nd = dptr[c + 1] - dptr[c];
          for (int32_t k = ptr[static_cast<size_t>(c) * r.rpc]; k < ptr[std::min<int64_t>(r.nrow, (c + 1) * r.rpc)] && g; ++k)
            g = cid[k] < nd && dict[dptr[c] + cid[k]] == idx[k];
        }
        good[t] = g ? 1 : 0;
      });
      for (char g : good) ok = ok && g;
    }
    for (int64_t k = 0; k < r.ndict && ok; ++k) ok = dict[k] >= 0 && dict[k] < r.ncol;
    ok = ok && static_cast<size_t>(r.chunk_nnz + r.max_dict) * 8 <= 64 * 1024;
  } else if (ok && r.rpc > 0) {
    ok = static_cast<size_t>(r.chunk_nnz) * 8 <= 64 * 1024;
  }
  if (!ok) return fail(ctx, HF_ERR_ARG, "hf_amg_install: an operator of the blob holds an index outside its range");
  d.nrow = r.nrow; d.ncol = r.ncol; d.nnz = r.nnz; d.lanes = r.lanes; d.max_row = max_row; d.rpc = r.rpc; d.nchunks = r.nchunks; d.chunk_nnz = r.chunk_nnz;
  d.max_dict = r.has_c16 ? r.max_dict : 0; d.ndict = r.has_c16 ? r.ndict : 0;
  HF_TRY(dev_alloc(ctx, &d.ptr, static_cast<size_t>(r.nrow) + 1));
  HF_TRY(dev_alloc(ctx, &d.idx, static_cast<size_t>(r.nnz)));
  HF_HIP(hipMemcpyAsync(d.ptr, ptr, sizeof(int32_t) * (static_cast<size_t>(r.nrow) + 1), hipMemcpyHostToDevice, ctx->stream));
  if (r.nnz) HF_HIP(hipMemcpyAsync(d.idx, idx, sizeof(int32_t) * static_cast<size_t>(r.nnz), hipMemcpyHostToDevice, ctx->stream));
  if (r.val_kind == 2) {
    HF_TRY(dev_alloc(ctx, &d.valf, static_cast<size_t>(r.nnz)));
    if (r.nnz) HF_HIP(hipMemcpyAsync(d.valf, val, sizeof(float) * static_cast<size_t>(r.nnz), hipMemcpyHostToDevice, ctx->stream));
  } else {
    HF_TRY(dev_alloc(ctx, &d.val, static_cast<size_t>(r.nnz)));
    if (r.nnz) HF_HIP(hipMemcpyAsync(d.val, val, sizeof(double) * static_cast<size_t>(r.nnz), hipMemcpyHostToDevice, ctx->stream));
  }
  if (r.has_c16) {
    HF_TRY(dev_alloc(ctx, &d.dptr, static_cast<size_t>(r.nchunks) + 1));
    HF_TRY(dev_alloc(ctx, &d.dict, static_cast<size_t>(r.ndict)));
    HF_TRY(dev_alloc(ctx, &d.cid, static_cast<size_t>(r.nnz)));
    HF_HIP(hipMemcpyAsync(d.dptr, dptr, sizeof(int32_t) * (static_cast<size_t>(r.nchunks) + 1), hipMemcpyHostToDevice, ctx->stream));
    HF_HIP(hipMemcpyAsync(d.dict, dict, sizeof(int32_t) * static_cast<size_t>(r.ndict), hipMemcpyHostToDevice, ctx->stream));
    HF_HIP(hipMemcpyAsync(d.cid, cid, sizeof(uint16_t) * static_cast<size_t>(r.nnz), hipMemcpyHostToDevice, ctx->stream));
  }
  return HF_OK;
}

int install_hierarchy(hf_ctx* ctx, const unsigned char* blob, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  BlobIn in{blob, bytes};
  const AmgBlobHeader* hp = static_cast<const AmgBlobHeader*>(in.take(sizeof(AmgBlobHeader)));
  if (!hp || std::memcmp(hp->magic, AMG_MAGIC, 8) != 0) return fail(ctx, HF_ERR_ARG, "hf_amg_install: not a hierarchy blob of this library version");
  const AmgBlobHeader h = *hp;
  if (h.total_bytes != static_cast<int64_t>(bytes)) return fail(ctx, HF_ERR_ARG, "hf_amg_install: blob says %lld bytes, %lld given", (long long)h.total_bytes, (long long)bytes);
  if (h.n != ctx->n || h.nnz != ctx->nnz) return fail(ctx, HF_ERR_ARG, "hf_amg_install: hierarchy of a %d-row operator with %lld nonzeros, this mesh has %d / %lld", h.n, (long long)h.nnz, ctx->n, (long long)ctx->nnz);
  if (h.nl < 1 || h.nl > 32 || h.tab_len != ctx->tab_len || h.coarse_n < 0 || h.coarse_n > 4096 || (h.coarse_n > 0 && (h.coarse_ld < h.coarse_n || (h.coarse_ld & 3))))
    return fail(ctx, HF_ERR_ARG, "hf_amg_install: header does not fit this context");
  OperatorPrint theirs;
  theirs.dt = h.dt; theirs.nbc = h.nbc; theirs.bc_hash = h.bc_hash;
  const double* pk = static_cast<const double*>(in.take(sizeof(double) * h.tab_len));
  const double* pc = static_cast<const double*>(in.take(sizeof(double) * h.tab_len));
  if (!in.ok) return fail(ctx, HF_ERR_ARG, "hf_amg_install: blob truncated");
  theirs.kappa.assign(pk, pk + h.tab_len);
  theirs.rhoc.assign(pc, pc + h.tab_len);
  free_amg(ctx);
  ctx->amg.resize(h.nl);
  int prev_n = ctx->n;
  auto bail = [&](int rc) { free_amg(ctx); return rc; };
  for (int l = 0; l < h.nl; ++l) {
    DevLevel& L = ctx->amg[l];
    struct Lv { int32_t n, pad_; double omega; };
    const Lv* lv = static_cast<const Lv*>(in.take(sizeof(Lv)));
    if (!lv || lv->n <= 0 || (l == 0 && lv->n != ctx->n) || !(lv->omega > 0.0)) return bail(fail(ctx, HF_ERR_ARG, "hf_amg_install: bad level record"));
    L.n = lv->n; L.omega = lv->omega;
    if (l > 0) {
      const double* dv = static_cast<const double*>(in.take(sizeof(double) * static_cast<size_t>(L.n)));
      if (!dv) return bail(fail(ctx, HF_ERR_ARG, "hf_amg_install: blob truncated"));
      int rc = dev_alloc(ctx, &L.dinv, L.n);
      if (rc != HF_OK) return bail(rc);
      if (hipMemcpyAsync(L.dinv, dv, sizeof(double) * L.n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(fail(ctx, HF_ERR_HIP, "hf_amg_install: copy failed"));
    }
    int rc = get_csr(ctx, in, L.A, l == 0 ? -1 : L.n, l == 0 ? -1 : L.n);
    if (rc == HF_OK && (l == 0) != (L.A.nrow == 0)) rc = fail(ctx, HF_ERR_ARG, "hf_amg_install: level operator missing or unexpected");
    if (rc == HF_OK) rc = get_csr(ctx, in, L.P, L.n, -1);
    if (rc == HF_OK) rc = get_csr(ctx, in, L.R, -1, L.n);
    if (rc == HF_OK) rc = get_csr(ctx, in, L.Rt, -1, L.n);
    if (rc == HF_OK) rc = get_csr(ctx, in, L.GP, L.n, -1);
    if (rc != HF_OK) return bail(rc);
    const bool last = l + 1 == h.nl;
    bool ok = last ? (L.P.nrow == 0 && L.R.nrow == 0 && L.Rt.nrow == 0 && L.GP.nrow == 0)
                   : (L.P.nrow == L.n && L.R.ncol == L.n && L.R.nrow == L.P.ncol && (L.Rt.nrow == 0 || L.Rt.nrow == L.P.ncol) &&
                      (L.GP.nrow == 0 || L.GP.ncol == L.n + L.P.ncol));
    if (ok && !last && l > 0) ok = L.Rt.nrow > 0 && L.GP.nrow > 0;       // intermediate levels run through their fused legs
    if (ok && l == 0 && (L.Rt.nrow > 0 || L.GP.nrow > 0)) ok = L.Rt.rpc > 0 && (L.GP.nrow == 0 || L.GP.rpc > 0) && L.Rt.nrow > 0;
    if (ok && l > 0) ok = L.n == prev_n;
    if (!ok) return bail(fail(ctx, HF_ERR_ARG, "hf_amg_install: the operators of level %d do not fit together", l));
    prev_n = last ? 0 : L.P.ncol;
  }
  if (h.nl > 1 && h.coarse_n != ctx->amg.back().n) return bail(fail(ctx, HF_ERR_ARG, "hf_amg_install: dense inverse of %d rows for a coarsest level of %d", h.coarse_n, ctx->amg.back().n));
  ctx->coarse_n = 0;
  if (h.coarse_n > 0) {
    const size_t cnt = static_cast<size_t>(h.coarse_n) * h.coarse_ld;
    const double* inv = static_cast<const double*>(in.take(sizeof(double) * cnt));
    if (!inv) return bail(fail(ctx, HF_ERR_ARG, "hf_amg_install: blob truncated"));
    int rc = dev_alloc(ctx, &ctx->d_coarse_inv, cnt);
    if (rc != HF_OK) return bail(rc);
    if (hipMemcpyAsync(ctx->d_coarse_inv, inv, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(fail(ctx, HF_ERR_HIP, "hf_amg_install: copy failed"));
    if (h.f32) {
      rc = dev_alloc(ctx, &ctx->d_coarse_inv_f, cnt);
      if (rc != HF_OK) return bail(rc);
      hipLaunchKernelGGL(k_to_float, dim3(1024), dim3(256), 0, ctx->stream, cnt, ctx->d_coarse_inv, ctx->d_coarse_inv_f);
    }
    ctx->coarse_n = h.coarse_n; ctx->coarse_ld = h.coarse_ld;
  }
  ctx->amg_fuse0 = h.fuse0; ctx->amg_f32 = h.f32 != 0; ctx->amg_opc = h.opc;
  int rc = wire_levels(ctx);
  if (rc != HF_OK) return bail(rc);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(fail(ctx, HF_ERR_HIP, "hf_amg_install: copies failed"));
  ctx->amg_print = std::move(theirs);
  ctx->amg_ready = true;
  ctx->amg_fine_stale = true;                    // until hf_assemble has compared this context's operator with the fingerprint
  ctx->amg_setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return HF_OK;
}

}  // namespace
