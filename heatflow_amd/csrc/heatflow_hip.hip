// libheatflow_hip.so - HIP/CDNA4 (gfx950) implementation of include/heatflow_hip.h: the C ABI.
//
// Hot path of cebarker1000/heatflow re-designed for MI355X:
//   assembly   per-element P1 kernel (r-weighted axisymmetric mass + stiffness), owner-computes
//              scatter-add into a CSR slab staged in LDS, streamed out once        (hf_kernels.hpp)
//   time step  b = M u^n (CSR SpMV) -> lifting -> set_bc -> PCG (CSR SpMV with LDS-staged products,
//              wavefront shuffles + fixed-order block partials, device-resident scalars), Jacobi or
//              smoothed-aggregation multigrid preconditioner         (hf_solver.hpp, amg_host.hpp)
// One translation unit: hf_context.hpp -> hf_kernels.hpp -> hf_pattern.hpp -> hf_solver.hpp -> this file.
// Reference semantics being reproduced: run_with_diamond.py:321-337 (forms), :381-394 (assemble
// once, symmetric Dirichlet elimination, solve), :469-481 (loop body).

#include "hf_amg_io.hpp"

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {

// Everything the device needs that derives from the mesh connectivity alone.  hf_set_mesh builds it on the host,
// hf_set_mesh_prebuilt takes it from a blob another context exported (hf_pattern_export): one rank builds, the
// others of a sweep receive it over RCCL together with the mesh (reference parameter_sweep.py:401-446 re-reads
// mesh.msh in every worker instead).
struct MeshTables {
  std::vector<int32_t> rowptr, colidx;
  ColDict spmv;                 // compressed columns per SRPC-row SpMV chunk
  RowGather rg;                 // row-gather assembly lists per RBA-row block (rg.ok = false: not available)
  std::vector<char> tag_used;   // cell tags present in the mesh (index = tag)
  int max_blk_nnz = 0;
};

constexpr char BLOB_MAGIC[8] = {'H', 'F', 'P', 'A', 'T', '0', '2', 0};
struct BlobHeader {
  char magic[8];
  int64_t total_bytes, nnz;
  int32_t n, ne, rba, ts, max_blk_nnz, tab_len, rg_ok, rg_max_dict, spmv_max_dict, reserved;
  int64_t count[12];            // elements per section, in the order written below
};

// Upload the tables and size every buffer of the context for the mesh.
int install_mesh(hf_ctx* ctx, int32_t n, int32_t ne, const double* zr, const int32_t* tri, const int32_t* tag, MeshTables& T) {
  free_batch(ctx);
  free_batch_state(ctx->fluxb);
  free_batch_cols(ctx);
  proj_free(ctx);
  ctx->n = n; ctx->ne = ne; ctx->nnz = static_cast<int64_t>(T.colidx.size());
  ctx->nchunks = (n + RB - 1) / RB;
  ctx->nblk_a = (n + RBA - 1) / RBA;
  ctx->P = std::min(ctx->nchunks, MAXP);
  if (ctx->P >= 64) ctx->P &= ~7;          // multiple of 8: one equal group of workgroups per XCD
  ctx->nchunks_s = (n + SRPC - 1) / SRPC;
  ctx->Ps = std::min(ctx->nchunks_s, MAXP);
  if (ctx->Ps >= 64) ctx->Ps &= ~7;
  ctx->max_chunk_nnz_s = 0;
  for (int c = 0; c < ctx->nchunks_s; ++c)
    ctx->max_chunk_nnz_s = std::max(ctx->max_chunk_nnz_s, T.rowptr[std::min<int64_t>(n, (c + 1LL) * SRPC)] - T.rowptr[static_cast<size_t>(c) * SRPC]);
  ctx->max_cdict = T.spmv.max_dict;
  ctx->c16 = true;
  if (const char* e = std::getenv("HEATFLOW_SPMV_C16")) ctx->c16 = (e[0] != '0');
  if (static_cast<size_t>(ctx->max_chunk_nnz_s + ctx->max_cdict) * 8 > 64 * 1024) ctx->c16 = false;   // LDS window of the kernel
  // own rows of a chunk contiguous in its sorted column list <=> every row stores its diagonal (P1 patterns do)
  ctx->cdict_own = static_cast<int>(T.spmv.ptr.size()) == ctx->nchunks_s + 1;
  for (int c = 0; c < ctx->nchunks_s && ctx->cdict_own; ++c) {
    const int32_t r0 = c * SRPC, r1 = std::min<int32_t>(n, r0 + SRPC);
    const int32_t* lo = T.spmv.dict.data() + T.spmv.ptr[c];
    const int32_t* hi = T.spmv.dict.data() + T.spmv.ptr[c + 1];
    const int32_t* at = std::lower_bound(lo, hi, r0);
    ctx->cdict_own = hi - at >= r1 - r0 && at[0] == r0 && at[r1 - r0 - 1] == r1 - 1;
  }
  ctx->max_blk_nnz = T.max_blk_nnz;
  if (static_cast<size_t>((ctx->max_blk_nnz + 1) & ~1) * 16 + (RBA + 1) * 4 > 160 * 1024)
    return fail(ctx, HF_ERR_ARG, "row block holds %d nonzeros: LDS slab too large", ctx->max_blk_nnz);
  ctx->tab_len = static_cast<int>(T.tag_used.size());
  ctx->h_tag_used = T.tag_used;
  ctx->assembled = false; ctx->have_mat = false;
  ctx->h_tri.assign(tri, tri + 3 * static_cast<size_t>(ne));
  ctx->h_tag.assign(tag, tag + ne);
  ctx->owner_ready = false;
  ctx->ncolors = 0; ctx->elist_len = 0;
  dev_free(&ctx->d_blk_eptr); dev_free(&ctx->d_blk_cptr); dev_free(&ctx->d_blk_ent);

  std::vector<int4> elem(ne);
  for (int32_t e = 0; e < ne; ++e) elem[e] = make_int4(tri[3 * e], tri[3 * e + 1], tri[3 * e + 2], tag[e]);
  HF_TRY(dev_alloc(ctx, &ctx->d_zr, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_elem, ne));
  HF_TRY(dev_alloc(ctx, &ctx->d_kappa, ctx->tab_len));
  HF_TRY(dev_alloc(ctx, &ctx->d_rhoc, ctx->tab_len));
  HF_TRY(dev_alloc(ctx, &ctx->d_rowptr, n + 1));
  HF_TRY(dev_alloc(ctx, &ctx->d_colidx, ctx->nnz));
  HF_TRY(dev_alloc(ctx, &ctx->d_cdict_ptr, T.spmv.ptr.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_cdict, T.spmv.dict.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_cid, T.spmv.id.size()));
  ctx->n_cdict = static_cast<int64_t>(T.spmv.dict.size());
  HF_HIP(copy_sync(ctx, ctx->d_zr, zr, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  ctx->rg_ok = T.rg.ok && rowgather_smem_bytes(T.max_blk_nnz, T.rg.cols.max_dict) <= 160 * 1024;
  ctx->rg_grid = 0;
  if (ctx->rg_ok) {
    const RowGather& G = T.rg;
    ctx->rg_max_dict = G.cols.max_dict;
    ctx->h_rg_tags = G.tags;
    ctx->n_rg_ell = static_cast<int64_t>(G.ell.size());
    ctx->n_rg_dict = static_cast<int64_t>(G.cols.dict.size());
    HF_TRY(dev_alloc(ctx, &ctx->d_rg_hdr, G.hdr.size()));
    HF_TRY(dev_alloc(ctx, &ctx->d_rg_ell, G.ell.size()));
    HF_TRY(dev_alloc(ctx, &ctx->d_rg_cid, G.cols.id.size() + 16));          // read in 16-byte vectors from an 8-aligned start
    HF_TRY(dev_alloc(ctx, &ctx->d_rg_dict, G.cols.dict.size()));
    HF_TRY(dev_alloc(ctx, &ctx->d_rg_zrb, G.cols.dict.size()));
    HF_TRY(dev_alloc(ctx, &ctx->d_kappa_rg, 64));
    HF_TRY(dev_alloc(ctx, &ctx->d_rhoc_rg, 64));
    HF_HIP(copy_sync(ctx, ctx->d_rg_hdr, G.hdr.data(), sizeof(int4) * G.hdr.size(), hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, ctx->d_rg_ell, G.ell.data(), sizeof(uint16_t) * G.ell.size(), hipMemcpyHostToDevice));
    HF_HIP(hipMemsetAsync(ctx->d_rg_cid + G.cols.id.size(), 0, sizeof(uint16_t) * 16, ctx->stream));
    HF_HIP(copy_sync(ctx, ctx->d_rg_cid, G.cols.id.data(), sizeof(uint16_t) * G.cols.id.size(), hipMemcpyHostToDevice));
    HF_HIP(copy_sync(ctx, ctx->d_rg_dict, G.cols.dict.data(), sizeof(int32_t) * G.cols.dict.size(), hipMemcpyHostToDevice));
    // per-block copies of the column lists' coordinates, gathered on the device
    const int64_t total = ctx->n_rg_dict;
    hipLaunchKernelGGL(k_gather_coords, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, ctx->stream, total, ctx->d_rg_dict,
                       ctx->d_zr, ctx->d_rg_zrb);
    HF_HIP(hipGetLastError());
  }
  HF_TRY(dev_alloc(ctx, &ctx->d_M, ctx->nnz));
  HF_TRY(dev_alloc(ctx, &ctx->d_A, ctx->nnz));
  HF_TRY(dev_alloc(ctx, &ctx->d_dinv, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_u, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_uprev, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_ustart, n));
  ctx->have_prev = false;
  free_responses(ctx);
  HF_TRY(dev_alloc(ctx, &ctx->d_b, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_r, 2 * static_cast<size_t>(n) + 4));   // the level-1 result of a fused finest level lives behind r: [r; x_1]
  HF_HIP(hipMemsetAsync(ctx->d_r, 0, sizeof(double) * (2 * static_cast<size_t>(n) + 4), ctx->stream));
  HF_TRY(dev_alloc(ctx, &ctx->d_p, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_Ap, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_tmp, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_z, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_z2, n));
  free_amg(ctx);
  HF_HIP(copy_sync(ctx, ctx->d_elem, elem.data(), sizeof(int4) * ne, hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_rowptr, T.rowptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_colidx, T.colidx.data(), sizeof(int32_t) * ctx->nnz, hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_cdict_ptr, T.spmv.ptr.data(), sizeof(int32_t) * T.spmv.ptr.size(), hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_cdict, T.spmv.dict.data(), sizeof(int32_t) * T.spmv.dict.size(), hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_cid, T.spmv.id.data(), sizeof(uint16_t) * T.spmv.id.size(), hipMemcpyHostToDevice));
  HF_HIP(hipMemsetAsync(ctx->d_u, 0, sizeof(double) * n, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->h_rowptr.swap(T.rowptr);
  ctx->h_colidx.swap(T.colidx);
  // a new mesh invalidates the Dirichlet set
  ctx->nbc = 0; ctx->nlift = 0; ctx->nlift_rows = 0;
  ctx->have_mesh = true;
  ctx->pred_iters = 0;
  ctx->flux_ready = false;
  if (!ctx->rg_ok) HF_TRY(ensure_owner_lists(ctx));   // the scatter kernels are this mesh's only assembly path: report their limits now
  return HF_OK;
}

// ---- serialised tables (hf_pattern_export / hf_set_mesh_prebuilt): header, then 16-byte aligned sections
struct Section { const void* src; size_t bytes; bool on_device; };

void list_sections(const hf_ctx* c, Section out[12], int64_t count[12]) {
  const int nblk = c->nblk_a, nch = c->nchunks_s;
  const bool rg = c->rg_ok;
  const Section s[12] = {
      {c->h_rowptr.data(), sizeof(int32_t) * (static_cast<size_t>(c->n) + 1), false},
      {c->h_colidx.data(), sizeof(int32_t) * static_cast<size_t>(c->nnz), false},
      {c->d_cdict_ptr, sizeof(int32_t) * (static_cast<size_t>(nch) + 1), true},
      {c->d_cdict, sizeof(int32_t) * static_cast<size_t>(c->n_cdict), true},
      {c->d_cid, sizeof(uint16_t) * static_cast<size_t>(c->nnz), true},
      {c->h_tag_used.data(), c->h_tag_used.size(), false},
      {c->d_rg_hdr, rg ? sizeof(int4) * 2 * static_cast<size_t>(nblk) : 0, true},
      {c->d_rg_ell, rg ? sizeof(uint16_t) * static_cast<size_t>(c->n_rg_ell) : 0, true},
      {c->h_rg_tags.data(), rg ? sizeof(int32_t) * c->h_rg_tags.size() : 0, false},
      {c->d_rg_dict, rg ? sizeof(int32_t) * static_cast<size_t>(c->n_rg_dict) : 0, true},
      {c->d_rg_cid, rg ? sizeof(uint16_t) * static_cast<size_t>(c->nnz) : 0, true},
      {nullptr, 0, false}};
  const size_t unit[12] = {4, 4, 4, 4, 2, 1, 16, 2, 4, 4, 2, 1};
  for (int k = 0; k < 12; ++k) { out[k] = s[k]; count[k] = static_cast<int64_t>(s[k].bytes / unit[k]); }
}

int64_t mesh_tables_bytes(const hf_ctx* c) {
  Section s[12];
  int64_t cnt[12];
  list_sections(c, s, cnt);
  size_t total = pad16(sizeof(BlobHeader));
  for (int k = 0; k < 12; ++k) total += pad16(s[k].bytes);
  return static_cast<int64_t>(total);
}

int write_mesh_tables(hf_ctx* ctx, unsigned char* dst) {
  Section s[12];
  BlobHeader h{};
  list_sections(ctx, s, h.count);
  std::memcpy(h.magic, BLOB_MAGIC, 8);
  h.total_bytes = mesh_tables_bytes(ctx);
  h.nnz = ctx->nnz; h.n = ctx->n; h.ne = ctx->ne; h.rba = RBA; h.ts = SRPC; h.max_blk_nnz = ctx->max_blk_nnz;
  h.tab_len = ctx->tab_len; h.rg_ok = ctx->rg_ok ? 1 : 0; h.rg_max_dict = ctx->rg_max_dict; h.spmv_max_dict = ctx->max_cdict;
  std::memset(dst, 0, static_cast<size_t>(h.total_bytes));
  std::memcpy(dst, &h, sizeof h);
  size_t at = pad16(sizeof(BlobHeader));
  for (int k = 0; k < 12; ++k) {
    if (s[k].bytes) {
      if (s[k].on_device) HF_HIP(hipMemcpyAsync(dst + at, s[k].src, s[k].bytes, hipMemcpyDeviceToHost, ctx->stream));
      else std::memcpy(dst + at, s[k].src, s[k].bytes);
    }
    at += pad16(s[k].bytes);
  }
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

template <typename T>
bool take_section(const unsigned char* blob, int64_t bytes, size_t& at, int64_t count, std::vector<T>& out) {
  if (count < 0) return false;
  const size_t need = sizeof(T) * static_cast<size_t>(count);
  if (at + need > static_cast<size_t>(bytes)) return false;
  out.resize(static_cast<size_t>(count));
  if (need) std::memcpy(out.data(), blob + at, need);
  at += pad16(need);
  return true;
}

int parse_mesh_tables(hf_ctx* ctx, const unsigned char* blob, int64_t bytes, int32_t n, int32_t ne, MeshTables& T) {
  BlobHeader h;
  if (bytes < static_cast<int64_t>(sizeof h)) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: blob too short");
  std::memcpy(&h, blob, sizeof h);
  if (std::memcmp(h.magic, BLOB_MAGIC, 8) != 0) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: not a pattern blob of this library version");
  if (h.total_bytes != bytes) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: blob says %lld bytes, %lld given", (long long)h.total_bytes, (long long)bytes);
  if (h.n != n || h.ne != ne) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: blob was exported for a mesh of %d nodes / %d triangles, not %d / %d", h.n, h.ne, n, ne);
  if (h.rba != RBA || h.ts != SRPC) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: blob built for other block sizes");
  size_t at = pad16(sizeof(BlobHeader));
  std::vector<int4> hdr;
  bool ok = take_section(blob, bytes, at, h.count[0], T.rowptr) && take_section(blob, bytes, at, h.count[1], T.colidx) &&
            take_section(blob, bytes, at, h.count[2], T.spmv.ptr) && take_section(blob, bytes, at, h.count[3], T.spmv.dict) &&
            take_section(blob, bytes, at, h.count[4], T.spmv.id) && take_section(blob, bytes, at, h.count[5], T.tag_used) &&
            take_section(blob, bytes, at, h.count[6], T.rg.hdr) && take_section(blob, bytes, at, h.count[7], T.rg.ell) &&
            take_section(blob, bytes, at, h.count[8], T.rg.tags) && take_section(blob, bytes, at, h.count[9], T.rg.cols.dict) &&
            take_section(blob, bytes, at, h.count[10], T.rg.cols.id);
  const int nblk = (n + RBA - 1) / RBA, nch = (n + SRPC - 1) / SRPC;
  ok = ok && T.rowptr.size() == static_cast<size_t>(n) + 1 && T.rowptr[0] == 0 && T.rowptr[n] == h.nnz &&
       T.colidx.size() == static_cast<size_t>(h.nnz) && T.spmv.ptr.size() == static_cast<size_t>(nch) + 1 &&
       T.spmv.id.size() == T.colidx.size() && static_cast<int>(T.tag_used.size()) == h.tab_len && h.tab_len > 0 &&
       !T.spmv.ptr.empty() && T.spmv.ptr.back() == static_cast<int32_t>(T.spmv.dict.size());
  if (ok && h.rg_ok)
    ok = T.rg.hdr.size() == 2 * static_cast<size_t>(nblk) && T.rg.cols.id.size() == T.colidx.size() && !T.rg.tags.empty() &&
         T.rg.tags.size() <= 64 && h.rg_max_dict > 0 && h.rg_max_dict <= RBA * RG_NX;
  if (!ok) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: blob sections do not fit the mesh");
  // every index the kernels follow without a bounds check is verified once here
  for (int32_t i = 0; i < n && ok; ++i) ok = T.rowptr[i + 1] > T.rowptr[i];
  for (size_t k = 0; k < T.colidx.size() && ok; ++k) ok = T.colidx[k] >= 0 && T.colidx[k] < n;
  for (size_t k = 0; k < T.spmv.dict.size() && ok; ++k) ok = T.spmv.dict[k] >= 0 && T.spmv.dict[k] < n;
  for (int c = 0; c < nch && ok; ++c) {
    const int nd = T.spmv.ptr[c + 1] - T.spmv.ptr[c];
    ok = nd > 0 && nd <= h.spmv_max_dict;
    for (int32_t k = T.rowptr[static_cast<size_t>(c) * SRPC]; k < T.rowptr[std::min<int64_t>(n, (c + 1LL) * SRPC)] && ok; ++k) ok = T.spmv.id[k] < nd;
  }
  if (ok && h.rg_ok) {
    for (size_t k = 0; k < T.rg.cols.dict.size() && ok; ++k) ok = T.rg.cols.dict[k] >= 0 && T.rg.cols.dict[k] < n;
    const int64_t ell16 = static_cast<int64_t>(T.rg.ell.size() / 8);
    for (int b = 0; b < nblk && ok; ++b) {
      const int4 A = T.rg.hdr[2 * b], B = T.rg.hdr[2 * b + 1];
      const int32_t r0 = b * RBA, r1 = std::min<int32_t>(n, r0 + RBA);
      ok = A.x == T.rowptr[r0] && A.y == T.rowptr[r1] - T.rowptr[r0] && A.z >= 0 && A.w > 0 && A.w <= h.rg_max_dict &&
           static_cast<size_t>(A.z) + A.w <= T.rg.cols.dict.size() && B.x >= 0 && B.y >= 1 &&
           static_cast<int64_t>(B.x) + static_cast<int64_t>(B.y) * RBA <= ell16 && B.z >= 0 && B.z + (r1 - r0) <= A.w && B.w == r1 - r0 &&
           A.y <= h.max_blk_nnz;
      for (int32_t k = A.x; k < A.x + A.y && ok; ++k) ok = T.rg.cols.id[k] < A.w;
      for (int32_t i = r0; i < r1 && ok; ++i) {
        const int len = T.rowptr[i + 1] - T.rowptr[i];
        ok = len <= 32;
        for (int g = 0; g < B.y && ok; ++g)
          for (int j = 0; j < 8 && ok; ++j) {
            const unsigned e = T.rg.ell[(static_cast<size_t>(B.x) + static_cast<size_t>(g) * RBA + (i - r0)) * 8 + j];
            ok = e == 0xFFFFu || (static_cast<int>(e & 31u) < len && static_cast<int>((e >> 5) & 31u) < len && (e >> 10) < T.rg.tags.size());
          }
      }
    }
  }
  if (!ok) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: blob holds an index outside its range");
  T.spmv.max_dict = h.spmv_max_dict;
  T.max_blk_nnz = h.max_blk_nnz;
  T.rg.ok = h.rg_ok != 0;
  T.rg.cols.max_dict = h.rg_max_dict;
  return HF_OK;
}

// A blob handed to hf_set_mesh_prebuilt / hf_amg_install may live in host memory (read in place) or in device memory (what an
// RCCL broadcast leaves: staged to the host once - its indices are verified there before any kernel follows them)
int blob_on_host(hf_ctx* ctx, const void* blob, int64_t bytes, std::vector<unsigned char>& staged, const unsigned char** src) {
  hipPointerAttribute_t at{};
  const hipError_t e = hipPointerGetAttributes(&at, blob);
  if (e != hipSuccess) (void)hipGetLastError();            // plain (unregistered) host memory is reported as an error by some runtimes
  const bool on_device = e == hipSuccess && at.type == hipMemoryTypeDevice;
  if (!on_device) { *src = static_cast<const unsigned char*>(blob); return HF_OK; }
  staged.resize(static_cast<size_t>(bytes));
  HF_HIP(hipMemcpy(staged.data(), blob, static_cast<size_t>(bytes), hipMemcpyDeviceToHost));
  *src = staged.data();
  return HF_OK;
}

// coefficient tables of the row-gather kernel: indexed by position in the mesh's tag dictionary
int upload_rg_tables(hf_ctx* ctx, const std::vector<double>& by_tag_k, const std::vector<double>* by_tag_c) {
  if (!ctx->rg_ok) return HF_OK;
  std::vector<double> k(64, 0.0), c(64, 0.0);
  for (size_t q = 0; q < ctx->h_rg_tags.size(); ++q) {
    k[q] = by_tag_k[ctx->h_rg_tags[q]];
    if (by_tag_c) c[q] = (*by_tag_c)[ctx->h_rg_tags[q]];
  }
  HF_HIP(copy_sync(ctx, ctx->d_kappa_rg, k.data(), sizeof(double) * 64, hipMemcpyHostToDevice));
  if (by_tag_c) HF_HIP(copy_sync(ctx, ctx->d_rhoc_rg, c.data(), sizeof(double) * 64, hipMemcpyHostToDevice));
  return HF_OK;
}
// nv-column Jacobi-PCG state of the batched loop's read-flux projection (one gradient component of every column per solve,
// on the unit-coefficient mass matrix of hf_flux_setup); `uprev` keeps the z component's last projection when both are asked for
int ensure_batch_flux(hf_ctx* ctx, int nv, int ncomp) {
  hf_ctx::Batch& F = ctx->fluxnb;
  if (F.nv == nv && (ncomp < 2 || F.uprev != nullptr)) return HF_OK;
  free_batch_state(F);
  const size_t vec = static_cast<size_t>(ctx->n) * nv;
  for (double** v : {&F.u, &F.b, &F.r, &F.p, &F.Ap, &F.z}) {
    HF_TRY(dev_alloc(ctx, v, vec));
    HF_HIP(hipMemsetAsync(*v, 0, sizeof(double) * vec, ctx->stream));
  }
  if (ncomp == 2) {
    HF_TRY(dev_alloc(ctx, &F.uprev, vec));
    HF_HIP(hipMemsetAsync(F.uprev, 0, sizeof(double) * vec, ctx->stream));
  }
  HF_TRY(dev_alloc(ctx, &F.part_pAp, static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &F.part_rz, 2 * static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &F.part_zz, static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &F.part_bn, static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &F.scal, nv));
  HF_TRY(dev_alloc(ctx, &F.red, 1));
  HF_HIP(hipMemsetAsync(F.scal, 0, sizeof(Scal) * nv, ctx->stream));
  HF_HIP(hipMemsetAsync(F.red, 0, sizeof(BRed), ctx->stream));
  if (hipHostMalloc(reinterpret_cast<void**>(&F.h_scal), sizeof(Scal) * nv) != hipSuccess) return fail(ctx, HF_ERR_ALLOC, "hipHostMalloc failed");
  const int rpb = TPB / nv;
  F.Pb = static_cast<int>(std::min<size_t>((static_cast<size_t>(ctx->n) + rpb - 1) / rpb, MAXP));
  if (F.Pb >= 64) F.Pb &= ~7;
  F.opk = 0; F.pred_iters = 0;
  F.sysA = ctx->d_M1; F.sysDinv = ctx->d_dinv1;
  F.lds = ctx->bcols.nv == nv;        // same mesh, same nv: the main batch's staged SpMV serves the projection too
  HF_HIP(hipStreamSynchronize(ctx->stream));
  F.nv = nv;
  return HF_OK;
}

// After a batched step: per wanted component, the gradient right-hand sides of all columns (k_grad_rows on column j of the
// interleaved state), one nv-column Jacobi-PCG warm-started from that component's last projection, and the samples
// out[component][column][node] (reference run_no_diamond.py:543-566, once per run and step).
int batch_flux_step(hf_ctx* ctx, int components, double rtol, int max_it, int nfs, double* out, int32_t* it_out) {
  hf_ctx::Batch& B = ctx->batch;
  hf_ctx::Batch& F = ctx->fluxnb;
  const int nv = B.nv;
  const int capd = ctx->rg_max_dict;
  const size_t sm = static_cast<size_t>(capd) * 16 + static_cast<size_t>(capd + (capd & 1)) * 8 + (RBA + 4) * 4 +
                    (static_cast<size_t>((ctx->max_blk_nnz + 1) & ~1) / 8 + 3) * 16;
  const int grid = std::min(ctx->nblk_a, 2048);
  int done = 0;
  for (int comp = 0; comp < 2; ++comp) {
    if (!((components >> comp) & 1)) continue;
    const bool second = done == 1;                     // the r component after the z component: its own warm start
    if (second) std::swap(F.u, F.uprev);
    for (int j = 0; j < nv; ++j)
      hipLaunchKernelGGL(k_grad_rows, dim3(grid), dim3(RBA), sm, ctx->stream, ctx->nblk_a, capd, ctx->d_rg_hdr,
                         reinterpret_cast<const uint4*>(ctx->d_rg_ell), reinterpret_cast<const uint4*>(ctx->d_rg_cid), ctx->d_rg_zrb,
                         ctx->d_rg_dict, ctx->d_rowptr, B.u, comp == 0 ? F.b + j : static_cast<double*>(nullptr),
                         comp == 1 ? F.b + j : static_cast<double*>(nullptr), nv, nv, j);
    HF_HIP(hipGetLastError());
    std::swap(ctx->batch, ctx->fluxnb);
    const int rc = batch_dispatch(ctx, [&](auto ops) { return decltype(ops)::pcg(ctx, false, rtol, 0.0, max_it); });
    int itmax = 0;
    for (int j = 0; j < nv; ++j) itmax = std::max(itmax, ctx->batch.h_scal[j].iters);
    std::swap(ctx->batch, ctx->fluxnb);
    if (it_out) it_out[done] = itmax;
    if (rc != HF_OK) { if (second) std::swap(F.u, F.uprev); return rc; }
    double* o = out + static_cast<size_t>(done) * nv * nfs;
    const int thr = nfs * nv;
    switch (nv) {
      case 2: hipLaunchKernelGGL((kb_gather<2>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, nfs, ctx->d_fsamp_idx, F.u, o); break;
      case 4: hipLaunchKernelGGL((kb_gather<4>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, nfs, ctx->d_fsamp_idx, F.u, o); break;
      case 8: hipLaunchKernelGGL((kb_gather<8>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, nfs, ctx->d_fsamp_idx, F.u, o); break;
      default: hipLaunchKernelGGL((kb_gather<16>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, nfs, ctx->d_fsamp_idx, F.u, o); break;
    }
    HF_HIP(hipGetLastError());
    if (second) std::swap(F.u, F.uprev);
    ++done;
  }
  return HF_OK;
}
}  // namespace

extern "C" {

const char* hf_version(void) { return "heatflow_hip 0.1 (gfx950)"; }

const char* hf_last_error(const hf_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int hf_create(int device_id, hf_ctx** out) {
  if (!out) return HF_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HF_ERR_HIP;  // no CPU fallback by design
  if (device_id < 0 || device_id >= count) return HF_ERR_ARG;
  hf_ctx* ctx = new hf_ctx();
  ctx->dev = device_id;
  auto bail = [&](int rc) { *out = ctx; return rc; };  // keep ctx so the caller can read the message
  if (hipSetDevice(device_id) != hipSuccess) return bail(fail(ctx, HF_ERR_HIP, "hipSetDevice(%d) failed", device_id));
  // non-blocking: no implicit ordering with the legacy stream, so contexts driven from different host
  // threads (concurrent sweep points) do not serialise on it
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(ctx, HF_ERR_HIP, "hipStreamCreate failed"));
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess)
    return bail(fail(ctx, HF_ERR_HIP, "hipEventCreate failed"));
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_scal), sizeof(Scal)) != hipSuccess)
    return bail(fail(ctx, HF_ERR_ALLOC, "hipHostMalloc failed"));
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_mirror), sizeof(ScalMirror), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_mirror), ctx->h_mirror, 0) != hipSuccess)
    return bail(fail(ctx, HF_ERR_ALLOC, "hipHostMalloc (mapped) failed"));
  *ctx->h_mirror = ScalMirror{};               // epoch 0: no solve of this context (they count from 1) has written it
  int rc = dev_alloc(ctx, &ctx->d_scal, 1);
  if (rc == HF_OK) rc = dev_alloc(ctx, &ctx->d_part_pAp, MAXP);
  if (rc == HF_OK) rc = dev_alloc(ctx, &ctx->d_part_rz, 2 * MAXP);
  if (rc == HF_OK) rc = dev_alloc(ctx, &ctx->d_part_zz, MAXP);
  if (rc == HF_OK) rc = dev_alloc(ctx, &ctx->d_part_bn, MAXP);
  if (rc == HF_OK && reset_scal(ctx) != HF_OK) rc = HF_ERR_HIP;
  if (rc == HF_OK) {     // an empty launch: the library's device code is loaded now (tens of ms once per process), not inside the first solve
    hipLaunchKernelGGL(k_zero_entries, dim3(1), dim3(64), 0, ctx->stream, 0, static_cast<const int32_t*>(nullptr), static_cast<double*>(nullptr));
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, HF_ERR_HIP, "first kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  }
  *out = ctx;
  return rc;
}

int hf_destroy(hf_ctx* ctx) {
  if (!ctx) return HF_ERR_ARG;
  (void)hipSetDevice(ctx->dev);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  dev_free(&ctx->d_zr); dev_free(&ctx->d_elem); dev_free(&ctx->d_kappa); dev_free(&ctx->d_rhoc);
  dev_free(&ctx->d_rowptr); dev_free(&ctx->d_colidx); dev_free(&ctx->d_cdict_ptr); dev_free(&ctx->d_cdict); dev_free(&ctx->d_cid); dev_free(&ctx->d_blk_eptr); dev_free(&ctx->d_blk_cptr); dev_free(&ctx->d_blk_ent); dev_free(&ctx->d_rg_hdr); dev_free(&ctx->d_rg_ell); dev_free(&ctx->d_rg_cid); dev_free(&ctx->d_rg_dict); dev_free(&ctx->d_rg_zrb); dev_free(&ctx->d_kappa_rg); dev_free(&ctx->d_rhoc_rg); dev_free(&ctx->d_M); dev_free(&ctx->d_A); dev_free(&ctx->d_dinv);
  dev_free(&ctx->d_bc_dofs); dev_free(&ctx->d_g); dev_free(&ctx->d_lift_rows); dev_free(&ctx->d_lift_ptr);
  dev_free(&ctx->d_lift_bc); dev_free(&ctx->d_lift_slot); dev_free(&ctx->d_lift_val);
  dev_free(&ctx->d_uprev); dev_free(&ctx->d_ustart);
  dev_free(&ctx->d_u); dev_free(&ctx->d_b); dev_free(&ctx->d_r); dev_free(&ctx->d_p); dev_free(&ctx->d_Ap);
  free_batch(ctx); free_batch_state(ctx->fluxb); free_batch_cols(ctx); free_amg(ctx); free_responses(ctx); proj_free(ctx); dev_free(&ctx->d_z); dev_free(&ctx->d_z2);
  dev_free(&ctx->d_M1); dev_free(&ctx->d_dinv1); dev_free(&ctx->d_gz); dev_free(&ctx->d_gr); dev_free(&ctx->d_bz); dev_free(&ctx->d_br);
  dev_free(&ctx->d_tmp); dev_free(&ctx->d_part_pAp); dev_free(&ctx->d_part_rz); dev_free(&ctx->d_part_zz);
  dev_free(&ctx->d_part_bn); dev_free(&ctx->d_scal); dev_free(&ctx->d_samp_idx); dev_free(&ctx->d_samp); dev_free(&ctx->d_fsamp_idx);
  for (auto& e : ctx->prof_ev) (void)hipEventDestroy(e);
  if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
  if (ctx->h_mirror) (void)hipHostFree(ctx->h_mirror);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return HF_OK;
}

int hf_set_mesh(hf_ctx* ctx, int32_t n, int32_t ne, const double* zr, const int32_t* tri, const int32_t* tag) {
  if (!ctx) return HF_ERR_ARG;
  if (!zr || !tri || !tag || n <= 0 || ne <= 0) return fail(ctx, HF_ERR_ARG, "hf_set_mesh: null pointer or empty mesh");
  HF_HIP(hipSetDevice(ctx->dev));
  int32_t maxtag = 0;
  for (int64_t k = 0; k < 3LL * ne; ++k)
    if (tri[k] < 0 || tri[k] >= n) return fail(ctx, HF_ERR_ARG, "hf_set_mesh: triangle %lld references node %d outside [0,%d)", (long long)(k / 3), tri[k], n);
  for (int32_t e = 0; e < ne; ++e) {
    if (tag[e] < 0) return fail(ctx, HF_ERR_ARG, "hf_set_mesh: negative cell tag at cell %d", e);
    maxtag = std::max(maxtag, tag[e]);
    const double* p0 = zr + 2 * tri[3 * e], *p1 = zr + 2 * tri[3 * e + 1], *p2 = zr + 2 * tri[3 * e + 2];
    const double d = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]);
    if (!(d != 0.0)) return fail(ctx, HF_ERR_ARG, "hf_set_mesh: degenerate triangle %d", e);
  }
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (std::getenv("HEATFLOW_DEBUG")) std::fprintf(stderr, "[set_mesh] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  };
  lap("checked");
  MeshTables T;
  Pattern P;
  HF_TRY(build_csr(ctx, n, ne, tri, P));
  lap("+ csr pattern");
  build_rowgather(n, ne, tri, tag, P);
  lap("+ row-gather lists");
  if (!build_coldict(P.rowptr, P.colidx, n, SRPC, T.spmv)) return fail(ctx, HF_ERR_ARG, "an SpMV chunk touches more than 65535 columns (16-bit positions)");
  lap("+ spmv column lists");
  T.rowptr.swap(P.rowptr);
  T.colidx.swap(P.colidx);
  T.max_blk_nnz = P.max_blk_nnz;
  T.rg = std::move(P.rg);
  T.tag_used.assign(static_cast<size_t>(maxtag) + 1, 0);
  for (int32_t e = 0; e < ne; ++e) T.tag_used[tag[e]] = 1;
  const int rc = install_mesh(ctx, n, ne, zr, tri, tag, T);
  lap("+ installed");
  return rc;
}

int hf_set_mesh_prebuilt(hf_ctx* ctx, int32_t n, int32_t ne, const double* zr, const int32_t* tri, const int32_t* tag,
                         const void* blob, int64_t bytes) {
  if (!ctx) return HF_ERR_ARG;
  if (!zr || !tri || !tag || !blob || n <= 0 || ne <= 0 || bytes <= 0) return fail(ctx, HF_ERR_ARG, "hf_set_mesh_prebuilt: null pointer or empty mesh");
  HF_HIP(hipSetDevice(ctx->dev));
  std::vector<unsigned char> staged;
  const unsigned char* src = nullptr;
  HF_TRY(blob_on_host(ctx, blob, bytes, staged, &src));
  MeshTables T;
  HF_TRY(parse_mesh_tables(ctx, src, bytes, n, ne, T));
  return install_mesh(ctx, n, ne, zr, tri, tag, T);
}

int hf_pattern_export_size(hf_ctx* ctx, int64_t* bytes) {
  if (!ctx || !bytes) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_pattern_export_size before hf_set_mesh");
  *bytes = mesh_tables_bytes(ctx);
  return HF_OK;
}

int hf_pattern_export(hf_ctx* ctx, void* blob, int64_t bytes) {
  if (!ctx || !blob) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_pattern_export before hf_set_mesh");
  if (bytes != mesh_tables_bytes(ctx)) return fail(ctx, HF_ERR_ARG, "hf_pattern_export: buffer of %lld bytes, need %lld", (long long)bytes, (long long)mesh_tables_bytes(ctx));
  HF_HIP(hipSetDevice(ctx->dev));
  std::vector<unsigned char> host(static_cast<size_t>(bytes));
  HF_TRY(write_mesh_tables(ctx, host.data()));
  HF_HIP(hipMemcpy(blob, host.data(), static_cast<size_t>(bytes), hipMemcpyDefault));   // host or device destination
  return HF_OK;
}

int hf_set_materials(hf_ctx* ctx, int32_t n_mat, const int32_t* tags, const double* kappa, const double* rho_c) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_set_materials before hf_set_mesh");
  if (n_mat <= 0 || !tags || !kappa || !rho_c) return fail(ctx, HF_ERR_ARG, "hf_set_materials: bad arguments");
  HF_HIP(hipSetDevice(ctx->dev));
  std::vector<double> tk(ctx->tab_len, std::nan("")), tc(ctx->tab_len, std::nan(""));
  for (int32_t i = 0; i < n_mat; ++i) {
    if (tags[i] < 0) return fail(ctx, HF_ERR_ARG, "hf_set_materials: negative tag");
    if (!(kappa[i] > 0.0) || !(rho_c[i] > 0.0)) return fail(ctx, HF_ERR_ARG, "hf_set_materials: kappa and rho_c must be positive");
    if (tags[i] < ctx->tab_len) { tk[tags[i]] = kappa[i]; tc[tags[i]] = rho_c[i]; }
  }
  // every tag present in the mesh must be mapped (the reference raises KeyError, run_with_diamond.py:291)
  for (int t = 0; t < ctx->tab_len; ++t)
    if (ctx->h_tag_used[t] && std::isnan(tk[t])) return fail(ctx, HF_ERR_ARG, "hf_set_materials: cell tag %d has no material", t);
  HF_HIP(copy_sync(ctx, ctx->d_kappa, tk.data(), sizeof(double) * ctx->tab_len, hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_rhoc, tc.data(), sizeof(double) * ctx->tab_len, hipMemcpyHostToDevice));
  HF_TRY(upload_rg_tables(ctx, tk, &tc));
  ctx->have_mat = true;
  ctx->assembled = false;
  return HF_OK;
}

int hf_update_kappa(hf_ctx* ctx, int32_t n_mat, const int32_t* tags, const double* kappa) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mat || !ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_update_kappa needs a completed hf_assemble first");
  if (n_mat <= 0 || !tags || !kappa) return fail(ctx, HF_ERR_ARG, "hf_update_kappa: bad arguments");
  HF_HIP(hipSetDevice(ctx->dev));
  std::vector<double> tk(ctx->tab_len);
  HF_HIP(copy_sync(ctx, tk.data(), ctx->d_kappa, sizeof(double) * ctx->tab_len, hipMemcpyDeviceToHost));
  for (int32_t i = 0; i < n_mat; ++i) {
    if (tags[i] < 0 || tags[i] >= ctx->tab_len || !ctx->h_tag_used[tags[i]])
      return fail(ctx, HF_ERR_ARG, "hf_update_kappa: tag %d is not a cell tag of the mesh", tags[i]);
    if (!(kappa[i] > 0.0)) return fail(ctx, HF_ERR_ARG, "hf_update_kappa: kappa must be positive");
    tk[tags[i]] = kappa[i];
  }
  HF_HIP(copy_sync(ctx, ctx->d_kappa, tk.data(), sizeof(double) * ctx->tab_len, hipMemcpyHostToDevice));
  HF_TRY(upload_rg_tables(ctx, tk, nullptr));
  return hf_assemble(ctx, ctx->dt, ctx->mode);
}

int hf_set_dirichlet(hf_ctx* ctx, int32_t n_bc, const int32_t* dofs) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_set_dirichlet before hf_set_mesh");
  if (n_bc < 0 || (n_bc > 0 && !dofs)) return fail(ctx, HF_ERR_ARG, "hf_set_dirichlet: bad arguments");
  HF_HIP(hipSetDevice(ctx->dev));
  std::vector<char> seen(ctx->n, 0);
  for (int32_t q = 0; q < n_bc; ++q) {
    if (dofs[q] < 0 || dofs[q] >= ctx->n) return fail(ctx, HF_ERR_ARG, "hf_set_dirichlet: dof %d outside [0,%d)", dofs[q], ctx->n);
    if (seen[dofs[q]]) return fail(ctx, HF_ERR_ARG, "hf_set_dirichlet: dof %d listed twice (resolve overlaps on the host)", dofs[q]);
    seen[dofs[q]] = 1;
  }
  free_batch(ctx);
  ctx->nbc = n_bc;
  HF_TRY(dev_alloc(ctx, &ctx->d_bc_dofs, n_bc));
  HF_TRY(dev_alloc(ctx, &ctx->d_g, n_bc));
  if (n_bc > 0) HF_HIP(copy_sync(ctx, ctx->d_bc_dofs, dofs, sizeof(int32_t) * n_bc, hipMemcpyHostToDevice));
  HF_TRY(build_lift(ctx));
  free_amg(ctx);
  free_responses(ctx);
  ctx->assembled = false;  // A_hat depends on the BC set
  return HF_OK;
}

int hf_assemble(hf_ctx* ctx, double dt, int32_t mode) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh || !ctx->have_mat) return fail(ctx, HF_ERR_STATE, "hf_assemble needs hf_set_mesh and hf_set_materials first");
  if (!(dt > 0.0)) return fail(ctx, HF_ERR_ARG, "hf_assemble: dt must be positive");
  if (mode < 0 || mode > 3) return fail(ctx, HF_ERR_ARG, "hf_assemble: unknown mode %d", mode);
  HF_HIP(hipSetDevice(ctx->dev));
  ctx->dt = dt;
  ctx->mode = mode;
  HF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  HF_TRY(launch_assemble(ctx));
  if (ctx->nbc > 0) {
    if (ctx->nlift > 0)
      hipLaunchKernelGGL(k_take_lift, dim3((ctx->nlift + 255) / 256), dim3(256), 0, ctx->stream, ctx->nlift,
                         ctx->d_lift_slot, ctx->d_A, ctx->d_lift_val);
    hipLaunchKernelGGL(k_bc_rows, dim3((ctx->nbc + 255) / 256), dim3(256), 0, ctx->stream, ctx->nbc, ctx->d_bc_dofs,
                       ctx->d_rowptr, ctx->d_colidx, ctx->d_A);
  }
  hipLaunchKernelGGL(k_dinv, dim3((ctx->n + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, ctx->d_rowptr,
                     ctx->d_colidx, ctx->d_A, ctx->d_dinv);
  HF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HF_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->last_ms = ms;
  if (ctx->precond == 1) {
    OperatorPrint now;
    HF_TRY(operator_print(ctx, now));
    if (!(ctx->amg_ready && ctx->amg_reuse)) {
      HF_TRY(build_amg_auto(ctx));
      ctx->amg_print = std::move(now);
    } else {
      // hierarchy kept (reuse) or installed from another context (hf_amg_install): its fused fine-level operators hold the
      // operator it was built from - usable only if that is this one
      ctx->amg_fine_stale = !same_print(now, ctx->amg_print);
    }
  }
  if (ctx->precond == 1 && ctx->amg_ready) {  // level 0 aliases the fine operator: refresh its pointers
    ctx->amg[0].A.val = ctx->d_A;
    ctx->amg[0].dinv = ctx->d_dinv;
  }
  ctx->assembled = true;
  ctx->pred_iters = 0;
  ctx->have_prev = false;
  free_responses(ctx);   // R depends on the operator
  return HF_OK;
}

int hf_set_precond(hf_ctx* ctx, int32_t kind, int32_t reuse) {
  if (!ctx) return HF_ERR_ARG;
  if (kind < 0 || kind > 1) return fail(ctx, HF_ERR_ARG, "hf_set_precond: unknown preconditioner %d", kind);
  if (kind != ctx->precond) { ctx->assembled = false; ctx->pred_iters = 0; (void)hipSetDevice(ctx->dev); free_batch(ctx); }
  if (kind == 0) { (void)hipSetDevice(ctx->dev); free_amg(ctx); }
  ctx->precond = kind;
  ctx->amg_reuse = reuse ? 1 : 0;
  return HF_OK;
}

int hf_set_start_vector(hf_ctx* ctx, int32_t kind) {
  if (!ctx) return HF_ERR_ARG;
  if (kind < 0 || kind > 3) return fail(ctx, HF_ERR_ARG, "hf_set_start_vector: unknown kind %d", kind);
  ctx->start_kind = kind;
  ctx->extrapolate = kind >= 1 ? 1 : 0;
  if (kind == 0) ctx->have_prev = false;
  return HF_OK;
}

int hf_get_response_solves(hf_ctx* ctx, int64_t* count) {
  if (!ctx || !count) return HF_ERR_ARG;
  *count = ctx->resp_solves;
  return HF_OK;
}

int hf_get_amg_fallbacks(hf_ctx* ctx, int64_t* count) {
  if (!ctx || !count) return HF_ERR_ARG;
  *count = ctx->amg_fallbacks;
  return HF_OK;
}

int hf_get_amg_info(hf_ctx* ctx, int32_t* n_levels, int32_t* level_rows, int32_t max_levels, double* op_complexity,
                    double* setup_seconds) {
  if (!ctx) return HF_ERR_ARG;
  const int nl = ctx->amg_ready ? static_cast<int>(ctx->amg.size()) : 0;
  if (n_levels) *n_levels = nl;
  if (level_rows)
    for (int l = 0; l < nl && l < max_levels; ++l) level_rows[l] = ctx->amg[l].n;
  if (op_complexity) *op_complexity = ctx->amg_ready ? ctx->amg_opc : 0.0;
  if (setup_seconds) *setup_seconds = ctx->amg_ready ? ctx->amg_setup_s : 0.0;
  return HF_OK;
}

int hf_amg_export_size(hf_ctx* ctx, int64_t* bytes) {
  if (!ctx || !bytes) return HF_ERR_ARG;
  if (!ctx->amg_ready || !ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_amg_export_size: no multigrid hierarchy (hf_set_precond(1, ...) and hf_assemble first)");
  BlobOut o;
  o.ctx = ctx;
  HF_TRY(walk_hierarchy(ctx, o, ctx->amg_print));
  *bytes = static_cast<int64_t>(o.at);
  return HF_OK;
}

int hf_amg_export(hf_ctx* ctx, void* blob, int64_t bytes) {
  if (!ctx || !blob) return HF_ERR_ARG;
  if (!ctx->amg_ready || !ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_amg_export: no multigrid hierarchy (hf_set_precond(1, ...) and hf_assemble first)");
  HF_HIP(hipSetDevice(ctx->dev));
  BlobOut sz;
  sz.ctx = ctx;
  HF_TRY(walk_hierarchy(ctx, sz, ctx->amg_print));
  if (bytes != static_cast<int64_t>(sz.at)) return fail(ctx, HF_ERR_ARG, "hf_amg_export: buffer of %lld bytes, need %lld", (long long)bytes, (long long)sz.at);
  std::vector<unsigned char> host(sz.at, 0);
  BlobOut o;
  o.ctx = ctx;
  o.dst = host.data();
  HF_TRY(walk_hierarchy(ctx, o, ctx->amg_print));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  HF_HIP(hipMemcpy(blob, host.data(), host.size(), hipMemcpyDefault));     // host or device destination
  return HF_OK;
}

int hf_amg_install(hf_ctx* ctx, const void* blob, int64_t bytes) {
  if (!ctx || !blob || bytes <= 0) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_amg_install before hf_set_mesh");
  if (ctx->precond != 1 || !ctx->amg_reuse) return fail(ctx, HF_ERR_STATE, "hf_amg_install needs hf_set_precond(1, reuse = 1): an installed hierarchy is a kept one");
  HF_HIP(hipSetDevice(ctx->dev));
  free_batch(ctx);
  std::vector<unsigned char> staged;
  const unsigned char* src = nullptr;
  HF_TRY(blob_on_host(ctx, blob, bytes, staged, &src));
  HF_TRY(install_hierarchy(ctx, src, static_cast<size_t>(bytes)));
  ctx->assembled = false;            // the next hf_assemble compares its operator with the hierarchy's fingerprint
  return HF_OK;
}

int hf_flux_setup(hf_ctx* ctx) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_flux_setup before hf_set_mesh");
  HF_HIP(hipSetDevice(ctx->dev));
  const int n = ctx->n;
  HF_TRY(dev_alloc(ctx, &ctx->d_M1, ctx->nnz));
  HF_TRY(dev_alloc(ctx, &ctx->d_dinv1, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_gz, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_gr, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_bz, n));
  HF_TRY(dev_alloc(ctx, &ctx->d_br, n));
  // M_r(1): the element kernel with rho_c = 1, kappa = 0, dt = 0 (its A output = M goes to scratch)
  const int tab = ctx->rg_ok ? 64 : ctx->tab_len;
  DevTemp<double> t_one, t_zero, t_scratch;
  double *&d_one = t_one.p, *&d_zero = t_zero.p, *&d_scratch = t_scratch.p;
  HF_TRY(dev_alloc(ctx, &d_one, tab));
  HF_TRY(dev_alloc(ctx, &d_zero, tab));
  HF_TRY(dev_alloc(ctx, &d_scratch, ctx->nnz));
  std::vector<double> ones(tab, 1.0), zeros(tab, 0.0);
  HF_HIP(copy_sync(ctx, d_one, ones.data(), sizeof(double) * tab, hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, d_zero, zeros.data(), sizeof(double) * tab, hipMemcpyHostToDevice));
  if (ctx->rg_ok) HF_TRY(launch_assemble_rows(ctx, d_zero, d_one, 0.0, ctx->d_M1, d_scratch));
  else HF_TRY(launch_assemble_lds(ctx, true, d_zero, d_one, 0.0, ctx->d_M1, d_scratch));
  hipLaunchKernelGGL(k_dinv, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ctx->d_rowptr, ctx->d_colidx,
                     ctx->d_M1, ctx->d_dinv1);
  HF_HIP(hipMemsetAsync(ctx->d_gz, 0, sizeof(double) * n, ctx->stream));
  HF_HIP(hipMemsetAsync(ctx->d_gr, 0, sizeof(double) * n, ctx->stream));
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  // two-column PCG state: both gradient components share every pass over M_r(1) (Jacobi-PCG of the batched loop)
  {
    hf_ctx::Batch& F = ctx->fluxb;
    free_batch_state(F);
    const size_t vec = static_cast<size_t>(n) * 2;
    for (double** v : {&F.u, &F.b, &F.r, &F.p, &F.Ap, &F.z}) {
      HF_TRY(dev_alloc(ctx, v, vec));
      HF_HIP(hipMemsetAsync(*v, 0, sizeof(double) * vec, ctx->stream));
    }
    HF_TRY(dev_alloc(ctx, &F.part_pAp, 2 * static_cast<size_t>(MAXP)));
    HF_TRY(dev_alloc(ctx, &F.part_rz, 4 * static_cast<size_t>(MAXP)));
    HF_TRY(dev_alloc(ctx, &F.part_zz, 2 * static_cast<size_t>(MAXP)));
    HF_TRY(dev_alloc(ctx, &F.part_bn, 2 * static_cast<size_t>(MAXP)));
    HF_TRY(dev_alloc(ctx, &F.scal, 2));
    HF_TRY(dev_alloc(ctx, &F.red, 1));
    HF_HIP(hipMemsetAsync(F.scal, 0, sizeof(Scal) * 2, ctx->stream));
    HF_HIP(hipMemsetAsync(F.red, 0, sizeof(BRed), ctx->stream));
    if (hipHostMalloc(reinterpret_cast<void**>(&F.h_scal), sizeof(Scal) * 2) != hipSuccess) return fail(ctx, HF_ERR_ALLOC, "hipHostMalloc failed");
    F.Pb = static_cast<int>(std::min<size_t>((static_cast<size_t>(n) + 127) / 128, MAXP));
    if (F.Pb >= 64) F.Pb &= ~7;
    F.nv = 2; F.opk = 0; F.pred_iters = 0;
    F.sysA = ctx->d_M1; F.sysDinv = ctx->d_dinv1;
    HF_HIP(hipStreamSynchronize(ctx->stream));
  }
  ctx->flux_ready = true;
  ctx->pred_flux[0] = ctx->pred_flux[1] = 0;
  return HF_OK;
}

int hf_flux_solve(hf_ctx* ctx, int32_t components, double rtol, int32_t max_it, int32_t* iters) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->flux_ready) return fail(ctx, HF_ERR_STATE, "hf_flux_solve before hf_flux_setup");
  if (max_it <= 0 || rtol < 0 || components < 0 || components > 3) return fail(ctx, HF_ERR_ARG, "hf_flux_solve: bad arguments");
  HF_HIP(hipSetDevice(ctx->dev));
  const bool both = components == 3 && ctx->rg_ok && ctx->fluxb.nv == 2;
  if (ctx->rg_ok) {
    const int capd = ctx->rg_max_dict;
    const size_t sm = static_cast<size_t>(capd) * 16 + static_cast<size_t>(capd + (capd & 1)) * 8 + (RBA + 4) * 4 +
                      (static_cast<size_t>((ctx->max_blk_nnz + 1) & ~1) / 8 + 3) * 16;
    const int grid = std::min(ctx->nblk_a, 2048);
    hipLaunchKernelGGL(k_grad_rows, dim3(grid), dim3(RBA), sm, ctx->stream, ctx->nblk_a, capd, ctx->d_rg_hdr,
                       reinterpret_cast<const uint4*>(ctx->d_rg_ell), reinterpret_cast<const uint4*>(ctx->d_rg_cid), ctx->d_rg_zrb,
                       ctx->d_rg_dict, ctx->d_rowptr, ctx->d_u, both ? ctx->fluxb.b : ctx->d_bz, both ? ctx->fluxb.b + 1 : ctx->d_br,
                       both ? 2 : 1, 1, 0);
  } else {
    HF_TRY(ensure_owner_lists(ctx));
    hipLaunchKernelGGL(k_grad_rhs, dim3(ctx->nblk_a), dim3(RBA), 0, ctx->stream, ctx->n, ctx->d_blk_eptr, ctx->d_blk_ent,
                       ctx->d_zr, ctx->d_u, ctx->d_bz, ctx->d_br);
  }
  HF_HIP(hipGetLastError());
  if (both) {
    // both components as the two interleaved columns of one Jacobi-PCG on M_r(1): every pass over the matrix serves
    // both (reference: one 2n x 2n vector-P1 solve, run_no_diamond.py:479-489, 544-550); warm start = last projection
    std::swap(ctx->batch, ctx->fluxb);
    const int rc = BatchOps<2, OP_SHARED>::pcg(ctx, false, rtol, 0.0, max_it);
    const int it0 = ctx->batch.h_scal[0].iters, it1 = ctx->batch.h_scal[1].iters;
    std::swap(ctx->batch, ctx->fluxb);
    if (iters) { iters[0] = it0; iters[1] = it1; }
    ctx->flux_valid = 0;
    if (rc != HF_OK) return rc;
    hipLaunchKernelGGL(kb_get_column, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->n), 2, 0, ctx->fluxb.u, ctx->d_gz);
    hipLaunchKernelGGL(kb_get_column, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->n), 2, 1, ctx->fluxb.u, ctx->d_gr);
    HF_HIP(hipGetLastError());
    ctx->flux_valid = 3;
    return HF_OK;
  }
  if (iters) iters[0] = iters[1] = 0;
  ctx->flux_valid = 0;
  // one scalar mass-matrix solve per wanted component, each warm-started from its previous projection
  if (components & 1) {
    const LinSys sz{ctx->d_M1, ctx->d_dinv1, ctx->d_gz, ctx->d_bz};
    const int rc = pcg_solve(ctx, sz, false, rtol, 0.0, max_it, &ctx->pred_flux[0]);
    if (iters) iters[0] = ctx->h_scal->iters;
    if (rc != HF_OK) return rc;
    ctx->flux_valid |= 1;
  }
  if (components & 2) {
    const LinSys sr{ctx->d_M1, ctx->d_dinv1, ctx->d_gr, ctx->d_br};
    const int rc = pcg_solve(ctx, sr, false, rtol, 0.0, max_it, &ctx->pred_flux[1]);
    if (iters) iters[1] = ctx->h_scal->iters;
    if (rc != HF_OK) return rc;
    ctx->flux_valid |= 2;
  }
  return HF_OK;
}

int hf_flux_project(hf_ctx* ctx, double rtol, int32_t max_it, double* grad_z, double* grad_r, int32_t* iters) {
  if (!ctx) return HF_ERR_ARG;
  const int rc = hf_flux_solve(ctx, (grad_z ? 1 : 0) | (grad_r ? 2 : 0), rtol, max_it, iters);
  if (rc != HF_OK) return rc;
  const int n = ctx->n;
  if (grad_z) HF_HIP(hipMemcpyAsync(grad_z, ctx->d_gz, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (grad_r) HF_HIP(hipMemcpyAsync(grad_r, ctx->d_gr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

int hf_flux_sample(hf_ctx* ctx, int32_t ns, const int32_t* nodes, double* grad_z, double* grad_r) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->flux_ready) return fail(ctx, HF_ERR_STATE, "hf_flux_sample before hf_flux_setup");
  if (ns < 0 || (ns > 0 && !nodes)) return fail(ctx, HF_ERR_ARG, "hf_flux_sample: bad arguments");
  if ((grad_z && !(ctx->flux_valid & 1)) || (grad_r && !(ctx->flux_valid & 2)))
    return fail(ctx, HF_ERR_STATE, "hf_flux_sample: that component was not solved by the last hf_flux_solve / hf_flux_project");
  for (int32_t q = 0; q < ns; ++q)
    if (nodes[q] < 0 || nodes[q] >= ctx->n) return fail(ctx, HF_ERR_ARG, "hf_flux_sample: node %d outside [0,%d)", nodes[q], ctx->n);
  if (ns == 0) return HF_OK;
  HF_HIP(hipSetDevice(ctx->dev));
  HF_TRY(ensure_samples(ctx, ns));
  HF_HIP(hipMemcpyAsync(ctx->d_samp_idx, nodes, sizeof(int32_t) * ns, hipMemcpyHostToDevice, ctx->stream));
  for (int comp = 0; comp < 2; ++comp) {
    double* out = comp ? grad_r : grad_z;
    if (!out) continue;
    hipLaunchKernelGGL(k_gather, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx,
                       comp ? ctx->d_gr : ctx->d_gz, ctx->d_samp);
    HF_HIP(hipMemcpyAsync(out, ctx->d_samp, sizeof(double) * ns, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
  }
  return HF_OK;
}

int hf_set_state(hf_ctx* ctx, const double* u) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh || !u) return fail(ctx, HF_ERR_STATE, "hf_set_state: no mesh or null pointer");
  HF_HIP(hipSetDevice(ctx->dev));
  HF_HIP(hipMemcpyAsync(ctx->d_u, u, sizeof(double) * ctx->n, hipMemcpyHostToDevice, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->have_prev = false;
  ctx->g_hist = 0;       // the state no longer continues the recursion the boundary history belongs to
  proj_clear(ctx, true);
  return HF_OK;
}

int hf_get_state(hf_ctx* ctx, double* u) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh || !u) return fail(ctx, HF_ERR_STATE, "hf_get_state: no mesh or null pointer");
  HF_HIP(hipSetDevice(ctx->dev));
  HF_HIP(hipMemcpyAsync(u, ctx->d_u, sizeof(double) * ctx->n, hipMemcpyDeviceToHost, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

int hf_sample(hf_ctx* ctx, int32_t ns, const int32_t* nodes, double* out) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh || ns <= 0 || !nodes || !out) return fail(ctx, HF_ERR_ARG, "hf_sample: bad arguments");
  for (int32_t q = 0; q < ns; ++q)
    if (nodes[q] < 0 || nodes[q] >= ctx->n) return fail(ctx, HF_ERR_ARG, "hf_sample: node %d outside [0,%d)", nodes[q], ctx->n);
  HF_HIP(hipSetDevice(ctx->dev));
  HF_TRY(ensure_samples(ctx, ns));
  HF_HIP(hipMemcpyAsync(ctx->d_samp_idx, nodes, sizeof(int32_t) * ns, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_gather, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx, ctx->d_u, ctx->d_samp);
  HF_HIP(hipMemcpyAsync(out, ctx->d_samp, sizeof(double) * ns, hipMemcpyDeviceToHost, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  return HF_OK;
}

int hf_step(hf_ctx* ctx, const double* g_bc, double rtol, double atol, int32_t max_it, int32_t* iters, double* resid) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_step before hf_assemble");
  if (ctx->nbc > 0 && !g_bc) return fail(ctx, HF_ERR_ARG, "hf_step: g_bc is null");
  if (max_it <= 0 || rtol < 0 || atol < 0) return fail(ctx, HF_ERR_ARG, "hf_step: bad tolerances");
  HF_HIP(hipSetDevice(ctx->dev));
  HF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  const int rc = step_device(ctx, g_bc, nullptr, rtol, atol, max_it);
  if (rc == HF_ERR_HIP) return rc;
  HF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  HF_HIP(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HF_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->last_ms = ms;
  if (iters) *iters = ctx->h_scal->iters;
  if (resid) *resid = std::sqrt(ctx->h_scal->zz / std::max(ctx->h_scal->bn2, 1e-300));
  return rc;
}

int hf_run(hf_ctx* ctx, int32_t n_steps, const double* g_all, double rtol, double atol, int32_t max_it, int32_t ns,
           const int32_t* nodes, double* samples, int32_t* iters) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_run before hf_assemble");
  if (n_steps <= 0 || (ctx->nbc > 0 && !g_all) || max_it <= 0) return fail(ctx, HF_ERR_ARG, "hf_run: bad arguments");
  if (ns < 0 || (ns > 0 && (!nodes || !samples))) return fail(ctx, HF_ERR_ARG, "hf_run: bad sample arguments");
  for (int32_t q = 0; q < ns; ++q)
    if (nodes[q] < 0 || nodes[q] >= ctx->n) return fail(ctx, HF_ERR_ARG, "hf_run: node %d outside [0,%d)", nodes[q], ctx->n);
  HF_HIP(hipSetDevice(ctx->dev));
  DevTemp<double> t_gall, t_sall;
  double *&d_gall = t_gall.p, *&d_sall = t_sall.p;
  if (ctx->nbc > 0) {   // all boundary vectors in one transfer; the steps copy device to device
    HF_TRY(dev_alloc(ctx, &d_gall, static_cast<size_t>(n_steps) * ctx->nbc));
    HF_HIP(copy_sync(ctx, d_gall, g_all, sizeof(double) * n_steps * ctx->nbc, hipMemcpyHostToDevice));
  }
  if (ns > 0) {
    HF_TRY(ensure_samples(ctx, ns));
    HF_TRY(dev_alloc(ctx, &d_sall, static_cast<size_t>(n_steps) * ns));
    HF_HIP(copy_sync(ctx, ctx->d_samp_idx, nodes, sizeof(int32_t) * ns, hipMemcpyHostToDevice));
  }
  int rc = HF_OK;
  HF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  for (int32_t s = 0; s < n_steps && rc == HF_OK; ++s) {
    rc = step_device(ctx, ctx->nbc > 0 ? g_all + static_cast<size_t>(s) * ctx->nbc : nullptr,
                     ctx->nbc > 0 ? d_gall + static_cast<size_t>(s) * ctx->nbc : nullptr, rtol, atol, max_it);
    if (iters) iters[s] = ctx->h_scal->iters;
    if (ns > 0 && rc == HF_OK)
      hipLaunchKernelGGL(k_gather, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx, ctx->d_u,
                         d_sall + static_cast<size_t>(s) * ns);
  }
  (void)hipEventRecord(ctx->ev1, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->last_ms = ms;
  if (ns > 0 && rc == HF_OK) (void)copy_sync(ctx, samples, d_sall, sizeof(double) * n_steps * ns, hipMemcpyDeviceToHost);
  return rc;
}

int hf_batch_begin(hf_ctx* ctx, int32_t nv, int32_t operator_kind) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_batch_begin before hf_assemble");
  if (nv != 2 && nv != 4 && nv != 8 && nv != 16) return fail(ctx, HF_ERR_ARG, "hf_batch_begin: 2, 4, 8 or 16 columns (got %d)", nv);
  if (operator_kind < 0 || operator_kind > 2) return fail(ctx, HF_ERR_ARG, "hf_batch_begin: unknown operator kind %d", operator_kind);
  if (operator_kind == HF_BATCH_PER_COLUMN && ctx->precond == 1 && !ctx->amg_reuse)
    return fail(ctx, HF_ERR_STATE, "hf_batch_begin: per-column operators need the frozen hierarchy (hf_set_precond(1, reuse = 1))");
  HF_HIP(hipSetDevice(ctx->dev));
  free_batch(ctx);
  hf_ctx::Batch& B = ctx->batch;
  const size_t n = static_cast<size_t>(ctx->n), vec = n * nv;
  B.opk = operator_kind;
  for (double& d : B.delta) d = 0.0;
  if (B.opk == HF_BATCH_PER_COLUMN) {
    HF_TRY(dev_alloc(ctx, &B.A, static_cast<size_t>(ctx->nnz) * nv));
    HF_TRY(dev_alloc(ctx, &B.lift_val, static_cast<size_t>(std::max(ctx->nlift, 1)) * nv));
  }
  if (B.opk == HF_BATCH_AFFINE) {
    HF_TRY(dev_alloc(ctx, &B.A1, static_cast<size_t>(ctx->nnz)));
    HF_TRY(dev_alloc(ctx, &B.lift1, static_cast<size_t>(std::max(ctx->nlift, 1))));
  }
  if (B.opk != HF_BATCH_SHARED) HF_TRY(dev_alloc(ctx, &B.dinv, vec));
  for (double** v : {&B.u, &B.uprev, &B.ustart, &B.b, &B.r, &B.p, &B.Ap, &B.z, &B.z2, &B.tmp}) {
    HF_TRY(dev_alloc(ctx, v, vec));
    HF_HIP(hipMemsetAsync(*v, 0, sizeof(double) * vec, ctx->stream));
  }
  HF_TRY(dev_alloc(ctx, &B.part_pAp, static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &B.part_rz, 2 * static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &B.part_zz, static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &B.part_bn, static_cast<size_t>(nv) * MAXP));
  HF_TRY(dev_alloc(ctx, &B.scal, nv));
  HF_HIP(hipMemsetAsync(B.scal, 0, sizeof(Scal) * nv, ctx->stream));
  HF_TRY(dev_alloc(ctx, &B.red, 1));
  HF_HIP(hipMemsetAsync(B.red, 0, sizeof(BRed), ctx->stream));
  for (int k = 0; k < PROJ_MH; ++k) {
    HF_TRY(dev_alloc(ctx, &B.pV[k], vec));
    HF_TRY(dev_alloc(ctx, &B.pF[k], vec));
    B.pused[k] = false;
  }
  HF_TRY(dev_alloc(ctx, &B.pG, static_cast<size_t>(nv) * PROJ_MT * PROJ_MT));
  HF_TRY(dev_alloc(ctx, &B.palpha, static_cast<size_t>(nv) * (PROJ_MT + 1)));
  HF_TRY(dev_alloc(ctx, &B.ppart, static_cast<size_t>(nv) * 2 * PROJ_MT * MAXP));
  HF_HIP(hipMemsetAsync(B.pG, 0, sizeof(double) * nv * PROJ_MT * PROJ_MT, ctx->stream));
  B.pnext = 0; B.ppending = -1;
  if (hipHostMalloc(reinterpret_cast<void**>(&B.h_scal), sizeof(Scal) * nv) != hipSuccess) return fail(ctx, HF_ERR_ALLOC, "hipHostMalloc failed");
  if (hipHostMalloc(reinterpret_cast<void**>(&B.h_mirror), sizeof(ScalMirror) * nv, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&B.d_mirror), B.h_mirror, 0) != hipSuccess)
    return fail(ctx, HF_ERR_ALLOC, "hipHostMalloc (mapped) failed");
  for (int j = 0; j < nv; ++j) B.h_mirror[j] = ScalMirror{};
  B.epoch = 0;
  const int rpb = TPB / nv;
  B.Pb = static_cast<int>(std::min<size_t>((n + rpb - 1) / rpb, MAXP));
  if (B.Pb >= 64) B.Pb &= ~7;
  if (ctx->precond == 1 && ctx->amg_ready) {   // level vectors, laid out as build_amg lays out the single-column ones
    const size_t nl = ctx->amg.size();
    B.lev.resize(nl);
    for (size_t l = 1; l < nl; ++l) {
      const DevLevel& L = ctx->amg[l];
      hf_ctx::BatchLevel& Q = B.lev[l];
      if (l + 1 < nl) {
        const size_t len = (static_cast<size_t>(L.n) + L.P.ncol + 2) * nv;
        HF_TRY(dev_alloc(ctx, &Q.cat, len));
        HF_HIP(hipMemsetAsync(Q.cat, 0, sizeof(double) * len, ctx->stream));
        Q.b = Q.cat;
      } else {
        HF_TRY(dev_alloc(ctx, &Q.b, (static_cast<size_t>(L.n) + 2) * nv));
        Q.own_b = true;
        HF_HIP(hipMemsetAsync(Q.b, 0, sizeof(double) * (static_cast<size_t>(L.n) + 2) * nv, ctx->stream));
      }
      if (l == 1) {
        HF_TRY(dev_alloc(ctx, &Q.x, (static_cast<size_t>(L.n) + 2) * nv));
        Q.res = Q.x;
      } else {
        Q.res = B.lev[l - 1].cat + static_cast<size_t>(ctx->amg[l - 1].n) * nv;
      }
    }
  }
  HF_HIP(hipStreamSynchronize(ctx->stream));
  HF_TRY(ensure_batch_cols(ctx, nv));
  B.lds = ctx->bcols.nv == nv;
  B.nv = nv;
  B.have_prev = false;
  B.pred_iters = 0;
  B.loaded = 0;
  return HF_OK;
}

int hf_batch_end(hf_ctx* ctx) {
  if (!ctx) return HF_ERR_ARG;
  (void)hipSetDevice(ctx->dev);
  free_batch(ctx);
  return HF_OK;
}

int hf_batch_load_column(hf_ctx* ctx, int32_t j) {
  if (!ctx) return HF_ERR_ARG;
  hf_ctx::Batch& B = ctx->batch;
  if (B.nv == 0 || B.opk != HF_BATCH_PER_COLUMN) return fail(ctx, HF_ERR_STATE, "hf_batch_load_column: no batch with per-column operators is open");
  if (!ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_batch_load_column: the context holds no assembled operator");
  if (j < 0 || j >= B.nv) return fail(ctx, HF_ERR_ARG, "hf_batch_load_column: column %d outside [0,%d)", j, B.nv);
  HF_HIP(hipSetDevice(ctx->dev));
  hipLaunchKernelGGL(kb_put_column, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->nnz), B.nv, j, ctx->d_A, B.A);
  hipLaunchKernelGGL(kb_put_column, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->n), B.nv, j, ctx->d_dinv, B.dinv);
  if (ctx->nlift > 0)
    hipLaunchKernelGGL(kb_put_column, dim3(64), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->nlift), B.nv, j, ctx->d_lift_val, B.lift_val);
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  B.loaded |= 1u << j;
  for (bool& u_ : B.pused) u_ = false;     // stored solutions belong to the operators they were computed with
  B.pnext = 0; B.ppending = -1;
  return HF_OK;
}

int hf_batch_set_affine(hf_ctx* ctx, int32_t n_tags, const int32_t* tags, const double* delta) {
  if (!ctx) return HF_ERR_ARG;
  hf_ctx::Batch& B = ctx->batch;
  if (B.nv == 0 || B.opk != HF_BATCH_AFFINE) return fail(ctx, HF_ERR_STATE, "hf_batch_set_affine: no batch with affine operators is open");
  if (!ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_batch_set_affine: the context holds no assembled operator");
  if (n_tags <= 0 || !tags || !delta) return fail(ctx, HF_ERR_ARG, "hf_batch_set_affine: bad arguments");
  for (int32_t q = 0; q < n_tags; ++q)
    if (tags[q] < 0 || tags[q] >= ctx->tab_len || !ctx->h_tag_used[tags[q]])
      return fail(ctx, HF_ERR_ARG, "hf_batch_set_affine: tag %d is not a cell tag of the mesh", tags[q]);
  HF_HIP(hipSetDevice(ctx->dev));
  // A1 = dt K with unit conductivity on the listed materials and zero elsewhere (no mass part): the element kernel
  // with an indicator table; its M output (zero) goes to scratch
  DevTemp<double> t_k, t_c, t_scratch;
  const int tab = ctx->rg_ok ? 64 : ctx->tab_len;
  std::vector<double> ind(tab, 0.0), zeros(tab, 0.0);
  for (int32_t q = 0; q < n_tags; ++q) {
    if (ctx->rg_ok) {
      const auto it = std::lower_bound(ctx->h_rg_tags.begin(), ctx->h_rg_tags.end(), tags[q]);
      ind[it - ctx->h_rg_tags.begin()] = 1.0;
    } else {
      ind[tags[q]] = 1.0;
    }
  }
  HF_TRY(dev_alloc(ctx, &t_k.p, tab));
  HF_TRY(dev_alloc(ctx, &t_c.p, tab));
  HF_TRY(dev_alloc(ctx, &t_scratch.p, ctx->nnz));
  HF_HIP(copy_sync(ctx, t_k.p, ind.data(), sizeof(double) * tab, hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, t_c.p, zeros.data(), sizeof(double) * tab, hipMemcpyHostToDevice));
  if (ctx->rg_ok) HF_TRY(launch_assemble_rows(ctx, t_k.p, t_c.p, ctx->dt, t_scratch.p, B.A1));
  else HF_TRY(launch_assemble_lds(ctx, true, t_k.p, t_c.p, ctx->dt, t_scratch.p, B.A1));
  // Dirichlet elimination of A1: lifting entries kept aside, Dirichlet rows and columns zero (the unit diagonal is A's)
  if (ctx->nbc > 0) {
    if (ctx->nlift > 0)
      hipLaunchKernelGGL(k_zero_slots, dim3((ctx->nlift + 255) / 256), dim3(256), 0, ctx->stream, ctx->nlift, ctx->d_lift_slot, B.A1, B.lift1);
    hipLaunchKernelGGL(k_zero_rows, dim3((ctx->nbc + 255) / 256), dim3(256), 0, ctx->stream, ctx->nbc, ctx->d_bc_dofs, ctx->d_rowptr, B.A1);
  }
  for (int j = 0; j < B.nv; ++j) B.delta[j] = delta[j];
  BOp op{};
  op.v0 = ctx->d_A; op.v1 = B.A1;
  for (int j = 0; j < NV_MAX; ++j) op.delta[j] = B.delta[j];
  const int thr = ctx->n * B.nv;
  switch (B.nv) {
    case 2: hipLaunchKernelGGL((kb_affine_dinv<2>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, ctx->d_rowptr, ctx->d_colidx, op, B.dinv); break;
    case 4: hipLaunchKernelGGL((kb_affine_dinv<4>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, ctx->d_rowptr, ctx->d_colidx, op, B.dinv); break;
    case 8: hipLaunchKernelGGL((kb_affine_dinv<8>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, ctx->d_rowptr, ctx->d_colidx, op, B.dinv); break;
    default: hipLaunchKernelGGL((kb_affine_dinv<16>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ctx->n, ctx->d_rowptr, ctx->d_colidx, op, B.dinv); break;
  }
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  B.loaded = 1;
  for (bool& u_ : B.pused) u_ = false;
  B.pnext = 0; B.ppending = -1;
  return HF_OK;
}

int hf_batch_set_state(hf_ctx* ctx, int32_t j, const double* u) {
  if (!ctx) return HF_ERR_ARG;
  hf_ctx::Batch& B = ctx->batch;
  if (B.nv == 0) return fail(ctx, HF_ERR_STATE, "hf_batch_set_state: no batch is open");
  if (j < 0 || j >= B.nv || !u) return fail(ctx, HF_ERR_ARG, "hf_batch_set_state: bad column or null pointer");
  HF_HIP(hipSetDevice(ctx->dev));
  HF_HIP(copy_sync(ctx, ctx->d_tmp, u, sizeof(double) * ctx->n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kb_put_column, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->n), B.nv, j, ctx->d_tmp, B.u);
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  B.have_prev = false;
  for (bool& u_ : B.pused) u_ = false;     // a new state: the old trajectory's solutions leave every column's basis
  B.pnext = 0; B.ppending = -1;
  return HF_OK;
}

int hf_batch_get_state(hf_ctx* ctx, int32_t j, double* u) {
  if (!ctx) return HF_ERR_ARG;
  hf_ctx::Batch& B = ctx->batch;
  if (B.nv == 0) return fail(ctx, HF_ERR_STATE, "hf_batch_get_state: no batch is open");
  if (j < 0 || j >= B.nv || !u) return fail(ctx, HF_ERR_ARG, "hf_batch_get_state: bad column or null pointer");
  HF_HIP(hipSetDevice(ctx->dev));
  hipLaunchKernelGGL(kb_get_column, dim3(1024), dim3(256), 0, ctx->stream, static_cast<size_t>(ctx->n), B.nv, j, B.u, ctx->d_tmp);
  HF_HIP(hipGetLastError());
  HF_HIP(copy_sync(ctx, u, ctx->d_tmp, sizeof(double) * ctx->n, hipMemcpyDeviceToHost));
  return HF_OK;
}

int hf_batch_run(hf_ctx* ctx, int32_t n_steps, const double* g_all, double rtol, double atol, int32_t max_it, int32_t ns,
                 const int32_t* nodes, double* samples, int32_t* iters) {
  return hf_batch_run_flux(ctx, n_steps, g_all, rtol, atol, max_it, ns, nodes, samples, iters, 0, 0.0, 0, 0, nullptr, nullptr, nullptr);
}

int hf_batch_run_flux(hf_ctx* ctx, int32_t n_steps, const double* g_all, double rtol, double atol, int32_t max_it, int32_t ns,
                      const int32_t* nodes, double* samples, int32_t* iters, int32_t flux_components, double flux_rtol,
                      int32_t flux_max_it, int32_t nfs, const int32_t* flux_nodes, double* flux_samples, int32_t* flux_iters) {
  if (!ctx) return HF_ERR_ARG;
  hf_ctx::Batch& B = ctx->batch;
  if (B.nv == 0) return fail(ctx, HF_ERR_STATE, "hf_batch_run: no batch is open");
  if (!ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_batch_run before hf_assemble");
  if (B.opk == HF_BATCH_PER_COLUMN && B.loaded != (1u << B.nv) - 1u) return fail(ctx, HF_ERR_STATE, "hf_batch_run: not every column's operator has been loaded");
  if (B.opk == HF_BATCH_AFFINE && B.loaded == 0) return fail(ctx, HF_ERR_STATE, "hf_batch_run: hf_batch_set_affine has not been called");
  if (n_steps <= 0 || (ctx->nbc > 0 && !g_all) || max_it <= 0 || rtol < 0 || atol < 0) return fail(ctx, HF_ERR_ARG, "hf_batch_run: bad arguments");
  if (ns < 0 || (ns > 0 && (!nodes || !samples))) return fail(ctx, HF_ERR_ARG, "hf_batch_run: bad sample arguments");
  for (int32_t q = 0; q < ns; ++q)
    if (nodes[q] < 0 || nodes[q] >= ctx->n) return fail(ctx, HF_ERR_ARG, "hf_batch_run: node %d outside [0,%d)", nodes[q], ctx->n);
  const int ncomp = (flux_components & 1) + ((flux_components >> 1) & 1);
  if (flux_components != 0) {
    if (flux_components < 0 || flux_components > 3 || flux_max_it <= 0 || flux_rtol < 0 || nfs <= 0 || !flux_nodes || !flux_samples)
      return fail(ctx, HF_ERR_ARG, "hf_batch_run_flux: bad flux arguments");
    if (!ctx->flux_ready) return fail(ctx, HF_ERR_STATE, "hf_batch_run_flux before hf_flux_setup");
    if (!ctx->rg_ok) return fail(ctx, HF_ERR_ARG, "hf_batch_run_flux: the row-gather lists are not available for this mesh (a row of more than 32 entries or more than 64 cell tags): project run by run");
    for (int32_t q = 0; q < nfs; ++q)
      if (flux_nodes[q] < 0 || flux_nodes[q] >= ctx->n) return fail(ctx, HF_ERR_ARG, "hf_batch_run_flux: node %d outside [0,%d)", flux_nodes[q], ctx->n);
  }
  HF_HIP(hipSetDevice(ctx->dev));
  const int nv = B.nv;
  const size_t gstep = static_cast<size_t>(ctx->nbc) * nv;
  DevTemp<double> t_sall, t_fall;
  if (flux_components != 0) {
    HF_TRY(ensure_batch_flux(ctx, nv, ncomp));
    if (nfs > ctx->fsamp_cap) { HF_TRY(dev_alloc(ctx, &ctx->d_fsamp_idx, nfs)); ctx->fsamp_cap = nfs; }
    HF_HIP(copy_sync(ctx, ctx->d_fsamp_idx, flux_nodes, sizeof(int32_t) * nfs, hipMemcpyHostToDevice));
    HF_TRY(dev_alloc(ctx, &t_fall.p, static_cast<size_t>(n_steps) * ncomp * nv * nfs));
  }
  if (ctx->nbc > 0) {
    HF_TRY(dev_alloc(ctx, &B.g, gstep * n_steps));
    HF_HIP(copy_sync(ctx, B.g, g_all, sizeof(double) * gstep * n_steps, hipMemcpyHostToDevice));
  }
  if (ns > 0) {
    HF_TRY(ensure_samples(ctx, ns));
    HF_TRY(dev_alloc(ctx, &t_sall.p, static_cast<size_t>(n_steps) * nv * ns));
    HF_HIP(copy_sync(ctx, ctx->d_samp_idx, nodes, sizeof(int32_t) * ns, hipMemcpyHostToDevice));
  }
  int rc = HF_OK;
  HF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  for (int32_t s = 0; s < n_steps && rc == HF_OK; ++s) {
    const double* gs = ctx->nbc > 0 ? B.g + gstep * s : nullptr;
    rc = batch_dispatch(ctx, [&](auto ops) { return decltype(ops)::step(ctx, gs, rtol, atol, max_it); });
    if (iters)
      for (int j = 0; j < nv; ++j) iters[static_cast<size_t>(s) * nv + j] = B.h_scal[j].iters;
    if (ns > 0 && rc == HF_OK) {
      double* out = t_sall.p + static_cast<size_t>(s) * nv * ns;
      const int thr = ns * nv;
      switch (nv) {
        case 2: hipLaunchKernelGGL((kb_gather<2>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx, B.u, out); break;
        case 4: hipLaunchKernelGGL((kb_gather<4>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx, B.u, out); break;
        case 8: hipLaunchKernelGGL((kb_gather<8>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx, B.u, out); break;
        default: hipLaunchKernelGGL((kb_gather<16>), dim3((thr + 255) / 256), dim3(256), 0, ctx->stream, ns, ctx->d_samp_idx, B.u, out); break;
      }
    }
    if (flux_components != 0 && rc == HF_OK)
      rc = batch_flux_step(ctx, flux_components, flux_rtol, flux_max_it, nfs, t_fall.p + static_cast<size_t>(s) * ncomp * nv * nfs,
                           flux_iters ? flux_iters + static_cast<size_t>(s) * ncomp : nullptr);
  }
  (void)hipEventRecord(ctx->ev1, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->last_ms = ms;
  if (ns > 0 && rc == HF_OK) (void)copy_sync(ctx, samples, t_sall.p, sizeof(double) * n_steps * nv * ns, hipMemcpyDeviceToHost);
  if (flux_components != 0 && rc == HF_OK)
    (void)copy_sync(ctx, flux_samples, t_fall.p, sizeof(double) * n_steps * ncomp * nv * nfs, hipMemcpyDeviceToHost);
  return rc;
}

int hf_get_sizes(hf_ctx* ctx, int32_t* n, int32_t* ne, int64_t* nnz, int32_t* nbc) {
  if (!ctx) return HF_ERR_ARG;
  if (n) *n = ctx->n;
  if (ne) *ne = ctx->ne;
  if (nnz) *nnz = ctx->nnz;
  if (nbc) *nbc = ctx->nbc;
  return HF_OK;
}

int hf_get_csr(hf_ctx* ctx, int32_t* rowptr, int32_t* colidx, double* A, double* M) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->have_mesh) return fail(ctx, HF_ERR_STATE, "hf_get_csr before hf_set_mesh");
  if ((A || M) && !ctx->assembled) return fail(ctx, HF_ERR_STATE, "hf_get_csr: values requested before hf_assemble");
  HF_HIP(hipSetDevice(ctx->dev));
  if (rowptr) std::memcpy(rowptr, ctx->h_rowptr.data(), sizeof(int32_t) * (ctx->n + 1));
  if (colidx) std::memcpy(colidx, ctx->h_colidx.data(), sizeof(int32_t) * ctx->nnz);
  if (A) HF_HIP(copy_sync(ctx, A, ctx->d_A, sizeof(double) * ctx->nnz, hipMemcpyDeviceToHost));
  if (M) HF_HIP(copy_sync(ctx, M, ctx->d_M, sizeof(double) * ctx->nnz, hipMemcpyDeviceToHost));
  return HF_OK;
}

int hf_spmv(hf_ctx* ctx, int32_t which, const double* x, double* y) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->assembled || !x || !y || which < 0 || which > 1) return fail(ctx, HF_ERR_STATE, "hf_spmv: not assembled or bad arguments");
  HF_HIP(hipSetDevice(ctx->dev));
  HF_HIP(copy_sync(ctx, ctx->d_tmp, x, sizeof(double) * ctx->n, hipMemcpyHostToDevice));
  launch_spmv<0>(ctx, which ? ctx->d_M : ctx->d_A, ctx->d_tmp, ctx->d_Ap);
  HF_HIP(hipGetLastError());
  HF_HIP(hipStreamSynchronize(ctx->stream));
  HF_HIP(copy_sync(ctx, y, ctx->d_Ap, sizeof(double) * ctx->n, hipMemcpyDeviceToHost));
  return HF_OK;
}

int hf_time_kernel(hf_ctx* ctx, int32_t which, int32_t reps, double* ms_avg) {
  if (!ctx) return HF_ERR_ARG;
  if (!ctx->assembled || reps <= 0 || !ms_avg) return fail(ctx, HF_ERR_STATE, "hf_time_kernel: not assembled or bad arguments");
  HF_HIP(hipSetDevice(ctx->dev));
  // Scratch operands only (d_tmp, d_Ap, d_r, d_p are overwritten by the next step anyway);
  // the state u and the matrices are left intact except HF_K_ASSEMBLE, which re-runs the
  // element kernel into M / A and is followed by a full hf_assemble by the caller.
  HF_TRY(reset_scal(ctx));
  HF_HIP(hipMemcpyAsync(ctx->d_tmp, ctx->d_u, sizeof(double) * ctx->n, hipMemcpyDeviceToDevice, ctx->stream));
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 = warm-up
    const int nrep = pass == 0 ? std::min(reps, 3) : reps;
    HF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int k = 0; k < nrep; ++k) {
      switch (which) {
        case HF_K_SPMV: launch_spmv<0>(ctx, ctx->d_A, ctx->d_tmp, ctx->d_Ap); break;
        case HF_K_RHS: launch_spmv<0>(ctx, ctx->d_M, ctx->d_tmp, ctx->d_b); break;
        case HF_K_PCG_SPMV:  // iteration head as in the loop (beta = 1 from the benign partials set below)
          launch_spmv<9>(ctx, ctx->d_A, ctx->d_tmp, ctx->d_Ap, ctx->d_part_pAp, nullptr, ctx->d_p, ctx->d_part_rz,
                         ctx->d_part_zz);
          break;
        case HF_K_PCG_UPDATE:
          // alpha from whatever the partial slots hold: make them benign (pAp = P, rz = 0 -> alpha = 0)
          hipLaunchKernelGGL(k_pcg_update, dim3(ctx->P), dim3(TPB), 0, ctx->stream, ctx->n, ctx->nchunks, ctx->P, 0,
                             ctx->d_scal, ctx->d_part_bn, ctx->d_part_rz, ctx->d_part_zz, ctx->d_r, ctx->d_p,
                             ctx->d_tmp, ctx->d_Ap, ctx->d_dinv, ctx->d_z);
          break;
        case HF_K_PCG_DIR:
          return fail(ctx, HF_ERR_ARG, "HF_K_PCG_DIR: the direction update is fused into the PCG SpMV (HF_K_PCG_SPMV)");
        case HF_K_ASSEMBLE: HF_TRY(launch_assemble(ctx)); break;
        case HF_K_STREAM_READ:
          hipLaunchKernelGGL(k_stream_read, dim3(MAXP), dim3(TS), 0, ctx->stream, static_cast<size_t>(ctx->nnz) / 2,
                             reinterpret_cast<const double2*>(ctx->d_A), static_cast<size_t>(ctx->nnz) / 4,
                             reinterpret_cast<const double2*>(ctx->d_colidx), ctx->d_tmp);
          break;
        default: return fail(ctx, HF_ERR_ARG, "hf_time_kernel: unknown kernel %d", which);
      }
    }
    HF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    HF_HIP(hipGetLastError());
    HF_HIP(hipStreamSynchronize(ctx->stream));
    if (pass == 0 && (which == HF_K_PCG_UPDATE || which == HF_K_PCG_SPMV)) {
      // benign scalars for the timed pass: p.Ap partials = 1, r.z partials = tiny, tol2 = 0, not done
      HF_HIP(hipMemsetAsync(ctx->d_p, 0, sizeof(double) * ctx->n, ctx->stream));
      HF_HIP(hipMemsetAsync(ctx->d_Ap, 0, sizeof(double) * ctx->n, ctx->stream));
      std::vector<double> ones(MAXP, 1.0), tiny(2 * MAXP, 1e-300);
      HF_HIP(copy_sync(ctx, ctx->d_part_bn, ones.data(), sizeof(double) * MAXP, hipMemcpyHostToDevice));
      HF_HIP(copy_sync(ctx, ctx->d_part_rz, tiny.data(), sizeof(double) * 2 * MAXP, hipMemcpyHostToDevice));
      HF_HIP(copy_sync(ctx, ctx->d_part_zz, ones.data(), sizeof(double) * MAXP, hipMemcpyHostToDevice));
      HF_TRY(reset_scal(ctx));
    }
    if (pass == 1) {
      float ms = 0.f;
      HF_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      *ms_avg = static_cast<double>(ms) / nrep;
    }
  }
  if (which == HF_K_ASSEMBLE) ctx->assembled = false;  // BC elimination was undone: caller re-assembles
  return HF_OK;
}

int hf_set_profile(hf_ctx* ctx, int32_t on) {
  if (!ctx) return HF_ERR_ARG;
  HF_HIP(hipSetDevice(ctx->dev));
  if (on && ctx->prof_ev.empty()) {
    ctx->prof_ev.resize(2 * PROF_PAIRS);
    for (auto& e : ctx->prof_ev) HF_HIP(hipEventCreate(&e));
  }
  ctx->prof = on != 0;
  ctx->prof_used = 0;
  ctx->prof_spmv_ms = 0.0;
  ctx->prof_spmv_n = 0;
  return HF_OK;
}

int hf_get_profile(hf_ctx* ctx, double* spmv_ms_sum, int64_t* spmv_launches) {
  if (!ctx || !spmv_ms_sum || !spmv_launches) return HF_ERR_ARG;
  *spmv_ms_sum = ctx->prof_spmv_ms;
  *spmv_launches = ctx->prof_spmv_n;
  return HF_OK;
}

int hf_last_gpu_ms(hf_ctx* ctx, double* ms) {
  if (!ctx || !ms) return HF_ERR_ARG;
  *ms = ctx->last_ms;
  return HF_OK;
}

}  // extern "C"

#if HF_PHASE_CLOCK >= 0
// measurement builds only (see hf_kernels.hpp): the phase stamps of the last k_spmv<HF_PHASE_CLOCK, C16> launch
extern "C" int hf_debug_phases(unsigned long long* out /* MAXP * 16 */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * MAXP * 16) == hipSuccess ? 0 : -1;
}
#endif
