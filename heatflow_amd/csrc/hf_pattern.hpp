// Part of libheatflow_hip.so (see heatflow_hip.hip): host code - sparsity pattern, owner lists and colouring, launch helpers of the Jacobi-PCG loop
#pragma once
#include <thread>

#include "hf_kernels.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// host side: sparsity pattern, owner lists, colouring
// ------------------------------------------------------------------------------------------
// Per-chunk compressed columns: the sorted list of the columns a chunk of consecutive rows touches (its own rows
// plus a halo, ~1.3 entries per row) and a 16-bit position in that list per nonzero.
struct ColDict {
  std::vector<int32_t> ptr, dict;
  std::vector<uint16_t> id;
  int max_dict = 0;
};

// Row-gather assembly lists (k_assemble_rows): per block of RBA rows an ELL slab of 16-bit entries in groups of
// eight visits, laid out [group][lane][8] so that a lane reads its eight entries with one 16-byte load and a
// wavefront reads 1 KB contiguously.  Entry q of lane t describes the q-th triangle at node r0 + t by the
// positions of its next and previous vertex (in the triangle's own cyclic order) inside row t's sorted column
// list (5 bits each) and the index of its cell tag in the mesh's tag dictionary (6 bits); 0xFFFF pads.
// `hdr` holds two int4 per block: (k0, nk, d0, nd) = the block's slice of the CSR arrays and of the column
// lists, and (ELL offset in 16-byte units, groups, position of row r0 in the block's column list, rows).
struct RowGather {
  bool ok = false;
  std::vector<int4> hdr;
  std::vector<uint16_t> ell;
  std::vector<int32_t> tags;       // tag dictionary: dictionary index -> cell tag
  ColDict cols;                    // column lists per RBA-row block
};

// Owner-computes element lists of the LDS scatter kernels (k_assemble_lds, k_grad_rhs): built on demand only,
// the default path (row gather) does not need them.
struct OwnerLists {
  std::vector<int32_t> blk_eptr, blk_cptr, blk_elist;
  std::vector<int2> blk_ent;       // 3 x int2 per list entry: element record + nine slot offsets + ownership mask
  int ncolors = 0;
};

struct Pattern {
  std::vector<int32_t> rowptr, colidx;
  std::vector<int32_t> nptr, nlist;   // node -> incident elements (ascending), the input of the list builders
  int max_blk_nnz = 0;
  RowGather rg;
};

// The connectivity tables of hf_set_mesh are built by a few host threads (HEATFLOW_HOST_THREADS, default 16, at most the hardware threads): every table is a
// loop over independent row blocks whose output sizes are known after a counting pass, so each thread works on a contiguous
// range of blocks and writes a contiguous slice - no per-thread arrays of mesh size, no concatenation.
inline int host_threads() {
  static const int nt = [] {
    int v = std::getenv("HEATFLOW_HOST_THREADS") ? std::atoi(std::getenv("HEATFLOW_HOST_THREADS")) : 16;
    const int hw = static_cast<int>(std::thread::hardware_concurrency());
    if (hw > 0) v = std::min(v, hw);
    return std::max(1, v);
  }();
  return nt;
}

// f(begin, end, thread index) over [0, n) split into contiguous ranges; small loops stay on the calling thread
template <typename F>
void parallel_ranges(int64_t n, int64_t min_per_thread, F&& f) {
  const int nt = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / std::max<int64_t>(1, min_per_thread))));
  if (nt <= 1) { f(static_cast<int64_t>(0), n, 0); return; }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (int t = 1; t < nt; ++t) th.emplace_back([&, t]() { f(n * t / nt, n * (t + 1) / nt, t); });
  f(static_cast<int64_t>(0), n / nt, 0);
  for (auto& x : th) x.join();
}

// false when a chunk touches more than 65535 columns (16-bit positions)
bool build_coldict(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int32_t n, int rows, ColDict& D,
                   int32_t ncol = -1) {
  if (ncol < 0) ncol = n;
  const int nchunk = (n + rows - 1) / rows;
  D.ptr.assign(static_cast<size_t>(nchunk) + 1, 0);
  D.id.resize(colidx.size());
  D.max_dict = 0;
  // per chunk: the distinct columns (marked in a column-indexed array of the thread's own: one pass, no sorting of the
  // chunk's thousands of nonzeros), sorted (the chunk's list), and every nonzero's position in it
  std::vector<std::vector<int32_t>> lists(host_threads());          // each thread's chunks' lists, back to back
  std::vector<int64_t> first_chunk(host_threads() + 1, 0);
  std::vector<char> bad(host_threads(), 0);
  parallel_ranges(nchunk, 64, [&](int64_t c0, int64_t c1, int t) {
    std::vector<int32_t>& out = lists[t];
    std::vector<int32_t> seen(ncol, -1), lid(ncol, 0), list;
    first_chunk[t] = c0;
    for (int64_t c = c0; c < c1; ++c) {
      const int64_t k0 = rowptr[static_cast<size_t>(c) * rows], k1 = rowptr[std::min<int64_t>(n, (c + 1) * rows)];
      list.clear();
      for (int64_t k = k0; k < k1; ++k) {
        const int32_t col = colidx[k];
        if (seen[col] != c) { seen[col] = static_cast<int32_t>(c); list.push_back(col); }
      }
      std::sort(list.begin(), list.end());
      if (list.size() > 65535) { bad[t] = 1; return; }
      for (size_t q = 0; q < list.size(); ++q) lid[list[q]] = static_cast<int32_t>(q);
      for (int64_t k = k0; k < k1; ++k) D.id[k] = static_cast<uint16_t>(lid[colidx[k]]);
      D.ptr[c + 1] = static_cast<int32_t>(list.size());             // lengths now, offsets after the prefix sum below
      out.insert(out.end(), list.begin(), list.end());
    }
  });
  for (char b : bad)
    if (b) return false;
  for (int c = 0; c < nchunk; ++c) {
    D.max_dict = std::max(D.max_dict, D.ptr[c + 1]);
    D.ptr[c + 1] += D.ptr[c];
  }
  D.dict.resize(static_cast<size_t>(D.ptr[nchunk]));
  for (size_t t = 0; t < lists.size(); ++t)
    if (!lists[t].empty()) std::memcpy(D.dict.data() + D.ptr[first_chunk[t]], lists[t].data(), sizeof(int32_t) * lists[t].size());
  return true;
}

// Row-gather lists from the node -> element adjacency (nptr / nlist, elements in ascending order per node).
// Not available (rg.ok = false; the LDS scatter kernels are used instead) when a row holds more than 32 entries,
// the mesh carries more than 64 distinct cell tags or a block's column list is longer than the kernel prefetches.
void build_rowgather(int32_t n, int32_t ne, const int32_t* tri, const int32_t* tag, Pattern& P) {
  const std::vector<int32_t>& nptr = P.nptr;
  const std::vector<int32_t>& nlist = P.nlist;
  RowGather& G = P.rg;
  G.ok = false;
  for (int32_t i = 0; i < n; ++i)
    if (P.rowptr[i + 1] - P.rowptr[i] > 32) return;
  {
    std::vector<int32_t> t(tag, tag + ne);
    std::sort(t.begin(), t.end());
    t.erase(std::unique(t.begin(), t.end()), t.end());
    if (t.size() > 64) return;
    G.tags = t;
  }
  if (!build_coldict(P.rowptr, P.colidx, n, RBA, G.cols)) return;
  if (G.cols.max_dict > RBA * RG_NX || P.max_blk_nnz + 8 > 8 * RBA * RG_NC) return;
  const int nblk = (n + RBA - 1) / RBA;
  G.hdr.resize(2 * static_cast<size_t>(nblk));
  // offsets of the blocks' ELL slabs first (a block's slab is as wide as its busiest node), then the blocks in parallel
  std::vector<size_t> ell_off(static_cast<size_t>(nblk) + 1, 0);
  for (int b = 0; b < nblk; ++b) {
    const int32_t r0 = b * RBA, r1 = std::min<int32_t>(n, r0 + RBA);
    int w = 0;
    for (int32_t i = r0; i < r1; ++i) w = std::max(w, nptr[i + 1] - nptr[i]);
    ell_off[b + 1] = ell_off[b] + static_cast<size_t>(std::max(1, (w + 7) / 8)) * RBA * 8;
  }
  if (ell_off[nblk] / 8 > static_cast<size_t>(INT32_MAX)) return;
  G.ell.resize(ell_off[nblk]);
  parallel_ranges(nblk, 64, [&](int64_t b0, int64_t b1, int) {
  std::fill(G.ell.begin() + ell_off[b0], G.ell.begin() + ell_off[b1], static_cast<uint16_t>(0xFFFF));
  for (int64_t b = b0; b < b1; ++b) {
    const int32_t r0 = static_cast<int32_t>(b) * RBA, r1 = std::min<int32_t>(n, r0 + RBA);
    const size_t off = ell_off[b];
    const int groups = static_cast<int>((ell_off[b + 1] - off) / (static_cast<size_t>(RBA) * 8));
    const int32_t* dict = &G.cols.dict[G.cols.ptr[b]];
    const int32_t* dend = &G.cols.dict[G.cols.ptr[b + 1]];
    G.hdr[2 * b] = make_int4(P.rowptr[r0], P.rowptr[r1] - P.rowptr[r0], G.cols.ptr[b], G.cols.ptr[b + 1] - G.cols.ptr[b]);
    G.hdr[2 * b + 1] = make_int4(static_cast<int>(off / 8), groups, static_cast<int>(std::lower_bound(dict, dend, r0) - dict), r1 - r0);
    for (int32_t i = r0; i < r1; ++i) {
      const int32_t* rb = &P.colidx[P.rowptr[i]];
      const int32_t* re = &P.colidx[P.rowptr[i + 1]];
      const int t = i - r0;
      for (int32_t q = nptr[i]; q < nptr[i + 1]; ++q) {
        const int32_t e = nlist[q];
        int a = 0;
        while (tri[3 * e + a] != i) ++a;
        const uint32_t pj = static_cast<uint32_t>(std::lower_bound(rb, re, tri[3 * e + (a + 1) % 3]) - rb);
        const uint32_t pk = static_cast<uint32_t>(std::lower_bound(rb, re, tri[3 * e + (a + 2) % 3]) - rb);
        const uint32_t tg = static_cast<uint32_t>(std::lower_bound(G.tags.begin(), G.tags.end(), tag[e]) - G.tags.begin());
        const int v = q - nptr[i];
        G.ell[off + (static_cast<size_t>(v >> 3) * RBA + t) * 8 + (v & 7)] = static_cast<uint16_t>(pj | (pk << 5) | (tg << 10));
      }
    }
  }
  });
  G.ok = true;
}

// CSR sparsity pattern of the P1 operators + the node -> element adjacency it is built from.
int build_csr(hf_ctx* ctx, int32_t n, int32_t ne, const int32_t* tri, Pattern& P) {
  std::vector<int32_t>& nptr = P.nptr;
  std::vector<int32_t>& nlist = P.nlist;
  nptr.assign(static_cast<size_t>(n) + 1, 0);
  for (int64_t k = 0; k < 3LL * ne; ++k) nptr[tri[k] + 1]++;
  for (int32_t i = 0; i < n; ++i) nptr[i + 1] += nptr[i];
  nlist.resize(static_cast<size_t>(3) * ne);
  {
    std::vector<int32_t> cur(nptr.begin(), nptr.end() - 1);
    for (int32_t e = 0; e < ne; ++e)
      for (int a = 0; a < 3; ++a) nlist[cur[tri[3 * e + a]]++] = e;
  }
  P.rowptr.assign(static_cast<size_t>(n) + 1, 0);
  // rows in parallel: every thread sorts the vertex lists of a contiguous range of rows into a buffer of its own (the rows'
  // lengths go to rowptr), the prefix sum places the buffers one behind the other
  const int NT = host_threads();
  std::vector<std::vector<int32_t>> cols(NT);
  std::vector<int64_t> first_row(NT, 0);
  std::vector<int32_t> orphan(NT, -1);
  parallel_ranges(n, 4096, [&](int64_t i0, int64_t i1, int t) {
    std::vector<int32_t>& out = cols[t];
    out.reserve(static_cast<size_t>(8) * (i1 - i0));
    first_row[t] = i0;
    int32_t tmp[3 * 64];
    std::vector<int32_t> big;
    for (int64_t i = i0; i < i1; ++i) {
      const int deg = nptr[i + 1] - nptr[i];
      if (deg == 0) { orphan[t] = static_cast<int32_t>(i); return; }
      int32_t* buf = tmp;
      if (deg > 64) { big.resize(static_cast<size_t>(3) * deg); buf = big.data(); }
      int m = 0;
      for (int32_t q = nptr[i]; q < nptr[i + 1]; ++q) {
        const int32_t e = nlist[q];
        buf[m++] = tri[3 * e]; buf[m++] = tri[3 * e + 1]; buf[m++] = tri[3 * e + 2];
      }
      std::sort(buf, buf + m);
      m = static_cast<int>(std::unique(buf, buf + m) - buf);
      out.insert(out.end(), buf, buf + m);
      P.rowptr[i + 1] = m;
    }
  });
  for (int t = 0; t < NT; ++t)
    if (orphan[t] >= 0) return fail(ctx, HF_ERR_ARG, "node %d belongs to no triangle", orphan[t]);
  int64_t total = 0;
  for (int32_t i = 0; i < n; ++i) {
    total += P.rowptr[i + 1];
    if (total > static_cast<int64_t>(INT32_MAX)) return fail(ctx, HF_ERR_ARG, "nnz exceeds int32");
    P.rowptr[i + 1] = static_cast<int32_t>(total);
  }
  P.colidx.resize(static_cast<size_t>(total));
  for (int t = 0; t < NT; ++t)
    if (!cols[t].empty()) std::memcpy(P.colidx.data() + P.rowptr[first_row[t]], cols[t].data(), sizeof(int32_t) * cols[t].size());
  P.max_blk_nnz = 0;
  for (int32_t r0 = 0; r0 < n; r0 += RBA) P.max_blk_nnz = std::max(P.max_blk_nnz, P.rowptr[std::min<int32_t>(n, r0 + RBA)] - P.rowptr[r0]);
  return HF_OK;
}

// Owner lists + greedy colouring per row block (LDS scatter kernels).
int build_owner_lists(hf_ctx* ctx, int32_t n, int32_t ne, const int32_t* tri, const int32_t* tag, const Pattern& P, OwnerLists& O) {
  const std::vector<int32_t>& nptr = P.nptr;
  const std::vector<int32_t>& nlist = P.nlist;
  const int nblk = (n + RBA - 1) / RBA;
  O.blk_eptr.assign(static_cast<size_t>(nblk) + 1, 0);
  O.blk_cptr.assign(static_cast<size_t>(nblk) * (NCOL + 1), 0);
  O.blk_elist.clear();
  O.blk_elist.reserve(static_cast<size_t>(ne) * 3 / 2);
  std::vector<int32_t> stamp(ne, -1), list, color;
  std::vector<uint32_t> mask(RBA);
  O.ncolors = 0;
  for (int b = 0; b < nblk; ++b) {
    const int32_t r0 = b * RBA, r1 = std::min<int32_t>(n, r0 + RBA);
    list.clear();
    for (int32_t i = r0; i < r1; ++i)
      for (int32_t q = nptr[i]; q < nptr[i + 1]; ++q) {
        const int32_t e = nlist[q];
        if (stamp[e] != b) { stamp[e] = b; list.push_back(e); }
      }
    std::sort(list.begin(), list.end());
    std::fill(mask.begin(), mask.end(), 0u);
    color.resize(list.size());
    int counts[NCOL] = {0};
    for (size_t k = 0; k < list.size(); ++k) {
      const int32_t e = list[k];
      uint32_t used = 0;
      for (int a = 0; a < 3; ++a) {
        const int32_t v = tri[3 * e + a];
        if (v >= r0 && v < r1) used |= mask[v - r0];
      }
      if (used == 0xFFFFFFFFu) return fail(ctx, HF_ERR_ARG, "more than %d elements share a node", NCOL);
      const int c = __builtin_ctz(~used);
      color[k] = c;
      counts[c]++;
      O.ncolors = std::max(O.ncolors, c + 1);
      for (int a = 0; a < 3; ++a) {
        const int32_t v = tri[3 * e + a];
        if (v >= r0 && v < r1) mask[v - r0] |= (1u << c);
      }
    }
    const int32_t base = static_cast<int32_t>(O.blk_elist.size());
    int32_t* cp = &O.blk_cptr[static_cast<size_t>(b) * (NCOL + 1)];
    cp[0] = base;
    for (int c = 0; c < NCOL; ++c) cp[c + 1] = cp[c] + counts[c];
    O.blk_elist.resize(O.blk_elist.size() + list.size());
    int32_t cur[NCOL];
    for (int c = 0; c < NCOL; ++c) cur[c] = cp[c];
    for (size_t k = 0; k < list.size(); ++k) O.blk_elist[cur[color[k]]++] = list[k];
    O.blk_eptr[b] = base;
    O.blk_eptr[b + 1] = static_cast<int32_t>(O.blk_elist.size());
  }
  // widen every list entry with the offsets of its nine contributions inside the CSR rows
  O.blk_ent.resize(3 * O.blk_elist.size());
  for (int b = 0; b < nblk; ++b) {
    const int32_t r0 = b * RBA, r1 = std::min<int32_t>(n, r0 + RBA);
    for (int32_t q = O.blk_eptr[b]; q < O.blk_eptr[b + 1]; ++q) {
      const int32_t e = O.blk_elist[q];
      const int32_t nd[3] = {tri[3 * e], tri[3 * e + 1], tri[3 * e + 2]};
      uint32_t pos[9] = {0}, owned = 0;
      for (int a = 0; a < 3; ++a) {
        if (nd[a] < r0 || nd[a] >= r1) continue;
        owned |= 1u << a;
        const int32_t* rb = &P.colidx[P.rowptr[nd[a]]];
        const int32_t* re = &P.colidx[P.rowptr[nd[a] + 1]];
        if (re - rb > 255) return fail(ctx, HF_ERR_ARG, "row %d holds more than 255 entries", nd[a]);
        for (int c = 0; c < 3; ++c) pos[a * 3 + c] = static_cast<uint32_t>(std::lower_bound(rb, re, nd[c]) - rb);
      }
      if (tag[e] >= (1 << 21)) return fail(ctx, HF_ERR_ARG, "cell tag %d does not fit the packed list entry (max 2^21 - 1)", tag[e]);
      const uint32_t w3 = pos[8] | (owned << 8) | (static_cast<uint32_t>(tag[e]) << 11);
      O.blk_ent[3 * q] = make_int2(nd[0], nd[1]);
      O.blk_ent[3 * q + 1] = make_int2(nd[2], static_cast<int>(w3));
      O.blk_ent[3 * q + 2] = make_int2(static_cast<int>(pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24)),
                                       static_cast<int>(pos[4] | (pos[5] << 8) | (pos[6] << 16) | (pos[7] << 24)));
    }
  }
  return HF_OK;
}

// The lists of the LDS scatter kernels on the device, built from the host copies of the mesh the first time a
// kernel that needs them is asked for.
int ensure_owner_lists(hf_ctx* ctx) {
  if (ctx->owner_ready) return HF_OK;
  if (ctx->h_tri.empty()) return fail(ctx, HF_ERR_STATE, "element lists requested without a host copy of the mesh");
  Pattern P;   // adjacency again (cheap next to the colouring), pattern from the context
  HF_TRY(build_csr(ctx, ctx->n, ctx->ne, ctx->h_tri.data(), P));
  OwnerLists O;
  HF_TRY(build_owner_lists(ctx, ctx->n, ctx->ne, ctx->h_tri.data(), ctx->h_tag.data(), P, O));
  ctx->ncolors = O.ncolors;
  ctx->elist_len = static_cast<int64_t>(O.blk_elist.size());
  HF_TRY(dev_alloc(ctx, &ctx->d_blk_eptr, O.blk_eptr.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_blk_cptr, O.blk_cptr.size()));
  HF_TRY(dev_alloc(ctx, &ctx->d_blk_ent, O.blk_ent.size()));
  HF_HIP(copy_sync(ctx, ctx->d_blk_eptr, O.blk_eptr.data(), sizeof(int32_t) * O.blk_eptr.size(), hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_blk_cptr, O.blk_cptr.data(), sizeof(int32_t) * O.blk_cptr.size(), hipMemcpyHostToDevice));
  HF_HIP(copy_sync(ctx, ctx->d_blk_ent, O.blk_ent.data(), sizeof(int2) * O.blk_ent.size(), hipMemcpyHostToDevice));
  ctx->owner_ready = true;
  return HF_OK;
}

size_t spmv_smem_bytes(const hf_ctx* c) { return static_cast<size_t>(c->max_chunk_nnz_s + (c->c16 ? c->max_cdict : 0)) * 8; }

// LDS-staged element kernel into (Mout, Aout) with the given coefficient tables.
int launch_assemble_lds(hf_ctx* ctx, bool colored, const double* kappa_tab, const double* rhoc_tab, double dt,
                        double* Mout, double* Aout) {
  HF_TRY(ensure_owner_lists(ctx));
  const int cap = (ctx->max_blk_nnz + 1) & ~1;  // keep the int array 8-byte aligned
  const size_t sm = static_cast<size_t>(cap) * 16 + (RBA + 1) * 4;
  if (sm > 64 * 1024) {  // beyond the default dynamic-LDS window: opt in (160 KB per CU on gfx950)
    HF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_lds<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sm)));
    HF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_lds<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sm)));
  }
  if (colored)
    hipLaunchKernelGGL(k_assemble_lds<true>, dim3(ctx->nblk_a), dim3(RBA), sm, ctx->stream, ctx->n, cap, ctx->d_rowptr,
                       ctx->d_blk_eptr, ctx->d_blk_cptr, ctx->d_blk_ent, ctx->d_zr, kappa_tab, rhoc_tab, dt, Mout, Aout);
  else
    hipLaunchKernelGGL(k_assemble_lds<false>, dim3(ctx->nblk_a), dim3(RBA), sm, ctx->stream, ctx->n, cap, ctx->d_rowptr,
                       ctx->d_blk_eptr, ctx->d_blk_cptr, ctx->d_blk_ent, ctx->d_zr, kappa_tab, rhoc_tab, dt, Mout, Aout);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

size_t rowgather_smem_bytes(int max_blk_nnz, int max_dict) {
  const size_t cap = static_cast<size_t>((max_blk_nnz + 1) & ~1);
  return cap * 16 + static_cast<size_t>(max_dict) * 16 + (RBA + 4) * 4 + (cap / 8 + 3) * 16;
}

// Row-gather element kernel into (Mout, Aout); coefficient tables indexed by the tag dictionary.
int launch_assemble_rows(hf_ctx* ctx, const double* kappa_idx, const double* rhoc_idx, double dt, double* Mout, double* Aout) {
  const int cap = (ctx->max_blk_nnz + 1) & ~1;
  const int capd = ctx->rg_max_dict;
  const size_t sm = rowgather_smem_bytes(ctx->max_blk_nnz, capd);
  if (ctx->rg_grid == 0) {   // persistent workgroups: as many as fit the chip at this LDS footprint
    if (sm > 64 * 1024)
      HF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 static_cast<int>(sm)));
    int per_cu = 0, ncu = 0;
    HF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_assemble_rows), RBA, sm));
    HF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->dev));
    ctx->rg_grid = std::max(1, std::min(ctx->nblk_a, std::max(1, per_cu) * std::max(1, ncu)));
  }
  hipLaunchKernelGGL(k_assemble_rows, dim3(ctx->rg_grid), dim3(RBA), sm, ctx->stream, ctx->nblk_a, cap, capd, ctx->d_rowptr,
                     ctx->d_rg_hdr, reinterpret_cast<const uint4*>(ctx->d_rg_ell), reinterpret_cast<const uint4*>(ctx->d_rg_cid),
                     ctx->d_rg_zrb, kappa_idx, rhoc_idx, dt, Mout, Aout);
  HF_HIP(hipGetLastError());
  return HF_OK;
}

int launch_assemble(hf_ctx* ctx) {
  if (ctx->mode == HF_ASM_ROW_GATHER && ctx->rg_ok)
    return launch_assemble_rows(ctx, ctx->d_kappa_rg, ctx->d_rhoc_rg, ctx->dt, ctx->d_M, ctx->d_A);
  if (ctx->mode == HF_ASM_ROW_GATHER)   // lists not available for this mesh (row > 32 entries or > 64 cell tags): deterministic LDS variant
    return launch_assemble_lds(ctx, true, ctx->d_kappa, ctx->d_rhoc, ctx->dt, ctx->d_M, ctx->d_A);
  if (ctx->mode == HF_ASM_LDS_COLORED || ctx->mode == HF_ASM_LDS_ATOMIC) {
    return launch_assemble_lds(ctx, ctx->mode == HF_ASM_LDS_COLORED, ctx->d_kappa, ctx->d_rhoc, ctx->dt, ctx->d_M, ctx->d_A);
  } else {
    HF_HIP(hipMemsetAsync(ctx->d_M, 0, sizeof(double) * ctx->nnz, ctx->stream));
    HF_HIP(hipMemsetAsync(ctx->d_A, 0, sizeof(double) * ctx->nnz, ctx->stream));
    hipLaunchKernelGGL(k_assemble_global, dim3((ctx->ne + TPB - 1) / TPB), dim3(TPB), 0, ctx->stream, ctx->ne,
                       ctx->d_rowptr, ctx->d_colidx, ctx->d_elem, ctx->d_zr, ctx->d_kappa, ctx->d_rhoc, ctx->dt,
                       ctx->d_M, ctx->d_A);
  }
  HF_HIP(hipGetLastError());
  return HF_OK;
}

template <int MODE, typename VT = double>
void launch_spmv(hf_ctx* c, const VT* vals, const double* x, double* y, double* part0 = nullptr,
                 const double* bvec = nullptr, double* pvec = nullptr, double* part1 = nullptr,
                 double* part2 = nullptr, double w = 0.0, const double* dinv = nullptr, hipEvent_t ev_start = nullptr,
                 hipEvent_t ev_stop = nullptr, int parity = 0) {
  // With events: the launch carries them (hipExtLaunchKernelGGL), so they bracket the kernel's own
  // execution on the device - the same interval rocprofv3 reports - not the launch gap before it.
  const ColComp comp{c->d_cdict_ptr, c->d_cdict, c->d_cid, c->max_chunk_nnz_s, c->cdict_own ? 1 : 0};
#define HF_SPMV_ARGS c->n, c->nchunks_s, static_cast<int>(SRPC), static_cast<const int32_t*>(c->d_rowptr),                    \
                     static_cast<const int32_t*>(c->d_colidx), vals, x, y, (MODE == 0 ? nullptr : c->d_scal), part0, bvec,  \
                     dinv ? dinv : static_cast<const double*>(c->d_dinv), pvec, part1, part2, w, c->P, parity, comp
  const std::uint32_t smem = static_cast<std::uint32_t>(spmv_smem_bytes(c));
  if (c->c16) {
    if (ev_start != nullptr) hipExtLaunchKernelGGL((k_spmv<MODE, true, VT, HF_SPMV_UN>), dim3(c->Ps), dim3(TS), smem, c->stream, ev_start, ev_stop, 0u, HF_SPMV_ARGS);
    else hipLaunchKernelGGL((k_spmv<MODE, true, VT, HF_SPMV_UN>), dim3(c->Ps), dim3(TS), smem, c->stream, HF_SPMV_ARGS);
  } else {
    if (ev_start != nullptr) hipExtLaunchKernelGGL((k_spmv<MODE, false, VT>), dim3(c->Ps), dim3(TS), smem, c->stream, ev_start, ev_stop, 0u, HF_SPMV_ARGS);
    else hipLaunchKernelGGL((k_spmv<MODE, false, VT>), dim3(c->Ps), dim3(TS), smem, c->stream, HF_SPMV_ARGS);
  }
#undef HF_SPMV_ARGS
}

constexpr int PROF_PAIRS = 64;

// A linear system on the context's sparsity pattern: values, inverse diagonal, unknown, right-hand side.
struct LinSys { const double* A; const double* dinv; double* x; const double* b; };

// One Jacobi-PCG iteration = 2 kernels: [convergence, beta, Ap/p by recurrence, p.Ap] + [alpha, x, r, z, r.z, z.z]
void launch_pcg_iteration(hf_ctx* c, const LinSys& s, int parity) {
  const bool timed = c->prof && c->prof_used < PROF_PAIRS;
  hipEvent_t e0 = timed ? c->prof_ev[2 * c->prof_used] : nullptr, e1 = timed ? c->prof_ev[2 * c->prof_used + 1] : nullptr;
  launch_spmv<9>(c, s.A, c->d_z, c->d_Ap, c->d_part_pAp, nullptr, c->d_p, c->d_part_rz, c->d_part_zz, 0.0, nullptr, e0, e1,
                 parity);
  if (timed) c->prof_used++;
  hipLaunchKernelGGL(k_pcg_update, dim3(c->P), dim3(TPB), 0, c->stream, c->n, c->nchunks, c->P, parity, c->d_scal,
                     c->d_part_pAp, c->d_part_rz, c->d_part_zz, s.x, c->d_r, c->d_p, c->d_Ap, s.dinv, c->d_z);
}

}  // namespace
