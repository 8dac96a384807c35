/* libheatflow_host.so - host-only helpers of the mesh layer (see include/heatflow_host.h). */
#define _POSIX_C_SOURCE 200809L
#include "heatflow_host.h"

#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int hfh_version(void) { return 1; }

/* decimal digits of an int, returns the new write position */
static char* put_int(char* p, long v) {
  char tmp[24];
  int k = 0;
  if (v < 0) { *p++ = '-'; v = -v; }
  if (v == 0) tmp[k++] = '0';
  while (v > 0) { tmp[k++] = (char)('0' + v % 10); v /= 10; }
  while (k > 0) *p++ = tmp[--k];
  return p;
}

int hfh_write_msh22(const char* path, int32_t n, int32_t ne, const double* coords, const int32_t* tris,
                    const int32_t* tags, int32_t n_names, const char* const* names, const int32_t* name_tags) {
  if (!path || n < 0 || ne < 0 || (n > 0 && !coords) || (ne > 0 && (!tris || !tags))) return -EINVAL;
  FILE* f = fopen(path, "w");
  if (!f) return -errno;
  const size_t cap = 1u << 20;
  char* buf = (char*)malloc(cap + 256);
  if (!buf) { fclose(f); return -ENOMEM; }
  int rc = 0;
  fprintf(f, "$MeshFormat\n2.2 0 8\n$EndMeshFormat\n");
  if (n_names > 0 && names && name_tags) {
    fprintf(f, "$PhysicalNames\n%d\n", n_names);
    for (int32_t k = 0; k < n_names; ++k) fprintf(f, "2 %d \"%s\"\n", name_tags[k], names[k]);
    fprintf(f, "$EndPhysicalNames\n");
  }
  fprintf(f, "$Nodes\n%d\n", n);
  char* p = buf;
  for (int32_t i = 0; i < n; ++i) {
    p = put_int(p, (long)i + 1);
    p += snprintf(p, 64, " %.17g %.17g 0\n", coords[2 * (size_t)i], coords[2 * (size_t)i + 1]);
    if ((size_t)(p - buf) > cap) { if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO; p = buf; }
  }
  if (p != buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO;
  fprintf(f, "$EndNodes\n$Elements\n%d\n", ne);
  p = buf;
  for (int32_t e = 0; e < ne; ++e) {
    p = put_int(p, (long)e + 1);
    memcpy(p, " 2 2 ", 5); p += 5;
    p = put_int(p, tags[e]); *p++ = ' ';
    p = put_int(p, tags[e]);
    for (int a = 0; a < 3; ++a) { *p++ = ' '; p = put_int(p, (long)tris[3 * (size_t)e + a] + 1); }
    *p++ = '\n';
    if ((size_t)(p - buf) > cap) { if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO; p = buf; }
  }
  if (p != buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO;
  fprintf(f, "$EndElements\n");
  free(buf);
  if (fclose(f) != 0 && rc == 0) rc = -EIO;
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Quadtree level map of the mesher (heatflow_amd/mesh.py Mesh.build_mesh holds the same algorithm in
 * numpy; tests compare the two).  Dense int8 maps over the padded base grid, row-major [nzp][nrp],
 * nzp and nrp multiples of 2^lmax.
 * ------------------------------------------------------------------------------------------------ */

typedef struct { int8_t* p; int32_t nz, nr; } grid8;

/* The dense passes over the base grid (1e8 cells for a 1M-node mesh) are row-parallel: body(r0, r1, arg) on row ranges,
 * on up to HEATFLOW_HOST_THREADS (default 16) threads when the grid is big enough to pay for starting them. */
typedef void (*row_body)(int32_t r0, int32_t r1, void* arg);
typedef struct { row_body f; void* arg; int32_t r0, r1; } row_job;
static void* row_thread(void* p) { row_job* j = (row_job*)p; j->f(j->r0, j->r1, j->arg); return NULL; }
static int host_threads(void) {
  static int nt = 0;
  if (nt == 0) {
    const char* e = getenv("HEATFLOW_HOST_THREADS");
    int v = e ? atoi(e) : 16;
    const long hw = sysconf(_SC_NPROCESSORS_ONLN);
    if (hw > 0 && v > hw) v = (int)hw;
    nt = v < 1 ? 1 : (v > 64 ? 64 : v);
  }
  return nt;
}
static void par_rows(int32_t nrows, size_t cells, row_body f, void* arg) {
  int nt = host_threads();
  if (cells < ((size_t)1 << 21) || nrows < 2 * nt) nt = 1;
  if (nt == 1) { f(0, nrows, arg); return; }
  pthread_t th[64];
  row_job job[64];
  int started = 0;
  for (int t = 1; t < nt; ++t) {
    job[t].f = f; job[t].arg = arg;
    job[t].r0 = (int32_t)((int64_t)nrows * t / nt); job[t].r1 = (int32_t)((int64_t)nrows * (t + 1) / nt);
    if (pthread_create(&th[t], NULL, row_thread, &job[t]) != 0) { f(job[t].r0, job[t].r1, arg); th[t] = 0; job[t].f = NULL; }
    else ++started;
  }
  f(0, (int32_t)((int64_t)nrows / nt), arg);
  for (int t = 1; t < nt; ++t)
    if (job[t].f) pthread_join(th[t], NULL);
  (void)started;
}

static int grid_alloc(grid8* g, int32_t nz, int32_t nr) {
  g->nz = nz; g->nr = nr;
  g->p = (int8_t*)malloc((size_t)nz * (size_t)nr > 0 ? (size_t)nz * (size_t)nr : 1);
  return g->p ? 0 : -ENOMEM;
}

static inline int8_t min4(int8_t a, int8_t b, int8_t c, int8_t d) { int8_t m = a < b ? a : b; int8_t n = c < d ? c : d; return m < n ? m : n; }
static inline int8_t max4(int8_t a, int8_t b, int8_t c, int8_t d) { int8_t m = a > b ? a : b; int8_t n = c > d ? c : d; return m > n ? m : n; }

/* 2x2 block minimum / maximum of src into dst (dst is half the size) */
typedef struct { const grid8* s; grid8* d; } reduce_arg;
static void reduce_min_rows(int32_t i0, int32_t i1, void* p) {
  const grid8* s = ((reduce_arg*)p)->s; grid8* d = ((reduce_arg*)p)->d;
  for (int32_t i = i0; i < i1; ++i) {
    const int8_t* r0 = s->p + (size_t)(2 * i) * s->nr; const int8_t* r1 = r0 + s->nr; int8_t* o = d->p + (size_t)i * d->nr;
    for (int32_t j = 0; j < d->nr; ++j) o[j] = min4(r0[2 * j], r0[2 * j + 1], r1[2 * j], r1[2 * j + 1]);
  }
}
static void reduce_max_rows(int32_t i0, int32_t i1, void* p) {
  const grid8* s = ((reduce_arg*)p)->s; grid8* d = ((reduce_arg*)p)->d;
  for (int32_t i = i0; i < i1; ++i) {
    const int8_t* r0 = s->p + (size_t)(2 * i) * s->nr; const int8_t* r1 = r0 + s->nr; int8_t* o = d->p + (size_t)i * d->nr;
    for (int32_t j = 0; j < d->nr; ++j) o[j] = max4(r0[2 * j], r0[2 * j + 1], r1[2 * j], r1[2 * j + 1]);
  }
}
static void reduce_min(const grid8* s, grid8* d) { reduce_arg a = {s, d}; par_rows(d->nz, (size_t)s->nz * s->nr, reduce_min_rows, &a); }
static void reduce_max(const grid8* s, grid8* d) { reduce_arg a = {s, d}; par_rows(d->nz, (size_t)s->nz * s->nr, reduce_max_rows, &a); }

typedef struct { const grid8 *cur, *ok; grid8* nxt; int lv; const int8_t* mat; int8_t* level; int32_t nrp; } topdown_arg;
static void topdown_rows(int32_t i0, int32_t i1, void* p) {            /* a cell's level = the highest admissible block above it */
  const topdown_arg* a = (const topdown_arg*)p;
  for (int32_t i = i0; i < i1; ++i)
    for (int32_t j = 0; j < a->nxt->nr; ++j) {
      const int8_t par = a->cur->p ? a->cur->p[(size_t)(i >> 1) * a->cur->nr + (j >> 1)] : (int8_t)-1;
      a->nxt->p[(size_t)i * a->nxt->nr + j] = par >= 0 ? par : (a->ok->p[(size_t)i * a->nxt->nr + j] ? (int8_t)a->lv : (int8_t)-1);
    }
}
static void level0_rows(int32_t i0, int32_t i1, void* p) {
  const topdown_arg* a = (const topdown_arg*)p;
  for (int32_t i = i0; i < i1; ++i)
    for (int32_t j = 0; j < a->nrp; ++j) {
      const int8_t par = a->cur->p ? a->cur->p[(size_t)(i >> 1) * a->cur->nr + (j >> 1)] : (int8_t)-1;
      a->level[(size_t)i * a->nrp + j] = par >= 0 ? par : (a->mat[(size_t)i * a->nrp + j] >= 0 ? (int8_t)0 : (int8_t)-1);
    }
}
typedef struct { const int8_t* level; int8_t* out; int32_t nrp; } pyr0_arg;
static void pyr0_rows(int32_t i0, int32_t i1, void* p) {
  const pyr0_arg* a = (const pyr0_arg*)p;
  for (size_t q = (size_t)i0 * a->nrp; q < (size_t)i1 * a->nrp; ++q) a->out[q] = a->level[q] < 0 ? 127 : a->level[q];
}

typedef struct { const int8_t *mn, *mx, *al; int8_t* ok; int32_t nr; int lv; int any; } ok_arg;
static void ok_rows(int32_t i0, int32_t i1, void* p) {                  /* one material and nothing finer asked for over the block */
  ok_arg* a = (ok_arg*)p;
  int any = 0;
  for (size_t q = (size_t)i0 * a->nr; q < (size_t)i1 * a->nr; ++q) {
    const int8_t v = (int8_t)(a->mn[q] == a->mx[q] && a->mn[q] >= 0 && a->al[q] >= a->lv);
    a->ok[q] = v; any |= v;
  }
  if (any) __atomic_store_n(&a->any, 1, __ATOMIC_RELAXED);
}
/* 2:1 balance, one level: blocks of level lv whose 8 neighbours hold something finer than lv - 1 are marked (read-only pass on
 * the unpatched map of that level), then demoted: the block on the finer pyramid levels and in the level map is rewritten.
 * Blocks are disjoint, so both passes are row-parallel. */
typedef struct { grid8* g; grid8* pyr; int8_t* level; uint8_t* mark; int32_t nrp; int lv; int changed; } bal_arg;
static void bal_mark_rows(int32_t i0, int32_t i1, void* p) {
  bal_arg* a = (bal_arg*)p;
  const grid8* g = a->g;
  const int lv = a->lv;
  for (int32_t i = i0; i < i1; ++i)
    for (int32_t j = 0; j < g->nr; ++j) {
      uint8_t dem = 0;
      if (g->p[(size_t)i * g->nr + j] == lv) {
        int8_t m = 127;
        for (int di = -1; di <= 1; ++di) {
          const int32_t r = i + di; if (r < 0 || r >= g->nz) continue;
          for (int dj = -1; dj <= 1; ++dj) {
            const int32_t c = j + dj; if (c < 0 || c >= g->nr || (di == 0 && dj == 0)) continue;
            const int8_t v = g->p[(size_t)r * g->nr + c]; if (v < m) m = v;
          }
        }
        dem = (uint8_t)(m < lv - 1);
      }
      a->mark[(size_t)i * g->nr + j] = dem;
    }
}
static void bal_apply_rows(int32_t i0, int32_t i1, void* p) {
  bal_arg* a = (bal_arg*)p;
  grid8* g = a->g;
  const int lv = a->lv;
  int changed = 0;
  for (int32_t i = i0; i < i1; ++i)
    for (int32_t j = 0; j < g->nr; ++j) {
      if (!a->mark[(size_t)i * g->nr + j]) continue;
      g->p[(size_t)i * g->nr + j] = (int8_t)(lv - 1);
      for (int l = 0; l < lv; ++l) {              /* the block on the finer pyramid levels and in the level map */
        const int32_t s = 1 << (lv - l);
        grid8* f = &a->pyr[l];
        for (int32_t r = i * s; r < (i + 1) * s; ++r) memset(f->p + (size_t)r * f->nr + (size_t)j * s, lv - 1, (size_t)s);
      }
      const int32_t s0 = 1 << lv;
      for (int32_t r = i * s0; r < (i + 1) * s0; ++r) memset(a->level + (size_t)r * a->nrp + (size_t)j * s0, lv - 1, (size_t)s0);
      changed = 1;
    }
  if (changed) __atomic_store_n(&a->changed, 1, __ATOMIC_RELAXED);
}

int hfh_quadtree_levels(int32_t nzp, int32_t nrp, int32_t lmax, const int8_t* mat, const int8_t* allowed, int8_t* level) {
  if (nzp <= 0 || nrp <= 0 || lmax < 0 || lmax > 30 || !mat || !allowed || !level) return -EINVAL;
  if ((nzp & ((1 << lmax) - 1)) || (nrp & ((1 << lmax) - 1))) return -EINVAL;
  const size_t N = (size_t)nzp * (size_t)nrp;
  int rc = 0;
  /* ---- admissibility per level: one material (>= 0) and allowed >= lv over the aligned block ---- */
  grid8* ok = (grid8*)calloc((size_t)lmax + 1, sizeof(grid8));       /* ok[lv].p: 1 / 0, lv = 1..top */
  if (!ok) return -ENOMEM;
  int top = 0;
  {
    grid8 amin = {0}, mmin = {0}, mmax = {0};
    const grid8 a0 = {(int8_t*)allowed, nzp, nrp}, m0 = {(int8_t*)mat, nzp, nrp};
    const grid8 *pa = &a0, *pmin = &m0, *pmax = &m0;
    for (int lv = 1; lv <= lmax && rc == 0; ++lv) {
      grid8 na = {0}, nmin = {0}, nmax = {0};
      const int32_t hz = nzp >> lv, hr = nrp >> lv;
      if (grid_alloc(&na, hz, hr) || grid_alloc(&nmin, hz, hr) || grid_alloc(&nmax, hz, hr) || grid_alloc(&ok[lv], hz, hr)) { rc = -ENOMEM; free(na.p); free(nmin.p); free(nmax.p); break; }
      reduce_min(pa, &na); reduce_min(pmin, &nmin); reduce_max(pmax, &nmax);
      ok_arg oa = {nmin.p, nmax.p, na.p, ok[lv].p, hr, lv, 0};
      par_rows(hz, (size_t)hz * hr * 4, ok_rows, &oa);
      const int any = oa.any;
      free(amin.p); free(mmin.p); free(mmax.p);
      amin = na; mmin = nmin; mmax = nmax; pa = &amin; pmin = &mmin; pmax = &mmax;
      if (!any) { free(ok[lv].p); ok[lv].p = NULL; break; }
      top = lv;
    }
    free(amin.p); free(mmin.p); free(mmax.p);
  }
  /* ---- a cell's level = the highest admissible block above it: top-down ---- */
  if (rc == 0) {
    grid8 cur = {0};
    for (int lv = top; lv >= 1 && rc == 0; --lv) {     /* cur: assigned level (or -1) on the grid of level lv */
      grid8 nxt;
      if (grid_alloc(&nxt, nzp >> lv, nrp >> lv)) { rc = -ENOMEM; break; }
      topdown_arg ta = {&cur, &ok[lv], &nxt, lv, mat, level, nrp};
      par_rows(nxt.nz, (size_t)nxt.nz * nxt.nr, topdown_rows, &ta);
      free(cur.p);
      cur = nxt;
    }
    if (rc == 0) {
      topdown_arg ta = {&cur, NULL, NULL, 0, mat, level, nrp};
      par_rows(nzp, N, level0_rows, &ta);
    }
    free(cur.p);
  }
  for (int lv = 0; lv <= lmax; ++lv) free(ok[lv].p);
  free(ok);
  if (rc) return rc;
  /* ---- 2:1 balance with smooth grading: a level-L leaf needs each of its 8 same-size neighbour blocks to
   * hold nothing finer than L-1.  Each sweep walks the levels upwards on a min-pyramid that is patched as
   * blocks are demoted; downward ripples take another sweep. ---- */
  grid8* pyr = (grid8*)calloc((size_t)lmax + 1, sizeof(grid8));
  if (!pyr) return -ENOMEM;
  for (int lv = 0; lv <= lmax; ++lv)
    if (grid_alloc(&pyr[lv], nzp >> lv, nrp >> lv)) rc = -ENOMEM;
  /* the min-pyramid is built once and kept consistent: a demotion rewrites its block on every finer level, the
   * coarser levels are re-reduced on the way up (levels 0 and 1 never change a decision, so a sweep costs N/4) */
  if (rc == 0) {
    pyr0_arg pa0 = {level, pyr[0].p, nrp};
    par_rows(nzp, N, pyr0_rows, &pa0);
    if (lmax >= 1) reduce_min(&pyr[0], &pyr[1]);
  }
  uint8_t* mark = (uint8_t*)malloc(lmax >= 2 ? (size_t)(nzp >> 2) * (size_t)(nrp >> 2) : 1);   /* level 2 has the most blocks */
  if (!mark) rc = -ENOMEM;
  int converged = 0;
  for (int sweep = 0; sweep < 2 * (lmax + 2) && rc == 0; ++sweep) {
    int changed = 0;
    for (int lv = 2; lv <= lmax; ++lv) {
      reduce_min(&pyr[lv - 1], &pyr[lv]);
      grid8* g = &pyr[lv];
      /* demotions of one level are decided on the unpatched map of that level (as the vectorised original does) */
      bal_arg ba = {g, pyr, level, mark, nrp, lv, 0};
      par_rows(g->nz, (size_t)g->nz * g->nr * 8, bal_mark_rows, &ba);
      par_rows(g->nz, (size_t)g->nz * g->nr * 8, bal_apply_rows, &ba);
      changed |= ba.changed;
    }
    if (!changed) { converged = 1; break; }
  }
  free(mark);
  for (int lv = 0; lv <= lmax; ++lv) free(pyr[lv].p);
  free(pyr);
  if (rc) return rc;
  return converged ? 0 : -EDOM;
}

/* Leaves of the level map in the mesher's order: by level, row-major within a level.  Writes up to `cap`
 * (i0, j0, lev) triples and returns the total count (call with cap = 0 to size the arrays).  Row-parallel: a counting pass
 * per level gives every row its place in the list, a second pass fills it. */
typedef struct { const int8_t* level; int32_t nrp, hr; int lv; int64_t* rowcnt; int64_t cap; int64_t *i0, *j0, *lev; } leaves_arg;
static void leaves_count_rows(int32_t a, int32_t b, void* p) {
  leaves_arg* g = (leaves_arg*)p;
  for (int32_t i = a; i < b; ++i) {
    const int8_t* row = g->level + ((size_t)i << g->lv) * g->nrp;
    int64_t c = 0;
    for (int32_t j = 0; j < g->hr; ++j) c += row[(size_t)j << g->lv] == g->lv;
    g->rowcnt[i] = c;
  }
}
static void leaves_fill_rows(int32_t a, int32_t b, void* p) {
  leaves_arg* g = (leaves_arg*)p;
  for (int32_t i = a; i < b; ++i) {
    const int8_t* row = g->level + ((size_t)i << g->lv) * g->nrp;
    int64_t at = g->rowcnt[i];                     /* after the prefix sum: the row's first position in the list */
    for (int32_t j = 0; j < g->hr && at < g->cap; ++j)
      if (row[(size_t)j << g->lv] == g->lv) { g->i0[at] = (int64_t)i << g->lv; g->j0[at] = (int64_t)j << g->lv; g->lev[at] = g->lv; ++at; }
  }
}
int64_t hfh_quadtree_leaves(int32_t nzp, int32_t nrp, int32_t lmax, const int8_t* level, int64_t cap, int64_t* i0,
                            int64_t* j0, int64_t* lev) {
  if (nzp <= 0 || nrp <= 0 || lmax < 0 || lmax > 30 || !level) return -EINVAL;
  if (cap > 0 && (!i0 || !j0 || !lev)) return -EINVAL;
  int64_t* rowcnt = (int64_t*)malloc(sizeof(int64_t) * (size_t)nzp);
  if (!rowcnt) return -ENOMEM;
  int64_t cnt = 0;
  for (int lv = 0; lv <= lmax; ++lv) {
    const int32_t hz = nzp >> lv, hr = nrp >> lv;
    if (hz <= 0 || hr <= 0) break;
    leaves_arg g = {level, nrp, hr, lv, rowcnt, cap, i0, j0, lev};
    par_rows(hz, (size_t)hz * hr, leaves_count_rows, &g);
    for (int32_t i = 0; i < hz; ++i) { const int64_t c = rowcnt[i]; rowcnt[i] = cnt; cnt += c; }
    if (cap > 0) par_rows(hz, (size_t)hz * hr, leaves_fill_rows, &g);
  }
  free(rowcnt);
  return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * Nodes and triangles of the mesh from the quadtree's leaves (the second half of Mesh.build_mesh; its numpy
 * statement stays in heatflow_amd/mesh.py and the tests compare the two bit for bit).
 *
 * Node lattice: (nzp + 1) x (nrp + 1) points, key = i * (nrp + 1) + j.  Two bitmaps over it - corners of leaves, and
 * corners + centres of the leaves that fan - replace the sorted key arrays of the numpy path: "is this edge midpoint a
 * node" is a bit test, and a node's number is perm[rank of its bit], with the rank from a per-word prefix count.  Nodes
 * are numbered by the Morton code of (i, j), triangles are listed leaf by leaf in the Morton order of the leaf centres
 * (both codes from the low 16 bits, as in the numpy path: callers take that path for lattices beyond 65535).
 * ------------------------------------------------------------------------------------------------ */
struct hfh_mesh {
  int64_t n_nodes, n_tris, n_fan;
  double* coords;      /* n_nodes x 2 */
  int64_t* node_ij;    /* n_nodes x 2 */
  int32_t* tris;       /* n_tris x 3 */
  int32_t* tags;       /* n_tris */
};

static inline uint32_t spread16(uint32_t v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
static inline uint32_t morton16(int64_t i, int64_t j) { return spread16((uint32_t)i) | (spread16((uint32_t)j) << 1); }

/* stable LSD radix sort of (code, payload) pairs by the 32-bit code, three passes of 11 bits; result in (code, pay) */
static int radix_sort_u32(int64_t n, uint32_t* code, int32_t* pay) {
  uint32_t* c2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* p2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  size_t* cnt = (size_t*)malloc(sizeof(size_t) * 2048);
  if (!c2 || !p2 || !cnt) { free(c2); free(p2); free(cnt); return -ENOMEM; }
  uint32_t *ca = code, *cb = c2;
  int32_t *pa = pay, *pb = p2;
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = 11 * pass;
    memset(cnt, 0, sizeof(size_t) * 2048);
    for (int64_t k = 0; k < n; ++k) ++cnt[(ca[k] >> sh) & 2047u];
    size_t at = 0;
    for (int b = 0; b < 2048; ++b) { const size_t c = cnt[b]; cnt[b] = at; at += c; }
    for (int64_t k = 0; k < n; ++k) { const size_t d = cnt[(ca[k] >> sh) & 2047u]++; cb[d] = ca[k]; pb[d] = pa[k]; }
    uint32_t* tc = ca; ca = cb; cb = tc;
    int32_t* tp = pa; pa = pb; pb = tp;
  }
  /* three passes: the result sits in the scratch arrays */
  memcpy(code, ca, sizeof(uint32_t) * (size_t)n);
  memcpy(pay, pa, sizeof(int32_t) * (size_t)n);
  free(c2); free(p2); free(cnt);
  return 0;
}

typedef struct {
  int64_t nleaf;
  const int64_t *i0, *j0, *lev;
  int64_t stride;
  uint64_t *corner, *node;
  uint8_t* flags;          /* bit 0..3: hanging node on the bottom / z1 / top / z0 edge */
  int32_t* ntri;
  /* numbering */
  const uint32_t* rank;    /* set bits of `node` before each word */
  const int32_t* perm;     /* row-major rank -> node number */
  /* output */
  const int32_t* lorder;   /* position -> leaf */
  const int64_t* toff;     /* position -> first triangle */
  const int8_t* mat;
  int32_t nrp;
  const double* coords;
  int32_t* tris;
  int32_t* tags;
  int degenerate;
} mesh_arg;

static inline void bit_set(uint64_t* b, int64_t k) { __atomic_fetch_or(&b[k >> 6], (uint64_t)1 << (k & 63), __ATOMIC_RELAXED); }
static inline int bit_get(const uint64_t* b, int64_t k) { return (int)((b[k >> 6] >> (k & 63)) & 1u); }

static void mesh_corner_rows(int32_t a, int32_t b, void* p) {
  mesh_arg* m = (mesh_arg*)p;
  for (int64_t q = a; q < b; ++q) {
    const int64_t i0 = m->i0[q], j0 = m->j0[q], sz = (int64_t)1 << m->lev[q];
    bit_set(m->corner, i0 * m->stride + j0);
    bit_set(m->corner, (i0 + sz) * m->stride + j0);
    bit_set(m->corner, (i0 + sz) * m->stride + j0 + sz);
    bit_set(m->corner, i0 * m->stride + j0 + sz);
  }
}

static void mesh_flag_rows(int32_t a, int32_t b, void* p) {
  mesh_arg* m = (mesh_arg*)p;
  for (int64_t q = a; q < b; ++q) {
    uint8_t f = 0;
    if (m->lev[q] >= 1) {
      const int64_t i0 = m->i0[q], j0 = m->j0[q], sz = (int64_t)1 << m->lev[q], h = sz >> 1;
      f = (uint8_t)(bit_get(m->corner, (i0 + h) * m->stride + j0) | (bit_get(m->corner, (i0 + sz) * m->stride + j0 + h) << 1) |
                    (bit_get(m->corner, (i0 + h) * m->stride + j0 + sz) << 2) | (bit_get(m->corner, i0 * m->stride + j0 + h) << 3));
      if (f) bit_set(m->node, (i0 + h) * m->stride + j0 + h);      /* the fan's centre node */
    }
    m->flags[q] = f;
    m->ntri[q] = f ? 4 + (f & 1) + ((f >> 1) & 1) + ((f >> 2) & 1) + ((f >> 3) & 1) : 2;
  }
}

static inline int32_t node_id(const mesh_arg* m, int64_t key) {
  const uint64_t w = m->node[key >> 6];
  return m->perm[m->rank[key >> 6] + (uint32_t)__builtin_popcountll(w & (((uint64_t)1 << (key & 63)) - 1))];
}

static inline void put_tri(mesh_arg* m, int64_t t, int32_t a, int32_t b, int32_t c, int32_t tag) {
  const double* pa = m->coords + 2 * (size_t)a;
  const double* pb = m->coords + 2 * (size_t)b;
  const double* pc = m->coords + 2 * (size_t)c;
  const double area2 = (pb[0] - pa[0]) * (pc[1] - pa[1]) - (pc[0] - pa[0]) * (pb[1] - pa[1]);
  if (area2 == 0.0) m->degenerate = 1;
  int32_t* o = m->tris + 3 * (size_t)t;
  o[0] = a;
  if (area2 < 0.0) { o[1] = c; o[2] = b; } else { o[1] = b; o[2] = c; }     /* counter-clockwise in the (z, r) plane */
  m->tags[t] = tag;
}

static void mesh_tri_rows(int32_t a, int32_t b, void* p) {
  mesh_arg* m = (mesh_arg*)p;
  for (int64_t pos = a; pos < b; ++pos) {
    const int64_t q = m->lorder[pos];
    const int64_t i0 = m->i0[q], j0 = m->j0[q], sz = (int64_t)1 << m->lev[q], h = sz >> 1, i1 = i0 + sz, j1 = j0 + sz;
    const int32_t tag = (int32_t)m->mat[(size_t)i0 * m->nrp + j0] + 1;
    const int32_t c00 = node_id(m, i0 * m->stride + j0), c10 = node_id(m, i1 * m->stride + j0);
    const int32_t c11 = node_id(m, i1 * m->stride + j1), c01 = node_id(m, i0 * m->stride + j1);
    int64_t t = m->toff[pos];
    const uint8_t f = m->flags[q];
    if (!f) {                                   /* two right triangles sharing the (c00, c11) diagonal */
      put_tri(m, t, c00, c10, c11, tag);
      put_tri(m, t + 1, c00, c11, c01, tag);
      continue;
    }
    const int32_t cc = node_id(m, (i0 + h) * m->stride + j0 + h);
    const int32_t ea[4] = {c00, c10, c11, c01}, eb[4] = {c10, c11, c01, c00};
    const int64_t mk[4] = {(i0 + h) * m->stride + j0, i1 * m->stride + j0 + h, (i0 + h) * m->stride + j1, i0 * m->stride + j0 + h};
    for (int s = 0; s < 4; ++s) {
      if ((f >> s) & 1) {
        const int32_t mid = node_id(m, mk[s]);
        put_tri(m, t++, ea[s], mid, cc, tag);
        put_tri(m, t++, mid, eb[s], cc, tag);
      } else {
        put_tri(m, t++, ea[s], eb[s], cc, tag);
      }
    }
  }
}

void hfh_mesh_free(hfh_mesh* h) {
  if (!h) return;
  free(h->coords); free(h->node_ij); free(h->tris); free(h->tags);
  free(h);
}

int hfh_mesh_build(int64_t nleaf, const int64_t* i0, const int64_t* j0, const int64_t* lev, int32_t nzp, int32_t nrp,
                   const int8_t* mat, int32_t nz, int32_t nr, const double* zc, const double* rc, hfh_mesh** out) {
  if (!out) return -EINVAL;
  *out = NULL;
  if (nleaf <= 0 || nleaf > 0x7fffffffLL / 8 || !i0 || !j0 || !lev || !mat || !zc || !rc || nzp <= 0 || nrp <= 0 || nzp > 65535 ||
      nrp > 65535 || nz <= 0 || nr <= 0 || nz > nzp || nr > nrp)
    return -EINVAL;
  for (int64_t q = 0; q < nleaf; ++q) {
    const int64_t sz = lev[q] >= 0 && lev[q] <= 30 ? (int64_t)1 << lev[q] : -1;
    if (sz < 0 || i0[q] < 0 || j0[q] < 0 || i0[q] + sz > nzp || j0[q] + sz > nrp) return -EINVAL;
  }
  const int64_t stride = (int64_t)nrp + 1, nkeys = ((int64_t)nzp + 1) * stride;
  const size_t nwords = (size_t)((nkeys + 63) >> 6);
  int rc_ = -ENOMEM;
  hfh_mesh* h = (hfh_mesh*)calloc(1, sizeof(hfh_mesh));
  uint64_t* corner = (uint64_t*)calloc(nwords, sizeof(uint64_t));
  uint64_t* node = (uint64_t*)malloc(nwords * sizeof(uint64_t));
  uint32_t* rank = (uint32_t*)malloc((nwords + 1) * sizeof(uint32_t));
  uint8_t* flags = (uint8_t*)malloc((size_t)nleaf);
  int32_t* ntri = (int32_t*)malloc(sizeof(int32_t) * (size_t)nleaf);
  uint32_t *ncode = NULL, *lcode = NULL;
  int32_t *npay = NULL, *perm = NULL, *lorder = NULL;
  int64_t* toff = NULL;
  mesh_arg m;
  memset(&m, 0, sizeof m);
  if (!h || !corner || !node || !rank || !flags || !ntri) goto done;
  m.nleaf = nleaf; m.i0 = i0; m.j0 = j0; m.lev = lev; m.stride = stride; m.corner = corner; m.node = node; m.flags = flags; m.ntri = ntri;
  m.mat = mat; m.nrp = nrp;
  const size_t big = (size_t)1 << 22;      /* par_rows: enough work to pay for threads */
  par_rows((int32_t)nleaf, nleaf >= 200000 ? big : 0, mesh_corner_rows, &m);
  memcpy(node, corner, nwords * sizeof(uint64_t));
  par_rows((int32_t)nleaf, nleaf >= 200000 ? big : 0, mesh_flag_rows, &m);
  /* ranks of the node bits (row-major), then the nodes in Morton order */
  int64_t nn = 0;
  for (size_t w = 0; w < nwords; ++w) { rank[w] = (uint32_t)nn; nn += __builtin_popcountll(node[w]); }
  rank[nwords] = (uint32_t)nn;
  if (nn > 0x7fffffffLL) { rc_ = -EOVERFLOW; goto done; }
  ncode = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nn);
  npay = (int32_t*)malloc(sizeof(int32_t) * (size_t)nn);
  perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)nn);
  h->coords = (double*)malloc(sizeof(double) * 2 * (size_t)nn);
  h->node_ij = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)nn);
  if (!ncode || !npay || !perm || !h->coords || !h->node_ij) goto done;
  {
    int64_t r = 0;
    for (size_t w = 0; w < nwords; ++w) {
      uint64_t bits = node[w];
      while (bits) {
        const int64_t key = (int64_t)(w << 6) + __builtin_ctzll(bits);
        bits &= bits - 1;
        ncode[r] = morton16(key / stride, key % stride);
        npay[r] = (int32_t)r;
        ++r;
      }
    }
  }
  if ((rc_ = radix_sort_u32(nn, ncode, npay)) != 0) goto done;
  for (int64_t id = 0; id < nn; ++id) perm[npay[id]] = (int32_t)id;
  {
    int64_t r = 0;
    for (size_t w = 0; w < nwords; ++w) {
      uint64_t bits = node[w];
      while (bits) {
        const int64_t key = (int64_t)(w << 6) + __builtin_ctzll(bits);
        bits &= bits - 1;
        const int64_t ki = key / stride, kj = key % stride;
        const int32_t id = perm[r++];
        h->coords[2 * (size_t)id] = zc[ki < nz ? ki : nz];
        h->coords[2 * (size_t)id + 1] = rc[kj < nr ? kj : nr];
        h->node_ij[2 * (size_t)id] = ki;
        h->node_ij[2 * (size_t)id + 1] = kj;
      }
    }
  }
  /* leaves in the Morton order of their centres, triangle offsets */
  lcode = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nleaf);
  lorder = (int32_t*)malloc(sizeof(int32_t) * (size_t)nleaf);
  toff = (int64_t*)malloc(sizeof(int64_t) * ((size_t)nleaf + 1));
  if (!lcode || !lorder || !toff) { rc_ = -ENOMEM; goto done; }
  for (int64_t q = 0; q < nleaf; ++q) {
    const int64_t hh = ((int64_t)1 << lev[q]) >> 1;
    lcode[q] = morton16(i0[q] + hh, j0[q] + hh);
    lorder[q] = (int32_t)q;
  }
  if ((rc_ = radix_sort_u32(nleaf, lcode, lorder)) != 0) goto done;
  int64_t nt = 0, nfan = 0;
  for (int64_t pos = 0; pos < nleaf; ++pos) { toff[pos] = nt; nt += ntri[lorder[pos]]; nfan += flags[lorder[pos]] != 0; }
  toff[nleaf] = nt;
  if (nt > 0x7fffffffLL / 3) { rc_ = -EOVERFLOW; goto done; }
  h->tris = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nt);
  h->tags = (int32_t*)malloc(sizeof(int32_t) * (size_t)nt);
  if (!h->tris || !h->tags) { rc_ = -ENOMEM; goto done; }
  m.rank = rank; m.perm = perm; m.lorder = lorder; m.toff = toff; m.coords = h->coords; m.tris = h->tris; m.tags = h->tags;
  par_rows((int32_t)nleaf, nleaf >= 200000 ? big : 0, mesh_tri_rows, &m);
  if (m.degenerate) { rc_ = -EDOM; goto done; }
  h->n_nodes = nn; h->n_tris = nt; h->n_fan = nfan;
  rc_ = 0;
done:
  free(corner); free(node); free(rank); free(flags); free(ntri); free(ncode); free(npay); free(perm); free(lcode); free(lorder); free(toff);
  if (rc_ != 0) { hfh_mesh_free(h); return rc_; }
  *out = h;
  return 0;
}

int hfh_mesh_sizes(const hfh_mesh* h, int64_t* n_nodes, int64_t* n_tris, int64_t* n_fan) {
  if (!h || !n_nodes || !n_tris || !n_fan) return -EINVAL;
  *n_nodes = h->n_nodes; *n_tris = h->n_tris; *n_fan = h->n_fan;
  return 0;
}

int hfh_mesh_fetch(const hfh_mesh* h, double* coords, int64_t* node_ij, int32_t* tris, int32_t* tags) {
  if (!h || !coords || !node_ij || !tris || !tags) return -EINVAL;
  memcpy(coords, h->coords, sizeof(double) * 2 * (size_t)h->n_nodes);
  memcpy(node_ij, h->node_ij, sizeof(int64_t) * 2 * (size_t)h->n_nodes);
  memcpy(tris, h->tris, sizeof(int32_t) * 3 * (size_t)h->n_tris);
  memcpy(tags, h->tags, sizeof(int32_t) * (size_t)h->n_tris);
  return 0;
}
