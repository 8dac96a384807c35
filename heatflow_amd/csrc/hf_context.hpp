// hf_context.hpp - constants, the context struct and small host helpers of libheatflow_hip.so
// (HIP/CDNA4 gfx950 implementation of include/heatflow_hip.h).
//

#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "amg_host.hpp"
#include "heatflow_hip.h"

namespace {

constexpr int RB = 256;        // rows per chunk of the vector kernels
#ifndef HF_RBA
#define HF_RBA 256
#endif
constexpr int RBA = HF_RBA;    // CSR rows owned by one assembly workgroup (= its thread count); 512 measured 7 % slower
constexpr int TPB = 256;       // threads per workgroup = 4 wavefronts of 64
constexpr int NCOL = 32;       // max colours per row block (uint32 mask)
#ifndef HF_STAGE_U
#define HF_STAGE_U 2   // column-list entries per lane and pass while a chunk's operand slice is staged (k_spmv, C16)
#endif
#ifndef HF_UNROLL
#define HF_UNROLL 4
#endif
constexpr int MAXP = 1024;     // max workgroups per launch = partial-sum slots per array
constexpr int MAXRESP = 4;     // boundary-response directions kept per operator
// projection start vector (k_proj_* in hf_kernels.hpp): solutions kept in the ring, + boundary responses = vectors in all
#ifndef HF_PROJ_MH
#define HF_PROJ_MH 6
#endif
constexpr int PROJ_MH = HF_PROJ_MH;
constexpr int PROJ_MT = PROJ_MH + MAXRESP;
constexpr int TS = 512;        // SpMV workgroup: 512 threads = 8 wavefronts own 512 consecutive rows per chunk
// rows per chunk of the fine operator's LDS-staged SpMV (a workgroup of TS threads, one row per lane in the row phase;
// fewer rows than TS: more, shorter chunks per workgroup for the chunk pipeline) and its stream entries per lane in flight
#ifndef HF_SPMV_RPC
#define HF_SPMV_RPC 512
#endif
#ifndef HF_SPMV_UN
#define HF_SPMV_UN 8
#endif
constexpr int SRPC = HF_SPMV_RPC;
                               // (4 such workgroups per CU = 32 waves/CU; measured 20 % faster than 256x16)

// Host-visible progress of the running solve (pinned, mapped memory): written by the one thread that changes the
// device-resident scalars, so the host follows the convergence tests by reading its own memory - no copy, no
// synchronisation - and queues the next iteration while the current V-cycle is still running.
// `done` and `tested` carry the epoch of the solve that wrote them in their upper half (Scal::epoch, bumped by the host for
// every solve): launches of an earlier solve that are still queued when the host has moved on (the blind first burst can
// run past convergence) cannot be mistaken for progress of the current one, and the host never has to reset the mirror
// while something that writes it may still be in flight.
struct ScalMirror {
  double zz, bn2;
  int iters;                   // updates done when the last test ran
  int pad_;
  unsigned long long done_st;    // (epoch << 32) | Scal::done
  unsigned long long tested_st;  // (epoch << 32) | (iteration count whose iterate was tested last + 1); the start kernel writes + 1 = "0 tested"
};
inline int mirror_tested(const ScalMirror* m, unsigned epoch) {   // host side: -1 until the current solve's start kernel has run
  const unsigned long long v = __atomic_load_n(&m->tested_st, __ATOMIC_ACQUIRE);
  return static_cast<unsigned>(v >> 32) == epoch ? static_cast<int>(v & 0xffffffffu) - 1 : -1;
}
inline int mirror_done(const ScalMirror* m, unsigned epoch) {
  const unsigned long long v = __atomic_load_n(&m->done_st, __ATOMIC_ACQUIRE);
  return static_cast<unsigned>(v >> 32) == epoch ? static_cast<int>(v & 0xffffffffu) : 0;
}

struct Scal {                  // device-resident PCG scalars
  double tol2;                 // (max(rtol*||D^-1 b||, atol))^2
  double bn2;                  // ||D^-1 b||^2
  double zz;                   // ||D^-1 r||^2 of the last iterate
  int iters;
  int done;                    // 0 running, 1 converged, 2 breakdown
  int first;                   // 1 until the first update of a solve: the first direction is p = z (beta = 0)
  unsigned epoch;              // the host's count of solves on this context (k_pcg_begin / kb_begin): stamps what goes to the mirror
  ScalMirror* mirror;          // device address of the host mirror, or null (batched columns)
};

// Per-column scalars of the batched PCG (hf_batch.hpp), each reduced once per producer by a one-workgroup kernel
constexpr int NV_MAX = 16;    // columns of the widest batch
struct BRed { double pAp[NV_MAX], rz[2][NV_MAX], zz[NV_MAX], bn[NV_MAX]; };

}  // namespace

struct hf_ctx {
  int dev = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  double last_ms = 0.0;

  int32_t n = 0, ne = 0, nbc = 0;
  int64_t nnz = 0;
  int nchunks = 0, P = 0;      // 256-row chunks and grid of the vector kernels / assembly
  int nchunks_s = 0, Ps = 0;   // 512-row chunks and grid of the SpMV kernel (Ps <= P partials)
  int max_chunk_nnz_s = 0;
  // compressed column indices of the fine operator (ColComp in hf_kernels.hpp): per SpMV chunk a sorted column list,
  // per nonzero a 16-bit position in it
  int32_t *d_cdict_ptr = nullptr, *d_cdict = nullptr;
  uint16_t* d_cid = nullptr;
  int max_cdict = 0;
  bool cdict_own = false;      // every chunk's own rows sit contiguously in its column list (each row stores its diagonal): the kernels take x[row] from the staged slice
  bool c16 = true;             // HEATFLOW_SPMV_C16=0 keeps the 32-bit column stream
  bool have_mesh = false, have_mat = false, assembled = false;
  double dt = 0.0;
  int mode = 0;

  // host copies of the pattern (needed to build lifting structures and the multigrid hierarchy) and of the
  // elements (the lists of the LDS scatter kernels are built from them on first use)
  std::vector<int32_t> h_rowptr, h_colidx, h_tri, h_tag;
  std::vector<char> h_tag_used;
  bool owner_ready = false;

  // device: mesh
  double2* d_zr = nullptr;
  int4* d_elem = nullptr;
  int tab_len = 0;
  double *d_kappa = nullptr, *d_rhoc = nullptr;
  // device: pattern + owner lists
  int32_t *d_rowptr = nullptr, *d_colidx = nullptr;
  int32_t *d_blk_eptr = nullptr, *d_blk_cptr = nullptr;
  int2* d_blk_ent = nullptr;     // 3 x int2 per owner-list entry
  int nblk_a = 0;
  int max_blk_nnz = 0, ncolors = 0;
  int64_t elist_len = 0;
  // device: row-gather assembly lists (RowGather in hf_pattern.hpp) and coefficient tables by tag-dictionary index
  bool rg_ok = false;
  int4* d_rg_hdr = nullptr;
  uint16_t *d_rg_ell = nullptr, *d_rg_cid = nullptr;
  int32_t* d_rg_dict = nullptr;     // the blocks' column lists (global node ids)
  double2* d_rg_zrb = nullptr;      // coordinates of every block's column list (own rows + halo)
  int rg_max_dict = 0, rg_grid = 0;
  int64_t n_rg_ell = 0, n_rg_dict = 0, n_cdict = 0;
  std::vector<int32_t> h_rg_tags;
  double *d_kappa_rg = nullptr, *d_rhoc_rg = nullptr;
  // device: matrices
  double *d_M = nullptr, *d_A = nullptr, *d_dinv = nullptr;
  // device: Dirichlet
  int32_t* d_bc_dofs = nullptr;
  double* d_g = nullptr;
  int32_t nlift_rows = 0, nlift = 0;
  int32_t *d_lift_rows = nullptr, *d_lift_ptr = nullptr, *d_lift_bc = nullptr, *d_lift_slot = nullptr;
  double* d_lift_val = nullptr;
  // device: vectors
  double *d_u = nullptr, *d_b = nullptr, *d_r = nullptr, *d_p = nullptr, *d_Ap = nullptr;
  double *d_uprev = nullptr, *d_ustart = nullptr;   // u^{n-1} and the buffer of the next start vector (rotated with d_u)
  bool have_prev = false;
  int extrapolate = 1;         // start PCG from 2 u^n - u^{n-1} (same answer, fewer iterations)
  // hf_set_start_vector kind 2: boundary-response correction of the start vector.  Host copies of the last two
  // boundary vectors, an orthonormal set of directions seen in their second difference and, per direction d,
  // the device vector w = R d (A_hat w = -lift(d), w_B = d).
  int start_kind = 3;
  std::vector<double> h_g0, h_g1;   // g^n, g^{n-1}
  int g_hist = 0;
  struct BcResponse { std::vector<double> dir; double* w = nullptr; };
  std::vector<BcResponse> resp;
  long long resp_solves = 0;
  // hf_set_start_vector kind 3: Galerkin projection on the last solutions and the boundary responses (k_proj_*)
  struct Proj {
    double *V[PROJ_MT] = {nullptr}, *F[PROJ_MT] = {nullptr};   // slots 0..PROJ_MH-1: ring of (solution with zeroed Dirichlet entries, its right-hand side); then the responses
    bool used[PROJ_MT] = {false};
    int next = 0, pending = -1;   // ring slot to overwrite next; slot whose Gram column is still to be computed
    double *G = nullptr, *alpha = nullptr, *part = nullptr;
    bool ready = false;
  } proj;
  double *d_tmp = nullptr;
  // device: reductions
  double *d_part_pAp = nullptr, *d_part_rz = nullptr, *d_part_zz = nullptr, *d_part_bn = nullptr;
  Scal* d_scal = nullptr;
  Scal* h_scal = nullptr;      // pinned
  ScalMirror* h_mirror = nullptr;   // pinned + mapped; d_mirror is its device address
  ScalMirror* d_mirror = nullptr;
  unsigned epoch = 0;          // solves started on this context (Scal::epoch)
  int32_t* d_samp_idx = nullptr;
  double* d_samp = nullptr;
  int samp_cap = 0;
  int pred_iters = 0;
  // multigrid preconditioner (hf_set_precond): device hierarchy
  int precond = 0;             // 0 Jacobi, 1 smoothed-aggregation AMG V(1,1)
  int amg_reuse = 0;           // 1: keep the coarse levels across hf_assemble calls (kappa sweeps)
  bool amg_ready = false;
  struct DevCsr {
    int nrow = 0, ncol = 0, lanes = 8; int64_t nnz = 0; int32_t *ptr = nullptr, *idx = nullptr; double* val = nullptr;
    float* valf = nullptr;                     // single-precision values (transfer operators of the preconditioner); val is then null
    int max_row = 0;
    int rpc = 0, nchunks = 0, chunk_nnz = 0;   // LDS-staged (stream) kernel geometry; rpc = 0 -> use the sub-wave kernel
    int32_t *dptr = nullptr, *dict = nullptr;  // stream kernel with compressed columns: per-chunk column lists ...
    uint16_t* cid = nullptr;                   // ... and a 16-bit position per nonzero
    int max_dict = 0;
    int64_t ndict = 0;                         // entries of `dict` (all chunks)
  };
  // Levels 1..nl-2 run the cycle through the fused legs Rt / GP (amg_host.hpp): `cat` = [b_l ; result of level l+1]
  // is GP's operand, `b` aliases its head; a level's result goes to `res` (the tail of the finer level's cat,
  // or the level's own x on level 1).  The coarsest level owns a zero-padded b (the dense solve reads pairs).
  struct DevLevel {
    DevCsr A, P, R, Rt, GP;
    double *dinv = nullptr, *x = nullptr, *cat = nullptr, *b = nullptr, *res = nullptr;
    bool own_b = false;
    double omega = 0; int n = 0;
  };
  std::vector<DevLevel> amg;
  double* d_coarse_inv = nullptr;
  float* d_coarse_inv_f = nullptr;   // the same in single precision (amg_f32)
  int coarse_n = 0, coarse_ld = 0;   // dense inverse, row-major, leading dimension a multiple of 4 (16-byte row loads)
  bool amg_fine_stale = false;       // frozen hierarchy (reuse) and the fine operator re-valued since it was built: the fused down leg Rt_0 no longer matches A
  int amg_fuse0 = 0;                 // finest level of the V-cycle: 0 explicit sweeps, 1 fused legs Rt_0 / GP_0, 2 fused down leg only (chosen by size in build_amg; HEATFLOW_AMG_FUSE0 overrides)
  bool amg_f32 = true;               // operators of the preconditioner below the fine level stored in float (HEATFLOW_AMG_F32=0: double)
  double amg_opc = 0.0, amg_setup_s = 0.0;
  // what the fine operator the hierarchy was built from depends on besides the mesh (hf_amg_io.hpp): time step, coefficient
  // tables, Dirichlet set - compared with the context's own operator whenever a kept or installed hierarchy meets a new hf_assemble
  struct OperatorPrint { double dt = 0.0; std::vector<double> kappa, rhoc; int32_t nbc = 0; uint64_t bc_hash = 0; } amg_print;
  long long amg_fallbacks = 0;   // steps finished by Jacobi-PCG after a multigrid-PCG breakdown
  double *d_z = nullptr, *d_z2 = nullptr;
  // read-flux projection (hf_flux_setup): unit-rho_c r-weighted mass matrix and the projected gradient
  bool flux_ready = false;
  int flux_valid = 0;          // components (bit 0 z, bit 1 r) the last projection solved
  double *d_M1 = nullptr, *d_dinv1 = nullptr, *d_gz = nullptr, *d_gr = nullptr, *d_bz = nullptr, *d_br = nullptr;
  int pred_flux[2] = {0, 0};
  // batched time loop (hf_batch_*, hf_batch.hpp): NV sweep points as interleaved columns
  struct BatchLevel { double *x = nullptr, *cat = nullptr, *b = nullptr, *res = nullptr; bool own_b = false; };
  struct Batch {
    int nv = 0;                  // 0: no batch open
    int opk = 0;                 // fine operator of the columns: 0 shared (ctx->d_A), 1 one per column, 2 affine A + d_j A1
    const double *sysA = nullptr, *sysDinv = nullptr;   // shared operator other than ctx->d_A (the flux projection's mass matrix)
    double *A = nullptr, *dinv = nullptr, *lift_val = nullptr;            // per-column operator data (opk 1; dinv also opk 2)
    double *A1 = nullptr, *lift1 = nullptr;                               // affine part and its lifting values (opk 2)
    double delta[NV_MAX] = {};
    double *g = nullptr;                                                   // boundary values of all steps
    double *u = nullptr, *uprev = nullptr, *ustart = nullptr, *b = nullptr, *r = nullptr, *p = nullptr, *Ap = nullptr;
    double *z = nullptr, *z2 = nullptr, *tmp = nullptr;
    double *part_pAp = nullptr, *part_rz = nullptr, *part_zz = nullptr, *part_bn = nullptr;
    Scal *scal = nullptr, *h_scal = nullptr;
    ScalMirror *h_mirror = nullptr, *d_mirror = nullptr;   // per-column progress for the host (pinned + mapped), hf_batch_begin only
    unsigned epoch = 0;          // solves started on this batch state (Scal::epoch of every column)
    BRed* red = nullptr;         // per-column reduced scalars
    std::vector<BatchLevel> lev;
    int Pb = 0, pred_iters = 0;
    bool have_prev = false;
    // projection start vector per column: ring of (solutions with zeroed Dirichlet entries, right-hand sides)
    double *pV[PROJ_MH] = {nullptr}, *pF[PROJ_MH] = {nullptr}, *pG = nullptr, *palpha = nullptr, *ppart = nullptr;
    bool pused[PROJ_MH] = {false};
    int pnext = 0, ppending = -1;
    unsigned loaded = 0;         // bit j: column j's operator has been loaded (percol)
    bool lds = false;            // fine-pattern products through kb_spmv_lds (the context's `bcols` tables are for this nv)
  } batch;
  // compressed columns of the batched loop's LDS-staged SpMV (BComp in hf_batch.hpp): per chunk of `rpc` rows a sorted
  // column list and per nonzero a 16-bit position in it; built on the first hf_batch_begin with a given nv, kept per mesh
  struct BatchCols {
    int nv = 0, rpc = 0, nchunks = 0, cap_nnz = 0, cap_dict = 0;
    int32_t *ptr = nullptr, *dict = nullptr, *own = nullptr;
    uint16_t* id = nullptr;
  } bcols;
  Batch fluxb;                   // two-column state of the read-flux projection (both components in one PCG), swapped into `batch` while it runs
  Batch fluxnb;                  // nv-column state of the batched loop's read-flux projection (hf_batch_run_flux): one gradient component of every column per PCG
  int32_t* d_fsamp_idx = nullptr;   // its sample nodes
  int fsamp_cap = 0;
  // optional in-situ kernel timing (hf_set_profile): event pairs around each PCG SpMV launch
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;
  int prof_used = 0, prof_base = 0;
  double prof_spmv_ms = 0.0;
  long long prof_spmv_n = 0;
};

namespace {

int fail(hf_ctx* c, int code, const char* fmt, ...);
inline size_t pad16(size_t b) { return (b + 15) & ~static_cast<size_t>(15); }

// a polite spin: the polling threads of concurrent sessions share host cores with the threads that launch kernels
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

hipError_t copy_sync(hf_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (bytes == 0) return hipSuccess;
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}

int fail(hf_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HF_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(ctx, HF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <typename T>
int dev_alloc(hf_ctx* ctx, T** p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) return fail(ctx, HF_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
  return HF_OK;
}
#define HF_TRY(expr) do { int rc_ = (expr); if (rc_ != HF_OK) return rc_; } while (0)

// Host<->device copy that is complete on return, issued on the context's own stream (never the legacy
// stream: contexts driven from other host threads must not serialise on it).
hipError_t copy_sync(hf_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);

template <typename T>
void dev_free(T** p) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
}

// Scratch device buffer released on every exit path of the function that owns it.
template <typename T>
struct DevTemp {
  T* p = nullptr;
  ~DevTemp() { dev_free(&p); }
  DevTemp() = default;
  DevTemp(const DevTemp&) = delete;
  DevTemp& operator=(const DevTemp&) = delete;
};

}  // namespace
