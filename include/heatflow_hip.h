/*
 * heatflow_hip.h - C ABI of libheatflow_hip.so (MI355X / gfx950, HIP).
 *
 * Drop-in boundary for the hot path of cebarker1000/heatflow.  The reference has no
 * FFI layer: its drivers call dolfinx / PETSc directly from Python.  Each entry point
 * below names the reference call(s) it stands in for (file:line in the reference):
 *
 *   hf_set_mesh        gmshio.model_to_mesh(...)            run_with_diamond.py:240-245
 *                      fem.functionspace(domain, P1 / DG0)  run_with_diamond.py:279-280
 *   hf_set_mesh_prebuilt / hf_pattern_export   the same for the 2nd..Nth worker of a sweep, which in the
 *                      reference re-reads mesh.msh and rebuilds everything   parameter_sweep.py:401-446
 *   hf_set_materials   kappa.x.array[:] / rho_cv.x.array[:] run_with_diamond.py:286-301
 *   hf_set_dirichlet   fem.dirichletbc(g, row_dofs) x 4     dirichlet_bc/bc.py:104-113,
 *                                                           run_with_diamond.py:362-374
 *   hf_assemble        fem.form(lhs) + assemble_matrix(lhs_form, bcs) + KSP/PC setup
 *                                                           run_with_diamond.py:328-337, 381-394
 *   hf_set_state       u_n.x.array[:] = ic_temp             run_with_diamond.py:317-319
 *   hf_step            b.set(0); assemble_vector(b, rhs_form); apply_lifting; ghostUpdate;
 *                      set_bc; solver.solve(b, u_n)         run_with_diamond.py:474-481
 *   hf_sample          u_n.x.array[node_idx]                run_with_diamond.py:485-493
 *   hf_get_state       u_n.x.array (what xdmf.write_function would write)   :483-484
 *   hf_flux_setup      assemble_matrix(a_proj) + KSP/LU set-up       run_no_diamond.py:471-491
 *   hf_flux_project    assemble_vector(rhs_proj) + solver_proj.solve run_no_diamond.py:543-550
 *
 * Conventions
 *   - All functions return 0 (HF_OK) or a negative HF_ERR_* code; hf_last_error(ctx)
 *     returns a message for the last failure on that context.
 *   - Host pointers are borrowed for the duration of the call only and copied to the
 *     device; outputs go to caller-allocated host buffers.  No torch / numpy types.
 *   - float64 values, int32 indices.  Node coordinates are (z, r) pairs: mesh x = z
 *     (axial), mesh y = r (radial), weight r = x[1] as in run_with_diamond.py:321-322.
 *   - One ctx = one HIP device + one stream.  A ctx is not thread-safe; different
 *     ctxs may be driven from different threads or processes (one per GPU).
 *   - hf_step / hf_run / hf_batch_run block until their steps are done; meanwhile the calling
 *     thread polls pinned host memory that the solver kernels update (no other thread is
 *     created).  No progress for HEATFLOW_POLL_TIMEOUT_S seconds (default 60) -> HF_ERR_HIP.
 *   - There is no CPU fallback: without a HIP device hf_create fails.
 */
#ifndef HEATFLOW_HIP_H
#define HEATFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hf_ctx hf_ctx;

enum {
  HF_OK = 0,
  HF_ERR_ARG = -1,      /* bad argument (null pointer, index out of range, ...) */
  HF_ERR_STATE = -2,    /* call out of order (e.g. hf_step before hf_assemble) */
  HF_ERR_HIP = -3,      /* a HIP runtime call failed; see hf_last_error */
  HF_ERR_NOCONV = -4,   /* PCG hit max_it (or broke down) before reaching the tolerance */
  HF_ERR_ALLOC = -5
};

/* Assembly variants (all are per-element kernels, results agree to rounding):
 *   HF_ASM_LDS_ATOMIC    a workgroup owns 256 CSR rows, stages their value slab in LDS,
 *                        scatter-adds the incident elements with LDS f64 atomics and
 *                        streams the slab out once (coalesced).
 *   HF_ASM_LDS_COLORED   same staging, elements processed colour by colour with plain
 *                        LDS read-modify-write: bitwise reproducible.
 *   HF_ASM_GLOBAL_ATOMIC one thread per element, f64 atomics straight into global CSR
 *                        (baseline / cross-check).
 *   HF_ASM_ROW_GATHER    a lane owns one CSR row and visits the triangles at its node (16-bit
 *                        list entries = positions of the two other vertices in the row); the
 *                        slab, the column positions and the coordinates of the block live in
 *                        LDS, every global access is a coalesced stream; no atomics, no
 *                        colours, bitwise reproducible.  Default of the entry points.  Falls
 *                        back to HF_ASM_LDS_COLORED for meshes with a row of more than 32
 *                        entries or more than 64 distinct cell tags. */
enum { HF_ASM_LDS_ATOMIC = 0, HF_ASM_LDS_COLORED = 1, HF_ASM_GLOBAL_ATOMIC = 2, HF_ASM_ROW_GATHER = 3 };

/* Kernels addressable by hf_time_kernel */
enum {
  HF_K_SPMV = 0,        /* y = A x, CSR, LDS-staged products            */
  HF_K_PCG_SPMV = 1,    /* PCG iteration head: beta, Ap <- A z + beta Ap, p <- z + beta p, p.Ap partials */
  HF_K_PCG_UPDATE = 2,  /* x += a p; r -= a Ap; z = D^-1 r; r.z, z.z    */
  HF_K_PCG_DIR = 3,     /* retired: the direction update is fused into HF_K_PCG_SPMV (returns HF_ERR_ARG) */
  HF_K_ASSEMBLE = 4,    /* element kernel in the mode of the last hf_assemble */
  HF_K_RHS = 5,         /* b = M u^n                                    */
  HF_K_STREAM_READ = 6  /* plain streaming read of the operator's values + column indices (12 nnz bytes, 16-byte loads):
                           the read bandwidth this device reaches on the SpMV's own arrays - its practical ceiling */
};

const char* hf_version(void);

int hf_create(int device_id, hf_ctx** out);
int hf_destroy(hf_ctx* ctx);
const char* hf_last_error(const hf_ctx* ctx);

/* Mesh: n nodes, n_e P1 triangles.  zr = n x 2 (z, r); tri = n_e x 3 node ids;
 * tag = n_e cell tags (>= 0).  Builds the CSR sparsity pattern and the row-block
 * element lists once (host side) and uploads everything. */
int hf_set_mesh(hf_ctx* ctx, int32_t n, int32_t n_e, const double* zr, const int32_t* tri, const int32_t* tag);

/* The tables hf_set_mesh derives from the connectivity (CSR pattern, compressed column lists of the SpMV chunks,
 * row-gather assembly lists) in serialised form, so that ONE context builds them and every other context on the
 * same mesh - other ranks of a sweep, further contexts of the same rank - installs them without rebuilding:
 * the MI355X counterpart of the reference's workers each re-reading mesh.msh (parameter_sweep.py:401-446).
 * hf_pattern_export writes exactly hf_pattern_export_size bytes; hf_set_mesh_prebuilt = hf_set_mesh with the
 * tables taken from such a blob (validated against n, n_e and its own index ranges; the mesh arrays themselves
 * are trusted to be the ones the blob was exported for).  `blob` may be a host or a device pointer in both calls
 * (hipMemcpyDefault), so an RCCL broadcast buffer can be handed over as it is. */
int hf_pattern_export_size(hf_ctx* ctx, int64_t* bytes);
int hf_pattern_export(hf_ctx* ctx, void* blob, int64_t bytes);
int hf_set_mesh_prebuilt(hf_ctx* ctx, int32_t n, int32_t n_e, const double* zr, const int32_t* tri, const int32_t* tag,
                         const void* blob, int64_t bytes);

/* Cell-tag -> coefficient tables: kappa[c] = kappa[i], rho_c[c] = rho_c[i] for cells with
 * tag == tags[i].  May be called again (kappa sweep) followed by hf_assemble. */
int hf_set_materials(hf_ctx* ctx, int32_t n_mat, const int32_t* tags, const double* kappa, const double* rho_c);

/* Kappa sweep step: overwrite the conductivity of the listed cell tags (rho_c, mesh, pattern, Dirichlet
 * set, dt and assembly mode stay) and re-value M, A, D^-1 - i.e. hf_set_materials + hf_assemble for the
 * entries that changed (reference: a new run_simulation per kappa, sweep_test.py:55-75). */
int hf_update_kappa(hf_ctx* ctx, int32_t n_mat, const int32_t* tags, const double* kappa);

/* Dirichlet DOFs (unique; the host resolves overlaps "later BC wins" beforehand).
 * The order defines the order of g_bc in hf_step.  n_bc = 0 removes all BCs. */
int hf_set_dirichlet(hf_ctx* ctx, int32_t n_bc, const int32_t* dofs);

/* M = M_r(rho_c), A = M + dt K_r(kappa); then rows+columns of the Dirichlet DOFs are
 * zeroed with unit diagonal (the lifting columns are kept aside) and D^-1 is formed. */
int hf_assemble(hf_ctx* ctx, double dt, int32_t mode);

/* Preconditioner of the PCG solve: kind 0 = Jacobi (D^-1, the north-star path), kind 1 =
 * smoothed-aggregation multigrid V(1,1) with damped-Jacobi smoothing, built on the host from the
 * assembled operator at hf_assemble time and applied on the GPU with CSR SpMV kernels.  With
 * reuse != 0 the coarse levels are kept across later hf_assemble calls (kappa sweeps on one mesh:
 * the fine level always uses the current matrix, the frozen coarse levels remain a valid SPD
 * preconditioner).  Call before hf_assemble.  The stopping criterion of hf_step is the same for both. */
int hf_set_precond(hf_ctx* ctx, int32_t kind, int32_t reuse);
/* Hierarchy of the last AMG set-up: level count, rows per level (up to max_levels entries),
 * operator complexity sum(nnz_l)/nnz_0 and host set-up time.  Any pointer may be NULL. */
int hf_get_amg_info(hf_ctx* ctx, int32_t* n_levels, int32_t* level_rows, int32_t max_levels, double* op_complexity,
                    double* setup_seconds);

/* The hierarchy of a context as one blob, for other contexts on the same mesh (the other sessions of a sweep, on this
 * GPU or - broadcast with the mesh - on the other ranks): what the host set-up computes is shipped instead of recomputed.
 * The reference's analogue is the per-worker MUMPS factorisation (run_with_diamond.py:389-394; every pool worker of
 * parameter_sweep.py:401-446 factorises for itself).  hf_amg_export needs a completed hf_assemble with the multigrid
 * preconditioner; blob = host or device memory of hf_amg_export_size bytes.  hf_amg_install needs hf_set_mesh,
 * hf_set_dirichlet and hf_set_precond(1, reuse = 1) on the same mesh and is followed by hf_assemble, which keeps the
 * installed hierarchy instead of building one and compares its own operator with the fingerprint in the blob (time step,
 * coefficient tables, Dirichlet set): the same operator -> the cycle is the one the exporting context runs, bit for bit;
 * another point of a sweep -> the hierarchy is a frozen one (see hf_set_precond).  Every index in the blob is
 * verified; HF_ERR_ARG if it does not belong to this mesh. */
int hf_amg_export_size(hf_ctx* ctx, int64_t* bytes);
int hf_amg_export(hf_ctx* ctx, void* blob, int64_t bytes);
int hf_amg_install(hf_ctx* ctx, const void* blob, int64_t bytes);

/* Start vector of every hf_step / hf_run solve (the converged answer does not depend on it, only the
 * iteration count does): kind 0 = u^n (what KSP.solve sees in the reference, run_with_diamond.py:480,
 * where it is irrelevant because the solve is direct); 1 = 2 u^n - u^{n-1}; 2 = that plus the
 * response to the second difference of the boundary values: the loop is linear,
 * u^{n+1} = T u^n + R g^{n+1}, so  u^{n+1} - 2u^n + u^{n-1} = T(...) + R (g^{n+1} - 2g^n + g^{n-1});  R d is
 * obtained by one extra solve the first time a new direction d of that second difference appears (the
 * heated line's Gaussian profile: once per assembled operator) and re-used, scaled, afterwards;
 * 3 (default) = Galerkin projection: the combination of the last six solutions and of those boundary responses
 * that is closest to the new solution in the A-norm (each of them solves A v = f with a known f, so the normal
 * equations cost one pass over the stored vectors; Fischer 1998).  It contains kinds 1 and 2 as special
 * combinations and needs fewer iterations than either (13.1 -> ~10.5 per step on the 1M-DOF mesh). */
int hf_set_start_vector(hf_ctx* ctx, int32_t kind);
/* Number of extra response solves spent so far (diagnostics). */
int hf_get_response_solves(hf_ctx* ctx, int64_t* count);

/* Number of hf_step solves that hit a breakdown (p.Ap <= 0) in the multigrid-preconditioned loop and
 * were finished with the Jacobi preconditioner instead (still on the GPU).  0 in every case tested. */
int hf_get_amg_fallbacks(hf_ctx* ctx, int64_t* count);

int hf_set_state(hf_ctx* ctx, const double* u);
int hf_get_state(hf_ctx* ctx, double* u);
int hf_sample(hf_ctx* ctx, int32_t n_s, const int32_t* nodes, double* out);

/* One backward-Euler step: b = M u^n - A[:,B] g, b_B = g, solve A_hat u^{n+1} = b by
 * Jacobi-PCG started from u^n (with u_B = g), in place.  Stops when
 * ||D^-1 r||_2 <= max(rtol * ||D^-1 b||_2, atol)  (a zero right-hand side - the answer is then zero - is measured
 * against the start residual instead).  iters / resid (relative) may be NULL. */
int hf_step(hf_ctx* ctx, const double* g_bc, double rtol, double atol, int32_t max_it, int32_t* iters, double* resid);

/* n_steps steps in one call: g_bc_all = n_steps x n_bc; after every step the n_s nodes
 * are sampled into samples (n_steps x n_s).  iters = n_steps entries (may be NULL). */
int hf_run(hf_ctx* ctx, int32_t n_steps, const double* g_bc_all, double rtol, double atol, int32_t max_it,
           int32_t n_s, const int32_t* nodes, double* samples, int32_t* iters);

/* Batched time loop: nv = 2, 4, 8 or 16 sweep points of ONE mesh, Dirichlet set and rho_c advance together as the
 * columns of a multi-vector PCG (reference: the independent runs of the parameter grid, parameter_sweep.py:195-235,
 * and of the kappa list, sweep_test.py:47-52).  Vectors are stored interleaved on the device, every index and every
 * shared matrix value is read once for nv products, and each column keeps its own alpha / beta / tolerance /
 * iteration count / done flag; per column the arithmetic and the stopping rule are hf_step's.
 *   HF_BATCH_SHARED      all columns share the context's assembled operator (points that differ in their boundary
 *                        values only: fwhm, heating curve)
 *   HF_BATCH_PER_COLUMN  every column has its own A_hat: assemble a point's operator in the context as usual
 *                        (hf_update_kappa), then hf_batch_load_column(j) copies A_hat, D^-1 and the lifting values
 *                        into column j
 *   HF_BATCH_AFFINE      A_hat_j = A_hat + delta_j * A1 with A1 = dt K restricted to the listed materials at unit
 *                        conductivity (hf_batch_set_affine): a sweep over ONE conductivity (sweep_test.py's kappa_sample
 *                        list) needs two shared value arrays and a scalar per column instead of nv operators;
 *                        delta_j = kappa_j - the conductivity the context's operator was assembled with
 * With per-column or affine operators the multigrid hierarchy is the frozen one (hf_set_precond(1, reuse = 1)), shared
 * by all columns.
 * hf_batch_begin needs a completed hf_assemble; hf_set_mesh / hf_set_dirichlet / hf_set_precond close the batch.
 * hf_batch_run: g_bc_all = n_steps x n_bc x nv ([step][bc][column]); samples = n_steps x nv x n_s; iters =
 * n_steps x nv.  The start vector of every solve is 2 u^n - u^{n-1}.  HF_ERR_NOCONV if any column fails. */
enum { HF_BATCH_SHARED = 0, HF_BATCH_PER_COLUMN = 1, HF_BATCH_AFFINE = 2 };
int hf_batch_begin(hf_ctx* ctx, int32_t nv, int32_t operator_kind);
int hf_batch_load_column(hf_ctx* ctx, int32_t j);
int hf_batch_set_affine(hf_ctx* ctx, int32_t n_tags, const int32_t* tags, const double* delta /* nv */);
int hf_batch_set_state(hf_ctx* ctx, int32_t j, const double* u);
int hf_batch_get_state(hf_ctx* ctx, int32_t j, double* u);
int hf_batch_run(hf_ctx* ctx, int32_t n_steps, const double* g_bc_all, double rtol, double atol, int32_t max_it,
                 int32_t n_s, const int32_t* nodes, double* samples, int32_t* iters);
/* hf_batch_run with run_no_diamond's per-step read-flux projection for every column (reference run_no_diamond.py:543-566,
 * which the sweep of parameter_sweep.py:43,157-166 runs at every grid point): after each step the gradient of each column's
 * new state is L2-projected with the r-weighted unit mass matrix (hf_flux_setup first) - per wanted component (bit 0 = z,
 * bit 1 = r; the reference's outputs read d/dr only) the nv columns are the interleaved columns of ONE Jacobi-PCG,
 * warm-started from the previous step's projection, stopping rule of hf_step with flux_rtol - and sampled at n_fs nodes:
 * flux_samples = n_steps x n_comp x nv x n_fs ([step][component, z before r][column][node]); flux_iters = n_steps x n_comp
 * (largest count among the columns; may be NULL).  flux_components = 0 is hf_batch_run. */
int hf_batch_run_flux(hf_ctx* ctx, int32_t n_steps, const double* g_bc_all, double rtol, double atol, int32_t max_it,
                      int32_t n_s, const int32_t* nodes, double* samples, int32_t* iters, int32_t flux_components,
                      double flux_rtol, int32_t flux_max_it, int32_t n_fs, const int32_t* flux_nodes, double* flux_samples,
                      int32_t* flux_iters);
int hf_batch_end(hf_ctx* ctx);

/* Read-flux projection of run_no_diamond (reference run_no_diamond.py:471-491 set-up, :543-550 per
 * step): grad_smooth = L2 projection of grad(T) onto vector P1 with weight r.  hf_flux_setup
 * assembles the unit-coefficient r-weighted mass matrix on the mesh's pattern (once per mesh);
 * hf_flux_project projects the CURRENT state: the 2n x 2n system of the reference is block diagonal, so the
 * components decouple exactly into two scalar solves with M_r(1); when both are wanted they run as the two
 * interleaved columns of ONE Jacobi-PCG (every pass over the matrix serves both; per column the stopping rule
 * of hf_step), a single wanted component is solved on its own; results copied to grad_z / grad_r (n values each).  A NULL output skips that component's
 * solve altogether (run_no_diamond's outputs only use d/dr, :553-566).  iters = 2 entries (may be NULL).
 * hf_flux_solve does the same without any copy (components: bit 0 = z, bit 1 = r); hf_flux_sample then
 * reads the projected gradient at n_s nodes (either output may be NULL) - what the band / axis averages
 * of run_no_diamond.py:494-513, 553-566 need, instead of two n-vectors per step. */
int hf_flux_setup(hf_ctx* ctx);
int hf_flux_project(hf_ctx* ctx, double rtol, int32_t max_it, double* grad_z, double* grad_r, int32_t* iters);
int hf_flux_solve(hf_ctx* ctx, int32_t components, double rtol, int32_t max_it, int32_t* iters);
int hf_flux_sample(hf_ctx* ctx, int32_t n_s, const int32_t* nodes, double* grad_z, double* grad_r);

int hf_get_sizes(hf_ctx* ctx, int32_t* n, int32_t* n_e, int64_t* nnz, int32_t* n_bc);
/* Any pointer may be NULL.  A is the matrix as it stands (eliminated when BCs are set). */
int hf_get_csr(hf_ctx* ctx, int32_t* rowptr, int32_t* colidx, double* A, double* M);
/* y = A x (which = 0) or y = M x (which = 1) through the SpMV kernel; host vectors. */
int hf_spmv(hf_ctx* ctx, int32_t which, const double* x, double* y);

/* Average duration (ms) of `reps` back-to-back launches of one kernel on the ctx stream,
 * bracketed by HIP events on that stream. */
int hf_time_kernel(hf_ctx* ctx, int32_t which, int32_t reps, double* ms_avg);
/* In-situ timing of the dominant kernel: while on, up to 64 PCG SpMV launches per host check
 * carry a HIP event pair on the ctx stream (hipExtLaunchKernelGGL start/stop events: the
 * kernel's own execution interval; launches skipped after convergence are not counted).
 * hf_get_profile returns the summed duration and the number of launches. */
int hf_set_profile(hf_ctx* ctx, int32_t on);
int hf_get_profile(hf_ctx* ctx, double* spmv_ms_sum, int64_t* spmv_launches);
/* GPU time (ms, HIP events on the ctx stream) of the last hf_step / hf_run / hf_assemble. */
int hf_last_gpu_ms(hf_ctx* ctx, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* HEATFLOW_HIP_H */
