"""Backward-Euler heat problem on the HIP backend.

Host-side equivalent of the block *forms -> assemble_matrix -> KSP -> time loop* of the
reference drivers (run_with_diamond.py:321-394, 456-504), with the seven dolfinx/PETSc
calls replaced by libheatflow_hip.so (see include/heatflow_hip.h for the mapping).

The loop order is the reference's: ``bc.update(t)`` -> RHS (M u^n, lifting, set_bc) ->
solve in place -> sample watchers.  The solver differs by design: Jacobi-PCG on the
GPU instead of a cached MUMPS LU; it stops at ``||D^-1 r|| <= rtol ||D^-1 b||``
(Kelvin-scaled residual) and reports the iteration count of every step.
"""
from __future__ import annotations

import time

import numpy as np

from .bc import gather_bc_values, gather_plan, merge_bcs
from .hip_backend import ASM_ROW_GATHER, PC_AMG, PC_JACOBI, HeatflowHIP

# Default PCG tolerance: at rtol = 1e-10 the temperature field agrees with a sparse
# direct solve of the same system to ~3e-6 K (measured on the stock and the 1M-DOF
# meshes, tests/test_gpu_parity.py); the stated parity bound is 1e-4 K absolute.
DEFAULT_RTOL = 1e-10
DEFAULT_MAX_IT = 20000


class HeatProblem:
    """Mesh + coefficients + Dirichlet rows on one GPU context.

    Parameters
    ----------
    coords (n,2) [z,r], tris (n_e,3), tags (n_e,)
    tag_to_k, tag_to_rho_cv : {cell tag: value}   (run_with_diamond.py:286-287)
    dt : time step
    bcs : list of RowDirichletBC in application order (later wins on shared DOFs)
    u0 : scalar or (n,) initial temperature
    backend : an object with the HeatflowHIP interface; default = a new HeatflowHIP
    assembly_mode : ASM_ROW_GATHER (default: a lane per CSR row, no atomics, bitwise reproducible matrices),
              ASM_LDS_COLORED (LDS scatter by colours, reproducible), ASM_LDS_ATOMIC (LDS atomics, diagonals
              summed in arrival order) or ASM_GLOBAL_ATOMIC (baseline)
    pattern : connectivity tables exported by another context on the same mesh (HeatflowHIP.export_pattern)
    amg : multigrid hierarchy exported by another context on the same mesh (HeatflowHIP.amg_export; needs precond=PC_AMG
              and amg_reuse=True): installed instead of being built
    precond : PC_JACOBI (Jacobi-PCG, the north-star solver) or PC_AMG (PCG preconditioned by a
              smoothed-aggregation V-cycle: same stopping rule and answer, ~50x fewer iterations)
    """

    def __init__(self, coords, tris, tags, tag_to_k, tag_to_rho_cv, dt, bcs, u0, *, backend=None, device_id=0,
                 assembly_mode=ASM_ROW_GATHER, rtol=DEFAULT_RTOL, atol=0.0, max_it=DEFAULT_MAX_IT,
                 precond=PC_JACOBI, amg_reuse=False, pattern=None, amg=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.n = self.coords.shape[0]
        self.dt = float(dt)
        self.bcs = list(bcs)
        self.rtol, self.atol, self.max_it = float(rtol), float(atol), int(max_it)
        self.assembly_mode = assembly_mode
        self.precond = precond
        self.backend = backend if backend is not None else HeatflowHIP(device_id)
        self._own_backend = backend is None

        t0 = time.perf_counter()
        if pattern is None:
            self.backend.set_mesh(self.coords, tris, tags)
        else:   # connectivity tables built once elsewhere (another rank / context of the sweep): install, do not rebuild
            self.backend.set_mesh(self.coords, tris, tags, pattern=pattern)
        self.mesh_seconds = time.perf_counter() - t0
        self.set_materials(tag_to_k, tag_to_rho_cv, assemble=False)
        if self.bcs:
            self.bc_dofs, self._owner, self._pos = merge_bcs(self.bcs)
        else:
            self.bc_dofs = np.zeros(0, dtype=np.int32)
            self._owner = self._pos = np.zeros(0, dtype=np.int64)
        self.backend.set_dirichlet(self.bc_dofs)
        self.backend.set_precond(precond, amg_reuse)
        if amg is not None and precond == PC_AMG and amg_reuse:
            self.backend.amg_install(amg)
        self.backend.assemble(self.dt, self.assembly_mode)
        u = np.full(self.n, float(u0)) if np.isscalar(u0) else np.asarray(u0, dtype=np.float64)
        self.backend.set_state(u)
        self.setup_seconds = time.perf_counter() - t0
        self.iters = []

    def set_materials(self, tag_to_k, tag_to_rho_cv, assemble=True):
        """(Re)load the coefficient tables; with ``assemble`` re-value M, A (kappa sweeps reuse
        the mesh, the pattern and the Dirichlet set)."""
        tags = sorted(tag_to_k)
        self.backend.set_materials(np.array(tags, dtype=np.int32),
                                   np.array([tag_to_k[t] for t in tags], dtype=np.float64),
                                   np.array([tag_to_rho_cv[t] for t in tags], dtype=np.float64))
        if assemble:
            self.backend.assemble(self.dt, self.assembly_mode)

    def close(self):
        if self._own_backend:
            self.backend.close()

    # -- boundary values -------------------------------------------------------------------
    def bc_values(self, t, only=None):
        """g_B(t).  ``only`` = BCs to refresh (the reference refreshes all once at t=0 and then
        only the heated line, run_with_diamond.py:458-459, 472)."""
        for bc in (self.bcs if only is None else only):
            bc.update(t)
        if getattr(self, "_plan", None) is None or len(self._plan) != len(self.bcs):
            self._plan = gather_plan(len(self.bcs), self._owner, self._pos)
        return gather_bc_values(self.bcs, self._owner, self._pos, self._plan)

    # -- stepping --------------------------------------------------------------------------
    def set_state(self, u):
        u = np.full(self.n, float(u)) if np.isscalar(u) else u
        self.backend.set_state(u)

    def state(self):
        return self.backend.get_state()

    def step(self, t, only=None):
        g = self.bc_values(t, only)
        it, res = self.backend.step(g, self.rtol, self.atol, self.max_it)
        self.iters.append(it)
        return it, res

    def run(self, num_steps, watcher_nodes=None, time_varying=None, first_step=0):
        """``num_steps`` steps t_k = (k+1) dt in one backend call (hf_run): the boundary values
        of all steps are tabulated on the host first.  Returns (times, samples, iters)."""
        for bc in self.bcs:
            bc.update(0.0)
        times = (np.arange(first_step, first_step + num_steps) + 1) * self.dt
        g_all = np.empty((num_steps, len(self.bc_dofs)), dtype=np.float64)
        for k, t in enumerate(times):
            g_all[k] = self.bc_values(t, time_varying)
        samples, iters = self.backend.run(g_all, self.rtol, self.atol, self.max_it, watcher_nodes)
        self.iters.extend(int(i) for i in iters)
        return times, samples, iters


def nearest_nodes(coords, points):
    """Nearest mesh node of each (z, r) point (run_with_diamond.py:443-449 asks a cKDTree over geometry.x[:, :2] for it).
    A handful of watcher points does not repay a tree over the whole mesh (60 ms to build at 2e5 nodes, plus the import of
    scipy.spatial): one pass of squared distances per point; of equidistant nodes the lowest-numbered one is taken."""
    xy = np.asarray(coords, dtype=np.float64)[:, :2]
    out = np.empty(len(points), dtype=np.int32)
    for q, p in enumerate(points):
        dz, dr = xy[:, 0] - float(p[0]), xy[:, 1] - float(p[1])
        out[q] = int(np.argmin(dz * dz + dr * dr))
    return out
