"""Shared implementation of the ``run_simulation`` entry points.

``run_with_diamond.run_simulation`` / ``run_no_diamond.run_simulation`` keep the reference
signature (run_with_diamond.py:27, run_no_diamond.py:29):

    run_simulation(cfg, mesh_folder, rebuild_mesh=False, visualize_mesh=False,
                   output_folder=None, watcher_points=None, write_xdmf=True, suppress_print=False)

and the same side effects: mesh cache ``<mesh_folder>/mesh.msh`` + ``mesh_cfg.yaml`` (cfg copy
plus ``material_tags``, :194-230), ``<out>/used_config.yaml`` (:403-404), ``<out>/watcher_points.csv``
with columns ``time,<watcher names>`` (:510-515), default output folder
``sim_outputs/refactor_test`` (:405-409), ``FileNotFoundError`` when the cache is missing
(:219-226), ``ValueError`` for a bad ``watcher_points`` (:439) or heating CSV (:268-271),
``RuntimeError("No DOFs found ...")`` from the BC location (bc.py:105-106).

What differs by design: the mesh comes from heatflow_amd.mesh (no gmsh), assembly and the time
loop run on the GPU through libheatflow_hip.so, and a :class:`SimulationSession` keeps mesh,
sparsity pattern and mass matrix resident so a sweep re-values only what changed.
"""
from __future__ import annotations

import contextlib
import copy
import csv
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np
import yaml

_YAML_DUMPER = getattr(yaml, "CSafeDumper", yaml.SafeDumper)     # libyaml when PyYAML was built with it (10x faster)


def _dump_yaml(obj, f):
    yaml.dump(obj, f, Dumper=_YAML_DUMPER)


from .bc import P1Space, RowDirichletBC
from .geometry import stack_no_diamond, stack_with_diamond
from .heating import HeatingCurve
from .mesh import Mesh, load_mesh_arrays
from .solver import DEFAULT_MAX_IT, DEFAULT_RTOL, HeatProblem

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_quiet_lock = threading.Lock()
_quiet_depth = 0
_quiet_saved = None


@contextlib.contextmanager
def suppress_output(enabled):
    """Redirect stdout/stderr to devnull while ``enabled`` (reference run_with_diamond.py:18-25).
    Re-entrant across threads (concurrent sweep points): the first entrant redirects, the last one
    out restores."""
    global _quiet_depth, _quiet_saved
    if not enabled:
        yield
        return
    with _quiet_lock:
        if _quiet_depth == 0:
            sink = open(os.devnull, "w")
            _quiet_saved = (sys.stdout, sys.stderr, sink)
            sys.stdout = sys.stderr = sink
        _quiet_depth += 1
    try:
        yield
    finally:
        with _quiet_lock:
            _quiet_depth -= 1
            if _quiet_depth == 0:
                sys.stdout, sys.stderr, sink = _quiet_saved
                sink.close()
                _quiet_saved = None


def _resolve(path):
    """Heating file paths in the cfgs are relative to the repository root (cwd in the reference)."""
    if os.path.isabs(path) or os.path.isfile(path):
        return path
    alt = os.path.join(_PKG_ROOT, path)
    return alt if os.path.isfile(alt) else path


def _parse_watchers(watcher_points):
    if watcher_points is None:
        return [], []
    if isinstance(watcher_points, dict):
        return list(watcher_points.keys()), [tuple(v) for v in watcher_points.values()]
    if isinstance(watcher_points, list):
        return [pt["name"] for pt in watcher_points], [tuple(pt["coords"]) for pt in watcher_points]
    raise ValueError("watcher_points must be a dict or list of dicts")


_pending_mesh_writes = []


def flush_mesh_writes():
    """Wait for the mesh-cache files a ``prepare_mesh(..., defer_write=True)`` left to a background thread."""
    while _pending_mesh_writes:
        th, err = _pending_mesh_writes.pop()
        th.join()
        if err:
            raise err[0]


def prepare_mesh(cfg, mesh_folder, rebuild_mesh, stack, defer_write=False):
    """Build-and-cache or load the mesh.  Returns (coords, tris, tags, material_tags).  ``defer_write``: the cache files
    (``mesh.msh``, its sidecar, ``mesh_cfg.yaml``: 0.1 s of ASCII at 2e5 nodes) are written by a background thread while
    the caller goes on with the arrays - a sweep broadcasts them and never reads the files back; :func:`flush_mesh_writes`
    waits for them."""
    mesh_cfg_path = os.path.join(mesh_folder, "mesh_cfg.yaml")
    mesh_file_path = os.path.join(mesh_folder, "mesh.msh")
    if rebuild_mesh:
        mesh = Mesh(name="mesh.msh", boundaries=stack.bounds, materials=stack.materials)
        mesh.build_mesh()
        tag_map = {m.name: int(getattr(m, "_tag", -1)) for m in stack.materials}
        os.makedirs(mesh_folder, exist_ok=True)
        mesh_cfg = copy.deepcopy(cfg)
        mesh_cfg["material_tags"] = tag_map

        def write_files():
            with open(mesh_cfg_path, "w") as f:
                _dump_yaml(mesh_cfg, f)
            mesh.write(mesh_file_path)

        if defer_write:
            err = []

            def guarded():
                try:
                    write_files()
                except BaseException as e:      # noqa: BLE001 - re-raised by flush_mesh_writes
                    err.append(e)

            th = threading.Thread(target=guarded, name="heatflow-mesh-write", daemon=False)
            th.start()
            _pending_mesh_writes.append((th, err))
        else:
            write_files()
        return mesh.coords, mesh.tris, mesh.tags, tag_map
    missing = [nm for nm, p in (("mesh.msh", mesh_file_path), ("mesh_cfg.yaml", mesh_cfg_path)) if not os.path.isfile(p)]
    if missing:
        raise FileNotFoundError(f"Missing required file(s) in {mesh_folder}: {', '.join(missing)}")
    with open(mesh_cfg_path) as f:
        mesh_cfg = yaml.safe_load(f)
    tag_map = mesh_cfg["material_tags"]
    coords, tris, tags = load_mesh_arrays(mesh_file_path)
    return coords, tris, tags, tag_map


def build_pattern_blob(coords, tris, tags, device_id=0):
    """The connectivity-derived tables of a mesh (CSR pattern, compressed column lists, row-gather assembly
    lists), built ONCE on this GPU and returned as a uint8 array: a sweep hands it to every solver session of
    every rank (``SimulationSession(..., pattern=blob)``), which then installs instead of rebuilding."""
    from .hip_backend import HeatflowHIP

    with HeatflowHIP(device_id) as be:
        be.set_mesh(coords, tris, tags)
        return be.export_pattern()


class SimulationSession:
    """A mesh resident on one GPU, reusable for many runs of the same geometry.

    ``run(cfg, ...)`` re-values the coefficient tables / matrices only when they differ from
    the previous run and re-tabulates the boundary values (fwhm, heating curve) on the host.
    """

    def __init__(self, coords, tris, tags, material_tags, *, device_id=0, backend=None, rtol=DEFAULT_RTOL,
                 max_it=DEFAULT_MAX_IT, assembly_mode=3, precond=1, pattern=None, hierarchy=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.tris = np.ascontiguousarray(tris, dtype=np.int32)
        self.tags = np.ascontiguousarray(tags, dtype=np.int32)
        self.material_tags = dict(material_tags)
        self.device_id, self.backend = device_id, backend
        self.rtol, self.max_it, self.assembly_mode = rtol, max_it, assembly_mode
        self.precond = precond           # 1 = multigrid-preconditioned CG (default), 0 = Jacobi-PCG
        self.pattern = pattern           # connectivity tables built once for this mesh (build_pattern_blob), or None
        # multigrid hierarchy another session built on this mesh: {"blob": HeatflowHIP.amg_export() output, "k": {cell tag:
        # conductivity it was built for}}; installed by the first problem this session creates instead of a host set-up
        self.hierarchy = hierarchy
        self.problem = None
        self._key = None
        self._tree = None
        self._space = None
        self._dof_cache = {}

    def close(self):
        if self.problem is not None:
            self.problem.close()
            self.problem = None

    def export_hierarchy(self, into=None):
        """The multigrid hierarchy of the resident problem for other sessions on this mesh (``hierarchy=`` of their
        constructor): {"blob": uint8 array (or None when written ``into`` = (address, nbytes)), "k": {cell tag: conductivity}}."""
        if self.problem is None or self.precond != 1:
            raise RuntimeError("export_hierarchy: no resident problem with the multigrid preconditioner")
        return {"blob": self.problem.backend.amg_export(into=into), "k": dict(self._k_hier)}

    def prepare(self, cfg, stack):
        """Create the resident problem for ``cfg`` (mesh tables, matrices, multigrid hierarchy) without running a step."""
        num_steps = int(cfg["timing"]["num_steps"])
        dt = float(cfg["timing"]["t_final"]) / num_steps
        bcs = self._boundary_conditions(cfg, stack)
        tag_to_k, tag_to_rc = self._tables(stack)
        self._ensure_problem(self._problem_key(dt, tag_to_rc, bcs), tag_to_k, tag_to_rc, dt, bcs, float(cfg["heating"]["ic_temp"]))

    def _tables(self, stack):
        tag_to_k = {self.material_tags[m.name]: m.properties["k"] for m in stack.materials}
        tag_to_rc = {self.material_tags[m.name]: m.properties["rho_cv"] for m in stack.materials}
        return tag_to_k, tag_to_rc

    def _boundary_conditions(self, cfg, stack, two_sided=False):
        """[left, right, top, heated line(s)] of one configuration (reference run_with_diamond.py:343-374)."""
        ic_temp = float(cfg["heating"]["ic_temp"])
        heat = HeatingCurve(_resolve(cfg["heating"]["file"]), ic_temp, float(cfg["heating"]["fwhm"]))
        if self._space is None:
            self._space = P1Space(self.coords)
        V = self._space

        def located(location, value, **kw):
            # the DOF set of an edge / line depends on the mesh and the location arguments only: located once per session
            key = (location,) + tuple(sorted(kw.items()))
            if key not in self._dof_cache:
                self._dof_cache[key] = RowDirichletBC(V, location, value=value, **kw).row_dofs
            return RowDirichletBC(V, location, value=value, row_dofs=self._dof_cache[key], **kw)

        bcs = [
            located("left", ic_temp),
            located("right", ic_temp),
            located("top", ic_temp),      # named bottom_bc in the reference, location 'top'
            located("x", heat.gaussian, coord=float(stack.heated_z), length=abs(stack.r_sample) * 2, center=0.0),
        ]
        if two_sided:
            # EXTENSION without a reference implementation (BASELINE config 4 "konopkova two-sided",
            # SURVEY 8d C4): a second Gaussian Dirichlet line on the outer face of the o-side coupler,
            # driven by the CSV's `oside` column with the same offset-to-ic_temp convention.
            heat_o = HeatingCurve(_resolve(cfg["heating"]["file"]), ic_temp, float(cfg["heating"]["fwhm"]), column="oside")
            bcs.append(located("x", heat_o.gaussian, coord=float(stack.heated_z_oside), length=abs(stack.r_sample) * 2, center=0.0))
        return bcs

    def _problem_key(self, dt, tag_to_rc, bcs):
        # the resident problem is reusable only for exactly the same Dirichlet DOF sets, in the same order
        return (dt, tuple(sorted(tag_to_rc.items())),
                tuple(hashlib.sha1(np.ascontiguousarray(b.row_dofs, dtype=np.int64).tobytes()).hexdigest() for b in bcs))

    def _ensure_problem(self, key, tag_to_k, tag_to_rc, dt, bcs, ic_temp):
        """The resident HeatProblem for ``key`` (built if absent), its operator valued for ``tag_to_k``."""
        if self.problem is None or key != self._key:
            self.close()
            print("Assigning material properties...")
            shared = self.hierarchy if (self.hierarchy is not None and self.precond == 1) else None
            self.problem = HeatProblem(self.coords, self.tris, self.tags, tag_to_k, tag_to_rc, dt, bcs, ic_temp,
                                       backend=self.backend, device_id=self.device_id, rtol=self.rtol,
                                       max_it=self.max_it, assembly_mode=self.assembly_mode, precond=self.precond,
                                       amg_reuse=True, pattern=self.pattern, amg=shared["blob"] if shared else None)
            self._key = key
            self._k = dict(tag_to_k)
            # conductivities the multigrid levels were built for: this problem's, or those of the session that shared them
            self._k_hier = dict(shared["k"]) if shared else dict(tag_to_k)
            self.hierarchy = None                    # a later problem of this session (other dt / Dirichlet set) builds its own
            if shared and set(self._k_hier) == set(tag_to_k):
                drift = max(max(tag_to_k[t] / self._k_hier[t], self._k_hier[t] / tag_to_k[t]) for t in tag_to_k)
                if drift > 2.0:                      # too far from this point's operator to be a good frozen hierarchy: rebuild
                    self.problem.backend.set_precond(1, False)
                    self.problem.set_materials(tag_to_k, tag_to_rc)
                    self.problem.backend.set_precond(1, True)
                    self._k_hier = dict(tag_to_k)
            print("Material properties assigned.")
            return
        self.problem.bcs = bcs
        if tag_to_k != self._k:
            self._revalue(tag_to_k, tag_to_rc)

    def _revalue(self, tag_to_k, tag_to_rc, allow_rebuild=True):
        # re-value A on the resident pattern; the frozen coarse levels stay a good preconditioner
        # while no conductivity moved by more than 2x from the values they were built for
        # (measured: 3.8 -> 60 W/m/K triples the iteration count), beyond that rebuild them
        drift = max(max(tag_to_k[t] / self._k_hier[t], self._k_hier[t] / tag_to_k[t]) for t in tag_to_k)
        if drift > 2.0 and self.precond == 1 and allow_rebuild:
            self.problem.backend.set_precond(1, False)
            self.problem.set_materials(tag_to_k, tag_to_rc)
            self.problem.backend.set_precond(1, True)
            self._k_hier = dict(tag_to_k)
        else:
            self.problem.set_materials(tag_to_k, tag_to_rc)
        self._k = dict(tag_to_k)

    def _watcher_nodes(self, watcher_points):
        names, coords_w = _parse_watchers(watcher_points)
        if not names:
            return names, None
        from .solver import nearest_nodes
        return names, nearest_nodes(self.coords, coords_w)

    def run_batch(self, cfgs, stacks, watcher_points=None, read_flux=False):
        """``len(cfgs)`` in (2, 4, 8, 16) simulations of this mesh advanced together (hf_batch_*): the points of a
        sweep that share geometry, time stepping and rho_c; they may differ in the boundary values (fwhm,
        heating curve: one shared operator) and in the conductivities (one operator per column, shared
        frozen multigrid hierarchy).  ``read_flux`` adds run_no_diamond's per-step gradient projection for every
        column (hf_batch_run_flux).  Returns one result dict per configuration, as :meth:`run` does."""
        nv = len(cfgs)
        if nv not in (2, 4, 8, 16):
            raise ValueError("run_batch: 2, 4, 8 or 16 configurations at a time")
        t_start = time.time()
        cfg0 = cfgs[0]
        num_steps = int(cfg0["timing"]["num_steps"])
        dt = float(cfg0["timing"]["t_final"]) / num_steps
        ic_temp = float(cfg0["heating"]["ic_temp"])
        cols = []
        for cfg, stack in zip(cfgs, stacks):
            bcs = self._boundary_conditions(cfg, stack)
            tk, trc = self._tables(stack)
            key = self._problem_key(float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"]), trc, bcs)
            cols.append((bcs, tk, trc, key))
            if int(cfg["timing"]["num_steps"]) != num_steps or float(cfg["heating"]["ic_temp"]) != ic_temp or key != cols[0][3]:
                raise ValueError("run_batch: the configurations must share time stepping, ic_temp, rho_c and the Dirichlet sets")
        percol = any(c[1] != cols[0][1] for c in cols)
        mid = cols[nv // 2]
        # conductivities that differ between the columns; a single one (sweep_test.py's kappa_sample list) makes
        # the operators an affine family A + (kappa_j - kappa_ref) A1: two shared matrices instead of nv
        varying = [t for t in mid[1] if any(c[1][t] != mid[1][t] for c in cols)]
        affine = percol and len(varying) == 1
        self._ensure_problem(mid[3], mid[1], mid[2], dt, mid[0], ic_temp)
        prob = self.problem
        be = prob.backend
        if percol and self.precond == 1:             # all columns share the hierarchy: keep every column within its range
            worst = max(max(max(c[1][t] / self._k_hier[t], self._k_hier[t] / c[1][t]) for t in c[1]) for c in cols)
            if worst > 2.0:
                be.set_precond(1, False)
                prob.set_materials(mid[1], mid[2])
                be.set_precond(1, True)
                self._k_hier, self._k = dict(mid[1]), dict(mid[1])
        times = (np.arange(num_steps) + 1) * dt
        g_all = np.empty((num_steps, len(prob.bc_dofs), nv), dtype=np.float64)
        tabulated = {}                                   # columns with the same boundary definition share one table
        for j, (bcs, _, _, _) in enumerate(cols):
            h = cfgs[j]["heating"]
            sig = (str(h["file"]), float(h["ic_temp"]), float(h["fwhm"]), len(bcs))
            if sig not in tabulated:
                prob.bcs = bcs
                for bc in bcs:
                    bc.update(0.0)
                tabulated[sig] = np.stack([prob.bc_values(t, bcs[3:]) for t in times])
            g_all[:, :, j] = tabulated[sig]
        prob.bcs = cols[-1][0]
        names, nodes = self._watcher_nodes(watcher_points)
        if affine and mid[1] != self._k:
            self._revalue(mid[1], mid[2], allow_rebuild=False)     # the context's operator is the reference of the family
        flux0 = FluxSampler(self.coords) if read_flux else None
        if flux0 is not None and not getattr(prob, "_flux_ready", False):
            print("Setting up radial heat flux sampling...")
            be.flux_setup()                    # once per mesh: the unit-coefficient mass matrix does not depend on kappa
            prob._flux_ready = True
        be.batch_begin(nv, per_column_operator=2 if affine else (1 if percol else 0))
        try:
            if affine:
                be.batch_set_affine(varying, [c[1][varying[0]] - mid[1][varying[0]] for c in cols])
            elif percol:
                for j, (_, tk, trc, _) in enumerate(cols):
                    if tk != self._k:
                        self._revalue(tk, trc, allow_rebuild=False)
                    be.batch_load_column(j)
            for j in range(nv):
                be.batch_set_state(j, np.full(prob.n, ic_temp))
            print("Beginning loop...")
            t_loop = time.time()
            if flux0 is None:
                samples, iters = be.batch_run(g_all, self.rtol, 0.0, self.max_it, nodes)
            else:                              # d/dr only, as the single run (run_no_diamond.py:553-566 reads nothing else)
                samples, iters, grad = be.batch_run(g_all, self.rtol, 0.0, self.max_it, nodes, flux_nodes=flux0.nodes,
                                                    flux_components=2, flux_rtol=self.rtol, flux_max_it=5000)
            loop_time = time.time() - t_loop
        finally:
            be.batch_end()
        fluxes = [None] * nv
        if flux0 is not None:
            for j in range(nv):
                fluxes[j] = flux0 if j == 0 else flux0.clone()
                for s, t in enumerate(times):
                    fluxes[j].record_sampled(t, grad[s, 0, j])
        print(f"Simulation progress: 100% (step {num_steps}/{num_steps}) | {nv} runs together | Avg time/step: "
              f"{loop_time / num_steps:.4f} s | PCG iterations/step: mean {np.mean(iters):.0f}, max {int(np.max(iters))}")
        return [{"times": times.copy(), "watcher_names": names,
                 "watchers": {nm: samples[:, j, k].copy() for k, nm in enumerate(names)},
                 "iters": iters[:, j].copy(), "loop_time": loop_time / nv, "startup_time": (t_loop - t_start) / nv,
                 "n_dof": prob.n, "dt": dt, "flux": fluxes[j], "batch": nv} for j in range(nv)]

    def run(self, cfg, stack, watcher_points=None, field_sink=None, read_flux=False, two_sided=False):
        """One simulation (reference loop run_with_diamond.py:456-504).  Returns a dict with
        ``times``, ``watchers`` {name: array}, ``iters``, timing numbers.  ``read_flux`` adds the
        per-step gradient projection of run_no_diamond.py:543-566 (``flux`` entry of the result)."""
        t_start = time.time()
        t_final = float(cfg["timing"]["t_final"])
        num_steps = int(cfg["timing"]["num_steps"])
        dt = t_final / num_steps
        ic_temp = float(cfg["heating"]["ic_temp"])
        bcs = self._boundary_conditions(cfg, stack, two_sided)
        varying = bcs[3:]
        tag_to_k, tag_to_rc = self._tables(stack)
        key = self._problem_key(dt, tag_to_rc, bcs)
        fresh = self.problem is None or key != self._key
        self._ensure_problem(key, tag_to_k, tag_to_rc, dt, bcs, ic_temp)
        if not fresh:
            self.problem.set_state(ic_temp)
            self.problem.iters = []
        prob = self.problem
        names, nodes = self._watcher_nodes(watcher_points)

        flux = FluxSampler(self.coords) if read_flux else None
        if flux is not None and not getattr(prob, "_flux_ready", False):
            print("Setting up radial heat flux sampling...")
            prob.backend.flux_setup()          # once per mesh: the unit-coefficient mass matrix does not depend on kappa
            prob._flux_ready = True
        print("Beginning loop...")
        t_loop = time.time()
        if field_sink is None and flux is None:
            times, samples, iters = prob.run(num_steps, watcher_nodes=nodes, time_varying=varying)
        else:  # step-wise: every field goes to the sink (visualisation) and / or through the flux projection
            for bc in bcs:
                bc.update(0.0)
            times, rows, iters = [], [], []
            for step in range(num_steps):
                t = (step + 1) * dt
                it, _ = prob.step(t, only=varying)
                if flux is not None:
                    prob.backend.flux_solve(self.rtol, 5000, want_z=False)          # d/dr only, on the device
                    flux.record_sampled(t, prob.backend.flux_sample(flux.nodes, want_z=False)[1])
                if field_sink is not None:
                    u = prob.state()
                    field_sink(t, u)
                    rows.append(u[nodes] if nodes is not None else np.zeros(0))
                else:
                    rows.append(prob.backend.sample(nodes) if nodes is not None else np.zeros(0))
                times.append(t)
                iters.append(it)
            times, samples, iters = np.array(times), np.array(rows), np.array(iters)
        loop_time = time.time() - t_loop
        print(f"Simulation progress: 100% (step {num_steps}/{num_steps}) | Avg time/step: {loop_time / num_steps:.4f} s"
              f" | PCG iterations/step: mean {np.mean(iters):.0f}, max {int(np.max(iters))}")
        return {
            "times": np.asarray(times), "watcher_names": names,
            "watchers": {nm: samples[:, k] for k, nm in enumerate(names)},
            "iters": np.asarray(iters), "loop_time": loop_time, "startup_time": t_loop - t_start,
            "n_dof": prob.n, "dt": dt, "flux": flux,
        }


class FluxSampler:
    """Bookkeeping of run_no_diamond's radial-gradient outputs (reference run_no_diamond.py):
    * smoothed: mean of d T/d r over the nodes with 0 < r <= 0.25 um that fall into each 0.2-um z bin
      (:494-513, :553-556) -> ``radial_gradient.csv`` (columns = bin centres, :603-608);
    * raw: d T/d r at the nodes on the axis (|r| <= 1e-12), ordered by z (:457-465, :559-566)
      -> ``radial_gradient_raw.csv`` (columns = their z, :611-617)."""

    DZ_BIN = 0.2e-6
    BAND = 0.25e-6
    R_TOL = 1e-12

    def __init__(self, coords):
        z, r = coords[:, 0], coords[:, 1]
        edges = np.arange(z.min(), z.max() + self.DZ_BIN, self.DZ_BIN)
        band = np.nonzero((r > 0.0) & (r <= self.BAND))[0]
        k = np.searchsorted(edges, z[band]) - 1
        ok = (k >= 0) & (k < len(edges) - 1)
        band, k = band[ok], k[ok]
        self.z_centres, self.groups = [], []
        for b in np.unique(k):
            self.z_centres.append(0.5 * (edges[b] + edges[b + 1]))
            self.groups.append(band[k == b])
        axis = np.nonzero(np.abs(r) <= self.R_TOL)[0]
        order = np.argsort(z[axis], kind="stable")
        self.axis_nodes = axis[order]
        self.axis_z = z[self.axis_nodes]
        self.times, self.rows, self.raw_rows = [], [], []
        # the only nodes whose gradient is ever read: band groups first, axis nodes last (hf_flux_sample order)
        self.nodes = np.concatenate(self.groups + [self.axis_nodes]).astype(np.int32) if self.groups else self.axis_nodes.astype(np.int32)
        self._cuts = np.cumsum([len(g) for g in self.groups])

    def clone(self):
        """An empty sampler of the same mesh (the node groups are shared, the recorded rows are not)."""
        import copy as _copy
        c = _copy.copy(self)
        c.times, c.rows, c.raw_rows = [], [], []
        return c

    def record(self, t, grad_r):
        """grad_r: the full nodal field."""
        self.record_sampled(t, np.asarray(grad_r)[self.nodes])

    def record_sampled(self, t, values):
        """values: d T/d r at ``self.nodes`` (what hf_flux_sample returns)."""
        self.times.append(float(t))
        parts = np.split(values, self._cuts) if len(self._cuts) else [values]
        self.rows.append([float(np.mean(p)) for p in parts[:len(self.groups)]])
        self.raw_rows.append(parts[-1].astype(float).tolist())

    @staticmethod
    def _write(path, times, columns, rows):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["time"] + [repr(float(c)) for c in columns])
            for t, row in zip(times, rows):
                w.writerow([repr(t)] + [repr(v) for v in row])

    def write(self, folder):
        if self.rows:
            self._write(os.path.join(folder, "radial_gradient.csv"), self.times, self.z_centres, self.rows)
            self._write(os.path.join(folder, "radial_gradient_raw.csv"), self.times, self.axis_z, self.raw_rows)


def write_watcher_csv(path, times, names, watchers):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["time"] + list(names))
        for k, t in enumerate(times):
            w.writerow([repr(float(t))] + [repr(float(watchers[nm][k])) for nm in names])


class _FieldWriter:
    """``output.xdmf`` time series (reference :414-424, :483-484 via dolfinx.io.XDMFFile).
    dolfinx stores the heavy data in HDF5; h5py is not available here, so the same XDMF 3 layout
    (mesh grid + temporal collection of the nodal attribute "Temperature (K)") points at raw
    little-endian binary files instead (``Format="Binary"``, readable by ParaView/VisIt):
    ``output_xy.bin`` (n x 2 float64), ``output_conn.bin`` (n_e x 3 int32) and
    ``output_fields.f64`` (one field per step, initial state first, addressed with ``Seek``)."""

    def __init__(self, folder, coords, tris, tags):
        self.folder = folder
        self.n, self.ne = len(coords), len(tris)
        np.ascontiguousarray(coords, dtype="<f8").tofile(os.path.join(folder, "output_xy.bin"))
        np.ascontiguousarray(tris, dtype="<i4").tofile(os.path.join(folder, "output_conn.bin"))
        np.ascontiguousarray(tags, dtype="<i4").tofile(os.path.join(folder, "output_cell_tags.bin"))
        self.f = open(os.path.join(folder, "output_fields.f64"), "wb")
        self.times = []

    def __call__(self, t, u):
        np.asarray(u, dtype="<f8").tofile(self.f)
        self.times.append(float(t))

    def close(self):
        self.f.close()
        n, ne = self.n, self.ne
        item = 'Format="Binary" Endian="Little"'
        out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0" xmlns:xi="http://www.w3.org/2001/XInclude">', " <Domain>",
               '  <Grid Name="mesh" GridType="Uniform">',
               f'   <Topology TopologyType="Triangle" NumberOfElements="{ne}" NodesPerElement="3">',
               f'    <DataItem {item} DataType="Int" Precision="4" Dimensions="{ne} 3">output_conn.bin</DataItem>',
               "   </Topology>", '   <Geometry GeometryType="XY">',
               f'    <DataItem {item} DataType="Float" Precision="8" Dimensions="{n} 2">output_xy.bin</DataItem>',
               "   </Geometry>", "  </Grid>",
               '  <Grid Name="Temperature (K)" GridType="Collection" CollectionType="Temporal">']
        for k, t in enumerate(self.times):
            out += ['   <Grid Name="Temperature (K)" GridType="Uniform">',
                    "    <xi:include xpointer=\"xpointer(/Xdmf/Domain/Grid[@GridType='Uniform'][1]/*[self::Topology or self::Geometry])\" />",
                    f'    <Time Value="{t!r}" />',
                    '    <Attribute Name="Temperature (K)" AttributeType="Scalar" Center="Node">',
                    f'     <DataItem {item} DataType="Float" Precision="8" Seek="{8 * n * k}" Dimensions="{n} 1">output_fields.f64</DataItem>',
                    "    </Attribute>", "   </Grid>"]
        out += ["  </Grid>", " </Domain>", "</Xdmf>"]
        with open(os.path.join(self.folder, "output.xdmf"), "w") as f:
            f.write("\n".join(out) + "\n")
        with open(os.path.join(self.folder, "output_fields.json"), "w") as f:
            json.dump({"name": "Temperature (K)", "n": n, "times": self.times, "dtype": "float64"}, f)


def run_simulation_impl(kind, cfg, mesh_folder, rebuild_mesh=False, visualize_mesh=False, output_folder=None,
                        watcher_points=None, write_xdmf=True, suppress_print=False, *, device_id=0, backend=None,
                        session=None, rtol=DEFAULT_RTOL, max_it=DEFAULT_MAX_IT, read_flux=True, precond=None,
                        two_sided=False):
    with suppress_output(suppress_print):
        program_start = time.time()
        stack = stack_with_diamond(cfg) if kind == "with_diamond" else stack_no_diamond(cfg)
        own_session = session is None
        if own_session:
            coords, tris, tags, tag_map = prepare_mesh(cfg, mesh_folder, rebuild_mesh, stack)
            session = SimulationSession(coords, tris, tags, tag_map, device_id=device_id, backend=backend, rtol=rtol,
                                        max_it=max_it, precond=1 if precond is None else precond)
        if visualize_mesh:
            print("visualize_mesh: the gmsh GUI is not part of this build; open mesh.msh in gmsh instead.")
        _parse_watchers(watcher_points)  # validate before any work, as the reference does at :431-439

        if output_folder is not None:
            save_folder = output_folder
            os.makedirs(save_folder, exist_ok=True)
            with open(os.path.join(save_folder, "used_config.yaml"), "w") as f:
                _dump_yaml(cfg, f)
        else:
            save_folder = os.path.join(os.getcwd(), "sim_outputs", "refactor_test")
            os.makedirs(save_folder, exist_ok=True)

        sink = _FieldWriter(save_folder, session.coords, session.tris, session.tags) if write_xdmf else None
        try:
            if sink is not None:
                sink(0.0, np.full(len(session.coords), float(cfg["heating"]["ic_temp"])))
            result = session.run(cfg, stack, watcher_points, field_sink=sink, read_flux=read_flux and kind == "no_diamond",
                                 two_sided=two_sided)
        finally:
            if sink is not None:
                sink.close()
            if own_session:
                session.close()

        if watcher_points is not None:
            write_watcher_csv(os.path.join(save_folder, "watcher_points.csv"), result["times"], result["watcher_names"],
                              result["watchers"])
        if result.get("flux") is not None:
            result["flux"].write(save_folder)
            print(f"Saved raw gradient data at r=0 nodes to {os.path.join(save_folder, 'radial_gradient_raw.csv')}")
        total = time.time() - program_start
        n_steps = len(result["times"])
        print("\n--- Timing Summary ---")
        print(f"Total time: {total:.2f} s")
        print(f"Startup time: {total - result['loop_time']:.2f} s")
        print(f"Loop time: {result['loop_time']:.2f} s")
        print(f"Average time per step: {result['loop_time'] / max(n_steps, 1):.4f} s")
        print("----------------------\n")
        result["save_folder"] = save_folder
        result["total_time"] = total
        return result


def run_simulation_batch_impl(kind, cfgs, output_folders, watcher_points_list, session, suppress_print=True, read_flux=False):
    """``run_simulation`` for 2, 4, 8 or 16 configurations of one resident mesh at once (SimulationSession.run_batch):
    the same per-run artefacts (``used_config.yaml``, ``watcher_points.csv``, with ``read_flux`` also run_no_diamond's
    ``radial_gradient.csv`` / ``radial_gradient_raw.csv``) in each output folder, no XDMF.  The columns of a batch are
    sampled at the same nodes: all configurations must name the same watcher points (``ValueError`` otherwise - the
    points of a sweep group share their geometry and therefore do).  Returns the list of result dicts."""
    with suppress_output(suppress_print):
        t0 = time.time()
        stacks = [stack_with_diamond(c) if kind == "with_diamond" else stack_no_diamond(c) for c in cfgs]
        parsed = [_parse_watchers(wp) for wp in watcher_points_list]
        for names_j, coords_j in parsed[1:]:
            if names_j != parsed[0][0] or not np.array_equal(np.asarray(coords_j, dtype=float), np.asarray(parsed[0][1], dtype=float)):
                raise ValueError("run_simulation_batch_impl: the configurations of a batch must share their watcher points "
                                 "(one set of sample nodes serves every column)")
        for cfg, folder in zip(cfgs, output_folders):
            os.makedirs(folder, exist_ok=True)
            with open(os.path.join(folder, "used_config.yaml"), "w") as f:
                _dump_yaml(cfg, f)
        results = session.run_batch(cfgs, stacks, watcher_points_list[0], read_flux=read_flux and kind == "no_diamond")
        for res, folder, wp in zip(results, output_folders, watcher_points_list):
            if wp is not None:
                write_watcher_csv(os.path.join(folder, "watcher_points.csv"), res["times"], res["watcher_names"], res["watchers"])
            if res.get("flux") is not None:
                res["flux"].write(folder)
            res["save_folder"] = folder
            res["total_time"] = (time.time() - t0) / len(cfgs)
        return results


def cli(kind, argv=None):
    """Command line of the drivers.  (The reference's own CLI cannot start: argparse is given
    ``type='dict'``, run_with_diamond.py:540; ``--watcher-points`` takes JSON here.)"""
    import argparse

    p = argparse.ArgumentParser(description="Heatflow simulation runner (MI355X)")
    p.add_argument("--config", type=str, default="simulation_template.yaml")
    p.add_argument("--mesh-folder", type=str, default="meshes")
    p.add_argument("--rebuild-mesh", action="store_true")
    p.add_argument("--visualize-mesh", action="store_true")
    p.add_argument("--output-folder", type=str)
    p.add_argument("--watcher-points", type=json.loads, help='JSON, e.g. {"pside": [z, r]}')
    p.add_argument("--write-xdmf", action="store_true")
    p.add_argument("--suppress-print", action="store_true")
    p.add_argument("--device", type=int, default=0)
    a = p.parse_args(argv)
    with open(a.config) as f:
        cfg = yaml.safe_load(f)
    run_simulation_impl(kind, cfg, a.mesh_folder, a.rebuild_mesh, a.visualize_mesh, a.output_folder, a.watcher_points,
                        a.write_xdmf, a.suppress_print, device_id=a.device)
    return 0


if __name__ == "__main__":
    sys.exit(cli("with_diamond"))
