"""Parameter sweeps, sharded one point per GPU.

Keeps the reference's sweep surface (parameter_sweep.py): ``create_parameter_grid`` (:195-235,
logspace fwhm x logspace k x linspace width, grouped by width), ``modify_config_for_parameters``
(:238-266), ``get_mesh_folder_for_width`` (:269-286), ``get_watcher_points`` (:69-120),
``run_single_simulation`` (:123-192, one status row per point, exceptions become
``status='failed'`` rows), ``run_parameter_sweep`` (:289-536) with its artefacts
``sweep_metadata.json``, ``successful_runs.csv``, ``failed_runs.csv`` and one folder per run.

MI355X-first differences:
* the reference farms points to a ``multiprocessing.Pool`` whose workers each re-read
  ``mesh.msh``; here every rank (= one GPU, launched by ``torch.distributed.run``) takes the
  points ``combos[rank::world]`` of a width group, the group's mesh is built once by rank 0 and
  **broadcast** to the other ranks (RCCL over xGMI with the nccl backend, gloo on CPU), and a
  :class:`~heatflow_amd.driver.SimulationSession` keeps mesh + CSR pattern + mass matrix
  resident on the GPU across that rank's points (only kappa / fwhm dependent data is redone);
* there is no per-time-step communication; status rows are gathered on rank 0 at the end;
* ``modify_config_for_parameters`` deep-copies (the reference's shallow copy mutates the
  shared base dict, which is harmless there only because every call overwrites all three keys).
"""
from __future__ import annotations

import copy
import itertools
import json
import os
import time
from datetime import datetime

import numpy as np
import yaml

import sys

from .driver import (SimulationSession, build_pattern_blob, flush_mesh_writes, prepare_mesh, run_simulation_batch_impl,
                     run_simulation_impl)
from .geometry import build_stack, watcher_points as _watcher_points
from .hip_backend import HipError, NotConverged


def batch_failure(e, n_points):
    """What to do when a batched time loop raised ``e``.  Returns ``(retry_singly, text)``: a batch that did not converge
    as a whole (``NotConverged``: one hard column stops all of them) or that the configurations do not admit
    (``ValueError``) is re-run point by point, and ``text`` goes into every such row as ``batch_error`` and to stderr - a
    batch that always fails must show in the rows, not only as a slow sweep.  Anything else (a HIP error, an allocation
    failure, a bug) is not retried: the device or the session is suspect, the points become failed rows (reference:
    parameter_sweep.py:154-192, every exception becomes a ``status='failed'`` row, :516-518 ``failed_runs.csv``)."""
    text = f"{type(e).__name__}: {e}"
    retry = isinstance(e, (NotConverged, ValueError)) or (isinstance(e, HipError) and getattr(e, "code", None) == -4)
    print(f"heatflow_amd.parameter_sweep: batched time loop of {n_points} points failed ({text}); "
          + ("re-running them one by one" if retry else "not retried"), file=sys.stderr)
    return retry, text


def get_watcher_points(cfg):
    return _watcher_points(cfg)


def create_parameter_grid(fwhm_range, k_range, width_range, num_points):
    num_fwhm, num_k, num_width = num_points
    fwhm_vals = np.logspace(np.log10(fwhm_range[0]), np.log10(fwhm_range[1]), num_fwhm)
    k_vals = np.logspace(np.log10(k_range[0]), np.log10(k_range[1]), num_k)
    width_vals = np.linspace(width_range[0], width_range[1], num_width)
    combos = [{"fwhm": fwhm, "k": k, "width": width}
              for width in width_vals for fwhm, k in itertools.product(fwhm_vals, k_vals)]
    return combos, fwhm_vals, k_vals, width_vals


def modify_config_for_parameters(base_config, fwhm, k, width):
    config = copy.deepcopy(base_config)
    config["heating"]["fwhm"] = float(fwhm)
    config["mats"]["p_sample"]["k"] = float(k)
    config["mats"]["p_sample"]["z"] = float(width)
    return config


def get_mesh_folder_for_width(base_mesh_folder, width):
    width_str = f"{width:.3e}".replace("+", "").replace("-0", "-")
    return os.path.join(base_mesh_folder, f"width_{width_str}")


def run_name_for(fwhm, k, width):
    return f"fwhm_{fwhm:.2e}_k_{k:.2f}_width_{width:.2e}".replace("+", "").replace("-0", "-")


# -- process group helpers (work without torch.distributed as a world of 1) ------------------
def _dist():
    """torch.distributed if a process group is up.  A process that never imported it has none: a world of one must not
    pay for ``import torch`` (0.9 s, most of a small sweep's set-up)."""
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist
    return None


def local_device():
    """GPU of this rank: LOCAL_RANK, wrapped onto the GPUs that exist when a rehearsal (HEATFLOW_SWEEP_BACKEND=gloo)
    runs more ranks than the box has devices."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HEATFLOW_SWEEP_BACKEND") == "gloo":
        try:
            import torch
            return lr % max(torch.cuda.device_count(), 1)
        except ImportError:
            return 0
    return lr


def world_info():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d is not None else (0, 1)


def broadcast_mesh(arrays, tag_map=None, src=0):
    """Broadcast (coords f64 (n,2), tris i32 (ne,3), tags i32 (ne,)) and the material tag map
    {material name: cell tag} from ``src`` to all ranks; returns (arrays, tag_map).  The tag map
    belongs to the mesh (``mesh_cfg.yaml``: gmsh surface ids for a reference-written mesh, list
    positions for ours), so every rank must use rank ``src``'s copy and never guess it.
    With the nccl backend the arrays travel GPU-to-GPU (RCCL over xGMI), with gloo on the host.  They end on the host on
    every rank because the host logic needs them there: Dirichlet DOF location, nearest-node search of the watchers, the
    flux bands, and the C ABI's hf_set_mesh(_prebuilt), which takes the mesh arrays as host pointers.  The two big blobs of
    a sweep - connectivity tables and multigrid hierarchy - stay in device memory (:func:`broadcast_bytes`)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return arrays, tag_map
    import torch

    on_gpu = d.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    rank = d.get_rank()
    sizes = torch.tensor([len(arrays[0]), len(arrays[1])] if rank == src else [0, 0], dtype=torch.int64, device=dev)
    d.broadcast(sizes, src)
    n, ne = int(sizes[0]), int(sizes[1])
    shapes = [((n, 2), torch.float64), ((ne, 3), torch.int32), ((ne,), torch.int32)]
    out = []
    for k, (shape, dtype) in enumerate(shapes):
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arrays[k], dtype=np.float64 if k == 0 else np.int32)).to(dev)
        else:
            t = torch.empty(shape, dtype=dtype, device=dev)
        d.broadcast(t, src)
        out.append(t.cpu().numpy())
    box = [dict(tag_map) if rank == src else None]
    d.broadcast_object_list(box, src)
    return tuple(out), box[0]


class DeviceBlob:
    """A blob in device memory (the tensor RCCL delivered it into), handed to the solver library by address:
    ``hf_set_mesh_prebuilt`` / ``hf_amg_install`` read it from there, it never passes through host memory on this side."""

    def __init__(self, tensor):
        self.tensor = tensor                      # keeps the memory alive
        self.address = int(tensor.data_ptr())
        self.nbytes = int(tensor.numel() * tensor.element_size())

    def __len__(self):
        return self.nbytes


def broadcast_bytes(blob, src=0):
    """Broadcast a uint8 array (pattern blob, hierarchy blob) from ``src``.  With nccl it travels GPU-to-GPU (RCCL over
    xGMI) and STAYS on the device: every rank gets a :class:`DeviceBlob`; with gloo a numpy array on the host."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return blob
    import torch

    on_gpu = d.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    rank = d.get_rank()
    size = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    d.broadcast(size, src)
    t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)).to(dev) if rank == src else \
        torch.empty(int(size[0]), dtype=torch.uint8, device=dev)
    d.broadcast(t, src)
    if on_gpu:
        torch.cuda.synchronize()                 # the library reads the buffer on its own stream
        return DeviceBlob(t)
    return t.numpy()


def shared_pattern(arrays, device_id, session_factory, pattern_builder, timing=None, session=None, cfg=None, stack=None):
    """Rank 0 builds the connectivity tables of the mesh once and broadcasts them ("mesh/CSR pattern
    from rank 0"); every other session of every rank installs them.  With ``session`` (rank 0's first solver session,
    created without a pattern) the tables are the ones that session builds for its own resident problem of ``cfg`` -
    no extra solver context, no second installation on rank 0.  Without a builder (tests that inject their own
    session factory and no builder) nothing is shared and each session builds its own."""
    rank, _ = world_info()
    t0 = time.perf_counter()
    if pattern_builder is None:
        if session_factory is not None:
            return None
        if session is not None and rank == 0:
            from .driver import suppress_output
            with suppress_output(True):
                session.prepare(cfg, stack)
            pattern_builder = lambda *a: session.problem.backend.export_pattern()   # noqa: E731
        else:
            pattern_builder = build_pattern_blob
    blob = pattern_builder(*arrays, device_id) if rank == 0 else np.zeros(0, np.uint8)
    t1 = time.perf_counter()
    blob = broadcast_bytes(blob)
    if timing is not None:
        timing["pattern_build_s"] = t1 - t0
        timing["pattern_broadcast_s"] = time.perf_counter() - t1
        timing["pattern_bytes"] = int(len(blob))
    return blob


def make_session(arrays, tag_map, device_id, session_factory, pattern, hierarchy=None):
    if session_factory is not None:
        kw = {}
        if pattern is not None:
            kw["pattern"] = pattern
        if hierarchy is not None:
            kw["hierarchy"] = hierarchy
        return session_factory(*arrays, tag_map, **kw)
    return SimulationSession(*arrays, tag_map, device_id=device_id, pattern=pattern, hierarchy=hierarchy)


def _factory_takes_hierarchy(session_factory):
    if session_factory is None:
        return True
    import inspect
    try:
        params = inspect.signature(session_factory).parameters
    except (TypeError, ValueError):
        return False
    return "hierarchy" in params or any(p.kind == p.VAR_KEYWORD for p in params.values())


def shared_hierarchy(session, cfg, stack, session_factory, timing=None):
    """The multigrid hierarchy of the sweep's first configuration, built ONCE: rank 0's first session creates its resident
    problem for ``cfg`` (tables, matrices, host set-up of the hierarchy) and exports the hierarchy; the blob is broadcast
    like the connectivity tables (RCCL: device to device) and every other session of every rank installs it instead of
    repeating the set-up (the reference's analogue: each pool worker factorises for itself, run_with_diamond.py:389-394).
    Returns {"blob", "k"} for ``make_session(..., hierarchy=)``, or None when the sessions do not take one."""
    if not _factory_takes_hierarchy(session_factory) or getattr(session, "precond", 1) != 1:
        return None
    rank, world = world_info()
    t0 = time.perf_counter()
    share = None
    if rank == 0:
        from .driver import suppress_output
        with suppress_output(True):
            session.prepare(cfg, stack)
        share = session.export_hierarchy()
    t1 = time.perf_counter()
    if world > 1:
        d = _dist()
        box = [share["k"] if rank == 0 else None]
        d.broadcast_object_list(box, 0)
        blob = broadcast_bytes(share["blob"] if rank == 0 else np.zeros(0, np.uint8))
        share = {"blob": blob, "k": box[0]}
    if timing is not None:
        timing["hierarchy_build_s"] = t1 - t0
        timing["hierarchy_broadcast_s"] = time.perf_counter() - t1
        timing["hierarchy_bytes"] = int(len(share["blob"]))
    return share


def warm_device(device_id, session_factory=None):
    """Initialise the HIP runtime on this rank's GPU and load the library's kernels now (0.1-0.2 s the first time in a
    process), on a thread of its own - while rank 0 meshes and the others wait for the mesh - instead of inside the first
    solver session.  Returns the thread (join it before the first session is made), or None."""
    if session_factory is not None:
        return None
    import threading

    def init():
        from .hip_backend import HeatflowHIP
        try:
            HeatflowHIP(device_id).close()
        except Exception:            # noqa: BLE001 - the first real session reports what is wrong with the device
            pass

    th = threading.Thread(target=init, name="heatflow-warm-device", daemon=True)
    th.start()
    return th


_EMPTY_MESH = (np.zeros((0, 2)), np.zeros((0, 3), np.int32), np.zeros(0, np.int32))


def batch_groups(items, batch):
    """Split ``items`` into consecutive groups of 16, 8, 4 or 2 (at most ``batch``) and singles: the group sizes the
    batched time loop takes (hf_batch_begin)."""
    out, i = [], 0
    sizes = [s for s in (16, 8, 4, 2) if s <= max(int(batch), 1)]
    while i < len(items):
        left = len(items) - i
        take = next((s for s in sizes if s <= left), 1)
        out.append(items[i:i + take])
        i += take
    return out


def shard(items, rank, world):
    """Round-robin shard: item i -> rank i mod world (8 points per GPU for 64 points on 8 GPUs)."""
    return [(i, it) for i, it in enumerate(items) if i % world == rank]


def run_single_simulation(args, session=None, kind=None):
    """(combo, base_config, mesh_folder, output_dir, write_xdmf, suppress_print, run_id) -> status row."""
    combo, base_config, mesh_folder, output_dir, write_xdmf, suppress_print, run_id = args
    fwhm, k, width = combo["fwhm"], combo["k"], combo["width"]
    run_name = run_name_for(fwhm, k, width)
    run_output_dir = os.path.join(output_dir, run_name)
    config = modify_config_for_parameters(base_config, fwhm, k, width)
    row = {"run_id": run_id, "run_name": run_name, "fwhm": fwhm, "k": k, "width": width,
           "output_dir": run_output_dir, "runtime": 0.0, "status": "failed", "error": None}
    try:
        t0 = time.time()
        kind = kind or build_stack(config).kind
        res = run_simulation_impl(kind, config, mesh_folder, False, False, run_output_dir,
                                  get_watcher_points(config), write_xdmf, suppress_print, session=session)
        row.update(runtime=time.time() - t0, status="success",
                   pcg_iters_mean=float(np.mean(res["iters"])), pcg_iters_max=int(np.max(res["iters"])))
    except Exception as e:  # a failed / non-converged point is a row, never a silent result
        row.update(error=str(e))
    return row


def run_simulation_group(items, base_config, mesh_folder, output_dir, write_xdmf, suppress_print, session, kind):
    """``items`` = [(run_id, combo), ...]: two or more points advance together through the batched time loop
    (same status rows as run_single_simulation), run_no_diamond's per-step read-flux projection included; with XDMF
    output every point is run on its own.  A batch that fails as a whole: see :func:`batch_failure`."""
    def singly(batch_error=None):
        out = []
        for run_id, combo in items:
            row = run_single_simulation((combo, base_config, mesh_folder, output_dir, write_xdmf, suppress_print, run_id),
                                        session=session, kind=kind)
            if batch_error is not None:
                row["batch_error"] = batch_error
            out.append(row)
        return out

    if len(items) == 1 or write_xdmf:
        return singly()
    rows, cfgs, folders = [], [], []
    for run_id, combo in items:
        fwhm, k, width = combo["fwhm"], combo["k"], combo["width"]
        name = run_name_for(fwhm, k, width)
        folders.append(os.path.join(output_dir, name))
        cfgs.append(modify_config_for_parameters(base_config, fwhm, k, width))
        rows.append({"run_id": run_id, "run_name": name, "fwhm": fwhm, "k": k, "width": width, "output_dir": folders[-1],
                     "runtime": 0.0, "status": "failed", "error": None})
    try:
        t0 = time.time()
        results = run_simulation_batch_impl(kind, cfgs, folders, [get_watcher_points(c) for c in cfgs], session, suppress_print,
                                            read_flux=(kind == "no_diamond"))
        for row, res in zip(rows, results):
            row.update(runtime=(time.time() - t0) / len(items), status="success", pcg_iters_mean=float(np.mean(res["iters"])),
                       pcg_iters_max=int(np.max(res["iters"])), batch=len(items))
        return rows
    except Exception as e:  # noqa: BLE001 - classified by batch_failure, never swallowed
        retry, text = batch_failure(e, len(items))
        if retry:
            return singly(batch_error=text)
        for row in rows:
            row.update(error=f"batched time loop failed: {text}", batch_error=text)
        return rows


def run_parameter_sweep(base_config_path, output_dir, fwhm_range, k_range, width_range, num_points,
                        base_mesh_folder="meshes", write_xdmf=False, suppress_print=True, num_processes=None, *,
                        session_factory=None, device_id=None, pattern_builder=None, batch=1):
    """Sweep driver.  ``num_processes`` is accepted for signature parity; the degree of
    parallelism is the size of the torch.distributed world (one rank per GPU).
    ``session_factory(coords, tris, tags, tag_map)`` lets tests substitute the solver session.
    ``batch`` > 1: up to that many (16, 8, 4 or 2) consecutive points of a rank advance together through the batched
    time loop (points that share k share one operator, others get one operator per column)."""
    rank, world = world_info()
    with open(base_config_path) as f:
        base_config = yaml.safe_load(f)
    combos, fwhm_vals, k_vals, width_vals = create_parameter_grid(fwhm_range, k_range, width_range, num_points)
    if device_id is None:
        device_id = local_device()

    if rank == 0:
        os.makedirs(output_dir, exist_ok=True)
        meta = {
            "base_config": base_config_path, "fwhm_range": list(fwhm_range), "k_range": list(k_range),
            "width_range": list(width_range), "num_points": list(num_points), "fwhm_values": fwhm_vals.tolist(),
            "k_values": k_vals.tolist(), "width_values": width_vals.tolist(), "total_runs": len(combos),
            "num_processes": world, "timestamp": datetime.now().isoformat(),
            "watcher_points": {
                "description": "Temperature monitoring points positioned halfway through iridium coupler layers",
                "locations": {"pside": "Center of p-side iridium coupler (r=0)",
                              "oside": "Center of o-side iridium coupler (r=0)"},
                "coordinates": "Relative to mesh geometry, calculated for each parameter combination"},
        }
        with open(os.path.join(output_dir, "sweep_metadata.json"), "w") as f:
            json.dump(meta, f, indent=2)
        print(f"Starting parameter sweep with {len(combos)} total runs on {world} rank(s)")

    groups = {}
    for c in combos:
        groups.setdefault(c["width"], []).append(c)

    rows, done = [], 0
    for width, group in groups.items():
        mesh_folder = get_mesh_folder_for_width(base_mesh_folder, width)
        cfg0 = modify_config_for_parameters(base_config, group[0]["fwhm"], group[0]["k"], width)
        stack0 = build_stack(cfg0)
        arrays, tag_map = _EMPTY_MESH, None
        if rank == 0:
            have = os.path.exists(os.path.join(mesh_folder, "mesh.msh")) and \
                os.path.exists(os.path.join(mesh_folder, "mesh_cfg.yaml"))
            coords, tris, tags, tag_map = prepare_mesh(cfg0, mesh_folder, not have, stack0)
            arrays = (coords, tris, tags)
        arrays, tag_map = broadcast_mesh(arrays, tag_map)
        pattern = shared_pattern(arrays, device_id, session_factory, pattern_builder)
        # the group's first configuration: hierarchy built once (rank 0) and installed by the other ranks' sessions
        session = make_session(arrays, tag_map, device_id, session_factory, pattern) if rank == 0 else None
        hierarchy = shared_hierarchy(session, cfg0, stack0, session_factory) if world > 1 else None
        if session is None:
            session = make_session(arrays, tag_map, device_id, session_factory, pattern, hierarchy)
        try:
            mine = [(done + idx + 1, combo) for idx, combo in shard(group, rank, world)]
            for items in batch_groups(mine, batch):
                for row in run_simulation_group(items, base_config, mesh_folder, output_dir, write_xdmf, suppress_print,
                                                session, stack0.kind):
                    row["rank"] = rank
                    rows.append(row)
        finally:
            session.close()
        done += len(group)

    d = _dist()
    if d is not None and world > 1:
        gathered = [None] * world
        d.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
    rows.sort(key=lambda r: r["run_id"])
    results = [r for r in rows if r["status"] == "success"]
    failed = [r for r in rows if r["status"] != "success"]
    if rank == 0:
        _write_rows(os.path.join(output_dir, "successful_runs.csv"), results)
        _write_rows(os.path.join(output_dir, "failed_runs.csv"), failed)
        print(f"PARAMETER SWEEP COMPLETE: total {len(combos)}, successful {len(results)}, failed {len(failed)}")
    return results, failed


def _write_rows(path, rows):
    if not rows:
        return
    import csv

    keys = list(rows[0].keys())
    for r in rows[1:]:                      # rows of a sweep may carry extra fields (batch, batch_error): keep every column
        keys += [k for k in r.keys() if k not in keys]
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys, extrasaction="ignore")
        w.writeheader()
        for r in rows:
            w.writerow(r)


# -- kappa-only sweep on a shared mesh (reference sweep_test.py) ------------------------------
def get_k_values(k0=3.8, half_width=0.5, step=0.02, count=None):
    """sweep_test.py:47-52: np.round(np.arange(k0-0.5, k0+0.5+step, step), 4); ``count`` picks an
    evenly spaced grid of that many values over the same interval instead (64 for BASELINE C5)."""
    if count is not None:
        return np.round(np.linspace(k0 - half_width, k0 + half_width, int(count)), 4)
    return np.round(np.arange(k0 - half_width, k0 + half_width + step, step), 4)


def run_kappa_sweep(cfg, mesh_folder, k_values, output_dir, *, rebuild_mesh=False, session_factory=None,
                    device_id=None, exp_csv=None, concurrent=1, warmup_steps=0, on_ready=None, on_done=None,
                    timing=None, pattern_builder=None, batch=1):
    """k_sample sweep on one mesh: point i -> rank i mod world; the mesh (arrays + tag map) is broadcast
    once, each rank keeps it resident and only re-values A per point.  Returns rows [{k, rmse, runtime,...}]
    (rmse of the normalised o-side watcher against the experiment, sweep_test.py:76-93).

    ``batch`` > 1 advances up to that many (16, 8, 4 or 2) of a rank's points together through the batched time loop
    (hf_batch_*: one operator per column, shared frozen multigrid hierarchy): at stock mesh sizes a single run is
    launch-bound, a batch of 8 costs little more than one run.
    ``concurrent`` > 1 runs that many of a rank's points (or batches) at once, each on its own solver context
    (own HIP stream, own copy of the mesh): at stock mesh sizes one point cannot fill an MI355X
    (its kernels are latency-bound), so overlapping points raises the per-GPU throughput.

    Measurement hooks (bench.py): ``warmup_steps`` > 0 makes every session run that many untimed steps of the
    first point before the point loop (mesh, pattern, matrices and multigrid levels are then resident);
    ``on_ready()`` / ``on_done()`` are called right before / after the point loop (barrier + clock);
    ``timing`` (dict) receives the wall times of the phases on this rank."""
    from .analysis_utils import calculate_rmse

    rank, world = world_info()
    if device_id is None:
        device_id = local_device()
    stack = build_stack(cfg)
    t_phase = time.perf_counter()
    arrays, tag_map = _EMPTY_MESH, None
    warming = warm_device(device_id, session_factory)    # overlaps rank 0's meshing / the wait for the mesh
    if rank == 0:
        coords, tris, tags, tag_map = prepare_mesh(cfg, mesh_folder, rebuild_mesh, stack, defer_write=True)
        arrays = (coords, tris, tags)
        os.makedirs(output_dir, exist_ok=True)
    if timing is not None:
        timing["mesh_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()
    arrays, tag_map = broadcast_mesh(arrays, tag_map)
    if timing is not None:
        timing["broadcast_s"] = time.perf_counter() - t_phase
        timing["t_mesh_received"] = time.perf_counter()
        timing["n_dof"] = int(len(arrays[0]))
    if warming is not None:
        warming.join()
    # the first configuration of the sweep: rank 0's first session builds its connectivity tables and its multigrid hierarchy;
    # both are broadcast and installed by every other session of every rank
    cfg_first = copy.deepcopy(cfg)
    if len(k_values):
        cfg_first["mats"]["p_sample"]["k"] = float(list(k_values)[0])
    stack_first = build_stack(cfg_first)
    first = pattern_builder is None and session_factory is None      # product path: the tables come from rank 0's own session
    session = make_session(arrays, tag_map, device_id, session_factory, None) if (rank == 0 and first) else None
    pattern = shared_pattern(arrays, device_id, session_factory, pattern_builder, timing, session=session, cfg=cfg_first, stack=stack_first)
    t_phase = time.perf_counter()
    if session is None and rank == 0:
        session = make_session(arrays, tag_map, device_id, session_factory, pattern)
    hierarchy = shared_hierarchy(session, cfg_first, stack_first, session_factory, timing)
    if session is None:
        session = make_session(arrays, tag_map, device_id, session_factory, pattern, hierarchy)
    exp = None
    if exp_csv is not None:
        exp = np.genfromtxt(exp_csv, delimiter=",", names=True)
    # folder per point as sweep_test.py:63 names it ("3.80"); more digits only if two points would share a folder
    digits = 2 if len({f"{k:.2f}" for k in k_values}) == len(list(k_values)) else 4

    def rmse_against_experiment(c, res):
        ps, os_ = res["watchers"]["pside"], res["watchers"]["oside"]
        span = ps.max() - ps.min()
        sim_o = (os_ - os_[0]) / span
        ic = float(c["heating"]["ic_temp"])
        exp_o = exp["oside"] - exp["oside"][0] + ic
        exp_o = (exp_o - exp_o[0]) / (exp["temp"].max() - exp["temp"].min())
        return calculate_rmse(exp["time"], exp_o, res["times"], sim_o)

    def one_point(k, sess):
        c = copy.deepcopy(cfg)
        c["mats"]["p_sample"]["k"] = float(k)
        outdir = os.path.join(output_dir, f"{k:.{digits}f}")
        t0 = time.time()
        row = {"k": float(k), "rmse": float("nan"), "runtime": 0.0, "status": "failed", "error": None, "rank": rank}
        try:
            res = run_simulation_impl(stack.kind, c, mesh_folder, False, False, outdir, get_watcher_points(c),
                                      False, True, session=sess, read_flux=False)
            row.update(status="success", runtime=time.time() - t0, pcg_iters_mean=float(np.mean(res["iters"])))
            if exp is not None:
                row["rmse"] = rmse_against_experiment(c, res)
        except Exception as e:
            row.update(error=str(e))
        return row

    def many_points(ks, sess):
        """len(ks) in (2, 4, 8, 16): one batched loop; every point is re-run on its own if the batch fails."""
        cfgs = []
        for k in ks:
            c = copy.deepcopy(cfg)
            c["mats"]["p_sample"]["k"] = float(k)
            cfgs.append(c)
        t0 = time.time()
        try:
            results = run_simulation_batch_impl(stack.kind, cfgs, [os.path.join(output_dir, f"{k:.{digits}f}") for k in ks],
                                                [get_watcher_points(c) for c in cfgs], sess, True)
        except Exception as e:  # noqa: BLE001 - classified by batch_failure, never swallowed
            retry, text = batch_failure(e, len(ks))
            if retry:
                return [dict(one_point(k, sess), batch_error=text) for k in ks]
            return [{"k": float(k), "rmse": float("nan"), "runtime": 0.0, "status": "failed", "rank": rank,
                     "error": f"batched time loop failed: {text}", "batch_error": text} for k in ks]
        rows_ = []
        for k, c, res in zip(ks, cfgs, results):
            row = {"k": float(k), "rmse": float("nan"), "runtime": (time.time() - t0) / len(ks), "status": "success", "error": None,
                   "rank": rank, "pcg_iters_mean": float(np.mean(res["iters"])), "batch": len(ks)}
            if exp is not None:
                row["rmse"] = rmse_against_experiment(c, res)
            rows_.append(row)
        return rows_

    mine = batch_groups([k for _, k in shard(list(k_values), rank, world)], batch)
    sessions = [session]
    rows = []
    try:
        n_sess = min(concurrent, len(mine)) if concurrent > 1 else 1
        for _ in range(max(n_sess, 1) - 1):
            sessions.append(make_session(arrays, tag_map, device_id, session_factory, pattern, hierarchy))
        if timing is not None:
            timing["session_s"] = time.perf_counter() - t_phase
        if warmup_steps > 0 and len(k_values):
            t_phase = time.perf_counter()
            cw = copy.deepcopy(cfg)
            dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
            cw["timing"]["num_steps"] = int(warmup_steps)
            cw["timing"]["t_final"] = dt0 * int(warmup_steps)          # same dt: the resident problem is reused
            cw["mats"]["p_sample"]["k"] = float(list(k_values)[0])
            from .driver import suppress_output
            with suppress_output(True):
                if len(sessions) > 1:                    # each session on its own thread: installs and first launches overlap
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(max_workers=len(sessions)) as pool:
                        list(pool.map(lambda s: s.run(cw, build_stack(cw), get_watcher_points(cw)), sessions))
                else:
                    session.run(cw, build_stack(cw), get_watcher_points(cw))
            if timing is not None:
                timing["warmup_s"] = time.perf_counter() - t_phase
        if timing is not None:
            timing["mesh_received_to_ready_s"] = time.perf_counter() - timing["t_mesh_received"]
            del timing["t_mesh_received"]
        if on_ready is not None:
            on_ready()
        t_phase = time.perf_counter()
        if len(sessions) > 1:
            import queue
            from concurrent.futures import ThreadPoolExecutor

            free = queue.Queue()
            for sess in sessions:
                free.put(sess)

            def task(ks):
                sess = free.get()
                try:
                    return [one_point(ks[0], sess)] if len(ks) == 1 else many_points(ks, sess)
                finally:
                    free.put(sess)

            with ThreadPoolExecutor(max_workers=len(sessions)) as pool:
                rows = [r for part in pool.map(task, mine) for r in part]
        else:
            rows = [r for ks in mine for r in ([one_point(ks[0], session)] if len(ks) == 1 else many_points(ks, session))]
        if on_done is not None:
            on_done()
        if timing is not None:
            timing["points_s"] = time.perf_counter() - t_phase
            timing["points_here"] = sum(len(ks) for ks in mine)
            timing["batches"] = [len(ks) for ks in mine]
            timing["sessions"] = len(sessions)
    finally:
        t_phase = time.perf_counter()
        for sess in sessions:
            sess.close()
        flush_mesh_writes()
        if timing is not None:
            timing["close_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()
    d = _dist()
    if d is not None and world > 1:
        gathered = [None] * world
        d.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
    rows.sort(key=lambda r: r["k"])
    if rank == 0:
        _write_rows(os.path.join(output_dir, "rmse_summary.csv"), rows)
    if timing is not None:
        timing["gather_write_s"] = time.perf_counter() - t_phase
    return rows


def main(argv=None):
    import argparse

    p = argparse.ArgumentParser(description="Parameter sweep (one point per GPU under torch.distributed.run)")
    p.add_argument("--config", required=True)
    p.add_argument("--output-dir", required=True)
    p.add_argument("--fwhm-range", nargs=2, type=float, required=True)
    p.add_argument("--k-range", nargs=2, type=float, required=True)
    p.add_argument("--width-range", nargs=2, type=float, required=True)
    p.add_argument("--num-points", nargs=3, type=int, required=True)
    p.add_argument("--mesh-folder", default="meshes")
    p.add_argument("--write-xdmf", action="store_true")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--batch", type=int, default=16,
                   help="points of a rank advanced together by the batched time loop (16, 8, 4, 2; 1 = one run per point)")
    a = p.parse_args(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        # HEATFLOW_SWEEP_BACKEND=gloo: rehearsal with more ranks than GPUs (ranks share devices, the mesh travels on the host)
        use_gpu = torch.cuda.is_available() and os.environ.get("HEATFLOW_SWEEP_BACKEND") != "gloo"
        if use_gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if use_gpu else "gloo")
    run_parameter_sweep(a.config, a.output_dir, tuple(a.fwhm_range), tuple(a.k_range), tuple(a.width_range),
                        tuple(a.num_points), a.mesh_folder, a.write_xdmf, not a.verbose, batch=a.batch)
    d = _dist()
    if d is not None:
        d.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
