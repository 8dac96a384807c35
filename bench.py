#!/usr/bin/env python3
"""Headline benchmark: DOF-updates/s of the backward-Euler time loop on geballe_with_diamond
refined to ~1M DOF (BASELINE.json configs[2] / BASELINE.md C3).

    python bench.py --gpus N --steps K --warmup W [--workload c3|sweep64]

* ``--gpus N`` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
  (``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a fresh child, before torch or HIP
  is touched here) and forwards the child's single JSON line.  Under an outside launcher (the driver's
  ``torch.distributed.run``) WORLD_SIZE is set and the ranks run directly.  ``n_gpus`` in the JSON is the
  world size the process group actually has.
* workload ``c3`` (default, the configuration the metric is quoted on): a "step" is one time step of the hot
  path: b = M u^n, lifting, set_bc, PCG solve (run_with_diamond.py:469-481), all inputs resident in HBM.  The W
  warm-up steps are the first W steps of the simulation (steps 0-3 carry no heating yet and converge in zero
  iterations), the K timed steps follow them.  N > 1: every rank solves its own sweep point
  (kappa_sample = 3.8 + 0.02*rank, the sweep_test.py grid) on the same mesh, which rank 0 builds and
  broadcasts over RCCL together with its tag map; no data-path collective.  value = all ranks' DOF-updates /
  max-over-ranks time ("weak" scaling).
* workload ``sweep64`` (BASELINE C5): 64 kappa_sample values (parameter_sweep.get_k_values(count=64)) at stock
  mesh size, point i -> rank i mod world, batches of up to 16 points per time loop (hf_batch_*), 2 loops in flight per rank; K = time steps per point (default: the
  config's 100), W = untimed steps every solver session runs first.  value = 64*n*K / wall of the point loop
  (max over ranks), "strong" scaling (the 64 points are fixed).  The same sweep is also run as a side
  measurement of the default workload (``config.sweep64``; ``--sweep-points 0`` skips it).
* roofline: the dominant kernel is the PCG iteration head k_spmv<9> (CSR SpMV A z with the direction
  update p <- z + beta p, Ap <- A z + beta Ap fused).  achieved = algorithmic bytes per launch
  (SpMV 12*nnz + 20*n of SURVEY.md section 8d, plus 24*n for reading the old p and Ap and writing p)
  / its average duration INSIDE the loop (kernel-attached HIP events on the solver's stream on the launches
  of extra steps right after the timed region; agrees with rocprofv3's in-loop figure, profiles/).  The
  back-to-back figure (100 launches) and the bytes the compressed column format really moves are beside it.
  ``roofline.traffic`` = HBM bytes per launch of that kernel from the PMC counters, measured in the same invocation: two
  child runs of this script (``--pmc-child``) under ``rocprofv3 --pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE`` (separate passes,
  counters only), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; ``roofline.traffic_source`` says so, or names the kept file it fell
  back to (``--traffic file``) when the profiler is not available.
  At 1M DOF the iteration's working set sits in the 256 MiB Infinity Cache; ``roofline.hbm_resident`` repeats
  the measurement on a mesh whose matrix is far beyond it (``--hbm-scale``, 16M DOF).  ``roofline.assembly``
  holds the element kernel's variants.
* cpu_baseline: the oracle (reference algorithm: assemble once, sparse LU once, two
  triangular solves per step; SciPy SuperLU, 1 thread) on the same mesh, rank 0, N = 1 only.
* C5's CPU baseline (``cpu_baseline`` of workload sweep64, ``config.sweep64.cpu_farm_baseline`` of the default one):
  ALL 64 points through the oracle on a pool of single-threaded processes, one per core up to 64, as the reference farms
  its points over ``mp.Pool(processes=mp.cpu_count())`` (parameter_sweep.py:389-390, 423-446); the round-2 figure (8 points
  on 8 cores) is kept beside it (``cpu_farm_8_points``).  Both run in child processes before this one touches the GPU.
* roofline of workload sweep64 (``config.sweep64.roofline`` of the default one): the batched loop's iteration head
  kb_spmv_lds<9> (16 columns, affine operator family), in-loop events on one batch of 16 at stock size, and the same kernel
  on the 1M-DOF mesh, where the batch's vectors (66 MB each) are far beyond the Infinity Cache (``hbm_resident``).
"""
import os

os.environ.setdefault("NCCL_DEBUG", "WARN")   # keep RCCL banners out of stdout: one JSON line only
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")      # the reference pins its workers to 1 thread (parameter_sweep.py:46-53)

import argparse
import json
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
TARGET_DOF = 1.0e6
MESH_SCALE = 0.43          # all `mesh:` values x 0.43 -> 1.04 M nodes (within +-5 % of 1.0e6)
HBM_SCALE = 0.1075         # -> 16 M nodes: matrix 1.3 GB, vectors 128 MB each, nothing stays in the 256 MiB Infinity Cache
SWEEP_POINTS = 64          # BASELINE C5
SWEEP_BATCH = 16           # points per batched time loop (hf_batch_*): columns of one multi-vector PCG; a rank with fewer points takes 8 / 4 / 2
SWEEP_CONCURRENT = 2       # time loops in flight per rank (64 points on one GPU, round 3: batches of 16, 1 / 2 in flight = 1.09 / 1.20e9 DOF-updates/s;
                           # batches of 8: 0.90 / 1.08e9; round 2, batches of 8: 7.1 / 9.0e8; unbatched, 6 in flight: 4.5e8)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 60; sweep64: steps per point, default 100)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c3", "sweep64"], default="c3")
    ap.add_argument("--device-warmup-s", type=float, default=3.0,
                    help="seconds of back-to-back SpMV launches before the warm-up steps: the first process on a box that has been "
                         "idle sees one-off stalls of several ms in its first seconds (5-8 %% on a 0.1 s timed region), which are "
                         "not a property of the time loop (0 = off)")
    ap.add_argument("--scale", type=float, default=MESH_SCALE, help="factor on every mats.*.mesh (0.43 -> ~1.04M DOF)")
    ap.add_argument("--cpu-steps", type=int, default=60, help="steps of the CPU baseline sample (0 = skip): with the LU factorisation ~10-15 s of host work")
    ap.add_argument("--profile-steps", type=int, default=4, help="extra steps with in-situ SpMV event timing")
    ap.add_argument("--precond", choices=["amg", "jacobi"], default="amg",
                    help="PCG preconditioner of the timed run: smoothed-aggregation V-cycle (default) or plain Jacobi")
    ap.add_argument("--jacobi-steps", type=int, default=10,
                    help="with --precond amg: also time this many Jacobi-PCG steps for the record (0 = skip)")
    ap.add_argument("--sweep-points", type=int, default=SWEEP_POINTS,
                    help="c3 workload: also run this many kappa points of the C5 sweep as a side measurement (0 = skip)")
    ap.add_argument("--sweep-concurrent", type=int, default=SWEEP_CONCURRENT)
    ap.add_argument("--sweep-batch", type=int, default=SWEEP_BATCH,
                    help="sweep points advanced together by the batched time loop (16, 8, 4, 2; 1 = one run per point)")
    ap.add_argument("--hbm-scale", type=float, default=HBM_SCALE,
                    help="mesh factor of the HBM-resident roofline point (N = 1 only; 0 = skip)")
    ap.add_argument("--cpu-farm-points", type=int, default=SWEEP_POINTS,
                    help="C5 CPU baseline (N = 1 only): this many sweep points through the reference algorithm on a pool of single-threaded "
                         "processes, one per usable core up to this many (mirrors the reference's mp.Pool(processes=mp.cpu_count()), "
                         "parameter_sweep.py:389-390, 423-446); 0 = skip.  Runs before anything touches the GPU")
    ap.add_argument("--cpu-farm-procs", type=int, default=0, help="processes of that pool (0 = min(points, usable cores))")
    ap.add_argument("--batch-roofline", type=int, default=1, help="N = 1: roofline of the batched iteration head (0 = skip)")
    ap.add_argument("--traffic", choices=["live", "file", "none"], default="live",
                    help="roofline.traffic (HBM bytes of the dominant kernel per launch, PMC): live = measured now by two child runs of this "
                         "script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (N = 1; +15-20 s), falling back to the figures kept "
                         "under profiles/ when the profiler is not available; file = those figures only")
    ap.add_argument("--pmc-child", choices=["c3", "batch"], default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-farm-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="form the process group, report its size and exit: checks the launcher plumbing (no GPU work)")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 100 if args.workload == "sweep64" else 60
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus and --steps must be positive, --warmup non-negative")
    return args


def spawn_ranks(args, argv):
    """--gpus N without an outside launcher: start the N ranks as a fresh child (this process has not
    imported torch or touched HIP), forward its JSON line, return its exit code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode == 0 and lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
        return 0
    sys.stderr.write(f"bench.py: the {args.gpus}-rank child run failed (exit code {p.returncode})\n")
    return p.returncode or 1


def build_problem_inputs(scale):
    import yaml
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes
    from heatflow_amd.mesh import Mesh

    with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg = scale_mesh_sizes(cfg, scale)
    stack = build_stack(cfg)
    mesh = Mesh("mesh.msh", stack.bounds, stack.materials).build_mesh()
    return cfg, stack, mesh


def make_problem(cfg, stack, coords, tris, tags, material_tags, k_sample, device_id, precond, pattern=None):
    from heatflow_amd.bc import P1Space, RowDirichletBC
    from heatflow_amd.heating import HeatingCurve
    from heatflow_amd.solver import HeatProblem

    ic = float(cfg["heating"]["ic_temp"])
    heat = HeatingCurve(os.path.join(ROOT, cfg["heating"]["file"]), ic, float(cfg["heating"]["fwhm"]))
    V = P1Space(coords)
    bcs = [RowDirichletBC(V, "left", value=ic), RowDirichletBC(V, "right", value=ic), RowDirichletBC(V, "top", value=ic),
           RowDirichletBC(V, "x", coord=stack.heated_z, length=abs(stack.r_sample) * 2, center=0.0, value=heat.gaussian)]
    tag_to_k = {material_tags[m.name]: m.properties["k"] for m in stack.materials}
    tag_to_rc = {material_tags[m.name]: m.properties["rho_cv"] for m in stack.materials}
    if k_sample is not None:
        tag_to_k[material_tags["p_sample"]] = float(k_sample)
    dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    return HeatProblem(coords, tris, tags, tag_to_k, tag_to_rc, dt, bcs, ic, device_id=device_id, precond=precond, pattern=pattern)


def cpu_baseline(cfg, mesh, n_sample_steps, first_step):
    """Reference algorithm on the host: factor once, then `n_sample_steps` steps (timed)."""
    from oracle import heat_oracle as ho

    t0 = time.perf_counter()
    res = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags,
                                     os.path.join(ROOT, cfg["heating"]["file"]), num_steps=0)
    sol = res["solver"]
    t_asm = time.perf_counter() - t0
    sol.factor()
    # advance to the same point of the simulation as the GPU's timed region, then time the sample
    for s in range(first_step):
        sol.step((s + 1) * res["dt"])
    t0 = time.perf_counter()
    for s in range(first_step, first_step + n_sample_steps):
        sol.step((s + 1) * res["dt"])
    t_steps = time.perf_counter() - t0
    n = len(mesh.coords)
    return {
        "value": n * n_sample_steps / t_steps, "unit": "DOF-updates/s", "cores": 1, "kind": "port",
        "sample": (f"reference algorithm restated in oracle/heat_oracle.py (SciPy SuperLU, not FEniCS/MUMPS; 1 thread of "
                   f"{os.cpu_count()} host cores): same {n}-node mesh, steps {first_step}..{first_step + n_sample_steps - 1}; "
                   f"s/step = {t_steps / n_sample_steps:.4f}; one-off costs not in value: numpy assembly {t_asm:.1f} s, "
                   f"LU factorisation {sol.t_factor:.1f} s"),
        "s_per_step": t_steps / n_sample_steps, "factor_s": sol.t_factor,
    }


def _farm_point(job):
    """One sweep point on one host core through the oracle (child of the farm process; GPU-free)."""
    cfg, k, arrays, mtags = job
    import copy
    from oracle import heat_oracle as ho
    c = copy.deepcopy(cfg)
    c["mats"]["p_sample"]["k"] = float(k)
    t0 = time.perf_counter()
    ho.run_reference_algorithm(c, arrays[0], arrays[1], arrays[2], mtags, os.path.join(ROOT, c["heating"]["file"]),
                               num_steps=int(c["timing"]["num_steps"]), watcher_nodes=[0])
    return time.perf_counter() - t0


def cpu_farm_worker(n_points, steps_per_point, max_procs=0):
    """Body of `bench.py --cpu-farm-worker` (a process of its own, started before the parent touches the GPU): mesh once,
    then `n_points` kappa_sample points of BASELINE C5 on a pool of one process per point (at most the box's cores)."""
    import multiprocessing as mp
    import yaml
    from heatflow_amd import parameter_sweep as ps
    from heatflow_amd.geometry import build_stack
    from heatflow_amd.mesh import Mesh

    for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):   # parameter_sweep.py:46-53
        os.environ[v] = "1"
    with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
        cfg = yaml.safe_load(f)
    dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    cfg["timing"]["num_steps"] = int(steps_per_point)
    cfg["timing"]["t_final"] = dt0 * int(steps_per_point)
    stack = build_stack(cfg)
    mesh = Mesh("mesh.msh", stack.bounds, stack.materials).build_mesh()
    arrays, mtags = (mesh.coords, mesh.tris, mesh.tags), mesh.material_tags
    ks = ps.get_k_values(count=64)[:n_points]
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(n_points, max_procs if max_procs > 0 else usable))
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        per_point = pool.map(_farm_point, [(cfg, k, arrays, mtags) for k in ks], chunksize=1)
    wall = time.perf_counter() - t0
    n = len(mesh.coords)
    print(json.dumps({
        "value": n_points * n * steps_per_point / wall, "unit": "DOF-updates/s", "cores": procs, "kind": "port",
        "sample": (f"{n_points} of the 64 kappa_sample points of BASELINE C5 (stock mesh, {n} DOF, {steps_per_point} steps each) through "
                   f"oracle/heat_oracle.py (SciPy SuperLU, not FEniCS/MUMPS) on a pool of {procs} single-threaded processes "
                   f"({usable} cores usable by this process, os.cpu_count() = {os.cpu_count()}), each assembling and factorising for its "
                   f"point as the reference's pool workers do (parameter_sweep.py:123-192, pool of mp.cpu_count() processes :389-390); "
                   f"mesh built once and handed over; process start-up included"),
        "wall_s": wall, "per_point_s_mean": float(sum(per_point) / len(per_point)), "points": n_points,
        "usable_cores": usable, "cpu_count": os.cpu_count()}))
    return 0


def run_cpu_farm(args, steps_per_point=100, points=None, procs=None):
    """Start the farm as a child process and return its JSON (an error entry when it fails: the baseline is a report, not a gate)."""
    points = args.cpu_farm_points if points is None else points
    procs = args.cpu_farm_procs if procs is None else procs
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-farm-worker", "--cpu-farm-points", str(points), "--cpu-farm-procs", str(procs),
           "--steps", str(steps_per_point)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        return json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception as e:          # noqa: BLE001 - reported in the JSON line
        return {"error": f"{type(e).__name__}: {e}"}


SOLVER_TEXT = ("PCG preconditioned by a smoothed-aggregation multigrid V(1,1) cycle with damped-Jacobi smoothing; operator A, mass matrix, "
               "all vectors, residual, dot products and the stopping rule in f64; the operators that act only inside the preconditioner "
               "(transfer operators P / R, fused legs, dense inverse of the coarsest level) are stored in f32 and applied to f64 vectors "
               "with f64 accumulation")
PRECISION_TEXT = ("preconditioner operators below the fine level f32 (values only; vectors and accumulation f64); fine operator, "
                  "vectors, residual, stopping rule f64: the converged answer is the f64 PCG answer (HEATFLOW_AMG_F32=0 stores them in f64)")


def attach_farms(cfg_out, gpu_value, farm, farm8):
    """CPU farm baselines of C5 and the GPU / farm ratios, each with the core count it ran on."""
    if farm and "value" in farm:
        cfg_out["gpu_over_cpu_farm"] = gpu_value / farm["value"]
        cfg_out["gpu_over_cpu_farm_cores"] = farm.get("cores")
    if farm8:
        cfg_out["cpu_farm_8_points"] = farm8
        if "value" in farm8:
            cfg_out["gpu_over_cpu_farm_8_points"] = gpu_value / farm8["value"]


def batch_head_roofline(dev_index, nv, profile_steps=8, traffic_mode="file"):
    """Roofline of the batched loop's dominant kernel, the iteration head kb_spmv_lds<9, nv, affine> (the C5 sweep: nv
    kappa_sample points as the columns of one multi-vector PCG).  In-loop: kernel-attached HIP events on its launches
    inside the time loop of one batch (hf_set_profile), on the stock mesh (the C5 size; the batch's working set of
    ~135 MB sits in the 256 MiB Infinity Cache) and on the 1M-DOF mesh of C3 (vectors of 66 MB each: HBM-resident).
    Algorithmic bytes per launch, SURVEY 8d conventions (f64 values, i32 indices, every array once per pass):
    (4 + 16)*nnz [column index + the two shared value arrays of the affine family] + 4*n [row pointers]
    + 40*n*nv [per row and column: z 8, Ap and p read-modify-write 32]."""
    import copy
    import yaml
    from heatflow_amd.driver import SimulationSession
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes, watcher_points
    from heatflow_amd.mesh import Mesh
    from heatflow_amd import parameter_sweep as ps

    def one(scale, steps):
        with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
            cfg = scale_mesh_sizes(yaml.safe_load(f), scale)
        cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
        dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
        cfg["timing"]["num_steps"] = int(steps)
        cfg["timing"]["t_final"] = dt0 * int(steps)
        stack = build_stack(cfg)
        mesh = Mesh("mesh.msh", stack.bounds, stack.materials).build_mesh()
        sess = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, device_id=dev_index)
        try:
            cfgs = []
            for k in ps.get_k_values(count=64)[:nv]:
                c = copy.deepcopy(cfg)
                c["mats"]["p_sample"]["k"] = float(k)
                cfgs.append(c)
            stacks = [build_stack(c) for c in cfgs]
            from heatflow_amd.driver import suppress_output
            with suppress_output(True):
                sess.run_batch(cfgs, stacks, watcher_points(cfgs[0]))          # set-up + warm caches
                be = sess.problem.backend
                be.set_profile(True)
                res = sess.run_batch(cfgs, stacks, watcher_points(cfgs[0]))
                ms_sum, cnt = be.get_profile()
                be.set_profile(False)
            n, nnz = be.n, be.nnz
            byts = 20 * nnz + 4 * n + 40 * n * nv
            us = 1e3 * ms_sum / cnt if cnt else None
            return {"workload": f"one batch of {nv} kappa_sample points, cfgs/geballe_with_diamond.yaml, every mats.*.mesh x {scale}, steps 0..{steps - 1}",
                    "n_dof": n, "nnz": nnz, "nv": nv, "bytes_per_launch": byts, "us_per_launch_in_loop_events": us, "in_loop_launches": int(cnt),
                    "achieved": byts / (us * 1e-6) / 1e9 if us else None, "frac": byts / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us else None,
                    "pcg_iters_per_step_mean": float(sum(float(r["iters"].mean()) for r in res) / len(res)),
                    "batch_ms_per_step": 1e3 * res[0]["loop_time"] * nv / steps,
                    "working_set_bytes_vectors": 9 * 8 * n * nv, "cache_resident": bool(20 * nnz + 9 * 8 * n * nv < 256 * 2**20)}
        finally:
            sess.close()

    stock = one(1.0, 30)
    big = one(MESH_SCALE, 10)
    traffic, traffic_note = None, "not measured"
    if traffic_mode == "live":
        traffic, traffic_note = live_traffic("batch", rf"kb_spmv_lds<9, {nv}, 2>", ["--sweep-batch", str(nv)])
        if traffic is None:
            traffic_note = f"live measurement not available ({traffic_note}); "
    if traffic is None and traffic_mode != "none":
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_batch_latest.json")) as f:
                pmc = json.load(f)
            if pmc["n"] == stock["n_dof"] and pmc["nnz"] == stock["nnz"] and pmc["nv"] == nv:
                traffic = pmc["kernels"][f"kb_spmv_lds<9,nv{nv},op2>"]["hbm_bytes"]
                traffic_note = (traffic_note if traffic_note.startswith("live") else "") + \
                    "figure kept under profiles/pmc_traffic_batch_latest.json (same matrix and nv; scripts/profile_batch.sh)"
        except (OSError, KeyError, ValueError):
            pass
    return {"bound": "hbm", "achieved": stock["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": stock["frac"], "traffic": traffic,
            "kernel": f"kb_spmv_lds<9, {nv}, affine> (batched PCG iteration head: CSR SpMV on {nv} interleaved columns with the direction update fused; "
                      "chunk operands staged in LDS, 16-bit column positions)",
            "bytes_per_launch": stock["bytes_per_launch"], "us_per_launch": stock["us_per_launch_in_loop_events"],
            "formula": "(4 + 16)*nnz + 4*n + 40*n*nv",
            "timing": "in-loop: kernel-attached HIP events on the kb_spmv_lds<9> launches of one batch's time loop",
            "traffic_source": traffic_note,
            "stock_size": stock, "hbm_resident": big}


def pmc_child(args):
    """Body of `bench.py --pmc-child c3|batch`, run under `rocprofv3 --pmc <counter>`: a few steps of the measured loop, nothing else."""
    if args.pmc_child == "c3":
        cfg, stack, mesh = build_problem_inputs(args.scale)
        prob = make_problem(cfg, stack, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, None, 0, 1)
        for bc in prob.bcs:
            bc.update(0.0)
        prob.run(9, time_varying=[prob.bcs[3]], first_step=0)          # steps 0-4 carry no heating, 5-8 iterate
        print(json.dumps({"n": prob.backend.n, "nnz": prob.backend.nnz}))
        prob.close()
    else:
        import copy
        import yaml
        from heatflow_amd import parameter_sweep as ps
        from heatflow_amd.driver import SimulationSession, suppress_output
        from heatflow_amd.geometry import build_stack, watcher_points
        from heatflow_amd.mesh import Mesh
        with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
            cfg = yaml.safe_load(f)
        cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
        dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
        cfg["timing"]["num_steps"], cfg["timing"]["t_final"] = 12, dt0 * 12
        stack = build_stack(cfg)
        mesh = Mesh("mesh.msh", stack.bounds, stack.materials).build_mesh()
        sess = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags)
        cfgs = []
        for k in ps.get_k_values(count=64)[:args.sweep_batch]:
            c = copy.deepcopy(cfg)
            c["mats"]["p_sample"]["k"] = float(k)
            cfgs.append(c)
        with suppress_output(True):
            sess.run_batch(cfgs, [build_stack(c) for c in cfgs], watcher_points(cfgs[0]))
        print(json.dumps({"n": sess.problem.backend.n, "nnz": sess.problem.backend.nnz}))
        sess.close()
    return 0


def live_traffic(kind, kernel_regex, extra_args=()):
    """HBM bytes per launch of one kernel, measured now: this script is run twice as a child under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes, counters only: MI355X_MICROARCH.md, HBM section); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
    (gfx950 counts 64 B per 128-B request in FETCH_SIZE), each counter's 90th percentile over the launches (launches that return at
    their first instruction after convergence read nothing).  Returns (bytes, note) or (None, why not)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        return None, "rocprofv3 not found"
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")):
        return None, "this run is itself under rocprofv3 (a nested profiler would inherit its preloaded tool library)"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="hf_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "run", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", kind] + list(extra_args)
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180, cwd=tmp, env=dict(os.environ, TMPDIR=tmp))
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} timed out"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (exit {p.returncode})"
            v = []
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if re.search(kernel_regex, r["Kernel_Name"]):
                        v.append(float(r["Counter_Value"]))
            if not v:
                return None, f"no launch of {kernel_regex} in the {counter} pass"
            v.sort()
            vals[counter] = v[max(0, int(0.9 * len(v)) - 1)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "measured in this run: child runs of bench.py under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, (2*FETCH_SIZE + WRITE_SIZE)*1024, p90 over the launches"


class SideGuard:
    """Deadline for a side measurement that runs collectives AFTER the headline is measured (N > 1): if it has not been
    disarmed in time, rank 0 writes the headline's line with the failure noted in the side measurement's place and every
    rank ends its process (exit code 0: the line is complete) - before the process group's own watchdog aborts the job."""

    def __init__(self, out, emit, seconds):
        import threading
        self.lock, self.state = threading.Lock(), "armed"
        self.out, self.emit, self.seconds = out, emit, seconds
        self.timer = threading.Timer(seconds, self._expired)
        self.timer.daemon = True
        self.timer.start()

    def _expired(self):
        with self.lock:
            if self.state != "armed":
                return
            self.state = "expired"
        sys.stderr.write(f"bench.py: side measurement not finished after {self.seconds:.0f} s: reporting the headline without it\n")
        if self.out is not None:
            self.out["config"]["sweep64"] = {"error": f"not finished after {self.seconds:.0f} s (collective or rank stuck); headline unaffected"}
            self.out.setdefault("cpu_baseline", None)
            self.emit(self.out)
        sys.stderr.flush()
        os._exit(0)

    def disarm(self):
        """True if the guard was still armed (the caller goes on), False if it has fired."""
        with self.lock:
            if self.state != "armed":
                return False
            self.state = "disarmed"
        self.timer.cancel()
        return True


class Ranks:
    """The process group of this run (a world of one without torch.distributed)."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist, self.torch = None, None
        self.backend, self.dev_index = "nccl", self.local_rank
        if self.world > 1 or os.environ.get("HEATFLOW_BENCH_FORCE_DIST") == "1":   # the latter: rehearse the RCCL path on 1 GPU
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)   # a forced world of one started without a launcher
            import torch                      # torch first: its bundled HIP runtime must be the one both sides use
            import torch.distributed as dist
            # HEATFLOW_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
            # (ranks share devices, the mesh travels through host tensors); the measured runs use RCCL
            self.backend = os.environ.get("HEATFLOW_BENCH_BACKEND", "nccl")
            ndev = torch.cuda.device_count()
            if self.backend == "nccl" and ndev < int(os.environ.get("LOCAL_WORLD_SIZE", self.world)):
                raise SystemExit(f"bench.py: {self.world} ranks but {ndev} GPU(s) visible: RCCL needs one GPU per rank "
                                 "(HEATFLOW_BENCH_BACKEND=gloo rehearses the code path on fewer GPUs)")
            self.dev_index = self.local_rank % max(ndev, 1)
            # a collective that never completes (a rank that died or raised between two of them) must end the run, not hang it
            import datetime
            limit = datetime.timedelta(seconds=int(os.environ.get("HEATFLOW_BENCH_DIST_TIMEOUT_S", "240")))
            if self.backend == "nccl":
                torch.cuda.set_device(self.dev_index)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev_index), timeout=limit)
            else:
                if ndev > 0:
                    torch.cuda.set_device(self.dev_index)
                dist.init_process_group(self.backend, timeout=limit)
            self.dist, self.torch = dist, torch
            self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def device(self):
        return self.torch.device("cuda", self.dev_index) if self.backend == "nccl" else self.torch.device("cpu")

    def barrier_sync(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.torch.cuda.is_available():
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def run_sweep64(ranks, n_points, steps_per_point, warmup_steps, concurrent, batch=1):
    """BASELINE C5: `n_points` kappa_sample values on geballe_with_diamond at stock mesh size, point i ->
    rank i mod world, mesh + tag map broadcast once, `concurrent` points in flight per rank
    (reference parameter_sweep.py:423-446, sweep_test.py:47-115).  Returns the measurement (rank 0) or None."""
    import shutil
    import tempfile

    import yaml
    from heatflow_amd import parameter_sweep as ps

    with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
        cfg = yaml.safe_load(f)
    dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    cfg["timing"]["num_steps"] = int(steps_per_point)
    cfg["timing"]["t_final"] = dt0 * int(steps_per_point)        # steps 0..K-1 of the configured run, same dt
    ks = ps.get_k_values(count=n_points)
    tmp = tempfile.mkdtemp(prefix=f"hf_sweep64_r{ranks.rank}_")
    clock, timing = {}, {}

    def on_ready():
        ranks.barrier_sync()
        clock["t0"] = time.perf_counter()

    def on_done():
        clock["mine"] = time.perf_counter() - clock["t0"]
        ranks.barrier_sync()
        clock["all"] = time.perf_counter() - clock["t0"]

    try:
        t_all = time.perf_counter()
        rows = ps.run_kappa_sweep(cfg, os.path.join(tmp, "mesh"), ks, os.path.join(tmp, "out"), rebuild_mesh=True,
                                  device_id=ranks.dev_index, concurrent=concurrent, warmup_steps=warmup_steps,
                                  on_ready=on_ready, on_done=on_done, timing=timing, batch=batch)
        t_all = time.perf_counter() - t_all
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    wall = ranks.max_over_ranks(clock["all"])
    whole = ranks.max_over_ranks(t_all)          # mesh + broadcasts + connectivity tables + multigrid set-up + warm-up + point loop, slowest rank
    setup_max = ranks.max_over_ranks(t_all - clock["all"])
    phase_keys = ("mesh_s", "broadcast_s", "pattern_build_s", "pattern_broadcast_s", "hierarchy_build_s", "hierarchy_broadcast_s",
                  "session_s", "warmup_s", "mesh_received_to_ready_s", "points_s", "close_s", "gather_write_s")
    mine = {k: timing.get(k) for k in phase_keys}
    by_rank = [mine]
    if ranks.dist is not None:
        by_rank = [None] * ranks.world
        ranks.dist.all_gather_object(by_rank, mine)
    if ranks.rank != 0:
        return None
    bad = [r for r in rows if r["status"] != "success"]
    if bad:
        raise SystemExit(f"bench.py: {len(bad)} sweep point(s) failed, first: {bad[0]['error']}")
    import numpy as np
    n_dof = timing.get("n_dof")                  # rank 0 built the mesh
    if not n_dof:
        with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
            n_dof = _stock_dof(yaml.safe_load(f))
    return {
        "workload": (f"{n_points}-point kappa_sample sweep [{ks[0]}..{ks[-1]}] on cfgs/geballe_with_diamond.yaml, stock mesh "
                     f"(BASELINE C5), point i -> rank i mod {ranks.world}, batches of up to {batch} points per time loop, "
                     f"{concurrent} loops in flight per rank, steps 0..{steps_per_point - 1} per point"),
        "points": n_points, "n_dof": n_dof, "steps_per_point": steps_per_point, "concurrent_per_rank": concurrent,
        "batch": batch, "rank0_batches": timing.get("batches"),
        "iteration_head_bytes_per_dof_update": {
            "what": ("algorithmic bytes of the PCG iteration head per row and column at nnz = 7 n: one run per point "
                     "12*nnz + 44*n; batched with the affine operator family (two shared value arrays, shared indices) "
                     "(4 + 16)*nnz/nv + 44*n"),
            "one_run_per_point": 12 * 7 + 44, "batched": (4 + 16) * 7 / max(batch, 1) + 44 if batch > 1 else 12 * 7 + 44},
        "wall_s": wall, "value": n_points * n_dof * steps_per_point / wall, "unit": "DOF-updates/s",
        "points_per_s": n_points / wall, "pcg_iters_per_step_mean": float(np.mean([r["pcg_iters_mean"] for r in rows])),
        "rank0_phases_s": by_rank[0], "other_ranks_phases_s": by_rank[1:],
        "n_dof_check": timing.get("n_dof"),
        "whole_call_s": whole, "setup_s_max_over_ranks": setup_max,
        "value_whole_call": n_points * n_dof * steps_per_point / whole,
        "value_whole_call_note": ("the same DOF-updates over the whole call of the slowest rank: mesh, broadcasts, connectivity tables, "
                                  "multigrid set-up and the untimed warm-up steps included - what a user of the sweep waits for"),
    }


_STOCK_DOF = {}


def _stock_dof(cfg):
    """Node count of the stock mesh (rank 0 built it a moment ago; meshing again is 0.2 s)."""
    if "n" not in _STOCK_DOF:
        from heatflow_amd.geometry import build_stack
        from heatflow_amd.mesh import Mesh
        st = build_stack(cfg)
        _STOCK_DOF["n"] = len(Mesh("mesh.msh", st.bounds, st.materials).build_mesh().coords)
    return _STOCK_DOF["n"]


def kernel_rooflines(be, prob, hb, heated, first_step, profile_steps):
    """Durations of the dominant kernels on the live operator, HIP events on the solver's own stream.
    In-loop: kernel-attached start/stop events (hipExtLaunchKernelGGL) on the k_spmv<9> launches of
    `profile_steps` extra time steps.  Back to back: 100 launches in a row (what a cache-resident working
    set flatters).  Returns a dict of microseconds."""
    out = {"spmv9_in_loop": None}
    if profile_steps > 0:
        be.set_profile(True)
        prob.run(profile_steps, time_varying=heated, first_step=first_step)
        ms_sum, cnt = be.get_profile()
        be.set_profile(False)
        if cnt > 0:
            out["spmv9_in_loop"] = 1e3 * ms_sum / cnt
            out["spmv9_in_loop_launches"] = int(cnt)
    for nm, k in (("spmv9_back_to_back", hb.K_PCG_SPMV), ("update_back_to_back", hb.K_PCG_UPDATE),
                  ("plain_spmv_back_to_back", hb.K_SPMV), ("stream_read", hb.K_STREAM_READ)):
        out[nm] = 1e3 * be.time_kernel(k, 100)
    return out


def assembly_rooflines(be, prob, hb, n, ne, nnz):
    """Element kernel variants: 20 back-to-back launches, and single launches with an SpMV pass in between
    (what one launch costs when the previous kernel was something else, as in a kappa sweep)."""
    import numpy as np
    asm_bytes = 16 * ne + 16 * n + 16 * nnz           # tri+tag, coords, each CSR value of M and A written once
    res = {"bytes_per_launch": asm_bytes, "formula": "16*n_e + 16*n + 2*8*nnz (SURVEY 8d, M and A)"}
    for name, mode in (("lds_atomic", hb.ASM_LDS_ATOMIC), ("lds_colored", hb.ASM_LDS_COLORED)) + \
            ((("row_gather", hb.ASM_ROW_GATHER),) if hasattr(hb, "ASM_ROW_GATHER") else ()):
        be.assemble(prob.dt, mode)
        b2b = 1e3 * be.time_kernel(hb.K_ASSEMBLE, 20)
        singles = []
        for _ in range(7):
            be.assemble(prob.dt, mode)
            be.time_kernel(hb.K_SPMV, 2)
            singles.append(1e3 * be.time_kernel(hb.K_ASSEMBLE, 1))
        us1 = float(np.median(singles))
        res[name] = {"us_back_to_back": b2b, "us_single_launch_median": us1,
                     "achieved": asm_bytes / (us1 * 1e-6) / 1e9, "frac": asm_bytes / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "frac_back_to_back": asm_bytes / (b2b * 1e-6) / 1e9 / HBM_PEAK_GBS}
    res["default_mode"] = {0: "lds_atomic", 1: "lds_colored", 3: "row_gather"}.get(prob.assembly_mode, str(prob.assembly_mode))
    be.assemble(prob.dt, prob.assembly_mode)          # restore the operator of the run
    return res


def hbm_resident_point(scale, dev_index, steps):
    """The same kernels on a mesh whose operator is far larger than the 256 MiB Infinity Cache:
    Jacobi-PCG steps (no multigrid set-up needed) with in-loop events, plus back-to-back figures."""
    import numpy as np
    from heatflow_amd import hip_backend as hb

    t0 = time.perf_counter()
    cfg, stack, mesh = build_problem_inputs(scale)
    t_mesh = time.perf_counter() - t0
    t0 = time.perf_counter()
    prob = make_problem(cfg, stack, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, None, dev_index, 0)
    t_setup = time.perf_counter() - t0
    be = prob.backend
    try:
        n, ne, nnz = be.n, be.n_e, be.nnz
        for bc in prob.bcs:
            bc.update(0.0)
        heated = [prob.bcs[3]]
        prob.run(5, time_varying=heated, first_step=0)              # reach the heated steps
        be.set_profile(True)
        _, _, iters = prob.run(steps, time_varying=heated, first_step=5)
        ms_sum, cnt = be.get_profile()
        be.set_profile(False)
        us_loop = 1e3 * ms_sum / cnt if cnt else None
        us = {nm: 1e3 * be.time_kernel(k, 20) for nm, k in
              (("spmv9_back_to_back", hb.K_PCG_SPMV), ("update_back_to_back", hb.K_PCG_UPDATE),
               ("plain_spmv_back_to_back", hb.K_SPMV), ("stream_read", hb.K_STREAM_READ))}
        asm = assembly_rooflines(be, prob, hb, n, ne, nnz)
        b9 = 12 * nnz + 44 * n
        t9 = us_loop if us_loop else us["spmv9_back_to_back"]
        return {
            "workload": f"cfgs/geballe_with_diamond.yaml, every mats.*.mesh x {scale}", "n_dof": n, "nnz": nnz,
            "matrix_bytes": 12 * nnz, "kernel": "k_spmv<9>", "bytes_per_launch": b9,
            "us_per_launch_in_loop_events": us_loop, "in_loop_launches": int(cnt), "us_back_to_back": us,
            "achieved": b9 / (t9 * 1e-6) / 1e9, "frac": b9 / (t9 * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "plain_spmv_frac": (12 * nnz + 20 * n) / (us["plain_spmv_back_to_back"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "stream_read_GBs": 12 * nnz / (us["stream_read"] * 1e-6) / 1e9,
            "assembly": asm, "jacobi_pcg_iters_per_step_mean": float(np.mean(iters)),
            "host_setup_s": {"mesher": t_mesh, "set_mesh_assemble": t_setup},
        }
    finally:
        prob.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.cpu_farm_worker:
        return cpu_farm_worker(args.cpu_farm_points, args.steps or 100, args.cpu_farm_procs)
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    # C5's CPU baseline is a pool of processes: it runs to completion here, before this process touches the GPU
    farm = farm8 = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_farm_points > 0 and not args.rendezvous_only
            and (args.workload == "sweep64" or (args.sweep_points > 0 and args.cpu_steps > 0))):
        farm_steps = args.steps if args.workload == "sweep64" else 100
        farm = run_cpu_farm(args, farm_steps)
        if args.cpu_farm_points > 8:           # the round-2 figure beside it: 8 points on 8 cores
            farm8 = run_cpu_farm(args, farm_steps, points=8, procs=8)

    # stdout carries exactly ONE line (the JSON): everything else - RCCL's version banner, library
    # chatter - is sent to stderr by pointing fd 1 at fd 2 until the result is written
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    ranks = Ranks(args)
    world, rank, dev_index = ranks.world, ranks.rank, ranks.dev_index
    if world != args.gpus and rank == 0:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}\n")

    def emit(out):
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    if args.rendezvous_only:
        total = ranks.max_over_ranks(float(rank)) + 1
        if rank == 0:
            emit({"rendezvous_only": True, "n_gpus": world, "max_rank_plus_1": int(total), "backend": ranks.backend if ranks.dist else None})
        ranks.close()
        return 0

    import numpy as np
    import yaml
    from heatflow_amd import hip_backend as hb

    common = {"unit": "DOF-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
              "vs_baseline": None, "dtype": "f64", "data": "synthetic"}

    if args.workload == "sweep64":
        sw = run_sweep64(ranks, SWEEP_POINTS, args.steps, args.warmup, args.sweep_concurrent, args.sweep_batch)
        if rank == 0:
            roof = batch_head_roofline(dev_index, args.sweep_batch, traffic_mode=args.traffic) if (world == 1 and args.batch_roofline and args.sweep_batch > 1) else None
            out = dict(common, metric="DOF-updates/s (timesteps/s x nDOF) on geballe_with_diamond", value=sw["value"],
                       ms_per_step=1e3 * sw["wall_s"] / args.steps, scaling="strong",
                       config={"workload": sw["workload"], "solver": SOLVER_TEXT, "precision": PRECISION_TEXT,
                               **{k: v for k, v in sw.items() if k not in ("workload", "value", "unit")}},
                       roofline=roof, cpu_baseline=farm)
            attach_farms(out["config"], sw["value"], farm, farm8)
            emit(out)
        ranks.close()
        return 0

    # ---- mesh: built once (rank 0) and broadcast over RCCL with its tag map, as a sweep shares it
    from heatflow_amd import parameter_sweep as ps
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes
    arrays, mtags, mesh = ps._EMPTY_MESH, None, None
    if rank == 0:
        cfg, stack, mesh = build_problem_inputs(args.scale)
        arrays, mtags = (mesh.coords, mesh.tris, mesh.tags), mesh.material_tags
    else:
        with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
            cfg = scale_mesh_sizes(yaml.safe_load(f), args.scale)
        stack = build_stack(cfg)
    (coords, tris, tags), mtags = ps.broadcast_mesh(arrays, mtags)
    # N > 1: the connectivity tables (CSR pattern, column lists, assembly lists) are built by rank 0 only and broadcast too
    share = {}
    pattern = ps.shared_pattern((coords, tris, tags), dev_index, None, None, share) if world > 1 else None

    k_sample = None if world == 1 else 3.8 + 0.02 * rank
    precond = 1 if args.precond == "amg" else 0
    prob = make_problem(cfg, stack, coords, tris, tags, mtags, k_sample, dev_index, precond, pattern)
    setup_s = {"set_mesh": prob.mesh_seconds, "whole_problem": prob.setup_seconds, **share}
    be = prob.backend
    n, nnz, ne = be.n, be.nnz, be.n_e
    for bc in prob.bcs:
        bc.update(0.0)
    heated = [prob.bcs[3]]

    # ---- device warm-up (scratch operands only, the state is untouched), warm-up steps (untimed), then exactly K timed steps
    if args.device_warmup_s > 0:
        tw = time.perf_counter()
        while time.perf_counter() - tw < args.device_warmup_s:
            be.time_kernel(hb.K_SPMV, 500)
    if args.warmup > 0:
        prob.run(args.warmup, time_varying=heated, first_step=0)
    ranks.barrier_sync()
    t0 = time.perf_counter()
    _, _, iters = prob.run(args.steps, time_varying=heated, first_step=args.warmup)
    ranks.barrier_sync()
    elapsed = time.perf_counter() - t0
    gpu_ms = be.last_gpu_ms()
    elapsed = ranks.max_over_ranks(elapsed)

    # ---- dominant kernel k_spmv<9>: vals 8 + colidx 4 per nnz; per row rowptr 4, z 8 (gathered operand),
    # Ap 8+8 and p 8+8 (read-modify-write by the direction recurrence) = 12*nnz + 44*n (SURVEY 8d formula bytes)
    k_us = kernel_rooflines(be, prob, hb, heated, args.warmup + args.steps, args.profile_steps)
    spmv_bytes = 12 * nnz + 44 * n
    spmv_us = k_us["spmv9_in_loop"] if k_us["spmv9_in_loop"] else k_us["spmv9_back_to_back"]
    achieved = spmv_bytes / (spmv_us * 1e-6) / 1e9
    stream_gbs = 12 * nnz / (k_us["stream_read"] * 1e-6) / 1e9
    # HBM traffic of that kernel from the PMC passes kept under profiles/ (rocprofv3 cannot wrap itself):
    # only quoted when it was collected on exactly this matrix
    traffic, traffic_note = None, "not measured"
    if args.traffic == "live" and world == 1:
        traffic, traffic_note = live_traffic("c3", r"k_spmv<9, ", ["--scale", str(args.scale)])
        if traffic is None:
            traffic_note = f"live measurement not available ({traffic_note}); "
    if traffic is None and args.traffic != "none":
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
                pmc = json.load(f)
            if pmc["n"] == n and pmc["nnz"] == nnz:
                traffic = pmc["kernels"]["k_spmv<9>"]["hbm_bytes"]
                traffic_note = (traffic_note if traffic_note.startswith("live") else "") + \
                    "figure kept under profiles/pmc_traffic_latest.json (same matrix; rocprofv3 --pmc passes of scripts/profile_round.sh)"
        except (OSError, KeyError, ValueError):
            pass
    moved = traffic if traffic else 10 * nnz + 44 * n
    amg_info = be.amg_info() if precond == 1 else None
    asm_roof = None
    if world == 1:
        be.set_precond(0, False)          # the element kernel is timed without a multigrid set-up behind every re-assembly
        asm_roof = assembly_rooflines(be, prob, hb, n, ne, nnz)

    # ---- for the record: the plain Jacobi-PCG loop (north-star solver) on the same steps
    jacobi = None
    if precond == 1 and args.jacobi_steps > 0 and world == 1:
        be.assemble(prob.dt, prob.assembly_mode)
        prob.set_state(float(cfg["heating"]["ic_temp"]))
        if args.warmup > 0:
            prob.run(args.warmup, time_varying=heated, first_step=0)
        tj = time.perf_counter()
        _, _, itj = prob.run(args.jacobi_steps, time_varying=heated, first_step=args.warmup)
        tj = time.perf_counter() - tj
        jacobi = {"steps": args.jacobi_steps, "ms_per_step": 1e3 * tj / args.jacobi_steps,
                  "pcg_iters_per_step_mean": float(np.mean(itj)), "value": n * args.jacobi_steps / tj}
    prob.close()

    if rank == 0:
        out = dict(common, metric="DOF-updates/s (timesteps/s x nDOF) on geballe_with_diamond",
                   value=world * n * args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps, scaling="weak")
        out["config"] = {
            "workload": f"cfgs/geballe_with_diamond.yaml, every mats.*.mesh x {args.scale} (BASELINE C3, ~1M DOF), "
                        f"steps {args.warmup}..{args.warmup + args.steps - 1} of 100, dt=7.5e-8 s",
            "n_dof": n, "n_elem": ne, "nnz": nnz, "n_dirichlet": be.n_bc,
            "solver": (SOLVER_TEXT if precond else "Jacobi-PCG, everything in f64"),
            "pcg_rtol": prob.rtol, "pcg_iters_per_step_mean": float(np.mean(iters)), "pcg_iters_per_step_max": int(np.max(iters)),
            "points": "1 sweep point per GPU (kappa_sample = 3.8 + 0.02*rank)" if world > 1 else "1 run",
            "gpu_ms_per_step_events": gpu_ms / args.steps, "rank0_setup_s": setup_s}
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": "k_spmv<9> (PCG iteration head: CSR SpMV with the direction update p, Ap fused)",
            "bytes_per_launch": spmv_bytes, "us_per_launch": spmv_us,
            "timing": ("in-loop: kernel-attached HIP events on the k_spmv<9> launches of the time loop"
                       if k_us["spmv9_in_loop"] else "back-to-back launches (no in-loop sample)"),
            "cache_resident": bool(12 * nnz + 64 * n < 256 * 2**20),
            "note": ("at this size the iteration's working set sits in the 256 MiB Infinity Cache: frac is against the HBM "
                     "peak but is not an HBM measurement - see hbm_resident"),
            "us_back_to_back": {k: v for k, v in k_us.items() if k.endswith("back_to_back")},
            "frac_back_to_back": spmv_bytes / (k_us["spmv9_back_to_back"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "bytes_moved_per_launch": moved,
            "frac_bytes_moved": moved / (spmv_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "bytes_moved_note": "PMC traffic when collected on this matrix, else 10*nnz + 44*n (16-bit column stream)",
            "traffic_source": traffic_note,
            "measured_stream_read": {"GB/s": stream_gbs, "us": k_us["stream_read"], "bytes": 12 * nnz,
                                     "what": "16-byte-load read of the operator's values + column indices on this box",
                                     "frac_of_it": achieved / stream_gbs},
            "assembly": asm_roof, "hbm_resident": None}
        if precond == 1:
            # the whole multigrid-PCG iteration (iteration head, update, V-cycle: 10 launches at C3): HBM bytes = the per-launch PMC
            # means kept under profiles/ (scripts/pmc_summary.py; only quoted for exactly this matrix), over the time the loop
            # spends per iteration - right-hand side, start vector, first residual and the extra cycle of every step included
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
                    pmc_it = json.load(f)
                if pmc_it["n"] == n and pmc_it["nnz"] == nnz and pmc_it.get("multigrid_iteration"):
                    it_bytes = float(pmc_it["multigrid_iteration"]["bytes"])
                    it_us = 1e3 * (1e3 * elapsed / args.steps) / max(float(np.mean(iters)), 1.0)
                    out["roofline"]["multigrid_pcg_iteration"] = {
                        "bytes": it_bytes, "bytes_source": "profiles/pmc_traffic_latest.json (rocprofv3 --pmc passes of scripts/profile_round.sh, same matrix): mean HBM bytes per launch, summed over the launches of one iteration",
                        "us_per_iteration_in_loop": it_us, "achieved": it_bytes / (it_us * 1e-6) / 1e9,
                        "frac": it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "note": "time per iteration = time per step / iterations per step: includes the step's right-hand side, start vector, first residual and first cycle; the launches of one iteration alone take 119-122 us under rocprofv3 (profiles/r03_c3_iteration_breakdown.txt)"}
            except (OSError, KeyError, ValueError):
                pass
        if jacobi is not None:
            # the whole Jacobi-PCG iteration = iteration head + update, 12*nnz + 84*n bytes, over the time the
            # loop really spends per iteration (steps incl. right-hand side and start vector / iterations)
            it_us = 1e3 * jacobi["ms_per_step"] / max(jacobi["pcg_iters_per_step_mean"], 1.0)
            it_bytes = 12 * nnz + 84 * n
            out["roofline"]["jacobi_pcg_iteration"] = {
                "bytes": it_bytes, "formula": "12*nnz + 84*n (SURVEY 8d: SpMV + x, r, p read-modify-write + Ap, dinv reads)",
                "us_per_iteration_in_loop": it_us, "achieved": it_bytes / (it_us * 1e-6) / 1e9,
                "frac": it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "us_kernels_back_to_back": k_us["spmv9_back_to_back"] + k_us["update_back_to_back"]}
        if precond == 1:
            out["config"]["amg"] = dict(amg_info, precision=PRECISION_TEXT)
        if jacobi is not None:
            out["config"]["jacobi_pcg"] = jacobi
    # ---- side measurements: the C5 sweep (every N), the HBM-resident roofline point (N = 1)
    # (the headline's timing and every collective it needs are done: a failure of a side measurement is reported in its place,
    # it does not take the line with it.  N > 1: the sweep's collectives have a deadline of their own - a rank that hangs in one
    # would otherwise end in the process group's watchdog abort and take the finished headline with it)
    sweep = None
    if args.sweep_points > 0:
        guard = SideGuard(out if rank == 0 else None, emit, float(os.environ.get("HEATFLOW_BENCH_SIDE_TIMEOUT_S", "150"))) if world > 1 else None
        try:
            sweep = run_sweep64(ranks, args.sweep_points, 100, max(1, args.warmup), args.sweep_concurrent, args.sweep_batch)
        except BaseException as e:          # noqa: BLE001 - SystemExit of a failed point included
            sys.stderr.write(f"bench.py: the C5 side measurement failed on rank {rank}: {type(e).__name__}: {e}\n")
            sweep = {"error": f"{type(e).__name__}: {e}"} if rank == 0 else None
        if guard is not None and not guard.disarm():
            return 0                          # the guard has written the line (and is ending the process)
    hbm = hbm_resident_point(args.hbm_scale, dev_index, 3) if (world == 1 and args.hbm_scale > 0) else None
    batch_roof = batch_head_roofline(dev_index, args.sweep_batch, traffic_mode="file" if args.traffic == "live" else args.traffic) if (world == 1 and sweep is not None and "error" not in sweep and args.batch_roofline and args.sweep_batch > 1) else None

    if rank == 0:
        out["roofline"]["hbm_resident"] = hbm
        if sweep is not None and "error" in sweep:
            out["config"]["sweep64"] = sweep
        elif sweep is not None:
            if farm:
                sweep["cpu_farm_baseline"] = farm
            attach_farms(sweep, sweep["value"], farm, farm8)
            if batch_roof is not None:
                sweep["roofline"] = batch_roof
            out["config"]["sweep64"] = sweep
        if world == 1 and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, mesh, args.cpu_steps, args.warmup)
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if jacobi is not None:
                out["config"]["jacobi_pcg"]["gpu_over_cpu"] = jacobi["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        emit(out)
    try:
        ranks.close()
    except Exception as e:                  # noqa: BLE001 - the line is out; a process group broken by a failed side measurement must not change the exit code
        sys.stderr.write(f"bench.py: closing the process group failed: {e}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
