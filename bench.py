#!/usr/bin/env python3
"""Headline benchmark: DOF-updates/s of the backward-Euler time loop on geballe_with_diamond
refined to ~1M DOF (BASELINE.json configs[2] / BASELINE.md C3).

    python bench.py --gpus N --steps K --warmup W

* A "step" is one time step of the hot path: b = M u^n, lifting, set_bc, Jacobi-PCG solve
  (run_with_diamond.py:469-481), all inputs resident in HBM.  The W warm-up steps are the
  first W steps of the simulation (steps 0-3 carry no heating yet and converge in zero
  iterations), the K timed steps follow them.
* N > 1 (launched by torch.distributed.run, one rank per GPU): every rank solves its own
  sweep point (kappa_sample = 3.8 + 0.02*rank, the sweep_test.py grid) on the same mesh,
  which rank 0 builds and broadcasts over RCCL; no data-path collective.  value = all ranks'
  DOF-updates / max-over-ranks time ("weak" scaling).
* roofline: the dominant kernel is the PCG iteration head k_spmv<9> (CSR SpMV A z with the direction
  update p <- z + beta p, Ap <- A z + beta Ap fused).  achieved = algorithmic bytes per launch
  (SpMV 12*nnz + 20*n of SURVEY.md section 8d, plus 24*n for reading the old p and Ap and writing p)
  / its average duration over 100 launches on the live matrix, HIP events on the solver's stream,
  right after the timed region (agrees with rocprofv3's in-loop average for the kernel); the in-loop
  kernel-attached event timing, which also contains the gap to the previous kernel, is reported too.
* cpu_baseline: the oracle (reference algorithm: assemble once, sparse LU once, two
  triangular solves per step; SciPy SuperLU, 1 thread) on the same mesh, rank 0, N = 1 only.
"""
import os

os.environ.setdefault("NCCL_DEBUG", "WARN")   # keep RCCL banners out of stdout: one JSON line only
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")      # the reference pins its workers to 1 thread (parameter_sweep.py:46-53)

import argparse
import json
import sys
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
TARGET_DOF = 1.0e6
MESH_SCALE = 0.43          # all `mesh:` values x 0.43 -> 1.04 M nodes (within +-5 % of 1.0e6)


def build_problem_inputs(scale):
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes
    from heatflow_amd.mesh import Mesh

    with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg = scale_mesh_sizes(cfg, scale)
    stack = build_stack(cfg)
    mesh = Mesh("mesh.msh", stack.bounds, stack.materials).build_mesh()
    return cfg, stack, mesh


def make_problem(cfg, stack, coords, tris, tags, material_tags, k_sample, device_id, precond):
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
    return HeatProblem(coords, tris, tags, tag_to_k, tag_to_rc, dt, bcs, ic, device_id=device_id, precond=precond)


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


def main():
    # stdout carries exactly ONE line (the JSON): everything else - RCCL's version banner, library
    # chatter - is sent to stderr by pointing fd 1 at fd 2 until the result is written
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=float, default=MESH_SCALE, help="factor on every mats.*.mesh (0.43 -> ~1.04M DOF)")
    ap.add_argument("--cpu-steps", type=int, default=60, help="steps of the CPU baseline sample (0 = skip): with the LU factorisation ~10-15 s of host work")
    ap.add_argument("--profile-steps", type=int, default=4, help="extra steps with in-situ SpMV event timing")
    ap.add_argument("--precond", choices=["amg", "jacobi"], default="amg",
                    help="PCG preconditioner of the timed run: smoothed-aggregation V-cycle (default) or plain Jacobi")
    ap.add_argument("--jacobi-steps", type=int, default=10,
                    help="with --precond amg: also time this many Jacobi-PCG steps for the record (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend, dev_index = "nccl", local_rank
    if world > 1 or os.environ.get("HEATFLOW_BENCH_FORCE_DIST") == "1":   # the latter: rehearse the RCCL path on 1 GPU
        import torch                      # torch first: its bundled HIP runtime must be the one both sides use
        import torch.distributed as dist
        # HEATFLOW_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
        # (ranks share devices, the mesh travels through host tensors); the measured runs use RCCL
        backend = os.environ.get("HEATFLOW_BENCH_BACKEND", "nccl")
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    def barrier_sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # ---- mesh: built once (rank 0) and broadcast over RCCL, as a sweep shares it
    if rank == 0:
        cfg, stack, mesh = build_problem_inputs(args.scale)
        coords, tris, tags, mtags = mesh.coords, mesh.tris, mesh.tags, mesh.material_tags
    if dist is not None:
        import torch
        from heatflow_amd.geometry import build_stack, scale_mesh_sizes
        if rank != 0:
            with open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")) as f:
                cfg = scale_mesh_sizes(yaml.safe_load(f), args.scale)
            stack = build_stack(cfg)
            mtags = {m.name: k + 1 for k, m in enumerate(stack.materials)}
        dev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
        sizes = torch.tensor([len(coords), len(tris)] if rank == 0 else [0, 0], dtype=torch.int64, device=dev)
        dist.broadcast(sizes, 0)
        n_, ne_ = int(sizes[0]), int(sizes[1])
        tc = torch.from_numpy(coords).to(dev) if rank == 0 else torch.empty((n_, 2), dtype=torch.float64, device=dev)
        tt = torch.from_numpy(tris).to(dev) if rank == 0 else torch.empty((ne_, 3), dtype=torch.int32, device=dev)
        tg = torch.from_numpy(tags).to(dev) if rank == 0 else torch.empty((ne_,), dtype=torch.int32, device=dev)
        for t in (tc, tt, tg):
            dist.broadcast(t, 0)
        coords, tris, tags = tc.cpu().numpy(), tt.cpu().numpy(), tg.cpu().numpy()
        del tc, tt, tg

    k_sample = None if world == 1 else 3.8 + 0.02 * rank
    precond = 1 if args.precond == "amg" else 0
    prob = make_problem(cfg, stack, coords, tris, tags, mtags, k_sample, dev_index, precond)
    be = prob.backend
    n, nnz = be.n, be.nnz
    for bc in prob.bcs:
        bc.update(0.0)
    heated = [prob.bcs[3]]

    # ---- warm-up steps (untimed), then exactly K timed steps
    if args.warmup > 0:
        prob.run(args.warmup, time_varying=heated, first_step=0)
    barrier_sync()
    t0 = time.perf_counter()
    _, _, iters = prob.run(args.steps, time_varying=heated, first_step=args.warmup)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    gpu_ms = be.last_gpu_ms()
    if dist is not None:
        import torch
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])

    # ---- duration of the dominant kernel, two HIP-event measurements on the solver's own stream:
    #  (a) 100 back-to-back launches of the kernel on the live matrix right after the timed region: this is the
    #      number rocprofv3 reports for the same kernel inside the loop (its begin/end hardware timestamps);
    #  (b) in the loop itself: kernel-attached start/stop events on launches of extra steps - these also
    #      contain the ~4 us dependency gap to the previous kernel, so they read higher than (a) and rocprofv3.
    #  roofline.achieved uses (a); (b) is reported beside it.
    spmv_us_loop = None
    if args.profile_steps > 0:
        be.set_profile(True)
        prob.run(args.profile_steps, time_varying=heated, first_step=args.warmup + args.steps)
        ms_sum, cnt = be.get_profile()
        be.set_profile(False)
        if cnt > 0:
            spmv_us_loop = 1e3 * ms_sum / cnt
    # iteration-head kernel k_spmv<9>: vals 8 + colidx 4 per nnz; per row rowptr 4, z 8 (gathered operand),
    # Ap 8+8 and p 8+8 (read-modify-write by the direction recurrence) = 12*nnz + 44*n
    spmv_bytes = 12 * nnz + 44 * n
    from heatflow_amd import hip_backend as hb
    k_us = {nm: 1e3 * be.time_kernel(k, 100) for nm, k in
            (("spmv", hb.K_PCG_SPMV), ("update", hb.K_PCG_UPDATE), ("plain_spmv", hb.K_SPMV))}
    spmv_us = k_us["spmv"]
    achieved = spmv_bytes / (spmv_us * 1e-6) / 1e9
    # the device's own read ceiling on this operator's arrays ("measured device bandwidth on the box
    # alongside the nominal 8 TB/s"): a plain 16-byte-load streaming read of values + column indices, 12*nnz bytes
    read_us = 1e3 * be.time_kernel(hb.K_STREAM_READ, 100)
    stream_gbs = 12 * nnz / (read_us * 1e-6) / 1e9
    # HBM traffic of that kernel from the PMC passes kept under profiles/ (rocprofv3 cannot wrap itself):
    # only quoted when it was collected on exactly this matrix
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
            pmc = json.load(f)
        if pmc["n"] == n and pmc["nnz"] == nnz:
            traffic = pmc["kernels"]["k_spmv<9>"]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass

    # ---- for the record: the plain Jacobi-PCG loop (north-star solver) on the same steps
    jacobi = None
    if precond == 1 and args.jacobi_steps > 0 and world == 1:
        amg_info = be.amg_info()
        be.set_precond(0, False)
        be.assemble(prob.dt, prob.assembly_mode)
        prob.set_state(float(cfg["heating"]["ic_temp"]))
        if args.warmup > 0:
            prob.run(args.warmup, time_varying=heated, first_step=0)
        tj = time.perf_counter()
        _, _, itj = prob.run(args.jacobi_steps, time_varying=heated, first_step=args.warmup)
        tj = time.perf_counter() - tj
        jacobi = {"steps": args.jacobi_steps, "ms_per_step": 1e3 * tj / args.jacobi_steps,
                  "pcg_iters_per_step_mean": float(np.mean(itj)), "value": n * args.jacobi_steps / tj}
    elif precond == 1:
        amg_info = be.amg_info()

    if rank == 0:
        out = {
            "metric": "DOF-updates/s (timesteps/s x nDOF) on geballe_with_diamond", "value": world * n * args.steps / elapsed,
            "unit": "DOF-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cfgs/geballe_with_diamond.yaml, every mats.*.mesh x {args.scale} (BASELINE C3, ~1M DOF), "
                                   f"steps {args.warmup}..{args.warmup + args.steps - 1} of 100, dt=7.5e-8 s",
                       "n_dof": n, "n_elem": be.n_e, "nnz": nnz, "n_dirichlet": be.n_bc,
                       "solver": ("PCG + smoothed-aggregation multigrid V(1,1), damped-Jacobi smoothing" if precond
                                  else "Jacobi-PCG"),
                       "pcg_rtol": prob.rtol, "pcg_iters_per_step_mean": float(np.mean(iters)),
                       "pcg_iters_per_step_max": int(np.max(iters)),
                       "points": "1 sweep point per GPU (kappa_sample = 3.8 + 0.02*rank)" if world > 1 else "1 run",
                       "gpu_ms_per_step_events": gpu_ms / args.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_spmv<9> (PCG iteration head: CSR SpMV with the direction update p, Ap fused; the column stream is 16-bit compressed, so the measured traffic is below the formula bytes)",
                         "bytes_per_launch": spmv_bytes, "us_per_launch": spmv_us,
                         "us_per_launch_in_loop_events": spmv_us_loop, "us_back_to_back": k_us,
                         "measured_stream_read": {"GB/s": stream_gbs, "us": read_us, "bytes": 12 * nnz,
                                                  "what": "16-byte-load read of the operator's values + column indices on this box",
                                                  "frac_of_it": achieved / stream_gbs,
                                                  "plain_spmv_frac_of_it": (12 * nnz + 20 * n) / (k_us["plain_spmv"] * 1e-6) / 1e9 / stream_gbs}},
        }
        if precond == 1:
            out["config"]["amg"] = amg_info
        if jacobi is not None:
            out["config"]["jacobi_pcg"] = jacobi
        if world == 1 and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, mesh, args.cpu_steps, args.warmup)
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    prob.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
