"""C5 rehearsal on one GPU: kappa_sample sweep on the stock geballe_with_diamond mesh.
    python scripts/sweep_bench.py [n_points] [scale]
Under torch.distributed.run each rank takes points i mod world (heatflow_amd.parameter_sweep)."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, yaml
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    import torch, torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
from heatflow_amd import parameter_sweep as ps
from heatflow_amd.geometry import scale_mesh_sizes

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
conc = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = scale_mesh_sizes(yaml.safe_load(open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml"))), scale)
cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
tmp = tempfile.mkdtemp()
t0 = time.time()
rows = ps.run_kappa_sweep(cfg, os.path.join(tmp, "mesh"), ps.get_k_values(count=npts), os.path.join(tmp, "out"),
                          rebuild_mesh=True, exp_csv=cfg["heating"]["file"], concurrent=conc)
t1 = time.time()
rank, world = ps.world_info()
if rank == 0:
    ok = [r for r in rows if r["status"] == "success"]
    print(json.dumps({"points": npts, "world": world, "concurrent": conc, "ok": len(ok), "wall_s": round(t1 - t0, 2),
                      "sum_runtime_s": round(sum(r["runtime"] for r in ok), 2),
                      "per_point_s": round(np.mean([r["runtime"] for r in ok]), 3),
                      "iters_mean": round(np.mean([r["pcg_iters_mean"] for r in ok]), 1),
                      "best_k": min(ok, key=lambda r: r["rmse"])["k"], "errors": [r["error"] for r in rows if r["error"]][:2]}))
