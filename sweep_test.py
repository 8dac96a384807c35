"""kappa_sample sweep on the shared geballe_with_diamond mesh (role of the reference's sweep_test.py:
k0 = 3.8 +- 0.5 in steps of 0.02, normalised o-side RMSE per point, rmse_summary.csv).

    python sweep_test.py                      # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 sweep_test.py   # 8 GPUs
Points go to ranks i mod world; the mesh is built by rank 0 and broadcast over RCCL."""
import os
import time

import yaml

from heatflow_amd import parameter_sweep as ps


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    with open(os.path.join(here, "cfgs", "geballe_with_diamond.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["heating"]["file"] = os.path.join(here, cfg["heating"]["file"])
    t0 = time.time()
    rows = ps.run_kappa_sweep(cfg, "meshes/test1", ps.get_k_values(), "outputs/sweep_test", rebuild_mesh=True,
                              exp_csv=cfg["heating"]["file"], batch=8, concurrent=2)   # batches of 8 points, 2 time loops in flight per GPU
    if ps.world_info()[0] == 0:
        ok = [r for r in rows if r["status"] == "success"]
        best = min(ok, key=lambda r: r["rmse"])
        print(f"Lowest RMSE: {best['rmse']:.6f} at k = {best['k']:.2f}")
        print(f"Total sweep time: {time.time() - t0:.2f}s for {len(rows)} points ({len(rows) - len(ok)} failed)")


if __name__ == "__main__":
    main()
