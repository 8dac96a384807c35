"""What one rank of an 8-GPU C5 sweep runs: 8 kappa points on the stock mesh.  One batch of 8 (one loop), two batches of 4 in
flight, four of 2: which keeps a GPU busiest when there is only one batch worth of points?
    python scripts/rank_share_probe.py [points]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml
from heatflow_amd import parameter_sweep as ps

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")))
cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
ks = ps.get_k_values(count=64)[:npts]
for batch, conc in ((npts, 1), (npts // 2, 2), (npts // 4, 4), (npts // 2, 1), (1, 4)):
    if batch < 1:
        continue
    tmp = tempfile.mkdtemp()
    mark, timing = {}, {}
    rows = ps.run_kappa_sweep(cfg, os.path.join(tmp, "mesh"), ks, os.path.join(tmp, "out"), rebuild_mesh=True, concurrent=conc, batch=batch,
                              warmup_steps=5, on_ready=lambda: mark.setdefault("t0", time.perf_counter()),
                              on_done=lambda: mark.setdefault("t1", time.perf_counter()), timing=timing)
    assert all(r["status"] == "success" for r in rows), rows
    wall = mark["t1"] - mark["t0"]
    print(f"{npts} points, batches of {batch}, {conc} in flight: point loop {wall:.3f} s = {npts * timing['n_dof'] * 100 / wall:.3e} DOF-updates/s "
          f"(batches {timing['batches']}, sessions {timing['sessions']})", flush=True)
