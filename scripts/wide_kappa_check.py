"""kappa_sample sweep over two decades (0.5 .. 60 W/m/K) on the stock with-diamond mesh, one run per point and as one batch
of 8: exercises the hierarchy rebuild of a session when a conductivity drifts past 2x and a batch whose columns share one
frozen hierarchy far from some of them.    python scripts/wide_kappa_check.py"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml
from heatflow_amd import parameter_sweep as ps

cfg = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")))
cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
for batch in (1, 8):
    tmp = tempfile.mkdtemp()
    rows = ps.run_kappa_sweep(cfg, os.path.join(tmp, "mesh"), [1.0, 2.0, 3.8, 8.0, 20.0, 60.0, 0.5, 3.9], os.path.join(tmp, "out"),
                              rebuild_mesh=True, exp_csv=cfg["heating"]["file"], batch=batch)
    print("batch", batch, [(r["k"], r["status"], round(r.get("pcg_iters_mean", -1), 1)) for r in rows], [r["error"] for r in rows if r["error"]][:2])
