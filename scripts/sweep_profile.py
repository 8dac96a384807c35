"""Host-side profile (cProfile) of an 8-point kappa sweep = one batch on the stock mesh, second call on a warm session
cache is not available here, so the first call pays mesh + set-up; the profile lists what surrounds the time loop.
    python scripts/sweep_profile.py"""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml
from heatflow_amd import parameter_sweep as ps

cfg = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "geballe_with_diamond.yaml")))
cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
tmp = tempfile.mkdtemp()
timing = {}
pr = cProfile.Profile()
marks = {}
pr.enable()
rows = ps.run_kappa_sweep(cfg, os.path.join(tmp, "mesh"), ps.get_k_values(count=8), os.path.join(tmp, "out"), rebuild_mesh=True,
                          exp_csv=cfg["heating"]["file"], concurrent=1, batch=8, warmup_steps=5,
                          on_ready=lambda: marks.setdefault("t0", time.perf_counter()), on_done=lambda: marks.setdefault("t1", time.perf_counter()),
                          timing=timing)
pr.disable()
print("points loop %.3f s" % (marks["t1"] - marks["t0"]), timing)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
