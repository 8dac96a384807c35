"""Wall-clock of the read-flux config (BASELINE C4) through run_no_diamond.run_simulation."""
import sys, os, time, tempfile, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml
import run_no_diamond as run
from heatflow_amd.geometry import watcher_points
cfg = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "geballe_no_diamond_read_flux.yaml")))
tmp = tempfile.mkdtemp()
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    res = run.run_simulation(cfg, os.path.join(tmp, "mesh"), rebuild_mesh=True, output_folder=os.path.join(tmp, "out"),
                             watcher_points=watcher_points(cfg), write_xdmf=False, suppress_print=True)
    pr.disable()
    print("run %d wall %.3f s" % (rep, time.perf_counter() - t0), {k: v for k, v in res.items() if k in ("loop_s", "iters_mean")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
