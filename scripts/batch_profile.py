"""Host-side profile (cProfile) of one batched run of 8 kappa points on the stock mesh: where a rank's wall time goes
besides the time loop itself.    python scripts/batch_profile.py"""
import copy, cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case, HEATING_CSV
from heatflow_amd.driver import SimulationSession
from heatflow_amd.geometry import build_stack, watcher_points

cfg, stack, mesh = build_case("geballe_with_diamond", 1.0)
cfg["heating"]["file"] = HEATING_CSV
sess = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags)
cfgs = []
for j in range(8):
    c = copy.deepcopy(cfg)
    c["mats"]["p_sample"]["k"] = 3.3 + j / 7
    cfgs.append(c)
stacks = [build_stack(c) for c in cfgs]
sess.run_batch(cfgs, stacks, watcher_points(cfgs[0]))          # first batch: set-up included
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = sess.run_batch(cfgs, stacks, watcher_points(cfgs[0]))
pr.disable()
print(f"second batch: {time.perf_counter() - t0:.3f} s wall, loop {res[0]['loop_time'] * 8:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
sess.close()
