"""Batched time loop probe: python scripts/batch_probe.py <mesh scale> <nv> <steps> [percol=1] -> phase timings of one batch."""
import copy, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case, HEATING_CSV
from heatflow_amd.driver import SimulationSession
from heatflow_amd.geometry import build_stack, watcher_points

scale, nv, steps = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
percol = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
cfg["heating"]["file"] = HEATING_CSV
dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
cfg["timing"]["num_steps"] = steps
cfg["timing"]["t_final"] = dt0 * steps
sess = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags)
cfgs = []
for j in range(nv):
    c = copy.deepcopy(cfg)
    if percol:
        c["mats"]["p_sample"]["k"] = 3.3 + j / max(nv - 1, 1)
    else:
        c["heating"]["fwhm"] = 1.0e-5 * (1 + 0.2 * j)
    cfgs.append(c)
stacks = [build_stack(c) for c in cfgs]
t0 = time.time(); r1 = sess.run(cfgs[0], stacks[0], watcher_points(cfgs[0])); t_single = time.time() - t0
t0 = time.time(); r1 = sess.run(cfgs[-1], stacks[-1], watcher_points(cfgs[-1])); t_single2 = time.time() - t0
print(f"n = {len(mesh.coords)}  nnz = {sess.problem.backend.nnz}  single run: first {t_single:.3f} s (set-up included), second {t_single2:.3f} s, loop {r1['loop_time']:.3f} s, "
      f"gpu {sess.problem.backend.last_gpu_ms():.1f} ms, iters/step {np.mean(r1['iters']):.1f}")
for rep in range(2):
    t0 = time.time(); res = sess.run_batch(cfgs, stacks, watcher_points(cfgs[0])); t_b = time.time() - t0
    print(f"batch of {nv} ({'per-column' if percol else 'shared'} operator): {t_b:.3f} s wall, loop {res[0]['loop_time'] * nv:.3f} s, "
          f"gpu {sess.problem.backend.last_gpu_ms():.1f} ms = {sess.problem.backend.last_gpu_ms() / steps:.3f} ms/step, "
          f"iters/step {np.mean([r['iters'] for r in res]):.1f}")
d = max(np.abs(res[-1]["watchers"]["oside"] - r1["watchers"]["oside"]).max(), np.abs(res[-1]["watchers"]["pside"] - r1["watchers"]["pside"]).max())
print(f"last column vs its single run: max |dT| at the watchers = {d:.2e} K")
sess.close()
