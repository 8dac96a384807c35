"""A long time loop (default 1500 steps, dt as configured, the heating curve clamps after its last sample) on a coarse
with-diamond mesh: iteration counts must stay bounded and the final field must match the oracle's - the projection
ring's incrementally kept Gram matrix and the boundary responses are re-used for the whole run.
    python scripts/long_run_check.py [steps] [scale]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case
from helpers import make_problem, oracle_run

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
for precond in (1, 0):
    prob = make_problem(cfg, stack, mesh, precond=precond)
    t0 = time.time()
    _, _, iters = prob.run(steps, time_varying=[prob.bcs[3]])
    t1 = time.time() - t0
    u = prob.state()
    prob.close()
    it = np.asarray(iters)
    print(f"precond {precond}: n = {len(mesh.coords)}, {steps} steps in {t1:.2f} s, iterations/step first 100: {it[5:100].mean():.1f}, "
          f"last 100: {it[-100:].mean():.1f}, max {it.max()}, T range [{u.min():.3f}, {u.max():.3f}]")
    if precond == 1:
        u_amg = u
print("multigrid vs Jacobi final field: max |dT| = %.3e K" % np.abs(u_amg - u).max())
cfg2 = dict(cfg); cfg2["timing"] = dict(cfg["timing"])
dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
cfg2["timing"]["num_steps"] = steps; cfg2["timing"]["t_final"] = dt * steps
nodes = np.array([0, len(mesh.coords) // 3, len(mesh.coords) // 2, len(mesh.coords) - 1])
ref = oracle_run(cfg2, mesh, steps, keep_fields=False, watcher_nodes=nodes)
print("vs the oracle (direct solves) at 4 nodes after %d steps: max |dT| = %.3e K" % (steps, np.abs(ref["watchers"][-1] - u_amg[nodes]).max()))
