"""How much do K solver contexts on one GPU overlap?  K threads, each stepping its own stock-size problem."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import build_case
from helpers import make_problem
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
probs = [make_problem(cfg, stack, mesh, precond=1) for _ in range(8)]
for p in probs:
    for bc in p.bcs: bc.update(0.0)
gs = []
p0 = probs[0]
g_all = np.empty((100, len(p0.bc_dofs)))
for k in range(100):
    g_all[k] = p0.bc_values((k + 1) * p0.dt, [p0.bcs[3]])
def work(p):
    p.set_state(300.0)
    p.backend.run(g_all, 1e-10, 0.0, 20000, None)
for K in (1, 2, 4, 8):
    ts = [threading.Thread(target=work, args=(probs[i],)) for i in range(K)]
    t0 = time.time()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.time() - t0
    print(f"K={K}: {dt*1e3:.0f} ms for {K} x 100 steps -> {dt*1e3/K:.1f} ms per run", flush=True)
for p in probs: p.close()
