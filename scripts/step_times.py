"""Wall time of every time step of the C3 run, one hf_run call per step (diagnosis of cold-start effects).
    python scripts/step_times.py [scale] [steps]"""
import os, sys, time
T0 = time.perf_counter()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case
from helpers import make_problem

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.43
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 65
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
prob = make_problem(cfg, stack, mesh, assembly_mode=3, precond=1)
heated = [prob.bcs[3]]
warm_s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if warm_s > 0:
    from heatflow_amd import hip_backend as hb
    tw = time.perf_counter()
    while time.perf_counter() - tw < warm_s:
        prob.backend.time_kernel(hb.K_SPMV, 500)
print("process age at loop start %.1f s" % (time.perf_counter() - T0))
rows = []
for s in range(nsteps):
    t0 = time.perf_counter()
    _, _, it = prob.run(1, time_varying=heated, first_step=s)
    rows.append((1e3 * (time.perf_counter() - t0), prob.backend.last_gpu_ms(), int(it[0])))
for s, (w, g, it) in enumerate(rows):
    if s < 5 or w > 2.3:
        print(f"step {s:3d} wall {w:7.3f} ms gpu {g:7.3f} ms iters {it}")
w = np.array([r[0] for r in rows[5:]]); g = np.array([r[1] for r in rows[5:]])
print(f"steps 5..: wall mean {w.mean():.3f} median {np.median(w):.3f}  gpu mean {g.mean():.3f} median {np.median(g):.3f}")
