import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from conftest import build_case
from helpers import make_problem
from heatflow_amd import hip_backend as hb
cfg, stack, mesh = build_case("geballe_with_diamond", 2.0)
lib = hb.load_library()
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
vals = []
for rep in range(12):
    prob = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)
    prob.run(12, time_varying=[prob.bcs[3]])
    be = prob.backend
    be.batch_begin(8, hb.BATCH_SHARED)
    for j in range(8): be.batch_set_state(j, np.full(prob.n, 300.0))
    g = np.stack([np.repeat(prob.bc_values((s + 1) * prob.dt, [prob.bcs[3]])[:, None], 8, axis=1) for s in range(6)])
    be.batch_run(g, prob.rtol, 0.0, prob.max_it, None)
    be.batch_end()
    be.flux_setup(); be.flux_solve(prob.rtol, 5000)
    prob.close()
    vals.append(free_mb())
print("free MB after each create/run/destroy cycle:", [round(v) for v in vals])
print("drift over the last 10 cycles: %.1f MB" % (vals[1] - vals[-1]))
