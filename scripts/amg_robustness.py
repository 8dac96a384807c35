"""Multigrid-PCG vs Jacobi-PCG over a spread of operators (mesh scale, kappa_sample, dt): iterations,
agreement of the fields after 8 steps from the onset of the heating, fallback count."""
import os, sys, copy, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import build_case
from helpers import make_problem

for name, scale in (("geballe_with_diamond", 1.0), ("geballe_with_diamond", 3.0), ("geballe_no_diamond", 1.0)):
    cfg0, stack0, mesh = build_case(name, scale)
    for ks, dtf in itertools.product((0.3, 3.8, 60.0), (0.05, 1.0, 20.0)):
        cfg = copy.deepcopy(cfg0)
        cfg["mats"]["p_sample"]["k"] = ks
        cfg["timing"]["t_final"] = float(cfg["timing"]["t_final"]) * dtf
        from heatflow_amd.geometry import build_stack
        stack = build_stack(cfg)
        out = {}
        for pc in (1, 0):
            prob = make_problem(cfg, stack, mesh, precond=pc, max_it=50000)
            for bc in prob.bcs: bc.update(0.0)
            try:
                k0 = int(3.6e-7 / prob.dt)           # start where the heating curve starts (short steps would see a constant field)
                for k in range(k0, k0 + 8): prob.step((k + 1) * prob.dt)
                out[pc] = (prob.state(), max(prob.iters), prob.backend.amg_info()["jacobi_fallbacks"] if pc else 0)
            except Exception as e:
                out[pc] = (None, str(e)[:80], -1)
            prob.close()
        if out[1][0] is not None and out[0][0] is not None:
            d = np.abs(out[1][0] - out[0][0]).max()
        else:
            d = float("nan")
        print(f"{name} scale {scale} k_sample {ks:5.1f} dt x{dtf:5.2f}: amg its {out[1][1]} jacobi its {out[0][1]} fallbacks {out[1][2]} |dT| {d:.2e}", flush=True)
