import os, sys, copy
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import build_case
from helpers import make_problem, material_tables
cfg, stack, mesh = build_case("geballe_with_diamond", 1.0)
prob = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)
tk, trc = material_tables(stack, mesh)
for ks in (3.8, 0.3, 60.0, 3.8):
    t=dict(tk); t[mesh.material_tags["p_sample"]]=ks
    prob.set_materials(t, trc); prob.set_state(300.0); prob.iters=[]
    for bc in prob.bcs: bc.update(0.0)
    for k in range(12): prob.step((k+1)*prob.dt)
    print("k_sample", ks, "iters", prob.iters[4:], prob.backend.amg_info()["setup_s"])
prob.close()
