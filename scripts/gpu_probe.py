"""Quick GPU probe: kernel timings and a short time loop on a scaled geballe_with_diamond mesh."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case
from helpers import make_problem
from heatflow_amd import hip_backend as hb

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.43
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
precond = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t0 = time.time()
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
print("mesh", mesh.stats, "%.1fs" % (time.time() - t0), flush=True)
t0 = time.time()
prob = make_problem(cfg, stack, mesh, assembly_mode=mode, precond=precond)
be = prob.backend
print("setup %.2fs  n=%d ne=%d nnz=%d nbc=%d  assemble(+bc) gpu ms=%.3f" % (time.time() - t0, be.n, be.n_e, be.nnz, be.n_bc, be.last_gpu_ms()), flush=True)
n, nnz, ne = be.n, be.nnz, be.n_e
if precond: print("amg", be.amg_info(), flush=True)
prob.run(5, watcher_nodes=None, time_varying=[prob.bcs[3]])          # steps 0..4: boundary values still at the initial temperature
times, samples, iters = prob.run(nsteps, watcher_nodes=None, time_varying=[prob.bcs[3]], first_step=5)
ms = be.last_gpu_ms()
print("run steps 5..%d: %.2f ms total, %.3f ms/step, iters %s" % (4 + nsteps, ms, ms / nsteps, [int(i) for i in iters]), flush=True)
tot_it = int(np.sum(iters))
print("  per PCG iteration (incl. rhs etc): %.2f us" % (1e3 * ms / max(tot_it, 1)))
names = {hb.K_SPMV: ("spmv", 12 * nnz + 20 * n), hb.K_PCG_SPMV: ("pcg_spmv", 12 * nnz + 44 * n), hb.K_PCG_UPDATE: ("pcg_update", 64 * n),
         hb.K_RHS: ("rhs", 12 * nnz + 20 * n)}
for k, (nm, byt) in names.items():
    t = be.time_kernel(k, 200)
    print("  %-11s %8.2f us  %7.1f GB/s (algorithmic %d B)" % (nm, t * 1e3, byt / t / 1e6, byt), flush=True)
for m in (0, 1, 2, 3):
    be.assemble(prob.dt, m)
    t = be.time_kernel(hb.K_ASSEMBLE, 20)
    byt = 16 * ne + 16 * n + 16 * nnz
    print("  assemble mode %d %8.2f us  %7.1f GB/s (algorithmic %d B)" % (m, t * 1e3, byt / t / 1e6, byt), flush=True)
