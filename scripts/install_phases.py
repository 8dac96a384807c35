"""Wall time of every call a sweep session makes to become ready when the connectivity tables and the multigrid hierarchy
are installed from blobs (what a non-zero rank / a second session does):   python scripts/install_phases.py [mesh scale]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case
from helpers import make_problem, material_tables, reference_bcs
from heatflow_amd import hip_backend as hb
from heatflow_amd.bc import merge_bcs

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
t0 = time.perf_counter()
a = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)
t_build = time.perf_counter() - t0
t0 = time.perf_counter(); pat = a.backend.export_pattern(); t_pe = time.perf_counter() - t0
t0 = time.perf_counter(); amg = a.backend.amg_export(); t_ae = time.perf_counter() - t0
print(f"n = {a.n}: building session {t_build:.3f} s (set_mesh {a.mesh_seconds:.3f}, hierarchy {a.backend.amg_info()['setup_s']:.3f}); "
      f"export pattern {1e3 * t_pe:.1f} ms ({pat.nbytes / 1e6:.1f} MB), hierarchy {1e3 * t_ae:.1f} ms ({amg.nbytes / 1e6:.1f} MB)")
bcs, ic, _ = reference_bcs(cfg, stack, mesh)
tk, trc = material_tables(stack, mesh)
tags = sorted(tk)
dofs = merge_bcs(bcs)[0]
dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
for rep in range(3):
    lap = []
    def tick(name, t=[time.perf_counter()]):
        now = time.perf_counter(); lap.append((name, now - t[0])); t[0] = now
    tick("-")
    be = hb.HeatflowHIP(0); tick("hf_create")
    be.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=pat); tick("set_mesh_prebuilt")
    be.set_materials(np.array(tags, dtype=np.int32), np.array([tk[t] for t in tags]), np.array([trc[t] for t in tags])); tick("set_materials")
    be.set_dirichlet(dofs); tick("set_dirichlet")
    be.set_precond(1, True); be.amg_install(amg); tick("amg_install")
    be.assemble(dt, 3); tick("assemble")
    be.set_state(np.full(be.n, 300.0)); tick("set_state")
    g = np.full((5, be.n_bc), 300.0); be.run(g, 1e-10, 0.0, 100, None); tick("5 steps")
    be.close(); tick("close")
    print("  " + ", ".join(f"{nm} {1e3 * s:.1f}" for nm, s in lap[1:]) + f"  | total {1e3 * sum(s for _, s in lap[1:]):.1f} ms")
a.close()
