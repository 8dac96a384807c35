"""A/B kernel timing: one process per library build (HEATFLOW_HIP_LIB), same mesh from an npz cache."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from conftest import build_case
    from helpers import make_problem
    from heatflow_amd import hip_backend as hb
    cfg, stack, mesh = build_case("geballe_with_diamond", float(sys.argv[2]))
    prob = make_problem(cfg, stack, mesh, precond=int(sys.argv[3]))
    be = prob.backend
    out = {}
    for nm, k in (("spmv", hb.K_SPMV), ("pcg_spmv", hb.K_PCG_SPMV), ("update", hb.K_PCG_UPDATE)):
        out[nm] = round(min(be.time_kernel(k, 200) for _ in range(3)) * 1e3, 2)
    t, s, it = prob.run(16, time_varying=[prob.bcs[3]])
    out["ms_per_step_16"] = round(be.last_gpu_ms() / 16, 3); out["iters"] = int(np.sum(it))
    for m in (0, 1):
        ts = []
        for _ in range(3):
            be.assemble(prob.dt, m)
            ts.append(be.time_kernel(hb.K_ASSEMBLE, 20))
        out["asm%d" % m] = round(min(ts) * 1e3, 1)
        be.assemble(prob.dt, m)
    print(json.dumps(out))
else:
    scale, precond = sys.argv[1], sys.argv[2]
    for lib in sys.argv[3:]:
        env = dict(os.environ)
        if lib != "default":
            env["HEATFLOW_HIP_LIB"] = os.path.join(ROOT, "heatflow_amd", "csrc", lib)
        p = subprocess.run([sys.executable, __file__, "--child", scale, precond], env=env, capture_output=True, text=True)
        print(lib.ljust(16), p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-500:], flush=True)
