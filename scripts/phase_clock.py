"""usage (GPU box): HEATFLOW_HIP_LIB=<measurement build> python scripts/phase_clock.py <label> [scale]
Phase stamps of one k_spmv launch inside the C3 time loop (library built with -DHF_PHASE_CLOCK=<mode> [-DHF_PHASE_CONV=1],
see hf_kernels.hpp): lane 0 of every workgroup stamps the 100 MHz clock at entry and, per chunk, after the operand slice is
staged, after the products are parked in LDS and after the rows are summed.  Prints where the launch's time goes."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    label = sys.argv[1]
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.43
    from conftest import build_case
    from heatflow_amd import hip_backend as hb
    from helpers import make_problem

    batch_nv = int(os.environ.get("PHASE_BATCH_NV", "0"))       # > 0: one batch of that many kappa points (kb_spmv_lds stamps)
    cfg, stack, mesh = build_case("geballe_with_diamond", scale)
    lib = hb.load_library()
    buf = np.zeros(1024 * 16, np.uint64)
    if batch_nv > 0:
        import copy
        from conftest import HEATING_CSV
        from heatflow_amd.driver import SimulationSession
        from heatflow_amd.geometry import build_stack, watcher_points
        cfg["heating"]["file"] = HEATING_CSV
        dt0 = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
        cfg["timing"]["num_steps"], cfg["timing"]["t_final"] = 12, dt0 * 12
        cfgs = []
        for j in range(batch_nv):
            c = copy.deepcopy(cfg)
            c["mats"]["p_sample"]["k"] = 3.3 + j / max(batch_nv - 1, 1)
            cfgs.append(c)
        sess = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags)
        sess.run_batch(cfgs, [build_stack(c) for c in cfgs], watcher_points(cfgs[0]))
        assert lib.hf_debug_phases(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        sess.close()
    else:
        prob = make_problem(cfg, stack, mesh, assembly_mode=0, precond=1)
        heated = [prob.bcs[3]]
        prob.run(12, watcher_nodes=None, time_varying=heated)
        assert lib.hf_debug_phases(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        prob.close()
    t = buf.reshape(1024, 16).astype(np.float64) * 0.01      # us
    used = t[:, 0] > 0
    t = t[used]
    t0 = np.median(t[:, 0])
    t = t[np.abs(t[:, 0] - t0) < 500.0]             # workgroups stamped by the last launch
    t0 = t[:, 0].min()
    t[(t < t0) | (t > t0 + 1000.0)] = 0.0           # slots this launch did not write (stamps of earlier launches)
    nst = (t > 0).sum(axis=1)
    end = t.max(axis=1)
    print(f"== {label}: {len(t)} workgroups stamped; first entry -> last exit {end.max() - t0:.2f} us; entries spread over "
          f"{t[:, 0].max() - t0:.2f} us (median entry +{np.median(t[:, 0]) - t0:.2f})")
    print(f"   stamps per workgroup: {dict(zip(*np.unique(nst, return_counts=True)))}  (1 + 3 per chunk)")
    names = ["stream parked", "slice staged", "rows"] if batch_nv > 0 else ["staged", "products", "rows"]
    for c in range(4):
        rows = t[nst >= 1 + 3 * (c + 1)]
        if len(rows) == 0:
            break
        prev = rows[:, 3 * c]
        msg = [f"chunk {c} ({len(rows)} wg): starts +{np.median(prev) - t0:.2f}"]
        for k in range(3):
            d = rows[:, 3 * c + 1 + k] - rows[:, 3 * c + k]
            msg.append(f"{names[k]} {np.median(d):.2f} (p90 {np.percentile(d, 90):.2f})")
        print("   " + "; ".join(msg))
    print(f"   workgroup exit: median +{np.median(end) - t0:.2f}, p90 +{np.percentile(end, 90):.2f}, max +{end.max() - t0:.2f}")


if __name__ == "__main__":
    main()
