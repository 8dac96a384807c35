"""The reference's production sweep (parameter_sweep.py: fwhm x k grid over run_no_diamond with its read-flux outputs) on the stock
no-diamond mesh, 16 points on one GPU: through the batched loop (one batch of 16) and point by point.
    python scripts/no_diamond_sweep_probe.py"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, yaml
from heatflow_amd import parameter_sweep as ps

cfg = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "geballe_no_diamond.yaml")))
cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
tmp = tempfile.mkdtemp()
cfg_path = os.path.join(tmp, "cfg.yaml")
yaml.safe_dump(cfg, open(cfg_path, "w"))
width = float(cfg["mats"]["p_sample"]["z"])
grid = ((1.0e-5, 1.6e-5), (3.2, 4.4), (width, width), (4, 4, 1))
for batch in (16, 1, 16):
    out = os.path.join(tmp, f"out_b{batch}_{time.time_ns()}")
    t0 = time.perf_counter()
    ok, failed = ps.run_parameter_sweep(cfg_path, out, *grid, base_mesh_folder=os.path.join(tmp, "meshes"), batch=batch)
    wall = time.perf_counter() - t0
    assert len(ok) == 16 and not failed, failed[:1]
    n = len(np.load(os.path.join(ps.get_mesh_folder_for_width(os.path.join(tmp, "meshes"), width), "mesh.npz"))["coords"])
    steps = int(cfg["timing"]["num_steps"])
    print(f"16 points x {n} DOF x {steps} steps, batch {batch}: {wall:.2f} s whole call = {16 * n * steps / wall:.3e} DOF-updates/s, "
          f"PCG iterations/step {np.mean([r['pcg_iters_mean'] for r in ok]):.1f}", flush=True)
a = np.genfromtxt(os.path.join(tmp, [d for d in os.listdir(tmp) if d.startswith("out_b16")][0], ok[5]["run_name"], "radial_gradient_raw.csv"), delimiter=",", skip_header=1)
b = np.genfromtxt(os.path.join(tmp, [d for d in os.listdir(tmp) if d.startswith("out_b1_")][0], ok[5]["run_name"], "radial_gradient_raw.csv"), delimiter=",", skip_header=1)
print("batched vs point-by-point radial_gradient_raw.csv of one point: max |diff| / max |value| =", np.abs(a - b).max() / np.abs(b).max())
shutil.rmtree(tmp, ignore_errors=True)
