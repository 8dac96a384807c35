"""hf_set_mesh (tables built on the host) against hf_set_mesh_prebuilt (tables installed from a blob) at C3 size:
python scripts/pattern_share_probe.py [mesh scale = 0.43]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case
from heatflow_amd import hip_backend as hb

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.43
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
with hb.HeatflowHIP(0) as a:
    a.set_mesh(mesh.coords, mesh.tris, mesh.tags)          # first call on the device: includes one-off runtime set-up
    t0 = time.perf_counter(); a.set_mesh(mesh.coords, mesh.tris, mesh.tags); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); blob = a.export_pattern(); t_exp = time.perf_counter() - t0
with hb.HeatflowHIP(0) as b:
    b.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=blob)
    t0 = time.perf_counter(); b.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=blob); t_inst = time.perf_counter() - t0
print(f"n = {len(mesh.coords)}: hf_set_mesh (build tables) {t_build * 1e3:.1f} ms | hf_pattern_export {t_exp * 1e3:.1f} ms, blob {blob.nbytes / 1e6:.1f} MB | "
      f"hf_set_mesh_prebuilt (install) {t_inst * 1e3:.1f} ms")
