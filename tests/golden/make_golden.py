"""Generate tests/golden/with_diamond_tiny.npz from the CPU oracle.

The reference cannot be run here (dolfinx/PETSc/gmsh absent) and ships no golden vectors, so
this fixture is produced by oracle/heat_oracle.py itself on a tiny nine-material mesh
(geballe_with_diamond with every mats.*.mesh x 16 -> ~2k nodes) made by heatflow_amd.mesh.
It freezes: the mesh arrays, the Dirichlet DOF list, and the full temperature field of the
first 12 time steps (dt = 7.5e-8 s; heating starts at step 5).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import HEATING_CSV, build_case  # noqa: E402
from oracle import heat_oracle as ho  # noqa: E402

SCALE, STEPS = 16.0, 12

cfg, stack, mesh = build_case("geballe_with_diamond", SCALE)
res = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV,
                                 num_steps=STEPS, keep_fields=True)
names = sorted(mesh.material_tags)
np.savez_compressed(
    os.path.join(HERE, "with_diamond_tiny.npz"),
    mesh_scale=SCALE, coords=mesh.coords, tris=mesh.tris, tags=mesh.tags,
    material_names=np.array(names), material_tag_values=np.array([mesh.material_tags[n] for n in names]),
    bc_dofs=res["solver"].bc_dofs, times=res["times"], fields=res["fields"],
)
print("wrote with_diamond_tiny.npz:", mesh.stats, "max T", res["fields"][-1].max())
