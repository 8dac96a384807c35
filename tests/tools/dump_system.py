"""Study tool: write the eliminated fine operator A_hat of a scaled geballe_with_diamond case (oracle assembly) as a
binary CSR file for scripts/micro/amg_study.cpp.    python tests/tools/dump_system.py <scale> <out.bin>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_case
from helpers import reference_bcs, material_tables
from oracle import heat_oracle as ho

scale, out = float(sys.argv[1]), sys.argv[2]
cfg, stack, mesh = build_case("geballe_with_diamond", scale)
bcs, ic, _ = reference_bcs(cfg, stack, mesh)
tag_to_k, tag_to_rc = material_tables(stack, mesh)
dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
obcs = [{"dofs": np.asarray(b.row_dofs), "value": 300.0} for b in bcs]
st = ho.OracleSolver(mesh.coords, mesh.tris, mesh.tags, tag_to_k, tag_to_rc, dt, obcs, np.full(len(mesh.coords), 300.0))
A = st.Ahat.tocsr(); A.sort_indices()
with open(out, "wb") as f:
    np.array([A.shape[0]], dtype=np.int32).tofile(f)
    np.array([A.nnz], dtype=np.int64).tofile(f)
    A.indptr.astype(np.int32).tofile(f); A.indices.astype(np.int32).tofile(f); A.data.astype(np.float64).tofile(f)
print("n", A.shape[0], "nnz", A.nnz)
