"""Shared set-up for the parity tests: the same problem on the oracle and on the HIP path."""
import numpy as np

from conftest import HEATING_CSV


def material_tables(stack, mesh):
    tag_to_k = {mesh.material_tags[m.name]: m.properties["k"] for m in stack.materials}
    tag_to_rc = {mesh.material_tags[m.name]: m.properties["rho_cv"] for m in stack.materials}
    return tag_to_k, tag_to_rc


def reference_bcs(cfg, stack, mesh):
    """[left, right, top, inner] exactly as run_with_diamond.py:362-373."""
    from heatflow_amd.bc import P1Space, RowDirichletBC
    from heatflow_amd.heating import HeatingCurve

    ic = float(cfg["heating"]["ic_temp"])
    heat = HeatingCurve(HEATING_CSV, ic, float(cfg["heating"]["fwhm"]))
    V = P1Space(mesh.coords)
    bcs = [
        RowDirichletBC(V, "left", value=ic),
        RowDirichletBC(V, "right", value=ic),
        RowDirichletBC(V, "top", value=ic),
        RowDirichletBC(V, "x", coord=stack.heated_z, length=abs(stack.r_sample) * 2, center=0.0, value=heat.gaussian),
    ]
    return bcs, ic, heat


def make_problem(cfg, stack, mesh, **kw):
    from heatflow_amd.solver import HeatProblem

    bcs, ic, _ = reference_bcs(cfg, stack, mesh)
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    return HeatProblem(mesh.coords, mesh.tris, mesh.tags, tag_to_k, tag_to_rc, dt, bcs, ic, **kw)


def oracle_run(cfg, mesh, num_steps, keep_fields=True, watcher_nodes=None):
    from oracle import heat_oracle as ho

    return ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV,
                                      num_steps=num_steps, keep_fields=keep_fields, watcher_nodes=watcher_nodes)


def csr_values_on_pattern(S, rowptr, colidx):
    """Values of scipy CSR matrix S laid out on the (rowptr, colidx) pattern (must be a superset)."""
    import scipy.sparse as sp

    n = len(rowptr) - 1
    P = sp.csr_matrix((np.arange(1, len(colidx) + 1, dtype=np.float64), colidx, rowptr), shape=(n, n))
    S = S.tocsr()
    S.sort_indices()
    out = np.zeros(len(colidx))
    # every stored entry of S must exist in the pattern
    Sc = S.tocoo()
    slot = np.asarray(P[Sc.row, Sc.col]).ravel().astype(np.int64)
    if (slot[Sc.data != 0] == 0).any():
        raise AssertionError("oracle matrix has an entry outside the device pattern")
    ok = slot > 0
    out[slot[ok] - 1] = Sc.data[ok]
    return out
