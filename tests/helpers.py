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


# --------------------------------------------------------------------------------------
# The one output the reference holds: clean_with_ir.ipynb cell 22 prints
#   RMSE = 0.015309 / Max error = 0.030814
# for the normalised o-side curve of the nine-box stack of its cells 5-6 (200 steps to the last time of
# the heating CSV, fwhm 13.2 um, heated line on the p-side insulator/coupler interface, watchers at
# (heated line, r=0) and (outer face of the o-side coupler, r=0), nearest-vertex sampling including the
# t=0 state: cells 11, 15-21).  The notebook reads `experimental_data/raw_temp_time_curve.csv` (time in us),
# which the reference does not ship; `geballe_heat_data.csv` (time in s, same three columns, 51 rows)
# stands in for it.  The notebook's y runs over [-100 um, 0] with the axis at y = 0 (r = |y - 0|, cell 18):
# here r = -y.
# --------------------------------------------------------------------------------------
NOTEBOOK_RMSE, NOTEBOOK_MAX_ERR = 0.015309, 0.030814


def notebook_clean_with_ir_case(scale=1.0):
    """(materials, mesh bounds, heated_z, oside_z, r_sample) of clean_with_ir.ipynb cells 5-6;
    every mesh_size is multiplied by ``scale``."""
    from heatflow_amd.materials import Material

    d_ins_o, d_ins_p, d_s, d_ir, d_diam = 6.3e-6, 3.2e-6, 1.84e-6, 0.062e-6, 40e-6
    r_s, r_gask, r_insg = 20e-6, 75e-6, 5e-6
    z0 = -((d_ins_o + d_ins_p + d_s + 2 * d_ir) / 2)          # cell 5: mesh_xmin (the materials define the real extent)
    z1 = z0 + d_diam
    z2 = z1 + d_ins_p
    z3 = z2 + d_ir
    z4 = z3 + d_s
    z5 = z4 + d_ir
    z6 = z5 + d_ins_o
    z7 = z6 + d_diam
    rmax = r_s + r_gask + r_insg
    spec = [("pside diamond", [z0, z1, 0, rmax], 3500 * 510, 2000, 1e-6),
            ("pside ins", [z1, z2, 0, r_s], 4131 * 668, 10, 0.1e-6),
            ("pside ir", [z2, z3, 0, r_s], 26504 * 130, 352, 0.02e-6),
            ("sample", [z3, z4, 0, r_s], 5164 * 1158, 3.8, 0.08e-6),
            ("oside ir", [z4, z5, 0, r_s], 26504 * 130, 352, 0.02e-6),
            ("oside ins", [z5, z6, 0, r_s], 4131 * 668, 10, 0.1e-6),
            ("oside diamond", [z6, z7, 0, rmax], 3500 * 510, 2000, 1e-6),
            ("gside ins", [z1, z6, r_s, r_s + r_insg], 4131 * 668, 10, 0.02e-6),
            ("gasket", [z1, z6, r_s + r_insg, rmax], 21000 * 140, 100, 1e-6)]
    mats = [Material(nm, b, {"rho_cv": float(rc), "k": float(k)}, h * scale) for nm, b, rc, k, h in spec]
    return mats, [z0, z7, 0.0, rmax], z2, z5, r_s


def notebook_rmse(times, pside, oside):
    """Cells 11, 21, 22: each o-side curve normalised by its own p-side span, the simulation interpolated
    onto the experiment's times; returns (rmse, max error)."""
    exp = np.genfromtxt(HEATING_CSV, delimiter=",", names=True)
    te, pe, oe = exp["time"], exp["temp"], exp["oside"]
    oe_n = (oe - oe[0]) / (pe.max() - pe.min())
    nos = (oside - oside[0]) / (pside.max() - pside.min())
    err = np.interp(te, times, nos) - oe_n
    return float(np.sqrt(np.mean(err ** 2))), float(np.abs(err).max())
