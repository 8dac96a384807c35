"""The only output of the real FEniCS path that the reference holds (clean_with_ir.ipynb cell 22:
RMSE = 0.015309, Max error = 0.030814 of the normalised o-side curve against the Geballe data),
reproduced as a PHYSICS-LEVEL check - see tests/helpers.py for what is restated and what stands in
(the notebook's input CSV is absent; its gmsh mesh cannot be rebuilt).

What the numbers say: the max error is one sample of (simulation - experiment), so it pins the simulated
o-side temperature at that time: reproduced to 4 digits (0.03085 here on our mesh at the notebook's sizes,
0.030814 at half those sizes, 0.030814 in the notebook).  The RMSE comes out 0.0141 against 0.015309 (8 %
apart; sqrt(51/43) apart to 4 digits, i.e. as if the mean ran over 43 of the 51 rows): the notebook's state
is inconsistent and its CSV is not available, so this stays unexplained and the tolerance on it is wide.
This is evidence, not a parity pin: parity stays "unpinned" (DESIGN.md section 5).
"""
import numpy as np
import pytest

from conftest import HEATING_CSV
from helpers import NOTEBOOK_MAX_ERR, NOTEBOOK_RMSE, notebook_clean_with_ir_case, notebook_rmse

MAX_ERR_TOL = 5e-4      # |max error - 0.030814|: measured 4e-5 (mesh sizes x1), 1.6e-4 (x2)
RMSE_TOL = 1.5e-3       # |RMSE - 0.015309|: measured 1.24e-3 (unexplained, see above)


def _setup(scale):
    from heatflow_amd.mesh import Mesh
    from oracle import heat_oracle as ho

    mats, bounds, heated_z, oside_z, r_s = notebook_clean_with_ir_case(scale)
    mesh = Mesh("nb.msh", bounds, mats).build_mesh()
    h_time, h_temp = ho.read_heating_csv(HEATING_CSV)
    nsteps = 200
    dt = h_time.max() / nsteps                    # cell 18: time_stop = max(df_exact['time']), 200 steps
    return mats, mesh, heated_z, oside_z, r_s, h_time, h_temp, nsteps, dt


def test_oracle_reproduces_the_notebook_numbers_on_a_coarse_mesh():
    """Oracle (reference algorithm on the CPU), every notebook mesh size x 2 (60 k nodes)."""
    from oracle import heat_oracle as ho
    from scipy.spatial import cKDTree

    mats, mesh, heated_z, oside_z, r_s, h_time, h_temp, nsteps, dt = _setup(2.0)
    c = mesh.coords
    ic, fwhm = 300.0, 13.2e-6
    tk = {mesh.material_tags[m.name]: m.properties["k"] for m in mats}
    trc = {mesh.material_tags[m.name]: m.properties["rho_cv"] for m in mats}
    bcs = [{"dofs": ho.locate_row_dofs(c, "left"), "value": ic}, {"dofs": ho.locate_row_dofs(c, "right"), "value": ic},
           {"dofs": ho.locate_row_dofs(c, "top"), "value": ic},     # the notebook's 'bottom' (y = -100 um) is r = r_max
           {"dofs": ho.locate_row_dofs(c, "x", coord=heated_z, length=2 * r_s, center=0.0),
            "value": lambda r, t: ho.gaussian_bc_values(r, t, h_time, h_temp, ic, fwhm)}]
    sol = ho.OracleSolver(c, mesh.tris, mesh.tags, tk, trc, dt, bcs, np.full(len(c), ic))
    tree = cKDTree(c)
    wp = [tree.query((heated_z, 0.0))[1], tree.query((oside_z, 0.0))[1]]
    times, ps, os_ = [0.0], [ic], [ic]            # cell 19 writes the initial state at t = 0
    for s in range(nsteps):
        u = sol.step((s + 1) * dt)
        times.append((s + 1) * dt)
        ps.append(u[wp[0]])
        os_.append(u[wp[1]])
    rmse, emax = notebook_rmse(np.array(times), np.array(ps), np.array(os_))
    print(f"oracle, sizes x2: RMSE = {rmse:.6f}  Max error = {emax:.6f}  (notebook {NOTEBOOK_RMSE} / {NOTEBOOK_MAX_ERR})")
    assert abs(emax - NOTEBOOK_MAX_ERR) <= MAX_ERR_TOL
    assert abs(rmse - NOTEBOOK_RMSE) <= RMSE_TOL


@pytest.mark.gpu
def test_hip_path_reproduces_the_notebook_numbers_at_the_notebook_mesh_sizes(hip):
    """HIP path through the C ABI at the notebook's own mesh sizes (266 k nodes on our mesher; gmsh made
    426 k), all 200 steps in one hf_run."""
    from heatflow_amd.bc import P1Space, RowDirichletBC
    from heatflow_amd.heating import HeatingCurve
    from heatflow_amd.solver import HeatProblem, nearest_nodes

    mats, mesh, heated_z, oside_z, r_s, h_time, h_temp, nsteps, dt = _setup(1.0)
    ic = 300.0
    heat = HeatingCurve(HEATING_CSV, ic, 13.2e-6)
    V = P1Space(mesh.coords)
    bcs = [RowDirichletBC(V, "left", value=ic), RowDirichletBC(V, "right", value=ic), RowDirichletBC(V, "top", value=ic),
           RowDirichletBC(V, "x", coord=heated_z, length=2 * r_s, center=0.0, value=heat.gaussian)]
    assert len(bcs[3].row_dofs) == 1001            # cell 17: "Row BC #3 ... (n = 1001 DOFs)"
    tk = {mesh.material_tags[m.name]: m.properties["k"] for m in mats}
    trc = {mesh.material_tags[m.name]: m.properties["rho_cv"] for m in mats}
    prob = HeatProblem(mesh.coords, mesh.tris, mesh.tags, tk, trc, dt, bcs, ic, precond=1)
    try:
        nodes = nearest_nodes(mesh.coords, [(heated_z, 0.0), (oside_z, 0.0)])
        times, samples, iters = prob.run(nsteps, watcher_nodes=nodes, time_varying=[bcs[3]])
    finally:
        prob.close()
    times = np.concatenate([[0.0], times])
    ps = np.concatenate([[ic], samples[:, 0]])
    os_ = np.concatenate([[ic], samples[:, 1]])
    rmse, emax = notebook_rmse(times, ps, os_)
    print(f"HIP, notebook sizes: RMSE = {rmse:.6f}  Max error = {emax:.6f}  (notebook {NOTEBOOK_RMSE} / {NOTEBOOK_MAX_ERR}); "
          f"{len(mesh.coords)} nodes, PCG iterations/step mean {np.mean(iters):.1f}")
    assert abs(emax - NOTEBOOK_MAX_ERR) <= MAX_ERR_TOL
    assert abs(rmse - NOTEBOOK_RMSE) <= RMSE_TOL
