"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (float64):
  * matrix entries (M, A, A_hat)          relative 1e-13 of the row's largest entry
  * SpMV                                   relative 1e-14 (products are summed in CSR order)
  * temperature field, every time step     |T_hip - T_oracle| <= 1e-4 K  (fields are 300..2700 K,
    i.e. ~4e-8 relative; measured ~3e-6 K at the default PCG rtol = 1e-10)
"""
import os
import re
import numpy as np
import pytest

from conftest import ROOT
from helpers import csr_values_on_pattern, make_problem, material_tables, oracle_run, reference_bcs

pytestmark = pytest.mark.gpu

FIELD_TOL_K = 1e-4


def _rel_row_err(dev, ref, rowptr):
    scale = np.maximum.reduceat(np.abs(ref), rowptr[:-1])
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    return np.max(np.abs(dev - ref) / scale[rows])


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("case", ["with_diamond", "no_diamond"])
def test_assembly_matches_oracle(hip, case, mode, case_with_diamond_small, case_no_diamond_small):
    from oracle import heat_oracle as ho

    cfg, stack, mesh = case_with_diamond_small if case == "with_diamond" else case_no_diamond_small
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    with hip.HeatflowHIP(0) as be:
        be.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        tags = sorted(tag_to_k)
        be.set_materials(tags, [tag_to_k[t] for t in tags], [tag_to_rc[t] for t in tags])
        be.assemble(dt, mode)
        rowptr, colidx, A, M = be.get_csr()
    kappa, rho_c = ho.cell_coefficients(mesh.tags, tag_to_k, tag_to_rc)
    Me, Ke = ho.element_matrices(mesh.coords, mesh.tris.astype(np.int64), rho_c, kappa)
    M_ref = ho.assemble_csr(len(mesh.coords), mesh.tris.astype(np.int64), Me)
    A_ref = ho.assemble_csr(len(mesh.coords), mesh.tris.astype(np.int64), Me + dt * Ke)
    # same pattern
    assert np.array_equal(rowptr, M_ref.indptr) and np.array_equal(colidx, M_ref.indices)
    assert _rel_row_err(M, M_ref.data, rowptr) < 1e-13
    assert _rel_row_err(A, A_ref.data, rowptr) < 1e-13
    # symmetry is exact: both triangle contributions commute
    import scipy.sparse as sp
    Ad = sp.csr_matrix((A, colidx, rowptr))
    assert abs(Ad - Ad.T).max() == 0.0


@pytest.mark.parametrize("mode", [1, 3])
def test_deterministic_assembly_modes_are_bitwise_reproducible(hip, mode, case_with_diamond_small):
    cfg, stack, mesh = case_with_diamond_small
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    tags = sorted(tag_to_k)
    out = []
    for _ in range(2):
        with hip.HeatflowHIP(0) as be:
            be.set_mesh(mesh.coords, mesh.tris, mesh.tags)
            be.set_materials(tags, [tag_to_k[t] for t in tags], [tag_to_rc[t] for t in tags])
            be.assemble(1e-7, mode)
            out.append(be.get_csr())
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])


def _fan_mesh(nfan, ntags):
    """A closed fan of `nfan` triangles around node 0 (row 0 holds nfan + 1 entries) next to a strip of quads
    cut into triangles that carry `ntags` different cell tags."""
    ang = 2 * np.pi * np.arange(nfan) / nfan
    pts = [[5e-6, 5e-6]] + [[5e-6 + 1e-6 * np.cos(a), 5e-6 + 1e-6 * np.sin(a)] for a in ang]
    tris = [[0, 1 + q, 1 + (q + 1) % nfan] for q in range(nfan)]
    tags = [1] * nfan
    base = len(pts)
    for q in range(ntags + 1):
        pts += [[20e-6 + q * 1e-6, 1e-6], [20e-6 + q * 1e-6, 2e-6]]
    for q in range(ntags):
        a, b, c, d = base + 2 * q, base + 2 * q + 1, base + 2 * q + 2, base + 2 * q + 3
        tris += [[a, c, d], [a, d, b]]
        tags += [2 + q, 2 + q]
    return np.array(pts), np.array(tris, dtype=np.int32), np.array(tags, dtype=np.int32)


@pytest.mark.parametrize("nfan,ntags", [(31, 3), (32, 3), (8, 70)])
def test_row_gather_limits_and_its_fallback(hip, nfan, ntags):
    """Row gather packs row positions into 5 bits and the cell tag into a 6-bit dictionary index: a row of 32
    entries and 64 tags are the limits (first case: exactly at the row limit); beyond them HF_ASM_ROW_GATHER
    runs the coloured LDS kernel instead.  Either way the matrices must equal the oracle's."""
    from oracle import heat_oracle as ho
    import scipy.sparse as sp

    coords, tris, tags = _fan_mesh(nfan, ntags)
    utags = sorted(set(tags.tolist()))
    tk = {t: 1.0 + 0.37 * t for t in utags}
    trc = {t: 2.0e6 + 1.0e4 * t for t in utags}
    dt = 1e-7
    with hip.HeatflowHIP(0) as be:
        be.set_mesh(coords, tris, tags)
        be.set_materials(utags, [tk[t] for t in utags], [trc[t] for t in utags])
        be.assemble(dt, hip.ASM_ROW_GATHER)
        rowptr, colidx, A, M = be.get_csr()
    assert np.diff(rowptr).max() == nfan + 1
    kappa, rho_c = ho.cell_coefficients(tags, tk, trc)
    Me, Ke = ho.element_matrices(coords, tris.astype(np.int64), rho_c, kappa)
    M_ref = ho.assemble_csr(len(coords), tris.astype(np.int64), Me)
    A_ref = ho.assemble_csr(len(coords), tris.astype(np.int64), Me + dt * Ke)
    assert np.array_equal(colidx, M_ref.indices)
    assert _rel_row_err(M, M_ref.data, rowptr) < 1e-13 and _rel_row_err(A, A_ref.data, rowptr) < 1e-13
    Ad = sp.csr_matrix((A, colidx, rowptr))
    assert abs(Ad - Ad.T).max() == 0.0


def test_dirichlet_elimination_and_spmv(hip, case_no_diamond_small):
    from oracle import heat_oracle as ho

    cfg, stack, mesh = case_no_diamond_small
    prob = make_problem(cfg, stack, mesh)
    try:
        res = oracle_run(cfg, mesh, 0)
        sol = res["solver"]
        assert np.array_equal(prob.bc_dofs, sol.bc_dofs)
        rowptr, colidx, A, M = prob.backend.get_csr()
        Ahat_ref = csr_values_on_pattern(sol.Ahat, rowptr, colidx)
        assert _rel_row_err(A, Ahat_ref, rowptr) < 1e-13
        assert _rel_row_err(M, sol.M.data, rowptr) < 1e-13
        rng = np.random.default_rng(7)
        x = rng.standard_normal(len(mesh.coords))
        import scipy.sparse as sp
        Ad = sp.csr_matrix((A, colidx, rowptr))
        y_ref = Ad @ x
        y = prob.backend.spmv(x, 0)
        assert np.max(np.abs(y - y_ref)) <= 1e-14 * np.max(np.abs(y_ref))
        Md = sp.csr_matrix((M, colidx, rowptr))
        assert np.max(np.abs(prob.backend.spmv(x, 1) - Md @ x)) <= 1e-14 * np.max(np.abs(Md @ x))
    finally:
        prob.close()


@pytest.mark.parametrize("precond", [0, 1])
@pytest.mark.parametrize("case", ["with_diamond", "no_diamond"])
def test_time_loop_matches_oracle_every_step(hip, case, precond, case_with_diamond_small, case_no_diamond_small):
    cfg, stack, mesh = case_with_diamond_small if case == "with_diamond" else case_no_diamond_small
    nsteps = 16
    ref = oracle_run(cfg, mesh, nsteps)
    prob = make_problem(cfg, stack, mesh, precond=precond)
    try:
        for bc in prob.bcs:
            bc.update(0.0)
        worst = 0.0
        for k in range(nsteps):
            t = (k + 1) * prob.dt
            it, res = prob.step(t, only=[prob.bcs[3]])
            u = prob.state()
            err = np.max(np.abs(u - ref["fields"][k]))
            worst = max(worst, err)
            assert err <= FIELD_TOL_K, f"step {k}: |dT| = {err:.3e} K after {it} iterations"
        # the heated steps must really have been solved iteratively
        assert max(prob.iters) > (10 if precond == 0 else 3)
        assert ref["fields"][-1].max() > 310.0
        if precond == 1:
            info = prob.backend.amg_info()
            assert info["levels"] >= 2 and info["op_complexity"] < 2.0
        print(f"{case} precond={precond}: worst |dT| = {worst:.3e} K, iterations/step = {prob.iters}")
    finally:
        prob.close()


def test_run_batch_equals_stepwise_and_watchers(hip, case_with_diamond_small):
    from heatflow_amd.geometry import watcher_points
    from heatflow_amd.solver import nearest_nodes

    cfg, stack, mesh = case_with_diamond_small
    wp = watcher_points(cfg)
    nodes = nearest_nodes(mesh.coords, list(wp.values()))
    nsteps = 12
    ref = oracle_run(cfg, mesh, nsteps, keep_fields=False, watcher_nodes=nodes)
    prob = make_problem(cfg, stack, mesh)
    try:
        times, samples, iters = prob.run(nsteps, watcher_nodes=nodes, time_varying=[prob.bcs[3]])
        assert np.allclose(times, ref["times"], rtol=0, atol=1e-20)
        assert np.max(np.abs(samples - ref["watchers"])) <= FIELD_TOL_K
        assert len(iters) == nsteps
    finally:
        prob.close()


def test_constant_field_is_preserved_before_heating_starts(hip, case_with_diamond_small):
    """For t < 3.566e-7 s every BC value is ic_temp, so u stays 300 (K 1 = 0, A 1 = M 1)."""
    cfg, stack, mesh = case_with_diamond_small
    prob = make_problem(cfg, stack, mesh)
    try:
        for bc in prob.bcs:
            bc.update(0.0)
        for k in range(4):
            it, _ = prob.step((k + 1) * prob.dt)
            assert it == 0
            assert np.max(np.abs(prob.state() - 300.0)) < 1e-9
    finally:
        prob.close()


def test_kappa_update_reuses_pattern(hip, case_with_diamond_small):
    cfg, stack, mesh = case_with_diamond_small
    import copy
    prob = make_problem(cfg, stack, mesh)
    try:
        prob.backend.update_kappa([mesh.material_tags["p_sample"]], [4.4])      # hf_update_kappa
        with pytest.raises(ValueError):
            prob.backend.update_kappa([77], [1.0])                               # not a cell tag of this mesh
        cfg2 = copy.deepcopy(cfg)
        cfg2["mats"]["p_sample"]["k"] = 4.4
        ref = oracle_run(cfg2, mesh, 10)
        for bc in prob.bcs:
            bc.update(0.0)
        for k in range(10):
            prob.step((k + 1) * prob.dt)
        assert np.max(np.abs(prob.state() - ref["fields"][-1])) <= FIELD_TOL_K
    finally:
        prob.close()


def test_error_behaviour(hip, case_no_diamond_small):
    cfg, stack, mesh = case_no_diamond_small
    with hip.HeatflowHIP(0) as be:
        with pytest.raises(hip.HipError):
            be.assemble(1e-7, 0)                      # no mesh yet
        bad = mesh.tris.copy()
        bad[0, 0] = len(mesh.coords) + 5
        with pytest.raises(ValueError):
            be.set_mesh(mesh.coords, bad, mesh.tags)   # index out of range is caught on the host
        be.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        with pytest.raises(ValueError):
            be.set_materials([1], [1.0], [1.0])        # tags 2..5 unmapped
        with pytest.raises(ValueError):
            be.set_dirichlet([0, 0])                   # duplicates must be resolved by the caller
        with pytest.raises(hip.HipError):
            be.step(np.zeros(0))                       # not assembled


def test_not_converged_is_reported(hip, case_no_diamond_small):
    cfg, stack, mesh = case_no_diamond_small
    prob = make_problem(cfg, stack, mesh, max_it=4)
    try:
        for bc in prob.bcs:
            bc.update(0.0)
        with pytest.raises(hip.NotConverged):
            for k in range(8):
                prob.step((k + 1) * prob.dt)
    finally:
        prob.close()


def test_multigrid_loop_reports_non_convergence_and_recovers(hip, case_with_diamond_small):
    """The polled multigrid loop with too few iterations allowed: NotConverged with the iteration count in the message;
    the same context then runs on normally (the progress mirror of the failed solve does not leak into the next one:
    a stale `done` would end it after zero iterations, a stale counter would stall it)."""
    cfg, stack, mesh = case_with_diamond_small
    prob = make_problem(cfg, stack, mesh, precond=1)
    try:
        prob.run(8, time_varying=[prob.bcs[3]])
        g = prob.bc_values(9 * prob.dt, [prob.bcs[3]])
        with pytest.raises(hip.NotConverged, match="not converged in 2 iterations"):
            prob.backend.step(g, prob.rtol, 0.0, 2)
        g = prob.bc_values(10 * prob.dt, [prob.bcs[3]])
        it, res = prob.backend.step(g, prob.rtol, 0.0, prob.max_it)
        assert 3 <= it <= 40 and res <= prob.rtol
        _, _, iters = prob.run(4, time_varying=[prob.bcs[3]], first_step=10)
        assert (np.asarray(iters) >= 3).all() and np.isfinite(prob.state()).all()
    finally:
        prob.close()


@pytest.mark.parametrize("fine_level", ["default", "0", "1", "2"])
def test_amg_cuts_iterations_and_frozen_hierarchy_survives_a_kappa_change(hip, case_with_diamond_small, fine_level, monkeypatch):
    """Same answer with ~10x fewer iterations; with reuse=True a kappa change re-values only the
    fine operator (coarse levels frozen) and the result still matches the oracle.  Every form of the finest level
    (HEATFLOW_AMG_FUSE0, with the LDS-staged kernel admitted on this small mesh so that the fused legs really run): a
    fused down leg alone holds the operator it was built from and must hand over to the explicit one after the
    re-valuation, or the cycle loses its symmetry and PCG breaks down."""
    import copy
    if fine_level != "default":
        monkeypatch.setenv("HEATFLOW_STREAM_MIN_ROWS", "1000")
        monkeypatch.setenv("HEATFLOW_AMG_FUSE0", fine_level)
    cfg, stack, mesh = case_with_diamond_small
    pj = make_problem(cfg, stack, mesh, precond=0)
    pa = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)
    try:
        for prob in (pj, pa):
            for bc in prob.bcs:
                bc.update(0.0)
            for k in range(10):
                prob.step((k + 1) * prob.dt)
        assert np.max(np.abs(pj.state() - pa.state())) <= 2e-5
        assert max(pa.iters) * 4 < max(pj.iters)
        setup_before = pa.backend.amg_info()["setup_s"]
        tag_to_k, tag_to_rc = material_tables(stack, mesh)
        tag_to_k = dict(tag_to_k)
        tag_to_k[mesh.material_tags["p_sample"]] = 4.3
        pa.set_materials(tag_to_k, tag_to_rc)
        assert pa.backend.amg_info()["setup_s"] == setup_before       # not rebuilt
        pa.set_state(300.0)
        cfg2 = copy.deepcopy(cfg)
        cfg2["mats"]["p_sample"]["k"] = 4.3
        ref = oracle_run(cfg2, mesh, 10)
        for bc in pa.bcs:
            bc.update(0.0)
        n_before = len(pa.iters)
        for k in range(10):
            pa.step((k + 1) * pa.dt)
        assert np.max(np.abs(pa.state() - ref["fields"][-1])) <= FIELD_TOL_K
        assert max(pa.iters[n_before:]) <= 3 * max(pa.iters[:n_before]) + 5     # still a preconditioner, not a fallback crawl
        assert pa.backend.amg_info()["jacobi_fallbacks"] == 0
    finally:
        pj.close()
        pa.close()


def test_flux_projection_of_a_field_gone_flat_after_a_warm_start(hip, case_no_diamond_small):
    """Second simulation on one context (a sweep re-uses it): the state is reset to the uniform initial temperature, so
    grad T = 0 and the projection's right-hand side is exactly zero, while its warm start still holds the last gradient
    of the previous simulation.  The solve must return (numerically) zero - the stopping rule falls back to the start
    residual when ||b|| = 0 - and not run to max_it (the failure every point after the first of a no-diamond sweep
    showed: 'PCG not converged in 5000 iterations')."""
    cfg, stack, mesh = case_no_diamond_small
    prob = make_problem(cfg, stack, mesh)
    try:
        prob.backend.flux_setup()
        prob.run(8, time_varying=[prob.bcs[3]])
        for comps, kw in ((2, dict(want_z=False)), (3, dict())):
            it = prob.backend.flux_solve(prob.rtol, 5000, **kw)
            g_hot = prob.backend.flux_sample(np.arange(0, prob.n, 7, dtype=np.int32), **kw)[1]
            assert np.abs(g_hot).max() > 1e3 and max(np.atleast_1d(it)) >= 3          # a real gradient is in the warm start
            prob.set_state(300.0)
            it0 = prob.backend.flux_solve(prob.rtol, 5000, **kw)                       # b = 0 exactly
            g_flat = prob.backend.flux_sample(np.arange(0, prob.n, 7, dtype=np.int32), **kw)[1]
            assert max(np.atleast_1d(it0)) < 200
            assert np.abs(g_flat).max() <= 1e-8 * np.abs(g_hot).max()
            prob.run(8, time_varying=[prob.bcs[3]])                                      # heat up again for the next variant
    finally:
        prob.close()


def test_read_flux_projection_matches_oracle(hip, case_no_diamond_small):
    """run_no_diamond's per-step L2 projection of grad T (r-weighted vector-P1 mass system)."""
    from oracle import heat_oracle as ho

    cfg, stack, mesh = case_no_diamond_small
    ref = oracle_run(cfg, mesh, 8)
    proj = ho.GradientProjector(mesh.coords, mesh.tris)
    prob = make_problem(cfg, stack, mesh)
    try:
        prob.backend.flux_setup()
        for bc in prob.bcs:
            bc.update(0.0)
        for k in range(8):
            prob.step((k + 1) * prob.dt)
            g_ref = proj.project(ref["fields"][k])
            scale = max(np.abs(g_ref).max(), 1.0)                     # K/m (the field is still flat at step 0)
            # end to end: the 1e-6 K solver difference is amplified by 1/h ~ 1e8 /m
            gz, gr = prob.backend.flux_project(rtol=1e-12)
            assert np.abs(gr - g_ref[:, 1]).max() <= 1e-4 * scale + 1e-3
            # the projection itself, on identical input
            prob.set_state(ref["fields"][k])
            gz, gr = prob.backend.flux_project(rtol=1e-12)
            # 1e-3 K/m absolute: roundoff of a 300 K field (1e-13 K) over h ~ 1e-8 m is ~1e-5 K/m
            assert np.abs(gz - g_ref[:, 0]).max() <= 1e-8 * scale + 1e-3
            assert np.abs(gr - g_ref[:, 1]).max() <= 1e-8 * scale + 1e-3
        assert scale > 1e6                                            # steep gradients near the heated line
        assert prob.backend.last_flux_iters.max() < 200
    finally:
        prob.close()


@pytest.mark.parametrize("module,name", [("run_with_diamond", "geballe_with_diamond"), ("run_no_diamond", "geballe_no_diamond")])
def test_entry_points_end_to_end_on_gpu(hip, tmp_path, module, name):
    """The reference-named entry points on the real backend: watcher CSV (and, for run_no_diamond,
    the read-flux CSVs) against the oracle run on the mesh the driver cached."""
    import csv
    import importlib
    import os

    from conftest import load_cfg
    from heatflow_amd.geometry import scale_mesh_sizes, watcher_points
    from heatflow_amd.mesh import load_mesh_arrays
    from heatflow_amd.solver import nearest_nodes
    from oracle import heat_oracle as ho
    import yaml

    run = importlib.import_module(module)
    cfg = scale_mesh_sizes(load_cfg(name), 8.0)
    cfg["timing"]["num_steps"] = 12
    cfg["timing"]["t_final"] = 12 * (7.5e-8 if "with" in name else 1.875e-7)
    mesh_folder, out = str(tmp_path / "mesh"), str(tmp_path / "out")
    wp = watcher_points(cfg)
    res = run.run_simulation(cfg, mesh_folder, rebuild_mesh=True, output_folder=out, watcher_points=wp,
                             write_xdmf=False, suppress_print=True)
    coords, tris, tags = load_mesh_arrays(os.path.join(mesh_folder, "mesh.msh"))
    with open(os.path.join(mesh_folder, "mesh_cfg.yaml")) as f:
        mtags = yaml.safe_load(f)["material_tags"]
    nodes = nearest_nodes(coords, list(wp.values()))
    from conftest import HEATING_CSV
    ref = ho.run_reference_algorithm(cfg, coords, tris, tags, mtags, HEATING_CSV, keep_fields=True, watcher_nodes=nodes)
    with open(os.path.join(out, "watcher_points.csv")) as f:
        rows = list(csv.reader(f))
    got = np.array(rows[1:], dtype=float)
    assert rows[0] == ["time", "pside", "oside"]
    assert np.allclose(got[:, 0], ref["times"], rtol=0, atol=1e-20)
    assert np.abs(got[:, 1:] - ref["watchers"]).max() <= FIELD_TOL_K
    assert res["iters"].max() < 40                                     # multigrid-preconditioned by default
    if module == "run_no_diamond":
        proj = ho.GradientProjector(coords, tris)
        with open(os.path.join(out, "radial_gradient_raw.csv")) as f:
            raw = list(csv.reader(f))
        zcol = np.array(raw[0][1:], dtype=float)
        axis = np.nonzero(np.abs(coords[:, 1]) <= 1e-12)[0]
        axis = axis[np.argsort(coords[axis, 0], kind="stable")]
        assert np.array_equal(zcol, coords[axis, 0])
        g_last = proj.project(ref["fields"][-1])[axis, 1]
        got_last = np.array(raw[-1][1:], dtype=float)
        assert np.abs(got_last - g_last).max() <= 1e-4 * max(np.abs(g_last).max(), 1.0) + 1e-3
        assert os.path.isfile(os.path.join(out, "radial_gradient.csv"))


def test_kappa_sweep_on_gpu_reuses_mesh_and_hierarchy(hip, tmp_path):
    """sweep_test.py analogue on the real backend (world of 1): one session, three kappa values;
    every point must match an oracle run of the same cfg, with the multigrid levels built once."""
    import copy

    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd import parameter_sweep as ps
    from heatflow_amd.geometry import scale_mesh_sizes, watcher_points
    from heatflow_amd.mesh import load_mesh_arrays
    from heatflow_amd.solver import nearest_nodes
    from oracle import heat_oracle as ho
    import yaml, os

    cfg = scale_mesh_sizes(load_cfg("geballe_with_diamond"), 8.0)
    cfg["heating"]["file"] = HEATING_CSV
    cfg["timing"]["num_steps"] = 10
    cfg["timing"]["t_final"] = 10 * 7.5e-8
    mesh_folder, out = str(tmp_path / "mesh"), str(tmp_path / "out")
    rows = ps.run_kappa_sweep(cfg, mesh_folder, [3.3, 3.8, 4.3], out, rebuild_mesh=True)
    assert [r["status"] for r in rows] == ["success"] * 3 and [r["k"] for r in rows] == [3.3, 3.8, 4.3]
    coords, tris, tags = load_mesh_arrays(os.path.join(mesh_folder, "mesh.msh"))
    mtags = yaml.safe_load(open(os.path.join(mesh_folder, "mesh_cfg.yaml")))["material_tags"]
    nodes = nearest_nodes(coords, list(watcher_points(cfg).values()))
    for k in (3.3, 4.3):
        c = copy.deepcopy(cfg)
        c["mats"]["p_sample"]["k"] = k
        ref = ho.run_reference_algorithm(c, coords, tris, tags, mtags, HEATING_CSV, watcher_nodes=nodes)
        got = np.genfromtxt(os.path.join(out, f"{k:.2f}", "watcher_points.csv"), delimiter=",", names=True)
        assert np.abs(got["oside"] - ref["watchers"][:, 1]).max() <= FIELD_TOL_K
        assert np.abs(got["pside"] - ref["watchers"][:, 0]).max() <= FIELD_TOL_K
    assert os.path.isfile(os.path.join(out, "rmse_summary.csv"))


def test_no_diamond_grid_sweep_every_point_succeeds_on_one_context(hip, tmp_path):
    """parameter_sweep.run_parameter_sweep on the no-diamond read-flux configuration (the reference's production sweep,
    parameter_sweep.py:289-536): fwhm x k grid on one mesh, every point after the first re-uses the solver context - and
    with it the warm start of the flux projection, which meets a zero right-hand side in the new run's first steps."""
    import csv
    import os
    import yaml

    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd import parameter_sweep as ps
    from heatflow_amd.geometry import scale_mesh_sizes

    cfg = scale_mesh_sizes(load_cfg("geballe_no_diamond_read_flux"), 3.0)
    cfg["heating"]["file"] = HEATING_CSV
    cfg["timing"]["num_steps"] = 12
    cfg["timing"]["t_final"] = 12 * 1.875e-7
    cfg_path = str(tmp_path / "cfg.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    out = str(tmp_path / "out")
    ps.run_parameter_sweep(cfg_path, out, (1.0e-5, 1.4e-5), (3.4, 4.2), (1.9e-6, 1.9e-6), (2, 3, 1),
                           base_mesh_folder=str(tmp_path / "meshes"))
    with open(os.path.join(out, "successful_runs.csv")) as f:
        rows = list(csv.DictReader(f))
    failed = os.path.join(out, "failed_runs.csv")
    assert len(rows) == 6, open(failed).read() if os.path.isfile(failed) else rows
    for r in rows:
        assert r["status"] == "success" and os.path.isfile(os.path.join(r["output_dir"], "radial_gradient.csv"))


def test_no_diamond_grid_sweep_through_the_batched_loop_matches_the_oracle_at_every_point(hip, tmp_path):
    """The reference's production sweep (parameter_sweep.py:43,157-166,195-235: fwhm x k x width over run_no_diamond with its
    per-step read-flux projection, run_no_diamond.py:543-566,603-617) through the batched time loop: a 2 x 3 grid on one
    width = one batch of 4 and one of 2 columns (affine operator family in k, boundary values per column in fwhm), the
    projection as the columns of one Jacobi-PCG.  Every point: watcher curves and the last row of
    radial_gradient_raw.csv (dT/dr on the axis nodes) against the oracle's run of that point; the files of the batched
    sweep equal those of a point-by-point sweep to solver tolerance."""
    import csv
    import os
    import yaml

    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd import parameter_sweep as ps
    from heatflow_amd.geometry import scale_mesh_sizes, watcher_points
    from heatflow_amd.mesh import load_mesh_arrays
    from heatflow_amd.solver import nearest_nodes
    from oracle import heat_oracle as ho

    cfg = scale_mesh_sizes(load_cfg("geballe_no_diamond_read_flux"), 3.0)
    cfg["heating"]["file"] = HEATING_CSV
    nsteps = 12
    cfg["timing"]["num_steps"] = nsteps
    cfg["timing"]["t_final"] = nsteps * 1.875e-7
    cfg_path = str(tmp_path / "cfg.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    grid = ((1.0e-5, 1.4e-5), (3.4, 4.2), (1.9e-6, 1.9e-6), (2, 3, 1))
    out_b, out_s = str(tmp_path / "out_batched"), str(tmp_path / "out_single")
    ps.run_parameter_sweep(cfg_path, out_b, *grid, base_mesh_folder=str(tmp_path / "meshes"), batch=8)
    ps.run_parameter_sweep(cfg_path, out_s, *grid, base_mesh_folder=str(tmp_path / "meshes"), batch=1)
    with open(os.path.join(out_b, "successful_runs.csv")) as f:
        rows = list(csv.DictReader(f))
    failed = os.path.join(out_b, "failed_runs.csv")
    assert len(rows) == 6, open(failed).read() if os.path.isfile(failed) else rows
    assert sorted(int(r["batch"]) for r in rows) == [2, 2, 4, 4, 4, 4] and not any(r.get("batch_error") for r in rows)
    mesh_folder = ps.get_mesh_folder_for_width(str(tmp_path / "meshes"), 1.9e-6)
    coords, tris, tags = load_mesh_arrays(os.path.join(mesh_folder, "mesh.msh"))
    mtags = yaml.safe_load(open(os.path.join(mesh_folder, "mesh_cfg.yaml")))["material_tags"]
    proj = ho.GradientProjector(coords, tris)
    axis = np.nonzero(np.abs(coords[:, 1]) <= 1e-12)[0]
    axis = axis[np.argsort(coords[axis, 0], kind="stable")]
    seen = set()
    for r in rows:
        c = ps.modify_config_for_parameters(cfg, float(r["fwhm"]), float(r["k"]), float(r["width"]))
        nodes = nearest_nodes(coords, list(watcher_points(c).values()))
        ref = ho.run_reference_algorithm(c, coords, tris, tags, mtags, HEATING_CSV, watcher_nodes=nodes, keep_fields=True)
        got = np.genfromtxt(os.path.join(r["output_dir"], "watcher_points.csv"), delimiter=",", names=True)
        assert len(got) == nsteps
        assert np.abs(got["pside"] - ref["watchers"][:, 0]).max() <= FIELD_TOL_K
        assert np.abs(got["oside"] - ref["watchers"][:, 1]).max() <= FIELD_TOL_K
        with open(os.path.join(r["output_dir"], "radial_gradient_raw.csv")) as f:
            raw = list(csv.reader(f))
        assert len(raw) == nsteps + 1 and np.allclose(np.array(raw[0][1:], dtype=float), coords[axis, 0], rtol=0, atol=0)
        g_ref = proj.project(ref["fields"][-1])[axis, 1]
        g_got = np.array(raw[-1][1:], dtype=float)
        scale = max(np.abs(g_ref).max(), 1.0)
        assert scale > 1e5 and np.abs(g_got - g_ref).max() <= 1e-4 * scale + 1e-3
        # the same point of the point-by-point sweep: same files to solver tolerance
        single = os.path.join(out_s, os.path.basename(r["output_dir"]))
        got_s = np.genfromtxt(os.path.join(single, "watcher_points.csv"), delimiter=",", names=True)
        assert np.abs(got["oside"] - got_s["oside"]).max() <= 1e-5 and np.abs(got["pside"] - got_s["pside"]).max() <= 1e-5
        for name in ("radial_gradient.csv", "radial_gradient_raw.csv"):
            a = np.genfromtxt(os.path.join(r["output_dir"], name), delimiter=",", skip_header=1)
            b = np.genfromtxt(os.path.join(single, name), delimiter=",", skip_header=1)
            assert a.shape == b.shape and np.abs(a - b).max() <= 1e-4 * scale + 1e-3
        seen.add((round(float(r["fwhm"]), 9), round(float(r["k"]), 6)))
    assert len(seen) == 6


def _unit_square_mesh(nz, nr):
    z = np.linspace(0.0, 1.0e-6, nz + 1)
    r = np.linspace(0.0, 2.0e-6, nr + 1)
    Z, R = np.meshgrid(z, r, indexing="ij")
    coords = np.column_stack([Z.ravel(), R.ravel()])
    idx = lambda i, j: i * (nr + 1) + j
    tris = [[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(nz) for j in range(nr)] + \
           [[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(nz) for j in range(nr)]
    return coords, np.array(tris, dtype=np.int32)


@pytest.mark.parametrize("precond", [0, 1])
def test_no_dirichlet_rows_conserves_heat_content(hip, precond):
    """Edge case n_bc = 0 (pure Neumann): K 1 = 0, so 1^T M u is invariant under backward Euler; the
    field relaxes towards its r-weighted mean.  Also exercises hf_run with no samples."""
    from oracle import heat_oracle as ho

    coords, tris = _unit_square_mesh(37, 53)                       # 2052 nodes: not a multiple of any chunk size
    tags = np.where(coords[tris].mean(axis=1)[:, 0] < 0.5e-6, 3, 7).astype(np.int32)    # sparse tag values
    rng = np.random.default_rng(11)
    u0 = 300.0 + 50.0 * rng.random(len(coords))
    dt = 2e-9
    with hip.HeatflowHIP(0) as be:
        be.set_mesh(coords, tris, tags)
        be.set_materials([3, 7], [10.0, 352.0], [2.76e6, 3.44e6])
        be.set_dirichlet([])
        be.set_precond(precond)
        be.assemble(dt, hip.ASM_LDS_COLORED)
        rowptr, colidx, A, M = be.get_csr()
        import scipy.sparse as sp
        Md = sp.csr_matrix((M, colidx, rowptr))
        be.set_state(u0)
        heat0 = (Md @ u0).sum()
        samples, iters = be.run(np.zeros((25, 0)), rtol=1e-12)
        u = be.get_state()
        assert samples.shape == (25, 0) and iters.max() > 0
        assert abs((Md @ u).sum() - heat0) <= 1e-9 * abs(heat0)
        assert u.max() - u.min() < 0.9 * (u0.max() - u0.min())       # diffusing
    sol = ho.OracleSolver(coords, tris, tags, {3: 10.0, 7: 352.0}, {3: 2.76e6, 7: 3.44e6}, dt, [], u0)
    sol.bc_dofs = np.zeros(0, dtype=np.int64)
    for k in range(25):
        b = sol.M @ sol.u
        sol.u = sol.factor().solve(b)
    assert np.abs(u - sol.u).max() <= 1e-6


def test_two_triangle_mesh_and_context_reuse(hip):
    """Smallest possible mesh, then a different mesh on the same context."""
    coords = np.array([[0.0, 0.0], [1e-6, 0.0], [1e-6, 1e-6], [0.0, 1e-6]])
    tris = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    with hip.HeatflowHIP(0) as be:
        be.set_mesh(coords, tris, np.array([1, 1], dtype=np.int32))
        be.set_materials([1], [5.0], [1e6])
        be.set_dirichlet([0])
        be.assemble(1e-9, 0)
        be.set_state(np.full(4, 300.0))
        it, _ = be.step(np.array([400.0]))
        u = be.get_state()
        rowptr, colidx, A, M = be.get_csr()
        assert be.n == 4 and be.nnz == 14 and rowptr.tolist() == [0, 4, 7, 11, 14]
        # dense check of the one step (consistent mass + tiny dt: the far node may undershoot, that is the scheme)
        from oracle import heat_oracle as ho
        sol = ho.OracleSolver(coords, tris, np.array([1, 1]), {1: 5.0}, {1: 1e6}, 1e-9,
                              [{"dofs": np.array([0]), "value": 400.0}], np.full(4, 300.0))
        ref = sol.step(1e-9)
        assert u[0] == 400.0 and it >= 1 and np.abs(u - ref).max() < 1e-8
        # same context, new mesh: everything is rebuilt
        c2, t2 = _unit_square_mesh(8, 8)
        be.set_mesh(c2, t2, np.ones(len(t2), dtype=np.int32))
        with pytest.raises(hip.HipError):
            be.step(np.zeros(0))                                   # materials / assembly of the old mesh are gone
        be.set_materials([1], [5.0], [1e6])
        be.set_dirichlet(np.arange(9))
        be.assemble(1e-9, 1)
        be.set_state(np.full(81, 300.0))
        be.step(np.full(9, 350.0))
        assert be.get_state().max() <= 350.0 + 1e-9


def test_two_contexts_on_two_threads_agree_with_sequential(hip, case_with_diamond_small):
    """Contexts are independent (own stream, stream-ordered copies only): two of them stepped from two
    host threads at once - each polling its own progress mirror - give the sequential results."""
    import threading

    cfg, stack, mesh = case_with_diamond_small

    def run(precond, out, key):
        # coloured assembly: bitwise reproducible matrices (the LDS-atomic default sums diagonals in arrival order)
        prob = make_problem(cfg, stack, mesh, precond=precond, assembly_mode=1)
        try:
            for bc in prob.bcs:
                bc.update(0.0)
            for k in range(10):
                prob.step((k + 1) * prob.dt)
            out[key] = prob.state()
        finally:
            prob.close()

    seq, par = {}, {}
    run(1, seq, "amg")
    run(0, seq, "jac")
    threads = [threading.Thread(target=run, args=(1, par, "amg")), threading.Thread(target=run, args=(0, par, "jac"))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert set(par) == {"amg", "jac"}
    assert np.array_equal(par["amg"], seq["amg"]) and np.array_equal(par["jac"], seq["jac"])   # deterministic reductions


def test_two_sided_heating_extension_matches_oracle(hip, tmp_path):
    """BASELINE config 4's "two-sided heating" has no reference implementation (cfgs/konopkova.yaml is a
    stub); the extension (second Gaussian line on the o-side coupler face, `oside` column) is checked
    against the oracle's restatement of the same extension, together with the read-flux outputs."""
    import os
    import yaml
    import run_no_diamond as run
    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes
    from heatflow_amd.mesh import load_mesh_arrays
    from oracle import heat_oracle as ho

    cfg = scale_mesh_sizes(load_cfg("geballe_no_diamond_read_flux"), 8.0)
    cfg["timing"]["num_steps"] = 14
    cfg["timing"]["t_final"] = 14 * 1.5e-7
    mesh_folder, out = str(tmp_path / "mesh"), str(tmp_path / "out")
    res = run.run_simulation(cfg, mesh_folder, rebuild_mesh=True, output_folder=out, watcher_points=None,
                             write_xdmf=True, suppress_print=True, two_sided=True)
    coords, tris, tags = load_mesh_arrays(os.path.join(mesh_folder, "mesh.msh"))
    mtags = yaml.safe_load(open(os.path.join(mesh_folder, "mesh_cfg.yaml")))["material_tags"]
    stack = build_stack(cfg)
    ref = ho.run_reference_algorithm(cfg, coords, tris, tags, mtags, HEATING_CSV, keep_fields=True,
                                     second_line=stack.heated_z_oside)
    fld = np.fromfile(os.path.join(out, "output_fields.f64"), dtype="<f8").reshape(15, len(coords))
    assert np.abs(fld[1:] - ref["fields"]).max() <= FIELD_TOL_K
    one_sided = ho.run_reference_algorithm(cfg, coords, tris, tags, mtags, HEATING_CSV, keep_fields=True)
    assert np.abs(ref["fields"][-1] - one_sided["fields"][-1]).max() > 0.5        # the second line matters
    assert os.path.isfile(os.path.join(out, "radial_gradient.csv"))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_general_triangles_random_meshes(hip, seed):
    """gmsh meshes hold arbitrary triangles, the built-in mesher only right ones: jittered nodes, mixed
    orientations (CW and CCW), random tags / coefficients / Dirichlet sets, r starting at 0 or off the
    axis.  Matrices, three time steps (both preconditioners) and the gradient projection vs the oracle."""
    from oracle import heat_oracle as ho
    import scipy.sparse as sp

    rng = np.random.default_rng(100 + seed)
    nz, nr = int(rng.integers(9, 30)), int(rng.integers(9, 30))
    coords, tris = _unit_square_mesh(nz, nr)
    hz, hr = 1.0e-6 / nz, 2.0e-6 / nr
    inner = (coords[:, 0] > 0) & (coords[:, 0] < 1.0e-6) & (coords[:, 1] > 0) & (coords[:, 1] < 2.0e-6)
    coords = coords.copy()
    coords[inner, 0] += rng.uniform(-0.3, 0.3, inner.sum()) * hz
    coords[inner, 1] += rng.uniform(-0.3, 0.3, inner.sum()) * hr
    if seed == 1:
        coords[:, 1] += 0.7e-6                                       # annulus: no node on the axis
    flip = rng.random(len(tris)) < 0.5
    tris = tris.copy()
    tris[flip] = tris[flip][:, [0, 2, 1]]                            # clockwise elements
    tris = tris[rng.permutation(len(tris))]                          # no ordering assumption
    tags = rng.integers(1, 5, len(tris)).astype(np.int32)
    tag_to_k = {t: float(10.0 ** rng.uniform(0, 3)) for t in range(1, 5)}
    tag_to_rc = {t: float(10.0 ** rng.uniform(5.5, 7)) for t in range(1, 5)}
    dt = 3e-9
    bc_dofs = np.sort(rng.choice(len(coords), size=int(rng.integers(1, 40)), replace=False)).astype(np.int32)
    u0 = 300.0 + 100.0 * rng.random(len(coords))
    g_fn = lambda t: 300.0 + 1e10 * t + 50.0 * np.sin(np.arange(len(bc_dofs)))
    sol = ho.OracleSolver(coords, tris, tags, tag_to_k, tag_to_rc, dt,
                          [{"dofs": bc_dofs, "value": lambda r, t: g_fn(t)}], u0)
    refs = [sol.step((k + 1) * dt).copy() for k in range(3)]
    proj = ho.GradientProjector(coords, tris)
    for precond in (0, 1):
        with hip.HeatflowHIP(0) as be:
            be.set_mesh(coords, tris, tags)
            tl = sorted(tag_to_k)
            be.set_materials(tl, [tag_to_k[t] for t in tl], [tag_to_rc[t] for t in tl])
            be.set_dirichlet(bc_dofs)
            be.set_precond(precond)
            be.assemble(dt, hip.ASM_LDS_ATOMIC if precond else hip.ASM_LDS_COLORED)
            rowptr, colidx, A, M = be.get_csr()
            assert _rel_row_err(A, csr_values_on_pattern(sol.Ahat, rowptr, colidx), rowptr) < 1e-12
            assert _rel_row_err(M, csr_values_on_pattern(sol.M, rowptr, colidx), rowptr) < 1e-12
            be.set_state(u0)
            for k in range(3):
                be.step(g_fn((k + 1) * dt), rtol=1e-12)
                assert np.abs(be.get_state() - refs[k]).max() <= 1e-6
            be.flux_setup()
            be.set_state(refs[2])
            gz, gr = be.flux_project(rtol=1e-12)
            g_ref = proj.project(refs[2])
            scale = np.abs(g_ref).max()
            assert np.abs(gz - g_ref[:, 0]).max() <= 1e-8 * scale and np.abs(gr - g_ref[:, 1]).max() <= 1e-8 * scale


def test_call_order_state_machine(hip, case_no_diamond_small):
    """Out-of-order and repeated calls return error codes (never crash) and leave the context usable;
    re-assembling with another dt / preconditioner / Dirichlet set between steps gives the oracle's numbers."""
    from oracle import heat_oracle as ho

    cfg, stack, mesh = case_no_diamond_small
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    tl = sorted(tag_to_k)
    n = len(mesh.coords)
    with hip.HeatflowHIP(0) as be:
        for call in (lambda: be.set_materials(tl, [1.0] * 5, [1.0] * 5), lambda: be.set_dirichlet([0]),
                     lambda: be.flux_setup(), lambda: be.get_csr(), lambda: be.time_kernel(hip.K_SPMV, 2)):
            with pytest.raises((hip.HipError, ValueError)):
                call()                                               # nothing works before set_mesh
        be.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        with pytest.raises(hip.HipError):
            be.flux_project()
        with pytest.raises(hip.HipError):
            be.update_kappa([1], [2.0])
        be.set_materials(tl, [tag_to_k[t] for t in tl], [tag_to_rc[t] for t in tl])
        with pytest.raises(ValueError):
            be.assemble(-1.0)
        with pytest.raises(ValueError):
            be.assemble(1e-7, 9)
        be.assemble(1e-7, 1)
        with pytest.raises(ValueError):
            be.step(np.zeros(3))                                     # wrong number of boundary values
        be.set_dirichlet([5, 9, 2])                                  # invalidates the assembled operator
        with pytest.raises(hip.HipError):
            be.step(np.zeros(3))
        with pytest.raises(ValueError):
            be.set_precond(7)
        # a valid sequence with changes between steps, checked against the oracle
        u = np.full(n, 300.0)
        dofs = np.array([5, 9, 2], dtype=np.int32)
        plan = [(1e-7, 0), (1e-7, 1), (3e-8, 1), (3e-8, 0)]
        be.set_state(u)
        for dt, pc in plan:
            be.set_precond(pc)
            be.assemble(dt, 1)
            g = np.array([350.0, 320.0, 400.0])
            be.step(g, rtol=1e-12)
            sol = ho.OracleSolver(mesh.coords, mesh.tris, mesh.tags, tag_to_k, tag_to_rc, dt,
                                  [{"dofs": dofs[[0]], "value": 350.0}, {"dofs": dofs[[1]], "value": 320.0},
                                   {"dofs": dofs[[2]], "value": 400.0}], u)
            u = sol.step(dt)
            assert np.abs(be.get_state() - u).max() <= 1e-6
            be.set_state(u)
        with pytest.raises(ValueError):
            be.sample([n])                                           # node out of range
        assert be.sample([5])[0] == 350.0


@pytest.mark.gpu
@pytest.mark.parametrize("precond", [0, 1])
def test_start_vector_kinds_same_answer_fewer_iterations(hip, precond, case_with_diamond_small):
    """hf_set_start_vector: the converged field does not depend on the start vector; the boundary-response
    correction (kind 2) costs one extra solve per operator (the heated line's profile is the only direction
    the second difference of the boundary values ever has) and needs fewer iterations than plain
    extrapolation (kind 1), which needs fewer than starting from u^n (kind 0); the A-norm projection on the last
    solutions and that response (kind 3, the default) contains both as special cases and needs fewest."""
    cfg, stack, mesh = case_with_diamond_small
    nsteps = 40
    out = {}
    for kind in (0, 1, 2, 3):
        prob = make_problem(cfg, stack, mesh, precond=precond)
        try:
            prob.backend.set_start_vector(kind)
            _, _, iters = prob.run(nsteps, time_varying=[prob.bcs[3]])
            out[kind] = (prob.state(), int(np.sum(iters)), prob.backend.response_solves())
        finally:
            prob.close()
    for kind in (1, 2, 3):
        assert np.abs(out[kind][0] - out[0][0]).max() <= 2e-5, kind
    assert out[0][2] == 0 and out[1][2] == 0 and out[2][2] == 1 and out[3][2] == 1
    assert out[3][1] < out[2][1] < out[1][1] <= out[0][1], {k: v[1] for k, v in out.items()}
    print({k: v[1] for k, v in out.items()})
    if precond == 1:   # multigrid-PCG contracts at a fixed rate per iteration: a 30x smaller start residual shows
        assert out[2][1] <= 0.95 * out[1][1], {k: v[1] for k, v in out.items()}
        assert out[3][1] <= 0.92 * out[2][1], {k: v[1] for k, v in out.items()}
    prob2 = make_problem(cfg, stack, mesh, precond=precond)
    try:
        with pytest.raises(ValueError):
            prob2.backend.set_start_vector(4)
    finally:
        prob2.close()


@pytest.mark.gpu
def test_time_kernel_ids_and_state_preservation(hip, case_with_diamond_small):
    """hf_time_kernel: every live kernel id returns a positive time and leaves the state and the operator
    alone (HF_K_ASSEMBLE excepted: it asks for a re-assembly); the retired id is refused."""
    cfg, stack, mesh = case_with_diamond_small
    prob = make_problem(cfg, stack, mesh, precond=1)
    try:
        prob.run(8, time_varying=[prob.bcs[3]])
        u0 = prob.state()
        A0 = prob.backend.get_csr()[2].copy()
        for k in (hip.K_SPMV, hip.K_PCG_SPMV, hip.K_PCG_UPDATE, hip.K_RHS, hip.K_STREAM_READ):
            assert prob.backend.time_kernel(k, 3) > 0.0
        assert np.array_equal(prob.state(), u0)
        assert np.array_equal(prob.backend.get_csr()[2], A0)
        with pytest.raises(ValueError):
            prob.backend.time_kernel(hip.K_PCG_DIR, 1)
        with pytest.raises(ValueError):
            prob.backend.time_kernel(17, 1)
        it, _ = prob.step(9 * prob.dt, only=[prob.bcs[3]])          # the loop continues unharmed
        assert it < 60
    finally:
        prob.close()


@pytest.mark.gpu
def test_flux_solve_and_sample_match_the_full_projection(hip, case_no_diamond_small):
    """hf_flux_solve + hf_flux_sample (what the driver uses: d/dr at the band and axis nodes only) against
    hf_flux_project's full vectors; a component that was not solved cannot be sampled."""
    cfg, stack, mesh = case_no_diamond_small
    prob = make_problem(cfg, stack, mesh, precond=1)
    try:
        prob.run(12, time_varying=[prob.bcs[3]])
        be = prob.backend
        be.flux_setup()
        gz, gr = be.flux_project(1e-12, 5000)
        nodes = np.random.default_rng(3).integers(0, prob.n, size=257).astype(np.int32)
        it = be.flux_solve(1e-12, 5000, want_z=False)
        assert it[0] == 0 and it[1] >= 0
        _, sr = be.flux_sample(nodes, want_z=False)
        assert np.abs(sr - gr[nodes]).max() <= 1e-9 * np.abs(gr).max()
        with pytest.raises(hip.HipError):
            be.flux_sample(nodes, want_z=True, want_r=False)       # z was skipped by the last solve
        be.flux_solve(1e-12, 5000)
        sz, sr = be.flux_sample(nodes)
        assert np.abs(sz - gz[nodes]).max() <= 1e-9 * np.abs(gz).max()
        assert np.abs(sr - gr[nodes]).max() <= 1e-9 * np.abs(gr).max()
        with pytest.raises(ValueError):
            be.flux_sample(np.array([prob.n], dtype=np.int32))
        assert be.flux_sample(np.zeros(0, dtype=np.int32)) [0].shape == (0,)
    finally:
        prob.close()


@pytest.mark.gpu
def test_compressed_and_plain_column_streams_give_identical_results(hip, tmp_path):
    """The 16-bit column positions (default) and the 32-bit column stream (HEATFLOW_SPMV_C16=0) form the same
    products in the same order: SpMV and a short time loop agree bit for bit.  One subprocess per setting, since
    the switch is read when the mesh is set."""
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(os.path.join(ROOT, 'tests'))!r})\n"
        "from conftest import build_case\n"
        "from helpers import make_problem\n"
        "cfg, stack, mesh = build_case('geballe_with_diamond', 4.0)\n"
        "prob = make_problem(cfg, stack, mesh, precond=1)\n"
        "x = np.random.default_rng(1).standard_normal(prob.n)\n"
        "y = prob.backend.spmv(x, 0)\n"
        "prob.run(10, time_varying=[prob.bcs[3]])\n"
        "np.savez(sys.argv[1], y=y, u=prob.state(), iters=np.array(prob.iters))\n"
        "prob.close()\n")
    out = {}
    for flag in ("1", "0"):
        env = dict(os.environ, HEATFLOW_SPMV_C16=flag)
        res = subprocess.run([sys.executable, str(script), str(tmp_path / f"o{flag}.npz")], env=env, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        out[flag] = np.load(tmp_path / f"o{flag}.npz")
    assert np.array_equal(out["1"]["y"], out["0"]["y"])
    assert np.array_equal(out["1"]["u"], out["0"]["u"])
    assert np.array_equal(out["1"]["iters"], out["0"]["iters"])


@pytest.mark.gpu
def test_polled_and_copied_back_loops_give_identical_results(hip, tmp_path):
    """The multigrid-PCG loops queue iterations one convergence test ahead by polling the host-visible progress mirror
    (default); HEATFLOW_POLL=0 keeps the bursts + copy-back of the scalars.  Same kernels in the same order on the
    same data, so fields and iteration counts agree bit for bit - single-column and batched (affine family).  One
    subprocess per setting: the switch is read once."""
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(os.path.join(ROOT, 'tests'))!r})\n"
        "from conftest import build_case\n"
        "from helpers import make_problem\n"
        "from heatflow_amd import hip_backend as hb\n"
        "cfg, stack, mesh = build_case('geballe_with_diamond', 4.0)\n"
        "prob = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)\n"
        "_, _, it = prob.run(12, time_varying=[prob.bcs[3]])\n"
        "u = prob.state()\n"
        "be = prob.backend\n"
        "be.batch_begin(4, hb.BATCH_SHARED)\n"
        "g = np.stack([np.repeat(prob.bc_values((s + 1) * prob.dt, [prob.bcs[3]])[:, None], 4, axis=1) + np.arange(4) for s in range(12, 18)])\n"
        "for j in range(4): be.batch_set_state(j, u + j)\n"
        "_, bit = be.batch_run(g, prob.rtol, prob.atol, prob.max_it, None)\n"
        "ub = np.stack([be.batch_get_state(j) for j in range(4)])\n"
        "be.batch_end()\n"
        "np.savez(sys.argv[1], u=u, it=np.array(it), ub=ub, bit=np.array(bit))\n"
        "prob.close()\n")
    out = {}
    # "whole": polled, but whole iterations queued to the end (HEATFLOW_HOLD_BACK=0) instead of the last cycles held back until
    # their test asks for them - the launches that differ are the ones that return at their first instruction
    for flag, extra in (("1", {"HEATFLOW_POLL": "1"}), ("0", {"HEATFLOW_POLL": "0"}), ("whole", {"HEATFLOW_POLL": "1", "HEATFLOW_HOLD_BACK": "0"})):
        res = subprocess.run([sys.executable, str(script), str(tmp_path / f"o{flag}.npz")], env=dict(os.environ, **extra), capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        out[flag] = np.load(tmp_path / f"o{flag}.npz")
    for key in ("u", "it", "ub", "bit"):
        assert np.array_equal(out["1"][key], out["0"][key]), key
        assert np.array_equal(out["1"][key], out["whole"][key]), key
    assert out["1"]["it"].max() >= 5 and out["1"]["bit"].max() >= 3


@pytest.mark.gpu
def test_chunk_pipeline_of_the_transfer_operators_on_small_and_oversized_chunks(hip, tmp_path):
    """The LDS-staged kernel runs the single-precision transfer operators with a software pipeline over a workgroup's
    chunks (next chunk's list head and stream in flight, hf_kernels.hpp).  By default only operators of >= 20 000 rows
    take it (stock mesh and larger: test_gpu_fullsize.py); here HEATFLOW_STREAM_MIN_ROWS sends every level of a 50 k-node
    hierarchy through it, with the default chunk limit (4 or 8 stream entries per lane in flight), with tiny chunks (many
    per workgroup, ragged ends) and with chunks beyond what the pipeline keeps in registers (entries past 8 per lane
    and column lists past 1024 entries take the second, unpipelined pass).  Same preconditioner up to the summation
    order inside a row: fields agree to PCG tolerance, iteration counts to +-1.  One subprocess per setting (the
    switches are read once)."""
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(os.path.join(ROOT, 'tests'))!r})\n"
        "from conftest import build_case\n"
        "from helpers import make_problem\n"
        "cfg, stack, mesh = build_case('geballe_with_diamond', 2.0)\n"
        "prob = make_problem(cfg, stack, mesh, precond=1)\n"
        "_, _, it = prob.run(14, time_varying=[prob.bcs[3]])\n"
        "info = prob.backend.amg_info()\n"
        "np.savez(sys.argv[1], u=prob.state(), it=np.array(it), rows=np.array(info['rows']))\n"
        "prob.close()\n")
    settings = {"default": {"HEATFLOW_STREAM_MIN_ROWS": "1000000"}, "stream": {"HEATFLOW_STREAM_MIN_ROWS": "300"},
                "tiny": {"HEATFLOW_STREAM_MIN_ROWS": "300", "HEATFLOW_STREAM_NNZ": "700"},
                "oversized": {"HEATFLOW_STREAM_MIN_ROWS": "300", "HEATFLOW_STREAM_NNZ": "6500"}}
    out, staged = {}, {}
    for name, extra in settings.items():
        res = subprocess.run([sys.executable, str(script), str(tmp_path / f"{name}.npz")],
                             env=dict(os.environ, HEATFLOW_DEBUG="1", **extra), capture_output=True, text=True)
        assert res.returncode == 0, (name, res.stderr[-2000:])
        out[name] = np.load(tmp_path / f"{name}.npz")
        # the set-up's operator table ("[amg] level 1 GP ... stream rpc 64 lanes 32 f32 c16"): which operators the kernel runs
        staged[name] = sorted(set(re.findall(r"\[amg\] level (\d+) (\w+) .* stream rpc (\d+) lanes \d+ f32 c16", res.stdout + res.stderr)))
    ref = out["default"]
    assert ref["it"].max() >= 5 and len(ref["rows"]) >= 3
    assert staged["default"] == [] and len(staged["stream"]) >= 4, staged
    assert max(int(r) for _, _, r in staged["tiny"]) < max(int(r) for _, _, r in staged["stream"]) <= max(int(r) for _, _, r in staged["oversized"]), staged
    for name in ("stream", "tiny", "oversized"):
        assert np.array_equal(out[name]["rows"], ref["rows"]), name
        assert np.abs(out[name]["it"].astype(int) - ref["it"].astype(int)).max() <= 1, (name, out[name]["it"], ref["it"])
        assert np.abs(out[name]["u"] - ref["u"]).max() <= 1e-5, (name, np.abs(out[name]["u"] - ref["u"]).max())


@pytest.mark.gpu
def test_batched_polled_loop_after_a_hard_solve_followed_by_easy_ones(hip, tmp_path):
    """The polled batched loop queues a blind first burst of (previous iteration count - 2) iterations.  When the next
    solves need far fewer (here: the same steps again at a much looser tolerance), the rest of the burst is still queued
    when the host moves on to the next step; whatever those launches publish must not be taken for progress of the new
    solve (ScalMirror epochs; columns that have converged publish nothing).  On a mesh big enough that the device lags the
    host, polled and copied-back loops must agree bit for bit in iteration counts and fields."""
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(os.path.join(ROOT, 'tests'))!r})\n"
        "from conftest import build_case\n"
        "from helpers import make_problem\n"
        "from heatflow_amd import hip_backend as hb\n"
        "cfg, stack, mesh = build_case('geballe_with_diamond', 1.0)\n"
        "prob = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)\n"
        "be = prob.backend\n"
        "nv = 8\n"
        "be.batch_begin(nv, hb.BATCH_SHARED)\n"
        "g = np.stack([np.repeat(prob.bc_values((s + 1) * prob.dt, [prob.bcs[3]])[:, None], nv, axis=1) + 3.0 * np.arange(nv) for s in range(8, 40, 4)])\n"
        "for j in range(nv): be.batch_set_state(j, np.full(prob.n, 300.0 + j))\n"
        "its = []\n"
        "for rtol in (1e-12, 1e-3, 1e-12, 1e-2, 1e-3):\n"
        "    _, bit = be.batch_run(g, rtol, 0.0, prob.max_it, None)\n"
        "    its.append(np.array(bit))\n"
        "ub = np.stack([be.batch_get_state(j) for j in range(nv)])\n"
        "be.batch_end()\n"
        "np.savez(sys.argv[1], ub=ub, bit=np.concatenate(its))\n"
        "prob.close()\n")
    out = {}
    for flag in ("1", "0"):
        env = dict(os.environ, HEATFLOW_POLL=flag)
        res = subprocess.run([sys.executable, str(script), str(tmp_path / f"o{flag}.npz")], env=env, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        out[flag] = np.load(tmp_path / f"o{flag}.npz")
    bit = out["1"]["bit"]
    assert np.array_equal(bit, out["0"]["bit"])
    assert np.array_equal(out["1"]["ub"], out["0"]["ub"])
    per_call = bit.reshape(5, -1, 8).max(axis=(1, 2))
    assert per_call[0] >= per_call[1] + 3 and per_call[2] >= per_call[3] + 3, per_call     # the drops that leave launches queued


def test_two_heated_lines_take_two_response_directions(hip, case_no_diamond_small):
    """Two independently driven Dirichlet lines (the two-sided extension): the second difference of the boundary
    vector spans two directions, the library learns exactly two responses and still reproduces the kind-0 answer."""
    from conftest import HEATING_CSV
    from heatflow_amd.bc import P1Space, RowDirichletBC
    from heatflow_amd.heating import HeatingCurve
    from heatflow_amd.solver import HeatProblem

    cfg, stack, mesh = case_no_diamond_small
    bcs, ic, _ = reference_bcs(cfg, stack, mesh)
    heat_o = HeatingCurve(HEATING_CSV, ic, float(cfg["heating"]["fwhm"]), column="oside")
    bcs.append(RowDirichletBC(P1Space(mesh.coords), "x", coord=stack.heated_z_oside, length=abs(stack.r_sample) * 2,
                              center=0.0, value=heat_o.gaussian))
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    out = {}
    for kind in (0, 2):
        prob = HeatProblem(mesh.coords, mesh.tris, mesh.tags, tag_to_k, tag_to_rc, dt, bcs, ic, precond=1)
        try:
            prob.backend.set_start_vector(kind)
            _, _, iters = prob.run(30, time_varying=bcs[3:])
            out[kind] = (prob.state(), int(np.sum(iters)), prob.backend.response_solves())
        finally:
            prob.close()
    assert out[0][2] == 0 and out[2][2] == 2
    assert np.abs(out[2][0] - out[0][0]).max() <= 2e-5
    assert out[2][1] < out[0][1]


def test_golden_fixture_fields_are_reproduced_by_the_hip_path(hip):
    """tests/golden/with_diamond_tiny.npz (mesh arrays, Dirichlet DOFs, 12 full fields; written by
    tests/golden/make_golden.py from the oracle, which tests/test_oracle.py re-checks): the HIP path runs on
    exactly those arrays and must give those fields to <= 1e-4 K at every step."""
    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes
    from heatflow_amd.bc import P1Space, RowDirichletBC
    from heatflow_amd.heating import HeatingCurve
    from heatflow_amd.solver import HeatProblem

    g = np.load(os.path.join(ROOT, "tests", "golden", "with_diamond_tiny.npz"))
    cfg = scale_mesh_sizes(load_cfg("geballe_with_diamond"), float(g["mesh_scale"]))
    stack = build_stack(cfg)
    mtags = {str(k): int(v) for k, v in zip(g["material_names"], g["material_tag_values"])}
    ic = float(cfg["heating"]["ic_temp"])
    heat = HeatingCurve(HEATING_CSV, ic, float(cfg["heating"]["fwhm"]))
    V = P1Space(g["coords"])
    bcs = [RowDirichletBC(V, "left", value=ic), RowDirichletBC(V, "right", value=ic), RowDirichletBC(V, "top", value=ic),
           RowDirichletBC(V, "x", coord=stack.heated_z, length=abs(stack.r_sample) * 2, center=0.0, value=heat.gaussian)]
    tk = {mtags[m.name]: m.properties["k"] for m in stack.materials}
    trc = {mtags[m.name]: m.properties["rho_cv"] for m in stack.materials}
    dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    for precond in (0, 1):
        prob = HeatProblem(g["coords"], g["tris"], g["tags"], tk, trc, dt, bcs, ic, precond=precond)
        try:
            assert np.array_equal(prob.bc_dofs, g["bc_dofs"])                # the fixture's Dirichlet set
            for b in prob.bcs:
                b.update(0.0)
            worst = 0.0
            for k in range(g["fields"].shape[0]):
                t = (k + 1) * prob.dt
                assert t == pytest.approx(float(g["times"][k]), rel=0, abs=1e-20)
                prob.step(t, only=[prob.bcs[3]])
                worst = max(worst, float(np.abs(prob.state() - g["fields"][k]).max()))
            assert worst <= FIELD_TOL_K, f"precond {precond}: worst |dT| = {worst:.3e} K"
            assert g["fields"][-1].max() > 302.0           # the heating has started inside the fixture's 12 steps
        finally:
            prob.close()


def test_mesh_read_through_the_msh41_reader_runs_on_hip_and_matches_the_oracle(hip, tmp_path):
    """A mesh in the reference's file format (MSH 4.1 ASCII, physical group per surface, tags that are not
    list positions) goes read_msh -> reorder -> HIP; five steps past the onset of the heating against the
    oracle on the same arrays."""
    from conftest import HEATING_CSV, build_case
    from heatflow_amd.mesh import load_mesh_arrays, write_msh41
    from oracle import heat_oracle as ho

    cfg, stack, mesh = build_case("geballe_no_diamond", 4.0)
    perm = {int(t): 30 + 2 * int(t) for t in np.unique(mesh.tags)}
    path = str(tmp_path / "mesh.msh")
    write_msh41(path, mesh.coords, mesh.tris, np.array([perm[int(t)] for t in mesh.tags]),
                {nm: perm[t] for nm, t in mesh.material_tags.items()}, surface_ids={v: k for k, v in perm.items()})
    coords, tris, tags = load_mesh_arrays(path)                 # no npz sidecar: the 4.1 reader + Morton reorder
    assert len(coords) == len(mesh.coords) and not np.array_equal(coords, mesh.coords)
    mtags = {nm: perm[t] for nm, t in mesh.material_tags.items()}

    class M:                                                    # what helpers.make_problem reads
        pass
    m = M()
    m.coords, m.tris, m.tags, m.material_tags = coords, tris, tags, mtags
    nsteps = 7                                                  # no-diamond dt = 1.875e-7: heating from step 2
    ref = ho.run_reference_algorithm(cfg, coords, tris, tags, mtags, HEATING_CSV, num_steps=nsteps, keep_fields=True)
    prob = make_problem(cfg, stack, m, precond=1)
    try:
        for b in prob.bcs:
            b.update(0.0)
        for k in range(nsteps):
            prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
            assert np.abs(prob.state() - ref["fields"][k]).max() <= FIELD_TOL_K
        assert ref["fields"][-1].max() > 320.0 and max(prob.iters) >= 3
    finally:
        prob.close()


def test_hierarchy_export_and_install_give_a_bit_identical_time_loop(hip):
    """hf_amg_export / hf_amg_install: the multigrid hierarchy one context built is installed by another on the same mesh
    (host blob, and a blob in device memory as an RCCL broadcast leaves it) instead of repeating the host set-up.
    Same operator -> the same cycle, bit for bit (fields and iteration counts of a time loop, single and batched);
    another point of a kappa sweep -> the installed hierarchy is a frozen one: same answer to solver tolerance.
    Damaged blobs and blobs of another mesh are refused."""
    import copy
    import ctypes as C
    from conftest import build_case

    cfg, stack, mesh = build_case("geballe_with_diamond", 2.0)
    assert len(mesh.coords) > 30000

    def loop(prob, nsteps=10):
        for bc in prob.bcs:
            bc.update(0.0)
        for k in range(nsteps):
            prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
        return prob.state(), list(prob.iters)

    a = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)
    try:
        info_a = a.backend.amg_info()
        assert info_a["levels"] >= 3
        blob = a.backend.amg_export()
        assert blob.dtype == np.uint8 and bytes(blob[:6]) == b"HFAMG0"
        ua, ita = loop(a)
    finally:
        a.close()
    b = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True, amg=blob)
    try:
        info_b = b.backend.amg_info()
        assert info_b["rows"] == info_a["rows"] and info_b["op_complexity"] == info_a["op_complexity"]
        assert np.array_equal(b.backend.amg_export(), blob)          # and exports the same blob again
        ub, itb = loop(b)
        assert itb == ita and np.array_equal(ub, ua) and max(ita) >= 4
    finally:
        b.close()
    # the blob in device memory, read from there by address
    rt = hip.load_library()
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    d_in = C.c_void_p()
    assert rt.hipMalloc(C.byref(d_in), blob.nbytes) == 0
    try:
        assert rt.hipMemcpy(d_in, C.c_void_p(blob.ctypes.data), blob.nbytes, 1) == 0
        c = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True, amg=(d_in.value, blob.nbytes))
        try:
            uc, itc = loop(c)
            assert itc == ita and np.array_equal(uc, ua)
        finally:
            c.close()
    finally:
        rt.hipFree(d_in)
    # another point of a sweep: kappa_sample 3.8 -> 4.3 under the hierarchy built for 3.8 = what the context does itself when
    # it keeps its levels (frozen hierarchy): bit-identical to that, and equal to a hierarchy of its own within the tolerance
    cfg2 = copy.deepcopy(cfg)
    cfg2["mats"]["p_sample"]["k"] = 4.3
    from heatflow_amd.geometry import build_stack
    stack2 = build_stack(cfg2)
    own = make_problem(cfg2, stack2, mesh, precond=1, amg_reuse=True)
    try:
        u_own, it_own = loop(own)
    finally:
        own.close()
    kept = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)
    try:
        tk, trc = material_tables(stack2, mesh)
        kept.set_materials(tk, trc)
        u_kept, it_kept = loop(kept)
    finally:
        kept.close()
    inst = make_problem(cfg2, stack2, mesh, precond=1, amg_reuse=True, amg=blob)
    try:
        u_inst, it_inst = loop(inst)
    finally:
        inst.close()
    assert it_inst == it_kept and np.array_equal(u_inst, u_kept)
    assert np.abs(u_inst - u_own).max() <= 1e-5 and max(it_inst) <= max(it_own) + 4
    # refused
    small = build_case("geballe_with_diamond", 8.0)
    with pytest.raises(ValueError, match="this mesh has"):
        make_problem(small[0], small[1], small[2], precond=1, amg_reuse=True, amg=blob).close()
    bad = blob.copy(); bad[0] ^= 1
    with pytest.raises(ValueError, match="not a hierarchy blob"):
        make_problem(cfg, stack, mesh, precond=1, amg_reuse=True, amg=bad).close()
    with pytest.raises(ValueError, match="bytes"):
        make_problem(cfg, stack, mesh, precond=1, amg_reuse=True, amg=blob[:-16]).close()
    hdr = 8 + 16 + 8 * 4 + 16 + 8                        # AmgBlobHeader; the coefficient tables and level 0 follow
    bad = blob.copy()
    words = bad[(hdr + 15) // 16 * 16:].view(np.int32)
    # first operator's column indices lie somewhere behind the tables: corrupt a large run of int32 words to out-of-range values
    words[4096:4096 + 64] = 2 ** 30
    with pytest.raises(ValueError, match="hf_amg_install"):
        make_problem(cfg, stack, mesh, precond=1, amg_reuse=True, amg=bad).close()
    with hip.HeatflowHIP(0) as e:
        e.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        with pytest.raises(hip.HipError, match="reuse = 1"):
            e.amg_install(blob)


def test_device_built_hierarchy_is_bit_identical_to_the_host_built_one(hip, tmp_path):
    """The multigrid set-up forms its sparse products, transposes, fused legs and compressed column streams on the device
    (hf_amg_gpu.hpp) in the order the host routines of amg_host.hpp use; HEATFLOW_AMG_SETUP=host keeps the host-only
    set-up.  The exported hierarchies (every operator's pointers, indices, values, kernel geometry, column streams, dense
    inverse) are equal byte for byte, on the stock mesh (two device levels) and on a small one, and so is a time loop."""
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(os.path.join(ROOT, 'tests'))!r})\n"
        "from conftest import build_case\n"
        "from helpers import make_problem\n"
        "out = {}\n"
        "for scale in (1.0, 6.0):\n"
        "    cfg, stack, mesh = build_case('geballe_with_diamond', scale)\n"
        "    prob = make_problem(cfg, stack, mesh, precond=1, amg_reuse=True)\n"
        "    out[f'blob{scale}'] = prob.backend.amg_export()\n"
        "    out[f'rows{scale}'] = np.array(prob.backend.amg_info()['rows'])\n"
        "    _, _, it = prob.run(10, time_varying=[prob.bcs[3]])\n"
        "    out[f'u{scale}'] = prob.state(); out[f'it{scale}'] = np.array(it)\n"
        "    prob.close()\n"
        "np.savez(sys.argv[1], **out)\n")
    got = {}
    for mode in ("device", "host"):
        env = dict(os.environ)
        env.pop("HEATFLOW_AMG_SETUP", None)
        if mode == "host":
            env["HEATFLOW_AMG_SETUP"] = "host"
        res = subprocess.run([sys.executable, str(script), str(tmp_path / f"{mode}.npz")], env=env, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        got[mode] = np.load(tmp_path / f"{mode}.npz")
    for scale in ("1.0", "6.0"):
        assert np.array_equal(got["device"][f"rows{scale}"], got["host"][f"rows{scale}"])
        a, b = got["device"][f"blob{scale}"], got["host"][f"blob{scale}"]
        assert a.shape == b.shape
        if not np.array_equal(a, b):
            first = int(np.flatnonzero(a != b)[0])
            raise AssertionError(f"scale {scale}: hierarchy blobs differ in {int((a != b).sum())} of {a.size} bytes, first at offset {first}")
        assert np.array_equal(got["device"][f"it{scale}"], got["host"][f"it{scale}"])
        assert np.array_equal(got["device"][f"u{scale}"], got["host"][f"u{scale}"])
    assert len(got["device"]["rows1.0"]) >= 4 and got["device"]["rows1.0"][1] > 20000


def test_sweep_blobs_in_torch_device_memory_are_installed_by_address(hip, tmp_path):
    """What a non-zero rank of an RCCL sweep does after the broadcasts: connectivity tables and multigrid hierarchy sit in torch
    tensors on the device (parameter_sweep.DeviceBlob) and the solver session installs both by address - no host copy on the
    Python side.  torch is imported before the library is loaded (one HIP runtime for both), hence the subprocess.  The run
    equals the one of a session that built everything itself, bit for bit."""
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        "import torch\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(os.path.join(ROOT, 'tests'))!r})\n"
        "from conftest import build_case, HEATING_CSV\n"
        "from heatflow_amd.driver import SimulationSession\n"
        "from heatflow_amd.geometry import watcher_points\n"
        "from heatflow_amd.parameter_sweep import DeviceBlob\n"
        "cfg, stack, mesh = build_case('geballe_with_diamond', 4.0)\n"
        "cfg['heating']['file'] = HEATING_CSV\n"
        "cfg['timing']['num_steps'] = 12; cfg['timing']['t_final'] = 12 * 7.5e-8\n"
        "a = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags)\n"
        "ra = a.run(cfg, stack, watcher_points(cfg))\n"
        "pat = a.problem.backend.export_pattern(); share = a.export_hierarchy()\n"
        "dev = torch.device('cuda', 0)\n"
        "tp = torch.from_numpy(pat).to(dev); th = torch.from_numpy(share['blob']).to(dev); torch.cuda.synchronize()\n"
        "b = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, pattern=DeviceBlob(tp),\n"
        "                      hierarchy={'blob': DeviceBlob(th), 'k': share['k']})\n"
        "rb = b.run(cfg, stack, watcher_points(cfg))\n"
        "info = (a.problem.backend.amg_info(), b.problem.backend.amg_info())\n"
        "ua, ub = a.problem.state(), b.problem.state()\n"
        "a.close(); b.close()\n"
        "assert info[0]['rows'] == info[1]['rows'], info\n"
        "assert np.array_equal(ra['iters'], rb['iters']) and np.array_equal(ua, ub), (ra['iters'], rb['iters'])\n"
        "assert np.array_equal(ra['watchers']['oside'], rb['watchers']['oside']) and ra['iters'].max() >= 4\n"
        "print('ok', len(DeviceBlob(tp)), len(DeviceBlob(th)))\n")
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0 and res.stdout.strip().splitlines()[-1].startswith("ok"), res.stderr[-3000:]


def test_pattern_blob_export_and_prebuilt_install(hip, case_with_diamond_small):
    """hf_pattern_export / hf_set_mesh_prebuilt: the connectivity tables one context built are installed by another
    (host blob, and a blob held in device memory as an RCCL broadcast leaves it); matrices, lazily built scatter
    lists and a short time loop are bit-identical to the context that built them.  Damaged blobs are refused."""
    cfg, stack, mesh = case_with_diamond_small
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    tags = sorted(tag_to_k)

    def matrices(be, mode):
        be.set_materials(tags, [tag_to_k[t] for t in tags], [tag_to_rc[t] for t in tags])
        be.assemble(1e-7, mode)
        return be.get_csr()

    with hip.HeatflowHIP(0) as a:
        a.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        blob = a.export_pattern()
        assert blob.dtype == np.uint8 and blob.nbytes == a.pattern_bytes() and bytes(blob[:6]) == b"HFPAT0"
        ref = {m: matrices(a, m) for m in (3, 1, 0)}
    with hip.HeatflowHIP(0) as b:
        b.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=blob)
        assert (b.n, b.n_e, b.nnz) == (len(mesh.coords), len(mesh.tris), len(ref[3][1]))
        for m in (3, 1):                                  # mode 1 builds the scatter lists on first use
            got = matrices(b, m)
            for x, y in zip(got, ref[m]):
                assert np.array_equal(x, y)
        assert np.array_equal(b.export_pattern(), blob)   # and exports the same blob again
    # a blob in device memory (what an RCCL broadcast leaves behind): plain hipMalloc through the HIP runtime the
    # library itself is linked against (dlsym on its handle searches its dependencies; another copy of the runtime,
    # e.g. torch's bundled one, may be loaded in this process too)
    import ctypes as C
    rt = hip.load_library()
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    d_in, d_out = C.c_void_p(), C.c_void_p()
    assert rt.hipMalloc(C.byref(d_in), blob.nbytes) == 0 and rt.hipMalloc(C.byref(d_out), blob.nbytes) == 0
    try:
        assert rt.hipMemcpy(d_in, C.c_void_p(blob.ctypes.data), blob.nbytes, 1) == 0          # host -> device
        with hip.HeatflowHIP(0) as c:
            c.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=(d_in.value, blob.nbytes))
            got = matrices(c, 3)
            assert all(np.array_equal(x, y) for x, y in zip(got, ref[3]))
            c.export_pattern(into=(d_out.value, blob.nbytes))
        back = np.empty_like(blob)
        assert rt.hipMemcpy(C.c_void_p(back.ctypes.data), d_out, blob.nbytes, 2) == 0          # device -> host
        assert np.array_equal(back, blob)
    finally:
        rt.hipFree(d_in)
        rt.hipFree(d_out)
    # whole problems: built vs installed tables give the same time loop, bit for bit
    states = []
    for pat in (None, blob):
        prob = make_problem(cfg, stack, mesh, precond=1, pattern=pat)
        try:
            for bc in prob.bcs:
                bc.update(0.0)
            for k in range(8):
                prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
            states.append((prob.state(), list(prob.iters)))
        finally:
            prob.close()
    assert states[0][1] == states[1][1] and np.array_equal(states[0][0], states[1][0])
    # refused: wrong magic, wrong size, other mesh, an index out of range
    with hip.HeatflowHIP(0) as d:
        bad = blob.copy(); bad[0] ^= 1
        with pytest.raises(ValueError, match="not a pattern blob"):
            d.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=bad)
        with pytest.raises(ValueError, match="bytes"):
            d.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=blob[:-16])
        with pytest.raises(ValueError, match="exported for a mesh"):
            d.set_mesh(mesh.coords[:-1], mesh.tris[mesh.tris.max(axis=1) < len(mesh.coords) - 1], mesh.tags[mesh.tris.max(axis=1) < len(mesh.coords) - 1], pattern=blob)
        bad = blob.copy()
        hdr = 8 + 16 + 10 * 4 + 12 * 8
        off = (hdr + 15) // 16 * 16 + (4 * (len(mesh.coords) + 1) + 15) // 16 * 16     # first column index
        bad[off:off + 4] = np.frombuffer(np.int32(len(mesh.coords) + 5).tobytes(), dtype=np.uint8)
        with pytest.raises(ValueError, match="outside its range"):
            d.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=bad)
        d.set_mesh(mesh.coords, mesh.tris, mesh.tags, pattern=blob)        # the context is still usable
        assert d.n == len(mesh.coords)


@pytest.mark.parametrize("kind", ["per_column", "affine"])
@pytest.mark.parametrize("precond", [0, 1])
@pytest.mark.parametrize("nv", [2, 4, 8, 16])
def test_batched_time_loop_with_per_column_operators_matches_single_runs_and_oracle(hip, nv, precond, kind, case_with_diamond_small):
    """hf_batch_*: nv kappa_sample values advance together as interleaved columns (per-column fine operator,
    shared frozen hierarchy).  Every column must match the oracle's run for its kappa at every step (<= 1e-4 K)
    and the single-column run of the same context to solver tolerance."""
    import copy
    from conftest import HEATING_CSV
    from oracle import heat_oracle as ho

    cfg, stack, mesh = case_with_diamond_small
    nsteps = 12
    ks = [3.3 + 0.14 * j for j in range(nv)]
    tag_s = mesh.material_tags["p_sample"]
    prob = make_problem(cfg, stack, mesh, precond=precond, amg_reuse=True)
    be = prob.backend
    try:
        for bc in prob.bcs:
            bc.update(0.0)
        g_one = np.array([prob.bc_values((k + 1) * prob.dt, [prob.bcs[3]]) for k in range(nsteps)])      # same boundary values for all
        tk, trc = material_tables(stack, mesh)
        singles = []
        for kap in ks:                                   # reference: one column at a time through hf_run
            be.update_kappa([tag_s], [kap])
            prob.set_state(300.0)
            _, it1 = be.run(g_one, prob.rtol, 0.0, prob.max_it, None)
            singles.append((prob.state(), it1))
        if kind == "affine":                             # A_j = A(kappa_ref) + (kappa_j - kappa_ref) dt K_sample
            ref_k = ks[nv // 2]
            be.update_kappa([tag_s], [ref_k])
            be.batch_begin(nv, per_column_operator=hip.BATCH_AFFINE)
            be.batch_set_affine([tag_s], [kap - ref_k for kap in ks])
        else:
            be.batch_begin(nv, per_column_operator=hip.BATCH_PER_COLUMN)
            for j, kap in enumerate(ks):
                be.update_kappa([tag_s], [kap])
                be.batch_load_column(j)
        for j in range(nv):
            be.batch_set_state(j, np.full(prob.n, 300.0))
        g_all = np.repeat(g_one[:, :, None], nv, axis=2)
        nodes = np.array([0, prob.n // 2, prob.n - 1], dtype=np.int32)
        fields = []
        for k in range(nsteps):                          # step by step, so that every field can be checked
            samples, iters = be.batch_run(g_all[k:k + 1], prob.rtol, 0.0, prob.max_it, nodes)
            fields.append([be.batch_get_state(j) for j in range(nv)])
            for j in range(nv):
                assert np.array_equal(samples[0, j], fields[-1][j][nodes])
        be.batch_end()
        for j, kap in enumerate(ks):
            c = copy.deepcopy(cfg)
            c["mats"]["p_sample"]["k"] = kap
            ref = ho.run_reference_algorithm(c, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV,
                                             num_steps=nsteps, keep_fields=True)
            worst = max(np.abs(fields[k][j] - ref["fields"][k]).max() for k in range(nsteps))
            assert worst <= FIELD_TOL_K, f"column {j} (kappa {kap}): worst |dT| = {worst:.3e} K"
            assert np.abs(fields[-1][j] - singles[j][0]).max() <= 1e-5
        assert not np.array_equal(fields[-1][0], fields[-1][nv - 1])          # the columns really differ
    finally:
        prob.close()


@pytest.mark.parametrize("precond", [0, 1])
def test_batched_time_loop_with_a_shared_operator_matches_single_runs(hip, precond, case_no_diamond_small):
    """Columns that differ only in their boundary values (four fwhm values of the heated line's Gaussian, as the
    fwhm axis of parameter_sweep.py:195-235): one shared operator, all steps in one hf_batch_run."""
    from conftest import HEATING_CSV
    from heatflow_amd.heating import HeatingCurve

    cfg, stack, mesh = case_no_diamond_small
    nsteps, nv = 14, 4
    fwhms = [8e-6, 1.32e-5, 2e-5, 4e-5]
    prob = make_problem(cfg, stack, mesh, precond=precond)
    be = prob.backend
    try:
        ic = float(cfg["heating"]["ic_temp"])
        g_cols, singles = [], []
        nodes = np.array([1, prob.n // 3, prob.n - 2], dtype=np.int32)
        for f in fwhms:
            heat = HeatingCurve(HEATING_CSV, ic, f)
            prob.bcs[3]._value = heat.gaussian
            for bc in prob.bcs:
                bc.update(0.0)
            g = np.array([prob.bc_values((k + 1) * prob.dt, [prob.bcs[3]]) for k in range(nsteps)])
            g_cols.append(g)
            prob.set_state(ic)
            samp, it1 = be.run(g, prob.rtol, 0.0, prob.max_it, nodes)
            singles.append((prob.state(), samp, it1))
        be.batch_begin(nv, per_column_operator=False)
        for j in range(nv):
            be.batch_set_state(j, np.full(prob.n, ic))
        samples, iters = be.batch_run(np.stack(g_cols, axis=2), prob.rtol, 0.0, prob.max_it, nodes)
        for j in range(nv):
            assert np.abs(be.batch_get_state(j) - singles[j][0]).max() <= 1e-5
            assert np.abs(samples[:, j, :] - singles[j][1]).max() <= 1e-5
            assert iters[:, j].max() >= 3 and int(iters[:, j].sum()) <= 1.5 * int(singles[j][2].sum()) + nsteps   # the batch starts from 2u^n - u^(n-1), the single run from the projected start vector
        assert np.abs(singles[0][0] - singles[3][0]).max() > 1.0
        be.batch_end()
        # the context still steps on its own after the batch
        prob.set_state(ic)
        be.run(g_cols[1][:3], prob.rtol, 0.0, prob.max_it, None)
    finally:
        prob.close()


@pytest.mark.parametrize("precond", [0, 1])
def test_batched_loop_reports_non_convergence_and_the_context_recovers(hip, precond, case_with_diamond_small):
    """hf_batch_run with too few iterations allowed: NotConverged (polled multigrid loop and burst loop alike); after
    hf_batch_end the same context runs a normal single-column loop."""
    cfg, stack, mesh = case_with_diamond_small
    prob = make_problem(cfg, stack, mesh, precond=precond, amg_reuse=True)
    be = prob.backend
    try:
        for bc in prob.bcs:
            bc.update(0.0)
        g = np.array([prob.bc_values((k + 1) * prob.dt, [prob.bcs[3]]) for k in range(8)])
        be.batch_begin(4, per_column_operator=hip.BATCH_SHARED)
        for j in range(4):
            be.batch_set_state(j, np.full(prob.n, 300.0))
        with pytest.raises(hip.NotConverged, match="not converged|breakdown"):
            be.batch_run(np.repeat(g[:, :, None], 4, axis=2), prob.rtol, 0.0, 2, None)
        be.batch_end()
        prob.set_state(300.0)
        _, _, iters = prob.run(8, time_varying=[prob.bcs[3]])
        assert np.asarray(iters).max() >= 3 and np.isfinite(prob.state()).all()
    finally:
        prob.close()


def test_batch_call_order_and_argument_errors(hip, case_with_diamond_small):
    """hf_batch_*: wrong order / bad arguments are refused with a message, the context stays usable, and a batch is
    closed by anything that invalidates what it was sized for."""
    cfg, stack, mesh = case_with_diamond_small
    tk, trc = material_tables(stack, mesh)
    tags = sorted(tk)
    with hip.HeatflowHIP(0) as be:
        be.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        be.set_materials(tags, [tk[t] for t in tags], [trc[t] for t in tags])
        with pytest.raises(hip.HipError, match="before hf_assemble"):
            be.batch_begin(4)
    prob = make_problem(cfg, stack, mesh, precond=1)              # hierarchy not frozen (amg_reuse = False)
    be = prob.backend
    try:
        with pytest.raises(ValueError, match="2, 4, 8 or 16"):
            be.batch_begin(3)
        with pytest.raises(ValueError, match="unknown operator kind"):
            be.batch_begin(4, per_column_operator=5)
        with pytest.raises(hip.HipError, match="frozen hierarchy"):
            be.batch_begin(4, per_column_operator=hip.BATCH_PER_COLUMN)
        with pytest.raises(hip.HipError, match="no batch is open"):
            be.batch_run(np.zeros((1, be.n_bc, 4)))
        be.batch_begin(4)                                           # shared operator: fine without a frozen hierarchy
        with pytest.raises(hip.HipError, match="per-column"):
            be.batch_load_column(0)
        with pytest.raises(hip.HipError, match="affine"):
            be.batch_set_affine([mesh.material_tags["p_sample"]], [0.0, 0.1, 0.2, 0.3])
        with pytest.raises(ValueError, match="column"):
            be.batch_set_state(4, np.zeros(prob.n))
        with pytest.raises(ValueError):
            be.batch_run(np.zeros((2, be.n_bc + 1, 4)))             # wrong boundary-value shape (checked by the binding)
        be.set_precond(1, True)                                     # same kind: the batch survives ...
        be.batch_set_state(0, np.full(prob.n, 300.0))
        be.set_dirichlet(prob.bc_dofs)                              # ... a new Dirichlet set closes it
        with pytest.raises(hip.HipError, match="no batch is open"):
            be.batch_set_state(0, np.full(prob.n, 300.0))
        be.assemble(prob.dt, prob.assembly_mode)
        be.batch_begin(2, per_column_operator=hip.BATCH_AFFINE)
        with pytest.raises(hip.HipError, match="hf_batch_set_affine has not been called"):
            be.batch_run(np.zeros((1, be.n_bc, 2)))
        with pytest.raises(ValueError, match="not a cell tag"):
            be.batch_set_affine([999], [0.0, 0.1])
        be.batch_end()
        prob.set_state(300.0)                                       # and the context steps on its own again
        for bc in prob.bcs:
            bc.update(0.0)
        prob.step(prob.dt)
    finally:
        prob.close()


@pytest.mark.parametrize("precond", [0, 1])
def test_time_loop_is_bitwise_reproducible(hip, precond, case_with_diamond_small):
    """Fixed-order reductions everywhere (partial sums, projection start vector, row-gather assembly): two runs of
    the same problem give the same bits and the same iteration counts."""
    cfg, stack, mesh = case_with_diamond_small
    out = []
    for _ in range(2):
        prob = make_problem(cfg, stack, mesh, precond=precond)
        try:
            _, _, iters = prob.run(25, time_varying=[prob.bcs[3]])
            out.append((prob.state(), iters.copy()))
        finally:
            prob.close()
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])


def test_fused_finest_level_gives_the_same_answer(hip, case_with_diamond_small, monkeypatch):
    """HEATFLOW_AMG_FUSE0: the V-cycle's finest level as two fused operators (1: Rt_0 and GP_0, the library's choice
    above ~3M DOF) or with the down leg fused only (2: its choice below) against the explicit sweeps over A (0): the same
    preconditioner in exact arithmetic, so the same iteration counts to within one and the same field to solver
    tolerance."""
    cfg, stack, mesh = case_with_diamond_small
    monkeypatch.setenv("HEATFLOW_STREAM_MIN_ROWS", "1000")     # let the LDS-staged kernel run this small mesh's operators
    out = {}
    for flag in ("0", "1", "2"):
        monkeypatch.setenv("HEATFLOW_AMG_FUSE0", flag)
        prob = make_problem(cfg, stack, mesh, precond=1)
        try:
            _, _, iters = prob.run(20, time_varying=[prob.bcs[3]])
            out[flag] = (prob.state(), iters.copy())
        finally:
            prob.close()
    for flag in ("1", "2"):
        assert np.abs(out["0"][0] - out[flag][0]).max() <= 2e-5
        assert np.abs(out["0"][1].astype(int) - out[flag][1].astype(int)).max() <= 1 and out[flag][1].max() >= 5
