"""GPU tests at BASELINE.json's full sizes.

C2 (stock geballe_no_diamond, ~1.5e5 DOF): the oracle still finishes in seconds, so the whole
40-step run is compared field by field.  C3 (geballe_with_diamond refined to ~1.04e6 DOF): the
oracle's LU is too slow for a unit test, so the HIP path is checked through size-independent
properties of the discretisation (SURVEY section 8c pins 1-3, 6) and through agreement of its
two independent solvers (Jacobi-PCG and multigrid-PCG).
"""
import os

import numpy as np
import pytest

from conftest import build_case
from helpers import make_problem, material_tables, oracle_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    return build_case("geballe_no_diamond", 1.0)


@pytest.fixture(scope="module")
def c3():
    return build_case("geballe_with_diamond", 0.43)


def test_c2_stock_no_diamond_all_40_steps_match_oracle(hip, c2):
    cfg, stack, mesh = c2
    assert int(cfg["timing"]["num_steps"]) == 40 and 1.0e5 < len(mesh.coords) < 2.5e5
    ref = oracle_run(cfg, mesh, 40)
    for precond in (1, 0):
        prob = make_problem(cfg, stack, mesh, precond=precond)
        try:
            for bc in prob.bcs:
                bc.update(0.0)
            worst = 0.0
            for k in range(40):
                prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
                if precond == 1 or k % 8 == 7:           # every step for the default solver, samples for Jacobi
                    worst = max(worst, float(np.abs(prob.state() - ref["fields"][k]).max()))
            assert worst <= 1e-4, f"precond={precond}: {worst:.3e} K"
            print(f"C2 precond={precond}: worst |dT| = {worst:.2e} K, mean iterations/step = {np.mean(prob.iters):.1f}")
        finally:
            prob.close()
    assert ref["fields"][-1].max() > 400.0


def test_c3_one_million_dof_discretisation_properties(hip, c3):
    import scipy.sparse as sp

    cfg, stack, mesh = c3
    n = len(mesh.coords)
    assert abs(n - 1.0e6) <= 0.05e6
    tag_to_k, tag_to_rc = material_tables(stack, mesh)
    tags = sorted(tag_to_k)
    dt = float(cfg["timing"]["t_final"]) / int(cfg["timing"]["num_steps"])
    with hip.HeatflowHIP(0) as be:
        be.set_mesh(mesh.coords, mesh.tris, mesh.tags)
        be.set_materials(tags, [tag_to_k[t] for t in tags], [tag_to_rc[t] for t in tags])
        be.assemble(dt, hip.ASM_LDS_ATOMIC)
        rowptr, colidx, A, M = be.get_csr()
        # total mass = sum over the material boxes of rho_c * dz * (r2^2 - r1^2)/2
        expect = sum(m.properties["rho_cv"] * (m.boundaries[1] - m.boundaries[0]) *
                     (m.boundaries[3] ** 2 - m.boundaries[2] ** 2) / 2 for m in stack.materials)
        assert np.isclose(M.sum(), expect, rtol=1e-10)
        # exact symmetry; K 1 = 0  <=>  A 1 = M 1
        Ad = sp.csr_matrix((A, colidx, rowptr), shape=(n, n))
        Md = sp.csr_matrix((M, colidx, rowptr), shape=(n, n))
        assert abs(Ad - Ad.T).max() == 0.0 and abs(Md - Md.T).max() == 0.0
        ones = np.ones(n)
        a1, m1 = be.spmv(ones, 0), be.spmv(ones, 1)
        kdiag = (A[rowptr[:-1] + 0] * 0)  # placeholder to keep shapes explicit
        scale = np.abs(Ad).sum(axis=1).A1
        assert np.max(np.abs(a1 - m1) / scale) < 1e-12
        # SpMV against scipy and linearity
        rng = np.random.default_rng(3)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        ax = be.spmv(x, 0)
        assert np.max(np.abs(ax - Ad @ x)) <= 1e-13 * np.max(np.abs(ax))
        lin = be.spmv(2.0 * x - 3.0 * y, 0) - (2.0 * ax - 3.0 * be.spmv(y, 0))
        assert np.max(np.abs(lin)) <= 1e-12 * np.max(np.abs(ax))
        # the three assembly variants agree; the coloured one twice bit for bit
        be.assemble(dt, hip.ASM_LDS_COLORED)
        _, _, A1, M1 = be.get_csr()
        be.assemble(dt, hip.ASM_LDS_COLORED)
        _, _, A1b, M1b = be.get_csr()
        assert np.array_equal(A1, A1b) and np.array_equal(M1, M1b)
        be.assemble(dt, hip.ASM_GLOBAL_ATOMIC)
        _, _, A2, M2 = be.get_csr()
        rows = np.repeat(np.arange(n), np.diff(rowptr))
        rmax = np.maximum.reduceat(np.abs(A), rowptr[:-1])[rows]
        assert np.max(np.abs(A1 - A) / rmax) < 1e-14 and np.max(np.abs(A2 - A) / rmax) < 1e-14
        offdiag = colidx != rows
        assert np.array_equal(A1[offdiag], A[offdiag])            # two addends commute: order-independent bits
        # row gather (the default): evaluates every triangle with its vertices in ascending node order and with
        # reciprocal constants, so it agrees with the scatter kernels to rounding, is exactly symmetric and
        # reproduces itself bit for bit
        be.assemble(dt, hip.ASM_ROW_GATHER)
        _, _, A3, M3 = be.get_csr()
        be.assemble(dt, hip.ASM_ROW_GATHER)
        _, _, A3b, M3b = be.get_csr()
        assert np.array_equal(A3, A3b) and np.array_equal(M3, M3b)
        mmax = np.maximum.reduceat(np.abs(M), rowptr[:-1])[rows]
        assert np.max(np.abs(A3 - A) / rmax) < 1e-13 and np.max(np.abs(M3 - M) / mmax) < 1e-13
        A3d = sp.csr_matrix((A3, colidx, rowptr), shape=(n, n))
        M3d = sp.csr_matrix((M3, colidx, rowptr), shape=(n, n))
        assert abs(A3d - A3d.T).max() == 0.0 and abs(M3d - M3d.T).max() == 0.0
        assert np.isclose(M3.sum(), expect, rtol=1e-10)
        del kdiag


def test_c3_one_million_dof_time_loop_properties(hip, c3):
    """Constant preservation before the heating starts, and the two solvers agree afterwards."""
    cfg, stack, mesh = c3
    fields = {}
    for precond in (1, 0):
        prob = make_problem(cfg, stack, mesh, precond=precond)
        try:
            for bc in prob.bcs:
                bc.update(0.0)
            for k in range(4):
                it, _ = prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
                assert it == 0
            assert np.abs(prob.state() - 300.0).max() < 1e-9
            for k in range(4, 9):
                prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
            fields[precond] = prob.state()
            iters = prob.iters[4:]
            # the first heated step may need no iteration at all: its increment is exactly the boundary response
            # the start vector carries (hf_set_start_vector kind 2); the later ones show each solver's regime
            assert (max(iters) < 40) if precond == 1 else (max(iters) > 200 and sorted(iters)[1] > 200)
        finally:
            prob.close()
    assert np.abs(fields[1] - 300.0).max() > 0.5       # the curve first dips below its start value
    assert np.abs(fields[0] - fields[1]).max() <= 2e-5


def test_c3_one_million_dof_first_steps_match_oracle(hip, c3):
    """The headline configuration against the oracle itself: sparse LU of the 1.04M-DOF operator
    (a few seconds on the GPU box's host), then the first 10 steps field by field."""
    cfg, stack, mesh = c3
    nsteps = 10
    ref = oracle_run(cfg, mesh, nsteps)
    prob = make_problem(cfg, stack, mesh, precond=1)
    try:
        for bc in prob.bcs:
            bc.update(0.0)
        worst = 0.0
        for k in range(nsteps):
            prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
            worst = max(worst, float(np.abs(prob.state() - ref["fields"][k]).max()))
        assert worst <= 1e-4, f"{worst:.3e} K"
        assert np.abs(ref["fields"][-1] - 300.0).max() > 0.5
        print(f"C3: worst |dT| over {nsteps} steps = {worst:.2e} K, iterations/step = {prob.iters}")
    finally:
        prob.close()


def test_c3_one_million_dof_whole_run_watchers_and_final_field_match_oracle(hip, c3):
    """The whole headline run (all 100 steps of BASELINE C3 through hf_run, as bench.py drives it) against the
    oracle: the two watcher curves at every step and the final field.  Error does not accumulate: every step
    is solved to the same tolerance and the loop is a contraction."""
    from heatflow_amd.geometry import watcher_points
    from heatflow_amd.solver import nearest_nodes

    cfg, stack, mesh = c3
    nsteps = int(cfg["timing"]["num_steps"])
    assert nsteps == 100
    wp = watcher_points(cfg)
    nodes = nearest_nodes(mesh.coords, [wp["pside"], wp["oside"]])
    ref = oracle_run(cfg, mesh, nsteps, keep_fields=False, watcher_nodes=nodes)
    prob = make_problem(cfg, stack, mesh, precond=1)
    try:
        _, samples, iters = prob.run(nsteps, watcher_nodes=nodes, time_varying=[prob.bcs[3]])
        assert np.abs(samples - ref["watchers"]).max() <= 1e-4
        assert np.abs(prob.state() - ref["solver"].u).max() <= 1e-4
        assert ref["watchers"][:, 0].max() - 300.0 > 100.0          # the p-side coupler really heats up
        assert prob.backend.response_solves() == 1 and max(iters) < 40
    finally:
        prob.close()


def test_c4_stock_read_flux_matches_oracle(hip, c2):
    """BASELINE config 4 (geballe_no_diamond_read_flux.yaml: the C2 mesh, 50 steps, gradient
    projection every step): projected dT/dr on the axis against the oracle's projection."""
    from conftest import load_cfg
    from oracle import heat_oracle as ho

    _, stack, mesh = c2
    cfg = load_cfg("geballe_no_diamond_read_flux")
    assert int(cfg["timing"]["num_steps"]) == 50
    nsteps = 12
    ref = oracle_run(cfg, mesh, nsteps)
    proj = ho.GradientProjector(mesh.coords, mesh.tris)
    axis = np.nonzero(np.abs(mesh.coords[:, 1]) <= 1e-12)[0]
    prob = make_problem(cfg, stack, mesh, precond=1)
    try:
        prob.backend.flux_setup()
        for bc in prob.bcs:
            bc.update(0.0)
        for k in range(nsteps):
            prob.step((k + 1) * prob.dt, only=[prob.bcs[3]])
            _, gr = prob.backend.flux_project(rtol=1e-11, want_z=False)
            if k % 4 == 3:
                g_ref = proj.project(ref["fields"][k])[:, 1]
                scale = max(np.abs(g_ref).max(), 1.0)
                assert np.abs(gr - g_ref).max() <= 1e-4 * scale + 1e-3
                assert np.abs(gr[axis] - g_ref[axis]).max() <= 1e-4 * scale + 1e-3
        assert scale > 1e6 and prob.backend.last_flux_iters.max() < 100
    finally:
        prob.close()


def test_c4_two_sided_heating_extension_at_stock_size(hip, c2):
    """BASELINE config 4's "two-sided heating" on the stock no-diamond mesh.  EXTENSION WITHOUT A REFERENCE IMPLEMENTATION
    (cfgs/konopkova.yaml is an unparseable stub, no two-sided code exists in the reference): a second Gaussian Dirichlet
    line on the outer face of the o-side coupler driven by the CSV's `oside` column; checked against the oracle's
    restatement of the same extension, 12 steps, every field."""
    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd.driver import SimulationSession
    from oracle import heat_oracle as ho

    _, stack, mesh = c2
    cfg = load_cfg("geballe_no_diamond_read_flux")
    cfg["heating"]["file"] = HEATING_CSV
    nsteps = 12
    cfg["timing"]["num_steps"] = nsteps
    cfg["timing"]["t_final"] = nsteps * 1.5e-7
    ref = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV, keep_fields=True,
                                     second_line=stack.heated_z_oside)
    one_sided = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV, keep_fields=True)
    fields = []
    sess = SimulationSession(mesh.coords, mesh.tris, mesh.tags, mesh.material_tags)
    try:
        res = sess.run(cfg, stack, None, field_sink=lambda t, u: fields.append(u.copy()), read_flux=True, two_sided=True)
    finally:
        sess.close()
    assert len(fields) == nsteps and len(sess.coords) > 1.0e5
    worst = max(float(np.abs(f - r).max()) for f, r in zip(fields, ref["fields"]))
    assert worst <= 1e-4, worst
    assert np.abs(ref["fields"][-1] - one_sided["fields"][-1]).max() > 0.5        # the second line matters
    assert res["flux"] is not None and len(res["flux"].rows) == nsteps
    print(f"two-sided extension at stock size: worst |dT| = {worst:.2e} K over {nsteps} steps, {len(sess.coords)} DOF")


def test_c5_sixty_four_point_kappa_sweep_on_one_gpu(hip, tmp_path):
    """BASELINE C5 at its stated size on one GPU, as bench.py runs it (a world of 1 takes all 64 points: batches of 16
    through the batched time loop, 2 loops in flight): 64 kappa_sample values on the stock geballe_with_diamond mesh,
    100 steps each (reference parameter_sweep.py:423-446, sweep_test.py:47-115).  Every point must succeed; three of
    them (both ends and the middle of the grid) are re-run by the oracle on the same mesh and compared at every step."""
    import copy
    import yaml
    from conftest import HEATING_CSV, load_cfg
    from heatflow_amd import parameter_sweep as ps
    from heatflow_amd.geometry import watcher_points
    from heatflow_amd.mesh import load_mesh_arrays
    from heatflow_amd.solver import nearest_nodes
    from oracle import heat_oracle as ho

    cfg = load_cfg("geballe_with_diamond")
    cfg["heating"]["file"] = HEATING_CSV
    ks = ps.get_k_values(count=64)
    assert len(ks) == 64 and len({f"{k:.2f}" for k in ks}) == 64
    mesh_folder, out = str(tmp_path / "mesh"), str(tmp_path / "out")
    timing = {}
    rows = ps.run_kappa_sweep(cfg, mesh_folder, ks, out, rebuild_mesh=True, concurrent=2, batch=16, exp_csv=HEATING_CSV, timing=timing)
    assert len(rows) == 64 and all(r["status"] == "success" for r in rows), [r["error"] for r in rows if r["error"]][:1]
    assert [r["k"] for r in rows] == sorted(ks.tolist()) and timing["sessions"] == 2 and timing["batches"] == [16] * 4
    assert all(r.get("batch") == 16 and not r.get("batch_error") for r in rows)
    assert all(np.isfinite(r["rmse"]) and 0.0 < r["rmse"] < 0.2 for r in rows)
    coords, tris, tags = load_mesh_arrays(os.path.join(mesh_folder, "mesh.msh"))
    mtags = yaml.safe_load(open(os.path.join(mesh_folder, "mesh_cfg.yaml")))["material_tags"]
    assert 150_000 < len(coords) < 450_000                       # stock size
    nodes = nearest_nodes(coords, list(watcher_points(cfg).values()))
    finals = []
    for k in (ks[0], ks[31], ks[63]):
        c = copy.deepcopy(cfg)
        c["mats"]["p_sample"]["k"] = float(k)
        ref = ho.run_reference_algorithm(c, coords, tris, tags, mtags, HEATING_CSV, watcher_nodes=nodes)
        got = np.genfromtxt(os.path.join(out, f"{k:.2f}", "watcher_points.csv"), delimiter=",", names=True)   # 64 distinct 2-digit names
        assert len(got) == 100
        assert np.abs(got["pside"] - ref["watchers"][:, 0]).max() <= 1e-4
        assert np.abs(got["oside"] - ref["watchers"][:, 1]).max() <= 1e-4
        finals.append(got["oside"][-1])
    assert finals[0] < finals[1] < finals[2]                      # a better conductor heats the far side more
    print(f"64 points x {len(coords)} DOF x 100 steps in {timing['points_s']:.2f} s of point loop "
          f"({64 * len(coords) * 100 / timing['points_s']:.3e} DOF-updates/s)")


def test_bench_line_keeps_its_contract_on_a_small_mesh(hip):
    """bench.py end to end on a coarse mesh (seconds): exactly one JSON line with the driver's keys, value consistent with
    ms_per_step, a roofline object for the dominant kernel with live in-loop timing, and a cpu_baseline from the oracle."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "3.0", "--steps", "8", "--warmup", "5", "--cpu-steps", "3",
           "--jacobi-steps", "2", "--sweep-points", "0", "--hbm-scale", "0", "--cpu-farm-points", "0", "--device-warmup-s", "0",
           "--profile-steps", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 8 and out["warmup"] == 5 and out["dtype"] == "f64" and out["vs_baseline"] is None
    n = out["config"]["n_dof"]
    assert abs(out["value"] - n * 8 / (out["ms_per_step"] * 8e-3)) <= 1e-6 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["us_per_launch"] > 0 and "in-loop" in r["timing"] and r["assembly"]["default_mode"] == "row_gather"
    assert r["jacobi_pcg_iteration"]["bytes"] == 12 * out["config"]["nnz"] + 84 * n
    # roofline.traffic: measured in this very invocation (child runs under rocprofv3 --pmc) - or the line says why not
    if r["traffic"] is not None:
        assert r["traffic_source"].startswith("measured in this run") and 0.3 < r["traffic"] / r["bytes_per_launch"] < 3.0
    else:
        assert "not available" in r["traffic_source"] or "not measured" in r["traffic_source"]
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == out["unit"]
    assert out["config"]["gpu_over_cpu"] == pytest.approx(out["value"] / c["value"])
