"""Pins for the CPU oracle.  The reference has no golden vectors and its
numerics (dolfinx/PETSc) are not importable, so the oracle is pinned by analytic identities
and known-answer tests instead, plus the fixture in tests/golden/."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from hypothesis import given, settings, strategies as st
from scipy.special import j0, jn_zeros

from conftest import HEATING_CSV, ROOT, build_case, load_cfg
from oracle import heat_oracle as ho

coord = st.floats(min_value=0.0, max_value=1.0, allow_nan=False, allow_infinity=False)


@settings(max_examples=200, derandomize=True, deadline=None)
@given(st.tuples(coord, coord, coord, coord, coord, coord), st.floats(0.1, 1e7), st.floats(0.1, 1e4))
def test_element_identities_on_random_triangles(p, rho_c, kappa):
    zr = np.array(p).reshape(3, 2)
    area = 0.5 * abs((zr[1, 0] - zr[0, 0]) * (zr[2, 1] - zr[0, 1]) - (zr[2, 0] - zr[0, 0]) * (zr[1, 1] - zr[0, 1]))
    if area < 1e-4:
        return
    tri = np.array([[0, 1, 2]])
    Me, Ke = ho.element_matrices(zr, tri, np.array([rho_c]), np.array([kappa]))
    Me, Ke = Me[0], Ke[0]
    rbar = zr[:, 1].mean()
    # sum of the mass matrix = rho_c |K| rbar ; stiffness annihilates constants ; both symmetric
    assert np.isclose(Me.sum(), rho_c * area * rbar, rtol=1e-12, atol=0)
    assert np.allclose(Ke @ np.ones(3), 0.0, atol=1e-9 * max(1.0, np.abs(Ke).max()))
    assert np.array_equal(Me, Me.T) and np.allclose(Ke, Ke.T, rtol=1e-15, atol=0)
    # row sums of M: int rho_c phi_i r = rho_c |K| (2 r_i + r_j + r_k)/12
    assert np.allclose(Me.sum(axis=1), rho_c * area * (zr[:, 1] + zr[:, 1].sum()) / 12.0, rtol=1e-12)
    # closed forms == quadrature of the weak form as written in the reference
    Mq, Kq = ho.element_matrices_quadrature(zr, tri, np.array([rho_c]), np.array([kappa]))
    assert np.allclose(Me, Mq[0], rtol=1e-10, atol=1e-14 * np.abs(Me).max())
    assert np.allclose(Ke, Kq[0], rtol=1e-9, atol=1e-12 * max(np.abs(Ke).max(), 1e-300))
    # orientation does not matter
    Me2, Ke2 = ho.element_matrices(zr, np.array([[0, 2, 1]]), np.array([rho_c]), np.array([kappa]))
    perm = [0, 2, 1]
    assert np.allclose(Me2[0][np.ix_(perm, perm)], Me, rtol=1e-13) and np.allclose(Ke2[0][np.ix_(perm, perm)], Ke, rtol=1e-12)


def test_stiffness_energy_of_linear_fields():
    """u = z: grad = (1, 0), a(u,u) = kappa |K| rbar ; u = r likewise."""
    zr = np.array([[0.1, 0.2], [0.9, 0.3], [0.4, 0.8]])
    Me, Ke = ho.element_matrices(zr, np.array([[0, 1, 2]]), np.array([1.0]), np.array([2.5]))
    area = 0.5 * abs((0.8) * (0.6) - (0.3) * (0.1))
    rbar = zr[:, 1].mean()
    for u in (zr[:, 0], zr[:, 1]):
        assert np.isclose(u @ Ke[0] @ u, 2.5 * area * rbar, rtol=1e-13)


@pytest.fixture(scope="module")
def small_with_diamond():
    return build_case("geballe_with_diamond", 16.0)


def test_global_mass_equals_box_integrals(small_with_diamond):
    """sum(M) = sum_boxes rho_c * dz * (r2^2 - r1^2)/2."""
    cfg, stack, mesh = small_with_diamond
    tag_to_k = {mesh.material_tags[m.name]: m.properties["k"] for m in stack.materials}
    tag_to_rc = {mesh.material_tags[m.name]: m.properties["rho_cv"] for m in stack.materials}
    kappa, rho_c = ho.cell_coefficients(mesh.tags, tag_to_k, tag_to_rc)
    Me, Ke = ho.element_matrices(mesh.coords, mesh.tris.astype(np.int64), rho_c, kappa)
    M = ho.assemble_csr(len(mesh.coords), mesh.tris.astype(np.int64), Me)
    K = ho.assemble_csr(len(mesh.coords), mesh.tris.astype(np.int64), Ke)
    expect = sum(m.properties["rho_cv"] * (m.boundaries[1] - m.boundaries[0]) *
                 (m.boundaries[3] ** 2 - m.boundaries[2] ** 2) / 2 for m in stack.materials)
    assert np.isclose(M.sum(), expect, rtol=1e-11)
    assert np.abs(K @ np.ones(K.shape[0])).max() <= 1e-12 * np.abs(K.diagonal()).max()
    assert abs(M - M.T).max() == 0 and abs(K - K.T).max() <= 1e-16 * abs(K).max()


def test_constant_field_is_preserved_until_heating_starts(small_with_diamond):
    cfg, stack, mesh = small_with_diamond
    res = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV,
                                     num_steps=6, keep_fields=True)
    # t_k = (k+1)*7.5e-8 < 3.566e-7 for k = 0..3
    for k in range(4):
        assert np.abs(res["fields"][k] - 300.0).max() < 1e-9
    assert res["fields"][5].max() > 300.0 + 1e-6


def test_bc_overlap_later_entry_wins_in_no_diamond():
    """In the no-diamond geometry the node (z*, r_max) is in 'top' (300 K) and in the heated line;
    the inner BC is last in the list, so its Gaussian value applies."""
    cfg, stack, mesh = build_case("geballe_no_diamond", 8.0)
    res = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV, num_steps=0)
    sol = res["solver"]
    top = set(sol.bcs[2]["dofs"].tolist())
    inner = set(sol.bcs[3]["dofs"].tolist())
    shared = sorted(top & inner)
    assert len(shared) == 1
    node = shared[0]
    assert np.isclose(mesh.coords[node, 1], 20e-6) and np.isclose(mesh.coords[node, 0], stack.heated_z, atol=1e-12)
    t = 3.0e-6
    g = sol.bc_values(t)
    k = int(np.nonzero(sol.bc_dofs == node)[0][0])
    h_time, h_temp = ho.read_heating_csv(HEATING_CSV)
    assert np.isclose(g[k], ho.gaussian_bc_values(np.array([20e-6]), t, h_time, h_temp, 300.0, 1.32e-5)[0])
    assert g[k] > 300.5
    assert sol.bc_owner[k] == 3


def test_heating_curve_offset_and_clamping():
    h_time, h_temp = ho.read_heating_csv(HEATING_CSV)
    assert len(h_time) == 51 and np.all(np.diff(h_time) > 0)
    assert ho.heating_amplitude(0.0, h_time, h_temp, 300.0) == 300.0
    assert ho.heating_amplitude(h_time[0], h_time, h_temp, 300.0) == 300.0
    assert np.isclose(ho.heating_amplitude(1.0, h_time, h_temp, 300.0), h_temp[-1] - h_temp[0] + 300.0)
    g = ho.gaussian_bc_values(np.array([0.0, 6.6e-6]), 5e-6, h_time, h_temp, 300.0, 1.32e-5)
    assert np.isclose((g[1] - 300.0) / (g[0] - 300.0), 0.5)   # half maximum at r = fwhm/2


def _structured_mesh(nz, nr, z0, z1, r0, r1):
    z = np.linspace(z0, z1, nz + 1)
    r = np.linspace(r0, r1, nr + 1)
    Z, R = np.meshgrid(z, r, indexing="ij")
    coords = np.column_stack([Z.ravel(), R.ravel()])
    idx = lambda i, j: i * (nr + 1) + j
    tris = []
    for i in range(nz):
        for j in range(nr):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            tris += [[a, b, c], [a, c, d]]
    return coords, np.array(tris), z


def test_r_independent_slab_converges_to_1d_solution():
    """BC values uniform in r => the continuous axisymmetric solution does not depend on r and
    equals the 1-D slab solution along z.  On a triangulated grid the diagonal breaks the
    symmetry of the r-weighted stencil, so the discrete fields agree only up to O(h^2): the
    r-dependence and the distance to the 1-D P1 backward-Euler solution (same z nodes, same dt)
    must fall ~4x per simultaneous halving of h (SURVEY 8c pin 4, stated as a convergence test)."""
    dt, tag_to_k, tag_to_rc = 2e-8, {1: 10.0, 2: 352.0}, {1: 2.76e6, 2: 3.44e6}
    rdep, dist = [], []
    for n in (24, 48, 96):
        coords, tris, z = _structured_mesh(n, n, 0.0, 2e-6, 0.0, 3e-6)
        tags = np.where(coords[tris].mean(axis=1)[:, 0] < 1e-6, 1, 2)
        bcs = [{"dofs": ho.locate_row_dofs(coords, "left"), "value": lambda r, t: 300.0 + 4e9 * t + 0 * r},
               {"dofs": ho.locate_row_dofs(coords, "right"), "value": 300.0}]
        sol = ho.OracleSolver(coords, tris, tags, tag_to_k, tag_to_rc, dt, bcs, np.full(len(coords), 300.0))
        mid = 0.5 * (z[1:] + z[:-1])
        ref = ho.solve_1d_slab(z, np.where(mid < 1e-6, 2.76e6, 3.44e6), np.where(mid < 1e-6, 10.0, 352.0), dt,
                               np.full(n + 1, 300.0), [0, n], lambda t: np.array([300.0 + 4e9 * t, 300.0]), 10)
        for k in range(10):
            u = sol.step((k + 1) * dt).reshape(n + 1, n + 1)
        rdep.append(np.abs(u - u[:, :1]).max())
        dist.append(np.abs(u[:, n // 2] - ref[-1]).max())
    assert ref[-1][1] > 300.5
    assert rdep[0] / rdep[1] > 3.5 and rdep[1] / rdep[2] > 3.5 and rdep[2] < 0.06
    assert dist[0] / dist[1] > 3.5 and dist[1] / dist[2] > 3.5 and dist[2] < 0.006


def test_manufactured_bessel_mode_converges():
    """T = J0(alpha r) cos(beta z) exp(-lambda t) solves rho_c T_t = kappa (T_rr + T_r/r + T_zz) with
    natural BCs at r = 0, z = 0 and Dirichlet data elsewhere; the error falls ~4x per halving of
    h with dt ~ h^2 (O(h^2) + O(dt))."""
    kappa, rho_c, R, L = 2.0, 3.0, 1.0, 1.0
    alpha = jn_zeros(0, 1)[0] / R
    beta = np.pi / (2 * L)
    lam = kappa * (alpha ** 2 + beta ** 2) / rho_c
    errs = []
    for n, dt in ((8, 4e-3), (16, 1e-3), (32, 2.5e-4)):
        coords, tris, _ = _structured_mesh(n, n, 0.0, L, 0.0, R)
        exact = lambda t: j0(alpha * coords[:, 1]) * np.cos(beta * coords[:, 0]) * np.exp(-lam * t)
        right = ho.locate_row_dofs(coords, "right")      # z = L: cos = 0
        top = ho.locate_row_dofs(coords, "top")          # r = R: J0 = 0
        bcs = [{"dofs": right, "value": 0.0}, {"dofs": top, "value": 0.0}]
        sol = ho.OracleSolver(coords, tris, np.ones(len(tris), dtype=int), {1: kappa}, {1: rho_c}, dt, bcs, exact(0.0))
        steps = int(round(0.08 / dt))
        for k in range(steps):
            sol.step((k + 1) * dt)
        errs.append(np.abs(sol.u - exact(steps * dt)).max())
    assert errs[0] / errs[1] > 3.0 and errs[1] / errs[2] > 3.0
    assert errs[2] < 2e-3


def test_direct_solve_equals_pcg_on_the_same_system(small_with_diamond):
    cfg, stack, mesh = small_with_diamond
    res = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV,
                                     num_steps=8, keep_fields=True)
    sol = res["solver"]
    # redo the last step with Jacobi-PCG on A_hat
    sol.u = res["fields"][6].copy()
    g = sol.bc_values(8 * res["dt"])
    b = sol.rhs(g)
    A = sol.Ahat
    dinv = 1.0 / A.diagonal()
    x = sol.u.copy()
    x[sol.bc_dofs] = g
    r = b - A @ x
    z = dinv * r
    p = z.copy()
    rz = r @ z
    for _ in range(5000):
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p
        r -= a * Ap
        z = dinv * r
        if np.sqrt(z @ z) <= 1e-12 * np.linalg.norm(dinv * b):
            break
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    assert np.abs(x - res["fields"][7]).max() < 1e-7


def test_eliminated_matrix_has_unit_rows_and_zero_columns(small_with_diamond):
    cfg, stack, mesh = small_with_diamond
    sol = ho.run_reference_algorithm(cfg, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags, HEATING_CSV,
                                     num_steps=0)["solver"]
    B = sol.bc_dofs
    Ah = sol.Ahat.tocsr()
    assert np.allclose(Ah[B][:, B].toarray(), np.eye(len(B)))
    free = np.setdiff1d(np.arange(Ah.shape[0]), B)
    assert abs(Ah[free][:, B]).max() == 0 and abs(Ah[B][:, free]).max() == 0
    assert abs(Ah[free][:, free] - sol.A[free][:, free]).max() == 0


def test_golden_fixture_is_reproduced():
    """tests/golden/with_diamond_tiny.npz was produced by tests/golden/make_golden.py from this
    oracle; it guards the oracle (and through the GPU tests the HIP path) against drift."""
    path = os.path.join(ROOT, "tests", "golden", "with_diamond_tiny.npz")
    g = np.load(path)
    cfg = load_cfg("geballe_with_diamond")
    from heatflow_amd.geometry import scale_mesh_sizes
    cfg = scale_mesh_sizes(cfg, float(g["mesh_scale"]))
    mtags = {str(k): int(v) for k, v in zip(g["material_names"], g["material_tag_values"])}
    res = ho.run_reference_algorithm(cfg, g["coords"], g["tris"], g["tags"], mtags, HEATING_CSV,
                                     num_steps=int(g["fields"].shape[0]), keep_fields=True)
    assert np.abs(res["fields"] - g["fields"]).max() < 1e-9
    assert np.allclose(res["times"], g["times"], rtol=0, atol=1e-20)
