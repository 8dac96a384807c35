"""CPU oracle: float64 numpy/scipy restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``heatflow_amd/`` imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the timed CPU baseline.

PARITY UNPINNED by the reference itself: the reference (cebarker1000/heatflow)
has no golden vectors or known-answer tests, and its numerics live in
dolfinx/ufl/ffcx/petsc4py+MUMPS/gmsh, none of which is importable here (plain
ModuleNotFoundError, no version pins in the repo).  The oracle is therefore
pinned by analytic identities and known-answer tests in tests/test_oracle.py
(constant preservation, closed forms vs quadrature, global mass integrals,
1-D slab equivalence, a manufactured Bessel mode) and by fixtures generated
from it (tests/golden/, script tests/golden/make_golden.py).

What it restates (file:line in /root/reference):
  * forms        run_with_diamond.py:321-337 == run_no_diamond.py:271-287 ==
                 space/space_and_forms.py:98-116
  * coefficients run_with_diamond.py:286-301 (cell tag -> kappa, rho*cv)
  * BC location  dirichlet_bc/bc.py:32-118, values bc.py:128-137,
                 run_with_diamond.py:343-374 (list order left,right,top,inner)
  * assemble-once + symmetric elimination + LU   run_with_diamond.py:379-394
  * time loop    run_with_diamond.py:456-504
  * geometry     run_with_diamond.py:60-96, run_no_diamond.py:62-83
"""
from __future__ import annotations

import csv
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# --------------------------------------------------------------------------------------
# Element matrices (reference forms, run_with_diamond.py:328-335)
#   a(u,v) = int rho_c u v r dx + dt int kappa grad u . grad v r dx,   r = x[1]
# P1 on an affine triangle: the mass integrand is cubic, the stiffness integrand linear,
# so the degree-3 rule FFCx picks integrates both exactly; closed forms:
#   M_ii = rho_c |K| (3 r_i + r_j + r_k)/30,  M_ij = rho_c |K| (2 r_i + 2 r_j + r_k)/60
#   K_ij = kappa |K| rbar grad(phi_i).grad(phi_j),  rbar = (r_1+r_2+r_3)/3
# --------------------------------------------------------------------------------------


def element_matrices(zr, tri, rho_c, kappa):
    """Closed-form r-weighted P1 mass and stiffness.  Returns (Me, Ke), each (n_e, 3, 3)."""
    p = zr[tri]                                   # (ne, 3, 2)
    z, r = p[:, :, 0], p[:, :, 1]
    # signed double area; |K| uses abs so orientation does not matter
    d = (z[:, 1] - z[:, 0]) * (r[:, 2] - r[:, 0]) - (z[:, 2] - z[:, 0]) * (r[:, 1] - r[:, 0])
    area = 0.5 * np.abs(d)
    # grad phi_i = (b_i, c_i)/d with b_i = r_j - r_k, c_i = z_k - z_j  (i,j,k cyclic)
    b = np.stack([r[:, 1] - r[:, 2], r[:, 2] - r[:, 0], r[:, 0] - r[:, 1]], axis=1)
    c = np.stack([z[:, 2] - z[:, 1], z[:, 0] - z[:, 2], z[:, 1] - z[:, 0]], axis=1)
    rsum = r.sum(axis=1)
    rbar = rsum / 3.0
    gg = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (d * d)[:, None, None]
    Ke = (kappa * area * rbar)[:, None, None] * gg
    Me = np.empty_like(Ke)
    for i in range(3):
        for j in range(3):
            if i == j:
                w = (2.0 * r[:, i] + rsum) / 30.0              # (3 r_i + r_j + r_k)/30
            else:
                w = (rsum + (r[:, i] + r[:, j])) / 60.0        # (2 r_i + 2 r_j + r_k)/60, symmetric in i,j
            Me[:, i, j] = rho_c * area * w
    return Me, Ke


# Dunavant 6-point rule, exact to degree 4 (barycentric points, weights sum to 1)
_QA, _QB = 0.445948490915965, 0.091576213509771
_QW1, _QW2 = 0.223381589678011, 0.109951743655322
_QPTS = np.array([
    [_QA, _QA, 1 - 2 * _QA], [_QA, 1 - 2 * _QA, _QA], [1 - 2 * _QA, _QA, _QA],
    [_QB, _QB, 1 - 2 * _QB], [_QB, 1 - 2 * _QB, _QB], [1 - 2 * _QB, _QB, _QB]])
_QWTS = np.array([_QW1] * 3 + [_QW2] * 3)


def element_matrices_quadrature(zr, tri, rho_c, kappa):
    """Same matrices by numerical quadrature of the weak form as written in the
    reference (what an FFCx-generated kernel does); pins the closed forms."""
    p = zr[tri]
    z, r = p[:, :, 0], p[:, :, 1]
    d = (z[:, 1] - z[:, 0]) * (r[:, 2] - r[:, 0]) - (z[:, 2] - z[:, 0]) * (r[:, 1] - r[:, 0])
    area = 0.5 * np.abs(d)
    b = np.stack([r[:, 1] - r[:, 2], r[:, 2] - r[:, 0], r[:, 0] - r[:, 1]], axis=1) / d[:, None]
    c = np.stack([z[:, 2] - z[:, 1], z[:, 0] - z[:, 2], z[:, 1] - z[:, 0]], axis=1) / d[:, None]
    ne = len(tri)
    Me = np.zeros((ne, 3, 3))
    Ke = np.zeros((ne, 3, 3))
    for q in range(len(_QWTS)):
        lam = _QPTS[q]                                  # P1 basis values at the point
        rq = r @ lam
        w = _QWTS[q] * area
        for i in range(3):
            for j in range(3):
                Me[:, i, j] += w * rho_c * lam[i] * lam[j] * rq
                Ke[:, i, j] += w * kappa * (b[:, i] * b[:, j] + c[:, i] * c[:, j]) * rq
    return Me, Ke


def assemble_csr(n, tri, Ae):
    """Scatter-add element matrices into CSR (sorted column indices, duplicates summed).
    Restates dolfinx ``assemble_matrix`` without BCs (run_with_diamond.py:381)."""
    rows = np.repeat(tri, 3, axis=1).ravel()
    cols = np.tile(tri, (1, 3)).ravel()
    A = sp.coo_matrix((Ae.reshape(-1), (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def cell_coefficients(tags, tag_to_k, tag_to_rho_cv):
    """kappa[c] = k[tag[c]], rho_cv[c] = rho_cv[tag[c]] (run_with_diamond.py:286-292)."""
    tags = np.asarray(tags)
    tk = np.zeros(int(tags.max()) + 1)
    tc = np.zeros(int(tags.max()) + 1)
    for t in np.unique(tags):
        tk[int(t)] = tag_to_k[int(t)]           # KeyError for an unmapped tag, like the dict lookup
        tc[int(t)] = tag_to_rho_cv[int(t)]
    return tk[tags], tc[tags]


# --------------------------------------------------------------------------------------
# Dirichlet rows (dirichlet_bc/bc.py)
# --------------------------------------------------------------------------------------


def locate_row_dofs(coords, location, coord=None, length=None, center=None, width=1e-10):
    """DOF indices of a RowDirichletBC (bc.py:32-106): np.isclose(..., atol=width) on the
    edge / line coordinate, optional |s - center| <= length/2 + 1e-14 clip."""
    x0, x1 = coords[:, 0], coords[:, 1]
    xmin, xmax, ymin, ymax = x0.min(), x0.max(), x1.min(), x1.max()
    xmid, ymid = 0.5 * (xmin + xmax), 0.5 * (ymin + ymax)
    half = None if length is None else 0.5 * length
    if location in ("x", "y") and center is None:
        center = xmid if location == "x" else ymid

    def clip(vals, c):
        if half is None:
            return np.ones_like(vals, dtype=bool)
        return np.abs(vals - c) <= half + 1e-14

    if location == "left":
        mask = np.isclose(x0, xmin, atol=width) & clip(x1, ymid)
    elif location == "right":
        mask = np.isclose(x0, xmax, atol=width) & clip(x1, ymid)
    elif location == "bottom":
        mask = np.isclose(x1, ymin, atol=width) & clip(x0, xmid)
    elif location == "top":
        mask = np.isclose(x1, ymax, atol=width) & clip(x0, xmid)
    elif location == "x":
        mask = np.isclose(x0, float(coord), atol=width) & clip(x1, center)
    elif location == "y":
        mask = np.isclose(x1, float(coord), atol=width) & clip(x0, center)
    else:
        raise ValueError("Unknown location keyword.")
    dofs = np.nonzero(mask)[0]
    if dofs.size == 0:
        raise RuntimeError("No DOFs found for requested BC location/length.")
    return dofs


def read_heating_csv(path, column="temp"):
    """(time, temp) sorted by time, non-numeric rows dropped (run_with_diamond.py:254-265)."""
    t, T = [], []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        if "temp" not in rd.fieldnames:
            raise ValueError(f"Heating CSV file {path} must contain a 'temp' column")
        if "time" not in rd.fieldnames:
            raise ValueError(f"Heating CSV file {path} must contain a 'time' column")
        for row in rd:
            try:
                a, b = float(row["time"]), float(row[column])
            except (TypeError, ValueError):
                continue
            if np.isnan(a) or np.isnan(b):
                continue
            t.append(a)
            T.append(b)
    t, T = np.array(t), np.array(T)
    o = np.argsort(t, kind="stable")
    return t[o], T[o]


def heating_amplitude(t, h_time, h_temp, ic_temp):
    """heating_offset(t) = interp(t) - (temp[0] - ic_temp), ends clamped
    (run_with_diamond.py:343-351)."""
    return float(np.interp(t, h_time, h_temp, left=h_temp[0], right=h_temp[-1])) - (h_temp[0] - ic_temp)


def gaussian_bc_values(r, t, h_time, h_temp, ic_temp, fwhm):
    """(amp - ic) exp(-4 ln2 r^2 / fwhm^2) + ic  (run_with_diamond.py:354-359)."""
    coeff = -4.0 * np.log(2.0) / fwhm ** 2
    amp = heating_amplitude(t, h_time, h_temp, ic_temp)
    return (amp - ic_temp) * np.exp(coeff * (r - 0.0) ** 2) + ic_temp


def merge_bcs(bc_list):
    """Union of BC dof sets; where sets overlap the LATER list entry wins
    (dolfinx applies bcs in list order in set_bc / apply_lifting).
    ``bc_list`` = [(dofs, owner_id), ...].  Returns (dofs sorted, owner per dof)."""
    owner = {}
    for dofs, who in bc_list:
        for d in dofs:
            owner[int(d)] = who
    dofs = np.array(sorted(owner), dtype=np.int64)
    return dofs, np.array([owner[int(d)] for d in dofs], dtype=np.int64)


def eliminate_dirichlet(A, bc_dofs):
    """Rows and columns of the BC dofs zeroed, unit diagonal - what dolfinx
    ``assemble_matrix(form, bcs)`` leaves (run_with_diamond.py:381)."""
    n = A.shape[0]
    keep = np.ones(n)
    keep[bc_dofs] = 0.0
    D = sp.diags(keep)
    Ah = (D @ A @ D).tocsr()
    Ah = Ah + sp.diags(1.0 - keep)
    Ah = Ah.tocsr()
    Ah.sort_indices()
    return Ah


# --------------------------------------------------------------------------------------
# Geometry (restated independently of heatflow_amd.geometry)
# --------------------------------------------------------------------------------------


def stack_boxes(cfg):
    """{name: [zmin,zmax,rmin,rmax]}, heated_z, r_sample for the with/no-diamond stacks
    (run_with_diamond.py:60-96, run_no_diamond.py:62-83)."""
    g = lambda m, k: float(cfg["mats"][m][k])
    zs, zc, zpi, zoi = g("p_sample", "z"), g("p_coupler", "z"), g("p_ins", "z"), g("o_ins", "z")
    rs = g("p_sample", "r")
    if "p_diam" in cfg["mats"]:
        zd = g("p_diam", "z")
        rmax = rs + g("gasket", "r") + g("g_ins", "r")
        zmin = -(zs / 2) - zpi - zc - zd
        zmax = (zs / 2) + zoi + zc + zd
        pd = [zmin, zmin + zd, 0.0, rmax]
        od = [zmax - zd, zmax, 0.0, rmax]
        pi = [pd[1], pd[1] + zpi, 0.0, rs]
        oi = [od[0] - zoi, od[0], 0.0, rs]
        pc = [pi[1], pi[1] + zc, 0.0, rs]
        oc = [oi[0] - zc, oi[0], 0.0, rs]
        sa = [pc[1], pc[1] + zs, 0.0, rs]
        gi = [pd[1], od[0], rs, rs + g("g_ins", "r")]
        ga = [pd[1], od[0], gi[3], rmax]
        boxes = {"p_diam": pd, "p_ins": pi, "p_coupler": pc, "p_sample": sa, "o_coupler": oc,
                 "o_ins": oi, "o_diam": od, "gasket": ga, "g_ins": gi}
    else:
        zmin = -(zs / 2) - zpi - zc
        pi = [zmin, zmin + zpi, 0.0, g("p_ins", "r")]
        pc = [pi[1], pi[1] + zc, 0.0, g("p_coupler", "r")]
        sa = [pc[1], pc[1] + zs, 0.0, rs]
        oc = [sa[1], sa[1] + zc, 0.0, g("p_coupler", "r")]
        oi = [oc[1], oc[1] + zoi, 0.0, g("o_ins", "r")]
        boxes = {"p_ins": pi, "p_coupler": pc, "p_sample": sa, "o_coupler": oc, "o_ins": oi}
    return boxes, boxes["p_coupler"][0], rs


# --------------------------------------------------------------------------------------
# The reference algorithm: assemble once, factor once, solve per step
# --------------------------------------------------------------------------------------


class OracleSolver:
    """Backward-Euler loop of run_with_diamond.py:379-394, 469-481 on given mesh arrays.

    Parameters
    ----------
    coords (n,2), tris (n_e,3), tags (n_e,) : the mesh (x = z, y = r)
    tag_to_k, tag_to_rho_cv : {tag: value}
    dt : time step
    bcs : list of dicts {"dofs": int array, "value": float | callable(r_array, t) -> array}
          in the reference's list order (later wins on overlap)
    """

    def __init__(self, coords, tris, tags, tag_to_k, tag_to_rho_cv, dt, bcs, u0):
        t0 = time.perf_counter()
        self.coords = np.asarray(coords, dtype=np.float64)
        self.tris = np.asarray(tris, dtype=np.int64)
        n = len(self.coords)
        kappa, rho_c = cell_coefficients(np.asarray(tags), tag_to_k, tag_to_rho_cv)
        Me, Ke = element_matrices(self.coords, self.tris, rho_c, kappa)
        self.M = assemble_csr(n, self.tris, Me)
        self.K = assemble_csr(n, self.tris, Ke)
        self.A = assemble_csr(n, self.tris, Me + dt * Ke)          # unconstrained a(u,v)
        self.dt = dt
        self.bcs = bcs
        self.bc_dofs, self.bc_owner = merge_bcs([(b["dofs"], k) for k, b in enumerate(bcs)])
        self.Ahat = eliminate_dirichlet(self.A, self.bc_dofs)
        self.A_lift = self.A[:, self.bc_dofs].tocsr()              # columns used by apply_lifting
        self.u = np.array(u0, dtype=np.float64).copy()
        self.t_assemble = time.perf_counter() - t0
        self._lu = None
        self.t_factor = 0.0

    def bc_values(self, t):
        """g_B(t) in the order of ``self.bc_dofs`` (bc.py:128-137 per BC, later BC wins)."""
        g = np.empty(len(self.bc_dofs))
        for k, b in enumerate(self.bcs):
            sel = self.bc_owner == k
            if not sel.any():
                continue
            v = b["value"]
            if callable(v):
                g[sel] = v(self.coords[self.bc_dofs[sel], 1], t)
            else:
                g[sel] = float(v)
        return g

    def rhs(self, g):
        """b = M u^n ; b -= A[:,B] g ; b[B] = g   (run_with_diamond.py:474-479)."""
        b = self.M @ self.u
        b -= self.A_lift @ g
        b[self.bc_dofs] = g
        return b

    def factor(self):
        if self._lu is None:
            t0 = time.perf_counter()
            self._lu = spla.splu(self.Ahat.tocsc(), permc_spec="MMD_AT_PLUS_A",
                                 options=dict(SymmetricMode=True))
            self.t_factor = time.perf_counter() - t0
        return self._lu

    def step(self, t):
        """Advance to time ``t`` (solve in place into u, run_with_diamond.py:480)."""
        g = self.bc_values(t)
        b = self.rhs(g)
        self.u = self.factor().solve(b)
        return self.u


def run_reference_algorithm(cfg, coords, tris, tags, material_tags, heating_csv, num_steps=None,
                            watcher_nodes=None, keep_fields=False, second_line=None):
    """cfg + mesh -> per-step temperatures, following run_with_diamond.run_simulation
    (:254-274 heating, :286-301 coefficients, :307-319 dt/ic, :343-374 BCs, :469-504 loop).

    Returns dict(times, watchers (steps, n_w), fields (steps, n) if keep_fields, solver).
    """
    boxes, heated_z, r_sample = stack_boxes(cfg)
    g = lambda m, k: float(cfg["mats"][m][k])
    tag_to_k = {material_tags[m]: g(m, "k") for m in boxes}
    tag_to_rc = {material_tags[m]: g(m, "rho") * g(m, "cv") for m in boxes}
    t_final = float(cfg["timing"]["t_final"])
    nsteps_cfg = int(cfg["timing"]["num_steps"])
    dt = t_final / nsteps_cfg
    ic = float(cfg["heating"]["ic_temp"])
    fwhm = float(cfg["heating"]["fwhm"])
    h_time, h_temp = read_heating_csv(heating_csv)
    coords = np.asarray(coords, dtype=np.float64)

    bcs = [
        {"dofs": locate_row_dofs(coords, "left"), "value": ic},
        {"dofs": locate_row_dofs(coords, "right"), "value": ic},
        {"dofs": locate_row_dofs(coords, "top"), "value": ic},
        {"dofs": locate_row_dofs(coords, "x", coord=heated_z, length=abs(r_sample) * 2, center=0.0),
         "value": lambda r, t: gaussian_bc_values(r, t, h_time, h_temp, ic, fwhm)},
    ]
    if second_line is not None:
        # extension (no reference implementation): second Gaussian line at z = second_line driven by `oside`
        o_time, o_temp = read_heating_csv(heating_csv, column="oside")
        bcs.append({"dofs": locate_row_dofs(coords, "x", coord=second_line, length=abs(r_sample) * 2, center=0.0),
                    "value": lambda r, t: gaussian_bc_values(r, t, o_time, o_temp, ic, fwhm)})
    sol = OracleSolver(coords, tris, tags, tag_to_k, tag_to_rc, dt, bcs, np.full(len(coords), ic))
    steps = nsteps_cfg if num_steps is None else int(num_steps)
    times, watch, fields = [], [], []
    for s in range(steps):
        t = (s + 1) * dt
        u = sol.step(t)
        times.append(t)
        if watcher_nodes is not None:
            watch.append(u[np.asarray(watcher_nodes)].copy())
        if keep_fields:
            fields.append(u.copy())
    return {"times": np.array(times), "watchers": np.array(watch), "fields": np.array(fields) if keep_fields else None,
            "solver": sol, "dt": dt}


# --------------------------------------------------------------------------------------
# Read-flux projection (run_no_diamond.py:471-491 set-up, :543-550 per step)
#   a_proj = inner(g, w) r dx on vector P1, rhs = inner(grad(u_n), w) r dx  ->  grad_smooth
# The 2n x 2n block system is block-diagonal in the two components: each is M_r(1) g_c = b_c.
# --------------------------------------------------------------------------------------


class GradientProjector:
    def __init__(self, coords, tris):
        self.coords = np.asarray(coords, dtype=np.float64)
        self.tris = np.asarray(tris, dtype=np.int64)
        ne = len(self.tris)
        Me, _ = element_matrices(self.coords, self.tris, np.ones(ne), np.zeros(ne))
        self.M1 = assemble_csr(len(self.coords), self.tris, Me)
        self._lu = spla.splu(self.M1.tocsc())
        p = self.coords[self.tris]
        z, r = p[:, :, 0], p[:, :, 1]
        self.d = (z[:, 1] - z[:, 0]) * (r[:, 2] - r[:, 0]) - (z[:, 2] - z[:, 0]) * (r[:, 1] - r[:, 0])
        self.bz = np.stack([r[:, 1] - r[:, 2], r[:, 2] - r[:, 0], r[:, 0] - r[:, 1]], axis=1) / self.d[:, None]
        self.br = np.stack([z[:, 2] - z[:, 1], z[:, 0] - z[:, 2], z[:, 1] - z[:, 0]], axis=1) / self.d[:, None]
        area = 0.5 * np.abs(self.d)
        self.w = area[:, None] * (r + r.sum(axis=1)[:, None]) / 12.0      # int_e phi_i r dx

    def project(self, u):
        """(n, 2) array [dT/dz, dT/dr] = grad_smooth.x.array.reshape(-1, 2) (run_no_diamond.py:553)."""
        ue = np.asarray(u)[self.tris]
        gz = (ue * self.bz).sum(axis=1)
        gr = (ue * self.br).sum(axis=1)
        n = len(self.coords)
        rhs_z = np.bincount(self.tris.ravel(), weights=(gz[:, None] * self.w).ravel(), minlength=n)
        rhs_r = np.bincount(self.tris.ravel(), weights=(gr[:, None] * self.w).ravel(), minlength=n)
        return np.column_stack([self._lu.solve(rhs_z), self._lu.solve(rhs_r)])


# --------------------------------------------------------------------------------------
# 1-D P1 interval model (run_no_diamond_1d.py:537-546): no r weight
# --------------------------------------------------------------------------------------


def solve_1d_slab(z, rho_c_cells, kappa_cells, dt, u0, bc_nodes, bc_value_fn, num_steps, source_fn=None):
    """Backward Euler for rho_c u_t = (kappa u_z)_z + s on nodes ``z`` (sorted) with P1
    intervals, Dirichlet nodes ``bc_nodes`` (values ``bc_value_fn(t)`` -> array),
    consistent mass matrix - the un-weighted forms of run_no_diamond_1d.py:537-546;
    ``source_fn(t)`` -> nodal values of the P1 source (dt * int s v dx, :546).
    Returns (num_steps, n) array."""
    n = len(z)
    h = np.diff(z)
    main_m = np.zeros(n)
    off_m = rho_c_cells * h / 6.0
    main_m[:-1] += rho_c_cells * h / 3.0
    main_m[1:] += rho_c_cells * h / 3.0
    main_k = np.zeros(n)
    off_k = -kappa_cells / h
    main_k[:-1] += kappa_cells / h
    main_k[1:] += kappa_cells / h
    M = sp.diags([off_m, main_m, off_m], [-1, 0, 1]).tocsr()
    A = (M + dt * sp.diags([off_k, main_k, off_k], [-1, 0, 1])).tocsr()
    main_1 = np.zeros(n)
    main_1[:-1] += h / 3.0
    main_1[1:] += h / 3.0
    M1 = sp.diags([h / 6.0, main_1, h / 6.0], [-1, 0, 1]).tocsr()
    bc_nodes = np.asarray(bc_nodes)
    Ah = eliminate_dirichlet(A, bc_nodes)
    lu = spla.splu(Ah.tocsc())
    Al = A[:, bc_nodes].tocsr()
    u = np.array(u0, dtype=np.float64).copy()
    out = []
    for s in range(num_steps):
        g = np.asarray(bc_value_fn((s + 1) * dt), dtype=np.float64)
        b = M @ u - Al @ g
        if source_fn is not None:
            b = b + dt * (M1 @ np.asarray(source_fn((s + 1) * dt), dtype=np.float64))
        b[bc_nodes] = g
        u = lu.solve(b)
        out.append(u.copy())
    return np.array(out)
