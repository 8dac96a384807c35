"""Oracle-backed stand-in for HeatflowHIP, for CPU-only tests of the host logic (drivers,
sweeps, sharding).  TEST CODE: it lives under tests/ and is never imported by heatflow_amd."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import heat_oracle as ho


def fake_pattern_blob(coords, tris, tags, device_id=0):
    """What a test passes as ``pattern_builder``: a byte string that depends on the mesh (the real one is
    HeatflowHIP.export_pattern)."""
    import hashlib
    h = hashlib.sha1(np.ascontiguousarray(tris, dtype=np.int32).tobytes() + np.ascontiguousarray(tags, dtype=np.int32).tobytes())
    return np.frombuffer(b"FAKEPATT" + h.digest(), dtype=np.uint8).copy()


class OracleBackend:
    def __init__(self, device_id=0):
        self.n = self.n_e = self.n_bc = 0
        self.nnz = 0
        self.bc_dofs = np.zeros(0, dtype=np.int64)
        self._lu = None
        self.closed = False
        self.assemble_calls = 0
        self.set_mesh_calls = 0

    def close(self):
        self.closed = True

    def set_mesh(self, coords, tris, tags, pattern=None):
        if pattern is not None:                  # stand-in for hf_set_mesh_prebuilt: the blob must belong to this mesh
            if not np.array_equal(np.asarray(pattern), fake_pattern_blob(coords, tris, tags)):
                raise ValueError("pattern blob does not belong to this mesh")
            self.prebuilt_calls = getattr(self, "prebuilt_calls", 0) + 1
        self.coords = np.asarray(coords, dtype=np.float64)
        self.tris = np.asarray(tris, dtype=np.int64)
        self.tags = np.asarray(tags)
        self.n, self.n_e = len(self.coords), len(self.tris)
        self.u = np.zeros(self.n)
        self.set_mesh_calls += 1

    # -- hierarchy hand-over stand-in (hf_amg_export / hf_amg_install): the oracle has no hierarchy, the calls are recorded
    def amg_export(self, into=None):
        self.amg_export_calls = getattr(self, "amg_export_calls", 0) + 1
        return np.frombuffer(b"FAKEAMG!" + np.int64(self.n).tobytes(), dtype=np.uint8).copy()

    def amg_install(self, blob):
        b = np.asarray(blob, dtype=np.uint8).tobytes()
        if b[:8] != b"FAKEAMG!" or np.frombuffer(b[8:16], dtype=np.int64)[0] != self.n:
            raise ValueError("hierarchy blob does not belong to this mesh")
        self.amg_install_calls = getattr(self, "amg_install_calls", 0) + 1

    def set_materials(self, tags, kappa, rho_c):
        self.tag_to_k = {int(t): float(k) for t, k in zip(tags, kappa)}
        self.tag_to_rc = {int(t): float(c) for t, c in zip(tags, rho_c)}

    def set_precond(self, kind=0, reuse=False):
        self.precond = kind

    def set_dirichlet(self, dofs):
        self.bc_dofs = np.asarray(dofs, dtype=np.int64)
        self.n_bc = len(self.bc_dofs)

    def assemble(self, dt, mode=0):
        self._dt = dt
        kappa, rho_c = ho.cell_coefficients(self.tags, self.tag_to_k, self.tag_to_rc)
        Me, Ke = ho.element_matrices(self.coords, self.tris, rho_c, kappa)
        self.M = ho.assemble_csr(self.n, self.tris, Me)
        self.A = ho.assemble_csr(self.n, self.tris, Me + dt * Ke)
        self.nnz = self.A.nnz
        self.Ahat = ho.eliminate_dirichlet(self.A, self.bc_dofs) if self.n_bc else self.A
        self.A_lift = self.A[:, self.bc_dofs].tocsr() if self.n_bc else None
        self._lu = spla.splu(self.Ahat.tocsc())
        self.assemble_calls += 1

    def set_state(self, u):
        self.u = np.array(u, dtype=np.float64).copy()

    def get_state(self):
        return self.u.copy()

    def sample(self, nodes):
        return self.u[np.asarray(nodes)].copy()

    def step(self, g, rtol=1e-10, atol=0.0, max_it=20000):
        b = self.M @ self.u
        if self.n_bc:
            b -= self.A_lift @ g
            b[self.bc_dofs] = g
        self.u = self._lu.solve(b)
        return 1, 0.0

    def run(self, g_all, rtol=1e-10, atol=0.0, max_it=20000, nodes=None):
        ns = 0 if nodes is None else len(nodes)
        samples = np.empty((len(g_all), ns))
        for k, g in enumerate(g_all):
            self.step(g)
            if ns:
                samples[k] = self.u[np.asarray(nodes)]
        return samples, np.ones(len(g_all), dtype=np.int32)

    # -- batched loop stand-in: every column solved on its own with its own factorisation --------
    def batch_begin(self, nv, per_column_operator=False):
        self.batch_nv, self._bpercol = int(nv), bool(per_column_operator)
        self._bcols = [dict(u=None, M=self.M, lu=self._lu, lift=self.A_lift) for _ in range(nv)]
        self.batch_begin_calls = getattr(self, "batch_begin_calls", 0) + 1

    def batch_load_column(self, j):
        self._bcols[j].update(M=self.M, lu=self._lu, lift=self.A_lift)

    def batch_set_affine(self, tags, delta):
        ref = dict(self.tag_to_k)
        for j, d in enumerate(delta):
            self.tag_to_k = {t: (k + float(d) if t in set(int(x) for x in tags) else k) for t, k in ref.items()}
            self.assemble(self._dt)
            self.batch_load_column(j)
        self.tag_to_k = ref
        self.assemble(self._dt)
        self.batch_affine_calls = getattr(self, "batch_affine_calls", 0) + 1

    def batch_set_state(self, j, u):
        self._bcols[j]["u"] = np.array(u, dtype=np.float64).copy()

    def batch_get_state(self, j):
        return self._bcols[j]["u"].copy()

    def batch_run(self, g_all, rtol=1e-10, atol=0.0, max_it=20000, nodes=None, flux_nodes=None, flux_components=2,
                  flux_rtol=None, flux_max_it=5000):
        nsteps, _, nv = g_all.shape
        ns = 0 if nodes is None else len(nodes)
        comps = [c for c in (0, 1) if (int(flux_components) >> c) & 1]
        flux = None if flux_nodes is None else np.empty((nsteps, len(comps), nv, len(flux_nodes)))
        samples = np.empty((nsteps, nv, ns))
        for s in range(nsteps):
            for j, col in enumerate(self._bcols):
                b = col["M"] @ col["u"]
                if self.n_bc:
                    g = np.ascontiguousarray(g_all[s, :, j])
                    b -= col["lift"] @ g
                    b[self.bc_dofs] = g
                col["u"] = col["lu"].solve(b)
                if ns:
                    samples[s, j] = col["u"][np.asarray(nodes)]
                if flux is not None:
                    g2 = self._proj.project(col["u"])[np.asarray(flux_nodes)]
                    for q, c in enumerate(comps):
                        flux[s, q, j] = g2[:, c]
        self.batch_flux_calls = getattr(self, "batch_flux_calls", 0) + (flux is not None)
        if flux is not None:
            return samples, np.ones((nsteps, nv), dtype=np.int32), flux
        return samples, np.ones((nsteps, nv), dtype=np.int32)

    def batch_end(self):
        self.batch_nv = 0

    def flux_setup(self):
        self._proj = ho.GradientProjector(self.coords, self.tris)

    def flux_project(self, rtol=1e-10, max_it=5000, want_z=True, want_r=True):
        g = self._proj.project(self.u)
        return (g[:, 0].copy() if want_z else None), (g[:, 1].copy() if want_r else None)

    def flux_solve(self, rtol=1e-10, max_it=5000, want_z=True, want_r=True):
        self._grad = self._proj.project(self.u)
        return np.ones(2, dtype=np.int32)

    def flux_sample(self, nodes, want_z=True, want_r=True):
        g = self._grad[np.asarray(nodes)]
        return (g[:, 0].copy() if want_z else None), (g[:, 1].copy() if want_r else None)

    def last_gpu_ms(self):
        return 0.0
