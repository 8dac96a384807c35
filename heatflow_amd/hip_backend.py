"""ctypes binding of libheatflow_hip.so (include/heatflow_hip.h).

This is the only device path: there is no CPU fallback.  If the shared library
is missing or no HIP device is present, construction raises ``HipUnavailable``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HEATFLOW_HIP_LIB") or os.path.join(_HERE, "csrc", "libheatflow_hip.so")  # env: A/B builds

HF_OK, HF_ERR_ARG, HF_ERR_STATE, HF_ERR_HIP, HF_ERR_NOCONV, HF_ERR_ALLOC = 0, -1, -2, -3, -4, -5
ASM_LDS_ATOMIC, ASM_LDS_COLORED, ASM_GLOBAL_ATOMIC, ASM_ROW_GATHER = 0, 1, 2, 3
PC_JACOBI, PC_AMG = 0, 1
K_SPMV, K_PCG_SPMV, K_PCG_UPDATE, K_PCG_DIR, K_ASSEMBLE, K_RHS, K_STREAM_READ = range(7)
BATCH_SHARED, BATCH_PER_COLUMN, BATCH_AFFINE = 0, 1, 2

EXPORTS = [
    "hf_version", "hf_create", "hf_destroy", "hf_last_error", "hf_set_mesh", "hf_set_mesh_prebuilt", "hf_pattern_export_size",
    "hf_pattern_export", "hf_amg_export_size", "hf_amg_export", "hf_amg_install", "hf_set_materials",
    "hf_update_kappa", "hf_set_dirichlet", "hf_assemble", "hf_set_precond", "hf_set_start_vector", "hf_get_response_solves", "hf_get_amg_info", "hf_get_amg_fallbacks", "hf_set_state", "hf_get_state", "hf_sample", "hf_step", "hf_run",
    "hf_batch_begin", "hf_batch_load_column", "hf_batch_set_affine", "hf_batch_set_state", "hf_batch_get_state", "hf_batch_run", "hf_batch_run_flux", "hf_batch_end",
    "hf_flux_setup", "hf_flux_project", "hf_flux_solve", "hf_flux_sample", "hf_get_sizes", "hf_get_csr", "hf_spmv", "hf_time_kernel", "hf_set_profile", "hf_get_profile", "hf_last_gpu_ms",
]


class HipUnavailable(RuntimeError):
    """libheatflow_hip.so is not built or no HIP device is usable."""


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"heatflow_hip error {code}: {msg}")
        self.code = code


class NotConverged(HipError):
    pass


def build_library(force=False, verbose=False):
    """Compile csrc/heatflow_hip.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc", "heatflow_hip.hip")
    hdr = os.path.join(_HERE, "..", "include", "heatflow_hip.h")
    if not force and os.path.isfile(LIB_PATH):
        csrc = os.path.join(_HERE, "csrc")
        newest = max([os.path.getmtime(src), os.path.getmtime(hdr)] +
                     [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(".hpp")])
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "libheatflow_hip.so"]
    if force:
        cmd.insert(1, "-B")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libheatflow_hip.so failed:\n" + res.stderr)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen the library and declare the prototypes (no device is touched)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipUnavailable(f"{LIB_PATH} not found - run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C heatflow_amd/csrc`")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.hf_version.restype = C.c_char_p
    lib.hf_last_error.restype = C.c_char_p
    lib.hf_last_error.argtypes = [vp]
    protos = {
        "hf_create": [C.c_int, C.POINTER(vp)],
        "hf_destroy": [vp],
        "hf_set_mesh": [vp, i32, i32, pd, pi, pi],
        "hf_set_mesh_prebuilt": [vp, i32, i32, pd, pi, pi, vp, i64],
        "hf_pattern_export_size": [vp, C.POINTER(i64)],
        "hf_pattern_export": [vp, vp, i64],
        "hf_amg_export_size": [vp, C.POINTER(i64)],
        "hf_amg_export": [vp, vp, i64],
        "hf_amg_install": [vp, vp, i64],
        "hf_set_materials": [vp, i32, pi, pd, pd],
        "hf_update_kappa": [vp, i32, pi, pd],
        "hf_set_dirichlet": [vp, i32, pi],
        "hf_assemble": [vp, dbl, i32],
        "hf_set_precond": [vp, i32, i32],
        "hf_set_start_vector": [vp, i32],
        "hf_get_response_solves": [vp, C.POINTER(i64)],
        "hf_get_amg_info": [vp, pi, pi, i32, pd, pd],
        "hf_get_amg_fallbacks": [vp, C.POINTER(i64)],
        "hf_set_state": [vp, pd],
        "hf_get_state": [vp, pd],
        "hf_sample": [vp, i32, pi, pd],
        "hf_step": [vp, pd, dbl, dbl, i32, pi, pd],
        "hf_run": [vp, i32, pd, dbl, dbl, i32, i32, pi, pd, pi],
        "hf_batch_begin": [vp, i32, i32],
        "hf_batch_load_column": [vp, i32],
        "hf_batch_set_affine": [vp, i32, pi, pd],
        "hf_batch_set_state": [vp, i32, pd],
        "hf_batch_get_state": [vp, i32, pd],
        "hf_batch_run": [vp, i32, pd, dbl, dbl, i32, i32, pi, pd, pi],
        "hf_batch_run_flux": [vp, i32, pd, dbl, dbl, i32, i32, pi, pd, pi, i32, dbl, i32, i32, pi, pd, pi],
        "hf_batch_end": [vp],
        "hf_flux_setup": [vp],
        "hf_flux_project": [vp, dbl, i32, pd, pd, pi],
        "hf_flux_solve": [vp, i32, dbl, i32, pi],
        "hf_flux_sample": [vp, i32, pi, pd, pd],
        "hf_get_sizes": [vp, pi, pi, C.POINTER(i64), pi],
        "hf_get_csr": [vp, pi, pi, pd, pd],
        "hf_spmv": [vp, i32, pd, pd],
        "hf_time_kernel": [vp, i32, i32, pd],
        "hf_last_gpu_ms": [vp, pd],
        "hf_set_profile": [vp, i32],
        "hf_get_profile": [vp, pd, C.POINTER(i64)],
    }
    for name, args in protos.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def blob_address(blob):
    """(address, nbytes) of a blob handed to set_mesh(pattern=) / amg_install: a uint8 numpy array, an ``(address, nbytes)``
    tuple, or any object with ``address`` / ``nbytes`` attributes (a device-resident buffer kept alive by its owner, e.g. the
    tensor a sweep received over RCCL: :class:`heatflow_amd.parameter_sweep.DeviceBlob`)."""
    if isinstance(blob, tuple):
        return int(blob[0]), int(blob[1]), blob
    if hasattr(blob, "address") and hasattr(blob, "nbytes") and not isinstance(blob, np.ndarray):
        return int(blob.address), int(blob.nbytes), blob
    arr = np.ascontiguousarray(blob, dtype=np.uint8)
    return arr.ctypes.data, arr.nbytes, arr


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class HeatflowHIP:
    """One solver context = one HIP device + one stream (not thread-safe)."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.hf_create(int(device_id), C.byref(self._ctx))
        if rc != HF_OK:
            msg = self._lib.hf_last_error(self._ctx).decode() if self._ctx else "no usable HIP device"
            if self._ctx:
                self._lib.hf_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise HipUnavailable(f"hf_create(device {device_id}) failed ({rc}): {msg}")
        self.n = self.n_e = self.n_bc = 0
        self.nnz = 0
        self.batch_nv = 0

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.hf_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc == HF_OK:
            return
        msg = self._lib.hf_last_error(self._ctx).decode()
        if rc == HF_ERR_NOCONV:
            raise NotConverged(rc, msg)
        if rc == HF_ERR_ARG:
            raise ValueError(f"heatflow_hip: {msg}")
        raise HipError(rc, msg)

    # -- set-up --------------------------------------------------------------------------
    def set_mesh(self, coords, tris, tags, pattern=None):
        """``pattern``: the tables another context exported for the same mesh (:meth:`export_pattern`: a uint8
        array, or ``(address, nbytes)`` of a host or device buffer); they are installed instead of being rebuilt."""
        zr, tri, tag = _f64(coords), _i32(tris), _i32(tags)
        if zr.ndim != 2 or zr.shape[1] != 2 or tri.ndim != 2 or tri.shape[1] != 3 or tag.shape != (tri.shape[0],):
            raise ValueError("set_mesh: coords (n,2), tris (n_e,3), tags (n_e,) expected")
        if pattern is None:
            self._check(self._lib.hf_set_mesh(self._ctx, zr.shape[0], tri.shape[0], _pd(zr), _pi(tri), _pi(tag)))
        else:
            addr, nbytes, _keep = blob_address(pattern)
            self._check(self._lib.hf_set_mesh_prebuilt(self._ctx, zr.shape[0], tri.shape[0], _pd(zr), _pi(tri), _pi(tag),
                                                       C.c_void_p(addr), nbytes))
        self._refresh_sizes()

    def pattern_bytes(self):
        nb = C.c_int64()
        self._check(self._lib.hf_pattern_export_size(self._ctx, C.byref(nb)))
        return int(nb.value)

    def export_pattern(self, into=None):
        """The connectivity-derived tables of this context's mesh (CSR pattern, compressed column lists, row-gather
        lists) as one uint8 array, for ``set_mesh(..., pattern=)`` of other contexts.  ``into = (address, nbytes)``
        writes to that host or device buffer instead and returns None."""
        nb = self.pattern_bytes()
        if into is not None:
            if int(into[1]) != nb:
                raise ValueError(f"export_pattern: buffer of {into[1]} bytes, need {nb}")
            self._check(self._lib.hf_pattern_export(self._ctx, C.c_void_p(int(into[0])), nb))
            return None
        blob = np.empty(nb, dtype=np.uint8)
        self._check(self._lib.hf_pattern_export(self._ctx, C.c_void_p(blob.ctypes.data), nb))
        return blob

    def amg_export(self, into=None):
        """The multigrid hierarchy of this context (assembled with PC_AMG) as one uint8 array, for ``amg_install`` of other
        contexts on the same mesh.  ``into = (address, nbytes)`` writes to that host or device buffer and returns None."""
        nb = C.c_int64()
        self._check(self._lib.hf_amg_export_size(self._ctx, C.byref(nb)))
        if into is not None:
            if int(into[1]) != nb.value:
                raise ValueError(f"amg_export: buffer of {into[1]} bytes, need {nb.value}")
            self._check(self._lib.hf_amg_export(self._ctx, C.c_void_p(int(into[0])), nb.value))
            return None
        blob = np.empty(nb.value, dtype=np.uint8)
        self._check(self._lib.hf_amg_export(self._ctx, C.c_void_p(blob.ctypes.data), nb.value))
        return blob

    def amg_install(self, blob):
        """Install a hierarchy another context exported (after set_dirichlet and set_precond(PC_AMG, reuse=True), before
        assemble): nothing is built on the host; assemble() then decides from the blob's fingerprint whether this context's
        operator is the one the hierarchy was built from."""
        addr, nbytes, _keep = blob_address(blob)
        self._check(self._lib.hf_amg_install(self._ctx, C.c_void_p(addr), nbytes))

    def _refresh_sizes(self):
        n, ne, nbc, nnz = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        self._check(self._lib.hf_get_sizes(self._ctx, C.byref(n), C.byref(ne), C.byref(nnz), C.byref(nbc)))
        self.n, self.n_e, self.nnz, self.n_bc = n.value, ne.value, nnz.value, nbc.value

    def set_materials(self, tags, kappa, rho_c):
        t, k, c = _i32(tags), _f64(kappa), _f64(rho_c)
        if not (t.shape == k.shape == c.shape) or t.ndim != 1:
            raise ValueError("set_materials: three 1-D arrays of equal length expected")
        self._check(self._lib.hf_set_materials(self._ctx, len(t), _pi(t), _pd(k), _pd(c)))

    def update_kappa(self, tags, kappa):
        """Overwrite the conductivity of some cell tags and re-assemble (kappa sweeps)."""
        t, k = _i32(tags), _f64(kappa)
        self._check(self._lib.hf_update_kappa(self._ctx, len(t), _pi(t), _pd(k)))

    def set_dirichlet(self, dofs):
        d = _i32(dofs)
        self._check(self._lib.hf_set_dirichlet(self._ctx, len(d), _pi(d) if len(d) else None))
        self._refresh_sizes()

    def set_precond(self, kind=PC_JACOBI, reuse=False):
        """PC_JACOBI (0) or PC_AMG (1); call before assemble()."""
        self._check(self._lib.hf_set_precond(self._ctx, int(kind), 1 if reuse else 0))

    def set_start_vector(self, kind=3):
        """0: u^n, 1: 2u^n - u^{n-1}, 2: that + response to the boundary values' second difference, 3 (default):
        A-norm projection on the last solutions and the boundary responses."""
        self._check(self._lib.hf_set_start_vector(self._ctx, int(kind)))

    def response_solves(self):
        c = C.c_int64()
        self._check(self._lib.hf_get_response_solves(self._ctx, C.byref(c)))
        return int(c.value)

    def amg_info(self):
        nl, opc, secs = C.c_int32(), C.c_double(), C.c_double()
        rows = np.zeros(16, dtype=np.int32)
        self._check(self._lib.hf_get_amg_info(self._ctx, C.byref(nl), _pi(rows), 16, C.byref(opc), C.byref(secs)))
        fb = C.c_int64()
        self._check(self._lib.hf_get_amg_fallbacks(self._ctx, C.byref(fb)))
        return {"levels": nl.value, "rows": rows[:nl.value].tolist(), "op_complexity": opc.value, "setup_s": secs.value,
                "jacobi_fallbacks": fb.value}

    def assemble(self, dt, mode=ASM_LDS_ATOMIC):
        self._check(self._lib.hf_assemble(self._ctx, float(dt), int(mode)))

    # -- state ---------------------------------------------------------------------------
    def set_state(self, u):
        u = _f64(u)
        if u.shape != (self.n,):
            raise ValueError(f"set_state: expected {self.n} values")
        self._check(self._lib.hf_set_state(self._ctx, _pd(u)))

    def get_state(self):
        u = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.hf_get_state(self._ctx, _pd(u)))
        return u

    def sample(self, nodes):
        idx = _i32(nodes)
        out = np.empty(len(idx), dtype=np.float64)
        self._check(self._lib.hf_sample(self._ctx, len(idx), _pi(idx), _pd(out)))
        return out

    # -- time stepping -------------------------------------------------------------------
    def step(self, g_bc, rtol=1e-10, atol=0.0, max_it=20000):
        g = _f64(g_bc)
        if g.shape != (self.n_bc,):
            raise ValueError(f"step: expected {self.n_bc} boundary values")
        it, res = C.c_int32(), C.c_double()
        rc = self._lib.hf_step(self._ctx, _pd(g) if self.n_bc else None, rtol, atol, int(max_it), C.byref(it), C.byref(res))
        self.last_iters, self.last_resid = it.value, res.value
        self._check(rc)
        return it.value, res.value

    def run(self, g_all, rtol=1e-10, atol=0.0, max_it=20000, nodes=None):
        g = _f64(g_all)
        if g.ndim != 2 or g.shape[1] != self.n_bc:
            raise ValueError(f"run: g_all must be (n_steps, {self.n_bc})")
        nsteps = g.shape[0]
        idx = _i32(nodes) if nodes is not None and len(nodes) else None
        ns = 0 if idx is None else len(idx)
        samples = np.empty((nsteps, ns), dtype=np.float64)
        iters = np.zeros(nsteps, dtype=np.int32)
        rc = self._lib.hf_run(self._ctx, nsteps, _pd(g) if self.n_bc else None, rtol, atol, int(max_it), ns, _pi(idx),
                              _pd(samples) if ns else None, _pi(iters))
        self.last_run_iters = iters
        self._check(rc)
        return samples, iters

    # -- batched time loop: nv sweep points as the columns of one multi-vector PCG ------------------
    def batch_begin(self, nv, per_column_operator=False):
        """``per_column_operator``: BATCH_SHARED (False), BATCH_PER_COLUMN (True) or BATCH_AFFINE."""
        self.batch_nv = 0
        self._check(self._lib.hf_batch_begin(self._ctx, int(nv), int(per_column_operator)))
        self.batch_nv = int(nv)

    def batch_set_affine(self, tags, delta):
        """Operators A_j = A + delta[j] * dt K(unit conductivity on the cell tags ``tags``), A = the context's operator."""
        t, d = _i32(tags), _f64(delta)
        if d.shape != (self.batch_nv,):
            raise ValueError(f"batch_set_affine: {self.batch_nv} deltas expected")
        self._check(self._lib.hf_batch_set_affine(self._ctx, len(t), _pi(t), _pd(d)))

    def batch_load_column(self, j):
        """Copy the context's current (assembled, eliminated) operator into column j of the batch."""
        self._check(self._lib.hf_batch_load_column(self._ctx, int(j)))

    def batch_set_state(self, j, u):
        u = _f64(u)
        if u.shape != (self.n,):
            raise ValueError(f"batch_set_state: expected {self.n} values")
        self._check(self._lib.hf_batch_set_state(self._ctx, int(j), _pd(u)))

    def batch_get_state(self, j):
        u = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.hf_batch_get_state(self._ctx, int(j), _pd(u)))
        return u

    def batch_run(self, g_all, rtol=1e-10, atol=0.0, max_it=20000, nodes=None, flux_nodes=None, flux_components=2,
                  flux_rtol=None, flux_max_it=5000):
        """g_all: (n_steps, n_bc, nv).  Returns samples (n_steps, nv, n_s) and iters (n_steps, nv).
        With ``flux_nodes`` every step is followed by the read-flux projection of every column (flux_setup() first;
        ``flux_components``: 1 = d/dz, 2 = d/dr, 3 = both) and a third array is returned: the projected gradient at those
        nodes, (n_steps, n_comp, nv, len(flux_nodes)), z before r."""
        g = _f64(g_all)
        if g.ndim != 3 or g.shape[1] != self.n_bc or (self.batch_nv and g.shape[2] != self.batch_nv):
            raise ValueError(f"batch_run: g_all must be (n_steps, {self.n_bc}, {self.batch_nv or 'nv'})")
        nv = self.batch_nv or g.shape[2]        # no batch open on this side: the library reports it
        nsteps = g.shape[0]
        idx = _i32(nodes) if nodes is not None and len(nodes) else None
        ns = 0 if idx is None else len(idx)
        samples = np.empty((nsteps, nv, ns), dtype=np.float64)
        iters = np.zeros((nsteps, nv), dtype=np.int32)
        if flux_nodes is None:
            rc = self._lib.hf_batch_run(self._ctx, nsteps, _pd(g) if self.n_bc else None, rtol, atol, int(max_it), ns, _pi(idx),
                                        _pd(samples) if ns else None, _pi(iters))
            self.last_run_iters = iters
            self._check(rc)
            return samples, iters
        fidx = _i32(flux_nodes)
        comps = int(flux_components)
        ncomp = (comps & 1) + ((comps >> 1) & 1)
        flux = np.empty((nsteps, ncomp, nv, len(fidx)), dtype=np.float64)
        fit = np.zeros((nsteps, max(ncomp, 1)), dtype=np.int32)
        rc = self._lib.hf_batch_run_flux(self._ctx, nsteps, _pd(g) if self.n_bc else None, rtol, atol, int(max_it), ns, _pi(idx),
                                         _pd(samples) if ns else None, _pi(iters), comps, float(rtol if flux_rtol is None else flux_rtol),
                                         int(flux_max_it), len(fidx), _pi(fidx), _pd(flux), _pi(fit))
        self.last_run_iters, self.last_flux_iters = iters, fit
        self._check(rc)
        return samples, iters, flux

    def batch_end(self):
        self._check(self._lib.hf_batch_end(self._ctx))
        self.batch_nv = 0

    # -- read-flux projection (run_no_diamond) ----------------------------------------------
    def flux_setup(self):
        self._check(self._lib.hf_flux_setup(self._ctx))

    def flux_project(self, rtol=1e-10, max_it=5000, want_z=True, want_r=True):
        """(grad_z, grad_r) of the current state, L2-projected onto P1 with weight r."""
        gz = np.empty(self.n, dtype=np.float64) if want_z else None
        gr = np.empty(self.n, dtype=np.float64) if want_r else None
        it = np.zeros(2, dtype=np.int32)
        self._check(self._lib.hf_flux_project(self._ctx, rtol, int(max_it), _pd(gz), _pd(gr), _pi(it)))
        self.last_flux_iters = it
        return gz, gr

    def flux_solve(self, rtol=1e-10, max_it=5000, want_z=True, want_r=True):
        """Project the current state's gradient on the device only (no copy); returns the iteration counts."""
        it = np.zeros(2, dtype=np.int32)
        self._check(self._lib.hf_flux_solve(self._ctx, (1 if want_z else 0) | (2 if want_r else 0), rtol, int(max_it), _pi(it)))
        self.last_flux_iters = it
        return it

    def flux_sample(self, nodes, want_z=True, want_r=True):
        """(grad_z[nodes], grad_r[nodes]) of the last projection."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        gz = np.empty(len(nodes), dtype=np.float64) if want_z else None
        gr = np.empty(len(nodes), dtype=np.float64) if want_r else None
        self._check(self._lib.hf_flux_sample(self._ctx, len(nodes), _pi(nodes), _pd(gz), _pd(gr)))
        return gz, gr

    # -- inspection ----------------------------------------------------------------------
    def get_csr(self, values=True):
        rowptr = np.empty(self.n + 1, dtype=np.int32)
        colidx = np.empty(self.nnz, dtype=np.int32)
        A = np.empty(self.nnz, dtype=np.float64) if values else None
        M = np.empty(self.nnz, dtype=np.float64) if values else None
        self._check(self._lib.hf_get_csr(self._ctx, _pi(rowptr), _pi(colidx), _pd(A), _pd(M)))
        return rowptr, colidx, A, M

    def spmv(self, x, which=0):
        x = _f64(x)
        y = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.hf_spmv(self._ctx, int(which), _pd(x), _pd(y)))
        return y

    def time_kernel(self, which, reps=50):
        ms = C.c_double()
        self._check(self._lib.hf_time_kernel(self._ctx, int(which), int(reps), C.byref(ms)))
        return ms.value

    def set_profile(self, on=True):
        self._check(self._lib.hf_set_profile(self._ctx, 1 if on else 0))

    def get_profile(self):
        """(summed ms, launches) of the PCG SpMV launches bracketed since set_profile(True)."""
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._lib.hf_get_profile(self._ctx, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def last_gpu_ms(self):
        ms = C.c_double()
        self._check(self._lib.hf_last_gpu_ms(self._ctx, C.byref(ms)))
        return ms.value

    @staticmethod
    def version():
        return load_library().hf_version().decode()
