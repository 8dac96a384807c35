"""ctypes binding of libheatflow_host.so (include/heatflow_host.h): host-only native helpers of the mesh
layer.  Plain C built with gcc - no GPU involved; `available()` is False when it cannot be built, and the
callers keep a (slow) numpy path for that case, which is I/O plumbing, not the compute path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libheatflow_host.so")
EXPORTS = ["hfh_version", "hfh_write_msh22", "hfh_quadtree_levels", "hfh_quadtree_leaves", "hfh_mesh_build", "hfh_mesh_sizes",
           "hfh_mesh_fetch", "hfh_mesh_free"]

_lib = None
_failed = False


def build_library(force=False):
    src = os.path.join(_HERE, "csrc", "host", "heatflow_host.c")
    hdr = os.path.join(_HERE, "..", "include", "heatflow_host.h")
    if not force and os.path.isfile(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return LIB_PATH
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "libheatflow_host.so"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libheatflow_host.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def load_library():
    global _lib, _failed
    if _lib is not None:
        return _lib
    if _failed:
        return None
    try:
        lib = C.CDLL(build_library())
    except (OSError, RuntimeError):
        _failed = True
        return None
    i32 = C.c_int32
    lib.hfh_version.restype = C.c_int
    lib.hfh_write_msh22.restype = C.c_int
    lib.hfh_write_msh22.argtypes = [C.c_char_p, i32, i32, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(i32), i32,
                                    C.POINTER(C.c_char_p), C.POINTER(i32)]
    p8, p64 = C.POINTER(C.c_int8), C.POINTER(C.c_int64)
    lib.hfh_quadtree_levels.restype = C.c_int
    lib.hfh_quadtree_levels.argtypes = [i32, i32, i32, p8, p8, p8]
    lib.hfh_quadtree_leaves.restype = C.c_int64
    lib.hfh_quadtree_leaves.argtypes = [i32, i32, i32, p8, C.c_int64, p64, p64, p64]
    pd = C.POINTER(C.c_double)
    lib.hfh_mesh_build.restype = C.c_int
    lib.hfh_mesh_build.argtypes = [C.c_int64, p64, p64, p64, i32, i32, p8, i32, i32, pd, pd, C.POINTER(C.c_void_p)]
    lib.hfh_mesh_sizes.restype = C.c_int
    lib.hfh_mesh_sizes.argtypes = [C.c_void_p, p64, p64, p64]
    lib.hfh_mesh_fetch.restype = C.c_int
    lib.hfh_mesh_fetch.argtypes = [C.c_void_p, pd, p64, C.POINTER(i32), C.POINTER(i32)]
    lib.hfh_mesh_free.restype = None
    lib.hfh_mesh_free.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def available():
    return load_library() is not None


def write_msh22(filename, coords, tris, tags, names=None):
    """MSH 2.2 ASCII through the native writer (see heatflow_host.h)."""
    lib = load_library()
    if lib is None:
        raise RuntimeError("libheatflow_host.so is not available")
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    tris = np.ascontiguousarray(tris, dtype=np.int32)
    tags = np.ascontiguousarray(tags, dtype=np.int32)
    items = sorted((names or {}).items(), key=lambda kv: kv[1])
    nm = (C.c_char_p * max(len(items), 1))(*[k.encode() for k, _ in items])
    nt = np.array([t for _, t in items] or [0], dtype=np.int32)
    rc = lib.hfh_write_msh22(os.fsencode(filename), len(coords), len(tris), coords.ctypes.data_as(C.POINTER(C.c_double)),
                             tris.ctypes.data_as(C.POINTER(C.c_int32)), tags.ctypes.data_as(C.POINTER(C.c_int32)),
                             len(items), nm, nt.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise OSError(-rc, os.strerror(-rc), filename)


def quadtree_levels(mat, allowed, lmax):
    """Balanced quadtree level map (int8, -1 outside) from the material / allowed-level maps (heatflow_host.h)."""
    lib = load_library()
    if lib is None:
        raise RuntimeError("libheatflow_host.so is not available")
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    allowed = np.ascontiguousarray(allowed, dtype=np.int8)
    level = np.empty(mat.shape, dtype=np.int8)
    p8 = C.POINTER(C.c_int8)
    rc = lib.hfh_quadtree_levels(mat.shape[0], mat.shape[1], int(lmax), mat.ctypes.data_as(p8), allowed.ctypes.data_as(p8),
                                 level.ctypes.data_as(p8))
    if rc != 0:
        raise RuntimeError(f"hfh_quadtree_levels failed ({os.strerror(-rc)})")
    return level


def quadtree_leaves(level, lmax):
    """(i0, j0, lev) int64 arrays of the leaves of a level map, by level and row-major within a level."""
    lib = load_library()
    if lib is None:
        raise RuntimeError("libheatflow_host.so is not available")
    level = np.ascontiguousarray(level, dtype=np.int8)
    p8, p64 = C.POINTER(C.c_int8), C.POINTER(C.c_int64)
    n = lib.hfh_quadtree_leaves(level.shape[0], level.shape[1], int(lmax), level.ctypes.data_as(p8), 0, None, None, None)
    if n < 0:
        raise RuntimeError(f"hfh_quadtree_leaves failed ({os.strerror(-n)})")
    i0, j0, lev = (np.empty(n, dtype=np.int64) for _ in range(3))
    lib.hfh_quadtree_leaves(level.shape[0], level.shape[1], int(lmax), level.ctypes.data_as(p8), n, i0.ctypes.data_as(p64),
                            j0.ctypes.data_as(p64), lev.ctypes.data_as(p64))
    return i0, j0, lev


MESH_LATTICE_MAX = 65535      # Morton codes of the node / leaf order take the low 16 bits of a lattice index


def mesh_from_leaves(i0, j0, lev, mat, zc, rc):
    """Nodes and triangles from the quadtree's leaves (heatflow_host.h: hfh_mesh_build): returns
    (coords (n, 2) f64, node_ij (n, 2) i64, tris (nt, 3) i32 counter-clockwise, tags (nt,) i32 = material index + 1, n_fan),
    nodes and triangles in Morton order - bit for bit what the numpy statement in Mesh.build_mesh gives."""
    lib = load_library()
    if lib is None:
        raise RuntimeError("libheatflow_host.so is not available")
    i0, j0, lev = (np.ascontiguousarray(a, dtype=np.int64) for a in (i0, j0, lev))
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    zc, rc = np.ascontiguousarray(zc, dtype=np.float64), np.ascontiguousarray(rc, dtype=np.float64)
    p8, p64, pd, p32 = C.POINTER(C.c_int8), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    h = C.c_void_p()
    rc_ = lib.hfh_mesh_build(len(i0), i0.ctypes.data_as(p64), j0.ctypes.data_as(p64), lev.ctypes.data_as(p64), mat.shape[0],
                             mat.shape[1], mat.ctypes.data_as(p8), len(zc) - 1, len(rc) - 1, zc.ctypes.data_as(pd),
                             rc.ctypes.data_as(pd), C.byref(h))
    if rc_ != 0:
        raise RuntimeError(f"hfh_mesh_build failed ({os.strerror(-rc_)})")
    try:
        nn, nt, nf = C.c_int64(), C.c_int64(), C.c_int64()
        lib.hfh_mesh_sizes(h, C.byref(nn), C.byref(nt), C.byref(nf))
        coords = np.empty((nn.value, 2), dtype=np.float64)
        node_ij = np.empty((nn.value, 2), dtype=np.int64)
        tris = np.empty((nt.value, 3), dtype=np.int32)
        tags = np.empty(nt.value, dtype=np.int32)
        lib.hfh_mesh_fetch(h, coords.ctypes.data_as(pd), node_ij.ctypes.data_as(p64), tris.ctypes.data_as(p32), tags.ctypes.data_as(p32))
    finally:
        lib.hfh_mesh_free(h)
    return coords, node_ij, tris, tags, int(nf.value)
