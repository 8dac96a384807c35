"""The C-ABI library loads on a CPU-only host, exports every symbol include/heatflow_hip.h
declares, and refuses to create a context without a HIP device (no CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "heatflow_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hf_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    syms = _declared_symbols()
    for must in ("hf_create", "hf_destroy", "hf_set_mesh", "hf_set_materials", "hf_set_dirichlet", "hf_assemble",
                 "hf_step", "hf_run", "hf_get_state", "hf_sample", "hf_get_csr", "hf_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from heatflow_amd import hip_backend

    lib = hip_backend.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in heatflow_hip.h but not exported"
    assert set(hip_backend.EXPORTS) == set(_declared_symbols())
    assert b"gfx950" in lib.hf_version()


def test_no_cpu_fallback_without_a_device():
    import ctypes

    from heatflow_amd import hip_backend

    lib = hip_backend.load_library()
    ctx = ctypes.c_void_p()
    rc = lib.hf_create(0, ctypes.byref(ctx))
    if rc == 0:  # running on a GPU box: the context must work, then be released
        assert lib.hf_destroy(ctx) == 0
        return
    assert rc < 0
    with pytest.raises(hip_backend.HipUnavailable):
        hip_backend.HeatflowHIP(0)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "heatflow_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{fn} imports the oracle"


def test_host_library_exports_every_declared_symbol_and_writes_the_same_msh(tmp_path):
    """libheatflow_host.so (include/heatflow_host.h): symbols, and the native MSH 2.2 writer against the
    numpy writer it replaces - byte-identical files, exact coordinate round trip."""
    import numpy as np

    from heatflow_amd import hostlib, mesh as M

    with open(os.path.join(ROOT, "include", "heatflow_host.h")) as f:
        declared = set(re.findall(r"^\s*(?:int|int64_t|void)\s+(hfh_\w+)\s*\(", f.read(), flags=re.M))
    lib = hostlib.load_library()
    assert lib is not None, "gcc is part of the image: the host library must build"
    assert declared == set(hostlib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)
    rng = np.random.default_rng(5)
    coords = rng.standard_normal((500, 2)) * 1e-5
    tris = rng.integers(0, 500, size=(900, 3)).astype(np.int32)
    tags = rng.integers(1, 10, size=900).astype(np.int32)
    names = {"a b": 3, "p_diam": 1}
    hostlib.write_msh22(str(tmp_path / "native.msh"), coords, tris, tags, names)
    try:
        hostlib._lib, hostlib._failed = None, True          # force the numpy path
        M.write_msh(str(tmp_path / "numpy.msh"), coords, tris, tags, names)
    finally:
        hostlib._failed = False
    assert (tmp_path / "native.msh").read_bytes() == (tmp_path / "numpy.msh").read_bytes()
    c2, t2, g2 = M.read_msh(str(tmp_path / "native.msh"))
    assert np.array_equal(c2, coords) and np.array_equal(t2, tris) and np.array_equal(g2, tags)
    with pytest.raises(OSError):
        hostlib.write_msh22(str(tmp_path / "no_such_dir" / "x.msh"), coords, tris, tags)
