"""Multigrid set-up algebra (heatflow_amd/csrc/amg_host.hpp) checked on the CPU: the header is plain C++,
so a small harness (tests/cpp/amg_host_check.cpp) is compiled with g++ and run - Galerkin products,
R = P^T, and the fused down / up legs of the intermediate levels against the explicit V(1,1) steps."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("nx,ny", [(96, 80), (50, 131)])
def test_amg_host_algebra(tmp_path, nx, ny):
    exe = str(tmp_path / "amg_host_check")
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "heatflow_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "amg_host_check.cpp"), "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe, str(nx), str(ny)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "OK worst" in run.stdout, run.stdout + run.stderr
