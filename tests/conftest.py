import os
import sys

import numpy as np
import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HEATING_CSV = os.path.join(ROOT, "experimental_data", "geballe_heat_data.csv")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cfg(name):
    with open(os.path.join(ROOT, "cfgs", f"{name}.yaml")) as f:
        return yaml.safe_load(f)


def build_case(name, scale):
    """cfg (mesh sizes scaled), stack, mesh for one of the shipped configs."""
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes
    from heatflow_amd.mesh import Mesh

    cfg = scale_mesh_sizes(load_cfg(name), scale)
    stack = build_stack(cfg)
    mesh = Mesh("mesh.msh", stack.bounds, stack.materials).build_mesh()
    return cfg, stack, mesh


@pytest.fixture(scope="session")
def case_with_diamond_small():
    return build_case("geballe_with_diamond", 8.0)


@pytest.fixture(scope="session")
def case_no_diamond_small():
    return build_case("geballe_no_diamond", 8.0)


@pytest.fixture(scope="session")
def hip():
    """The HIP backend module; GPU tests fail (not skip) if the library cannot be loaded."""
    from heatflow_amd import hip_backend

    hip_backend.load_library()
    return hip_backend
