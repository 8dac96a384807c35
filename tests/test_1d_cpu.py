"""BASELINE config C1 (cfgs/geballe_1d.yaml via run_no_diamond_1d.run_1d): CPU plumbing."""
import os

import numpy as np
import pytest

from conftest import HEATING_CSV, load_cfg
from heatflow_amd.geometry import build_stack, watcher_points
from heatflow_amd.driver import prepare_mesh
from oracle import heat_oracle as ho


def test_run_1d_matches_the_oracle_slab_solver(tmp_path):
    import run_no_diamond_1d as r1

    cfg = load_cfg("geballe_1d")
    assert int(cfg["timing"]["num_steps"]) == 50
    stack = build_stack(cfg)
    mesh_folder = str(tmp_path / "mesh")
    coords, tris, tags, tag_map = prepare_mesh(cfg, mesh_folder, True, stack)     # stock 2-D mesh
    with pytest.raises(FileNotFoundError):
        r1.run_1d(cfg, str(tmp_path / "nope"), use_radial_correction=False)
    with pytest.raises(NotImplementedError):
        r1.run_1d(cfg, mesh_folder)                                                # correction is f3
    wp = watcher_points(cfg)
    out = str(tmp_path / "out")
    res = r1.run_1d(cfg, mesh_folder, output_folder=out, watcher_points=wp, write_xdmf=False, suppress_print=True,
                    use_radial_correction=False)
    z = res["z"]
    assert 200 < len(z) < 1500 and np.all(np.diff(z) > 0)          # SURVEY: ~350-1000 nodes with gmsh; 291 here
    assert z[0] == pytest.approx(-4.182e-6) and z[-1] == pytest.approx(7.282e-6)
    # material of every interval = material box containing its midpoint
    mid = 0.5 * (z[1:] + z[:-1])
    for m in stack.materials:
        sel = (mid > m.boundaries[0]) & (mid < m.boundaries[1])
        assert (res["cell_tags"][sel] == tag_map[m.name]).all()
    # same numbers as the oracle's 1-D slab solver
    rc = np.array([{tag_map[m.name]: m.properties["rho_cv"] for m in stack.materials}[int(t)] for t in res["cell_tags"]])
    kp = np.array([{tag_map[m.name]: m.properties["k"] for m in stack.materials}[int(t)] for t in res["cell_tags"]])
    h_time, h_temp = ho.read_heating_csv(HEATING_CSV)
    dt = 7.5e-6 / 50
    heat_node = int(np.argmin(np.abs(z - (-0.982e-6))))
    assert abs(z[heat_node] + 0.982e-6) < 1e-12
    bc_nodes = [0, heat_node, len(z) - 1]
    ref = ho.solve_1d_slab(z, rc, kp, dt, np.full(len(z), 300.0), bc_nodes,
                           lambda t: np.array([300.0, ho.heating_amplitude(t, h_time, h_temp, 300.0), 300.0]), 50)
    assert np.abs(res["u"] - ref[-1]).max() < 1e-8
    assert ref[-1].max() > 400.0
    assert os.path.isfile(os.path.join(out, "watcher_points.csv")) and os.path.isfile(os.path.join(out, "used_config.yaml"))
    wn = int(np.argmin(np.abs(z - wp["oside"][0])))
    assert res["watchers"]["oside"][-1] == pytest.approx(ref[-1][wn], abs=1e-8)
