"""Host logic of the entry points and of the sweep sharding, on CPU.  The solver session is
given an oracle-backed backend (tests/oracle_backend.py) through the injection points; the
product default (HeatflowHIP) is exercised by the GPU tests."""
import copy
import csv
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import yaml

from conftest import ROOT, load_cfg
from heatflow_amd import parameter_sweep as ps
from heatflow_amd.geometry import scale_mesh_sizes, watcher_points
from oracle_backend import OracleBackend


def _cfg(name="geballe_with_diamond", scale=16.0, steps=8):
    cfg = scale_mesh_sizes(load_cfg(name), scale)
    cfg["timing"]["num_steps"] = steps
    cfg["timing"]["t_final"] = 7.5e-8 * steps if name == "geballe_with_diamond" else 1.875e-7 * steps
    return cfg


@pytest.mark.parametrize("module,name", [("run_with_diamond", "geballe_with_diamond"), ("run_no_diamond", "geballe_no_diamond")])
def test_run_simulation_outputs_and_mesh_cache(tmp_path, module, name):
    import importlib

    run = importlib.import_module(module)            # the root-level drop-in module names
    cfg = _cfg(name)
    mesh_folder, out = str(tmp_path / "mesh"), str(tmp_path / "out")
    with pytest.raises(FileNotFoundError, match="mesh.msh, mesh_cfg.yaml"):
        run.run_simulation(cfg, mesh_folder, rebuild_mesh=False, backend=OracleBackend())
    wp = watcher_points(cfg)
    res = run.run_simulation(cfg, mesh_folder, rebuild_mesh=True, output_folder=out, watcher_points=wp,
                             write_xdmf=False, suppress_print=True, backend=OracleBackend())
    assert sorted(os.listdir(mesh_folder)) == ["mesh.msh", "mesh.npz", "mesh_cfg.yaml"]
    has_flux = os.path.isfile(os.path.join(out, "radial_gradient.csv"))
    assert has_flux == (module == "run_no_diamond")                  # read-flux CSVs only from run_no_diamond
    if has_flux:
        with open(os.path.join(out, "radial_gradient.csv")) as f:
            g = list(csv.reader(f))
        with open(os.path.join(out, "radial_gradient_raw.csv")) as f:
            graw = list(csv.reader(f))
        assert g[0][0] == "time" and len(g) == 9 and len(graw) == 9
        zc = np.array(g[0][1:], dtype=float)
        assert np.all(np.diff(zc) > 0) and np.allclose(np.diff(zc) / 0.2e-6, np.round(np.diff(zc) / 0.2e-6))
        zr = np.array(graw[0][1:], dtype=float)
        assert np.all(np.diff(zr) > 0) and zr[0] == pytest.approx(-4.182e-6)
    with open(os.path.join(mesh_folder, "mesh_cfg.yaml")) as f:
        mcfg = yaml.safe_load(f)
    assert mcfg["material_tags"]["p_ins"] >= 1 and "mats" in mcfg
    with open(os.path.join(out, "used_config.yaml")) as f:
        assert yaml.safe_load(f)["timing"]["num_steps"] == 8
    with open(os.path.join(out, "watcher_points.csv")) as f:
        rows = list(csv.reader(f))
    assert rows[0] == ["time", "pside", "oside"] and len(rows) == 9
    dt = float(cfg["timing"]["t_final"]) / 8
    assert float(rows[1][0]) == dt and float(rows[8][0]) == 8 * dt
    assert abs(float(rows[8][1]) - 300.0) > 0.1   # heated (a coarse consistent-mass mesh may undershoot)
    # second run reuses the cache, a list-of-dicts watcher spec works too, XDMF stand-in is written
    res2 = run.run_simulation(cfg, mesh_folder, output_folder=out, write_xdmf=True, suppress_print=True,
                              watcher_points=[{"name": "a", "coords": wp["pside"]}], backend=OracleBackend())
    assert np.allclose(res2["watchers"]["a"], res["watchers"]["pside"], rtol=0, atol=1e-9)
    meta = json.load(open(os.path.join(out, "output_fields.json")))
    assert len(meta["times"]) == 9 and os.path.getsize(os.path.join(out, "output_fields.f64")) == 9 * 8 * meta["n"]
    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(out, "output.xdmf")).getroot()            # well-formed XDMF 3, binary heavy data
    grids = root.findall("./Domain/Grid")
    assert grids[0].find("Topology").get("TopologyType") == "Triangle" and grids[1].get("CollectionType") == "Temporal"
    steps = grids[1].findall("Grid")
    assert len(steps) == 9 and float(steps[0].find("Time").get("Value")) == 0.0
    last = steps[-1].find("Attribute/DataItem")
    assert last.get("Format") == "Binary" and int(last.get("Seek")) == 8 * 8 * meta["n"]
    fld = np.fromfile(os.path.join(out, "output_fields.f64"), dtype="<f8").reshape(9, meta["n"])
    assert (fld[0] == 300.0).all() and np.abs(fld[-1] - 300.0).max() > 0.1
    with pytest.raises(ValueError, match="watcher_points must be a dict or list of dicts"):
        run.run_simulation(cfg, mesh_folder, watcher_points=3, backend=OracleBackend())


def test_default_backend_fails_loudly_without_gpu(tmp_path):
    """No silent CPU path: without a device the product entry point raises."""
    from heatflow_amd import hip_backend
    import ctypes
    lib = hip_backend.load_library()
    ctx = ctypes.c_void_p()
    if lib.hf_create(0, ctypes.byref(ctx)) == 0:
        lib.hf_destroy(ctx)
        pytest.skip("a GPU is present")
    import run_with_diamond as run
    with pytest.raises(hip_backend.HipUnavailable):
        run.run_simulation(_cfg(), str(tmp_path / "m"), rebuild_mesh=True, write_xdmf=False, suppress_print=True)


def test_grid_helpers_match_reference_conventions():
    combos, f, k, w = ps.create_parameter_grid((5e-6, 2e-5), (2.0, 8.0), (1.5e-6, 2.5e-6), (3, 2, 2))
    assert len(combos) == 12 and np.allclose(f, np.logspace(np.log10(5e-6), np.log10(2e-5), 3))
    assert [c["width"] for c in combos[:6]] == [w[0]] * 6            # grouped by width
    base = load_cfg("geballe_no_diamond")
    snap = copy.deepcopy(base)
    c = ps.modify_config_for_parameters(base, 1e-5, 4.0, 2e-6)
    assert c["heating"]["fwhm"] == 1e-5 and c["mats"]["p_sample"]["k"] == 4.0 and c["mats"]["p_sample"]["z"] == 2e-6
    assert base == snap                                               # base untouched (deep copy)
    assert ps.get_mesh_folder_for_width("meshes", 1.84e-6) == os.path.join("meshes", "width_1.840e-6")
    assert ps.run_name_for(1.32e-5, 3.8, 1.84e-6) == "fwhm_1.32e-5_k_3.80_width_1.84e-6"
    ks = ps.get_k_values()
    assert len(ks) == 51 and ks[0] == 3.3 and ks[-1] == 4.3
    assert len(ps.get_k_values(count=64)) == 64
    assert ps.shard(list("abcdefgh"), 1, 3) == [(1, "b"), (4, "e"), (7, "h")]


def _session_factory(coords, tris, tags, tag_map, pattern=None):
    from heatflow_amd.driver import SimulationSession
    return SimulationSession(coords, tris, tags, tag_map, backend=OracleBackend(), pattern=pattern)


def _sharing_session_factory(coords, tris, tags, tag_map, pattern=None, hierarchy=None):
    """As above, and takes the sweep's shared multigrid hierarchy (parameter_sweep.shared_hierarchy)."""
    from heatflow_amd.driver import SimulationSession
    return SimulationSession(coords, tris, tags, tag_map, backend=OracleBackend(), pattern=pattern, hierarchy=hierarchy)


def test_sweep_single_process_writes_artefacts_and_failed_rows(tmp_path):
    cfg = _cfg("geballe_no_diamond", 16.0, 6)
    cfg_path = str(tmp_path / "base.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    out = str(tmp_path / "sweep")
    ok, failed = ps.run_parameter_sweep(cfg_path, out, (8e-6, 2e-5), (3.0, 5.0), (1.84e-6, 1.84e-6), (2, 2, 1),
                                        base_mesh_folder=str(tmp_path / "meshes"), session_factory=_session_factory)
    assert len(ok) == 4 and not failed
    meta = json.load(open(os.path.join(out, "sweep_metadata.json")))
    assert meta["total_runs"] == 4 and len(meta["k_values"]) == 2
    rows = list(csv.DictReader(open(os.path.join(out, "successful_runs.csv"))))
    assert [int(r["run_id"]) for r in rows] == [1, 2, 3, 4] and all(r["status"] == "success" for r in rows)
    for r in rows:
        assert os.path.isfile(os.path.join(r["output_dir"], "watcher_points.csv"))
    # a failing point becomes a row in failed_runs.csv, the others still succeed
    bad = copy.deepcopy(cfg)
    bad["heating"]["file"] = "does/not/exist.csv"
    with open(cfg_path, "w") as f:
        yaml.safe_dump(bad, f)
    ok, failed = ps.run_parameter_sweep(cfg_path, out + "2", (8e-6, 2e-5), (3.0, 5.0), (1.84e-6, 1.84e-6), (1, 2, 1),
                                        base_mesh_folder=str(tmp_path / "meshes"), session_factory=_session_factory)
    assert not ok and len(failed) == 2 and "exist.csv" in failed[0]["error"]
    assert os.path.isfile(os.path.join(out + "2", "failed_runs.csv"))


def test_session_keeps_mesh_resident_across_kappa_points(tmp_path):
    """Within a width group only the coefficient tables / A are redone: one set_mesh, k assembles."""
    from heatflow_amd.driver import SimulationSession, prepare_mesh
    from heatflow_amd.geometry import build_stack

    cfg = _cfg("geballe_with_diamond", 16.0, 5)
    stack = build_stack(cfg)
    coords, tris, tags, tag_map = prepare_mesh(cfg, str(tmp_path / "m"), True, stack)
    be = OracleBackend()
    sess = SimulationSession(coords, tris, tags, tag_map, backend=be)
    outs = []
    for k in (3.3, 3.8, 4.3):
        c = copy.deepcopy(cfg)
        c["mats"]["p_sample"]["k"] = k
        outs.append(sess.run(c, build_stack(c), watcher_points(c))["watchers"]["oside"][-1])
    assert be.set_mesh_calls == 1 and be.assemble_calls == 3
    assert outs[0] != outs[2]


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from heatflow_amd import parameter_sweep as ps
    from test_driver_sweep_cpu import _session_factory
    dist.init_process_group("gloo")
    ok, failed = ps.run_parameter_sweep({cfg!r}, {out!r}, (8e-6, 2e-5), (3.0, 5.0), (1.84e-6, 2.2e-6), (2, 2, 2),
                                        base_mesh_folder={meshes!r}, session_factory=_session_factory)
    if dist.get_rank() == 0:
        json.dump({{"ok": ok, "failed": failed}}, open({res!r}, "w"))
    dist.destroy_process_group()
""")


def test_sweep_world_size_2_gloo_shards_points_and_broadcasts_the_mesh(tmp_path):
    """Two ranks (gloo, CPU): rank 0 meshes each width group and broadcasts it, points go
    i -> i mod 2, rows are gathered on rank 0; results equal the single-process sweep."""
    cfg = _cfg("geballe_no_diamond", 16.0, 5)
    cfg_path = str(tmp_path / "base.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    out2, res2 = str(tmp_path / "sweep2"), str(tmp_path / "res2.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, cfg=cfg_path, out=out2, meshes=str(tmp_path / "meshes2"), res=res2))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    got = json.load(open(res2))
    assert len(got["ok"]) == 8 and not got["failed"]
    assert sorted(r["rank"] for r in got["ok"]) == [0, 0, 0, 0, 1, 1, 1, 1]
    assert [r["run_id"] for r in got["ok"]] == list(range(1, 9))
    out1 = str(tmp_path / "sweep1")
    ok1, _ = ps.run_parameter_sweep(cfg_path, out1, (8e-6, 2e-5), (3.0, 5.0), (1.84e-6, 2.2e-6), (2, 2, 2),
                                    base_mesh_folder=str(tmp_path / "meshes1"), session_factory=_session_factory)
    for a, b in zip(ok1, got["ok"]):
        assert a["run_name"] == b["run_name"]
        wa = np.genfromtxt(os.path.join(a["output_dir"], "watcher_points.csv"), delimiter=",", names=True)
        wb = np.genfromtxt(os.path.join(b["output_dir"], "watcher_points.csv"), delimiter=",", names=True)
        assert np.array_equal(wa["oside"], wb["oside"]) and np.array_equal(wa["pside"], wb["pside"])


KAPPA_WORKER = textwrap.dedent("""
    import os, sys, json, yaml
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from heatflow_amd import parameter_sweep as ps
    from test_driver_sweep_cpu import _sharing_session_factory
    from oracle_backend import fake_pattern_blob
    dist.init_process_group("gloo")
    cfg = yaml.safe_load(open({cfg!r}))
    timing = {{}}
    made, sessions = [], []
    def factory(*a, **kw):
        made.append(kw.get("pattern") is not None)
        sessions.append(_sharing_session_factory(*a, **kw))
        return sessions[-1]
    rows = ps.run_kappa_sweep(cfg, {mesh!r}, [3.3, 3.6, 3.9, 4.2], {out!r}, session_factory=factory,
                              concurrent=2, warmup_steps=2, timing=timing, pattern_builder=fake_pattern_blob)
    timing["sessions_with_pattern"] = sum(made)
    timing["hierarchy_exports"] = [getattr(s.backend, "amg_export_calls", 0) for s in sessions]
    timing["hierarchy_installs"] = [getattr(s.backend, "amg_install_calls", 0) for s in sessions]
    gathered = [None, None]
    dist.all_gather_object(gathered, timing)
    timing = {{"rank0": gathered[0], "rank1": gathered[1], **gathered[0]}}
    if dist.get_rank() == 0:
        json.dump({{"rows": rows, "timing": timing}}, open({res!r}, "w"))
    dist.destroy_process_group()
""")


def test_kappa_sweep_world_size_2_uses_rank0s_tag_map_for_a_cached_mesh(tmp_path):
    """A cached mesh.msh + mesh_cfg.yaml whose material tags are NOT the list positions (as a gmsh /
    reference-written mesh has them: physical-group ids): rank 0 reads the tag map from mesh_cfg.yaml and
    broadcasts it with the arrays; rank 1 must not guess {name: k+1}.  Two ranks, two points in flight
    each, the measurement hooks on; the watcher curves equal a single-process run on the original tags."""
    from heatflow_amd.driver import prepare_mesh
    from heatflow_amd.geometry import build_stack
    from heatflow_amd.mesh import write_msh41

    cfg = _cfg("geballe_with_diamond", 16.0, 10)
    cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
    stack = build_stack(cfg)
    m1 = str(tmp_path / "mesh1")
    coords, tris, tags, tag_map = prepare_mesh(cfg, m1, True, stack)
    rows1 = ps.run_kappa_sweep(cfg, m1, [3.3, 3.6, 3.9, 4.2], str(tmp_path / "out1"), session_factory=_session_factory)
    assert [r["status"] for r in rows1] == ["success"] * 4
    # the same mesh with permuted tags, written as MSH 4.1 without an npz sidecar (forces the reader)
    perm = {t: 20 - t for t in tag_map.values()}
    m2 = str(tmp_path / "mesh2")
    os.makedirs(m2)
    write_msh41(os.path.join(m2, "mesh.msh"), coords, tris, np.array([perm[t] for t in tags]),
                {nm: perm[t] for nm, t in tag_map.items()})
    mcfg = copy.deepcopy(cfg)
    mcfg["material_tags"] = {nm: perm[t] for nm, t in tag_map.items()}
    with open(os.path.join(m2, "mesh_cfg.yaml"), "w") as f:
        yaml.safe_dump(mcfg, f)
    cfg_path, res = str(tmp_path / "cfg.yaml"), str(tmp_path / "res.json")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    script = tmp_path / "kworker.py"
    script.write_text(KAPPA_WORKER.format(root=ROOT, cfg=cfg_path, mesh=m2, out=str(tmp_path / "out2"), res=res))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    got = json.load(open(res))
    assert [r["status"] for r in got["rows"]] == ["success"] * 4, got["rows"]
    assert [r["rank"] for r in got["rows"]] == [0, 1, 0, 1]
    assert got["timing"]["sessions"] == 2 and got["timing"]["points_here"] == 2 and got["timing"]["warmup_s"] > 0
    # the connectivity tables: built by rank 0 only, broadcast, installed by every session of both ranks
    for r in ("rank0", "rank1"):
        assert got["timing"][r]["sessions_with_pattern"] == 2 and got["timing"][r]["pattern_bytes"] == 28
    # the multigrid hierarchy: built and exported by rank 0's first session only, broadcast, installed by its second session and
    # by both sessions of rank 1 - no other session runs the host set-up
    assert got["timing"]["rank0"]["hierarchy_exports"] == [1, 0] and got["timing"]["rank0"]["hierarchy_installs"] == [0, 1]
    assert got["timing"]["rank1"]["hierarchy_exports"] == [0, 0] and got["timing"]["rank1"]["hierarchy_installs"] == [1, 1]
    assert got["timing"]["rank1"]["hierarchy_bytes"] == 16 and got["timing"]["rank1"]["mesh_received_to_ready_s"] > 0
    for r in got["rows"]:
        a = np.genfromtxt(os.path.join(str(tmp_path / "out1"), f"{r['k']:.2f}", "watcher_points.csv"), delimiter=",", names=True)
        b = np.genfromtxt(os.path.join(str(tmp_path / "out2"), f"{r['k']:.2f}", "watcher_points.csv"), delimiter=",", names=True)
        # same physics on a renumbered mesh (the 4.1 file is re-ordered on load): direct solves agree to rounding.
        # (Only the o-side watcher is compared: the p-side point lies midway between two nodes of this coarse
        # mesh, and which of the two is "nearest" depends on the node numbering.)
        assert np.abs(a["oside"] - b["oside"]).max() < 1e-8
        assert np.abs(a["oside"] - 300.0).max() > 1e-6          # the heating has reached the watcher


def test_session_is_not_reused_for_a_different_dirichlet_set_with_the_same_dof_sum(tmp_path):
    """The resident problem is keyed on the Dirichlet DOF sets themselves: two sets with equal sums
    (what the key used to fingerprint) must not share a HeatProblem."""
    import heatflow_amd.driver as drv
    from heatflow_amd.driver import SimulationSession, prepare_mesh
    from heatflow_amd.geometry import build_stack

    cfg = _cfg("geballe_with_diamond", 16.0, 3)
    stack = build_stack(cfg)
    coords, tris, tags, tag_map = prepare_mesh(cfg, str(tmp_path / "m"), True, stack)
    be = OracleBackend()
    sess = SimulationSession(coords, tris, tags, tag_map, backend=be)
    sess.run(cfg, stack, watcher_points(cfg))
    key1, dofs1 = sess._key, sess.problem.bc_dofs.copy()
    real = drv.RowDirichletBC

    class SameSum(real):                        # heated line: two DOFs swapped for two others with the same sum
        def __init__(self, V, location, **kw):
            super().__init__(V, location, **kw)
            if location == "x":
                d = self.row_dofs.astype(np.int64)
                have, total = set(d.tolist()), int(d[0] + d[1])
                a = next(a for a in range(len(coords)) if a not in have and total - a not in have
                         and 0 <= total - a < len(coords) and a != total - a)
                d[0], d[1] = a, total - a
                self.row_dofs = np.sort(d).astype(np.int32)
                self.dof_coords = self.V.coords[self.row_dofs]

    drv.RowDirichletBC = SameSum
    try:
        sess.run(cfg, stack, watcher_points(cfg))
    finally:
        drv.RowDirichletBC = real
    assert sess._key != key1 and sess._key[:2] == key1[:2]
    assert int(dofs1.sum()) == int(sess.problem.bc_dofs.sum()) and not np.array_equal(dofs1, sess.problem.bc_dofs)
    assert be.set_mesh_calls == 2               # a new HeatProblem was built for the new Dirichlet set


def test_batched_sweeps_give_the_same_rows_and_files_as_point_by_point(tmp_path):
    """batch = 8 in both sweep drivers: points are grouped 8 / 4 / 2 / 1, advance together (here through the oracle
    stand-in of the batched loop) and leave the same artefacts and numbers as the unbatched sweep."""
    assert [len(g) for g in ps.batch_groups(list(range(15)), 8)] == [8, 4, 2, 1]
    assert [len(g) for g in ps.batch_groups(list(range(7)), 4)] == [4, 2, 1]
    assert [len(g) for g in ps.batch_groups(list(range(3)), 1)] == [1, 1, 1]
    from heatflow_amd.driver import prepare_mesh
    from heatflow_amd.geometry import build_stack

    cfg = _cfg("geballe_with_diamond", 16.0, 10)
    cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
    mesh = str(tmp_path / "mesh")
    prepare_mesh(cfg, mesh, True, build_stack(cfg))
    ks = [3.3, 3.45, 3.6, 3.75, 3.9, 4.05, 4.2]
    made = []

    def factory(*a, **kw):
        made.append(_session_factory(*a, **kw))
        return made[-1]

    timing = {}
    rows_b = ps.run_kappa_sweep(cfg, mesh, ks, str(tmp_path / "outb"), session_factory=factory, batch=4, timing=timing)
    rows_1 = ps.run_kappa_sweep(cfg, mesh, ks, str(tmp_path / "out1"), session_factory=_session_factory)
    assert timing["batches"] == [4, 2, 1] and made[0].backend.batch_begin_calls == 2
    assert [r["status"] for r in rows_b] == ["success"] * 7 and [r["k"] for r in rows_b] == ks
    assert [r.get("batch") for r in rows_b] == [4, 4, 4, 4, 2, 2, None]
    for k in ks:
        a = np.genfromtxt(os.path.join(str(tmp_path / "outb"), f"{k:.2f}", "watcher_points.csv"), delimiter=",", names=True)
        b = np.genfromtxt(os.path.join(str(tmp_path / "out1"), f"{k:.2f}", "watcher_points.csv"), delimiter=",", names=True)
        assert np.abs(a["oside"] - b["oside"]).max() < 1e-9 and np.abs(a["pside"] - b["pside"]).max() < 1e-9
        assert os.path.isfile(os.path.join(str(tmp_path / "outb"), f"{k:.2f}", "used_config.yaml"))
    # fwhm x k grid of parameter_sweep on the with-diamond stack: 3 x 2 points, batched 4 + 2
    cfg_path = str(tmp_path / "base.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    args = (cfg_path, None, (8e-6, 2e-5), (3.0, 5.0), (1.84e-6, 1.84e-6), (3, 2, 1))
    ok_b, failed_b = ps.run_parameter_sweep(args[0], str(tmp_path / "gb"), *args[2:], base_mesh_folder=str(tmp_path / "gm"),
                                            session_factory=_session_factory, batch=8)
    ok_1, failed_1 = ps.run_parameter_sweep(args[0], str(tmp_path / "g1"), *args[2:], base_mesh_folder=str(tmp_path / "gm"),
                                            session_factory=_session_factory)
    assert len(ok_b) == 6 and not failed_b and not failed_1 and [r.get("batch") for r in ok_b] == [4, 4, 4, 4, 2, 2]
    for a, b in zip(ok_b, ok_1):
        assert a["run_name"] == b["run_name"] and a["run_id"] == b["run_id"]
        wa = np.genfromtxt(os.path.join(a["output_dir"], "watcher_points.csv"), delimiter=",", names=True)
        wb = np.genfromtxt(os.path.join(b["output_dir"], "watcher_points.csv"), delimiter=",", names=True)
        assert np.abs(wa["oside"] - wb["oside"]).max() < 1e-9
    # the reference's production sweep runs run_no_diamond (parameter_sweep.py:43): batched too, read-flux CSVs per point
    cfgn = _cfg("geballe_no_diamond", 16.0, 6)
    cfgn["heating"]["file"] = os.path.join(ROOT, cfgn["heating"]["file"])
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfgn, f)
    made.clear()
    ok_b, failed_b = ps.run_parameter_sweep(cfg_path, str(tmp_path / "nb"), *args[2:], base_mesh_folder=str(tmp_path / "nm"),
                                            session_factory=factory, batch=8)
    ok_1, failed_1 = ps.run_parameter_sweep(cfg_path, str(tmp_path / "n1"), *args[2:], base_mesh_folder=str(tmp_path / "nm"),
                                            session_factory=_session_factory)
    assert len(ok_b) == 6 and not failed_b and not failed_1 and [r.get("batch") for r in ok_b] == [4, 4, 4, 4, 2, 2]
    assert made[0].backend.batch_flux_calls == 2
    for a, b in zip(ok_b, ok_1):
        for name in ("watcher_points.csv", "radial_gradient.csv", "radial_gradient_raw.csv"):
            fa = np.genfromtxt(os.path.join(a["output_dir"], name), delimiter=",", skip_header=1)
            fb = np.genfromtxt(os.path.join(b["output_dir"], name), delimiter=",", skip_header=1)
            assert fa.shape == fb.shape and np.allclose(fa, fb, rtol=1e-9, atol=1e-6), name
        with open(os.path.join(a["output_dir"], "radial_gradient_raw.csv")) as f1, open(os.path.join(b["output_dir"], "radial_gradient_raw.csv")) as f2:
            assert f1.readline() == f2.readline()            # same header: time + z of the axis nodes


def test_a_failed_batch_is_recorded_in_the_rows_and_the_points_are_rerun_one_by_one(tmp_path, capsys):
    """SURVEY 5 (failure detection; reference parameter_sweep.py:154-192, :516-518): a batched time loop that does not
    converge as a whole is re-run point by point, and every such row says so in `batch_error` (also in the CSV and on
    stderr) - a batch that always fails must not look like a merely slow sweep.  A failure that is not a solver outcome
    (here: a HIP error) is not retried: the points become failed rows."""
    import csv
    from heatflow_amd.driver import SimulationSession, prepare_mesh
    from heatflow_amd.geometry import build_stack
    from heatflow_amd.hip_backend import HipError, NotConverged

    class FailingBatch(OracleBackend):
        error = None
        calls = 0

        def batch_run(self, *a, **kw):
            FailingBatch.calls += 1
            raise FailingBatch.error

    def factory(coords, tris, tags, tag_map, pattern=None):
        return SimulationSession(coords, tris, tags, tag_map, backend=FailingBatch(), pattern=pattern)

    # the no-diamond production grid (read-flux projection batched too): 2 x 2 points on one width
    cfg = _cfg("geballe_no_diamond", 16.0, 6)
    cfg["heating"]["file"] = os.path.join(ROOT, cfg["heating"]["file"])
    cfg_path = str(tmp_path / "base.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    grid = ((8e-6, 2e-5), (3.0, 5.0), (1.84e-6, 1.84e-6), (2, 2, 1))
    FailingBatch.error = NotConverged(-4, "batched PCG not converged in 7 iterations")
    ok, failed = ps.run_parameter_sweep(cfg_path, str(tmp_path / "o1"), *grid, base_mesh_folder=str(tmp_path / "m"),
                                        session_factory=factory, batch=8)
    assert FailingBatch.calls == 1 and len(ok) == 4 and not failed
    assert all("not converged in 7 iterations" in r["batch_error"] and r["status"] == "success" for r in ok)
    assert all(os.path.isfile(os.path.join(r["output_dir"], "radial_gradient_raw.csv")) for r in ok)
    with open(os.path.join(str(tmp_path / "o1"), "successful_runs.csv")) as f:
        assert all("NotConverged" in r["batch_error"] for r in csv.DictReader(f))
    assert "batched time loop of 4 points failed" in capsys.readouterr().err
    # a device-side failure is not retried
    FailingBatch.error, FailingBatch.calls = HipError(-3, "hipStreamSynchronize failed: illegal memory access"), 0
    ok, failed = ps.run_parameter_sweep(cfg_path, str(tmp_path / "o2"), *grid, base_mesh_folder=str(tmp_path / "m"),
                                        session_factory=factory, batch=8)
    assert FailingBatch.calls == 1 and not ok and len(failed) == 4
    assert all("illegal memory access" in r["error"] and r["batch_error"] for r in failed)
    # the kappa-only sweep driver does the same
    cfgw = _cfg("geballe_with_diamond", 16.0, 6)
    cfgw["heating"]["file"] = os.path.join(ROOT, cfgw["heating"]["file"])
    mesh = str(tmp_path / "mw")
    prepare_mesh(cfgw, mesh, True, build_stack(cfgw))
    FailingBatch.error, FailingBatch.calls = NotConverged(-4, "batched PCG breakdown"), 0
    rows = ps.run_kappa_sweep(cfgw, mesh, [3.4, 3.8, 4.2, 4.6], str(tmp_path / "o3"), session_factory=factory, batch=4)
    assert FailingBatch.calls == 1 and [r["status"] for r in rows] == ["success"] * 4 and all("breakdown" in r["batch_error"] for r in rows)
    # configurations that do not share their watcher points cannot share a batch
    from heatflow_amd.driver import run_simulation_batch_impl
    sess = SimulationSession(*prepare_mesh(cfgw, mesh, False, build_stack(cfgw)), backend=OracleBackend())
    wp = watcher_points(cfgw)
    with pytest.raises(ValueError, match="share their watcher points"):
        run_simulation_batch_impl("with_diamond", [cfgw, cfgw], [str(tmp_path / "a"), str(tmp_path / "b")],
                                  [wp, {"pside": wp["pside"], "oside": (wp["oside"][0] + 1e-7, 0.0)}], sess)


def test_root_level_modules_keep_the_reference_names():
    """A user of the reference imports `parameter_sweep`, `run_with_diamond`, `run_no_diamond`, `run_no_diamond_1d` from the
    repository root (parameter_sweep.py:43, with_diamond.py:1, no_diamond.py:1, no_diamond_1d.py:1): the same names resolve
    here, and the experiment scripts exist beside them."""
    import importlib
    import os
    import sys

    from conftest import ROOT

    sys.path.insert(0, ROOT)
    try:
        for name, attrs in (("parameter_sweep", ("run_parameter_sweep", "run_single_simulation", "create_parameter_grid", "main")),
                            ("run_with_diamond", ("run_simulation",)), ("run_no_diamond", ("run_simulation",)),
                            ("run_no_diamond_1d", ("run_1d",))):
            mod = importlib.import_module(name)
            for a in attrs:
                assert hasattr(mod, a), (name, a)
    finally:
        sys.path.remove(ROOT)
    for script in ("with_diamond.py", "no_diamond.py", "no_diamond_1d.py", "sweep_test.py", "parameter_sweep.py"):
        assert os.path.isfile(os.path.join(ROOT, script)), script
