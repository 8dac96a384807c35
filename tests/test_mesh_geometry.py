"""Host logic: YAML coercions, layer-stack arithmetic, the quadtree mesher, mesh I/O."""
import os

import numpy as np
import pytest
import yaml

from conftest import ROOT, build_case, load_cfg
from heatflow_amd.geometry import build_stack, scale_mesh_sizes, stack_no_diamond, stack_with_diamond, watcher_points
from heatflow_amd.materials import Material
from heatflow_amd.mesh import Mesh, load_mesh_arrays, read_msh, reorder_mesh


def test_yaml_mantissas_without_a_dot_load_as_strings_and_are_coerced():
    cfg = load_cfg("geballe_with_diamond")
    assert isinstance(cfg["mats"]["g_ins"]["r"], str)          # '5e-6' under YAML 1.1
    assert isinstance(cfg["mats"]["p_coupler"]["z"], float)    # 6.2e-08
    st = stack_with_diamond(cfg)
    assert st.by_name("g_ins").boundaries[3] - st.by_name("g_ins").boundaries[2] == pytest.approx(5e-6)


def test_with_diamond_boxes():
    st = stack_with_diamond(load_cfg("geballe_with_diamond"))
    assert [m.name for m in st.materials] == ["p_diam", "p_ins", "p_coupler", "p_sample", "o_coupler", "o_ins",
                                              "o_diam", "gasket", "g_ins"]
    assert st.bounds[0] == pytest.approx(-44.182e-6) and st.bounds[1] == pytest.approx(47.282e-6)
    assert st.bounds[3] == pytest.approx(80e-6)
    assert st.heated_z == pytest.approx(-0.982e-6)
    assert st.heated_z == st.by_name("p_coupler").boundaries[0] == st.by_name("p_ins").boundaries[1]
    assert st.by_name("p_diam").properties == {"rho_cv": 3500.0 * 510.0, "k": 2000.0}
    assert st.by_name("gasket").boundaries == pytest.approx([-4.182e-6, 7.282e-6, 25e-6, 80e-6])


def test_no_diamond_boxes_and_unmeshed_bound():
    st = stack_no_diamond(load_cfg("geballe_no_diamond"))
    assert len(st.materials) == 5
    assert st.bounds == pytest.approx([-4.182e-6, 7.282e-6, 0.0, 40e-6])   # r_sample + r_ins_oside, not meshed
    assert max(m.boundaries[3] for m in st.materials) == pytest.approx(20e-6)
    assert st.heated_z == pytest.approx(-0.982e-6)


def test_watcher_points_sit_in_the_coupler_mid_planes():
    wp = watcher_points(load_cfg("geballe_with_diamond"))
    assert wp["pside"] == pytest.approx((-0.951e-6, 0.0)) and wp["oside"] == pytest.approx((0.951e-6, 0.0))
    wp2 = watcher_points(load_cfg("geballe_no_diamond"))
    assert wp2["pside"][0] == pytest.approx(wp["pside"][0])


def test_konopkova_stub_cannot_be_parsed_into_a_run():
    cfg = load_cfg("konopkova")
    with pytest.raises((KeyError, ValueError, TypeError)):
        build_stack(cfg)


def test_material_validation_matches_reference():
    with pytest.raises(TypeError):
        Material(3, [0, 1, 0, 1])
    with pytest.raises(ValueError):
        Material("a", [0, 1, 0])
    with pytest.raises(ValueError):
        Material("a", [1, 0, 0, 1])
    m = Material("a", [0, 1, 0, 2], {"k": 1.0}, 0.1)
    assert m.contains(0.5, 1.0) and not m.contains(1.5, 1.0)


def _edges(tris):
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    e.sort(axis=1)
    return e


@pytest.mark.parametrize("name,scale", [("geballe_with_diamond", 6.0), ("geballe_no_diamond", 4.0)])
def test_mesh_is_conforming_tagged_and_sized(name, scale):
    cfg, st, mesh = build_case(name, scale)
    c, t = mesh.coords, mesh.tris
    p0, p1, p2 = c[t[:, 0]], c[t[:, 1]], c[t[:, 2]]
    area = 0.5 * ((p1[:, 0] - p0[:, 0]) * (p2[:, 1] - p0[:, 1]) - (p2[:, 0] - p0[:, 0]) * (p1[:, 1] - p0[:, 1]))
    assert (area > 0).all()                                                   # CCW, non-degenerate
    assert area.sum() == pytest.approx(sum((m.boundaries[1] - m.boundaries[0]) * (m.boundaries[3] - m.boundaries[2])
                                           for m in st.materials), rel=1e-12)
    # conforming: every edge is shared by exactly two triangles, or lies on the outer boundary (one)
    e = _edges(t)
    uniq, cnt = np.unique(e, axis=0, return_counts=True)
    assert set(cnt.tolist()) <= {1, 2}
    bnd = uniq[cnt == 1]
    zmin, zmax, rmax = c[:, 0].min(), c[:, 0].max(), c[:, 1].max()
    mid = 0.5 * (c[bnd[:, 0]] + c[bnd[:, 1]])
    on_hull = np.isclose(mid[:, 0], zmin) | np.isclose(mid[:, 0], zmax) | np.isclose(mid[:, 1], 0.0) | np.isclose(mid[:, 1], rmax)
    assert on_hull.all()
    # every node is used
    assert len(np.unique(t)) == len(c)
    # tags: triangle centroid inside its material's box; legs no longer than mesh_size
    cen = (p0 + p1 + p2) / 3
    for k, m in enumerate(st.materials):
        sel = mesh.tags == k + 1
        assert sel.any()
        b = m.boundaries
        assert ((cen[sel, 0] > b[0]) & (cen[sel, 0] < b[1]) & (cen[sel, 1] > b[2]) & (cen[sel, 1] < b[3])).all()
        legs = np.sort(np.stack([np.linalg.norm(p1 - p0, axis=1), np.linalg.norm(p2 - p1, axis=1),
                                 np.linalg.norm(p0 - p2, axis=1)], axis=1)[sel], axis=1)
        assert legs[:, 1].max() <= m.mesh_size * (1 + 1e-9)                  # the two legs of the right triangle
    # shape quality: plain cells give right triangles; fans of (near-square) rectangles stay well
    # away from degenerate: every angle in [25, 130] degrees
    def ang(a, b, c_):
        u, v = b - a, c_ - a
        cosv = (u * v).sum(1) / np.sqrt((u * u).sum(1) * (v * v).sum(1))
        return np.degrees(np.arccos(np.clip(cosv, -1, 1)))
    angles = np.stack([ang(p0, p1, p2), ang(p1, p2, p0), ang(p2, p0, p1)], axis=1)
    assert angles.min() > 25.0 and angles.max() < 130.0
    assert np.isclose(angles.max(axis=1), 90.0).mean() > 0.7
    assert mesh.material_tags == {m.name: k + 1 for k, m in enumerate(st.materials)}


def test_stock_no_diamond_has_1001_nodes_on_the_heated_line():
    """Reference notebooks report 1001 inner-BC DOFs for r in [0, 20 um] at h = 0.02 um."""
    cfg, st, mesh = build_case("geballe_no_diamond", 1.0)
    on_line = np.isclose(mesh.coords[:, 0], st.heated_z, atol=1e-10)
    assert on_line.sum() == 1001
    assert 1.0e5 < len(mesh.coords) < 2.5e5


def test_mesher_is_deterministic_and_grades_into_the_diamond():
    a = build_case("geballe_with_diamond", 8.0)[2]
    b = build_case("geballe_with_diamond", 8.0)[2]
    assert np.array_equal(a.coords, b.coords) and np.array_equal(a.tris, b.tris) and np.array_equal(a.tags, b.tags)
    assert np.array_equal(a.node_ij, b.node_ij) and a.stats == b.stats and a.tris.dtype == b.tris.dtype and a.tags.dtype == b.tags.dtype
    assert a.stats["max_level"] >= 5 and a.stats["n_fan_cells"] > 0
    # far fewer nodes than a tensor grid at the finest spacing would need
    assert a.stats["n_nodes"] < 0.02 * a.stats["base_grid"][0] * a.stats["base_grid"][1]


def test_duplicate_rectangles_are_rejected():
    m1 = Material("a", [0, 1, 0, 1], {}, 0.5)
    m2 = Material("b", [0, 1, 0, 1], {}, 0.5)
    with pytest.raises(RuntimeError):
        Mesh("m", [0, 2, 0, 2], [m1, m2]).build_mesh()


def test_msh_roundtrip(tmp_path):
    cfg, st, mesh = build_case("geballe_with_diamond", 16.0)
    path = str(tmp_path / "mesh.msh")
    mesh.write(path)
    c1, t1, g1 = load_mesh_arrays(path)                   # npz sidecar
    c2, t2, g2 = read_msh(path)                           # ASCII MSH 2.2
    for c, t, g in ((c1, t1, g1), (c2, t2, g2)):
        assert np.array_equal(c, mesh.coords) and np.array_equal(t, mesh.tris) and np.array_equal(g, mesh.tags)
    with open(path) as f:
        head = f.read(200)
    assert head.startswith("$MeshFormat\n2.2 0 8") and '"p_diam"' in head


def test_scale_reaches_one_million_dof_band():
    """BASELINE C3: one factor on every mesh: value so that n = 1.0e6 +- 5 % (bench.py uses 0.43).
    Checked through the cell count formula on a cheap proxy: halving h quadruples the nodes."""
    n1 = build_case("geballe_with_diamond", 4.0)[2].stats["n_nodes"]
    n2 = build_case("geballe_with_diamond", 2.0)[2].stats["n_nodes"]
    assert 3.3 < n2 / n1 < 4.3


MSH41 = """$MeshFormat
4.1 0 8
$EndMeshFormat
$PhysicalNames
2
2 7 "left_mat"
2 9 "right_mat"
$EndPhysicalNames
$Entities
6 7 2 0
1 0 0 0 0
2 1 0 0 0
3 1 1 0 0
4 0 1 0 0
5 2 0 0 0
6 2 1 0 0
1 0 0 0 1 0 0 0 2 1 -2
2 1 0 0 1 1 0 0 2 2 -3
3 0 1 0 1 1 0 0 2 3 -4
4 0 0 0 0 1 0 0 2 4 -1
5 1 0 0 2 0 0 0 2 2 -5
6 2 0 0 2 1 0 0 2 5 -6
7 1 1 0 2 1 0 0 2 6 -3
1 0 0 0 1 1 0 1 7 4 1 2 3 4
2 1 0 0 2 1 0 1 9 4 5 6 7 -2
$EndEntities
$Nodes
8 6 1 6
0 1 0 1
1
0 0 0
0 2 0 1
2
1 0 0
0 3 0 1
3
1 1 0
0 4 0 1
4
0 1 0
0 5 0 1
5
2 0 0
0 6 0 1
6
2 1 0
2 1 0 0
2 2 0 0
$EndNodes
$Elements
3 5 1 5
1 2 1 1
1 2 3
2 1 2 2
2 1 2 3
3 1 3 4
2 2 2 2
4 2 5 6
5 2 6 3
$EndElements
"""


def test_msh41_ascii_reader(tmp_path):
    """A reference-generated mesh.msh (MSH 4.1, physical groups per surface) can be fed in."""
    path = tmp_path / "mesh41.msh"
    path.write_text(MSH41)
    coords, tris, tags = read_msh(str(path))
    assert coords.shape == (6, 2) and tris.shape == (4, 3)
    assert tags.tolist() == [7, 7, 9, 9]                       # physical group of the owning surface
    assert np.allclose(coords[tris[2]], [[1, 0], [2, 0], [2, 1]])
    from heatflow_amd.mesh import MeshError
    bad = tmp_path / "bin.msh"
    bad.write_text("$MeshFormat\n4.1 1 8\n$EndMeshFormat\n")
    with pytest.raises(MeshError, match="binary"):
        read_msh(str(bad))


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_mesher_on_random_layer_stacks(seed):
    """Random touching-rectangle stacks (thin and thick layers, mesh sizes spread over 2.5 decades,
    an L-shaped union): conforming, every triangle inside its box, sizes honoured, areas add up."""
    rng = np.random.default_rng(seed)
    nlay = int(rng.integers(2, 6))
    thick = 10.0 ** rng.uniform(-7.5, -5.0, nlay)
    z = np.concatenate([[0.0], np.cumsum(thick)]) - 1e-6
    r_in = float(10.0 ** rng.uniform(-6, -5))
    mats = [Material(f"l{k}", [z[k], z[k + 1], 0.0, r_in], {"k": 1.0, "rho_cv": 1.0},
                     float(thick[k] / rng.uniform(1.5, 12.0))) for k in range(nlay)]
    if seed % 2 == 0:                       # an outer ring around all layers, coarse
        mats.append(Material("ring", [z[0], z[-1], r_in, r_in * rng.uniform(1.5, 4.0)], {"k": 1.0, "rho_cv": 1.0},
                             float(max(m.mesh_size for m in mats) * rng.uniform(2.0, 30.0))))
    mesh = Mesh("m", [z[0], z[-1], 0.0, max(m.boundaries[3] for m in mats)], mats).build_mesh()
    # the native quadtree passes (libheatflow_host.so) and their numpy statement give the same mesh, bit for bit
    other = Mesh("m", [z[0], z[-1], 0.0, max(m.boundaries[3] for m in mats)], mats).build_mesh(use_native=False)
    assert np.array_equal(mesh.coords, other.coords) and np.array_equal(mesh.tris, other.tris)
    assert np.array_equal(mesh.tags, other.tags) and np.array_equal(mesh.node_ij, other.node_ij) and mesh.stats == other.stats
    c, t = mesh.coords, mesh.tris
    p0, p1, p2 = c[t[:, 0]], c[t[:, 1]], c[t[:, 2]]
    area = 0.5 * ((p1[:, 0] - p0[:, 0]) * (p2[:, 1] - p0[:, 1]) - (p2[:, 0] - p0[:, 0]) * (p1[:, 1] - p0[:, 1]))
    assert (area > 0).all()
    assert area.sum() == pytest.approx(sum((m.boundaries[1] - m.boundaries[0]) * (m.boundaries[3] - m.boundaries[2])
                                           for m in mats), rel=1e-11)
    uniq, cnt = np.unique(_edges(t), axis=0, return_counts=True)
    assert set(cnt.tolist()) <= {1, 2}
    assert len(np.unique(t)) == len(c)
    cen = (p0 + p1 + p2) / 3
    for k, m in enumerate(mats):
        sel = mesh.tags == k + 1
        b = m.boundaries
        assert sel.any() and ((cen[sel, 0] > b[0]) & (cen[sel, 0] < b[1]) & (cen[sel, 1] > b[2]) & (cen[sel, 1] < b[3])).all()
        legs = np.sort(np.stack([np.linalg.norm(p1 - p0, axis=1), np.linalg.norm(p2 - p1, axis=1),
                                 np.linalg.norm(p0 - p2, axis=1)], axis=1)[sel], axis=1)
        assert legs[:, 1].max() <= m.mesh_size * (1 + 1e-9)


def test_reorder_mesh_is_a_consistent_renumbering():
    cfg, st, mesh = build_case("geballe_with_diamond", 16.0)
    rng = np.random.default_rng(0)
    shuffle = rng.permutation(len(mesh.coords))                  # scramble like an external mesher would
    inv = np.empty_like(shuffle)
    inv[shuffle] = np.arange(len(shuffle))
    c_s, t_s = mesh.coords[shuffle], inv[mesh.tris][rng.permutation(len(mesh.tris))]
    tags_s = np.arange(len(t_s), dtype=np.int32)                 # unique tags: track every triangle
    c2, t2, g2, perm = reorder_mesh(c_s, t_s, tags_s)
    assert np.array_equal(c2, c_s[perm]) and sorted(g2.tolist()) == list(range(len(t_s)))
    # every triangle keeps its three vertex positions
    assert np.array_equal(np.sort(c2[t2].reshape(len(t2), -1), axis=1), np.sort(c_s[t_s][g2].reshape(len(t2), -1), axis=1))
    # locality: mean |node index spread| inside a triangle shrinks by a large factor
    spread = lambda t: np.mean(t.max(axis=1) - t.min(axis=1))
    assert spread(t2) < 0.1 * spread(t_s)


@pytest.mark.parametrize("name,scale", [("geballe_with_diamond", 2.0), ("geballe_no_diamond", 2.0), ("geballe_with_diamond", 8.0)])
def test_native_and_numpy_quadtree_agree_on_the_reference_stacks(name, scale):
    from heatflow_amd import hostlib
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes

    assert hostlib.available(), "gcc is part of the image: libheatflow_host.so must build"
    stack = build_stack(scale_mesh_sizes(load_cfg(name), scale))
    a = Mesh("a", stack.bounds, stack.materials).build_mesh(use_native=True)
    b = Mesh("b", stack.bounds, stack.materials).build_mesh(use_native=False)
    assert np.array_equal(a.coords, b.coords) and np.array_equal(a.tris, b.tris) and np.array_equal(a.tags, b.tags)


def _tri_key(coords, tris, tags):
    """Triangles as sorted coordinate tuples + tag, in a canonical order (node numbering independent)."""
    a = np.concatenate([np.sort(coords[tris].reshape(len(tris), -1), axis=1), np.asarray(tags, dtype=float)[:, None]], axis=1)
    return a[np.lexsort(a.T[::-1])]


def test_msh41_writer_round_trip_keeps_physical_tags(tmp_path):
    """MSH 4.1 as the reference's gmsh.write lays it out (one surface + one physical group per material,
    mesh_and_materials/mesh.py:114-126, :191-195): written by write_msh41, read back by read_msh.  Surface
    ids and physical tags are numbered differently on purpose: the cell tag must be the physical one."""
    from heatflow_amd.mesh import write_msh41

    cfg, stack, mesh = build_case("geballe_with_diamond", 16.0)
    path = str(tmp_path / "m41.msh")
    write_msh41(path, mesh.coords, mesh.tris, mesh.tags, mesh.material_tags,
                surface_ids={int(t): 100 - int(t) for t in np.unique(mesh.tags)})
    head = open(path).read(4000)
    assert head.startswith("$MeshFormat\n4.1 0 8\n") and '2 4 "p_sample"' in head and "\n0 0 9 0\n" in head
    coords, tris, tags = read_msh(path)
    assert coords.shape == mesh.coords.shape and tris.shape == mesh.tris.shape
    assert np.array_equal(_tri_key(coords, tris, tags), _tri_key(mesh.coords, mesh.tris, mesh.tags))   # bit-exact (%.17g)
    # Mesh.write: a 2.2 file keeps this mesh's node and element order, so its npz sidecar and a read of the file agree
    # element by element; a 4.1 file groups elements by surface and therefore carries no sidecar (and removes a stale one):
    # both reload paths of one file always see one numbering
    from heatflow_amd.mesh import load_mesh_arrays
    p2 = str(tmp_path / "mesh.msh")
    mesh.write(p2)
    assert (tmp_path / "mesh.npz").is_file()
    c2, t2, g2 = read_msh(p2)
    cs, ts, gs = load_mesh_arrays(p2)
    assert np.array_equal(c2, cs) and np.array_equal(t2, ts) and np.array_equal(g2, gs)
    mesh.write(p2, version="4.1")
    assert not (tmp_path / "mesh.npz").exists()
    c2, t2, g2 = read_msh(p2)
    assert np.array_equal(_tri_key(c2, t2, g2), _tri_key(mesh.coords, mesh.tris, mesh.tags))
    ca, ta, ga = load_mesh_arrays(p2)
    cb, tb, gb = load_mesh_arrays(p2)
    assert np.array_equal(ta, tb) and np.array_equal(ga, gb) and np.array_equal(ca, cb)
    assert np.array_equal(_tri_key(ca, ta, ga), _tri_key(mesh.coords, mesh.tris, mesh.tags))


def test_native_mesh_construction_limits_and_the_numpy_route_beyond_them(monkeypatch):
    """hfh_mesh_build numbers nodes and triangles by Morton codes of the low 16 bits of the lattice indices, like the numpy
    statement; beyond 65535 lattice lines per direction build_mesh keeps the numpy route (same arrays), and the native entry
    point refuses such a lattice and malformed leaves instead of reading outside its bitmaps."""
    from heatflow_amd import hostlib
    from heatflow_amd.geometry import build_stack, scale_mesh_sizes

    stack = build_stack(scale_mesh_sizes(load_cfg("geballe_with_diamond"), 8.0))
    ref = Mesh("a", stack.bounds, stack.materials).build_mesh(use_native=True)
    monkeypatch.setattr(hostlib, "MESH_LATTICE_MAX", 16)          # every real lattice is "too large" now
    called = []
    monkeypatch.setattr(hostlib, "mesh_from_leaves", lambda *a, **k: called.append(1))
    other = Mesh("b", stack.bounds, stack.materials).build_mesh(use_native=True)
    assert not called
    assert np.array_equal(ref.coords, other.coords) and np.array_equal(ref.tris, other.tris) and np.array_equal(ref.tags, other.tags)
    monkeypatch.undo()
    i0 = np.array([0, 2], dtype=np.int64)
    j0 = np.array([0, 0], dtype=np.int64)
    lev = np.array([1, 1], dtype=np.int64)
    mat = np.zeros((4, 2), dtype=np.int8)
    zc, rc = np.linspace(0.0, 1.0, 5), np.linspace(0.0, 1.0, 3)
    coords, node_ij, tris, tags, n_fan = hostlib.mesh_from_leaves(i0, j0, lev, mat, zc, rc)       # two 2 x 2 leaves side by side
    assert len(coords) == 6 and len(tris) == 4 and n_fan == 0 and set(tags) == {1}
    with pytest.raises(RuntimeError):                                  # a leaf sticking out of the lattice
        hostlib.mesh_from_leaves(np.array([4], dtype=np.int64), j0[:1], lev[:1], mat, zc, rc)
    with pytest.raises(RuntimeError):                                  # more lattice lines than 16-bit Morton codes order
        hostlib.mesh_from_leaves(i0, j0, lev, np.zeros((4, 70000), dtype=np.int8), zc, np.linspace(0.0, 1.0, 70001))
