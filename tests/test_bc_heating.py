import numpy as np
import pytest

from conftest import HEATING_CSV, build_case
from heatflow_amd.bc import P1Space, RowDirichletBC, gather_bc_values, merge_bcs
from heatflow_amd.heating import HeatingCurve
from oracle import heat_oracle as ho


@pytest.fixture(scope="module")
def nd():
    return build_case("geballe_no_diamond", 8.0)


def test_row_dofs_agree_with_the_oracle_restatement(nd):
    cfg, st, mesh = nd
    V = P1Space(mesh.coords)
    for loc, kw in (("left", {}), ("right", {}), ("top", {}), ("bottom", {}),
                    ("x", dict(coord=st.heated_z, length=40e-6, center=0.0)), ("y", dict(coord=0.0))):
        bc = RowDirichletBC(V, loc, **kw)
        assert np.array_equal(bc.row_dofs, ho.locate_row_dofs(mesh.coords, loc, **kw))
    assert np.allclose(RowDirichletBC(V, "left").dof_coords[:, 0], mesh.coords[:, 0].min())


def test_length_clip_and_errors(nd):
    cfg, st, mesh = nd
    V = P1Space(mesh.coords)
    full = RowDirichletBC(V, "x", coord=st.heated_z, length=40e-6, center=0.0)
    half = RowDirichletBC(V, "x", coord=st.heated_z, length=20e-6, center=0.0)
    assert half.row_dofs.size < full.row_dofs.size and mesh.coords[half.row_dofs, 1].max() <= 10e-6 + 1e-14
    with pytest.raises(RuntimeError, match="No DOFs found"):
        RowDirichletBC(V, "x", coord=1.0)
    with pytest.raises(ValueError):
        RowDirichletBC(V, "x")
    with pytest.raises(ValueError):
        RowDirichletBC(V, "diagonal")
    outer = RowDirichletBC(V, "outer")
    assert outer.row_dofs.size > RowDirichletBC(V, "left").row_dofs.size


def test_values_callable_vectorised_and_scalar_fallback(nd):
    cfg, st, mesh = nd
    V = P1Space(mesh.coords)
    heat = HeatingCurve(HEATING_CSV, 300.0, 1.32e-5)
    a = RowDirichletBC(V, "x", coord=st.heated_z, length=40e-6, center=0.0, value=heat.gaussian)
    b = RowDirichletBC(V, "x", coord=st.heated_z, length=40e-6, center=0.0,
                       value=lambda x, y, t: float(heat.gaussian(float(x), float(y), t)) if np.ndim(x) == 0 else (_ for _ in ()).throw(TypeError()))
    va, vb = a.update(4e-6).copy(), b.update(4e-6).copy()
    assert np.array_equal(va, vb)
    h_time, h_temp = ho.read_heating_csv(HEATING_CSV)
    assert np.allclose(va, ho.gaussian_bc_values(a.dof_coords[:, 1], 4e-6, h_time, h_temp, 300.0, 1.32e-5), rtol=1e-15)
    c = RowDirichletBC.constant(V, "left", 300.0)
    assert (c.values == 300.0).all()


def test_merge_is_last_wins_and_matches_oracle(nd):
    cfg, st, mesh = nd
    V = P1Space(mesh.coords)
    heat = HeatingCurve(HEATING_CSV, 300.0, 1.32e-5)
    bcs = [RowDirichletBC(V, "left", value=300.0), RowDirichletBC(V, "right", value=300.0),
           RowDirichletBC(V, "top", value=300.0),
           RowDirichletBC(V, "x", coord=st.heated_z, length=40e-6, center=0.0, value=heat.gaussian)]
    dofs, owner, pos = merge_bcs(bcs)
    odofs, oowner = ho.merge_bcs([(b.row_dofs, k) for k, b in enumerate(bcs)])
    assert np.array_equal(dofs, odofs) and np.array_equal(owner, oowner)
    assert len(dofs) == len(np.unique(dofs)) == sum(b.row_dofs.size for b in bcs) - 3   # 2 corners + (z*, rmax)
    for b in bcs:
        b.update(5e-6)
    g = gather_bc_values(bcs, owner, pos)
    corner = np.nonzero(np.isclose(mesh.coords[dofs, 0], st.heated_z, atol=1e-10) & np.isclose(mesh.coords[dofs, 1], 20e-6))[0]
    assert len(corner) == 1 and owner[corner[0]] == 3 and g[corner[0]] > 300.0


def test_heating_curve(tmp_path):
    h = HeatingCurve(HEATING_CSV, 300.0, 1.32e-5)
    assert h.amplitude(0.0) == 300.0 and h.amplitude(1.0) == pytest.approx(h.temp[-1] - h.temp[0] + 300.0)
    assert h.temp_normed.max() <= 1.0
    bad = tmp_path / "bad.csv"
    bad.write_text("time,oside\n1,2\n")
    with pytest.raises(ValueError, match="'temp' column"):
        HeatingCurve(str(bad), 300.0, 1e-5)
    messy = tmp_path / "messy.csv"
    messy.write_text("time,temp\n2e-6,400\nx,1\n1e-6,350\n,\n")
    hm = HeatingCurve(str(messy), 300.0, 1e-5)
    assert hm.time.tolist() == [1e-6, 2e-6] and hm.temp.tolist() == [350.0, 400.0]
