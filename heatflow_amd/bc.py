"""Dirichlet rows on a domain edge or an interior line.

Host-side mirror of the reference ``RowDirichletBC`` (dirichlet_bc/bc.py:6-146): same
constructor keywords, same DOF-location predicates (``np.isclose(..., atol=width)`` on
the edge coordinate, optional ``|s - center| <= length/2 + 1e-14`` clip, bc.py:50-101),
same ``RuntimeError`` when nothing is found (bc.py:105-106).  ``V`` is anything with a
``coords`` (n, 2) array - here the P1 space is the mesh's node set, DOF index == node
index (the reference relies on the same identity for its watchers,
run_with_diamond.py:443-449).

The values live in ``self.values`` (one per ``row_dofs`` entry) and are refreshed by
``update(t)``; :func:`merge_bcs` resolves overlapping BCs the way dolfinx applies a
list of them (later entry wins).
"""
from __future__ import annotations

import numpy as np


class P1Space:
    """Minimal stand-in for ``fem.functionspace(domain, ("Lagrange", 1))``: the node set."""

    def __init__(self, coords):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        if self.coords.ndim != 2 or self.coords.shape[1] != 2:
            raise ValueError("coords must be (n, 2) [z, r]")

    @property
    def num_dofs(self):
        return self.coords.shape[0]


class RowDirichletBC:
    def __init__(self, V, location, *, coord=None, length=None, center=None, width=1e-10, value=0.0, row_dofs=None):
        """``row_dofs`` (not in the reference): the DOF set of an earlier BC with the same location arguments on
        the same space - a sweep locates each edge once instead of once per point."""
        self.V = V if hasattr(V, "coords") else P1Space(V)
        xy = self.V.coords
        self.width = float(width)
        self.center = center
        self.length = length
        self.location = location
        if row_dofs is not None:
            if location in ("x", "y") and center is None:
                raise ValueError("row_dofs needs an explicit center for location 'x' / 'y'")
            self.row_dofs = np.ascontiguousarray(row_dofs, dtype=np.int32)
            self._finish(xy, value)
            return

        x0, x1 = xy[:, 0], xy[:, 1]
        xmin, xmax, ymin, ymax = x0.min(), x0.max(), x1.min(), x1.max()
        xmid, ymid = 0.5 * (xmin + xmax), 0.5 * (ymin + ymax)
        half = None if length is None else 0.5 * length
        if location in ("x", "y") and center is None:
            self.center = xmid if location == "x" else ymid

        def along(vals, c):
            if half is None:
                return np.ones(vals.shape, dtype=bool)
            return np.abs(vals - c) <= half + 1e-14

        def on(vals, c):
            return np.isclose(vals, c, atol=self.width)

        if location == "left":
            mask = on(x0, xmin) & along(x1, ymid)
        elif location == "right":
            mask = on(x0, xmax) & along(x1, ymid)
        elif location == "bottom":
            mask = on(x1, ymin) & along(x0, xmid)
        elif location == "top":
            mask = on(x1, ymax) & along(x0, xmid)
        elif location == "outer":
            mask = ((on(x0, xmin) | on(x0, xmax)) & along(x1, ymid)) | ((on(x1, ymin) | on(x1, ymax)) & along(x0, xmid))
        elif location == "x":
            if coord is None:
                raise ValueError("coord required when location='x'.")
            mask = on(x0, float(coord)) & along(x1, self.center)
        elif location == "y":
            if coord is None:
                raise ValueError("coord required when location='y'.")
            mask = on(x1, float(coord)) & along(x0, self.center)
        else:
            raise ValueError("Unknown location keyword.")

        self.row_dofs = np.nonzero(mask)[0].astype(np.int32)
        self._finish(xy, value)

    def _finish(self, xy, value):
        if self.row_dofs.size == 0:
            raise RuntimeError("No DOFs found for requested BC location/length.")
        self.dof_coords = xy[self.row_dofs]
        self._value = value if callable(value) else None
        self._const = None if callable(value) else float(value)
        self.values = np.zeros(self.row_dofs.size, dtype=np.float64)

    def update(self, t):
        """Refill the boundary values at time ``t`` (reference bc.py:128-137).  A callable
        ``value(x, y, t)`` is tried on whole coordinate arrays first and falls back to the
        reference's per-DOF calls if it does not broadcast."""
        if self._value is None:
            self.values[:] = self._const
            return self.values
        x, y = self.dof_coords[:, 0], self.dof_coords[:, 1]
        try:
            v = np.asarray(self._value(x, y, t), dtype=np.float64)
            if v.shape == ():
                v = np.full(x.shape, float(v))
            if v.shape != x.shape:
                raise ValueError
        except (TypeError, ValueError):
            v = np.array([self._value(a, b, t) for a, b in zip(x, y)], dtype=np.float64)
        self.values[:] = v
        return self.values

    @staticmethod
    def constant(V, location, value, *, coord=None, length=None, width=1e-12):
        bc = RowDirichletBC(V, location, coord=coord, length=length, width=width, value=value)
        bc.update(0.0)
        return bc

    @staticmethod
    def describe_row_bcs(bc_list, *, label="Row BC"):
        for k, bc in enumerate(bc_list):
            if not isinstance(bc, RowDirichletBC):
                continue
            xy = bc.dof_coords
            print(f"{label} #{k}: x in [{xy[:, 0].min():.3e}, {xy[:, 0].max():.3e}]  "
                  f"y in [{xy[:, 1].min():.3e}, {xy[:, 1].max():.3e}]  (n = {xy.shape[0]} DOFs)")


def merge_bcs(bc_list):
    """Unique Dirichlet DOFs of a BC list plus, per DOF, (index of the BC that owns it,
    position inside that BC's row_dofs).  Overlaps go to the LATER list entry, which is
    what dolfinx's set_bc / apply_lifting do when handed ``[bc.bc for bc in obj_bcs]``
    (run_with_diamond.py:373-374, 477-479)."""
    n_max = max(int(bc.row_dofs.max()) for bc in bc_list) + 1
    owner = np.full(n_max, -1, dtype=np.int64)
    pos = np.zeros(n_max, dtype=np.int64)
    for k, bc in enumerate(bc_list):
        owner[bc.row_dofs] = k
        pos[bc.row_dofs] = np.arange(bc.row_dofs.size)
    dofs = np.nonzero(owner >= 0)[0].astype(np.int32)
    return dofs, owner[dofs], pos[dofs]


def gather_plan(n_bcs, owner, pos):
    """Per BC the positions it fills in the merged list and the entries of its values that go there."""
    plan = []
    for k in range(n_bcs):
        sel = np.nonzero(owner == k)[0]
        plan.append((sel, pos[sel]))
    return plan


def gather_bc_values(bc_list, owner, pos, plan=None):
    """Current values of the merged DOF list (after each bc.update(t)); ``plan`` = gather_plan(...) computed once."""
    g = np.empty(owner.shape, dtype=np.float64)
    if plan is None:
        plan = gather_plan(len(bc_list), owner, pos)
    for bc, (sel, src) in zip(bc_list, plan):
        if sel.size:
            g[sel] = bc.values[src]
    return g
