"""Conforming triangulation of a stack of touching material rectangles.

Fills the role of the reference's ``Mesh.build_mesh`` (mesh_and_materials/
mesh.py:81-149), which drives gmsh: one surface per material (cell tag =
surface id, 1-based in list order, mesh.py:114), target size = min over the
material boxes of (h_mat inside, h_max outside).  gmsh is not available and its
Frontal-Delaunay output cannot be reproduced, so this is an own, deterministic
mesher built for the same contract:

* every triangle lies in exactly one material box and carries that box's tag;
* element size <= the material's ``mesh_size`` (within a factor 2 below it);
* the mesh is conforming across all interfaces, including the heated line;
* sizes grade 2:1 from the fine layers into the coarse diamonds/gasket, so the
  coarse regions hold isotropic elements (no high-aspect slivers that would
  stiffen the Jacobi-preconditioned operator).

Method: a 2:1-balanced quadtree over a virtual tensor-product *base grid*
(spacing ~ the smallest ``mesh_size``; base lines are placed per slab between
material break points, so interfaces are grid lines).  A leaf of level L spans
2^L x 2^L base cells.  Leaves without hanging nodes are cut into 2 right
triangles; leaves with hanging mid-edge nodes get a centre node and a fan.
Plain cells give right triangles; fan triangles of the (near-square) graded
cells keep all angles within ~[30, 120] degrees.  Nodes and triangles are
numbered along a Morton curve for cache locality of the SpMV gathers.

The quadtree lives in a dense int8 level map over the base cells (20-80 M
cells), processed with numpy block reductions - no Python loop over cells.
"""
from __future__ import annotations

import math
import os

import numpy as np

SCALE = 1e6  # kept for API parity with the reference module (1 model unit = 1 um); unused


class MeshError(RuntimeError):
    pass


def _round_key(x):
    return round(float(x), 12)  # 1 pm, the precision reference _check_mesh uses (mesh.py:54)


def _unique_breaks(values):
    seen = {}
    for v in values:
        seen.setdefault(_round_key(v), float(v))
    return np.array([seen[k] for k in sorted(seen)], dtype=np.float64)


def _roundup(x, m):
    return ((x + m - 1) // m) * m


def _morton(i, j):
    """Interleave the low 16 bits of i and j -> 32-bit Morton code (uint64)."""
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    return spread(i) | (spread(j) << np.uint64(1))


def _blockreduce(a, fn):
    """2x2 block reduction of a 2-D array with even dims (fn = np.minimum / np.maximum)."""
    return fn(fn(a[0::2, 0::2], a[1::2, 0::2]), fn(a[0::2, 1::2], a[1::2, 1::2]))


def _min3x3(a, fill):
    """3x3 minimum filter with ``fill`` outside."""
    p = np.full((a.shape[0] + 2, a.shape[1] + 2), fill, dtype=a.dtype)
    p[1:-1, 1:-1] = a
    out = p[1:-1, 1:-1].copy()
    for di in (0, 1, 2):
        for dj in (0, 1, 2):
            if di == 1 and dj == 1:
                continue
            np.minimum(out, p[di:di + a.shape[0], dj:dj + a.shape[1]], out=out)
    return out


class Mesh:
    """Triangulated layer stack.

    Parameters mirror the reference ``Mesh(name, boundaries, materials)``
    (mesh.py:36-44).  After :meth:`build_mesh`:

    ``coords`` (n, 2) float64 [z, r] - ``tris`` (n_e, 3) int32, CCW -
    ``tags`` (n_e,) int32 - ``material_tags`` {name: tag}.
    """

    MAX_LEVEL = 12

    def __init__(self, name, boundaries, materials):
        if not isinstance(name, str):
            raise TypeError("name must be a string")
        if len(boundaries) != 4:
            raise ValueError("boundaries must be 4 floats")
        self.name = name
        self.boundaries = [float(b) for b in boundaries]
        self.materials = list(materials)
        self.material_tags = {}
        self.coords = None
        self.tris = None
        self.tags = None
        self.node_ij = None
        self.stats = {}

    # -- validation (same three checks as reference mesh.py:46-77) -----------------
    def _check_mesh(self, base_bounds):
        seen = {tuple(_round_key(x) for x in base_bounds): "BASE"}
        for m in self.materials:
            key = tuple(_round_key(x) for x in m.boundaries)
            if key in seen:
                raise RuntimeError(
                    f"Duplicate rectangle:\n    {m.name} has boundaries {key}\n    already used by {seen[key]}")
            seen[key] = m.name
        for m in self.materials:
            a, b, c, d = m.boundaries
            if b - a <= 0 or d - c <= 0:
                raise ValueError(f"{m.name}: invalid rectangle (bx,BX,by,BY) = {m.boundaries}")

    # -- base grid ------------------------------------------------------------------
    @staticmethod
    def _axis(breaks, slab_h, h0):
        """Base lines along one axis.  Returns (line coords, slab start indices)."""
        lines = [np.array([breaks[0]])]
        starts = [0]
        idx = 0
        for k in range(len(breaks) - 1):
            length = breaks[k + 1] - breaks[k]
            nmin = max(1, int(math.ceil(length / h0 - 1e-9)))
            lev = int(math.floor(math.log2(slab_h[k] / h0) + 1e-9))
            lev = max(0, min(lev, int(math.floor(math.log2(nmin)))))
            end = _roundup(idx + nmin, 1 << lev)
            n = end - idx
            seg = breaks[k] + length * (np.arange(1, n + 1, dtype=np.float64) / n)
            seg[-1] = breaks[k + 1]
            lines.append(seg)
            idx = end
            starts.append(idx)
        return np.concatenate(lines), np.array(starts, dtype=np.int64)

    @staticmethod
    def _quadtree_numpy(mat, allowed, lmax, nzp, nrp):
        """Level map + leaves in numpy: the steps hfh_quadtree_levels / hfh_quadtree_leaves restate in C."""
        # ---- initial levels: largest aligned block that is one material and small enough
        # "block of level lv is admissible" is monotone (an admissible block has admissible children), so a
        # cell's level is the highest admissible block above it: collect the maps bottom-up, assign top-down
        # with dense 2x upsampling (no scattered writes into the 28 M-cell base grid of a 1 M-node mesh).
        oks = []
        amin, mmin, mmax = allowed, mat, mat
        for lv in range(1, lmax + 1):
            amin = _blockreduce(amin, np.minimum)
            mmin = _blockreduce(mmin, np.minimum)
            mmax = _blockreduce(mmax, np.maximum)
            ok = (mmin == mmax) & (mmin >= 0) & (amin >= lv)
            if not ok.any():
                break
            oks.append(ok)
        del amin, mmin, mmax, allowed
        level = np.where(mat >= 0, 0, -1).astype(np.int8)
        if oks:
            cur = np.where(oks[-1], np.int8(len(oks)), np.int8(-1))
            for lv in range(len(oks) - 1, 0, -1):
                cur = cur.repeat(2, axis=0).repeat(2, axis=1)
                np.copyto(cur, np.int8(lv), where=(cur < 0) & oks[lv - 1])
            cur = cur.repeat(2, axis=0).repeat(2, axis=1)
            np.copyto(level, cur, where=cur > 0)
            del cur, oks

        # ---- 2:1 balance with smooth grading: a level-L leaf needs every one of its 8
        # same-size neighbour blocks to hold nothing finer than L-1.
        # Each sweep walks the levels upwards on a min-pyramid that is patched as blocks
        # are demoted, so upward ripples are caught in the same sweep; downward ripples
        # (new L-1 leaves next to L-3 cells) take another sweep.
        big = np.int8(127)
        for _sweep in range(2 * (lmax + 2)):
            changed = False
            pyr = np.where(level < 0, big, level)
            for lv in range(1, lmax + 1):
                pyr = _blockreduce(pyr, np.minimum)
                if lv < 2:
                    continue
                demote = (pyr == lv) & (_min3x3(pyr, big) < lv - 1)
                if demote.any():
                    ii, jj = np.nonzero(demote)
                    view = level.reshape(nzp >> lv, 1 << lv, nrp >> lv, 1 << lv)
                    view[ii, :, jj, :] = lv - 1
                    pyr[ii, jj] = lv - 1
                    changed = True
            if not changed:
                break
        else:
            raise MeshError("quadtree balance did not converge")

        # ---- leaves
        li, lj, ll = [], [], []
        pyr = np.where(level < 0, big, level)
        top = level
        for lv in range(0, lmax + 1):
            if lv > 0:
                pyr = _blockreduce(pyr, np.minimum)
                top = _blockreduce(top, np.maximum)
                is_leaf = (pyr == lv) & (top == lv)
            else:
                is_leaf = level == 0
            ii, jj = np.nonzero(is_leaf)
            li.append(ii.astype(np.int64) << lv)
            lj.append(jj.astype(np.int64) << lv)
            ll.append(np.full(ii.shape, lv, dtype=np.int64))
        i0 = np.concatenate(li)
        j0 = np.concatenate(lj)
        lev = np.concatenate(ll)
        return i0, j0, lev

    def build_mesh(self, verbose=False, use_native=None):
        mats = self.materials
        if not mats:
            raise MeshError("no materials")
        for m in mats:
            if m.mesh_size is None or m.mesh_size <= 0:
                raise MeshError(f"{m.name}: mesh_size required")
        self._check_mesh(self.boundaries)

        zb = _unique_breaks([v for m in mats for v in m.boundaries[:2]])
        rb = _unique_breaks([v for m in mats for v in m.boundaries[2:]])
        h0 = min(m.mesh_size for m in mats)

        def slab_sizes(breaks, lo, hi):
            out = []
            for k in range(len(breaks) - 1):
                mid = 0.5 * (breaks[k] + breaks[k + 1])
                hs = [m.mesh_size for m in mats if m.boundaries[lo] < mid < m.boundaries[hi]]
                out.append(min(hs) if hs else h0)
            return out

        zc, zs = self._axis(zb, slab_sizes(zb, 0, 1), h0)
        rc, rs = self._axis(rb, slab_sizes(rb, 2, 3), h0)
        nz, nr = len(zc) - 1, len(rc) - 1
        dz, dr = np.diff(zc), np.diff(rc)

        # material / allowed-level maps, filled per (z-slab, r-slab) rectangle
        lmax = 0
        blocks = []
        for a in range(len(zb) - 1):
            zm = 0.5 * (zb[a] + zb[a + 1])
            for b in range(len(rb) - 1):
                rm = 0.5 * (rb[b] + rb[b + 1])
                owner = [k for k, m in enumerate(mats) if m.boundaries[0] < zm < m.boundaries[1]
                         and m.boundaries[2] < rm < m.boundaries[3]]
                if len(owner) > 1:
                    raise MeshError(f"materials {[mats[k].name for k in owner]} overlap")
                if not owner:
                    continue
                k = owner[0]
                cell = max(dz[zs[a]:zs[a + 1]].max(), dr[rs[b]:rs[b + 1]].max())
                lev = int(math.floor(math.log2(mats[k].mesh_size / cell) + 1e-9))
                lev = max(0, min(lev, self.MAX_LEVEL))
                lmax = max(lmax, lev)
                blocks.append((a, b, k, lev))
        pad = 1 << lmax
        nzp, nrp = _roundup(nz, pad), _roundup(nr, pad)
        mat = np.full((nzp, nrp), -1, dtype=np.int8)
        allowed = np.zeros((nzp, nrp), dtype=np.int8)
        for a, b, k, lev in blocks:
            mat[zs[a]:zs[a + 1], rs[b]:rs[b + 1]] = k
            allowed[zs[a]:zs[a + 1], rs[b]:rs[b + 1]] = lev

        # ---- quadtree: initial levels, 2:1 balance, leaves.  Dense passes over the base grid (98 M cells for a
        # 1 M-node mesh): native (libheatflow_host.so) when available, else the numpy statement of the same steps.
        from . import hostlib

        if use_native is None:
            use_native = hostlib.available()
        if use_native:
            i0, j0, lev = hostlib.quadtree_leaves(hostlib.quadtree_levels(mat, allowed, lmax), lmax)
        else:
            i0, j0, lev = self._quadtree_numpy(mat, allowed, lmax, nzp, nrp)
        if use_native and max(nzp, nrp) <= hostlib.MESH_LATTICE_MAX:
            # nodes, hanging-node fans, triangles, Morton numbering: native too (heatflow_host.h); the numpy statement below
            # gives the same arrays bit for bit (tests/test_mesh_geometry.py) and serves lattices beyond 16-bit indices
            try:
                coords, node_ij, tris, tags, n_fan = hostlib.mesh_from_leaves(i0, j0, lev, mat, zc, rc)
            except RuntimeError as e:
                raise MeshError(str(e)) from e
            del mat, allowed
            return self._finish(coords, node_ij, tris, tags, len(i0), n_fan, nz, nr, h0, lmax, verbose)
        size = np.int64(1) << lev
        i1, j1 = i0 + size, j0 + size
        leaf_mat = mat[i0, j0].astype(np.int32)
        del mat, allowed

        stride = np.int64(nrp + 1)

        def key(i, j):
            return i * stride + j

        corner_keys = np.concatenate([key(i0, j0), key(i1, j0), key(i1, j1), key(i0, j1)])
        node_keys = np.unique(corner_keys)

        def present(k):
            pos = np.searchsorted(node_keys, k)
            pos[pos >= len(node_keys)] = 0
            return node_keys[pos] == k

        half = size >> 1
        im, jm = i0 + half, j0 + half
        can_hang = lev >= 1
        hb = can_hang & present(key(im, j0))   # bottom edge (r = r0), mid in z
        hr = can_hang & present(key(i1, jm))   # z = z1 edge
        ht = can_hang & present(key(im, j1))
        hl = can_hang & present(key(i0, jm))
        fan = hb | hr | ht | hl
        centre_keys = key(im[fan], jm[fan])
        all_keys = np.concatenate([node_keys, centre_keys])  # centres are never corners of a leaf
        ki = all_keys // stride
        kj = all_keys % stride
        order = np.argsort(_morton(ki, kj), kind="stable")
        sorted_keys = all_keys[order]
        # key -> new node id
        ks = np.argsort(sorted_keys, kind="stable")
        keys_lookup = sorted_keys[ks]

        def nid(k):
            return ks[np.searchsorted(keys_lookup, k)].astype(np.int64)

        n_nodes = len(sorted_keys)
        coords = np.empty((n_nodes, 2), dtype=np.float64)
        coords[:, 0] = zc[np.minimum(ki[order], nz)]
        coords[:, 1] = rc[np.minimum(kj[order], nr)]
        node_ij = np.stack([ki[order], kj[order]], axis=1)

        c00, c10, c11, c01 = nid(key(i0, j0)), nid(key(i1, j0)), nid(key(i1, j1)), nid(key(i0, j1))
        tri_parts, mat_parts, cell_parts = [], [], []
        plain = ~fan
        cell_id = np.arange(len(i0), dtype=np.int64)
        # two right triangles sharing the (c00, c11) diagonal
        tri_parts.append(np.stack([c00[plain], c10[plain], c11[plain]], axis=1))
        tri_parts.append(np.stack([c00[plain], c11[plain], c01[plain]], axis=1))
        mat_parts += [leaf_mat[plain], leaf_mat[plain]]
        cell_parts += [cell_id[plain], cell_id[plain]]
        if fan.any():
            cc = nid(key(im[fan], jm[fan]))
            fm = leaf_mat[fan]
            fid = cell_id[fan]
            sides = [
                (c00[fan], c10[fan], hb[fan], key(im[fan], j0[fan])),
                (c10[fan], c11[fan], hr[fan], key(i1[fan], jm[fan])),
                (c11[fan], c01[fan], ht[fan], key(im[fan], j1[fan])),
                (c01[fan], c00[fan], hl[fan], key(i0[fan], jm[fan])),
            ]
            for a, b, hang, mkey in sides:
                nh = ~hang
                tri_parts.append(np.stack([a[nh], b[nh], cc[nh]], axis=1))
                mat_parts.append(fm[nh])
                cell_parts.append(fid[nh])
                if hang.any():
                    mid = nid(mkey[hang])
                    tri_parts.append(np.stack([a[hang], mid, cc[hang]], axis=1))
                    tri_parts.append(np.stack([mid, b[hang], cc[hang]], axis=1))
                    mat_parts += [fm[hang], fm[hang]]
                    cell_parts += [fid[hang], fid[hang]]
        tris = np.concatenate(tri_parts)
        tmat = np.concatenate(mat_parts)
        tcell = np.concatenate(cell_parts)

        # CCW orientation in the (z, r) plane
        p0, p1, p2 = coords[tris[:, 0]], coords[tris[:, 1]], coords[tris[:, 2]]
        area2 = (p1[:, 0] - p0[:, 0]) * (p2[:, 1] - p0[:, 1]) - (p2[:, 0] - p0[:, 0]) * (p1[:, 1] - p0[:, 1])
        if (area2 == 0).any():
            raise MeshError("degenerate triangle generated")
        flip = area2 < 0
        tris[flip, 1], tris[flip, 2] = tris[flip, 2].copy(), tris[flip, 1].copy()

        # element order: Morton code of the owning leaf cell, then the (stable) template order
        eorder = np.argsort(_morton(i0[tcell] + (size[tcell] >> 1), j0[tcell] + (size[tcell] >> 1)), kind="stable")
        # cell tag = gmsh surface id: 1-based in material list order (reference mesh.py:114)
        return self._finish(coords, node_ij.astype(np.int64), tris[eorder].astype(np.int32), (tmat[eorder] + 1).astype(np.int32),
                            len(i0), int(fan.sum()), nz, nr, h0, lmax, verbose)

    def _finish(self, coords, node_ij, tris, tags, n_leaves, n_fan, nz, nr, h0, lmax, verbose):
        self.coords = coords
        self.node_ij = node_ij
        self.tris = tris
        self.tags = tags
        for k, m in enumerate(self.materials):
            m._tag = k + 1
            m.tag = k + 1
            self.material_tags[m.name] = k + 1
        self.stats = {
            "n_nodes": int(len(coords)), "n_tris": int(len(tris)), "n_leaves": int(n_leaves),
            "base_grid": (int(nz), int(nr)), "h0": float(h0), "max_level": int(lmax),
            "n_fan_cells": int(n_fan),
        }
        if verbose:
            print(f"mesh: {len(coords)} nodes, {len(tris)} triangles, base grid {nz}x{nr}, levels 0..{lmax}")
        return self

    # -- I/O ----------------------------------------------------------------------------
    def write(self, filename: str, version: str = "2.2"):
        """Write the mesh.  ``*.msh`` -> Gmsh ASCII, MSH 2.2 (default, native writer) or MSH 4.1 (the
        format the reference's ``gmsh.write`` produces, mesh.py:191-195); both are readable by gmsh and
        by :func:`read_msh`.  Beside a 2.2 file, which lists nodes and triangles in this mesh's own order, a binary
        ``.npz`` sidecar is written for fast reload (:func:`load_mesh_arrays`); a 4.1 file groups its elements by
        surface, so no sidecar goes with it (a stale one is removed): every reload of that file goes through
        :func:`read_msh` and sees one element numbering."""
        if self.coords is None:
            raise RuntimeError("Mesh not built - call build_mesh() first.")
        side = os.path.splitext(filename)[0] + ".npz"
        if str(version).startswith("4"):
            write_msh41(filename, self.coords, self.tris, self.tags, self.material_tags)
            if os.path.isfile(side):
                os.remove(side)
        else:
            write_msh(filename, self.coords, self.tris, self.tags, self.material_tags)
            np.savez(side, coords=self.coords, tris=self.tris, tags=self.tags)


def write_msh(filename, coords, tris, tags, names=None):
    """Gmsh MSH 2.2 ASCII: nodes (x=z, y=r, z=0) and 3-node triangles (type 2) whose
    first tag (physical group) and second tag (surface id) are the material tag."""
    from . import hostlib

    if hostlib.available():        # native writer (libheatflow_host.so): ~20x faster than numpy.savetxt
        hostlib.write_msh22(filename, coords, tris, tags, names)
        return
    n, ne = len(coords), len(tris)
    with open(filename, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n")
        if names:
            f.write(f"$PhysicalNames\n{len(names)}\n")
            for nm, t in sorted(names.items(), key=lambda kv: kv[1]):
                f.write(f'2 {t} "{nm}"\n')
            f.write("$EndPhysicalNames\n")
        f.write(f"$Nodes\n{n}\n")
        ids = np.arange(1, n + 1)
        np.savetxt(f, np.column_stack([ids, coords[:, 0], coords[:, 1], np.zeros(n)]), fmt="%d %.17g %.17g %.17g")
        f.write(f"$EndNodes\n$Elements\n{ne}\n")
        eid = np.arange(1, ne + 1)
        rows = np.column_stack([eid, np.full(ne, 2), np.full(ne, 2), tags, tags, tris + 1])
        np.savetxt(f, rows, fmt="%d")
        f.write("$EndElements\n")


def write_msh41(filename, coords, tris, tags, names=None, surface_ids=None):
    """Gmsh MSH 4.1 ASCII as ``gmsh.write`` lays it out for the reference's meshes (one plane surface per
    material, one physical group per surface, mesh.py:114-126): $PhysicalNames, $Entities (surfaces only,
    each with its bounding box and its physical tag), $Nodes and $Elements in one block per surface.
    ``tags`` are the physical tags (= cell tags); ``surface_ids`` {physical tag: surface entity id} lets a
    caller separate the two numberings (default: the same number).  Every node is listed once, in the block
    of the lowest-numbered surface that uses it (gmsh files interface nodes under curve / point entities,
    which carry no cells; readers only need each node exactly once)."""
    coords = np.asarray(coords, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64)
    tags = np.asarray(tags, dtype=np.int64)
    phys = [int(t) for t in np.unique(tags)]
    sid = {t: int(surface_ids[t]) if surface_ids else t for t in phys}
    order = sorted(phys, key=lambda t: sid[t])
    n, ne = len(coords), len(tris)
    owner = np.full(n, -1, dtype=np.int64)
    for t in reversed(order):                                  # the lowest surface id wins
        owner[np.unique(tris[tags == t])] = sid[t]
    if (owner < 0).any():
        raise MeshError("write_msh41: a node belongs to no triangle")
    with open(filename, "w") as f:
        f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
        if names:
            inv = {int(v): k for k, v in names.items()}
            f.write(f"$PhysicalNames\n{len(phys)}\n")
            for t in phys:
                f.write(f'2 {t} "{inv.get(t, "surface_%d" % t)}"\n')
            f.write("$EndPhysicalNames\n")
        f.write(f"$Entities\n0 0 {len(order)} 0\n")
        for t in order:
            p = coords[np.unique(tris[tags == t])]
            lo, hi = p.min(axis=0), p.max(axis=0)
            f.write(f"{sid[t]} {float(lo[0])!r} {float(lo[1])!r} 0 {float(hi[0])!r} {float(hi[1])!r} 0 1 {t} 0\n")
        f.write("$EndEntities\n")
        f.write(f"$Nodes\n{len(order)} {n} 1 {n}\n")
        for t in order:
            idx = np.nonzero(owner == sid[t])[0]
            f.write(f"2 {sid[t]} 0 {len(idx)}\n")
            if len(idx):
                np.savetxt(f, idx + 1, fmt="%d")
                np.savetxt(f, np.column_stack([coords[idx], np.zeros(len(idx))]), fmt="%.17g %.17g %.17g")
        f.write("$EndNodes\n")
        f.write(f"$Elements\n{len(order)} {ne} 1 {ne}\n")
        at = 1
        for t in order:
            sel = np.nonzero(tags == t)[0]
            f.write(f"2 {sid[t]} 2 {len(sel)}\n")
            np.savetxt(f, np.column_stack([np.arange(at, at + len(sel)), tris[sel] + 1]), fmt="%d")
            at += len(sel)
        f.write("$EndElements\n")


def read_msh(filename):
    """Read a Gmsh ASCII mesh: MSH 2.2 (what :func:`write_msh` writes) or MSH 4.1 (what the
    reference's ``gmsh.write`` produces by default).  Returns (coords (n,2), tris (ne,3) int32,
    tags (ne,) int32).  The cell tag is the triangle's *physical group* (the tag
    ``gmshio.model_to_mesh`` hands to dolfinx, run_with_diamond.py:244), falling back to the
    surface id when a surface has no physical group.  Points / lines are skipped."""
    with open(filename) as f:
        lines = f.read().split("\n")
    pos = {ln.strip(): k for k, ln in enumerate(lines) if ln.startswith("$")}
    if "$MeshFormat" not in pos:
        raise MeshError(f"{filename}: not a Gmsh mesh file")
    fmt = lines[pos["$MeshFormat"] + 1].split()
    if len(fmt) >= 2 and fmt[1] != "0":
        raise MeshError(f"{filename}: binary MSH files are not supported, export ASCII")
    if fmt and fmt[0].startswith("2"):
        return _read_msh2(lines, pos)
    if fmt and fmt[0].startswith("4.1"):
        return _read_msh41(lines, pos)
    raise MeshError(f"{filename}: unsupported MSH version {fmt[:1]} (2.2 and 4.1 ASCII are read)")


def _numbers(block, dtype=np.float64):
    """All whitespace-separated numbers of a list of lines, parsed by numpy's C tokenizer."""
    return np.fromstring(" ".join(block), dtype=dtype, sep=" ")


def _read_msh2(lines, pos):
    k = pos["$Nodes"]
    n = int(lines[k + 1])
    nodes = _numbers(lines[k + 2:k + 2 + n])
    if nodes.size != 4 * n:
        raise MeshError("MSH 2.2 $Nodes: expected 4 numbers per line")
    nodes = nodes.reshape(n, 4)
    ids = nodes[:, 0].astype(np.int64)
    remap = np.full(ids.max() + 1, -1, dtype=np.int64)
    remap[ids] = np.arange(n)
    k = pos["$Elements"]
    ne = int(lines[k + 1])
    block = lines[k + 2:k + 2 + ne]
    # gmsh stores points and lines before the triangles; skip them, then try the uniform fast path
    first = 0
    while first < ne and int(block[first].split()[1]) != 2:
        first += 1
    tri_rows = None
    if first < ne:
        ncol = len(block[first].split())
        flat = _numbers(block[first:], dtype=np.int64)
        if flat.size == ncol * (ne - first):
            rows = flat.reshape(ne - first, ncol)
            if (rows[:, 1] == 2).all() and (rows[:, 2] == rows[0, 2]).all():
                tri_rows = rows
    if tri_rows is not None:
        ntag = int(tri_rows[0, 2])
        tags = tri_rows[:, 3] if ntag else np.zeros(len(tri_rows), dtype=np.int64)
        tris = tri_rows[:, 3 + ntag:3 + ntag + 3]
    else:                                          # mixed element types / tag counts: line by line
        tris, tags = [], []
        for ln in block:
            p = ln.split()
            if int(p[1]) != 2:
                continue
            ntag = int(p[2])
            tags.append(int(p[3]) if ntag else 0)
            tris.append([int(v) for v in p[3 + ntag:3 + ntag + 3]])
        tris = np.array(tris, dtype=np.int64).reshape(-1, 3)
    if len(tris) == 0:
        raise MeshError("no 3-node triangles in the MSH 2.2 file")
    tris = remap[np.asarray(tris, dtype=np.int64)].astype(np.int32)
    return nodes[:, 1:3].copy(), tris, np.asarray(tags, dtype=np.int32)


def _read_msh41(lines, pos):
    # surface entity -> physical tag
    surf_phys = {}
    if "$Entities" in pos:
        k = pos["$Entities"] + 1
        npnt, ncur, nsur, _nvol = (int(v) for v in lines[k].split())
        k += 1 + npnt + ncur
        for ln in lines[k:k + nsur]:
            p = ln.split()
            nphys = int(p[7])
            if nphys:
                surf_phys[int(p[0])] = abs(int(p[8]))
    k = pos["$Nodes"] + 1
    nblocks, nnodes = (int(v) for v in lines[k].split()[:2])
    k += 1
    ids = np.empty(nnodes, dtype=np.int64)
    xyz = np.empty((nnodes, 3), dtype=np.float64)
    at = 0
    for _ in range(nblocks):
        _dim, _tag, parametric, nb = (int(v) for v in lines[k].split())
        k += 1
        if nb:
            ids[at:at + nb] = _numbers(lines[k:k + nb], dtype=np.int64)
            k += nb
            vals = _numbers(lines[k:k + nb])
            if vals.size % nb:
                raise MeshError("MSH 4.1 $Nodes: ragged coordinate block")
            xyz[at:at + nb] = vals.reshape(nb, -1)[:, :3]     # parametric blocks carry extra columns
            k += nb
        at += nb
    remap = np.full(ids.max() + 1, -1, dtype=np.int64)
    remap[ids] = np.arange(nnodes)
    k = pos["$Elements"] + 1
    nblocks = int(lines[k].split()[0])
    k += 1
    tris, tags = [], []
    for _ in range(nblocks):
        dim, etag, etype, nb = (int(v) for v in lines[k].split())
        k += 1
        if dim == 2 and etype == 2 and nb:
            rows = _numbers(lines[k:k + nb], dtype=np.int64)
            if rows.size != 4 * nb:
                raise MeshError("MSH 4.1 $Elements: a 3-node triangle line must hold 4 integers")
            tris.append(rows.reshape(nb, 4)[:, 1:4])
            tags.append(np.full(nb, surf_phys.get(etag, etag), dtype=np.int64))
        k += nb
    if not tris:
        raise MeshError("no 3-node triangles in the MSH 4.1 file")
    tris = remap[np.concatenate(tris)]
    tags = np.concatenate(tags)
    used = np.unique(tris)                       # gmsh may store nodes that only points/curves use
    compact = np.full(nnodes, -1, dtype=np.int64)
    compact[used] = np.arange(len(used))
    return xyz[used, :2].copy(), compact[tris].astype(np.int32), tags.astype(np.int32)


def reorder_mesh(coords, tris, tags):
    """Renumber nodes along a Morton curve of their (quantised) coordinates and sort the triangles by
    the curve position of their centroids.  The device kernels work on any numbering, but a spatially
    coherent one keeps the SpMV gathers in L2 and the assembly's row blocks compact; meshes from
    gmsh (the reference's ``mesh.msh``) arrive in insertion order.  Returns (coords, tris, tags, perm)
    with ``coords_new = coords[perm]``."""
    coords = np.asarray(coords, dtype=np.float64)
    lo, span = coords.min(axis=0), np.ptp(coords, axis=0)
    span[span == 0] = 1.0
    q = np.minimum(((coords - lo) / span * 65535.0).astype(np.int64), 65535)
    perm = np.argsort(_morton(q[:, 0], q[:, 1]), kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    tris_new = inv[np.asarray(tris, dtype=np.int64)]
    cq = q[perm][tris_new].mean(axis=1).astype(np.int64)
    eorder = np.argsort(_morton(cq[:, 0], cq[:, 1]), kind="stable")
    return coords[perm], tris_new[eorder].astype(np.int32), np.asarray(tags)[eorder].astype(np.int32), perm


def load_mesh_arrays(mesh_file_path, reorder_external=True):
    """(coords, tris, tags) from ``mesh.msh`` - through the ``.npz`` sidecar when it is
    at least as new as the ``.msh`` (a mesh written by this package, already Morton-ordered);
    a bare ``.msh`` (e.g. generated by gmsh for the reference) is read and renumbered."""
    side = os.path.splitext(mesh_file_path)[0] + ".npz"
    if os.path.isfile(side) and os.path.getmtime(side) >= os.path.getmtime(mesh_file_path) - 1.0:
        with np.load(side) as d:
            return d["coords"], d["tris"], d["tags"]
    coords, tris, tags = read_msh(mesh_file_path)
    if reorder_external:
        coords, tris, tags, _ = reorder_mesh(coords, tris, tags)
    return coords, tris, tags
