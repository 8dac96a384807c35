"""1-D model on the r = 0 line of a 2-D mesh (reference run_no_diamond_1d.py; BASELINE config C1).

BASELINE.json lists this configuration as "plumbing, ~1k DOF, no GPU": it is a CPU model by
definition (a few hundred unknowns, tridiagonal), so it runs on the host here as well - through
a banded LU (scipy.linalg.solve_banded semantics via the Thomas algorithm below), not through
libheatflow_hip.so.  It mirrors ``run_1d`` (:166): same signature, mesh cache check (:203-213),
axis submesh (:30-164: edges with both vertices at |r| <= 1e-10, 1-D cell tag = tag of the first
2-D cell that owns the edge), un-weighted forms (:537-546), Dirichlet rows left / right = ic_temp
and heating_offset(t) at z = mesh_zmin + z_ins_pside (:563-591, no Gaussian in 1-D), assemble
once, loop (:712-790), outputs ``used_config.yaml`` / ``watcher_points.csv`` in
``sim_outputs/1d_simulation`` by default.

The radial-loss source term (``use_radial_correction``) follows
:316-378 (CSV discovery, bilinear (t, z) interpolation of dT/dr with clamping), :469-480 (delta_r =
0.1 um for ``radial_gradient.csv``, 0.07 um for the raw file), :676-700 (kappa per node) and
:718-747 (per-step source 2 kappa (dT/dr)/delta_r, x0.1 on nodes whose z was clamped), entering
the right-hand side as dt * int s v dx (:546).  One reference quirk is kept selectable: it looks
kappa up as ``kappa_per_cell[cell_tags_1d.values[cell_idx]]`` (:692), i.e. it indexes the
per-cell array by a *tag value*; ``kappa_lookup="reference"`` (default, for identical numbers)
does the same, ``kappa_lookup="cell"`` uses the conductivity of the node's own cell.
"""
from __future__ import annotations

import os
import time

import numpy as np
import yaml

from .bc import P1Space, RowDirichletBC, gather_bc_values, merge_bcs
from .driver import _parse_watchers, _resolve, suppress_output, write_watcher_csv
from .heating import HeatingCurve
from .mesh import load_mesh_arrays


def extract_1d_submesh_from_2d(coords, tris, tags, tolerance=1e-10):
    """Nodes and intervals of the r = 0 line.  Returns (z sorted, cell tags of the intervals,
    node ids in the 2-D mesh).  An interval exists where a 2-D edge has both vertices on the axis;
    its tag is the tag of the lowest-numbered 2-D cell containing that edge (reference :124-138)."""
    on_axis = np.abs(coords[:, 1]) <= tolerance
    edges, owner = {}, {}
    for c in np.nonzero(on_axis[tris].sum(axis=1) >= 2)[0]:
        t = tris[c]
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            if on_axis[a] and on_axis[b]:
                key = (min(a, b), max(a, b))
                if key not in owner:
                    owner[key] = c
    if not owner:
        raise ValueError("No facets found on the r=0 axis. Check tolerance or mesh.")
    nodes = np.array(sorted({v for e in owner for v in e}, key=lambda v: coords[v, 0]), dtype=np.int64)
    pos = {int(v): k for k, v in enumerate(nodes)}
    cell_tags = np.zeros(len(nodes) - 1, dtype=np.int32)
    seen = np.zeros(len(nodes) - 1, dtype=bool)
    for (a, b), c in owner.items():
        k = min(pos[int(a)], pos[int(b)])
        if abs(pos[int(a)] - pos[int(b)]) != 1:
            raise ValueError("axis edges do not form a chain")
        cell_tags[k] = tags[c]
        seen[k] = True
    if not seen.all():
        raise ValueError("the r=0 line is not covered by mesh edges")
    return coords[nodes, 0].copy(), cell_tags, nodes


def _thomas_factor(lower, diag, upper):
    """LU of a tridiagonal matrix (no pivoting: the operator is SPD)."""
    n = len(diag)
    c = np.zeros(n - 1)
    d = np.zeros(n)
    d[0] = diag[0]
    for i in range(1, n):
        c[i - 1] = lower[i - 1] / d[i - 1]
        d[i] = diag[i] - c[i - 1] * upper[i - 1]
    return c, d, upper


def _thomas_solve(fac, b):
    c, d, upper = fac
    y = b.copy()
    for i in range(1, len(y)):
        y[i] -= c[i - 1] * y[i - 1]
    y[-1] /= d[-1]
    for i in range(len(y) - 2, -1, -1):
        y[i] = (y[i] - upper[i] * y[i + 1]) / d[i]
    return y


def run_1d(cfg, mesh_folder_2d, mesh_folder_1d=None, rebuild_mesh=False, visualize_mesh=False, output_folder=None,
           watcher_points=None, write_xdmf=True, suppress_print=False, use_radial_correction=True,
           radial_gradient_path=None, *, kappa_lookup="reference"):
    with suppress_output(suppress_print):
        t_start = time.time()
        mesh_cfg_path = os.path.join(mesh_folder_2d, "mesh_cfg.yaml")
        mesh_file_path = os.path.join(mesh_folder_2d, "mesh.msh")
        missing = [nm for nm, p in (("mesh.msh", mesh_file_path), ("mesh_cfg.yaml", mesh_cfg_path)) if not os.path.isfile(p)]
        if missing:
            raise FileNotFoundError(f"Missing required file(s) in {mesh_folder_2d}: {', '.join(missing)}")
        print("Radial heating correction: %s (user choice)" % ("ENABLED" if use_radial_correction else "DISABLED"))
        with open(mesh_cfg_path) as f:
            mat_tag_map = yaml.safe_load(f).get("material_tags", {})
        coords, tris, tags = load_mesh_arrays(mesh_file_path)
        print("Loaded 2D mesh successfully")
        z, cell_tags, nodes_2d = extract_1d_submesh_from_2d(coords, tris, tags)
        n = len(z)

        names = ["p_ins", "p_coupler", "p_sample", "o_coupler", "o_ins"]
        tag_to_k = {mat_tag_map[m]: float(cfg["mats"][m]["k"]) for m in names}
        tag_to_rc = {mat_tag_map[m]: float(cfg["mats"][m]["rho"]) * float(cfg["mats"][m]["cv"]) for m in names}
        kappa = np.array([tag_to_k[int(t)] for t in cell_tags])
        rho_c = np.array([tag_to_rc[int(t)] for t in cell_tags])

        t_final = float(cfg["timing"]["t_final"])
        num_steps = int(cfg["timing"]["num_steps"])
        dt = t_final / num_steps
        ic_temp = float(cfg["heating"]["ic_temp"])
        heat = HeatingCurve(_resolve(cfg["heating"]["file"]), ic_temp, float(cfg["heating"]["fwhm"]))

        # P1 intervals, no r weight: M_e = rho_c h/6 [[2,1],[1,2]], K_e = kappa/h [[1,-1],[-1,1]]
        h = np.diff(z)
        m_off = rho_c * h / 6.0
        m_diag = np.zeros(n)
        m_diag[:-1] += rho_c * h / 3.0
        m_diag[1:] += rho_c * h / 3.0
        k_off = -kappa / h
        k_diag = np.zeros(n)
        k_diag[:-1] += kappa / h
        k_diag[1:] += kappa / h
        a_off = m_off + dt * k_off
        a_diag = m_diag + dt * k_diag

        z_sample = float(cfg["mats"]["p_sample"]["z"])
        z_ins_pside = float(cfg["mats"]["p_ins"]["z"])
        z_coupler = float(cfg["mats"]["p_coupler"]["z"])
        heating_z = -(z_sample / 2) - z_ins_pside - z_coupler + z_ins_pside
        V = P1Space(np.column_stack([z, np.zeros(n)]))
        bcs = [RowDirichletBC(V, "left", value=ic_temp), RowDirichletBC(V, "right", value=ic_temp),
               RowDirichletBC(V, "x", coord=heating_z, value=lambda x, y, t: heat.amplitude(t) + 0.0 * x)]
        bc_dofs, owner, pos = merge_bcs(bcs)
        is_bc = np.zeros(n, dtype=bool)
        is_bc[bc_dofs] = True
        # symmetric elimination with unit diagonal (dolfinx assemble_matrix(form, bcs))
        lo, up, dg = a_off.copy(), a_off.copy(), a_diag.copy()
        dg[is_bc] = 1.0
        kill = is_bc[:-1] | is_bc[1:]
        lo[kill] = 0.0
        up[kill] = 0.0
        fac = _thomas_factor(lo, dg, up)

        if output_folder is not None:
            save_folder = output_folder
            os.makedirs(save_folder, exist_ok=True)
            with open(os.path.join(save_folder, "used_config.yaml"), "w") as f:
                yaml.safe_dump(cfg, f)
        else:
            save_folder = os.path.join(os.getcwd(), "sim_outputs", "1d_simulation")
            os.makedirs(save_folder, exist_ok=True)

        # ---- radial heating correction from a 2-D run's gradient CSV (reference :316-378)
        grad = None
        if use_radial_correction:
            grad_file = radial_gradient_path
            if grad_file is None:
                bases = [os.path.join(mesh_folder_2d, "..", "outputs", "geballe_no_diamond_read_flux"),
                         os.path.join(mesh_folder_2d, "..", "..", "outputs", "geballe_no_diamond_read_flux"),
                         os.path.join(os.getcwd(), "outputs", "geballe_no_diamond_read_flux"),
                         os.path.join(os.getcwd(), "sim_outputs", "geballe_no_diamond_read_flux")]
                for fname in ("radial_gradient.csv", "radial_gradient_raw.csv"):
                    hits = [os.path.join(b, fname) for b in bases if os.path.exists(os.path.join(b, fname))]
                    if hits:
                        grad_file = hits[0]
                        break
            if grad_file is None:
                print("No radial gradient CSV found: radial correction switched off")   # as the reference does
                use_radial_correction = False
            else:
                from scipy.interpolate import RegularGridInterpolator

                with open(grad_file) as f:
                    header = f.readline().strip().split(",")
                table = np.loadtxt(grad_file, delimiter=",", skiprows=1, ndmin=2)
                g_times, g_z, g_vals = table[:, 0], np.array(header[1:], dtype=float), table[:, 1:]
                interp = RegularGridInterpolator((g_times, g_z), g_vals, method="linear")
                delta_r = 0.1e-6 if "radial_gradient.csv" in grad_file else 0.07e-6
                # node -> first cell (in cell order) whose span contains it (:676-686)
                node_cell = np.maximum(np.arange(n) - 1, 0)
                if kappa_lookup == "reference":
                    idx = np.minimum(cell_tags[node_cell], len(kappa) - 1)      # tag value used as a cell index (:692)
                    node_kappa = kappa[idx]
                elif kappa_lookup == "cell":
                    node_kappa = kappa[node_cell]
                else:
                    raise ValueError("kappa_lookup must be 'reference' or 'cell'")
                z_cl = np.clip(z, g_z.min(), g_z.max())
                clamped = z != z_cl
                grad = (interp, g_times.min(), g_times.max(), z_cl, clamped, node_kappa, delta_r)
                # un-weighted unit mass matrix for dt * int s v dx
                s_off = h / 6.0
                s_diag = np.zeros(n)
                s_diag[:-1] += h / 3.0
                s_diag[1:] += h / 3.0

        wnames, wcoords = _parse_watchers(watcher_points)
        wnodes = [int(np.argmin(np.abs(z - c[0]))) for c in wcoords]

        u = np.full(n, ic_temp)
        for bc in bcs:
            bc.update(0.0)
        times, rows, fields = [], [], []
        t_loop = time.time()
        for step in range(num_steps):
            t = (step + 1) * dt
            bcs[2].update(t)
            g = gather_bc_values(bcs, owner, pos)
            b = m_diag * u
            b[:-1] += m_off * u[1:]
            b[1:] += m_off * u[:-1]
            if grad is not None:
                interp, t_lo, t_hi, z_cl, clamped, node_kappa, delta_r = grad
                gv = interp(np.column_stack([np.full(n, np.clip(t, t_lo, t_hi)), z_cl]))
                gv[clamped] *= 0.1
                src = 2.0 * node_kappa * gv / delta_r
                b += dt * s_diag * src
                b[:-1] += dt * s_off * src[1:]
                b[1:] += dt * s_off * src[:-1]
            # lifting with the unconstrained operator, then set_bc
            gfull = np.zeros(n)
            gfull[bc_dofs] = g
            lift = a_diag * gfull
            lift[:-1] += a_off * gfull[1:]
            lift[1:] += a_off * gfull[:-1]
            b -= lift
            b[bc_dofs] = g
            u = _thomas_solve(fac, b)
            times.append(t)
            rows.append(u[wnodes].copy() if wnodes else np.zeros(0))
            if write_xdmf:
                fields.append(u.copy())
        loop_time = time.time() - t_loop
        if write_xdmf:  # stand-in for the XDMF series
            np.savez(os.path.join(save_folder, "output_1d.npz"), z=z, times=np.array(times), fields=np.array(fields))
        samples = np.array(rows)
        if watcher_points is not None:
            write_watcher_csv(os.path.join(save_folder, "watcher_points.csv"), times, wnames,
                              {nm: samples[:, k] for k, nm in enumerate(wnames)})
        print("\n--- Timing Summary ---")
        print(f"Total time: {time.time() - t_start:.2f} s")
        print(f"Loop time: {loop_time:.2f} s")
        print(f"Average time per step: {loop_time / max(num_steps, 1):.4f} s")
        print("----------------------\n")
        return {"z": z, "cell_tags": cell_tags, "nodes_2d": nodes_2d, "times": np.array(times), "u": u,
                "watchers": {nm: samples[:, k] for k, nm in enumerate(wnames)}, "save_folder": save_folder}
