"""Experiment script for the 1-D model (role of the reference's no_diamond_1d.py): run_1d on the r = 0 line of a 2-D
no-diamond mesh, with the radial-loss source taken from a 2-D run's radial_gradient.csv when there is one, and the
normalised RMSE of the 1-D watchers against that 2-D run.  Plots are out of scope.

    python no_diamond_1d.py [--from-run outputs/geballe_no_diamond_read_flux] [--no-correction]

`--from-run` names the output folder of a 2-D run (no_diamond.py writes one): its used_config.yaml, watcher_points.csv
and radial_gradient.csv are read; the 2-D mesh is the one that run cached under meshes/.  The reference's script is
wired to one particular folder of an earlier sweep (no_diamond_1d.py:7-42); the folder is an argument here.
"""
import argparse
import os

import numpy as np
import yaml

import run_no_diamond_1d as run
from heatflow_amd.analysis_utils import calculate_rmse
from heatflow_amd.geometry import watcher_points


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--from-run", default="outputs/geballe_no_diamond_read_flux")
    ap.add_argument("--mesh-folder-2d", default=None, help="default: meshes/<name of the run folder>")
    ap.add_argument("--no-correction", action="store_true")
    a = ap.parse_args(argv)
    name = os.path.basename(os.path.normpath(a.from_run))
    with open(os.path.join(a.from_run, "used_config.yaml")) as f:
        cfg = yaml.safe_load(f)
    grad = os.path.join(a.from_run, "radial_gradient.csv")
    use_corr = (not a.no_correction) and os.path.isfile(grad)
    out = f"outputs/{name}_1d"
    run.run_1d(cfg, mesh_folder_2d=a.mesh_folder_2d or f"meshes/{name}", mesh_folder_1d=f"meshes/1d_meshes/{name}",
               rebuild_mesh=True, visualize_mesh=False, output_folder=out, watcher_points=watcher_points(cfg),
               write_xdmf=False, suppress_print=False, use_radial_correction=use_corr,
               radial_gradient_path=grad if use_corr else None)
    print(f"Simulation completed! Check {out}/ for results.")
    w1 = np.genfromtxt(os.path.join(out, "watcher_points.csv"), delimiter=",", names=True)
    w2 = np.genfromtxt(os.path.join(a.from_run, "watcher_points.csv"), delimiter=",", names=True)
    span1, span2 = w1["pside"].max() - w1["pside"].min(), w2["pside"].max() - w2["pside"].min()
    for nm in ("pside", "oside"):                          # no_diamond_1d.py:61-78 normalisation, 2-D run as the reference curve
        s1 = (w1[nm] - w1[nm][0]) / span1
        s2 = (w2[nm] - w2[nm][0]) / span2
        print(f"{nm} RMSE (1-D vs 2-D run): {calculate_rmse(w2['time'], s2, w1['time'], s1):.4f}")


if __name__ == "__main__":
    main()
