"""Experiment script for cfgs/geballe_no_diamond_read_flux.yaml (role of the reference's no_diamond.py: load the
config, put the watchers in the coupler mid-planes at r = 0, run with the read-flux projection, and report the
normalised o-side RMSE against the experimental curve).  Plots are out of scope.

    python no_diamond.py [--scale S] [--device D]
"""
import argparse
import os

import numpy as np
import yaml

import run_no_diamond as run
from heatflow_amd.analysis_utils import calculate_rmse
from heatflow_amd.geometry import scale_mesh_sizes, watcher_points

sim_name = "geballe_no_diamond_read_flux"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="factor on every mats.*.mesh")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "cfgs", f"{sim_name}.yaml")) as f:
        cfg = scale_mesh_sizes(yaml.safe_load(f), a.scale)
    wp = watcher_points(cfg)                               # coupler mid-planes at r = 0 (no_diamond.py:17-38)
    res = run.run_simulation(cfg=cfg, mesh_folder=f"meshes/{sim_name}", rebuild_mesh=True, visualize_mesh=False,
                             output_folder=f"outputs/{sim_name}", watcher_points=wp, write_xdmf=False,
                             suppress_print=False, device_id=a.device)
    print(f"Simulation completed! Check outputs/{sim_name}/ for results (watcher_points.csv, radial_gradient*.csv).")
    exp = np.genfromtxt(os.path.join(here, cfg["heating"]["file"]), delimiter=",", names=True)
    ps, os_ = res["watchers"]["pside"], res["watchers"]["oside"]
    span = ps.max() - ps.min()
    sim_o = (os_ - os_[0]) / span                          # no_diamond.py:66-78 normalisation
    ic = float(cfg["heating"]["ic_temp"])
    exp_o = exp["oside"] - exp["oside"][0] + ic
    exp_o = (exp_o - exp_o[0]) / (exp["temp"].max() - exp["temp"].min())
    print("\n--- RMSE Analysis ---")
    print(f"O-side RMSE: {calculate_rmse(exp['time'], exp_o, res['times'], sim_o):.4f}")
    print("-------------------\n")
    return res


if __name__ == "__main__":
    main()
