"""Experiment script for cfgs/geballe_with_diamond.yaml (role of the reference's with_diamond.py:
load the config, put the watchers in the coupler mid-planes at r = 0, run, and report the
normalised RMSE against the experimental curves).  Plots are out of scope.

    python with_diamond.py [--scale S] [--device D]
"""
import argparse
import os

import numpy as np
import yaml

import run_with_diamond as run
from heatflow_amd.analysis_utils import calculate_rmse
from heatflow_amd.geometry import scale_mesh_sizes, watcher_points

sim_name = "geballe_with_diamond"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="factor on every mats.*.mesh")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "cfgs", f"{sim_name}.yaml")) as f:
        cfg = scale_mesh_sizes(yaml.safe_load(f), a.scale)
    wp = watcher_points(cfg)
    res = run.run_simulation(cfg=cfg, mesh_folder=f"meshes/{sim_name}", rebuild_mesh=True, visualize_mesh=False,
                             output_folder=f"outputs/{sim_name}", watcher_points=wp, write_xdmf=False,
                             suppress_print=False, device_id=a.device)
    print(f"Simulation completed! Check outputs/{sim_name}/ for results.")
    exp = np.genfromtxt(os.path.join(here, cfg["heating"]["file"]), delimiter=",", names=True)
    ps, os_ = res["watchers"]["pside"], res["watchers"]["oside"]
    span = ps.max() - ps.min()
    sim_p, sim_o = (ps - ps[0]) / span, (os_ - os_[0]) / span            # with_diamond.py:63-70 normalisation
    exp_span = exp["temp"].max() - exp["temp"].min()
    exp_p = (exp["temp"] - exp["temp"][0]) / exp_span
    exp_o = (exp["oside"] - exp["oside"][0]) / exp_span
    print(f"pside RMSE: {calculate_rmse(exp['time'], exp_p, res['times'], sim_p):.6f}")
    print(f"oside RMSE: {calculate_rmse(exp['time'], exp_o, res['times'], sim_o):.6f}")
    return res


if __name__ == "__main__":
    main()
