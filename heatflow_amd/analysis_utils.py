"""RMSE between an experimental curve and a simulated one (reference analysis_utils.py:66-93):
the simulation is interpolated onto the experimental time base with np.interp."""
import numpy as np


def calculate_rmse(exp_time, exp_data, sim_time, sim_data):
    exp_time, exp_data = np.asarray(exp_time, dtype=float), np.asarray(exp_data, dtype=float)
    sim_on_exp = np.interp(exp_time, np.asarray(sim_time, dtype=float), np.asarray(sim_data, dtype=float))
    return float(np.sqrt(np.mean((exp_data - sim_on_exp) ** 2)))
