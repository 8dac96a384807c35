"""Entry point for the five-material stack without diamonds/gasket (reference run_no_diamond.py).

Same call as the reference ``run_simulation(cfg, mesh_folder, ...)``; see
:mod:`heatflow_amd.run_with_diamond` for the extras.
"""
from .driver import cli, run_simulation_impl, suppress_output  # noqa: F401


def run_simulation(cfg, mesh_folder, rebuild_mesh=False, visualize_mesh=False, output_folder=None,
                   watcher_points=None, write_xdmf=True, suppress_print=False, **extra):
    return run_simulation_impl("no_diamond", cfg, mesh_folder, rebuild_mesh, visualize_mesh, output_folder,
                               watcher_points, write_xdmf, suppress_print, **extra)


if __name__ == "__main__":
    raise SystemExit(cli("no_diamond"))
