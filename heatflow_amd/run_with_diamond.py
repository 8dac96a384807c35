"""Entry point for the nine-material diamond-anvil-cell stack (reference run_with_diamond.py).

Same call as the reference: ``run_simulation(cfg, mesh_folder, rebuild_mesh=False,
visualize_mesh=False, output_folder=None, watcher_points=None, write_xdmf=True,
suppress_print=False)``.  Keyword-only extras (``device_id``, ``session``, ``rtol`` ...) select
the GPU and let a sweep keep the mesh resident.  Returns the result dict of
:func:`heatflow_amd.driver.run_simulation_impl` (the reference returns None).
"""
from .driver import cli, run_simulation_impl, suppress_output  # noqa: F401


def run_simulation(cfg, mesh_folder, rebuild_mesh=False, visualize_mesh=False, output_folder=None,
                   watcher_points=None, write_xdmf=True, suppress_print=False, **extra):
    return run_simulation_impl("with_diamond", cfg, mesh_folder, rebuild_mesh, visualize_mesh, output_folder,
                               watcher_points, write_xdmf, suppress_print, **extra)


if __name__ == "__main__":
    raise SystemExit(cli("with_diamond"))
