"""Drop-in name for the reference's run_with_diamond module: ``import run_with_diamond as run``."""
from heatflow_amd.run_with_diamond import cli, run_simulation, suppress_output  # noqa: F401

if __name__ == "__main__":
    raise SystemExit(cli("with_diamond"))
