"""Drop-in name for the reference's run_no_diamond module: ``from run_no_diamond import run_simulation``."""
from heatflow_amd.run_no_diamond import cli, run_simulation, suppress_output  # noqa: F401

if __name__ == "__main__":
    raise SystemExit(cli("no_diamond"))
