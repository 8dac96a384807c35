"""Drop-in name for the reference's run_no_diamond_1d module (``run_1d``)."""
from heatflow_amd.run_no_diamond_1d import extract_1d_submesh_from_2d, run_1d  # noqa: F401
