"""Drop-in name for the reference's parameter_sweep module and command line (parameter_sweep.py:289-600):

    python parameter_sweep.py --config cfgs/geballe_no_diamond.yaml --output-dir outputs/sweep --fwhm-range A B \\
        --k-range A B --width-range A B --num-points NF NK NW [--mesh-folder meshes] [--batch 8]

One rank per GPU under torch.distributed.run (points i mod world); see heatflow_amd/parameter_sweep.py."""
from heatflow_amd.parameter_sweep import *  # noqa: F401,F403
from heatflow_amd.parameter_sweep import main  # noqa: F401

if __name__ == "__main__":
    raise SystemExit(main())
