"""Helper of tests/test_variants_spawn.py (not a test): steps a 3-member model 40 times and writes member 1's registry to an
.npz -- run in a process of its own because the kernel-variant switches it is started with are read once per process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd.model import SHAPES, EnsembleModel  # noqa: E402


def main():
    bc = np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))
    model = EnsembleModel(pyspeedy_amd.ModSpectral(), 3)
    model.set_bc({k: bc[k] for k in bc.files})
    if os.environ.get("DUMP_STATE_CFG5"):  # BASELINE cfg 5: SPPT on, fp32 column physics (with its fp32 storage)
        model.set_sppt(True, seed=5, first_member_id=0)
        model.set_physics_precision(True)
    model.run(40)
    np.savez(sys.argv[1], **{n: model.get(n, 1) for n in SHAPES})


if __name__ == "__main__":
    main()
