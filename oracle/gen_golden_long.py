"""Generate tests/golden/run10.npz from the flang-compiled REFERENCE: the state after 10 simulated days (360 steps) from
1982-01-01 with the example boundary conditions and zero SST anomaly -- the horizon over which SURVEY.md section 8c found no
error amplification between two builds of the reference itself.  TEST INFRASTRUCTURE.

Run in the build container:  python oracle/gen_golden_long.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refmodel as R  # noqa: E402


def main():
    bc = np.load(os.path.join(HERE, "..", "pyspeedy_amd", "data", "example_bc.npz"))
    m = R.RefModel(start=(1982, 1, 1, 0, 0), end=(1982, 1, 12, 0, 0))
    m.set_bc(bc)
    for _ in range(360):
        assert m.step() == 0
    out = {}
    for v in ("vor", "div", "t", "ps"):
        out[v] = m.get(v)[..., 0]
    out["tr"] = m.get("tr")[..., 0, 0]
    for v in ("land_temp", "sst_am", "tice_am", "snowc", "olr", "precnv", "precls", "tsr"):
        out[v] = m.get(v)
    dst = os.path.join(HERE, "..", "tests", "golden", "run10.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst)


if __name__ == "__main__":
    main()
