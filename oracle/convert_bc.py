"""Convert the reference's boundary-condition file to a small .npz fixture.

TEST INFRASTRUCTURE (oracle side).  Run once, in the build container, with the
conda interpreter (the only one that has h5py):

    /opt/conda/bin/python3.9 oracle/convert_bc.py

Input : /root/reference/pyspeedy/data/example_bc.nc   (HDF5 / NetCDF4, data only)
Output: pyspeedy_amd/data/example_bc.npz  -- the 12 fields read by the reference's
        Speedy.set_bc (pyspeedy/speedy.py:277-296), float32, dims (lon, lat[, month]),
        latitude south -> north, exactly as stored.
"""
import os
import sys

import h5py
import numpy as np

SRC = "/root/reference/pyspeedy/data/example_bc.nc"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyspeedy_amd", "data", "example_bc.npz")

FIELDS = ["alb", "icec", "lsm", "orog", "snowd", "sst", "stl", "swl1", "swl2", "swl3", "vegh", "vegl"]


def main():
    out = {}
    with h5py.File(SRC, "r") as f:
        for k in FIELDS:
            a = np.asarray(f[k][...])
            assert a.dtype == np.float32, (k, a.dtype)
            out[k] = a
            print(k, a.shape, a.dtype, float(np.nanmin(a)), float(np.nanmax(a)))
        for k in ("lon", "lat"):
            if k in f:
                out[k] = np.asarray(f[k][...])
    np.savez_compressed(DST, **out)
    print("wrote", DST, os.path.getsize(DST), "bytes")


if __name__ == "__main__":
    sys.exit(main())
