"""Convert a boundary-condition (or SST-anomaly) file of the reference -- NetCDF-4 / HDF5 -- into a file pyspeedy_amd reads without
any extra package: .npz (default) or NetCDF-3 classic.

    python tools/convert_bc.py [SRC [DST]]       with an interpreter that has h5py, netCDF4 or xarray
    /opt/conda/bin/python3.9 tools/convert_bc.py   (the build container: converts the reference's example_bc.nc into the packaged
                                                    pyspeedy_amd/data/example_bc.npz)

`Speedy.set_bc(bc_file=...)` reads HDF5 files directly when one of those packages is importable (pyspeedy_amd/speedy.py:
_read_hdf5); this tool is for environments where none is -- the default test image here -- and is what the error message of that
reader names.  Product data preparation, not test infrastructure: every variable of the file is copied as stored (the 12 fields
the reference's Speedy.set_bc reads, pyspeedy/speedy.py:277-296, are float32 with dims (lon, lat[, month]), latitude south ->
north; an anomaly file carries `ssta` (lon, lat, time) and `time`).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/pyspeedy/data/example_bc.nc"
DST = os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz")


def main(argv):
    src = argv[0] if argv else SRC
    dst = argv[1] if len(argv) > 1 else (DST if not argv else os.path.splitext(src)[0] + ".npz")
    sys.path.insert(0, ROOT)
    from pyspeedy_amd.speedy import _read_hdf5
    fields = _read_hdf5(src)
    for k, a in fields.items():
        print(k, a.shape, a.dtype)
    if dst.endswith(".npz"):
        np.savez_compressed(dst, **fields)
    else:
        from pyspeedy_amd.dataset import Dataset, Variable
        dims = {"time": ("time",), "lon": ("lon",), "lat": ("lat",)}
        data = {k: Variable(dims.get(k, ("lon", "lat", "month" if k != "ssta" else "time")[:a.ndim]), a) for k, a in fields.items()}
        Dataset({k: v for k, v in data.items() if k not in dims}, {k: v for k, v in data.items() if k in dims}).to_netcdf(dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
