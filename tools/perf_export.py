"""Cost of one daily output of a 64-member ensemble through the facade: the host-side path (fp64 device -> host copies, float32
narrowing, byte order and level reversal in numpy) against the packed path (all of that on the GPU, the file's payload copied into
pinned memory), and the NetCDF-3 write itself.  Usage (GPU box): python tools/perf_export.py [members]"""
import os
import sys
import tempfile
import time
from datetime import datetime

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ens = SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 2))
ens.set_bc()
ens.run()


def timed(fn, n=6):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return sorted(out)[len(out) // 2] * 1e3


with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "day.nc")
    plain = ens.to_dataframe()
    packed = ens.to_dataframe(packed=True)
    print("M = %d, one day's output (%.1f MB as float32)" % (M, sum(v.values.nbytes for v in packed.variables.values()) / 1e6))
    print("  to_dataframe()              %7.2f ms" % timed(lambda: ens.to_dataframe()))
    print("  to_dataframe(packed=True)   %7.2f ms" % timed(lambda: ens.to_dataframe(packed=True)))
    print("  to_netcdf of the plain one  %7.2f ms" % timed(lambda: plain.to_netcdf(path)))
    print("  to_netcdf of the packed one %7.2f ms" % timed(lambda: packed.to_netcdf(path)))
    print("  both, plain                 %7.2f ms" % timed(lambda: ens.to_dataframe().to_netcdf(path)))
    print("  both, packed                %7.2f ms" % timed(lambda: ens.to_dataframe(packed=True).to_netcdf(path)))
