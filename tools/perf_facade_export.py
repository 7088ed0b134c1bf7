"""Where a day of SpeedyEns(M).run(callbacks=[XarrayExporter()]) goes: the stretch on the GPU, the exporter's wait for a free buffer,
the packed export (transform + pack kernels + copy to pinned memory), the header, and -- in its own thread -- the file write.

    python tools/perf_facade_export.py [members] [days] [output dir]
"""
import os
import sys
import tempfile
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyspeedy_amd import callbacks as CB  # noqa: E402
from pyspeedy_amd import dataset as DS  # noqa: E402
from pyspeedy_amd import speedy as SP  # noqa: E402
from pyspeedy_amd import speedy_driver as DRV  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
days = int(sys.argv[2]) if len(sys.argv) > 2 else 5
where = sys.argv[3] if len(sys.argv) > 3 else None
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc.setdefault(name, []).append(time.perf_counter() - t0)
    return wrapper


DRV.parallel_steps_end = timed("stretch: wait for the device (parallel_steps_end)", DRV.parallel_steps_end)
DRV.parallel_steps_begin = timed("stretch: enqueue (parallel_steps_begin)", DRV.parallel_steps_begin)
CB.XarrayExporter._wait = timed("export: wait for a free buffer", CB.XarrayExporter._wait)
SP.SpeedyEns.to_dataframe = timed("export: to_dataframe(packed) = transforms + pack + copy out", SP.SpeedyEns.to_dataframe)
DS.prepare_netcdf = timed("export: NetCDF header", DS.prepare_netcdf)
DS.write_prepared = timed("writer thread: write the file", DS.write_prepared)
CB.XarrayExporter.fire = timed("export: fire() in all", CB.XarrayExporter.fire)

_act_ahead = SP._act_ahead
# (the first run of each kind pays what a process pays once: pinned buffers, code objects; ahead = the exporter enqueues its device
# work behind the stretch before the host has waited for it, speedy._act_ahead -- off: it acts after the wait, as until round 6)
for export, ahead in ((False, True), (True, False), (True, False), (True, False), (True, True), (True, True), (True, True)):
    SP._act_ahead = _act_ahead if ahead else (lambda due, model: None if due else [])
    ens = SP.SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1) + timedelta(days=days))
    ens.set_bc()
    with tempfile.TemporaryDirectory(prefix="pyspeedy_perf_", dir=where) as tmp:
        acc.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens.run(callbacks=[CB.XarrayExporter(output_dir=tmp)] if export else [])
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        nbytes = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
    print("%d members, %d days, daily export %s%s: %.4f ms per step, %.1f MB written"
          % (M, days, "on" if export else "off", (", enqueued ahead" if ahead else ", after the wait") if export else "", total / (36 * days) * 1e3, nbytes / 1e6))
    for name, ts in sorted(acc.items()):
        print("   %-62s n=%3d  total %8.2f ms  mean %8.3f ms  max %8.3f ms" % (name, len(ts), sum(ts) * 1e3, sum(ts) / len(ts) * 1e3, max(ts) * 1e3))
    del ens
