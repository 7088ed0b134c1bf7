"""Wall time of the pySPEEDY-style host path (SpeedyEns.run: parallel_step + range check every step, optional daily
NetCDF export with its device -> host copies) next to the bare device loop (EnsembleModel.run).  Usage: perf_facade.py [M]"""
import os
import sys
import tempfile
import time
from datetime import datetime

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.callbacks import XarrayExporter  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
days = 2


def run(callbacks):
    ens = SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1 + days))
    t0 = time.perf_counter()
    for member in ens:
        member.set_bc()
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    t0 = time.perf_counter()
    ens.run(callbacks=callbacks)
    torch.cuda.synchronize()
    return t_init, (time.perf_counter() - t0) / (36 * days) * 1e3, ens


t_init, ms_plain, ens = run([])
print("M=%d  set_bc of all members %.2f s;  SpeedyEns.run (step + check each step): %.3f ms/step" % (M, t_init, ms_plain))
models = [m for m, _ in ens._device_models()]  # (32 or more members live in two device models: time them all, side by side)
torch.cuda.synchronize()
t0 = time.perf_counter()
streams = [torch.cuda.Stream() for _ in models]
for model, stream in zip(models, streams):
    with torch.cuda.stream(stream):
        model.run(72)
torch.cuda.synchronize()
print("      EnsembleModel.run of the %d device model(s) (no per-step check): %.3f ms/step" % (
    len(models), (time.perf_counter() - t0) / 72 * 1e3))
with tempfile.TemporaryDirectory() as tmp:
    _, ms_exp, _ = run([XarrayExporter(output_dir=tmp)])
    size = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
print("      with daily NetCDF export of u, v, t, q, phi, ps for all members: %.3f ms/step (%.1f MB written)" % (ms_exp, size / 1e6))
