"""SpeedyEns(M).run(callbacks=[ModelCheckpoint()]): what the reference's in-memory checkpoint hook costs per model step.

    python tools/experiments/r06_checkpoint_hook.py [members] [days]
"""
import cProfile
import os
import pstats
import sys
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd.callbacks import ModelCheckpoint  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
days = int(sys.argv[2]) if len(sys.argv) > 2 else 10
start = datetime(1982, 1, 1)


def run(hooks, profile=False):
    ens = SpeedyEns(M, start_date=start, end_date=start + timedelta(days=days))
    ens.set_bc()
    torch.cuda.synchronize()
    pr = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    ens.run(callbacks=hooks)
    if pr:
        pr.disable()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (36 * days) * 1e3
    if pr:
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    return dt


run([])
print("%d members, %d days: bare %.4f ms per step" % (M, days, run([])), flush=True)
for _ in range(2):
    keep = ModelCheckpoint(interval=36)
    print("with ModelCheckpoint(interval=36): %.4f ms per step; dataframe t %s %s" % (run([keep]), keep.dataframe["t"].values.shape, keep.dataframe["t"].values.dtype), flush=True)
run([ModelCheckpoint(interval=36)], profile=True)
