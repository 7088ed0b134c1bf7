"""SpeedyEns(M).run() per model step, warm, for the launch plans the outer boundary can give the stretches of the time loop:
two device models per GPU from 32 containers up (the default since round 3, made for hosts that call once per step) against one
(PYSPEEDY_AMD_DRIVER_SPLIT=0).  One process per setting (the switch is read once):
    for s in 32 0; do PYSPEEDY_AMD_DRIVER_SPLIT=$s python tools/experiments/r06_facade_plans.py 64 256; done"""
import os
import sys
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd.callbacks import DiagnosticCheck  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

start = datetime(1982, 1, 1)


def timed(M, days, callbacks):
    ens = SpeedyEns(M, start_date=start, end_date=start + timedelta(days=days))
    ens.set_bc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ens.run(callbacks=callbacks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(ens._device_models())
    del ens
    return dt / (36 * days) * 1e3, n


for M in [int(a) for a in sys.argv[1:]] or [64]:
    timed(M, 1, [])
    days = 10 if M <= 64 else 4
    bare = min(timed(M, days, [])[0] for _ in range(2))
    daily, models = min(timed(M, days, [DiagnosticCheck(interval=36)]) for _ in range(2))
    print("PYSPEEDY_AMD_DRIVER_SPLIT=%s  M=%-4d device models %d:  run() %.4f ms/step (%.3f us per member-step);  with a daily hook "
          "(stretches of 36) %.4f ms/step" % (os.environ.get("PYSPEEDY_AMD_DRIVER_SPLIT", "default"), M, models, bare, bare / M * 1e3, daily), flush=True)
