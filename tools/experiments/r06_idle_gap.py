"""What a host-side pause between two stretches costs the NEXT stretch: SpeedyEns(64).run() with a daily hook that only sleeps."""
import os
import sys
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd import speedy_driver as DRV  # noqa: E402
from pyspeedy_amd.callbacks import BaseCallback  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

acc = {"begin": [], "end": []}
_b, _e = DRV.parallel_steps_begin, DRV.parallel_steps_end


def begin(*a):
    t0 = time.perf_counter()
    try:
        return _b(*a)
    finally:
        acc["begin"].append(time.perf_counter() - t0)


def end(*a):
    t0 = time.perf_counter()
    try:
        return _e(*a)
    finally:
        acc["end"].append(time.perf_counter() - t0)


DRV.parallel_steps_begin, DRV.parallel_steps_end = begin, end


class Pause(BaseCallback):
    def __init__(self, seconds):
        super().__init__(interval=36)
        self.seconds = seconds

    def fire(self, model):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < self.seconds:
            pass


start = datetime(1982, 1, 1)
for pause_ms in (0.0, 0.0, 0.5, 1.5, 3.0, 0.0):
    ens = SpeedyEns(64, start_date=start, end_date=start + timedelta(days=6))
    ens.set_bc()
    torch.cuda.synchronize()
    acc["begin"].clear()
    acc["end"].clear()
    t0 = time.perf_counter()
    ens.run(callbacks=[Pause(pause_ms * 1e-3)])
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    b, e = acc["begin"][1:], acc["end"][1:]
    print("pause %.1f ms per day: run %.4f ms/step;  per stretch (after the first): begin %.2f ms + end %.2f ms = %.2f ms" % (
        pause_ms, total / 216 * 1e3, sum(b) / len(b) * 1e3, sum(e) / len(e) * 1e3, (sum(b) + sum(e)) / len(b) * 1e3), flush=True)
    del ens
