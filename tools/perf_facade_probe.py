"""Per-step wall time of Speedy.run and SpeedyEns.run, alternating.  Usage (GPU box): python tools/perf_facade_probe.py [members] [days]"""
import os
import sys
import time
from datetime import datetime

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyspeedy_amd.speedy import Speedy, SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
days = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def timed(model):
    model.set_bc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (36 * days) * 1e3


for rep in range(3):
    a = timed(SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1 + days)))
    b = timed(Speedy(start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1 + days)))
    print("rep %d  SpeedyEns(%d).run %.4f ms/step   Speedy.run %.4f ms/step" % (rep, M, a, b), flush=True)
