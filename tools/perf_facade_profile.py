"""cProfile of SpeedyEns.run (host side of the per-step loop).  Usage (GPU box): python tools/perf_facade_profile.py [members] [days]"""
import cProfile
import os
import pstats
import sys
from datetime import datetime

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
days = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ens = SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1 + days))
ens.set_bc()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
ens.run()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
