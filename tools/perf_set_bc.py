import sys, time
sys.path.insert(0, "/root/repo")
from datetime import datetime
import torch
from pyspeedy_amd.speedy import SpeedyEns
for n in (64, 256):
    t0 = time.perf_counter(); ens = SpeedyEns(n, end_date=datetime(1982, 1, 2)); ens.set_bc(); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    del ens
    t0 = time.perf_counter(); ens = SpeedyEns(n, end_date=datetime(1982, 1, 2))
    for m in ens: m.set_bc()
    torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    del ens
    print("SpeedyEns(%d): ens.set_bc() %.3f s; for member in ens: member.set_bc() %.3f s" % (n, t1, t2), flush=True)
