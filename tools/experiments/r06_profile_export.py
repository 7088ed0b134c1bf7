import cProfile, pstats, sys, os, time
from datetime import datetime, timedelta
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pyspeedy_amd.speedy import SpeedyEns
ens = SpeedyEns(64, start_date=datetime(1982,1,1), end_date=datetime(1982,1,2))
ens.set_bc(); ens.run()
bufs = {}
for i in range(4):
    f = ens.to_dataframe(packed=True, slot=i % 2, buffers=bufs, wait=False)
    for e in f.ready: e.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    f = ens.to_dataframe(packed=True, slot=i % 2, buffers=bufs, wait=False)
    for e in f.ready: e.synchronize()
pr.disable()
print("per call %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
