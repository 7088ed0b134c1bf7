"""What a fresh SpeedyEns pays in its first run(): the same ensemble run twice (the second time from where the first ended)."""
import os
import sys
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
start = datetime(1982, 1, 1)
ens = SpeedyEns(M, start_date=start, end_date=start + timedelta(days=10))
ens.set_bc()
torch.cuda.synchronize()
for leg in range(3):
    t0 = time.perf_counter()
    ens.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("SPLIT=%s M=%d run %d of the same ensemble: %.4f ms/step (%.2f ms in all)" % (os.environ.get("PYSPEEDY_AMD_DRIVER_SPLIT", "default"), M, leg + 1, dt / 360 * 1e3, dt * 1e3), flush=True)
    # (the next run starts where this one ended: the control containers take new dates, the state stays)
    new_start = ens.current_date
    for member in ens:
        member.start_date, member.end_date = new_start, new_start + timedelta(days=10)
        member.current_date = new_start
    ens.current_date = new_start
