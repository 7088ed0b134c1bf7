"""Set-up cost of a host that follows the REFERENCE's call sequence (no ensemble extensions): per member modelstate_init,
the 12 set_<field> of set_bc, init; then parallel_step (its first call gathers the one-member models into batched ones).
Usage (GPU box): python tools/perf_reference_flow.py [members ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.model import BC_MAP  # noqa: E402

bc = np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))
fields = [(name, np.asfortranarray(bc[key], dtype=np.float64)) for name, key in BC_MAP]


def clock():
    import torch
    torch.cuda.synchronize()
    return time.perf_counter()


for n in [int(a) for a in sys.argv[1:]] or [8, 64]:
    for attempt in range(2):
        start, end = drv.create_datetime(1982, 1, 1, 0, 0), drv.create_datetime(1982, 1, 4, 0, 0)
        t = [clock()]
        states = [drv.modelstate_init() for _ in range(n)]
        controls = [drv.controlparams_init(start, end) for _ in range(n)]
        t.append(clock())
        for s in states:
            for name, value in fields:
                getattr(drv, "set_" + name)(s, value)
        t.append(clock())
        for s, c in zip(states, controls):
            assert drv.init(s, c) == 0
        t.append(clock())
        assert (np.asarray(drv.parallel_step(states, controls)) == 0).all()
        t.append(clock())
        for _ in range(36):
            assert (np.asarray(drv.parallel_step(states, controls)) == 0).all()
        t.append(clock())
        for s, c in zip(states, controls):
            drv.modelstate_close(s)
            drv.controlparams_close(c)
        t.append(clock())
    d = np.diff(t)
    print("n=%3d  modelstate_init %.3f s (%.2f ms each)  set_bc fields %.3f s (%.2f ms per member)  init %.3f s (%.2f ms each)  "
          "first parallel_step (gather) %.3f s  36 steps %.4f s  close %.3f s" %
          (n, d[0], d[0] / n * 1e3, d[1], d[1] / n * 1e3, d[2], d[2] / n * 1e3, d[3], d[4], d[5]), flush=True)
