"""What would hiding the range-check launch of the drop-in path be worth?  spd_parallel_step_begin / _end once per model step over
n containers, total wall time of 360 steps with the device synchronised at the end -- with the committed library, and with a
build that launches no check kernel at all and reports "all fine" from the host (tools/experiments/r04_no_check_launch_bound.patch,
build_variants/lib_nocheck.so): the upper bound of any scheme that takes the check off the step's stream.
Usage (GPU box, repository root): [PYSPEEDY_AMD_LIB=build_variants/lib_nocheck.so] python tools/experiments/r04_dropin_bound.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.model import BC_MAP  # noqa: E402

bc = np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))
for n in (64, 8, 1):
    states = drv.modelstate_init_ensemble(n)
    start, end = drv.create_datetime(1982, 1, 1, 0, 0), drv.create_datetime(1982, 3, 1, 0, 0)
    controls = [drv.controlparams_init(start, end) for _ in states]
    for name, key in BC_MAP:
        getattr(drv, "set_" + name)(states[0], np.asfortranarray(bc[key], dtype=np.float64))
    drv.broadcast_boundary(states, 0)
    assert (np.asarray(drv.init_ensemble(states, controls)) == 0).all()
    s, c = np.asarray(states, dtype=np.int64), np.asarray(controls, dtype=np.int64)
    best = 1e30
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        token = drv.parallel_step_begin(s, c)
        for _ in range(359):
            nxt = drv.parallel_step_begin(s, c)
            drv.parallel_step_end(token)
            token = nxt
        drv.parallel_step_end(token)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 360)
    print("%s  %2d containers: begin/end %.4f ms per step (360 steps, device synchronised at the end)" % (
        os.path.basename(os.environ.get("PYSPEEDY_AMD_LIB", "committed")), n, best * 1e3), flush=True)
    for st, ct in zip(states, controls):
        drv.modelstate_close(st)
        drv.controlparams_close(ct)
