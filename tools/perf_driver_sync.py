"""Cost of the reference-shaped synchronous loop -- spd_parallel_step (step + range check + error codes back) once per model
step, as an f2py / Fortran host of the reference would call it -- next to the overlapped form (the bare device loop is bench.py's headline).
Usage (GPU box): python tools/perf_driver_sync.py [members]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from datetime import datetime  # noqa: E402

from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = 360


def main():
    ens = SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 3, 1))
    for member in ens:
        member.set_bc()
    states = [m._state_cnt for m in ens.members]
    controls = [m._control_cnt for m in ens.members]
    assert (drv.parallel_step(states, controls) == 0).all()
    for _ in range(36):
        drv.parallel_step(states, controls)
    # per-step wall times, median reported: the runtime stalls ONCE per process for ~40 ms some 20-30 ms after the first launches
    # (a pool of its own growing), which an average over a few hundred steps would smear over whichever loop it falls into
    def median_ms(times):
        times = sorted(times)
        return times[len(times) // 2] * 1e3

    torch.cuda.synchronize()
    per_step = []
    for _ in range(STEPS):
        t0 = time.perf_counter()
        codes = drv.parallel_step(states, controls)
        per_step.append(time.perf_counter() - t0)
    assert (codes == 0).all()
    sync_ms, stalls = median_ms(per_step), sum(1 for t in per_step if t > 5e-3)
    per_step = []
    token = drv.parallel_step_begin(states, controls)
    for _ in range(STEPS):
        t0 = time.perf_counter()
        nxt = drv.parallel_step_begin(states, controls)
        assert (drv.parallel_step_end(token) == 0).all()
        token = nxt
        per_step.append(time.perf_counter() - t0)
    assert (drv.parallel_step_end(token) == 0).all()
    ovl_ms, stalls = median_ms(per_step), stalls + sum(1 for t in per_step if t > 5e-3)
    print("M=%d  parallel_step (synchronous) %.4f ms/step;  begin / end overlapped %.4f  (medians; %d steps longer than 5 ms; "
          "%d device models)" % (M, sync_ms, ovl_ms, stalls, len({drv.device_model(s)[0]._m.value for s in states})))


main()
