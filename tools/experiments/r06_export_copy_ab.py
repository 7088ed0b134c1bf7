"""Does the copy of a day's output to pinned memory hold up the next stretch of steps?  SpeedyEns(64), 10 days, daily export, with
the copy (a) enqueued behind the pack kernels on a side stream, as until the middle of round 6 (the runtime then uses its blit
kernel), (b) not made at all, (c) handed to the runtime by the writer thread once the pack kernels are through (the stream is
idle then: an SDMA engine; speedy_driver._CopyOut, what the package does), (d) made by a copy kernel of the library's own with
4 .. 256 workgroups -- only with tools/experiments/r06_copy_out_kernel.patch applied and the library rebuilt; then the rate of each
way of copying on an idle GPU.

    python tools/experiments/r06_export_copy_ab.py
"""
import os
import sys
import tempfile
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd import callbacks as CB  # noqa: E402
from pyspeedy_amd import speedy as SP  # noqa: E402
from pyspeedy_amd import speedy_driver as DRV  # noqa: E402

import ctypes as C  # noqa: E402

from pyspeedy_amd import _lib  # noqa: E402

stock = DRV._CopyOut
try:
    own_kernel = _lib.lib().spd_model_export_copy_out
    own_kernel.restype, own_kernel.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
except AttributeError:
    own_kernel = None
WORKGROUPS = [0]


class Enqueued:
    """the copy enqueued at once, behind the pack kernels (an event wait on the side stream)"""

    def __init__(self, device, packed, buf, stage, pieces, side, how="blit"):
        with torch.cuda.device(device):
            side.wait_event(packed)
            with torch.cuda.stream(side):
                for start, nbytes in pieces:
                    if how == "blit":
                        buf[start:start + nbytes].copy_(stage[start:start + nbytes], non_blocking=True)
                    elif how == "own":
                        model = [m for m in MODELS if m.sp.device == device][0]
                        rc = own_kernel(model._m, C.c_void_p(buf.data_ptr() + start), C.c_void_p(stage.data_ptr() + start), nbytes,
                                        WORKGROUPS[0], C.c_void_p(side.cuda_stream))
                        assert rc == 0, rc
                self.done = torch.cuda.Event()
                self.done.record(side)

    def synchronize(self):
        self.done.synchronize()


def blit(*a):
    return Enqueued(*a, how="blit")


def none(*a):
    return Enqueued(*a, how="none")


def own(*a):
    return Enqueued(*a, how="own")


MODELS = []


def run(days, export=True):
    ens = SP.SpeedyEns(64, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1) + timedelta(days=days))
    ens.set_bc()
    MODELS[:] = [g[0] for g in DRV._group_by_model([m._state_cnt for m in ens])[1].values()]
    with tempfile.TemporaryDirectory(prefix="pyspeedy_ab_") as tmp:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens.run(callbacks=[CB.XarrayExporter(output_dir=tmp)] if export else [])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (36 * days) * 1e3


run(2)
cases = [("bare run, no export", None, 0), ("enqueued behind the pack kernels (blit kernel)", blit, 0), ("no copy", none, 0),
         ("handed over by the writer thread (SDMA)", stock, 0)]
if own_kernel is not None:
    cases += [("own kernel, %d workgroups" % w, own, w) for w in (4, 8, 16, 32, 64, 256)]
for name, fn, wgs in cases + cases[:4]:
    if fn is not None:
        DRV._CopyOut, WORKGROUPS[0] = fn, wgs
    print("%-48s %.4f ms per step" % (name, run(10, fn is not None)), flush=True)

# the copies alone, idle GPU: 48 MB device -> pinned
run(1)
device = MODELS[0].sp.device
n = 48 << 20
stage = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
buf = torch.empty(n, dtype=torch.uint8, pin_memory=True)
side = torch.cuda.Stream(device=device)
for name, fn, wgs in cases[1:2] + cases[3:]:
    WORKGROUPS[0] = wgs

    def once():
        packed = torch.cuda.Event()
        packed.record(torch.cuda.current_stream(device))
        return fn(device, packed, buf, stage, [[0, n]], side)
    buf.zero_()
    once().synchronize()
    assert torch.equal(buf, stage.cpu()), name
    t0 = time.perf_counter()
    for _ in range(10):
        once().synchronize()
    print("%-48s %.1f GB/s alone (host clock, one copy at a time)" % (name, 10 * n / (time.perf_counter() - t0) / 1e9), flush=True)
