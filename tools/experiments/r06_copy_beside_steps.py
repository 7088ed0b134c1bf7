"""What does a kernel on another stream do to a 36-step call of a 64-member model?  The copy of 48 MB to pinned memory by a copy
kernel of the library's own with 8 workgroups (only with tools/experiments/r06_copy_out_kernel.patch applied and the library
rebuilt), by hipMemcpyAsync, device -> device copies of the same duration, a kernel that only sleeps -- each on three side
streams (streams share the few hardware queues: one that shares with a member group's shows as +0.4 ms whatever it carries).
Times by HIP events on the stepping stream and on the side stream.

    python tools/experiments/r06_copy_beside_steps.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd import _lib  # noqa: E402
from pyspeedy_amd.model import EnsembleModel  # noqa: E402

L = _lib.lib()
bc = np.load(os.path.join(os.path.dirname(pyspeedy_amd.__file__), "data", "example_bc.npz"))
model = EnsembleModel(pyspeedy_amd.ModSpectral(), 64)
model.set_bc(bc)
model.init()
model.run(36)
torch.cuda.synchronize()
n = 48 << 20
stage = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
other = torch.empty(n, dtype=torch.uint8, device="cuda")
pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
sides = [torch.cuda.Stream() for _ in range(6)]
main = torch.cuda.current_stream()


def copy_out(side, wgs=8):
    L.spd_model_export_copy_out.restype = C.c_int
    L.spd_model_export_copy_out.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    rc = L.spd_model_export_copy_out(model._m, C.c_void_p(pinned.data_ptr()), C.c_void_p(stage.data_ptr()), n, wgs, C.c_void_p(side.cuda_stream))
    assert rc == 0, rc


def copy_device(side):
    with torch.cuda.stream(side):
        other.copy_(stage, non_blocking=True)
        for _ in range(40):
            other.copy_(stage, non_blocking=True)


def copy_sdma(side):
    # (hipMemcpyAsync on a stream with nothing pending: the runtime hands it to an SDMA engine; behind a pending command of the
    # stream -- an event wait is one -- it uses its blit kernel instead)
    with torch.cuda.stream(side):
        pinned.copy_(stage, non_blocking=True)


def copy_blit(side):
    with torch.cuda.stream(side):
        pinned.copy_(stage, non_blocking=True)


def sleep(side):
    with torch.cuda.stream(side):
        torch.cuda._sleep(2_000_000)  # cycles


def measure(what, side):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(main)
    if what is not None:
        if what is not copy_sdma:
            side.wait_event(a)
        c.record(side)
        what(side)
        d.record(side)
    model.run(36)
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b), (c.elapsed_time(d) if what is not None else 0.0), (a.elapsed_time(d) if what not in (None, copy_sdma) else 0.0)


CASES = [("nothing beside", None)]
if hasattr(L, "spd_model_export_copy_out"):
    CASES.append(("copy to pinned memory, own kernel", copy_out))
CASES += [("hipMemcpyAsync behind an event wait", copy_blit), ("hipMemcpyAsync on an idle stream", copy_sdma),
          ("device -> device copies (torch)", copy_device), ("a sleeping kernel", sleep), ("nothing beside", None)]
for name, what in CASES:
    for k, side in enumerate(sides[:3] if what is not None else sides[:1]):
        model.init()
        model.run(2)
        measure(what, side)
        steps, beside, end_beside = min(measure(what, side) for _ in range(3))
        print("%-36s side stream %d: 36 steps %.3f ms; the kernel beside them %.3f ms, over at %.3f ms" % (name, k, steps, beside, end_beside), flush=True)
