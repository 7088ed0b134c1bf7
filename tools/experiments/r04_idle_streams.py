"""Does an idle stream slow down a model that steps its member groups on streams created after it?  bench.py's cfg 5 leg
(32 members, 3 member groups on 3 HIP streams) with k idle streams (hipStreamCreate, never used) created first.
This is what the driver's pool of idle streams did to the cfg 5 leg of bench.py's default run (0.187 instead of 0.142 ms/step):
HIP spreads streams over a few hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order, idle ones included.
The library now measures, when it creates a group stream, whether it runs side by side with the groups before it, and replaces it
if not (csrc/stream_apart.hpp); PYSPEEDY_AMD_STREAMS_APART=0 switches that off (what this script showed before), =2 reports.
A negative k: |k| streams from torch's pool (torch.cuda.Stream()) instead of bare hipStreamCreate.
Usage (GPU box, repository root): [GPU_MAX_HW_QUEUES=8] python tools/experiments/r04_idle_streams.py [k ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = bench.parse(["--no-cpu-baseline", "--min-seconds", "1"])
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
torch.zeros(1, device=device)
hip = ctypes.CDLL("libamdhip64.so")


def barrier():
    torch.cuda.synchronize()


def cfg(config, members):
    return bench.config_leg(args, config, members, config, device, None, 0, device, barrier)["ms_per_step"]


idle = []
torch_streams = []
for k in [int(a) for a in sys.argv[1:]] or [2, 0, 1, 3, 4, 6]:
    if k < 0:
        torch_streams = [torch.cuda.Stream() for _ in range(-k)]
        k = 0
    while len(idle) > k:
        assert hip.hipStreamDestroy(idle.pop()) == 0
    while len(idle) < k:
        s = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(s)) == 0
        idle.append(s)
    print("GPU_MAX_HW_QUEUES=%s  %d idle streams + %d of torch:  cfg5 (32 members, 3 groups) %.4f   cfg4 (64 members, 2 groups) %.4f ms/step" % (
        os.environ.get("GPU_MAX_HW_QUEUES", "default"), k, len(torch_streams), cfg("cfg5", 32), cfg("cfg4", 64)), flush=True)
