"""A 36-step call of a 64-member model with and without the range check of every step recorded (spd_model_step_checked_begin / _end
against spd_model_step): what do the checks -- blocks in front of the next step's spectral -> grid launch that write their codes
to pinned host memory -- cost?

    python tools/experiments/r06_checked_vs_plain.py [members] [steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd.model import EnsembleModel  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 36
bc = np.load(os.path.join(os.path.dirname(pyspeedy_amd.__file__), "data", "example_bc.npz"))
model = EnsembleModel(pyspeedy_amd.ModSpectral(), M)
model.set_bc(bc)


def timed(fn, reps=max(2, 800 // K - 1)):
    model.init()
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / K * 1e3


for _ in range(3):
    print("%d members, calls of %d steps:  plain %.4f ms per step   checked %.4f ms per step"
          % (M, K, timed(lambda: model.run(K)), timed(lambda: model.run_checked(K))), flush=True)
