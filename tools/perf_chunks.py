"""Experiment: the same 64 members as several smaller models stepped concurrently on separate HIP streams (one host thread
each), so that a chunk's intermediate grids may stay in the 256 MiB Infinity Cache between producer and consumer kernels.
Usage: perf_chunks.py [total_members]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd.model import EnsembleModel  # noqa: E402

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = 72
sp = pyspeedy_amd.ModSpectral()
with np.load(pyspeedy_amd.example_bc_file()) as z:
    bc = {k: z[k] for k in z.files}

for nchunks in (1, 2, 4, 8, 16):
    if TOTAL % nchunks:
        continue
    m = TOTAL // nchunks
    models = [EnsembleModel(sp, m) for _ in range(nchunks)]
    for mod in models:
        mod.set_bc(bc)
        mod.run(6)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(nchunks)]

    def work(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):  # step by step, so that the chunks interleave in time
                models[i].run(1)

    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        threads = [threading.Thread(target=work, args=(i, STEPS)) for i in range(nchunks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS * 1e3
    print("%2d chunks x %3d members on %2d streams: %.3f ms per step of all %d members (%.2f us per member-step)"
          % (nchunks, m, nchunks, dt, TOTAL, dt * 1e3 / TOTAL), flush=True)
    for mod in models:
        mod.close()
