"""Time of spd_model_init (set_bc of all members + init) for an M-member model.
Usage (GPU box): [PYSPEEDY_AMD_LIB=build_variants/lib_x.so] python tools/perf_init.py [members ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd.model import EnsembleModel  # noqa: E402

bc = np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))
sp = pyspeedy_amd.ModSpectral(0)
for M in [int(a) for a in sys.argv[1:]] or [32, 256]:
    for attempt in range(2):  # (the second creation runs with a warm allocator)
        model = EnsembleModel(sp, M)
        model.init_sst_anom(2)
        for state_name, bc_name in pyspeedy_amd.model.BC_MAP:
            model.set(state_name, np.asarray(bc[bc_name], dtype=np.float64), -1)
        if os.environ.get("PERF_INIT_DISTINCT"):  # every member with an SST climatology of its own
            for i in range(1, M):
                model.set("sst12", np.asarray(bc["sst"], dtype=np.float64) + 0.01 * i, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.init((1982, 1, 1, 0, 0))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        codes = model.check(2)
        assert (codes == 0).all()
        model.close()
    print("%s  M=%4d  spd_model_init %.3f s  (%.2f ms per member)" % (os.path.basename(os.environ.get("PYSPEEDY_AMD_LIB", "default")),
                                                                       M, dt, dt / M * 1e3), flush=True)
sp.close()
