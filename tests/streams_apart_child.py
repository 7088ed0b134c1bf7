"""Child process of tests/test_variants_spawn.py::test_group_streams_end_up_on_hardware_queues_of_their_own: two idle streams
first (the case in which HIP hands two of three group streams the same hardware queue), then a 24-member model stepped as three
member groups and, through the outer boundary, two device models side by side.  The library reports its measurements on stderr
(PYSPEEDY_AMD_STREAMS_APART=2)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.model import BC_MAP, EnsembleModel  # noqa: E402

torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
idle = []
for _ in range(int(sys.argv[1])):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(s)) == 0
    idle.append(s)
bc = np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))
sp = pyspeedy_amd.ModSpectral(0)
model = EnsembleModel(sp, 24)
assert model.config()["chunks"] == 3
model.set_bc(bc)
model.run(12)  # (a call of several steps: three member groups on three streams)
assert (model.check(2) == 0).all()
print("MODEL DONE", file=sys.stderr, flush=True)
# the outer boundary: 4 containers in two device models (PYSPEEDY_AMD_DRIVER_SPLIT=2), each stepped on a stream of its own
states = drv.modelstate_init_ensemble(4)
start, end = drv.create_datetime(1982, 1, 1, 0, 0), drv.create_datetime(1982, 1, 2, 0, 0)
controls = [drv.controlparams_init(start, end) for _ in states]
for name, key in BC_MAP:
    getattr(drv, "set_" + name)(states[0], np.asfortranarray(bc[key], dtype=np.float64))
drv.broadcast_boundary(states, 0)
assert (np.asarray(drv.init_ensemble(states, controls)) == 0).all()
for _ in range(3):
    assert (np.asarray(drv.parallel_step(states, controls)) == 0).all()
assert drv.driver_stats(states[0])[1] == 2
print("DRIVER DONE", file=sys.stderr, flush=True)
