"""Time the device-resident model step for M members, starting from the golden step state.  Usage: perf_step.py [M] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyspeedy_amd
from pyspeedy_amd.model import EnsembleModel, DELT
from test_step_gpu import load_initial
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 36
gold = np.load(os.path.join(ROOT, "tests", "golden", "step.npz"))
sp = pyspeedy_amd.ModSpectral()
model = EnsembleModel(sp, M)
load_initial(model, gold)
model.set_time_step(2 * DELT)
for i in range(6):
    model.step_dynamics(2, 2, 2 * DELT, i % 3 == 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    model.step_dynamics(2, 2, 2 * DELT, i % 3 == 0)
torch.cuda.synchronize()
el = time.perf_counter() - t0
codes = model.check(2)
print("M=%d  %.3f ms/step  %.2f us/member-step  %.0f sim-years/day  errors=%d" % (M, el / steps * 1e3, el / steps / M * 1e6, M * 86400 / (el / steps * 13140), int((codes != 0).sum())))
