"""Bound experiments on the fused column kernel of small ensembles (profiles/r04_small_shard_bound.txt).

The variant libraries (tools/experiments/r04_small_shard_split.patch, tools/build_variant.sh) compute WRONG results on purpose, so
they cannot spin a model up themselves.  Step 1, with the committed library:   python tools/exp_column.py save M state.npz
spins an M-member ensemble of the bench workload up for two days and saves every member's state.  Step 2, per variant:
    PYSPEEDY_AMD_LIB=build_variants/lib_exp_x.so python tools/exp_column.py time state.npz
loads that state, takes ONE model step (its spectral -> grid launch leaves valid inputs for the column kernel in the work arrays;
what the variant's column kernel then writes is never read again) and times `spd_exp_column`: back-to-back launches of the
fused column kernel alone on that frozen input, between two HIP events -- shortwave steps, other steps, and the 1 : 2 mix of
the model."""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd.model import EnsembleModel  # noqa: E402


def save(M, path):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sp, model = bench.build_ensemble(types.SimpleNamespace(config="cfg4"), M, 0, dev, None, 0, dev)
    model.run(72)
    assert (model.check(2) == 0).all()
    out = {"members": np.asarray(M)}
    for i in range(M):
        for k, v in model.state_dict(i).items():
            out["m%d/%s" % (i, k)] = v
    np.savez(path, **out)
    print("saved %d members to %s" % (M, path))


def time_variant(path, launches=400):
    z = np.load(path)
    M = int(z["members"])
    torch.cuda.set_device(0)
    sp = pyspeedy_amd.ModSpectral(0)
    model = EnsembleModel(sp, M)
    for i in range(M):
        prefix = "m%d/" % i
        model.load_state_dict({k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}, member=i)
    model.run(1)
    torch.cuda.synchronize()
    L = sp._lib
    L.spd_exp_column.restype = C.c_int
    L.spd_exp_column.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    row = []
    for mode, name in ((0, "other"), (1, "shortwave"), (-1, "mix")):
        assert L.spd_exp_column(model._m, 30, mode, st) == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            a.record()
            assert L.spd_exp_column(model._m, launches, mode, st) == 0
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b) / launches * 1e3)
        row.append("%s %.2f us" % (name, best))
    print("%-26s M=%-3d column kernel alone, back to back: %s" % (os.path.basename(os.environ.get("PYSPEEDY_AMD_LIB", "committed")), M,
                                                               "  ".join(row)), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "save":
        save(int(sys.argv[2]), sys.argv[3])
    else:
        time_variant(sys.argv[2])
