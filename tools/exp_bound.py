"""Per-kernel times of the bench workload with an experimental build of the library (results may be WRONG on purpose:
bounding experiments).  PYSPEEDY_AMD_LIB=build_variants/lib_<x>.so python tools/exp_bound.py [members] [steps] [cfg4|cfg5]
Prints ms/step (one call of `steps` steps, median of 5) and the level-2 per-kernel HIP-event table of one simulated day."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 360
    tag = os.path.basename(os.environ.get("PYSPEEDY_AMD_LIB", "committed"))
    args = types.SimpleNamespace(config=sys.argv[3] if len(sys.argv) > 3 else "cfg4")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sp, model = bench.build_ensemble(args, M, 0, dev, None, 0, dev)
    model.run(36)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.run(steps)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    ts.sort()
    model.profile(2)
    model.run(36)
    prof = model.profile_read_kernels()
    model.profile(0)
    row = " ".join("%s=%.1f" % (k, v[0] * 1e3) for k, v in prof.items())
    print("%-28s M=%d ms/step median %.4f min %.4f | us: %s" % (tag, M, ts[2], ts[0], row), flush=True)


if __name__ == "__main__":
    main()
