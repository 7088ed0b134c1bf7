"""How much of a launch is its tail: the step's kernels at member counts that are whole and fractional rounds of the machine.

    python tools/exp_tail.py [M ...]        (default 57 64 85 114 128; serial plan, dispatch-packet events, spd_model_profile(2))

The column kernel keeps 8 wavefronts on a CU (256 VGPRs): 2048 on the GPU, 72 per member.  57 members are 2.00 rounds of resident
wavefronts, 64 are 2.25, 85 are 2.99, 114 are 4.01.  If the last fractional round costs as much as a whole one the time per
member at 64 is well above the one at 57 / 85 / 114; if wavefronts retire and start continuously it is not.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [57, 64, 85, 114, 128]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    print("kernel times in the serial plan (one member group), mean us per launch [us per member]; 72 steps after 36 of warm-up")
    for M in sizes:
        sp, model = bench.build_ensemble(types.SimpleNamespace(config="cfg4"), M, 0, dev, None, 0, dev)
        model.set_option("member_groups", 1)
        model.set_option("block_members", 0)
        model.run(36)
        torch.cuda.synchronize()
        model.profile(2)
        model.run(72)
        torch.cuda.synchronize()
        k = model.profile_read_kernels()
        model.profile(0)
        waves = M * 72
        line = "M=%-4d column wavefronts %5d = %.2f rounds of 2048: " % (M, waves, waves / 2048.0)
        for name in ("column", "spec2grid", "grid2spec", "spectral_step"):
            hit = [v for n, v in k.items() if n.startswith(name)]
            if hit:
                mean = sum(v[0] * v[2] for v in hit) / sum(v[2] for v in hit) * 1e3
                line += " %s %.1f [%.3f]" % (name, mean, mean / M)
        print(line, flush=True)
        print("      all: " + "  ".join("%s %.1f" % (n, v[0] * 1e3) for n, v in sorted(k.items())), flush=True)
        model.close()
        del model, sp


if __name__ == "__main__":
    main()
