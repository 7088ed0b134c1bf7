"""Does a hipGraph replay of the model step beat eager launches for small ensembles?  
35 consecutive steps inside one simulated day (no daily forcing) are captured from the launch stream into one graph (140
kernel nodes at up to 8 members) and replayed; the per-step time is compared with the same 35 steps launched eagerly.
The replay re-runs the SAME launches (dates and shortwave pattern baked in), so the state drifts from the calendar: timing only.
Usage (GPU box): python tools/exp_graph.py [members ...]"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1, 8]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for M in sizes:
        args = types.SimpleNamespace(config="cfg4")
        sp, model = bench.build_ensemble(args, M, 0, dev, None, 0, dev)
        model.run(37)  # current_step = 37: the next 35 steps stay inside the day
        torch.cuda.synchronize()
        eager = []
        for _ in range(20):
            model.run(36)  # finish the day (one step with forcing) so that every timed block is steps 1 ... 35 of a day
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.run(35)
            torch.cuda.synchronize()
            eager.append((time.perf_counter() - t0) / 35 * 1e6)
            model.run(1)
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        model.run(36)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            g.capture_begin()
            model.run(35)
            g.capture_end()
        torch.cuda.synchronize()
        replay = []
        for _ in range(20):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            replay.append((time.perf_counter() - t0) / 35 * 1e6)
        eager.sort(), replay.sort()
        print("M=%d: eager %.2f us/step (min %.2f)   graph replay %.2f us/step (min %.2f)" % (
            M, eager[len(eager) // 2], eager[0], replay[len(replay) // 2], replay[0]), flush=True)
        del g
        model.close()
        sp.close()


if __name__ == "__main__":
    main()
