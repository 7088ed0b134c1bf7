"""The Legendre stage on its own at B fields (default 16 384: in- and outputs of 650 MB, the HBM regime): `iters` launches (default
1000: a run of a few milliseconds ends before the GPU has reached its clocks -- 60 launches read 0.58 / 0.49 where 1000 read
0.71 / 0.51 in one session) each of
spd_legendre_inv (spec2grid_kernel<LegendreOnly>) and spd_legendre (grid2spec_kernel<LegendreOnly>) through the operator-level C
ABI -- the command tools/collect_profiles.sh traces with rocprofv3 (kernel stats, then the FETCH_SIZE / WRITE_SIZE passes) for
north_star's literal target: >= 40 % of the HBM roofline on the Legendre transform, 39 680 algorithmic bytes per field.
Usage (GPU box): python tools/perf_legendre_only.py [B] [iters]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402

S, F = 15872, 23808


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    sp = pyspeedy_amd.ModSpectral()
    L, h = sp._lib, sp.handle
    gen = torch.Generator(device="cuda").manual_seed(1234)
    spec = torch.view_as_complex(torch.randn((B, 32, 31, 2), dtype=torch.float64, device="cuda", generator=gen))
    four = torch.empty((B, 48, 62), dtype=torch.float64, device="cuda")
    out_spec = torch.empty_like(spec)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for name, fn in (("legendre_inv", lambda: L.spd_legendre_inv(h, p(spec), p(four), B, st)),
                     ("legendre", lambda: L.spd_legendre(h, p(four), p(out_spec), B, st))):
        for _ in range(min(100, iters)):
            assert fn() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            assert fn() == 0
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / iters * 1e-3
        print("B=%d %-13s %8.2f us/launch  %6.2f ns/field  %7.1f GB/s algorithmic (%d B/field)  = %.3f of 8 TB/s"
              % (B, name, t * 1e6, t / B * 1e9, (S + F) * B / t / 1e9, S + F, (S + F) * B / t / 8e12), flush=True)
    sp.close()


if __name__ == "__main__":
    main()
