"""Micro-benchmark of the transform kernels: algorithmic GB/s vs batch size.
Usage (GPU box): [PERF_LIB=path/to/variant.so] python tools/perf_transforms.py [B ...]
PERF_LIB points the loader at an experimental build of the same C ABI (build_variants/, not committed)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd._lib as _L  # noqa: E402

if os.environ.get("PERF_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["PERF_LIB"])
import pyspeedy_amd  # noqa: E402

S, F, G = 15872, 23808, 36864


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [64, 512, 728, 4096, 5824, 16384]
    tag = os.path.basename(os.environ.get("PERF_LIB", "default"))
    sp = pyspeedy_amd.ModSpectral()
    L = sp._lib
    for B in sizes:
        spec = torch.view_as_complex(torch.randn((B, 32, 31, 2), dtype=torch.float64, device="cuda"))
        grid = torch.randn((B, 48, 96), dtype=torch.float64, device="cuda")
        four = torch.randn((B, 48, 62), dtype=torch.float64, device="cuda")
        og, osp, of = torch.empty_like(grid), torch.empty_like(spec), torch.empty_like(four)
        h, st = sp.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        rows = [
            ("spec2grid", lambda: L.spd_spec2grid(h, p(spec), p(og), 1, B, st), S + G),
            ("grid2spec", lambda: L.spd_grid2spec(h, p(grid), p(osp), B, st), S + G),
            ("legendre_inv", lambda: L.spd_legendre_inv(h, p(spec), p(of), B, st), S + F),
            ("legendre", lambda: L.spd_legendre(h, p(four), p(osp), B, st), S + F),
            ("fourier_inv", lambda: L.spd_fourier_inv(h, p(four), p(og), 1, B, st), F + G),
            ("fourier", lambda: L.spd_fourier(h, p(grid), p(of), B, st), F + G),
        ]
        for name, fn, bytes_per in rows:
            t = timeit(fn)
            print("%s B=%6d %-13s %9.2f us  %8.1f GB/s  %6.2f ns/field" %
                  (tag, B, name, t * 1e6, bytes_per * B / t / 1e9, t / B * 1e9), flush=True)
    sp.close()
    # device copy bandwidth for reference
    n = 1 << 28
    x = torch.empty(n, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x), iters=20)
    print("device copy: %.1f GB/s (read+write)" % (2 * n / t / 1e9))


if __name__ == "__main__":
    main()
