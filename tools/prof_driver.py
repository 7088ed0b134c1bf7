"""Run each transform kernel a few times (for rocprofv3).  Usage: python tools/prof_driver.py B [iters]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sp = pyspeedy_amd.ModSpectral()
spec = torch.view_as_complex(torch.randn((B, 32, 31, 2), dtype=torch.float64, device="cuda"))
grid = torch.randn((B, 48, 96), dtype=torch.float64, device="cuda")
four = torch.randn((B, 48, 62), dtype=torch.float64, device="cuda")
og, osp, of = torch.empty_like(grid), torch.empty_like(spec), torch.empty_like(four)
L, h, st = sp._lib, sp.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(iters):
    L.spd_spec2grid(h, p(spec), p(og), 1, B, st)
    L.spd_grid2spec(h, p(grid), p(osp), B, st)
    L.spd_legendre_inv(h, p(spec), p(of), B, st)
    L.spd_legendre(h, p(four), p(osp), B, st)
    L.spd_fourier_inv(h, p(four), p(og), 1, B, st)
    L.spd_fourier(h, p(grid), p(of), B, st)
torch.cuda.synchronize()
print("done")
