"""Per-phase latency of the fused transform kernels (needs a library built with -DSPD_TRACE):
    hipcc ... -DSPD_TRACE -c pyspeedy_amd/csrc/transforms.hip ; link as build_variants/lib_trace.so
    PERF_LIB=build_variants/lib_trace.so python tools/trace_transforms.py [B ...]
Prints the mean time from kernel entry to each phase boundary (thread 0 of every workgroup, 100 MHz wall clock)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd._lib as _L  # noqa: E402

_L.LIB_PATH = os.path.abspath(os.environ["PERF_LIB"])
import pyspeedy_amd  # noqa: E402

INV = ["S staged", "Legendre done", "FFT group done", "FFT block done", "stores issued"]
FWD = ["grid staged", "FFT block done", "FFT group done", "sym/antisym done", "Legendre done", "stores issued"]


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [728, 5824]
    sp = pyspeedy_amd.ModSpectral()
    L = sp._lib
    L.spd_trace_read.restype = C.c_int
    L.spd_trace_read.argtypes = [C.c_void_p, C.c_void_p]
    out, cnt = np.zeros(16), np.zeros(2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B in sizes:
        spec = torch.view_as_complex(torch.randn((B, 32, 31, 2), dtype=torch.float64, device="cuda"))
        grid = torch.randn((B, 48, 96), dtype=torch.float64, device="cuda")
        og, osp = torch.empty_like(grid), torch.empty_like(spec)
        p = lambda t: C.c_void_p(t.data_ptr())
        for name, fn, labels, d in (("spec2grid", lambda: L.spd_spec2grid(sp.handle, p(spec), p(og), 1, B, st), INV, 0),
                                    ("grid2spec", lambda: L.spd_grid2spec(sp.handle, p(grid), p(osp), B, st), FWD, 1)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            L.spd_trace_read(out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            L.spd_trace_read(out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
            t = out[d * 8:d * 8 + len(labels)] * 0.01  # us
            prev = 0.0
            print("%s B=%d (%d workgroups sampled)" % (name, B, int(cnt[d])))
            for lab, v in zip(labels, t):
                print("   %-18s +%6.2f us  (at %6.2f us)" % (lab, v - prev, v))
                prev = v


if __name__ == "__main__":
    main()
