import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "pyspeedy_amd", sys.argv[1] if len(sys.argv) > 1 else "libpyspeedy_amd.so"))
import torch
print("torch avail", torch.cuda.is_available())
h = C.c_void_p()
lib.spd_last_error.restype = C.c_char_p
rc = lib.spd_create(C.byref(h), 0)
print("spd_create rc", rc, lib.spd_last_error() if rc else "")
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l}))
