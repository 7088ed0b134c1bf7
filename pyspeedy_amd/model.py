"""Ensemble model object: the device-resident state of M members and the model time step (`step` of
speedy.f90/time_stepping.f90:38-147 for all members at once).

Host arrays use the reference's shapes and Fortran order (what the f2py getters of speedy_driver.f90.j2:250-334
return); on the device every variable is member-major with that same order inside a member, so get/set are plain copies.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

# registry shapes (registry/model_state_def.py:121-495) of the variables the model object holds
_C, _F = np.complex128, np.float64
SHAPES = {
    "vor": (_C, (31, 32, 8, 2)), "div": (_C, (31, 32, 8, 2)), "t": (_C, (31, 32, 8, 2)), "tr": (_C, (31, 32, 8, 2)),
    "ps": (_C, (31, 32, 2)), "phi": (_C, (31, 32, 8)), "phis": (_C, (31, 32)), "tcorh": (_C, (31, 32)), "qcorh": (_C, (31, 32)),
    "rad_st4a": (_F, (96, 48, 8, 2)), "rad_flux": (_F, (96, 48, 4)), "tt_rsw": (_F, (96, 48, 8)),
    "rad_tau2": (_F, (96, 48, 8, 4)), "rad_strat_corr": (_F, (96, 48, 2)),
}
for _n in ("fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp", "soil_avail_water",
           "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction", "stratospheric_correction",
           "alb_surface", "precnv", "precls", "cbmf", "slrd", "slr", "olr", "tsr", "ssrd", "ssr", "qcloud_equiv"):
    SHAPES[_n] = (_F, (96, 48))
for _n in ("slru", "ustr", "vstr", "shf", "evap", "hfluxn"):
    SHAPES[_n] = (_F, (96, 48, 3))

DELT = 86400.0 / 36  # params.f90:33


class EnsembleModel:
    def __init__(self, spectral, nmembers):
        self.sp = spectral
        self.nmembers = int(nmembers)
        self._lib = _lib.lib()
        self._m = C.c_void_p()
        with torch.cuda.device(spectral.device):
            check(self._lib.spd_model_create(spectral.handle, self.nmembers, C.byref(self._m)), "spd_model_create")

    def close(self):
        if getattr(self, "_m", None) is not None and self._m:
            self._lib.spd_model_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- registry access (speedy_driver.f90.j2:250-334) --------------------------------------------------
    def set(self, name, value, member=-1):
        """Copy a host array (reference shape) into one member, or into every member when member == -1."""
        dtype, shape = SHAPES[name]
        a = np.asarray(value, dtype=dtype)
        if a.shape != shape:
            raise ValueError("Array shape missmatch: %s expects %s, got %s" % (name, shape, a.shape))  # speedy.py:153
        flat = np.ascontiguousarray(a.ravel(order="F"))
        check(self._lib.spd_model_set(self._m, name.encode(), int(member), flat.ctypes.data_as(C.c_void_p), flat.nbytes),
              "spd_model_set(%s)" % name)

    def get(self, name, member=0):
        dtype, shape = SHAPES[name]
        flat = np.empty(int(np.prod(shape)), dtype=dtype)
        check(self._lib.spd_model_get(self._m, name.encode(), int(member), flat.ctypes.data_as(C.c_void_p), flat.nbytes),
              "spd_model_get(%s)" % name)
        return flat.reshape(shape, order="F")

    def set_co2(self, value):
        check(self._lib.spd_model_set_co2(self._m, float(value)), "spd_model_set_co2")

    # ---- time stepping -------------------------------------------------------------------------------------
    def set_time_step(self, dt):
        check(self._lib.spd_model_set_time_step(self._m, float(dt)), "spd_model_set_time_step")

    def step_dynamics(self, j1, j2, dt, compute_shortwave):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_step_dynamics(self._m, int(j1), int(j2), float(dt), int(bool(compute_shortwave)), stream),
              "spd_model_step_dynamics")

    def check(self, time_level=2, with_diag=False):
        """diagnostics.f90 range check; returns int32 codes per member (0 ok, -2 out of range) [and the diagnostics]."""
        codes = np.zeros(self.nmembers, dtype=np.int32)
        diag = np.zeros((self.nmembers, 3, 8)) if with_diag else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_check(self._m, int(time_level), codes.ctypes.data_as(C.c_void_p),
                                        diag.ctypes.data_as(C.c_void_p) if with_diag else None, stream), "spd_model_check")
        return (codes, diag) if with_diag else codes
