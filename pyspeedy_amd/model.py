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

for _n in ("stl12", "snowd12", "soilw12", "sst12", "sea_ice_frac12", "soil_wc_l1", "soil_wc_l2", "soil_wc_l3"):
    SHAPES[_n] = (_F, (96, 48, 12))
SHAPES["sst_anom"] = (_F, (96, 48, 3))
for _n in ("stlcl_obs", "snowdcl_obs", "soilwcl_obs", "stl_lm", "snow_depth", "cdland", "rhcapl", "sstcl_ob", "sicecl_ob",
           "ticecl_ob", "sstan_ob", "sst_om", "tice_om", "sice_om", "sstan_am", "sice_am", "tice_am", "ssti_om", "cdsea",
           "cdice", "rhcaps", "rhcapi", "hfseacl", "fmask_sea", "alb0", "orog", "phi0", "fmask_orig", "veg_high", "veg_low",
           "bmask_land", "bmask_sea"):
    SHAPES[_n] = (_F, (96, 48))

DELT = 86400.0 / 36  # params.f90:33

# boundary-condition file variable -> registry variable (pyspeedy/speedy.py:279-296)
BC_MAP = (("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"),
          ("stl12", "stl"), ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"), ("soil_wc_l3", "swl3"),
          ("sst12", "sst"), ("sea_ice_frac12", "icec"))


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

    # ---- lifecycle (pyspeedy/speedy.py:217-301, 375-405) ----------------------------------------------------
    def set_bc(self, bc, start_date=(1982, 1, 1, 0, 0), member=-1):
        """Load the 12 boundary fields (mapping like example_bc.nc, dims (lon, lat[, month])) and run the reference's
        `init`: land/sea preprocessing, rest atmosphere, coupler, forcing, first_step.  Zero SST anomaly unless
        `sst_anom` was set before."""
        for state_name, bc_name in BC_MAP:
            self.set(state_name, np.asarray(bc[bc_name], dtype=np.float64), member)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_init(self._m, *[int(v) for v in start_date], stream), "spd_model_init")

    def run(self, nsteps):
        """`nsteps` model steps (40 simulated minutes each) for every member; asynchronous."""
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_step(self._m, int(nsteps), stream), "spd_model_step")

    @property
    def current_step(self):
        return int(self._lib.spd_model_current_step(self._m))

    @property
    def current_date(self):
        buf = (C.c_int * 5)()
        check(self._lib.spd_model_get_date(self._m, buf), "spd_model_get_date")
        return tuple(buf)

    def profile(self, enable=True):
        check(self._lib.spd_model_profile(self._m, int(bool(enable))), "spd_model_profile")

    def profile_read(self):
        """(mean launch time in ms, number of launches, fields per launch) of the spectral->grid kernel since profile(True)."""
        ms, n, f = C.c_double(), C.c_int(), C.c_int()
        check(self._lib.spd_model_profile_read(self._m, C.byref(ms), C.byref(n), C.byref(f)), "spd_model_profile_read")
        return ms.value, n.value, f.value

    def set_flags(self, land_coupling_flag=True, sst_anomaly_coupling_flag=True, increase_co2=False):
        check(self._lib.spd_model_set_flags(self._m, int(land_coupling_flag), int(sst_anomaly_coupling_flag),
                                            int(increase_co2)), "spd_model_set_flags")

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
