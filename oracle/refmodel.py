"""ctypes front-end to oracle/_ref/libspeedy_ref.so (the flang-compiled *reference* Fortran).

TEST INFRASTRUCTURE -- not part of the product.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module, and only as checker / baseline.

It plays the role the f2py module ``pyspeedy.speedy_driver`` plays for the reference's Python layer
(pyspeedy/__init__.py:14): the generated driver procedures (speedy.f90/speedy_driver.f90) use
explicit-shape arguments passed by reference, so they are callable from C through their flang
module-procedure symbols ``_QMspeedy_driverP<name>``.  Operator-level entry points come from
oracle/ref_shim.f90 (bind(C) forwarding wrappers, no numerics).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libspeedy_ref.so")

IX, IL, KX, MX, NX, IY = 96, 48, 8, 31, 32, 24

# name -> (dtype, shape) of the registry arrays this harness touches (registry/model_state_def.py:121-495)
SHAPES = {
    "vor": (np.complex128, (MX, NX, KX, 2)), "div": (np.complex128, (MX, NX, KX, 2)),
    "t": (np.complex128, (MX, NX, KX, 2)), "ps": (np.complex128, (MX, NX, 2)),
    "tr": (np.complex128, (MX, NX, KX, 2, 1)), "phi": (np.complex128, (MX, NX, KX)),
    "phis": (np.complex128, (MX, NX)),
    "fband": (np.float64, (301, 4)),
    "xgeop1": (np.float64, (KX,)), "xgeop2": (np.float64, (KX,)),
    "rad_flux": (np.float64, (IX, IL, 4)), "rad_tau2": (np.float64, (IX, IL, KX, 4)),
    "rad_st4a": (np.float64, (IX, IL, KX, 2)), "rad_strat_corr": (np.float64, (IX, IL, 2)),
}
for _n in ("u_grid", "v_grid", "t_grid", "q_grid", "phi_grid", "tt_rsw"):
    SHAPES[_n] = (np.float64, (IX, IL, KX))
for _n in ("slru", "ustr", "vstr", "shf", "evap", "hfluxn"):
    SHAPES[_n] = (np.float64, (IX, IL, 3))
for _n in ("stl12", "snowd12", "soilw12", "soil_wc_l1", "soil_wc_l2", "soil_wc_l3", "sst12", "sea_ice_frac12",
           "sstom12"):
    SHAPES[_n] = (np.float64, (IX, IL, 12))
for _n in ("ps_grid", "precnv", "precls", "snowcv", "snowls", "cbmf", "tsr", "ssrd", "ssr", "slrd", "slr", "olr",
           "phi0", "orog", "phis0", "alb0", "forog", "fmask_orig", "veg_low", "veg_high",
           "flux_solar_in", "flux_ozone_lower", "flux_ozone_upper", "zenit_correction", "stratospheric_correction",
           "qcloud_equiv", "rhcapl", "cdland", "stlcl_obs", "snowdcl_obs", "soilwcl_obs", "land_temp", "snow_depth",
           "soil_avail_water", "stl_lm", "fmask_land", "bmask_land", "rhcaps", "rhcapi", "cdsea", "cdice",
           "fmask_sea", "bmask_sea", "hfseacl", "sstcl_ob", "sicecl_ob", "ticecl_ob", "sstan_ob", "sstcl_om",
           "sst_am", "sstan_am", "sice_am", "tice_am", "sst_om", "sice_om", "tice_om", "ssti_om", "wsst_ob",
           "alb_land", "alb_sea", "alb_surface", "snowc"):
    SHAPES[_n] = (np.float64, (IX, IL))

SCALARS = {"current_step": C.c_int, "increase_co2": C.c_int, "compute_shortwave": C.c_int,
           "air_absortivity_co2": C.c_double, "land_coupling_flag": C.c_int,
           "sst_anomaly_coupling_flag": C.c_int, "ablco2_ref": C.c_double}

BC_MAP = [("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"),
          ("stl12", "stl"), ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"),
          ("soil_wc_l3", "swl3"), ("sst12", "sst"), ("sea_ice_frac12", "icec")]  # pyspeedy/speedy.py:279-296


def available():
    return os.path.isfile(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _drv(name):
    return getattr(lib(), "_QMspeedy_driverP" + name)


def _f(a, dtype):
    return np.asfortranarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefModel:
    """One reference ModelState_t + ControlParams_t, driven like pyspeedy.Speedy drives them."""

    def __init__(self, start=(1982, 1, 1, 0, 0), end=(1982, 1, 4, 0, 0)):
        self.cnt = C.c_int64(0)
        _drv("modelstate_init")(C.byref(self.cnt))
        self._dates = []
        for d in (start, end):
            dc = C.c_int64(0)
            args = [C.byref(C.c_int(v)) for v in d]
            _drv("create_datetime")(*args, C.byref(dc))
            self._dates.append(dc)
        self.ctl = C.c_int64(0)
        _drv("controlparams_init")(C.byref(self.ctl), C.byref(self._dates[0]), C.byref(self._dates[1]))
        self.n_months = (end[0] - start[0]) * 12 + (end[1] - start[1]) + 1

    # -- registry access -------------------------------------------------
    def get(self, name):
        if name in SCALARS:
            v = SCALARS[name]()
            _drv("get_" + name)(C.byref(self.cnt), C.byref(v))
            return v.value
        dtype, shape = SHAPES[name]
        out = np.zeros(shape, dtype=dtype, order="F")
        _drv("get_" + name)(C.byref(self.cnt), _p(out))
        return out

    def set(self, name, value):
        if name in SCALARS:
            v = SCALARS[name](value)
            _drv("set_" + name)(C.byref(self.cnt), C.byref(v))
            return
        dtype, shape = SHAPES[name]
        a = _f(value, dtype)
        assert a.shape == shape, (name, a.shape, shape)
        _drv("set_" + name)(C.byref(self.cnt), _p(a))

    # -- lifecycle (pyspeedy/speedy.py:217-301) ---------------------------
    def set_bc(self, bc):
        """bc: mapping with the 12 example_bc fields; zero SST anomaly (sst_anomaly.nc is absent upstream)."""
        _drv("modelstate_init_sst_anom")(C.byref(self.cnt), C.byref(C.c_int(self.n_months)))
        for state_name, bc_name in BC_MAP:
            self.set(state_name, np.asarray(bc[bc_name], dtype=np.float64))
        err = C.c_int(0)
        _drv("init")(C.byref(self.cnt), C.byref(self.ctl), C.byref(err))
        if err.value != 0:
            raise RuntimeError("reference init failed: %d" % err.value)

    def step(self):
        err = C.c_int(0)
        _drv("step")(C.byref(self.cnt), C.byref(self.ctl), C.byref(err))
        return err.value

    def spectral2grid(self):
        _drv("transform_spectral2grid")(C.byref(self.cnt))

    def check(self):
        err = C.c_int(0)
        _drv("check")(C.byref(self.cnt), C.byref(err))
        return err.value

    # -- operator level (oracle/ref_shim.f90) ------------------------------
    def call(self, name, *args):
        """Call shim_<name>(cnt, *args); numpy arrays are passed by pointer, ints/floats by value."""
        fn = getattr(lib(), "shim_" + name)
        cargs = [self.cnt]
        for a in args:
            if isinstance(a, np.ndarray):
                assert a.flags.f_contiguous or a.ndim <= 1
                cargs.append(_p(a))
            elif isinstance(a, (int, np.integer)):
                cargs.append(C.c_int(int(a)))
            elif isinstance(a, float):
                cargs.append(C.c_double(a))
            else:
                raise TypeError(type(a))
        fn(*cargs)

    def spec2grid(self, spec, kcos=1):
        out = np.zeros((IX, IL), order="F")
        self.call("spec2grid", _f(spec, np.complex128), out, kcos)
        return out

    def grid2spec(self, grid):
        out = np.zeros((MX, NX), dtype=np.complex128, order="F")
        self.call("grid2spec", _f(grid, np.float64), out)
        return out
