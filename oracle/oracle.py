"""ctypes front-end to oracle/liboracle.so (plain-C restatement of the reference hot path).

TEST INFRASTRUCTURE -- not part of the product.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module, and only as checker / baseline.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

IX, IL, KX, MX, NX, IY = 96, 48, 8, 31, 32, 24


class Tables(C.Structure):
    _fields_ = [
        ("hsg", C.c_double * 9), ("dhs", C.c_double * 8), ("fsg", C.c_double * 8), ("dhsr", C.c_double * 8),
        ("fsgr", C.c_double * 8),
        ("radang", C.c_double * 48), ("coriol", C.c_double * 48), ("sia", C.c_double * 48), ("coa", C.c_double * 48),
        ("sia_half", C.c_double * 24), ("coa_half", C.c_double * 24), ("cosgr", C.c_double * 48),
        ("cosgr2", C.c_double * 48),
        ("sigl", C.c_double * 8), ("sigh", C.c_double * 9), ("grdsig", C.c_double * 8), ("grdscp", C.c_double * 8),
        ("wvi", C.c_double * 16),
        ("epsi", C.c_double * (32 * 33)), ("repsi", C.c_double * (32 * 33)), ("cpol", C.c_double * (62 * 32 * 24)),
        ("wt", C.c_double * 24), ("nsh2", C.c_int * 32),
        ("work", C.c_double * 96), ("ifac", C.c_int * 15),
        ("el2", C.c_double * 992), ("elm2", C.c_double * 992), ("el4", C.c_double * 992), ("trfilt", C.c_double * 992),
        ("gradym", C.c_double * 992), ("gradyp", C.c_double * 992), ("uvdx", C.c_double * 992),
        ("uvdym", C.c_double * 992), ("uvdyp", C.c_double * 992), ("vddym", C.c_double * 992),
        ("vddyp", C.c_double * 992), ("gradx", C.c_double * 31),
        ("fband", C.c_double * (301 * 4)),
    ]


TABLE_SHAPES = {
    "wvi": (8, 2), "epsi": (32, 33), "repsi": (32, 33), "cpol": (62, 32, 24), "fband": (301, 4),
    **{k: (31, 32) for k in ("el2", "elm2", "el4", "trfilt", "gradym", "gradyp", "uvdx", "uvdym", "uvdyp", "vddym",
                             "vddyp")},
}


class PhysIO(C.Structure):
    _fields_ = (
        [(n, C.c_void_p) for n in ("ug", "vg", "tg", "qg_in", "phig", "pslg", "utend", "vtend", "ttend", "qtend",
                                   "fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc",
                                   "land_temp", "soil_avail_water",
                                   "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction",
                                   "stratospheric_correction", "alb_surface")]
        + [("air_absortivity_co2", C.c_double), ("compute_shortwave", C.c_int)]
        + [(n, C.c_void_p) for n in ("precnv", "precls", "cbmf", "slrd", "slr", "olr",
                                     "slru", "ustr", "vstr", "shf", "evap", "hfluxn", "rad_st4a", "rad_flux",
                                     "tt_rsw", "rad_tau2", "rad_strat_corr", "tsr", "ssrd", "ssr", "qcloud_equiv",
                                     "iptop", "icltop", "ts", "tskin", "u0", "v0", "t0", "cloudc", "clstr")]
    )


def build():
    """(Re)build liboracle.so with gcc.  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", HERE, "liboracle.so"], check=True)


_lib = None
_tables = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def tables():
    global _tables
    if _tables is None:
        _tables = Tables()
        lib().orc_tables_init(C.byref(_tables))
    return _tables


def table(name):
    """Oracle table as a Fortran-ordered numpy array (copy)."""
    t = tables()
    a = np.ctypeslib.as_array(getattr(t, name)).copy()
    if name in TABLE_SHAPES:
        a = a.reshape(TABLE_SHAPES[name], order="F")
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _call(name, *args):
    fn = getattr(lib(), "orc_" + name)
    cargs = [C.byref(tables())]
    for a in args:
        if isinstance(a, np.ndarray):
            cargs.append(_p(a))
        elif isinstance(a, (int, np.integer)):
            cargs.append(C.c_int(int(a)))
        else:
            raise TypeError(type(a))
    fn(*cargs)


def _r(a):
    return np.asfortranarray(a, dtype=np.float64)


def _c(a):
    return np.asfortranarray(a, dtype=np.complex128)


def legendre_inv(x):
    out = np.zeros((62, 48), order="F")
    _call("legendre_inv", _r(x), out)
    return out


def legendre(x):
    out = np.zeros((62, 32), order="F")
    _call("legendre", _r(x), out)
    return out


def fourier_inv(x, kcos=1):
    out = np.zeros((96, 48), order="F")
    _call("fourier_inv", _r(x), out, kcos)
    return out


def fourier(x):
    out = np.zeros((62, 48), order="F")
    _call("fourier", _r(x), out)
    return out


def spec2grid(spec, kcos=1):
    out = np.zeros((96, 48), order="F")
    _call("spec2grid", _c(spec), out, kcos)
    return out


def grid2spec(grid):
    out = np.zeros((31, 32), dtype=np.complex128, order="F")
    _call("grid2spec", _r(grid), out)
    return out


def spec2grid_batch(spec, kcos=1):
    """spec: C-contiguous [B, 32, 31] complex (i.e. Fortran (31,32) per field) -> [B, 48, 96]."""
    spec = np.ascontiguousarray(spec, dtype=np.complex128)
    B = spec.shape[0]
    out = np.zeros((B, 48, 96))
    _call("spec2grid_batch", spec, out, kcos, B)
    return out


def grid2spec_batch(grid):
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    B = grid.shape[0]
    out = np.zeros((B, 32, 31), dtype=np.complex128)
    _call("grid2spec_batch", grid, out, B)
    return out


def vort2vel(vor, div):
    u = np.zeros((31, 32), dtype=np.complex128, order="F")
    v = np.zeros_like(u)
    _call("vort2vel", _c(vor), _c(div), u, v)
    return u, v


def vel2vort(u, v):
    vor = np.zeros((31, 32), dtype=np.complex128, order="F")
    div = np.zeros_like(vor)
    _call("vel2vort", _c(u), _c(v), vor, div)
    return vor, div


def grid_vel2vort(ug, vg, kcos):
    vor = np.zeros((31, 32), dtype=np.complex128, order="F")
    div = np.zeros_like(vor)
    _call("grid_vel2vort", _r(ug), _r(vg), vor, div, kcos)
    return vor, div


def gradient(psi):
    dx = np.zeros((31, 32), dtype=np.complex128, order="F")
    dy = np.zeros_like(dx)
    _call("gradient", _c(psi), dx, dy)
    return dx, dy


def laplacian(x, inverse=False):
    out = np.zeros((31, 32), dtype=np.complex128, order="F")
    _call("laplacian", _c(x), out, int(inverse))
    return out


def truncate(x):
    out = _c(x).copy(order="F")
    _call("truncate", out)
    return out


def grid_filter(g):
    out = np.zeros((96, 48), order="F")
    _call("grid_filter", _r(g), out)
    return out


# ---------------------------------------------------------------------------------------------------
# column physics
# ---------------------------------------------------------------------------------------------------
PHYS_IN_3D = ("ug", "vg", "tg", "qg_in", "phig")
PHYS_IN_2D = ("pslg", "fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp",
              "soil_avail_water", "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction",
              "stratospheric_correction", "alb_surface")
PHYS_TEND = ("utend", "vtend", "ttend", "qtend")
PHYS_OUT_SHAPES = {
    "precnv": (96, 48), "precls": (96, 48), "cbmf": (96, 48), "slrd": (96, 48), "slr": (96, 48), "olr": (96, 48),
    "slru": (96, 48, 3), "ustr": (96, 48, 3), "vstr": (96, 48, 3), "shf": (96, 48, 3), "evap": (96, 48, 3),
    "hfluxn": (96, 48, 3), "rad_st4a": (96, 48, 8, 2), "rad_flux": (96, 48, 4),
}
PHYS_PERSIST_SHAPES = {
    "tt_rsw": (96, 48, 8), "rad_tau2": (96, 48, 8, 4), "rad_strat_corr": (96, 48, 2), "tsr": (96, 48),
    "ssrd": (96, 48), "ssr": (96, 48), "qcloud_equiv": (96, 48),
}
PHYS_DIAG_F = ("ts", "tskin", "u0", "v0", "t0", "cloudc", "clstr")
PHYS_DIAG_I = ("iptop", "icltop")


def physics(inputs, compute_shortwave, air_absortivity_co2):
    """Run orc_physics on one member.

    inputs: dict with PHYS_IN_3D (96,48,8), PHYS_IN_2D (96,48), PHYS_TEND (96,48,8) and -- for a non-shortwave
    step -- the persisted fields of PHYS_PERSIST_SHAPES (rad_flux is fully rewritten by the longwave scheme).
    Returns dict of all outputs (tendencies, PHYS_OUT_SHAPES, PHYS_PERSIST_SHAPES, diagnostics), Fortran order.
    """
    io = PhysIO()
    keep = {}

    def put(name, arr):
        keep[name] = arr
        setattr(io, name, arr.ctypes.data_as(C.c_void_p).value)

    for n in PHYS_IN_3D + PHYS_IN_2D:
        put(n, np.asfortranarray(inputs[n], dtype=np.float64))
    for n in PHYS_TEND:
        put(n, np.array(inputs[n], dtype=np.float64, order="F", copy=True))
    for n, shp in PHYS_OUT_SHAPES.items():
        put(n, np.zeros(shp, order="F"))
    for n, shp in PHYS_PERSIST_SHAPES.items():
        if n in inputs:
            put(n, np.array(inputs[n], dtype=np.float64, order="F", copy=True))
        else:
            put(n, np.zeros(shp, order="F"))
    for n in PHYS_DIAG_F:
        put(n, np.zeros((96, 48), order="F"))
    for n in PHYS_DIAG_I:
        put(n, np.zeros((96, 48), dtype=np.int32, order="F"))
    io.air_absortivity_co2 = float(air_absortivity_co2)
    io.compute_shortwave = int(bool(compute_shortwave))
    lib().orc_physics(C.byref(tables()), C.byref(io))
    out = {n: keep[n] for n in PHYS_TEND + tuple(PHYS_OUT_SHAPES) + tuple(PHYS_PERSIST_SHAPES) + PHYS_DIAG_F + PHYS_DIAG_I}
    return out


# ---------------------------------------------------------------------------------------------------
# dynamics / time stepping (callers of the hot path)
# ---------------------------------------------------------------------------------------------------

def land_sea_init(fields, fmean=0.0):
    """land_model_init + sea_model_init (orc_surface.c) on a dict of registry inputs (oracle/init_cases.py); returns the dict of
    the 16 outputs and the running mean fill_missing_values is left with."""
    f = {k: np.array(v, dtype=np.float64, order="F") for k, v in fields.items()}
    n_planes = f["sst_anom"].shape[2] if "sst_anom" in f else 0
    anom = f.get("sst_anom", np.zeros((96, 48, 0), order="F"))
    out = {k: np.zeros((96, 48), order="F") for k in ("fmask_land", "bmask_land", "fmask_sea", "bmask_sea", "rhcapl", "cdland",
                                                      "rhcaps", "rhcapi", "cdsea", "cdice")}
    out["soilw12"] = np.zeros((96, 48, 12), order="F")
    carry = np.array([fmean], dtype=np.float64)
    _call("land_sea_init", n_planes, f["fmask_orig"], f["alb0"], f["veg_high"], f["veg_low"], f["soil_wc_l1"], f["soil_wc_l2"],
          f["stl12"], f["snowd12"], f["sst12"], f["sea_ice_frac12"], anom, out["soilw12"], out["fmask_land"], out["bmask_land"],
          out["fmask_sea"], out["bmask_sea"], out["rhcapl"], out["cdland"], out["rhcaps"], out["rhcapi"], out["cdsea"],
          out["cdice"], carry)
    for k in ("stl12", "snowd12", "sst12", "sea_ice_frac12"):
        out[k] = f[k]
    out["sst_anom"] = anom
    return out, float(carry[0])


class DynTables(C.Structure):
    _fields_ = [(n, C.c_double * 992) for n in ("dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s")] + [
        ("tcorv", C.c_double * 8), ("qcorv", C.c_double * 8), ("tref", C.c_double * 8), ("tref2", C.c_double * 8),
        ("tref3", C.c_double * 8), ("dhsx", C.c_double * 8), ("xc", C.c_double * 64), ("xd", C.c_double * 64),
        ("xj", C.c_double * 4096), ("elz", C.c_double * 992), ("xgeop1", C.c_double * 8), ("xgeop2", C.c_double * 8)]


DYN_SHAPES = {**{k: (31, 32) for k in ("dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s", "elz")},
              "xc": (8, 8), "xd": (8, 8), "xj": (8, 8, 64)}


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("vor", "div", "t", "tr", "ps", "phi", "phis", "tcorh", "qcorh")] + [("ph", PhysIO)]


STATE_SPEC = {"vor": (31, 32, 8, 2), "div": (31, 32, 8, 2), "t": (31, 32, 8, 2), "tr": (31, 32, 8, 2), "ps": (31, 32, 2),
              "phi": (31, 32, 8), "phis": (31, 32), "tcorh": (31, 32), "qcorh": (31, 32)}


def dyn_tables(dt=None):
    """orc_dyn_tables after initialisation and (optionally) set_time_step(dt)."""
    d = DynTables()
    lib().orc_dyn_tables_init(C.byref(tables()), C.byref(d))
    if dt is not None:
        lib().orc_dyn_set_time_step(C.byref(tables()), C.byref(d), C.c_double(dt))
    return d


def dyn_table(d, name):
    a = np.ctypeslib.as_array(getattr(d, name)).copy()
    if name in DYN_SHAPES:
        a = a.reshape(DYN_SHAPES[name], order="F")
    return a


class ModelState:
    """Host copy of the part of ModelState_t that one model step touches (Fortran layouts)."""

    def __init__(self, arrays, compute_shortwave, air_absortivity_co2):
        self.a = {}
        self.c = State()
        for n, shp in STATE_SPEC.items():
            v = np.array(arrays[n], dtype=np.complex128, order="F", copy=True) if n in arrays else np.zeros(shp, np.complex128, order="F")
            assert v.shape == shp, (n, v.shape)
            self.a[n] = v
            setattr(self.c, n, v.ctypes.data_as(C.c_void_p).value)
        ph = self.c.ph
        f2 = PHYS_IN_2D[1:]  # everything but pslg
        for n in f2:
            self._put(ph, n, np.array(arrays[n], dtype=np.float64, order="F", copy=True))
        for n, shp in {**PHYS_OUT_SHAPES, **PHYS_PERSIST_SHAPES}.items():
            self._put(ph, n, np.array(arrays[n], dtype=np.float64, order="F", copy=True) if n in arrays else np.zeros(shp, order="F"))
        ph.air_absortivity_co2 = float(air_absortivity_co2)
        ph.compute_shortwave = int(bool(compute_shortwave))

    def _put(self, ph, n, v):
        self.a[n] = v
        setattr(ph, n, v.ctypes.data_as(C.c_void_p).value)

    def set_shortwave(self, flag):
        self.c.ph.compute_shortwave = int(bool(flag))


def step(state, dyn, j1, j2, dt):
    lib().orc_step(C.byref(tables()), C.byref(dyn), C.byref(state.c), C.c_int(j1), C.c_int(j2), C.c_double(dt))


def get_tendencies(state, dyn, j2):
    out = {n: np.zeros((31, 32, 8), np.complex128, order="F") for n in ("vordt", "divdt", "tdt", "trdt")}
    out["psdt"] = np.zeros((31, 32), np.complex128, order="F")
    lib().orc_get_tendencies(C.byref(tables()), C.byref(dyn), C.byref(state.c), _p(out["vordt"]), _p(out["divdt"]),
                             _p(out["tdt"]), _p(out["psdt"]), _p(out["trdt"]), C.c_int(j2))
    return out


def check_diagnostics(state, time_lev):
    diag = np.zeros((8, 3), order="F")
    lib().orc_check_diagnostics.restype = C.c_int
    rc = lib().orc_check_diagnostics(C.byref(tables()), C.byref(state.c), C.c_int(time_lev), _p(diag))
    return rc, diag


# ---- the whole model (orc_model.c) ---------------------------------------------------------------------------------------
MODEL_SHAPES = {"vor": (31, 32, 8, 2), "div": (31, 32, 8, 2), "t": (31, 32, 8, 2), "tr": (31, 32, 8, 2), "ps": (31, 32, 2),
                "phi": (31, 32, 8), "phis": (31, 32), "tcorh": (31, 32), "qcorh": (31, 32)}  # complex; everything else is real
BC_MAP = (("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"), ("stl12", "stl"),
          ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"), ("soil_wc_l3", "swl3"), ("sst12", "sst"),
          ("sea_ice_frac12", "icec"))  # pyspeedy/speedy.py:279-296


class Model:
    """One whole oracle model: boundary fields in, `init`, `step` (do_single_step), registry arrays out by name."""

    def __init__(self, n_months=1):
        L = lib()
        L.orc_model_new.restype = C.c_void_p
        L.orc_model_field.restype = C.POINTER(C.c_double)
        L.orc_model_field.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
        L.orc_model_get_scalar.restype = C.c_double
        L.orc_model_get_scalar.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_model_set_scalar.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.orc_model_init.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.orc_model_step.argtypes = [C.c_void_p]
        L.orc_model_free.argtypes = [C.c_void_p]
        L.orc_model_calendar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.planes = n_months + 2
        self._m = C.c_void_p(L.orc_model_new(self.planes))

    def __del__(self):
        if getattr(self, "_m", None):
            lib().orc_model_free(self._m)
            self._m = None

    def _view(self, name):
        n = C.c_long(0)
        p = lib().orc_model_field(self._m, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def get(self, name):
        if name in ("current_step", "air_absortivity_co2", "ablco2_ref", "compute_shortwave"):
            return lib().orc_model_get_scalar(self._m, name.encode())
        flat = self._view(name).copy()
        if name in MODEL_SHAPES:
            return flat.view(np.complex128).reshape(MODEL_SHAPES[name], order="F")
        return flat.reshape((96, 48) + ((flat.size // 4608,) if flat.size > 4608 else ()), order="F")

    def set(self, name, value):
        if name in ("land_coupling_flag", "sst_anomaly_coupling_flag", "increase_co2", "air_absortivity_co2"):
            assert lib().orc_model_set_scalar(self._m, name.encode(), float(value)) == 0
            return
        view = self._view(name)
        a = np.asfortranarray(value, dtype=np.complex128 if name in MODEL_SHAPES else np.float64)
        flat = a.reshape(-1, order="F").view(np.float64)
        assert flat.size == view.size, (name, flat.size, view.size)
        view[:] = flat

    def set_bc(self, bc, sst_anom=None):
        for state_name, bc_name in BC_MAP:
            self.set(state_name, np.asarray(bc[bc_name], dtype=np.float64))
        if sst_anom is not None:
            self.set("sst_anom", sst_anom)

    def init(self, year, month, day, hour=0, minute=0):
        return lib().orc_model_init(self._m, year, month, day, hour, minute)

    def step(self):
        return lib().orc_model_step(self._m)

    def calendar(self):
        ymdhm = (C.c_int * 5)()
        mi, im, tm, ty = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        lib().orc_model_calendar(self._m, ymdhm, C.byref(mi), C.byref(im), C.byref(tm), C.byref(ty))
        return list(ymdhm), mi.value, im.value, tm.value, ty.value
