"""Model-state registry of the device-resident model: what `ModelState_t` exposes through the generated getters and
setters of the reference (registry/model_state_def.py:121-495 -> speedy_driver.f90.j2:250-334), restated as a table.

Every entry: name -> Var(dtype, shape, where, nc_dims, alt_name, units, long_name).
  shape   reference (Fortran-order) shape; `N_MONTHS` marks the SST-anomaly axis of length n_months + 2
  where   "device"  the array lives in HBM inside the spd_model (spd_model_get / spd_model_set)
          "table"   read-only table owned by the context (spd_get_table_host); float32 where the reference's is
          "scalar"  host-side control value of the model object
          "host"    allocated by the reference but never read by its step: kept as a plain host array
  nc_dims / alt_name / units / long_name feed the NetCDF export (pyspeedy/speedy.py:415-477).
"""
from collections import namedtuple

import numpy as np

Var = namedtuple("Var", "dtype shape where nc_dims alt_name units long_name")

IX, IL, KX, MX, NX = 96, 48, 8, 31, 32
N_MONTHS = "n_months+2"

_C, _F, _F4 = np.complex128, np.float64, np.float32

REGISTRY = {}


def _add(names, dtype, shape, where="device", nc_dims=None, units=None, long_names=None):
    for i, n in enumerate(names.split()):
        alt = n
        REGISTRY[n] = Var(dtype, shape, where, nc_dims, alt, units, long_names[i] if long_names else n)


_S2, _S3, _S4 = (MX, NX), (MX, NX, KX), (MX, NX, KX, 2)
_G2, _G3 = (IX, IL), (IX, IL, KX)
_LL, _LLK = ["lon", "lat"], ["lon", "lat", "lev"]

# prognostic spectral state
_add("vor div t tr", _C, _S4, nc_dims=["mx", "nx", "lev", "t_levs"])
_add("ps", _C, (MX, NX, 2), nc_dims=["mx", "nx", "t_levs"])
_add("phi", _C, _S3, nc_dims=["mx", "nx", "lev"])
_add("phis tcorh qcorh", _C, _S2, nc_dims=["mx", "nx"])
# grid-space prognostics in output units (the default export set)
for _n, _alt, _u, _ln in (("u_grid", "u", "m/s", "eastward_wind"), ("v_grid", "v", "m/s", "northward_wind"),
                          ("t_grid", "t", "K", "air_temperature"), ("q_grid", "q", None, "specific_humidity"),
                          ("phi_grid", "phi", None, "geopotential_height")):
    REGISTRY[_n] = Var(_F, _G3, "device", _LLK, _alt, _u, _ln)
REGISTRY["ps_grid"] = Var(_F, _G2, "device", _LL, "ps", None, "surface_air_pressure")
# physics: persisted radiation state, fluxes and diagnostics
_add("rad_st4a", _F, (IX, IL, KX, 2))
_add("rad_flux", _F, (IX, IL, 4))
_add("tt_rsw", _F, _G3, nc_dims=_LLK)
_add("rad_tau2", _F, (IX, IL, KX, 4))
_add("rad_strat_corr", _F, (IX, IL, 2))
_add("fmask_land phis0 forog sst_am alb_land alb_sea snowc land_temp soil_avail_water flux_solar_in flux_ozone_upper "
     "flux_ozone_lower zenit_correction stratospheric_correction alb_surface precnv precls cbmf slrd slr olr tsr ssrd ssr "
     "qcloud_equiv", _F, _G2, nc_dims=_LL)
_add("slru ustr vstr shf evap hfluxn", _F, (IX, IL, 3))
# boundary conditions and the land / sea / ice slab models
_add("stl12 snowd12 soilw12 sst12 sea_ice_frac12 soil_wc_l1 soil_wc_l2 soil_wc_l3", _F, (IX, IL, 12))
_add("sst_anom", _F, (IX, IL, N_MONTHS))
_add("stlcl_obs snowdcl_obs soilwcl_obs stl_lm snow_depth cdland rhcapl sstcl_ob sicecl_ob ticecl_ob sstan_ob sst_om "
     "tice_om sice_om sstan_am sice_am tice_am ssti_om cdsea cdice rhcaps rhcapi hfseacl fmask_sea alb0 orog phi0 "
     "fmask_orig veg_high veg_low bmask_land bmask_sea", _F, _G2, nc_dims=_LL)
# allocated by the reference, never read by its time step
_add("snowcv snowls sstcl_om wsst_ob", _F, _G2, where="host", nc_dims=_LL)
_add("sstom12", _F, (IX, IL, 12), where="host")
# tables
REGISTRY["lon"] = Var(_F4, (IX,), "table", ["lon"], "lon", "degrees_east", "longitude")
REGISTRY["lat"] = Var(_F4, (IL,), "table", ["lat"], "lat", "degrees_north", "latitude")
REGISTRY["lev"] = Var(_F4, (KX,), "table", ["lev"], "lev", None, "Vertical sigma coordinate")
_add("deglat_s", _F, (IL,), where="table")
_add("fband", _F, (301, 4), where="table")
_add("xgeop1 xgeop2", _F, (KX,), where="table")
# scalars
for _n, _t in (("current_step", np.int32), ("increase_co2", np.bool_), ("compute_shortwave", np.bool_),
               ("air_absortivity_co2", _F), ("land_coupling_flag", np.bool_), ("sst_anomaly_coupling_flag", np.bool_),
               ("ablco2_ref", _F)):
    REGISTRY[_n] = Var(_t, None, "scalar", None, _n, None, _n)

DEFAULT_OUTPUT_VARS = ("u_grid", "v_grid", "t_grid", "q_grid", "phi_grid", "ps_grid")  # pyspeedy/__init__.py:18-25


def shape_of(name, n_months=1):
    """Concrete shape of a registry array (sst_anom depends on the simulated period)."""
    v = REGISTRY[name]
    if v.shape is None:
        return ()
    return tuple(n_months + 2 if s == N_MONTHS else s for s in v.shape)


def is_array(name):
    return REGISTRY[name].shape is not None


_FORTRAN_TYPE = {np.dtype(np.complex128): "complex(8)", np.dtype(np.float64): "real(8)", np.dtype(np.float32): "real",
                 np.dtype(np.int32): "integer", np.dtype(np.bool_): "logical"}
_DIM_NAME = {IX: "ix", IL: "il", KX: "kx", MX: "mx", NX: "nx"}


def model_state_def():
    """The registry in the layout of the reference's `pyspeedy/data/model_state.json` (what `pyspeedy.speedy.MODEL_STATE_DEF`
    holds: name -> dtype, dims, desc, time_dim, units, nc_dims, alt_name, std_name), generated from the table above: host code
    that looks variables up there (`MODEL_STATE_DEF[var]["alt_name"]`, `["nc_dims"]`, `["time_dim"]`) keeps working.  `dims` is
    the Fortran shape with the reference's dimension names where a size has one; `desc` is the long name used in the export."""
    out = {}
    for name, v in REGISTRY.items():
        timed = v.shape is not None and N_MONTHS in v.shape
        dims = None
        if v.shape is not None:
            dims = "(" + ", ".join("0:n_months+1" if s == N_MONTHS else _DIM_NAME.get(s, str(s)) for s in v.shape) + ")"
        nc_dims = None if v.nc_dims is None else ["0:n_months+1" if (timed and i == len(v.shape) - 1) else d
                                                  for i, d in enumerate(v.nc_dims)]
        out[name] = {"dtype": _FORTRAN_TYPE[np.dtype(v.dtype)], "dims": dims, "desc": v.long_name,
                     "time_dim": "n_months" if timed else None, "units": v.units, "nc_dims": nc_dims, "alt_name": v.alt_name,
                     "std_name": name}
    return out
