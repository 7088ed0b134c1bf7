"""Boundary-field sets for the initialisation goldens (tests/golden/init.npz).  TEST INFRASTRUCTURE.

The example boundary file marks missing values with 9.97e36, which fill_missing_values (boundaries.f90:69-113, `value < 0`)
does not treat as missing: the reference's own input never takes that routine's branches.  `holes` is the example with
patterns added that do: scattered negative values, whole rows without a valid point (the running mean is then carried over
from the row before -- from the previous month at the first row visited, and from the land fields into the sea fields,
boundaries.f90:77), mask / albedo values on and next to the thresholds of land_model_init and sea_model_init, negative
vegetation and sea-ice fractions, soil water beyond saturation, non-zero SST anomalies on five planes (three are masked).
Used by oracle/gen_golden_init.py (the reference computes the expected outputs) and by the tests (the same inputs again).
"""
import numpy as np

IX, IL = 96, 48
# registry name -> key of the example boundary file (pyspeedy/speedy.py:279-296)
BC_MAP = [("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"),
          ("stl12", "stl"), ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"),
          ("soil_wc_l3", "swl3"), ("sst12", "sst"), ("sea_ice_frac12", "icec")]
N_MONTHS = 3  # 1982-01-01 .. 1982-03-04: sst_anom has n_months + 2 = 5 planes
START, END = (1982, 1, 1, 0, 0), (1982, 3, 4, 0, 0)
OUTPUTS = ("stl12", "snowd12", "soilw12", "sst12", "sea_ice_frac12", "sst_anom", "fmask_land", "bmask_land", "fmask_sea",
           "bmask_sea", "rhcapl", "cdland", "rhcaps", "rhcapi", "cdsea", "cdice")


def example(bc):
    """The registry inputs of the example boundary file, fp64, (ix, il[, 12]); zero SST anomaly."""
    f = {name: np.asfortranarray(bc[key], dtype=np.float64) for name, key in BC_MAP}
    f["sst_anom"] = np.zeros((IX, IL, N_MONTHS + 2), order="F")
    return f


def holes(bc, seed=20260410):
    f = example(bc)
    rng = np.random.default_rng(seed)
    stl, sst = f["stl12"], f["sst12"]
    stl[:, 10, 2] = -999.0                       # a whole row, mid-hemisphere
    stl[:, 23, 3] = -1.0                         # the first row visited: mean carried over from month 2's last row
    stl[:, 0:4, 5] = -999.0                      # the last rows of the southern sweep
    stl[:, :, 7][rng.random((IX, IL)) < 0.2] = -5.0
    stl[::2, 30, 8] = -999.0                     # every other point: both neighbours valid
    stl[1:, 31, 8] = -999.0                      # one valid point in the row
    stl[:, 47, 11] = -999.0                      # the very last row of the land sequence ...
    sst[:, 23, 0] = -999.0                       # ... hands its mean to the first row of the sea sequence
    sst[:, :, 4][rng.random((IX, IL)) < 0.3] = -0.5
    sst[:, 47, 9] = -999.0
    sst[0, :, 10] = -999.0                       # the wrap-around column
    sst[IX - 1, ::3, 10] = -999.0
    f32 = np.float32
    m = f["fmask_orig"]
    special = [0.1, float(f32(0.1)), float(np.nextafter(f32(0.1), f32(0))), 0.9, 1.0 - float(f32(0.1)),
               float(np.nextafter(1.0 - float(f32(0.1)), 2.0)), float(f32(1.0) / f32(3.0)), 1.0 - float(f32(1.0) / f32(3.0)),
               1.0 / 3.0, 2.0 / 3.0, 0.05, 0.95]
    for k, v in enumerate(special):
        m[5 + 7 * k, 20 + (k % 9)] = v
    f["alb0"][3, 5] = float(f32(0.4))
    f["alb0"][4, 5] = float(np.nextafter(f32(0.4), f32(0)))
    f["alb0"][5, 5] = 0.4
    f["veg_high"][rng.random((IX, IL)) < 0.05] = -0.3
    f["veg_low"][rng.random((IX, IL)) < 0.05] = -0.9
    f["soil_wc_l1"][:] = np.where(f["soil_wc_l1"] > 1e30, f["soil_wc_l1"], f["soil_wc_l1"] * 4.0)
    ice = f["sea_ice_frac12"]
    ice[:, :, 1][rng.random((IX, IL)) < 0.1] = -0.25
    ice[7, 7, 2] = -0.0
    f["snowd12"][:, :, 6][rng.random((IX, IL)) < 0.1] = -3.0
    f["sst_anom"][:] = rng.standard_normal(f["sst_anom"].shape)
    return f


CASES = {"example": example, "holes": holes}
