"""Generate tests/golden/export.npz from the flang-compiled REFERENCE (oracle/_ref/libspeedy_ref.so).

TEST INFRASTRUCTURE.  What the reference's own tests pin (pyspeedy/tests/test_speedy.py: 1-day and 3-day runs exported
as u, v, t, q, phi, ps on the grid) regenerated here from the same Fortran with ZERO SST anomalies (the reference's
sst_anomaly.nc is not distributed with it), plus the grid <-> spectral conversions of prognostics.f90:125-219:

  d1_<v>   float64  grid-space prognostics after 36 steps  (transform_spectral2grid)
  d3_<v>   float32  the same after 108 steps (what the reference's exporter would write, before the lev reversal)
  lon, lat, lev     float32 coordinates (initialization.f90:85-87)
  rt_<v>            spectral state (time level 1) produced by transform_grid2spectral from the d1 grid fields perturbed by
                    1e-3 relative noise (numpy default_rng(20260101), fields in export order; tests regenerate the input)
  gf_<v>            float64  apply_grid_filter of those perturbed grid fields (u_grid, t_grid, ps_grid kept)
  chk_zero_t        error code of `check` after t := 0 (test_exceptions)

Run in the build container:  python oracle/gen_golden_export.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refmodel as R  # noqa: E402

GRID = ("u_grid", "v_grid", "t_grid", "q_grid", "phi_grid", "ps_grid")


def main():
    bc = np.load(os.path.join(HERE, "..", "pyspeedy_amd", "data", "example_bc.npz"))
    m = R.RefModel(start=(1982, 1, 1, 0, 0), end=(1982, 1, 4, 0, 0))
    m.set_bc(bc)
    out = {}
    for name in ("lon", "lat", "lev"):
        n = {"lon": 96, "lat": 48, "lev": 8}[name]
        a = np.zeros(n, dtype=np.float32)
        R._drv("get_" + name)(C.byref(m.cnt), R._p(a))
        out[name] = a
    for _ in range(36):
        assert m.step() == 0
    m.spectral2grid()
    for v in GRID:
        out["d1_" + v] = m.get(v)
    # grid -> spectral round trip on perturbed fields, and the grid filter
    rng = np.random.default_rng(20260101)
    pert = {}
    for v in GRID:
        g = out["d1_" + v]
        pert[v] = g * (1.0 + 1e-3 * rng.standard_normal(g.shape))  # tests regenerate this from d1_<v> with the same seed
    state0 = {v: m.get(v) for v in ("vor", "div", "t", "tr", "ps", "phi")}
    for v in GRID:
        m.set(v, pert[v])
    R._drv("transform_grid2spectral")(C.byref(m.cnt))
    for v in ("vor", "div", "t", "tr", "ps", "phi"):
        a = m.get(v)
        out["rt_" + v] = a if v == "phi" else a[..., 0, 0] if v == "tr" else a[..., 0]  # time level 1
    R._drv("apply_grid_filter")(C.byref(m.cnt))
    for v in ("u_grid", "t_grid", "ps_grid"):
        out["gf_" + v] = m.get(v)
    # restore and continue to day 3
    for v, a in state0.items():
        m.set(v, a)
    for _ in range(72):
        assert m.step() == 0
    m.spectral2grid()
    for v in GRID:
        out["d3_" + v] = m.get(v).astype(np.float32)
    t = m.get("t")
    t[:] = 0
    m.set("t", t)
    out["chk_zero_t"] = np.int32(m.check())
    dst = os.path.join(HERE, "..", "tests", "golden", "export.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: (v.shape, str(v.dtype)) for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
