"""Generate the long-horizon goldens from the flang-compiled REFERENCE (one unperturbed member, example boundary conditions,
zero SST anomaly, 1982-01-01 + one year = 13 140 steps).  TEST INFRASTRUCTURE.

  tests/golden/run30.npz        the state after 1080 steps (30 days): prognostic spectra at time level 1 and surface fields --
                                the horizon at which SURVEY.md section 8c found two builds of the reference itself 2.7e-14 apart
  tests/golden/climate_year.npz zonal means sampled at 00:00 of each of the last 60 days of the year (days 306 ... 365):
                                t_grid, u_grid (lat, lev), precipitation precnv + precls (lat), and their time means; the
                                global-mean temperature at 500 hPa-ish level 4 at the end of every month

Run in the build container (about three minutes):  python oracle/gen_golden_year.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refmodel as R  # noqa: E402

STEPS_PER_DAY, DAYS = 36, 365
SAMPLE_FROM_DAY = 306


def zonal(a):
    """(lon, lat[, lev]) -> (lat[, lev])"""
    return np.asarray(a).mean(axis=0)


def main():
    bc = np.load(os.path.join(HERE, "..", "pyspeedy_amd", "data", "example_bc.npz"))
    m = R.RefModel(start=(1982, 1, 1, 0, 0), end=(1983, 1, 2, 0, 0))
    m.set_bc(bc)
    gold = os.path.join(HERE, "..", "tests", "golden")
    t_z, u_z, p_z, month_t = [], [], [], []
    for day in range(1, DAYS + 1):
        for _ in range(STEPS_PER_DAY):
            assert m.step() == 0, "reference left the accepted range on day %d" % day
        if day == 30:
            out = {}
            for v in ("vor", "div", "t", "ps"):
                out[v] = m.get(v)[..., 0]
            out["tr"] = m.get("tr")[..., 0, 0]
            for v in ("land_temp", "sst_am", "tice_am", "snowc", "olr", "tsr"):
                out[v] = m.get(v)
            np.savez_compressed(os.path.join(gold, "run30.npz"), **out)
            print("wrote run30.npz", flush=True)
        if day % 30 == 0 or day >= SAMPLE_FROM_DAY:
            m.spectral2grid()
            t = m.get("t_grid")
            if day % 30 == 0:
                month_t.append(float(zonal(t)[:, 3].mean()))
            if day >= SAMPLE_FROM_DAY:
                t_z.append(zonal(t))
                u_z.append(zonal(m.get("u_grid")))
                p_z.append(zonal(m.get("precnv") + m.get("precls")))
    t_z, u_z, p_z = np.array(t_z), np.array(u_z), np.array(p_z)
    np.savez_compressed(os.path.join(gold, "climate_year.npz"), t_zonal_mean=t_z.mean(axis=0), u_zonal_mean=u_z.mean(axis=0),
                        precip_zonal_mean=p_z.mean(axis=0), t_zonal_daily_std=t_z.std(axis=0), u_zonal_daily_std=u_z.std(axis=0),
                        precip_zonal_daily_std=p_z.std(axis=0), t_level4_monthly=np.array(month_t),
                        sample_days=np.arange(SAMPLE_FROM_DAY, DAYS + 1))
    print("wrote climate_year.npz")


if __name__ == "__main__":
    main()
