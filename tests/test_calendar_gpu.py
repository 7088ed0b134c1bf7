"""GPU tier: the calendar's edge cases and the two runtime coupling flags against the reference Fortran
(tests/golden/calendar.npz, oracle/gen_golden_calendar.py):

  leap      1980-02-26 -> 1980-03-02: February keeps 29 days when mod(year, 4) == 0 (model_control.f90:135-142), tmonth exceeds 1
            on Feb 29 (:182 divides by 28) -- the period the reference's own notebooks run over;
  newyear   1982-12-29 -> 1983-01-03: month 12 -> 1 with the year carried (:153-157), month_idx keeps counting into the SST
            anomaly planes, tyear wraps, CO2 trend on;
  land_off  land_coupling_flag = .false. (land_model.f90:179-186);   ssta_off  sst_anomaly_coupling_flag = .false.
            (sea_model.f90:218-222, 279) with a non-zero anomaly in the state.

Every case runs twice: container by container through the C boundary (`spd_step`, the reference's `step(state_cnt,
control_cnt)`), with the control container's date and month_idx compared after EVERY step, and through the facade
(`Speedy.run()` with a daily callback).  Tolerance: 1e-10 of each field's max norm (<= 180 steps; fp64); the zonal forcing
profiles 1e-13 (they are bitwise on the host, tests/test_calendar_host.py).  The largest error seen is written to
gpurun_out/calendar_parity.txt when that directory exists.
"""
import os
from datetime import datetime

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")

CASES = {  # name: (start, end, anomaly?, {state scalar: value})   -- oracle/gen_golden_calendar.py: CASES
    "leap": (datetime(1980, 2, 26), datetime(1980, 3, 2), True, {}),
    "newyear": (datetime(1982, 12, 29), datetime(1983, 1, 3), True, {"increase_co2": True}),
    "land_off": (datetime(1982, 1, 1), datetime(1982, 1, 3), False, {"land_coupling_flag": False}),
    "ssta_off": (datetime(1982, 1, 1), datetime(1982, 1, 3), True, {"sst_anomaly_coupling_flag": False}),
}
ZONAL = ("flux_solar_in", "flux_ozone_lower", "flux_ozone_upper", "zenit_correction", "stratospheric_correction")
SPECTRAL = {"vor": lambda a: a[..., 0], "div": lambda a: a[..., 0], "t": lambda a: a[..., 0], "ps": lambda a: a[..., 0],
            "tr": lambda a: a[..., 0], "phi": lambda a: a}  # (the facade's tr is (mx, nx, kx, 2): one tracer)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "calendar.npz"))


def months_before_and_after(start, end):
    """first day of every month from (start - 1 month) to (end + 1 month): the window pyspeedy/speedy.py:338-372 selects"""
    first = start.year * 12 + start.month - 2
    last = end.year * 12 + end.month
    return np.array(["%04d-%02d-01" % (k // 12, k % 12 + 1) for k in range(first, last + 1)], dtype="datetime64[s]")


def synthetic_ssta(lat, planes):
    i = np.arange(96)[:, None, None]
    t = np.arange(planes)[None, None, :]
    return 1.5 * np.sin(2 * np.pi * i / 96 + 0.7 * t) * np.cos(np.deg2rad(lat.astype(np.float64)))[None, :, None] + 0.3 * t - 0.4


def make_model(gold, name):
    from pyspeedy_amd.speedy import Speedy
    start, end, anomaly, scalars = CASES[name]
    model = Speedy(start_date=start, end_date=end)
    for k, v in scalars.items():
        model[k] = v
    planes = int(gold[name + "_planes"])
    times = months_before_and_after(start, end)
    assert len(times) == planes
    ssta = {"ssta": synthetic_ssta(gold["lat"], planes), "time": times} if anomaly else None
    model.set_bc(sst_anomaly=ssta)
    assert model.get_shape("sst_anom") == (96, 48, planes)
    return model


def close(got, ref, what, tol=1e-10):
    scale = np.abs(ref).max()
    err = np.abs(np.asarray(got) - ref).max() / (scale if scale > 0 else 1.0)
    assert err <= tol, "%s: scaled max error %.3e" % (what, err)
    return err


WORST = {}  # case -> largest scaled error seen (written to gpurun_out/ by the last test, for DESIGN.md)


def compare_day(model, gold, name, day):
    worst = _compare_day(model, gold, name, day)
    WORST[name] = max(WORST.get(name, 0.0), worst)
    return worst


def _compare_day(model, gold, name, day):
    p = "%s_d%d_" % (name, day)
    assert model.get_current_step() == int(gold[p + "current_step"]) == 36 * day
    worst = 0.0
    for key in gold.files:
        if not key.startswith(p):
            continue
        var = key[len(p):]
        if var == "current_step":
            continue
        ref = gold[key]
        if var == "air_absortivity_co2":
            assert abs(model[var] - float(ref)) <= 1e-13 * abs(float(ref)), var
        elif var in SPECTRAL:
            worst = max(worst, close(SPECTRAL[var](model[var]), ref, key))
        elif var in ZONAL:
            field = model[var]
            assert np.all(field == field[:1, :]), var
            worst = max(worst, close(field[0], ref, key, tol=1e-13))
        else:
            worst = max(worst, close(model[var], ref, key))
    return worst


@pytest.mark.parametrize("name", list(CASES))
def test_step_by_step_through_the_c_boundary(gold, name):
    """`spd_step` container by container: the control container carries the reference's date and month_idx after every step,
    the state matches the reference at every day."""
    from pyspeedy_amd import speedy_driver as drv
    model = make_model(gold, name)
    ymdhm, month_idx = gold[name + "_cal_ymdhm"], gold[name + "_cal_month_idx"]
    n = len(month_idx) - 1
    date, idx = drv.get_model_datetime(model._control_cnt)
    assert tuple(date) == tuple(ymdhm[0]) and idx == month_idx[0] == 1
    for s in range(1, n + 1):
        assert drv.step(model._state_cnt, model._control_cnt) == 0
        date, idx = drv.get_model_datetime(model._control_cnt)
        assert tuple(date) == tuple(ymdhm[s]), (name, s)
        assert idx == month_idx[s], (name, s)
        if s % 36 == 0:
            compare_day(model, gold, name, s // 36)
    if name == "leap":
        days = [tuple(d[:3]) for d in ymdhm[::36]]
        assert (1980, 2, 29) in days and days[-1] == (1980, 3, 2)
    if name == "newyear":
        assert tuple(ymdhm[-1][:3]) == (1983, 1, 3) and month_idx[-1] == 2


@pytest.mark.parametrize("name", list(CASES))
def test_facade_run(gold, name):
    """`Speedy.run()` (the time loop of pyspeedy/speedy.py:396-405 over parallel_step begin / end) with a callback once a day."""
    model = make_model(gold, name)
    seen = []

    def daily(m):
        if m.get_current_step() % 36 == 0 and (m.current_date.hour, m.current_date.minute) == (0, 0):
            seen.append((m.current_date, compare_day(m, gold, name, m.get_current_step() // 36)))

    model.run(callbacks=[daily])
    ymdhm = gold[name + "_cal_ymdhm"]
    assert [d for d, _ in seen] == [datetime(*[int(v) for v in row]) for row in ymdhm[36::36]]
    assert model.current_date == CASES[name][1]


def test_flags_matter(gold):
    """the two flag cases are not vacuous: with the flags at their defaults the same runs end somewhere else"""
    base = make_model(gold, "land_off")
    base["land_coupling_flag"] = True
    base.run()
    assert np.abs(base["land_temp"] - gold["land_off_d2_land_temp"]).max() > 0.05
    assert np.array_equal(gold["land_off_d2_land_temp"], gold["land_off_d2_stlcl_obs"])
    assert np.all(gold["ssta_off_d2_sstan_am"] == 0) and np.abs(gold["ssta_off_d2_sstan_ob"]).max() == 0


def test_zz_report():
    out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    assert set(WORST) == set(CASES)
    if os.path.isdir(out):
        with open(os.path.join(out, "calendar_parity.txt"), "w") as f:
            for name, err in WORST.items():
                f.write("%-9s largest scaled error over all days and fields: %.3e (tolerance 1e-10)\n" % (name, err))
