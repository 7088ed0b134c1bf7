"""CPU tier: the host side of the model's calendar and daily forcing (pyspeedy_amd/csrc/surface_host.cpp, through the C ABI's
host-only entry points spd_calendar_walk / spd_daily_forcing_host -- no device is touched) against the reference Fortran
(tests/golden/calendar.npz, oracle/gen_golden_calendar.py).  Bitwise.

* model_control.f90:79-185 after every step of four runs: leap February 1980 (the reference notebooks' period), the 1982/83 year
  end, two January runs;
* get_zonal_average_fields (shortwave_radiation.f90:218-322) for every day of the 365-day year;
* the leap rule against the Gregorian calendar over 1979-1981 (Python's datetime; mod(year, 4) agrees with it there);
* the reference library's own stderr text for a state out of range (diagnostics.f90:69-70) -- the text the product's driver
  prints (tests/test_speedy_gpu.py::test_exceptions asserts it on the GPU).
"""
import ctypes as C
import os
import re
import subprocess
import sys
from datetime import datetime, timedelta

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STARTS = {"leap": (1980, 2, 26, 0, 0), "newyear": (1982, 12, 29, 0, 0), "land_off": (1982, 1, 1, 0, 0),
          "ssta_off": (1982, 1, 1, 0, 0)}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir + "/calendar.npz")


def walk(lib, start, n):
    ymdhm = np.zeros((n + 1, 5), dtype=np.int32)
    month_idx, imont1 = np.zeros(n + 1, dtype=np.int32), np.zeros(n + 1, dtype=np.int32)
    tmonth, tyear = np.zeros(n + 1), np.zeros(n + 1)
    rc = lib.spd_calendar_walk(*start, n, ymdhm.ctypes.data, month_idx.ctypes.data, imont1.ctypes.data, tmonth.ctypes.data,
                               tyear.ctypes.data)
    assert rc == 0
    return ymdhm, month_idx, imont1, tmonth, tyear


@pytest.mark.parametrize("name", list(STARTS))
def test_calendar_after_every_step_bitwise(hip_lib, gold, name):
    n = len(gold[name + "_cal_month_idx"]) - 1
    ymdhm, month_idx, imont1, tmonth, tyear = walk(hip_lib, STARTS[name], n)
    assert np.array_equal(ymdhm, gold[name + "_cal_ymdhm"])
    assert np.array_equal(month_idx, gold[name + "_cal_month_idx"])
    assert np.array_equal(imont1, gold[name + "_cal_imont1"])
    assert np.array_equal(tmonth.view(np.uint64), gold[name + "_cal_tmonth"].view(np.uint64))
    assert np.array_equal(tyear.view(np.uint64), gold[name + "_cal_tyear"].view(np.uint64))


def test_golden_reaches_the_edges(gold):
    leap = [tuple(r) for r in gold["leap_cal_ymdhm"][::36]]
    assert (1980, 2, 29, 0, 0) in leap and leap[-1] == (1980, 3, 2, 0, 0)
    assert gold["leap_cal_tmonth"].max() > 1.0  # Feb 29: (29 - 0.5) / 28, model_control.f90:182
    assert list(gold["leap_cal_month_idx"][::36]) == [1, 1, 1, 1, 2, 2]
    ny = gold["newyear_cal_ymdhm"]
    assert tuple(ny[108]) == (1983, 1, 1, 0, 0) and tuple(ny[107]) == (1982, 12, 31, 23, 20)
    assert list(gold["newyear_cal_imont1"][::36]) == [12, 12, 12, 1, 1, 1] and list(gold["newyear_cal_month_idx"][::36]) == [1, 1, 1, 2, 2, 2]
    assert gold["newyear_cal_tyear"][107] > 0.998 and gold["newyear_cal_tyear"][108] < 0.002


def test_leap_rule_against_the_gregorian_calendar(hip_lib):
    start = datetime(1979, 12, 30)
    n = 36 * 400  # through February 1980 (29 days) and February 1981 (28)
    ymdhm, month_idx, _, _, _ = walk(hip_lib, (start.year, start.month, start.day, 0, 0), n)
    for s in range(0, n + 1, 7):
        d = start + s * timedelta(minutes=40)
        assert tuple(ymdhm[s]) == (d.year, d.month, d.day, d.hour, d.minute), s
    last = start + n * timedelta(minutes=40)
    assert month_idx[-1] == 1 + (last.year - start.year) * 12 + last.month - start.month


def test_a_century_year_bitwise(hip_lib, gold):
    """1899-12-30 -> 1901-01-03 after every one of 13 320 steps against the reference's advance_date alone: its leap rule is
    mod(year, 4) == 0 and nothing else, so the MODEL's 1900 has a February 29 (the Gregorian calendar has none), and month_idx
    counts through 14 months."""
    n = len(gold["century_cal_month_idx"]) - 1
    ymdhm, month_idx, imont1, tmonth, tyear = walk(hip_lib, (1899, 12, 30, 0, 0), n)
    assert np.array_equal(ymdhm, gold["century_cal_ymdhm"]) and np.array_equal(month_idx, gold["century_cal_month_idx"])
    assert np.array_equal(imont1, gold["century_cal_imont1"])
    assert np.array_equal(tmonth.view(np.uint64), gold["century_cal_tmonth"].view(np.uint64))
    assert np.array_equal(tyear.view(np.uint64), gold["century_cal_tyear"].view(np.uint64))
    days = {tuple(r[:3]) for r in gold["century_cal_ymdhm"]}
    assert (1900, 2, 29) in days and (1901, 2, 29) not in days and gold["century_cal_month_idx"][-1] == 14


def test_daily_forcing_every_day_of_the_year_bitwise(hip_lib, gold):
    out = np.zeros((5, 48))
    for d, tyear in enumerate(gold["forcing_tyear"]):
        assert hip_lib.spd_daily_forcing_host(C.c_double(float(tyear)), out.ctypes.data) == 0
        assert np.array_equal(out.view(np.uint64), gold["forcing_fields"][d].view(np.uint64)), "day %d of the year" % (d + 1)
    # the sweep's tyear values are the calendar's own
    _, _, _, _, tyear = walk(hip_lib, (1982, 1, 1, 0, 0), 36 * 364)
    assert np.array_equal(tyear[::36], gold["forcing_tyear"])


REF_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1] + "/oracle")
import refmodel as R
bc = np.load(sys.argv[1] + "/pyspeedy_amd/data/example_bc.npz")
hot = R.RefModel()
hot.set_bc(bc)
t = hot.get("t")
t[0, 0] *= 2.0
hot.set("t", t)
print(hot.step())
cold = R.RefModel()
cold.set_bc(bc)
for _ in range(36):
    assert cold.step() == 0
t = cold.get("t")
t[:] = 0
cold.set("t", t)
print(cold.check())
"""


def test_reference_stderr_text_on_a_range_failure():
    """diagnostics.f90:69-70 as the compiled reference prints it (list-directed: a leading blank; flang puts one blank before the
    integer where gfortran right-justifies it in 12 columns -- the product prints gfortran's form, both match the pattern)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import refmodel as R
    if not R.available():
        pytest.skip("oracle/_ref/libspeedy_ref.so not built")
    r = subprocess.run([sys.executable, "-c", REF_CHILD, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["-2", "-2"]
    lines = r.stderr.splitlines()
    assert lines[0] == lines[2] == " Model variables out of accepted range"
    steps = [re.fullmatch(r" step =\s+(\d+)", ln) for ln in (lines[1], lines[3])]
    assert all(steps) and [int(m.group(1)) for m in steps] == [1, 36]
    for product_line, step in ((" step =%12d" % 1, 1), (" step =%12d" % 36, 36)):  # what csrc/driver.cpp writes
        assert int(re.fullmatch(r" step =\s+(\d+)", product_line).group(1)) == step
