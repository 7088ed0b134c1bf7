#!/usr/bin/env python3
"""One deterministic forecast on one GPU, start to finish, through the pySPEEDY-shaped facade.

What it shows: a `Speedy` object over the MI355X backend, two hooks riding along in `run()` -- daily NetCDF-3 files once a
spin-up period is over, and the same days kept in memory --, and direct access to the state by registry name afterwards.

    python examples/winter_week.py [--from 1982-12-20] [--days 9] [--discard 3] [--dir ./winter_week]

The defaults cross the 1982/83 year end on purpose: the calendar, the monthly interpolation of the boundary climatologies and
the solar forcing all change year there (tests/test_calendar_gpu.py checks that stretch against the reference Fortran).
"""
import argparse
import glob
import os
import sys
from datetime import datetime, timedelta

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyspeedy_amd import Speedy  # noqa: E402
from pyspeedy_amd.callbacks import ModelCheckpoint, XarrayExporter  # noqa: E402

STEPS_PER_DAY = 36  # the model step is 40 minutes


def parse():
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("--from", dest="first_day", default="1982-12-20")
    p.add_argument("--days", type=int, default=9, help="length of the forecast")
    p.add_argument("--discard", type=int, default=3, help="days at the start that are not written (spin-up from rest)")
    p.add_argument("--dir", default="./winter_week", help="where the daily files go")
    return p.parse_args()


def band_mean(field, lat, south, north):
    """area-weighted mean of a (lon, lat) field between two latitudes"""
    rows = (lat >= south) & (lat <= north)
    weights = np.cos(np.deg2rad(lat[rows]))
    return float((field[:, rows].mean(axis=0) * weights).sum() / weights.sum())


def main():
    args = parse()
    t0 = datetime.strptime(args.first_day, "%Y-%m-%d")
    t1 = t0 + timedelta(days=args.days)
    keep_from = t0 + timedelta(days=args.discard)

    forecast = Speedy(start_date=t0, end_date=t1)
    forecast.set_bc()  # packaged climatological boundary fields; the atmosphere starts at rest

    to_disk = XarrayExporter(output_dir=args.dir, interval=STEPS_PER_DAY, spinup_date=keep_from, verbose=False)
    in_memory = ModelCheckpoint(interval=STEPS_PER_DAY, spinup_date=keep_from, verbose=False)
    print("writing", ", ".join(to_disk.variables), "to", args.dir)
    forecast.run(callbacks=[to_disk, in_memory])

    files = sorted(glob.glob(os.path.join(args.dir, "*.nc")))
    history = in_memory.dataframe
    print("forecast ended %s after %d steps; %d daily files, %d days in memory"
          % (forecast.current_date, forecast.get_current_step(), len(files), history.dims["time"]))

    # the state stays on the GPU; model["name"] copies one registry variable out, (lon, lat[, lev]) with lev top-down
    lat = forecast["lat"].astype(np.float64)
    lowest_t = forecast["t_grid"][:, :, -1]
    surface_p = forecast["ps_grid"]
    print("grid: %d x %d, first latitude %.3f" % (forecast["lon"].size, lat.size, lat[0]))
    for name, south, north in (("southern extratropics", -90, -30), ("tropics", -30, 30), ("northern extratropics", 30, 90)):
        print("  %-22s T(lowest level) %6.1f K   p(surface) %7.1f hPa"
              % (name, band_mean(lowest_t, lat, south, north), band_mean(surface_p, lat, south, north) / 100))
    day_to_day = np.abs(np.diff(history["t"].values, axis=0)).mean(axis=(1, 2, 3))
    print("mean |dT| from one kept day to the next [K]:", " ".join("%.2f" % v for v in day_to_day))


if __name__ == "__main__":
    main()
