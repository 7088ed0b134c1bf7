#!/usr/bin/env python3
"""How fast do nearly identical forecasts drift apart?  A small perturbed ensemble on one GPU.

Every member gets the same boundary fields and a tiny random change of its grid-point temperature; the members then run
together (one set of kernel launches per model step for the whole ensemble) and the script tabulates how the spread between
them grows, level by level and day by day, from the daily checkpoints kept in memory.

    python examples/ensemble_spread.py [--size 12] [--from 1982-01-01] [--days 10] [--quiet-days 1] [--noise 0.01] [--seed 7]

API surface used: iteration over `SpeedyEns`, in-place update of a state variable (`member["t_grid"] += ...`) followed by
`grid2spectral()`, `ModelCheckpoint` + `DiagnosticCheck` hooks, and the reductions of the checkpoint `Dataset`
(`var`, `std`, `mean`, `apply`, `isel`).  The same forecast over several GPUs: ensemble_one_process.py, ensemble_multi_gpu.py.
"""
import argparse
import os
import sys
from datetime import datetime, timedelta

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyspeedy_amd import SpeedyEns  # noqa: E402
from pyspeedy_amd.callbacks import DiagnosticCheck, ModelCheckpoint  # noqa: E402


def parse():
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("--size", type=int, default=12, help="number of members")
    p.add_argument("--from", dest="first_day", default="1982-01-01")
    p.add_argument("--days", type=int, default=10)
    p.add_argument("--quiet-days", type=int, default=1, help="days before the first checkpoint")
    p.add_argument("--noise", type=float, default=0.01, help="standard deviation of the temperature perturbation [K]")
    p.add_argument("--seed", type=int, default=7)
    return p.parse_args()


def perturb(ensemble, sigma, seed):
    """independent Gaussian noise on every member's grid-point temperature, then back to the spectral prognostics"""
    rng = np.random.default_rng(seed)
    for member in ensemble:
        member.set_bc()
        member["t_grid"] += rng.normal(0.0, sigma, size=member.get_shape("t_grid"))
        member.grid2spectral()


def main():
    args = parse()
    t0 = datetime.strptime(args.first_day, "%Y-%m-%d")
    t1 = t0 + timedelta(days=args.days)
    ensemble = SpeedyEns(args.size, start_date=t0, end_date=t1)
    perturb(ensemble, args.noise, args.seed)

    daily = ModelCheckpoint(interval=36, spinup_date=t0 + timedelta(days=args.quiet_days))
    watchdog = DiagnosticCheck(interval=72)  # the range check of the reference, every second day, on top of the per-step one
    ensemble.run(callbacks=[daily, watchdog])

    record = daily.dataframe  # dims (time, ens, lev, lat, lon); lev counts upwards from the surface
    print(record)

    # spread = square root of the ensemble variance, averaged over the whole domain
    domain_spread = record.var(dim="ens").mean(dim=["lev", "lat", "lon"]).apply(np.sqrt)
    print("domain-mean spread per kept day")
    for name in domain_spread:
        print("  %-4s %s" % (name, " ".join("%.4g" % v for v in domain_spread[name].values)))

    # the same at the lowest model level only, as a map statistic of the last day
    near_surface = record.std(dim="ens").isel(lev=0)
    last_day = near_surface["t"].values[-1]
    print("lowest level, last day: temperature spread max %.4g K, median %.4g K" % (last_day.max(), float(np.median(last_day))))
    growth = domain_spread["t"].values[-1] / domain_spread["t"].values[0]
    print("temperature spread grew by a factor %.2f over %d days" % (growth, len(domain_spread["t"].values) - 1))


if __name__ == "__main__":
    main()
