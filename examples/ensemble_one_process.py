#!/usr/bin/env python3
"""The reference's own shape of an ensemble forecast (examples/Ensemble_forecast.ipynb: ONE Python process, `SpeedyEns`, every
member stepped by one `parallel_step`) on all GPUs of a node:

    python examples/ensemble_one_process.py --members 64 --devices 8 --days 3 --out out/

`SpeedyEns(n, devices=k)` places the members in blocks on GPUs 0 .. k-1; `ens.set_bc()` reads the boundary conditions once,
hands them to the other GPUs with one RCCL broadcast over xGMI (local copies on each GPU) and initialises every device model
in one pass; `ens.run()` drives all GPUs with one parallel_step per model step, each GPU's launches issued by a host thread of
its own.  Members are perturbed with seed = member id, so the forecast is the one examples/ensemble_multi_gpu.py (one process
per GPU) produces.  The ensemble mean and spread of the temperature are formed on the GPU at every output time and written as
NetCDF-3.
"""
import argparse
import os
import sys
from datetime import datetime, timedelta

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspeedy_amd import ensemble as E  # noqa: E402
from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.callbacks import BaseCallback  # noqa: E402
from pyspeedy_amd.dataset import Dataset, Variable  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402


class TemperatureStatistics(BaseCallback):
    """daily ensemble mean and spread of t_grid, computed on the device over all members whatever GPU they live on"""

    def __init__(self, output_dir, interval=36):
        super().__init__(interval=interval)
        self.output_dir = output_dir

    def fire(self, ens):
        mean, spread = E.ensemble_mean_spread(ens.device_view("t_grid", spectral2grid=True), None)
        m0, dims = ens.members[0], ("time", "lev", "lat", "lon")
        ds = Dataset({"t_mean": Variable(dims, mean.cpu().numpy()[None, ::-1].astype(np.float32)),
                      "t_spread": Variable(dims, spread.cpu().numpy()[None, ::-1].astype(np.float32))},
                     {"time": Variable(("time",), np.array([np.datetime64(ens.current_date, "s")])),
                      "lev": Variable(("lev",), m0["lev"][::-1].copy()), "lat": Variable(("lat",), m0["lat"]),
                      "lon": Variable(("lon",), m0["lon"])})
        path = os.path.join(self.output_dir, ens.current_date.strftime("tstat_%Y-%m-%d_%H%M.nc"))
        ds.to_netcdf(path)
        self.print_msg("%s: max spread %.4f K -> %s" % (ens.current_date, float(spread.max()), path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=16)
    ap.add_argument("--devices", type=int, default=0, help="GPUs to spread the members over (0: all this process can see)")
    ap.add_argument("--days", type=int, default=1)
    ap.add_argument("--out", default="ensemble_out")
    args = ap.parse_args()
    devices = args.devices or drv.device_count()
    os.makedirs(args.out, exist_ok=True)
    start = datetime(1982, 1, 1)
    ens = SpeedyEns(args.members, start_date=start, end_date=start + timedelta(days=args.days), devices=devices)
    ens.set_bc()
    for i, member in enumerate(ens):
        t = member["t_grid"]
        member["t_grid"] = t + np.random.default_rng(i).normal(0.0, 0.01, t.shape)
        member.grid2spectral()
    ens.run(callbacks=[TemperatureStatistics(args.out)])
    torch.cuda.synchronize()
    placed = sorted({drv.modelstate_device(m._state_cnt) for m in ens})
    print("%d members on device(s) %s, boundary hand-over (peer copies, local copies, GPUs reached collectively) = %s, %d steps"
          % (args.members, placed, drv.broadcast_boundary_stats(), ens.get_current_step()))


if __name__ == "__main__":
    main()
