#!/usr/bin/env python3
"""A perturbed-initial-condition ensemble forecast sharded over the GPUs of one node (cf. the reference's
examples/Ensemble_forecast.ipynb, which steps its members with OpenMP on the CPU).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/ensemble_multi_gpu.py \\
        --members 64 --days 3 --out out/

One process per GPU.  Rank 0 reads the boundary conditions and broadcasts them (one RCCL broadcast over xGMI); every rank
builds ONE batched device model for its block of members, perturbs them (seed = global member id, so the forecast does not
depend on the number of GPUs), and runs; members never exchange data.  At every output time the ensemble mean and spread of
the temperature are formed on the GPUs (two all-reduces of one field) and rank 0 writes them as NetCDF-3.
PYSPEEDY_AMD_BACKEND=gloo rehearses the same program when the ranks have to share a GPU.
"""
import argparse
import os
import sys
from datetime import datetime, timedelta

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd  # noqa: E402
from pyspeedy_amd import ensemble as E  # noqa: E402
from pyspeedy_amd import speedy_driver as drv  # noqa: E402
from pyspeedy_amd.dataset import Dataset, Variable  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=16, help="ensemble size over all GPUs")
    ap.add_argument("--days", type=int, default=1)
    ap.add_argument("--out", default="ensemble_out")
    args = ap.parse_args()

    world, rank, local = E.dist_env()
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = os.environ.get("PYSPEEDY_AMD_BACKEND", "nccl")
    dist = E.init_process_group(backend, device)
    coll = device if backend == "nccl" else torch.device("cpu")

    bc = None
    if rank == 0:
        with np.load(pyspeedy_amd.example_bc_file()) as z:
            bc = {k: z[k] for k in z.files}
    bc = E.broadcast_boundary_conditions(bc, dist, coll)

    first, count = E.shard_members(args.members, world, rank)
    start = datetime(1982, 1, 1)
    ens = SpeedyEns(count, start_date=start, end_date=start + timedelta(days=args.days))
    for i, member in enumerate(ens):
        member.member_id = first + i
        member.set_bc(bc_file=bc)
        rng = np.random.default_rng(first + i)  # global member id: independent of the sharding
        t = member["t_grid"]
        member["t_grid"] = t + rng.normal(0.0, 0.01, t.shape)
        member.grid2spectral()

    def write_statistics(_ens):
        if _ens.get_current_step() % 36:
            return
        view = _ens.device_view("t_grid", spectral2grid=True)  # [local members, lev, lat, lon] in HBM
        if backend != "nccl":
            view = view.cpu()
        mean, spread = E.ensemble_mean_spread(view, dist)
        if rank == 0:
            m0 = _ens.members[0]
            ds = Dataset({"t_mean": Variable(("time", "lev", "lat", "lon"), mean.cpu().numpy()[None, ::-1].astype(np.float32)),
                          "t_spread": Variable(("time", "lev", "lat", "lon"), spread.cpu().numpy()[None, ::-1].astype(np.float32))},
                         {"time": Variable(("time",), np.array([np.datetime64(_ens.current_date, "s")])),
                          "lev": Variable(("lev",), m0["lev"][::-1].copy()), "lat": Variable(("lat",), m0["lat"]),
                          "lon": Variable(("lon",), m0["lon"])})
            os.makedirs(args.out, exist_ok=True)
            path = os.path.join(args.out, _ens.current_date.strftime("tstat_%Y-%m-%d_%H%M.nc"))
            ds.to_netcdf(path)
            print("%s: %d members on %d rank(s), max spread %.4f K -> %s" % (_ens.current_date, args.members, world,
                                                                              float(spread.max()), path), flush=True)

    ens.run(callbacks=[write_statistics])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
