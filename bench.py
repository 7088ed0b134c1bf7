#!/usr/bin/env python3
"""Benchmark of pyspeedy_amd on MI355X: ensemble SPEEDY T30L8 (96x48x8, fp64), whole model step on the GPU.

    python bench.py --gpus N --steps K --warmup W [--members M]

One "step" = one call of the reference's do_single_step (speedy.f90:20-74) for EVERY one of the M ensemble members
resident on each GPU: daily forcing when due, shortwave every third step, the leapfrog step (91 spectral->grid transforms,
grid-point dynamics, fused column physics, 73 grid->spectral transforms, spectral tendencies, semi-implicit correction,
horizontal diffusion, Robert-Asselin-Williams filter), date advance and the land / sea-ice coupling -- 6 kernel launches.  Nothing crosses
PCIe inside a step; the state of all members stays in HBM.  The hot path of BASELINE.json (transforms + column physics)
is what dominates it (profiles/).

Workload: BASELINE.json cfg 4's ensemble, 64 members per GPU by default (weak scaling: every rank owns its own M members,
members never exchange data -- speedy_driver.f90.j2:71-77 -- so there is no data-path collective; the only collective is
the start-up broadcast of the boundary fields from rank 0, outside the timed region).  Members start from the
reference's own initial state (example boundary conditions, resting atmosphere, first_step) with a small temperature
perturbation per member, spun up for `warmup` steps.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEPS_PER_YEAR = 36 * 365  # model_control.f90:57-60, params.f90:32
S_BYTES, G_BYTES = 15872, 36864  # one spectral / one grid field


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=360)
    ap.add_argument("--warmup", type=int, default=36)
    ap.add_argument("--members", type=int, default=64, help="ensemble members resident per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def load_bc():
    return np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))


def build_ensemble(M, device, seed, dist, rank, coll_device):
    import pyspeedy_amd
    from pyspeedy_amd import ensemble as E
    from pyspeedy_amd.model import EnsembleModel
    sp = pyspeedy_amd.ModSpectral(device.index)
    model = EnsembleModel(sp, M)
    # rank 0 reads the boundary file; one RCCL broadcast (~3.4 MB over xGMI) hands it to the other GPUs
    bc = E.broadcast_boundary_conditions(dict(load_bc()) if rank == 0 else None, dist, coll_device)
    model.set_bc(bc, start_date=(1982, 1, 1, 0, 0))
    # member perturbations (examples/Ensemble_forecast.ipynb perturbs t_grid with N(0, 0.01) K): here a relative 1e-5
    # perturbation of the spectral temperature of both time levels, seed = global member id
    t0 = model.get("t", 0)
    for i in range(1, M):
        rng = np.random.default_rng(seed * 100003 + i)
        t = t0 * (1.0 + 1e-5 * rng.standard_normal((31, 32, 8, 1)))
        t[0] = t[0].real  # zonal-mean coefficients stay real
        model.set("t", t, member=i)
    return sp, model


def cpu_baseline(seconds):
    """The reference Fortran itself (flang-compiled, oracle/_ref/libspeedy_ref.so, built in the build container from
    /root/reference) timed on one host core: the same do_single_step for ONE member.  Falls back to the C port
    (oracle/liboracle.so) when the reference library did not travel."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    bc = load_bc()
    try:
        import refmodel as R
        if not R.available():
            raise OSError("no reference library")
        m = R.RefModel(end=(1983, 1, 1, 0, 0))
        m.set_bc(bc)
        for _ in range(6):
            m.step()
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(36):
                assert m.step() == 0
            n += 36
            el = time.perf_counter() - t0
            if el >= seconds:
                break
        kind, what = "reference", "reference Fortran (amdflang -O2) do_single_step"
    except (OSError, AttributeError):
        import oracle as orc
        orc.build()
        g = np.load(os.path.join(ROOT, "tests", "golden", "step.npz"))
        arr = {k[3:]: g[k] for k in g.files if k.startswith("s0_")}
        arr["tcorh"], arr["qcorh"] = g["tab_tcorh"], g["tab_qcorh"]
        st = orc.ModelState(arr, True, float(g["air_absortivity_co2"]))
        d = orc.dyn_tables(2 * 2400.0)
        n, t0 = 0, time.perf_counter()
        while True:
            st.set_shortwave(n % 3 == 0)
            orc.step(st, d, 2, 2, 2 * 2400.0)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds and n >= 3:
                break
        kind, what = "port", "C port (oracle/liboracle.so) of time_stepping.f90 step incl. transforms and physics"
    ms = el / n * 1e3
    return {"value": 86400.0 / (ms * 1e-3 * STEPS_PER_YEAR), "unit": "simulated-years/day (1 member)", "cores": 1,
            "kind": kind, "ms_per_member_step": ms,
            "sample": "%d model steps of one member with the %s in %.1f s on one host core" % (n, what, el)}


def main():
    args = parse()
    from pyspeedy_amd import ensemble as E
    world, rank, local = E.dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    local = local % torch.cuda.device_count()  # (several ranks may share a GPU in a rehearsal on a one-GPU box)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # RCCL ("nccl" on ROCm): barrier, max-over-ranks time and the start-up broadcast of the boundary fields.
    # PYSPEEDY_AMD_BENCH_BACKEND=gloo rehearses the same control flow when the ranks cannot each have their own GPU.
    backend = os.environ.get("PYSPEEDY_AMD_BENCH_BACKEND", "nccl")
    dist = E.init_process_group(backend, device)
    coll_device = device if backend == "nccl" else torch.device("cpu")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sp, model = build_ensemble(args.members, device, 1 + rank, dist, rank, coll_device)
    model.run(args.warmup)
    model.profile(True)
    barrier()
    t0 = time.perf_counter()
    model.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = E.max_over_ranks(elapsed, dist, coll_device)
    kern_ms, launches, nfields = model.profile_read()
    codes = model.check(2)
    if (codes != 0).any():
        raise SystemExit("bench.py: %d members left the accepted range (diagnostics.f90)" % int((codes != 0).sum()))

    achieved = (S_BYTES + G_BYTES) * nfields / (kern_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC measurement of this very kernel inside this bench (rocprofv3 cannot run
    # inside bench.py): 5824 fields per launch at 64 members, FETCH_SIZE doubled as the gfx950 guide prescribes; scaled
    # per field for other member counts.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_model_step.json")) as fh:
            tj = json.load(fh)["kernels"]["spd::spec2grid_table_kernel"]
        traffic = tj["hbm_bytes_per_launch"] * nfields / 5824.0
        traffic_src = "profiles/r01_pmc_model_step.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py, 5824 fields/launch)"
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        total_members = args.members * world
        value = E.simulated_years_per_day(total_members, ms_step * 1e-3, STEPS_PER_YEAR)
        line = {
            "metric": "simulated-years/day (whole node), T30L8", "value": value, "unit": "simulated-years/day",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "reference example_bc boundary fields (committed fixture, no download); state generated by the model: "
                    "resting atmosphere + first_step, per-member temperature perturbation, %d spin-up steps" % args.warmup,
            "config": {
                "workload": "BASELINE cfg 4 ensemble shard: %d members per GPU, T30L8 96x48x8 fp64, full do_single_step per "
                            "member (91 spec2grid + grid-point dynamics + fused column physics + 73 grid2spec + spectral "
                            "tendencies/semi-implicit/diffusion/RAW filter + coupler + daily forcing), all on the GPU"
                            % args.members,
                "members_per_gpu": args.members, "members_total": total_members,
                "ms_per_member_step": ms_step / args.members, "simulated_days": args.steps / 36.0,
                "parallelism": "ensemble members sharded per GPU, no collective",
            },
            "roofline": {
                "kernel": "spec2grid_table_kernel (inverse Legendre + inverse FFT-96, with vort2vel / gradient applied while staging "
                          "the 34 wind and pressure-gradient fields of each member), %d fields/launch" % nfields,
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "algorithmic_bytes_per_field": S_BYTES + G_BYTES, "avg_launch_ms": kern_ms, "launches_timed": launches,
                "traffic": traffic, "traffic_source": traffic_src,
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the host baseline is measured on single-GPU runs only
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
