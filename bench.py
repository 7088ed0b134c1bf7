#!/usr/bin/env python3
"""Benchmark of the MI355X-native SPEEDY hot path (spectral transforms + fused column physics), T30L8, fp64.

    python bench.py --gpus N --steps K --warmup W [--members M]

One "step" = one pass of the hot path of ONE model time step over the M ensemble members resident on each GPU,
issued through the C ABI exactly as a host model would (SURVEY.md App. C transform census of do_single_step):

    16M vort2vel + 1M gradient                      (spectral.f90:190-214, 275-296)
    91M spec2grid  = 57M with kcos=1 + 34M kcos=2   (tendencies.f90:109-146, physics.f90:89-101)
    column physics of M members, shortwave every 3rd step   (physics.f90:107-256)
    73M grid2spec  = 24M grid_vel2vort pairs (48M, +24M vel2vort) + 25M plain   (tendencies.f90:148, 242-266)
    16M laplacian                                   (tendencies.f90:247-249, 345-349)

Inputs are synthetic but physically plausible fields resident in HBM before the timed region.  The grid-point
dynamics algebra, implicit solver and time filter that sit between these calls in the reference are NOT part of this
hot path (SURVEY.md section 8f "next") and are not executed; `config.workload` says so.  Members are independent
(speedy_driver.f90.j2:71-77): every rank runs its own M members, no data-path collective -> weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
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
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--members", type=int, default=64, help="ensemble members resident per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def physical_member_grids(M, device, seed):
    """Plausible grid-point state for M members (device layout): u, v, T, q, phi [M,8,48,96], ln ps [M,48,96]."""
    import pyspeedy_amd.physics as P
    base = P.synthetic_member(seed=seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for n in P.STATE_IN_3D + P.STATE_IN_2D + P.SURFACE_IN + P.SHORTWAVE_IN:
        a = torch.from_numpy(P.to_device_layout(base[n]))
        rep = a[None].repeat(M, *([1] * a.dim()))
        if n in ("tg", "ug", "vg"):
            rep = rep + 0.01 * torch.randn(rep.shape, generator=g, dtype=torch.float64)  # member perturbations
        out[n] = rep.to(device).contiguous()
    return out


class HotPath:
    """All buffers are allocated once and laid out so that every batched call reads/writes them in place: the model
    state lives inside the transform batch buffers ([variable][member][level] order), physics reads the transform
    outputs directly and writes its tendencies straight into the forward-transform input batch."""

    def __init__(self, M, device, seed):
        import pyspeedy_amd
        import pyspeedy_amd.physics as P
        self.M, self.P = M, P
        self.sp = pyspeedy_amd.ModSpectral(device.index)
        self.phys = P.ColumnPhysics(self.sp)
        self.lib, self.h = self.sp._lib, self.sp.handle
        f64 = dict(dtype=torch.float64, device=device)
        c128 = dict(dtype=torch.complex128, device=device)
        sp = self.sp
        # ---- inverse batches.  kcos=1: [vor2, div2, t2, q2, t1, q1, phi][M][8] + ps1[M] = 57M fields
        self.in_k1 = torch.empty((57 * M, 32, 31), **c128)
        self.st1 = self.in_k1[:56 * M].view(7, M, 8, 32, 31)
        self.ps1 = self.in_k1[56 * M:]
        # kcos=2: [dyn|phys][ucos|vcos][M][8] + grad(ln ps)[2][M] = 34M fields, written by vort2vel / gradient
        self.in_k2 = torch.empty((34 * M, 32, 31), **c128)
        self.sv = self.in_k2[:32 * M].view(2, 2, M, 8, 32, 31)
        self.gps = self.in_k2[32 * M:].view(2, M, 32, 31)
        self.g_k1 = torch.empty((57 * M, 48, 96), **f64)
        self.g_k2 = torch.empty((34 * M, 48, 96), **f64)
        g1 = self.g_k1[:56 * M].view(7, M, 8, 48, 96)
        g2 = self.g_k2[:32 * M].view(2, 2, M, 8, 48, 96)
        self.fields = {"ug": g2[1, 0], "vg": g2[1, 1], "tg": g1[4], "qg": g1[5], "phig": g1[6],
                       "pslg": self.g_k1[56 * M:]}
        # ---- forward batches.  grid_vel2vort pairs: [utend|vtend, -uT|-vT, -uq|-vq][M][8] = 24M pairs
        self.fw_uv = torch.empty((2, 24 * M, 48, 96), **f64)
        # plain grid2spec: [ttend, qtend, KE][M][8] + ps tendency [M] = 25M fields
        self.fw_sc = torch.empty((25 * M, 48, 96), **f64)
        self.tend = {"utend": self.fw_uv[0, :8 * M].view(M, 8, 48, 96), "vtend": self.fw_uv[1, :8 * M].view(M, 8, 48, 96),
                     "ttend": self.fw_sc[:8 * M].view(M, 8, 48, 96), "qtend": self.fw_sc[8 * M:16 * M].view(M, 8, 48, 96)}
        self.o_vor = torch.empty((24 * M, 32, 31), **c128)
        self.o_div = torch.empty((24 * M, 32, 31), **c128)
        self.o_sc = torch.empty((25 * M, 32, 31), **c128)
        self.o_lap = torch.empty((16 * M, 32, 31), **c128)
        # ---- synthetic but physical content: analyse plausible grids into the spectral state
        grids = physical_member_grids(M, device, seed)
        self.st1[4].copy_(sp.grid2spec(grids["tg"])); self.st1[5].copy_(sp.grid2spec(grids["qg"]))
        self.st1[6].copy_(sp.grid2spec(grids["phig"])); self.ps1.copy_(sp.grid2spec(grids["pslg"]))
        ucos, vcos = sp.grid2spec(grids["ug"] * 0.7), sp.grid2spec(grids["vg"] * 0.7)
        vor, div = sp.vel2vort(ucos, vcos)
        self.vor1, self.div1 = vor.contiguous(), div.contiguous()           # time level 1 (physics)
        self.st1[0].copy_(vor * 1.01); self.st1[1].copy_(div * 0.99)         # time level 2 (dynamics)
        self.st1[2].copy_(self.st1[4] * 1.001); self.st1[3].copy_(self.st1[5] * 0.999)
        self.ps2 = (self.ps1 * 1.0).contiguous()
        self.forcing = {n: grids[n] for n in P.SURFACE_IN + P.SHORTWAVE_IN}
        self.state = P.PhysicsState(M, device)
        g = torch.Generator(device=device).manual_seed(seed)
        self.fw_uv.normal_(generator=g).mul_(1e-4)
        self.fw_sc.normal_(generator=g).mul_(1e-4)
        self.step_no = 0
        self.ev = []  # (start, end) events around the dominant kernel
        self.step(record=False, force_sw=True)  # prime the persisted radiation state with one shortwave step
        self.step_no = 0

    def step(self, record=True, force_sw=False):
        M, L, h = self.M, self.lib, self.h
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        ck = self._ck
        # spectral-space pre-processing: u,v from vor,div at both time levels; grad ln ps
        ck(L.spd_vort2vel(h, p(self.st1[0]), p(self.st1[1]), p(self.sv[0, 0]), p(self.sv[0, 1]), 8 * M, st))
        ck(L.spd_vort2vel(h, p(self.vor1), p(self.div1), p(self.sv[1, 0]), p(self.sv[1, 1]), 8 * M, st))
        ck(L.spd_gradient(h, p(self.ps2), p(self.gps[0]), p(self.gps[1]), M, st))
        # the 91 inverse transforms of every member, two launches
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ck(L.spd_spec2grid(h, p(self.in_k1), p(self.g_k1), 1, 57 * M, st))
        if record:
            e1.record()
            self.ev.append((e0, e1))
        ck(L.spd_spec2grid(h, p(self.in_k2), p(self.g_k2), 2, 34 * M, st))
        # column physics on the time-level-1 grids, tendencies accumulated in the forward-transform input batch
        sw = force_sw or (self.step_no % 3 == 0)  # speedy.f90:53
        self.phys(self.fields, self.tend, self.forcing, self.state, sw, 0.3)
        # the 73 forward transforms: 24 grid_vel2vort pairs (48 transforms + 24 vel2vort) + 25 scalars
        ck(L.spd_grid_vel2vort(h, p(self.fw_uv[0]), p(self.fw_uv[1]), p(self.o_vor), p(self.o_div), 2, 24 * M, st))
        ck(L.spd_grid2spec(h, p(self.fw_sc), p(self.o_sc), 25 * M, st))
        ck(L.spd_laplacian(h, p(self.o_sc), p(self.o_lap), 0, 16 * M, st))
        self.step_no += 1

    @staticmethod
    def _ck(rc):
        if rc != 0:
            from pyspeedy_amd._lib import check
            check(rc, "hot path call")


def cpu_baseline(seconds):
    """The CPU oracle (plain-C restatement, bitwise-pinned to the flang-built reference) on the same per-member-step
    workload, one core.  Bounded sample: whole member-steps until `seconds` of CPU time are spent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    import pyspeedy_amd.physics as P
    orc.build()
    m = P.synthetic_member(seed=0)
    o_in = {("qg_in" if k == "qg" else k): v for k, v in m.items()}
    prev = orc.physics(o_in, True, 0.3)
    for k in orc.PHYS_PERSIST_SHAPES:
        o_in[k] = prev[k]
    rng = np.random.default_rng(0)
    spec = rng.standard_normal((91, 32, 31)) + 1j * rng.standard_normal((91, 32, 31))
    grid = rng.standard_normal((73, 48, 96))
    a, b = spec[0].T.copy(), spec[1].T.copy()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(16):
            orc.vort2vel(a, b)
        orc.gradient(a)
        orc.spec2grid_batch(spec[:57], 1)
        orc.spec2grid_batch(spec[57:], 2)
        orc.physics(o_in, n % 3 == 0, 0.3)
        orc.grid2spec_batch(grid)
        for _ in range(24):
            orc.vel2vort(a, b)
        for _ in range(16):
            orc.laplacian(a)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds and n >= 3:
            break
    ms = el / n * 1e3
    return {"value": 86400.0 / (ms * 1e-3 * STEPS_PER_YEAR), "unit": "simulated-years/day (1 member)", "cores": 1,
            "kind": "port", "ms_per_member_step": ms,
            "sample": "%d member-steps of the same hot path (91 spec2grid + physics + 73 grid2spec + spectral ops) "
                      "with oracle/liboracle.so in %.1f s" % (n, el)}


def main():
    args = parse()
    from pyspeedy_amd import ensemble as E
    world, rank, local = E.dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = E.init_process_group("nccl", device)  # RCCL; only used for the barrier and the max-over-ranks time

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    hp = HotPath(args.members, device, seed=1000 + rank)
    for _ in range(args.warmup):
        hp.step(record=False)
    hp.ev.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hp.step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = E.max_over_ranks(elapsed, dist, device)

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in hp.ev]))
    nfields = 57 * args.members
    achieved = (S_BYTES + G_BYTES) * nfields / (kern_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC measurement of this kernel (rocprofv3 cannot run inside bench.py):
    # bytes per field measured at 3648 fields/launch, FETCH_SIZE doubled as the gfx950 guide prescribes.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            tj = json.load(fh)
        traffic = tj["spec2grid_fused_hbm_bytes_per_field"] * nfields
        traffic_src = "profiles/r01_pmc_transforms_v2.csv (per-field bytes measured at %d fields/launch)" % tj["fields_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        total_members = args.members * world
        value = E.simulated_years_per_day(total_members, ms_step * 1e-3, STEPS_PER_YEAR)
        line = {
            "metric": "simulated-years/day (whole node), T30L8", "value": value, "unit": "simulated-years/day",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "cfg4-style ensemble, %d members/GPU, T30L8 96x48x8 fp64: HOT PATH ONLY per member-step = "
                            "91 spec2grid + fused column physics (shortwave every 3rd step) + 73 grid2spec + 16 vort2vel + "
                            "24 vel2vort + 1 gradient + 16 laplacian; grid-point dynamics algebra / implicit solver / "
                            "time filter of the full model step are not included" % args.members,
                "members_per_gpu": args.members, "members_total": total_members,
                "ms_per_member_step": ms_step / args.members, "parallelism": "ensemble members sharded per GPU, no collective",
            },
            "roofline": {
                "kernel": "spec2grid_kernel<Fused> (inverse Legendre + inverse FFT-96), %d fields/launch" % nfields,
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "algorithmic_bytes_per_field": S_BYTES + G_BYTES, "avg_launch_ms": kern_ms, "traffic": traffic,
                "traffic_source": traffic_src,
            },
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
