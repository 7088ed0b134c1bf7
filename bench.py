#!/usr/bin/env python3
"""Benchmark of pyspeedy_amd on MI355X: ensemble SPEEDY T30L8 (96x48x8), whole model step on the GPU.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--config cfg4|cfg5] [--members M]

One "step" = one call of the reference's do_single_step (speedy.f90:20-74) for EVERY ensemble member resident on each
GPU: daily forcing when due, shortwave every third step, the leapfrog step (spectral->grid transforms, grid-point
dynamics, fused column physics, 73 grid->spectral transforms, spectral tendencies, semi-implicit correction, horizontal
diffusion, Robert-Asselin-Williams filter), date advance and the land / sea-ice coupling.  Nothing crosses PCIe inside a
step; the state of all members stays in HBM.

Workloads (BASELINE.json):
  --config cfg4 (default)  fp64 everywhere.  --scaling weak (default): 64 members PER GPU; --scaling strong: 64 members in
                           total, block-sharded over the ranks (8 per GPU at N = 8, cfg 4 to the letter).
  --config cfg5            SPPT on + fp32 arithmetic in the column physics.  weak: 32 members per GPU; strong: 256 in total.
  --members M overrides the per-GPU (weak) or total (strong) member count.
Members never exchange data (speedy_driver.f90.j2:71-77): no data-path collective; the one collective is the start-up
broadcast of the boundary fields from rank 0 (RCCL, device buffers), outside the timed region.  Members start from the
reference's own initial state (example boundary fields, resting atmosphere, first_step), are perturbed as
examples/Ensemble_forecast.ipynb does (t_grid += N(0, 0.01 K), grid2spectral; seed = global member id) and spun up
`warmup` steps.

What the ONE JSON line of rank 0 holds:
  value / ms_per_step   the headline: `spd_model_step(m, K)` regions in the library's default launch plan (two member groups
                        on two HIP streams from 32 members per GPU up, `config.plan`), median region.
  roofline              the dominant transform kernel (spec2grid) timed by HIP events on its launch stream in further
                        regions of K steps issued in the SERIAL plan (one member group: with overlapping groups a
                        kernel's duration is not its own); `serial_plan_ms_per_step` beside it; `kernels[]` = every
                        kernel of the step from a bracketed one-day pass; `dominant` = the kernel with the largest share of
                        the step, the fused column kernel, priced the same way (its PMC traffic beside it).
  cpu_baseline          the reference Fortran itself (oracle/_ref; the C port when it did not travel) on one host core
                        and on all host cores, measured BEFORE any process touches the GPU -- by the launcher (N > 1 through
                        this script), or by rank 0 while the other ranks wait (N > 1 under torch.distributed.run).
  vs_baseline           value / cpu_baseline.all_cores.value: BASELINE.md holds no published number; north_star's target is
                        stated against the host-CPU reference ("core count stated"), so that is the ratio given (+ note).
  cfg4_strong (N > 1)   BASELINE cfg 4 to the letter beside the weak headline: 64 members in total, block-sharded.
  collective            what the collective layer saw, gathered through it (RCCL on the GPU box): ranks_seen, device_of_rank[],
                        the checksum of the boundary fields every rank received in the start-up broadcast and whether they agree.
                        `rccl_preflight`: before the ranks commit to RCCL each forms the same RCCL world in a bounded CHILD process
                        (all-reduce, broadcast, barrier on device buffers); rank 0 decides for all; when it fails the line is
                        produced over gloo and `backend_fallback` says why (the step has no collective: `value` is unaffected).
  one_process (N > 1)   the reference's own shape beside the process-per-GPU headline: ONE process drives all N GPUs -- containers
                        in blocks per device, boundary fields device to device, spd_parallel_step[_begin / _end] once per model
                        step over all containers.  Measured by a child process of rank 0 (`bench.py --one-process`) after the
                        ranks have released their models and while they sit idle on the host; a failure there is recorded in
                        the object and does not cost the line its headline.  `--one-process` alone runs only this and makes it
                        the line's value.
  drop_in_step (N = 1)  the reference-shaped host loop: spd_parallel_step once per model step (step + range check + codes
                        back), synchronous and in the overlapped begin / end form, over independent containers.
  every_step_stores (N = 1)  the step with every store of the reference restored (all 91 spectral->grid transforms, the
                        diagnostics-only physics outputs on every step).
  cfg3, cfg4_shard8, cfg5 (N = 1)  the other BASELINE configs on the same clock: 1 member fp64; 8 members fp64 (one GPU's share
                        of cfg 4 as worded); 32 members with SPPT + fp32 column physics (one GPU's share of cfg 5) -- ms_per_step,
                        launch plan, the step's own roofline fraction (algorithmic bytes of all its launches / time / 8 TB/s)
                        and per-kernel microseconds.
  cfg2_transforms (N = 1)  BASELINE cfg 2: spec2grid / grid2spec alone over batches of B = 1 ... 16 384 fields: ns per field and
                        fraction of the HBM peak.

  facade_run (N = 1)    the reference's own entry points on the same clock: SpeedyEns(64).run() and Speedy().run()
                        (pyspeedy/speedy.py:572-586, :398-405) over ten simulated days, without callbacks and with the default daily
                        XarrayExporter; ms per model step, Python loop, range check and file output included.
  roofline.stream_ceiling   what kernels of this library that only move bytes reach on this box (spd_stream_probe: copy, 2r:1w,
                        3r:2w, read, write; best shape and the column kernel's shape; TB/s) and `column_twin`, the column kernel's
                        launch with the arithmetic taken out; every kernel row carries frac_of_stream_ceiling beside its frac of 8 TB/s.
  roofline.frac_beyond_infinity_cache   the line's kernel at 256 members in the serial plan: `frac` at 64 members is HBM + Infinity
                        Cache.  roofline.traffic_stale: the committed PMC figure was taken with other device sources than this tree's.
  projected_8gpu_cfg4   (N = 1) BASELINE cfg 4 as worded on 8 GPUs, PROJECTED from this box: 8 members per GPU step in
                        cfg4_shard8.ms_per_step whatever the other 7 GPUs do (no collective in the step), so the node's
                        throughput is 64 members / that time; speedup and efficiency against this line's 64-members-on-one-GPU
                        headline.  A projection, labelled so: the first measured 8-GPU record is to be checked against it.
  config.step_contract_* / config.facade_* / config.projected_8gpu_cfg4_*   the scalars of drop_in_step, facade_run and
                        projected_8gpu_cfg4 once more, flat, where a reader that keeps only the contract's objects sees them.
                        `ms_per_step` is the multi-step device loop spd_model_step(m, K); the reference's step() contract -- one
                        call per model step with the range check and the codes back -- is config.step_contract_ms_per_step_*.

Wall-clock budget: `--budget SECONDS` (default 300) bounds the WHOLE run.  The headline, the roofline regions and the host
baseline always run; every secondary object (the legs above, cfg4_strong, one_process and its cfg4_strong) starts only while
the time that is left covers its allowance, and says `{"skipped": "budget"}` otherwise.  The child processes of the one_process
object get timeouts that sum to at most 150 s, and the ranks that wait for them wait exactly that long: a secondary leg that
hangs on the first contact with a second GPU cannot out-wait the driver's own timeout and cost the line its headline.

Launching: with N > 1 and no torchrun environment, this process only starts N rank processes (before touching the GPU),
relays rank 0's JSON line and fails if any rank fails.  Under `python -m torch.distributed.run ... bench.py --gpus N` each
process is one rank.  PYSPEEDY_AMD_BENCH_BACKEND=gloo rehearses the N-rank control flow when the ranks share one GPU.

Timing: W warm-up steps, then regions of EXACTLY K steps, each bracketed by barrier + synchronize on both sides and reduced
with MAX over ranks.  Regions are repeated until `--min-seconds` (default 5) of timed GPU work have accumulated (the run
stays inside the SST-anomaly months it allocates) and `ms_per_step` is the MEDIAN region (minimum and count beside it): a
20-step run is then several hundred 5-ms samples, and the GPU is busy for seconds, not milliseconds.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_PROCESS_START = time.time()  # the wall-clock budget counts from here (PYSPEEDY_AMD_BENCH_T0: the launcher's own start)

STEPS_PER_YEAR = 36 * 365  # model_control.f90:57-60, params.f90:32
S_BYTES, G_BYTES = 15872, 36864  # one spectral / one grid field
ANOM_MONTHS = 38                  # sst_anom(ix, il, 0:39), zero: December 1981 ... January 1985
MAX_STEPS = 3 * STEPS_PER_YEAR    # the run stays inside the months of (zero) SST anomalies the bench allocates
NG = 96 * 48
CFG2_SIZES = (1, 8, 64, 512, 4096, 16384)  # fields per launch of the cfg2_transforms leg (cfg 2)

# Algorithmic HBM bytes of each step kernel PER MEMBER, counted from the kernels' argument lists (DESIGN.md section 5 has
# the itemised lists).  Transforms: SURVEY 8d's contract figure S + G per field.
COLUMN_DOUBLES = {"column_sw": 254, "column": 243}  # doubles moved per column (dynamics 50 in / 55 out + physics)
DIAG_DOUBLES = 39           # of those, stores of diagnostics nothing on the device reads: only on the last step of a call
# cfg 5 with fp32 storage: values per column that travel as 4 bytes instead of 8 -- the 27 time-level-1 inputs, the
# persisted radiation state (read on steps without shortwave: tt_rsw 8 + rad_tau2 30 + rad_strat_corr 2; written on shortwave
# steps: 8 + 32 + 2), 35 of the 39 diagnostics-only stores; and the fields per member the spectral -> grid launch writes as fp32
COLUMN_FLOATS = {"column_sw": 27 + 42, "column": 27 + 40}
DIAG_FLOATS = 35
S2G_FLOAT_FIELDS = 27
COUPLER_DOUBLES = (55, 29)  # per column: with the climatologies interpolated (first coupling of a day) / re-used
ALGO_BYTES = {
    "geopotential": 17 * S_BYTES,
    "spectral_step": 215 * S_BYTES,
    "forcing": 19 * 8 * NG + 2 * (S_BYTES + G_BYTES),
    "dyn_grid": (50 + 55 + 18) * 8 * NG,
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=360)
    ap.add_argument("--warmup", type=int, default=36)
    ap.add_argument("--config", choices=("cfg4", "cfg5"), default="cfg4")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--members", type=int, default=None, help="members per GPU (weak) or in total (strong)")
    ap.add_argument("--regions", type=int, default=0, help="timed regions of `steps` steps (0 = automatic)")
    ap.add_argument("--min-seconds", type=float, default=5.0, help="timed regions are repeated until this much GPU time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="headline and roofline only: no drop_in_step / every_step_stores / cfg4_strong objects (profiles of "
                         "this command then hold launches of one size and one plan per region kind)")
    ap.add_argument("--serial-plan", action="store_true", help="headline in the serial plan too (one member group)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--leg-seconds", type=float, default=0.6,
                    help="GPU time of each of the cfg3 / cfg4_shard8 / cfg5 / cfg2_transforms legs of the one-GPU line")
    ap.add_argument("--cfg2-sizes", type=int, nargs="+", default=list(CFG2_SIZES), help="batch sizes of the cfg2_transforms leg")
    ap.add_argument("--one-process", action="store_true",
                    help="ONE process drives all --gpus devices through spd_parallel_step (the reference's own shape) and that is the "
                         "line's value; without it, an N > 1 line carries the same measurement as its `one_process` object")
    ap.add_argument("--budget", type=float, default=300.0,
                    help="wall-clock seconds for the whole run: secondary legs start only while the remaining time covers them")
    ap.add_argument("--cpu-worker", type=float, nargs=2, default=None, metavar=("T_START", "SECONDS"), help=argparse.SUPPRESS)
    ap.add_argument("--rccl-probe", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


class Budget:
    """The run's wall-clock budget.  `allows(seconds)`: is there room for a leg with that allowance?  Every refusal is kept so
    that the line can say what was left out and why."""

    def __init__(self, seconds):
        self.seconds = float(seconds)
        self.t0 = float(os.environ.get("PYSPEEDY_AMD_BENCH_T0", T_PROCESS_START))
        self.skipped = []

    def left(self):
        return self.seconds - (time.time() - self.t0)

    def allows(self, what, allowance):
        if self.left() >= allowance:
            return True
        self.skipped.append({"leg": what, "allowance_s": allowance, "left_s": round(self.left(), 1)})
        return False

    def record(self):
        return {"budget_s": self.seconds, "used_s": round(time.time() - self.t0, 1), "skipped": self.skipped}


# allowances of the secondary legs [s]: generous multiples of what they take on a healthy box (the whole default N = 1 line
# runs in 40-60 s); ONE_PROCESS_* are also the timeouts of the child processes, and sum to 150 s
LEG_ALLOWANCE = {"every_step_stores": 15, "drop_in_step": 25, "facade_run": 30, "cfg2_transforms": 10, "cfg3": 10, "cfg4_shard8": 10,
                 "cfg5": 15, "cfg4_strong": 30, "stream_ceiling": 8, "beyond_infinity_cache": 12}
ONE_PROCESS_TIMEOUT, ONE_PROCESS_STRONG_TIMEOUT = 100, 50


def load_bc():
    import numpy as np
    return np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ----------------------------------------------------------------------------------------------------------------------
# launcher: N rank processes, started before this process touches the GPU (it never does)
# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    port = str(free_port())
    procs = []
    baseline_file = None
    if not args.no_cpu_baseline:  # the host baseline of an N-rank line: measured here, before any rank exists
        import tempfile
        fd, baseline_file = tempfile.mkstemp(prefix="pyspeedy_bench_cpu_", suffix=".json")
        with os.fdopen(fd, "w") as fh:
            json.dump(cpu_baseline(args.cpu_seconds), fh)
    for r in range(args.gpus):
        env = dict(os.environ, WORLD_SIZE=str(args.gpus), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, PYSPEEDY_AMD_BENCH_T0=repr(T_PROCESS_START))
        if baseline_file:
            env["PYSPEEDY_AMD_BENCH_CPU_BASELINE"] = baseline_file
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE,
                                      text=True))
    failed = None
    give_up = T_PROCESS_START + args.budget + 90.0  # (the ranks bound themselves by the budget; this is the launcher's backstop)
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        if failed is None and time.time() > give_up:
            failed = next(r for r, p in enumerate(procs) if p.poll() is None)
        time.sleep(0.05)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:  # a dead rank leaves the others waiting in a collective: stop exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    outs = [p.stdout.read() for p in procs]
    if baseline_file:
        os.unlink(baseline_file)
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %s\n%s" % (failed, procs[failed].returncode, outs[failed]))
        raise SystemExit(1)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        raise SystemExit("bench.py: rank 0 printed %d result lines" % len(lines))
    print(lines[0], flush=True)


# ----------------------------------------------------------------------------------------------------------------------
# host baseline: the reference Fortran itself (oracle/_ref), one member on one core and one member per core on all cores
# ----------------------------------------------------------------------------------------------------------------------
def host_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def _reference_member():
    """(step function of one member, kind, description): the flang-compiled reference when it travelled, else the C port."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    bc = load_bc()
    try:
        import refmodel as R
        if not R.available():
            raise OSError("no reference library")
        m = R.RefModel(end=(1983, 1, 1, 0, 0))
        m.set_bc(bc)

        def step():
            assert m.step() == 0
        return step, "reference", "reference Fortran (amdflang -O2) do_single_step"
    except (OSError, AttributeError):
        import oracle as orc
        orc.build()
        m = orc.Model(n_months=12)  # the whole model restated in C (oracle/orc_model.c), bitwise the reference's trajectory
        m.set_bc(bc)
        assert m.init(1982, 1, 1) == 0

        def step():
            assert m.step() == 0
        return step, "port", "C port (oracle/liboracle.so) of do_single_step: the whole model, bit for bit the reference's results"


def _time_member(step, seconds, t_start=None):
    for _ in range(6):
        step()
    if t_start is not None:  # all workers of the all-core leg start their timed loop together
        while time.time() < t_start:
            time.sleep(0.001)
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(12):
            step()
        n += 12
        el = time.perf_counter() - t0
        if el >= seconds:
            return n, el


def cpu_worker(t_start, seconds):
    step, kind, _ = _reference_member()
    n, el = _time_member(step, seconds, t_start)
    print("CPUWORKER %s %d %.6f" % (kind, n, el), flush=True)


def cpu_baseline(seconds):
    """Runs BEFORE this process initialises the GPU (it starts child processes).  Leg 1: one member on one core.  Leg 2: one
    member per host core, all cores at once (members are independent: the reference's own OpenMP loop,
    speedy_driver.f90.j2:71-77, does exactly this), started together; throughput = sum over workers."""
    step, kind, what = _reference_member()
    n1, el1 = _time_member(step, seconds)
    ms1 = el1 / n1 * 1e3
    out = {"value": 86400.0 / (ms1 * 1e-3 * STEPS_PER_YEAR), "unit": "simulated-years/day", "cores": 1, "kind": kind,
           "ms_per_member_step": ms1,
           "sample": "%d model steps of one member with the %s in %.1f s on one host core" % (n1, what, el1)}
    cores = host_cores()
    t_start = time.time() + 6.0 + 0.02 * cores  # allowance for interpreter start-up, library load and model initialisation
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", repr(t_start), repr(seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
    rate, steps, longest, ok = 0.0, 0, 0.0, 0
    for p in procs:
        try:
            o = p.communicate(timeout=max(5.0, t_start + seconds + 45.0 - time.time()))[0]
        except subprocess.TimeoutExpired:  # (a worker that does not finish is left out of the sum, not waited for)
            p.kill()
            o = ""
        for ln in o.splitlines():
            if ln.startswith("CPUWORKER"):
                _, _, n, el = ln.split()
                rate += int(n) / float(el)
                steps += int(n)
                longest = max(longest, float(el))
                ok += 1
    if ok:
        out["all_cores"] = {
            "value": rate * 86400.0 / STEPS_PER_YEAR, "unit": "simulated-years/day", "cores": ok, "kind": kind,
            "member_steps_per_second": rate, "ms_per_member_step_per_core": ok / rate * 1e3,
            "sample": "%d member-steps: one member per core on %d cores at once (%d workers started), %.1f s each" % (
                steps, ok, cores, longest)}
    return out


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------
def workload(args, world, rank):
    """(members on this rank, global id of its first member, members in total)"""
    from pyspeedy_amd import ensemble as E
    per_gpu_default, total_default = (64, 64) if args.config == "cfg4" else (32, 256)
    if args.scaling == "weak":
        m = args.members if args.members is not None else per_gpu_default
        return m, rank * m, m * world
    total = args.members if args.members is not None else total_default
    first, count = E.shard_members(total, world, rank)
    if count == 0:
        raise SystemExit("bench.py: rank %d would own no member (%d members on %d ranks)" % (rank, total, world))
    return count, first, total


def build_ensemble(args, M, first_id, device, dist, rank, coll_device):
    import numpy as np
    import torch
    import pyspeedy_amd
    from pyspeedy_amd import ensemble as E
    from pyspeedy_amd.model import EnsembleModel
    sp = pyspeedy_amd.ModSpectral(device.index)
    model = EnsembleModel(sp, M)
    model.init_sst_anom(ANOM_MONTHS)  # zero anomalies over the whole run (speedy.py:338-372)
    # rank 0 reads the boundary file; one RCCL broadcast (~3.4 MB over xGMI) hands it to the other GPUs
    bc = E.broadcast_boundary_conditions(dict(load_bc()) if rank == 0 else None, dist, coll_device)
    model.bc_checksum, model.bc_bytes = boundary_checksum(bc)  # (of what THIS rank received)
    model.set_bc(bc, start_date=(1982, 1, 1, 0, 0))
    # cfg 4 perturbation (examples/Ensemble_forecast.ipynb): t_grid += N(0, 0.01), then grid2spectral;
    # the generator is seeded with the GLOBAL member id, so the ensemble does not depend on how it is sharded
    model.spectral2grid()
    t_grid = model.device_view("t_grid")  # [M, lev, lat, lon] in HBM
    noise = np.stack([np.random.default_rng(first_id + i).normal(0.0, 0.01, (96, 48, 8)).transpose(2, 1, 0) for i in range(M)])
    t_grid += torch.from_numpy(np.ascontiguousarray(noise)).to(device)
    model.grid2spectral()
    if args.config == "cfg5":
        model.set_sppt(True, seed=2024, first_member_id=first_id)
        model.set_physics_precision(True)
    return sp, model


def boundary_checksum(bc):
    """(60-bit checksum, bytes) of a set of boundary fields: sha256 over names, shapes and fp64 values in name order."""
    import hashlib
    import numpy as np
    h, n = hashlib.sha256(), 0
    for k in sorted(bc):
        a = np.ascontiguousarray(np.asarray(bc[k], dtype=np.float64))
        h.update(k.encode())
        h.update(repr(a.shape).encode())
        h.update(a.tobytes())
        n += a.nbytes
    return int(h.hexdigest()[:15], 16), n


def collective_record(dist, backend, device, coll_device, rank, checksum, nbytes):
    """What the collective layer actually saw, gathered THROUGH it: every rank contributes (rank, HIP device index, checksum of
    the boundary fields it received in the start-up broadcast) as an int64 tensor to one all_gather -- RCCL on device buffers on
    the GPU box -- and its GPU's identity to one all_gather_object.  A line that holds this object proves that N ranks met, which
    device each one computed on, and that the broadcast delivered the same bytes everywhere."""
    import torch
    index = device.index if device.type == "cuda" else -1  # (a CPU "device": the gloo rehearsal of the CPU test tier)
    ident = {"rank": rank, "pid": os.getpid(), "device": index}
    if device.type == "cuda":
        props = torch.cuda.get_device_properties(device)
        ident.update(name=props.name, uuid=str(getattr(props, "uuid", "")), pci_bus_id=getattr(props, "pci_bus_id", None))
    mine = torch.tensor([rank, index, checksum], dtype=torch.int64, device=coll_device)
    if dist is None:
        rows, idents, world = [mine.tolist()], [ident], 1
    else:
        world = dist.get_world_size()
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        rows = [p.tolist() for p in parts]
        idents = [None] * world
        try:
            dist.all_gather_object(idents, ident)
        except Exception as exc:  # (the tensor path above is the proof; the identities are an extra)
            idents = [{"error": repr(exc)}]
    gpus = {(i.get("uuid") or "", i.get("pci_bus_id"), i.get("device")) for i in idents if i and "rank" in i}
    return {"backend": backend if dist is not None else "none (single process)", "world_size": world,
            "ranks_seen": len({r[0] for r in rows}), "device_of_rank": [r[1] for r in sorted(rows)],
            "boundary_bytes": nbytes, "boundary_checksum_of_rank": ["%015x" % r[2] for r in sorted(rows)],
            "boundary_checksum_equal": len({r[2] for r in rows}) == 1, "distinct_gpus": len(gpus), "gpu_of_rank": idents}


def kernel_table(model, M, inv_per_member, sppt):
    """roofline.kernels[]: every kernel of the step with its algorithmic bytes (argument lists), HIP-event time, fraction of
    the 8 TB/s HBM peak.  Measured in a separate one-day pass with a bracket around every launch."""
    from pyspeedy_amd.model import KERNEL_NAMES  # noqa: F401
    model.profile(2)
    model.run(36)  # one call, one simulated day: 12 shortwave steps, one daily forcing, one midnight coupling
    prof = model.profile_read_kernels()
    model.profile(0)
    extra = 8 if sppt else 0  # the column kernel also reads the SPPT pattern
    # average over the 36 launches of the pass: the diagnostics-only stores happen on the last step only (unless the model
    # is told to store them every step), the coupler interpolates its climatologies on one step of the day
    skipped = 0.0 if model.config()["diag_every_step"] else 35.0 / 36.0  # share of the launches that do not store the diagnostics
    diag = DIAG_DOUBLES * skipped
    store32 = model.config()["physics_storage32"]
    coupler = (COUPLER_DOUBLES[0] + 35 * COUPLER_DOUBLES[1]) / 36.0
    rows = []
    for name, (mean_ms, min_ms, n, units) in prof.items():
        if name == "spec2grid":
            members = units // (inv_per_member + extra)
            algo = (S_BYTES + G_BYTES) * units - (G_BYTES // 2) * S2G_FLOAT_FIELDS * members * (1 if store32 else 0)
        elif name == "grid2spec":
            algo = (S_BYTES + G_BYTES) * units
        elif name in COLUMN_DOUBLES:
            floats = (COLUMN_FLOATS[name] + DIAG_FLOATS * (1.0 - skipped)) if store32 else 0.0
            algo = ((COLUMN_DOUBLES[name] + extra - diag) * 8 - floats * 4) * NG * units
        elif name in ("physics_sw", "physics"):
            algo = (COLUMN_DOUBLES["column_sw" if name == "physics_sw" else "column"] - 105 + 18 + extra) * 8 * NG * units
        elif name == "sppt":  # the AR(1) update of the spectral pattern; its 8 transforms per member ride in spec2grid
            algo = 2 * S_BYTES * units
        elif name == "coupler":
            algo = coupler * 8 * NG * units
        elif name == "spectral_step":  # (+ the coupling and / or the next geopotential when they ride in this launch)
            cfg = model.config()
            algo = (ALGO_BYTES[name] + (coupler * 8 * NG if cfg["coupler_in_spectral"] else 0) +
                    (8 * S_BYTES if cfg["fold_geo"] else 0)) * units
        elif name == "geopotential":  # (+ the SPPT pattern update of 8 spectral fields per member, which rides in this launch)
            algo = (ALGO_BYTES[name] + (2 * 8 * S_BYTES if sppt else 0)) * units
        else:
            algo = ALGO_BYTES[name] * units
        gbs = algo / (mean_ms * 1e-3) / 1e9
        rows.append({"kernel": name, "launches_timed": n, "avg_launch_us": mean_ms * 1e3, "min_launch_us": min_ms * 1e3,
                     "algorithmic_bytes_per_launch": int(round(algo)), "achieved": gbs, "frac": gbs / 8000.0})
    return rows


def timed_regions(model, args, barrier, dist, coll_device, steps_left, min_seconds, max_regions=1000):
    """Regions of EXACTLY args.steps steps between barrier + synchronize, MAX over ranks; repeated until `min_seconds` of
    timed work (or --regions, or the months of SST anomaly run out).  Returns the list of region seconds."""
    from pyspeedy_amd import ensemble as E
    regions, out, r = max(args.regions, 1), [], 0
    while r < regions:
        barrier()
        t0 = time.perf_counter()
        model.run(args.steps)
        barrier()
        out.append(E.max_over_ranks(time.perf_counter() - t0, dist, coll_device))
        if r == 0 and args.regions == 0:  # identical on every rank: it is the all-reduced time
            regions = min(max_regions, int(math.ceil(min_seconds / max(out[0], 1e-6))))
        regions = max(1, min(regions, steps_left // max(args.steps, 1)))
        r += 1
    return out


def median(values):
    v = sorted(values)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def fidelity_leg(args, M, first_id, device, dist, rank, coll_device, barrier):
    """ms/step of the same ensemble with every store of the reference restored: all 91 spectral->grid transforms
    (PYSPEEDY_AMD_PRUNE_DEAD=0) and the diagnostics-only physics outputs on every step."""
    os.environ["PYSPEEDY_AMD_PRUNE_DEAD"] = "0"  # read when the model is created
    try:
        sp, model = build_ensemble(args, M, first_id, device, dist, rank, coll_device)
    finally:
        del os.environ["PYSPEEDY_AMD_PRUNE_DEAD"]
    model.set_option("diag_every_step", 1)
    model.run(args.warmup)
    secs = timed_regions(model, args, barrier, dist, coll_device, MAX_STEPS - args.warmup - 36, 0.5, 200)
    ok = (model.check(2) == 0).all()
    cfg = model.config()
    model.close()
    sp.close()
    if not ok:
        raise SystemExit("bench.py: members left the accepted range in the every-step-stores leg")
    return {"ms_per_step": median(secs) / args.steps * 1e3, "regions": len(secs), "spec2grid_per_member": cfg["inv_per_member"],
            "plan": plan_name(cfg, M),
            "note": "the reference's every store: 91 spectral->grid transforms per member (14 of them feed nothing) and the 39 "
                    "diagnostics-only physics outputs per column stored on every step of the call"}


LEG_REGION_STEPS = 360  # the legs time regions of ten simulated days whatever --steps the headline was given


def config_leg(args, config, members, what, device, dist, rank, coll_device, barrier):
    """One more BASELINE config on the same clock as the headline: the ensemble of `config` with `members` members on this GPU,
    built, perturbed and spun up exactly like the headline's, timed in regions of LEG_REGION_STEPS steps between barrier +
    synchronize (default plan of the library for that size) until --leg-seconds of GPU work, median region; then one simulated
    day in the serial plan with events attached to every dispatch for the per-kernel durations.  `step_roofline` prices the
    WHOLE step: the algorithmic bytes of all its launches / ms_per_step against the 8 TB/s HBM peak."""
    from pyspeedy_amd import ensemble as E
    leg = argparse.Namespace(**dict(vars(args), config=config, scaling="strong", members=members, steps=LEG_REGION_STEPS, regions=0))
    sp, model = build_ensemble(leg, members, 0, device, dist, rank, coll_device)
    cfg = model.config()
    model.run(leg.warmup)
    secs = timed_regions(model, leg, barrier, dist, coll_device, MAX_STEPS - leg.warmup - 72, args.leg_seconds, 500)
    cfg = model.config()
    model.profile(2)  # a first bracketed day that is not read: it creates the events and takes the one-off costs of the
    model.run(36)     # profiling path (the headline's table comes behind its serial-plan regions, which do the same for it)
    model.sync()
    rows = kernel_table(model, members, cfg["inv_per_member"], config == "cfg5")
    ok = (model.check(2) == 0).all()
    model.close()
    sp.close()
    if not ok:
        raise SystemExit("bench.py: members left the accepted range in the %s leg" % what)
    ms = median(secs) / leg.steps * 1e3
    step_bytes = sum(r["algorithmic_bytes_per_launch"] * r["launches_timed"] for r in rows) / 36.0
    gbs = step_bytes / (ms * 1e-3) / 1e9
    out = {
        "workload": what, "members": members, "ms_per_step": ms, "us_per_member_step": ms * 1e3 / members,
        "ms_per_step_min": min(secs) / leg.steps * 1e3, "regions": len(secs), "steps_per_region": leg.steps,
        "timed_seconds": sum(secs), "plan": plan_name(cfg, members),
        "value": E.simulated_years_per_day(members, ms * 1e-3, STEPS_PER_YEAR), "unit": "simulated-years/day",
        "step_roofline": {"bound": "hbm", "algorithmic_bytes_per_step": int(round(step_bytes)), "achieved": gbs, "peak": 8000.0,
                          "unit": "GB/s", "frac": gbs / 8000.0},
        "kernel_us": {r["kernel"]: round(r["avg_launch_us"], 3) for r in rows},
        "kernel_us_min": {r["kernel"]: round(r["min_launch_us"], 3) for r in rows},
        "kernel_frac": {r["kernel"]: round(r["frac"], 4) for r in rows},
        "kernel_us_sum_per_step": sum(r["avg_launch_us"] * r["launches_timed"] for r in rows) / 36.0,
    }
    return out


def transforms_leg(args, device):
    """BASELINE cfg 2: the two fused transform kernels on their own through the operator-level C ABI (spd_spec2grid /
    spd_grid2spec, include/pyspeedy_amd.h) over contiguous batches of B fields.  Inputs as SURVEY 8d words them: spectra =
    triangular-truncated complex normal(0, 1) / (1 + l), Im(m = 0) = 0, numpy default_rng(1234); grids = spec2grid of those
    spectra (band-limited).  Per (kernel, B): back-to-back launches on one stream between two HIP events, repeated until
    --leg-seconds / 12 of GPU time; ns per field and the fraction of 8 TB/s at 52 736 algorithmic bytes per field."""
    import ctypes as C
    import numpy as np
    import torch
    import pyspeedy_amd
    sp = pyspeedy_amd.ModSpectral(device.index)
    L, h = sp._lib, sp.handle
    bmax = max(args.cfg2_sizes)
    rng = np.random.default_rng(1234)
    nn, mm = np.meshgrid(np.arange(32), np.arange(31), indexing="ij")  # [n][m]; total wavenumber l = m + n
    spec = (rng.standard_normal((bmax, 32, 31)) + 1j * rng.standard_normal((bmax, 32, 31))) / (1.0 + nn + mm)
    spec[:, (nn + mm) > 30] = 0
    spec[:, :, 0] = spec[:, :, 0].real
    spec_d = torch.from_numpy(spec).to(device)
    grid_d = torch.empty((bmax, 48, 96), dtype=torch.float64, device=device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    if L.spd_spec2grid(h, p(spec_d), p(grid_d), 1, bmax, st) != 0:
        raise SystemExit("bench.py: cfg2 leg: " + L.spd_last_error().decode())
    out_spec, out_grid = torch.empty_like(spec_d), torch.empty_like(grid_d)
    # the Legendre stage on its own (north_star's literal target: >= 40 % of the HBM roofline on the Legendre transform, 39 680
    # algorithmic bytes per field = spectral field + Fourier plane): the same kernels with the FFT stage disabled, at the two
    # largest batch sizes; the Fourier planes are those of the same fields
    four_d = torch.empty((bmax, 48, 62), dtype=torch.float64, device=device)
    if L.spd_legendre_inv(h, p(spec_d), p(four_d), bmax, st) != 0:
        raise SystemExit("bench.py: cfg2 leg: " + L.spd_last_error().decode())
    out_four = torch.empty_like(four_d)
    torch.cuda.synchronize()
    legendre_sizes = sorted(args.cfg2_sizes)[-2:]
    share = max(args.leg_seconds / (2.0 * len(args.cfg2_sizes) + 2.0 * len(legendre_sizes)), 1e-3)
    rows, total = [], 0.0
    F_BYTES = 48 * 62 * 8
    for B in args.cfg2_sizes:
        calls = {"spec2grid": (lambda: L.spd_spec2grid(h, p(spec_d), p(out_grid), 1, B, st), S_BYTES + G_BYTES),
                 "grid2spec": (lambda: L.spd_grid2spec(h, p(grid_d), p(out_spec), B, st), S_BYTES + G_BYTES)}
        if B in legendre_sizes:
            calls["legendre_inv"] = (lambda: L.spd_legendre_inv(h, p(spec_d), p(out_four), B, st), S_BYTES + F_BYTES)
            calls["legendre"] = (lambda: L.spd_legendre(h, p(four_d), p(out_spec), B, st), S_BYTES + F_BYTES)
        for name, (fn, field_bytes) in calls.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            n, elapsed, a, b = 8, 0.0, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            while True:
                a.record()
                for _ in range(n):
                    if fn() != 0:
                        raise SystemExit("bench.py: cfg2 leg: " + L.spd_last_error().decode())
                b.record()
                b.synchronize()
                elapsed = a.elapsed_time(b) * 1e-3
                total += elapsed
                if elapsed >= share or n >= (1 << 20):
                    break
                n = min(1 << 20, max(2 * n, int(1.2 * n * share / max(elapsed, 1e-6))))
            per_launch = elapsed / n
            gbs = field_bytes * B / per_launch / 1e9
            rows.append({"kernel": name, "fields": B, "launches_timed": n, "us_per_launch": per_launch * 1e6,
                         "ns_per_field": per_launch / B * 1e9, "algorithmic_bytes_per_field": field_bytes, "achieved": gbs,
                         "frac": gbs / 8000.0})
    # BASELINE cfg 2 as worded keeps the physics -- and with it the state -- on the HOST: every batch of transforms then crosses
    # PCIe both ways.  The same two kernels with their in- and outputs in pinned host memory (copy in, transform, copy out, one
    # stream): what the operator-level boundary costs a host-resident model.  Never `value`.
    Bh = min(4096, bmax)
    spec_h, grid_h = spec_d[:Bh].cpu().pin_memory(), grid_d[:Bh].cpu().pin_memory()
    back_grid, back_spec = torch.empty_like(grid_h).pin_memory(), torch.empty_like(spec_h).pin_memory()
    pcie = {}
    for name in ("spec2grid", "grid2spec"):
        def once():
            if name == "spec2grid":
                spec_d[:Bh].copy_(spec_h, non_blocking=True)
                rc = L.spd_spec2grid(h, p(spec_d), p(out_grid), 1, Bh, st)
                back_grid.copy_(out_grid[:Bh], non_blocking=True)
            else:
                grid_d[:Bh].copy_(grid_h, non_blocking=True)
                rc = L.spd_grid2spec(h, p(grid_d), p(out_spec), Bh, st)
                back_spec.copy_(out_spec[:Bh], non_blocking=True)
            if rc != 0:
                raise SystemExit("bench.py: cfg2 leg: " + L.spd_last_error().decode())
        once()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            once()
        b.record()
        b.synchronize()
        per = a.elapsed_time(b) * 1e-3 / 5
        total += 5 * per
        pcie[name] = {"fields": Bh, "ns_per_field": per / Bh * 1e9, "achieved": (S_BYTES + G_BYTES) * Bh / per / 1e9, "unit": "GB/s"}
    sp.close()
    return {"pcie_inclusive": dict(pcie, note="in- and outputs in pinned HOST memory (copy in, transform, copy out on one stream): "
                                               "BASELINE cfg 2 as worded, physics and state on the host; bound by PCIe, not by the GPU"),
            "workload": "BASELINE cfg 2: the fused transform kernels alone, contiguous batches of B fields (spd_spec2grid / "
                        "spd_grid2spec), SURVEY 8d inputs (seed 1234, band-limited grids)",
            "algorithmic_bytes_per_field": S_BYTES + G_BYTES, "algorithmic_bytes_per_field_legendre_stage": S_BYTES + F_BYTES,
            "peak": 8000.0, "unit": "GB/s", "timed_seconds": total, "rows": rows,
            "note": "back-to-back launches on one stream between two HIP events; at B = 1 and 8 a launch is one dependent chain of "
                    "one workgroup (load, Legendre, FFT, store) and the figure is that latency, not a bandwidth; the in- and outputs "
                    "of a batch of up to 4096 fields (216 MB fused, 162 MB Legendre stage) largely stay in the 256 MB Infinity "
                    "Cache between launches, so the 16 384-field rows (864 / 650 MB) are the HBM figures; legendre_inv / legendre = "
                    "the Legendre stage on its own (39 680 algorithmic bytes per field: north_star's >= 40 % target)"}


def _time_container_loop(ens, steps):
    """Per-step wall times of spd_parallel_step over all containers of `ens`, synchronous and begin / end (medians and means).
    The calls are the C entry points themselves, on argument arrays built once: what a Fortran / C host of the reference's loop
    pays per step (the Python veneer would add its own list -> array conversions to every call)."""
    import ctypes as C
    import torch
    L = pyspeedy_amd_lib()
    n = len(ens.members)
    states = (C.c_int64 * n)(*[m._state_cnt for m in ens.members])
    controls = (C.c_int64 * n)(*[m._control_cnt for m in ens.members])
    codes = (C.c_int32 * n)()

    def ok(rc):
        if rc != 0:
            raise SystemExit("bench.py: parallel_step loop: " + L.spd_last_error().decode())

    def sync_all():
        for d in range(torch.cuda.device_count()):
            torch.cuda.synchronize(d)

    for _ in range(12):
        ok(L.spd_parallel_step(states, controls, codes, n))
    assert not any(codes)
    sync_all()
    sync = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ok(L.spd_parallel_step(states, controls, codes, n))
        sync.append(time.perf_counter() - t0)
    assert not any(codes)
    ovl, worst = [], 0
    token, nxt = C.c_int64(), C.c_int64()
    ok(L.spd_parallel_step_begin(states, controls, n, C.byref(token)))
    for _ in range(steps):
        t0 = time.perf_counter()
        ok(L.spd_parallel_step_begin(states, controls, n, C.byref(nxt)))
        ok(L.spd_parallel_step_end(token, codes))
        token.value = nxt.value
        ovl.append(time.perf_counter() - t0)
        worst |= int(any(codes))
    ok(L.spd_parallel_step_end(token, codes))
    assert not worst and not any(codes)
    return {"steps_timed": steps, "sync_ms_per_step": median(sync) * 1e3, "begin_end_ms_per_step": median(ovl) * 1e3,
            "sync_ms_per_step_mean": sum(sync) / len(sync) * 1e3, "begin_end_ms_per_step_mean": sum(ovl) / len(ovl) * 1e3}


def drop_in_leg(M, steps):
    """The reference-shaped host loop over M independent containers through the outer C boundary (include/pyspeedy_amd_driver.h):
    spd_parallel_step once per model step -- step, range check, error codes back -- synchronously and in the overlapped
    begin / end form.  Per-step wall times, medians (the runtime stalls once per process for ~40 ms shortly after the first
    launches; a median keeps that out)."""
    from datetime import datetime
    from pyspeedy_amd import speedy_driver as drv
    from pyspeedy_amd.speedy import SpeedyEns
    ens = SpeedyEns(M, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 3, 1))
    for member in ens:
        member.set_bc()
    out = {"containers": M}
    out.update(_time_container_loop(ens, steps))
    out["device_models"] = len({drv.device_model(m._state_cnt)[0]._m.value for m in ens.members})  # (as this loop's habit left them)
    del ens
    out["note"] = ("spd_parallel_step(state_cnts, control_cnts, error_codes, n) once per model step over independent containers "
                   "(one device model up to 31 containers, two from 32 up, both enqueued before either is waited for): every "
                   "step stores all diagnostics and runs the range check; sync = the call returns the codes (the check is a launch "
                   "behind the step), begin_end = the check of step k is collected after step k + 1 has been enqueued (and rides in "
                   "that step's first launch); medians of per-step wall times")
    return out


def small_drop_in_legs(steps):
    """The same host loop for the small cases of BASELINE.json: one container (cfg 3 through the reference's `step()` contract --
    what Speedy.run() pays per step) and 8 (cfg 4's share of one GPU)."""
    out = {}
    for containers in (1, 8):
        leg = drop_in_leg(containers, steps)
        out["containers_%d" % containers] = {k: leg[k] for k in ("containers", "device_models", "steps_timed", "sync_ms_per_step",
                                                                 "begin_end_ms_per_step")}
    return out


def facade_leg():
    """The reference's own entry points, as a user of pySPEEDY calls them (pyspeedy/speedy.py:572-586 and :398-405):
    `SpeedyEns(64)` with `for member in ens: member.set_bc()` and `ens.run(callbacks)`, and `Speedy().set_bc(); run(callbacks)` --
    the Python time loop, one parallel_step per model step with its range check, and (second figure) the default daily
    XarrayExporter: u, v, t, q, phi, ps of every member, device -> host -> NetCDF-3 file, once per simulated day.  Wall time of
    run() / model steps, the better of two runs.  A one-day run of the same shape goes first, untimed (one-off costs)."""
    import tempfile
    from datetime import datetime, timedelta
    import torch
    from pyspeedy_amd.callbacks import ModelCheckpoint, XarrayExporter
    from pyspeedy_amd.speedy import Speedy, SpeedyEns
    start = datetime(1982, 1, 1)

    def make(members, days):
        end = start + timedelta(days=days)
        if members == 1:
            model = Speedy(start_date=start, end_date=end)
            model.set_bc()
            return model
        ens = SpeedyEns(members, start_date=start, end_date=end)
        if members > 64:
            ens.set_bc()  # (this library's extension: the file is read once and handed on device to device)
            return ens
        for member in ens:  # (the reference's way)
            member.set_bc()
        return ens

    def timed(members, days, export):
        model = make(members, days)
        with tempfile.TemporaryDirectory(prefix="pyspeedy_bench_") as tmp:
            callbacks = ([ModelCheckpoint()] if export == "checkpoint" else [XarrayExporter(output_dir=tmp)]) if export else []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.run(callbacks=callbacks)
            torch.cuda.synchronize()
            seconds = time.perf_counter() - t0
            files = [os.path.join(tmp, f) for f in os.listdir(tmp)]
            written = sum(os.path.getsize(f) for f in files)
        assert model.get_current_step() == 36 * days
        del model
        return seconds / (36 * days) * 1e3, len(files), written

    out = {}
    for key, members, days_plain, days_export in (("ens64", 64, 10, 10), ("single", 1, 10, 10), ("ens256", 256, 4, 0)):
        if key == "ens256":  # (large ensembles through the facade: the 64-member rate per member-step, no file output)
            timed(members, 1, False)
            plain = min(timed(members, days_plain, False)[0] for _ in range(2))
            out[key] = {"members": members, "run_ms_per_step": plain, "run_steps": 36 * days_plain, "us_per_member_step": plain / members * 1e3}
            continue
        timed(members, 1, False)
        # the better of two runs each: the ROCm runtime stalls once per process for ~40 ms shortly after its first launches, and
        # one such stall is a tenth of a millisecond per step of a 360-step run
        plain = min(timed(members, days_plain, False)[0] for _ in range(2))
        exported, files, written = min(timed(members, days_export, True) for _ in range(2))
        out[key] = {"members": members, "run_ms_per_step": plain, "run_steps": 36 * days_plain,
                    "run_daily_export_ms_per_step": exported, "run_daily_export_steps": 36 * days_export, "files_written": files,
                    "megabytes_written": written / 1e6}
        if key == "ens64":  # (the reference's in-memory hook: the snapshots wait on the GPU until the series is read)
            out[key]["run_daily_checkpoint_ms_per_step"] = min(timed(members, days_export, "checkpoint")[0] for _ in range(2))
    out["note"] = ("SpeedyEns(64).run() / Speedy().run() of the facade (pyspeedy_amd/speedy.py = the reference's classes over the C "
                   "boundary): wall time of run() per model step; the steps between two due callbacks are ONE call "
                   "(spd_parallel_steps_begin / _end) with the range check of every step recorded on the device; "
                   "run_daily_export = with callbacks=[XarrayExporter()] (defaults: every 36 steps, u v t q phi ps, NetCDF-3 files, "
                   "written behind the time loop); run_daily_checkpoint = with callbacks=[ModelCheckpoint()] (the same variables kept "
                   "as a time series in memory); ens256: the same for 256 members (rounds of 64 inside every call)")
    return out


def step_contract_keys(legs, headline_ms, members_total):
    """The scalars of drop_in_step / facade_run / projected_8gpu_cfg4 once more, flat, for `config` (readers that keep the
    contract's objects and scalar members only)."""
    flat = {}
    d = legs.get("drop_in_step") or {}
    if "sync_ms_per_step" in d:
        n = d["containers"]
        flat["step_contract_ms_per_step_sync_%d" % n] = d["sync_ms_per_step"]
        flat["step_contract_ms_per_step_begin_end_%d" % n] = d["begin_end_ms_per_step"]
        for sub in ("containers_8", "containers_1"):
            if sub in d and d[sub]["containers"] != n:
                flat["step_contract_ms_per_step_sync_%d" % d[sub]["containers"]] = d[sub]["sync_ms_per_step"]
                flat["step_contract_ms_per_step_begin_end_%d" % d[sub]["containers"]] = d[sub]["begin_end_ms_per_step"]
        flat["step_contract_note"] = ("ms_per_step is spd_model_step(m, K), K steps per call; the reference's step() contract (one call "
                                      "per step, range check, codes back) is step_contract_ms_per_step_* over that many containers")
    f = legs.get("facade_run") or {}
    for key in ("ens64", "single", "ens256"):
        if key in f:
            flat["facade_%s_run_ms_per_step" % key] = f[key]["run_ms_per_step"]
            if "run_daily_export_ms_per_step" in f[key]:
                flat["facade_%s_run_daily_export_ms_per_step" % key] = f[key]["run_daily_export_ms_per_step"]
            if "run_daily_checkpoint_ms_per_step" in f[key]:
                flat["facade_%s_run_daily_checkpoint_ms_per_step" % key] = f[key]["run_daily_checkpoint_ms_per_step"]
    pr = legs.get("projected_8gpu_cfg4") or {}
    for key in ("value", "speedup_over_1gpu", "efficiency"):
        if key in pr:
            flat["projected_8gpu_cfg4_%s" % key] = pr[key]
    return flat


def scaling_keys(collective, legs):
    """What tools/check_scale.py reads of an N-rank line, once more as flat scalars in `config` (a reader that keeps the contract's
    objects only -- the driver's SCALE records -- still holds them): what the collective layer saw, BASELINE cfg 4 as worded
    (cfg4_strong) and the one-process shape."""
    flat = {}
    c = collective or {}
    for key in ("backend", "ranks_seen", "distinct_gpus", "boundary_checksum_equal"):
        if key in c:
            flat["collective_" + key] = c[key]
    if "rccl_preflight" in c:
        flat["collective_rccl_preflight_ok"] = bool(c["rccl_preflight"].get("ok"))
    flat["collective_backend_fallback"] = c.get("backend_fallback")
    st = legs.get("cfg4_strong") or {}
    for key in ("value", "ms_per_step", "members_per_gpu"):
        if key in st:
            flat["cfg4_strong_" + key] = st[key]
    op = legs.get("one_process") or {}
    for key in ("ms_per_step", "value", "devices_used", "containers"):
        if key in op:
            flat["one_process_" + key] = op[key]
    if "boundary_broadcast" in op:
        flat["one_process_boundary_broadcast_note"] = op["boundary_broadcast"].get("note")
    if "error" in op:
        flat["one_process_error"] = str(op["error"])[:200]
    if "value" in (op.get("cfg4_strong") or {}):
        flat["one_process_cfg4_strong_value"] = op["cfg4_strong"]["value"]
    return flat


def projection_8gpu(shard8, headline_value, headline_members, all_cores):
    """BASELINE cfg 4 as worded (64 members, 8 per GPU on 8 GPUs), PROJECTED from one GPU: members never exchange data in a step
    (no collective: speedy_driver.f90.j2:71-77), so each GPU steps its 8 members in cfg4_shard8.ms_per_step whatever the other
    seven do, and the node's throughput is 64 members / that time.  Against this line's 64 members on ONE GPU."""
    from pyspeedy_amd import ensemble as E
    ms = shard8["ms_per_step"]
    value = E.simulated_years_per_day(64, ms * 1e-3, STEPS_PER_YEAR)
    one_gpu_64 = headline_value * 64.0 / headline_members  # (the headline is 64 members on one GPU unless --members said otherwise)
    out = {"projection": True, "n_gpus": 8, "members_total": 64, "members_per_gpu": 8, "ms_per_step": ms, "value": value,
           "unit": "simulated-years/day", "speedup_over_1gpu": value / one_gpu_64, "efficiency": value / one_gpu_64 / 8.0,
           "note": "PROJECTED, not measured: 64 members / cfg4_shard8.ms_per_step (8 members on this GPU; the step has no collective, "
                   "the other GPUs do not enter), against 64 members on one GPU at this line's headline rate.  Strong scaling of 64 "
                   "members tops out here because an 8-member launch fills a fraction of a GPU (DESIGN: small shards)"}
    if all_cores:
        out["vs_cpu_all_cores"] = value / all_cores["value"]
    return out


def agreed(dist, value):
    """rank 0's `value` on every rank (decisions about what to run must not differ between ranks that meet in collectives)"""
    if dist is None:
        return value
    box = [value]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def one_process_leg(n_devices, members_total, steps, what):
    """The reference's own shape (speedy_driver.f90.j2:58-79, SpeedyEns.run): ONE process owns every container and hands them all
    to parallel_step.  The containers are placed in blocks on the first `n_devices` GPUs this process can see
    (spd_modelstate_init_ensemble_on: member e of n on device e k / n), the boundary fields are set on container 0 only and
    handed to the others device to device (spd_broadcast_boundary: one crossing per GPU), every member is initialised and
    perturbed like the headline's (t_grid += N(0, 0.01 K), seed = member id), and spd_parallel_step[_begin / _end] is called
    once per model step over ALL containers: every GPU's step and range check are enqueued before the host waits for any."""
    import numpy as np
    import torch
    from datetime import datetime
    from pyspeedy_amd import speedy_driver as drv
    from pyspeedy_amd import speedy as S
    before = torch.cuda.current_device()
    used = max(1, min(n_devices, drv.device_count()))
    t0 = time.perf_counter()
    ens = S.SpeedyEns(members_total, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 3, 1), devices=used)
    cnts = [m._state_cnt for m in ens.members]
    ens.set_bc()  # member 0 reads the packaged boundary file; every other member receives the fields device to device
    peer, local, collective = drv.broadcast_boundary_stats()
    transport_note = drv.broadcast_boundary_note()  # (in words, with the reason when the collective was not used)
    bc = load_bc()
    probe = ens.members[-1]["orog"]  # what arrived in the last container (on the last device) is what the file holds
    arrived = bool(np.array_equal(probe, np.asarray(bc["orog"], dtype=np.float64)))
    for i, member in enumerate(ens.members):
        noise = np.random.default_rng(i).normal(0.0, 0.01, (96, 48, 8))
        member["t_grid"] = member["t_grid"] + noise
        member.grid2spectral()
    t_setup = time.perf_counter() - t0
    devices = sorted({drv.modelstate_device(c) for c in cnts})
    timing = _time_container_loop(ens, steps)
    # (counted after the loop: SpeedyEns makes one device model per GPU for its own multi-step stretches, and a host that steps one
    # by one -- this loop -- gets two per GPU from 32 containers up at its first step, csrc/driver.cpp: regroup)
    models = len({drv.device_model(c)[0]._m.value for c in cnts})
    ms = timing["begin_end_ms_per_step"]
    out = {"workload": what, "processes": 1, "devices_asked": n_devices, "devices_used": len(devices), "containers": members_total,
           "members_per_device": [sum(1 for c in cnts if drv.modelstate_device(c) == d) for d in devices],
           "device_models": models, "boundary_broadcast": {"collective_devices": collective, "peer_copies": peer, "local_copies": local,
                                                         "transport": "rccl" if collective else ("peer copies" if peer else "local copies only"),
                                                         "arrived_intact": arrived, "note": transport_note},
           "setup_seconds": t_setup, "ms_per_step": ms,
           "value": members_total * 86400.0 / (ms * 1e-3 * STEPS_PER_YEAR), "unit": "simulated-years/day",
           "current_device_preserved": torch.cuda.current_device() == before}
    out.update(timing)
    out["note"] = ("one process, spd_parallel_step_begin / _end once per model step over all containers (ms_per_step, value) and the "
                   "synchronous spd_parallel_step (sync_ms_per_step); per-step wall times, medians; every step stores all diagnostics "
                   "and runs the range check")
    del ens
    return out


def pyspeedy_amd_lib():
    import pyspeedy_amd
    return pyspeedy_amd.lib()


def plan_name(cfg, M):
    """cfg: model.config(), read AFTER the run for the last clause (the group streams exist from the first multi-step call on)"""
    g = cfg["chunks"]
    if g <= 1:
        return "serial: one member group on one stream"
    rounds = cfg.get("rounds", 1)
    if rounds > 1:  # (multi-step calls of large ensembles: model.hip, block_members)
        per = (M + rounds - 1) // rounds
        plan = "%d rounds of %d members, each through all steps of a call in %d member groups on %d HIP streams" % (rounds, per, g, g)
    else:
        plan = "%d member groups of %d / %d members on %d HIP streams" % (g, (M + g - 1) // g, M // g, g)
    if cfg.get("group_streams", 0) > 1:  # (csrc/stream_apart.hpp: measured when the streams were created)
        plan += " (measured: side by side)" if cfg["group_streams_apart"] else " (NOT measured to be all side by side -- two may share a hardware queue)"
    return plan


def stream_ceiling_object(sp, kernels):
    """roofline.stream_ceiling: what kernels of this library that only move bytes reach on THIS box, in the shapes of the step's
    kernels (pyspeedy_amd/stream_probe.py, spd_stream_probe) -- the ceiling to read the fractions of 8 TB/s against -- and the
    column kernel's twin: its own launch (streams, bytes per lane, occupancy, rows per wavefront, size) with the arithmetic taken
    out.  Every row of `kernels` gets `frac_of_stream_ceiling` (the best shape of its stream mix)."""
    from pyspeedy_amd import stream_probe as P
    L = sp._lib
    out = P.ceiling(L, sp.handle)
    mix_of = {"column": "column_2r1w", "column_sw": "column_2r1w", "physics": "column_2r1w", "physics_sw": "column_2r1w",
              "spec2grid": "copy_1r1w", "grid2spec": "copy_1r1w", "spectral_step": "copy_1r1w", "geopotential": "read",
              "coupler": "copy_1r1w", "forcing": "write", "dyn_grid": "copy_1r1w", "sppt": "copy_1r1w"}
    for k in kernels or []:
        mix = mix_of.get(k["kernel"])
        if mix:
            k["frac_of_stream_ceiling"] = k["achieved"] / 1e3 / out[mix]
            k["stream_ceiling_mix"] = mix
    col = [k for k in kernels or [] if k["kernel"] == "column"]
    if col:
        twin = P.column_twin(L, sp.handle, col[0]["algorithmic_bytes_per_launch"])
        out["column_twin"] = {
            "us": round(twin["us"], 2), "us_best": round(twin["us_best"], 2), "tb_s": round(twin["tb_s"], 3), "bytes": twin["bytes"],
            "column_kernel_us": col[0]["avg_launch_us"], "column_kernel_over_twin": col[0]["avg_launch_us"] / twin["us"],
            "note": "the column kernel's launch with the arithmetic taken out: 2 reads : 1 write, one double per lane, two "
                    "wavefronts per SIMD, 243 rows per wavefront each in an array of its own, non-temporal, the same bytes"}
    return out


def beyond_infinity_cache_leg(args, device, dist, rank, coll_device, barrier):
    """roofline.frac_beyond_infinity_cache: the line's kernel (spec2grid_table_kernel) at 256 members, serial plan, no rounds -- its
    input (the spectral state the previous launch wrote) and output are then far beyond the 256 MB Infinity Cache, which at 64
    members still holds a part of them (FETCH_SIZE counts such hits as fabric reads: `frac` is HBM + Infinity Cache)."""
    M = 256
    sp, model = build_ensemble(args, M, 0, device, dist, rank, coll_device)
    model.set_option("member_groups", 1)
    model.set_option("block_members", 0)
    model.run(36)
    model.profile(1)
    model.run(36)
    import torch
    torch.cuda.synchronize()
    kern_ms, launches, nfields = model.profile_read()
    model.profile(0)
    ok = (model.check(2) == 0).all()
    model.close()
    sp.close()
    if not ok:
        raise SystemExit("bench.py: members left the accepted range in the beyond_infinity_cache leg")
    nbytes = (S_BYTES + G_BYTES) * nfields
    return {"members": M, "fields_per_launch": nfields, "avg_launch_ms": kern_ms, "launches_timed": launches,
            "achieved": nbytes / (kern_ms * 1e-3) / 1e9, "unit": "GB/s", "frac": nbytes / (kern_ms * 1e-3) / 1e9 / 8000.0}


def kernel_sources_sha():
    """sha256 over the device sources of the library (csrc/*.hip, *.hpp in name order): what a committed PMC measurement was taken
    with (tools/pmc_summary.py writes it into every profiles/*_pmc_*.json) against what this tree holds"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pyspeedy_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pyspeedy_amd", "csrc", "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_FILES = ("r06_pmc_model_step.json", "r05_pmc_model_step.json", "r04_pmc_model_step.json", "r03_pmc_model_step.json",
             "r02_pmc_model_step.json", "r01_pmc_model_step.json")


def load_traffic(nfields):
    """HBM bytes per spec2grid launch from the committed PMC measurement of this very kernel inside this bench (rocprofv3
    cannot run inside bench.py): FETCH_SIZE doubled as the gfx950 guide prescribes, scaled per field."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                doc = json.load(fh)
            tj = doc["kernels"]["spd::spec2grid_table_kernel"]
            per_field = tj["hbm_bytes_per_launch"] / float(tj.get("fields_per_launch", 5824))
            # (the PMC passes cannot run inside bench.py: the figure is the committed one -- stale when the device sources have
            # changed since it was taken; None: the file is older than the stamp)
            taken_with = doc.get("kernel_sources_sha")
            stale = None if taken_with is None else taken_with != kernel_sources_sha()
            return per_field * nfields, "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py)" % name, stale
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def dominant_kernel(kernels, M, config, stream_ceiling=None):
    """The kernel that takes the largest share of the step -- the fused column kernel (grid-point dynamics + physics) -- priced
    like `roofline`: the launches of one simulated day (shortwave and other steps together), algorithmic bytes of its argument
    list / its dispatch-attached HIP-event time, and the committed PMC traffic of the same kernel per member."""
    rows = [k for k in kernels or [] if k["kernel"] in ("column", "column_sw")]
    if not rows:
        return None
    t_us = sum(k["avg_launch_us"] * k["launches_timed"] for k in rows)
    nbytes = sum(k["algorithmic_bytes_per_launch"] * k["launches_timed"] for k in rows)
    n = sum(k["launches_timed"] for k in rows)
    all_us = sum(k["avg_launch_us"] * k["launches_timed"] for k in kernels)
    out = {"kernel": "physics_kernel, fused (grid-point dynamics + column physics), %d members per launch" % M,
           "share_of_kernel_time": t_us / all_us, "launches_timed": n, "avg_launch_us": t_us / n,
           "algorithmic_bytes_per_launch": int(round(nbytes / n)), "bound": "hbm", "achieved": nbytes / (t_us * 1e-6) / 1e9,
           "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (t_us * 1e-6) / 1e9 / 8000.0, "traffic": None}
    if stream_ceiling and "column_2r1w" in stream_ceiling:  # (against this box's own 2r : 1w copy kernels, best shape / the kernel's shape)
        out["frac_of_stream_ceiling"] = out["achieved"] / 1e3 / stream_ceiling["column_2r1w"]
        out["frac_of_stream_ceiling_in_its_shape"] = out["achieved"] / 1e3 / stream_ceiling["column_2r1w_column_shape"]
    names, key, members = ((PMC_FILES[:2], "spd::physics_kernel<2, true, false, double, false>", 64) if config == "cfg4" else
                           (("r06_pmc_cfg5_storage32_1.json", "r05_pmc_cfg5_storage32_1.json"), "spd::physics_kernel<3, true, true, float, true>", 32))
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                doc = json.load(fh)
            out["traffic"] = doc["kernels"][key]["hbm_bytes_per_launch"] / float(members) * M
            out["traffic_source"] = "profiles/%s (%d members per launch there; scaled per member)" % (name, members)
            out["traffic_stale"] = None if doc.get("kernel_sources_sha") is None else doc["kernel_sources_sha"] != kernel_sources_sha()
            break
        except (OSError, KeyError, ValueError):
            continue
    return out


def job_file(what):
    """path under /tmp of a host-side meeting point of the ranks of ONE launch: keyed by the rendezvous port, the launcher's pid
    (the ranks of one job are children of one launcher process) and -- a worker group that torchrun restarts keeps both -- the
    launcher's restart count and run id, so that the ranks of a restarted group never read what an earlier attempt left there"""
    import tempfile
    nonce = "%s_%s" % (os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    nonce = "".join(ch if ch.isalnum() else "-" for ch in nonce)[:48]
    return os.path.join(tempfile.gettempdir(), "pyspeedy_bench_%s_%d_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), nonce, what))


def wait_for_ranks(rank, world, what, seconds=600.0):
    """A file barrier under /tmp keyed by the rendezvous port: used ONCE, before any process group exists, so that rank 0
    measures the host baseline while the other ranks have finished importing torch and sit idle."""
    base = job_file(what)
    open("%s.%d" % (base, rank), "w").close()
    deadline = time.time() + seconds
    while time.time() < deadline:
        if all(os.path.exists("%s.%d" % (base, r)) for r in range(world)):
            if rank != 0:  # (rank 0 reads the files last: it is the one that waits for all of them)
                return True
            time.sleep(0.5)
            for r in range(world):
                try:
                    os.unlink("%s.%d" % (base, r))
                except OSError:
                    pass
            return True
        time.sleep(0.05)
    return False


def file_flag(what, set_it=False, wait_seconds=0.0):
    """A flag file shared by the ranks of one job (keyed like wait_for_ranks): set it, or wait on the HOST until it is there."""
    path = job_file(what)
    if set_it:
        open(path, "w").close()
        return True
    deadline = time.time() + wait_seconds
    while time.time() < deadline:
        if os.path.exists(path):
            return True
        time.sleep(0.05)
    return False


def run_bounded_child(cmd, env, timeout):
    """Run `cmd` to completion or for `timeout` seconds, whichever comes first; -> (returncode or None, stdout, stderr).  Unlike
    subprocess.run(timeout=...) this never waits for a child it could not end: output goes to files, not pipes, and a child that
    survives SIGKILL for 10 s (stuck in the kernel on a wedged GPU) is left behind."""
    import tempfile
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        child = subprocess.Popen(cmd, env=env, stdout=out, stderr=err, text=True, start_new_session=True)
        deadline = time.time() + timeout
        while child.poll() is None and time.time() < deadline:
            time.sleep(0.05)
        code = child.poll()
        if code is None:
            child.kill()  # (exactly the process started here)
            try:
                child.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        out.seek(0)
        err.seek(0)
        return code, out.read(), err.read()


def one_process_child(n_gpus, extra, timeout=ONE_PROCESS_TIMEOUT):
    """`bench.py --one-process --gpus N ...` as a child process; its `one_process` object, or {"error": ...}.
    (PYSPEEDY_AMD_BENCH_ONE_PROCESS_CMD: a command to start instead -- the tests put a child there that never answers.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                             "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "PYSPEEDY_AMD_BENCH_T0")}
    cmd = [sys.executable, os.path.abspath(__file__), "--one-process", "--gpus", str(n_gpus), "--steps", "200", "--no-cpu-baseline",
           "--budget", str(int(timeout))]
    if os.environ.get("PYSPEEDY_AMD_BENCH_ONE_PROCESS_CMD"):
        import shlex
        cmd, extra = shlex.split(os.environ["PYSPEEDY_AMD_BENCH_ONE_PROCESS_CMD"]), []
    try:
        t0 = time.time()
        code, out, err = run_bounded_child(cmd + extra, env, timeout)
        if code is None:
            return {"error": "no answer within %d s (the child was killed after %.1f s); the line's other objects do not depend on it"
                             % (timeout, time.time() - t0)}
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if code != 0 or len(lines) != 1:
            return {"error": "exit code %s: %s" % (code, (err or out)[-600:])}
        return json.loads(lines[0])["one_process"]
    except (OSError, ValueError, KeyError) as exc:
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


def one_process_timeouts(budget):
    """(timeout of the one_process child, timeout of its cfg4_strong child) for the time that is left: at most 100 + 50 s, at most
    a third and a sixth of the budget, and 0 = do not start when less than 20 s would remain for it."""
    left = budget.left() - 15.0  # (what the line still needs afterwards: the last barrier, printing)
    t1 = min(ONE_PROCESS_TIMEOUT, budget.seconds / 3.0, left)
    t2 = min(ONE_PROCESS_STRONG_TIMEOUT, budget.seconds / 6.0, left - t1)
    floor = min(20.0, budget.seconds / 15.0)
    return (t1 if t1 >= floor else 0.0), (t2 if t1 >= floor and t2 >= floor / 2 else 0.0)


def run_one_process(args):
    """`--one-process`: this process alone drives --gpus devices; the line's value is the begin / end loop over all containers."""
    baseline = None if args.no_cpu_baseline else cpu_baseline(args.cpu_seconds)  # (before the GPU is touched)
    import torch
    import pyspeedy_amd
    pyspeedy_amd.lib()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(0)
    per_gpu_default, total_default = (64, 64)
    if args.config != "cfg4":
        raise SystemExit("bench.py: --one-process measures cfg4")
    if args.scaling == "weak":
        per = args.members if args.members is not None else per_gpu_default
        total = per * args.gpus
    else:
        total = args.members if args.members is not None else total_default
    op = one_process_leg(args.gpus, total, max(args.steps, 20), "BASELINE cfg4 (%s scaling), %d members, ONE process over %d GPUs"
                         % (args.scaling, total, args.gpus))
    all_cores = (baseline or {}).get("all_cores")
    line = {"metric": "simulated-years/day (whole node), T30L8", "value": op["value"], "unit": "simulated-years/day",
            "n_gpus": args.gpus, "steps": op["steps_timed"], "warmup": 12, "ms_per_step": op["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": (op["value"] / all_cores["value"]) if all_cores else None,
            "dtype": "f64", "data": "reference example_bc boundary fields (committed fixture); members perturbed with t_grid += "
                                    "N(0, 0.01 K), seed = member id",
            "config": {"workload": op["workload"], "members_total": total, "parallelism": "one process, spd_parallel_step over all "
                       "containers, members in blocks per GPU, no collective in the step", "backend": "none (one process; boundary "
                       "fields device to device, hipMemcpyPeerAsync)"},
            "one_process": op}
    if baseline is not None:
        line["cpu_baseline"] = baseline
    print(json.dumps(line), flush=True)


def rccl_probe_child():
    """`bench.py --rccl-probe` (a child of a rank, with the rank's environment and a rendezvous port of its own): the ranks' RCCL
    world in miniature -- process group with a device id, an all-reduce and a broadcast on device buffers, a barrier."""
    from datetime import timedelta
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=device, timeout=timedelta(seconds=60))
    t = torch.full((1 << 16,), float(rank + 1), dtype=torch.float64, device=device)
    dist.all_reduce(t)
    dist.broadcast(t, src=0)
    torch.cuda.synchronize()
    ok = abs(float(t[0]) - world * (world + 1) / 2.0) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_PROBE_OK" if ok else "RCCL_PROBE_WRONG_SUM", flush=True)


def rccl_preflight(rank, world, wall):
    """Before the ranks commit to RCCL for their barriers, their all-reduced times and the start-up broadcast: every rank starts a
    CHILD that forms the same RCCL world on another port and runs three collectives, bounded in time.  Rank 0 collects the
    verdicts (files under /tmp, as for the other host-side meeting points) and decides for everybody: RCCL when every child came
    back good, gloo otherwise -- with the reason in the line (`collective.backend_fallback`).  A node on which RCCL cannot be
    initialised, or hangs at its first collective, then still produces the scaling line, timed the same way; the members never
    exchange data in a step, so the transport of the barrier does not enter `value`.  -> (use RCCL?, reason, seconds spent)"""
    t0 = time.time()
    timeout = max(20.0, min(90.0, wall.left() / 4.0))
    # (without torchrun's own variables: under TORCHELASTIC_USE_AGENT_STORE the ranks expect the launcher's agent to host the
    # rendezvous store, and nobody hosts one on the probe's port -- the children would wait for it until their time is up)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k != "PYSPEEDY_AMD_BENCH_T0"}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for stale in ["%s.%d" % (job_file("rccl"), rank), "%s.ack.%d" % (job_file("rccl"), rank)] + ([job_file("rccl") + ".decision"] if rank == 0 else []):
        try:  # (what an earlier attempt under the same key may have left: a rank's own verdict and acknowledgement, rank 0's decision)
            os.unlink(stale)
        except OSError:
            pass
    code, out, err = run_bounded_child([sys.executable, os.path.abspath(__file__), "--rccl-probe"], env, timeout)
    def reason(text):  # the exception's own line, not the warnings the runtime prints on its way out
        lines = [ln.strip() for ln in text.splitlines() if ln.strip() and not ln.lstrip().startswith(("[W", "[I", "warnings.warn"))]
        import re
        named = [ln for ln in lines if re.search(r"\b\w+(Error|Exception)\b", ln)]  # "torch.distributed.DistBackendError: ..."
        errors = [ln for ln in lines if "error" in ln.lower() and not ln.lower().startswith("last error")]
        return (named or errors or lines or [""])[-1][:240]
    mine = "ok" if code == 0 and "RCCL_PROBE_OK" in out else (
        "no answer within %d s" % timeout if code is None else "exit code %s: %s" % (code, reason(err or out)))
    base = job_file("rccl")
    with open("%s.%d.tmp" % (base, rank), "w") as fh:
        fh.write(mine)
    os.replace("%s.%d.tmp" % (base, rank), "%s.%d" % (base, rank))  # (never seen half written)
    deadline = time.time() + timeout + 30.0
    if rank == 0:
        verdicts = {}
        while time.time() < deadline and len(verdicts) < world:
            for r in range(world):
                if r not in verdicts and os.path.exists("%s.%d" % (base, r)):
                    text = open("%s.%d" % (base, r)).read()
                    if text:
                        verdicts[r] = text
            time.sleep(0.05)
        bad = ["rank %d: %s" % (r, verdicts.get(r, "no verdict")) for r in range(world) if verdicts.get(r) != "ok"]
        decision = "ok" if not bad else "; ".join(bad)
        with open(base + ".decision.tmp", "w") as fh:
            fh.write(decision)
        os.replace(base + ".decision.tmp", base + ".decision")
        # everybody acknowledges the decision; then the files go (a later launch that ends up with the same key finds nothing)
        seen = time.time() + 20.0
        while time.time() < seen and not all(os.path.exists("%s.ack.%d" % (base, r)) for r in range(1, world)):
            time.sleep(0.05)
        for path in [base + ".decision"] + ["%s.%d" % (base, r) for r in range(world)] + ["%s.ack.%d" % (base, r) for r in range(1, world)]:
            try:
                os.unlink(path)
            except OSError:
                pass
    else:
        decision = None
        while time.time() < deadline + 15.0 and decision is None:
            if os.path.exists(base + ".decision"):
                decision = open(base + ".decision").read()
            else:
                time.sleep(0.05)
        if decision is None:
            decision = "rank %d saw no decision of rank 0" % rank
        open("%s.ack.%d" % (base, rank), "w").close()
    return decision == "ok", decision, time.time() - t0


def run_rank(args):
    baseline = None
    wall = Budget(args.budget)
    world_env, rank_env = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    import torch  # (importing torch does not initialise the GPU; the first import on a fresh box takes a minute or two)
    if os.environ.get("PYSPEEDY_AMD_BENCH_CPU_BASELINE"):  # measured by bench.py's launcher before it started the ranks
        if rank_env == 0:
            with open(os.environ["PYSPEEDY_AMD_BENCH_CPU_BASELINE"]) as fh:
                baseline = json.load(fh)
    elif not args.no_cpu_baseline:
        # spawns processes: must come before the first GPU call of this process.  Under torch.distributed.run rank 0 measures
        # it while the other ranks -- torch imported, GPU untouched -- wait for it in the rendezvous of the process group.
        if world_env > 1:
            wait_for_ranks(rank_env, world_env, "imported", seconds=max(30.0, wall.left() - 60.0))
        if rank_env == 0:
            baseline = cpu_baseline(args.cpu_seconds)

    backend = os.environ.get("PYSPEEDY_AMD_BENCH_BACKEND", "nccl")
    preflight = None
    if backend == "nccl" and world_env > 1 and os.environ.get("PYSPEEDY_AMD_BENCH_RCCL_PROBE", "1") != "0":
        # starts a child process: like the host baseline, before this process makes its first GPU call
        use_rccl, why, spent = rccl_preflight(rank_env, world_env, wall)
        preflight = {"ok": use_rccl, "seconds": round(spent, 1)}
        if not use_rccl:
            backend, preflight["why"] = "gloo", why

    import pyspeedy_amd
    from pyspeedy_amd import ensemble as E
    pyspeedy_amd.lib()  # load (or fail loudly) before the GPU is initialised; there is no CPU fallback
    world, rank, local = E.dist_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    local = local % torch.cuda.device_count()  # (several ranks may share a GPU in a rehearsal on a one-GPU box)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # RCCL ("nccl" on ROCm): barrier, max-over-ranks time and the start-up broadcast of the boundary fields.
    # PYSPEEDY_AMD_BENCH_BACKEND=gloo rehearses the same control flow when the ranks cannot each have their own GPU.
    # (`backend`, `preflight`: decided above, before the first GPU call of this process)
    # the process group is also created for an explicit one-rank world (WORLD_SIZE=1 in the environment): RCCL end to end
    dist = E.init_process_group(backend, device, force="WORLD_SIZE" in os.environ)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    n_gpus = dist.get_world_size() if dist is not None else 1
    assert n_gpus == args.gpus, (n_gpus, args.gpus)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    M, first_id, total_members = workload(args, world, rank)
    sp, model = build_ensemble(args, M, first_id, device, dist, rank, coll_device)
    collective = collective_record(dist, backend, device, coll_device, rank, model.bc_checksum, model.bc_bytes)
    if preflight is not None:
        collective["rccl_preflight"] = preflight
        if not preflight["ok"]:
            collective["backend_fallback"] = ("RCCL did not pass its pre-flight (%s): barriers, all-reduced times and the start-up "
                                              "broadcast go through gloo; the step itself has no collective" % preflight["why"])
    if not collective["boundary_checksum_equal"] or collective["ranks_seen"] != n_gpus:
        raise SystemExit("bench.py: the start-up broadcast did not deliver the same boundary fields to %d ranks: %r" % (n_gpus, collective))
    if args.serial_plan:
        model.set_option("member_groups", 1)
    cfg = model.config()
    sppt = args.config == "cfg5"
    plan = plan_name(cfg, M)
    model.run(args.warmup)
    budget = MAX_STEPS - args.warmup - 36  # model steps this ensemble may still take (36: the bracketed one-day pass)
    # ---- headline: the library's default plan
    region_s = timed_regions(model, args, barrier, dist, coll_device, budget * 3 // 4, args.min_seconds)
    budget -= len(region_s) * args.steps
    plan = plan_name(model.config(), M)
    # (every rank's group streams, not only rank 0's: one all_gather_object; a rank whose streams share a hardware queue is the
    # slowest rank, and the line's time is the slowest rank's)
    apart_by_rank = [bool(model.config()["group_streams_apart"])]
    if dist is not None:
        apart_by_rank = [None] * n_gpus
        dist.all_gather_object(apart_by_rank, bool(model.config()["group_streams_apart"]))
    # ---- roofline: the dominant transform kernel by HIP events on its launch stream, serial plan (profiling implies it)
    model.profile(1)
    saved_regions, args.regions = args.regions, (min(args.regions, 10) if args.regions else 0)
    serial_s = timed_regions(model, args, barrier, dist, coll_device, budget, min(args.min_seconds, 1.0), 200)
    args.regions = saved_regions
    kern_ms, launches, nfields = model.profile_read()
    codes = model.check(2)
    if (codes != 0).any():
        raise SystemExit("bench.py: %d members left the accepted range (diagnostics.f90)" % int((codes != 0).sum()))
    model.profile(0)
    kernels = kernel_table(model, M, nfields // M, sppt) if rank == 0 or dist is None else None
    model.close()
    stream_ceiling = None
    if rank == 0:  # (own probe kernels on the idle GPU of rank 0; the other ranks' GPUs have nothing timed in flight either)
        stream_ceiling = (stream_ceiling_object(sp, kernels) if wall.allows("stream_ceiling", LEG_ALLOWANCE["stream_ceiling"])
                          else {"skipped": "budget"})
    sp.close()
    legs = {}
    if not args.no_legs:
        def leg(name, fn):  # a secondary object of the one-GPU line: only while the budget covers its allowance
            legs[name] = fn() if wall.allows(name, LEG_ALLOWANCE[name]) else {"skipped": "budget"}

        if n_gpus == 1 and args.config == "cfg4":
            leg("every_step_stores", lambda: fidelity_leg(args, M, first_id, device, dist, rank, coll_device, barrier))
            leg("beyond_infinity_cache", lambda: beyond_infinity_cache_leg(args, device, dist, rank, coll_device, barrier))

            def drop_in():
                out = drop_in_leg(M, 360)
                out.update(small_drop_in_legs(360))
                return out
            leg("drop_in_step", drop_in)
            leg("facade_run", facade_leg)
            # the other BASELINE configs on the same clock (SURVEY 8d "Configs as concrete inputs")
            leg("cfg2_transforms", lambda: transforms_leg(args, device))
            leg("cfg3", lambda: config_leg(args, "cfg4", 1, "BASELINE cfg 3: one member, fp64, the full step on the GPU", device, dist, rank,
                                           coll_device, barrier))
            leg("cfg4_shard8", lambda: config_leg(args, "cfg4", 8, "BASELINE cfg 4 as worded: one GPU's share of 64 members on 8 GPUs = 8 "
                                                  "members, fp64", device, dist, rank, coll_device, barrier))
            leg("cfg5", lambda: config_leg(args, "cfg5", 32, "BASELINE cfg 5: one GPU's share of 256 members on 8 GPUs = 32 members, SPPT "
                                           "on, fp32 arithmetic in the column physics (fp64 state)", device, dist, rank, coll_device, barrier))
        if n_gpus > 1 and args.scaling == "weak" and args.config == "cfg4" and args.members is None and not agreed(
                dist, wall.allows("cfg4_strong", LEG_ALLOWANCE["cfg4_strong"])):
            legs["cfg4_strong"] = {"skipped": "budget"}
        elif n_gpus > 1 and args.scaling == "weak" and args.config == "cfg4" and args.members is None:
            # BASELINE cfg 4 to the letter next to the weak headline: 64 members in total, block-sharded over the ranks
            strong = argparse.Namespace(**dict(vars(args), scaling="strong", members=None))
            Ms, first_s, total_s = workload(strong, world, rank)
            sp2, m2 = build_ensemble(strong, Ms, first_s, device, dist, rank, coll_device)
            m2.run(args.warmup)
            secs = timed_regions(m2, strong, barrier, dist, coll_device, MAX_STEPS - args.warmup - 36, 1.0, 500)
            ok = (m2.check(2) == 0).all()
            scfg = m2.config()
            m2.close()
            sp2.close()
            if not ok:
                raise SystemExit("bench.py: members left the accepted range in the cfg4_strong leg")
            ms_s = median(secs) / args.steps * 1e3
            legs["cfg4_strong"] = {
                "members_total": total_s, "members_per_gpu": Ms, "ms_per_step": ms_s, "regions": len(secs), "scaling": "strong",
                "value": E.simulated_years_per_day(total_s, ms_s * 1e-3, STEPS_PER_YEAR), "unit": "simulated-years/day",
                "plan": plan_name(scfg, Ms),
                "note": "BASELINE cfg 4 as worded: 64 members sharded %d per GPU over %d GPUs, same timing rules" % (Ms, n_gpus)}
        if n_gpus > 1 and args.config == "cfg4":
            # The reference's own shape beside the process-per-GPU headline: ONE process (rank 0) drives every GPU through
            # spd_parallel_step.  The other ranks have released their models and wait on the host (a file, not a collective: a
            # pending RCCL barrier would keep a kernel spinning on the very GPUs that are being measured).
            # The measurement runs in a CHILD process of rank 0 (`bench.py --one-process`): it is ONE process by construction,
            # and whatever its first contact with a second GPU does, the headline of this line survives it.
            # Its children get timeouts cut from what is left of the budget (at most 100 + 50 s), rank 0 decides and tells the
            # others, and they wait exactly that long: whatever the first contact with a second GPU does inside this SECONDARY
            # object, every rank is back in time to print the line.
            t_child, t_strong = agreed(dist, one_process_timeouts(wall))
            if not (args.scaling == "weak" and args.members is None):
                t_strong = 0.0
            barrier()
            if t_child <= 0:
                legs["one_process"] = {"skipped": "budget"}
                if rank == 0:
                    wall.skipped.append({"leg": "one_process", "allowance_s": ONE_PROCESS_TIMEOUT, "left_s": round(wall.left(), 1)})
            elif rank == 0:
                op = one_process_child(n_gpus, ["--scaling", "strong", "--members", str(total_members)], timeout=t_child)
                if "error" not in op and t_strong > 0:
                    op["cfg4_strong"] = one_process_child(n_gpus, ["--scaling", "strong", "--members", "64"], timeout=t_strong)
                op["timeouts_s"] = [round(t_child, 1), round(t_strong, 1)]
                legs["one_process"] = op
                file_flag("one_process_done", set_it=True)
            else:
                file_flag("one_process_done", wait_seconds=t_child + t_strong + 25.0)  # (+ the grace of two killed children)

    if rank == 0:
        ms_step, ms_min = median(region_s) / args.steps * 1e3, min(region_s) / args.steps * 1e3
        value = E.simulated_years_per_day(total_members, ms_step * 1e-3, STEPS_PER_YEAR)
        # (cfg 5 stores 27 of a member's fields as fp32: those count S + G / 2)
        s2g_bytes = (S_BYTES + G_BYTES) * nfields - (G_BYTES // 2) * S2G_FLOAT_FIELDS * M * (1 if cfg["physics_storage32"] else 0)
        achieved = s2g_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_stale = load_traffic(nfields)
        beyond = legs.pop("beyond_infinity_cache", None)
        physics = "fp64 column physics" if args.config == "cfg4" else "SPPT on, fp32 arithmetic in the column physics (fp64 state)"
        all_cores = (baseline or {}).get("all_cores")
        line = {
            "metric": "simulated-years/day (whole node), T30L8", "value": value, "unit": "simulated-years/day",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": (value / all_cores["value"]) if all_cores else None,
            "vs_baseline_note": "BASELINE.md holds no published number for this metric; the ratio is to cpu_baseline.all_cores (the "
                                "reference Fortran, one member per core on the %s host cores of this box), the comparison "
                                "north_star asks for" % (all_cores["cores"] if all_cores else "?"),
            "dtype": "f64" if args.config == "cfg4" else "f64 state and dynamics, f32 column physics",
            "data": "reference example_bc boundary fields (committed fixture, no download); state generated by the model: "
                    "resting atmosphere + first_step, members perturbed with t_grid += N(0, 0.01 K) (seed = global member "
                    "id), %d spin-up steps" % args.warmup,
            "ms_per_step_min": ms_min, "regions": len(region_s), "timed_seconds": sum(region_s),
            "config": {
                "workload": "BASELINE %s ensemble (%s scaling): %d members per GPU, %d in total, T30L8 96x48x8, %s; full "
                            "do_single_step per member (%d spec2grid + grid-point dynamics + fused column physics + 73 "
                            "grid2spec + spectral tendencies/semi-implicit/diffusion/RAW filter + coupler + daily "
                            "forcing), all on the GPU" % (args.config, args.scaling, M, total_members, physics, nfields // M),
                "members_per_gpu": M, "members_total": total_members, "plan": plan,
                "group_streams_side_by_side_by_rank": apart_by_rank,
                "ms_per_member_step": ms_step * n_gpus / total_members, "simulated_days_per_region": args.steps / 36.0,
                "parallelism": "ensemble members sharded per GPU, no collective in the step",
                "backend": backend if dist is not None else "none (single process)",
            },
            "collective": collective,
            "roofline": {
                "kernel": "spec2grid_table_kernel (inverse Legendre + inverse FFT-96, with vort2vel / gradient applied while "
                          "staging the wind and pressure-gradient fields of each member), %d fields/launch" % nfields,
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "algorithmic_bytes_per_field": S_BYTES + G_BYTES, "algorithmic_bytes_per_launch": s2g_bytes,
                "avg_launch_ms": kern_ms, "launches_timed": launches,
                "measured_in": "%d further regions of %d steps issued in the serial plan (one member group on one stream: the "
                               "duration of a kernel that shares the GPU with another group's kernels is not its own); HIP events "
                               "attached to the dispatch of every spec2grid launch of those regions" % (len(serial_s), args.steps),
                "serial_plan_ms_per_step": median(serial_s) / args.steps * 1e3,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                "kernel_sources_sha": kernel_sources_sha(),
                # `frac` is HBM + Infinity Cache at 64 members (a part of what this launch reads and writes is still in the 256 MB
                # cache); the same kernel where nothing is: 256 members, serial plan
                "frac_beyond_infinity_cache": (beyond or {}).get("frac"), "beyond_infinity_cache": beyond,
                "stream_ceiling": stream_ceiling,
                "frac_of_stream_ceiling": (achieved / 1e3 / stream_ceiling["copy_1r1w"]) if stream_ceiling and "copy_1r1w" in stream_ceiling else None,
                "dominant": dominant_kernel(kernels, M, args.config, stream_ceiling), "kernels": kernels,
                "kernels_note": "one simulated day (36 steps, serial plan) with events attached to every kernel's dispatch (the "
                                "kernels' own begin / end time stamps, as in rocprofv3's kernel trace); the rows sum to less than "
                                "serial_plan_ms_per_step by the gaps between dependent launches",
            },
        }
        if "ms_per_step" in (legs.get("cfg4_shard8") or {}):
            legs["projected_8gpu_cfg4"] = projection_8gpu(legs["cfg4_shard8"], value, total_members, all_cores)
        line["config"].update(step_contract_keys(legs, ms_step, total_members))
        if n_gpus > 1:
            line["config"].update(scaling_keys(collective, legs))
        line.update(legs)
        line["budget"] = wall.record()
        if baseline is not None:
            line["cpu_baseline"] = baseline
            if all_cores:  # (flat, beside the one-core figure: the comparison north_star asks for is against all host cores)
                baseline["all_cores_value"], baseline["all_cores_cores"] = all_cores["value"], all_cores["cores"]
            if "cfg4_strong" in legs and all_cores:
                legs["cfg4_strong"]["vs_cpu_all_cores"] = legs["cfg4_strong"]["value"] / all_cores["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.cpu_worker is not None:
        return cpu_worker(*args.cpu_worker)
    if args.rccl_probe:
        return rccl_probe_child()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.one_process:
        return run_one_process(args)  # one process whatever --gpus says: no ranks are started
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)  # this process never touches the GPU
    run_rank(args)


if __name__ == "__main__":
    main()
