"""Kernel variants that are chosen once per process (launch-bounds builds of the column kernel) must give the same bits: a
member's trajectory must not depend on which variant stepped it.  The device code is built with -ffp-contract=on for exactly
that (csrc/Makefile); with the compiler's default the 1- and 2-wave builds of the column kernel differed in 35 of 92 registry
variables after 40 steps.  Each variant runs in a process of its own (tests/dump_state.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _state(tmp_path, tag, **env):
    out = str(tmp_path / (tag + ".npz"))
    run = subprocess.run([sys.executable, os.path.join(HERE, "dump_state.py"), out], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env))
    assert run.returncode == 0, run.stderr[-2000:]
    return np.load(out)


@pytest.mark.gpu
def test_launch_bounds_builds_of_the_column_kernel_agree_bitwise(tmp_path):
    a = _state(tmp_path, "w2", PYSPEEDY_AMD_PHYS_WAVES="2")
    b = _state(tmp_path, "w1", PYSPEEDY_AMD_PHYS_WAVES="1")
    c = _state(tmp_path, "split", PYSPEEDY_AMD_SPLIT_DYN="1")
    assert len(a.files) > 80
    for n in a.files:
        assert np.array_equal(a[n], b[n]), ("1-wave build", n)
    # separate dynamics / physics launches write the dynamics' tendencies to memory and read them back: same values
    differing = [n for n in a.files if not np.array_equal(a[n], c[n])]
    assert not differing, ("split launches", differing[:5])


@pytest.mark.gpu
def test_launch_bounds_builds_of_the_fp32_column_kernel_agree_bitwise(tmp_path):
    """cfg 5: the 2- and 3-wave builds of the fp32 kernel are chosen by the size of the launch (physics.hip: physics_waves32),
    so a member's trajectory must not depend on which one stepped it; and keeping the physics-only arrays as fp64 instead of
    fp32 (PYSPEEDY_AMD_PHYS_STORE32=0) must not change a bit either."""
    a = _state(tmp_path, "f32w2", DUMP_STATE_CFG5="1", PYSPEEDY_AMD_PHYS_WAVES32="2")
    b = _state(tmp_path, "f32w3", DUMP_STATE_CFG5="1", PYSPEEDY_AMD_PHYS_WAVES32="3")
    c = _state(tmp_path, "f32w3s64", DUMP_STATE_CFG5="1", PYSPEEDY_AMD_PHYS_WAVES32="3", PYSPEEDY_AMD_PHYS_STORE32="0")
    plain = _state(tmp_path, "f64")
    assert len(a.files) > 80
    for other, what in ((b, "3-wave build"), (c, "fp64 storage")):
        differing = [n for n in a.files if not np.array_equal(a[n], other[n])]
        assert not differing, (what, differing[:5])
    assert not np.array_equal(a["t"], plain["t"])  # (the switch did switch something)


@pytest.mark.gpu
@pytest.mark.parametrize("idle_streams", [2, 1])
def test_group_streams_end_up_on_hardware_queues_of_their_own(idle_streams):
    """HIP spreads the streams of a process over a few hardware queues, idle ones counted, ties broken arbitrarily; two streams
    on one queue run strictly one after the other -- for a model's member groups the difference between 0.25 and 0.32 ms per
    step at 64 members.  The library measures what it was given and replaces a stream that does not run side by side with the
    others (csrc/stream_apart.hpp).  With one or two idle streams created first the first hand-out does collide (that is how the
    effect was found, tools/experiments/r04_idle_streams.py): every stream must end up side by side with every other."""
    run = subprocess.run([sys.executable, os.path.join(HERE, "streams_apart_child.py"), str(idle_streams)], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, PYSPEEDY_AMD_STREAMS_APART="2", PYSPEEDY_AMD_DRIVER_SPLIT="2"))
    assert run.returncode == 0, run.stderr[-3000:]
    model_part, driver_part = run.stderr.split("MODEL DONE")
    assert "DRIVER DONE" in driver_part

    def final_verdicts(text):
        """{(streams compared with, index): verdict of the LAST attempt} per created stream, in order of creation"""
        created, current = [], {}
        for line in text.splitlines():
            if not line.startswith("create_stream_apart: attempt"):
                continue
            attempt = int(line.split("attempt ")[1].split(",")[0])
            index, n = (int(v) for v in line.split("against stream ")[1].split(":")[0].split(" of "))
            if attempt == 0 and index == 0 and current:
                created.append(current)
                current = {}
            if attempt > current.get("attempt", 0):
                current = {"attempt": attempt}
            current["attempt"] = attempt
            current[(n, index)] = line.rstrip().endswith("side by side")
        return created + ([current] if current else [])

    groups = final_verdicts(model_part)
    assert len(groups) == 2, model_part  # (the second and the third group stream are measured; the first has nobody to meet)
    assert [sorted(k for k in g if k != "attempt") for g in groups] == [[(1, 0)], [(2, 0), (2, 1)]], groups
    assert all(v for g in groups for k, v in g.items() if k != "attempt"), model_part
    models = final_verdicts(driver_part)
    assert len(models) == 1 and all(v for k, v in models[0].items() if k != "attempt"), driver_part


@pytest.mark.gpu
def test_single_forecast_example(tmp_path):
    """examples/winter_week.py: a caller of the facade written for this repository -- Speedy, set_bc, XarrayExporter +
    ModelCheckpoint with a spin-up date, run, state access by name -- over the 1982/83 year end (5 days, 2 of them discarded);
    in a process of its own (it is a script)."""
    out = tmp_path / "data"
    run = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "examples", "winter_week.py"), "--from", "1982-12-29", "--days", "5",
                          "--discard", "2", "--dir", str(out)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    files = sorted(f for f in os.listdir(out) if f.endswith(".nc"))
    assert files == ["1982-12-31_0000.nc", "1983-01-01_0000.nc", "1983-01-02_0000.nc", "1983-01-03_0000.nc"], files
    assert "forecast ended 1983-01-03 00:00:00 after 180 steps; 4 daily files, 4 days in memory" in run.stdout
    assert "grid: 96 x 48, first latitude -87." in run.stdout
    from pyspeedy_amd.dataset import open_dataset
    last = open_dataset(str(out / files[-1]))
    assert set(("u", "v", "t", "q", "phi", "ps")) <= set(last.variables) and last.variables["t"].values.shape == (1, 8, 48, 96)
    t_low = last.variables["t"].values[0, 0]  # (lev is written bottom-up reversed: index 0 = sigma 0.95)
    assert 200.0 < t_low.min() < t_low.max() < 330.0


@pytest.mark.gpu
def test_ensemble_spread_example():
    """examples/ensemble_spread.py: members set up one by one (set_bc, `member["t_grid"] += ...`, grid2spectral), ModelCheckpoint +
    DiagnosticCheck, the spread statistics on the checkpoint dataframe (var / std / mean / apply / isel).  4 members and 7 days
    here; the spread of the perturbed members grows from day to day."""
    run = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "examples", "ensemble_spread.py"), "--size", "4",
                          "--days", "7", "--quiet-days", "1"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "ens: 4" in run.stdout and "time: 7" in run.stdout, run.stdout[-2000:]
    tail = run.stdout.split("domain-mean spread per kept day")[1]
    line = [ln for ln in tail.splitlines() if ln.strip().startswith("t ")][0]
    spread = [float(v) for v in line.split()[1:]]
    assert len(spread) == 7 and all(v > 0 for v in spread) and spread[-1] > spread[0], spread
    assert "temperature spread grew by a factor" in run.stdout
