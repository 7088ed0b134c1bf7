"""bench.py as a launcher: `--gpus N` starts N rank processes (before touching the GPU), relays rank 0's line and fails when
a rank fails.  The GPU tests rehearse N = 2 on the one GPU of the test box (gloo carries the collectives, both ranks compute
on the same card) and run one explicit one-rank world through RCCL, so that the `nccl` path -- init with device_id, device-
buffer broadcast of the boundary fields, barrier, all-reduced time -- has executed on hardware."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_workload_partition():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--gpus", "8", "--scaling", "strong"])
    shards = [bench.workload(a, 8, r) for r in range(8)]
    assert [s[0] for s in shards] == [8] * 8 and [s[1] for s in shards] == list(range(0, 64, 8)) and shards[0][2] == 64
    a = bench.parse(["--gpus", "8"])
    assert bench.workload(a, 8, 3) == (64, 192, 512)  # weak: 64 per GPU, global ids do not overlap
    a = bench.parse(["--gpus", "8", "--config", "cfg5", "--scaling", "strong"])
    assert bench.workload(a, 8, 7) == (32, 224, 256)
    a = bench.parse(["--gpus", "4", "--config", "cfg5"])
    assert bench.workload(a, 4, 1) == (32, 32, 128)
    a = bench.parse(["--gpus", "3", "--scaling", "strong", "--members", "2"])
    with pytest.raises(SystemExit):
        bench.workload(a, 3, 2)


def test_launcher_reports_a_failing_rank():
    """Without a GPU every rank stops with an error: the launcher must return non-zero and print no result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    run = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode != 0
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert "rank" in run.stderr


def _result(run):
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu():
    env = dict(os.environ, PYSPEEDY_AMD_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    common = ["--steps", "6", "--warmup", "3", "--regions", "2", "--no-cpu-baseline"]
    strong = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--scaling", "strong", "--members", "6"] + common,
                                    capture_output=True, text=True, timeout=900, env=env))
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["config"]["members_per_gpu"] == 3 and strong["config"]["members_total"] == 6
    assert strong["regions"] == 2 and strong["ms_per_step_min"] <= strong["ms_per_step"]
    weak = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--members", "4"] + common, capture_output=True,
                                  text=True, timeout=900, env=env))
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak"
    assert weak["config"]["members_per_gpu"] == 4 and weak["config"]["members_total"] == 8
    years = 8 * 86400.0 / (weak["ms_per_step"] * 1e-3 * 13140)
    assert abs(weak["value"] - years) < 1e-6 * years
    names = {k["kernel"] for k in weak["roofline"]["kernels"]}
    assert {"spec2grid", "column_sw", "column", "grid2spec", "spectral_step"} <= names


@pytest.mark.gpu
def test_bench_one_rank_world_through_rccl():
    """WORLD_SIZE=1 in the environment makes the rank create a real `nccl` process group: the broadcast of the boundary
    fields, the barriers and the all-reduced time go through RCCL on device buffers."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    env.pop("PYSPEEDY_AMD_BENCH_BACKEND", None)
    env.pop("MASTER_PORT", None)
    res = _result(subprocess.run([sys.executable, BENCH, "--gpus", "1", "--members", "4", "--steps", "6", "--warmup", "3",
                                  "--regions", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env))
    assert res["n_gpus"] == 1 and res["config"]["backend"] == "nccl"
    assert res["config"]["members_total"] == 4 and res["value"] > 0
