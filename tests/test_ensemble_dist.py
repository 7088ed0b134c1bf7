"""CPU tier: the N>1 path (member sharding, barrier, max-over-ranks timing, boundary broadcast, ensemble statistics)
with world_size 2 on gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_members_partitions_everything():
    from pyspeedy_amd.ensemble import shard_members
    for total in (0, 1, 7, 64, 65, 256):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count = shard_members(total, world, r)
                seen.extend(range(first, first + count))
            assert seen == list(range(total))
            counts = [shard_members(total, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_members(8, 2, 2)


def test_throughput_formula():
    from pyspeedy_amd.ensemble import simulated_years_per_day
    # one member at 14 ms/step (the flang reference on one core) -> ~470 simulated years per day
    assert abs(simulated_years_per_day(1, 14e-3) - 86400 / (14e-3 * 13140)) < 1e-9
    assert simulated_years_per_day(64, 1e-3) == 64 * simulated_years_per_day(1, 1e-3)


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %(root)r)
    from pyspeedy_amd import ensemble as E
    world, rank, local = E.dist_env()
    dist = E.init_process_group("gloo")
    assert dist is not None and dist.get_world_size() == 2
    first, count = E.shard_members(65, world, rank)
    dist.barrier()
    elapsed = 0.010 * (rank + 1)          # rank 1 is the slow one
    slowest = E.max_over_ranks(elapsed, dist)
    total = E.sum_over_ranks(count, dist)
    assert abs(slowest - 0.020) < 1e-12, slowest
    assert total == 65, total
    # start-up broadcast of the shared boundary fields (rank 0 "reads the file") ...
    import numpy as np, torch
    bc = None
    if rank == 0:
        g = np.random.default_rng(5)
        bc = {"orog": g.standard_normal((96, 48)), "sst": g.standard_normal((96, 48, 12)), "lsm": g.random((96, 48))}
    got = E.broadcast_boundary_conditions(bc, dist, "cpu")
    g = np.random.default_rng(5)
    for k, shape in (("orog", (96, 48)), ("sst", (96, 48, 12))):
        ref = g.standard_normal(shape)
        assert got[k].shape == shape and (got[k] == ref).all(), k
    assert sorted(got) == ["lsm", "orog", "sst"]
    # ... and ensemble statistics over the members of both ranks (3 on rank 0, 2 on rank 1)
    allv = np.random.default_rng(9).standard_normal((5, 4, 6))
    mine = torch.from_numpy(allv[:3] if rank == 0 else allv[3:])
    mean, spread = E.ensemble_mean_spread(mine, dist)
    assert np.allclose(mean.numpy(), allv.mean(axis=0), rtol=0, atol=1e-15)
    assert np.allclose(spread.numpy(), allv.std(axis=0, ddof=1), rtol=1e-14, atol=0)
    # ... and what bench.py's N-rank line records about the collective layer, gathered through it: both ranks seen, the
    # checksum of what each rank received in the broadcast, equal -- and unequal when a rank's fields were tampered with
    import bench
    cpu = torch.device("cpu")
    checksum, nbytes = bench.boundary_checksum(got)
    rec = bench.collective_record(dist, "gloo", cpu, cpu, rank, checksum, nbytes)
    assert rec["backend"] == "gloo" and rec["world_size"] == 2 and rec["ranks_seen"] == 2 and rec["device_of_rank"] == [-1, -1]
    assert rec["boundary_checksum_equal"] is True and len(set(rec["boundary_checksum_of_rank"])) == 1
    assert rec["boundary_bytes"] == nbytes == 8 * (96 * 48 * 14) and [g["rank"] for g in rec["gpu_of_rank"]] == [0, 1]
    if rank == 1:
        got["orog"][3, 4] += 1e-12
    bad = bench.collective_record(dist, "gloo", cpu, cpu, rank, bench.boundary_checksum(got)[0], nbytes)
    assert bad["boundary_checksum_equal"] is False and bad["ranks_seen"] == 2
    if rank == 0:
        print("RESULT", first, count, E.simulated_years_per_day(total, slowest))
    dist.barrier()
    dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = str(free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 0 and int(line[2]) == 33
    assert abs(float(line[3]) - 65 * 86400.0 / (0.020 * 13140)) < 1e-6


@pytest.mark.gpu
def test_sharded_ensemble_forecast_is_independent_of_the_sharding(tmp_path):
    """examples/ensemble_multi_gpu.py with 1 rank and with 2 ranks (gloo, both on the one GPU of the test box: the control
    flow of the multi-GPU run -- boundary broadcast, block sharding, per-rank batched model, all-reduced statistics): the
    ensemble mean and spread written by rank 0 agree (seeds are global member ids; only the reduction order differs)."""
    import numpy as np
    from pyspeedy_amd.dataset import open_dataset
    script = os.path.join(ROOT, "examples", "ensemble_multi_gpu.py")
    out1, out2 = str(tmp_path / "one"), str(tmp_path / "two")
    run = subprocess.run([sys.executable, script, "--members", "5", "--days", "1", "--out", out1], capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    env = dict(os.environ, PYSPEEDY_AMD_BACKEND="gloo")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), script, "--members", "5", "--days", "1", "--out", out2],
                         capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    a = open_dataset(os.path.join(out1, "tstat_1982-01-02_0000.nc"))
    b = open_dataset(os.path.join(out2, "tstat_1982-01-02_0000.nc"))
    for v in ("t_mean", "t_spread"):
        np.testing.assert_allclose(a[v].values, b[v].values, rtol=1e-6, atol=1e-7)
    assert float(a["t_spread"].values.max()) > 1e-3
    # ... and the reference's own shape, ONE process over the GPUs it can see (examples/ensemble_one_process.py: SpeedyEns with
    # devices=k, ens.set_bc(), one parallel_step per model step), gives the same forecast
    out3 = str(tmp_path / "one_process")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "ensemble_one_process.py"), "--members", "5", "--days", "1",
                          "--out", out3], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "5 members on device(s) [0]" in run.stdout and "36 steps" in run.stdout
    c = open_dataset(os.path.join(out3, "tstat_1982-01-02_0000.nc"))
    for v in ("t_mean", "t_spread"):
        np.testing.assert_allclose(a[v].values, c[v].values, rtol=1e-6, atol=1e-7)
