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
    if os.path.exists("/dev/kfd"):  # (not torch.cuda.is_available(): that would initialise the GPU in the test process, which
        pytest.skip("needs a machine without a GPU")  # must stay able to start the other spawning tests)
    run = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode != 0
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert "rank" in run.stderr


def test_a_failing_one_process_child_costs_the_line_an_object_not_its_headline():
    """The one-process measurement of an N-rank line runs in a child process of rank 0; whatever happens to it (here: no GPU at
    all) comes back as {"error": ...} for the `one_process` object."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("needs a machine without a GPU")
    sys.path.insert(0, ROOT)
    import bench
    op = bench.one_process_child(2, ["--scaling", "strong", "--members", "4"], timeout=300)
    assert set(op) == {"error"} and "exit code" in op["error"]
    assert bench.file_flag("test_flag_%d" % os.getpid(), wait_seconds=0.1) is False
    assert bench.file_flag("test_flag_%d" % os.getpid(), set_it=True) and bench.file_flag("test_flag_%d" % os.getpid(), wait_seconds=1.0)


def test_budget_arithmetic():
    """The wall-clock budget of the line: secondary legs are refused once the remaining time does not cover their allowance, and
    the timeouts of the one_process children never sum to more than 150 s, whatever the budget."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse([]).budget == 300.0 and bench.parse(["--budget", "77"]).budget == 77.0
    import time
    b = bench.Budget(300.0)
    assert b.t0 == bench.T_PROCESS_START and b.left() <= 300.0  # (counted from the moment the bench module was loaded)
    b.t0 = time.time() - 150.0
    assert b.allows("cfg3", 10) and not b.allows("big", 200)
    assert b.record()["skipped"] == [{"leg": "big", "allowance_s": 200, "left_s": b.skipped[0]["left_s"]}] and 148 < b.skipped[0]["left_s"] < 151
    assert bench.ONE_PROCESS_TIMEOUT + bench.ONE_PROCESS_STRONG_TIMEOUT <= 150
    for seconds, used in ((300, 0), (300, 100), (300, 250), (300, 290), (600, 0), (60, 10), (30, 0), (10, 9)):
        b = bench.Budget(float(seconds))
        b.t0 = time.time() - used
        t1, t2 = bench.one_process_timeouts(b)
        assert 0 <= t1 <= bench.ONE_PROCESS_TIMEOUT and 0 <= t2 <= bench.ONE_PROCESS_STRONG_TIMEOUT and t1 + t2 <= 150
        assert t1 + t2 + 15 <= max(seconds - used, 15) + 1e-6, (seconds, used, t1, t2)  # the children end inside what is left
        if t1 == 0:
            assert t2 == 0
    fresh = bench.Budget(300.0)
    fresh.t0 = time.time()
    assert bench.one_process_timeouts(fresh) == (100, 50)
    spent = bench.Budget(300.0)
    spent.t0 = time.time() - 290
    assert bench.one_process_timeouts(spent) == (0.0, 0.0)
    assert sum(bench.LEG_ALLOWANCE.values()) < 200  # (all legs of the default one-GPU line fit the default budget)


def test_a_child_that_never_answers_is_cut_off():
    """The one_process object of an N-rank line is measured by a child process.  One that never answers (here: a sleeper put in
    its place) is ended at its timeout and comes back as {"error": ...}; the caller is not kept any longer."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    os.environ["PYSPEEDY_AMD_BENCH_ONE_PROCESS_CMD"] = "%s -c 'import time; time.sleep(600)'" % sys.executable
    try:
        t0 = time.time()
        op = bench.one_process_child(8, ["--scaling", "strong"], timeout=2)
        waited = time.time() - t0
    finally:
        del os.environ["PYSPEEDY_AMD_BENCH_ONE_PROCESS_CMD"]
    assert set(op) == {"error"} and "no answer within 2 s" in op["error"], op
    assert 1.9 <= waited < 15.0
    code, out, err = bench.run_bounded_child([sys.executable, "-c", "print('hello'); import sys; sys.exit(3)"], dict(os.environ), 30)
    assert code == 3 and out.strip() == "hello"


PREFLIGHT_CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
ok, why, spent = bench.rccl_preflight(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), bench.Budget(120.0))
print(json.dumps({"ok": ok, "why": why, "spent": spent}))
"""


def test_rccl_preflight_reaches_one_decision_for_all_ranks():
    """The pre-flight's meeting point without a GPU: every rank's probe child fails here (no device), rank 0 collects the verdicts
    and writes the decision, the other ranks read it -- all of them come back with the same answer (gloo) and the same reason."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("needs a machine without a GPU (on a GPU box the GPU tier runs the real thing)")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, "-c", PREFLIGHT_CHILD, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(3)]
    outs = [json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in procs]
    assert [o["ok"] for o in outs] == [False] * 3
    assert len({o["why"] for o in outs}) == 1 and "rank 0:" in outs[0]["why"] and "rank 2:" in outs[0]["why"]
    assert all(o["spent"] < 120 for o in outs)


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
    assert strong["vs_baseline"] is None and "cpu_baseline" not in strong and "cfg4_strong" not in strong
    assert strong["collective"]["ranks_seen"] == 2 and strong["collective"]["boundary_checksum_equal"] is True
    assert strong["one_process"]["containers"] == 6 and strong["one_process"]["device_models"] == 1
    assert "cfg4_strong" not in strong["one_process"]
    weak = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--members", "4"] + common, capture_output=True,
                                  text=True, timeout=900, env=env))
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak"
    assert weak["config"]["members_per_gpu"] == 4 and weak["config"]["members_total"] == 8
    assert weak["config"]["plan"].startswith("serial")  # 4 members per GPU: one group
    years = 8 * 86400.0 / (weak["ms_per_step"] * 1e-3 * 13140)
    assert abs(weak["value"] - years) < 1e-6 * years
    names = {k["kernel"] for k in weak["roofline"]["kernels"]}
    assert {"spec2grid", "column_sw", "column", "grid2spec", "spectral_step"} <= names
    assert weak["roofline"]["serial_plan_ms_per_step"] > 0


@pytest.mark.gpu
def test_a_hanging_one_process_child_does_not_cost_the_line():
    """First contact with a second GPU happens in the SECONDARY legs of an N-rank line (one process over all devices: peer
    copies, RCCL inside the library).  Here the child that measures it never answers: the 2-rank line still comes out, inside
    its wall-clock budget, with the reason in `one_process.error`, and rank 1 has waited no longer than rank 0's timeouts."""
    import time
    env = dict(os.environ, PYSPEEDY_AMD_BENCH_BACKEND="gloo",
               PYSPEEDY_AMD_BENCH_ONE_PROCESS_CMD="%s -c 'import time; time.sleep(3600)'" % sys.executable)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    res = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--members", "4", "--steps", "6", "--warmup", "3", "--regions", "2",
                                  "--no-cpu-baseline", "--budget", "90"], capture_output=True, text=True, timeout=600, env=env))
    wall = time.time() - t0
    assert wall < 90 + 30, wall
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["roofline"]["frac"] > 0
    op = res["one_process"]
    assert "no answer within" in op["error"] and op["timeouts_s"][0] <= 30.0 and op["timeouts_s"][1] == 0.0
    assert res["budget"]["budget_s"] == 90.0 and res["budget"]["used_s"] < 90.0
    # ... and with no time left at all nothing secondary is started: the headline alone
    res = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "3", "--regions", "2",
                                  "--no-cpu-baseline", "--budget", "1"], capture_output=True, text=True, timeout=600, env=env))
    assert res["value"] > 0 and res["one_process"] == {"skipped": "budget"} and res["cfg4_strong"] == {"skipped": "budget"}
    assert {k["leg"] for k in res["budget"]["skipped"]} == {"cfg4_strong", "one_process", "stream_ceiling"}
    assert res["roofline"]["stream_ceiling"] == {"skipped": "budget"}


@pytest.mark.gpu
def test_rccl_preflight_falls_back_to_gloo_when_rccl_cannot_form_the_world():
    """Before the ranks commit to RCCL each of them forms the same RCCL world in a bounded child process; when that fails on any
    rank, rank 0 decides for all and the line is produced over gloo with the reason in it.  Two ranks on the ONE GPU of this box
    are a world RCCL refuses (two ranks on one device): exactly that case, with the backend left at its default."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "PYSPEEDY_AMD_BENCH_BACKEND"):
        env.pop(k, None)
    res = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--members", "4", "--steps", "6", "--warmup", "3", "--regions", "2",
                                  "--no-cpu-baseline", "--no-legs"], capture_output=True, text=True, timeout=900, env=env))
    c = res["collective"]
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["config"]["backend"] == "gloo" and c["backend"] == "gloo"
    assert c["rccl_preflight"]["ok"] is False and c["rccl_preflight"]["why"] and "pre-flight" in c["backend_fallback"]
    assert c["ranks_seen"] == 2 and c["boundary_checksum_equal"] is True


@pytest.mark.gpu
def test_the_n_rank_line_is_complete():
    """The driver's own command at N > 1 (default members: the weak headline, 64 per GPU) -- here with 2 ranks sharing the one
    GPU: the line carries the host baseline with its core count (measured by the launcher before any rank exists), the ratio to
    it, and BASELINE cfg 4 to the letter (64 members in total, sharded) beside the weak headline."""
    env = dict(os.environ, PYSPEEDY_AMD_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    res = _result(subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "3", "--regions", "2",
                                  "--cpu-seconds", "1.5"], capture_output=True, text=True, timeout=1200, env=env))
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["members_total"] == 128
    assert res["config"]["plan"].startswith("2 member groups")
    cores = res["cpu_baseline"]["all_cores"]["cores"]
    assert cores >= 1 and res["cpu_baseline"]["cores"] == 1 and res["cpu_baseline"]["kind"] in ("reference", "port")
    assert abs(res["vs_baseline"] - res["value"] / res["cpu_baseline"]["all_cores"]["value"]) < 1e-9 * res["vs_baseline"]
    strong = res["cfg4_strong"]
    assert strong["members_total"] == 64 and strong["members_per_gpu"] == 32 and strong["scaling"] == "strong"
    assert abs(strong["value"] - 64 * 86400.0 / (strong["ms_per_step"] * 1e-3 * 13140)) < 1e-6 * strong["value"]
    assert strong["vs_cpu_all_cores"] > 0
    assert "drop_in_step" not in res  # (a leg of the one-GPU line)
    # the line proves what the collective layer saw: two ranks, the device of each, the same boundary fields on both
    c = res["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["ranks_seen"] == 2 and c["device_of_rank"] == [0, 0]
    assert c["boundary_checksum_equal"] is True and len(set(c["boundary_checksum_of_rank"])) == 1 and c["boundary_bytes"] > 3.0e6
    assert [g["rank"] for g in c["gpu_of_rank"]] == [0, 1] and c["distinct_gpus"] == 1  # (the rehearsal shares one card)
    # ... and carries the reference's own shape, one process over all devices, measured by rank 0 while rank 1 idles
    op = res["one_process"]
    assert "error" not in op, op
    assert op["processes"] == 1 and op["devices_asked"] == 2 and op["devices_used"] == 1 and op["containers"] == 128
    assert op["device_models"] == 2 and op["members_per_device"] == [128] and op["current_device_preserved"] is True
    assert op["boundary_broadcast"] == {"collective_devices": 0, "peer_copies": 0, "local_copies": 127,
                                        "transport": "local copies only", "arrived_intact": True,
                                        "note": "local copies only (one device)"}
    assert 0 < op["begin_end_ms_per_step"] == op["ms_per_step"] and 0 < op["sync_ms_per_step"]
    assert abs(op["value"] - 128 * 86400.0 / (op["ms_per_step"] * 1e-3 * 13140)) < 1e-6 * op["value"]
    assert op["cfg4_strong"]["containers"] == 64 and op["cfg4_strong"]["device_models"] == 2


@pytest.mark.gpu
def test_the_one_gpu_line_carries_the_drop_in_and_every_store_legs():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "PYSPEEDY_AMD_BENCH_BACKEND"):
        env.pop(k, None)
    res = _result(subprocess.run([sys.executable, BENCH, "--members", "8", "--steps", "6", "--warmup", "3", "--regions", "2",
                                  "--cpu-seconds", "1.5", "--leg-seconds", "0.05"], capture_output=True, text=True, timeout=1200,
                                 env=env))
    assert res["n_gpus"] == 1 and res["config"]["members_total"] == 8
    d = res["drop_in_step"]
    assert d["containers"] == 8 and 0 < d["begin_end_ms_per_step"] and 0 < d["sync_ms_per_step"]
    assert d["containers_1"]["containers"] == 1 and 0 < d["containers_1"]["begin_end_ms_per_step"] < d["containers_1"]["sync_ms_per_step"]
    assert d["containers_8"]["containers"] == 8 and d["containers_8"]["device_models"] == 1
    # the reference's own entry points on the same clock, and the step() contract's figures flat in `config`
    f = res["facade_run"]
    for key, members in (("ens64", 64), ("single", 1)):
        assert f[key]["members"] == members and 0 < f[key]["run_ms_per_step"] and 0 < f[key]["run_daily_export_ms_per_step"]
        assert f[key]["files_written"] == f[key]["run_daily_export_steps"] // 36 and f[key]["megabytes_written"] > 0.7 * members * f[key]["files_written"]
    assert f["ens64"]["run_ms_per_step"] < f["ens64"]["run_daily_export_ms_per_step"]  # (48 MB per simulated day)
    # ... and a large ensemble through the facade: its stretches are multi-step calls, so the members go in rounds of 64
    assert f["ens256"]["members"] == 256 and abs(f["ens256"]["us_per_member_step"] - f["ens256"]["run_ms_per_step"] / 256 * 1e3) < 1e-9
    assert f["ens256"]["us_per_member_step"] < 1.15 * f["ens64"]["run_ms_per_step"] / 64 * 1e3
    cfg = res["config"]
    assert cfg["step_contract_ms_per_step_sync_8"] == d["sync_ms_per_step"] and cfg["step_contract_ms_per_step_begin_end_8"] == d["begin_end_ms_per_step"]
    assert cfg["step_contract_ms_per_step_sync_1"] == d["containers_1"]["sync_ms_per_step"]
    assert cfg["facade_ens64_run_ms_per_step"] == f["ens64"]["run_ms_per_step"]
    assert cfg["facade_single_run_daily_export_ms_per_step"] == f["single"]["run_daily_export_ms_per_step"]
    pr = res["projected_8gpu_cfg4"]
    assert pr["projection"] is True and pr["n_gpus"] == 8 and pr["ms_per_step"] == res["cfg4_shard8"]["ms_per_step"]
    assert abs(pr["value"] - 64 * 86400.0 / (pr["ms_per_step"] * 1e-3 * 13140)) < 1e-6 * pr["value"]
    assert abs(pr["efficiency"] * 8 - pr["speedup_over_1gpu"]) < 1e-9 and cfg["projected_8gpu_cfg4_value"] == pr["value"]
    assert res["budget"]["skipped"] == [] and res["budget"]["used_s"] < res["budget"]["budget_s"] == 300.0
    assert res["cpu_baseline"]["all_cores_cores"] == res["cpu_baseline"]["all_cores"]["cores"]
    e = res["every_step_stores"]
    assert e["spec2grid_per_member"] == 91 and e["ms_per_step"] > 0
    dom = res["roofline"]["dominant"]  # the fused column kernel: the largest share of the step, priced like the line's kernel
    assert dom["share_of_kernel_time"] > 0.25 and 0 < dom["frac"] < 1 and dom["traffic"] > 0
    assert abs(dom["achieved"] - dom["algorithmic_bytes_per_launch"] / (dom["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * dom["achieved"]
    assert res["cpu_baseline"]["all_cores"]["cores"] >= 1 and res["vs_baseline"] > 0
    # the ceiling the fractions of 8 TB/s are to be read against, measured with the library's own probe kernels on this box
    sc = res["roofline"]["stream_ceiling"]
    for mix in ("copy_1r1w", "column_2r1w", "mix_3r2w", "read", "write"):
        assert 2.0 < sc[mix] < 9.0 and 2.0 < sc[mix + "_column_shape"] <= sc[mix] + 1e-9 and sc[mix + "_shape"], (mix, sc)
    twin = sc["column_twin"]  # the column kernel's launch with the arithmetic taken out: the kernel cannot be faster than that
    assert twin["us"] > 0 and twin["column_kernel_over_twin"] > 0.9 and twin["bytes"] > 0
    assert all("frac_of_stream_ceiling" in k for k in res["roofline"]["kernels"] if k["kernel"] in ("spec2grid", "column", "grid2spec"))
    assert 0 < dom["frac_of_stream_ceiling"] < dom["frac_of_stream_ceiling_in_its_shape"] and dom["frac"] < dom["frac_of_stream_ceiling"]
    assert res["roofline"]["frac_of_stream_ceiling"] > res["roofline"]["frac"]
    # the committed PMC traffic is tied to the device sources it was taken with
    sys.path.insert(0, ROOT)
    import bench
    assert res["roofline"]["kernel_sources_sha"] == bench.kernel_sources_sha() and res["roofline"]["traffic_stale"] in (True, False, None)
    assert dom["traffic_stale"] in (True, False, None)
    # the line's kernel where nothing of its traffic can sit in the Infinity Cache (256 members)
    b = res["roofline"]["beyond_infinity_cache"]
    assert b["members"] == 256 and 0 < res["roofline"]["frac_beyond_infinity_cache"] == b["frac"] < 1
    # every BASELINE config on the same clock (SURVEY 8d "Configs as concrete inputs")
    for key, members in (("cfg3", 1), ("cfg4_shard8", 8), ("cfg5", 32)):
        leg = res[key]
        assert leg["members"] == members and leg["ms_per_step"] > 0 and leg["steps_per_region"] == 360 and leg["regions"] >= 1
        assert abs(leg["us_per_member_step"] - leg["ms_per_step"] * 1e3 / members) < 1e-9
        assert abs(leg["value"] - members * 86400.0 / (leg["ms_per_step"] * 1e-3 * 13140)) < 1e-6 * leg["value"]
        r = leg["step_roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
        assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (leg["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        assert {"spec2grid", "column_sw", "column", "grid2spec", "spectral_step"} <= set(leg["kernel_us"])
        assert set(leg["kernel_frac"]) == set(leg["kernel_us"]) and leg["plan"]
    # a member-step moves ~20 MB: the legs' byte counts must scale with the members
    per_member = [res[k]["step_roofline"]["algorithmic_bytes_per_step"] / res[k]["members"] for k in ("cfg3", "cfg4_shard8", "cfg5")]
    assert all(1.8e7 < b < 2.4e7 for b in per_member), per_member
    assert "sppt" in res["cfg5"]["workload"].lower() and res["cfg5"]["plan"].startswith("3 member groups")
    t = res["cfg2_transforms"]
    assert t["algorithmic_bytes_per_field"] == 52736
    assert sorted({r["fields"] for r in t["rows"]}) == [1, 8, 64, 512, 4096, 16384]
    assert {r["kernel"] for r in t["rows"]} == {"spec2grid", "grid2spec", "legendre_inv", "legendre"} and len(t["rows"]) == 16
    assert sorted(r["fields"] for r in t["rows"] if r["kernel"] == "legendre") == [4096, 16384]  # the Legendre stage on its own
    on_device = {r["kernel"]: r["ns_per_field"] for r in t["rows"] if r["fields"] == 4096}
    for k in ("spec2grid", "grid2spec"):  # cfg 2 as worded (state on the host): the same kernels through PCIe, far slower
        assert t["pcie_inclusive"][k]["fields"] == 4096 and t["pcie_inclusive"][k]["ns_per_field"] > 5 * on_device[k]
    for r in t["rows"]:
        want = 39680 if r["kernel"].startswith("legendre") else 52736
        assert r["algorithmic_bytes_per_field"] == want
        assert r["ns_per_field"] > 0 and abs(r["frac"] - want / r["ns_per_field"] / 8000.0) < 1e-6


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
    c = res["collective"]  # gathered through RCCL, on device buffers
    assert c["backend"] == "nccl" and c["ranks_seen"] == 1 and c["device_of_rank"] == [0] and c["boundary_checksum_equal"] is True


@pytest.mark.gpu
def test_bench_one_process_mode():
    """`--one-process`: no ranks, this process drives the devices through spd_parallel_step -- rehearsed on the one GPU of the
    box, where 40 containers are kept as two device models (both enqueued before either is waited for)."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PYSPEEDY_AMD_BENCH_BACKEND"):
        env.pop(k, None)
    res = _result(subprocess.run([sys.executable, BENCH, "--one-process", "--gpus", "1", "--members", "40", "--steps", "30",
                                  "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env))
    op = res["one_process"]
    assert res["n_gpus"] == 1 and res["value"] == op["value"] and res["ms_per_step"] == op["begin_end_ms_per_step"]
    assert op["containers"] == 40 and op["device_models"] == 2 and op["devices_used"] == 1 and op["steps_timed"] == 30
    assert op["boundary_broadcast"] == {"collective_devices": 0, "peer_copies": 0, "local_copies": 39,
                                        "transport": "local copies only", "arrived_intact": True,
                                        "note": "local copies only (one device)"}
    assert op["current_device_preserved"] is True and res["vs_baseline"] is None


@pytest.mark.gpu
def test_bench_under_torchrun_measures_the_host_baseline_on_rank_0():
    """The driver's launch line for N > 1: every process is a rank; rank 0 measures the host baseline before it touches the GPU
    while the other ranks wait in the rendezvous, and the line is as complete as the launcher's."""
    sys.path.insert(0, ROOT)
    import bench
    env = dict(os.environ, PYSPEEDY_AMD_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(bench.free_port()), BENCH, "--gpus", "2", "--members", "4", "--steps", "6", "--warmup", "3",
           "--regions", "2", "--cpu-seconds", "1.5"]
    res = _result(subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env))
    assert res["n_gpus"] == 2 and res["config"]["members_total"] == 8
    assert res["cpu_baseline"]["all_cores"]["cores"] >= 1 and res["vs_baseline"] > 0
    # (the streaming ceiling is rank 0's; the 256-member leg belongs to the one-GPU line)
    sc = res["roofline"]["stream_ceiling"]
    assert 2.0 < sc["copy_1r1w"] < 9.0 and sc["column_twin"]["us"] > 0
    assert res["roofline"]["frac_beyond_infinity_cache"] is None and res["roofline"]["traffic_stale"] in (True, False, None)
    # ... and what tools/check_scale.py reads travels flat in `config`
    cfg = res["config"]
    assert cfg["collective_ranks_seen"] == 2 and cfg["collective_backend"] == "gloo" and cfg["collective_boundary_checksum_equal"] is True


def _line(n, value, **config):
    """a bench line as tools/check_scale.py reads it: the contract's scalars and `config` only (what the driver's records keep)"""
    return {"metric": "simulated-years/day (whole node), T30L8", "value": value, "n_gpus": n, "scaling": "weak",
            "config": dict({"group_streams_side_by_side_by_rank": [True] * n}, **config)}


def _good(n, value):
    return _line(n, value, collective_backend="nccl", collective_ranks_seen=n, collective_distinct_gpus=n,
                 collective_boundary_checksum_equal=True, collective_rccl_preflight_ok=True, collective_backend_fallback=None,
                 one_process_boundary_broadcast_note="one RCCL broadcast to %d other device(s)" % (n - 1))


def test_check_scale_reads_the_first_multi_gpu_record_as_design_says(tmp_path, capsys):
    """tools/check_scale.py: the reading rules of the scaling record as code.  On the committed 4-rank rehearsal (four ranks on
    ONE GPU over gloo) it reports exactly the deviations such a rehearsal must show; on a record as the design expects it --
    weak scaling flat, cfg 4 as worded at its projection, RCCL everywhere -- none; and it names what is off when something is."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_scale
    rehearsal = os.path.join(ROOT, "profiles", "r05_bench_rehearsal_4ranks_one_gpu.json")
    assert check_scale.main(["--line", rehearsal]) == 1
    out = capsys.readouterr().out
    deviations = [ln for ln in out.splitlines() if ln.startswith("DEVIATION")]
    assert len(deviations) == 3 and "3 deviation(s)" in out
    assert "distinct GPUs: 1 of 4" in deviations[0] and "backend 'gloo'" in deviations[1] and "local copies only" in deviations[2]
    assert "ok         N=4 ranks that met in the collective layer: 4 of 4" in out and "member groups side by side on every rank" in out
    one = _line(1, 1.70e6, projected_8gpu_cfg4_value=7.4e6)
    runs = [{"n": 1, "parsed": one}] + [{"n": n, "parsed": _good(n, 1.70e6 * n * f)} for n, f in ((2, 0.99), (4, 1.01), (8, 0.98))]
    runs[3]["parsed"]["config"]["cfg4_strong_value"] = 7.2e6
    scale, bench_file = tmp_path / "SCALE.json", tmp_path / "BENCH.json"
    scale.write_text(json.dumps({"runs": runs}))
    bench_file.write_text(json.dumps({"n": 1, "parsed": _line(1, 1.66e6)}))
    assert check_scale.main([str(scale), str(bench_file)]) == 0
    out = capsys.readouterr().out
    assert "0 deviation(s)" in out and "4.24 x" in out and "N=1 of the scaling record against BENCH" in out
    # ... a node where RCCL fell back at 8 ranks, two ranks shared a GPU and the weak curve sags
    bad = json.loads(scale.read_text())
    cfg8 = bad["runs"][3]["parsed"]["config"]
    cfg8.update(collective_backend="gloo", collective_rccl_preflight_ok=False, collective_backend_fallback="RCCL did not pass", collective_distinct_gpus=7)
    bad["runs"][3]["parsed"]["value"] = 1.70e6 * 8 * 0.90
    cfg8["one_process_boundary_broadcast_note"] = "peer copies, because: RCCL not loadable"
    scale.write_text(json.dumps(bad))
    assert check_scale.main([str(scale), str(bench_file)]) == 1
    out = capsys.readouterr().out
    assert "DEVIATION  weak scaling at N=8 flat" in out and "DEVIATION  N=8 distinct GPUs: 7 of 8" in out
    assert "DEVIATION  N=8 RCCL pre-flight passed" in out and "DEVIATION  N=8 one process: ONE RCCL broadcast" in out
    assert check_scale.main([os.path.join(ROOT, "SCALE_r05.json")]) == 0  # (a skipped record is said to be one)


def test_scaling_facts_travel_flat_in_config_and_the_traffic_is_tied_to_the_sources(tmp_path, monkeypatch):
    """What check_scale reads is repeated as flat scalars in `config` (the driver's records keep the contract's objects only), the
    host-side meeting points of the ranks carry a nonce per launch attempt, and the committed PMC traffic says when it was taken
    with other device sources than the tree's."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    flat = bench.scaling_keys({"backend": "nccl", "ranks_seen": 8, "distinct_gpus": 8, "boundary_checksum_equal": True,
                               "rccl_preflight": {"ok": True, "seconds": 3.0}},
                              {"cfg4_strong": {"value": 7.1e6, "ms_per_step": 0.058, "members_per_gpu": 8},
                               "one_process": {"ms_per_step": 0.3, "value": 1.0e7, "devices_used": 8, "containers": 512,
                                               "boundary_broadcast": {"note": "one RCCL broadcast to 7 other device(s)"},
                                               "cfg4_strong": {"value": 6.9e6}}})
    assert flat["collective_ranks_seen"] == 8 and flat["collective_rccl_preflight_ok"] is True and flat["collective_backend_fallback"] is None
    assert flat["cfg4_strong_value"] == 7.1e6 and flat["one_process_cfg4_strong_value"] == 6.9e6
    assert flat["one_process_boundary_broadcast_note"].startswith("one RCCL broadcast to 7")
    assert all(not isinstance(v, (dict, list)) for v in flat.values())
    # the meeting points of one launch attempt
    monkeypatch.setenv("MASTER_PORT", "29123")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    first = bench.job_file("rccl")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert bench.job_file("rccl") != first and "29123" in first and str(os.getppid()) in first
    # the stamp of the PMC files
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    kernels = {"spd::spec2grid_table_kernel": {"hbm_bytes_per_launch": 4928 * 50000, "fields_per_launch": 4928}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_sha", lambda: sha)
    for stamp, want in ((sha, False), ("0" * 16, True), (None, None)):
        doc = {"kernels": kernels}
        if stamp:
            doc["kernel_sources_sha"] = stamp
        (prof / bench.PMC_FILES[0]).write_text(json.dumps(doc))
        traffic, source, stale = bench.load_traffic(4928)
        assert traffic == 4928 * 50000 and bench.PMC_FILES[0] in source and stale is want
