"""Pretty-print bench.py JSON lines.  python tools/show_bench.py file.log ..."""
import json, sys
for f in sys.argv[1:]:
    for ln in open(f):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        c = d["config"]
        print("%s: n_gpus %d M/gpu %d total %d  ms/step %.4f (min %.4f, %d regions)  value %.0f  %s" % (
            f, d["n_gpus"], c["members_per_gpu"], c["members_total"], d["ms_per_step"], d.get("ms_per_step_min", 0), d.get("regions", 1), d["value"], d["scaling"]))
        r = d["roofline"]
        print("   spec2grid: %.1f us  frac %.3f" % (r["avg_launch_ms"] * 1e3, r["frac"]))
        tot = 0
        for k in r.get("kernels") or []:
            print("   %-14s n=%3d avg %7.1f us  min %7.1f us  algo %6.1f MB  %6.0f GB/s  frac %.3f" % (
                k["kernel"], k["launches_timed"], k["avg_launch_us"], k["min_launch_us"], k["algorithmic_bytes_per_launch"] / 1e6, k["achieved"], k["frac"]))
        if "overlapped_member_groups" in d:
            o = d["overlapped_member_groups"]
            print("   two member groups on two streams: %.4f ms/step  value %.0f" % (o["ms_per_step"], o["value"]))
        if "cpu_baseline" in d:
            b = d["cpu_baseline"]
            print("   cpu 1 core: %.1f sy/d (%.2f ms/step);" % (b["value"], b["ms_per_member_step"]), end=" ")
            if "all_cores" in b:
                a = b["all_cores"]
                print("all cores (%d): %.1f sy/d, %.2f ms/member-step/core" % (a["cores"], a["value"], a["ms_per_member_step_per_core"]))
            else:
                print()
