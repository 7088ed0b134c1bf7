"""Pretty-print bench.py JSON lines.  python tools/show_bench.py file.log ..."""
import json
import sys

for f in sys.argv[1:]:
    for ln in open(f):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        c = d["config"]
        print("%s: n_gpus %d M/gpu %d total %d  ms/step %.4f (min %.4f, %d regions)  value %.0f  %s  vs_baseline %s" % (
            f, d["n_gpus"], c["members_per_gpu"], c["members_total"], d["ms_per_step"], d.get("ms_per_step_min", 0),
            d.get("regions", 1), d["value"], d["scaling"], ("%.1f" % d["vs_baseline"]) if d.get("vs_baseline") else "-"))
        print("   plan: %s" % c.get("plan", "?"))
        r = d["roofline"]
        print("   spec2grid: %.1f us  frac %.3f   (serial plan %.4f ms/step)" % (
            r["avg_launch_ms"] * 1e3, r["frac"], r.get("serial_plan_ms_per_step", 0.0)))
        for k in r.get("kernels") or []:
            print("   %-14s n=%3d avg %7.1f us  min %7.1f us  algo %6.1f MB  %6.0f GB/s  frac %.3f" % (
                k["kernel"], k["launches_timed"], k["avg_launch_us"], k["min_launch_us"], k["algorithmic_bytes_per_launch"] / 1e6,
                k["achieved"], k["frac"]))
        if "drop_in_step" in d:
            o = d["drop_in_step"]
            print("   drop-in spd_parallel_step per model step (%d containers, %d device models): sync %.4f ms, begin/end %.4f ms" % (
                o["containers"], o.get("device_models", 0), o["sync_ms_per_step"], o["begin_end_ms_per_step"]))
        if "every_step_stores" in d:
            print("   every store of the reference restored: %.4f ms/step" % d["every_step_stores"]["ms_per_step"])
        if "cfg4_strong" in d:
            o = d["cfg4_strong"]
            print("   cfg4 strong: %d members, %d per GPU: %.4f ms/step  value %.0f  vs all host cores %s" % (
                o["members_total"], o["members_per_gpu"], o["ms_per_step"], o["value"],
                ("%.1f" % o["vs_cpu_all_cores"]) if "vs_cpu_all_cores" in o else "-"))
        if "cpu_baseline" in d:
            b = d["cpu_baseline"]
            print("   cpu 1 core: %.1f sy/d (%.2f ms/step);" % (b["value"], b["ms_per_member_step"]), end=" ")
            if "all_cores" in b:
                a = b["all_cores"]
                print("all cores (%d): %.1f sy/d, %.2f ms/member-step/core" % (a["cores"], a["value"], a["ms_per_member_step_per_core"]))
            else:
                print()
