"""Pretty-print bench.py JSON lines.  python tools/show_bench.py file.log ..."""
import json
import sys

for f in sys.argv[1:]:
    for ln in open(f):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        c = d["config"]
        print("%s: n_gpus %d M/gpu %s total %d  ms/step %.4f (min %.4f, %d regions)  value %.0f  %s  vs_baseline %s" % (
            f, d["n_gpus"], c.get("members_per_gpu", "-"), c["members_total"], d["ms_per_step"], d.get("ms_per_step_min", 0),
            d.get("regions", 1), d["value"], d["scaling"], ("%.1f" % d["vs_baseline"]) if d.get("vs_baseline") else "-"))
        print("   plan: %s" % c.get("plan", "?"))
        for key in ("one_process",):
            if key in d:
                o = d[key]
                if "error" in o:
                    print("   one process: FAILED %s" % o["error"])
                else:
                    print("   one process over %d device(s), %d containers in %d device models: begin/end %.4f ms/step, sync %.4f; "
                          "boundary broadcast %s" % (o["devices_used"], o["containers"], o["device_models"],
                                                     o["begin_end_ms_per_step"], o["sync_ms_per_step"], o["boundary_broadcast"]))
        if "collective" in d:
            o = d["collective"]
            print("   collective: %s, ranks seen %d, devices %s, boundary checksums equal: %s" % (
                o["backend"], o["ranks_seen"], o["device_of_rank"], o["boundary_checksum_equal"]))
        if "roofline" not in d:
            continue
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
            for key in ("containers_1", "containers_8"):
                if key in o:
                    print("      %d container(s): sync %.4f ms, begin/end %.4f ms" % (o[key]["containers"], o[key]["sync_ms_per_step"],
                                                                                      o[key]["begin_end_ms_per_step"]))
        if "every_step_stores" in d:
            print("   every store of the reference restored: %.4f ms/step" % d["every_step_stores"]["ms_per_step"])
        for key in ("cfg3", "cfg4_shard8", "cfg5"):
            if key in d:
                o = d[key]
                print("   %-12s %3d members: %.4f ms/step (%.2f us per member-step), step at %.3f of the HBM peak, plan: %s" % (
                    key, o["members"], o["ms_per_step"], o["us_per_member_step"], o["step_roofline"]["frac"], o["plan"]))
                print("                kernels us: %s" % "  ".join("%s %.1f" % kv for kv in o["kernel_us"].items()))
        if "cfg2_transforms" in d:
            for name in ("spec2grid", "grid2spec", "legendre_inv", "legendre"):
                print("   cfg2 %-9s ns/field (frac): %s" % (name, "  ".join("B=%d %.1f (%.3f)" % (r["fields"], r["ns_per_field"], r["frac"])
                                                                              for r in d["cfg2_transforms"]["rows"] if r["kernel"] == name)))
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
