"""The figures DESIGN.md / README.md quote, recomputed from the files of a collection (tools/collect_profiles.sh <tag>):

    python tools/design_figures.py [directory with <tag>_*.json / .csv, default profiles] [tag, default r05]
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
where = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
load = lambda name: json.load(open(os.path.join(where, "%s_%s.json" % (tag, name))))  # noqa: E731

d, d20, ds = load("bench"), load("bench_20"), load("bench_serial_plan")
pmc = load("pmc_model_step")["kernels"]
stats = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(where, tag + "_model_bench_kernel_stats.csv")))}
rows = (("spec2grid_table_kernel", 259.883), ("physics_kernel<2, true, false, double, false>", 492.5), ("grid2spec_table_kernel", 246.383),
        ("spectral_step_kernel<false, false, spd::CouplerArgs>", 288.5), ("geopotential_kernel<false>", 17.27))
print("kernel (serial plan, 64 members)                       avg us   algorithmic MB   PMC MB   TB/s   frac of 8 TB/s")
for key, mb in rows:
    us = [v for k, v in stats.items() if key in k][0]
    p = [v["hbm_bytes_per_launch"] / 1e6 for k, v in pmc.items() if key in k][0]
    print("  %-52s %7.2f  %10.1f  %10.1f  %6.2f  %6.3f" % (key[:52], us, mb, p, mb / us, mb / us / 8))
c = d["config"]
print("headline %.4f ms/step = %.3f M sim-years/day, %.0f x all %d host cores (one core %.2f ms per member-step); 20-step command %.4f; "
      "serial plan %.4f; every_step_stores %.4f" % (d["ms_per_step"], d["value"] / 1e6, d["vs_baseline"], d["cpu_baseline"]["all_cores"]["cores"],
                                                      d["cpu_baseline"]["ms_per_member_step"], d20["ms_per_step"], ds["ms_per_step"],
                                                      d["every_step_stores"]["ms_per_step"]))
print("step() contract sync / begin-end: 64: %.4f / %.4f; 8: %.4f / %.4f; 1: %.4f / %.4f" % tuple(
    c["step_contract_ms_per_step_%s_%d" % (k, n)] for n in (64, 8, 1) for k in ("sync", "begin_end")))
print("facade: SpeedyEns(64).run %.4f, with daily files %.4f; Speedy().run %.4f / %.4f" % (
    c["facade_ens64_run_ms_per_step"], c["facade_ens64_run_daily_export_ms_per_step"], c["facade_single_run_ms_per_step"],
    c["facade_single_run_daily_export_ms_per_step"]))
print("cfg3 %.4f (step frac %.3f)  cfg4_shard8 %.4f (%.3f)  cfg5 %.4f (%.3f)" % tuple(
    v for k in ("cfg3", "cfg4_shard8", "cfg5") for v in (d[k]["ms_per_step"], d[k]["step_roofline"]["frac"])))
print("cfg2 at 16 384 fields: " + ", ".join("%s %.3f" % (k, [r["frac"] for r in d["cfg2_transforms"]["rows"] if r["kernel"] == k and r["fields"] == 16384][0])
                                            for k in ("spec2grid", "grid2spec", "legendre_inv", "legendre")))
p = d["projected_8gpu_cfg4"]
print("projected 8 GPUs, cfg 4: %.2f M sim-years/day, %.2f x one GPU, efficiency %.2f" % (p["value"] / 1e6, p["speedup_over_1gpu"], p["efficiency"]))
for name, label in (("model_bench_kernel_stats_8members", "8 members"), ("model_bench_kernel_stats_1member", "1 member")):
    st = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(where, "%s_%s.csv" % (tag, name))))}
    pick = lambda key: [v for k, v in st.items() if key in k and "NoCoupler" not in k][0]  # noqa: E731
    print("%s: column %.1f, grid2spec %.1f, spec2grid %.1f, spectral step %.1f us" % (
        label, pick("physics_kernel"), pick("grid2spec_table"), pick("spec2grid_table"), pick("spectral_step_kernel")))
leg = os.path.join(where, tag + "_legendre_only_kernel_stats.csv")
if os.path.exists(leg):
    for r in csv.DictReader(open(leg)):
        if "spd::" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("Legendre only: %-40s %7.2f us over %s launches  frac %.3f" % (r["Name"][:40], us, r["Calls"], 650.117 / us / 8))
