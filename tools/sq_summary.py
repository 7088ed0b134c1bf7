"""Per-kernel means of rocprofv3 --pmc passes (SQ / TCC counters; any number of separate passes under one directory) and the
ratios DESIGN.md quotes from them.

    python tools/sq_summary.py <dir with one sub-directory per pass> [skip] [kernel-name filter ...]

For every kernel (name up to the first '(') and counter: the mean over its launches after the first `skip` (default 8).  Derived,
where the counters are there:  of SQ_WAVE_CYCLES -- parked (SQ_WAIT_ANY: s_waitcnt / barrier), issue-stalled (SQ_WAIT_INST_ANY),
issuing (SQ_ACTIVE_INST_ANY), VALU / LDS shares;  LDS bank-conflict share of LDS-active cycles;  L2 hit rate
TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, section L2);  instructions per workgroup.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8
    filters = [a for a in sys.argv[2:] if not a.isdigit()]
    vals, groups = defaultdict(lambda: defaultdict(list)), {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if filters and not any(k in name for k in filters):
                continue
            vals[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            groups[name] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    for name in sorted(vals):
        c = {}
        for counter, rows in vals[name].items():
            rows = sorted(rows)
            rows = rows[skip:] if len(rows) > 2 * skip else rows
            c[counter] = sum(v for _, v in rows) / len(rows)
        n = max(len(v) for v in vals[name].values())
        print("%s   (%d workgroups per launch, up to %d launches seen)" % (name, groups[name], n))
        w = c.get("SQ_WAVE_CYCLES")
        if w:
            parts = [("parked in waitcnt / barrier", "SQ_WAIT_ANY"), ("issue-stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"),
                     ("VALU", "SQ_ACTIVE_INST_VALU"), ("LDS", "SQ_ACTIVE_INST_LDS"), ("LDS issue stall", "SQ_WAIT_INST_LDS")]
            print("   of wave cycles: " + ", ".join("%s %.1f %%" % (t, 100.0 * c[k] / w) for t, k in parts if k in c))
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in c:
            print("   LDS bank conflicts: %.1f %% of LDS-active cycles" % (100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0) > 0:
            print("   L2: hit rate %.1f %% (%.3g hits, %.3g misses per launch)" % (
                100.0 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), c["TCC_HIT_sum"], c["TCC_MISS_sum"]))
        insts = [(t, k) for t, k in (("valu", "SQ_INSTS_VALU"), ("lds", "SQ_INSTS_LDS"), ("salu", "SQ_INSTS_SALU"), ("smem", "SQ_INSTS_SMEM"),
                                     ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR")) if k in c]
        if insts:
            print("   wave-instructions per workgroup: " + ", ".join("%s %.0f" % (t, c[k] / groups[name]) for t, k in insts))
        rest = sorted(k for k in c if not k.startswith(("SQ_WAVE", "SQ_WAIT", "SQ_ACTIVE", "SQ_INSTS", "SQ_LDS", "TCC_HIT", "TCC_MISS")))
        if rest:
            print("   " + ", ".join("%s %.4g" % (k, c[k]) for k in rest))


if __name__ == "__main__":
    main()
