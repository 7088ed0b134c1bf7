"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into the JSON that
bench.py reads for roofline.traffic.

    python tools/pmc_summary.py <dir with fetch/ and write/ sub-directories> <out.json> "<command that was profiled>" [skip]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is reported in KiB and, on gfx950, counts 64 B per
128-B request of a wide coalesced read (MI355X_MICROARCH.md, section HBM: double it); WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  The first `skip` launches of every kernel (initialisation, warm-up) are left out of the mean.
`kernel_sources_sha` (bench.kernel_sources_sha: sha256 over csrc/*.hip, *.hpp) ties the file to the device code it was taken with.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(path, counter):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"]),
                                               int(r["Workgroup_Size"])))
    return rows


def main():
    root, out, command = sys.argv[1], sys.argv[2], sys.argv[3]
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    fetch, write = read(os.path.join(root, "fetch"), "FETCH_SIZE"), read(os.path.join(root, "write"), "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) & set(write)):
        f, w = sorted(fetch[name]), sorted(write[name])
        if len(f) > 2 * skip:
            f, w = f[skip:], w[skip:len(f) + skip]
        fk, wk = sum(v for _, v, _, _ in f) / len(f), sum(v for _, v, _, _ in w) / len(w)
        short = name.split("(")[0].replace("void ", "")
        entry = {"launches_averaged": len(f), "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                 "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "workgroups": f[-1][2] // max(f[-1][3], 1)}
        if "table_kernel" in short:
            entry["fields_per_launch"] = entry["workgroups"]
        kernels[short] = entry
    # what these numbers were taken with: bench.py reports `traffic_stale` when the tree's device sources have changed since
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"kernel_sources_sha": bench.kernel_sources_sha(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- " + command +
                         "; mean over the launches after the first %d of each kernel; HBM bytes = (2*FETCH_SIZE + "
                         "WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md)" % skip, "kernels": kernels},
              open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-40s n=%4d  %9.1f MB/launch" % (k[:40], v["launches_averaged"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
