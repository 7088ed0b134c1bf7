"""What the GPU does around the daily output of SpeedyEns(64).run(callbacks=[XarrayExporter()]): a time line of kernels and copies
around every stretch boundary, from a rocprofv3 kernel + memory-copy trace (add --hip-runtime-trace for the host's calls -- but the
host then falls behind the device, and the picture is no longer that of an untraced run).

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/boundary -- python3 tools/experiments/r06_export_boundary.py run
    python3 tools/experiments/r06_export_boundary.py read gpurun_out/boundary
"""
import csv
import glob
import os
import sys
import tempfile
from datetime import datetime, timedelta

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    from pyspeedy_amd import callbacks as CB
    from pyspeedy_amd import speedy as SP
    for _ in range(2):
        ens = SP.SpeedyEns(64, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1) + timedelta(days=4))
        ens.set_bc()
        with tempfile.TemporaryDirectory(prefix="pyspeedy_boundary_") as tmp:
            ens.run(callbacks=[CB.XarrayExporter(output_dir=tmp)])
        del ens


def read(where):
    rows = []
    for f in glob.glob(os.path.join(where, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")))
    for f in glob.glob(os.path.join(where, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy")), ""))
    for f in glob.glob(os.path.join(where, "**", "*hip_api_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H", r["Function"] + " tid=" + r.get("Thread_Id", ""), ""))
    rows.sort()
    # a boundary: the first export kernel (its name holds "export" or "pack") after a run of step kernels
    marks = [i for i, r in enumerate(rows) if r[2] == "K" and ("pack" in r[3].lower() or "export" in r[3].lower())]
    starts = [i for n, i in enumerate(marks) if n == 0 or rows[i][0] - rows[marks[n - 1]][0] > 2_000_000]
    print("%d events, %d boundaries" % (len(rows), len(starts)))
    for i in starts[-3:]:
        t0 = rows[i][0]
        print("--- boundary at first pack kernel; times in us relative to it")
        lo = i
        while lo > 0 and t0 - rows[lo][0] < 400_000:
            lo -= 1
        hi = i
        while hi < len(rows) - 1 and rows[hi][0] - t0 < 2_500_000:
            hi += 1
        last_end = None
        squeezed, run_of = [], 0
        for r in rows[lo:hi]:  # (runs of quick launch calls of one thread: one line)
            quick = r[2] == "H" and r[1] - r[0] < 20_000 and squeezed and squeezed[-1][2] == "H" and squeezed[-1][3] == r[3]
            if quick:
                run_of += 1
                squeezed[-1] = (squeezed[-1][0], r[1], "H", r[3], "x%d" % (run_of + 1))
            else:
                run_of = 0
                squeezed.append(r)
        for r in squeezed:
            gap = "" if last_end is None else "  (idle %.1f)" % ((r[0] - last_end) / 1e3) if r[0] > last_end + 3000 else ""
            print("  %9.1f .. %9.1f  %s q=%-3s %s%s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[2], r[4], r[3], gap))
            if r[2] == "K":
                last_end = r[1] if last_end is None else max(last_end, r[1])


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else read(sys.argv[2])
