"""SpeedyEns(64).run() with and without daily files for runs of 5, 10 and 20 days: what a further simulated day costs.

    python tools/experiments/r06_export_days.py
"""
import os
import sys
import tempfile
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd.callbacks import DiagnosticCheck, XarrayExporter  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

start = datetime(1982, 1, 1)


def run(days, kind):
    ens = SpeedyEns(64, start_date=start, end_date=start + timedelta(days=days))
    ens.set_bc()
    with tempfile.TemporaryDirectory(prefix="pyspeedy_days_") as tmp:
        hooks = {"bare": [], "check": [DiagnosticCheck(interval=36)], "files": [XarrayExporter(output_dir=tmp)]}[kind]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens.run(callbacks=hooks)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3


run(2, "files")
table = {}
for kind in ("bare", "check", "files"):
    for days in (5, 10, 20):
        table[kind, days] = min(run(days, kind) for _ in range(2))
        print("%-6s %2d days: %8.2f ms  = %.4f ms per step" % (kind, days, table[kind, days], table[kind, days] / (36 * days)), flush=True)
for kind in ("bare", "check", "files"):
    per_day = (table[kind, 20] - table[kind, 5]) / 15
    print("%-6s a further simulated day: %.3f ms (%.4f ms per step); per run, beside its days: %.2f ms"
          % (kind, per_day, per_day / 36, table[kind, 5] - 5 * per_day))
print("(bare: no callbacks, one device call per ten days; check: DiagnosticCheck(36), stretches of a day; files: XarrayExporter(), 48 MB per day)")
