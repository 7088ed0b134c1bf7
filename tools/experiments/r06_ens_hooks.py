"""SpeedyEns(64).run() with each of the reference's hooks, daily, ten days: ms per model step."""
import os
import sys
import tempfile
import time
from datetime import datetime, timedelta

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyspeedy_amd.callbacks import DiagnosticCheck, ModelCheckpoint, XarrayExporter  # noqa: E402
from pyspeedy_amd.speedy import SpeedyEns  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
start, days = datetime(1982, 1, 1), 10


def run(make):
    ens = SpeedyEns(M, start_date=start, end_date=start + timedelta(days=days))
    ens.set_bc()
    with tempfile.TemporaryDirectory() as tmp:
        hooks = make(tmp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens.run(callbacks=hooks)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (36 * days) * 1e3


run(lambda tmp: [XarrayExporter(36, output_dir=tmp)])
for name, make in (("no hooks", lambda tmp: []), ("DiagnosticCheck(36)", lambda tmp: [DiagnosticCheck(36)]),
                   ("ModelCheckpoint(36)", lambda tmp: [ModelCheckpoint(36)]), ("ModelCheckpoint(36, device_bytes=0)", lambda tmp: [ModelCheckpoint(36, device_bytes=0)]),
                   ("XarrayExporter(36)", lambda tmp: [XarrayExporter(36, output_dir=tmp)]),
                   ("ModelCheckpoint + XarrayExporter", lambda tmp: [ModelCheckpoint(36), XarrayExporter(36, output_dir=tmp)]),
                   ("all three", lambda tmp: [DiagnosticCheck(36), ModelCheckpoint(36), XarrayExporter(36, output_dir=tmp)])):
    print("%3d members  %-38s %.4f ms per step" % (M, name, min(run(make) for _ in range(2))), flush=True)
