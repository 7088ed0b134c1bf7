"""64 members, 3 days, a file every 12 steps: the files of the default exporter (device work enqueued ahead, copy by SDMA, written by a
thread) against those of background=False (everything inside the callback): byte for byte."""
import os, sys, tempfile, hashlib
from datetime import datetime, timedelta
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyspeedy_amd.callbacks import XarrayExporter
from pyspeedy_amd.speedy import SpeedyEns
start = datetime(1982, 1, 1)
sums = {}
for background in (None, False):
    ens = SpeedyEns(64, start_date=start, end_date=start + timedelta(days=3))
    ens.set_bc()
    t = ens.members[5]["t"]; t[3, 3] *= 1.001; ens.members[5]["t"] = t
    with tempfile.TemporaryDirectory() as tmp:
        ens.run(callbacks=[XarrayExporter(output_dir=tmp, interval=12, background=background)])
        sums[background] = {n: hashlib.sha256(open(os.path.join(tmp, n), "rb").read()).hexdigest()[:16] for n in sorted(os.listdir(tmp))}
print(len(sums[None]), "files;", "identical" if sums[None] == sums[False] else "DIFFERENT", "; distinct contents:", len(set(sums[None].values())))
