"""The streaming ceiling of the box, measured with the library's own probe kernels (pyspeedy_amd/stream_probe.py, spd_stream_probe).

    python tools/stream_ceiling.py [--quick] [--json FILE]

Prints TB/s (mean over the timed launches; the best launch in brackets) for every shape the step's kernels come in: stream mix
(copy 1:1, the column kernel's 2r:1w, 3r:2w, read only, write only), bytes moved per launch (0.25 / 1.3 / 4 GB: inside the 256 MB
Infinity Cache, the 64-member step, far beyond), 8 or 16 bytes per lane, non-temporal hint off / on, wavefronts per SIMD (two: the
column kernel's occupancy at 256 VGPRs), rows in flight per lane, a wavefront's life (one batch and out, or the column kernel's 243
rows) and where its rows lie (a contiguous chunk of its own, or every row in an array of its own as in the column kernel); then the
column kernel's twin at its own size, and the compact object bench.py puts into its line as roofline.stream_ceiling.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from pyspeedy_amd import _lib
    from pyspeedy_amd.stream_probe import GB, MIXES, ceiling, column_twin, probe
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.spd_create(C.byref(h), 0), "spd_create")
    sizes = (1.3,) if args.quick else (0.25, 1.3, 4.0)
    rows = []
    print("spd_stream_probe on device 0: TB/s mean over 10 launches [best launch]; one wavefront per 64-thread workgroup")
    print("columns: bytes per lane / wavefronts per SIMD / rows a wavefront lives (all streams) / rows in flight per stream / layout")
    print("(layout c: a wavefront walks a contiguous chunk of its own; layout a: the column kernel's -- row r of every wavefront in")
    print(" array r, consecutive wavefronts side by side); each cell: plain | non-temporal")
    shapes = [(16, 8, 1, 1, 0), (8, 8, 1, 1, 0), (16, 8, 64, 4, 0), (8, 8, 64, 8, 0), (8, 2, 243, 8, 0), (16, 2, 243, 8, 0),
              (8, 2, 243, 4, 1), (8, 2, 243, 8, 1), (8, 2, 243, 16, 1), (8, 1, 243, 16, 1), (8, 4, 243, 8, 1), (8, 8, 243, 8, 1),
              (8, 8, 24, 8, 1), (16, 2, 243, 8, 1)]
    for gb in sizes:
        print("\n== %.2f GB per launch" % gb)
        print("%-12s" % "mix" + "".join("%24s" % ("%dB/%dw/%dr/%df/%s" % (s[:4] + ("ca"[s[4]],))) for s in shapes))
        for name, (r, w) in MIXES.items():
            line = "%-12s" % name
            for (lb, wv, rw, fl, lay) in shapes:
                cell = []
                for nt in (0, 1):
                    res = probe(L, h, r, w, gb * GB, lb, fl, nt, wv, rw, 10, lay)
                    rows.append({"mix": name, "gb": gb, "lane_bytes": lb, "waves_per_simd": wv, "rows_per_wave": rw, "in_flight": fl,
                                 "layout": lay, "nontemporal": nt, **res})
                    cell.append("%.2f[%.2f]" % (res["tb_s"], res["tb_s_best"]))
                line += "%24s" % " | ".join(cell)
            print(line, flush=True)
    print("\nthe column kernel's twin (2r:1w, 8 B per lane, two wavefronts per SIMD, 243 rows per wavefront in arrays of their own, non-temporal)")
    print("at the kernel's own algorithmic bytes per launch, 64 / 32 / 8 / 1 members (7.695 MB per member, DESIGN 4):")
    twins = {}
    for members in (64, 32, 8, 1):
        t = column_twin(L, h, 7.695e6 * members)
        twins[members] = t
        print("   %3d members: %7.1f us mean, %7.1f us best launch, %.2f TB/s, %d wavefronts" % (members, t["us"], t["us_best"], t["tb_s"], t["workgroups"]))
    print("\ncompact (bench.py roofline.stream_ceiling):")
    c = ceiling(L, h)
    print(json.dumps(c))
    if args.json:
        json.dump({"rows": rows, "ceiling": c, "column_twin": {str(k): v for k, v in twins.items()}}, open(args.json, "w"), indent=1)
    L.spd_destroy(h)


if __name__ == "__main__":
    main()
