"""The streaming ceiling of the device, measured with the library's own probe kernels (spd_stream_probe, csrc/stream_probe.hip).

The step's kernels are priced against the contract's 8 TB/s (bench.py: `roofline`).  How much of the distance to that figure is the
kernels' own and how much is the memory system's is read against kernels that do nothing but move bytes in the same SHAPE: stream
mix (copy 1r:1w, the column kernel's 2r:1w, 3r:2w, read only, write only), bytes per lane, wavefronts per SIMD, rows in flight, how
long a wavefront lives and where its rows lie.  `probe` measures one shape, `ceiling` the compact set bench.py puts into its line
as `roofline.stream_ceiling`, `column_twin` the column kernel's own launch with the arithmetic taken out.  tools/stream_ceiling.py
prints the whole table (profiles/r06_stream_ceiling.txt).  Measurement infrastructure: no counterpart in the reference.
"""
import ctypes as C

from . import _lib

MIXES = {"copy_1r1w": (1, 1), "column_2r1w": (2, 1), "mix_3r2w": (3, 2), "read": (1, 0), "write": (0, 1)}
GB = 1e9
# (bytes per lane, wavefronts per SIMD, rows a wavefront lives, rows in flight per stream, layout) of `ceiling`
SHAPES = ((16, 8, 1, 1, 0), (8, 8, 1, 1, 0), (16, 8, 64, 4, 0), (8, 8, 64, 8, 0), (8, 8, 24, 8, 1), (8, 2, 243, 8, 1),
          (8, 2, 243, 16, 1), (8, 8, 243, 8, 1))
COLUMN_SHAPE = (8, 2, 243, 8, 1)  # one double per lane, two wavefronts per SIMD, 243 rows per wavefront, row r in array r


def probe(L, handle, reads, writes, total_bytes, lane_bytes=8, in_flight=8, nontemporal=1, waves_per_simd=2, rows_per_wave=243,
          reps=10, layout=0):
    """One shape -> {"tb_s": mean, "tb_s_best": best launch, "us": mean microseconds, "bytes": moved per launch}."""
    a = _lib.StreamProbeArgs(reads, writes, lane_bytes, in_flight, nontemporal, waves_per_simd, rows_per_wave, reps, layout, 0,
                             int(total_bytes))
    mean, best, moved, wgs = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
    _lib.check(L.spd_stream_probe(handle, C.byref(a), C.byref(mean), C.byref(best), C.byref(moved), C.byref(wgs)),
               "spd_stream_probe")
    return {"tb_s": moved.value / mean.value / 1e6, "tb_s_best": moved.value / best.value / 1e6, "us": mean.value,
            "us_best": best.value, "bytes": moved.value, "workgroups": wgs.value}


def shape_key(lane_bytes, waves, rows, in_flight, layout, nontemporal):
    return "%dB_%dw_%drows_%dfl_%s_%s" % (lane_bytes, waves, rows, in_flight, "arrays" if layout else "chunk", "nt" if nontemporal else "plain")


def ceiling(L, handle, total_bytes=1.3 * GB, reps=8):
    """The compact form for the bench line: for every stream mix the best shape at the step's size (1.3 GB: what the 64-member step
    moves) and the column kernel's own shape with the non-temporal hint, in TB/s."""
    out = {"bytes_per_launch": int(total_bytes), "unit": "TB/s",
           "note": "own HIP kernels (spd_stream_probe), mean of %d launches timed by their dispatch packets; *_column_shape: one "
                   "double per lane, two wavefronts per SIMD, 243 rows per wavefront each in an array of its own, non-temporal" % reps}
    for name, (r, w) in MIXES.items():
        best = None
        for lane_bytes, waves, rows, fl, lay in SHAPES:
            for nt in (0, 1):
                res = probe(L, handle, r, w, total_bytes, lane_bytes, fl, nt, waves, rows, reps, lay)
                if best is None or res["tb_s"] > best[1]:
                    best = (shape_key(lane_bytes, waves, rows, fl, lay, nt), res["tb_s"])
                if (lane_bytes, waves, rows, fl, lay) == COLUMN_SHAPE and nt == 1:
                    out[name + "_column_shape"] = round(res["tb_s"], 3)
        out[name] = round(best[1], 3)
        out[name + "_shape"] = best[0]
    return out


def column_twin(L, handle, nbytes, reps=10):
    """The column kernel's launch with the arithmetic taken out: 2 reads : 1 write, one double per lane, two wavefronts per SIMD,
    243 rows per wavefront each in an array of its own, the non-temporal hint, and exactly `nbytes` (the kernel's algorithmic bytes
    per launch).  What the kernel takes beyond this is its own -- dependent arithmetic, its phases' round trips -- not the memory
    system's.  -> {"us", "us_best", "tb_s", "bytes", "workgroups"}"""
    lb, wv, rw, fl, lay = COLUMN_SHAPE
    return probe(L, handle, 2, 1, nbytes, lb, fl, 1, wv, rw, reps, lay)
