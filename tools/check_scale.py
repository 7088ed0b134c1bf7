"""Reads the first multi-GPU record the way DESIGN.md section 6 says it should be read -- as code, not prose.

    python tools/check_scale.py SCALE_rNN.json [BENCH_rNN.json]        (exit code 0: every expectation holds; 1: deviations listed)
    python tools/check_scale.py --line rehearsal.json                   (one bench.py line, e.g. a committed rehearsal)

The scaling record arrives without the builder in the room (the driver runs bench.py --gpus 1, 2, 4, 8 on a whole node at the end of
a round).  Expectations, each reported as ok / DEVIATION with the numbers:
  * N = 1 of the scaling record within 10 % of the round's BENCH record (same command, another box);
  * weak scaling flat: value(N) / N within 5 % of value(1) (the step has no collective: speedy_driver.f90.j2:71-77);
  * BASELINE cfg 4 as worded (64 members, 8 per GPU on 8 GPUs; `cfg4_strong` of the 8-rank line) against the projection of the
    1-rank line (`projected_8gpu_cfg4`: 4.4-4.5 x one GPU) within 10 %;
  * the collective layer saw what it should: ranks_seen == N, distinct_gpus == N, the RCCL pre-flight passed (backend nccl, no
    fall-back), the boundary checksums of all ranks equal;
  * the one-process shape delivered the boundary fields with ONE RCCL broadcast ("one RCCL broadcast to <N-1> other device(s)");
  * every rank's member groups ran side by side (config.group_streams_side_by_side_by_rank all true).
Records are looked for wherever the driver may have put them: a list, {"runs": [...]}, {"results": [...]}, a dict keyed by N, each
entry the bench line itself or a driver record with the line under "parsed" (flat scalars in `config`) -- and, failing that, a JSON
line inside "tail".
"""
import json
import sys


def lines_of(doc):
    """every bench.py line found in a driver document, as dicts"""
    found = []

    def visit(x):
        if isinstance(x, dict):
            if "metric" in x and "n_gpus" in x and "value" in x:
                found.append(x)
                return
            if isinstance(x.get("parsed"), dict) and "n_gpus" in x["parsed"]:
                found.append(x["parsed"])
                return
            if isinstance(x.get("tail"), str):
                for ln in x["tail"].splitlines():
                    ln = ln.strip()
                    if ln.startswith("{") and '"n_gpus"' in ln:
                        try:
                            found.append(json.loads(ln))
                            return
                        except ValueError:
                            pass
            for v in x.values():
                visit(v)
        elif isinstance(x, list):
            for v in x:
                visit(v)
    visit(doc)
    return found


def get(line, flat_key, *path):
    """a scalar: from config (flat) when it is there, else from the nested object"""
    cfg = line.get("config") or {}
    if flat_key in cfg:
        return cfg[flat_key]
    x = line
    for p in path:
        if not isinstance(x, dict) or p not in x:
            return None
        x = x[p]
    return x


class Report:
    def __init__(self):
        self.rows = []

    def expect(self, ok, what, detail):
        self.rows.append((bool(ok), what, detail))

    def skip(self, what, why):
        self.rows.append((None, what, why))

    def deviations(self):
        return [r for r in self.rows if r[0] is False]

    def show(self, out=sys.stdout):
        for ok, what, detail in self.rows:
            out.write("%-10s %s: %s\n" % ("ok" if ok else ("--" if ok is None else "DEVIATION"), what, detail))


def check_line(line, rep):
    """what a single N-rank line must say about its own collective layer"""
    n = int(line["n_gpus"])
    if n < 2:
        return
    tag = "N=%d" % n
    ranks = get(line, "collective_ranks_seen", "collective", "ranks_seen")
    rep.expect(ranks == n, tag + " ranks that met in the collective layer", "%r of %d" % (ranks, n))
    gpus = get(line, "collective_distinct_gpus", "collective", "distinct_gpus")
    rep.expect(gpus == n, tag + " distinct GPUs", "%r of %d" % (gpus, n))
    same = get(line, "collective_boundary_checksum_equal", "collective", "boundary_checksum_equal")
    rep.expect(same is True, tag + " boundary checksums equal on all ranks", repr(same))
    pre = get(line, "collective_rccl_preflight_ok", "collective", "rccl_preflight", "ok")
    backend = get(line, "collective_backend", "collective", "backend")
    fallback = get(line, "collective_backend_fallback", "collective", "backend_fallback")
    rep.expect(pre is True and backend == "nccl" and not fallback, tag + " RCCL pre-flight passed, barriers over RCCL",
               "preflight ok %r, backend %r%s" % (pre, backend, ", fall-back: " + str(fallback)[:120] if fallback else ""))
    side = (line.get("config") or {}).get("group_streams_side_by_side_by_rank")
    rep.expect(isinstance(side, list) and len(side) == n and all(side), tag + " member groups side by side on every rank", repr(side))
    note = get(line, "one_process_boundary_broadcast_note", "one_process", "boundary_broadcast", "note")
    err = get(line, "one_process_error", "one_process", "error")
    if note is None and err is None:
        rep.skip(tag + " one-process boundary broadcast", "no one_process object in the line (budget, or not cfg4)")
    else:
        want = "one RCCL broadcast to %d other device" % (n - 1)
        rep.expect(isinstance(note, str) and note.startswith(want), tag + " one process: ONE RCCL broadcast over xGMI",
                   repr(note) if err is None else "failed: %s" % err)


def check(lines, bench_line, rep):
    by_n = {}
    for ln in lines:
        by_n.setdefault(int(ln["n_gpus"]), ln)
    if not by_n:
        rep.expect(False, "bench lines in the scaling record", "none found")
        return
    one = by_n.get(1)
    if one is not None and bench_line is not None:
        r = one["value"] / bench_line["value"]
        rep.expect(abs(r - 1.0) <= 0.10, "N=1 of the scaling record against BENCH", "%.0f / %.0f = %.3f (within 10 %%)" % (one["value"], bench_line["value"], r))
    elif bench_line is not None:
        rep.skip("N=1 against BENCH", "no N=1 line in the scaling record")
    base = one or bench_line
    for n in sorted(by_n):
        ln = by_n[n]
        if n > 1 and base is not None and ln.get("scaling") == "weak":
            r = ln["value"] / n / base["value"]
            rep.expect(abs(r - 1.0) <= 0.05, "weak scaling at N=%d flat" % n, "value / N / value(1) = %.3f (within 5 %%)" % r)
        check_line(ln, rep)
    proj = None
    for ln in ([one] if one else []) + ([bench_line] if bench_line else []):
        proj = proj or get(ln, "projected_8gpu_cfg4_value", "projected_8gpu_cfg4", "value")
    eight = by_n.get(8)
    if eight is None:
        rep.skip("BASELINE cfg 4 as worded (8 GPUs) against its projection", "no N=8 line")
    else:
        strong = get(eight, "cfg4_strong_value", "cfg4_strong", "value")
        if strong is None or proj is None:
            rep.skip("BASELINE cfg 4 as worded (8 GPUs) against its projection", "cfg4_strong %r, projection %r" % (strong, proj))
        else:
            rep.expect(abs(strong / proj - 1.0) <= 0.10, "BASELINE cfg 4 as worded (64 members, 8 per GPU) against projected_8gpu_cfg4",
                       "%.0f measured / %.0f projected = %.3f (within 10 %%)" % (strong, proj, strong / proj))
            if base is not None:
                rep.rows.append((None, "cfg 4 as worded, speed-up over 64 members on one GPU", "%.2f x (expected 4.4-4.5)" % (strong / base["value"])))


def main(argv):
    rep = Report()
    if len(argv) >= 2 and argv[0] == "--line":
        doc = [json.loads(ln) for ln in open(argv[1]) if ln.lstrip().startswith("{")]
        lines = lines_of(doc)
        for ln in lines:
            check_line(ln, rep)
        if not lines:
            rep.expect(False, "bench line", "none found in " + argv[1])
    else:
        if not argv:
            print(__doc__)
            return 2
        scale = json.load(open(argv[0]))
        if isinstance(scale, dict) and scale.get("skipped"):
            print("skipped record: %s" % scale.get("reason"))
            return 0
        bench = lines_of(json.load(open(argv[1])))[:1] if len(argv) > 1 else []
        check(lines_of(scale), bench[0] if bench else None, rep)
    rep.show()
    bad = rep.deviations()
    print("%d expectation(s) checked, %d deviation(s)" % (sum(1 for r in rep.rows if r[0] is not None), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
