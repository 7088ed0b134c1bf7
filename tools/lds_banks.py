"""LDS bank conflicts of the transform kernels' access patterns, enumerated on the host (no GPU needed).

    python tools/lds_banks.py

For every LDS access site of csrc/transforms.hip that is not lane = row (those are conflict-free by the odd row strides, also
checked here) the script lists the byte address of every lane of every wavefront, groups the lanes as the gfx950 LDS services
the instruction (MI355X_MICROARCH.md, section LDS: ds_read_b64 2 x 32 lanes over 64 banks, ds_write_b64 4 x 16 contiguous lanes
over 32 banks, ds_read_b128 4 x 16 lanes over 64 banks, ds_write_b128 8 x 8 contiguous lanes over 32 banks) and counts LDS-array
cycles: a group costs as many cycles as the most distinct dword addresses it puts on one bank.  `before` is the mapping the
kernels use (round 5), `after` the conflict-free maps of round 6 (tools/experiments/r06_lds_lane_maps.patch): measured to buy
nothing -- profiles/r06_lds_conflicts.txt -- and not adopted; both are kept so that the table can be regenerated.
"""
IX, IL, IY, MX, NX = 96, 48, 24, 31, 32
NSPEC = MX * NX
K_ROW, K_C = 97, 63          # row strides (doubles) of the FFT row buffer R and of the compact Fourier buffer C
CBUF = IL * K_C              # S (spectral staging) starts here, in doubles
THREADS = 512

B128_READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_READ_GROUPS += [[l + 32 for l in g] for g in B128_READ_GROUPS]


def groups(op):
    if op == "read_b64":
        return [list(range(0, 32)), list(range(32, 64))], 64, 2
    if op == "write_b64":
        return [list(range(g * 16, g * 16 + 16)) for g in range(4)], 32, 2
    if op == "read_b128":
        return B128_READ_GROUPS, 64, 4
    if op == "write_b128":
        return [list(range(g * 8, g * 8 + 8)) for g in range(8)], 32, 4
    raise ValueError(op)


def cycles(op, addr_of_lane):
    """addr_of_lane: 64 byte addresses or None (inactive).  Returns (LDS-array cycles, conflict-free cycles)."""
    grp, nbanks, ndw = groups(op)
    total = ideal = 0
    for g in grp:
        per_bank = {}
        active = False
        for lane in g:
            a = addr_of_lane[lane]
            if a is None:
                continue
            active = True
            for d in range(ndw):
                dw = a // 4 + d
                per_bank.setdefault(dw % nbanks, set()).add(dw)
        if active:
            ideal += 1
            total += max(len(v) for v in per_bank.values())
    return total, ideal


def site(name, op, fn, items=THREADS, iters=1):
    """fn(tid, it) -> byte address or None for every thread of the 512-thread workgroup."""
    tot = idl = 0
    for it in range(iters):
        for w in range(THREADS // 64):
            c, i = cycles(op, [fn(w * 64 + l, it) for l in range(64)])
            tot += c
            idl += i
    print("  %-74s %-10s %4d cycles, conflict-free %4d  (x%.2f)" % (name, op, tot, idl, tot / max(idl, 1)))
    return tot, idl


def pos_re(m):
    return 0 if m == 0 else 2 * m - 1


def pos_im(m):
    return 61 if m == 0 else 2 * m


# ---------------------------------------------------------------------------- inverse Legendre epilogue (spec2grid)
def inv_epilogue_before():
    """lane = (m, jq) = (tid / 12, tid % 12) owns latitude pairs 2 jq, 2 jq + 1 (round 5)."""
    t = [0, 0]
    for q in range(2):
        for north in (1, 0):
            for part in (pos_re, pos_im):
                def fn(tid, it, q=q, north=north, part=part):
                    if tid >= MX * 12:
                        return None
                    m, jq = divmod(tid, 12)
                    if part is pos_im and m == 0:
                        return None
                    js = 2 * jq + q
                    row = IL - 1 - js if north else js
                    return 8 * (row * K_C + part(m))
                c, i = site("inv. Legendre epilogue q=%d %s %s" % (q, "north" if north else "south", part.__name__), "write_b64", fn)
                t[0] += c
                t[1] += i
    return t


def inv_epilogue_after():
    """lane = (m, jq) = (tid / 16, tid % 16), jq < 12 active, owns latitude pairs jq and jq + 12."""
    t = [0, 0]
    for q in range(2):
        for north in (1, 0):
            for part in (pos_re, pos_im):
                def fn(tid, it, q=q, north=north, part=part):
                    m, jq = divmod(tid, 16)
                    if m >= MX or jq >= 12:
                        return None
                    if part is pos_im and m == 0:
                        return None
                    js = jq + 12 * q
                    row = IL - 1 - js if north else js
                    return 8 * (row * K_C + part(m))
                c, i = site("inv. Legendre epilogue q=%d %s %s" % (q, "north" if north else "south", part.__name__), "write_b64", fn)
                t[0] += c
                t[1] += i
    return t


# ---------------------------------------------------------------------------- grid rows out of R (spec2grid) / into R (grid2spec)
def grid_task_before(idx):
    return divmod(idx, IX // 2) if idx < IL * IX // 2 else None          # (row, 16-byte piece)


def grid_task_after(idx):
    """32 consecutive lanes = 16 pieces of row 2p and the same 16 pieces of row 2p + 1 (rows 97 doubles apart: bank offset 2)."""
    if idx >= IL * IX // 2:
        return None
    g32, l = divmod(idx, 32)
    pair, chunk = divmod(g32, 3)
    return 2 * pair + (l >> 4), chunk * 16 + (l & 15)


def grid_readout(task):
    t = [0, 0]
    for half in range(2):
        def fn(tid, it, half=half):
            rp = task(tid + it * THREADS)
            return None if rp is None else 8 * (rp[0] * K_ROW + 2 * rp[1] + half)
        c, i = site("spec2grid: grid row read-out, double %d of the lane's 16 bytes" % half, "read_b64", fn, iters=5)
        t[0] += c
        t[1] += i
    return t


def pair_task_before(idx):
    return divmod(idx, IX // 2) if idx < IY * IX // 2 else None          # (latitude pair j, piece)


def pair_task_after(idx):
    """16 consecutive lanes = 8 pieces of pair 2p and the same 8 pieces of pair 2p + 1."""
    if idx >= IY * IX // 2:
        return None
    g16, l = divmod(idx, 16)
    pp, chunk = divmod(g16, 6)
    return 2 * pp + (l >> 3), chunk * 8 + (l & 7)


def grid_stagein(task):
    t = [0, 0]
    for north in (1, 0):
        for half in range(2):
            def fn(tid, it, north=north, half=half):
                jp = task(tid + it * THREADS)
                if jp is None:
                    return None
                row = IL - 1 - jp[0] if north else jp[0]
                return 8 * (row * K_ROW + 2 * jp[1] + half)
            c, i = site("grid2spec: staging store %s row, double %d" % ("north" if north else "south", half), "write_b64", fn, iters=3)
            t[0] += c
            t[1] += i
    return t


# ---------------------------------------------------------------------------- direct Legendre (grid2spec)
def dir_lanes():
    lanes = []
    for m in range(MX):
        for par in range(2):
            ns = [n for n in range(par, 31) if m + n <= 31]
            for i in range(0, len(ns), 2):
                lanes.append((m, par, ns[i], ns[i + 1] if i + 1 < len(ns) else -1))
    return lanes


def direct_legendre():
    lanes = dir_lanes()
    t = [0, 0]
    for part in (pos_re, pos_im):
        def fn(tid, it, part=part):
            if tid >= len(lanes):
                return None
            m, par, na, nb = lanes[tid]
            row = it if par else IL - 1 - it
            return 8 * (row * K_C + part(m))
        c, i = site("direct Legendre: read of %s over the 24 latitude pairs" % part.__name__, "read_b64", fn, iters=IY)
        t[0] += c
        t[1] += i
    for which in (2, 3):
        def fn(tid, it, which=which):
            if tid >= len(lanes):
                return None
            n = lanes[tid][which]
            return None if n < 0 else 8 * CBUF + 16 * (n * MX + lanes[tid][0])
        c, i = site("direct Legendre: store of coefficient %s into S" % ("a" if which == 2 else "b"), "write_b128", fn)
        t[0] += c
        t[1] += i
    return t


def lane_is_row():
    """The FFT stages: lane = latitude row, every lane the same column.  One representative column per buffer."""
    t = [0, 0]
    for nm, stride, op in (("R (stride 97)", K_ROW, "read_b64"), ("R (stride 97)", K_ROW, "write_b64"),
                           ("C (stride 63)", K_C, "read_b64"), ("C (stride 63)", K_C, "write_b64")):
        def fn(tid, it, stride=stride):
            lane = tid & 63
            return None if lane >= IL else 8 * (lane * stride + 5)
        c, i = site("FFT stage, lane = row, buffer %s" % nm, op, fn)
        t[0] += c
        t[1] += i
    return t


def main():
    print("LDS-array cycles per workgroup (one field), gfx950 banking rules; x = cycles / conflict-free cycles")
    print("\nlane = row accesses (unchanged):")
    lane_is_row()
    print("\nBEFORE (the kernels' maps):")
    b = [inv_epilogue_before(), grid_readout(grid_task_before), grid_stagein(pair_task_before), direct_legendre()]
    print("\nAFTER (the conflict-free maps of tools/experiments/r06_lds_lane_maps.patch, not adopted):")
    a = [inv_epilogue_after(), grid_readout(grid_task_after), grid_stagein(pair_task_after), direct_legendre()]
    print("\ntotals (cycles / conflict-free):")
    for nm, x, y in zip(("inverse Legendre epilogue", "spec2grid read-out", "grid2spec staging stores", "direct Legendre"), b, a):
        print("  %-28s before %5d / %5d   after %5d / %5d" % (nm, x[0], x[1], y[0], y[1]))
    for nm, task in (("grid", grid_task_after), ("pair", pair_task_after)):
        n = IL * IX // 2 if nm == "grid" else IY * IX // 2
        seen = sorted(task(i) for i in range(n))
        full = sorted((r, p) for r in range(IL if nm == "grid" else IY) for p in range(IX // 2))
        assert seen == full, nm + " task mapping is not a permutation"
    print("task mappings are permutations: ok")


if __name__ == "__main__":
    main()
