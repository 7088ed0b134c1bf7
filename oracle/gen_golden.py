"""Generate the committed golden vectors under tests/golden/ from the *reference itself*.

TEST INFRASTRUCTURE.  Runs only in the build container (needs oracle/_ref/libspeedy_ref.so, i.e.
/root/reference compiled by oracle/build_ref.sh).  The vectors are data: inputs and the outputs the
flang-compiled reference Fortran produced for them.

    python oracle/gen_golden.py [tables] [transforms] [physics] [steps]

Inputs follow SURVEY.md section 8(d) cfg 2: triangular spectra = complex normal(0,1)*(1+l)^-1 with
Im(m=0)=0, numpy default_rng(1234); raw grids normal(0,1), default_rng(4321); plus real model fields
from the example_bc run started 1982-01-01 with zero SST anomaly.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refmodel as R  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
IX, IL, KX, MX, NX, IY = 96, 48, 8, 31, 32, 24


def z(*s, dt=np.float64):
    return np.zeros(s, dtype=dt, order="F")


def cz():
    return z(MX, NX, dt=np.complex128)


def new_model():
    bc = np.load(os.path.join(GOLD, "..", "..", "pyspeedy_amd", "data", "example_bc.npz"))
    m = R.RefModel()
    m.set_bc(bc)
    return m


def tri_spectra(n, seed=1234):
    """[n, 31(m), 32(n)] complex, triangular truncation l = m + n <= 30 (0-based m, n)."""
    rng = np.random.default_rng(seed)
    mm, nn = np.meshgrid(np.arange(MX), np.arange(NX), indexing="ij")
    ll = mm + nn
    s = (rng.standard_normal((n, MX, NX)) + 1j * rng.standard_normal((n, MX, NX))) / (1.0 + ll)
    s[:, ll > 30] = 0.0
    s[:, 0, :] = s[:, 0, :].real
    return s


def gen_tables(m):
    g = dict(hsg=z(9), dhs=z(8), fsg=z(8), dhsr=z(8), fsgr=z(8), radang=z(48), coriol=z(48), sia=z(48), coa=z(48),
             sia_half=z(24), coa_half=z(24), cosgr=z(48), cosgr2=z(48), sigl=z(8), sigh=z(9), grdsig=z(8),
             grdscp=z(8), wvi=z(8, 2))
    m.call("geometry", *g.values())
    leg = dict(epsi=z(32, 33), repsi=z(32, 33), cpol=z(62, 32, 24), nsh2=z(32, dt=np.int32), wt=z(24))
    m.call("legendre_tables", *leg.values())
    # cpol(2m-1,n,j) == cpol(2m,n,j) (legendre.f90:102-105): store the unique polynomials only
    assert np.array_equal(leg["cpol"][0::2], leg["cpol"][1::2])
    leg["poly"] = leg.pop("cpol")[0::2].copy()
    f = dict(work=z(96), ifac=z(15, dt=np.int32))
    m.call("fft_tables", *f.values())
    # slots rffti1 never writes are uninitialised in the reference: zero them in the fixture
    for i in (47, 48, 59, 60, 71, 72, 83, 84, 87, 90, 93, 94, 95, 96):
        f["work"][i - 1] = 0.0
    f["ifac"][6:] = 0
    s = {k: z(31, 32) for k in ("el2", "elm2", "el4", "trfilt")}
    s["gradx"] = z(31)
    for k in ("gradym", "gradyp", "uvdx", "uvdym", "uvdyp", "vddym", "vddyp"):
        s[k] = z(31, 32)
    m.call("spectral_tables", *s.values())
    s["gradym"][:, 0] = 0.0  # never initialised by the reference (spectral.f90:97-107)
    out = {**g, **leg, **f, **s, "fband": m.get("fband"), "xgeop1": m.get("xgeop1"), "xgeop2": m.get("xgeop2")}
    np.savez_compressed(os.path.join(GOLD, "tables.npz"), **out)
    print("tables.npz:", len(out), "arrays")


def gen_transforms(m):
    out = {}
    spec = tri_spectra(6)
    # two non-triangular spectra (full rhomboid incl. n=32 column) exercise nsh2 / l=31 handling
    rng = np.random.default_rng(99)
    full = rng.standard_normal((2, MX, NX)) + 1j * rng.standard_normal((2, MX, NX))
    spec = np.concatenate([spec, full])
    # real model fields after one day: T and vorticity at level 4, ln(ps)
    for _ in range(36):
        assert m.step() == 0
    t = m.get("t"); vor = m.get("vor"); div = m.get("div"); ps = m.get("ps")
    real = np.stack([t[:, :, 3, 0], vor[:, :, 3, 0], ps[:, :, 0]])
    spec = np.concatenate([spec, real])
    nb = spec.shape[0]
    out["spec_in"] = spec
    out["spec2grid_k1"] = np.stack([m.spec2grid(spec[b], 1) for b in range(nb)])
    out["spec2grid_k2"] = np.stack([m.spec2grid(spec[b], 2) for b in range(nb)])
    leg = []
    for b in range(nb):
        o = z(62, 48)
        sr = np.ascontiguousarray(spec[b].T).view(np.float64).reshape(32, 62).T.copy(order="F")
        m.call("legendre_inv", sr, o)
        leg.append(o)
    out["legendre_inv"] = np.stack(leg)
    four = []
    for b in range(nb):
        o = z(96, 48)
        m.call("fourier_inv", np.asfortranarray(out["legendre_inv"][b]), o, 1)
        four.append(o)
    assert np.array_equal(np.stack(four), out["spec2grid_k1"])

    rng = np.random.default_rng(4321)
    grids = np.concatenate([out["spec2grid_k1"][:4], rng.standard_normal((3, IX, IL)),
                            np.full((1, IX, IL), 280.0), out["spec2grid_k1"][-3:]])
    ng = grids.shape[0]
    out["grid_in"] = grids
    out["grid2spec"] = np.stack([m.grid2spec(grids[b]) for b in range(ng)])
    fo, lo = [], []
    for b in range(ng):
        o = z(62, 48)
        m.call("fourier", np.asfortranarray(grids[b]), o)
        fo.append(o)
        o2 = z(62, 32)
        m.call("legendre", o, o2)
        lo.append(o2)
    out["fourier"] = np.stack(fo)
    out["legendre"] = np.stack(lo)

    # spectral-space operators on (spec[a], spec[b]) pairs
    pairs = [(0, 1), (6, 7), (9, 8)]
    v2v, vv2, gr, lap, lapi, trn = [], [], [], [], [], []
    for a, b in pairs:
        u, v = cz(), cz()
        m.call("vort2vel", np.asfortranarray(spec[a]), np.asfortranarray(spec[b]), u, v)
        v2v.append(np.stack([u, v]))
        u, v = cz(), cz()
        m.call("vel2vort", np.asfortranarray(spec[a]), np.asfortranarray(spec[b]), u, v)
        vv2.append(np.stack([u, v]))
        dx, dy = cz(), cz()
        m.call("gradient", np.asfortranarray(spec[a]).copy(order="F"), dx, dy)
        gr.append(np.stack([dx, dy]))
        o = cz(); m.call("laplacian", np.asfortranarray(spec[a]), o, 0); lap.append(o)
        o = cz(); m.call("laplacian", np.asfortranarray(spec[a]), o, 1); lapi.append(o)
        o = np.asfortranarray(spec[a]).copy(order="F"); m.call("truncate", o); trn.append(o)
    out["pairs"] = np.array(pairs)
    out["vort2vel"] = np.stack(v2v); out["vel2vort"] = np.stack(vv2); out["gradient"] = np.stack(gr)
    out["laplacian"] = np.stack(lap); out["laplacian_inv"] = np.stack(lapi); out["truncate"] = np.stack(trn)
    gv = []
    for kc in (1, 2):
        u, v = cz(), cz()
        m.call("grid_vel2vort", np.asfortranarray(grids[0]), np.asfortranarray(grids[5]), u, v, kc)
        gv.append(np.stack([u, v]))
    out["grid_vel2vort_k1k2"] = np.stack(gv)  # inputs grid_in[0], grid_in[5]
    gf = []
    for b in (4, 8):
        o = z(96, 48)
        m.call("grid_filter", np.asfortranarray(grids[b]).copy(order="F"), o)
        gf.append(o)
    out["grid_filter"] = np.stack(gf)  # inputs grid_in[4], grid_in[8]
    np.savez_compressed(os.path.join(GOLD, "transforms.npz"), **out)
    print("transforms.npz:", {k: v.shape for k, v in out.items()})
    # known answers recorded in SURVEY 8c (constant 280 field)
    k = 7
    print("KAT const 280: fourier(1,j)=%.17g  s(1,1)=%.16g" % (out["fourier"][k][0, 0], out["grid2spec"][k][0, 0].real))


def main(argv):
    what = set(argv) or {"tables", "transforms", "physics", "steps"}
    if "tables" in what:
        gen_tables(new_model())
    if "transforms" in what:
        gen_transforms(new_model())
    if "physics" in what or "steps" in what:
        import gen_golden_physics as GP
        if "physics" in what:
            GP.gen_physics()
        if "steps" in what:
            GP.gen_steps()


if __name__ == "__main__":
    main(sys.argv[1:])
