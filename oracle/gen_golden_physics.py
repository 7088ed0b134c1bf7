"""Physics / model-step golden vectors from the reference (see gen_golden.py).  TEST INFRASTRUCTURE.

physics_sw.npz / physics_nosw.npz: complete inputs and outputs of get_physical_tendencies (physics.f90:14) taken
from the running example_bc model (1982-01-01, zero SST anomaly), on every third longitude (32 x 48 columns; the
physics is column-local, so tests tile the columns back to 96 longitudes).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refmodel as R  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
IX, IL, KX, MX, NX = 96, 48, 8, 31, 32
SUB = 3  # keep every third longitude

STATE_2D_IN = ("fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp",
               "soil_avail_water", "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction",
               "stratospheric_correction", "alb_surface")
OUT_FIELDS = ("precnv", "precls", "cbmf", "slrd", "slr", "olr", "slru", "ustr", "vstr", "shf", "evap", "hfluxn",
              "rad_st4a", "rad_flux")
PERSIST = ("tt_rsw", "rad_tau2", "rad_strat_corr", "tsr", "ssrd", "ssr", "qcloud_equiv")


def z(*s, dt=np.float64):
    return np.zeros(s, dtype=dt, order="F")


def physics_inputs_from_state(m):
    """What physics.f90:89-101 computes from the spectral state at time level j1 = 1."""
    vor, div, t, tr, phi, ps = (m.get(n) for n in ("vor", "div", "t", "tr", "phi", "ps"))
    ug, vg, tg, qg, phig = (z(IX, IL, KX) for _ in range(5))
    for k in range(KX):
        u, v = z(MX, NX, dt=np.complex128), z(MX, NX, dt=np.complex128)
        m.call("vort2vel", np.asfortranarray(vor[:, :, k, 0]), np.asfortranarray(div[:, :, k, 0]), u, v)
        ug[:, :, k] = m.spec2grid(u, 2)
        vg[:, :, k] = m.spec2grid(v, 2)
        tg[:, :, k] = m.spec2grid(t[:, :, k, 0], 1)
        qg[:, :, k] = m.spec2grid(tr[:, :, k, 0, 0], 1)
        phig[:, :, k] = m.spec2grid(phi[:, :, k], 1)
    pslg = m.spec2grid(ps[:, :, 0], 1)
    return dict(ug=ug, vg=vg, tg=tg, qg_in=qg, phig=phig, pslg=pslg)


def snapshot(m, compute_shortwave, seed):
    rng = np.random.default_rng(seed)
    inp = physics_inputs_from_state(m)
    for n in STATE_2D_IN:
        inp[n] = m.get(n)
    pre = {n: m.get(n) for n in PERSIST}
    tend = {n: np.asfortranarray(1e-5 * rng.standard_normal((IX, IL, KX))) for n in ("utend", "vtend", "ttend", "qtend")}
    # repeat every SUB-th longitude so that the sub-sampled problem is self-contained
    m.set("compute_shortwave", 1 if compute_shortwave else 0)
    tin = {n: np.asfortranarray(np.repeat(a[::SUB], SUB, axis=0)) for n, a in tend.items()}
    tout = {n: a.copy(order="F") for n, a in tin.items()}
    m.call("physics", 1, tout["utend"], tout["vtend"], tout["ttend"], tout["qtend"])
    out = {n: m.get(n) for n in OUT_FIELDS + PERSIST}
    sub = lambda a: np.ascontiguousarray(a[::SUB])
    data = {"in_" + n: sub(a) for n, a in inp.items()}
    data.update({"in_" + n: sub(a) for n, a in tin.items()})
    data.update({"pre_" + n: sub(a) for n, a in pre.items()})
    data.update({"out_" + n: sub(a) for n, a in tout.items()})
    data.update({"out_" + n: sub(a) for n, a in out.items()})
    data["air_absortivity_co2"] = np.float64(m.get("air_absortivity_co2"))
    data["compute_shortwave"] = np.int32(1 if compute_shortwave else 0)
    return data


def gen_physics():
    bc = np.load(os.path.join(GOLD, "example_bc.npz"))
    m = R.RefModel()
    m.set_bc(bc)
    for _ in range(39):  # into day 2: step 39 is a shortwave step (mod(39,3)==0), moist processes active
        assert m.step() == 0
    d = snapshot(m, True, 1)
    np.savez_compressed(os.path.join(GOLD, "physics_sw.npz"), **d)
    print("physics_sw.npz", sum(v.nbytes for v in d.values()) // 1024, "KiB raw")
    assert m.step() == 0
    d = snapshot(m, False, 2)
    np.savez_compressed(os.path.join(GOLD, "physics_nosw.npz"), **d)
    print("physics_nosw.npz", sum(v.nbytes for v in d.values()) // 1024, "KiB raw")
    return m


def gen_steps():
    print("step snapshots: not generated yet (dynamics glue is a later row of SURVEY section 8f)")


if __name__ == "__main__":
    gen_physics()
