"""The column-physics schemes one by one, in the order and with the hand-overs of get_physical_tendencies
(speedy.f90/physics.f90:107-231): the same chain driven against the REFERENCE routines (oracle/ref_shim.f90 through
refmodel.RefModel) to generate tests/golden/schemes_*.npz, and against the C restatement (oracle/orc_physics.c) to check it
scheme by scheme.  TEST INFRASTRUCTURE.

Every scheme gets exactly the inputs stored in the golden file (the reference's own outputs of the schemes before it), so a
difference is attributed to the scheme it appears in -- not to whatever fed it.
"""
import ctypes as C

import numpy as np

IX, IL, KX = 96, 48, 8
SURFACE = ("fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp", "soil_avail_water")
SHORTWAVE_IN = ("flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction", "stratospheric_correction",
                "alb_surface")
# inputs of the chain that are not produced by a scheme (stored in the golden files as in_<name>)
CHAIN_INPUTS = ("tg", "qg", "phig", "ua", "va", "psg", "se", "gse", "air_absortivity_co2") + SURFACE + SHORTWAVE_IN
# outputs per scheme, in call order (stored as <scheme>_<name>)
SCHEME_OUTPUTS = (
    ("qsat", ("qsat", "rh")),
    ("convection", ("itop", "cbmf", "precnv", "dfse", "dfqa")),
    ("lsc", ("iptop", "precls", "dtlsc", "dqlsc")),
    ("clouds", ("icltop", "cloudc", "clstr", "qcloud_equiv")),
    ("shortwave", ("tsr", "ssrd", "ssr", "tt_rsw", "rad_tau2", "rad_strat_corr")),
    ("lw_down", ("slrd", "dfabs", "rad_flux", "rad_st4a")),
    ("surface_fluxes", ("ustr", "vstr", "shf", "evap", "slru", "hfluxn", "ts", "tskin", "u0", "v0", "t0")),
    ("lw_up", ("slr", "olr", "dfabs", "rad_flux")),
    ("vdiff", ("ut", "vt", "tt", "qt")),
)


def z(*shape, dtype=np.float64):
    return np.zeros(shape, dtype=dtype, order="F")


def f(a, dtype=np.float64):
    return np.array(a, dtype=dtype, order="F", copy=True)


class ReferenceBackend:
    """The reference's own subroutines (flang-compiled), through the bind(C) shims."""

    def __init__(self, model):
        self.m = model

    def qsat(self, ta, ps, sig):
        out = z(IX, IL)
        fn = getattr(__import__("refmodel").lib(), "shim_qsat")
        fn(ta.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p), C.c_double(sig), out.ctypes.data_as(C.c_void_p))
        return out

    def convection(self, psg, se, qg, qsat, itop, cbmf, precnv, dfse, dfqa):
        self.m.call("convection", psg, se, qg, qsat, itop, cbmf, precnv, dfse, dfqa)

    def lsc(self, psg, qg, qsat, iptop, precls, dtlsc, dqlsc):
        self.m.call("lsc", psg, qg, qsat, iptop, precls, dtlsc, dqlsc)

    def clouds(self, qg, rh, precnv, precls, iptop, gse, fmask, icltop, cloudc, clstr, qcloud):
        fn = getattr(__import__("refmodel").lib(), "shim_clouds")
        fn(*[a.ctypes.data_as(C.c_void_p) for a in (qg, rh, precnv, precls, iptop, gse, fmask, icltop, cloudc, clstr, qcloud)])

    def shortwave(self, forcing, co2, psg, qg, icltop, cloudc, clstr, qcloud, out):
        for n in SHORTWAVE_IN + ("fmask_land",):
            self.m.set(n, forcing[n])
        self.m.set("qcloud_equiv", qcloud)  # state%qcloud_equiv, written by clouds() just before (physics.f90:154-155)
        self.m.set("air_absortivity_co2", float(np.asarray(co2).item()))
        self.m.call("shortwave", psg, qg, icltop, cloudc, clstr)
        for n in out:
            out[n][...] = self.m.get(n)

    def lw_down(self, ta, slrd, dfabs, rad_flux, rad_tau2, rad_st4a):
        self.m.call("lw_down", ta, slrd, dfabs, rad_flux, rad_tau2, rad_st4a)

    def lw_up(self, ta, ts, slrd, slru3, slr, olr, dfabs, rad_flux, rad_tau2, rad_st4a, strat):
        self.m.call("lw_up", ta, ts, slrd, slru3, slr, olr, dfabs, rad_flux, rad_tau2, rad_st4a, strat)

    def surface_fluxes(self, *args):
        self.m.call("surface_fluxes", *args)

    def vdiff(self, se, rh, qg, qsat, phig, icnv, ut, vt, tt, qt):
        self.m.call("vdiff", se, rh, qg, qsat, phig, icnv, ut, vt, tt, qt)


class OracleBackend:
    """The plain-C restatement (oracle/orc_physics.c)."""

    def __init__(self):
        import oracle as orc
        self.orc = orc

    def qsat(self, ta, ps, sig):
        out = z(IX, IL)
        self.orc.lib().orc_qsat(ta.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p), C.c_double(sig),
                                out.ctypes.data_as(C.c_void_p), C.c_int(IX * IL))
        return out

    def convection(self, *a):
        self.orc._call("convection", *a)

    def lsc(self, *a):
        self.orc._call("lsc", *a)

    def clouds(self, *a):
        self.orc.lib().orc_clouds(*[x.ctypes.data_as(C.c_void_p) for x in a])

    def shortwave(self, forcing, co2, psg, qg, icltop, cloudc, clstr, qcloud, out):
        io = self.orc.PhysIO()
        keep = [qcloud]
        io.qcloud_equiv = qcloud.ctypes.data_as(C.c_void_p).value
        for n in SHORTWAVE_IN + ("fmask_land",):
            a = f(forcing[n])
            keep.append(a)
            setattr(io, n, a.ctypes.data_as(C.c_void_p).value)
        for n, a in out.items():
            setattr(io, n, a.ctypes.data_as(C.c_void_p).value)
        rf = z(IX, IL, 4)
        io.rad_flux = rf.ctypes.data_as(C.c_void_p).value
        io.air_absortivity_co2 = float(np.asarray(co2).item())
        io.compute_shortwave = 1
        self.orc.lib().orc_shortwave(C.byref(self.orc.tables()), C.byref(io), *[x.ctypes.data_as(C.c_void_p)
                                                                               for x in (psg, qg, icltop, cloudc, clstr)])

    def lw_down(self, *a):
        self.orc._call("lw_down", a[0], a[1], a[2], a[3], a[4], a[5])

    def lw_up(self, ta, ts, slrd, slru3, slr, olr, dfabs, rad_flux, rad_tau2, rad_st4a, strat):
        self.orc._call("lw_up", ta, ts, slrd, slru3, slr, olr, dfabs, rad_flux, rad_tau2, rad_st4a, strat)

    def surface_fluxes(self, *a):
        self.orc._call("surface_fluxes", *a)

    def vdiff(self, *a):
        self.orc._call("vdiff", *a)


def run_chain(backend, inp, fsg, feed=None):
    """Run every scheme once.  inp: CHAIN_INPUTS.  feed: golden outputs {scheme: {name: array}} handed to the FOLLOWING
    schemes instead of this backend's own results (None: the backend feeds itself -- how the goldens are generated).
    Returns {scheme: {name: array}}."""
    res = {}
    src = lambda scheme, name: (feed if feed is not None else res)[scheme][name]
    tg, qg, phig, psg, se = (f(inp[n]) for n in ("tg", "qg", "phig", "psg", "se"))
    # humidity.f90:17-28 spec_hum_to_rel_hum, level by level
    qsat, rh = z(IX, IL, KX), z(IX, IL, KX)
    for k in range(KX):
        qsat[:, :, k] = backend.qsat(f(tg[:, :, k]), psg, float(fsg[k]))
    rh[...] = qg / qsat
    res["qsat"] = dict(qsat=qsat, rh=rh)
    qsat, rh = f(src("qsat", "qsat")), f(src("qsat", "rh"))
    # convection
    o = dict(itop=z(IX, IL, dtype=np.int32), cbmf=z(IX, IL), precnv=z(IX, IL), dfse=z(IX, IL, KX), dfqa=z(IX, IL, KX))
    backend.convection(psg, se, qg, qsat, o["itop"], o["cbmf"], o["precnv"], o["dfse"], o["dfqa"])
    res["convection"] = o
    # large-scale condensation (itop is updated in place -> iptop)
    o = dict(iptop=f(src("convection", "itop"), np.int32), precls=z(IX, IL), dtlsc=z(IX, IL, KX), dqlsc=z(IX, IL, KX))
    backend.lsc(psg, qg, qsat, o["iptop"], o["precls"], o["dtlsc"], o["dqlsc"])
    res["lsc"] = o
    # clouds + shortwave
    o = dict(icltop=z(IX, IL, dtype=np.int32), cloudc=z(IX, IL), clstr=z(IX, IL), qcloud_equiv=z(IX, IL))
    backend.clouds(qg, rh, f(src("convection", "precnv")), f(src("lsc", "precls")), f(src("lsc", "iptop"), np.int32),
                   f(inp["gse"]), f(inp["fmask_land"]), o["icltop"], o["cloudc"], o["clstr"], o["qcloud_equiv"])
    res["clouds"] = o
    o = dict(tsr=z(IX, IL), ssrd=z(IX, IL), ssr=z(IX, IL), tt_rsw=z(IX, IL, KX), rad_tau2=z(IX, IL, KX, 4),
             rad_strat_corr=z(IX, IL, 2))
    backend.shortwave(inp, inp["air_absortivity_co2"], psg, qg, f(src("clouds", "icltop"), np.int32),
                      f(src("clouds", "cloudc")), f(src("clouds", "clstr")), f(src("clouds", "qcloud_equiv")), o)
    res["shortwave"] = o
    # longwave down
    o = dict(slrd=z(IX, IL), dfabs=z(IX, IL, KX), rad_flux=z(IX, IL, 4), rad_st4a=z(IX, IL, KX, 2))
    backend.lw_down(tg, o["slrd"], o["dfabs"], o["rad_flux"], f(src("shortwave", "rad_tau2")), o["rad_st4a"])
    res["lw_down"] = o
    # surface fluxes
    o = dict(ustr=z(IX, IL, 3), vstr=z(IX, IL, 3), shf=z(IX, IL, 3), evap=z(IX, IL, 3), slru=z(IX, IL, 3), hfluxn=z(IX, IL, 3),
             ts=z(IX, IL), tskin=z(IX, IL), u0=z(IX, IL), v0=z(IX, IL), t0=z(IX, IL))
    backend.surface_fluxes(psg, f(inp["ua"]), f(inp["va"]), tg, qg, rh, phig, f(inp["phis0"]), f(inp["fmask_land"]),
                           f(inp["forog"]), f(inp["sst_am"]), f(src("shortwave", "ssrd")), f(src("lw_down", "slrd")),
                           o["ustr"], o["vstr"], o["shf"], o["evap"], o["slru"], o["hfluxn"], o["ts"], o["tskin"], o["u0"],
                           o["v0"], o["t0"], f(inp["alb_land"]), f(inp["alb_sea"]), f(inp["snowc"]), f(inp["land_temp"]),
                           f(inp["soil_avail_water"]))
    res["surface_fluxes"] = o
    # longwave up (dfabs and rad_flux continue from the downward sweep)
    o = dict(slr=z(IX, IL), olr=z(IX, IL), dfabs=f(src("lw_down", "dfabs")), rad_flux=f(src("lw_down", "rad_flux")))
    backend.lw_up(tg, f(src("surface_fluxes", "ts")), f(src("lw_down", "slrd")), f(src("surface_fluxes", "slru")[:, :, 2]),
                  o["slr"], o["olr"], o["dfabs"], o["rad_flux"], f(src("shortwave", "rad_tau2")),
                  f(src("lw_down", "rad_st4a")), f(src("shortwave", "rad_strat_corr")))
    res["lw_up"] = o
    # vertical diffusion
    o = dict(ut=z(IX, IL, KX), vt=z(IX, IL, KX), tt=z(IX, IL, KX), qt=z(IX, IL, KX))
    icnv = f(KX - src("convection", "itop"), np.int32)
    backend.vdiff(se, rh, qg, qsat, phig, icnv, o["ut"], o["vt"], o["tt"], o["qt"])
    res["vdiff"] = o
    return res
