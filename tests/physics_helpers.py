"""Test infrastructure (NOT part of the product package): synthetic-but-physical inputs for one member of the column
physics and the HIP-vs-oracle check that __graft_entry__.smoke(), tests/ and tools/perf_physics.py share."""
import numpy as np
import torch

from pyspeedy_amd.physics import (IL, IX, KX, SHORTWAVE_IN, STATE_IN_2D, STATE_IN_3D, SURFACE_IN, TENDENCIES, ColumnPhysics,
                                  PhysicsState, from_device_layout, to_device_layout)


def smoke_check(sp, orc):
    """One member of column physics on synthetic-but-physical inputs, HIP vs the CPU oracle.  Returns the worst
    scaled error over the tendencies."""
    inp = synthetic_member(seed=0)
    ref = orc.physics({("qg_in" if k == "qg" else k): v for k, v in inp.items()}, True, 0.3)
    phys = ColumnPhysics(sp)
    st = PhysicsState(1, sp.device)
    dev = lambda a: torch.from_numpy(to_device_layout(a)[None]).to(sp.device)
    fields = {n: dev(inp[n]) for n in STATE_IN_3D + STATE_IN_2D}
    tend = {n: dev(inp[n]) for n in TENDENCIES}
    forcing = {n: dev(inp[n]) for n in SURFACE_IN + SHORTWAVE_IN}
    phys(fields, tend, forcing, st, True, 0.3)
    torch.cuda.synchronize()
    worst = 0.0
    for n in TENDENCIES:
        got = from_device_layout(tend[n][0])
        worst = max(worst, float(np.abs(got - ref[n]).max() / max(np.abs(ref[n]).max(), 1e-300)))
    assert worst < 1e-11, worst
    return worst


def synthetic_member(seed=0):
    """Physically plausible synthetic inputs for one member, reference host layout (ix, il[, kx])."""
    rng = np.random.default_rng(seed)
    fsg = np.array([0.025, 0.095, 0.2, 0.34, 0.51, 0.685, 0.835, 0.95])
    lat = np.linspace(-87.2, 87.2, IL)[None, :, None] * np.ones((IX, 1, 1))
    tsfc = 288.0 - 40.0 * np.sin(np.deg2rad(lat)) ** 2
    tg = tsfc * fsg[None, None, :] ** 0.19 + rng.standard_normal((IX, IL, KX))
    tg = np.maximum(tg, 205.0)
    qg = 12.0 * np.exp(-(1 - fsg[None, None, :]) * 6.0) * np.cos(np.deg2rad(lat)) ** 2 * rng.uniform(0.3, 1.1, (IX, IL, KX))
    phig = 287.0 * 260.0 * np.log(1.0 / fsg)[None, None, :] + 50.0 * rng.standard_normal((IX, IL, KX))
    u = 10.0 * rng.standard_normal((IX, IL, KX))
    v = 5.0 * rng.standard_normal((IX, IL, KX))
    two = lambda lo, hi: rng.uniform(lo, hi, (IX, IL))
    fmask = (two(0, 1) > 0.6) * two(0.2, 1.0)
    inp = dict(ug=u, vg=v, tg=tg, qg=qg, phig=phig, pslg=np.log(two(0.85, 1.03)),
               utend=1e-5 * rng.standard_normal((IX, IL, KX)), vtend=1e-5 * rng.standard_normal((IX, IL, KX)),
               ttend=1e-5 * rng.standard_normal((IX, IL, KX)), qtend=1e-6 * rng.standard_normal((IX, IL, KX)),
               fmask_land=fmask, phis0=two(0, 3000.0) * (fmask > 0), forog=two(1.0, 1.3),
               sst_am=tsfc[:, :, 0] + two(-2, 2), alb_land=two(0.1, 0.5), alb_sea=two(0.07, 0.3), snowc=two(0, 1) ** 4,
               land_temp=tsfc[:, :, 0] + two(-5, 5), soil_avail_water=two(0, 1),
               flux_solar_in=two(0, 450.0), flux_ozone_upper=two(0, 8.0), flux_ozone_lower=two(0, 8.0),
               zenit_correction=two(1.0, 1.8), stratospheric_correction=two(0, 6.0), alb_surface=two(0.07, 0.5))
    return {k: np.asfortranarray(v.astype(np.float64)) for k, v in inp.items()}
