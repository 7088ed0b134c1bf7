"""Host-side mirror of the reference's column-physics driver ``get_physical_tendencies``
(speedy.f90/physics.f90:14-256) on MI355X.

The reference routine does two things: (1) 41 spectral->grid transforms of the time-level-1 state
(physics.f90:89-101) and (2) the per-column schemes.  Here (1) is issued through ModSpectral (batched over levels
and ensemble members) and (2) is ONE fused HIP kernel (pyspeedy_amd/csrc/physics.hip) behind spd_physics.

Arrays are torch float64 CUDA tensors, member-major, the reference's Fortran order inside one member:
    (ix,il)       -> [M, 48, 96]          (ix,il,kx)    -> [M, 8, 48, 96]
    (ix,il,3)     -> [M, 3, 48, 96]       (ix,il,kx,4)  -> [M, 4, 8, 48, 96]   (ix,il,kx,2) -> [M, 2, 8, 48, 96]
"""
import ctypes as C

import torch

from . import _lib
from ._lib import IL, IX, KX, MX, NX, PhysicsArgs, check

STATE_IN_3D = ("ug", "vg", "tg", "qg", "phig")
STATE_IN_2D = ("pslg",)
TENDENCIES = ("utend", "vtend", "ttend", "qtend")
SURFACE_IN = ("fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp",
              "soil_avail_water")
SHORTWAVE_IN = ("flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction",
                "stratospheric_correction", "alb_surface")
OUT_2D = ("precnv", "precls", "cbmf", "slrd", "slr", "olr")
OUT_AUX = ("slru", "ustr", "vstr", "shf", "evap", "hfluxn")
PERSIST_2D = ("tsr", "ssrd", "ssr", "qcloud_equiv")
DIAG_F = ("ts", "tskin", "u0", "v0", "t0", "cloudc", "clstr")
DIAG_I = ("iptop", "icltop")


def shapes(M):
    s = {}
    for n in STATE_IN_3D + TENDENCIES + ("tt_rsw",):
        s[n] = (M, KX, IL, IX)
    for n in STATE_IN_2D + SURFACE_IN + SHORTWAVE_IN + OUT_2D + PERSIST_2D + DIAG_F + DIAG_I:
        s[n] = (M, IL, IX)
    for n in OUT_AUX:
        s[n] = (M, 3, IL, IX)
    s["rad_st4a"] = (M, 2, KX, IL, IX)
    s["rad_flux"] = (M, 4, IL, IX)
    s["rad_tau2"] = (M, 4, KX, IL, IX)
    s["rad_strat_corr"] = (M, 2, IL, IX)
    s["sppt_pattern"] = (M, KX, IL, IX)
    return s


class PhysicsState:
    """Device-resident output / persisted-radiation arrays of `nmembers` members (the physics part of ModelState_t)."""

    def __init__(self, nmembers, device, diagnostics=False):
        self.nmembers = nmembers
        shp = shapes(nmembers)
        names = OUT_2D + OUT_AUX + PERSIST_2D + ("rad_st4a", "rad_flux", "tt_rsw", "rad_tau2", "rad_strat_corr")
        for n in names:
            setattr(self, n, torch.zeros(shp[n], dtype=torch.float64, device=device))
        self.diagnostics = diagnostics
        if diagnostics:
            for n in DIAG_F:
                setattr(self, n, torch.zeros(shp[n], dtype=torch.float64, device=device))
            for n in DIAG_I:
                setattr(self, n, torch.zeros(shp[n], dtype=torch.int32, device=device))


class ColumnPhysics:
    """``get_physical_tendencies`` for a batch of ensemble members on one GPU."""

    def __init__(self, spectral):
        self.sp = spectral
        self.device = spectral.device
        self._lib = _lib.lib()

    def grid_fields_from_spectral(self, vor, div, t, q, phi, ps):
        """physics.f90:89-101 for M members: spectral time-level-1 state -> ug, vg, tg, qg, phig, pslg.

        vor, div, t, q, phi: complex128 [M, 8, 32, 31]; ps: [M, 32, 31].  41*M transforms in three launches."""
        ucos, vcos = self.sp.vort2vel(vor, div)
        uv = self.sp.spec2grid(torch.stack([ucos, vcos]), 2)
        tqp = self.sp.spec2grid(torch.stack([t, q, phi]), 1)
        pslg = self.sp.spec2grid(ps, 1)
        return dict(ug=uv[0], vg=uv[1], tg=tqp[0], qg=tqp[1], phig=tqp[2], pslg=pslg)

    def __call__(self, fields, tend, forcing, state, compute_shortwave, air_absortivity_co2, sppt_pattern=None, fp32=False):
        """Run the fused column kernel.

        fields : dict ug, vg, tg, qg, phig [M,8,48,96], pslg [M,48,96]
        tend   : dict utend, vtend, ttend, qtend [M,8,48,96] -- updated IN PLACE like the reference's arguments
        forcing: dict with SURFACE_IN (always) and SHORTWAVE_IN (needed on shortwave steps) [M,48,96]
        state  : PhysicsState (outputs and persisted radiation fields, updated in place)
        sppt_pattern : optional [M,8,48,96] multiplicative noise r; the physical part of every tendency is scaled by
                 1 + clip(r, -1, 1) (physics.f90:234-248; off in the reference)
        fp32   : column arithmetic in single precision (BASELINE cfg 5); every array stays float64
        """
        M = state.nmembers
        shp = shapes(M)
        args = PhysicsArgs()
        keep = []

        def put(name, tensor, dtype=torch.float64):
            if tensor is None:
                setattr(args, name, None)
                return
            if tensor.dtype != dtype or tuple(tensor.shape) != shp[name] or tensor.device != self.device:
                raise ValueError("%s: expected %s %s on %s, got %s %s on %s" % (
                    name, dtype, shp[name], self.device, tensor.dtype, tuple(tensor.shape), tensor.device))
            if not tensor.is_contiguous():
                raise ValueError("%s must be contiguous (it is passed to the kernel by pointer)" % name)
            keep.append(tensor)
            setattr(args, name, tensor.data_ptr())

        for n in STATE_IN_3D + STATE_IN_2D:
            put(n, fields[n])
        for n in TENDENCIES:
            put(n, tend[n])
        for n in SURFACE_IN:
            put(n, forcing[n])
        for n in SHORTWAVE_IN:
            put(n, forcing.get(n) if compute_shortwave else forcing.get(n, None))
        for n in OUT_2D + OUT_AUX + PERSIST_2D + ("rad_st4a", "rad_flux", "tt_rsw", "rad_tau2", "rad_strat_corr"):
            put(n, getattr(state, n))
        for n in DIAG_F:
            put(n, getattr(state, n, None) if state.diagnostics else None)
        for n in DIAG_I:
            put(n, getattr(state, n, None) if state.diagnostics else None, torch.int32)
        put("sppt_pattern", sppt_pattern)
        args.air_absortivity_co2 = float(air_absortivity_co2)
        args.compute_shortwave = 1 if compute_shortwave else 0
        args.fp32 = 1 if fp32 else 0
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_physics(self.sp.handle, C.byref(args), M, stream), "spd_physics")
        return tend


# ---- reference host layout <-> device layout -----------------------------
def to_device_layout(a):
    """reference host array (ix, il[, ...]) -> [..., il, ix] contiguous numpy (reversed axis order)."""
    import numpy as np
    a = np.asarray(a)
    return np.ascontiguousarray(a.transpose(tuple(range(a.ndim - 1, -1, -1))))


def from_device_layout(t):
    a = t.cpu().numpy()
    return a.transpose(tuple(range(a.ndim - 1, -1, -1)))
