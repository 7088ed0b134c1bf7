"""pyspeedy_amd -- MI355X-native (gfx950) hot path of the SPEEDY T30L8 model as exposed by pySPEEDY.

Spectral transforms (Legendre + zonal FFT), spectral-space operators and the fused column physics run as
hand-written HIP kernels behind the C ABI in include/pyspeedy_amd.h; this package is the Python host side that
mirrors the reference's operator interface (ModSpectral_t procedures, get_physical_tendencies).
"""
from ._lib import (IL, IX, IY, KX, MX, NX, TRUNC, SpeedyHipError, build, lib)  # noqa: F401

__all__ = ["ModSpectral", "ColumnPhysics", "SpeedyHipError", "build", "lib"]


def __getattr__(name):
    # torch is only needed for device-side classes; keep `import pyspeedy_amd` light for the CPU test tier
    if name == "ModSpectral":
        from .spectral import ModSpectral
        return ModSpectral
    if name == "ColumnPhysics":
        from .physics import ColumnPhysics
        return ColumnPhysics
    raise AttributeError(name)
