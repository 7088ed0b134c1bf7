"""pyspeedy_amd -- MI355X-native (gfx950) hot path of the SPEEDY T30L8 model as exposed by pySPEEDY.

Spectral transforms (Legendre + zonal FFT), spectral-space operators and the fused column physics run as
hand-written HIP kernels behind the C ABI in include/pyspeedy_amd.h; this package is the Python host side that
mirrors the reference's interfaces: the operator level (ModSpectral_t procedures, get_physical_tendencies), the f2py
driver function set (speedy_driver) and the user-facing Speedy / SpeedyEns classes with their callbacks.
"""
from ._lib import (IL, IX, IY, KX, MX, NX, TRUNC, SpeedyHipError, build, lib)  # noqa: F401

__all__ = ["ModSpectral", "ColumnPhysics", "Speedy", "SpeedyEns", "example_bc_file", "example_sst_anomaly_file",
           "MODEL_STATE_DEF", "DEFAULT_OUTPUT_VARS", "SpeedyHipError", "build", "lib"]


def __getattr__(name):
    # torch is only needed for device-side classes; keep `import pyspeedy_amd` light for the CPU test tier
    if name == "ModSpectral":
        from .spectral import ModSpectral
        return ModSpectral
    if name == "ColumnPhysics":
        from .physics import ColumnPhysics
        return ColumnPhysics
    if name in ("Speedy", "SpeedyEns", "example_bc_file", "example_sst_anomaly_file", "MODEL_STATE_DEF"):
        from . import speedy
        return getattr(speedy, name)
    if name == "DEFAULT_OUTPUT_VARS":
        from .registry import DEFAULT_OUTPUT_VARS
        return DEFAULT_OUTPUT_VARS
    raise AttributeError(name)
