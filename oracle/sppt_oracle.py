"""numpy restatement of the deterministic SPPT scheme of pyspeedy_amd/csrc/sppt.hip.

TEST INFRASTRUCTURE.  PARITY UNPINNED with respect to the reference: its sppt.f90 is compiled out (params.f90:44) and cannot
work as written (see the header of csrc/sppt.hip), so this file restates OUR definition of the scheme the reference documents
(sppt.f90:27-36, 62-112; physics.f90:234-248), for checking the HIP kernels against an independent implementation.
"""
import numpy as np

MASK = (1 << 64) - 1
KX, NSPEC, TRUNC = 8, 992, 30


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def constants():
    time_decorr, len_decorr, stddev, rearth = 6.0, 500000.0, float(np.float32(0.33)), float(np.float32(6.371e6))
    phi = np.exp(-(24.0 / 36.0) / time_decorr)
    n = np.arange(1, TRUNC + 1)
    total = 0.0
    for k in n:  # same summation order as the host code
        total += (2 * k + 1) * np.exp(-0.5 * (len_decorr / rearth) * (len_decorr / rearth) * k * (k + 1))
    f0 = np.sqrt((stddev * stddev * (1.0 - phi * phi)) / (2.0 * total))
    return phi, f0, 0.25 * len_decorr * len_decorr


def eta(seed, member, step):
    """Clipped complex normals for one member and step: array [KX, NSPEC] (level, coefficient index m + 31 n)."""
    with np.errstate(over="ignore"):
        idx = np.arange(KX * NSPEC, dtype=np.uint64)
        counter = (np.uint64(member) << np.uint64(40)) | (np.uint64(step) << np.uint64(14)) | idx
        h1 = mix64(np.uint64(seed) ^ mix64(counter))
        h2 = mix64(h1 + np.uint64(0x9E3779B97F4A7C15))
    u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53
    u2 = (h2 >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 6.283185307179586 * u2
    er, ei = rad * np.cos(ang), rad * np.sin(ang)
    clip = lambda x: np.minimum(10.0, np.abs(x)) * np.where(x < 0.0, -1.0, 1.0)
    return (clip(er) + 1j * clip(ei)).reshape(KX, NSPEC)


def advance(spec, el2, seed, member, step):
    """One AR(1) step of the spectral pattern [KX, NSPEC]; spec = None on the first step.  el2: [NSPEC]."""
    phi, f0, q = constants()
    sigma = f0 * np.exp(-q * el2)[None, :]
    e = eta(seed, member, step)
    if spec is None:
        return sigma / np.sqrt(1.0 - phi * phi) * e
    return phi * spec + sigma * e


def perturb(tend, tend_dyn, pattern):
    r = np.minimum(1.0, np.abs(pattern)) * np.where(pattern < 0.0, -1.0, 1.0)
    return (1.0 + r) * (tend - tend_dyn) + tend_dyn
