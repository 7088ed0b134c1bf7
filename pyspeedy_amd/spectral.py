"""Host-side mirror of the reference's ``ModSpectral_t`` (speedy.f90/spectral.f90:11-33) on MI355X.

Same procedure names and argument meaning as the reference type-bound procedures, with a leading batch
dimension.  All arrays are torch tensors resident on the GPU; every method only enqueues HIP kernels through the
C ABI (include/pyspeedy_amd.h) on the current torch stream -- PyTorch is used for device memory and streams only.

Layouts (C-contiguous torch tensors == the reference's Fortran order inside one field):
    spectral field  complex128 [..., 32, 31]   (reference ``complex(8) (mx=31, nx=32)``: index [n][m])
    Fourier plane   float64    [..., 48, 62]   (reference ``real(8) (2*mx, il)``)
    grid field      float64    [..., 48, 96]   (reference ``real(8) (ix, il)``: index [lat j][lon i], j=0 south)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import IL, IX, MX, NX, check


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class ModSpectral:
    """Batched spectral transforms and spectral-space operators on one GPU."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise _lib.SpeedyHipError("pyspeedy_amd needs a HIP device (torch.cuda.is_available() is False)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._lib = _lib.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._lib.spd_create(C.byref(self._h), self.device.index), "spd_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.spd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- tables ---------------------------------------------------------------------------------
    def table(self, name):
        """Host copy of a constant table under the reference's name (1-D, Fortran order)."""
        n = self._lib.spd_get_table_host(self._h, name.encode(), None, 0)
        if n < 0:
            check(int(n), "spd_get_table_host(%s)" % name)
        out = np.empty(int(n), dtype=np.float64)
        n2 = self._lib.spd_get_table_host(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.size)
        if n2 < 0:
            check(int(n2), "spd_get_table_host(%s)" % name)
        return out

    # ---- argument plumbing ------------------------------------------------------------------------
    def _spec(self, t, name):
        if t.dtype != torch.complex128 or t.shape[-2:] != (NX, MX):
            raise ValueError("%s: expected complex128 [..., %d, %d], got %s %s" % (name, NX, MX, t.dtype, tuple(t.shape)))
        return self._dev(t, name)

    def _grid(self, t, name):
        if t.dtype != torch.float64 or t.shape[-2:] != (IL, IX):
            raise ValueError("%s: expected float64 [..., %d, %d], got %s %s" % (name, IL, IX, t.dtype, tuple(t.shape)))
        return self._dev(t, name)

    def _four(self, t, name):
        if t.dtype != torch.float64 or t.shape[-2:] != (IL, 2 * MX):
            raise ValueError("%s: expected float64 [..., %d, %d], got %s %s" % (name, IL, 2 * MX, t.dtype, tuple(t.shape)))
        return self._dev(t, name)

    def _dev(self, t, name):
        if t.device != self.device:
            raise ValueError("%s: tensor is on %s, this ModSpectral is bound to %s" % (name, t.device, self.device))
        return t.contiguous()

    @staticmethod
    def _count(t):
        n = 1
        for s in t.shape[:-2]:
            n *= s
        return n

    def _new_spec(self, like):
        return torch.empty(like.shape[:-2] + (NX, MX), dtype=torch.complex128, device=self.device)

    def _new_grid(self, like):
        return torch.empty(like.shape[:-2] + (IL, IX), dtype=torch.float64, device=self.device)

    def _new_four(self, like):
        return torch.empty(like.shape[:-2] + (IL, 2 * MX), dtype=torch.float64, device=self.device)

    # ---- transforms (spectral.f90:251-273) ---------------------------------------------------------
    def spec2grid(self, vorm, kcos=1, out=None):
        vorm = self._spec(vorm, "vorm")
        out = self._new_grid(vorm) if out is None else self._grid(out, "out")
        check(self._lib.spd_spec2grid(self._h, _ptr(vorm), _ptr(out), int(kcos), self._count(vorm), _stream_ptr()),
              "spd_spec2grid")
        return out

    def grid2spec(self, vorg, out=None):
        vorg = self._grid(vorg, "vorg")
        out = self._new_spec(vorg) if out is None else self._spec(out, "out")
        check(self._lib.spd_grid2spec(self._h, _ptr(vorg), _ptr(out), self._count(vorg), _stream_ptr()),
              "spd_grid2spec")
        return out

    # stage level (legendre.f90:130-221, fourier.f90:63-123)
    def legendre_inv(self, spec):
        spec = self._spec(spec, "input")
        out = self._new_four(spec)
        check(self._lib.spd_legendre_inv(self._h, _ptr(spec), _ptr(out), self._count(spec), _stream_ptr()),
              "spd_legendre_inv")
        return out

    def legendre(self, four):
        four = self._four(four, "input")
        out = self._new_spec(four)
        check(self._lib.spd_legendre(self._h, _ptr(four), _ptr(out), self._count(four), _stream_ptr()), "spd_legendre")
        return out

    def fourier_inv(self, four, kcos=1):
        four = self._four(four, "input")
        out = self._new_grid(four)
        check(self._lib.spd_fourier_inv(self._h, _ptr(four), _ptr(out), int(kcos), self._count(four), _stream_ptr()),
              "spd_fourier_inv")
        return out

    def fourier(self, grid):
        grid = self._grid(grid, "input")
        out = self._new_four(grid)
        check(self._lib.spd_fourier(self._h, _ptr(grid), _ptr(out), self._count(grid), _stream_ptr()), "spd_fourier")
        return out

    # ---- spectral-space operators (spectral.f90:134-317) ---------------------------------------------
    def vort2vel(self, vorm, divm):
        vorm, divm = self._spec(vorm, "vorm"), self._spec(divm, "divm")
        ucosm, vcosm = self._new_spec(vorm), self._new_spec(vorm)
        check(self._lib.spd_vort2vel(self._h, _ptr(vorm), _ptr(divm), _ptr(ucosm), _ptr(vcosm), self._count(vorm),
                                     _stream_ptr()), "spd_vort2vel")
        return ucosm, vcosm

    def vel2vort(self, ucosm, vcosm):
        ucosm, vcosm = self._spec(ucosm, "ucosm"), self._spec(vcosm, "vcosm")
        vorm, divm = self._new_spec(ucosm), self._new_spec(ucosm)
        check(self._lib.spd_vel2vort(self._h, _ptr(ucosm), _ptr(vcosm), _ptr(vorm), _ptr(divm), self._count(ucosm),
                                     _stream_ptr()), "spd_vel2vort")
        return vorm, divm

    def grid_vel2vort(self, ug, vg, kcos):
        ug, vg = self._grid(ug, "ug"), self._grid(vg, "vg")
        vorm, divm = self._new_spec(ug), self._new_spec(ug)
        check(self._lib.spd_grid_vel2vort(self._h, _ptr(ug), _ptr(vg), _ptr(vorm), _ptr(divm), int(kcos),
                                          self._count(ug), _stream_ptr()), "spd_grid_vel2vort")
        return vorm, divm

    def gradient(self, psi):
        psi = self._spec(psi, "psi")
        psdx, psdy = self._new_spec(psi), self._new_spec(psi)
        check(self._lib.spd_gradient(self._h, _ptr(psi), _ptr(psdx), _ptr(psdy), self._count(psi), _stream_ptr()),
              "spd_gradient")
        return psdx, psdy

    def laplacian(self, x):
        x = self._spec(x, "input")
        out = self._new_spec(x)
        check(self._lib.spd_laplacian(self._h, _ptr(x), _ptr(out), 0, self._count(x), _stream_ptr()), "spd_laplacian")
        return out

    def laplacian_inv(self, x):
        x = self._spec(x, "input")
        out = self._new_spec(x)
        check(self._lib.spd_laplacian(self._h, _ptr(x), _ptr(out), 1, self._count(x), _stream_ptr()), "spd_laplacian")
        return out

    def truncate(self, vor):
        """In place, like the reference (spectral.f90:134-138).  Returns its argument."""
        if not vor.is_contiguous():
            raise ValueError("truncate works in place and needs a contiguous tensor")
        vor = self._spec(vor, "vor")
        check(self._lib.spd_truncate(self._h, _ptr(vor), self._count(vor), _stream_ptr()), "spd_truncate")
        return vor

    def grid_filter(self, fg1):
        fg1 = self._grid(fg1, "fg1")
        fg2 = self._new_grid(fg1)
        check(self._lib.spd_grid_filter(self._h, _ptr(fg1), _ptr(fg2), self._count(fg1), _stream_ptr()),
              "spd_grid_filter")
        return fg2


# ---- layout helpers between the reference's host arrays and the device layout ----------------------------
def spec_from_ref(a):
    """numpy complex (..., mx=31, nx=32) as the reference's getters return it -> [..., 32, 31] C-contiguous."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(a, dtype=np.complex128), -1, -2))


def spec_to_ref(a):
    return np.swapaxes(np.asarray(a), -1, -2)


def grid_from_ref(a):
    """numpy float (..., ix=96, il=48) -> [..., 48, 96] C-contiguous."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(a, dtype=np.float64), -1, -2))


def grid_to_ref(a):
    return np.swapaxes(np.asarray(a), -1, -2)
