"""RealSHT / InverseRealSHT with the reference constructor and call signature
(fme/sht_fix.py:61-139, 141-226), computed by the HIP library: folded longitude DFT
and triangular Legendre quadrature on the fp32 MFMA tile engine."""

import ctypes
import os

import torch
import torch.nn as nn

from . import _lib


class _Plan:
    """Owns an ace_sht_plan (tables live on the current device)."""

    def __init__(self, nlat, nlon, lmax, mmax, grid, precision=None):
        handle = ctypes.c_void_p()
        # arithmetic of the Legendre stage: "fp32" (exact, the reference's) or "f16x3" (the network's default mode);
        # not part of the reference API, selected by ACE_SHT_PRECISION or the module attribute `precision`
        precision = precision or os.environ.get("ACE_SHT_PRECISION", "fp32")
        if precision not in ("fp32", "f16x3"):
            raise ValueError("precision must be 'fp32' or 'f16x3'")
        _lib.check(_lib.lib().ace_sht_plan_create_ex(nlat, nlon, lmax or 0, mmax or 0, grid.encode(),
                                                     1 if precision == "f16x3" else 0, ctypes.byref(handle)))
        self.handle = handle
        d = [ctypes.c_int() for _ in range(4)]
        _lib.check(_lib.lib().ace_sht_plan_dims(handle, *[ctypes.byref(v) for v in d]))
        self.nlat, self.nlon, self.lmax, self.mmax = (v.value for v in d)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().ace_sht_plan_destroy(self.handle)
        except Exception:
            pass


def _default_dims(nlat, nlon, lmax, mmax, grid):
    # fme/sht_fix.py:86-104
    if grid == "healpix":
        raise NotImplementedError("'healpix' grid not supported")
    if grid not in ("legendre-gauss", "lobatto", "equiangular"):
        raise ValueError("Unknown quadrature mode")
    lmax = lmax or (nlat - 1 if grid == "lobatto" else nlat)
    mmax = mmax or nlon // 2 + 1
    return lmax, mmax


class RealSHT(nn.Module):
    """Forward real SHT over the last two dimensions: (..., nlat, nlon) -> (..., lmax, mmax) complex64."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="lobatto", norm="ortho", csphase=True, precision=None):
        super().__init__()
        self.precision = precision   # None: ACE_SHT_PRECISION or "fp32"; "f16x3": the network's default arithmetic
        if norm != "ortho" or not csphase:
            raise NotImplementedError("only norm='ortho', csphase=True (the reference's usage) is implemented")
        self.nlat, self.nlon, self.grid, self.norm, self.csphase = nlat, nlon, grid, norm, csphase
        self.lmax, self.mmax = _default_dims(nlat, nlon, lmax, mmax, grid)
        self._plan = None  # tables are plain attributes, not buffers (fme/sht_fix.py:111)

    def _get_plan(self):
        if self._plan is None:
            self._plan = _Plan(self.nlat, self.nlon, self.lmax, self.mmax, self.grid, getattr(self, "precision", None))
        return self._plan

    def extra_repr(self):
        return f"nlat={self.nlat}, nlon={self.nlon},\n lmax={self.lmax}, mmax={self.mmax},\n grid={self.grid}, csphase={self.csphase}"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.nlat
        assert x.shape[-1] == self.nlon
        x = x.float().contiguous()
        lead = x.shape[:-2]
        n = 1
        for s in lead:
            n *= s
        out = torch.empty(*lead, self.lmax, self.mmax, dtype=torch.complex64, device=x.device)
        if n > 0:
            _lib.check(_lib.lib().ace_sht_forward(self._get_plan().handle, _lib.ptr(x), _lib.ptr(out), n,
                                                  _lib.current_stream()))
        return out


class InverseRealSHT(nn.Module):
    """Inverse real SHT: (..., lmax, mmax) complex64 -> (..., nlat, nlon)."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="lobatto", norm="ortho", csphase=True, precision=None):
        super().__init__()
        self.precision = precision   # None: ACE_SHT_PRECISION or "fp32"; "f16x3": the network's default arithmetic
        if norm != "ortho" or not csphase:
            raise NotImplementedError("only norm='ortho', csphase=True (the reference's usage) is implemented")
        self.nlat, self.nlon, self.grid, self.norm, self.csphase = nlat, nlon, grid, norm, csphase
        self.lmax, self.mmax = _default_dims(nlat, nlon, lmax, mmax, grid)
        self._plan = None

    def _get_plan(self):
        if self._plan is None:
            self._plan = _Plan(self.nlat, self.nlon, self.lmax, self.mmax, self.grid, getattr(self, "precision", None))
        return self._plan

    def extra_repr(self):
        return f"nlat={self.nlat}, nlon={self.nlon},\n lmax={self.lmax}, mmax={self.mmax},\n grid={self.grid}, csphase={self.csphase}"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.lmax
        assert x.shape[-1] == self.mmax
        x = x.to(torch.complex64).contiguous()
        lead = x.shape[:-2]
        n = 1
        for s in lead:
            n *= s
        out = torch.empty(*lead, self.nlat, self.nlon, dtype=torch.float32, device=x.device)
        if n > 0:
            _lib.check(_lib.lib().ace_sht_inverse(self._get_plan().handle, _lib.ptr(x), _lib.ptr(out), n,
                                                  _lib.current_stream()))
        return out
