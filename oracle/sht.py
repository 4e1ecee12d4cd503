"""Real spherical harmonic transform pair (oracle; test infrastructure only).

Restates ``fme/sht_fix.py:61-139`` (RealSHT), ``:141-226`` (InverseRealSHT) and
the rfft/irfft wrappers ``fme/fft.py:61-96`` with the same torch-CPU op
sequence (``torch.fft.rfft`` + ``einsum``), so it also serves as the timed CPU
baseline.  ``dtype=torch.float64`` gives a higher-precision "truth" used to
put both fp32 implementations' round-off in context.
"""

import numpy as np
import torch

from .legendre import precompute_legpoly
from .quadrature import quadrature


def _default_lmax(grid, nlat):
    # fme/sht_fix.py:87-96
    return nlat - 1 if grid == "lobatto" else nlat


def rfft(x, nmodes=None, dim=-1, **kwargs):
    """fme/fft.py:61-76."""
    x = torch.fft.rfft(x, dim=dim, **kwargs)
    if nmodes is not None and nmodes > x.shape[dim]:
        pad = [0] * (2 * x.ndim)
        d = dim if dim >= 0 else x.ndim + dim
        pad[(x.ndim - 1 - d) * 2 + 1] = nmodes - x.shape[dim]
        x = torch.nn.functional.pad(x, tuple(pad), value=0.0)
    elif nmodes is not None and nmodes < x.shape[dim]:
        x = x.narrow(dim, 0, nmodes)
    return x


def irfft(x, n=None, dim=-1, **kwargs):
    """fme/fft.py:78-96 (zero Im of m=0 and Nyquist, then irfft)."""
    if n is None:
        n = 2 * (x.size(dim) - 1)
    x = x.clone()
    x[..., 0].imag = 0.0
    if (n % 2 == 0) and (n // 2 < x.size(dim)):
        x[..., n // 2].imag = 0.0
    return torch.fft.irfft(x, n=n, dim=dim, **kwargs)


class RealSHT:
    """fme/sht_fix.py:61-139."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="lobatto", norm="ortho",
                 csphase=True, dtype=torch.float32):
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        cost, w = quadrature(grid, nlat)
        self.lmax = lmax or _default_lmax(grid, nlat)
        tq = np.flip(np.arccos(cost))
        self.mmax = mmax or nlon // 2 + 1
        pct = precompute_legpoly(self.mmax, self.lmax, tq, norm=norm, csphase=csphase)
        weights = np.einsum("mlk,k->mlk", pct, w)
        self.weights64 = torch.from_numpy(np.ascontiguousarray(weights))
        self.dtype = dtype
        self.weights = self.weights64.to(dtype)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.nlat
        assert x.shape[-1] == self.nlon
        x = x.to(self.dtype)
        x = 2.0 * torch.pi * rfft(x, nmodes=self.mmax, dim=-1, norm="forward")
        x = x.transpose(-2, -1).contiguous()
        x = torch.view_as_real(x)
        out_shape = list(x.size())
        out_shape[-3] = self.lmax
        out_shape[-2] = self.mmax
        xout = torch.zeros(out_shape, dtype=x.dtype)
        xout[..., 0] = torch.einsum("...mk,mlk->...lm", x[..., : self.mmax, :, 0], self.weights)
        xout[..., 1] = torch.einsum("...mk,mlk->...lm", x[..., : self.mmax, :, 1], self.weights)
        return torch.view_as_complex(xout)


class InverseRealSHT:
    """fme/sht_fix.py:141-226."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="lobatto", norm="ortho",
                 csphase=True, dtype=torch.float32):
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        cost, _ = quadrature(grid, nlat)
        self.lmax = lmax or _default_lmax(grid, nlat)
        t = np.flip(np.arccos(cost))
        self.mmax = mmax or nlon // 2 + 1
        pct = precompute_legpoly(self.mmax, self.lmax, t, norm=norm, inverse=True, csphase=csphase)
        self.pct64 = torch.from_numpy(np.array(pct, order="C"))   # own copy: the table is shared and read-only
        self.dtype = dtype
        self.pct = self.pct64.to(dtype)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.lmax
        assert x.shape[-1] == self.mmax
        x = x.transpose(-1, -2).contiguous()
        x = torch.view_as_real(x).to(self.dtype)
        rl = torch.einsum("...ml,mlk->...km", x[..., 0], self.pct)
        im = torch.einsum("...ml,mlk->...km", x[..., 1], self.pct)
        xs = torch.stack((rl, im), -1)
        x = torch.view_as_complex(xs)
        return irfft(x, n=self.nlon, dim=-1, norm="forward")
