"""Quadrature rules on [-1, 1] (oracle; test infrastructure only).

Restates ``torch_harmonics.quadrature`` (torch-harmonics 0.8.0, pinned at
``pyproject.toml:41``; not vendored) as called from ``fme/sht_fix.py:87-107``.
In-tree near-copies followed here: ``fme/core/disco/_quadrature.py:25-71``
(legendre-gauss, clenshaw-curtiss).  Lobatto is not vendored; it is the
published Gauss-Lobatto construction (Newton iteration from Chebyshev-Lobatto
nodes, w = 2 / (n (n-1) P_{n-1}(x)^2)) and is pinned by the reference golden
``fme/core/benchmark/testdata/sht-regression.pt``.

All results are fp64 numpy arrays: (nodes ascending in cos(theta), weights).
"""

import numpy as np


def legendre_gauss_weights(n: int, a: float = -1.0, b: float = 1.0):
    """fme/core/disco/_quadrature.py:25-33 (numpy leggauss, affine map)."""
    xlg, wlg = np.polynomial.legendre.leggauss(n)
    xlg = (b - a) * 0.5 * xlg + (b + a) * 0.5
    wlg = wlg * (b - a) * 0.5
    return xlg.astype(np.float64), wlg.astype(np.float64)


def lobatto_weights(n: int, a: float = -1.0, b: float = 1.0, tol=1e-16, maxiter=100):
    """Gauss-Lobatto nodes/weights (torch-harmonics 0.8.0 ``lobatto_weights``).

    Newton iteration on the Chebyshev-Gauss-Lobatto first guess using the
    Legendre Vandermonde recurrence; weights 2 / (n (n-1) P_{n-1}(x)^2).
    """
    tlg = -np.cos(np.pi * np.arange(n) / (n - 1))
    vdm = np.zeros((n, n), dtype=np.float64)
    for _ in range(maxiter):
        tmp = tlg
        vdm[:, 0] = 1.0
        vdm[:, 1] = tlg
        for k in range(2, n):
            vdm[:, k] = ((2 * k - 1) * tlg * vdm[:, k - 1] - (k - 1) * vdm[:, k - 2]) / k
        tlg = tmp - (tlg * vdm[:, n - 1] - vdm[:, n - 2]) / (n * vdm[:, n - 1])
        if np.max(np.abs(tlg - tmp)) < tol:
            break
    wlg = 2.0 / ((n * (n - 1)) * (vdm[:, n - 1] ** 2))
    tlg = (b - a) * 0.5 * tlg + (b + a) * 0.5
    wlg = wlg * (b - a) * 0.5
    return tlg, wlg


def clenshaw_curtiss_weights(n: int, a: float = -1.0, b: float = 1.0):
    """fme/core/disco/_quadrature.py:36-71 (FFT-based Clenshaw-Curtis)."""
    assert n > 1
    tcc = np.cos(np.linspace(np.pi, 0, n, dtype=np.float64))
    if n == 2:
        wcc = np.array([1.0, 1.0], dtype=np.float64)
    else:
        n1 = n - 1
        N = np.arange(1, n1, 2, dtype=np.float64)
        ll = len(N)
        m = n1 - ll
        v = np.concatenate([2 / N / (N - 2), 1 / N[-1:], np.zeros(m)])
        v = 0 - v[:-1] - v[-1:0:-1]
        g0 = -np.ones(n1)
        g0[ll] = g0[ll] + n1
        g0[m] = g0[m] + n1
        g = g0 / (n1**2 - 1 + (n1 % 2))
        wcc = np.fft.ifft(v + g).real
        wcc = np.concatenate((wcc, wcc[:1]))
    tcc = (b - a) * 0.5 * tcc + (b + a) * 0.5
    wcc = wcc * (b - a) * 0.5
    return tcc, wcc


def quadrature(grid: str, nlat: int):
    """Dispatch used by RealSHT/InverseRealSHT (fme/sht_fix.py:86-102)."""
    if grid == "legendre-gauss":
        return legendre_gauss_weights(nlat, -1, 1)
    if grid == "lobatto":
        return lobatto_weights(nlat, -1, 1)
    if grid == "equiangular":
        return clenshaw_curtiss_weights(nlat, -1, 1)
    if grid == "healpix":
        raise NotImplementedError("'healpix' grid not supported")
    raise ValueError("Unknown quadrature mode")
