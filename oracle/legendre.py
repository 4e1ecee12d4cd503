"""Orthonormal associated Legendre table (oracle; test infrastructure only).

Restates ``torch_harmonics.legendre._precompute_legpoly`` (torch-harmonics
0.8.0, not vendored) as called at ``fme/sht_fix.py:110,189``.  The in-tree
near-copy is ``fme/core/cuhpx/tools.py:288-336`` - followed here for the
recursion, EXCEPT its Condon-Shortley line (``vdm[m] *= 1`` at ``:334``): the
reference goldens require odd m multiplied by -1.
"""

import numpy as np


def legpoly(mmax, lmax, x, norm="ortho", inverse=False, csphase=True):
    """P-bar_l^m(x) as fp64 array [mmax, lmax, len(x)]; zero for l < m."""
    x = np.asarray(x, dtype=np.float64)
    nmax = max(mmax, lmax)
    vdm = np.zeros((nmax, nmax, len(x)), dtype=np.float64)

    norm_factor = 1.0 if norm == "ortho" else np.sqrt(4 * np.pi)
    norm_factor = 1.0 / norm_factor if inverse else norm_factor

    vdm[0, 0, :] = norm_factor / np.sqrt(4 * np.pi)

    # diagonal and first off-diagonal (cuhpx/tools.py:299-304)
    for l in range(1, nmax):
        vdm[l - 1, l, :] = np.sqrt(2 * l + 1) * x * vdm[l - 1, l - 1, :]
        vdm[l, l, :] = (
            np.sqrt((2 * l + 1) * (1 + x) * (1 - x) / 2 / l) * vdm[l - 1, l - 1, :]
        )

    # three-term recursion in l, vectorised over m (cuhpx/tools.py:306-322)
    for l in range(2, nmax):
        m = np.arange(0, l - 1, dtype=np.float64)[:, None]
        f1 = np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m))
        f2 = np.sqrt(
            (l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m)
        )
        vdm[: l - 1, l, :] = x[None, :] * f1 * vdm[: l - 1, l - 1, :] - f2 * vdm[: l - 1, l - 2, :]

    if norm == "schmidt":
        for l in range(0, nmax):
            if inverse:
                vdm[:, l, :] = vdm[:, l, :] * np.sqrt(2 * l + 1)
            else:
                vdm[:, l, :] = vdm[:, l, :] / np.sqrt(2 * l + 1)

    vdm = vdm[:mmax, :lmax]

    if csphase:
        vdm[1::2] *= -1.0

    return vdm


_LAST = {}   # the most recent table (the 721-latitude one takes most of a minute and several tests build it twice: analysis + synthesis)


def precompute_legpoly(mmax, lmax, t, norm="ortho", inverse=False, csphase=True):
    """cuhpx/tools.py:374-375: table on colatitudes t.  The returned array is read-only (one shared copy per geometry)."""
    t = np.asarray(t, dtype=np.float64)
    # with norm="ortho" the `inverse` flag changes nothing (norm_factor = 1 both ways)
    key = (mmax, lmax, t.tobytes(), norm, bool(inverse) and norm != "ortho", csphase)
    if _LAST.get("key") != key:
        _LAST.clear()
        table = legpoly(mmax, lmax, np.cos(t), norm=norm, inverse=inverse, csphase=csphase)
        table.setflags(write=False)
        _LAST.update(key=key, table=table)
    return _LAST["table"]
