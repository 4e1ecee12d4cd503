#!/usr/bin/env python
"""Derivation of the fast erf used by the f16x3 GEMM epilogue (ace_amd/csrc/kernels.hip: fast_erf).
erf(x) = sign(x) * (1 - 2^-q(|x|)); q = weighted least-squares degree-9 fit of -log2(erfc(t)) on [0, 4] at Chebyshev
nodes (weight erfc(t): the abs error of erf is erfc * ln2 * dq).  Prints float32 coefficients and the max abs error of
the float32 evaluation against scipy's erf in fp64."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc

T = 4.0


def fit(deg=9, npts=40001):
    t = np.cos(np.pi * (np.arange(npts) + 0.5) / npts) * T / 2 + T / 2
    q, w = -np.log2(erfc(t)), erfc(t) + 1e-9
    V = C.chebvander(2 * t / T - 1, deg)
    coef, *_ = np.linalg.lstsq(V * w[:, None], q * w, rcond=None)
    pc, pt, base, cur = C.cheb2poly(coef), np.zeros(1), np.array([-1.0, 2.0 / T]), np.ones(1)
    for c in pc:
        pt, cur = P.polyadd(pt, c * cur), P.polymul(cur, base)
    pt[0] = 0.0
    return pt.astype(np.float32)


if __name__ == "__main__":
    c = fit()
    print(", ".join("%.9ef" % v for v in c))
    xs = np.linspace(-6, 6, 4000001)
    t = np.minimum(np.abs(xs), T).astype(np.float32)
    r = np.full_like(t, c[-1])
    for k in range(len(c) - 2, -1, -1):
        r = (r.astype(np.float64) * t.astype(np.float64) + np.float64(c[k])).astype(np.float32)
    y = np.copysign((np.float32(1) - np.exp2(-r.astype(np.float64)).astype(np.float32)).astype(np.float32), xs)
    print("max abs err", np.abs(y - erf(xs)).max())
    # --gelu: the coefficients of strip_common.h's gelu_stage1: q as a function of |v| = sqrt 2 t (c_k / 2^(k / 2)), and the error
    # of GELU(v) = v / 2 (1 + erf(v / sqrt 2)) evaluated that way in float32
    import sys
    if "--gelu" in sys.argv:
        d = np.array([np.float32(float(c[k]) * 2.0 ** (-k / 2.0)) for k in range(len(c))])
        print("gelu:", ", ".join("%.9ef" % v for v in d[:0:-1]))
        v = np.linspace(-8, 8, 2000001)
        t = np.minimum(np.abs(v), T * np.sqrt(2.0)).astype(np.float32)
        r = np.full_like(t, d[-1])
        for k in range(len(d) - 2, 0, -1):
            r = (r * t + d[k]).astype(np.float32)
        r = (r * t).astype(np.float32)
        # gelu_stage2: (v + |v| - t 2^-q) / 2, the multiply-add rounded once
        v32 = v.astype(np.float32)
        s = (v32 + np.abs(v32)).astype(np.float32)
        inner = (s.astype(np.float64) - t.astype(np.float64) * np.exp2(-r).astype(np.float32).astype(np.float64)).astype(np.float32)
        print("gelu max abs err", np.abs((inner * np.float32(0.5)).astype(np.float64) - 0.5 * v * (1 + erf(v / np.sqrt(2.0)))).max())
