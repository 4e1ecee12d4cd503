"""The index algebra of ace_amd/csrc/fft.hip (two-level Cooley-Tukey form of the longitude DFT), restated in numpy and
checked against the direct sums the matrix kernels / the reference compute (fme/fft.py:61-96 under sht_fix's 2 pi scaling):
level 2 as ONE full N2-point transform per column k1 <= N1/2 whose outputs are stored twice (q <= N2/2 directly, conjugated
as the mirrored column N1 - k1 at k2 = N2 - 1 - q), the inverse's per-k2 choice of stored entry (direct below N2/2, the
conjugate of column N1 - k1 of block N2 - 1 - k2 above, W/2 itself or mirrored at N2/2), the dropped imaginary parts at
m = 0 / Nyquist, the truncation at mmax and the paired (P, Q) outputs.  The small transforms themselves
(ace_amd/csrc/small_fft.h) are compiled for the host in test_strip_emul_cpu.py; the kernels are exercised by the GPU SHT /
network tests at W = 16, 24, 48 and 360."""
import numpy as np
import pytest

CASES = [(360, 20, 18), (1440, 40, 36), (720, 30, 24), (48, 8, 6), (24, 6, 4), (16, 4, 4)]   # the instantiated factorisations W = N1 * N2


def w(j, n):
    return np.exp(-2j * np.pi * (j % n) / n)


def forward_two_level(x, N1, N2, Mm):
    W = N1 * N2
    H1 = N1 // 2 + 1
    Z = np.zeros((N2, H1), complex)
    for b in range(N2):
        for k1 in range(H1):
            Z[b, k1] = sum(x[N2 * a + b] * w(a * k1, N1) for a in range(N1)) * (2 * np.pi / W) * w(b * k1, W)
    X = np.zeros(Mm, complex)
    written = np.zeros(Mm, int)
    for k1 in range(H1):
        out = [sum(Z[b, k1] * w(b * q, N2) for b in range(N2)) for q in range(N2)]     # all N2 outputs of column k1
        for q in range(N2):
            if q <= N2 // 2:
                m = k1 + N1 * q
                if m < Mm:
                    X[m] = out[q]
                    written[m] += 1
            if q >= N2 // 2 and 0 < k1 < N1 // 2:
                m = (N1 - k1) + N1 * (N2 - 1 - q)
                if m < Mm:
                    X[m] = np.conj(out[q])
                    written[m] += 1
    assert (written == 1).all()             # every stored wavenumber exactly once
    return X


def inverse_two_level(S, N1, N2, Mm):
    W = N1 * N2
    H1, M2 = N1 // 2 + 1, N2 // 2
    U = np.zeros((H1, N2), complex)
    for k1 in range(H1):
        f = np.zeros(N2, complex)
        for k2 in range(N2):
            mir = k2 > N2 // 2 or (k2 == N2 // 2 and k1 > 0)
            blk = (N2 - 1 - k2) if mir else k2
            m = ((N1 - k1) if mir else k1) + N1 * blk
            assert m == (W - (k1 + N1 * k2) if 2 * (k1 + N1 * k2) > W else k1 + N1 * k2)
            re, im = (S[m].real, S[m].imag) if m < Mm else (0.0, 0.0)
            if (k2 == 0 and k1 == 0) or (k2 == N2 // 2 and k1 == 0):
                im = 0.0
            f[k2] = complex(re, -im if mir else im)
        for j in range(M2):
            e = sum(f[2 * i] * np.conj(w(i * j, M2)) for i in range(M2))
            o = sum(f[2 * i + 1] * np.conj(w(i * j, M2)) for i in range(M2)) * np.conj(w(j, N2))
            U[k1, j] = (e + o) * np.conj(w(k1 * j, W))
            U[k1, j + M2] = (e - o) * np.conj(w(k1 * (j + M2), W))
    y = np.zeros(W)
    hN = N1 // 2
    for b in range(N2):
        u = U[:, b]
        y[b] = u[0].real + u[hN].real + 2 * sum(u[k].real for k in range(1, hN))
        y[N2 * hN + b] = u[0].real + (-1) ** hN * u[hN].real + sum((-2 if k % 2 else 2) * u[k].real for k in range(1, hN))
        for a in range(1, hN):
            P = u[0].real + (-1) ** a * u[hN].real + sum(2 * u[k].real * w(k * a, N1).real for k in range(1, hN))
            Q = sum(2 * u[k].imag * w(k * a, N1).imag for k in range(1, hN))
            y[N2 * a + b], y[N2 * (N1 - a) + b] = P + Q, P - Q
    return y


@pytest.mark.parametrize("W,N1,N2", CASES)
def test_forward_two_level_equals_scaled_rfft(W, N1, N2):
    rng = np.random.default_rng(W)
    x = rng.standard_normal(W)
    for Mm in (W // 2 + 1, W // 2 - 1, 3):
        got = forward_two_level(x, N1, N2, Mm)
        ref = 2 * np.pi * np.fft.rfft(x, norm="forward")[:Mm]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    assert forward_two_level(x, N1, N2, W // 2 + 1)[0].imag == 0.0


@pytest.mark.parametrize("W,N1,N2", CASES)
def test_inverse_two_level_equals_irfft_of_truncated_spectrum(W, N1, N2):
    rng = np.random.default_rng(W + 1)
    S = rng.standard_normal(W // 2 + 1) + 1j * rng.standard_normal(W // 2 + 1)
    for Mm in (W // 2 + 1, W // 2 - 1, 3):
        T = np.zeros(W // 2 + 1, complex)
        T[:Mm] = S[:Mm]
        T[0] = T[0].real                       # irfft ignores these imaginary parts (fft.py:78-96)
        T[-1] = T[-1].real
        ref = np.fft.irfft(T, n=W, norm="forward")
        got = inverse_two_level(S, N1, N2, Mm)
        assert np.abs(got - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
