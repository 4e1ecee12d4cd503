// Compile-time small FFTs for the two-level longitude transform (fft.hip): every twiddle a constexpr constant, every loop
// unrolled by template recursion, trivial twiddles (1, -1, +-i) folded into adds.  Round 2 evaluated both levels as direct
// O(N^2) sums (640 packed FMAs per thread at W = 360 = 20 x 18) and the kernels turned out bound by vector-ALU issue (r02 PMC:
// 78 % of the SIMD issue cycles); the factorised forms below need about a third of that.
//   cfft<N, INV>      complex, all N outputs; mixed radix, largest prime factor first (so that the radix-2 leaves are add / sub)
//   rfft<N>           real input, outputs k = 0 .. N / 2 (Hermitian half); radix 2 down to an odd length, direct there
// Plain C++ over a 2-vector type V with .x / .y (float2-like: clang ext_vector_type(2) on the device, so that complex
// multiply-adds become v_pk_fma_f32), also compiled for the host by tests/emul/small_fft_emul.cpp.
#pragma once

#include <type_traits>

namespace ace {
namespace sfft {

#if defined(__HIPCC__)
#define SFFT_FN __device__ __forceinline__
#else
#define SFFT_FN inline
#endif

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double c_sin_small(double x) {  // |x| <= pi / 4
    double term = x, sum = x;
    for (int n = 1; n < 12; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double c_cos_small(double x) {
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 12; ++n) {
        term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
struct cdbl { double re, im; };
// e^(-2 pi i j / N), exact on the axes
constexpr cdbl unit_root(long j, long N) {
    j %= N;
    if (j < 0) j += N;
    if (j == 0) return {1.0, 0.0};
    if (2 * j == N) return {-1.0, 0.0};
    if (4 * j == N) return {0.0, -1.0};
    if (4 * j == 3 * N) return {0.0, 1.0};
    const long q = (8 * j + N) / (2 * N);   // quadrant q = round(4 j / N), residual angle in [-pi/4, pi/4]
    const double r = 2.0 * kPi * ((double)j / (double)N - 0.25 * (double)q);
    const double c = c_cos_small(r), s = c_sin_small(r);
    double co = 0, si = 0;
    switch (q & 3) {
        case 0: co = c; si = s; break;
        case 1: co = -s; si = c; break;
        case 2: co = -c; si = -s; break;
        default: co = s; si = -c; break;
    }
    return {co, -si};
}

constexpr int largest_prime_factor(int n) {
    int best = 1;
    for (int p = 2; p <= n; ++p)
        while (n % p == 0) { best = p; n /= p; }
    return best;
}

template <int I0, int I1, class F>
SFFT_FN void unroll(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        unroll<I0 + 1, I1>(f);
    }
}

// acc += x * w for the compile-time root w = w_N^J (conjugated for the inverse transform); trivial roots become adds
template <int J, int N, bool INV, class V>
SFFT_FN void cmac(V& acc, const V x) {
    constexpr cdbl w0 = unit_root(J, N);
    constexpr float wr = (float)w0.re, wi = (float)(INV ? -w0.im : w0.im);
    if constexpr (wr == 1.0f && wi == 0.0f) acc = acc + x;
    else if constexpr (wr == -1.0f && wi == 0.0f) acc = acc - x;
    else if constexpr (wr == 0.0f && wi == 1.0f) acc = acc + V{-x.y, x.x};      // * i
    else if constexpr (wr == 0.0f && wi == -1.0f) acc = acc + V{x.y, -x.x};     // * -i
    else {
        acc = acc + V{x.x, x.x} * V{wr, wi};
        acc = acc + V{x.y, x.y} * V{-wi, wr};
    }
}
// acc += x * w for a REAL x
template <int J, int N, class V>
SFFT_FN void rmac(V& acc, const float x) {
    constexpr cdbl w0 = unit_root(J, N);
    acc = acc + V{x, x} * V{(float)w0.re, (float)w0.im};
}

// ---- complex FFT, all N outputs: out(k, sum_j in(j) w_N^(+-jk)).  Inputs are fetched once each (in(j) may be a load), outputs
// are handed over as they complete (out may store them): neither array needs to be live as a whole.
template <int N, bool INV, class V>
struct CFft {
    template <class In, class Out>
    SFFT_FN static void run(In&& in, Out&& out) {   // in(j) -> V, out(k, V)
        if constexpr (N == 1) {
            out(0, in(0));
        } else if constexpr (largest_prime_factor(N) == N) {   // prime length: direct
            V x[N];
            unroll<0, N>([&](auto jc) { x[decltype(jc)::value] = in(decltype(jc)::value); });
            unroll<0, N>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                V acc = x[0];
                unroll<1, N>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    cmac<(j * k) % N, N, INV>(acc, x[j]);
                });
                out(k, acc);
            });
        } else {
            constexpr int P = largest_prime_factor(N), M = N / P;
            V sub[P][M];
            unroll<0, P>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                CFft<M, INV, V>::run([&](int j) { return in(P * j + c); }, [&](int k, V v) { sub[c][k] = v; });
            });
            unroll<0, N>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                V acc = sub[0][k % M];
                unroll<1, P>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    cmac<(c * k) % N, N, INV>(acc, sub[c][k % M]);
                });
                out(k, acc);
            });
        }
    }
};

// ---- real-input FFT, outputs k = 0 .. N / 2
template <int N, class V>
struct RFft {
    static constexpr int NOUT = N / 2 + 1;
    template <class In, class Out>
    SFFT_FN static void run(In&& in, Out&& out) {   // in(j) -> float, out(k, V)
        if constexpr (N == 1) {
            out(0, V{in(0), 0.f});
        } else if constexpr (N % 2 != 0) {        // odd length: direct, Hermitian half
            float x[N];
            unroll<0, N>([&](auto jc) { x[decltype(jc)::value] = in(decltype(jc)::value); });
            unroll<0, NOUT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                V acc = V{x[0], 0.f};
                unroll<1, N>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    rmac<(j * k) % N, N>(acc, x[j]);
                });
                out(k, acc);
            });
        } else {
            constexpr int M = N / 2, MO = M / 2 + 1;
            V e[MO], o[MO];
            RFft<M, V>::run([&](int j) { return in(2 * j); }, [&](int k, V v) { e[k] = v; });
            RFft<M, V>::run([&](int j) { return in(2 * j + 1); }, [&](int k, V v) { o[k] = v; });
            unroll<0, NOUT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int km = k % M;                       // sub-transform index; beyond M / 2 use E[M - km] = conj E[km]
                constexpr bool mir = km > M / 2;
                constexpr int kk = mir ? M - km : km;
                const V ev = mir ? V{e[kk].x, -e[kk].y} : e[kk];
                const V ov = mir ? V{o[kk].x, -o[kk].y} : o[kk];
                V acc = ev;
                cmac<k % N, N, false>(acc, ov);
                out(k, acc);
            });
        }
    }
};

}  // namespace sfft
}  // namespace ace
