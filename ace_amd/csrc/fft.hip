// Longitude real DFT as a two-level Cooley-Tukey transform on the vector ALUs (gfx950).
//
// The matrix form (dft_forward_kernel / dft_inverse_kernel in kernels.hip) spends 2 * (W/2+1)^2 multiply-adds per row
// on the fp32 MFMA pipe (64 flops/cycle/SIMD); for the 1-degree grid (W = 360) that is 65 k MACs per row and the
// kernels sit on the fp32 matrix rate.  With W = N1 * N2, n = N2 a + b, k = k1 + N1 k2:
//     Y[b][k1] = sum_a x[N2 a + b] w_N1^(a k1)              N1-point DFTs of real data (k1 <= N1/2 by symmetry)
//     Z[b][k1] = Y[b][k1] w_W^(b k1)                        twiddle
//     X[k1 + N1 k2] = sum_b Z[b][k1] w_N2^(b k2)            N2-point DFTs
// costs ~21 k fp32 FMAs per row (360 = 20 x 18) with every small-DFT root a compile-time constant, no MFMA, and the
// kernel becomes a streaming pass over the field.  Semantics are those of the matrix kernels (fme/fft.py:61-96 under
// sht_fix's 2 pi scaling): forward X[m] = (2 pi / W) sum_w x[w] e^(-2 pi i m w / W) for m < Mm, with the fused
// instance-norm affine on x.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace ace {
namespace {

#define FDEV __device__ __forceinline__

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- compile-time roots of unity, exact on the axes -------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double c_sin_small(double x) {  // |x| <= pi/4
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
// e^(-2 pi i j / N)
constexpr cdbl unit_root(long j, long N) {
    j %= N;
    if (j < 0) j += N;
    if (j == 0) return {1.0, 0.0};
    if (2 * j == N) return {-1.0, 0.0};
    if (4 * j == N) return {0.0, -1.0};
    if (4 * j == 3 * N) return {0.0, 1.0};
    // quadrant q = round(4 j / N), residual angle in [-pi/4, pi/4]
    const long q = (8 * j + N) / (2 * N);
    const double r = 2.0 * kPi * ((double)j / (double)N - 0.25 * (double)q);
    const double c = c_cos_small(r), s = c_sin_small(r);
    double co = 0, si = 0;  // cos / sin of the full angle
    switch (q & 3) {
        case 0: co = c; si = s; break;
        case 1: co = -s; si = c; break;
        case 2: co = -c; si = -s; break;
        default: co = s; si = -c; break;
    }
    return {co, -si};
}
template <int N>
struct RootTab {  // w_N^j, j = 0 .. N-1, optionally scaled
    float re[N], im[N];
    constexpr explicit RootTab(double scale = 1.0) : re(), im() {
        for (int j = 0; j < N; ++j) {
            const cdbl w = unit_root(j, N);
            re[j] = (float)(scale * w.re);
            im[j] = (float)(scale * w.im);
        }
    }
};

template <int W>
__device__ constexpr RootTab<W> kScaledRoots{2.0 * kPi / (double)W};   // (2 pi / W) w_W^j: runtime-indexed, lives in device memory

#ifndef ACE_FFT_ABL
#define ACE_FFT_ABL 0   // measurement only: 1 no stores, 2 no second-level DFT, 3 no global loads, 4 no first-level DFT
#endif
#ifndef ACE_FFT_ROWS
#define ACE_FFT_ROWS 16
#endif
constexpr int FFT_ROWS = ACE_FFT_ROWS;  // channel rows per workgroup: 64-byte runs of the spectral output

// ---- forward ----------------------------------------------------------------------------------------------------------
// grid = (ceil(C / 16), H, Bt); block = 16 * max(N1, N2).  LDS: the rows (pitch W + 1), then - aliased - Z.
template <int N1, int N2>
__global__ __launch_bounds__(FFT_ROWS * (N1 > N2 ? N1 : N2)) void dft_forward_fft_kernel(DftArgs p) {
    constexpr int W = N1 * N2, R = FFT_ROWS, H1 = N1 / 2 + 1, NT = R * (N1 > N2 ? N1 : N2);
    constexpr int PITCH = W + 1;
    constexpr int K2N = N2 / 2 + 1;  // k = k1 + N1 k2 <= W / 2  =>  k2 <= N2 / 2
    static_assert(N1 % 2 == 0, "N1 even");
    constexpr int XS = R * PITCH, ZS = 2 * R * H1 * N2;
    __shared__ __attribute__((aligned(16))) float smem[XS > ZS ? XS : ZS];
    float* xs = smem;
    v2f* Zs = reinterpret_cast<v2f*>(smem);

    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * R, k = blockIdx.y, b = blockIdx.z;
    const int kb = k * p.Bt + b;
    const long HW = (long)p.H * W;

    // ---- rows -> LDS (16 B per lane), fused instance-norm affine
    for (int idx = tid; idx < R * (W / 4); idx += NT) {
        const int r = idx / (W / 4), j = idx % (W / 4);
        int c = c0 + r;
        c = c < p.C ? c : p.C - 1;
        const long bc = (long)b * p.C + c;
#if ACE_FFT_ABL == 3
        const float4 v = make_float4(1.f + idx, 2.f, 3.f, 4.f);
#else
        const float4 v = *reinterpret_cast<const float4*>(p.x + bc * HW + (long)k * W + 4 * j);
#endif
        float* d = xs + r * PITCH + 4 * j;
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
        d[3] = v.w;
    }
    __syncthreads();

    // ---- step 1: thread (b1, r): N1-point DFT of x[N2 a + b1], outputs k1 = 0 .. N1/2, times w_W^(b1 k1) (2 pi / W folded in)
    {
        const int r = tid % R, b1 = tid / R;
        const bool act = b1 < N2;
        // fused instance-norm affine of this thread's row (one load pair per thread, applied in registers)
        const int cr = c0 + r < p.C ? c0 + r : p.C - 1;
        const float sc = p.sc ? p.sc[(long)b * p.C + cr] : 1.f, sh = p.sc ? p.sh[(long)b * p.C + cr] : 0.f;
        float xv[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) xv[a] = act ? fmaf(xs[r * PITCH + N2 * a + b1], sc, sh) : 0.f;
        __syncthreads();   // Z aliases the rows: every row value is in registers before any Z is written
        if (act) {
            constexpr RootTab<N1> T1{};
#pragma unroll
            for (int k1 = 0; k1 < H1; ++k1) {
                v2f acc = {0.f, 0.f};
#pragma unroll
                for (int a = 0; a < (ACE_FFT_ABL == 4 ? 2 : N1); ++a) {
                    const int j = (a * k1) % N1;
                    const v2f w = {T1.re[j], T1.im[j]};
                    acc += v2f{xv[a], xv[a]} * w;
                }
                const int jw = b1 * k1;   // < N2 * (N1 / 2 + 1) <= W for N1 >= 2
                const float wr = kScaledRoots<W>.re[jw], wi = kScaledRoots<W>.im[jw];
                Zs[(r * H1 + k1) * N2 + b1] = v2f{acc.x * wr - acc.y * wi, acc.x * wi + acc.y * wr};
            }
        }
    }
    __syncthreads();

    // ---- step 2: thread (k1, r): N2-point DFT over b of Z[b][k1]; k1 > N1/2 from the Hermitian symmetry of Y:
    //      Z[b][k1] = conj(Z[b][N1 - k1]) w_N2^b
    float vmax = 0.f;
    {
        const int r = tid % R, k1 = tid / R;
        if (k1 < N1) {
            constexpr RootTab<N2> T2{};
            const bool cj = k1 > N1 / 2;
            const int k1p = cj ? N1 - k1 : k1;
            v2f z[N2];
#pragma unroll
            for (int bb = 0; bb < N2; ++bb) {
                const v2f t = Zs[(r * H1 + k1p) * N2 + bb];
                const v2f tc = {t.x * T2.re[bb] + t.y * T2.im[bb], t.x * T2.im[bb] - t.y * T2.re[bb]};   // conj(t) * w
                z[bb] = cj ? tc : t;
            }
            const int c = c0 + r;
            const long N2c = (long)p.Bt * 2 * p.C;
            float* ob = p.spec_out + (long)kb * 2 * p.C + c;
#pragma unroll
            for (int k2 = 0; k2 < K2N; ++k2) {
                const int m = k1 + N1 * k2;
                if (m < p.Mm) {
                    v2f acc = {0.f, 0.f};
#pragma unroll
                    for (int bb = 0; bb < (ACE_FFT_ABL == 2 ? 2 : N2); ++bb) {
                        const int j = (bb * k2) % N2;
                        const v2f w = {T2.re[j], T2.im[j]};
                        const v2f wp = {-T2.im[j], T2.re[j]};
                        acc += v2f{z[bb].x, z[bb].x} * w;
                        acc += v2f{z[bb].y, z[bb].y} * wp;
                    }
                    if (c < p.C && (ACE_FFT_ABL != 1 || acc.x == 1.2345e-30f)) {
                        float* o = ob + (long)m * p.H * N2c;
                        o[0] = acc.x;
                        o[p.C] = acc.y;
                        vmax = fmaxf(vmax, fmaxf(fabsf(acc.x), fabsf(acc.y)));
                    }
                }
            }
        }
    }
    if (p.omax) {   // one atomic per workgroup (Z is dead: reduce the wave maxima through LDS)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads();
        if ((tid & 63) == 0) smem[tid >> 6] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < (NT + 63) / 64; ++w) m = fmaxf(m, smem[w]);
            atomicMax(p.omax + ((blockIdx.x + blockIdx.y) & 63), __float_as_uint(m));
        }
    }
}

template <int N1, int N2>
hipError_t launch_fwd(const DftArgs& a, hipStream_t s) {
    constexpr int NT = FFT_ROWS * (N1 > N2 ? N1 : N2);
    dim3 grid((unsigned)((a.C + FFT_ROWS - 1) / FFT_ROWS), (unsigned)a.H, (unsigned)a.Bt), block(NT);
    hipLaunchKernelGGL((dft_forward_fft_kernel<N1, N2>), grid, block, 0, s, a);
    return hipGetLastError();
}

bool fft_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("ACE_NO_FFT");
        return !(e && e[0] && e[0] != '0');
    }();
    return on;
}

}  // namespace

// true when the FFT form handled the launch (sizes with an instantiated factorisation, 16-byte aligned rows)
bool launch_dft_forward_fft(const DftArgs& a, hipStream_t s, hipError_t* err) {
    if (!fft_enabled() || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0 || a.Mm > a.W / 2 + 1 || a.H > 65535 || a.Bt > 65535)
        return false;
    switch (a.W) {
        case 360: *err = launch_fwd<20, 18>(a, s); return true;
        case 48: *err = launch_fwd<8, 6>(a, s); return true;
        case 24: *err = launch_fwd<6, 4>(a, s); return true;
        case 16: *err = launch_fwd<4, 4>(a, s); return true;
        default: return false;
    }
}

}  // namespace ace
