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

// entries (complex values) per first-level index k1 of the intermediate Z / U: N2 * R, padded to R modulo 32
template <int N2, int R>
struct ZPitch {
    static constexpr int raw = N2 * R;
    static constexpr int value = R >= 32 ? raw : raw + ((R - raw) % 32 + 32) % 32;
};

// ---- forward ----------------------------------------------------------------------------------------------------------
// grid = (ceil(C / R), ceil(H / NLAT), Bt); block = R * max(N1, N2).  LDS: the rows, then - aliased - Z.
// NLAT > 1: a workgroup transforms NLAT consecutive latitudes and fetches the rows of the next one into registers while the
// two DFT levels of the current one run (the r02 ablation showed load, level 1, level 2 and store phases adding up).
// XT: the rows are staged transposed, xs[n][r] (r fastest): the strided reads of level 1 become contiguous runs.
#ifndef ACE_FFT_NLAT
#define ACE_FFT_NLAT 1
#endif
#ifndef ACE_FFT_XT
#define ACE_FFT_XT 0
#endif
#ifndef ACE_FFT_PFU
#define ACE_FFT_PFU 1
#endif
#ifndef ACE_FFT_MINWG
#define ACE_FFT_MINWG 0
#endif
// workgroups per CU the register allocation must allow (0: the compiler's choice); only applied to the 16-row shapes
template <int N1, int N2, int R, int NLAT, bool XT>
__global__ __launch_bounds__(R * (N1 > N2 ? N1 : N2), (R == 16 && N1 * N2 <= 360) ? ACE_FFT_MINWG : 0) void dft_forward_fft_kernel(DftArgs p) {
    constexpr int W = N1 * N2, H1 = N1 / 2 + 1, NT = R * (N1 > N2 ? N1 : N2);
    constexpr int PITCH = W + 1;
    constexpr int K2N = N2 / 2 + 1;  // k = k1 + N1 k2 <= W / 2  =>  k2 <= N2 / 2
    static_assert(N1 % 2 == 0, "N1 even");
    // Z[k1][b][r], r fastest: step 1 (thread (b, r), fixed k1) writes and step 2 (thread (k1, r), fixed b) reads whole
    // contiguous runs of R entries.  ZP = entries per k1, padded so that the 32 lanes of one ds_read_b64 group (32 / R
    // consecutive k1 x R rows) fall on 64 distinct banks: ZP = R (mod 32).  (r02: the [r][k1][b] order cost 53 % of the
    // kernel's LDS cycles in bank conflicts.)
    constexpr int ZP = ZPitch<N2, R>::value;
    constexpr int XS = XT ? W * R : R * PITCH, ZS = 2 * H1 * ZP;
    __shared__ __attribute__((aligned(16))) float smem[XS > ZS ? XS : ZS];
    float* xs = smem;
    v2f* Zs = reinterpret_cast<v2f*>(smem);

    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * R, klat0 = blockIdx.y * NLAT, b = blockIdx.z;
    const long HW = (long)p.H * W;
    constexpr int NPF = (R * (W / 4) + NT - 1) / NT;   // 16-byte row pieces per thread

    // piece idx of the workgroup: (row r, 16-byte column j).  XT: rows fastest, so that the four LDS writes of a lane pair up
    // at most two lanes per bank; otherwise columns fastest (128-byte runs of one row per 8 lanes)
    auto piece_rj = [&](int idx, int& r, int& j) {
        if (XT) { r = idx % R; j = idx / R; }
        else { r = idx / (W / 4); j = idx % (W / 4); }
    };
    float4 pf[NPF];
    auto fetch = [&](int k) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = tid + q * NT;
            if (idx < R * (W / 4)) {
                int r, j;
                piece_rj(idx, r, j);
                int c = c0 + r;
                c = c < p.C ? c : p.C - 1;
                const long bc = (long)b * p.C + c;
#if ACE_FFT_ABL == 3
                pf[q] = make_float4(1.f + idx, 2.f, 3.f, 4.f);
#else
                pf[q] = *reinterpret_cast<const float4*>(p.x + bc * HW + (long)k * W + 4 * j);
#endif
            }
        }
    };
    auto stage_rows = [&]() {   // registers -> LDS
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int idx = tid + q * NT;
            if (idx < R * (W / 4)) {
                int r, j;
                piece_rj(idx, r, j);
                const float4 v = pf[q];
                if (XT) {
                    float* d = xs + (4 * j) * R + r;
                    d[0] = v.x; d[R] = v.y; d[2 * R] = v.z; d[3 * R] = v.w;
                } else {
                    float* d = xs + r * PITCH + 4 * j;   // (8-way bank conflicts on these four writes - rotating the element
                    d[0] = v.x;                          //  order per lane group removes them and made the kernel 5 % SLOWER:
                    d[1] = v.y;                          //  not on the critical path, r02 same-box A/B)
                    d[2] = v.z;
                    d[3] = v.w;
                }
            }
        }
    };
    // fused instance-norm affine of this thread's row in step 1 (one load pair per thread, applied in registers)
    const int r1 = tid % R, b1 = tid / R;
    const int cr = c0 + r1 < p.C ? c0 + r1 : p.C - 1;
    const float sc = p.sc ? p.sc[(long)b * p.C + cr] : 1.f, sh = p.sc ? p.sh[(long)b * p.C + cr] : 0.f;

#if ACE_FFT_PFU
    fetch(klat0);        // all row pieces of this thread in flight at once
    stage_rows();
#else
    for (int idx = tid; idx < R * (W / 4); idx += NT) {   // r02 form: one piece at a time (fewest registers)
        int r, j;
        piece_rj(idx, r, j);
        int c = c0 + r;
        c = c < p.C ? c : p.C - 1;
        const float4 v = *reinterpret_cast<const float4*>(p.x + ((long)b * p.C + c) * HW + (long)klat0 * W + 4 * j);
        if (XT) {
            float* d = xs + (4 * j) * R + r;
            d[0] = v.x; d[R] = v.y; d[2 * R] = v.z; d[3 * R] = v.w;
        } else {
            float* d = xs + r * PITCH + 4 * j;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
#endif
    __syncthreads();

    float vmax = 0.f;
    auto one_latitude = [&](const int it) {
        const int k = klat0 + it;
        const bool more = NLAT > 1 && it + 1 < NLAT && k + 1 < p.H;
        if (more) fetch(k + 1);                    // in flight under both DFT levels
        const int kb = k * p.Bt + b;

        // ---- step 1: thread (b1, r): N1-point DFT of x[N2 a + b1], outputs k1 = 0 .. N1/2, times w_W^(b1 k1) (2 pi / W folded in)
        {
            const int r = r1;
            const bool act = b1 < N2;
            float xv[N1];
#pragma unroll
            for (int a = 0; a < N1; ++a)
                xv[a] = act ? fmaf(XT ? xs[(N2 * a + b1) * R + r] : xs[r * PITCH + N2 * a + b1], sc, sh) : 0.f;
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
                    Zs[k1 * ZP + b1 * R + r] = v2f{acc.x * wr - acc.y * wi, acc.x * wi + acc.y * wr};
                }
            }
        }
        __syncthreads();

        // ---- step 2: thread (k1, r): N2-point DFT over b of Z[b][k1]; k1 > N1/2 from the Hermitian symmetry of Y:
        //      Z[b][k1] = conj(Z[b][N1 - k1]) w_N2^b
        {
            const int r = tid % R, k1 = tid / R;
            if (k1 < N1) {
                constexpr RootTab<N2> T2{};
                const bool cj = k1 > N1 / 2;
                const int k1p = cj ? N1 - k1 : k1;
                v2f z[N2];
#pragma unroll
                for (int bb = 0; bb < N2; ++bb) {
                    const v2f t = Zs[k1p * ZP + bb * R + r];
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
        if (more) {
            __syncthreads();   // Z is dead
            stage_rows();
            __syncthreads();
        }
    };
    if constexpr (NLAT == 1) {
        one_latitude(0);
    } else {
#pragma unroll 1
        for (int it = 0; it < NLAT && klat0 + it < p.H; ++it) one_latitude(it);
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

template <int N1, int N2, int R = FFT_ROWS, int NLAT = 1, bool XT = false>
hipError_t launch_fwd(const DftArgs& a, hipStream_t s) {
    constexpr int NT = R * (N1 > N2 ? N1 : N2);
    dim3 grid((unsigned)((a.C + R - 1) / R), (unsigned)((a.H + NLAT - 1) / NLAT), (unsigned)a.Bt), block(NT);
    hipLaunchKernelGGL((dft_forward_fft_kernel<N1, N2, R, NLAT, XT>), grid, block, 0, s, a);
    return hipGetLastError();
}


// ---- inverse ----------------------------------------------------------------------------------------------------------
// y[n] = sum_k F[k] e^(+2 pi i k n / W) with F the Hermitian extension of the stored half spectrum S[m], m < Mm
// (F[0] = Re S[0], F[W/2] = Re S[W/2]: irfft ignores those imaginary parts, fft.py:78-96; zero beyond Mm).  With
// k = k1 + N1 k2, n = N2 a + b:
//     T[k1][b] = sum_k2 F[k1 + N1 k2] w_N2^(-k2 b)            N2-point inverse DFTs (radix 2 x N2/2), k1 <= N1/2 only:
//     U[k1][b] = w_W^(-k1 b) T[k1][b]                         U[N1 - k1][b] = conj U[k1][b]
//     y[N2 a + b] = Re U[0] + (-1)^a Re U[N1/2] + 2 sum_{0<k1<N1/2} Re(w_N1^(-k1 a) U[k1][b])      (pairs a, N1 - a share sums)
// grid = (ceil(C / 16), H, Bt); block = 16 * max(N2, N1/2 + 1).  The spectral-filter bias is added on the way out.
template <int W>
__device__ constexpr RootTab<W> kRoots{1.0};

template <int N1, int N2, int R>
__global__ __launch_bounds__(R * (N2 > N1 / 2 + 1 ? N2 : N1 / 2 + 1)) void dft_inverse_fft_kernel(DftArgs p) {
    constexpr int W = N1 * N2, H1 = N1 / 2 + 1, M2 = N2 / 2, NT = R * (N2 > H1 ? N2 : H1);
    constexpr int PITCH = W + 1;
    static_assert(N1 % 2 == 0 && N2 % 2 == 0 && N1 >= 4, "even factors");
    constexpr int ZP = ZPitch<N2, R>::value;   // U[k1][b][r], as Z of the forward kernel
    constexpr int YS = R * PITCH, US = 2 * H1 * ZP;
    __shared__ __attribute__((aligned(16))) float smem[YS > US ? YS : US];
    float* ys = smem;
    v2f* Us = reinterpret_cast<v2f*>(smem);

    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * R, k = blockIdx.y, b = blockIdx.z;
    const int kb = k * p.Bt + b;
    const long HW = (long)p.H * W;
    const long N2c = (long)p.Bt * 2 * p.C;

    // ---- step A: thread (k1, r): the N2 inputs F[k1 + N1 k2] straight from memory (64-byte runs over r), N2-point inverse DFT
    {
        const int r = tid % R, k1 = tid / R;
        if (k1 < H1) {
            int c = c0 + r;
            c = c < p.C ? c : p.C - 1;
            const float* sb = p.spec + (long)kb * 2 * p.C + c;
            v2f f[N2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                const int kk = k1 + N1 * k2;
                const bool mir = 2 * kk > W;
                const int m = mir ? W - kk : kk;
                float re = 0.f, im = 0.f;
                if (m < p.Mm) {
                    const float* s = sb + (long)m * p.H * N2c;
                    re = s[0];
                    im = s[p.C];
                }
                if (m == 0 || 2 * m == W) im = 0.f;
                f[k2] = v2f{re, mir ? -im : im};
            }
            // radix 2: even / odd k2 -> two (N2/2)-point inverse DFTs, then T[j] = E[j] + w O[j], T[j + N2/2] = E[j] - w O[j]
            constexpr RootTab<M2> TM{};
            constexpr RootTab<N2> T2{};
#pragma unroll
            for (int j = 0; j < M2; ++j) {
                v2f e = {0.f, 0.f}, o = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < M2; ++i) {
                    const int q = (i * j) % M2;
                    const v2f w = {TM.re[q], -TM.im[q]};     // w_M2^(-i j)
                    const v2f wp = {TM.im[q], TM.re[q]};     // i * w
                    e += v2f{f[2 * i].x, f[2 * i].x} * w;
                    e += v2f{f[2 * i].y, f[2 * i].y} * wp;
                    o += v2f{f[2 * i + 1].x, f[2 * i + 1].x} * w;
                    o += v2f{f[2 * i + 1].y, f[2 * i + 1].y} * wp;
                }
                const float wr = T2.re[j], wi = -T2.im[j];   // w_N2^(-j)
                const v2f wo = {o.x * wr - o.y * wi, o.x * wi + o.y * wr};
                const v2f t0 = e + wo, t1 = e - wo;
                // twiddle w_W^(-k1 b): runtime index (k1 varies over the workgroup)
                const int j0 = k1 * j, j1 = k1 * (j + M2);
                const float a0 = kRoots<W>.re[j0], b0 = -kRoots<W>.im[j0];
                const float a1 = kRoots<W>.re[j1], b1 = -kRoots<W>.im[j1];
                Us[k1 * ZP + j * R + r] = v2f{t0.x * a0 - t0.y * b0, t0.x * b0 + t0.y * a0};
                Us[k1 * ZP + (j + M2) * R + r] = v2f{t1.x * a1 - t1.y * b1, t1.x * b1 + t1.y * a1};
            }
        }
    }
    __syncthreads();

    // ---- step B: thread (b1, r): N1 real outputs y[N2 a + b1] from U[0 .. N1/2][b1]
    {
        const int r = tid % R, b1 = tid / R;
        const bool act = b1 < N2;
        v2f u[H1];
#pragma unroll
        for (int k1 = 0; k1 < H1; ++k1) u[k1] = act ? Us[k1 * ZP + b1 * R + r] : v2f{0.f, 0.f};
        __syncthreads();   // the output rows alias U
        if (act) {
            constexpr RootTab<N1> T1{};
            const int cr = c0 + r < p.C ? c0 + r : p.C - 1;
            const float bias = p.bias ? p.bias[cr] : 0.f;
            float* yr = ys + r * PITCH + b1;
            float s0 = u[0].x + u[N1 / 2].x + bias, sh = u[0].x + ((N1 / 2) % 2 ? -u[N1 / 2].x : u[N1 / 2].x) + bias;
#pragma unroll
            for (int k1 = 1; k1 < N1 / 2; ++k1) {
                s0 += 2.f * u[k1].x;
                sh += (k1 % 2 ? -2.f : 2.f) * u[k1].x;
            }
            yr[0] = s0;
            yr[N2 * (N1 / 2)] = sh;
#pragma unroll
            for (int a = 1; a < N1 / 2; ++a) {
                // (P, Q) = sum_k1 (Ur, Ui) * (2 cos, -2 sin)(2 pi k1 a / N1);  w_N1^j = (cos, -sin)
                v2f pq = {u[0].x + (a % 2 ? -u[N1 / 2].x : u[N1 / 2].x) + bias, 0.f};
#pragma unroll
                for (int k1 = 1; k1 < N1 / 2; ++k1) {
                    const int j = (k1 * a) % N1;
                    pq += u[k1] * v2f{2.f * T1.re[j], 2.f * T1.im[j]};
                }
                yr[N2 * a] = pq.x + pq.y;
                yr[N2 * (N1 - a)] = pq.x - pq.y;
            }
        }
    }
    __syncthreads();

    // ---- rows -> memory, 16 B per lane
    float vmax = 0.f;
    for (int idx = tid; idx < R * (W / 4); idx += NT) {
        const int r = idx / (W / 4), j = idx % (W / 4);
        const int c = c0 + r;
        if (c < p.C) {
            // (same rotation of the element order on the way out: four conflict-free reads instead of four 8-way ones)
            const float* d = ys + r * PITCH + 4 * j;
            const int rot = (tid >> 3) & 3;
            float t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = d[(q + rot) & 3];
            // t[q] holds element (q + rot) & 3: element e sits in t[(e - rot) & 3]
            const float e0 = rot == 0 ? t[0] : rot == 1 ? t[3] : rot == 2 ? t[2] : t[1];
            const float e1 = rot == 0 ? t[1] : rot == 1 ? t[0] : rot == 2 ? t[3] : t[2];
            const float e2 = rot == 0 ? t[2] : rot == 1 ? t[1] : rot == 2 ? t[0] : t[3];
            const float e3 = rot == 0 ? t[3] : rot == 1 ? t[2] : rot == 2 ? t[1] : t[0];
            const float4 v = make_float4(e0, e1, e2, e3);
            *reinterpret_cast<float4*>(p.y + ((long)b * p.C + c) * HW + (long)k * W + 4 * j) = v;
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
    if (p.omax) {   // one atomic per workgroup
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

template <int N1, int N2, int R = FFT_ROWS>
hipError_t launch_inv(const DftArgs& a, hipStream_t s) {
    constexpr int H1 = N1 / 2 + 1, NT = R * (N2 > H1 ? N2 : H1);
    dim3 grid((unsigned)((a.C + R - 1) / R), (unsigned)a.H, (unsigned)a.Bt), block(NT);
    hipLaunchKernelGGL((dft_inverse_fft_kernel<N1, N2, R>), grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace

// true when the FFT form handled the launch (sizes with an instantiated factorisation, 16-byte aligned rows)
bool launch_dft_forward_fft(const DftArgs& a, hipStream_t s, hipError_t* err) {
    if (a.no_fft || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0 || a.Mm > a.W / 2 + 1 || a.H > 65535 || a.Bt > 65535)
        return false;
    switch (a.W) {
        case 360: *err = launch_fwd<20, 18, FFT_ROWS, ACE_FFT_NLAT, ACE_FFT_XT != 0>(a, s); return true;
        case 1440: *err = launch_fwd<40, 36, 8>(a, s); return true;    // 0.25-degree grid: 8 channel rows per workgroup (46 KiB)
        case 720: *err = launch_fwd<30, 24, 8>(a, s); return true;
        case 48: *err = launch_fwd<8, 6>(a, s); return true;
        case 24: *err = launch_fwd<6, 4>(a, s); return true;
        case 16: *err = launch_fwd<4, 4>(a, s); return true;
        default: return false;
    }
}

bool launch_dft_inverse_fft(const DftArgs& a, hipStream_t s, hipError_t* err) {
    if (a.no_fft || (reinterpret_cast<uintptr_t>(a.y) & 15) != 0 || a.Mm > a.W / 2 + 1 || a.H > 65535 || a.Bt > 65535)
        return false;
    switch (a.W) {
        case 360: *err = launch_inv<20, 18, 32>(a, s); return true;   // 32 channel rows per workgroup: 128-byte runs on the spectral side (r02: 81.7 -> 75.6 us; the forward kernel is faster with 16)
        case 1440: *err = launch_inv<40, 36, 8>(a, s); return true;
        case 720: *err = launch_inv<30, 24, 8>(a, s); return true;
        case 48: *err = launch_inv<8, 6>(a, s); return true;
        case 24: *err = launch_inv<6, 4>(a, s); return true;
        case 16: *err = launch_inv<4, 4>(a, s); return true;
        default: return false;
    }
}

}  // namespace ace
