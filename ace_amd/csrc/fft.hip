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
#include "pack_frag.h"
#include "small_fft.h"
#include "strip_common.h"

namespace ace {
namespace {

#define FDEV __device__ __forceinline__

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Every global access of these kernels is (uniform 64-bit base, in a buffer descriptor) + (32-bit lane offset): left to the
// compiler as pointer arithmetic, each of the ~40 scattered 4-byte spectral accesses per thread cost two 64-bit vector
// instructions of address arithmetic (r03: the kernels are bound by vector-ALU issue, not by bytes).
// y * w, as two packed operations (the swizzle and the sign of {-w.y, w.x} fold into the instruction's op_sel / neg modifiers)
FDEV v2f cmul(const v2f y, const v2f w) {
    const v2f t = v2f{y.x, y.x} * w;
    return v2f{y.y, y.y} * v2f{-w.y, w.x} + t;
}
FDEV auto wide_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000); }

using sfft::cdbl;
using sfft::kPi;
using sfft::unit_root;

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

// Level-to-level twiddles, laid out for the thread that applies them: forward [b][k1] = (2 pi / W) w_W^(b k1) (thread (b, r) walks
// k1), inverse [k1][b] = w_W^(-k1 b) (thread (k1, r) walks b); rows padded to an even count so that two entries load as 16 bytes.
template <int N1, int N2, bool INV>
struct alignas(16) TwTab {
    static constexpr int ROWS = INV ? N1 / 2 + 1 : N2, COLS = INV ? N2 : N1 / 2 + 1, CP = (COLS + 1) & ~1;
    float v[ROWS * CP * 2];
    constexpr TwTab() : v() {
        constexpr long W = (long)N1 * N2;
        for (int r = 0; r < ROWS; ++r)
            for (int c = 0; c < COLS; ++c) {
                const cdbl w = unit_root((long)r * c, W);
                const double scale = INV ? 1.0 : 2.0 * kPi / (double)W;
                v[2 * (r * CP + c)] = (float)(scale * w.re);
                v[2 * (r * CP + c) + 1] = (float)(scale * (INV ? -w.im : w.im));
            }
    }
};
template <int N1, int N2, bool INV>
__device__ constexpr TwTab<N1, N2, INV> kTw{};

#ifndef ACE_FFT_ABL
#define ACE_FFT_ABL 0   // measurement only (forward kernel): 1 no spectral stores, 3 no grid loads
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


// Workgroup -> (channel block, latitude).  Workgroups go round-robin over the 8 XCDs by linear id, and every XCD has its own L2:
// in launch order the R-row (64- or 128-byte) spectral runs of neighbouring channel blocks would be written from eight different
// L2s as partial lines.  With a unit count divisible by 8, XCD x takes the contiguous unit range [x T/8, (x+1) T/8) instead, so
// the channel blocks of one latitude share an L2 and leave it as whole lines.
// (r03 same-box A/B at 1 degree: forward 71.7 -> 68.0 us; the inverse kernel, whose spectral side is READ in 128-byte runs,
// got slower, 54.6 -> 59.1 us, and keeps launch order.)
#ifndef ACE_FFT_XCD
#define ACE_FFT_XCD 1
#endif
#ifndef ACE_FFT_XCD_INV
#define ACE_FFT_XCD_INV 0
#endif
template <bool XCD>
FDEV void fft_unit(int& cblk, int& lat) {
    const int gx = (int)gridDim.x, id = (int)blockIdx.x + gx * (int)blockIdx.y, total = gx * (int)gridDim.y;
    int u = id;
    if (XCD && (total & 7) == 0) u = (id & 7) * (total >> 3) + (id >> 3);
    cblk = u % gx;
    lat = u / gx;
}
// The forward kernel's form with RIDER workgroups (DftArgs::nride, pack_frag.h): the grid's y extent carries ceil(nride / gx) extra
// rows; the riders are the FIRST units of every XCD's range - dispatched first, their slots go back to the transform's own
// workgroups (at the end they would sit in the launch's tail; all on one XCD they cost its range 2 us).  Returns true for a rider
// (index rid, idle when rid >= nride).
template <bool XCD>
FDEV bool fft_unit_ride(int nride, int& cblk, int& lat, int& rid) {
    const int gx = (int)gridDim.x, id = (int)blockIdx.x + gx * (int)blockIdx.y, total = gx * (int)gridDim.y;
    const int nextra = nride > 0 ? (nride + gx - 1) / gx * gx : 0;
    int u;
    if (XCD && (total & 7) == 0 && (nextra & 7) == 0) {
        const int xcd = id & 7, pos = id >> 3, nr8 = nextra >> 3;
        if (pos < nr8) { rid = pos * 8 + xcd; return true; }
        u = xcd * ((total >> 3) - nr8) + (pos - nr8);
    } else {
        if (id < nextra) { rid = id; return true; }
        u = id - nextra;
    }
    cblk = u % gx;
    lat = u / gx;
    return false;
}

// ---- forward ----------------------------------------------------------------------------------------------------------
// grid = (ceil(C / R), H, Bt); block = R * N2.  LDS: the rows, then - aliased - Z.
// Level 1: thread (b, r) runs the real-input FFT of x[N2 a + b] (small_fft.h: radix 2 down to an odd length), k1 <= N1/2.
// Level 2: thread (k1 <= N1/2, r) runs ONE full N2-point complex FFT of column k1 and stores it twice: outputs q <= N2/2 are
// X[k1 + N1 q]; and because Z[b][N1 - k1] = conj(Z[b][k1]) w_N2^b (Y is Hermitian in k1), the same outputs conjugated are the
// mirrored column: X[(N1 - k1) + N1 k2] = conj(out[N2 - 1 - k2]).  (Round 2 ran both levels as direct sums on all N1 columns:
// 977 vector instructions per wave and 78 % ALU issue; this form needs about a third of the arithmetic.)
#ifndef ACE_FFT_FWD_WAVES
#define ACE_FFT_FWD_WAVES 0
#endif
#ifndef ACE_FFT_FWD_PLN_WAVES
#define ACE_FFT_FWD_PLN_WAVES 7   // planes-input forward kernel at W = 360: five 5-wave workgroups per CU need <= 72 registers
#endif
#ifndef ACE_FFT_QROWS
#define ACE_FFT_QROWS 8    // channel rows per workgroup at W = 1440 (46 KiB of LDS per 8 rows)
#endif
#ifndef ACE_FFT_INV_ROWS
#define ACE_FFT_INV_ROWS 32
#endif
#ifndef ACE_FFT_INV_WAVES
#define ACE_FFT_INV_WAVES 7   // three 9-wave workgroups per CU (LDS allows three) need <= 72 registers
#endif
// FULLM: Mm == W / 2 + 1 (every wavenumber kept).  Then every output of a column is either stored or has the magnitude of a
// stored entry (its Hermitian mirror), so the range maximum is the plain maximum over the column - no per-store selects.
template <int N1, int N2, int R, bool PLN, bool FULLM>
__global__ __launch_bounds__(R * N2, N1 * N2 == 360 ? (PLN ? ACE_FFT_FWD_PLN_WAVES : ACE_FFT_FWD_WAVES) : (N1 * N2 == 1440 && R == 8 ? 4 : 0)) void dft_forward_fft_kernel(DftArgs p) {
    constexpr int W = N1 * N2, H1 = N1 / 2 + 1, NT = R * N2;
    constexpr int PITCH = W + 4;     // 16-byte aligned rows; PITCH = 4 (mod 8): the 16 rows x 4 b of a wave's level-1 read hit 64 banks
    constexpr int K2N = N2 / 2 + 1;  // k = k1 + N1 k2 <= W / 2  =>  k2 <= N2 / 2
    static_assert(N1 % 2 == 0 && N2 % 2 == 0 && N2 >= H1 && W % 4 == 0, "even factors, level 1 at least as wide as level 2");
    // Z[k1][b][r], r fastest: level 1 (thread (b, r), fixed k1) writes and level 2 (thread (k1, r), fixed b) reads whole
    // contiguous runs of R entries.  ZP = entries per k1, padded so that the 32 lanes of one ds_read_b64 group (32 / R
    // consecutive k1 x R rows) fall on 64 distinct banks: ZP = R (mod 32).  (r02: the [r][k1][b] order cost 53 % of the
    // kernel's LDS cycles in bank conflicts.)
    constexpr int ZP = ZPitch<N2, R>::value;
    constexpr int XS = R * PITCH, ZS = 2 * H1 * ZP;
    __shared__ __attribute__((aligned(16))) float smem[XS > ZS ? XS : ZS];
    float* xs = smem;
    v2f* Zs = reinterpret_cast<v2f*>(smem);

    const int tid = threadIdx.x;
    int cblk = 0, k = 0, rid = 0;
    if (fft_unit_ride<ACE_FFT_XCD != 0>(p.nride, cblk, k, rid)) {
        // rider workgroup: one block of a weight-packing job (pack_frag.h), for every sample, from plane z = 0
        if (rid >= p.nride || blockIdx.z != 0) return;
        float* red = smem;
        for (int smp = 0; smp < p.ride.nsamples; ++smp) {
            if (smp) __syncthreads();
            pack_frag_block(p.ride, rid, smp, tid, red);
        }
        return;
    }
    const int c0 = cblk * R, b = blockIdx.z;
    const long HW = (long)p.H * W;
    constexpr int NPF = (R * (W / 4) + NT - 1) / NT;   // 16-byte row pieces per thread
    const int rlast = (p.C - c0 < R ? p.C - c0 : R) - 1;   // ragged last channel block: its missing rows repeat the last one

    // fused instance-norm affine of this thread's row in level 1 (one load pair per thread, applied in registers).  The three small
    // loads (scale, shift, range slot) are REQUESTED behind the rows' own loads: in front of them hipcc waited for the slot's value
    // (a whole memory latency, its wave reduction needs it) before the first row request went out.
    const int r1 = tid % R, b1 = tid / R;
    const int cr = c0 + (r1 < rlast ? r1 : rlast);
    float sc = 1.f, sh = 0.f;
    using Tw = TwTab<N1, N2, false>;
    float4 tw[Tw::CP / 2];                          // level-1 twiddles of this thread's b1: requested with them, used behind two barriers
    constexpr bool TW_EARLY = Tw::CP / 2 <= 6;      // (the 40-point level 1 of W = 1440 has 11 of them: held from the top they spill)
    auto load_tw = [&]() {
        const float4* twp = reinterpret_cast<const float4*>(kTw<N1, N2, false>.v) + b1 * (Tw::CP / 2);
#pragma unroll
        for (int i = 0; i < Tw::CP / 2; ++i) tw[i] = twp[i];
    };
    auto load_affine = [&]() {
        sc = p.sc ? p.sc[(long)b * p.C + cr] : 1.f;
        sh = p.sc ? p.sh[(long)b * p.C + cr] : 0.f;
        if constexpr (TW_EARLY) load_tw();
    };
    if constexpr (PLN) {
        // the field arrives as P-format planes [C/8][H W][8] (hi | lo): an entry = 8 channels of one pixel, 16 bytes per plane;
        // this workgroup's R rows are R / 8 k-groups.  (hi + lo) / scale is the producer's 22-bit value, exactly.
        static_assert(R % 8 == 0, "whole k-groups");
        constexpr int NE = (R / 8) * W, NPE = (NE + NT - 1) / NT;
        const int kgmax = p.C / 8 - 1;
        const auto rsh = wide_rsrc(p.xhi + (long)b * p.sxp + (long)k * W * 8);
        const auto rsl = wide_rsrc(p.xlo + (long)b * p.sxp + (long)k * W * 8);
        u32x4 eh[NPE], el[NPE];
#pragma unroll
        for (int q = 0; q < NPE; ++q) {
            const int idx = tid + q * NT;
            if (idx < NE) {
                int kg = c0 / 8 + idx / W;
                kg = kg < kgmax ? kg : kgmax;
                const int off = (int)(((unsigned)kg * (unsigned)HW + (unsigned)(idx % W)) * 16u);
                eh[q] = __builtin_amdgcn_raw_buffer_load_b128(rsh, off, 0, 0);
                el[q] = __builtin_amdgcn_raw_buffer_load_b128(rsl, off, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_affine();
        // the producer's power-of-two scale comes off in the level-1 affine (an exact scaling: same values as dividing here)
        sc *= ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(slot_load(p.xslot + (tid & 63)))));
#pragma unroll
        for (int q = 0; q < NPE; ++q) {
            const int idx = tid + q * NT;
            if (idx < NE) {
                const half8 h8 = __builtin_bit_cast(half8, eh[q]), l8 = __builtin_bit_cast(half8, el[q]);
                float* d = xs + (8 * (idx / W)) * PITCH + idx % W;
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e * PITCH] = (float)h8[e] + (float)l8[e];
            }
        }
    } else {
    // rows -> registers (all pieces of the thread in flight at once) -> LDS, 16 bytes per lane each way.  Every global access
    // of the kernel is (uniform 64-bit base) + (32-bit lane offset): no per-access 64-bit vector arithmetic.
    const auto rsx = wide_rsrc(p.x + ((long)b * p.C + c0) * HW + (long)k * W);
    float4 pf[NPF];
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int idx = tid + q * NT;
        if (idx < R * (W / 4)) {
            const int r = idx / (W / 4), j = idx % (W / 4);
            const unsigned off = ((unsigned)(r < rlast ? r : rlast) * (unsigned)HW + 4u * j) * 4u;
#if ACE_FFT_ABL == 3
            pf[q] = make_float4(1.f + off, 2.f, 3.f, 4.f);
#else
            pf[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0));
#endif
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    load_affine();
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int idx = tid + q * NT;
        if (idx < R * (W / 4)) {
            const int r = idx / (W / 4), j = idx % (W / 4);
            *reinterpret_cast<float4*>(xs + r * PITCH + 4 * j) = pf[q];
        }
    }
    }
    __syncthreads();

    const int kb = k * p.Bt + b;
    // ---- level 1: thread (b1, r1): outputs k1 = 0 .. N1/2, times w_W^(b1 k1) (2 pi / W folded in)
    {
        float xv[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) xv[a] = fmaf(xs[r1 * PITCH + N2 * a + b1], sc, sh);
        if constexpr (!TW_EARLY) load_tw();
        __syncthreads();   // Z aliases the rows: every row value is in registers before any Z is written
        sfft::RFft<N1, v2f>::run([&](int a) { return xv[a]; }, [&](int k1, v2f y) {
            const v2f w = k1 % 2 ? v2f{tw[k1 / 2].z, tw[k1 / 2].w} : v2f{tw[k1 / 2].x, tw[k1 / 2].y};
            Zs[k1 * ZP + b1 * R + r1] = cmul(y, w);
        });
    }
    __syncthreads();

    // ---- level 2: thread (k1 <= N1/2, r)
    float vmax = 0.f;
    {
        const int r = tid % R, k1 = tid / R;
        if (k1 < H1 && r <= rlast) {
            // spec_out[m][lat][b][re | im][c]: plane stride ms floats.  Store address = (uniform base of wavenumber N1 q) + (lane
            // offset of column k1 or N1 - k1, real or imaginary row)
            const long ms = (long)p.H * p.Bt * 2 * p.C;
            char* obase = reinterpret_cast<char*>(p.spec_out + (long)kb * 2 * p.C + c0);
            // Entries beyond Mm (and the mirror of the self-conjugate columns 0 and N1/2) are dropped by the descriptor's range
            // check: their lane offset is replaced by one past num_records - no branch per store.
            constexpr unsigned kDrop = 0xF0000000u;
            const unsigned od = (unsigned)(k1 * ms + r) * 4u, om = (unsigned)((N1 - k1) * ms + r) * 4u;
            const int imoff = p.C * 4;
            const bool has_mirror = k1 > 0 && k1 < N1 / 2;
            const v2f* zcol = Zs + k1 * ZP + r;
            sfft::CFft<N2, false, v2f>::run([&](int bb) { return zcol[bb * R]; }, [&](int q, v2f out) {
                const float mag = FULLM ? 0.f : fmaxf(fabsf(out.x), fabsf(out.y));
                if (FULLM) vmax = fmaxf(vmax, fmaxf(fabsf(out.x), fabsf(out.y)));   // v_max3_f32
                if (q < K2N) {
                    const bool ok = k1 + N1 * q < p.Mm && (ACE_FFT_ABL != 1 || out.x == 1.2345e-30f);
                    const auto rs = wide_rsrc(obase + (long)q * N1 * ms * 4);
                    const int off = (int)(ok ? od : kDrop);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out.x), rs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out.y), rs, off, imoff, 0);
                    if (!FULLM) vmax = fmaxf(vmax, ok ? mag : 0.f);
                }
                if (q >= N2 / 2) {   // k2 = N2 - 1 - q <= N2 / 2 - 1 (k2 = N2 / 2 would be beyond W / 2 for every mirrored column)
                    const bool ok = has_mirror && (N1 - k1) + N1 * (N2 - 1 - q) < p.Mm && (ACE_FFT_ABL != 1 || out.x == 1.2345e-30f);
                    const auto rs = wide_rsrc(obase + (long)(N2 - 1 - q) * N1 * ms * 4);
                    const int off = (int)(ok ? om : kDrop);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out.x), rs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-out.y), rs, off, imoff, 0);
                    if (!FULLM) vmax = fmaxf(vmax, ok ? mag : 0.f);
                }
            });
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

template <int N1, int N2, int R = FFT_ROWS>
hipError_t launch_fwd(const DftArgs& a0, hipStream_t s) {
    DftArgs a = a0;
    const int gx = (a.C + R - 1) / R;
    // riders need a workgroup of at least four waves and a y extent that stays in range; otherwise the caller launches the job itself
    if (a.nride > 0 && (R * N2 < 256 || !pack_frag_args_ok(a.ride) || a.nride != pack_frag_blocks(a.ride) || a.H + (a.nride + gx - 1) / gx > 65535)) a.nride = 0;
    if (a.rode) *a.rode = a.nride > 0;
    dim3 grid((unsigned)gx, (unsigned)(a.H + (a.nride > 0 ? (a.nride + gx - 1) / gx : 0)), (unsigned)a.Bt), block(R * N2);
    const bool full = a.Mm == a.W / 2 + 1;
    if (a.xhi) {
        if (!a.xlo || !a.xslot || a.C % 8 != 0) return hipErrorInvalidValue;
        if (full) hipLaunchKernelGGL((dft_forward_fft_kernel<N1, N2, R, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((dft_forward_fft_kernel<N1, N2, R, true, false>), grid, block, 0, s, a);
    } else {
        if (full) hipLaunchKernelGGL((dft_forward_fft_kernel<N1, N2, R, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((dft_forward_fft_kernel<N1, N2, R, false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}


// ---- inverse ----------------------------------------------------------------------------------------------------------
// y[n] = sum_k F[k] e^(+2 pi i k n / W) with F the Hermitian extension of the stored half spectrum S[m], m < Mm
// (F[0] = Re S[0], F[W/2] = Re S[W/2]: irfft ignores those imaginary parts, fft.py:78-96; zero beyond Mm).  With
// k = k1 + N1 k2, n = N2 a + b:
//     T[k1][b] = sum_k2 F[k1 + N1 k2] w_N2^(-k2 b)            N2-point inverse FFTs (small_fft.h), k1 <= N1/2 only:
//     U[k1][b] = w_W^(-k1 b) T[k1][b]                         U[N1 - k1][b] = conj U[k1][b]
//     y[N2 a + b] = Re U[0] + (-1)^a Re U[N1/2] + 2 sum_{0<k1<N1/2} Re(w_N1^(-k1 a) U[k1][b])      (pairs a, N1 - a share sums)
// grid = (ceil(C / 16), H, Bt); block = 16 * max(N2, N1/2 + 1).  The spectral-filter bias is added on the way out.
template <int N1, int N2, int R, bool XCD>
__global__ __launch_bounds__(R * N2, N1 * N2 == 360 ? ACE_FFT_INV_WAVES : (N1 * N2 == 1440 && R == 8 ? 4 : 0)) void dft_inverse_fft_kernel(DftArgs p) {
    constexpr int W = N1 * N2, H1 = N1 / 2 + 1, NT = R * N2;
    // The output rows are staged COLUMN-major, element (row r, longitude c) at c * RP + r with RP = R + 1: step B's threads (b1, r)
    // write one longitude of R consecutive rows per instruction - consecutive banks (r03: row-major rows of W + 4 floats put the
    // 32 rows of a write on 8 banks, 41 % of the kernel's LDS cycles were bank conflicts); the copy-out reads four longitudes
    // of a row with four 4-byte reads, lanes = 8 row pieces x 8 rows, RP = 1 (mod 8): 32 distinct banks per half wave.
    constexpr int RP = R + 1;
    static_assert(N1 % 2 == 0 && N2 % 2 == 0 && N1 >= 4 && N2 >= H1 && W % 4 == 0 && R % 8 == 0, "even factors, whole row groups");
    constexpr int ZP = ZPitch<N2, R>::value;   // U[k1][b][r], as Z of the forward kernel
    constexpr int YS = W * RP, US = 2 * H1 * ZP;
    __shared__ __attribute__((aligned(16))) float smem[YS > US ? YS : US];
    float* ys = smem;
    v2f* Us = reinterpret_cast<v2f*>(smem);

    const int tid = threadIdx.x;
    int cblk, k;
    fft_unit<XCD>(cblk, k);
    const int c0 = cblk * R, b = blockIdx.z;
    const int kb = k * p.Bt + b;
    const long HW = (long)p.H * W;
    const int rlast = (p.C - c0 < R ? p.C - c0 : R) - 1;
    // the filter bias of this thread's row (step B's (b1, r) mapping has the same r = tid % R): requested here, used behind two
    // barriers - loaded where it is used, every workgroup waited a memory latency for it in the middle of its life
    const int crb = c0 + ((tid % R) < rlast ? (tid % R) : rlast);
    const float bias = p.bias ? p.bias[crb] : 0.f;

    // ---- step A: thread (k1, r): the N2 inputs F[k1 + N1 k2] straight from memory (runs of R floats over r), N2-point inverse
    //      FFT, twiddle.  Which stored entry is F[k1 + N1 k2] is known per k2 at compile time: below N2/2 the entry itself (column
    //      k1 of wavenumber block k2), above it the conjugate of column N1 - k1 of block N2 - 1 - k2; k2 = N2/2 is the entry
    //      W/2 for k1 = 0 and mirrored otherwise.  Address = (uniform base of the block) + (32-bit lane offset of the column).
    {
        const int r = tid % R, k1 = tid / R;
        if (k1 < H1) {
            const int rr = r < rlast ? r : rlast;
            const long ms = (long)p.H * p.Bt * 2 * p.C;
            const char* sbase = reinterpret_cast<const char*>(p.spec + (long)kb * 2 * p.C + c0);
            // entries beyond Mm read as zero through the descriptor's range check (lane offset one past num_records): no branches
            constexpr unsigned kDrop = 0xF0000000u;
            const unsigned od = (unsigned)(k1 * ms + rr) * 4u, om = (unsigned)((N1 - k1) * ms + rr) * 4u;
            const int imoff = p.C * 4;
            v2f f[N2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                const bool mir = k2 > N2 / 2 || (k2 == N2 / 2 && k1 > 0);
                const int blk = k2 < N2 / 2 ? k2 : N2 - 1 - k2;      // wavenumber block of the stored entry (k2 = N2/2, k1 = 0: below)
                const int m = (mir ? N1 - k1 : k1) + N1 * blk;
                if (k2 == N2 / 2) {
                    // k1 = 0: the entry W/2 itself (block N2/2, real); otherwise mirrored from block N2/2 - 1
                    const auto rs0 = wide_rsrc(sbase + (long)(N2 / 2) * N1 * ms * 4);
                    const auto rs1 = wide_rsrc(sbase + (long)(N2 / 2 - 1) * N1 * ms * 4);
                    const int off0 = (int)((k1 == 0 && N1 * (N2 / 2) < p.Mm) ? od : kDrop);
                    const int off1 = (int)((k1 > 0 && m < p.Mm) ? om : kDrop);
                    const float re0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs0, off0, 0, 0));
                    const float re1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, off1, 0, 0));
                    const float im1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, off1, imoff, 0));
                    f[k2] = v2f{re0 + re1, -im1};                     // (one of the two is a dropped load: zero)
                } else {
                    const auto rs = wide_rsrc(sbase + (long)blk * N1 * ms * 4);
                    const int off = (int)(m < p.Mm ? (mir ? om : od) : kDrop);
                    const float re = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                    const float im = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, imoff, 0));
                    f[k2] = v2f{re, mir ? -im : ((k2 == 0 && k1 == 0) ? 0.f : im)};   // F[0] is real
                }
            }
            using Tw = TwTab<N1, N2, true>;
            const float4* twp = reinterpret_cast<const float4*>(kTw<N1, N2, true>.v) + k1 * (Tw::CP / 2);
            sfft::CFft<N2, true, v2f>::run([&](int j) { return f[j]; }, [&](int j, v2f t) {
                const float4 w2 = twp[j / 2];                                    // w_W^(-k1 j), two per 16 bytes
                Us[k1 * ZP + j * R + r] = cmul(t, j % 2 ? v2f{w2.z, w2.w} : v2f{w2.x, w2.y});
            });
        }
    }
    __syncthreads();

    // ---- step B: thread (b1, r): N1 real outputs y[N2 a + b1] from U[0 .. N1/2][b1]
    {
        const int r = tid % R, b1 = tid / R;
        v2f u[H1];
#pragma unroll
        for (int k1 = 0; k1 < H1; ++k1) u[k1] = Us[k1 * ZP + b1 * R + r];
        __syncthreads();   // the output rows alias U
        constexpr RootTab<N1> T1{};
        float* yr = ys + b1 * RP + r;                 // longitude N2 a + b1 of row r at yr[N2 a RP]
        float s0 = u[0].x + u[N1 / 2].x + bias, sh = u[0].x + ((N1 / 2) % 2 ? -u[N1 / 2].x : u[N1 / 2].x) + bias;
#pragma unroll
        for (int k1 = 1; k1 < N1 / 2; ++k1) {
            s0 += 2.f * u[k1].x;
            sh += (k1 % 2 ? -2.f : 2.f) * u[k1].x;
        }
        yr[0] = s0;
        yr[N2 * (N1 / 2) * RP] = sh;
#pragma unroll
        for (int a = 1; a < N1 / 2; ++a) {
            // (P, Q) = sum_k1 (Ur, Ui) * (2 cos, -2 sin)(2 pi k1 a / N1);  w_N1^j = (cos, -sin)
            v2f pq = {u[0].x + (a % 2 ? -u[N1 / 2].x : u[N1 / 2].x) + bias, 0.f};
#pragma unroll
            for (int k1 = 1; k1 < N1 / 2; ++k1) {
                const int j = (k1 * a) % N1;
                pq += u[k1] * v2f{2.f * T1.re[j], 2.f * T1.im[j]};
            }
            yr[N2 * a * RP] = pq.x + pq.y;
            yr[N2 * (N1 - a) * RP] = pq.x - pq.y;
        }
    }
    __syncthreads();

    // ---- rows -> memory, 16 B per lane both ways
    float vmax = 0.f;
    const auto rsy = wide_rsrc(p.y + ((long)b * p.C + c0) * HW + (long)k * W);
    constexpr int JG = (W / 4 + 7) / 8;            // groups of eight 16-byte pieces per row
    for (int idx = tid; idx < JG * 8 * R; idx += NT) {
        const int j = (idx / (8 * R)) * 8 + (idx & 7), r = (idx >> 3) % R;
        if (r <= rlast && j < W / 4) {
            const float* src = ys + (4 * j) * RP + r;
            const float4 v = make_float4(src[0], src[RP], src[2 * RP], src[3 * RP]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, (int)(((unsigned)r * (unsigned)HW + 4u * j) * 4u), 0, 0);
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
    dim3 grid((unsigned)((a.C + R - 1) / R), (unsigned)a.H, (unsigned)a.Bt), block(R * N2);
    // The spectral side is read in runs of R floats per wavenumber.  With R = 32 a run is a whole 128-byte line and launch order is
    // the faster mapping (r03).  With fewer rows per workgroup (the 0.25-degree grid: R = 8, 32-byte runs) FOUR neighbouring channel
    // blocks share every line: dealt round-robin they sit on four XCDs and each fetches the whole line; the XCD-contiguous unit
    // ranges of the forward kernel put them on one L2.
    constexpr bool XCD = ACE_FFT_XCD_INV != 0 || R * 4 < 128;
    hipLaunchKernelGGL((dft_inverse_fft_kernel<N1, N2, R, XCD>), grid, block, 0, s, a);
    return hipGetLastError();
}

// the kernels address with 32-bit lane offsets: at most (N1 + 1) spectral planes, and 32 grid rows, from a workgroup's base
bool lane_offsets_fit(const DftArgs& a) {
    const double plane = (double)a.H * a.Bt * 2.0 * a.C * 4.0, rows = 33.0 * (double)a.H * a.W * 4.0;
    return 42.0 * plane < 2147483647.0 && rows < 2147483647.0;
}

}  // namespace

// true when the FFT form handled the launch (sizes with an instantiated factorisation, 16-byte aligned rows)
bool launch_dft_forward_fft(const DftArgs& a, hipStream_t s, hipError_t* err) {
    if (a.no_fft || (!a.xhi && (reinterpret_cast<uintptr_t>(a.x) & 15) != 0) || a.Mm > a.W / 2 + 1 || a.H > 65535 || a.Bt > 65535 || !lane_offsets_fit(a))
        return false;
    switch (a.W) {
        case 360: *err = launch_fwd<20, 18>(a, s); return true;
        case 1440: *err = launch_fwd<40, 36, ACE_FFT_QROWS>(a, s); return true;    // 0.25-degree grid: 8 channel rows per workgroup (46 KiB)
        case 720: *err = launch_fwd<30, 24, 8>(a, s); return true;
        case 48: *err = launch_fwd<8, 6>(a, s); return true;
        case 24: *err = launch_fwd<6, 4>(a, s); return true;
        case 16: *err = launch_fwd<4, 4>(a, s); return true;
        default: return false;
    }
}

bool dft_fft_has_width(int W) { return W == 360 || W == 1440 || W == 720 || W == 48 || W == 24 || W == 16; }

bool launch_dft_inverse_fft(const DftArgs& a, hipStream_t s, hipError_t* err) {
    if (a.no_fft || (reinterpret_cast<uintptr_t>(a.y) & 15) != 0 || a.Mm > a.W / 2 + 1 || a.H > 65535 || a.Bt > 65535 || !lane_offsets_fit(a))
        return false;
    switch (a.W) {
        case 360: *err = launch_inv<20, 18, ACE_FFT_INV_ROWS>(a, s); return true;   // 32 channel rows per workgroup: 128-byte runs on the spectral side (r02: 81.7 -> 75.6 us; the forward kernel is faster with 16)
        case 1440: *err = launch_inv<40, 36, ACE_FFT_QROWS>(a, s); return true;
        case 720: *err = launch_inv<30, 24, 8>(a, s); return true;
        case 48: *err = launch_inv<8, 6>(a, s); return true;
        case 24: *err = launch_inv<6, 4>(a, s); return true;
        case 16: *err = launch_inv<4, 4>(a, s); return true;
        default: return false;
    }
}

}  // namespace ace
