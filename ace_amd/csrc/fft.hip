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
#include <map>
#include <mutex>

#include "kernels.h"
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
// The base is wave-uniform by construction (unit coordinates, kernel arguments), but inside the persistent kernels' unit loop the
// compiler evaluates part of the 64-bit address arithmetic on the vector ALU and would then wrap every access in a waterfall loop
// over "different" descriptors: hand both halves back as scalars.
FDEV auto wide_rsrc(const void* base) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
}
FDEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

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

#ifdef ACE_FFT_TRACE   // measurement builds only (tools/trace_fft.py): s_memtime stamps of wave 0 of every workgroup, 8 per workgroup
__device__ unsigned long long fft_trace[8192 * 8];
#define FT(unit, ev) do { if (threadIdx.x == 0) { const unsigned wg_ = (unsigned)(unit); if (wg_ < 8192) fft_trace[wg_ * 8 + (ev)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define FT_XCC(unit) do { if (threadIdx.x == 0) { const unsigned wg_ = (unsigned)(unit); unsigned x_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_)); if (wg_ < 8192) fft_trace[wg_ * 8 + 7] = ((unsigned long long)(x_ & 0xf) << 32) | blockIdx.x; } } while (0)
#else
#define FT(unit, ev) do { } while (0)
#define FT_XCC(unit) do { } while (0)
#endif
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
// Round 4: PERSISTENT workgroups.  The in-kernel timelines (profiles/r04_s3_fft_inkernel_timeline.txt) showed a workgroup of the
// inverse kernel waiting for its spectral entries for half of its life (14 k of 29 k cycles) and one of the forward kernel for its
// rows for 37 % - with every phase of a unit strictly after its loads and only 3 - 5 workgroups per CU to overlap.  Now a
// workgroup walks several units and issues the loads of the NEXT unit (into registers) as soon as the current unit's values have
// left them, so they are in flight under the current unit's arithmetic, LDS exchanges and stores.  grid = G workgroups, G chosen by
// the launcher so that all of them are resident at once and every one gets the same number of units (+- 1).
// unit u = (channel block, latitude, sample); workgroup w's j-th unit, XCD form: XCD (w % 8) owns the contiguous range
// [x T / 8, (x + 1) T / 8) and walks it with stride G / 8 - the channel blocks of one latitude share an L2 (see above).
template <bool XCD>
FDEV bool fft_unit(int j, int total, int gx, int gxy, int& cblk, int& lat, int& b) {
    const int w = (int)blockIdx.x, G = (int)gridDim.x;
    int u;
    if (XCD && (total & 7) == 0 && (G & 7) == 0) {
        const int idx = (w >> 3) + j * (G >> 3);
        if (idx >= (total >> 3)) return false;
        u = (w & 7) * (total >> 3) + idx;
    } else {
        u = w + j * G;
        if (u >= total) return false;
    }
    // (the integer divisions run on the vector ALU: hand the results back as scalars, or every buffer descriptor built from them
    // counts as divergent and each load / store becomes a waterfall loop)
    b = __builtin_amdgcn_readfirstlane(u / gxy);
    const int rem = u - b * gxy;
    lat = __builtin_amdgcn_readfirstlane(rem / gx);
    cblk = rem - lat * gx;
    return true;
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
// PERSIST: several units per workgroup with the next unit's loads in flight (the 1-degree width, where it was measured); otherwise one
// unit per workgroup and grid = units, as in round 3 (the prefetch registers cost the 0.25-degree kernels two thirds of their occupancy)
// MEASURED AND SWITCHED OFF (round 4, same box, profiles/r04_s4_*, r04_s5_*): at W = 360 the persistent forms ran 64 / 80 us
// (forward / inverse) against 46 / 49 us for one unit per workgroup.  The per-unit timelines show why: with the next unit's
// loads issued the wave sits in the ISSUE of those loads (the memory pipeline accepts them at the rate it drains - these kernels
// move 200 MB in 46 us, i.e. the "load latency" of the one-unit form is bandwidth queueing, not exposed latency), and the
// prefetch registers cut the resident workgroups per CU from 5 to 3, which is what overlapped the phases before.  The code stays
// for the record and for other widths' experiments; every width runs one unit per workgroup.
template <int N1, int N2>
constexpr bool fft_persistent() { return false; }

template <int N1, int N2, int R, bool PLN, bool FULLM>
__global__ __launch_bounds__(R * N2, N1 * N2 == 360 ? (PLN ? 7 : ACE_FFT_FWD_WAVES) : (N1 * N2 == 1440 && R == 8 ? 4 : 0)) void dft_forward_fft_kernel(DftArgs p, int total, int gx, int gxy) {
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
    // The level-to-level twiddles sit in LDS for the life of the workgroup: inside the unit loop the kernel issues NO global load
    // that is younger than the prefetch of the next unit's rows and needed before it (loads return in order - the first persistent
    // form read its twiddles from memory after issuing the prefetch and waited for the whole of it in level 1: 14.5 k cycles
    // instead of 3 k, slower than the one-unit-per-workgroup kernel; profiles/r04_s4_fft_inkernel_timeline.txt).
    using Tw = TwTab<N1, N2, false>;
    constexpr bool PERSIST = fft_persistent<N1, N2>();
    constexpr int TWN = Tw::ROWS * Tw::CP / 2;    // float4 entries
    __shared__ float4 twl[PERSIST ? TWN : 1];     // (one unit per workgroup: straight from memory, as in round 3 - no LDS spent on them)
    if constexpr (PERSIST)
        for (int t = threadIdx.x; t < TWN; t += NT) twl[t] = reinterpret_cast<const float4*>(kTw<N1, N2, false>.v)[t];
    const float4* twsrc = PERSIST ? twl : reinterpret_cast<const float4*>(kTw<N1, N2, false>.v);

    // Every per-lane index below is derived from a LAUNDERED copy of the thread id taken inside the unit loop: as loop invariants the
    // compiler hoists all of them (LDS addresses of both levels, load / store offsets, predicates) and keeps them alive for the whole
    // kernel - 127 instead of 36 registers, spills under the occupancy cap.  Recomputing them per unit is a few dozen integer ops.
    auto fresh_tid = [] { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; };
    const long HW = (long)p.H * W;
    constexpr int NPF = (R * (W / 4) + NT - 1) / NT;            // fp32 rows: 16-byte row pieces per thread
    constexpr int NE = (R / 8) * W, NPE = (NE + NT - 1) / NT;   // planes: 16-byte entries (8 channels of a pixel) per thread and plane

    // ---- the loads of one unit: issued into registers, staged into LDS by the SAME thread one unit later
    float4 pf[PLN ? 1 : NPF];
    u32x4 eh[PLN ? NPE : 1], el[PLN ? NPE : 1];
    float scale_p = 1.f;
    if constexpr (PLN) {
        // the producer's power-of-two scale comes off in the level-1 affine (an exact scaling: same values as dividing at the load)
        static_assert(R % 8 == 0, "whole k-groups");
        scale_p = ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(slot_load(p.xslot + (threadIdx.x & 63)))));
    }
    auto issue_loads = [&](int cblk, int k, int b) {
        const int tid = fresh_tid();
        const int c0 = cblk * R;
        const int rlast = (p.C - c0 < R ? p.C - c0 : R) - 1;   // ragged last channel block: its missing rows repeat the last one
        if constexpr (PLN) {
            // the field arrives as P-format planes [C/8][H W][8] (hi | lo): an entry = 8 channels of one pixel, 16 bytes per plane;
            // this workgroup's R rows are R / 8 k-groups.  (hi + lo) / scale is the producer's 22-bit value, exactly.
            const int kgmax = p.C / 8 - 1;
            const auto rsh = wide_rsrc(p.xhi + (long)b * p.sxp + (long)k * W * 8);
            const auto rsl = wide_rsrc(p.xlo + (long)b * p.sxp + (long)k * W * 8);
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
        } else {
            // rows -> registers (all pieces of the thread in flight at once), 16 bytes per lane.  Every global access of the kernel
            // is (uniform 64-bit base) + (32-bit lane offset): no per-access 64-bit vector arithmetic.
            const auto rsx = wide_rsrc(p.x + ((long)b * p.C + c0) * HW + (long)k * W);
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
        }
    };
    auto stage_rows = [&]() {   // registers -> LDS rows
        const int tid = fresh_tid();
        if constexpr (PLN) {
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
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int idx = tid + q * NT;
                if (idx < R * (W / 4)) {
                    const int r = idx / (W / 4), j = idx % (W / 4);
                    *reinterpret_cast<float4*>(xs + r * PITCH + 4 * j) = pf[q];
                }
            }
        }
    };

    int cblk, k, b;
    if (!fft_unit<ACE_FFT_XCD != 0>(0, total, gx, gxy, cblk, k, b)) return;
    issue_loads(cblk, k, b);
    float vmax = 0.f;
    for (int j = 0;; ++j) {
        FT(cblk + gx * (k + p.H * b), 0); FT_XCC(cblk + gx * (k + p.H * b));
        const int tid = fresh_tid();
        const int r1 = tid % R, b1 = tid / R;                       // level-1 role (b1, r1); level-2 role (k1 = b1, r = r1)
        const int c0 = cblk * R;
        const int rlast = (p.C - c0 < R ? p.C - c0 : R) - 1;
        // fused instance-norm affine of this thread's row in level 1 (one load pair per thread, applied in registers)
        const int cr = c0 + (r1 < rlast ? r1 : rlast);
        const float sc = (p.sc ? p.sc[(long)b * p.C + cr] : 1.f) * scale_p;
        const float sh = p.sc ? p.sh[(long)b * p.C + cr] : 0.f;
        // (the affine above is this unit's only other global load: issued before the next unit's rows are requested)
        stage_rows();
        __syncthreads();
        FT(cblk + gx * (k + p.H * b), 1);   // rows landed and staged
        // the next unit's rows: in flight under both levels of this one
        int cblk2 = 0, k2 = 0, b2 = 0;
        const bool more = PERSIST && fft_unit<ACE_FFT_XCD != 0>(j + 1, total, gx, gxy, cblk2, k2, b2);
        if (more) issue_loads(cblk2, k2, b2);

        const int kb = k * p.Bt + b;
        // ---- level 1: thread (b1, r1): outputs k1 = 0 .. N1/2, times w_W^(b1 k1) (2 pi / W folded in)
        {
            float xv[N1];
#pragma unroll
            for (int a = 0; a < N1; ++a) xv[a] = fmaf(xs[r1 * PITCH + N2 * a + b1], sc, sh);
            float4 tw[Tw::CP / 2];
#pragma unroll
            for (int i = 0; i < Tw::CP / 2; ++i) tw[i] = twsrc[b1 * (Tw::CP / 2) + i];
            __syncthreads();   // Z aliases the rows: every row value is in registers before any Z is written
            sfft::RFft<N1, v2f>::run([&](int a) { return xv[a]; }, [&](int k1, v2f y) {
                const v2f w = k1 % 2 ? v2f{tw[k1 / 2].z, tw[k1 / 2].w} : v2f{tw[k1 / 2].x, tw[k1 / 2].y};
                Zs[k1 * ZP + b1 * R + r1] = cmul(y, w);
            });
        }
        __syncthreads();
        FT(cblk + gx * (k + p.H * b), 2);   // level 1 done

        // ---- level 2: thread (k1 <= N1/2, r)
        {
            const int r = r1, k1 = b1;
            if (k1 < H1 && r <= rlast) {
                // spec_out[m][lat][b][re | im][c]: plane stride ms floats.  Store address = (uniform base of wavenumber N1 q) + (lane
                // offset of column k1 or N1 - k1, real or imaginary row)
                const int ms = p.H * p.Bt * 2 * p.C;      // floats per wavenumber plane (lane_offsets_fit: 42 planes of bytes < 2 GiB)
                char* obase = reinterpret_cast<char*>(p.spec_out + (long)kb * 2 * p.C + c0);
                // Entries beyond Mm (and the mirror of the self-conjugate columns 0 and N1/2) are dropped by the descriptor's range
                // check: their lane offset is replaced by one past num_records - no branch per store.
                constexpr unsigned kDrop = 0xF0000000u;
                const unsigned od = (unsigned)(k1 * ms + r) * 4u, om = (unsigned)((N1 - k1) * ms + r) * 4u;
                const int imoff = p.C * 4;
                const bool has_mirror = k1 > 0 && k1 < N1 / 2;
                const v2f* zcol = Zs + k1 * ZP + r;
                // ONE descriptor per unit; the wavenumber block N1 q goes into the instruction's SCALAR offset (one SGPR each instead of
                // a descriptor each: inside the unit loop the per-block descriptors of this and of the prefetched unit spilled SGPRs)
                const auto rs = wide_rsrc(obase);
                int blockb = N1 * ms * 4;                 // bytes between wavenumber blocks (lane_offsets_fit: 42 of them < 2 GiB)
                asm volatile("" : "+s"(blockb));          // (its multiples are not to be hoisted out of the unit loop into 19 SGPRs)
                sfft::CFft<N2, false, v2f>::run([&](int bb) { return zcol[bb * R]; }, [&](int q, v2f out) {
                    const float mag = FULLM ? 0.f : fmaxf(fabsf(out.x), fabsf(out.y));
                    if (FULLM) vmax = fmaxf(vmax, fmaxf(fabsf(out.x), fabsf(out.y)));   // v_max3_f32
                    if (q < K2N) {
                        const bool ok = k1 + N1 * q < p.Mm && (ACE_FFT_ABL != 1 || out.x == 1.2345e-30f);
                        const int off = (int)(ok ? od : kDrop);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out.x), rs, off, uni(q * blockb), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out.y), rs, off, uni(q * blockb + imoff), 0);
                        if (!FULLM) vmax = fmaxf(vmax, ok ? mag : 0.f);
                    }
                    if (q >= N2 / 2) {   // k2 = N2 - 1 - q <= N2 / 2 - 1 (k2 = N2 / 2 would be beyond W / 2 for every mirrored column)
                        const bool ok = has_mirror && (N1 - k1) + N1 * (N2 - 1 - q) < p.Mm && (ACE_FFT_ABL != 1 || out.x == 1.2345e-30f);
                        const int off = (int)(ok ? om : kDrop);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out.x), rs, off, uni((N2 - 1 - q) * blockb), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-out.y), rs, off, uni((N2 - 1 - q) * blockb + imoff), 0);
                        if (!FULLM) vmax = fmaxf(vmax, ok ? mag : 0.f);
                    }
                });
            }
        }
        FT(cblk + gx * (k + p.H * b), 3);   // level 2 done, stores issued
        if (!more) break;
        __syncthreads();    // every Z value has been read: the next unit's rows may overwrite them
        cblk = cblk2; k = k2; b = b2;
    }
    if (p.omax) {   // one atomic per workgroup (Z is dead: reduce the wave maxima through LDS)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads();
        const int tid = threadIdx.x;
        if ((tid & 63) == 0) smem[tid >> 6] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < (NT + 63) / 64; ++w) m = fmaxf(m, smem[w]);
            atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(m));
        }
    }
}

// grid of a persistent launch: every workgroup resident at once (occupancy of THIS kernel x the CUs of the device), the same
// number of units each (+- 1), a multiple of 8 when the unit count is (the XCD-contiguous unit ranges need both)
template <class K>
int fft_grid(K kernel, int block, int total) {
    static int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    static std::mutex mu;
    static std::map<const void*, int> cache;   // workgroups per CU of each instantiation, asked once
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(reinterpret_cast<const void*>(kernel));
        if (it == cache.end()) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu <= 0) per_cu = 1;
            cache.emplace(reinterpret_cast<const void*>(kernel), per_cu);
        } else {
            per_cu = it->second;
        }
    }
    const long slots = (long)ncu * per_cu;
    const long rounds = (total + slots - 1) / slots;
    long G = (total + rounds - 1) / rounds;
    if ((total & 7) == 0) G = (G + 7) & ~7L;
    return (int)(G < total ? G : total);
}

template <int N1, int N2, int R = FFT_ROWS>
hipError_t launch_fwd(const DftArgs& a, hipStream_t s) {
    const int gx = (a.C + R - 1) / R, gxy = gx * a.H, total = gxy * a.Bt;
    dim3 block(R * N2);
    const bool full = a.Mm == a.W / 2 + 1;
    auto go = [&](auto kernel) {
        dim3 grid((unsigned)(fft_persistent<N1, N2>() ? fft_grid(kernel, R * N2, total) : total));
        hipLaunchKernelGGL(kernel, grid, block, 0, s, a, total, gx, gxy);
    };
    if (a.xhi) {
        if (!a.xlo || !a.xslot || a.C % 8 != 0) return hipErrorInvalidValue;
        if constexpr (R % 8 == 0) {
            if (full) go(dft_forward_fft_kernel<N1, N2, R, true, true>);
            else go(dft_forward_fft_kernel<N1, N2, R, true, false>);
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        if (full) go(dft_forward_fft_kernel<N1, N2, R, false, true>);
        else go(dft_forward_fft_kernel<N1, N2, R, false, false>);
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
__global__ __launch_bounds__(R * N2, N1 * N2 == 360 ? ACE_FFT_INV_WAVES : (N1 * N2 == 1440 && R == 8 ? 4 : 0)) void dft_inverse_fft_kernel(DftArgs p, int total, int gx, int gxy) {
    constexpr bool PERSIST = fft_persistent<N1, N2>();
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
    using Tw = TwTab<N1, N2, true>;               // persistent form: twiddles in LDS for the life of the workgroup (see the forward kernel)
    constexpr int TWN = Tw::ROWS * Tw::CP / 2;
    __shared__ float4 twl[PERSIST ? TWN : 1];
    if constexpr (PERSIST) {
        for (int t = threadIdx.x; t < TWN; t += NT) twl[t] = reinterpret_cast<const float4*>(kTw<N1, N2, true>.v)[t];
        __syncthreads();
    }
    const float4* twsrc = PERSIST ? twl : reinterpret_cast<const float4*>(kTw<N1, N2, true>.v);

    auto fresh_tid = [] { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; };   // see the forward kernel
    const long HW = (long)p.H * W;

    // ---- the loads of one unit (step A's inputs): thread (k1, r) takes the N2 entries F[k1 + N1 k2] straight from memory (runs of R
    //      floats over r).  Which stored entry is F[k1 + N1 k2] is known per k2 at compile time: below N2/2 the entry itself (column
    //      k1 of wavenumber block k2), above it the conjugate of column N1 - k1 of block N2 - 1 - k2; k2 = N2/2 is the entry
    //      W/2 for k1 = 0 and mirrored otherwise.  Address = (uniform base of the block) + (32-bit lane offset of the column).
    v2f f[N2];
    auto issue_loads = [&](int cblk, int k, int b) {
        const int tid = fresh_tid();
        const int r = tid % R, k1 = tid / R;
        if (k1 >= H1) return;
        const int c0 = cblk * R;
        const int rlast = (p.C - c0 < R ? p.C - c0 : R) - 1;
        const int kb = k * p.Bt + b;
        const int rr = r < rlast ? r : rlast;
        const int ms = p.H * p.Bt * 2 * p.C;
        const char* sbase = reinterpret_cast<const char*>(p.spec + (long)kb * 2 * p.C + c0);
        // entries beyond Mm read as zero through the descriptor's range check (lane offset one past num_records): no branches
        constexpr unsigned kDrop = 0xF0000000u;
        const unsigned od = (unsigned)(k1 * ms + rr) * 4u, om = (unsigned)((N1 - k1) * ms + rr) * 4u;
        const int imoff = p.C * 4;
        // ONE descriptor per unit; the wavenumber block goes into the instruction's SCALAR offset (see the forward kernel's stores)
        const auto rs = wide_rsrc(sbase);
        int blockb = N1 * ms * 4;
        asm volatile("" : "+s"(blockb));
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const bool mir = k2 > N2 / 2 || (k2 == N2 / 2 && k1 > 0);
            const int blk = k2 < N2 / 2 ? k2 : N2 - 1 - k2;      // wavenumber block of the stored entry (k2 = N2/2, k1 = 0: below)
            const int m = (mir ? N1 - k1 : k1) + N1 * blk;
            if (k2 == N2 / 2) {
                // k1 = 0: the entry W/2 itself (block N2/2, real); otherwise mirrored from block N2/2 - 1
                const int off0 = (int)((k1 == 0 && N1 * (N2 / 2) < p.Mm) ? od : kDrop);
                const int off1 = (int)((k1 > 0 && m < p.Mm) ? om : kDrop);
                const float re0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off0, uni((N2 / 2) * blockb), 0));
                const float re1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off1, uni((N2 / 2 - 1) * blockb), 0));
                const float im1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off1, uni((N2 / 2 - 1) * blockb + imoff), 0));
                f[k2] = v2f{re0 + re1, -im1};                     // (one of the two is a dropped load: zero)
            } else {
                const int off = (int)(m < p.Mm ? (mir ? om : od) : kDrop);
                const float re = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, uni(blk * blockb), 0));
                const float im = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, uni(blk * blockb + imoff), 0));
                f[k2] = v2f{re, mir ? -im : ((k2 == 0 && k1 == 0) ? 0.f : im)};   // F[0] is real
            }
        }
    };

    int cblk, k, b;
    if (!fft_unit<XCD>(0, total, gx, gxy, cblk, k, b)) return;
    issue_loads(cblk, k, b);
    float vmax = 0.f;
    for (int j = 0;; ++j) {
        const int unit_id = cblk + gx * (k + p.H * b);
        FT(unit_id, 0); FT_XCC(unit_id);
        const int tid = fresh_tid();
        const int r = tid % R, k1 = tid / R;     // step A role (k1, r); step B role (b1 = k1, r)
        const int c0 = cblk * R;
        const int rlast = (p.C - c0 < R ? p.C - c0 : R) - 1;
        // (loaded before the next unit's entries are requested: loads return in order - see the forward kernel)
        const float bias = p.bias ? p.bias[c0 + (r < rlast ? r : rlast)] : 0.f;
        // ---- step A: N2-point inverse FFT of the loaded entries, twiddle
        if (k1 < H1) {
            const float4* twp = twsrc + k1 * (Tw::CP / 2);
            sfft::CFft<N2, true, v2f>::run([&](int jj) { return f[jj]; }, [&](int jj, v2f t) {
                const float4 w2 = twp[jj / 2];                                    // w_W^(-k1 j), two per 16 bytes
                Us[k1 * ZP + jj * R + r] = cmul(t, jj % 2 ? v2f{w2.z, w2.w} : v2f{w2.x, w2.y});
            });
        }
        __syncthreads();
        FT(unit_id, 1);   // step A done
        // the next unit's entries: in flight under step B and the copy-out of this one
        int cblk2 = 0, k2n = 0, b2 = 0;
        const bool more = PERSIST && fft_unit<XCD>(j + 1, total, gx, gxy, cblk2, k2n, b2);
        if (more) issue_loads(cblk2, k2n, b2);

        // ---- step B: thread (b1, r): N1 real outputs y[N2 a + b1] from U[0 .. N1/2][b1]
        {
            const int b1 = k1;
            v2f u[H1];
#pragma unroll
            for (int q = 0; q < H1; ++q) u[q] = Us[q * ZP + b1 * R + r];
            __syncthreads();   // the output rows alias U
            constexpr RootTab<N1> T1{};
            float* yr = ys + b1 * RP + r;                 // longitude N2 a + b1 of row r at yr[N2 a RP]
            float s0 = u[0].x + u[N1 / 2].x + bias, sh = u[0].x + ((N1 / 2) % 2 ? -u[N1 / 2].x : u[N1 / 2].x) + bias;
#pragma unroll
            for (int q = 1; q < N1 / 2; ++q) {
                s0 += 2.f * u[q].x;
                sh += (q % 2 ? -2.f : 2.f) * u[q].x;
            }
            yr[0] = s0;
            yr[N2 * (N1 / 2) * RP] = sh;
#pragma unroll
            for (int a = 1; a < N1 / 2; ++a) {
                // (P, Q) = sum_k1 (Ur, Ui) * (2 cos, -2 sin)(2 pi k1 a / N1);  w_N1^j = (cos, -sin)
                v2f pq = {u[0].x + (a % 2 ? -u[N1 / 2].x : u[N1 / 2].x) + bias, 0.f};
#pragma unroll
                for (int q = 1; q < N1 / 2; ++q) {
                    const int jr = (q * a) % N1;
                    pq += u[q] * v2f{2.f * T1.re[jr], 2.f * T1.im[jr]};
                }
                yr[N2 * a * RP] = pq.x + pq.y;
                yr[N2 * (N1 - a) * RP] = pq.x - pq.y;
            }
        }
        __syncthreads();
        FT(unit_id, 2);   // step B done

        // ---- rows -> memory, 16 B per lane
        const auto rsy = wide_rsrc(p.y + ((long)b * p.C + c0) * HW + (long)k * W);
        constexpr int JG = (W / 4 + 7) / 8;            // groups of eight 16-byte pieces per row
        for (int idx = tid; idx < JG * 8 * R; idx += NT) {
            const int jq = (idx / (8 * R)) * 8 + (idx & 7), rw = (idx >> 3) % R;
            if (rw <= rlast && jq < W / 4) {
                const float* src = ys + (4 * jq) * RP + rw;
                const float4 v = make_float4(src[0], src[RP], src[2 * RP], src[3 * RP]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, (int)(((unsigned)rw * (unsigned)HW + 4u * jq) * 4u), 0, 0);
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
        }
        FT(unit_id, 3);   // copy-out issued
        if (!more) break;
        __syncthreads();    // the rows have been read: the next unit's U may overwrite them
        cblk = cblk2; k = k2n; b = b2;
    }
    if (p.omax) {   // one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads();
        const int tid = threadIdx.x;
        if ((tid & 63) == 0) smem[tid >> 6] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < (NT + 63) / 64; ++w) m = fmaxf(m, smem[w]);
            atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(m));
        }
    }
}

template <int N1, int N2, int R = FFT_ROWS>
hipError_t launch_inv(const DftArgs& a, hipStream_t s) {
    const int gx = (a.C + R - 1) / R, gxy = gx * a.H, total = gxy * a.Bt;
    auto kernel = dft_inverse_fft_kernel<N1, N2, R, ACE_FFT_XCD_INV != 0>;
    dim3 grid((unsigned)(fft_persistent<N1, N2>() ? fft_grid(kernel, R * N2, total) : total)), block(R * N2);
    hipLaunchKernelGGL(kernel, grid, block, 0, s, a, total, gx, gxy);
    return hipGetLastError();
}

// the kernels address with 32-bit lane offsets: at most (N1 + 1) spectral planes, and 32 grid rows, from a workgroup's base
bool lane_offsets_fit(const DftArgs& a) {
    const double plane = (double)a.H * a.Bt * 2.0 * a.C * 4.0, rows = 33.0 * (double)a.H * a.W * 4.0;
    return 42.0 * plane < 2147483647.0 && rows < 2147483647.0;
}

}  // namespace

#ifdef ACE_FFT_TRACE
extern "C" int ace_debug_fft_trace(void* dst, int clear) {
    if (clear) {
        void* sym = nullptr;
        hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(fft_trace));
        if (e != hipSuccess) return (int)e;
        return (int)hipMemset(sym, 0, sizeof(fft_trace));
    }
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fft_trace), sizeof(fft_trace));
}
#endif

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
