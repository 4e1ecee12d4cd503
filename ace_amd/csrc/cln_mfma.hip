// Conditional layer norm of the NoiseConditionedSFNO (fme/core/models/conditional_sfno/layers.py:95-141, 245-318) as ONE pass:
//   y[c][p] = ((x[c][p] - mean_p) * rstd_p * gamma_c + beta_c) * (1 + sum_j Ws[c][j] cond[j][p]) + sum_j Wb[c][j] cond[j][p]
// with per-PIXEL statistics over the channels (biased variance, eps inside the sqrt).
//
// kernels.hip does this in two kernels (cln_stats_kernel + cln_apply_kernel): a statistics pass over x, then an apply pass that
// evaluates the two 1x1 "conditioning" convolutions on the vector ALUs - 2 J C multiply-adds per pixel with their weights read from
// LDS, 0.15 ms per norm at the shipped configuration (C = 512, J = 32; 16 norms per step = 20 % of the step).  Here a workgroup
// owns a pixel tile with ALL channels: its waves hold the tile in registers in the MFMA accumulator layout (32 channels per row
// tile), the per-pixel sums are reduced across the waves through LDS in fp64 (as the two-kernel form accumulates), the two
// conditioning convolutions run on v_mfma_f32_32x32x16_f16 with error-compensated operands (weights pre-packed as A fragments at
// upload; cond split once per workgroup into B fragments in LDS, shared by both convolutions, all waves and row tiles), and the
// epilogue applies everything to the resident x and stores y: x is read once and y written once (266 MB instead of 399 MB at
// C = 512).  Two forms:
//   * C = 256 / 512, H W % 4 == 0: 128-pixel tiles, 16 bytes per lane, C / 32 waves (cln_mfma_wide_kernel) - 86 us per norm at
//     C = 512, 180 x 360; optionally writes y as the P-format planes the packed convolutions read (no pack pass, no fp32 y);
//   * C = 768 / 1024, H W % 32 == 0: 32-pixel tiles, 4 bytes per lane, 8 waves x C / 256 row tiles (cln_mfma_kernel).
// J <= 128.  Other shapes keep the two-kernel form.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.h"

namespace ace {
namespace {

#define CDEV __device__ __forceinline__

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

CDEV unsigned slot_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CDEV int pow2_exponent_for(float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 12 - e; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
CDEV float wave_max_bits(unsigned raw) {
    float mx = __uint_as_float(raw);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
}
// lanes of ONE wave exchanging data through LDS: the hardware runs a wave's LDS instructions in order; this keeps the compiler from
// reordering them around the exchange (wavefront-scope release / acquire, no cross-wave barrier)
CDEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
CDEV int acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

constexpr int CLN_MAX_NK = 8;   // J <= 128

CDEV auto wide_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000); }

// C <= 512: 128 registers per lane (two 8-wave workgroups per CU; C = 768 / 1024 take 256): the tile stays resident (16 RT registers), the conditioning convolutions
// and the apply run one 32-channel row tile at a time (32 accumulator registers, B fragments read back from LDS per k-step).
// All addressing is (uniform base in a buffer descriptor) + 32-bit offset.
template <int RT>
__global__ __launch_bounds__(512, RT <= 2 ? 4 : 2) void cln_mfma_kernel(ClnMfmaArgs p) {
    __shared__ double red[2][8][32];
    __shared__ float stat[2][32];
    __shared__ half8 Bf[CLN_MAX_NK][2][64];                         // cond as B fragments: [k-step][hi | lo][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * 32;
    const unsigned HWb = (unsigned)p.HW * 4u;                       // bytes per channel row (eligibility: C HW 4 < 2 GiB)
    const auto rsx = wide_rsrc(p.x + (long)b * p.sx + p0);
    const auto rsy = wide_rsrc(p.y + (long)b * p.sx + p0);
    const unsigned raw_c = p.cslot ? slot_load(p.cslot + lane) : 0u;

    // ---- the tile: wave's RT row tiles of 32 channels x 32 pixels, accumulator layout (register r of lane (i, g) = row acc_row(r, g))
    float xr[RT][16];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xr[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, (int)((unsigned)((wave * RT + t) * 32 + acc_row(r, g)) * HWb + 4u * i), 0, 0));

    // ---- cond -> B fragments in LDS, once per workgroup (wave ks does k-step ks): lane (i, g) holds j = 16 ks + 8 g .. + 7 of pixel i
    float inv_c = 1.f;
    const int nk = (p.J + 15) / 16;
    if (p.As) {
        const int ec = pow2_exponent_for(wave_max_bits(raw_c));
        const float cscale = ldexpf(1.0f, ec);
        inv_c = ldexpf(1.0f, -ec);
        const auto rsc = wide_rsrc(p.cond + (long)b * p.scond + p0);
        for (int ks = wave; ks < nk; ks += 8) {
            half8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 16 * ks + 8 * g + e;
                // (a conditioning index at or beyond J reads row J - 1 and is zeroed: no out-of-range address)
                const float raw = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, (int)((unsigned)(j < p.J ? j : p.J - 1) * HWb + 4u * i), 0, 0));
                const float v = j < p.J ? raw * cscale : 0.f;
                const _Float16 h = (_Float16)v;
                bh[e] = h;
                bl[e] = (_Float16)(v - (float)h);
            }
            Bf[ks][0][lane] = bh;
            Bf[ks][1][lane] = bl;
        }
    }

    // ---- per-pixel statistics over all channels, fp64 sums in a fixed order (lane, lane ^ 32, waves 0..7)
    {
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s += (double)xr[t][r]; ss += (double)xr[t][r] * (double)xr[t][r]; }
        s += __shfl_xor(s, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) { red[0][wave][i] = s; red[1][wave][i] = ss; }
    }
    __syncthreads();
    if (tid < 32) {
        double a = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += red[0][w][tid]; q += red[1][w][tid]; }
        const double mu = a / p.C;
        double var = q / p.C - mu * mu;
        if (var < 0.0) var = 0.0;
        stat[0][tid] = (float)mu;
        stat[1][tid] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    const float mu = stat[0][i], rstd = stat[1][i];
    const float os = inv_c / p.ascale_s, ob = inv_c / p.ascale_b;
    float vmax = 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        // ---- the two conditioning convolutions of this row tile on MFMA: S = Ws cond, Bv = Wb cond (error-compensated fp16)
        f32x16 accS, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accS[r] = 0.f; accB[r] = 0.f; }
        if (p.As) {
            for (int ks = 0; ks < nk; ++ks) {
                const half8 bh = Bf[ks][0][lane], bl = Bf[ks][1][lane];
                const long blk = ((long)(wave * RT + t) * nk + ks) * 1024 + lane * 8;
                const half8 sh = *reinterpret_cast<const half8*>(p.As + blk), sl = *reinterpret_cast<const half8*>(p.As + blk + 512);
                const half8 wh = *reinterpret_cast<const half8*>(p.Ab + blk), wl = *reinterpret_cast<const half8*>(p.Ab + blk + 512);
                accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl, bh, accS, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, accB, 0, 0, 0);
                accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, bl, accS, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, accB, 0, 0, 0);
                accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, bh, accS, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, accB, 0, 0, 0);
            }
        }
        // ---- apply to the resident tile, store
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (wave * RT + t) * 32 + acc_row(r, g);
            float v = (xr[t][r] - mu) * rstd;
            if (p.gamma) v = v * p.gamma[c] + p.beta[c];
            const float o = v * (1.0f + accS[r] * os) + accB[r] * ob;
            vmax = fmaxf(vmax, fabsf(o));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rsy, (int)((unsigned)c * HWb + 4u * i), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // one row tile at a time: interleaving the next tile's loads with this tile's apply is what spills
    }
    if (p.omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) atomicMax(p.omax + ((blockIdx.x + wave) & 63), __float_as_uint(vmax));
    }
}

// ---- the wide form (C = 256 / 512): a workgroup owns 128 pixels x all channels, one 32-channel row tile per wave (C / 32 waves).
// 128-byte row segments (the 32-pixel tile above) keep HBM at ~2.4 TB/s; here every lane moves 16 bytes, 512 contiguous bytes per
// channel row.  Lane (i, g) holds pixels 4 i .. 4 i + 3 of its 16 accumulator rows, so the tile is FOUR MFMA column blocks with
// the strided pixel assignment block cb = pixels {4 i + cb}: component cb of a lane's 16-byte load is exactly its accumulator
// element of block cb - no transposition anywhere.  64 tile registers; the two conditioning convolutions run one after the other
// (one 16-register accumulator): S first, applied in place with the normalisation, then Bv, added, then 16-byte stores.
// The last tile of a row may be ragged (H W % 128 != 0): its lanes beyond the row read zeros / store nothing (range-checked
// descriptors sized to the sample, out-of-range offsets for those lanes).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned CLN_OOB = 0x7ffffff0u;

CDEV auto sized_rsrc(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }

template <int NW, bool AFF>
__global__ __launch_bounds__(NW * 64, 4) void cln_mfma_wide_kernel(ClnMfmaArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cln_smem[];
    double* red = reinterpret_cast<double*>(cln_smem);                          // [2][NW][128]
    float* stat = reinterpret_cast<float*>(cln_smem + NW * 2048);               // [2][128]
    half8* Bf = reinterpret_cast<half8*>(cln_smem + NW * 2048 + 1024);          // [k-step][column block][hi | lo][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int b = blockIdx.y;
    const long px = (long)blockIdx.x * 128 + 4 * i;                             // this lane's four pixels
    const bool valid = px < p.HW;
    const unsigned HWb = (unsigned)p.HW * 4u;
    const unsigned pxb = valid ? (unsigned)px * 4u : CLN_OOB;
    const auto rsx = sized_rsrc(p.x + (long)b * p.sx, (unsigned)p.C * HWb);
    const auto rsy = sized_rsrc(p.y + (long)b * p.sx, (unsigned)p.C * HWb);
    const unsigned raw_c = p.cslot ? slot_load(p.cslot + lane) : 0u;

    // one per-lane offset (row tile base + 4 g rows, this lane's pixels); register r's rows are a UNIFORM distance on: soffset operand
    // (the range check looks at the per-lane part only - CLN_OOB alone is beyond every sample)
    const unsigned vbase = valid ? (unsigned)(wave * 32 + 4 * g) * HWb + pxb : CLN_OOB;
    float xr[4][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)vbase, (int)((unsigned)acc_row(r, 0) * HWb), 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) xr[cb][r] = __uint_as_float(v[cb]);
    }

    float inv_c = 1.f;
    const int nk = (p.J + 15) / 16;
    const float cmax = p.As ? wave_max_bits(raw_c) : 0.f;
    if (p.As) {
        const int ec = pow2_exponent_for(cmax);
        const float cscale = ldexpf(1.0f, ec);
        inv_c = ldexpf(1.0f, -ec);
        const auto rsc = sized_rsrc(p.cond + (long)b * p.scond, (unsigned)p.J * HWb);
        // one (k-step, element) row of the conditioning field per wave and turn: 16 bytes per lane, split, eight 2-byte LDS stores
        _Float16* Bh = reinterpret_cast<_Float16*>(Bf);
        for (int item = wave; item < nk * 8; item += NW) {
            const int ks = item >> 3, e = item & 7, j = 16 * ks + 8 * g + e;
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsc, (int)((valid && j < p.J) ? (unsigned)(8 * g) * HWb + pxb : CLN_OOB),
                                                                    (int)((unsigned)(16 * ks + e) * HWb), 0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const float v = __uint_as_float(raw[cb]) * cscale;
                const _Float16 h = (_Float16)v;
                Bh[((((ks * 4 + cb) * 2 + 0) * 64 + lane) << 3) + e] = h;
                Bh[((((ks * 4 + cb) * 2 + 1) * 64 + lane) << 3) + e] = (_Float16)(v - (float)h);
            }
        }
    }

    // ---- per-pixel statistics (fp64, fixed order: lane, lane ^ 32, waves 0 .. NW - 1); pixel 4 i + cb lives in slot cb * 32 + i
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s += (double)xr[cb][r]; ss += (double)xr[cb][r] * (double)xr[cb][r]; }
        s += __shfl_xor(s, 32, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) { red[(0 * NW + wave) * 128 + cb * 32 + i] = s; red[(1 * NW + wave) * 128 + cb * 32 + i] = ss; }
    }
    __syncthreads();
    if (tid < 128) {
        double a = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a += red[(0 * NW + w) * 128 + tid]; q += red[(1 * NW + w) * 128 + tid]; }
        const double mu = a / p.C;
        double var = q / p.C - mu * mu;
        if (var < 0.0) var = 0.0;
        stat[tid] = (float)mu;
        stat[128 + tid] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    const float os = inv_c / p.ascale_s, ob = inv_c / p.ascale_b;
    const long ablk = (long)wave * nk * 1024 + lane * 8;

    // ---- normalise + elementwise affine in place (one gamma / beta load per row, used by the four column blocks)
    {
        const auto rsg = sized_rsrc(AFF ? p.gamma : p.x, (unsigned)p.C * 4u), rsb = sized_rsrc(AFF ? p.beta : p.x, (unsigned)p.C * 4u);
        float mu[4], rstd[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { mu[cb] = stat[cb * 32 + i]; rstd[cb] = stat[128 + cb * 32 + i]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ga = 1.f, be = 0.f;
            if (AFF) {
                ga = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsg, (wave * 32 + 4 * g) * 4, acc_row(r, 0) * 4, 0));
                be = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsb, (wave * 32 + 4 * g) * 4, acc_row(r, 0) * 4, 0));
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) { xr[cb][r] = (xr[cb][r] - mu[cb]) * rstd[cb] * ga + be; asm volatile("" : "+v"(xr[cb][r])); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- S = Ws cond per column block, applied in place: x (1 + S)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (p.As) {
            for (int ks = 0; ks < nk; ++ks) {
                const half8 bh = Bf[((ks * 4 + cb) * 2 + 0) * 64 + lane], bl = Bf[((ks * 4 + cb) * 2 + 1) * 64 + lane];
                const half8 ah = *reinterpret_cast<const half8*>(p.As + ablk + (long)ks * 1024), al = *reinterpret_cast<const half8*>(p.As + ablk + (long)ks * 1024 + 512);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { xr[cb][r] *= 1.0f + acc[r] * os; asm volatile("" : "+v"(xr[cb][r])); }   // pinned here: sunk past the
        __builtin_amdgcn_sched_barrier(0);                                                                        // other blocks, four accumulators stay live
    }
    // ---- Bv = Wb cond per column block, added
    if (p.As) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int ks = 0; ks < nk; ++ks) {
                const half8 bh = Bf[((ks * 4 + cb) * 2 + 0) * 64 + lane], bl = Bf[((ks * 4 + cb) * 2 + 1) * 64 + lane];
                const half8 ah = *reinterpret_cast<const half8*>(p.Ab + ablk + (long)ks * 1024), al = *reinterpret_cast<const half8*>(p.Ab + ablk + (long)ks * 1024 + 512);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { xr[cb][r] += acc[r] * ob; asm volatile("" : "+v"(xr[cb][r])); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.y) {
        float vmax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            u32x4 o;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) { o[cb] = __float_as_uint(xr[cb][r]); vmax = fmaxf(vmax, fabsf(xr[cb][r])); }
            __builtin_amdgcn_raw_buffer_store_b128(o, rsy, (int)vbase, (int)((unsigned)acc_row(r, 0) * HWb), 0);
        }
        if (p.omax && !p.Phi) {
            if (!valid) vmax = 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
            if (lane == 0) atomicMax(p.omax + ((blockIdx.x + wave) & 63), __float_as_uint(vmax));
        }
    }
    if (p.Phi) {
        // ---- the same tile as P-format planes.  v_permlane32_swap(a, b) exchanges a[lanes 32..63] with b[lanes 0..31]: swapping
        // registers (e, 4 + e) and (8 + e, 12 + e) leaves lane (i, g) with channels 8 g .. 8 g + 7 of its row tile in registers
        // 0..7 and 16 + 8 g .. + 7 in 8..15 - whole 16-byte entries (8 channels of one pixel) of channel groups g and 2 + g.
        const float bound = ((sqrtf((float)p.C) * p.gmax + p.bmax) * (1.0f + p.ws_inf * cmax) + p.wb_inf * cmax) * 1.0001f;
        const float ys = ldexpf(1.0f, pow2_exponent_for(bound));
        if (p.omax && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) p.omax[lane] = __float_as_uint(bound);
        const auto rsh = sized_rsrc(p.Phi + (long)b * p.sP, (unsigned)p.C * (HWb / 2));
        const auto rsl = sized_rsrc(p.Plo + (long)b * p.sP, (unsigned)p.C * (HWb / 2));
        // A lane holds FOUR ADJACENT pixels (64 contiguous bytes of a plane), so storing its entries directly makes every store
        // instruction a 64-byte-strided scatter of 16-byte pieces - quarter-filled L2 write requests, +25 us per norm.  Each wave
        // turns its entries through a private LDS buffer instead ([column block][36] entries: pitch 36 is conflict-free for the
        // 16-byte writes and the transposed reads), so that store c covers pixels 32 c .. 32 c + 31 with one entry per lane.
        // The buffer reuses the statistics / B-fragment LDS: every wave is past them after this barrier.
        __syncthreads();
        half8* tb = reinterpret_cast<half8*>(cln_smem) + (wave * 2 + g) * 144;
        const long tile0 = (long)blockIdx.x * 128;
        unsigned vst[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
            vst[c4] = tile0 + 32 * c4 + i < p.HW ? ((unsigned)(wave * 4 + g) * (unsigned)p.HW + (unsigned)(tile0 + 32 * c4 + i)) * 16u : CLN_OOB;
        const int rd = (i & 3) * 36 + (i >> 2);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xr[cb][8 * hq + e]), __float_as_uint(xr[cb][8 * hq + 4 + e]), false, false);
                    xr[cb][8 * hq + e] = __uint_as_float(sw[0]);
                    xr[cb][8 * hq + 4 + e] = __uint_as_float(sw[1]);
                }
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const int so = (int)((unsigned)(2 * hq) * (unsigned)p.HW * 16u);
            half8 ll[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                half8 hh;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = __builtin_amdgcn_fmed3f(xr[cb][8 * hq + e] * ys, -65504.f, 65504.f);
                    const _Float16 a = (_Float16)x;
                    hh[e] = a;
                    ll[cb][e] = (_Float16)(x - (float)a);
                }
                tb[cb * 36 + i] = hh;
            }
            wave_lds_sync();
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tb[rd + 8 * c4]), rsh, (int)vst[c4], so, 0);
            wave_lds_sync();
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) tb[cb * 36 + i] = ll[cb];
            wave_lds_sync();
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tb[rd + 8 * c4]), rsl, (int)vst[c4], so, 0);
            wave_lds_sync();
        }
    }
}

size_t cln_wide_lds_bytes(int NW, int J) { return std::max((size_t)NW * 2048 + 1024 + (size_t)((J + 15) / 16) * 8192, (size_t)NW * 2 * 144 * 16); }
bool cln_wide_ok(const ClnMfmaArgs& a) { return (a.C == 256 || a.C == 512) && a.HW % 4 == 0; }

}  // namespace

bool cln_mfma_planes_ok(const ClnMfmaArgs& a) { return cln_wide_ok(a); }

bool cln_mfma_eligible(const ClnMfmaArgs& a) {
    if (a.Phi && !(a.Plo && cln_wide_ok(a))) return false;
    const bool common = a.x && (a.y || a.Phi) && a.C % 256 == 0 && a.C / 256 >= 1 && a.C / 256 <= 4 && a.HW >= 1 && a.nbatch >= 1 && a.nbatch <= 65535 &&
                        (double)a.C * (double)a.HW * 4.0 < 2147483000.0 && (double)(a.J > 0 ? a.J : 1) * (double)a.HW * 4.0 < 2147483000.0 &&
                        (!a.As || (a.Ab && a.cond && a.cslot && a.J >= 1 && a.J <= 16 * CLN_MAX_NK)) && (!a.gamma || a.beta);
    return common && (cln_wide_ok(a) || a.HW % 32 == 0);
}

hipError_t launch_cln_mfma(const ClnMfmaArgs& a, hipStream_t s) {
    if (!cln_mfma_eligible(a)) return hipErrorInvalidValue;
    if (cln_wide_ok(a)) {
        const int NW = a.C / 32;
        const size_t lds = cln_wide_lds_bytes(NW, a.As ? a.J : 0);
        dim3 grid((unsigned)((a.HW + 127) / 128), (unsigned)a.nbatch), block(NW * 64);
        const bool aff = a.gamma != nullptr;
        auto go = [&](auto kern) -> hipError_t {
            static bool raised = false;   // (one flag per instantiation of this generic lambda)
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cln_wide_lds_bytes(NW, 16 * CLN_MAX_NK));
                if (e != hipSuccess) return e;
                raised = true;
            }
            hipLaunchKernelGGL(kern, grid, block, lds, s, a);
            return hipSuccess;
        };
        hipError_t e = NW == 8 ? (aff ? go(cln_mfma_wide_kernel<8, true>) : go(cln_mfma_wide_kernel<8, false>))
                               : (aff ? go(cln_mfma_wide_kernel<16, true>) : go(cln_mfma_wide_kernel<16, false>));
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    dim3 grid((unsigned)(a.HW / 32), (unsigned)a.nbatch), block(512);
    // (C = 256 / 512 never come here: same-box A/B at C = 512, 114 us per norm in this form against 86 in the 128-pixel one)
    if (a.C / 256 == 3) hipLaunchKernelGGL(cln_mfma_kernel<3>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(cln_mfma_kernel<4>, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace ace
