// 1x1 convolution with the INPUT strip resident in registers, for gfx950:  out = act(A . P(x) + bias [+ R])  written as
// P-format fp16 hi/lo planes (+ row statistics), the form the block's inner skip (sfnonet.py:229-232) and the first MLP
// convolution (layers.py:117-124) take on the packed path.  A is (M x C) with C = 128 / 256 / 384.
//
// Same idea as the Legendre strip kernels (strip.hip), sized for two waves per SIMD:
//   * a wave owns a 32-pixel strip: the C x 32 input (P-format planes written by the producer) is loaded once as MFMA B
//     fragments, 192 VGPRs at C = 384, and stays there while the wave walks all M / 32 output tiles;
//   * eight waves (256 pixels) share the weight stream: one output tile = 2 C / 16 KiB of pre-packed A fragments
//     (strip_pack.h layout, instance-norm affine folded in by pack_conv_frag_kernel), two-slot LDS ring, the pieces of tile
//     t + 1 issued between the MFMAs of tile t.  256 pixels per weight fetch halve the L2 -> LDS traffic of a 128 x 128 tile
//     engine, and the 254 workgroups of a 180 x 360 field are one wave of workgroups on the 256 CUs;
//   * the epilogue never leaves the registers: four v_permlane32_swap turn the accumulator tile into whole 8-row P entries
//     (bias, residual, activation, split to fp16 hi/lo, two 16-byte stores per plane); row statistics for the next
//     instance norm go through a per-wave LDS transpose;
//   * the second wave of each SIMD hides the DMA issue slots, the LDS latency and the epilogue of its partner (the fused MLP
//     kernel, one wave per SIMD, pays for each of them: profiles/r02_mlp_ablation.txt).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "strip_common.h"

namespace ace {
namespace {

// NC = C / 32; RES: fp32 residual added before the activation; STATS: per-(32-pixel strip, row) statistics
template <int NC, int ACT, bool RES, bool STATS>
__global__ __launch_bounds__(512, 2) void conv_strip_kernel(ConvStripArgs p) {
    constexpr int KS = 2 * NC;              // k16-steps
    constexpr int SLOT = KS * 2048;         // one 32-row tile of A fragments
    constexpr int NSLOT = 2;
    constexpr int PW = KS / 4;              // 1-KiB pieces per wave per tile (2 KS pieces, 8 waves)
    constexpr int BMAX = 2048;              // output rows whose bias is kept in LDS
    constexpr int ST = STATS ? 8 * 33 * 32 * 4 : 0;
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLOT + BMAX * 4 + ST];
    float* bs = reinterpret_cast<float*>(smem + NSLOT * SLOT);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* St = reinterpret_cast<float*>(smem + NSLOT * SLOT + BMAX * 4) + wave * (33 * 32);
    const int i = lane & 31, g = lane >> 5;
    const int wgs = (p.HW + 255) / 256;
    const int smp = blockIdx.x / wgs;
    const int n0 = (blockIdx.x % wgs) * 256 + wave * 32;
    const int n = n0 + i;
    const int nc = n < p.HW ? n : p.HW - 1;
    const bool nok = n < p.HW;
    const int ntiles = p.M / 32;

    const unsigned raw_x = slot_load(p.xslot + lane);
    const unsigned raw_a = p.aslot ? slot_load(p.aslot + lane) : 0u;
    const unsigned raw_c = p.cinb ? slot_load(p.cinb + lane) : 0u;
    const unsigned raw_r = p.rmax ? slot_load(p.rmax + lane) : 0u;

    const _Float16* A = p.A + (long)smp * p.sA;
    auto piece = [&](int t, int k) {   // piece k (of PW) of this wave for tile t -> slot t % NSLOT; tiles past the end re-fetch the last
        const int tt = t < ntiles ? t : ntiles - 1;
        const int pc = wave + 8 * k;
        glds16(A + (long)tt * (KS * 1024) + pc * 512 + lane * 8, smem + (t % NSLOT) * SLOT + pc * 1024);
    };
#pragma unroll
    for (int k = 0; k < PW; ++k) piece(0, k);

    // ---- resident input strip
    half8 xh[KS], xl[KS];
    {
        const _Float16* Xh = p.Xhi + (long)smp * p.sX;
        const _Float16* Xl = p.Xlo + (long)smp * p.sX;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const long off = ((long)(2 * j + g) * p.ldn + nc) * 8;
            xh[j] = *reinterpret_cast<const half8*>(Xh + off);
            xl[j] = *reinterpret_cast<const half8*>(Xl + off);
        }
    }
    {   // bias of this sample -> LDS
        const float* b = p.bias + (long)smp * p.sbias;
        float bv[BMAX / 512];
#pragma unroll
        for (int k = 0; k < BMAX / 512; ++k) bv[k] = b[(tid + 512 * k) < p.M ? tid + 512 * k : 0];
#pragma unroll
        for (int k = 0; k < BMAX / 512; ++k)
            if (tid + 512 * k < p.M) bs[tid + 512 * k] = bv[k];
    }
    const float xbound = wave_max_bits(raw_x);
    const float inv_x = ldexpf(1.0f, -pow2_exponent_for(xbound));
    const float inv_a = p.aslot ? ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_a))) : 1.0f / p.ascale;
    const float s_acc = inv_x * inv_a;
    // bound of this launch's output, identical in every workgroup; the consumer reads it from cslot
    const float inb = p.cinb ? wave_max_bits(raw_c) : xbound;
    const float resb = p.rmax ? wave_max_bits(raw_r) : 0.f;
    const float cbound = fmaf(p.cw, inb, p.cb) + resb;
    const float cscale = ldexpf(1.0f, pow2_exponent_for(cbound));
    if (tid == 0) atomicMax(p.cslot + (blockIdx.x & 63), __float_as_uint(cbound));

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tiles 0, 1, the input strip, the bias
#pragma unroll
    for (int j = 0; j < KS; ++j) asm volatile("" : "+v"(xh[j]), "+v"(xl[j]));
    __syncthreads();

    const auto rsR = __builtin_amdgcn_make_buffer_rsrc(RES ? const_cast<float*>(p.R + (long)smp * p.sR) : nullptr, 0,
                                                       RES ? p.M * p.HW * 4 : 0, 0x00020000);
    const int voff4 = (4 * g * p.HW + nc) * 4;     // accumulator layout: lane half g starts 4 rows down
    const int rowb = p.HW * 4;
    _Float16* Chi = p.Chi + (long)smp * p.sCp;
    _Float16* Clo = p.Clo + (long)smp * p.sCp;
    const int ncols_ok = p.HW - n0 < 32 ? (p.HW - n0 > 0 ? p.HW - n0 : 0) : 32;

    // Two slots, epilogue one tile late, residual through the accumulator.  Iteration t: barrier - epilogue of tile t - 1 -
    // residual of tile t loaded INTO the accumulator registers (pre-divided by the accumulator scale, an exact power of two:
    // acc * s + bias then carries the residual; no second register tile) - the PW pieces of tile t + 1 (into the slot of
    // tile t - 1, free since the barrier) - MFMAs of tile t.  vmcnt counts stores on gfx950 and loads / stores retire out
    // of order with respect to each other, so only vmcnt(0) proves that the pieces of a tile have landed; at the top of an
    // iteration everything it waits for was issued a whole tile of MFMAs earlier.  The residual loads are inline asm so
    // that THEIR wait can leave the pieces issued after them in flight (hipcc would wait vmcnt(0) at the first use of a
    // plain load: it does not see the LDS-DMA pieces).  The second wave of the SIMD runs its MFMAs meanwhile.
    auto epilogue = [&](int t, f32x16& v) {
        rows_to_kgroups(v);                // rows 8 g + e and 16 + 8 g + e: whole P entries
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const int row0 = 32 * t + 16 * hq + 8 * g;
            const f32x4 ba = *reinterpret_cast<const f32x4*>(bs + row0), bb = *reinterpret_cast<const f32x4*>(bs + row0 + 4);
            half8 hh, ll;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float val = fmaf(v[8 * hq + e], s_acc, e < 4 ? ba[e & 3] : bb[e & 3]);
                val = act_fn<ACT>(val);
                if (STATS) St[i * 33 + 16 * hq + 8 * g + e] = val;
                const float xs = val * cscale;
                const _Float16 a16 = (_Float16)xs;
                hh[e] = a16;
                ll[e] = (_Float16)(xs - (float)a16);
            }
            if (nok) {
                const long eo = ((long)(row0 >> 3) * p.HW + n) * 8;
                *reinterpret_cast<half8*>(Chi + eo) = hh;
                *reinterpret_cast<half8*>(Clo + eo) = ll;
            }
        }
        if (STATS) {
            // row statistics over this wave's 32 pixels (sum, sum of squares, min, max): value (row, column) sits at
            // column * 33 + row, lane (i, g) walks row i over the columns 16 g .. 16 g + 15, the half-waves meet in one exchange
            float sm = 0.f, sq = 0.f, mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                const float x = St[(16 * g + cc) * 33 + i];
                const bool ok = 16 * g + cc < ncols_ok;
                sm += ok ? x : 0.f;
                sq = ok ? fmaf(x, x, sq) : sq;
                mn = ok ? fminf(mn, x) : mn;
                mx = ok ? fmaxf(mx, x) : mx;
            }
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            mn = fminf(mn, __shfl_xor(mn, 32, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (g == 0) p.part[((long)smp * p.nstrips32 + (n0 >> 5)) * p.M + 32 * t + i] = make_float4(sm, sq, mn, mx);
        }
    };
    const float inv_s = 1.0f / s_acc;
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // tile t landed in every wave's share; every wave is done with tile t - 1
        if (t > 0) epilogue(t - 1, v);
        if (RES) {   // accumulator register r holds row acc_row(r, g) = (r & 3) + 8 (r >> 2) + 4 g of the tile, column i
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = (32 * t + (r & 3) + 8 * (r >> 2)) * rowb;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v[r]) : "v"(voff4), "s"(rsR), "s"(soff));
            }
        }
#pragma unroll
        for (int k = 0; k < PW; ++k) piece(t + 1, k);
        if (RES) {
            asm volatile("s_waitcnt vmcnt(%16)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                           "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                         : "n"(PW));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= inv_s;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
        }
        const unsigned sl = (unsigned)(size_t)(lds_cptr)(smem + (t % NSLOT) * SLOT) + lane * 16;
        pipelined_steps<KS, 1>(sl, [&](auto ss, const Frag& f) {
            constexpr int j = decltype(ss)::value;
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l, xh[j], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, xl[j], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, xh[j], v, 0, 0, 0);
        });
    }
    epilogue(ntiles - 1, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy refills of the tail land before the LDS is released
}

template <int NC>
hipError_t launch_nc(const ConvStripArgs& a, hipStream_t s) {
    const int wgs = (a.HW + 255) / 256;
    dim3 grid((unsigned)(wgs * a.nbatch)), block(512);
    const bool res = a.R != nullptr, stats = a.part != nullptr;
    if (a.act == ACT_GELU || a.act == ACT_GELU_FAST) {
        if (res && stats) hipLaunchKernelGGL((conv_strip_kernel<NC, ACT_GELU_FAST, true, true>), grid, block, 0, s, a);
        else if (!res && !stats) hipLaunchKernelGGL((conv_strip_kernel<NC, ACT_GELU_FAST, false, false>), grid, block, 0, s, a);
        else return hipErrorInvalidValue;
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

bool conv_strip_eligible(int C, int M, int act) {
    static const bool off = std::getenv("ACE_NO_CONV_STRIP") != nullptr;   // A/B switch for measurements
    if (off) return false;
    if (!(act == ACT_GELU || act == ACT_GELU_FAST)) return false;
    return (C == 128 || C == 256 || C == 384) && M % 32 == 0 && M >= 32 && M <= 2048;
}

hipError_t launch_conv_strip(const ConvStripArgs& a, hipStream_t s) {
    if (!conv_strip_eligible(a.C, a.M, a.act) || !a.Chi || !a.Clo || !a.cslot || !a.bias || !a.xslot) return hipErrorInvalidValue;
    switch (a.C / 32) {
        case 4: return launch_nc<4>(a, s);
        case 8: return launch_nc<8>(a, s);
        case 12: return launch_nc<12>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ace
