// First MLP convolution (layers.py:117-124: GELU(W1 . x + b1), the widest of the block's 1x1 convolutions: M = 768 rows) with
// the WEIGHTS in LDS, NO synchronisation in the main loop and the epilogue riding in the MFMA stream, for gfx950:
//     U = GELU( W1f . P(t) + b1f )        W1f: (M x K) norm-folded weight, t: K x HW as P-format fp16 hi/lo planes, U likewise
// conv_ws.hip keeps the weights in registers and streams the activation through an LDS ring shared by eight waves: a barrier,
// an accumulator exchange between the two contraction halves and 6 LDS-DMA issues per wave and 36-MFMA stage, all in lock
// step - r03 same-box variants (two accumulators, earlier stores, all 256 CUs) moved its time by < 2 %, the PMC view is one
// third issuing / one third issue-stalled / one third waiting at the barrier, matrix pipe 40 % busy.  Here the roles are
// swapped:
//   * a workgroup owns 32 RT output rows; their A fragments (strip_pack.h order 0: one contiguous 2 KiB x RT x K/16 chunk of
//     the packed weight) are copied to LDS once (144 KiB at K = 384, RT = 3) and only READ afterwards;
//   * every wave works alone: for its pixel tile it loads the B fragments straight from the P-format planes (a lane's 16
//     bytes ARE its fragment: coalesced global_load_dwordx4, no LDS, no DMA), a few k-steps ahead, and feeds each to 3 RT
//     MFMAs on RT independent accumulators; then the GELU / split epilogue of its tile and 16-byte P-entry stores;
//   * the GELU / split epilogue of tile n - 1 is cut into quarters of a value (~7 VALU each) and one quarter follows each MFMA
//     of tile n: on gfx950 vector-ALU work hides behind matrix work only inside ONE wave's instruction stream (a second wave's
//     VALU instructions do not issue while the first keeps the matrix pipe fed: profiles/r03_mfma_valu_overlap_probe.txt);
//     the first version of this kernel (12 waves, epilogue after the MFMAs) ran at conv_ws.hip's 134 us;
//   * 8 waves per workgroup (2 per SIMD, <= 256 registers: two sets of accumulators), no barrier to align them;
//   * the 8 row slices (M = 768) that need the same pixel tiles sit on one XCD and walk that XCD's tile range in the same
//     order, so the activation comes from HBM once and from L2 otherwise; 8 slices x 4 pixel groups = the XCD's 32 CUs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "strip_common.h"

namespace ace {
namespace {

#ifndef ACE_WL_WAVES
#define ACE_WL_WAVES 8
#endif
#ifndef ACE_WL_D
#define ACE_WL_D 4
#endif
#ifndef ACE_WL_ABL
#define ACE_WL_ABL 0                   // measurement builds only (wrong results): bit 0 no epilogue chunks, bit 1 no B-fragment loads in the loop
#endif
constexpr int WL_WAVES = ACE_WL_WAVES;   // 8: two per SIMD, <= 256 registers each
#ifdef ACE_X_TRACE   // measurement builds: first / last s_memtime of every workgroup's wave 0, its XCC id
__device__ unsigned long long wl_wg_span[4096][3];
#define WG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) wl_wg_span[blockIdx.x][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif
constexpr int WL_OOBV = 0x7fffff00;

// KS: k16-steps (K / 16), RT: 32-row tiles per workgroup, D: k-steps of B-fragment read-ahead
// FOLD: the instance norm in front of this convolution (norm1 of the block) is folded into the weights while the workgroup copies
// its slice to LDS - W diag(a) scaled by 2^(12 - exponent(max|W| max|a|)) and split into hi / lo fragments, bias + W b - from the
// fp32 weight and the norm's affine: element for element the arithmetic of pack_conv_frag_kernel, whose launch (7 us behind
// every norm1) it replaces.
template <int KS, int RT, int D, bool FOLD = false>
__global__ __launch_bounds__(64 * WL_WAVES) void conv_wl_kernel(ConvStripArgs p, int nslice, int groups_per_xcd, int tpx) {
    constexpr int FTB = FOLD ? (2 * 16 * KS + WL_WAVES * 32 * RT + 16) * 4 : 0;   // a | b | per-wave partial bias dots | max|a| per wave
    __shared__ __attribute__((aligned(16))) char smem[RT * KS * 2048 + 32 * RT * 4 + FTB];   // the slice's fragments, then its bias
    float* Pb = reinterpret_cast<float*>(smem + RT * KS * 2048);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;

    // ---- workgroup -> (sample, XCD, pixel group of the XCD, row slice); block b runs on XCD b % 8
    const int per_xcd = nslice * groups_per_xcd;
    const int per_smp = 8 * per_xcd;
    const int smp = blockIdx.x / per_smp;
    const int bb = blockIdx.x % per_smp;
    const int xcd = bb & 7, w = bb >> 3;
    const int slice = w % nslice, grp = w / nslice;
    const int tiles_px = (p.HW + 31) / 32;
    const int x0 = xcd * tpx;
    const int x1 = x0 + tpx < tiles_px ? x0 + tpx : tiles_px;

    MT(0);
    WG_STAMP(0);
    const unsigned raw_x = slot_load(p.xslot + lane);
    const unsigned raw_a = (!FOLD && p.aslot) ? slot_load(p.aslot + lane) : 0u;
    const unsigned raw_c = p.cinb ? slot_load(p.cinb + lane) : 0u;

    // ---- weights of this slice -> LDS (the slice's fragments are one contiguous chunk of the packed operand)
    float fscale = 1.f;
    if constexpr (FOLD) {
        static_assert(WL_WAVES == 8 && KS % 8 == 0, "fold: wave w takes the k-steps w, w + 8, ... of every row tile");
        float* Ta = Pb + 32 * RT;                 // a[0 .. K), b[0 .. K)
        float* Tb = Ta + 16 * KS;
        float* Pd = Tb + 16 * KS;                 // [wave][32 RT] partial dots of W b
        float* red = Pd + WL_WAVES * 32 * RT;
        const float* fa = p.fa + (long)smp * p.sfa;
        const float* fb = p.fb + (long)smp * p.sfa;
        float am = 0.f;
        for (int r = tid; r < 16 * KS; r += 64 * WL_WAVES) {
            const float av = fa[r];
            Ta[r] = av; Tb[r] = fb[r];
            am = fmaxf(am, fabsf(av));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
        if (lane == 0) red[wave] = am;
        __syncthreads();
        am = red[0];
#pragma unroll
        for (int k = 1; k < WL_WAVES; ++k) am = fmaxf(am, red[k]);
        fscale = ldexpf(1.0f, pow2_exponent_for(p.wabs * am));   // this sample's own scale (pack_conv_frag_kernel: one over the batch)
        // block (t, J): lane (i, g) owns row 32 (slice RT + t) + i, columns 16 J + 8 g + e
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float* Wr = p.Wraw + (long)(32 * (slice * RT + t) + i) * p.ldw + 8 * g;
            float dot = 0.f;
#pragma unroll
            for (int jj = 0; jj < KS / 8; ++jj) {
                const int J = wave + 8 * jj;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wr + 16 * J), w1 = *reinterpret_cast<const f32x4*>(Wr + 16 * J + 4);
                const int col = 16 * J + 8 * g;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ta + col), a1 = *reinterpret_cast<const f32x4*>(Ta + col + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(Tb + col), b1 = *reinterpret_cast<const f32x4*>(Tb + col + 4);
                half8 fh, fl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float wv = e < 4 ? w0[e & 3] : w1[e & 3];
                    dot = fmaf(wv, e < 4 ? b0[e & 3] : b1[e & 3], dot);
                    float x = wv * fscale;
                    x *= e < 4 ? a0[e & 3] : a1[e & 3];
                    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
                    const _Float16 hv = (_Float16)x;
                    fh[e] = hv;
                    fl[e] = (_Float16)(x - (float)hv);
                }
                *reinterpret_cast<half8*>(smem + (t * KS + J) * 2048 + lane * 16) = fh;
                *reinterpret_cast<half8*>(smem + (t * KS + J) * 2048 + 1024 + lane * 16) = fl;
            }
            dot += __shfl_xor(dot, 32, 64);
            if (g == 0) Pd[wave * (32 * RT) + 32 * t + i] = dot;
        }
        __syncthreads();
        if (tid < 32 * RT) {   // folded bias: bias + W b, the eight waves' partial sums in a fixed order
            float acc = Pd[tid];
#pragma unroll
            for (int k = 1; k < WL_WAVES; ++k) acc += Pd[k * (32 * RT) + tid];
            Pb[tid] = p.bias[slice * 32 * RT + tid] + acc;
        }
    } else {
        const char* A = reinterpret_cast<const char*>(p.A + (long)smp * p.sA) + (long)slice * RT * KS * 2048;
        for (int o = tid * 16; o < RT * KS * 2048; o += 64 * WL_WAVES * 16)
            *reinterpret_cast<u32x4*>(smem + o) = *reinterpret_cast<const u32x4*>(A + o);
        const float* b = p.bias + (long)smp * p.sbias + slice * 32 * RT;
        if (tid < 32 * RT) Pb[tid] = b[tid];
    }
    const float xbound = wave_max_bits(raw_x);
    const float inv_x = ldexpf(1.0f, -pow2_exponent_for(xbound));
    const float inv_a = FOLD ? 1.0f / fscale : (p.aslot ? ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_a))) : 1.0f / p.ascale);
    const float s_acc = inv_x * inv_a;
    // bound of this launch's output, identical in every workgroup; the consumer reads it from cslot
    const float inb = p.cinb ? wave_max_bits(raw_c) : xbound;
    const float cbound = fmaf(p.cw, inb, p.cb);
    const float cscale = ldexpf(1.0f, pow2_exponent_for(cbound));
    if (tid == 0) atomicMax(p.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
    __syncthreads();
    MT(1);
    int tev = 2;   // (measurement builds: timeline stamps of wave 0 - tile start, k-loop done)

    const _Float16* Xh = p.Xhi + (long)smp * p.sX;
    const _Float16* Xl = p.Xlo + (long)smp * p.sX;
    const int pbytes = p.M * p.HW * 2;
    const int xbytes = p.C * (int)p.ldn * 2;     // one input plane: C / 8 k-groups x ldn entries of 16 bytes
    const auto rsXh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Xh), 0, xbytes, 0x00020000);
    const auto rsXl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Xl), 0, xbytes, 0x00020000);
    const auto rsH = __builtin_amdgcn_make_buffer_rsrc(p.Chi + (long)smp * p.sCp, 0, pbytes, 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(p.Clo + (long)smp * p.sCp, 0, pbytes, 0x00020000);
    const char* aw = smem + lane * 16;

    // ---- the finished tile whose epilogue rides in the current tile's MFMA stream: rows in k-group order (rows_to_kgroups),
    //      lane (i, g) holds rows 8 g + e (r0..r7) and 16 + 8 g + e (r8..r15) of its pixel column: two whole P entries
    //      (k-groups 4 T + g and 4 T + 2 + g) per row tile.  The first tile carries a dummy (stores suppressed).
    f32x16 prev[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[t][r] = 0.f;
    int prev_vo = WL_OOBV;
    GeluStage gst = {0.f, 0.f, 0.f};
    unsigned hh[4] = {0u, 0u, 0u, 0u}, ll[4] = {0u, 0u, 0u, 0u};   // the P entry being assembled: fp16 hi / lo of rows 8 g + e, packed
    float bias_next = Pb[8 * g];
    // quarter `ch` of value v = 16 t + 8 q + e of the finished tile (gfx950 hides vector-ALU work behind matrix work only when
    // the two alternate in ONE wave's instruction stream, ~6 VALU per MFMA: profiles/r03_mfma_valu_overlap_probe.txt)
    auto chunk = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int v = k / 4, ch = k % 4, t = v / 16, q = (v % 16) / 8, e = v % 8;
        if constexpr (ch == 0) {
            gelu_stage0(gst, fmaf(prev[t][8 * q + e], s_acc, bias_next));
            asm volatile("" : "+v"(gst.val), "+v"(gst.t));                // pinned to this slot (pure arithmetic would
        } else if constexpr (ch == 1) {                                    //  otherwise sink to its last use)
            gelu_stage1(gst);
            constexpr int vn = (v + 1) % (16 * RT);                          // bias of the next value, one value ahead
            bias_next = Pb[32 * (vn / 16) + 16 * ((vn % 16) / 8) + 8 * g + vn % 8];
            asm volatile("" : "+v"(gst.q));
        } else if constexpr (ch == 2) {
            gst.val = gelu_stage2(gst);
            asm volatile("" : "+v"(gst.val));
        } else {
            split_put<e>(hh, ll, cscale, gst.val);
            if constexpr (e == 7) {   // the entry is complete: k-group 4 T + 2 q (+ g in the lane part of the offset)
                const int soff = (4 * (slice * RT + t) + 2 * q) * p.HW * 16;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{hh[0], hh[1], hh[2], hh[3]}, rsH, prev_vo, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ll[0], ll[1], ll[2], ll[3]}, rsL, prev_vo, soff, 0);
            }
        }
    };
    constexpr int NCH = 4 * 16 * RT;     // chunks per tile
    constexpr int NM = 9 * KS;           // MFMA slots per tile (9 = 3 products x RT at RT = 3; see below)
    static_assert(RT == 3 || RT == 2, "slot arithmetic below");
    constexpr int MPS = 3 * RT;          // MFMAs per k-step
    auto slot = [&](auto mc) {           // what rides behind MFMA m of the tile
        constexpr int m = decltype(mc)::value;
        constexpr int NMT = MPS * KS;
        if constexpr (!(ACE_WL_ABL & 1)) static_for<(m * NCH) / NMT, ((m + 1) * NCH) / NMT>([&](auto kc) { chunk(kc); });
        __builtin_amdgcn_sched_barrier(0);
    };
    (void)NM;

    // ---- this wave's pixel tiles: x0 + (grp * 8 + wave) + 8 groups_per_xcd * n
    const int stride = WL_WAVES * groups_per_xcd;
    for (int tile = x0 + grp * WL_WAVES + wave; tile < x1; tile += stride) {
        MT(tev);
        int n = 32 * tile + i;
        const bool nok = n < p.HW;
        n = nok ? n : p.HW - 1;
        // k-group 2 j + g of k-step j: lane part (g ldn + n) entries of 16 bytes, k-step part 2 j ldn entries - through buffer
        // descriptors, the k-step as the SCALAR offset (as 64-bit lane addresses every load cost a v_lshl_add_u64: one per value)
        const int bvo = (g * (int)p.ldn + n) * 16;
        const int kstep = 2 * (int)p.ldn * 16;
        f32x16 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        half8 bh[D], bl[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            bh[d] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsXh, bvo, d * kstep, 0));
            bl[d] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsXl, bvo, d * kstep, 0));
        }
        // A fragments double-buffered by hand and scheduling barriers: left alone, hipcc hoists all the LDS reads of the
        // unrolled loop to the top (554 VGPRs spilled in the first version)
        half8 ah[2][RT], al[2][RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            ah[0][t] = *reinterpret_cast<const half8*>(aw + (t * KS) * 2048);
            al[0][t] = *reinterpret_cast<const half8*>(aw + (t * KS) * 2048 + 1024);
        }
        static_for<0, KS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const half8 ch = bh[j % D], cl = bl[j % D];
            if constexpr (j + 1 < KS) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    ah[(j + 1) & 1][t] = *reinterpret_cast<const half8*>(aw + (t * KS + j + 1) * 2048);
                    al[(j + 1) & 1][t] = *reinterpret_cast<const half8*>(aw + (t * KS + j + 1) * 2048 + 1024);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // the next step's fragment reads go out BEFORE this step's MFMAs, which cover them
            static_for<0, RT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j & 1][t], ch, acc[t], 0, 0, 0);
                slot(std::integral_constant<int, MPS * j + t>{});
            });
            static_for<0, RT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][t], cl, acc[t], 0, 0, 0);
                slot(std::integral_constant<int, MPS * j + RT + t>{});
            });
            static_for<0, RT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][t], ch, acc[t], 0, 0, 0);
                if constexpr (t == RT - 1 && j + D < KS && !(ACE_WL_ABL & 2)) {   // this k-step's B registers are free: the fragment of step j + D
                    bh[j % D] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsXh, bvo, (j + D) * kstep, 0));
                    bl[j % D] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsXl, bvo, (j + D) * kstep, 0));
                }
                slot(std::integral_constant<int, MPS * j + 2 * RT + t>{});
            });
        });
        MT(tev + 1);
        tev += 2;
        // ---- this tile becomes the finished one
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            rows_to_kgroups(acc[t]);
            prev[t] = acc[t];
        }
        prev_vo = nok ? ((g * p.HW + 32 * tile + i) * 16) : WL_OOBV;
    }
    // ---- the last tile's epilogue, on its own
    static_for<0, NCH>([&](auto kc) { chunk(kc); });
    MT(tev);
    WG_STAMP(1);
#ifdef ACE_X_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        wl_wg_span[blockIdx.x][2] = xcc & 0xf;
    }
#endif
}

template <int KS, int RT, int D, bool FOLD = false>
hipError_t launch_wl(const ConvStripArgs& a, hipStream_t s) {
    const int nslice = a.M / (32 * RT);
    int gpx = 32 / nslice;                      // pixel groups per XCD (32 CUs each)
    if (gpx < 1) gpx = 1;
    const int tiles_px = (int)((a.HW + 31) / 32);
    const int tpx = (tiles_px + 7) / 8;
    if (gpx * WL_WAVES > tpx) gpx = (tpx + WL_WAVES - 1) / WL_WAVES;   // small fields: no idle workgroups
    dim3 grid((unsigned)(8 * nslice * gpx * a.nbatch)), block(64 * WL_WAVES);
    hipLaunchKernelGGL((conv_wl_kernel<KS, RT, D, FOLD>), grid, block, 0, s, a, nslice, gpx, tpx);
    return hipGetLastError();
}

}  // namespace

#ifdef ACE_X_TRACE   // measurement builds only (tools/trace_wl.py): the s_memtime stamps of wave 0 of workgroup ACE_X_TRACE
extern "C" int ace_debug_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mlp_trace), sizeof(mlp_trace)); }
extern "C" int ace_debug_wg_spans(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wl_wg_span), sizeof(wl_wg_span)); }
#endif

// K: input channels, M: output channels.  GELU + P-format output, no residual, no statistics (the first MLP convolution)
bool conv_wl_eligible(int K, int M, long HW) {
    if (K == 384) return M % 96 == 0 && M / 96 <= 32 && (long)M * HW * 2 < 0x7fffff00L;
    if (K == 512 || K == 256 || K == 128) return M % 64 == 0 && M / 64 <= 32 && (long)M * HW * 2 < 0x7fffff00L;   // two row tiles per workgroup (K = 512: 128 KiB of weights)
    return false;
}

hipError_t launch_conv_wl(const ConvStripArgs& a, hipStream_t s) {
    if (!conv_wl_eligible(a.C, a.M, a.HW) || !a.bias || !a.xslot || !(a.A || a.Wraw) || !a.Chi || !a.Clo || !a.cslot || a.R || a.part || a.Cf ||
        !(a.act == ACT_GELU || a.act == ACT_GELU_FAST))
        return hipErrorInvalidValue;
    if (a.Wraw) {   // weight fold in the prologue (the ACE2 width): fp32 weight with 16-byte rows, the norm's affine, the raw bias
        if (!a.fa || !a.fb || (a.ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(a.Wraw) & 15) != 0 || a.sbias != 0 || a.C != 384) return hipErrorInvalidValue;
        return launch_wl<24, 3, ACE_WL_D, true>(a, s);
    }
    switch (a.C) {
        case 512: return launch_wl<32, 2, 4>(a, s);
        case 384: return launch_wl<24, 3, ACE_WL_D>(a, s);
        case 256: return launch_wl<16, 2, 4>(a, s);
        case 128: return launch_wl<8, 2, 4>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ace
