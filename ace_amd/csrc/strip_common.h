// Device-side building blocks shared by the register-resident "strip" kernels (mlp_strip.hip, conv_strip.hip): dynamic-range
// slot helpers, the LDS-DMA piece, the hand-pipelined A-fragment reader and the register epilogue pieces.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"

namespace ace {
namespace {

#define MDEV __device__ __forceinline__
#ifndef ACE_MLP_FDEPTH
#define ACE_MLP_FDEPTH 2
#endif
#ifndef ACE_MLP_ABL
#define ACE_MLP_ABL 0   // measurement only (wrong results): bit 0 no DMA in the loop, bit 1 no fragment reads, bit 2 no GELU,
#endif                  // bit 3 no barrier / DMA wait in the loop

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_cptr;

MDEV unsigned slot_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
MDEV int pow2_exponent_for(float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 12 - e; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
MDEV float wave_max_bits(unsigned raw) {
    float mx = __uint_as_float(raw);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
}
MDEV void glds16(const void* gsrc, const char* lds_dst_uniform) {
    // m0 = LDS base of the copy; declared clobbered instead of saved and restored around every piece (two scalar instructions
    // per piece: the compiler keeps nothing in m0 across these kernels' loops)
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)lds_dst_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(addr)
                 : "memory", "m0");
}
// The same piece with the address split into a wave-uniform base (SGPR pair: which k-group, plane, tile - scalar arithmetic) and a
// 32-bit lane offset that every piece of a tile shares: as one 64-bit lane address per piece the requests cost the wave eight vector
// instructions each (two of them quarter rate) - 760 .. 1050 cycles in front of every 36-MFMA stage of fc2 (tools/trace_ws.py).
MDEV void glds16s(const void* sbase_uniform, unsigned lane_off, const char* lds_dst_uniform) {
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)lds_dst_uniform);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_off), "s"(sbase_uniform), "s"(addr)
                 : "memory", "m0");
}
// ... with the LDS destination as a byte address (lds_addr_of(smem) + offset): the generic -> LDS pointer conversion of the char* forms
// above carries a null check (s_cmp_lg_u64 + s_cselect) that hipcc repeats for every piece - four scalar instructions per piece
MDEV unsigned lds_addr_of(const char* p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)p); }
MDEV void glds16a(const void* sbase_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_off), "s"(sbase_uniform), "s"(lds_addr_uniform)
                 : "memory", "m0");
}
MDEV float fast_erf(float x) {   // kernels.hip: max abs error 1.1e-7
    const float t = fminf(fabsf(x), 4.0f);
    float q = -1.150086973e-05f;
    q = fmaf(q, t, 1.518900972e-04f);
    q = fmaf(q, t, -8.436889620e-04f);
    q = fmaf(q, t, 2.264559502e-03f);
    q = fmaf(q, t, -7.151089812e-05f);
    q = fmaf(q, t, -2.773463540e-02f);
    q = fmaf(q, t, 1.483123451e-01f);
    q = fmaf(q, t, 9.184418917e-01f);
    q = fmaf(q, t, 1.627907395e+00f);
    q = q * t;
    return copysignf(1.0f - __builtin_amdgcn_exp2f(-q), x);
}
// GELU(v) = v / 2 (1 + erf(v / sqrt 2)) with the 1 / sqrt 2 folded into the fit's coefficients (c_k / 2^(k / 2): q is the same function of
// |v| / sqrt 2; max abs error of the GELU unchanged, tools/fit_fast_erf.py --gelu) - one multiplication and one register per value less
// than erf(v * 0.7071).  In three stages (identical operations in identical order wherever it is used), for epilogues that are spread
// over an MFMA stream a few instructions at a time: gfx950 hides vector-ALU work behind matrix work only when the two alternate in ONE
// wave's instruction stream, about 6 VALU per MFMA (profiles/r03_mfma_valu_overlap_probe.txt)
struct GeluStage { float val, t, q; };
MDEV void gelu_stage0(GeluStage& g, float val) {
    g.val = val;
    g.t = fminf(fabsf(val), 5.65685424949238f);  // erf(4) rounds to 1
}
MDEV void gelu_stage1(GeluStage& g) {
    float q = -5.082714551e-07f;
    q = fmaf(q, g.t, 9.493131074e-06f);
    q = fmaf(q, g.t, -7.457227184e-05f);
    q = fmaf(q, g.t, 2.830699377e-04f);
    q = fmaf(q, g.t, -1.264146067e-05f);
    q = fmaf(q, g.t, -6.933658849e-03f);
    q = fmaf(q, g.t, 5.243633315e-02f);
    q = fmaf(q, g.t, 4.592209458e-01f);
    q = fmaf(q, g.t, 1.151104331e+00f);
    g.q = q;
}
// v / 2 (1 + erf) = (v + |v| - |v| erfc(|v| / sqrt 2)) / 2 with erfc = 2^-q: an addition, a multiply-add and a multiplication where
// 1 - 2^-q, the copy of the sign, v / 2 and the multiply-add were four (max abs error 4.8e-7 on [-8, 8] against 6.5e-7; NaN and +Inf
// propagate, the tail beyond the fit's range is -t erfc(4) / 2 = -4e-8 instead of 0: tools/fit_fast_erf.py --gelu)
// (the planes' power-of-two scale does not belong into the 1 / 2: hipcc already folds it into the v_fma_mix conversions of the split,
// and without the multiplication in front of them it falls back to cvt / cvt / sub / cvt / perm: measured on the ISA, +2.5 per value)
MDEV float gelu_stage2(const GeluStage& g) {
    const float q = g.q * g.t;
    const float s = g.val + fabsf(g.val);
    return 0.5f * fmaf(-g.t, __builtin_amdgcn_exp2f(-q), s);
}

// hi / lo split of value E (0 .. 7) of a P entry, straight into the entry's packed words: H[E / 2], L[E / 2] get fp16(cs v) and
// fp16(cs v - hi) in their low (E even) or high (E odd) half, the other half is preserved.  The arithmetic is the one hipcc emits
// for `h = (_Float16)(cs * v); l = (_Float16)(cs * v - (float)h)` (v_fma_mix: the product is never rounded to fp32) - but written
// through half8 element inserts it costs 3.5 instructions per value (even values go through a temporary and a v_bfi each), here 2.5.
// Pinned where it stands (volatile): these ride in MFMA streams a few instructions at a time.
template <int E>
MDEV void split_put(unsigned (&H)[4], unsigned (&L)[4], const float cs, const float v) {
    constexpr int d = E >> 1;
    if constexpr ((E & 1) == 0) {
        asm volatile("v_fma_mixlo_f16 %0, %2, %3, 0\n\tv_fma_mixlo_f16 %1, %2, %3, -%0 op_sel_hi:[0,0,1]" : "+v"(H[d]), "+v"(L[d]) : "v"(cs), "v"(v));
    } else {
        unsigned tmp;   // the high half written by v_fma_mixhi is not read back in the next instruction: its hi goes through a temporary
        asm volatile("v_fma_mixlo_f16 %2, %3, %4, 0\n\tv_fma_mixhi_f16 %0, %3, %4, 0\n\tv_fma_mixhi_f16 %1, %3, %4, -%2 op_sel_hi:[0,0,1]"
                     : "+v"(H[d]), "+v"(L[d]), "=&v"(tmp)
                     : "v"(cs), "v"(v));
    }
}

template <int ACT>
MDEV float act_fn(float v) {
    if (ACT == ACT_GELU_FAST || ACT == ACT_GELU) { GeluStage g; gelu_stage0(g, v); gelu_stage1(g); return gelu_stage2(g); }
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_SILU) return v / (1.0f + expf(-v));
    return v;
}

// After the four swaps a lane (i, g) holds rows 8 g .. 8 g + 7 in r0..r7 and rows 16 + 8 g .. + 7 in r8..r15 of its column
// (v_permlane32_swap exchanges vdst[32..63] with src[0..31]; validated on the part through gemm4's ACE_G4_REGEPI build).
MDEV void rows_to_kgroups(f32x16& v) {
#pragma unroll
    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * hq + e]), __float_as_uint(v[8 * hq + 4 + e]), false, false);
            v[8 * hq + e] = __uint_as_float(sw[0]);
            v[8 * hq + 4 + e] = __uint_as_float(sw[1]);
        }
}

#ifdef ACE_X_TRACE   // measurement build only (tools/trace_wl.py): s_memtime stamps of wave 0 of one workgroup
__device__ unsigned long long mlp_trace[512];
#define MT(ev) do { if (blockIdx.x == ACE_X_TRACE && threadIdx.x == 0 && (ev) < 512) mlp_trace[ev] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MT(ev) do { } while (0)
#endif

template <int I0, int I1, class F>
MDEV void static_for(F&& f);

// A-fragment reads with a hand-managed pipeline.  hipcc, left alone at 480 registers per lane, reuses ONE fragment register
// set and waits lgkmcnt(0) before every MFMA (r02 profile: half of the kernel's time in s_waitcnt); as inline asm the reads
// are invisible to its bookkeeping, so they are issued DEPTH k-steps ahead and retired by a counted wait that names the
// destination registers (the data dependence keeps the consuming MFMAs below it).  LDS operations retire in order, so
// "lgkmcnt <= N" means all but the newest N have landed, whatever scalar loads are in flight.
struct Frag { half8 h, l; };
template <int OFF>
MDEV void frag_issue(Frag& f, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(f.h), "=&v"(f.l)
                 : "v"(lds_addr), "n"(OFF), "n"(OFF + 1024));
}
template <int N>
MDEV void frag_wait(Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.h), "+v"(f.l) : "n"(N));
}
// NS k-steps of one group: body(step, fragment) with the fragments of the steps s + 1 .. s + DEPTH already requested
template <int NS, int DEPTH, class F>
MDEV void pipelined_steps(unsigned lds_addr, F&& body) {
    Frag fr[DEPTH + 1];
    static_for<0, (DEPTH < NS ? DEPTH : NS)>([&](auto ss) {
        constexpr int s0 = decltype(ss)::value;
        frag_issue<s0 * 2048>(fr[s0 % (DEPTH + 1)], lds_addr);
    });
    static_for<0, NS>([&](auto ss) {
        constexpr int st = decltype(ss)::value;
        if constexpr (st + DEPTH < NS) frag_issue<(st + DEPTH) * 2048>(fr[(st + DEPTH) % (DEPTH + 1)], lds_addr);
        constexpr int newer = (NS - 1 - st) < DEPTH ? (NS - 1 - st) : DEPTH;   // requested after step st's fragments
        frag_wait<2 * newer>(fr[st % (DEPTH + 1)]);
        body(ss, fr[st % (DEPTH + 1)]);
    });
}

// The same in UNITS of two k-steps, one unit of read-ahead: body(unit, fragment of step 2u, fragment of step 2u + 1).  Two
// fragments per unit give the MFMA stream two independent accumulators to alternate between: a v_mfma_f32_32x32x16_f16 that
// accumulates into the result of the MFMA right before it waits for it (64 instead of 32 cycles - r02 in-kernel timeline:
// three dependent MFMAs per tile made the fc2 phases 1.7x their issue time).
template <int NS, class F>
MDEV void pipelined_pairs(unsigned lds_addr, F&& body) {
    static_assert(NS % 2 == 0, "units of two k-steps");
    constexpr int NU = NS / 2;
    Frag fr[2][2];
#if ACE_MLP_ABL & 2
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { asm volatile("" : "=v"(fr[a][b].h), "=v"(fr[a][b].l)); }
    static_for<0, NU>([&](auto uu) { constexpr int u = decltype(uu)::value; body(uu, fr[u % 2][0], fr[u % 2][1]); });
    return;
#endif
    frag_issue<0>(fr[0][0], lds_addr);
    frag_issue<2048>(fr[0][1], lds_addr);
    static_for<0, NU>([&](auto uu) {
        constexpr int u = decltype(uu)::value;
        if constexpr (u + 1 < NU) {
            frag_issue<(2 * u + 2) * 2048>(fr[(u + 1) % 2][0], lds_addr);
            frag_issue<(2 * u + 3) * 2048>(fr[(u + 1) % 2][1], lds_addr);
        }
        constexpr int newer = u + 1 < NU ? 4 : 0;
        asm volatile("s_waitcnt lgkmcnt(%4)"
                     : "+v"(fr[u % 2][0].h), "+v"(fr[u % 2][0].l), "+v"(fr[u % 2][1].h), "+v"(fr[u % 2][1].l)
                     : "n"(newer));
        body(uu, fr[u % 2][0], fr[u % 2][1]);
    });
}

template <int I0, int I1, class F>
MDEV void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}


}  // namespace
}  // namespace ace
