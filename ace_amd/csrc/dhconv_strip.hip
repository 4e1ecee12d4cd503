// Spectral filter contraction (fme/ace/models/modulus/contractions.py:183-195, "dhconv") for gfx950 with the weights
// streamed ONCE, straight from HBM into MFMA B fragments:
//     E_l[(m, b)][o] = sum_i D_l[(m, b)][i] W_l[i][o]      complex, one (C x C) filter matrix per degree l, rows m <= l
// The filter is 212 MB at the ACE2 shape and every element is used by at most 181 rows: the stage is bound by streaming
// the weights.  The 128 x 128 tile engine moves each weight through L2 -> LDS once per 128-row tile and each coefficient
// once per 128-column tile (1.1 GB of LDS-DMA traffic per launch, 465 MB of HBM reads for 262 MB of operands: r02 PMC).
// Here a workgroup owns (degree l, 128 output channels, a chunk of up to 96 rows): each of its four waves holds 32 output channels,
// real AND imaginary part, for the chunk's rows (1 .. 3 strips of 32, 6 accumulator tiles per wave).  Round 5 gave a workgroup up
// to 192 rows: 552 workgroups of very unequal size on 256 CUs, the heavy ones matrix-bound at ~40 us, the light ones latency-bound,
// 98 us per launch against a 50 us stream of the operands.  The chunks of one (degree, column group) are NEIGHBOURS in launch order
// on one XCD: the second reads the filter slice the first is streaming from that XCD's L2.
//   * the weights are the MFMA B operand: their stored form (k-packed "P format", compact complex: Wr | Wi per l) IS the
//     fragment of a lane, so they go global -> registers with coalesced 16-byte loads, two stages ahead, and never touch LDS;
//     a stage is 32 rows of Wr and the same 32 rows of Wi, each used for two products (with D_re and with D_im), so every
//     filter element is fetched exactly once per launch;
//   * the coefficients D_l (fp16 hi/lo planes written by the Legendre stage, row-major (D_re | D_im)) are the A operand
//     shared by the four waves: a stage holds the 32-wide k slice of BOTH halves of the row, as 1-KiB LDS-DMA pieces of
//     16 rows x 32 k with a source-side XOR swizzle, three-stage ring, the pieces of stage t + 2 requested while stage t computes (as
//     the filter fragments; one or two requests behind every six MFMAs), ONE barrier per stage, counted waits (all vector-memory
//     operations of the loop are inline asm, no stores: the count is exact);
//   * TWO workgroups per CU (round 6, second half): with requests two stages ahead instead of three a wave needs 256 registers and a
//     workgroup 72 KiB of LDS, so a CU holds two units: one's prologue (7 - 13 k cycles of memory latency) and epilogue (6 - 11 k)
//     run under the other's MFMAs.  In-kernel trace: a pair of 96-row units takes 88 k cycles where one alone took 55 k, the
//     stages of a pair issue an MFMA every 37 cycles (87 % of the pipe); the launch 93.6 -> 87 us same box with the XCD-balanced
//     list (dhconv_units.h, order 1).  While all CUs run pairs of 96-row units the filter stream alone asks HBM for ~6 TB/s:
//     what is left is the ramp at both ends of a unit and the second, half-empty round of the dispatch
//     (profiles/r06_dhconv_occupancy.txt: host-packed persistent lists, longest-first unit orders and one-strip units at three
//     workgroups per CU were measured and not kept);
//   * complex structure: real += D_re Wr - D_im Wi, imaginary += D_re Wi + D_im Wr; the minus sign is put on the D_im
//     fragment (8 v_xor per strip and k16 step);
//   * row strips beyond l are skipped (1 .. 3 active strips of 32 rows per chunk).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dhconv_units.h"
#include "strip_common.h"

namespace ace {
namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#ifdef ACE_DH_TRACE   // measurement builds (tools/trace_dh.py): per workgroup, wave 0: start, after the main loop, end (s_memtime), degree, XCC
__device__ unsigned long long dh_wg_span[8192][6];
#define DH_STAMP(k, v) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) dh_wg_span[blockIdx.x][k] = (v); } while (0)
__device__ unsigned long long dh_stage_trace[4][64];   // per traced workgroup (ACE_DH_TRACE_WG + 8 k, k < 4): stamps of wave 0
#ifndef ACE_DH_TRACE_WG
#define ACE_DH_TRACE_WG 0
#endif
#define DH_MT(ev) do { const int tw_ = ((int)blockIdx.x - ACE_DH_TRACE_WG) / 264; if (threadIdx.x == 0 && ((int)blockIdx.x - ACE_DH_TRACE_WG) % 264 == 0 && tw_ >= 0 && tw_ < 4 && (ev) < 64) dh_stage_trace[tw_][ev] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DH_STAMP(k, v) do { } while (0)
#define DH_MT(ev) do { } while (0)
#endif


MDEV void gload16(half8& dst, const _Float16* p) {   // un-waited 16-byte global load (retired by wait_b below)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
MDEV half8 neg8(half8 v) {
    u32x4v u = __builtin_bit_cast(u32x4v, v);
    u ^= 0x80008000u;
    return __builtin_bit_cast(half8, u);
}


// NS: active 32-row strips (1 .. 3) of this workgroup's row chunk [row0, row0 + rows) of degree l
#ifndef ACE_DH_PB
#define ACE_DH_PB 2
#endif
#ifndef ACE_DH_WGS
#define ACE_DH_WGS 2
#endif
template <int NS>
MDEV void dhconv_body(const DhconvStripArgs& p, char* smem, const int l, const int j, const int row0, const int rows, const int tid) {
    constexpr int PB = ACE_DH_PB;                 // A pieces and B fragments of stage t + PB are issued while stage t computes
    constexpr int NSTG = PB + 1;                  // ring depth (stages): the slot refilled during stage t held stage t - 1, which every
                                                  // wave had finished before any wave passed the barrier of stage t
    constexpr int STAGE = NS * 8192;              // bytes: NS strips x 2 slices (re, im) x (2 row pieces x 2 planes) x 1 KiB
    constexpr int NA = 2 * NS;                    // A pieces per wave per stage
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int C = p.C, K2 = 2 * C;
    // Block-diagonal filter (G groups of C / G channels: output channel o sees input channels of its own group only, the dense
    // form holds exact zeros elsewhere): this workgroup's 128 output channels need the input range [k_lo, k_lo + kspan) only -
    // its group (C / G a multiple of 128) or the 128 / (C / G) whole groups under its outputs.  Skipping exact zeros changes no bit.
    const int cg = C / p.groups;
    const bool skip = p.groups > 1 && (cg % 128 == 0 || 128 % cg == 0);
    const int kspan = skip ? (cg > 128 ? cg : 128) : C;
    const int tb = skip ? ((128 * j) / kspan) * (kspan / 32) : 0;   // first stage
    const int nstages = kspan / 32;               // stage t = k in [32 (tb + t), + 32) of BOTH halves of the row (D_re | D_im)
    const int oc = 128 * j + 32 * wave;           // this wave's 32 output channels (real and imaginary part)
    // Diagonal blocks only (kstore = C / G rows per column, the reference's grouped parameter as it is): row kk of column o is
    // input channel (o / cg) cg + kk.  The stages outside this WAVE's group (cg < 128: the workgroup's 128 columns span 128 / cg
    // groups) multiply exact zeros in the dense form: their fragments are fetched from a clamped address (uniform operation
    // counts for the counted waits) and their MFMAs skipped (wave-uniform) - no bit changes, half / a quarter of the matrix work.
    const int kstore = p.kstore > 0 ? p.kstore : C;
    const bool native = kstore != C;
    const int kbase = native ? (oc / cg) * cg : 0;

    const unsigned raw_a = slot_load(p.amax + lane);

    // ---- A pieces: piece q of a stage = (strip s, slice sl, row half rh, plane pl) = 8 s + 4 sl + 2 rh + pl; this wave issues
    //      (rh, pl) = (wave / 2, wave % 2) of every (s, sl).  Lane L fetches the XOR-swizzled 16-byte slot of row L / 4
    const _Float16* Dh = p.Dhi + (long)l * p.sD + (long)row0 * K2;
    const _Float16* Dl = p.Dlo + (long)l * p.sD + (long)row0 * K2;
    const _Float16* asrc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int s = k >> 1, sl = k & 1, rh = wave >> 1, pl = wave & 1;
        int row = 32 * s + 16 * rh + (lane >> 2);
        const int ls = (lane & 3) ^ ((row >> 2) & 3);
        row = row < rows ? row : rows - 1;
        asrc[k] = (pl ? Dl : Dh) + (long)row * K2 + sl * C + 8 * ls;
    }
    auto issue_a1 = [&](int t, int k) {   // piece k of stage t -> ring slot t % NSTG; stages past the end re-fetch the last (uniform count)
        const int tt = tb + (t < nstages ? t : nstages - 1);
        glds16(asrc[k] + 32 * tt, smem + (t % NSTG) * STAGE + (wave + 4 * k) * 1024);
    };
    auto issue_a = [&](int t) {
#pragma unroll
        for (int k = 0; k < NA; ++k) issue_a1(t, k);
    };
    // ---- B fragments of stage t: rows [32 t, 32 t + 32) of Wr and of Wi, this wave's 32 columns.  Each filter element is
    //      loaded by exactly one wave of one workgroup, once (the first version walked k over (D_re | D_im) with the real
    //      columns in waves 0 - 1 and the imaginary ones in 2 - 3: Wr and Wi were each fetched twice, half a kernel apart -
    //      567 MB of HBM reads for 262 MB of operands, r02 PMC)
    const _Float16* Wh = p.Whi + (long)l * p.sW;
    const _Float16* Wl = p.Wlo + (long)l * p.sW;
    struct BSet { half8 h[2][2], l[2][2]; };      // [k16 step][Wr | Wi]
    auto issue_b1 = [&](BSet& b, int t, auto qc) {   // load q of the eight of a set: (k16 step c, Wr | Wi, hi | lo)
        constexpr int q = decltype(qc)::value, c = q >> 2, blk = (q >> 1) & 1, lo = q & 1;
        const int tt = tb + (t < nstages ? t : nstages - 1);
        int kg = 4 * tt + 2 * c + g - kbase / 8;          // k-group relative to this wave's first stored row
        if (native) kg = kg < 0 ? 0 : (kg >= kstore / 8 ? kstore / 8 - 1 : kg);
        const long off = (long)blk * kstore * C + ((long)kg * C + oc + i) * 8;
        if constexpr (lo) gload16(b.l[c][blk], Wl + off);
        else gload16(b.h[c][blk], Wh + off);
    };
    auto issue_b = [&](BSet& b, int t) {
        static_for<0, 8>([&](auto qc) { issue_b1(b, t, qc); });
    };
    auto wait_b = [&](BSet& b, auto nn) {         // retire this set: at most nn newer operations stay in flight
        constexpr int N = decltype(nn)::value;
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(b.h[0][0]), "+v"(b.h[0][1]), "+v"(b.h[1][0]), "+v"(b.h[1][1]), "+v"(b.l[0][0]), "+v"(b.l[0][1]),
                       "+v"(b.l[1][0]), "+v"(b.l[1][1])
                     : "n"(N)
                     : "memory");
    };

    f32x16 acc[NS][2];                            // [strip][real | imaginary part of the output]
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][tl][r] = 0.f;

    // prologue in the steady-state issue order: A0 B0 | A1 B1 | A2 B2
    BSet bs[4];                                   // ring of fragment sets, indexed with compile-time constants only
    issue_a(0); issue_b(bs[0], 0);
    if constexpr (PB >= 2) { issue_a(1); issue_b(bs[1], 1); }
    if constexpr (PB == 3) { issue_a(2); issue_b(bs[2], 2); }

    const int key = (i >> 2) & 3;                 // swizzle key of this lane's fragment rows (rows i and i + 16 ... share it mod 4)

    // The operands of stage t + PB are requested WHILE stage t computes, one or two operations behind each group of six MFMAs
    // (in front of the stage they cost a wave 800 - 1800 cycles of issue with the matrix pipe idle: tools/trace_dh.py).  Issue
    // order: ... A(t+1) B(t+1) | A(t+2) B(t+2) | A(t+3) B(t+3): at the top of stage t exactly 2 (NA + 8) operations are newer
    // than B(t).  All of them are loads (in-order retirement), the loop has no stores: vmcnt(2 (NA + 8)) is exact.  B(t + 3)
    // goes into the set of stage t - 1, A(t + 3) into its ring slot.
    constexpr int NOPS = NA + 8, NSLOT = 4 * NS;
    auto stage = [&](const int t, BSet& b, BSet& bnew) {
        DH_MT(4 * t);
        DH_MT(4 * t + 1);
        wait_b(b, std::integral_constant<int, (PB - 1) * (NA + 8)>{});
        DH_MT(4 * t + 2);
        __builtin_amdgcn_s_barrier();             // every wave's pieces of stage t landed; every wave is done with stage t - 1
        DH_MT(4 * t + 3);
        auto issue_slot = [&](auto sc) {          // the operations of slot sc of the NSLOT MFMA groups of this stage
            constexpr int sl_ = decltype(sc)::value;
            static_for<(sl_ * NOPS) / NSLOT, ((sl_ + 1) * NOPS) / NSLOT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < NA) issue_a1(t + PB, k);
                else issue_b1(bnew, t + PB, std::integral_constant<int, k - NA>{});
            });
        };
        const unsigned sl = (unsigned)(size_t)(lds_cptr)(smem + (t % NSTG) * STAGE);
        const int k0 = 32 * (tb + t);
        const bool live = !native || (k0 >= kbase && k0 < kbase + cg);   // wave-uniform: this stage's k range lies in this wave's group
        // A fragment of (strip s, slice sl), k16 step c: rows 32 s + i (row piece i >> 4), logical slot 2 c + g, physical ^ key
        const unsigned la = sl + ((i >> 4) * 2) * 1024 + (i & 15) * 64;
        // four phases (c, slice) = (0, re) (0, im) (1, re) (1, im); the fragments of phase ph + 1 are requested strip by strip
        // while phase ph computes: 2 (NS - 1) LDS reads are in flight behind the fragment being waited for, always
        Frag fa[NS], fb[NS];
        auto frag_addr = [&](int ph, int s) { return la + (2 * s + (ph & 1)) * 4096 + (unsigned)(((2 * (ph >> 1) + g) ^ key) * 16); };
        auto issue_f = [&](Frag& f, unsigned addr) {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(f.h), "=&v"(f.l) : "v"(addr));
        };
#pragma unroll
        for (int s = 0; s < NS; ++s) issue_f(fa[s], frag_addr(0, s));
        static_for<0, 4>([&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            constexpr int c = ph >> 1;
            constexpr bool im = ph & 1;
            static_for<0, NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                Frag& f = (ph & 1) ? fb[s] : fa[s];
                constexpr int newer = ph < 3 ? 2 * (NS - 1) : 2 * (NS - 1 - s);
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.h), "+v"(f.l) : "n"(newer));
                if (!live) {
                } else if constexpr (!im) {   // D_re: real += D_re Wr, imaginary += D_re Wi
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l, b.h[c][0], acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l, b.h[c][1], acc[s][1], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, b.l[c][0], acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, b.l[c][1], acc[s][1], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, b.h[c][0], acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, b.h[c][1], acc[s][1], 0, 0, 0);
                } else {               // D_im: imaginary += D_im Wr, real -= D_im Wi (sign on the coefficient fragment)
                    const half8 nh = neg8(f.h), nl = neg8(f.l);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l, b.h[c][0], acc[s][1], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(nl, b.h[c][1], acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, b.l[c][0], acc[s][1], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(nh, b.l[c][1], acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, b.h[c][0], acc[s][1], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(nh, b.h[c][1], acc[s][0], 0, 0, 0);
                }
                if constexpr (ph < 3) issue_f((ph & 1) ? fa[s] : fb[s], frag_addr(ph + 1, s));
                __builtin_amdgcn_sched_barrier(0);
                issue_slot(std::integral_constant<int, ph * NS + s>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    for (int t0 = 0; t0 < nstages; t0 += 4)       // nstages (C / 32, or the group span / 32) is a multiple of 4
        static_for<0, 4>([&](auto u) { stage(t0 + u, bs[u], bs[(u + PB) % 4]); });
    // The dummy tail fetches are still in flight into fragment sets that are dead as far as hipcc knows: the wait that retires
    // them NAMES their registers, or the allocator may hand those registers to the epilogue's values above the wait and the
    // returning loads would land on them (tools/store_hazard.hip, variant 3: the mechanism behind round 2's "store data"
    // corruption).  Two statements: an asm takes at most 30 operands.
    DH_STAMP(1, __builtin_amdgcn_s_memtime());
    DH_MT(48);
    auto name_set = [](BSet& b) {
        asm volatile("" : "+v"(b.h[0][0]), "+v"(b.h[0][1]), "+v"(b.h[1][0]), "+v"(b.h[1][1]), "+v"(b.l[0][0]), "+v"(b.l[0][1]),
                          "+v"(b.l[1][0]), "+v"(b.l[1][1]));
    };
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(bs[0].h[0][0]), "+v"(bs[0].h[0][1]), "+v"(bs[0].h[1][0]), "+v"(bs[0].h[1][1]), "+v"(bs[0].l[0][0]),
                   "+v"(bs[0].l[0][1]), "+v"(bs[0].l[1][0]), "+v"(bs[0].l[1][1])
                 :
                 : "memory");
    name_set(bs[1]); name_set(bs[2]); name_set(bs[3]);

    // ---- epilogue: lane = column, registers = rows; fp32 E[l][row][part * C + column], 128-byte row segments.
    // Order matters on gfx950: the range maximum (shuffles and an LDS round trip, i.e. loads that LAND in registers) comes
    // first, the stores last and from the accumulator registers themselves, which nothing writes afterwards.  With the stores
    // first and their data in temporaries, 7 of 1000 forwards of the 1-degree network came out wrong (r02): a load issued
    // later returned into a register whose store had not read it yet.
    // (the range slot is USED only here: consumed in the prologue, hipcc's wait for it - a vmcnt(0), the asm loads are invisible to its
    // bookkeeping - sat behind the three stages of operand requests and cost every workgroup 8 - 16 k cycles before its first MFMA)
    const float inv_a = ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_a)));
    const float oscale = inv_a / p.bscale;
    float* E = p.E + (long)l * p.sE + (long)row0 * K2;
    float vmax = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * s + (r & 3) + 8 * (r >> 2) + 4 * g;
                acc[s][tl][r] *= oscale;
                vmax = fmaxf(vmax, row < rows ? fabsf(acc[s][tl][r]) : 0.f);
            }
    if (p.omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int col = tl * C + oc + i;   // tl: real | imaginary part
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * s + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (row < rows) E[(long)row * K2 + col] = acc[s][tl][r];
            }
        }
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" ::"v"(acc[s][0]), "v"(acc[s][1]));   // the store data stays put to the end
}

__global__ __launch_bounds__(256, ACE_DH_WGS) void dhconv_strip_kernel(DhconvStripArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[(ACE_DH_PB + 1) * DH_CHUNK_STRIPS * 8192];
    // Workgroup b runs on XCD b % 8 (each XCD has its own L2) and, measured (tools/trace_dh.py, profiles/r06_dhconv_dispatch.txt), on
    // shader engine (b / 8) % 4 of that XCD: the dispatcher deals a launch's workgroups to the four engines in turn WHATEVER their
    // load, only the 8 CUs of an engine (16 resident workgroups) share work dynamically.  dhconv_build_units (host) lists, per XCD,
    // the units (degree, column group, row chunk) of the degrees it owns, longest group first, the chunks of one (degree, column
    // group) next to each other (they stream the same 393 KB filter slice: once from HBM, then from this XCD's L2), in the chunk
    // order that keeps the four engines even.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int4 u = reinterpret_cast<const int4*>(p.units)[xcd * p.units_per_xcd + idx];   // (l, j, row0, rows); rows = 0: padding
    const int l = u.x, j = u.y, row0 = u.z, rows = u.w;
    if (rows <= 0) return;
#ifdef ACE_DH_TRACE
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    // active 32-row strips: 1 .. 3
    if (rows <= 32) dhconv_body<1>(p, smem, l, j, row0, rows, tid);
    else if constexpr (DH_CHUNK_STRIPS < 2) return;
    else if (rows <= 64) dhconv_body<2>(p, smem, l, j, row0, rows, tid);
    else if constexpr (DH_CHUNK_STRIPS >= 3) dhconv_body<3>(p, smem, l, j, row0, rows, tid);
#ifdef ACE_DH_TRACE
    DH_MT(49);
    DH_STAMP(0, t_start);
    DH_STAMP(2, __builtin_amdgcn_s_memtime());
    DH_STAMP(3, (unsigned long long)rows);
    { unsigned xcc, hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); DH_STAMP(4, xcc & 0xf); DH_STAMP(5, hwid); }
#endif
}

}  // namespace

#ifdef ACE_DH_TRACE
extern "C" int ace_debug_dh_stages(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dh_stage_trace), sizeof(dh_stage_trace)); }
extern "C" int ace_debug_dh_spans(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dh_wg_span), sizeof(dh_wg_span)); }
#endif

bool dhconv_native_groups_ok(int C, int groups) {
    if (groups <= 1 || C % groups != 0 || C % 128 != 0) return false;
    const int cg = C / groups;
    return cg % 32 == 0 && (cg % 128 == 0 || 128 % cg == 0);
}

#ifndef ACE_DH_ORDER
#define ACE_DH_ORDER DH_ORDER_DEFAULT
#endif
int dhconv_build_units(int L, int Mrows, int trimul, int C, std::vector<int>& out) {
    int order = ACE_DH_ORDER;
#ifdef ACE_MEASUREMENT_SWITCHES
    if (const char* e = std::getenv("ACE_DH_ORDER")) order = std::atoi(e);
#endif
    return dhconv_units(L, Mrows, trimul, C, out, order);
}

bool dhconv_strip_eligible(const DhconvStripArgs& a) {
    if (a.kstore > 0 && a.kstore != a.C && !(a.groups > 1 && a.kstore == a.C / a.groups && dhconv_native_groups_ok(a.C, a.groups))) return false;
    return a.C % 128 == 0 && a.C >= 128 && a.groups >= 1 && a.C % a.groups == 0 && a.Mrows >= 1 && (a.Mrows + 191) / 192 <= 65535 && a.L >= 1 && a.Dhi && a.Dlo && a.Whi && a.Wlo && a.E && a.amax;
}

hipError_t launch_dhconv_strip(const DhconvStripArgs& a, hipStream_t s) {
    if (!dhconv_strip_eligible(a) || !a.units || a.units_per_xcd < 1 || (long)a.units_per_xcd * 8 > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)(8 * a.units_per_xcd)), block(256);
    hipLaunchKernelGGL(dhconv_strip_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace ace
