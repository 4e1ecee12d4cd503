// Fused MLP for gfx950: h' = W2 . GELU(W1f . P(T) + b1f) + b2 + (a0 h + b0)   (fme/ace/models/modulus/layers.py:117-137 with the
// block's outer skip, sfnonet.py:234-250), one launch, the hidden activation U never leaves the chip.
//
// A wave owns a 32-pixel strip for the whole MLP:
//   * the strip of the input (P-format fp16 hi/lo planes written by the inner-skip GEMM) is loaded once as MFMA B fragments
//     and stays in registers (C = 384: 24 k-steps x 8 VGPRs);
//   * the hidden dimension is walked in chunks of 32 units: fc1 of the chunk (24 k-steps x 3 MFMAs into one 32 x 32 tile),
//     bias + GELU + hi/lo split IN REGISTERS - after four v_permlane32_swap the accumulator tile IS the B operand of two
//     k-steps of fc2 - then fc2 of the chunk (12 output tiles x 2 k-steps x 3 MFMAs into the 12 resident accumulators);
//   * both weight matrices stream through LDS as pre-packed A fragments (strip_pack.h layout), one 48 KiB slot each,
//     shared by the four waves of the workgroup: the slot of W1 is refilled while fc2 runs and vice versa;
//   * the epilogue (bias, residual with its own affine, fp32 store, P-format planes of h', row statistics of the next
//     instance norm) runs from the 12 accumulators.
// One wave per SIMD (the strip + the 12 accumulators need ~430 registers of the unified 512-entry file).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "strip_common.h"

namespace ace {
namespace {

// NC = C / 32: KS1 = 2 NC k16-steps of fc1, NT2 = NC 32-row output tiles of fc2.  The two weight matrices are one stream
// of GROUPS of NC k-step blocks (NC * 2 KiB): per hidden chunk c the groups 4c, 4c+1 are the two k-halves of the W1 chunk
// (32 hidden rows x C) and 4c+2, 4c+3 the two row-halves of the W2 chunk (C rows x 32 hidden).  A ring of six groups in
// LDS keeps five groups (> 1 chunk, ~3 us of MFMA work) in flight: an LDS-DMA piece issued under full-chip load lands
// 2 - 3 us later (r02 measurement: with one group of lookahead the kernel waited half of its time).
template <int NC, int ACT, bool PK>
__global__ __launch_bounds__(256, 1) void mlp_strip_kernel(MlpStripArgs p) {
    constexpr int KS1 = 2 * NC, NT2 = NC;
    constexpr int GRP = NC * 2048;          // bytes per group
    constexpr int NSLOT = 6, AHEAD = NSLOT - 1;
    constexpr int PW = NC / 2;              // 1-KiB pieces per wave per group
    constexpr int RING = NSLOT * GRP;
    constexpr int B1MAX = 4096;             // hidden units whose folded bias is kept in LDS
    constexpr int FDEPTH = ACE_MLP_FDEPTH;  // k-steps of fragment read-ahead
    __shared__ __attribute__((aligned(16))) char smem[RING + B1MAX * 4];
    float* b1s = reinterpret_cast<float*>(smem + RING);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int strips = (p.HW + 127) / 128;
    const int smp = blockIdx.x / strips;
    const int n0 = (blockIdx.x % strips) * 128 + wave * 32;
    const int n = n0 + i;
    const int nc = n < p.HW ? n : p.HW - 1;
    const int nchunks = p.hid / 32;
    const int ngroups = 4 * nchunks;

    MT(0);
    const unsigned raw_x = slot_load(p.xslot + lane);
    const unsigned raw_a = slot_load(p.a1slot + lane);
    const unsigned raw_c = slot_load(p.cinb + lane);
    const unsigned raw_r = PK ? slot_load(p.rmax + lane) : 0u;

    const _Float16* A1 = p.A1 + (long)smp * p.sA1;
    const _Float16* A2 = p.A2;
    auto issue = [&](int q) {   // group q -> ring slot q % NSLOT; groups past the end re-fetch the last one (keeps the count uniform)
        const int qq = q < ngroups ? q : ngroups - 1;
        const int c = qq >> 2, ph = qq & 3;
        const _Float16* src = (ph < 2 ? A1 + (long)c * (KS1 * 1024) + (long)ph * (NC * 1024)
                                      : A2 + (long)c * (NT2 * 2 * 1024) + (long)(ph - 2) * (NC * 1024)) + lane * 8;
        const char* dst = smem + (q % NSLOT) * GRP;
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int pc = wave + 4 * k;
            glds16(src + pc * 512, dst + pc * 1024);
        }
    };
    issue(0);
    issue(1);

    // ---- resident input strip (already split and k-packed by its producer)
    half8 xh[KS1], xl[KS1];
    {
        const _Float16* Xh = p.Xhi + (long)smp * p.sX;
        const _Float16* Xl = p.Xlo + (long)smp * p.sX;
#pragma unroll
        for (int j = 0; j < KS1; ++j) {
            const long off = ((long)(2 * j + g) * p.ldn + nc) * 8;
            xh[j] = *reinterpret_cast<const half8*>(Xh + off);
            xl[j] = *reinterpret_cast<const half8*>(Xl + off);
        }
    }
    {   // folded fc1 bias of this sample -> LDS (read back per chunk with ds_read: no vector-memory load inside the loop)
        const float* b1 = p.b1 + (long)smp * p.sb1;
        float bv[B1MAX / 256];
#pragma unroll
        for (int k = 0; k < B1MAX / 256; ++k) bv[k] = b1[(tid + 256 * k) < p.hid ? tid + 256 * k : 0];   // one batch of loads
#pragma unroll
        for (int k = 0; k < B1MAX / 256; ++k)
            if (tid + 256 * k < p.hid) b1s[tid + 256 * k] = bv[k];
    }
    const float xbound = wave_max_bits(raw_x);
    const float inv_x = ldexpf(1.0f, -pow2_exponent_for(xbound));
    const float inv_a1 = ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_a)));
    const float s_fc1 = inv_x * inv_a1;
    // bound of the hidden activation, identical in every workgroup: |U| <= cw1 * bound(norm1(T)) + cb1
    const float ubound = fmaf(p.cw1, wave_max_bits(raw_c), p.cb1);
    const int eu = pow2_exponent_for(ubound);
    const float uscale = ldexpf(1.0f, eu);
    const float s_fc2 = ldexpf(1.0f, -eu) / p.a2scale;

    f32x16 out[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[t][r] = 0.f;

    // Everything loaded so far is consumed HERE: hipcc places its own (counted) waits for a plain load at its first use
    // and knows nothing of the LDS-DMA pieces, so a first use inside the loop would drain the ring every iteration.
    MT(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MT(2);
#pragma unroll
    for (int j = 0; j < KS1; ++j) asm volatile("" : "+v"(xh[j]), "+v"(xl[j]));
    __syncthreads();                       // bias table visible; groups 0 and 1 landed in every wave's share
    MT(3);
#pragma unroll
    for (int q = 2; q < AHEAD; ++q) issue(q);

    // LDS-DMA bookkeeping: the only vector-memory operations inside the loop are the DMA pieces (no plain loads, no
    // stores), so the counted wait is exact.  At the top of group q the queue holds the groups q .. q + AHEAD - 1 that
    // have not landed; vmcnt((AHEAD - 1) * PW) retires exactly group q.
    auto top = [&](int q) {
        MT(8 + 4 * q);
        if (!(ACE_MLP_ABL & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PW) : "memory");
        MT(9 + 4 * q);
        if (!(ACE_MLP_ABL & 8)) __builtin_amdgcn_s_barrier();      // group q landed in every wave's share; every wave is done with group q - 1
        MT(10 + 4 * q);
    };
    // ... whose slot is refilled DURING the steps of group q: one 1-KiB piece every NC / PW steps, between the MFMA triples
    // (an LDS-DMA issue costs ~90 cycles of the wave's issue slot; six of them at the top of a group stalled the matrix pipe
    // for 500 cycles per group, 19 % of the loop - r02 in-kernel timeline)
    auto piece_src = [&](int q) -> const _Float16* {
        const int qq = q < ngroups ? q : ngroups - 1;
        const int c = qq >> 2, ph = qq & 3;
        return (ph < 2 ? A1 + (long)c * (KS1 * 1024) + (long)ph * (NC * 1024)
                       : A2 + (long)c * (NT2 * 2 * 1024) + (long)(ph - 2) * (NC * 1024)) + lane * 8 + wave * 512;
    };
    for (int c = 0; c < nchunks; ++c) {
        f32x16 u0, u1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { u0[r] = 0.f; u1[r] = 0.f; }
        // ---- fc1 of chunk c: two k-halves
        static_for<0, 2>([&](auto phh) {
            constexpr int ph = decltype(phh)::value;
            const int q = 4 * c + ph;
            top(q);
            const unsigned sl = (unsigned)(size_t)(lds_cptr)(smem + (q % NSLOT) * GRP) + lane * 16;
            const _Float16* nsrc = piece_src(q + AHEAD);
            const char* ndst = smem + ((q + AHEAD) % NSLOT) * GRP + wave * 1024;
            pipelined_pairs<NC>(sl, [&](auto uu, const Frag& f0, const Frag& f1) {
                constexpr int un = decltype(uu)::value;
                constexpr int j0 = ph * NC + 2 * un, j1 = j0 + 1;
                // six MFMAs, the two accumulators strictly alternating (all six products are summed in the end)
                u0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.l, xh[j0], u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.h, xl[j0], u1, 0, 0, 0);
                u0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.h, xh[j0], u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.l, xh[j1], u1, 0, 0, 0);
                u0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.h, xl[j1], u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.h, xh[j1], u1, 0, 0, 0);
                if (!(ACE_MLP_ABL & 1)) glds16(nsrc + un * 4 * 512, ndst + un * 4 * 1024);
            });
        });
        f32x16 u = u0 + u1;
        // bias of this lane's rows after the swap: rows 8 g + e (e < 8) and 16 + 8 g + e
        const f32x4 bA = *reinterpret_cast<const f32x4*>(b1s + 32 * c + 8 * g);
        const f32x4 bB = *reinterpret_cast<const f32x4*>(b1s + 32 * c + 8 * g + 4);
        const f32x4 bC = *reinterpret_cast<const f32x4*>(b1s + 32 * c + 16 + 8 * g);
        const f32x4 bD = *reinterpret_cast<const f32x4*>(b1s + 32 * c + 16 + 8 * g + 4);
        rows_to_kgroups(u);
        half8 uh[2], ul[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float bv = kk == 0 ? (e < 4 ? bA[e & 3] : bB[e & 3]) : (e < 4 ? bC[e & 3] : bD[e & 3]);
                const float y = ((ACE_MLP_ABL & 4) ? fmaf(u[8 * kk + e], s_fc1, bv) : act_fn<ACT>(fmaf(u[8 * kk + e], s_fc1, bv))) * uscale;
                const _Float16 h = (_Float16)y;
                uh[kk][e] = h;
                ul[kk][e] = (_Float16)(y - (float)h);
            }
        // ---- fc2 of chunk c: k-step 0 of all NT2 output tiles, then k-step 1 (the GELU of the second 16 hidden rows runs
        //      under the MFMAs of the first; three dependent MFMAs per accumulator and phase instead of six)
        static_for<0, 2>([&](auto phh) {
            constexpr int kk = decltype(phh)::value;
            const int q = 4 * c + 2 + kk;
            top(q);
            const unsigned sl = (unsigned)(size_t)(lds_cptr)(smem + (q % NSLOT) * GRP) + lane * 16;
            const _Float16* nsrc = piece_src(q + AHEAD);
            const char* ndst = smem + ((q + AHEAD) % NSLOT) * GRP + wave * 1024;
            pipelined_pairs<NC>(sl, [&](auto uu, const Frag& f0, const Frag& f1) {
                constexpr int un = decltype(uu)::value;
                constexpr int t0 = 2 * un, t1 = t0 + 1;
                out[t0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.l, uh[kk], out[t0], 0, 0, 0);
                out[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.l, uh[kk], out[t1], 0, 0, 0);
                out[t0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.h, ul[kk], out[t0], 0, 0, 0);
                out[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.h, ul[kk], out[t1], 0, 0, 0);
                out[t0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.h, uh[kk], out[t0], 0, 0, 0);
                out[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.h, uh[kk], out[t1], 0, 0, 0);
                if (!(ACE_MLP_ABL & 1)) glds16(nsrc + un * 4 * 512, ndst + un * 4 * 1024);
            });
        });
    }
    MT(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy refills of the tail
    MT(5);

    // ---- epilogue: rows of tile t are output channels 32 t + acc_row(r, g); lane column = pixel n
    float cscale = 1.f;
    if (PK) {
        const float resb = wave_max_bits(raw_r);
        const float cbound = fmaf(p.cw2, ubound, p.cb2) + resb;
        cscale = ldexpf(1.0f, pow2_exponent_for(cbound));
        if (tid == 0) atomicMax(p.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
    }
    const float* R = p.R + (long)smp * p.sR + nc;
    const float* rsc = p.rsc ? p.rsc + (long)smp * p.srs : nullptr;
    const float* rsh = p.rsc ? p.rsh + (long)smp * p.srs : nullptr;
    float* Cc = p.C + (long)smp * p.sC + nc;
    float vmax = 0.f;
    const bool nok = n < p.HW;
    const long HW = p.HW;
    // per-wave transpose buffer of the row statistics and the per-row epilogue parameters, in the ring: every wave's tail
    // refills have landed and every wave is out of the main loop once the barrier below has been passed
    __syncthreads();
    float* St = reinterpret_cast<float*>(smem) + wave * (33 * 32);
    float* Pb = reinterpret_cast<float*>(smem) + 4 * (33 * 32);      // [Cch] b2 + rsh
    float* Ps = Pb + NC * 32;                                         // [Cch] rsc
    for (int k = tid; k < NC * 32; k += 256) {
        Pb[k] = p.b2[k] + (rsh ? rsh[k] : 0.f);
        Ps[k] = rsc ? rsc[k] : 1.f;
    }
    __syncthreads();
    const int ncols_ok = p.HW - n0 < 32 ? (p.HW - n0 > 0 ? p.HW - n0 : 0) : 32;
    if (ncols_ok == 32) {
        // ---- whole strip (all but the last workgroup): no masks, buffer addressing (per-lane offset once, per-row
        //      offsets wave-uniform in soffset: no address arithmetic per access)
        const int plane_bytes = NC * 32 * p.HW * 4;
        const auto rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R + (long)smp * p.sR), 0, plane_bytes, 0x00020000);
        const auto rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + (long)smp * p.sC, 0, plane_bytes, 0x00020000);
        const auto rsH = __builtin_amdgcn_make_buffer_rsrc(PK ? p.Chi + (long)smp * p.sCp : nullptr, 0, plane_bytes, 0x00020000);
        const auto rsL = __builtin_amdgcn_make_buffer_rsrc(PK ? p.Clo + (long)smp * p.sCp : nullptr, 0, plane_bytes, 0x00020000);
        const int voff = (8 * g * p.HW + n) * 4;          // fp32 element (row 8 g, column n)
        const int voffp = (g * p.HW + n) * 16;            // P entry (k group g, column n)
        const int rowb = p.HW * 4;                        // bytes per channel row
        static_for<0, 2>([&](auto hh) {
            constexpr int th = decltype(hh)::value;
            float resv[NT2 / 2][16];
            static_for<0, NT2 / 2>([&](auto tt) {
                constexpr int tl = decltype(tt)::value;
                constexpr int t = th * (NT2 / 2) + tl;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    resv[tl][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsR, voff, (32 * t + 16 * (r >> 3) + (r & 7)) * rowb, 0));
            });
            static_for<0, NT2 / 2>([&](auto tt) {
                constexpr int tl = decltype(tt)::value;
                constexpr int t = th * (NT2 / 2) + tl;
                f32x16 v = out[t];
                rows_to_kgroups(v);
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const int row0 = 32 * t + 16 * hq + 8 * g;
                    const f32x4 ba = *reinterpret_cast<const f32x4*>(Pb + row0), bb = *reinterpret_cast<const f32x4*>(Pb + row0 + 4);
                    const f32x4 sa = *reinterpret_cast<const f32x4*>(Ps + row0), sb = *reinterpret_cast<const f32x4*>(Ps + row0 + 4);
                    float val[8];
                    half8 hh8, ll8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float bq = e < 4 ? ba[e & 3] : bb[e & 3];
                        const float sq_ = e < 4 ? sa[e & 3] : sb[e & 3];
                        val[e] = fmaf(resv[tl][8 * hq + e], sq_, fmaf(v[8 * hq + e], s_fc2, bq));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[e]), rsC, voff, (32 * t + 16 * hq + e) * rowb, 0);
                        vmax = fmaxf(vmax, fabsf(val[e]));
                        if (PK) {
                            const float xs = val[e] * cscale;
                            const _Float16 a16 = (_Float16)xs;
                            hh8[e] = a16;
                            ll8[e] = (_Float16)(xs - (float)a16);
                        }
                        if (p.part) St[i * 33 + 16 * hq + 8 * g + e] = val[e];
                    }
                    if (PK) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hh8), rsH, voffp, (4 * t + 2 * hq) * p.HW * 16, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ll8), rsL, voffp, (4 * t + 2 * hq) * p.HW * 16, 0);
                    }
                }
                if (p.part) {   // row statistics over this wave's 32 pixels: see the masked path below
                    float sm = 0.f, sq = 0.f, mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) {
                        const float x = St[(16 * g + cc) * 33 + i];
                        sm += x;
                        sq = fmaf(x, x, sq);
                        mn = fminf(mn, x);
                        mx = fmaxf(mx, x);
                    }
                    sm += __shfl_xor(sm, 32, 64);
                    sq += __shfl_xor(sq, 32, 64);
                    mn = fminf(mn, __shfl_xor(mn, 32, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    if (g == 0) p.part[((long)smp * p.nstrips32 + (n0 >> 5)) * p.Cch + 32 * t + i] = make_float4(sm, sq, mn, mx);
                }
            });
        });
    } else {
    // ---- ragged strip (last workgroup): masked, plain addressing
    // the input strip is dead: its registers take the residual strip in TWO batches of loads (per-tile loads cost a memory
    // round trip under full load each: 70 k cycles of epilogue in the r02 timeline)
    static_for<0, 2>([&](auto hh) {
    constexpr int th = decltype(hh)::value;
    float resv[NT2 / 2][16];
    static_for<0, NT2 / 2>([&](auto tt) {
        constexpr int t = th * (NT2 / 2) + decltype(tt)::value;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq)
#pragma unroll
            for (int e = 0; e < 8; ++e) resv[t - th * (NT2 / 2)][8 * hq + e] = R[(long)(32 * t + 16 * hq + 8 * g + e) * HW];
    });
    static_for<0, NT2 / 2>([&](auto tt) {
        constexpr int tl = decltype(tt)::value;
        constexpr int t = th * (NT2 / 2) + tl;
        f32x16 v = out[t];
        rows_to_kgroups(v);      // rows 8 g + e and 16 + 8 g + e: whole P entries, 8 consecutive channel rows
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const int row0 = 32 * t + 16 * hq + 8 * g;
            const f32x4 b2a = *reinterpret_cast<const f32x4*>(p.b2 + row0);
            const f32x4 b2b = *reinterpret_cast<const f32x4*>(p.b2 + row0 + 4);
            f32x4 sa = {1.f, 1.f, 1.f, 1.f}, sb = sa, ta = {0.f, 0.f, 0.f, 0.f}, tb = ta;
            if (rsc) {
                sa = *reinterpret_cast<const f32x4*>(rsc + row0); sb = *reinterpret_cast<const f32x4*>(rsc + row0 + 4);
                ta = *reinterpret_cast<const f32x4*>(rsh + row0); tb = *reinterpret_cast<const f32x4*>(rsh + row0 + 4);
            }
            float val[8];
            half8 hh, ll;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float bb = e < 4 ? b2a[e & 3] : b2b[e & 3];
                const float rs = e < 4 ? sa[e & 3] : sb[e & 3];
                const float rt = e < 4 ? ta[e & 3] : tb[e & 3];
                val[e] = fmaf(v[8 * hq + e], s_fc2, bb) + fmaf(resv[tl][8 * hq + e], rs, rt);
                if (nok) Cc[(long)(row0 + e) * HW] = val[e];
                vmax = fmaxf(vmax, nok ? fabsf(val[e]) : 0.f);
                if (PK) {
                    const float xs = val[e] * cscale;
                    const _Float16 a = (_Float16)xs;
                    hh[e] = a;
                    ll[e] = (_Float16)(xs - (float)a);
                }
            }
            if (PK && nok) {
                const long eo = ((long)(row0 >> 3) * HW + n) * 8;
                *reinterpret_cast<half8*>(p.Chi + (long)smp * p.sCp + eo) = hh;
                *reinterpret_cast<half8*>(p.Clo + (long)smp * p.sCp + eo) = ll;
            }
            if (p.part) {   // wave-uniform: park the final values transposed for the row statistics below
#pragma unroll
                for (int e = 0; e < 8; ++e) St[i * 33 + 16 * hq + 8 * g + e] = val[e];
            }
        }
        if (p.part) {
            // Row statistics over this wave's 32 pixels (sum, sum of squares, min, max) through a per-wave LDS transpose:
            // value (row, column) sits at column * 33 + row, lane (i, g) walks row i over the columns 16 g .. 16 g + 15
            // (conflict-free both ways), the two half-waves meet in one exchange.  ~110 instructions per tile instead of
            // ~600 for a shuffle butterfly over 16 registers x 4 quantities.
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
            if (g == 0) p.part[((long)smp * p.nstrips32 + (n0 >> 5)) * p.Cch + 32 * t + i] = make_float4(sm, sq, mn, mx);
        }
    });
    });
    }
    MT(6);
    if (p.omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
}

}  // namespace

#ifdef ACE_X_TRACE
extern "C" int ace_debug_mlp_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mlp_trace), sizeof(mlp_trace)); }
#endif

// A-fragment packing of a conv weight W (O x I, row pitch ldw), optionally with a per-input-channel scale folded in
// (instance-norm affine: W diag(a)), as fp16 hi/lo blocks of 64 lanes x 8 halves (strip_pack.h layout: lane = i + 32 g holds
// row 32 T + i, columns 16 J + 8 g .. + 7):
//   order 0 (streamed by output-row chunk, fc1): block (T, J) at T * (I / 16) + J
//   order 1 (streamed by 16-column step, fc2):   block (T, J) at J * (O / 32) + T
// O % 32 == 0, I % 16 == 0.  scale: a power of two, or derived from `bound` = wmax * max|a| (published to wslot).
__global__ __launch_bounds__(256) void pack_conv_frag_kernel(const float* __restrict__ W, long ldw, int O, int I, int order,
                                                             const float* __restrict__ a, float wmax, float scale_static,
                                                             unsigned* wslot, _Float16* __restrict__ dst, long sDst,
                                                             const float* __restrict__ b, const float* __restrict__ bias,
                                                             float* __restrict__ bf) {
    const int smp = blockIdx.y;
    float scale = scale_static;
    if (a) {   // one scale for all samples (as fold_affine_f16_kernel)
        __shared__ float red[4];
        float am = 0.f;
        for (int q = threadIdx.x; q < I * (int)gridDim.y; q += 256) am = fmaxf(am, fabsf(a[q]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
        __syncthreads();
        am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float bound = wmax * am;
        scale = ldexpf(1.0f, pow2_exponent_for(bound));
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(wslot + (smp & 63), __float_as_uint(bound));
    }
    const int nJ = I / 16, nT = O / 32;
    const int blk = blockIdx.x;                 // one workgroup = one (T, J) block: 512 elements, two per thread
    const int T = blk / nJ, J = blk % nJ;
    const long bidx = order == 0 ? (long)T * nJ + J : (long)J * nT + T;
    _Float16* out = dst + (long)smp * sDst + bidx * 1024;
    const float* as = a ? a + (long)smp * I : nullptr;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = threadIdx.x + 256 * u;    // = lane * 8 + e
        const int e = t & 7, lane = t >> 3, i = lane & 31, g = lane >> 5;
        const int row = 32 * T + i, col = 16 * J + 8 * g + e;
        float x = W[(long)row * ldw + col] * scale;
        if (as) x *= as[col];
        x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
        const _Float16 h = (_Float16)x;
        out[t] = h;
        out[512 + t] = (_Float16)(x - (float)h);
    }
    if (bf && J == 0) {   // folded bias of the 32 rows of this tile: bias + W b (8 threads per row)
        const int r = threadIdx.x >> 3, l8 = threadIdx.x & 7;
        const long row = 32L * T + r;
        const float* bs = b + (long)smp * I;
        float acc = 0.f;
        for (int q = l8; q < I; q += 8) acc = fmaf(W[row * ldw + q], bs[q], acc);
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (l8 == 0) bf[(long)smp * O + row] = (bias ? bias[row] : 0.f) + acc;
    }
}

hipError_t launch_pack_conv_frag(const float* W, long ldw, int O, int I, int order, const float* a, float wmax,
                                 float scale_static, unsigned* wslot, void* dst, long sDst, int nsamples, hipStream_t s,
                                 const float* b, const float* bias, float* bf) {
    if (O % 32 != 0 || I % 16 != 0 || (order == 1 && I % 32 != 0) || (a && !wslot) || (bf && !b)) return hipErrorInvalidValue;
    dim3 grid((unsigned)((O / 32) * (I / 16)), (unsigned)nsamples);
    hipLaunchKernelGGL(pack_conv_frag_kernel, grid, dim3(256), 0, s, W, ldw, O, I, order, a, wmax, scale_static, wslot,
                       static_cast<_Float16*>(dst), sDst, b, bias, bf);
    return hipGetLastError();
}

bool mlp_strip_shape_ok(int C, int hid) { return (C == 128 || C == 256 || C == 384) && hid % 32 == 0 && hid >= 64 && hid <= 4096; }

bool mlp_strip_eligible(int C, int hid, int act) {
    // Opt-in (ACE_MLP_FUSED=1, read per call so that tests can flip it): at the ACE2 shape the fused kernel (one wave per
    // SIMD, 306 us) is slower than fc1 on the strip convolution + fc2 on the v4 tile engine (126 + 179 us) - r02 measurements
    const char* e = std::getenv("ACE_MLP_FUSED");
    if (!(e && e[0] && e[0] != '0')) return false;
    if (!(act == ACT_GELU || act == ACT_GELU_FAST)) return false;
    return mlp_strip_shape_ok(C, hid);
}

template <int NC>
static hipError_t launch_mlp_nc(const MlpStripArgs& a, hipStream_t s) {
    const int strips = (a.HW + 127) / 128;
    dim3 grid((unsigned)(strips * a.nbatch)), block(256);
    if (a.Chi) hipLaunchKernelGGL((mlp_strip_kernel<NC, ACT_GELU_FAST, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((mlp_strip_kernel<NC, ACT_GELU_FAST, false>), grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_mlp_strip(const MlpStripArgs& a, hipStream_t s) {
    if (!mlp_strip_shape_ok(a.Cch, a.hid) || !(a.act == ACT_GELU || a.act == ACT_GELU_FAST) || !a.C || !a.R) return hipErrorInvalidValue;
    if (a.Chi && (!a.cslot || !a.rmax)) return hipErrorInvalidValue;
    switch (a.Cch / 32) {
        case 4: return launch_mlp_nc<4>(a, s);
        case 8: return launch_mlp_nc<8>(a, s);
        case 12: return launch_mlp_nc<12>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ace
