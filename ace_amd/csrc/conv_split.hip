// 1x1 convolutions of the SFNO block on the packed path, contraction SPLIT over the two waves of a SIMD, for gfx950:
//     out = epilogue(A . P(x) + bias [+ residual])        A: (M x K) conv weight, x: K x HW activation (P-format fp16 hi/lo planes)
// covering the block's inner skip (sfnonet.py:229-232), the first MLP convolution (layers.py:117-124) and the second one
// with the outer skip (layers.py:117-124, sfnonet.py:246-250), K up to 768.
//
// conv_strip.hip keeps a whole K x 32-pixel strip in one wave's registers (192 VGPRs at K = 384): no room is left for a
// second accumulator tile, so the GELU / split / statistics epilogue of a tile (about as many VALU cycles as the tile has MFMA
// cycles) runs between the MFMA phases instead of under them, and K = 768 (the second MLP convolution) does not fit at all.
// Here the two waves that share a SIMD share a 32-pixel strip and each holds HALF of the contraction:
//   * wave (strip s, half h) keeps k16-steps [h KH, (h + 1) KH) of the strip as MFMA B fragments (96 VGPRs at K = 384, 192
//     at K = 768) and accumulates a partial 32 x 32 tile per output tile over its half;
//   * at the end of a tile the partners swap halves of their accumulators through LDS (2 KiB per wave, double buffered, no
//     extra barrier: the stage barrier of the ring orders it), so that each FINISHES 16 of the 32 rows: the epilogue work is
//     halved per wave and - for the GELU epilogues - spread over the k-steps of the NEXT tile, between its MFMAs, in one
//     basic block (no masks: ragged strips are handled by out-of-range buffer offsets);
//   * the weights arrive as pre-packed A fragments (strip_pack.h layout, launch_pack_conv_frag order 0) through a two-stage
//     LDS ring of 1-KiB LDS-DMA pieces; a stage is KSW k-steps of BOTH halves (48 KiB at KSW = 12), a tile is NSTG stages
//     (1 for K <= 384, 2 for K = 512 / 768);
//   * residual rows are fetched a stage ahead into registers (buffer loads, retired by the stage-top wait).
// Workgroup = 8 waves = 4 strips = 128 pixels; 507 workgroups at 180 x 360, two rounds on 256 CUs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "strip_common.h"

namespace ace {
namespace {

#ifndef ACE_X_TRACE_MODE
#define ACE_X_TRACE_MODE 1   // measurement builds (-DACE_X_TRACE=<block>): which MODE records its timeline
#endif
#define MTQ(ev) do { if (MODE == ACE_X_TRACE_MODE) MT(ev); } while (0)

constexpr int OOBV = 0x7fffff00;   // buffer offset beyond every resource of these kernels: loads return 0, stores are dropped

// MODE 0: inner skip  (fp32 residual, GELU, P-format planes + row statistics)
//      1: fc1         (GELU, P-format planes)
//      2: fc2, mid    (residual with per-row affine, fp32 output + P-format planes + row statistics)
//      3: fc2, last   (residual with per-row affine, fp32 output, optional range maximum)
template <int KSW, int NSTG, int MODE>
struct SplitGeom {
    static constexpr int KH = KSW * NSTG;          // k16-steps per wave: its half of the contraction
    static constexpr int SLOT = 2 * KSW * 2048;    // one stage of A fragments, both halves
    static constexpr bool F32 = MODE >= 2, STATS = MODE == 0 || MODE == 2;
    static constexpr int BMAX = F32 ? 1024 : 2048; // output rows with LDS-resident epilogue parameters
    static constexpr int TAB = BMAX * 4 * (F32 ? 2 : 1);
    static constexpr int XCH = 2 * 8 * 2048;       // accumulator exchange: [tile parity][wave][2 planes][64 lanes][16 B]
    static constexpr int STP = 36;                 // pitch of the statistics transpose (floats)
    static constexpr int STB = STATS ? 8 * 16 * STP * 4 : 0;
    static constexpr int LDS = 2 * SLOT + TAB + XCH + STB;
    static_assert(KSW % 2 == 0 && LDS <= 160 * 1024, "LDS budget");
};

// H: contraction half of the calling wave (compile time: the accumulator halves kept / handed over are then plain register
// names; a run-time h made hipcc index the register file through M0 or spill)
template <int KSW, int NSTG, int MODE, int H>
MDEV void conv_split_body(const ConvStripArgs& p, char* smem) {
    using G = SplitGeom<KSW, NSTG, MODE>;
    constexpr int KH = G::KH, SLOT = G::SLOT, BMAX = G::BMAX, TAB = G::TAB, XCH = G::XCH, STP = G::STP;
    constexpr int PW = KSW / 2;             // 1-KiB pieces per wave per stage
    constexpr bool GELU = MODE <= 1, RES = MODE != 1, PK = MODE != 3, F32 = MODE >= 2, STATS = G::STATS;
    constexpr bool INTER = MODE <= 1;       // epilogue of tile t - 1 between the MFMAs of tile t
    constexpr int h = H;
    float* Pb = reinterpret_cast<float*>(smem + 2 * SLOT);     // bias (+ residual shift)
    float* Ps = Pb + BMAX;                                      // residual scale (F32 modes)
    char* xch = smem + 2 * SLOT + TAB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    float* St = reinterpret_cast<float*>(smem + 2 * SLOT + TAB + XCH) + wave * (16 * STP);
    const int wgs = (p.HW + 127) / 128;
    const int smp = blockIdx.x / wgs;
    const int n0 = (blockIdx.x % wgs) * 128 + (wave & 3) * 32;
    const int n = n0 + i;
    const int nc = n < p.HW ? n : p.HW - 1;
    const bool nok = n < p.HW;
    const int ntiles = p.M / 32;
    const int NU = ntiles * NSTG;

    MTQ(0);
    const unsigned raw_x = slot_load(p.xslot + lane);
    const unsigned raw_a = p.aslot ? slot_load(p.aslot + lane) : 0u;
    const unsigned raw_c = p.cinb ? slot_load(p.cinb + lane) : 0u;
    const unsigned raw_r = p.rmax ? slot_load(p.rmax + lane) : 0u;

    const _Float16* A = p.A + (long)smp * p.sA;
    auto piece = [&](int u, int k) {   // piece k (of PW) of this wave for stage u -> slot u % 2; stages past the end re-fetch the last
        const int uu = u < NU ? u : NU - 1;
        const int t = uu / NSTG, q = uu % NSTG;
        const int pc = wave + 8 * k;                 // 0 .. 4 KSW - 1: block pc / 2 of the stage, hi / lo KiB
        const int bl = pc >> 1, hf = pc & 1;
        const int hh = bl / KSW, jj = bl % KSW;
        const long blk = (long)t * (2 * KH) + hh * KH + q * KSW + jj;
        glds16(A + blk * 1024 + hf * 512 + lane * 8, smem + (u & 1) * SLOT + pc * 1024);
    };
#pragma unroll
    for (int k = 0; k < PW; ++k) piece(0, k);

    // ---- resident input strip: this wave's half of the contraction
    half8 xh[KH], xl[KH];
    {
        const _Float16* Xh = p.Xhi + (long)smp * p.sX;
        const _Float16* Xl = p.Xlo + (long)smp * p.sX;
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            const long off = ((long)(2 * (h * KH + j) + g) * p.ldn + nc) * 8;
            xh[j] = *reinterpret_cast<const half8*>(Xh + off);
            xl[j] = *reinterpret_cast<const half8*>(Xl + off);
        }
    }
    {   // epilogue parameters of this sample -> LDS
        const float* b = p.bias + (long)smp * p.sbias;
        const float* rsc = (F32 && p.rsc) ? p.rsc + (long)smp * p.srs : nullptr;
        const float* rsh = (F32 && p.rsc) ? p.rsh + (long)smp * p.srs : nullptr;
#pragma unroll
        for (int k = 0; k < BMAX / 512; ++k) {
            const int r = tid + 512 * k;
            if (r < p.M) {
                Pb[r] = b[r] + (rsh ? rsh[r] : 0.f);
                if (F32) Ps[r] = rsc ? rsc[r] : 1.f;
            }
        }
    }
    MTQ(1);
    const float xbound = wave_max_bits(raw_x);
    const float inv_x = ldexpf(1.0f, -pow2_exponent_for(xbound));
    const float inv_a = p.aslot ? ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_a))) : 1.0f / p.ascale;
    const float s_acc = inv_x * inv_a;
    float cscale = 1.f;
    if (PK) {   // bound of this launch's output, identical in every workgroup; the consumer reads it from cslot
        const float inb = p.cinb ? wave_max_bits(raw_c) : xbound;
        const float resb = p.rmax ? wave_max_bits(raw_r) : 0.f;
        const float cbound = fmaf(p.cw, inb, p.cb) + resb;
        cscale = ldexpf(1.0f, pow2_exponent_for(cbound));
        if (tid == 0) atomicMax(p.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
    }

    MTQ(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage 0, the input strip, the tables' sources
#pragma unroll
    for (int j = 0; j < KH; ++j) asm volatile("" : "+v"(xh[j]), "+v"(xl[j]));
    MTQ(3);
    __syncthreads();
    MTQ(4);

    const int fbytes = p.M * p.HW * 4, pbytes = p.M * p.HW * 2;
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc(RES ? const_cast<float*>(p.R + (long)smp * p.sR) : nullptr, 0, RES ? fbytes : 0, 0x00020000);
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc(F32 ? p.Cf + (long)smp * p.sCf : nullptr, 0, F32 ? fbytes : 0, 0x00020000);
    const auto rsH = __builtin_amdgcn_make_buffer_rsrc(PK ? p.Chi + (long)smp * p.sCp : nullptr, 0, PK ? pbytes : 0, 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(PK ? p.Clo + (long)smp * p.sCp : nullptr, 0, PK ? pbytes : 0, 0x00020000);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(
        STATS ? p.part + ((long)smp * p.nstrips32 + (n0 >> 5)) * p.M : nullptr, 0, STATS ? p.M * 16 : 0, 0x00020000);
    const int rowb = p.HW * 4;
    const int vf_ok = nok ? (8 * g * p.HW + n) * 4 : OOBV;      // fp32 element (row 8 g, column n)
    const int vp_ok = nok ? (g * p.HW + n) * 16 : OOBV;         // P entry (k group g, column n)
    const int vs_ok = lane < 16 ? lane * 16 : OOBV;             // statistics row (lane & 15), written by the lanes of column quarter 0
    const int ncols_ok = p.HW - n0 < 32 ? (p.HW - n0 > 0 ? p.HW - n0 : 0) : 32;

    // ---- epilogue state of the tile being finished: rows 32 tp + 16 h + 8 g + e of column i
    float own[8], res[8], resn[8];
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = pa;
    struct EpiOut {                        // data registers of the epilogue's stores (see the note at the end of the stage)
        half8 hh8, ll8;
        float vals[8];
        f32x4 stv;
    };
    EpiOut eo;
    float vmax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { own[e] = 0.f; res[e] = 0.f; resn[e] = 0.f; eo.vals[e] = 0.f; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { eo.hh8[e] = eo.ll8[e] = (_Float16)0.f; }
    eo.stv = f32x4{0.f, 0.f, 0.f, 0.f};

    // item k < 8: value e = k; item 8: the stores of whole P entries and the row statistics
    auto epi_item = [&](auto kc, const int tp, const int vo_f, const int vo_p, const int vo_s, EpiOut& o) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 8) {
            constexpr int e = k;
            const int row = 32 * tp + 16 * h + 8 * g + e;
            float val = fmaf(own[e] + (e < 4 ? pa[e & 3] : pb[e & 3]), s_acc, Pb[row]);
            if (RES) val = F32 ? fmaf(resn[e], Ps[row], val) : val + res[e];   // light epilogues consume the fetched rows in place
            if (GELU) val = act_fn<ACT_GELU_FAST>(val);
            if (F32) {
                o.vals[e] = val;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.vals[e]), rsC, vo_f, (32 * tp + 16 * h + e) * rowb, 0);
                vmax = fmaxf(vmax, vo_f != OOBV ? fabsf(val) : 0.f);
            }
            if (STATS) St[(8 * g + e) * STP + i] = val;
            if (PK) {
                const float xs = val * cscale;
                const _Float16 a16 = (_Float16)xs;
                o.hh8[e] = a16;
                o.ll8[e] = (_Float16)(xs - (float)a16);
            }
        } else {
            if (PK) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.hh8), rsH, vo_p, (4 * tp + 2 * h) * p.HW * 16, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.ll8), rsL, vo_p, (4 * tp + 2 * h) * p.HW * 16, 0);
            }
            if (STATS) {
                // lane (r = lane & 15, cq = lane >> 4) reduces columns 8 cq .. 8 cq + 7 of row r; the four quarters meet in two exchanges
                const int r = lane & 15, cq = lane >> 4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(St + r * STP + 8 * cq);
                const f32x4 b = *reinterpret_cast<const f32x4*>(St + r * STP + 8 * cq + 4);
                float sm = 0.f, sq = 0.f, mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = c < 4 ? a[c & 3] : b[c & 3];
                    const bool ok = 8 * cq + c < ncols_ok;
                    sm += ok ? x : 0.f;
                    sq = ok ? fmaf(x, x, sq) : sq;
                    mn = ok ? fminf(mn, x) : mn;
                    mx = ok ? fmaxf(mx, x) : mx;
                }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    sm += __shfl_xor(sm, off, 64);
                    sq += __shfl_xor(sq, off, 64);
                    mn = fminf(mn, __shfl_xor(mn, off, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                }
                o.stv = f32x4{sm, sq, mn, mx};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.stv), rsP, vo_s, (32 * tp + 16 * h) * 16, 0);
            }
        }
    };
    auto load_residual = [&](int t) {   // rows of tile t this wave will finish -> resn (un-waited; retired by the next stage-top wait)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(resn[e]) : "v"(vf_ok), "s"(rsR), "s"((32 * t + 16 * h + e) * rowb));
    };
    auto stage_top = [&](int u = 0) {
        MTQ(12 + 6 * u);
        // lgkmcnt: the accumulator halves written for the partner must be IN the LDS before the barrier releases it (a bare
        // s_barrier does not wait for them; hipcc only adds that wait to __syncthreads)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(resn[0]), "+v"(resn[1]), "+v"(resn[2]), "+v"(resn[3]), "+v"(resn[4]), "+v"(resn[5]), "+v"(resn[6]), "+v"(resn[7])
                     :
                     : "memory");
        MTQ(13 + 6 * u);
        __builtin_amdgcn_s_barrier();      // the stage landed in every wave's share; every wave is done with the previous one
    };
    auto read_partner = [&](int t) {       // the partner's half of tile t
        const char* xr = xch + (((t & 1) * 8) + (wave ^ 4)) * 2048 + lane * 16;
        pa = *reinterpret_cast<const f32x4*>(xr);
        pb = *reinterpret_cast<const f32x4*>(xr + 1024);
    };

    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
    // epilogue items of the previous tile: the eight values over the first VSPAN k-steps, the stores right after - early
    // enough that the stage-closing vmcnt(0) (gfx950 counts stores) finds them retired
    constexpr int VSPAN = 2 * KSW / 3;
    constexpr int FDEPTH = (MODE == 2 && KH == 24) ? 0 : 1;   // fragment read-ahead; 0 where the registers are gone (K = 768 + planes + statistics)
    // The loop body is one stage FOLLOWED by the wait + barrier that opens the next one (stage 0 was opened by the prologue):
    // the residual rows are requested, retired and copied inside ONE iteration.  An asm-loaded register that is still in
    // flight must not be live across the back edge - hipcc believes the value is there and is free to move it (it placed
    // v_mov copies of in-flight registers in the middle of the stage: r02, nondeterministic results).
    for (int t = 0; t < ntiles; ++t) {
        const bool live = t > 0;
        const int tp = live ? t - 1 : 0;
        const int vo_f = live ? vf_ok : OOBV, vo_p = live ? vp_ok : OOBV, vo_s = live ? vs_ok : OOBV;
        static_for<0, NSTG>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int u = t * NSTG + q;
            MTQ(8 + 6 * u);
            // the PW pieces of stage u + 1 (into the slot of stage u - 1, free since the barrier): one per k-step from the first on,
            // between the MFMAs (an LDS-DMA issue holds the wave's issue slot for ~80 cycles: 470 of 4550 cycles per stage
            // when all six went out at the top - r02 in-kernel timeline); up front when the stage starts with an epilogue
            constexpr bool SPREAD = INTER;          // (the K = 768 kernels have no registers for the piece addresses)
            // K = 768: no registers to hold the store data of the epilogue through the stage either - the stores are retired
            // before the fragment pipeline starts, the pieces go out after that
            constexpr bool TIGHT = !INTER && KH == 24;
            if constexpr (!SPREAD && !(TIGHT && q == 0)) {
#pragma unroll
                for (int k = 0; k < PW; ++k) piece(u + 1, k);
            }
            MTQ(9 + 6 * u);
            if constexpr (q == 0) {
                read_partner(tp);
                if constexpr (!INTER)   // light epilogue: all of it before the MFMAs of this tile
                    static_for<0, 9>([&](auto kc) { epi_item(kc, tp, vo_f, vo_p, vo_s, eo); });
                if constexpr (TIGHT) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // (the store data stays where it is until here: see the note at the end of the stage)
                    if constexpr (PK) asm volatile("" ::"v"(eo.hh8), "v"(eo.ll8));
                    if constexpr (STATS) asm volatile("" ::"v"(eo.stv));
                    asm volatile("" ::"v"(eo.vals[0]), "v"(eo.vals[1]), "v"(eo.vals[2]), "v"(eo.vals[3]), "v"(eo.vals[4]), "v"(eo.vals[5]), "v"(eo.vals[6]), "v"(eo.vals[7]));
#pragma unroll
                    for (int k = 0; k < PW; ++k) piece(u + 1, k);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = 0.f;
            }
            if constexpr (RES && q == NSTG - 1) load_residual(t);
            MTQ(10 + 6 * u);
            const unsigned sl = (unsigned)(size_t)(lds_cptr)(smem + (u & 1) * SLOT) + h * (KSW * 2048) + lane * 16;
            pipelined_steps<KSW, FDEPTH>(sl, [&](auto ss, const Frag& f) {
                constexpr int st = decltype(ss)::value;
                constexpr int j = q * KSW + st;
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.l, xh[j], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, xl[j], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, xh[j], v, 0, 0, 0);
                if constexpr (SPREAD && st < PW) piece(u + 1, st);
                if constexpr (INTER && q == 0) {
                    static_for<0, 8>([&](auto kc) {
                        constexpr int e = decltype(kc)::value;
                        if constexpr ((e * VSPAN) / 8 == st) epi_item(kc, tp, vo_f, vo_p, vo_s, eo);
                    });
                    if constexpr (st == VSPAN) epi_item(std::integral_constant<int, 8>{}, tp, vo_f, vo_p, vo_s, eo);
                }
            });
            MTQ(11 + 6 * u);
            if constexpr (q == NSTG - 1) {   // tile complete: keep the rows this wave finishes, hand the others to the partner
                rows_to_kgroups(v);
                f32x4 sa, sb;
#pragma unroll
                for (int e = 0; e < 4; ++e) { sa[e] = v[8 * (1 - H) + e]; sb[e] = v[8 * (1 - H) + 4 + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) own[e] = v[8 * H + e];
                char* xw = xch + (((t & 1) * 8) + wave) * 2048 + lane * 16;
                *reinterpret_cast<f32x4*>(xw) = sa;
                *reinterpret_cast<f32x4*>(xw + 1024) = sb;
            }
            stage_top(u);                  // opens stage u + 1 (after the last one: the partner's half and the residual rows are there)
            // The registers a buffer store takes its data from stay untouched until the store has RETIRED (the vmcnt(0) above).
            // Measured on gfx950 (r02): with the store data in registers that the fragment pipeline re-used a few
            // instructions later (an inline-asm ds_read_b128 whose data lands asynchronously), lanes 12 - 15 of each 16 of
            // the second data dword reached memory with the NEW contents; holding the registers until here removed it.
            constexpr bool KEEP = q == 0 && !(!INTER && KH == 24);
            if constexpr (PK && KEEP) asm volatile("" ::"v"(eo.hh8), "v"(eo.ll8));
            if constexpr (STATS && KEEP) asm volatile("" ::"v"(eo.stv));
            if constexpr (F32 && KEEP) asm volatile("" ::"v"(eo.vals[0]), "v"(eo.vals[1]), "v"(eo.vals[2]), "v"(eo.vals[3]), "v"(eo.vals[4]), "v"(eo.vals[5]), "v"(eo.vals[6]), "v"(eo.vals[7]));
            if constexpr (INTER && RES && q == NSTG - 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) res[e] = resn[e];
            }
        });
    }
    // ---- last tile (its own set of store-data registers, every field written before it is stored: holding THIS set to the end
    //      of the program does not pin registers inside the loop)
    EpiOut eo_last;
    MTQ(5);
    read_partner(ntiles - 1);
    static_for<0, 9>([&](auto kc) { epi_item(kc, ntiles - 1, vf_ok, vp_ok, vs_ok, eo_last); });
    // ... and the data registers of these stores are held to the end of the program (nothing may land in them: the
    // statistics' LDS reads did, r02); no wait - it would keep the workgroup on the CU for a store round trip
    if constexpr (PK) asm volatile("" ::"v"(eo_last.hh8), "v"(eo_last.ll8));
    if constexpr (STATS) asm volatile("" ::"v"(eo_last.stv));
    if constexpr (F32) asm volatile("" ::"v"(eo_last.vals[0]), "v"(eo_last.vals[1]), "v"(eo_last.vals[2]), "v"(eo_last.vals[3]), "v"(eo_last.vals[4]), "v"(eo_last.vals[5]), "v"(eo_last.vals[6]), "v"(eo_last.vals[7]));
    MTQ(6);
    if (F32 && p.omax) {                   // one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float* red = reinterpret_cast<float*>(smem);   // the ring is dead
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
            for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
            atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(m));
        }
    }
}

template <int KSW, int NSTG, int MODE>
__global__ __launch_bounds__(512) void conv_split_kernel(ConvStripArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[SplitGeom<KSW, NSTG, MODE>::LDS];
    if (threadIdx.x < 256) conv_split_body<KSW, NSTG, MODE, 0>(p, smem);   // waves 0 - 3: first half of the contraction
    else conv_split_body<KSW, NSTG, MODE, 1>(p, smem);
}

template <int KSW, int NSTG>
hipError_t launch_split_k(const ConvStripArgs& a, int mode, hipStream_t s) {
    const int wgs = (a.HW + 127) / 128;
    dim3 grid((unsigned)(wgs * a.nbatch)), block(512);
    if constexpr (NSTG == 1) {   // the GELU modes exist for the single-stage contractions only (K <= 384)
        if (mode == 0) hipLaunchKernelGGL((conv_split_kernel<KSW, NSTG, 0>), grid, block, 0, s, a);
        if (mode == 1) hipLaunchKernelGGL((conv_split_kernel<KSW, NSTG, 1>), grid, block, 0, s, a);
    } else if (mode <= 1) {
        return hipErrorInvalidValue;
    }
    if (mode == 2) hipLaunchKernelGGL((conv_split_kernel<KSW, NSTG, 2>), grid, block, 0, s, a);
    if (mode == 3) hipLaunchKernelGGL((conv_split_kernel<KSW, NSTG, 3>), grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace

#ifdef ACE_X_TRACE
extern "C" int ace_debug_split_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mlp_trace), sizeof(mlp_trace)); }
#endif

// K: input channels, M: output rows
bool conv_split_eligible(int K, int M, long HW, int role) {
    // Which convolutions run here: ACE_CONV_SPLIT = list of roles ("skip,fc1,fc2"), "all" or "none".  Default "fc2": same-box
    // A/B at the 1-degree shape (r02) has this kernel 1 - 5 % ahead of the 128 x 128 tile engine on fc2 and 7 - 15 % behind
    // conv_strip.hip on the inner skip and fc1.
    const int on = [] {   // read per call (host side, a few times per forward): tests switch it per case
        const char* e = std::getenv("ACE_CONV_SPLIT");
        if (!e) return 4;
        const std::string v(e);
        if (v == "all" || v == "1") return 7;
        int m = 0;
        if (v.find("skip") != std::string::npos) m |= 1;
        if (v.find("fc1") != std::string::npos) m |= 2;
        if (v.find("fc2") != std::string::npos) m |= 4;
        return m;
    }();
    if (role >= 0 && role < 3 && !(on >> role & 1)) return false;
    if (role == -1 && on == 0) return false;
    if (!(K == 128 || K == 256 || K == 384 || K == 512 || K == 768)) return false;
    return M % 32 == 0 && M >= 32 && M <= 2048 && (long)M * HW * 4 < 0x7fffff00L;
}

hipError_t launch_conv_split(const ConvStripArgs& a, hipStream_t s) {
    if (!conv_split_eligible(a.C, a.M, a.HW, -2) || !a.bias || !a.xslot || !a.A) return hipErrorInvalidValue;
    const bool f32 = a.Cf != nullptr, pk = a.Chi != nullptr, stats = a.part != nullptr, res = a.R != nullptr;
    const bool gelu = a.act == ACT_GELU || a.act == ACT_GELU_FAST;
    int mode = -1;
    if (gelu && res && pk && stats && !f32) mode = 0;
    else if (gelu && !res && pk && !stats && !f32) mode = 1;
    else if (a.act == ACT_NONE && res && f32 && pk && stats) mode = 2;
    else if (a.act == ACT_NONE && res && f32 && !pk && !stats) mode = 3;
    if (mode < 0 || (pk && (!a.Clo || !a.cslot)) || (f32 && a.M > 1024)) return hipErrorInvalidValue;
    if (mode <= 1 && a.C > 384) return hipErrorInvalidValue;   // only the fc2 modes are instantiated for the wide contractions
    switch (a.C) {
        case 128: return launch_split_k<4, 1>(a, mode, s);
        case 256: return launch_split_k<8, 1>(a, mode, s);
        case 384: return launch_split_k<12, 1>(a, mode, s);
        case 512: return launch_split_k<8, 2>(a, mode, s);
        case 768: return launch_split_k<12, 2>(a, mode, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ace
