// The 1x1 convolutions of an FNO block (inner skip sfnonet.py:229-232, MLP layers.py:117-124, outer skip sfnonet.py:246-250),
// WEIGHT-stationary, for gfx950:
//     C = epilogue( W . P(x) )        W: (M x K), x: K x HW activation as P-format fp16 hi/lo planes, K in {128 .. 768}
//   * a workgroup owns 128 output channels (four 32-row tiles of W) and a contiguous range of 32-pixel tiles; wave (tile T,
//     half h) keeps rows 32 T .. 32 T + 31 of W over its half of the contraction as MFMA A fragments for the whole range
//     (96 VGPRs at K = 384, 192 at K = 768), loaded once from the packed-fragment form (launch_pack_conv_frag order 0);
//   * the activation streams: a stage is KSW k16-steps of both contraction halves of ONE pixel tile, fetched by 1-KiB LDS-DMA
//     pieces straight from the P-format planes (a piece is 2 k-groups x 32 pixels = one MFMA B fragment) into a two-stage
//     ring; each fragment is read by the four waves that own the four channel tiles;
//   * accumulators: rows = output channels, columns = pixels; the two waves of a pair (the halves of the contraction, on one
//     SIMD) swap accumulator halves through LDS and each finishes 16 rows: scale, bias, residual, GELU, hi/lo split, whole
//     16-byte P entries out (v_permlane32_swap turns accumulator registers into 8-row groups), row statistics;
//   * store data is HELD: the registers a store reads are not written again before the stage-closing vmcnt(0) (gfx950
//     store-data rule, tools/store_hazard.hip);
//   * grid: persistent, 32 workgroups per XCD = every CU.  The channel slices that need the same pixel tiles run on the same
//     XCD (block -> XCD is block % 8) and walk them together, so the activation comes from HBM once and from that XCD's
//     L2 otherwise.  When the slice count does not divide 32 (M = 384: 3 slices) the remaining workgroups of the XCD share
//     one more tile range, each taking a contiguous run of (slice, tile) units - r02 left 16 of the 256 CUs idle.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "kernels.h"
#include "pack_frag.h"
#include "strip_common.h"
#include "ws_plan.h"

namespace ace {
namespace {

constexpr int WS_OOBV = 0x7fffff00;   // buffer offset beyond every resource of these kernels: loads return 0, stores are dropped

#ifndef ACE_WS_ACC2
#define ACE_WS_ACC2 0     // modes (bit 0 inner skip, bit 1 fc1) whose MFMAs alternate between two accumulators (no dependent-issue stalls)
#endif
#ifndef ACE_WS_FINE
#define ACE_WS_FINE 1     // GELU modes: the epilogue of the previous tile goes into the MFMA stream a quarter of a value (~7 VALU) per
#endif                    // MFMA instead of a whole value (~30 VALU) per k-step
#ifndef ACE_WS_HOLD4
#define ACE_WS_HOLD4 0   // mode 4 at K = 768: hold the store data through the stage instead of retiring the stores first
#endif
#ifndef ACE_WS_FD4
#define ACE_WS_FD4 1     // fragment read-ahead of mode 4 at K = 768 (mode 2 there has no registers for it) (r03 same-box: mode 4 155.8 -> 149.1 us; holding the store data instead: 153.2)
#endif
#ifndef ACE_WS_ABL
#define ACE_WS_ABL 0     // measurement builds only (WRONG results), fc2 modes (NSTG == 2): bit 0 no epilogue (values, stores, statistics),
#endif                   // bit 1 no store-retire wait before the next stage's pieces, bit 2 no residual loads, bit 3 no barrier / piece wait
#ifndef ACE_WS_REARLY
#define ACE_WS_REARLY 1   // fc2 modes with the residual as planes: its two loads per tile go out in the tile's first stage (see the loop)
#endif
#ifndef ACE_WS_VSPAN
#define ACE_WS_VSPAN 8    // interleaved epilogue: its eight values are spread over the first VSPAN twelfths of the stage
#endif


#ifndef ACE_WS_MINMAX
#define ACE_WS_MINMAX 0   // 1: the statistics epilogues also track a row's running minimum / maximum (rounds 3-6: the norm finaliser's bound came
#endif                    // from them; it now comes from the variance - kernels.hip: instnorm_finalize_kernel - and the records carry +-3e38 there)
constexpr bool WS_MINMAX = ACE_WS_MINMAX != 0;

// v_min_f32 / v_max_f32 as single instructions: hipcc puts a canonicalising `v_max x, x` in front of fminf / fmaxf on a computed value
__device__ __forceinline__ float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int KSW, int NSTG, int MODE>
struct WsGeom {
    static constexpr int KH = KSW * NSTG;          // k16-steps per wave: its half of the contraction
    static constexpr int SLOT = 2 * KSW * 2048;    // one stage of activation fragments, both halves
    static constexpr bool F32 = MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7, STATS = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 8;
    static constexpr bool AFFRES = MODE >= 2;      // fc2 modes: residual with a per-row affine (scale table Ps, shift folded into Pb)
    static constexpr int BMAX = AFFRES ? 1024 : 2048; // output rows with LDS-resident epilogue parameters
    static constexpr int TAB = BMAX * 4 * (AFFRES ? 2 : 1);
    static constexpr int XCH = 2 * 8 * 2048;       // accumulator exchange: [tile parity][wave][2 planes][64 lanes][16 B]
    static constexpr int STP = 36;                 // pitch of the statistics transpose (floats)
    static constexpr bool RSTATS = MODE == 0;      // statistics accumulated in registers over the workgroup's pixel range
    static constexpr int STB = (STATS && !RSTATS) ? 8 * 16 * STP * 4 : 0;
    static constexpr int LACC = (STATS && !RSTATS) ? 8 * 16 * 16 : 0;   // fc2 modes: running row statistics of a segment, per wave 16 rows x float4
    static constexpr int LDS = 2 * SLOT + TAB + XCH + STB + LACC;
    static_assert(KSW % 2 == 0 && LDS <= 160 * 1024, "LDS budget");
};

// MODE 0: inner skip  (fp32 residual, GELU, P-format planes + row statistics)       1: fc1 (GELU, P-format planes)
//      2: fc2, mid block (residual with per-row affine, fp32 output + P-format planes + row statistics)
//      3: fc2, last block (fp32 output, optional range maximum)
//      4: as 2 with the residual read from P-format planes and NO fp32 output: the residual stream of the blocks exists as planes
//         only (520 -> 420 MB of traffic per launch)                                 5: as 3 with the residual from planes
//      8: as 2 without the fp32 output: the encoder's last convolution when block 0 reads its input as planes everywhere (round 6)
//      6: as 3 with GELU: the inner skip of the noise-conditioned nets (fp32 residual, fp32 output for the conditional norm that
//         follows; any K of this file - round 4 ran it on the packed tile engine at C = 512)
//      7: measurement builds (-DACE_MEASUREMENT_SWITCHES): bias only, fp32 output - the inner skip's GEMM without its epilogue, for the
//         fork experiment of profiles/r06_skip_fork.txt
// Modes 0 / 1 (K <= 384, registers to spare): the epilogue of pixel tile t - 1 runs between the MFMAs of tile t.
// H: contraction half of the calling wave (compile time, see conv_split.hip)
template <int KSW, int NSTG, int MODE, int H>
MDEV void conv_ws_body(const ConvStripArgs& p, char* smem, const WsPlan pl) {
    using G = WsGeom<KSW, NSTG, MODE>;
    constexpr int KH = G::KH, SLOT = G::SLOT, BMAX = G::BMAX, TAB = G::TAB, XCH = G::XCH, STP = G::STP;
    constexpr int PW = KSW / 2;             // 1-KiB pieces per wave per stage
    constexpr bool GELU = MODE <= 1 || MODE == 6, RES = MODE != 1 && MODE != 7, PK = MODE != 3 && MODE != 5 && MODE != 6 && MODE != 7, F32 = G::F32, STATS = G::STATS, RSTATS = G::RSTATS;
    constexpr bool AFFRES = G::AFFRES, RPL = MODE == 4 || MODE == 5;   // RPL: the residual comes as planes
    constexpr bool INTER = MODE <= 1;       // epilogue of tile t - 1 between the MFMAs of tile t
    constexpr bool HOLD = KH < 24 || (MODE == 4 && ACE_WS_HOLD4);   // registers to hold the store data through a stage (else: stores retired first; holding them at K = 768 in mode 2 spills 22 VGPRs)
    constexpr bool ACC2 = INTER && ((ACE_WS_ACC2 >> MODE) & 1);
    static_assert(!INTER || NSTG == 1, "interleaved epilogue: single-stage tiles");
    constexpr int h = H;
    float* Pb = reinterpret_cast<float*>(smem + 2 * SLOT);     // bias (+ residual shift)
    float* Ps = Pb + BMAX;                                      // residual scale (fc2 modes)
    char* xch = smem + 2 * SLOT + TAB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    float* St = reinterpret_cast<float*>(smem + 2 * SLOT + TAB + XCH) + wave * (16 * STP);
    f32x4* Lacc = reinterpret_cast<f32x4*>(smem + 2 * SLOT + TAB + XCH + G::STB) + wave * 16;   // LSTATS: this wave's 16 rows

    // ---- which sample, XCD and workgroup of the XCD; its segments (runs of pixel tiles of one channel slice): ws_plan.h
    const int per_smp = 8 * ws_workgroups_per_xcd(pl);
    const int smp = blockIdx.x / per_smp;
    const int bb = blockIdx.x % per_smp;
    const int xcd = bb & 7, w = bb >> 3;
    const WsWork wk = ws_work(pl, (p.HW + 31) / 32, xcd, w);
    const int part_q = wk.part_q, nseg = wk.nseg;
    auto segment = [&](int k) { return ws_segment(pl, wk, w, k); };
    if (STATS) {
        // The norm finaliser sums slot part_q over ALL rows.  An extra workgroup owns its slot alone: rows of the slices it
        // does not reach get a neutral partial; a group's slot is shared by its nslice workgroups, each answers for its slice.
        float4* pq = p.part + ((long)smp * p.nstrips32 + part_q) * p.M;
        for (int r = tid; r < p.M; r += 512)
            if (ws_answers_for(pl, wk, w, r >> 7) && !ws_reaches(pl, wk, w, r >> 7)) pq[r] = make_float4(0.f, 0.f, 3.0e38f, -3.0e38f);
    }
    if (nseg <= 0) return;   // (whole workgroup: no barrier has been executed yet)

    const unsigned raw_x = slot_load(p.xslot + lane);
    const unsigned raw_a = p.aslot ? slot_load(p.aslot + lane) : 0u;
    const unsigned raw_c = p.cinb ? slot_load(p.cinb + lane) : 0u;
    const unsigned raw_r = p.rmax ? slot_load(p.rmax + lane) : 0u;
    const unsigned raw_p = (RPL && p.rslot) ? slot_load(p.rslot + lane) : 0u;

    // scale of the residual planes (a power of two), undone in the residual's scale table (RPL) - exact
    const float inv_r = RPL ? ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_p))) : 1.f;
    {   // epilogue parameters of this sample -> LDS
        const float* b = p.bias + (long)smp * p.sbias;
        const float* rsc = (AFFRES && p.rsc) ? p.rsc + (long)smp * p.srs : nullptr;
        const float* rsh = (AFFRES && p.rsc) ? p.rsh + (long)smp * p.srs : nullptr;
#pragma unroll
        for (int k = 0; k < BMAX / 512; ++k) {
            const int r = tid + 512 * k;
            if (r < p.M) {
                Pb[r] = b[r] + (rsh ? rsh[r] : 0.f);
                if (AFFRES) Ps[r] = (rsc ? rsc[r] : 1.f) * inv_r;
            }
        }
    }
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

    const int fbytes = p.M * p.HW * 4, pbytes = p.M * p.HW * 2;
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc((RES && !RPL) ? const_cast<float*>(p.R + (long)smp * p.sR) : nullptr, 0, (RES && !RPL) ? fbytes : 0, 0x00020000);
    const auto rsRh = __builtin_amdgcn_make_buffer_rsrc(RPL ? const_cast<_Float16*>(p.Rhi + (long)smp * p.sRp) : nullptr, 0, RPL ? pbytes : 0, 0x00020000);
    const auto rsRl = __builtin_amdgcn_make_buffer_rsrc(RPL ? const_cast<_Float16*>(p.Rlo + (long)smp * p.sRp) : nullptr, 0, RPL ? pbytes : 0, 0x00020000);
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc(F32 ? p.Cf + (long)smp * p.sCf : nullptr, 0, F32 ? fbytes : 0, 0x00020000);
    const auto rsH = __builtin_amdgcn_make_buffer_rsrc(PK ? p.Chi + (long)smp * p.sCp : nullptr, 0, PK ? pbytes : 0, 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(PK ? p.Clo + (long)smp * p.sCp : nullptr, 0, PK ? pbytes : 0, 0x00020000);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(STATS ? p.part + (long)smp * p.nstrips32 * p.M : nullptr, 0,
                                                       STATS ? p.nstrips32 * p.M * 16 : 0, 0x00020000);
    float vmax = 0.f;
    // ---- one segment: np pixel tiles from tile0 on, channel slice `slice` (weights loaded once per segment)
    auto run_segment = [&](const WsSeg sg, const bool first) {
    const int tile0 = sg.tile0, np = sg.np;
    const int T = sg.slice * 4 + (wave & 3);   // 32-row tile of the output channels
    const int NU = np * NSTG;
    if (!first) __syncthreads();               // the ring and the exchange buffers of the previous segment are free
    if constexpr (STATS && !RSTATS) {          // running row statistics of this segment (lanes 0 - 15 of a wave own its 16 rows)
        if (lane < 16) Lacc[lane] = f32x4{0.f, 0.f, 3.0e38f, -3.0e38f};
    }
    // ---- streamed activation: piece k (of PW) of this wave for stage u -> slot u % 2; stages past the end re-fetch the last.
    //      Piece pc = (block pc / 2 of the stage: half hh, k-step jj; plane pc % 2): lane (i, g) fetches k-group 2 J + g, pixel i
    const _Float16* Xh = p.Xhi + (long)smp * p.sX;
    const _Float16* Xl = p.Xlo + (long)smp * p.sX;
    const unsigned smem_lds = lds_addr_of(smem);
    // wave-uniform part of a piece's address that does not change from stage to stage: plane and k-group pair of (wave, k); the
    // stage adds ONE 64-bit offset (its contraction segment q, its pixel tile) that all pieces of the stage share
    const _Float16* pbase[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        const int pc = wave + 8 * k;
        const int bl = pc >> 1, hf = pc & 1;
        const int hh = bl / KSW, jj = bl % KSW;
        pbase[k] = (hf ? Xl : Xh) + (long)(2 * (hh * KH + jj)) * p.ldn * 8;
    }
    auto piece = [&](int u, int k) {
        const int uu = u < NU ? u : NU - 1;
        const int pt = uu / NSTG, q = uu % NSTG;
        const int pc = wave + 8 * k;
        // address = (uniform: plane, k-group pair 2 J, first pixel of the tile) + (lane: k-group parity g, pixel i - the last tile's
        // missing pixels repeat its last one)
        const int n0 = 32 * (tile0 + pt);
        const int ic = i < p.HW - 1 - n0 ? i : p.HW - 1 - n0;
        const unsigned lane_off = (unsigned)(g * (int)p.ldn + ic) * 16u;
        const long stage_off = ((long)(2 * q * KSW) * p.ldn + n0) * 8;
        glds16a(pbase[k] + stage_off, lane_off, smem_lds + (unsigned)((u & 1) * SLOT + pc * 1024));
    };
#pragma unroll
    for (int k = 0; k < PW; ++k) piece(0, k);

    // ---- resident weights: rows 32 T .. 32 T + 31, this wave's half of the contraction, as A fragments
    half8 wh[KH], wl[KH];
    {
        const _Float16* A = p.A + (long)smp * p.sA + ((long)T * (2 * KH) + h * KH) * 1024 + lane * 8;
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            wh[j] = *reinterpret_cast<const half8*>(A + (long)j * 1024);
            wl[j] = *reinterpret_cast<const half8*>(A + (long)j * 1024 + 512);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage 0, the weights, the tables' sources
#pragma unroll
    for (int j = 0; j < KH; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
    __syncthreads();

    const int row0 = 32 * T + 16 * h;                            // first of the 16 rows this wave finishes
    // per-lane buffer offsets, recomputed where they are used from an opaque copy of the lane index: as loop invariants they
    // (and what hipcc derives from them) would sit in registers this kernel does not have at K = 768
    auto lane_offsets = [&](int& vf_lane, int& vp_lane, int& vs_lane, int& li, int& lg) {
        int lz = lane;
        asm volatile("" : "+v"(lz));
        li = lz & 31; lg = lz >> 5;
        vf_lane = ((row0 + 8 * lg) * p.HW + li) * 4;             // fp32 element (row row0 + 8 g, pixel i of tile 0)
        vp_lane = (((row0 >> 3) + lg) * p.HW + li) * 16;         // P entry (k group row0 / 8 + g, pixel i)
        vs_lane = lz < 16 ? (row0 + lz) * 16 : WS_OOBV;          // statistics row (lane & 15), lanes of column quarter 0
    };

    // ---- epilogue state of the tile being finished: rows row0 + 8 g + e, pixel 32 (tile0 + pt) + i
    float own[8], res[8], resn[8];
    u32x4 rph = {0u, 0u, 0u, 0u}, rpl = rph;   // RPL: the residual's P entry (hi, lo) of the tile being finished
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = pa;
    struct EpiOut {                        // data registers of the epilogue's stores (held until the stores retired)
        unsigned hh[4], ll[4];             // the P entry being assembled: fp16 hi / lo of the eight rows, packed (split_put)
        float vals[8];
        f32x4 stv;
    };
    struct TileCtx { int n0, vo_f, vo_p, vo_s, ncols_ok, i, g; };   // per pixel tile: offsets / validity of this lane
    EpiOut eo;
    // RSTATS (inner skip: registers to spare): running sum / sum of squares / min / max of this lane's pixel column for the
    // eight rows it finishes; reduced over the 32 pixel lanes once, at the end: ONE partial per workgroup and row instead of
    // one per 32-pixel tile (80 instead of 2025 for the norm finaliser at 180 x 360), no LDS transpose, no per-tile store
    float rsm[8], rsq[8], rmn[8], rmx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { rsm[e] = 0.f; rsq[e] = 0.f; rmn[e] = 3.0e38f; rmx[e] = -3.0e38f; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { own[e] = 0.f; res[e] = 0.f; resn[e] = 0.f; eo.vals[e] = 0.f; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { eo.hh[e] = 0u; eo.ll[e] = 0u; }
    eo.stv = f32x4{0.f, 0.f, 0.f, 0.f};

    auto tile_ctx = [&](const int pt, const bool live) {   // live = false: every store of the tile suppressed
        TileCtx c;
        c.n0 = 32 * (tile0 + pt);
        int vf_lane, vp_lane, vs_lane;
        lane_offsets(vf_lane, vp_lane, vs_lane, c.i, c.g);
        const bool nok = live && c.n0 + c.i < p.HW;
        c.vo_f = nok ? vf_lane : WS_OOBV; c.vo_p = nok ? vp_lane : WS_OOBV; c.vo_s = live ? vs_lane : WS_OOBV;
        c.ncols_ok = p.HW - c.n0 < 32 ? (p.HW - c.n0 > 0 ? p.HW - c.n0 : 0) : 32;
        return c;
    };
    // item k < 8: value e = k; item 8: the stores of whole P entries and the row statistics
    auto epi_item = [&](auto kc, const TileCtx& c, EpiOut& o) {
        constexpr int k = decltype(kc)::value;
        const int i = c.i, g = c.g;
        if constexpr (k == 0 && PK) {
            // a NEW entry: split_put modifies the words in place, and the words of the tile before may still be the data of stores in
            // flight (held through a stage, `hold`) - defining them afresh lets the allocator keep those where they are (with the
            // in-place chain alone the 0.25-degree net came out wrong in its last rows: the stores read the next tile's values)
#pragma unroll
            for (int d = 0; d < 4; ++d) asm volatile("" : "=v"(o.hh[d]), "=v"(o.ll[d]));
        }
        if constexpr (k < 8) {
            constexpr int e = k;
            const int row = row0 + 8 * g + e;
            float val = fmaf(own[e] + (e < 4 ? pa[e & 3] : pb[e & 3]), s_acc, Pb[row]);
            if (RES) {
                if constexpr (RPL) {   // rows row0 + 8 g + e of this pixel = ONE P entry: (hi + lo) / scale
                    const half8 rh8 = __builtin_bit_cast(half8, rph), rl8 = __builtin_bit_cast(half8, rpl);
                    val = fmaf((float)rl8[e], Ps[row], fmaf((float)rh8[e], Ps[row], val));   // two v_fma_mix (the table carries 1 / scale)
                } else {
                    val = AFFRES ? fmaf(resn[e], Ps[row], val) : val + res[e];   // light epilogues consume the fetched rows in place
                }
            }
            if (GELU) val = act_fn<ACT_GELU_FAST>(val);
            if (F32) {
                o.vals[e] = val;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o.vals[e]), rsC, c.vo_f, (e * p.HW + c.n0) * 4, 0);
                vmax = fmaxf(vmax, c.vo_f != WS_OOBV ? fabsf(val) : 0.f);
            }
            if (STATS && !RSTATS) St[(8 * g + e) * STP + i] = val;
            if (RSTATS) {
                const bool ok = c.vo_p != WS_OOBV;   // a stored pixel of a live tile
                rsm[e] += ok ? val : 0.f;
                rsq[e] = ok ? fmaf(val, val, rsq[e]) : rsq[e];
                if constexpr (WS_MINMAX) {
                    const float lo_ = raw_min(rmn[e], val), hi_ = raw_max(rmx[e], val);
                    rmn[e] = ok ? lo_ : rmn[e];
                    rmx[e] = ok ? hi_ : rmx[e];
                }
            }
            if (PK) split_put<e>(o.hh, o.ll, cscale, val);
        } else {
            if (PK) {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{o.hh[0], o.hh[1], o.hh[2], o.hh[3]}, rsH, c.vo_p, c.n0 * 16, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{o.ll[0], o.ll[1], o.ll[2], o.ll[3]}, rsL, c.vo_p, c.n0 * 16, 0);
            }
            if (STATS && !RSTATS) {
                // lane (r = lane & 15, cq = lane >> 4) reduces columns 8 cq .. 8 cq + 7 of row r; the four quarters meet in two exchanges
                const int ln = i + 32 * g, r = ln & 15, cq = ln >> 4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(St + r * STP + 8 * cq);
                const f32x4 b = *reinterpret_cast<const f32x4*>(St + r * STP + 8 * cq + 4);
                float sm = 0.f, sq = 0.f, mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const float x = cc < 4 ? a[cc & 3] : b[cc & 3];
                    const bool ok = 8 * cq + cc < c.ncols_ok;
                    const float xs_ = ok ? x : 0.f;                                         // 0 is neutral for the sums
                    sm += xs_;
                    sq = fmaf(xs_, xs_, sq);
                    if constexpr (WS_MINMAX) {
                        const float xm_ = ok ? x : __builtin_nanf("");                      // a NaN for min / max
                        mn = raw_min(mn, xm_);
                        mx = raw_max(mx, xm_);
                    }
                }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    sm += __shfl_xor(sm, off, 64);
                    sq += __shfl_xor(sq, off, 64);
                    if constexpr (WS_MINMAX) {
                        mn = raw_min(mn, __shfl_xor(mn, off, 64));
                        mx = raw_max(mx, __shfl_xor(mx, off, 64));
                    }
                }
                // ONE partial per workgroup and row instead of one per 32-pixel tile (r03: the norm finaliser read 2025 partials per
                // channel, 6 KiB apart, and took 10 us): the row's running statistics live in LDS, always updated by the same lane
                if (c.vo_s != WS_OOBV) {
                    f32x4 acc = Lacc[ln];
                    acc[0] += sm; acc[1] += sq;
                    if constexpr (WS_MINMAX) { acc[2] = raw_min(acc[2], mn); acc[3] = raw_max(acc[3], mx); }
                    Lacc[ln] = acc;
                }
            }
        }
    };
    // quarter `ch` of value e (GELU modes): 0 scale / bias / residual + first GELU stage, 1 second stage, 2 last stage +
    // statistics, 3 hi / lo split
    GeluStage gst = {0.f, 0.f, 0.f};
    // bias of the value about to be finished: read from the LDS table one value ahead (in the previous value's polynomial
    // quarter), so that no LDS latency sits in front of an MFMA
    float bias_next = INTER ? Pb[row0 + 8 * g] : 0.f;
    auto epi_chunk = [&](auto ec, auto cc, const TileCtx& c, EpiOut& o) {
        constexpr int e = decltype(ec)::value, ch = decltype(cc)::value;
        if constexpr (ch == 0) {
            float val = fmaf(own[e] + (e < 4 ? pa[e & 3] : pb[e & 3]), s_acc, bias_next);
            if (RES) val = val + res[e];
            gelu_stage0(gst, val);
            asm volatile("" : "+v"(gst.val), "+v"(gst.t));   // pinned to this slot (pure arithmetic would otherwise
        } else if constexpr (ch == 1) {                                  //  sink to its last use)
            gelu_stage1(gst);
            bias_next = Pb[row0 + 8 * c.g + ((e + 1) & 7)];
            asm volatile("" : "+v"(gst.q));
        } else if constexpr (ch == 2) {
            const float val = gelu_stage2(gst);
            gst.val = val;
            if (RSTATS) {
                const bool ok = c.vo_p != WS_OOBV;   // a stored pixel of a live tile
                // two selects instead of four: 0 is neutral for the sums, a NaN for v_min_f32 / v_max_f32 (they return the other operand)
                const float vs = ok ? val : 0.f;
                rsm[e] += vs;
                rsq[e] = fmaf(vs, vs, rsq[e]);
                if constexpr (WS_MINMAX) {
                    const float vm = ok ? val : __builtin_nanf("");
                    rmn[e] = raw_min(rmn[e], vm);
                    rmx[e] = raw_max(rmx[e], vm);
                    asm volatile("" : "+v"(rmn[e]), "+v"(rmx[e]));
                }
                asm volatile("" : "+v"(rsm[e]), "+v"(rsq[e]));
            }
            asm volatile("" : "+v"(gst.val));
        } else {
            split_put<e>(o.hh, o.ll, cscale, gst.val);
        }
    };
    auto epilogue = [&](const int pt, const bool live, EpiOut& o) {   // all of it at once
        const TileCtx c = tile_ctx(pt, live);
        static_for<0, 9>([&](auto kc) { epi_item(kc, c, o); });
    };
    auto hold = [&](EpiOut& o) {           // gfx950 store-data rule (profiles/r02_store_data_hazard.txt)
        if constexpr (PK) asm volatile("" ::"v"(o.hh[0]), "v"(o.hh[1]), "v"(o.hh[2]), "v"(o.hh[3]), "v"(o.ll[0]), "v"(o.ll[1]), "v"(o.ll[2]), "v"(o.ll[3]));
        if constexpr (F32) asm volatile("" ::"v"(o.vals[0]), "v"(o.vals[1]), "v"(o.vals[2]), "v"(o.vals[3]), "v"(o.vals[4]), "v"(o.vals[5]), "v"(o.vals[6]), "v"(o.vals[7]));
    };
    auto load_residual = [&](int pt) {     // rows of pixel tile pt this wave will finish -> resn (retired by the next stage-top wait)
        const int n0 = 32 * (tile0 + pt);
        int vf_lane, vp_lane, vs_lane, i, g;
        lane_offsets(vf_lane, vp_lane, vs_lane, i, g);
        if constexpr (RPL) {
            const int vo = n0 + i < p.HW ? vp_lane : WS_OOBV;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rph) : "v"(vo), "s"(rsRh), "s"(n0 * 16));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rpl) : "v"(vo), "s"(rsRl), "s"(n0 * 16));
        } else {
            const int vo = n0 + i < p.HW ? vf_lane : WS_OOBV;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(resn[e]) : "v"(vo), "s"(rsR), "s"((e * p.HW + n0) * 4));
        }
    };
    auto stage_top = [&]() {
        // lgkmcnt: the accumulator halves written for the partner must be IN the LDS before the barrier releases it
        if constexpr (RPL)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rph), "+v"(rpl) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(resn[0]), "+v"(resn[1]), "+v"(resn[2]), "+v"(resn[3]), "+v"(resn[4]), "+v"(resn[5]), "+v"(resn[6]), "+v"(resn[7])
                         :
                         : "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto read_partner = [&](int pt) {      // the partner's half of pixel tile pt
        const char* xr = xch + (((pt & 1) * 8) + (wave ^ 4)) * 2048 + lane * 16;
        pa = *reinterpret_cast<const f32x4*>(xr);
        pb = *reinterpret_cast<const f32x4*>(xr + 1024);
    };

    f32x16 v, v2;   // v2: second accumulator of the ACC2 form (the MFMA stream alternates v, v2, v, v2, ...)
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r] = 0.f; v2[r] = 0.f; }
    constexpr int FDEPTH = (MODE == 2 && KH == 24) ? 0 : (((MODE == 4 || MODE == 8) && KH == 24) ? ACE_WS_FD4 : 1);   // fragment read-ahead; 0 where the registers are gone (K = 768 + planes + statistics)
    // Loop body = one stage followed by the wait + barrier that opens the next (stage 0 was opened by the prologue); an
    // asm-loaded register still in flight is never live across the back edge (hipcc believes the value is there and may copy it).
    constexpr int VSPAN = ACE_WS_VSPAN * KSW / 12 > 0 ? ACE_WS_VSPAN * KSW / 12 : 1;   // interleaved epilogue: the eight values over the first VSPAN k-steps, the stores right after
    for (int pt = 0; pt < np; ++pt) {
        static_for<0, NSTG>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int u = pt * NSTG + q;
            if constexpr (MODE == 4) MT(16 * pt + 8 * q);            // (measurement builds: tools/trace_ws.py)
            if constexpr (!INTER && !(q == 0 && !HOLD)) {
#pragma unroll
                for (int k = 0; k < PW; ++k) piece(u + 1, k);
            }
            TileCtx ctx = {0, WS_OOBV, WS_OOBV, WS_OOBV, 0, 0, 0};
            if constexpr (q == 0) {
                read_partner(pt > 0 ? pt - 1 : 0);
                if constexpr (INTER) {
                    ctx = tile_ctx(pt > 0 ? pt - 1 : 0, pt > 0);
                } else {
                    if constexpr (!(ACE_WS_ABL & 1)) epilogue(pt > 0 ? pt - 1 : 0, pt > 0, eo);
                    if constexpr (MODE == 4) MT(16 * pt + 1);
                    if constexpr (!HOLD) {   // no registers to hold the store data through the stage: retire the stores first
                        if constexpr (!(ACE_WS_ABL & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if constexpr (MODE == 4) MT(16 * pt + 2);
                        hold(eo);
#pragma unroll
                        for (int k = 0; k < PW; ++k) piece(u + 1, k);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { v[r] = 0.f; if (ACC2) v2[r] = 0.f; }
            }
            // the residual of the tile being computed: requested in its LAST stage - or, when it comes as planes (two registers sets that
            // the epilogue above has just released) and the tile has two stages, already in the first: requested a stage before the
            // wait that retires it, the 1.2 k cycles that wait took (tools/trace_ws.py) were the loads' latency
            constexpr int RQ = (RPL && NSTG == 2 && ACE_WS_REARLY) ? 0 : NSTG - 1;
            if constexpr (RES && q == RQ && !(ACE_WS_ABL & 4)) load_residual(pt);
            if constexpr (MODE == 4) MT(16 * pt + 8 * q + 3);
            const unsigned sl = (unsigned)(size_t)(lds_cptr)(smem + (u & 1) * SLOT) + h * (KSW * 2048) + lane * 16;
            pipelined_steps<KSW, FDEPTH>(sl, [&](auto ss, const Frag& f) {
                constexpr int st = decltype(ss)::value;
                constexpr int j = q * KSW + st;
                constexpr bool FINE = INTER && ACE_WS_FINE && GELU && !F32 && !(STATS && !RSTATS);
                if constexpr (FINE) {
                    // 33 epilogue chunks (8 values x 4 quarters, then the stores) over the 3 KSW MFMA slots of the stage, a
                    // scheduling barrier per slot: MFMA, ~7 VALU, MFMA, ~7 VALU, ...
                    constexpr int NSL = 3 * KSW;
                    auto slot = [&](auto sc) {
                        constexpr int sl_ = decltype(sc)::value;
                        static_for<(sl_ * 33) / NSL, ((sl_ + 1) * 33) / NSL>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            if constexpr (k < 32) epi_chunk(std::integral_constant<int, k / 4>{}, std::integral_constant<int, k % 4>{}, ctx, eo);
                            else epi_item(std::integral_constant<int, 8>{}, ctx, eo);
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], f.h, v, 0, 0, 0);
                    slot(std::integral_constant<int, 3 * st>{});
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.l, v, 0, 0, 0);
                    slot(std::integral_constant<int, 3 * st + 1>{});
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.h, v, 0, 0, 0);
                    if constexpr (st < PW) piece(u + 1, st);   // the pieces of the next stage, one per k-step from the first on
                    slot(std::integral_constant<int, 3 * st + 2>{});
                } else if constexpr (!ACC2) {
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], f.h, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.l, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.h, v, 0, 0, 0);
                } else if constexpr ((st & 1) == 0) {   // a dependent MFMA on ONE accumulator issues every 44 cycles, not 32
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], f.h, v, 0, 0, 0);
                    v2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.l, v2, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.h, v, 0, 0, 0);
                } else {
                    v2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], f.h, v2, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.l, v, 0, 0, 0);
                    v2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], f.h, v2, 0, 0, 0);
                }
                if constexpr (INTER && !(ACE_WS_FINE && GELU && !F32 && !(STATS && !RSTATS))) {
                    if constexpr (st < PW) piece(u + 1, st);   // the pieces of the next stage, one per k-step from the first on
                    static_for<0, 8>([&](auto kc) {
                        constexpr int e = decltype(kc)::value;
                        if constexpr ((e * VSPAN) / 8 == st) epi_item(kc, ctx, eo);
                    });
                    if constexpr (st == VSPAN) epi_item(std::integral_constant<int, 8>{}, ctx, eo);
                }
            });
            if constexpr (q == NSTG - 1) {   // tile complete: keep the rows this wave finishes, hand the others to the partner
                if constexpr (ACC2) v += v2;
                rows_to_kgroups(v);
                f32x4 sa, sb;
#pragma unroll
                for (int e = 0; e < 4; ++e) { sa[e] = v[8 * (1 - H) + e]; sb[e] = v[8 * (1 - H) + 4 + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) own[e] = v[8 * H + e];
                char* xw = xch + (((pt & 1) * 8) + wave) * 2048 + lane * 16;
                *reinterpret_cast<f32x4*>(xw) = sa;
                *reinterpret_cast<f32x4*>(xw + 1024) = sb;
            }
            if constexpr (MODE == 4) MT(16 * pt + 8 * q + 4);
            if constexpr (!(ACE_WS_ABL & 8)) stage_top();
            if constexpr (MODE == 4) MT(16 * pt + 8 * q + 5);
            if constexpr (INTER && RES) {
#pragma unroll
                for (int e = 0; e < 8; ++e) res[e] = resn[e];
            }
            if constexpr (q == 0 && (HOLD || INTER)) hold(eo);
        });
    }
    // ---- last tile (its own set of store-data registers, held until every store of the segment has retired)
    EpiOut eo_last;
    f32x4 st8[8];
    read_partner(np - 1);
    epilogue(np - 1, true, eo_last);
    if constexpr (RSTATS) {   // statistics of the whole pixel range: reduce over the 32 pixel lanes (same g), one store per row
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int off = 1; off <= 16; off <<= 1) {
                rsm[e] += __shfl_xor(rsm[e], off, 64);
                rsq[e] += __shfl_xor(rsq[e], off, 64);
                if constexpr (WS_MINMAX) {
                    rmn[e] = fminf(rmn[e], __shfl_xor(rmn[e], off, 64));
                    rmx[e] = fmaxf(rmx[e], __shfl_xor(rmx[e], off, 64));
                }
            }
        // lanes 0 and 32 hold rows row0 + 8 g + e; part[(sample, slot) x M + row], slot = part_q of this workgroup
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc(p.part + ((long)smp * p.nstrips32 + part_q) * p.M, 0, p.M * 16, 0x00020000);
        const int gq = lane >> 5;
        const int vo = (lane & 31) == 0 ? (row0 + 8 * gq) * 16 : WS_OOBV;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            st8[e] = f32x4{rsm[e], rsq[e], rmn[e], rmx[e]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, st8[e]), rsG, vo, e * 16, 0);
        }
    }
    f32x4 lst = {0.f, 0.f, 0.f, 0.f};
    if constexpr (STATS && !RSTATS) {   // the segment's row statistics: one partial per row, slot part_q of this workgroup
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc(p.part + ((long)smp * p.nstrips32 + part_q) * p.M, 0, p.M * 16, 0x00020000);
        lst = Lacc[lane & 15];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lst), rsG, lane < 16 ? (row0 + lane) * 16 : WS_OOBV, 0, 0);
    }
    // the next segment loads weights into registers (and the range reduction below shuffles): every store of this segment
    // has read its data by then
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (STATS && !RSTATS) asm volatile("" ::"v"(lst));
    if constexpr (RSTATS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(st8[e]));
    }
    hold(eo_last);
    };
    if constexpr (NSTG == 1) {
#pragma unroll 1
        for (int sgi = 0; sgi < nseg; ++sgi) run_segment(segment(sgi), sgi == 0);
    } else {   // K >= 512: no registers for a segment loop (ws_plan gives these launches no extra workgroups)
        run_segment(segment(0), true);
    }
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
__global__ __launch_bounds__(512) void conv_ws_kernel(ConvStripArgs p, WsPlan pl) {
    __shared__ __attribute__((aligned(16))) char smem[WsGeom<KSW, NSTG, MODE>::LDS];
    if (threadIdx.x < 256) conv_ws_body<KSW, NSTG, MODE, 0>(p, smem, pl);   // waves 0 - 3: first half of the contraction
    else conv_ws_body<KSW, NSTG, MODE, 1>(p, smem, pl);
}

template <int KSW, int NSTG>
hipError_t launch_ws_k(const ConvStripArgs& a, int mode, hipStream_t s) {
    const WsPlan pl = ws_plan(a.M, a.HW, NSTG == 1);
    dim3 grid((unsigned)(8 * ws_workgroups_per_xcd(pl) * a.nbatch)), block(512);
    if constexpr (NSTG == 1) {   // the GELU modes exist for the single-stage contractions only (K <= 384)
        if (mode == 0) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 0>), grid, block, 0, s, a, pl);
        if (mode == 1) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 1>), grid, block, 0, s, a, pl);
    } else if (mode <= 1) {
        return hipErrorInvalidValue;
    }
    if (mode == 2) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 2>), grid, block, 0, s, a, pl);
    if (mode == 8) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 8>), grid, block, 0, s, a, pl);
    if (mode == 3) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 3>), grid, block, 0, s, a, pl);
    if (mode == 4) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 4>), grid, block, 0, s, a, pl);
    if (mode == 5) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 5>), grid, block, 0, s, a, pl);
    if (mode == 6) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 6>), grid, block, 0, s, a, pl);
#ifdef ACE_MEASUREMENT_SWITCHES
    if (mode == 7) hipLaunchKernelGGL((conv_ws_kernel<KSW, NSTG, 7>), grid, block, 0, s, a, pl);
#endif
    return hipGetLastError();
}

}  // namespace

// A-fragment packing of a conv weight (pack_frag.h), one (T, J) block or one folded-bias tile per workgroup
__global__ __launch_bounds__(256) void pack_conv_frag_kernel(PackFragArgs q) {
    __shared__ float red[4];
    pack_frag_block(q, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, red);
}

bool pack_frag_args_ok(const PackFragArgs& q) {
    if (q.O % 32 != 0 || q.I % 16 != 0 || (q.order == 1 && q.I % 32 != 0) || (q.a && !q.wslot) || (q.bf && !q.b)) return false;
    if (q.bf && ((q.ldw & 3) != 0 || (reinterpret_cast<uintptr_t>(q.W) & 15) != 0 || (reinterpret_cast<uintptr_t>(q.b) & 15) != 0)) return false;
    return q.W && q.dst && q.nsamples >= 1;
}

hipError_t launch_pack_conv_frag(const float* W, long ldw, int O, int I, int order, const float* a, float wmax,
                                 float scale_static, unsigned* wslot, void* dst, long sDst, int nsamples, hipStream_t s,
                                 const float* b, const float* bias, float* bf) {
    PackFragArgs q;
    q.W = W; q.ldw = ldw; q.O = O; q.I = I; q.order = order; q.a = a; q.wmax = wmax; q.scale_static = scale_static; q.wslot = wslot;
    q.dst = static_cast<_Float16*>(dst); q.sDst = sDst; q.b = b; q.bias = bias; q.bf = bf; q.nsamples = nsamples;
    if (!pack_frag_args_ok(q)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_conv_frag_kernel, dim3((unsigned)pack_frag_blocks(q), (unsigned)nsamples), dim3(256), 0, s, q);
    return hipGetLastError();
}

#ifdef ACE_X_TRACE   // measurement builds only (tools/trace_ws.py): this file's copy of the s_memtime stamps
extern "C" int ace_debug_trace_ws(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mlp_trace), sizeof(mlp_trace)); }
#endif

// statistics partials per row the launch described by `a` writes (a.nstrips32 must be at least this)
int conv_ws_stat_parts(const ConvStripArgs& a) {
    // one partial per workgroup of a channel slice (inner skip: accumulated in registers; fc2 modes: in LDS)
    const WsPlan pl = ws_plan(a.M, a.HW, a.C <= 384);
    return ws_stat_slots(pl);
}

// K: input channels (contraction), M: output channels; role 0 inner skip, 1 fc1, 2 fc2 (-1: any)
bool conv_ws_eligible(int K, int M, long HW, int role, int roles_on) {
    if (role >= 0 && role < 3 && !(roles_on >> role & 1)) return false;
    if (role == -1 && roles_on == 0) return false;
    if (!(K == 128 || K == 256 || K == 384 || K == 512 || K == 768)) return false;
    if (role >= 0 && role <= 1 && K > 384) return false;
    return M % 128 == 0 && M >= 128 && M <= 2048 && (long)M * HW * 4 < 0x7fffff00L && ((HW + 31) / 32 + 8) * (long)M * 16 < 0x7fffff00L;
}

hipError_t launch_conv_ws(const ConvStripArgs& a, hipStream_t s) {
    if (!conv_ws_eligible(a.C, a.M, a.HW, -2) || !a.bias || !a.xslot || !a.A) return hipErrorInvalidValue;
    const bool f32 = a.Cf != nullptr, pk = a.Chi != nullptr, stats = a.part != nullptr, res = a.R != nullptr;
    const bool rpl = a.Rhi != nullptr && a.Rlo != nullptr && a.rslot != nullptr && !a.R;
    const bool gelu = a.act == ACT_GELU || a.act == ACT_GELU_FAST;
    int mode = -1;
    if (gelu && res && pk && stats && !f32) mode = 0;
    else if (gelu && res && f32 && !pk && !stats) mode = 6;
    else if (gelu && !res && pk && !stats && !f32) mode = 1;
    else if (a.act == ACT_NONE && res && f32 && pk && stats) mode = 2;
    else if (a.act == ACT_NONE && res && !f32 && pk && stats) mode = 8;
    else if (a.act == ACT_NONE && res && f32 && !pk && !stats) mode = 3;
    else if (a.act == ACT_NONE && rpl && !f32 && pk && stats) mode = 4;
    else if (a.act == ACT_NONE && rpl && f32 && !pk && !stats) mode = 5;
#ifdef ACE_MEASUREMENT_SWITCHES
    else if (a.act == ACT_NONE && !res && !rpl && f32 && !pk && !stats) mode = 7;
#endif
    if (mode < 0 || (pk && (!a.Clo || !a.cslot)) || (f32 && a.M > 1024)) return hipErrorInvalidValue;
    if (mode <= 1 && a.C > 384) return hipErrorInvalidValue;
    switch (a.C) {
        case 128: return launch_ws_k<4, 1>(a, mode, s);
        case 256: return launch_ws_k<8, 1>(a, mode, s);
        case 384: return launch_ws_k<12, 1>(a, mode, s);
        case 512: return launch_ws_k<8, 2>(a, mode, s);
        case 768: return launch_ws_k<12, 2>(a, mode, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ace
