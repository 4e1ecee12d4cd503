// Spectral filter contraction (fme/ace/models/modulus/contractions.py:183-195, "dhconv") for gfx950 with the weights
// streamed ONCE, straight from HBM into MFMA B fragments:
//     E_l[(m, b)][o] = sum_i D_l[(m, b)][i] W_l[i][o]      complex, one (C x C) filter matrix per degree l, rows m <= l
// The filter is 212 MB at the ACE2 shape and every element is used by at most 181 rows: the stage is bound by streaming
// the weights.  The 128 x 128 tile engine moves each weight through L2 -> LDS once per 128-row tile and each coefficient
// once per 128-column tile (1.1 GB of LDS-DMA traffic per launch, 465 MB of HBM reads for 262 MB of operands: r02 PMC).
// Here a workgroup owns (degree l, 128 output channels): each of its four waves holds 32 output channels, real AND imaginary
// part, for ALL rows of the degree (up to 192 = 6 strips of 32, 12 accumulator tiles per wave; degrees with more rows - batch
// > 1, the 0.25-degree grid - are cut into 192-row chunks, one workgroup each):
//   * the weights are the MFMA B operand: their stored form (k-packed "P format", compact complex: Wr | Wi per l) IS the
//     fragment of a lane, so they go global -> registers with coalesced 16-byte loads, two stages ahead, and never touch LDS;
//     a stage is 32 rows of Wr and the same 32 rows of Wi, each used for two products (with D_re and with D_im), so every
//     filter element is fetched exactly once per launch;
//   * the coefficients D_l (fp16 hi/lo planes written by the Legendre stage, row-major (D_re | D_im)) are the A operand
//     shared by the four waves: a stage holds the 32-wide k slice of BOTH halves of the row, as 1-KiB LDS-DMA pieces of
//     16 rows x 32 k with a source-side XOR swizzle, three-stage ring, counted waits (all vector-memory operations of the
//     loop are inline asm, no stores: the count is exact);
//   * complex structure: real += D_re Wr - D_im Wi, imaginary += D_re Wi + D_im Wr; the minus sign is put on the D_im
//     fragment (8 v_xor per strip and k16 step);
//   * row strips beyond l are skipped (1 .. 6 active strips of 32 rows).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "strip_common.h"

namespace ace {
namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

MDEV void gload16(half8& dst, const _Float16* p) {   // un-waited 16-byte global load (retired by wait_b below)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
MDEV half8 neg8(half8 v) {
    u32x4v u = __builtin_bit_cast(u32x4v, v);
    u ^= 0x80008000u;
    return __builtin_bit_cast(half8, u);
}

// NS: active 32-row strips (1 .. 6) of this workgroup's row chunk [row0, row0 + rows) of degree l
template <int NS>
MDEV void dhconv_body(const DhconvStripArgs& p, char* smem, const int l, const int j, const int row0, const int rows) {
    constexpr int NSTG = 3;                       // ring depth (stages)
    constexpr int STAGE = NS * 8192;              // bytes: NS strips x 2 slices (re, im) x (2 row pieces x 2 planes) x 1 KiB
    constexpr int NA = 2 * NS;                    // A pieces per wave per stage
    constexpr int PB = 2;                         // B fragments run PB stages ahead (as the A pieces)
    const int tid = threadIdx.x;
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
    auto issue_a = [&](int t) {   // stage t -> ring slot t % NSTG; stages past the end re-fetch the last (uniform count)
        const int tt = tb + (t < nstages ? t : nstages - 1);
        char* dst = smem + (t % NSTG) * STAGE;
#pragma unroll
        for (int k = 0; k < NA; ++k) glds16(asrc[k] + 32 * tt, dst + (wave + 4 * k) * 1024);
    };
    // ---- B fragments of stage t: rows [32 t, 32 t + 32) of Wr and of Wi, this wave's 32 columns.  Each filter element is
    //      loaded by exactly one wave of one workgroup, once (the first version walked k over (D_re | D_im) with the real
    //      columns in waves 0 - 1 and the imaginary ones in 2 - 3: Wr and Wi were each fetched twice, half a kernel apart -
    //      567 MB of HBM reads for 262 MB of operands, r02 PMC)
    const _Float16* Wh = p.Whi + (long)l * p.sW;
    const _Float16* Wl = p.Wlo + (long)l * p.sW;
    struct BSet { half8 h[2][2], l[2][2]; };      // [k16 step][Wr | Wi]
    auto issue_b = [&](BSet& b, int t) {
        const int tt = tb + (t < nstages ? t : nstages - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                int kg = 4 * tt + 2 * c + g - kbase / 8;          // k-group relative to this wave's first stored row
                if (native) kg = kg < 0 ? 0 : (kg >= kstore / 8 ? kstore / 8 - 1 : kg);
                const long off = (long)blk * kstore * C + ((long)kg * C + oc + i) * 8;
                gload16(b.h[c][blk], Wh + off);
                gload16(b.l[c][blk], Wl + off);
            }
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

    // prologue in the steady-state issue order (stage t issues A(t + 2) then B(t + 2)): A0 B0 | A1 B1
    BSet bs[4];                                   // ring of fragment sets, indexed with compile-time constants only
    issue_a(0); issue_b(bs[0], 0);
    issue_a(1); issue_b(bs[1], 1);

    const float inv_a = ldexpf(1.0f, -pow2_exponent_for(wave_max_bits(raw_a)));
    const float oscale = inv_a / p.bscale;
    const int key = (i >> 2) & 3;                 // swizzle key of this lane's fragment rows (rows i and i + 16 ... share it mod 4)

    // Stage t issues A(t + 2) and B(t + 2).  Issue order: ... A(t) B(t) | A(t+1) B(t+1) | A(t+2) B(t+2): after this stage's
    // issue exactly 2 (NA + 8) operations are newer than B(t).  All of them are loads (in-order retirement), the loop has
    // no stores: vmcnt(2 (NA + 8)) is exact.
    auto stage = [&](const int t, BSet& b, BSet& bnew) {
        issue_a(t + 2);
        issue_b(bnew, t + PB);
        wait_b(b, std::integral_constant<int, 2 * (NA + 8)>{});
        __builtin_amdgcn_s_barrier();             // every wave's pieces of stage t landed
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
            });
        });
        __builtin_amdgcn_s_barrier();             // every wave is done reading the slot of stage t (refilled by A(t + 3))
    };
    for (int t0 = 0; t0 < nstages; t0 += 4)       // nstages (C / 32, or the group span / 32) is a multiple of 4
        static_for<0, 4>([&](auto u) { stage(t0 + u, bs[u], bs[(u + PB) % 4]); });
    // The dummy tail fetches are still in flight into fragment sets that are dead as far as hipcc knows: the wait that retires
    // them NAMES their registers, or the allocator may hand those registers to the epilogue's values above the wait and the
    // returning loads would land on them (tools/store_hazard.hip, variant 3: the mechanism behind round 2's "store data"
    // corruption).  Two statements: an asm takes at most 30 operands.
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

__global__ __launch_bounds__(256, 1) void dhconv_strip_kernel(DhconvStripArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 6 * 8192];
    // Workgroup b runs on XCD b % 8 (each XCD has its own L2).  The C / 128 column groups of one degree all read the degree's
    // coefficient rows D[l] (the A operand: up to 181 rows x 768 values as hi / lo planes): deal whole DEGREES to XCDs so that the
    // groups of a degree follow each other on ONE XCD and the later ones find D[l] in its L2 (round 3 dealt consecutive blocks -
    // the groups of a degree - to three different XCDs: 176 MB of D traffic for 100 MB of D, profiles/r04_pmc_table.txt).
    // Heavy degrees first within every XCD: XCD x takes l = L - 1 - x, L - 9 - x, ...
    const int ncg = p.C / 128;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int l = p.L - 1 - ((idx / ncg) * 8 + xcd);
    const int j = idx % ncg;
    if (l < 0) return;
    const int rows_l = (l + 1) * p.trimul < p.Mrows ? (l + 1) * p.trimul : p.Mrows;
    // degrees with more than 192 rows (a batch of 2 at the 1-degree grid: 362; the 0.25-degree grid: 721) are cut into chunks
    // of 192 rows, one workgroup each (blockIdx.y); the filter slice of a (degree, column group) is then streamed once per
    // chunk - the later ones from L2 / MALL (393 KB per slice)
    const int row0 = (int)blockIdx.y * 192;
    if (row0 >= rows_l) return;
    const int rows = rows_l - row0 < 192 ? rows_l - row0 : 192;
    // active 32-row strips: 1 .. 6 (row granularity 32: 14 % fewer MFMAs than strip pairs at the 1-degree shape, and the
    // workgroups' sizes - there are only 2.1 per CU - balance better)
    if (rows <= 32) dhconv_body<1>(p, smem, l, j, row0, rows);
    else if (rows <= 64) dhconv_body<2>(p, smem, l, j, row0, rows);
    else if (rows <= 96) dhconv_body<3>(p, smem, l, j, row0, rows);
    else if (rows <= 128) dhconv_body<4>(p, smem, l, j, row0, rows);
    else if (rows <= 160) dhconv_body<5>(p, smem, l, j, row0, rows);
    else dhconv_body<6>(p, smem, l, j, row0, rows);
}

}  // namespace

bool dhconv_native_groups_ok(int C, int groups) {
    if (groups <= 1 || C % groups != 0 || C % 128 != 0) return false;
    const int cg = C / groups;
    return cg % 32 == 0 && (cg % 128 == 0 || 128 % cg == 0);
}

bool dhconv_strip_eligible(const DhconvStripArgs& a) {
    if (a.kstore > 0 && a.kstore != a.C && !(a.groups > 1 && a.kstore == a.C / a.groups && dhconv_native_groups_ok(a.C, a.groups))) return false;
    return a.C % 128 == 0 && a.C >= 128 && a.groups >= 1 && a.C % a.groups == 0 && a.Mrows >= 1 && (a.Mrows + 191) / 192 <= 65535 && a.L >= 1 && a.Dhi && a.Dlo && a.Whi && a.Wlo && a.E && a.amax;
}

hipError_t launch_dhconv_strip(const DhconvStripArgs& a, hipStream_t s) {
    if (!dhconv_strip_eligible(a)) return hipErrorInvalidValue;
    dim3 grid((unsigned)(((a.L + 7) / 8) * 8 * (a.C / 128)), (unsigned)((a.Mrows + 191) / 192)), block(256);   // whole degrees per XCD (see the kernel)
    hipLaunchKernelGGL(dhconv_strip_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace ace
