// Equatorially folded "strip" Legendre kernels for gfx950 (fme/sht_fix.py:134-138 forward, :208-219 inverse).
//
// strip.hip contracts over all nlat latitudes.  On a grid that is symmetric about the equator the associated Legendre functions
// are even or odd in latitude, P_l^m(-x) = (-1)^(l+m) P_l^m(x), and the contraction halves (strip_pack.h, "folded form"):
//     forward   c[l] = sum_{kf < Hh} wt[l][kf] (X[kf] +- X[H-1-kf])          + for (l - m) even, - for odd
//     inverse   X[kf] = E[kf] + O[kf],   X[H-1-kf] = E[kf] - O[kf]           E / O: the even / odd degrees' partial sums
// Half the MFMAs, half the table bytes (23.5 -> 8.6 MB of fragments at 1 degree) and half the LDS per table tile for the same
// data traffic.  The fold itself costs one add and one subtract per loaded pair (forward, in the strip load) or per output pair
// (inverse, in the epilogue); the longitude FFT kernels and the layout of X are untouched.
//
// Structure as strip.hip: a wave owns a 32-column strip of the data operand for the whole contraction - here as TWO resident
// operands, even and odd (forward: sums and differences of mirror rows; inverse: the coefficient rows l = m, m + 2, ... and
// l = m + 1, m + 3, ...), each nkp2 <= 6 k16-steps of fp16 hi / lo B fragments; the table streams as 1-KiB LDS-DMA pieces into a
// three-slot ring, one slot = one UNIT = the 32-row tile tp of one parity (nkp2 x 2 KiB); units alternate even, odd, so the
// loop is unrolled by two and each half names its operand at compile time.  Forward: every unit is an output tile (rows
// l = m + p + 2 r - the two units of a pair interleave into 64 consecutive degrees).  Inverse: a pair of units gives the rows
// kf (E + O) and their mirror rows (E - O).  Epilogues are deferred by one unit and issued under the next unit's MFMAs; the
// vmcnt accounting is strip.hip's (a unit's pieces are waited for with the count of the pieces of the unit after it).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "strip_pack.h"

namespace ace {
namespace {

#define FDEV __device__ __forceinline__

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char* lds_cptr;

constexpr int NH_MAX = FOLD_NKP_SMALL / 2;         // small form: nkp2 <= 6: Hh <= 96 (forward), ceil(lmax / 2) <= 96 (inverse)
constexpr int NH_BIG = FOLD_NKP_BIG / 2;           // big form: nkp2 = 12, 18, 24 (one workgroup per CU, 512 registers per lane)
constexpr int FNSLOT = 3;                          // unit u + 2 is in flight while unit u is consumed
constexpr int FTS_BYTES = 2048;                    // per-wave transpose buffer: 16 rows x 32 dwords
constexpr int fold_lds_bytes(int nh) { return FNSLOT * 2 * nh * 2048 + 4 * FTS_BYTES; }   // NH 3: 44 KiB; NH 12: 152 KiB
constexpr int FLDS_BYTES = fold_lds_bytes(NH_MAX);

FDEV unsigned slot_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
FDEV int pow2_exponent_for(float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 12 - e; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
FDEV float wave_max_bits(unsigned raw) {
    float mx = __uint_as_float(raw);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
}
// one 1-KiB LDS-DMA piece: lane L's 16 bytes at gsrc land at lds_dst + 16 L (m0 clobbered, see strip.hip)
FDEV void glds16(const void* gsrc, const char* lds_dst_uniform) {
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)lds_dst_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(addr)
                 : "memory", "m0");
}
FDEV int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// where the 32 rows of an output tile go: row(rl) = rbase + rstep * rl, stored when vlo <= row < vhi
struct RowMap { int rbase, rstep, vlo, vhi; };

// OUT: 0 = fp32, 16-byte stores (N % 4 == 0, aligned); 1 = fp16 hi/lo planes, 8-byte stores; 2 = fp32 scalar stores
template <int NH, int OUT, bool FULLN, int MODE>
FDEV void fold_body(const LegStripArgs& p, char* smem, const int m, const int grp, const FoldGeom gm) {
    constexpr int NKP = 2 * NH;       // k16-steps per unit; a unit = 4 NH pieces, NH per wave
    constexpr int FSLOT_BYTES = NKP * 2048;   // one unit of A fragments (hi + lo)
    constexpr bool BIG = NH > NH_MAX; // 0.25-degree form: one wave per SIMD, both operands (2 x NKP x 8 registers) still resident
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int N = p.N, K = p.K, R = p.R;
    const int H = MODE == 0 ? K : R;
    const int n0 = grp * 128 + wave * 32;
    const int n = n0 + i;
    const int nc = n < N ? n : N - 1;
    char* ring = smem;
    char* Ts = smem + FNSLOT * FSLOT_BYTES + wave * FTS_BYTES;
    const int nunits = 2 * gm.npairs;

    const unsigned raw_b = slot_load(p.bmax + lane);

    // ---- table stream
    const _Float16* Am = p.A + (long)p.tile_off[m] * 1024;
    auto issue_unit = [&](int u) {   // unit u -> slot u % 3; units past the end re-fetch the last one (uniform count)
        const int uu = u < nunits ? u : nunits - 1;
        const _Float16* src = Am + (long)uu * NKP * 1024 + lane * 8;
        const char* dst = ring + (u % FNSLOT) * FSLOT_BYTES;
#pragma unroll
        for (int c = 0; c < NH; ++c) {
            const int pc = wave + 4 * c;
            glds16(src + pc * 512, dst + pc * 1024);
        }
    };
    if (nunits > 0) { issue_unit(0); issue_unit(1); }

    // ---- resident operands: fp32 rows -> fp16 hi/lo B fragments; lane (i, g) holds k = 16 jj + 8 g .. + 7 of column i
    const long ks = p.b_kstride;
    float bbound = wave_max_bits(raw_b);
    if (MODE == 0) bbound *= 2.f;                     // |X[kf] +- X[H-1-kf]| <= 2 max|X|
    const int eb = pow2_exponent_for(bbound);
    const float bscale = ldexpf(1.0f, eb), inv_b = ldexpf(1.0f, -eb);
    half8 bh[2][NKP], bl[2][NKP];
#define ACE_FOLD_SPLIT(x, H_, L_) do { const float x_ = (x); const _Float16 h_ = (_Float16)x_; (H_) = h_; (L_) = (_Float16)(x_ - (float)h_); } while (0)
    if constexpr (MODE == 0) {
        // Direct rows kf through a descriptor over the first Hh rows of this wavenumber, mirror rows H - 1 - kf through one over
        // the remaining H - Hh rows (based at row Hh): a k index at or beyond Hh, the mirror of the middle row of an odd H, and
        // every padded k-step fall outside their descriptor and read as exact zeros - no masks.  32-bit offsets; the mirror
        // offset of such an index is negative and wraps to the top of the unsigned range (far beyond num_records).
        const float* Xm = p.B + (long)m * p.b_moff;
        const long rowb = ks * 4;
        const auto rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xm), 0, (unsigned)((long)gm.Hh * rowb), 0x00020000);
        const auto rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xm + (long)gm.Hh * ks), 0, (unsigned)((long)(H - gm.Hh) * rowb), 0x00020000);
        unsigned vd[8], vm[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            vd[e] = (unsigned)(((long)(8 * g + e) * ks + nc) * 4);
            vm[e] = (unsigned)(((long)(H - 1 - gm.Hh - (8 * g + e)) * ks + nc) * 4);   // row (H - 1 - kf) - Hh of the mirror descriptor
        }
        // small form: every row of the strip in flight at once; big form: in chunks of CHK k-steps (the raw fp32 values of a whole
        // 0.25-degree strip would not fit beside the fragments they become)
        constexpr int CHK = BIG ? 4 : NKP;
#pragma unroll
        for (int j0 = 0; j0 < NKP; j0 += CHK) {
            float ra[CHK][8], rb[CHK][8];
#pragma unroll
            for (int jc = 0; jc < CHK; ++jc) {
                const unsigned so = (unsigned)((long)(16 * (j0 + jc)) * rowb);     // wave-uniform; in the VECTOR offset (range-checked)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ra[jc][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsD, vd[e] + so, 0, 0));
                    rb[jc][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsM, vm[e] - so, 0, 0));
                }
            }
#pragma unroll
            for (int jc = 0; jc < CHK; ++jc)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = ra[jc][e] * bscale, b = rb[jc][e] * bscale;
                    ACE_FOLD_SPLIT(a + b, bh[0][j0 + jc][e], bl[0][j0 + jc][e]);
                    ACE_FOLD_SPLIT(a - b, bh[1][j0 + jc][e], bl[1][j0 + jc][e]);
                }
            if (BIG) __builtin_amdgcn_sched_barrier(0);   // keep the chunks in order: hoisting every load to the top is what overflows
        }
    } else {
        // coefficient rows l = m + p + 2 (16 jj + 8 g + e) of this wavenumber; rows at or beyond lmax are outside the descriptor
        const float* Em = p.B + (long)m * p.b_moff;
        const long rowb = ks * 4;
        const long span = ((long)(K - 1) * ks + N) * 4;     // up to the end of row lmax - 1 of this wavenumber
        const auto rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Em), 0, (unsigned)(span > 0 ? span : 0), 0x00020000);
        unsigned ve[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ve[e] = (unsigned)(((long)(m + 2 * (8 * g + e)) * ks + nc) * 4);
        constexpr int CHK = BIG ? 2 : NKP;
#pragma unroll
        for (int j0 = 0; j0 < NKP; j0 += CHK) {
            float r0[CHK][8], r1[CHK][8];
#pragma unroll
            for (int jc = 0; jc < CHK; ++jc) {
                const unsigned so = (unsigned)((long)(32 * (j0 + jc)) * rowb);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    r0[jc][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsE, ve[e] + so, 0, 0));
                    r1[jc][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsE, ve[e] + so + (unsigned)rowb, 0, 0));
                }
            }
#pragma unroll
            for (int jc = 0; jc < CHK; ++jc)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ACE_FOLD_SPLIT(r0[jc][e] * bscale, bh[0][j0 + jc][e], bl[0][j0 + jc][e]);
                    ACE_FOLD_SPLIT(r1[jc][e] * bscale, bh[1][j0 + jc][e], bl[1][j0 + jc][e]);
                }
            if (BIG) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- scales
    const float inv_a = 1.0f / p.ascale;
    float oscale = inv_a * inv_b;
    if (OUT == 1) {   // planes: bound of this launch's output, identical in every workgroup; the consumer reads it from cslot
        const float cbound = p.cw * wave_max_bits(raw_b);
        oscale *= ldexpf(1.0f, pow2_exponent_for(cbound));
        if (tid == 0) atomicMax(p.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
    }

    float vmax = 0.f;
    const long cm = (long)m * p.c_moff;
    auto tile_max = [&](const RowMap rm, const f32x16& acc, auto inner_tag) {   // range of the fp32 outputs (valid entries only)
        constexpr bool NOMASK = decltype(inner_tag)::value && FULLN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rm.rbase + rm.rstep * acc_row(r, g);
            const bool ok = NOMASK || (row >= rm.vlo && row < rm.vhi && n < N);
            vmax = fmaxf(vmax, ok ? fabsf(acc[r] * oscale) : 0.f);
        }
    };
    auto store_tile = [&](const RowMap rm, const f32x16& acc, auto inner_tag) {
        constexpr bool NOMASK = decltype(inner_tag)::value && FULLN;
        if (OUT == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rm.rbase + rm.rstep * acc_row(r, g);
                const float v = acc[r] * oscale;
                if (row >= rm.vlo && row < rm.vhi && n < N) p.C[cm + (long)row * p.c_rstride + n] = v;
            }
            return;
        }
        // row-major through this wave's transpose buffer, half a tile (16 rows = 8 accumulator registers) at a time: dword
        // (row, column) at row * 32 + column, read back as four consecutive columns of one row per lane
        unsigned* T32 = reinterpret_cast<unsigned*>(Ts);
        const int c4 = (lane & 7) * 4;
        const int ncol = n0 + c4;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = 8 * hf + r8;
                const float v = acc[r] * oscale;
                unsigned w;
                if (OUT == 1) {
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    w = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
                } else {
                    w = __float_as_uint(v);
                }
                T32[(acc_row(r, g) - 16 * hf) * 32 + i] = w;
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int rl = ps * 8 + (lane >> 3);           // row within this half
                const int row = rm.rbase + rm.rstep * (16 * hf + rl);
                const u32x4 d = *reinterpret_cast<const u32x4*>(T32 + rl * 32 + c4);
                const bool ok = NOMASK || (row >= rm.vlo && row < rm.vhi && ncol < N);
                if (OUT == 1) {
                    u32x2 hi2, lo2;
                    hi2[0] = (d[0] & 0xffffu) | (d[1] << 16);
                    hi2[1] = (d[2] & 0xffffu) | (d[3] << 16);
                    lo2[0] = (d[0] >> 16) | (d[1] & 0xffff0000u);
                    lo2[1] = (d[2] >> 16) | (d[3] & 0xffff0000u);
                    if (ok) {
                        const long off = cm + (long)row * p.c_rstride + ncol;
                        *reinterpret_cast<u32x2*>(p.Chi + off) = hi2;
                        *reinterpret_cast<u32x2*>(p.Clo + off) = lo2;
                    }
                } else {
                    if (ok) {
                        f32x4 v;
                        v[0] = __uint_as_float(d[0]); v[1] = __uint_as_float(d[1]);
                        v[2] = __uint_as_float(d[2]); v[3] = __uint_as_float(d[3]);
                        *reinterpret_cast<f32x4*>(p.C + cm + (long)row * p.c_rstride + ncol) = v;
                    }
                }
            }
        }
    };
    // row maps.  Forward, unit (tp, par): degrees l = m + par + 2 (32 tp + rl), stored below lmax.  Inverse, pair tp: north rows
    // kf = 32 tp + rl below Hh; south rows H - 1 - kf, stored when they are not a north row (at or beyond Hh)
    auto fwd_map = [&](int tp, int par) { return RowMap{m + par + 64 * tp, 2, 0, R}; };
    auto north_map = [&](int tp) { return RowMap{32 * tp, 1, 0, gm.Hh}; };
    auto south_map = [&](int tp) { return RowMap{H - 1 - 32 * tp, -1, gm.Hh, H}; };

    // one unit: wait for its pieces, barrier, refill the slot of the unit before it, `under()` (the deferred epilogue), MFMAs
    auto unit = [&](int u, auto par_tag, auto&& under) -> f32x16 {
        constexpr int PAR = decltype(par_tag)::value;
        // queue at the top of unit u >= 1 (issue order): [pieces of u | stores] from unit u - 2, [pieces of u + 1 | stores] from
        // unit u - 1; loads retire in order, so "a piece of u pending" implies all NH pieces of u + 1 pending: vmcnt(NH)
        // guarantees unit u has landed (strip.hip)
        if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // units 0 and 1 and the data strips
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH) : "memory");
        __syncthreads();                                     // every share landed; every wave is done reading unit u - 1
        issue_unit(u + 2);                                   // ... whose slot is refilled now (a dummy past the end)
        under();
        const char* slot = ring + (u % FNSLOT) * FSLOT_BYTES + lane * 16;
        f32x16 a0, a1, a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }
        (void)a1;
        // table fragments: the whole unit up front (small form: 48 registers at most) or in groups of FG k-steps, the next group
        // requested before the MFMAs of the current one (big form)
        constexpr int FG = BIG ? (MODE == 1 ? 1 : 2) : NKP;   // (the big inverse form also holds the even parity's tile: one k-step of read-ahead)
        half8 fh[2][FG], fl[2][FG];
#pragma unroll
        for (int q = 0; q < FG; ++q) {
            fh[0][q] = *reinterpret_cast<const half8*>(slot + q * 2048);
            fl[0][q] = *reinterpret_cast<const half8*>(slot + q * 2048 + 1024);
        }
#pragma unroll
        for (int g0 = 0; g0 < NKP; g0 += FG) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = (g0 / FG) & 1;
            if (g0 + FG < NKP) {
#pragma unroll
                for (int q = 0; q < FG; ++q) {
                    fh[cur ^ 1][q] = *reinterpret_cast<const half8*>(slot + (g0 + FG + q) * 2048);
                    fl[cur ^ 1][q] = *reinterpret_cast<const half8*>(slot + (g0 + FG + q) * 2048 + 1024);
                }
            }
#pragma unroll
            for (int q = 0; q < FG; ++q) {
                if constexpr (BIG) {   // two accumulators (both small terms in one): 16 registers the big form needs; the chains still alternate
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[cur][q], bh[PAR][g0 + q], a0, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur][q], bh[PAR][g0 + q], a2, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur][q], bl[PAR][g0 + q], a0, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[cur][q], bh[PAR][g0 + q], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur][q], bl[PAR][g0 + q], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[cur][q], bh[PAR][g0 + q], a2, 0, 0, 0);
                }
            }
        }
        if constexpr (BIG) return a0 + a2;
        else return (a0 + a1) + a2;
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;

    f32x16 pa, pb;   // held outputs: forward pb = the odd unit of the pair before; inverse pa / pb = north / south rows of the pair before
#pragma unroll
    for (int r = 0; r < 16; ++r) { pa[r] = 0.f; pb[r] = 0.f; }
    if constexpr (MODE == 0) {
        for (int tp = 0; tp < gm.npairs; ++tp) {
            const f32x16 ev = unit(2 * tp, T0{}, [&] {
                if (tp > 0) {
                    if (OUT != 1) tile_max(fwd_map(tp - 1, 1), pb, std::true_type{});
                    store_tile(fwd_map(tp - 1, 1), pb, std::true_type{});
                }
            });
            const bool last = tp + 1 == gm.npairs;
            pb = unit(2 * tp + 1, T1{}, [&] {
                if (!last) {
                    if (OUT != 1) tile_max(fwd_map(tp, 0), ev, std::true_type{});
                    store_tile(fwd_map(tp, 0), ev, std::true_type{});
                }
            });
            if (last) pa = ev;
        }
    } else if constexpr (BIG) {
        // big form: no deferred epilogue (its two held tiles are 32 registers the 384-register operands leave no room for); the rows
        // of a pair are stored as soon as both parities are in - the stores themselves still retire under the next pair's MFMAs
        for (int tp = 0; tp < gm.npairs; ++tp) {
            const f32x16 ev = unit(2 * tp, T0{}, [] {});
            const f32x16 od = unit(2 * tp + 1, T1{}, [] {});
            f32x16 t = ev + od;
            const bool inner = tp + 1 < gm.npairs;
            if (inner) { tile_max(north_map(tp), t, std::true_type{}); store_tile(north_map(tp), t, std::true_type{}); }
            else { tile_max(north_map(tp), t, std::false_type{}); store_tile(north_map(tp), t, std::false_type{}); }
            __builtin_amdgcn_sched_barrier(0);   // one 16-register tile at a time through the transpose buffer
            t = ev - od;
            if (inner) { tile_max(south_map(tp), t, std::true_type{}); store_tile(south_map(tp), t, std::true_type{}); }
            else { tile_max(south_map(tp), t, std::false_type{}); store_tile(south_map(tp), t, std::false_type{}); }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int tp = 0; tp < gm.npairs; ++tp) {
            const f32x16 ev = unit(2 * tp, T0{}, [&] {
                if (tp > 0) {
                    tile_max(north_map(tp - 1), pa, std::true_type{});
                    store_tile(north_map(tp - 1), pa, std::true_type{});
                }
            });
            const f32x16 od = unit(2 * tp + 1, T1{}, [&] {
                if (tp > 0) {
                    tile_max(south_map(tp - 1), pb, std::true_type{});
                    store_tile(south_map(tp - 1), pb, std::true_type{});
                }
            });
            pa = ev + od;
            pb = ev - od;
        }
    }
    // the dummy pieces issued past the end land in the ring: retire them before its first slot doubles as the reduction scratch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // last pair: masked.  The range reduction (shuffles, an LDS round trip) runs before the last stores (strip.hip)
    constexpr bool STORED_IN_LOOP = BIG && MODE == 1;   // (the big inverse form has stored every pair already)
    const int tl = gm.npairs - 1;
    const RowMap ma = MODE == 0 ? fwd_map(tl, 0) : north_map(tl), mb = MODE == 0 ? fwd_map(tl, 1) : south_map(tl);
    if (OUT != 1 && gm.npairs > 0 && !STORED_IN_LOOP) {
        tile_max(ma, pa, std::false_type{});
        tile_max(mb, pb, std::false_type{});
    }
    if (OUT != 1 && p.omax) {   // one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        __syncthreads();        // ... before the transpose buffer of the last tiles re-uses the LDS
    }
    if (gm.npairs > 0 && !STORED_IN_LOOP) {
        store_tile(ma, pa, std::false_type{});
        store_tile(mb, pb, std::false_type{});
    }
}

#ifdef ACE_LF_TRACE   // measurement builds (tools/trace_lf.py): per workgroup, wave 0: start / end (s_memtime), wavenumber, HW_ID, XCC
__device__ unsigned long long lf_wg_span[2][4096][5];
#endif

template <int OUT, bool FULLN, int MODE>
__global__ __launch_bounds__(256, 2) void legendre_fold_kernel(LegStripArgs p, int G) {
    __shared__ __attribute__((aligned(16))) char smem[FLDS_BYTES];
#ifdef ACE_LF_TRACE
    const unsigned long long lf_t0 = __builtin_amdgcn_s_memtime();
#endif
    // workgroup b runs on XCD b % 8: deal the wavenumbers round-robin so that all groups of one m share an L2 and the
    // triangular load is even across XCDs; ascending m = heaviest first
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int m = (idx / G) * 8 + xcd;
    const int grp = idx % G;
    if (m >= p.nbatch) return;
    const FoldGeom gm = fold_geom(MODE, m, p.R, p.K);
    switch (gm.nkp2) {
        case 2: fold_body<1, OUT, FULLN, MODE>(p, smem, m, grp, gm); break;
        case 4: fold_body<2, OUT, FULLN, MODE>(p, smem, m, grp, gm); break;
        case 6: fold_body<3, OUT, FULLN, MODE>(p, smem, m, grp, gm); break;
        default: break;
    }
#ifdef ACE_LF_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned long long* e = lf_wg_span[MODE][blockIdx.x];
        e[0] = lf_t0; e[1] = __builtin_amdgcn_s_memtime(); e[2] = (unsigned long long)m; e[3] = hwid; e[4] = xcc & 0xf;
    }
#endif
}

// The 0.25-degree form: one workgroup per CU (dynamic LDS: three units of up to 48 KiB), one wave per SIMD with the whole register
// file - both folded operands of a 721-latitude strip are 384 registers.  Whole 128-column groups and 16-byte stores only.
template <int OUT, int MODE>
__global__ __launch_bounds__(256, 1) void legendre_fold_big_kernel(LegStripArgs p, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem_big[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int m = (idx / G) * 8 + xcd;
    const int grp = idx % G;
    if (m >= p.nbatch) return;
    const FoldGeom gm = fold_geom(MODE, m, p.R, p.K);
    switch (gm.nkp2) {
        case 12: fold_body<6, OUT, true, MODE>(p, smem_big, m, grp, gm); break;
        case 18: fold_body<9, OUT, true, MODE>(p, smem_big, m, grp, gm); break;
        case 24: fold_body<12, OUT, true, MODE>(p, smem_big, m, grp, gm); break;
        default:
            if constexpr (MODE == 1) {   // the inverse's contraction shrinks with m: the high wavenumbers need the small unit sizes too
                switch (gm.nkp2) {
                    case 2: fold_body<1, OUT, true, MODE>(p, smem_big, m, grp, gm); break;
                    case 4: fold_body<2, OUT, true, MODE>(p, smem_big, m, grp, gm); break;
                    case 6: fold_body<3, OUT, true, MODE>(p, smem_big, m, grp, gm); break;
                    default: break;
                }
            }
            break;
    }
}

}  // namespace

#ifdef ACE_LF_TRACE
extern "C" int ace_debug_lf_spans(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(lf_wg_span), sizeof(lf_wg_span)); }
#endif

bool legendre_fold_eligible(const LegStripArgs& a) {
    if (a.K < 1 || a.R < 1 || a.N < 1 || a.nbatch < 1 || (a.mode != 0 && a.mode != 1)) return false;
    const FoldGeom g0 = fold_geom(a.mode, 0, a.R, a.K);
    if (g0.nkp2 > 2 * NH_BIG) return false;
    if (!a.A || !a.tile_off || !a.B || !a.bmax) return false;
    if (g0.nkp2 > 2 * NH_MAX) {   // big form: whole column groups, 16-byte stores
        const bool vec = a.N % 4 == 0 && a.c_rstride % 4 == 0 && a.c_moff % 4 == 0;
        if (a.N % 128 != 0 || !vec || (!a.Chi && (reinterpret_cast<uintptr_t>(a.C) & 15) != 0)) return false;
    }
    // 32-bit UNSIGNED byte offsets from the wavenumber's base.  Forward: rows up to 16 nkp2 (<= nlat / 2 + 96 + 15), and the
    // out-of-range mirror offsets wrap to within that many rows below 2^32 - they must stay above the descriptor's num_records.
    // Inverse: rows up to m + 32 nkp2 <= lmax + 192 (the rounding of the unit size), all of them plain positive offsets.
    const double rowb = (double)a.b_kstride * 4.0;
    if (a.mode == 0 ? ((double)a.K + 2.0 * (16.0 * FOLD_NKP_BIG + 64.0)) * rowb >= 4.0e9 : ((double)a.K + 256.0) * rowb >= 4.0e9) return false;
    if (a.Chi) return a.mode == 0 && a.N % 4 == 0 && a.c_rstride % 4 == 0 && a.c_moff % 4 == 0 && a.cslot &&
                      (reinterpret_cast<uintptr_t>(a.Chi) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.Clo) & 7) == 0;
    return a.C != nullptr;
}

bool legendre_fold_is_big(const LegStripArgs& a) { return fold_geom(a.mode, 0, a.R, a.K).nkp2 > 2 * NH_MAX; }

template <int MODE>
static hipError_t launch_fold_mode(const LegStripArgs& a, hipStream_t s) {
    const int G = (a.N + 127) / 128;
    const int mgroups = (a.nbatch + 7) / 8;
    dim3 grid((unsigned)(mgroups * 8 * G)), block(256);
    const bool fulln = a.N % 128 == 0;
    if (a.Chi) {
        if (fulln) hipLaunchKernelGGL((legendre_fold_kernel<1, true, MODE>), grid, block, 0, s, a, G);
        else hipLaunchKernelGGL((legendre_fold_kernel<1, false, MODE>), grid, block, 0, s, a, G);
    } else {
        const bool vec = a.N % 4 == 0 && a.c_rstride % 4 == 0 && a.c_moff % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
        if (vec && fulln) hipLaunchKernelGGL((legendre_fold_kernel<0, true, MODE>), grid, block, 0, s, a, G);
        else if (vec) hipLaunchKernelGGL((legendre_fold_kernel<0, false, MODE>), grid, block, 0, s, a, G);
        else hipLaunchKernelGGL((legendre_fold_kernel<2, false, MODE>), grid, block, 0, s, a, G);
    }
    return hipGetLastError();
}

template <int OUT, int MODE>
static hipError_t launch_fold_big(const LegStripArgs& a, int nh, hipStream_t s) {
    const int G = a.N / 128;
    const int mgroups = (a.nbatch + 7) / 8;
    const int lds = fold_lds_bytes(nh);
    static bool configured = false;   // per instantiation: the dynamic LDS ceiling is a property of the function
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(legendre_fold_big_kernel<OUT, MODE>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, fold_lds_bytes(NH_BIG));
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL((legendre_fold_big_kernel<OUT, MODE>), dim3((unsigned)(mgroups * 8 * G)), dim3(256), (size_t)lds, s, a, G);
    return hipGetLastError();
}

hipError_t launch_legendre_fold(const LegStripArgs& a, hipStream_t s) {
    const FoldGeom g0 = fold_geom(a.mode, 0, a.R, a.K);
    if (legendre_fold_is_big(a)) {
        const int nh = g0.nkp2 / 2;
        if (a.mode == 0) return a.Chi ? launch_fold_big<1, 0>(a, nh, s) : launch_fold_big<0, 0>(a, nh, s);
        return launch_fold_big<0, 1>(a, nh, s);
    }
    return a.mode == 0 ? launch_fold_mode<0>(a, s) : launch_fold_mode<1>(a, s);
}

}  // namespace ace
