// "Strip" Legendre kernels for gfx950: the latitude quadrature / synthesis of the spherical harmonic transform
// (fme/sht_fix.py:134-138 forward, :208-219 inverse) with the DATA operand resident in registers.
//
// Per zonal wavenumber m the transform is a small GEMM (K <= 192 at the 1-degree grid):
//     forward   D_m[l][n] = sum_k wt_m[l][k] X_m[k][n]      l in [m, L)
//     inverse   X_m[k][n] = sum_l pt_m[k][l] E_m[l][n]      l in [m, L)
// with n = (sample, re|im, channel) 768 wide at the ACE2 shape.  Both are memory bound (223 MB of traffic for 13 GF).
// A tile engine that stages both operands through LDS spends most of a K = 180 tile in its prologue and epilogue
// (DESIGN.md section 3.4); here instead
//   * a wave owns a 32-column strip of the data operand for the WHOLE contraction: it loads it once (fp32, 128-byte row
//     segments), splits it into fp16 hi/lo MFMA B fragments in registers (96 VGPRs at K = 192) and never touches it again;
//   * the table operand arrives as ready-made A fragments (strip_pack.h) by 1-KiB LDS-DMA pieces into a three-slot ring
//     shared by the four waves of the workgroup, one slot = one 32-row output tile, two tiles of lookahead (an LDS-DMA
//     piece issued under full-chip load lands 2 - 3 us later);
//   * per tile: one barrier, 3 x 12 MFMAs (f16x3: lo.hi + hi.lo + hi.hi, fp32 accumulate) against the resident strip;
//   * the epilogue of tile t - 1 (scale, split to fp16 planes or fp32, 8/16-byte row-contiguous stores through a 4 KiB
//     per-wave LDS transpose) is issued under the MFMAs of tile t, right after the DMA of tile t + 2; vmcnt counts stores
//     on gfx950 and loads / stores may retire out of order with respect to each other, so the wait at the top of an
//     iteration is the count that is safe under that rule (see the loop);
//   * workgroups are dealt to XCDs by m mod 8: the table slice of one m stays in one L2 and every XCD gets the same mix of
//     heavy (small m) and light (large m) workgroups, heavy ones first.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "strip_pack.h"

namespace ace {
namespace {

#define SDEV __device__ __forceinline__

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char* lds_cptr;

constexpr int KS_MAX = 12;                  // resident k16-steps (K <= 192)
constexpr int SLOT_BYTES = KS_MAX * 2048;   // one 32-row tile of A fragments (hi + lo)
constexpr int NSLOT = 3;                    // ring depth: tile t + 2 is in flight while tile t is consumed
constexpr int TS_BYTES = 2048;              // per-wave transpose buffer: 16 rows x 32 dwords (half a tile at a time)
constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + 4 * TS_BYTES;   // 80 KiB: two workgroups per CU

SDEV unsigned slot_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SDEV int pow2_exponent_for(float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 12 - e; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
SDEV float wave_max_bits(unsigned raw) {
    float mx = __uint_as_float(raw);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
}
// one 1-KiB LDS-DMA piece: lane L's 16 bytes at gsrc land at lds_dst + 16 L (see kernels.hip glds16)
SDEV void glds16(const void* gsrc, const char* lds_dst_uniform) {
    // m0 = LDS base of the copy; declared clobbered instead of saved and restored around every piece (two scalar instructions
    // per piece: the compiler keeps nothing in m0 across these kernels' loops)
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)lds_dst_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(addr)
                 : "memory", "m0");
}
SDEV int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// OUT: 0 = fp32, 16-byte stores (N % 4 == 0, aligned); 1 = fp16 hi/lo planes, 8-byte stores; 2 = fp32 scalar stores
template <int NG, int OUT, bool FULLN>
SDEV void strip_body(const LegStripArgs& p, char* smem, const int m, const int grp, const StripGeom gm) {
    constexpr int NK = 4 * NG;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int N = p.N, K = p.K, R = p.R;
    const int n0 = grp * 128 + wave * 32;
    const int n = n0 + i;
    const int nc = n < N ? n : N - 1;
    char* ring = smem;
    char* Ts = smem + NSLOT * SLOT_BYTES + wave * TS_BYTES;

    const unsigned raw_b = p.bmax ? slot_load(p.bmax + lane) : 0u;

    // ---- table stream: tile t = nks4 consecutive 2-KiB k-step blocks = 2 nks4 pieces, 2 NG per wave
    const _Float16* Am = p.A + (long)p.tile_off[m] * 1024;
    auto issue_tile = [&](int t) {   // tile t -> slot t % NSLOT; tiles past the end re-fetch the last one (uniform count)
        const int tt = t < gm.ntiles ? t : gm.ntiles - 1;
        const _Float16* src = Am + (long)tt * NK * 1024 + lane * 8;
        const char* dst = ring + (t % NSLOT) * SLOT_BYTES;
#pragma unroll
        for (int c = 0; c < 2 * NG; ++c) {
            const int pc = wave + 4 * c;
            glds16(src + pc * 512, dst + pc * 1024);
        }
    };
    if (gm.ntiles > 0) { issue_tile(0); issue_tile(1); }

    // ---- resident strip: fp32 rows -> fp16 hi/lo B fragments (lane (i, g) holds k = 16 jj + 8 g .. + 7 of column i)
    const float* Bm = p.B + (long)m * p.b_moff + nc;
    const long ks = p.b_kstride;
    float bbound = 0.f, inv_b = 1.f, bscale = 1.f;
    if (p.bmax) {
        bbound = wave_max_bits(raw_b);
        const int eb = pow2_exponent_for(bbound);
        bscale = ldexpf(1.0f, eb);
        inv_b = ldexpf(1.0f, -eb);
    }
    // Loads are un-clamped per lane (uniform offsets); the VALUES outside [klo, K) are zeroed after the load.  Indices >= K
    // read at most 15 rows past K: the next batch's rows or the 16 rows of slack the library's buffers carry
    // (LEG_STRIP_SLACK_ROWS); padded k-steps re-read the last real one (wave-uniform clamp).  Indices below klo (inverse:
    // l < m, never written by the producer - possibly stale bits of another layout) only occur in the first resident k-step.
    half8 bh[NK], bl[NK];
    {
        // Through a range-checked buffer descriptor over the K rows of THIS batch: a row at or beyond K (the tail of the last real
        // k-step, every padded k-step) is out of range and reads as an exact zero - no clamp, no per-element mask, no stale data
        // of an earlier, larger call (the reason the mask existed), and 32-bit offsets: per lane one offset per row e of its
        // k-group plus the k-step's (uniform) offset, one 32-bit add per load.
        const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B + (long)m * p.b_moff), 0,
                                                           (unsigned)((long)K * ks * 4 < 0xFFFFFFFFl ? (long)K * ks * 4 : 0xFFFFFFFFl), 0x00020000);
        unsigned vo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vo[e] = (unsigned)(((long)(8 * g + e) * ks + nc) * 4);
        float raw[NK][8];
#pragma unroll
        for (int jj = 0; jj < NK; ++jj) {
            // the k-step's offset goes into the VECTOR offset: the hardware range check of a raw buffer covers the instruction and
            // vector offsets only, not the scalar one
            const unsigned so = (unsigned)((long)(16 * (gm.j0 + jj)) * ks * 4);     // wave-uniform
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[jj][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, vo[e] + so, 0, 0));
        }
#pragma unroll
        for (int jj = 0; jj < NK; ++jj)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = raw[jj][e] * bscale;
                if (jj == 0) {   // indices below klo (inverse: l < m, never written by the producer) only occur in the first resident k-step
                    const int kk = 16 * gm.j0 + 8 * g + e;
                    x = kk >= gm.klo ? x : 0.f;
                }
                const _Float16 h = (_Float16)x;
                bh[jj][e] = h;
                bl[jj][e] = (_Float16)(x - (float)h);
            }
    }

    // ---- scales
    const float inv_a = 1.0f / p.ascale;
    float oscale = inv_a * inv_b;
    if (OUT == 1) {   // planes: bound of this launch's output, identical in every workgroup; the consumer reads it from cslot
        const float cbound = p.cw * bbound;
        oscale *= ldexpf(1.0f, pow2_exponent_for(cbound));
        if (tid == 0) atomicMax(p.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
    }

    float vmax = 0.f;
    const long cm = (long)m * p.c_moff;
    // `inner`: a tile that is not the last one of its column.  Its rows are all below R, and rows below the triangle
    // (forward, l < m) hold exact zeros at addresses nobody reads, so with whole strips (FULLN) the stores need no mask:
    // the in-loop epilogue is straight-line code that the scheduler can lay under the MFMAs of the next tile.
    auto tile_max = [&](int t, const f32x16& acc, auto inner_tag) {   // range of the fp32 outputs of tile t (valid entries only)
        constexpr bool NOMASK = decltype(inner_tag)::value && FULLN;
        const int rbase = gm.row0 + 32 * t;
        const int rlo = p.mode == 0 ? m : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + acc_row(r, g);
            const bool ok = NOMASK || (row >= rlo && row < R && n < N);
            vmax = fmaxf(vmax, ok ? fabsf(acc[r] * oscale) : 0.f);
        }
    };
    auto store_tile = [&](int t, const f32x16& acc, auto inner_tag) {
        constexpr bool NOMASK = decltype(inner_tag)::value && FULLN;
        const int rbase = gm.row0 + 32 * t;
        const int rlo = p.mode == 0 ? m : 0;
        if (OUT == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + acc_row(r, g);
                const float v = acc[r] * oscale;
                if (row >= rlo && row < R && n < N) p.C[cm + (long)row * p.c_rstride + n] = v;
            }
            return;
        }
        // row-major through this wave's transpose buffer, half a tile (16 rows = 8 accumulator registers) at a time:
        // dword (row, column) at row * 32 + column, read back as four consecutive columns of one row per lane (8 lanes
        // cover a row, 8 rows per instruction)
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
                const int row = rbase + 16 * hf + rl;
                const u32x4 d = *reinterpret_cast<const u32x4*>(T32 + rl * 32 + c4);
                const bool ok = NOMASK || (row >= rlo && row < R && ncol < N);
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

    // ---- tiles
    f32x16 prev;
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[r] = 0.f;
    // vmcnt counts stores too and loads / stores may retire out of order with respect to each other, so the wait is not a
    // plain count of newer operations.  Queue at the top of iteration t >= 1 (issue order): [pieces of tile t | stores of
    // tile t - 3] from iteration t - 2, [pieces of tile t + 1 | stores of tile t - 2] from iteration t - 1.  Loads retire in
    // order, so "a piece of tile t pending" implies "all 2 NG pieces of tile t + 1 pending", i.e. more than 2 NG
    // operations outstanding: vmcnt(2 NG) therefore guarantees tile t has landed, while tile t + 1 may stay in flight.
    for (int t = 0; t < gm.ntiles; ++t) {
        if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tiles 0 and 1 and the data strip
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NG) : "memory");
        __syncthreads();                                     // every share landed; every wave is done reading tile t - 1
        issue_tile(t + 2);                                   // ... whose slot is refilled now (a dummy past the end)
        if (t > 0) {
            if (OUT != 1) tile_max(t - 1, prev, std::true_type{});
            store_tile(t - 1, prev, std::true_type{});
        }
        const char* slot = ring + (t % NSLOT) * SLOT_BYTES + lane * 16;
        f32x16 a0, a1, a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }
        half8 fh[NK], fl[NK];
#pragma unroll
        for (int jj = 0; jj < NK; ++jj) {
            fh[jj] = *reinterpret_cast<const half8*>(slot + jj * 2048);
            fl[jj] = *reinterpret_cast<const half8*>(slot + jj * 2048 + 1024);
        }
#pragma unroll
        for (int jj = 0; jj < NK; ++jj) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[jj], bh[jj], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[jj], bl[jj], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[jj], bh[jj], a2, 0, 0, 0);
        }
        prev = (a0 + a1) + a2;
    }
    // gfx950 store-data rule (profiles/r02_store_data_hazard.txt): the range reduction (shuffles, an LDS round trip - loads that
    // LAND in registers) runs before the last tile's stores, which are the last thing the wave does
    if (OUT != 1 && gm.ntiles > 0) tile_max(gm.ntiles - 1, prev, std::false_type{});
    if (OUT != 1 && p.omax) {   // one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        __syncthreads();        // ... before the transpose buffer of the last tile re-uses the LDS
    }
    if (gm.ntiles > 0) store_tile(gm.ntiles - 1, prev, std::false_type{});
}

template <int OUT, bool FULLN>
__global__ __launch_bounds__(256, 2) void legendre_strip_kernel(LegStripArgs p, int G) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    // workgroup b runs on XCD b % 8: deal the wavenumbers round-robin so that all groups of one m share an L2 and the
    // triangular load is even across XCDs; ascending m = heaviest first
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int m = (idx / G) * 8 + xcd;
    const int grp = idx % G;
    if (m >= p.nbatch) return;
    const StripGeom gm = strip_geom(p.mode, m, p.R, p.K);
    switch (gm.nks4) {
        case 4: strip_body<1, OUT, FULLN>(p, smem, m, grp, gm); break;
        case 8: strip_body<2, OUT, FULLN>(p, smem, m, grp, gm); break;
        case 12: strip_body<3, OUT, FULLN>(p, smem, m, grp, gm); break;
        default: break;   // nks == 0: nothing to contract (cannot happen for m < mmax <= lmax)
    }
}

}  // namespace

bool legendre_strip_eligible(const LegStripArgs& a) {
    if (a.K < 1 || a.R < 1 || a.N < 1 || a.nbatch < 1) return false;
    if ((a.K + 15) / 16 > KS_MAX) return false;
    if (!a.A || !a.tile_off || !a.B || !a.bmax) return false;
    // the data operand is addressed with 32-bit byte offsets through a range-checked buffer descriptor (rows up to 15 past the
    // last k-step, clamped to 4 GiB - 1): a batch whose K rows do not fit wraps the offsets and reads in-range wrong rows
    // (inverse transform: b_kstride = mmax * 2 * B * C, i.e. B * C >= ~16 k at 1 degree).  Larger operands go to the tile engine.
    if (((double)a.K + 16.0) * (double)a.b_kstride * 4.0 >= 4.0e9) return false;
    if (a.Chi) return a.N % 4 == 0 && a.c_rstride % 4 == 0 && a.c_moff % 4 == 0 && a.cslot &&
                      (reinterpret_cast<uintptr_t>(a.Chi) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.Clo) & 7) == 0;
    return a.C != nullptr;
}

hipError_t launch_legendre_strip(const LegStripArgs& a, hipStream_t s) {
    const int G = (a.N + 127) / 128;
    const int mgroups = (a.nbatch + 7) / 8;
    dim3 grid((unsigned)(mgroups * 8 * G)), block(256);
    const bool fulln = a.N % 128 == 0;
    if (a.Chi) {
        if (fulln) hipLaunchKernelGGL((legendre_strip_kernel<1, true>), grid, block, 0, s, a, G);
        else hipLaunchKernelGGL((legendre_strip_kernel<1, false>), grid, block, 0, s, a, G);
    } else {
        const bool vec = a.N % 4 == 0 && a.c_rstride % 4 == 0 && a.c_moff % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
        if (vec && fulln) hipLaunchKernelGGL((legendre_strip_kernel<0, true>), grid, block, 0, s, a, G);
        else if (vec) hipLaunchKernelGGL((legendre_strip_kernel<0, false>), grid, block, 0, s, a, G);
        else hipLaunchKernelGGL((legendre_strip_kernel<2, false>), grid, block, 0, s, a, G);
    }
    return hipGetLastError();
}

}  // namespace ace
