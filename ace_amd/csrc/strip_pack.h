// Operand packing of the "strip" Legendre kernels (strip.hip): host-side, header-only, no HIP dependency so that the
// index algebra can be exercised on the CPU (tests/emul/strip_emul.cpp).
//
// The strip kernels keep the DATA operand (a 32-column strip of X_m or E_m, all of K) in registers as MFMA B fragments
// and stream the TABLE operand (wt_m / pt_m) through LDS as ready-made MFMA A fragments:
//
//   fragment (m, tile t, k-step jj) = 64 lanes x 8 halves, lane = i + 32 g:
//       forward  (sht_fix.py:134-138)  row l = row0(m) + 32 t + i,  column k = 16 jj + 8 g + e        value wt[m][l][k]
//       inverse  (sht_fix.py:208-219)  row k = 32 t + i,            column l = 16 (j0(m) + jj) + 8 g + e   value pt[m][k][l]
//   i.e. exactly what lane (i, g) feeds to v_mfma_f32_32x32x16_f16 as its A operand.  A k-step block is the hi fragment
//   (1 KiB) followed by the lo fragment (1 KiB); a tile is nks4(m) consecutive k-step blocks, so one tile is one
//   contiguous run that a wave moves to LDS with 1-KiB global_load_lds pieces and reads back with conflict-free
//   ds_read_b128 at lane * 16 B.  Rows outside the triangle (l < m), beyond the table, and padded k-steps are zero.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ace {

struct StripGeom {   // per-m geometry shared by the packer, the kernel and the emulator
    int row0;    // first output row of tile 0
    int ntiles;  // 32-row output tiles
    int j0;      // first resident k16-step
    int nks;     // k16-steps that carry data
    int nks4;    // nks rounded up to 4 (the kernel works in groups of four k-steps)
    int klo;     // first valid contraction index
};

// mode 0: forward (rows l in [m, R), contraction over k in [0, K));  mode 1: inverse (rows k in [0, R), contraction
// over l in [m, K))
#if defined(__HIPCC__)
__host__ __device__
#endif
inline StripGeom strip_geom(int mode, int m, int R, int K) {
    StripGeom g;
    if (mode == 0) {
        g.row0 = (m / 32) * 32;
        g.ntiles = (R - g.row0 + 31) / 32;
        g.j0 = 0;
        g.nks = (K + 15) / 16;
        g.klo = 0;
    } else {
        g.row0 = 0;
        g.ntiles = (R + 31) / 32;
        g.j0 = m / 16;
        g.nks = (K + 15) / 16 - g.j0;
        g.klo = m;
    }
    if (g.ntiles < 0) g.ntiles = 0;
    if (g.nks < 0) g.nks = 0;
    // at least one (all-zero) group: a wavenumber with nothing to contract (inverse, m >= 16 ceil(L / 16)) still has to
    // write its zeros
    g.nks4 = g.nks > 0 ? (g.nks + 3) & ~3 : 4;
    return g;
}

// fp32 -> fp16 bits, round to nearest even (normal / subnormal / overflow to inf); host only
inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((em > 0x7f800000u) ? 0x200u : 0u));
    if (em >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   // >= 65520 rounds to inf
    if (em < 0x33000001u) return (uint16_t)sign;                 // < 2^-25 (or exactly 2^-25: ties to even = 0)
    int e = (int)(em >> 23) - 127;
    uint32_t man = (em & 0x7fffffu) | 0x800000u;                 // 24-bit significand
    int shift;                                                   // bits dropped from the 24-bit significand
    uint32_t base;
    if (e >= -14) { shift = 13; base = (uint32_t)(e + 15) << 10; man &= 0x7fffffu; }
    else { shift = 13 + (-14 - e); base = 0; }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;            // carries propagate into the exponent correctly
    return (uint16_t)(sign | (base + q));
}
inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)man, -24);
    else if (e == 31) v = man ? NAN : INFINITY;
    else v = std::ldexp((float)(man | 0x400u), (int)e - 25);
    uint32_t b;
    std::memcpy(&b, &v, 4);
    b |= sign;
    std::memcpy(&v, &b, 4);
    return v;
}

struct StripPack {
    std::vector<uint16_t> frags;   // fp16 bits: [k-step block][hi 512 | lo 512]
    std::vector<int> tile_off;     // per m: index of the first k-step block of tile 0
    long blocks = 0;
};

// tab: forward wt[m][l][pitch] (rows = l, cols = k); inverse pt[m][k][pitch] (rows = k, cols = l).
// nrows / ncols: logical extents (forward: L, H; inverse: H, L).  Values are multiplied by `scale` (a power of two).
inline void pack_legendre_strip(const float* tab, int mmax, int nrows, int ncols, int pitch, int mode, float scale,
                                StripPack& out) {
    out.tile_off.assign((size_t)mmax, 0);
    long blocks = 0;
    for (int m = 0; m < mmax; ++m) {
        const StripGeom g = strip_geom(mode, m, nrows, ncols);
        out.tile_off[m] = (int)blocks;
        blocks += (long)g.ntiles * g.nks4;
    }
    out.blocks = blocks;
    out.frags.assign((size_t)blocks * 1024, 0);
    for (int m = 0; m < mmax; ++m) {
        const StripGeom g = strip_geom(mode, m, nrows, ncols);
        for (int t = 0; t < g.ntiles; ++t)
            for (int jj = 0; jj < g.nks; ++jj) {
                uint16_t* blk = out.frags.data() + ((size_t)out.tile_off[m] + (size_t)t * g.nks4 + jj) * 1024;
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, gg = lane >> 5;
                    const int row = g.row0 + 32 * t + i;
                    for (int e = 0; e < 8; ++e) {
                        const int col = 16 * (g.j0 + jj) + 8 * gg + e;
                        float v = 0.f;
                        // the contraction index is l for the inverse (valid from m), the row index is l for the forward
                        const bool ok = row < nrows && col < ncols && (mode == 0 ? row >= m : col >= m);
                        if (ok) v = tab[((size_t)m * nrows + row) * pitch + col] * scale;
                        const uint16_t hi = f32_to_f16_bits(v);
                        const uint16_t lo = f32_to_f16_bits(v - f16_bits_to_f32(hi));
                        blk[lane * 8 + e] = hi;
                        blk[512 + lane * 8 + e] = lo;
                    }
                }
            }
    }
}

// ---- equatorially folded form (strip_fold.hip) ------------------------------------------------------------------------------
// For grids whose nodes and weights are symmetric about the equator (legendre-gauss, lobatto, equiangular - checked on the host
// table, fold_symmetry_error), P_l^m(-x) = (-1)^(l+m) P_l^m(x) halves the latitude contraction:
//     forward  c[l][m] = sum_{kf < Hh} wt[m][l][kf] (X[kf] + s X[H-1-kf]),  s = +1 for (l - m) even, -1 for (l - m) odd
//     inverse  X[kf] = E[kf] + O[kf],  X[H-1-kf] = E[kf] - O[kf],  E / O = sum over l of even / odd (l - m) of pt[m][kf][l] c[l]
// with Hh = ceil(H / 2) (an odd H has a middle row that is its own mirror image: it enters once).  The wave keeps BOTH folded
// data operands resident (even: the sums / the even-degree coefficient rows, odd: the differences / the odd-degree rows) and the
// table streams as "units": unit u = 2 tp + p is the 32-row tile tp of parity p over nkp2 k16-steps.  Forward: unit (tp, p)
// yields the output rows l = m + p + 2 (32 tp + r); inverse: the pair of units tp yields rows kf = 32 tp + r (E + O) and their
// mirror images H - 1 - kf (E - O).  Contraction indices start AT the triangle (l = m + p + 2 j), so there is no l < m masking.
struct FoldGeom {
    int Hh;      // folded latitude count ceil(H / 2)
    int nkp;     // k16-steps per parity that carry data
    int nkp2;    // ... rounded up to an instantiated count (fold_round_nkp; a unit is nkp2 2-KiB blocks = nkp2 / 2 pieces per wave)
    int npairs;  // pairs of units (32-row tiles per parity)
};
// k16-steps per unit the kernels are instantiated for: 2, 4, 6 (two workgroups per CU: 1-degree grids) and 12, 18, 24 (one
// workgroup per CU with all 512 registers per lane: up to 384 folded latitudes / degrees per parity - the 0.25-degree grid).  At
// least one (all-zero) group: an inverse wavenumber with nothing to contract still writes its zeros.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int fold_round_nkp(int nkp) {
    if (nkp <= 6) return nkp > 0 ? (nkp + 1) & ~1 : 2;
    return ((nkp + 5) / 6) * 6;
}
constexpr int FOLD_NKP_SMALL = 6, FOLD_NKP_BIG = 24;

#if defined(__HIPCC__)
__host__ __device__
#endif
inline FoldGeom fold_geom(int mode, int m, int R, int K) {
    FoldGeom g;
    int n16;
    if (mode == 0) {            // R = lmax rows, K = nlat
        g.Hh = (K + 1) / 2;
        n16 = g.Hh;
        const int nr0 = (R - m + 1) / 2;          // rows of even parity l = m, m + 2, ... < R
        g.npairs = nr0 > 0 ? (nr0 + 31) / 32 : 0;
    } else {                    // R = nlat rows, K = lmax
        g.Hh = (R + 1) / 2;
        n16 = (K - m + 1) / 2;                    // even-parity degrees l = m, m + 2, ... < K
        if (n16 < 0) n16 = 0;
        g.npairs = (g.Hh + 31) / 32;
    }
    g.nkp = (n16 + 15) / 16;
    g.nkp2 = fold_round_nkp(g.nkp);
    return g;
}

// max |tab[..][H-1-k] - s tab[..][k]| / max |tab| over the table (forward layout wt[m][l][pitch] or inverse pt[m][k][pitch]):
// the fold is used when this is at rounding level
inline double fold_symmetry_error(const float* tab, int mmax, int nlat, int lmax, int pitch, int mode) {
    double worst = 0.0, mx = 0.0;
    for (int m = 0; m < mmax; ++m)
        for (int l = m; l < lmax; ++l) {
            const double s = ((l - m) & 1) ? -1.0 : 1.0;
            for (int k = 0; k < nlat; ++k) {
                const double a = mode == 0 ? tab[((size_t)m * lmax + l) * pitch + k] : tab[((size_t)m * nlat + k) * pitch + l];
                const double b = mode == 0 ? tab[((size_t)m * lmax + l) * pitch + (nlat - 1 - k)] : tab[((size_t)m * nlat + (nlat - 1 - k)) * pitch + l];
                worst = std::fmax(worst, std::fabs(b - s * a));
                mx = std::fmax(mx, std::fabs(a));
            }
        }
    return mx > 0.0 ? worst / mx : 0.0;
}

// tab as for pack_legendre_strip.  nlat / lmax: logical extents.  The folded entry is the mean of the two mirror entries (they
// differ by fp32 rounding of the fp64 recursion at most); the middle row of an odd nlat keeps its own value.
inline void pack_legendre_fold(const float* tab, int mmax, int nlat, int lmax, int pitch, int mode, float scale, StripPack& out) {
    const int R = mode == 0 ? lmax : nlat, K = mode == 0 ? nlat : lmax;
    out.tile_off.assign((size_t)mmax, 0);
    long blocks = 0;
    for (int m = 0; m < mmax; ++m) {
        const FoldGeom g = fold_geom(mode, m, R, K);
        out.tile_off[m] = (int)blocks;
        blocks += (long)g.npairs * 2 * g.nkp2;
    }
    out.blocks = blocks;
    out.frags.assign((size_t)blocks * 1024, 0);
    for (int m = 0; m < mmax; ++m) {
        const FoldGeom g = fold_geom(mode, m, R, K);
        for (int u = 0; u < 2 * g.npairs; ++u) {
            const int tp = u >> 1, p = u & 1;
            for (int jj = 0; jj < g.nkp; ++jj) {
                uint16_t* blk = out.frags.data() + ((size_t)out.tile_off[m] + (size_t)u * g.nkp2 + jj) * 1024;
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, gg = lane >> 5;
                    for (int e = 0; e < 8; ++e) {
                        const int c = 16 * jj + 8 * gg + e;
                        // forward: row = degree index, column = folded latitude; inverse: row = folded latitude, column = degree index
                        const int kf = mode == 0 ? c : 32 * tp + i;
                        const int l = m + p + 2 * (mode == 0 ? 32 * tp + i : c);
                        float v = 0.f;
                        if (kf < g.Hh && l < lmax) {
                            const int km = nlat - 1 - kf;
                            const double s = p ? -1.0 : 1.0;
                            const double a = mode == 0 ? tab[((size_t)m * lmax + l) * pitch + kf] : tab[((size_t)m * nlat + kf) * pitch + l];
                            const double b = mode == 0 ? tab[((size_t)m * lmax + l) * pitch + km] : tab[((size_t)m * nlat + km) * pitch + l];
                            v = (float)((km == kf ? a : 0.5 * (a + s * b)) * (double)scale);
                        }
                        const uint16_t hi = f32_to_f16_bits(v);
                        const uint16_t lo = f32_to_f16_bits(v - f16_bits_to_f32(hi));
                        blk[lane * 8 + e] = hi;
                        blk[512 + lane * 8 + e] = lo;
                    }
                }
            }
        }
    }
}

// ---- conditional layer norm on MFMA (cln_mfma.hip): a (C x J) 1x1-convolution weight as error-compensated A fragments of
// v_mfma_f32_32x32x16_f16, [C / 32 row tiles][ceil(J / 16) k-steps][hi 512 | lo 512] halves; lane (i, g) element e of a block is
// W[32 rt + i][16 ks + 8 g + e] * scale (zero beyond J).  `scale` = cln_frag_scale(max |W|): max |W| scale in [2^9, 2^10).
inline float cln_frag_scale(float wabs) {
    int e = 0;
    if (wabs > 0.f && std::isfinite(wabs)) { (void)std::frexp(wabs, &e); e = 10 - e; }
    return std::ldexp(1.0f, e);
}
inline size_t cln_frag_halves(int C, int J) { return (size_t)(C / 32) * (size_t)((J + 15) / 16) * 1024; }
inline void pack_cln_frags(const float* W, int C, int J, float scale, uint16_t* out) {
    const int nk = (J + 15) / 16;
    for (int rt = 0; rt < C / 32; ++rt)
        for (int ks = 0; ks < nk; ++ks) {
            uint16_t* blk = out + ((size_t)rt * nk + ks) * 1024;
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = 32 * rt + (lane & 31), j = 16 * ks + 8 * (lane >> 5) + e;
                    const float v = j < J ? W[(size_t)row * J + j] * scale : 0.f;
                    const uint16_t hi = f32_to_f16_bits(v);
                    blk[lane * 8 + e] = hi;
                    blk[512 + lane * 8 + e] = f32_to_f16_bits(v - f16_bits_to_f32(hi));
                }
        }
}

}  // namespace ace
