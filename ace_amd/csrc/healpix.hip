// HEALPix variant (BASELINE configs[4]): the operators of the reference's HEALPix UNet
// (fme/ace/models/healpix/) as gfx950 kernels behind the C ABI - neighbourhood convolutions on the 12-face mesh in place of the
// spherical harmonic transform, same stepper API.
//   * face padding (healpix_paddings.py:239-611, Karlbauer et al. 2024; the reference states that its default "earth2grid"
//     backend gives the same result): every halo cell of a padded face is ONE cell of a neighbouring face (rotated for the
//     polar faces) or, on the diagonal of the two corners an equatorial face has no neighbour for, the mean of two.  Built
//     once on the host as a gather table (ace_hpx_pad_table_host), applied by one gather kernel;
//   * k x k (dilated) convolution of a padded face = ONE contraction over (tap, input channel): activations of one UNet
//     level are stored as [channels][rows][P] with the row pitch P (a multiple of 4) of that level's padded faces, so row
//     (tap, i) of the B operand is channel i of the padded tensor shifted by the tap's constant offset - a row-offset table
//     (GemmArgs::brow) instead of an im2col copy.  The engine is the compensated-fp16 one (kernels.hip gemm3: weights
//     pre-split into fp16 hi/lo planes once per parameter, activations split on the fly, fp32 accumulation - the same
//     fp32-class arithmetic as the SFNO path), with bias, residual, capped GELU and the output's magnitude bound in the
//     epilogue.  (Round 3's first version ran k^2 accumulated fp32-MFMA GEMMs per convolution: 9 read-modify-write passes
//     over the output; 16.8 ms per forward at nside 64.)
//   * dynamic range: every tensor carries a 64-shard slot with a bound on max|x| (written by its producer: the padding
//     gather, a convolution's epilogue, the transposed convolution's scatter; pooling passes its input's bound on);
//   * gap columns [W, P) of a row are always DEFINED (zeros from the padding kernel / zero-initialised buffers, finite values
//     from convolutions): the contraction runs over whole rows and its results in the gaps are never read as data;
//   * 2 x 2 average / max pooling; 2 x 2 stride-2 transposed convolution = one GEMM with the four taps stacked along the
//     output rows + an interleaving scatter with the bias and activation; channel concatenation by writing two sources
//     into one padded tensor / two row sources of one GEMM.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/ace_sfno.h"
#include "kernels.h"
#include "strip_common.h"

using namespace ace;

#define ACE_HPX_SLACK 16   // floats the caller keeps behind a padded tensor (read, never used, by the last taps of the last row)

static thread_local std::string g_herr;
static int hfail(int code, const std::string& m) { g_herr = m; return code; }
extern "C" const char* ace_hpx_last_error(void) { return g_herr.c_str(); }
#define HPX_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) return hfail(ACE_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// gather table: for every cell (face f, row y, column x) of the padded mesh [12][H + 2p][H + 2p] two source cells of the
// unpadded mesh, packed (face << 24 | row << 12 | column); out = 0.5 a + 0.5 b (b = a where the cell is a plain copy)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Face {           // a view of source cells: n x n packed indices, row-major
    int n;
    std::vector<int> v;
    int& at(int y, int x) { return v[(size_t)y * n + x]; }
    int at(int y, int x) const { return v[(size_t)y * n + x]; }
};
Face source_face(int f, int n) {
    Face F{n, std::vector<int>((size_t)n * n)};
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) F.at(y, x) = (f << 24) | (y << 12) | x;
    return F;
}
// torch.rot90(k, dims=(-2, -1)): k = 1 rotates counter-clockwise: out[i][j] = in[j][n - 1 - i]
Face rot90(const Face& a, int k) {
    k = ((k % 4) + 4) % 4;
    Face r = a;
    for (int t = 0; t < k; ++t) {
        Face o{r.n, std::vector<int>(r.v.size())};
        for (int i = 0; i < r.n; ++i)
            for (int j = 0; j < r.n; ++j) o.at(i, j) = r.at(j, r.n - 1 - i);
        r = o;
    }
    return r;
}
struct Padded {         // (n + 2p)^2 cells, two sources each
    int n, p;
    std::vector<int> a, b;
    void set(int y, int x, int s) { a[(size_t)y * (n + 2 * p) + x] = s; b[(size_t)y * (n + 2 * p) + x] = s; }
    void set2(int y, int x, int s, int t) { a[(size_t)y * (n + 2 * p) + x] = s; b[(size_t)y * (n + 2 * p) + x] = t; }
};
// block copy helper: dst[y0 + i][x0 + j] = src view [sy + i][sx + j] for i < h, j < w
void blit(Padded& P, int y0, int x0, const Face& s, int sy, int sx, int h, int w) {
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) P.set(y0 + i, x0 + j, s.at(sy + i, sx + j));
}
enum Hemi { NORTH, EQUATOR, SOUTH };
// healpix_paddings.py:379-543 (pn / pe / ps): c centre, t top, tl top-left, l left, bl bottom-left, b bottom, br bottom-right,
// r right, tr top-right.  For equatorial faces tl / br do not exist: the corner blocks come from corner_tl / corner_br.
Padded pad_face(Hemi h, int n, int p, const Face& c, const Face& t, const Face& tl, const Face& l, const Face& bl, const Face& b,
                const Face& br, const Face& r, const Face& tr, const Padded* ctl, const Padded* cbr) {
    const int m = n + 2 * p;
    Padded P{n, p, std::vector<int>((size_t)m * m, 0), std::vector<int>((size_t)m * m, 0)};
    const Face tt = h == NORTH ? rot90(t, 1) : t;
    const Face bb = h == SOUTH ? rot90(b, 1) : b;
    blit(P, 0, p, tt, n - p, 0, p, n);            // top strip: last p rows of the (rotated) top neighbour
    blit(P, p, p, c, 0, 0, n, n);
    blit(P, p + n, p, bb, 0, 0, p, n);            // bottom strip: first p rows
    // left column: [tl corner; left strip; bl corner]
    if (h == NORTH) {
        blit(P, 0, 0, rot90(tl, 2), n - p, n - p, p, p);
        blit(P, p, 0, rot90(l, -1), 0, n - p, n, p);
    } else {
        if (h == EQUATOR) {
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) P.set2(i, j, ctl->a[(size_t)i * p + j], ctl->b[(size_t)i * p + j]);
        } else {
            blit(P, 0, 0, tl, n - p, n - p, p, p);
        }
        blit(P, p, 0, l, 0, n - p, n, p);
    }
    blit(P, p + n, 0, bl, 0, n - p, p, p);
    // right column: [tr corner; right strip; br corner]
    blit(P, 0, p + n, tr, n - p, 0, p, p);
    if (h == SOUTH) {
        blit(P, p, p + n, rot90(r, -1), 0, 0, n, p);
        blit(P, p + n, p + n, rot90(br, 2), 0, 0, p, p);
    } else {
        blit(P, p, p + n, r, 0, 0, n, p);
        if (h == EQUATOR) {
            for (int i = 0; i < p; ++i)
                for (int j = 0; j < p; ++j) P.set2(p + n + i, p + n + j, cbr->a[(size_t)i * p + j], cbr->b[(size_t)i * p + j]);
        } else {
            blit(P, p + n, p + n, br, 0, 0, p, p);
        }
    }
    return P;
}
// healpix_paddings.py:545-580: the p x p top-left corner of an equatorial face from its top and left neighbours
Padded corner_tl(const Face& top, const Face& lft, int p) {
    const int n = top.n;
    Padded C{0, 0, std::vector<int>((size_t)p * p, 0), std::vector<int>((size_t)p * p, 0)};
    auto put = [&](int y, int x, int s, int t) { C.a[(size_t)y * p + x] = s; C.b[(size_t)y * p + x] = t; };
    // negative indices of the reference are relative to the block (p) for ret and to the face (n) for top / lft
    put(p - 1, p - 1, top.at(n - 1, 0), lft.at(0, n - 1));
    for (int i = 1; i < p; ++i) {
        for (int j = 0; j < i; ++j) {
            put(p - i - 1, p - i + j, top.at(n - i - 1, j), top.at(n - i - 1, j));      // above the diagonal: from the top face
            put(p - i + j, p - i - 1, lft.at(j, n - i - 1), lft.at(j, n - i - 1));      // below: from the left face
        }
        put(p - i - 1, p - i - 1, top.at(n - i - 1, 0), lft.at(0, n - i - 1));          // diagonal: mean of the two
    }
    return C;
}
// healpix_paddings.py:582-610: the bottom-right corner from the bottom and right neighbours
Padded corner_br(const Face& b, const Face& r, int p) {
    const int n = b.n;
    Padded C{0, 0, std::vector<int>((size_t)p * p, 0), std::vector<int>((size_t)p * p, 0)};
    auto put = [&](int y, int x, int s, int t) { C.a[(size_t)y * p + x] = s; C.b[(size_t)y * p + x] = t; };
    put(0, 0, b.at(0, n - 1), r.at(n - 1, 0));
    for (int i = 1; i < p; ++i) {
        for (int j = 0; j < i; ++j) {
            put(j, i, r.at(n - i + j, i), r.at(n - i + j, i));      // above the diagonal: from the right face
            put(i, j, b.at(i, n - i + j), b.at(i, n - i + j));      // below: from the bottom face
        }
        put(i, i, b.at(i, n - 1), r.at(n - 1, i));
    }
    return C;
}

}  // namespace

extern "C" int ace_hpx_pad_table_host(int nside, int p, int* idx_a_host, int* idx_b_host) {
    if (nside < 1 || nside > 4095 || p < 1 || p > nside || !idx_a_host || !idx_b_host)
        return hfail(ACE_ERR_INVALID, "ace_hpx_pad_table_host: need 1 <= padding <= nside <= 4095");
    std::vector<Face> f;
    for (int k = 0; k < 12; ++k) f.push_back(source_face(k, nside));
    // neighbour tables of healpix_paddings.py:299-367, in the order (t, tl, l, bl, b, br, r, tr)
    static const int N[4][8] = {{1, 2, 3, 3, 4, 8, 5, 1}, {2, 3, 0, 0, 5, 9, 6, 2}, {3, 0, 1, 1, 6, 10, 7, 3}, {0, 1, 2, 2, 7, 11, 4, 0}};
    static const int E[4][8] = {{0, -1, 3, 7, 11, -1, 8, 5}, {1, -1, 0, 4, 8, -1, 9, 6}, {2, -1, 1, 5, 9, -1, 10, 7}, {3, -1, 2, 6, 10, -1, 11, 4}};
    static const int S[4][8] = {{5, 0, 4, 11, 11, 10, 9, 9}, {6, 1, 5, 8, 8, 11, 10, 10}, {7, 2, 6, 9, 9, 8, 11, 11}, {4, 3, 7, 10, 10, 9, 8, 8}};
    const int m = nside + 2 * p;
    for (int k = 0; k < 12; ++k) {
        const int q = k % 4;
        Padded P;
        if (k < 4) {
            const int* n8 = N[q];
            P = pad_face(NORTH, nside, p, f[k], f[n8[0]], f[n8[1]], f[n8[2]], f[n8[3]], f[n8[4]], f[n8[5]], f[n8[6]], f[n8[7]], nullptr, nullptr);
        } else if (k < 8) {
            const int* n8 = E[q];
            const Padded ctl = corner_tl(f[n8[0]], f[n8[2]], p), cbr = corner_br(f[n8[4]], f[n8[6]], p);
            P = pad_face(EQUATOR, nside, p, f[k], f[n8[0]], f[k], f[n8[2]], f[n8[3]], f[n8[4]], f[k], f[n8[6]], f[n8[7]], &ctl, &cbr);
        } else {
            const int* n8 = S[q];
            P = pad_face(SOUTH, nside, p, f[k], f[n8[0]], f[n8[1]], f[n8[2]], f[n8[3]], f[n8[4]], f[n8[5]], f[n8[6]], f[n8[7]], nullptr, nullptr);
        }
        for (size_t c = 0; c < (size_t)m * m; ++c) {
            idx_a_host[(size_t)k * m * m + c] = P.a[c];
            idx_b_host[(size_t)k * m * m + c] = P.b[c];
        }
    }
    return ACE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// device kernels
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// one atomicMax per workgroup into the 64-shard bound slot (a wave-level atomic per 64 cells serialises at the L2: 0.35 ms per call)
__device__ __forceinline__ void block_amax(float vmax, unsigned* __restrict__ amax) {
    __shared__ float wmax[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// y[item][face][c0 + c][row][col] (padded: m rows of pitch mp >= m, gap columns zero) = 0.5 x[a] + 0.5 x[b];
// x: [item * 12 + face][c][rows][x_pitch].  amax (optional): 64-shard atomicMax of bits(max|y|).  The last workgroup also
// zeroes `slack` floats behind the tensor (the k x k contraction reads (k - 1) dil elements past the last row).
__global__ __launch_bounds__(256) void hpx_pad_kernel(const float* __restrict__ x, long x_img_stride, long x_chan_stride, int x_pitch,
                                                      float* __restrict__ y, int y_chans, int c0, int c, int m, int mp,
                                                      const int* __restrict__ ia, const int* __restrict__ ib, int items,
                                                      unsigned* __restrict__ amax, int slack) {
    const long cells = (long)m * mp;
    const long total = (long)items * 12 * c * cells;
    float vmax = 0.f;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cellp = (int)(t % cells);
        const int row = cellp / mp, col = cellp % mp;
        long q = t / cells;
        const int ch = (int)(q % c);
        q /= c;
        const int face = (int)(q % 12), item = (int)(q / 12);
        float v = 0.f;
        if (col < m) {
            const long cell = (long)row * m + col;
            const int a = ia[(long)face * m * m + cell], b = ib[(long)face * m * m + cell];
            auto src = [&](int s) {
                const int sf = s >> 24, sy = (s >> 12) & 4095, sx = s & 4095;
                return x[(long)(item * 12 + sf) * x_img_stride + (long)ch * x_chan_stride + (long)sy * x_pitch + sx];
            };
            const float va = src(a);
            v = a == b ? va : 0.5f * va + 0.5f * src(b);
        }
        y[((long)(item * 12 + face) * y_chans + c0 + ch) * cells + cellp] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    if (slack > 0 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < slack && c0 + c == y_chans)
        y[(long)items * 12 * y_chans * cells + threadIdx.x] = 0.f;
    if (amax) block_amax(vmax, amax);
}

// The same gather, written as the PACKED convolution engine's operand: fp16 hi / lo "P-format" planes [item * 12 + face][cg][cell][8]
// (cell = padded row * mp + column; entry = channels 8 cg .. 8 cg + 7 of one cell: the MFMA B fragment of a lane), scaled by
// 2^(12 - exponent(bound)) with bound = max(bound of x, bound of x2) - padding copies / averages cells, it cannot exceed its sources -
// published to pmax.  Channels [0, cin) come from x, [cin, cin + cin2) from x2 (the skip concatenation), the rest of the last
// group is zero.  `slack` zero entries behind every plane pair (the last taps of the last row read past the end).
__global__ __launch_bounds__(256) void hpx_pad_planes_kernel(const float* __restrict__ x, long x_img_stride, long x_chan_stride, int x_pitch,
                                                             const float* __restrict__ x2, long x2_img_stride, long x2_chan_stride, int x2_pitch,
                                                             int cin, int cin2, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int m, int mp,
                                                             const int* __restrict__ ia, const int* __restrict__ ib, int items,
                                                             const unsigned* __restrict__ xmax, const unsigned* __restrict__ x2max,
                                                             unsigned* __restrict__ pmax, int slack) {
    const int lane = threadIdx.x & 63;
    float bound = wave_max_bits(slot_load(xmax + lane));
    if (x2max) bound = fmaxf(bound, wave_max_bits(slot_load(x2max + lane)));
    const float scale = ldexpf(1.0f, pow2_exponent_for(bound));
    if (blockIdx.x == 0 && threadIdx.x < 64) pmax[threadIdx.x] = __float_as_uint(bound);
    const int ctot = cin + cin2, cg8 = (ctot + 7) / 8;
    const long cells = (long)m * mp;
    const long total = (long)items * 12 * cg8 * cells;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cellp = (int)(t % cells);
        const int row = cellp / mp, col = cellp % mp;
        long q = t / cells;
        const int cg = (int)(q % cg8);
        q /= cg8;
        const int face = (int)(q % 12), item = (int)(q / 12);
        half8 hh, ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hh[e] = (_Float16)0.f; ll[e] = (_Float16)0.f; }
        if (col < m) {
            const long cell = (long)row * m + col;
            const int a = ia[(long)face * m * m + cell], b = ib[(long)face * m * m + cell];
            const int af = a >> 24, ay = (a >> 12) & 4095, ax = a & 4095, bf = b >> 24, by = (b >> 12) & 4095, bx = b & 4095;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = 8 * cg + e;
                if (ch >= ctot) break;
                const bool second = ch >= cin;
                const float* src = second ? x2 : x;
                const long is = second ? x2_img_stride : x_img_stride, cs = second ? x2_chan_stride : x_chan_stride;
                const int pt = second ? x2_pitch : x_pitch, cc = second ? ch - cin : ch;
                const float va = src[(long)(item * 12 + af) * is + (long)cc * cs + (long)ay * pt + ax];
                float v = va;
                if (a != b) v = 0.5f * va + 0.5f * src[(long)(item * 12 + bf) * is + (long)cc * cs + (long)by * pt + bx];
                v = __builtin_amdgcn_fmed3f(v * scale, -65504.f, 65504.f);
                const _Float16 h = (_Float16)v;
                hh[e] = h;
                ll[e] = (_Float16)(v - (float)h);
            }
        }
        const long eo = (((long)(item * 12 + face) * cg8 + cg) * cells + cellp) * 8;
        *reinterpret_cast<half8*>(hi + eo) = hh;
        *reinterpret_cast<half8*>(lo + eo) = ll;
    }
    if (slack > 0 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < slack) {
        half8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
        const long eo = ((long)items * 12 * cg8 * cells + threadIdx.x) * 8;
        *reinterpret_cast<half8*>(hi + eo) = z;
        *reinterpret_cast<half8*>(lo + eo) = z;
    }
}

// Face padding IN PLACE on planes whose interior a convolution's epilogue has just written (ace_hpx_conv_packed with the padded
// plane pitch and the interior's shift): every halo cell is gathered from interior cells of the neighbouring faces' planes (same
// channel group, same scale: copies of 16-byte entries; the two-source corner cells are averaged and re-split), the gap columns
// [m, mp) and the slack entries are zeroed (the epilogue's gap-column results spilled into them).  Reads interiors only, writes
// everything else: no ordering between threads matters.
__global__ __launch_bounds__(256) void hpx_halo_planes_kernel(_Float16* __restrict__ hi, _Float16* __restrict__ lo, int cg8, int nside, int p,
                                                              int mp, const int* __restrict__ ia, const int* __restrict__ ib, int items, int slack) {
    const int m = nside + 2 * p;
    const long cells = (long)m * mp;
    const long total = (long)items * 12 * cg8 * cells;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cellp = (int)(t % cells);
        const int row = cellp / mp, col = cellp % mp;
        if (row >= p && row < p + nside && col >= p && col < p + nside) continue;   // interior: the convolution's own result
        long q = t / cells;
        const int cg = (int)(q % cg8);
        q /= cg8;
        const int face = (int)(q % 12), item = (int)(q / 12);
        half8 hh, ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hh[e] = (_Float16)0.f; ll[e] = (_Float16)0.f; }
        if (col < m) {
            const long cell = (long)row * m + col;
            const int a = ia[(long)face * m * m + cell], b = ib[(long)face * m * m + cell];
            auto entry = [&](int s_) { return (((long)(item * 12 + (s_ >> 24)) * cg8 + cg) * cells + (long)(((s_ >> 12) & 4095) + p) * mp + (s_ & 4095) + p) * 8; };
            const long ea = entry(a);
            hh = *reinterpret_cast<const half8*>(hi + ea);
            ll = *reinterpret_cast<const half8*>(lo + ea);
            if (a != b) {
                const long eb = entry(b);
                const half8 h2 = *reinterpret_cast<const half8*>(hi + eb), l2 = *reinterpret_cast<const half8*>(lo + eb);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = 0.5f * ((float)hh[e] + (float)ll[e]) + 0.5f * ((float)h2[e] + (float)l2[e]);
                    const _Float16 h = (_Float16)v;
                    hh[e] = h;
                    ll[e] = (_Float16)(v - (float)h);
                }
            }
        }
        const long eo = (((long)(item * 12 + face) * cg8 + cg) * cells + cellp) * 8;
        *reinterpret_cast<half8*>(hi + eo) = hh;
        *reinterpret_cast<half8*>(lo + eo) = ll;
    }
    if (slack > 0 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < slack) {
        half8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
        const long eo = ((long)items * 12 * cg8 * cells + threadIdx.x) * 8;
        *reinterpret_cast<half8*>(hi + eo) = z;
        *reinterpret_cast<half8*>(lo + eo) = z;
    }
}

// 2 x 2 pooling, stride 2: x [imgs * c][H][px] -> y [imgs * c][H / 2][py]
template <bool MAX>
__global__ __launch_bounds__(256) void hpx_pool2_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W,
                                                        int px, long sx, int py, long sy) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = planes * Ho * Wo;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int xo = (int)(t % Wo), yo = (int)((t / Wo) % Ho);
        const long pl = t / ((long)Wo * Ho);
        const float* s = x + pl * sx + (long)(2 * yo) * px + 2 * xo;
        const float a = s[0], b = s[1], c = s[px], d = s[px + 1];
        y[pl * sy + (long)yo * py + xo] = MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (((a + b) + c) + d) * 0.25f;
    }
}

// nn.Upsample(scale_factor = 2) of every (image, channel) plane (healpix_blocks.py:197-252, 699-759: the "Interpolate" block and the
// resize of SmoothedInterpolate): mode 0 "nearest" (every cell a 2 x 2 block of itself), 1 "bilinear" with torch's source index
// (align_corners false: max(0.5 (o + 0.5) - 0.5, 0); true: o (n - 1) / (2 n - 1)), neighbours clamped to the plane.  Both are convex
// combinations of the input: the input's bound is the output's.
template <int MODE>
__global__ __launch_bounds__(256) void hpx_upsample2_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W,
                                                            int px, long sx, int py, long sy, int align_corners) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = planes * Ho * Wo;
    const float ry = (align_corners && Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.5f;
    const float rx = (align_corners && Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.5f;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int xo = (int)(t % Wo), yo = (int)((t / Wo) % Ho);
        const long pl = t / ((long)Wo * Ho);
        const float* s = x + pl * sx;
        float v;
        if (MODE == 0) {
            v = s[(long)(yo >> 1) * px + (xo >> 1)];
        } else {
            const float fy = align_corners ? ry * (float)yo : fmaxf(ry * ((float)yo + 0.5f) - 0.5f, 0.f);
            const float fx = align_corners ? rx * (float)xo : fmaxf(rx * ((float)xo + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            v = ly0 * (lx0 * s[(long)y0 * px + x0] + lx1 * s[(long)y0 * px + x1]) + ly1 * (lx0 * s[(long)y1 * px + x0] + lx1 * s[(long)y1 * px + x1]);
        }
        y[pl * sy + (long)yo * py + xo] = v;
    }
}

__device__ __forceinline__ float hpx_act(float v, int act, float cap) {
    if (act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    else if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
    return fminf(v, cap);
}
// interleave the four (dy, dx) row blocks of a 2 x 2 stride-2 transposed convolution's GEMM, add the bias, activate:
// t [imgs][4][cout][H][pin] -> y [imgs][cout][2 H][pout];  amax: bound of the result
__global__ __launch_bounds__(256) void hpx_tconv_scatter_kernel(const float* __restrict__ t, const float* __restrict__ bias,
                                                                float* __restrict__ y, long imgs, int cout, int H, int W, int pin,
                                                                int pout, long s_out_plane, int act, float cap,
                                                                unsigned* __restrict__ amax) {
    const long total = imgs * cout * (long)(2 * H) * (2 * W);
    float vmax = 0.f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int xo = (int)(q % (2 * W)), yo = (int)((q / (2 * W)) % (2 * H));
        const long pl = q / ((long)4 * W * H);       // image * cout + channel
        const long img = pl / cout;
        const int o = (int)(pl % cout);
        const int tap = (yo & 1) * 2 + (xo & 1);
        float v = t[((img * 4 + tap) * cout + o) * (long)H * pin + (long)(yo >> 1) * pin + (xo >> 1)];
        v += bias ? bias[o] : 0.f;
        v = hpx_act(v, act, cap);
        y[pl * s_out_plane + (long)yo * pout + xo] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    if (amax) block_amax(vmax, amax);
}

unsigned grid_for(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));   // grid-stride kernels: 16 workgroups per CU are plenty
}

}  // namespace

extern "C" int ace_hpx_pad(const float* x, long x_img_stride, long x_chan_stride, int x_pitch, float* y, int y_chans, int c0, int c,
                           const int* idx_a_dev, const int* idx_b_dev, int items, int nside, int p, int y_pitch, unsigned* amax,
                           void* stream) {
    const int m = nside + 2 * p;
    if (!x || !y || !idx_a_dev || !idx_b_dev || items < 1 || c < 1 || c0 < 0 || c0 + c > y_chans || nside < 1 || p < 1 || y_pitch < m)
        return hfail(ACE_ERR_INVALID, "ace_hpx_pad: bad argument");
    const long total = (long)items * 12 * c * m * y_pitch;
    hipLaunchKernelGGL(hpx_pad_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, x_img_stride,
                       x_chan_stride, x_pitch, y, y_chans, c0, c, m, y_pitch, idx_a_dev, idx_b_dev, items, amax, ACE_HPX_SLACK);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}

extern "C" int ace_hpx_absmax(const float* x, long n, unsigned* amax, void* stream) {
    if (!x || n < 1 || !amax) return hfail(ACE_ERR_INVALID, "ace_hpx_absmax: bad argument");
    HPX_TRY(launch_absmax(x, n, amax, static_cast<hipStream_t>(stream)));
    return ACE_OK;
}

// ---- prepared weights: fp16 hi/lo planes [rows16][pitch] scaled by a power of two (what ace_sfno_set_weight does per parameter)
struct ace_hpx_weight {
    _Float16* hi = nullptr; _Float16* lo = nullptr;
    _Float16* thi = nullptr; _Float16* tlo = nullptr;   // the same planes in the packed engine's A-tile order (ace_hpx_conv_packed)
    int rows = 0, cols = 0, pitch = 0;
    float ascale = 1.f;
    float winf = 0.f;   // max row sum of |w|: |W x| <= winf max|x| (the bound a P-format output is scaled by)
};
extern "C" void ace_hpx_weight_destroy(ace_hpx_weight* w) {
    if (!w) return;
    if (w->hi) (void)hipFree(w->hi);
    if (w->lo) (void)hipFree(w->lo);
    if (w->thi) (void)hipFree(w->thi);
    if (w->tlo) (void)hipFree(w->tlo);
    delete w;
}
extern "C" int ace_hpx_weight_create(const float* w_dev, int rows, int cols, void* stream, ace_hpx_weight** out) {
    if (!w_dev || rows < 1 || cols < 1 || !out) return hfail(ACE_ERR_INVALID, "ace_hpx_weight_create: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<float> host((size_t)rows * cols);
    HPX_TRY(hipStreamSynchronize(s));
    HPX_TRY(hipMemcpy(host.data(), w_dev, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (float v : host) {
        if (!std::isfinite(v)) return hfail(ACE_ERR_INVALID, "ace_hpx_weight_create: non-finite weight");
        mx = std::max(mx, std::fabs(v));
    }
    int e = 0;
    if (mx > 0.f) { (void)std::frexp(mx, &e); e = 10 - e; }
    ace_hpx_weight* w = new ace_hpx_weight;
    w->rows = rows; w->cols = cols; w->pitch = (cols + 31) & ~31;
    w->ascale = std::ldexp(1.0f, e);
    for (int r = 0; r < rows; ++r) {
        double rs = 0.0;
        for (int c = 0; c < cols; ++c) rs += std::fabs((double)host[(size_t)r * cols + c]);
        w->winf = std::max(w->winf, (float)(rs * (1.0 + 1e-6)));
    }
    const size_t halves = (size_t)((rows + 15) / 16 * 16) * w->pitch;
    if (hipMalloc(reinterpret_cast<void**>(&w->hi), halves * 2) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&w->lo), halves * 2) != hipSuccess) {
        ace_hpx_weight_destroy(w);
        return hfail(ACE_ERR_RUNTIME, "ace_hpx_weight_create: out of device memory");
    }
    hipError_t er = hipMemsetAsync(w->hi, 0, halves * 2, s);
    if (er == hipSuccess) er = hipMemsetAsync(w->lo, 0, halves * 2, s);
    if (er == hipSuccess) er = launch_split_f16(w_dev, cols, w->hi, w->lo, w->pitch, rows, cols, w->ascale, s);
    if (er == hipSuccess && cols % 8 == 0) {   // a (tap, padded channel) ordered matrix: also as tiles for the packed engine
        if (hipMalloc(reinterpret_cast<void**>(&w->thi), halves * 2) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&w->tlo), halves * 2) != hipSuccess) {
            ace_hpx_weight_destroy(w);
            return hfail(ACE_ERR_RUNTIME, "ace_hpx_weight_create: out of device memory");
        }
        er = hipMemsetAsync(w->thi, 0, halves * 2, s);
        if (er == hipSuccess) er = hipMemsetAsync(w->tlo, 0, halves * 2, s);
        if (er == hipSuccess) er = launch_split_f16_tiled(w_dev, cols, w->thi, w->tlo, w->pitch, rows, cols, w->ascale, s);
    }
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (er != hipSuccess) { ace_hpx_weight_destroy(w); return hfail(ACE_ERR_RUNTIME, std::string("ace_hpx_weight_create: ") + hipGetErrorString(er)); }
    *out = w;
    return ACE_OK;
}

// y[img][o][r][c] = act( bias[o] + sum_{ky, kx, i} w[o][(ky k + kx) cin + i] x[img][i][r + ky dil][c + kx dil] (+ R[img][o][r][c]) ), capped.
// x: [imgs][cin][rows_in][pitch] with rows_in = H + (k - 1) dil (+ ACE_HPX_SLACK floats behind it when k > 1); k = 1 may take a
// second source x2 [imgs][cin2][H][pitch] (rows cin .. of the contraction) and a residual R; y / R: [imgs][cout][H][pitch].
// row_off (k > 1): k k cin element offsets (ky dil pitch + kx dil + i rows_in pitch), device.  xmax / x2max: bound slots of the
// sources; ymax (optional): bound slot of the result (zeroed by the caller).  pitch % 4 == 0, all bases 16-byte aligned.
extern "C" int ace_hpx_conv(const float* x, const float* x2, int cin, int cin2, const ace_hpx_weight* w, const long* row_off,
                            const float* bias, const float* R, float* y, int imgs, int cout, int H, int W, int pitch, int k, int dil,
                            int act, float cap, const unsigned* xmax, const unsigned* x2max, unsigned* ymax, void* stream) {
    if (!x || !w || !y || !xmax || imgs < 1 || cin < 1 || cout < 1 || H < 1 || W < 1 || pitch < W + (k - 1) * dil || (pitch & 3) || k < 1 ||
        dil < 1 || cin2 < 0 || (cin2 > 0 && (!x2 || !x2max || k != 1)) || (R && k != 1) || (k > 1 && !row_off))
        return hfail(ACE_ERR_INVALID, "ace_hpx_conv: bad argument (two sources and a residual only with k = 1; pitch % 4 == 0)");
    if (!(act == ACT_NONE || act == ACT_GELU || act == ACT_RELU)) return hfail(ACE_ERR_INVALID, "ace_hpx_conv: activation must be none, gelu or relu");
    const int K = (cin + cin2) * k * k;
    if (w->rows != cout || w->cols != K) return hfail(ACE_ERR_INVALID, "ace_hpx_conv: prepared weight is " + std::to_string(w->rows) + " x " +
                                                      std::to_string(w->cols) + ", expected " + std::to_string(cout) + " x " + std::to_string(K));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows_in = H + (k - 1) * dil;
    GemmArgs g;
    g.lda = w->pitch; g.sA = 0; g.a_kpad = w->pitch;
    g.B = x; g.ldb = (long)rows_in * pitch; g.sB = (long)cin * rows_in * pitch;
    if (k > 1) g.brow = row_off;
    if (cin2 > 0) { g.B2 = x2; g.ldb2 = (long)H * pitch; g.sB2 = (long)cin2 * H * pitch; g.K1 = cin; }
    g.C = y; g.ldc = (long)H * pitch; g.sC = (long)cout * H * pitch;
    g.bias = bias;
    if (R) { g.R = R; g.ldr = g.ldc; g.sR = g.sC; }
    g.M = cout; g.N = H * pitch; g.K = K; g.nbatch = imgs;
    g.act = act; g.cap = cap;
    if (!gemm_f16x3_eligible(g)) return hfail(ACE_ERR_INVALID, "ace_hpx_conv: operands must be 16-byte aligned");
    HPX_TRY(launch_gemm_f16x3(g, w->hi, w->lo, w->ascale, 1.0f, s, xmax, ymax, cin2 > 0 ? x2max : nullptr));
    return ACE_OK;
}

// The k x k convolution on the packed-operand engine (kernels.hip gemm4, implicit-GEMM form): the padded input comes as P-format
// planes from ace_hpx_pad_planes (cpad = 8 ceil(channels / 8) channels per image, `pitch` entries per padded row, H + (k - 1) dil rows),
// the weight as a prepared (cout x k k cpad) matrix in (tap, padded channel) order (zero columns for the padding channels).
// Both operands stream by LDS-DMA; no fp32 activation is split in the kernel (gemm3 spends its loader waves on that).
// x_plane_cells / y_plane_cells (0: the natural (H + (k - 1) dil) pitch / H pitch): entries per channel-group plane when the planes are
// those of a LARGER padded tensor and xhi / yhi point at a shifted origin inside it - a 1 x 1 convolution reading the interior of
// planes padded for a k x k one (k = 1, origin at (p, p)), or a result written into the interior of the next convolution's padded
// planes (ace_hpx_halo_planes then fills the halo).
extern "C" int ace_hpx_conv_packed(const void* xhi, const void* xlo, int cpad, long x_plane_cells, const ace_hpx_weight* w, const float* bias,
                                   float bias_max, float* y, void* yhi, void* ylo, long y_plane_cells, int imgs, int cout, int H, int W, int pitch,
                                   int k, int dil, int act, float cap, const unsigned* pmax, unsigned* ymax, void* stream) {
    if (!xhi || !xlo || !w || (!y && !yhi) || (yhi && (!ylo || !ymax || (cout & 7))) || !pmax || imgs < 1 || cpad < 8 || (cpad & 7) || cout < 1 ||
        H < 1 || W < 1 || pitch < W + (k - 1) * dil || (pitch & 3) || k < 1 || dil < 1 || (k - 1) * dil > ACE_HPX_SLACK || !(bias_max >= 0.f) ||
        x_plane_cells < 0 || y_plane_cells < 0 || (x_plane_cells > 0 && x_plane_cells < (long)(H + (k - 1) * dil) * pitch) ||
        (y_plane_cells > 0 && y_plane_cells < (long)H * pitch))
        return hfail(ACE_ERR_INVALID, "ace_hpx_conv_packed: bad argument (channels padded to 8, pitch % 4 == 0; planes out: cout % 8 == 0 and a slot)");
    if (!(act == ACT_NONE || act == ACT_GELU || act == ACT_RELU)) return hfail(ACE_ERR_INVALID, "ace_hpx_conv_packed: activation must be none, gelu or relu");
    const int K = cpad * k * k;
    if (w->rows != cout || w->cols != K || !w->thi)
        return hfail(ACE_ERR_INVALID, "ace_hpx_conv_packed: prepared weight is " + std::to_string(w->rows) + " x " + std::to_string(w->cols) +
                                          ", expected " + std::to_string(cout) + " x " + std::to_string(K) + " (tap, padded channel) columns");
    const int rows_in = H + (k - 1) * dil;
    const long cells = x_plane_cells > 0 ? x_plane_cells : (long)rows_in * pitch;   // entries per channel-group plane of x
    Gemm4Args a;
    a.Ahi = w->thi; a.Alo = w->tlo; a.lda = w->pitch; a.sA = 0; a.ascale = w->ascale; a.a_tiled = 1;
    a.Bhi = static_cast<const _Float16*>(xhi); a.Blo = static_cast<const _Float16*>(xlo);
    a.ldn = cells; a.sB = (long)(cpad / 8) * cells * 8; a.bmax = pmax;
    if (yhi) {   // P-format output: the operand of a following convolution; its bound goes to ymax
        const long ycells = y_plane_cells > 0 ? y_plane_cells : (long)H * pitch;
        a.Chi = static_cast<_Float16*>(yhi); a.Clo = static_cast<_Float16*>(ylo); a.ldnc = ycells; a.sCp = (long)(cout / 8) * ycells * 8;
        a.cw = w->winf; a.cb = bias_max; a.cslot = ymax;
    }
    if (y) { a.C = y; a.ldc = (long)H * pitch; a.sC = (long)cout * H * pitch; if (!yhi) a.omax = ymax; }
    a.bias = bias; a.sbias = 0;
    a.M = cout; a.N = H * pitch; a.K = K; a.nbatch = imgs;
    a.act = act; a.cap = cap;
    a.impl_k = k; a.impl_cg8 = cpad / 8; a.impl_pitch = pitch; a.impl_dil = dil;
    // 128 x 128 tiles unless they waste over 20 % more rows than 64 x 256 ones: the launcher's own rule (least padding) picks 64-row
    // tiles for cout = 544 (576 against 640 rows), the 128-row tile is the faster engine (same box, nside 64: 4.47 -> 4.08 ms per forward)
    if (5 * ((cout + 127) / 128 * 128) <= 6 * ((cout + 63) / 64 * 64)) a.tile = 1;
    HPX_TRY(launch_gemm_f16x3_packed(a, static_cast<hipStream_t>(stream)));
    return ACE_OK;
}

// 1 x 1 convolution (+ bias, + residual, activation none / GELU / ReLU without a cap) whose input is the P-format output of
// ace_hpx_conv_packed: x planes [imgs][cin / 8][H pitch][8] with bound slot xslot; R / y: [imgs][cout][H][pitch] fp32.
extern "C" int ace_hpx_conv1_packed(const void* xhi, const void* xlo, int cin, const ace_hpx_weight* w, const float* bias, const float* R, float* y,
                                    int imgs, int cout, int H, int W, int pitch, int act, const unsigned* xslot, unsigned* ymax, void* stream) {
    if (!xhi || !xlo || !w || !y || !xslot || imgs < 1 || cin < 8 || (cin & 7) || cout < 1 || H < 1 || W < 1 || pitch < W || (pitch & 3))
        return hfail(ACE_ERR_INVALID, "ace_hpx_conv1_packed: bad argument (cin % 8 == 0, pitch % 4 == 0)");
    if (!(act == ACT_NONE || act == ACT_GELU || act == ACT_RELU)) return hfail(ACE_ERR_INVALID, "ace_hpx_conv1_packed: activation must be none, gelu or relu");
    if (w->rows != cout || w->cols != cin || !w->thi)
        return hfail(ACE_ERR_INVALID, "ace_hpx_conv1_packed: prepared weight is " + std::to_string(w->rows) + " x " + std::to_string(w->cols) +
                                          ", expected " + std::to_string(cout) + " x " + std::to_string(cin));
    const long N = (long)H * pitch;
    Gemm4Args a;
    a.Ahi = w->thi; a.Alo = w->tlo; a.lda = w->pitch; a.sA = 0; a.ascale = w->ascale; a.a_tiled = 1;
    a.Bhi = static_cast<const _Float16*>(xhi); a.Blo = static_cast<const _Float16*>(xlo); a.ldn = N; a.sB = (long)cin * N; a.bmax = xslot;
    a.C = y; a.ldc = N; a.sC = (long)cout * N; a.omax = ymax;
    a.bias = bias; a.sbias = 0;
    if (R) { a.R = R; a.ldr = N; a.sR = (long)cout * N; }
    a.M = cout; a.N = (int)N; a.K = cin; a.nbatch = imgs;
    a.act = act;
    HPX_TRY(launch_gemm_f16x3_packed(a, static_cast<hipStream_t>(stream)));
    return ACE_OK;
}

extern "C" int ace_hpx_pad_planes(const float* x, long x_img_stride, long x_chan_stride, int x_pitch, const float* x2, long x2_img_stride,
                                  long x2_chan_stride, int x2_pitch, int cin, int cin2, void* hi, void* lo, const int* idx_a_dev,
                                  const int* idx_b_dev, int items, int nside, int p, int y_pitch, const unsigned* xmax, const unsigned* x2max,
                                  unsigned* pmax, void* stream) {
    const int m = nside + 2 * p;
    if (!x || !hi || !lo || !idx_a_dev || !idx_b_dev || items < 1 || cin < 1 || cin2 < 0 || (cin2 > 0 && (!x2 || !x2max)) || nside < 1 || p < 1 ||
        y_pitch < m || !xmax || !pmax)
        return hfail(ACE_ERR_INVALID, "ace_hpx_pad_planes: bad argument");
    const long total = (long)items * 12 * ((cin + cin2 + 7) / 8) * m * y_pitch;
    hipLaunchKernelGGL(hpx_pad_planes_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, x_img_stride, x_chan_stride,
                       x_pitch, cin2 > 0 ? x2 : x, x2_img_stride, x2_chan_stride, x2_pitch, cin, cin2, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), m, y_pitch, idx_a_dev, idx_b_dev, items, xmax, cin2 > 0 ? x2max : nullptr, pmax, ACE_HPX_SLACK);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}

extern "C" int ace_hpx_halo_planes(void* hi, void* lo, int cpad, const int* idx_a_dev, const int* idx_b_dev, int items, int nside, int p, int y_pitch,
                                   void* stream) {
    const int m = nside + 2 * p;
    if (!hi || !lo || !idx_a_dev || !idx_b_dev || items < 1 || cpad < 8 || (cpad & 7) || nside < 1 || p < 1 || y_pitch < m)
        return hfail(ACE_ERR_INVALID, "ace_hpx_halo_planes: bad argument");
    const long total = (long)items * 12 * (cpad / 8) * m * y_pitch;
    hipLaunchKernelGGL(hpx_halo_planes_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), cpad / 8, nside, p, y_pitch, idx_a_dev, idx_b_dev, items, ACE_HPX_SLACK);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}

extern "C" int ace_hpx_pool2(const float* x, float* y, long planes, int H, int W, int pitch_in, long plane_stride_in, int pitch_out,
                             long plane_stride_out, int is_max, void* stream) {
    if (!x || !y || planes < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return hfail(ACE_ERR_INVALID, "ace_hpx_pool2: bad argument");
    const long total = planes * (H / 2) * (W / 2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (is_max) hipLaunchKernelGGL(hpx_pool2_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, x, y, planes, H, W, pitch_in, plane_stride_in, pitch_out, plane_stride_out);
    else hipLaunchKernelGGL(hpx_pool2_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, x, y, planes, H, W, pitch_in, plane_stride_in, pitch_out, plane_stride_out);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}

extern "C" int ace_hpx_upsample2(const float* x, float* y, long planes, int H, int W, int pitch_in, long plane_stride_in, int pitch_out,
                                 long plane_stride_out, int mode, int align_corners, void* stream) {
    if (!x || !y || planes < 1 || H < 1 || W < 1 || (mode != 0 && mode != 1)) return hfail(ACE_ERR_INVALID, "ace_hpx_upsample2: bad argument");
    const long total = planes * (2L * H) * (2L * W);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 0) hipLaunchKernelGGL(hpx_upsample2_kernel<0>, dim3(grid_for(total)), dim3(256), 0, s, x, y, planes, H, W, pitch_in, plane_stride_in, pitch_out, plane_stride_out, 0);
    else hipLaunchKernelGGL(hpx_upsample2_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, x, y, planes, H, W, pitch_in, plane_stride_in, pitch_out, plane_stride_out, align_corners);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}

// nn.ConvTranspose2d(cin, cout, 2, stride 2) + activation: w prepared from [(dy, dx, cout)][cin] (4 cout rows); tmp: imgs * 4 cout * H *
// pitch_in floats; x: [imgs][cin][H][pitch_in] with bound slot xmax; ymax (optional, zeroed by the caller): bound of the result
extern "C" int ace_hpx_tconv2(const float* x, const ace_hpx_weight* w, const float* bias, float* tmp, float* y, int imgs, int cin, int cout,
                              int H, int W, int pitch_in, int pitch_out, long plane_stride_out, int act, float cap, const unsigned* xmax,
                              unsigned* ymax, void* stream) {
    if (!x || !w || !tmp || !y || !xmax || imgs < 1 || cin < 1 || cout < 1 || H < 1 || W < 1 || (pitch_in & 3))
        return hfail(ACE_ERR_INVALID, "ace_hpx_tconv2: bad argument");
    if (w->rows != 4 * cout || w->cols != cin) return hfail(ACE_ERR_INVALID, "ace_hpx_tconv2: prepared weight has the wrong shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GemmArgs g;
    g.lda = w->pitch; g.sA = 0; g.a_kpad = w->pitch;
    g.B = x; g.ldb = (long)H * pitch_in; g.sB = (long)cin * H * pitch_in;
    g.C = tmp; g.ldc = (long)H * pitch_in; g.sC = (long)4 * cout * H * pitch_in;
    g.M = 4 * cout; g.N = H * pitch_in; g.K = cin; g.nbatch = imgs;
    if (!gemm_f16x3_eligible(g)) return hfail(ACE_ERR_INVALID, "ace_hpx_tconv2: operands must be 16-byte aligned");
    HPX_TRY(launch_gemm_f16x3(g, w->hi, w->lo, w->ascale, 1.0f, s, xmax, nullptr, nullptr));
    const long total = (long)imgs * cout * 4 * H * W;
    hipLaunchKernelGGL(hpx_tconv_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, s, tmp, bias, y, (long)imgs, cout, H, W, pitch_in,
                       pitch_out, plane_stride_out, act, cap, ymax);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}
