// HEALPix variant (BASELINE configs[4]): the operators of the reference's HEALPix UNet
// (fme/ace/models/healpix/) as gfx950 kernels behind the C ABI - neighbourhood convolutions on the 12-face mesh in place of the
// spherical harmonic transform, same stepper API.
//   * face padding (healpix_paddings.py:239-611, Karlbauer et al. 2024; the reference states that its default "earth2grid"
//     backend gives the same result): every halo cell of a padded face is ONE cell of a neighbouring face (rotated for the
//     polar faces) or, on the diagonal of the two corners an equatorial face has no neighbour for, the mean of two.  Built
//     once on the host as a gather table (ace_hpx_pad_table_host), applied by one gather kernel;
//   * k x k (dilated) convolution of a padded face = k^2 accumulated GEMMs W_tap (Cout x Cin) . X shifted by the tap's
//     constant offset: activations of one UNet level are stored as [channels][rows][P] with the row pitch P = W + 2 p of that
//     level's padded faces, so a shifted view of the padded tensor IS a plain row-major operand of the fp32 MFMA engine
//     (kernels.hip: exact fp32 on v_mfma_f32_32x32x2_f32; bias on the first tap, residual / capped GELU on the last);
//   * 2 x 2 average / max pooling, 2 x 2 stride-2 transposed convolution (four GEMMs + an interleaving scatter with the
//     bias and activation), channel concatenation by writing two sources into one padded tensor.
// First version: correct and on the matrix cores, not yet tuned (k^2 passes over the output per convolution).
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/ace_sfno.h"
#include "kernels.h"

using namespace ace;

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

// y[item][face][c0 + c][row][col] (padded, compact m x m) = 0.5 x[a] + 0.5 x[b];  x: [item * 12 + face][c][rows][x_pitch]
__global__ __launch_bounds__(256) void hpx_pad_kernel(const float* __restrict__ x, long x_img_stride, long x_chan_stride, int x_pitch,
                                                      float* __restrict__ y, int y_chans, int c0, int c, int m,
                                                      const int* __restrict__ ia, const int* __restrict__ ib, int items) {
    const long cells = (long)m * m;
    const long total = (long)items * 12 * c * cells;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cell = (int)(t % cells);
        long q = t / cells;
        const int ch = (int)(q % c);
        q /= c;
        const int face = (int)(q % 12), item = (int)(q / 12);
        const int a = ia[(long)face * cells + cell], b = ib[(long)face * cells + cell];
        auto src = [&](int s) {
            const int sf = s >> 24, sy = (s >> 12) & 4095, sx = s & 4095;
            return x[(long)(item * 12 + sf) * x_img_stride + (long)ch * x_chan_stride + (long)sy * x_pitch + sx];
        };
        const float va = src(a);
        const float v = a == b ? va : 0.5f * va + 0.5f * src(b);
        y[((long)(item * 12 + face) * y_chans + c0 + ch) * cells + cell] = v;
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

__device__ __forceinline__ float hpx_act(float v, int act, float cap) {
    if (act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    else if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
    return fminf(v, cap);
}
// interleave the four (dy, dx) GEMM results of a 2 x 2 stride-2 transposed convolution, add the bias, activate:
// t [4][imgs][cout][H][pin] -> y [imgs][cout][2 H][pout]
__global__ __launch_bounds__(256) void hpx_tconv_scatter_kernel(const float* __restrict__ t, const float* __restrict__ bias,
                                                                float* __restrict__ y, long imgs, int cout, int H, int W, int pin,
                                                                int pout, long s_out_plane, int act, float cap) {
    const long total = imgs * cout * (long)(2 * H) * (2 * W);
    const long tap_stride = imgs * cout * (long)H * pin;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int xo = (int)(q % (2 * W)), yo = (int)((q / (2 * W)) % (2 * H));
        const long pl = q / ((long)4 * W * H);       // image * cout + channel
        const int tap = (yo & 1) * 2 + (xo & 1);
        float v = t[tap * tap_stride + pl * (long)H * pin + (long)(yo >> 1) * pin + (xo >> 1)];
        v += bias ? bias[pl % cout] : 0.f;
        y[pl * s_out_plane + (long)yo * pout + xo] = hpx_act(v, act, cap);
    }
}

unsigned grid_for(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" int ace_hpx_pad(const float* x, long x_img_stride, long x_chan_stride, int x_pitch, float* y, int y_chans, int c0, int c,
                           const int* idx_a_dev, const int* idx_b_dev, int items, int nside, int p, void* stream) {
    if (!x || !y || !idx_a_dev || !idx_b_dev || items < 1 || c < 1 || c0 < 0 || c0 + c > y_chans || nside < 1 || p < 1)
        return hfail(ACE_ERR_INVALID, "ace_hpx_pad: bad argument");
    const int m = nside + 2 * p;
    const long total = (long)items * 12 * c * m * m;
    hipLaunchKernelGGL(hpx_pad_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, x_img_stride,
                       x_chan_stride, x_pitch, y, y_chans, c0, c, m, idx_a_dev, idx_b_dev, items);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}

// y[img][o][r][c] = act( bias[o] + sum_{i, ky, kx} w[ky][kx][o][i] x[img][i][r + ky dil][c + kx dil] (+ R[img][o][r][c]) ), capped.
// x: [imgs][cin (+ cin2 from x2)][rows_in][pitch] with rows_in = H + (k - 1) dil; y / R: [imgs][cout][H][pitch].  wt: tap-major.
extern "C" int ace_hpx_conv(const float* x, const float* x2, int cin, int cin2, const float* wt, const float* bias, const float* R,
                            float* y, int imgs, int cout, int H, int W, int pitch, int k, int dil, int act, float cap, void* stream) {
    if (!x || !wt || !y || imgs < 1 || cin < 1 || cout < 1 || H < 1 || W < 1 || pitch < W + (k - 1) * dil || k < 1 || dil < 1 || cin2 < 0 ||
        (cin2 > 0 && (!x2 || k != 1)) || (R && k != 1))
        return hfail(ACE_ERR_INVALID, "ace_hpx_conv: bad argument (two sources and a residual only with k = 1)");
    if (!(act == ACT_NONE || act == ACT_GELU || act == ACT_RELU)) return hfail(ACE_ERR_INVALID, "ace_hpx_conv: activation must be none, gelu or relu");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows_in = H + (k - 1) * dil;
    const int K = cin + cin2;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const int tap = ky * k + kx;
            const bool first = tap == 0, last = tap == k * k - 1;
            GemmArgs g;
            g.A = wt + (long)tap * cout * K; g.lda = K; g.sA = 0;
            const long off = (long)ky * dil * pitch + (long)kx * dil;
            g.B = x + off; g.ldb = (long)rows_in * pitch; g.sB = (long)cin * rows_in * pitch;
            if (cin2 > 0) { g.B2 = x2; g.ldb2 = (long)rows_in * pitch; g.sB2 = (long)cin2 * rows_in * pitch; g.K1 = cin; }
            g.C = y; g.ldc = (long)H * pitch; g.sC = (long)cout * H * pitch;
            g.bias = first ? bias : nullptr;
            if (!first) { g.R = y; g.ldr = g.ldc; g.sR = g.sC; }        // accumulate over the taps, in place
            else if (R) { g.R = R; g.ldr = g.ldc; g.sR = g.sC; }
            g.M = cout; g.N = (H - 1) * pitch + W; g.K = K; g.nbatch = imgs;
            g.act = last ? act : ACT_NONE;
            g.cap = last ? cap : INFINITY;
            HPX_TRY(launch_gemm(g, s));
        }
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

// nn.ConvTranspose2d(cin, cout, 2, stride 2) + activation: wt tap-major [dy][dx][cout][cin]; tmp: 4 * imgs * cout * H * pitch_in floats
extern "C" int ace_hpx_tconv2(const float* x, const float* wt, const float* bias, float* tmp, float* y, int imgs, int cin, int cout, int H,
                              int W, int pitch_in, int pitch_out, long plane_stride_out, int act, float cap, void* stream) {
    if (!x || !wt || !tmp || !y || imgs < 1 || cin < 1 || cout < 1 || H < 1 || W < 1) return hfail(ACE_ERR_INVALID, "ace_hpx_tconv2: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long tap_stride = (long)imgs * cout * H * pitch_in;
    for (int tap = 0; tap < 4; ++tap) {
        GemmArgs g;
        g.A = wt + (long)tap * cout * cin; g.lda = cin;
        g.B = x; g.ldb = (long)H * pitch_in; g.sB = (long)cin * H * pitch_in;
        g.C = tmp + tap * tap_stride; g.ldc = (long)H * pitch_in; g.sC = (long)cout * H * pitch_in;
        g.M = cout; g.N = (H - 1) * pitch_in + W; g.K = cin; g.nbatch = imgs;
        HPX_TRY(launch_gemm(g, s));
    }
    const long total = (long)imgs * cout * 4 * H * W;
    hipLaunchKernelGGL(hpx_tconv_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, s, tmp, bias, y, (long)imgs, cout, H, W, pitch_in,
                       pitch_out, plane_stride_out, act, cap);
    HPX_TRY(hipGetLastError());
    return ACE_OK;
}
