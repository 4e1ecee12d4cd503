// Launch interface of the gfx950 kernels (kernels.hip).  Internal to the library;
// the public boundary is include/ace_sfno.h.
#pragma once
#include "tuning_guard.h"

#include <hip/hip_runtime.h>

#include <vector>

namespace ace {

// Measurement switches (environment variables, A/B timing only - every path is a gfx950 HIP kernel).  Read ONCE per handle, at
// ace_sfno_create / ace_sht_plan_create, and kept there: which kernels run never changes between weight upload and forward.
struct Switches {
    bool no_fft = false;           // ACE_NO_FFT: longitude DFT on the matrix kernels
    bool no_strip = false;         // ACE_NO_STRIP: Legendre stages on the tile engine
    bool no_fold = false;          // ACE_NO_FOLD: Legendre stages on strip.hip (all latitudes) instead of the equatorially folded strip_fold.hip
    bool no_dhconv_strip = false;  // ACE_NO_DHCONV_STRIP: spectral filter contraction on the tile engine
    bool no_pk = false;            // measurement builds only (-DACE_MEASUREMENT_SWITCHES) ACE_NO_PK: 1x1 convolutions on the on-the-fly-split engine (v3)
    bool no_pk_sht = false;        // measurement builds only: ACE_NO_PK_SHT: fp32 D, expanded filter operand
    bool no_enc_ws = false;        // ACE_NO_ENC_WS: last encoder convolution on the v3 engine
    bool no_enc_pk = false;        // ACE_NO_ENC_PK: first encoder convolution writes fp32 + a pack pass (r03) instead of planes from the packed engine
    bool no_cln_mfma = false;      // ACE_NO_CLN_MFMA: conditional layer norms as statistics + apply passes (kernels.hip) instead of cln_mfma.hip
    bool no_cln_planes = false;    // ACE_NO_CLN_PLANES: ... which write fp32 only, followed by a pack pass (instead of P-format planes from the norm kernel)
    int conv_ws_roles = 7;         // ACE_CONV_WS=skip,fc1,fc2|all|none: roles on conv_ws.hip (bit 0 inner skip, 1 fc1, 2 fc2)
    bool conv_wl = true;           // ACE_CONV_WL=0: fc1 on conv_ws.hip instead of conv_wl.hip (weights in LDS, unsynchronised waves)
    bool fused_pack = true;        // ACE_FUSED_PACK=0: the norm-folded weights of inner skip / fc1 from a pack_conv_frag launch behind every norm (r05) instead of
                                   // rider workgroups of the forward FFT (inner skip) / the convolution's own prologue (fc1 on conv_wl.hip)
    bool planes_stream = true;     // ACE_PLANES_STREAM=0: fc2 also writes the block output as fp32 (the residual stream round-trips twice)
    bool dense_grouped_filter = false;   // ACE_DENSE_GROUPED_FILTER: a grouped (block-diagonal) csfno filter expanded to the dense (C x C) form (r04)
                                         // instead of stored as the reference stores it, (G, L, C/G, C/G, 2) blocks only
};
Switches read_switches();

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SILU = 3, ACT_GELU_FAST = 4 /* erf to 1.1e-7 abs */ };
enum Tri {
    TRI_NONE = 0,
    TRI_ROWS_GE_BATCH = 1,  // forward Legendre: rows l >= m (tiles wholly below are skipped)
    TRI_K_GE_BATCH = 2,     // inverse Legendre: k (= l) starts at batch rounded down to the k-tile
    TRI_ROWS_LE_BATCH = 3   // dhconv: rows (m, b) with m <= l  => M_eff = (batch + 1) * trimul
};

// C[b] = epilogue( A[b] (M x K, row-major) * B[b] (K x N, row-major) ), batched.
// All arithmetic is exact fp32 on v_mfma_f32_32x32x2_f32 (k-ordered fma chain).
struct GemmArgs {
    const float* A = nullptr; long lda = 0; long sA = 0;
    const float* B = nullptr; long ldb = 0; long sB = 0;
    // rows k >= K1 of B come from B2 (row k - K1): the big-skip concat without a copy.  K1 < 0: unused
    const float* B2 = nullptr; long ldb2 = 0; long sB2 = 0; int K1 = -1;
    // optional (f16x3 on-the-fly-split engine only): row k of B starts at B + brow[k] instead of B + k * ldb - the k x k taps of
    // a convolution as ONE contraction over (tap, channel) without an im2col copy; rows need 4-byte alignment only
    const long* brow = nullptr;
    // per-k-row affine applied to B on load: b = b * bsc[k] + bsh[k] (fused instance norm / normalisation)
    const float* bsc = nullptr; const float* bsh = nullptr; long sbs = 0;
    float* C = nullptr; long ldc = 0; long sC = 0;
    const float* bias = nullptr; long sbias = 0;  // per row (optionally per batch)
    // number of readable columns per A row that are either data (k < K) or zero padding; the direct-to-LDS
    // engine needs a_kpad >= roundup(K, stage depth).  0 = unknown (register-staged engine is used)
    int a_kpad = 0;
    // residual added after bias: r = R[row][col] (optionally r * rsc[row] + rsh[row])
    const float* R = nullptr; long ldr = 0; long sR = 0;
    const float* rsc = nullptr; const float* rsh = nullptr; long srs = 0;
    // per-row affine applied last (fused de-normalisation): v = v * osc[row] + osh[row]
    const float* osc = nullptr; const float* osh = nullptr;
    int M = 0, N = 0, K = 0, nbatch = 1;
    int tri = TRI_NONE; int trimul = 1;
    int act = ACT_NONE;
    float cap = __builtin_inff();   // the activated value is clamped from above (HEALPix CappedGELU, healpix_activations.py:41-85)
    // optional: atomicMax of the bit pattern of max|C| over the launch (dynamic range of the f16x3 consumers);
    // must be zeroed before the launch; AMAX_SHARDS words
    unsigned* omax = nullptr;
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);

// Compensated-fp16 engine (f16x3): same contract as launch_gemm, but the A operand is given pre-split into two
// fp16 planes (hi, lo; pitch g.lda HALVES, rows zero-padded to a multiple of 32 columns: g.a_kpad), scaled by the
// power of two `ascale`; B is scaled by the power of two `bscale` on the fly.  g.A is ignored.
bool gemm_f16x3_eligible(const GemmArgs& g);
// Dynamic range: if `bmax` is given (device word holding the bit pattern of max|B|, produced by an earlier launch's
// `omax`), the kernel derives the power-of-two B scale itself and `bscale` is ignored; `omax` (optional) receives
// atomicMax(bits(max|C|)) and must be zeroed before the launch.
// Slots are arrays of AMAX_SHARDS words (producers spread their atomics, consumers reduce).
constexpr int AMAX_SHARDS = 64;
hipError_t launch_gemm_f16x3(const GemmArgs& g, const void* Ahi, const void* Alo, float ascale, float bscale,
                             hipStream_t s, const unsigned* bmax = nullptr, unsigned* omax = nullptr,
                             const unsigned* bmax2 = nullptr);
// same engine, C written as fp16 hi/lo planes in C's row-major layout (the v4 engine's A format), scaled from the bound
// cw * bound(B) which is published to cslot
hipError_t launch_gemm_f16x3_planes(const GemmArgs& g, const void* Ahi, const void* Alo, float ascale, const unsigned* bmax,
                                    void* Chi, void* Clo, float cw, unsigned* cslot, hipStream_t s);
// Mirrored roles: A = fp32 activations (row-major, split on the fly, scale from the `amax` slot), B = static operand
// pre-packed as fp16 hi/lo planes [batch][K/8][ldn][8] scaled by the power of two `bscale_static`.
hipError_t launch_gemm_f16x3_adyn(const GemmArgs& g, const void* Bhi, const void* Blo, long ldn, long sB_halves,
                                  float bscale_static, const unsigned* amax, unsigned* omax, hipStream_t s);
hipError_t launch_pack_dhconv_f16(const float* w, void* hi, void* lo, int Cin, int Cout, int L, float scale, hipStream_t s);
// NoiseConditionedSFNO filter weight (G, L, C/G, C/G, 2) -> dense (Cin, Cout, L, 2)
hipError_t launch_csfno_weight_to_dense(const float* w, float* dense, int C, int G, int L, hipStream_t s);
// compact form for Gemm4Args::cplx: planes [l][2 (re | im)][Cin/8][Cout][8]
hipError_t launch_pack_dhconv_f16c(const float* w, void* hi, void* lo, int Cin, int Cout, int L, float scale, hipStream_t s);
hipError_t launch_zero_u32(unsigned* p, long n, hipStream_t s);
// nn.LayerNorm over (H, W) with an (H, W) affine, per (sample, channel) plane (sfnonet.py:584-592); in place allowed
hipError_t launch_spatial_layer_norm(const float* x, const float* gamma, const float* beta, float eps, long planes, long HW, float* y,
                                     unsigned* omax, hipStream_t s);
// max|x| of a plain tensor into a slot
hipError_t launch_absmax(const float* x, long n, unsigned* omax, hipStream_t s);
// fp32 matrix (rows x cols, pitch lds) -> fp16 hi/lo planes (pitch ldd halves, zero padded), values scaled by `scale`
hipError_t launch_split_f16(const float* src, long lds, void* hi, void* lo, long ldd, long rows, int cols, float scale,
                            hipStream_t s);  // tests / A-B: route everything through the register-staged engine

// Compensated-fp16 engine with BOTH operands pre-split ("v4", see kernels.hip): A = row-major fp16 hi/lo planes,
// B = k-packed "P format" planes [K/8][ldn][8 halves]; optional P-format output over the rows of C.
struct Gemm4Args {
    const _Float16* Ahi = nullptr; const _Float16* Alo = nullptr; long lda = 0; long sA = 0;   // halves
    const _Float16* Bhi = nullptr; const _Float16* Blo = nullptr; long ldn = 0; long sB = 0;   // ldn entries per k group; sB halves
    int a_tiled = 0;                            // A planes are in split_f16_tiled order (lda = padded K)
    // complex-structured B (dhconv): cplx = C > 0 means K = N = 2C and B = [[Wr, Wi], [-Wi, Wr]] is stored compactly as
    // two C x C P-format blocks per batch (Wr then Wi; ldn = C, sB = 2 C C); the kernel picks the block per (k half,
    // n half) and negates the A fragments for the (-Wi) quadrant.  Needs C % 128 == 0 and the 128 x 128 tile.
    int cplx = 0;
    int tile = 0;                               // 0 = pick by M, 1 = 128 x 128, 2 = 64 x 256
    float ascale = 1.f, bscale = 1.f;           // static power-of-two scales (used when the slot pointer is null)
    const unsigned* amax = nullptr;             // dynamic operands: slot holding the bound the producer scaled with
    const unsigned* bmax = nullptr;
    float* C = nullptr; long ldc = 0; long sC = 0;                 // fp32 output (optional when PK)
    _Float16* Chi = nullptr; _Float16* Clo = nullptr; long ldnc = 0; long sCp = 0;  // P-format output over rows of C
    // PK: |C| <= cw * bound(conv input) + cb + bound(residual); published to cslot, the output scale is derived from it.
    // bound(conv input) is read from `cinb` (e.g. the bound of the NORMALISED input when the norm is folded into A)
    // or, if null, from bmax; bound(residual) from `rmax` or 0
    float cw = 0.f, cb = 0.f;
    unsigned* cslot = nullptr;
    const unsigned* cinb = nullptr;
    const unsigned* rmax = nullptr;
    // optional per-row statistics of the final values, one float4 (sum, sum sq, min, max) per (batch, 64-column strip,
    // row): part[((batch * tilesN * WN + strip) * M + row)]
    float4* part = nullptr;
    unsigned* omax = nullptr;                   // fp32 output: atomicMax of bits(max|C|)
    const float* bias = nullptr; long sbias = 0;
    const float* R = nullptr; long ldr = 0; long sR = 0;
    const float* rsc = nullptr; const float* rsh = nullptr; long srs = 0;
    int M = 0, N = 0, K = 0, nbatch = 1;
    int tri = TRI_NONE; int trimul = 1;
    int act = ACT_NONE;
    // implicit-GEMM B (impl_k > 0; the HEALPix k x k convolutions on padded faces): the contraction runs over (tap, channel group)
    // - k group kg = tap * impl_cg8 + cg - and reads the P-format planes of the PADDED tensor, plane group cg, at the pixel offset
    // ((tap / impl_k) * impl_pitch + tap % impl_k) * impl_dil from the output pixel: a shifted window per tap, no im2col tensor.
    // ldn = entries per plane group (the caller keeps (impl_k - 1) * impl_dil entries of slack behind the last one); the result is
    // clamped from above by `cap` (capped GELU).  fp32 and / or P-format output; no residual, no triangular form.
    int impl_k = 0, impl_cg8 = 0, impl_pitch = 0, impl_dil = 1;
    float cap = 3.0e38f;
};
hipError_t launch_gemm_f16x3_packed(const Gemm4Args& a, hipStream_t s);
int gemm4_strips(int M, int N);
hipError_t launch_split_f16_tiled(const float* src, long lds_, void* hi, void* lo, long ldd, long rows, int cols,
                                  float scale, hipStream_t s);
// fp32 [K][N] -> P-format planes with optional per-row affine; scale = 2^(12 - exponent(slot))
hipError_t launch_pack_pformat(const float* src, long ldb, long sSrc, int K, int N, int nbatch, const float* sc,
                               const float* sh, long sbs, const unsigned* slot, void* hi, void* lo, long ldn, long sPl,
                               hipStream_t s);

// "Strip" Legendre kernels (strip.hip): the data operand (a 32-column strip of X_m / E_m over the whole contraction) is
// resident in registers, the table arrives as pre-packed MFMA A fragments (strip_pack.h).  Batched over m.
struct LegStripArgs {
    const float* B = nullptr;        // fp32 data operand: element (m, k, n) at m * b_moff + k * b_kstride + n
    long b_kstride = 0, b_moff = 0;
    const _Float16* A = nullptr;     // packed table fragments, scaled by the power of two `ascale`
    const int* tile_off = nullptr;   // per m: first k-step block (device array)
    float ascale = 1.f;
    const unsigned* bmax = nullptr;  // dynamic-range slot holding max|B| (required)
    // output: fp32 C, or fp16 hi/lo planes (element (m, row, n) at m * c_moff + row * c_rstride + n in either form)
    float* C = nullptr; _Float16* Chi = nullptr; _Float16* Clo = nullptr;
    long c_rstride = 0, c_moff = 0;
    float cw = 0.f; unsigned* cslot = nullptr;   // planes: |C| <= cw * bound(B), published to cslot
    unsigned* omax = nullptr;                    // fp32: atomicMax of bits(max|C|)
    int N = 0;        // columns
    int K = 0;        // contraction extent (forward: nlat, inverse: lmax)
    int R = 0;        // row extent (forward: lmax, inverse: nlat)
    int nbatch = 0;   // mmax
    int mode = 0;     // 0 forward (rows l >= m), 1 inverse (contraction over l >= m)
};
// the kernels read up to 15 rows (of b_kstride elements) past the last contraction row of the last batch: buffers they read
// must carry this many rows of slack
constexpr int LEG_STRIP_SLACK_ROWS = 16;
bool legendre_strip_eligible(const LegStripArgs& a);
hipError_t launch_legendre_strip(const LegStripArgs& a, hipStream_t s);
// the equatorially folded form (strip_fold.hip): same arguments, A / tile_off from pack_legendre_fold (strip_pack.h); K, R <= 192
bool legendre_fold_eligible(const LegStripArgs& a);
bool legendre_fold_is_big(const LegStripArgs& a);   // an eligible launch takes legendre_fold_big_kernel (more than 96 folded rows)
hipError_t launch_legendre_fold(const LegStripArgs& a, hipStream_t s);

// conv weight (O x I) -> packed MFMA A fragments (fp16 hi/lo), optionally W diag(a) per sample with the scale derived from
// wmax * max|a| (published to wslot); order 0: blocks by 32-row tile then k16-step (what conv_ws.hip keeps resident), 1: by k16-step then tile
// optional: bf[sample][row] = bias[row] + sum_i W[row][i] b[sample][i] (the folded bias of the same affine)
hipError_t launch_pack_conv_frag(const float* W, long ldw, int O, int I, int order, const float* a, float wmax,
                                 float scale_static, unsigned* wslot, void* dst, long sDst, int nsamples, hipStream_t s,
                                 const float* b = nullptr, const float* bias = nullptr, float* bf = nullptr);

// The block's 1x1 convolutions on the weight-stationary kernel (conv_ws.hip): P-format planes in; P-format planes and / or
// fp32 out.  The mode is derived from the outputs requested (GELU + planes [+ residual + statistics], or residual + fp32
// [+ planes + statistics]).
struct ConvStripArgs {
    const _Float16* Xhi = nullptr; const _Float16* Xlo = nullptr; long ldn = 0; long sX = 0;   // input planes [C/8][ldn][8]
    const unsigned* xslot = nullptr;                  // bound the producer scaled the input planes with
    const _Float16* A = nullptr; long sA = 0;         // packed A fragments (launch_pack_conv_frag order 0), per-sample stride
    const unsigned* aslot = nullptr; float ascale = 1.f;   // dynamic (folded) or static weight scale
    // ... or (GELU modes of conv_ws.hip, conv_wl.hip) the instance norm in front folded into the weights in the kernel's own prologue:
    // fp32 weight (M x C, row pitch ldw floats, 16-byte aligned rows), max|W|, the norm's per-(sample, input channel) affine
    // a (fa) / b (fb); `bias` is then the raw bias (sbias = 0) and A / aslot / ascale are not read
    const float* Wraw = nullptr; long ldw = 0; float wabs = 0.f;
    const float* fa = nullptr; const float* fb = nullptr; long sfa = 0;
    const float* bias = nullptr; long sbias = 0;
    const float* R = nullptr; long sR = 0;            // optional fp32 residual (M x HW per sample), added before the activation
    // ... or the residual as P-format planes [M/8][HW][8] (hi | lo) scaled from the bound in rslot: the fc2 modes of a block whose
    // input exists as planes only (the residual stream never round-trips as fp32)
    const _Float16* Rhi = nullptr; const _Float16* Rlo = nullptr; long sRp = 0; const unsigned* rslot = nullptr;
    float cw = 0.f, cb = 0.f; const unsigned* cinb = nullptr; const unsigned* rmax = nullptr;   // output bound (see Gemm4Args)
    _Float16* Chi = nullptr; _Float16* Clo = nullptr; long sCp = 0; unsigned* cslot = nullptr;  // output planes [M/8][HW][8]
    float4* part = nullptr; int nstrips32 = 0;        // optional row statistics per (sample, 32-pixel strip, row)
    int C = 0, M = 0, HW = 0, nbatch = 1, act = ACT_NONE;
    // second MLP convolution: fp32 output, per-row affine of the residual, range maximum
    float* Cf = nullptr; long sCf = 0;
    const float* rsc = nullptr; const float* rsh = nullptr; long srs = 0;
    unsigned* omax = nullptr;
};
// weights resident in registers, persistent grid over pixel tiles; K in {128, 256, 384, 512, 768}, M % 128 == 0;
// statistics per 32-pixel tile (fc2) or per pixel group (inner skip): nstrips32 >= conv_ws_stat_parts()
bool conv_ws_eligible(int K, int M, long HW, int role, int roles_on = 7);   // role 0 inner skip, 1 fc1, 2 fc2 (-1: any); roles_on: Switches::conv_ws_roles
hipError_t launch_conv_ws(const ConvStripArgs& a, hipStream_t s);
int conv_ws_stat_parts(const ConvStripArgs& a);
// the first MLP convolution (GELU, planes in / planes out, no residual, no statistics) with the weights in LDS and unsynchronised
// waves (conv_wl.hip); K in {128, 256, 384}
bool conv_wl_eligible(int K, int M, long HW);
hipError_t launch_conv_wl(const ConvStripArgs& a, hipStream_t s);   // statistics partials per row that launch writes (nstrips32 >= this)

// dhconv with the filter streamed once into MFMA B fragments (dhconv_strip.hip).  Rows (m, b), m <= l; K = N = 2 C.
struct DhconvStripArgs {
    const _Float16* Dhi = nullptr; const _Float16* Dlo = nullptr; long sD = 0;   // coefficient planes [l][row][2C], per-l stride (halves)
    const unsigned* amax = nullptr;                                               // bound the planes were scaled with
    const _Float16* Whi = nullptr; const _Float16* Wlo = nullptr; long sW = 0;   // compact filter planes [l][Wr|Wi][C/8][C][8], per-l stride
    float bscale = 1.f;                                                           // static power-of-two scale of the filter
    float* E = nullptr; long sE = 0;                                              // fp32 output [l][row][2C]
    unsigned* omax = nullptr;
    int C = 0, L = 0, Mrows = 0, trimul = 1;                                      // rows of degree l: min((l + 1) * trimul, Mrows)
    int groups = 1;                                                               // block-diagonal filter (grouped csfno filter, s2convolutions.py:119-135): zero blocks are skipped
    const int* units = nullptr; int units_per_xcd = 0;                            // work list of the launch: dhconv_build_units(L, Mrows, trimul, C), on the device
    int kstore = 0;                                                               // rows (input channels) each Wr / Wi block holds per output column: 0 / C = dense form;
                                                                                  // C / groups = ONLY the diagonal blocks (the reference's (G, L, C/G, C/G, 2) parameter):
                                                                                  // [l][Wr|Wi][(C/G)/8][C][8], row index relative to the column's own group
};
bool dhconv_native_groups_ok(int C, int groups);   // can the strip kernel read the diagonal-blocks-only form?
hipError_t launch_pack_dhconv_f16g(const float* w_grouped, void* hi, void* lo, int C, int G, int L, float scale, hipStream_t s);
bool dhconv_strip_eligible(const DhconvStripArgs& a);
// host: the launch's work list, 8 x (returned count) entries of 4 ints (l, j, row0, rows), see dhconv_strip.hip
int dhconv_build_units(int L, int Mrows, int trimul, int C, std::vector<int>& out);
hipError_t launch_dhconv_strip(const DhconvStripArgs& a, hipStream_t s);

// A-fragment packing of a conv weight W (O x I, row pitch ldw), optionally with a per-input-channel scale folded in
// (instance-norm affine: W diag(a)), as fp16 hi/lo blocks of 64 lanes x 8 halves (strip_pack.h layout: lane = i + 32 g holds
// row 32 T + i, columns 16 J + 8 g .. + 7):
//   order 0 (streamed by output-row chunk, fc1): block (T, J) at T * (I / 16) + J
//   order 1 (streamed by 16-column step, fc2):   block (T, J) at J * (O / 32) + T
// O % 32 == 0, I % 16 == 0.  scale: a power of two, or derived from `bound` = wmax * max|a| (published to wslot).
// Workgroups [0, (O/32)(I/16)) pack one block each; with `bf`, O / 32 more compute the folded bias bias + W b of a 32-row tile.
struct PackFragArgs {
    const float* W = nullptr; long ldw = 0; int O = 0, I = 0, order = 0;
    const float* a = nullptr; float wmax = 0.f, scale_static = 1.f; unsigned* wslot = nullptr;
    _Float16* dst = nullptr; long sDst = 0;
    const float* b = nullptr; const float* bias = nullptr; float* bf = nullptr;
    int nsamples = 1;
};
inline int pack_frag_blocks(const PackFragArgs& q) { return (q.O / 32) * (q.I / 16) + (q.bf ? q.O / 32 : 0); }

bool pack_frag_args_ok(const PackFragArgs& q);

// Spectral-space layout used between the kernels ("channel-fastest planar"):
//   X[m][k][b][ri][c]  (after the longitude DFT)     index ((m*H + k)*Bt + b)*2C + ri*C + c
//   D[l][m][b][ri][c]  (after the Legendre stage)    index ((l*Mm + m)*Bt + b)*2C + ri*C + c
struct DftArgs {
    const float* x = nullptr;  // (Bt, C, H, W) grid-space field
    // ... or (forward, FFT form only, C % 8 == 0) the field as P-format fp16 planes [C/8][H W][8] hi | lo, scaled from the bound
    // in xslot; sxp = halves per sample
    const _Float16* xhi = nullptr; const _Float16* xlo = nullptr; long sxp = 0; const unsigned* xslot = nullptr;
    float* y = nullptr;        // grid-space output (inverse)
    const float* spec = nullptr;  // spectral input (inverse)
    float* spec_out = nullptr;    // spectral output (forward)
    const float* tc = nullptr; const float* ts = nullptr; int ldt = 0;  // cos / sin tables [Mm][ldt]
    const float* sc = nullptr; const float* sh = nullptr;  // forward: per-(b,c) affine on x (fused instance norm)
    const float* bias = nullptr;                           // inverse: per-c bias added to the output
    int Bt = 1, C = 0, H = 0, W = 0, Mm = 0;
    unsigned* omax = nullptr;  // atomicMax of bits(max|output|)
    bool no_fft = false;       // Switches::no_fft of the owning plan
    // forward, FFT form only: a weight-packing job whose workgroups ride in this launch (pack_frag.h); nride = pack_frag_blocks()
    // per sample, 0 = none.  launch_dft_forward reports through `rode` whether the launch took them along.
    PackFragArgs ride; int nride = 0; bool* rode = nullptr;
};
hipError_t launch_dft_forward(const DftArgs& a, hipStream_t s);
hipError_t launch_dft_inverse(const DftArgs& a, hipStream_t s);
// two-level FFT forms on the vector ALUs (fft.hip); return false when the size / alignment is not covered
bool launch_dft_forward_fft(const DftArgs& a, hipStream_t s, hipError_t* err);
bool launch_dft_inverse_fft(const DftArgs& a, hipStream_t s, hipError_t* err);
bool dft_fft_has_width(int W);   // widths with an instantiated two-level FFT

// Conditional layer norm in one pass (cln_mfma.hip): statistics + the two conditioning 1x1 convolutions on MFMA + apply, for
// C % 256 == 0 and H W % 32 == 0.  As / Ab: W_scale / W_bias as packed MFMA A fragments [C / 32][ceil(J / 16)][hi 512 | lo 512] halves
// scaled by the powers of two ascale_s / ascale_b (pack_cln_frags, host); NULL: plain channel layer norm.  cslot: range slot of cond.
struct ClnMfmaArgs {
    const float* x = nullptr; float* y = nullptr; long sx = 0;       // (nbatch, C, HW); y may alias x
    const float* cond = nullptr; long scond = 0; int J = 0;          // conditioning field (nbatch, J, HW)
    const unsigned* cslot = nullptr;
    const _Float16* As = nullptr; const _Float16* Ab = nullptr; float ascale_s = 1.f, ascale_b = 1.f;
    const float* gamma = nullptr; const float* beta = nullptr;       // optional elementwise affine (C)
    float eps = 1e-5f; int C = 0; long HW = 0; int nbatch = 1;
    unsigned* omax = nullptr;                                        // atomicMax of bits(max|y|); with planes: the BOUND they are scaled by
    // optional P-format output (fp16 hi / lo planes [C / 8][HW][8], the B operand of the packed convolutions), 128-pixel form only
    // (cln_mfma_planes_ok).  y may then be null.  The planes' scale cannot be the true max |y| (unknown until every tile is done):
    // it is the bound (sqrt(C) gmax + bmax)(1 + ws_inf max|cond|) + wb_inf max|cond| - fp16 hi + lo keep full precision as
    // long as the bound is within 2^14 of the true max (it is ~20x) - published to omax, from which consumers derive the scale.
    _Float16* Phi = nullptr; _Float16* Plo = nullptr; long sP = 0;   // per-sample stride in halves
    float gmax = 1.f, bmax = 0.f;                                    // max |gamma|, max |beta| (1, 0 without the affine)
    float ws_inf = 0.f, wb_inf = 0.f;                                // max row sums of |W_scale|, |W_bias|
};
bool cln_mfma_eligible(const ClnMfmaArgs& a);
bool cln_mfma_planes_ok(const ClnMfmaArgs& a);
hipError_t launch_cln_mfma(const ClnMfmaArgs& a, hipStream_t s);

// per-(b,c) instance-norm statistics over H*W -> affine (scale, shift):
//   scale = gamma[c] * rsqrt(var + eps),  shift = beta[c] - mean * scale   (biased var, fp64 accumulation)
hipError_t launch_instnorm_stats(const float* x, const float* gamma, const float* beta, float eps, int Bt, int C,
                                 long HW, float* scale, float* shift, hipStream_t s, unsigned* omax = nullptr);

// fused instance norm, second half: reduce Gemm4Args::part statistics to the per-(sample, channel) affine (+ bound)
hipError_t launch_instnorm_finalize(const float4* part, int nparts, int Bt, int C, long HW, const float* gamma,
                                    const float* beta, float eps, float* scale, float* shift, unsigned* omax,
                                    hipStream_t s);
// W diag(a) as tiled fp16 hi/lo planes + folded bias, bound published to wslot (see kernels.hip)
hipError_t launch_fold_affine_f16(const float* W, long ldw, float wmax, const float* a, const float* b, const float* bias,
                                  void* hi, void* lo, float* bf, int nsamples, int O, int I, long ldd, long sPl,
                                  unsigned* wslot, hipStream_t s);

// conditional layer norm (per-pixel statistics over channels, noise-conditioned scale / bias; see kernels.hip).
// stats: workspace of 2 * Bt * HW floats.  ws / wb (C x J) may be null (plain channel layer norm); gamma / beta may be null
hipError_t launch_cond_layer_norm(const float* x, const float* noise, const float* gamma, const float* beta,
                                  const float* ws, const float* wb, float eps, float* stats, float* y, int Bt, int C,
                                  int J, long HW, hipStream_t s, unsigned* omax = nullptr);

// layout converters between the internal spectral layout and the reference's (n, L, M) complex64
hipError_t launch_spec_to_ref(const float* D, float* out, int Bt, int C, int L, int Mm, hipStream_t s);
// triangular_consumer: the scratch is read by an exactly triangular kernel only (entries with m > l may stay unwritten)
hipError_t launch_ref_to_spec(const float* in, float* E, int Bt, int C, int L, int Mm, hipStream_t s, unsigned* omax = nullptr, bool triangular_consumer = false);

// "diagonal" operator (contractions.py:169-180): E[l][m][b][:, o] = sum_i D[l][m][b][:, i] * w[i][o][l][m] (complex)
hipError_t launch_contract_diagonal(const float* D, const float* w, float* E, int Bt, int Cin, int Cout, int L, int Mm,
                                    hipStream_t s, unsigned* omax = nullptr);

// dhconv weight (Cin, Cout, L, 2) -> expanded real matrices Wx[l][2*Cin][2*Cout]
hipError_t launch_expand_dhconv_weight(const float* w, float* wx, int Cin, int Cout, int L, hipStream_t s);

// y = x * sc[row] + sh[row] + (r ? r * rsc[row] + rsh[row] : 0), rows of length n  (use_mlp=False tail, rare)
hipError_t launch_rowaffine_add(const float* x, const float* sc, const float* sh, const float* r, const float* rsc,
                                const float* rsh, float* y, long rows, long n, hipStream_t s);

// fold a per-(sample, input-channel) affine (a, b) into a conv weight: Wf = W diag(a) (pitch ldw, zero padded),
// bf = bias + W b.  W is (O, I) with pitch ldw.
hipError_t launch_fold_affine(const float* W, long ldw, const float* a, const float* b, const float* bias, float* Wf,
                              float* bf, int nsamples, int O, int I, hipStream_t s);

// stepper glue (packer.py:45-52 + normalizer.py:213-236 fused)
//   pack:   dst[b][j][:] = (src_j[b][:] - mean[j]) / std[j]   where src_j = srcs[j] + b*strides[j]
//   unpack: dst_j[b][:] = src[b][j][:] * std[j] + mean[j]
hipError_t launch_pack_normalize(const float* const* srcs, const long* strides, const float* mean, const float* stdv,
                                 float* dst, int Bt, int nch, long HW, hipStream_t s);
hipError_t launch_unpack_denormalize(const float* src, const float* mean, const float* stdv, float* const* dsts,
                                     const long* strides, int Bt, int nch, long HW, hipStream_t s);

}  // namespace ace
