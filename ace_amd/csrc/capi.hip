// C ABI of libace_sfno.so (include/ace_sfno.h): SHT plans, building blocks and the
// SFNO forward schedule.  Host code only; the kernels live in kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ace_sfno.h"
#include "kernels.h"
#include "strip_pack.h"
#include "tables.h"

using namespace ace;

// ---------------------------------------------------------------------------------------------
// error plumbing: never abort, return a code and keep a thread-local message
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            return fail(ACE_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e__));              \
    } while (0)
#define ACE_TRY(expr)                \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != ACE_OK) return rc__; \
    } while (0)

extern "C" const char* ace_last_error(void) { return g_err.c_str(); }

namespace ace {
Switches read_switches() {
    auto on = [](const char* name) { const char* e = std::getenv(name); return e && e[0] && !(e[0] == '0' && !e[1]); };
    Switches sw;
    sw.no_fft = on("ACE_NO_FFT");
    sw.no_strip = on("ACE_NO_STRIP");
    sw.no_fold = on("ACE_NO_FOLD");
    sw.no_dhconv_strip = on("ACE_NO_DHCONV_STRIP");
#ifdef ACE_MEASUREMENT_SWITCHES   // historical routings: measurement builds only (tools/mkvar.sh NAME -DACE_MEASUREMENT_SWITCHES)
    sw.no_pk = on("ACE_NO_PK");
    sw.no_pk_sht = on("ACE_NO_PK_SHT");
#endif
    sw.no_enc_ws = on("ACE_NO_ENC_WS");
    sw.no_enc_pk = on("ACE_NO_ENC_PK");
    sw.no_cln_mfma = on("ACE_NO_CLN_MFMA");
    sw.no_cln_planes = on("ACE_NO_CLN_PLANES");
    sw.dense_grouped_filter = on("ACE_DENSE_GROUPED_FILTER");
    if (const char* e = std::getenv("ACE_CONV_WL")) sw.conv_wl = !(e[0] == '0' && !e[1]);
    if (const char* e = std::getenv("ACE_PLANES_STREAM")) sw.planes_stream = !(e[0] == '0' && !e[1]);
    if (const char* e = std::getenv("ACE_FUSED_PACK")) sw.fused_pack = !(e[0] == '0' && !e[1]);
    if (const char* e = std::getenv("ACE_CONV_WS")) {
        const std::string v(e);
        if (v == "all" || v == "1") sw.conv_ws_roles = 7;
        else sw.conv_ws_roles = (v.find("skip") != std::string::npos ? 1 : 0) | (v.find("fc1") != std::string::npos ? 2 : 0) |
                                (v.find("fc2") != std::string::npos ? 4 : 0);
    }
    return sw;
}
}  // namespace ace
extern "C" int ace_version(void) { return 100; }

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count, bool zero = true) {
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(float));
        if (e != hipSuccess) { p = nullptr; return e; }
        n = count;
        if (zero) e = hipMemset(p, 0, count * sizeof(float));
        return e;
    }
    hipError_t ensure(size_t count) { return count <= n ? hipSuccess : alloc(count); }
    hipError_t upload(const std::vector<float>& h) {
        hipError_t e = alloc(h.size(), false);
        if (e != hipSuccess) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    }
};

// ---------------------------------------------------------------------------------------------
// SHT plan
// ---------------------------------------------------------------------------------------------
struct ace_sht_plan {
    int nlat = 0, nlon = 0, lmax = 0, mmax = 0, Hp = 0, Lp = 0, Kfp = 0;
    Grid grid = GRID_LEGENDRE_GAUSS;
    DevBuf wt, pt, fc, fs, gc, gs;
    DevBuf X, D;  // scratch of the standalone transforms (plan-owned, grown on demand)
    // f16x3 mode: hi/lo fp16 planes of wt / pt (same pitches, in halves) scaled by a power of two
    DevBuf wt_hi, wt_lo, pt_hi, pt_lo;
    float wt_scale = 1.f, pt_scale = 1.f;
    float wt_winf = 0.f;   // max over (m, l) of sum_k |wt[m][l][k]|: |Legendre-forward output| <= wt_winf * max|X|
    bool f16 = false;
    // strip kernels (strip.hip): wt / pt as pre-packed MFMA A fragments + per-m block offsets (nlat, lmax <= 192)
    DevBuf wt_frag, pt_frag, wt_off, pt_off;
    bool strip = false;
    // ... and in the equatorially folded form (strip_fold.hip) when the tables are mirror-symmetric about the equator
    DevBuf wt_ffrag, pt_ffrag, wt_foff, pt_foff;
    bool fold = false;
    DevBuf slots;   // standalone transforms in f16x3 mode: dynamic-range slots (max|X|, max|coefficients|)
    Switches sw;    // measurement switches, read when the plan was built
    // which kernel family the last Legendre launch of each direction took (ace_sht_plan_route): 0 tile engine, 1 strip.hip,
    // 2 strip_fold.hip, 3 strip_fold.hip big form (more than 96 folded latitudes); -1 = no launch yet
    mutable int route[2] = {-1, -1};
};
static int fold_route(const LegStripArgs& f) { return legendre_fold_is_big(f) ? 3 : 2; }

static float pow2_scale_for(const std::vector<float>& v) {  // puts max|v| in [2^9, 2^10)
    float mx = 0.f;
    for (float x : v) mx = std::max(mx, std::fabs(x));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) { (void)std::frexp(mx, &e); e = 10 - e; }
    return std::ldexp(1.0f, e);
}

static int plan_build(int nlat, int nlon, int lmax, int mmax, Grid g, std::unique_ptr<ace_sht_plan>& out,
                      bool f16 = false) {
    ShtTables t;
    std::string err = build_sht_tables(nlat, nlon, lmax, mmax, g, t);
    if (!err.empty()) return fail(ACE_ERR_INVALID, err);
    auto p = std::make_unique<ace_sht_plan>();
    p->sw = read_switches();
    p->nlat = t.nlat; p->nlon = t.nlon; p->lmax = t.lmax; p->mmax = t.mmax;
    p->Hp = t.Hp; p->Lp = t.Lp; p->Kfp = t.Kfp; p->grid = g;
    HIP_TRY(p->wt.upload(t.wt));
    HIP_TRY(p->pt.upload(t.pt));
    HIP_TRY(p->fc.upload(t.fc));
    HIP_TRY(p->fs.upload(t.fs));
    HIP_TRY(p->gc.upload(t.gc));
    HIP_TRY(p->gs.upload(t.gs));
    if (f16) {
        p->f16 = true;
        p->wt_scale = pow2_scale_for(t.wt);
        for (size_t r = 0; r < (size_t)t.mmax * t.lmax; ++r) {
            double rs = 0.0;
            for (int k = 0; k < t.nlat; ++k) rs += std::fabs((double)t.wt[r * t.Hp + k]);
            p->wt_winf = std::max(p->wt_winf, (float)(rs * (1.0 + 1e-6)));
        }
        p->pt_scale = pow2_scale_for(t.pt);
        const size_t hw = (t.wt.size() + 1) / 2, hp = (t.pt.size() + 1) / 2;
        HIP_TRY(p->wt_hi.alloc(hw, false)); HIP_TRY(p->wt_lo.alloc(hw, false));
        HIP_TRY(p->pt_hi.alloc(hp, false)); HIP_TRY(p->pt_lo.alloc(hp, false));
        HIP_TRY(launch_split_f16(p->wt.p, t.Hp, p->wt_hi.p, p->wt_lo.p, t.Hp, (long)t.mmax * t.lmax, t.Hp, p->wt_scale, nullptr));
        HIP_TRY(launch_split_f16(p->pt.p, t.Lp, p->pt_hi.p, p->pt_lo.p, t.Lp, (long)t.mmax * t.nlat, t.Lp, p->pt_scale, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        auto up = [](const StripPack& sp, DevBuf& frag, DevBuf& off) -> hipError_t {
            hipError_t e = frag.alloc((sp.frags.size() + 1) / 2, false);   // halves -> floats
            if (e != hipSuccess) return e;
            e = hipMemcpy(frag.p, sp.frags.data(), sp.frags.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
            e = off.alloc(sp.tile_off.size(), false);
            if (e != hipSuccess) return e;
            return hipMemcpy(off.p, sp.tile_off.data(), sp.tile_off.size() * sizeof(int), hipMemcpyHostToDevice);
        };
        if (t.nlat <= 192 && t.lmax <= 192) {
            StripPack sp;
            pack_legendre_strip(t.wt.data(), t.mmax, t.lmax, t.nlat, t.Hp, 0, p->wt_scale, sp);
            HIP_TRY(up(sp, p->wt_frag, p->wt_off));
            pack_legendre_strip(t.pt.data(), t.mmax, t.nlat, t.lmax, t.Lp, 1, p->pt_scale, sp);
            HIP_TRY(up(sp, p->pt_frag, p->pt_off));
            p->strip = true;
        }
        // P_l^m(-x) = (-1)^(l+m) P_l^m(x) on a grid that is symmetric about the equator (all three quadratures are): checked
        // on the fp32 tables themselves - mirror entries agree to the rounding of the fp64 recursion.  Folded, a parity's contraction
        // fits the strip kernels' resident operand up to 768 latitudes / degrees (strip_fold.hip: 0.25-degree grid included).
        if ((t.nlat + 1) / 2 <= 16 * FOLD_NKP_BIG && (t.lmax + 1) / 2 <= 16 * FOLD_NKP_BIG && !p->sw.no_fold &&
            fold_symmetry_error(t.wt.data(), t.mmax, t.nlat, t.lmax, t.Hp, 0) < 2e-6 &&
            fold_symmetry_error(t.pt.data(), t.mmax, t.nlat, t.lmax, t.Lp, 1) < 2e-6) {
            StripPack sp;
            pack_legendre_fold(t.wt.data(), t.mmax, t.nlat, t.lmax, t.Hp, 0, p->wt_scale, sp);
            HIP_TRY(up(sp, p->wt_ffrag, p->wt_foff));
            pack_legendre_fold(t.pt.data(), t.mmax, t.nlat, t.lmax, t.Lp, 1, p->pt_scale, sp);
            HIP_TRY(up(sp, p->pt_ffrag, p->pt_foff));
            p->fold = true;
        }
    }
    out = std::move(p);
    return ACE_OK;
}

// X[m][k][b][ri][c] <- longitude DFT of x (Bt, C, H, W), optional per-(b,c) affine on load
static int run_dft_forward(const ace_sht_plan& pl, const float* x, const float* sc, const float* sh, float* X, int Bt,
                           int C, hipStream_t s, unsigned* xmax = nullptr, const _Float16* xhi = nullptr,
                           const _Float16* xlo = nullptr, long sxp = 0, const unsigned* xslot = nullptr,
                           const PackFragArgs* ride = nullptr, bool* rode = nullptr) {
    DftArgs a;
    if (rode) *rode = false;
    if (ride && rode) { a.ride = *ride; a.nride = pack_frag_blocks(*ride); a.rode = rode; }   // a weight-packing job riding in the launch (pack_frag.h)
    a.omax = xmax;
    a.xhi = xhi; a.xlo = xlo; a.sxp = sxp; a.xslot = xslot;   // the field as P-format planes instead of fp32 (FFT form only)
    a.x = xhi ? nullptr : x; a.spec_out = X; a.tc = pl.fc.p; a.ts = pl.fs.p; a.ldt = pl.Kfp; a.sc = sc; a.sh = sh;
    a.Bt = Bt; a.C = C; a.H = pl.nlat; a.W = pl.nlon; a.Mm = pl.mmax; a.no_fft = pl.sw.no_fft;
    HIP_TRY(launch_dft_forward(a, s));
    return ACE_OK;
}
static int run_dft_inverse(const ace_sht_plan& pl, const float* X, const float* bias, float* y, int Bt, int C,
                           hipStream_t s, unsigned* ymax = nullptr) {
    DftArgs a;
    a.omax = ymax;
    a.spec = X; a.y = y; a.tc = pl.gc.p; a.ts = pl.gs.p; a.ldt = pl.Kfp; a.bias = bias;
    a.Bt = Bt; a.C = C; a.H = pl.nlat; a.W = pl.nlon; a.Mm = pl.mmax; a.no_fft = pl.sw.no_fft;
    HIP_TRY(launch_dft_inverse(a, s));
    return ACE_OK;
}
// D[l][m][n2] = sum_k wt[m][l][k] X[m][k][n2]   (sht_fix.py:134-138), batched over m, rows l >= m only
static int run_legendre_forward(const ace_sht_plan& pl, const float* X, float* D, long N2, hipStream_t s,
                                const unsigned* xmax = nullptr, unsigned* dmax = nullptr, bool planes = false) {
    GemmArgs g;
    g.omax = planes ? nullptr : dmax;
    g.A = pl.wt.p; g.lda = pl.Hp; g.sA = (long)pl.lmax * pl.Hp;
    g.B = X; g.ldb = N2; g.sB = (long)pl.nlat * N2;
    g.C = D; g.ldc = (long)pl.mmax * N2; g.sC = N2;
    g.M = pl.lmax; g.N = (int)N2; g.K = pl.nlat; g.nbatch = pl.mmax; g.a_kpad = pl.Hp;
    g.tri = TRI_ROWS_GE_BATCH;
    if ((pl.strip || pl.fold) && !pl.sw.no_strip && xmax) {   // register-resident strip kernels (strip_fold.hip, strip.hip)
        LegStripArgs a;
        a.B = X; a.b_kstride = N2; a.b_moff = (long)pl.nlat * N2;
        a.A = reinterpret_cast<const _Float16*>(pl.wt_frag.p); a.tile_off = reinterpret_cast<const int*>(pl.wt_off.p);
        a.ascale = pl.wt_scale; a.bmax = xmax;
        a.c_rstride = (long)pl.mmax * N2; a.c_moff = N2;
        a.N = (int)N2; a.K = pl.nlat; a.R = pl.lmax; a.nbatch = pl.mmax; a.mode = 0;
        if (planes) {
            a.Chi = reinterpret_cast<_Float16*>(D);
            a.Clo = a.Chi + (size_t)pl.lmax * pl.mmax * N2;
            a.cw = pl.wt_winf; a.cslot = dmax;
        } else {
            a.C = D; a.omax = dmax;
        }
        if (pl.fold && !pl.sw.no_fold) {   // half the contraction: sums / differences of mirror latitudes (strip_fold.hip)
            LegStripArgs f = a;
            f.A = reinterpret_cast<const _Float16*>(pl.wt_ffrag.p); f.tile_off = reinterpret_cast<const int*>(pl.wt_foff.p);
            if (legendre_fold_eligible(f)) {
                HIP_TRY(launch_legendre_fold(f, s));
                pl.route[0] = fold_route(f);
                return ACE_OK;
            }
        }
        if (pl.strip && legendre_strip_eligible(a)) {
            HIP_TRY(launch_legendre_strip(a, s));
            pl.route[0] = 1;
            return ACE_OK;
        }
    }
    pl.route[0] = 0;
    if (planes) {
        // D as fp16 hi/lo planes in D's own layout [l][m][n2] (the buffer holds two planes instead of one fp32 tensor):
        // the dhconv contracts over n2's channel index, so this IS the v4 engine's A operand.  dmax receives the bound.
        if (!(pl.f16 && xmax && dmax && gemm_f16x3_eligible(g))) return fail(ACE_ERR_STATE, "legendre planes path not eligible");
        _Float16* Dh = reinterpret_cast<_Float16*>(D);
        _Float16* Dl = Dh + (size_t)pl.lmax * pl.mmax * N2;
        HIP_TRY(launch_gemm_f16x3_planes(g, pl.wt_hi.p, pl.wt_lo.p, pl.wt_scale, xmax, Dh, Dl, pl.wt_winf, dmax, s));
        return ACE_OK;
    }
    if (pl.f16 && xmax && gemm_f16x3_eligible(g)) {
        HIP_TRY(launch_gemm_f16x3(g, pl.wt_hi.p, pl.wt_lo.p, pl.wt_scale, 1.f, s, xmax, dmax));
        return ACE_OK;
    }
    HIP_TRY(launch_gemm(g, s));
    return ACE_OK;
}
// X[m][k][n2] = sum_{l>=m} pt[m][k][l] E[l][m][n2]   (sht_fix.py:208-219), batched over m
// dry: decide the route (pl.route[1]) without launching anything
static int run_legendre_inverse(const ace_sht_plan& pl, const float* E, float* X, long N2, hipStream_t s,
                                const unsigned* emax = nullptr, bool dry = false) {
    GemmArgs g;
    g.A = pl.pt.p; g.lda = pl.Lp; g.sA = (long)pl.nlat * pl.Lp;
    g.B = E; g.ldb = (long)pl.mmax * N2; g.sB = N2;
    g.C = X; g.ldc = N2; g.sC = (long)pl.nlat * N2;
    g.M = pl.nlat; g.N = (int)N2; g.K = pl.lmax; g.nbatch = pl.mmax; g.a_kpad = pl.Lp;
    g.tri = TRI_K_GE_BATCH;
    if ((pl.strip || pl.fold) && !pl.sw.no_strip && emax) {   // register-resident strip kernels (strip_fold.hip, strip.hip)
        LegStripArgs a;
        a.B = E; a.b_kstride = (long)pl.mmax * N2; a.b_moff = N2;
        a.A = reinterpret_cast<const _Float16*>(pl.pt_frag.p); a.tile_off = reinterpret_cast<const int*>(pl.pt_off.p);
        a.ascale = pl.pt_scale; a.bmax = emax;
        a.C = X; a.c_rstride = N2; a.c_moff = (long)pl.nlat * N2;
        a.N = (int)N2; a.K = pl.lmax; a.R = pl.nlat; a.nbatch = pl.mmax; a.mode = 1;
        if (pl.fold && !pl.sw.no_fold) {   // even / odd degrees separately, rows and mirror rows from their sum and difference
            LegStripArgs f = a;
            f.A = reinterpret_cast<const _Float16*>(pl.pt_ffrag.p); f.tile_off = reinterpret_cast<const int*>(pl.pt_foff.p);
            if (legendre_fold_eligible(f)) {
                if (!dry) HIP_TRY(launch_legendre_fold(f, s));
                pl.route[1] = fold_route(f);
                return ACE_OK;
            }
        }
        if (pl.strip && legendre_strip_eligible(a)) {
            if (!dry) HIP_TRY(launch_legendre_strip(a, s));
            pl.route[1] = 1;
            return ACE_OK;
        }
    }
    pl.route[1] = 0;
    if (dry) return ACE_OK;
    if (pl.f16 && emax && gemm_f16x3_eligible(g)) {
        HIP_TRY(launch_gemm_f16x3(g, pl.pt_hi.p, pl.pt_lo.p, pl.pt_scale, 1.f, s, emax, nullptr));
        return ACE_OK;
    }
    HIP_TRY(launch_gemm(g, s));
    return ACE_OK;
}

extern "C" int ace_sht_plan_create_ex(int nlat, int nlon, int lmax, int mmax, const char* grid, int precision,
                                      ace_sht_plan** plan) {
    if (!plan || !grid) return fail(ACE_ERR_INVALID, "null argument");
    if (precision != 0 && precision != 1) return fail(ACE_ERR_INVALID, "precision must be 0 (fp32) or 1 (f16x3)");
    Grid g;
    if (std::string(grid) == "healpix") return fail(ACE_ERR_INVALID, "'healpix' grid not supported");
    if (!parse_grid(grid, &g)) return fail(ACE_ERR_INVALID, "Unknown quadrature mode");
    std::unique_ptr<ace_sht_plan> p;
    ACE_TRY(plan_build(nlat, nlon, lmax, mmax, g, p, precision == 1));
    if (precision == 1) HIP_TRY(p->slots.alloc((size_t)2 * AMAX_SHARDS));
    *plan = p.release();
    return ACE_OK;
}
extern "C" int ace_sht_plan_create(int nlat, int nlon, int lmax, int mmax, const char* grid, ace_sht_plan** plan) {
    return ace_sht_plan_create_ex(nlat, nlon, lmax, mmax, grid, 0, plan);
}
extern "C" void ace_sht_plan_destroy(ace_sht_plan* plan) { delete plan; }
extern "C" int ace_sht_plan_dims(const ace_sht_plan* p, int* nlat, int* nlon, int* lmax, int* mmax) {
    if (!p) return fail(ACE_ERR_INVALID, "null plan");
    if (nlat) *nlat = p->nlat;
    if (nlon) *nlon = p->nlon;
    if (lmax) *lmax = p->lmax;
    if (mmax) *mmax = p->mmax;
    return ACE_OK;
}

extern "C" int ace_sht_plan_route(const ace_sht_plan* p, int* forward, int* inverse) {
    if (!p) return fail(ACE_ERR_INVALID, "null plan");
    if (forward) *forward = p->route[0];
    if (inverse) *inverse = p->route[1];
    return ACE_OK;
}

extern "C" int ace_sht_forward(ace_sht_plan* p, const float* x, float* coeffs, int n, void* stream) {
    if (!p || !x || !coeffs || n <= 0) return fail(ACE_ERR_INVALID, "ace_sht_forward: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long N2 = 2L * n;
    HIP_TRY(p->X.ensure(((size_t)p->mmax * p->nlat + LEG_STRIP_SLACK_ROWS) * N2));
    HIP_TRY(p->D.ensure(((size_t)p->lmax + LEG_STRIP_SLACK_ROWS) * p->mmax * N2));
    // coefficients with l < m are never written by the triangular Legendre stage: the layout converter writes their zeros
    // without reading the scratch (no memset)
    unsigned* xmax = nullptr;
    if (p->f16 && p->slots.p) {
        xmax = reinterpret_cast<unsigned*>(p->slots.p);
        HIP_TRY(launch_zero_u32(xmax, 2 * AMAX_SHARDS, s));
    }
    ACE_TRY(run_dft_forward(*p, x, nullptr, nullptr, p->X.p, 1, n, s, xmax));
    ACE_TRY(run_legendre_forward(*p, p->X.p, p->D.p, N2, s, xmax, xmax ? xmax + AMAX_SHARDS : nullptr));
    HIP_TRY(launch_spec_to_ref(p->D.p, coeffs, 1, n, p->lmax, p->mmax, s));
    return ACE_OK;
}
extern "C" int ace_sht_inverse(ace_sht_plan* p, const float* coeffs, float* x, int n, void* stream) {
    if (!p || !x || !coeffs || n <= 0) return fail(ACE_ERR_INVALID, "ace_sht_inverse: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long N2 = 2L * n;
    HIP_TRY(p->X.ensure(((size_t)p->mmax * p->nlat + LEG_STRIP_SLACK_ROWS) * N2));
    HIP_TRY(p->D.ensure(((size_t)p->lmax + LEG_STRIP_SLACK_ROWS) * p->mmax * N2));
    unsigned* emax = nullptr;
    if (p->f16 && p->slots.p) {   // f16x3: range of the coefficients, taken by the layout converter as it reads them (round 5: a pass
        emax = reinterpret_cast<unsigned*>(p->slots.p);   // of its own over the converted tensor, 45 us at the reference's benchmark size)
        HIP_TRY(launch_zero_u32(emax, 2 * AMAX_SHARDS, s));
    }
    // the strip kernels are exactly triangular: the converter then neither reads nor writes the entries with m > l (half the tensor)
    ACE_TRY(run_legendre_inverse(*p, p->D.p, p->X.p, N2, s, emax, true));
    HIP_TRY(launch_ref_to_spec(coeffs, p->D.p, 1, n, p->lmax, p->mmax, s, emax, p->route[1] != 0));
    ACE_TRY(run_legendre_inverse(*p, p->D.p, p->X.p, N2, s, emax));
    ACE_TRY(run_dft_inverse(*p, p->X.p, nullptr, x, 1, n, s));
    return ACE_OK;
}

extern "C" int ace_sht_tables_host(int nlat, int nlon, int lmax, int mmax, const char* grid, int which,
                                   void* out_host) {
    if (!grid || !out_host) return fail(ACE_ERR_INVALID, "null argument");
    Grid g;
    if (!parse_grid(grid, &g)) return fail(ACE_ERR_INVALID, "Unknown quadrature mode");
    if (which == 2 || which == 3) {
        std::vector<double> x, w;
        quadrature(g, nlat, x, w);
        std::memcpy(out_host, (which == 2 ? x : w).data(), sizeof(double) * nlat);
        return ACE_OK;
    }
    ShtTables t;
    std::string err = build_sht_tables(nlat, nlon, lmax, mmax, g, t);
    if (!err.empty()) return fail(ACE_ERR_INVALID, err);
    float* o = static_cast<float*>(out_host);
    for (int m = 0; m < t.mmax; ++m)
        for (int l = 0; l < t.lmax; ++l)
            for (int k = 0; k < t.nlat; ++k) {
                const size_t dst = ((size_t)m * t.lmax + l) * t.nlat + k;
                if (which == 0) o[dst] = t.wt[((size_t)m * t.lmax + l) * t.Hp + k];
                else if (which == 1) o[dst] = t.pt[((size_t)m * t.nlat + k) * t.Lp + l];
                else return fail(ACE_ERR_INVALID, "unknown table id");
            }
    return ACE_OK;
}

// ---------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------
extern "C" int ace_conv1x1(const float* x, const float* weight, const float* bias, float* y, int n, int cin, int cout,
                           long hw, int act, void* stream) {
    if (!x || !weight || !y || n <= 0 || cin <= 0 || cout <= 0 || hw <= 0)
        return fail(ACE_ERR_INVALID, "ace_conv1x1: bad argument");
    GemmArgs g;
    g.A = weight; g.lda = cin; g.sA = 0;
    g.B = x; g.ldb = hw; g.sB = (long)cin * hw;
    g.C = y; g.ldc = hw; g.sC = (long)cout * hw;
    g.bias = bias; g.M = cout; g.N = (int)hw; g.K = cin; g.nbatch = n; g.act = act;
    g.a_kpad = cin;  // caller's weight is unpadded: only K % stage-depth == 0 qualifies for the direct-to-LDS engine
    HIP_TRY(launch_gemm(g, static_cast<hipStream_t>(stream)));
    return ACE_OK;
}

extern "C" int ace_conv1x1_f16x3(const float* x, const float* weight, const float* bias, float* y, int n, int cin,
                                 int cout, long hw, int act, void* stream) {
    if (!x || !weight || !y || n <= 0 || cin <= 0 || cout <= 0 || hw <= 0)
        return fail(ACE_ERR_INVALID, "ace_conv1x1_f16x3: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int pitch = (cin + 31) & ~31;
    // one-off weight preparation (what ace_sfno_set_weight does once per parameter): absmax -> power-of-two scale -> split
    std::vector<float> host((size_t)cout * cin);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(host.data(), weight, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (float v : host) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) { (void)std::frexp(mx, &e); e = 10 - e; }
    const float ascale = std::ldexp(1.0f, e);
    DevBuf hi, lo;
    const size_t halves = (size_t)cout * pitch;
    HIP_TRY(hi.alloc((halves + 1) / 2, false));
    HIP_TRY(lo.alloc((halves + 1) / 2, false));
    HIP_TRY(launch_split_f16(weight, cin, hi.p, lo.p, pitch, cout, cin, ascale, s));
    GemmArgs g;
    g.lda = pitch; g.sA = 0; g.a_kpad = pitch;
    g.B = x; g.ldb = hw; g.sB = (long)cin * hw;
    g.C = y; g.ldc = hw; g.sC = (long)cout * hw;
    g.bias = bias; g.M = cout; g.N = (int)hw; g.K = cin; g.nbatch = n; g.act = act;
    if (!gemm_f16x3_eligible(g)) return fail(ACE_ERR_INVALID, "ace_conv1x1_f16x3: needs 16-byte aligned x and hw % 4 == 0");
    if (g.act == ACT_GELU) g.act = ACT_GELU_FAST;
    DevBuf slotbuf;
    HIP_TRY(slotbuf.alloc(AMAX_SHARDS));
    unsigned* xslot = reinterpret_cast<unsigned*>(slotbuf.p);
    HIP_TRY(launch_absmax(x, (long)n * cin * hw, xslot, s));
    HIP_TRY(launch_gemm_f16x3(g, hi.p, lo.p, ascale, 1.0f, s, xslot, nullptr));
    HIP_TRY(hipStreamSynchronize(s));  // temporaries are freed on return
    return ACE_OK;
}

// one-off preparation of a 1x1-conv weight for the f16x3 engines (what ace_sfno_set_weight does once per parameter)
struct PreparedWeight { DevBuf hi, lo; int pitch = 0; float ascale = 1.f, winf = 0.f; };
static int prepare_weight(const float* weight, int rows, int cols, hipStream_t s, PreparedWeight& out, bool tiled) {
    std::vector<float> host((size_t)rows * cols);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(host.data(), weight, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (int r = 0; r < rows; ++r) {
        double rs = 0.0;
        for (int c = 0; c < cols; ++c) { const float v = std::fabs(host[(size_t)r * cols + c]); mx = std::max(mx, v); rs += v; }
        out.winf = std::max(out.winf, (float)(rs * (1.0 + 1e-6)));
    }
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) { (void)std::frexp(mx, &e); e = 10 - e; }
    out.ascale = std::ldexp(1.0f, e);
    out.pitch = (cols + 31) & ~31;
    const size_t halves = (size_t)((rows + 15) / 16 * 16) * out.pitch;
    HIP_TRY(out.hi.alloc((halves + 1) / 2, false));
    HIP_TRY(out.lo.alloc((halves + 1) / 2, false));
    if (tiled) HIP_TRY(launch_split_f16_tiled(weight, cols, out.hi.p, out.lo.p, out.pitch, rows, cols, out.ascale, s));
    else HIP_TRY(launch_split_f16(weight, cols, out.hi.p, out.lo.p, out.pitch, rows, cols, out.ascale, s));
    return ACE_OK;
}

// The reference's MLP (fme/ace/models/modulus/layers.py:97-137: 1x1 conv -> activation -> 1x1 conv, drop_rate 0) on
// the packed-operand f16x3 engine: x is split once into P format, fc1 writes the hidden activation straight in
// P format (bound-scaled), fc2 consumes it by DMA.  Operator-level entry for tests / benchmarks.
extern "C" int ace_mlp_f16x3(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y,
                             int n, int cin, int hid, int cout, long hw, int act, void* stream) {
    if (!x || !w1 || !w2 || !y || n <= 0 || cin <= 0 || hid <= 0 || cout <= 0 || hw <= 0)
        return fail(ACE_ERR_INVALID, "ace_mlp_f16x3: bad argument");
    if (cin % 8 != 0 || hid % 8 != 0 || hw % 4 != 0)
        return fail(ACE_ERR_INVALID, "ace_mlp_f16x3: needs cin % 8 == 0, hid % 8 == 0, hw % 4 == 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PreparedWeight p1, p2;
    ACE_TRY(prepare_weight(w1, hid, cin, s, p1, true));
    ACE_TRY(prepare_weight(w2, cout, hid, s, p2, true));
    float b1max = 0.f;
    if (b1) {
        std::vector<float> hb((size_t)hid);
        HIP_TRY(hipMemcpy(hb.data(), b1, hb.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (float v : hb) b1max = std::max(b1max, std::fabs(v));
    }
    DevBuf slots, xp, up;
    HIP_TRY(slots.alloc(2 * AMAX_SHARDS, true));
    HIP_TRY(xp.alloc((size_t)n * cin * hw, true));   // two fp16 planes = one fp32 tensor's bytes
    HIP_TRY(up.alloc((size_t)n * hid * hw, true));
    unsigned* xslot = reinterpret_cast<unsigned*>(slots.p);
    unsigned* uslot = xslot + AMAX_SHARDS;
    _Float16* xh = reinterpret_cast<_Float16*>(xp.p);
    _Float16* xl = xh + (size_t)n * cin * hw;
    _Float16* uh = reinterpret_cast<_Float16*>(up.p);
    _Float16* ul = uh + (size_t)n * hid * hw;
    HIP_TRY(launch_absmax(x, (long)n * cin * hw, xslot, s));
    HIP_TRY(launch_pack_pformat(x, hw, (long)cin * hw, cin, (int)hw, n, nullptr, nullptr, 0, xslot, xh, xl, hw,
                                (long)cin * hw, s));
    Gemm4Args a;
    a.Ahi = reinterpret_cast<const _Float16*>(p1.hi.p); a.Alo = reinterpret_cast<const _Float16*>(p1.lo.p);
    a.lda = p1.pitch; a.ascale = p1.ascale; a.a_tiled = 1;
    a.Bhi = xh; a.Blo = xl; a.ldn = hw; a.sB = (long)cin * hw; a.bmax = xslot;
    a.Chi = uh; a.Clo = ul; a.ldnc = hw; a.sCp = (long)hid * hw; a.cw = p1.winf; a.cb = b1max; a.cslot = uslot;
    a.bias = b1; a.M = hid; a.N = (int)hw; a.K = cin; a.nbatch = n;
    a.act = act == ACT_GELU ? ACT_GELU_FAST : act;
    HIP_TRY(launch_gemm_f16x3_packed(a, s));
    Gemm4Args b;
    b.Ahi = reinterpret_cast<const _Float16*>(p2.hi.p); b.Alo = reinterpret_cast<const _Float16*>(p2.lo.p);
    b.lda = p2.pitch; b.ascale = p2.ascale; b.a_tiled = 1;
    b.Bhi = uh; b.Blo = ul; b.ldn = hw; b.sB = (long)hid * hw; b.bmax = uslot;
    b.C = y; b.ldc = hw; b.sC = (long)cout * hw;
    b.bias = b2; b.M = cout; b.N = (int)hw; b.K = hid; b.nbatch = n; b.act = ACT_NONE;
    HIP_TRY(launch_gemm_f16x3_packed(b, s));
    HIP_TRY(hipStreamSynchronize(s));  // temporaries are freed on return
    return ACE_OK;
}

// the work list of a dhconv_strip launch (dhconv_build_units) on the device
static hipError_t upload_dhconv_units(int L, int Mrows, int trimul, int C, DevBuf& buf, int* per_xcd) {
    std::vector<int> u;
    *per_xcd = dhconv_build_units(L, Mrows, trimul, C, u);
    hipError_t e = buf.alloc(u.size(), false);
    if (e != hipSuccess) return e;
    return hipMemcpy(buf.p, u.data(), u.size() * sizeof(int), hipMemcpyHostToDevice);
}

// _contract_dhconv (fme/ace/models/modulus/contractions.py:183-195): einsum("bixy,iox->boxy") on complex coefficients, on the
// kernel the network uses (dhconv_strip.hip: compensated fp16, filter streamed once).  Operator-level entry for tests and
// micro-benchmarks: converts to the internal layouts, prepares the filter planes on every call and synchronises.
extern "C" int ace_dhconv_f16x3(const float* coeffs, const float* weight, float* out, int n, int c, int L, int Mm, void* stream) {
    if (!coeffs || !weight || !out || n <= 0 || c <= 0 || L <= 0 || Mm <= 0) return fail(ACE_ERR_INVALID, "ace_dhconv_f16x3: bad argument");
    if (c % 128 != 0) return fail(ACE_ERR_INVALID, "ace_dhconv_f16x3: needs channels % 128 == 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long N2 = 2L * n * c;
    const size_t spec = (size_t)L * Mm * N2;
    DevBuf D, Dp, E, Wh, Wl, slots;
    HIP_TRY(D.alloc(spec, true));
    HIP_TRY(Dp.alloc(spec, true));          // hi | lo planes: two halves per element = one float
    HIP_TRY(E.alloc(spec, true));           // rows m > l are never written: they stay zero
    HIP_TRY(slots.alloc(2 * AMAX_SHARDS, true));
    const size_t wh = (size_t)L * 2 * c * c;
    HIP_TRY(Wh.alloc((wh + 1) / 2, false));
    HIP_TRY(Wl.alloc((wh + 1) / 2, false));
    unsigned* dslot = reinterpret_cast<unsigned*>(slots.p);
    unsigned* wslot = dslot + AMAX_SHARDS;
    HIP_TRY(launch_ref_to_spec(coeffs, D.p, n, c, L, Mm, s));
    HIP_TRY(launch_absmax(D.p, (long)spec, dslot, s));
    HIP_TRY(launch_absmax(weight, (long)c * c * L * 2, wslot, s));
    unsigned bits[2 * AMAX_SHARDS];
    HIP_TRY(hipMemcpyAsync(bits, slots.p, sizeof(bits), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    float dmx = 0.f, wmx = 0.f;
    for (int k = 0; k < AMAX_SHARDS; ++k) {
        float f; std::memcpy(&f, &bits[k], 4); dmx = std::max(dmx, f);
        std::memcpy(&f, &bits[AMAX_SHARDS + k], 4); wmx = std::max(wmx, f);
    }
    auto pow2 = [](float mx, int top) { int e = 0; if (mx > 0.f && std::isfinite(mx)) { (void)std::frexp(mx, &e); e = top - e; } return std::ldexp(1.0f, e); };
    const float dscale = pow2(dmx, 12), wscale = pow2(wmx, 10);   // the kernel derives 2^(12 - exponent) from the slot itself
    HIP_TRY(launch_split_f16(D.p, N2, Dp.p, reinterpret_cast<_Float16*>(Dp.p) + spec, N2, (long)L * Mm, (int)N2, dscale, s));
    HIP_TRY(launch_pack_dhconv_f16c(weight, Wh.p, Wl.p, c, c, L, wscale, s));
    DhconvStripArgs ds;
    ds.Dhi = reinterpret_cast<const _Float16*>(Dp.p); ds.Dlo = ds.Dhi + spec;
    ds.sD = (long)Mm * N2; ds.amax = dslot;
    ds.Whi = reinterpret_cast<const _Float16*>(Wh.p); ds.Wlo = reinterpret_cast<const _Float16*>(Wl.p);
    ds.sW = (long)2 * c * c; ds.bscale = wscale;
    ds.E = E.p; ds.sE = (long)Mm * N2;
    ds.C = c; ds.L = L; ds.Mrows = Mm * n; ds.trimul = n;
    DevBuf units;
    HIP_TRY(upload_dhconv_units(L, Mm * n, n, c, units, &ds.units_per_xcd));
    ds.units = reinterpret_cast<const int*>(units.p);
    if (!dhconv_strip_eligible(ds)) return fail(ACE_ERR_INVALID, "ace_dhconv_f16x3: shape not covered by dhconv_strip.hip");
    HIP_TRY(launch_dhconv_strip(ds, s));
    HIP_TRY(launch_spec_to_ref(E.p, out, n, c, L, Mm, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ACE_OK;
}

extern "C" int ace_instance_norm(const float* x, const float* gamma, const float* beta, float eps, float* y, int n,
                                 int c, long hw, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || hw <= 0) return fail(ACE_ERR_INVALID, "ace_instance_norm: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* st = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st), sizeof(float) * 2 * n * c));
    hipError_t e = launch_instnorm_stats(x, gamma, beta, eps, n, c, hw, st, st + (size_t)n * c, s);
    if (e == hipSuccess) e = launch_rowaffine_add(x, st, st + (size_t)n * c, nullptr, nullptr, nullptr, y, (long)n * c, hw, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(st);
    HIP_TRY(e);
    return ACE_OK;
}

extern "C" int ace_conditional_layer_norm(const float* x, const float* noise, const float* gamma, const float* beta,
                                          const float* w_scale, const float* w_bias, float eps, float* y, int n, int c,
                                          int noise_dim, long hw, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || hw <= 0 || (w_scale && (!w_bias || !noise || noise_dim <= 0)))
        return fail(ACE_ERR_INVALID, "ace_conditional_layer_norm: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DevBuf stats;
    HIP_TRY(stats.alloc((size_t)2 * n * hw, false));
    HIP_TRY(launch_cond_layer_norm(x, noise, gamma, beta, w_scale, w_bias, eps, stats.p, y, n, c, noise_dim, hw, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ACE_OK;
}

extern "C" int ace_conditional_layer_norm_f16x3(const float* x, const float* noise, const float* gamma, const float* beta,
                                                const float* w_scale, const float* w_bias, float eps, float* y, int n, int c,
                                                int noise_dim, long hw, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || hw <= 0 || (w_scale && (!w_bias || !noise || noise_dim <= 0)))
        return fail(ACE_ERR_INVALID, "ace_conditional_layer_norm_f16x3: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ClnMfmaArgs a;
    a.x = x; a.y = y; a.sx = (long)c * hw; a.C = c; a.HW = hw; a.nbatch = n; a.eps = eps; a.gamma = gamma; a.beta = beta;
    DevBuf fs, fb, cslot;
    {   // the shape requirement, checked BEFORE any weight is copied or packed (same predicate as the launch: the conditioning
        // operands are stood in for by the input pointer, only their presence matters to it)
        ClnMfmaArgs probe = a;
        if (w_scale) {
            probe.cond = noise; probe.scond = (long)noise_dim * hw; probe.J = noise_dim;
            probe.As = probe.Ab = reinterpret_cast<const _Float16*>(x); probe.cslot = reinterpret_cast<const unsigned*>(x);
        }
        if (!cln_mfma_eligible(probe))
            return fail(ACE_ERR_INVALID, "ace_conditional_layer_norm_f16x3: needs c % 256 == 0 (c <= 1024), hw % 4 == 0 (c > 512: hw % 32 == 0), noise_dim <= 128");
    }
    if (w_scale) {   // one-off weight preparation (what ace_sfno_set_weight does once per parameter) + the conditioning field's bound
        const size_t halves = cln_frag_halves(c, noise_dim);
        std::vector<float> host((size_t)c * noise_dim);
        std::vector<uint16_t> frags(halves);
        int k = 0;
        for (const float* wsrc : {w_scale, w_bias}) {
            HIP_TRY(hipMemcpy(host.data(), wsrc, host.size() * sizeof(float), hipMemcpyDeviceToHost));
            float mx = 0.f;
            for (float v : host) mx = std::max(mx, std::fabs(v));
            const float sc = cln_frag_scale(mx);
            pack_cln_frags(host.data(), c, noise_dim, sc, frags.data());
            DevBuf& dst = k == 0 ? fs : fb;
            HIP_TRY(dst.alloc((halves + 1) / 2, false));
            HIP_TRY(hipMemcpy(dst.p, frags.data(), halves * sizeof(uint16_t), hipMemcpyHostToDevice));
            (k == 0 ? a.ascale_s : a.ascale_b) = sc;
            ++k;
        }
        HIP_TRY(cslot.alloc(AMAX_SHARDS));
        HIP_TRY(launch_absmax(noise, (long)n * noise_dim * hw, reinterpret_cast<unsigned*>(cslot.p), s));
        a.cond = noise; a.scond = (long)noise_dim * hw; a.J = noise_dim; a.cslot = reinterpret_cast<const unsigned*>(cslot.p);
        a.As = reinterpret_cast<const _Float16*>(fs.p); a.Ab = reinterpret_cast<const _Float16*>(fb.p);
    }
    if (!cln_mfma_eligible(a))
        return fail(ACE_ERR_INVALID, "ace_conditional_layer_norm_f16x3: needs c % 256 == 0 (c <= 1024), hw % 4 == 0 (c > 512: hw % 32 == 0), noise_dim <= 128");
    HIP_TRY(launch_cln_mfma(a, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ACE_OK;
}

extern "C" int ace_pack_normalize(const float* const* srcs, const long* strides, const float* mean, const float* std_,
                                  float* dst, int batch, int nch, long hw, void* stream) {
    if (!srcs || !strides || !mean || !std_ || !dst) return fail(ACE_ERR_INVALID, "ace_pack_normalize: null argument");
    HIP_TRY(launch_pack_normalize(srcs, strides, mean, std_, dst, batch, nch, hw, static_cast<hipStream_t>(stream)));
    return ACE_OK;
}
extern "C" int ace_unpack_denormalize(const float* src, const float* mean, const float* std_, float* const* dsts,
                                      const long* strides, int batch, int nch, long hw, void* stream) {
    if (!src || !strides || !mean || !std_ || !dsts)
        return fail(ACE_ERR_INVALID, "ace_unpack_denormalize: null argument");
    HIP_TRY(launch_unpack_denormalize(src, mean, std_, dsts, strides, batch, nch, hw, static_cast<hipStream_t>(stream)));
    return ACE_OK;
}

// ---------------------------------------------------------------------------------------------
// the network
// ---------------------------------------------------------------------------------------------
struct Weight {
    std::string name;
    long numel = 0;
    DevBuf buf;    // library copy (reference layout; conv weights: rows zero-padded to `pitch`)
    bool set = false;
    int block = -1;
    bool is_filter = false;
    long ext_numel = 0;  // numel of the caller's tensor when it differs from the library copy (grouped csfno filter)
    int rows = 0, cols = 0, pitch = 0;  // conv weights (rows x cols), pitch = cols rounded up to 32
    DevBuf hi, lo;      // f16x3 mode: fp16 planes of the conv weight scaled by `ascale` (pitch halves)
    float ascale = 1.f;
    DevBuf thi, tlo;    // the same planes in the v4 engine's A-tile order (rows padded to 16)
    DevBuf frag0;       // inner_skip / mlp.fwd.2 / last encoder convolution: packed MFMA A fragments (conv_ws.hip, static weights)
    float wabs = 0.f;   // conv weights: max |w|
    float winf = 0.f;   // conv weights: max row sum of |w| (bounds |W x| by winf * max|x|)
    float absmax = 0.f; // small parameters (biases): max |value|
};

struct GraphKey {
    const float* in; float* out; int batch;
    bool operator<(const GraphKey& o) const { return std::tie(in, out, batch) < std::tie(o.in, o.out, o.batch); }
};

struct ace_sfno {
    ace_sfno_config cfg;
    int H = 0, W = 0, C = 0, L = 0, Mm = 0, hid = 0, Bmax = 1;
    long HW = 0;
    // scale_factor != 1 (sfnonet.py:467-515): the blocks between the first filter's inverse transform and the last filter's work on
    // the (nlat / sf) x (nlon / sf) Gauss-Legendre grid; hw_now is the resolution of the launches being enqueued (the 1x1-convolution
    // helpers read it), HWs the inner one
    mutable long hw_now = 0;
    long HWs = 0;
    int Hs = 0, Ws = 0;
    std::unique_ptr<ace_sht_plan> plan_lg, plan_data_own;
    ace_sht_plan* plan_data = nullptr;  // == plan_lg.get() when data_grid is legendre-gauss
    std::vector<std::unique_ptr<Weight>> weights;
    std::map<std::string, int> index;
    std::vector<DevBuf> wx;  // per block: dhconv weight expanded to real [L][2C][2C] (fp32 engines)
    std::vector<DevBuf> wx_hi, wx_lo;  // per block: the same operand k-packed as fp16 hi/lo planes (f16x3 engine)
    std::vector<float> wx_scale;
    std::vector<char> wx_compact;   // per block: wx_hi/lo hold the compact (Wr | Wi) form of Gemm4Args::cplx
    std::vector<DevBuf> dh_units; std::vector<int> dh_units_per_xcd;   // work lists of the filter contraction, one per batch size 1 .. max_batch
    std::vector<char> wx_native;    // per block: ... of a grouped filter with its diagonal blocks only (1 / G of the dense form)
    // workspace
    DevBuf h0, h1, Y, T, R, U, X, D, E, stats;
    DevBuf P;  // f16x3: a C-channel activation as P-format fp16 hi/lo planes (input of the packed-operand GEMM)
    DevBuf cln_stats;    // conditional layer norm: per-pixel mean | rstd
    DevBuf inN;          // conditional layer norm of the network input (normalize_big_skip)
    // residual_filter_factor > 1 (sfnonet.py:473-497): exact-fp32 plan on the data grid with lmax = nlat / r, mmax = nlon / r / 2 + 1,
    // its scratch and the band-limited copy of the network input the decoder's big skip reads
    std::unique_ptr<ace_sht_plan> plan_res;
    DevBuf resX, resD, inF;
    DevBuf P2;           // second one: the block input h as written by the previous block's fc2 epilogue
    DevBuf part;         // per-strip row statistics from the GEMM epilogues (fused instance norm), two tensors
    DevBuf Wp0, Wp1;     // folded (norm affine) skip / fc1 weights as tiled fp16 planes, per sample
    DevBuf Wq1;          // folded (norm affine) fc1 weights as packed MFMA A fragments, per sample (conv_ws.hip)
    DevBuf Wq0;          // folded inner-skip weights likewise
    DevBuf zero_c;       // C zeros (bias of the bias-free last encoder convolution on conv_ws.hip)
    DevBuf inP;          // the network input as P-format planes (in_chans rounded up to 8 channels): operand of the first encoder convolution
    DevBuf pe_slot;      // dynamic-range slot (64 words) holding max|pos_embed|: the residual bound of that convolution
    DevBuf phys;         // fused post-step physics (ace_step_physics): fp64 global sums, parameters
    int nstrips = 0;
    DevBuf Wf0, bf0, Wf1, bf1;
    DevBuf amax;  // [8] uint words: bit patterns of max|X|, max|D|, max|E| of the current block (f16x3 dynamic range)  // instance-norm affine folded into inner_skip / mlp.fc1 weights, per sample
    bool taps_on = false;
    std::vector<DevBuf> taps;
    std::map<GraphKey, hipGraphExec_t> graphs;
    hipStream_t capture_stream = nullptr;
#ifdef ACE_MEASUREMENT_SWITCHES   // fork experiment (profiles/r06_skip_fork.txt): ACE_SKIP_FORK=1 the inner skip's bias-only GEMM on a side
    int skip_fork = 0;            // stream beside the spectral chain (an EXTRA kernel: results unchanged), 2 the same kernel in line
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
#endif
    long weights_generation = 0; // bumped by every ace_sfno_set_weight (ace_sfno_weights_generation)
    bool graphs_stale = false;   // a parameter was uploaded since the graphs were captured: re-capture lazily
    Switches sw;                 // measurement switches, read once at ace_sfno_create

    const float* w(const std::string& name) const {
        auto it = index.find(name);
        return it == index.end() ? nullptr : weights[it->second]->buf.p;
    }
    const Weight* find(const std::string& name) const {
        auto it = index.find(name);
        return it == index.end() ? nullptr : weights[it->second].get();
    }
};

static void add_weight(ace_sfno* n, const std::string& name, long numel, int block = -1, bool is_filter = false) {
    auto w = std::make_unique<Weight>();
    w->name = name; w->numel = numel; w->block = block; w->is_filter = is_filter;
    n->index[name] = (int)n->weights.size();
    n->weights.push_back(std::move(w));
}
// 1x1-conv weight (rows x cols): kept zero-padded to a multiple of 32 columns so the direct-to-LDS GEMM can
// read whole k-stages
static void add_conv_weight(ace_sfno* n, const std::string& name, int rows, int cols) {
    add_weight(n, name, (long)rows * cols);
    Weight& w = *n->weights.back();
    w.rows = rows; w.cols = cols; w.pitch = (cols + 31) & ~31;
}

extern "C" int ace_sfno_create(const ace_sfno_config* cfg, ace_sfno** out) {
    if (!cfg || !out) return fail(ACE_ERR_INVALID, "null argument");
    const ace_sfno_config& c = *cfg;
    if (c.scale_factor < 1) return fail(ACE_ERR_INVALID, "scale_factor must be >= 1");
    if ((c.scale_factor != 1 || c.residual_filter_factor > 1) && c.normalization_layer == 2)
        return fail(ACE_ERR_INVALID, "scale_factor / residual_filter_factor != 1 are not built for the noise-conditioned nets");
    if (c.residual_filter_factor < 0) return fail(ACE_ERR_INVALID, "residual_filter_factor must be >= 1");
    if (c.in_chans <= 0 || c.out_chans <= 0 || c.embed_dim <= 0 || c.num_layers <= 0 || c.nlat < 2 || c.nlon < 2)
        return fail(ACE_ERR_INVALID, "non-positive dimension in ace_sfno_config");
    if (c.operator_type != 0 && c.operator_type != 1) return fail(ACE_ERR_INVALID, "Unsupported operator type");
    if (c.normalization_layer < 0 || c.normalization_layer > 3)
        return fail(ACE_ERR_INVALID, "normalization_layer must be 'none', 'instance_norm', conditional layer norm or 'layer_norm'");
    const bool cln = c.normalization_layer == 2;
    if (cln) {
        if (c.operator_type != 1) return fail(ACE_ERR_INVALID, "Only 'dhconv' operator_type is supported for NoiseConditionedSFNO models.");
        if (c.noise_embed_dim < 0 || c.noise_embed_dim > 512) return fail(ACE_ERR_INVALID, "noise_embed_dim must be in [0, 512]");
        if (c.filter_num_groups < 1 || c.embed_dim % c.filter_num_groups != 0)
            return fail(ACE_ERR_INVALID, "embed_dim must be divisible by filter_num_groups");
    }
    if (c.activation_function < 1 || c.activation_function > 3)
        return fail(ACE_ERR_INVALID, "Unknown activation function");
    // 'lobatto' is what fme.sht_fix.RealSHT defaults to and what the reference's block-level goldens were made on
    // (conditional_sfno/benchmark.py:85-86); the registered builders only ever pass the other two
    if (c.data_grid != GRID_LEGENDRE_GAUSS && c.data_grid != GRID_EQUIANGULAR && c.data_grid != GRID_LOBATTO)
        return fail(ACE_ERR_INVALID, "data_grid must be 'legendre-gauss', 'equiangular' or 'lobatto'");
    if (c.encoder_layers < 0) return fail(ACE_ERR_INVALID, "encoder_layers must be >= 0");   // 0: one bias-free convolution each side (sfnonet.py:566-577, 660-671)
    if (c.precision != 0 && c.precision != 1) return fail(ACE_ERR_INVALID, "precision must be 0 (fp32) or 1 (f16x3)");

    auto n = std::make_unique<ace_sfno>();
    n->cfg = c;
    n->sw = read_switches();
    n->H = c.nlat; n->W = c.nlon; n->C = c.embed_dim; n->HW = (long)c.nlat * c.nlon;
    n->hw_now = n->HW;
    n->Bmax = c.max_batch > 0 ? c.max_batch : 1;
    // sfnonet.py:467-472: the downscaled image and the modes kept
    const int sf = c.scale_factor;
    n->Hs = c.nlat / sf; n->Ws = c.nlon / sf; n->HWs = (long)n->Hs * n->Ws;
    if (n->Hs < 1 || n->Ws < 2) return fail(ACE_ERR_INVALID, "scale_factor leaves no grid");
    n->L = (int)(n->Hs * c.hard_thresholding_fraction);
    n->Mm = (int)((n->Ws / 2 + 1) * c.hard_thresholding_fraction);
    if (n->L < 1 || n->Mm < 1) return fail(ACE_ERR_INVALID, "hard_thresholding_fraction leaves no modes");
    n->hid = (int)(c.embed_dim * c.mlp_ratio);

    // trans / itrans on the inner Gauss-Legendre grid; trans_down / itrans_up on the data grid (sfnonet.py:498-515)
    ACE_TRY(plan_build(n->Hs, n->Ws, n->L, n->Mm, GRID_LEGENDRE_GAUSS, n->plan_lg, c.precision == 1));
    if (c.data_grid == GRID_LEGENDRE_GAUSS && sf == 1) {
        n->plan_data = n->plan_lg.get();
    } else {
        ACE_TRY(plan_build(c.nlat, c.nlon, n->L, n->Mm, (Grid)c.data_grid, n->plan_data_own, c.precision == 1));
        n->plan_data = n->plan_data_own.get();
    }

    if (c.residual_filter_factor > 1 && c.big_skip) {
        const int rl = c.nlat / c.residual_filter_factor, rm = c.nlon / c.residual_filter_factor / 2 + 1;
        if (rl < 1 || rm < 1) return fail(ACE_ERR_INVALID, "residual_filter_factor leaves no modes");
        ACE_TRY(plan_build(c.nlat, c.nlon, rl, rm, (Grid)c.data_grid, n->plan_res, false));
        const size_t n2 = (size_t)n->Bmax * 2 * c.in_chans;
        HIP_TRY(n->resX.alloc(((size_t)rm * c.nlat + LEG_STRIP_SLACK_ROWS) * n2));
        HIP_TRY(n->resD.alloc(((size_t)rl + LEG_STRIP_SLACK_ROWS) * rm * n2));
        HIP_TRY(n->inF.alloc((size_t)n->Bmax * c.in_chans * n->HW));
    }

    // parameters in the reference's state_dict order
    const long C = n->C, HW = n->HW;
    if (c.pos_embed) add_weight(n.get(), "pos_embed", C * HW);
    long cur = c.in_chans;
    for (int j = 0; j < c.encoder_layers; ++j) {
        add_conv_weight(n.get(), "encoder." + std::to_string(2 * j) + ".weight", (int)C, (int)cur);
        add_weight(n.get(), "encoder." + std::to_string(2 * j) + ".bias", C);
        cur = C;
    }
    add_conv_weight(n.get(), "encoder." + std::to_string(2 * c.encoder_layers) + ".weight", (int)C, (int)cur);
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        if (c.normalization_layer == 1) { add_weight(n.get(), p + "norm0.weight", C); add_weight(n.get(), p + "norm0.bias", C); }
        // "layer_norm" (sfnonet.py:584-592): nn.LayerNorm over (H, W), its affine is a pair of (H, W) fields
        const long hw_n0 = i == 0 ? HW : n->HWs;                                  // sfnonet.py:616-623: norm0 sees the block's input grid,
        const long hw_n1 = (i == 0 || i + 1 < c.num_layers) ? n->HWs : HW;        // norm1 its output grid (a first block's: the inner one)
        if (c.normalization_layer == 3) { add_weight(n.get(), p + "norm0.weight", hw_n0); add_weight(n.get(), p + "norm0.bias", hw_n0); }
        auto add_cln = [&](const std::string& q, long ch) {   // ConditionalLayerNorm parameters in state_dict order
            if (c.noise_embed_dim > 0) {
                add_weight(n.get(), q + "W_scale_2d.weight", ch * c.noise_embed_dim);
                add_weight(n.get(), q + "W_bias_2d.weight", ch * c.noise_embed_dim);
            }
            if (c.affine_norms) { add_weight(n.get(), q + "norm.weight", ch); add_weight(n.get(), q + "norm.bias", ch); }
        };
        if (cln) add_cln(p + "norm0.", C);
        const long fw = c.operator_type == 1 ? C * C * n->L * 2 : C * C * (long)n->L * n->Mm * 2;
        add_weight(n.get(), p + "filter.filter.weight", fw, i, true);
        if (cln) n->weights.back()->ext_numel = fw / c.filter_num_groups;   // (G, L, C/G, C/G, 2)
        add_weight(n.get(), p + "filter.filter.bias", C);
        add_conv_weight(n.get(), p + "inner_skip.weight", (int)C, (int)C);
        add_weight(n.get(), p + "inner_skip.bias", C);
        if (c.normalization_layer == 1) { add_weight(n.get(), p + "norm1.weight", C); add_weight(n.get(), p + "norm1.bias", C); }
        if (c.normalization_layer == 3) { add_weight(n.get(), p + "norm1.weight", hw_n1); add_weight(n.get(), p + "norm1.bias", hw_n1); }
        if (cln) add_cln(p + "norm1.", C);
        if (c.use_mlp) {
            add_conv_weight(n.get(), p + "mlp.fwd.0.weight", n->hid, (int)C);
            add_weight(n.get(), p + "mlp.fwd.0.bias", n->hid);
            add_conv_weight(n.get(), p + "mlp.fwd.2.weight", (int)C, n->hid);
            add_weight(n.get(), p + "mlp.fwd.2.bias", C);
        }
    }
    cur = C + (c.big_skip ? c.in_chans : 0);
    for (int j = 0; j < c.encoder_layers; ++j) {
        add_conv_weight(n.get(), "decoder." + std::to_string(2 * j) + ".weight", (int)C, (int)cur);
        add_weight(n.get(), "decoder." + std::to_string(2 * j) + ".bias", C);
        cur = C;
    }
    add_conv_weight(n.get(), "decoder." + std::to_string(2 * c.encoder_layers) + ".weight", c.out_chans, (int)cur);
    if (cln && c.normalize_big_skip && c.big_skip) {
        if (c.noise_embed_dim > 0) {
            add_weight(n.get(), "norm_big_skip.W_scale_2d.weight", (long)c.in_chans * c.noise_embed_dim);
            add_weight(n.get(), "norm_big_skip.W_bias_2d.weight", (long)c.in_chans * c.noise_embed_dim);
        }
        if (c.affine_norms) { add_weight(n.get(), "norm_big_skip.norm.weight", c.in_chans); add_weight(n.get(), "norm_big_skip.norm.bias", c.in_chans); }
    }

    n->wx.resize(c.num_layers);
    n->wx_hi.resize(c.num_layers);
    n->wx_lo.resize(c.num_layers);
    n->wx_scale.assign(c.num_layers, 1.f);
    n->wx_compact.assign(c.num_layers, 0);
    n->wx_native.assign(c.num_layers, 0);
    for (int i = 0; i < c.num_layers; ++i) {
        // blocks whose input/output grid differs from the internal one keep D in fp32 (residual round trip) and use
        // the expanded operand
        const bool mixed = (n->plan_data != n->plan_lg.get()) && (i == 0 || i == c.num_layers - 1);
        n->wx_compact[i] = (!n->sw.no_pk_sht && c.precision == 1 && c.operator_type == 1 && n->C % 128 == 0 && !mixed &&
                            ((long)n->Bmax * 2 * n->C) % 4 == 0) ? 1 : 0;
        // grouped filter of the NoiseConditionedSFNO kept as the reference keeps it: the strip kernel is then the ONLY reader
        // (every runtime precondition of that kernel is decided HERE, from the shape and max_batch: plan arithmetic = c.precision,
        // column alignment = wx_compact, and its row-chunk grid limit for the largest batch - a handle that stores the native form
        // can always run it)
        const bool strip_rows_ok = ((long)n->Mm * n->Bmax + 191) / 192 <= 65535;
        n->wx_native[i] = (n->wx_compact[i] && cln && c.filter_num_groups > 1 && !n->sw.dense_grouped_filter && !n->sw.no_dhconv_strip &&
                           strip_rows_ok && dhconv_native_groups_ok(n->C, c.filter_num_groups)) ? 1 : 0;
    }
#ifdef ACE_MEASUREMENT_SWITCHES
    if (const char* e = std::getenv("ACE_SKIP_FORK")) {
        n->skip_fork = std::atoi(e);
        if (n->skip_fork) {
            HIP_TRY(hipStreamCreateWithFlags(&n->side_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
        }
    }
#endif
    if (c.operator_type == 1 && c.precision == 1 && n->C % 128 == 0) {   // dhconv_strip.hip's work lists (depend on the batch size)
        n->dh_units.resize(n->Bmax);
        n->dh_units_per_xcd.assign(n->Bmax, 0);
        for (int b = 1; b <= n->Bmax; ++b)
            HIP_TRY(upload_dhconv_units(n->L, n->Mm * b, b, n->C, n->dh_units[b - 1], &n->dh_units_per_xcd[b - 1]));
    }
    const size_t act = (size_t)n->Bmax * C * HW;
    // + slack rows read (never used) by the strip Legendre kernels past the last contraction row
    const size_t spec_x = ((size_t)n->Mm * n->H + LEG_STRIP_SLACK_ROWS) * n->Bmax * 2 * C;
    const size_t spec_d = ((size_t)n->L + LEG_STRIP_SLACK_ROWS) * n->Mm * n->Bmax * 2 * C;
    HIP_TRY(n->h0.alloc(act));
    HIP_TRY(n->h1.alloc(act));
    HIP_TRY(n->Y.alloc(act));
    HIP_TRY(n->T.alloc(act));
    if (n->plan_data != n->plan_lg.get()) HIP_TRY(n->R.alloc(act));
    if (c.use_mlp) HIP_TRY(n->U.alloc((size_t)n->Bmax * n->hid * HW, true));
    if (c.precision == 1) {
        HIP_TRY(n->P.alloc(act, true));
        if (c.normalization_layer == 1 && c.use_mlp) {
            HIP_TRY(n->P2.alloc(act, true));
            n->nstrips = (int)((HW + 31) / 32) + 8;     // >= tilesN * WN of either tile shape and conv_ws's 32-pixel tiles
            if (conv_ws_eligible((int)C, n->hid, HW, 1, n->sw.conv_ws_roles) || (n->sw.conv_wl && conv_wl_eligible((int)C, n->hid, HW)))
                HIP_TRY(n->Wq1.alloc((size_t)n->Bmax * n->hid * C, true));   // hi + lo halves = one float per element
            if (conv_ws_eligible((int)C, (int)C, HW, 0, n->sw.conv_ws_roles)) HIP_TRY(n->Wq0.alloc((size_t)n->Bmax * C * C, true));
            HIP_TRY(n->zero_c.alloc((size_t)C, true));
            HIP_TRY(n->inP.alloc((size_t)n->Bmax * ((c.in_chans + 7) / 8 * 8) * n->HW, true));   // two planes of halves = one float per element
            HIP_TRY(n->pe_slot.alloc(64, true));
            HIP_TRY(n->part.alloc((size_t)2 * n->Bmax * n->nstrips * C * 4, true));
            const size_t cp = (size_t)((C + 31) & ~31);
            HIP_TRY(n->Wp0.alloc((size_t)n->Bmax * ((C + 15) / 16 * 16) * cp, true));       // 2 planes of halves = 1 float per element
            HIP_TRY(n->Wp1.alloc((size_t)n->Bmax * ((n->hid + 15) / 16 * 16) * cp, true));
        }
    }
    HIP_TRY(n->X.alloc(spec_x));
    HIP_TRY(n->D.alloc(spec_d));
    HIP_TRY(n->E.alloc(spec_d));
    HIP_TRY(n->stats.alloc((size_t)4 * n->Bmax * C));
    if (cln) {
        HIP_TRY(n->cln_stats.alloc((size_t)2 * n->Bmax * HW));
        if (c.normalize_big_skip && c.big_skip) HIP_TRY(n->inN.alloc((size_t)n->Bmax * c.in_chans * HW));
    }
    HIP_TRY(n->amax.alloc((size_t)(16 + 12 * c.num_layers + 1) * AMAX_SHARDS));   // + the conditioning field's slot
    if (c.normalization_layer == 1) {
        const size_t cp = (size_t)((C + 31) & ~31);
        HIP_TRY(n->Wf0.alloc((size_t)n->Bmax * C * cp));
        HIP_TRY(n->bf0.alloc((size_t)n->Bmax * C));
        if (c.use_mlp) {
            HIP_TRY(n->Wf1.alloc((size_t)n->Bmax * n->hid * cp));
            HIP_TRY(n->bf1.alloc((size_t)n->Bmax * n->hid));
        }
    }
    *out = n.release();
    return ACE_OK;
}

extern "C" void ace_sfno_destroy(ace_sfno* n) {
    if (!n) return;
    for (auto& kv : n->graphs) (void)hipGraphExecDestroy(kv.second);
    if (n->capture_stream) (void)hipStreamDestroy(n->capture_stream);
#ifdef ACE_MEASUREMENT_SWITCHES
    if (n->side_stream) (void)hipStreamDestroy(n->side_stream);
    if (n->ev_fork) (void)hipEventDestroy(n->ev_fork);
    if (n->ev_join) (void)hipEventDestroy(n->ev_join);
#endif
    delete n;
}

extern "C" long ace_sfno_weights_generation(const ace_sfno* n) { return n ? n->weights_generation : -1; }

// Device bytes the library owns for this handle at batch sizes up to `batch` (workspace + both operand forms of the weights
// + tables): SURVEY 8(b) workspace_size.  The workspace is sized by ace_sfno_config.max_batch at creation.
extern "C" long ace_sfno_workspace_size(const ace_sfno* n, int batch) {
    if (!n) { (void)fail(ACE_ERR_INVALID, "null argument"); return -1; }
    if (batch <= 0 || batch > n->Bmax) {
        (void)fail(ACE_ERR_INVALID, "batch " + std::to_string(batch) + " outside [1, max_batch=" + std::to_string(n->Bmax) + "]");
        return -1;
    }
    size_t fl = 0;
    auto add = [&](const DevBuf& b) { fl += b.n; };
    for (const DevBuf* b : {&n->h0, &n->h1, &n->Y, &n->T, &n->R, &n->U, &n->X, &n->D, &n->E, &n->stats, &n->P, &n->cln_stats, &n->inN,
                            &n->P2, &n->part, &n->Wp0, &n->Wp1, &n->Wq1, &n->Wq0, &n->zero_c, &n->inP, &n->pe_slot, &n->Wf0, &n->bf0, &n->Wf1,
                            &n->bf1, &n->amax, &n->phys})
        add(*b);
    for (const auto& t : n->taps) add(t);
    for (const auto& v : {&n->wx, &n->wx_hi, &n->wx_lo})
        for (const auto& b : *v) add(b);
    for (const auto& w : n->weights)
        for (const DevBuf* b : {&w->buf, &w->hi, &w->lo, &w->thi, &w->tlo, &w->frag0}) add(*b);
    auto plan_bytes = [&](const ace_sht_plan* p) {
        if (!p) return;
        for (const DevBuf* b : {&p->wt, &p->pt, &p->fc, &p->fs, &p->gc, &p->gs, &p->X, &p->D, &p->wt_hi, &p->wt_lo, &p->pt_hi, &p->pt_lo,
                                &p->wt_frag, &p->pt_frag, &p->wt_off, &p->pt_off, &p->slots})
            add(*b);
    };
    plan_bytes(n->plan_lg.get());
    plan_bytes(n->plan_data_own.get());
    plan_bytes(n->plan_res.get());
    for (const DevBuf* b : {&n->resX, &n->resD, &n->inF}) add(*b);
    return (long)(fl * sizeof(float));
}

extern "C" int ace_sfno_num_weights(const ace_sfno* n) { return n ? (int)n->weights.size() : 0; }
extern "C" const char* ace_sfno_weight_name(const ace_sfno* n, int i) {
    if (!n || i < 0 || i >= (int)n->weights.size()) return nullptr;
    return n->weights[i]->name.c_str();
}
extern "C" long ace_sfno_weight_numel(const ace_sfno* n, int i) {
    if (!n || i < 0 || i >= (int)n->weights.size()) return -1;
    return n->weights[i]->ext_numel ? n->weights[i]->ext_numel : n->weights[i]->numel;
}

// The inner skip of a noise-conditioned net on conv_ws.hip (mode 6: no per-step weight fold, any K of that file).  ONE predicate for
// the upload (which packs the static fragments) and the forward (which reads them): the "skip" bit of ACE_CONV_WS switches it.
static bool cln_skip_on_conv_ws(const ace_sfno* n, int K, int M) {
    return (n->sw.conv_ws_roles & 1) && conv_ws_eligible(K, M, n->HW, -1, n->sw.conv_ws_roles);
}

extern "C" int ace_sfno_set_weight(ace_sfno* n, const char* name, const float* src, long numel, void* stream) {
    if (!n || !name || !src) return fail(ACE_ERR_INVALID, "null argument");
    auto it = n->index.find(name);
    if (it == n->index.end()) return fail(ACE_ERR_INVALID, std::string("unknown parameter '") + name + "'");
    Weight& w = *n->weights[it->second];
    const long expect = w.ext_numel ? w.ext_numel : w.numel;
    if (numel != expect)
        return fail(ACE_ERR_INVALID, std::string("size mismatch for ") + name + ": expected " +
                                         std::to_string(expect) + " elements, got " + std::to_string(numel));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool native_groups = w.ext_numel && w.is_filter && w.block >= 0 && n->wx_native[w.block];
    if (native_groups) {   // grouped csfno filter kept AS IT IS, (G, L, C/G, C/G, 2): fp32 copy and packed planes hold the diagonal blocks only
        if (!w.buf.p) HIP_TRY(w.buf.alloc((size_t)w.ext_numel, false));
        HIP_TRY(hipMemcpyAsync(w.buf.p, src, sizeof(float) * w.ext_numel, hipMemcpyDeviceToDevice, s));
        src = nullptr;
        numel = w.ext_numel;
    } else if (w.ext_numel) {   // ... or (shapes the strip kernel does not read that way) -> dense (Cin, Cout, L, 2), then as any dhconv weight
        if (!w.buf.p) HIP_TRY(w.buf.alloc((size_t)w.numel, false));
        HIP_TRY(launch_csfno_weight_to_dense(src, w.buf.p, n->C, n->cfg.filter_num_groups, n->L, s));
        src = nullptr;
        numel = w.numel;
    }
    if (w.pitch > 0) {
        if (!w.buf.p) HIP_TRY(w.buf.alloc((size_t)w.rows * w.pitch, true));  // zero padding columns
        HIP_TRY(hipMemcpy2DAsync(w.buf.p, sizeof(float) * w.pitch, src, sizeof(float) * w.cols, sizeof(float) * w.cols,
                                 w.rows, hipMemcpyDeviceToDevice, s));
    } else if (src) {
        if (!w.buf.p) HIP_TRY(w.buf.alloc((size_t)numel, false));
        HIP_TRY(hipMemcpyAsync(w.buf.p, src, sizeof(float) * numel, hipMemcpyDeviceToDevice, s));
    }
    if (w.is_filter && n->cfg.operator_type == 1) {
        const size_t cnt = (size_t)n->L * 2 * n->C * 2 * n->C;
        const bool packed = n->cfg.precision == 1 && (2 * n->C) % 32 == 0;
        if (packed) {  // f16x3: k-packed fp16 hi/lo planes scaled so that max|w| lands in [2^9, 2^10)
            DevBuf slotbuf;
            HIP_TRY(slotbuf.alloc(AMAX_SHARDS));
            HIP_TRY(launch_absmax(w.buf.p, numel, reinterpret_cast<unsigned*>(slotbuf.p), s));
            unsigned bits[AMAX_SHARDS];
            HIP_TRY(hipMemcpyAsync(bits, slotbuf.p, sizeof(bits), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            float mx = 0.f;
            for (unsigned b : bits) { float f; std::memcpy(&f, &b, 4); mx = std::max(mx, f); }
            int e = 0;
            if (mx > 0.f && std::isfinite(mx)) { (void)std::frexp(mx, &e); e = 10 - e; }
            n->wx_scale[w.block] = std::ldexp(1.0f, e);
            const bool compact = n->wx_compact[w.block] != 0;
            const size_t halves = native_groups ? cnt / 2 / n->cfg.filter_num_groups : (compact ? cnt / 2 : cnt);
            if (!n->wx_hi[w.block].p) HIP_TRY(n->wx_hi[w.block].alloc((halves + 1) / 2, false));
            if (!n->wx_lo[w.block].p) HIP_TRY(n->wx_lo[w.block].alloc((halves + 1) / 2, false));
            if (native_groups)
                HIP_TRY(launch_pack_dhconv_f16g(w.buf.p, n->wx_hi[w.block].p, n->wx_lo[w.block].p, n->C, n->cfg.filter_num_groups, n->L,
                                                n->wx_scale[w.block], s));
            else if (compact)
                HIP_TRY(launch_pack_dhconv_f16c(w.buf.p, n->wx_hi[w.block].p, n->wx_lo[w.block].p, n->C, n->C, n->L,
                                                n->wx_scale[w.block], s));
            else
                HIP_TRY(launch_pack_dhconv_f16(w.buf.p, n->wx_hi[w.block].p, n->wx_lo[w.block].p, n->C, n->C, n->L,
                                               n->wx_scale[w.block], s));
        } else {
            DevBuf& wx = n->wx[w.block];
            if (!wx.p) HIP_TRY(wx.alloc(cnt, false));
            HIP_TRY(launch_expand_dhconv_weight(w.buf.p, wx.p, n->C, n->C, n->L, s));
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (w.pitch > 0 && n->cfg.precision == 1) {
        // power-of-two scale that puts max|w| in [2^9, 2^10): hi and lo parts stay in fp16's normal range
        std::vector<float> host((size_t)w.rows * w.pitch);
        HIP_TRY(hipMemcpy(host.data(), w.buf.p, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        float mx = 0.f;
        for (float v : host) mx = std::max(mx, std::fabs(v));
        w.wabs = mx;
        w.winf = 0.f;
        for (int r = 0; r < w.rows; ++r) {
            double rs = 0.0;
            for (int cidx = 0; cidx < w.cols; ++cidx) rs += std::fabs((double)host[(size_t)r * w.pitch + cidx]);
            w.winf = std::max(w.winf, (float)(rs * (1.0 + 1e-6)));
        }
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { (void)std::frexp(mx, &e); e = 10 - e; }
        w.ascale = std::ldexp(1.0f, e);
        const size_t halves = (size_t)w.rows * w.pitch;
        if (!w.hi.p) HIP_TRY(w.hi.alloc((halves + 1) / 2, false));
        if (!w.lo.p) HIP_TRY(w.lo.alloc((halves + 1) / 2, false));
        HIP_TRY(launch_split_f16(w.buf.p, w.pitch, w.hi.p, w.lo.p, w.pitch, w.rows, w.pitch, w.ascale, s));
        const size_t thalves = (size_t)((w.rows + 15) / 16 * 16) * w.pitch;
        if (!w.thi.p) HIP_TRY(w.thi.alloc((thalves + 1) / 2, false));
        if (!w.tlo.p) HIP_TRY(w.tlo.alloc((thalves + 1) / 2, false));
        HIP_TRY(launch_split_f16_tiled(w.buf.p, w.pitch, w.thi.p, w.tlo.p, w.pitch, w.rows, w.cols, w.ascale, s));
        const std::string& wn = w.name;
        const bool is_enc2 = wn == "encoder." + std::to_string(2 * n->cfg.encoder_layers) + ".weight";
        const bool is_skip = wn.size() > 17 && wn.compare(wn.size() - 17, 17, "inner_skip.weight") == 0;
        const bool is_fc2 = wn.size() > 16 && wn.compare(wn.size() - 16, 16, "mlp.fwd.2.weight") == 0;
        // fc1 without an instance norm in front of it (nothing to fold per step): static fragments for conv_wl.hip
        const bool is_fc1 = wn.size() > 16 && wn.compare(wn.size() - 16, 16, "mlp.fwd.0.weight") == 0;
        if (is_fc1 && n->cfg.normalization_layer != 1 && n->sw.conv_wl && conv_wl_eligible(w.cols, w.rows, n->HW)) {
            if (!w.frag0.p) HIP_TRY(w.frag0.alloc((size_t)w.rows * w.cols, false));
            HIP_TRY(launch_pack_conv_frag(w.buf.p, w.pitch, w.rows, w.cols, 0, nullptr, 0.f, w.ascale, nullptr, w.frag0.p,
                                          0, 1, s));
        }
        // (the inner skip of a noise-conditioned net has no per-step weight fold and any K of conv_ws.hip: its mode 6)
        const bool cln_skip = is_skip && n->cfg.normalization_layer == 2;
        const bool ws_ok = cln_skip ? cln_skip_on_conv_ws(n, w.cols, w.rows)
                                    : conv_ws_eligible(w.cols, w.rows, n->HW, is_skip ? 0 : 2, n->sw.conv_ws_roles);
        if ((is_skip || is_fc2 || is_enc2) && ws_ok) {
            if (!w.frag0.p) HIP_TRY(w.frag0.alloc((size_t)w.rows * w.cols, false));
            HIP_TRY(launch_pack_conv_frag(w.buf.p, w.pitch, w.rows, w.cols, 0, nullptr, 0.f, w.ascale, nullptr, w.frag0.p,
                                          0, 1, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (w.name == "pos_embed" && n->pe_slot.p) {   // its bound, as a dynamic-range slot (the residual of the last encoder convolution)
        std::vector<float> host((size_t)numel);
        HIP_TRY(hipMemcpy(host.data(), w.buf.p, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        float mx = 0.f;
        for (float v : host) mx = std::max(mx, std::fabs(v));
        std::vector<float> rep(64, mx);
        HIP_TRY(hipMemcpy(n->pe_slot.p, rep.data(), rep.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    {   // conditioning convolutions of a conditional layer norm: MFMA A fragments for cln_mfma.hip (f16x3 networks, C % 256 == 0)
        const std::string& wn = w.name;
        const bool is_cln_w = (wn.size() > 17 && wn.compare(wn.size() - 17, 17, "W_scale_2d.weight") == 0) ||
                              (wn.size() > 16 && wn.compare(wn.size() - 16, 16, "W_bias_2d.weight") == 0);
        const int J = n->cfg.noise_embed_dim;
        if (is_cln_w && n->cfg.precision == 1 && J >= 1 && J <= 128 && numel == (long)n->C * J && n->C % 256 == 0 && n->C <= 1024) {
            std::vector<float> host((size_t)numel);
            HIP_TRY(hipMemcpy(host.data(), w.buf.p, host.size() * sizeof(float), hipMemcpyDeviceToHost));
            float mx = 0.f;
            for (float v : host) mx = std::max(mx, std::fabs(v));
            w.wabs = mx;
            w.winf = 0.f;   // max row sum of |w|: bounds the convolution's output by winf * max |cond|
            for (int r = 0; r < n->C; ++r) {
                double rs = 0.0;
                for (int j = 0; j < J; ++j) rs += std::fabs((double)host[(size_t)r * J + j]);
                w.winf = std::max(w.winf, (float)(rs * (1.0 + 1e-6)));
            }
            w.ascale = cln_frag_scale(mx);
            std::vector<uint16_t> frags(cln_frag_halves(n->C, J));
            pack_cln_frags(host.data(), n->C, J, w.ascale, frags.data());
            if (!w.frag0.p) HIP_TRY(w.frag0.alloc((frags.size() + 1) / 2, false));
            HIP_TRY(hipMemcpy(w.frag0.p, frags.data(), frags.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    }
    if (w.pitch == 0 && !w.is_filter && numel <= (1 << 16)) {  // biases: bound used by the P-format producers
        std::vector<float> host((size_t)numel);
        HIP_TRY(hipMemcpy(host.data(), w.buf.p, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        w.absmax = 0.f;
        for (float v : host) w.absmax = std::max(w.absmax, std::fabs(v));
    }
    w.set = true;
    // Captured graphs keep pointing at the same library buffers, but the host-side scalars derived from the weights
    // (ascale, winf, wabs, bias bounds, filter scale) are baked into their kernel arguments by value: drop the graphs so
    // that the next ace_sfno_forward_graph re-captures with the new scalars.
    // Marked stale here, destroyed in ace_sfno_forward_graph after a stream synchronise (an exec may still be in flight).
    n->graphs_stale = true;
    n->weights_generation += 1;
    return ACE_OK;
}

// one 1x1 convolution = one batched GEMM launch.  A operand: (ptr, pitch, per-sample stride); bias: (ptr, per-sample stride)
struct ConvW { const float* w; int pitch; long sw; const float* bias; long sbias; const float* hi; const float* lo; float ascale; };
static ConvW conv_weight(const ace_sfno* n, const std::string& wname, const std::string& bname) {
    const Weight& w = *n->weights[n->index.at(wname)];
    return ConvW{w.buf.p, w.pitch, 0, bname.empty() ? nullptr : n->w(bname), 0, w.hi.p, w.lo.p, w.ascale};
}
static int conv(const ace_sfno* n, const ConvW& cw, const float* in, long in_bstride, int cin, const float* in2,
                long in2_bstride, int K1, float* out, int cout, const float* R, long r_bstride, const float* rsc,
                const float* rsh, int act, int batch, hipStream_t s, const float* bsc = nullptr,
                const float* bsh = nullptr, const unsigned* bmax = nullptr, const unsigned* bmax2 = nullptr,
                unsigned* omax = nullptr) {
    GemmArgs g;
    g.bsc = bsc; g.bsh = bsh; g.sbs = bsc ? cin : 0;
    g.omax = omax;
    g.A = cw.w; g.lda = cw.pitch; g.sA = cw.sw; g.a_kpad = cw.pitch;
    g.B = in; g.ldb = n->hw_now; g.sB = in_bstride;
    g.B2 = in2; g.ldb2 = n->hw_now; g.sB2 = in2_bstride; g.K1 = in2 ? K1 : -1;
    g.C = out; g.ldc = n->hw_now; g.sC = (long)cout * n->hw_now;
    g.bias = cw.bias; g.sbias = cw.sbias;
    g.R = R; g.ldr = n->hw_now; g.sR = r_bstride;
    g.rsc = rsc; g.rsh = rsh; g.srs = rsc ? cout : 0;
    g.M = cout; g.N = (int)n->hw_now; g.K = cin; g.nbatch = batch; g.act = act;
    if (n->cfg.precision == 1 && cw.hi && cw.sw == 0 && bmax && gemm_f16x3_eligible(g)) {
        // compensated fp16 with the B scale derived in-kernel from max|B| (slot written by B's producer)
        if (g.act == ACT_GELU) g.act = ACT_GELU_FAST;  // the epilogue is on the critical path of this engine
        HIP_TRY(launch_gemm_f16x3(g, cw.hi, cw.lo, cw.ascale, 1.0f, s, bmax, omax, (in2 ? bmax2 : nullptr)));
        return ACE_OK;
    }
    if (bsc && g.sA == 0 && n->cfg.precision == 1) {
        // f16x3 configuration but this launch fell back to fp32 with an un-folded affine: the register-staged
        // engine applies it on load
    }
    HIP_TRY(launch_gemm(g, s));
    return ACE_OK;
}
// the instance-norm affine (a, b) of the conv input folded into its weight: returns the per-sample operand
static int fold(const ace_sfno* n, const ConvW& cw, int O, int I, const float* a, const float* b, float* Wf, float* bf,
                int batch, hipStream_t s, ConvW* out) {
    HIP_TRY(launch_fold_affine(cw.w, cw.pitch, a, b, cw.bias, Wf, bf, batch, O, I, s));
    *out = ConvW{Wf, cw.pitch, (long)O * cw.pitch, bf, O};
    return ACE_OK;
}

// f16x3 "v4": 1x1 convolution whose input already is in P format (fp16 hi/lo planes [cin/8][HW][8] per sample, scaled
// from the bound in `in_slot`).  Output: fp32 (+ omax) and/or P-format planes for the next convolution.
static bool packed_ok(const ace_sfno* n, int cin) {
    return !n->sw.no_pk && n->cfg.precision == 1 && n->P.p && cin % 8 == 0 && n->hw_now % 4 == 0;
}
static int pack_act(const ace_sfno* n, const float* x, long x_bs, int cin, const float* sc, const float* sh,
                    const unsigned* slot, void* hi, void* lo, int batch, hipStream_t s) {
    HIP_TRY(launch_pack_pformat(x, n->hw_now, x_bs, cin, (int)n->hw_now, batch, sc, sh, sc ? cin : 0, slot, hi, lo, n->hw_now,
                                (long)cin * n->hw_now, s));
    return ACE_OK;
}
static int conv_pk(const ace_sfno* n, const Weight& w, const float* bias, const void* bhi, const void* blo, int cin,
                   const unsigned* in_slot, float* out, int cout, const float* R, long r_bs, const float* rsc,
                   const float* rsh, int act, int batch, hipStream_t s, unsigned* omax, void* ohi = nullptr,
                   void* olo = nullptr, float cb = 0.f, unsigned* cslot = nullptr) {
    Gemm4Args a;
    a.Ahi = static_cast<const _Float16*>((const void*)w.thi.p); a.Alo = static_cast<const _Float16*>((const void*)w.tlo.p);
    a.lda = w.pitch; a.sA = 0; a.ascale = w.ascale; a.a_tiled = 1;
    a.Bhi = static_cast<const _Float16*>(bhi); a.Blo = static_cast<const _Float16*>(blo);
    a.ldn = n->hw_now; a.sB = (long)cin * n->hw_now; a.bmax = in_slot;
    a.C = out; a.ldc = n->hw_now; a.sC = (long)cout * n->hw_now; a.omax = omax;
    a.Chi = static_cast<_Float16*>(ohi); a.Clo = static_cast<_Float16*>(olo); a.ldnc = n->hw_now; a.sCp = (long)cout * n->hw_now;
    a.cw = w.winf; a.cb = cb; a.cslot = cslot;
    a.bias = bias; a.sbias = 0;
    a.R = R; a.ldr = n->hw_now; a.sR = r_bs; a.rsc = rsc; a.rsh = rsh; a.srs = rsc ? cout : 0;
    a.M = cout; a.N = (int)n->hw_now; a.K = cin; a.nbatch = batch;
    a.act = act == ACT_GELU ? ACT_GELU_FAST : act;
    HIP_TRY(launch_gemm_f16x3_packed(a, s));
    return ACE_OK;
}

struct PkOpts {
    // A: static tiled planes of `w`, or per-sample folded planes (fhi/flo with scale slot fslot)
    const Weight* w = nullptr;
    const void* fhi = nullptr; const void* flo = nullptr; const unsigned* fslot = nullptr; long f_stride = 0;
    const float* bias = nullptr; long sbias = 0;
    const void* bhi = nullptr; const void* blo = nullptr; int cin = 0; const unsigned* in_slot = nullptr;
    float* out = nullptr; int cout = 0; unsigned* omax = nullptr;
    const float* R = nullptr; long r_bs = 0; const float* rsc = nullptr; const float* rsh = nullptr;
    int act = ACT_NONE;
    void* ohi = nullptr; void* olo = nullptr; float cb = 0.f; unsigned* cslot = nullptr;
    const unsigned* cinb = nullptr; const unsigned* rmax = nullptr;
    float* part = nullptr;
};
static int conv_pk2(const ace_sfno* n, const PkOpts& o, int batch, hipStream_t s) {
    Gemm4Args a;
    if (o.fhi) {
        a.Ahi = static_cast<const _Float16*>(o.fhi); a.Alo = static_cast<const _Float16*>(o.flo);
        a.sA = o.f_stride; a.amax = o.fslot;
    } else {
        a.Ahi = reinterpret_cast<const _Float16*>(o.w->thi.p); a.Alo = reinterpret_cast<const _Float16*>(o.w->tlo.p);
        a.sA = 0; a.ascale = o.w->ascale;
    }
    a.lda = o.w->pitch; a.a_tiled = 1;
    a.Bhi = static_cast<const _Float16*>(o.bhi); a.Blo = static_cast<const _Float16*>(o.blo);
    a.ldn = n->hw_now; a.sB = (long)o.cin * n->hw_now; a.bmax = o.in_slot;
    a.C = o.out; a.ldc = n->hw_now; a.sC = (long)o.cout * n->hw_now; a.omax = o.omax;
    a.Chi = static_cast<_Float16*>(o.ohi); a.Clo = static_cast<_Float16*>(o.olo); a.ldnc = n->hw_now; a.sCp = (long)o.cout * n->hw_now;
    a.cw = o.w->winf; a.cb = o.cb; a.cslot = o.cslot; a.cinb = o.cinb; a.rmax = o.rmax;
    a.part = reinterpret_cast<float4*>(o.part);
    a.bias = o.bias; a.sbias = o.sbias;
    a.R = o.R; a.ldr = n->hw_now; a.sR = o.r_bs; a.rsc = o.rsc; a.rsh = o.rsh; a.srs = o.rsc ? o.cout : 0;
    a.M = o.cout; a.N = (int)n->hw_now; a.K = o.cin; a.nbatch = batch;
    a.act = o.act == ACT_GELU ? ACT_GELU_FAST : o.act;
    HIP_TRY(launch_gemm_f16x3_packed(a, s));
    return ACE_OK;
}

// hipEvent stage timer: one event after each launch group; the elapsed time between consecutive events
// is charged to the stage that just ended (the reference's CUDATimer children: fme/core/benchmark/timer.py:105-168,
// conditional_sfno/sfnonet.py:388-437, s2convolutions.py:372-431).
enum Stage {
    ST_ENCODER = 0, ST_NORM0, ST_DFT_FWD, ST_LEGENDRE_FWD, ST_CONTRACT, ST_LEGENDRE_INV, ST_DFT_INV, ST_INNER_SKIP,
    ST_NORM1, ST_MLP_FC1, ST_MLP_FC2, ST_DECODER, ST_COUNT
};
static const char* kStageNames[ST_COUNT] = {
    "encoder", "norm0_stats", "forward_transform.dft", "forward_transform.legendre", "dhconv",
    "inverse_transform.legendre", "inverse_transform.dft", "inner_skip+activation", "norm1_stats", "mlp.fc1",
    "mlp.fc2+outer_skip", "decoder"};

struct StageTimer {
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    std::vector<int> stage;
    hipError_t err = hipSuccess;
    explicit StageTimer(hipStream_t st) : s(st) { push(-1); }
    void push(int st) {
        hipEvent_t e;
        hipError_t r = hipEventCreate(&e);
        if (r == hipSuccess) r = hipEventRecord(e, s);
        if (r != hipSuccess && err == hipSuccess) err = r;
        ev.push_back(e);
        stage.push_back(st);
    }
    hipError_t finish(float* ms, int* calls) {
        hipError_t r = hipStreamSynchronize(s);
        if (r != hipSuccess) return r;
        for (int i = 0; i < ST_COUNT; ++i) { ms[i] = 0.f; if (calls) calls[i] = 0; }
        for (size_t i = 1; i < ev.size(); ++i) {
            float t = 0.f;
            r = hipEventElapsedTime(&t, ev[i - 1], ev[i]);
            if (r != hipSuccess) return r;
            ms[stage[i]] += t;
            if (calls) calls[stage[i]] += 1;
        }
        return err;
    }
    ~StageTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
};
#define MARK(st) do { if (tm) tm->push(st); } while (0)

static int forward_impl(ace_sfno* n, const float* in, float* out, int B, hipStream_t s, StageTimer* tm = nullptr,
                        const float* noise = nullptr) {
    const ace_sfno_config& c = n->cfg;
    const int C = n->C, Cin = c.in_chans, act = c.activation_function;
    long HW = n->HW;                 // resolution of the launches being enqueued: the data grid, or (scale_factor != 1, between the
    n->hw_now = HW;                  // first filter's inverse transform and the last one's) the inner grid - see the blocks
    const long N2 = (long)B * 2 * C;
    long actB = (long)C * HW;        // per-sample stride of a C-channel activation
    auto W = [&](const std::string& name) { return n->w(name); };
    // dynamic-range slots of the f16x3 engine (AMAX_SHARDS words each): 0 network input, 1..7 encoder/decoder hidden,
    // 8 + i block input h_i, 16 + 12 i + {0 X, 1 D, 2 E, 3 N0, 4 T, 5 N1, 6 U, 7 Y, 8 folded skip weight, 9 folded fc1 weight}
    const bool f16 = c.precision == 1;
    unsigned* amax = reinterpret_cast<unsigned*>(n->amax.p);
    auto slot = [&](int k) -> unsigned* { return f16 ? amax + (size_t)k * AMAX_SHARDS : nullptr; };
    if (f16) {
        HIP_TRY(launch_zero_u32(amax, (long)n->amax.n, s));
        HIP_TRY(launch_absmax(in, (long)B * Cin * HW, slot(0), s));
    }
    if (c.num_layers + 8 > 16) { /* block-input slots 8..15 are shared cyclically for deep nets */ }
    auto hslot = [&](int i) { return slot(8 + (i % 8)); };

    // Residual stream as planes only (see the block loop): the condition is needed by the encoder already
    const bool stream_ok = n->sw.planes_stream && f16 && !n->taps_on && n->plan_data == n->plan_lg.get() && C % 16 == 0 &&
                           !n->plan_lg->sw.no_fft && dft_fft_has_width(n->W);
    // ---- encoder (sfnonet.py:721-733): [conv+bias, act] x encoder_layers, conv (no bias), + pos_embed
    const float* cur = in;
    long cur_bs = (long)Cin * HW;
    int curC = Cin;
    float* ping[2] = {n->Y.p, n->T.p};
    // Will the last encoder convolution run on conv_ws.hip (below)?  Then a ONE-hidden-layer encoder (the ACE2 shape) never
    // materialises its hidden activation as fp32: the network input is packed to P-format planes (in_chans rounded up to a k-group,
    // 11 MB) and the first convolution runs on the packed-operand engine, whose epilogue writes GELU(W x + b) straight as the
    // planes that convolution reads (r03: fp32 write + pack pass = 55 + 41 us per step).
    const bool enc_ws_ok = [&] {
        const Weight& we = *n->weights[n->index.at("encoder." + std::to_string(2 * c.encoder_layers) + ".weight")];
        const bool block0_fused = f16 && c.normalization_layer == 1 && c.use_mlp && n->P2.p && packed_ok(n, C) && n->hid % 8 == 0 &&
                                  (n->plan_data == n->plan_lg.get() || c.num_layers == 1);
        return !n->sw.no_enc_ws && block0_fused && c.encoder_layers >= 1 && c.pos_embed && we.frag0.p && n->zero_c.p &&
               n->pe_slot.p && n->part.p && conv_ws_eligible(C, C, HW, 2, n->sw.conv_ws_roles);
    }();
    bool enc_hidden_planes = false;   // n->P holds the last hidden activation of the encoder as planes scaled from slot(encoder_layers)
    if (enc_ws_ok && c.encoder_layers == 1 && n->inP.p && !n->sw.no_enc_pk) {
        const Weight& w0 = *n->weights[n->index.at("encoder.0.weight")];
        const Weight& b0 = *n->weights[n->index.at("encoder.0.bias")];
        const int cin8 = (Cin + 7) / 8 * 8;
        if (w0.thi.p && w0.pitch >= cin8) {
            _Float16* Ih = reinterpret_cast<_Float16*>(n->inP.p);
            _Float16* Il = Ih + (size_t)n->Bmax * cin8 * HW;
            HIP_TRY(launch_pack_pformat(in, HW, (long)Cin * HW, Cin, (int)HW, B, nullptr, nullptr, 0, slot(0), Ih, Il, HW, (long)cin8 * HW, s));
            _Float16* Th = reinterpret_cast<_Float16*>(n->P.p);
            _Float16* Tl = Th + (size_t)n->Bmax * C * HW;
            ACE_TRY(conv_pk(n, w0, b0.buf.p, Ih, Il, cin8, slot(0), nullptr, C, nullptr, 0, nullptr, nullptr, act, B, s, nullptr, Th, Tl,
                            b0.absmax, slot(1)));
            enc_hidden_planes = true;
            curC = C;
        }
    }
    for (int j = 0; j < c.encoder_layers && !enc_hidden_planes; ++j) {
        const std::string p = "encoder." + std::to_string(2 * j);
        ACE_TRY(conv(n, conv_weight(n, p + ".weight", p + ".bias"), cur, cur_bs, curC, nullptr, 0, -1, ping[j & 1], C,
                     nullptr, 0, nullptr, nullptr, act, B, s, nullptr, nullptr, slot(j == 0 ? 0 : j), nullptr,
                     slot(1 + j)));
        cur = ping[j & 1]; cur_bs = actB; curC = C;
    }
    float* h = n->h0.p;
    float* hn = n->h1.p;
    // The last encoder convolution (C -> C, no bias, + pos_embed) on the weight-stationary strip kernel when block 0 takes the
    // packed path: its epilogue then writes h0 as fp32 AND as P-format planes with the row statistics of norm0, so block 0
    // starts like every other block (no pack pass, no statistics pass over h0).
    bool enc_planes = false, enc_planes_only = false;
    int enc_nparts = 0;          // statistics partials per row the last encoder convolution wrote
    {
        const Weight& we = *n->weights[n->index.at("encoder." + std::to_string(2 * c.encoder_layers) + ".weight")];
        const bool enc_off = n->sw.no_enc_ws;
        const bool block0_fused = f16 && c.normalization_layer == 1 && c.use_mlp && n->P2.p && packed_ok(n, C) && n->hid % 8 == 0 &&
                                  (n->plan_data == n->plan_lg.get() || c.num_layers == 1);
        if (!enc_off && block0_fused && c.encoder_layers >= 1 && c.pos_embed && curC == C && we.frag0.p && n->zero_c.p &&
            n->pe_slot.p && n->part.p && conv_ws_eligible(C, C, HW, 2, n->sw.conv_ws_roles)) {
            _Float16* Th = reinterpret_cast<_Float16*>(n->P.p);
            _Float16* Tl = Th + (size_t)n->Bmax * C * HW;
            _Float16* Hh = reinterpret_cast<_Float16*>(n->P2.p);
            _Float16* Hl = Hh + (size_t)n->Bmax * C * HW;
            if (!enc_hidden_planes) ACE_TRY(pack_act(n, cur, cur_bs, C, nullptr, nullptr, slot(c.encoder_layers), Th, Tl, B, s));
            ConvStripArgs k;
            k.Xhi = Th; k.Xlo = Tl; k.ldn = HW; k.sX = (long)C * HW; k.xslot = slot(c.encoder_layers);
            k.A = reinterpret_cast<const _Float16*>(we.frag0.p); k.sA = 0; k.ascale = we.ascale;
            k.bias = n->zero_c.p; k.sbias = 0;
            k.R = W("pos_embed"); k.sR = 0;
            // fp32 copy of h0: only where something reads it - with the residual stream as planes (stream_ok below) block 0 takes its
            // input from the planes everywhere (longitude FFT, inner skip, outer-skip residual) and the 100 MB write is dropped
            // (block 0 is on the fused path here - block0_fused above - and its fc2 must be the conv_ws one, whose residual comes from the planes)
            if (stream_ok && c.num_layers >= 1) {
                const Weight& w2b0 = *n->weights[n->index.at("blocks.0.mlp.fwd.2.weight")];
                enc_planes_only = w2b0.frag0.p && C <= 1024 && conv_ws_eligible(n->hid, C, HW, 2, n->sw.conv_ws_roles);
            }
            if (!enc_planes_only) { k.Cf = h; k.sCf = actB; }
            k.Chi = Hh; k.Clo = Hl; k.sCp = (long)C * HW; k.cslot = hslot(0);
            k.cw = we.winf; k.cb = 0.f; k.rmax = reinterpret_cast<const unsigned*>(n->pe_slot.p);
            k.C = C; k.M = C; k.HW = (int)HW; k.nbatch = B; k.act = ACT_NONE;
            k.part = reinterpret_cast<float4*>(n->part.p); k.nstrips32 = conv_ws_stat_parts(k); enc_nparts = k.nstrips32;
            HIP_TRY(launch_conv_ws(k, s));
            enc_planes = true;
        }
    }
    if (!enc_planes)
    ACE_TRY(conv(n, conv_weight(n, "encoder." + std::to_string(2 * c.encoder_layers) + ".weight", ""), cur, cur_bs, curC,
                 nullptr, 0, -1, h, C, c.pos_embed ? W("pos_embed") : nullptr, 0, nullptr, nullptr, ACT_NONE, B, s,
                 nullptr, nullptr, slot(c.encoder_layers), nullptr, hslot(0)));
    MARK(ST_ENCODER);
    if (n->taps_on) HIP_TRY(hipMemcpyAsync(n->taps[0].p, h, sizeof(float) * B * actB, hipMemcpyDeviceToDevice, s));

    float* sc0 = n->stats.p;
    float* sh0 = sc0 + (size_t)n->Bmax * C;
    float* sc1 = sh0 + (size_t)n->Bmax * C;
    float* sh1 = sc1 + (size_t)n->Bmax * C;
    const bool norm = c.normalization_layer == 1;
    // NoiseConditionedSFNO: conditional layer norms are materialised in place (fp32) with their true max in the slot of
    // the instance-norm bound; the convolutions then run on the packed-operand engine (one pack pass per conv input)
    const bool cln = c.normalization_layer == 2;
    // "layer_norm": nn.LayerNorm over (H, W) with an (H, W) affine (sfnonet.py:584-592) - a per-PIXEL affine does not fold into the
    // consumers' weights, so it is materialised in place like the conditional norms (statistics + apply in one launch per norm,
    // kernels.hip: spatial_layer_norm_kernel) and the convolutions run on the packed-operand engine
    const bool lnrm = c.normalization_layer == 3;
    const float* skip_in = in;               // second source of the big-skip concat
    const unsigned* skip_in_slot = slot(0);
    // one conditional layer norm (layers.py:245-318): a single MFMA pass (cln_mfma.hip) where the shape allows it and the
    // conditioning weights were packed at upload, else statistics + apply (kernels.hip)
    unsigned* noise_slot = slot(16 + 12 * c.num_layers);
    if (cln && f16 && noise) HIP_TRY(launch_absmax(noise, (long)B * c.noise_embed_dim * HW, noise_slot, s));
    // planes (optional): where the single-pass kernel can, it also writes y as the P-format operand of the packed convolutions
    // (scaled by a bound it publishes to omax) - *planes_done says so, the caller then skips its pack pass; with
    // fp32_needed = false that is the ONLY output (y untouched).
    auto cond_norm = [&](const float* x, float* y, const std::string& q, int Cc, unsigned* omax, _Float16* phi = nullptr,
                         _Float16* plo = nullptr, bool fp32_needed = true, bool* planes_done = nullptr) -> int {
        const Weight* wsc = n->find(q + "W_scale_2d.weight");
        const Weight* wbi = n->find(q + "W_bias_2d.weight");
        const Weight* wga = n->find(q + "norm.weight");
        const Weight* wbe = n->find(q + "norm.bias");
        if (planes_done) *planes_done = false;
        ClnMfmaArgs a;
        a.x = x; a.y = y; a.sx = (long)Cc * HW; a.C = Cc; a.HW = HW; a.nbatch = B; a.eps = 1e-5f; a.omax = omax;
        a.gamma = W(q + "norm.weight"); a.beta = W(q + "norm.bias");
        if (f16 && noise && wsc && wbi && wsc->frag0.p && wbi->frag0.p && !n->sw.no_cln_mfma) {
            a.cond = noise; a.scond = (long)c.noise_embed_dim * HW; a.J = c.noise_embed_dim; a.cslot = noise_slot;
            a.As = reinterpret_cast<const _Float16*>(wsc->frag0.p); a.Ab = reinterpret_cast<const _Float16*>(wbi->frag0.p);
            a.ascale_s = wsc->ascale; a.ascale_b = wbi->ascale;
            if (phi && plo && omax && !n->sw.no_cln_planes && cln_mfma_planes_ok(a) && (!a.gamma || (wga && wbe))) {
                a.Phi = phi; a.Plo = plo; a.sP = (long)Cc * HW;
                a.gmax = a.gamma ? wga->absmax : 1.f; a.bmax = a.gamma ? wbe->absmax : 0.f;
                a.ws_inf = wsc->winf; a.wb_inf = wbi->winf;
                if (!fp32_needed) a.y = nullptr;
            }
            if (cln_mfma_eligible(a)) {
                HIP_TRY(launch_cln_mfma(a, s));
                if (planes_done) *planes_done = a.Phi != nullptr;
                return ACE_OK;
            }
            a.Phi = a.Plo = nullptr; a.y = y;
        }
        HIP_TRY(launch_cond_layer_norm(x, noise, a.gamma, a.beta, W(q + "W_scale_2d.weight"), W(q + "W_bias_2d.weight"), 1e-5f,
                                       n->cln_stats.p, y, B, Cc, c.noise_embed_dim, HW, s, omax));
        return ACE_OK;
    };
    if (n->plan_res) {   // residual = residual_filter_up(residual_filter_down(x)) (sfnonet.py:715-716), exact fp32; its range into slot 7
        const long NR = (long)B * 2 * Cin;
        ACE_TRY(run_dft_forward(*n->plan_res, in, nullptr, nullptr, n->resX.p, B, Cin, s));
        ACE_TRY(run_legendre_forward(*n->plan_res, n->resX.p, n->resD.p, NR, s));
        ACE_TRY(run_legendre_inverse(*n->plan_res, n->resD.p, n->resX.p, NR, s));
        ACE_TRY(run_dft_inverse(*n->plan_res, n->resX.p, nullptr, n->inF.p, B, Cin, s, slot(7)));
        skip_in = n->inF.p;
        skip_in_slot = slot(7);
    }
    if (cln && c.big_skip && c.normalize_big_skip) {
        ACE_TRY(cond_norm(in, n->inN.p, "norm_big_skip.", Cin, slot(7)));
        skip_in = n->inN.p;
        skip_in_slot = slot(7);
    }

    // fused-norm state of the packed-operand path: does P2 hold the block input in P format / `part_h` its statistics?
    bool have_ph = enc_planes, have_hstats = enc_planes;
    int h_nparts = enc_planes ? enc_nparts : 0;   // statistics partials per row in part_h (depends on which kernel produced them)
    float* part_h = n->part.p;
    float* part_t = n->part.p ? n->part.p + (size_t)n->Bmax * n->nstrips * C * 4 : nullptr;
    _Float16* PAh = reinterpret_cast<_Float16*>(n->P.p);                  // T planes / pack fallback
    _Float16* PAl = PAh ? PAh + (size_t)n->Bmax * C * HW : nullptr;
    _Float16* PBh = reinterpret_cast<_Float16*>(n->P2.p);                 // h planes
    _Float16* PBl = PBh ? PBh + (size_t)n->Bmax * C * HW : nullptr;

    // Residual stream as planes only: once a block's fc2 (conv_ws mode 4) has written h' as P-format planes, the next block reads
    // them everywhere - longitude FFT, inner skip, outer-skip residual - and no fp32 copy of h' exists (fc2: 520 -> 420 MB).
    bool h_planes_only = enc_planes && enc_planes_only;   // the encoder wrote h0 as planes only (implies stream_ok)
    // ---- blocks (sfnonet.py:217-252)
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        const ace_sht_plan& fwd = (i == 0) ? *n->plan_data : *n->plan_lg;
        const ace_sht_plan& inv = (i == c.num_layers - 1) ? *n->plan_data : *n->plan_lg;
        const bool scale_residual = (&fwd != &inv);  // grids differ (s2convolutions.py:82-86)

        // norm0 as an affine applied on load by every consumer (never materialised)
        const int sb = 16 + 12 * i;  // slot base of this block
        if (f16 && i + 1 >= 8) HIP_TRY(launch_zero_u32(hslot(i + 1), AMAX_SHARDS, s));
        const float *a0 = nullptr, *b0 = nullptr;
        bool n0_planes = false, n1_planes = false;   // the conditional norms wrote their output as the packed convolutions' operand
        if (cln) {   // x_norm = CLN0(h; noise), in place: the block never needs the un-normalised h again (sfnonet.py:388-437)
            const bool pk_ahead = f16 && !scale_residual && packed_ok(n, C) && (!c.use_mlp || n->hid % 8 == 0);   // = `pk` below
            _Float16* P0 = reinterpret_cast<_Float16*>(n->P.p);
            ACE_TRY(cond_norm(h, h, p + "norm0.", C, slot(sb + 3), pk_ahead ? P0 : nullptr,
                              pk_ahead ? P0 + (size_t)n->Bmax * C * HW : nullptr, true, &n0_planes));
            MARK(ST_NORM0);
        }
        if (lnrm) {   // x_norm = LayerNorm_(H, W)(h), in place
            HIP_TRY(launch_spatial_layer_norm(h, W(p + "norm0.weight"), W(p + "norm0.bias"), 1e-6f, (long)B * C, HW, h, slot(sb + 3), s));
            MARK(ST_NORM0);
        }
        if (norm) {
            if (have_hstats)   // statistics of h came out of the previous block's fc2 epilogue
                HIP_TRY(launch_instnorm_finalize(reinterpret_cast<const float4*>(part_h), h_nparts, B, C, HW,
                                                 W(p + "norm0.weight"), W(p + "norm0.bias"), 1e-6f, sc0, sh0, slot(sb + 3), s));
            else
                HIP_TRY(launch_instnorm_stats(h, W(p + "norm0.weight"), W(p + "norm0.bias"), 1e-6f, B, C, HW, sc0, sh0, s,
                                              slot(sb + 3)));
            a0 = sc0; b0 = sh0;
            MARK(ST_NORM0);
        }
        // (Round 4 tried packing the inner skip's folded weights on a side stream beside the SHT chain - they depend on norm0's
        // statistics only.  Same box, whole step in a hipGraph: 5.76 -> 5.86 ms; the fork / join edges cost more than the 7 us
        // kernel they hide.  profiles/r04_s3_kdur_side_pack_*.txt)
        // spectral filter (s2convolutions.py:162-197): SHT -> contraction -> inverse SHT + bias
        unsigned *xmax = slot(sb + 0), *dmax = slot(sb + 1), *emax = slot(sb + 2);
        // The folded weights of this block's inner skip (W diag(a0), bs + W b0: conv_ws.hip reads them as packed fragments) depend on
        // norm0's affine only: where that convolution will run on the weight-stationary kernel (the conditions of `ws_skip` below),
        // the ~300 workgroups that pack them ride in the forward FFT launch instead of a dependent 7 us launch of their own.
        PackFragArgs skip_pack;
        bool skip_pack_rode = false, skip_pack_wanted = false;
        if (n->sw.fused_pack && f16 && norm && !scale_residual && have_ph && c.use_mlp && n->P2.p && packed_ok(n, C) && n->hid % 8 == 0 &&
            c.activation_function == ACT_GELU && n->Wq0.p) {
            const Weight& wsk = *n->weights[n->index.at(p + "inner_skip.weight")];
            const Weight& bsk = *n->weights[n->index.at(p + "inner_skip.bias")];
            if (wsk.frag0.p && conv_ws_eligible(C, C, HW, 0, n->sw.conv_ws_roles)) {
                skip_pack.W = wsk.buf.p; skip_pack.ldw = wsk.pitch; skip_pack.O = C; skip_pack.I = C; skip_pack.order = 0;
                skip_pack.a = a0; skip_pack.wmax = wsk.wabs; skip_pack.scale_static = 1.f; skip_pack.wslot = slot(sb + 8);
                skip_pack.dst = reinterpret_cast<_Float16*>(n->Wq0.p); skip_pack.sDst = (long)C * C * 2;
                skip_pack.b = b0; skip_pack.bias = bsk.buf.p; skip_pack.bf = n->bf0.p; skip_pack.nsamples = B;
                skip_pack_wanted = true;
            }
        }
        const PackFragArgs* ride = skip_pack_wanted ? &skip_pack : nullptr;
#ifdef ACE_MEASUREMENT_SWITCHES
        bool forked = false, packed_early = false;
        if (n->skip_fork && skip_pack_wanted && !tm) {
            // the skip GEMM without its epilogue (mode 7: S = W diag(a0) h + bias, fp32 into the unused T buffer) beside the chain
            const Weight& wsk = *n->weights[n->index.at(p + "inner_skip.weight")];
            hipStream_t q = n->skip_fork == 1 ? n->side_stream : s;
            if (n->skip_fork == 1) { HIP_TRY(hipEventRecord(n->ev_fork, s)); HIP_TRY(hipStreamWaitEvent(q, n->ev_fork, 0)); }
            HIP_TRY(launch_pack_conv_frag(wsk.buf.p, wsk.pitch, C, C, 0, a0, wsk.wabs, 1.f, slot(sb + 8), n->Wq0.p, (long)C * C * 2, B, q,
                                          b0, skip_pack.bias, n->bf0.p));
            ConvStripArgs k;
            k.Xhi = PBh; k.Xlo = PBl; k.ldn = HW; k.sX = (long)C * HW; k.xslot = hslot(i);
            k.A = reinterpret_cast<const _Float16*>(n->Wq0.p); k.sA = (long)C * C * 2; k.aslot = slot(sb + 8);
            k.bias = n->bf0.p; k.sbias = C;
            k.Cf = n->T.p; k.sCf = actB;
            k.C = C; k.M = C; k.HW = (int)HW; k.nbatch = B; k.act = ACT_NONE;
            HIP_TRY(launch_conv_ws(k, q));
            forked = n->skip_fork == 1;
            ride = nullptr;   // (packed above)
            packed_early = true;
        }
#endif
        if (h_planes_only)   // the block input exists as planes only (written by the previous block's fc2, mode 4)
            ACE_TRY(run_dft_forward(fwd, nullptr, a0, b0, n->X.p, B, C, s, xmax, PBh, PBl, (long)C * HW, hslot(i), ride, &skip_pack_rode));
        else
            ACE_TRY(run_dft_forward(fwd, h, a0, b0, n->X.p, B, C, s, xmax, nullptr, nullptr, 0, nullptr, ride, &skip_pack_rode));
#ifdef ACE_MEASUREMENT_SWITCHES
        if (packed_early) skip_pack_rode = true;
#endif
        MARK(ST_DFT_FWD);
        // packed dhconv: D goes from the Legendre epilogue to the filter GEMM as fp16 planes (never fp32)
        const bool dplanes = f16 && !n->sw.no_pk_sht && c.operator_type == 1 && n->wx_hi[i].p && !scale_residual && fwd.f16 &&
                             N2 % 4 == 0 && (2 * C) % 32 == 0;
        ACE_TRY(run_legendre_forward(fwd, n->X.p, n->D.p, N2, s, xmax, dmax, dplanes));
        MARK(ST_LEGENDRE_FWD);
        const float* res = h;           // residual = x_norm, applied as (h, a0, b0)
        const float *ra = a0, *rb = b0;
        if (scale_residual) {           // residual = inverse(forward(x_norm)) on the output grid
            ACE_TRY(run_legendre_inverse(inv, n->D.p, n->X.p, N2, s, dmax));
            MARK(ST_LEGENDRE_INV);
            ACE_TRY(run_dft_inverse(inv, n->X.p, nullptr, n->R.p, B, C, s));
            MARK(ST_DFT_INV);
            res = n->R.p; ra = nullptr; rb = nullptr;
        }
        if (c.operator_type == 1) {  // dhconv (contractions.py:183-195): per l, (m,b) x 2C times real 2C x 2C
            GemmArgs g;
            g.A = n->D.p; g.lda = 2 * C; g.sA = (long)n->Mm * N2;
            g.B = n->wx[i].p; g.ldb = 2 * C; g.sB = (long)2 * C * 2 * C;
            g.C = n->E.p; g.ldc = 2 * C; g.sC = (long)n->Mm * N2;
            g.M = n->Mm * B; g.N = 2 * C; g.K = 2 * C; g.nbatch = n->L; g.a_kpad = 2 * C;
            g.tri = TRI_ROWS_LE_BATCH; g.trimul = B;
            g.omax = emax;
            DhconvStripArgs ds;
            if (dplanes && n->wx_compact[i]) {
                ds.Dhi = reinterpret_cast<const _Float16*>(n->D.p);
                ds.Dlo = ds.Dhi + (size_t)n->L * n->Mm * N2;
                ds.sD = (long)n->Mm * N2; ds.amax = dmax;
                ds.Whi = reinterpret_cast<const _Float16*>(n->wx_hi[i].p);
                ds.Wlo = reinterpret_cast<const _Float16*>(n->wx_lo[i].p);
                ds.sW = (long)2 * C * C; ds.bscale = n->wx_scale[i];
                ds.E = n->E.p; ds.sE = (long)n->Mm * N2; ds.omax = emax;
                ds.C = C; ds.L = n->L; ds.Mrows = n->Mm * B; ds.trimul = B;
                ds.groups = cln ? c.filter_num_groups : 1;
                if (n->wx_native[i]) { ds.kstore = C / ds.groups; ds.sW = (long)2 * ds.kstore * C; }
                if ((int)n->dh_units.size() >= B) { ds.units = reinterpret_cast<const int*>(n->dh_units[B - 1].p); ds.units_per_xcd = n->dh_units_per_xcd[B - 1]; }
            }
            if (n->wx_native[i] && !(dplanes && ds.units && dhconv_strip_eligible(ds)))
                return fail(ACE_ERR_STATE, "grouped filter stored as diagonal blocks, but the strip kernel cannot run this launch");
            if (dplanes && n->wx_compact[i] && !n->sw.no_dhconv_strip && ds.units && dhconv_strip_eligible(ds)) {
                HIP_TRY(launch_dhconv_strip(ds, s));
            } else if (dplanes) {
                Gemm4Args a;
                a.Ahi = reinterpret_cast<const _Float16*>(n->D.p);
                a.Alo = a.Ahi + (size_t)n->L * n->Mm * N2;
                a.lda = 2 * C; a.sA = (long)n->Mm * N2; a.amax = dmax;
                a.Bhi = reinterpret_cast<const _Float16*>(n->wx_hi[i].p);
                a.Blo = reinterpret_cast<const _Float16*>(n->wx_lo[i].p);
                a.ldn = 2 * C; a.sB = (long)2 * C * 2 * C; a.bscale = n->wx_scale[i];
                if (n->wx_compact[i]) { a.cplx = C; a.ldn = C; a.sB = (long)2 * C * C; a.tile = 1; }
                a.C = n->E.p; a.ldc = 2 * C; a.sC = (long)n->Mm * N2; a.omax = emax;
                a.M = n->Mm * B; a.N = 2 * C; a.K = 2 * C; a.nbatch = n->L;
                a.tri = TRI_ROWS_LE_BATCH; a.trimul = B;
                HIP_TRY(launch_gemm_f16x3_packed(a, s));
            } else if (f16 && n->wx_hi[i].p && n->wx_compact[i]) {
                return fail(ACE_ERR_STATE, "compact dhconv operand without the packed path");
            } else if (f16 && n->wx_hi[i].p) {
                HIP_TRY(launch_gemm_f16x3_adyn(g, n->wx_hi[i].p, n->wx_lo[i].p, 2 * C, (long)2 * C * 2 * C, n->wx_scale[i],
                                               dmax, emax, s));
            } else {
                HIP_TRY(launch_gemm(g, s));
            }
        } else {
            HIP_TRY(launch_contract_diagonal(n->D.p, W(p + "filter.filter.weight"), n->E.p, B, C, C, n->L, n->Mm, s, emax));
        }
        MARK(ST_CONTRACT);
        ACE_TRY(run_legendre_inverse(inv, n->E.p, n->X.p, N2, s, emax));
        MARK(ST_LEGENDRE_INV);
        ACE_TRY(run_dft_inverse(inv, n->X.p, W(p + "filter.filter.bias"), n->Y.p, B, C, s, slot(sb + 7)));
        MARK(ST_DFT_INV);
        // everything behind the filter works on ITS output grid (sfnonet.py:217-252: the skips, norm1 and the MLP see what the
        // inverse transform produced) - the inner grid after the first block of a scale_factor != 1 net, the data grid after the last
        HW = (long)inv.nlat * inv.nlon; actB = (long)C * HW; n->hw_now = HW;
#ifdef ACE_MEASUREMENT_SWITCHES
        if (forked) { HIP_TRY(hipEventRecord(n->ev_join, n->side_stream)); HIP_TRY(hipStreamWaitEvent(s, n->ev_join, 0)); }
#endif

        // x = act(filter + inner_skip(residual))   (sfnonet.py:229-232)
        ConvW wskip = conv_weight(n, p + "inner_skip.weight", p + "inner_skip.bias");
        // B of the inner skip: the normalised block input (bound from the norm statistics) or, without a norm, the raw
        // block input; the spectrally round-tripped residual of mixed-grid blocks has no range slot -> fp32 engine
        const unsigned* skip_max = scale_residual ? nullptr : ((norm || cln || lnrm) ? slot(sb + 3) : hslot(i));
        const bool skip_f16 = f16 && skip_max != nullptr;
        const bool pk = skip_f16 && packed_ok(n, C) && (!c.use_mlp || n->hid % 8 == 0);
        _Float16* Ph = reinterpret_cast<_Float16*>(n->P.p);
        _Float16* Pl = Ph + (size_t)n->Bmax * C * HW;
        const bool fused = pk && norm && c.use_mlp && n->P2.p != nullptr;
        const bool have_ph_in = have_ph;   // P2 holds the RAW block input as planes (bound in hslot(i)), from a producer epilogue
        if (fused) {
            // Packed-operand path with the instance norms fused away: every conv input is P-format planes written by
            // its producer's epilogue, the norm affine is folded into the consumer's weights, the norm statistics come
            // out of the producer's epilogue.  No pass over an activation other than the GEMMs themselves.
            const Weight& ws = *n->weights[n->index.at(p + "inner_skip.weight")];
            const Weight& bsw = *n->weights[n->index.at(p + "inner_skip.bias")];
            const Weight& w1 = *n->weights[n->index.at(p + "mlp.fwd.0.weight")];
            const Weight& b1w = *n->weights[n->index.at(p + "mlp.fwd.0.bias")];
            const Weight& w2 = *n->weights[n->index.at(p + "mlp.fwd.2.weight")];
            const Weight& b2w = *n->weights[n->index.at(p + "mlp.fwd.2.bias")];
            const long cp = ws.pitch;
            _Float16* F0h = reinterpret_cast<_Float16*>(n->Wp0.p);
            const long f0s = (long)((C + 15) / 16 * 16) * cp;
            _Float16* F0l = F0h + (size_t)n->Bmax * f0s;
            _Float16* F1h = reinterpret_cast<_Float16*>(n->Wp1.p);
            const long f1s = (long)((n->hid + 15) / 16 * 16) * cp;
            _Float16* F1l = F1h + (size_t)n->Bmax * f1s;
            _Float16* Uh = reinterpret_cast<_Float16*>(n->U.p);
            _Float16* Ul = Uh + (size_t)n->Bmax * n->hid * HW;
            const bool last = (i + 1 == c.num_layers);
            // inner skip: T = act(Y + Ws norm0(h) + bs), written as P-format planes only (+ its row statistics)
            PkOpts o;
            o.w = &ws; o.cin = C; o.cout = C; o.act = act;
            o.R = n->Y.p; o.r_bs = actB;
            o.ohi = PAh; o.olo = PAl; o.cb = bsw.absmax; o.cslot = slot(sb + 4); o.cinb = slot(sb + 3); o.rmax = slot(sb + 7);
            o.part = part_t;
            const bool gelu = c.activation_function == ACT_GELU;
            const int roles = n->sw.conv_ws_roles;
            // which of the three convolutions run on the weight-stationary kernel (conv_ws.hip); the others on the tile engine
            const bool ws_skip = gelu && n->Wq0.p && ws.frag0.p && conv_ws_eligible(C, C, HW, 0, roles);
            const bool wl_fc1 = gelu && n->Wq1.p && n->sw.conv_wl && conv_wl_eligible(C, n->hid, HW);
            const bool ws_fc1 = wl_fc1 || (gelu && n->Wq1.p && conv_ws_eligible(C, n->hid, HW, 1, roles));
            const bool ws_fc2 = w2.frag0.p && C <= 1024 && conv_ws_eligible(n->hid, C, HW, 2, roles);
            if (have_ph) {   // h planes from the previous fc2; the affine goes into the weights
                if (!ws_skip)
                    HIP_TRY(launch_fold_affine_f16(ws.buf.p, ws.pitch, ws.wabs, ra, rb, bsw.buf.p, F0h, F0l, n->bf0.p, B, C, C, cp,
                                                   f0s, slot(sb + 8), s));
                o.fhi = F0h; o.flo = F0l; o.fslot = slot(sb + 8); o.f_stride = f0s;
                o.bias = n->bf0.p; o.sbias = C;
                o.bhi = PBh; o.blo = PBl; o.in_slot = hslot(i);
            } else {         // first block: normalise + split h in one pass
                ACE_TRY(pack_act(n, res, actB, C, ra, rb, skip_max, PBh, PBl, B, s));
                o.bias = bsw.buf.p;
                o.bhi = PBh; o.blo = PBl; o.in_slot = skip_max;
            }
            int t_nparts = gemm4_strips(C, (int)HW);
            if (ws_skip) {
                ConvStripArgs k;
                k.Xhi = PBh; k.Xlo = PBl; k.ldn = HW; k.sX = (long)C * HW; k.xslot = o.in_slot;
                if (have_ph) {
                    _Float16* Q0 = reinterpret_cast<_Float16*>(n->Wq0.p);
                    if (!skip_pack_rode)   // (packed by rider workgroups of this block's forward FFT otherwise)
                        HIP_TRY(launch_pack_conv_frag(ws.buf.p, ws.pitch, C, C, 0, ra, ws.wabs, 1.f, slot(sb + 8), Q0, (long)C * C * 2, B, s,
                                                      rb, bsw.buf.p, n->bf0.p));
                    k.A = Q0; k.sA = (long)C * C * 2; k.aslot = slot(sb + 8);
                } else {
                    k.A = reinterpret_cast<const _Float16*>(ws.frag0.p); k.sA = 0; k.ascale = ws.ascale;
                }
                k.bias = o.bias; k.sbias = o.sbias;
                k.R = n->Y.p; k.sR = actB;
                k.cw = ws.winf; k.cb = bsw.absmax; k.cinb = slot(sb + 3); k.rmax = slot(sb + 7);
                k.Chi = PAh; k.Clo = PAl; k.sCp = (long)C * HW; k.cslot = slot(sb + 4);
                k.part = reinterpret_cast<float4*>(part_t);
                k.C = C; k.M = C; k.HW = (int)HW; k.nbatch = B; k.act = ACT_GELU;
                k.nstrips32 = conv_ws_stat_parts(k);   // one partial per pixel group (statistics accumulated in registers)
                HIP_TRY(launch_conv_ws(k, s));
                t_nparts = k.nstrips32;
            } else {
                ACE_TRY(conv_pk2(n, o, B, s));
            }
            MARK(ST_INNER_SKIP);
            // norm1 statistics -> affine -> folded fc1 weights
            HIP_TRY(launch_instnorm_finalize(reinterpret_cast<const float4*>(part_t), t_nparts, B, C, HW,
                                             W(p + "norm1.weight"), W(p + "norm1.bias"), 1e-6f, sc1, sh1, slot(sb + 5), s));
            if (ws_fc1) {
                _Float16* Q1 = reinterpret_cast<_Float16*>(n->Wq1.p);
                const long q1s = (long)n->hid * C * 2;
                // W1 diag(a1), b1 + W1 b1' folded in the convolution's own prologue (conv_wl.hip at the ACE2 width: +1.4 us against a 6.7 us launch)
                const bool pfold = n->sw.fused_pack && wl_fc1 && C == 384 && w1.pitch % 4 == 0;
                if (!pfold)
                    HIP_TRY(launch_pack_conv_frag(w1.buf.p, w1.pitch, n->hid, C, 0, sc1, w1.wabs, 1.f, slot(sb + 9), Q1, q1s, B, s,
                                                  sh1, b1w.buf.p, n->bf1.p));
                MARK(ST_NORM1);
                ConvStripArgs k;
                k.Xhi = PAh; k.Xlo = PAl; k.ldn = HW; k.sX = (long)C * HW; k.xslot = slot(sb + 4);
                if (pfold) {
                    k.Wraw = w1.buf.p; k.ldw = w1.pitch; k.wabs = w1.wabs; k.fa = sc1; k.fb = sh1; k.sfa = C;
                    k.bias = b1w.buf.p; k.sbias = 0;
                } else {
                    k.A = Q1; k.sA = q1s; k.aslot = slot(sb + 9);
                    k.bias = n->bf1.p; k.sbias = n->hid;
                }
                k.cw = w1.winf; k.cb = b1w.absmax; k.cinb = slot(sb + 5);
                k.Chi = Uh; k.Clo = Ul; k.sCp = (long)n->hid * HW; k.cslot = slot(sb + 6);
                k.C = C; k.M = n->hid; k.HW = (int)HW; k.nbatch = B; k.act = ACT_GELU;
                HIP_TRY(wl_fc1 ? launch_conv_wl(k, s) : launch_conv_ws(k, s));
            } else {
                HIP_TRY(launch_fold_affine_f16(w1.buf.p, w1.pitch, w1.wabs, sc1, sh1, b1w.buf.p, F1h, F1l, n->bf1.p, B, n->hid, C,
                                               cp, f1s, slot(sb + 9), s));
                MARK(ST_NORM1);
                PkOpts f1;
                f1.w = &w1; f1.fhi = F1h; f1.flo = F1l; f1.fslot = slot(sb + 9); f1.f_stride = f1s;
                f1.bias = n->bf1.p; f1.sbias = n->hid;
                f1.bhi = PAh; f1.blo = PAl; f1.cin = C; f1.in_slot = slot(sb + 4);
                f1.cout = n->hid; f1.act = act;
                f1.ohi = Uh; f1.olo = Ul; f1.cb = b1w.absmax; f1.cslot = slot(sb + 6); f1.cinb = slot(sb + 5);
                ACE_TRY(conv_pk2(n, f1, B, s));
            }
            MARK(ST_MLP_FC1);
            if (ws_fc2) {
                // fc2 + outer skip: fp32 h' (+ planes and statistics for the next block)
                ConvStripArgs k;
                k.Xhi = Uh; k.Xlo = Ul; k.ldn = HW; k.sX = (long)n->hid * HW; k.xslot = slot(sb + 6);
                k.A = reinterpret_cast<const _Float16*>(w2.frag0.p); k.sA = 0; k.ascale = w2.ascale;
                k.bias = b2w.buf.p; k.sbias = 0;
                // residual = a0 h + b0: from the fp32 h, or - when the block input came as planes from a producer epilogue (raw h,
                // bound in hslot(i)) - from those planes; then a mid block writes h' as planes only
                const bool rpl = stream_ok && have_ph_in;
                k.rsc = ra; k.rsh = rb; k.srs = C;
                if (rpl) { k.Rhi = PBh; k.Rlo = PBl; k.sRp = (long)C * HW; k.rslot = hslot(i); }
                else { k.R = res; k.sR = actB; }
                if (last || !rpl) { k.Cf = hn; k.sCf = actB; }
                h_planes_only = !last && rpl;
                k.C = n->hid; k.M = C; k.HW = (int)HW; k.nbatch = B; k.act = ACT_NONE;
                const int nparts = conv_ws_stat_parts(k);   // one partial per workgroup and row (accumulated in LDS)
                if (!last) {
                    k.Chi = PBh; k.Clo = PBl; k.sCp = (long)C * HW; k.cslot = hslot(i + 1);
                    k.cw = w2.winf; k.cb = b2w.absmax; k.rmax = slot(sb + 3);
                    k.part = reinterpret_cast<float4*>(part_h); k.nstrips32 = nparts;
                } else {
                    k.omax = hslot(i + 1);
                }
                HIP_TRY(launch_conv_ws(k, s));
                h_nparts = nparts;
            } else {
                PkOpts f2;
                f2.w = &w2; f2.bias = b2w.buf.p;
                f2.bhi = Uh; f2.blo = Ul; f2.cin = n->hid; f2.in_slot = slot(sb + 6);
                f2.out = hn; f2.cout = C; f2.act = ACT_NONE;
                f2.R = res; f2.r_bs = actB; f2.rsc = ra; f2.rsh = rb;
                if (!last) {   // next block: h planes + statistics, bound instead of the true max in the h slot
                    f2.ohi = PBh; f2.olo = PBl; f2.cb = b2w.absmax; f2.cslot = hslot(i + 1); f2.rmax = slot(sb + 3);
                    f2.part = part_h;
                } else {
                    f2.omax = hslot(i + 1);
                }
                ACE_TRY(conv_pk2(n, f2, B, s));
                h_nparts = gemm4_strips(C, (int)HW);
                h_planes_only = false;
            }
            have_ph = have_hstats = !last;
        } else if (pk) {
            have_ph = have_hstats = false;
            const Weight& ws = *n->weights[n->index.at(p + "inner_skip.weight")];
            if (!n0_planes) ACE_TRY(pack_act(n, res, actB, C, ra, rb, skip_max, Ph, Pl, B, s));
            const bool gelu0 = act == ACT_GELU || act == ACT_GELU_FAST;
            if (cln && gelu0 && ws.frag0.p && cln_skip_on_conv_ws(n, C, C)) {
                // weight-stationary kernel, mode 6: GELU(W . planes + bias + Y) -> fp32 T (the conditional norm that follows reads fp32)
                ConvStripArgs k;
                k.Xhi = Ph; k.Xlo = Pl; k.ldn = HW; k.sX = (long)C * HW; k.xslot = skip_max;
                k.A = reinterpret_cast<const _Float16*>(ws.frag0.p); k.sA = 0; k.ascale = ws.ascale;
                k.bias = W(p + "inner_skip.bias"); k.sbias = 0;
                k.R = n->Y.p; k.sR = actB;
                k.Cf = n->T.p; k.sCf = actB; k.omax = slot(sb + 4);
                k.C = C; k.M = C; k.HW = (int)HW; k.nbatch = B; k.act = act;
                HIP_TRY(launch_conv_ws(k, s));
            } else {
                ACE_TRY(conv_pk(n, ws, W(p + "inner_skip.bias"), Ph, Pl, C, skip_max, n->T.p, C, n->Y.p, actB, nullptr, nullptr,
                                act, B, s, slot(sb + 4)));
            }
        } else {
            if (ra && !skip_f16) ACE_TRY(fold(n, wskip, C, C, ra, rb, n->Wf0.p, n->bf0.p, B, s, &wskip));
            ACE_TRY(conv(n, wskip, res, actB, C, nullptr, 0, -1, n->T.p, C, n->Y.p, actB, nullptr, nullptr, act, B, s,
                         skip_f16 ? ra : nullptr, skip_f16 ? rb : nullptr, skip_max, nullptr, slot(sb + 4)));
        }
        if (!fused) MARK(ST_INNER_SKIP);
        // norm1 -> MLP -> + residual   (sfnonet.py:234-250)
        const float *a1 = nullptr, *b1 = nullptr;
        if (cln) {   // followed by the packed MLP, norm1's output exists as planes only
            const bool to_mlp = c.use_mlp && pk && !n->taps_on;
            ACE_TRY(cond_norm(n->T.p, n->T.p, p + "norm1.", C, slot(sb + 5), to_mlp ? Ph : nullptr, to_mlp ? Pl : nullptr, false, &n1_planes));
            MARK(ST_NORM1);
        }
        if (lnrm) {
            HIP_TRY(launch_spatial_layer_norm(n->T.p, W(p + "norm1.weight"), W(p + "norm1.bias"), 1e-6f, (long)B * C, HW, n->T.p, slot(sb + 5), s));
            MARK(ST_NORM1);
        }
        if (norm && !fused) {
            HIP_TRY(launch_instnorm_stats(n->T.p, W(p + "norm1.weight"), W(p + "norm1.bias"), 1e-6f, B, C, HW, sc1, sh1, s,
                                          slot(sb + 5)));
            a1 = sc1; b1 = sh1;
            MARK(ST_NORM1);
        }
        if (fused) {
            // done above
        } else if (c.use_mlp && pk) {
            // fc1 reads norm1(T) packed; its epilogue writes U = act(.) straight in P format (bound-scaled), which fc2
            // consumes by DMA: the hidden activation never exists in fp32
            const Weight& w1 = *n->weights[n->index.at(p + "mlp.fwd.0.weight")];
            const Weight& b1w = *n->weights[n->index.at(p + "mlp.fwd.0.bias")];
            const Weight& w2 = *n->weights[n->index.at(p + "mlp.fwd.2.weight")];
            const unsigned* tmax = (norm || cln || lnrm) ? slot(sb + 5) : slot(sb + 4);
            _Float16* Uh = reinterpret_cast<_Float16*>(n->U.p);
            _Float16* Ul = Uh + (size_t)n->Bmax * n->hid * HW;
            if (!n1_planes) ACE_TRY(pack_act(n, n->T.p, actB, C, a1, b1, tmax, Ph, Pl, B, s));
            const bool gelu1 = act == ACT_GELU || act == ACT_GELU_FAST;
            if (w1.frag0.p && !a1 && gelu1 && n->sw.conv_wl && conv_wl_eligible(C, n->hid, HW)) {
                // weights in LDS, unsynchronised waves (conv_wl.hip): the noise-conditioned nets' fc1 (no norm affine to fold)
                ConvStripArgs k;
                k.Xhi = Ph; k.Xlo = Pl; k.ldn = HW; k.sX = (long)C * HW; k.xslot = tmax;
                k.A = reinterpret_cast<const _Float16*>(w1.frag0.p); k.sA = 0; k.ascale = w1.ascale;
                k.bias = b1w.buf.p; k.sbias = 0;
                k.cw = w1.winf; k.cb = b1w.absmax;
                k.Chi = Uh; k.Clo = Ul; k.sCp = (long)n->hid * HW; k.cslot = slot(sb + 6);
                k.C = C; k.M = n->hid; k.HW = (int)HW; k.nbatch = B; k.act = ACT_GELU;
                HIP_TRY(launch_conv_wl(k, s));
            } else {
                ACE_TRY(conv_pk(n, w1, b1w.buf.p, Ph, Pl, C, tmax, nullptr, n->hid, nullptr, 0, nullptr, nullptr, act, B, s,
                                nullptr, Uh, Ul, b1w.absmax, slot(sb + 6)));
            }
            MARK(ST_MLP_FC1);
            ACE_TRY(conv_pk(n, w2, W(p + "mlp.fwd.2.bias"), Uh, Ul, n->hid, slot(sb + 6), hn, C, res, actB, ra, rb, ACT_NONE,
                            B, s, hslot(i + 1)));
        } else if (c.use_mlp) {
            ConvW wfc1 = conv_weight(n, p + "mlp.fwd.0.weight", p + "mlp.fwd.0.bias");
            if (a1 && !f16) ACE_TRY(fold(n, wfc1, n->hid, C, a1, b1, n->Wf1.p, n->bf1.p, B, s, &wfc1));
            ACE_TRY(conv(n, wfc1, n->T.p, actB, C, nullptr, 0, -1, n->U.p, n->hid, nullptr, 0, nullptr, nullptr, act, B, s,
                         f16 ? a1 : nullptr, f16 ? b1 : nullptr, (norm || cln || lnrm) ? slot(sb + 5) : slot(sb + 4), nullptr,
                         slot(sb + 6)));
            MARK(ST_MLP_FC1);
            ACE_TRY(conv(n, conv_weight(n, p + "mlp.fwd.2.weight", p + "mlp.fwd.2.bias"), n->U.p, (long)n->hid * HW,
                         n->hid, nullptr, 0, -1, hn, C, res, actB, ra, rb, ACT_NONE, B, s, nullptr, nullptr, slot(sb + 6),
                         nullptr, hslot(i + 1)));
        } else {
            HIP_TRY(launch_rowaffine_add(n->T.p, a1, b1, res, ra, rb, hn, (long)B * C, HW, s));
        }
        MARK(ST_MLP_FC2);
        std::swap(h, hn);
        if (n->taps_on)
            HIP_TRY(hipMemcpyAsync(n->taps[i + 1].p, h, sizeof(float) * B * actB, hipMemcpyDeviceToDevice, s));
    }

    // ---- decoder on cat(x, input) (sfnonet.py:741-747); the concat is two row sources of one GEMM
    cur = h; cur_bs = actB; curC = C;
    for (int j = 0; j < c.encoder_layers; ++j) {
        const std::string p = "decoder." + std::to_string(2 * j);
        const bool cat = (j == 0 && c.big_skip);
        ACE_TRY(conv(n, conv_weight(n, p + ".weight", p + ".bias"), cur, cur_bs, cat ? C + Cin : curC,
                     cat ? skip_in : nullptr, (long)Cin * HW, C, ping[j & 1], C, nullptr, 0, nullptr, nullptr, act, B, s, nullptr,
                     nullptr, j == 0 ? (c.use_mlp ? hslot(c.num_layers) : nullptr) : slot(3 + j), skip_in_slot, slot(4 + j)));
        cur = ping[j & 1]; cur_bs = actB; curC = C;
    }
    {
        // without hidden decoder layers the last convolution is the one that sees cat(x, input) and the last block's range slot
        const bool first = c.encoder_layers == 0, cat = first && c.big_skip;
        ACE_TRY(conv(n, conv_weight(n, "decoder." + std::to_string(2 * c.encoder_layers) + ".weight", ""), cur, cur_bs,
                     cat ? C + Cin : curC, cat ? skip_in : nullptr, (long)Cin * HW, cat ? C : -1, out, c.out_chans, nullptr, 0, nullptr,
                     nullptr, ACT_NONE, B, s, nullptr, nullptr,
                     first ? (c.use_mlp ? hslot(c.num_layers) : nullptr) : slot(3 + c.encoder_layers), cat ? skip_in_slot : nullptr, nullptr));
    }
    MARK(ST_DECODER);
    return ACE_OK;
}

static int check_ready(ace_sfno* n, const float* in, float* out, int batch) {
    if (!n || !in || !out) return fail(ACE_ERR_INVALID, "null argument");
    if (batch <= 0 || batch > n->Bmax)
        return fail(ACE_ERR_INVALID, "batch " + std::to_string(batch) + " outside [1, max_batch=" +
                                         std::to_string(n->Bmax) + "]");
    for (auto& w : n->weights)
        if (!w->set) return fail(ACE_ERR_STATE, "parameter '" + w->name + "' has not been set");
    return ACE_OK;
}

extern "C" int ace_sfno_forward(ace_sfno* n, const float* in, float* out, int batch, void* stream) {
    ACE_TRY(check_ready(n, in, out, batch));
    if (n->cfg.normalization_layer == 2)
        return fail(ACE_ERR_STATE, "this net is noise conditioned: call ace_sfno_forward_conditioned");
    return forward_impl(n, in, out, batch, static_cast<hipStream_t>(stream));
}

extern "C" int ace_sfno_forward_conditioned(ace_sfno* n, const float* in, const float* noise, float* out, int batch,
                                            void* stream) {
    ACE_TRY(check_ready(n, in, out, batch));
    if (n->cfg.normalization_layer != 2)
        return fail(ACE_ERR_STATE, "ace_sfno_forward_conditioned needs a net created with conditional layer norms");
    if (!noise && n->cfg.noise_embed_dim > 0) return fail(ACE_ERR_INVALID, "null noise");
    return forward_impl(n, in, out, batch, static_cast<hipStream_t>(stream), nullptr, noise);
}

#ifdef ACE_DEBUG_WS   // debugging builds only (tools/mkvar.sh dbg -DACE_DEBUG_WS): copy of an internal workspace after a forward
extern "C" long ace_debug_read_workspace(ace_sfno* n, const char* which, void* dst, long max_bytes) {
    const std::string w(which);
    DevBuf* b = w == "U" ? &n->U : w == "P" ? &n->P : w == "P2" ? &n->P2 : w == "part" ? &n->part : w == "Y" ? &n->Y : nullptr;
    if (!b || !b->p) return -1;
    const long bytes = std::min<long>((long)(b->n * sizeof(float)), max_bytes);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpy(dst, b->p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return bytes;
}
#endif

extern "C" int ace_sfno_sht_route(const ace_sfno* n, int* forward, int* inverse) {
    if (!n || !n->plan_lg) return fail(ACE_ERR_INVALID, "null argument");
    return ace_sht_plan_route(n->plan_lg.get(), forward, inverse);
}
extern "C" int ace_sfno_num_stages(void) { return ST_COUNT; }
extern "C" const char* ace_sfno_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : nullptr; }
extern "C" int ace_sfno_forward_timed(ace_sfno* n, const float* in, float* out, int batch, void* stream, float* ms_host,
                                      int* calls_host) {
    ACE_TRY(check_ready(n, in, out, batch));
    if (n->cfg.normalization_layer == 2) return fail(ACE_ERR_STATE, "this net is noise conditioned: call ace_sfno_forward_conditioned");
    if (!ms_host) return fail(ACE_ERR_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    StageTimer tm(s);
    ACE_TRY(forward_impl(n, in, out, batch, s, &tm));
    HIP_TRY(tm.finish(ms_host, calls_host));
    return ACE_OK;
}

extern "C" int ace_sfno_forward_conditioned_timed(ace_sfno* n, const float* in, const float* noise, float* out, int batch,
                                                  void* stream, float* ms_host, int* calls_host) {
    ACE_TRY(check_ready(n, in, out, batch));
    if (n->cfg.normalization_layer != 2)
        return fail(ACE_ERR_STATE, "ace_sfno_forward_conditioned_timed needs a net created with conditional layer norms");
    if (!ms_host || (!noise && n->cfg.noise_embed_dim > 0)) return fail(ACE_ERR_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    StageTimer tm(s);
    ACE_TRY(forward_impl(n, in, out, batch, s, &tm, noise));
    HIP_TRY(tm.finish(ms_host, calls_host));
    return ACE_OK;
}

extern "C" int ace_sfno_set_taps(ace_sfno* n, int enable) {
    if (!n) return fail(ACE_ERR_INVALID, "null argument");
    if (enable && n->cfg.scale_factor != 1)   // the taps are (batch, C, nlat, nlon) copies: the inner blocks of such a net live on another grid
        return fail(ACE_ERR_STATE, "per-block taps are not available for scale_factor != 1");
    if (enable && n->taps.empty()) {
        n->taps = std::vector<DevBuf>(n->cfg.num_layers + 1);
        for (auto& t : n->taps) HIP_TRY(t.alloc((size_t)n->Bmax * n->C * n->HW));
    }
    n->taps_on = enable != 0;
    return ACE_OK;
}
extern "C" int ace_sfno_get_tap(ace_sfno* n, int i, float* dst, int batch, void* stream) {
    if (!n || !dst) return fail(ACE_ERR_INVALID, "null argument");
    if (n->taps.empty() || i < -1 || i >= n->cfg.num_layers) return fail(ACE_ERR_STATE, "taps not enabled / bad index");
    HIP_TRY(hipMemcpyAsync(dst, n->taps[i + 1].p, sizeof(float) * batch * n->C * n->HW, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return ACE_OK;
}

extern "C" int ace_sfno_forward_graph(ace_sfno* n, const float* in, float* out, int batch, void* stream) {
    ACE_TRY(check_ready(n, in, out, batch));
    if (n->cfg.normalization_layer == 2) return fail(ACE_ERR_STATE, "this net is noise conditioned: call ace_sfno_forward_conditioned");
    if (n->taps_on) return fail(ACE_ERR_STATE, "disable taps before using the graph path");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n->graphs_stale) {   // parameters changed since capture: the scalars baked into the kernel arguments are out of date
        HIP_TRY(hipStreamSynchronize(s));   // no exec of this handle may be in flight when it is destroyed
        for (auto& kv : n->graphs) (void)hipGraphExecDestroy(kv.second);
        n->graphs.clear();
        n->graphs_stale = false;
    }
    GraphKey key{in, out, batch};
    auto it = n->graphs.find(key);
    if (it == n->graphs.end()) {
        if (!n->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&n->capture_stream, hipStreamNonBlocking));
        // everything queued on the caller's stream must be visible to the first replay; capture itself runs nothing
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(n->capture_stream, hipStreamCaptureModeThreadLocal));
        int rc = forward_impl(n, in, out, batch, n->capture_stream);
        hipError_t e = hipStreamEndCapture(n->capture_stream, &graph);
        if (rc != ACE_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIP_TRY(e);
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(e);
        it = n->graphs.emplace(key, exec).first;
    }
    HIP_TRY(hipGraphLaunch(it->second, s));
    return ACE_OK;
}
