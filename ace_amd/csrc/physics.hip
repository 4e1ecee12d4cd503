// Post-step physics of the stepper as gfx950 kernels: what SingleModuleStep.step_with_adjustments does
// between the network and the next step (fme/core/step/single_module.py:669-716) -
//   AtmosphereCorrector   fme/core/corrector/atmosphere.py:349-398 (order), 404-700 (the corrections):
//       force positive -> conserve dry air (fp64 global means, reference mass carried from the first step)
//       -> zero global-mean moisture advection -> moisture budget (+ frozen-precipitation clip) -> total energy budget
//   Ocean (prescribed SST) fme/core/ocean.py:167-222, fme/core/prescriber.py:54-117
//   prescribed prognostics single_module.py:700-716
// on the denormalised output planes, in place.  Round 2 ran this as ~60 captured ATen launches per step (+29 % step time);
// here it is FOUR launches: the corrections form a chain of global (area-weighted) means, each needing the fields as corrected
// by the previous link, so one pass over the columns per link:
//   P1  clamp the force-positive fields; per column: dry-air surface pressure of the output and of the step's input
//       -> partial sums
//   P2  (dry-air means known) new surface pressure per column (fp64, as the reference); with it: total-water-path tendency,
//       evaporation, precipitation, advective tendency -> partial sums
//   P3  (moisture means known) rescale precipitation / evaporation, rebuild or re-centre the advective tendency, clip frozen
//       precipitation; per column: total-energy path of output and input, net energy flux, correction factor -> partial sums
//   P4  (energy means known) uniform temperature increment; prescribed SST over ocean; prescribed prognostics
// A column is one thread; every quantity a column needs (8 - 16 levels of temperature and water, a dozen fluxes) is read once
// per pass: ~5 MB per pass at 1 degree, the passes are latency-, not bandwidth-bound.
// Reductions are DETERMINISTIC: each workgroup writes one fp64 partial per quantity (fixed tree inside the workgroup), the
// consumer pass re-sums the partials of its sample in a fixed order (no atomics, nothing to zero).  Per-column arithmetic is
// fp32 in the reference's operation order (fp64 exactly where the reference casts: the dry-air closure).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ace_sfno.h"

// the reference's torch ops round after every multiply and add: no fused multiply-adds in the per-column arithmetic
#pragma clang fp contract(off)

namespace {

constexpr int MAXL = ACE_PHYS_MAX_LEVELS;
constexpr int NQ = 6;            // partial-sum slots per pass
constexpr int NBLK_MAX = 512;    // workgroups per sample (grid-stride beyond)
constexpr int NT = 256;

// fme/core/constants.py
constexpr float LATENT_HEAT_OF_VAPORIZATION = 2.5e6f;
constexpr float LATENT_HEAT_OF_FREEZING = 334000.0f;
constexpr float GRAVITY = 9.80665f;
constexpr float RDGAS = 287.05f;
constexpr float CV_DRY = (float)(1004.6 - 287.05);          // SPECIFIC_HEAT_OF_DRY_AIR_CONST_VOLUME: python float arithmetic, then fp32
constexpr float RV_OVER_RD_M1 = (float)(461.5 / 287.05 - 1.0);   // (RVGAS / RDGAS - 1.0) likewise

struct PhysParams {            // static per plan
    int H, W, L;
    float ak[MAXL + 1], bk[MAXL + 1];   // interface coefficients (fp32 as the reference's tensors)
    double dak[MAXL], dbk[MAXL];        // their fp32 differences, cast to fp64 (atmosphere.py:456-457)
    float dt;
    double inv_wsum;                    // 1 / sum of the area weights over the grid
    int conserve_dry_air, zero_adv, moisture, clip_frozen, energy, ocean;
    float heating;
    const float* wlat;                  // area weight per latitude row (device)
    double* part;                       // [3 passes][max_batch][NQ][NBLK_MAX] partial sums
    double* ref_mass;                   // [max_batch] dry-air reference (global mean)
    int* have_ref;                      // [1] device flag: reference seeded
    int max_batch;
};

__device__ __forceinline__ float ld(const ace_phys_plane& f, int b, long px) { return f.p[(long)b * f.stride + px]; }
__device__ __forceinline__ void st(const ace_phys_plane& f, int b, long px, float v) { f.p[(long)b * f.stride + px] = v; }

// HybridSigmaPressureCoordinate.vertical_integral (fme/core/coordinates.py:262-280) of f[k] at surface pressure ps
template <class F>
__device__ __forceinline__ float vertical_integral(const PhysParams& P, float ps, F&& f) {
    float s = 0.f;
    float lo = P.ak[0] + P.bk[0] * ps;
    for (int k = 0; k < P.L; ++k) {
        const float hi = P.ak[k + 1] + P.bk[k + 1] * ps;
        s += f(k) * (hi - lo);
        lo = hi;
    }
    return s / GRAVITY;
}

// fixed-order workgroup reduction of NQ doubles; thread 0 of the workgroup stores them
__device__ __forceinline__ void block_store_partials(double (&v)[NQ], int nq, double* dst /* [NQ][NBLK_MAX] of this sample */, int blk) {
    __shared__ double red[NQ][NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < nq; ++q) {
        double x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) red[q][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < nq) {
        const int q = threadIdx.x;
        dst[(long)q * NBLK_MAX + blk] = ((red[q][0] + red[q][1]) + (red[q][2] + red[q][3]));
    }
}
// every workgroup re-sums the partials of its sample in the same fixed order: thread t adds blocks t, t + 256, ...; then the
// same tree as above.  Result broadcast through LDS.
__device__ __forceinline__ void block_load_sums(const double* src /* [NQ][NBLK_MAX] */, int nq, int nblk, double (&out)[NQ]) {
    __shared__ double red2[NQ][NT / 64];
    __shared__ double tot[NQ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < nq; ++q) {
        double x = 0.0;
        for (int i = threadIdx.x; i < nblk; i += NT) x += src[(long)q * NBLK_MAX + i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) red2[q][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < nq) tot[threadIdx.x] = ((red2[threadIdx.x][0] + red2[threadIdx.x][1]) + (red2[threadIdx.x][2] + red2[threadIdx.x][3]));
    __syncthreads();
    for (int q = 0; q < nq; ++q) out[q] = tot[q];
    __syncthreads();
}

__device__ __forceinline__ float total_water_path(const PhysParams& P, const ace_phys_plane* wat, int b, long px, float ps) {
    return vertical_integral(P, ps, [&](int k) { return ld(wat[k], b, px); });
}

// ---- P1 -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void phys_p1(PhysParams P, ace_phys_fields F, int nblk) {
    const int b = blockIdx.y;
    const long HW = (long)P.H * P.W;
    double acc[NQ] = {0, 0, 0, 0, 0, 0};
    for (long px = (long)blockIdx.x * NT + threadIdx.x; px < HW; px += (long)nblk * NT) {
        for (int i = 0; i < F.npositive; ++i) {   // force_positive (fme/core/corrector/utils.py:26-44)
            const float v = ld(F.positive[i], b, px);
            st(F.positive[i], b, px, fmaxf(v, 0.0f));
        }
        if (P.conserve_dry_air) {
            const float w = P.wlat[px / P.W];
            const float ps = ld(F.ps, b, px);
            const float dry = ps - GRAVITY * total_water_path(P, F.wat, b, px, ps);          // metrics.py:283-296
            const float psi = ld(F.ps_in, b, px);
            const float dryi = psi - GRAVITY * total_water_path(P, F.wat_in, b, px, psi);
            if (w != 0.0f) {
                acc[0] += (double)dry * (double)w;      // atmosphere.py:447: the mean is taken in fp64
                acc[1] += (double)dryi * (double)w;
            }
        }
    }
    if (P.conserve_dry_air) block_store_partials(acc, 2, P.part + ((long)(0 * P.max_batch + b) * NQ) * NBLK_MAX, blockIdx.x);
}

// ---- P2 -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void phys_p2(PhysParams P, ace_phys_fields F, int nblk) {
    const int b = blockIdx.y;
    const long HW = (long)P.H * P.W;
    double error = 0.0;
    if (P.conserve_dry_air) {
        double s[NQ];
        block_load_sums(P.part + ((long)(0 * P.max_batch + b) * NQ) * NBLK_MAX, 2, nblk, s);
        const double gen_mean = s[0] * P.inv_wsum, in_mean = s[1] * P.inv_wsum;
        // the reference mass is seeded from the first step's INPUT (atmosphere.py:404-428) and kept (CorrectorState)
        const double target = *P.have_ref ? P.ref_mass[b] : in_mean;
        error = gen_mean - target;
        if (blockIdx.x == 0 && threadIdx.x == 0 && !*P.have_ref) P.ref_mass[b] = in_mean;   // (flag set by P4, after every reader)
    }
    const bool need = P.moisture != 0 || P.zero_adv;
    double acc[NQ] = {0, 0, 0, 0, 0, 0};
    for (long px = (long)blockIdx.x * NT + threadIdx.x; px < HW; px += (long)nblk * NT) {
        float ps = (P.conserve_dry_air || P.moisture) ? ld(F.ps, b, px) : 0.f;   // (absent when only the advection mean is removed)
        if (P.conserve_dry_air) {   // atmosphere.py:431-467
            const double dry = (double)(ps - GRAVITY * total_water_path(P, F.wat, b, px, ps)) - error;
            double sa = 0.0, sb = 0.0;
            for (int k = 0; k < P.L; ++k) {
                const double wk = (double)ld(F.wat[k], b, px);
                sa += P.dak[k] * wk;
                sb += P.dbk[k] * wk;
            }
            ps = (float)((dry + sa) / (1.0 - sb));
            st(F.ps, b, px, ps);
        }
        if (need) {
            const float w = P.wlat[px / P.W];
            if (w != 0.0f) {
                if (P.moisture) {   // atmosphere.py:511-560: the three global means
                    const float psi = ld(F.ps_in, b, px);
                    const float tend = (total_water_path(P, F.wat, b, px, ps) - total_water_path(P, F.wat_in, b, px, psi)) / P.dt;
                    acc[0] += (double)(tend * w);
                    acc[1] += (double)((ld(F.lhf, b, px) / LATENT_HEAT_OF_VAPORIZATION) * w);
                    acc[2] += (double)(ld(F.precip, b, px) * w);
                }
                if (P.zero_adv) acc[3] += (double)(ld(F.adv, b, px) * w);
            }
        }
    }
    if (need) block_store_partials(acc, 4, P.part + ((long)(1 * P.max_batch + b) * NQ) * NBLK_MAX, blockIdx.x);
}

// ---- energy helpers (atmosphere_data.py:340-416, atmosphere.py:666-692) -------------------------------------------------
struct EnergyColumn { float path, factor; };
__device__ __forceinline__ EnergyColumn energy_column(const PhysParams& P, const ace_phys_plane* Tf, const ace_phys_plane* Wf,
                                                      int b, long px, float ps, float hsfc_raw, bool want_factor, float tshift) {
    float lt[MAXL], T[MAXL], q[MAXL];
    float lo = logf(fmaxf(P.ak[0] + P.bk[0] * ps, 1.0f));
    for (int k = 0; k < P.L; ++k) {
        T[k] = ld(Tf[k], b, px) + tshift;
        q[k] = ld(Wf[k], b, px);
        const float hi = logf(fmaxf(P.ak[k + 1] + P.bk[k + 1] * ps, 1.0f));
        const float tv = T[k] * (1.0f + RV_OVER_RD_M1 * q[k]);
        lt[k] = (hi - lo) * RDGAS * tv / GRAVITY;      // compute_layer_thickness
        lo = hi;
    }
    const float hsfc = hsfc_raw < 0.0f ? 0.0f : hsfc_raw;
    // height at the interfaces: cumulative thickness from the bottom (flip - cumsum - flip), + surface height
    float hi_next = hsfc;       // interface L
    float cum = 0.f;
    float e[MAXL], qd[MAXL], cum2[MAXL];
    float c2 = 0.f;
    for (int k = P.L - 1; k >= 0; --k) {
        cum += lt[k];
        const float hi_k = cum + hsfc;
        const float hmid = 0.5f * (hi_k + hi_next);
        e[k] = T[k] * CV_DRY + q[k] * LATENT_HEAT_OF_VAPORIZATION + hmid * GRAVITY;     // total_energy_ace2
        hi_next = hi_k;
        qd[k] = lt[k] * GRAVITY / T[k];
        c2 += qd[k];
        cum2[k] = c2;
    }
    EnergyColumn r;
    r.path = vertical_integral(P, ps, [&](int k) { return e[k]; });
    r.factor = want_factor ? vertical_integral(P, ps, [&](int k) { return CV_DRY - 0.5f * qd[k] + cum2[k]; }) : 0.f;
    return r;
}
__device__ __forceinline__ float frozen_rate(const ace_phys_fields& F, int b, long px) {
    if (F.frozen.p) return ld(F.frozen, b, px);
    if (F.frozen_parts[0].p) return ld(F.frozen_parts[0], b, px) + ld(F.frozen_parts[1], b, px) + ld(F.frozen_parts[2], b, px);
    return 0.f;
}

// ---- P3 -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void phys_p3(PhysParams P, ace_phys_fields F, int nblk) {
    const int b = blockIdx.y;
    const long HW = (long)P.H * P.W;
    float tend_gm = 0.f, evap_gm = 0.f, precip_gm = 0.f, adv_gm = 0.f;
    if (P.moisture || P.zero_adv) {
        double s[NQ];
        block_load_sums(P.part + ((long)(1 * P.max_batch + b) * NQ) * NBLK_MAX, 4, nblk, s);
        tend_gm = (float)(s[0] * P.inv_wsum); evap_gm = (float)(s[1] * P.inv_wsum);
        precip_gm = (float)(s[2] * P.inv_wsum); adv_gm = (float)(s[3] * P.inv_wsum);
    }
    const int mb = P.moisture;   // 1 precipitation, 2 evaporation, 3 advection_and_precipitation, 4 advection_and_evaporation
    const float pscale = (mb == 1 || mb == 3) ? (evap_gm - tend_gm) / precip_gm : 1.0f;
    const float escale = (mb == 2 || mb == 4) ? (tend_gm + precip_gm) / evap_gm : 1.0f;
    double acc[NQ] = {0, 0, 0, 0, 0, 0};
    for (long px = (long)blockIdx.x * NT + threadIdx.x; px < HW; px += (long)nblk * NT) {
        const float ps = (mb >= 3 || P.energy) ? ld(F.ps, b, px) : 0.f;
        float precip = 0.f, lhf = 0.f;
        if (P.zero_adv) st(F.adv, b, px, ld(F.adv, b, px) - adv_gm);          // atmosphere.py:470-487
        if (mb) {
            precip = ld(F.precip, b, px);
            lhf = ld(F.lhf, b, px);
            if (mb == 1 || mb == 3) { precip = precip * pscale; st(F.precip, b, px, precip); }
            if (mb == 2 || mb == 4) { lhf = ((lhf / LATENT_HEAT_OF_VAPORIZATION) * escale) * LATENT_HEAT_OF_VAPORIZATION; st(F.lhf, b, px, lhf); }
            if (mb >= 3) {
                const float psi = ld(F.ps_in, b, px);
                const float tend = (total_water_path(P, F.wat, b, px, ps) - total_water_path(P, F.wat_in, b, px, psi)) / P.dt;
                st(F.adv, b, px, tend - (lhf / LATENT_HEAT_OF_VAPORIZATION - precip));
            }
            if (P.clip_frozen && F.frozen.p) st(F.frozen, b, px, fminf(ld(F.frozen, b, px), precip));   // atmosphere.py:490-508
        }
        if (P.energy) {   // atmosphere.py:611-663
            const float w = P.wlat[px / P.W];
            if (w != 0.0f) {
                const EnergyColumn g = energy_column(P, F.T, F.wat, b, px, ps, ld(F.hgt_next, b, px) * F.hgt_next_scale, true, 0.f);
                const EnergyColumn in = energy_column(P, F.T_in, F.wat_in, b, px, ld(F.ps_in, b, px), ld(F.hgt_in, b, px) * F.hgt_in_scale, false, 0.f);
                const float net_sfc = (ld(F.dswsfc, b, px) - ld(F.uswsfc, b, px) + ld(F.dlwsfc, b, px) - ld(F.ulwsfc, b, px))
                                      + (-ld(F.lhf, b, px) - ld(F.shf, b, px)) - frozen_rate(F, b, px) * LATENT_HEAT_OF_FREEZING;
                const float net_toa = ld(F.dswtoa_next, b, px) - ld(F.uswtoa, b, px) - ld(F.ulwtoa, b, px);
                acc[0] += (double)(g.path * w);
                acc[1] += (double)(in.path * w);
                acc[2] += (double)((net_toa - net_sfc) * w);
                acc[3] += (double)(g.factor * w);
            }
        }
    }
    if (P.energy) block_store_partials(acc, 4, P.part + ((long)(2 * P.max_batch + b) * NQ) * NBLK_MAX, blockIdx.x);
}

// ---- P4 -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void phys_p4(PhysParams P, ace_phys_fields F, int nblk) {
    const int b = blockIdx.y;
    const long HW = (long)P.H * P.W;
    float tcorr = 0.f;
    if (P.energy) {
        double s[NQ];
        block_load_sums(P.part + ((long)(2 * P.max_batch + b) * NQ) * NBLK_MAX, 4, nblk, s);
        const float gen_gm = (float)(s[0] * P.inv_wsum), in_gm = (float)(s[1] * P.inv_wsum);
        const float flux_gm = (float)(s[2] * P.inv_wsum), factor_gm = (float)(s[3] * P.inv_wsum);
        const float desired = in_gm + (flux_gm + P.heating) * P.dt;
        tcorr = (desired - gen_gm) / factor_gm;
    }
    for (long px = (long)blockIdx.x * NT + threadIdx.x; px < HW; px += (long)nblk * NT) {
        if (P.energy)
            for (int k = 0; k < P.L; ++k) st(F.T[k], b, px, ld(F.T[k], b, px) + tcorr);
        if (P.ocean) {   // prescriber.py:84-117 on the next step's mask / target (ocean.py:196-215)
            const float mask = ld(F.ocean_fraction, b, px), target = ld(F.sst_target, b, px), gen = ld(F.sst, b, px);
            float out;
            if (P.ocean == 2) out = mask * target + (1.0f - mask) * gen;
            else out = ((int)rintf(mask) == 1) ? target : gen;      // torch.round: half to even
            st(F.sst, b, px, out);
        }
        for (int i = 0; i < F.nprescribed; ++i) st(F.prescribed_dst[i], b, px, ld(F.prescribed_src[i], b, px));
    }
    if (P.conserve_dry_air && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *P.have_ref = 1;
}

}  // namespace

struct ace_physics {
    PhysParams P;
    void* dev = nullptr;       // one allocation: wlat | part | ref_mass | have_ref
    std::string err;
};

static thread_local std::string g_perr;
static int pfail(int code, const std::string& m) { g_perr = m; return code; }
extern "C" const char* ace_physics_last_error(void) { return g_perr.c_str(); }

extern "C" int ace_physics_create(const ace_phys_config* c, const float* area_weights_lat_host, const float* ak_host,
                                  const float* bk_host, ace_physics** out) {
    if (!c || !out) return pfail(ACE_ERR_INVALID, "null argument");
    if (c->nlat < 1 || c->nlon < 1 || c->max_batch < 1) return pfail(ACE_ERR_INVALID, "bad grid / batch");
    const bool geom = c->conserve_dry_air || c->zero_global_mean_moisture_advection || c->moisture_budget || c->energy_budget;
    const bool vert = c->conserve_dry_air || c->moisture_budget || c->energy_budget;
    if (geom && !area_weights_lat_host) return pfail(ACE_ERR_INVALID, "area weights are required for the conservation corrections");
    if (vert && (!ak_host || !bk_host || c->nlev < 1 || c->nlev > MAXL))
        return pfail(ACE_ERR_INVALID, "the conservation corrections need ak / bk with 1 <= nlev <= " + std::to_string(MAXL));
    if (c->moisture_budget < 0 || c->moisture_budget > 4 || c->energy_budget < 0 || c->energy_budget > 1 || c->ocean < 0 || c->ocean > 2)
        return pfail(ACE_ERR_INVALID, "unknown correction variant");
    if ((c->moisture_budget || c->energy_budget) && !(c->timestep_seconds > 0)) return pfail(ACE_ERR_INVALID, "timestep required");
    auto h = std::make_unique<ace_physics>();
    PhysParams& P = h->P;
    std::memset(&P, 0, sizeof(P));
    P.H = c->nlat; P.W = c->nlon; P.L = vert ? c->nlev : 0;
    for (int k = 0; k <= P.L && vert; ++k) { P.ak[k] = ak_host[k]; P.bk[k] = bk_host[k]; }
    for (int k = 0; k < P.L; ++k) { P.dak[k] = (double)(P.ak[k + 1] - P.ak[k]); P.dbk[k] = (double)(P.bk[k + 1] - P.bk[k]); }
    P.dt = (float)c->timestep_seconds;
    P.conserve_dry_air = c->conserve_dry_air; P.zero_adv = c->zero_global_mean_moisture_advection;
    P.moisture = c->moisture_budget; P.clip_frozen = c->clip_frozen_precipitation; P.energy = c->energy_budget;
    P.ocean = c->ocean; P.heating = (float)c->unaccounted_heating; P.max_batch = c->max_batch;
    double wsum = 0.0;
    std::vector<float> wl((size_t)c->nlat, 0.f);
    if (area_weights_lat_host)
        for (int i = 0; i < c->nlat; ++i) { wl[i] = area_weights_lat_host[i]; wsum += (double)wl[i] * c->nlon; }
    P.inv_wsum = wsum > 0 ? 1.0 / wsum : 0.0;
    const size_t b_w = ((size_t)c->nlat * 4 + 255) & ~(size_t)255;
    const size_t b_part = (size_t)3 * c->max_batch * NQ * NBLK_MAX * 8;
    const size_t b_ref = ((size_t)c->max_batch * 8 + 255) & ~(size_t)255;
    if (hipMalloc(&h->dev, b_w + b_part + b_ref + 256) != hipSuccess) return pfail(ACE_ERR_RUNTIME, "hipMalloc failed");
    char* d = static_cast<char*>(h->dev);
    if (hipMemset(d, 0, b_w + b_part + b_ref + 256) != hipSuccess || hipMemcpy(d, wl.data(), wl.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(h->dev);
        return pfail(ACE_ERR_RUNTIME, "device initialisation failed");
    }
    P.wlat = reinterpret_cast<const float*>(d);
    P.part = reinterpret_cast<double*>(d + b_w);
    P.ref_mass = reinterpret_cast<double*>(d + b_w + b_part);
    P.have_ref = reinterpret_cast<int*>(d + b_w + b_part + b_ref);
    *out = h.release();
    return ACE_OK;
}
extern "C" void ace_physics_destroy(ace_physics* h) {
    if (!h) return;
    if (h->dev) (void)hipFree(h->dev);
    delete h;
}
// new initial condition: the next step re-seeds the dry-air reference (stream-ordered)
extern "C" int ace_physics_reset(ace_physics* h, void* stream) {
    if (!h) return pfail(ACE_ERR_INVALID, "null argument");
    if (hipMemsetAsync(h->P.have_ref, 0, sizeof(int), static_cast<hipStream_t>(stream)) != hipSuccess) return pfail(ACE_ERR_RUNTIME, "memset failed");
    return ACE_OK;
}
// carried CorrectorState (fme/core/corrector/state.py): ref_dev = (batch) fp64 on the device
extern "C" int ace_physics_set_reference(ace_physics* h, const double* ref_dev, int batch, void* stream) {
    if (!h || !ref_dev || batch < 1 || batch > h->P.max_batch) return pfail(ACE_ERR_INVALID, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int one = 1;
    if (hipMemcpyAsync(h->P.ref_mass, ref_dev, sizeof(double) * batch, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(h->P.have_ref, &one, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return pfail(ACE_ERR_RUNTIME, "copy failed");
    return ACE_OK;
}
extern "C" int ace_physics_get_reference(ace_physics* h, double* ref_dev, int* have_host, int batch, void* stream) {
    if (!h || !ref_dev || batch < 1 || batch > h->P.max_batch) return pfail(ACE_ERR_INVALID, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int have = 0;
    if (hipMemcpyAsync(ref_dev, h->P.ref_mass, sizeof(double) * batch, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(&have, h->P.have_ref, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return pfail(ACE_ERR_RUNTIME, "copy failed");
    if (have_host) *have_host = have;
    return ACE_OK;
}

extern "C" int ace_physics_apply(ace_physics* h, const ace_phys_fields* f, int batch, void* stream) {
    if (!h || !f) return pfail(ACE_ERR_INVALID, "null argument");
    const PhysParams& P = h->P;
    if (batch < 1 || batch > P.max_batch) return pfail(ACE_ERR_INVALID, "batch outside [1, max_batch]");
    if (f->npositive < 0 || f->npositive > ACE_PHYS_MAX_POSITIVE || f->nprescribed < 0 || f->nprescribed > ACE_PHYS_MAX_PRESCRIBED)
        return pfail(ACE_ERR_INVALID, "too many force-positive / prescribed fields");
    auto need = [&](const ace_phys_plane& p) { return p.p != nullptr; };
    auto levels = [&](const ace_phys_plane* a) { for (int k = 0; k < P.L; ++k) if (!a[k].p) return false; return true; };
    if (P.conserve_dry_air && !(need(f->ps) && need(f->ps_in) && levels(f->wat) && levels(f->wat_in)))
        return pfail(ACE_ERR_INVALID, "conserve_dry_air needs surface pressure and specific_total_water of output and input");
    if (P.zero_adv && !need(f->adv)) return pfail(ACE_ERR_INVALID, "tendency_of_total_water_path_due_to_advection is missing");
    if (P.moisture && !(need(f->ps) && need(f->ps_in) && levels(f->wat) && levels(f->wat_in) && need(f->lhf) && need(f->precip) &&
                        (P.moisture < 3 || need(f->adv))))
        return pfail(ACE_ERR_INVALID, "moisture budget correction: a required field is missing");
    if (P.energy && !(need(f->ps) && need(f->ps_in) && levels(f->wat) && levels(f->wat_in) && levels(f->T) && levels(f->T_in) &&
                      need(f->hgt_in) && need(f->hgt_next) && need(f->dswtoa_next) && need(f->lhf) && need(f->shf) && need(f->dswsfc) &&
                      need(f->uswsfc) && need(f->dlwsfc) && need(f->ulwsfc) && need(f->ulwtoa) && need(f->uswtoa)))
        return pfail(ACE_ERR_INVALID, "total energy budget correction: a required field is missing");
    if (P.ocean && !(need(f->sst) && need(f->sst_target) && need(f->ocean_fraction))) return pfail(ACE_ERR_INVALID, "ocean: a required field is missing");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long HW = (long)P.H * P.W;
    int nblk = (int)((HW + NT - 1) / NT);
    if (nblk > NBLK_MAX) nblk = NBLK_MAX;
    dim3 grid((unsigned)nblk, (unsigned)batch), block(NT);
    const bool p1 = f->npositive > 0 || P.conserve_dry_air;
    const bool p2 = P.conserve_dry_air || P.moisture || P.zero_adv;
    const bool p3 = P.moisture || P.zero_adv || P.energy;
    const bool p4 = P.energy || P.ocean || f->nprescribed > 0 || P.conserve_dry_air;
    if (p1) hipLaunchKernelGGL(phys_p1, grid, block, 0, s, P, *f, nblk);
    if (p2) hipLaunchKernelGGL(phys_p2, grid, block, 0, s, P, *f, nblk);
    if (p3) hipLaunchKernelGGL(phys_p3, grid, block, 0, s, P, *f, nblk);
    if (p4) hipLaunchKernelGGL(phys_p4, grid, block, 0, s, P, *f, nblk);
    if (hipGetLastError() != hipSuccess) return pfail(ACE_ERR_RUNTIME, "physics kernel launch failed");
    return ACE_OK;
}
