// Host emulation of the work decomposition of csrc/cln_mfma.hip (conditional layer norm, single pass), both forms: the workgroup's waves,
// their row tiles, the A fragments packed by pack_cln_frags (strip_pack.h), the B fragments each wave builds from the conditioning
// field, v_mfma_f32_32x32x16_f16's lane layout (A lane (i, g): row i, k = 8 g .. + 7; B lane (i, g): column i, same k; accumulator
// register r of lane (i, g): row (r & 3) + 8 (r >> 2) + 4 g, column i) and the apply.  Checks every output of the tile against
// the direct formula in fp64.  Test infrastructure (no GPU): catches index algebra, not rounding.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../ace_amd/csrc/strip_pack.h"

using namespace ace;

static int acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

static double run_case(int C, int J, int HW, unsigned seed, int PV) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x((size_t)C * HW), cond((size_t)J * HW), Ws((size_t)C * J), Wb((size_t)C * J), gamma(C), beta(C);
    for (auto& v : x) v = 3.f * nd(rng) + 1.5f;
    for (auto& v : cond) v = 1.7f * nd(rng);
    for (auto& v : Ws) v = 0.3f * nd(rng);
    for (auto& v : Wb) v = 0.02f * nd(rng);
    for (auto& v : gamma) v = 1.f + 0.3f * nd(rng);
    for (auto& v : beta) v = 0.2f * nd(rng);
    float ms = 0.f, mb = 0.f, mc = 0.f;
    for (float v : Ws) ms = std::max(ms, std::fabs(v));
    for (float v : Wb) mb = std::max(mb, std::fabs(v));
    for (float v : cond) mc = std::max(mc, std::fabs(v));
    const float ss = cln_frag_scale(ms), sb = cln_frag_scale(mb);
    int ec = 0; (void)std::frexp(mc, &ec); ec = 12 - ec;
    const float cscale = std::ldexp(1.f, ec), inv_c = std::ldexp(1.f, -ec);
    std::vector<uint16_t> As(cln_frag_halves(C, J)), Ab(cln_frag_halves(C, J));
    pack_cln_frags(Ws.data(), C, J, ss, As.data());
    pack_cln_frags(Wb.data(), C, J, sb, Ab.data());
    const int nk = (J + 15) / 16, NW = PV == 4 ? C / 32 : 8, RT = C / 32 / NW, TP = 32 * PV;
    std::vector<int> written((size_t)C * HW, 0);
    double worst = 0.0;
    for (int tile = 0; tile < (HW + TP - 1) / TP; ++tile)
      for (int cb = 0; cb < PV; ++cb) {
        const int p0 = tile * TP + cb;   // column block cb of the tile: lane i <-> pixel p0 + PV i (beyond the row: zeros in, nothing out)
        auto PX = [&](int i) { return p0 + PV * i; };
        // statistics of the 32 pixels
        std::vector<double> mu(32), rstd(32);
        for (int i = 0; i < 32; ++i) {
            double s = 0, q = 0;
            for (int c = 0; c < C; ++c) { const double v = PX(i) < HW ? x[(size_t)c * HW + PX(i)] : 0.0; s += v; q += v * v; }
            mu[i] = s / C;
            rstd[i] = 1.0 / std::sqrt(std::max(0.0, q / C - mu[i] * mu[i]) + 1e-5);
        }
        // B fragments: [ks][hi | lo][lane][e]
        std::vector<float> Bh((size_t)nk * 64 * 8), Bl((size_t)nk * 64 * 8);
        for (int ks = 0; ks < nk; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int i = lane & 31, g = lane >> 5, j = 16 * ks + 8 * g + e;
                    const float v = (j < J && PX(i) < HW) ? cond[(size_t)j * HW + PX(i)] * cscale : 0.f;
                    const float h = f16_bits_to_f32(f32_to_f16_bits(v));
                    Bh[((size_t)ks * 64 + lane) * 8 + e] = h;
                    Bl[((size_t)ks * 64 + lane) * 8 + e] = f16_bits_to_f32(f32_to_f16_bits(v - h));
                }
        for (int wave = 0; wave < NW; ++wave)
            for (int t = 0; t < RT; ++t) {
                const int rt = wave * RT + t;
                // D[row][col] = sum over k-steps, lanes' k ranges: A lane (row, g) x B lane (col, g)
                double S[32][32] = {}, Bv[32][32] = {};
                for (int ks = 0; ks < nk; ++ks) {
                    const uint16_t* bs = As.data() + ((size_t)rt * nk + ks) * 1024;
                    const uint16_t* bb = Ab.data() + ((size_t)rt * nk + ks) * 1024;
                    for (int row = 0; row < 32; ++row)
                        for (int col = 0; col < 32; ++col)
                            for (int g = 0; g < 2; ++g)
                                for (int e = 0; e < 8; ++e) {
                                    const int la = g * 32 + row, lb = g * 32 + col;
                                    const double sh = f16_bits_to_f32(bs[la * 8 + e]), sl = f16_bits_to_f32(bs[512 + la * 8 + e]);
                                    const double wh = f16_bits_to_f32(bb[la * 8 + e]), wl = f16_bits_to_f32(bb[512 + la * 8 + e]);
                                    const double bh = Bh[((size_t)ks * 64 + lb) * 8 + e], bl = Bl[((size_t)ks * 64 + lb) * 8 + e];
                                    S[row][col] += sl * bh + sh * bl + sh * bh;
                                    Bv[row][col] += wl * bh + wh * bl + wh * bh;
                                }
                }
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 16; ++r) {
                        const int i = lane & 31, g = lane >> 5, row = acc_row(r, g), c = rt * 32 + row;
                        if (PX(i) >= HW) continue;
                        written[(size_t)c * HW + PX(i)] += 1;
                        const double xv = x[(size_t)c * HW + PX(i)];
                        const double v = (xv - mu[i]) * rstd[i] * gamma[c] + beta[c];
                        const double o = v * (1.0 + S[row][i] * ((double)inv_c / ss)) + Bv[row][i] * ((double)inv_c / sb);
                        double sref = 0, bref = 0;
                        for (int j = 0; j < J; ++j) {
                            sref += (double)Ws[(size_t)c * J + j] * cond[(size_t)j * HW + PX(i)];
                            bref += (double)Wb[(size_t)c * J + j] * cond[(size_t)j * HW + PX(i)];
                        }
                        const double ref = v * (1.0 + sref) + bref;
                        worst = std::max(worst, std::fabs(o - ref) / (1.0 + std::fabs(ref)));
                    }
            }
      }
    for (int w : written) if (w != 1) return 1.0;   // every output exactly once
    return worst;
}

int main() {
    // {C, J, HW, pixels per lane}: 1 = the 32-pixel form (8 waves, C / 256 row tiles each), 4 = the 128-pixel form (C / 32 waves, ragged rows)
    const int cases[][4] = {{256, 33, 64, 1}, {512, 33, 32, 1}, {512, 16, 32, 1}, {256, 5, 96, 1}, {768, 40, 32, 1}, {1024, 128, 32, 1},
                            {256, 33, 128, 4}, {512, 33, 160, 4}, {512, 16, 36, 4}, {256, 5, 300, 4}, {512, 128, 132, 4}};
    double worst = 0.0;
    for (auto& c : cases) {
        const double w = run_case(c[0], c[1], c[2], 7u + c[0] + c[1], c[3]);
        std::printf("C %d J %d HW %d PV %d: worst %.3e\n", c[0], c[1], c[2], c[3], w);
        worst = std::max(worst, w);
    }
    std::printf("worst %.3e\n", worst);
    return worst < 5e-6 ? 0 : 1;   // dropped lo x lo terms: ~2^-22 per product, sqrt(J) of them; an index error is O(1)
}
