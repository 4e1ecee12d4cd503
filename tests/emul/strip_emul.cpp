// CPU emulation of the strip Legendre kernel's data movement (ace_amd/csrc/strip.hip) on top of the REAL operand packer
// (ace_amd/csrc/strip_pack.h): lanes, MFMA fragment ownership, B-strip loading, tile loop and output mapping are
// restated with the kernel's index formulas; arithmetic is plain double.  Checks the index algebra (geometry, packing,
// triangular masks, tile offsets) against a direct sum - the part of the kernel that can be verified without a GPU.
// Test infrastructure only (built and run by tests/test_strip_emul_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../ace_amd/csrc/strip_pack.h"

using namespace ace;

static int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// mode 0: tab = wt[m][l][pitch] (R = L rows, K = H); mode 1: tab = pt[m][k][pitch] (R = H rows, K = L)
static double run_case(int mode, int H, int L, int M, int N, unsigned seed) {
    const int R = mode == 0 ? L : H, K = mode == 0 ? H : L;
    const int pitch = ((K + 31) / 32) * 32;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> tab((size_t)M * R * pitch, 0.f);
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < K; ++c) {
                const int l = mode == 0 ? r : c;
                tab[((size_t)m * R + r) * pitch + c] = l >= m ? U(rng) : 0.f;   // the real tables are zero for l < m
            }
    // data operand in the kernel's addressing: element (m, k, n) at m * b_moff + k * b_kstride + n
    const long b_kstride = mode == 0 ? N : (long)M * N, b_moff = mode == 0 ? (long)H * N : N;
    // + 16 rows of zero slack past the last contraction row (LEG_STRIP_SLACK_ROWS in kernels.h)
    const size_t bsize = mode == 0 ? ((size_t)M * K + 16) * N : ((size_t)K + 16) * M * N;
    std::vector<float> B(bsize, 0.f);
    for (size_t q = 0; q < (size_t)M * K * N; ++q) B[q] = U(rng);
    if (mode == 1)   // entries with l < m are never written by the producer: poison them
        for (int l = 0; l < K; ++l)
            for (int m = l + 1; m < M; ++m)
                for (int n = 0; n < N; ++n) B[(size_t)m * b_moff + (size_t)l * b_kstride + n] = 1e30f;
    const long c_rstride = mode == 0 ? (long)M * N : N, c_moff = mode == 0 ? N : (long)H * N;
    std::vector<double> C((size_t)M * R * N, -7.0), ref((size_t)M * R * N, -7.0);

    StripPack sp;
    pack_legendre_strip(tab.data(), M, R, K, pitch, mode, 1.0f, sp);

    const int G = (N + 127) / 128;
    for (int m = 0; m < M; ++m) {
        const StripGeom gm = strip_geom(mode, m, R, K);
        if (gm.nks4 > 12 || gm.nks4 % 4) { std::printf("bad nks4 %d\n", gm.nks4); return 1e9; }
        const uint16_t* Am = sp.frags.data() + (size_t)sp.tile_off[m] * 1024;
        for (int grp = 0; grp < G; ++grp)
            for (int wave = 0; wave < 4; ++wave) {
                const int n0 = grp * 128 + wave * 32;
                // resident strip
                std::vector<double> bfrag((size_t)gm.nks4 * 64 * 8);
                const int klast16 = (K + 15) / 16 - 1;
                for (int jj = 0; jj < gm.nks4; ++jj)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, g = lane >> 5;
                        const int n = n0 + i, nc = n < N ? n : N - 1;
                        const int kstep = gm.j0 + jj < klast16 ? gm.j0 + jj : klast16;   // padded k-steps re-read the last real one
                        for (int e = 0; e < 8; ++e) {
                            const size_t at = (size_t)m * b_moff + (size_t)(16 * kstep + 8 * g + e) * b_kstride + nc;
                            if (at >= B.size()) { std::printf("B read out of bounds\n"); return 1e9; }
                            double x = B[at];                                        // un-masked: zero table rows kill k >= K
                            if (jj == 0 && 16 * gm.j0 + 8 * g + e < gm.klo) x = 0.0;  // ... only l < m needs the explicit zero
                            bfrag[((size_t)jj * 64 + lane) * 8 + e] = x;
                        }
                    }
                for (int t = 0; t < gm.ntiles; ++t) {
                    const uint16_t* tile = Am + (size_t)t * gm.nks4 * 1024;
                    // MFMA 32x32x16: D[row i'][col j'] += sum_{g,e} A(lane i' + 32 g)[e] * B(lane j' + 32 g)[e]
                    double D[32][32] = {};
                    for (int jj = 0; jj < gm.nks4; ++jj)
                        for (int ii = 0; ii < 32; ++ii)
                            for (int jc = 0; jc < 32; ++jc)
                                for (int g = 0; g < 2; ++g)
                                    for (int e = 0; e < 8; ++e) {
                                        const uint16_t* blk = tile + (size_t)jj * 1024;
                                        const double a = (double)f16_bits_to_f32(blk[(ii + 32 * g) * 8 + e]) +
                                                         (double)f16_bits_to_f32(blk[512 + (ii + 32 * g) * 8 + e]);
                                        D[ii][jc] += a * bfrag[((size_t)jj * 64 + jc + 32 * g) * 8 + e];
                                    }
                    // accumulator ownership and the store mapping
                    const bool inner = t + 1 < gm.ntiles;
                    const int rbase = gm.row0 + 32 * t, rlo = mode == 0 ? m : 0;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 16; ++r) {
                            const int i = lane & 31, g = lane >> 5;
                            const int row = rbase + acc_row(r, g), n = n0 + i;
                            const bool nomask = inner && (N % 128 == 0);
                            if (nomask || (row >= rlo && row < R && n < N)) {
                                if (row < 0 || row >= R || n >= N) { std::printf("out of bounds store\n"); return 1e9; }
                                C[(size_t)m * c_moff + (size_t)row * c_rstride + n] = D[acc_row(r, g)][i];
                            }
                        }
                }
            }
    }
    // direct sums; fp16 hi+lo of a float in [-1, 1] is exact to ~2^-22, so compare loosely
    double err = 0.0;
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < R; ++r)
            for (int n = 0; n < N; ++n) {
                const int lrow = mode == 0 ? r : -1;
                double s = 0.0;
                bool written = true;
                if (mode == 0 && lrow < m) written = false;   // forward: rows l < m are not part of the result
                for (int c = 0; c < K; ++c) {
                    const int l = mode == 0 ? r : c;
                    if (l < m) continue;
                    s += (double)tab[((size_t)m * R + r) * pitch + c] * (double)B[(size_t)m * b_moff + (size_t)c * b_kstride + n];
                }
                const double got = C[(size_t)m * c_moff + (size_t)r * c_rstride + n];
                if (!written) {   // either untouched or an exact zero (unmasked inner tiles)
                    if (got != -7.0 && got != 0.0) err = 1e9;
                    continue;
                }
                const double d = got - s;
                err = std::max(err, d < 0 ? -d : d);
            }
    return err;
}

int main() {
    struct Case { int mode, H, L, M, N; } cases[] = {
        {0, 20, 18, 12, 40},  {1, 20, 18, 12, 40},  {0, 45, 45, 46, 32},   {1, 45, 45, 46, 32},  {0, 64, 40, 50, 128},
        {1, 64, 40, 50, 128}, {0, 9, 8, 10, 6},     {1, 9, 8, 10, 6},      {0, 180, 180, 181, 128}, {1, 180, 180, 181, 128},
        {0, 100, 90, 51, 256}, {1, 100, 90, 51, 256}, {0, 16, 16, 17, 48}, {1, 16, 16, 17, 48}, {0, 64, 48, 60, 32}, {1, 64, 48, 60, 32},
    };
    double worst = 0.0;
    for (auto& c : cases) {
        const double e = run_case(c.mode, c.H, c.L, c.M, c.N, 1234u + c.H);
        std::printf("mode %d H %d L %d M %d N %d  err %.3e\n", c.mode, c.H, c.L, c.M, c.N, e);
        worst = std::max(worst, e);
    }
    std::printf("worst %.3e\n", worst);
    return worst < 2e-4 ? 0 : 1;
}
