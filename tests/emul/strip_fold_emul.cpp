// CPU emulation of the folded strip Legendre kernel's data movement (ace_amd/csrc/strip_fold.hip) on top of the REAL operand
// packer and geometry (ace_amd/csrc/strip_pack.h: fold_geom, pack_legendre_fold): lanes, MFMA fragment ownership, the two
// range-checked descriptors of the forward strip load (32-bit unsigned offsets, out of range = zero), the coefficient-row load of
// the inverse, the unit / pair loop, the row maps and the masks of the deferred and the final stores are restated with the
// kernel's index formulas; arithmetic is plain double.  Compared with the UNFOLDED direct sums over all latitudes on tables that
// are exactly symmetric (odd nlat included: the middle row is its own mirror image).  Test infrastructure only
// (built and run by tests/test_strip_emul_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../ace_amd/csrc/strip_pack.h"

using namespace ace;

static int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct RowMap { int rbase, rstep, vlo, vhi; };

// a raw buffer load: base element index, num_records in bytes, 32-bit unsigned byte offset
static double buf_load(const std::vector<float>& mem, size_t base, unsigned num_records, unsigned off, bool* fault) {
    if ((unsigned long long)off + 4ull > (unsigned long long)num_records) return 0.0;
    const size_t at = base + off / 4;
    if (at >= mem.size()) { *fault = true; return 0.0; }
    return mem[at];
}

// mode 0: tab = wt[m][l][pitch] (lmax rows, nlat columns); mode 1: tab = pt[m][k][pitch] (nlat rows, lmax columns)
static double run_case(int mode, int H, int L, int M, int N, unsigned seed) {
    const int R = mode == 0 ? L : H, K = mode == 0 ? H : L;
    const int pitch = ((K + 31) / 32) * 32;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> tab((size_t)M * R * pitch, 0.f);
    const int Hh = (H + 1) / 2;
    for (int m = 0; m < M; ++m)
        for (int l = m; l < L; ++l)
            for (int k = 0; k < Hh; ++k) {
                const float s = ((l - m) & 1) ? -1.f : 1.f;
                const int km = H - 1 - k;
                // the middle row of an odd nlat: an odd function vanishes there (rounding-level entries in the real tables)
                const float v = (km == k && s < 0.f) ? U(rng) * 1e-9f : U(rng);
                auto at = [&](int kk) { return mode == 0 ? ((size_t)m * R + l) * pitch + kk : ((size_t)m * R + kk) * pitch + l; };
                tab[at(k)] = v;
                if (km != k) tab[at(km)] = s * v;
            }
    const long b_kstride = mode == 0 ? N : (long)M * N, b_moff = mode == 0 ? (long)H * N : N;
    const size_t bsize = mode == 0 ? (size_t)M * K * N : (size_t)K * M * N;
    std::vector<float> B(bsize, 0.f);
    for (auto& x : B) x = U(rng);
    if (mode == 1)   // entries with l < m are never written by the producer: poison them
        for (int l = 0; l < K; ++l)
            for (int m = l + 1; m < M; ++m)
                for (int n = 0; n < N; ++n) B[(size_t)m * b_moff + (size_t)l * b_kstride + n] = 1e30f;
    const long c_rstride = mode == 0 ? (long)M * N : N, c_moff = mode == 0 ? N : (long)H * N;
    std::vector<double> C((size_t)M * R * N, -7.0);
    std::vector<int> writes((size_t)M * R * N, 0);

    if (fold_symmetry_error(tab.data(), M, H, L, pitch, mode) > 1e-7) { std::printf("table not symmetric\n"); return 1e9; }
    StripPack sp;
    pack_legendre_fold(tab.data(), M, H, L, pitch, mode, 1.0f, sp);

    bool fault = false;
    const int G = (N + 127) / 128;
    for (int m = 0; m < M; ++m) {
        const FoldGeom gm = fold_geom(mode, m, R, K);
        // instantiated unit sizes: 2, 4, 6 (small form) and 12, 18, 24 (big form, one workgroup per CU)
        if (!(gm.nkp2 == 2 || gm.nkp2 == 4 || gm.nkp2 == 6 || gm.nkp2 == 12 || gm.nkp2 == 18 || gm.nkp2 == 24) || gm.nkp2 < gm.nkp) {
            std::printf("bad nkp2 %d (nkp %d)\n", gm.nkp2, gm.nkp);
            return 1e9;
        }
        const int NKP = gm.nkp2;
        const uint16_t* Am = sp.frags.data() + (size_t)sp.tile_off[m] * 1024;
        const int nunits = 2 * gm.npairs;
        for (int grp = 0; grp < G; ++grp)
            for (int wave = 0; wave < 4; ++wave) {
                const int n0 = grp * 128 + wave * 32;
                // resident operands bf[par][jj][lane][e]
                std::vector<double> bf((size_t)2 * NKP * 64 * 8);
                auto BF = [&](int par, int jj, int lane, int e) -> double& { return bf[(((size_t)par * NKP + jj) * 64 + lane) * 8 + e]; };
                const long ks = b_kstride;
                const unsigned rowb = (unsigned)(ks * 4);
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, g = lane >> 5;
                    const int n = n0 + i, nc = n < N ? n : N - 1;
                    if (mode == 0) {
                        const size_t baseD = (size_t)m * b_moff, baseM = baseD + (size_t)gm.Hh * ks;
                        const unsigned nrD = (unsigned)((long)gm.Hh * rowb), nrM = (unsigned)((long)(H - gm.Hh) * rowb);
                        for (int e = 0; e < 8; ++e) {
                            const unsigned vd = (unsigned)(((long)(8 * g + e) * ks + nc) * 4);
                            const unsigned vm = (unsigned)(((long)(H - 1 - gm.Hh - (8 * g + e)) * ks + nc) * 4);
                            for (int jj = 0; jj < NKP; ++jj) {
                                const unsigned so = (unsigned)((long)(16 * jj) * rowb);
                                const double a = buf_load(B, baseD, nrD, vd + so, &fault);
                                const double b = buf_load(B, baseM, nrM, vm - so, &fault);
                                BF(0, jj, lane, e) = a + b;
                                BF(1, jj, lane, e) = a - b;
                            }
                        }
                    } else {
                        const size_t baseE = (size_t)m * b_moff;
                        const long span = ((long)(K - 1) * ks + N) * 4;
                        const unsigned nrE = (unsigned)(span > 0 ? span : 0);
                        for (int e = 0; e < 8; ++e) {
                            const unsigned ve = (unsigned)(((long)(m + 2 * (8 * g + e)) * ks + nc) * 4);
                            for (int jj = 0; jj < NKP; ++jj) {
                                const unsigned so = (unsigned)((long)(32 * jj) * rowb);
                                BF(0, jj, lane, e) = buf_load(B, baseE, nrE, ve + so, &fault);
                                BF(1, jj, lane, e) = buf_load(B, baseE, nrE, ve + so + rowb, &fault);
                            }
                        }
                    }
                }
                auto mfma_unit = [&](int u, double D[32][32]) {
                    const int par = u & 1;
                    const uint16_t* unit = Am + (size_t)u * NKP * 1024;
                    for (int a = 0; a < 32; ++a) for (int b = 0; b < 32; ++b) D[a][b] = 0.0;
                    for (int jj = 0; jj < NKP; ++jj)
                        for (int ii = 0; ii < 32; ++ii)
                            for (int jc = 0; jc < 32; ++jc)
                                for (int g = 0; g < 2; ++g)
                                    for (int e = 0; e < 8; ++e) {
                                        const uint16_t* blk = unit + (size_t)jj * 1024;
                                        const double a = (double)f16_bits_to_f32(blk[(ii + 32 * g) * 8 + e]) +
                                                         (double)f16_bits_to_f32(blk[512 + (ii + 32 * g) * 8 + e]);
                                        D[ii][jc] += a * BF(par, jj, jc + 32 * g, e);
                                    }
                };
                auto store = [&](const RowMap rm, double D[32][32], bool inner) {
                    const bool nomask = inner && (N % 128 == 0);
                    if (N % 4 != 0) {   // OUT == 2: scalar stores straight from the accumulator registers
                        for (int lane = 0; lane < 64; ++lane)
                            for (int r = 0; r < 16; ++r) {
                                const int i = lane & 31, g = lane >> 5, n = n0 + i;
                                const int row = rm.rbase + rm.rstep * acc_row(r, g);
                                if (!(row >= rm.vlo && row < rm.vhi && n < N)) continue;
                                if (row < 0 || row >= R) { fault = true; continue; }
                                const size_t at = (size_t)m * c_moff + (size_t)row * c_rstride + n;
                                C[at] = D[acc_row(r, g)][i];
                                writes[at]++;
                            }
                        return;
                    }
                    // the transposing store: lane reads 4 consecutive columns of row rl of each half
                    for (int hf = 0; hf < 2; ++hf)
                        for (int ps = 0; ps < 2; ++ps)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int rl = ps * 8 + (lane >> 3), c4 = (lane & 7) * 4, ncol = n0 + c4;
                                const int row = rm.rbase + rm.rstep * (16 * hf + rl);
                                const bool ok = nomask || (row >= rm.vlo && row < rm.vhi && ncol < N);
                                if (!ok) continue;
                                for (int q = 0; q < 4; ++q) {
                                    if (row < 0 || row >= R || ncol + q >= N) { fault = true; continue; }
                                    const size_t at = (size_t)m * c_moff + (size_t)row * c_rstride + ncol + q;
                                    C[at] = D[16 * hf + rl][c4 + q];
                                    writes[at]++;
                                }
                            }
                };
                // accumulator ownership check: register r of lane (i, g) is row acc_row(r, g), column i - the transpose buffer
                // is written at (acc_row - 16 hf) * 32 + i and read back at rl * 32 + c4: both index D[row][col] here
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 16; ++r)
                        if (acc_row(r, lane >> 5) < 0 || acc_row(r, lane >> 5) > 31) fault = true;
                auto fwd_map = [&](int tp, int par) { return RowMap{m + par + 64 * tp, 2, 0, R}; };
                auto north_map = [&](int tp) { return RowMap{32 * tp, 1, 0, gm.Hh}; };
                auto south_map = [&](int tp) { return RowMap{H - 1 - 32 * tp, -1, gm.Hh, H}; };
                static double Dev[32][32], Dod[32][32], Pa[32][32], Pb[32][32];
                for (int tp = 0; tp < gm.npairs; ++tp) {
                    if (2 * tp + 1 >= nunits) { std::printf("unit past the end\n"); return 1e9; }
                    const bool last = tp + 1 == gm.npairs;
                    if (mode == 0) {
                        if (tp > 0) store(fwd_map(tp - 1, 1), Pb, true);
                        mfma_unit(2 * tp, Dev);
                        if (!last) store(fwd_map(tp, 0), Dev, true);
                        mfma_unit(2 * tp + 1, Dod);
                        for (int a = 0; a < 32; ++a) for (int b = 0; b < 32; ++b) { Pb[a][b] = Dod[a][b]; if (last) Pa[a][b] = Dev[a][b]; }
                    } else {
                        if (tp > 0) { store(north_map(tp - 1), Pa, true); store(south_map(tp - 1), Pb, true); }
                        mfma_unit(2 * tp, Dev);
                        mfma_unit(2 * tp + 1, Dod);
                        for (int a = 0; a < 32; ++a) for (int b = 0; b < 32; ++b) { Pa[a][b] = Dev[a][b] + Dod[a][b]; Pb[a][b] = Dev[a][b] - Dod[a][b]; }
                    }
                }
                if (gm.npairs > 0) {
                    const int tl = gm.npairs - 1;
                    store(mode == 0 ? fwd_map(tl, 0) : north_map(tl), Pa, false);
                    store(mode == 0 ? fwd_map(tl, 1) : south_map(tl), Pb, false);
                }
            }
    }
    if (fault) { std::printf("out of bounds access\n"); return 1e9; }
    // unfolded direct sums
    double err = 0.0;
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < R; ++r)
            for (int n = 0; n < N; ++n) {
                const size_t at = (size_t)m * c_moff + (size_t)r * c_rstride + n;
                const bool expected = !(mode == 0 && r < m);
                if (!expected) {
                    if (writes[at] != 0) { std::printf("write below the triangle m %d l %d\n", m, r); return 1e9; }
                    continue;
                }
                if (writes[at] != 1) { std::printf("mode %d m %d row %d n %d written %d times\n", mode, m, r, n, writes[at]); return 1e9; }
                double s = 0.0;
                for (int c = 0; c < K; ++c) {
                    const int l = mode == 0 ? r : c;
                    if (l < m) continue;
                    s += (double)tab[((size_t)m * R + r) * pitch + c] * (double)B[(size_t)m * b_moff + (size_t)c * b_kstride + n];
                }
                const double d = C[at] - s;
                err = std::max(err, d < 0 ? -d : d);
            }
    return err;
}

int main() {
    struct Case { int mode, H, L, M, N; } cases[] = {
        {0, 20, 18, 12, 40},   {1, 20, 18, 12, 40},   {0, 45, 45, 46, 32},     {1, 45, 45, 46, 32},     {0, 64, 40, 50, 128},
        {1, 64, 40, 50, 128},  {0, 9, 8, 10, 6},      {1, 9, 8, 10, 6},        {0, 180, 180, 181, 128}, {1, 180, 180, 181, 128},
        {0, 100, 90, 51, 256}, {1, 100, 90, 51, 256}, {0, 16, 16, 17, 48},     {1, 16, 16, 17, 48},     {0, 65, 48, 60, 32},
        {1, 65, 48, 60, 32},   {0, 192, 192, 97, 128}, {1, 192, 192, 97, 128}, {0, 181, 180, 91, 64},   {1, 181, 180, 91, 64},
        {0, 24, 24, 13, 12},   {1, 24, 24, 13, 12},   {0, 12, 12, 13, 20},     {1, 12, 12, 13, 20},     {0, 33, 32, 17, 36}, {1, 33, 32, 17, 36},
        // big form (more than 96 folded latitudes / degrees per parity): unit sizes 12, 18, 24, and the inverse's descent through every
        // size as the wavenumber grows; whole 128-column groups only (the big form's eligibility)
        {0, 400, 390, 12, 128}, {1, 400, 390, 12, 128}, {0, 721, 721, 3, 128}, {1, 721, 721, 3, 128}, {1, 230, 225, 226, 128}, {0, 201, 200, 9, 128},
    };
    double worst = 0.0;
    for (auto& c : cases) {
        const double e = run_case(c.mode, c.H, c.L, c.M, c.N, 4321u + c.H);
        std::printf("mode %d H %d L %d M %d N %d  err %.3e\n", c.mode, c.H, c.L, c.M, c.N, e);
        worst = std::max(worst, e);
    }
    std::printf("worst %.3e\n", worst);
    return worst < 4e-4 ? 0 : 1;
}
