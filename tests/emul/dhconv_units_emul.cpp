// CPU check of the work list of a dhconv_strip.hip launch (ace_amd/csrc/dhconv_units.h): every row of every (degree, column
// group) is covered by exactly one unit, on the XCD that owns the degree, chunks are at most 96 rows, the lists are padded with
// empty entries only at their ends, the four shader engines of an XCD (entry index % 4) get the same work to within two units or 8 %, and
// the chunks of one (degree, column group) are neighbours.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../ace_amd/csrc/dhconv_units.h"

static int check(int L, int Mm, int B, int C, int order) {
    std::vector<int> u;
    const int per = ace::dhconv_units(L, Mm * B, B, C, u, order);
    const int ncg = C / 128;
    std::map<std::pair<int, int>, std::vector<char>> seen;
    int bad = 0;
    long worst_spread = 0, least = 1L << 60;
    long xcd_strips[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::map<int, int> xcd_of_degree;
    for (int x = 0; x < 8; ++x) {
        long strips[4] = {0, 0, 0, 0};
        bool padding = false;
        int prev_l = -1, prev_j = -1, group_open = 0;
        std::map<std::pair<int, int>, int> last_pos;
        for (int k = 0; k < per; ++k) {
            const int* e = &u[((size_t)x * per + k) * 4];
            const int l = e[0], j = e[1], row0 = e[2], rows = e[3];
            if (rows == 0) { padding = true; continue; }
            if (padding) { std::printf("unit after padding (xcd %d, entry %d)\n", x, k); ++bad; }
            const long want = (long)(l + 1) * B;
            const int rows_l = (int)(want < (long)Mm * B ? want : (long)Mm * B);
            if (l < 0 || l >= L || (order == 0 && (L - 1 - l) % 8 != x) || j < 0 || j >= ncg || rows < 1 || rows > ace::DH_CHUNK_ROWS || row0 % ace::DH_CHUNK_ROWS ||
                row0 + rows > rows_l) { std::printf("bad unit l %d j %d row0 %d rows %d\n", l, j, row0, rows); ++bad; continue; }
            auto& s = seen[{l, j}];
            s.resize(rows_l, 0);
            for (int r = row0; r < row0 + rows; ++r) { if (s[r]) ++bad; s[r] = 1; }
            auto it = last_pos.find({l, j});
            if (it != last_pos.end() && it->second != k - 1) { std::printf("chunks of (l %d, j %d) apart\n", l, j); ++bad; }
            last_pos[{l, j}] = k;
            strips[k % 4] += (rows + 31) / 32;
            xcd_strips[x] += (rows + 31) / 32;
            if (xcd_of_degree.count(l) && xcd_of_degree[l] != x) { std::printf("degree %d on two XCDs\n", l); ++bad; }
            xcd_of_degree[l] = x;
            (void)prev_l; (void)prev_j; (void)group_open;
        }
        long lo = strips[0], hi = strips[0];
        for (int e = 1; e < 4; ++e) { lo = strips[e] < lo ? strips[e] : lo; hi = strips[e] > hi ? strips[e] : hi; }
        worst_spread = hi - lo > worst_spread ? hi - lo : worst_spread;
        least = lo < least ? lo : least;
    }
    for (int l = 0; l < L; ++l)
        for (int j = 0; j < ncg; ++j) {
            auto it = seen.find({l, j});
            if (it == seen.end()) { std::printf("(l %d, j %d) missing\n", l, j); ++bad; continue; }
            for (char c : it->second) if (!c) ++bad;
        }
    long xlo = xcd_strips[0], xhi = xcd_strips[0];
    for (int x = 1; x < 8; ++x) { xlo = xcd_strips[x] < xlo ? xcd_strips[x] : xlo; xhi = xcd_strips[x] > xhi ? xcd_strips[x] : xhi; }
    std::printf("order %d L %4d Mm %4d B %d C %4d: %5d entries per XCD, engine spread %ld strips, XCD strips %ld .. %ld, %s\n", order, L, Mm, B, C, per, worst_spread, xlo, xhi,
                bad ? "FAILED" : "ok");
    if (worst_spread > 2 * ace::DH_CHUNK_STRIPS && worst_spread * 100 > 8 * least) { std::printf("engines out of balance\n"); ++bad; }
    // orders 1, 2: degrees dealt longest first to the least loaded XCD - the lists differ by at most one degree's strips
    if (order != 0 && xhi - xlo > (long)(C / 128) * ((Mm * B + 31) / 32 + 1)) { std::printf("XCDs out of balance\n"); ++bad; }
    return bad;
}

int main() {
    int bad = 0;
    for (int order = 0; order <= 1; ++order) {
        bad += check(180, 181, 1, 384, order);
        bad += check(180, 181, 2, 384, order);
        bad += check(180, 181, 3, 512, order);
        bad += check(721, 721, 1, 384, order);
        bad += check(24, 25, 3, 128, order);
        bad += check(40, 41, 5, 256, order);
        bad += check(9, 10, 1, 128, order);
        bad += check(1, 1, 1, 128, order);
    }
    std::printf(bad ? "FAILED\n" : "worst: all lists consistent\n");
    return bad ? 1 : 0;
}
