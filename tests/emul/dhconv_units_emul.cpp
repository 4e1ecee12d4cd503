// CPU check of the work list of a dhconv_strip.hip launch (ace_amd/csrc/dhconv_units.h): every row of every (degree, column
// group) is covered by exactly one unit, on the XCD that owns the degree, chunks are at most 96 rows, the lists are padded with
// empty entries only at their ends, the four shader engines of an XCD (entry index % 4) get the same work to within two units or 8 %, and
// the chunks of one (degree, column group) are neighbours.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../ace_amd/csrc/dhconv_units.h"

static int check(int L, int Mm, int B, int C) {
    std::vector<int> u;
    const int per = ace::dhconv_units(L, Mm * B, B, C, u);
    const int ncg = C / 128;
    std::map<std::pair<int, int>, std::vector<char>> seen;
    int bad = 0;
    long worst_spread = 0, least = 1L << 60;
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
            if (l < 0 || l >= L || (L - 1 - l) % 8 != x || j < 0 || j >= ncg || rows < 1 || rows > ace::DH_CHUNK_ROWS || row0 % ace::DH_CHUNK_ROWS ||
                row0 + rows > rows_l) { std::printf("bad unit l %d j %d row0 %d rows %d\n", l, j, row0, rows); ++bad; continue; }
            auto& s = seen[{l, j}];
            s.resize(rows_l, 0);
            for (int r = row0; r < row0 + rows; ++r) { if (s[r]) ++bad; s[r] = 1; }
            auto it = last_pos.find({l, j});
            if (it != last_pos.end() && it->second != k - 1) { std::printf("chunks of (l %d, j %d) apart\n", l, j); ++bad; }
            last_pos[{l, j}] = k;
            strips[k % 4] += (rows + 31) / 32;
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
    std::printf("L %4d Mm %4d B %d C %4d: %5d entries per XCD, engine spread %ld strips, %s\n", L, Mm, B, C, per, worst_spread, bad ? "FAILED" : "ok");
    if (worst_spread > 2 * ace::DH_CHUNK_STRIPS && worst_spread * 100 > 8 * least) { std::printf("engines out of balance\n"); ++bad; }
    return bad;
}

int main() {
    int bad = 0;
    bad += check(180, 181, 1, 384);
    bad += check(180, 181, 2, 384);
    bad += check(180, 181, 3, 512);
    bad += check(721, 721, 1, 384);
    bad += check(24, 25, 3, 128);
    bad += check(40, 41, 5, 256);
    bad += check(9, 10, 1, 128);
    bad += check(1, 1, 1, 128);
    std::printf(bad ? "FAILED\n" : "worst: all lists consistent\n");
    return bad ? 1 : 0;
}
