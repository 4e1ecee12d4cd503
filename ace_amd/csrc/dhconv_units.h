// Work list of a dhconv_strip.hip launch: plain C++ shared by the launcher (host) and the CPU test (tests/emul/dhconv_units_emul.cpp).
#pragma once
#include <algorithm>
#include <vector>

namespace ace {

constexpr int DH_CHUNK_STRIPS = 3;                // 32-row strips per workgroup
constexpr int DH_CHUNK_ROWS = 32 * DH_CHUNK_STRIPS;

// Work list of one launch (see the kernel): 8 lists of `units_per_xcd` entries (l, j, row0, rows), padded with rows = 0.
// XCD x owns the degrees l = L - 1 - x, L - 9 - x, ...; a (degree, column group) with R rows is cut into ceil(R / 96) chunks.
// Groups are sorted by the size of their largest chunk, then by their row count (largest first), and their chunks emitted next to
// each other, the starting chunk rotating with the group's position: engine e = position % 4 then sees full and partial chunks alike.
inline int dhconv_units(int L, int Mrows, int trimul, int C, std::vector<int>& out) {
    struct Grp { int l, j, rows; };
    const int ncg = C / 128;
    std::vector<std::vector<int>> lists(8);
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) {
        std::vector<Grp> grps;
        for (int l = L - 1 - x; l >= 0; l -= 8) {
            const long want = (long)(l + 1) * trimul;
            const int rows_l = (int)(want < Mrows ? want : Mrows);
            for (int j = 0; j < ncg; ++j) grps.push_back({l, j, rows_l});
        }
        auto first_chunk = [](const Grp& g) { return g.rows < DH_CHUNK_ROWS ? (g.rows + 31) / 32 : DH_CHUNK_STRIPS; };
        std::stable_sort(grps.begin(), grps.end(), [&](const Grp& a, const Grp& b) {
            if (first_chunk(a) != first_chunk(b)) return first_chunk(a) > first_chunk(b);
            return a.rows > b.rows;
        });
        std::vector<int>& li = lists[x];
        for (size_t gi = 0; gi < grps.size(); ++gi) {
            const Grp& g = grps[gi];
            const int nch = (g.rows + DH_CHUNK_ROWS - 1) / DH_CHUNK_ROWS;
            const int rot = (int)((gi >> 1) % (size_t)nch);
            for (int k = 0; k < nch; ++k) {
                const int ch = (k + rot) % nch;
                const int row0 = ch * DH_CHUNK_ROWS;
                const int rows = g.rows - row0 < DH_CHUNK_ROWS ? g.rows - row0 : DH_CHUNK_ROWS;
                li.insert(li.end(), {g.l, g.j, row0, rows});
            }
        }
        longest = std::max(longest, li.size() / 4);
    }
    out.assign(8 * longest * 4, 0);
    for (int x = 0; x < 8; ++x) std::copy(lists[x].begin(), lists[x].end(), out.begin() + (size_t)x * longest * 4);
    return (int)longest;
}


}  // namespace ace
