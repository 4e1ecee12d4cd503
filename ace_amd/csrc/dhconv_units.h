// Work list of a dhconv_strip.hip launch: plain C++ shared by the launcher (host) and the CPU test (tests/emul/dhconv_units_emul.cpp).
#pragma once
#include <algorithm>
#include <vector>

#include "tuning_guard.h"   // checked before the default below is defined

namespace ace {

#ifndef ACE_DH_STRIPS
#define ACE_DH_STRIPS 3
#endif
constexpr int DH_CHUNK_STRIPS = ACE_DH_STRIPS;    // 32-row strips per workgroup
constexpr int DH_CHUNK_ROWS = 32 * DH_CHUNK_STRIPS;

// Work list of one launch (see the kernel): 8 lists of `units_per_xcd` entries (l, j, row0, rows), padded with rows = 0.
// A (degree, column group) with R rows is cut into ceil(R / 96) chunks, emitted next to each other (the second one reads the
// filter slice the first is streaming from that XCD's L2); all column groups of a degree live on one XCD (they share the
// coefficient rows through its L2).  The dispatcher deals entry k of an XCD's stream to shader engine k % 4 whatever the engines'
// load (tools/trace_dh.py); only the CUs of one engine share its queue dynamically.
//   order 1 (shipped; two workgroups per CU): degrees dealt to the XCDs longest first onto the least loaded one (225 strips each at
//     the ACE2 shape; l = L - 1 - x, L - 9 - x, ... gives 234 / 216), groups in descending order of their rows, the chunk order of a
//     group whichever leaves the four engines most even.
//   order 0 (round 6's first form, one workgroup per CU): XCD x owns l = L - 1 - x, L - 9 - x, ...; groups sorted by the size of
//     their largest chunk, then by their row count, the starting chunk rotating with the group's position.
// Measured and not kept (profiles/r06_dhconv_occupancy.txt): units longest first regardless of their group, 64 + rest chunks for
// 97 .. 128 rows, alternating long / short groups, persistent workgroups walking host-packed per-CU lists.
constexpr int DH_ORDER_DEFAULT = 1;
inline int dhconv_units(int L, int Mrows, int trimul, int C, std::vector<int>& out, int order = DH_ORDER_DEFAULT) {
    struct Grp { int l, j, rows; };
    const int ncg = C / 128;
    auto rows_of = [&](int l) { const long want = (long)(l + 1) * trimul; return (int)(want < Mrows ? want : Mrows); };
    auto chunk_rows = [](int rows, int ch) { const int row0 = ch * DH_CHUNK_ROWS; return rows - row0 < DH_CHUNK_ROWS ? rows - row0 : DH_CHUNK_ROWS; };
    auto strips_of = [&](int rows) {
        int s = 0;
        for (int ch = 0; ch * DH_CHUNK_ROWS < rows; ++ch) s += (chunk_rows(rows, ch) + 31) / 32;
        return s;
    };
    std::vector<std::vector<int>> owner(8);          // degrees of each XCD
    if (order == 0) {
        for (int x = 0; x < 8; ++x)
            for (int l = L - 1 - x; l >= 0; l -= 8) owner[x].push_back(l);
    } else {
        long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = L - 1; l >= 0; --l) {            // rows_of is non-decreasing in l: longest first
            int x = 0;
            for (int k = 1; k < 8; ++k) x = load[k] < load[x] ? k : x;
            owner[x].push_back(l);
            load[x] += strips_of(rows_of(l));
        }
    }
    std::vector<std::vector<int>> lists(8);
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) {
        std::vector<Grp> grps;
        for (int l : owner[x])
            for (int j = 0; j < ncg; ++j) grps.push_back({l, j, rows_of(l)});
        auto first_chunk = [](const Grp& g) { return g.rows < DH_CHUNK_ROWS ? (g.rows + 31) / 32 : DH_CHUNK_STRIPS; };
        if (order == 0)
            std::stable_sort(grps.begin(), grps.end(), [&](const Grp& a, const Grp& b) {
                if (first_chunk(a) != first_chunk(b)) return first_chunk(a) > first_chunk(b);
                return a.rows > b.rows;
            });
        else
            std::stable_sort(grps.begin(), grps.end(), [&](const Grp& a, const Grp& b) { return a.rows > b.rows; });
        std::vector<int>& li = lists[x];
        long eng[4] = {0, 0, 0, 0};                   // strips dealt to each engine so far (entry k -> engine k % 4)
        for (size_t gi = 0; gi < grps.size(); ++gi) {
            const Grp& g = grps[gi];
            const int nch = (g.rows + DH_CHUNK_ROWS - 1) / DH_CHUNK_ROWS;
            int rot = (int)((gi >> 1) % (size_t)nch);
            if (order != 0) {                         // the rotation that leaves the engines most even
                const size_t pos = li.size() / 4;
                long best = -1;
                for (int r = 0; r < nch; ++r) {
                    long e2[4] = {eng[0], eng[1], eng[2], eng[3]}, cost = 0;
                    for (int k = 0; k < nch; ++k) e2[(pos + k) % 4] += (chunk_rows(g.rows, (k + r) % nch) + 31) / 32;
                    for (int e = 0; e < 4; ++e) cost += e2[e] * e2[e];
                    if (best < 0 || cost < best) { best = cost; rot = r; }
                }
            }
            for (int k = 0; k < nch; ++k) {
                const int ch = (k + rot) % nch;
                const int rows = chunk_rows(g.rows, ch);
                eng[(li.size() / 4) % 4] += (rows + 31) / 32;
                li.insert(li.end(), {g.l, g.j, ch * DH_CHUNK_ROWS, rows});
            }
        }
        longest = std::max(longest, li.size() / 4);
    }
    out.assign(8 * longest * 4, 0);
    for (int x = 0; x < 8; ++x) std::copy(lists[x].begin(), lists[x].end(), out.begin() + (size_t)x * longest * 4);
    return (int)longest;
}


}  // namespace ace
