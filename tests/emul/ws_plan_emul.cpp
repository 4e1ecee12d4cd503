// CPU check of the conv_ws.hip work decomposition (ace_amd/csrc/ws_plan.h, the very code the kernel runs): for a sweep of
// (M, HW) every (channel slice, pixel tile) unit is walked exactly once, segments of one workgroup never overlap, every
// (statistics slot, row) receives exactly one partial - real or neutral - and no workgroup gets more than its fair share.
#include <cstdio>
#include <map>
#include <vector>

#include "../../ace_amd/csrc/ws_plan.h"

using namespace ace;

int main() {
    int checked = 0;
    long worst_num = 0, worst_den = 1;
    for (int M : {128, 256, 384, 512, 640, 768, 896, 1024, 1152, 2048})
        for (long HW : {1L, 31L, 32L, 33L, 164L, 800L, 1152L, 4050L, 4132L, 16200L, 64800L, 259200L, 1038240L})
            for (int allow = 0; allow < 2; ++allow) {
                const WsPlan pl = ws_plan(M, HW, allow != 0);
                const int tiles = (int)((HW + 31) / 32), S = pl.nslice, wpx = ws_workgroups_per_xcd(pl);
                if (wpx > 32 && pl.tpx > pl.F) { printf("FAIL wpx %d > 32 (M %d HW %ld)\n", wpx, M, HW); return 1; }
                std::vector<int> cover((size_t)S * tiles, 0);
                std::map<std::pair<int, int>, int> slot_rows;   // (slot, slice) -> partials written
                long maxwork = 0;
                for (int xcd = 0; xcd < 8; ++xcd)
                    for (int w = 0; w < wpx; ++w) {
                        const WsWork k = ws_work(pl, tiles, xcd, w);
                        long work = 0;
                        for (int s = 0; s < S; ++s) {
                            const bool a = ws_answers_for(pl, k, w, s), r = ws_reaches(pl, k, w, s);
                            if (a) slot_rows[{k.part_q, s}] += 1;        // neutral or real: one partial per row of the slice
                            if (r && !a) { printf("FAIL reaches a slice it does not answer for\n"); return 1; }
                        }
                        std::vector<char> reached(S, 0);
                        for (int i = 0; i < k.nseg; ++i) {
                            const WsSeg sg = ws_segment(pl, k, w, i);
                            if (sg.np <= 0 || sg.slice < 0 || sg.slice >= S || sg.tile0 < 0 || sg.tile0 + sg.np > tiles) {
                                printf("FAIL bad segment M %d HW %ld xcd %d w %d: slice %d tile0 %d np %d\n", M, HW, xcd, w, sg.slice, sg.tile0, sg.np);
                                return 1;
                            }
                            if (reached[sg.slice]) { printf("FAIL two segments in one slice\n"); return 1; }
                            reached[sg.slice] = 1;
                            if (!ws_reaches(pl, k, w, sg.slice)) { printf("FAIL segment outside the reached range\n"); return 1; }
                            for (int t = 0; t < sg.np; ++t) cover[(size_t)sg.slice * tiles + sg.tile0 + t] += 1;
                            work += sg.np;
                        }
                        for (int s = 0; s < S; ++s)
                            if (ws_reaches(pl, k, w, s) && !reached[s]) { printf("FAIL reached slice without a segment\n"); return 1; }
                        maxwork = work > maxwork ? work : maxwork;
                    }
                for (size_t q = 0; q < cover.size(); ++q)
                    if (cover[q] != 1) { printf("FAIL M %d HW %ld allow %d: unit %zu covered %d times\n", M, HW, allow, q, cover[q]); return 1; }
                for (int q = 0; q < ws_stat_slots(pl); ++q)
                    for (int s = 0; s < S; ++s)
                        if (slot_rows[{q, s}] != 1) { printf("FAIL M %d HW %ld: slot %d slice %d gets %d partials\n", M, HW, q, s, slot_rows[{q, s}]); return 1; }
                // balance: the busiest workgroup against the ideal share of 256 CUs (large fields only)
                if (tiles >= 2000 && (allow || pl.XG > 0)) {
                    const long ideal_num = (long)S * tiles, ideal_den = 256;
                    if (maxwork * ideal_den * worst_den > worst_num * ideal_num) { worst_num = maxwork * ideal_den; worst_den = ideal_num; }
                }
                ++checked;
            }
    printf("ws_plan: %d shapes ok, worst busiest-workgroup / ideal share = %.3f\n", checked, (double)worst_num / (double)worst_den);
    return 0;
}
