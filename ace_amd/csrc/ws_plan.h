// Work decomposition of one conv_ws.hip launch: which pixel tiles of which channel slice a workgroup walks.  Plain C++ shared by
// the kernel (device), its launcher (host) and the CPU emulation test (tests/emul/ws_plan_emul.cpp).
#pragma once

#if defined(__HIPCC__)
#define WS_HD __host__ __device__ inline
#else
#define WS_HD inline
#endif

namespace ace {

// XCD x owns the pixel tiles [x tpx, (x + 1) tpx) (workgroup b runs on XCD b % 8).  Its first `e` tiles are the EXTRA range
// shared by the R workgroups left over when nslice does not divide the 32 CUs of an XCD; the rest is cut into F groups of g
// tiles, each walked by nslice workgroups (one per channel slice of 128 output channels) at the same time, so that the
// activation is fetched from HBM once and from that XCD's L2 by the other slices.
// Launches that cannot loop over segments (K >= 512: no registers, allow_extra = false) and whose slice count does not divide 32
// (M = 384: 3 slices, 2 CUs of every XCD left over) use the 8 x (32 - F nslice) leftover workgroups as XG CROSS groups instead: group q
// = leftover workgroups q nslice .. q nslice + nslice - 1 (numbered x Lx + k over the XCDs, so its members sit on up to two XCDs and
// their tiles' activation is fetched by each of those L2s), one slice each, over the tile range [xg0 + q xgt, + xgt) behind the XCDs'
// own ranges.  r05 left those 16 CUs idle in fc2 (K = 768 -> M = 384): 26 tiles per workgroup, now 24.
struct WsPlan {
    int nslice, F, R, tpx, g, e;
    int Lx, XG, xg0, xgt;   // leftover workgroups per XCD, cross groups, their first tile, tiles per cross group
};
struct WsSeg {
    int slice, tile0, np;
};
// what workgroup w (0 .. F nslice + R - 1) of XCD xcd does: nseg segments, statistics slot part_q
struct WsWork {
    int x0, nx, ex, u0, u1, nseg, part_q;
    bool extra;
    int cross;            // -1, or the cross group this (leftover) workgroup belongs to; then xs / xt0 / xnp = its slice and tiles
    int xs, xt0, xnp;
};

WS_HD WsPlan ws_plan(int M, long HW, bool allow_extra) {
    WsPlan pl;
    pl.nslice = M / 128;
    pl.Lx = 0; pl.XG = 0; pl.xg0 = 0; pl.xgt = 0;
    const int tiles_px = (int)((HW + 31) / 32);
    pl.tpx = (tiles_px + 7) / 8;
    const int F0 = 32 / pl.nslice > 0 ? 32 / pl.nslice : 1;   // groups per XCD (32 CUs each)
    if (pl.tpx <= F0) {            // small fields: one tile per group, no leftovers worth sharing
        pl.F = pl.tpx; pl.R = 0; pl.g = 1; pl.e = 0;
        return pl;
    }
    pl.F = F0;
    const int left = 32 - F0 * pl.nslice > 0 ? 32 - F0 * pl.nslice : 0;
    if (!allow_extra && left > 0 && 8 * left >= pl.nslice) {   // cross groups of the leftover workgroups
        pl.R = 0; pl.e = 0;
        pl.Lx = left;
        pl.XG = 8 * left / pl.nslice;
        pl.g = (tiles_px + 8 * pl.F + pl.XG - 1) / (8 * pl.F + pl.XG);
        pl.tpx = pl.F * pl.g;                                   // an XCD's own range: F groups of g tiles
        pl.xg0 = 8 * pl.tpx;
        const int rest = tiles_px - pl.xg0 > 0 ? tiles_px - pl.xg0 : 0;
        pl.xgt = (rest + pl.XG - 1) / pl.XG;
        return pl;
    }
    pl.R = (allow_extra && left > 0) ? left : 0;
    pl.e = pl.R > 0 ? (pl.tpx * pl.R + 16) / 32 : 0;           // the extra workgroups take their share of the XCD's tiles
    if (pl.e * pl.nslice < pl.R) { pl.R = 0; pl.e = 0; }
    pl.g = (pl.tpx - pl.e + pl.F - 1) / pl.F;
    return pl;
}

WS_HD int ws_workgroups_per_xcd(const WsPlan& pl) { return pl.F * pl.nslice + pl.R + pl.Lx; }
WS_HD int ws_stat_slots(const WsPlan& pl) { return 8 * (pl.F + pl.R) + pl.XG; }   // statistics partials per row

WS_HD WsSeg ws_segment(const WsPlan& pl, const WsWork& k, int w, int i) {
    WsSeg sg;
    if (k.cross >= 0) {
        sg.slice = k.xs; sg.tile0 = k.xt0; sg.np = k.xnp;
    } else if (!k.extra) {
        const int grp = w / pl.nslice;
        sg.slice = w % pl.nslice;
        sg.tile0 = k.x0 + k.ex + grp * pl.g;
        const int end = k.x0 + k.ex + (grp + 1) * pl.g < k.x0 + k.nx ? k.x0 + k.ex + (grp + 1) * pl.g : k.x0 + k.nx;
        sg.np = end - sg.tile0;
    } else {
        const int r = w - pl.F * pl.nslice;
        const int ii = (r & 1) ? k.nseg - 1 - i : i;   // odd workgroups walk their slices backwards: pairs meet on the same tiles
        sg.slice = k.u0 / k.ex + ii;
        const int lo = k.u0 > sg.slice * k.ex ? k.u0 - sg.slice * k.ex : 0;
        const int hi = k.u1 < (sg.slice + 1) * k.ex ? k.u1 - sg.slice * k.ex : k.ex;
        sg.tile0 = k.x0 + lo;
        sg.np = hi - lo;
    }
    return sg;
}

WS_HD WsWork ws_work(const WsPlan& pl, int tiles_px, int xcd, int w) {
    WsWork k;
    k.x0 = xcd * pl.tpx;
    k.nx = tiles_px - k.x0 < pl.tpx ? (tiles_px - k.x0 > 0 ? tiles_px - k.x0 : 0) : pl.tpx;   // tiles of this XCD
    k.ex = pl.e < k.nx ? pl.e : k.nx;                                                         // ... of them in the extra range
    k.cross = -1; k.xs = 0; k.xt0 = 0; k.xnp = 0;
    if (pl.Lx > 0 && w >= pl.F * pl.nslice) {   // a leftover workgroup: member r % nslice of cross group r / nslice (or idle)
        const int r = xcd * pl.Lx + (w - pl.F * pl.nslice);
        k.extra = false; k.u0 = 0; k.u1 = 0; k.ex = 0;
        if (r >= pl.XG * pl.nslice) { k.nseg = 0; k.part_q = 0; k.cross = pl.XG; return k; }   // (answers for nothing)
        k.cross = r / pl.nslice;
        k.xs = r % pl.nslice;
        k.xt0 = pl.xg0 + k.cross * pl.xgt;
        const int end = k.xt0 + pl.xgt < tiles_px ? k.xt0 + pl.xgt : tiles_px;
        k.xnp = end - k.xt0 > 0 ? end - k.xt0 : 0;
        if (k.xnp == 0) k.xt0 = 0;
        k.nseg = k.xnp > 0 ? 1 : 0;
        k.part_q = 8 * (pl.F + pl.R) + k.cross;
        return k;
    }
    k.extra = w >= pl.F * pl.nslice;
    k.part_q = xcd * (pl.F + pl.R) + (k.extra ? pl.F + (w - pl.F * pl.nslice) : w / pl.nslice);
    k.u0 = 0; k.u1 = 0; k.nseg = 1;
    if (k.extra) {   // a contiguous run [u0, u1) of the units (slice, tile) = (u / ex, u % ex) of the extra range
        const int r = w - pl.F * pl.nslice, U = pl.nslice * k.ex;
        k.u0 = (int)((long)r * U / pl.R);
        k.u1 = (int)((long)(r + 1) * U / pl.R);
        k.nseg = k.u1 > k.u0 ? (k.u1 - 1) / k.ex - k.u0 / k.ex + 1 : 0;
    } else if (ws_segment(pl, k, w, 0).np <= 0) {
        k.nseg = 0;
    }
    return k;
}

// rows [128 s, 128 s + 128) of slot part_q are finished by this workgroup iff its segments reach slice s; the others of the
// rows it answers for (all rows for an extra workgroup, its own slice for a group member) get a neutral partial
WS_HD bool ws_answers_for(const WsPlan& pl, const WsWork& k, int w, int slice) {
    if (k.cross >= 0) return k.cross < pl.XG && slice == k.xs;
    return k.extra || slice == w % pl.nslice;
}
WS_HD bool ws_reaches(const WsPlan& pl, const WsWork& k, int w, int slice) {
    if (k.nseg <= 0) return false;
    if (k.cross >= 0) return slice == k.xs;
    const int lo = k.extra ? k.u0 / k.ex : w % pl.nslice;
    const int hi = k.extra ? (k.u1 - 1) / k.ex : w % pl.nslice;
    return slice >= lo && slice <= hi;
}

}  // namespace ace
