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
struct WsPlan {
    int nslice, F, R, tpx, g, e;
};
struct WsSeg {
    int slice, tile0, np;
};
// what workgroup w (0 .. F nslice + R - 1) of XCD xcd does: nseg segments, statistics slot part_q
struct WsWork {
    int x0, nx, ex, u0, u1, nseg, part_q;
    bool extra;
};

WS_HD WsPlan ws_plan(int M, long HW, bool allow_extra) {
    WsPlan pl;
    pl.nslice = M / 128;
    const int tiles_px = (int)((HW + 31) / 32);
    pl.tpx = (tiles_px + 7) / 8;
    const int F0 = 32 / pl.nslice > 0 ? 32 / pl.nslice : 1;   // groups per XCD (32 CUs each)
    if (pl.tpx <= F0) {            // small fields: one tile per group, no leftovers worth sharing
        pl.F = pl.tpx; pl.R = 0; pl.g = 1; pl.e = 0;
        return pl;
    }
    pl.F = F0;
    pl.R = (allow_extra && 32 - F0 * pl.nslice > 0) ? 32 - F0 * pl.nslice : 0;
    pl.e = pl.R > 0 ? (pl.tpx * pl.R + 16) / 32 : 0;           // the extra workgroups take their share of the XCD's tiles
    if (pl.e * pl.nslice < pl.R) { pl.R = 0; pl.e = 0; }
    pl.g = (pl.tpx - pl.e + pl.F - 1) / pl.F;
    return pl;
}

WS_HD int ws_workgroups_per_xcd(const WsPlan& pl) { return pl.F * pl.nslice + pl.R; }
WS_HD int ws_stat_slots(const WsPlan& pl) { return 8 * (pl.F + pl.R); }   // statistics partials per row (inner-skip mode)

WS_HD WsSeg ws_segment(const WsPlan& pl, const WsWork& k, int w, int i) {
    WsSeg sg;
    if (!k.extra) {
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
WS_HD bool ws_answers_for(const WsPlan& pl, const WsWork& k, int w, int slice) { return k.extra || slice == w % pl.nslice; }
WS_HD bool ws_reaches(const WsPlan& pl, const WsWork& k, int w, int slice) {
    if (k.nseg <= 0) return false;
    const int lo = k.extra ? k.u0 / k.ex : w % pl.nslice;
    const int hi = k.extra ? (k.u1 - 1) / k.ex : w % pl.nslice;
    return slice >= lo && slice <= hi;
}

}  // namespace ace
