// gfx950 (MI355X / CDNA4) kernels for the SFNO forward step.
//
// One fp32 tile engine - LDS-staged operands, v_mfma_f32_32x32x2_f32 accumulation
// (exact fp32, k-ordered fma chain, 64-lane wavefronts, 64x64 output per wave) -
// carries every contraction on the path:
//   * 1x1 convolutions (encoder, inner skip, MLP, decoder)  sfnonet.py:229, layers.py:117-124
//   * Legendre quadrature / synthesis, batched over m        sht_fix.py:134-138, 208-219
//   * dhconv spectral filter, batched over l                 contractions.py:183-195
//   * folded real DFT along longitude (forward and inverse)  fft.py:61-96
// Elementwise work (instance-norm affine, bias, exact GELU, residual adds,
// (de)normalisation) is fused into the operand loaders and epilogues so that
// no normalised / concatenated / transposed copy of an activation is ever
// written to HBM.
//
// Written for gfx950 only: wavefront = 64, no portability paths.
#include "kernels.h"

#include <cstdint>
#include <cstdlib>

namespace ace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DEVINL __device__ __forceinline__

// erf(x) = sign(x) (1 - 2^-q(|x|)), q a degree-9 fit of -log2(erfc(t)) on [0, 4] (erf(4) = 1 - 1.5e-8 rounds to 1).
// Max abs error 1.1e-7 (tools/fit_fast_erf.py), i.e. ~2 fp32 ulps of the result, for ~14 instructions instead of
// the ~60 of libm's erff.  Used by the compensated-fp16 engine, whose MFMA time no longer hides the epilogue.
DEVINL float fast_erf(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float q = -1.150086973e-05f;
    q = fmaf(q, t, 1.518900972e-04f);
    q = fmaf(q, t, -8.436889620e-04f);
    q = fmaf(q, t, 2.264559502e-03f);
    q = fmaf(q, t, -7.151089812e-05f);
    q = fmaf(q, t, -2.773463540e-02f);
    q = fmaf(q, t, 1.483123451e-01f);
    q = fmaf(q, t, 9.184418917e-01f);
    q = fmaf(q, t, 1.627907395e+00f);
    q = q * t;
    return copysignf(1.0f - __builtin_amdgcn_exp2f(-q), x);
}

// Dynamic-range slots are written with device-scope atomics (performed at the memory side on this multi-XCD part);
// read them the same way so that a copy of the line left in this XCD's L2 by an earlier forward is never used.
DEVINL unsigned slot_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// power of two that maps a bound to [2^11, 2^12) (fp16 keeps 11 more bits below; lo parts stay normal for every
// element within 2^-13 of the bound and lose absolute, not relative, accuracy below that)
DEVINL int pow2_exponent_for(float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 12 - e; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}

DEVINL float act_apply(float v, int act) {
    switch (act) {
        case ACT_GELU_FAST: return 0.5f * v * (1.0f + fast_erf(v * 0.70710678118654752440f));
        case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // exact erf GELU (nn.GELU)
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SILU: return v / (1.0f + expf(-v));
        default: return v;
    }
}

// Activation chosen once per kernel instead of once per element: inside an epilogue loop a runtime `switch (act)` puts
// a branch chain between every LDS read and its use (the reads serialise on s_waitcnt, nothing is batched).  A < 0: the
// runtime code (rare activations).
template <int A>
DEVINL float act_const(float v, int act_rt) { return A < 0 ? act_apply(v, act_rt) : act_apply(v, A); }
template <int A> struct ActTag { static constexpr int value = A; };
template <class F>
DEVINL void dispatch_act(int act, F&& f) {
    if (act == ACT_NONE) f(ActTag<ACT_NONE>{});
    else if (act == ACT_GELU_FAST) f(ActTag<ACT_GELU_FAST>{});
    else f(ActTag<-1>{});
}

// Workgroup b is observed to run on XCD b % 8.  Give each XCD a contiguous chunk of
// logical tiles so tiles that share an operand panel also share an L2 (speed only;
// bijective for any nblk).
DEVINL int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// row of accumulator register r for lane-half h in a 32x32 MFMA tile
DEVINL int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int BK = 16;  // k-depth of one LDS stage

// ---------------------------------------------------------------------------------------------
// operand stagers: global -> registers (issued early, never touched before the MFMAs) -> LDS (after them).
// Loads are branch-free: out-of-range lanes read a clamped in-range address and the value is zeroed by a
// select, so the compiler emits straight-line global_load_dwordx4 with no wait between them.
// ---------------------------------------------------------------------------------------------

// A tile from a row-major [M][K] matrix, stored k-major in LDS: As[k][m] (pitch SA, SA % 8 == 2 so the
// four transposing ds_write_b32 of a lane group hit 32 distinct banks).
// VEC requires: base 16B aligned, lda % 4 == 0, kend % 4 == 0.
template <int BM, int NT, bool VEC>
struct StageA {
    static constexpr int G = BM * BK / 4 / NT;
    static constexpr int SA = BM + 2;
    float4 r[G];
    // raw loads only (clamped addresses): nothing here may consume a loaded value, or the compiler has to
    // wait for the whole stage before the MFMAs.  Out-of-range elements are zeroed in store().
    DEVINL void load(const float* __restrict__ A, long lda, int m0, int M, int k0, int kend, int tid) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int row = f / (BK / 4), kc = f % (BK / 4);
            const int m = m0 + row, k = k0 + 4 * kc;
            const float* prow = A + (long)(m < M ? m : 0) * lda;
            if (VEC) {
                r[g] = *reinterpret_cast<const float4*>(prow + (k < kend ? k : 0));
            } else {
                r[g] = make_float4(prow[k < kend ? k : 0], prow[k + 1 < kend ? k + 1 : 0], prow[k + 2 < kend ? k + 2 : 0],
                                   prow[k + 3 < kend ? k + 3 : 0]);
            }
        }
    }
    DEVINL void store(float* As, int m0, int M, int k0, int kend, int tid) const {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int row = f / (BK / 4), kc = f % (BK / 4);
            const int k = k0 + 4 * kc;
            const bool rok = m0 + row < M;
            float* d = As + (4 * kc) * SA + row;
            d[0] = (rok && k < kend) ? r[g].x : 0.f;
            d[SA] = (rok && k + 1 < kend) ? r[g].y : 0.f;
            d[2 * SA] = (rok && k + 2 < kend) ? r[g].z : 0.f;
            d[3 * SA] = (rok && k + 3 < kend) ? r[g].w : 0.f;
        }
    }
};

// B tile from a row-major [K][N] matrix: Bs[k][n] (pitch SB = BN + 4, 16-byte rows -> ds_write_b128).
// Optional second source for rows k >= K1 (the big-skip concat) and optional per-row affine (fused instance
// norm): the scale/shift are fetched with the tile and applied when the tile is written to LDS.
// VEC requires: bases 16B aligned, ldb % 4 == 0, N % 4 == 0.
template <int BN, int NT, bool VEC, bool AFF>
struct StageB {
    static constexpr int G = BN * BK / 4 / NT;
    static constexpr int SB = BN + 4;
    float4 r[G];
    float sc[AFF ? G : 1], sh[AFF ? G : 1];
    // raw loads only (see StageA::load)
    DEVINL void load(const float* __restrict__ B, long ldb, const float* __restrict__ B2, long ldb2, int K1,
                     const float* __restrict__ bsc, const float* __restrict__ bsh, int n0, int N, int k0, int kend,
                     int tid, int kvalid = 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int kr = f / (BN / 4), nc = f % (BN / 4);
            const int k = k0 + kr, n = n0 + 4 * nc;
            const int kk = ((k < kend) && (k >= kvalid)) ? k : kvalid;  // kvalid < kend whenever the tile is reached
            const float* prow = (K1 >= 0 && kk >= K1) ? (B2 + (long)(kk - K1) * ldb2) : (B + (long)kk * ldb);
            if (VEC) {
                r[g] = *reinterpret_cast<const float4*>(prow + (n < N ? n : 0));
            } else {
                r[g] = make_float4(prow[n < N ? n : 0], prow[n + 1 < N ? n + 1 : 0], prow[n + 2 < N ? n + 2 : 0],
                                   prow[n + 3 < N ? n + 3 : 0]);
            }
            if (AFF) { sc[g] = bsc[kk]; sh[g] = bsh[kk]; }
        }
    }
    DEVINL void store(float* Bs, int n0, int N, int k0, int kend, int tid, int kvalid = 0) const {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int kr = f / (BN / 4), nc = f % (BN / 4);
            const int k = k0 + kr, n = n0 + 4 * nc;
            const bool kok = (k < kend) && (k >= kvalid);
            float4 v = r[g];
            if (AFF) {
                v.x = fmaf(v.x, sc[g], sh[g]);
                v.y = fmaf(v.y, sc[g], sh[g]);
                v.z = fmaf(v.z, sc[g], sh[g]);
                v.w = fmaf(v.w, sc[g], sh[g]);
            }
            v.x = (kok && n < N) ? v.x : 0.f;
            v.y = (kok && n + 1 < N) ? v.y : 0.f;
            v.z = (kok && n + 2 < N) ? v.z : 0.f;
            v.w = (kok && n + 3 < N) ? v.w : 0.f;
            *reinterpret_cast<float4*>(Bs + kr * SB + 4 * nc) = v;
        }
    }
};

// one LDS stage of MFMAs for a 64x64 wave tile: 2x2 tiles of 32x32, K = 2 per instruction; lane half h supplies
// k = ks + h (A[i][k] / B[k][j] fragments are one VGPR each).  Fragments are fetched PF k-steps ahead of the
// MFMAs that consume them so the LDS round trip hides under the previous MFMAs.
template <int SA, int SB>
DEVINL void mma_stage(const float* As, const float* Bs, int arow, int bcol, int h, f32x16 (&acc)[2][2]) {
    constexpr int NS = BK / 2;  // k-steps per stage
#ifndef ACE_PF
#define ACE_PF 2
#endif
    constexpr int PF = ACE_PF;  // prefetch distance in k-steps
    float fa[NS][2], fb[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        fa[s][0] = As[(2 * s + h) * SA + arow];
        fa[s][1] = As[(2 * s + h) * SA + arow + 32];
        fb[s][0] = Bs[(2 * s + h) * SB + bcol];
        fb[s][1] = Bs[(2 * s + h) * SB + bcol + 32];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][1], acc[1][1], 0, 0, 0);
    }
    // pin the interleave: fragments for k-step s+PF are read (2 ds_read2_b32) while the MFMAs of step s run
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (s + PF < NS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// batched GEMM with fused epilogue
// ---------------------------------------------------------------------------------------------
#ifndef ACE_LB
#define ACE_LB 1
#endif
template <int WM, int WN, bool VEC, bool AFF, bool RES>
__global__ __launch_bounds__(64 * WM * WN, ACE_LB) void gemm_f32_kernel(GemmArgs p, int tilesM, int tilesN) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    using SA_t = StageA<BM, NT, VEC>;
    using SB_t = StageB<BN, NT, VEC, AFF>;
    constexpr int SA = SA_t::SA, SB = SB_t::SB;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * SA + 2 * BK * SB];
    float* As = smem;
    float* Bs = smem + 2 * BK * SA;

    const int nblk = tilesM * tilesN * p.nbatch;
    const int lid = xcd_remap(blockIdx.x, nblk);
    const int tile_m = lid % tilesM;
    const int rest = lid / tilesM;
    const int tile_n = rest % tilesN;
    const int batch = rest / tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int M = p.M, kbeg = 0, kvalid = 0;
    if (p.tri == TRI_ROWS_GE_BATCH) {
        if (m0 + BM <= batch) return;  // rows l < m: table is zero there, nothing downstream reads them
    } else if (p.tri == TRI_K_GE_BATCH) {
        kbeg = (batch / BK) * BK;
        kvalid = batch;  // rows l < m of the spectral operand are never written: read them as zero
    } else if (p.tri == TRI_ROWS_LE_BATCH) {
        const int me = (batch + 1) * p.trimul;
        M = me < M ? me : M;
        if (m0 >= M) return;
    }
    const int kend = p.K;

    const float* A = p.A + (long)batch * p.sA;
    const float* B = p.B + (long)batch * p.sB;
    const float* B2 = p.B2 ? p.B2 + (long)batch * p.sB2 : nullptr;
    const float* bsc = AFF ? p.bsc + (long)batch * p.sbs : nullptr;
    const float* bsh = AFF ? p.bsh + (long)batch * p.sbs : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int arow = wm * 64 + i, bcol = wn * 64 + i;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    SA_t sa;
    SB_t sb;
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        sa.load(A, p.lda, m0, M, kbeg, kend, tid);
        sb.load(B, p.ldb, B2, p.ldb2, p.K1, bsc, bsh, n0, p.N, kbeg, kend, tid, kvalid);
        sa.store(As, m0, M, kbeg, kend, tid);
        sb.store(Bs, n0, p.N, kbeg, kend, tid, kvalid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        const int k0 = kbeg + (kt + 1) * BK;
#ifndef ACE_EXP_NOGLOBAL
        if (more) {  // next stage's global loads fly under this stage's MFMAs
            sa.load(A, p.lda, m0, M, k0, kend, tid);
            sb.load(B, p.ldb, B2, p.ldb2, p.K1, bsc, bsh, n0, p.N, k0, kend, tid, kvalid);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);  // nothing that consumes the loads may move above the MFMAs
#ifdef ACE_EXP_NOLDSREAD
        {   // timing probe: operands from registers (no LDS traffic in the loop)
            const float xa = (float)(tid & 7) * 0.25f, xb = (float)(tid & 3) * 0.5f;
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, acc[1][1], 0, 0, 0);
            }
        }
#else
        mma_stage<SA, SB>(As + cur * BK * SA, Bs + cur * BK * SB, arow, bcol, h, acc);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef ACE_EXP_NOLDSWRITE
        if (more) {
            sa.store(As + (cur ^ 1) * BK * SA, m0, M, k0, kend, tid);
            sb.store(Bs + (cur ^ 1) * BK * SB, n0, p.N, k0, kend, tid, kvalid);
        }
#endif
#ifndef ACE_EXP_NOBARRIER
        __syncthreads();
#endif
    }

    // epilogue: lanes run along columns (128-byte row segments per store instruction).  Per 4-row register quad:
    // all row parameters and all residual values are fetched first (clamped addresses, no branches), then the
    // 8 outputs are finished and stored.
    float* C = p.C + (long)batch * p.sC;
    const float* R = RES ? p.R + (long)batch * p.sR : nullptr;
    const float* rsc = p.rsc ? p.rsc + (long)batch * p.srs : nullptr;
    const float* rsh = p.rsc ? p.rsh + (long)batch * p.srs : nullptr;
    const float* bias1 = p.bias ? p.bias + (long)batch * p.sbias : nullptr;
    const float* osc = p.osc;
    const float* osh = p.osh;
    const long ldc = p.ldc, ldr = p.ldr;
    const int actk = p.act;
    float vmax = 0.f;
    const int col0 = n0 + wn * 64 + i, col1 = col0 + 32;
    const bool c0ok = col0 < p.N, c1ok = col1 < p.N;
    const int cc0 = c0ok ? col0 : 0, cc1 = c1ok ? col1 : 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rbase = m0 + wm * 64 + tm * 32 + 8 * q + 4 * h;
            float bv[4], rs[4], rt[4], os[4], ot[4], rv0[4], rv1[4];
            int rr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rr[e] = (rbase + e < M) ? rbase + e : 0;
                // optional row parameters without branches: absent ones read a valid dummy address (A) and are
                // replaced by the neutral value with a select, so all loads of the quad issue back to back
                const float t0 = (bias1 ? bias1 : A)[bias1 ? rr[e] : 0];
                const float t1 = (rsc ? rsc : A)[rsc ? rr[e] : 0];
                const float t2 = (rsc ? rsh : A)[rsc ? rr[e] : 0];
                const float t3 = (osc ? osc : A)[osc ? rr[e] : 0];
                const float t4 = (osc ? osh : A)[osc ? rr[e] : 0];
                bv[e] = bias1 ? t0 : 0.f;
                rs[e] = rsc ? t1 : 1.f;
                rt[e] = rsc ? t2 : 0.f;
                os[e] = osc ? t3 : 1.f;
                ot[e] = osc ? t4 : 0.f;
                if (RES) {
                    rv0[e] = R[(long)rr[e] * ldr + cc0];
                    rv1[e] = R[(long)rr[e] * ldr + cc1];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool rok = rbase + e < M;
                float v0 = acc[tm][0][4 * q + e] + bv[e];
                float v1 = acc[tm][1][4 * q + e] + bv[e];
                if (RES) {
                    v0 += fmaf(rv0[e], rs[e], rt[e]);
                    v1 += fmaf(rv1[e], rs[e], rt[e]);
                }
                v0 = fminf(act_apply(v0, actk), p.cap);
                v1 = fminf(act_apply(v1, actk), p.cap);
                v0 = fmaf(v0, os[e], ot[e]);
                v1 = fmaf(v1, os[e], ot[e]);
                float* crow = C + (long)rr[e] * ldc;
                if (rok && c0ok) { crow[col0] = v0; vmax = fmaxf(vmax, fabsf(v0)); }
                if (rok && c1ok) { crow[col1] = v1; vmax = fmaxf(vmax, fabsf(v1)); }
            }
        }
    }
    if (p.omax) {  // bit patterns of non-negative floats order like the floats
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(vmax));
    }
}

// ---------------------------------------------------------------------------------------------
// batched GEMM, direct-to-LDS variant ("v2"): both operand tiles are brought in with
// global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPR staging, no LDS write pass).
//   * B tile  Bs[k][n]  (n contiguous): lane-linear image of BK rows of the row-major source.
//   * A tile  As[m][k]  (k contiguous, pitch BK): lane-linear too; the 16-byte slot a lane FETCHES is
//     XOR-swizzled on the source side (slot ^ (row / rows-per-256B) mod slots) so that the per-lane
//     ds_read_b128 fragment reads of 16 different rows spread over all 64 banks.
//   * k permutation: lane half h owns k in [h*BK/2, (h+1)*BK/2) of the stage, so an A fragment is BK/8
//     consecutive float4 of one row; MFMA sums are order-insensitive up to fp32 rounding of the chain.
// Requirements (checked by the launcher): 16B-aligned bases, lda/ldb % 4 == 0, N % 4 == 0, every A row
// readable AND zero for k in [K, roundup(K, BK)) (library-owned padded weights/tables, or K % BK == 0).
// Out-of-range rows/columns are CLAMPED to valid addresses: the zero A columns annihilate the extra B rows,
// extra A rows / B columns only feed outputs that are never stored.  No per-k affine (the caller folds
// the instance-norm affine into the weights).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char* lds_cp;

// One 1-KiB LDS-DMA piece: lane L's 16 bytes at gsrc land at lds_base + 16*L.  Issued as inline asm so that
// hipcc does not order later ds_reads of the OTHER buffer behind it (it would insert vmcnt(0) before every
// fragment read); completion is waited for by hand (vmcnt(0) + barrier) at the end of the stage.
DEVINL void glds16(const float* gsrc, float* lds_base_uniform) {
    unsigned keep;
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cp)lds_base_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(addr)
                 : "memory");
}

// 4 bytes per lane variant: used as an L2 prefetch ("touch": one lane per 128-byte line, data discarded into a
// scratch corner of LDS - no VGPR destination that a late return could clobber)
DEVINL void glds4(const void* gsrc, void* lds_base_uniform) {
    unsigned keep;
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cp)lds_base_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(addr)
                 : "memory");
}

template <int WM, int WN, int BKT, bool RES>
__global__ __launch_bounds__(64 * WM * WN) void gemm2_f32_kernel(GemmArgs p, int tilesM, int tilesN) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
    constexpr int SL = BKT / 4;             // 16-byte slots per A row
    constexpr int RW = 64 / BKT;            // A rows per 256-byte bank window
    constexpr int ACH = BM * BKT / 256;     // 1 KiB chunks of the A tile
    constexpr int BCH = BKT * BN / 256;     // 1 KiB chunks of the B tile
    constexpr int AROWS = 256 / BKT;        // A rows per chunk
    constexpr int ABUF = BM * BKT, BBUF = BKT * BN;  // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    float* As = smem2;              // [2][BM][BKT]
    float* Bs = smem2 + 2 * ABUF;   // [2][BKT][BN]

    const int nblk = tilesM * tilesN * p.nbatch;
    const int lid = xcd_remap(blockIdx.x, nblk);
    const int tile_m = lid % tilesM;
    const int rest = lid / tilesM;
    const int tile_n = rest % tilesN;
    const int batch = rest / tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int M = p.M, kbeg = 0;
    if (p.tri == TRI_ROWS_GE_BATCH) {
        if (m0 + BM <= batch) return;
    } else if (p.tri == TRI_K_GE_BATCH) {
        kbeg = (batch / BKT) * BKT;
    } else if (p.tri == TRI_ROWS_LE_BATCH) {
        const int me = (batch + 1) * p.trimul;
        M = me < M ? me : M;
        if (m0 >= M) return;
    }
    const int K = p.K;
    const float* A = p.A + (long)batch * p.sA;
    const float* B = p.B + (long)batch * p.sB;
    const float* B2 = p.B2 ? p.B2 + (long)batch * p.sB2 : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- per-lane source descriptors of the chunks this wave fetches (fixed over the k loop)
    constexpr int ACW = (ACH + NW - 1) / NW, BCW = (BCH + NW - 1) / NW;
    const float* asrc[ACW];
    int bn[BCW];   // clamped column of the 16-byte piece this lane fetches
    int bkr[BCW];  // its k row inside the stage
#pragma unroll
    for (int c = 0; c < ACW; ++c) {
        const int ca = wave + c * NW;
        const int row = ca * AROWS + lane / SL;
        const int ps = lane % SL;
        const int ls = ps ^ ((row / RW) % SL);
        int m = m0 + row;
        m = m < M ? m : M - 1;
        asrc[c] = A + (long)m * p.lda + 4 * ls;
    }
#pragma unroll
    for (int c = 0; c < BCW; ++c) {
        const int cb = wave + c * NW;
        bkr[c] = (cb * 256 + lane * 4) / BN;
        const int n = n0 + (cb * 256 + lane * 4) % BN;
        bn[c] = n < p.N ? n : p.N - 4;
    }

    // kernel-argument fields used inside the loop, as SSA values: the asm "memory" clobber of glds16 would
    // otherwise force a reload (global_load + vmcnt(0)) of each of them after every DMA piece
    const long ldb = p.ldb, ldb2 = p.ldb2;
    const int K1 = p.K1;

    auto issue = [&](int k0, int buf) {
        float* Ab = As + buf * ABUF;
        float* Bb = Bs + buf * BBUF;
#pragma unroll
        for (int c = 0; c < ACW; ++c) {
            const int ca = wave + c * NW;
            if (ACH % NW == 0 || ca < ACH)
                glds16(asrc[c] + k0, Ab + ca * 256);
        }
#pragma unroll
        for (int c = 0; c < BCW; ++c) {
            const int cb = wave + c * NW;
            if (BCH % NW == 0 || cb < BCH) {
                int k = k0 + bkr[c];
                k = k < K ? k : K - 1;
                const long n = bn[c];
                const float* src = (K1 >= 0 && k >= K1) ? (B2 + (long)(k - K1) * ldb2 + n) : (B + (long)k * ldb + n);
                glds16(src, Bb + cb * 256);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addressing
    const int arow0 = wm * 64 + i, arow1 = arow0 + 32;
    const int key0 = (arow0 / RW) % SL, key1 = (arow1 / RW) % SL;
    const int bcol = wn * 64 + i;
    constexpr int NS = BKT / 2;   // k-steps per stage
    constexpr int NV = BKT / 8;   // float4 fragments per row per stage

    const int nk = (K - kbeg + BKT - 1) / BKT;
    if (nk > 0) issue(kbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kbeg + (kt + 1) * BKT, cur ^ 1);
        const float* Ab = As + cur * ABUF;
        const float* Bb = Bs + cur * BBUF + (h * NS) * BN + bcol;
        float4 a0[NV], a1[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            a0[j] = *reinterpret_cast<const float4*>(Ab + arow0 * BKT + 4 * ((h * NV + j) ^ key0));
            a1[j] = *reinterpret_cast<const float4*>(Ab + arow1 * BKT + 4 * ((h * NV + j) ^ key1));
        }
        float b0[NS], b1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            b0[s] = Bb[s * BN];
            b1[s] = Bb[s * BN + 32];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 va = a0[s / 4], vb = a1[s / 4];
            const float x0 = (s % 4 == 0) ? va.x : (s % 4 == 1) ? va.y : (s % 4 == 2) ? va.z : va.w;
            const float x1 = (s % 4 == 0) ? vb.x : (s % 4 == 1) ? vb.y : (s % 4 == 2) ? vb.z : vb.w;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b1[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, b0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, b1[s], acc[1][1], 0, 0, 0);
        }
        // pin the interleave: all A fragments + the first PF B pairs up front, then one B pair per MFMA quad
        constexpr int PF2 = 3;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NV + PF2, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (s + PF2 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's next-stage DMA has landed ...
        __syncthreads();                                  // ... and so has everyone's; cur may be overwritten
    }

    // epilogue (same as v1)
    float* C = p.C + (long)batch * p.sC;
    const float* R = RES ? p.R + (long)batch * p.sR : nullptr;
    const float* rsc = p.rsc ? p.rsc + (long)batch * p.srs : nullptr;
    const float* rsh = p.rsc ? p.rsh + (long)batch * p.srs : nullptr;
    const float* bias = p.bias ? p.bias + (long)batch * p.sbias : nullptr;
    const float* osc = p.osc;
    const float* osh = p.osh;
    const long ldc = p.ldc, ldr = p.ldr;
    const int actk = p.act;
    float vmax = 0.f;
    const int col0 = n0 + wn * 64 + i, col1 = col0 + 32;
    const bool c0ok = col0 < p.N, c1ok = col1 < p.N;
    const int cc0 = c0ok ? col0 : 0, cc1 = c1ok ? col1 : 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rbase = m0 + wm * 64 + tm * 32 + 8 * q + 4 * h;
            float bv[4], rs[4], rt[4], os[4], ot[4], rv0[4], rv1[4];
            int rr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rr[e] = (rbase + e < M) ? rbase + e : 0;
                // optional row parameters without branches: absent ones read a valid dummy address (A) and are
                // replaced by the neutral value with a select, so all loads of the quad issue back to back
                const float t0 = (bias ? bias : A)[bias ? rr[e] : 0];
                const float t1 = (rsc ? rsc : A)[rsc ? rr[e] : 0];
                const float t2 = (rsc ? rsh : A)[rsc ? rr[e] : 0];
                const float t3 = (osc ? osc : A)[osc ? rr[e] : 0];
                const float t4 = (osc ? osh : A)[osc ? rr[e] : 0];
                bv[e] = bias ? t0 : 0.f;
                rs[e] = rsc ? t1 : 1.f;
                rt[e] = rsc ? t2 : 0.f;
                os[e] = osc ? t3 : 1.f;
                ot[e] = osc ? t4 : 0.f;
                if (RES) {
                    rv0[e] = R[(long)rr[e] * ldr + cc0];
                    rv1[e] = R[(long)rr[e] * ldr + cc1];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool rok = rbase + e < M;
                float v0 = acc[tm][0][4 * q + e] + bv[e];
                float v1 = acc[tm][1][4 * q + e] + bv[e];
                if (RES) {
                    v0 += fmaf(rv0[e], rs[e], rt[e]);
                    v1 += fmaf(rv1[e], rs[e], rt[e]);
                }
                v0 = fminf(act_apply(v0, actk), p.cap);
                v1 = fminf(act_apply(v1, actk), p.cap);
                v0 = fmaf(v0, os[e], ot[e]);
                v1 = fmaf(v1, os[e], ot[e]);
                float* crow = C + (long)rr[e] * ldc;
                if (rok && c0ok) { crow[col0] = v0; vmax = fmaxf(vmax, fabsf(v0)); }
                if (rok && c1ok) { crow[col1] = v1; vmax = fmaxf(vmax, fabsf(v1)); }
            }
        }
    }
    if (p.omax) {  // bit patterns of non-negative floats order like the floats
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(vmax));
    }
}

// ---------------------------------------------------------------------------------------------
// batched GEMM, compensated-fp16 variant ("v3", f16x3): fp32-class accuracy on the 16x faster fp16 MFMA.
// Every fp32 operand is split x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits after a
// power-of-two pre-scale that keeps hi/lo in fp16's normal range); fp16 x fp16 products are exact in fp32, so
//     A.B  ~=  Ahi.Bhi + Ahi.Blo + Alo.Bhi          (dropped lo.lo term ~ 2^-22 relative)
// on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 MFMAs of 32 cycles do the work of 8 fp32 MFMAs of 64.
//   * A (weights / tables) is pre-split once by split_f16_kernel into two fp16 planes [M][lda] (zero padded to the
//     stage depth) and DMA'd to LDS with global_load_lds, source-swizzled for conflict-free ds_read_b128.
//   * B (activations, fp32 in HBM) is split on the fly: each thread loads 8 consecutive k rows of 2/4 columns,
//     applies the optional per-row affine (fused instance norm) and the power-of-two scale, converts, and writes
//     16-byte k-packed entries Bs[k/8][n][8 halves] - exactly the MFMA B fragment of one lane.
//   * epilogue as v1/v2 with the product of the two scales undone (exact, powers of two).
// ---------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

struct Gemm3Args {
    GemmArgs g;                 // B, C, epilogue, shapes (g.A unused)
    const _Float16* Ahi = nullptr; const _Float16* Alo = nullptr;  // [M][lda] fp16 planes, lda in halves (= g.lda)
    float bscale = 1.f;         // power of two applied to B before the split
    float oscale = 1.f;         // 1 / (ascale * bscale), applied to the accumulator
    // dynamic range management: if bmax != nullptr it holds the bit pattern of max|B| (written by the producer's
    // epilogue); the kernel then derives bscale = 2^(11 - exponent(max)) itself and oscale = oscale_a / bscale
    const unsigned* bmax = nullptr;
    const unsigned* bmax2 = nullptr;  // optional second source (B2 of the concat)
    unsigned* omax = nullptr;   // if set: atomicMax of the bit pattern of max|C| over this launch (64 shards)
    // optional: write C as two fp16 planes (hi, lo) in C's own row-major layout instead of fp32 - the "A format" of the
    // v4 engine (the consumer contracts over C's columns).  Scaled by 2^(12 - exponent(cw * bound(B))), bound published
    // to cslot.  Vector epilogue only (N % 4 == 0, ldc % 4 == 0).
    _Float16* Chi = nullptr; _Float16* Clo = nullptr;
    float cw = 0.f;
    unsigned* cslot = nullptr;
};

constexpr int G3_KAFF = 1024;  // largest K whose per-row affine is kept in LDS by the f16x3 engine

template <int WM, int WN, bool AFF, bool RES, bool ADYN = false>
__global__ __launch_bounds__(128 * WM * WN, 4) void gemm3_f16x3_kernel(Gemm3Args q, int tilesM, int tilesN) {
    // Wave-specialised: waves [0, NW) are MMA waves (LDS fragments -> MFMA -> epilogue), waves [NW, 2 NW) are loader
    // waves (A DMA issue, B row loads, hi/lo split, LDS writes).  The split's VALU work and the MFMAs run on the
    // same SIMDs at the same time (separate pipes); one workgroup barrier per stage hands buffers over.
    constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN, NT = 128 * NW, NL = 64 * NW;
    static_assert(NL == 256, "B stager mapping assumes 256 loader threads");
    constexpr int BKT = 32;                   // k per stage
    constexpr int NPT = BN / 64;              // columns per loader thread (2 or 4)
    constexpr int APL = BM * BKT;             // halves per A plane per buffer
    constexpr int BPL = BKT * BN;             // halves per B plane per buffer
    constexpr int ACH = BM * BKT * 2 / 1024;  // 1 KiB DMA pieces per A plane
    constexpr int ACW = (ACH + NW - 1) / NW;  // pieces per loader wave per plane
    static_assert(ACH % NW == 0, "A pieces must divide evenly over the loader waves (counted vmcnt)");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem3[];
    _Float16* As = smem3;                     // [2 buf][2 plane][BM][32]        DMA target, one stage ahead
    _Float16* Bs = smem3 + 2 * 2 * APL;       // [2 buf][2 plane][4 kg][BN][8]   written from registers
    float* Aff = reinterpret_cast<float*>(smem3 + 2 * 2 * APL + 2 * 2 * BPL);  // [2][G3_KAFF] scale | shift (AFF only)
    const GemmArgs& p = q.g;

    const int nblk = tilesM * tilesN * p.nbatch;
    const int lid = xcd_remap(blockIdx.x, nblk);
    const int tile_m = lid % tilesM;
    const int rest = lid / tilesM;
    const int tile_n = rest % tilesN;
    const int batch = rest / tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int M = p.M, kbeg = 0;
    if (p.tri == TRI_ROWS_GE_BATCH) {
        if (m0 + BM <= batch) return;
    } else if (p.tri == TRI_K_GE_BATCH) {
        kbeg = (batch / BKT) * BKT;
    } else if (p.tri == TRI_ROWS_LE_BATCH) {
        const int me = (batch + 1) * p.trimul;
        M = me < M ? me : M;
        if (m0 >= M) return;
    }
    const int K = p.K, N = p.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = (K - kbeg + BKT - 1) / BKT;
    float bscale_k = q.bscale, oscale_k = q.oscale;
    float bbound_k = 0.f;
    if (q.bmax) {  // power of two that puts max|B| in [2^11, 2^12): exact to undo, no overflow, lo parts stay normal
        float mx = __uint_as_float(slot_load(q.bmax + lane));  // 64 shards (one per producer workgroup residue), reduce in the wave
        if (q.bmax2) mx = fmaxf(mx, __uint_as_float(slot_load(q.bmax2 + lane)));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        mx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
        bbound_k = mx;
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 12 - e; }
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        bscale_k = ldexpf(1.0f, e);
        oscale_k = q.oscale * ldexpf(1.0f, -e);  // q.oscale = 1 / ascale in this mode
    }

    if (AFF) {  // per-row affine of B (fused instance norm), pre-multiplied by the power-of-two scale
        const float* bsc = p.bsc + (long)batch * p.sbs;
        const float* bsh = p.bsh + (long)batch * p.sbs;
        for (int k = tid; k < K; k += NT) {
            Aff[k] = bsc[k] * bscale_k;
            Aff[G3_KAFF + k] = bsh[k] * bscale_k;
        }
        __syncthreads();
    }

    if (ADYN && wave >= NW) {
        // ============ loader waves, mirrored roles: A is the fp32 activation (row-major, k contiguous), split on
        // the fly; B is a static operand pre-packed on the host side as fp16 hi/lo planes [K/8][ldn][8] and DMA'd ====
        const int lw = wave - NW;
        const int ltid = tid - NL;
        const long lda = p.lda;
        const float* A = p.A + (long)batch * p.sA;
        const _Float16* Bhi = q.Ahi + (long)batch * p.sB;   // planes of the packed B operand (sB in halves)
        const _Float16* Blo = q.Alo + (long)batch * p.sB;
        const long ldn = p.ldb;                             // columns per k-group row of the packed operand
        constexpr int AE = BM * 4 / NL;                     // (row, 8-k chunk) entries per loader thread (1 or 2)
        constexpr int BCH = 4 * BN / 64;                    // 1 KiB pieces per B plane per stage
        constexpr int BCW = BCH / NW;
        static_assert(BCH % NW == 0, "B pieces must divide evenly over the loader waves");
        const float* arow[AE];
        int aslot[AE];
#pragma unroll
        for (int e = 0; e < AE; ++e) {
            const int idx = ltid + e * NL;
            const int row = idx / 4, kc = idx % 4;
            int m = m0 + row;
            m = m < M ? m : M - 1;
            arow[e] = A + (long)m * lda + 8 * kc;
            aslot[e] = row * 32 + 8 * (kc ^ ((row >> 2) & 3));   // halves; same XOR swizzle the MMA waves read with
        }
        long boff[BCW];
#pragma unroll
        for (int c = 0; c < BCW; ++c) {
            const int cb = lw + c * NW;                     // piece -> (k group, 64 columns)
            const int kgi = cb / (BN / 64), nb = cb % (BN / 64);
            int nn = n0 + nb * 64 + lane;
            nn = nn < N ? nn : N - 1;
            boff[c] = ((long)kgi * ldn + nn) * 8;
        }
        struct ARegs { float4 v[AE][2]; };
        auto issue_b = [&](int k0, int bbuf) {              // 2*BCW DMA pieces
            _Float16* Bb = Bs + bbuf * 2 * BPL;
            const long kgoff = (long)(k0 / 8) * ldn * 8;
#pragma unroll
            for (int c = 0; c < BCW; ++c) {
                const int cb = lw + c * NW;
                glds16(reinterpret_cast<const float*>(Bhi + kgoff + boff[c]), reinterpret_cast<float*>(Bb + cb * 512));
                glds16(reinterpret_cast<const float*>(Blo + kgoff + boff[c]), reinterpret_cast<float*>(Bb + BPL + cb * 512));
            }
        };
        auto issue_a = [&](int k0, ARegs& r) {              // exactly 2*AE loads
#pragma unroll
            for (int e = 0; e < AE; ++e) {
                r.v[e][0] = *reinterpret_cast<const float4*>(arow[e] + k0);
                r.v[e][1] = *reinterpret_cast<const float4*>(arow[e] + k0 + 4);
            }
        };
        const float ascale_dyn = bscale_k;                  // the dynamic operand's power-of-two scale
        auto stash = [&](int abuf, const ARegs& r) {
            _Float16* Ab = As + abuf * 2 * APL;
#pragma unroll
            for (int e = 0; e < AE; ++e) {
                const float xs[8] = {r.v[e][0].x, r.v[e][0].y, r.v[e][0].z, r.v[e][0].w,
                                     r.v[e][1].x, r.v[e][1].y, r.v[e][1].z, r.v[e][1].w};
                half8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = __builtin_amdgcn_fmed3f(xs[j] * ascale_dyn, -65504.f, 65504.f);
                    const _Float16 hh = (_Float16)x;
                    hi[j] = hh;
                    lo[j] = (_Float16)(x - (float)hh);
                }
                *reinterpret_cast<half8*>(Ab + aslot[e]) = hi;
                *reinterpret_cast<half8*>(Ab + APL + aslot[e]) = lo;
            }
        };
        // same pipeline as below with the roles swapped: B DMA one stage ahead, A registers two stages ahead;
        // each stage issues [B DMA of t+1][2*AE A loads of t+2] -> vmcnt(2*AE)
        ARegs r0, r1;
        if (nk > 0) { issue_b(kbeg, 0); issue_a(kbeg, r0); }
        if (nk > 1) issue_a(kbeg + BKT, r1);
        if (nk > 0) stash(0, r0);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            {
                const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
                if (more1) issue_b(kbeg + (kt + 1) * BKT, 1);
                if (more2) issue_a(kbeg + (kt + 2) * BKT, r0);
                if (more1) stash(1, r1);
                if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (kt + 1 >= nk) break;
            {
                const bool more1 = kt + 2 < nk, more2 = kt + 3 < nk;
                if (more1) issue_b(kbeg + (kt + 2) * BKT, 0);
                if (more2) issue_a(kbeg + (kt + 3) * BKT, r1);
                if (more1) stash(0, r0);
                if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        return;
    }
    if (!ADYN && wave >= NW) {
        // =========================== loader waves ===========================
        const int lw = wave - NW;            // loader wave index
        const int ltid = tid - NL;           // 0..255
        const long lda = p.lda, ldb = p.ldb, ldb2 = p.ldb2;
        const int K1 = p.K1;
        const _Float16* Ahi = q.Ahi + (long)batch * p.sA;
        const _Float16* Alo = q.Alo + (long)batch * p.sA;
        const float* B = p.B + (long)batch * p.sB;
        const float* B2 = p.B2 ? p.B2 + (long)batch * p.sB2 : nullptr;
        const long* brow = p.brow;
        const float bscale = bscale_k;
        // A DMA: piece = 16 rows x 64 B; lane -> (row, physical slot), fetches the XOR-swizzled logical slot
        long aoff[ACW];
#pragma unroll
        for (int c = 0; c < ACW; ++c) {
            const int ca = lw + c * NW;
            const int row = ca * 16 + lane / 4;
            const int ls = (lane % 4) ^ ((row >> 2) & 3);
            int m = m0 + row;
            m = m < M ? m : M - 1;
            aoff[c] = (long)m * lda + 8 * ls;
        }
        // B stager: thread -> (k group of 8, NPT consecutive columns)
        const int bkg = ltid / 64;           // 0..3
        const int bnl = (ltid % 64) * NPT;
        int bn = n0 + bnl;
        bn = bn + NPT <= N ? bn : N - NPT;   // clamp (N % NPT == 0): extra columns feed outputs never stored

        struct BRegs { float v[8][NPT]; };
        auto issue_a = [&](int k0, int abuf) {
            _Float16* Ab = As + abuf * 2 * APL;
#pragma unroll
            for (int c = 0; c < ACW; ++c) {
                const int ca = lw + c * NW;
                glds16(reinterpret_cast<const float*>(Ahi + aoff[c] + k0), reinterpret_cast<float*>(Ab + ca * 512));
                glds16(reinterpret_cast<const float*>(Alo + aoff[c] + k0), reinterpret_cast<float*>(Ab + APL + ca * 512));
            }
        };
        // row-offset table (convolution taps): the offsets of the NEXT call's rows are fetched by this call, so that a stage's
        // row loads do not wait for a dependent table load (the calls walk k0 = kbeg, kbeg + BKT, ... in order).  The table loads
        // are issued BEFORE the call's 8 row loads: VMEM returns in order, so reading them back at the next call waits for nothing
        // younger, and the counted vmcnt(8) below still means "everything but this stage's row loads has landed"
        long rowoff[8];
        auto fetch_rowoff = [&](int k0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int k = k0 + 8 * bkg + e;
                k = k < K ? k : K - 1;
                rowoff[e] = brow[k];
            }
        };
        if (brow) fetch_rowoff(kbeg);
        auto issue_b = [&](int k0, BRegs& r) {  // exactly 8 row loads (+ 8 table loads with a row table)
            long cur[8];
            if (brow) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cur[e] = rowoff[e];
                fetch_rowoff(k0 + BKT);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int k = k0 + 8 * bkg + e;
                k = k < K ? k : K - 1;  // rows past K meet zero A columns
                const float* src = brow ? (B + cur[e] + bn) : (K1 >= 0 && k >= K1) ? (B2 + (long)(k - K1) * ldb2 + bn) : (B + (long)k * ldb + bn);
                if (NPT == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    r.v[e][0] = v.x; r.v[e][1] = v.y; r.v[e][NPT - 2] = v.z; r.v[e][NPT - 1] = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    r.v[e][0] = v.x; r.v[e][1] = v.y;
                }
            }
        };
        auto stash = [&](int k0, int bbuf, const BRegs& r) {
            _Float16* Bb = Bs + bbuf * 2 * BPL;
            float sc[8], sh[8];
            if (AFF) {
                int kk = k0 + 8 * bkg;
                kk = kk + 8 <= G3_KAFF ? kk : G3_KAFF - 8;
                const float4 s0 = *reinterpret_cast<const float4*>(Aff + kk), s1 = *reinterpret_cast<const float4*>(Aff + kk + 4);
                const float4 t0 = *reinterpret_cast<const float4*>(Aff + G3_KAFF + kk),
                             t1 = *reinterpret_cast<const float4*>(Aff + G3_KAFF + kk + 4);
                sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
                sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
            }
#pragma unroll
            for (int c = 0; c < NPT; ++c) {
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = r.v[e][c];
                    x = AFF ? fmaf(x, sc[e], sh[e]) : x * bscale;
                    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
                    const _Float16 hh = (_Float16)x;
                    hi[e] = hh;
                    lo[e] = (_Float16)(x - (float)hh);
                }
                *reinterpret_cast<half8*>(Bb + ((long)bkg * BN + bnl + c) * 8) = hi;
                *reinterpret_cast<half8*>(Bb + BPL + ((long)bkg * BN + bnl + c) * 8) = lo;
            }
        };
        // Stage t is computed from As[t & 1] / Bs[t & 1].  A (weights, L2 resident): DMA one stage ahead.  B
        // (activations): row loads two stages ahead into register set t & 1, split into Bs[(t+1) & 1] during stage t.
        // VMEM retires in order and each stage issues [A DMA of t+1][8 B loads of t+2]: "A of t+1 landed, B of t+2 may
        // still fly" is exactly vmcnt(8).
        BRegs r0, r1;
        if (nk > 0) { issue_a(kbeg, 0); issue_b(kbeg, r0); }
        if (nk > 1) issue_b(kbeg + BKT, r1);
        if (nk > 0) stash(kbeg, 0, r0);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            {
                const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
                if (more1) issue_a(kbeg + (kt + 1) * BKT, 1);
                if (more2) issue_b(kbeg + (kt + 2) * BKT, r0);
                if (more1) stash(kbeg + (kt + 1) * BKT, 1, r1);
                if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (kt + 1 >= nk) break;
            {
                const bool more1 = kt + 2 < nk, more2 = kt + 3 < nk;
                if (more1) issue_a(kbeg + (kt + 2) * BKT, 0);
                if (more2) issue_b(kbeg + (kt + 3) * BKT, r1);
                if (more1) stash(kbeg + (kt + 2) * BKT, 0, r0);
                if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        return;
    }

    // =========================== MMA waves ===========================
    const int i = lane & 31, g = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int arow0 = wm * 64 + i, arow1 = arow0 + 32;
    const int key0 = (arow0 >> 2) & 3, key1 = (arow1 >> 2) & 3;
    const int bcol0 = wn * 64 + i, bcol1 = bcol0 + 32;

    __syncthreads();  // stage 0 is in LDS
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const _Float16* Ah = As + buf * 2 * APL;
        const _Float16* Al = Ah + APL;
        const _Float16* Bh = Bs + buf * 2 * BPL;
        const _Float16* Bl = Bh + BPL;
        half8 ah0[2], ah1[2], al0[2], al1[2], bh0[2], bh1[2], bl0[2], bl1[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {  // all fragments of the stage up front: 16 ds_read_b128
            const int ls = 2 * c + g;  // lane group g owns k = 16c + 8g .. +7
            ah0[c] = *reinterpret_cast<const half8*>(Ah + arow0 * 32 + 8 * (ls ^ key0));
            ah1[c] = *reinterpret_cast<const half8*>(Ah + arow1 * 32 + 8 * (ls ^ key1));
            al0[c] = *reinterpret_cast<const half8*>(Al + arow0 * 32 + 8 * (ls ^ key0));
            al1[c] = *reinterpret_cast<const half8*>(Al + arow1 * 32 + 8 * (ls ^ key1));
            bh0[c] = *reinterpret_cast<const half8*>(Bh + ((long)ls * BN + bcol0) * 8);
            bh1[c] = *reinterpret_cast<const half8*>(Bh + ((long)ls * BN + bcol1) * 8);
            bl0[c] = *reinterpret_cast<const half8*>(Bl + ((long)ls * BN + bcol0) * 8);
            bl1[c] = *reinterpret_cast<const half8*>(Bl + ((long)ls * BN + bcol1) * 8);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {  // small cross terms first, the hi.hi term last
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0[c], bh0[c], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0[c], bh1[c], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1[c], bh0[c], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1[c], bh1[c], acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[c], bl0[c], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[c], bl1[c], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[c], bl0[c], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[c], bl1[c], acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[c], bh0[c], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[c], bh1[c], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[c], bh0[c], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[c], bh1[c], acc[1][1], 0, 0, 0);
        }
        __syncthreads();  // fragments are in registers (lgkmcnt(0) precedes the barrier): the buffer may be refilled
    }

    // epilogue
    const float osc_acc = oscale_k;
    float vmax = 0.f;
    const float* B = ADYN ? p.A : p.B + (long)batch * p.sB;
    float* C = p.C + (long)batch * p.sC;
    const float* R = RES ? p.R + (long)batch * p.sR : nullptr;
    const float* rsc = p.rsc ? p.rsc + (long)batch * p.srs : nullptr;
    const float* rsh = p.rsc ? p.rsh + (long)batch * p.srs : nullptr;
    const float* bias = p.bias ? p.bias + (long)batch * p.sbias : nullptr;
    const float* osc = p.osc;
    const float* osh = p.osh;
    const float* dummy = B;
    const long ldc = p.ldc, ldr = p.ldr;
    const int actk = p.act;
    const bool vec_epi = (N % 4 == 0) && (ldc % 4 == 0) && (p.sC % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                         (!RES || ((ldr % 4 == 0) && (p.sR % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.R) & 15) == 0)));
    if (vec_epi) {
        // through LDS (as the v4 engine): park the 64x64 tile in this wave's 16 KiB of the idle operand buffers, read it
        // back row-contiguous: residual loads and stores are 16 B per lane.  The loader waves are done with LDS (their
        // last barrier is the one the MMA waves just passed).
        float cscale = 1.f;
        if (q.Chi) {
            const float cbound = q.cw * bbound_k;
            cscale = ldexpf(1.0f, pow2_exponent_for(cbound));
            if (tid == 0) atomicMax(q.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
        }
        float* Ts = reinterpret_cast<float*>(smem3) + wave * 4096;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(tm * 32 + acc_row(r, g)) * 64 + tn * 32 + i] = acc[tm][tn][r];
        const int c4 = (lane & 15) * 4;
        const int colb = n0 + wn * 64 + c4;
        const bool cok = colb < N;
        const int colc = cok ? colb : 0;
        _Float16* Ph = q.Chi ? q.Chi + (long)batch * p.sC : nullptr;
        _Float16* Pl = q.Chi ? q.Clo + (long)batch * p.sC : nullptr;
        // four rows at a time: this kernel is capped at 128 VGPRs (8 waves per workgroup, 2 workgroups per CU)
        const float capv = p.cap;
        dispatch_act(actk, [&](auto at) {
        constexpr int AC = decltype(at)::value;
#pragma unroll 1
        for (int jb = 0; jb < 16; jb += 4) {
            float4 tv[4], rv[4];
            float bvv[4], rsv[4], rtv[4], osv[4], otv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int rl = (lane >> 4) + 4 * (jb + jj);
                const int row = m0 + wm * 64 + rl;
                const int rr = row < M ? row : 0;
                const float t0 = (bias ? bias : dummy)[bias ? rr : 0];
                const float t1 = (rsc ? rsc : dummy)[rsc ? rr : 0];
                const float t2 = (rsc ? rsh : dummy)[rsc ? rr : 0];
                const float t3 = (osc ? osc : dummy)[osc ? rr : 0];
                const float t4 = (osc ? osh : dummy)[osc ? rr : 0];
                bvv[jj] = bias ? t0 : 0.f;
                rsv[jj] = rsc ? t1 : 1.f;
                rtv[jj] = rsc ? t2 : 0.f;
                osv[jj] = osc ? t3 : 1.f;
                otv[jj] = osc ? t4 : 0.f;
                if (RES) rv[jj] = *reinterpret_cast<const float4*>(R + (long)rr * ldr + colc);
                tv[jj] = *reinterpret_cast<const float4*>(Ts + rl * 64 + c4);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int row = m0 + wm * 64 + (lane >> 4) + 4 * (jb + jj);
                float o[4] = {tv[jj].x, tv[jj].y, tv[jj].z, tv[jj].w};
                const float r4[4] = {RES ? rv[jj].x : 0.f, RES ? rv[jj].y : 0.f, RES ? rv[jj].z : 0.f, RES ? rv[jj].w : 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(o[e], osc_acc, bvv[jj]);
                    if (RES) v += fmaf(r4[e], rsv[jj], rtv[jj]);
                    v = fminf(act_const<AC>(v, actk), capv);
                    o[e] = fmaf(v, osv[jj], otv[jj]);
                }
                if (row < M && cok) {
                    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                    if (Ph) {
                        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                        half4 hh, ll;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = o[e] * cscale;
                            const _Float16 a = (_Float16)x;
                            hh[e] = a;
                            ll[e] = (_Float16)(x - (float)a);
                        }
                        *reinterpret_cast<half4*>(Ph + (long)row * ldc + colb) = hh;
                        *reinterpret_cast<half4*>(Pl + (long)row * ldc + colb) = ll;
                    } else {
                        *reinterpret_cast<float4*>(C + (long)row * ldc + colb) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
        });
        if (q.omax) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
            if (lane == 0) atomicMax(q.omax + (blockIdx.x & 63), __float_as_uint(vmax));
        }
        return;
    }
    const int col0 = n0 + wn * 64 + i, col1 = col0 + 32;
    const bool c0ok = col0 < N, c1ok = col1 < N;
    const int cc0 = c0ok ? col0 : 0, cc1 = c1ok ? col1 : 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int rbase = m0 + wm * 64 + tm * 32 + 8 * qd + 4 * g;
            float bv[4], rs[4], rt[4], os[4], ot[4], rv0[4], rv1[4];
            int rr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rr[e] = (rbase + e < M) ? rbase + e : 0;
                const float t0 = (bias ? bias : dummy)[bias ? rr[e] : 0];
                const float t1 = (rsc ? rsc : dummy)[rsc ? rr[e] : 0];
                const float t2 = (rsc ? rsh : dummy)[rsc ? rr[e] : 0];
                const float t3 = (osc ? osc : dummy)[osc ? rr[e] : 0];
                const float t4 = (osc ? osh : dummy)[osc ? rr[e] : 0];
                bv[e] = bias ? t0 : 0.f;
                rs[e] = rsc ? t1 : 1.f;
                rt[e] = rsc ? t2 : 0.f;
                os[e] = osc ? t3 : 1.f;
                ot[e] = osc ? t4 : 0.f;
                if (RES) {
                    rv0[e] = R[(long)rr[e] * ldr + cc0];
                    rv1[e] = R[(long)rr[e] * ldr + cc1];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool rok = rbase + e < M;
                float v0 = fmaf(acc[tm][0][4 * qd + e], osc_acc, bv[e]);
                float v1 = fmaf(acc[tm][1][4 * qd + e], osc_acc, bv[e]);
                if (RES) {
                    v0 += fmaf(rv0[e], rs[e], rt[e]);
                    v1 += fmaf(rv1[e], rs[e], rt[e]);
                }
                v0 = fminf(act_apply(v0, actk), p.cap);
                v1 = fminf(act_apply(v1, actk), p.cap);
                v0 = fmaf(v0, os[e], ot[e]);
                v1 = fmaf(v1, os[e], ot[e]);
                float* crow = C + (long)rr[e] * ldc;
                if (rok && c0ok) { crow[col0] = v0; vmax = fmaxf(vmax, fabsf(v0)); }
                if (rok && c1ok) { crow[col1] = v1; vmax = fmaxf(vmax, fabsf(v1)); }
            }
        }
    }
    if (q.omax) {  // one atomic per wave: bit patterns of non-negative floats order like the floats
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) atomicMax(q.omax + (blockIdx.x & 63), __float_as_uint(vmax));
    }
}

// fp32 (rows x cols, pitch lds) -> two fp16 planes hi/lo (pitch ldd halves, columns >= cols zero), x scaled by `scale`
__global__ void split_f16_kernel(const float* __restrict__ src, long lds_, _Float16* __restrict__ hi,
                                 _Float16* __restrict__ lo, long ldd, long rows, int cols, float scale) {
    const long total = rows * ldd;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long r = t / ldd;
        const int c = (int)(t % ldd);
        float x = 0.f;
        if (c < cols) x = __builtin_amdgcn_fmed3f(src[r * lds_ + c] * scale, -65504.f, 65504.f);
        const _Float16 h = (_Float16)x;
        hi[t] = h;
        lo[t] = (_Float16)(x - (float)h);
    }
}

// fp32 (rows x cols, pitch lds) -> fp16 hi/lo planes in the v4 engine's A-tile order: [rows/16][ldd/32][16][4 slots][8],
// slot s of row r holds the logical 8-k chunk s ^ ((r >> 2) & 3) (the LDS image the fragment reads expect); ldd =
// cols rounded up to 32, rows rounded up to 16, padding zero
__global__ void split_f16_tiled_kernel(const float* __restrict__ src, long lds_, _Float16* __restrict__ hi,
                                       _Float16* __restrict__ lo, long ldd, long rows, int cols, float scale) {
    const long rows16 = (rows + 15) / 16 * 16;
    const long total = rows16 * ldd;
    const long nst = ldd / 32;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long e = t % 8, ps = (t / 8) % 4, r = (t / 32) % 16, st = (t / 512) % nst, rb = t / (512 * nst);
        const long row = rb * 16 + r;
        const long ls = ps ^ ((row >> 2) & 3);
        const long c = st * 32 + ls * 8 + e;
        float x = 0.f;
        if (row < rows && c < cols) x = __builtin_amdgcn_fmed3f(src[row * lds_ + c] * scale, -65504.f, 65504.f);
        const _Float16 h = (_Float16)x;
        hi[t] = h;
        lo[t] = (_Float16)(x - (float)h);
    }
}
hipError_t launch_split_f16_tiled(const float* src, long lds_, void* hi, void* lo, long ldd, long rows, int cols,
                                  float scale, hipStream_t s) {
    const long total = (rows + 15) / 16 * 16 * ldd;
    if (total <= 0 || ldd % 32 != 0) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
    hipLaunchKernelGGL(split_f16_tiled_kernel, dim3(grid), dim3(256), 0, s, src, lds_, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), ldd, rows, cols, scale);
    return hipGetLastError();
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// Historical engine switches (ACE_FORCE_V1, ACE_G4_TILE, ACE_NO_PK, ACE_NO_PK_SHT) only exist in measurement builds
// (-DACE_MEASUREMENT_SWITCHES, tools/mkvar.sh): the shipped library has one routing per shape.
#ifdef ACE_MEASUREMENT_SWITCHES
static const bool g_force_v1 = (getenv("ACE_FORCE_V1") != nullptr);
#else
static constexpr bool g_force_v1 = false;
#endif

template <int WM, int WN, bool VEC>
static hipError_t launch_gemm_vec(const GemmArgs& a, hipStream_t s, int tilesM, int tilesN, dim3 grid, dim3 block) {
    const bool aff = a.bsc != nullptr, res = a.R != nullptr;
    if (aff && res)
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, VEC, true, true>), grid, block, 0, s, a, tilesM, tilesN);
    else if (aff)
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, VEC, true, false>), grid, block, 0, s, a, tilesM, tilesN);
    else if (res)
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, VEC, false, true>), grid, block, 0, s, a, tilesM, tilesN);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, VEC, false, false>), grid, block, 0, s, a, tilesM, tilesN);
    return hipGetLastError();
}

template <int WM, int WN, int BKT>
static hipError_t launch_gemm2_cfg(const GemmArgs& a, hipStream_t s, int tilesM, int tilesN, dim3 grid, dim3 block) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr size_t lds = (size_t)2 * (BM * BKT + BKT * BN) * sizeof(float);
    static bool configured[2] = {false, false};
    const bool res = a.R != nullptr;
    if (!configured[res]) {
        hipError_t e = res ? hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_f32_kernel<WM, WN, BKT, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_f32_kernel<WM, WN, BKT, false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[res] = true;
    }
    if (res)
        hipLaunchKernelGGL((gemm2_f32_kernel<WM, WN, BKT, true>), grid, block, lds, s, a, tilesM, tilesN);
    else
        hipLaunchKernelGGL((gemm2_f32_kernel<WM, WN, BKT, false>), grid, block, lds, s, a, tilesM, tilesN);
    return hipGetLastError();
}

#ifndef ACE_GEMM2_BK
#define ACE_GEMM2_BK 32
#endif

template <int WM, int WN>
static hipError_t launch_gemm_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    constexpr int BK2 = ACE_GEMM2_BK;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    const long nblk = (long)tilesM * tilesN * a.nbatch;
    if (nblk <= 0) return hipSuccess;
    bool vec = al16(a.A) && (a.lda % 4 == 0) && (a.sA % 4 == 0) && (a.K % 4 == 0) && al16(a.B) && (a.ldb % 4 == 0) &&
               (a.sB % 4 == 0) && (a.N % 4 == 0);
    if (a.B2) vec = vec && al16(a.B2) && (a.ldb2 % 4 == 0) && (a.sB2 % 4 == 0);
    dim3 grid((unsigned)nblk), block(NT);
    // direct-to-LDS engine: needs zero-padded (or exact) A rows up to the stage depth and no per-k affine
    const int kround = ((a.K + BK2 - 1) / BK2) * BK2;
    const bool v2 = !g_force_v1 && al16(a.A) && (a.lda % 4 == 0) && (a.sA % 4 == 0) && al16(a.B) && (a.ldb % 4 == 0) &&
                    (a.sB % 4 == 0) && (a.N % 4 == 0) && (a.N >= 4) && (a.bsc == nullptr) && (a.a_kpad >= kround) &&
                    (!a.B2 || (al16(a.B2) && (a.ldb2 % 4 == 0) && (a.sB2 % 4 == 0))) && a.M >= 1 && a.K >= 1;
    if (v2) return launch_gemm2_cfg<WM, WN, BK2>(a, s, tilesM, tilesN, grid, block);
    if (vec) return launch_gemm_vec<WM, WN, true>(a, s, tilesM, tilesN, grid, block);
    return launch_gemm_vec<WM, WN, false>(a, s, tilesM, tilesN, grid, block);
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.brow) return hipErrorInvalidValue;   // row tables: f16x3 engine only
    // 128x128 when the row count tiles well by 128, else 64x256 (M = 180/181-row spectral problems, M = 50)
    const int waste128 = ((a.M + 127) / 128) * 128 - a.M;
    const int waste64 = ((a.M + 63) / 64) * 64 - a.M;
    if (a.M >= 128 && waste128 <= waste64) return launch_gemm_cfg<2, 2>(a, s);
    return launch_gemm_cfg<1, 4>(a, s);
}


#ifndef ACE_G3_W128_PCT
#define ACE_G3_W128_PCT 0   // f16x3 engine: take the 128 x 128 tile when it wastes at most this share of its rows (else 64 x 256)
#endif
template <int WM, int WN>
static hipError_t launch_gemm3_cfg(const Gemm3Args& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 128 * WM * WN;  // MMA waves + loader waves
    const int tilesM = (a.g.M + BM - 1) / BM, tilesN = (a.g.N + BN - 1) / BN;
    const long nblk = (long)tilesM * tilesN * a.g.nbatch;
    if (nblk <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * 2 * (BM * 32 + 32 * BN) * sizeof(_Float16) + (a.g.bsc ? 2 * G3_KAFF * sizeof(float) : 0);
    const bool aff = a.g.bsc != nullptr, res = a.g.R != nullptr;
    const void* fn = aff ? (res ? (const void*)gemm3_f16x3_kernel<WM, WN, true, true> : (const void*)gemm3_f16x3_kernel<WM, WN, true, false>)
                         : (res ? (const void*)gemm3_f16x3_kernel<WM, WN, false, true> : (const void*)gemm3_f16x3_kernel<WM, WN, false, false>);
    static bool configured[4] = {false, false, false, false};
    const int ci = (aff ? 2 : 0) + (res ? 1 : 0);
    if (!configured[ci]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[ci] = true;
    }
    dim3 grid((unsigned)nblk), block(NT);
    if (aff && res) hipLaunchKernelGGL((gemm3_f16x3_kernel<WM, WN, true, true>), grid, block, lds, s, a, tilesM, tilesN);
    else if (aff) hipLaunchKernelGGL((gemm3_f16x3_kernel<WM, WN, true, false>), grid, block, lds, s, a, tilesM, tilesN);
    else if (res) hipLaunchKernelGGL((gemm3_f16x3_kernel<WM, WN, false, true>), grid, block, lds, s, a, tilesM, tilesN);
    else hipLaunchKernelGGL((gemm3_f16x3_kernel<WM, WN, false, false>), grid, block, lds, s, a, tilesM, tilesN);
    return hipGetLastError();
}

bool gemm_f16x3_eligible(const GemmArgs& a) {
    const int npt = 4;  // strictest alignment of the two tile shapes
    bool ok = al16(a.B) && (a.ldb % 4 == 0) && (a.sB % 4 == 0) && (a.N % npt == 0) && (a.N >= npt) && (a.lda % 8 == 0) &&
              (a.sA % 8 == 0) && (a.a_kpad >= ((a.K + 31) / 32) * 32) && a.M >= 1 && a.K >= 1;
    if (a.B2) ok = ok && al16(a.B2) && (a.ldb2 % 4 == 0) && (a.sB2 % 4 == 0);
    if (a.bsc) ok = ok && (((a.K + 31) / 32) * 32 <= G3_KAFF);
    return ok;
}

hipError_t launch_gemm_f16x3(const GemmArgs& g, const void* Ahi, const void* Alo, float ascale, float bscale,
                             hipStream_t s, const unsigned* bmax, unsigned* omax, const unsigned* bmax2) {
    Gemm3Args a;
    a.bmax2 = bmax2;
    a.g = g;
    a.Ahi = static_cast<const _Float16*>(Ahi);
    a.Alo = static_cast<const _Float16*>(Alo);
    a.bscale = bscale;
    a.oscale = bmax ? 1.0f / ascale : 1.0f / (ascale * bscale);
    a.bmax = bmax;
    a.omax = omax;
    const int waste128 = ((g.M + 127) / 128) * 128 - g.M;
    const int waste64 = ((g.M + 63) / 64) * 64 - g.M;
    if (g.M >= 128 && (waste128 <= waste64 || waste128 * 100 <= ACE_G3_W128_PCT * (g.M + waste128))) return launch_gemm3_cfg<2, 2>(a, s);
    return launch_gemm3_cfg<1, 4>(a, s);
}

// as launch_gemm_f16x3, but C is written as fp16 hi/lo planes in C's own layout (see Gemm3Args::Chi)
hipError_t launch_gemm_f16x3_planes(const GemmArgs& g, const void* Ahi, const void* Alo, float ascale, const unsigned* bmax,
                                    void* Chi, void* Clo, float cw, unsigned* cslot, hipStream_t s) {
    if (!bmax || !Chi || !Clo || !cslot || g.N % 4 != 0 || g.ldc % 4 != 0 || g.sC % 4 != 0 || g.R || !al16(Chi) || !al16(Clo))
        return hipErrorInvalidValue;
    Gemm3Args a;
    a.g = g;
    a.g.C = reinterpret_cast<float*>(Chi);   // only its alignment is looked at
    a.Ahi = static_cast<const _Float16*>(Ahi);
    a.Alo = static_cast<const _Float16*>(Alo);
    a.bscale = 1.f;
    a.oscale = 1.0f / ascale;
    a.bmax = bmax;
    a.Chi = static_cast<_Float16*>(Chi); a.Clo = static_cast<_Float16*>(Clo); a.cw = cw; a.cslot = cslot;
    const long waste128 = (long)((g.M + 127) / 128) * 128, waste64 = (long)((g.M + 63) / 64) * 64;
    if (g.M >= 128 && waste128 <= waste64) return launch_gemm3_cfg<2, 2>(a, s);
    return launch_gemm3_cfg<1, 4>(a, s);
}

hipError_t launch_split_f16(const float* src, long lds_, void* hi, void* lo, long ldd, long rows, int cols, float scale,
                            hipStream_t s) {
    const long total = rows * ldd;
    long gsz = (total + 255) / 256;
    if (gsz > 16384) gsz = 16384;
    if (gsz < 1) gsz = 1;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)gsz), dim3(256), 0, s, src, lds_, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), ldd, rows, cols, scale);
    return hipGetLastError();
}

// NoiseConditionedSFNO filter weight (G, L, C/G [out], C/G [in], 2) (conditional_sfno/s2convolutions.py:232-239,
// einsum "bgixy,gxoi->bgoxy") -> the dense layout (Cin, Cout, L, 2) the rest of the library works with; entries that
// couple different groups are zero
__global__ void csfno_weight_to_dense_kernel(const float* __restrict__ w, float* __restrict__ dense, int C, int G, int L) {
    const long total = (long)C * C * L;
    const int Cg = C / G;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int l = t % L;
        const int o = (t / L) % C;
        const int i = t / ((long)L * C);
        float2 v = make_float2(0.f, 0.f);
        if (i / Cg == o / Cg) {
            const int g = i / Cg;
            v = *reinterpret_cast<const float2*>(w + ((((long)g * L + l) * Cg + (o % Cg)) * Cg + (i % Cg)) * 2);
        }
        *reinterpret_cast<float2*>(dense + t * 2) = v;
    }
}
hipError_t launch_csfno_weight_to_dense(const float* w, float* dense, int C, int G, int L, hipStream_t s) {
    if (G < 1 || C % G != 0) return hipErrorInvalidValue;
    const long total = (long)C * C * L;
    long gsz = (total + 255) / 256;
    if (gsz > 32768) gsz = 32768;
    hipLaunchKernelGGL(csfno_weight_to_dense_kernel, dim3((unsigned)gsz), dim3(256), 0, s, w, dense, C, G, L);
    return hipGetLastError();
}

// dhconv weight (Cin, Cout, L, 2) -> per-l real 2Cin x 2Cout operand in the f16x3 engine's k-packed form:
//   planes hi/lo [l][K/8][ldn][8] halves, K = 2 Cin (rows (ri, i)), ldn = 2 Cout (columns (r', o)), values scaled by `scale`
//   [ out_re | out_im ] = [ x_re | x_im ] * [[ w_re, w_im ], [ -w_im, w_re ]]
__global__ void pack_dhconv_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                       int Cin, int Cout, int L, float scale) {
    const int K = 2 * Cin, N = 2 * Cout;
    const long total = (long)L * K * N;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int e = t % 8;          // k within the group (fastest in memory)
        long q = t / 8;
        const int n = q % N;
        q /= N;
        const int kg = q % (K / 8);
        const int l = q / (K / 8);
        const int k = kg * 8 + e;
        const int ri = k / Cin, i = k % Cin, rp = n / Cout, o = n % Cout;
        const float2 wv = *reinterpret_cast<const float2*>(w + (((long)i * Cout + o) * L + l) * 2);
        float v = (ri == rp) ? wv.x : (ri == 0 ? wv.y : -wv.y);
        v = __builtin_amdgcn_fmed3f(v * scale, -65504.f, 65504.f);
        const _Float16 h = (_Float16)v;
        hi[t] = h;
        lo[t] = (_Float16)(v - (float)h);
    }
}
hipError_t launch_pack_dhconv_f16(const float* w, void* hi, void* lo, int Cin, int Cout, int L, float scale, hipStream_t s) {
    const long total = (long)L * 2 * Cin * 2 * Cout;
    long gsz = (total + 255) / 256;
    if (gsz > 32768) gsz = 32768;
    hipLaunchKernelGGL(pack_dhconv_f16_kernel, dim3((unsigned)gsz), dim3(256), 0, s, w, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), Cin, Cout, L, scale);
    return hipGetLastError();
}

// the same weight, compact: two Cin x Cout P-format blocks per l (re, then im); see Gemm4Args::cplx
__global__ void pack_dhconv_f16c_kernel(const float* __restrict__ w, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                        int Cin, int Cout, int L, float scale) {
    const long total = (long)L * 2 * Cin * Cout;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int e = t % 8;
        long q = t / 8;
        const int o = q % Cout;
        q /= Cout;
        const int kg = q % (Cin / 8);
        q /= (Cin / 8);
        const int blk = q % 2;
        const int l = q / 2;
        const int i = kg * 8 + e;
        const float2 wv = *reinterpret_cast<const float2*>(w + (((long)i * Cout + o) * L + l) * 2);
        const float v = __builtin_amdgcn_fmed3f((blk == 0 ? wv.x : wv.y) * scale, -65504.f, 65504.f);
        const _Float16 h = (_Float16)v;
        hi[t] = h;
        lo[t] = (_Float16)(v - (float)h);
    }
}
hipError_t launch_pack_dhconv_f16c(const float* w, void* hi, void* lo, int Cin, int Cout, int L, float scale, hipStream_t s) {
    if (Cin % 8 != 0) return hipErrorInvalidValue;
    const long total = (long)L * 2 * Cin * Cout;
    long gsz = (total + 255) / 256;
    if (gsz > 32768) gsz = 32768;
    hipLaunchKernelGGL(pack_dhconv_f16c_kernel, dim3((unsigned)gsz), dim3(256), 0, s, w, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), Cin, Cout, L, scale);
    return hipGetLastError();
}

// the grouped csfno filter AS THE REFERENCE STORES IT, (G, L, C/G [out], C/G [in], 2) (conditional_sfno/s2convolutions.py:119-135, 229-240),
// compact per l like the kernel above but with the diagonal blocks only: [l][Wr | Wi][kg < (C/G)/8][o < C][8], row kg * 8 + e = input
// channel RELATIVE to the group of output column o.  1 / G of the dense form's memory and upload; csrc/dhconv_strip.hip reads it.
__global__ void pack_dhconv_f16g_kernel(const float* __restrict__ w, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                        int C, int G, int L, float scale) {
    const int cg = C / G;
    const long total = (long)L * 2 * cg * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int e = t % 8;
        long q = t / 8;
        const int o = q % C;
        q /= C;
        const int kg = q % (cg / 8);
        q /= (cg / 8);
        const int blk = q % 2;
        const int l = q / 2;
        const int grp = o / cg, orel = o % cg, irel = kg * 8 + e;
        const float2 wv = *reinterpret_cast<const float2*>(w + ((((long)grp * L + l) * cg + orel) * cg + irel) * 2);
        const float v = __builtin_amdgcn_fmed3f((blk == 0 ? wv.x : wv.y) * scale, -65504.f, 65504.f);
        const _Float16 h = (_Float16)v;
        hi[t] = h;
        lo[t] = (_Float16)(v - (float)h);
    }
}
hipError_t launch_pack_dhconv_f16g(const float* w, void* hi, void* lo, int C, int G, int L, float scale, hipStream_t s) {
    if (G < 1 || C % G != 0 || (C / G) % 8 != 0) return hipErrorInvalidValue;
    const long total = (long)L * 2 * (C / G) * C;
    long gsz = (total + 255) / 256;
    if (gsz > 32768) gsz = 32768;
    hipLaunchKernelGGL(pack_dhconv_f16g_kernel, dim3((unsigned)gsz), dim3(256), 0, s, w, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), C, G, L, scale);
    return hipGetLastError();
}

// f16x3 with the roles mirrored: A = fp32 activations (split on the fly, dynamic scale from amax), B = packed static
// operand (planes Bhi/Blo, [K/8][ldn][8] per batch with batch stride sB_halves).  Requirements: K % 32 == 0, lda % 4 == 0,
// A 16B aligned.  g.B / g.ldb are ignored (ldn and the plane pointers describe B).
template <int WM, int WN>
static hipError_t launch_gemm3_adyn_cfg(const Gemm3Args& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 128 * WM * WN;
    const int tilesM = (a.g.M + BM - 1) / BM, tilesN = (a.g.N + BN - 1) / BN;
    const long nblk = (long)tilesM * tilesN * a.g.nbatch;
    if (nblk <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * 2 * (BM * 32 + 32 * BN) * sizeof(_Float16);
    const bool res = a.g.R != nullptr;
    static bool configured[2] = {false, false};
    if (!configured[res]) {
        const void* fn = res ? (const void*)gemm3_f16x3_kernel<WM, WN, false, true, true>
                             : (const void*)gemm3_f16x3_kernel<WM, WN, false, false, true>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[res] = true;
    }
    dim3 grid((unsigned)nblk), block(NT);
    if (res) hipLaunchKernelGGL((gemm3_f16x3_kernel<WM, WN, false, true, true>), grid, block, lds, s, a, tilesM, tilesN);
    else hipLaunchKernelGGL((gemm3_f16x3_kernel<WM, WN, false, false, true>), grid, block, lds, s, a, tilesM, tilesN);
    return hipGetLastError();
}
hipError_t launch_gemm_f16x3_adyn(const GemmArgs& g, const void* Bhi, const void* Blo, long ldn, long sB_halves,
                                  float bscale_static, const unsigned* amax, unsigned* omax, hipStream_t s) {
    Gemm3Args a;
    a.g = g;
    a.g.ldb = ldn;
    a.g.sB = sB_halves;
    a.Ahi = static_cast<const _Float16*>(Bhi);
    a.Alo = static_cast<const _Float16*>(Blo);
    a.oscale = 1.0f / bscale_static;   // dynamic part is derived in-kernel from amax
    a.bmax = amax;
    a.omax = omax;
    const int waste128 = ((g.M + 127) / 128) * 128 - g.M;
    const int waste64 = ((g.M + 63) / 64) * 64 - g.M;
    if (g.M >= 128 && waste128 <= waste64) return launch_gemm3_adyn_cfg<2, 2>(a, s);
    return launch_gemm3_adyn_cfg<1, 4>(a, s);
}

// ---------------------------------------------------------------------------------------------
// batched GEMM, compensated-fp16, both operands pre-split ("v4").  Same arithmetic as v3 (three fp16 MFMA products
// per tile, fp32 accumulate), but neither operand is converted here:
//   * A: fp16 hi/lo planes, row-major [M][lda] (k contiguous, rows zero-padded to the stage depth) - weights/tables
//     split once at upload, or activations written this way by their producer;
//   * B: fp16 hi/lo planes in the k-packed "P format" [K/8][ldn][8 halves]: entry (kg, n) holds k = 8kg..8kg+7 of
//     column n, i.e. exactly the MFMA B fragment of one lane.  Activations are written in this format by their
//     producer's epilogue (PK output below) or by pack_pformat_kernel; 4 bytes per element, like fp32.
// Both tiles come in by LDS-DMA.  4 waves (no loader waves), up to 256 VGPRs: the fragments of stage t+1 are read
// from LDS into a second register set while the MFMAs of stage t run; one barrier per stage; the DMA of stage t+2
// is issued right after that barrier.  Epilogue as v3, plus the optional P-format output over the rows of C.
// ---------------------------------------------------------------------------------------------

DEVINL float slot_reduce(const unsigned* slot, int lane) {
    float mx = __uint_as_float(slot_load(slot + lane));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
}

#ifdef ACE_X_TRACE
__device__ unsigned long long g4_trace[64];
extern "C" int ace_debug_g4_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g4_trace), sizeof(g4_trace)); }
#define G4T(ev) do { if (blockIdx.x == 700 && threadIdx.x == 0) g4_trace[ev] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G4T(ev) do { } while (0)
#endif
#ifndef ACE_G4_PFD
#define ACE_G4_PFD 0   // L2 prefetch distance of the v4 engine (0 = off: measured neutral-to-negative, r01 profiles)
#endif
#ifndef ACE_G4_RTOUCH
#define ACE_G4_RTOUCH 0
#endif
#ifndef ACE_G4_PIN
#define ACE_G4_PIN 1
#endif
#ifndef ACE_G4_REGEPI
#define ACE_G4_REGEPI 0   // P-format-only epilogue from registers (v_permlane32_swap): written, lane mapping probed, NOT yet
#endif                    // validated on the GPU (round 2: tools/ab.sh with -DACE_G4_REGEPI=1)
struct Frags4 { half8 ah[2], al[2], bh[2], bl[2]; };  // one 16-deep k half: [tile]

template <int WM, int WN, bool RES, bool PK, bool IMPL>
DEVINL void gemm4_body(const Gemm4Args& q, const int tilesM, const int tilesN) {
    static_assert(!IMPL || !RES, "implicit-GEMM form: no residual");
    constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
    static_assert(NW == 4, "piece distribution assumes 4 waves");
    constexpr int BKT = 32;
    constexpr int APL = BM * BKT;             // halves per A plane per buffer
    constexpr int BPL = BKT * BN;             // halves per B plane per buffer
    constexpr int ACW = BM * BKT * 2 / 1024 / NW;   // 1 KiB pieces per wave per A plane (2 or 1)
    constexpr int BCW = 4 * BN / 64 / NW;           // 1 KiB pieces per wave per B plane (2 or 4)
    constexpr int PCS = 2 * (ACW + BCW);            // DMA pieces per wave per stage
    constexpr int PFD = ACE_G4_PFD;                 // prefetch distance (stages beyond the DMA'd one)
    extern __shared__ __attribute__((aligned(16))) _Float16 smem4[];
    _Float16* As = smem4;                     // [2 buf][2 plane][BM][32]
    _Float16* Bs = smem4 + 2 * 2 * APL;       // [2 buf][2 plane][4 kg][BN][8]
    _Float16* Scratch = Bs + 2 * 2 * BPL;     // [NW][128]: landing zone of the prefetch touches (never read)

    const int nblk = tilesM * tilesN * q.nbatch;
    const int lid = xcd_remap(blockIdx.x, nblk);
    const int tile_m = lid % tilesM;
    const int rest = lid / tilesM;
    const int tile_n = rest % tilesN;
    const int batch = rest / tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int M = q.M, kbeg = 0;
    if (q.tri == TRI_ROWS_GE_BATCH) {
        if (m0 + BM <= batch) return;
    } else if (q.tri == TRI_K_GE_BATCH) {
        kbeg = (batch / BKT) * BKT;
    } else if (q.tri == TRI_ROWS_LE_BATCH) {
        const int me = (batch + 1) * q.trimul;
        M = me < M ? me : M;
        if (m0 >= M) return;
    }
    G4T(0);
    const int K = q.K, N = q.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = (K - kbeg + BKT - 1) / BKT;
    const int nkg = (K + 7) / 8;              // valid k groups of the packed operand

    // dynamic-range slots: loaded now, reduced after the first DMAs are in flight
    const unsigned raw_a = q.amax ? slot_load(q.amax + lane) : 0u;
    const unsigned raw_b = q.bmax ? slot_load(q.bmax + lane) : 0u;
    const unsigned raw_c = q.cinb ? slot_load(q.cinb + lane) : 0u;
    const unsigned raw_r = q.rmax ? slot_load(q.rmax + lane) : 0u;

    // ---- DMA descriptors of this wave's pieces (fixed over the k loop)
    const _Float16* Ahi = q.Ahi + (long)batch * q.sA;
    const _Float16* Alo = q.Alo + (long)batch * q.sA;
    const _Float16* Bhi = q.Bhi + (long)batch * q.sB;
    const _Float16* Blo = q.Blo + (long)batch * q.sB;
    const long lda = q.lda, ldn = q.ldn;
    long aoff[ACW];
    const int a_tiled = q.a_tiled;
    const long astage = a_tiled ? 512 : BKT;   // halves between consecutive k stages of one piece
#pragma unroll
    for (int c = 0; c < ACW; ++c) {   // piece = 16 rows x 64 B
        const int ca = wave + c * NW;
        if (a_tiled) {   // pre-tiled static operand: the piece is 1 KiB contiguous, already in its (swizzled) LDS image
            int rb = m0 / 16 + ca;
            const int nrb = (q.M + 15) / 16;
            rb = rb < nrb ? rb : nrb - 1;
            aoff[c] = (long)rb * lda * 16 + lane * 8;     // lda = padded K (multiple of 32)
        } else {         // row-major planes: the lane fetches the XOR-swizzled logical slot of its row
            const int row = ca * 16 + lane / 4;
            const int ls = (lane % 4) ^ ((row >> 2) & 3);
            int m = m0 + row;
            m = m < M ? m : M - 1;
            aoff[c] = (long)m * lda + 8 * ls;
        }
    }
    int bkg[BCW];
    long bcol[BCW];
    const int cplx = q.cplx;                    // complex-structured B: see Gemm4Args::cplx
    const int nhalf = cplx ? n0 / cplx : 0;     // 0: real output columns, 1: imaginary (tiles never straddle)
#pragma unroll
    for (int c = 0; c < BCW; ++c) {   // piece = one k group x 64 columns
        const int cb = wave + c * NW;
        bkg[c] = cb / (BN / 64);
        int nn = n0 + (cb % (BN / 64)) * 64 + lane;
        nn = nn < N ? nn : N - 1;
        if (cplx) nn -= nhalf * cplx;
        bcol[c] = (long)nn * 8;
    }
    // implicit-GEMM B: (plane group, tap row, tap column) of the k group each of this wave's pieces fetches NEXT; issue() is called
    // once per stage, in stage order, and steps them by the stage's four k groups (wave-uniform scalars, no table in memory)
    int icg[BCW], ity[BCW], itx[BCW];
    if constexpr (IMPL) {
#pragma unroll
        for (int c = 0; c < BCW; ++c) {
            const int tap = bkg[c] / q.impl_cg8;
            icg[c] = bkg[c] % q.impl_cg8; ity[c] = tap / q.impl_k; itx[c] = tap % q.impl_k;
        }
    }
    auto issue = [&](int k0, int buf) {
        _Float16* Ab = As + buf * 2 * APL;
        _Float16* Bb = Bs + buf * 2 * BPL;
#pragma unroll
        for (int c = 0; c < ACW; ++c) {
            const int ca = wave + c * NW;
            const long ak = (long)(k0 / BKT) * astage;
            glds16(reinterpret_cast<const float*>(Ahi + aoff[c] + ak), reinterpret_cast<float*>(Ab + ca * 512));
            glds16(reinterpret_cast<const float*>(Alo + aoff[c] + ak), reinterpret_cast<float*>(Ab + APL + ca * 512));
        }
#pragma unroll
        for (int c = 0; c < BCW; ++c) {
            const int cb = wave + c * NW;
            int kg = k0 / 8 + bkg[c];
            kg = kg < nkg ? kg : nkg - 1;   // groups past K meet zero A columns (values there are finite)
            long off = (long)kg * ldn * 8 + bcol[c];
            if (cplx) {                     // block (k half == n half ? Wr : Wi), k group inside the block
                const int khalf = k0 >= cplx;
                off = (long)(khalf != nhalf) * cplx * cplx + (long)(kg - khalf * (cplx / 8)) * ldn * 8 + bcol[c];
            }
            if constexpr (IMPL) {           // plane group at the tap's pixel shift; k groups past K: the last one (zero A columns)
                const bool past = ity[c] >= q.impl_k;
                const int cgq = past ? q.impl_cg8 - 1 : icg[c], ty = past ? q.impl_k - 1 : ity[c], tx = past ? q.impl_k - 1 : itx[c];
                off = ((long)cgq * ldn + (long)(ty * q.impl_pitch + tx) * q.impl_dil) * 8 + bcol[c];
                icg[c] += 4;
                while (icg[c] >= q.impl_cg8) {
                    icg[c] -= q.impl_cg8;
                    if (++itx[c] == q.impl_k) { itx[c] = 0; ++ity[c]; }
                }
            }
            glds16(reinterpret_cast<const float*>(Bhi + off), reinterpret_cast<float*>(Bb + cb * 512));
            glds16(reinterpret_cast<const float*>(Blo + off), reinterpret_cast<float*>(Bb + BPL + cb * 512));
        }
        // L2 prefetch of the B lines of stage (k0 / BKT + PFD): the DMA above has one stage of lookahead, enough for an
        // L2 hit but not for an HBM miss.  One lane per 128-byte line (BN lines per stage: 4 k groups x 2 planes x BN/8),
        // BN/4 lanes per wave; always exactly one instruction per call so that the counted waits stay uniform.
        if (PFD > 0) {
            const int li = wave * (BN / 4) + (lane % (BN / 4));
            const int plane = li / (BN / 2), rem = li % (BN / 2);
            int kg = (k0 + PFD * BKT) / 8 + rem / (BN / 8);
            kg = kg < nkg ? kg : nkg - 1;
            int nn = n0 + (rem % (BN / 8)) * 8;
            nn = nn < N ? nn : N - 1;
            const _Float16* src = (plane ? Blo : Bhi) + ((long)kg * ldn + nn) * 8;
            glds4(src, Scratch + wave * 128);
        }
    };

    const int i = lane & 31, g = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int arow0 = wm * 64 + i, arow1 = arow0 + 32;
    const int key0 = (arow0 >> 2) & 3, key1 = (arow1 >> 2) & 3;
    const int bcol0 = wn * 64 + i, bcol1 = bcol0 + 32;
    // Triangular launches (dhconv: rows m <= l) leave whole 64-row strips of a 128-row tile beyond M: such a wave keeps
    // its DMA share and the barriers but issues no fragment reads, no MFMAs and no epilogue (25 % of the dhconv's waves).
    const bool strip_on = m0 + wm * 64 < M;
    auto neg8 = [](half8 v) {   // -v: flip the eight sign bits
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 u = __builtin_bit_cast(u32x4, v);
        u ^= 0x80008000u;
        return __builtin_bit_cast(half8, u);
    };
    auto load_frags = [&](Frags4& f, int buf, int c, bool neg = false) {
        if (!strip_on) return;
        const _Float16* Ah = As + buf * 2 * APL;
        const _Float16* Al = Ah + APL;
        const _Float16* Bh = Bs + buf * 2 * BPL;
        const _Float16* Bl = Bh + BPL;
        const int ls = 2 * c + g;  // lane group g owns k = 16c + 8g .. +7
        f.al[0] = *reinterpret_cast<const half8*>(Al + arow0 * 32 + 8 * (ls ^ key0));
        f.al[1] = *reinterpret_cast<const half8*>(Al + arow1 * 32 + 8 * (ls ^ key1));
        f.bh[0] = *reinterpret_cast<const half8*>(Bh + ((long)ls * BN + bcol0) * 8);
        f.bh[1] = *reinterpret_cast<const half8*>(Bh + ((long)ls * BN + bcol1) * 8);
        f.ah[0] = *reinterpret_cast<const half8*>(Ah + arow0 * 32 + 8 * (ls ^ key0));
        f.ah[1] = *reinterpret_cast<const half8*>(Ah + arow1 * 32 + 8 * (ls ^ key1));
        f.bl[0] = *reinterpret_cast<const half8*>(Bl + ((long)ls * BN + bcol0) * 8);
        f.bl[1] = *reinterpret_cast<const half8*>(Bl + ((long)ls * BN + bcol1) * 8);
        if (neg) {   // wave-uniform: the (-Wi) quadrant of the complex-structured operand
            f.al[0] = neg8(f.al[0]); f.al[1] = neg8(f.al[1]);
            f.ah[0] = neg8(f.ah[0]); f.ah[1] = neg8(f.ah[1]);
        }
    };
    auto negq = [&](int kt) { return cplx != 0 && nhalf == 0 && kbeg + kt * BKT >= cplx; };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto mma = [&](const Frags4& f) {  // 12 MFMAs: small cross terms first, the hi.hi term last
        if (!strip_on) return;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[a], f.bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bh[b], acc[a][b], 0, 0, 0);
    };
    auto interleave = [&]() {  // 8 fragment reads spread under 12 MFMAs
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };

    // Software pipeline at half-stage granularity: while the 12 MFMAs of one 16-deep half run, the fragments of the
    // next half are read from LDS into the other register set.  One barrier per stage, between the two halves: by
    // then every wave has read all of buffer t & 1, so the DMA of stage t + 2 may overwrite it, and the DMA of
    // stage t + 1 (issued one stage earlier) has landed.
    Frags4 f0, f1;
    G4T(1);
    if (RES && ACE_G4_RTOUCH) {  // touch the residual tile now: BM rows x (BN * 4 / 128) lines, it is needed only in the epilogue
        const float* Rt = q.R + (long)batch * q.sR;
        constexpr int LPR = BN / 32;                       // 128-byte lines per tile row
#pragma unroll
        for (int t = 0; t < BM * LPR / (64 * NW); ++t) {
            const int li = (t * NW + wave) * 64 + lane;
            int row = m0 + li / LPR;
            row = row < M ? row : M - 1;
            int col = n0 + (li % LPR) * 32;
            col = col < N ? col : N - 1;
            glds4(Rt + (long)row * q.ldr + col, Scratch + wave * 128);
        }
    }
    if (nk > 0) issue(kbeg, 0);
    if (nk > 1) issue(kbeg + BKT, 1);
    // scales: the producer of a dynamic operand scaled it by 2^(12 - exponent(bound)); undo both here (exact)
    auto wave_max = [&](unsigned raw) {
        float mx = __uint_as_float(raw);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    };
    float bbound = 0.f;
    float inv_a = 1.0f / q.ascale, inv_b = 1.0f / q.bscale;
    if (q.amax) inv_a = ldexpf(1.0f, -pow2_exponent_for(wave_max(raw_a)));
    if (q.bmax) { bbound = wave_max(raw_b); inv_b = ldexpf(1.0f, -pow2_exponent_for(bbound)); }
    float cscale = 1.f;
    if (PK) {  // bound of this launch's output, identical in every workgroup; consumers read it from cslot
        const float inb = q.cinb ? wave_max(raw_c) : bbound;      // bound of the (normalised) conv input
        const float resb = q.rmax ? wave_max(raw_r) : 0.f;        // bound of the residual term
        float cbound = fmaf(q.cw, inb, q.cb) + resb;
        if constexpr (IMPL) { if (q.act != ACT_NONE) cbound = fminf(cbound, fmaxf(q.cap, 0.25f)); }   // capped GELU / ReLU: -0.17 <= value <= cap
        cscale = ldexpf(1.0f, pow2_exponent_for(cbound));
        if (tid == 0) atomicMax(q.cslot + (blockIdx.x & 63), __float_as_uint(cbound));
    }
    constexpr int NTOUCH = PFD > 0 ? 1 : 0;   // prefetch touches per issue() that may stay outstanding
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS + NTOUCH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    G4T(2);
    if (nk > 0) load_frags(f0, 0, 0, negq(0));
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        load_frags(f1, cur, 1, negq(kt));
        mma(f0);
        interleave();
#if ACE_G4_PIN
        __builtin_amdgcn_sched_barrier(0);   // keep the 12 MFMAs above the wait: without it the scheduler sinks 11 of them below
#endif                                       // the barrier and the DMA of the next stage gets half the lookahead
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NTOUCH) : "memory");   // all but the newest prefetch touch
        __syncthreads();
        if (kt + 2 < nk) issue(kbeg + (kt + 2) * BKT, cur);
        if (kt + 1 < nk) load_frags(f0, cur ^ 1, 0, negq(kt + 1));
        mma(f1);
        interleave();
        G4T(3 + (kt < 40 ? kt : 40));
    }
    G4T(50);

    // ---- epilogue, through LDS: each wave parks its 64x64 tile (accumulator layout) in its own 16 KiB of the now idle
    // operand buffers.  No barrier: all LDS reads of the main loop completed before its last barrier, and waves use
    // disjoint regions.
    //   phase 2 - read back row-contiguous (4 rows x 256 B per wave instruction): scales, bias, residual (16-byte loads),
    //             activation; optional fp32 store (16 B per lane); optional per-row statistics of the final values
    //             (sum, sum of squares, min, max over this wave's 64 columns -> `part`, reduced later by
    //             instnorm_finalize_kernel: the fused instance norm of the NEXT layer needs no pass over the tensor);
    //   phase 3 - optional P-format output: final values go back to LDS and are re-read 8 rows x 1 column per lane,
    //             split hi/lo and stored as whole 16-byte entries (1 KiB per wave instruction).
    float vmax = 0.f;
    float* C = q.C ? q.C + (long)batch * q.sC : nullptr;
    const float* R = RES ? q.R + (long)batch * q.sR : nullptr;
    const float* rsc = q.rsc ? q.rsc + (long)batch * q.srs : nullptr;
    const float* rsh = q.rsc ? q.rsh + (long)batch * q.srs : nullptr;
    const float* bias = q.bias ? q.bias + (long)batch * q.sbias : nullptr;
    const float* dummy = reinterpret_cast<const float*>(q.Ahi);
    const long ldc = q.ldc, ldr = q.ldr;
    const int actk = q.act;
    float4* part = q.part ? q.part + ((long)batch * tilesN * WN + (long)tile_n * WN + wn) * q.M : nullptr;
    if (!strip_on) {
        // nothing to write: every row of this wave's strip is beyond M
    } else if (ACE_G4_REGEPI && PK && !RES && !C && !part) {
        // Register-only variant.  A lane of the 32x32 accumulator tile holds rows {0-3, 8-11, 16-19, 24-27} (+4 in the upper
        // half-wave) of its column; v_permlane32_swap(vdst, src) exchanges vdst[32..63] with src[0..31] (probed:
        // tools/permlane_probe.hip), so swapping r(4j+e) pairs (r0<->r4 .. r3<->r7, r8<->r12 .. r11<->r15) leaves rows
        // 8g .. 8g+7 in r0..r7 and rows 16+8g .. 16+8g+7 in r8..r15: whole 8-row P entries, no LDS.
        _Float16* Chi = q.Chi + (long)batch * q.sCp;
        _Float16* Clo = q.Clo + (long)batch * q.sCp;
        const int brow = m0 + wm * 64 + lane;
        const float bl = bias ? bias[brow < M ? brow : 0] : 0.f;
        dispatch_act(actk, [&](auto at) {
            constexpr int AC = decltype(at)::value;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    f32x16 v = acc[tm][tn];
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * hq + e]),
                                                                             __float_as_uint(v[8 * hq + 4 + e]), false, false);
                            v[8 * hq + e] = __uint_as_float(sw[0]);
                            v[8 * hq + 4 + e] = __uint_as_float(sw[1]);
                        }
                    const int col = n0 + wn * 64 + tn * 32 + i;
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        const int rl = tm * 32 + hq * 16 + g * 8;       // first of this lane's 8 rows within the wave's 64
                        const int rbase = m0 + wm * 64 + rl;
                        half8 hh, ll;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            // bias of row rl + e: the two half-waves need different rows of the coalesced load
                            const float b0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bl), tm * 32 + hq * 16 + e));
                            const float b1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bl), tm * 32 + hq * 16 + 8 + e));
                            const float x = act_const<AC>(fmaf(v[8 * hq + e] * inv_a, inv_b, g ? b1 : b0), actk) * cscale;
                            const _Float16 a = (_Float16)x;
                            hh[e] = a;
                            ll[e] = (_Float16)(x - (float)a);
                        }
                        if (rbase < M && col < N) {
                            const long eo = ((long)(rbase >> 3) * q.ldnc + col) * 8;
                            *reinterpret_cast<half8*>(Chi + eo) = hh;
                            *reinterpret_cast<half8*>(Clo + eo) = ll;
                        }
                    }
                }
        });
    } else if (PK && !RES && !C && !part) {
        // only the P-format output (fc1 -> hidden activation): park, then lane = column, 8 rows per 16-byte entry
        float* Ts = reinterpret_cast<float*>(smem4) + wave * 4096;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(tm * 32 + acc_row(r, g)) * 64 + tn * 32 + i] = acc[tm][tn][r];
        _Float16* Chi = q.Chi + (long)batch * q.sCp;
        _Float16* Clo = q.Clo + (long)batch * q.sCp;
        const int col = n0 + wn * 64 + lane;
        const bool colok = col < N;
        // the 64 row biases of this wave: one coalesced load, broadcast per row with v_readlane
        const int brow = m0 + wm * 64 + lane;
        const float bl = bias ? bias[brow < M ? brow : 0] : 0.f;
        dispatch_act(actk, [&](auto at) {
            constexpr int AC = decltype(at)::value;
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) {
                const int rbase = m0 + wm * 64 + 8 * rg;
                float tv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) tv[e] = Ts[(8 * rg + e) * 64 + lane];
                half8 hh, ll;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float bvs = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bl), 8 * rg + e));
                    float v = act_const<AC>(fmaf(tv[e] * inv_a, inv_b, bvs), actk);
                    if constexpr (IMPL) v = fminf(v, q.cap);
                    const float x = v * cscale;
                    const _Float16 a = (_Float16)x;
                    hh[e] = a;
                    ll[e] = (_Float16)(x - (float)a);
                }
                if (rbase < M && colok) {
                    const long eo = ((long)(rbase >> 3) * q.ldnc + col) * 8;
                    *reinterpret_cast<half8*>(Chi + eo) = hh;
                    *reinterpret_cast<half8*>(Clo + eo) = ll;
                }
            }
        });
    } else {
        float* Ts = reinterpret_cast<float*>(smem4) + wave * 4096;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(tm * 32 + acc_row(r, g)) * 64 + tn * 32 + i] = acc[tm][tn][r];
        const int c4 = (lane & 15) * 4;
        const int colb = n0 + wn * 64 + c4;
        const bool cok = colb < N;           // N % 4 == 0 (checked by the launcher)
        const int colc = cok ? colb : 0;
        // four rows at a time: loads, math and stores of successive chunks overlap, and the register footprint stays small
        dispatch_act(actk, [&](auto at) {
        constexpr int AC = decltype(at)::value;
#pragma unroll 1
        for (int jb = 0; jb < 16; jb += 4) {
            float4 tv[4], rv[4];
            float bvv[4], rsv[4], rtv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int rl = (lane >> 4) + 4 * (jb + jj);
                const int row = m0 + wm * 64 + rl;
                const int rr = row < M ? row : 0;
                const float t0 = (bias ? bias : dummy)[bias ? rr : 0];
                const float t1 = (rsc ? rsc : dummy)[rsc ? rr : 0];
                const float t2 = (rsc ? rsh : dummy)[rsc ? rr : 0];
                bvv[jj] = bias ? t0 : 0.f;
                rsv[jj] = rsc ? t1 : 1.f;
                rtv[jj] = rsc ? t2 : 0.f;
                if (RES) rv[jj] = *reinterpret_cast<const float4*>(R + (long)rr * ldr + colc);
                tv[jj] = *reinterpret_cast<const float4*>(Ts + rl * 64 + c4);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int rl = (lane >> 4) + 4 * (jb + jj);
                const int row = m0 + wm * 64 + rl;
                float o[4] = {tv[jj].x, tv[jj].y, tv[jj].z, tv[jj].w};
                const float r4[4] = {RES ? rv[jj].x : 0.f, RES ? rv[jj].y : 0.f, RES ? rv[jj].z : 0.f, RES ? rv[jj].w : 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(o[e] * inv_a, inv_b, bvv[jj]);
                    if (RES) v += fmaf(r4[e], rsv[jj], rtv[jj]);
                    o[e] = act_const<AC>(v, actk);
                    if constexpr (IMPL) o[e] = fminf(o[e], q.cap);
                }
                const bool ok = row < M && cok;
                if (C && ok) *reinterpret_cast<float4*>(C + (long)row * ldc + colb) = make_float4(o[0], o[1], o[2], o[3]);
                if (ok) vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                if (PK) *reinterpret_cast<float4*>(Ts + rl * 64 + c4) = make_float4(o[0], o[1], o[2], o[3]);
                if (part) {   // wave-uniform branch
                    float sm = ok ? (o[0] + o[1]) + (o[2] + o[3]) : 0.f;
                    float sq = ok ? fmaf(o[0], o[0], o[1] * o[1]) + fmaf(o[2], o[2], o[3] * o[3]) : 0.f;
                    float mn = ok ? fminf(fminf(o[0], o[1]), fminf(o[2], o[3])) : 3.0e38f;
                    float mx = ok ? fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3])) : -3.0e38f;
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) {   // the 16 lanes that share this row
                        sm += __shfl_xor(sm, off, 64);
                        sq += __shfl_xor(sq, off, 64);
                        mn = fminf(mn, __shfl_xor(mn, off, 64));
                        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                    }
                    if ((lane & 15) == 0 && row < M) part[row] = make_float4(sm, sq, mn, mx);
                }
            }
        }
        });
        if (PK) {
            _Float16* Chi = q.Chi + (long)batch * q.sCp;
            _Float16* Clo = q.Clo + (long)batch * q.sCp;
            const int col = n0 + wn * 64 + lane;
            const bool colok = col < N;
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) {
                const int rbase = m0 + wm * 64 + 8 * rg;
                half8 hh, ll;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = Ts[(8 * rg + e) * 64 + lane] * cscale;
                    const _Float16 a = (_Float16)x;
                    hh[e] = a;
                    ll[e] = (_Float16)(x - (float)a);
                }
                if (rbase < M && colok) {   // M % 8 == 0: the 8-row group is wholly inside
                    const long eo = ((long)(rbase >> 3) * q.ldnc + col) * 8;
                    *reinterpret_cast<half8*>(Chi + eo) = hh;
                    *reinterpret_cast<half8*>(Clo + eo) = ll;
                }
            }
        }
    }
    G4T(53);
    if (q.omax) {   // one atomic per workgroup: the 64 slots share two cache lines and same-line atomics serialise
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float* red = reinterpret_cast<float*>(smem4);
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
            atomicMax(q.omax + (blockIdx.x & 63), __float_as_uint(m));
        }
    }
#ifdef ACE_X_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G4T(54);
#endif
}

template <int WM, int WN, bool RES, bool PK>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm4_f16x3_kernel(Gemm4Args q, int tilesM, int tilesN) {
    gemm4_body<WM, WN, RES, PK, false>(q, tilesM, tilesN);
}
// the same engine with the implicit-GEMM B operand (Gemm4Args::impl_k): HEALPix k x k convolutions (PK: P-format output)
template <int WM, int WN, bool PK>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm4_implicit_kernel(Gemm4Args q, int tilesM, int tilesN) {
    gemm4_body<WM, WN, false, PK, true>(q, tilesM, tilesN);
}

// fp32 [K][N] (row pitch ldb) -> P-format fp16 hi/lo planes [ceil(K/8)][ldn][8], optional per-row affine (fused
// instance norm), scaled by 2^(12 - exponent(slot)).  Thread = (k group, 4 consecutive columns).
__global__ __launch_bounds__(256) void pack_pformat_kernel(const float* __restrict__ src, long ldb, long sSrc, int K,
                                                            int N, const float* __restrict__ sc,
                                                            const float* __restrict__ sh, long sbs,
                                                            const unsigned* __restrict__ slot, _Float16* __restrict__ hi,
                                                            _Float16* __restrict__ lo, long ldn, long sPl) {
    const int lane = threadIdx.x & 63;
    const float scale = ldexpf(1.0f, pow2_exponent_for(slot_reduce(slot, lane)));
    const int nq = N / 4;
    const int batch = blockIdx.y;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int kg = (int)(t / nq);
    const int n = (int)(t % nq) * 4;
    if (kg >= (K + 7) / 8) return;
    const float* base = src + (long)batch * sSrc + n;
    float4 v[8];
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * kg + e;
        const int kc = k < K ? k : K - 1;
        v[e] = *reinterpret_cast<const float4*>(base + (long)kc * ldb);
        a[e] = sc ? sc[(long)batch * sbs + kc] * scale : scale;
        b[e] = sc ? sh[(long)batch * sbs + kc] * scale : 0.f;
        if (k >= K) { a[e] = 0.f; b[e] = 0.f; }
    }
    half8 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xs[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = __builtin_amdgcn_fmed3f(fmaf(xs[j], a[e], b[e]), -65504.f, 65504.f);
            const _Float16 hh = (_Float16)x;
            h[j][e] = hh;
            l[j][e] = (_Float16)(x - (float)hh);
        }
    }
    _Float16* ph = hi + (long)batch * sPl + ((long)kg * ldn + n) * 8;
    _Float16* pl = lo + (long)batch * sPl + ((long)kg * ldn + n) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<half8*>(ph + 8 * j) = h[j];
        *reinterpret_cast<half8*>(pl + 8 * j) = l[j];
    }
}

hipError_t launch_pack_pformat(const float* src, long ldb, long sSrc, int K, int N, int nbatch, const float* sc,
                               const float* sh, long sbs, const unsigned* slot, void* hi, void* lo, long ldn, long sPl,
                               hipStream_t s) {
    if (N % 4 != 0 || !al16(src) || ldb % 4 != 0 || sSrc % 4 != 0) return hipErrorInvalidValue;
    const long total = (long)((K + 7) / 8) * (N / 4);
    dim3 grid((unsigned)((total + 255) / 256), (unsigned)nbatch);
    hipLaunchKernelGGL(pack_pformat_kernel, grid, dim3(256), 0, s, src, ldb, sSrc, K, N, sc, sh, sbs, slot,
                       static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), ldn, sPl);
    return hipGetLastError();
}

template <int WM, int WN>
static hipError_t launch_gemm4_cfg(const Gemm4Args& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr size_t lds = (size_t)2 * 2 * (BM * 32 + 32 * BN) * sizeof(_Float16) + ((ACE_G4_PFD > 0 || ACE_G4_RTOUCH) ? 4 * 256 : 0);
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    const long nblk = (long)tilesM * tilesN * a.nbatch;
    if (nblk <= 0) return hipSuccess;
    const bool res = a.R != nullptr, pk = a.Chi != nullptr;
    static bool configured[4] = {false, false, false, false};
    const void* fn = pk ? (res ? (const void*)gemm4_f16x3_kernel<WM, WN, true, true> : (const void*)gemm4_f16x3_kernel<WM, WN, false, true>)
                        : (res ? (const void*)gemm4_f16x3_kernel<WM, WN, true, false> : (const void*)gemm4_f16x3_kernel<WM, WN, false, false>);
    const int ci = (pk ? 2 : 0) + (res ? 1 : 0);
    if (!configured[ci]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[ci] = true;
    }
    dim3 grid((unsigned)nblk), block(64 * WM * WN);
    if (a.impl_k > 0) {
        static bool configured_impl[2] = {false, false};
        const void* fi = pk ? (const void*)gemm4_implicit_kernel<WM, WN, true> : (const void*)gemm4_implicit_kernel<WM, WN, false>;
        if (!configured_impl[pk]) {
            hipError_t e = hipFuncSetAttribute(fi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured_impl[pk] = true;
        }
        if (pk) hipLaunchKernelGGL((gemm4_implicit_kernel<WM, WN, true>), grid, block, lds, s, a, tilesM, tilesN);
        else hipLaunchKernelGGL((gemm4_implicit_kernel<WM, WN, false>), grid, block, lds, s, a, tilesM, tilesN);
        return hipGetLastError();
    }
    if (pk && res) hipLaunchKernelGGL((gemm4_f16x3_kernel<WM, WN, true, true>), grid, block, lds, s, a, tilesM, tilesN);
    else if (pk) hipLaunchKernelGGL((gemm4_f16x3_kernel<WM, WN, false, true>), grid, block, lds, s, a, tilesM, tilesN);
    else if (res) hipLaunchKernelGGL((gemm4_f16x3_kernel<WM, WN, true, false>), grid, block, lds, s, a, tilesM, tilesN);
    else hipLaunchKernelGGL((gemm4_f16x3_kernel<WM, WN, false, false>), grid, block, lds, s, a, tilesM, tilesN);
    return hipGetLastError();
}

// number of 64-column strips (= statistics partials per row) the launcher's tile choice produces for an M x N problem
int gemm4_strips(int M, int N) {
    const long waste128 = (long)((M + 127) / 128) * 128, waste64 = (long)((M + 63) / 64) * 64;
    const bool big = M >= 128 && waste128 <= waste64;
    return big ? ((N + 127) / 128) * 2 : ((N + 255) / 256) * 4;
}
hipError_t launch_gemm_f16x3_packed(const Gemm4Args& a, hipStream_t s) {
    if (!al16(a.Ahi) || !al16(a.Alo) || !al16(a.Bhi) || !al16(a.Blo) || a.lda % 8 != 0 || a.sA % 8 != 0 || a.sB % 8 != 0)
        return hipErrorInvalidValue;
    if (a.Chi && (a.M % 8 != 0 || !a.cslot)) return hipErrorInvalidValue;
    if (!a.Chi && !a.C) return hipErrorInvalidValue;
    if (a.N % 4 != 0 || (a.C && (!al16(a.C) || a.ldc % 4 != 0 || a.sC % 4 != 0))) return hipErrorInvalidValue;
    if (a.R && (!al16(a.R) || a.ldr % 4 != 0 || a.sR % 4 != 0)) return hipErrorInvalidValue;
    if (a.impl_k > 0 && (a.R || a.cplx || a.tri != TRI_NONE || a.part || a.impl_cg8 < 1 || a.impl_dil < 1 || a.impl_pitch < 1 ||
                         a.K != a.impl_k * a.impl_k * a.impl_cg8 * 8 || !a.a_tiled))
        return hipErrorInvalidValue;
    const long waste128 = (long)((a.M + 127) / 128) * 128, waste64 = (long)((a.M + 63) / 64) * 64;
#ifdef ACE_MEASUREMENT_SWITCHES
    static const int env_force = getenv("ACE_G4_TILE") ? atoi(getenv("ACE_G4_TILE")) : 0;   // A/B switch: 1 = 128x128, 2 = 64x256
#else
    constexpr int env_force = 0;
#endif
    const int force = a.tile ? a.tile : env_force;
    if (a.cplx && (a.cplx % 128 != 0 || a.K != 2 * a.cplx || a.N != 2 * a.cplx || a.ldn != a.cplx || force != 1)) return hipErrorInvalidValue;
    if (force == 1 || (force == 0 && a.M >= 128 && waste128 <= waste64)) return launch_gemm4_cfg<2, 2>(a, s);
    return launch_gemm4_cfg<1, 4>(a, s);
}

// ---------------------------------------------------------------------------------------------
// forward longitude DFT (2*pi*rfft(norm="forward"), fft.py:61-76 via sht_fix.py:127), folded:
//   Re X[m] = sum_{w<=W/2} fc[m][w] * (x[w] + x[W-w]),  Im X[m] = sum fs[m][w] * (x[w] - x[W-w])
// rows = m, columns = (k, b, c).  One workgroup produces BOTH the real and the imaginary rows of its columns
// (two accumulator sets fed by the even / odd folds of the same loaded x values), so x is read once per
// m-tile.  The instance-norm affine and the fold happen when the tile is written to LDS, after the MFMAs
// of the previous stage.  Output goes straight to the channel-fastest spectral layout X[m][k][b][ri][c].
// ---------------------------------------------------------------------------------------------
template <int BN, int NT, bool VEC>
struct StageFold {
    static constexpr int G = BN * BK / 4 / NT;
    static constexpr int SB = BN + 2;  // transposing b32 writes: pitch % 8 == 2
    float4 fw[G];   // x[w .. w+3]
    float4 mr[G];   // x[W-w-4 .. W-w-1] (reversed neighbours of the mirror), VEC path
    float m0v[G];   // x[W-w] (mirror of element 0)
    const float* base[G];
    float sc[G], sh[G];
    DEVINL void init(const DftArgs& p, int n0, int ncols, int tid) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int n = n0 + f / (BK / 4);
            const int nn = n < ncols ? n : 0;
            const int c = nn % p.C, kb = nn / p.C;
            const int b = kb % p.Bt, k = kb / p.Bt;
            base[g] = p.x + ((long)(b * p.C + c) * p.H + k) * p.W;
            sc[g] = 1.f;
            sh[g] = 0.f;
            if (p.sc) { sc[g] = p.sc[b * p.C + c]; sh[g] = p.sh[b * p.C + c]; }
            if (n >= ncols) { sc[g] = 0.f; sh[g] = 0.f; }  // columns past the end contribute exact zeros
        }
    }
    // element e of this group is longitude w+e; its mirror is W-w-e
    DEVINL void load(int W, int w0, int tid) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int w = w0 + 4 * (f % (BK / 4));
            const float* bp = base[g];
            if (VEC) {  // W % 4 == 0: w+3 < W always holds for w <= W/2 rounded to 4 when W >= 8
                const int wc = (w + 3 < W) ? w : 0;
                fw[g] = *reinterpret_cast<const float4*>(bp + wc);
                const int mo = W - w - 4;  // >= 0 iff w + 4 <= W
                mr[g] = *reinterpret_cast<const float4*>(bp + (mo >= 0 ? mo : 0));
                m0v[g] = bp[(w > 0 && w < W) ? W - w : 0];
            } else {
                float t[4], u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int we = w + e;
                    t[e] = bp[we < W ? we : 0];
                    u[e] = bp[(we > 0 && we < W) ? W - we : 0];
                }
                fw[g] = make_float4(t[0], t[1], t[2], t[3]);
                m0v[g] = u[0];
                mr[g] = make_float4(0.f, u[3], u[2], u[1]);  // same packing as the VEC path: (.w,.z,.y) = mirrors of e=1,2,3
            }
        }
    }
    // writes the even fold into Be and the odd fold into Bo
    DEVINL void store(float* Be, float* Bo, int W, int Kf, int w0, int tid) const {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int nl = f / (BK / 4), wc = f % (BK / 4);
            const int w = w0 + 4 * wc;
            const float xs[4] = {fw[g].x, fw[g].y, fw[g].z, fw[g].w};
            const float ms[4] = {m0v[g], mr[g].w, mr[g].z, mr[g].y};
            float* de = Be + (4 * wc) * SB + nl;
            float* dd = Bo + (4 * wc) * SB + nl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int we = w + e;
                const float a = fmaf(xs[e], sc[g], sh[g]);
                const float m = fmaf(ms[e], sc[g], sh[g]);
                const bool in = we < Kf;
                const bool unpaired = (we == 0) || (2 * we == W);
                de[e * SB] = in ? (unpaired ? a : a + m) : 0.f;
                dd[e * SB] = (in && !unpaired) ? a - m : 0.f;
            }
        }
    }
};

template <bool VX>
__global__ __launch_bounds__(256) void dft_forward_kernel(DftArgs p, int tilesM, int tilesN) {
    constexpr int BM = 64, BN = 128, NT = 256;     // 4 waves along n; each wave: 64 (m) x 32 (n) x {re, im}
    using SA_t = StageA<BM, NT, true>;  // tables are library-owned: 16B aligned, pitch % 4 == 0, zero padded
    using SB_t = StageFold<BN, NT, VX>;
    constexpr int SA = SA_t::SA, SB = SB_t::SB;
    constexpr int ABUF = 2 * BK * SA, BBUF = 2 * BK * SB;  // per buffer: {cos, sin} tables, {even, odd} folds
    __shared__ __attribute__((aligned(16))) float smem[2 * ABUF + 2 * BBUF];
    float* As = smem;
    float* Bs = smem + 2 * ABUF;

    const int nblk = tilesM * tilesN;
    const int lid = xcd_remap(blockIdx.x, nblk);
    const int tile_m = lid % tilesM;  // the m-tiles of one column panel run back to back (shared x lines in L2)
    const int tile_n = lid / tilesM;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ncols = p.H * p.Bt * p.C;
    const int Kf = p.W / 2 + 1;
    const int Kp = p.ldt;  // table pitch = Kf rounded up to 4, zero padded

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int bcol = wave * 32 + i;

    f32x16 Re[2], Im[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Re[t][r] = 0.f; Im[t][r] = 0.f; }

    SA_t sac, sas;
    SB_t sb;
    sb.init(p, n0, ncols, tid);
    const int nk = (Kf + BK - 1) / BK;
    sac.load(p.tc, p.ldt, m0, p.Mm, 0, Kp, tid);
    sas.load(p.ts, p.ldt, m0, p.Mm, 0, Kp, tid);
    sb.load(p.W, 0, tid);
    sac.store(As, m0, p.Mm, 0, Kp, tid);
    sas.store(As + BK * SA, m0, p.Mm, 0, Kp, tid);
    sb.store(Bs, Bs + BK * SB, p.W, Kf, 0, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
            const int k0 = (kt + 1) * BK;
            sac.load(p.tc, p.ldt, m0, p.Mm, k0, Kp, tid);
            sas.load(p.ts, p.ldt, m0, p.Mm, k0, Kp, tid);
            sb.load(p.W, k0, tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* Ac = As + cur * ABUF;
            const float* Asn = Ac + BK * SA;
            const float* Be = Bs + cur * BBUF;
            const float* Bo = Be + BK * SB;
#pragma unroll
            for (int ks = 0; ks < BK; ks += 2) {
                const float c0 = Ac[(ks + h) * SA + i];
                const float c1 = Ac[(ks + h) * SA + i + 32];
                const float s0 = Asn[(ks + h) * SA + i];
                const float s1 = Asn[(ks + h) * SA + i + 32];
                const float ev = Be[(ks + h) * SB + bcol];
                const float od = Bo[(ks + h) * SB + bcol];
                Re[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, ev, Re[0], 0, 0, 0);
                Re[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, ev, Re[1], 0, 0, 0);
                Im[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, od, Im[0], 0, 0, 0);
                Im[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, od, Im[1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            const int k0 = (kt + 1) * BK;
            sac.store(As + (cur ^ 1) * ABUF, m0, p.Mm, k0, Kp, tid);
            sas.store(As + (cur ^ 1) * ABUF + BK * SA, m0, p.Mm, k0, Kp, tid);
            sb.store(Bs + (cur ^ 1) * BBUF, Bs + (cur ^ 1) * BBUF + BK * SB, p.W, Kf, k0, tid);
        }
        __syncthreads();
    }

    const long N2 = (long)p.Bt * 2 * p.C;
    const int n = n0 + bcol;
    float vmax = 0.f;
    if (VX && (p.C % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.spec_out) & 15) == 0)) {
        // through LDS, one plane (Re, then Im) at a time: each wave parks its 64 (m) x 32 (n) tile and reads it back
        // row-contiguous, so the stores are 16 B per lane (8 rows x 128 B per wave instruction instead of 2 x 128 B).
        // The operand buffers are idle: the main loop ended on a barrier.
        float* Ts = smem + wave * 2048;
        const int c4 = (lane & 7) * 4;
        const int nb = n0 + wave * 32 + c4;          // first of this lane's 4 columns (same kb: C % 4 == 0)
        const int nbc = nb < ncols ? nb : 0;
        const int cb = nbc % p.C, kbb = nbc / p.C;
        float* ob = p.spec_out + (long)kbb * 2 * p.C + cb;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) Ts[(tm * 32 + acc_row(r, h)) * 32 + i] = pl == 0 ? Re[tm][r] : Im[tm][r];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ml = (lane >> 3) + 8 * j;
                const float4 v = *reinterpret_cast<const float4*>(Ts + ml * 32 + c4);
                const int m = m0 + ml;
                if (m < p.Mm && nb < ncols) {
                    *reinterpret_cast<float4*>(ob + (long)m * p.H * N2 + pl * p.C) = v;
                    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
            }
        }
    } else if (n < ncols) {
        const int c = n % p.C, kb = n / p.C;  // kb = k * Bt + b
        float* obase = p.spec_out + (long)kb * 2 * p.C + c;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + tm * 32 + acc_row(r, h);
                if (m < p.Mm) {
                    float* o = obase + (long)m * p.H * N2;
                    o[0] = Re[tm][r];
                    o[p.C] = Im[tm][r];
                    vmax = fmaxf(vmax, fmaxf(fabsf(Re[tm][r]), fabsf(Im[tm][r])));
                }
            }
    }
    if (p.omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(vmax));
    }
}

hipError_t launch_dft_forward(const DftArgs& a, hipStream_t s) {
    hipError_t ferr = hipSuccess;
    if (launch_dft_forward_fft(a, s, &ferr)) return ferr;
    if (!a.x) return hipErrorInvalidValue;   // planes input: FFT form only
    constexpr int BM = 64, BN = 128;
    const int ncols = a.H * a.Bt * a.C;
    const int tilesM = (a.Mm + BM - 1) / BM, tilesN = (ncols + BN - 1) / BN;
    const bool vx = al16(a.x) && (a.W % 4 == 0) && (a.W >= 8);
    dim3 grid((unsigned)(tilesM * tilesN)), block(256);
    if (vx)
        hipLaunchKernelGGL((dft_forward_kernel<true>), grid, block, 0, s, a, tilesM, tilesN);
    else
        hipLaunchKernelGGL((dft_forward_kernel<false>), grid, block, 0, s, a, tilesM, tilesN);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// inverse longitude DFT (irfft(norm="forward") after zeroing Im of m=0/Nyquist, fft.py:78-96), folded:
//   P[w] = sum_m Re S[m] gc[m][w],  Q[w] = sum_m Im S[m] gs[m][w];  y[w] = P + Q,  y[W-w] = P - Q
// rows = (k, b, c) flattened, columns = w <= W/2.  gs[0][:] = gs[W/2][:] = 0 exactly, which is the
// reference's "zero the imaginary part" step.  The spectral-filter bias is added on the way out.
// ---------------------------------------------------------------------------------------------
template <int BM, int NT, bool VEC>
struct StageSpecK {  // A operand, k-major source: element (row nn, k = m) at spec[m*ms + rb(nn) + ri*C]
    static constexpr int G = BM * BK / 4 / NT;
    static constexpr int SA = BM + 4;
    float4 r[2][G];
    long rb[G][VEC ? 1 : 4];
    bool ok[G][VEC ? 1 : 4];
    DEVINL void init(int nn0, int nrows, int C, int tid) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int mc = f % (BM / 4);
#pragma unroll
            for (int e = 0; e < (VEC ? 1 : 4); ++e) {
                const int nn = nn0 + 4 * mc + e;
                ok[g][e] = nn < nrows;  // VEC: C % 4 == 0 so the four rows are valid together
                const int nc = ok[g][e] ? nn : 0;
                rb[g][e] = (long)(nc / C) * 2 * C + (nc % C);
            }
        }
    }
    DEVINL void load(const float* __restrict__ S, long ms, int C, int Mm, int k0, int tid) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int m = k0 + f / (BM / 4);
            const float* p0 = S + (long)(m < Mm ? m : 0) * ms;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                const float* p = p0 + (long)ri * C;
                if (VEC) r[ri][g] = *reinterpret_cast<const float4*>(p + rb[g][0]);
                else r[ri][g] = make_float4(p[rb[g][0]], p[rb[g][1]], p[rb[g][2]], p[rb[g][3]]);
            }
        }
    }
    DEVINL void store(float* As /* [2][BK][SA] */, int Mm, int k0, int tid) const {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = tid + g * NT;
            const int kr = f / (BM / 4), mc = f % (BM / 4);
            const bool mok = k0 + kr < Mm;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                float4 v = r[ri][g];
                v.x = (mok && ok[g][0]) ? v.x : 0.f;
                v.y = (mok && ok[g][VEC ? 0 : 1]) ? v.y : 0.f;
                v.z = (mok && ok[g][VEC ? 0 : 2]) ? v.z : 0.f;
                v.w = (mok && ok[g][VEC ? 0 : 3]) ? v.w : 0.f;
                *reinterpret_cast<float4*>(As + ri * BK * SA + kr * SA + 4 * mc) = v;
            }
        }
    }
};

template <bool VS>
__global__ __launch_bounds__(256) void dft_inverse_kernel(DftArgs p, int tilesM, int tilesN) {
    constexpr int BM = 128, BN = 64, NT = 256;
    using SA_t = StageSpecK<BM, NT, VS>;
    using SB_t = StageB<BN, NT, true, false>;
    constexpr int SA = SA_t::SA, SB = SB_t::SB;
    // per buffer: A re/im planes [2][BK][SA], tables cos/sin [2][BK][SB]
    constexpr int ABUF = 2 * BK * SA, BBUF = 2 * BK * SB;
    __shared__ __attribute__((aligned(16))) float smem[2 * ABUF + 2 * BBUF];
    float* As = smem;
    float* Bs = smem + 2 * ABUF;

    const int nblk = tilesM * tilesN;
    const int lid = xcd_remap(blockIdx.x, nblk);
    const int tile_n = lid % tilesN;  // the w-tiles of one row panel run back to back
    const int tile_m = lid / tilesN;
    const int nn0 = tile_m * BM, n0 = tile_n * BN;
    const int nrows = p.H * p.Bt * p.C;
    const int Kf = p.W / 2 + 1;
    const int Np = p.ldt;  // table pitch (Kf rounded up to 4, zero padded): vector loads never straddle the end
    const long N2 = (long)p.Bt * 2 * p.C;
    const long ms = (long)p.H * N2;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int arow = wave * 32 + i;

    f32x16 P[2], Q[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[t][r] = 0.f; Q[t][r] = 0.f; }

    SA_t sa;
    SB_t sbc, sbs;
    sa.init(nn0, nrows, p.C, tid);
    const int nk = (p.Mm + BK - 1) / BK;
    sa.load(p.spec, ms, p.C, p.Mm, 0, tid);
    sbc.load(p.tc, p.ldt, nullptr, 0, -1, nullptr, nullptr, n0, Np, 0, p.Mm, tid);
    sbs.load(p.ts, p.ldt, nullptr, 0, -1, nullptr, nullptr, n0, Np, 0, p.Mm, tid);
    sa.store(As, p.Mm, 0, tid);
    sbc.store(Bs, n0, Np, 0, p.Mm, tid);
    sbs.store(Bs + BK * SB, n0, Np, 0, p.Mm, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
            const int k0 = (kt + 1) * BK;
            sa.load(p.spec, ms, p.C, p.Mm, k0, tid);
            sbc.load(p.tc, p.ldt, nullptr, 0, -1, nullptr, nullptr, n0, Np, k0, p.Mm, tid);
            sbs.load(p.ts, p.ldt, nullptr, 0, -1, nullptr, nullptr, n0, Np, k0, p.Mm, tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* Ar = As + cur * ABUF;
            const float* Ai = Ar + BK * SA;
            const float* Bc = Bs + cur * BBUF;
            const float* Bsn = Bc + BK * SB;
#pragma unroll
            for (int ks = 0; ks < BK; ks += 2) {
                const float ar = Ar[(ks + h) * SA + arow];
                const float ai = Ai[(ks + h) * SA + arow];
                const float c0 = Bc[(ks + h) * SB + i];
                const float c1 = Bc[(ks + h) * SB + i + 32];
                const float s0 = Bsn[(ks + h) * SB + i];
                const float s1 = Bsn[(ks + h) * SB + i + 32];
                P[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, c0, P[0], 0, 0, 0);
                P[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, c1, P[1], 0, 0, 0);
                Q[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, s0, Q[0], 0, 0, 0);
                Q[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, s1, Q[1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            const int k0 = (kt + 1) * BK;
            sa.store(As + (cur ^ 1) * ABUF, p.Mm, k0, tid);
            sbc.store(Bs + (cur ^ 1) * BBUF, n0, Np, k0, p.Mm, tid);
            sbs.store(Bs + (cur ^ 1) * BBUF + BK * SB, n0, Np, k0, p.Mm, tid);
        }
        __syncthreads();
    }

    const int w0c = n0 + i, w1c = n0 + 32 + i;
    float vmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nn = nn0 + wave * 32 + acc_row(r, h);
        if (nn >= nrows) continue;
        const int c = nn % p.C, kb = nn / p.C;
        const int b = kb % p.Bt, k = kb / p.Bt;
        float* yrow = p.y + ((long)(b * p.C + c) * p.H + k) * p.W;
        const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int w = tn == 0 ? w0c : w1c;
            if (w >= Kf) continue;
            const float pv = P[tn][r], qv = Q[tn][r];
            const float ya = (pv + qv) + bv;
            yrow[w] = ya;
            vmax = fmaxf(vmax, fabsf(ya));
            if (w != 0 && 2 * w != p.W) {
                const float yb = (pv - qv) + bv;
                yrow[p.W - w] = yb;
                vmax = fmaxf(vmax, fabsf(yb));
            }
        }
    }
    if (p.omax) {   // one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads();
        if ((tid & 63) == 0) smem[tid >> 6] = vmax;
        __syncthreads();
        if (tid == 0) atomicMax(p.omax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]))));
    }
}

hipError_t launch_dft_inverse(const DftArgs& a, hipStream_t s) {
    hipError_t ferr = hipSuccess;
    if (launch_dft_inverse_fft(a, s, &ferr)) return ferr;
    constexpr int BM = 128, BN = 64;
    const int nrows = a.H * a.Bt * a.C;
    const int Kf = a.W / 2 + 1;
    const int tilesM = (nrows + BM - 1) / BM, tilesN = (Kf + BN - 1) / BN;
    const bool vs = al16(a.spec) && (a.C % 4 == 0);
    dim3 grid((unsigned)(tilesM * tilesN)), block(256);
    if (vs)
        hipLaunchKernelGGL((dft_inverse_kernel<true>), grid, block, 0, s, a, tilesM, tilesN);
    else
        hipLaunchKernelGGL((dft_inverse_kernel<false>), grid, block, 0, s, a, tilesM, tilesN);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// instance-norm statistics (nn.InstanceNorm2d eps=1e-6 affine, sfnonet.py:593-601): one workgroup per
// (b, c) plane, fp64 accumulation, emits the affine (scale, shift) that consumers apply on load.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void instnorm_stats_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int C,
                                                             long HW, float* __restrict__ scale,
                                                             float* __restrict__ shift, unsigned* omax) {
    const int plane = blockIdx.x;
    const float* xp = x + (long)plane * HW;
    double s = 0.0, ss = 0.0;
    float lo = 3.0e38f, hi = -3.0e38f;
    const bool vec = ((reinterpret_cast<uintptr_t>(xp) & 15) == 0) && (HW % 4 == 0);
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        const long n4 = HW / 4;
        for (long j = threadIdx.x; j < n4; j += blockDim.x) {
            const float4 v = x4[j];
            lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
            hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            const float ps = (v.x + v.y) + (v.z + v.w);
            const float pq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
            s += (double)ps;
            ss += (double)pq;
        }
    } else {
        for (long j = threadIdx.x; j < HW; j += blockDim.x) {
            const double v = (double)xp[j];
            lo = fminf(lo, xp[j]);
            hi = fmaxf(hi, xp[j]);
            s += v;
            ss += v * v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
        lo = fminf(lo, __shfl_down(lo, off, 64));
        hi = fmaxf(hi, __shfl_down(hi, off, 64));
    }
    __shared__ double red[2][8];
    __shared__ float redm[2][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; redm[0][wave] = lo; redm[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tss = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            ts += red[0][w]; tss += red[1][w];
            lo = fminf(lo, redm[0][w]); hi = fmaxf(hi, redm[1][w]);
        }
        const double mean = ts / (double)HW;
        double var = tss / (double)HW - mean * mean;  // biased variance
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const int c = plane % C;
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double bt = beta ? (double)beta[c] : 0.0;
        const double sc = g * rstd;
        scale[plane] = (float)sc;
        shift[plane] = (float)(bt - mean * sc);
        if (omax) {  // bound of |x * scale + shift| over the plane: consumers of the normalised field scale by it
            const float a = (float)sc, b = (float)(bt - mean * sc);
            const float bound = fmaxf(fabsf(fmaf(lo, a, b)), fabsf(fmaf(hi, a, b)));
            atomicMax(omax + (plane & 63), __float_as_uint(bound));
        }
    }
}

// Second half of the fused instance norm: reduce the per-strip row statistics written by the producing GEMM's epilogue
// (Gemm4Args::part) to the per-(sample, channel) affine, in fp64 and in a fixed order (deterministic).
// The bound of the normalised plane published in `omax` is |a| sqrt(HW var) + |beta| >= max |a (x - mean) + beta| (no sample
// lies further from the mean than sqrt(sum (x - mean)^2)): it needs no running minimum / maximum in the producers' epilogues
// (12 % of their vector instructions).  It is up to sqrt(HW) / (max |x - mean| / sigma) - 2^5 .. 2^6 at 180 x 360 - looser than
// the true maximum; the hi / lo operands keep their 22 bits while a bound is within 2^14 of the maximum.  The
// (min, max) fields of a record are ignored.
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float4* __restrict__ part, int nparts, int C,
                                                                long HW, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                float* __restrict__ scale, float* __restrict__ shift,
                                                                unsigned* omax) {
    // one workgroup per 4 channels: thread = (channel sub-index, strip residue of 64); strips are summed thread-strided
    // in fp64, then across the residues with a fixed shuffle tree and a fixed order over the 4 waves (deterministic).
    // With the 2025 partials per channel that fc2 writes (one per 32-pixel tile) it takes 10 us, with the 256 of the inner skip
    // 5: its 64-byte reads are 6 KiB apart, one DRAM page each.  r03 same-box: 1024 threads per workgroup 11.2 us, eight
    // range-checked loads in flight per thread 16.7 us - neither helps; fewer partials would.
    const int b = blockIdx.y;
    const int cs = threadIdx.x & 3, pr = threadIdx.x >> 2;   // 0..3, 0..63
    const int c = blockIdx.x * 4 + cs;
    double s = 0.0, ss = 0.0;
    float lo = 3.0e38f, hi = -3.0e38f;
    if (c < C) {
        const float4* base = part + (long)b * nparts * C + c;
#pragma unroll 4
        for (int p = pr; p < nparts; p += 64) {
            const float4 v = base[(long)p * C];
            s += (double)v.x;
            ss += (double)v.y;
            lo = fminf(lo, v.z);
            hi = fmaxf(hi, v.w);
        }
    }
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {   // lanes of this wave with the same channel sub-index
        s += __shfl_xor(s, off, 64);
        ss += __shfl_xor(ss, off, 64);
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    __shared__ double rs[4][4], rss[4][4];
    __shared__ float rlo[4][4], rhi[4][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 4) { rs[wave][lane] = s; rss[wave][lane] = ss; rlo[wave][lane] = lo; rhi[wave][lane] = hi; }
    __syncthreads();
    if (threadIdx.x < 4 && c < C) {
        s = rs[0][cs]; ss = rss[0][cs]; lo = rlo[0][cs]; hi = rhi[0][cs];
        for (int k = 1; k < 4; ++k) {
            s += rs[k][cs]; ss += rss[k][cs];
            lo = fminf(lo, rlo[k][cs]); hi = fmaxf(hi, rhi[k][cs]);
        }
        const double mean = s / (double)HW;
        double var = ss / (double)HW - mean * mean;  // biased variance
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double bt = beta ? (double)beta[c] : 0.0;
        const double sc = g * rstd;
        const float a = (float)sc, bb = (float)(bt - mean * sc);
        scale[(long)b * C + c] = a;
        shift[(long)b * C + c] = bb;
        (void)lo; (void)hi;
        if (omax) atomicMax(omax + ((b * C + c) & 63), __float_as_uint((float)(fabs(sc) * sqrt((double)HW * var) * (1.0 + 1e-6) + fabs(bt))));
    }
}
hipError_t launch_instnorm_finalize(const float4* part, int nparts, int Bt, int C, long HW, const float* gamma,
                                    const float* beta, float eps, float* scale, float* shift, unsigned* omax,
                                    hipStream_t s) {
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)((C + 3) / 4), (unsigned)Bt), dim3(256), 0, s, part, nparts,
                       C, HW, gamma, beta, eps, scale, shift, omax);
    return hipGetLastError();
}

// Instance-norm affine folded into the conv that consumes the normalised tensor, for the packed-operand engine:
//   Wf[o][i] = W[o][i] * a[i]   -> fp16 hi/lo planes in A-tile order (split_f16_tiled layout), scaled by 2^(12 - exponent(bound))
//   bf[o]    = bias[o] + sum_i W[o][i] * b[i]
// bound = max|W| * max|a| is published to `wslot` (the GEMM derives the same scale from it).  One workgroup per 16 output
// rows; per sample.
__global__ __launch_bounds__(256) void fold_affine_f16_kernel(const float* __restrict__ W, long ldw, float wmax,
                                                              const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ bias, _Float16* __restrict__ hi,
                                                              _Float16* __restrict__ lo, float* __restrict__ bf, int O,
                                                              int I, long ldd, long sPl, unsigned* wslot) {
    // workgroup = (16 output rows, one 32-column stage, sample): 512 plane elements, two per thread
    const int rb = blockIdx.x, st = blockIdx.y, smp = blockIdx.z;
    const float* as = a + (long)smp * I;
    const float* bs = b + (long)smp * I;
    __shared__ float red[4];
    float am = 0.f;
    for (int i = threadIdx.x; i < I * (int)gridDim.z; i += 256) am = fmaxf(am, fabsf(a[i]));   // over ALL samples: one scale
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float bound = wmax * am;
    const float scale = ldexpf(1.0f, pow2_exponent_for(bound));
    if (rb == 0 && st == 0 && threadIdx.x == 0) atomicMax(wslot + (smp & 63), __float_as_uint(bound));
    const long nst = ldd / 32;
    _Float16* ph = hi + (long)smp * sPl + ((long)rb * nst + st) * 512;
    _Float16* pl = lo + (long)smp * sPl + ((long)rb * nst + st) * 512;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = threadIdx.x + 256 * u;
        const int e = t & 7, ps = (t >> 3) & 3, r = t >> 5;
        const long row = (long)rb * 16 + r;
        const int ls = ps ^ ((int)(row >> 2) & 3);
        const int c = st * 32 + ls * 8 + e;
        float x = 0.f;
        if (row < O && c < I) x = __builtin_amdgcn_fmed3f(W[row * ldw + c] * as[c] * scale, -65504.f, 65504.f);
        const _Float16 h = (_Float16)x;
        ph[t] = h;
        pl[t] = (_Float16)(x - (float)h);
    }
    if (st == 0) {  // folded bias: 16 rows, 16 threads per row
        const int r = threadIdx.x >> 4, l16 = threadIdx.x & 15;
        const long row = (long)rb * 16 + r;
        float acc = 0.f;
        if (row < O)
            for (int i = l16; i < I; i += 16) acc = fmaf(W[row * ldw + i], bs[i], acc);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (l16 == 0 && row < O) bf[(long)smp * O + row] = (bias ? bias[row] : 0.f) + acc;
    }
}
hipError_t launch_fold_affine_f16(const float* W, long ldw, float wmax, const float* a, const float* b, const float* bias,
                                  void* hi, void* lo, float* bf, int nsamples, int O, int I, long ldd, long sPl,
                                  unsigned* wslot, hipStream_t s) {
    if (ldd % 32 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_affine_f16_kernel, dim3((unsigned)((O + 15) / 16), (unsigned)(ldd / 32), (unsigned)nsamples), dim3(256), 0, s, W, ldw,
                       wmax, a, b, bias, static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), bf, O, I, ldd, sPl, wslot);
    return hipGetLastError();
}

hipError_t launch_instnorm_stats(const float* x, const float* gamma, const float* beta, float eps, int Bt, int C,
                                 long HW, float* scale, float* shift, hipStream_t s, unsigned* omax) {
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3((unsigned)(Bt * C)), dim3(512), 0, s, x, gamma, beta, eps, C, HW,
                       scale, shift, omax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// layout converters (API boundary only; the network path never leaves the internal layout)
// ---------------------------------------------------------------------------------------------
// Both directions are a tiled transpose through LDS between the internal matrix [r = (l, m)][b][ri][c] (c fastest) and the
// reference's [b][c][r][ri] (complex64, r fastest): a 32 x 32 (r, c) tile is read and written in 128 / 256-byte runs (the
// first version walked one side with a stride of 2 C Bt floats per lane: the reference's own `sht` benchmark, 1024 fields,
// spent 1.1 of its 1.5 ms here).  TO_REF: entries with m > l are written as zeros without reading the internal buffer (the
// triangular Legendre stage never writes them), so the standalone transform needs no memset of its scratch.
// omax (from-reference direction, optional): atomicMax of the bit pattern of max |value| over the launch - the range of the
// coefficients for the f16x3 Legendre stage, taken here instead of in a pass of its own over the converted tensor.
template <bool TO_REF>
__global__ __launch_bounds__(256) void spec_layout_kernel(const float* __restrict__ src, float* __restrict__ dst, int Bt, int C, int L, int Mm,
                                                          unsigned* omax, int tri) {
    __shared__ float tile[2][32][33];
    const long R = (long)L * Mm;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const long r0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, b = blockIdx.z;
    if (TO_REF) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {          // rows r0 + ty + 8 k, channel c0 + tx: 128-byte runs over c
            const long r = r0 + ty + 8 * k;
            const int c = c0 + tx;
            float re = 0.f, im = 0.f;
            if (r < R && c < C && (int)(r % Mm) <= (int)(r / Mm)) {
                const float* p = src + (r * Bt + b) * 2 * C + c;
                re = p[0];
                im = p[C];
            }
            tile[0][ty + 8 * k][tx] = re;
            tile[1][ty + 8 * k][tx] = im;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {          // channel c0 + ty + 8 k, row r0 + tx: (re, im) pairs, 256-byte runs over r
            const int c = c0 + ty + 8 * k;
            const long r = r0 + tx;
            if (r < R && c < C)
                *reinterpret_cast<float2*>(dst + (((long)b * C + c) * R + r) * 2) = make_float2(tile[0][tx][ty + 8 * k], tile[1][tx][ty + 8 * k]);
        }
    } else {
        // Entries with m > l are never used by a synthesis (their Legendre values are zero): they are not read.  `tri` < 0: the
        // consumer is one of the exactly triangular strip kernels, which never touch them - a tile that lies wholly above the
        // diagonal is skipped, the others leave them unwritten; otherwise they are written as zeros (the tile engines read whole
        // k-tiles of the scratch and 0 x garbage is not 0).  The range maximum is over the used entries only.
        const bool exact = tri < 0;
        {
            const long l0 = r0 / Mm, m0 = r0 % Mm;
            if (exact && m0 > l0 && m0 + 32 <= Mm) return;   // 32 consecutive r of one degree, all above the diagonal
        }
        float vmax = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k;
            const long r = r0 + tx;
            float2 v = make_float2(0.f, 0.f);
            if (r < R && c < C && (int)(r % Mm) <= (int)(r / Mm)) v = *reinterpret_cast<const float2*>(src + (((long)b * C + c) * R + r) * 2);
            tile[0][tx][ty + 8 * k] = v.x;
            tile[1][tx][ty + 8 * k] = v.y;
            vmax = fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
        if (omax) {   // one atomic per workgroup
            __shared__ float red[4];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
            __syncthreads();
            if (threadIdx.x == 0) {   // 32 k workgroups at the reference's benchmark size: the atomic only where it would raise the slot (64
                const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));   // words in one line - unconditional, they serialise to 70 us)
                unsigned* slot = omax + ((blockIdx.x + blockIdx.y) & (AMAX_SHARDS - 1));
                if (__float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(m));
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long r = r0 + ty + 8 * k;
            const int c = c0 + tx;
            if (r < R && c < C && (!exact || (int)(r % Mm) <= (int)(r / Mm))) {
                float* p = dst + (r * Bt + b) * 2 * C + c;
                p[0] = tile[0][ty + 8 * k][tx];
                p[C] = tile[1][ty + 8 * k][tx];
            }
        }
    }
}
static inline unsigned grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}
hipError_t launch_spec_to_ref(const float* D, float* out, int Bt, int C, int L, int Mm, hipStream_t s) {
    const long R = (long)L * Mm;
    if ((R + 31) / 32 > 0x7fffffffL || (C + 31) / 32 > 65535 || Bt > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(spec_layout_kernel<true>, dim3((unsigned)((R + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)Bt), dim3(256), 0, s, D, out, Bt, C, L, Mm,
                       static_cast<unsigned*>(nullptr), 0);
    return hipGetLastError();
}
hipError_t launch_ref_to_spec(const float* in, float* E, int Bt, int C, int L, int Mm, hipStream_t s, unsigned* omax, bool triangular_consumer) {
    const long R = (long)L * Mm;
    if ((R + 31) / 32 > 0x7fffffffL || (C + 31) / 32 > 65535 || Bt > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(spec_layout_kernel<false>, dim3((unsigned)((R + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)Bt), dim3(256), 0, s, in, E, Bt, C, L, Mm, omax, triangular_consumer ? -1 : 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// "diagonal" operator: per-(l, m) Cin x Cout complex mat-vec (contractions.py:169-180).  Pure weight
// bandwidth (the weight is Cin*Cout*L*M complex); only used by small nets and three of the goldens.
// ---------------------------------------------------------------------------------------------
__global__ void contract_diagonal_kernel(const float* __restrict__ D, const float* __restrict__ w,
                                         float* __restrict__ E, int Bt, int Cin, int Cout, int L, int Mm,
                                         unsigned* omax) {
    const long total = (long)L * Mm * Bt * Cout;
    float vmax = 0.f;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int m = t % Mm;  // m fastest: weight reads coalesce along m
        long q = t / Mm;
        const int l = q % L;
        q /= L;
        const int o = q % Cout;
        const int b = q / Cout;
        const float* xin = D + (((long)l * Mm + m) * Bt + b) * 2 * Cin;
        float re = 0.f, im = 0.f;
        for (int i = 0; i < (m <= l ? Cin : 0); ++i) {  // coefficients with m > l are identically zero
            const float xr = xin[i], xi = xin[Cin + i];
            const float2 wv = *reinterpret_cast<const float2*>(w + ((((long)i * Cout + o) * L + l) * Mm + m) * 2);
            re = fmaf(xr, wv.x, re);
            re = fmaf(-xi, wv.y, re);
            im = fmaf(xr, wv.y, im);
            im = fmaf(xi, wv.x, im);
        }
        float* eo = E + (((long)l * Mm + m) * Bt + b) * 2 * Cout;
        eo[o] = re;
        eo[Cout + o] = im;
        vmax = fmaxf(vmax, fmaxf(fabsf(re), fabsf(im)));
    }
    if (omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(omax + (blockIdx.x & 63), __float_as_uint(vmax));
    }
}
hipError_t launch_contract_diagonal(const float* D, const float* w, float* E, int Bt, int Cin, int Cout, int L, int Mm,
                                    hipStream_t s, unsigned* omax) {
    const long total = (long)L * Mm * Bt * Cout;
    hipLaunchKernelGGL(contract_diagonal_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, D, w, E, Bt, Cin, Cout, L,
                       Mm, omax);
    return hipGetLastError();
}

// dhconv weight (Cin, Cout, L, 2) -> per-l real 2Cin x 2Cout matrix acting on planar (re | im) vectors:
//   [ out_re | out_im ] = [ x_re | x_im ] * [[ w_re, w_im ], [ -w_im, w_re ]]
__global__ void expand_dhconv_weight_kernel(const float* __restrict__ w, float* __restrict__ wx, int Cin, int Cout,
                                            int L) {
    const long total = (long)L * Cin * Cout;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int o = t % Cout;
        long q = t / Cout;
        const int i = q % Cin;
        const int l = q / Cin;
        const float2 wv = *reinterpret_cast<const float2*>(w + (((long)i * Cout + o) * L + l) * 2);
        float* base = wx + (long)l * (2 * Cin) * (2 * Cout);
        base[(long)i * 2 * Cout + o] = wv.x;
        base[(long)i * 2 * Cout + Cout + o] = wv.y;
        base[(long)(Cin + i) * 2 * Cout + o] = -wv.y;
        base[(long)(Cin + i) * 2 * Cout + Cout + o] = wv.x;
    }
}
hipError_t launch_expand_dhconv_weight(const float* w, float* wx, int Cin, int Cout, int L, hipStream_t s) {
    const long total = (long)L * Cin * Cout;
    hipLaunchKernelGGL(expand_dhconv_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, w, wx, Cin, Cout, L);
    return hipGetLastError();
}

// nn.LayerNorm over the two spatial dimensions, as the reference's "layer_norm" normalisation builds it (sfnonet.py:584-592:
// normalized_shape = (H, W), eps 1e-6, elementwise affine of shape (H, W) shared by all channels): per (sample, channel) plane
//     y[p] = (x[p] - mean) / sqrt(var + eps) * gamma[p] + beta[p]        (biased variance, fp64 sums in a fixed order)
// One workgroup per plane, two passes (the second one's reads come from L2); in place allowed.  omax (optional): atomicMax of the bit
// pattern of max |y| - the range slot of the packed convolutions that read y.
__global__ __launch_bounds__(512) void spatial_layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, long HW, float* __restrict__ y,
                                                                 unsigned* omax) {
    const float* xp = x + (long)blockIdx.x * HW;
    float* yp = y + (long)blockIdx.x * HW;
    double s = 0.0, ss = 0.0;
    for (long j = threadIdx.x; j < HW; j += blockDim.x) {
        const double v = (double)xp[j];
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
    }
    __shared__ double red[2][8];
    __shared__ float stat[2], redm[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tss = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += red[0][w]; tss += red[1][w]; }
        const double mean = ts / (double)HW;
        double var = tss / (double)HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        stat[0] = (float)rstd;
        stat[1] = (float)(-mean * rstd);
    }
    __syncthreads();
    const float a = stat[0], b = stat[1];
    float vmax = 0.f;
    for (long j = threadIdx.x; j < HW; j += blockDim.x) {
        const float v = fmaf(fmaf(xp[j], a, b), gamma ? gamma[j] : 1.f, beta ? beta[j] : 0.f);
        yp[j] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    if (omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
        if (lane == 0) redm[wave] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, redm[w]);
            atomicMax(omax + (blockIdx.x & (AMAX_SHARDS - 1)), __float_as_uint(m));
        }
    }
}
hipError_t launch_spatial_layer_norm(const float* x, const float* gamma, const float* beta, float eps, long planes, long HW, float* y,
                                     unsigned* omax, hipStream_t s) {
    if (planes < 1 || planes > 0x7fffffffL || HW < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(spatial_layer_norm_kernel, dim3((unsigned)planes), dim3(512), 0, s, x, gamma, beta, eps, HW, y, omax);
    return hipGetLastError();
}

__global__ void rowaffine_add_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                     const float* __restrict__ sh, const float* __restrict__ r,
                                     const float* __restrict__ rsc, const float* __restrict__ rsh,
                                     float* __restrict__ y, long rows, long n) {
    const long total = rows * n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long row = t / n;
        float v = x[t];
        if (sc) v = fmaf(v, sc[row], sh[row]);
        if (r) {
            float rv = r[t];
            if (rsc) rv = fmaf(rv, rsc[row], rsh[row]);
            v += rv;
        }
        y[t] = v;
    }
}
hipError_t launch_rowaffine_add(const float* x, const float* sc, const float* sh, const float* r, const float* rsc,
                                const float* rsh, float* y, long rows, long n, hipStream_t s) {
    hipLaunchKernelGGL(rowaffine_add_kernel, dim3(grid_for(rows * n, 256)), dim3(256), 0, s, x, sc, sh, r, rsc, rsh, y,
                       rows, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// stepper glue: gather + normalise into the packed network input, scatter + de-normalise the output
// ---------------------------------------------------------------------------------------------
__global__ void pack_normalize_kernel(const float* const* __restrict__ srcs, const long* __restrict__ strides,
                                      const float* __restrict__ mean, const float* __restrict__ stdv,
                                      float* __restrict__ dst, int nch, long HW) {
    const int j = blockIdx.y, b = blockIdx.z;
    const float* src = srcs[j] + (long)b * strides[j];
    float* d = dst + ((long)b * nch + j) * HW;
    const float mu = mean[j], sd = stdv[j];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < HW; t += (long)gridDim.x * blockDim.x)
        d[t] = __fdiv_rn(__fsub_rn(src[t], mu), sd);  // normalizer.py:221: (t - means[k]) / stds[k], unfused
}
__global__ void unpack_denormalize_kernel(const float* __restrict__ src, const float* __restrict__ mean,
                                          const float* __restrict__ stdv, float* const* __restrict__ dsts,
                                          const long* __restrict__ strides, int nch, long HW) {
    const int j = blockIdx.y, b = blockIdx.z;
    const float* s = src + ((long)b * nch + j) * HW;
    float* d = dsts[j] + (long)b * strides[j];
    const float mu = mean[j], sd = stdv[j];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < HW; t += (long)gridDim.x * blockDim.x)
        d[t] = __fadd_rn(__fmul_rn(s[t], sd), mu);  // normalizer.py:236: t * stds[k] + means[k], two roundings
}
hipError_t launch_pack_normalize(const float* const* srcs, const long* strides, const float* mean, const float* stdv,
                                 float* dst, int Bt, int nch, long HW, hipStream_t s) {
    unsigned gx = (unsigned)((HW + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pack_normalize_kernel, dim3(gx, nch, Bt), dim3(256), 0, s, srcs, strides, mean, stdv, dst, nch,
                       HW);
    return hipGetLastError();
}
hipError_t launch_unpack_denormalize(const float* src, const float* mean, const float* stdv, float* const* dsts,
                                     const long* strides, int Bt, int nch, long HW, hipStream_t s) {
    unsigned gx = (unsigned)((HW + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(unpack_denormalize_kernel, dim3(gx, nch, Bt), dim3(256), 0, s, src, mean, stdv, dsts, strides,
                       nch, HW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fold a per-(sample, input-channel) affine into a 1x1-conv weight:  W (a x + b) + bias = (W diag(a)) x + (W b + bias)
//   Wf[s][o][i] = W[o][i] * a[s][i]  (columns i in [I, ldw) are zero),   bf[s][o] = bias[o] + sum_i W[o][i] * b[s][i]
// This is how the instance norm (sfnonet.py:218-221, 234-238) reaches the direct-to-LDS GEMM, which cannot touch its
// B operand: ~0.6-1.2 MB of work per conv instead of a 100 MB normalised activation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void fold_affine_kernel(const float* __restrict__ W, long ldw,
                                                          const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ bias, float* __restrict__ Wf,
                                                          float* __restrict__ bf, int O, int I) {
    const int o = blockIdx.x, smp = blockIdx.y;
    const float* wr = W + (long)o * ldw;
    float* wo = Wf + ((long)smp * O + o) * ldw;
    const float* as = a + (long)smp * I;
    const float* bs = b + (long)smp * I;
    double acc = 0.0;
    for (int i = threadIdx.x; i < (int)ldw; i += blockDim.x) {
        float v = 0.f;
        if (i < I) {
            const float w = wr[i];
            v = w * as[i];
            acc += (double)w * (double)bs[i];
        }
        wo[i] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double red[2];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bf[(long)smp * O + o] = (float)((bias ? (double)bias[o] : 0.0) + red[0] + red[1]);
}
hipError_t launch_fold_affine(const float* W, long ldw, const float* a, const float* b, const float* bias, float* Wf,
                              float* bf, int nsamples, int O, int I, hipStream_t s) {
    hipLaunchKernelGGL(fold_affine_kernel, dim3(O, nsamples), dim3(128), 0, s, W, ldw, a, b, bias, Wf, bf, O, I);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Conditional layer norm of the NoiseConditionedSFNO (fme/core/models/conditional_sfno/layers.py:95-141, 245-318):
//   y[c][p] = ((x[c][p] - mean_p) * rstd_p * gamma_c + beta_c) * (1 + sum_j Ws[c][j] noise[j][p]) + sum_j Wb[c][j] noise[j][p]
// with per-PIXEL statistics over the channels (biased variance, eps inside the sqrt).  Two kernels:
//   cln_stats_kernel : mean / rstd per pixel; 8 channel groups x 64 pixel quads per workgroup, fp64 accumulation of
//                      sum and sum of squares (one pass over x), reduced through LDS
//   cln_apply_kernel : thread = 8 channels x 4 pixels; the two 1x1 "noise" convolutions are computed on the fly
//                      (2 J FMAs per element; their weights for the 8 channels sit in LDS), fp32 output + max|y|
// ---------------------------------------------------------------------------------------------
// V pixels per lane: 4 (16-byte accesses, H W % 4 == 0) or 1 (any field size, e.g. the 9 x 18 grids of the reference's own goldens)
template <int V>
struct PixVec {
    float v[V];
    __device__ __forceinline__ void load(const float* p) {
        if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else v[0] = *p;
    }
    __device__ __forceinline__ void store(float* p) const {
        if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else *p = v[0];
    }
};

template <int V>
__global__ __launch_bounds__(512) void cln_stats_kernel(const float* __restrict__ x, int C, long HW, float eps,
                                                        float* __restrict__ mean, float* __restrict__ rstd) {
    const int b = blockIdx.y;
    const int q = threadIdx.x & 63, cg = threadIdx.x >> 6;      // pixel group, channel group (0..7)
    const long p = ((long)blockIdx.x * 64 + q) * V;
    const bool ok = p < HW;                                     // HW % V == 0
    const float* xb = x + (long)b * C * HW + (ok ? p : 0);
    double s[V], ss[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s[e] = 0.0; ss[e] = 0.0; }
    for (int c = cg; c < C; c += 8) {
        PixVec<V> xv;
        xv.load(xb + (long)c * HW);
#pragma unroll
        for (int e = 0; e < V; ++e) { s[e] += (double)xv.v[e]; ss[e] += (double)xv.v[e] * (double)xv.v[e]; }
    }
    __shared__ double rs[8][64][V], rss[8][64][V];
#pragma unroll
    for (int e = 0; e < V; ++e) { rs[cg][q][e] = s[e]; rss[cg][q][e] = ss[e]; }
    __syncthreads();
    if (cg == 0 && ok) {
        PixVec<V> m4, r4;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            double a = 0.0, bq = 0.0;
            for (int k = 0; k < 8; ++k) { a += rs[k][q][e]; bq += rss[k][q][e]; }
            const double mu = a / C;
            double var = bq / C - mu * mu;
            if (var < 0.0) var = 0.0;
            m4.v[e] = (float)mu;
            r4.v[e] = (float)(1.0 / sqrt(var + (double)eps));
        }
        m4.store(mean + (long)b * HW + p);
        r4.store(rstd + (long)b * HW + p);
    }
}

template <int V>
__global__ __launch_bounds__(256) void cln_apply_kernel(const float* x, const float* __restrict__ noise,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ ws, const float* __restrict__ wb,
                                                        float* y, int C, int J, long HW, unsigned* omax) {   // y may alias x
    extern __shared__ float wsm[];                // [2][8][J]: scale / bias conv weights of this workgroup's 8 channels
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 8;
    for (int t = threadIdx.x; t < 2 * 8 * J; t += 256) {
        const int which = t / (8 * J), r = (t / J) % 8, j = t % J;
        const int c = c0 + r;
        const float* src = which == 0 ? ws : wb;
        wsm[t] = (src != nullptr && c < C) ? src[(long)c * J + j] : 0.f;
    }
    __syncthreads();
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    float vmax = 0.f;
    if (p < HW) {
        float sc[8][V], bi[8][V];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int e = 0; e < V; ++e) { sc[r][e] = 1.f; bi[r][e] = 0.f; }
        if (ws != nullptr) {
            const float* nb = noise + (long)b * J * HW + p;
            for (int j = 0; j < J; ++j) {
                PixVec<V> nv;
                nv.load(nb + (long)j * HW);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float a = wsm[r * J + j], bq = wsm[8 * J + r * J + j];
#pragma unroll
                    for (int e = 0; e < V; ++e) { sc[r][e] = fmaf(a, nv.v[e], sc[r][e]); bi[r][e] = fmaf(bq, nv.v[e], bi[r][e]); }
                }
            }
        }
        PixVec<V> mu, rs;
        mu.load(mean + (long)b * HW + p);
        rs.load(rstd + (long)b * HW + p);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c = c0 + r;
            if (c >= C) break;
            PixVec<V> xv, o;
            xv.load(x + ((long)b * C + c) * HW + p);
            const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float t = (xv.v[e] - mu.v[e]) * rs.v[e];
                if (gamma) t = t * g + bt;
                o.v[e] = t * sc[r][e] + bi[r][e];
                vmax = fmaxf(vmax, fabsf(o.v[e]));
            }
            o.store(y + ((long)b * C + c) * HW + p);
        }
    }
    if (omax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(omax + ((blockIdx.x + blockIdx.y) & 63), __float_as_uint(vmax));
    }
}

template <int V>
static void launch_cln(const float* x, const float* noise, const float* gamma, const float* beta, const float* ws, const float* wb,
                       float eps, float* mean, float* rstd, float* y, int Bt, int C, int Jm, long HW, hipStream_t s, unsigned* omax) {
    const long nv = HW / V;
    hipLaunchKernelGGL(cln_stats_kernel<V>, dim3((unsigned)((nv + 63) / 64), (unsigned)Bt), dim3(512), 0, s, x, C, HW, eps, mean, rstd);
    hipLaunchKernelGGL(cln_apply_kernel<V>, dim3((unsigned)((nv + 255) / 256), (unsigned)((C + 7) / 8), (unsigned)Bt), dim3(256),
                       (size_t)2 * 8 * Jm * sizeof(float), s, x, noise, mean, rstd, gamma, beta, ws, wb, y, C, Jm, HW, omax);
}

hipError_t launch_cond_layer_norm(const float* x, const float* noise, const float* gamma, const float* beta,
                                  const float* ws, const float* wb, float eps, float* stats, float* y, int Bt, int C,
                                  int J, long HW, hipStream_t s, unsigned* omax) {
    if (ws && (!noise || J <= 0 || J > 512)) return hipErrorInvalidValue;
    float* mean = stats;
    float* rstd = stats + (long)Bt * HW;
    const int Jm = ws ? J : 1;
    const bool vec = HW % 4 == 0 && al16(x) && al16(y) && al16(stats) && (!ws || al16(noise));
    if (vec) launch_cln<4>(x, noise, gamma, beta, ws, wb, eps, mean, rstd, y, Bt, C, Jm, HW, s, omax);
    else launch_cln<1>(x, noise, gamma, beta, ws, wb, eps, mean, rstd, y, Bt, C, Jm, HW, s, omax);
    return hipGetLastError();
}

// slot reset as a kernel (not hipMemsetAsync): inside a captured graph a memset node is not ordered/coherent with the
// device-scope atomics on the same words the way a kernel node is (observed: stale bounds on replay)
__global__ void zero_u32_kernel(unsigned* p, long n) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) __hip_atomic_store(p + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t launch_zero_u32(unsigned* p, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
    return hipGetLastError();
}

// max |x| into 64 shards: 16-byte loads when the base is aligned, one atomic per workgroup (r03: one per wave from 1400 workgroups
// of 4-byte loads took 29 us on the 11 MB network input - the atomics, not the bytes)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* omax) {
    __shared__ float red[4];
    float m = 0.f;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long t = tid; t < n4; t += nth) {
            const float4 v = x4[t];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (long t = (n4 << 2) + tid; t < n; t += nth) m = fmaxf(m, fabsf(x[t]));
    } else {
        for (long t = tid; t < n; t += nth) m = fmaxf(m, fabsf(x[t]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(omax + (blockIdx.x & 63), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
hipError_t launch_absmax(const float* x, long n, unsigned* omax, hipStream_t s) {
    long blocks = (n + 256 * 16 - 1) / (256 * 16);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, omax);
    return hipGetLastError();
}

}  // namespace ace
