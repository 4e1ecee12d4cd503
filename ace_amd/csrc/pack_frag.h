// One workgroup's share of the A-fragment packing of a 1x1-convolution weight (pack_conv_frag_kernel, conv_ws.hip), as a device
// function: the stand-alone kernel calls it, and so do RIDER workgroups of the forward longitude FFT (fft.hip) - the folded weights
// of a block's inner skip depend only on norm0's affine, which exists before the spectral chain starts, so the 300 tiny workgroups
// that pack them ride in the first kernel of that chain instead of costing a dependent 7 us launch in front of the convolution.
#pragma once
#include <hip/hip_runtime.h>

#include "strip_common.h"

namespace ace {

#if defined(__HIPCC__)
// blk: workgroup index within the sample; 256 threads (tid < 256; further threads of the calling workgroup only take part in the
// barrier); `red`: 4 floats of LDS
__device__ __forceinline__ void pack_frag_block(const PackFragArgs& q, const int blk, const int smp, const int tid, float* red) {
    const int O = q.O, I = q.I;
    const bool active = tid < 256;
    if (blk >= (O / 32) * (I / 16)) {   // folded bias of one 32-row tile, bias + W b (8 threads per row, 16 bytes per load)
        if (!active) return;
        const int T = blk - (O / 32) * (I / 16);
        const int r = tid >> 3, l8 = tid & 7;
        const long row = 32L * T + r;
        const float4* w4 = reinterpret_cast<const float4*>(q.W + row * q.ldw);
        const float4* b4 = reinterpret_cast<const float4*>(q.b + (long)smp * I);
        float acc = 0.f;
        for (int k = l8; k < I / 4; k += 8) {
            const float4 w = w4[k], v = b4[k];
            acc = fmaf(w.x, v.x, acc); acc = fmaf(w.y, v.y, acc); acc = fmaf(w.z, v.z, acc); acc = fmaf(w.w, v.w, acc);
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (l8 == 0) q.bf[(long)smp * O + row] = (q.bias ? q.bias[row] : 0.f) + acc;
        return;
    }
    float scale = q.scale_static;
    if (q.a) {   // one scale for all samples (as fold_affine_f16_kernel)
        float am = 0.f;
        if (active)
            for (int k = tid; k < I * q.nsamples; k += 256) am = fmaxf(am, fabsf(q.a[k]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
        if (active && (tid & 63) == 0) red[tid >> 6] = am;
        __syncthreads();
        am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float bound = q.wmax * am;
        scale = ldexpf(1.0f, pow2_exponent_for(bound));
        if (blk == 0 && tid == 0) atomicMax(q.wslot + (smp & 63), __float_as_uint(bound));
    }
    if (!active) return;
    const int nJ = I / 16, nT = O / 32;
    const int T = blk / nJ, J = blk % nJ;         // one workgroup = one (T, J) block: 512 elements, two per thread
    const long bidx = q.order == 0 ? (long)T * nJ + J : (long)J * nT + T;
    _Float16* out = q.dst + (long)smp * q.sDst + bidx * 1024;
    const float* as = q.a ? q.a + (long)smp * I : nullptr;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + 256 * u;              // = lane * 8 + e
        const int e = t & 7, lane = t >> 3, i = lane & 31, g = lane >> 5;
        const int row = 32 * T + i, col = 16 * J + 8 * g + e;
        float x = q.W[(long)row * q.ldw + col] * scale;
        if (as) x *= as[col];
        x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
        const _Float16 h = (_Float16)x;
        out[t] = h;
        out[512 + t] = (_Float16)(x - (float)h);
    }
}
#endif

}  // namespace ace
