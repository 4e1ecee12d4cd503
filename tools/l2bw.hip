// L2 -> CU fill-rate probe for gfx950: every workgroup streams a small (L2-resident) or large (HBM) region
// either with global_load_lds_dwordx4 (LDS-DMA) or with global_load_dwordx4 into VGPRs.
// build: hipcc --offload-arch=gfx950 -O3 -o l2bw tools/l2bw.hip ; run: ./l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) char* lds_cp;
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_base_uniform) {
    unsigned keep;
    const unsigned addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cp)lds_base_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(addr) : "memory");
}

// region_floats: size of the region each XCD-group of workgroups cycles through
template <int MODE>  // 0 = LDS-DMA, 1 = VGPR loads
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, long region_floats, int iters, float* sink, int share) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // `share` workgroups (consecutive ids on one XCD: ids congruent mod 8) stream the SAME addresses
    const long grp = ((long)(blockIdx.x >> 3) / share) * 8 + (blockIdx.x & 7);
    const long base = (grp * 8192 * 64) % region_floats;  // distinct groups are 2 MiB apart
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const long off = (base + (long)it * 8192) % region_floats;   // 32 KB per iteration per WG (iters * 32 KB <= 2 MiB apart)
        const float* p = src + off + wave * 2048 + lane * 4;          // wave: 8 KB = 8 pieces of 1 KB
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) glds16(p + c * 256, lds + (it & 1) * 8192 + wave * 2048 + c * 256);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            float4 v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const float4*>(p + c * 256);
#pragma unroll
            for (int c = 0; c < 8; ++c) { acc.x += v[c].x; acc.y += v[c].y; acc.z += v[c].z; acc.w += v[c].w; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) { __syncthreads(); acc.x = lds[threadIdx.x]; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main() {
    const long total = 1L << 28;  // 1 GiB of floats = 4 GiB? no: 2^28 floats = 1 GiB
    float *src, *sink;
    hipMalloc(&src, total * sizeof(float));
    hipMalloc(&sink, 64);
    hipMemset(src, 0, total * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 60;
    const long regions[] = {1L << 24 /*64 MiB*/, 1L << 28 /*1 GiB: the whole buffer*/};
    for (int share : {1, 2, 3, 6})
        for (long reg : regions)
            for (int mode = 0; mode < 1; ++mode) {
                const int wgs_per_cu = 2;
                const int grid = 256 * wgs_per_cu;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 65536, 0, src, reg, iters, sink, share);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)grid * iters * 32768.0;
                printf("share %d region %6ld KiB lds-dma: %.1f us  %.2f TB/s into CUs  %.1f B/clk/CU @2.1GHz  (distinct bytes %.2f TB/s)\n", share, reg * 4 / 1024,
                       ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9, bytes / share / ms / 1e9);
            }
    return 0;
}
