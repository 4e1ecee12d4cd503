// What clock does a gfx950 kernel actually run at, and which counter tells?  (Round 5: VERDICT r04 reads GRBM_GUI_ACTIVE / wall
// = 2.1 GHz for the 1x1-convolution kernels and concludes "not power-limited"; the in-kernel s_memtime spans of round 3 give
// 1.1 GHz for the same kernels.  One of the two is not the shader clock under load.)
// Every workgroup stamps s_memtime (ISA: "free-running counter based on the shader core clock") AND s_memrealtime (fixed
// 100 MHz reference) at its start and end; the host brackets the launch with events.  Bodies:
//   0 spin    one wave per CU spinning on s_memtime (no pipes busy)
//   1 mfma    8 waves per CU, v_mfma_f32_32x32x16_f16 back to back on 4 accumulators (random operands)
//   2 mixed   as 1 with 6 v_fma_f32 after every MFMA and one ds_read_b128 per 3 MFMAs (the convolutions' diet)
//   3 mfma0   as 1 with all-zero operands (the guide's DVFS give-back check)
// Output: per body, wall us, median ticks of both counters, ticks / wall, and MFMA issue cycles needed / s_memtime ticks.
// Run it a second time under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES` to see what those say
// about the SAME launches.
// build: hipcc --offload-arch=gfx950 -O3 -o exp/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned long long realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int BODY>
__global__ __launch_bounds__(512) void probe(float* out, const float* in, int nm, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = in[k & 511];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = realtime();
    float s = 0.f;
    if (BODY == 0) {
        const unsigned long long want = (unsigned long long)nm;
        while (__builtin_amdgcn_s_memtime() - t0 < want) { asm volatile("s_sleep 1"); }
    } else {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        half8 av, bv;
        for (int e = 0; e < 8; ++e) {
            av[e] = BODY == 3 ? (_Float16)0.f : (_Float16)in[(threadIdx.x * 8 + e) & 511];
            bv[e] = BODY == 3 ? (_Float16)0.f : (_Float16)in[(threadIdx.x * 8 + e + 77) & 511];
        }
        float x[6];
        for (int e = 0; e < 6; ++e) x[e] = in[(threadIdx.x + e) & 511];
        const float fa = in[3] * 1e-3f + 0.999f, fb = in[5] * 1e-6f;
        f32x4 ld = {0.f, 0.f, 0.f, 0.f};
        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
        for (int it = 0; it < nm / 12; ++it) {
#pragma unroll
            for (int a = 0; a < 12; ++a) {
                acc[a & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a & 3], 0, 0, 0);
                if (BODY == 2) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(fa), "v"(fb));
                    if (a % 3 == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld) : "v"(la), "n"(1024 * (a / 3)));
                }
            }
            if (BODY == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
        }
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        for (int e = 0; e < 6; ++e) s += x[e];
        s += ld[0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = realtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int BODY>
static void run(const char* name, float* out, const float* in, int nm, unsigned long long* dstamps, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL((probe<BODY>), dim3(grid), dim3(threads), 0, 0, out, in, nm, dstamps);   // warm
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<BODY>), dim3(grid), dim3(threads), 0, 0, out, in, nm, dstamps);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    std::vector<unsigned long long> st(2 * grid);
    (void)hipMemcpy(st.data(), dstamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b;
    for (int g = 0; g < grid; ++g) { a.push_back(st[2 * g]); b.push_back(st[2 * g + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    const double mt = (double)a[grid / 2], rt = (double)b[grid / 2];
    // waves per SIMD = threads / 256; MFMA issue cycles one SIMD needs = waves per SIMD * nm * 32
    const double need = BODY == 0 ? 0.0 : (threads / 256.0) * nm * 32.0;
    printf("%-6s wall %9.1f us | s_memtime span %10.0f ticks = %6.3f ticks/ns of wall | s_memrealtime span %8.0f ticks = %6.1f ticks/us "
           "(=> kernel body %8.1f us at 100 MHz) | shader clock by the two counters %6.3f GHz",
           name, us, mt, mt / (us * 1e3), rt, rt / us, rt / 100.0, mt / (rt / 100.0) * 1e-3);
    if (need > 0) printf(" | MFMA issue cycles needed / s_memtime ticks = %.3f", need / mt);
    printf("\n");
}

int main() {
    float *out, *in;
    unsigned long long* st;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&in, 512 * 4);
    (void)hipMalloc(&st, 2 * 256 * 8);
    std::vector<float> h(512);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    (void)hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("spin", out, in, 2000000, st, 64);
        run<1>("mfma", out, in, 24000, st, 512);
        run<2>("mixed", out, in, 24000, st, 512);
        run<3>("mfma0", out, in, 24000, st, 512);
    }
    return 0;
}
