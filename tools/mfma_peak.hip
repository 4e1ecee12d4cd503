// Register-only MFMA throughput probe (fp32 32x32x2 / 16x16x4) with shader-clock readout (s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, const float* in, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, const float* in, int iters, unsigned long long* clk) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// fp16 32x32x16 (the instruction of the f16x3 engines): NACC independent accumulators, operands from memory (random fp16)
template <int NACC>
__global__ __launch_bounds__(512) void kh(float* out, const float* in, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    half8 av, bv;
    for (int e = 0; e < 8; ++e) {
        av[e] = (_Float16)in[(threadIdx.x * 8 + e) & 511];
        bv[e] = (_Float16)in[(threadIdx.x * 8 + e + 77) & 511];
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <typename F>
void run(const char* name, F launch, double flops_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f TF", name, blocks, ms, flops_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float *out, *in;
    unsigned long long* clk;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&in, 512 * 4);
    hipMalloc(&clk, 8);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) for (auto& v : h) v = 0.f;
        hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
        printf("---- operands: %s\n", zero ? "zero" : "random");
        for (int blocks : {256, 512, 1024}) {
            const int iters = 20000;
            unsigned long long c = 0;
            run("mfma_f32_32x32x2 x4acc", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, in, it, clk); },
                4.0 * 4 * 2 * 32 * 32 * 2, blocks, iters);
            hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("  cycles/mfma(wave) %.1f\n", (double)c / (iters * 4.0));
            run("mfma_f32_32x32x16_f16 x4acc", [&](int it) { hipLaunchKernelGGL(kh<4>, dim3(blocks), dim3(256), 0, 0, out, in, it, clk); },
                4.0 * 4 * 2 * 32 * 32 * 16, blocks, iters);
            hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("  cycles/mfma(wave) %.1f\n", (double)c / (iters * 4.0));
            run("mfma_f32_16x16x4 x4acc", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, in, it, clk); },
                4.0 * 4 * 2 * 16 * 16 * 4, blocks, iters);
            hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("  cycles/mfma(wave) %.1f\n", (double)c / (iters * 4.0));
        }
    }
    // dependent-accumulator latency of v_mfma_f32_32x32x16_f16: NACC independent chains, one wave per SIMD (256 blocks of
    // 256 threads) and two waves per SIMD (256 blocks of 512 threads)
    for (auto& v : h) v = 0.f;
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    printf("---- dependent chains (fp16 32x32x16, random operands)\n");
    for (int threads : {256, 512}) {
        const int iters = 20000;
        unsigned long long c = 0;
        auto rep = [&](const char* nm, int nacc, auto kern) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, in, iters, clk);
            hipDeviceSynchronize();
            hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            printf("%s threads/block %d: cycles per MFMA (one wave) %.1f\n", nm, threads, (double)c / (iters * (double)nacc));
        };
        rep("1 accumulator ", 1, kh<1>);
        rep("2 accumulators", 2, kh<2>);
        rep("3 accumulators", 3, kh<3>);
        rep("4 accumulators", 4, kh<4>);
    }
    return 0;
}
