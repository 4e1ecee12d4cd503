// Does vector-ALU work overlap with matrix-core work on one gfx950 SIMD?  (Round 3: two structurally different fc1 kernels -
// conv_ws.hip with a barrier per 36 MFMAs, conv_wl.hip without any - both ran 134 us for 51 us of MFMA time and 43 us of
// epilogue VALU issue, which is what "the two do not overlap" would give.)
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD).  Per wave: NM v_mfma_f32_32x32x16_f16 on 4 independent
// accumulators, or NV independent v_fma_f32, or both interleaved.
//   M    every wave MFMAs                                  V    every wave VALU
//   MV   waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
//   I<k> every wave: k VALU after each MFMA, in its own instruction stream
// time(MV) ~ max(time of its MFMA half, time of its VALU half)  => the pipes overlap across waves;  ~ sum => they serialise.
// build: hipcc --offload-arch=gfx950 -O3 -o exp/mfma_valu_overlap tools/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define FMA8(x, a, b)                                                                                                       \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
                 "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9" \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])          \
                 : "v"(a), "v"(b))

// MODE 0: M, 1: V, 2: MV, 3: interleaved with KV VALU per MFMA
// BIGLDS: the workgroup also declares 147 KiB of LDS (as conv_wl.hip / conv_ws.hip do): does launching 256 such workgroups
// cost wall time that no wave sees?
template <int MODE, int KV, bool BIGLDS = false>
__global__ __launch_bounds__(512) void probe(float* out, const float* in, int nm, int nv, unsigned long long* ticks) {
    __shared__ float big[BIGLDS ? 147 * 256 : 1];
    if (BIGLDS) { big[threadIdx.x] = in[threadIdx.x]; __syncthreads(); }
    const int wave = threadIdx.x >> 6;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    half8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (_Float16)in[(threadIdx.x * 8 + e) & 511]; bv[e] = (_Float16)in[(threadIdx.x * 8 + e + 77) & 511]; }
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = in[(threadIdx.x + e) & 511];
    const float fa = in[3] * 1e-3f + 0.999f, fb = in[5] * 1e-6f;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 1 || (MODE == 2 && wave >= 4);
    if (MODE == 3) {
        for (int it = 0; it < nm / 4; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a], 0, 0, 0);
                if (KV >= 8) FMA8(x, fa, fb);
                else {
#pragma unroll
                    for (int k = 0; k < KV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(fa), "v"(fb));
                }
            }
        }
    } else {
        if (do_m)
            for (int it = 0; it < nm / 4; ++it) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a], 0, 0, 0);
            }
        if (do_v)
            for (int it = 0; it < nv / 8; ++it) FMA8(x, fa, fb);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int e = 0; e < 8; ++e) s += x[e];
    if (BIGLDS) s += big[(threadIdx.x * 7) & 511];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t_start;   // s_memtime ticks of wave 0
}

static unsigned long long* g_ticks = nullptr;
static unsigned long long last_ticks() { unsigned long long t = 0; (void)hipMemcpy(&t, g_ticks, 8, hipMemcpyDeviceToHost); return t; }
template <int MODE, int KV, bool BIGLDS = false>
static double run(float* out, const float* in, int nm, int nv) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, KV, BIGLDS>), dim3(256), dim3(512), 0, 0, out, in, nm, nv, g_ticks);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<MODE, KV, BIGLDS>), dim3(256), dim3(512), 0, 0, out, in, nm, nv, g_ticks);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / 5;   // us per launch
}

int main() {
    float *in, *out;
    (void)hipMalloc(reinterpret_cast<void**>(&in), 512 * 4);
    (void)hipMalloc(reinterpret_cast<void**>(&out), 256 * 512 * 4);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0.25f + 0.001f * (float)((i * 37) % 101);
    (void)hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(reinterpret_cast<void**>(&g_ticks), 8);
    const int nm = 40000;              // MFMAs per wave
    {   // what does one s_memtime tick measure?  8 waves x nm MFMAs = 2 nm x 32 matrix-pipe cycles per SIMD at least
        const double us = run<0, 0>(out, in, nm, 0);
        const unsigned long long tk = last_ticks();
        printf("clock check: M run %.1f us, wave 0 spans %llu s_memtime ticks = %.3f ticks/ns; the matrix pipe needs %d cycles "
               "=> >= %.2f GHz if a tick is a shader cycle\n", us, tk, (double)tk / (us * 1e3), 2 * nm * 32, 2.0 * nm * 32 / (us * 1e3));
    }
    for (int small : {2000, 4000}) {   // conv-sized launches (a 1-degree fc1 is ~3500 MFMAs per wave): wall time vs what wave 0 sees
        const double us = run<0, 0, false>(out, in, small, 0);
        const unsigned long long tk = last_ticks();
        const double usb = run<0, 0, true>(out, in, small, 0);
        const unsigned long long tkb = last_ticks();
        printf("launch of 256 workgroups x 8 waves x %d MFMAs: %.1f us wall, wave 0 lives %llu cycles;  with 147 KiB of LDS per workgroup: "
               "%.1f us wall, %llu cycles\n", small, us, tk, usb, tkb);
    }
    for (int ratio : {4, 7, 8}) {      // VALU per MFMA (fc1's epilogue: 6.7)
        const int nv = nm * ratio;
        const double tm = run<0, 0>(out, in, nm, nv), tv = run<1, 0>(out, in, nm, nv), tmv = run<2, 0>(out, in, nm, nv);
        printf("ratio %d VALU per MFMA:  M (8 waves MFMA) %8.1f us   V (8 waves VALU) %8.1f us   MV (4 + 4) %8.1f us   "
               "[halves alone: M/2 = %.1f, V/2 = %.1f; serialised %.1f, overlapped %.1f]\n",
               ratio, tm, tv, tmv, tm / 2, tv / 2, tm / 2 + tv / 2, tm / 2 > tv / 2 ? tm / 2 : tv / 2);
    }
    printf("interleaved in ONE stream (8 waves, each %d MFMAs with k VALU after every MFMA):\n", nm);
    printf("  k = 0: %8.1f us\n", run<3, 0>(out, in, nm, 0));
    printf("  k = 2: %8.1f us\n", run<3, 2>(out, in, nm, 0));
    printf("  k = 4: %8.1f us\n", run<3, 4>(out, in, nm, 0));
    printf("  k = 6: %8.1f us\n", run<3, 6>(out, in, nm, 0));
    printf("  k = 8: %8.1f us\n", run<3, 8>(out, in, nm, 0));
    return 0;
}
