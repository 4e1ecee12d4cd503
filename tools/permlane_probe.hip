// Prints what v_permlane32_swap_b32 does on this part (which half of which operand moves where): the register-only
// P-format epilogue planned for the v4 engine relies on swap(vdst[32..63], src0[0..31]).
// Build: hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o exp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d;
    unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("vdst  lanes 0,31,32,63: %u %u %u %u   (in: 1000+lane)\n", h[0], h[31], h[32], h[63]);
    printf("src0  lanes 0,31,32,63: %u %u %u %u   (in: 2000+lane)\n", h[64], h[95], h[96], h[127]);
    return 0;
}
