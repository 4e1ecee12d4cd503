// Minimal reproducer of the gfx950 "store data must stay put" rule (profiles/r02_store_data_hazard.txt).
// Question it answers: can a 16-byte global store reach memory with the contents that an ASYNCHRONOUSLY returning LDS read
// wrote into the store's data VGPRs AFTER the store was issued?  (Round 2 saw this in three kernels, 3 % of entries down to
// 7 forwards in 1000; the kernels now hold store data until a vmcnt wait.)  The store and the read are ONE inline-asm block
// with the same register operand, so the register re-use is by construction and hipcc's bookkeeping is not involved.
//   variant 0: store, then ds_read_b128 into the same registers straight away
//   variant 1: store, NOPS wait states, then the read      (distance, what strip.hip relied on in round 2)
//   variant 2: store, s_waitcnt vmcnt(0), then the read    (the "held" form every kernel uses now)
//   variant 3: the SOFTWARE mechanism, by construction: an LDS read is in flight into registers that the store data is then
//              written to (what the register allocator may do around an inline-asm load it believes has completed: a dead
//              destination, a destination not named by the wait that retires it); the LDS return lands before the store is
//              issued and the store carries the read's contents.  Expected: every entry wrong
// Every workgroup keeps the vector-memory pipeline busy (all waves store 16 B per lane per iteration, whole chip).
// build: hipcc --offload-arch=gfx950 -O3 -o store_hazard tools/store_hazard.hip      run: ./store_hazard [iters] [nops]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef NOPS
#define NOPS 8
#endif
#define STR2(x) #x
#define STR(x) STR2(x)

template <int VARIANT>
__global__ __launch_bounds__(256) void hazard_kernel(u32x4* out, int iters) {
    __shared__ u32x4 lds[256];
    const int tid = threadIdx.x;
    lds[tid] = u32x4{0xB0000000u + tid, 0xB1111111u, 0xB2222222u, 0xB3333333u};   // what the LDS read returns ("new")
    __syncthreads();
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[tid];
    u32x4* dst = out + ((size_t)blockIdx.x * iters) * 256 + tid;
    for (int it = 0; it < iters; ++it) {
        u32x4 r = {0xA0000000u + (unsigned)it, 0xA1000000u + (unsigned)tid, 0xA2000000u + blockIdx.x, 0xA3333333u};   // "old"
        u32x4* p = dst + (size_t)it * 256;
        if (VARIANT == 0)
            asm volatile("global_store_dwordx4 %1, %0, off\n\tds_read_b128 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(p), "v"(la) : "memory");
        if (VARIANT == 1)
            asm volatile("global_store_dwordx4 %1, %0, off\n\ts_nop " STR(NOPS) "\n\tds_read_b128 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(p), "v"(la) : "memory");
        if (VARIANT == 2)
            asm volatile("global_store_dwordx4 %1, %0, off\n\ts_waitcnt vmcnt(0)\n\tds_read_b128 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(p), "v"(la) : "memory");
        if (VARIANT == 3) {
            // what the register allocator may legally produce around an inline-asm load whose result it believes is already
            // there (dead, or copied early): the read is issued into v, the store data is then built IN v, the LDS return lands,
            // the store goes out.  One asm block so that the interleaving is by construction.
            *p = r;                                       // dwords 1 - 3 (and 0, re-written below)
            unsigned q = 0;
            asm volatile("ds_read_b32 %0, %2\n\t"
                         "v_mov_b32 %0, %3\n\t"          // "new value" written over the in-flight destination
                         "s_sleep 4\n\t"                  // ... some hundred cycles of other work
                         "global_store_dword %1, %0, off\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(q) : "v"(p), "v"(la), "v"(0xA0000000u + (unsigned)it) : "memory");
            continue;
        }
        if (r[1] != 0xB1111111u) asm volatile("s_trap 2");   // the read itself must have landed
    }
}

template <int VARIANT>
static long run(int iters, int blocks, u32x4* dev, std::vector<u32x4>& host) {
    const size_t n = (size_t)blocks * iters * 256;
    (void)hipMemset(dev, 0, n * sizeof(u32x4));
    hipLaunchKernelGGL(hazard_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, dev, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("variant %d: launch failed\n", VARIANT); return -1; }
    (void)hipMemcpy(host.data(), dev, n * sizeof(u32x4), hipMemcpyDeviceToHost);
    long bad = 0, newer = 0;
    int shown = 0;
    for (size_t k = 0; k < n; ++k) {
        const int tid = (int)(k % 256), it = (int)((k / 256) % iters), blk = (int)(k / 256 / iters);
        const u32x4 e = {0xA0000000u + (unsigned)it, 0xA1000000u + (unsigned)tid, 0xA2000000u + (unsigned)blk, 0xA3333333u};
        const u32x4 g = host[k];
        bool ok = true, nw = false;
        for (int c = 0; c < 4; ++c) { ok &= g[c] == e[c]; nw |= (g[c] >> 28) == 0xBu; }
        bad += !ok; newer += nw;
        if (!ok && shown++ < 4) printf("  block %d iter %d lane %d: got %08x %08x %08x %08x\n", blk, it, tid, g[0], g[1], g[2], g[3]);
    }
    printf("variant %d: %ld of %zu stored entries wrong, %ld of them carry the LDS read's (later) contents\n", VARIANT, bad, n, newer);
    return bad;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 128, blocks = 256 * 8;
    const size_t n = (size_t)blocks * iters * 256;
    u32x4* dev = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dev), n * sizeof(u32x4)) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
    std::vector<u32x4> host(n);
    printf("store-data hazard probe: %d workgroups x 256 lanes x %d stores of 16 B, NOPS=%d\n", blocks, iters, NOPS);
    long tot[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        tot[0] += run<0>(iters, blocks, dev, host); tot[1] += run<1>(iters, blocks, dev, host);
        tot[2] += run<2>(iters, blocks, dev, host); tot[3] += run<3>(iters, blocks, dev, host);
    }
    printf("SUMMARY immediate=%ld distance=%ld held=%ld dead-asm-destination=%ld (wrong entries over 3 runs)\n", tot[0], tot[1], tot[2], tot[3]);
    (void)hipFree(dev);
    return 0;
}
