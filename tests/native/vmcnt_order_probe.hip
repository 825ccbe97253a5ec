// vmcnt_order_probe.hip — do vector-memory loads and stores of one wave retire IN ORDER with respect to each other on gfx950?
// TEST INFRASTRUCTURE (tests/test_gpu_vmcnt_order.py builds and runs it on the GPU box).
//
// gfx9-family chips count loads and stores in ONE counter (vmcnt).  The 4-state walk kernels (beast-mcmc_amd/csrc/
// kernels_walk4.hip, tools/gen_walk4_fast.py) wait for the loads of micro-operation k with "s_waitcnt vmcnt(N)", N =
// everything issued after them: the STORES of micro-operation k-1 and the loads of k+1.  That is only correct if a younger
// store can never be acknowledged — counted out — before an older load has returned its data.  The ISA guides promise
// in-order return for loads among themselves; this probe looks for a counter-example to the stronger property, in the
// situations the kernels create:
//   A  cold load (random line of a 2 GiB array: an HBM miss), then four HOT non-temporal 16-byte stores (a line the lane
//      owns), s_waitcnt vmcnt(4): the register must hold the loaded value, not the sentinel it was preset to
//   B  the same with four COLD stores (random lines) — slow stores behind a slow load
//   C  four cold stores FIRST (older, slow, still in the queue), then the cold load, then four hot stores, vmcnt(4)
//   D  a cold LDS-DMA load (global_load_lds_dwordx4: the kernels' matrix-table fetch, the first load of every stage), then
//      four hot stores, vmcnt(4): the LDS word must hold the loaded value
// BEAGLE_MI355_STRICT_WAITS=1 makes the engine independent of the property (engine_walk.cpp runPlan).
// Output: one "PROBE <name> <violations> <trials>" line per experiment and load level.
// Build: hipcc --offload-arch=gfx950 -O3 tests/native/vmcnt_order_probe.hip -o /tmp/vp && /tmp/vp [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned SENTINEL = 0xDEADBEEFu;

__global__ void k(const unsigned* cold, size_t coldWords, u32x4* coldSt, size_t coldStLines, u32x4* hot, unsigned long long* out, int iters, int mode) {
    __shared__ u32x4 ldsBuf[256];
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long state = 0x9E3779B97F4A7C15ull * (gid + 1);
    unsigned long long violations = 0, sum = 0;
    u32x4* mine = hot + (size_t)gid * 4;
    u32x4 data = {gid, 1u, 2u, 3u};
    const unsigned ldsWave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ldsBuf + (threadIdx.x & ~63u)));
    for (int i = 0; i < iters; i++) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned* src = cold + (size_t)((state >> 20) % coldWords);
        u32x4* c0 = coldSt + (size_t)((state >> 13) % coldStLines) * 4;     // a random 64-byte line of the store arena
        u32x4* c1 = coldSt + (size_t)((state >> 29) % coldStLines) * 4;
        unsigned got = SENTINEL;
        if (mode == 0)
            asm volatile("global_load_dword %0, %1, off\n\t"
                         "global_store_dwordx4 %2, %3, off nt\n\tglobal_store_dwordx4 %2, %3, off offset:16 nt\n\t"
                         "global_store_dwordx4 %2, %3, off offset:32 nt\n\tglobal_store_dwordx4 %2, %3, off offset:48 nt\n\t"
                         "s_waitcnt vmcnt(4)\n\ts_nop 0"
                         : "+v"(got) : "v"(src), "v"(mine), "v"(data) : "memory");
        else if (mode == 1)
            asm volatile("global_load_dword %0, %1, off\n\t"
                         "global_store_dwordx4 %2, %3, off nt\n\tglobal_store_dwordx4 %2, %3, off offset:16 nt\n\t"
                         "global_store_dwordx4 %4, %3, off offset:32 nt\n\tglobal_store_dwordx4 %4, %3, off offset:48 nt\n\t"
                         "s_waitcnt vmcnt(4)\n\ts_nop 0"
                         : "+v"(got) : "v"(src), "v"(c0), "v"(data), "v"(c1) : "memory");
        else if (mode == 2)
            asm volatile("global_store_dwordx4 %4, %3, off nt\n\tglobal_store_dwordx4 %4, %3, off offset:16 nt\n\t"
                         "global_store_dwordx4 %5, %3, off offset:32 nt\n\tglobal_store_dwordx4 %5, %3, off offset:48 nt\n\t"
                         "global_load_dword %0, %1, off\n\t"
                         "global_store_dwordx4 %2, %3, off nt\n\tglobal_store_dwordx4 %2, %3, off offset:16 nt\n\t"
                         "global_store_dwordx4 %2, %3, off offset:32 nt\n\tglobal_store_dwordx4 %2, %3, off offset:48 nt\n\t"
                         "s_waitcnt vmcnt(4)\n\ts_nop 0"
                         : "+v"(got) : "v"(src), "v"(mine), "v"(data), "v"(c0), "v"(c1) : "memory");
        else {
            // the wave's 64 x 16 bytes of LDS preset to the sentinel, then the DMA of 16 bytes per lane from a cold line
            ldsBuf[threadIdx.x] = u32x4{SENTINEL, SENTINEL, SENTINEL, SENTINEL};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned* src16 = (const unsigned*)((size_t)src & ~(size_t)15);
            asm volatile("s_mov_b32 m0, %4\n\t"
                         "global_load_lds_dwordx4 %0, off\n\t"
                         "global_store_dwordx4 %1, %2, off nt\n\tglobal_store_dwordx4 %1, %2, off offset:16 nt\n\t"
                         "global_store_dwordx4 %1, %2, off offset:32 nt\n\tglobal_store_dwordx4 %1, %2, off offset:48 nt\n\t"
                         "s_waitcnt vmcnt(4)\n\ts_nop 0"
                         : : "v"(src16), "v"(mine), "v"(data), "v"(0), "s"(ldsWave) : "memory");
            got = ((volatile unsigned*)&ldsBuf[threadIdx.x])[0];
        }
        const unsigned seen = got;              // read BEFORE draining
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (seen == SENTINEL) violations++;
        sum += seen;
        data.y += seen;
    }
    atomicAdd(&out[0], violations); atomicAdd(&out[1], sum);
}

int main(int argc, char** argv) {
    const size_t coldWords = (size_t)1 << 29;                 // 2 GiB
    const size_t coldStLines = (size_t)1 << 24;               // 1 GiB of 64-byte lines
    unsigned* cold; u32x4 *hot, *coldSt; unsigned long long* out;
    const int maxBlocks = 2048, threads = 256, iters = argc > 1 ? atoi(argv[1]) : 600;
    if (hipMalloc(&cold, coldWords * 4) != hipSuccess || hipMalloc(&coldSt, coldStLines * 64) != hipSuccess) { printf("no memory\n"); return 2; }
    hipMemset(cold, 0x11, coldWords * 4);                      // never the sentinel
    hipMalloc(&hot, (size_t)maxBlocks * threads * 64); hipMemset(hot, 0, (size_t)maxBlocks * threads * 64);
    hipMalloc(&out, 16);
    unsigned long long h[2], total = 0;
    double trials = 0;
    const char* names[4] = {"A:cold-load,hot-stores", "B:cold-load,cold-stores", "C:older-cold-stores,cold-load,hot-stores", "D:cold-lds-dma,hot-stores"};
    for (int blocks : {64, 512, 2048})                         // light load (stores much faster than the miss) to saturation
        for (int mode = 0; mode < 4; mode++) {
            hipMemset(out, 0, 16);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cold, coldWords, coldSt, coldStLines, hot, out, iters, mode);
            if (hipMemcpy(h, out, 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
            const double n = (double)blocks * threads * iters;
            printf("PROBE %s blocks=%d %llu %.0f\n", names[mode], blocks, h[0], n);
            total += h[0]; trials += n;
        }
    printf("TOTAL %llu violations in %.4g lane-trials\n", total, trials);
    printf("%s\n", total ? "OUT OF ORDER: younger stores were acknowledged before an older load returned"
                        : "in order: a younger store is never counted out before an older load");
    return total ? 1 : 0;
}
