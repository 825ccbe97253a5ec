// vmcnt_order_probe.hip — do vector-memory loads and stores of one wave retire IN ORDER with respect to each other on gfx950?
// (tools/: hardware facts the kernels rest on.)  gfx9-family chips count loads and stores in ONE counter (vmcnt).  The
// walk kernel wants to wait for the loads of micro-operation k with "s_waitcnt vmcnt(N)", N = everything issued after
// them: the STORES of micro-operation k-1 and the loads of k+1.  That is only correct if a younger store can never be
// acknowledged before an older load has returned its data.
// Test: every wave repeatedly issues a COLD load (random line of a 2 GiB array: an HBM miss) into a register preset to a
// sentinel, then four non-temporal 16-byte STORES to a line it owns (hot in L2), then waits vmcnt(4).  If stores could
// overtake the load, the wait would pass with the load still pending and the register would still hold the sentinel.
// The second experiment swaps the roles (cold store first, hot load second, vmcnt(1) must not pass on the load alone is
// not needed by the kernel and not tested).  Prints the number of violations and the average wait in clocks.
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmcnt_order_probe.hip -o /tmp/vp && /tmp/vp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* cold, size_t coldWords, u32x4* hot, unsigned long long* out, int iters, int mode) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long state = 0x9E3779B97F4A7C15ull * (gid + 1);
    unsigned long long violations = 0, sum = 0, clocks = 0;
    u32x4* mine = hot + (size_t)gid * 4;
    u32x4 data = {gid, 1u, 2u, 3u};
    for (int i = 0; i < iters; i++) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const size_t w = (size_t)((state >> 20) % coldWords);
        const unsigned* src = cold + w;
        unsigned got = 0xDEADBEEFu;
        const long long t0 = clock64();
        if (mode == 1)            // control: the four hot stores alone
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx4 %0, %1, off offset:16 nt\n\t"
                         "global_store_dwordx4 %0, %1, off offset:32 nt\n\tglobal_store_dwordx4 %0, %1, off offset:48 nt\n\t"
                         "s_waitcnt vmcnt(0)" : : "v"(mine), "v"(data) : "memory");
        else if (mode == 2)       // control: the cold load alone
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(got) : "v"(src) : "memory");
        else
        asm volatile(
            "global_load_dword %0, %1, off\n\t"
            "global_store_dwordx4 %2, %3, off nt\n\t"
            "global_store_dwordx4 %2, %3, off offset:16 nt\n\t"
            "global_store_dwordx4 %2, %3, off offset:32 nt\n\t"
            "global_store_dwordx4 %2, %3, off offset:48 nt\n\t"
            "s_waitcnt vmcnt(4)\n\t"
            "s_nop 0"
            : "+v"(got) : "v"(src), "v"(mine), "v"(data) : "memory");
        const unsigned seen = got;              // read BEFORE draining
        const long long t1 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (seen == 0xDEADBEEFu) violations++;
        sum += seen; clocks += (unsigned long long)(t1 - t0);
        data.y += seen;
    }
    atomicAdd(&out[0], violations); atomicAdd(&out[1], sum); atomicAdd(&out[2], clocks);
}
int main() {
    const size_t coldWords = (size_t)1 << 29;                 // 2 GiB
    unsigned* cold; u32x4* hot; unsigned long long* out;
    const int maxBlocks = 2048, threads = 256, iters = 2000;
    hipMalloc(&cold, coldWords * 4); hipMemset(cold, 0x11, coldWords * 4);       // never the sentinel
    hipMalloc(&hot, (size_t)maxBlocks * threads * 64); hipMemset(hot, 0, (size_t)maxBlocks * threads * 64);
    hipMalloc(&out, 24); hipMemset(out, 0, 24);
    unsigned long long h[3];
    unsigned long long total = 0;
    for (int blocks : {64, 512, 2048}) {                       // light load (stores much faster than the miss) to saturation
    const double n = (double)blocks * threads * iters;
    printf("-- %d workgroups of %d threads\n", blocks, threads);
    for (int mode = 1; mode <= 2; mode++) {
        hipMemset(out, 0, 24);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cold, coldWords, hot, out, iters, mode);
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        printf("control, %s: average %.0f clocks until vmcnt(0)\n", mode == 1 ? "four hot nt stores alone" : "cold load alone", (double)h[2] / n);
    }
    hipMemset(out, 0, 24);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cold, coldWords, hot, out, iters, 0);
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("vmcnt order probe: %llu violations in %.3g lane-trials (cold load, then 4 hot nt stores, vmcnt(4)); average wait %.0f clocks\n",
           h[0], n, (double)h[2] / n);
    total += h[0];
    }
    printf("%s\n", total ? "OUT OF ORDER: younger stores were acknowledged before an older load returned"
                        : "in order: a younger store is never counted out before an older load");
    return 0;
}
