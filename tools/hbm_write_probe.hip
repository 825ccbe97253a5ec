// hbm_write_probe.hip — what the MI355X write path sustains for the store shapes the walk kernel can choose from
// (tools/: hardware facts the kernels rest on).  Every variant writes the same 4 GiB once with global_store_dwordx4:
//   contiguous : lane l writes the 16 bytes at 16 l of its wave's 1 KiB (every instruction = 8 full 128-byte lines)
//   halfline   : lane l writes 16 bytes at 32 l, then 16 bytes at 32 l + 16 (the [pattern][4 doubles] layout: every
//                instruction touches 16 lines and fills half of each; the other half comes with the next instruction)
// each with default and non-temporal cache policy; plus a read-only and a copy pass for reference.
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_write_probe.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ void k(u32x4* dst, const u32x4* src, size_t n16, unsigned* sink) {      // n16 = number of 16-byte items
    const size_t wavesTotal = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    u32x4 val = {lane, 1u, 2u, 3u};
    unsigned acc = 0;
    // a wave owns chunks of 2 KiB (128 items), grid-strided
    for (size_t chunk = wave; chunk < n16 / 128; chunk += wavesTotal) {
        u32x4* p = dst + chunk * 128;
        if (MODE == 0) {            // contiguous: two instructions, 1 KiB each
            if (NT) { __builtin_nontemporal_store(val, p + lane); __builtin_nontemporal_store(val, p + 64 + lane); }
            else { p[lane] = val; p[64 + lane] = val; }
        } else if (MODE == 1) {     // half lines
            if (NT) { __builtin_nontemporal_store(val, p + 2 * lane); __builtin_nontemporal_store(val, p + 2 * lane + 1); }
            else { p[2 * lane] = val; p[2 * lane + 1] = val; }
        } else if (MODE == 2) {     // read only
            const u32x4* q = src + chunk * 128;
            const u32x4 a = q[lane], b = q[64 + lane];
            acc += a.x + b.y;
        } else {                    // copy
            const u32x4* q = src + chunk * 128;
            const u32x4 a = q[lane], b = q[64 + lane];
            if (NT) { __builtin_nontemporal_store(a, p + lane); __builtin_nontemporal_store(b, p + 64 + lane); }
            else { p[lane] = a; p[64 + lane] = b; }
        }
    }
    if (MODE == 2 && acc == 0x12345678u) *sink = acc;
}
template <int MODE, bool NT>
double run(u32x4* d, const u32x4* s, size_t n16, unsigned* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, NT>), dim3(256 * 8), dim3(256), 0, 0, d, s, n16, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return (double)n16 * 16 / (best * 1e-3) / 1e9;
}
int main() {
    const size_t bytes = (size_t)4 << 30, n16 = bytes / 16;
    u32x4 *d, *s; unsigned* sink;
    hipMalloc(&d, bytes); hipMalloc(&s, bytes); hipMalloc(&sink, 4);
    hipMemset(d, 0, bytes); hipMemset(s, 1, bytes);
    printf("write, contiguous 1 KiB per instruction        : %7.0f GB/s\n", run<0, false>(d, s, n16, sink));
    printf("write, contiguous, non-temporal                : %7.0f GB/s\n", run<0, true>(d, s, n16, sink));
    printf("write, half lines (16 B at 32-B stride, twice) : %7.0f GB/s\n", run<1, false>(d, s, n16, sink));
    printf("write, half lines, non-temporal                : %7.0f GB/s\n", run<1, true>(d, s, n16, sink));
    printf("read only                                      : %7.0f GB/s\n", run<2, false>(d, s, n16, sink));
    printf("copy (GB/s of bytes written; x2 moved)         : %7.0f GB/s\n", run<3, false>(d, s, n16, sink));
    printf("copy, non-temporal stores                      : %7.0f GB/s\n", run<3, true>(d, s, n16, sink));
    return 0;
}
