// vmem_rate_probe.hip — how many cycles one CU needs per vector-memory INSTRUCTION (not per byte), by access shape, and
// whether an EXEC = 0 instruction is free.  Decides how the walk kernel's fetch stage is laid out.
// build: hipcc --offload-arch=gfx950 -O3 tools/vmem_rate_probe.hip -o tools/vmem_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { X4_COAL = 0, X2_COAL, X2_SPREAD, U8_COAL, X4_EXEC0, U8_EXEC0, ST4_EXEC0, X4_SAMELINE, NKIND };
static const char* kindName[NKIND] = {"dwordx4 coalesced (1 KiB/wave)", "dwordx2 coalesced (512 B/wave)", "dwordx2 16 addresses in one 128-B line",
                                      "ubyte coalesced (64 B/wave)", "dwordx4 with EXEC=0", "ubyte with EXEC=0", "store dwordx4 with EXEC=0",
                                      "dwordx4 all lanes in one 256-B window"};

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(const char* __restrict__ base, double* out, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned off;
    if (KIND == X4_COAL || KIND == X4_EXEC0 || KIND == ST4_EXEC0) off = (unsigned)((wave & 63) * 1024 + lane * 16);
    else if (KIND == X2_COAL) off = (unsigned)((wave & 63) * 512 + lane * 8);
    else if (KIND == X2_SPREAD) off = (unsigned)((wave & 63) * 128 + (lane & 15) * 8);
    else if (KIND == X4_SAMELINE) off = (unsigned)((wave & 63) * 256 + (lane & 15) * 16);
    else off = (unsigned)((wave & 63) * 64 + lane);
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d a0 = {0, 0}, a1 = a0, a2 = a0, a3 = a0;
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (int i = 0; i < iters; i++) {
        if (KIND == X4_COAL || KIND == X4_SAMELINE)
            asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5\n\tglobal_load_dwordx4 %2, %4, %5\n\tglobal_load_dwordx4 %3, %4, %5\n\ts_waitcnt vmcnt(2)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(off), "s"(base) : "memory");
        else if (KIND == X2_COAL || KIND == X2_SPREAD)
            asm volatile("global_load_dwordx2 %0, %4, %5\n\tglobal_load_dwordx2 %1, %4, %5\n\tglobal_load_dwordx2 %2, %4, %5\n\tglobal_load_dwordx2 %3, %4, %5\n\ts_waitcnt vmcnt(2)"
                         : "=&v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "=&v"(a3.x) : "v"(off), "s"(base) : "memory");
        else if (KIND == U8_COAL)
            asm volatile("global_load_ubyte %0, %4, %5\n\tglobal_load_ubyte %1, %4, %5\n\tglobal_load_ubyte %2, %4, %5\n\tglobal_load_ubyte %3, %4, %5\n\ts_waitcnt vmcnt(2)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(off), "s"(base) : "memory");
        else if (KIND == X4_EXEC0)
            asm volatile("s_mov_b64 exec, 0\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5\n\tglobal_load_dwordx4 %2, %4, %5\n\tglobal_load_dwordx4 %3, %4, %5\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(2)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(off), "s"(base) : "memory");
        else if (KIND == U8_EXEC0)
            asm volatile("s_mov_b64 exec, 0\n\tglobal_load_ubyte %0, %4, %5\n\tglobal_load_ubyte %1, %4, %5\n\tglobal_load_ubyte %2, %4, %5\n\tglobal_load_ubyte %3, %4, %5\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(2)"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(off), "s"(base) : "memory");
        else
            asm volatile("s_mov_b64 exec, 0\n\tglobal_store_dwordx4 %4, %0, %5\n\tglobal_store_dwordx4 %4, %1, %5\n\tglobal_store_dwordx4 %4, %2, %5\n\tglobal_store_dwordx4 %4, %3, %5\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(2)"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(off), "s"(base) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + b0 + b1 + b2 + b3;
}

template <int KIND> static void run(const char* base, double* out, int blocks, int iters, int cus, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, base, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = 4.0 * iters * blocks * 4;                    // wave-instructions
    printf("%-42s %8.3f ms  %6.2f cycles per wave-instruction per CU (at %.1f GHz, %d CUs)\n", kindName[KIND], ms,
           ms * 1e-3 * ghz * 1e9 * cus / instr, ghz, cus);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate / 1e6;
    char* base; double* out;
    hipMalloc(&base, 1 << 20); hipMemset(base, 1, 1 << 20);
    const int blocks = cus * 8, iters = 2000;                           // 8 workgroups x 4 waves = 8 waves per SIMD
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    run<X4_COAL>(base, out, blocks, iters, cus, ghz);
    run<X2_COAL>(base, out, blocks, iters, cus, ghz);
    run<X2_SPREAD>(base, out, blocks, iters, cus, ghz);
    run<X4_SAMELINE>(base, out, blocks, iters, cus, ghz);
    run<U8_COAL>(base, out, blocks, iters, cus, ghz);
    run<X4_EXEC0>(base, out, blocks, iters, cus, ghz);
    run<U8_EXEC0>(base, out, blocks, iters, cus, ghz);
    run<ST4_EXEC0>(base, out, blocks, iters, cus, ghz);
    return 0;
}
