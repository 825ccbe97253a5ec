// walk_probe.hip — hardware facts the pipelined walk kernel relies on, checked on the GPU box (gfx950):
//   1. v_fmac_f64_dpp ... row_newbcast:N multiplies by lane N of each 16-lane row (DPP64), and its rate vs plain v_fmac_f64
//   2. a VMEM load issued with EXEC = 0 still counts in vmcnt (so "s_waitcnt vmcnt(N)" with a fixed N stays exact when
//      some of the loads of a fixed-length sequence are masked off)
// build: hipcc --offload-arch=gfx950 -O3 tools/walk_probe.hip -o tools/walk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__global__ void k_dpp(double* out, const double* m, const double* x) {
    const int lane = threadIdx.x & 63;
    double sp = m[lane & 15];
    double x0 = x[lane], y[16];
#define ONE(N) { double a = 0.0; asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(sp), "v"(x0)); y[N] = a; }
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15)
    for (int n = 0; n < 16; n++) out[n * 64 + lane] = y[n];
}

template <bool DPP>
__global__ void k_rate(double* out, const double* m, int iters) {
    const int lane = threadIdx.x & 63;
    double sp = m[lane & 15];
    double a0 = 0.1 * lane, a1 = 0.2, a2 = 0.3, a3 = 0.4, x = 1.0000001;
    for (int i = 0; i < iters; i++) {
        if (DPP) asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %1, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %2, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sp), "v"(x));
        else asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sp), "v"(x));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

// every wave: one real load from cold memory, then 9 loads with EXEC = 0, then s_waitcnt vmcnt(9), then use the value
__global__ void k_exec0(const double* __restrict__ src, double* __restrict__ out, int* __restrict__ bad, size_t stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double* p = src + i * stride;
    double v = -1.0, d0 = 0, d1 = 0, d2 = 0;
    asm volatile(
        "global_load_dwordx2 %0, %4, off\n\t"
        "s_mov_b64 s[20:21], exec\n\t"
        "s_mov_b64 exec, 0\n\t"
        "global_load_dwordx2 %1, %4, off\n\t"
        "global_load_dwordx2 %2, %4, off\n\t"
        "global_load_dwordx2 %3, %4, off\n\t"
        "global_load_dwordx2 %1, %4, off\n\t"
        "global_load_dwordx2 %2, %4, off\n\t"
        "global_load_dwordx2 %3, %4, off\n\t"
        "global_load_dwordx2 %1, %4, off\n\t"
        "global_load_dwordx2 %2, %4, off\n\t"
        "global_load_dwordx2 %3, %4, off\n\t"
        "s_mov_b64 exec, s[20:21]\n\t"
        "s_waitcnt vmcnt(9)"
        : "+v"(v), "+v"(d0), "+v"(d1), "+v"(d2) : "v"(p) : "s20", "s21", "memory");
    const double got = v;            // must be the loaded value if masked loads count in vmcnt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[i] = got;
    if (got != (double)(i % 1000003)) atomicAdd(bad, 1);
}

int main() {
    double hm[16], hx[64], *dm, *dx, *dout;
    for (int i = 0; i < 16; i++) hm[i] = 1.0 + i;
    for (int i = 0; i < 64; i++) hx[i] = 0.5 + 0.01 * i;
    hipMalloc(&dm, sizeof hm); hipMalloc(&dx, sizeof hx); hipMalloc(&dout, 16 * 64 * 8);
    hipMemcpy(dm, hm, sizeof hm, hipMemcpyHostToDevice); hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, dout, dm, dx);
    std::vector<double> ho(16 * 64);
    hipMemcpy(ho.data(), dout, ho.size() * 8, hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int n = 0; n < 16; n++) for (int l = 0; l < 64; l++) if (ho[n * 64 + l] != hm[n] * hx[l]) wrong++;
    printf("dpp row_newbcast semantics: %s (%d mismatches of 1024)\n", wrong ? "WRONG" : "as assumed: lane N of each 16-lane row", wrong);
    if (wrong) for (int l = 0; l < 20; l++) printf("  n=3 lane %d got %.6f expect %.6f\n", l, ho[3 * 64 + l], hm[3] * hx[l]);

    const int blocks = 256 * 8, iters = 20000;
    double* dbig; hipMalloc(&dbig, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dpp = 0; dpp < 2; dpp++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (dpp) hipLaunchKernelGGL(k_rate<true>, dim3(blocks), dim3(256), 0, 0, dbig, dm, iters);
            else hipLaunchKernelGGL(k_rate<false>, dim3(blocks), dim3(256), 0, 0, dbig, dm, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 4 * iters * (double)blocks * 256;
        printf("%s: %.3f ms, %.1f TFLOP/s fp64\n", dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64 (plain)     ", ms, flops / ms / 1e9);
    }

    // EXEC = 0 loads and vmcnt: 1M threads, each reads its own cold cache line (stride 16 doubles = 128 B)
    const size_t n = 1 << 20, stride = 16;
    std::vector<double> hs(n * stride, -7.0);
    for (size_t i = 0; i < n; i++) hs[i * stride] = (double)(i % 1000003);
    double *dsrc, *dres; int* dbad;
    hipMalloc(&dsrc, hs.size() * 8); hipMalloc(&dres, n * 8); hipMalloc(&dbad, 4);
    hipMemcpy(dsrc, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
    int totalBad = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(dbad, 0, 4);
        hipLaunchKernelGGL(k_exec0, dim3(n / 256), dim3(256), 0, 0, dsrc, dres, dbad, stride);
        int hb = 0; hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost);
        totalBad += hb;
    }
    printf("EXEC=0 loads count in vmcnt: %s (%d stale values over 5M loads)\n", totalBad ? "NO - masked loads are NOT counted" : "yes", totalBad);
    return (wrong || totalBad) ? 1 : 0;
}
