// mfma_f64_probe.hip — measures what the 20-/61-state kernels are designed around, on the GPU box:
//   1. the lane <-> matrix-element maps of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 (derived empirically
//      from one-hot inputs, printed as formulas checked over all lanes);
//   2. their sustained issue rate (FLOP/clk/CU) next to a plain v_fma_f64 loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

// out[la][lane][r] for A = one-hot at lane la, B lane value = lane+1
__global__ void k_map16(double* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++) {
        v4d c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(lane == la ? 1.0 : 0.0, (double)(lane + 1), c, 0, 0, 0);
        for (int r = 0; r < 4; r++) out[(la * 64 + lane) * 4 + r] = c[r];
    }
}
__global__ void k_map4(double* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++) {
        double c = 0.0;
        c = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, (double)(lane + 1), c, 0, 0, 0);
        out[la * 64 + lane] = c;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
    if (MODE == 0) {
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (MODE == 1) {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
        }
        out[blockIdx.x * 256 + lane] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_fma(a, b, c0); c1 = __builtin_fma(a, b, c1); c2 = __builtin_fma(a, b, c2); c3 = __builtin_fma(a, b, c3);
            c4 = __builtin_fma(a, b, c4); c5 = __builtin_fma(a, b, c5); c6 = __builtin_fma(a, b, c6); c7 = __builtin_fma(a, b, c7);
        }
        out[blockIdx.x * 256 + lane] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    }
}

template <int MODE>
double rate(double* d, double flopPerIterPerWave) {
    const int blocks = 256 * 4, iters = 20000;
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0;
    return flopPerIterPerWave * iters * waves / (ms * 1e-3) / 1e12;
}

int main() {
    double* d; hipMalloc(&d, 64 * 64 * 4 * sizeof(double) + 256 * 1024 * sizeof(double));
    std::vector<double> h(64 * 64 * 4);
    // ---- 16x16x4: A[i][k] at lane la, B[k][j] at lane lb, D[row][col] at (lane, reg)
    hipLaunchKernelGGL(k_map16, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, 64 * 64 * 4 * sizeof(double), hipMemcpyDeviceToHost);
    {
        // hypothesis: A lane la = (i = la & 15, k = la >> 4); B lane lb = (k = lb >> 4, j = lb & 15);
        //             D(lane, r) = (row = (lane >> 4) + 4 r, col = lane & 15)
        int bad = 0;
        for (int la = 0; la < 64; la++) for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 4; r++) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            const int i = la & 15, k = la >> 4;
            const double expect = (row == i) ? (double)((k << 4 | col) + 1) : 0.0;   // B lane with (k, j = col)
            if (h[(la * 64 + lane) * 4 + r] != expect) bad++;
        }
        printf("mfma_f64_16x16x4: A(i=l&15,k=l>>4) B(k=l>>4,j=l&15) D(row=(l>>4)+4r,col=l&15): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
        if (bad) for (int la = 0; la < 4; la++) { printf(" la=%d:", la); for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 4; r++) { double v = h[(la * 64 + lane) * 4 + r]; if (v != 0) printf(" (l%d r%d)=%g", lane, r, v); } printf("\n"); }
    }
    // ---- 4x4x4 x 4 blocks
    hipLaunchKernelGGL(k_map4, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, 64 * 64 * sizeof(double), hipMemcpyDeviceToHost);
    {
        // map (first derived from this probe's one-hot dump on an MI355X, now asserted):
        //   lane l: block = (l >> 2) & 3;  A(i = l & 3, k = l >> 4);  B(k = l >> 4, j = l & 3);  D(row = l >> 4, col = l & 3)
        int bad = 0;
        for (int la = 0; la < 64; la++) for (int lane = 0; lane < 64; lane++) {
            const int blkA = (la >> 2) & 3, i = la & 3, k = la >> 4;
            const int blkD = (lane >> 2) & 3, row = lane >> 4, col = lane & 3;
            const double expect = (blkA == blkD && row == i) ? (double)((k << 4 | blkD << 2 | col) + 1) : 0.0;
            if (h[la * 64 + lane] != expect) bad++;
        }
        printf("mfma_f64_4x4x4_4b: blk=(l>>2)&3 A(i=l&3,k=l>>4) B(k=l>>4,j=l&3) D(row=l>>4,col=l&3): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
        if (bad) for (int la = 0; la < 20; la++) { printf(" la=%d:", la); for (int lane = 0; lane < 64; lane++) { double v = h[la * 64 + lane]; if (v != 0) printf(" l%d=%g", lane, v); } printf("\n"); }
    }
    double* r = d + 64 * 64 * 4;
    printf("rate mfma_f64_16x16x4 : %.1f TFLOP/s (4 independent accumulators per wave, 4 waves/SIMD)\n", rate<0>(r, 4 * 2.0 * 16 * 16 * 4));
    printf("rate mfma_f64_4x4x4_4b: %.1f TFLOP/s (8 independent accumulators)\n", rate<1>(r, 8 * 2.0 * 4 * 4 * 4 * 4));
    printf("rate v_fma_f64        : %.1f TFLOP/s (8 independent accumulators)\n", rate<2>(r, 8 * 2.0 * 64));
    hipFree(d);
    return 0;
}
