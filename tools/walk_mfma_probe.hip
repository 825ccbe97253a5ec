// walk_mfma_probe.hip — what would the 4-state walk's inner arithmetic cost with the 4x4 mat-vec on the matrix pipe?
// (tools/: timing experiment, not product.)  Two emulations of the same micro-operation "node = (M1 . ACC) * column(M2, tip)
// [* 1/scale]" for 128 patterns x 1 category per wave, at the walk kernel's occupancy (4 waves per SIMD, set by dynamic LDS):
//   V  today's layout: a lane owns 2 patterns x 4 states; mat-vec = 8 zeroing moves + 32 v_fmac_f64_dpp row_newbcast;
//      tip column = 4 VALU + 4 ds_read_b128; 8 + 8 v_mul_f64
//   M  one STATE per lane (lane l = state l >> 4 of pattern slot l & 15, 8 registers = 8 x 16 patterns): mat-vec = 8
//      v_mfma_f64_4x4x4_4b_f64 (D's lane map equals B's, so results chain without any shuffle; C = 0: no zeroing); tip
//      column entry = 1 add (byte of the pre-multiplied state) + 1 ds_read_b64 per register; 8 (+ 8) v_mul_f64
// Prints ns per micro-operation and SIMD for both.  Build: hipcc --offload-arch=gfx950 -O3 tools/walk_mfma_probe.hip -o /tmp/wmp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void matvecDpp2(const double sp, const v4d xa, const v4d xb, v4d& ya, v4d& yb) {
    double a0, a1, a2, a3, b0, b1, b2, b3;
    const double p0 = xa.x, p1 = xa.y, p2 = xa.z, p3 = xa.w, q0 = xb.x, q1 = xb.y, q2 = xb.z, q3 = xb.w;
#define FM(Y, N, X) "v_fmac_f64_dpp %[" #Y "], %[sp], %[" #X "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile(
        "v_mov_b64 %[a0], 0\n\tv_mov_b64 %[a1], 0\n\tv_mov_b64 %[a2], 0\n\tv_mov_b64 %[a3], 0\n\t"
        "v_mov_b64 %[b0], 0\n\tv_mov_b64 %[b1], 0\n\tv_mov_b64 %[b2], 0\n\tv_mov_b64 %[b3], 0\n\t"
        FM(a0, 0, p0) FM(a1, 4, p0) FM(a2, 8, p0) FM(a3, 12, p0) FM(b0, 0, q0) FM(b1, 4, q0) FM(b2, 8, q0) FM(b3, 12, q0)
        FM(a0, 1, p1) FM(a1, 5, p1) FM(a2, 9, p1) FM(a3, 13, p1) FM(b0, 1, q1) FM(b1, 5, q1) FM(b2, 9, q1) FM(b3, 13, q1)
        FM(a0, 2, p2) FM(a1, 6, p2) FM(a2, 10, p2) FM(a3, 14, p2) FM(b0, 2, q2) FM(b1, 6, q2) FM(b2, 10, q2) FM(b3, 14, q2)
        FM(a0, 3, p3) FM(a1, 7, p3) FM(a2, 11, p3) FM(a3, 15, p3) FM(b0, 3, q3) FM(b1, 7, q3) FM(b2, 11, q3) FM(b3, 15, q3)
        "s_nop 0"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
        : [sp] "v"(sp), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3));
#undef FM
    ya = v4d{a0, a1, a2, a3}; yb = v4d{b0, b1, b2, b3};
}
__device__ __forceinline__ v4d tipColumn(const char* tbl, unsigned s) {
    const v2d* p = reinterpret_cast<const v2d*>(tbl + (s << 5));
    const v2d lo = p[0], hi = p[1];
    return v4d{lo.x, lo.y, hi.x, hi.y};
}

// MODE 0: V;  MODE 1: M.  SCALE: also multiply by a reciprocal.  TIPS: second child is a tip (else a second mat-vec on a held value)
template <int MODE, int SCALE, int TIPS>
__global__ __launch_bounds__(256, 4) void k(double* out, const double* tbl, int iters) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 40; i += 256) lds[i] = tbl[i];
    __syncthreads();
    const char* tb = reinterpret_cast<const char*>(lds);
    const double inv = 2.5;
    if (MODE == 0) {
        const int spOff = (((lane & 3) * 4) + ((lane & 15) >> 2)) * 8;
        v4d accA = v4d{1.0, 1.1, 1.2, 1.3} + 0.001 * lane, accB = accA * 1.01, hA = accA, hB = accB;
        unsigned t = (lane & 3) | (((lane >> 2) & 3) << 8);
        for (int it = 0; it < iters; it++) {
            v4d fa, fb, ga, gb;
            matvecDpp2(*reinterpret_cast<const double*>(tb + spOff), accA, accB, fa, fb);
            if (TIPS) { ga = tipColumn(tb + 160, t & 0xff); gb = tipColumn(tb + 160, t >> 8); }
            else matvecDpp2(*reinterpret_cast<const double*>(tb + 160 + spOff), hA, hB, ga, gb);
            accA = fa * ga; accB = fb * gb;
            if (SCALE) { accA = accA * inv; accB = accB * inv; }
            asm volatile("" : "+v"(accA), "+v"(accB), "+v"(t));
            t = ((t + 1) & 0x303);
        }
        out[blockIdx.x * 256 + threadIdx.x] = accA.x + accA.y + accA.z + accA.w + accB.x + accB.y + accB.z + accB.w;
    } else {
        // A operand: lane l holds M[i = l & 3][k = l >> 4] (tools/mfma_f64_probe.hip); table column k = 4 doubles T[k][i] = M[i][k]
        const double* T = reinterpret_cast<const double*>(tb);
        const double a1 = T[(lane >> 4) * 4 + (lane & 3)], a2 = T[20 + (lane >> 4) * 4 + (lane & 3)];
        double acc[8], h[8];
        for (int r = 0; r < 8; r++) { acc[r] = 1.0 + 0.1 * r + 0.001 * lane; h[r] = acc[r] * 1.01; }
        const unsigned rowOff = (unsigned)(lane >> 4) * 8 + 160;     // this lane's state row inside a column
        unsigned long long t = 0x0020406000204060ull ^ ((unsigned long long)(lane & 3) << 5);   // 8 bytes: state * 32 of the lane's 8 patterns
        for (int it = 0; it < iters; it++) {
            double f[8], g[8];
#pragma unroll
            for (int r = 0; r < 8; r++) f[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, acc[r], 0.0, 0, 0, 0);
            if (TIPS) {
#pragma unroll
                for (int r = 0; r < 8; r++) g[r] = *reinterpret_cast<const double*>(tb + rowOff + (unsigned)((t >> (8 * r)) & 0xff));
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) g[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, h[r], 0.0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) { acc[r] = f[r] * g[r]; if (SCALE) acc[r] *= inv; }
#pragma unroll
            for (int r = 0; r < 8; r++) asm volatile("" : "+v"(acc[r]));
            asm volatile("" : "+v"(t));
            t = (t + 0x2020202020202020ull) & 0x6060606060606060ull;
        }
        double s = 0; for (int r = 0; r < 8; r++) s += acc[r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int MODE, int SCALE, int TIPS>
double run(double* dout, const double* dtbl, size_t ldsBytes) {
    const int blocks = 256 * 4 * 3, iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, SCALE, TIPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipLaunchKernelGGL((k<MODE, SCALE, TIPS>), dim3(blocks), dim3(256), ldsBytes, 0, dout, dtbl, 50);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, SCALE, TIPS>), dim3(blocks), dim3(256), ldsBytes, 0, dout, dtbl, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-operations per SIMD = blocks * 4 waves * iters / 1024 SIMDs
    return ms * 1e6 / ((double)blocks * 4 * iters / 1024.0);
}

int main() {
    double h[40];
    for (int m = 0; m < 2; m++) for (int c = 0; c < 5; c++) for (int i = 0; i < 4; i++) h[m * 20 + c * 4 + i] = c < 4 ? (c == i ? 0.7 : 0.1) : 1.0;
    double *dtbl, *dout; hipMalloc(&dtbl, sizeof h); hipMalloc(&dout, (size_t)256 * 4 * 3 * 256 * 8);
    hipMemcpy(dtbl, h, sizeof h, hipMemcpyHostToDevice);
    for (size_t ldsBytes : {(size_t)36 << 10, (size_t)18 << 10}) {
        printf("-- %zu KiB of LDS per workgroup (%d waves per SIMD)\n", ldsBytes >> 10, ldsBytes > (20 << 10) ? 4 : 8);
        printf("V  mat-vec x tip column           : %6.1f ns per micro-operation and SIMD\n", run<0, 0, 1>(dout, dtbl, ldsBytes));
        printf("V  mat-vec x tip column x 1/scale : %6.1f\n", run<0, 1, 1>(dout, dtbl, ldsBytes));
        printf("V  mat-vec x mat-vec x 1/scale    : %6.1f\n", run<0, 1, 0>(dout, dtbl, ldsBytes));
        printf("M  mat-vec x tip column           : %6.1f\n", run<1, 0, 1>(dout, dtbl, ldsBytes));
        printf("M  mat-vec x tip column x 1/scale : %6.1f\n", run<1, 1, 1>(dout, dtbl, ldsBytes));
        printf("M  mat-vec x mat-vec x 1/scale    : %6.1f\n", run<1, 1, 0>(dout, dtbl, ldsBytes));
    }
    return 0;
}
