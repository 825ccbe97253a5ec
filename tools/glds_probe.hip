// glds_probe.hip — what an EXEC-masked LDS-DMA load does on gfx950 (tools/: hardware facts the kernels rest on).
// Question: with only lanes 0..19 enabled, does `global_load_lds_dwordx4` write lane l's 16 bytes at M0 + 16 l (by lane id,
// not compacted), leave the rest of LDS alone, and is the data readable by the issuing wave after s_waitcnt vmcnt(0) alone?
// Build: hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* src, double* out, int waveStride) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -1.0;
    __syncthreads();
    unsigned off = (unsigned)(w * waveStride + lane * 16);
    unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + 256 * w + 8));
    unsigned long long base = (unsigned long long)src;
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xfffff\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(0)"
                 :: "v"(off), "s"(base), "s"(dst) : "memory");
    for (int i = lane; i < 256; i += 64) out[w * 256 + i] = lds[256 * w + i];
}
int main() {
    const int W = 4, stride = 320;
    std::vector<double> h(W * 40 + 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)i;
    double *d, *o;
    hipMalloc(&d, h.size() * 8); hipMalloc(&o, W * 256 * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * W), 1024 * 8, 0, d, o, stride);
    std::vector<double> r(W * 256);
    hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < W; w++)
        for (int i = 0; i < 256; i++) {
            const double want = (i >= 8 && i < 48) ? (double)(w * 40 + (i - 8)) : -1.0;
            if (r[w * 256 + i] != want) { if (bad < 10) printf("wave %d lds[%d] = %g, expected %g\n", w, i, r[w * 256 + i], want); bad++; }
        }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAILED" : "ok: lane-linear by lane id, masked lanes write nothing, vmcnt(0) suffices", bad);
    return bad != 0;
}
