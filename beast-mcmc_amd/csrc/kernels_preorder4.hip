// kernels_preorder4.hip — 4 states: the gradient pass as ONE sweep per tree level.
//
// What the reference asks for, call by call (src/dr/evomodel/treedatalikelihood/preorder/AbstractBeagleGradientDelegate.java:
// 115-151, 207-221; AbstractBeagleBranchGradientDelegate.java:82-140): updatePrePartials with one operation per non-root node
//     pre(child)[j] = sum_i P_child[i][j] * ( pre(parent)[i] * sum_k P_sib[i][k] post(sib)[k] )
// and then calculateEdgeDifferentials over the same nodes
//     num = sum_c w_c sum_j pre[j] sum_k D_c[j][k] post[k],   den = sum_c w_c sum_j pre[j] post[j],   d lnL / d t = sum_p weight_p num / den.
// Done operation by operation that is 3 buffers moved per node for the pre-order pass and 2 more per edge (25 GB for 1000 taxa x
// 20 000 patterns).  The two children of a node share everything they read — pre(parent), post(a), post(b) — and an edge's
// derivative needs exactly the pre-order partial its node has just received, so k_preNode4 takes a NODE at a time: it reads
// pre(n), post(a), post(b) once, writes pre(a) and pre(b), and accumulates both edges' derivative sums on the way (the engine
// holds the pre-order operations back until the edge call arrives, engine_preorder.cpp).  A thread owns one pattern and walks
// the rate categories; partials move as 32-byte vectors (the direct kernels of kernels_preorder.hip read them 8 bytes at a
// 32-byte stride); the branch and differential matrices are wave-uniform.  k_edge4 is the edge derivative alone, in the same
// shape, for calls the fused pass does not cover (second derivatives on partials that exist already).
#include "kernels.h"

namespace mi355 {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d matvec4(const double* __restrict__ M, const v4d x) {          // y_i = sum_k M[i][k] x_k
    v4d y;
    y.x = M[0] * x.x + M[1] * x.y + M[2] * x.z + M[3] * x.w;
    y.y = M[4] * x.x + M[5] * x.y + M[6] * x.z + M[7] * x.w;
    y.z = M[8] * x.x + M[9] * x.y + M[10] * x.z + M[11] * x.w;
    y.w = M[12] * x.x + M[13] * x.y + M[14] * x.z + M[15] * x.w;
    return y;
}
__device__ __forceinline__ v4d matvecT4(const double* __restrict__ M, const v4d x) {         // y_j = sum_i M[i][j] x_i
    v4d y;
    y.x = M[0] * x.x + M[4] * x.y + M[8] * x.z + M[12] * x.w;
    y.y = M[1] * x.x + M[5] * x.y + M[9] * x.z + M[13] * x.w;
    y.z = M[2] * x.x + M[6] * x.y + M[10] * x.z + M[14] * x.w;
    y.w = M[3] * x.x + M[7] * x.y + M[11] * x.z + M[15] * x.w;
    return y;
}
// a compact tip as a partial: the unit vector of its state, all ones when the state is missing
__device__ __forceinline__ v4d tipVector(int s) {
    return s >= 4 ? v4d{1.0, 1.0, 1.0, 1.0} : v4d{s == 0 ? 1.0 : 0.0, s == 1 ? 1.0 : 0.0, s == 2 ? 1.0 : 0.0, s == 3 ? 1.0 : 0.0};
}
__device__ __forceinline__ double dot4(const v4d a, const v4d b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// per wave: blockSums[(slot * nBlocks + block64) * 2] = { sum weight deriv, sum weight deriv^2 } over its 64 patterns (fixed-shape
// butterfly: deterministic; the layout k_edgeFinal sums)
__device__ __forceinline__ void edgeBlockSum(double w1, double w2, double* __restrict__ blockSums, int slot, int nBlocks, int block64) {
    for (int off = 32; off > 0; off >>= 1) { w1 += __shfl_xor(w1, off, 64); w2 += __shfl_xor(w2, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        double* b = blockSums + ((size_t)slot * nBlocks + block64) * 2;
        b[0] = w1; b[1] = w2;
    }
}

__global__ __launch_bounds__(256) void k_preNode4(const PreNodeJob* __restrict__ jobs, const double* __restrict__ matrices,
                                                  const double* __restrict__ catWeights, const double* __restrict__ patternWeights,
                                                  double* __restrict__ blockSums, int P, int C, int nBlocks) {
    const PreNodeJob& jb = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < P;
    const int q = valid ? p : P - 1;                               // lanes past the end recompute the last pattern, store nothing
    const bool stA = jb.statesA != 0, stB = jb.statesB != 0;
    int sa = 4, sb = 4;
    if (stA) sa = gptr(reinterpret_cast<const uint8_t*>(jb.postA))[q];
    if (stB) sb = gptr(reinterpret_cast<const uint8_t*>(jb.postB))[q];
    const v4d MI355_GLOBAL* preN = gptr(reinterpret_cast<const v4d*>(jb.preParent));
    const v4d MI355_GLOBAL* postA = gptr(reinterpret_cast<const v4d*>(jb.postA));
    const v4d MI355_GLOBAL* postB = gptr(reinterpret_cast<const v4d*>(jb.postB));
    v4d MI355_GLOBAL* preA = gptr(reinterpret_cast<v4d*>(jb.preA));
    v4d MI355_GLOBAL* preB = gptr(reinterpret_cast<v4d*>(jb.preB));
    double numA = 0.0, denA = 0.0, numB = 0.0, denB = 0.0;
    for (int c = 0; c < C; c++) {
        const size_t e = (size_t)c * P + q;
        const double* MA = matrices + ((size_t)jb.matA * C + c) * 16;
        const double* MB = matrices + ((size_t)jb.matB * C + c) * 16;
        const v4d pn = preN[e];
        const v4d xa = stA ? tipVector(sa) : postA[e];
        const v4d xb = stB ? tipVector(sb) : postB[e];
        const v4d ua = matvec4(MA, xa), ub = matvec4(MB, xb);
        const v4d pa = matvecT4(MA, pn * ub), pb = matvecT4(MB, pn * ua);
        if (valid) {
            if (jb.preA) preA[e] = pa;
            if (jb.preB) preB[e] = pb;
        }
        const double w = catWeights[c];
        if (jb.slotA >= 0) { numA += w * dot4(pa, matvec4(matrices + ((size_t)jb.dA * C + c) * 16, xa)); denA += w * dot4(pa, xa); }
        if (jb.slotB >= 0) { numB += w * dot4(pb, matvec4(matrices + ((size_t)jb.dB * C + c) * 16, xb)); denB += w * dot4(pb, xb); }
    }
    const double pw = valid ? patternWeights[p] : 0.0;
    const int block64 = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (block64 < nBlocks) {
        if (jb.slotA >= 0) { const double d = valid ? numA / denA : 0.0; edgeBlockSum(pw * d, pw * d * d, blockSums, jb.slotA, nBlocks, block64); }
        if (jb.slotB >= 0) { const double d = valid ? numB / denB : 0.0; edgeBlockSum(pw * d, pw * d * d, blockSums, jb.slotB, nBlocks, block64); }
    }
}

void launchPreNodes4(hipStream_t stream, const PreNodeJob* dJobs, int nJobs, const double* matrices, const double* catWeights,
                     const double* patternWeights, double* blockSums, int P, int C) {
    if (nJobs <= 0) return;
    for (int o = 0; o < nJobs; o += 65535) {
        const int n = nJobs - o < 65535 ? nJobs - o : 65535;
        hipLaunchKernelGGL(k_preNode4, dim3((P + 255) / 256, n), dim3(256), 0, stream, dJobs + o, matrices, catWeights, patternWeights,
                           blockSums, P, C, edgeBlocks(P));
    }
}

__global__ __launch_bounds__(256) void k_edge4(const EdgeDesc* __restrict__ edges, const double* __restrict__ matrices,
                                               const double* __restrict__ catWeights, const double* __restrict__ patternWeights,
                                               double* __restrict__ perPattern, double* __restrict__ blockSums, int P, int C, int nBlocks) {
    const EdgeDesc& ed = edges[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < P;
    const int q = valid ? p : P - 1;
    const bool st = ed.postIsStates != 0;
    int s = 4;
    if (st) s = gptr(reinterpret_cast<const uint8_t*>(ed.post))[q];
    const v4d MI355_GLOBAL* pre = gptr(reinterpret_cast<const v4d*>(ed.pre));
    const v4d MI355_GLOBAL* post = gptr(reinterpret_cast<const v4d*>(ed.post));
    double num = 0.0, den = 0.0;
    for (int c = 0; c < C; c++) {
        const size_t e = (size_t)c * P + q;
        const v4d u = pre[e];
        const v4d x = st ? tipVector(s) : post[e];
        const double w = catWeights[c];
        num += w * dot4(u, matvec4(matrices + ((size_t)ed.dmat * C + c) * 16, x)); den += w * dot4(u, x);
    }
    const double d = valid ? num / den : 0.0;
    if (valid && perPattern) perPattern[(size_t)ed.slot * P + p] = d;
    const double pw = valid ? patternWeights[p] : 0.0;
    const int block64 = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (block64 < nBlocks) edgeBlockSum(pw * d, pw * d * d, blockSums, ed.slot, nBlocks, block64);
}

void launchEdgeDifferentials4(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* matrices, const double* catWeights,
                              const double* patternWeights, double* perPattern, double* blockSums, int P, int C) {
    if (nEdges <= 0) return;
    for (int o = 0; o < nEdges; o += 65535) {
        const int n = nEdges - o < 65535 ? nEdges - o : 65535;
        hipLaunchKernelGGL(k_edge4, dim3((P + 255) / 256, n), dim3(256), 0, stream, dEdges + o, matrices, catWeights, patternWeights,
                           perPattern, blockSums, P, C, edgeBlocks(P));
    }
}

}  // namespace mi355
