// kernels.hip — hand-written CDNA4 (gfx950) kernels of the tree-likelihood hot path.
//
// Data layout in HBM:
//   partials  double[C][P][S]  — pattern-major inside each rate category: consecutive lanes own consecutive
//                                patterns, so a wave's loads/stores of one category plane are one contiguous
//                                run (S=4: 64 lanes x 32 B = 2 KiB per plane per wave)
//   tip states uint8[P]        — 1 byte per pattern, value == S means missing/ambiguous
//   matrices  double[M][C][S][S]
//   scale     double[P]        — RAW per-pattern factors for per-node buffers (read mode needs 1/f, not exp),
//                                LOG for cumulative buffers
//
// What each kernel restates (reference = /root/reference):
//   k_prune4 / k_pruneGeneral   src/dr/oldevomodel/treelikelihood/GeneralLikelihoodCore.java:52-203
//                               (+ 4-state unrolling as NucleotideLikelihoodCore.java:54-270), rescale as
//                               AbstractLikelihoodCore.java:406-440 applied unconditionally (BEAGLE semantics)
//   k_root*                     GeneralLikelihoodCore.java:358-406
//   k_transition                src/dr/evomodel/substmodel/BaseSubstitutionModel.java:206-245,
//                               lib/beagle.jar!beagle/GeneralBeagleImpl#updateTransitionMatrices
//   k_accumulate                AbstractLikelihoodCore.java:442-458 as a persistent cumulative buffer
#include "kernels.h"
#include "root_site4.h"
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace mi355 {

bool grantDynamicLds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;     // (kernel, device) -> bytes granted so far
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    size_t& g = granted[std::make_pair(kernel, dev)];
    if (bytes <= g) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    g = bytes;
    return true;
}

// ------------------------------------------------------------------------------------------------
// host -> device copies of small arrays (kernels.h HostCopyList): a workgroup moves 4 KiB of one entry
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hostCopies(const HostCopyList L) { hostCopyBlock(L, blockIdx.x); }
void launchHostCopies(hipStream_t stream, const HostCopyList& list, int blocks) {
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_hostCopies, dim3((unsigned)blocks), dim3(256), 0, stream, list);
}

__global__ __launch_bounds__(64) void k_publish(const double* __restrict__ src, int n, double* __restrict__ dst, unsigned long long* flag, unsigned long long seq) {
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
    __threadfence_system();                        // every lane's values before the flag (one wave: program order does the rest)
    if (threadIdx.x == 0) __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
}
void launchPublish(hipStream_t stream, const double* src, int n, double* dst, unsigned long long* flag, unsigned long long seq) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream, src, n, dst, flag, seq);
}

// ------------------------------------------------------------------------------------------------
// transition matrices
// ------------------------------------------------------------------------------------------------
// complexEigen (an EIGEN_COMPLEX instance: BeagleTreeLikelihood.java:353-355, the asymmetric discrete-trait models): the
// eigenvalue array holds S real parts, then S imaginary parts, and U / U^-1 are the REAL block form — a conjugate pair
// a +/- b i sits in rows i, i+1 (imaginary parts b, -b) and contributes exp(a t) [cos b t, sin b t; -sin b t, cos b t] on those
// two rows of U^-1 (ComplexSubstitutionModel.java:121-173).  Row k of a pair is its FIRST row iff an even number of rows
// with a nonzero imaginary part precede it without a gap (the reference walks the rows in order and skips the second).
__device__ __forceinline__ double iexpEntry(const double* __restrict__ Ui, const double* __restrict__ lam, int S, int k, int j, double dist, int complexEigen) {
    const double im = complexEigen ? lam[S + k] : 0.0;
    if (im == 0.0) return Ui[k * S + j] * exp(dist * lam[k]);
    int run = 0;
    for (int q = k - 1; q >= 0 && lam[S + q] != 0.0; q--) run++;
    const int first = (run & 1) ? k - 1 : k;
    const double b = lam[S + first];
    const double expat = exp(dist * lam[first]), c = expat * cos(dist * b), sn = expat * sin(dist * b);
    return first == k ? c * Ui[k * S + j] + sn * Ui[(k + 1) * S + j] : c * Ui[k * S + j] - sn * Ui[(k - 1) * S + j];
}

__global__ __launch_bounds__(256) void k_transition(double* __restrict__ matrices, const double* __restrict__ eigen,
                                                    const double* __restrict__ rates, const int* __restrict__ dIdx,
                                                    const double* __restrict__ dLen, const int* __restrict__ dEig,
                                                    const int* __restrict__ dRate, int S, int C, int complexEigen) {
    extern __shared__ double sh[];            // iexp[S][S] | exp(dist lambda_k)[S]
    const int u = blockIdx.x, c = blockIdx.y;
    const size_t eigStride = (size_t)2 * S * S + (complexEigen ? 2 : 1) * S;
    const double* U = eigen + eigStride * dEig[u];
    const double* Ui = U + (size_t)S * S;
    const double* lam = Ui + (size_t)S * S;
    const double dist = dLen[u] * rates[(size_t)dRate[u] * C + c];
    // (one exponential per eigenvalue, not per entry: S instead of S x S of them — 3 721 at 61 states — the same values)
    double* ek = sh + (size_t)S * S;
    if (!complexEigen) {
        for (int k = threadIdx.x; k < S; k += blockDim.x) ek[k] = exp(dist * lam[k]);
        __syncthreads();
    }
    for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
        const int k = e / S;
        sh[e] = complexEigen ? iexpEntry(Ui, lam, S, k, e - k * S, dist, 1) : Ui[e] * ek[k];
    }
    __syncthreads();
    double* M = matrices + ((size_t)dIdx[u] * C + c) * S * S;
    // a thread forms a 4 x 4 block of entries: eight loads per sixteen multiply-adds instead of two per one (round 6: the 1 592 matrices
    // of a 61-state evaluation 127 -> ~40 us); every entry still sums over k in ascending order — the same bits as entry by entry
    const int nb = (S + 3) >> 2;
    for (int b = threadIdx.x; b < nb * nb; b += blockDim.x) {
        const int bi = b / nb, bj = b - bi * nb, i0 = 4 * bi, j0 = 4 * bj;
        const double* u0 = U + (size_t)(i0 < S ? i0 : S - 1) * S;
        const double* u1 = U + (size_t)(i0 + 1 < S ? i0 + 1 : S - 1) * S;
        const double* u2 = U + (size_t)(i0 + 2 < S ? i0 + 2 : S - 1) * S;
        const double* u3 = U + (size_t)(i0 + 3 < S ? i0 + 3 : S - 1) * S;
        const int c0 = j0 < S ? j0 : S - 1, c1 = j0 + 1 < S ? j0 + 1 : S - 1, c2 = j0 + 2 < S ? j0 + 2 : S - 1, c3 = j0 + 3 < S ? j0 + 3 : S - 1;
        double a[4][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        for (int k = 0; k < S; k++) {
            const double* row = sh + (size_t)k * S;
            const double v[4] = {row[c0], row[c1], row[c2], row[c3]};
            const double w[4] = {u0[k], u1[k], u2[k], u3[k]};
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) a[r][q] += w[r] * v[q];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i0 + r < S && j0 + q < S) M[(size_t)(i0 + r) * S + j0 + q] = a[r][q] > 0.0 ? a[r][q] : 0.0;
    }
}

// More than 90 states (S x S doubles exceed what a workgroup may ask of the LDS): per eigenvalue only its exponential — for
// a complex pair exp(a t) cos(b t), exp(a t) sin(b t) and which row of the pair it is — stays in LDS; a thread forms one matrix
// entry, summing over k in the same order as k_transition.
__global__ __launch_bounds__(256) void k_transitionBig(double* __restrict__ matrices, const double* __restrict__ eigen,
                                                       const double* __restrict__ rates, const int* __restrict__ dIdx,
                                                       const double* __restrict__ dLen, const int* __restrict__ dEig,
                                                       const int* __restrict__ dRate, int S, int C, int complexEigen) {
    extern __shared__ double sh[];            // ec[S] | es[S] | role[S] (0 real, 1 first row of a pair, 2 second)
    double* ec = sh; double* es = sh + S; double* role = sh + 2 * S;
    const int u = blockIdx.x, c = blockIdx.y;
    const size_t eigStride = (size_t)2 * S * S + (complexEigen ? 2 : 1) * S;
    const double* U = eigen + eigStride * dEig[u];
    const double* Ui = U + (size_t)S * S;
    const double* lam = Ui + (size_t)S * S;
    const double dist = dLen[u] * rates[(size_t)dRate[u] * C + c];
    for (int k = threadIdx.x; k < S; k += blockDim.x) {
        const double im = complexEigen ? lam[S + k] : 0.0;
        if (im == 0.0) { ec[k] = exp(dist * lam[k]); es[k] = 0.0; role[k] = 0.0; continue; }
        int run = 0;
        for (int q = k - 1; q >= 0 && lam[S + q] != 0.0; q--) run++;
        const int first = (run & 1) ? k - 1 : k;
        const double b = lam[S + first], expat = exp(dist * lam[first]);
        ec[k] = expat * cos(dist * b); es[k] = expat * sin(dist * b); role[k] = first == k ? 1.0 : 2.0;
    }
    __syncthreads();
    double* M = matrices + ((size_t)dIdx[u] * C + c) * S * S;
    for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
        const int i = e / S, j = e - i * S;
        double s = 0.0;
        for (int k = 0; k < S; k++) {
            const double r = role[k];
            const double ie = r == 0.0 ? Ui[k * S + j] * ec[k]
                            : r == 1.0 ? ec[k] * Ui[k * S + j] + es[k] * Ui[(k + 1) * S + j]
                                       : ec[k] * Ui[k * S + j] - es[k] * Ui[(k - 1) * S + j];
            s += U[i * S + k] * ie;
        }
        M[e] = s > 0.0 ? s : 0.0;
    }
}

// 4 states: one THREAD per (branch, category) — a workgroup per matrix would be 64 lanes for 16 outputs (and, for the 12 872
// matrices of a four-partition 1610-taxon evaluation, 51 488 workgroups).  Same arithmetic and summation order as k_transition.
__global__ __launch_bounds__(256) void k_transition4(double* __restrict__ matrices, const double* __restrict__ eigen,
                                                     const double* __restrict__ rates, const int* __restrict__ dIdx,
                                                     const double* __restrict__ dLen, const int* __restrict__ dEig,
                                                     const int* __restrict__ dRate, int count, int C, int complexEigen) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count * C) return;
    const int u = t / C, c = t - u * C;
    const double* U = eigen + (size_t)(complexEigen ? 40 : 36) * dEig[u];
    const double* Ui = U + 16;
    const double* lam = U + 32;
    const double dist = dLen[u] * rates[(size_t)dRate[u] * C + c];
    double ie[16];
    if (complexEigen) { for (int e = 0; e < 16; e++) ie[e] = iexpEntry(Ui, lam, 4, e >> 2, e & 3, dist, 1); }
    else for (int k = 0; k < 4; k++) { const double ex = exp(dist * lam[k]); for (int j = 0; j < 4; j++) ie[k * 4 + j] = Ui[k * 4 + j] * ex; }
    double* M = matrices + ((size_t)dIdx[u] * C + c) * 16;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += U[i * 4 + k] * ie[k * 4 + j];
            M[i * 4 + j] = s > 0.0 ? s : 0.0;
        }
}


__global__ __launch_bounds__(256) void k_transition4Fused(double* __restrict__ matrices, const double* __restrict__ eigSrc,
                                                          const double* __restrict__ ratesSrc, const int* __restrict__ idx,
                                                          const double* __restrict__ len, int count, int C, int complexEigen,
                                                          const HostCopyList L, int transitionBlocks) {
    if ((int)blockIdx.x >= transitionBlocks) { hostCopyBlock(L, blockIdx.x - (unsigned)transitionBlocks); return; }
    __shared__ double sEig[40], sRate[16];
    const int nEig = complexEigen ? 40 : 36;
    // (the branch's length and matrix index are asked for BEFORE the barrier: they come out of the host's staging ring, as a freshly set
    // eigen system does, and two round trips across PCIe one after the other were a fifth of this kernel's 10 us)
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool live = t < count * C;
    const int u = live ? t / C : 0, c = live ? t - u * C : 0;
    const double myLen = len[u];
    const int myIdx = idx[u];
    if ((int)threadIdx.x < nEig) sEig[threadIdx.x] = eigSrc[threadIdx.x];
    else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + C && C <= 16) sRate[threadIdx.x - 64] = ratesSrc[threadIdx.x - 64];
    __syncthreads();
    if (!live) return;
    const double* U = sEig;
    const double* Ui = U + 16;
    const double* lam = U + 32;
    const double dist = myLen * (C <= 16 ? sRate[c] : ratesSrc[c]);
    double ie[16];
    if (complexEigen) { for (int e = 0; e < 16; e++) ie[e] = iexpEntry(Ui, lam, 4, e >> 2, e & 3, dist, 1); }
    else for (int k = 0; k < 4; k++) { const double ex = exp(dist * lam[k]); for (int j = 0; j < 4; j++) ie[k * 4 + j] = Ui[k * 4 + j] * ex; }
    double* M = matrices + ((size_t)myIdx * C + c) * 16;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += U[i * 4 + k] * ie[k * 4 + j];
            M[i * 4 + j] = s > 0.0 ? s : 0.0;
        }
}

void launchTransitionMatrices4Fused(hipStream_t stream, double* matrices, const double* eigSrc, const double* ratesSrc, const int* idx,
                                    const double* len, int count, int C, bool complexEigen, const HostCopyList& pending, int copyBlocks) {
    if (count <= 0) return;
    const int tb = (int)(((size_t)count * C + 255) / 256);
    hipLaunchKernelGGL(k_transition4Fused, dim3((unsigned)(tb + copyBlocks)), dim3(256), 0, stream, matrices, eigSrc, ratesSrc, idx, len,
                       count, C, complexEigen ? 1 : 0, pending, tb);
}

void launchTransitionMatrices(hipStream_t stream, double* matrices, const double* eigen, const double* rates,
                              const int* dIdx, const double* dLen, const int* dEig, const int* dRate,
                              int count, int S, int C, bool complexEigen) {
    if (count <= 0) return;
    if (S == 4) {
        hipLaunchKernelGGL(k_transition4, dim3((unsigned)(((size_t)count * C + 255) / 256)), dim3(256), 0, stream,
                           matrices, eigen, rates, dIdx, dLen, dEig, dRate, count, C, complexEigen ? 1 : 0);
        return;
    }
    const int threads = S * S >= 256 ? 256 : 64;
    if (((size_t)S * S + S) * sizeof(double) > 64 * 1024) {      // more than 90 states: exp(lambda t) per eigenvalue in LDS, not the S x S product
        hipLaunchKernelGGL(k_transitionBig, dim3(count, C), dim3(256), (size_t)3 * S * sizeof(double), stream,
                           matrices, eigen, rates, dIdx, dLen, dEig, dRate, S, C, complexEigen ? 1 : 0);
        return;
    }
    hipLaunchKernelGGL(k_transition, dim3(count, C), dim3(threads), ((size_t)S * S + S) * sizeof(double), stream,
                       matrices, eigen, rates, dIdx, dLen, dEig, dRate, S, C, complexEigen ? 1 : 0);
}

__global__ __launch_bounds__(256) void k_addMatrices(double* __restrict__ matrices, const int* __restrict__ dFirst, const int* __restrict__ dSecond,
                                                     const int* __restrict__ dResult, int elems) {
    const double* a = matrices + (size_t)dFirst[blockIdx.x] * elems;
    const double* b = matrices + (size_t)dSecond[blockIdx.x] * elems;
    double* r = matrices + (size_t)dResult[blockIdx.x] * elems;
    for (int e = threadIdx.x; e < elems; e += 256) r[e] = a[e] + b[e];
}
void launchAddMatrices(hipStream_t stream, double* matrices, const int* dFirst, const int* dSecond, const int* dResult, int count, int S, int C) {
    if (count > 0) hipLaunchKernelGGL(k_addMatrices, dim3(count), dim3(256), 0, stream, matrices, dFirst, dSecond, dResult, C * S * S);
}

__global__ __launch_bounds__(256) void k_convolve(double* __restrict__ matrices, const int* __restrict__ dFirst,
                                                  const int* __restrict__ dSecond, const int* __restrict__ dResult,
                                                  int S, int C) {
    const int u = blockIdx.x, c = blockIdx.y;
    const double* A = matrices + ((size_t)dFirst[u] * C + c) * S * S;
    const double* B = matrices + ((size_t)dSecond[u] * C + c) * S * S;
    double* R = matrices + ((size_t)dResult[u] * C + c) * S * S;
    for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
        const int i = e / S, j = e - i * S;
        double s = 0.0;
        for (int k = 0; k < S; k++) s += A[i * S + k] * B[k * S + j];
        R[e] = s;
    }
}

void launchConvolveMatrices(hipStream_t stream, double* matrices, const int* dFirst, const int* dSecond,
                            const int* dResult, int count, int S, int C) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_convolve, dim3(count, C), dim3(S * S >= 256 ? 256 : 64), 0, stream,
                       matrices, dFirst, dSecond, dResult, S, C);
}

// (one 64-thread workgroup per matrix moved a 61-state matrix block — 119 KB — 8 bytes per thread and step: 72 us for the ~600 snapshots
// of a config-C evaluation; now a matrix is cut into pieces of 2 048 doubles, a workgroup of 256 threads each)
__global__ __launch_bounds__(256) void k_snapshot(double* __restrict__ matrices, const int* __restrict__ srcDst, int elems) {
    const double* s = matrices + (size_t)srcDst[2 * blockIdx.x] * elems;
    double* d = matrices + (size_t)srcDst[2 * blockIdx.x + 1] * elems;
    const int e0 = blockIdx.y * 2048, e1 = e0 + 2048 < elems ? e0 + 2048 : elems;
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) d[e] = s[e];
}

void launchSnapshotMatrices(hipStream_t stream, double* matrices, const int* dSrcDst, int n, int elems) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_snapshot, dim3(n, (elems + 2047) / 2048), dim3(256), 0, stream, matrices, dSrcDst, elems);
}

// ------------------------------------------------------------------------------------------------
// general state count (20, 61, ...): VALU kernel, any S <= 256 and any C.  A workgroup owns a strided set
// of pattern tiles of one op; per category it stages both transposed matrices in LDS once and then
// streams its tiles through: thread (pl, i) computes parent state i of pattern pl.
// Rescaling in write mode is two-phase (running max per pattern, then a divide pass over the
// workgroup's own tiles) because the max spans categories.
// ------------------------------------------------------------------------------------------------
constexpr int GEN_BLOCK = 256;

// STAGED = false (more than ~90 states: two transposed S x S matrices no longer fit a CU's LDS): the matrices are read where
// they lie, row i of M by the thread that owns parent state i — from L2 after the first touch; a correctness path for the
// large discrete-trait state spaces (GeneralLikelihoodCore.java:41-50 takes any stateCount), same arithmetic and order.
template <bool STAGED>
__global__ __launch_bounds__(GEN_BLOCK) void k_pruneGeneral(const OpDesc* __restrict__ ops, const double* __restrict__ matrices,
                                                            int P, int S, int C) {
    extern __shared__ double sh[];
    const int ppb = GEN_BLOCK / S;                   // patterns per pass
    double* mT1 = sh;                                // [S+1][S]  (row S = ones: unknown state)
    double* mT2 = mT1 + (STAGED ? (size_t)(S + 1) * S : 0);
    double* x1 = mT2 + (STAGED ? (size_t)(S + 1) * S : 0);          // [ppb][S]
    double* x2 = x1 + (size_t)ppb * S;
    double* red = x2 + (size_t)ppb * S;              // [ppb][S] products for the max reduction
    const OpDesc op = ops[blockIdx.y];
    const int range = op.pEnd - op.pStart;
    const int tiles = (range + ppb - 1) / ppb;
    if ((int)blockIdx.x >= tiles) return;
    const int pl = threadIdx.x / S, i = threadIdx.x - pl * S;
    const bool lane = pl < ppb;
    const bool st1 = op.kind & KIND_STATES1, st2 = op.kind & KIND_STATES2;
    const uint8_t* s1 = reinterpret_cast<const uint8_t*>(op.child1);
    const uint8_t* s2 = reinterpret_cast<const uint8_t*>(op.child2);
    const double* c1 = reinterpret_cast<const double*>(op.child1);
    const double* c2 = reinterpret_cast<const double*>(op.child2);

    for (int c = 0; c < C; c++) {
        __syncthreads();
        const double* M1 = matrices + ((size_t)op.mat1 * C + c) * S * S;
        const double* M2 = matrices + ((size_t)op.mat2 * C + c) * S * S;
        if (STAGED) {
            for (int e = threadIdx.x; e < S * S; e += GEN_BLOCK) {
                const int r = e / S, q = e - r * S;
                mT1[q * S + r] = M1[e];
                mT2[q * S + r] = M2[e];
            }
            for (int e = threadIdx.x; e < S; e += GEN_BLOCK) { mT1[S * S + e] = 1.0; mT2[S * S + e] = 1.0; }
        }
        for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
            const int pBase = op.pStart + tile * ppb;
            const int np = min(ppb, op.pEnd - pBase);
            __syncthreads();
            if (!st1) for (int e = threadIdx.x; e < np * S; e += GEN_BLOCK) x1[e] = c1[((size_t)c * P + pBase) * S + e];
            if (!st2) for (int e = threadIdx.x; e < np * S; e += GEN_BLOCK) x2[e] = c2[((size_t)c * P + pBase) * S + e];
            __syncthreads();
            const bool act = lane && pl < np;
            double v = 0.0;
            if (act) {
                const int p = pBase + pl;
                double sum1, sum2;
                if (STAGED) {
                    if (st1) sum1 = mT1[(int)s1[p] * S + i];
                    else { sum1 = 0.0; for (int j = 0; j < S; j++) sum1 += mT1[j * S + i] * x1[pl * S + j]; }
                    if (st2) sum2 = mT2[(int)s2[p] * S + i];
                    else { sum2 = 0.0; for (int j = 0; j < S; j++) sum2 += mT2[j * S + i] * x2[pl * S + j]; }
                } else {
                    const double* r1 = M1 + (size_t)i * S;
                    const double* r2 = M2 + (size_t)i * S;
                    if (st1) { const int t = (int)s1[p]; sum1 = t < S ? r1[t] : 1.0; }
                    else { sum1 = 0.0; for (int j = 0; j < S; j++) sum1 += r1[j] * x1[pl * S + j]; }
                    if (st2) { const int t = (int)s2[p]; sum2 = t < S ? r2[t] : 1.0; }
                    else { sum2 = 0.0; for (int j = 0; j < S; j++) sum2 += r2[j] * x2[pl * S + j]; }
                }
                v = sum1 * sum2;
                if (op.scaleRead) v *= 1.0 / op.scaleRead[p];
                op.dest[((size_t)c * P + p) * S + i] = v;
            }
            if (op.scaleWrite) {
                if (lane) red[pl * S + i] = act ? v : 0.0;
                __syncthreads();
                if (act && i == 0) {
                    double m = 0.0;
                    for (int j = 0; j < S; j++) m = fmax(m, red[pl * S + j]);
                    const int p = pBase + pl;
                    if (c > 0) m = fmax(m, op.scaleWrite[p]);
                    op.scaleWrite[p] = m;
                }
            }
        }
    }
    if (op.scaleWrite) {
        __syncthreads();
        for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
            const int pBase = op.pStart + tile * ppb;
            const int np = min(ppb, op.pEnd - pBase);
            __syncthreads();
            if (lane && pl < np && i == 0) {
                const int p = pBase + pl;
                double m = op.scaleWrite[p];
                if (!(m > 0.0)) { m = 1.0; op.scaleWrite[p] = 1.0; }
                red[pl] = 1.0 / m;
            }
            __syncthreads();
            if (lane && pl < np) {
                const int p = pBase + pl;
                const double inv = red[pl];
                for (int c = 0; c < C; c++) op.dest[((size_t)c * P + p) * S + i] *= inv;
            }
        }
    }
}

int pruneBlocksForRange(int S, int range) {
    const int ppb = GEN_BLOCK / S;
    const int tiles = (range + ppb - 1) / ppb;
    // enough workgroups to fill the chip, few enough that the per-category matrix staging is amortised
    const int cap = 1024;
    return tiles < cap ? tiles : cap;
}

void launchPruneLevel(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices,
                      int P, int S, int C, int maxRange) {
    if (nOps <= 0 || maxRange <= 0) return;
    const int ppb = GEN_BLOCK / S;
    const size_t ldsStaged = ((size_t)2 * (S + 1) * S + (size_t)3 * ppb * S) * sizeof(double);
    const bool staged = ldsStaged <= 150 * 1024;         // two transposed matrices fit a CU's 160 KiB up to ~95 states
    const size_t lds = staged ? ldsStaged : (size_t)3 * ppb * S * sizeof(double);
    int blocks = pruneBlocksForRange(S, maxRange);
    // keep total workgroups around a few per CU when many ops share the launch
    if (nOps > 1) { int per = (4096 + nOps - 1) / nOps; if (per < 1) per = 1; if (blocks > per) blocks = per; }
    if (staged) {
        if (lds > 48 * 1024 && !grantDynamicLds(reinterpret_cast<const void*>(k_pruneGeneral<true>), lds)) return;   // S = 61 needs 66 KB of the CU's 160 KB
        hipLaunchKernelGGL(k_pruneGeneral<true>, dim3(blocks, nOps), dim3(GEN_BLOCK), lds, stream, dOps, matrices, P, S, C);
    } else
        hipLaunchKernelGGL(k_pruneGeneral<false>, dim3(blocks, nOps), dim3(GEN_BLOCK), lds, stream, dOps, matrices, P, S, C);
}

// ------------------------------------------------------------------------------------------------
// root: integrate over categories and states, log, add cumulative scale, weighted deterministic sum
// ------------------------------------------------------------------------------------------------
constexpr int ROOT_BLOCK = 256;
struct __attribute__((aligned(32))) d4 { double x, y, z, w; };

__device__ __forceinline__ double blockSum(double v, double* sh) {
    // fixed-shape tree: wave shuffle (64 lanes) then LDS across the 4 waves — same order every run
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) { for (int k = 0; k < ROOT_BLOCK / 64; k++) t += sh[k]; }
    return t;
}

__global__ __launch_bounds__(ROOT_BLOCK) void k_rootSite(const double* __restrict__ root, const double* __restrict__ catWeights,
                                                         const double* __restrict__ freqs, const double* __restrict__ cum,
                                                         int cumIsRaw, const double* __restrict__ patternWeights,
                                                         double* __restrict__ siteLogL, double* __restrict__ blockSums,
                                                         int P, int S, int C, int pStart, int pEnd,
                                                         unsigned* counter, double* __restrict__ out, unsigned long long* flag, unsigned long long seq) {
    __shared__ double sh[ROOT_BLOCK / 64];
    __shared__ bool lastBlock;
    const int p = pStart + blockIdx.x * ROOT_BLOCK + threadIdx.x;
    double contrib = 0.0;
    if (p < pEnd) {
        double sum = 0.0;
        if (S == 4) {
            for (int c = 0; c < C; c++) {
                const d4 v = *reinterpret_cast<const d4*>(root + ((size_t)c * P + p) * 4);
                sum += catWeights[c] * (freqs[0] * v.x + freqs[1] * v.y + freqs[2] * v.z + freqs[3] * v.w);
            }
        } else {
            for (int c = 0; c < C; c++) {
                const double* r = root + ((size_t)c * P + p) * S;
                double s = 0.0;
                for (int i = 0; i < S; i++) s += freqs[i] * r[i];
                sum += catWeights[c] * s;
            }
        }
        double site = log(sum);
        if (cum) site += cumIsRaw ? log(cum[p]) : cum[p];
        siteLogL[p] = site;
        contrib = site * patternWeights[p];
    }
    const double t = blockSum(contrib, sh);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = t;
    if (!counter) return;                          // (the sum over the blocks is a launch of its own: k_rootFinal)
    // The workgroup that finishes LAST adds the block sums up, in index order (the same fixed-order sum k_rootFinal forms: the
    // result does not depend on which workgroup that is) — one launch per evaluation less on a path where a launch is 4 us of
    // the GPU's time and as much of the host's.
    if (threadIdx.x == 0) {
        __threadfence();                                                       // my block sum before my ticket
        lastBlock = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!lastBlock) return;
    __threadfence();
    double v = 0.0;
    for (int k = threadIdx.x; k < (int)gridDim.x; k += ROOT_BLOCK)
        v += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(blockSums) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const double total = blockSum(v, sh);
    if (threadIdx.x == 0) {
        out[0] = total;
        *counter = 0u;                                                         // ready for the next launch (same stream: ordered)
        if (flag) { __threadfence_system(); __atomic_store_n(flag, seq, __ATOMIC_RELEASE); }
    }
}

// `flag` (nullable): a word next to `out` in host-visible memory that receives `seq` after the sum — the host polls it
// instead of paying a stream-synchronisation wake-up once per evaluation.
__global__ __launch_bounds__(ROOT_BLOCK) void k_rootFinal(const double* __restrict__ blockSums, int n, double* __restrict__ out,
                                                          unsigned long long* flag, unsigned long long seq) {
    __shared__ double sh[ROOT_BLOCK / 64];
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += ROOT_BLOCK) v += blockSums[k];
    const double t = blockSum(v, sh);
    if (threadIdx.x == 0) {
        out[0] = t;
        if (flag) { __threadfence_system(); __atomic_store_n(flag, seq, __ATOMIC_RELEASE); }
    }
}

// 4-state walk instances (root_site4.h): a wave per 128 patterns with the assembly loop's lane map, so that this launch and the
// epilogue of the walk's root slice (kernels_walk4.hip) give the same bits
__global__ __launch_bounds__(256) void k_rootSite4W(const double* __restrict__ root, const double* __restrict__ catWeights,
                                                    const double* __restrict__ freqs, const double* __restrict__ cum, int cumIsRaw,
                                                    const double* __restrict__ patternWeights, double* __restrict__ siteLogL,
                                                    double* __restrict__ blockSums, int P, int C, int pStart, int pEnd, int groups,
                                                    unsigned* counter, double* __restrict__ out, unsigned long long* flag, unsigned long long seq) {
    const int lane = threadIdx.x & 63, group = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (group >= groups) return;
    const int p0 = pStart + group * 128, q = lane >> 1, r = lane & 1;
    const int pa = p0 + q + 32 * r, pb = pa + 64;
    const int la = pa < pEnd ? pa : pEnd - 1, lb = pb < pEnd ? pb : pEnd - 1;
    double sumA = 0.0, sumB = 0.0;
    for (int c = 0; c < C; c++) {
        const d4 a = *reinterpret_cast<const d4*>(root + ((size_t)c * P + la) * 4);
        const d4 b = *reinterpret_cast<const d4*>(root + ((size_t)c * P + lb) * 4);
        sumA = __builtin_fma(catWeights[c], rootDot4(freqs, a.x, a.y, a.z, a.w), sumA);
        sumB = __builtin_fma(catWeights[c], rootDot4(freqs, b.x, b.y, b.z, b.w), sumB);
    }
    const double g = rootWaveSum(rootFinishPair(sumA, sumB, pa, pb, pEnd, cum, cumIsRaw, patternWeights, siteLogL));
    rootPublishGroup(g, lane, group, groups, blockSums, counter, out, flag, seq);
}

void launchRootLogLikelihood4W(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                               const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                               double* blockSums, double* out, int P, int C, int pStart, int pEnd,
                               unsigned long long* flag, unsigned long long seq, unsigned* counter) {
    const int groups = (pEnd - pStart + 127) / 128;
    hipLaunchKernelGGL(k_rootSite4W, dim3((groups + 3) / 4), dim3(256), 0, stream, root, catWeights, freqs, cum, cumIsRaw,
                       patternWeights, siteLogL, blockSums, P, C, pStart, pEnd, groups, counter, out, flag, seq);
}

__global__ void k_rootFinalParts(const double* __restrict__ blockSums, const RootParts parts, double* __restrict__ out, unsigned long long* flag, unsigned long long seq);
// ... and for up to ROOT_MAX_PARTS partitions in one launch (grid row = partition): the launch of its own beside the epilogues of the
// partitions' top slices (kernels_walk4.hip, RootFusedParts) — same functions, same order, same bits
__global__ __launch_bounds__(256) void k_rootSite4WParts(const RootParts parts, const double* __restrict__ patternWeights, double* __restrict__ siteLogL,
                                                         double* __restrict__ blockSums, int P, int C, int totalGroups, unsigned* counter,
                                                         double* __restrict__ out, unsigned long long* flag, unsigned long long seq) {
    const RootPart& q = parts.p[blockIdx.y];
    const int lane = threadIdx.x & 63, group = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int groups = (q.pEnd - q.pStart + 127) / 128;
    if (group >= groups) return;
    const int p0 = q.pStart + group * 128, h = lane >> 1, r = lane & 1;
    const int pa = p0 + h + 32 * r, pb = pa + 64;
    const int la = pa < q.pEnd ? pa : q.pEnd - 1, lb = pb < q.pEnd ? pb : q.pEnd - 1;
    double sumA = 0.0, sumB = 0.0;
    for (int c = 0; c < C; c++) {
        const d4 a = *reinterpret_cast<const d4*>(q.root + ((size_t)c * P + la) * 4);
        const d4 b = *reinterpret_cast<const d4*>(q.root + ((size_t)c * P + lb) * 4);
        sumA = __builtin_fma(q.catWeights[c], rootDot4(q.freqs, a.x, a.y, a.z, a.w), sumA);
        sumB = __builtin_fma(q.catWeights[c], rootDot4(q.freqs, b.x, b.y, b.z, b.w), sumB);
    }
    const double g = rootWaveSum(rootFinishPair(sumA, sumB, pa, pb, q.pEnd, q.cum, q.cumIsRaw, patternWeights, siteLogL));
    rootPublishGroupParts(g, lane, q.blockOff + group, totalGroups, parts.n, parts,
                          [](const RootParts& s, int i) { return (s.p[i].pEnd - s.p[i].pStart + 127) / 128; }, [](const RootParts& s, int i) { return s.p[i].blockOff; },
                          blockSums, counter, out, flag, seq);
}
void launchRootLogLikelihoodParts4W(hipStream_t stream, const RootParts& parts, const double* patternWeights, double* siteLogL, double* blockSums,
                                    double* out, int P, int C, unsigned long long* flag, unsigned long long seq, unsigned* counter) {
    int maxGroups = 1, total = 0;
    for (int k = 0; k < parts.n; k++) { const int g = (std::max(0, parts.p[k].pEnd - parts.p[k].pStart) + 127) / 128; maxGroups = std::max(maxGroups, g); total += g; }
    if (total == 0) {                                  // nothing but empty ranges (a shard that holds none of these partitions' patterns): zeros, and the word
        hipLaunchKernelGGL(k_rootFinalParts, dim3(1), dim3(ROOT_BLOCK), 0, stream, blockSums, parts, out, flag, seq);
        return;
    }
    hipLaunchKernelGGL(k_rootSite4WParts, dim3((maxGroups + 3) / 4, parts.n), dim3(256), 0, stream, parts, patternWeights, siteLogL, blockSums, P, C, total,
                       counter, out, flag, seq);
}

void launchRootLogLikelihood(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                             const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                             double* blockSums, double* out, int P, int S, int C, int pStart, int pEnd,
                             unsigned long long* flag, unsigned long long seq, unsigned* counter) {
    const int n = (pEnd - pStart + ROOT_BLOCK - 1) / ROOT_BLOCK;
    hipLaunchKernelGGL(k_rootSite, dim3(n), dim3(ROOT_BLOCK), 0, stream, root, catWeights, freqs, cum, cumIsRaw,
                       patternWeights, siteLogL, blockSums, P, S, C, pStart, pEnd, counter, out, flag, seq);
    if (!counter) hipLaunchKernelGGL(k_rootFinal, dim3(1), dim3(ROOT_BLOCK), 0, stream, blockSums, n, out, flag, seq);
}

// the sums of parts.n partitions from their workgroups' sums, one after the other, by ONE workgroup (k_rootFinalParts, and the last
// workgroup of k_rootSiteParts: the same order of additions, the same bits)
template <bool ATOMIC>
__device__ __forceinline__ void rootFinalPartsBody(const double* __restrict__ blockSums, const RootParts& parts, double* __restrict__ out, double* sh) {
    for (int k = 0; k < parts.n; k++) {
        const RootPart& q = parts.p[k];
        const int n = (q.pEnd - q.pStart + ROOT_BLOCK - 1) / ROOT_BLOCK;
        double v = 0.0;
        for (int j = threadIdx.x; j < n; j += ROOT_BLOCK)
            v += ATOMIC ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(blockSums) + q.blockOff + j,
                                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        : blockSums[q.blockOff + j];
        __syncthreads();
        const double t = blockSum(v, sh);
        if (threadIdx.x == 0) out[k] = t;
    }
}

// counter (nullable): as k_rootSite's — the workgroup that finishes last forms the partitions' sums (no k_rootFinalParts launch)
__global__ __launch_bounds__(ROOT_BLOCK) void k_rootSiteParts(const RootParts parts, const double* __restrict__ patternWeights,
                                                              double* __restrict__ siteLogL, double* __restrict__ blockSums, int P, int S, int C,
                                                              unsigned* counter, double* __restrict__ out, unsigned long long* flag, unsigned long long seq) {
    __shared__ double sh[ROOT_BLOCK / 64];
    __shared__ bool lastBlock;
    const RootPart& q = parts.p[blockIdx.y];
    const int p = q.pStart + blockIdx.x * ROOT_BLOCK + threadIdx.x;
    const bool active = q.pStart + (int)blockIdx.x * ROOT_BLOCK < q.pEnd;    // (the whole workgroup)
    if (!active && !counter) return;
    double contrib = 0.0;
    if (active && p < q.pEnd) {
        double sum = 0.0;
        for (int c = 0; c < C; c++) {
            const double* r = q.root + ((size_t)c * P + p) * S;
            double s = 0.0;
            if (S == 4) {           // (the multiply-adds of root_site4.h: a site value has the same bits whichever kernel formed it)
                const d4 v = *reinterpret_cast<const d4*>(r);
                sum = __builtin_fma(q.catWeights[c], rootDot4(q.freqs, v.x, v.y, v.z, v.w), sum);
                continue;
            }
            for (int i = 0; i < S; i++) s += q.freqs[i] * r[i];
            sum += q.catWeights[c] * s;
        }
        double site = log(sum);
        if (q.cum) site += q.cumIsRaw ? log(q.cum[p]) : q.cum[p];
        siteLogL[p] = site;
        contrib = site * patternWeights[p];
    }
    if (active) {                                                            // (uniform over the workgroup)
        const double t = blockSum(contrib, sh);
        if (threadIdx.x == 0) blockSums[q.blockOff + blockIdx.x] = t;
    }
    if (!counter) return;
    if (threadIdx.x == 0) {
        __threadfence();                                                       // my block sum before my ticket
        lastBlock = atomicAdd(counter, 1u) == gridDim.x * gridDim.y - 1;       // (every workgroup of the grid takes one, the idle ones too)
    }
    __syncthreads();
    if (!lastBlock) return;
    __threadfence();
    rootFinalPartsBody<true>(blockSums, parts, out, sh);
    if (threadIdx.x == 0) {
        *counter = 0u;                                                         // ready for the next launch (same stream: ordered)
        if (flag) { __threadfence_system(); __atomic_store_n(flag, seq, __ATOMIC_RELEASE); }
    }
}

__global__ __launch_bounds__(ROOT_BLOCK) void k_rootFinalParts(const double* __restrict__ blockSums, const RootParts parts, double* __restrict__ out,
                                                               unsigned long long* flag, unsigned long long seq) {
    __shared__ double sh[ROOT_BLOCK / 64];
    rootFinalPartsBody<false>(blockSums, parts, out, sh);
    if (threadIdx.x == 0 && flag) { __threadfence_system(); __atomic_store_n(flag, seq, __ATOMIC_RELEASE); }
}

void launchRootLogLikelihoodParts(hipStream_t stream, const RootParts& parts, const double* patternWeights, double* siteLogL, double* blockSums,
                                  double* out, int P, int S, int C, unsigned long long* flag, unsigned long long seq, unsigned* counter) {
    int maxBlocks = 1;
    for (int k = 0; k < parts.n; k++) maxBlocks = std::max(maxBlocks, (parts.p[k].pEnd - parts.p[k].pStart + ROOT_BLOCK - 1) / ROOT_BLOCK);
    hipLaunchKernelGGL(k_rootSiteParts, dim3(maxBlocks, parts.n), dim3(ROOT_BLOCK), 0, stream, parts, patternWeights, siteLogL, blockSums, P, S, C,
                       counter, out, flag, seq);
    if (!counter) hipLaunchKernelGGL(k_rootFinalParts, dim3(1), dim3(ROOT_BLOCK), 0, stream, blockSums, parts, out, flag, seq);
}

void launchRootFinal(hipStream_t stream, const double* blockSums, int n, double* out, unsigned long long* flag, unsigned long long seq) {
    hipLaunchKernelGGL(k_rootFinal, dim3(1), dim3(ROOT_BLOCK), 0, stream, blockSums, n, out, flag, seq);
}

// ------------------------------------------------------------------------------------------------
// scale-factor bookkeeping
// ------------------------------------------------------------------------------------------------
// cum[p] += sign * sum_k log f_k[p] (raw factors) or sum_k l_k[p] (buffers that hold logarithms already).  A thread owns a
// pattern and walks the buffers; the sum of logarithms of the raw factors is formed as ONE logarithm of their product, the
// product kept as (mantissa in [0.5, 1), binary exponent) so that it cannot leave the range: per factor a mask, an add and a
// multiply instead of a 40-instruction fp64 log (config A, 999 buffers x 1e5 patterns: 727 -> see profiles/r03_experiments.txt).
__global__ __launch_bounds__(256) void k_accumulate(double* __restrict__ cum, const double* const* __restrict__ srcs,
                                                    const int* __restrict__ raw, int count, double sign, int pStart, int pEnd) {
    const int p = pStart + blockIdx.x * 256 + threadIdx.x;
    if (p >= pEnd) return;
    double logs = 0.0, mant = 1.0;
    long expo = 0;
    auto take = [&](double v, int isRaw, int k) {
        if (!isRaw) { logs += v; return; }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        const int field = (int)((bits >> 52) & 0x7ffull);
        if (field == 0 || field == 0x7ff || (long long)bits < 0) { logs += log(v); return; }       // zero, denormal, inf / nan, negative: as written
        mant *= __longlong_as_double((long long)((bits & 0x800fffffffffffffull) | 0x3fe0000000000000ull));
        expo += field - 1022;
        if ((k & 255) == 255) {                        // 256 mantissas >= 0.5 each: the product is still >= 2^-256
            const unsigned long long mb = (unsigned long long)__double_as_longlong(mant);
            expo += (int)((mb >> 52) & 0x7ffull) - 1022;
            mant = __longlong_as_double((long long)((mb & 0x800fffffffffffffull) | 0x3fe0000000000000ull));
        }
    };
    int k = 0;
    for (; k + 4 <= count; k += 4) {                   // four independent loads in flight
        const double v0 = gptr(srcs[k])[p], v1 = gptr(srcs[k + 1])[p], v2 = gptr(srcs[k + 2])[p], v3 = gptr(srcs[k + 3])[p];
        take(v0, raw[k], k); take(v1, raw[k + 1], k + 1); take(v2, raw[k + 2], k + 2); take(v3, raw[k + 3], k + 3);
    }
    for (; k < count; k++) take(gptr(srcs[k])[p], raw[k], k);
    cum[p] += sign * (logs + (log(mant) + (double)expo * 0.69314718055994530942));
}

// The same for a long list of buffers (a whole tree's factors into the cumulative buffer: 999 for config A).  One thread per
// pattern walking 999 buffers is a chain of dependent-latency loads on a handful of workgroups — 250 us whatever the pattern
// count.  Here a workgroup is 64 patterns x 16 waves; wave j takes buffers j, j + 16, ...; the sixteen partial sums of logs
// meet in LDS and are added in wave order (fixed: deterministic).
constexpr int ACC_CHUNKS = 16;
__global__ __launch_bounds__(64 * ACC_CHUNKS) void k_accumulateWide(double* __restrict__ cum, const double* const* __restrict__ srcs,
                                                                    const int* __restrict__ raw, int count, double sign, int pStart, int pEnd) {
    __shared__ double part[ACC_CHUNKS][64];
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int p = pStart + blockIdx.x * 64 + lane;
    const bool valid = p < pEnd;
    const int q = valid ? p : pEnd - 1;
    double logs = 0.0, mant = 1.0;
    long expo = 0;
    int taken = 0;
    for (int k = j; k < count; k += ACC_CHUNKS) {
        const double v = gptr(srcs[k])[q];
        if (!raw[k]) { logs += v; continue; }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        const int field = (int)((bits >> 52) & 0x7ffull);
        if (field == 0 || field == 0x7ff || (long long)bits < 0) { logs += log(v); continue; }
        mant *= __longlong_as_double((long long)((bits & 0x800fffffffffffffull) | 0x3fe0000000000000ull));
        expo += field - 1022;
        if ((++taken & 255) == 0) {
            const unsigned long long mb = (unsigned long long)__double_as_longlong(mant);
            expo += (int)((mb >> 52) & 0x7ffull) - 1022;
            mant = __longlong_as_double((long long)((mb & 0x800fffffffffffffull) | 0x3fe0000000000000ull));
        }
    }
    part[j][lane] = logs + (log(mant) + (double)expo * 0.69314718055994530942);
    __syncthreads();
    if (j == 0 && valid) {
        double t = 0.0;
        for (int c = 0; c < ACC_CHUNKS; c++) t += part[c][lane];
        cum[p] += sign * t;
    }
}

void launchAccumulateScale(hipStream_t stream, double* cum, const double* const* dSrcs, const int* dRaw,
                           int count, double sign, int pStart, int pEnd) {
    if (count <= 0 || pEnd <= pStart) return;
    if (count >= 4 * ACC_CHUNKS)
        hipLaunchKernelGGL(k_accumulateWide, dim3((pEnd - pStart + 63) / 64), dim3(64 * ACC_CHUNKS), 0, stream, cum, dSrcs, dRaw, count, sign, pStart, pEnd);
    else
        hipLaunchKernelGGL(k_accumulate, dim3((pEnd - pStart + 255) / 256), dim3(256), 0, stream, cum, dSrcs, dRaw, count, sign, pStart, pEnd);
}

// accumulateScaleFactors over the factors one write-mode walk has just written (engine_abi.cpp accumulate): the walk's slices each left the
// product of their factors per pattern (tools/gen_walk4_fast.py rescale_block / the loop's exit; k_walk4 likewise), so the sum of the logarithms
// of a thousand factors is a few dozen logarithms and 12 bytes per slice and pattern instead of 8 bytes per NODE and pattern
// (config A with ALWAYS rescaling: 0.8 GB and 250 us per evaluation).  Rows are added in the order given: deterministic.
__global__ __launch_bounds__(256) void k_accumulateSlices(double* __restrict__ cum, const double* __restrict__ mant, const int* __restrict__ expo,
                                                          const int* __restrict__ rows, int n, size_t stride, const unsigned* __restrict__ pairPos,
                                                          double sign, int pStart, int pEnd) {
    const int p = pStart + blockIdx.x * 256 + threadIdx.x;
    if (p >= pEnd) return;
    const size_t q = pairPos ? (size_t)pairPos[p] : walkPairIndex((size_t)p);
    double t = 0.0;
    for (int k = 0; k < n; k++) {
        const size_t at = (size_t)rows[k] * stride + q;
        t += log(mant[at]) + (double)expo[at] * 0.69314718055994530942;
    }
    cum[p] += sign * t;
}
void launchAccumulateSlices(hipStream_t stream, double* cum, const double* mant, const int* expo, const int* dRows, int n, size_t stride,
                            const unsigned* dPairPos, double sign, int pStart, int pEnd) {
    if (n <= 0 || pEnd <= pStart) return;
    hipLaunchKernelGGL(k_accumulateSlices, dim3((pEnd - pStart + 255) / 256), dim3(256), 0, stream, cum, mant, expo, dRows, n, stride, dPairPos, sign, pStart, pEnd);
}

__global__ void k_fill(double* dst, double value, int pStart, int pEnd) {
    const int p = pStart + blockIdx.x * 256 + threadIdx.x;
    if (p < pEnd) dst[p] = value;
}
void launchFill(hipStream_t stream, double* dst, double value, int pStart, int pEnd) {
    if (pEnd <= pStart) return;
    hipLaunchKernelGGL(k_fill, dim3((pEnd - pStart + 255) / 256), dim3(256), 0, stream, dst, value, pStart, pEnd);
}

// Folded reciprocal vectors of read-mode walk programs (engine_walk.cpp foldScales): job j multiplies the reciprocal halves
// srcs[start[j]] .. srcs[start[j + 1] - 1] of per-node scale buffers, entry by entry in that order, into dst[j]; worst[j] receives the
// largest product of the job as a bit pattern (positive doubles order like their bit patterns; anything not finite counts as
// +infinity) — the host refuses a fold whose products leave the safe range.  worst[] must be zero before the launch.
// invert: the sources are the factors themselves (<= 1: the T32 walk divides by them), the range check looks at 1 / product.
__global__ __launch_bounds__(256) void k_foldReciprocals(const double* const* __restrict__ srcs, const int* __restrict__ start, double* const* __restrict__ dst,
                                                         int len, unsigned long long* __restrict__ worst, int invert) {
    const int j = (int)blockIdx.y, i = (int)(blockIdx.x * 256 + threadIdx.x);
    double prod = 1.0;
    if (i < len) {
        // (four members' values in flight at once, multiplied in member order: a load per multiplication, each behind its pointer's, left
        // the 0.74 GB of config A's 58 folds at 1.5 TB/s — 0.5 ms of every rescaling cycle)
        int m = start[j];
        const int e = start[j + 1];
        for (; m + 4 <= e; m += 4) {
            const double* p0 = srcs[m]; const double* p1 = srcs[m + 1]; const double* p2 = srcs[m + 2]; const double* p3 = srcs[m + 3];
            const double a = p0[i], b = p1[i], c = p2[i], d = p3[i];
            prod *= a; prod *= b; prod *= c; prod *= d;
        }
        for (; m < e; m++) prod *= srcs[m][i];
        dst[j][i] = prod;
        if (invert) prod = 1.0 / prod;
    }
    if (!(prod <= 1.7976931348623157e308)) prod = __longlong_as_double(0x7ff0000000000000ll);
    unsigned long long bits = (unsigned long long)__double_as_longlong(prod);
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(bits, o); bits = other > bits ? other : bits; }
    if ((threadIdx.x & 63) == 0) atomicMax(worst + j, bits);
}
// The same, two neighbouring elements per thread (16 bytes per lane and load; every vector starts on a 16-byte boundary and `len` is even:
// launchFoldReciprocals checks): the one-element form left config A's 0.74 GB at 1.7 TB/s.  Element by element the same products in the
// same order.
__global__ __launch_bounds__(256) void k_foldReciprocals2(const double* const* __restrict__ srcs, const int* __restrict__ start, double* const* __restrict__ dst,
                                                          int len, unsigned long long* __restrict__ worst, int invert) {
    const int j = (int)blockIdx.y, i = 2 * (int)(blockIdx.x * 256 + threadIdx.x);
    double2 prod = make_double2(1.0, 1.0);
    if (i < len) {
        int m = start[j];
        const int e = start[j + 1];
        for (; m + 4 <= e; m += 4) {
            const double* p0 = srcs[m]; const double* p1 = srcs[m + 1]; const double* p2 = srcs[m + 2]; const double* p3 = srcs[m + 3];
            const double2 a = *reinterpret_cast<const double2*>(p0 + i), b = *reinterpret_cast<const double2*>(p1 + i),
                          c = *reinterpret_cast<const double2*>(p2 + i), d = *reinterpret_cast<const double2*>(p3 + i);
            prod.x *= a.x; prod.y *= a.y; prod.x *= b.x; prod.y *= b.y; prod.x *= c.x; prod.y *= c.y; prod.x *= d.x; prod.y *= d.y;
        }
        for (; m < e; m++) { const double2 a = *reinterpret_cast<const double2*>(srcs[m] + i); prod.x *= a.x; prod.y *= a.y; }
        *reinterpret_cast<double2*>(dst[j] + i) = prod;
        if (invert) { prod.x = 1.0 / prod.x; prod.y = 1.0 / prod.y; }
    }
    double big = !(prod.x <= 1.7976931348623157e308) || !(prod.y <= 1.7976931348623157e308) ? __longlong_as_double(0x7ff0000000000000ll)
                                                                                               : (prod.x > prod.y ? prod.x : prod.y);
    unsigned long long bits = (unsigned long long)__double_as_longlong(big);
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(bits, o); bits = other > bits ? other : bits; }
    if ((threadIdx.x & 63) == 0) atomicMax(worst + j, bits);
}
void launchFoldReciprocals(hipStream_t stream, const double* const* dSrcs, const int* dStart, double* const* dDst, int nJobs, int len, unsigned long long* dWorst, bool invert, bool pairs) {
    if (nJobs <= 0 || len <= 0) return;
    if (pairs && (len & 1) == 0) {
        hipLaunchKernelGGL(k_foldReciprocals2, dim3((unsigned)((len / 2 + 255) / 256), (unsigned)nJobs), dim3(256), 0, stream, dSrcs, dStart, dDst, len, dWorst, invert ? 1 : 0);
        return;
    }
    hipLaunchKernelGGL(k_foldReciprocals, dim3((unsigned)((len + 255) / 256), (unsigned)nJobs), dim3(256), 0, stream, dSrcs, dStart, dDst, len, dWorst, invert ? 1 : 0);
}

__global__ void k_logScale(const double* in, double* out, int raw, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) out[p] = raw ? log(in[p]) : in[p];
}
void launchLogScale(hipStream_t stream, const double* in, double* out, int raw, int P) {
    hipLaunchKernelGGL(k_logScale, dim3((P + 255) / 256), dim3(256), 0, stream, in, out, raw, P);
}

// API-layout export of a partials buffer (either device layout) with the optional scale factor folded in
__global__ __launch_bounds__(256) void k_exportPartials(const double* __restrict__ src, const double* __restrict__ scale, int scaleIsRaw,
                                                        double* __restrict__ out, int P, int S, int C, int tiled) {
    const size_t n = (size_t)C * P * S;
    const int ntile = (P + 31) >> 5;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % S);
        const size_t cp = e / S;
        const int p = (int)(cp % P), c = (int)(cp / P);
        double v = tiled ? src[(((size_t)c * ntile + (p >> 5)) * S + i) * 32 + (p & 31)] : src[e];
        if (scale) v *= scaleIsRaw ? scale[p] : exp(scale[p]);
        out[e] = v;
    }
}
void launchExportPartials(hipStream_t stream, const double* partials, const double* scale, int scaleIsRaw, double* out,
                          int P, int S, int C, bool tiled) {
    const size_t n = (size_t)C * P * S;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_exportPartials, dim3(blocks), dim3(256), 0, stream, partials, scale, scaleIsRaw, out, P, S, C, tiled ? 1 : 0);
}

__global__ void k_relayoutStates(const uint8_t* oldPlain, uint8_t* newPlain, uint8_t* newPair, const unsigned* pos, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const uint8_t v = oldPlain[p];
    newPlain[p] = v; newPair[pos[p]] = v;
}
void launchRelayoutStates(hipStream_t stream, const uint8_t* oldPlain, uint8_t* newPlain, uint8_t* newPair, const unsigned* dPairPos, int P) {
    hipLaunchKernelGGL(k_relayoutStates, dim3((P + 255) / 256), dim3(256), 0, stream, oldPlain, newPlain, newPair, dPairPos, P);
}
// every tip of an instance in two launches (a partitioned analysis re-lays 1 610 tips at set-up: one fill and one scatter each were
// 3 200 launches of 2 us): first the pair-interleaved arrays are filled with "missing" (their padding must read as missing), then
// the states are scattered into them
__global__ void k_fillStatesBatch(const RelayoutJob* jobs, int pairLen, uint8_t value) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < pairLen) jobs[blockIdx.y].newPair[q] = value;
}
__global__ void k_relayoutStatesBatch(const RelayoutJob* jobs, const unsigned* pos, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const RelayoutJob j = jobs[blockIdx.y];
    const uint8_t v = j.oldPlain[p];
    j.newPlain[p] = v; j.newPair[pos[p]] = v;
}
void launchRelayoutStatesBatch(hipStream_t stream, const RelayoutJob* dJobs, int nJobs, const unsigned* dPairPos, int P, int pairLen, int missing) {
    if (nJobs <= 0) return;
    hipLaunchKernelGGL(k_fillStatesBatch, dim3((pairLen + 255) / 256, nJobs), dim3(256), 0, stream, dJobs, pairLen, (uint8_t)missing);
    hipLaunchKernelGGL(k_relayoutStatesBatch, dim3((P + 255) / 256, nJobs), dim3(256), 0, stream, dJobs, dPairPos, P);
}
__global__ void k_recipFromFactors(const double* f, double* recip, const unsigned* pos, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) recip[pos[p]] = 1.0 / f[p];
}
void launchRecipFromFactors(hipStream_t stream, const double* factors, double* recip, const unsigned* dPairPos, int P) {
    hipLaunchKernelGGL(k_recipFromFactors, dim3((P + 255) / 256), dim3(256), 0, stream, factors, recip, dPairPos, P);
}

__global__ void k_replicate(const double* src, double* dst, size_t n, int C) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const double v = src[e];
    for (int c = 0; c < C; c++) dst[(size_t)c * n + e] = v;
}
void launchReplicateCategories(hipStream_t stream, const double* src, double* dst, int P, int S, int C) {
    const size_t n = (size_t)P * S;
    hipLaunchKernelGGL(k_replicate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n, C);
}

}  // namespace mi355
