// Pre-order partials, edge derivatives and cross products for gfx950.
//
// The kernels in this file are the DIRECT forms: any state count, both partials layouts, one thread = one pattern, a
// workgroup = one wavefront (64 patterns) whose matrices live in LDS (read at wave-uniform addresses) next to the wave's own
// operand columns.  They are what 4-state (and S < 16, S > 64) instances run — at 4 states they are pure streaming and
// sit near the HBM roofline.  For 16..64 states the engine routes the O(S^2) work through the MFMA pruning kernel instead
// (engine_preorder.cpp preLevelTwoPass / edgeDifferentials: a pre-order op = two pruning passes, an internal edge = one pass +
// k_edgeReduce) and uses the direct kernels only for tip edges, which are O(S) per pattern.
//
// Arithmetic (callers: src/dr/evomodel/treedatalikelihood/preorder/AbstractBeagleGradientDelegate.java:207-221,
// AbstractBeagleBranchGradientDelegate.java:82-92 with the formula spelled out at :103-140):
//   pre(child)[j] = sum_i P_child[i][j] * ( pre(parent)[i] * sum_k P_sib[i][k] post(sib)[k] )
//   num = sum_c w_c sum_j pre[j] sum_k D_c[j][k] post[k],   den = sum_c w_c sum_j pre[j] post[j],   derivative = num / den
#include "kernels.h"

namespace mi355 {

constexpr int PRE_BLOCK = 64;

// element (category c, pattern p, state i) of a partials buffer
template <bool TILED>
__device__ __forceinline__ size_t pidx(int c, int p, int i, int P, int S, int ntile) {
    return TILED ? (((size_t)c * ntile + (p >> 5)) * S + i) * 32 + (p & 31) : ((size_t)c * P + p) * S + i;
}

template <bool TILED>
__global__ __launch_bounds__(PRE_BLOCK) void k_prePartials(const OpDesc* __restrict__ ops, const double* __restrict__ matrices,
                                                           int P, int S, int C, long recipOff) {
    extern __shared__ double sh[];                 // Ms[S*S] | Mc[S*S] | v[S][64] | x[S][64]
    double* Ms = sh; double* Mc = sh + S * S; double* v = Mc + S * S; double* x = v + S * PRE_BLOCK;
    const OpDesc& op = ops[blockIdx.y];
    const int base = op.pStart + blockIdx.x * PRE_BLOCK;
    if (base >= op.pEnd) return;
    const int tid = threadIdx.x, p = base + tid, ntile = (P + 31) >> 5;
    const bool valid = p < op.pEnd, sibStates = op.kind & KIND_STATES2;
    const double MI355_GLOBAL* parent = gptr(reinterpret_cast<const double*>(op.child1));
    double MI355_GLOBAL* dest = gptr(op.dest);
    int s = S;
    if (valid && sibStates) s = gptr(reinterpret_cast<const uint8_t*>(op.child2))[p];
    double inv = 1.0;
    if (valid && !op.scaleWrite && op.scaleRead) inv = 1.0 / gptr(op.scaleRead)[p];
    for (int c = 0; c < C; c++) {
        const double* GS = matrices + ((size_t)op.mat2 * C + c) * S * S;
        const double* GC = matrices + ((size_t)op.mat1 * C + c) * S * S;
        __syncthreads();
        for (int e = tid; e < S * S; e += PRE_BLOCK) { Ms[e] = GS[e]; Mc[e] = GC[e]; }
        if (valid) {
            for (int i = 0; i < S; i++) v[i * PRE_BLOCK + tid] = parent[pidx<TILED>(c, p, i, P, S, ntile)];
            if (!sibStates) {
                const double MI355_GLOBAL* sib = gptr(reinterpret_cast<const double*>(op.child2));
                for (int i = 0; i < S; i++) x[i * PRE_BLOCK + tid] = sib[pidx<TILED>(c, p, i, P, S, ntile)];
            }
        }
        __syncthreads();
        if (!valid) continue;
        for (int i = 0; i < S; i++) {
            double f;
            if (sibStates) f = s < S ? Ms[i * S + s] : 1.0;
            else { f = 0.0; for (int k = 0; k < S; k++) f += Ms[i * S + k] * x[k * PRE_BLOCK + tid]; }
            v[i * PRE_BLOCK + tid] *= f;
        }
        for (int j = 0; j < S; j++) {
            double acc = 0.0;
            for (int i = 0; i < S; i++) acc += v[i * PRE_BLOCK + tid] * Mc[i * S + j];
            dest[pidx<TILED>(c, p, j, P, S, ntile)] = acc * inv;
        }
    }
    if (op.scaleWrite && valid) {                  // rescale now: max over categories and states, divide, keep the raw factor
        double m = 0.0;
        for (int c = 0; c < C; c++) for (int j = 0; j < S; j++) m = fmax(m, dest[pidx<TILED>(c, p, j, P, S, ntile)]);
        if (!(m > 0.0)) m = 1.0;
        gptr(op.scaleWrite)[p] = m;
        const double im = 1.0 / m;
        if (recipOff) gptr(op.scaleWrite)[recipOff + (long)walkPairIndex((size_t)p)] = im;
        for (int c = 0; c < C; c++) for (int j = 0; j < S; j++) dest[pidx<TILED>(c, p, j, P, S, ntile)] *= im;
    }
}

static size_t preLds(int S) { return ((size_t)2 * S * S + (size_t)2 * S * PRE_BLOCK) * sizeof(double); }

// ---- more states than the kernels above can stage (65..255: discrete-trait models with many locations) ----------------------
// The same arithmetic with both matrices read from global memory (L2: every workgroup of a level reads the same few) and the
// operand columns of SIXTEEN patterns at a time in LDS; a workgroup still covers 64 patterns (four rounds), its 64 threads are
// 16 patterns x 4 parts that share a pattern's states i = part, part + 4, ...  A correctness path, as k_pruneGeneral is for the
// likelihood (kernels.hip): what SubstitutionModelCrossProductDelegate / the branch-rate gradients ask for at any stateCount
// (discrete/SubstitutionModelCrossProductDelegate.java:153-178), not a tuned one.
constexpr int BIG_PT = 16, BIG_PARTS = PRE_BLOCK / BIG_PT;

__global__ __launch_bounds__(PRE_BLOCK) void k_prePartialsBig(const OpDesc* __restrict__ ops, const double* __restrict__ matrices, int P, int S, int C) {
    extern __shared__ double sh[];                 // v[S][16] | x[S][16] | red[64]
    double* v = sh; double* x = v + S * BIG_PT; double* red = x + S * BIG_PT;
    const OpDesc& op = ops[blockIdx.y];
    const int tid = threadIdx.x, pt = tid & (BIG_PT - 1), part = tid / BIG_PT;
    const bool sibStates = op.kind & KIND_STATES2;
    const double MI355_GLOBAL* parent = gptr(reinterpret_cast<const double*>(op.child1));
    const double MI355_GLOBAL* sib = gptr(reinterpret_cast<const double*>(op.child2));
    double MI355_GLOBAL* dest = gptr(op.dest);
    for (int round = 0; round < PRE_BLOCK / BIG_PT; round++) {
        const int base = op.pStart + blockIdx.x * PRE_BLOCK + round * BIG_PT;
        if (base >= op.pEnd) break;                                   // (uniform over the workgroup)
        const int p = base + pt;
        const bool valid = p < op.pEnd;
        int s = S;
        if (valid && sibStates) s = gptr(reinterpret_cast<const uint8_t*>(op.child2))[p];
        double inv = 1.0;
        if (valid && !op.scaleWrite && op.scaleRead) inv = 1.0 / gptr(op.scaleRead)[p];
        double mx = 0.0;
        for (int c = 0; c < C; c++) {
            const double* GS = matrices + ((size_t)op.mat2 * C + c) * S * S;
            const double* GC = matrices + ((size_t)op.mat1 * C + c) * S * S;
            __syncthreads();
            for (int i = part; i < S; i += BIG_PARTS) {
                v[i * BIG_PT + pt] = valid ? parent[((size_t)c * P + p) * S + i] : 0.0;
                if (!sibStates) x[i * BIG_PT + pt] = valid ? sib[((size_t)c * P + p) * S + i] : 0.0;
            }
            __syncthreads();
            for (int i = part; i < S; i += BIG_PARTS) {
                double f;
                if (sibStates) f = s < S ? GS[(size_t)i * S + s] : 1.0;
                else { f = 0.0; for (int k = 0; k < S; k++) f += GS[(size_t)i * S + k] * x[k * BIG_PT + pt]; }
                v[i * BIG_PT + pt] *= f;
            }
            __syncthreads();
            for (int j = part; j < S; j += BIG_PARTS) {
                double acc = 0.0;
                for (int i = 0; i < S; i++) acc += v[i * BIG_PT + pt] * GC[(size_t)i * S + j];
                acc *= inv;
                if (valid) dest[((size_t)c * P + p) * S + j] = acc;
                mx = fmax(mx, acc);
            }
        }
        if (op.scaleWrite) {                       // rescale now: max over categories and states, divide, keep the raw factor
            __syncthreads();
            red[tid] = mx;
            __syncthreads();
            double m = 0.0;
            for (int q = 0; q < BIG_PARTS; q++) m = fmax(m, red[q * BIG_PT + pt]);
            if (!(m > 0.0)) m = 1.0;
            if (valid && part == 0) gptr(op.scaleWrite)[p] = m;
            const double im = 1.0 / m;
            if (valid)
                for (int c = 0; c < C; c++) for (int j = part; j < S; j += BIG_PARTS) dest[((size_t)c * P + p) * S + j] *= im;
        }
    }
}

void launchPrePartials(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C, bool tiled,
                       int maxRange, long recipOff) {
    if (nOps <= 0 || maxRange <= 0) return;
    if (preLds(S) > 160 * 1024 && !tiled) {
        const size_t ldsBig = ((size_t)2 * S * BIG_PT + PRE_BLOCK) * sizeof(double);
        if (!grantDynamicLds(reinterpret_cast<const void*>(k_prePartialsBig), 160 * 1024)) return;
        for (int o = 0; o < nOps; o += 65535) {
            const int n = nOps - o < 65535 ? nOps - o : 65535;
            hipLaunchKernelGGL(k_prePartialsBig, dim3((maxRange + PRE_BLOCK - 1) / PRE_BLOCK, n), dim3(PRE_BLOCK), ldsBig, stream, dOps + o, matrices, P, S, C);
        }
        return;
    }
    const size_t lds = preLds(S);
    if (!grantDynamicLds(reinterpret_cast<const void*>(k_prePartials<false>), 160 * 1024) ||
        !grantDynamicLds(reinterpret_cast<const void*>(k_prePartials<true>), 160 * 1024)) return;
    for (int o = 0; o < nOps; o += 65535) {
        const int n = nOps - o < 65535 ? nOps - o : 65535;
        dim3 grid((maxRange + PRE_BLOCK - 1) / PRE_BLOCK, n), block(PRE_BLOCK);
        if (tiled) hipLaunchKernelGGL(k_prePartials<true>, grid, block, lds, stream, dOps + o, matrices, P, S, C, recipOff);
        else hipLaunchKernelGGL(k_prePartials<false>, grid, block, lds, stream, dOps + o, matrices, P, S, C, recipOff);
    }
}

// ---- edge derivatives ---------------------------------------------------------------------------------------------
template <bool TILED>
__global__ __launch_bounds__(PRE_BLOCK) void k_edgeDifferentials(const EdgeDesc* __restrict__ edges, const double* __restrict__ matrices,
                                                                 const double* __restrict__ catWeights,
                                                                 const double* __restrict__ patternWeights,
                                                                 double* __restrict__ perPattern, double* __restrict__ blockSums,
                                                                 int P, int S, int C) {
    extern __shared__ double sh[];                 // D[S*S] | rowsum[S] | u[S][64] | x[S][64]
    double* D = sh; double* rs = sh + S * S; double* u = rs + S; double* x = u + S * PRE_BLOCK;
    const EdgeDesc& ed = edges[blockIdx.y];
    const int tid = threadIdx.x, p = blockIdx.x * PRE_BLOCK + tid, ntile = (P + 31) >> 5;
    const bool valid = p < P, postStates = ed.postIsStates != 0;
    const double MI355_GLOBAL* pre = gptr(ed.pre);
    const size_t row = (size_t)ed.slot;
    int s = S;
    if (valid && postStates) s = gptr(reinterpret_cast<const uint8_t*>(ed.post))[p];
    double num = 0.0, den = 0.0;
    for (int c = 0; c < C; c++) {
        const double* G = matrices + ((size_t)ed.dmat * C + c) * S * S;
        __syncthreads();
        for (int e = tid; e < S * S; e += PRE_BLOCK) D[e] = G[e];
        if (postStates) {                           // a missing tip state is the all-ones vector: D . 1 = the row sums
            __syncthreads();
            for (int j = tid; j < S; j += PRE_BLOCK) { double t = 0.0; for (int k = 0; k < S; k++) t += D[j * S + k]; rs[j] = t; }
        }
        if (valid) {
            for (int i = 0; i < S; i++) u[i * PRE_BLOCK + tid] = pre[pidx<TILED>(c, p, i, P, S, ntile)];
            if (!postStates) {
                const double MI355_GLOBAL* post = gptr(reinterpret_cast<const double*>(ed.post));
                for (int i = 0; i < S; i++) x[i * PRE_BLOCK + tid] = post[pidx<TILED>(c, p, i, P, S, ntile)];
            }
        }
        __syncthreads();
        if (!valid) continue;
        double n = 0.0, d = 0.0;
        for (int j = 0; j < S; j++) {
            double t, xj;
            if (postStates) {
                if (s < S) { t = D[j * S + s]; xj = j == s ? 1.0 : 0.0; }
                else { t = rs[j]; xj = 1.0; }
            } else {
                t = 0.0; for (int k = 0; k < S; k++) t += D[j * S + k] * x[k * PRE_BLOCK + tid];
                xj = x[j * PRE_BLOCK + tid];
            }
            n += u[j * PRE_BLOCK + tid] * t; d += u[j * PRE_BLOCK + tid] * xj;
        }
        num += catWeights[c] * n; den += catWeights[c] * d;
    }
    double w1 = 0.0, w2 = 0.0;
    if (valid) {
        const double deriv = num / den;
        if (perPattern) perPattern[row * P + p] = deriv;
        w1 = patternWeights[p] * deriv; w2 = w1 * deriv;
    }
    // fixed-shape butterfly over the wave: deterministic
    for (int off = 32; off > 0; off >>= 1) { w1 += __shfl_xor(w1, off, 64); w2 += __shfl_xor(w2, off, 64); }
    if (tid == 0) {
        double* b = blockSums + (row * gridDim.x + blockIdx.x) * 2;
        b[0] = w1; b[1] = w2;
    }
}

// (65..255 states: see k_prePartialsBig) the differential matrix read from global memory, sixteen patterns' columns in LDS at a time
__global__ __launch_bounds__(PRE_BLOCK) void k_edgeDifferentialsBig(const EdgeDesc* __restrict__ edges, const double* __restrict__ matrices,
                                                                    const double* __restrict__ catWeights, const double* __restrict__ patternWeights,
                                                                    double* __restrict__ perPattern, double* __restrict__ blockSums, int P, int S, int C) {
    extern __shared__ double sh[];                 // u[S][16] | x[S][16] | rs[S] | red[2][64]
    double* u = sh; double* x = u + S * BIG_PT; double* rs = x + S * BIG_PT; double* red = rs + S;
    const EdgeDesc& ed = edges[blockIdx.y];
    const int tid = threadIdx.x, pt = tid & (BIG_PT - 1), part = tid / BIG_PT;
    const bool postStates = ed.postIsStates != 0;
    const double MI355_GLOBAL* pre = gptr(ed.pre);
    const double MI355_GLOBAL* post = gptr(reinterpret_cast<const double*>(ed.post));
    const size_t row = (size_t)ed.slot;
    double w1 = 0.0, w2 = 0.0;                      // (threads of part 0: their pattern's weighted derivative and its square)
    for (int round = 0; round < PRE_BLOCK / BIG_PT; round++) {
        const int p = blockIdx.x * PRE_BLOCK + round * BIG_PT + pt;
        const bool valid = p < P;
        int s = S;
        if (valid && postStates) s = gptr(reinterpret_cast<const uint8_t*>(ed.post))[p];
        double num = 0.0, den = 0.0;
        for (int c = 0; c < C; c++) {
            const double* G = matrices + ((size_t)ed.dmat * C + c) * S * S;
            __syncthreads();
            if (postStates) for (int j = tid; j < S; j += PRE_BLOCK) { double t = 0.0; for (int k = 0; k < S; k++) t += G[(size_t)j * S + k]; rs[j] = t; }
            for (int i = part; i < S; i += BIG_PARTS) {
                u[i * BIG_PT + pt] = valid ? pre[((size_t)c * P + p) * S + i] : 0.0;
                if (!postStates) x[i * BIG_PT + pt] = valid ? post[((size_t)c * P + p) * S + i] : 0.0;
            }
            __syncthreads();
            double n = 0.0, d = 0.0;
            for (int j = part; j < S; j += BIG_PARTS) {
                double t, xj;
                if (postStates) {
                    if (s < S) { t = G[(size_t)j * S + s]; xj = j == s ? 1.0 : 0.0; }
                    else { t = rs[j]; xj = 1.0; }
                } else {
                    t = 0.0; for (int k = 0; k < S; k++) t += G[(size_t)j * S + k] * x[k * BIG_PT + pt];
                    xj = x[j * BIG_PT + pt];
                }
                n += u[j * BIG_PT + pt] * t; d += u[j * BIG_PT + pt] * xj;
            }
            num += catWeights[c] * n; den += catWeights[c] * d;
        }
        __syncthreads();
        red[tid] = num; red[PRE_BLOCK + tid] = den;
        __syncthreads();
        if (part == 0 && valid) {
            double nn = 0.0, dd = 0.0;
            for (int q = 0; q < BIG_PARTS; q++) { nn += red[q * BIG_PT + pt]; dd += red[PRE_BLOCK + q * BIG_PT + pt]; }
            const double deriv = nn / dd;
            if (perPattern) perPattern[row * P + p] = deriv;
            const double w = patternWeights[p] * deriv;
            w1 += w; w2 += w * deriv;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { w1 += __shfl_xor(w1, off, 64); w2 += __shfl_xor(w2, off, 64); }
    if (tid == 0) {
        double* b = blockSums + (row * gridDim.x + blockIdx.x) * 2;
        b[0] = w1; b[1] = w2;
    }
}

// out[e] = sum over the edge's workgroups, in a fixed order
__global__ __launch_bounds__(64) void k_edgeFinal(const double* __restrict__ blockSums, int nBlocks, double* __restrict__ out) {
    const double* b = blockSums + (size_t)blockIdx.x * nBlocks * 2;
    double w1 = 0.0, w2 = 0.0;
    for (int k = threadIdx.x; k < nBlocks; k += 64) { w1 += b[2 * k]; w2 += b[2 * k + 1]; }
    for (int off = 32; off > 0; off >>= 1) { w1 += __shfl_xor(w1, off, 64); w2 += __shfl_xor(w2, off, 64); }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = w1; out[2 * blockIdx.x + 1] = w2; }
}

int edgeBlocks(int P) { return (P + PRE_BLOCK - 1) / PRE_BLOCK; }

void launchEdgeDifferentials(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* matrices, const double* catWeights,
                             const double* patternWeights, double* perPattern, double* blockSums,
                             int P, int S, int C, bool tiled) {
    if (nEdges <= 0) return;
    const size_t lds = ((size_t)S * S + S + (size_t)2 * S * PRE_BLOCK) * sizeof(double);
    if (lds > 160 * 1024 && !tiled) {
        const size_t ldsBig = ((size_t)2 * S * BIG_PT + S + 2 * PRE_BLOCK) * sizeof(double);
        if (!grantDynamicLds(reinterpret_cast<const void*>(k_edgeDifferentialsBig), 160 * 1024)) return;
        hipLaunchKernelGGL(k_edgeDifferentialsBig, dim3(edgeBlocks(P), nEdges), dim3(PRE_BLOCK), ldsBig, stream, dEdges, matrices, catWeights, patternWeights,
                           perPattern, blockSums, P, S, C);
        return;
    }
    if (!grantDynamicLds(reinterpret_cast<const void*>(k_edgeDifferentials<false>), 160 * 1024) ||
        !grantDynamicLds(reinterpret_cast<const void*>(k_edgeDifferentials<true>), 160 * 1024)) return;
    dim3 grid(edgeBlocks(P), nEdges), block(PRE_BLOCK);
    if (tiled) hipLaunchKernelGGL(k_edgeDifferentials<true>, grid, block, lds, stream, dEdges, matrices, catWeights, patternWeights, perPattern, blockSums, P, S, C);
    else hipLaunchKernelGGL(k_edgeDifferentials<false>, grid, block, lds, stream, dEdges, matrices, catWeights, patternWeights, perPattern, blockSums, P, S, C);
}

// The streaming half of the two-step edge derivative (the O(S^2) half ran on the matrix cores as a pruning pass)
template <bool TILED>
__global__ __launch_bounds__(PRE_BLOCK) void k_edgeReduce(const EdgeDesc* __restrict__ edges, const double* __restrict__ catWeights,
                                                          const double* __restrict__ patternWeights, double* __restrict__ perPattern,
                                                          double* __restrict__ blockSums, int P, int S, int C) {
    const EdgeDesc& ed = edges[blockIdx.y];
    const int tid = threadIdx.x, p = blockIdx.x * PRE_BLOCK + tid, ntile = (P + 31) >> 5;
    const bool valid = p < P;
    const double MI355_GLOBAL* pre = gptr(ed.pre);
    const double MI355_GLOBAL* post = gptr(reinterpret_cast<const double*>(ed.post));
    const double MI355_GLOBAL* tmp = gptr(ed.tmp);
    const size_t row = (size_t)ed.slot;
    double w1 = 0.0, w2 = 0.0;
    if (valid) {
        double num = 0.0, den = 0.0;
        for (int c = 0; c < C; c++) {
            double n = 0.0, d = 0.0;
            for (int j = 0; j < S; j++) {
                const size_t a = pidx<TILED>(c, p, j, P, S, ntile);
                n += tmp[a]; d += pre[a] * post[a];
            }
            num += catWeights[c] * n; den += catWeights[c] * d;
        }
        const double deriv = num / den;
        if (perPattern) perPattern[row * P + p] = deriv;
        w1 = patternWeights[p] * deriv; w2 = w1 * deriv;
    }
    for (int off = 32; off > 0; off >>= 1) { w1 += __shfl_xor(w1, off, 64); w2 += __shfl_xor(w2, off, 64); }
    if (tid == 0) {
        double* b = blockSums + (row * gridDim.x + blockIdx.x) * 2;
        b[0] = w1; b[1] = w2;
    }
}

void launchEdgeReduce(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* catWeights, const double* patternWeights,
                      double* perPattern, double* blockSums, int P, int S, int C, bool tiled) {
    if (nEdges <= 0) return;
    dim3 grid(edgeBlocks(P), nEdges), block(PRE_BLOCK);
    if (tiled) hipLaunchKernelGGL(k_edgeReduce<true>, grid, block, 0, stream, dEdges, catWeights, patternWeights, perPattern, blockSums, P, S, C);
    else hipLaunchKernelGGL(k_edgeReduce<false>, grid, block, 0, stream, dEdges, catWeights, patternWeights, perPattern, blockSums, P, S, C);
}

void launchEdgeFinal(hipStream_t stream, const double* blockSums, int nRows, int P, double* outSums) {
    if (nRows > 0) hipLaunchKernelGGL(k_edgeFinal, dim3(nRows), dim3(64), 0, stream, blockSums, edgeBlocks(P), outSums);
}

// ---- cross products (calculateCrossProductDifferentials) -------------------------------------------------------------
// out[i][j] = sum_e t_e sum_p weight_p (sum_c w_c r_c pre_e[c,p,i] post_e[c,p,j]) / (sum_c w_c pre_e . post_e)
// One wave = one block of 64 patterns, looping over every edge: per (edge, category) the scaled pre column and the post
// column of the 64 patterns go to LDS and thread t accumulates the outputs t, t+64, ... as 64-long dot products — a small
// S x 64 by 64 x S product per step, accumulators in registers.  partial[block][S*S] is then summed over the blocks in
// a fixed order (k_crossFinal).  First correct version (VALU; the shape is MFMA-able).
constexpr int CROSS_MAXM = 64;     // outputs per thread: ceil(64 * 64 / 64)

template <bool TILED>
__global__ __launch_bounds__(PRE_BLOCK) void k_crossProducts(const EdgeDesc* __restrict__ edges, int nEdges,
                                                             const double* __restrict__ edgeLengths,
                                                             const double* __restrict__ catWeights, const double* __restrict__ catRates,
                                                             const double* __restrict__ patternWeights,
                                                             double* __restrict__ partial, int P, int S, int C) {
    extern __shared__ double sh[];                 // u[S][64] | x[S][64]
    double* u = sh; double* x = sh + S * PRE_BLOCK;
    const int tid = threadIdx.x, p = blockIdx.x * PRE_BLOCK + tid, ntile = (P + 31) >> 5, nOut = S * S;
    const bool valid = p < P;
    double acc[CROSS_MAXM];
#pragma unroll
    for (int m = 0; m < CROSS_MAXM; m++) acc[m] = 0.0;
    const double pw = valid ? patternWeights[p] : 0.0;
    for (int e = 0; e < nEdges; e++) {
        const EdgeDesc& ed = edges[e];
        const bool postStates = ed.postIsStates != 0;
        const double MI355_GLOBAL* pre = gptr(ed.pre);
        const double MI355_GLOBAL* post = gptr(reinterpret_cast<const double*>(ed.post));
        int s = S;
        if (valid && postStates) s = gptr(reinterpret_cast<const uint8_t*>(ed.post))[p];
        double den = 0.0;
        if (valid)
            for (int c = 0; c < C; c++) {
                double d = 0.0;
                for (int k = 0; k < S; k++) {
                    const double xk = postStates ? (s < S ? (k == s ? 1.0 : 0.0) : 1.0) : post[pidx<TILED>(c, p, k, P, S, ntile)];
                    d += pre[pidx<TILED>(c, p, k, P, S, ntile)] * xk;
                }
                den += catWeights[c] * d;
            }
        const double f = valid ? edgeLengths[e] * pw / den : 0.0;
        for (int c = 0; c < C; c++) {
            const double g = f * catWeights[c] * catRates[c];
            __syncthreads();
            for (int k = 0; k < S; k++) {
                u[k * PRE_BLOCK + tid] = valid ? g * pre[pidx<TILED>(c, p, k, P, S, ntile)] : 0.0;
                x[k * PRE_BLOCK + tid] = !valid ? 0.0 : postStates ? (s < S ? (k == s ? 1.0 : 0.0) : 1.0) : post[pidx<TILED>(c, p, k, P, S, ntile)];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < CROSS_MAXM; m++) {
                const int o = tid + m * PRE_BLOCK;
                if (o < nOut) {
                    const int i = o / S, j = o - i * S;
                    double t = 0.0;
                    for (int l = 0; l < PRE_BLOCK; l++) t += u[i * PRE_BLOCK + l] * x[j * PRE_BLOCK + l];
                    acc[m] += t;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < CROSS_MAXM; m++) {
        const int o = tid + m * PRE_BLOCK;
        if (o < nOut) partial[(size_t)blockIdx.x * nOut + o] = acc[m];
    }
}

// (65..255 states) S x S outputs do not fit one thread's 64 accumulators x 64 threads: the outputs are cut into tiles of 4096
// (blockIdx.y), and the columns of THIRTY-TWO patterns at a time go to LDS (two rounds per 64-pattern block)
__global__ __launch_bounds__(PRE_BLOCK) void k_crossProductsBig(const EdgeDesc* __restrict__ edges, int nEdges, const double* __restrict__ edgeLengths,
                                                                const double* __restrict__ catWeights, const double* __restrict__ catRates,
                                                                const double* __restrict__ patternWeights, double* __restrict__ partial, int P, int S, int C) {
    constexpr int HALF = 32;
    extern __shared__ double sh[];                 // u[S][32] | x[S][32] | red[64]
    double* u = sh; double* x = u + S * HALF; double* red = x + S * HALF;
    const int tid = threadIdx.x, pt = tid & (HALF - 1), part = tid / HALF, nOut = S * S;
    const int tile0 = (int)blockIdx.y * CROSS_MAXM * PRE_BLOCK;
    double acc[CROSS_MAXM];
#pragma unroll
    for (int m = 0; m < CROSS_MAXM; m++) acc[m] = 0.0;
    for (int e = 0; e < nEdges; e++) {
        const EdgeDesc& ed = edges[e];
        const bool postStates = ed.postIsStates != 0;
        const double MI355_GLOBAL* pre = gptr(ed.pre);
        const double MI355_GLOBAL* post = gptr(reinterpret_cast<const double*>(ed.post));
        for (int round = 0; round < PRE_BLOCK / HALF; round++) {
            const int p = blockIdx.x * PRE_BLOCK + round * HALF + pt;
            const bool valid = p < P;
            int s = S;
            if (valid && postStates) s = gptr(reinterpret_cast<const uint8_t*>(ed.post))[p];
            auto postAt = [&](int c, int k) { return postStates ? (s < S ? (k == s ? 1.0 : 0.0) : 1.0) : post[((size_t)c * P + p) * S + k]; };
            double d = 0.0;
            if (valid)
                for (int c = 0; c < C; c++) {
                    double dc = 0.0;
                    for (int k = part; k < S; k += 2) dc += pre[((size_t)c * P + p) * S + k] * postAt(c, k);
                    d += catWeights[c] * dc;
                }
            __syncthreads();
            red[tid] = d;
            __syncthreads();
            const double den = red[pt] + red[HALF + pt];
            const double f = valid ? edgeLengths[e] * patternWeights[p] / den : 0.0;
            for (int c = 0; c < C; c++) {
                const double g = f * catWeights[c] * catRates[c];
                __syncthreads();
                for (int k = part; k < S; k += 2) {
                    u[k * HALF + pt] = valid ? g * pre[((size_t)c * P + p) * S + k] : 0.0;
                    x[k * HALF + pt] = valid ? postAt(c, k) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < CROSS_MAXM; m++) {
                    const int o = tile0 + tid + m * PRE_BLOCK;
                    if (o < nOut) {
                        const int i = o / S, j = o - i * S;
                        double t = 0.0;
                        for (int l = 0; l < HALF; l++) t += u[i * HALF + l] * x[j * HALF + l];
                        acc[m] += t;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < CROSS_MAXM; m++) {
        const int o = tile0 + tid + m * PRE_BLOCK;
        if (o < nOut) partial[(size_t)blockIdx.x * nOut + o] = acc[m];
    }
}

__global__ void k_crossFinal(const double* __restrict__ partial, int nBlocks, int nOut, double* __restrict__ out) {
    const int o = blockIdx.x * 64 + threadIdx.x;
    if (o >= nOut) return;
    double t = 0.0;
    for (int b = 0; b < nBlocks; b++) t += partial[(size_t)b * nOut + o];
    out[o] = t;
}

void launchCrossProducts(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* dEdgeLengths, const double* catWeights,
                         const double* catRates, const double* patternWeights, double* partial, double* out, int P, int S, int C, bool tiled) {
    if (nEdges <= 0) return;
    const size_t lds = (size_t)2 * S * PRE_BLOCK * sizeof(double);
    if (!grantDynamicLds(reinterpret_cast<const void*>(k_crossProducts<false>), 160 * 1024) ||
        !grantDynamicLds(reinterpret_cast<const void*>(k_crossProducts<true>), 160 * 1024)) return;
    const int nb = edgeBlocks(P);
    if (S > 64) {
        const size_t ldsBig = ((size_t)2 * S * 32 + PRE_BLOCK) * sizeof(double);
        if (!grantDynamicLds(reinterpret_cast<const void*>(k_crossProductsBig), 160 * 1024)) return;
        const int tiles = (S * S + CROSS_MAXM * PRE_BLOCK - 1) / (CROSS_MAXM * PRE_BLOCK);
        hipLaunchKernelGGL(k_crossProductsBig, dim3(nb, tiles), dim3(PRE_BLOCK), ldsBig, stream, dEdges, nEdges, dEdgeLengths, catWeights, catRates, patternWeights, partial, P, S, C);
        hipLaunchKernelGGL(k_crossFinal, dim3((S * S + 63) / 64), dim3(64), 0, stream, partial, nb, S * S, out);
        return;
    }
    if (tiled) hipLaunchKernelGGL(k_crossProducts<true>, dim3(nb), dim3(PRE_BLOCK), lds, stream, dEdges, nEdges, dEdgeLengths, catWeights, catRates, patternWeights, partial, P, S, C);
    else hipLaunchKernelGGL(k_crossProducts<false>, dim3(nb), dim3(PRE_BLOCK), lds, stream, dEdges, nEdges, dEdgeLengths, catWeights, catRates, patternWeights, partial, P, S, C);
    hipLaunchKernelGGL(k_crossFinal, dim3((S * S + 63) / 64), dim3(64), 0, stream, partial, nb, S * S, out);
}

// ---- small helpers ------------------------------------------------------------------------------------------------
// matrices[dst] = transpose(matrices[src]) per category
__global__ void k_transposeMatrices(double* __restrict__ matrices, const int* __restrict__ srcDst, int S, int C) {
    const int src = srcDst[2 * blockIdx.x], dst = srcDst[2 * blockIdx.x + 1];
    const double* a = matrices + (size_t)src * C * S * S;
    double* b = matrices + (size_t)dst * C * S * S;
    for (int e = threadIdx.x; e < C * S * S; e += blockDim.x) {
        const int c = e / (S * S), r = e - c * S * S, i = r / S, j = r - i * S;
        b[(size_t)c * S * S + j * S + i] = a[e];
    }
}
void launchTransposeMatrices(hipStream_t stream, double* matrices, const int* dSrcDst, int count, int S, int C) {
    if (count > 0) hipLaunchKernelGGL(k_transposeMatrices, dim3(count), dim3(256), 0, stream, matrices, dSrcDst, S, C);
}

// dest[c][p][i] = freqs[i]
template <bool TILED>
__global__ void k_fillFrequencies(double* __restrict__ dest, const double* __restrict__ freqs, int P, int S, int C) {
    const int p = blockIdx.x * 256 + threadIdx.x, ntile = (P + 31) >> 5;
    if (p >= P) return;
    for (int c = 0; c < C; c++) for (int i = 0; i < S; i++) dest[pidx<TILED>(c, p, i, P, S, ntile)] = freqs[i];
}
void launchFillFrequencies(hipStream_t stream, double* dest, const double* freqs, int P, int S, int C, bool tiled) {
    dim3 grid((P + 255) / 256), block(256);
    if (tiled) hipLaunchKernelGGL(k_fillFrequencies<true>, grid, block, 0, stream, dest, freqs, P, S, C);
    else hipLaunchKernelGGL(k_fillFrequencies<false>, grid, block, 0, stream, dest, freqs, P, S, C);
}

}  // namespace mi355
