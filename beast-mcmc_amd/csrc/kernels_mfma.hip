// kernels_mfma.hip — 20-state (amino-acid) and 61-state (codon) pruning on the fp64 matrix cores.
//
// For these state counts the per-node update IS a batched small GEMM: per category c,
//     sum1[i][p] = sum_j M1[c][i][j] * X1[c][j][p]          (S x S) x (S x patterns)
// so it goes on MFMA — v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction.  Measured on
// MI355X (tools/mfma_f64_probe.hip, profiles/r01_mfma_f64_probe.txt): 73.9 TFLOP/s for the 4x4x4 form vs 45.5 for
// v_mfma_f64_16x16x4_f64 and 61.4 for a plain v_fma_f64 loop, and 4-wide tiles fit S = 20 exactly (5 x 5 tiles,
// no padding) and S = 61 with one partial tile (16 x 16).
//
// Lane map of the instruction (measured by the probe): block b = (l >> 2) & 3,
//     A operand: A_b[i = l & 3][k = l >> 4]      B operand: B_b[k = l >> 4][j = l & 3]      D: D_b[row = l >> 4][col = l & 3]
// The kernel gives the four blocks the SAME matrix tile (A is independent of b) and four different groups of four
// patterns, so one instruction multiplies a 4x4 tile of M by 4 states x 16 patterns.
//
// HBM layout for these state counts ("T32"): partials[c][tile][state j][q], q = pattern within a tile of
// 32 — pattern-major inside a tile, so lane l = (g = l >> 4, m = l & 15) loads the 16 bytes
//     { X[4*jt + g][2m], X[4*jt + g][2m + 1] }
// and those two doubles ARE its B operands for the even-pattern and odd-pattern MFMA of state tile jt: a wave's
// load of one state tile is 1 KiB contiguous, nothing is staged or transposed, and D comes out in the same
// layout, ready for a 16-byte store.  Only the transition matrices go through LDS (as ready-made A fragments).
//
// Restates src/dr/oldevomodel/treelikelihood/GeneralLikelihoodCore.java:52-203 (same arithmetic, different
// summation order: 4-wide partial sums accumulate in the matrix core).
#include "kernels.h"

namespace mi355 {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TILE = 32;
constexpr int MF_BLOCK = 256;

__device__ __forceinline__ double mfma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// One child of one (tile, category): out_e/out_o[it] = sum_j M[4it+g][j] * X[j][2m / 2m+1]
template <int NTMAX>
__device__ __forceinline__ void tiledChild(const double* __restrict__ frag, int nt, int S, bool isStates,
                                           const void* __restrict__ src, const double* __restrict__ Mc, size_t tileBase,
                                           int pe, int P, int g, int m, int lane,
                                           double (&oe)[NTMAX], double (&oo)[NTMAX]) {
    if (isStates) {
        const uint8_t* st = reinterpret_cast<const uint8_t*>(src);
        const int se = pe < P ? st[pe] : S, so = pe + 1 < P ? st[pe + 1] : S;
#pragma unroll
        for (int it = 0; it < NTMAX; it++) {
            const int i = 4 * it + g;
            const bool row = it < nt && i < S;
            oe[it] = (row && se < S) ? Mc[(size_t)i * S + se] : 1.0;
            oo[it] = (row && so < S) ? Mc[(size_t)i * S + so] : 1.0;
        }
        return;
    }
    const double* x = reinterpret_cast<const double*>(src) + tileBase;
    v2d b[NTMAX];
#pragma unroll
    for (int jt = 0; jt < NTMAX; jt++) {
        const int j = 4 * jt + g;
        b[jt] = (jt < nt && j < S) ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(x + (size_t)j * TILE + 2 * m))
                                   : v2d{0.0, 0.0};
    }
#pragma unroll
    for (int it = 0; it < NTMAX; it++) { oe[it] = 0.0; oo[it] = 0.0; }
    const int fl = g * 4 + (lane & 3);
#pragma unroll
    for (int jt = 0; jt < NTMAX; jt++) {
        if (jt < nt) {
#pragma unroll
            for (int it = 0; it < NTMAX; it++) {
                if (it < nt) {
                    const double a = frag[(it * nt + jt) * 16 + fl];
                    oe[it] = mfma4(a, b[jt].x, oe[it]);
                    oo[it] = mfma4(a, b[jt].y, oo[it]);
                }
            }
        }
    }
}

template <int NTMAX>
__global__ __launch_bounds__(MF_BLOCK) void k_pruneTiled(const OpDesc* __restrict__ ops, const double* __restrict__ matrices,
                                                         int P, int S, int C) {
    extern __shared__ double frag[];          // [2][nt*nt][16] A fragments of the two branch matrices, current category
    const OpDesc& op = ops[blockIdx.y];
    const int nt = (S + 3) >> 2;
    const int ntile = (P + TILE - 1) / TILE;
    const int tile0 = op.pStart / TILE, tile1 = (op.pEnd + TILE - 1) / TILE;
    if (tile0 + (int)blockIdx.x * 4 >= tile1) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const bool st1 = op.kind & KIND_STATES1, st2 = op.kind & KIND_STATES2;
    const int fragN = nt * nt * 16;

    for (int c = 0; c < C; c++) {
        const double* M1 = matrices + ((size_t)op.mat1 * C + c) * S * S;
        const double* M2 = matrices + ((size_t)op.mat2 * C + c) * S * S;
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * fragN; e += MF_BLOCK) {
            const int child = e >= fragN, r = e - child * fragN;
            const int f = r >> 4, q = r & 15;
            const int it = f / nt, jt = f - it * nt;
            const int i = 4 * it + (q & 3), j = 4 * jt + (q >> 2);
            frag[e] = (i < S && j < S) ? (child ? M2 : M1)[(size_t)i * S + j] : 0.0;
        }
        __syncthreads();
        for (int tile = tile0 + blockIdx.x * 4 + wave; tile < tile1; tile += gridDim.x * 4) {
            const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
            const int pe = tile * TILE + 2 * m;           // even pattern of this lane; odd = pe + 1
            double re[NTMAX], ro[NTMAX], te[NTMAX], to[NTMAX];
            tiledChild<NTMAX>(frag, nt, S, st1, op.child1, M1, tileBase, pe, P, g, m, lane, re, ro);
            tiledChild<NTMAX>(frag + fragN, nt, S, st2, op.child2, M2, tileBase, pe, P, g, m, lane, te, to);
            double inve = 1.0, invo = 1.0;
            if (!op.scaleWrite && op.scaleRead) {
                if (pe < P) inve = 1.0 / op.scaleRead[pe];
                if (pe + 1 < P) invo = 1.0 / op.scaleRead[pe + 1];
            }
            const bool ine = pe >= op.pStart && pe < op.pEnd, ino = pe + 1 >= op.pStart && pe + 1 < op.pEnd;
            double* d = op.dest + tileBase;
#pragma unroll
            for (int it = 0; it < NTMAX; it++) {
                const int i = 4 * it + g;
                if (it < nt && i < S) {
                    v2d o; o.x = re[it] * te[it] * inve; o.y = ro[it] * to[it] * invo;
                    double* q = d + (size_t)i * TILE + 2 * m;
                    if (ine && ino) __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(q));
                    else { if (ine) q[0] = o.x; if (ino) q[1] = o.y; }
                }
            }
        }
    }
}

// Second pass of a WRITE-mode rescale (only on evaluations that recompute the scalers, 1 in `beagle.rescale`): per
// pattern, the max over categories and states of the freshly written destination becomes the scale factor and the
// destination is divided by it.  One wave per tile.
__global__ __launch_bounds__(MF_BLOCK) void k_rescaleTiled(const OpDesc* __restrict__ ops, int P, int S, int C) {
    const OpDesc& op = ops[blockIdx.y];
    if (!op.scaleWrite) return;
    const int ntile = (P + TILE - 1) / TILE;
    const int tile0 = op.pStart / TILE, tile1 = (op.pEnd + TILE - 1) / TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    for (int tile = tile0 + blockIdx.x * 4 + wave; tile < tile1; tile += gridDim.x * 4) {
        const int pe = tile * TILE + 2 * m;
        double me = 0.0, mo = 0.0;
        for (int c = 0; c < C; c++) {
            const double* d = op.dest + ((size_t)c * ntile + tile) * S * TILE;
            for (int i = g; i < S; i += 4) {
                const v2d v = *reinterpret_cast<const v2d*>(d + (size_t)i * TILE + 2 * m);
                me = fmax(me, v.x); mo = fmax(mo, v.y);
            }
        }
        me = fmax(me, __shfl_xor(me, 16, 64)); me = fmax(me, __shfl_xor(me, 32, 64));
        mo = fmax(mo, __shfl_xor(mo, 16, 64)); mo = fmax(mo, __shfl_xor(mo, 32, 64));
        if (!(me > 0.0)) me = 1.0;
        if (!(mo > 0.0)) mo = 1.0;
        const bool ine = pe >= op.pStart && pe < op.pEnd, ino = pe + 1 >= op.pStart && pe + 1 < op.pEnd;
        if (g == 0) { if (ine) op.scaleWrite[pe] = me; if (ino) op.scaleWrite[pe + 1] = mo; }
        const double ie = 1.0 / me, io = 1.0 / mo;
        for (int c = 0; c < C; c++) {
            double* d = op.dest + ((size_t)c * ntile + tile) * S * TILE;
            for (int i = g; i < S; i += 4) {
                double* q = d + (size_t)i * TILE + 2 * m;
                if (ine) q[0] *= ie;
                if (ino) q[1] *= io;
            }
        }
    }
}

int tiledBlocksPerOp(int P, int nOps) {
    const int tiles = (P + TILE - 1) / TILE;
    int blocks = (tiles + 3) / 4;
    int per = 1024 / (nOps > 0 ? nOps : 1);          // ~4 workgroups per CU in total; each restages the matrices per category
    if (per < 1) per = 1;
    return blocks < per ? blocks : per;
}

void launchPruneLevelTiled(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C,
                           bool anyScaleWrite) {
    if (nOps <= 0) return;
    const int nt = (S + 3) / 4;
    const size_t lds = (size_t)2 * nt * nt * 16 * sizeof(double);
    dim3 grid(tiledBlocksPerOp(P, nOps), nOps), block(MF_BLOCK);
    if (nt <= 5) {
        hipLaunchKernelGGL(k_pruneTiled<5>, grid, block, lds, stream, dOps, matrices, P, S, C);
    } else {
        static bool granted = false;
        if (!granted && lds > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pruneTiled<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            granted = true;
        }
        hipLaunchKernelGGL(k_pruneTiled<16>, grid, block, lds, stream, dOps, matrices, P, S, C);
    }
    if (anyScaleWrite) hipLaunchKernelGGL(k_rescaleTiled, grid, block, 0, stream, dOps, P, S, C);
}

// root integration on the T32 layout: thread per pattern, consecutive lanes = consecutive patterns of a tile
__global__ __launch_bounds__(256) void k_rootSiteTiled(const double* __restrict__ root, const double* __restrict__ catWeights,
                                                       const double* __restrict__ freqs, const double* __restrict__ cum, int cumIsRaw,
                                                       const double* __restrict__ patternWeights, double* __restrict__ siteLogL,
                                                       double* __restrict__ blockSums, int P, int S, int C, int pStart, int pEnd) {
    __shared__ double sh[4];
    const int p = pStart + blockIdx.x * 256 + threadIdx.x;
    const int ntile = (P + TILE - 1) / TILE;
    double contrib = 0.0;
    if (p < pEnd) {
        const int tile = p / TILE, q = p - tile * TILE;
        double sum = 0.0;
        for (int c = 0; c < C; c++) {
            const double* r = root + ((size_t)c * ntile + tile) * S * TILE + q;
            double s = 0.0;
            for (int i = 0; i < S; i++) s += freqs[i] * r[(size_t)i * TILE];
            sum += catWeights[c] * s;
        }
        double site = log(sum);
        if (cum) site += cumIsRaw ? log(cum[p]) : cum[p];
        siteLogL[p] = site;
        contrib = site * patternWeights[p];
    }
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) blockSums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

void launchRootSiteTiled(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                         const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                         double* blockSums, int P, int S, int C, int pStart, int pEnd) {
    const int n = (pEnd - pStart + 255) / 256;
    hipLaunchKernelGGL(k_rootSiteTiled, dim3(n), dim3(256), 0, stream, root, catWeights, freqs, cum, cumIsRaw,
                       patternWeights, siteLogL, blockSums, P, S, C, pStart, pEnd);
}

}  // namespace mi355
