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
#include <cstdlib>

namespace mi355 {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TILE = 32;
constexpr int MF_BLOCK = 256;

__device__ __forceinline__ double mfma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// B operands of one partials child for one (tile, category): b[jt] = { X[4jt+g][2m], X[4jt+g][2m+1] }
template <int NTMAX, bool EXACT>
__device__ __forceinline__ void tiledLoadB(const void* __restrict__ src, size_t tileBase, int S, int g, int m, v2d (&b)[NTMAX]) {
    // wave-uniform base + one 32-bit lane offset; the tile rows are immediates
    const char* x = reinterpret_cast<const char*>(reinterpret_cast<const double*>(src) + tileBase);
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;
#pragma unroll
    for (int jt = 0; jt < NTMAX; jt++) {
        if (EXACT && jt < NTMAX - 1) {
            b[jt] = __builtin_nontemporal_load(gptr(reinterpret_cast<const v2d*>(x + (lane8 + (unsigned)jt * 4u * TILE * 8u))));
        } else {
            // rows >= S do not exist in the buffer: read the last real row instead (branch-free) and zero the operand
            const int j = 4 * jt + g, jc = j < S ? j : S - 1;
            const v2d v = __builtin_nontemporal_load(gptr(reinterpret_cast<const v2d*>(x + (unsigned)(jc * TILE + 2 * m) * 8u)));
            b[jt] = j < S ? v : v2d{0.0, 0.0};
        }
    }
}

// child 1 of tile `tile`: its B operands (partials) or its two state codes (compact tip)
template <int NTMAX, bool EXACT>
__device__ __forceinline__ void tiledFetch1(const OpDesc& op, bool st1, int c, int ntile, int tile, int P, int S, int g, int m,
                                            v2d (&b)[NTMAX], int& se, int& so) {
    if (st1) {
        const uint8_t MI355_GLOBAL* st = gptr(reinterpret_cast<const uint8_t*>(op.child1));
        const int pe = tile * TILE + 2 * m;
        se = pe < P ? st[pe] : S;
        so = pe + 1 < P ? st[pe + 1] : S;
    } else tiledLoadB<NTMAX, EXACT>(op.child1, ((size_t)c * ntile + tile) * S * TILE, S, g, m, b);
}

// A virtual-cherry child (kernels.h CherryDesc): X[j][p] = A[j][sA(p)] * B[j][sB(p)] * (1 / scale[p]) — the arithmetic, and
// its order, of the op that would have stored the cherry (GeneralLikelihoodCore.java:52-105; a missing state contributes 1).
// What comes from memory for a lane's two patterns (the part worth prefetching): four state codes, two factors.
struct CherryRaw { int ae, ao, be, bo; double inve, invo; };
__device__ __forceinline__ CherryRaw cherryFetch(const CherryDesc& cd, int tile, int P, int S, int m) {
    const uint8_t MI355_GLOBAL* ta = gptr(cd.tipA);
    const uint8_t MI355_GLOBAL* tb = gptr(cd.tipB);
    const int pe = tile * TILE + 2 * m;
    CherryRaw r;
    r.ae = r.ao = r.be = r.bo = S; r.inve = r.invo = 1.0;
    if (pe < P) { r.ae = ta[pe]; r.be = tb[pe]; }
    if (pe + 1 < P) { r.ao = ta[pe + 1]; r.bo = tb[pe + 1]; }
    r.ae = r.ae < S ? r.ae : S; r.ao = r.ao < S ? r.ao : S; r.be = r.be < S ? r.be : S; r.bo = r.bo < S ? r.bo : S;
    if (cd.scale) {
        const double MI355_GLOBAL* sr = gptr(cd.scale);
        if (pe < P) r.inve = 1.0 / sr[pe];
        if (pe + 1 < P) r.invo = 1.0 / sr[pe + 1];
    }
    return r;
}
// The cherry's two matrices sit in LDS as vm[column s = 0..S][g][jt] = M[4 jt + g][s] (column S: ones for the rows that
// exist, so a missing state needs no select; rows >= S: zeros): the five values a lane needs of one column are contiguous.
template <int NTMAX>
__device__ __forceinline__ void cherryOperands(const CherryRaw& r, const double* __restrict__ vmA, const double* __restrict__ vmB, int g, v2d (&b)[NTMAX]) {
    const double* ae = vmA + (r.ae * 4 + g) * NTMAX;
    const double* ao = vmA + (r.ao * 4 + g) * NTMAX;
    const double* be = vmB + (r.be * 4 + g) * NTMAX;
    const double* bo = vmB + (r.bo * 4 + g) * NTMAX;
#pragma unroll
    for (int jt = 0; jt < NTMAX; jt++) b[jt] = v2d{ae[jt] * be[jt] * r.inve, ao[jt] * bo[jt] * r.invo};
}
template <int NTMAX>
__device__ __forceinline__ void cherryStage(double* __restrict__ vm, const double* __restrict__ M, int S) {
    const int n = (S + 1) * 4 * NTMAX;
    for (int e = threadIdx.x; e < n; e += MF_BLOCK) {
        const int jt = e % NTMAX, g = (e / NTMAX) & 3, col = e / (4 * NTMAX), i = 4 * jt + g;
        vm[e] = i < S ? (col < S ? M[(size_t)i * S + col] : 1.0) : 0.0;
    }
}

// One child's factor for the parent-state tiles [it0, it0 + IH): oe/oo[k] = sum_j M[4(it0+k)+g][j] * X[j][2m / 2m+1]
template <int NTMAX, int IH>
__device__ __forceinline__ void tiledChild(const double* __restrict__ frag, int nt, int S, bool isStates, int se, int so,
                                           const double* __restrict__ Mc, const v2d (&b)[NTMAX], int it0, int g, int fl,
                                           double (&oe)[IH], double (&oo)[IH]) {
    if (isStates) {
        // column `state` of the matrix, straight from the staged fragments: M[4it+g][s] = frag[(it, s>>2)][(s&3)*4 + g]
        const bool ge = se < S, go = so < S;
        const double* fe = frag + (ge ? (se >> 2) * 16 + (se & 3) * 4 + g : 0);
        const double* fo = frag + (go ? (so >> 2) * 16 + (so & 3) * 4 + g : 0);
#pragma unroll
        for (int k = 0; k < IH; k++) {
            const double ve = fe[(it0 + k) * NTMAX * 16], vo = fo[(it0 + k) * NTMAX * 16];
            oe[k] = ge ? ve : 1.0;
            oo[k] = go ? vo : 1.0;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < IH; k++) { oe[k] = 0.0; oo[k] = 0.0; }
#pragma unroll
    for (int jt = 0; jt < NTMAX; jt++) {
        if (jt < nt) {
#pragma unroll
            for (int k = 0; k < IH; k++) {
                if (it0 + k < nt) {
                    const double a = frag[((it0 + k) * NTMAX + jt) * 16 + fl];
#if defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_NOMFMA)      // TIMING EXPERIMENTS ONLY: one multiply-add instead of the two matrix instructions
                    oe[k] += a * b[jt].x; oo[k] += a * b[jt].y;
#else
                    oe[k] = mfma4(a, b[jt].x, oe[k]);
                    oo[k] = mfma4(a, b[jt].y, oo[k]);
#endif
                }
            }
        }
    }
}

// Child 1's factors for all parent-state tiles are accumulated first; child 2's are then produced IH tiles at a time,
// multiplied in and stored.  The loads are software-pipelined against the MFMAs: child 2's B operands are requested before
// child 1's MFMAs start, the next tile's child-1 operands before child 2's (and the first tile's before the matrix staging).
// NTMAX = 5 (<= 20 states) runs at 4 waves per SIMD, NTMAX = 16 (<= 64 states) at 2.
// EXACT: the state count fills all NTMAX tiles (20 and 61..64 states), so every tile bound folds at compile time.
// PIPE: 0 = loads where they are needed, 1 = child 2 early, 2 = child 2 early + next tile's child 1 (register budget permitting).
template <int NTMAX, bool EXACT, int PIPE, bool CHERRY>
__global__ __launch_bounds__(MF_BLOCK, (NTMAX > 5 ? 2 : 4)) void k_pruneTiled(const OpDesc* __restrict__ ops, const double* __restrict__ matrices,
                                                                              int P, int S, int C, const CherryDesc* __restrict__ cherries,
                                                                              const double* __restrict__ cherryTables) {
    constexpr int IH = NTMAX > 5 ? 4 : NTMAX;
    extern __shared__ double frag[];          // [2][NTMAX*NTMAX][16] A fragments of the two branch matrices, current category
    const OpDesc& op = ops[blockIdx.y / C];      // one (op, rate category) per grid row: single-op levels still fill the chip
    const int c = blockIdx.y % C;
    const int nt = EXACT ? NTMAX : (S + 3) >> 2;
    const int ntile = (P + TILE - 1) / TILE;
    const int tile0 = op.pStart / TILE, tile1 = (op.pEnd + TILE - 1) / TILE;
    if (tile0 + (int)blockIdx.x * 4 >= tile1) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: everything derived from the tile index stays in SGPRs
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const bool st1 = op.kind & KIND_STATES1, st2 = op.kind & KIND_STATES2;
    // virtual-cherry children.  Their two matrices as column tables (cherryOperands): NTMAX = 5 — staged in LDS (four more S x S
    // matrices fit the LDS budget of 4 workgroups per CU); NTMAX = 16 — read from a table in global memory that k_cherryTables
    // built for this list (32 KB per matrix and category: it lives in L2; the LDS is full with the branch matrices)
    const bool vt1 = CHERRY && (op.kind & KIND_CHERRY1), vt2 = CHERRY && (op.kind & KIND_CHERRY2);
    constexpr int fragN = NTMAX * NTMAX * 16;
    constexpr bool LDS_TABLES = NTMAX <= 5;
    const int vmN = (S + 1) * 4 * NTMAX;
    CherryDesc cd1 = {}, cd2 = {};
    if (vt1) cd1 = cherries[(size_t)op.child1];
    if (vt2) cd2 = cherries[(size_t)op.child2];
    const double* vm = frag + 2 * fragN;      // LDS: [4][vmN]: child 1's (A, B), child 2's (A, B)
    const double *vm1A = vm, *vm1B = vm + vmN, *vm2A = vm + 2 * vmN, *vm2B = vm + 3 * vmN;
    if (CHERRY && !LDS_TABLES) {              // global: [cherry][A, B][category][vmN]
        const size_t per = (size_t)C * vmN;
        vm1A = cherryTables + ((size_t)op.child1 * 2) * per + (size_t)(blockIdx.y % C) * vmN; vm1B = vm1A + per;
        vm2A = cherryTables + ((size_t)op.child2 * 2) * per + (size_t)(blockIdx.y % C) * vmN; vm2B = vm2A + per;
    }
    const int tstep = gridDim.x * 4;
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;

    {
        const double* M1 = matrices + ((size_t)op.mat1 * C + c) * S * S;
        const double* M2 = matrices + ((size_t)op.mat2 * C + c) * S * S;
        int tile = tile0 + blockIdx.x * 4 + wave;
        v2d b1[NTMAX], b2[NTMAX];
        int se1 = S, so1 = S;
        if (PIPE == 2 && !vt1 && tile < tile1) tiledFetch1<NTMAX, EXACT>(op, st1, c, ntile, tile, P, S, g, m, b1, se1, so1);   // in flight across the staging
#if defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_NOSTAGE)     // TIMING EXPERIMENTS ONLY (tools/build_mfma_variant.sh; wrong results): what the fragment staging costs
        for (int e = threadIdx.x; e < 2 * fragN; e += MF_BLOCK) frag[e] = 0.5;
        if (false)
#endif
        for (int e = threadIdx.x; e < 2 * fragN; e += MF_BLOCK) {
            const int child = e >= fragN, r = e - child * fragN;
            const int f = r >> 4, q = r & 15;
            const int it = f / NTMAX, jt = f - it * NTMAX;
            const int i = 4 * it + (q & 3), j = 4 * jt + (q >> 2);
            frag[e] = (i < S && j < S) ? (child ? M2 : M1)[(size_t)i * S + j] : 0.0;
        }
        if (LDS_TABLES) {
            double* vmw = frag + 2 * fragN;
            if (vt1) {
                cherryStage<NTMAX>(vmw, matrices + ((size_t)cd1.matA * C + c) * S * S, S);
                cherryStage<NTMAX>(vmw + vmN, matrices + ((size_t)cd1.matB * C + c) * S * S, S);
            }
            if (vt2) {
                cherryStage<NTMAX>(vmw + 2 * vmN, matrices + ((size_t)cd2.matA * C + c) * S * S, S);
                cherryStage<NTMAX>(vmw + 3 * vmN, matrices + ((size_t)cd2.matB * C + c) * S * S, S);
            }
        }
        __syncthreads();
        CherryRaw raw1 = {};
        if (vt1 && tile < tile1) raw1 = cherryFetch(cd1, tile, P, S, m);
        for (; tile < tile1; tile += tstep) {
            const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
            const int pe = tile * TILE + 2 * m;           // even pattern of this lane; odd = pe + 1
            if (vt1) cherryOperands<NTMAX>(raw1, vm1A, vm1B, g, b1);        // (its states and factors were fetched a tile ago)
            else if (PIPE < 2) tiledFetch1<NTMAX, EXACT>(op, st1, c, ntile, tile, P, S, g, m, b1, se1, so1);
            // child 2's operands fly while child 1's MFMAs run
            int se2 = S, so2 = S;
            CherryRaw raw2 = {};
            if (st2) {
                const uint8_t MI355_GLOBAL* st = gptr(reinterpret_cast<const uint8_t*>(op.child2));
                if (pe < P) se2 = st[pe];
                if (pe + 1 < P) so2 = st[pe + 1];
            } else if (vt2) raw2 = cherryFetch(cd2, tile, P, S, m);
            else if (PIPE >= 1) tiledLoadB<NTMAX, EXACT>(op.child2, tileBase, S, g, m, b2);
            double inve = 1.0, invo = 1.0;
            if (!op.scaleWrite && op.scaleRead) {
                const double MI355_GLOBAL* sr = gptr(op.scaleRead);
                if (pe < P) inve = 1.0 / sr[pe];
                if (pe + 1 < P) invo = 1.0 / sr[pe + 1];
            }
            // Child 1's factors are accumulated for RH parent-state tiles at a time (all of them up to 20 states; HALF of the 16
            // tiles above: 32 instead of 64 accumulator registers next to the two children's 2 x 64 operand registers — the
            // one-pass version needed 136 bytes of scratch per lane at 256 registers, profiles/r02_C_*), child 2's are then
            // produced IH tiles at a time, multiplied in and stored.
            constexpr int RH = NTMAX > 5 ? NTMAX / 2 : NTMAX;
            const bool ine = pe >= op.pStart && pe < op.pEnd, ino = pe + 1 >= op.pStart && pe + 1 < op.pEnd;
            double* d = op.dest + tileBase;
#pragma unroll
            for (int h0 = 0; h0 < NTMAX; h0 += RH) {
                double re[RH], ro[RH];
                tiledChild<NTMAX, RH>(frag, nt, S, st1, se1, so1, M1, b1, h0, g, fl, re, ro);
                __builtin_amdgcn_sched_barrier(0);
                if (h0 == 0 && PIPE == 0 && !st2 && !vt2) tiledLoadB<NTMAX, EXACT>(op.child2, tileBase, S, g, m, b2);
                if (h0 + RH >= NTMAX) {
                    // child 1's operands are dead: the next tile's fly while child 2's MFMAs run
                    if (vt1) { if (tile + tstep < tile1) raw1 = cherryFetch(cd1, tile + tstep, P, S, m); }
                    else if (PIPE == 2 && tile + tstep < tile1) tiledFetch1<NTMAX, EXACT>(op, st1, c, ntile, tile + tstep, P, S, g, m, b1, se1, so1);
                }
                if (h0 == 0 && vt2) cherryOperands<NTMAX>(raw2, vm2A, vm2B, g, b2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it0 = h0; it0 < h0 + RH; it0 += IH) {
                    if (it0 < nt) {
                        double te[IH], to[IH];
                        tiledChild<NTMAX, IH>(frag + fragN, nt, S, st2, se2, so2, M2, b2, it0, g, fl, te, to);
#pragma unroll
                        for (int k = 0; k < IH; k++) {
                            const int i = 4 * (it0 + k) + g;
                            if (it0 + k < nt && i < S) {
                                v2d o; o.x = re[it0 - h0 + k] * te[k] * inve; o.y = ro[it0 - h0 + k] * to[k] * invo;
                                double MI355_GLOBAL* q = gptr(reinterpret_cast<double*>(reinterpret_cast<char*>(d) + (lane8 + (unsigned)(it0 + k) * 4u * TILE * 8u)));
#if defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_NOSTORE)     // TIMING EXPERIMENTS ONLY: the result is stored only where it cannot be (keeps the arithmetic alive)
                                if (o.x == -1.0) q[0] = o.x;
#else
                                if (ine && ino) __builtin_nontemporal_store(o, reinterpret_cast<v2d MI355_GLOBAL*>(q));
                                else { if (ine) q[0] = o.x; if (ino) q[1] = o.y; }
#endif
                            }
                        }
                    }
                }
            }
        }
    }
}

// WRITE-mode rescaling in ONE pass (<= 20 states, <= 4 rate categories): a pattern's factor is the maximum over its states AND
// categories, so here a wave takes its tile through all categories, keeps the unscaled results in registers (5 x 16 bytes
// per category), forms the factor (the four state rows of a tile column sit in lanes l, l ^ 16, l ^ 32, l ^ 48), divides
// and stores once.  The two-pass form below re-reads and re-writes every node (config B with ALWAYS rescaling: 15 ms per
// evaluation).  Grid row = operation; the two branch matrices of every category are staged as A fragments.
template <bool EXACT>
__global__ __launch_bounds__(MF_BLOCK, 2) void k_pruneTiledWrite(const OpDesc* __restrict__ ops, const double* __restrict__ matrices, int P, int S, int C) {
    constexpr int NT = 5, MAXC = 4, fragN = NT * NT * 16;
    extern __shared__ double frag[];          // [C][2][fragN]
    const OpDesc& op = ops[blockIdx.y];
    const int ntile = (P + TILE - 1) / TILE;
    const int tile0 = op.pStart / TILE, tile1 = (op.pEnd + TILE - 1) / TILE;
    if (tile0 + (int)blockIdx.x * 4 >= tile1) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const bool st1 = op.kind & KIND_STATES1, st2 = op.kind & KIND_STATES2;
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;
    for (int e = threadIdx.x; e < C * 2 * fragN; e += MF_BLOCK) {
        const int c = e / (2 * fragN), r0 = e - c * 2 * fragN, child = r0 >= fragN, r = r0 - child * fragN;
        const int f = r >> 4, q = r & 15, it = f / NT, jt = f - it * NT, i = 4 * it + (q & 3), j = 4 * jt + (q >> 2);
        const double* M = matrices + ((size_t)(child ? op.mat2 : op.mat1) * C + c) * S * S;
        frag[e] = (i < S && j < S) ? M[(size_t)i * S + j] : 0.0;
    }
    __syncthreads();
    for (int tile = tile0 + blockIdx.x * 4 + wave; tile < tile1; tile += gridDim.x * 4) {
        const int pe = tile * TILE + 2 * m;
        const bool ine = pe >= op.pStart && pe < op.pEnd, ino = pe + 1 >= op.pStart && pe + 1 < op.pEnd;
        int se1 = S, so1 = S, se2 = S, so2 = S;
        if (st1) { const uint8_t MI355_GLOBAL* st = gptr(reinterpret_cast<const uint8_t*>(op.child1)); if (pe < P) se1 = st[pe]; if (pe + 1 < P) so1 = st[pe + 1]; }
        if (st2) { const uint8_t MI355_GLOBAL* st = gptr(reinterpret_cast<const uint8_t*>(op.child2)); if (pe < P) se2 = st[pe]; if (pe + 1 < P) so2 = st[pe + 1]; }
        double inve = 1.0, invo = 1.0;
        if (!op.scaleWrite && op.scaleRead) {
            const double MI355_GLOBAL* sr = gptr(op.scaleRead);
            if (pe < P) inve = 1.0 / sr[pe];
            if (pe + 1 < P) invo = 1.0 / sr[pe + 1];
        }
        v2d res[MAXC][NT];
        double me = 0.0, mo = 0.0;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < C) {
                const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
                v2d b1[NT], b2[NT];
                if (!st1) tiledLoadB<NT, EXACT>(op.child1, tileBase, S, g, m, b1);
                if (!st2) tiledLoadB<NT, EXACT>(op.child2, tileBase, S, g, m, b2);
                double re[NT], ro[NT], te[NT], to[NT];
                tiledChild<NT, NT>(frag + (size_t)c * 2 * fragN, NT, S, st1, se1, so1, nullptr, b1, 0, g, fl, re, ro);
                tiledChild<NT, NT>(frag + (size_t)c * 2 * fragN + fragN, NT, S, st2, se2, so2, nullptr, b2, 0, g, fl, te, to);
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    res[c][j] = v2d{re[j] * te[j] * inve, ro[j] * to[j] * invo};
                    if (EXACT || 4 * j + g < S) { me = fmax(me, res[c][j].x); mo = fmax(mo, res[c][j].y); }
                }
            }
        }
        double ie = 1.0, io = 1.0;
        if (op.scaleWrite) {
            me = fmax(me, __shfl_xor(me, 16, 64)); me = fmax(me, __shfl_xor(me, 32, 64));
            mo = fmax(mo, __shfl_xor(mo, 16, 64)); mo = fmax(mo, __shfl_xor(mo, 32, 64));
            if (!(me > 0.0)) me = 1.0;
            if (!(mo > 0.0)) mo = 1.0;
            if (g == 0) { if (ine) op.scaleWrite[pe] = me; if (ino) op.scaleWrite[pe + 1] = mo; }
            ie = 1.0 / me; io = 1.0 / mo;
        }
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < C) {
                char* dst = reinterpret_cast<char*>(op.dest + ((size_t)c * ntile + tile) * S * TILE);
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    if (EXACT || 4 * j + g < S) {
                        // (the two-pass form multiplies the stored value by the reciprocal: the same two roundings)
                        const v2d o = v2d{res[c][j].x * ie, res[c][j].y * io};
                        double MI355_GLOBAL* q = gptr(reinterpret_cast<double*>(dst + (lane8 + (unsigned)j * 4u * TILE * 8u)));
                        if (ine && ino) __builtin_nontemporal_store(o, reinterpret_cast<v2d MI355_GLOBAL*>(q));
                        else { if (ine) q[0] = o.x; if (ino) q[1] = o.y; }
                    }
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

// workgroups per grid row (one row = one (op, category) for pruning, one op for the rescale pass): enough rows x groups for
// about `target` workgroups in the launch, each restaging its two matrices once and then streaming its share of the tiles
static int tiledBlocksPerRow(int P, int rows, int target) {
    const int tiles = (P + TILE - 1) / TILE;
    const int blocks = (tiles + 3) / 4;
    int per = target / (rows > 0 ? rows : 1);
    if (per < 1) per = 1;
    return blocks < per ? blocks : per;
}

// the column tables of a list's virtual cherries for the 21..64-state kernel: out[cherry][A, B][category][(S + 1) * 4 * 16], laid out as
// cherryStage lays a matrix out in LDS
__global__ void k_cherryTables(const CherryDesc* __restrict__ cherries, int n, const double* __restrict__ matrices, int S, int C, double* __restrict__ out) {
    constexpr int NT = 16;
    const size_t vmN = (size_t)(S + 1) * 4 * NT, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * 2 * C * vmN) return;
    const int e = (int)(t % vmN), c = (int)((t / vmN) % C), which = (int)((t / (vmN * C)) & 1), k = (int)(t / (vmN * C * 2));
    const int jt = e % NT, g = (e / NT) & 3, col = e / (4 * NT), i = 4 * jt + g;
    const double* M = matrices + ((size_t)(which ? cherries[k].matB : cherries[k].matA) * C + c) * S * S;
    out[t] = i < S ? (col < S ? M[(size_t)i * S + col] : 1.0) : 0.0;
}
size_t cherryTableBytes(int nCherries, int S, int C) { return (size_t)nCherries * 2 * C * (S + 1) * 4 * 16 * sizeof(double); }
void launchCherryTables(hipStream_t stream, const CherryDesc* dCherries, int n, const double* matrices, int S, int C, double* out) {
    if (n <= 0) return;
    const size_t total = (size_t)n * 2 * C * (S + 1) * 4 * 16;
    hipLaunchKernelGGL(k_cherryTables, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dCherries, n, matrices, S, C, out);
}

void launchPruneLevelTiled(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C,
                           bool anyScaleWrite, const CherryDesc* dCherries, const double* dCherryTables) {
    if (nOps <= 0) return;
    const int maxOps = 65535 / C;                    // grid.y limit
    if (nOps > maxOps) {
        for (int o = 0; o < nOps; o += maxOps)
            launchPruneLevelTiled(stream, dOps + o, nOps - o < maxOps ? nOps - o : maxOps, matrices, P, S, C, anyScaleWrite, dCherries, dCherryTables);
        return;
    }
    const int nt = (S + 3) / 4;
    static const bool twoPass = getenv("BEAGLE_MI355_RESCALE_TWO_PASS") && atoi(getenv("BEAGLE_MI355_RESCALE_TWO_PASS")) != 0;
    if (anyScaleWrite && !dCherries && nt <= 5 && S >= 16 && C <= 4 && !twoPass) {
        // a level that rescales in write mode, <= 20 states, <= 4 categories: one pass (k_pruneTiledWrite)
        const size_t lds = (size_t)C * 2 * 5 * 5 * 16 * sizeof(double);
        if (grantDynamicLds(reinterpret_cast<const void*>(k_pruneTiledWrite<true>), lds) && grantDynamicLds(reinterpret_cast<const void*>(k_pruneTiledWrite<false>), lds)) {
            dim3 grid(tiledBlocksPerRow(P, nOps, 2048), nOps), block(MF_BLOCK);
            if (S == 20) hipLaunchKernelGGL(k_pruneTiledWrite<true>, grid, block, lds, stream, dOps, matrices, P, S, C);
            else hipLaunchKernelGGL(k_pruneTiledWrite<false>, grid, block, lds, stream, dOps, matrices, P, S, C);
            return;
        }
    }
    // resident workgroups: 4 per CU at <= 20 states (4 waves/SIMD), 2 per CU above (2 waves/SIMD, 64 KiB LDS each); two rounds
    // (three above 20 states: 256 / 512 / 1024 / 1536 / 2048 / 4096 workgroups per launch -> 134 / 203 / 214 / 229 / 228 / 212 evals/s on config C)
    static const int target = [] { const char* e = labEnv("BEAGLE_MI355_TILED_TARGET"); return e ? atoi(e) : 0; }();
    dim3 grid(tiledBlocksPerRow(P, nOps * C, target > 0 ? target : (nt <= 5 ? 2048 : 1536)), nOps * C), block(MF_BLOCK);
    static const int pipe = [] { const char* e = getenv("BEAGLE_MI355_MFMA_PIPE"); return e ? atoi(e) : 2; }();
#define TILED_LAUNCH(NT, EX, PI) do { if (dCherries && (NT <= 5 || dCherryTables)) hipLaunchKernelGGL((k_pruneTiled<NT, EX, PI, true>), grid, block, lds, stream, dOps, matrices, P, S, C, dCherries, dCherryTables); \
                                        else hipLaunchKernelGGL((k_pruneTiled<NT, EX, PI, false>), grid, block, lds, stream, dOps, matrices, P, S, C, dCherries, dCherryTables); } while (0)
    if (nt <= 5) {
        const size_t lds = (size_t)2 * 5 * 5 * 16 * sizeof(double) + (dCherries ? (size_t)4 * (S + 1) * 4 * 5 * sizeof(double) : 0);    // + the cherries' matrices
        if (nt < 5) TILED_LAUNCH(5, false, 0);
        else if (pipe >= 2) TILED_LAUNCH(5, true, 2);
        else if (pipe == 1) TILED_LAUNCH(5, true, 1);
        else TILED_LAUNCH(5, true, 0);
    } else {
        const size_t lds = (size_t)2 * 16 * 16 * 16 * sizeof(double);          // 64 KiB: above the default 48 KiB cap
        const void* fns[] = {(const void*)k_pruneTiled<16, false, 0, false>, (const void*)k_pruneTiled<16, true, 0, false>,
                             (const void*)k_pruneTiled<16, true, 1, false>, (const void*)k_pruneTiled<16, true, 2, false>,
                             (const void*)k_pruneTiled<16, false, 0, true>, (const void*)k_pruneTiled<16, true, 0, true>,
                             (const void*)k_pruneTiled<16, true, 1, true>, (const void*)k_pruneTiled<16, true, 2, true>};
        for (const void* f : fns) if (!grantDynamicLds(f, lds)) return;
        if (nt < 16) TILED_LAUNCH(16, false, 0);
        else if (pipe >= 2) TILED_LAUNCH(16, true, 2);
        else if (pipe == 1) TILED_LAUNCH(16, true, 1);
        else TILED_LAUNCH(16, true, 0);
    }
#undef TILED_LAUNCH
    if (anyScaleWrite) hipLaunchKernelGGL(k_rescaleTiled, dim3(tiledBlocksPerRow(P, nOps, 2048), nOps), block, 0, stream, dOps, P, S, C);
}

// ---- a pre-order operation on the matrix cores (gradients at 16..64 states) ----------------------------------------------------
// pre(child) = P_child^T . ( pre(parent) o (P_sib . post(sib)) )   (AbstractBeagleGradientDelegate.java:207-221), OpDesc fields as
// launchPrePartials: dest = pre(child), child1 = pre(parent), mat1 = the child's branch matrix (used transposed), child2 =
// post(sib) (partials, or compact states: KIND_STATES2), mat2 = the sibling's matrix; scaleRead divides the result.
// Rounds 2-3 expressed this as TWO passes of the pruning kernel (an identity matrix in the first, an all-missing tip in the
// second: engine_preorder.cpp preLevelTwoPass) — five buffer transfers and four matrix products per operation, two of them
// with the identity.  Here a wave takes a tile through both products: t = P_sib . post(sib) leaves the matrix cores in the
// B-operand layout, is multiplied by pre(parent) in place and fed straight into the product with P_child^T — three transfers,
// two products.  Same helper functions (fragment staging, tiledLoadB, tiledChild) as k_pruneTiled.
template <int NTMAX, bool EXACT>
__global__ __launch_bounds__(MF_BLOCK, (NTMAX > 5 ? 2 : 4)) void k_preOpTiled(const OpDesc* __restrict__ ops, const double* __restrict__ matrices,
                                                                              int P, int S, int C) {
    constexpr int IH = NTMAX > 5 ? 4 : NTMAX;
    constexpr int fragN = NTMAX * NTMAX * 16;
    extern __shared__ double frag[];          // [2][fragN]: A fragments of P_sib, then of P_child TRANSPOSED
    const OpDesc& op = ops[blockIdx.y / C];
    const int c = blockIdx.y % C;
    const int nt = EXACT ? NTMAX : (S + 3) >> 2;
    const int ntile = (P + TILE - 1) / TILE;
    const int tile0 = op.pStart / TILE, tile1 = (op.pEnd + TILE - 1) / TILE;
    if (tile0 + (int)blockIdx.x * 4 >= tile1) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const bool st2 = op.kind & KIND_STATES2;
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;
    const double* Mc = matrices + ((size_t)op.mat1 * C + c) * S * S;          // the child's branch matrix
    const double* Ms = matrices + ((size_t)op.mat2 * C + c) * S * S;          // the sibling's
    for (int e = threadIdx.x; e < 2 * fragN; e += MF_BLOCK) {
        const int second = e >= fragN, r = e - second * fragN;
        const int f = r >> 4, q = r & 15;
        const int it = f / NTMAX, jt = f - it * NTMAX;
        const int i = 4 * it + (q & 3), j = 4 * jt + (q >> 2);
        frag[e] = (i < S && j < S) ? (second ? Mc[(size_t)j * S + i] : Ms[(size_t)i * S + j]) : 0.0;
    }
    __syncthreads();
    for (int tile = tile0 + blockIdx.x * 4 + wave; tile < tile1; tile += gridDim.x * 4) {
        const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
        const int pe = tile * TILE + 2 * m;
        v2d bs[NTMAX], u[NTMAX];
        int se = S, so = S;
        if (st2) {
            const uint8_t MI355_GLOBAL* st = gptr(reinterpret_cast<const uint8_t*>(op.child2));
            if (pe < P) se = st[pe];
            if (pe + 1 < P) so = st[pe + 1];
        } else tiledLoadB<NTMAX, EXACT>(op.child2, tileBase, S, g, m, bs);
        tiledLoadB<NTMAX, EXACT>(op.child1, tileBase, S, g, m, u);                // pre(parent): becomes u = pre(parent) o t in place
        double inve = 1.0, invo = 1.0;
        if (op.scaleRead) {
            const double MI355_GLOBAL* sr = gptr(op.scaleRead);
            if (pe < P) inve = 1.0 / sr[pe];
            if (pe + 1 < P) invo = 1.0 / sr[pe + 1];
        }
#pragma unroll
        for (int it0 = 0; it0 < NTMAX; it0 += IH) {
            if (it0 < nt) {
                double te[IH], to[IH];
                tiledChild<NTMAX, IH>(frag, nt, S, st2, se, so, Ms, bs, it0, g, fl, te, to);
#pragma unroll
                for (int k = 0; k < IH; k++) { u[it0 + k].x *= te[k]; u[it0 + k].y *= to[k]; }
            }
        }
        const bool ine = pe >= op.pStart && pe < op.pEnd, ino = pe + 1 >= op.pStart && pe + 1 < op.pEnd;
        double* d = op.dest + tileBase;
#pragma unroll
        for (int it0 = 0; it0 < NTMAX; it0 += IH) {
            if (it0 < nt) {
                double re[IH], ro[IH];
                tiledChild<NTMAX, IH>(frag + fragN, nt, S, false, S, S, Mc, u, it0, g, fl, re, ro);
#pragma unroll
                for (int k = 0; k < IH; k++) {
                    const int i = 4 * (it0 + k) + g;
                    if (it0 + k < nt && i < S) {
                        v2d o; o.x = re[k] * inve; o.y = ro[k] * invo;
                        double MI355_GLOBAL* q = gptr(reinterpret_cast<double*>(reinterpret_cast<char*>(d) + (lane8 + (unsigned)(it0 + k) * 4u * TILE * 8u)));
                        if (ine && ino) __builtin_nontemporal_store(o, reinterpret_cast<v2d MI355_GLOBAL*>(q));
                        else { if (ine) q[0] = o.x; if (ino) q[1] = o.y; }
                    }
                }
            }
        }
    }
}

bool launchPreOpsTiled(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C) {
    if (nOps <= 0) return true;
    const int maxOps = 65535 / C;
    if (nOps > maxOps) {
        for (int o = 0; o < nOps; o += maxOps)
            if (!launchPreOpsTiled(stream, dOps + o, nOps - o < maxOps ? nOps - o : maxOps, matrices, P, S, C)) return false;
        return true;
    }
    const int nt = (S + 3) / 4;
    dim3 grid(tiledBlocksPerRow(P, nOps * C, nt <= 5 ? 2048 : 1536), nOps * C), block(MF_BLOCK);
    if (nt <= 5) {
        const size_t lds = (size_t)2 * 5 * 5 * 16 * sizeof(double);
        if (nt < 5) hipLaunchKernelGGL((k_preOpTiled<5, false>), grid, block, lds, stream, dOps, matrices, P, S, C);
        else hipLaunchKernelGGL((k_preOpTiled<5, true>), grid, block, lds, stream, dOps, matrices, P, S, C);
    } else {
        const size_t lds = (size_t)2 * 16 * 16 * 16 * sizeof(double);
        if (!grantDynamicLds((const void*)k_preOpTiled<16, false>, lds) || !grantDynamicLds((const void*)k_preOpTiled<16, true>, lds)) return false;
        if (nt < 16) hipLaunchKernelGGL((k_preOpTiled<16, false>), grid, block, lds, stream, dOps, matrices, P, S, C);
        else hipLaunchKernelGGL((k_preOpTiled<16, true>), grid, block, lds, stream, dOps, matrices, P, S, C);
    }
    return true;
}

// ---- edge derivatives on the matrix cores (gradients at 16..64 states) ------------------------------------------------------------
// derivative[p] = (sum_c w_c sum_i pre[c,p,i] (D_c . post[c,p])[i]) / (sum_c w_c sum_i pre[c,p,i] post[c,p,i])   for one edge
// (AbstractBeagleGradientDelegate.java:207-221 -> Beagle.calculateEdgeDifferentials; same outputs as k_edgeDifferentials).
// Rounds 2-3 took the O(S^2) product through a pass of the pruning kernel into a scratch buffer and a streaming reduction
// (6 buffer transfers per internal edge).  Here a wave owns a block of 64 patterns (two tiles) for ALL categories: pre and post
// are read once (2 transfers), the product stays in registers, the per-state partial sums are folded across the four state
// rows of the lane map at the end.  A tip edge's compact states become a one-hot B operand (a missing state: all ones).
// The workgroup stages the A fragments of D_c once per category for its 4 x UNITS blocks.
template <int NTMAX, bool EXACT, int UNITS>
__global__ __launch_bounds__(MF_BLOCK, (NTMAX > 5 ? 2 : 4)) void k_edgeTiled(const EdgeDesc* __restrict__ edges, const double* __restrict__ matrices,
                                                                             const double* __restrict__ catWeights,
                                                                             const double* __restrict__ patternWeights,
                                                                             double* __restrict__ perPattern, double* __restrict__ blockSums,
                                                                             int P, int S, int C, int nBlocks) {
    constexpr int IH = NTMAX > 5 ? 4 : NTMAX;
    constexpr int fragN = NTMAX * NTMAX * 16;
    extern __shared__ double frag[];          // [fragN]: A fragments of the differential matrix of the category at hand
    const EdgeDesc& ed = edges[blockIdx.y];
    const int nt = EXACT ? NTMAX : (S + 3) >> 2;
    const int ntile = (P + TILE - 1) / TILE;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const bool postStates = ed.postIsStates != 0;
    const int unit0 = ((int)blockIdx.x * 4 + wave) * UNITS;
    double num[UNITS][2][2], den[UNITS][2][2];
#pragma unroll
    for (int un = 0; un < UNITS; un++)
#pragma unroll
        for (int h = 0; h < 2; h++) { num[un][h][0] = num[un][h][1] = 0.0; den[un][h][0] = den[un][h][1] = 0.0; }
    for (int c = 0; c < C; c++) {
        const double* G = matrices + ((size_t)ed.dmat * C + c) * S * S;
        __syncthreads();
        for (int e = threadIdx.x; e < fragN; e += MF_BLOCK) {
            const int f = e >> 4, q = e & 15;
            const int it = f / NTMAX, jt = f - it * NTMAX;
            const int i = 4 * it + (q & 3), j = 4 * jt + (q >> 2);
            frag[e] = (i < S && j < S) ? G[(size_t)i * S + j] : 0.0;
        }
        __syncthreads();
        const double wc = catWeights[c];
#pragma unroll
        for (int un = 0; un < UNITS; un++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int tile = (unit0 + un) * 2 + h;
                if (tile >= ntile) continue;                                        // (wave-uniform)
                const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
                const int pe = tile * TILE + 2 * m;
                v2d bs[NTMAX], u[NTMAX];
                if (postStates) {
                    const uint8_t MI355_GLOBAL* st = gptr(reinterpret_cast<const uint8_t*>(ed.post));
                    const int se = pe < P ? st[pe] : S, so = pe + 1 < P ? st[pe + 1] : S;
#pragma unroll
                    for (int jt = 0; jt < NTMAX; jt++) {
                        const int j = 4 * jt + g;
                        bs[jt].x = (j < S && (se >= S || j == se)) ? 1.0 : 0.0;
                        bs[jt].y = (j < S && (so >= S || j == so)) ? 1.0 : 0.0;
                    }
                } else tiledLoadB<NTMAX, EXACT>(ed.post, tileBase, S, g, m, bs);
                tiledLoadB<NTMAX, EXACT>(ed.pre, tileBase, S, g, m, u);
                double ne = 0.0, no = 0.0, de = 0.0, dodd = 0.0;
#pragma unroll
                for (int jt = 0; jt < NTMAX; jt++) { de += u[jt].x * bs[jt].x; dodd += u[jt].y * bs[jt].y; }
#pragma unroll
                for (int it0 = 0; it0 < NTMAX; it0 += IH) {
                    if (it0 < nt) {
                        double te[IH], to[IH];
                        tiledChild<NTMAX, IH>(frag, nt, S, false, S, S, G, bs, it0, g, fl, te, to);
#pragma unroll
                        for (int k = 0; k < IH; k++) { ne += u[it0 + k].x * te[k]; no += u[it0 + k].y * to[k]; }
                    }
                }
                num[un][h][0] += wc * ne; num[un][h][1] += wc * no;
                den[un][h][0] += wc * de; den[un][h][1] += wc * dodd;
            }
        }
    }
    const size_t row = (size_t)ed.slot;
#pragma unroll
    for (int un = 0; un < UNITS; un++) {
        const int unit = unit0 + un;
        double w1 = 0.0, w2 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int pe = (unit * 2 + h) * TILE + 2 * m;
#pragma unroll
            for (int par = 0; par < 2; par++) {
                double n = num[un][h][par], d = den[un][h][par];
                n += __shfl_xor(n, 16, 64); d += __shfl_xor(d, 16, 64);            // the four state rows of the lane map
                n += __shfl_xor(n, 32, 64); d += __shfl_xor(d, 32, 64);
                const int p = pe + par;
                if (g == 0 && p < P) {
                    const double deriv = n / d;
                    if (perPattern) perPattern[row * P + p] = deriv;
                    const double t = patternWeights[p] * deriv;
                    w1 += t; w2 += t * deriv;
                }
            }
        }
        // fixed-shape butterfly over the wave (only the lanes of state row 0 hold anything): deterministic
        for (int off = 32; off > 0; off >>= 1) { w1 += __shfl_xor(w1, off, 64); w2 += __shfl_xor(w2, off, 64); }
        if (lane == 0 && unit < nBlocks) {
            double* b = blockSums + (row * nBlocks + unit) * 2;
            b[0] = w1; b[1] = w2;
        }
    }
}

bool launchEdgeTiled(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* matrices, const double* catWeights,
                     const double* patternWeights, double* perPattern, double* blockSums, int P, int S, int C) {
    if (nEdges <= 0) return true;
    const int nt = (S + 3) / 4, nBlocks = edgeBlocks(P);
    dim3 block(MF_BLOCK);
    if (nt <= 5) {
        dim3 grid((nBlocks + 3) / 4, nEdges);
        const size_t lds = (size_t)5 * 5 * 16 * sizeof(double);
        if (nt < 5) hipLaunchKernelGGL((k_edgeTiled<5, false, 1>), grid, block, lds, stream, dEdges, matrices, catWeights, patternWeights, perPattern, blockSums, P, S, C, nBlocks);
        else hipLaunchKernelGGL((k_edgeTiled<5, true, 1>), grid, block, lds, stream, dEdges, matrices, catWeights, patternWeights, perPattern, blockSums, P, S, C, nBlocks);
    } else {
        dim3 grid((nBlocks + 7) / 8, nEdges);
        const size_t lds = (size_t)16 * 16 * 16 * sizeof(double);
        if (nt < 16) hipLaunchKernelGGL((k_edgeTiled<16, false, 2>), grid, block, lds, stream, dEdges, matrices, catWeights, patternWeights, perPattern, blockSums, P, S, C, nBlocks);
        else hipLaunchKernelGGL((k_edgeTiled<16, true, 2>), grid, block, lds, stream, dEdges, matrices, catWeights, patternWeights, perPattern, blockSums, P, S, C, nBlocks);
    }
    return true;
}

// ---- the pattern walk on the T32 layout (17..20 states) ------------------------------------------------------------------
// The level kernel above is bound by the bytes of storing every node and reading it back (config B rebuilt without its stores:
// 4.2 -> 2.0 ms, profiles/r03_experiments.txt 11).  A pattern tile never needs another tile's data either, so the 4-state
// walk's idea carries over: a wave that owns (tile of 32 patterns, rate category) executes a whole post-order program of
// micro-operations (planner.h — the same programs, kinds, hold slots and definitions as kernels_walk4.hip), its running result
// in registers (ACC: the MFMA's D layout IS the B-operand layout, so a result feeds the next product as it is), values that
// wait for a sibling's subtree in LDS hold slots, and only the nodes the planner wants stored go to memory.
// What does NOT carry over is the size of a branch matrix: 3.2 KB per category against 128 bytes.  A workgroup is therefore
// four tiles of ONE category: its four waves share the two matrices of a micro-operation, which arrive as ready-made A
// fragments in program order (k_gatherFragments lays them out once per evaluation) and are staged through LDS one
// micro-operation ahead, double-buffered, one barrier per micro-operation.  Read-mode rescaling only (a factor per pattern
// needs all categories of the pattern: write-mode lists take the level path).
// A fragments of a matrix for the walk: frag[it][fl][6] — the five state-tile columns jt of (row tile it, lane slot fl) are
// CONTIGUOUS (two 16-byte LDS reads and one 8-byte read per row tile instead of five), padded to six doubles so that the 16
// slots of a row tile fall into distinct banks
constexpr int WT_NT = 5, WT_ROW = 5, WT_FRAG = WT_NT * 16 * WT_ROW;   // doubles per matrix (400): k_walkT32
constexpr int WT_HOLD_V2D = WT_NT * 64;                              // v2d per wave and hold slot

__global__ void k_gatherFragments(const WalkOp* __restrict__ prog, int n, int C, int S, double* __restrict__ stream) {
    constexpr int ROW = WT_ROW, FRAG = WT_FRAG;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * C * 2 * FRAG) return;
    const int r = (int)(t % FRAG), child = (int)((t / FRAG) & 1), c = (int)((t / (2 * FRAG)) % C), k = (int)(t / ((size_t)2 * FRAG * C));
    const int jt = r % ROW, q = (r / ROW) & 15, it = r / (16 * ROW), i = 4 * it + (q & 3), j = 4 * jt + (q >> 2);
    const double MI355_GLOBAL* M = gptr(child ? prog[k].m2 : prog[k].m1) + (size_t)c * S * S;
    stream[t] = (jt < WT_NT && i < S && j < S) ? M[(size_t)i * S + j] : 0.0;
}

// One child's factor for all five parent-state tiles from the walk's fragment layout: oe/oo[it] = sum_j M[4 it + g][j] X[j][2m / 2m+1]
// (a compact tip: column `state` of the matrix, ones for a missing state)
__device__ __forceinline__ void walkChild5(const double* __restrict__ frag, int S, bool isStates, int se, int so, const v2d (&b)[WT_NT],
                                           int g, int fl, double (&oe)[WT_NT], double (&oo)[WT_NT]) {
    if (isStates) {
        const bool ge = se < S, go = so < S;
        const double* fe = frag + (ge ? (((se & 3) * 4 + g) * WT_ROW + (se >> 2)) : 0);
        const double* fo = frag + (go ? (((so & 3) * 4 + g) * WT_ROW + (so >> 2)) : 0);
#pragma unroll
        for (int it = 0; it < WT_NT; it++) {
            const double ve = fe[it * 16 * WT_ROW], vo = fo[it * 16 * WT_ROW];
            oe[it] = ge ? ve : 1.0;
            oo[it] = go ? vo : 1.0;
        }
        return;
    }
    // row tile by row tile (round 6): a row's five fragments (40 bytes in LDS: rows of 5 doubles fall into distinct banks as they
    // are — lane slot fl starts at bank 10 fl mod 64 —, so no padding: 12.5 KiB of fragment buffers, and with the hold slots' 40 KiB a
    // workgroup fits a CU three times), then its ten MFMAs — two accumulators (even / odd patterns), each a chain over the five column
    // tiles; the matrix pipe takes a wave's instructions one after the other anyway.  Rounds 3-5 read all 25 fragments first (50
    // registers live at once: 208 VGPRs, two waves per SIMD).
#pragma unroll
    for (int it = 0; it < WT_NT; it++) {
        const double* row = frag + (it * 16 + fl) * WT_ROW;
        double a[WT_NT];
#pragma unroll
        for (int jt = 0; jt < WT_NT; jt++) a[jt] = row[jt];
        double e = 0.0, o = 0.0;
#pragma unroll
        for (int jt = 0; jt < WT_NT; jt++) { e = mfma4(a[jt], b[jt].x, e); o = mfma4(a[jt], b[jt].y, o); }
        oe[it] = e; oo[it] = o;
    }
}

template <bool EXACT>
__global__ __launch_bounds__(MF_BLOCK, 3) void k_walkT32(const WalkOp* __restrict__ prog, const WalkSeg* __restrict__ segs,
                                                         const double* __restrict__ fragStream, int P, int S, int C) {
    extern __shared__ double wtLds[];              // frag[2][2 * WT_FRAG] doubles, then hold[3][4 waves][WT_NT][64] v2d
    const WalkSeg& sg = segs[blockIdx.y / C];
    const int c = blockIdx.y % C;
    const int ntile = (P + TILE - 1) / TILE;
    const int tile1 = (sg.pEnd + TILE - 1) / TILE;
    const int tileB = sg.pStart / TILE + (int)blockIdx.x * 4;
    if (tileB >= tile1) return;                    // the whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const bool active = tileB + wave < tile1;      // a wave past the end walks the last tile along (barriers, staging) and stores nothing
    const int tile = active ? tileB + wave : tile1 - 1;
    const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
    const int pe = tile * TILE + 2 * m;
    const bool ine = active && pe >= sg.pStart && pe < sg.pEnd, ino = active && pe + 1 >= sg.pStart && pe + 1 < sg.pEnd;
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;
    v2d* hold = reinterpret_cast<v2d*>(wtLds + 4 * WT_FRAG) + (size_t)wave * WT_HOLD_V2D + lane;     // + slot * 4 * WT_HOLD_V2D, tile row k at + 64 k
    const int nOps = sg.progCount;
    const WalkOp* dp = prog + sg.progStart;
    const v2d MI355_GLOBAL* fs = gptr(reinterpret_cast<const v2d*>(fragStream)) + ((size_t)sg.progStart * C + c) * WT_FRAG;   // (2 * WT_FRAG doubles = WT_FRAG v2d per entry and category)
    const size_t fsStep = (size_t)C * WT_FRAG;
    v2d* fragV = reinterpret_cast<v2d*>(wtLds);
    {   // the first micro-operation's fragments
        const int t = threadIdx.x;                 // 2 matrices x 400 doubles = WT_FRAG v2d
        fragV[t] = fs[t];
        if (t < WT_FRAG - 256) fragV[t + 256] = fs[t + 256];
    }
    __syncthreads();
    v2d ACC[WT_NT];
#pragma unroll
    for (int k = 0; k < WT_NT; k++) ACC[k] = v2d{1.0, 1.0};
    // What a micro-operation needs from memory as a matter of course — the host points unused operands at dummies, engine_walk.cpp, so
    // the same instructions go out whatever the kinds are and the waits are constants:
    //  * its 6.4 KB of fragments (400 v2d): ONE micro-operation ahead, by LDS-DMA (round 6; global_load_lds_dwordx4 — 64 lanes x 16
    //    bytes land in 1 KB of LDS at M0: no registers, no ds_write; rounds 3-5 took them through registers two ahead), two per wave:
    //    256 threads x 2 x 16 bytes over the 400 x 16 — the second piece is short (144 lanes of the workgroup): wave 2 issues it with
    //    a quarter of its EXEC, wave 3 not at all;
    //  * the two children's state codes (one ushort each: the lane's two patterns are neighbours) and the raw scale factors (one
    //    16-byte load): ONE ahead as well, into ONE set of six registers (rounds 3-5: two sets, two ahead; at three waves per SIMD
    //    the second set made the register allocator split the live range of a register a load was still writing — tools/
    //    check_walk_isa.py caught it — and a stage is ~1 us: one ahead covers an L2 hit several times over).
    // Issue order in stage k: [wait: operands of k = everything outstanding] DMA(k + 1) x 2, operands(k + 1) x 3, ... [wait: DMA(k + 1) =
    // all but the three youngest] barrier.  Loads return in issue order; stores and the compiler's own loads in the queue only make a
    // wait stricter (a stage that stored its result finds its last stores waited for at the next stage's start).  The compiler cannot express that (it drains the queue at the first use), hence inline assembly, as in
    // kernels_walk4.hip.  A child's PARTIALS in memory are rare (20 of 499 micro-operations of config B) and are read where needed.
    struct Flight { unsigned t1, t2; v2d sc; };                      // one micro-operation's operand loads (registers written asynchronously)
    const unsigned oFrag = (unsigned)threadIdx.x * 16u, oFrag2 = oFrag + 4096u, oPe = (unsigned)pe, oPe8 = (unsigned)pe * 8u;
    const int n1 = WT_FRAG - 256 - 64 * wave;                                             // lanes of this wave that carry a second piece (400 - 256 = 144: two waves and a quarter)
    const unsigned long long mask2 = n1 >= 64 ? ~0ull : n1 > 0 ? (1ull << n1) - 1ull : 0ull;
    const unsigned ldsBase = (unsigned)__builtin_amdgcn_groupstaticsize();                // (the dynamic LDS starts behind the static: there is none)
    const unsigned ldsW = ldsBase + (unsigned)wave * 1024u, fragBytes = (unsigned)(2 * WT_FRAG) * 8u;
    auto issue = [&](Flight& f, const WalkOp& d) {
        asm volatile(
            "global_load_ushort %[t1], %[oP], %[s1]\n\t"
            "global_load_ushort %[t2], %[oP], %[s2]\n\t"
            "global_load_dwordx4 %[sc], %[oS], %[ss]"
            : [t1] "=&v"(f.t1), [t2] "=&v"(f.t2), [sc] "=&v"(f.sc)      /* (pure outputs: a tie to the old value would make the compiler shuffle
                  registers around the wait — and read a destination before its load has landed) */
            : [oP] "v"(oPe), [oS] "v"(oPe8), [s1] "s"(d.src1), [s2] "s"(d.src2), [ss] "s"(d.scale)
            : "memory");
    };
    auto dma = [&](const v2d MI355_GLOBAL* fptr, unsigned parity) {    // the entry at fptr into fragment buffer `parity`
        unsigned keep;
        unsigned long long ex;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(ldsW + parity * fragBytes), l1 = l0 + 4096u;
        // (the waits count the OPERAND loads behind the DMAs, not the DMAs: a wave without a second piece simply issues one)
        if (n1 > 0)
            asm volatile(
                "s_mov_b32 %[keep], m0\n\t"
                "s_mov_b32 m0, %[l0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[fp]\n\t"
                "s_mov_b32 m0, %[l1]\n\ts_mov_b64 %[ex], exec\n\ts_and_b64 exec, %[ex], %[m2]\n\tglobal_load_lds_dwordx4 %[o1], %[fp]\n\ts_mov_b64 exec, %[ex]\n\t"
                "s_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep), [ex] "=&s"(ex)
                : [l0] "s"(l0), [l1] "s"(l1), [o0] "v"(oFrag), [o1] "v"(oFrag2), [fp] "s"(fptr), [m2] "s"(mask2)
                : "memory", "scc");
        else
            asm volatile(
                "s_mov_b32 %[keep], m0\n\t"
                "s_mov_b32 m0, %[l0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[fp]\n\t"
                "s_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep)
                : [l0] "s"(l0), [o0] "v"(oFrag), [fp] "s"(fptr)
                : "memory");
    };
    // "At most the three loads issued last are outstanding".  The registers of a Flight are written asynchronously, so the
    // compiler must never be given a reason to copy one between issue and wait: the wait below only READS them (no tied
    // operands — a tie made the compiler move a destination to another register BEFORE the wait), and what has to outlive the
    // set's next issue is copied out inside the same statement, after the wait.  tools/check_walk_isa.py verifies the result.
    auto landedOperands = [&](const Flight& f, unsigned& t1, unsigned& t2, double& fe, double& fo) {
        asm volatile("s_waitcnt vmcnt(0) ; retires %[i1] %[i2] %[ie] %[io]\n\t"
                     "v_mov_b32 %[t1], %[i1]\n\tv_mov_b32 %[t2], %[i2]\n\tv_mov_b64 %[fe], %[ie]\n\tv_mov_b64 %[fo], %[io]"
                     : [t1] "=&v"(t1), [t2] "=&v"(t2), [fe] "=&v"(fe), [fo] "=&v"(fo)
                     : [i1] "v"(f.t1), [i2] "v"(f.t2), [ie] "v"(f.sc.x), [io] "v"(f.sc.y) : "memory");
    };
    Flight F;
    F.sc = v2d{1.0, 1.0}; F.t1 = F.t2 = 0u;
    issue(F, dp[0]);

    // F: the one set of operand registers — micro-operation k's when the stage begins, handed to k + 1's loads as soon as they are read out
#define WT_STAGE()                                                                                                  \
    {                                                                                                                     \
        const WalkOp& d = dp[k];                                                                                          \
        const unsigned flg = d.flags;                                                                                     \
        const int k1 = (flg >> 5) & 7, k2 = (flg >> 8) & 7, hslot = (flg >> 11) & 3;                                      \
        unsigned t1, t2;                                                                                                  \
        double fe, fo;                                                                                                    \
        landedOperands(F, t1, t2, fe, fo);                            /* (read out before the set is handed to the next loads) */ \
        dma(fs + (size_t)(k + 1) * fsStep, (unsigned)((k + 1) & 1));                                                      \
        issue(F, dp[k + 1]);                                                                                              \
        const int se1 = (int)(t1 & 0xffu), so1 = (int)(t1 >> 8) & 0xff, se2 = (int)(t2 & 0xffu), so2 = (int)(t2 >> 8) & 0xff; \
        const double* frag = wtLds + (size_t)(k & 1) * 2 * WT_FRAG;                                                       \
        /* the second child first: the running result (ACC) is consumed where it stands */                                \
        double te[WT_NT], to[WT_NT];                                                                                      \
        if (k2 == WK_ACC) walkChild5(frag + WT_FRAG, S, false, S, S, ACC, g, fl, te, to);                                 \
        else {                                                                                                            \
            v2d b2[WT_NT];                                                                                                \
            if (k2 == WK_MEM) tiledLoadB<WT_NT, EXACT>(d.src2, tileBase, S, g, m, b2);                                    \
            walkChild5(frag + WT_FRAG, S, k2 == WK_TIPS, se2, so2, b2, g, fl, te, to);                                    \
        }                                                                                                                 \
        const bool rd = ((flg >> 13) & 3) == WS_READ;                                                                     \
        const double inve = rd ? 1.0 / fe : 1.0, invo = rd ? 1.0 / fo : 1.0;                                              \
        {                                                                                                                 \
            v2d b1[WT_NT];                                                                                                \
            if (k1 == WK_MEM) tiledLoadB<WT_NT, EXACT>(d.src1, tileBase, S, g, m, b1);                                    \
            else if (k1 >= WK_H0) {                                                                                       \
                const v2d* h = hold + (size_t)(k1 - WK_H0) * 4 * WT_HOLD_V2D;                                             \
                _Pragma("unroll") for (int j = 0; j < WT_NT; j++) b1[j] = h[64 * j];                                      \
            }                                                                                                             \
            double re[WT_NT], ro[WT_NT];                                                                                  \
            walkChild5(frag, S, k1 == WK_TIPS, se1, so1, b1, g, fl, re, ro);                                              \
            _Pragma("unroll") for (int j = 0; j < WT_NT; j++) ACC[j] = v2d{re[j] * te[j] * inve, ro[j] * to[j] * invo};   \
        }                                                                                                                 \
        if (flg & WF_STORE) {                                                                                             \
            char* dst = reinterpret_cast<char*>(d.store + tileBase);                                                      \
            _Pragma("unroll") for (int j = 0; j < WT_NT; j++) {                                                           \
                if (EXACT || 4 * j + g < S) {                                                                             \
                    double MI355_GLOBAL* q = gptr(reinterpret_cast<double*>(dst + (lane8 + (unsigned)j * 4u * TILE * 8u))); \
                    if (ine && ino) __builtin_nontemporal_store(ACC[j], reinterpret_cast<v2d MI355_GLOBAL*>(q));           \
                    else { if (ine) q[0] = ACC[j].x; if (ino) q[1] = ACC[j].y; }                                          \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
        if (hslot) {                                                                                                      \
            v2d* h = hold + (size_t)(hslot - 1) * 4 * WT_HOLD_V2D;                                                        \
            _Pragma("unroll") for (int j = 0; j < WT_NT; j++) h[64 * j] = ACC[j];                                         \
        }                                                                                                                 \
        /* the next micro-operation's fragments have landed in the other buffer (this wave's share: the barrier makes it everybody's) */ \
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
    }
    for (int k = 0; k < nOps; k++) {               // (one more readable descriptor and stream entry follow the segment: its trailing no-ops)
        WT_STAGE()
    }
#undef WT_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the same walk with WRITE-mode rescaling (PartialsRescalingScheme ALWAYS, and DYNAMIC's every-100th evaluation) --------------
// A pattern's factor is the maximum over its states AND its rate categories (GeneralLikelihoodCore.java:281-318 scalePartials), and
// k_walkT32 keeps a workgroup to ONE category so that its waves can share the matrices: it has no write mode, and until round 6 a
// list that rescaled ran level by level — every node stored and read back, 32 GB per evaluation of config B, 6.5 ms against the
// read-mode walk's 2.5.  k_walkT32W1 below puts all C (<= 4) categories of a tile into one workgroup: its waves leave their
// per-pattern maxima in LDS on the way into the stage's barrier (which the fragment staging needs anyway: no barrier more), every
// wave reads the C of them behind it and multiplies its result by the reciprocal of their maximum — the two roundings of
// k_pruneTiledWrite: the stored value times the reciprocal of the factor — and category 0's wave stores the factor.  The exchange
// runs for every micro-operation, straight-line (a program of this kernel rescales nearly everywhere; a multiplication by 1.0 changes
// no bit); read-mode and unscaled micro-operations of the same list behave as in k_walkT32.
// (The first form, earlier in round 6: TWO tiles x C categories per workgroup, the hold slots in LDS — 145 KB, one workgroup of eight
// waves per CU: config B under ALWAYS 154 -> 282 evaluations/s; this one: 328.  profiles/r06_experiments.txt 11, 16.)
// max over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the four state rows of a tile column), in every one of them: gfx950's lane-row
// swaps — two instructions per 32-bit half and step, no LDS round trip (ds_bpermute, what __shfl_xor compiles to, is one per half and step)
__device__ __forceinline__ double maxOverRows(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto a1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double w = fmax(__hiloint2double((int)a1[0], (int)a0[0]), __hiloint2double((int)a1[1], (int)a0[1]));
    lo = (unsigned)__double2loint(w); hi = (unsigned)__double2hiint(w);
    auto b0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return fmax(__hiloint2double((int)b1[0], (int)b0[0]), __hiloint2double((int)b1[1], (int)b0[1]));
}

// ONE tile per workgroup, the hold slots in registers.
// What held the first form at one workgroup of eight waves per CU was 80 KB of LDS hold slots.  A hold slot is lane-private — the wave
// that parks a result is the wave that takes it back, lane for lane — so it can be ten registers instead of 5 KB of LDS: with the
// changes that took k_walkT32 to 120 registers (fragments row tile by row tile, one set of operand registers, one stage per loop
// iteration) two slots fit a 168-register budget.  The workgroup is then the C categories of ONE tile (wave = category), its LDS
// the two fragment buffers (rows of 5: 12.5 KB per category and buffer, every wave DMAs its own category's) and the maxima:
// 52 KB at C = 4 — THREE workgroups per CU, three waves per SIMD, and a barrier couples four waves instead of eight.
template <bool EXACT>
__global__ __launch_bounds__(256, 3) void k_walkT32W1(const WalkOp* __restrict__ prog, const WalkSeg* __restrict__ segs,
                                                      const double* __restrict__ fragStream, int P, int S, int C) {
    extern __shared__ double wtLds[];              // frag[2][C][2 * WT_FRAG] doubles | mx[2][C][16] v2d
    const WalkSeg& sg = segs[blockIdx.y];
    const int ntile = (P + TILE - 1) / TILE;
    const int tile1 = (sg.pEnd + TILE - 1) / TILE;
    const int tile = sg.pStart / TILE + (int)blockIdx.x;
    if (tile >= tile1) return;                     // the whole workgroup
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // wave = rate category
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
    const int pe = tile * TILE + 2 * m;
    const bool ine = pe >= sg.pStart && pe < sg.pEnd, ino = pe + 1 >= sg.pStart && pe + 1 < sg.pEnd;
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;
    const int fragD = C * 2 * WT_FRAG;             // doubles per micro-operation (all categories, both children)
    v2d* mx = reinterpret_cast<v2d*>(wtLds + 2 * fragD);                     // [parity][category][16] v2d: (even, odd) maxima of pattern pair m
    const int nOps = sg.progCount;
    const WalkOp* dp = prog + sg.progStart;
    // this wave's category of every entry: WT_FRAG v2d at [entry][category]
    const v2d MI355_GLOBAL* fs = gptr(reinterpret_cast<const v2d*>(fragStream)) + ((size_t)sg.progStart * C + c) * WT_FRAG;
    const size_t fsStep = (size_t)C * WT_FRAG;
    {   // the first micro-operation's fragments (this wave's category)
        v2d* fw = reinterpret_cast<v2d*>(wtLds) + (size_t)c * WT_FRAG;
        for (int i = lane; i < WT_FRAG; i += 64) fw[i] = fs[i];
    }
    __syncthreads();
    v2d ACC[WT_NT], H0[WT_NT], H1[WT_NT];
#pragma unroll
    for (int k = 0; k < WT_NT; k++) { ACC[k] = v2d{1.0, 1.0}; H0[k] = ACC[k]; H1[k] = ACC[k]; }
    // Per stage and wave: seven LDS-DMAs (its category's 400 v2d of the NEXT micro-operation: six of 64 lanes and one of 16) into the
    // other buffer, then the three operand loads of the next micro-operation into the one set of operand registers (k_walkT32).
    // Waits: everything outstanding at the stage's start (the operands), all but the three youngest before the barrier (the DMAs).
    struct Flight { unsigned t1, t2; v2d sc; };
    const unsigned oFa = (unsigned)lane * 16u, oFb = oFa + 4096u, oPe = (unsigned)pe, oPe8 = (unsigned)pe * 8u;
    const unsigned ldsBase = (unsigned)__builtin_amdgcn_groupstaticsize();                // (no static LDS: the dynamic block starts here)
    const unsigned fragBytes = (unsigned)fragD * 8u, ldsC = ldsBase + (unsigned)c * (unsigned)(2 * WT_FRAG) * 8u;
    auto issue = [&](Flight& f, const WalkOp& d) {
        asm volatile(
            "global_load_ushort %[t1], %[oP], %[s1]\n\t"
            "global_load_ushort %[t2], %[oP], %[s2]\n\t"
            "global_load_dwordx4 %[sc], %[oS], %[ss]"
            : [t1] "=&v"(f.t1), [t2] "=&v"(f.t2), [sc] "=&v"(f.sc)
            : [oP] "v"(oPe), [oS] "v"(oPe8), [s1] "s"(d.src1), [s2] "s"(d.src2), [ss] "s"(d.scale)
            : "memory");
    };
    auto dma = [&](const v2d MI355_GLOBAL* fptr, unsigned parity) {
        unsigned keep;
        unsigned long long ex;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(ldsC + parity * fragBytes);
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            /* (the instruction's offset field moves BOTH ends: memory address and LDS address — measured: with M0 advanced as well the   \
               pieces landed 1 KB too far per step) */                                                                              \
            "s_mov_b32 m0, %[l0]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[oa], %[fp]\n\t"
            "global_load_lds_dwordx4 %[oa], %[fp] offset:1024\n\t"
            "global_load_lds_dwordx4 %[oa], %[fp] offset:2048\n\t"
            "global_load_lds_dwordx4 %[oa], %[fp] offset:3072\n\t"
            "s_add_u32 m0, %[l0], 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[ob], %[fp]\n\t"
            "global_load_lds_dwordx4 %[ob], %[fp] offset:1024\n\t"
            "s_mov_b64 %[ex], exec\n\ts_and_b64 exec, %[ex], 0xffff\n\tglobal_load_lds_dwordx4 %[ob], %[fp] offset:2048\n\ts_mov_b64 exec, %[ex]\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep), [ex] "=&s"(ex)
            : [l0] "s"(l0), [oa] "v"(oFa), [ob] "v"(oFb), [fp] "s"(fptr)
            : "memory", "scc");
    };
    auto landedOperands = [&](const Flight& f, unsigned& t1, unsigned& t2, double& fe, double& fo) {
        asm volatile("s_waitcnt vmcnt(0) ; retires %[i1] %[i2] %[ie] %[io]\n\t"
                     "v_mov_b32 %[t1], %[i1]\n\tv_mov_b32 %[t2], %[i2]\n\tv_mov_b64 %[fe], %[ie]\n\tv_mov_b64 %[fo], %[io]"
                     : [t1] "=&v"(t1), [t2] "=&v"(t2), [fe] "=&v"(fe), [fo] "=&v"(fo)
                     : [i1] "v"(f.t1), [i2] "v"(f.t2), [ie] "v"(f.sc.x), [io] "v"(f.sc.y) : "memory");
    };
    Flight F;
    F.sc = v2d{1.0, 1.0}; F.t1 = F.t2 = 0u;
    issue(F, dp[0]);
    for (int k = 0; k < nOps; k++) {               // (one more readable descriptor and stream entry follow the segment: its trailing no-ops)
        const WalkOp& d = dp[k];
        const unsigned flg = d.flags;
        const int k1 = (flg >> 5) & 7, k2 = (flg >> 8) & 7, hslot = (flg >> 11) & 3, smode = (flg >> 13) & 3;
        unsigned t1, t2;
        double fe, fo;
        landedOperands(F, t1, t2, fe, fo);
        dma(fs + (size_t)(k + 1) * fsStep, (unsigned)((k + 1) & 1));
        issue(F, dp[k + 1]);
        const int se1 = (int)(t1 & 0xffu), so1 = (int)(t1 >> 8) & 0xff, se2 = (int)(t2 & 0xffu), so2 = (int)(t2 >> 8) & 0xff;
        const double* frag = wtLds + (size_t)(k & 1) * fragD + (size_t)c * 2 * WT_FRAG;
        double te[WT_NT], to[WT_NT];
        if (k2 == WK_ACC) walkChild5(frag + WT_FRAG, S, false, S, S, ACC, g, fl, te, to);
        else {
            v2d b2[WT_NT];
            if (k2 == WK_MEM) tiledLoadB<WT_NT, EXACT>(d.src2, tileBase, S, g, m, b2);
            walkChild5(frag + WT_FRAG, S, k2 == WK_TIPS, se2, so2, b2, g, fl, te, to);
        }
        const bool rd = smode == WS_READ, wr = smode == WS_WRITE;
        const double inve = rd ? 1.0 / fe : 1.0, invo = rd ? 1.0 / fo : 1.0;
        {
            v2d b1[WT_NT];
            if (k1 == WK_MEM) tiledLoadB<WT_NT, EXACT>(d.src1, tileBase, S, g, m, b1);
            else if (k1 == WK_H0) {
#pragma unroll
                for (int j = 0; j < WT_NT; j++) b1[j] = H0[j];
            } else if (k1 > WK_H0) {
#pragma unroll
                for (int j = 0; j < WT_NT; j++) b1[j] = H1[j];
            }
            double re[WT_NT], ro[WT_NT];
            walkChild5(frag, S, k1 == WK_TIPS, se1, so1, b1, g, fl, re, ro);
#pragma unroll
            for (int j = 0; j < WT_NT; j++) ACC[j] = v2d{re[j] * te[j] * inve, ro[j] * to[j] * invo};
        }
        // the maxima over the states, exchanged by every micro-operation (the header above: straight-line, a multiplication by 1.0 changes no bit)
        v2d* mxk = mx + (size_t)(k & 1) * C * 16;
        {
            double me = 0.0, mo = 0.0;
#pragma unroll
            for (int j = 0; j < WT_NT; j++)
                if (EXACT || 4 * j + g < S) { me = fmax(me, ACC[j].x); mo = fmax(mo, ACC[j].y); }
            me = maxOverRows(me); mo = maxOverRows(mo);
            mxk[c * 16 + m] = v2d{me, mo};
        }
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            double me = 0.0, mo = 0.0;
#pragma unroll
            for (int cc = 0; cc < WALK_T32_WRITE_MAX_CATEGORIES; cc++) {
                const v2d v = mxk[(cc < C ? cc : C - 1) * 16 + m];
                me = fmax(me, v.x); mo = fmax(mo, v.y);
            }
            if (!(me > 0.0)) me = 1.0;
            if (!(mo > 0.0)) mo = 1.0;
            if (wr && c == 0 && g == 0) { if (ine) d.scaleW[pe] = me; if (ino) d.scaleW[pe + 1] = mo; }
            const double ie = wr ? 1.0 / me : 1.0, io = wr ? 1.0 / mo : 1.0;
#pragma unroll
            for (int j = 0; j < WT_NT; j++) ACC[j] = v2d{ACC[j].x * ie, ACC[j].y * io};
        }
        if (flg & WF_STORE) {
            char* dst = reinterpret_cast<char*>(d.store + tileBase);
#pragma unroll
            for (int j = 0; j < WT_NT; j++) {
                if (EXACT || 4 * j + g < S) {
                    double MI355_GLOBAL* q = gptr(reinterpret_cast<double*>(dst + (lane8 + (unsigned)j * 4u * TILE * 8u)));
                    if (ine && ino) __builtin_nontemporal_store(ACC[j], reinterpret_cast<v2d MI355_GLOBAL*>(q));
                    else { if (ine) q[0] = ACC[j].x; if (ino) q[1] = ACC[j].y; }
                }
            }
        }
        if (hslot == 1) {
#pragma unroll
            for (int j = 0; j < WT_NT; j++) H0[j] = ACC[j];
        } else if (hslot == 2) {
#pragma unroll
            for (int j = 0; j < WT_NT; j++) H1[j] = ACC[j];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
static size_t walkT32W1Lds(int C) { return (size_t)2 * C * 2 * WT_FRAG * sizeof(double) + (size_t)2 * C * 16 * sizeof(v2d); }

// ---- the pattern walk at 21..64 states (codon models): k_walkT64 ---------------------------------------------------------------
// The level kernel k_pruneTiled<16> stores every node and reads it back — 16 GB per evaluation of config C, and rebuilt without its
// stores it runs in 2.5 ms instead of 4.1 (profiles/r03_experiments.txt 11).  The walk's idea carries over once more, with two
// differences that follow from the size of things at 64 states:
//  * a wave's running result (ACC) is 16 v2d = 64 registers, a hold slot would be 16 KiB of LDS per wave: there are NO hold slots.
//    The planner is initialised with none (planner.h): its definitions are ladders (a tip-tip node under further tips), a node over
//    two evaluated children takes one of them through ACC and the other one from memory — every real node is stored, a stored node
//    is read back at most once, and only as the sibling of a subtree that was evaluated in registers.
//  * a branch matrix is 32 KiB of A fragments per category.  A workgroup is four tiles of ONE category, 64 KiB of LDS for the two
//    matrices of the current micro-operation (two workgroups per CU).  The halves are refilled separately by LDS-DMA
//    (global_load_lds_dwordx4: no registers, no ds_write): a stage multiplies by the second child's matrix first; behind the barrier
//    that ends that phase the next micro-operation's second matrix streams into the half just freed while the first child's MFMAs
//    run, and behind the barrier that ends those the next first matrix follows, under the products, the stores and the next second
//    phase.  Two barriers per micro-operation, each in front of ~512 MFMAs per wave (3.4 us) unless the child is a tip.
// Fragment layout of a matrix in the stream and in LDS: [jt][it / 2][q][it & 1] doubles — the A operands of row tiles 2i and 2i + 1
// for column tile jt are ONE 16-byte LDS read per lane (16 distinct addresses = 256 contiguous bytes per wave: every bank once),
// q = 4 * (column in tile) + (row in tile) as everywhere in this file.  Accumulation order per result element: column tiles
// ascending, exactly k_pruneTiled's — the walk and the level kernel agree bit for bit (tests/test_gpu_t64_walk.py).
// Read-mode rescaling and unscaled lists only (a pattern's write-mode factor needs all categories of the pattern): lists that
// rescale in write mode take the level path (engine_levels.cpp runOperations), on operands materialised first.
constexpr int W64_NT = 16, W64_FRAG = W64_NT * W64_NT * 16;          // doubles per matrix (4096 = 32 KiB)
#ifndef W64_RING
#define W64_RING 4                                                    // row tiles of a streamed operand in flight
#endif
#if defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_T64_NOMFMA)       // TIMING EXPERIMENTS ONLY: one of every sixteen matrix instructions (keeps the operands alive)
#define W64_MFMA(a, b, c, sel) ((sel) ? (c) : mfma4(a, b, c))
#else
#define W64_MFMA(a, b, c, sel) mfma4(a, b, c)
#endif

__global__ void k_gatherFragments64(const WalkOp* __restrict__ prog, int n, int C, int S, v2d* __restrict__ stream) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 16-byte pair
    constexpr int HALF = W64_FRAG / 2;
    if (t >= (size_t)n * C * 2 * HALF) return;
    const int r = (int)(t % HALF), child = (int)((t / HALF) & 1), c = (int)((t / (2 * HALF)) % C), k = (int)(t / ((size_t)2 * HALF * C));
    const int q = r & 15, ip = (r >> 4) & 7, jt = r >> 7;
    const int i0 = 8 * ip + (q & 3), j = 4 * jt + (q >> 2);                  // rows of tiles 2 ip and 2 ip + 1: i0, i0 + 4
    const double MI355_GLOBAL* M = gptr(child ? prog[k].m2 : prog[k].m1) + (size_t)c * S * S;
    v2d v;
    v.x = (i0 < S && j < S) ? M[(size_t)i0 * S + j] : 0.0;
    v.y = (i0 + 4 < S && j < S) ? M[(size_t)(i0 + 4) * S + j] : 0.0;
    stream[t] = v;
}

// A child's partials in memory are STREAMED: row tile jt of the operand (16 bytes per lane) is requested four column steps — 128
// MFMAs — before its products, through a ring of four registers pairs, instead of all 64 registers up front: with the whole operand
// loaded first a stage held an operand, two sets of sixteen results and the fragments — the compiler wanted 338 registers, spilled
// at 256 and serialised the loads behind the spills (10 us per operand: profiles/r06_experiments.txt 17).
template <bool EXACT>
__device__ __forceinline__ v2d w64LoadRowTile(const char* __restrict__ x, unsigned lane8, int S, int g, int m, int jt) {
    if (jt < W64_NT - 1 || !EXACT) {
        if (EXACT) return __builtin_nontemporal_load(gptr(reinterpret_cast<const v2d*>(x + (lane8 + (unsigned)jt * 4u * TILE * 8u))));
    }
    const int j = 4 * jt + g, jc = j < S ? j : S - 1;          // rows >= S do not exist in the buffer: the last real row, zeroed
    const v2d v = __builtin_nontemporal_load(gptr(reinterpret_cast<const v2d*>(x + (unsigned)(jc * TILE + 2 * m) * 8u)));
    return j < S ? v : v2d{0.0, 0.0};
}

// T[it] = { sum_j M[4 it + g][j] X[j][2m], ... X[j][2m + 1] } for all sixteen parent-state tiles.  The operand X: registers (b; STREAM
// false) or memory (x: the tile's first byte; q holds row tiles 0..3 on entry).  64 steps of two 16-byte fragment reads and eight MFMAs
// (128 cycles of the matrix pipe); the next step's fragments are requested before this step's MFMAs, and the scheduler is kept from
// pulling more forward.  Even- and odd-pattern sums share a 16-byte register tuple: that is what is stored and multiplied later.
template <bool EXACT, bool STREAM>
__device__ __forceinline__ void walkAcc16(const v2d* __restrict__ frag, int nt, int S, const v2d (&b)[W64_NT], const char* __restrict__ x, v2d (&q)[W64_RING],
                                          unsigned lane8, int g, int m, int fl, v2d (&T)[W64_NT]) {
#pragma unroll
    for (int it = 0; it < W64_NT; it++) T[it] = v2d{0.0, 0.0};
    v2d a0 = frag[fl], a1 = frag[16 + fl];
    v2d bj = v2d{0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * W64_NT; s++) {
        const int jt = s >> 2, ip = (s & 3) * 2;
        const int sn = s + 1, jn = sn >> 2, in = (sn & 3) * 2;
        if ((s & 3) == 0) {
            if (STREAM) {
                bj = q[jt % W64_RING];
                if (jt + W64_RING < W64_NT && (EXACT || jt + W64_RING < nt)) q[jt % W64_RING] = w64LoadRowTile<EXACT>(x, lane8, S, g, m, jt + W64_RING);
            } else bj = b[jt];
        }
        v2d n0 = a0, n1 = a1;
        if (sn < 4 * W64_NT && (EXACT || jn < nt)) { n0 = frag[(jn * 8 + in) * 16 + fl]; n1 = frag[(jn * 8 + in + 1) * 16 + fl]; }
        if (EXACT || (jt < nt && 2 * ip < nt)) {
            T[2 * ip].x = W64_MFMA(a0.x, bj.x, T[2 * ip].x, jt != 0);
            T[2 * ip].y = W64_MFMA(a0.x, bj.y, T[2 * ip].y, jt != 0);
            T[2 * ip + 1].x = W64_MFMA(a0.y, bj.x, T[2 * ip + 1].x, jt != 0);
            T[2 * ip + 1].y = W64_MFMA(a0.y, bj.y, T[2 * ip + 1].y, jt != 0);
        }
        if (EXACT || (jt < nt && 2 * ip + 2 < nt)) {
            T[2 * ip + 2].x = W64_MFMA(a1.x, bj.x, T[2 * ip + 2].x, jt != 0);
            T[2 * ip + 2].y = W64_MFMA(a1.x, bj.y, T[2 * ip + 2].y, jt != 0);
            T[2 * ip + 3].x = W64_MFMA(a1.y, bj.x, T[2 * ip + 3].x, jt != 0);
            T[2 * ip + 3].y = W64_MFMA(a1.y, bj.y, T[2 * ip + 3].y, jt != 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0; a1 = n1;
    }
}
// a compact tip: column `state` of the matrix (ones for a missing state)
__device__ __forceinline__ void walkTip16(const v2d* __restrict__ frag, int S, int se, int so, int g, v2d (&T)[W64_NT]) {
    const bool ge = se < S, go = so < S;
    const v2d* fe = frag + (ge ? (se >> 2) * 128 + (se & 3) * 4 + g : 0);
    const v2d* fo = frag + (go ? (so >> 2) * 128 + (so & 3) * 4 + g : 0);
#pragma unroll
    for (int ip = 0; ip < W64_NT / 2; ip++) {
        const v2d ve = fe[ip * 16], vo = fo[ip * 16];
        T[2 * ip] = v2d{ge ? ve.x : 1.0, go ? vo.x : 1.0};
        T[2 * ip + 1] = v2d{ge ? ve.y : 1.0, go ? vo.y : 1.0};
    }
}

template <bool EXACT>
__global__ __launch_bounds__(MF_BLOCK, 2) void k_walkT64(const WalkOp* __restrict__ prog, const WalkSeg* __restrict__ segs,
                                                         const double* __restrict__ fragStream, int P, int S, int C) {
    extern __shared__ double w64Lds[];             // [first child's matrix | second child's matrix], W64_FRAG doubles each
    const WalkSeg& sg = segs[blockIdx.y / C];
    const int c = blockIdx.y % C;
    const int nt = EXACT ? W64_NT : (S + 3) >> 2;
    const int ntile = (P + TILE - 1) / TILE;
    const int tile1 = (sg.pEnd + TILE - 1) / TILE;
    const int tileB = sg.pStart / TILE + (int)blockIdx.x * 4;
    if (tileB >= tile1) return;                    // the whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int fl = g * 4 + (lane & 3);
    const bool active = tileB + wave < tile1;      // a wave past the end walks the last tile along (barriers, staging) and stores nothing
    const int tile = active ? tileB + wave : tile1 - 1;
    const size_t tileBase = ((size_t)c * ntile + tile) * S * TILE;
    const int pe = tile * TILE + 2 * m;
    const bool ine = active && pe >= sg.pStart && pe < sg.pEnd, ino = active && pe + 1 >= sg.pStart && pe + 1 < sg.pEnd;
    const unsigned lane8 = (unsigned)(g * TILE + 2 * m) * 8u;
    const int nOps = sg.progCount;
    const WalkOp* dp = prog + sg.progStart;
    // stream entry of micro-operation k, this category: [m1's fragments | m2's fragments]
    const char MI355_GLOBAL* fs = reinterpret_cast<const char MI355_GLOBAL*>(gptr(fragStream)) + ((size_t)sg.progStart * C + c) * (2 * W64_FRAG * sizeof(double));
    const size_t fsStep = (size_t)C * 2 * W64_FRAG * sizeof(double);
    constexpr unsigned HALF_BYTES = W64_FRAG * sizeof(double);            // 32 KiB
    const unsigned ldsBase = (unsigned)__builtin_amdgcn_groupstaticsize();
    const unsigned oLane = (unsigned)lane * 16u;
    // a matrix's 32 KiB into LDS half `half` (0: first child's, 1: second child's): 32 pieces of 1 KiB, eight per wave
    auto dma = [&](const char MI355_GLOBAL* src, unsigned half) {
        unsigned keep;
        asm volatile("s_mov_b32 %[keep], m0" : [keep] "=&s"(keep) :: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned piece = (unsigned)(wave * 8 + i) * 1024u;
            const unsigned l = __builtin_amdgcn_readfirstlane(ldsBase + half * HALF_BYTES + piece);
            const char MI355_GLOBAL* p = src + piece;
            asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o], %[p]"
                         :: [l] "s"(l), [o] "v"(oLane), [p] "s"(p) : "memory");
        }
        asm volatile("s_mov_b32 m0, %[keep]" :: [keep] "s"(keep) : "memory");
    };
    const v2d* fragA = reinterpret_cast<const v2d*>(w64Lds);
    const v2d* fragB = fragA + W64_FRAG / 2;
    // What a stage needs from memory, and when it is asked for (loads return in the order they were issued; the waits in front of the
    // two barriers count on that, as kernels_walk4.hip's do):
    //  * its state codes and raw scale factors (two ushorts, 16 bytes) and its descriptor: a stage ahead, behind the barrier that ends
    //    the stage before (unused operands are readable dummies: the same loads go out whatever the kinds are);
    //  * a child's partials: streamed (walkAcc16); the first four row tiles of the SECOND child together with the small operands, the
    //    first four of the FIRST child in front of the barrier that ends the second child's phase — always four loads (a child that is
    //    no buffer: four loads from the fragment stream, never used), so that the barrier's wait is "all but the four youngest";
    //  * the matrices: see the header.
    const char* dummyRows = reinterpret_cast<const char*>(fragStream);
    struct Small { unsigned t1, t2; v2d sc; };
    auto small = [&](const WalkOp& d) {
        Small r;
        r.t1 = *reinterpret_cast<const unsigned short MI355_GLOBAL*>(gptr(reinterpret_cast<const char*>(d.src1)) + pe);
        r.t2 = *reinterpret_cast<const unsigned short MI355_GLOBAL*>(gptr(reinterpret_cast<const char*>(d.src2)) + pe);
        r.sc = *reinterpret_cast<const v2d MI355_GLOBAL*>(gptr(d.scale) + pe);
        return r;
    };
    auto head = [&](const char* x, v2d (&q)[W64_RING]) {
#pragma unroll
        for (int j = 0; j < W64_RING; j++) q[j] = __builtin_nontemporal_load(gptr(reinterpret_cast<const v2d*>(x + (lane8 + (unsigned)j * 4u * TILE * 8u))));
    };
    auto rowsOf = [&](const void* buf) { return reinterpret_cast<const char*>(reinterpret_cast<const double*>(buf) + tileBase); };
    dma(fs + HALF_BYTES, 1u);
    dma(fs, 0u);
    Small sm = small(dp[0]);
    v2d q2[W64_RING];
    head(((dp[0].flags >> 8) & 7) == WK_MEM ? rowsOf(dp[0].src2) : dummyRows, q2);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    v2d ACC[W64_NT];
#pragma unroll
    for (int k = 0; k < W64_NT; k++) ACC[k] = v2d{1.0, 1.0};
    for (int k = 0; k < nOps; k++) {
        const WalkOp& d = dp[k];
        const unsigned flg = d.flags;
        const int k1 = (flg >> 5) & 7, k2 = (flg >> 8) & 7;
        const bool rd = ((flg >> 13) & 3) == WS_READ;
        const int se1 = (int)(sm.t1 & 0xffu), so1 = (int)(sm.t1 >> 8) & 0xff, se2 = (int)(sm.t2 & 0xffu), so2 = (int)(sm.t2 >> 8) & 0xff;
        const double inve = rd ? 1.0 / sm.sc.x : 1.0, invo = rd ? 1.0 / sm.sc.y : 1.0;
        // ---- second child (the running result is consumed where it stands)
        v2d T[W64_NT];
        if (k2 == WK_TIPS) walkTip16(fragB, S, se2, so2, g, T);
        else if (k2 == WK_ACC) walkAcc16<EXACT, false>(fragB, nt, S, ACC, nullptr, q2, lane8, g, m, fl, T);
#if !(defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_T64_NOLOAD))      // TIMING EXPERIMENTS ONLY (tools/build_mfma_variant.sh; wrong results)
        else walkAcc16<EXACT, true>(fragB, nt, S, ACC, rowsOf(d.src2), q2, lane8, g, m, fl, T);
#else
        else walkAcc16<EXACT, false>(fragB, nt, S, ACC, nullptr, q2, lane8, g, m, fl, T);
#endif
        // the first child's first row tiles (or four loads of nothing), then: everybody is done with the second matrix, and this
        // micro-operation's first one has landed (requested a phase ago: older than the four)
        v2d q1[W64_RING];
        head(k1 == WK_MEM ? rowsOf(d.src1) : dummyRows, q1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(W64_RING) : "memory");
#if !(defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_T64_NODMA))
        if (k + 1 < nOps) dma(fs + (size_t)(k + 1) * fsStep + HALF_BYTES, 1u);
#endif
        // ---- first child, then the product: first * second * 1/factor, k_pruneTiled's order of operations
        {
            v2d R[W64_NT];
            if (k1 == WK_TIPS) walkTip16(fragA, S, se1, so1, g, R);
#if !(defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_T64_NOLOAD))
            else walkAcc16<EXACT, true>(fragA, nt, S, ACC, rowsOf(d.src1), q1, lane8, g, m, fl, R);
#else
            else walkAcc16<EXACT, false>(fragA, nt, S, T, nullptr, q1, lane8, g, m, fl, R);
#endif
#pragma unroll
            for (int j = 0; j < W64_NT; j++) ACC[j] = v2d{R[j].x * T[j].x * inve, R[j].y * T[j].y * invo};
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (k + 1 < nOps) {
#if !(defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_T64_NODMA))
            dma(fs + (size_t)(k + 1) * fsStep, 0u);
#endif
            const WalkOp& dn = dp[k + 1];
            sm = small(dn);
            head(((dn.flags >> 8) & 7) == WK_MEM ? rowsOf(dn.src2) : dummyRows, q2);
        }
#if defined(BEAGLE_MI355_LAB) && defined(MI355_EXP_T64_NOSTORE)
        if ((flg & WF_STORE) && ACC[0].x == -1.0) {
#else
        if (flg & WF_STORE) {
#endif
            char* dst = reinterpret_cast<char*>(d.store + tileBase);
#pragma unroll
            for (int j = 0; j < W64_NT; j++) {
                if ((EXACT && j < W64_NT - 1) || 4 * j + g < S) {
                    double MI355_GLOBAL* q = gptr(reinterpret_cast<double*>(dst + (lane8 + (unsigned)j * 4u * TILE * 8u)));
                    if (ine && ino) __builtin_nontemporal_store(ACC[j], reinterpret_cast<v2d MI355_GLOBAL*>(q));
                    else { if (ine) q[0] = ACC[j].x; if (ino) q[1] = ACC[j].y; }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// LDS per workgroup: 12.5 KiB of fragments + 20 KiB per hold slot (2 slots: 3 workgroups per CU, 3: 2)
static size_t walkT32Lds(int holdSlots) { return (size_t)4 * WT_FRAG * sizeof(double) + (size_t)holdSlots * 4 * WT_HOLD_V2D * sizeof(v2d); }

// the fragment stream of a device program of nEntries descriptors: [entry][category][child][25 tile pairs][16]
// (21..64 states, k_walkT64: [entry][category][child][16 x 8 tile pairs][16][2])
size_t walkT32StreamBytes(int nEntries, int C, int S) { return (size_t)nEntries * C * 2 * (S > 20 ? W64_FRAG : WT_FRAG) * sizeof(double); }
void launchGatherFragments(hipStream_t stream, const WalkOp* dProg, int nEntries, int C, int S, void* dStream) {
    if (nEntries <= 0) return;
    if (S > 20) {
        const size_t pairs = (size_t)nEntries * C * W64_FRAG;
        hipLaunchKernelGGL(k_gatherFragments64, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, dProg, nEntries, C, S, (v2d*)dStream);
        return;
    }
    const size_t total = (size_t)nEntries * C * 2 * WT_FRAG;
    hipLaunchKernelGGL(k_gatherFragments, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dProg, nEntries, C, S, (double*)dStream);
}
bool launchWalkT32(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, const void* dStream, int P, int S, int C, int holdSlots,
                   bool writeMode) {
    if (nSegs <= 0 || maxRange <= 0 || S < 16 || S > 64 || (size_t)nSegs * C > 65535) return false;
    if (S > 20) {                                  // 21..64 states: no hold slots, no write mode (k_walkT64)
        if (writeMode) return false;
        const size_t lds64 = (size_t)2 * W64_FRAG * sizeof(double);
        if (!grantDynamicLds(reinterpret_cast<const void*>(k_walkT64<true>), lds64) || !grantDynamicLds(reinterpret_cast<const void*>(k_walkT64<false>), lds64)) return false;
        const dim3 grid64((maxRange + 4 * TILE - 1) / (4 * TILE) + 1, nSegs * C), block64(MF_BLOCK);      // (+ 1: a range that starts inside a group of four tiles)
        if (S > 60) hipLaunchKernelGGL(k_walkT64<true>, grid64, block64, lds64, stream, dProg, dSegs, (const double*)dStream, P, S, C);
        else hipLaunchKernelGGL(k_walkT64<false>, grid64, block64, lds64, stream, dProg, dSegs, (const double*)dStream, P, S, C);
        return true;
    }
    if (writeMode) {                               // a program with write-mode micro-operations: all categories of a tile in one workgroup
        if (C < 1 || C > WALK_T32_WRITE_MAX_CATEGORIES || holdSlots > WALK_T32_WRITE_MAX_HOLD) return false;
        // one tile per workgroup, hold slots in registers (k_walkT32W1)
        if (!grantDynamicLds(reinterpret_cast<const void*>(k_walkT32W1<true>), 160 * 1024) ||
            !grantDynamicLds(reinterpret_cast<const void*>(k_walkT32W1<false>), 160 * 1024)) return false;
        const dim3 grid1((maxRange + TILE - 1) / TILE + 1, nSegs), block1(64 * C);      // (+ 1: a range that starts inside a tile)
        const size_t lds1 = walkT32W1Lds(C);
        if (S == 20) hipLaunchKernelGGL(k_walkT32W1<true>, grid1, block1, lds1, stream, dProg, dSegs, (const double*)dStream, P, S, C);
        else hipLaunchKernelGGL(k_walkT32W1<false>, grid1, block1, lds1, stream, dProg, dSegs, (const double*)dStream, P, S, C);
        return true;
    }
    const dim3 grid((maxRange + 4 * TILE - 1) / (4 * TILE), nSegs * C), block(MF_BLOCK);
    const size_t lds = walkT32Lds(holdSlots < 1 ? 1 : holdSlots > 3 ? 3 : holdSlots);
    if (S == 20) hipLaunchKernelGGL(k_walkT32<true>, grid, block, lds, stream, dProg, dSegs, (const double*)dStream, P, S, C);
    else hipLaunchKernelGGL(k_walkT32<false>, grid, block, lds, stream, dProg, dSegs, (const double*)dStream, P, S, C);
    return true;
}

// root integration on the T32 layout: thread per pattern, consecutive lanes = consecutive patterns of a tile
__global__ __launch_bounds__(256) void k_rootSiteTiled(const double* __restrict__ root, const double* __restrict__ catWeights,
                                                       const double* __restrict__ freqs, const double* __restrict__ cum, int cumIsRaw,
                                                       const double* __restrict__ patternWeights, double* __restrict__ siteLogL,
                                                       double* __restrict__ blockSums, int P, int S, int C, int pStart, int pEnd) {
    // a workgroup = 64 patterns; wave w takes the states i = w, w + 4, ... of every category (a thread used to walk all C x S
    // entries of its pattern alone: 244 dependent loads at 61 states, 79 workgroups for 20 000 patterns — 106 us for 39 MB)
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int p = pStart + blockIdx.x * 64 + lane;
    const int ntile = (P + TILE - 1) / TILE;
    double sum = 0.0;
    if (p < pEnd) {
        const int tile = p / TILE, q = p - tile * TILE;
        for (int c = 0; c < C; c++) {
            const double* r = root + ((size_t)c * ntile + tile) * S * TILE + q;
            double s = 0.0;
            for (int i = w; i < S; i += 4) s += freqs[i] * r[(size_t)i * TILE];
            sum += catWeights[c] * s;
        }
    }
    part[w][lane] = sum;
    __syncthreads();
    if (w) return;
    double contrib = 0.0;
    if (p < pEnd) {
        double site = log((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
        if (cum) site += cumIsRaw ? log(cum[p]) : cum[p];
        siteLogL[p] = site;
        contrib = site * patternWeights[p];
    }
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
    if (lane == 0) blockSums[blockIdx.x] = contrib;
}

int rootSiteTiledBlocks(int patterns) { return (patterns + 63) / 64; }

void launchRootSiteTiled(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                         const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                         double* blockSums, int P, int S, int C, int pStart, int pEnd) {
    const int n = rootSiteTiledBlocks(pEnd - pStart);
    hipLaunchKernelGGL(k_rootSiteTiled, dim3(n), dim3(256), 0, stream, root, catWeights, freqs, cum, cumIsRaw,
                       patternWeights, siteLogL, blockSums, P, S, C, pStart, pEnd);
}

}  // namespace mi355
