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

// ---- the derivative sums of a whole list, nothing stored --------------------------------------------------------------
// The chain that evaluates gradients asks for updatePrePartials and then for the SUMS of calculateEdgeDifferentials; the
// pre-order partials themselves are rarely looked at.  k_preWalk4 therefore runs the whole list as a depth-first walk, like
// the post-order pattern walk (kernels_walk4.hip): a thread owns (pattern, category) — wave w of the workgroup = category w —
// and carries the pre-order partial of the node it stands on in registers; at a node it reads post(a) and post(b), forms
// both children's pre-order partials and both edges' contributions, steps into one child and parks the other child's
// partial in an LDS hold slot until that subtree is done (the host orders the walk smaller subtree first, so the slots in
// use never exceed log2 of the node count).  No pre-order partial goes to memory (the engine writes them when somebody
// asks: engine_preorder.cpp), so the pass reads each post-order partial once and writes a few doubles per edge and wave.
//
// The sum over categories inside a pattern's derivative, sum_c w_c num_c / sum_c w_c den_c, would need the categories'
// waves to meet at every edge.  It does not have to: den = sum_c w_c pre . post is the pattern's likelihood — the same on
// every edge up to the scale factors of the post-order partials: den(edges below a) = den(edges below parent(a)) * f_a, with
// f_a the factor post(a) was divided by.  So the denominator is formed ONCE, at the root, and because everything downstream
// is linear in the pre-order partial, what a thread carries is not pre(n) but weight_p w_c / den * pre(n): the factor is
// multiplied in at the root, a step into an internal child multiplies by 1 / f_a (read from the child's reciprocal scale
// array; ones when it carries none), and every edge's contribution is a plain dot product to be summed.  (The partials this
// walk parks in hold slots or stores for other segments are therefore scaled ones; the engine rewrites the real buffers
// whenever somebody asks for them.)
//
// Loads are software-pipelined by hand exactly as in k_walk4: while descriptor k computes, the loads of k + 1 are in flight,
// and the wait before k's arithmetic is for "all but the loads of k + 1" (preWait).  The branch matrices and the differential matrices arrive spread over the
// lanes of one register each (lane l = entry l & 15) and are applied with v_fmac_f64_dpp row_newbcast — no LDS, no SGPRs.
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
#define MI355_CONST __attribute__((address_space(4)))

struct PreFetched { v2d a0, a1, b0, b1; unsigned sa, sb, ta, tb; double mA, mB, dA, dB, ra, rb; };     // (ta, tb: the second tip of a PW_CHERRY child)
struct PreDesc { u64 postA, postB, tipA, tipB, storeA, storeB, recipA, recipB; int matA, matB, dA, dB, slotA, slotB; unsigned flags; };

__device__ __forceinline__ PreDesc loadPreDesc(const PreWalkOp MI355_CONST* p) {
    PreDesc d;
    d.postA = (u64)p->postA; d.postB = (u64)p->postB; d.tipA = (u64)p->tipA; d.tipB = (u64)p->tipB; d.storeA = (u64)p->storeA; d.storeB = (u64)p->storeB; d.recipA = (u64)p->recipA; d.recipB = (u64)p->recipB;
    d.matA = p->matA; d.matB = p->matB; d.dA = p->dA; d.dB = p->dB; d.slotA = p->slotA; d.slotB = p->slotB; d.flags = p->flags;
    return d;
}
__device__ __forceinline__ void preIssue(PreFetched& f, const PreDesc& d, unsigned oPart, unsigned oTip, unsigned oMat, unsigned oRecip, u64 matrices, u64 products, unsigned matBytes) {
    const u64 mA = matrices + (u64)(unsigned)d.matA * matBytes, mB = matrices + (u64)(unsigned)d.matB * matBytes;
    const u64 dA = products + (u64)(unsigned)d.dA * matBytes, dB = products + (u64)(unsigned)d.dB * matBytes;       // (k_edgeProducts: branch matrix . differential matrix, per edge)
    // a compact tip is one byte, a child with partials two 16-byte loads: what is not needed is BRANCHED around (a vector-memory
    // instruction occupies the address unit whatever it fetches, and half the children of a tree are tips)
    // (an unstored node over two tips, PW_CHERRY: its two matrices interleaved in ONE 16-byte load per lane — into the first of the
    // registers the partials would have filled —, its tips' states into two registers: a d16 load into half a register clears the other
    // half on this chip)
    asm volatile(
        "s_bitcmp1_b32 %[fl], 25\n\t"
        "s_cbranch_scc1 .Lqa%=\n\t"
        "s_bitcmp1_b32 %[fl], 0\n\t"
        "s_cbranch_scc1 .Lpa%=\n\t"
        "global_load_dwordx4 %[a0], %[oP], %[pA]\n\t"
        "global_load_dwordx4 %[a1], %[oP], %[pA] offset:16\n\t"
        "s_branch .Lpb%=\n"
        ".Lqa%=:\n\t"
        "global_load_dwordx4 %[a0], %[oM2], %[pA]\n\t"
        "global_load_ubyte %[sa], %[oT], %[tA]\n\t"
        "global_load_ubyte %[ta], %[oT], %[uA]\n\t"
        "s_branch .Lpb%=\n"
        ".Lpa%=:\n\t"
        "global_load_ubyte %[sa], %[oT], %[tA]\n"
        ".Lpb%=:\n\t"
        "s_bitcmp1_b32 %[fl], 26\n\t"
        "s_cbranch_scc1 .Lqc%=\n\t"
        "s_bitcmp1_b32 %[fl], 1\n\t"
        "s_cbranch_scc1 .Lpc%=\n\t"
        "global_load_dwordx4 %[b0], %[oP], %[pB]\n\t"
        "global_load_dwordx4 %[b1], %[oP], %[pB] offset:16\n\t"
        "s_branch .Lpd%=\n"
        ".Lqc%=:\n\t"
        "global_load_dwordx4 %[b0], %[oM2], %[pB]\n\t"
        "global_load_ubyte %[sb], %[oT], %[tB]\n\t"
        "global_load_ubyte %[tb], %[oT], %[uB]\n\t"
        "s_branch .Lpd%=\n"
        ".Lpc%=:\n\t"
        "global_load_ubyte %[sb], %[oT], %[tB]\n"
        ".Lpd%=:\n\t"
        "global_load_dwordx2 %[mA], %[oM], %[smA]\n\t"
        "global_load_dwordx2 %[mB], %[oM], %[smB]\n\t"
        "global_load_dwordx2 %[dA], %[oM], %[sdA]\n\t"
        "global_load_dwordx2 %[dB], %[oM], %[sdB]\n\t"
        "global_load_dwordx2 %[ra], %[oR], %[srA]\n\t"
        "global_load_dwordx2 %[rb], %[oR], %[srB]"
        : [a0] "+v"(f.a0), [a1] "+v"(f.a1), [b0] "+v"(f.b0), [b1] "+v"(f.b1), [sa] "+v"(f.sa), [sb] "+v"(f.sb), [ta] "+v"(f.ta), [tb] "+v"(f.tb),
          [mA] "+v"(f.mA), [mB] "+v"(f.mB), [dA] "+v"(f.dA), [dB] "+v"(f.dB), [ra] "+v"(f.ra), [rb] "+v"(f.rb)
        : [fl] "s"(d.flags), [oP] "v"(oPart), [oT] "v"(oTip), [oM] "v"(oMat), [oM2] "v"(oMat * 2u), [oR] "v"(oRecip), [pA] "s"(d.postA), [pB] "s"(d.postB), [tA] "s"(d.tipA), [tB] "s"(d.tipB),
          [uA] "s"(d.storeA), [uB] "s"(d.storeB),
          [smA] "s"(mA), [smB] "s"(mB), [sdA] "s"(dA), [sdB] "s"(dB), [srA] "s"(d.recipA), [srB] "s"(d.recipB)
        : "memory", "scc");
}
// the loads of `f` have landed once at most as many loads as the FOLLOWING descriptor issued (8..12: kernels.h preWalkLoads, bits
// 28..31 of its flags) are outstanding: loads return in issue order, and stores in the queue only make the wait stricter
__device__ __forceinline__ void preWait(PreFetched& f, unsigned nextLoads) {
    asm volatile(
        "s_cmp_eq_u32 %[nl], 8\n\t"
        "s_cbranch_scc1 .Lw8%=\n\t"
        "s_cmp_eq_u32 %[nl], 9\n\t"
        "s_cbranch_scc1 .Lw9%=\n\t"
        "s_cmp_eq_u32 %[nl], 10\n\t"
        "s_cbranch_scc1 .Lw10%=\n\t"
        "s_cmp_eq_u32 %[nl], 11\n\t"
        "s_cbranch_scc1 .Lw11%=\n\t"
        "s_cmp_eq_u32 %[nl], 12\n\t"
        "s_cbranch_scc1 .Lw12%=\n\t"
        "s_waitcnt vmcnt(0)\n\t"                         // (no descriptor says anything else; waiting for everything is always right)
        "s_branch .Lwd%=\n"
        ".Lw12%=:\n\t"
        "s_waitcnt vmcnt(12)\n\t"
        "s_branch .Lwd%=\n"
        ".Lw11%=:\n\t"
        "s_waitcnt vmcnt(11)\n\t"
        "s_branch .Lwd%=\n"
        ".Lw10%=:\n\t"
        "s_waitcnt vmcnt(10)\n\t"
        "s_branch .Lwd%=\n"
        ".Lw9%=:\n\t"
        "s_waitcnt vmcnt(9)\n\t"
        "s_branch .Lwd%=\n"
        ".Lw8%=:\n\t"
        "s_waitcnt vmcnt(8)\n"
        ".Lwd%=: ; retires %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13"
        : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.sa), "+v"(f.sb), "+v"(f.mA), "+v"(f.mB), "+v"(f.dA), "+v"(f.dB), "+v"(f.ra), "+v"(f.rb), "+v"(f.ta), "+v"(f.tb)
        : [nl] "s"(nextLoads) : "memory", "scc");
}
// (Measured and dropped, round 5: touching the post-order partials of the descriptor TWO ahead — one byte per lane, so that the
// real loads a stage later find them on their way — 6.72 instead of 5.98 ms per gradient at 1e5 patterns: the touches are vector-
// memory instructions and L2 requests of their own.)
// (ya, yb) = (MA xa, MB xb), the matrices spread over the lanes (lane l = entry l & 15, row-major): eight independent chains
__device__ __forceinline__ void matvecDppPair(const double mA, const v4d xa, const double mB, const v4d xb, v4d& ya, v4d& yb) {
    double a0, a1, a2, a3, b0, b1, b2, b3;
    const double p0 = xa.x, p1 = xa.y, p2 = xa.z, p3 = xa.w, q0 = xb.x, q1 = xb.y, q2 = xb.z, q3 = xb.w;
#define FA(Y, N, X) "v_fmac_f64_dpp %[" #Y "], %[mA], %[" #X "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
#define FB(Y, N, X) "v_fmac_f64_dpp %[" #Y "], %[mB], %[" #X "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile(
        "v_mov_b64 %[a0], 0\n\tv_mov_b64 %[a1], 0\n\tv_mov_b64 %[a2], 0\n\tv_mov_b64 %[a3], 0\n\t"
        "v_mov_b64 %[b0], 0\n\tv_mov_b64 %[b1], 0\n\tv_mov_b64 %[b2], 0\n\tv_mov_b64 %[b3], 0\n\t"
        FA(a0, 0, p0) FA(a1, 4, p0) FA(a2, 8, p0) FA(a3, 12, p0) FB(b0, 0, q0) FB(b1, 4, q0) FB(b2, 8, q0) FB(b3, 12, q0)
        FA(a0, 1, p1) FA(a1, 5, p1) FA(a2, 9, p1) FA(a3, 13, p1) FB(b0, 1, q1) FB(b1, 5, q1) FB(b2, 9, q1) FB(b3, 13, q1)
        FA(a0, 2, p2) FA(a1, 6, p2) FA(a2, 10, p2) FA(a3, 14, p2) FB(b0, 2, q2) FB(b1, 6, q2) FB(b2, 10, q2) FB(b3, 14, q2)
        FA(a0, 3, p3) FA(a1, 7, p3) FA(a2, 11, p3) FA(a3, 15, p3) FB(b0, 3, q3) FB(b1, 7, q3) FB(b2, 11, q3) FB(b3, 15, q3)
        "s_nop 0"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
        : [mA] "v"(mA), [mB] "v"(mB), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3));
    ya = v4d{a0, a1, a2, a3}; yb = v4d{b0, b1, b2, b3};
}
// the transposes: (ya, yb) = (MA^T xa, MB^T xb)
__device__ __forceinline__ void matvecDppPairT(const double mA, const v4d xa, const double mB, const v4d xb, v4d& ya, v4d& yb) {
    double a0, a1, a2, a3, b0, b1, b2, b3;
    const double p0 = xa.x, p1 = xa.y, p2 = xa.z, p3 = xa.w, q0 = xb.x, q1 = xb.y, q2 = xb.z, q3 = xb.w;
    asm volatile(
        "v_mov_b64 %[a0], 0\n\tv_mov_b64 %[a1], 0\n\tv_mov_b64 %[a2], 0\n\tv_mov_b64 %[a3], 0\n\t"
        "v_mov_b64 %[b0], 0\n\tv_mov_b64 %[b1], 0\n\tv_mov_b64 %[b2], 0\n\tv_mov_b64 %[b3], 0\n\t"
        FA(a0, 0, p0) FA(a1, 1, p0) FA(a2, 2, p0) FA(a3, 3, p0) FB(b0, 0, q0) FB(b1, 1, q0) FB(b2, 2, q0) FB(b3, 3, q0)
        FA(a0, 4, p1) FA(a1, 5, p1) FA(a2, 6, p1) FA(a3, 7, p1) FB(b0, 4, q1) FB(b1, 5, q1) FB(b2, 6, q1) FB(b3, 7, q1)
        FA(a0, 8, p2) FA(a1, 9, p2) FA(a2, 10, p2) FA(a3, 11, p2) FB(b0, 8, q2) FB(b1, 9, q2) FB(b2, 10, q2) FB(b3, 11, q2)
        FA(a0, 12, p3) FA(a1, 13, p3) FA(a2, 14, p3) FA(a3, 15, p3) FB(b0, 12, q3) FB(b1, 13, q3) FB(b2, 14, q3) FB(b3, 15, q3)
        "s_nop 0"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
        : [mA] "v"(mA), [mB] "v"(mB), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3));
    ya = v4d{a0, a1, a2, a3}; yb = v4d{b0, b1, b2, b3};
}
#undef FA
#undef FB
// one matrix: y = M x and y = M^T x (four independent chains)
#define FM(Y, N, X) "v_fmac_f64_dpp %[" #Y "], %[m], %[" #X "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ v4d matvecDpp(const double m, const v4d x) {
    double a0, a1, a2, a3;
    const double p0 = x.x, p1 = x.y, p2 = x.z, p3 = x.w;
    asm volatile(
        "v_mov_b64 %[a0], 0\n\tv_mov_b64 %[a1], 0\n\tv_mov_b64 %[a2], 0\n\tv_mov_b64 %[a3], 0\n\t"
        FM(a0, 0, p0) FM(a1, 4, p0) FM(a2, 8, p0) FM(a3, 12, p0) FM(a0, 1, p1) FM(a1, 5, p1) FM(a2, 9, p1) FM(a3, 13, p1)
        FM(a0, 2, p2) FM(a1, 6, p2) FM(a2, 10, p2) FM(a3, 14, p2) FM(a0, 3, p3) FM(a1, 7, p3) FM(a2, 11, p3) FM(a3, 15, p3)
        "s_nop 0"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
        : [m] "v"(m), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3));
    return v4d{a0, a1, a2, a3};
}
__device__ __forceinline__ v4d matvecDppT(const double m, const v4d x) {
    double a0, a1, a2, a3;
    const double p0 = x.x, p1 = x.y, p2 = x.z, p3 = x.w;
    asm volatile(
        "v_mov_b64 %[a0], 0\n\tv_mov_b64 %[a1], 0\n\tv_mov_b64 %[a2], 0\n\tv_mov_b64 %[a3], 0\n\t"
        FM(a0, 0, p0) FM(a1, 1, p0) FM(a2, 2, p0) FM(a3, 3, p0) FM(a0, 4, p1) FM(a1, 5, p1) FM(a2, 6, p1) FM(a3, 7, p1)
        FM(a0, 8, p2) FM(a1, 9, p2) FM(a2, 10, p2) FM(a3, 11, p2) FM(a0, 12, p3) FM(a1, 13, p3) FM(a2, 14, p3) FM(a3, 15, p3)
        "s_nop 0"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
        : [m] "v"(m), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3));
    return v4d{a0, a1, a2, a3};
}
#undef FM
// v + (the value another lane holds, 0 where the pattern has no source lane / the row is masked off)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dppAdd(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
// lane 63 <- the sum over the wave, in a fixed order (row_shr 1, 2, 4, 8: lane 15 of every row holds the row's sum; then
// row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3)
__device__ __forceinline__ double waveSumTo63(double v) {
    v = dppAdd<0x111, 0xf>(v); v = dppAdd<0x112, 0xf>(v); v = dppAdd<0x114, 0xf>(v); v = dppAdd<0x118, 0xf>(v);
    v = dppAdd<0x142, 0xa>(v); v = dppAdd<0x143, 0xc>(v);
    return v;
}

// the value lane `byteAddr / 4` holds (the LDS crossbar, no memory: the vector pipe is what this kernel is short of)
__device__ __forceinline__ double laneValue(int byteAddr, double v) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byteAddr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byteAddr, __double2loint(v)));
}
// M e_s for a compact tip: column s (the lane's own state) of the matrix spread over the lanes of its row; a missing state
// (s >= 4: the all-ones partial) takes the row sums.  Eight lane reads instead of sixteen multiply-adds and what surrounds them —
// half the children of a tree are tips.
__device__ __forceinline__ v4d columnDpp(const double m, const unsigned s, const int lane) {
    const int row = (lane & 48) << 2, base = row + (int)((s & 3u) << 2);
    v4d y = v4d{laneValue(base, m), laneValue(base + 16, m), laneValue(base + 32, m), laneValue(base + 48, m)};
    if (__builtin_amdgcn_ballot_w64(s >= 4u)) {                            // (wave-uniform: nothing of this for an alignment without gaps)
        double r = dppAdd<0xB1, 0xf>(m);                                  // quad_perm [1,0,3,2], [2,3,0,1]: every lane of a quad holds the quad's sum
        r = dppAdd<0x4E, 0xf>(r);
        int first = base & ~12;                                           // (= row; opaque, or four loop-invariant address registers are kept for it)
        asm volatile("" : "+v"(first));
        const v4d z = v4d{laneValue(first, r), laneValue(first + 16, r), laneValue(first + 32, r), laneValue(first + 48, r)};
        if (s >= 4u) y = z;
    }
    return y;
}
// lane 31 <- the wave's sum of a, lane 63 <- the wave's sum of b, in a fixed order: the halves of the wave change what they
// hold (lanes < 32: a + a of the other half; lanes >= 32: b + b of the other half), then one reduction serves both
__device__ __forceinline__ double waveSumPair(const double a, const double b, const int lane) {
    const bool upper = lane >= 32;
    const double keep = upper ? b : a, send = upper ? a : b;
    double v = keep + laneValue((lane ^ 32) << 2, send);
    v = dppAdd<0x111, 0xf>(v); v = dppAdd<0x112, 0xf>(v); v = dppAdd<0x114, 0xf>(v); v = dppAdd<0x118, 0xf>(v);
    return dppAdd<0x142, 0xa>(v);
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_preWalk4(const PreWalkOp MI355_CONST* __restrict__ prog, const PreWalkSeg MI355_CONST* __restrict__ segs,
                                                   const double* __restrict__ listRootPre, const double* __restrict__ matrices,
                                                   const double* __restrict__ products,
                                                   const double* __restrict__ catWeights, const double* __restrict__ patternWeights,
                                                   double* __restrict__ sums, int P, int C, int rootSegment, int holdSlots) {
    extern __shared__ v2d preLds[];                   // hold[slot][C][2][64] (v2d); slot 0 doubles as the categories' exchange at the start
    const PreWalkSeg MI355_CONST& sg = segs[blockIdx.y];
    const int nOps = sg.progCount;
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = (int)blockIdx.x * 64 + lane;
    const bool valid = p < P;
    const int q = valid ? p : P - 1;                  // lanes past the end recompute the last pattern and count for nothing
    const unsigned oPart = (unsigned)(((size_t)c * P + q) * 32), oTip = (unsigned)q, oMat = (unsigned)(c * 128 + (lane & 15) * 8);
    const unsigned oRecip = (unsigned)(walkPairIndex((size_t)q) * 8);     // (one partition: the pair-interleaved position of pattern q)
    const unsigned matBytes = (unsigned)C * 128u;
    const u64 mats = (u64)matrices, prods = (u64)products;
    const size_t waves = (size_t)gridDim.x * C, w = (size_t)blockIdx.x * C + c;
    v2d* holdBase = preLds + (size_t)c * 128 + lane;  // + slot * C * 128, second half at + 64
    const size_t holdStride = (size_t)C * 128;
    v2d* postBase = holdBase + (size_t)holdSlots * holdStride;       // the post slots (PW_POSTOP) behind the hold slots, same shape

    const PreWalkOp MI355_CONST* dp = prog + sg.progStart;
    v4d ACC = gptr(reinterpret_cast<const v4d*>(sg.rootPre))[(size_t)c * P + q];
    // the pattern's likelihood, once (the segment that starts at the list's root; the others start from partials that carry the
    // factor already): den = sum_c w_c sum_i pre(root)_i (MA xa)_i (MB xb)_i through the categories' exchange
    if (rootSegment) {
        const PreWalkOp MI355_CONST& r = prog[0];
        const v4d root = gptr(reinterpret_cast<const v4d*>(listRootPre))[(size_t)c * P + q];
        const v4d xa = (r.flags & PW_TIP_A) ? tipVector(gptr(r.tipA)[q]) : gptr(reinterpret_cast<const v4d*>(r.postA))[(size_t)c * P + q];
        const v4d xb = (r.flags & PW_TIP_B) ? tipVector(gptr(r.tipB)[q]) : gptr(reinterpret_cast<const v4d*>(r.postB))[(size_t)c * P + q];
        const v4d ua = matvec4(matrices + ((size_t)r.matA * C + c) * 16, xa), ub = matvec4(matrices + ((size_t)r.matB * C + c) * 16, xb);
        double* exch = reinterpret_cast<double*>(preLds);
        exch[c * 64 + lane] = catWeights[c] * dot4(root, ua * ub);
        __syncthreads();
        double den = 0.0;
        for (int cc = 0; cc < C; cc++) den += exch[cc * 64 + lane];
        __syncthreads();
        const double coef = valid ? patternWeights[p] * catWeights[c] / den : 0.0;
        ACC = ACC * coef;
    } else if (!valid) ACC = v4d{0.0, 0.0, 0.0, 0.0};            // (a lane past the end counts for nothing: the root's factor was 0 for it)
    // (the pipeline is set up behind the block above: the kernel's register count is decided where the two overlap)
    PreDesc D0 = loadPreDesc(dp), D1 = loadPreDesc(dp + 1);
    PreFetched A, B;
    A.a0 = A.a1 = A.b0 = A.b1 = v2d{1.0, 1.0}; A.sa = A.sb = A.ta = A.tb = 4u; A.mA = A.mB = A.dA = A.dB = 0.0; A.ra = A.rb = 1.0;
    B = A;
    preIssue(A, D0, oPart, oTip, oMat, oRecip, mats, prods, matBytes);

// One descriptor.  What a child costs depends on what it is: a compact tip contributes COLUMNS (columnDpp: of its branch matrix
// for the sibling's side, of the edge's product matrix for its own derivative) and nothing goes down its edge; a child with
// partials costs three matrix-vector products (its contribution, its derivative, the pre-order partial handed down).  The
// derivative of edge a is  w . (E_a x_a)  with  w = pre(n) * (M_b x_b)  and  E_a = M_a . D_a  (k_edgeProducts) — the same number
// as  (M_a^T w) . (D_a x_a)  without needing the child's pre-order partial for it.
#define PRE_STAGE(CUR, NXT, DCUR, DNXT)                                                                                     \
    {                                                                                                                     \
        preIssue(NXT, DNXT, oPart, oTip, oMat, oRecip, mats, prods, matBytes);                                            \
        const unsigned fl = DCUR.flags;                                                                                   \
        const int slotA = DCUR.slotA, slotB = DCUR.slotB;                                                                 \
        const u64 stA = DCUR.storeA, stB = DCUR.storeB;                                                                   \
        const unsigned src = (fl >> PW_SRC_SHIFT) & 15u, contA = (fl >> PW_CONT_A_SHIFT) & 15u, contB = (fl >> PW_CONT_B_SHIFT) & 15u; \
        /* (a node taken from a hold slot follows the end of a subtree: what the registers carried is not needed any more) */ \
        if (src) { const v2d* h = holdBase + (size_t)(src - 1) * holdStride; const v2d lo = h[0], hi = h[64]; ACC = v4d{lo.x, lo.y, hi.x, hi.y}; } \
        preWait(CUR, DNXT.flags >> PW_LOADS_SHIFT);                                                                       \
        const bool tipA = (fl & (PW_TIP_A | PW_SLOT_A)) == PW_TIP_A, tipB = (fl & (PW_TIP_B | PW_SLOT_B)) == PW_TIP_B;    \
        /* (an unstored operand comes out of its post slot into the registers the loads would have filled) */              \
        if (fl & PW_SLOT_A) { const v2d* h = postBase + (size_t)((fl >> PW_SLOTA_SHIFT) & 3u) * holdStride; CUR.a0 = h[0]; CUR.a1 = h[64]; } \
        if (fl & PW_SLOT_B) { const v2d* h = postBase + (size_t)((fl >> PW_SLOTB_SHIFT) & 3u) * holdStride; CUR.b0 = h[0]; CUR.b1 = h[64]; } \
        v4d xa = v4d{CUR.a0.x, CUR.a0.y, CUR.a1.x, CUR.a1.y}, xb = v4d{CUR.b0.x, CUR.b0.y, CUR.b1.x, CUR.b1.y};            \
        /* an unstored node over two tips: (M1 e_s1) * (M2 e_s2) / its own factor, from what came in its stead (kernels.h PW_CHERRY) */ \
        if (fl & PW_CHERRY_A) xa = columnDpp(CUR.a0.x, CUR.sa, lane) * columnDpp(CUR.a0.y, CUR.ta, lane) * CUR.ra;        \
        if (fl & PW_CHERRY_B) xb = columnDpp(CUR.b0.x, CUR.sb, lane) * columnDpp(CUR.b0.y, CUR.tb, lane) * CUR.rb;        \
        v4d ua, ub;                                                                                                       \
        if (tipA) ua = columnDpp(CUR.mA, CUR.sa, lane); else ua = matvecDpp(CUR.mA, xa);                                  \
        if (tipB) ub = columnDpp(CUR.mB, CUR.sb, lane); else ub = matvecDpp(CUR.mB, xb);                                  \
        if (fl & PW_POSTOP) {                          /* an unstored post-order operand, re-evaluated: nothing else happens */ \
            const v4d r = ua * ub * CUR.ra;                                                                               \
            v2d* h = postBase + (size_t)((fl >> PW_DST_SHIFT) & 3u) * holdStride;                                         \
            h[0] = v2d{r.x, r.y}; h[64] = v2d{r.z, r.w};                                                                  \
            DCUR = loadPreDesc(dp + 2);                                                                                   \
            dp += 1;                                                                                                      \
        } else {                                                                                                          \
        const v4d wA = ACC * ub, wB = ACC * ua;                                                                           \
        v4d pa = wA, pb = wB;                                                                                             \
        double ga, gb;                                                                                                    \
        if (tipA) ga = dot4(wA, columnDpp(CUR.dA, CUR.sa, lane));                                                         \
        else { ga = dot4(wA, matvecDpp(CUR.dA, xa)); pa = matvecDppT(CUR.mA, wA) * CUR.ra; }   /* (what the child's edges see: divided by its own scale factor) */ \
        if (tipB) gb = dot4(wB, columnDpp(CUR.dB, CUR.sb, lane));                                                         \
        else { gb = dot4(wB, matvecDpp(CUR.dB, xb)); pb = matvecDppT(CUR.mB, wB) * CUR.rb; }                               \
        DCUR = loadPreDesc(dp + 2);                                                                                       \
        dp += 1;                                                                                                          \
        const double g = waveSumPair(ga, gb, lane);                                                                       \
        if (lane == 31) sums[(size_t)slotA * waves + w] = g;                                                              \
        if (lane == 63) sums[(size_t)slotB * waves + w] = g;                                                              \
        if (contA == PW_CONT_STORE) { if (valid) gptr(reinterpret_cast<v4d*>(stA))[(size_t)c * P + q] = pa; }             \
        else if (contA >= 2u) { v2d* h = holdBase + (size_t)(contA - 2) * holdStride; h[0] = v2d{pa.x, pa.y}; h[64] = v2d{pa.z, pa.w}; }   \
        if (contB == PW_CONT_STORE) { if (valid) gptr(reinterpret_cast<v4d*>(stB))[(size_t)c * P + q] = pb; }             \
        else if (contB >= 2u) { v2d* h = holdBase + (size_t)(contB - 2) * holdStride; h[0] = v2d{pb.x, pb.y}; h[64] = v2d{pb.z, pb.w}; }   \
        if (contA == 1u) ACC = pa;                                                                                        \
        if (contB == 1u) ACC = pb;                                                                                        \
        }                                                                                                                 \
    }
    for (int k = 0; k < nOps; k += 2) {
        PRE_STAGE(A, B, D0, D1)
        PRE_STAGE(B, A, D1, D0)
    }
#undef PRE_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int preWalkWaves(int P, int C) { return ((P + 63) / 64) * C; }

// products[e][c] = M[pairs[2 e]][c] . M[pairs[2 e + 1]][c] (4 x 4, row-major): an edge's branch matrix times its differential
// matrix — what k_preWalk4 applies to the child's post-order partial (PreWalkOp::dA / dB index this array); a pair of -1: zeros
__global__ __launch_bounds__(256) void k_edgeProducts(const double* __restrict__ matrices, const int* __restrict__ pairs, double* __restrict__ products, int C, int n) {
    const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (t >= n * C * 16) return;
    const int k = t & 3, i = (t >> 2) & 3, c = (t >> 4) % C, e = (t >> 4) / C;
    const int m = pairs[2 * e], d = pairs[2 * e + 1];
    double v = 0.0;
    if (m >= 0 && d >= 0) {
        const double* M = matrices + ((size_t)m * C + c) * 16 + 4 * i;
        const double* D = matrices + ((size_t)d * C + c) * 16 + k;
        v = M[0] * D[0] + M[1] * D[4] + M[2] * D[8] + M[3] * D[12];
    }
    products[t] = v;
}
__global__ __launch_bounds__(256) void k_cherryPairs(const double* __restrict__ matrices, const int* __restrict__ pairs, double* __restrict__ out, int C, int n) {
    const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;          // (k, c, l, which)
    if (t >= n * C * 32) return;
    const int which = t & 1, l = (t >> 1) & 15, c = (t >> 5) % C, k = (t >> 5) / C;
    out[t] = matrices[((size_t)pairs[2 * k + which] * C + c) * 16 + l];
}
void launchCherryPairs(hipStream_t stream, const double* matrices, const int* dPairs, double* out, int C, int n) {
    if (n > 0) hipLaunchKernelGGL(k_cherryPairs, dim3((n * C * 32 + 255) / 256), dim3(256), 0, stream, matrices, dPairs, out, C, n);
}
void launchEdgeProducts(hipStream_t stream, const double* matrices, const int* dPairs, double* products, int C, int n) {
    if (n > 0) hipLaunchKernelGGL(k_edgeProducts, dim3((n * C * 16 + 255) / 256), dim3(256), 0, stream, matrices, dPairs, products, C, n);
}

bool launchPreWalk4(hipStream_t stream, const PreWalkOp* dProg, const PreWalkSeg* dSegs, int nSegs, const double* listRootPre,
                    const double* matrices, const double* products, const double* catWeights, const double* patternWeights, double* sums, int P, int C,
                    int holdSlots, bool postSlots) {
    if (nSegs <= 0 || nSegs > 65536 || C < 1 || C > 16 || (size_t)C * P * 32 >= ((size_t)1 << 32)) return false;
    const int slots = holdSlots < 1 ? 1 : holdSlots;                 // (slot 0 doubles as the categories' exchange at the start)
    const size_t lds = (size_t)(slots + (postSlots ? PW_POST_SLOTS : 0)) * C * 128 * sizeof(v2d);
    if (lds > 160 * 1024) return false;
    const dim3 block(64 * C);
    // the segment that starts at the list's root, then (its stores visible at the kernel boundary) all the others
#define PRE_WALK_LAUNCH(T)                                                                                                  \
    { if (!grantDynamicLds(reinterpret_cast<const void*>(k_preWalk4<T>), 160 * 1024)) return false;                        \
      for (int part = 0; part < 2; part++) {                                                                              \
          const int n = part ? nSegs - 1 : 1;                                                                             \
          if (n > 0) hipLaunchKernelGGL(k_preWalk4<T>, dim3((P + 63) / 64, n), block, lds, stream, (const PreWalkOp MI355_CONST*)dProg,   \
                                        (const PreWalkSeg MI355_CONST*)(dSegs + part), listRootPre, matrices, products, catWeights, patternWeights, sums, P, C, part == 0 ? 1 : 0, slots); } }
    if (C <= 4) PRE_WALK_LAUNCH(256) else if (C <= 8) PRE_WALK_LAUNCH(512) else PRE_WALK_LAUNCH(1024)
#undef PRE_WALK_LAUNCH
    return true;
}

// out[e] = the sum of edge e's per-wave sums, in a fixed order
__global__ __launch_bounds__(64) void k_preWalkFinal(const double* __restrict__ sums, int waves, double* __restrict__ out) {
    const double* b = sums + (size_t)blockIdx.x * waves;
    double s = 0.0;
    for (int k = threadIdx.x; k < waves; k += 64) s += b[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
void launchPreWalkFinal(hipStream_t stream, const double* sums, int nSlots, int P, int C, double* out) {
    if (nSlots > 0) hipLaunchKernelGGL(k_preWalkFinal, dim3(nSlots), dim3(64), 0, stream, sums, preWalkWaves(P, C), out);
}

}  // namespace mi355
