// kernels_nuc4.hip — 4-state (nucleotide) pruning, the kernel the headline benchmark runs.
//
// One thread = one pattern, all C rate categories (so the per-pattern rescale max never leaves the thread's
// registers).  Loads/stores of the partials streams are non-temporal 16-byte accesses; a wave covers 2 KiB contiguous
// per category plane.
//
// A child is one of (block-uniform):
//   PARTIALS  32 B per category from the child's buffer, then the branch mat-vec
//   STATES    compact tip: a column of the branch matrix
//   VIRTUAL   the child's whole subtree is a handful of compact tips: recompute its partials in registers from the
//             state bytes (kernels.h, VStep) — no HBM read, and the child's own op wrote nothing either
//
// Where the transition matrices live: all in LDS, staged once per workgroup — as [i][j] row tables when they multiply a
// vector (read at wave-uniform addresses: LDS broadcast) and as [state][i] column tables, with an all-ones row for
// "missing" (state 4), when they are indexed by a lane's own tip state.  (Measured alternative: mat-vec operands
// through scalar loads/SGPRs was 10 % slower at equal occupancy — SMEM latency is exposed at 2-3 waves per SIMD.)
// A virtual child is evaluated one rate category at a time, so its two accumulators cost 16 VGPRs, not 16 * C.
//
// Arithmetic restated from src/dr/oldevomodel/treelikelihood/NucleotideLikelihoodCore.java:54-270 /
// GeneralLikelihoodCore.java:52-203; rescaling AbstractLikelihoodCore.java:406-440 applied unconditionally.
#include "kernels.h"
#include <stdlib.h>

namespace mi355 {

constexpr int NUC_BLOCK = 256;
typedef double v4d __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ v4d ldv4(const double* p) {
    const v4d MI355_GLOBAL* q = gptr(reinterpret_cast<const v4d*>(p));
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT> __device__ __forceinline__ void stv4(double* p, v4d v) {
    v4d MI355_GLOBAL* q = gptr(reinterpret_cast<v4d*>(p));
    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
}

// THE arithmetic of one node, shared by the ordinary path and by the virtual-child programs so both give bitwise the
// same numbers: y = M x (row-major 4x4), and the element-wise product of the two child factors times 1/scale.
__device__ __forceinline__ v4d matvec4(const double* __restrict__ m, v4d x) {
    v4d y;
    y.x = m[0] * x.x + m[1] * x.y + m[2] * x.z + m[3] * x.w;
    y.y = m[4] * x.x + m[5] * x.y + m[6] * x.z + m[7] * x.w;
    y.z = m[8] * x.x + m[9] * x.y + m[10] * x.z + m[11] * x.w;
    y.w = m[12] * x.x + m[13] * x.y + m[14] * x.z + m[15] * x.w;
    return y;
}
__device__ __forceinline__ v4d colvec4(const double* __restrict__ col5x4, int s) {   // [state][i], row 4 = ones
    const double* c = col5x4 + 4 * s;
    v4d y; y.x = c[0]; y.y = c[1]; y.z = c[2]; y.w = c[3];
    return y;
}
__device__ __forceinline__ v4d combine4(v4d a, v4d b, double inv) {
    v4d y;
    y.x = (a.x * b.x) * inv; y.y = (a.y * b.y) * inv; y.z = (a.z * b.z) * inv; y.w = (a.w * b.w) * inv;
    return y;
}

template <int C>
struct NucLds {
    double row[2][C][16];                          // [child][c][i*4+j]      the op's two branch matrices
    double col[2][C][20];                          // [child][c][state*4+i]  same as column tables (STATES children)
    double prog[2][VIRT_EMIT_STEPS][2][C][20];     // per child, step, operand: row table (16 used) or column table
};

enum { CH_PARTIALS = 0, CH_STATES = 1, CH_VIRTUAL = 2 };

template <int C>
__device__ __forceinline__ void stageRow(double (*dst)[20], const double* __restrict__ M) {
    for (int t = threadIdx.x; t < C * 16; t += NUC_BLOCK) dst[t >> 4][t & 15] = M[t];
}
template <int C>
__device__ __forceinline__ void stageCol(double (*dst)[20], const double* __restrict__ M) {
    for (int t = threadIdx.x; t < C * 16; t += NUC_BLOCK) { const int e = t & 15; dst[t >> 4][(e & 3) * 4 + (e >> 2)] = M[t]; }
    for (int t = threadIdx.x; t < C * 4; t += NUC_BLOCK) dst[t >> 2][16 + (t & 3)] = 1.0;
}

// What a virtual child needs from memory is requested before the staging barrier: the 1/scale of each step (two v4d,
// VIRT_MAX_STEPS == 8) and the tip-state bytes packed into two 64-bit words (byte s of `pa` = step s tipA state,
// byte s of `pb` = its tipB state).
static_assert(VIRT_MAX_STEPS == 8, "1/scale values are two v4d, states are packed 8 + 8 bytes");

struct VirtPark { v4d invLo, invHi; unsigned long long pa, pb; };

__device__ __forceinline__ void virtIssue(const VStep* __restrict__ prog, int p, VirtPark& k) {
    k.invLo = v4d{1.0, 1.0, 1.0, 1.0}; k.invHi = k.invLo;
    k.pa = 0x0404040404040404ull; k.pb = k.pa;
#pragma unroll
    for (int s = 0; s < VIRT_EMIT_STEPS; s++) {
        const int type = prog[s].type;
        if (type == VS_END) continue;
        if (type == VS_CHERRY_A || type == VS_CHERRY_B)
            k.pa = (k.pa & ~(0xffull << (8 * s))) | ((unsigned long long)gptr(prog[s].tipA)[p] << (8 * s));
        if (type != VS_JOIN)
            k.pb = (k.pb & ~(0xffull << (8 * s))) | ((unsigned long long)gptr(prog[s].tipB)[p] << (8 * s));
        if (prog[s].scale) { const double v = 1.0 / gptr(prog[s].scale)[p]; if (s < 4) k.invLo[s] = v; else k.invHi[s - 4] = v; }
    }
}

template <int C>
__device__ __forceinline__ void virtStage(NucLds<C>& L, int child, const VStep* __restrict__ prog, const double* __restrict__ matrices) {
#pragma unroll
    for (int s = 0; s < VIRT_EMIT_STEPS; s++) {
        const int type = prog[s].type;
        if (type == VS_END) continue;
        const double* MA = matrices + (size_t)prog[s].matA * (C * 16);
        const double* MB = matrices + (size_t)prog[s].matB * (C * 16);
        if (type == VS_CHERRY_A || type == VS_CHERRY_B) stageCol<C>(L.prog[child][s][0], MA); else stageRow<C>(L.prog[child][s][0], MA);
        if (type == VS_JOIN) stageRow<C>(L.prog[child][s][1], MB); else stageCol<C>(L.prog[child][s][1], MB);
    }
}

// run the program for ONE rate category; returns the child's partials for that category
template <int C>
__device__ __forceinline__ v4d virtEval(const NucLds<C>& L, int child, const VStep* __restrict__ prog, const VirtPark& k, int c) {
    v4d A = v4d{1.0, 1.0, 1.0, 1.0}, B = A;
    // Up to 4 categories: fully unrolled over the steps the host may emit (VIRT_EMIT_STEPS) — the compiler then
    // interleaves the LDS reads of different steps and categories (a runtime loop is ~20 % slower: every step waits for
    // its own LDS latency).  More categories: runtime loop, or the code (and the compile time) grows ~C * steps * 10.
#pragma unroll (C <= 4 ? VIRT_EMIT_STEPS : 1)
    for (int s = 0; s < VIRT_EMIT_STEPS; s++) {
        const int type = prog[s].type;
        if (type == VS_END) break;
        const double* t0 = L.prog[child][s][0][c];
        const double* t1 = L.prog[child][s][1][c];
        const int sa = (int)((k.pa >> (8 * s)) & 0xff), sb = (int)((k.pb >> (8 * s)) & 0xff);
        const double lo = (s & 2) ? ((s & 1) ? k.invLo.w : k.invLo.z) : ((s & 1) ? k.invLo.y : k.invLo.x);
        const double hi = (s & 2) ? ((s & 1) ? k.invHi.w : k.invHi.z) : ((s & 1) ? k.invHi.y : k.invHi.x);
        const double iv = (s & 4) ? hi : lo;
        if (type == VS_CHERRY_A)      A = combine4(colvec4(t0, sa), colvec4(t1, sb), iv);
        else if (type == VS_CHERRY_B) B = combine4(colvec4(t0, sa), colvec4(t1, sb), iv);
        else if (type == VS_EXTEND_A) A = combine4(matvec4(t0, A), colvec4(t1, sb), iv);
        else if (type == VS_EXTEND_B) B = combine4(matvec4(t0, B), colvec4(t1, sb), iv);
        else                          A = combine4(matvec4(t0, A), matvec4(t1, B), iv);     // VS_JOIN
    }
    return A;
}

// MINW = waves per SIMD the register allocator must leave room for
template <int C, int NT, int MINW>
__global__ __launch_bounds__(NUC_BLOCK, MINW) void k_prune4(const OpDesc* __restrict__ ops, const double* __restrict__ matrices, int P) {
    __shared__ NucLds<C> L;
    const OpDesc& op = ops[blockIdx.y];
    const int pEnd = op.pEnd;
    const int p0 = op.pStart + blockIdx.x * NUC_BLOCK;
    if (p0 >= pEnd) return;
    const int kindBits = op.kind;
    const int k1 = (kindBits & KIND_STATES1) ? CH_STATES : (kindBits & KIND_VIRT1) ? CH_VIRTUAL : CH_PARTIALS;
    const int k2 = (kindBits & KIND_STATES2) ? CH_STATES : (kindBits & KIND_VIRT2) ? CH_VIRTUAL : CH_PARTIALS;
    const int p = p0 + threadIdx.x;
    const bool valid = p < pEnd;

    // ---- everything that comes from memory is requested first ...
    // x1/x2: the child's partials (PARTIALS); w1/w2: a compact tip's state; v1/v2: a virtual child's parked inputs
    v4d x1[C], x2[C];
    int w1 = 4, w2 = 4;
    VirtPark v1, v2;
    double invRead = 1.0;
    if (valid) {
        if (k1 == CH_PARTIALS) {
            const double* x = reinterpret_cast<const double*>(op.child1);
#pragma unroll
            for (int c = 0; c < C; c++) x1[c] = ldv4<(NT & 1) != 0>(x + ((size_t)c * P + p) * 4);
        } else if (k1 == CH_STATES) w1 = gptr(reinterpret_cast<const uint8_t*>(op.child1))[p];
        else virtIssue(op.prog[0], p, v1);
        if (k2 == CH_PARTIALS) {
            const double* x = reinterpret_cast<const double*>(op.child2);
#pragma unroll
            for (int c = 0; c < C; c++) x2[c] = ldv4<(NT & 1) != 0>(x + ((size_t)c * P + p) * 4);
        } else if (k2 == CH_STATES) w2 = gptr(reinterpret_cast<const uint8_t*>(op.child2))[p];
        else virtIssue(op.prog[1], p, v2);
        if (!op.scaleWrite && op.scaleRead) invRead = 1.0 / gptr(op.scaleRead)[p];
    }
    // ---- ... then the column tables are staged while those requests are in flight
    const double* G1 = matrices + (size_t)op.mat1 * (C * 16);
    const double* G2 = matrices + (size_t)op.mat2 * (C * 16);
    for (int t = threadIdx.x; t < C * 16; t += NUC_BLOCK) { L.row[0][t >> 4][t & 15] = G1[t]; L.row[1][t >> 4][t & 15] = G2[t]; }
    if (k1 == CH_STATES) stageCol<C>(L.col[0], G1);
    if (k2 == CH_STATES) stageCol<C>(L.col[1], G2);
    if (k1 == CH_VIRTUAL) virtStage<C>(L, 0, op.prog[0], matrices);
    if (k2 == CH_VIRTUAL) virtStage<C>(L, 1, op.prog[1], matrices);
    __syncthreads();
    if (!valid) return;

    // child 1 -> its factor (x1 is dead afterwards), then child 2; virtual children one category at a time
    v4d a[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        if (k1 == CH_STATES) a[c] = colvec4(L.col[0][c], w1);
        else a[c] = matvec4(L.row[0][c], k1 == CH_VIRTUAL ? virtEval<C>(L, 0, op.prog[0], v1, c) : x1[c]);
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
        v4d f2;
        if (k2 == CH_STATES) f2 = colvec4(L.col[1][c], w2);
        else f2 = matvec4(L.row[1][c], k2 == CH_VIRTUAL ? virtEval<C>(L, 1, op.prog[1], v2, c) : x2[c]);
        a[c] = combine4(a[c], f2, 1.0);
    }
    if (op.scaleWrite) {
        double m = 0.0;
#pragma unroll
        for (int c = 0; c < C; c++) m = fmax(fmax(fmax(m, a[c].x), fmax(a[c].y, a[c].z)), a[c].w);
        if (!(m > 0.0)) m = 1.0;
        gptr(op.scaleWrite)[p] = m;
        const double inv = 1.0 / m;
#pragma unroll
        for (int c = 0; c < C; c++) a[c] = a[c] * inv;
    } else if (op.scaleRead) {
#pragma unroll
        for (int c = 0; c < C; c++) a[c] = a[c] * invRead;
    }
    if (kindBits & KIND_NO_STORE) return;     // virtual node in write-mode rescaling: only its scale factors are kept
#pragma unroll
    for (int c = 0; c < C; c++) stv4<(NT & 2) != 0>(op.dest + ((size_t)c * P + p) * 4, a[c]);
}

bool launchPruneLevelNuc4(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int C, int maxRange) {
    if (C > 8) return false;
    dim3 grid((maxRange + NUC_BLOCK - 1) / NUC_BLOCK, nOps), block(NUC_BLOCK);
    // BEAGLE_MI355_NT: bit 0 = non-temporal loads, bit 1 = non-temporal stores (default 3: +3 % on config A)
    static const int nt = getenv("BEAGLE_MI355_NT") ? (atoi(getenv("BEAGLE_MI355_NT")) & 3) : 3;
    // two variants per category count only (compile time): non-temporal streams on (default) or off
    // 4 waves per SIMD (128 VGPRs, 44 B/lane of scratch) beats 3 waves without spills by ~11 % on config A
    static const int minw = getenv("BEAGLE_MI355_WAVES") ? atoi(getenv("BEAGLE_MI355_WAVES")) : 4;
#define LAUNCH_NUC(CC)                                                                                          \
    if (nt == 0)        hipLaunchKernelGGL((k_prune4<CC, 0, 3>), grid, block, 0, stream, dOps, matrices, P);    \
    else if (minw >= 4) hipLaunchKernelGGL((k_prune4<CC, 3, 4>), grid, block, 0, stream, dOps, matrices, P);    \
    else                hipLaunchKernelGGL((k_prune4<CC, 3, 3>), grid, block, 0, stream, dOps, matrices, P)
    switch (C) {
        case 1: LAUNCH_NUC(1); break;
        case 2: LAUNCH_NUC(2); break;
        case 3: LAUNCH_NUC(3); break;
        case 4: LAUNCH_NUC(4); break;
        case 5: LAUNCH_NUC(5); break;
        case 6: LAUNCH_NUC(6); break;
        case 7: LAUNCH_NUC(7); break;
        default: LAUNCH_NUC(8); break;
    }
#undef LAUNCH_NUC
    return true;
}

}  // namespace mi355
