// kernels_walk4.hip — 4-state (nucleotide) pruning as ONE launch per operation list: the "pattern walk".
//
// A site pattern never needs another pattern's data, so a thread that owns (pattern p, rate category c) can execute a
// whole dependency-ordered operation list by itself, one operation after the other, with no grid-wide synchronisation:
// what it reads was either there before the launch or written by the same thread earlier in the same launch (program
// order makes a thread's own stores visible to its later loads).  The host (planner.cpp) turns an updatePartials list
// into a post-order program of micro-operations; the result of a micro-operation stays in the thread's registers (ACC)
// or in one of two LDS hold slots (a value that has to wait for its sibling's subtree), so a child that was computed by
// the previous micro-operation is not read back from HBM, and a node whose subtree is a few compact tips ("virtual"
// buffer) is never written at all.
//
// Mapping:  workgroup = 64 consecutive patterns x all C categories; wave w = category w; lane l = pattern p0 + l.
//           A wave's loads/stores of a partials buffer ([C][P][4] doubles) are 64 x 32 B = 2 KiB contiguous.
//           All C*P/64 waves of a 1e5-pattern alignment are resident at once (6.1 waves per SIMD at C = 4; the kernel is
//           held to 72 VGPRs = 7 waves per SIMD for that reason: a wave walks the whole list, so a second round of
//           workgroups would double the time).
//
// What bounds a wave is LATENCY: it executes ~T dependent micro-operations.  So the loop is software-pipelined by hand:
// while micro-operation k computes, everything k+1 needs from memory is already in flight — its child partials (32 B per
// lane), tip-state bytes, reciprocal scale factor and BOTH branch matrices.  The compiler cannot express that (its
// s_waitcnt insertion has to assume the worst path of the kind-dependent branches and drains the queue every iteration:
// measured 64 % of wave cycles parked, profiles/r02_*), therefore every vector-memory instruction of the loop is inline
// assembly, and the one wait per stage is EXACT: "s_waitcnt vmcnt(N)" with N = the number of vector-memory instructions
// issued after the loads of micro-operation k (the stores of k-1 and the loads of k+1), which the host knows when it
// builds the program and passes in the descriptor; the kernel jumps into a table of s_waitcnt instructions.
// Loads a micro-operation does not need are BRANCHED around, not masked: on this chip a vector-memory instruction with 8
// or 16 bytes per lane occupies the CU's address unit for ~16 cycles whatever its EXEC mask or coalescing, a byte load
// for ~4 (tools/vmem_rate_probe.hip) — with 24 waves per CU that unit, not HBM, is the first thing to saturate.
//
// Branch matrices are wave-uniform (one category per wave).  A matrix lives in ONE 64-bit VGPR, lane l holding entry
// l & 15 (one 8-byte load per lane, 128 B per wave), and the 4x4 mat-vec is 16 x v_fmac_f64_dpp row_newbcast:n (DPP64:
// every lane multiplies by lane n of its 16-lane row) — no SGPRs, no LDS, no scalar loads, full fp64 rate
// (tools/walk_probe.hip).  A compact tip child picks column `state` of the same register with ds_bpermute (crossbar
// only, no LDS storage).
// Rescaling in write mode needs the per-pattern maximum over all categories: the C waves exchange their maxima
// through 2 x C x 64 doubles of LDS and one barrier (double-buffered); read mode multiplies by the stored reciprocal.
//
// Arithmetic restated from src/dr/oldevomodel/treelikelihood/NucleotideLikelihoodCore.java:54-270 /
// GeneralLikelihoodCore.java:52-203; rescaling AbstractLikelihoodCore.java:406-440 applied unconditionally.
#include "kernels.h"

namespace mi355 {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
#define MI355_CONST __attribute__((address_space(4)))

// what a micro-operation needs from memory; two of these ping-pong (one being consumed, one in flight)
struct Fetched {
    v2d xa, xb;          // first child's partials (WF_X)
    unsigned s1, s2;     // tip states of the two children (WF_T1 / WF_T2)
    double inv;          // reciprocal scale factor (WF_INV)
    double sp1, sp2;     // the two branch matrices, lane l = entry l & 15
};

struct Desc {            // a WalkOp in SGPRs
    u64 src1, src2, store, scale, m1, m2;
    unsigned flags;
};
__device__ __forceinline__ Desc unpack(const u32x16 d) {
    Desc r;
    r.src1 = ((u64)d.s1 << 32) | d.s0; r.src2 = ((u64)d.s3 << 32) | d.s2; r.store = ((u64)d.s5 << 32) | d.s4;
    r.scale = ((u64)d.s7 << 32) | d.s6; r.m1 = ((u64)d.s9 << 32) | d.s8; r.m2 = ((u64)d.sb << 32) | d.sa;
    r.flags = d.sc;
    return r;
}

// issue the loads of one micro-operation: only the groups it needs (WF_* bits of the flags), then the two matrices
__device__ __forceinline__ void fetchIssue(Fetched& f, const Desc& d, unsigned oPart, unsigned oTip, unsigned oScale, unsigned oMat) {
    asm volatile(
        "s_bitcmp1_b32 %[fl], 0\n\t"
        "s_cbranch_scc0 .Lfx%=\n\t"
        "global_load_dwordx4 %[xa], %[oP], %[src1]\n\t"
        "global_load_dwordx4 %[xb], %[oP], %[src1] offset:16\n"
        ".Lfx%=:\n\t"
        "s_bitcmp1_b32 %[fl], 1\n\t"
        "s_cbranch_scc0 .Lft1%=\n\t"
        "global_load_ubyte %[s1], %[oT], %[src1]\n"
        ".Lft1%=:\n\t"
        "s_bitcmp1_b32 %[fl], 2\n\t"
        "s_cbranch_scc0 .Lft2%=\n\t"
        "global_load_ubyte %[s2], %[oT], %[src2]\n"
        ".Lft2%=:\n\t"
        "s_bitcmp1_b32 %[fl], 3\n\t"
        "s_cbranch_scc0 .Lfi%=\n\t"
        "global_load_dwordx2 %[inv], %[oS], %[scale]\n"
        ".Lfi%=:\n\t"
        "global_load_dwordx2 %[sp1], %[oM], %[m1]\n\t"
        "global_load_dwordx2 %[sp2], %[oM], %[m2]"
        : [xa] "=&v"(f.xa), [xb] "=&v"(f.xb), [s1] "=&v"(f.s1), [s2] "=&v"(f.s2), [inv] "=&v"(f.inv), [sp1] "=&v"(f.sp1), [sp2] "=&v"(f.sp2)
        : [fl] "s"(d.flags), [oP] "v"(oPart), [oT] "v"(oTip), [oS] "v"(oScale), [oM] "v"(oMat),
          [src1] "s"(d.src1), [src2] "s"(d.src2), [scale] "s"(d.scale), [m1] "s"(d.m1), [m2] "s"(d.m2)
        : "memory", "scc");
}
// The loads of `f` have landed once at most N younger vector-memory instructions are outstanding; jump = 8 N + 12 is the
// byte offset of "s_waitcnt vmcnt(N)" in the table below, counted from the instruction after s_getpc_b64 (every
// instruction here is 4 bytes; an entry is a wait and a branch).
__device__ __forceinline__ void fetchWait(Fetched& f, unsigned jump) {
    asm volatile(
        "s_getpc_b64 s[80:81]\n\t"
        "s_add_u32 s80, s80, %[jump]\n\t"
        "s_addc_u32 s81, s81, 0\n\t"
        "s_setpc_b64 s[80:81]\n\t"
        "s_waitcnt vmcnt(0)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(1)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(2)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(3)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(4)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(5)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(6)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(7)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(8)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(9)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(10)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(11)\n\ts_branch .Lwd%=\n\t"
        "s_waitcnt vmcnt(12)\n"
        ".Lwd%=:"
        : "+v"(f.xa), "+v"(f.xb), "+v"(f.s1), "+v"(f.s2), "+v"(f.inv), "+v"(f.sp1), "+v"(f.sp2)
        : [jump] "s"(jump) : "memory", "scc", "s80", "s81");
}
// the stores of one micro-operation (`mask` = the lanes that really store; nothing is issued when it has no destination)
__device__ __forceinline__ void storeIssue(const v4d r, unsigned flags, u64 mask, unsigned oPart, u64 base) {
    const v2d lo = v2d{r.x, r.y}, hi = v2d{r.z, r.w};
    asm volatile(
        "s_bitcmp1_b32 %[fl], 4\n\t"
        "s_cbranch_scc0 .Lst%=\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "global_store_dwordx4 %[oP], %[lo], %[base] nt\n\t"
        "global_store_dwordx4 %[oP], %[hi], %[base] offset:16 nt\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_nop 0\n"
        ".Lst%=:"
        : : [fl] "s"(flags), [m] "s"(mask), [oP] "v"(oPart), [lo] "v"(lo), [hi] "v"(hi), [base] "s"(base) : "memory", "scc");
}

// y = M x with M spread over the lanes of `sp` (lane l = entry l & 15, row-major).  The rounding sequence is fixed:
// y_i = fma(m_i3, x3, fma(m_i2, x2, fma(m_i1, x1, fma(m_i0, x0, 0)))) — the element order NucleotideLikelihoodCore uses.
__device__ __forceinline__ v4d matvecDpp(const double sp, const v4d x) {
    double y0, y1, y2, y3;
    const double x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
#define FM(Y, N, X) "v_fmac_f64_dpp %[" #Y "], %[sp], %[" #X "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile(
        "v_mov_b64 %[y0], 0\n\tv_mov_b64 %[y1], 0\n\tv_mov_b64 %[y2], 0\n\tv_mov_b64 %[y3], 0\n\t"
        FM(y0, 0, x0) FM(y1, 4, x0) FM(y2, 8, x0) FM(y3, 12, x0)
        FM(y0, 1, x1) FM(y1, 5, x1) FM(y2, 9, x1) FM(y3, 13, x1)
        FM(y0, 2, x2) FM(y1, 6, x2) FM(y2, 10, x2) FM(y3, 14, x2)
        FM(y0, 3, x3) FM(y1, 7, x3) FM(y2, 11, x3) FM(y3, 15, x3)
        "s_nop 0"
        : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3)
        : [sp] "v"(sp), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3));
#undef FM
    return v4d{y0, y1, y2, y3};
}

__device__ __forceinline__ double bperm(double v, int byteAddr) {
    const int lo = __builtin_amdgcn_ds_bpermute(byteAddr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byteAddr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// column `s` of the spread matrix: y[i] = M[i][s] = lane 4i + s, or 1 for a missing state (s >= 4)
__device__ __forceinline__ v4d column4(double sp, unsigned s) {
    const int base = (int)(s & 3u) * 4;
    const bool known = s < 4u;
    v4d y;
    y.x = bperm(sp, base); y.y = bperm(sp, base + 16); y.z = bperm(sp, base + 32); y.w = bperm(sp, base + 48);
    y.x = known ? y.x : 1.0; y.y = known ? y.y : 1.0; y.z = known ? y.z : 1.0; y.w = known ? y.w : 1.0;
    return y;
}

// MAXT = 64 * C threads; MINW = waves per SIMD the register allocation must allow (see the file header)
template <int MAXT, int MINW>
__global__ __launch_bounds__(MAXT, MINW) void k_walk4(const u32x16 MI355_CONST* __restrict__ prog, const WalkSeg MI355_CONST* __restrict__ segs,
                                                      int P, int C, long recipOff) {
    extern __shared__ v2d lds[];                      // hold[2][C][2][64] (v2d), then exch[2][C][64] (double)
    const WalkSeg MI355_CONST& sg = segs[blockIdx.y];
    const int progStart = sg.progStart, progCount = sg.progCount, pStart = sg.pStart, pEnd = sg.pEnd;
    const int p0 = pStart + (int)blockIdx.x * 64;
    if (p0 >= pEnd) return;                           // the whole workgroup
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool valid = p0 + lane < pEnd;
    const int p = valid ? p0 + lane : pEnd - 1;       // lanes past the end recompute the last pattern and store nothing
    const u64 validMask = __ballot(valid);
    // loop-invariant 32-bit byte offsets: every address of the loop is (64-bit SGPR base from the descriptor) + one of these
    const unsigned oPart = (unsigned)(((size_t)c * P + p) * 32), oTip = (unsigned)p, oScale = (unsigned)p * 8u;
    const unsigned oMat = (unsigned)(c * 128 + (lane & 15) * 8);
    v2d* holdBase = lds + (size_t)c * 128 + lane;     // + slot * C * 128 (+ 64 for the second half)
    double* exch = reinterpret_cast<double*>(lds + (size_t)2 * C * 128);
    int buf = 0;

    v4d ACC = v4d{1.0, 1.0, 1.0, 1.0};
    const u32x16 MI355_CONST* dp = prog + progStart;  // the host pads every segment: progCount is even and two more
    u32x16 D0 = dp[0], D1 = dp[1];                     // descriptors (no-ops) follow it, so k + 2 is always readable
    Fetched A, B;
    fetchIssue(A, unpack(D0), oPart, oTip, oScale, oMat);

    // one micro-operation: CUR holds its operands (issued one stage ago), NXT receives those of the following one.
    // The few descriptor fields the compute stage needs are moved out of DCUR first (explicit s_mov: the register
    // allocator then lets descriptor k + 2 land in DCUR's own registers instead of rotating 16 SGPRs per stage).
#define WALK_STAGE(CUR, NXT, DCUR, DNXT)                                                                                  \
    {                                                                                                                     \
        unsigned fl; u64 dStore, dScale, dSrc2;                                                                           \
        asm volatile("s_mov_b32 %0, %4\n\ts_mov_b64 %1, %5\n\ts_mov_b64 %2, %6\n\ts_mov_b64 %3, %7"                     \
                     : "=&s"(fl), "=&s"(dStore), "=&s"(dScale), "=&s"(dSrc2)                                              \
                     : "s"(DCUR.sc), "s"(((u64)DCUR.s5 << 32) | DCUR.s4), "s"(((u64)DCUR.s7 << 32) | DCUR.s6),            \
                       "s"(((u64)DCUR.s3 << 32) | DCUR.s2));                                                              \
        DCUR = dp[2];                                  /* descriptor k + 2 (used two stages on) */                        \
        fetchIssue(NXT, unpack(DNXT), oPart, oTip, oScale, oMat);                                                         \
        dp += 1;                                                                                                          \
        const int shape = (fl >> 5) & 63, hold = (fl >> 11) & 3, smode = (fl >> 13) & 3;                                  \
        fetchWait(CUR, (fl >> 16) & 0xffu);            /* 8 N + 12, N = younger loads (kernels.h walkWaitJump) */         \
        v4d f1, f2;                                                                                                       \
        switch (shape) {                                                                                                  \
            case WK_TIPS | (WK_TIPS << 3): f1 = column4(CUR.sp1, CUR.s1); f2 = column4(CUR.sp2, CUR.s2); break;           \
            case WK_TIPS | (WK_ACC << 3):  f1 = column4(CUR.sp1, CUR.s1); f2 = matvecDpp(CUR.sp2, ACC); break;            \
            case WK_MEM | (WK_ACC << 3):   f1 = matvecDpp(CUR.sp1, v4d{CUR.xa.x, CUR.xa.y, CUR.xb.x, CUR.xb.y});          \
                                           f2 = matvecDpp(CUR.sp2, ACC); break;                                           \
            case WK_MEM | (WK_TIPS << 3):  f1 = matvecDpp(CUR.sp1, v4d{CUR.xa.x, CUR.xa.y, CUR.xb.x, CUR.xb.y});          \
                                           f2 = column4(CUR.sp2, CUR.s2); break;                                          \
            case WK_H0 | (WK_ACC << 3): case WK_H1 | (WK_ACC << 3): {        /* the thread's own hold slot */             \
                const v2d* h = holdBase + (size_t)((shape & 7) - WK_H0) * C * 128;                                        \
                const v2d lo = h[0], hi = h[64];                                                                          \
                f1 = matvecDpp(CUR.sp1, v4d{lo.x, lo.y, hi.x, hi.y}); f2 = matvecDpp(CUR.sp2, ACC); break; }              \
            default: {                                 /* both children in memory (rare): the second one is not prefetched */ \
                v2d ya, yb;                                                                                               \
                asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16\n\ts_waitcnt vmcnt(0)" \
                             : "=&v"(ya), "=&v"(yb) : "v"(oPart), "s"(dSrc2) : "memory");                                 \
                f1 = matvecDpp(CUR.sp1, v4d{CUR.xa.x, CUR.xa.y, CUR.xb.x, CUR.xb.y});                                     \
                f2 = matvecDpp(CUR.sp2, v4d{ya.x, ya.y, yb.x, yb.y}); break; }                                            \
        }                                                                                                                 \
        v4d r = f1 * f2;                                                                                                  \
        if (smode == WS_READ) r = r * CUR.inv;                                                                            \
        else if (smode == WS_WRITE) {                                                                                     \
            double m = fmax(fmax(fmax(0.0, r.x), fmax(r.y, r.z)), r.w);                                                   \
            double* e = exch + (size_t)buf * C * 64;                                                                      \
            e[c * 64 + lane] = m;                                                                                         \
            __syncthreads();                                                                                              \
            m = 0.0;                                                                                                      \
            for (int cc = 0; cc < C; cc++) m = fmax(m, e[cc * 64 + lane]);                                                \
            buf ^= 1;                                                                                                     \
            if (!(m > 0.0)) m = 1.0;                                                                                      \
            const double im = 1.0 / m;                                                                                    \
            r = r * im;                                                                                                   \
            const u64 wm = c == 0 ? validMask : 0ull;  /* category 0 stores the factor and its reciprocal; then drain */  \
            asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dwordx2 %1, %3, %5\n\tglobal_store_dwordx2 %2, %4, %5\n\t"   \
                         "s_mov_b64 exec, -1\n\ts_waitcnt vmcnt(0)"                                                       \
                         : : "s"(wm), "v"(oScale), "v"(oScale + (unsigned)recipOff * 8u), "v"(m), "v"(im), "s"(dScale) : "memory"); \
        }                                                                                                                 \
        storeIssue(r, fl, validMask, oPart, dStore);                                                                      \
        if (hold) {                                    /* this value waits for its sibling's subtree */                   \
            v2d* h = holdBase + (size_t)(hold - 1) * C * 128;                                                             \
            h[0] = v2d{r.x, r.y}; h[64] = v2d{r.z, r.w};                                                                  \
        }                                                                                                                 \
        ACC = r;                                                                                                          \
    }

    for (int k = 0; k < progCount; k += 2) {
        WALK_STAGE(A, B, D0, D1)
        WALK_STAGE(B, A, D1, D0)
    }
#undef WALK_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void launchWalk4(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, int P, int C, long recipOff) {
    if (nSegs <= 0 || maxRange <= 0) return;
    const dim3 grid((maxRange + 63) / 64, nSegs), block(64 * C);
    const size_t lds = (size_t)2 * C * 128 * sizeof(v2d) + (size_t)2 * C * 64 * sizeof(double);
    const u32x16 MI355_CONST* prog = (const u32x16 MI355_CONST*)dProg;
    const WalkSeg MI355_CONST* segs = (const WalkSeg MI355_CONST*)dSegs;
    if (C <= 4) hipLaunchKernelGGL((k_walk4<256, 7>), grid, block, lds, stream, prog, segs, P, C, recipOff);
    else if (C <= 8) hipLaunchKernelGGL((k_walk4<512, 6>), grid, block, lds, stream, prog, segs, P, C, recipOff);
    else hipLaunchKernelGGL((k_walk4<1024, 4>), grid, block, lds, stream, prog, segs, P, C, recipOff);
}

}  // namespace mi355
