// kernels_walk4.hip — 4-state (nucleotide) pruning as ONE launch per operation list: the "pattern walk".
//
// A site pattern never needs another pattern's data, so a thread that owns (pattern p, rate category c) can execute a
// whole dependency-ordered operation list by itself, one operation after the other, with no grid-wide synchronisation:
// what it reads was either there before the launch or written by the same thread earlier in the same launch (program
// order makes a thread's own stores visible to its later loads).  The host (planner.cpp) turns an updatePartials list
// into a post-order program of micro-operations; the result of a micro-operation stays in the thread's registers (ACC)
// or in one of three hold slots (a value that has to wait for its sibling's subtree), so a child that was computed by
// the previous micro-operation is not read back from HBM, and a node whose subtree is a few compact tips ("virtual"
// buffer) is never written at all.
//
// Two kernels execute such programs (bit-identical results, tests/test_gpu_walk_kernels.py):
//   k_walk4_fast  the main loop as ONE block of generated gfx950 assembly (tools/gen_walk4_fast.py, walk4_fast_loop.inc; the
//                 design notes live in that generator) — every launch whose segments start at a multiple of 128 patterns
//                 (which the engine arranges: partitions are padded internally);
//   k_walk4       the C++ kernel below: the reference implementation of the same walk (BEAGLE_MI355_NO_FAST_WALK=1, tests).
//
// Common to both.  Workgroup = 128 consecutive patterns x all C categories; wave w = category w; a lane owns TWO patterns (here
// p0 + l and p0 + 64 + l), so what a micro-operation costs per wave whatever the lanes do — scalar and branch instructions,
// the matrix table — is shared by two patterns.  A wave walks its whole program in order, so the loop is software-pipelined
// by hand: while micro-operation k computes, everything k+1 needs from memory is in flight.  The compiler cannot express
// that (its s_waitcnt insertion assumes the worst path of the kind-dependent branches and drains the queue every
// iteration), therefore every vector-memory instruction of the loop is inline assembly and the one wait per stage is EXACT:
// "s_waitcnt vmcnt(N)", N = the loads of micro-operation k+1 (issued after those of k; loads return in issue order — with
// BEAGLE_MI355_STRICT_WAITS=0 also the stores of k-1, see engine_walk.cpp runPlan) — which the host knows when it
// builds the program and passes in the descriptor; this kernel jumps into a table of s_waitcnt instructions.  Loads a
// micro-operation does not need are BRANCHED around, not masked: a vector-memory instruction with 8 or 16 bytes per lane
// occupies the CU's address unit for ~16 cycles whatever its EXEC mask or coalescing (tools/vmem_rate_probe.hip).
//
// Branch matrices are wave-uniform (one category per wave).  k_gatherMatrices lays the two matrices of every
// micro-operation out as a 320-byte table (5 columns x 4 doubles each, column 4 = ones for a missing state) in program
// order; the fetch stage copies the next table into wave-private LDS with ONE LDS-DMA instruction (lanes 0..19, no
// registers: tools/glds_probe.hip).  A compact tip child's contribution is column `state` of the table (two
// ds_read_b128, no select); an internal child's 4x4 mat-vec is 16 x v_fmac_f64_dpp row_newbcast:n on the matrix spread
// over the lanes of ONE 64-bit VGPR (DPP64: every lane multiplies by lane n of its 16-lane row) — no SGPRs, no LDS
// operands, full fp64 rate (tools/walk_probe.hip).
// Tip states and reciprocal scale factors are stored pair-interleaved (kernels.h walkPairIndex) so that the assembly
// loop gets a lane's pair with one load; this kernel's lanes own other pairs and address the same layout per pattern.
//
// Arithmetic restated from src/dr/oldevomodel/treelikelihood/NucleotideLikelihoodCore.java:54-270 /
// GeneralLikelihoodCore.java:52-203; rescaling AbstractLikelihoodCore.java:406-440 applied unconditionally.
#include "kernels.h"
#include "root_site4.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>

namespace mi355 {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define MI355_CONST __attribute__((address_space(4)))

// what a micro-operation needs from memory, for the lane's two patterns (a, b); two of these ping-pong (one being
// consumed, one in flight)
struct Fetched {
    v2d xa0, xa1, xb0, xb1;   // first child's partials (WF_X) or the hold slot it comes from: pattern a, pattern b
    unsigned s1a, s1b, s2a, s2b;   // tip states of the two children (WF_T1 / WF_T2), pattern a / b
    double inva, invb;        // reciprocal scale factors of the pair (WF_INV)
};                            // (the two branch matrices go straight to LDS: fetchIssue)

struct Desc {                 // a WalkOp in SGPRs (11 dwords; the matrices come through the stream, not the descriptor)
    u64 src1, src2, store, scale, scaleW;
    unsigned flags;
};
// scalar loads of exactly the dwords in use: an unused lane of a wider load would be a register the allocator hands out
// while the load is still pending
__device__ __forceinline__ Desc loadDesc(const unsigned MI355_CONST* p) {
    const u32x8 a = *reinterpret_cast<const u32x8 MI355_CONST*>(p);
    Desc r;
    r.src1 = ((u64)a.s1 << 32) | a.s0; r.src2 = ((u64)a.s3 << 32) | a.s2; r.store = ((u64)a.s5 << 32) | a.s4;
    r.scale = ((u64)a.s7 << 32) | a.s6;
    r.flags = p[8];                                  // (kernels.h WalkOp: flags at byte 32, scaleW at 40)
    r.scaleW = ((u64)p[11] << 32) | p[10];
    return r;
}

// Loop-invariant 32-bit byte offsets of the lane.  Tip states and reciprocal scale factors are stored PAIR-INTERLEAVED
// (kernels.h walkPairIndex) for the assembly loop k_walk4_fast, whose lanes own other pattern pairs than this kernel's;
// here tipA / tipB / scaleA / scaleB are simply the positions of the lane's two patterns in that layout.
struct LaneOffsets { unsigned partA, partB, tipA, tipB, scaleA, scaleB, mat; };
// issue the loads of one micro-operation: its matrix table (320 bytes of the matrix stream, lanes 0..19, by LDS-DMA to
// the wave's table buffer `ldsDst` — no registers; tools/glds_probe.hip), then only the groups it needs (WF_* bits of the
// flags).  Registers of a skipped group keep their contents (a hold-slot operand is placed in xa0..xb1 by the caller).
#define MI355_TABLE_DMA                                                                                                  \
            "s_mov_b32 m0, %[dst]\n\t"                                                                                   \
            "s_mov_b64 exec, 0xfffff\n\t"                                                                                \
            "global_load_lds_dwordx4 %[oM], %[strm]\n\t"                                                                 \
            "s_mov_b64 exec, -1\n\t"
__device__ __forceinline__ void fetchIssue(Fetched& f, const Desc& d, const LaneOffsets& o, u64 strm, unsigned ldsDst) {
    asm volatile(MI355_TABLE_DMA
        "s_bitcmp1_b32 %[fl], 0\n\t"
        "s_cbranch_scc0 .Lfx%=\n\t"
        "global_load_dwordx4 %[xa0], %[oPA], %[src1]\n\t"
        "global_load_dwordx4 %[xa1], %[oPA], %[src1] offset:16\n\t"
        "global_load_dwordx4 %[xb0], %[oPB], %[src1]\n\t"
        "global_load_dwordx4 %[xb1], %[oPB], %[src1] offset:16\n"
        ".Lfx%=:\n\t"
        "s_bitcmp1_b32 %[fl], 1\n\t"
        "s_cbranch_scc0 .Lft1%=\n\t"
        "global_load_ubyte %[s1a], %[oTA], %[src1]\n\t"
        "global_load_ubyte %[s1b], %[oTB], %[src1]\n"
        ".Lft1%=:\n\t"
        "s_bitcmp1_b32 %[fl], 2\n\t"
        "s_cbranch_scc0 .Lft2%=\n\t"
        "global_load_ubyte %[s2a], %[oTA], %[src2]\n\t"
        "global_load_ubyte %[s2b], %[oTB], %[src2]\n"
        ".Lft2%=:\n\t"
        "s_bitcmp1_b32 %[fl], 3\n\t"
        "s_cbranch_scc0 .Lfi%=\n\t"
        "global_load_dwordx2 %[inva], %[oSA], %[scale]\n\t"
        "global_load_dwordx2 %[invb], %[oSB], %[scale]\n"
        ".Lfi%=:"
        : [xa0] "+v"(f.xa0), [xa1] "+v"(f.xa1), [xb0] "+v"(f.xb0), [xb1] "+v"(f.xb1), [s1a] "+v"(f.s1a), [s1b] "+v"(f.s1b),
          [s2a] "+v"(f.s2a), [s2b] "+v"(f.s2b), [inva] "+v"(f.inva), [invb] "+v"(f.invb)
        : [fl] "s"(d.flags), [oPA] "v"(o.partA), [oPB] "v"(o.partB), [oTA] "v"(o.tipA), [oTB] "v"(o.tipB), [oSA] "v"(o.scaleA),
          [oSB] "v"(o.scaleB), [oM] "v"(o.mat), [src1] "s"(d.src1), [src2] "s"(d.src2), [scale] "s"(d.scale), [strm] "s"(strm),
          [dst] "s"(ldsDst)
        : "memory", "scc");
}
// The loads of `f` have landed once at most N younger vector-memory instructions are outstanding; jump = 8 N + 12 is the
// byte offset of "s_waitcnt vmcnt(N)" in the table below, counted from the instruction after s_getpc_b64 (every
// instruction here is 4 bytes; an entry is a wait and a branch).
#define MI355_WAIT_TABLE                                                                                                 \
        "s_getpc_b64 s[80:81]\n\t"                                                                                       \
        "s_add_u32 s80, s80, %[jump]\n\t"                                                                                \
        "s_addc_u32 s81, s81, 0\n\t"                                                                                     \
        "s_setpc_b64 s[80:81]\n\t"                                                                                       \
        "s_waitcnt vmcnt(0)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(1)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(2)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(3)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(4)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(5)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(6)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(7)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(8)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(9)\n\ts_branch .Lwd%=\n\t"                                                                      \
        "s_waitcnt vmcnt(10)\n\ts_branch .Lwd%=\n\t"                                                                     \
        "s_waitcnt vmcnt(11)\n\ts_branch .Lwd%=\n\t"                                                                     \
        "s_waitcnt vmcnt(12)\n"                                                                                          \
        ".Lwd%=:"
__device__ __forceinline__ void fetchWait(Fetched& f, unsigned jump) {
    asm volatile(MI355_WAIT_TABLE
        : "+v"(f.xa0), "+v"(f.xa1), "+v"(f.xb0), "+v"(f.xb1), "+v"(f.s1a), "+v"(f.s1b), "+v"(f.s2a), "+v"(f.s2b), "+v"(f.inva), "+v"(f.invb)
        : [jump] "s"(jump) : "memory", "scc", "s80", "s81");
}
// the stores of one micro-operation (maskA / maskB = the lanes whose first / second pattern really stores; nothing is
// issued when the micro-operation has no destination).  Default cache policy: these are HALF-LINE stores (16 bytes at a
// 32-byte stride, twice), which the non-temporal hint makes slower (tools/hbm_write_probe.hip: 2.0 against 4.9 TB/s;
// config E, which runs entirely on this kernel: 179 -> 145 us; tools/generic_store_policy.sh).  The assembly loop's stores
// are full lines and keep the hint (12 500 patterns: 121 against 144 us, tools/fast_store_policy.sh).
__device__ __forceinline__ void storeIssue(const v4d ra, const v4d rb, unsigned flags, u64 maskA, u64 maskB, unsigned oPartA, unsigned oPartB, u64 base) {
    const v2d a0 = v2d{ra.x, ra.y}, a1 = v2d{ra.z, ra.w}, b0 = v2d{rb.x, rb.y}, b1 = v2d{rb.z, rb.w};
    asm volatile(
        "s_bitcmp1_b32 %[fl], 4\n\t"
        "s_cbranch_scc0 .Lst%=\n\t"
        "s_mov_b64 exec, %[ma]\n\t"
        "global_store_dwordx4 %[oPA], %[a0], %[base]\n\t"
        "global_store_dwordx4 %[oPA], %[a1], %[base] offset:16\n\t"
        "s_mov_b64 exec, %[mb]\n\t"
        "global_store_dwordx4 %[oPB], %[b0], %[base]\n\t"
        "global_store_dwordx4 %[oPB], %[b1], %[base] offset:16\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_nop 0\n"
        ".Lst%=:"
        : : [fl] "s"(flags), [ma] "s"(maskA), [mb] "s"(maskB), [oPA] "v"(oPartA), [oPB] "v"(oPartB), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
            [base] "s"(base) : "memory", "scc");
}

// y = M x for the lane's two patterns, M spread over the lanes of `sp` (lane l = entry l & 15, row-major).  The rounding
// sequence is fixed: y_i = fma(m_i3, x3, fma(m_i2, x2, fma(m_i1, x1, fma(m_i0, x0, 0)))) — the element order
// NucleotideLikelihoodCore uses.  Eight independent accumulator chains keep the fp64 pipe busy.
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

// A micro-operation's matrix table in LDS (written by the LDS-DMA of fetchIssue from the stream k_gatherMatrices lays
// out): for each of the two branch matrices five columns of four doubles, T[s][i] = M[i][s] for a state s < 4 and 1 for
// s = 4 (missing) — what a compact tip child in state s contributes, read with two ds_read_b128 and no select.
constexpr int WALK_TABLE_BYTES = 320, WALK_TABLE_M2 = 160;
__device__ __forceinline__ v4d tipColumn(const char* tbl, unsigned s) {
    const v2d* p = reinterpret_cast<const v2d*>(tbl + (s << 5));
    const v2d lo = p[0], hi = p[1];
    return v4d{lo.x, lo.y, hi.x, hi.y};
}

// MAXT = 64 * C threads; MINW = waves per SIMD the register allocation must allow (see the file header)
template <int MAXT, int MINW>
__global__ __launch_bounds__(MAXT, MINW) void k_walk4(const unsigned MI355_CONST* __restrict__ prog, const WalkSeg MI355_CONST* __restrict__ segs,
                                                      const v2d MI355_CONST* __restrict__ matStream, int P, int C, long recipOff) {
    extern __shared__ v2d lds[];                      // hold[walkHoldSlots(C)][C][4][64] (v2d), exch[C][128] (double), table[2][MAXT / 64][320 B]
    const WalkSeg MI355_CONST& sg = segs[blockIdx.y];
    const int progStart = sg.progStart, progCount = sg.progCount, pStart = sg.pStart, pEnd = sg.pEnd, tStart = sg.tStart;
    const int p0 = pStart + (int)blockIdx.x * 128;
    if (p0 >= pEnd) return;                           // the whole workgroup
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pa = p0 + lane, pb = p0 + 64 + lane;
    const u64 validA = __ballot(pa < pEnd), validB = __ballot(pb < pEnd);
    const int qa = pa < pEnd ? pa : pEnd - 1, qb = pb < pEnd ? pb : pEnd - 1;   // lanes past the end recompute the last pattern, store nothing
    // loop-invariant 32-bit byte offsets: every address of the loop is (64-bit SGPR base from the descriptor) + one of these
    LaneOffsets o;
    o.partA = (unsigned)(((size_t)c * P + qa) * 32); o.partB = (unsigned)(((size_t)c * P + qb) * 32);
    o.tipA = (unsigned)tStart + (unsigned)walkPairIndex((size_t)(qa - pStart)); o.tipB = (unsigned)tStart + (unsigned)walkPairIndex((size_t)(qb - pStart));
    o.scaleA = o.tipA * 8u; o.scaleB = o.tipB * 8u;
    o.mat = (unsigned)(c * WALK_TABLE_BYTES + lane * 16);          // lanes 0..19 copy the wave's 320-byte table
    v2d* holdBase = lds + (size_t)c * 256 + lane;     // + slot * C * 256, quarter q at + 64 q
    double* exch = reinterpret_cast<double*>(lds + (size_t)walkHoldSlots(C) * C * 256);
    // the wave's two matrix tables (ping-pong with the fetch stage); a lane's entry (l & 15) of matrix 1 for the DPP mat-vec
    // is T[k][i] with l & 15 = 4 i + k
    constexpr int MAXC = MAXT / 64;
    const char* tbl0 = reinterpret_cast<const char*>(exch + (size_t)C * 128) + c * WALK_TABLE_BYTES;
    const unsigned tblDst0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)tbl0);
    const int spOff = (((lane & 3) * 4) + ((lane & 15) >> 2)) * 8;

    v4d ACCa = v4d{1.0, 1.0, 1.0, 1.0}, ACCb = ACCa;
    // the product of the factors this slice writes for the lane's two patterns (category 0's wave), as mantissa x 2^exponent: what the
    // slice leaves behind for accumulateScaleFactors (kernels.hip k_accumulateSlices) — the same operations as the assembly loop's
    double pmA = 1.0, pmB = 1.0;
    int peA = 0, peB = 0;
    const unsigned MI355_CONST* dp = prog + (size_t)progStart * 16;   // the host pads every segment: progCount is even and
    Desc D0 = loadDesc(dp), D1 = loadDesc(dp + 16);                    // two more descriptors (no-ops) follow it
    // the matrix stream: entry k is the [C] matrix tables of the program's k-th micro-operation (k_gatherMatrices)
    const unsigned strmStep = (unsigned)C * WALK_TABLE_BYTES;
    u64 strm = (u64)(matStream) + (u64)progStart * strmStep;
    Fetched A, B;
    A.xa0 = A.xa1 = A.xb0 = A.xb1 = v2d{1.0, 1.0}; A.s1a = A.s1b = A.s2a = A.s2b = 0x404u; A.inva = A.invb = 1.0;
    B = A;
    fetchIssue(A, D0, o, strm, tblDst0);

    // one micro-operation: CUR holds its operands (issued one stage ago), NXT receives those of the following one
#define WALK_STAGE(CUR, NXT, DCUR, DNXT, TB)                                                                                \
    {                                                                                                                     \
        const unsigned fl = DCUR.flags;                                                                                   \
        const u64 dStore = DCUR.store, dScale = DCUR.scaleW, dSrc2 = DCUR.src2;                                            \
        const int k1n = (DNXT.flags >> 5) & 7;         /* a hold-slot operand of the NEXT micro-operation is read now */  \
        if (k1n >= WK_H0) {                                                                                               \
            const v2d* h = holdBase + (size_t)(k1n - WK_H0) * C * 256;                                                    \
            NXT.xa0 = h[0]; NXT.xa1 = h[64]; NXT.xb0 = h[128]; NXT.xb1 = h[192];                                          \
        }                                                                                                                 \
        strm += strmStep;                                                                                                 \
        fetchIssue(NXT, DNXT, o, strm, tblDst0 + (1 - TB) * MAXC * WALK_TABLE_BYTES);                             \
        const int k1 = (fl >> 5) & 7, k2 = (fl >> 8) & 7, hold = (fl >> 11) & 3, smode = (fl >> 13) & 3;                  \
        fetchWait(CUR, (fl >> 16) & 0xffu);    /* 8 N + 12, N = younger loads (kernels.h walkWaitJump) */         \
        const unsigned t1a = CUR.s1a, t1b = CUR.s1b, t2a = CUR.s2a, t2b = CUR.s2b;                                        \
        const char* tb = tbl0 + TB * MAXC * WALK_TABLE_BYTES;   /* this micro-operation's table landed with its loads */ \
        v4d fa, fb, ga, gb;                                                                                               \
        if (k1 == WK_TIPS) { fa = tipColumn(tb, t1a); fb = tipColumn(tb, t1b); }                                          \
        else matvecDpp2(*reinterpret_cast<const double*>(tb + spOff), v4d{CUR.xa0.x, CUR.xa0.y, CUR.xa1.x, CUR.xa1.y},    \
                        v4d{CUR.xb0.x, CUR.xb0.y, CUR.xb1.x, CUR.xb1.y}, fa, fb);                                         \
        if (k2 == WK_TIPS) { ga = tipColumn(tb + WALK_TABLE_M2, t2a); gb = tipColumn(tb + WALK_TABLE_M2, t2b); }          \
        else if (k2 == WK_ACC) matvecDpp2(*reinterpret_cast<const double*>(tb + WALK_TABLE_M2 + spOff), ACCa, ACCb, ga, gb);   \
        else {                                         /* both children in memory (rare): the second one is not prefetched */ \
            v2d y0, y1, y2, y3;                                                                                           \
            asm volatile("global_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %4, %6 offset:16\n\t"                \
                         "global_load_dwordx4 %2, %5, %6\n\tglobal_load_dwordx4 %3, %5, %6 offset:16\n\ts_waitcnt vmcnt(0)"   \
                         : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(o.partA), "v"(o.partB), "s"(dSrc2) : "memory"); \
            matvecDpp2(*reinterpret_cast<const double*>(tb + WALK_TABLE_M2 + spOff), v4d{y0.x, y0.y, y1.x, y1.y},         \
                       v4d{y2.x, y2.y, y3.x, y3.y}, ga, gb);                                                              \
        }                                                                                                                 \
        /* descriptor k + 2 (used two stages on).  Scalar loads share the LDS counter and return out of order, so every   \
           LDS wait also waits for them: issued HERE, behind the stage's LDS reads, they have the arithmetic below to land */ \
        v4d ra = fa * ga, rb = fb * gb;                                                                                   \
        asm volatile("" : "+v"(ra), "+v"(rb));         /* (keeps the loads below behind the products' LDS wait) */         \
        DCUR = loadDesc(dp + 32);                                                                                         \
        dp += 16;                                                                                                         \
        if (smode == WS_READ) { ra = ra * CUR.inva; rb = rb * CUR.invb; }                                                 \
        else if (smode == WS_WRITE) {                                                                                     \
            double ma = fmax(fmax(fmax(0.0, ra.x), fmax(ra.y, ra.z)), ra.w), mb = fmax(fmax(fmax(0.0, rb.x), fmax(rb.y, rb.z)), rb.w); \
            v2d* e = reinterpret_cast<v2d*>(exch);                                                                        \
            e[c * 64 + lane] = v2d{ma, mb};                                                                               \
            __syncthreads();                                                                                              \
            ma = 0.0; mb = 0.0;                                                                                           \
            for (int cc = 0; cc < C; cc++) { const v2d t = e[cc * 64 + lane]; ma = fmax(ma, t.x); mb = fmax(mb, t.y); }   \
            __syncthreads();                           /* the exchange buffer is free again */                            \
            if (!(ma > 0.0)) ma = 1.0;                                                                                    \
            if (!(mb > 0.0)) mb = 1.0;                                                                                    \
            const double ia = 1.0 / ma, ib = 1.0 / mb;                                                                    \
            ra = ra * ia; rb = rb * ib;                                                                                   \
            /* category 0 stores the pair's factors (8 bytes per pattern, plain layout) and their reciprocals (pair-       \
               interleaved layout, recipOff doubles further on); then drain */                                            \
            const unsigned oFA = o.partA - (unsigned)c * (unsigned)P * 32u, oFB = o.partB - (unsigned)c * (unsigned)P * 32u;   /* 32 q */ \
            const unsigned oRA = o.scaleA + (unsigned)recipOff * 8u;                                                      \
            const unsigned oRB = o.scaleB + (unsigned)recipOff * 8u;                                                  \
            asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dwordx2 %2, %6, %10\n\tglobal_store_dwordx2 %3, %7, %10\n\t"  \
                         "s_mov_b64 exec, %1\n\tglobal_store_dwordx2 %4, %8, %10\n\tglobal_store_dwordx2 %5, %9, %10\n\t"     \
                         "s_mov_b64 exec, -1\n\ts_waitcnt vmcnt(0)"                                                   \
                         : : "s"(c == 0 ? validA : 0ull), "s"(c == 0 ? validB : 0ull), "v"(oFA >> 2), "v"(oRA),       \
                             "v"(oFB >> 2), "v"(oRB), "v"(ma), "v"(ia), "v"(mb), "v"(ib), "s"(dScale) : "memory");    \
            if (c == 0) {                                                                                                 \
                pmA *= ma; pmB *= mb;                                                                                     \
                peA += __builtin_amdgcn_frexp_exp(pmA); peB += __builtin_amdgcn_frexp_exp(pmB);                           \
                pmA = __builtin_amdgcn_frexp_mant(pmA); pmB = __builtin_amdgcn_frexp_mant(pmB);                           \
            }                                                                                                             \
        }                                                                                                                 \
        storeIssue(ra, rb, fl, validA, validB, o.partA, o.partB, dStore);                                                 \
        if (hold) {                                    /* this value waits for its sibling's subtree */                   \
            v2d* h = holdBase + (size_t)(hold - 1) * C * 256;                                                             \
            h[0] = v2d{ra.x, ra.y}; h[64] = v2d{ra.z, ra.w}; h[128] = v2d{rb.x, rb.y}; h[192] = v2d{rb.z, rb.w};          \
        }                                                                                                                 \
        ACCa = ra; ACCb = rb;                                                                                             \
    }

    for (int k = 0; k < progCount; k += 2) {
        WALK_STAGE(A, B, D0, D1, 0)
        WALK_STAGE(B, A, D1, D0, 1)
    }
#undef WALK_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the last no-op behind the program names the vectors that receive the slice's product of factors (engine_walk.cpp runPlan; null: none)
    const unsigned MI355_CONST* last = prog + ((size_t)progStart + progCount + 1) * 16;
    const u64 mantTo = ((u64)last[11] << 32) | last[10], expTo = ((u64)last[5] << 32) | last[4];
    if (mantTo && c == 0) {
        double MI355_GLOBAL* m = (double MI355_GLOBAL*)mantTo;
        int MI355_GLOBAL* x = (int MI355_GLOBAL*)expTo;
        m[o.tipA] = pmA; m[o.tipB] = pmB; x[o.tipA] = peA; x[o.tipB] = peB;
    }
}

// stream[k][c] = the matrix table of micro-operation k, category c (see tipColumn): 2 x 5 columns x 4 doubles, in
// program order, so that the walk's fetch stage copies it to LDS with ONE instruction at an address it only increments
// one COLUMN of the stream (4 doubles, 32 bytes: column `col` of a matrix, or the table's fifth column of ones).  MAIN region, t < n C 10:
// entry k, category c, column j of the 10 of m1's and m2's tables — contiguous, every line written whole.  CHERRY region behind it (only
// where the program has fused cherries: t up to 2 n C 10), the same indexing: the tables of a fused cherry's two matrices where
// micro-operation k has one; nothing is written elsewhere (the walk never reads it).
// (Round 6's first form interleaved the two per entry, 640 bytes: every main half then ended in the middle of a line, and the gather of a
// small partitioned alignment moved three times its bytes — read-modify-write of the shared lines: config E 13 -> 20 us.)
__device__ __forceinline__ void gatherOne(const WalkOp* __restrict__ prog, double* __restrict__ stream, size_t t, int n, int C,
                                          const double* const* __restrict__ cherryMats) {
    const unsigned mainCount = (unsigned)n * C * 10;                   // (the stream stays below 4 GiB: 32-bit arithmetic throughout)
    const bool cherry = t >= mainCount;
    const unsigned u = cherry ? (unsigned)t - mainCount : (unsigned)t;
    const unsigned k = u / ((unsigned)C * 10), r = u - k * ((unsigned)C * 10), c = r / 10, j = r - c * 10, m = j >= 5 ? 1 : 0, col = j - m * 5;
    const double* src = cherry ? cherryMats[2 * k + m] : (m == 0 ? prog[k].m1 : prog[k].m2);
    if (!src) return;
    const double MI355_GLOBAL* M = gptr(src) + c * 16;
    double2 lo, hi;
    if (col < 4) { lo = make_double2(M[col], M[4 + col]); hi = make_double2(M[8 + col], M[12 + col]); }
    else lo = hi = make_double2(1.0, 1.0);
    double2* out = reinterpret_cast<double2*>(stream + (size_t)t * 4);
    out[0] = lo; out[1] = hi;
}
__global__ void k_gatherMatrices(const WalkOp* __restrict__ prog, int n, int C, double* __restrict__ stream, const double* const* __restrict__ cherryMats) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * C * (cherryMats ? 20 : 10)) return;
    gatherOne(prog, stream, t, n, C, cherryMats);
}

// The same, and in the same launch the matrix snapshots of the plan's new definitions (kernels.hip k_snapshot): the stream's
// entries of those definitions are gathered from the snapshots' SOURCES (engine_walk.cpp runPlan points m1 / m2 there), so the
// two halves do not depend on each other and one launch does for both.
// ... and (round 5) the uploads queued on the instance — the program itself, when it was staged by this call: `prog` and `srcDst` are
// then read through the host ring's device mapping, the copy blocks put the program where the walk will read it; a partial update
// or an evaluation on a list the engine has not seen is three launches instead of four.
__global__ __launch_bounds__(256) void k_gatherAndSnapshot(const WalkOp* __restrict__ prog, int n, int C, double* __restrict__ stream, int gatherBlocks,
                                    double* __restrict__ matrices, const int* __restrict__ srcDst, int elems, int nPairs, const HostCopyList L,
                                    const double* const* __restrict__ cherryMats) {
    // (a snapshot is C x 16 doubles — 512 bytes with four categories —: a thread moves 16 bytes of one, a workgroup 256 / (8 C) of them.
    // A workgroup per snapshot, as k_snapshot has it for the wide state counts, left three of its four waves idle, and a partitioned
    // alignment of 1 600 taxa takes 26 000 snapshots per evaluation: config E's launch 11.6 -> see profiles/r06_experiments.txt)
    const int unitsPerPair = elems >> 1, snapBlocks = (int)(((size_t)nPairs * unitsPerPair + 255) / 256);
    if ((int)blockIdx.x >= gatherBlocks + snapBlocks) { hostCopyBlock(L, blockIdx.x - (unsigned)(gatherBlocks + snapBlocks)); return; }
    if ((int)blockIdx.x >= gatherBlocks) {
        const unsigned u = ((unsigned)blockIdx.x - (unsigned)gatherBlocks) * 256u + threadIdx.x, k = u / (unsigned)unitsPerPair, e = u - k * (unsigned)unitsPerPair;
        if ((int)k >= nPairs) return;
        const double2* s = reinterpret_cast<const double2*>(matrices + (size_t)srcDst[2 * k] * elems);
        double2* d = reinterpret_cast<double2*>(matrices + (size_t)srcDst[2 * k + 1] * elems);
        d[e] = s[e];
        return;
    }
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * C * (cherryMats ? 20 : 10)) return;
    gatherOne(prog, stream, t, n, C, cherryMats);
}
void launchGatherAndSnapshot(hipStream_t stream, const WalkOp* dProg, int nOps, int C, void* dStream, double* matrices, const int* dSrcDst, int nPairs, int elems,
                             const HostCopyList* copies, int copyBlocks, const double* const* cherryMats) {
    HostCopyList none;
    none.n = 0;
    if (nPairs < 0) nPairs = 0;
    if (!copies || copyBlocks <= 0) { copies = &none; copyBlocks = 0; }
    if (nOps <= 0) { if (copyBlocks) launchHostCopies(stream, *copies, copyBlocks); launchSnapshotMatrices(stream, matrices, dSrcDst, nPairs, elems); return; }
    const size_t total = (size_t)nOps * C * (cherryMats ? 20 : 10);
    const int gatherBlocks = (int)((total + 255) / 256);
    const int snapBlocks = (int)(((size_t)nPairs * (elems >> 1) + 255) / 256);          // (elems = 16 C: even)
    hipLaunchKernelGGL(k_gatherAndSnapshot, dim3((unsigned)(gatherBlocks + snapBlocks + copyBlocks)), dim3(256), 0, stream, dProg, nOps, C, (double*)dStream,
                       gatherBlocks, matrices, dSrcDst, elems, nPairs, *copies, cherryMats);
}

void launchGatherMatrices(hipStream_t stream, const WalkOp* dProg, int nOps, int C, void* dStream, const double* const* cherryMats) {
    if (nOps <= 0) return;
    const size_t total = (size_t)nOps * C * (cherryMats ? 20 : 10);
    hipLaunchKernelGGL(k_gatherMatrices, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dProg, nOps, C, (double*)dStream, cherryMats);
}

// ---- the assembly loop ------------------------------------------------------------------------------------------------
#include "walk4_fast_loop.inc"
// Same mapping, LDS layout (hold slots, then the matrix tables; no exchange buffer) and arithmetic as k_walk4; the loop
// itself is one block of assembly with its own register map (tools/gen_walk4_fast.py says why and what it leaves to k_walk4).
// One launch for ALL slices of a program (flags != nullptr).  Slices of a wave are independent; a slice of a later wave reads
// what earlier slices stored — for the same 128 patterns only.  So instead of a launch per wave (a chip-wide barrier, with the
// tail of every wave running on a few CUs), a workgroup waits for exactly the slices it reads from: flags[slice][x] carries the
// launch's epoch once that slice's workgroup x has its results out.  gfx950 dispatches workgroups in index order (x fastest, then
// the slice) and the engine orders the slices critical path first with every slice behind the ones it waits for, so a waiting
// workgroup only ever waits for workgroups dispatched before it.  That order is observed behaviour, not a documented guarantee:
// the wait is therefore BOUNDED, and a workgroup whose wait runs out computes what it was waiting for itself (see the kernel) —
// no deadlock whatever the dispatcher does.
// Visibility across the 8 XCDs (private L2s): the loop's result stores and its loads of stored results carry the device-scope
// bit (sc1: written through to memory before the acknowledgement; read past any stale line — tools/gen_walk4_fast.py), so a
// producer only has to drain its stores (the loop ends with s_waitcnt vmcnt(0)) before the flag goes up and a consumer only has
// to see the flag.  A write-back / invalidate of the whole L2 per workgroup (buffer_wbl2 / buffer_inv, what a release / acquire
// fence at agent scope costs) made the launch five times slower instead of faster.
// the lane's index without a register that has to survive the assembly block (which leaves the compiler two VGPRs)
// (volatile: formed where it is used, every time — as a plain expression the compiler computes it once in front of the slice loop and
// keeps it in a vector register across the assembly block, which it can only do by spilling it)
__device__ __forceinline__ int walkLane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
#ifdef BEAGLE_MI355_LAB
// LAB builds, BEAGLE_MI355_WALK_TRACE=1: every workgroup's wall-clock ticks (100 MHz) at entry, behind its dependency wait and at
// its end, [slice][group][3] — where a small shard's launch spends its time (profiles/r05_experiments.txt 17)
__device__ unsigned long long* g_walkTrace = nullptr;
void setWalkTrace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walkTrace), &p, sizeof(p)); }
#endif
template <int MAXC>
__global__ __launch_bounds__(MAXC * 64, 4) void k_walk4_fast(const unsigned MI355_CONST* __restrict__ prog, const WalkSeg MI355_CONST* __restrict__ segs,
                                                             const v2d MI355_CONST* __restrict__ matStream, int P, int C, unsigned recipOffBytes,
                                                             const int MI355_CONST* __restrict__ deps, unsigned* __restrict__ flags, unsigned epoch, int flagStride,
                                                             unsigned long long spinLimit, unsigned* __restrict__ selfServed, const RootFused rootArgs,
                                                             unsigned* __restrict__ tickets, int xcdGroups, int nRows, unsigned cherryOff) {
    // hold[2][C][4 KiB], table[3][MAXC][320 B], then ONE region shared by the three 1 KiB maximum buffers of write-mode rescaling and
    // the cherry halves of the table buffers, [3][MAXC][320 B] (a program that rescales in write mode has no fused cherries: runPlan)
    extern __shared__ v2d lds[];
    // Which (slice row y, pattern group bx) this workgroup is.  Plain: the 2-D grid, x fastest.  XCD-aware (xcdGroups > 0; SMALL launches on
    // tickets only — launchWalk4Fast says when): a 1-D grid in which the rows are taken eight at a time and workgroup id = group * 8 + row
    // within its eight.  The hardware hands workgroup i to XCD i mod 8, so all pattern groups of a row run on ONE XCD and the row's
    // descriptors and matrix tables are fetched into one L2 instead of into as many as the row has groups.
    int y, bx;
    if (xcdGroups > 0) {
        const unsigned per = 8u * (unsigned)xcdGroups, batch = blockIdx.x / per, r = blockIdx.x % per;
        y = (int)(batch * 8u + (r & 7u)); bx = (int)(r >> 3);
        if (y >= nRows) return;
    } else { y = (int)blockIdx.y; bx = (int)blockIdx.x; }
#ifdef BEAGLE_MI355_LAB
    const int gridX = xcdGroups > 0 ? xcdGroups : (int)gridDim.x;
#endif
    const WalkSeg MI355_CONST& sg = segs[y];
    const int pStart = sg.pStart, pEnd = sg.pEnd;
    const int p0 = pStart + bx * 128;
    if (p0 >= pEnd || sg.progCount <= 0) return;      // (no workgroup waits for this one: its dependants leave the same way)
    const unsigned c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef BEAGLE_MI355_LAB
    unsigned long long* trace = g_walkTrace ? g_walkTrace + ((size_t)y * gridX + bx) * 3 : nullptr;
    if (trace && threadIdx.x == 0) trace[0] = wall_clock64();
#endif
    volatile int* word = reinterpret_cast<volatile int*>(lds);          // (LDS is free between programs)
    // Forward progress does not rest on the order in which the hardware dispatches workgroups.  A workgroup polls the flags of the
    // slices it reads from for at most `spinLimit` ticks of the 100 MHz clock; past that it stops waiting and SERVES ITSELF: it walks
    // every slice in front of its own (same pattern range, device order: a slice's dependencies come before it) whose flag for this
    // pattern group is not up, then its own.  Everything a slice reads was computed for the same 128 patterns, so a workgroup can
    // always finish alone; a result computed twice is stored twice with the same bits.  On gfx950 dispatch IS in index order and
    // the limit is never reached (selfServed counts the workgroups that did; tests force it with a limit of 0).
    int first = y;
    if (flags) {
        const int depCount = sg.depCount;
        if (depCount > 0) {
            if (c == 0) {
                const int MI355_CONST* dl = deps + sg.depStart;
                // (the clock is looked at on every 32nd poll: a poll that also reads the clock notices the flag a little later, and a
                // small evaluation's time is a chain of such hand-overs — config D: 2.5 us of 46)
                unsigned long long t0 = 0;
                bool late = false;
                if (spinLimit == 0) {                    // (the forward-progress test: nobody waits; whoever finds a flag down serves itself)
                    for (int d = walkLane(); d < depCount; d += 64)
                        late = late || __hip_atomic_load(flags + (size_t)dl[d] * flagStride + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch;
                } else
                for (int d = walkLane(); d < depCount && !late; d += 64) {
                    const unsigned* f = flags + (size_t)dl[d] * flagStride + bx;
                    unsigned polls = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        if ((++polls & 31u) == 0u) {
                            const unsigned long long now = wall_clock64();
                            if (t0 == 0) t0 = now;
                            else if (now - t0 >= spinLimit) { late = true; break; }
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                const bool anyLate = __ballot(late) != 0ull;
                if (walkLane() == 0) *word = anyLate ? 1 : 0;
            }
            __syncthreads();
            // (no second barrier on the way out: nothing writes LDS word 0 before the loop's own prologue barrier, which every wave
            // reaches behind this read)
            const int anyLate = __builtin_amdgcn_readfirstlane(*word);
            if (anyLate) {
                __syncthreads();
                first = 0;
                if (c == 0 && walkLane() == 0) atomicAdd(selfServed, 1u);
            }
        }
    }
#ifdef BEAGLE_MI355_LAB
    if (trace && threadIdx.x == 0) trace[1] = wall_clock64();
#endif
    const unsigned strmStep = (unsigned)C * WALK_TABLE_BYTES;
    const unsigned ldsBase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    const unsigned hold = ldsBase + c * 4096u, holdStride = (unsigned)C * 4096u;
    const unsigned tbl = ldsBase + 2u * holdStride + c * WALK_TABLE_BYTES;
    // TICKETS (tickets != nullptr; the engine's default whenever a program's slices form a forest, planner.h PlanSeg::next): no
    // workgroup ever waits.  The grid holds the slices without dependencies only; a workgroup that has finished slice s (its stores
    // acknowledged by memory: the loop ends with s_waitcnt vmcnt(0), the result stores are written through at device scope) counts
    // itself in at tickets[next(s)][x], and the workgroup whose count completes depCount(next) carries on with that slice itself —
    // every result it reads there was stored by a workgroup that counted in before it, and is read past the L2s (sc1 loads).  The
    // others leave.  A slice above the first wave thus starts the moment its last operand is out — no workgroup slot spent polling, no
    // dispatch behind a full chip, no flag hand-over — and forward progress needs no assumption at all.  The last arrival puts the
    // counter back to zero (nobody else touches it any more in this launch), so the words are zero between launches.
    int last = y;
    for (int s = first; s <= y || tickets; s++) {
        const WalkSeg MI355_CONST& ss = segs[s];
        if (s != y && !tickets) {                     // (self-serve only)
            if (ss.pStart != pStart || ss.pEnd != pEnd || ss.progCount <= 0) continue;
            if (c == 0 && walkLane() == 0)
                *word = __hip_atomic_load(flags + (size_t)s * flagStride + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch ? 1 : 0;
            __syncthreads();
            const int done = __builtin_amdgcn_readfirstlane(*word);
            __syncthreads();
            if (done) continue;
        }
        const int progStart = ss.progStart, progCount = ss.progCount;
        const u64 dp = (u64)(prog + (size_t)progStart * 16);
        const u64 strm = (u64)matStream + (u64)progStart * strmStep;
        asm volatile(WALK4_FAST_ASM
                     : : [dp] "s"(dp), [strm] "s"(strm), [cnt] "s"(progCount), [tbl] "s"(tbl), [tblStep] "s"((unsigned)(MAXC * WALK_TABLE_BYTES)),
                         [holdStride] "s"(holdStride), [strmStep] "s"(strmStep), [pEnd] "s"(pEnd), [p0] "s"(p0),
                         [cP32] "s"(c * (unsigned)P * 32u), [cM] "s"(c * (unsigned)WALK_TABLE_BYTES), [hold] "s"(hold),
                         [exch] "s"(ldsBase + 2u * holdStride + 3u * (unsigned)(MAXC * WALK_TABLE_BYTES)), [ncat] "s"((unsigned)C),
                         [roff] "s"(recipOffBytes), [cat] "s"(c), [t0] "s"(ss.tStart + (int)bx * 128), [choff] "s"(cherryOff)
                     : WALK4_FAST_CLOBBERS);
        if (flags) {
            // (the loop ends with s_waitcnt vmcnt(0): every store of this wave has been acknowledged by memory)
            __syncthreads();
            unsigned e = epoch;
            asm volatile("" : "+s"(e));               // (formed here: a vector register that waits across the assembly block would be spilled)
            if (c == 0 && walkLane() == 0)
                __hip_atomic_store(flags + (size_t)s * flagStride + bx, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tickets) {
            last = s;
            const int nxt = ss.next;
            if (nxt < 0) break;
#ifdef BEAGLE_MI355_LAB
            if (trace && walkLane() == 0 && c == 0) trace[2] = wall_clock64();
#endif
            __syncthreads();                          // every wave's stores are out
            if (c == 0 && walkLane() == 0) {
                unsigned* t = tickets + (size_t)nxt * flagStride + bx;
                const unsigned before = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int go = before + 1u == (unsigned)segs[nxt].depCount ? 1 : 0;
                if (go) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *word = go;
            }
            __syncthreads();
            const int go = __builtin_amdgcn_readfirstlane(*word);
            __syncthreads();                          // (word 0 is hold-slot space: nobody writes it before everybody has read it)
            if (!go) return;
            s = nxt - 1;                              // (the loop's increment makes it nxt)
#ifdef BEAGLE_MI355_LAB
            if (g_walkTrace) { trace = g_walkTrace + ((size_t)nxt * gridX + bx) * 3; if (threadIdx.x == 0) { trace[0] = wall_clock64(); trace[1] = trace[0]; } }
#endif
        }
    }
#ifdef BEAGLE_MI355_LAB
    if (trace && walkLane() == 0 && c == 0) trace[2] = wall_clock64();
#endif
    // The slice that ends at the root finishes the evaluation (engine_walk.cpp PendingWalk: the launch was held back until
    // calculateRootLogLikelihoods named this slice's last result as the root): every wave's last result sits in hold slot 0
    // (the loop's exit writes it there), wave c forms sum_i pi_i L[c][p][i] for its two patterns, category 0's wave adds the
    // categories up in order, takes the logarithm and folds the group's 128 site values; the last group adds the groups' sums.
    // Same functions, same order, same bits as the launch of its own (kernels.hip k_rootSite4W).
    // (a partitioned instance: the top slice of every partition finishes ITS partition — rootParts, kernels.h)
    int part = -1;
    const RootFusedParts MI355_GLOBAL* rootPartsP = gptr(rootArgs.parts);
    if (rootPartsP) {
        const int n = rootPartsP->n;
#pragma unroll 1
        for (int i = 0; i < n; i++) if (rootPartsP->p[i].rootSeg == last) part = i;
    }
    if (rootArgs.rootSeg == last || part >= 0) {
        const RootFusedParts MI355_GLOBAL& rootParts = *rootPartsP;               // (only read where part >= 0)
        const int lane = walkLane();
        const double* freqs = part >= 0 ? rootParts.p[part].freqs : rootArgs.freqs;
        const double* catWeights = part >= 0 ? rootParts.p[part].catWeights : rootArgs.catWeights;
        const double* cum = part >= 0 ? rootParts.p[part].cum : rootArgs.cum;
        const int cumIsRaw = part >= 0 ? rootParts.p[part].cumIsRaw : rootArgs.cumIsRaw;
        // Every wave of the workgroup has left the loop before any of them writes the exchange area below — it is hold slot 1 of the
        // FIRST waves, and a wave that is a stage or two behind may still have a value parked there.  On flags the barrier in front of
        // the flag store did that; on tickets the last slice leaves the loop through a `break` in front of every barrier, and with
        // another instance's workgroups sharing the CU's SIMDs unevenly (a 20-state write-mode walk, three workgroups per CU) the waves
        // drifted far enough apart: whole pattern groups of the 4-state instance came out wrong in the two-thread test, half the runs
        // (tools/r06_flake_diag.py; profiles/r06_experiments.txt 18).
        __syncthreads();
        const double* h = reinterpret_cast<const double*>(lds) + (size_t)c * 512 + (size_t)lane * 2;       // hold slot 0: [c][4 x 1 KiB][lane x 16 B]
        const double sa = rootDot4(freqs, h[0], h[1], h[128], h[129]);
        const double sb = rootDot4(freqs, h[256], h[257], h[384], h[385]);
        double* exch = reinterpret_cast<double*>(lds) + (size_t)C * 512;                                      // hold slot 1: free at the end of a program
        exch[(size_t)c * 128 + lane * 2] = sa;
        exch[(size_t)c * 128 + lane * 2 + 1] = sb;
        __syncthreads();
        if (c == 0) {
            double sumA = 0.0, sumB = 0.0;
            for (int cc = 0; cc < C; cc++) {
                sumA = __builtin_fma(catWeights[cc], exch[(size_t)cc * 128 + lane * 2], sumA);
                sumB = __builtin_fma(catWeights[cc], exch[(size_t)cc * 128 + lane * 2 + 1], sumB);
            }
            const int pa = p0 + (lane >> 1) + 32 * (lane & 1), pb = pa + 64;
            const double g = rootWaveSum(rootFinishPair(sumA, sumB, pa, pb, pEnd, cum, cumIsRaw, rootArgs.patternWeights, rootArgs.siteLogL));
            if (part < 0) rootPublishGroup(g, lane, (int)bx, rootArgs.groups, rootArgs.blockSums, rootArgs.counter, rootArgs.out, rootArgs.flag, rootArgs.seq);
            else rootPublishGroupParts(g, lane, rootParts.p[part].blockOff + (int)bx, rootParts.totalGroups, rootParts.n, rootParts,
                                       [](const RootFusedParts MI355_GLOBAL& q, int i) { return q.p[i].groups; }, [](const RootFusedParts MI355_GLOBAL& q, int i) { return q.p[i].blockOff; },
                                       rootArgs.blockSums, rootArgs.counter, rootArgs.out, rootArgs.flag, rootArgs.seq);
        }
    }
}

void launchWalk4Fast(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, const void* dStream, int P, int C,
                     long recipOff, const int* dDeps, unsigned* flags, unsigned epoch, int flagStride, const RootFused* root,
                     unsigned long long spinLimit, unsigned* selfServed, unsigned* tickets, int nLeaves, bool xcdAware, unsigned cherryOff) {
    if (nSegs <= 0 || maxRange <= 0) return;
    if (tickets) { if (nLeaves <= 0) return; nSegs = nLeaves; flags = nullptr; }          // (the grid: the slices that wait for nothing)
    // (x = pattern group, y = slice: the chip holds little more than one slice at a time.  Dispatching slice-index-fastest
    // instead — a mix of programs resident at any moment — is SLOWER, 667 against 621 us on config A: the workgroups of a
    // slice share its descriptors in the scalar cache and its matrix tables in L2; profiles/r03_experiments.txt)
    // XCD-aware layout of the rows (see the kernel): only for launches on tickets (their rows wait for nothing) that the chip holds ALL AT ONCE
    // and that have rows for every XCD.  A launch of more workgroups than places must not use it: the dispatcher is in order, and when the XCD
    // whose turn it is has no place free every workgroup behind it waits whichever XCD it is bound for — config A ran 60 % slower that way
    // (profiles/r06_experiments.txt 5).  Small partitioned alignments (config E: 93 rows of 3-6 groups) are the case it is for: 72 -> 68 us
    // and a third of the measured traffic, which was every XCD fetching the rows' tables for itself.
    const int groupsX = (maxRange + 127) / 128;
    const bool xcd = xcdAware && tickets && nSegs >= 16 && (long)nSegs * groupsX <= 1024;
    const int xcdGroups = xcd ? groupsX : 0;
    const dim3 grid = xcd ? dim3((unsigned)(((nSegs + 7) / 8) * 8 * groupsX)) : dim3(groupsX, nSegs), block(64 * C);
    const int maxC = C <= 4 ? 4 : C <= 8 ? 8 : 16;
    // LAB builds only, BEAGLE_MI355_WALK_LDS_PAD=<bytes> (timing experiments: DESIGN.md 4.1's occupancy curve): unused LDS on top, so
    // that fewer workgroups fit a CU — 37.5 KiB: 4 per CU (4 waves per SIMD); + 16 KiB: 3; + 40 KiB: 2; + 100 KiB: 1
    static const size_t ldsPad = labEnv("BEAGLE_MI355_WALK_LDS_PAD") ? (size_t)atol(labEnv("BEAGLE_MI355_WALK_LDS_PAD")) : 0;
    const size_t lds = (size_t)2 * C * 4096 + (size_t)3 * maxC * WALK_TABLE_BYTES + std::max((size_t)3 * 1024, (size_t)3 * maxC * WALK_TABLE_BYTES) + ldsPad;
    const unsigned recipOffBytes = (unsigned)(recipOff * 8);
    const unsigned MI355_CONST* prog = (const unsigned MI355_CONST*)dProg;
    const WalkSeg MI355_CONST* segs = (const WalkSeg MI355_CONST*)dSegs;
    const v2d MI355_CONST* ms = (const v2d MI355_CONST*)dStream;
    const int MI355_CONST* deps = (const int MI355_CONST*)dDeps;
    RootFused ra;
    memset(&ra, 0, sizeof(ra));
    ra.rootSeg = -1;
    if (root) ra = *root;
    if (C <= 4 && ldsPad) { if (!grantDynamicLds(reinterpret_cast<const void*>(k_walk4_fast<4>), lds)) return; }
    if (C <= 4) hipLaunchKernelGGL((k_walk4_fast<4>), grid, block, lds, stream, prog, segs, ms, P, C, recipOffBytes, deps, flags, epoch, flagStride, spinLimit, selfServed, ra, tickets, xcdGroups, nSegs, cherryOff);
    else if (C <= 8) { if (!grantDynamicLds(reinterpret_cast<const void*>(k_walk4_fast<8>), lds)) return;
                       hipLaunchKernelGGL((k_walk4_fast<8>), grid, block, lds, stream, prog, segs, ms, P, C, recipOffBytes, deps, flags, epoch, flagStride, spinLimit, selfServed, ra, tickets, xcdGroups, nSegs, cherryOff); }
    else { if (!grantDynamicLds(reinterpret_cast<const void*>(k_walk4_fast<16>), lds)) return;
           hipLaunchKernelGGL((k_walk4_fast<16>), grid, block, lds, stream, prog, segs, ms, P, C, recipOffBytes, deps, flags, epoch, flagStride, spinLimit, selfServed, ra, tickets, xcdGroups, nSegs, cherryOff); }
}

void launchWalk4(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, const void* dStream, int P, int C, long recipOff) {
    if (nSegs <= 0 || maxRange <= 0) return;
    const dim3 grid((maxRange + 127) / 128, nSegs), block(64 * C);
    const int maxC = C <= 4 ? 4 : C <= 8 ? 8 : 16;
    const int slots = walkHoldSlots(C);
    const size_t lds = (size_t)slots * C * 256 * sizeof(v2d) + (size_t)C * 128 * sizeof(double) + (size_t)2 * maxC * WALK_TABLE_BYTES;
    const unsigned MI355_CONST* prog = (const unsigned MI355_CONST*)dProg;
    const WalkSeg MI355_CONST* segs = (const WalkSeg MI355_CONST*)dSegs;
    const v2d MI355_CONST* ms = (const v2d MI355_CONST*)dStream;
    if (C <= 4) hipLaunchKernelGGL((k_walk4<256, 4>), grid, block, lds, stream, prog, segs, ms, P, C, recipOff);
    else if (C <= 8) { if (!grantDynamicLds(reinterpret_cast<const void*>(k_walk4<512, 4>), lds)) return;     // > 64 KiB of LDS: opt in per device
                       hipLaunchKernelGGL((k_walk4<512, 4>), grid, block, lds, stream, prog, segs, ms, P, C, recipOff); }
    else { if (!grantDynamicLds(reinterpret_cast<const void*>(k_walk4<1024, 4>), lds)) return;
           hipLaunchKernelGGL((k_walk4<1024, 4>), grid, block, lds, stream, prog, segs, ms, P, C, recipOff); }
}

}  // namespace mi355
