// kernels.h — device-side data structures and launcher prototypes of the MI355X engine.
// Everything here is gfx950-only HIP; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace mi355 {

// Environment switches.  The product library reads only the documented A/B switches (every one of them leaves the
// results unchanged and selects an older or a reference code path) with plain getenv.  Tuning knobs and timing experiments —
// slice lengths, scheduling orders, LDS padding, parts of a kernel left out (some of them give WRONG results by construction) —
// go through labEnv(), which only a LAB build (-DBEAGLE_MI355_LAB: `python beast-mcmc_amd/build.py --lab`, tools/build_variant.sh)
// connects to the environment; the library a maintainer ships cannot be talked into any of them.
#ifdef BEAGLE_MI355_LAB
inline const char* labEnv(const char* name) { return getenv(name); }
#else
inline const char* labEnv(const char*) { return nullptr; }
#endif

// One pruning operation as the level kernels see it (every state count but 4): pointers already resolved on the host
// from the reference's 7-int (or 9-int) tuple {dest, writeScale, readScale, child1, matrix1, child2, matrix2
// [, partition, cumulativeScale]} (src/dr/evomodel/treelikelihood/BeagleTreeLikelihood.java:1266-1299).
// Pointers read out of a descriptor are generic ("flat") to the compiler; flat loads count against the LDS counter as well
// as the vector-memory one and cannot use scalar-base addressing.  Every buffer the engine hands the kernels lives in
// HBM, so device code converts descriptor pointers with gptr() before dereferencing.
#if defined(__HIPCC__)
#define MI355_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ T MI355_GLOBAL* gptr(T* p) { return (T MI355_GLOBAL*)p; }
#endif

struct OpDesc {
    double*        dest;        // [C][P][S] partials, written on [pStart, pEnd)
    const void*    child1;      // double [C][P][S] partials, or uint8 [P] compact states (kind bit 0)
    const void*    child2;      // same (kind bit 1)
    double*        scaleWrite;  // per-pattern raw scale factors to WRITE (rescale now), or nullptr
    const double*  scaleRead;   // per-pattern raw scale factors to READ (divide by existing), or nullptr
    int            mat1, mat2;  // transition-matrix buffer indices of the two child branches
    int            kind;        // KIND_* bits
    int            pStart, pEnd;// pattern range of this op (whole buffer unless a ...ByPartition call)
    int            pad;
};
static_assert(sizeof(OpDesc) == 64, "OpDesc layout");
enum { KIND_STATES1 = 1, KIND_STATES2 = 2, KIND_CHERRY1 = 4, KIND_CHERRY2 = 8 };
// A child that is a "virtual cherry" (T32 instances with <= 20 states): its buffer is not stored, it IS
// node(tipA over matrix matA, tipB over matrix matB) [/ scale] and the parent's kernel rebuilds its MFMA operands from the
// two tips' states and LDS copies of the two matrices.  OpDesc.child1 / child2 then hold the INDEX of the descriptor in
// the array handed to launchPruneLevelTiled.
struct CherryDesc {
    const uint8_t* tipA;
    const uint8_t* tipB;
    const double*  scale;     // the cherry node's raw scale factors (read mode), or nullptr
    int            matA, matB;
};
static_assert(sizeof(CherryDesc) == 32, "CherryDesc layout");

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute is per DEVICE: granted once per (kernel, device) —
// several instances on different GPUs of one process (BEAST's -beagle_instances) each get theirs.  false: the runtime refused.
bool grantDynamicLds(const void* kernel, size_t bytes);

// ---- the pattern walk (4 states, kernels_walk4.hip) ------------------------------------------------------------------
// One micro-operation of a walk program: node = (M1 . child1) * (M2 . child2) [* 1/scale].  A child is read from a
// partials buffer (WK_MEM), is a compact tip (WK_TIPS), or is a value the same thread computed earlier in the program:
// the previous micro-operation's result (WK_ACC, second operand only) or one of two hold slots (WK_H0/WK_H1, first
// operand only; `hold` = 1 + slot: the result of THIS micro-operation is also parked there).  The product commutes
// bitwise, so the planner is free to order the two children that way; a child in memory comes first.
enum { WK_MEM = 0, WK_TIPS = 1, WK_ACC = 2, WK_H0 = 3, WK_H1 = 4, WK_H2 = 5, WK_CHERRY = 6 };
// WK_CHERRY (second operand only, k_walk4_fast only; round 6): the child is a node over two compact tips whose own micro-operation the
// engine has FUSED into this one (engine_walk.cpp runPlan): src2 = the states of its first tip, scale = those of its second (the
// micro-operation multiplies by no reciprocals: WF_INV and WK_CHERRY exclude each other), the same entry of the matrix stream's cherry
// region = its two branch-matrix tables.  The kernel forms the child's value — column x column, what the fused micro-operation would have left
// in ACC, bit for bit — and goes on as for WK_ACC.  A third of a binary tree's internal nodes are such cherries.
// hold slots the planner may use: k_walk4 keeps all of them in LDS (4 KiB per slot and category: three fit up to 8
// categories), k_walk4_fast two in LDS and the third in registers
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int walkHoldSlots(int C) { return C <= 8 ? 3 : 2; }
enum { WS_NONE = 0, WS_READ = 1, WS_WRITE = 2 };
// flags: what the kernel's fetch stage has to load for the micro-operation (WF_*), then the kinds
enum { WF_X = 1, WF_T1 = 2, WF_T2 = 4, WF_INV = 8, WF_STORE = 16, WF_CHERRY2 = 1 << 15 };
// (bit 14 = WS_WRITE of the scale mode: the assembly loop tests it directly)
// more bits for the assembly loop k_walk4_fast (tools/gen_walk4_fast.py): first child comes from a hold slot (and which),
// second child in memory, the result is parked in a hold slot; and in bits 16..23 — which belong to the kernel that runs the program:
// k_walk4 keeps its wait-table jump there (walkWaitJump) — "the fetch skips the first / second tip-state load" (WF_NOLOAD1 / 2: the
// child is no compact tip; set in programs that do not rescale in write mode) and the stage's wait as a 4-bit code (walkWaitCode)
enum : unsigned { WF_HREAD = 1u << 24, WF_HREAD1 = 1u << 25, WF_MEM2 = 1u << 26, WF_HWRITE = 1u << 27, WF_NOLOAD1 = 1u << 16, WF_NOLOAD2 = 1u << 17,
                  WF_WAIT_SHIFT = 18, WF_HREAD2 = 1u << 31 };
// k_walk4_fast's pipeline is three micro-operations deep: the wait of stage k is "at most N vector-memory instructions outstanding",
// N = what was issued behind the small loads of k and may stay in flight (engine_walk.cpp runPlan).  A fetch is three loads, four
// for a micro-operation that multiplies by reciprocal scale factors (WF_INV), six with a fused cherry (WF_CHERRY2: its table half and
// two more tip-state pairs), so N is 6..12, + 4 behind a first child from memory, or 3..6 when the stage's own first child comes from
// memory; a tip-state load that is skipped (WF_NOLOAD1 / 2) takes one off.  Every count from 1 to 16 has its code (tools/gen_walk4_fast.py
// WAIT_N: 0..15 = 4, 3, 5, 2, 6, 7, 8, 9, 10, 11, 12, 1, 13, 14, 15, 16); more than 16 waits as for 16, less than 1 as for 1 (a smaller N
// only waits longer).
inline unsigned walkWaitCode(int n) {
    static const int codeOf[17] = {11, 11, 3, 1, 0, 2, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 15};      // index: N (0 as 1)
    return (unsigned)codeOf[n < 0 ? 0 : n > 16 ? 16 : n] << WF_WAIT_SHIFT;
}
struct WalkOp {              // 64 bytes = one scalar-cache line; every field is an ADDRESS the kernel adds a 32-bit lane offset to
    const void*    src1;     // WK_MEM: first child's partials [C][P][4];  WK_TIPS: its uint8 states
    const void*    src2;     // WK_TIPS: second child's states;  WK_MEM (both children in memory): its partials
    double*        store;    // partials buffer the result is written to (WF_STORE)
    const double*  scale;    // WF_INV: the reciprocals this result is multiplied by — a scale buffer's reciprocal half or a folded vector
                             // (engine_walk.cpp); otherwise a readable all-ones array (the assembly loop loads and multiplies only under WF_INV,
                             // which also pads a fetch behind write-mode stores: runPlan)
    // (the first 48 bytes are what the assembly loop loads per micro-operation: s_load_dwordx8 at 0, s_load_dwordx4 at 32)
    unsigned       flags;    // WF_* | k1 << 5 | k2 << 8 | hold << 11 | scaleMode << 13 | waitJump << 16 (walkWaitJump)
    unsigned       pad0;
    double*        scaleW;   // WS_WRITE: the scale buffer that receives the factors (plain layout) and, recipOff doubles on, their reciprocals
    const double*  m1;       // first / second child's branch matrix, category 0 ([C][4][4] doubles): read by
    const double*  m2;       // k_gatherMatrices, which lays them out as the stream the walk reads
};
static_assert(sizeof(WalkOp) == 64, "WalkOp layout");
// Position of pattern p in a pair-interleaved per-pattern array (walk instances: compact tip states, reciprocal scale
// factors).  Of every block of 128 patterns, lane 2 q + r of the walk owns patterns q + 32 r and 64 + q + 32 r (that
// assignment makes its result stores full cache lines: tools/gen_walk4_fast.py) — the two are neighbours here, pair after
// pair in lane order.  Such arrays are padded to a multiple of 128 entries.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t walkPairIndex(size_t p) { return (p & ~(size_t)127) + 4 * (p & 31) + 2 * ((p >> 5) & 1) + ((p >> 6) & 1); }
inline unsigned walkFlags(int k1, int k2, int hold, int smode, bool store) {
    unsigned f = (unsigned)((k1 << 5) | (k2 << 8) | (hold << 11) | (smode << 13));
    if (k1 == WK_MEM) f |= WF_X;
    if (k1 == WK_TIPS) f |= WF_T1;
    if (k2 == WK_TIPS) f |= WF_T2;
    if (smode == WS_READ) f |= WF_INV;
    if (store) f |= WF_STORE;
    if (k1 >= WK_H0) f |= WF_HREAD | (k1 == WK_H1 ? WF_HREAD1 : 0u) | (k1 == WK_H2 ? WF_HREAD2 : 0u);
    if (k2 == WK_MEM) f |= WF_MEM2;
    if (k2 == WK_CHERRY) f |= WF_CHERRY2;
    if (hold) f |= WF_HWRITE;
    return f;
}
// vector-memory instructions the kernel's fetch stage issues for a micro-operation / its store stage
// of k_walk4 (k_walk4_fast: 3, + 1 with WF_INV, + 4 with WF_X)
inline int walkFetchCount(unsigned f) { return ((f & WF_X) ? 4 : 0) + ((f & WF_T1) ? 2 : 0) + ((f & WF_T2) ? 2 : 0) + ((f & WF_INV) ? 2 : 0) + 1; }
inline int walkStoreCount(unsigned f) { return (f & WF_STORE) ? 4 : 0; }
// flags field "waitJump" of micro-operation k: 8 N + 12 with N = walkFetchCount(k+1) (engine_walk.cpp runPlan, kernels_walk4.hip)
inline unsigned walkWaitJump(int n) { return (unsigned)(8 * n + 12) << 16; }
// A program slice and the pattern range that executes it (one per partition of a partitioned instance).  The kernel is
// software-pipelined two micro-operations deep: progCount must be EVEN and two more readable descriptors must follow
// (k_walk4, k_walkT32); the assembly loop k_walk4_fast is three deep and leaves behind any stage: any progCount, three more descriptors.
// tStart: where the segment's patterns begin in the PAIR-INTERLEAVED arrays — those are laid out partition by partition, every
// partition padded to whole blocks of 128 (engine_instance.cpp setPairLayout), so that a lane's pair is one aligned load wherever a
// partition starts; partials and plain per-pattern arrays keep the caller's pattern numbering.
// depStart / depCount (one fused launch of all slices, k_walk4_fast only): the slices — rows of this array — whose stored results
// this one reads, as a range of the launch's dependency list; its workgroups wait for their flags first.
// next (a launch on tickets, k_walk4_fast only): the row of the ONE slice that reads this slice's stored result, -1: none.  Only the
// rows without dependencies get workgroups then; a workgroup that finishes row s counts itself in at tickets[next][x], and the one
// that makes the count depCount(next) carries on with row `next` itself (planner.h PlanSeg::next).
struct WalkSeg { int progStart, progCount, pStart, pEnd, tStart, depStart, depCount, next; };
// one launch: every 128-pattern group of every segment walks its program; maxRange = max (pEnd - pStart).  A lane owns two
// patterns, 64 apart; its tip states and reciprocal scale factors are stored pair-interleaved (walkPairIndex).
// dStream = the matrix stream of the WHOLE device program (launchGatherMatrices), nOps * C * 16 {M1, M2} pairs.
void launchWalk4(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, const void* dStream,
                 int P, int C, long recipOff);
// cherryMats != nullptr (programs of k_walk4_fast with fused cherries): behind the stream's nOps * C entries a CHERRY region of as many, the
// tables of cherryMats[2 k], [2 k + 1] (a fused cherry's branch matrices, category 0) at entry k where they are not null
void launchGatherMatrices(hipStream_t stream, const WalkOp* dProg, int nOps, int C, void* dStream, const double* const* cherryMats = nullptr);
// ... and the plan's matrix snapshots (src, dst index pairs) in the same launch; m1 / m2 of freshly snapshotted matrices must point at the sources
// ... and queued host copies (copies / copyBlocks: kernels_walk4.hip k_gatherAndSnapshot); dProg / dSrcDst may then be the host ring's mapping
struct HostCopyList;
void launchGatherAndSnapshot(hipStream_t stream, const WalkOp* dProg, int nOps, int C, void* dStream, double* matrices, const int* dSrcDst, int nPairs, int elems,
                             const HostCopyList* copies = nullptr, int copyBlocks = 0, const double* const* cherryMats = nullptr);
// The same walk by the assembly loop (tools/gen_walk4_fast.py).  Requirement: EVERY descriptor carries readable addresses in
// src1, src2 and scale even where unused (the small loads are unconditional): all-missing tip states / all-one scale factors.
// deps / flags / epoch / flagStride: all slices of a program in ONE launch (slice y is dispatched before y + 1): a workgroup first
// waits until flags[d * flagStride + x] == epoch for every slice d of its dependency list, and sets flags[y * flagStride + x] =
// epoch when its results are out.  flags == nullptr: no waiting, no signalling (one launch per wave of independent slices).
// what the walk's root slice needs to finish the evaluation itself (kernels_walk4.hip, root_site4.h; rootSeg < 0: nothing)
struct RootFusedParts;
struct RootFused {
    const double* catWeights; const double* freqs; const double* cum; const double* patternWeights;
    double* siteLogL; double* blockSums; unsigned* counter; double* out; unsigned long long* flag; unsigned long long seq;
    int cumIsRaw; int rootSeg; int groups; int pad;
    const RootFusedParts* parts;     // DEVICE memory, or nullptr: the roots of a partitioned instance's partitions (below)
};
// ... and the same for the roots of up to ROOT_MAX_PARTS partitions (calculateRootLogLikelihoodsByPartition behind updatePartialsByPartition:
// every partition's top slice finishes its own partition; RootFused carries what the partitions share — pattern weights, site values,
// blockSums, counter, out[partition], flag, seq —, rootSeg = -1 and a pointer to this table in device memory: by value it kept more
// scalars alive across the assembly loop than the kernel has).  A pattern group's sum goes to blockSums[blockOff + group],
// the last of ALL groups adds every partition's up in index order (root_site4.h rootPublishGroupParts; k_rootSite4WParts: the same bits).
struct RootFusedPart { const double* catWeights; const double* freqs; const double* cum; int cumIsRaw, rootSeg, blockOff, groups; };
struct RootFusedParts { RootFusedPart p[8]; int n, totalGroups; };
// spinLimit: how long a workgroup polls those flags (ticks of the 100 MHz wall clock) before it computes what it waits for itself
// (kernels_walk4.hip: forward progress whatever the dispatch order); *selfServed counts the workgroups that did.
void launchWalk4Fast(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, const void* dStream, int P, int C,
                     long recipOff, const int* dDeps = nullptr, unsigned* flags = nullptr, unsigned epoch = 0, int flagStride = 0,
                     const RootFused* root = nullptr, unsigned long long spinLimit = 0, unsigned* selfServed = nullptr,
                     unsigned* tickets = nullptr, int nLeaves = 0, bool xcdAware = false, unsigned cherryOff = 0);
// (cherryOff: byte offset of the stream's cherry region — nOps * C * 320 of the WHOLE device program — where it has fused cherries)
// (tickets != nullptr: rows 0 .. nLeaves - 1 of dSegs are the slices without dependencies — the launch's grid —, the rest follow;
// tickets[row * flagStride + x] are zero before the launch and zero again behind it; deps / flags / epoch / spinLimit unused;
// xcdAware: a launch the chip holds all at once lays its rows out so that all pattern groups of a row run on one XCD — kernels_walk4.hip)
#ifdef BEAGLE_MI355_LAB
void setWalkTrace(unsigned long long* devicePointer);          // (kernels_walk4.hip g_walkTrace; nullptr: off)
#endif
// 4-state walk instances: the root integration as a launch of its own, bit-compatible with the walk's root epilogue (root_site4.h)
void launchRootLogLikelihood4W(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                               const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                               double* blockSums, double* out, int P, int C, int pStart, int pEnd,
                               unsigned long long* flag, unsigned long long seq, unsigned* counter);

// matrices[dst[k]] = matrices[src[k]] for k < n (each C*S*S doubles): private snapshots of branch matrices
void launchSnapshotMatrices(hipStream_t stream, double* matrices, const int* dSrcDst, int n, int elems);

// ---- launchers (all asynchronous on `stream`) -------------------------------------------------

// Host -> device copies of small arrays by a kernel (engine_instance.cpp flushUploads): entry k moves `bytes` from `src` — host
// memory the device maps — to `dst`; its workgroups are firstBlock .. firstBlock + ceil(bytes / 4096) - 1.
constexpr int HOST_COPY_MAX = 12;
struct HostCopyList { struct Entry { void* dst; const void* src; unsigned bytes, firstBlock; } e[HOST_COPY_MAX]; int n; };
void launchHostCopies(hipStream_t stream, const HostCopyList& list, int blocks);
#ifdef __HIPCC__
// one workgroup of 256 threads moves 4 KiB of entry k (block = its index among the list's blocks): k_hostCopies, and the kernels the
// queued copies ride along with (k_transition4Fused, k_gatherAndSnapshot)
__device__ __forceinline__ void hostCopyBlock(const HostCopyList& L, unsigned block) {
    int k = 0;
    while (k + 1 < L.n && block >= L.e[k + 1].firstBlock) k++;
    const size_t base = (size_t)(block - L.e[k].firstBlock) * 4096;
    if (base >= L.e[k].bytes) return;                 // (an entry superseded by a later one for the same destination: bytes = 0)
    const char* s = (const char*)L.e[k].src + base;
    char* d = (char*)L.e[k].dst + base;
    const unsigned left = L.e[k].bytes - (unsigned)base, n = left < 4096u ? left : 4096u;
    const unsigned t = threadIdx.x;
    if ((((size_t)s | (size_t)d) & 15) == 0) {
        if (t * 16 + 16 <= n) *reinterpret_cast<uint4*>(d + t * 16) = *reinterpret_cast<const uint4*>(s + t * 16);
        for (unsigned i = (n & ~15u) + t; i < n; i += 256) d[i] = s[i];
    } else if ((((size_t)s | (size_t)d) & 3) == 0) {
        for (unsigned i = t * 4; i + 4 <= n; i += 1024) *reinterpret_cast<unsigned*>(d + i) = *reinterpret_cast<const unsigned*>(s + i);
        for (unsigned i = (n & ~3u) + t; i < n; i += 256) d[i] = s[i];
    } else
        for (unsigned i = t; i < n; i += 256) d[i] = s[i];
}
#endif
// dst[0..n) = src[0..n) — dst in host memory the device maps —, then *flag = seq (release, system scope): a result handed to a
// polling host thread
void launchPublish(hipStream_t stream, const double* src, int n, double* dst, unsigned long long* flag, unsigned long long seq);

// P(t) = U diag(exp(lambda * t * r_c)) U^-1, negatives clamped to 0, for `count` branches.
// dIdx/dLen/dEig/dRate: device arrays of length `count`: destination matrix index, edge length,
// eigen-system index and category-rate-set index per branch.
// complexEigen: eigen systems are [U | U^-1 | Re lambda (S) | Im lambda (S)] in real block form (kernels.hip iexpEntry)
void launchTransitionMatrices(hipStream_t stream, double* matrices, const double* eigen, const double* rates,
                              const int* dIdx, const double* dLen, const int* dEig, const int* dRate,
                              int count, int S, int C, bool complexEigen = false);

// 4 states, ONE eigen system and ONE rate set for all `count` branches (beagleUpdateTransitionMatrices), and everything the call
// needs still in the host's staging ring: eigSrc / ratesSrc / idx / len may be HOST addresses the device maps (each workgroup
// stages the eigen system and the rates in LDS with one read over the link, a thread reads its own branch length and matrix
// index), and the queued host->device copies of `pending` (kernels.h HostCopyList: the persistent copies of the same eigen
// system and rates, whatever else was queued) ride in the same launch as extra workgroups — one launch where there were a
// copy kernel and a transition kernel.  Same arithmetic and summation order as launchTransitionMatrices.
void launchTransitionMatrices4Fused(hipStream_t stream, double* matrices, const double* eigSrc, const double* ratesSrc, const int* idx,
                                    const double* len, int count, int C, bool complexEigen, const HostCopyList& pending, int copyBlocks);

// C_c = A_c * B_c per category, `count` triples (device index arrays).
void launchConvolveMatrices(hipStream_t stream, double* matrices, const int* dFirst, const int* dSecond,
                            const int* dResult, int count, int S, int C);
// C_c = A_c + B_c per category (a result may be one of its own operands: entry by entry)
void launchAddMatrices(hipStream_t stream, double* matrices, const int* dFirst, const int* dSecond, const int* dResult, int count, int S, int C);

// One dependency level of pruning operations: `nOps` independent ops, descriptors on the device.
// maxRange = max over ops of (pEnd - pStart).
void launchPruneLevel(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices,
                      int P, int S, int C, int maxRange);

// The same for up to ROOT_MAX_PARTS pattern ranges in ONE pair of launches (calculateRootLogLikelihoodsByPartition): range k
// has its own root buffer, weights, frequencies and cumulative buffer; out[k] = its sum; `flag` as above, after the last.
constexpr int ROOT_MAX_PARTS = 8;
struct RootPart { const double* root; const double* catWeights; const double* freqs; const double* cum; int cumIsRaw, pStart, pEnd, blockOff; };
struct RootParts { RootPart p[ROOT_MAX_PARTS]; int n; };
void launchRootLogLikelihoodParts(hipStream_t stream, const RootParts& parts, const double* patternWeights, double* siteLogL, double* blockSums,
                                  double* out, int P, int S, int C, unsigned long long* flag, unsigned long long seq, unsigned* counter = nullptr);
// (counter: a zeroed device word, zero again behind the launch — the last workgroup of the site kernel forms the sums; nullptr: a second launch does)
// 4-state walk instances: a wave per 128 patterns with the assembly loop's lane map, the groups' sums as the walk's own root slices leave
// them (kernels_walk4.hip, RootFusedParts) — the same bits whether a partition's root is integrated there or here.  Uses RootPart::root,
// catWeights, freqs, cum, cumIsRaw, pStart, pEnd; blockOff counts 128-pattern groups.
void launchRootLogLikelihoodParts4W(hipStream_t stream, const RootParts& parts, const double* patternWeights, double* siteLogL, double* blockSums,
                                    double* out, int P, int C, unsigned long long* flag, unsigned long long seq, unsigned* counter);

// site[p] = log(sum_c w_c sum_i pi_i root[c][p][i]) + cum[p];  blockSums[b] = sum_p weight[p]*site[p] over block b
// then out[0] = sum_b blockSums[b] in a fixed order (deterministic).  cum may be nullptr; cumIsRaw says the
// buffer holds raw factors (log is taken on the fly).  Restricted to [pStart, pEnd).
void launchRootLogLikelihood(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                             const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                             double* blockSums, double* out, int P, int S, int C, int pStart, int pEnd,
                             unsigned long long* flag = nullptr, unsigned long long seq = 0, unsigned* counter = nullptr);
// (counter: a zeroed device word — the workgroup that finishes last forms the sum itself, one launch instead of two)

// cum[p] += sign * sum_k (raw_k ? log(src_k[p]) : src_k[p]) on [pStart, pEnd); srcs/raws are device arrays.
void launchAccumulateScale(hipStream_t stream, double* cum, const double* const* dSrcs, const int* dRaw,
                           int count, double sign, int pStart, int pEnd);

// The same from what a write-mode walk left behind per SLICE (walk instances, round 6): slice row r holds, per pattern at its position q in
// the pair-interleaved layout (pairPos[p]; nullptr: walkPairIndex(p)), the product of the factors the slice wrote as mantissa[r * stride + q]
// x 2^exponent[r * stride + q];  cum[p] += sign * sum over rows[0 .. n) of (log(mantissa) + exponent ln 2), in that order.
void launchAccumulateSlices(hipStream_t stream, double* cum, const double* mant, const int* expo, const int* dRows, int n, size_t stride,
                            const unsigned* dPairPos, double sign, int pStart, int pEnd);
void launchFill(hipStream_t stream, double* dst, double value, int pStart, int pEnd);
// dst[j][i] = prod_m srcs[m][i] over m in [start[j], start[j + 1]), i < len; worst[j] (zeroed by the caller) = bit pattern of job j's
// largest product, +infinity for anything not finite (kernels.hip k_foldReciprocals)
// invert: the sources are factors (<= 1), the range check is on 1 / product (the T32 walk, which divides by what it reads)
void launchFoldReciprocals(hipStream_t stream, const double* const* dSrcs, const int* dStart, double* const* dDst, int nJobs, int len, unsigned long long* dWorst,
                           bool invert = false, bool pairs = true);        // (pairs: two elements per thread where `len` is even — the same bits)
// out[p] = raw ? log(in[p]) : in[p]
void launchLogScale(hipStream_t stream, const double* in, double* out, int raw, int P);
// Read-back: out[c][p][i] (API layout) = partials * scale, from either device layout.  `scale` may be
// nullptr; scaleIsRaw: the buffer holds raw factors (else logs: the factor is exp).  One pass, then a single D2H.
// walk instances, after the pattern partitions changed: pair[pos[p]] = plain[p] for a tip's states (the rest "missing"), and
// the reciprocal half of a raw scale buffer rebuilt from its factors
void launchRelayoutStates(hipStream_t stream, const uint8_t* oldPlain, uint8_t* newPlain, uint8_t* newPair, const unsigned* dPairPos, int P);
// ... for a list of tips at once: newPair[0 .. pairLen) = missing, then the scatter (two launches whatever the number of tips)
struct RelayoutJob { const uint8_t* oldPlain; uint8_t* newPlain; uint8_t* newPair; };
void launchRelayoutStatesBatch(hipStream_t stream, const RelayoutJob* dJobs, int nJobs, const unsigned* dPairPos, int P, int pairLen, int missing);
void launchRecipFromFactors(hipStream_t stream, const double* factors, double* recip, const unsigned* dPairPos, int P);
void launchExportPartials(hipStream_t stream, const double* partials, const double* scale, int scaleIsRaw, double* out,
                          int P, int S, int C, bool tiled);
// dst[c][p][i] = src[p][i] for every category (setTipPartials replication)
void launchReplicateCategories(hipStream_t stream, const double* src, double* dst, int P, int S, int C);

// ---- pre-order partials and edge derivatives (kernels_preorder.hip) -----------------------------
// One level of pre-order ops.  The OpDesc fields are reused: dest = pre(child), child1 = pre(parent) (always partials),
// mat1 = the child's branch matrix (used transposed), child2 / mat2 = the sibling's post-order partials (or compact
// states, KIND_STATES2) and branch matrix.  Works on either partials layout.
// recipOff != 0: a rescaling op also stores the reciprocal of its factor at scaleWrite[recipOff + walkPairIndex(p)] (walk instances)
void launchPrePartials(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C, bool tiled,
                       int maxRange, long recipOff);
struct EdgeDesc {
    const void*   post;          // post-order partials of the node below the edge (double*) or its compact states (uint8*)
    const double* pre;           // pre-order partials of the same node
    const double* tmp;           // launchEdgeReduce only: pre[j] * (D . post)[j], produced by a pass of the pruning kernel
    int           dmat;          // differential matrix index (C*S*S doubles)
    int           postIsStates;
    int           slot;          // output row: perPattern[slot][P], blockSums[slot][blocks][2]
    int           pad;
};
// 16..64 states, T32 layout (kernels_mfma.hip k_edgeTiled): pre and post read once, the product with D on the matrix cores; same
// outputs as launchEdgeDifferentials (blockSums rows of edgeBlocks(P) entries).  EdgeDesc::tmp unused.
bool launchEdgeTiled(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* matrices, const double* catWeights,
                     const double* patternWeights, double* perPattern, double* blockSums, int P, int S, int C);
// 4 states, plain layout (kernels_preorder4.hip): a NODE of the pre-order pass — both children's pre-order partials from one
// read of pre(parent), post(a), post(b), and both edges' derivative sums on the way.  preA / preB may be NULL (not stored);
// slotA / slotB < 0: no derivative asked for that edge; dA / dB: the edges' differential matrices.
struct PreNodeJob {
    const double* preParent;
    const void*   postA;         // double [C][P][4], or uint8 states (statesA)
    const void*   postB;
    double*       preA;
    double*       preB;
    int           matA, matB, dA, dB;
    int           slotA, slotB, statesA, statesB;
    double        pad;
};
static_assert(sizeof(PreNodeJob) == 80, "PreNodeJob layout");
void launchPreNodes4(hipStream_t stream, const PreNodeJob* dJobs, int nJobs, const double* matrices, const double* catWeights,
                     const double* patternWeights, double* blockSums, int P, int C);
// 4 states: the edge derivative SUMS of a whole pre-order list in one launch, no pre-order partial written (kernels_preorder4.hip
// k_preWalk4).  One descriptor per node of the list, in depth-first order; the node's own pre-order partial is in the thread's
// registers (it was produced by the previous descriptor) or in one of the LDS hold slots.
struct PreWalkOp {
    const void*    postA;        // double [C][P][4]; any valid partials buffer when the child is a compact tip
    const void*    postB;
    const uint8_t* tipA;         // uint8 states [P]; any valid states array when the child has partials
    const uint8_t* tipB;
    double*        storeA;       // continuation PW_CONT_STORE: where the child's pre-order partial is written (it heads another segment)
    double*        storeB;
    const double*  recipA;       // 1 / (scale factor of post(a)), pair-interleaved (a walk instance's reciprocal array); all ones when
    const double*  recipB;       // the child is a tip or its partials carry no factor
    int            matA, matB;   // the children's branch matrices
    int            dA, dB;       // the edges' entries in the launch's product array (launchEdgeProducts): the slot, normally
    int            slotA, slotB; // where the edges' sums go (the launch's last slot = nobody asked)
    unsigned       flags;        // PW_* below
    int            pad;
};
static_assert(sizeof(PreWalkOp) == 96, "PreWalkOp layout");
constexpr unsigned PW_TIP_A = 1u, PW_TIP_B = 2u;          // the child is a compact tip
constexpr int PW_SRC_SHIFT = 4, PW_CONT_A_SHIFT = 8, PW_CONT_B_SHIFT = 12;   // 4 bits each
// source of the node's pre-order partial: 0 = the registers, 1 + k = hold slot k
// what becomes of a child's pre-order partial: 0 = nothing below it, 1 = the registers (the next descriptor is that child),
// 2 + k = hold slot k (its descriptor comes after the other child's subtree), PW_CONT_STORE = memory (storeA / storeB)
constexpr unsigned PW_CONT_STORE = 15u;
constexpr int PW_MAX_HOLD = 13;
// A post-order operand that is NOT stored (a gradient chain keeps its tip-tip nodes, and a tip-tip node under one more tip, as
// definitions: engine_preorder.cpp) is re-evaluated from the tips where the walk needs it, by descriptors of a second kind placed
// right in front of the node's own: PW_POSTOP: (M_A x_A) * (M_B tip_B) * recipA -> post slot `dst` (LDS, PW_POST_SLOTS per thread),
// x_A a tip (tipA) or post slot `slotA` (PW_SLOT_A: the inner tip-tip node of the longer kind); matA / matB: the definition's
// private matrix snapshots; recipA: the reciprocal of the node's own scale factor.  Such a descriptor asks for the same loads as
// a node over two tips, writes no sum and leaves the walk's own state alone.  A node descriptor takes an unstored child from its
// slot: PW_SLOT_A / PW_SLOT_B (+ PW_TIP_* so that nothing but a dummy state byte is loaded for it).
constexpr unsigned PW_POSTOP = 1u << 16, PW_SLOT_A = 1u << 17, PW_SLOT_B = 1u << 18;
constexpr int PW_SLOTA_SHIFT = 19, PW_SLOTB_SHIFT = 21, PW_DST_SHIFT = 23;      // 2 bits each
// The shortest unstored operand — a node over two compact tips — is evaluated INSIDE the descriptor of its parent (round 5, second
// form: a descriptor of its own costs a whole stage whatever it computes): PW_CHERRY_A / _B.  For such a child postX points at the
// definition's two branch matrices interleaved lane by lane ({M1[l], M2[l]}, 256 bytes per category: launchCherryPairs), tipX and
// storeX (a node over two tips never heads a segment: nothing is stored for it) at the two tips' states, recipX at the reciprocal of
// the node's OWN scale factor; three loads (one 16-byte, two single bytes) where a stored child asks for two.
constexpr unsigned PW_CHERRY_A = 1u << 25, PW_CHERRY_B = 1u << 26;
// bits 28..31: the vector-memory loads the descriptor asks for (8..12: four matrices, two reciprocal factors, one / two / three per
// child) — what the wait of the descriptor BEFORE it leaves outstanding
constexpr int PW_LOADS_SHIFT = 28;
inline unsigned preWalkLoads(unsigned flags) {
    auto child = [&](unsigned tip, unsigned cherry) { return (flags & cherry) ? 3u : (flags & tip) ? 1u : 2u; };
    return 6u + child(PW_TIP_A, PW_CHERRY_A) + child(PW_TIP_B, PW_CHERRY_B);
}
constexpr int PW_POST_SLOTS = 3;
// A walk is cut into SEGMENTS that run side by side (one more grid dimension): the first one starts at the list's root and
// stores the pre-order partials of the nodes that head the others; those run in a second launch.  progCount even, two more
// no-op descriptors behind every segment.
struct PreWalkSeg { int progStart, progCount; const double* rootPre; };
static_assert(sizeof(PreWalkSeg) == 16, "PreWalkSeg layout");
// sums [nSlots + 1][waves] with waves = preWalkWaves(P, C); dProg[0] is the list's root (what the likelihood is formed from),
// listRootPre its pre-order partial
int  preWalkWaves(int P, int C);
// products: [nSlots + 1][C][16], entry e = (branch matrix of edge e) . (its differential matrix), the spare last one zeros
// (launchEdgeProducts from pairs {matrix, differential matrix}, -1 -1 for zeros); PreWalkOp::dA / dB index it
void launchEdgeProducts(hipStream_t stream, const double* matrices, const int* dPairs, double* products, int C, int n);
// out[k][c][l] = { M[pairs[2 k]][c][l], M[pairs[2 k + 1]][c][l] }, l = 0..15: what a PW_CHERRY child's postX points at
void launchCherryPairs(hipStream_t stream, const double* matrices, const int* dPairs, double* out, int C, int n);
bool launchPreWalk4(hipStream_t stream, const PreWalkOp* dProg, const PreWalkSeg* dSegs, int nSegs, const double* listRootPre,
                    const double* matrices, const double* products, const double* catWeights, const double* patternWeights, double* sums, int P, int C,
                    int holdSlots, bool postSlots = false);
void launchPreWalkFinal(hipStream_t stream, const double* sums, int nSlots, int P, int C, double* out);
// the edge derivatives alone, same shape (32-byte vector accesses); outputs as launchEdgeDifferentials
void launchEdgeDifferentials4(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* matrices, const double* catWeights,
                              const double* patternWeights, double* perPattern, double* blockSums, int P, int C);
int  edgeBlocks(int P);          // workgroups per edge = entries per edge in blockSums (x2 doubles)
// per edge: blockSums[slot][b] = {sum_p w_p num/den, sum_p w_p (num/den)^2} over workgroup b; perPattern (nullable) [slot][P];
// finish with launchEdgeFinal
void launchEdgeDifferentials(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* matrices, const double* catWeights,
                             const double* patternWeights, double* perPattern, double* blockSums,
                             int P, int S, int C, bool tiled);
// same outputs from tmp = pre * (D . post) (see EdgeDesc): num = sum_c w_c sum_j tmp, den = sum_c w_c sum_j pre * post
void launchEdgeReduce(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* catWeights, const double* patternWeights,
                      double* perPattern, double* blockSums, int P, int S, int C, bool tiled);
// outSums[2r], outSums[2r+1] = fixed-order sums of row r's block sums, r < nRows
void launchEdgeFinal(hipStream_t stream, const double* blockSums, int nRows, int P, double* outSums);
// out[S*S] = sum over edges and patterns of the weighted pre x post cross products (see kernels_preorder.hip); partial: [edgeBlocks(P)][S*S]
void launchCrossProducts(hipStream_t stream, const EdgeDesc* dEdges, int nEdges, const double* dEdgeLengths, const double* catWeights,
                         const double* catRates, const double* patternWeights, double* partial, double* out, int P, int S, int C, bool tiled);
void launchTransposeMatrices(hipStream_t stream, double* matrices, const int* dSrcDst, int count, int S, int C);
void launchFillFrequencies(hipStream_t stream, double* dest, const double* freqs, int P, int S, int C, bool tiled);

int  pruneBlocksForRange(int S, int range);

// ---- T32 layout (20-/61-state MFMA path, kernels_mfma.hip): partials[c][tile][state][32 patterns] --------------
// One dependency level on the fp64 matrix cores; anyScaleWrite adds the second (max + divide) pass.
// 17..20 states: the walk's programs on the T32 layout (kernels_mfma.hip k_walkT32): same descriptors and segments as the 4-state
// walk (src = plain tip states / T32 partials, scale = the RAW factors, no write mode); dStream from launchGatherFragments over
// the same device program (walkT32StreamBytes)
size_t walkT32StreamBytes(int nEntries, int C, int S);
void launchGatherFragments(hipStream_t stream, const WalkOp* dProg, int nEntries, int C, int S, void* dStream);
// writeMode (a program with write-mode rescaling in it — k_walkT32W1: a workgroup is all categories of one tile, hold slots in registers;
// at most WALK_T32_WRITE_MAX_CATEGORIES of them and two hold slots: false otherwise)
constexpr int WALK_T32_WRITE_MAX_CATEGORIES = 4, WALK_T32_WRITE_MAX_HOLD = 2;
bool launchWalkT32(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange, const void* dStream, int P, int S, int C, int holdSlots,
                   bool writeMode = false);
// 21..64 states: the column tables of a list's virtual cherries (kernels_mfma.hip k_cherryTables), handed to launchPruneLevelTiled
size_t cherryTableBytes(int nCherries, int S, int C);
void launchCherryTables(hipStream_t stream, const CherryDesc* dCherries, int n, const double* matrices, int S, int C, double* out);
void launchPruneLevelTiled(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C,
                           bool anyScaleWrite, const CherryDesc* dCherries = nullptr, const double* dCherryTables = nullptr);
// A level of pre-order operations on the T32 layout in ONE pass each (kernels_mfma.hip k_preOpTiled; OpDesc fields as
// launchPrePartials; no write-mode rescaling: such operations take the two passes of the pruning kernel).  false: LDS refused.
bool launchPreOpsTiled(hipStream_t stream, const OpDesc* dOps, int nOps, const double* matrices, int P, int S, int C);
// per-pattern site log-likelihoods + per-block weighted sums (finish with launchRootFinal)
int  rootSiteTiledBlocks(int patterns);      // entries launchRootSiteTiled writes to blockSums for that many patterns
void launchRootSiteTiled(hipStream_t stream, const double* root, const double* catWeights, const double* freqs,
                         const double* cum, int cumIsRaw, const double* patternWeights, double* siteLogL,
                         double* blockSums, int P, int S, int C, int pStart, int pEnd);
// out[0] = sum of n block sums in a fixed order
void launchRootFinal(hipStream_t stream, const double* blockSums, int n, double* out, unsigned long long* flag = nullptr,
                     unsigned long long seq = 0);

}  // namespace mi355
