// engine_internal.h — what the engine's translation units share: the instance record and its staging / allocation helpers
// (engine_instance.cpp), the 4-state walk runner (engine_walk.cpp), the level scheduler for every other state count
// (engine_levels.cpp), the pre-order / gradient paths (engine_preorder.cpp); the C ABI itself is engine_abi.cpp.
// All arithmetic is in the .hip files; these files validate indices, resolve buffer indices to device pointers and enqueue
// kernels on the instance's HIP stream.  Results are only observed at calculateRootLogLikelihoods / get*, so every other
// call returns as soon as its work is enqueued (SURVEY 8b "Threading").
//
// There is no CPU path in this library: with no visible MI355X beagleCreateInstance returns BEAGLE_ERROR_NO_RESOURCE.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>
#include "../../include/beagle_mi355.h"
#include "kernels.h"
#include "planner.h"
#include "sharded.h"

namespace mi355 {
namespace eng {


#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            if (getenv("BEAGLE_MI355_DEBUG"))                                                 \
                fprintf(stderr, "[beagle-mi355] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return e__ == hipErrorOutOfMemory ? BEAGLE_ERROR_OUT_OF_MEMORY : BEAGLE_ERROR_GENERAL; \
        }                                                                                     \
    } while (0)

constexpr size_t RING_BYTES = 16u << 20;   // pinned host staging ring + its device mirror
constexpr int SLAB_BUFFERS = 32;           // partials buffers per hipMalloc
constexpr int PRE_SCRATCH = 32;            // pre-order ops per two-pass chunk on the T32 layout
constexpr int GRADIENT_VIRT_STEPS = 2;     // longest definition a gradient chain can leave unstored (Instance::gradientVirtual)
constexpr int GRADIENT_VIRT_DEFAULT = 1;   // ... and what it does leave unstored by default: nodes over two tips (engine_abi.cpp)

struct Instance {
    int device = 0;
    // 4 states: every operation list runs as ONE launch of the pattern-walk kernel (kernels_walk4.hip), programmed by the
    // walk planner (planner.h), which also owns the definitions of virtual buffers
    bool walk = false;
    mi355::WalkPlanner planner;
    mi355::Plan plan;                                    // scratch of the current call
    std::vector<mi355::WalkOp> walkOps;                  // scratch: resolved program
    size_t scaleStride = 0;                              // walk instances: a scale buffer is [factors | reciprocals (pair-interleaved,
                                                         // kernels.h walkPairIndex)], this many doubles apart
    size_t statePairOff = 0;                             // walk instances: a tip's pair-interleaved states follow its plain ones, this many bytes on
    // walk instances: position of pattern p in the pair-interleaved arrays (tip states, reciprocal scale factors).  They are
    // laid out partition by partition, each padded to whole blocks of 128 patterns (kernels.h WalkSeg), so that the assembly
    // loop runs whatever the caller's partition boundaries are; one partition: walkPairIndex(p).
    std::vector<unsigned> pairPos; size_t pairLen = 0; std::vector<int> padStart; unsigned* dPairPos = nullptr;
    char* matStream = nullptr; size_t matStreamBytes = 0;   // walk instances: the matrix stream of the program being run (k_gatherMatrices)
    uint8_t* dummyTips = nullptr; double* onesScale = nullptr;   // walk instances: all-missing states / all-one factors for the operands a
                                                                 // micro-operation does not use (the assembly loop loads them unconditionally)
    long statFastWalks = 0;
    // what runPlan derived from a cached plan (planner.h plannedTag): the device program with its addresses resolved
    struct Resolved {
        long tag = 0, epoch = -1;
        std::vector<mi355::WalkOp> w; std::vector<mi355::WalkSeg> segs; std::vector<int> deps;
        int maxRange = 0, sinks = 0;                     // (sinks: slices no other slice waits for — one: the whole program leads to its last slice)
        int leaves = 0;                                  // > 0: the device program is laid out for a launch on tickets — its first `leaves` slices wait for nothing
        std::vector<int> sumRows, wroteScale;            // write-mode programs: the device slices that leave their product of factors behind, the scale buffers the program writes (Instance::lastSums)
        std::vector<const double*> cm; long fused = 0;   // fused cherries (kernels.h WK_CHERRY): per device micro-operation the cherry's two branch matrices (empty: none fused), their number
        long memReads = 0, tipReads = 0, scaleReads = 0, scaleWrites = 0, stored = 0;
        char* dProg = nullptr; size_t dProgBytes = 0; bool dProgValid = false;    // the packed program, resident on the device
        std::vector<int> folds;                          // folded reciprocal vectors the program reads (Instance::folds)
        long foldEpoch = -1;                             // scaleWriteEpoch those vectors were last checked against
        long noFoldTag = 0;                              // the plan whose folds left the safe range: resolved with per-node factors
    } resolved[8];                                       // (as many as the planner's cache has ways: planner.h CACHE_WAYS)
    // Read mode, 4 states: a node that is not stored is seen by nobody but its parent, and a partial is linear in each child — so
    // the reciprocal scale factors of the unstored nodes below a stored one are applied ONCE, at that node, as one vector: the
    // entry-wise product of their reciprocal halves (a "fold"; built by k_foldReciprocals whenever a scale buffer has been written
    // since — scaleWriteEpoch — and kept by member list).  A cached full-evaluation program then reads one scale vector per STORED
    // node instead of one per node (config A: 78 + slice roots instead of 999; 0.8 of the evaluation's 2.2 GB).  The per-node
    // buffers stay what getLogScaleFactors, accumulateScaleFactors, partial updates and the gradient pass expect.  A fold whose
    // largest product leaves [1, 1e100] is refused and its plan falls back to per-node factors.  BEAGLE_MI355_NO_SCALE_FOLD=1 at
    // creation switches folding off: every node is then rounded the same way on every evaluation path (bitwise-equal partial
    // updates and full evaluations); with folding the paths agree to rounding (1e-15 relative).
    struct FoldVec { std::vector<int> members; double* recip = nullptr; long builtEpoch = -1; bool bad = false; };
    std::vector<FoldVec> folds;
    std::vector<std::pair<size_t, int>> foldIndex;       // (hash of the member list, index into folds), sorted by hash
    std::vector<double*> foldFree;                       // vectors of a dropped cache generation, reused
    long scaleWriteEpoch = 0;                            // bumped by everything that writes (or re-lays) a per-node scale buffer
    bool foldScales = true;
    unsigned long long* foldWorst = nullptr; size_t foldWorstCount = 0;      // device: k_foldReciprocals' range check
    long resolveEpoch = 0;                               // bumped when pattern ranges change
    bool fastWalk = true;                                // BEAGLE_MI355_NO_FAST_WALK=1 at creation: k_walk4 only (A/B runs, tests)
    // 4 states: a pre-order operation list is HELD BACK (engine_preorder.cpp): the chain that evaluates gradients wants the
    // edge-derivative sums that follow it, not the pre-order partials, and those sums can be had without writing a single
    // pre-order partial (kernels_preorder4.hip k_preWalk4).  The list stays held — as a closure over what it reads — until a
    // call reads something it writes or writes something it reads; then it runs first (GET_INSTANCE -> executeHeldPre).
    // Calls that provably do neither leave it alone (GET_INSTANCE_KEEP_PENDING + heldTouches* below), and a newer list that
    // rewrites everything it would have written replaces it unexecuted.
    struct HeldPreNode { int par, preA, preB, postA, postB, matA, matB, level, jobA, jobB, size; };
    struct HeldPreList {
        bool held = false;
        std::vector<int> ops;                            // as the caller wrote it
        std::vector<HeldPreNode> nodes;                  // one per parent: its two operations together
        std::vector<int> order;                          // depth-first, smaller subtree first (the walk's program order)
        std::vector<unsigned> walkFlags;                 // per entry of `order`: PW_* source / continuation bits
        std::vector<int> segStart, segRoot;              // the walk's segments: entries segStart[s] .. segStart[s + 1] of `order`, headed by job segRoot[s]
        int maxLevel = 0, holdSlots = 0, rootBuf = -1;
        std::vector<char> readsMatrix, readsBuf, writesBuf;   // by matrix / partials-buffer index
    } heldPre;
    double* preRootCopy = nullptr;                       // the held list's own copy of its root's pre-order partial
    uint8_t* preDummyStates = nullptr;                   // [P] "missing": what a descriptor's unused tip pointer points at
    void* dPreProg = nullptr; size_t dPreProgBytes = 0;  // the walk's program on the device (grow-only)
    // which scale buffer a partials buffer's last operation divided it by (-1: none / unknown), and that buffer's version then: the
    // pre-order walk multiplies a step into an internal node by the reciprocal of exactly that factor (kernels_preorder4.hip) and
    // is refused when the scale buffer has been written since (beagleUpdatePartials, the scale-factor calls)
    // Kept only once a pre-order list has arrived (trackScales; everything written before that is "unknown", -2, and the first
    // gradient of a chain takes the path that forms every edge's denominator itself).
    std::vector<int> scaleOfPartial; std::vector<unsigned> scaleVersionAtWrite, scaleVersion;
    bool trackScales = false;
    long statFusedGradients = 0, statPreLists = 0, statWalkedGradients = 0, statLateLists = 0;
    int storeAllEvaluations = 0;                         // > 0: a gradient chain is running — its post-order passes store what the pre-order
                                                         // pass will read: every node, or (gradientVirtual) every node but the short
                                                         // definitions the pre-order walk re-evaluates from the tips itself
    // 4 states: a gradient chain's post-order passes keep definitions of up to GRADIENT_VIRT_STEPS steps (tip-tip nodes, and those
    // under one more tip — half the nodes of a coalescent tree) instead of storing every node, and k_preWalk4 re-evaluates them where
    // it needs them (engine_preorder.cpp walkableDefinition).  BEAGLE_MI355_GRADIENT_VIRTUAL at creation (engine_abi.cpp): 0 = every
    // node stored, as in round 4; 1 = THE DEFAULT (GRADIENT_VIRT_DEFAULT): nodes over two tips stay unstored and are evaluated inside
    // their parents' descriptors (a third of the bytes of both passes: 6.1 -> 5.3-5.5 ms per gradient at 1e5 patterns); 2 = also such a
    // node under one more tip, by descriptors of their own (half the bytes, slower: a descriptor costs a stage whatever it computes).
    // (the initialisers below are overwritten at creation)
    bool gradientVirtual = false; int gradientVirtualSteps = GRADIENT_VIRT_STEPS;
    void* edgeScratch = nullptr; size_t edgeScratchBytes = 0;    // per-64-pattern derivative sums of the edges of one call (grow-only)
    bool preWalk = true;                                 // BEAGLE_MI355_NO_PRE_WALK=1 at creation: always write the pre-order partials
    bool fuseGradient = true;                            // BEAGLE_MI355_NO_FUSED_GRADIENT=1 at creation: operation by operation (A/B runs)
    // 16..20 states: operation lists without write-mode rescaling run as the walk's programs on the T32 layout (kernels_mfma.hip
    // k_walkT32: same planner, same descriptors; engine_walk.cpp); everything 4-state-specific (`walk`) stays off
    bool walkT = false;
    // ... and its write-mode form (k_walkT32W1: lists that rescale in write mode stay on the walk; at most four categories, two hold slots;
    // BEAGLE_MI355_NO_T32_WRITE_WALK=1: they run level by level, as until round 6)
    bool walkTWrite = false;
    // 4-state partitioned instances: the top slices of the partitions finish calculateRootLogLikelihoodsByPartition inside the walk's launch
    // (kernels.h RootFusedParts; BEAGLE_MI355_NO_ROOT_PARTS_FUSION=1: a launch of its own behind it, k_rootSite4WParts — the same bits)
    bool fuseRootParts = true;
    long statRootPartsFused = 0;                         // by-partition root calls that were finished inside the walk's launch
    // the tables the kernel reads and what each holds: a chain's evaluations alternate between a few (buffer flips: two programs, two
    // root slices), so four are kept and one is uploaded only when none of them matches
    static constexpr int ROOT_PARTS_TABLES = 4;
    mi355::RootFusedParts* rootPartsDev = nullptr;       // [ROOT_PARTS_TABLES]
    std::vector<char> rootPartsShadow[ROOT_PARTS_TABLES];
    int rootPartsNext = 0;
    bool fuseLaunches = true;                            // BEAGLE_MI355_NO_LAUNCH_FUSION=1: snapshot / gather and root site / final as separate launches (A/B runs)
    // A one-launch walk whose launch is held back until the next call: calculateRootLogLikelihoods on the result of one of its
    // slices launches it WITH that slice finishing the evaluation (no root kernel, no read-back of the root's partials); any other
    // call launches it as it is (live()).  Only full-range walks of unpartitioned instances outside a timer bracket are held.
    struct PendingWalk {
        bool valid = false;
        const mi355::WalkOp* prog = nullptr; const mi355::WalkSeg* segs = nullptr; const int* deps = nullptr;
        int nSegs = 0, range = 0, flagStride = 0; unsigned epoch = 0;
        int leaves = 0;                                  // > 0: launch on tickets, that many rows
        unsigned cherryOff = 0;                          // byte offset of the matrix stream's cherry region (0: the program has no fused cherries)
        std::vector<int> finalStore;                     // per device slice: the buffer its last micro-operation stores (-1: none)
        std::vector<int> finalPart;                      // ... and its partition; sinkRows: the slices nothing waits for (a partitioned
        std::vector<int> sinkRows;                       // instance: one per partition in the list — their epilogues finish the partitions' roots)
    } pendingWalk;
    std::vector<int> snapSourceOf;                       // runPlan's scratch: matrix slot -> the slot its snapshot is being taken from in this plan (-1 between calls)
    bool deferWalk = true;                               // BEAGLE_MI355_NO_ROOT_FUSION=1: never hold a launch back
    bool copyKeepsWalk = false;                          // (set around an upload the held walk does not read: engine_instance.cpp queueCopy)
    long statRootFused = 0;
    long statFoldedVectors = 0, statFoldBuilds = 0;      // read-mode programs: folded reciprocal vectors in use / (re)builds of them (engine_walk.cpp)
    unsigned* rootCounter = nullptr;                     // device word of k_rootSite's last-workgroup sum (kernels.hip)
    double* cherryTables = nullptr; size_t cherryTableBytes = 0;   // 21..64 states: column tables of a list's virtual cherries (grow-only)
    int holdSlots = 3;                                   // what the planner was given
    bool eigenComplex = false;                           // created with BEAGLE_FLAG_EIGEN_COMPLEX: eigenvalue arrays are [S real parts | S imaginary parts]
    bool strictWaits = true;                             // a stage's wait does not count on the previous stage's stores retiring behind its
                                                         // loads (runPlan); BEAGLE_MI355_STRICT_WAITS=0 at creation: it does (1 % faster)
    bool virt = false;                                   // some partials buffers may be virtual (walk instances; T32 instances: cherries)
    bool cherry = false;                                 // T32 instance with <= 20 states: tip-tip nodes are not stored (kernels.h CherryDesc)
    long statCherries = 0;
    double hostPlanUs = 0, hostRunUs = 0, hostPrepUs = 0, hostPlanHitUs = 0, hostRunHitUs = 0; long hostCalls = 0, hostHits = 0; bool hostTrace = false, lastResolveMiss = false;   // BEAGLE_MI355_HOST_TIMING=1: where updatePartials spends host time
    char* bigStage = nullptr; size_t bigStageBytes = 0;  // device staging for programs that do not fit the ring
    // read-back (getPartials): API-layout export buffers on the device and a pinned bounce buffer on the host
    double* exportDev[2] = {nullptr, nullptr}; double* exportHost[2] = {nullptr, nullptr}; size_t exportBytes = 0;   // two chunks in flight
    hipEvent_t exportEvent[2] = {nullptr, nullptr};
    long statMicroOps = 0, statStored = 0, statMemReads = 0, statTipReads = 0, statScaleReads = 0, statWalks = 0, statScaleWrites = 0;   // since the last timer reset
    hipStream_t stream = nullptr, ownStream = nullptr;
    int tipCount = 0, partialsCount = 0, compactCount = 0, S = 0, P = 0, eigenCount = 0, matrixCount = 0, C = 0, scaleCount = 0;
    size_t partialsBytes = 0;
    std::vector<double*> partials;
    std::vector<uint8_t*> tipStates;
    std::vector<void*> allocations;
    char* slabCur = nullptr; int slabLeft = 0;
    char* scaleSlabCur = nullptr; int scaleSlabLeft = 0;
    char* stateSlabCur = nullptr; int stateSlabLeft = 0;
    double* matrices = nullptr; double* eigen = nullptr; double* rates = nullptr; double* weights = nullptr;
    double* freqs = nullptr; double* patternWeights = nullptr; double* siteLogL = nullptr;
    std::vector<double*> scale; std::vector<char> scaleIsRaw;
    double* blockSums = nullptr; double* dResult = nullptr; double* hResult = nullptr; double* hResultDev = nullptr; unsigned long long resultSeq = 0;
    char* hRing = nullptr; char* dRing = nullptr; size_t ringHead = 0;
    // Small host arrays (model parameters, branch lengths, programs) do not go through a copy engine: they are staged in the
    // pinned ring, which the device maps (hRingDev), and queued here; ONE kernel (kernels.hip k_hostCopies) moves everything
    // queued so far right before the next kernel or copy of the stream (live()).  A copy-engine transfer per array costs a
    // dependent blit or SDMA packet each — 25 us of a 12 500-pattern evaluation's 190 (profiles/r04_experiments.txt).
    // BEAGLE_MI355_COPY_ENGINE_UPLOADS=1 at creation: hipMemcpyAsync per array, as before round 4 (A/B runs).
    struct PendingCopy { void* dst; size_t ringOff; size_t bytes; };
    std::vector<PendingCopy> pendingCopies;
    char* hRingDev = nullptr;
    bool kernelUploads = true;
    // one process per GPU, patterns sharded over the processes: the all-reduce of the root sum runs INSIDE the engine, on the
    // instance's stream right behind the reduction kernel, and its result reaches the host through the same mapped words as a
    // single-GPU sum (beagleMi355CommInit / beagleMi355CalculateRootLogLikelihoodsAllReduce)
    ncclComm_t comm = nullptr; int commRanks = 0;
    // BeagleTreeLikelihood reads the site log-likelihoods back after EVERY evaluation (BeagleTreeLikelihood.java:1050: P doubles, 800 KB at
    // the metric's size — through a pageable destination a quarter of a 4-state evaluation's step).  A caller that has done so after each
    // of the last two whole-alignment root sums gets the next one's site values sent to a pinned host buffer (hSites; the device writes it
    // through its mapping, a kernel right behind the root's: no copy engine, no staging) before it asks: beagleGetSiteLogLikelihoods is then
    // a wait for siteEvent — over by the time the call arrives — and one host copy.  Any later root sum of the instance makes the prefetched
    // values stale (sitePrefetched = false: rootEnqueue / the by-partition sums); a caller that stops asking stops being served
    // (siteReadStreak).  BEAGLE_MI355_NO_SITE_PREFETCH=1 at creation: always the stream-ordered download (A/B runs, tests).
    bool sitePrefetch = true, sitePrefetched = false, siteReadSinceRoot = true;
    int siteReadStreak = 0; long statSitePrefetched = 0;
    double* hSites = nullptr; double* hSitesDev = nullptr; hipEvent_t siteEvent = nullptr;
    ~Instance() { if (hSites) hipHostFree(hSites); if (siteEvent) hipEventDestroy(siteEvent); }      // (destroy() has drained the streams by then)
    // first error of a deferred operation; surfaces at the next call that observes results.  Atomic: the sharded handle's caller reads and
    // clears it (sharded.cpp takeAsyncError) while the shard's own worker thread may be setting it
    std::atomic<int> asyncError{0};
    // 4 states: every slice of a walk program in ONE launch (engine_walk.cpp runPlan): per (slice, pattern group) flag words the
    // workgroups signal and poll with the launch's epoch.  BEAGLE_MI355_NO_WALK_FUSION=1 at creation: one launch per wave of slices
    bool fuseWaves = true;
    unsigned* walkFlags = nullptr; size_t walkFlagBytes = 0; unsigned walkEpoch = 0;
    // ... or, when the slices of a program form a forest (every stored slice root has one reader: planner.h PlanSeg::next), with no
    // waiting at all: only the slices without dependencies get workgroups, and the workgroup that arrives last at a slice above runs it
    // (kernels_walk4.hip "tickets"; walkTickets: per (slice, pattern group) arrival counts, zero between launches, in the same
    // allocation behind the flags).  BEAGLE_MI355_NO_WALK_TICKETS=1 at creation: dependency flags always (A/B runs, tests)
    bool useTickets = true; unsigned* walkTickets = nullptr;
    // ... and such a launch, when the chip holds all of it at once and it has rows for every XCD, lays its rows out XCD by XCD (kernels_walk4.hip
    // launchWalk4Fast: small partitioned alignments).  BEAGLE_MI355_NO_XCD_MAP=1 at creation: always the plain 2-D grid (A/B runs)
    bool xcdAware = true;
    // 4 states, assembly loop: a node over two compact tips that is not stored, pays no scale factors and is consumed by the very next
    // micro-operation is not a micro-operation of the device program at all: its consumer evaluates it inside its own stage (kernels.h
    // WK_CHERRY; engine_walk.cpp runPlan).  Same arithmetic in the same order: bitwise the unfused program.  A third of a tree's nodes.
    // BEAGLE_MI355_NO_CHERRY_FUSION=1 at creation: every micro-operation of the plan is one of the device program (A/B runs, tests)
    bool fuseCherries = true; long statFused = 0;
    // ... and the loop's fetch loads tip states only for children that ARE compact tips (kernels.h WF_NOLOAD1 / 2; programs without
    // write-mode rescaling).  BEAGLE_MI355_NO_LOAD_SKIP=1 at creation: both tip-state loads in every fetch, as before round 6 (A/B runs)
    bool skipTipLoads = true;
    // Write-mode programs (4 states, one partition): every slice leaves the product of the factors it wrote per pattern — mantissa and binary
    // exponent, [slice row][pair position] — and an accumulateScaleFactors call over EXACTLY the scale buffers the last such program wrote (what
    // BEAST issues right behind it: BeagleTreeLikelihood.java:1013-1026) adds a few dozen logarithms per pattern instead of reading a factor per
    // node (kernels.hip k_accumulateSlices; config A, ALWAYS rescaling: 250 us and 0.8 GB per evaluation).  Anything else — another list, a
    // scale buffer written since (scaleWriteEpoch) — takes the general kernel.  BEAGLE_MI355_NO_SLICE_SUMS=1 at creation: always the general kernel.
    bool sliceSums = true;
    double* sliceMant = nullptr; int* sliceExp = nullptr; size_t sliceRows = 0;
    struct LastSums { bool valid = false; long epoch = -1, gen = 0; std::vector<int> rows; int nWritten = 0; } lastSums;
    std::vector<long> scaleGen, scaleSeen; long sliceGen = 0, seenCounter = 0, statSliceAccum = 0;
    long statTicketWalks = 0, statFlagWalks = 0, lastLaunchRows = 0, lastLaunchSlices = 0;      // (beagleMi355WalkLaunchInfo)
    // how long a workgroup of that launch polls before it computes what it waits for itself (kernels_walk4.hip: forward progress does
    // not rest on the dispatch order), in ticks of the device's 100 MHz wall clock: 20 ms — an evaluation of the largest alignment
    // this engine is measured on takes 0.6; BEAGLE_MI355_WALK_SPIN_US=<us> at creation (tests: 0 forces the self-serve path).
    // walkSelfServed: device word that counts the workgroups whose wait ran out (beagleMi355WalkHealth).
    unsigned long long walkSpinLimit = 2000000ull; unsigned* walkSelfServed = nullptr;
    int partitionCount = 1;
    std::vector<int> partStart, partEnd;
    // levelisation scratch
    std::vector<int> wStamp, wLevel, rStamp, rLevel, wOp; int stamp = 0;
    bool tiled = false; int ntile = 0;   // T32 partials layout (MFMA path)
    // pre-order on the T32 layout runs as two passes of the pruning kernel (see runPreOperations): scratch partials
    // buffers, an identity matrix and transposed-matrix slots behind the caller's matrices, an all-missing tip
    std::vector<double*> preScratch; uint8_t* preMissing = nullptr; int preIdentity = -1, preTransposed = -1;
    bool schedAlap = true;               // BEAGLE_MI355_SCHED=asap restores as-soon-as-possible levels
    int schedDfs = 0;                    // > 0 (BEAGLE_MI355_SCHED=dfs[:K]): launches of at most K operations in depth-first order (engine_levels.cpp)
    // kernel timer
    bool timing = false;
    int timingEvery = 1, timingTick = 0; long timedCalls = 0;      // every timingEvery-th updatePartials call is bracketed (beagleMi355KernelTimer)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events; size_t eventsUsed = 0;
    double timedMs = 0.0; long timedLaunches = 0, pendingLaunches = 0;
    size_t deviceBytes = 0;
    std::vector<double> shEigen, shFreqs, shWeights, shRates;      // host shadows of the small model arrays ...
    std::vector<char> okEigen, okFreqs, okWeights, okRates;         // ... valid flags per index
    std::string resourceName;
};

extern std::mutex g_mutex;
extern std::vector<Instance*> g_instances;
Instance* lookup(int h);

// ---- engine_instance.cpp: memory, staging, resources
int devAlloc(Instance* in, void** p, size_t bytes);
long stage(Instance* in, const void* src, size_t bytes, size_t reserve = 0);
int upload(Instance* in, void* dst, const void* src, size_t bytes);
int uploadTransient(Instance* in, const void* src, size_t bytes, void** dptr);
int download(Instance* in, void* dst, const void* src, size_t bytes);
int ensurePartials(Instance* in, int idx);
int ensureScale(Instance* in, int idx);
int ensureStates(Instance* in, int idx);
void destroy(Instance* in);
struct Resources {
    std::vector<std::string> names, descs;
    std::vector<BeagleResource> list;
    BeagleResourceList rl;
    int gpuCount = 0;
};
extern const long GPU_FLAGS;
Resources* resources();
void setPairLayout(Instance* in);
int executeHeldPre(Instance* in);
int holdPreList(Instance* in, const int* ops, int count);
bool supersedesHeld(Instance* in, const int* ops, int count);
int ensureWalkDummies(Instance* in);
// the instance's stream, with everything queued in pendingCopies enqueued on it first: what every launch, copy and
// synchronisation of the engine passes as its stream (only flushUploads itself and the stream setters touch in->stream)
int flushUploads(Instance* in);
// ... and a walk launch that is being held back for the root call (PendingWalk, engine_walk.cpp) launched behind them
int flushWalk(Instance* in, const mi355::RootFused* root = nullptr);
inline hipStream_t live(Instance* in) {
    if (!in->pendingCopies.empty()) flushUploads(in);
    if (in->pendingWalk.valid) flushWalk(in);
    return in->stream;
}
// queue `bytes` already staged at ring offset `off` for device address dst (or copy them now: kernelUploads off)
int queueCopy(Instance* in, void* dst, size_t off, size_t bytes);

// (a held-back pre-order list — Instance::heldPre — runs before anything else touches the instance; the few calls that cannot
// interact with it use GET_INSTANCE_KEEP_PENDING)
#define GET_INSTANCE_KEEP_PENDING(h)                             \
    Instance* in = lookup(h);                                    \
    if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;         \
    if (hipSetDevice(in->device) != hipSuccess) return BEAGLE_ERROR_GENERAL;
#define GET_INSTANCE(h)                                          \
    GET_INSTANCE_KEEP_PENDING(h)                                 \
    if (in->heldPre.held) { const int rcPending__ = executeHeldPre(in); if (rcPending__) return rcPending__; }

inline bool badIndex(int i, int n) { return i < 0 || i >= n; }
// is this updatePartials call one the kernel timer brackets with an event pair?  (an event costs a barrier packet on the stream:
// 6 us of an evaluation each — a sampled timer keeps that out of most of a timed region)
inline bool timeThisCall(Instance* in) {
    if (!in->timing) return false;
    if (++in->timingTick < in->timingEvery) return false;
    in->timingTick = 0;
    in->timedCalls++;
    return true;
}

// ---- the pattern walk (4 states) ------------------------------------------------------------------------------------
// A partials buffer is "virtual" when its content is DEFINED instead of stored (planner.h VirtDef): a few steps over
// compact tips, private snapshots of every branch matrix in the subtree (kept behind the caller's matrices) and each
// node's scale buffer.  Nothing is written to HBM for such a buffer; the walk recomputes it in registers where a parent
// needs it, bitwise as the ordinary operation would have.  The definition is self-contained: it never refers to other
// partials buffers or to the caller's matrix buffers, so buffer flips and matrix updates cannot invalidate it.  What CAN
// change a defining input — new tip states, a write to one of its scale buffers — and everything that needs the real
// data — getPartials, use as root, the pre-order kernels — first calls materializeList, which runs the definition with
// a store.
static_assert(mi355::PK_MEM == mi355::WK_MEM && mi355::PK_TIPS == mi355::WK_TIPS && mi355::PK_ACC == mi355::WK_ACC &&
              mi355::PK_H0 == mi355::WK_H0 && mi355::PK_H1 == mi355::WK_H1 && mi355::PK_H2 == mi355::WK_H2, "planner kinds = kernel kinds");
static_assert(mi355::PS_NONE == mi355::WS_NONE && mi355::PS_READ == mi355::WS_READ && mi355::PS_WRITE == mi355::WS_WRITE, "scale modes");

inline bool isVirt(const Instance* in, int X) { return in->virt && in->planner.isVirtual(X); }
inline void clearVirtual(Instance* in, int X) { if (in->virt) in->planner.clearVirtual(X); }
inline bool isCompactTip(const Instance* in, int X) { return in->tipStates[X] && X < in->tipCount; }
// what buffer X holds changed: compact tip states (on), or something else — in which case it is no uploaded-partials leaf
// either until setLeaf says so (planner.h leafPartials)
inline void setCompact(Instance* in, int X, bool on) { in->planner.setCompactTip(X, on); in->planner.setLeafPartials(X, false); }
inline void setLeaf(Instance* in, int X) { if (X < in->tipCount) in->planner.setLeafPartials(X, true); }

// ---- engine_walk.cpp
int runPlan(Instance* in, const mi355::Plan& plan, long planTag = 0, hipEvent_t recordBeforeWalk = nullptr);
inline void scalesWritten(Instance* in) { in->scaleWriteEpoch++; }       // (Instance::folds)
void forgetFolds(Instance* in);                                          // the layout of the scale buffers changes
int materializeList(Instance* in, const std::vector<int>& xs);
int materializeVirtual(Instance* in, int X);
int materializeScaleUsers(Instance* in, int scaleIdx);
int materializeTipUsers(Instance* in, int tip);
int walkChunkOps(const Instance* in, int opCount);
int runOperationsWalk(Instance* in, const int* ops, int count, int tuple, int globalCum);
// ---- engine_levels.cpp
int materializeCherries(Instance* in, const std::vector<int>& xs);
int runOperationsLevels(Instance* in, const int* ops, int count, int tuple, int globalCum);
int foldCumulative(Instance* in, const int* ops, int count, int tuple, int globalCum);
int runOperations(Instance* in, const int* ops, int count, int tuple, int globalCum);
// ---- engine_preorder.cpp
int ensurePreScratch(Instance* in);
int ensureEdgeScratch(Instance* in, size_t bytes);
// does a call that writes matrix m / reads or writes partials buffer b have to wait for the held pre-order list?
inline bool heldReadsMatrix(const Instance* in, int m) { return in->heldPre.held && m >= 0 && m < (int)in->heldPre.readsMatrix.size() && in->heldPre.readsMatrix[m]; }
inline bool heldWrites(const Instance* in, int b) { return in->heldPre.held && b >= 0 && b < (int)in->heldPre.writesBuf.size() && in->heldPre.writesBuf[b]; }
inline bool heldTouches(const Instance* in, int b) { return in->heldPre.held && b >= 0 && b < (int)in->heldPre.writesBuf.size() && (in->heldPre.writesBuf[b] || in->heldPre.readsBuf[b]); }
int runPreOperations(Instance* in, const int* ops, int count, int globalCum, bool mayHold = false, int tuple = BEAGLE_OP_COUNT);
int executeHeldPre(Instance* in);
int edgeDifferentials(Instance* in, const int* postIdx, const int* preIdx, const int* dIdx, int wIdx, int count,
                      double* outDerivatives, double* outSum, double* outSumSquared);
int crossProducts(Instance* in, const int* postIdx, const int* preIdx, int rateIdx, int wIdx, const double* lengths, int count, double* outSum);

}  // namespace eng
}  // namespace mi355
