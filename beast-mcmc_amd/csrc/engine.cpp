// engine.cpp — host side of the MI355X engine: instance table, HBM buffer management, the C ABI of
// include/beagle_mi355.h.  All arithmetic is in kernels.hip; this file validates indices, resolves
// buffer indices to device pointers, levelises operation lists and enqueues kernels on the
// instance's HIP stream.  Results are only observed at calculateRootLogLikelihoods / get*, so every
// other call returns as soon as its work is enqueued (SURVEY 8b "Threading").
//
// There is no CPU path in this library: with no visible MI355X beagleCreateInstance returns
// BEAGLE_ERROR_NO_RESOURCE.
#include <hip/hip_runtime.h>
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

using mi355::OpDesc;
using mi355::shardedStates;
using mi355::shardedCategories;

namespace {

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
        std::vector<mi355::WalkOp> w; std::vector<mi355::WalkSeg> segs;
        int maxRange = 0;
        long memReads = 0, tipReads = 0, scaleReads = 0, scaleWrites = 0, stored = 0;
        char* dProg = nullptr; size_t dProgBytes = 0; bool dProgValid = false;    // the packed program, resident on the device
    } resolved[4];
    long resolveEpoch = 0;                               // bumped when pattern ranges change
    bool fastWalk = true;                                // BEAGLE_MI355_NO_FAST_WALK=1 at creation: k_walk4 only (A/B runs, tests)
    bool eigenComplex = false;                           // created with BEAGLE_FLAG_EIGEN_COMPLEX: eigenvalue arrays are [S real parts | S imaginary parts]
    bool strictWaits = true;                             // a stage's wait does not count on the previous stage's stores retiring behind its
                                                         // loads (runPlan); BEAGLE_MI355_STRICT_WAITS=0 at creation: it does (1 % faster)
    bool virt = false;                                   // some partials buffers may be virtual (walk instances; T32 instances: cherries)
    bool cherry = false;                                 // T32 instance with <= 20 states: tip-tip nodes are not stored (kernels.h CherryDesc)
    long statCherries = 0;
    double hostPlanUs = 0, hostRunUs = 0, hostPrepUs = 0, hostPlanHitUs = 0, hostRunHitUs = 0; long hostCalls = 0, hostHits = 0;   // BEAGLE_MI355_HOST_TIMING=1: where updatePartials spends host time
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
    int partitionCount = 1;
    std::vector<int> partStart, partEnd;
    // levelisation scratch
    std::vector<int> wStamp, wLevel, rStamp, rLevel, wOp; int stamp = 0;
    bool tiled = false; int ntile = 0;   // T32 partials layout (MFMA path)
    // pre-order on the T32 layout runs as two passes of the pruning kernel (see runPreOperations): scratch partials
    // buffers, an identity matrix and transposed-matrix slots behind the caller's matrices, an all-missing tip
    std::vector<double*> preScratch; uint8_t* preMissing = nullptr; int preIdentity = -1, preTransposed = -1;
    bool schedAlap = true;               // BEAGLE_MI355_SCHED=asap restores as-soon-as-possible levels
    // kernel timer
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events; size_t eventsUsed = 0;
    double timedMs = 0.0; long timedLaunches = 0, pendingLaunches = 0;
    size_t deviceBytes = 0;
    std::vector<double> shEigen, shFreqs, shWeights, shRates;      // host shadows of the small model arrays ...
    std::vector<char> okEigen, okFreqs, okWeights, okRates;         // ... valid flags per index
    std::string resourceName;
};

std::mutex g_mutex;
std::vector<Instance*> g_instances;

Instance* lookup(int h) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (h < 0 || h >= (int)g_instances.size()) return nullptr;
    return g_instances[h];
}

int devAlloc(Instance* in, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    in->allocations.push_back(*p);
    in->deviceBytes += bytes;
    return 0;
}

// Stage `bytes` of host data into the pinned ring; returns the ring offset (or <0).  Wrapping first
// drains the stream, so a region is never overwritten while a copy from it is still in flight.
long stage(Instance* in, const void* src, size_t bytes, size_t reserve = 0) {
    const size_t need = (std::max(bytes, reserve) + 255) & ~(size_t)255;
    if (need > RING_BYTES) return -1;
    if (in->ringHead + need > RING_BYTES) {
        if (hipStreamSynchronize(in->stream) != hipSuccess) return -1;
        in->ringHead = 0;
    }
    const size_t off = in->ringHead;
    memcpy(in->hRing + off, src, bytes);
    in->ringHead += need;
    return (long)off;
}

// host array -> persistent device location, asynchronously when it fits the ring
int upload(Instance* in, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return 0;
    if (bytes <= RING_BYTES / 4) {
        long off = stage(in, src, bytes);
        if (off < 0) return BEAGLE_ERROR_GENERAL;
        HIP_TRY(hipMemcpyAsync(dst, in->hRing + off, bytes, hipMemcpyHostToDevice, in->stream));
        return 0;
    }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, in->stream));
    HIP_TRY(hipStreamSynchronize(in->stream));
    return 0;
}

// host array -> the device mirror of the ring (transient kernel arguments: op descriptors, index lists)
int uploadTransient(Instance* in, const void* src, size_t bytes, void** dptr) {
    long off = stage(in, src, bytes);
    if (off < 0) return BEAGLE_ERROR_GENERAL;
    HIP_TRY(hipMemcpyAsync(in->dRing + off, in->hRing + off, bytes, hipMemcpyHostToDevice, in->stream));
    *dptr = in->dRing + off;
    return 0;
}

int download(Instance* in, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, in->stream));
    HIP_TRY(hipStreamSynchronize(in->stream));
    in->ringHead = 0;   // everything staged so far has been consumed
    return 0;
}

int ensurePartials(Instance* in, int idx) {
    if (in->partials[idx]) return 0;
    if (in->slabLeft == 0) {
        int remaining = 0;
        for (double* p : in->partials) if (!p) remaining++;
        const int n = std::min(remaining, SLAB_BUFFERS);
        void* slab = nullptr;
        int rc = devAlloc(in, &slab, in->partialsBytes * n);
        if (rc) return rc;
        in->slabCur = (char*)slab; in->slabLeft = n;
    }
    in->partials[idx] = (double*)in->slabCur;
    in->slabCur += in->partialsBytes; in->slabLeft--;
    return 0;
}

int ensureScale(Instance* in, int idx) {
    if (in->scale[idx]) return 0;
    // walk instances keep [factors | reciprocals] so that read mode never divides (kernels_walk4.hip)
    const size_t bytes = in->walk ? 2 * in->scaleStride * sizeof(double) : (((size_t)in->P * sizeof(double) + 255) & ~(size_t)255);
    if (in->scaleSlabLeft == 0) {
        int remaining = 0;
        for (double* p : in->scale) if (!p) remaining++;
        const int n = std::min(remaining, 256);
        void* slab = nullptr;
        int rc = devAlloc(in, &slab, bytes * n);
        if (rc) return rc;
        if (hipMemsetAsync(slab, 0, bytes * n, in->stream) != hipSuccess) return BEAGLE_ERROR_GENERAL;
        in->scaleSlabCur = (char*)slab; in->scaleSlabLeft = n;
    }
    in->scale[idx] = (double*)in->scaleSlabCur;
    in->scaleSlabCur += bytes; in->scaleSlabLeft--;
    in->scaleIsRaw[idx] = 0;
    return 0;
}

int ensureStates(Instance* in, int idx) {
    if (in->tipStates[idx]) return 0;
    const size_t plain = ((size_t)in->P + 2 + 255) & ~(size_t)255;
    const size_t bytes = in->walk ? plain + ((in->pairLen + 255) & ~(size_t)255) : plain;      // walk instances: [plain | pair-interleaved] (the latter is what the walk reads)
    in->statePairOff = plain;
    if (in->stateSlabLeft == 0) {
        const int n = std::max(1, std::min(in->compactCount, 1024));
        void* slab = nullptr;
        int rc = devAlloc(in, &slab, bytes * n);
        if (rc) return rc;
        in->stateSlabCur = (char*)slab; in->stateSlabLeft = n;
    }
    in->tipStates[idx] = (uint8_t*)in->stateSlabCur;
    in->stateSlabCur += bytes; in->stateSlabLeft--;
    in->resolveEpoch++;                       // a kept device program may still point at the slot this tip had before (Instance::Resolved)
    return 0;
}

void destroy(Instance* in) {
    hipSetDevice(in->device);
    if (in->hostCalls && getenv("BEAGLE_MI355_HOST_TIMING"))
        fprintf(stderr, "[mi355] updatePartials host time per call over %ld calls: checks+materialise %.1f us, planner %.1f us, resolve+upload+launch %.1f us; %ld plans from the cache: planner %.1f us, resolve+upload+launch %.1f us\n",
                in->hostCalls, in->hostPrepUs / in->hostCalls, in->hostPlanUs / in->hostCalls, in->hostRunUs / in->hostCalls, in->planner.cacheHits,
                in->hostHits ? in->hostPlanHitUs / in->hostHits : 0.0, in->hostHits ? in->hostRunHitUs / in->hostHits : 0.0);
    if (in->ownStream) hipStreamSynchronize(in->ownStream);
    if (in->stream && in->stream != in->ownStream) hipStreamSynchronize(in->stream);
    for (void* p : in->allocations) hipFree(p);
    if (in->bigStage) hipFree(in->bigStage);
    if (in->matStream) hipFree(in->matStream);
    for (auto& r : in->resolved) if (r.dProg) hipFree(r.dProg);
    for (int k = 0; k < 2; k++) {
        if (in->exportDev[k]) hipFree(in->exportDev[k]);
        if (in->exportHost[k]) hipHostFree(in->exportHost[k]);
        if (in->exportEvent[k]) hipEventDestroy(in->exportEvent[k]);
    }
    if (in->hRing) hipHostFree(in->hRing);
    if (in->hResult) hipHostFree(in->hResult);
    for (auto& ev : in->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    if (in->ownStream) hipStreamDestroy(in->ownStream);
    delete in;
}

struct Resources {
    std::vector<std::string> names, descs;
    std::vector<BeagleResource> list;
    BeagleResourceList rl;
    int gpuCount = 0;
};
Resources* g_resources = nullptr;

const long GPU_FLAGS = BEAGLE_FLAG_PRECISION_DOUBLE | BEAGLE_FLAG_COMPUTATION_SYNCH | BEAGLE_FLAG_EIGEN_REAL | BEAGLE_FLAG_EIGEN_COMPLEX |
                       BEAGLE_FLAG_SCALING_MANUAL | BEAGLE_FLAG_SCALING_ALWAYS | BEAGLE_FLAG_SCALING_DYNAMIC |
                       BEAGLE_FLAG_SCALERS_RAW | BEAGLE_FLAG_VECTOR_NONE | BEAGLE_FLAG_THREADING_NONE |
                       BEAGLE_FLAG_PROCESSOR_GPU | BEAGLE_FLAG_PARALLELOPS_GRID;

Resources* resources() {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_resources) return g_resources;
    Resources* r = new Resources();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    r->gpuCount = n;
    // resource 0 is "the CPU" by BEAST convention (BeagleTreeLikelihood.java:90-92); this library has no
    // CPU implementation, the entry only keeps the numbering of the GPUs at 1..G.
    r->names.push_back("CPU"); r->descs.push_back("not provided by this library (MI355X engine only)");
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t prop;
        char buf[256];
        if (hipGetDeviceProperties(&prop, d) == hipSuccess) {
            snprintf(buf, sizeof(buf), "Global memory (MB): %zu | Compute units: %d | Arch: %s",
                     (size_t)(prop.totalGlobalMem >> 20), prop.multiProcessorCount, prop.gcnArchName);
            r->names.push_back(prop.name);
        } else {
            snprintf(buf, sizeof(buf), "device %d", d);
            r->names.push_back("AMD GPU");
        }
        r->descs.push_back(buf);
    }
    if (n >= 1) {      // resource G+1: every GPU of the node behind one instance, patterns sharded (sharded.cpp)
        char buf[256];
        const int shards = mi355::shardedDeviceCountOverride() > 0 ? mi355::shardedDeviceCountOverride() : n;
        snprintf(buf, sizeof(buf), "%d pattern shards over %d GPU(s) | one RCCL all-reduce of the log-likelihood per evaluation", shards, n);
        r->names.push_back("all GPUs (pattern-sharded)");
        r->descs.push_back(buf);
    }
    for (size_t i = 0; i < r->names.size(); i++) {
        BeagleResource br;
        br.name = (char*)r->names[i].c_str(); br.description = (char*)r->descs[i].c_str();
        br.supportFlags = i == 0 ? 0 : GPU_FLAGS; br.requiredFlags = 0;
        r->list.push_back(br);
    }
    r->rl.list = r->list.data(); r->rl.length = (int)r->list.size();
    g_resources = r;
    return r;
}

#define GET_INSTANCE(h)                                          \
    Instance* in = lookup(h);                                    \
    if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;         \
    if (hipSetDevice(in->device) != hipSuccess) return BEAGLE_ERROR_GENERAL;

inline bool badIndex(int i, int n) { return i < 0 || i >= n; }

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

// The pair-interleaved layout for the instance's current partitions (Instance::pairPos), and the scale-buffer stride that
// holds either half ([factors, plain | reciprocals, pair-interleaved]).
void setPairLayout(Instance* in) {
    const int K = in->partitionCount;
    in->padStart.assign(K, 0);
    in->pairPos.assign((size_t)in->P, 0u);
    size_t at = 0;
    // partitions in pattern order (they are contiguous ranges; an empty one takes no room)
    std::vector<int> order(K);
    for (int k = 0; k < K; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return in->partStart[a] < in->partStart[b]; });
    for (int k : order) {
        in->padStart[k] = (int)at;
        for (int p = in->partStart[k]; p < in->partEnd[k]; p++) in->pairPos[p] = (unsigned)(at + mi355::walkPairIndex((size_t)(p - in->partStart[k])));
        at += ((size_t)(in->partEnd[k] - in->partStart[k]) + 127) & ~(size_t)127;
    }
    in->pairLen = std::max<size_t>(at, 128);
    in->scaleStride = (std::max<size_t>((size_t)in->P, in->pairLen) + 2 + 127) & ~(size_t)127;
}

int ensureWalkDummies(Instance* in) {
    if (in->dummyTips) return 0;
    const size_t tipBytes = in->pairLen + 256, scaleBytes = in->scaleStride * sizeof(double);
    void* p = nullptr;
    int rc = devAlloc(in, &p, ((tipBytes + 255) & ~(size_t)255) + scaleBytes); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(p, in->S, tipBytes, in->stream));
    double* ones = (double*)((char*)p + ((tipBytes + 255) & ~(size_t)255));
    mi355::launchFill(in->stream, ones, 1.0, 0, (int)in->scaleStride);
    HIP_TRY(hipGetLastError());
    in->dummyTips = (uint8_t*)p; in->onesScale = ones;
    return 0;
}

// Resolve a planned program to device addresses, upload it (ONE host-to-device copy: snapshot pairs, segments and
// micro-operations travel together) and enqueue the snapshot copies and the walk.
int runPlan(Instance* in, const mi355::Plan& plan, long planTag = 0, hipEvent_t recordBeforeWalk = nullptr) {
    const size_t n = plan.prog.size();
    if (n == 0) {                                  // nothing to compute (every destination became virtual): the definitions'
        if (plan.snapPairs.empty()) return 0;      // matrix snapshots still have to be taken
        void* dPairs = nullptr;
        int rc = uploadTransient(in, plan.snapPairs.data(), plan.snapPairs.size() * sizeof(int), &dPairs); if (rc) return rc;
        mi355::launchSnapshotMatrices(in->stream, in->matrices, (const int*)dPairs, (int)(plan.snapPairs.size() / 2), in->C * in->S * in->S);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // Device program: per segment its micro-operations, a no-op when their number is odd, and two more no-ops the
    // kernel's descriptor prefetch may read (kernels.h WalkSeg).  For a plan that came out of the planner's cache the
    // resolved program is kept as well: buffer addresses never change once a buffer exists.
    static const int ablate = getenv("BEAGLE_MI355_ABLATE") ? atoi(getenv("BEAGLE_MI355_ABLATE")) : 0;
    Instance::Resolved* slot = planTag && !ablate ? &in->resolved[planTag & 3] : nullptr;
    const bool reuse = slot && slot->tag == planTag && slot->epoch == in->resolveEpoch;
    std::vector<mi355::WalkOp>& w = slot ? slot->w : in->walkOps;
    std::vector<mi355::WalkSeg> segsLocal;
    std::vector<mi355::WalkSeg>& segs = slot ? slot->segs : segsLocal;
    int maxRange = 0;
    if (reuse) {
        maxRange = slot->maxRange;
        in->statMemReads += slot->memReads; in->statTipReads += slot->tipReads; in->statScaleReads += slot->scaleReads;
        in->statScaleWrites += slot->scaleWrites; in->statStored += slot->stored;
    } else {
    const long s0[5] = {in->statMemReads, in->statTipReads, in->statScaleReads, in->statScaleWrites, in->statStored};
    if (slot) { slot->tag = 0; slot->dProgValid = false; }
    w.clear();
    w.reserve(n + 3 * plan.segs.size());
    segs.assign(plan.segs.size(), mi355::WalkSeg());
    const size_t matStride = (size_t)in->C * 16;
    { int rc = ensureWalkDummies(in); if (rc) return rc; }
    mi355::WalkOp nop;
    memset(&nop, 0, sizeof(nop));
    nop.m1 = in->matrices; nop.m2 = in->matrices;
    nop.src1 = in->dummyTips; nop.src2 = in->dummyTips; nop.scale = in->onesScale;
    nop.flags = (unsigned)((mi355::WK_TIPS << 5) | (mi355::WK_TIPS << 8));         // loads nothing, stores nothing
    for (size_t si = 0; si < plan.segs.size(); si++) {
        const mi355::PlanSeg& ps = plan.segs[si];
        segs[si].progStart = (int)w.size();
        for (int i = ps.progStart; i < ps.progStart + ps.progCount; i++) {
            mi355::MicroOp m = plan.prog[i];
            // The kernels request a first child's partials one stage early — before the previous micro-operation's store
            // is issued (kernels_walk4.hip WALK_STAGE).  The planner never emits that sequence (tests/native/plan_check.cpp
            // checks every program for it); should one arrive anyway, a no-op in between restores the distance.
            if (i > ps.progStart && m.k1 == mi355::PK_MEM && plan.prog[i - 1].storeBuf == m.a1) {
                if (m.k2 == mi355::PK_ACC) { m.k2 = mi355::PK_MEM; m.a2 = m.a1; }      // the no-op overwrites ACC; the value is in memory as well
                w.push_back(nop);
            }
            mi355::WalkOp d;
            memset(&d, 0, sizeof(d));
            d.src1 = in->dummyTips; d.src2 = in->dummyTips; d.scale = in->onesScale;     // unused operands stay readable (kernels.h launchWalk4Fast)
            if (m.k1 == mi355::PK_MEM) { d.src1 = in->partials[m.a1]; if (!d.src1 || isCompactTip(in, m.a1)) return BEAGLE_ERROR_OUT_OF_RANGE; in->statMemReads++; }
            else if (m.k1 == mi355::PK_TIPS) { if (!in->tipStates[m.a1]) return BEAGLE_ERROR_OUT_OF_RANGE; d.src1 = in->tipStates[m.a1] + in->statePairOff; in->statTipReads++; }
            if (m.k2 == mi355::PK_MEM) { d.src2 = in->partials[m.a2]; if (!d.src2 || isCompactTip(in, m.a2)) return BEAGLE_ERROR_OUT_OF_RANGE; in->statMemReads++; }
            else if (m.k2 == mi355::PK_TIPS) { if (!in->tipStates[m.a2]) return BEAGLE_ERROR_OUT_OF_RANGE; d.src2 = in->tipStates[m.a2] + in->statePairOff; in->statTipReads++; }
            if (m.smode != mi355::PS_NONE) {
                int rc = ensureScale(in, m.scaleIdx); if (rc) return rc;
                if (m.smode == mi355::PS_WRITE) { in->scaleIsRaw[m.scaleIdx] = 1; in->statScaleWrites++; d.scaleW = in->scale[m.scaleIdx]; }
                else {
                    if (!in->scaleIsRaw[m.scaleIdx]) return BEAGLE_ERROR_OUT_OF_RANGE;   // never written by a rescaling op
                    in->statScaleReads++;
                    d.scale = in->scale[m.scaleIdx] + in->scaleStride;                    // read mode multiplies by the reciprocal
                }
            }
            if (m.storeBuf >= 0) {
                int rc = ensurePartials(in, m.storeBuf); if (rc) return rc;
                d.store = in->partials[m.storeBuf];
                in->statStored++;
            }
            d.m1 = in->matrices + (size_t)m.mat1 * matStride; d.m2 = in->matrices + (size_t)m.mat2 * matStride;
            d.flags = mi355::walkFlags(m.k1, m.k2, m.hold, m.smode, m.storeBuf >= 0);
            if (ablate) {       // TIMING EXPERIMENTS ONLY (wrong results): 1 no stores, 2 no partials loads, 4 no scale traffic, 8 no tip traffic
                if (ablate & 1) d.flags &= ~(unsigned)mi355::WF_STORE;
                if (ablate & 2) d.flags &= ~(unsigned)mi355::WF_X;
                if (ablate & 4) d.scale = in->onesScale;
                if (ablate & 8) { if (m.k1 == mi355::PK_TIPS) d.src1 = in->dummyTips; if (m.k2 == mi355::PK_TIPS) d.src2 = in->dummyTips; }
            }
            w.push_back(d);
        }
        if (ps.progCount & 1) w.push_back(nop);
        segs[si].progCount = (int)w.size() - segs[si].progStart;
        w.push_back(nop); w.push_back(nop);
        // the wait of every stage: "at most N vector-memory instructions outstanding".  Loads and stores share the counter.
        // DEFAULT (strict): N = the loads of the NEXT micro-operation only.  Sufficient under the one ordering rule the ISA
        // guides state for this counter — vector-memory LOADS return in the order they were issued: when at most N operations
        // are outstanding and the N youngest loads are all younger than this stage's loads, an unfinished load of this stage
        // would leave N + 1 unfinished, whatever the stores (of this or any earlier stage) do.
        // BEAGLE_MI355_STRICT_WAITS=0: N also counts the previous micro-operation's stores, i.e. assumes that a younger store
        // is never counted out before an older load.  That held in > 1e9 lane-trials (tests/test_gpu_vmcnt_order.py) and saves a
        // stage the acknowledgement of four stores per stored node — 1 % of config A (profiles/r03_experiments.txt 7) — but it
        // is an observation, not a documented guarantee, so it is not what ships by default.
        // A smaller N than the true number only waits longer (the table ends at 12).
        for (int i = segs[si].progStart; i < segs[si].progStart + segs[si].progCount; i++) {
            const int stores = in->strictWaits ? 0 : (i > segs[si].progStart ? mi355::walkStoreCount(w[i - 1].flags) : 0);
            w[i].flags |= mi355::walkWaitJump(std::min(mi355::walkFetchCount(w[i + 1].flags) + stores, 12));
            // the assembly loop always issues four small loads per stage: its wait is 4, 8 or 12
            const int code = (stores ? 1 : 0) + ((w[i + 1].flags & mi355::WF_X) ? 1 : 0);
            if (code == 1) w[i].flags |= mi355::WF_WAIT8; else if (code == 2) w[i].flags |= mi355::WF_WAIT12;
        }
        segs[si].pStart = in->partStart[ps.partition]; segs[si].pEnd = in->partEnd[ps.partition]; segs[si].tStart = in->padStart[ps.partition];
        maxRange = std::max(maxRange, segs[si].pEnd - segs[si].pStart);
    }
    if (slot) {
        slot->tag = planTag; slot->epoch = in->resolveEpoch; slot->maxRange = maxRange;
        slot->memReads = in->statMemReads - s0[0]; slot->tipReads = in->statTipReads - s0[1]; slot->scaleReads = in->statScaleReads - s0[2];
        slot->scaleWrites = in->statScaleWrites - s0[3]; slot->stored = in->statStored - s0[4];
    }
    }
    in->statMicroOps += (long)n;
    // pack: [micro-ops (64 B each) | segments (16 B each) | snapshot pairs] — ONE host-to-device copy
    const size_t opBytes = w.size() * sizeof(mi355::WalkOp), segBytes = segs.size() * sizeof(mi355::WalkSeg);
    const size_t pairBytes = plan.snapPairs.size() * sizeof(int), total = opBytes + segBytes + pairBytes;
    char* dBase = nullptr;
    if (reuse && slot->dProgValid) dBase = slot->dProg;          // a cached plan's program is already on the device, bit for bit
    else if (total <= RING_BYTES / 4) {
        const long off = stage(in, w.data(), opBytes, total);                    // reserves `total` bytes, copies the ops ...
        if (off < 0) return BEAGLE_ERROR_GENERAL;
        memcpy(in->hRing + off + opBytes, segs.data(), segBytes);                // ... the rest is filled in behind them
        if (pairBytes) memcpy(in->hRing + off + opBytes + segBytes, plan.snapPairs.data(), pairBytes);
        HIP_TRY(hipMemcpyAsync(in->dRing + off, in->hRing + off, total, hipMemcpyHostToDevice, in->stream));
        dBase = in->dRing + off;
        if (slot) {                                   // keep a device copy for the next time this plan comes out of the cache
            if (slot->dProgBytes < total) {
                if (slot->dProg) { HIP_TRY(hipStreamSynchronize(in->stream)); hipFree(slot->dProg); }
                slot->dProg = nullptr; slot->dProgBytes = 0;
                HIP_TRY(hipMalloc((void**)&slot->dProg, total + total / 4));
                slot->dProgBytes = total + total / 4;
            }
            HIP_TRY(hipMemcpyAsync(slot->dProg, in->dRing + off, total, hipMemcpyDeviceToDevice, in->stream));
            slot->dProgValid = true;
        }
    } else {                                  // a tree of > ~60 000 nodes: its own staging buffer, synchronous copy
        HIP_TRY(hipStreamSynchronize(in->stream));
        if (in->bigStageBytes < total) {
            if (in->bigStage) hipFree(in->bigStage);
            in->bigStage = nullptr; in->bigStageBytes = 0;
            HIP_TRY(hipMalloc((void**)&in->bigStage, total));
            in->bigStageBytes = total;
        }
        HIP_TRY(hipMemcpy(in->bigStage, w.data(), opBytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(in->bigStage + opBytes, segs.data(), segBytes, hipMemcpyHostToDevice));
        if (pairBytes) HIP_TRY(hipMemcpy(in->bigStage + opBytes + segBytes, plan.snapPairs.data(), pairBytes, hipMemcpyHostToDevice));
        dBase = in->bigStage;
    }
    if (pairBytes)
        mi355::launchSnapshotMatrices(in->stream, in->matrices, (const int*)(dBase + opBytes + segBytes), (int)(plan.snapPairs.size() / 2),
                                      in->C * in->S * in->S);
    // the matrix stream: both branch matrices of every micro-operation, in program order (after the snapshots they may name)
    const size_t streamBytes = w.size() * (size_t)in->C * 40 * sizeof(double) + 1024;   // 2 x 5 columns x 4 per category (kernels_walk4.hip)
    if (in->matStreamBytes < streamBytes) {
        HIP_TRY(hipStreamSynchronize(in->stream));
        if (in->matStream) hipFree(in->matStream);
        in->matStream = nullptr; in->matStreamBytes = 0;
        const size_t want = std::max(streamBytes + streamBytes / 4, (size_t)1 << 20);
        HIP_TRY(hipMalloc((void**)&in->matStream, want));
        in->matStreamBytes = want;
    }
    mi355::launchGatherMatrices(in->stream, (const mi355::WalkOp*)dBase, (int)w.size(), in->C, in->matStream);
    if (getenv("BEAGLE_MI355_DUMP_PLAN")) {           // development: the slices of this program, wave by wave
        fprintf(stderr, "[mi355] plan: %zu micro-ops in %zu slices:", n, segs.size());
        for (size_t i = 0; i < segs.size(); i++) fprintf(stderr, " w%d:%d", plan.segs[i].wave, segs[i].progCount);
        fprintf(stderr, "\n");
    }
    if (recordBeforeWalk) HIP_TRY(hipEventRecord(recordBeforeWalk, in->stream));
    // one launch per wave of independent slices (a single one unless the planner cut the forest for a small shard)
    for (size_t b = 0; b < segs.size();) {
        size_t e = b + 1;
        while (e < segs.size() && plan.segs[e].wave == plan.segs[b].wave) e++;
        int range = 0;
        const bool fast = in->fastWalk;                 // the assembly loop (BEAGLE_MI355_NO_FAST_WALK=1: the C++ reference kernel)
        for (size_t i = b; i < e; i++) range = std::max(range, segs[i].pEnd - segs[i].pStart);
        if (fast) {
            mi355::launchWalk4Fast(in->stream, (const mi355::WalkOp*)dBase, (const mi355::WalkSeg*)(dBase + opBytes) + b, (int)(e - b), range,
                                   in->matStream, in->P, in->C, (long)in->scaleStride);
            in->statFastWalks++;
        } else
            mi355::launchWalk4(in->stream, (const mi355::WalkOp*)dBase, (const mi355::WalkSeg*)(dBase + opBytes) + b, (int)(e - b), range,
                               in->matStream, in->P, in->C, (long)in->scaleStride);
        in->statWalks++;
        b = e;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Give every definition of `xs` its real partials: one program, one launch.  xs are definition KEYS (planner.h: buffer x
// partitionCount + partition; the buffer index itself on an instance with one partition).
int materializeCherries(Instance* in, const std::vector<int>& xs);
int materializeList(Instance* in, const std::vector<int>& xs) {
    if (!in->virt || xs.empty()) return 0;
    if (!in->walk) return materializeCherries(in, xs);
    mi355::Plan mp;
    in->planner.planMaterialize(xs, mp);
    return runPlan(in, mp);
}
int materializeVirtual(Instance* in, int X) {         // every partition of buffer X
    if (!isVirt(in, X)) return 0;
    std::vector<int> keys;
    in->planner.keysOf(X, keys);
    return materializeList(in, keys);
}
int materializeScaleUsers(Instance* in, int scaleIdx) {
    if (!in->virt || in->planner.scaleUsers(scaleIdx).empty()) return 0;
    return materializeList(in, std::vector<int>(in->planner.scaleUsers(scaleIdx)));
}
int materializeTipUsers(Instance* in, int tip) {
    if (!in->virt || in->planner.tipUsers(tip).empty()) return 0;
    return materializeList(in, std::vector<int>(in->planner.tipUsers(tip)));
}

int foldCumulative(Instance* in, const int* ops, int count, int tuple, int globalCum);

// A walk is one workgroup per 128 patterns, and every wave executes its program one dependent step after the other.  The
// planner therefore cuts the forest into independent subtrees that run side by side, wave after wave (planner.h): with few
// patterns (a shard of a multi-GPU run, a small alignment) that is what fills the 256 CUs at all (12 500 patterns:
// 0.83 -> 0.33 ms per evaluation); with many it keeps more workgroups than the chip holds in the queue, so that a wave
// waiting for its stores is replaced by another one instead of idling (1e5 patterns: 1.73 -> 1.37 ms).  Returns the target
// number of micro-operations per subtree, 0 = one walk.  BEAGLE_MI355_CHUNK overrides (0 = never).
int walkChunkOps(const Instance* in, int opCount) {
    static const int forced = getenv("BEAGLE_MI355_CHUNK") ? atoi(getenv("BEAGLE_MI355_CHUNK")) : -1;
    if (forced >= 0) return forced;
    if (opCount < 64) return 0;
    const long groups = (in->P + 127) / 128;
    // about 2 560 workgroups per wave of slices: 2.5 rounds of the 1 024 the chip holds (4 per CU).  Measured with the
    // assembly loop (tools/chunk_sweep.sh): 12 500 patterns 129 us at 40 micro-operations per slice vs 145 at 99; flat
    // between 50 and 300 from 25 000 patterns up
    return (int)std::min<long>(150, std::max<long>(24, (long)opCount * groups / 2560));
}

// 4 states: the operation list becomes one (or, for a list with hazards, a few) pattern-walk launches.
int runOperationsWalk(Instance* in, const int* ops, int count, int tuple, int globalCum) {
    if (count <= 0) return 0;
    typedef std::chrono::steady_clock Clock;
    const Clock::time_point t0 = Clock::now();
    auto usSince = [](Clock::time_point a) { return std::chrono::duration<double, std::micro>(Clock::now() - a).count(); };
    in->hostCalls++;
    const int parts = in->partitionCount;
    // destinations may become virtual: 7-int lists of an unpartitioned instance, 9-int lists (definitions are per partition)
    const bool allowVirtual = tuple == BEAGLE_PARTITION_OP_COUNT || parts == 1;
    // The chain's steady state — the SAME full-evaluation list as one seen before, no rescaling in it — needs none of the
    // per-operation work below: it was range-checked and planned then, nothing has to be materialised or accumulated for it,
    // the planner re-establishes its definitions with one comparison per operation and the program is resident on the
    // device (config E, 6 436 operations: 116 -> 35 us of host time per call; profiles/r03_experiments.txt).
    {
        bool simple = false;
        if (in->planner.replayCached(ops, count, tuple, parts, allowVirtual, walkChunkOps(in, count), &simple)) {
            const double usPlan = usSince(t0);
            in->hostPlanUs += usPlan; in->hostPlanHitUs += usPlan; in->hostHits++;
            const Clock::time_point t1 = Clock::now();
            hipEvent_t a = nullptr, b = nullptr;
            const bool launches = !in->planner.planned->prog.empty();
            if (in->timing && launches) {
                if (in->eventsUsed == in->events.size()) { hipEvent_t x, y; HIP_TRY(hipEventCreate(&x)); HIP_TRY(hipEventCreate(&y)); in->events.emplace_back(x, y); }
                a = in->events[in->eventsUsed].first; b = in->events[in->eventsUsed].second; in->eventsUsed++;
            }
            int rc = runPlan(in, *in->planner.planned, in->planner.plannedTag, a); if (rc) return rc;
            if (b) { HIP_TRY(hipEventRecord(b, in->stream)); in->pendingLaunches++; }
            const double usRun = usSince(t1);
            in->hostRunUs += usRun; in->hostRunHitUs += usRun;
            return 0;
        }
    }
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        const int dest = op[0], wS = op[1], rS = op[2], c1 = op[3], m1 = op[4], c2 = op[5], m2 = op[6];
        int part = 0, cum = globalCum;
        if (tuple == BEAGLE_PARTITION_OP_COUNT) { part = op[7]; cum = op[8]; }
        if (badIndex(dest, in->partialsCount) || badIndex(c1, in->partialsCount) || badIndex(c2, in->partialsCount) ||
            badIndex(m1, in->matrixCount) || badIndex(m2, in->matrixCount) || badIndex(part, parts) ||
            (wS != BEAGLE_OP_NONE && badIndex(wS, in->scaleCount)) || (rS != BEAGLE_OP_NONE && badIndex(rS, in->scaleCount)) ||
            (cum != BEAGLE_OP_NONE && badIndex(cum, in->scaleCount)))
            return BEAGLE_ERROR_OUT_OF_RANGE;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (in->timing) {
        if (in->eventsUsed == in->events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
            in->events.emplace_back(a, b);
        }
        e0 = in->events[in->eventsUsed].first; e1 = in->events[in->eventsUsed].second; in->eventsUsed++;
    }
    int launches = 0;
    in->hostPrepUs += usSince(t0);
    for (int begin = 0; begin < count;) {
        Clock::time_point t1 = Clock::now();
        const int n = in->planner.hazardFreePrefix(ops, begin, count, tuple, parts);
        const int* sub = ops + (size_t)begin * tuple;
        for (int k = 0; k < n; k++) {                       // a tip index reused as a destination now holds partials
            const int dest = sub[(size_t)k * tuple];
            if (isCompactTip(in, dest) || in->planner.leafPartials[dest]) { int rcm = materializeTipUsers(in, dest); if (rcm) return rcm; in->tipStates[dest] = nullptr; setCompact(in, dest, false); }
        }
        std::vector<int> need;
        in->planner.mustMaterializeBefore(sub, n, tuple, need);
        if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
        in->hostPrepUs += usSince(t1); t1 = Clock::now();
        const long hitsBefore = in->planner.cacheHits;
        int rc = in->planner.plan(sub, n, tuple, parts, allowVirtual, in->plan, walkChunkOps(in, n));
        if (rc) return rc;
        const bool hit = in->planner.cacheHits != hitsBefore;
        { const double us = usSince(t1); in->hostPlanUs += us; if (hit) { in->hostPlanHitUs += us; in->hostHits++; } }
        t1 = Clock::now();
        // with the kernel timer on, ONE HIP-event pair brackets the walk launches of the call (the program upload and the
        // snapshot copies are outside: the events time the pruning kernel, which is what the roofline is about)
        if (!in->planner.planned->prog.empty()) {
            rc = runPlan(in, *in->planner.planned, in->planner.plannedTag, launches == 0 ? e0 : nullptr); if (rc) return rc;
            launches++;
        } else { rc = runPlan(in, *in->planner.planned, in->planner.plannedTag); if (rc) return rc; }
        { const double us = usSince(t1); in->hostRunUs += us; if (hit) in->hostRunHitUs += us; }
        begin += n;
    }
    if (e1 && launches > 0) { HIP_TRY(hipEventRecord(e1, in->stream)); in->pendingLaunches += launches; }
    else if (e1) in->eventsUsed--;        // nothing was launched: give the (unrecorded) event pair back
    return foldCumulative(in, ops, count, tuple, globalCum);
}

// T32 instances: give the virtual cherries of `xs` their real partials — each is one ordinary tip-tip operation on its
// snapshot matrices; all of them are independent (one level launch).
int materializeCherries(Instance* in, const std::vector<int>& xs) {
    std::vector<OpDesc> descs;
    for (int X : xs) {
        if (!in->planner.isVirtual(X)) continue;
        const mi355::VirtDef& v = in->planner.definition(X);
        const mi355::VirtStep& st = v.steps[0];
        if (v.nSteps != 1 || st.type != mi355::VT_CHERRY || !in->tipStates[st.tipA] || !in->tipStates[st.tipB]) return BEAGLE_ERROR_GENERAL;
        int rc = ensurePartials(in, X); if (rc) return rc;
        OpDesc d;
        memset(&d, 0, sizeof(d));
        d.dest = in->partials[X];
        d.child1 = in->tipStates[st.tipA]; d.child2 = in->tipStates[st.tipB];
        d.kind = mi355::KIND_STATES1 | mi355::KIND_STATES2;
        d.mat1 = in->planner.snapSlot(X, 0, 0); d.mat2 = in->planner.snapSlot(X, 0, 1);
        if (st.scaleIdx >= 0) { if (!in->scale[st.scaleIdx]) return BEAGLE_ERROR_GENERAL; d.scaleRead = in->scale[st.scaleIdx]; }
        d.pStart = 0; d.pEnd = in->P;
        descs.push_back(d);
        in->planner.clearVirtual(X);
    }
    if (descs.empty()) return 0;
    void* dOps = nullptr;
    int rc = uploadTransient(in, descs.data(), descs.size() * sizeof(OpDesc), &dOps); if (rc) return rc;
    mi355::launchPruneLevelTiled(in->stream, (const OpDesc*)dOps, (int)descs.size(), in->matrices, in->P, in->S, in->C, false);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Enqueue an op list level by level (every state count but 4).  `tuple` is 7 (updatePartials) or 9 (updatePartialsByPartition).
int runOperationsLevels(Instance* in, const int* ops, int count, int tuple, int globalCum) {
    if (count <= 0) return 0;
    const int parts = in->partitionCount;
    std::vector<OpDesc> descs;                   // one per op that launches (never reallocated: references stay valid)
    descs.reserve(count);
    std::vector<int> descOf(count, -1);
    std::vector<int> level(count);
    std::vector<int> predOff(count + 1, 0), predList;                 // RAW / WAW edges: producer op -> this op
    bool warSeen = false;                                             // a write-after-read hazard inside the list (never in BEAST's lists)
    predList.reserve((size_t)count * 3);
    // virtual cherries (in->cherry): definitions that read a scale buffer this list rewrites, or that the list updates in
    // place, get their data first; new ones are only made by single-partition 7-int lists
    const bool cherryList = in->cherry && parts == 1 && tuple == BEAGLE_OP_COUNT;
    std::vector<mi355::CherryDesc> cherries;
    std::vector<int> snapPairs;
    std::vector<char> skipped(count, 0);                             // ops that only defined a cherry
    if (in->virt) {
        for (int k = 0; k < count; k++) {                            // (range checks of these fields: same loop below, nothing is touched before it passes)
            const int* op = ops + (size_t)k * tuple;
            if (badIndex(op[0], in->partialsCount) || badIndex(op[3], in->partialsCount) || badIndex(op[5], in->partialsCount) ||
                (op[1] != BEAGLE_OP_NONE && badIndex(op[1], in->scaleCount))) return BEAGLE_ERROR_OUT_OF_RANGE;
        }
        std::vector<int> need;
        in->planner.mustMaterializeBefore(ops, count, tuple, need);
        if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    }
    auto cherryChild = [&](int c) -> size_t {                         // descriptor index of a virtual child
        const mi355::VirtStep& st = in->planner.definition(c).steps[0];
        mi355::CherryDesc cd;
        cd.tipA = in->tipStates[st.tipA]; cd.tipB = in->tipStates[st.tipB];
        cd.scale = st.scaleIdx >= 0 ? in->scale[st.scaleIdx] : nullptr;
        cd.matA = in->planner.snapSlot(c, 0, 0); cd.matB = in->planner.snapSlot(c, 0, 1);
        cherries.push_back(cd);
        return cherries.size() - 1;
    };
    in->stamp++;
    int maxLevel = 0;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        const int dest = op[0], wS = op[1], rS = op[2], c1 = op[3], m1 = op[4], c2 = op[5], m2 = op[6];
        int part = 0, cum = globalCum;
        if (tuple == BEAGLE_PARTITION_OP_COUNT) { part = op[7]; cum = op[8]; }
        if (badIndex(dest, in->partialsCount) || badIndex(c1, in->partialsCount) || badIndex(c2, in->partialsCount) ||
            badIndex(m1, in->matrixCount) || badIndex(m2, in->matrixCount) || badIndex(part, parts) ||
            (wS != BEAGLE_OP_NONE && badIndex(wS, in->scaleCount)) || (rS != BEAGLE_OP_NONE && badIndex(rS, in->scaleCount)) ||
            (cum != BEAGLE_OP_NONE && badIndex(cum, in->scaleCount)))
            return BEAGLE_ERROR_OUT_OF_RANGE;
        const bool tip1 = isCompactTip(in, c1), tip2 = isCompactTip(in, c2);
        const int ownScale = wS != BEAGLE_OP_NONE ? wS : rS;
        if (wS != BEAGLE_OP_NONE || rS != BEAGLE_OP_NONE) { int rcs = ensureScale(in, ownScale); if (rcs) return rcs; }
        // a tip-tip node that does not rescale now is DEFINED, not computed: nothing is launched for it (a definition
        // reads its scale buffer in read mode only, so that buffer must hold factors already)
        if (cherryList && tip1 && tip2 && wS == BEAGLE_OP_NONE && dest != c1 && dest != c2 && dest >= in->tipCount &&
            (rS == BEAGLE_OP_NONE || in->scaleIsRaw[rS]) && in->planner.defineCherry(dest, c1, m1, c2, m2, rS == BEAGLE_OP_NONE ? -1 : rS, snapPairs)) {
            skipped[k] = 1; level[k] = 0; predOff[k] = (int)predList.size();
            in->statCherries++;
            continue;
        }
        if (isVirt(in, dest)) clearVirtual(in, dest);                // whatever it was, this op gives it real data
        // traffic counters (beagleMi355WalkStats): one stored node; per child a partials read, a tip-state read or — for a
        // virtual cherry — two tip-state reads and its scale factors
        in->statMicroOps++; in->statStored++;
        for (int w = 0; w < 2; w++) {
            const int c = w ? c2 : c1;
            if (w ? tip2 : tip1) in->statTipReads++;
            else if (isVirt(in, c)) { in->statTipReads += 2; if (in->planner.definition(c).steps[0].scaleIdx >= 0) in->statScaleReads++; }
            else in->statMemReads++;
        }
        if (wS != BEAGLE_OP_NONE) in->statScaleWrites++; else if (rS != BEAGLE_OP_NONE) in->statScaleReads++;
        descOf[k] = (int)descs.size(); descs.emplace_back();
        OpDesc& d = descs.back();
        memset(&d, 0, sizeof(d));
        if (tip1) { d.child1 = in->tipStates[c1]; d.kind |= mi355::KIND_STATES1; }
        else if (isVirt(in, c1)) { d.child1 = (const void*)cherryChild(c1); d.kind |= mi355::KIND_CHERRY1; }
        else if (in->partials[c1]) d.child1 = in->partials[c1];
        else return BEAGLE_ERROR_OUT_OF_RANGE;
        if (tip2) { d.child2 = in->tipStates[c2]; d.kind |= mi355::KIND_STATES2; }
        else if (isVirt(in, c2)) { d.child2 = (const void*)cherryChild(c2); d.kind |= mi355::KIND_CHERRY2; }
        else if (in->partials[c2]) d.child2 = in->partials[c2];
        else return BEAGLE_ERROR_OUT_OF_RANGE;
        int rc = ensurePartials(in, dest); if (rc) return rc;
        d.dest = in->partials[dest];
        d.mat1 = m1; d.mat2 = m2;
        if (wS != BEAGLE_OP_NONE) {
            rc = ensureScale(in, wS); if (rc) return rc;
            d.scaleWrite = in->scale[wS]; in->scaleIsRaw[wS] = 1;
        } else if (rS != BEAGLE_OP_NONE) {
            rc = ensureScale(in, rS); if (rc) return rc;
            if (!in->scaleIsRaw[rS]) return BEAGLE_ERROR_OUT_OF_RANGE;   // never written by a rescaling op
            d.scaleRead = in->scale[rS];
        }
        d.pStart = in->partStart[part]; d.pEnd = in->partEnd[part];
        // dependency level: after the ops (of this call) that produced my children (RAW), that read my
        // destination (WAR) or that wrote it (WAW); hazards are tracked per (buffer, partition)
        int lvl = 0;
        const size_t kc1 = (size_t)c1 * parts + part, kc2 = (size_t)c2 * parts + part, kd = (size_t)dest * parts + part;
        predOff[k] = (int)predList.size();
        if (in->wStamp[kc1] == in->stamp) { lvl = std::max(lvl, in->wLevel[kc1] + 1); predList.push_back(in->wOp[kc1]); }
        if (in->wStamp[kc2] == in->stamp) { lvl = std::max(lvl, in->wLevel[kc2] + 1); predList.push_back(in->wOp[kc2]); }
        if (in->wStamp[kd] == in->stamp) { lvl = std::max(lvl, in->wLevel[kd] + 1); predList.push_back(in->wOp[kd]); }
        if (in->rStamp[kd] == in->stamp) { lvl = std::max(lvl, in->rLevel[kd] + 1); warSeen = true; }
        level[k] = lvl; maxLevel = std::max(maxLevel, lvl);
        in->wStamp[kd] = in->stamp; in->wLevel[kd] = lvl; in->wOp[kd] = k;
        if (in->rStamp[kc1] != in->stamp || in->rLevel[kc1] < lvl) { in->rStamp[kc1] = in->stamp; in->rLevel[kc1] = lvl; }
        if (in->rStamp[kc2] != in->stamp || in->rLevel[kc2] < lvl) { in->rStamp[kc2] = in->stamp; in->rLevel[kc2] = lvl; }
    }
    // ASAP levels put every tip-tip op ("cherry", write-only traffic) into the first launch and leave the read-heavy
    // ops to later ones, so the HBM sees a write-bound phase (~3.7 TB/s) followed by read-heavy phases.  ALAP levels
    // (= depth below the root, BEAST's own "reverse level order") spread the cherries over all launches: every launch
    // then mixes reads and writes, which is where the memory system is fastest.  Same number of launches either way.
    predOff[count] = (int)predList.size();
    if (in->schedAlap && !warSeen) {
        std::vector<int> alap(count, maxLevel);
        for (int k = count - 1; k >= 0; k--)
            for (int e = predOff[k]; e < predOff[k + 1]; e++) {
                const int a = predList[e];
                if (alap[a] > alap[k] - 1) alap[a] = alap[k] - 1;
            }
        level.swap(alap);
    }
    // counting sort by level (stable)
    std::vector<int> start(maxLevel + 2, 0);
    for (int k = 0; k < count; k++) if (!skipped[k]) start[level[k] + 1]++;
    for (int l = 0; l <= maxLevel; l++) start[l + 1] += start[l];
    const int launchCount = start[maxLevel + 1];
    std::vector<OpDesc> sorted(std::max(1, launchCount));
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int k = 0; k < count; k++) if (!skipped[k]) sorted[fill[level[k]]++] = descs[descOf[k]];
    // the cherries' matrix snapshots and the descriptors of the virtual children, ahead of the level launches
    const mi355::CherryDesc* dCherries = nullptr;
    if (!snapPairs.empty()) {
        void* dPairs = nullptr;
        int rc = uploadTransient(in, snapPairs.data(), snapPairs.size() * sizeof(int), &dPairs); if (rc) return rc;
        mi355::launchSnapshotMatrices(in->stream, in->matrices, (const int*)dPairs, (int)(snapPairs.size() / 2), in->C * in->S * in->S);
    }
    if (!cherries.empty()) {
        void* dC = nullptr;
        int rc = uploadTransient(in, cherries.data(), cherries.size() * sizeof(mi355::CherryDesc), &dC); if (rc) return rc;
        dCherries = (const mi355::CherryDesc*)dC;
    }
    // ONE descriptor upload for the whole list (every extra copy is a dependent blit kernel between two
    // level launches: ~4 us + two boundaries), chunked only when the list would not fit the ring; then one
    // launch per dependency level reading its slice.  With the kernel timer on, ONE HIP-event pair brackets
    // all level launches of the call (gaps between levels included — they are part of what the path costs).
    const size_t maxChunkOps = (RING_BYTES / 4) / sizeof(OpDesc);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (in->timing) {
        if (in->eventsUsed == in->events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
            in->events.emplace_back(a, b);
        }
        e0 = in->events[in->eventsUsed].first; e1 = in->events[in->eventsUsed].second; in->eventsUsed++;
    }
    int launches = 0;
    for (int chunkBegin = 0; chunkBegin < launchCount;) {
        const int chunkEnd = (int)std::min<size_t>((size_t)launchCount, (size_t)chunkBegin + maxChunkOps);
        void* dChunk = nullptr;
        int rc = uploadTransient(in, &sorted[chunkBegin], (size_t)(chunkEnd - chunkBegin) * sizeof(OpDesc), &dChunk);
        if (rc) return rc;
        if (e0 && chunkBegin == 0) HIP_TRY(hipEventRecord(e0, in->stream));
        for (int l = 0; l <= maxLevel; l++) {
            const int begin = std::max(start[l], chunkBegin), end = std::min(start[l + 1], chunkEnd);
            if (begin >= end) continue;
            int maxRange = 0;
            bool anyWrite = false;
            for (int k = begin; k < end; k++) {
                maxRange = std::max(maxRange, sorted[k].pEnd - sorted[k].pStart);
                anyWrite = anyWrite || sorted[k].scaleWrite != nullptr;
            }
            if (in->tiled)
                mi355::launchPruneLevelTiled(in->stream, (const OpDesc*)dChunk + (begin - chunkBegin), end - begin, in->matrices,
                                             in->P, in->S, in->C, anyWrite, dCherries);
            else
                mi355::launchPruneLevel(in->stream, (const OpDesc*)dChunk + (begin - chunkBegin), end - begin, in->matrices,
                                        in->P, in->S, in->C, maxRange);
            launches++;
        }
        chunkBegin = chunkEnd;
    }
    if (e1 && launches > 0) { HIP_TRY(hipEventRecord(e1, in->stream)); in->pendingLaunches += launches; }
    else if (e1) in->eventsUsed--;        // nothing was launched: give the (unrecorded) event pair back
    HIP_TRY(hipGetLastError());
    return foldCumulative(in, ops, count, tuple, globalCum);
}

// cumulative scale factors requested together with an update: fold the factors the list wrote into the cumulative
// buffer afterwards, in op order (deterministic; no cross-workgroup atomics)
int foldCumulative(Instance* in, const int* ops, int count, int tuple, int globalCum) {
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        const int wS = op[1];
        int part = 0, cum = globalCum;
        if (tuple == BEAGLE_PARTITION_OP_COUNT) { part = op[7]; cum = op[8]; }
        if (cum == BEAGLE_OP_NONE || wS == BEAGLE_OP_NONE) continue;
        int rc = materializeScaleUsers(in, cum); if (rc) return rc;
        rc = ensureScale(in, cum); if (rc) return rc;
        const double* src = in->scale[wS];
        int one = 1;
        void *dSrc = nullptr, *dRaw = nullptr;
        rc = uploadTransient(in, &src, sizeof(src), &dSrc); if (rc) return rc;
        rc = uploadTransient(in, &one, sizeof(one), &dRaw); if (rc) return rc;
        mi355::launchAccumulateScale(in->stream, in->scale[cum], (const double* const*)dSrc, (const int*)dRaw, 1, 1.0,
                                     in->partStart[part], in->partEnd[part]);
    }
    return 0;
}

int runOperations(Instance* in, const int* ops, int count, int tuple, int globalCum) {
    return in->walk ? runOperationsWalk(in, ops, count, tuple, globalCum) : runOperationsLevels(in, ops, count, tuple, globalCum);
}

// One dependency level of pre-order ops on the T32 layout, expressed with the tuned pruning kernel:
//   pass A   tmp        = (I . pre(parent)) * (P_sib . post(sib))         a pruning op whose first branch matrix is the identity
//   pass B   pre(child) = (P_child^T . tmp) * 1                            a pruning op whose second child is an all-missing tip
// (products with the identity's 0/1 entries and the sums of the resulting zeros are exact, so pass A adds no rounding).
// PRE_SCRATCH ops at a time: their tmp buffers and transposed matrices are reused by the next chunk in stream order.
int ensurePreScratch(Instance* in) {
    if (!in->preScratch.empty()) return 0;
    void* slab = nullptr;
    int rc = devAlloc(in, &slab, in->partialsBytes * PRE_SCRATCH); if (rc) return rc;
    void* miss = nullptr;
    rc = devAlloc(in, &miss, ((size_t)in->P + 255) & ~(size_t)255); if (rc) return rc;
    in->preMissing = (uint8_t*)miss;
    HIP_TRY(hipMemsetAsync(in->preMissing, in->S, (size_t)in->P, in->stream));
    in->preScratch.assign(PRE_SCRATCH, nullptr);
    for (int j = 0; j < PRE_SCRATCH; j++) in->preScratch[j] = (double*)((char*)slab + in->partialsBytes * j);
    return 0;
}

int preLevelTwoPass(Instance* in, const OpDesc* ops, int nOps) {
    { int rc0 = ensurePreScratch(in); if (rc0) return rc0; }
    std::vector<OpDesc> pass(2 * PRE_SCRATCH);
    std::vector<int> pairs(2 * PRE_SCRATCH);
    for (int b = 0; b < nOps; b += PRE_SCRATCH) {
        const int n = std::min(PRE_SCRATCH, nOps - b);
        bool anyWrite = false;
        for (int j = 0; j < n; j++) {
            const OpDesc& o = ops[b + j];
            pairs[2 * j] = o.mat1; pairs[2 * j + 1] = in->preTransposed + j;
            OpDesc& a = pass[j];
            memset(&a, 0, sizeof(a));
            a.dest = in->preScratch[j];
            a.child1 = o.child1; a.mat1 = in->preIdentity;
            a.child2 = o.child2; a.mat2 = o.mat2; a.kind = o.kind & mi355::KIND_STATES2;
            a.pStart = 0; a.pEnd = in->P;
            OpDesc& c = pass[n + j];
            memset(&c, 0, sizeof(c));
            c.dest = o.dest;
            c.child1 = in->preScratch[j]; c.mat1 = in->preTransposed + j;
            c.child2 = in->preMissing; c.mat2 = in->preIdentity; c.kind = mi355::KIND_STATES2;
            c.scaleWrite = o.scaleWrite; c.scaleRead = o.scaleRead;
            c.pStart = 0; c.pEnd = in->P;
            anyWrite = anyWrite || o.scaleWrite != nullptr;
        }
        void *dPairs = nullptr, *dPass = nullptr;
        int rc = uploadTransient(in, pairs.data(), (size_t)2 * n * sizeof(int), &dPairs); if (rc) return rc;
        rc = uploadTransient(in, pass.data(), (size_t)2 * n * sizeof(OpDesc), &dPass); if (rc) return rc;
        mi355::launchTransposeMatrices(in->stream, in->matrices, (const int*)dPairs, n, in->S, in->C);
        mi355::launchPruneLevelTiled(in->stream, (const OpDesc*)dPass, n, in->matrices, in->P, in->S, in->C, false);
        mi355::launchPruneLevelTiled(in->stream, (const OpDesc*)dPass + n, n, in->matrices, in->P, in->S, in->C, anyWrite);
    }
    return 0;
}

// Enqueue a pre-order op list (7-int tuples {pre(child), writeScale, readScale, pre(parent), matrix(child), post(sibling),
// matrix(sibling)}, AbstractBeagleGradientDelegate.java:207-221).  A parent's op precedes its children's; the list is
// levelised like a post-order one and each level is one launch.
int runPreOperations(Instance* in, const int* ops, int count, int globalCum) {
    if (count <= 0) return 0;
    if (in->partitionCount != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    const int n = in->partialsCount;
    // everything these ops read must be real data, and nothing they overwrite may still define a virtual buffer
    std::vector<int> need;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * BEAGLE_OP_COUNT;
        const int dest = op[0], wS = op[1], rS = op[2], par = op[3], mc = op[4], sib = op[5], ms = op[6];
        if (badIndex(dest, n) || badIndex(par, n) || badIndex(sib, n) || badIndex(mc, in->matrixCount) || badIndex(ms, in->matrixCount) ||
            (wS != BEAGLE_OP_NONE && badIndex(wS, in->scaleCount)) || (rS != BEAGLE_OP_NONE && badIndex(rS, in->scaleCount)) ||
            (globalCum != BEAGLE_OP_NONE && badIndex(globalCum, in->scaleCount)) || dest == par || dest == sib)
            return BEAGLE_ERROR_OUT_OF_RANGE;
        if (isVirt(in, sib)) need.push_back(sib);
        if (isVirt(in, par)) need.push_back(par);
        if (in->virt) {
            need.insert(need.end(), in->planner.tipUsers(dest).begin(), in->planner.tipUsers(dest).end());
            if (wS != BEAGLE_OP_NONE) need.insert(need.end(), in->planner.scaleUsers(wS).begin(), in->planner.scaleUsers(wS).end());
        }
    }
    if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    std::vector<OpDesc> descs(count);
    std::vector<int> level(count), wLevel(n, -1), rLevel(n, -1), opWrite(count, BEAGLE_OP_NONE);
    int maxLevel = 0;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * BEAGLE_OP_COUNT;
        const int dest = op[0], wS = op[1], rS = op[2], par = op[3], mc = op[4], sib = op[5], ms = op[6];
        OpDesc& d = descs[k];
        memset(&d, 0, sizeof(d));
        clearVirtual(in, dest);
        int rc = ensurePartials(in, dest); if (rc) return rc;
        in->tipStates[dest] = nullptr; setCompact(in, dest, false);
        if (!in->partials[par] || (in->tipStates[par] && par < in->tipCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        d.dest = in->partials[dest];
        d.child1 = in->partials[par];
        if (in->tipStates[sib] && sib < in->tipCount) { d.child2 = in->tipStates[sib]; d.kind = mi355::KIND_STATES2; }
        else if (in->partials[sib]) d.child2 = in->partials[sib];
        else return BEAGLE_ERROR_OUT_OF_RANGE;
        d.mat1 = mc; d.mat2 = ms;
        if (wS != BEAGLE_OP_NONE) {
            rc = ensureScale(in, wS); if (rc) return rc;
            d.scaleWrite = in->scale[wS]; in->scaleIsRaw[wS] = 1; opWrite[k] = wS;
        } else if (rS != BEAGLE_OP_NONE) {
            rc = ensureScale(in, rS); if (rc) return rc;
            if (!in->scaleIsRaw[rS]) return BEAGLE_ERROR_OUT_OF_RANGE;
            d.scaleRead = in->scale[rS];
        }
        d.pStart = 0; d.pEnd = in->P;
        const int lvl = std::max(std::max(wLevel[par], wLevel[sib]), std::max(wLevel[dest], rLevel[dest])) + 1;
        level[k] = lvl; maxLevel = std::max(maxLevel, lvl);
        wLevel[dest] = lvl; rLevel[par] = std::max(rLevel[par], lvl); rLevel[sib] = std::max(rLevel[sib], lvl);
    }
    std::vector<int> start(maxLevel + 2, 0);
    for (int k = 0; k < count; k++) start[level[k] + 1]++;
    for (int l = 0; l <= maxLevel; l++) start[l + 1] += start[l];
    std::vector<OpDesc> sorted(count);
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int k = 0; k < count; k++) sorted[fill[level[k]]++] = descs[k];
    const size_t maxChunkOps = (RING_BYTES / 4) / sizeof(OpDesc);
    // T32 layout (16..64 states): two passes of the MFMA pruning kernel per op instead of the VALU pre-order kernel
    // (BEAGLE_MI355_PRE_NAIVE=1 keeps the latter, for A/B runs)
    static const bool preNaive = getenv("BEAGLE_MI355_PRE_NAIVE") && atoi(getenv("BEAGLE_MI355_PRE_NAIVE")) != 0;
    const bool twoPass = in->tiled && !preNaive;
    for (int chunkBegin = 0; chunkBegin < count;) {
        const int chunkEnd = (int)std::min<size_t>((size_t)count, (size_t)chunkBegin + maxChunkOps);
        void* dChunk = nullptr;
        int rc = twoPass ? 0 : uploadTransient(in, &sorted[chunkBegin], (size_t)(chunkEnd - chunkBegin) * sizeof(OpDesc), &dChunk);
        if (rc) return rc;
        for (int l = 0; l <= maxLevel; l++) {
            const int begin = std::max(start[l], chunkBegin), end = std::min(start[l + 1], chunkEnd);
            if (begin >= end) continue;
            if (twoPass) { int rc2 = preLevelTwoPass(in, &sorted[begin], end - begin); if (rc2) return rc2; continue; }
            mi355::launchPrePartials(in->stream, (const OpDesc*)dChunk + (begin - chunkBegin), end - begin, in->matrices,
                                     in->P, in->S, in->C, in->tiled, in->P, in->walk ? (long)in->scaleStride : 0);
        }
        chunkBegin = chunkEnd;
    }
    HIP_TRY(hipGetLastError());
    if (globalCum != BEAGLE_OP_NONE)
        for (int k = 0; k < count; k++) {
            if (opWrite[k] == BEAGLE_OP_NONE) continue;
            int rc = ensureScale(in, globalCum); if (rc) return rc;
            const double* src = in->scale[opWrite[k]];
            int one = 1;
            void *dSrc = nullptr, *dRaw = nullptr;
            rc = uploadTransient(in, &src, sizeof(src), &dSrc); if (rc) return rc;
            rc = uploadTransient(in, &one, sizeof(one), &dRaw); if (rc) return rc;
            mi355::launchAccumulateScale(in->stream, in->scale[globalCum], (const double* const*)dSrc, (const int*)dRaw, 1, 1.0, 0, in->P);
        }
    return 0;
}

// Per-edge derivative sums (AbstractBeagleBranchGradientDelegate.java:82-92).  Edges are processed in chunks that bound
// the scratch memory (block sums, and the optional per-pattern matrix) to a few hundred MB.
int edgeDifferentials(Instance* in, const int* postIdx, const int* preIdx, const int* dIdx, int wIdx, int count,
                      double* outDerivatives, double* outSum, double* outSumSquared) {
    if (count <= 0) return 0;
    if (in->partitionCount != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    std::vector<int> need;
    for (int e = 0; e < count; e++) {
        if (badIndex(postIdx[e], in->partialsCount) || badIndex(preIdx[e], in->partialsCount) || badIndex(dIdx[e], in->matrixCount))
            return BEAGLE_ERROR_OUT_OF_RANGE;
        if (isVirt(in, postIdx[e])) need.push_back(postIdx[e]);
        if (isVirt(in, preIdx[e])) need.push_back(preIdx[e]);
    }
    if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    const int nb = mi355::edgeBlocks(in->P);
    const size_t perEdgeBytes = (size_t)nb * 2 * sizeof(double) + 2 * sizeof(double) + (outDerivatives ? (size_t)in->P * sizeof(double) : 0);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)256 << 20) / perEdgeBytes));
    chunk = std::min(chunk, 32768);
    double *dBlock = nullptr, *dSums = nullptr, *dPer = nullptr;
    HIP_TRY(hipMalloc((void**)&dBlock, (size_t)chunk * nb * 2 * sizeof(double)));
    hipError_t e1 = hipMalloc((void**)&dSums, (size_t)chunk * 2 * sizeof(double));
    hipError_t e2 = outDerivatives ? hipMalloc((void**)&dPer, (size_t)chunk * in->P * sizeof(double)) : hipSuccess;
    int rc = (e1 != hipSuccess || e2 != hipSuccess) ? BEAGLE_ERROR_OUT_OF_MEMORY : 0;
    std::vector<mi355::EdgeDesc> descs;
    std::vector<double> sums;
    static const bool preNaive = getenv("BEAGLE_MI355_PRE_NAIVE") && atoi(getenv("BEAGLE_MI355_PRE_NAIVE")) != 0;
    const bool twoStep = in->tiled && !preNaive;
    if (twoStep && !rc) rc = ensurePreScratch(in);
    for (int b = 0; b < count && !rc; b += chunk) {
        const int m = std::min(chunk, count - b);
        descs.assign(m, mi355::EdgeDesc());
        for (int e = 0; e < m && !rc; e++) {
            const int po = postIdx[b + e], pr = preIdx[b + e];
            mi355::EdgeDesc& d = descs[e];
            if (in->tipStates[po] && po < in->tipCount) { d.post = in->tipStates[po]; d.postIsStates = 1; }
            else if (in->partials[po]) { d.post = in->partials[po]; d.postIsStates = 0; }
            else rc = BEAGLE_ERROR_OUT_OF_RANGE;
            if (!in->partials[pr] || (in->tipStates[pr] && pr < in->tipCount)) rc = BEAGLE_ERROR_OUT_OF_RANGE;
            d.pre = in->partials[pr];
            d.dmat = dIdx[b + e];
        }
        if (rc) break;
        // 16..64 states: an edge below an internal node takes the O(S^2) part through one pass of the MFMA pruning kernel
        // (tmp = (I . pre) * (D . post)) and a streaming reduction; tip edges (O(S) per pattern) and every other state
        // count use the direct kernel.  Output rows are addressed by EdgeDesc::slot, so the two groups can interleave.
        std::vector<mi355::EdgeDesc> direct, viaPrune;
        for (int e = 0; e < m; e++) {
            descs[e].slot = e;
            (twoStep && !descs[e].postIsStates ? viaPrune : direct).push_back(descs[e]);
        }
        if (!direct.empty()) {
            void* dDesc = nullptr;
            rc = uploadTransient(in, direct.data(), direct.size() * sizeof(mi355::EdgeDesc), &dDesc); if (rc) break;
            mi355::launchEdgeDifferentials(in->stream, (const mi355::EdgeDesc*)dDesc, (int)direct.size(), in->matrices,
                                           in->weights + (size_t)wIdx * in->C, in->patternWeights, dPer, dBlock, in->P, in->S, in->C, in->tiled);
        }
        for (size_t q = 0; q < viaPrune.size() && !rc; q += PRE_SCRATCH) {
            const int n = (int)std::min<size_t>(PRE_SCRATCH, viaPrune.size() - q);
            std::vector<OpDesc> pass(n);
            for (int j = 0; j < n; j++) {
                mi355::EdgeDesc& ed = viaPrune[q + j];
                OpDesc& a = pass[j];
                memset(&a, 0, sizeof(a));
                a.dest = in->preScratch[j];
                a.child1 = ed.pre; a.mat1 = in->preIdentity;
                a.child2 = ed.post; a.mat2 = ed.dmat;
                a.pStart = 0; a.pEnd = in->P;
                ed.tmp = in->preScratch[j];
            }
            void *dPass = nullptr, *dDesc = nullptr;
            rc = uploadTransient(in, pass.data(), (size_t)n * sizeof(OpDesc), &dPass); if (rc) break;
            rc = uploadTransient(in, &viaPrune[q], (size_t)n * sizeof(mi355::EdgeDesc), &dDesc); if (rc) break;
            mi355::launchPruneLevelTiled(in->stream, (const OpDesc*)dPass, n, in->matrices, in->P, in->S, in->C, false);
            mi355::launchEdgeReduce(in->stream, (const mi355::EdgeDesc*)dDesc, n, in->weights + (size_t)wIdx * in->C, in->patternWeights,
                                    dPer, dBlock, in->P, in->S, in->C, in->tiled);
        }
        if (rc) break;
        mi355::launchEdgeFinal(in->stream, dBlock, m, in->P, dSums);
        sums.resize((size_t)m * 2);
        rc = download(in, sums.data(), dSums, sums.size() * sizeof(double)); if (rc) break;
        for (int e = 0; e < m; e++) {
            if (outSum) outSum[b + e] = sums[2 * e];
            if (outSumSquared) outSumSquared[b + e] = sums[2 * e + 1];
        }
        if (outDerivatives) rc = download(in, outDerivatives + (size_t)b * in->P, dPer, (size_t)m * in->P * sizeof(double));
    }
    hipStreamSynchronize(in->stream);
    hipFree(dBlock); hipFree(dSums); if (dPer) hipFree(dPer);
    return rc;
}

// calculateCrossProductDifferentials (semantics: include/beagle_mi355.h)
int crossProducts(Instance* in, const int* postIdx, const int* preIdx, int rateIdx, int wIdx, const double* lengths, int count, double* outSum) {
    if (count <= 0) return 0;
    if (in->partitionCount != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    std::vector<int> need;
    for (int e = 0; e < count; e++) {
        if (badIndex(postIdx[e], in->partialsCount) || badIndex(preIdx[e], in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (isVirt(in, postIdx[e])) need.push_back(postIdx[e]);
        if (isVirt(in, preIdx[e])) need.push_back(preIdx[e]);
    }
    if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    std::vector<mi355::EdgeDesc> descs(count);
    for (int e = 0; e < count; e++) {
        const int po = postIdx[e], pr = preIdx[e];
        mi355::EdgeDesc& d = descs[e];
        if (in->tipStates[po] && po < in->tipCount) { d.post = in->tipStates[po]; d.postIsStates = 1; }
        else if (in->partials[po]) { d.post = in->partials[po]; d.postIsStates = 0; }
        else return BEAGLE_ERROR_OUT_OF_RANGE;
        if (!in->partials[pr] || (in->tipStates[pr] && pr < in->tipCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        d.pre = in->partials[pr];
    }
    const size_t nOut = (size_t)in->S * in->S;
    const int nb = mi355::edgeBlocks(in->P);
    double *dPartial = nullptr, *dOut = nullptr;
    HIP_TRY(hipMalloc((void**)&dPartial, (size_t)nb * nOut * sizeof(double)));
    if (hipMalloc((void**)&dOut, nOut * sizeof(double)) != hipSuccess) { hipFree(dPartial); return BEAGLE_ERROR_OUT_OF_MEMORY; }
    std::vector<double> sums(nOut);
    int rc = 0;
    const size_t maxChunk = (RING_BYTES / 8) / sizeof(mi355::EdgeDesc);
    for (size_t b = 0; b < (size_t)count && !rc; b += maxChunk) {
        const size_t n = std::min(maxChunk, (size_t)count - b);
        void *dDesc = nullptr, *dLen = nullptr;
        rc = uploadTransient(in, &descs[b], n * sizeof(mi355::EdgeDesc), &dDesc); if (rc) break;
        rc = uploadTransient(in, lengths + b, n * sizeof(double), &dLen); if (rc) break;
        mi355::launchCrossProducts(in->stream, (const mi355::EdgeDesc*)dDesc, (int)n, (const double*)dLen, in->weights + (size_t)wIdx * in->C,
                                   in->rates + (size_t)rateIdx * in->C, in->patternWeights, dPartial, dOut, in->P, in->S, in->C, in->tiled);
        rc = download(in, sums.data(), dOut, nOut * sizeof(double)); if (rc) break;
        for (size_t k = 0; k < nOut; k++) outSum[k] += sums[k];
    }
    hipStreamSynchronize(in->stream);
    hipFree(dPartial); hipFree(dOut);
    return rc;
}

int accumulate(Instance* in, const int* idx, int count, int cum, double sign, int part) {
    if (badIndex(cum, in->scaleCount) || badIndex(part, in->partitionCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, cum); if (rc) return rc;
    rc = ensureScale(in, cum); if (rc) return rc;
    if (in->scaleIsRaw[cum]) return BEAGLE_ERROR_OUT_OF_RANGE;
    std::vector<const double*> srcs(count);
    std::vector<int> raw(count);
    for (int k = 0; k < count; k++) {
        if (badIndex(idx[k], in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        rc = ensureScale(in, idx[k]); if (rc) return rc;
        srcs[k] = in->scale[idx[k]]; raw[k] = in->scaleIsRaw[idx[k]];
    }
    const int chunk = 4096;
    for (int b = 0; b < count; b += chunk) {
        const int n = std::min(chunk, count - b);
        void *dSrc = nullptr, *dRaw = nullptr;
        rc = uploadTransient(in, &srcs[b], (size_t)n * sizeof(double*), &dSrc); if (rc) return rc;
        rc = uploadTransient(in, &raw[b], (size_t)n * sizeof(int), &dRaw); if (rc) return rc;
        mi355::launchAccumulateScale(in->stream, in->scale[cum], (const double* const*)dSrc, (const int*)dRaw, n, sign,
                                     in->partStart[part], in->partEnd[part]);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int rootEnqueue(Instance* in, int rootIdx, int wIdx, int fIdx, int cumIdx, int part, double* dOut,
                unsigned long long* flag = nullptr, unsigned long long seq = 0) {
    // part < 0: the whole pattern range
    if (badIndex(rootIdx, in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    { int rcv = materializeVirtual(in, rootIdx); if (rcv) return rcv; }
    if (!in->partials[rootIdx] || badIndex(wIdx, in->eigenCount) ||
        badIndex(fIdx, in->eigenCount) || (part >= 0 && badIndex(part, in->partitionCount))) return BEAGLE_ERROR_OUT_OF_RANGE;
    const int pStart = part < 0 ? 0 : in->partStart[part], pEnd = part < 0 ? in->P : in->partEnd[part];
    if (pEnd <= pStart) {                                 // an empty partition (a shard that holds none of its patterns) contributes 0
        HIP_TRY(hipMemsetAsync(dOut, 0, sizeof(double), in->stream));
        return 0;
    }
    const double* cum = nullptr; int cumRaw = 0;
    if (cumIdx != BEAGLE_OP_NONE) {
        if (badIndex(cumIdx, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        int rc = ensureScale(in, cumIdx); if (rc) return rc;
        cum = in->scale[cumIdx]; cumRaw = in->scaleIsRaw[cumIdx];
    }
    if (in->tiled) {
        mi355::launchRootSiteTiled(in->stream, in->partials[rootIdx], in->weights + (size_t)wIdx * in->C,
                                   in->freqs + (size_t)fIdx * in->S, cum, cumRaw, in->patternWeights, in->siteLogL,
                                   in->blockSums, in->P, in->S, in->C, pStart, pEnd);
        mi355::launchRootFinal(in->stream, in->blockSums, (pEnd - pStart + 255) / 256, dOut, flag, seq);
    } else {
        mi355::launchRootLogLikelihood(in->stream, in->partials[rootIdx], in->weights + (size_t)wIdx * in->C,
                                       in->freqs + (size_t)fIdx * in->S, cum, cumRaw, in->patternWeights, in->siteLogL,
                                       in->blockSums, dOut, in->P, in->S, in->C, pStart, pEnd, flag, seq);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// API layout double[C][P][S]  <->  T32 layout double[C][tile][S][32] (kernels_mfma.hip); padded patterns are zero
void toTiled(const Instance* in, const double* api, double* tiled, int categories) {
    const size_t S = in->S, P = in->P, nt = in->ntile;
    std::fill(tiled, tiled + (size_t)categories * nt * S * 32, 0.0);
    for (int c = 0; c < categories; c++)
        for (size_t p = 0; p < P; p++) {
            const double* src = api + ((size_t)c * P + p) * S;
            double* dst = tiled + ((size_t)c * nt + p / 32) * S * 32 + p % 32;
            for (size_t j = 0; j < S; j++) dst[j * 32] = src[j];
        }
}

}  // namespace

// per-partition root sums of ONE (single-GPU) instance left on the device: deviceOut[k], k < partitionCount
static int rootByPartitionDevice(int instance, const int* bufferIndices, const int* categoryWeightsIndices, const int* stateFrequenciesIndices,
                                 const int* cumulativeScaleIndices, const int* partitionIndices, int partitionCount, double* deviceOut) {
    GET_INSTANCE(instance);
    for (int k = 0; k < partitionCount; k++) {
        int rc = rootEnqueue(in, bufferIndices[k], categoryWeightsIndices[k], stateFrequenciesIndices[k], cumulativeScaleIndices[k],
                             partitionIndices[k], deviceOut + k);
        if (rc) return rc;
    }
    return 0;
}

extern "C" {

const char* beagleGetVersion(void) { return "4.0.0-mi355"; }

const char* beagleGetCitation(void) {
    return "MI355X-native tree-likelihood engine behind the beagle.Beagle surface (gfx950 HIP kernels).\n"
           "API after: Ayres et al. (2019) BEAGLE 3, Systematic Biology 68:1052-1061.";
}

BeagleResourceList* beagleGetResourceList(void) { return &resources()->rl; }

// -beagle_auto: a full-tree evaluation of a synthetic alignment of the caller's shape on every candidate resource.
// Balanced tree over `tipCount` compact tips with pseudo-random states, one stochastic matrix on every branch (no eigen
// system needed: setTransitionMatrix), rescaling as the benchmark flags ask; 2 warm-up + 5 timed evaluations.
BeagleBenchmarkedResourceList* beagleGetBenchmarkedResourceList(int tipCount, int compactBufferCount, int stateCount, int patternCount,
                                      int categoryCount, const int* resourceList, int resourceCount, long preferenceFlags,
                                      long requirementFlags, int eigenModelCount, int partitionCount, int calculateDerivatives,
                                      long benchmarkFlags) {
    (void)compactBufferCount; (void)eigenModelCount; (void)partitionCount; (void)calculateDerivatives;
    static std::mutex mu;
    static std::vector<BeagleBenchmarkedResource> entries;
    static std::vector<std::string> strings;
    static BeagleBenchmarkedResourceList out;
    std::lock_guard<std::mutex> lock(mu);
    Resources* res = resources();
    std::vector<int> candidates;
    if (resourceList && resourceCount > 0) { for (int i = 0; i < resourceCount; i++) if (resourceList[i] >= 1 && resourceList[i] < res->rl.length) candidates.push_back(resourceList[i]); }
    else for (int r = 1; r < res->rl.length; r++) candidates.push_back(r);
    entries.clear(); strings.clear();
    strings.reserve(candidates.size() * 3 + 1);
    const int T = std::max(2, tipCount), S = stateCount, P = std::max(1, patternCount), C = std::max(1, categoryCount);
    const bool always = (benchmarkFlags & BEAGLE_BENCHFLAG_SCALING_ALWAYS) != 0;
    for (int r : candidates) {
        BeagleBenchmarkedResource e;
        memset(&e, 0, sizeof(e));
        e.number = r; e.name = res->rl.list[r].name; e.description = res->rl.list[r].description;
        e.supportFlags = res->rl.list[r].supportFlags; e.requiredFlags = 0; e.benchedFlags = benchmarkFlags;
        BeagleInstanceDetails det = {0, nullptr, nullptr, nullptr, 0};
        const int h = beagleCreateInstance(T, T + (T - 1), T, S, P, 1, 2 * T, C, always ? T : 0, &r, 1, preferenceFlags, requirementFlags, &det);
        e.returnCode = h < 0 ? h : 0;
        strings.push_back(det.implName ? det.implName : "");
        e.implName = (char*)strings.back().c_str();
        e.benchmarkResult = 0.0;
        if (h >= 0) {
            int rc = 0;
            std::vector<int> st(P);
            unsigned long long x = 88172645463325252ull;
            for (int t = 0; t < T && !rc; t++) {
                for (int p = 0; p < P; p++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; st[p] = (int)(x % (unsigned)S); }
                rc = beagleSetTipStates(h, t, st.data());
            }
            std::vector<double> m((size_t)C * S * S), w(C, 1.0 / C), f(S, 1.0 / S), pw(P, 1.0);
            for (int c = 0; c < C; c++) for (int i = 0; i < S; i++) for (int j = 0; j < S; j++)
                m[((size_t)c * S + i) * S + j] = i == j ? 0.9 - 0.05 * c / C : (0.1 + 0.05 * c / C) / (S - 1);
            for (int b = 0; b < 2 * T - 1 && !rc; b++) rc = beagleSetTransitionMatrix(h, b, m.data(), 0.0);
            if (!rc) rc = beagleSetCategoryWeights(h, 0, w.data());
            if (!rc) rc = beagleSetStateFrequencies(h, 0, f.data());
            if (!rc) rc = beagleSetPatternWeights(h, pw.data());
            // balanced tree: nodes 0..T-1 tips; internal node T+k joins the two oldest unjoined nodes
            std::vector<int> ops, scaleIdx;
            std::vector<int> queue(T);
            for (int t = 0; t < T; t++) queue[t] = t;
            size_t head = 0;
            for (int k = 0; k < T - 1; k++) {
                const int a = queue[head++], b = queue[head++], d = T + k;
                ops.insert(ops.end(), {d, always ? k : BEAGLE_OP_NONE, BEAGLE_OP_NONE, a, a, b, b});
                scaleIdx.push_back(k);
                queue.push_back(d);
            }
            const int root = 2 * T - 2, cum = always ? T - 1 : BEAGLE_OP_NONE, zero = 0;
            double lnl = 0.0, best = 1e300;
            for (int rep = 0; rep < 7 && !rc; rep++) {
                const auto t0 = std::chrono::steady_clock::now();
                rc = beagleUpdatePartials(h, ops.data(), T - 1, BEAGLE_OP_NONE);
                if (!rc && always) { rc = beagleResetScaleFactors(h, cum); if (!rc) rc = beagleAccumulateScaleFactors(h, scaleIdx.data(), T - 1, cum); }
                if (!rc) rc = beagleCalculateRootLogLikelihoods(h, &root, &zero, &zero, &cum, 1, &lnl);
                if (rc == BEAGLE_ERROR_FLOATING_POINT) rc = 0;
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rep >= 2) best = std::min(best, ms);
            }
            e.returnCode = rc;
            e.benchmarkResult = rc ? 0.0 : best;
            beagleFinalizeInstance(h);
        }
        entries.push_back(e);
    }
    std::stable_sort(entries.begin(), entries.end(), [](const BeagleBenchmarkedResource& a, const BeagleBenchmarkedResource& b) {
        const bool oa = a.returnCode == 0 && a.benchmarkResult > 0, ob = b.returnCode == 0 && b.benchmarkResult > 0;
        if (oa != ob) return oa;
        return a.benchmarkResult < b.benchmarkResult; });
    const double fastest = !entries.empty() && entries[0].benchmarkResult > 0 ? entries[0].benchmarkResult : 1.0;
    for (auto& e : entries) e.performanceRatio = e.benchmarkResult > 0 ? e.benchmarkResult / fastest : 0.0;
    out.list = entries.data(); out.length = (int)entries.size();
    return &out;
}

int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                         int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                         int scaleBufferCount, const int* resourceList, int resourceCount,
                         long preferenceFlags, long requirementFlags, BeagleInstanceDetails* returnInfo) {
    if (tipCount < 0 || partialsBufferCount < 1 || compactBufferCount < 0 || stateCount < 2 || stateCount > 255 ||
        patternCount < 1 || eigenBufferCount < 0 || matrixBufferCount < 0 || categoryCount < 1 || scaleBufferCount < 0)
        return BEAGLE_ERROR_OUT_OF_RANGE;
    // more than 64 states: no kernel of this engine is built for it (the MFMA path tiles up to 64, the gradient kernels
    // accumulate 64 x 64 outputs, the general kernels stage S x S doubles in LDS) — refuse loudly instead of half-working
    if (stateCount > 64) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    // requirement flags this engine cannot honour
    if (requirementFlags & (BEAGLE_FLAG_PRECISION_SINGLE | BEAGLE_FLAG_PROCESSOR_CPU |
                            BEAGLE_FLAG_FRAMEWORK_CPU | BEAGLE_FLAG_FRAMEWORK_CUDA | BEAGLE_FLAG_FRAMEWORK_OPENCL |
                            BEAGLE_FLAG_SCALING_AUTO | BEAGLE_FLAG_VECTOR_SSE))
        return BEAGLE_ERROR_NO_RESOURCE;
    Resources* res = resources();
    int device = -1;
    if (resourceList == nullptr || resourceCount <= 0) {
        if (res->gpuCount > 0) device = 0;
    } else {
        for (int i = 0; i < resourceCount && device < 0; i++) {
            if (resourceList[i] >= 1 && resourceList[i] <= res->gpuCount) device = resourceList[i] - 1;
            else if (res->gpuCount > 0 && resourceList[i] == res->gpuCount + 1)       // "all GPUs": the pattern-sharded instance
                return mi355::shardedCreate(res->gpuCount, tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount,
                                            eigenBufferCount, matrixBufferCount, categoryCount, scaleBufferCount, preferenceFlags,
                                            requirementFlags, returnInfo);
        }
    }
    if (device < 0) return BEAGLE_ERROR_NO_RESOURCE;
    if (hipSetDevice(device) != hipSuccess) return BEAGLE_ERROR_NO_RESOURCE;

    Instance* in = new Instance();
    in->device = device;
    in->tipCount = tipCount; in->partialsCount = partialsBufferCount; in->compactCount = compactBufferCount;
    in->S = stateCount; in->P = patternCount; in->eigenCount = std::max(1, eigenBufferCount);
    // BEAST adds EIGEN_COMPLEX to the flags whenever the substitution model may have complex eigenvalues (the asymmetric
    // discrete-trait models: BeagleTreeLikelihood.java:353-355, BeagleDataLikelihoodDelegate.java:378); every eigen system of
    // such an instance then arrives in real block form with 2 S eigenvalue entries (ComplexSubstitutionModel.java:121-173)
    in->eigenComplex = (requirementFlags & BEAGLE_FLAG_EIGEN_COMPLEX) != 0;      // (a REQUIREMENT in both callers; a mere preference keeps EIGEN_REAL)
    (void)preferenceFlags;
    in->matrixCount = matrixBufferCount; in->C = categoryCount; in->scaleCount = scaleBufferCount;
    // 16..64 states: T32 layout + fp64 MFMA kernels (amino acids, codons); BEAGLE_MI355_NO_MFMA=1 keeps the VALU kernel
    in->tiled = stateCount >= 16 && stateCount <= 64 && !(getenv("BEAGLE_MI355_NO_MFMA") && atoi(getenv("BEAGLE_MI355_NO_MFMA")) != 0);
    in->ntile = (patternCount + 31) / 32;
    in->schedAlap = !(getenv("BEAGLE_MI355_SCHED") && strcmp(getenv("BEAGLE_MI355_SCHED"), "asap") == 0);
    // 4 states (nucleotides), up to 16 rate categories: the pattern walk.  BEAGLE_MI355_NO_VIRTUAL=1 keeps every buffer real,
    // BEAGLE_MI355_VSTEPS=n caps the length of a virtual definition (A/B runs).
    in->walk = stateCount == 4 && categoryCount <= 16 &&
               (size_t)categoryCount * patternCount * 32 < ((size_t)1 << 32);     // the kernel addresses a buffer with 32-bit lane offsets
    const bool noVirtual = getenv("BEAGLE_MI355_NO_VIRTUAL") && atoi(getenv("BEAGLE_MI355_NO_VIRTUAL")) != 0;
    // T32 instances with <= 20 states: tip-tip nodes ("cherries") are defined, not stored — their parent's kernel rebuilds
    // them from the tips' states (kernels_mfma.hip cherryOperands); a definition is ONE step here
    in->cherry = in->tiled && stateCount <= 20 && !noVirtual;
    const bool virtualOn = (in->walk && !noVirtual) || in->cherry;
    in->virt = virtualOn;
    // Size of a virtual definition (internal nodes; any subtree shape whose evaluation needs at most two hold slots).
    // Evaluations at alignment sizes that keep the chip busy are bound by the bytes of the STORED nodes and their time
    // follows the cap (config A, profiles/r02_experiments.txt: cap 8 -> 207 stored nodes; 16 -> 112, 0.70 ms; 24 -> 78,
    // 0.64 ms; 32 -> 62, 0.63 ms) while a branch move — which re-evaluates the virtual siblings it passes instead of reading
    // 32 C P bytes each — costs 139 / 141 / 162 us at 16 / 24 / 32.  Small alignments are latency-bound: there the extra
    // micro-operations of long definitions show (12 500 patterns: branch move 65 -> 71 us from cap 8 to 16).
    int maxVirtSteps = (size_t)categoryCount * patternCount * 32 >= ((size_t)2 << 20) ? 24 : 8;
    if (getenv("BEAGLE_MI355_VSTEPS")) maxVirtSteps = std::max(1, std::min(mi355::PLAN_MAX_STEPS, atoi(getenv("BEAGLE_MI355_VSTEPS"))));
    if (in->cherry) maxVirtSteps = 1;
    in->planner.init(partialsBufferCount, tipCount, matrixBufferCount, scaleBufferCount, maxVirtSteps, virtualOn,
                     getenv("BEAGLE_MI355_HOLD_SLOTS") ? std::min(atoi(getenv("BEAGLE_MI355_HOLD_SLOTS")), mi355::walkHoldSlots(categoryCount)) : mi355::walkHoldSlots(categoryCount));
    in->planner.cacheEnabled = !(getenv("BEAGLE_MI355_NO_PLAN_CACHE") && atoi(getenv("BEAGLE_MI355_NO_PLAN_CACHE")) != 0);
    in->fastWalk = !(getenv("BEAGLE_MI355_NO_FAST_WALK") && atoi(getenv("BEAGLE_MI355_NO_FAST_WALK")) != 0);
    in->strictWaits = !(getenv("BEAGLE_MI355_STRICT_WAITS") && atoi(getenv("BEAGLE_MI355_STRICT_WAITS")) == 0);
    // matrix storage: the caller's buffers, then the private snapshot slots of virtual definitions (planner.h)
    size_t matrixSlots = std::max<size_t>(std::max<size_t>(1, matrixBufferCount), (size_t)in->planner.matrixSlots());
    if (in->tiled) { in->preIdentity = (int)matrixSlots; in->preTransposed = (int)matrixSlots + 1; matrixSlots += 1 + PRE_SCRATCH; }
    const size_t patternSlots = in->tiled ? (size_t)in->ntile * 32 : (size_t)patternCount;
    in->partialsBytes = (((size_t)categoryCount * patternSlots * stateCount * sizeof(double)) + 255 + (in->walk ? 256 : 0)) & ~(size_t)255;
    in->partials.assign(partialsBufferCount, nullptr);
    in->tipStates.assign(partialsBufferCount, nullptr);
    in->scale.assign(std::max(1, scaleBufferCount), nullptr);
    in->scaleIsRaw.assign(std::max(1, scaleBufferCount), 0);
    in->partStart.assign(1, 0); in->partEnd.assign(1, patternCount);
    setPairLayout(in);                                                    // one partition: whole blocks of 128 patterns
    in->wStamp.assign(partialsBufferCount, 0); in->wLevel.assign(partialsBufferCount, 0); in->wOp.assign(partialsBufferCount, 0);
    in->rStamp.assign(partialsBufferCount, 0); in->rLevel.assign(partialsBufferCount, 0);
    in->resourceName = res->names[device + 1];

    bool ok = hipStreamCreateWithFlags(&in->ownStream, hipStreamNonBlocking) == hipSuccess;
    in->stream = in->ownStream;
    ok = ok && hipHostMalloc((void**)&in->hRing, RING_BYTES, hipHostMallocDefault) == hipSuccess;
    // result words live in coherent, device-mapped host memory: the final reduction kernel writes the sum straight into it
    // and the host only waits for the stream (no device-to-host copy behind the last kernel)
    ok = ok && hipHostMalloc((void**)&in->hResult, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void**)&in->hResultDev, in->hResult, 0) == hipSuccess;
    const size_t S = stateCount, C = categoryCount, E = in->eigenCount;
    const int rootBlocks = (patternCount + 255) / 256;
    ok = ok && devAlloc(in, (void**)&in->dRing, RING_BYTES) == 0;
    ok = ok && devAlloc(in, (void**)&in->matrices, matrixSlots * C * S * S * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->eigen, E * (2 * S * S + 2 * S) * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->rates, E * C * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->weights, E * C * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->freqs, E * S * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->patternWeights, (size_t)patternCount * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->siteLogL, (size_t)patternCount * sizeof(double)) == 0;
    ok = ok && devAlloc(in, (void**)&in->blockSums, ((size_t)rootBlocks + 1024) * sizeof(double)) == 0;    // (+ one partial block per partition)
    ok = ok && devAlloc(in, (void**)&in->dResult, 4096) == 0;
    if (ok) {
        // defaults: category rates 1, weights 1/C, pattern weights 1 (beagle.jar!GeneralBeagleImpl#<init>)
        std::vector<double> ones(std::max<size_t>((size_t)patternCount, E * C), 1.0);
        ok = upload(in, in->rates, ones.data(), E * C * sizeof(double)) == 0;
        ok = ok && upload(in, in->patternWeights, ones.data(), (size_t)patternCount * sizeof(double)) == 0;
        std::vector<double> w(E * C, 1.0 / (double)C);
        ok = ok && upload(in, in->weights, w.data(), E * C * sizeof(double)) == 0;
        ok = ok && hipMemsetAsync(in->matrices, 0, matrixSlots * C * S * S * sizeof(double), in->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(in->siteLogL, 0, (size_t)patternCount * sizeof(double), in->stream) == hipSuccess;
        if (ok && in->tiled) {   // identity matrix for the two-pass pre-order path
            std::vector<double> eye(C * S * S, 0.0);
            for (size_t c = 0; c < C; c++) for (size_t i = 0; i < S; i++) eye[c * S * S + i * S + i] = 1.0;
            ok = upload(in, in->matrices + (size_t)in->preIdentity * C * S * S, eye.data(), eye.size() * sizeof(double)) == 0;
        }
    }
    if (!ok) { destroy(in); return BEAGLE_ERROR_OUT_OF_MEMORY; }

    int handle = -1;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        for (size_t i = 0; i < g_instances.size(); i++) if (!g_instances[i]) { handle = (int)i; break; }
        if (handle < 0) { g_instances.push_back(nullptr); handle = (int)g_instances.size() - 1; }
        g_instances[handle] = in;
    }
    if (returnInfo) {
        returnInfo->resourceNumber = device + 1;
        returnInfo->resourceName = (char*)in->resourceName.c_str();
        returnInfo->implName = (char*)"HIP-gfx950-fp64";
        returnInfo->implDescription = (char*)"hand-written CDNA4 kernels, level-batched pruning";
        returnInfo->flags = GPU_FLAGS & ~(BEAGLE_FLAG_SCALING_ALWAYS | BEAGLE_FLAG_SCALING_DYNAMIC) &
                            ~(in->eigenComplex ? BEAGLE_FLAG_EIGEN_REAL : BEAGLE_FLAG_EIGEN_COMPLEX);
    }
    return handle;
}

int beagleFinalizeInstance(int instance) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedFinalize(instance); }
    Instance* in = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (instance < 0 || instance >= (int)g_instances.size() || !g_instances[instance]) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
        in = g_instances[instance];
        g_instances[instance] = nullptr;
    }
    destroy(in);
    return BEAGLE_SUCCESS;
}

int beagleSetCPUThreadCount(int instance, int threadCount) {
    if (mi355::isShardedHandle(instance)) { return BEAGLE_SUCCESS; }
    (void)threadCount;
    return lookup(instance) ? BEAGLE_SUCCESS : BEAGLE_ERROR_UNINITIALIZED_INSTANCE;   // no-op on a GPU instance
}

int beagleSetPatternWeights(int instance, const double* w) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedSetPerPatternDoubles(instance, w, 1, 1, [&](int h, const double* v) { return beagleSetPatternWeights(h, v); }); }
    GET_INSTANCE(instance);
    return upload(in, in->patternWeights, w, (size_t)in->P * sizeof(double));
}

int beagleSetPatternPartitions(int instance, int partitionCount, const int* partitions) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedSetPerPatternInts(instance, partitions, [&](int h, const int* v) { return beagleSetPatternPartitions(h, partitionCount, v); }); }
    GET_INSTANCE(instance);
    if (partitionCount < 1) return BEAGLE_ERROR_OUT_OF_RANGE;
    for (int x = 0; x < in->partialsCount; x++) { int rcv = materializeVirtual(in, x); if (rcv) return rcv; }   // whole-range definitions
    // partitions are contiguous pattern ranges in concatenation order
    // (MultiPartitionDataLikelihoodDelegate.java:520-535); anything else is rejected
    std::vector<int> s(partitionCount, -1), e(partitionCount, -1);
    for (int p = 0; p < in->P; p++) {
        const int k = partitions[p];
        if (badIndex(k, partitionCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (s[k] < 0) s[k] = p;
        else if (e[k] != p) return BEAGLE_ERROR_NO_IMPLEMENTATION;   // not contiguous
        e[k] = p + 1;
    }
    for (int k = 0; k < partitionCount; k++) if (s[k] < 0) { s[k] = 0; e[k] = 0; }
    in->partitionCount = partitionCount; in->partStart = s; in->partEnd = e; in->resolveEpoch++;
    if (in->walk) in->planner.setPartitionCount(partitionCount);
    if (in->walk && in->virt) {
        // definitions are kept per (buffer, partition): more snapshot slots behind the caller's matrices
        const size_t per = (size_t)in->C * in->S * in->S, slots = std::max<size_t>(std::max<size_t>(1, in->matrixCount), (size_t)in->planner.matrixSlots());
        double* grown = nullptr;
        int rcm = devAlloc(in, (void**)&grown, slots * per * sizeof(double)); if (rcm) return rcm;
        HIP_TRY(hipMemsetAsync(grown, 0, slots * per * sizeof(double), in->stream));
        HIP_TRY(hipMemcpyAsync(grown, in->matrices, (size_t)std::max(1, in->matrixCount) * per * sizeof(double), hipMemcpyDeviceToDevice, in->stream));
        in->matrices = grown;                              // (the old block stays owned by the instance until it is destroyed)
    }
    if (in->walk) {
        // the pair-interleaved arrays follow the partitions (Instance::pairPos): what exists already — tips are uploaded before
        // this call, MultiPartitionDataLikelihoodDelegate.java:544-553 — moves to the new layout on the device
        setPairLayout(in);
        if (!in->dPairPos) { int rc = devAlloc(in, (void**)&in->dPairPos, (size_t)in->P * sizeof(unsigned)); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(in->stream));
        HIP_TRY(hipMemcpy(in->dPairPos, in->pairPos.data(), (size_t)in->P * sizeof(unsigned), hipMemcpyHostToDevice));
        in->stateSlabLeft = 0; in->scaleSlabLeft = 0;                      // new slabs: the element sizes changed
        for (int t = 0; t < in->partialsCount; t++) {
            uint8_t* old = in->tipStates[t];
            if (!old) continue;
            in->tipStates[t] = nullptr;
            int rc = ensureStates(in, t); if (rc) return rc;
            HIP_TRY(hipMemsetAsync(in->tipStates[t] + in->statePairOff, in->S, in->pairLen, in->stream));
            mi355::launchRelayoutStates(in->stream, old, in->tipStates[t], in->tipStates[t] + in->statePairOff, in->dPairPos, in->P);
        }
        for (int k = 0; k < (int)in->scale.size(); k++) {
            double* old = in->scale[k];
            if (!old) continue;
            const char raw = in->scaleIsRaw[k];
            in->scale[k] = nullptr;
            int rc = ensureScale(in, k); if (rc) return rc;                // (zero-filled)
            in->scaleIsRaw[k] = raw;
            HIP_TRY(hipMemcpyAsync(in->scale[k], old, (size_t)in->P * sizeof(double), hipMemcpyDeviceToDevice, in->stream));
            if (raw) mi355::launchRecipFromFactors(in->stream, in->scale[k], in->scale[k] + in->scaleStride, in->dPairPos, in->P);
        }
        in->dummyTips = nullptr; in->onesScale = nullptr;                  // re-made at their new sizes on first use
        HIP_TRY(hipGetLastError());
    }
    const size_t n = (size_t)in->partialsCount * partitionCount;
    in->wStamp.assign(n, 0); in->wLevel.assign(n, 0); in->rStamp.assign(n, 0); in->rLevel.assign(n, 0); in->wOp.assign(n, 0);
    in->stamp = 0;
    return BEAGLE_SUCCESS;
}

int beagleSetTipStates(int instance, int tipIndex, const int* inStates) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedSetPerPatternInts(instance, inStates, [&](int h, const int* v) { return beagleSetTipStates(h, tipIndex, v); }); }
    GET_INSTANCE(instance);
    if (badIndex(tipIndex, in->tipCount) || badIndex(tipIndex, in->partialsCount) || tipIndex >= in->compactCount)
        return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeTipUsers(in, tipIndex); if (rc) return rc;   // virtual cherries defined by the OLD states
    rc = ensureStates(in, tipIndex); if (rc) return rc;
    setCompact(in, tipIndex, true);
    if (!in->walk) {
        std::vector<uint8_t> s(in->P);
        for (int p = 0; p < in->P; p++) s[p] = (inStates[p] >= 0 && inStates[p] < in->S) ? (uint8_t)inStates[p] : (uint8_t)in->S;
        return upload(in, in->tipStates[tipIndex], s.data(), (size_t)in->P);
    }
    // walk instances: plain states (pre-order kernels, getTipStates), then the pair-interleaved copy the walk reads; the
    // padding of the last block of 128 is "missing"
    std::vector<uint8_t> s(in->statePairOff + in->pairLen, (uint8_t)in->S);
    for (int p = 0; p < in->P; p++) {
        const uint8_t v = (inStates[p] >= 0 && inStates[p] < in->S) ? (uint8_t)inStates[p] : (uint8_t)in->S;
        s[p] = v; s[in->statePairOff + in->pairPos[p]] = v;
    }
    return upload(in, in->tipStates[tipIndex], s.data(), s.size());
}

int beagleGetTipStates(int instance, int tipIndex, int* outStates) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedGetPerPatternInts(instance, outStates, [&](int h, int* v) { return beagleGetTipStates(h, tipIndex, v); }); }
    GET_INSTANCE(instance);
    if (badIndex(tipIndex, in->partialsCount) || !in->tipStates[tipIndex]) return BEAGLE_ERROR_OUT_OF_RANGE;
    std::vector<uint8_t> s(in->P);
    int rc = download(in, s.data(), in->tipStates[tipIndex], (size_t)in->P); if (rc) return rc;
    for (int p = 0; p < in->P; p++) outStates[p] = s[p];
    return BEAGLE_SUCCESS;
}

int beagleSetTipPartials(int instance, int tipIndex, const double* inPartials) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedSetPerPatternDoubles(instance, inPartials, shardedStates(instance), 1, [&](int h, const double* v) { return beagleSetTipPartials(h, tipIndex, v); }); }
    GET_INSTANCE(instance);
    if (badIndex(tipIndex, in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeTipUsers(in, tipIndex); if (rc) return rc;
    clearVirtual(in, tipIndex);
    rc = ensurePartials(in, tipIndex); if (rc) return rc;
    const size_t n = (size_t)in->P * in->S * sizeof(double);
    if (in->tiled) {
        const size_t plane = (size_t)in->ntile * 32 * in->S;
        std::vector<double> t(plane * in->C);
        toTiled(in, inPartials, t.data(), 1);
        for (int c = 1; c < in->C; c++) memcpy(&t[plane * c], &t[0], plane * sizeof(double));
        rc = upload(in, in->partials[tipIndex], t.data(), t.size() * sizeof(double));
    } else if (in->C == 1) { rc = upload(in, in->partials[tipIndex], inPartials, n); }
    else {
        // upload one category plane to the LAST plane, replicate it into all planes on the device
        double* last = in->partials[tipIndex] + (size_t)(in->C - 1) * in->P * in->S;
        rc = upload(in, last, inPartials, n);
        if (!rc) mi355::launchReplicateCategories(in->stream, last, in->partials[tipIndex], in->P, in->S, in->C - 1);
    }
    in->tipStates[tipIndex] = nullptr;   // the buffer now holds partials (slab memory stays owned by the instance)
    setCompact(in, tipIndex, false);
    setLeaf(in, tipIndex);               // ... that no operation computes: definitions may read them (planner.h leafPartials)
    return rc;
}

int beagleSetPartials(int instance, int bufferIndex, const double* inPartials) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedSetPerPatternDoubles(instance, inPartials, shardedStates(instance), shardedCategories(instance), [&](int h, const double* v) { return beagleSetPartials(h, bufferIndex, v); }); }
    GET_INSTANCE(instance);
    if (badIndex(bufferIndex, in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeTipUsers(in, bufferIndex); if (rc) return rc;
    clearVirtual(in, bufferIndex);
    rc = ensurePartials(in, bufferIndex); if (rc) return rc;
    in->tipStates[bufferIndex] = nullptr; setCompact(in, bufferIndex, false);
    setLeaf(in, bufferIndex);
    if (in->tiled) {
        std::vector<double> t((size_t)in->C * in->ntile * 32 * in->S);
        toTiled(in, inPartials, t.data(), in->C);
        return upload(in, in->partials[bufferIndex], t.data(), t.size() * sizeof(double));
    }
    return upload(in, in->partials[bufferIndex], inPartials, (size_t)in->C * in->P * in->S * sizeof(double));
}

// Read-back of `count` partials buffers (AncestralStateBeagleTreeLikelihood.java:414-542 reads every
// internal node once per logged sample): virtual buffers are materialised by ONE walk, every buffer is converted to the
// API layout [C][P][S] on the device with its scale factors folded in, and the device-to-host copies stream through a
// pinned bounce buffer, a chunk of buffers at a time, with one synchronisation per chunk.
// host-side copy of a chunk out of the pinned bounce buffer, on a few threads when it is large: the destination is the
// caller's array, usually touched for the first time here, and faulting its pages in is what bounds a single thread
// (profiles/r02_readback.json: the 1.28 GB sweep ran at 8 GB/s, below the node-by-node loop)
struct HostCopy {
    std::vector<std::thread> th;
    void start(char* dst, const char* src, size_t bytes) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t k = bytes < ((size_t)8 << 20) ? 1 : std::min<size_t>(8, std::max<size_t>(1, hw / 2));
        if (k == 1) { memcpy(dst, src, bytes); return; }
        const size_t per = ((bytes / k) + 4095) & ~(size_t)4095;
        for (size_t i = 0; i * per < bytes; i++)
            th.emplace_back([=] { memcpy(dst + i * per, src + i * per, std::min(per, bytes - i * per)); });
    }
    void join() { for (auto& t : th) t.join(); th.clear(); }
    ~HostCopy() { join(); }
};

// out == nullptr (count must fit one chunk): the data is left in the pinned buffer exportHost[0] (beagleMi355GetPartialsPinned)
static int exportPartials(Instance* in, const int* bufferIndices, const int* scaleIndices, int count, double* out) {
    std::vector<int> need;
    for (int k = 0; k < count; k++) {
        const int b = bufferIndices[k];
        if (badIndex(b, in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (scaleIndices && scaleIndices[k] != BEAGLE_OP_NONE && badIndex(scaleIndices[k], in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (isVirt(in, b)) in->planner.keysOf(b, need);
    }
    if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    const size_t elems = (size_t)in->C * in->P * in->S, bytes = elems * sizeof(double);
    // chunks of about 32 MiB, two in flight: while the device converts and copies chunk k + 1, the host empties chunk k
    const size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)32 << 20) / bytes));
    if (!out && (size_t)count > chunk) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (in->exportBytes < chunk * bytes) {
        HIP_TRY(hipStreamSynchronize(in->stream));
        for (int k = 0; k < 2; k++) {
            if (in->exportDev[k]) hipFree(in->exportDev[k]);
            if (in->exportHost[k]) hipHostFree(in->exportHost[k]);
            in->exportDev[k] = nullptr; in->exportHost[k] = nullptr;
        }
        in->exportBytes = 0;
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipMalloc((void**)&in->exportDev[k], chunk * bytes));
            HIP_TRY(hipHostMalloc((void**)&in->exportHost[k], chunk * bytes, hipHostMallocDefault));
            if (!in->exportEvent[k]) HIP_TRY(hipEventCreateWithFlags(&in->exportEvent[k], hipEventDisableTiming));
        }
        in->exportBytes = chunk * bytes;
    }
    HostCopy copies[2];
    const size_t nChunks = ((size_t)count + chunk - 1) / chunk;
    auto chunkCount = [&](size_t c) { return std::min(chunk, (size_t)count - c * chunk); };
    for (size_t c = 0; c < nChunks; c++) {
        const int w = (int)(c & 1);
        copies[w].join();                                  // the host copy that was reading exportHost[w] (chunk c - 2)
        const size_t n = chunkCount(c);
        for (size_t k = 0; k < n; k++) {
            const int b = bufferIndices[c * chunk + k];
            if (!in->partials[b] || isCompactTip(in, b)) return BEAGLE_ERROR_OUT_OF_RANGE;
            const double* sc = nullptr; int raw = 0;
            if (scaleIndices && scaleIndices[c * chunk + k] != BEAGLE_OP_NONE) {
                int rc = ensureScale(in, scaleIndices[c * chunk + k]); if (rc) return rc;
                sc = in->scale[scaleIndices[c * chunk + k]]; raw = in->scaleIsRaw[scaleIndices[c * chunk + k]];
            }
            mi355::launchExportPartials(in->stream, in->partials[b], sc, raw, in->exportDev[w] + k * elems, in->P, in->S, in->C, in->tiled);
        }
        HIP_TRY(hipMemcpyAsync(in->exportHost[w], in->exportDev[w], n * bytes, hipMemcpyDeviceToHost, in->stream));
        HIP_TRY(hipEventRecord(in->exportEvent[w], in->stream));
        if (c >= 1 && out) {                               // chunk c - 1 has landed (or lands while this one is being produced)
            HIP_TRY(hipEventSynchronize(in->exportEvent[1 - w]));
            copies[1 - w].start((char*)(out + (c - 1) * chunk * elems), (const char*)in->exportHost[1 - w], chunkCount(c - 1) * bytes);
        }
    }
    const int last = (int)((nChunks - 1) & 1);
    HIP_TRY(hipEventSynchronize(in->exportEvent[last]));
    in->ringHead = 0;
    if (out) copies[last].start((char*)(out + (nChunks - 1) * chunk * elems), (const char*)in->exportHost[last], chunkCount(nChunks - 1) * bytes);
    copies[0].join(); copies[1].join();
    return BEAGLE_SUCCESS;
}

int beagleGetPartials(int instance, int bufferIndex, int scaleIndex, double* outPartials) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedGetPerPatternDoubles(instance, outPartials, shardedStates(instance), shardedCategories(instance), [&](int h, double* v) { return beagleGetPartials(h, bufferIndex, scaleIndex, v); }); }
    GET_INSTANCE(instance);
    return exportPartials(in, &bufferIndex, &scaleIndex, 1, outPartials);
}

// MI355X extension: `count` buffers in one call, out = [count][C][P][S]; scaleIndices may be NULL
int beagleMi355GetPartialsBatch(int instance, const int* bufferIndices, const int* scaleIndices, int count, double* outPartials) {
    if (mi355::isShardedHandle(instance)) {
        const size_t elems = (size_t)shardedCategories(instance) * mi355::shardedPatternCount(instance) * shardedStates(instance);
        for (int k = 0; k < count; k++) {
            const int rc = beagleGetPartials(instance, bufferIndices[k], scaleIndices ? scaleIndices[k] : BEAGLE_OP_NONE, outPartials + (size_t)k * elems);
            if (rc) return rc;
        }
        return BEAGLE_SUCCESS;
    }
    GET_INSTANCE(instance);
    if (count <= 0) return BEAGLE_SUCCESS;
    return exportPartials(in, bufferIndices, scaleIndices, count, outPartials);
}

// MI355X extensions for the JNI shim: the result stays in the engine's pinned bounce buffer (valid until the next call on the
// instance) and goes from there into the Java array with ONE copy.  Not for the sharded instance (NO_IMPLEMENTATION: the
// shim then takes the ordinary entry point).
int beagleMi355GetPartialsPinned(int instance, int bufferIndex, int scaleIndex, const double** outPinned, long* outCount) {
    if (mi355::isShardedHandle(instance)) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    GET_INSTANCE(instance);
    if (!outPinned || !outCount) return BEAGLE_ERROR_OUT_OF_RANGE;
    const int rc = exportPartials(in, &bufferIndex, &scaleIndex, 1, nullptr);
    if (rc) return rc;
    *outPinned = in->exportHost[0]; *outCount = (long)in->C * in->P * in->S;
    return BEAGLE_SUCCESS;
}
int beagleMi355GetSiteLogLikelihoodsPinned(int instance, const double** outPinned, long* outCount) {
    if (mi355::isShardedHandle(instance)) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    GET_INSTANCE(instance);
    if (!outPinned || !outCount) return BEAGLE_ERROR_OUT_OF_RANGE;
    const size_t bytes = (size_t)in->P * sizeof(double);
    if (bytes > RING_BYTES) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    HIP_TRY(hipMemcpyAsync(in->hRing, in->siteLogL, bytes, hipMemcpyDeviceToHost, in->stream));     // (the ring is pinned; everything staged in it
    HIP_TRY(hipStreamSynchronize(in->stream));                                                      //  has been consumed once the stream is idle)
    in->ringHead = (bytes + 255) & ~(size_t)255;
    *outPinned = (const double*)in->hRing; *outCount = in->P;
    return BEAGLE_SUCCESS;
}

int beagleGetLogScaleFactors(int instance, int scaleIndex, double* out) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedGetPerPatternDoubles(instance, out, 1, 1, [&](int h, double* v) { return beagleGetLogScaleFactors(h, scaleIndex, v); }); }
    GET_INSTANCE(instance);
    if (badIndex(scaleIndex, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = ensureScale(in, scaleIndex); if (rc) return rc;
    rc = download(in, out, in->scale[scaleIndex], (size_t)in->P * sizeof(double)); if (rc) return rc;
    if (in->scaleIsRaw[scaleIndex]) for (int p = 0; p < in->P; p++) out[p] = log(out[p]);
    return BEAGLE_SUCCESS;
}

// Small model arrays are re-sent by BEAST before every evaluation whether they changed or not (frequencies and category
// weights right before calculateRootLogLikelihoods, BeagleTreeLikelihood.java:1029-1030, i.e. behind the last pruning
// kernel in stream order): an identical value is not uploaded again.
static int uploadIfChanged(Instance* in, std::vector<double>& shadow, std::vector<char>& ok, int count, int idx, size_t n,
                           double* dst, const double* src) {
    if (shadow.empty()) { shadow.assign((size_t)count * n, 0.0); ok.assign(count, 0); }
    double* sh = &shadow[(size_t)idx * n];
    if (ok[idx] && memcmp(sh, src, n * sizeof(double)) == 0) return 0;
    memcpy(sh, src, n * sizeof(double));
    ok[idx] = 1;
    return upload(in, dst, src, n * sizeof(double));
}

int beagleSetEigenDecomposition(int instance, int eigenIndex, const double* U, const double* Uinv, const double* lambda) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetEigenDecomposition(h, eigenIndex, U, Uinv, lambda); }); }
    GET_INSTANCE(instance);
    if (badIndex(eigenIndex, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    const size_t S = in->S, nLambda = in->eigenComplex ? 2 * S : S, stride = 2 * S * S + nLambda;
    std::vector<double> pack(stride);
    memcpy(&pack[0], U, S * S * sizeof(double));
    memcpy(&pack[S * S], Uinv, S * S * sizeof(double));
    memcpy(&pack[2 * S * S], lambda, nLambda * sizeof(double));
    return uploadIfChanged(in, in->shEigen, in->okEigen, in->eigenCount, eigenIndex, stride, in->eigen + stride * eigenIndex, pack.data());
}

int beagleSetStateFrequencies(int instance, int idx, const double* f) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetStateFrequencies(h, idx, f); }); }
    GET_INSTANCE(instance);
    if (badIndex(idx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return uploadIfChanged(in, in->shFreqs, in->okFreqs, in->eigenCount, idx, in->S, in->freqs + (size_t)idx * in->S, f);
}

int beagleSetCategoryWeights(int instance, int idx, const double* w) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetCategoryWeights(h, idx, w); }); }
    GET_INSTANCE(instance);
    if (badIndex(idx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return uploadIfChanged(in, in->shWeights, in->okWeights, in->eigenCount, idx, in->C, in->weights + (size_t)idx * in->C, w);
}

int beagleSetCategoryRatesWithIndex(int instance, int idx, const double* r) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetCategoryRatesWithIndex(h, idx, r); }); }
    GET_INSTANCE(instance);
    if (badIndex(idx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return uploadIfChanged(in, in->shRates, in->okRates, in->eigenCount, idx, in->C, in->rates + (size_t)idx * in->C, r);
}

int beagleSetCategoryRates(int instance, const double* r) { return beagleSetCategoryRatesWithIndex(instance, 0, r); }

int beagleSetTransitionMatrix(int instance, int matrixIndex, const double* inMatrix, double paddedValue) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetTransitionMatrix(h, matrixIndex, inMatrix, paddedValue); }); }
    (void)paddedValue;
    GET_INSTANCE(instance);
    if (badIndex(matrixIndex, in->matrixCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    const size_t n = (size_t)in->C * in->S * in->S;
    return upload(in, in->matrices + n * matrixIndex, inMatrix, n * sizeof(double));
}

int beagleGetTransitionMatrix(int instance, int matrixIndex, double* outMatrix) {
    if (mi355::isShardedHandle(instance)) { bool first = true; std::mutex mu; return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleGetTransitionMatrix(h, matrixIndex, outMatrix); }); }
    GET_INSTANCE(instance);
    if (badIndex(matrixIndex, in->matrixCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    const size_t n = (size_t)in->C * in->S * in->S;
    return download(in, outMatrix, in->matrices + n * matrixIndex, n * sizeof(double));
}

int beagleConvolveTransitionMatrices(int instance, const int* first, const int* second, const int* result, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleConvolveTransitionMatrices(h, first, second, result, count); }); }
    GET_INSTANCE(instance);
    if (count <= 0) return BEAGLE_SUCCESS;
    for (int k = 0; k < count; k++) {
        if (badIndex(first[k], in->matrixCount) || badIndex(second[k], in->matrixCount) || badIndex(result[k], in->matrixCount) ||
            result[k] == first[k] || result[k] == second[k]) return BEAGLE_ERROR_OUT_OF_RANGE;
    }
    // a result may feed a later triple of the same call (epoch chains): run dependent triples in order
    int b = 0;
    while (b < count) {
        int e = b + 1;
        for (; e < count; e++) {
            bool dep = false;
            for (int k = b; k < e && !dep; k++)
                dep = result[k] == first[e] || result[k] == second[e] || result[k] == result[e] ||
                      first[k] == result[e] || second[k] == result[e];
            if (dep) break;
        }
        const int n = e - b;
        void *dF, *dS, *dR;
        int rc = uploadTransient(in, first + b, n * sizeof(int), &dF); if (rc) return rc;
        rc = uploadTransient(in, second + b, n * sizeof(int), &dS); if (rc) return rc;
        rc = uploadTransient(in, result + b, n * sizeof(int), &dR); if (rc) return rc;
        mi355::launchConvolveMatrices(in->stream, in->matrices, (const int*)dF, (const int*)dS, (const int*)dR, n, in->S, in->C);
        b = e;
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

static int transitionMatrices(Instance* in, const int* eigenIdx, int eigenScalar, const int* rateIdx,
                              const int* probIdx, const double* lens, int count) {
    if (count <= 0) return BEAGLE_SUCCESS;
    std::vector<int> eig(count), rate(count);
    for (int k = 0; k < count; k++) {
        eig[k] = eigenIdx ? eigenIdx[k] : eigenScalar;
        rate[k] = rateIdx ? rateIdx[k] : 0;
        if (badIndex(probIdx[k], in->matrixCount) || badIndex(eig[k], in->eigenCount) || badIndex(rate[k], in->eigenCount))
            return BEAGLE_ERROR_OUT_OF_RANGE;
    }
    // one packed upload: [lengths double[count] | matrix idx | eigen idx | rate idx] (each copy is a blit kernel)
    std::vector<char> pack((size_t)count * (sizeof(double) + 3 * sizeof(int)));
    double* pLen = (double*)pack.data();
    int* pIdx = (int*)(pack.data() + (size_t)count * sizeof(double));
    memcpy(pLen, lens, (size_t)count * sizeof(double));
    memcpy(pIdx, probIdx, (size_t)count * sizeof(int));
    memcpy(pIdx + count, eig.data(), (size_t)count * sizeof(int));
    memcpy(pIdx + 2 * (size_t)count, rate.data(), (size_t)count * sizeof(int));
    void* dPack;
    int rc = uploadTransient(in, pack.data(), pack.size(), &dPack); if (rc) return rc;
    const double* dLen = (const double*)dPack;
    const int* dIdx = (const int*)((const char*)dPack + (size_t)count * sizeof(double));
    mi355::launchTransitionMatrices(in->stream, in->matrices, in->eigen, in->rates, dIdx, dLen,
                                    dIdx + count, dIdx + 2 * (size_t)count, count, in->S, in->C, in->eigenComplex);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices,
                                   const int* firstDerivativeIndices, const int* secondDerivativeIndices,
                                   const double* edgeLengths, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdateTransitionMatrices(h, eigenIndex, probabilityIndices, firstDerivativeIndices, secondDerivativeIndices, edgeLengths, count); }); }
    GET_INSTANCE(instance);
    if (firstDerivativeIndices || secondDerivativeIndices) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    return transitionMatrices(in, nullptr, eigenIndex, nullptr, probabilityIndices, edgeLengths, count);
}

int beagleUpdateTransitionMatricesWithMultipleModels(int instance, const int* eigenIndices, const int* categoryRateIndices,
                                   const int* probabilityIndices, const int* firstDerivativeIndices,
                                   const int* secondDerivativeIndices, const double* edgeLengths, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdateTransitionMatricesWithMultipleModels(h, eigenIndices, categoryRateIndices, probabilityIndices, firstDerivativeIndices, secondDerivativeIndices, edgeLengths, count); }); }
    GET_INSTANCE(instance);
    if (firstDerivativeIndices || secondDerivativeIndices) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    if (!eigenIndices || !categoryRateIndices) return BEAGLE_ERROR_OUT_OF_RANGE;
    return transitionMatrices(in, eigenIndices, 0, categoryRateIndices, probabilityIndices, edgeLengths, count);
}

int beagleUpdatePartials(int instance, const int* operations, int operationCount, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdatePartials(h, operations, operationCount, cumulativeScaleIndex); }); }
    GET_INSTANCE(instance);
    return runOperations(in, operations, operationCount, BEAGLE_OP_COUNT, cumulativeScaleIndex);
}

int beagleUpdatePartialsByPartition(int instance, const int* operations, int operationCount) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdatePartialsByPartition(h, operations, operationCount); }); }
    GET_INSTANCE(instance);
    return runOperations(in, operations, operationCount, BEAGLE_PARTITION_OP_COUNT, BEAGLE_OP_NONE);
}

int beagleWaitForPartials(int instance, const int* destinationPartials, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleWaitForPartials(h, destinationPartials, count); }); }
    (void)destinationPartials; (void)count;
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    in->ringHead = 0;
    return BEAGLE_SUCCESS;
}

int beagleAccumulateScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleAccumulateScaleFactors(h, scaleIndices, count, cumulativeScaleIndex); }); }
    GET_INSTANCE(instance);
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, 1.0, 0);
}
int beagleAccumulateScaleFactorsByPartition(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex, int partitionIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleAccumulateScaleFactorsByPartition(h, scaleIndices, count, cumulativeScaleIndex, partitionIndex); }); }
    GET_INSTANCE(instance);
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, 1.0, partitionIndex);
}
int beagleRemoveScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleRemoveScaleFactors(h, scaleIndices, count, cumulativeScaleIndex); }); }
    GET_INSTANCE(instance);
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, -1.0, 0);
}
int beagleRemoveScaleFactorsByPartition(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex, int partitionIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleRemoveScaleFactorsByPartition(h, scaleIndices, count, cumulativeScaleIndex, partitionIndex); }); }
    GET_INSTANCE(instance);
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, -1.0, partitionIndex);
}

int beagleResetScaleFactorsByPartition(int instance, int cumulativeScaleIndex, int partitionIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleResetScaleFactorsByPartition(h, cumulativeScaleIndex, partitionIndex); }); }
    GET_INSTANCE(instance);
    if (badIndex(cumulativeScaleIndex, in->scaleCount) || badIndex(partitionIndex, in->partitionCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, cumulativeScaleIndex); if (rc) return rc;
    rc = ensureScale(in, cumulativeScaleIndex); if (rc) return rc;
    if (in->scaleIsRaw[cumulativeScaleIndex] && in->partitionCount > 1) {
        // a per-node (raw) buffer is being recycled as a cumulative one: clear all of it first
        mi355::launchFill(in->stream, in->scale[cumulativeScaleIndex], 0.0, 0, in->P);
    }
    if (in->scaleIsRaw[cumulativeScaleIndex]) in->resolveEpoch++;    // kept programs were validated against the raw flags
    in->scaleIsRaw[cumulativeScaleIndex] = 0;
    mi355::launchFill(in->stream, in->scale[cumulativeScaleIndex], 0.0, in->partStart[partitionIndex], in->partEnd[partitionIndex]);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}
int beagleResetScaleFactors(int instance, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleResetScaleFactors(h, cumulativeScaleIndex); }); }
    GET_INSTANCE(instance);
    if (badIndex(cumulativeScaleIndex, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, cumulativeScaleIndex); if (rc) return rc;
    rc = ensureScale(in, cumulativeScaleIndex); if (rc) return rc;
    if (in->scaleIsRaw[cumulativeScaleIndex]) in->resolveEpoch++;
    in->scaleIsRaw[cumulativeScaleIndex] = 0;
    mi355::launchFill(in->stream, in->scale[cumulativeScaleIndex], 0.0, 0, in->P);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleCopyScaleFactors(int instance, int dest, int src) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleCopyScaleFactors(h, dest, src); }); }
    GET_INSTANCE(instance);
    if (badIndex(dest, in->scaleCount) || badIndex(src, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, dest); if (rc) return rc;
    rc = ensureScale(in, dest); if (rc) return rc;
    rc = ensureScale(in, src); if (rc) return rc;
    const size_t scaleDoubles = in->walk ? 2 * in->scaleStride : (size_t)in->P;      // walk instances: factors and reciprocals
    HIP_TRY(hipMemcpyAsync(in->scale[dest], in->scale[src], scaleDoubles * sizeof(double), hipMemcpyDeviceToDevice, in->stream));
    if (in->scaleIsRaw[dest] != in->scaleIsRaw[src]) in->resolveEpoch++;
    in->scaleIsRaw[dest] = in->scaleIsRaw[src];
    return BEAGLE_SUCCESS;
}

int beagleCalculateRootLogLikelihoods(int instance, const int* bufferIndices, const int* categoryWeightsIndices,
                                      const int* stateFrequenciesIndices, const int* cumulativeScaleIndices,
                                      int count, double* outSumLogLikelihood) {
    if (mi355::isShardedHandle(instance)) { if (count != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;
        double v = 0.0;
        const int rc = mi355::shardedRootReduce(instance, 1, [&](int h, double* dOut) { return beagleMi355CalculateRootLogLikelihoodsDevice(h, bufferIndices[0],
                              categoryWeightsIndices[0], stateFrequenciesIndices[0], cumulativeScaleIndices[0], dOut); }, &v);
        if (rc) return rc;
        *outSumLogLikelihood = v;
        return (v != v) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS; }
    GET_INSTANCE(instance);
    if (count != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;   // BEAST always passes 1 (BeagleTreeLikelihood.java:1038)
    // the reduction kernel writes the sum and then a sequence number into mapped host memory; the kernel is the last
    // thing in the (in-order) stream, so seeing the number means everything before it has completed
    const unsigned long long seq = ++in->resultSeq;
    int rc = rootEnqueue(in, bufferIndices[0], categoryWeightsIndices[0], stateFrequenciesIndices[0],
                         cumulativeScaleIndices[0], -1, in->hResultDev, (unsigned long long*)(in->hResultDev + 8), seq);
    if (rc) return rc;
    {
        volatile unsigned long long* flag = (volatile unsigned long long*)(in->hResult + 8);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (*flag != seq) HIP_TRY(hipStreamSynchronize(in->stream));     // long evaluation (or an error): block instead of spinning
        if (*flag != seq) return BEAGLE_ERROR_GENERAL;
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    in->ringHead = 0;   // everything staged so far has been consumed
    const double v = in->hResult[0];
    *outSumLogLikelihood = v;
    return (v != v) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS;
}

int beagleCalculateRootLogLikelihoodsByPartition(int instance, const int* bufferIndices, const int* categoryWeightsIndices,
                                      const int* stateFrequenciesIndices, const int* cumulativeScaleIndices,
                                      const int* partitionIndices, int partitionCount, int count,
                                      double* outByPartition, double* outSum) {
    if (mi355::isShardedHandle(instance)) { if (count != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;
        const int rc = mi355::shardedRootReduce(instance, partitionCount, [&](int h, double* dOut) { return rootByPartitionDevice(h, bufferIndices, categoryWeightsIndices,
                              stateFrequenciesIndices, cumulativeScaleIndices, partitionIndices, partitionCount, dOut); }, outByPartition);
        if (rc) return rc;
        double tot = 0.0;
        for (int k = 0; k < partitionCount; k++) tot += outByPartition[k];
        *outSum = tot;
        return (tot != tot) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS; }
    GET_INSTANCE(instance);
    if (count != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    if (partitionCount < 1 || partitionCount > 512) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (!in->tiled && partitionCount <= 480) {
        // all partitions in ONE pair of launches per eight of them, the sums written straight into mapped host memory behind a
        // sequence word the host polls (as calculateRootLogLikelihoods): no device-to-host copy, no stream synchronisation
        std::vector<mi355::RootParts> chunks((partitionCount + mi355::ROOT_MAX_PARTS - 1) / mi355::ROOT_MAX_PARTS);
        int blockOff = 0;
        for (int k = 0; k < partitionCount; k++) {
            const int rootIdx = bufferIndices[k], wIdx = categoryWeightsIndices[k], fIdx = stateFrequenciesIndices[k], cumIdx = cumulativeScaleIndices[k], part = partitionIndices[k];
            if (badIndex(rootIdx, in->partialsCount) || badIndex(part, in->partitionCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
            { int rcv = materializeVirtual(in, rootIdx); if (rcv) return rcv; }
            if (!in->partials[rootIdx] || badIndex(wIdx, in->eigenCount) || badIndex(fIdx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
            mi355::RootParts& ch = chunks[k / mi355::ROOT_MAX_PARTS];
            mi355::RootPart& q = ch.p[k % mi355::ROOT_MAX_PARTS];
            ch.n = k % mi355::ROOT_MAX_PARTS + 1;
            q.root = in->partials[rootIdx]; q.catWeights = in->weights + (size_t)wIdx * in->C; q.freqs = in->freqs + (size_t)fIdx * in->S;
            q.cum = nullptr; q.cumIsRaw = 0;
            if (cumIdx != BEAGLE_OP_NONE) {
                if (badIndex(cumIdx, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
                int rc = ensureScale(in, cumIdx); if (rc) return rc;
                q.cum = in->scale[cumIdx]; q.cumIsRaw = in->scaleIsRaw[cumIdx];
            }
            q.pStart = in->partStart[part]; q.pEnd = in->partEnd[part]; q.blockOff = blockOff;
            blockOff += (std::max(0, q.pEnd - q.pStart) + 255) / 256;
        }
        const unsigned long long seq = ++in->resultSeq;
        for (size_t c = 0; c < chunks.size(); c++) {
            const bool last = c + 1 == chunks.size();
            mi355::launchRootLogLikelihoodParts(in->stream, chunks[c], in->patternWeights, in->siteLogL, in->blockSums,
                                                in->hResultDev + 16 + c * mi355::ROOT_MAX_PARTS, in->P, in->S, in->C,
                                                last ? (unsigned long long*)(in->hResultDev + 8) : nullptr, seq);
        }
        HIP_TRY(hipGetLastError());
        volatile unsigned long long* flag = (volatile unsigned long long*)(in->hResult + 8);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (*flag != seq) HIP_TRY(hipStreamSynchronize(in->stream));
        if (*flag != seq) return BEAGLE_ERROR_GENERAL;
        std::atomic_thread_fence(std::memory_order_acquire);
        in->ringHead = 0;
        double tot = 0.0;
        for (int k = 0; k < partitionCount; k++) { outByPartition[k] = in->hResult[16 + k]; tot += in->hResult[16 + k]; }
        *outSum = tot;
        return (tot != tot) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS;
    }
    for (int k = 0; k < partitionCount; k++) {
        int rc = rootEnqueue(in, bufferIndices[k], categoryWeightsIndices[k], stateFrequenciesIndices[k],
                             cumulativeScaleIndices[k], partitionIndices[k], in->dResult + k);
        if (rc) return rc;
    }
    int rc = download(in, in->hResult, in->dResult, (size_t)partitionCount * sizeof(double));
    if (rc) return rc;
    double tot = 0.0;
    for (int k = 0; k < partitionCount; k++) { outByPartition[k] = in->hResult[k]; tot += in->hResult[k]; }
    *outSum = tot;
    return (tot != tot) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS;
}

int beagleGetSiteLogLikelihoods(int instance, double* out) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedGetPerPatternDoubles(instance, out, 1, 1, [&](int h, double* v) { return beagleGetSiteLogLikelihoods(h, v); }); }
    GET_INSTANCE(instance);
    return download(in, out, in->siteLogL, (size_t)in->P * sizeof(double));
}

// ---- outside SURVEY 8 (a)-(e): exported so the JNI shim links ---------------------------------
// ---- pre-order partials and branch gradients ----
int beagleSetRootPrePartials(int instance, const int* bufferIndices, const int* stateFrequenciesIndices, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetRootPrePartials(h, bufferIndices, stateFrequenciesIndices, count); }); }
    GET_INSTANCE(instance);
    for (int k = 0; k < count; k++) {
        const int b = bufferIndices[k], f = stateFrequenciesIndices[k];
        if (badIndex(b, in->partialsCount) || badIndex(f, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        int rc = materializeTipUsers(in, b); if (rc) return rc;
        clearVirtual(in, b);
        rc = ensurePartials(in, b); if (rc) return rc;
        in->tipStates[b] = nullptr; setCompact(in, b, false);
        mi355::launchFillFrequencies(in->stream, in->partials[b], in->freqs + (size_t)f * in->S, in->P, in->S, in->C, in->tiled);
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleSetDifferentialMatrix(int instance, int matrixIndex, const double* inMatrix) {
    return beagleSetTransitionMatrix(instance, matrixIndex, inMatrix, 0.0);
}

int beagleAddTransitionMatrices(int, const int*, const int*, const int*, int) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }

int beagleTransposeTransitionMatrices(int instance, const int* inputIndices, const int* resultIndices, int matrixCount) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleTransposeTransitionMatrices(h, inputIndices, resultIndices, matrixCount); }); }
    GET_INSTANCE(instance);
    if (matrixCount <= 0) return BEAGLE_SUCCESS;
    std::vector<int> pairs((size_t)matrixCount * 2);
    for (int k = 0; k < matrixCount; k++) {
        if (badIndex(inputIndices[k], in->matrixCount) || badIndex(resultIndices[k], in->matrixCount) || inputIndices[k] == resultIndices[k])
            return BEAGLE_ERROR_OUT_OF_RANGE;
        pairs[2 * k] = inputIndices[k]; pairs[2 * k + 1] = resultIndices[k];
    }
    void* dPairs = nullptr;
    int rc = uploadTransient(in, pairs.data(), pairs.size() * sizeof(int), &dPairs); if (rc) return rc;
    mi355::launchTransposeMatrices(in->stream, in->matrices, (const int*)dPairs, matrixCount, in->S, in->C);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleUpdatePrePartials(int instance, const int* operations, int operationCount, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdatePrePartials(h, operations, operationCount, cumulativeScaleIndex); }); }
    GET_INSTANCE(instance);
    return runPreOperations(in, operations, operationCount, cumulativeScaleIndex);
}

int beagleCalculateCrossProductDifferentials(int instance, const int* postBufferIndices, const int* preBufferIndices,
                                             const int* categoryRateIndices, const int* categoryWeightsIndices,
                                             const double* edgeLengths, int count,
                                             double* outSumDerivatives, double* outSumSquaredDerivatives) {
    if (mi355::isShardedHandle(instance)) {
        if (!outSumDerivatives) return BEAGLE_ERROR_OUT_OF_RANGE;
        const int len = shardedStates(instance) * shardedStates(instance);
        std::vector<double> tot(len, 0.0);
        const int rc = mi355::shardedSumDoubles(instance, len, [&](int h, double* out) { return beagleCalculateCrossProductDifferentials(h, postBufferIndices,
                              preBufferIndices, categoryRateIndices, categoryWeightsIndices, edgeLengths, count, out, outSumSquaredDerivatives); }, tot.data());
        for (int k = 0; k < len && !rc; k++) outSumDerivatives[k] += tot[k];
        return rc;
    }
    GET_INSTANCE(instance);
    if (outSumSquaredDerivatives) return BEAGLE_ERROR_NO_IMPLEMENTATION;       // BEAST passes null
    if (!postBufferIndices || !preBufferIndices || !categoryRateIndices || !categoryWeightsIndices || !edgeLengths || !outSumDerivatives)
        return BEAGLE_ERROR_OUT_OF_RANGE;
    if (badIndex(categoryRateIndices[0], in->eigenCount) || badIndex(categoryWeightsIndices[0], in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return crossProducts(in, postBufferIndices, preBufferIndices, categoryRateIndices[0], categoryWeightsIndices[0], edgeLengths, count,
                         outSumDerivatives);
}

int beagleCalculateEdgeDifferentials(int instance, const int* postBufferIndices, const int* preBufferIndices,
                                     const int* derivativeMatrixIndices, const int* categoryWeightsIndices, int count,
                                     double* outDerivatives, double* outSumDerivatives, double* outSumSquaredDerivatives) {
    if (mi355::isShardedHandle(instance)) {
        if (count <= 0) return BEAGLE_SUCCESS;
        const int P = mi355::shardedPatternCount(instance), n = mi355::shardedShardCount(instance);
        std::vector<std::vector<double>> per(n);
        std::vector<int> handleOf(n, -1);
        std::vector<double> tot((size_t)2 * count, 0.0);
        std::mutex mu; int next = 0;
        const int rc = mi355::shardedSumDoubles(instance, 2 * count, [&](int h, double* out) {
            int k; { std::lock_guard<std::mutex> l(mu); k = next++; handleOf[k] = h; }
            int a, b; mi355::shardedBoundsOfHandle(instance, h, &a, &b);
            if (outDerivatives) per[k].assign((size_t)count * (b - a), 0.0);
            const int r = beagleCalculateEdgeDifferentials(h, postBufferIndices, preBufferIndices, derivativeMatrixIndices, categoryWeightsIndices, count,
                                                           outDerivatives ? per[k].data() : nullptr, out, out + count);
            if (!r && outDerivatives)
                for (int e = 0; e < count; e++) memcpy(outDerivatives + (size_t)e * P + a, &per[k][(size_t)e * (b - a)], (size_t)(b - a) * sizeof(double));
            return r; }, tot.data());
        if (rc) return rc;
        for (int e = 0; e < count; e++) { if (outSumDerivatives) outSumDerivatives[e] = tot[e]; if (outSumSquaredDerivatives) outSumSquaredDerivatives[e] = tot[count + e]; }
        return BEAGLE_SUCCESS;
    }
    GET_INSTANCE(instance);
    if (!postBufferIndices || !preBufferIndices || !derivativeMatrixIndices || !categoryWeightsIndices) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (badIndex(categoryWeightsIndices[0], in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return edgeDifferentials(in, postBufferIndices, preBufferIndices, derivativeMatrixIndices, categoryWeightsIndices[0], count,
                             outDerivatives, outSumDerivatives, outSumSquaredDerivatives);
}

int beagleUpdatePrePartialsByPartition(int, const int*, int) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }

// ---- MI355X extensions -----------------------------------------------------------------------
int beagleMi355SetStream(int instance, void* hipStream) {
    if (mi355::isShardedHandle(instance)) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    in->ringHead = 0;
    in->stream = hipStream ? (hipStream_t)hipStream : in->ownStream;
    return BEAGLE_SUCCESS;
}

int beagleMi355CalculateRootLogLikelihoodsDevice(int instance, int bufferIndex, int categoryWeightsIndex,
                                                 int stateFrequenciesIndex, int cumulativeScaleIndex, void* deviceOut) {
    if (mi355::isShardedHandle(instance)) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }
    GET_INSTANCE(instance);
    if (!deviceOut) return BEAGLE_ERROR_OUT_OF_RANGE;
    return rootEnqueue(in, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex, cumulativeScaleIndex, -1, (double*)deviceOut);
}

int beagleMi355Synchronize(int instance) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleMi355Synchronize(h); }); }
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    in->ringHead = 0;
    return BEAGLE_SUCCESS;
}

int beagleMi355KernelTimer(int instance, int enable, double* outMillis, long* outLaunches) {
    if (mi355::isShardedHandle(instance)) {             // the slowest shard's kernel time, the launches of all
        std::mutex mu; double ms = 0.0; long launches = 0;
        const int rc = mi355::shardedBroadcast(instance, [&](int h) { double m = 0.0; long l = 0; const int r = beagleMi355KernelTimer(h, enable, &m, &l);
                                                                       std::lock_guard<std::mutex> g(mu); ms = std::max(ms, m); launches += l; return r; });
        if (outMillis) *outMillis = ms;
        if (outLaunches) *outLaunches = launches;
        return rc;
    }
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    in->ringHead = 0;
    for (size_t k = 0; k < in->eventsUsed; k++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, in->events[k].first, in->events[k].second) == hipSuccess) in->timedMs += ms;
    }
    in->timedLaunches += in->pendingLaunches;
    in->pendingLaunches = 0;
    in->eventsUsed = 0;
    if (outMillis) *outMillis = in->timedMs;
    if (outLaunches) *outLaunches = in->timedLaunches;
    in->timedMs = 0.0; in->timedLaunches = 0;
    in->statMicroOps = in->statStored = in->statMemReads = in->statTipReads = in->statScaleReads = in->statWalks = in->statScaleWrites = in->statFastWalks = 0;
    in->timing = enable != 0;
    // event pairs for the calls to come are created here, not inside the region being timed
    while (enable && in->events.size() < 1024) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        in->events.emplace_back(a, b);
    }
    return BEAGLE_SUCCESS;
}

int beagleMi355WalkStats(int instance, long* out8) {
    if (mi355::isShardedHandle(instance)) {             // counters of shard 0 (every shard runs the same programs)
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355WalkStats(h, out8); });
    }
    Instance* in = lookup(instance);
    if (!in || !out8) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    out8[0] = in->statMicroOps; out8[1] = in->statStored; out8[2] = in->statMemReads; out8[3] = in->statTipReads;
    out8[4] = in->statScaleReads; out8[5] = in->statWalks; out8[6] = in->statScaleWrites; out8[7] = in->statFastWalks;
    return BEAGLE_SUCCESS;
}

long beagleMi355DeviceBytes(int instance) {
    if (mi355::isShardedHandle(instance)) {
        std::mutex mu; long total = 0;
        mi355::shardedBroadcast(instance, [&](int h) { const long b = beagleMi355DeviceBytes(h); std::lock_guard<std::mutex> l(mu); total += b; return 0; });
        return total;
    }
    Instance* in = lookup(instance);
    return in ? (long)in->deviceBytes : -1;
}

static const BeagleApi g_api = {
    beagleGetVersion,
    beagleCreateInstance,
    beagleFinalizeInstance,
    beagleSetPatternWeights,
    beagleSetTipStates,
    beagleSetTipPartials,
    beagleSetPartials,
    beagleGetPartials,
    beagleGetLogScaleFactors,
    beagleSetEigenDecomposition,
    beagleSetStateFrequencies,
    beagleSetCategoryWeights,
    beagleSetCategoryRates,
    beagleSetTransitionMatrix,
    beagleGetTransitionMatrix,
    beagleUpdateTransitionMatrices,
    beagleUpdatePartials,
    beagleAccumulateScaleFactors,
    beagleRemoveScaleFactors,
    beagleResetScaleFactors,
    beagleCopyScaleFactors,
    beagleCalculateRootLogLikelihoods,
    beagleGetSiteLogLikelihoods,
    beagleMi355CalculateRootLogLikelihoodsDevice,
};
const BeagleApi* beagleGetApiTable(void) { return &g_api; }

}  // extern "C"
