// engine_abi.cpp — the C ABI of include/beagle_mi355.h over the engine's internals (engine_internal.h): argument checks,
// buffer bookkeeping, and the calls that observe results (root log-likelihoods, read-back).
#include "engine_internal.h"

using mi355::OpDesc;
using mi355::labEnv;
using mi355::shardedStates;
using mi355::shardedCategories;
using namespace mi355::eng;

namespace {

int accumulate(Instance* in, const int* idx, int count, int cum, double sign, int part) {
    if (badIndex(cum, in->scaleCount) || badIndex(part, in->partitionCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, cum); if (rc) return rc;
    rc = ensureScale(in, cum); if (rc) return rc;
    if (in->scaleIsRaw[cum]) return BEAGLE_ERROR_OUT_OF_RANGE;
    // exactly the scale buffers the last write-mode walk wrote, none of them touched since: a few dozen slice products instead of a factor
    // per node (Instance::lastSums)
    if (in->walk && in->sliceSums && in->partitionCount == 1 && in->lastSums.valid && in->lastSums.epoch == in->scaleWriteEpoch &&
        count > 0 && count == in->lastSums.nWritten) {
        const long stamp = ++in->seenCounter;
        bool same = true;
        for (int k = 0; k < count && same; k++) {
            if (badIndex(idx[k], in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
            same = in->scaleGen[(size_t)idx[k]] == in->lastSums.gen && in->scaleSeen[(size_t)idx[k]] != stamp;
            in->scaleSeen[(size_t)idx[k]] = stamp;
        }
        if (same) {
            void* dRows = nullptr;
            rc = uploadTransient(in, in->lastSums.rows.data(), in->lastSums.rows.size() * sizeof(int), &dRows); if (rc) return rc;
            mi355::launchAccumulateSlices(live(in), in->scale[cum], in->sliceMant, in->sliceExp, (const int*)dRows, (int)in->lastSums.rows.size(), in->pairLen,
                                          in->dPairPos, sign, in->partStart[part], in->partEnd[part]);
            in->statSliceAccum++;
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
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
        mi355::launchAccumulateScale(live(in), in->scale[cum], (const double* const*)dSrc, (const int*)dRaw, n, sign,
                                     in->partStart[part], in->partEnd[part]);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The site log-likelihoods of the root sum just enqueued, on their way to the host before anybody asks (Instance::hSites): called by the
// whole-alignment root entry points right behind their last launch.  A caller that read the site values after each of the last two such
// sums (BeagleTreeLikelihood.java:1050 does after every one) is served; the copy is a kernel of the stream (k_hostCopies writing through
// the buffer's device mapping: the next evaluation's launches queue behind 800 KB of PCIe writes at the metric's size, ~20 us, which the
// caller's own traversal covers), its end an event.
int sitePrefetchAfterRoot(Instance* in) {
    if (!in->siteReadSinceRoot) in->siteReadStreak = 0;            // the sum before this one: nobody looked at its site values
    in->siteReadSinceRoot = false;
    const size_t bytes = (size_t)in->P * sizeof(double);
    if (!in->sitePrefetch || in->siteReadStreak < 2 || bytes > ((size_t)1 << 30)) return 0;
    if (!in->hSites) {
        if (hipHostMalloc((void**)&in->hSites, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { in->hSites = nullptr; in->sitePrefetch = false; (void)hipGetLastError(); return 0; }
        if (hipHostGetDevicePointer((void**)&in->hSitesDev, in->hSites, 0) != hipSuccess ||
            hipEventCreateWithFlags(&in->siteEvent, hipEventDisableTiming) != hipSuccess) { in->sitePrefetch = false; (void)hipGetLastError(); return 0; }
    }
    const unsigned blocks = (unsigned)((bytes + 4095) / 4096);
    mi355::HostCopyList L;
    L.n = 1;
    L.e[0].dst = in->hSitesDev; L.e[0].src = (const char*)in->siteLogL; L.e[0].bytes = (unsigned)bytes; L.e[0].firstBlock = 0;
    mi355::launchHostCopies(live(in), L, (int)blocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(in->siteEvent, live(in)));
    in->sitePrefetched = true;
    return 0;
}
// ... and the other end: true when `out` (nullable: the caller reads Instance::hSites itself) has the site values of the last root sum
bool sitePrefetchTake(Instance* in, double* out) {
    if (!in->siteReadSinceRoot) { in->siteReadStreak++; in->siteReadSinceRoot = true; }
    if (!in->sitePrefetched) return false;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    hipError_t st;
    while ((st = hipEventQuery(in->siteEvent)) == hipErrorNotReady) {
        __builtin_ia32_pause();
        if ((++spins & 0xff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) { st = hipEventSynchronize(in->siteEvent); break; }
    }
    if (st != hipSuccess) { (void)hipGetLastError(); in->sitePrefetched = false; return false; }      // (the stream-ordered download then says what went wrong)
    if (out) memcpy(out, in->hSites, (size_t)in->P * sizeof(double));
    in->statSitePrefetched++;
    return true;
}

int rootEnqueue(Instance* in, int rootIdx, int wIdx, int fIdx, int cumIdx, int part, double* dOut,
                unsigned long long* flag = nullptr, unsigned long long seq = 0) {
    // part < 0: the whole pattern range
    in->sitePrefetched = false;                           // (whatever this sum writes into siteLogL, the host's copy is of the one before)
    if (badIndex(rootIdx, in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    { int rcv = materializeVirtual(in, rootIdx); if (rcv) return rcv; }
    if (!in->partials[rootIdx] || badIndex(wIdx, in->eigenCount) ||
        badIndex(fIdx, in->eigenCount) || (part >= 0 && badIndex(part, in->partitionCount))) return BEAGLE_ERROR_OUT_OF_RANGE;
    const int pStart = part < 0 ? 0 : in->partStart[part], pEnd = part < 0 ? in->P : in->partEnd[part];
    if (pEnd <= pStart) {                                 // an empty partition (a shard that holds none of its patterns) contributes 0
        HIP_TRY(hipMemsetAsync(dOut, 0, sizeof(double), live(in)));
        return 0;
    }
    const double* cum = nullptr; int cumRaw = 0;
    if (cumIdx != BEAGLE_OP_NONE) {
        if (badIndex(cumIdx, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        int rc = ensureScale(in, cumIdx); if (rc) return rc;
        cum = in->scale[cumIdx]; cumRaw = in->scaleIsRaw[cumIdx];
    }
    if (in->pendingWalk.valid && part < 0 && !in->tiled) {
        // the walk that computes this root is still held back: launch it with the root's slice finishing the evaluation
        const Instance::PendingWalk& pw = in->pendingWalk;
        int seg = -1;
        for (size_t i = 0; i < pw.finalStore.size(); i++) if (pw.finalStore[i] == rootIdx) seg = (int)i;
        if (seg >= 0) {
            mi355::RootFused rf;
            memset(&rf, 0, sizeof(rf));
            rf.catWeights = in->weights + (size_t)wIdx * in->C; rf.freqs = in->freqs + (size_t)fIdx * in->S; rf.cum = cum; rf.cumIsRaw = cumRaw;
            rf.patternWeights = in->patternWeights; rf.siteLogL = in->siteLogL; rf.blockSums = in->blockSums; rf.counter = in->rootCounter;
            rf.out = dOut; rf.flag = flag; rf.seq = seq; rf.rootSeg = seg; rf.groups = (in->P + 127) / 128;
            return flushWalk(in, &rf);
        }
    }
    if (in->tiled) {
        mi355::launchRootSiteTiled(live(in), in->partials[rootIdx], in->weights + (size_t)wIdx * in->C,
                                   in->freqs + (size_t)fIdx * in->S, cum, cumRaw, in->patternWeights, in->siteLogL,
                                   in->blockSums, in->P, in->S, in->C, pStart, pEnd);
        mi355::launchRootFinal(live(in), in->blockSums, mi355::rootSiteTiledBlocks(pEnd - pStart), dOut, flag, seq);
    } else if (in->walk && in->fuseLaunches) {
        mi355::launchRootLogLikelihood4W(live(in), in->partials[rootIdx], in->weights + (size_t)wIdx * in->C,
                                         in->freqs + (size_t)fIdx * in->S, cum, cumRaw, in->patternWeights, in->siteLogL,
                                         in->blockSums, dOut, in->P, in->C, pStart, pEnd, flag, seq, in->rootCounter);
    } else {
        mi355::launchRootLogLikelihood(live(in), in->partials[rootIdx], in->weights + (size_t)wIdx * in->C,
                                       in->freqs + (size_t)fIdx * in->S, cum, cumRaw, in->patternWeights, in->siteLogL,
                                       in->blockSums, dOut, in->P, in->S, in->C, pStart, pEnd, flag, seq, in->fuseLaunches ? in->rootCounter : nullptr);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The last reduction kernel of a call writes its result and then `seq` into mapped host memory (Instance::hResult); the kernel
// is the last thing in the (in-order) stream, so seeing the number means everything before it has completed.  Polled — a stream
// synchronisation costs a wake-up per evaluation — for 20 ms, then blocking (a long evaluation, another rank's, or an error).
int waitResult(Instance* in, unsigned long long seq) {
    volatile unsigned long long* flag = (volatile unsigned long long*)(in->hResult + 8);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*flag != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    if (*flag != seq) HIP_TRY(hipStreamSynchronize(live(in)));
    if (*flag != seq) return BEAGLE_ERROR_GENERAL;
    std::atomic_thread_fence(std::memory_order_acquire);
    { const int rc = in->asyncError.exchange(0); if (rc) return rc; }
    return 0;
}

// Matrix slots of an instance: the caller's, the planner's private snapshot slots behind them and — T32 layout — one identity
// matrix and PRE_SCRATCH transposed-matrix slots behind those (the two-pass pre-order path, engine_preorder.cpp).  Sets
// preIdentity / preTransposed; used at creation and whenever the planner's slot count changes (beagleSetPatternPartitions).
size_t matrixSlotLayout(Instance* in) {
    size_t slots = std::max<size_t>(std::max<size_t>(1, in->matrixCount), (size_t)in->planner.matrixSlots());
    if (in->tiled) { in->preIdentity = (int)slots; in->preTransposed = (int)slots + 1; slots += 1 + PRE_SCRATCH; }
    return slots;
}
int uploadIdentityMatrix(Instance* in) {
    const size_t S = in->S, C = in->C;
    std::vector<double> eye(C * S * S, 0.0);
    for (size_t c = 0; c < C; c++) for (size_t i = 0; i < S; i++) eye[c * S * S + i * S + i] = 1.0;
    return upload(in, in->matrices + (size_t)in->preIdentity * C * S * S, eye.data(), eye.size() * sizeof(double));
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

namespace mi355 {
int publishAndWait(int instance, const double* dValues, int count, double* out) {
    GET_INSTANCE_KEEP_PENDING(instance);
    if (!dValues || !out || count < 1 || count > 480) return BEAGLE_ERROR_OUT_OF_RANGE;
    const unsigned long long seq = ++in->resultSeq;
    mi355::launchPublish(live(in), dValues, count, in->hResultDev + 16, (unsigned long long*)(in->hResultDev + 8), seq);
    HIP_TRY(hipGetLastError());
    { const int rcw = waitResult(in, seq); if (rcw) return rcw; }
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
    memcpy(out, in->hResult + 16, (size_t)count * sizeof(double));
    return BEAGLE_SUCCESS;
}
int takeAsyncError(int instance) {
    Instance* in = lookup(instance);
    if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    return in->asyncError.exchange(0);
}
}  // namespace mi355

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
    // (65..255 states — the large discrete-trait state spaces of phylogeography, GeneralLikelihoodCore.java:41-50 — run the
    // likelihood path on the general kernels, which read their matrices from L2 above ~90 states instead of staging them in LDS
    // (kernels.hip k_pruneGeneral<false>, k_transitionBig); the pre-order / gradient entry points run there too since round 5
    // (kernels_preorder.hip k_prePartialsBig, k_edgeDifferentialsBig, k_crossProductsBig: correctness paths, as k_pruneGeneral))
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
    // level order of the level kernels (engine_levels.cpp): as late as possible up to 20 states (every launch mixes the write-only
    // tip-tip nodes with read-heavy ones), as early as possible above (61 states: 214 -> 226 evals/s, profiles/r03_experiments.txt 11);
    // BEAGLE_MI355_SCHED=asap|alap overrides
    in->schedAlap = stateCount <= 20;
    if (labEnv("BEAGLE_MI355_SCHED")) {              // (LAB builds only: dfs is twice as slow, profiles/r04_experiments.txt 10)
        const char* sc = labEnv("BEAGLE_MI355_SCHED");
        in->schedAlap = strcmp(sc, "asap") != 0;
        if (strncmp(sc, "dfs", 3) == 0) in->schedDfs = sc[3] == ':' ? std::max(1, atoi(sc + 4)) : 4;
    }
    // 4 states (nucleotides), up to 16 rate categories: the pattern walk.  BEAGLE_MI355_NO_VIRTUAL=1 keeps every buffer real,
    // BEAGLE_MI355_VSTEPS=n caps the length of a virtual definition (A/B runs).
    in->walk = stateCount == 4 && categoryCount <= 16 &&
               (size_t)categoryCount * patternCount * 32 < ((size_t)1 << 32);     // the kernel addresses a buffer with 32-bit lane offsets
    const bool noVirtual = getenv("BEAGLE_MI355_NO_VIRTUAL") && atoi(getenv("BEAGLE_MI355_NO_VIRTUAL")) != 0;
    // T32 instances with <= 20 states: tip-tip nodes ("cherries") are defined, not stored — their parent's kernel rebuilds
    // them from the tips' states (kernels_mfma.hip cherryOperands); a definition is ONE step here
    // 16..20 states: the pattern walk on the T32 layout (BEAGLE_MI355_NO_T32_WALK=1: the level kernels with virtual cherries)
    // 21..64 states (round 6): the same walk without hold slots (kernels_mfma.hip k_walkT64; BEAGLE_MI355_NO_T64_WALK=1: the level kernels)
    const bool walk64 = in->tiled && stateCount > 20 && categoryCount <= 16 && !(getenv("BEAGLE_MI355_NO_T64_WALK") && atoi(getenv("BEAGLE_MI355_NO_T64_WALK")) != 0);
    in->walkT = in->tiled && categoryCount <= 16 && (stateCount <= 20 || walk64) && !(getenv("BEAGLE_MI355_NO_T32_WALK") && atoi(getenv("BEAGLE_MI355_NO_T32_WALK")) != 0);
    // (above 20 states the cherries' matrices do not fit the LDS; with the tables in global memory — BEAGLE_MI355_CHERRY61=1 — a third
    // of config C's nodes is never stored and the time does not move: 232 against 234 evals/s, profiles/r03_experiments.txt 14 — so
    // that stays an experiment)
    in->cherry = in->tiled && !noVirtual && !in->walkT &&
                 (stateCount <= 20 || (getenv("BEAGLE_MI355_CHERRY61") && atoi(getenv("BEAGLE_MI355_CHERRY61")) != 0));
    const bool virtualOn = ((in->walk || in->walkT) && !noVirtual) || in->cherry;
    in->virt = virtualOn;
    // Size of a virtual definition (internal nodes; any subtree shape whose evaluation needs at most two hold slots).
    // Evaluations at alignment sizes that keep the chip busy are bound by the bytes of the STORED nodes and their time
    // follows the cap (config A, profiles/r02_experiments.txt: cap 8 -> 207 stored nodes; 16 -> 112, 0.70 ms; 24 -> 78,
    // 0.64 ms; 32 -> 62, 0.63 ms) while a branch move — which re-evaluates the virtual siblings it passes instead of reading
    // 32 C P bytes each — costs 139 / 141 / 162 us at 16 / 24 / 32.  Small alignments are latency-bound: there the extra
    // micro-operations of long definitions show (12 500 patterns: branch move 65 -> 71 us from cap 8 to 16).
    // Round 6, the smallest alignments (a partials buffer under 64 KiB: the reference's benchmark1 alignment, 593 patterns): storing a
    // node costs next to nothing there, re-evaluating it costs stages — cap 2: a full evaluation 84.5 -> 78.6 us, a branch move 54 -> 46 us,
    // the mixed chain 12 070 -> 13 800 evaluations/s (tools/r06_vsteps_sweep.sh; at 5 565 patterns the full evaluation already prefers 8).
    const size_t bufferBytes = (size_t)categoryCount * std::max(patternCount, mi355::tlsWholePatternCount) * 32;      // (a shard of a sharded instance: as the whole would, sharded.h)
    int maxVirtSteps = bufferBytes >= ((size_t)2 << 20) ? 24 : bufferBytes < ((size_t)64 << 10) && stateCount == 4 ? 2 : 8;
    if (labEnv("BEAGLE_MI355_VSTEPS")) maxVirtSteps = std::max(1, std::min(mi355::PLAN_MAX_STEPS, atoi(labEnv("BEAGLE_MI355_VSTEPS"))));
    if (in->cherry) maxVirtSteps = 1;
    // (k_walkT64's definitions are ladders — no hold slots —, and a step's two matrix snapshots are 2 x 119 KB at 61 states and four categories)
    if (in->walkT && stateCount > 20) maxVirtSteps = std::min(maxVirtSteps, 8);
    // hold slots: three where the 4-state walk's LDS allows; TWO for the T32 walk (20 KiB each there: 3 workgroups per CU instead of 2)
    in->holdSlots = in->walkT ? (stateCount > 20 ? 0 : 2) : mi355::walkHoldSlots(categoryCount);
    if (labEnv("BEAGLE_MI355_HOLD_SLOTS") && !(in->walkT && stateCount > 20)) in->holdSlots = std::max(1, std::min(atoi(labEnv("BEAGLE_MI355_HOLD_SLOTS")), in->walkT ? 3 : mi355::walkHoldSlots(categoryCount)));
    in->fuseRootParts = !(getenv("BEAGLE_MI355_NO_ROOT_PARTS_FUSION") && atoi(getenv("BEAGLE_MI355_NO_ROOT_PARTS_FUSION")) != 0);
    in->walkTWrite = in->walkT && stateCount <= 20 && categoryCount <= mi355::WALK_T32_WRITE_MAX_CATEGORIES && in->holdSlots <= mi355::WALK_T32_WRITE_MAX_HOLD &&
                     !(getenv("BEAGLE_MI355_NO_T32_WRITE_WALK") && atoi(getenv("BEAGLE_MI355_NO_T32_WRITE_WALK")) != 0);
    in->planner.init(partialsBufferCount, tipCount, matrixBufferCount, scaleBufferCount, maxVirtSteps, virtualOn, in->holdSlots);
    in->planner.cacheEnabled = !(getenv("BEAGLE_MI355_NO_PLAN_CACHE") && atoi(getenv("BEAGLE_MI355_NO_PLAN_CACHE")) != 0);
    in->fastWalk = !(getenv("BEAGLE_MI355_NO_FAST_WALK") && atoi(getenv("BEAGLE_MI355_NO_FAST_WALK")) != 0);
    in->strictWaits = !(getenv("BEAGLE_MI355_STRICT_WAITS") && atoi(getenv("BEAGLE_MI355_STRICT_WAITS")) == 0);
    in->fuseGradient = !(getenv("BEAGLE_MI355_NO_FUSED_GRADIENT") && atoi(getenv("BEAGLE_MI355_NO_FUSED_GRADIENT")) != 0);
    in->preWalk = !(getenv("BEAGLE_MI355_NO_PRE_WALK") && atoi(getenv("BEAGLE_MI355_NO_PRE_WALK")) != 0);
    in->fuseLaunches = !(getenv("BEAGLE_MI355_NO_LAUNCH_FUSION") && atoi(getenv("BEAGLE_MI355_NO_LAUNCH_FUSION")) != 0);
    in->deferWalk = !(getenv("BEAGLE_MI355_NO_ROOT_FUSION") && atoi(getenv("BEAGLE_MI355_NO_ROOT_FUSION")) != 0);
    in->foldScales = !(getenv("BEAGLE_MI355_NO_SCALE_FOLD") && atoi(getenv("BEAGLE_MI355_NO_SCALE_FOLD")) != 0);
    // What a gradient chain's post-order passes leave unstored for the pre-order walk to re-evaluate (BEAGLE_MI355_GRADIENT_VIRTUAL):
    // 0 nothing; 1 (default) nodes over two compact tips — a third of a tree's nodes, evaluated INSIDE their parent's descriptor
    // (kernels.h PW_CHERRY); 2 also such a node under one more tip (descriptors of their own, PW_POSTOP: half the nodes, but a
    // descriptor costs a stage whatever it computes — slower than 1, for whoever needs the memory: profiles/r05_experiments.txt 5, 14)
    {
        const int gv = getenv("BEAGLE_MI355_GRADIENT_VIRTUAL") ? atoi(getenv("BEAGLE_MI355_GRADIENT_VIRTUAL")) : GRADIENT_VIRT_DEFAULT;
        in->gradientVirtual = in->walk && virtualOn && in->preWalk && in->fuseGradient && gv > 0;
        in->gradientVirtualSteps = std::max(1, std::min(GRADIENT_VIRT_STEPS, gv));
    }
    // matrix storage: the caller's buffers, then the private snapshot slots of virtual definitions (planner.h)
    const size_t matrixSlots = matrixSlotLayout(in);
    const size_t patternSlots = in->tiled ? (size_t)in->ntile * 32 : (size_t)patternCount;
    in->partialsBytes = (((size_t)categoryCount * patternSlots * stateCount * sizeof(double)) + 255 + (in->walk ? 256 : 0)) & ~(size_t)255;
    in->partials.assign(partialsBufferCount, nullptr);
    in->scaleOfPartial.assign(partialsBufferCount, -2); in->scaleVersionAtWrite.assign(partialsBufferCount, 0u);     // (-2: unknown)
    in->scaleVersion.assign(std::max(1, scaleBufferCount), 0u);
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
    ok = ok && hipHostMalloc((void**)&in->hRing, RING_BYTES, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void**)&in->hRingDev, in->hRing, 0) == hipSuccess;     // (the copies out of the ring are a kernel's: flushUploads)
    in->kernelUploads = !(getenv("BEAGLE_MI355_COPY_ENGINE_UPLOADS") && atoi(getenv("BEAGLE_MI355_COPY_ENGINE_UPLOADS")) != 0);
    in->sitePrefetch = !(getenv("BEAGLE_MI355_NO_SITE_PREFETCH") && atoi(getenv("BEAGLE_MI355_NO_SITE_PREFETCH")) != 0);
    in->fuseWaves = !(getenv("BEAGLE_MI355_NO_WALK_FUSION") && atoi(getenv("BEAGLE_MI355_NO_WALK_FUSION")) != 0);
    in->useTickets = !(getenv("BEAGLE_MI355_NO_WALK_TICKETS") && atoi(getenv("BEAGLE_MI355_NO_WALK_TICKETS")) != 0);
    in->xcdAware = !(getenv("BEAGLE_MI355_NO_XCD_MAP") && atoi(getenv("BEAGLE_MI355_NO_XCD_MAP")) != 0);
    in->fuseCherries = !(getenv("BEAGLE_MI355_NO_CHERRY_FUSION") && atoi(getenv("BEAGLE_MI355_NO_CHERRY_FUSION")) != 0);
    in->skipTipLoads = !(getenv("BEAGLE_MI355_NO_LOAD_SKIP") && atoi(getenv("BEAGLE_MI355_NO_LOAD_SKIP")) != 0);
    in->sliceSums = !(getenv("BEAGLE_MI355_NO_SLICE_SUMS") && atoi(getenv("BEAGLE_MI355_NO_SLICE_SUMS")) != 0);
    in->hostTrace = getenv("BEAGLE_MI355_HOST_TIMING") && atoi(getenv("BEAGLE_MI355_HOST_TIMING")) > 1;     // (a line per slow updatePartials call)
    if (getenv("BEAGLE_MI355_WALK_SPIN_US")) in->walkSpinLimit = (unsigned long long)std::max(0L, atol(getenv("BEAGLE_MI355_WALK_SPIN_US"))) * 100ull;
    if (in->walk && in->fuseWaves && in->fastWalk) {
        // (on tickets — the default — a slice above the first wave costs no workgroup slots and no polling, and the first wave is the whole
        // grid: 8 above / about twice as long first-wave slices measured best at 12 500 patterns, tools/r06_ticket_sweep.sh)
        in->planner.chunkTopOps = labEnv("BEAGLE_MI355_CHUNK_TOP") ? atoi(labEnv("BEAGLE_MI355_CHUNK_TOP")) : in->useTickets ? 8 : 16;
        // slices the chip holds side by side: 4 workgroups per CU over the pattern groups of a slice (planner.h launchMachines)
        hipDeviceProp_t prop;
        const int cus = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        in->planner.launchMachines = (double)(4 * cus) / (double)std::max(1, (patternCount + 127) / 128);
        if (labEnv("BEAGLE_MI355_SCHED_SIM") && atoi(labEnv("BEAGLE_MI355_SCHED_SIM")) == 0) in->planner.launchMachines = 0.0;
    }
    // result words live in coherent, device-mapped host memory: the final reduction kernel writes the sum straight into it
    // and the host only waits for the stream (no device-to-host copy behind the last kernel)
    ok = ok && hipHostMalloc((void**)&in->hResult, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void**)&in->hResultDev, in->hResult, 0) == hipSuccess;
    const size_t S = stateCount, C = categoryCount, E = in->eigenCount;
    const int rootBlocks = (patternCount + 63) / 64;          // (the T32 root kernel: a partial sum per 64 patterns; 4 states: per 256)
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
    ok = ok && devAlloc(in, (void**)&in->rootCounter, 256) == 0 && hipMemset(in->rootCounter, 0, 256) == hipSuccess;
    if (ok) in->walkSelfServed = in->rootCounter + 32;       // (its own 128-byte line of the same allocation)
    if (ok) {
        // defaults: category rates 1, weights 1/C, pattern weights 1 (beagle.jar!GeneralBeagleImpl#<init>)
        std::vector<double> ones(std::max<size_t>((size_t)patternCount, E * C), 1.0);
        ok = upload(in, in->rates, ones.data(), E * C * sizeof(double)) == 0;
        ok = ok && upload(in, in->patternWeights, ones.data(), (size_t)patternCount * sizeof(double)) == 0;
        std::vector<double> w(E * C, 1.0 / (double)C);
        ok = ok && upload(in, in->weights, w.data(), E * C * sizeof(double)) == 0;
        ok = ok && hipMemsetAsync(in->matrices, 0, matrixSlots * C * S * S * sizeof(double), live(in)) == hipSuccess;
        ok = ok && hipMemsetAsync(in->siteLogL, 0, (size_t)patternCount * sizeof(double), live(in)) == hipSuccess;
        if (ok && in->tiled) ok = uploadIdentityMatrix(in) == 0;   // for the two-pass pre-order path
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
    // (the T32 walk needs no layout change: a tile that straddles two partitions is walked once per partition, each walk storing
    // only its own patterns — kernels_mfma.hip k_walkT32 masks by the segment's range)
    // (every instance whose planner is in use: its tables are keyed by (buffer, partition) — the level kernels' cherry instances
    // too, although they define nothing while there are several partitions)
    if (in->walk || in->walkT || in->virt) in->planner.setPartitionCount(partitionCount);
    if (in->virt) {
        // definitions are kept per (buffer, partition): more snapshot slots behind the caller's matrices
        // (T32 instances keep an identity matrix and the transposed-matrix scratch of the two-pass pre-order path BEHIND the
        // snapshot slots — matrixSlotLayout, as at creation: they move with the block and the identity is sent again)
        const size_t per = (size_t)in->C * in->S * in->S, slots = matrixSlotLayout(in);
        double* grown = nullptr;
        int rcm = devAlloc(in, (void**)&grown, slots * per * sizeof(double)); if (rcm) return rcm;
        HIP_TRY(hipMemsetAsync(grown, 0, slots * per * sizeof(double), live(in)));
        HIP_TRY(hipMemcpyAsync(grown, in->matrices, (size_t)std::max(1, in->matrixCount) * per * sizeof(double), hipMemcpyDeviceToDevice, live(in)));
        in->matrices = grown;                              // (the old block stays owned by the instance until it is destroyed)
        if (in->tiled) { int rci = uploadIdentityMatrix(in); if (rci) return rci; }
    }
    if (in->walk) {
        // the pair-interleaved arrays follow the partitions (Instance::pairPos): what exists already — tips are uploaded before
        // this call, MultiPartitionDataLikelihoodDelegate.java:544-553 — moves to the new layout on the device
        forgetFolds(in);
        setPairLayout(in);
        in->sliceRows = 0; in->lastSums.valid = false;                    // (the per-slice factor products are [row][pairLen]: re-made at the new length on first use)
        if (!in->dPairPos) { int rc = devAlloc(in, (void**)&in->dPairPos, (size_t)in->P * sizeof(unsigned)); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(live(in)));
        HIP_TRY(hipMemcpy(in->dPairPos, in->pairPos.data(), (size_t)in->P * sizeof(unsigned), hipMemcpyHostToDevice));
        in->stateSlabLeft = 0; in->scaleSlabLeft = 0;                      // new slabs: the element sizes changed
        std::vector<mi355::RelayoutJob> jobs;                              // every tip in one pair of launches
        for (int t = 0; t < in->partialsCount; t++) {
            uint8_t* old = in->tipStates[t];
            if (!old) continue;
            in->tipStates[t] = nullptr;
            int rc = ensureStates(in, t); if (rc) return rc;
            jobs.push_back({old, in->tipStates[t], in->tipStates[t] + in->statePairOff});
        }
        for (size_t b = 0; b < jobs.size(); b += 16384) {                  // (the job list goes through the staging ring)
            const size_t e = std::min(jobs.size(), b + 16384);
            void* dJobs = nullptr;
            int rc = uploadTransient(in, jobs.data() + b, (e - b) * sizeof(mi355::RelayoutJob), &dJobs); if (rc) return rc;
            mi355::launchRelayoutStatesBatch(live(in), (const mi355::RelayoutJob*)dJobs, (int)(e - b), in->dPairPos, in->P, (int)in->pairLen, in->S);
        }
        for (int k = 0; k < (int)in->scale.size(); k++) {
            double* old = in->scale[k];
            if (!old) continue;
            const char raw = in->scaleIsRaw[k];
            in->scale[k] = nullptr;
            int rc = ensureScale(in, k); if (rc) return rc;                // (zero-filled)
            in->scaleIsRaw[k] = raw;
            HIP_TRY(hipMemcpyAsync(in->scale[k], old, (size_t)in->P * sizeof(double), hipMemcpyDeviceToDevice, live(in)));
            if (raw) mi355::launchRecipFromFactors(live(in), in->scale[k], in->scale[k] + in->scaleStride, in->dPairPos, in->P);
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
    in->scaleOfPartial[tipIndex] = -1;                     // (caller's data: no scale factor of ours in it)
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
        if (!rc) mi355::launchReplicateCategories(live(in), last, in->partials[tipIndex], in->P, in->S, in->C - 1);
    }
    in->tipStates[tipIndex] = nullptr;   // the buffer now holds partials (slab memory stays owned by the instance)
    setCompact(in, tipIndex, false);
    setLeaf(in, tipIndex);               // ... that no operation computes: definitions may read them (planner.h leafPartials)
    return rc;
}

int beagleSetPartials(int instance, int bufferIndex, const double* inPartials) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedSetPerPatternDoubles(instance, inPartials, shardedStates(instance), shardedCategories(instance), [&](int h, const double* v) { return beagleSetPartials(h, bufferIndex, v); }); }
    GET_INSTANCE_KEEP_PENDING(instance);
    if (badIndex(bufferIndex, in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    // (a held-back pre-order list has its own copy of its root's pre-order partial — the buffer the gradient delegates rewrite
    // before every list — and waits unless this is one of the other buffers it reads or writes)
    if (heldTouches(in, bufferIndex)) { int rcp = executeHeldPre(in); if (rcp) return rcp; }
    in->scaleOfPartial[bufferIndex] = -1;                  // (caller's data: no scale factor of ours in it)
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
        HIP_TRY(hipStreamSynchronize(live(in)));
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
            mi355::launchExportPartials(live(in), in->partials[b], sc, raw, in->exportDev[w] + k * elems, in->P, in->S, in->C, in->tiled);
        }
        HIP_TRY(hipMemcpyAsync(in->exportHost[w], in->exportDev[w], n * bytes, hipMemcpyDeviceToHost, live(in)));
        HIP_TRY(hipEventRecord(in->exportEvent[w], live(in)));
        if (c >= 1 && out) {                               // chunk c - 1 has landed (or lands while this one is being produced)
            HIP_TRY(hipEventSynchronize(in->exportEvent[1 - w]));
            copies[1 - w].start((char*)(out + (c - 1) * chunk * elems), (const char*)in->exportHost[1 - w], chunkCount(c - 1) * bytes);
        }
    }
    const int last = (int)((nChunks - 1) & 1);
    HIP_TRY(hipEventSynchronize(in->exportEvent[last]));
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
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
    if (sitePrefetchTake(in, nullptr)) { *outPinned = in->hSites; *outCount = in->P; return BEAGLE_SUCCESS; }      // (valid until the next root sum)
    if (bytes > RING_BYTES) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    HIP_TRY(hipMemcpyAsync(in->hRing, in->siteLogL, bytes, hipMemcpyDeviceToHost, live(in)));     // (the ring is pinned; everything staged in it
    HIP_TRY(hipStreamSynchronize(live(in)));                                                      //  has been consumed once the stream is idle)
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
    if (mi355::isShardedHandle(instance)) {          // (queued: every shard's thread applies its copy; sharded.h shardedPost)
        const size_t S = (size_t)shardedStates(instance);
        std::vector<double> u(U, U + S * S), ui(Uinv, Uinv + S * S), lam(lambda, lambda + (mi355::shardedEigenComplex(instance) ? 2 : 1) * S);
        return mi355::shardedPost(instance, [=](int h) { return beagleSetEigenDecomposition(h, eigenIndex, u.data(), ui.data(), lam.data()); });
    }
    GET_INSTANCE(instance);
    if (badIndex(eigenIndex, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    const size_t S = in->S, nLambda = in->eigenComplex ? 2 * S : S, stride = 2 * S * S + nLambda;
    if (in->eigenComplex) {
        // imaginary parts come as adjacent conjugate pairs (b, -b) — ComplexSubstitutionModel.java:121-173 walks them that way, and
        // the kernel (kernels.hip iexpEntry) reads the row after a pair's first row: a lone or unmatched entry is refused here
        for (size_t k = 0; k < S; k++) {
            const double im = lambda[S + k];
            if (im == 0.0) continue;
            if (k + 1 >= S || lambda[S + k + 1] != -im) return BEAGLE_ERROR_OUT_OF_RANGE;
            k++;
        }
    }
    std::vector<double> pack(stride);
    memcpy(&pack[0], U, S * S * sizeof(double));
    memcpy(&pack[S * S], Uinv, S * S * sizeof(double));
    memcpy(&pack[2 * S * S], lambda, nLambda * sizeof(double));
    return uploadIfChanged(in, in->shEigen, in->okEigen, in->eigenCount, eigenIndex, stride, in->eigen + stride * eigenIndex, pack.data());
}

int beagleSetStateFrequencies(int instance, int idx, const double* f) {
    if (mi355::isShardedHandle(instance)) { std::vector<double> v(f, f + shardedStates(instance)); return mi355::shardedPost(instance, [=](int h) { return beagleSetStateFrequencies(h, idx, v.data()); }); }
    GET_INSTANCE(instance);
    if (badIndex(idx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    in->copyKeepsWalk = true;                            // (the root's input, not the walk's: a held launch stays held)
    const int rc = uploadIfChanged(in, in->shFreqs, in->okFreqs, in->eigenCount, idx, in->S, in->freqs + (size_t)idx * in->S, f);
    in->copyKeepsWalk = false;
    return rc;
}

int beagleSetCategoryWeights(int instance, int idx, const double* w) {
    if (mi355::isShardedHandle(instance)) { std::vector<double> v(w, w + shardedCategories(instance)); return mi355::shardedPost(instance, [=](int h) { return beagleSetCategoryWeights(h, idx, v.data()); }); }
    GET_INSTANCE(instance);
    if (badIndex(idx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    in->copyKeepsWalk = true;
    const int rc = uploadIfChanged(in, in->shWeights, in->okWeights, in->eigenCount, idx, in->C, in->weights + (size_t)idx * in->C, w);
    in->copyKeepsWalk = false;
    return rc;
}

int beagleSetCategoryRatesWithIndex(int instance, int idx, const double* r) {
    if (mi355::isShardedHandle(instance)) { std::vector<double> v(r, r + shardedCategories(instance)); return mi355::shardedPost(instance, [=](int h) { return beagleSetCategoryRatesWithIndex(h, idx, v.data()); }); }
    GET_INSTANCE(instance);
    if (badIndex(idx, in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return uploadIfChanged(in, in->shRates, in->okRates, in->eigenCount, idx, in->C, in->rates + (size_t)idx * in->C, r);
}

int beagleSetCategoryRates(int instance, const double* r) { return beagleSetCategoryRatesWithIndex(instance, 0, r); }

int beagleSetTransitionMatrix(int instance, int matrixIndex, const double* inMatrix, double paddedValue) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleSetTransitionMatrix(h, matrixIndex, inMatrix, paddedValue); }); }
    (void)paddedValue;
    GET_INSTANCE_KEEP_PENDING(instance);
    if (badIndex(matrixIndex, in->matrixCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    // (setDifferentialMatrix arrives between updatePrePartials and calculateEdgeDifferentials: the held-back list stays held
    // unless this very matrix is one of its branch matrices)
    if (heldReadsMatrix(in, matrixIndex)) { int rcp = executeHeldPre(in); if (rcp) return rcp; }
    const size_t n = (size_t)in->C * in->S * in->S;
    return upload(in, in->matrices + n * matrixIndex, inMatrix, n * sizeof(double));
}

int beagleGetTransitionMatrix(int instance, int matrixIndex, double* outMatrix) {
    if (mi355::isShardedHandle(instance)) { bool first = true; std::mutex mu; return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleGetTransitionMatrix(h, matrixIndex, outMatrix); }); }
    GET_INSTANCE(instance);
    if (badIndex(matrixIndex, in->matrixCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    const size_t n = (size_t)in->C * in->S * in->S;
    // a small matrix (4 states: 64 doubles with four categories) comes back through the polled result page, as a root sum does: callers that
    // read one per branch (ancestral states, Markov jumps: AncestralStateBeagleTreeLikelihood.java:414-542) paid a device-to-host copy and a
    // stream synchronisation each — 27 us a matrix
    if (outMatrix && n <= 480) return mi355::publishAndWait(instance, in->matrices + n * matrixIndex, (int)n, outMatrix);
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
        mi355::launchConvolveMatrices(live(in), in->matrices, (const int*)dF, (const int*)dS, (const int*)dR, n, in->S, in->C);
        b = e;
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

static int transitionMatrices(Instance* in, const int* eigenIdx, int eigenScalar, const int* rateIdx,
                              const int* probIdx, const double* lens, int count) {
    if (count <= 0) return BEAGLE_SUCCESS;
    // range checks as minimum / maximum scans (they vectorise; a partitioned evaluation names 12 900 branches, and a loop with an early
    // return per branch cost more than the kernel it fed)
    auto outOfRange = [count](const int* v, int limit) {
        int lo = v[0], hi = v[0];
        for (int k = 1; k < count; k++) { lo = v[k] < lo ? v[k] : lo; hi = v[k] > hi ? v[k] : hi; }
        return lo < 0 || hi >= limit;
    };
    if (outOfRange(probIdx, in->matrixCount) || (eigenIdx ? outOfRange(eigenIdx, in->eigenCount) : badIndex(eigenScalar, in->eigenCount)) ||
        (rateIdx && outOfRange(rateIdx, in->eigenCount))) return BEAGLE_ERROR_OUT_OF_RANGE;
    // 4 states, one eigen system and one rate set (beagleUpdateTransitionMatrices — what every evaluation of a chain issues): the
    // branch lengths and matrix indices stay in the staging ring, which the device maps, and the kernel reads them — and an
    // eigen system / rate set whose upload is still queued — from there; the queued copies ride in the same launch
    // (kernels.hip k_transition4Fused).  One launch instead of a copy kernel and a transition kernel: 5 us of a 12 500-pattern
    // evaluation's 170 (profiles/r04_experiments.txt).
    // The queued copies would run side by side with the transition blocks and with each other, in no order: a queued copy INTO the
    // matrix block (beagleSetTransitionMatrix of a slot this call may rewrite: the later call has to win) or two queued copies whose
    // ranges overlap without being the same array are flushed first, in order, and the plain kernel takes this call.
    bool fusable = in->S == 4 && in->kernelUploads && in->fuseLaunches && !eigenIdx && !rateIdx && (size_t)count * 12 <= RING_BYTES / 4 &&
                   (int)in->pendingCopies.size() <= mi355::HOST_COPY_MAX;
    if (fusable) {
        const char* m0 = (const char*)in->matrices;
        const char* m1 = m0 + (size_t)std::max(1, in->matrixCount) * in->C * in->S * in->S * sizeof(double);
        const std::vector<Instance::PendingCopy>& pc = in->pendingCopies;
        for (size_t a = 0; a < pc.size() && fusable; a++) {
            const char* d0 = (const char*)pc[a].dst; const char* d1 = d0 + pc[a].bytes;
            if (d0 < m1 && m0 < d1) fusable = false;
            for (size_t b = a + 1; b < pc.size() && fusable; b++) {
                const char* e0 = (const char*)pc[b].dst; const char* e1 = e0 + pc[b].bytes;
                if (d0 < e1 && e0 < d1 && !(e0 <= d0 && d1 <= e1)) fusable = false;      // (an earlier copy fully covered by a later one is simply dropped below)
            }
        }
    }
    if (fusable) {
        const size_t lenBytes = (size_t)count * sizeof(double), idxBytes = (size_t)count * sizeof(int);
        const long off = stage(in, lens, lenBytes, lenBytes + idxBytes);
        if (off < 0) return BEAGLE_ERROR_GENERAL;
        memcpy(in->hRing + off + lenBytes, probIdx, idxBytes);
        const size_t eigStride = in->eigenComplex ? 40 : 36;
        const double* eigSrc = in->eigen + eigStride * eigenScalar;
        const double* ratesSrc = in->rates;                                   // (rate set 0)
        mi355::HostCopyList L;
        L.n = 0;
        unsigned blocks = 0;
        for (const Instance::PendingCopy& pc : in->pendingCopies) {
            mi355::HostCopyList::Entry& e = L.e[L.n++];
            e.dst = pc.dst; e.src = in->hRingDev + pc.ringOff; e.bytes = (unsigned)pc.bytes; e.firstBlock = blocks;
            blocks += (unsigned)((pc.bytes + 4095) / 4096);
            // (the LAST queued upload of an array is the one that counts)
            if (pc.dst == (void*)eigSrc && pc.bytes == eigStride * sizeof(double)) eigSrc = (const double*)(in->hRingDev + pc.ringOff);
            if (pc.dst == (void*)in->rates && pc.bytes >= (size_t)in->C * sizeof(double)) ratesSrc = (const double*)(in->hRingDev + pc.ringOff);
        }
        // (an array queued twice: the copies run side by side — an earlier one that a later one covers entirely is dropped; any
        // other overlap was excluded above)
        for (int a = 0; a < L.n; a++)
            for (int b = a + 1; b < L.n; b++)
                if ((const char*)L.e[b].dst <= (const char*)L.e[a].dst &&
                    (const char*)L.e[a].dst + L.e[a].bytes <= (const char*)L.e[b].dst + L.e[b].bytes) L.e[a].bytes = 0;
        in->pendingCopies.clear();
        mi355::launchTransitionMatrices4Fused(in->stream, in->matrices, eigSrc, ratesSrc, (const int*)(in->hRingDev + off + lenBytes),
                                              (const double*)(in->hRingDev + off), count, in->C, in->eigenComplex, L, (int)blocks);
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    // 4 states, several eigen systems or rate sets (beagleUpdateTransitionMatricesWithMultipleModels — what every evaluation of a
    // partitioned chain issues): lengths and the three index lists stay in the staging ring as above and k_transition4 reads them from
    // there; queued uploads (an eigen system, a rate set) go first, in their own launch — a chain's steady state has none.  One launch
    // instead of a copy kernel (the ring's 20 bytes per branch over PCIe: 9 us for config E's 12 900 branches) and a transition
    // kernel (5 us) behind it: the reads now overlap the arithmetic.
    if (in->S == 4 && in->kernelUploads && in->fuseLaunches && (size_t)count * 20 <= RING_BYTES / 4) {
        const size_t lenBytes = (size_t)count * sizeof(double), idxBytes = (size_t)count * sizeof(int);
        const long off = stage(in, lens, lenBytes, lenBytes + 3 * idxBytes);
        if (off < 0) return BEAGLE_ERROR_GENERAL;
        memcpy(in->hRing + off + lenBytes, probIdx, idxBytes);
        int* rEig = (int*)(in->hRing + off + lenBytes + idxBytes);
        int* rRate = rEig + count;
        if (eigenIdx) memcpy(rEig, eigenIdx, idxBytes); else std::fill(rEig, rEig + count, eigenScalar);
        if (rateIdx) memcpy(rRate, rateIdx, idxBytes); else std::fill(rRate, rRate + count, 0);
        const int* rIdx = (const int*)(in->hRingDev + off + lenBytes);
        mi355::launchTransitionMatrices(live(in), in->matrices, in->eigen, in->rates, rIdx, (const double*)(in->hRingDev + off),
                                        rIdx + count, rIdx + 2 * (size_t)count, count, in->S, in->C, in->eigenComplex);
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    // one packed upload: [lengths double[count] | matrix idx | eigen idx | rate idx] (each copy is a blit kernel)
    std::vector<char> pack((size_t)count * (sizeof(double) + 3 * sizeof(int)));
    double* pLen = (double*)pack.data();
    int* pIdx = (int*)(pack.data() + (size_t)count * sizeof(double));
    memcpy(pLen, lens, (size_t)count * sizeof(double));
    memcpy(pIdx, probIdx, (size_t)count * sizeof(int));
    if (eigenIdx) memcpy(pIdx + count, eigenIdx, (size_t)count * sizeof(int)); else std::fill(pIdx + count, pIdx + 2 * (size_t)count, eigenScalar);
    if (rateIdx) memcpy(pIdx + 2 * (size_t)count, rateIdx, (size_t)count * sizeof(int)); else std::fill(pIdx + 2 * (size_t)count, pIdx + 3 * (size_t)count, 0);
    void* dPack;
    int rc = uploadTransient(in, pack.data(), pack.size(), &dPack); if (rc) return rc;
    const double* dLen = (const double*)dPack;
    const int* dIdx = (const int*)((const char*)dPack + (size_t)count * sizeof(double));
    mi355::launchTransitionMatrices(live(in), in->matrices, in->eigen, in->rates, dIdx, dLen,
                                    dIdx + count, dIdx + 2 * (size_t)count, count, in->S, in->C, in->eigenComplex);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices,
                                   const int* firstDerivativeIndices, const int* secondDerivativeIndices,
                                   const double* edgeLengths, int count) {
    if (mi355::isShardedHandle(instance)) {
        if (firstDerivativeIndices || secondDerivativeIndices) return BEAGLE_ERROR_NO_IMPLEMENTATION;
        if (count <= 0) return BEAGLE_SUCCESS;
        std::vector<int> p(probabilityIndices, probabilityIndices + count); std::vector<double> t(edgeLengths, edgeLengths + count);
        return mi355::shardedPost(instance, [=](int h) { return beagleUpdateTransitionMatrices(h, eigenIndex, p.data(), nullptr, nullptr, t.data(), count); });
    }
    GET_INSTANCE_KEEP_PENDING(instance);                      // (a held-back pre-order list waits unless one of ITS matrices is rewritten)
    if (firstDerivativeIndices || secondDerivativeIndices) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    if (in->heldPre.held && probabilityIndices)
        for (int k = 0; k < count; k++) if (heldReadsMatrix(in, probabilityIndices[k])) { int rcp = executeHeldPre(in); if (rcp) return rcp; break; }
    return transitionMatrices(in, nullptr, eigenIndex, nullptr, probabilityIndices, edgeLengths, count);
}

int beagleUpdateTransitionMatricesWithMultipleModels(int instance, const int* eigenIndices, const int* categoryRateIndices,
                                   const int* probabilityIndices, const int* firstDerivativeIndices,
                                   const int* secondDerivativeIndices, const double* edgeLengths, int count) {
    if (mi355::isShardedHandle(instance)) {
        if (firstDerivativeIndices || secondDerivativeIndices) return BEAGLE_ERROR_NO_IMPLEMENTATION;
        if (!eigenIndices || !categoryRateIndices) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (count <= 0) return BEAGLE_SUCCESS;
        std::vector<int> e(eigenIndices, eigenIndices + count), r(categoryRateIndices, categoryRateIndices + count), p(probabilityIndices, probabilityIndices + count);
        std::vector<double> t(edgeLengths, edgeLengths + count);
        return mi355::shardedPost(instance, [=](int h) { return beagleUpdateTransitionMatricesWithMultipleModels(h, e.data(), r.data(), p.data(), nullptr, nullptr, t.data(), count); });
    }
    GET_INSTANCE(instance);
    if (firstDerivativeIndices || secondDerivativeIndices) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    if (!eigenIndices || !categoryRateIndices) return BEAGLE_ERROR_OUT_OF_RANGE;
    return transitionMatrices(in, eigenIndices, 0, categoryRateIndices, probabilityIndices, edgeLengths, count);
}

int beagleUpdatePartials(int instance, const int* operations, int operationCount, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) {
        if (operationCount <= 0) return BEAGLE_SUCCESS;
        std::vector<int> ops(operations, operations + (size_t)operationCount * BEAGLE_OP_COUNT);
        return mi355::shardedPost(instance, [=](int h) { return beagleUpdatePartials(h, ops.data(), operationCount, cumulativeScaleIndex); });
    }
    GET_INSTANCE_KEEP_PENDING(instance);
    if (operations && (in->heldPre.held || in->trackScales))               // (only on instances that have been asked for a pre-order pass)
        for (int k = 0; k < operationCount; k++) {
            const int* op = operations + (size_t)k * BEAGLE_OP_COUNT;
            // (a held-back pre-order list waits unless this list overwrites what it reads or touches what it writes)
            if (heldTouches(in, op[0]) || heldWrites(in, op[3]) || heldWrites(in, op[5])) { int rcp = executeHeldPre(in); if (rcp) return rcp; }
            // which scale factor the destination is divided by (the pre-order walk needs it: Instance::scaleOfPartial)
            if (badIndex(op[0], in->partialsCount)) continue;                  // (reported by runOperations)
            const int sIdx = op[1] != BEAGLE_OP_NONE ? op[1] : op[2];
            if (op[1] != BEAGLE_OP_NONE && !badIndex(op[1], in->scaleCount)) in->scaleVersion[op[1]]++;
            in->scaleOfPartial[op[0]] = (sIdx != BEAGLE_OP_NONE && !badIndex(sIdx, in->scaleCount)) ? sIdx : -1;
            in->scaleVersionAtWrite[op[0]] = in->scaleOfPartial[op[0]] >= 0 ? in->scaleVersion[sIdx] : 0u;
        }
    return runOperations(in, operations, operationCount, BEAGLE_OP_COUNT, cumulativeScaleIndex);
}

int beagleUpdatePartialsByPartition(int instance, const int* operations, int operationCount) {
    if (mi355::isShardedHandle(instance)) {
        if (operationCount <= 0) return BEAGLE_SUCCESS;
        std::vector<int> ops(operations, operations + (size_t)operationCount * BEAGLE_PARTITION_OP_COUNT);
        return mi355::shardedPost(instance, [=](int h) { return beagleUpdatePartialsByPartition(h, ops.data(), operationCount); });
    }
    GET_INSTANCE(instance);
    if (operations && in->trackScales)                         // (Instance::scaleOfPartial, as beagleUpdatePartials keeps it)
        for (int k = 0; k < operationCount; k++) {
            const int* op = operations + (size_t)k * BEAGLE_PARTITION_OP_COUNT;
            if (badIndex(op[0], in->partialsCount)) continue;
            const int sIdx = op[1] != BEAGLE_OP_NONE ? op[1] : op[2];
            if (op[1] != BEAGLE_OP_NONE && !badIndex(op[1], in->scaleCount)) in->scaleVersion[op[1]]++;
            // (with several partitions a buffer's pattern ranges may carry different factors: unknown to the walk, which is
            // single-partition anyway)
            in->scaleOfPartial[op[0]] = in->partitionCount > 1 ? -2 : (sIdx != BEAGLE_OP_NONE && !badIndex(sIdx, in->scaleCount)) ? sIdx : -1;
            in->scaleVersionAtWrite[op[0]] = in->scaleOfPartial[op[0]] >= 0 ? in->scaleVersion[sIdx] : 0u;
        }
    return runOperations(in, operations, operationCount, BEAGLE_PARTITION_OP_COUNT, BEAGLE_OP_NONE);
}

int beagleWaitForPartials(int instance, const int* destinationPartials, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleWaitForPartials(h, destinationPartials, count); }); }
    (void)destinationPartials; (void)count;
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
    return BEAGLE_SUCCESS;
}

int beagleAccumulateScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { std::vector<int> v(scaleIndices, scaleIndices + std::max(0, count)); return mi355::shardedPost(instance, [=](int h) { return beagleAccumulateScaleFactors(h, v.data(), count, cumulativeScaleIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);          // (scale buffers only: nothing a held-back pre-order list reads or writes)
    if (!badIndex(cumulativeScaleIndex, in->scaleCount)) in->scaleVersion[cumulativeScaleIndex]++;
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, 1.0, 0);
}
int beagleAccumulateScaleFactorsByPartition(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex, int partitionIndex) {
    if (mi355::isShardedHandle(instance)) { std::vector<int> v(scaleIndices, scaleIndices + std::max(0, count)); return mi355::shardedPost(instance, [=](int h) { return beagleAccumulateScaleFactorsByPartition(h, v.data(), count, cumulativeScaleIndex, partitionIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);          // (scale buffers only: nothing a held-back pre-order list reads or writes)
    if (!badIndex(cumulativeScaleIndex, in->scaleCount)) in->scaleVersion[cumulativeScaleIndex]++;
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, 1.0, partitionIndex);
}
int beagleRemoveScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { std::vector<int> v(scaleIndices, scaleIndices + std::max(0, count)); return mi355::shardedPost(instance, [=](int h) { return beagleRemoveScaleFactors(h, v.data(), count, cumulativeScaleIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);          // (scale buffers only: nothing a held-back pre-order list reads or writes)
    if (!badIndex(cumulativeScaleIndex, in->scaleCount)) in->scaleVersion[cumulativeScaleIndex]++;
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, -1.0, 0);
}
int beagleRemoveScaleFactorsByPartition(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex, int partitionIndex) {
    if (mi355::isShardedHandle(instance)) { std::vector<int> v(scaleIndices, scaleIndices + std::max(0, count)); return mi355::shardedPost(instance, [=](int h) { return beagleRemoveScaleFactorsByPartition(h, v.data(), count, cumulativeScaleIndex, partitionIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);          // (scale buffers only: nothing a held-back pre-order list reads or writes)
    if (!badIndex(cumulativeScaleIndex, in->scaleCount)) in->scaleVersion[cumulativeScaleIndex]++;
    return accumulate(in, scaleIndices, count, cumulativeScaleIndex, -1.0, partitionIndex);
}

int beagleResetScaleFactorsByPartition(int instance, int cumulativeScaleIndex, int partitionIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedPost(instance, [=](int h) { return beagleResetScaleFactorsByPartition(h, cumulativeScaleIndex, partitionIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);          // (scale buffers only: nothing a held-back pre-order list reads or writes)
    if (!badIndex(cumulativeScaleIndex, in->scaleCount)) in->scaleVersion[cumulativeScaleIndex]++;
    if (badIndex(cumulativeScaleIndex, in->scaleCount) || badIndex(partitionIndex, in->partitionCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, cumulativeScaleIndex); if (rc) return rc;
    rc = ensureScale(in, cumulativeScaleIndex); if (rc) return rc;
    if (in->scaleIsRaw[cumulativeScaleIndex] && in->partitionCount > 1) {
        // a per-node (raw) buffer is being recycled as a cumulative one: clear all of it first
        mi355::launchFill(live(in), in->scale[cumulativeScaleIndex], 0.0, 0, in->P);
    }
    if (in->scaleIsRaw[cumulativeScaleIndex]) { in->resolveEpoch++; scalesWritten(in); }    // kept programs were validated against the raw flags
    in->scaleIsRaw[cumulativeScaleIndex] = 0;
    mi355::launchFill(live(in), in->scale[cumulativeScaleIndex], 0.0, in->partStart[partitionIndex], in->partEnd[partitionIndex]);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}
int beagleResetScaleFactors(int instance, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedPost(instance, [=](int h) { return beagleResetScaleFactors(h, cumulativeScaleIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);          // (scale buffers only: nothing a held-back pre-order list reads or writes)
    if (!badIndex(cumulativeScaleIndex, in->scaleCount)) in->scaleVersion[cumulativeScaleIndex]++;
    if (badIndex(cumulativeScaleIndex, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, cumulativeScaleIndex); if (rc) return rc;
    rc = ensureScale(in, cumulativeScaleIndex); if (rc) return rc;
    if (in->scaleIsRaw[cumulativeScaleIndex]) { in->resolveEpoch++; scalesWritten(in); }
    in->scaleIsRaw[cumulativeScaleIndex] = 0;
    mi355::launchFill(live(in), in->scale[cumulativeScaleIndex], 0.0, 0, in->P);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleCopyScaleFactors(int instance, int dest, int src) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedPost(instance, [=](int h) { return beagleCopyScaleFactors(h, dest, src); }); }
    GET_INSTANCE_KEEP_PENDING(instance);
    if (!badIndex(dest, in->scaleCount)) in->scaleVersion[dest]++;
    if (badIndex(dest, in->scaleCount) || badIndex(src, in->scaleCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int rc = materializeScaleUsers(in, dest); if (rc) return rc;
    rc = ensureScale(in, dest); if (rc) return rc;
    rc = ensureScale(in, src); if (rc) return rc;
    const size_t scaleDoubles = in->walk ? 2 * in->scaleStride : (size_t)in->P;      // walk instances: factors and reciprocals
    HIP_TRY(hipMemcpyAsync(in->scale[dest], in->scale[src], scaleDoubles * sizeof(double), hipMemcpyDeviceToDevice, live(in)));
    if (in->scaleIsRaw[dest] != in->scaleIsRaw[src]) in->resolveEpoch++;
    if (in->scaleIsRaw[dest] || in->scaleIsRaw[src]) scalesWritten(in);
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
    GET_INSTANCE_KEEP_PENDING(instance);                      // (reads a post-order buffer: a held-back pre-order list writes none)
    if (count != 1) return BEAGLE_ERROR_NO_IMPLEMENTATION;   // BEAST always passes 1 (BeagleTreeLikelihood.java:1038)
    if (bufferIndices && heldWrites(in, bufferIndices[0])) { int rcp = executeHeldPre(in); if (rcp) return rcp; }
    // the reduction kernel writes the sum and then a sequence number into mapped host memory; the kernel is the last
    // thing in the (in-order) stream, so seeing the number means everything before it has completed
    const unsigned long long seq = ++in->resultSeq;
    int rc = rootEnqueue(in, bufferIndices[0], categoryWeightsIndices[0], stateFrequenciesIndices[0],
                         cumulativeScaleIndices[0], -1, in->hResultDev, (unsigned long long*)(in->hResultDev + 8), seq);
    if (rc) return rc;
    rc = sitePrefetchAfterRoot(in); if (rc) return rc;
    { const int rcw = waitResult(in, seq); if (rcw) return rcw; }
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;   // everything staged so far has been consumed
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
    in->sitePrefetched = false; in->siteReadStreak = 0;          // (per-partition sums rewrite siteLogL piece by piece: always the stream-ordered download)
    if (!in->tiled && partitionCount <= 480) {
        // 4-state walk instances, up to eight partitions: 128-pattern groups with the assembly loop's lane map (k_rootSite4WParts) — and
        // when the walk that computes these roots is still held back (engine_walk.cpp runPlan) and its last slices are exactly the
        // named roots, the slices' own epilogues do it: no root launch at all
        const bool groups128 = in->walk && in->fuseLaunches && partitionCount <= mi355::ROOT_MAX_PARTS;
        if (groups128 && in->pendingWalk.valid && in->fuseRootParts) {
            const Instance::PendingWalk& pw = in->pendingWalk;
            mi355::RootFusedParts rp;
            memset(&rp, 0, sizeof(rp));
            bool ok = (int)pw.sinkRows.size() == partitionCount && in->partitionCount > 1;
            int off = 0;
            for (int k = 0; k < partitionCount && ok; k++) {
                const int rootIdx = bufferIndices[k], wIdx = categoryWeightsIndices[k], fIdx = stateFrequenciesIndices[k], cumIdx = cumulativeScaleIndices[k], part = partitionIndices[k];
                if (badIndex(rootIdx, in->partialsCount) || badIndex(part, in->partitionCount) || badIndex(wIdx, in->eigenCount) || badIndex(fIdx, in->eigenCount) ||
                    (cumIdx != BEAGLE_OP_NONE && badIndex(cumIdx, in->scaleCount))) { ok = false; break; }
                int seg = -1;
                for (int row : pw.sinkRows) if (pw.finalStore[(size_t)row] == rootIdx && pw.finalPart[(size_t)row] == part) seg = row;
                for (int j = 0; j < k; j++) if (rp.p[j].rootSeg == seg) seg = -1;                       // (a root named twice: the plain path)
                if (seg < 0) { ok = false; break; }
                mi355::RootFusedPart& q = rp.p[k];
                q.catWeights = in->weights + (size_t)wIdx * in->C; q.freqs = in->freqs + (size_t)fIdx * in->S; q.cum = nullptr; q.cumIsRaw = 0;
                if (cumIdx != BEAGLE_OP_NONE) { int rc = ensureScale(in, cumIdx); if (rc) return rc; q.cum = in->scale[cumIdx]; q.cumIsRaw = in->scaleIsRaw[cumIdx]; }
                q.rootSeg = seg; q.blockOff = off; q.groups = (std::max(0, in->partEnd[part] - in->partStart[part]) + 127) / 128;
                off += q.groups;
            }
            if (ok && off > 0) {
                rp.n = partitionCount; rp.totalGroups = off;
                const unsigned long long seq = ++in->resultSeq;
                mi355::RootFused rf;
                memset(&rf, 0, sizeof(rf));
                rf.rootSeg = -1;
                rf.patternWeights = in->patternWeights; rf.siteLogL = in->siteLogL; rf.blockSums = in->blockSums; rf.counter = in->rootCounter;
                rf.out = in->hResultDev + 16; rf.flag = (unsigned long long*)(in->hResultDev + 8); rf.seq = seq;
                if (!in->rootPartsDev) { int rca = devAlloc(in, (void**)&in->rootPartsDev, Instance::ROOT_PARTS_TABLES * sizeof(rp)); if (rca) return rca; }
                int table = -1;
                for (int t = 0; t < Instance::ROOT_PARTS_TABLES; t++)
                    if (in->rootPartsShadow[t].size() == sizeof(rp) && memcmp(in->rootPartsShadow[t].data(), &rp, sizeof(rp)) == 0) table = t;
                if (table < 0) {
                    table = in->rootPartsNext; in->rootPartsNext = (table + 1) % Instance::ROOT_PARTS_TABLES;
                    in->copyKeepsWalk = true;                        // (the held walk's own input)
                    const int rcu = upload(in, in->rootPartsDev + table, &rp, sizeof(rp));
                    in->copyKeepsWalk = false;
                    if (rcu) return rcu;
                    in->rootPartsShadow[table].assign((const char*)&rp, (const char*)&rp + sizeof(rp));
                }
                rf.parts = in->rootPartsDev + table;
                in->statRootPartsFused++;
                { const int rcw = flushWalk(in, &rf); if (rcw) return rcw; }
                HIP_TRY(hipGetLastError());
                { const int rcw = waitResult(in, seq); if (rcw) return rcw; }
                if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
                double tot = 0.0;
                for (int k = 0; k < partitionCount; k++) { outByPartition[k] = in->hResult[16 + k]; tot += in->hResult[16 + k]; }
                *outSum = tot;
                return (tot != tot) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS;
            }
        }
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
            blockOff += (std::max(0, q.pEnd - q.pStart) + (groups128 ? 127 : 255)) / (groups128 ? 128 : 256);
        }
        const unsigned long long seq = ++in->resultSeq;
        for (size_t c = 0; c < chunks.size(); c++) {
            const bool last = c + 1 == chunks.size();
            if (groups128)
                mi355::launchRootLogLikelihoodParts4W(live(in), chunks[c], in->patternWeights, in->siteLogL, in->blockSums, in->hResultDev + 16, in->P, in->C,
                                                      (unsigned long long*)(in->hResultDev + 8), seq, in->rootCounter);
            else
            mi355::launchRootLogLikelihoodParts(live(in), chunks[c], in->patternWeights, in->siteLogL, in->blockSums,
                                                in->hResultDev + 16 + c * mi355::ROOT_MAX_PARTS, in->P, in->S, in->C,
                                                last ? (unsigned long long*)(in->hResultDev + 8) : nullptr, seq, in->fuseLaunches ? in->rootCounter : nullptr);
        }
        HIP_TRY(hipGetLastError());
        volatile unsigned long long* flag = (volatile unsigned long long*)(in->hResult + 8);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (*flag != seq) HIP_TRY(hipStreamSynchronize(live(in)));
        if (*flag != seq) return BEAGLE_ERROR_GENERAL;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
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
    if (!out) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (sitePrefetchTake(in, out)) return BEAGLE_SUCCESS;         // (already on the host: Instance::hSites)
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
        mi355::launchFillFrequencies(live(in), in->partials[b], in->freqs + (size_t)f * in->S, in->P, in->S, in->C, in->tiled);
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleSetDifferentialMatrix(int instance, int matrixIndex, const double* inMatrix) {
    return beagleSetTransitionMatrix(instance, matrixIndex, inMatrix, 0.0);
}

// result[k] = first[k] + second[k], entry by entry and category by category (declared by BeagleJNIWrapper next to
// convolveTransitionMatrices; no caller in BEAST today).  A result may feed a later triple of the same call: dependent triples in order.
int beagleAddTransitionMatrices(int instance, const int* first, const int* second, const int* result, int count) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleAddTransitionMatrices(h, first, second, result, count); }); }
    GET_INSTANCE(instance);
    if (count <= 0) return BEAGLE_SUCCESS;
    if (!first || !second || !result) return BEAGLE_ERROR_OUT_OF_RANGE;
    for (int k = 0; k < count; k++)
        if (badIndex(first[k], in->matrixCount) || badIndex(second[k], in->matrixCount) || badIndex(result[k], in->matrixCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    int b = 0;
    while (b < count) {
        int e = b + 1;
        for (; e < count; e++) {
            bool dep = false;
            for (int k = b; k < e && !dep; k++)
                dep = result[k] == first[e] || result[k] == second[e] || result[k] == result[e] || first[k] == result[e] || second[k] == result[e];
            if (dep) break;
        }
        const int n = e - b;
        void *dF, *dS, *dR;
        int rc = uploadTransient(in, first + b, n * sizeof(int), &dF); if (rc) return rc;
        rc = uploadTransient(in, second + b, n * sizeof(int), &dS); if (rc) return rc;
        rc = uploadTransient(in, result + b, n * sizeof(int), &dR); if (rc) return rc;
        mi355::launchAddMatrices(live(in), in->matrices, (const int*)dF, (const int*)dS, (const int*)dR, n, in->S, in->C);
        b = e;
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

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
    mi355::launchTransposeMatrices(live(in), in->matrices, (const int*)dPairs, matrixCount, in->S, in->C);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int beagleUpdatePrePartials(int instance, const int* operations, int operationCount, int cumulativeScaleIndex) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdatePrePartials(h, operations, operationCount, cumulativeScaleIndex); }); }
    GET_INSTANCE_KEEP_PENDING(instance);                      // (runPreOperations decides what becomes of a list still held back)
    return runPreOperations(in, operations, operationCount, cumulativeScaleIndex, true);
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
                                                           outDerivatives ? per[k].data() : nullptr, out, outSumSquaredDerivatives ? out + count : nullptr);   // (NULL lets a shard answer without writing pre-order partials)
            if (!r && outDerivatives)
                for (int e = 0; e < count; e++) memcpy(outDerivatives + (size_t)e * P + a, &per[k][(size_t)e * (b - a)], (size_t)(b - a) * sizeof(double));
            return r; }, tot.data());
        if (rc) return rc;
        for (int e = 0; e < count; e++) { if (outSumDerivatives) outSumDerivatives[e] = tot[e]; if (outSumSquaredDerivatives) outSumSquaredDerivatives[e] = tot[count + e]; }
        return BEAGLE_SUCCESS;
    }
    GET_INSTANCE_KEEP_PENDING(instance);                      // (a held-back pre-order list is what this call wants to run with)
    if (!postBufferIndices || !preBufferIndices || !derivativeMatrixIndices || !categoryWeightsIndices) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (badIndex(categoryWeightsIndices[0], in->eigenCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
    return edgeDifferentials(in, postBufferIndices, preBufferIndices, derivativeMatrixIndices, categoryWeightsIndices[0], count,
                             outDerivatives, outSumDerivatives, outSumSquaredDerivatives);
}

// 9-int tuples {pre(child), writeScale, readScale, pre(parent), matrix(child), post(sibling), matrix(sibling), partition,
// cumulativeScale}: beagleUpdatePrePartials over one partition's patterns (declared by BeagleJNIWrapper; no caller in BEAST today)
int beagleUpdatePrePartialsByPartition(int instance, const int* operations, int operationCount) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleUpdatePrePartialsByPartition(h, operations, operationCount); }); }
    GET_INSTANCE(instance);
    return runPreOperations(in, operations, operationCount, BEAGLE_OP_NONE, false, BEAGLE_PARTITION_OP_COUNT);
}

// ---- MI355X extensions -----------------------------------------------------------------------
int beagleMi355SetStream(int instance, void* hipStream) {
    if (mi355::isShardedHandle(instance)) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
    in->stream = hipStream ? (hipStream_t)hipStream : in->ownStream;
    return BEAGLE_SUCCESS;
}

int beagleMi355CalculateRootLogLikelihoodsDevice(int instance, int bufferIndex, int categoryWeightsIndex,
                                                 int stateFrequenciesIndex, int cumulativeScaleIndex, void* deviceOut) {
    if (mi355::isShardedHandle(instance)) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }
    GET_INSTANCE_KEEP_PENDING(instance);
    if (!deviceOut) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (heldWrites(in, bufferIndex)) { int rcp = executeHeldPre(in); if (rcp) return rcp; }
    const int rc = rootEnqueue(in, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex, cumulativeScaleIndex, -1, (double*)deviceOut);
    return rc ? rc : sitePrefetchAfterRoot(in);
}

// ---- one process per GPU: the collective inside the engine -------------------------------------------------------------
int beagleMi355GetCommUniqueId(void* out128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!out128) return BEAGLE_ERROR_OUT_OF_RANGE;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return BEAGLE_ERROR_GENERAL;
    memcpy(out128, &id, sizeof(id));
    return BEAGLE_SUCCESS;
}

int beagleMi355CommInit(int instance, const void* uniqueId128, int rank, int rankCount) {
    if (mi355::isShardedHandle(instance)) return BEAGLE_ERROR_NO_IMPLEMENTATION;      // (resource G+1 owns its own communicator)
    GET_INSTANCE(instance);
    if (!uniqueId128 || rankCount < 1 || rank < 0 || rank >= rankCount) return BEAGLE_ERROR_OUT_OF_RANGE;
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->comm) { ncclCommDestroy(in->comm); in->comm = nullptr; in->commRanks = 0; }
    ncclUniqueId id;
    memcpy(&id, uniqueId128, sizeof(id));
    if (ncclCommInitRank(&in->comm, rankCount, id, rank) != ncclSuccess) { in->comm = nullptr; return BEAGLE_ERROR_GENERAL; }
    in->commRanks = rankCount;
    return BEAGLE_SUCCESS;
}

int beagleMi355CommInfo(int instance, int* outRanks) {
    if (!outRanks) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (mi355::isShardedHandle(instance)) { *outRanks = mi355::shardedCommRanks(instance); return BEAGLE_SUCCESS; }
    Instance* in = lookup(instance);
    if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    int n = 0;
    if (in->comm && ncclCommCount(in->comm, &n) != ncclSuccess) return BEAGLE_ERROR_GENERAL;     // (what RCCL says, not what the caller asked for)
    *outRanks = n;
    return BEAGLE_SUCCESS;
}

int beagleMi355CalculateRootLogLikelihoodsAllReduce(int instance, int bufferIndex, int categoryWeightsIndex, int stateFrequenciesIndex,
                                                    int cumulativeScaleIndex, double* outGlobalSum) {
    if (mi355::isShardedHandle(instance)) return BEAGLE_ERROR_NO_IMPLEMENTATION;
    GET_INSTANCE_KEEP_PENDING(instance);
    if (!outGlobalSum) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (!in->comm) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    if (heldWrites(in, bufferIndex)) { int rcp = executeHeldPre(in); if (rcp) return rcp; }
    int rc = rootEnqueue(in, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex, cumulativeScaleIndex, -1, in->dResult);
    if (rc) return rc;
    // this shard's sum -> the sum over all ranks (RCCL over xGMI; a communicator of one rank still takes the call) -> the host's
    // mapped result words, all on the instance's stream
    if (ncclAllReduce(in->dResult, in->dResult, 1, ncclDouble, ncclSum, in->comm, live(in)) != ncclSuccess) return BEAGLE_ERROR_GENERAL;
    const unsigned long long seq = ++in->resultSeq;
    mi355::launchRootFinal(live(in), in->dResult, 1, in->hResultDev, (unsigned long long*)(in->hResultDev + 8), seq);
    HIP_TRY(hipGetLastError());
    rc = sitePrefetchAfterRoot(in); if (rc) return rc;          // (behind the publishing kernel: neither the collective nor the result waits for the copy)
    { const int rcw = waitResult(in, seq); if (rcw) return rcw; }
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
    const double v = in->hResult[0];
    *outGlobalSum = v;
    return (v != v) ? BEAGLE_ERROR_FLOATING_POINT : BEAGLE_SUCCESS;
}

int beagleMi355Synchronize(int instance) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleMi355Synchronize(h); }); }
    GET_INSTANCE_KEEP_PENDING(instance);                      // (a held-back pre-order list is not work in flight)
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
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
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;
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
    in->statMicroOps = in->statStored = in->statMemReads = in->statTipReads = in->statScaleReads = in->statWalks = in->statScaleWrites = in->statFastWalks = in->statFused = 0;
    in->timing = enable != 0;
    in->timingEvery = enable > 1 ? enable : 1; in->timingTick = 0;
    // event pairs for the calls to come are created here, not inside the region being timed
    while (enable && in->events.size() < 1024) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        in->events.emplace_back(a, b);
    }
    return BEAGLE_SUCCESS;
}

int beagleMi355GetDimensions(int instance, int* out8) {
    if (!out8) return BEAGLE_ERROR_OUT_OF_RANGE;
    if (mi355::isShardedHandle(instance)) {
        memset(out8, 0, 8 * sizeof(int));
        out8[2] = shardedStates(instance); out8[3] = mi355::shardedPatternCount(instance); out8[4] = shardedCategories(instance);
        return out8[3] > 0 ? BEAGLE_SUCCESS : BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    }
    Instance* in = lookup(instance);
    if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    out8[0] = in->tipCount; out8[1] = in->partialsCount; out8[2] = in->S; out8[3] = in->P; out8[4] = in->C;
    out8[5] = in->matrixCount; out8[6] = in->scaleCount; out8[7] = in->partitionCount;
    return BEAGLE_SUCCESS;
}

int beagleMi355KernelTimerRestart(int instance) {
    if (mi355::isShardedHandle(instance)) { return mi355::shardedBroadcast(instance, [&](int h) { return beagleMi355KernelTimerRestart(h); }); }
    Instance* in = lookup(instance);
    if (!in) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    in->eventsUsed = 0; in->timedMs = 0.0; in->timedLaunches = 0; in->pendingLaunches = 0; in->timingTick = 0; in->timedCalls = 0;
    in->statMicroOps = in->statStored = in->statMemReads = in->statTipReads = in->statScaleReads = in->statWalks = in->statScaleWrites = in->statFastWalks = in->statFused = 0;
    return BEAGLE_SUCCESS;
}

int beagleMi355KernelTimerCalls(int instance, long* outCalls) {
    if (mi355::isShardedHandle(instance)) {             // shard 0's (every shard brackets the same calls)
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355KernelTimerCalls(h, outCalls); });
    }
    Instance* in = lookup(instance);
    if (!in || !outCalls) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    *outCalls = in->timedCalls;
    in->timedCalls = 0;
    return BEAGLE_SUCCESS;
}

int beagleMi355RootFusedCount(int instance, long* outCount) {
    if (mi355::isShardedHandle(instance)) {
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355RootFusedCount(h, outCount); });
    }
    Instance* in = lookup(instance);
    if (!in || !outCount) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    *outCount = in->statRootFused;
    return BEAGLE_SUCCESS;
}

int beagleMi355SitePrefetchCount(int instance, long* outCount) {
    if (mi355::isShardedHandle(instance)) {
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355SitePrefetchCount(h, outCount); });
    }
    Instance* in = lookup(instance);
    if (!in || !outCount) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    *outCount = in->statSitePrefetched;
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

int beagleMi355WalkHealth(int instance, long* out4) {
    if (mi355::isShardedHandle(instance)) {             // shard 0's
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355WalkHealth(h, out4); });
    }
    GET_INSTANCE_KEEP_PENDING(instance);
    if (!out4) return BEAGLE_ERROR_OUT_OF_RANGE;
    unsigned served = 0;
    if (in->walkSelfServed) { int rc = download(in, &served, in->walkSelfServed, sizeof(served)); if (rc) return rc; }
    out4[0] = (long)served; out4[1] = (long)(in->walkSpinLimit / 100ull); out4[2] = in->statFoldedVectors; out4[3] = in->statFoldBuilds;
    return BEAGLE_SUCCESS;
}

int beagleMi355WalkLaunchInfo(int instance, long* out4) {       // (eight values)
    if (mi355::isShardedHandle(instance)) {             // shard 0's
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355WalkLaunchInfo(h, out4); });
    }
    Instance* in = lookup(instance);
    if (!in || !out4) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    out4[0] = in->statTicketWalks; out4[1] = in->statFlagWalks; out4[2] = in->lastLaunchRows; out4[3] = in->lastLaunchSlices;
    out4[4] = in->statFused; out4[5] = in->statMicroOps; out4[6] = in->statSliceAccum; out4[7] = in->statRootPartsFused;
    return BEAGLE_SUCCESS;
}

int beagleMi355GradientStats(int instance, long* out4) {
    if (mi355::isShardedHandle(instance)) {             // counters of shard 0 (every shard is driven the same way)
        bool first = true; std::mutex mu;
        return mi355::shardedBroadcast(instance, [&](int h) { { std::lock_guard<std::mutex> l(mu); if (!first) return 0; first = false; } return beagleMi355GradientStats(h, out4); });
    }
    Instance* in = lookup(instance);
    if (!in || !out4) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
    out4[0] = in->statFusedGradients; out4[1] = in->statPreLists; out4[2] = in->statWalkedGradients; out4[3] = in->statLateLists;
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
    beagleMi355CalculateRootLogLikelihoodsAllReduce,
};
const BeagleApi* beagleGetApiTable(void) { return &g_api; }

static const BeaglePartitionApi g_partitionApi = {
    beagleSetCategoryRatesWithIndex,
    beagleUpdateTransitionMatricesWithMultipleModels,
    beagleUpdatePartialsByPartition,
    beagleResetScaleFactorsByPartition,
    beagleAccumulateScaleFactorsByPartition,
    beagleCalculateRootLogLikelihoodsByPartition,
};
const BeaglePartitionApi* beagleGetPartitionApiTable(void) { return &g_partitionApi; }

}  // extern "C"
