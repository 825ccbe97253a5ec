// engine_instance.cpp — instance table, HBM buffer management (lazy slabs), the pinned staging ring, resources, the
// pair-interleaved layout of walk instances.  See engine_internal.h.
#include "engine_internal.h"

using mi355::OpDesc;

namespace mi355 {
namespace eng {

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
long stage(Instance* in, const void* src, size_t bytes, size_t reserve) {
    const size_t need = (std::max(bytes, reserve) + 255) & ~(size_t)255;
    if (need > RING_BYTES) return -1;
    if (in->ringHead + need > RING_BYTES) {
        if (hipStreamSynchronize(live(in)) != hipSuccess) return -1;         // (live: what is queued from the ring goes first)
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
        return queueCopy(in, dst, (size_t)off, bytes);
    }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, live(in)));
    HIP_TRY(hipStreamSynchronize(live(in)));
    return 0;
}

// host array -> the device mirror of the ring (transient kernel arguments: op descriptors, index lists)
int uploadTransient(Instance* in, const void* src, size_t bytes, void** dptr) {
    long off = stage(in, src, bytes);
    if (off < 0) return BEAGLE_ERROR_GENERAL;
    *dptr = in->dRing + off;
    return queueCopy(in, in->dRing + off, (size_t)off, bytes);
}

int queueCopy(Instance* in, void* dst, size_t off, size_t bytes) {
    if (bytes == 0) return 0;
    // a walk whose launch is held back must not see what the caller uploads after its updatePartials: it goes first — except for
    // the root's own inputs (category weights, state frequencies: the reference sets them between updatePartials and the root call)
    if (in->pendingWalk.valid && !in->copyKeepsWalk) { const int rcw = flushWalk(in); if (rcw) return rcw; }
    if (!in->kernelUploads) {
        HIP_TRY(hipMemcpyAsync(dst, in->hRing + off, bytes, hipMemcpyHostToDevice, in->stream));
        return 0;
    }
    Instance::PendingCopy pc; pc.dst = dst; pc.ringOff = off; pc.bytes = bytes;
    in->pendingCopies.push_back(pc);
    return 0;
}

// Everything queued, by one kernel per HOST_COPY_MAX arrays: each workgroup moves 4 KiB from the mapped ring (a coalesced
// read over the host link) to its device destination.
int flushUploads(Instance* in) {
    std::vector<Instance::PendingCopy>& pc = in->pendingCopies;
    int rc = 0;
    for (size_t i = 0; i < pc.size();) {
        mi355::HostCopyList L;
        L.n = 0;
        unsigned blocks = 0;
        for (; i < pc.size() && L.n < mi355::HOST_COPY_MAX; i++) {
            // the copies of one launch run side by side: an earlier one that this one covers entirely is dropped (the later upload of
            // an array wins); any other overlap with an earlier one of the launch sends this copy to the next launch, behind them
            const char* d0 = (const char*)pc[i].dst; const char* d1 = d0 + pc[i].bytes;
            bool partial = false;
            for (int a = 0; a < L.n; a++) {
                const char* e0 = (const char*)L.e[a].dst; const char* e1 = e0 + L.e[a].bytes;
                if (L.e[a].bytes == 0 || !(d0 < e1 && e0 < d1)) continue;
                if (d0 <= e0 && e1 <= d1) L.e[a].bytes = 0; else partial = true;
            }
            if (partial) break;
            mi355::HostCopyList::Entry& e = L.e[L.n++];
            e.dst = pc[i].dst; e.src = in->hRingDev + pc[i].ringOff; e.bytes = (unsigned)pc[i].bytes; e.firstBlock = blocks;
            blocks += (unsigned)((pc[i].bytes + 4095) / 4096);
        }
        mi355::launchHostCopies(in->stream, L, (int)blocks);
    }
    pc.clear();
    if (hipGetLastError() != hipSuccess) rc = BEAGLE_ERROR_GENERAL;
    if (rc) { int none = 0; in->asyncError.compare_exchange_strong(none, rc); }          // (the first error sticks)
    return rc;
}

int download(Instance* in, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, live(in)));
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->pendingCopies.empty() && !in->pendingWalk.valid) in->ringHead = 0;   // everything staged so far has been consumed
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
        if (hipMemsetAsync(slab, 0, bytes * n, live(in)) != hipSuccess) return BEAGLE_ERROR_GENERAL;
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
    in->pendingCopies.clear();
    if (in->ownStream) hipStreamSynchronize(in->ownStream);
    if (in->comm) { if (in->stream) hipStreamSynchronize(in->stream); ncclCommDestroy(in->comm); in->comm = nullptr; }
    if (in->stream && in->stream != in->ownStream) hipStreamSynchronize(in->stream);
    for (void* p : in->allocations) hipFree(p);
    if (in->bigStage) hipFree(in->bigStage);
    if (in->matStream) hipFree(in->matStream);
    if (in->walkFlags) hipFree(in->walkFlags);
    if (in->sliceMant) hipFree(in->sliceMant);
    if (in->sliceExp) hipFree(in->sliceExp);
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

Resources* g_resources = nullptr;

extern const long GPU_FLAGS;
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
    HIP_TRY(hipMemsetAsync(p, in->S, tipBytes, live(in)));
    double* ones = (double*)((char*)p + ((tipBytes + 255) & ~(size_t)255));
    mi355::launchFill(live(in), ones, 1.0, 0, (int)in->scaleStride);
    HIP_TRY(hipGetLastError());
    in->dummyTips = (uint8_t*)p; in->onesScale = ones;
    return 0;
}


}  // namespace eng
}  // namespace mi355
