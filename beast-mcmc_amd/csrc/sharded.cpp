// sharded.cpp — the pattern-sharded, multi-GPU instance INSIDE the library: resource G+1 = "all G GPUs of this node".
//
// BEAST already knows how to use several GPUs from one JVM: -beagle_instances G -beagle_order 1,..,G makes G
// BeagleDataLikelihoodDelegates over contiguous pattern blocks (src/dr/evolution/alignment/Patterns.java:142-167), a Java
// thread pool evaluates them and Java adds the G values (src/dr/evomodelxml/treedatalikelihood/TreeDataLikelihoodParser.java:
// 205-278, src/dr/inference/model/CompoundLikelihood.java:202-214).  This file offers the same split behind ONE instance
// handle (SURVEY 8b "Resource numbering", 8e): the caller asks for resource G+1 and drives it exactly like a single-GPU
// instance; the library
//   * cuts the P unique patterns into G contiguous blocks with BEAST's own block sizes (first P % G blocks get one more),
//   * keeps one ordinary engine instance per GPU for its block, each with its own HIP stream, driven by its own host thread
//     (the host-side planning of one evaluation takes about as long as a 12 500-pattern shard's kernels, so G shards
//     prepared one after the other by one thread would serialise the GPUs),
//   * replicates everything that is KB-sized (tree operations, eigen systems, matrices, rates, weights, frequencies),
//   * slices what is indexed by pattern (tip states / partials, pattern weights, partition map) on the way in and gathers
//     it on the way out (site log-likelihoods, partials, scale factors),
//   * and reduces the per-shard root log-likelihoods with ONE ncclAllReduce(sum, ncclDouble, count = 1, or partitionCount
//     for ...ByPartition) per evaluation over RCCL/xGMI, issued on each shard's stream right behind its reduction kernel;
//     the host reads the result from shard 0.  Deterministic: RCCL's ring order is fixed for a fixed communicator.
// Gradient sums (edge differentials, cross products) are small host-side sums of what each shard returns.
//
// BEAGLE_MI355_SHARDS=n (tests on a one-GPU box): n shards placed round-robin on the visible devices.  RCCL needs distinct
// devices per rank, so when two shards share a GPU the G partial sums are added on the host instead, in shard order.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/beagle_mi355.h"
#include "sharded.h"

namespace mi355 {

thread_local int tlsWholePatternCount = 0;
namespace {

// one host thread per shard: runs the closures the API thread posts, in order.  post() does not wait: calls that return
// nothing to the caller (model parameters, matrices, operation lists, scale-factor bookkeeping) are queued and the API thread
// moves on to queue the next one — a hand-off per call and shard would cost more than a 12 500-pattern shard's kernels
// (eight shards: two context switches x 8 x ~7 calls per evaluation); wait() drains the queue and returns the first error
// since the last wait (BEAGLE's own asynchronous semantics: an error surfaces at the next call that observes results).
class Worker {
public:
    Worker() : thread_([this] { loop(); }) {}
    ~Worker() {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; stopFlag_.store(true); }
        cv_.notify_all();
        thread_.join();
    }
    void post(std::function<int()> f) {
        { std::lock_guard<std::mutex> l(mu_); q_.push_back(std::move(f)); pending_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
    }
    int wait() {
        // the shard threads answer within microseconds: poll first (a condition-variable wake-up is 5-50 us of scheduler
        // latency per call under a container's CPU quota), block only when the work is long
        for (int spin = 0; spin < 20000 && pending_.load(std::memory_order_acquire) != 0; spin++) __builtin_ia32_pause();
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [this] { return q_.empty() && !busy_; });
        const int rc = rc_;
        rc_ = 0;
        return rc;
    }
private:
    void loop() {
        for (;;) {
            std::function<int()> f;
            // an evaluation is a burst of ~8 posted calls some microseconds apart: poll for the next one for a while before
            // going to sleep (BEAGLE_MI355_SHARD_SPIN_US, default 200; 0 = always block)
            for (long spin = 0; spin < spinIters_ && pending_.load(std::memory_order_acquire) == 0 && !stopFlag_.load(std::memory_order_relaxed); spin++)
                __builtin_ia32_pause();
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return !q_.empty() || stop_; });
                if (q_.empty()) return;                 // (stop_, and nothing left to run)
                f = std::move(q_.front()); q_.pop_front(); busy_ = true;
            }
            const int rc = f();
            { std::lock_guard<std::mutex> l(mu_); if (rc && !rc_) rc_ = rc; busy_ = false; pending_.fetch_sub(1, std::memory_order_release); }
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<int()>> q_;
    bool busy_ = false, stop_ = false;
    std::atomic<int> pending_{0};
    std::atomic<bool> stopFlag_{false};
    long spinIters_ = (getenv("BEAGLE_MI355_SHARD_SPIN_US") ? atol(getenv("BEAGLE_MI355_SHARD_SPIN_US")) : 200) * 25;     // ~40 ns per pause
    int rc_ = 0;
    std::thread thread_;
};

struct Shard {
    int handle = -1, device = 0;
    int pStart = 0, pEnd = 0;
    hipStream_t stream = nullptr;
    double* dResult = nullptr;          // device buffer the shard's root sums land in (all-reduced in place)
    ncclComm_t comm = nullptr;
    Worker* worker = nullptr;
};

struct Sharded {
    std::vector<Shard> shards;
    int tipCount = 0, S = 0, P = 0, C = 0;
    bool useRccl = false, eigenComplex = false;
    double* hResult = nullptr;          // pinned
    std::string name;
};

std::mutex g_mu;
std::vector<Sharded*> g_sharded;

Sharded* find(int handle) {
    std::lock_guard<std::mutex> l(g_mu);
    const int i = handle - SHARD_HANDLE_BASE;
    if (i < 0 || i >= (int)g_sharded.size()) return nullptr;
    return g_sharded[i];
}

// run f(shard index) on every shard's thread; first non-zero return code wins
int forAll(Sharded* sh, const std::function<int(int)>& f) {
    const int n = (int)sh->shards.size();
    for (int k = 0; k < n; k++) sh->shards[k].worker->post([&f, k] { return f(k); });
    int rc = 0;
    for (int k = 0; k < n; k++) { const int r = sh->shards[k].worker->wait(); if (r && !rc) rc = r; }
    return rc;
}

// everything a (possibly half-built) sharded instance owns: worker threads first (nothing is in flight afterwards), the
// shard instances — which must go before the streams they were pointed at —, then communicator, result buffer, stream
void destroySharded(Sharded* sh) {
    for (Shard& s : sh->shards) {
        delete s.worker;
        if (s.handle >= 0) beagleFinalizeInstance(s.handle);
        hipSetDevice(s.device);
        if (s.comm) ncclCommDestroy(s.comm);
        if (s.dResult) hipFree(s.dResult);
        if (s.stream) hipStreamDestroy(s.stream);
    }
    if (sh->hResult) hipHostFree(sh->hResult);
    delete sh;
}

#define GET_SHARDED(h) Sharded* sh = find(h); if (!sh) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE

}  // namespace

int shardedDeviceCountOverride() {
    const char* e = getenv("BEAGLE_MI355_SHARDS");
    return e ? atoi(e) : 0;
}

int shardedCreate(int gpuCount, int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount, int patternCount,
                  int eigenBufferCount, int matrixBufferCount, int categoryCount, int scaleBufferCount, long preferenceFlags,
                  long requirementFlags, BeagleInstanceDetails* returnInfo) {
    int n = shardedDeviceCountOverride();
    if (n <= 0) n = gpuCount;
    n = std::max(1, std::min(n, patternCount));
    Sharded* sh = new Sharded();
    sh->tipCount = tipCount; sh->S = stateCount; sh->P = patternCount; sh->C = categoryCount;
    sh->eigenComplex = (requirementFlags & BEAGLE_FLAG_EIGEN_COMPLEX) != 0;
    sh->useRccl = n <= gpuCount;                       // distinct devices: the all-reduce runs over RCCL
    const int div = patternCount / n, rem = patternCount % n;      // Patterns.java:142-167
    int start = 0, rc = 0;
    for (int k = 0; k < n && !rc; k++) {
        Shard s;
        s.device = k % gpuCount;
        s.pStart = start; s.pEnd = start + div + (k < rem ? 1 : 0); start = s.pEnd;
        const int res = s.device + 1;
        tlsWholePatternCount = patternCount;
        s.handle = beagleCreateInstance(tipCount, partialsBufferCount, compactBufferCount, stateCount, s.pEnd - s.pStart, eigenBufferCount,
                                        matrixBufferCount, categoryCount, scaleBufferCount, &res, 1, preferenceFlags, requirementFlags, nullptr);
        tlsWholePatternCount = 0;
        if (s.handle < 0) { rc = s.handle; break; }
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void**)&s.dResult, 4096) != hipSuccess) { rc = BEAGLE_ERROR_OUT_OF_MEMORY; sh->shards.push_back(s); break; }
        rc = beagleMi355SetStream(s.handle, s.stream);
        s.worker = new Worker();
        sh->shards.push_back(s);
    }
    if (!rc && hipHostMalloc((void**)&sh->hResult, 4096, hipHostMallocDefault) != hipSuccess) rc = BEAGLE_ERROR_OUT_OF_MEMORY;
    if (!rc && sh->useRccl) {
        std::vector<int> devs(n);
        std::vector<ncclComm_t> comms(n);
        for (int k = 0; k < n; k++) devs[k] = sh->shards[k].device;
        if (ncclCommInitAll(comms.data(), n, devs.data()) != ncclSuccess) rc = BEAGLE_ERROR_GENERAL;
        else for (int k = 0; k < n; k++) sh->shards[k].comm = comms[k];
    }
    if (rc) { destroySharded(sh); return rc; }
    sh->name = std::to_string(n) + " x MI355X, patterns sharded" + (sh->useRccl ? " (RCCL all-reduce)" : " (host sum)");
    int handle;
    {
        std::lock_guard<std::mutex> l(g_mu);
        g_sharded.push_back(sh);
        handle = SHARD_HANDLE_BASE + (int)g_sharded.size() - 1;
    }
    if (returnInfo) {
        returnInfo->resourceNumber = gpuCount + 1;
        returnInfo->resourceName = (char*)sh->name.c_str();
        returnInfo->implName = (char*)"HIP-gfx950-fp64-sharded";
        returnInfo->implDescription = (char*)"one engine instance per GPU over contiguous pattern blocks, one RCCL all-reduce per evaluation";
        returnInfo->flags = BEAGLE_FLAG_PRECISION_DOUBLE | BEAGLE_FLAG_COMPUTATION_SYNCH | BEAGLE_FLAG_EIGEN_REAL | BEAGLE_FLAG_SCALING_MANUAL |
                            BEAGLE_FLAG_SCALERS_RAW | BEAGLE_FLAG_VECTOR_NONE | BEAGLE_FLAG_THREADING_NONE | BEAGLE_FLAG_PROCESSOR_GPU |
                            BEAGLE_FLAG_PARALLELOPS_GRID;
    }
    return handle;
}

int shardedFinalize(int handle) {
    Sharded* sh;
    {
        std::lock_guard<std::mutex> l(g_mu);
        const int i = handle - SHARD_HANDLE_BASE;
        if (i < 0 || i >= (int)g_sharded.size() || !g_sharded[i]) return BEAGLE_ERROR_UNINITIALIZED_INSTANCE;
        sh = g_sharded[i]; g_sharded[i] = nullptr;
    }
    destroySharded(sh);
    return BEAGLE_SUCCESS;
}

int shardedShardCount(int handle) { Sharded* sh = find(handle); return sh ? (int)sh->shards.size() : -1; }
int shardedCommRanks(int handle) {
    Sharded* sh = find(handle);
    if (!sh || !sh->useRccl || sh->shards.empty() || !sh->shards[0].comm) return 0;
    int n = 0;
    return ncclCommCount(sh->shards[0].comm, &n) == ncclSuccess ? n : 0;
}

int shardedBroadcast(int handle, const std::function<int(int)>& call) {
    GET_SHARDED(handle);
    return forAll(sh, [&](int k) { return call(sh->shards[k].handle); });
}

int shardedPost(int handle, std::function<int(int)> call) {
    GET_SHARDED(handle);
    auto shared = std::make_shared<const std::function<int(int)>>(std::move(call));
    for (Shard& s : sh->shards) { const int h = s.handle; s.worker->post([shared, h] { return (*shared)(h); }); }
    return BEAGLE_SUCCESS;
}

// ---- inputs indexed by pattern ---------------------------------------------------------------------------------------
int shardedSetPerPatternInts(int handle, const int* v, const std::function<int(int, const int*)>& call) {
    GET_SHARDED(handle);
    return forAll(sh, [&](int k) { return call(sh->shards[k].handle, v + sh->shards[k].pStart); });
}
int shardedSetPerPatternDoubles(int handle, const double* v, int perPattern, int planes, const std::function<int(int, const double*)>& call) {
    // v is [planes][P][perPattern]; a shard gets [planes][its patterns][perPattern]
    GET_SHARDED(handle);
    return forAll(sh, [&](int k) {
        const Shard& s = sh->shards[k];
        const size_t n = (size_t)(s.pEnd - s.pStart) * perPattern;
        if (planes == 1) return call(s.handle, v + (size_t)s.pStart * perPattern);
        std::vector<double> tmp(n * planes);
        for (int c = 0; c < planes; c++) memcpy(&tmp[n * c], v + ((size_t)c * sh->P + s.pStart) * perPattern, n * sizeof(double));
        return call(s.handle, tmp.data());
    });
}
// ---- outputs indexed by pattern --------------------------------------------------------------------------------------
int shardedGetPerPatternDoubles(int handle, double* out, int perPattern, int planes, const std::function<int(int, double*)>& call) {
    GET_SHARDED(handle);
    return forAll(sh, [&](int k) {
        const Shard& s = sh->shards[k];
        const size_t n = (size_t)(s.pEnd - s.pStart) * perPattern;
        if (planes == 1) return call(s.handle, out + (size_t)s.pStart * perPattern);
        std::vector<double> tmp(n * planes);
        const int rc = call(s.handle, tmp.data());
        for (int c = 0; c < planes && !rc; c++) memcpy(out + ((size_t)c * sh->P + s.pStart) * perPattern, &tmp[n * c], n * sizeof(double));
        return rc;
    });
}
int shardedGetPerPatternInts(int handle, int* out, const std::function<int(int, int*)>& call) {
    GET_SHARDED(handle);
    return forAll(sh, [&](int k) { return call(sh->shards[k].handle, out + sh->shards[k].pStart); });
}

// ---- the reduction: per-shard root sums -> one all-reduce -> host -----------------------------------------------------
int shardedRootReduce(int handle, int count, const std::function<int(int shardHandle, double* deviceOut)>& enqueue, double* outValues) {
    GET_SHARDED(handle);
    if (count < 1 || count > 512) return BEAGLE_ERROR_OUT_OF_RANGE;
    const int n = (int)sh->shards.size();
    int rc = forAll(sh, [&](int k) { return enqueue(sh->shards[k].handle, sh->shards[k].dResult); });
    if (rc) return rc;
    if (sh->useRccl) {
        // every rank's call sits on its shard's stream, right behind the kernels that produce its operand
        if (ncclGroupStart() != ncclSuccess) return BEAGLE_ERROR_GENERAL;
        for (int k = 0; k < n; k++) {
            Shard& s = sh->shards[k];
            if (hipSetDevice(s.device) != hipSuccess ||
                ncclAllReduce(s.dResult, s.dResult, (size_t)count, ncclDouble, ncclSum, s.comm, s.stream) != ncclSuccess) { ncclGroupEnd(); return BEAGLE_ERROR_GENERAL; }
        }
        if (ncclGroupEnd() != ncclSuccess) return BEAGLE_ERROR_GENERAL;
        // the result reaches the host from shard 0: one small kernel behind its all-reduce writes the values and a sequence word
        // into mapped host memory, which this thread polls (engine_abi.cpp publishAndWait) — a device-to-host copy, a stream
        // synchronisation and a synchronisation of every other shard cost more than a small shard's kernels.  The other shards
        // are not waited for: their rank of the all-reduce has contributed when shard 0's completes, and what follows on their
        // streams is ordered behind it by the streams themselves (their staging rings drain when they wrap).
        // (the result page holds 480 doubles behind its header: a longer vector — up to 512 partitions — goes in two pieces)
        for (int b = 0; b < count; b += 480) {
            const int rcp = publishAndWait(sh->shards[0].handle, sh->shards[0].dResult + b, std::min(480, count - b), outValues + b);
            if (rcp) return rcp;
        }
        // a deferred error of another shard (an upload kernel that could not be launched) must not go unseen just because nobody
        // waits on that shard: the first one found fails this call, as shard 0's own would inside publishAndWait
        for (int k = 1; k < n; k++) { const int rce = takeAsyncError(sh->shards[k].handle); if (rce) return rce; }
        return BEAGLE_SUCCESS;
    } else {
        // no communicator (one GPU, or BEAGLE_MI355_SHARDS on one): every shard's sums reach the host the way a single instance's do — a
        // small kernel behind the shard's root kernels writes them and a sequence word into mapped host memory, which this thread polls
        // (engine_abi.cpp publishAndWait; it resets the shard's staging ring) — and are added in shard order.  A device-to-host copy and a
        // stream synchronisation per shard cost 20-50 us of an evaluation (round 6: the reference's benchmark1 alignment through this
        // handle on one GPU 7 080 -> see profiles/r06_experiments.txt 25).
        std::vector<double> acc(count, 0.0), part(count);
        for (int k = 0; k < n; k++) {
            Shard& s = sh->shards[k];
            if (hipSetDevice(s.device) != hipSuccess) return BEAGLE_ERROR_GENERAL;
            for (int b = 0; b < count; b += 480) {
                const int rcp = publishAndWait(s.handle, s.dResult + b, std::min(480, count - b), part.data() + b);
                if (rcp) return rcp;
            }
            for (int q = 0; q < count; q++) acc[q] += part[q];
        }
        memcpy(outValues, acc.data(), (size_t)count * sizeof(double));
    }
    return BEAGLE_SUCCESS;
}

// per-shard host results that simply add up (gradient sums): call fills `len` doubles per shard
int shardedSumDoubles(int handle, int len, const std::function<int(int shardHandle, double* out)>& call, double* outSum) {
    GET_SHARDED(handle);
    const int n = (int)sh->shards.size();
    std::vector<std::vector<double>> part(n, std::vector<double>(len, 0.0));
    int rc = forAll(sh, [&](int k) { return call(sh->shards[k].handle, part[k].data()); });
    if (rc) return rc;
    for (int q = 0; q < len; q++) { double a = 0.0; for (int k = 0; k < n; k++) a += part[k][q]; outSum[q] = a; }
    return BEAGLE_SUCCESS;
}

void shardedBounds(int handle, int shard, int* pStart, int* pEnd) {
    Sharded* sh = find(handle);
    if (!sh || shard < 0 || shard >= (int)sh->shards.size()) { *pStart = *pEnd = 0; return; }
    *pStart = sh->shards[shard].pStart; *pEnd = sh->shards[shard].pEnd;
}
void shardedBoundsOfHandle(int handle, int shardHandle, int* pStart, int* pEnd) {
    Sharded* sh = find(handle);
    *pStart = *pEnd = 0;
    if (!sh) return;
    for (const Shard& s : sh->shards) if (s.handle == shardHandle) { *pStart = s.pStart; *pEnd = s.pEnd; }
}
int shardedPatternCount(int handle) { Sharded* sh = find(handle); return sh ? sh->P : 0; }
int shardedStates(int handle) { Sharded* sh = find(handle); return sh ? sh->S : 0; }
bool shardedEigenComplex(int handle) { Sharded* sh = find(handle); return sh && sh->eigenComplex; }
int shardedCategories(int handle) { Sharded* sh = find(handle); return sh ? sh->C : 0; }

}  // namespace mi355
