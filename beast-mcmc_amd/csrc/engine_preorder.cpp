// engine_preorder.cpp — pre-order partials and the gradient sums built on them (kernels_preorder.hip).
// See engine_internal.h.
#include "engine_internal.h"

using mi355::OpDesc;

namespace mi355 {
namespace eng {

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
    HIP_TRY(hipMemsetAsync(in->preMissing, in->S, (size_t)in->P, live(in)));
    in->preScratch.assign(PRE_SCRATCH, nullptr);
    for (int j = 0; j < PRE_SCRATCH; j++) in->preScratch[j] = (double*)((char*)slab + in->partialsBytes * j);
    return 0;
}

// the block sums of the derivative calls: kept between calls (hipMalloc/hipFree per call cost more than the small trees' kernels)
int ensureEdgeScratch(Instance* in, size_t bytes) {
    if (bytes <= in->edgeScratchBytes) return 0;
    HIP_TRY(hipStreamSynchronize(live(in)));
    if (in->edgeScratch) {
        for (auto& a : in->allocations) if (a == in->edgeScratch) { a = in->allocations.back(); in->allocations.pop_back(); break; }
        hipFree(in->edgeScratch); in->deviceBytes -= in->edgeScratchBytes; in->edgeScratch = nullptr; in->edgeScratchBytes = 0;
    }
    bytes = (bytes * 5 / 4 + 4095) & ~(size_t)4095;
    int rc = devAlloc(in, &in->edgeScratch, bytes); if (rc) return rc;
    in->edgeScratchBytes = bytes;
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
        mi355::launchTransposeMatrices(live(in), in->matrices, (const int*)dPairs, n, in->S, in->C);
        mi355::launchPruneLevelTiled(live(in), (const OpDesc*)dPass, n, in->matrices, in->P, in->S, in->C, false);
        mi355::launchPruneLevelTiled(live(in), (const OpDesc*)dPass + n, n, in->matrices, in->P, in->S, in->C, anyWrite);
    }
    return 0;
}

// Does the pre-order walk (kernels_preorder4.hip k_preWalk4) re-evaluate this post-order operand by itself?  A definition of one
// step (a node over two compact tips) or two (such a node under one more compact tip): what a gradient chain's post-order passes
// leave unstored (engine_walk.cpp runOperationsWalk: planner.h stepLimit) — about half the nodes of a coalescent tree.
static bool walkableDefinition(const Instance* in, int buf) {
    if (!in->gradientVirtual || in->partitionCount != 1 || !isVirt(in, buf)) return false;
    const mi355::VirtDef& d = in->planner.definition(in->planner.key(buf, 0));
    if (!d.on || d.nSteps < 1 || d.nSteps > 2) return false;
    const mi355::VirtStep& s0 = d.steps[0];
    if (s0.type != mi355::VT_CHERRY || s0.memA || s0.memB || !in->tipStates[s0.tipA] || !in->tipStates[s0.tipB]) return false;
    if (d.nSteps == 2) {
        const mi355::VirtStep& s1 = d.steps[1];
        if (s1.type != mi355::VT_EXTEND || s1.subA != 0 || s1.memB || !in->tipStates[s1.tipB]) return false;
    }
    return true;
}
// the held list is about to run on kernels that read real partials: its unstored operands get theirs
static int materializeHeldOperands(Instance* in) {
    if (!in->virt) return 0;
    std::vector<int> need;
    for (const Instance::HeldPreNode& nd : in->heldPre.nodes) {
        if (isVirt(in, nd.postA)) in->planner.keysOf(nd.postA, need);
        if (isVirt(in, nd.postB)) in->planner.keysOf(nd.postB, need);
    }
    return need.empty() ? 0 : materializeList(in, need);
}

// Enqueue a pre-order op list (7-int tuples {pre(child), writeScale, readScale, pre(parent), matrix(child), post(sibling),
// matrix(sibling)}, AbstractBeagleGradientDelegate.java:207-221).  A parent's op precedes its children's; the list is
// levelised like a post-order one and each level is one launch.
// tuple = 9 (beagleUpdatePrePartialsByPartition): {.., partition, cumulativeScale} — the operation covers that partition's patterns.
// A partitioned instance takes 7-int lists too (whole pattern range); nothing is held back there (the walk is single-partition).
int runPreOperations(Instance* in, const int* ops, int count, int globalCum, bool mayHold, int tuple) {
    if (count <= 0) return 0;
    if (in->partitionCount != 1 || tuple != BEAGLE_OP_COUNT) mayHold = false;
    const int n = in->partialsCount;
    // The whole list is checked BEFORE anything changes — indices, and that every operand will resolve (real data, a definition
    // that can be materialised, or a destination of an earlier operation of this list; a scale buffer to read holds raw
    // factors or is written earlier in the list): a list that fails must leave the instance — and a list still held back,
    // which the new one may replace unexecuted below — exactly as they were.
    {
        std::vector<char> written(n, 0), scaleWritten(std::max(1, in->scaleCount), 0);
        for (int k = 0; k < count; k++) {
            const int* op = ops + (size_t)k * tuple;
            const int dest = op[0], wS = op[1], rS = op[2], par = op[3], mc = op[4], sib = op[5], ms = op[6];
            if (badIndex(dest, n) || badIndex(par, n) || badIndex(sib, n) || badIndex(mc, in->matrixCount) || badIndex(ms, in->matrixCount) ||
                (wS != BEAGLE_OP_NONE && badIndex(wS, in->scaleCount)) || (rS != BEAGLE_OP_NONE && badIndex(rS, in->scaleCount)) ||
                (globalCum != BEAGLE_OP_NONE && badIndex(globalCum, in->scaleCount)) || dest == par || dest == sib)
                return BEAGLE_ERROR_OUT_OF_RANGE;
            if (tuple > BEAGLE_OP_COUNT && (badIndex(op[7], in->partitionCount) || (op[8] != BEAGLE_OP_NONE && badIndex(op[8], in->scaleCount))))
                return BEAGLE_ERROR_OUT_OF_RANGE;
            auto hasData = [&](int b) { return written[b] || (in->partials[b] != nullptr && !isCompactTip(in, b)) || isVirt(in, b); };
            if (!hasData(par)) return BEAGLE_ERROR_OUT_OF_RANGE;
            if (!(hasData(sib) || (!written[sib] && isCompactTip(in, sib)))) return BEAGLE_ERROR_OUT_OF_RANGE;
            if (wS == BEAGLE_OP_NONE && rS != BEAGLE_OP_NONE && !(scaleWritten[rS] || (in->scale[rS] && in->scaleIsRaw[rS]))) return BEAGLE_ERROR_OUT_OF_RANGE;
            written[dest] = 1;
            if (wS != BEAGLE_OP_NONE) scaleWritten[wS] = 1;
        }
    }
    if (mayHold && in->S == 4 && in->fuseGradient) in->trackScales = true;     // (from now on updatePartials records scale indices: engine_abi.cpp)
    // everything these ops read must be real data — except, for a list that is going to be held back and walked, the short
    // definitions the walk re-evaluates itself —, and nothing they overwrite may still define a virtual buffer
    const bool mayWalk = mayHold && in->fuseGradient && in->S == 4 && !in->tiled && globalCum == BEAGLE_OP_NONE && in->gradientVirtual && in->preWalk;
    std::vector<int> need, deferred;
    long unstoredOperands = 0;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        const int dest = op[0], wS = op[1], par = op[3], sib = op[5];
        in->scaleOfPartial[dest] = -1;                    // (a pre-order partial: never a post-order operand of the walk)
        if (wS != BEAGLE_OP_NONE) in->scaleVersion[wS]++;
        if (isVirt(in, sib)) {
            unstoredOperands++;
            if (mayWalk && wS == BEAGLE_OP_NONE && op[2] == BEAGLE_OP_NONE && walkableDefinition(in, sib)) deferred.push_back(sib);
            else in->planner.keysOf(sib, need);
        }
        if (isVirt(in, par)) in->planner.keysOf(par, need);
        if (in->virt) {
            need.insert(need.end(), in->planner.tipUsers(dest).begin(), in->planner.tipUsers(dest).end());
            if (wS != BEAGLE_OP_NONE) need.insert(need.end(), in->planner.scaleUsers(wS).begin(), in->planner.scaleUsers(wS).end());
        }
    }
    // a list still held back: the new one replaces it unexecuted when it rewrites everything the old one would have written
    // without reading any of it first (the chain's next gradient: same destinations, other post-order buffers); otherwise the
    // old one runs now
    if (in->heldPre.held) {
        if (supersedesHeld(in, ops, count)) in->heldPre.held = false;
        else { int rc = executeHeldPre(in); if (rc) return rc; }
    }
    // A pre-order pass reads the post-order partials of (nearly) every node: when it has to have a good part of them
    // materialised first, the chain is evaluating gradients, and the next post-order passes store what they compute straight
    // away instead of leaving it to a second walk (runOperationsWalk; 1000 x 20 000: 1.86 -> 0.9 ms for the post-order half).
    // The hint is renewed by every pre-order list and runs out 16 post-order evaluations after the last one.
    if ((int)unstoredOperands >= std::max(4, count / 4) || in->storeAllEvaluations > 0) in->storeAllEvaluations = 16;
    if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    std::vector<OpDesc> descs(count);
    std::vector<int> level(count), wLevel(n, -1), rLevel(n, -1), opWrite(count, BEAGLE_OP_NONE);
    int maxLevel = 0;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
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
        else if (isVirt(in, sib)) d.child2 = nullptr;     // (one of `deferred`: filled in below should the list run here after all)
        else if (in->partials[sib]) d.child2 = in->partials[sib];
        else return BEAGLE_ERROR_OUT_OF_RANGE;
        d.mat1 = mc; d.mat2 = ms;
        if (wS != BEAGLE_OP_NONE) {
            rc = ensureScale(in, wS); if (rc) return rc;
            d.scaleWrite = in->scale[wS]; in->scaleIsRaw[wS] = 1; opWrite[k] = wS; scalesWritten(in);
        } else if (rS != BEAGLE_OP_NONE) {
            rc = ensureScale(in, rS); if (rc) return rc;
            if (!in->scaleIsRaw[rS]) return BEAGLE_ERROR_OUT_OF_RANGE;
            d.scaleRead = in->scale[rS];
        }
        d.pStart = 0; d.pEnd = in->P;
        if (tuple > BEAGLE_OP_COUNT) { d.pStart = in->partStart[op[7]]; d.pEnd = in->partEnd[op[7]]; }
        const int lvl = std::max(std::max(wLevel[par], wLevel[sib]), std::max(wLevel[dest], rLevel[dest])) + 1;
        level[k] = lvl; maxLevel = std::max(maxLevel, lvl);
        wLevel[dest] = lvl; rLevel[par] = std::max(rLevel[par], lvl); rLevel[sib] = std::max(rLevel[sib], lvl);
    }
    // 4 states, no rescaling in the list: hold it back (engine_internal.h HeldPreList).  The bookkeeping above (real operands,
    // destinations allocated and no longer tips or definitions) is done either way.
    if (mayHold && in->fuseGradient && in->S == 4 && !in->tiled && globalCum == BEAGLE_OP_NONE) {
        int rc = holdPreList(in, ops, count);
        if (rc <= 0) return rc;                                    // held (0) or failed (< 0); 1 = not the shape: run it now
    }
    if (!deferred.empty()) {                                       // it runs now, on kernels that read real partials
        std::vector<int> keys;
        for (int b : deferred) if (isVirt(in, b)) in->planner.keysOf(b, keys);
        if (!keys.empty()) { int rc = materializeList(in, keys); if (rc) return rc; }
        for (int k = 0; k < count; k++) {
            const int sib = ops[(size_t)k * tuple + 5];
            if (!descs[k].child2) { if (!in->partials[sib]) return BEAGLE_ERROR_OUT_OF_RANGE; descs[k].child2 = in->partials[sib]; }
        }
    }
    in->statPreLists++;
    std::vector<int> start(maxLevel + 2, 0);
    for (int k = 0; k < count; k++) start[level[k] + 1]++;
    for (int l = 0; l <= maxLevel; l++) start[l + 1] += start[l];
    std::vector<OpDesc> sorted(count);
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int k = 0; k < count; k++) sorted[fill[level[k]]++] = descs[k];
    const size_t maxChunkOps = (RING_BYTES / 4) / sizeof(OpDesc);
    // T32 layout (16..64 states): two passes of the MFMA pruning kernel per op instead of the VALU pre-order kernel
    // (BEAGLE_MI355_PRE_NAIVE=1 keeps the latter, for A/B runs)
    static const bool preNaive = labEnv("BEAGLE_MI355_PRE_NAIVE") && atoi(labEnv("BEAGLE_MI355_PRE_NAIVE")) != 0;
    const bool twoPass = in->tiled && !preNaive && tuple == BEAGLE_OP_COUNT;      // (a partition's range: the direct kernel, which takes any range)
    // a partitioned walk instance keeps its reciprocal halves in another order than the kernel's walkPairIndex: rebuilt below
    const bool recipLater = in->walk && in->partitionCount > 1;
    for (int chunkBegin = 0; chunkBegin < count;) {
        const int chunkEnd = (int)std::min<size_t>((size_t)count, (size_t)chunkBegin + maxChunkOps);
        void* dChunk = nullptr;
        int rc = twoPass ? 0 : uploadTransient(in, &sorted[chunkBegin], (size_t)(chunkEnd - chunkBegin) * sizeof(OpDesc), &dChunk);
        if (rc) return rc;
        for (int l = 0; l <= maxLevel; l++) {
            const int begin = std::max(start[l], chunkBegin), end = std::min(start[l + 1], chunkEnd);
            if (begin >= end) continue;
            if (twoPass) {
                // one pass per operation on the matrix cores (kernels_mfma.hip k_preOpTiled: three buffer transfers and two
                // products instead of five and four) unless the level rescales in write mode (BEAGLE_MI355_PRE_TWO_PASS=1: always
                // the two passes of the pruning kernel, for A/B runs)
                static const bool forceTwo = getenv("BEAGLE_MI355_PRE_TWO_PASS") && atoi(getenv("BEAGLE_MI355_PRE_TWO_PASS")) != 0;
                bool anyWrite = forceTwo;
                for (int k = begin; k < end && !anyWrite; k++) anyWrite = sorted[k].scaleWrite != nullptr;
                if (!anyWrite) {
                    void* dLevel = nullptr;
                    int rcu = uploadTransient(in, &sorted[begin], (size_t)(end - begin) * sizeof(OpDesc), &dLevel); if (rcu) return rcu;
                    if (mi355::launchPreOpsTiled(live(in), (const OpDesc*)dLevel, end - begin, in->matrices, in->P, in->S, in->C)) continue;
                }
                int rc2 = preLevelTwoPass(in, &sorted[begin], end - begin); if (rc2) return rc2; continue;
            }
            mi355::launchPrePartials(live(in), (const OpDesc*)dChunk + (begin - chunkBegin), end - begin, in->matrices,
                                     in->P, in->S, in->C, in->tiled, in->P, in->walk && !recipLater ? (long)in->scaleStride : 0);
        }
        chunkBegin = chunkEnd;
    }
    HIP_TRY(hipGetLastError());
    if (recipLater)
        for (int k = 0; k < count; k++)
            if (opWrite[k] != BEAGLE_OP_NONE && in->dPairPos)
                mi355::launchRecipFromFactors(live(in), in->scale[opWrite[k]], in->scale[opWrite[k]] + in->scaleStride, in->dPairPos, in->P);
    if (tuple > BEAGLE_OP_COUNT) {                  // per-operation cumulative buffers, over the partition's patterns
        for (int k = 0; k < count; k++) {
            const int* op = ops + (size_t)k * tuple;
            if (opWrite[k] == BEAGLE_OP_NONE || op[8] == BEAGLE_OP_NONE) continue;
            int rc = ensureScale(in, op[8]); if (rc) return rc;
            const double* src = in->scale[opWrite[k]];
            int one = 1;
            void *dSrc = nullptr, *dRaw = nullptr;
            rc = uploadTransient(in, &src, sizeof(src), &dSrc); if (rc) return rc;
            rc = uploadTransient(in, &one, sizeof(one), &dRaw); if (rc) return rc;
            mi355::launchAccumulateScale(live(in), in->scale[op[8]], (const double* const*)dSrc, (const int*)dRaw, 1, 1.0, in->partStart[op[7]], in->partEnd[op[7]]);
        }
        return 0;
    }
    if (globalCum != BEAGLE_OP_NONE)
        for (int k = 0; k < count; k++) {
            if (opWrite[k] == BEAGLE_OP_NONE) continue;
            int rc = ensureScale(in, globalCum); if (rc) return rc;
            const double* src = in->scale[opWrite[k]];
            int one = 1;
            void *dSrc = nullptr, *dRaw = nullptr;
            rc = uploadTransient(in, &src, sizeof(src), &dSrc); if (rc) return rc;
            rc = uploadTransient(in, &one, sizeof(one), &dRaw); if (rc) return rc;
            mi355::launchAccumulateScale(live(in), in->scale[globalCum], (const double* const*)dSrc, (const int*)dRaw, 1, 1.0, 0, in->P);
        }
    return 0;
}

// Per-edge derivative sums (AbstractBeagleBranchGradientDelegate.java:82-92).  Edges are processed in chunks that bound
// the scratch memory (block sums, and the optional per-pattern matrix) to a few hundred MB.
// ---- the held-back list (4 states) -----------------------------------------------------------------------------------
// Does the list `ops` write every buffer the held list would have written, reading none of them before it has written it
// itself?  (Operations come parent before child, so "its own destinations" are safe to read.)
bool supersedesHeld(Instance* in, const int* ops, int count) {
    const Instance::HeldPreList& h = in->heldPre;
    std::vector<char> dest(in->partialsCount, 0);
    for (int k = 0; k < count; k++) dest[ops[(size_t)k * BEAGLE_OP_COUNT]] = 1;
    for (size_t k = 0; k < h.ops.size(); k += BEAGLE_OP_COUNT) if (!dest[h.ops[k]]) return false;
    std::vector<char> written(in->partialsCount, 0);
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * BEAGLE_OP_COUNT;
        if (h.writesBuf[op[5]]) return false;                                       // a sibling's post-order partial
        if (h.writesBuf[op[3]] && !written[op[3]]) return false;                    // a parent it has not produced itself by then
        written[op[0]] = 1;
    }
    return true;
}

// Try to hold `ops` back: 0 = held, 1 = not the shape of a gradient pass (the caller runs it now), < 0 = error.  The shape: no
// scale indices; the two operations below a node come as a pair that agrees on the two branch matrices; ONE node of the list
// has a parent the list does not produce (the root).
int holdPreList(Instance* in, const int* ops, int count) {
    Instance::HeldPreList& h = in->heldPre;
    const int nBuf = in->partialsCount;
    for (int k = 0; k < count; k++) if (ops[(size_t)k * BEAGLE_OP_COUNT + 1] != BEAGLE_OP_NONE || ops[(size_t)k * BEAGLE_OP_COUNT + 2] != BEAGLE_OP_NONE) return 1;
    std::vector<int> jobOfParent(nBuf, -1), jobOfDest(nBuf, -1);
    std::vector<Instance::HeldPreNode> nodes;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * BEAGLE_OP_COUNT;
        const int dest = op[0], par = op[3], mc = op[4], sib = op[5], ms = op[6];
        if (jobOfDest[dest] >= 0) return 1;
        int j = jobOfParent[par];
        if (j < 0) {
            j = (int)nodes.size(); jobOfParent[par] = j;
            Instance::HeldPreNode nd; nd.par = par; nd.preA = dest; nd.matA = mc; nd.postB = sib; nd.matB = ms; nd.preB = -1; nd.postA = -1;
            nd.jobA = nd.jobB = -1; nd.size = 1;
            nd.level = jobOfDest[par] >= 0 ? nodes[jobOfDest[par]].level + 1 : 0;      // (a parent's operation precedes its children's)
            nodes.push_back(nd);
        } else {
            Instance::HeldPreNode& nd = nodes[j];
            if (nd.preB >= 0 || mc != nd.matB || ms != nd.matA) return 1;
            nd.preB = dest; nd.postA = sib;
        }
        jobOfDest[dest] = j;
    }
    int root = -1, maxLevel = 0;
    for (size_t j = 0; j < nodes.size(); j++) {
        Instance::HeldPreNode& nd = nodes[j];
        if (nd.preB < 0) return 1;
        if (jobOfDest[nd.par] < 0) { if (root >= 0) return 1; root = (int)j; }
        nd.jobA = jobOfParent[nd.preA]; nd.jobB = jobOfParent[nd.preB];
        maxLevel = std::max(maxLevel, nd.level);
        for (int w = 0; w < 2; w++) {                              // what the kernels will dereference has to be there
            const int po = w ? nd.postB : nd.postA;
            if (!(in->tipStates[po] && po < in->tipCount) && !in->partials[po] && !isVirt(in, po)) return BEAGLE_ERROR_OUT_OF_RANGE;
            if (jobOfDest[po] >= 0) return 1;                      // a post-order operand that is one of the list's own destinations
        }
        if (!in->partials[nd.par] || !in->partials[nd.preA] || !in->partials[nd.preB]) return BEAGLE_ERROR_OUT_OF_RANGE;
    }
    if (root < 0) return 1;
    // (operations come parent first: a node's job index is larger than its parent's) subtree sizes, bottom-up
    for (size_t j = nodes.size(); j-- > 0;) {
        const Instance::HeldPreNode& nd = nodes[j];
        if (nd.jobA >= 0 && nd.jobA <= (int)j) return 1;
        if (nd.jobB >= 0 && nd.jobB <= (int)j) return 1;
        nodes[j].size = 1 + (nd.jobA >= 0 ? nodes[nd.jobA].size : 0) + (nd.jobB >= 0 ? nodes[nd.jobB].size : 0);
    }
    // The walk's order: depth first, the smaller subtree first, the other child's partial parked in a hold slot meanwhile.
    // Like the post-order walk (engine_walk.cpp walkChunkOps) it is cut into segments that run side by side — a wave executes
    // its program one dependent step after the other, and 20 000 patterns are only 1.2 waves per SIMD: the subtrees of at
    // most `chunk` nodes hanging off the upper part of the tree become segments of their own; the upper part stores the
    // pre-order partials of their roots and runs first.
    h.order.clear(); h.walkFlags.clear(); h.segStart.clear(); h.segRoot.clear(); h.holdSlots = 0;
    {
        const long groups = (in->P + 63) / 64;
        static const int forced = labEnv("BEAGLE_MI355_PRE_CHUNK") ? atoi(labEnv("BEAGLE_MI355_PRE_CHUNK")) : -1;
        int chunk = nodes.size() >= 64 ? (int)std::min<long>(256, std::max<long>(24, (long)nodes.size() * groups / 2560)) : 0;
        if (forced >= 0) chunk = forced;
        static const int forcedMin = labEnv("BEAGLE_MI355_PRE_MINHEAD") ? atoi(labEnv("BEAGLE_MI355_PRE_MINHEAD")) : -1;
        const int minHead = forcedMin >= 0 ? forcedMin : std::max(8, chunk / 4);
        std::vector<char> head(nodes.size(), 0);
        std::vector<int> heads(1, root);
        if (chunk > 0)
            for (size_t j = 0; j < nodes.size(); j++) {
                const int pj = jobOfDest[nodes[j].par];
                if (pj >= 0 && nodes[j].size <= chunk && nodes[j].size >= minHead && nodes[pj].size > chunk) { head[j] = 1; heads.push_back((int)j); }
            }
        for (int hd : heads) {
            h.segStart.push_back((int)h.order.size()); h.segRoot.push_back(hd);
            std::vector<std::pair<int, int>> parked;               // (job, hold slot)
            int j = hd; unsigned src = 0;
            while (j >= 0) {
                const Instance::HeldPreNode& nd = nodes[j];
                const bool storeA = nd.jobA >= 0 && head[nd.jobA], storeB = nd.jobB >= 0 && head[nd.jobB];
                const int a = storeA ? -1 : nd.jobA, b = storeB ? -1 : nd.jobB;      // children the walk steps into
                unsigned contA = storeA ? mi355::PW_CONT_STORE : 0u, contB = storeB ? mi355::PW_CONT_STORE : 0u;
                int next = -1; unsigned nextSrc = 0;
                if (a >= 0 && b >= 0) {
                    const bool aFirst = nodes[a].size <= nodes[b].size;
                    const int slot = (int)parked.size();
                    h.holdSlots = std::max(h.holdSlots, slot + 1);
                    (aFirst ? contA : contB) = 1u; (aFirst ? contB : contA) = 2u + (unsigned)slot;
                    parked.emplace_back(aFirst ? b : a, slot);
                    next = aFirst ? a : b;
                } else if (a >= 0) { contA = 1u; next = a; }
                else if (b >= 0) { contB = 1u; next = b; }
                else if (!parked.empty()) { next = parked.back().first; nextSrc = 1u + (unsigned)parked.back().second; parked.pop_back(); }
                h.order.push_back(j);
                h.walkFlags.push_back((src << mi355::PW_SRC_SHIFT) | (contA << mi355::PW_CONT_A_SHIFT) | (contB << mi355::PW_CONT_B_SHIFT));
                j = next; src = nextSrc;
            }
        }
        h.segStart.push_back((int)h.order.size());
        if (h.order.size() != nodes.size()) return 1;              // (cannot happen for a tree)
    }
    // its own copy of the root's pre-order partial: the caller rewrites that buffer before every list (simulateRoot,
    // AbstractBeagleGradientDelegate.java:142-151), which must not force this list to run
    if (!in->preRootCopy) { void* q = nullptr; int rc = devAlloc(in, &q, in->partialsBytes); if (rc) return rc; in->preRootCopy = (double*)q; }
    HIP_TRY(hipMemcpyAsync(in->preRootCopy, in->partials[nodes[root].par], in->partialsBytes, hipMemcpyDeviceToDevice, live(in)));
    h.rootBuf = nodes[root].par;
    h.ops.assign(ops, ops + (size_t)count * BEAGLE_OP_COUNT);
    h.nodes.swap(nodes);
    h.maxLevel = maxLevel;
    h.readsMatrix.assign(in->matrixCount, 0); h.readsBuf.assign(nBuf, 0); h.writesBuf.assign(nBuf, 0);
    for (const Instance::HeldPreNode& nd : h.nodes) {
        h.readsMatrix[nd.matA] = h.readsMatrix[nd.matB] = 1;
        h.readsBuf[nd.postA] = h.readsBuf[nd.postB] = 1;
        h.writesBuf[nd.preA] = h.writesBuf[nd.preB] = 1;
    }
    h.held = true;
    return 0;
}

// jobs of the held list for k_preNode4, level by level; start[l] .. start[l + 1] = level l.  edgeOf (nullable): buffer -> edge slot
static int heldJobs(Instance* in, const int* edgeOf, const int* dIdx, std::vector<mi355::PreNodeJob>& jobs, std::vector<int>& start) {
    const Instance::HeldPreList& h = in->heldPre;
    start.assign(h.maxLevel + 2, 0);
    for (const Instance::HeldPreNode& nd : h.nodes) start[nd.level + 1]++;
    for (int l = 0; l <= h.maxLevel; l++) start[l + 1] += start[l];
    std::vector<int> fill(start.begin(), start.end() - 1);
    jobs.resize(h.nodes.size());
    for (const Instance::HeldPreNode& nd : h.nodes) {
        mi355::PreNodeJob& jb = jobs[fill[nd.level]++];
        memset(&jb, 0, sizeof(jb));
        jb.preParent = nd.par == h.rootBuf ? in->preRootCopy : in->partials[nd.par];
        jb.preA = in->partials[nd.preA]; jb.preB = in->partials[nd.preB];
        for (int w = 0; w < 2; w++) {
            const int po = w ? nd.postB : nd.postA;
            const bool st = in->tipStates[po] && po < in->tipCount;
            const void* ptr = st ? (const void*)in->tipStates[po] : (const void*)in->partials[po];
            const int e = edgeOf ? edgeOf[w ? nd.preB : nd.preA] : -1;
            if (w) { jb.postB = ptr; jb.statesB = st; jb.slotB = e; jb.dB = e >= 0 ? dIdx[e] : 0; }
            else { jb.postA = ptr; jb.statesA = st; jb.slotA = e; jb.dA = e >= 0 ? dIdx[e] : 0; }
        }
        jb.matA = nd.matA; jb.matB = nd.matB;
    }
    return 0;
}
static int launchHeldLevels(Instance* in, const std::vector<mi355::PreNodeJob>& jobs, const std::vector<int>& start, int wIdx, double* dBlock) {
    const size_t maxChunk = (RING_BYTES / 4) / sizeof(mi355::PreNodeJob);
    for (size_t b = 0; b < jobs.size(); b += maxChunk) {
        const size_t n = std::min(maxChunk, jobs.size() - b);
        void* dJobs = nullptr;
        int rc = uploadTransient(in, &jobs[b], n * sizeof(mi355::PreNodeJob), &dJobs); if (rc) return rc;
        for (size_t l = 0; l + 1 < start.size(); l++) {
            const size_t lo = std::max((size_t)start[l], b), hi = std::min((size_t)start[l + 1], b + n);
            if (lo >= hi) continue;
            mi355::launchPreNodes4(live(in), (const mi355::PreNodeJob*)dJobs + (lo - b), (int)(hi - lo), in->matrices,
                                   in->weights + (size_t)wIdx * in->C, in->patternWeights, dBlock, in->P, in->C);
        }
    }
    return 0;
}

// the held list runs: every pre-order partial it defines is written (one sweep per tree level, k_preNode4)
int executeHeldPre(Instance* in) {
    if (!in->heldPre.held) return 0;
    { int rcm = materializeHeldOperands(in); if (rcm) return rcm; }
    std::vector<mi355::PreNodeJob> jobs; std::vector<int> start;
    heldJobs(in, nullptr, nullptr, jobs, start);
    in->heldPre.held = false;
    in->statLateLists++;
    return launchHeldLevels(in, jobs, start, 0, nullptr);
}

// which edge asks for which destination of the held list; 1 = the edges are not (all) edges of the held list
static int heldEdges(Instance* in, const int* postIdx, const int* preIdx, const int* dIdx, int count, std::vector<int>& edgeOf) {
    const Instance::HeldPreList& h = in->heldPre;
    edgeOf.assign(in->partialsCount, -1);
    for (int e = 0; e < count; e++) {
        if (badIndex(postIdx[e], in->partialsCount) || badIndex(preIdx[e], in->partialsCount) || badIndex(dIdx[e], in->matrixCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (edgeOf[preIdx[e]] >= 0 || !h.writesBuf[preIdx[e]]) return 1;
        edgeOf[preIdx[e]] = e;
    }
    for (const Instance::HeldPreNode& nd : h.nodes)
        for (int w = 0; w < 2; w++) {
            const int e = edgeOf[w ? nd.preB : nd.preA];
            if (e >= 0 && postIdx[e] != (w ? nd.postB : nd.postA)) return 1;
        }
    return 0;
}

// The held list and the edge derivatives asked for now as ONE sweep per tree level (k_preNode4): pre-order partials written,
// sums and sums of squares of the per-pattern derivatives.  The list has run afterwards.
static int fusedGradient(Instance* in, const std::vector<int>& edgeOf, const int* dIdx, int wIdx, int count, double* outSum, double* outSumSquared) {
    const int nb = mi355::edgeBlocks(in->P);
    if ((size_t)count * nb * 2 * sizeof(double) > ((size_t)512 << 20)) return 1;
    { int rcm = materializeHeldOperands(in); if (rcm) return rcm; }
    std::vector<mi355::PreNodeJob> jobs; std::vector<int> start;
    heldJobs(in, edgeOf.data(), dIdx, jobs, start);
    int rc = ensureEdgeScratch(in, (size_t)count * (nb + 1) * 2 * sizeof(double)); if (rc) return rc;
    double *dBlock = (double*)in->edgeScratch, *dSums = dBlock + (size_t)count * nb * 2;
    in->heldPre.held = false;                                      // (whatever happens below, the destinations are being written)
    rc = launchHeldLevels(in, jobs, start, wIdx, dBlock); if (rc) return rc;
    std::vector<double> sums((size_t)count * 2);
    mi355::launchEdgeFinal(live(in), dBlock, count, in->P, dSums);
    rc = download(in, sums.data(), dSums, sums.size() * sizeof(double)); if (rc) return rc;
    for (int e = 0; e < count; e++) {
        if (outSum) outSum[e] = sums[2 * e];
        if (outSumSquared) outSumSquared[e] = sums[2 * e + 1];
    }
    in->statFusedGradients++;
    return 0;
}

// The sums alone, nothing written (k_preWalk4): the list STAYS held.  1 = this list / instance cannot be walked.
static int walkedGradient(Instance* in, const std::vector<int>& edgeOf, const int* dIdx, int wIdx, int count, double* outSum) {
    const Instance::HeldPreList& h = in->heldPre;
    if (!in->preWalk || !in->walk || h.holdSlots > mi355::PW_MAX_HOLD || in->C > 16) return 1;     // (walk instances: reciprocal scale arrays, dummies)
    { int rcd = ensureWalkDummies(in); if (rcd) return rcd; }              // (the all-ones reciprocal array)
    const int waves = mi355::preWalkWaves(in->P, in->C);
    const size_t sumBytes = (size_t)(count + 1) * waves * sizeof(double), outBytes = (size_t)count * sizeof(double);
    if (sumBytes > ((size_t)1 << 30)) return 1;
    if ((size_t)(h.holdSlots + mi355::PW_POST_SLOTS) * in->C * 128 * 16 > 160 * 1024) return 1;
    // the list's root is what the likelihood is formed from, straight from its operands' partials: those two are real
    {
        const Instance::HeldPreNode& rt = h.nodes[h.segRoot[0]];
        std::vector<int> keys;
        if (isVirt(in, rt.postA)) in->planner.keysOf(rt.postA, keys);
        if (isVirt(in, rt.postB)) in->planner.keysOf(rt.postB, keys);
        if (!keys.empty()) { int rcm = materializeList(in, keys); if (rcm) return rcm; }
    }
    if (!in->preDummyStates) {
        void* q = nullptr; int rc = devAlloc(in, &q, ((size_t)in->P + 255) & ~(size_t)255); if (rc) return rc;
        in->preDummyStates = (uint8_t*)q;
        HIP_TRY(hipMemsetAsync(in->preDummyStates, 4, (size_t)in->P, live(in)));
    }
    // every segment: its descriptors, padded to an even count, and two more no-ops (the kernel's look-ahead)
    const int nSegs = (int)h.segRoot.size();
    std::vector<mi355::PreWalkSeg> segs(nSegs);
    std::vector<mi355::PreWalkOp> prog;
    prog.reserve(2 * h.order.size() + 3 * (size_t)nSegs);
    mi355::PreWalkOp nop;
    memset(&nop, 0, sizeof(nop));
    nop.postA = nop.postB = in->preRootCopy; nop.tipA = nop.tipB = in->preDummyStates; nop.slotA = nop.slotB = count;   // valid memory, the spare slot
    nop.dA = nop.dB = count;                                         // (the spare product: zeros)
    // per edge: its branch matrix and its differential matrix — the walk applies their product (kernels_preorder4.hip k_edgeProducts)
    std::vector<int> pairs(2 * (size_t)(count + 1), -1);
    nop.recipA = nop.recipB = in->onesScale;
    // 1 / (the factor a post-order operand was divided by), or ones; false: its scale buffer has been written since
    auto reciprocalOf = [&](int po, const double*& out) {
        out = in->onesScale;
        const int sIdx = in->scaleOfPartial[po];
        if (sIdx == -1) return true;
        if (sIdx < 0) return false;                                 // written before the instance kept track
        if (in->scaleVersion[sIdx] != in->scaleVersionAtWrite[po] || !in->scale[sIdx] || !in->scaleIsRaw[sIdx]) return false;
        out = in->scale[sIdx] + in->scaleStride;
        return true;
    };
    nop.flags = mi355::PW_TIP_A | mi355::PW_TIP_B;
    // a node over two compact tips is evaluated inside its parent's descriptor (kernels.h PW_CHERRY): per such child the definition's two
    // snapshot matrices (interleaved for the kernel by launchCherryPairs) and where its descriptor wants their address
    std::vector<int> cherryPairs;
    struct CherryRef { size_t desc; int which; };
    std::vector<CherryRef> cherryRefs;
    // an unstored operand `po` (walkableDefinition) as descriptors in front of the node that reads it: its value ends up in post
    // slot `result` (the inner node of the two-step kind passes through slot 2)
    bool anyPost = false;
    auto stepReciprocal = [&](const mi355::VirtStep& st, const double*& out) {
        out = in->onesScale;
        if (st.scaleIdx == mi355::PLAN_NONE) return true;
        if (!in->scale[st.scaleIdx] || !in->scaleIsRaw[st.scaleIdx]) return false;
        out = in->scale[st.scaleIdx] + in->scaleStride;
        return true;
    };
    auto emitPost = [&](int po, unsigned result) {
        const int key = in->planner.key(po, 0);
        const mi355::VirtDef& d = in->planner.definition(key);
        const mi355::VirtStep& s0 = d.steps[0];
        mi355::PreWalkOp c = nop;
        c.flags = mi355::PW_TIP_A | mi355::PW_TIP_B | mi355::PW_POSTOP | ((d.nSteps == 2 ? 2u : result) << mi355::PW_DST_SHIFT);
        c.tipA = in->tipStates[s0.tipA]; c.tipB = in->tipStates[s0.tipB];
        c.matA = in->planner.snapSlot(key, 0, 0); c.matB = in->planner.snapSlot(key, 0, 1);
        if (!stepReciprocal(s0, c.recipA)) return false;
        prog.push_back(c);
        if (d.nSteps == 2) {
            const mi355::VirtStep& s1 = d.steps[1];
            mi355::PreWalkOp x = nop;
            x.flags = mi355::PW_TIP_A | mi355::PW_TIP_B | mi355::PW_POSTOP | mi355::PW_SLOT_A | (2u << mi355::PW_SLOTA_SHIFT) | (result << mi355::PW_DST_SHIFT);
            x.tipB = in->tipStates[s1.tipB];
            x.matA = in->planner.snapSlot(key, 1, 0); x.matB = in->planner.snapSlot(key, 1, 1);
            if (!stepReciprocal(s1, x.recipA)) return false;
            prog.push_back(x);
        }
        anyPost = true;
        return true;
    };
    for (int sgi = 0; sgi < nSegs; sgi++) {
        const int first = h.segStart[sgi], n = h.segStart[sgi + 1] - first;
        segs[sgi].progStart = (int)prog.size();
        segs[sgi].rootPre = sgi == 0 ? in->preRootCopy : in->partials[h.nodes[h.segRoot[sgi]].par];
        for (int k = first; k < first + n; k++) {
            const Instance::HeldPreNode& nd = h.nodes[h.order[k]];
            mi355::PreWalkOp op = nop;
            op.flags = h.walkFlags[k];
            for (int w = 0; w < 2; w++) {
                const int po = w ? nd.postB : nd.postA;
                const bool st = in->tipStates[po] && po < in->tipCount;
                const int e = edgeOf[w ? nd.preB : nd.preA];
                if (!st && isVirt(in, po)) {
                    if (!walkableDefinition(in, po)) return 1;
                    const mi355::VirtDef& def = in->planner.definition(in->planner.key(po, 0));
                    const unsigned cont = (op.flags >> (w ? mi355::PW_CONT_B_SHIFT : mi355::PW_CONT_A_SHIFT)) & 15u;
                    if (def.nSteps == 1 && cont != mi355::PW_CONT_STORE) {
                        // a node over two tips: evaluated inside this descriptor from its tips' states and its two matrices
                        const mi355::VirtStep& s0 = def.steps[0];
                        const double* rc = nullptr;
                        if (!stepReciprocal(s0, rc)) return 1;
                        const int key = in->planner.key(po, 0);
                        cherryRefs.push_back(CherryRef{(size_t)-1, w});          // (its descriptor's index: filled in when the descriptor is pushed)
                        cherryPairs.push_back(in->planner.snapSlot(key, 0, 0)); cherryPairs.push_back(in->planner.snapSlot(key, 0, 1));
                        if (w) { op.flags |= mi355::PW_CHERRY_B; op.tipB = in->tipStates[s0.tipA]; op.storeB = (double*)in->tipStates[s0.tipB]; op.recipB = rc; if (e >= 0) { op.slotB = e; op.dB = e; } }
                        else { op.flags |= mi355::PW_CHERRY_A; op.tipA = in->tipStates[s0.tipA]; op.storeA = (double*)in->tipStates[s0.tipB]; op.recipA = rc; if (e >= 0) { op.slotA = e; op.dA = e; } }
                        if (e >= 0) { pairs[2 * (size_t)e] = w ? nd.matB : nd.matA; pairs[2 * (size_t)e + 1] = dIdx[e]; }
                        continue;
                    }
                    // a longer one: re-evaluated in front of this descriptor, read from its post slot (no load but a dummy state byte)
                    if (!emitPost(po, (unsigned)w)) return 1;
                    const double* rc = nullptr;
                    if (!reciprocalOf(po, rc)) return 1;
                    if (w) { op.flags |= mi355::PW_TIP_B | mi355::PW_SLOT_B | (1u << mi355::PW_SLOTB_SHIFT); op.recipB = rc; if (e >= 0) { op.slotB = e; op.dB = e; } }
                    else { op.flags |= mi355::PW_TIP_A | mi355::PW_SLOT_A | (0u << mi355::PW_SLOTA_SHIFT); op.recipA = rc; if (e >= 0) { op.slotA = e; op.dA = e; } }
                    if (e >= 0) { pairs[2 * (size_t)e] = w ? nd.matB : nd.matA; pairs[2 * (size_t)e + 1] = dIdx[e]; }
                    continue;
                }
                if (w) { if (st) { op.tipB = in->tipStates[po]; op.flags |= mi355::PW_TIP_B; } else { op.postB = in->partials[po]; if (!reciprocalOf(po, op.recipB)) return 1; } if (e >= 0) { op.slotB = e; op.dB = e; } }
                else { if (st) { op.tipA = in->tipStates[po]; op.flags |= mi355::PW_TIP_A; } else { op.postA = in->partials[po]; if (!reciprocalOf(po, op.recipA)) return 1; } if (e >= 0) { op.slotA = e; op.dA = e; } }
                if (e >= 0) { pairs[2 * (size_t)e] = w ? nd.matB : nd.matA; pairs[2 * (size_t)e + 1] = dIdx[e]; }
            }
            if (!(op.flags & mi355::PW_CHERRY_A)) op.storeA = in->partials[nd.preA];
            if (!(op.flags & mi355::PW_CHERRY_B)) op.storeB = in->partials[nd.preB];
            op.matA = nd.matA; op.matB = nd.matB;
            prog.push_back(op);
            for (size_t q = cherryRefs.size(); q-- > 0 && cherryRefs[q].desc == (size_t)-1;) cherryRefs[q].desc = prog.size() - 1;
        }
        const int emitted = (int)prog.size() - segs[sgi].progStart;
        segs[sgi].progCount = (emitted + 1) & ~1;
        for (int k = emitted; k < segs[sgi].progCount + 2; k++) prog.push_back(nop);
    }
    for (mi355::PreWalkOp& d : prog) d.flags = (d.flags & ~(15u << mi355::PW_LOADS_SHIFT)) | (mi355::preWalkLoads(d.flags) << mi355::PW_LOADS_SHIFT);
    const size_t segBytes = ((size_t)nSegs * sizeof(mi355::PreWalkSeg) + 255) & ~(size_t)255;
    const size_t progBytes = prog.size() * sizeof(mi355::PreWalkOp);
    if (progBytes + segBytes > RING_BYTES / 2) return 1;
    if (progBytes + segBytes > in->dPreProgBytes) {
        HIP_TRY(hipStreamSynchronize(live(in)));
        void* q = nullptr; int rc = devAlloc(in, &q, (progBytes + segBytes) * 2); if (rc) return rc;   // (the old, smaller one stays allocated until the instance goes)
        in->dPreProg = q; in->dPreProgBytes = (progBytes + segBytes) * 2;
    }
    const size_t outPad = (outBytes + 255) & ~(size_t)255, prodBytes = (size_t)(count + 1) * in->C * 16 * sizeof(double), pairBytes = pairs.size() * sizeof(int);
    const size_t pairPad = (pairBytes + 255) & ~(size_t)255, nCherries = cherryPairs.size() / 2;
    const size_t cherryBytes = nCherries * in->C * 32 * sizeof(double), cherryPairBytes = cherryPairs.size() * sizeof(int);
    const size_t sumPad = (sumBytes + 255) & ~(size_t)255;              // (what follows the sums keeps a 256-byte alignment: 16-byte loads)
    int rc = ensureEdgeScratch(in, sumPad + outPad + prodBytes + pairPad + cherryBytes + cherryPairBytes); if (rc) return rc;
    double *dSums = (double*)in->edgeScratch, *dOut = (double*)((char*)in->edgeScratch + sumPad), *dProducts = (double*)((char*)in->edgeScratch + sumPad + outPad);
    int* dPairs = (int*)((char*)dProducts + prodBytes);
    double* dCherries = (double*)((char*)dPairs + pairPad);
    int* dCherryPairs = (int*)((char*)dCherries + cherryBytes);
    rc = upload(in, dPairs, pairs.data(), pairBytes); if (rc) return rc;
    mi355::launchEdgeProducts(live(in), in->matrices, dPairs, dProducts, in->C, count + 1);
    if (nCherries) {
        rc = upload(in, dCherryPairs, cherryPairs.data(), cherryPairBytes); if (rc) return rc;
        mi355::launchCherryPairs(live(in), in->matrices, dCherryPairs, dCherries, in->C, (int)nCherries);
        for (size_t k = 0; k < nCherries; k++) {          // (the scratch may have moved: the addresses go in now, the program is uploaded below)
            mi355::PreWalkOp& d = prog[cherryRefs[k].desc];
            (cherryRefs[k].which ? d.postB : d.postA) = dCherries + k * (size_t)in->C * 32;
        }
    }
    rc = upload(in, in->dPreProg, segs.data(), (size_t)nSegs * sizeof(mi355::PreWalkSeg)); if (rc) return rc;
    rc = upload(in, (char*)in->dPreProg + segBytes, prog.data(), progBytes); if (rc) return rc;
    if (!mi355::launchPreWalk4(live(in), (const mi355::PreWalkOp*)((char*)in->dPreProg + segBytes), (const mi355::PreWalkSeg*)in->dPreProg, nSegs,
                               in->preRootCopy, in->matrices, dProducts, in->weights + (size_t)wIdx * in->C, in->patternWeights, dSums, in->P, in->C, h.holdSlots, anyPost)) return 1;
    if (in->hostTrace) fprintf(stderr, "[mi355] pre-order walk: %d segments, %zu descriptors, %d hold slots (%zu KB of LDS per workgroup)\n", nSegs, prog.size(), h.holdSlots,
                               (size_t)(h.holdSlots + (anyPost ? mi355::PW_POST_SLOTS : 0)) * in->C * 2);
    mi355::launchPreWalkFinal(live(in), dSums, count, in->P, in->C, dOut);
    std::vector<double> out(count);
    rc = download(in, out.data(), dOut, outBytes); if (rc) return rc;
    // (scaled partials times reciprocals of factors: should a product have left the range on a very deep path, the sums show it —
    // the caller then gets them from the path that forms every edge's denominator itself)
    for (int e = 0; e < count; e++) if (!std::isfinite(out[e])) return 1;
    if (outSum) memcpy(outSum, out.data(), outBytes);
    in->statWalkedGradients++;
    return 0;
}

int edgeDifferentials(Instance* in, const int* postIdx, const int* preIdx, const int* dIdx, int wIdx, int count,
                      double* outDerivatives, double* outSum, double* outSumSquared) {
    if (count <= 0) return 0;
    if (in->heldPre.held) {
        // a held-back pre-order list (the usual case: these are its edges).  Sums only: no pre-order partial is written and the
        // list stays held; sums of squares as well: the list runs together with the derivatives; anything else: it runs first
        std::vector<int> edgeOf;
        int rc = outDerivatives ? 1 : heldEdges(in, postIdx, preIdx, dIdx, count, edgeOf);
        if (rc < 0) return rc;
        if (rc == 0 && !outSumSquared) { rc = walkedGradient(in, edgeOf, dIdx, wIdx, count, outSum); if (rc <= 0) return rc; rc = 0; }
        if (rc == 0) { rc = fusedGradient(in, edgeOf, dIdx, wIdx, count, outSum, outSumSquared); if (rc <= 0) return rc; }
        rc = executeHeldPre(in); if (rc) return rc;
    }
    std::vector<int> need;
    for (int e = 0; e < count; e++) {
        if (badIndex(postIdx[e], in->partialsCount) || badIndex(preIdx[e], in->partialsCount) || badIndex(dIdx[e], in->matrixCount))
            return BEAGLE_ERROR_OUT_OF_RANGE;
        if (isVirt(in, postIdx[e])) in->planner.keysOf(postIdx[e], need);
        if (isVirt(in, preIdx[e])) in->planner.keysOf(preIdx[e], need);
    }
    if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    const int nb = mi355::edgeBlocks(in->P);
    const size_t perEdgeBytes = (size_t)nb * 2 * sizeof(double) + 2 * sizeof(double) + (outDerivatives ? (size_t)in->P * sizeof(double) : 0);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)256 << 20) / perEdgeBytes));
    chunk = std::min(chunk, 32768);
    int rc = ensureEdgeScratch(in, (size_t)chunk * (nb + 1) * 2 * sizeof(double)); if (rc) return rc;
    double *dBlock = (double*)in->edgeScratch, *dSums = dBlock + (size_t)chunk * nb * 2, *dPer = nullptr;
    if (outDerivatives && hipMalloc((void**)&dPer, (size_t)chunk * in->P * sizeof(double)) != hipSuccess) rc = BEAGLE_ERROR_OUT_OF_MEMORY;
    std::vector<mi355::EdgeDesc> descs;
    std::vector<double> sums;
    static const bool preNaive = labEnv("BEAGLE_MI355_PRE_NAIVE") && atoi(labEnv("BEAGLE_MI355_PRE_NAIVE")) != 0;
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
        // (round 4: kernels_mfma.hip k_edgeTiled takes every edge of such an instance in one pass — pre and post read once;
        // BEAGLE_MI355_EDGE_TWO_STEP=1 keeps the older route, =2 sends only the tip edges to the direct kernel.)
        static const int edgeTwoStep = getenv("BEAGLE_MI355_EDGE_TWO_STEP") ? atoi(getenv("BEAGLE_MI355_EDGE_TWO_STEP")) : 0;
        std::vector<mi355::EdgeDesc> direct, viaPrune, onePass;
        for (int e = 0; e < m; e++) {
            descs[e].slot = e;
            if (twoStep && edgeTwoStep != 1 && !(edgeTwoStep == 2 && descs[e].postIsStates)) onePass.push_back(descs[e]);
            else (twoStep && !descs[e].postIsStates ? viaPrune : direct).push_back(descs[e]);
        }
        if (!onePass.empty()) {
            void* dDesc = nullptr;
            rc = uploadTransient(in, onePass.data(), onePass.size() * sizeof(mi355::EdgeDesc), &dDesc); if (rc) break;
            if (!mi355::launchEdgeTiled(live(in), (const mi355::EdgeDesc*)dDesc, (int)onePass.size(), in->matrices,
                                        in->weights + (size_t)wIdx * in->C, in->patternWeights, dPer, dBlock, in->P, in->S, in->C)) { rc = BEAGLE_ERROR_GENERAL; break; }
        }
        if (!direct.empty()) {
            void* dDesc = nullptr;
            rc = uploadTransient(in, direct.data(), direct.size() * sizeof(mi355::EdgeDesc), &dDesc); if (rc) break;
            if (in->S == 4 && !in->tiled)
                mi355::launchEdgeDifferentials4(live(in), (const mi355::EdgeDesc*)dDesc, (int)direct.size(), in->matrices,
                                                in->weights + (size_t)wIdx * in->C, in->patternWeights, dPer, dBlock, in->P, in->C);
            else
                mi355::launchEdgeDifferentials(live(in), (const mi355::EdgeDesc*)dDesc, (int)direct.size(), in->matrices,
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
            mi355::launchPruneLevelTiled(live(in), (const OpDesc*)dPass, n, in->matrices, in->P, in->S, in->C, false);
            mi355::launchEdgeReduce(live(in), (const mi355::EdgeDesc*)dDesc, n, in->weights + (size_t)wIdx * in->C, in->patternWeights,
                                    dPer, dBlock, in->P, in->S, in->C, in->tiled);
        }
        if (rc) break;
        mi355::launchEdgeFinal(live(in), dBlock, m, in->P, dSums);
        sums.resize((size_t)m * 2);
        rc = download(in, sums.data(), dSums, sums.size() * sizeof(double)); if (rc) break;
        for (int e = 0; e < m; e++) {
            if (outSum) outSum[b + e] = sums[2 * e];
            if (outSumSquared) outSumSquared[b + e] = sums[2 * e + 1];
        }
        if (outDerivatives) rc = download(in, outDerivatives + (size_t)b * in->P, dPer, (size_t)m * in->P * sizeof(double));
    }
    if (dPer) { hipStreamSynchronize(live(in)); hipFree(dPer); }
    return rc;
}

// calculateCrossProductDifferentials (semantics: include/beagle_mi355.h)
int crossProducts(Instance* in, const int* postIdx, const int* preIdx, int rateIdx, int wIdx, const double* lengths, int count, double* outSum) {
    if (count <= 0) return 0;
    std::vector<int> need;
    for (int e = 0; e < count; e++) {
        if (badIndex(postIdx[e], in->partialsCount) || badIndex(preIdx[e], in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
        if (isVirt(in, postIdx[e])) in->planner.keysOf(postIdx[e], need);
        if (isVirt(in, preIdx[e])) in->planner.keysOf(preIdx[e], need);
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
        mi355::launchCrossProducts(live(in), (const mi355::EdgeDesc*)dDesc, (int)n, (const double*)dLen, in->weights + (size_t)wIdx * in->C,
                                   in->rates + (size_t)rateIdx * in->C, in->patternWeights, dPartial, dOut, in->P, in->S, in->C, in->tiled);
        rc = download(in, sums.data(), dOut, nOut * sizeof(double)); if (rc) break;
        for (size_t k = 0; k < nOut; k++) outSum[k] += sums[k];
    }
    hipStreamSynchronize(live(in));
    hipFree(dPartial); hipFree(dOut);
    return rc;
}


}  // namespace eng
}  // namespace mi355
