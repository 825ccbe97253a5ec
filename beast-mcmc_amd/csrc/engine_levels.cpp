// engine_levels.cpp — every state count but 4: an operation list levelised and enqueued one dependency level per launch
// (kernels_mfma.hip for 16..64 states with virtual cherries, kernels.hip k_pruneGeneral otherwise).  See engine_internal.h.
#include "engine_internal.h"

using mi355::OpDesc;

namespace mi355 {
namespace eng {

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
    mi355::launchPruneLevelTiled(live(in), (const OpDesc*)dOps, (int)descs.size(), in->matrices, in->P, in->S, in->C, false);
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
            d.scaleWrite = in->scale[wS]; in->scaleIsRaw[wS] = 1; scalesWritten(in);
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
    // Depth-first launches (BEAGLE_MI355_SCHED=dfs[:K]; an experiment of round 4, profiles/r04_experiments.txt): a level-by-level
    // sweep writes a whole level — gigabytes — before anything reads it back, so every child partial comes from HBM.  In
    // depth-first order (the larger subtree of a node first, so that the smaller one and the node itself follow each other
    // closely) a parent runs within a few launches of its children: what it reads was written megabytes ago and can still sit in
    // the 256 MB memory-side cache.  Launches: greedily packed runs of at most K consecutive operations of that order that do
    // not depend on each other.
    if (in->schedDfs > 0 && !warSeen) {
        std::vector<int> size(count, 1), order, launchOf(count, -1);
        for (int k = 0; k < count; k++)                     // (producers precede consumers in the list)
            if (!skipped[k]) for (int e = predOff[k]; e < predOff[k + 1]; e++) size[k] += size[predList[e]];
        std::vector<char> consumed(count, 0), done(count, 0);
        for (int k = 0; k < count; k++) for (int e = predOff[k]; e < predOff[k + 1]; e++) consumed[predList[e]] = 1;
        std::vector<std::pair<int, int>> st;                // (op, next predecessor to visit), predecessors by descending subtree size
        auto preds = [&](int k) {
            std::vector<int> p(predList.begin() + predOff[k], predList.begin() + predOff[k + 1]);
            std::sort(p.begin(), p.end(), [&](int a, int b) { return size[a] > size[b]; });
            p.erase(std::unique(p.begin(), p.end()), p.end());
            return p;
        };
        for (int r = 0; r < count; r++) {
            if (skipped[r] || consumed[r] || done[r]) continue;
            st.emplace_back(r, 0);
            while (!st.empty()) {
                const int k = st.back().first;
                const std::vector<int> p = preds(k);
                int& next = st.back().second;
                while (next < (int)p.size() && (done[p[next]] || skipped[p[next]])) next++;
                if (next < (int)p.size()) { const int c = p[next++]; st.emplace_back(c, 0); continue; }
                if (!done[k]) { done[k] = 1; order.push_back(k); }
                st.pop_back();
            }
        }
        int cur = 0, inCur = 0;
        for (int k : order) {
            bool dep = inCur >= in->schedDfs;
            for (int e = predOff[k]; e < predOff[k + 1] && !dep; e++) dep = launchOf[predList[e]] == cur;
            if (dep) { cur++; inCur = 0; }
            launchOf[k] = cur; inCur++;
        }
        if ((int)order.size() == count - (int)std::count(skipped.begin(), skipped.end(), (char)1)) {
            for (int k = 0; k < count; k++) if (!skipped[k]) level[k] = launchOf[k];
            maxLevel = cur;
        }
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
    const double* dCherryTables = nullptr;
    if (!snapPairs.empty()) {
        void* dPairs = nullptr;
        int rc = uploadTransient(in, snapPairs.data(), snapPairs.size() * sizeof(int), &dPairs); if (rc) return rc;
        mi355::launchSnapshotMatrices(live(in), in->matrices, (const int*)dPairs, (int)(snapPairs.size() / 2), in->C * in->S * in->S);
    }
    if (!cherries.empty()) {
        void* dC = nullptr;
        int rc = uploadTransient(in, cherries.data(), cherries.size() * sizeof(mi355::CherryDesc), &dC); if (rc) return rc;
        dCherries = (const mi355::CherryDesc*)dC;
        if (in->S > 20) {                                // 21..64 states: the cherries' matrices as column tables in global memory
            const size_t bytes = mi355::cherryTableBytes((int)cherries.size(), in->S, in->C);
            if (bytes > in->cherryTableBytes) {
                HIP_TRY(hipStreamSynchronize(live(in)));
                if (in->cherryTables) {
                    for (auto& a : in->allocations) if (a == (void*)in->cherryTables) { a = in->allocations.back(); in->allocations.pop_back(); break; }
                    hipFree(in->cherryTables); in->deviceBytes -= in->cherryTableBytes; in->cherryTables = nullptr; in->cherryTableBytes = 0;
                }
                void* q = nullptr; rc = devAlloc(in, &q, bytes + bytes / 4); if (rc) return rc;
                in->cherryTables = (double*)q; in->cherryTableBytes = bytes + bytes / 4;
            }
            mi355::launchCherryTables(live(in), dCherries, (int)cherries.size(), in->matrices, in->S, in->C, in->cherryTables);
            dCherryTables = in->cherryTables;
        }
    }
    // ONE descriptor upload for the whole list (every extra copy is a dependent blit kernel between two
    // level launches: ~4 us + two boundaries), chunked only when the list would not fit the ring; then one
    // launch per dependency level reading its slice.  With the kernel timer on, ONE HIP-event pair brackets
    // all level launches of the call (gaps between levels included — they are part of what the path costs).
    const size_t maxChunkOps = (RING_BYTES / 4) / sizeof(OpDesc);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timeThisCall(in)) {
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
        if (e0 && chunkBegin == 0) HIP_TRY(hipEventRecord(e0, live(in)));
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
                mi355::launchPruneLevelTiled(live(in), (const OpDesc*)dChunk + (begin - chunkBegin), end - begin, in->matrices,
                                             in->P, in->S, in->C, anyWrite, dCherries, dCherryTables);
            else
                mi355::launchPruneLevel(live(in), (const OpDesc*)dChunk + (begin - chunkBegin), end - begin, in->matrices,
                                        in->P, in->S, in->C, maxRange);
            launches++;
        }
        chunkBegin = chunkEnd;
    }
    if (e1 && launches > 0) { HIP_TRY(hipEventRecord(e1, live(in))); in->pendingLaunches += launches; }
    else if (e1) { in->eventsUsed--; in->timedCalls--; }       // nothing was launched: give the (unrecorded) event pair back
    HIP_TRY(hipGetLastError());
    return foldCumulative(in, ops, count, tuple, globalCum);
}

// cumulative scale factors requested together with an update: fold the factors the list wrote into the cumulative
// buffer afterwards, in op order (deterministic; no cross-workgroup atomics)
int foldCumulative(Instance* in, const int* ops, int count, int tuple, int globalCum) {
    // one accumulation launch per (cumulative buffer, partition) met in the list — the factors of all its operations at once (the
    // kernel walks the list of sources in op order: deterministic) — not one per operation (999 launches for a 1000-taxon tree)
    struct Group { int cum, part; std::vector<const double*> srcs; };
    std::vector<Group> groups;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        const int wS = op[1];
        int part = 0, cum = globalCum;
        if (tuple == BEAGLE_PARTITION_OP_COUNT) { part = op[7]; cum = op[8]; }
        if (cum == BEAGLE_OP_NONE || wS == BEAGLE_OP_NONE) continue;
        Group* g = nullptr;
        for (Group& x : groups) if (x.cum == cum && x.part == part) { g = &x; break; }
        if (!g) { groups.push_back(Group{cum, part, {}}); g = &groups.back(); }
        g->srcs.push_back(in->scale[wS]);
    }
    for (const Group& g : groups) {
        int rc = materializeScaleUsers(in, g.cum); if (rc) return rc;
        rc = ensureScale(in, g.cum); if (rc) return rc;
        const int chunk = 4096;
        for (size_t b = 0; b < g.srcs.size(); b += chunk) {
            const int n = (int)std::min<size_t>(chunk, g.srcs.size() - b);
            std::vector<int> raw(n, 1);
            void *dSrc = nullptr, *dRaw = nullptr;
            rc = uploadTransient(in, &g.srcs[b], (size_t)n * sizeof(double*), &dSrc); if (rc) return rc;
            rc = uploadTransient(in, raw.data(), (size_t)n * sizeof(int), &dRaw); if (rc) return rc;
            mi355::launchAccumulateScale(live(in), in->scale[g.cum], (const double* const*)dSrc, (const int*)dRaw, n, 1.0,
                                         in->partStart[g.part], in->partEnd[g.part]);
        }
    }
    return 0;
}

int runOperations(Instance* in, const int* ops, int count, int tuple, int globalCum) {
    if (in->walk) return runOperationsWalk(in, ops, count, tuple, globalCum);
    if (in->walkT) {
        // a list that rescales in write mode: the walk's write-mode form (kernels_mfma.hip k_walkT32W1) up to four categories; beyond
        // that (a pattern's factor needs all its categories in one workgroup) level by level, on operands that exist in memory
        bool writes = false;
        for (int k = 0; k < count && !writes; k++) writes = ops[(size_t)k * tuple + 1] != BEAGLE_OP_NONE;
        if (!writes || in->walkTWrite) return runOperationsWalk(in, ops, count, tuple, globalCum);
        std::vector<int> need;
        for (int k = 0; k < count; k++) {
            const int* op = ops + (size_t)k * tuple;
            if (badIndex(op[0], in->partialsCount) || badIndex(op[3], in->partialsCount) || badIndex(op[5], in->partialsCount)) return BEAGLE_ERROR_OUT_OF_RANGE;
            if (isVirt(in, op[3])) in->planner.keysOf(op[3], need);
            if (isVirt(in, op[5])) in->planner.keysOf(op[5], need);
            // (the level path forgets a destination's definitions for ALL partitions; what another partition still defines there
            // has to exist before that)
            if (in->partitionCount > 1 && isVirt(in, op[0])) in->planner.keysOf(op[0], need);
        }
        if (!need.empty()) { int rc = materializeList(in, need); if (rc) return rc; }
    }
    return runOperationsLevels(in, ops, count, tuple, globalCum);
}


}  // namespace eng
}  // namespace mi355
