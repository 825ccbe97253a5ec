// engine_walk.cpp — 4 states: a planned program (planner.h) resolved to device addresses and run as pattern-walk launches
// (kernels_walk4.hip); materialisation of virtual buffers; updatePartials' steady-state fast path.  See engine_internal.h.
#include "engine_internal.h"
#include <atomic>
#include <unordered_map>

using mi355::OpDesc;

namespace mi355 {
namespace eng {

// ---- folded reciprocal vectors (Instance::folds) ------------------------------------------------------------------------
constexpr int FOLD_MAX_MEMBERS = 32;        // factors per fold: a longer run of unstored nodes is folded in pieces
// largest product of reciprocals a fold may hold (each is >= 1).  An unstored intermediate travels through registers and hold slots
// UNSCALED by the members gathered so far, i.e. as small as 1 / product of its per-node-scaled size: with 1e100 its entries stay
// normal numbers down to 1e-208 of the node's largest entry (per-node read mode keeps them down to 1e-308); beyond, the plan falls
// back to per-node factors (runPlan: noFoldTag)
constexpr double FOLD_SAFE_MAX = 1e100;

// forget every fold (their vectors are kept for reuse) and every resolved program that may point at one
static void dropFolds(Instance* in, bool keepVectors) {
    for (Instance::FoldVec& f : in->folds) if (keepVectors && f.recip) in->foldFree.push_back(f.recip);
    if (!keepVectors) in->foldFree.clear();                // (another layout: the old vectors stay with the instance until it is destroyed)
    in->folds.clear(); in->foldIndex.clear();
    for (Instance::Resolved& r : in->resolved) { r.tag = 0; r.dProgValid = false; r.folds.clear(); r.foldEpoch = -1; }
}
void forgetFolds(Instance* in) { dropFolds(in, false); scalesWritten(in); }

// the fold of exactly these scale buffers, in this order (index into Instance::folds; < 0: out of memory)
static int foldFor(Instance* in, const std::vector<int>& members) {
    size_t h = 1469598103934665603ull;
    for (int m : members) { h ^= (size_t)(unsigned)m; h *= 1099511628211ull; }
    auto it = std::lower_bound(in->foldIndex.begin(), in->foldIndex.end(), std::make_pair(h, -1));
    for (auto q = it; q != in->foldIndex.end() && q->first == h; ++q)
        if (in->folds[(size_t)q->second].members == members) return q->second;
    Instance::FoldVec f;
    f.members = members;
    if (!in->foldFree.empty()) { f.recip = in->foldFree.back(); in->foldFree.pop_back(); }
    else {
        if (devAlloc(in, (void**)&f.recip, in->scaleStride * sizeof(double))) return -1;
        // (the T32 walk reads whole tiles: what lies past the last pattern must be a number it can divide by)
        if (in->walkT) mi355::launchFill(live(in), f.recip, 1.0, 0, (int)in->scaleStride);
    }
    in->folds.push_back(f);
    const int id = (int)in->folds.size() - 1;
    in->foldIndex.insert(it, std::make_pair(h, id));
    return id;
}

// (re)build the folds of `ids` that are older than the last scale-buffer write; *anyBad: one of `ids` leaves the safe range.
// One launch for all of them, then ONE read-back of the range check — which stalls the stream once per write-mode evaluation
// (DYNAMIC rescaling: every 100th), not per evaluation.
static int refreshFolds(Instance* in, const std::vector<int>& ids, bool* anyBad) {
    std::vector<int> stale;
    for (int id : ids) {
        Instance::FoldVec& f = in->folds[(size_t)id];
        if (f.builtEpoch != in->scaleWriteEpoch) { f.builtEpoch = in->scaleWriteEpoch; stale.push_back(id); }
    }
    if (!stale.empty()) {
        std::vector<const double*> srcs; std::vector<int> start; std::vector<double*> dst;
        for (int id : stale) {
            const Instance::FoldVec& f = in->folds[(size_t)id];
            start.push_back((int)srcs.size());
            for (int m : f.members) srcs.push_back(in->scale[m] + (in->walkT ? 0 : in->scaleStride));    // (reciprocal halves; T32: the factors)
            dst.push_back(f.recip);
        }
        start.push_back((int)srcs.size());
        if (in->foldWorstCount < stale.size()) {
            const size_t want = std::max<size_t>(stale.size() + stale.size() / 2, 256);
            int rc = devAlloc(in, (void**)&in->foldWorst, want * sizeof(unsigned long long)); if (rc) return rc;
            in->foldWorstCount = want;
        }
        HIP_TRY(hipMemsetAsync(in->foldWorst, 0, stale.size() * sizeof(unsigned long long), live(in)));
        const int chunk = 2048;                             // (jobs per launch: their pointer lists go through the staging ring)
        for (size_t b = 0; b < stale.size(); b += chunk) {
            const size_t e = std::min(stale.size(), b + chunk);
            std::vector<int> st(start.begin() + b, start.begin() + e + 1);
            const int base = st[0];
            for (int& v : st) v -= base;
            void *dSrcs = nullptr, *dStart = nullptr, *dDst = nullptr;
            int rc = uploadTransient(in, srcs.data() + base, (size_t)st.back() * sizeof(double*), &dSrcs); if (rc) return rc;
            rc = uploadTransient(in, st.data(), st.size() * sizeof(int), &dStart); if (rc) return rc;
            rc = uploadTransient(in, dst.data() + b, (e - b) * sizeof(double*), &dDst); if (rc) return rc;
            mi355::launchFoldReciprocals(live(in), (const double* const*)dSrcs, (const int*)dStart, (double* const*)dDst, (int)(e - b),
                                         in->walkT ? in->P : (int)in->pairLen, in->foldWorst + b, in->walkT);
        }
        HIP_TRY(hipGetLastError());
        std::vector<unsigned long long> worst(stale.size());
        int rc = download(in, worst.data(), in->foldWorst, worst.size() * sizeof(unsigned long long)); if (rc) return rc;
        for (size_t k = 0; k < stale.size(); k++) {
            double v; memcpy(&v, &worst[k], sizeof(v));
            in->folds[(size_t)stale[k]].bad = !(v <= FOLD_SAFE_MAX);
        }
        in->statFoldBuilds += (long)stale.size();
    }
    *anyBad = false;
    for (int id : ids) if (in->folds[(size_t)id].bad) *anyBad = true;
    return 0;
}

// did this run of a plan write per-node scale buffers?  (folds built from them are stale then)
static inline bool anyScaleWriteIn(const Instance* in, const Instance::Resolved* slot, bool reuse, long writesAtEntry) {
    return reuse ? slot->scaleWrites > 0 : in->statScaleWrites != writesAtEntry;
}

// Resolve a planned program to device addresses, upload it (ONE host-to-device copy: snapshot pairs, segments and
// micro-operations travel together) and enqueue the snapshot copies and the walk.
int runPlan(Instance* in, const mi355::Plan& plan, long planTag, hipEvent_t recordBeforeWalk) {
    const size_t n = plan.prog.size();
    typedef std::chrono::steady_clock PhaseClock;                    // (BEAGLE_MI355_HOST_TIMING=2: where a call that resolves a program spends its time)
    PhaseClock::time_point ph0 = PhaseClock::now(), ph1 = ph0, ph2 = ph0, ph3 = ph0;
    const long statsAtEntry[5] = {in->statMemReads, in->statTipReads, in->statScaleReads, in->statScaleWrites, in->statStored};
    if (n == 0) {                                  // nothing to compute (every destination became virtual): the definitions'
        if (plan.snapPairs.empty()) return 0;      // matrix snapshots still have to be taken
        void* dPairs = nullptr;
        int rc = uploadTransient(in, plan.snapPairs.data(), plan.snapPairs.size() * sizeof(int), &dPairs); if (rc) return rc;
        mi355::launchSnapshotMatrices(live(in), in->matrices, (const int*)dPairs, (int)(plan.snapPairs.size() / 2), in->C * in->S * in->S);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // Device program: per segment its micro-operations, a no-op when their number is odd, and two more no-ops the
    // kernel's descriptor prefetch may read (kernels.h WalkSeg).  For a plan that came out of the planner's cache the
    // resolved program is kept as well: buffer addresses never change once a buffer exists.
    static const int ablate = labEnv("BEAGLE_MI355_ABLATE") ? atoi(labEnv("BEAGLE_MI355_ABLATE")) : 0;     // (LAB builds only: wrong results)
    Instance::Resolved* slot = planTag && !ablate ? &in->resolved[planTag & 7] : nullptr;
    const bool reuse = slot && slot->tag == planTag && slot->epoch == in->resolveEpoch;
    in->lastResolveMiss = !reuse;
    std::vector<mi355::WalkOp>& w = slot ? slot->w : in->walkOps;
    std::vector<mi355::WalkSeg> segsLocal;
    std::vector<mi355::WalkSeg>& segs = slot ? slot->segs : segsLocal;
    std::vector<int> depsLocal;
    std::vector<int>& devDeps = slot ? slot->deps : depsLocal;     // per device slice: the device slices it waits for (one fused launch)
    std::vector<const double*> cmLocal;
    std::vector<int> sumRowsLocal, wroteLocal;                     // (a program outside the cache: Resolved::sumRows / wroteScale)
    std::vector<const double*>& cm = slot ? slot->cm : cmLocal;    // per device micro-operation: a fused cherry's two matrices (k_gatherMatrices)
    // 4 states, assembly loop: ALL slices in one launch, dispatched critical path first, every workgroup waiting for the slices
    // whose stored results it reads (planner.h PlanSeg; kernels_walk4.hip) — instead of one launch per wave of slices
    const bool fused = in->fuseWaves && in->fastWalk && !in->walkT && plan.launchOrder.size() == plan.segs.size();
    const bool asmLoop = in->fastWalk && !in->walkT;          // k_walk4_fast runs this program (otherwise k_walk4 / k_walkT32)
    // ... on tickets when its slices form a forest: the slices without dependencies first (they are the launch's grid), in launch order
    const bool ticket = fused && in->useTickets && plan.leaves > 0;
    std::vector<int> order;
    if (fused) {
        order = plan.launchOrder;
        if (ticket) std::stable_partition(order.begin(), order.end(), [&](int s) { return plan.segs[(size_t)s].depCount == 0; });
    }
    int maxRange = 0;
    if (reuse) {
        maxRange = slot->maxRange;
        in->statMemReads += slot->memReads; in->statTipReads += slot->tipReads; in->statScaleReads += slot->scaleReads;
        in->statScaleWrites += slot->scaleWrites; in->statStored += slot->stored; in->statFused += slot->fused;
    } else {
    const long s0[5] = {in->statMemReads, in->statTipReads, in->statScaleReads, in->statScaleWrites, in->statStored};
    if (in->folds.size() > 4096) dropFolds(in, true);           // (tree shapes come and go; the vectors are reused)
    if (slot) { slot->tag = 0; slot->dProgValid = false; slot->folds.clear(); slot->foldEpoch = -1; }
    // read-mode programs of cached (full-evaluation) plans fold the reciprocals of unstored nodes (Instance::folds, planner.h FoldMap)
    mi355::FoldMap foldMap;
    const bool fold = slot && in->foldScales && (in->walk || in->walkT) && slot->noFoldTag != planTag &&
                      mi355::foldScaleFactors(plan, FOLD_MAX_MEMBERS, foldMap);
    std::vector<int> cur;
    // fused cherries (kernels.h WK_CHERRY): the assembly loop only, and no program that rescales in write mode anywhere (the cherry
    // halves of the kernel's table buffers share their LDS with the maximum buffers of write-mode rescaling)
    bool anyWrite = false;
    for (const mi355::MicroOp& q : plan.prog) if (q.smode == mi355::PS_WRITE) { anyWrite = true; break; }
    const bool noWrites = asmLoop && in->walk && !anyWrite;
    // write-mode programs: every slice leaves the product of its factors behind (Instance::lastSums); the vectors are named by the last no-op
    // behind the slice's program.  They are the instance's, so growing them invalidates what other kept programs point at.
    const bool sums = anyWrite && in->walk && !in->walkT && in->sliceSums && in->partitionCount == 1;
    std::vector<int>& sumRows = slot ? slot->sumRows : sumRowsLocal;
    std::vector<int>& wroteScale = slot ? slot->wroteScale : wroteLocal;
    sumRows.clear(); wroteScale.clear();
    if (sums && in->sliceRows < plan.segs.size()) {
        HIP_TRY(hipStreamSynchronize(live(in)));
        if (in->sliceMant) hipFree(in->sliceMant);
        if (in->sliceExp) hipFree(in->sliceExp);
        in->sliceMant = nullptr; in->sliceExp = nullptr; in->sliceRows = 0;
        const size_t rows = plan.segs.size() + plan.segs.size() / 2 + 8;
        HIP_TRY(hipMalloc((void**)&in->sliceMant, rows * in->pairLen * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&in->sliceExp, rows * in->pairLen * sizeof(int)));
        in->sliceRows = rows;
        in->resolveEpoch++;
        in->lastSums.valid = false;
    }
    const bool fuseOk = noWrites && in->fuseCherries;
    // ... and in such programs the loop's fetch skips the tip-state load of a child that is no compact tip (kernels.h WF_NOLOAD1 / 2): a
    // vector-memory instruction less on the CU's address unit for half the children of a tree.  Programs that rescale in write mode keep
    // every fetch at its full size: their stage waits count on it (below).
    const unsigned skipLoads = noWrites && in->skipTipLoads ? (mi355::WF_NOLOAD1 | mi355::WF_NOLOAD2) : 0u;
    struct FusedAt { size_t at; int matA, matB; };
    std::vector<FusedAt> fusedAt;
    auto paysFactors = [&](int j) { return fold ? foldMap.payStart[(size_t)j + 1] > foldMap.payStart[(size_t)j] : plan.prog[(size_t)j].smode == mi355::PS_READ; };
    w.clear();
    w.reserve(n + 6 * plan.segs.size());
    segs.assign(plan.segs.size(), mi355::WalkSeg());
    devDeps.clear();
    std::vector<int> posOf(plan.segs.size(), -1);
    const size_t matStride = (size_t)in->C * in->S * in->S;
    // matrices whose snapshot is taken by THIS plan are gathered from the snapshot's source (same values; lets the snapshot copies and
    // the gather run in one launch: kernels_walk4.hip k_gatherAndSnapshot)
    // (a flat table over the matrix slots, entries put back behind the loop: a program on a new tree takes ~1 800 snapshots, and a hash
    // map of them cost 200 of the 270 us this resolution took)
    std::vector<int>& snapSrc = in->snapSourceOf;
    if (snapSrc.size() < (size_t)in->planner.matrixSlots()) snapSrc.assign((size_t)in->planner.matrixSlots(), -1);
    for (size_t q = 0; q + 1 < plan.snapPairs.size(); q += 2) snapSrc[(size_t)plan.snapPairs[q + 1]] = plan.snapPairs[q];
    struct SnapReset { std::vector<int>& t; const std::vector<int>& pairs; ~SnapReset() { for (size_t q = 0; q + 1 < pairs.size(); q += 2) t[(size_t)pairs[q + 1]] = -1; } } snapReset{snapSrc, plan.snapPairs};
    auto gatherFrom = [&](int mat) { const int s = snapSrc[(size_t)mat]; return s < 0 ? mat : s; };
    const size_t tipOff = in->walkT ? 0 : in->statePairOff;          // the T32 walk reads the plain state arrays and the RAW scale factors
    { int rc = ensureWalkDummies(in); if (rc) return rc; }
    mi355::WalkOp nop;
    memset(&nop, 0, sizeof(nop));
    nop.m1 = in->matrices; nop.m2 = in->matrices;
    nop.src1 = in->dummyTips; nop.src2 = in->dummyTips; nop.scale = in->onesScale;
    nop.flags = (unsigned)((mi355::WK_TIPS << 5) | (mi355::WK_TIPS << 8)) | skipLoads;         // loads nothing, stores nothing
    for (size_t oi = 0; oi < plan.segs.size(); oi++) {
        const size_t si = oi;                                   // position in the device program
        const mi355::PlanSeg& ps = plan.segs[fused ? (size_t)order[oi] : oi];
        posOf[fused ? (size_t)order[oi] : oi] = (int)oi;
        segs[si].depStart = (int)devDeps.size();
        if (fused) for (int d = ps.depStart; d < ps.depStart + ps.depCount; d++) {
            if (posOf[plan.deps[d]] < 0) return BEAGLE_ERROR_GENERAL;      // (a slice behind one that waits for it: the planner's order forbids it)
            devDeps.push_back(posOf[plan.deps[d]]);
        }
        segs[si].depCount = (int)devDeps.size() - segs[si].depStart;
        segs[si].progStart = (int)w.size();
        int lastStore = -1, lastHold = 0, cherry = -1;    // what the micro-operation emitted last stores / parks (1 + slot); a cherry waiting to be fused into the next one
        for (int i = ps.progStart; i < ps.progStart + ps.progCount; i++) {
            mi355::MicroOp m = plan.prog[i];
            // A cherry that is consumed at once is fused into its consumer (kernels.h WK_CHERRY): tip x tip, not stored, not parked, pays no
            // factors; the next micro-operation of the slice takes it as its second operand (ACC), multiplies by no reciprocals itself (the
            // descriptor's scale field carries the cherry's second tip) and would not come to sit right behind the micro-operation that
            // stores or parks its first child (the loop requests a first child from memory or from an LDS hold slot in the MIDDLE of the
            // stage before — in front of that stage's store and hold-slot write: with the cherry's own stage between them that was safe).
            if (fuseOk && i + 1 < ps.progStart + ps.progCount && m.k1 == mi355::PK_TIPS && m.k2 == mi355::PK_TIPS && m.storeBuf < 0 && m.hold == 0 &&
                !paysFactors(i) && in->tipStates[m.a1] && in->tipStates[m.a2]) {
                const mi355::MicroOp& nx = plan.prog[(size_t)i + 1];
                if (nx.k2 == mi355::PK_ACC && !paysFactors(i + 1) && !(nx.k1 == mi355::PK_MEM && lastStore == nx.a1) &&
                    !(nx.k1 >= mi355::PK_H0 && lastHold == nx.k1 - mi355::PK_H0 + 1)) { cherry = i; continue; }
            }
            // The kernels request a first child's partials one stage early — before the previous micro-operation's store
            // is issued (kernels_walk4.hip WALK_STAGE).  The planner never emits that sequence (tests/native/plan_check.cpp
            // checks every program for it); should one arrive anyway, a no-op in between restores the distance.
            if ((int)w.size() > segs[si].progStart && m.k1 == mi355::PK_MEM && lastStore == m.a1) {
                if (m.k2 == mi355::PK_ACC) { m.k2 = mi355::PK_MEM; m.a2 = m.a1; }      // the no-op overwrites ACC; the value is in memory as well
                w.push_back(nop);
            }
            mi355::WalkOp d;
            memset(&d, 0, sizeof(d));
            d.src1 = in->dummyTips; d.src2 = in->dummyTips; d.scale = in->onesScale;     // unused operands stay readable (kernels.h launchWalk4Fast)
            if (m.k1 == mi355::PK_MEM) { d.src1 = in->partials[m.a1]; if (!d.src1 || isCompactTip(in, m.a1)) return BEAGLE_ERROR_OUT_OF_RANGE; in->statMemReads++; }
            else if (m.k1 == mi355::PK_TIPS) { if (!in->tipStates[m.a1]) return BEAGLE_ERROR_OUT_OF_RANGE; d.src1 = in->tipStates[m.a1] + tipOff; in->statTipReads++; }
            if (m.k2 == mi355::PK_MEM) { d.src2 = in->partials[m.a2]; if (!d.src2 || isCompactTip(in, m.a2)) return BEAGLE_ERROR_OUT_OF_RANGE; in->statMemReads++; }
            else if (m.k2 == mi355::PK_TIPS) { if (!in->tipStates[m.a2]) return BEAGLE_ERROR_OUT_OF_RANGE; d.src2 = in->tipStates[m.a2] + tipOff; in->statTipReads++; }
            if (m.smode != mi355::PS_NONE) {
                int rc = ensureScale(in, m.scaleIdx); if (rc) return rc;
                if (m.smode == mi355::PS_WRITE) { if (in->walkT && !in->walkTWrite) return BEAGLE_ERROR_GENERAL; in->scaleIsRaw[m.scaleIdx] = 1; in->statScaleWrites++; d.scaleW = in->scale[m.scaleIdx];
                                                  if (sums) { wroteScale.push_back(m.scaleIdx); if (sumRows.empty() || sumRows.back() != (int)si) sumRows.push_back((int)si); } }
                else {
                    if (!in->scaleIsRaw[m.scaleIdx]) return BEAGLE_ERROR_OUT_OF_RANGE;   // never written by a rescaling op
                    if (!fold) {
                        in->statScaleReads++;
                        d.scale = in->scale[m.scaleIdx] + (in->walkT ? 0 : in->scaleStride);  // read mode multiplies by the reciprocal
                    }
                }
            }
            int smodeNow = m.smode;
            if (fold) {                            // multiply by what the planner says this result pays for — nothing, one buffer's reciprocals, a fold
                const int b0 = foldMap.payStart[(size_t)i], b1 = foldMap.payStart[(size_t)i + 1];
                smodeNow = b1 > b0 ? mi355::PS_READ : mi355::PS_NONE;
                if (b1 - b0 == 1) d.scale = in->scale[foldMap.members[(size_t)b0]] + (in->walkT ? 0 : in->scaleStride);
                else if (b1 > b0) {
                    cur.assign(foldMap.members.begin() + b0, foldMap.members.begin() + b1);
                    const int f = foldFor(in, cur);
                    if (f < 0) return BEAGLE_ERROR_OUT_OF_MEMORY;
                    slot->folds.push_back(f);
                    d.scale = in->folds[(size_t)f].recip;
                }
                if (b1 > b0) in->statScaleReads++;
            }
            if (m.storeBuf >= 0) {
                int rc = ensurePartials(in, m.storeBuf); if (rc) return rc;
                d.store = in->partials[m.storeBuf];
                in->statStored++;
            }
            d.m1 = in->matrices + (size_t)gatherFrom(m.mat1) * matStride; d.m2 = in->matrices + (size_t)gatherFrom(m.mat2) * matStride;
            int k2 = m.k2;
            if (cherry >= 0) {
                const mi355::MicroOp& c = plan.prog[(size_t)cherry];
                d.src2 = in->tipStates[c.a1] + tipOff; d.scale = (const double*)(in->tipStates[c.a2] + tipOff);
                k2 = mi355::WK_CHERRY;
                fusedAt.push_back(FusedAt{w.size(), gatherFrom(c.mat1), gatherFrom(c.mat2)});
                in->statTipReads += 2;
                cherry = -1;
            }
            d.flags = mi355::walkFlags(m.k1, k2, m.hold, smodeNow, m.storeBuf >= 0);
            if (m.k1 != mi355::PK_TIPS) d.flags |= skipLoads & mi355::WF_NOLOAD1;
            if (k2 != mi355::PK_TIPS) d.flags |= skipLoads & mi355::WF_NOLOAD2;
            if (ablate) {       // TIMING EXPERIMENTS ONLY (wrong results): 1 no stores, 2 no partials loads, 4 no scale traffic, 8 no tip traffic
                if (ablate & 1) d.flags &= ~(unsigned)mi355::WF_STORE;
                if (ablate & 2) d.flags &= ~(unsigned)mi355::WF_X;
                if (ablate & 4) d.scale = in->onesScale;
                if (ablate & 8) { if (m.k1 == mi355::PK_TIPS) d.src1 = in->dummyTips; if (m.k2 == mi355::PK_TIPS) d.src2 = in->dummyTips; }
            }
            w.push_back(d);
            lastStore = m.storeBuf; lastHold = m.hold;
        }
        // (the kernels' software pipelines: the assembly loop is three micro-operations deep and leaves behind any stage — any
        // length, three more readable descriptors —; the C++ kernel and the T32 walk are two deep: an even length, two more)
        if (!asmLoop && !(in->walkT && in->S > 20) && ((int)w.size() - segs[si].progStart) % 2) w.push_back(nop);      // (k_walkT64 prefetches nothing across stages)
        segs[si].progCount = (int)w.size() - segs[si].progStart;
        for (int q = 0; q < (asmLoop ? 3 : 2); q++) w.push_back(nop);
        if (sums && !sumRows.empty() && sumRows.back() == (int)si) {      // (this slice writes factors: where its product of them goes)
            w.back().scaleW = in->sliceMant + si * in->pairLen;
            w.back().store = (double*)(in->sliceExp + si * in->pairLen);
        }
        // the wait of every stage: "at most N vector-memory instructions outstanding".  Loads and stores share the counter.
        // DEFAULT (strict): N = the LOADS issued behind this micro-operation's own.  Sufficient under the one ordering rule the ISA
        // guides state for this counter — vector-memory LOADS return in the order they were issued: when at most N operations
        // are outstanding and the N youngest loads are all younger than this stage's loads, an unfinished load of this stage
        // would leave N + 1 unfinished, whatever the stores (of this or any earlier stage) do.
        // BEAGLE_MI355_STRICT_WAITS=0: N also counts the stores issued in between, i.e. assumes that a younger store
        // is never counted out before an older load.  That held in > 1e9 lane-trials (tests/test_gpu_vmcnt_order.py) — but it
        // is an observation, not a documented guarantee, so it is not what ships by default.
        // A smaller N than the true number only waits longer.
        const int first = segs[si].progStart;
        if (asmLoop) {
            // The stage waits below count LOADS.  Stores share the counter: a stage that finds stores of the two micro-operations
            // before it still unacknowledged waits for as many of them as its N falls short of "loads + stores".  With four loads per
            // fetch that shortfall was covered by the loads of the fetch before (older than the stores, long landed); with three it
            // is not — a write-mode evaluation (every micro-operation stores its factors) lost 8 % to it.  So a micro-operation
            // whose fetch is in flight across such stores fetches four again: WF_INV on top, its scale address the all-ones array
            // (a multiplication by one where it multiplies).  Behind write-mode rescaling only: behind the (rare) stored results of a
            // read-mode program the padding costs what it saves (config A 503 / 503 / 507 us without, with, and with both; ALWAYS
            // 1 053 / 1 003 / 1 005: profiles/r05_experiments.txt).  LAB builds: BEAGLE_MI355_WALK_PAD_FETCH = 0 never, 1 (default), 2 both.
            static const int padMode = labEnv("BEAGLE_MI355_WALK_PAD_FETCH") ? atoi(labEnv("BEAGLE_MI355_WALK_PAD_FETCH")) : 1;
            for (int i = first + segs[si].progCount - 1; i >= first + 2; i--) {      // (backwards: the test reads unpadded flags only of earlier ones — WF_INV is not what it looks at)
                bool pad = false;
                for (int b = 2; b <= 4 && i - b >= first && !pad; b++) {
                    const unsigned f = w[i - b].flags;
                    pad = (padMode >= 1 && ((f >> 13) & 3u) == (unsigned)mi355::WS_WRITE) || (padMode >= 2 && (f & mi355::WF_STORE));
                }
                if (pad) w[i].flags |= mi355::WF_INV;
            }
        }
        for (int i = first; i < first + segs[si].progCount; i++) {
            // k_walk4 (two deep): N = the fetch of the next micro-operation (+ the previous one's stores)
            // (a micro-operation that rescales in write mode also stores its factors, from one of the workgroup's waves only: behind
            // it the count is the strict one whatever the mode)
            const bool prevWrites = i > first && ((w[i - 1].flags >> 13) & 3u) == (unsigned)mi355::WS_WRITE;
            const int stores1 = (in->strictWaits || prevWrites) ? 0 : (i > first ? mi355::walkStoreCount(w[i - 1].flags) : 0);
            if (!asmLoop) { w[i].flags |= mi355::walkWaitJump(std::min(mi355::walkFetchCount(w[i + 1].flags) + stores1, 12)); continue; }
            // k_walk4_fast (three deep; a fetch is THREE small loads, four with the reciprocal scale factors: WF_INV).  Issue order around
            // stage i: ... fetch(i) | first child of i - 1 from memory (4) | store(i - 2) | fetch(i + 1) | first child of i from memory
            // (4) | store(i - 1) | fetch(i + 2) | WAIT.  A first child of i in memory has to have landed as well: then only what
            // follows it counts.  (Loads only — the strict rule — whatever BEAGLE_MI355_STRICT_WAITS says: with two fetch sizes the
            // code space has no room for the store counts of the lax rule, which bought 1 %.)
            auto fetchLoads = [&](int j) { const unsigned f = w[j].flags;             // (matrix table; tip states, twice; reciprocals; a fused cherry's table half and tips)
                                           return 1 + ((f & mi355::WF_NOLOAD1) ? 0 : 1) + ((f & mi355::WF_NOLOAD2) ? 0 : 1) + ((f & mi355::WF_INV) ? 1 : 0) + ((f & mi355::WF_CHERRY2) ? 3 : 0); };
            const int x1 = i > first && (w[i - 1].flags & mi355::WF_X) ? 4 : 0;
            const int nWait = (w[i].flags & mi355::WF_X) ? fetchLoads(i + 2) : fetchLoads(i + 1) + fetchLoads(i + 2) + x1;
            w[i].flags |= mi355::walkWaitCode(nWait);
        }
        segs[si].pStart = in->partStart[ps.partition]; segs[si].pEnd = in->partEnd[ps.partition]; segs[si].tStart = in->padStart[ps.partition];
        maxRange = std::max(maxRange, segs[si].pEnd - segs[si].pStart);
    }
    for (size_t oi = 0; oi < plan.segs.size(); oi++) {          // (kernels.h WalkSeg::next: rows of THIS array)
        const mi355::PlanSeg& ps = plan.segs[fused ? (size_t)order[oi] : oi];
        segs[oi].next = ticket && ps.next >= 0 ? posOf[(size_t)ps.next] : -1;
    }
    if (slot) slot->leaves = ticket ? plan.leaves : 0;
    cm.clear();
    if (!fusedAt.empty()) {
        cm.assign(2 * w.size(), nullptr);
        for (const FusedAt& f : fusedAt) { cm[2 * f.at] = in->matrices + (size_t)f.matA * matStride; cm[2 * f.at + 1] = in->matrices + (size_t)f.matB * matStride; }
    }
    in->statFused += (long)fusedAt.size();
    if (slot) slot->fused = (long)fusedAt.size();
    if (slot) {
        slot->tag = planTag; slot->epoch = in->resolveEpoch; slot->maxRange = maxRange;
        slot->memReads = in->statMemReads - s0[0]; slot->tipReads = in->statTipReads - s0[1]; slot->scaleReads = in->statScaleReads - s0[2];
        slot->scaleWrites = in->statScaleWrites - s0[3]; slot->stored = in->statStored - s0[4];
    }
    }
    if (anyScaleWriteIn(in, slot, reuse, statsAtEntry[3])) scalesWritten(in);
    {   // the slices' products of factors this run leaves behind, and which scale buffers they cover (engine_abi.cpp accumulate)
        const std::vector<int>& rows = slot ? slot->sumRows : sumRowsLocal;
        const std::vector<int>& wrote = slot ? slot->wroteScale : wroteLocal;
        if (!rows.empty()) {
            if (in->scaleGen.size() < (size_t)in->scaleCount) { in->scaleGen.assign((size_t)in->scaleCount, 0); in->scaleSeen.assign((size_t)in->scaleCount, 0); }
            const long gen = ++in->sliceGen;
            for (int idx : wrote) in->scaleGen[(size_t)idx] = gen;
            in->lastSums.valid = true; in->lastSums.epoch = in->scaleWriteEpoch; in->lastSums.gen = gen; in->lastSums.rows = rows; in->lastSums.nWritten = (int)wrote.size();
        }
    }
    if (slot && !slot->folds.empty()) {
        if (slot->foldEpoch != in->scaleWriteEpoch) {
            bool bad = false;
            int rcf = refreshFolds(in, slot->folds, &bad); if (rcf) return rcf;
            slot->foldEpoch = in->scaleWriteEpoch;
            if (bad) {                                 // out of range: this plan keeps per-node factors from now on
                in->statMemReads = statsAtEntry[0]; in->statTipReads = statsAtEntry[1]; in->statScaleReads = statsAtEntry[2];
                in->statScaleWrites = statsAtEntry[3]; in->statStored = statsAtEntry[4];
                slot->noFoldTag = planTag; slot->tag = 0; slot->dProgValid = false; slot->folds.clear();
                return runPlan(in, plan, planTag, recordBeforeWalk);
            }
        }
        in->statFoldedVectors = (long)slot->folds.size();
    }
    in->statMicroOps += (long)n;
    ph1 = PhaseClock::now();
    // pack: [micro-ops (64 B each) | segments (32 B each) | dependency lists | snapshot pairs] — ONE host-to-device copy
    const size_t opBytes = w.size() * sizeof(mi355::WalkOp), segBytes = segs.size() * sizeof(mi355::WalkSeg) + ((devDeps.size() * sizeof(int) + 31) & ~(size_t)31);
    // (... | the matrices of fused cherries, two pointers per micro-operation, where the program has any)
    const size_t pairBytes = plan.snapPairs.size() * sizeof(int), cmOff = opBytes + segBytes + ((pairBytes + 15) & ~(size_t)15), cmBytes = cm.size() * sizeof(const double*);
    const size_t total = cmOff + cmBytes;
    const size_t depOff = opBytes + segs.size() * sizeof(mi355::WalkSeg);
    char* dBase = nullptr;
    const char* stagedProg = nullptr;                             // the program as this call staged it, seen through the ring's device mapping
    if (reuse && slot->dProgValid) dBase = slot->dProg;          // a cached plan's program is already on the device, bit for bit
    else if (total <= RING_BYTES / 4) {
        const long off = stage(in, w.data(), opBytes, total);                    // reserves `total` bytes, copies the ops ...
        if (off < 0) return BEAGLE_ERROR_GENERAL;
        memcpy(in->hRing + off + opBytes, segs.data(), segs.size() * sizeof(mi355::WalkSeg));     // ... the rest is filled in behind them
        if (!devDeps.empty()) memcpy(in->hRing + off + depOff, devDeps.data(), devDeps.size() * sizeof(int));
        if (pairBytes) memcpy(in->hRing + off + opBytes + segBytes, plan.snapPairs.data(), pairBytes);
        if (cmBytes) memcpy(in->hRing + off + cmOff, cm.data(), cmBytes);
        { int rcq = queueCopy(in, in->dRing + off, (size_t)off, total); if (rcq) return rcq; }
        dBase = in->dRing + off;
        if (in->kernelUploads) stagedProg = (const char*)in->hRingDev + off;
        if (slot) {                                   // keep a device copy for the next time this plan comes out of the cache
            if (slot->dProgBytes < total) {
                if (slot->dProg) { HIP_TRY(hipStreamSynchronize(live(in))); hipFree(slot->dProg); }
                slot->dProg = nullptr; slot->dProgBytes = 0;
                HIP_TRY(hipMalloc((void**)&slot->dProg, total + total / 4));
                slot->dProgBytes = total + total / 4;
            }
            if (in->kernelUploads) { int rcq = queueCopy(in, slot->dProg, (size_t)off, total); if (rcq) return rcq; }     // (from the same staged bytes)
            else HIP_TRY(hipMemcpyAsync(slot->dProg, in->dRing + off, total, hipMemcpyDeviceToDevice, live(in)));
            slot->dProgValid = true;
        }
    } else {                                  // a tree of > ~60 000 nodes: its own staging buffer, synchronous copy
        HIP_TRY(hipStreamSynchronize(live(in)));
        if (in->bigStageBytes < total) {
            if (in->bigStage) hipFree(in->bigStage);
            in->bigStage = nullptr; in->bigStageBytes = 0;
            HIP_TRY(hipMalloc((void**)&in->bigStage, total));
            in->bigStageBytes = total;
        }
        HIP_TRY(hipMemcpy(in->bigStage, w.data(), opBytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(in->bigStage + opBytes, segs.data(), segs.size() * sizeof(mi355::WalkSeg), hipMemcpyHostToDevice));
        if (!devDeps.empty()) HIP_TRY(hipMemcpy(in->bigStage + depOff, devDeps.data(), devDeps.size() * sizeof(int), hipMemcpyHostToDevice));
        if (pairBytes) HIP_TRY(hipMemcpy(in->bigStage + opBytes + segBytes, plan.snapPairs.data(), pairBytes, hipMemcpyHostToDevice));
        if (cmBytes) HIP_TRY(hipMemcpy(in->bigStage + cmOff, cm.data(), cmBytes, hipMemcpyHostToDevice));
        dBase = in->bigStage;
    }
    ph2 = PhaseClock::now();
    const bool fusedSnapshot = pairBytes && !in->walkT && in->fuseLaunches;          // 4 states: together with the gather below
    if (pairBytes && !fusedSnapshot)
        mi355::launchSnapshotMatrices(live(in), in->matrices, (const int*)(dBase + opBytes + segBytes), (int)(plan.snapPairs.size() / 2),
                                      in->C * in->S * in->S);
    // the matrix stream: both branch matrices of every micro-operation, in program order (after the snapshots they may name)
    const size_t streamBytes = in->walkT ? mi355::walkT32StreamBytes((int)w.size(), in->C, in->S) + 8192          // (the kernel's second fragment load reads up to 1.8 KB past an entry)
                                         : w.size() * (size_t)in->C * 40 * sizeof(double) * (cmBytes ? 2 : 1) + 1024;   // 2 x 5 columns x 4 per category; behind them the cherry region (kernels_walk4.hip)
    if (in->matStreamBytes < streamBytes) {
        HIP_TRY(hipStreamSynchronize(live(in)));
        if (in->matStream) hipFree(in->matStream);
        in->matStream = nullptr; in->matStreamBytes = 0;
        const size_t want = std::max(streamBytes + streamBytes / 4, (size_t)1 << 20);
        HIP_TRY(hipMalloc((void**)&in->matStream, want));
        in->matStreamBytes = want;
    }
    // 4 states, a program staged by this call (a partial update, a list the engine has not seen): its upload — and whatever else is
    // queued — rides in the gather's launch, which reads the program through the ring's mapping meanwhile (kernels_walk4.hip
    // k_gatherAndSnapshot): three launches per such evaluation instead of four.  Not when a queued copy lands in what the gather
    // reads or writes (the matrix block, the stream), or two queued copies overlap in part: then they go first, in order (live()).
    bool uploadsRide = !in->walkT && in->fuseLaunches && stagedProg && !in->pendingCopies.empty() && (int)in->pendingCopies.size() <= mi355::HOST_COPY_MAX;
    if (uploadsRide) {
        const char* m0 = (const char*)in->matrices;
        const char* m1 = m0 + (size_t)std::max(1, in->planner.matrixSlots()) * in->C * in->S * in->S * sizeof(double);
        const char* t0 = (const char*)in->matStream; const char* t1 = t0 + in->matStreamBytes;
        const std::vector<Instance::PendingCopy>& pc = in->pendingCopies;
        for (size_t a = 0; a < pc.size() && uploadsRide; a++) {
            const char* d0 = (const char*)pc[a].dst; const char* d1 = d0 + pc[a].bytes;
            if ((d0 < m1 && m0 < d1) || (d0 < t1 && t0 < d1)) uploadsRide = false;
            for (size_t b = a + 1; b < pc.size() && uploadsRide; b++) {
                const char* e0 = (const char*)pc[b].dst; const char* e1 = e0 + pc[b].bytes;
                if (d0 < e1 && e0 < d1 && !(e0 <= d0 && d1 <= e1)) uploadsRide = false;
            }
        }
    }
    if (uploadsRide) {
        if (in->pendingWalk.valid) { int rcw = flushWalk(in); if (rcw) return rcw; }          // (cannot be: queueCopy above launched it)
        mi355::HostCopyList L;
        L.n = 0;
        unsigned blocks = 0;
        for (const Instance::PendingCopy& pc : in->pendingCopies) {
            mi355::HostCopyList::Entry& e = L.e[L.n++];
            e.dst = pc.dst; e.src = in->hRingDev + pc.ringOff; e.bytes = (unsigned)pc.bytes; e.firstBlock = blocks;
            blocks += (unsigned)((pc.bytes + 4095) / 4096);
        }
        for (int a = 0; a < L.n; a++)                 // (an array queued twice: the later upload wins, the covered one is dropped)
            for (int b = a + 1; b < L.n; b++)
                if ((const char*)L.e[b].dst <= (const char*)L.e[a].dst && (const char*)L.e[a].dst + L.e[a].bytes <= (const char*)L.e[b].dst + L.e[b].bytes) L.e[a].bytes = 0;
        in->pendingCopies.clear();
        mi355::launchGatherAndSnapshot(in->stream, (const mi355::WalkOp*)stagedProg, (int)w.size(), in->C, in->matStream, in->matrices,
                                       (const int*)(stagedProg + opBytes + segBytes), (int)(plan.snapPairs.size() / 2), in->C * in->S * in->S, &L, (int)blocks,
                                       cmBytes ? (const double* const*)(stagedProg + cmOff) : nullptr);
    }
    else if (in->walkT) mi355::launchGatherFragments(live(in), (const mi355::WalkOp*)dBase, (int)w.size(), in->C, in->S, in->matStream);
    else if (fusedSnapshot) mi355::launchGatherAndSnapshot(live(in), (const mi355::WalkOp*)dBase, (int)w.size(), in->C, in->matStream, in->matrices,
                                                           (const int*)(dBase + opBytes + segBytes), (int)(plan.snapPairs.size() / 2), in->C * in->S * in->S, nullptr, 0,
                                                           cmBytes ? (const double* const*)(dBase + cmOff) : nullptr);
    else mi355::launchGatherMatrices(live(in), (const mi355::WalkOp*)dBase, (int)w.size(), in->C, in->matStream,
                                     cmBytes ? (const double* const*)(dBase + cmOff) : nullptr);
    if (labEnv("BEAGLE_MI355_DUMP_PLAN")) {           // development (LAB builds): the slices of this program, wave by wave
        fprintf(stderr, "[mi355] plan: %zu micro-ops in %zu slices:", n, segs.size());
        for (size_t i = 0; i < segs.size(); i++) {
            const mi355::PlanSeg& ps = plan.segs[fused ? (size_t)order[i] : i];
            fprintf(stderr, " w%d:%d", ps.wave, segs[i].progCount);
            if (fused) fprintf(stderr, "(t%d d%d)", ps.tail, ps.depCount);
        }
        fprintf(stderr, "\n");
        if (atoi(labEnv("BEAGLE_MI355_DUMP_PLAN")) > 1)
            for (size_t i = 0; i < w.size(); i++)
                fprintf(stderr, "[mi355]   %3zu: k1 %u k2 %u hold %u scale %u store %d\n", i, (w[i].flags >> 5) & 7, (w[i].flags >> 8) & 7, (w[i].flags >> 11) & 3,
                        (w[i].flags >> 13) & 3, (w[i].flags & mi355::WF_STORE) ? 1 : 0);
    }
    ph3 = PhaseClock::now();
    if (in->hostTrace) {                               // (BEAGLE_MI355_HOST_TIMING=2: what the first few gather launches of the process were made of)
        static std::atomic<int> left{6};
        if (left.fetch_sub(1) > 0)
            fprintf(stderr, "[mi355] gather launch: %zu micro-operations x %d categories%s, %zu matrix snapshots, program %s (%zu bytes), %zu queued uploads ride along\n",
                    w.size(), in->C, cmBytes ? " + cherry region" : "", plan.snapPairs.size() / 2, stagedProg ? "staged by this call" : "on the device",
                    total, uploadsRide ? (size_t)1 : (size_t)0);
    }
    if (in->hostTrace && !reuse && n >= 64) {
        auto us = [](PhaseClock::time_point a, PhaseClock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[mi355] program of %zu micro-operations resolved: descriptors and waits %.1f us, upload %.1f us, stream + gather launch %.1f us\n", n, us(ph0, ph1), us(ph1, ph2), us(ph2, ph3));
    }
    if (recordBeforeWalk) HIP_TRY(hipEventRecord(recordBeforeWalk, live(in)));
    if (fused) {
        // ONE launch: slice y of the device program is dispatched before slice y + 1 (x fastest), every slice behind the ones it
        // waits for; a workgroup signals flags[y][x] = epoch when its stores are out, its dependants poll for exactly that value
        int range = 0;
        for (size_t i = 0; i < segs.size(); i++) range = std::max(range, segs[i].pEnd - segs[i].pStart);
        const int flagStride = (in->P + 127) / 128 + 1;
        const size_t flagBytes = segs.size() * (size_t)flagStride * sizeof(unsigned);
        if (in->walkFlagBytes < flagBytes) {
            HIP_TRY(hipStreamSynchronize(live(in)));
            if (in->walkFlags) hipFree(in->walkFlags);
            in->walkFlags = nullptr; in->walkFlagBytes = 0; in->walkTickets = nullptr;
            const size_t want = (std::max(flagBytes + flagBytes / 2, (size_t)1 << 16) + 255) & ~(size_t)255;
            HIP_TRY(hipMalloc((void**)&in->walkFlags, 2 * want));               // [flags | tickets]
            HIP_TRY(hipMemsetAsync(in->walkFlags, 0, 2 * want, live(in)));
            in->walkFlagBytes = want;
            in->walkTickets = (unsigned*)((char*)in->walkFlags + want);
        }
        if (++in->walkEpoch == 0u) in->walkEpoch = 1u;
        Instance::PendingWalk& pw = in->pendingWalk;
        if (pw.valid) { int rcf = flushWalk(in); if (rcf) return rcf; }            // (cannot happen: every path here went through live())
        pw.prog = (const mi355::WalkOp*)dBase; pw.segs = (const mi355::WalkSeg*)(dBase + opBytes); pw.deps = (const int*)(dBase + depOff);
        pw.nSegs = (int)segs.size(); pw.range = range; pw.flagStride = flagStride; pw.epoch = in->walkEpoch;
        pw.leaves = slot && reuse ? slot->leaves : (ticket ? plan.leaves : 0);
        pw.cherryOff = cmBytes ? (unsigned)(w.size() * (size_t)in->C * 40 * sizeof(double)) : 0u;
        in->statFastWalks++; in->statWalks++;
        // hold the launch back for the root call?  (one partition, the whole range, not inside a timer bracket)
        // ... and only a program whose slices ALL lead to one last slice: the root call's result word then says that every workgroup
        // of the launch is done (waitResult and its callers reset the staging ring on seeing it) — a forest's other trees could
        // still be running behind the slice that publishes
        int sinks = slot && reuse ? slot->sinks : 0;
        if (!(slot && reuse)) {
            std::vector<char> feeds(segs.size(), 0);
            for (int d : devDeps) feeds[(size_t)d] = 1;
            for (size_t i = 0; i < segs.size(); i++) if (!feeds[i] && segs[i].progCount > 0) sinks++;
            if (slot) slot->sinks = sinks;
        }
        // (a partitioned instance, round 6: up to eight partitions in the list, every one ending in ONE slice — the by-partition root call
        // names them all or the launch goes out without it: engine_abi.cpp beagleCalculateRootLogLikelihoodsByPartition)
        const bool holdParts = in->partitionCount > 1 && sinks >= 1 && sinks <= mi355::ROOT_MAX_PARTS && in->fuseRootParts;
        const bool hold = in->deferWalk && in->fuseLaunches && !recordBeforeWalk && ((in->partitionCount == 1 && range == in->P && sinks == 1) || holdParts);
        if (hold) {
            pw.finalStore.assign(segs.size(), -1);
            pw.finalPart.assign(segs.size(), 0);
            pw.sinkRows.clear();
            std::vector<char> feeds(segs.size(), 0);
            for (int d : devDeps) feeds[(size_t)d] = 1;         // (the slot's own list: what it held when the program was resolved)
            for (size_t i = 0; i < segs.size(); i++) {
                const mi355::PlanSeg& ps = plan.segs[(size_t)order[i]];
                if (ps.progCount > 0) pw.finalStore[i] = plan.prog[(size_t)ps.progStart + ps.progCount - 1].storeBuf;
                pw.finalPart[i] = ps.partition;
                if (!feeds[i] && ps.progCount > 0) pw.sinkRows.push_back((int)i);
            }
            (void)live(in);                           // the program's copies and the gather are enqueued; only the walk itself waits
            pw.valid = true;
            return 0;
        }
        pw.valid = true;
        return flushWalk(in);
    }
    // one launch per wave of independent slices (a single one unless the planner cut the forest for a small shard)
    for (size_t b = 0; b < segs.size();) {
        size_t e = b + 1;
        while (e < segs.size() && plan.segs[e].wave == plan.segs[b].wave) e++;
        int range = 0;
        const bool fast = in->fastWalk;                 // the assembly loop (BEAGLE_MI355_NO_FAST_WALK=1: the C++ reference kernel)
        for (size_t i = b; i < e; i++) range = std::max(range, segs[i].pEnd - segs[i].pStart);
        if (in->walkT) {
            if (!mi355::launchWalkT32(live(in), (const mi355::WalkOp*)dBase, (const mi355::WalkSeg*)(dBase + opBytes) + b, (int)(e - b), range,
                                      in->matStream, in->P, in->S, in->C, in->holdSlots, anyScaleWriteIn(in, slot, reuse, statsAtEntry[3]))) return BEAGLE_ERROR_GENERAL;
        } else if (fast) {
            mi355::launchWalk4Fast(live(in), (const mi355::WalkOp*)dBase, (const mi355::WalkSeg*)(dBase + opBytes) + b, (int)(e - b), range,
                                   in->matStream, in->P, in->C, (long)in->scaleStride, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, false,
                                   cmBytes ? (unsigned)(w.size() * (size_t)in->C * 40 * sizeof(double)) : 0u);
            in->statFastWalks++;
        } else
            mi355::launchWalk4(live(in), (const mi355::WalkOp*)dBase, (const mi355::WalkSeg*)(dBase + opBytes) + b, (int)(e - b), range,
                               in->matStream, in->P, in->C, (long)in->scaleStride);
        in->statWalks++;
        b = e;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Launch the walk that was held back (engine_internal.h PendingWalk) — as it is, or with the slice `root` names finishing the evaluation.
int flushWalk(Instance* in, const mi355::RootFused* root) {
    Instance::PendingWalk& pw = in->pendingWalk;
    if (!pw.valid) return 0;
    pw.valid = false;                                  // (before anything that could come back here through live())
    if (!in->pendingCopies.empty()) { int rc = flushUploads(in); if (rc) return rc; }
#ifdef BEAGLE_MI355_LAB
    // BEAGLE_MI355_WALK_TRACE=<n>: the n-th held-or-not one-launch walk of the instance with more than 20 slices is timed workgroup by workgroup
    static const int traceAt = labEnv("BEAGLE_MI355_WALK_TRACE") ? atoi(labEnv("BEAGLE_MI355_WALK_TRACE")) : 0;
    static int traceSeen = 0;
    unsigned long long* dTrace = nullptr;
    const int groupsX = (pw.range + 127) / 128;
    if (traceAt > 0 && pw.nSegs > 20 && ++traceSeen == traceAt) {
        HIP_TRY(hipMalloc((void**)&dTrace, (size_t)pw.nSegs * groupsX * 3 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(dTrace, 0, (size_t)pw.nSegs * groupsX * 3 * sizeof(unsigned long long), in->stream));
        mi355::setWalkTrace(dTrace);
    }
#endif
    mi355::launchWalk4Fast(in->stream, pw.prog, pw.segs, pw.nSegs, pw.range, in->matStream, in->P, in->C, (long)in->scaleStride,
                           pw.deps, in->walkFlags, pw.epoch, pw.flagStride, root, in->walkSpinLimit, in->walkSelfServed,
                           pw.leaves > 0 ? in->walkTickets : nullptr, pw.leaves, in->xcdAware, pw.cherryOff);
#ifdef BEAGLE_MI355_LAB
    if (dTrace) {
        HIP_TRY(hipStreamSynchronize(in->stream));
        std::vector<unsigned long long> t((size_t)pw.nSegs * groupsX * 3);
        HIP_TRY(hipMemcpy(t.data(), dTrace, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        mi355::setWalkTrace(nullptr);
        hipFree(dTrace);
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < t.size(); i += 3) if (t[i] && t[i] < t0) t0 = t[i];
        fprintf(stderr, "[mi355] walk trace, %d slices x %d groups (us from the first workgroup's entry: entry min..max | wait over min..max | end min..max):\n", pw.nSegs, groupsX);
        for (int s = 0; s < pw.nSegs; s++) {
            double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {0, 0, 0};
            for (int g = 0; g < groupsX; g++)
                for (int k = 0; k < 3; k++) {
                    const unsigned long long v = t[((size_t)s * groupsX + g) * 3 + k];
                    if (!v) continue;
                    const double us = (double)(v - t0) / 100.0;
                    lo[k] = std::min(lo[k], us); hi[k] = std::max(hi[k], us);
                }
            fprintf(stderr, "[mi355]   slice %2d: %6.1f..%6.1f | %6.1f..%6.1f | %6.1f..%6.1f\n", s, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
        }
    }
#endif
    if (root) in->statRootFused++;
    if (pw.leaves > 0) in->statTicketWalks++; else in->statFlagWalks++;
    in->lastLaunchRows = pw.leaves > 0 ? pw.leaves : pw.nSegs; in->lastLaunchSlices = pw.nSegs;
    HIP_TRY(hipGetLastError());
    return 0;
}

// Give every definition of `xs` its real partials: one program, one launch.  xs are definition KEYS (planner.h: buffer x
// partitionCount + partition; the buffer index itself on an instance with one partition).
int materializeList(Instance* in, const std::vector<int>& xs) {
    if (!in->virt || xs.empty()) return 0;
    if (!in->walk && !in->walkT) return materializeCherries(in, xs);
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


// A walk is one workgroup per 128 patterns, and every wave executes its program one dependent step after the other.  The
// planner therefore cuts the forest into independent subtrees that run side by side, wave after wave (planner.h): with few
// patterns (a shard of a multi-GPU run, a small alignment) that is what fills the 256 CUs at all (12 500 patterns:
// 0.83 -> 0.33 ms per evaluation); with many it keeps more workgroups than the chip holds in the queue, so that a wave
// waiting for its stores is replaced by another one instead of idling (1e5 patterns: 1.73 -> 1.37 ms).  Returns the target
// number of micro-operations per subtree, 0 = one walk.  BEAGLE_MI355_CHUNK overrides (0 = never).
int walkChunkOps(const Instance* in, int opCount) {
    static const int forced = labEnv("BEAGLE_MI355_CHUNK") ? atoi(labEnv("BEAGLE_MI355_CHUNK")) : -1;
    if (forced >= 0) return forced;
    // (a short list is one walk — below 64 operations since round 2; on tickets, where a slice above the first wave costs no workgroup
    // slots and no polling, from 32: the reference's benchmark2 alignment — 62 taxa, 44 pattern groups — ran its 61 operations as ONE
    // serial slice, 32 us; cut into three and a top it takes 23: tools/r06_small_sweep.sh)
    if (opCount < (in->fuseWaves && in->fastWalk && !in->walkT && in->useTickets ? 32 : 64)) return 0;
    const long groups = (in->P + 127) / 128;
    // One launch per wave of slices (BEAGLE_MI355_NO_WALK_FUSION=1, the C++ walk, the T32 walk): about 2 560 workgroups per
    // wave, 2.5 rounds of the 1 024 the chip holds (4 per CU).  Measured with the assembly loop (tools/chunk_sweep.sh): 12 500
    // patterns 129 us at 40 micro-operations per slice vs 145 at 99; flat between 50 and 300 from 25 000 patterns up.
    // All slices in ONE launch (the default at 4 states): a wave of slices is no launch and no chip-wide barrier any more, so
    // slices can be longer — fewer stored slice roots, less polling — while the planner keeps the slices ABOVE the first wave
    // short (planner.h chunkTopOps: near the root few subtrees are left side by side).  12 500 patterns, kernel us per
    // evaluation at 40 / 56 / 72 / 96 / 128 micro-operations per first-wave slice: 130 / 122 / 113 / 114 / 114 with 16 above
    // (135 / 122 / 141 / 127 / 122 with the same length above); 25 000 and more: flat (profiles/r04_experiments.txt).
    // On tickets (round 6) the grid is the first wave alone: 12 500 patterns, kernel us at 40 / 56 / 70 / 96 / 128 / 150 per first-wave
    // slice with 8 above: 136 / 126 / 125 / 112 / 105 / 106 (16 above: 136 / 131 / 122 / 117 / 109 / 111; flags at their best: 108).
    // 6 250 patterns: 76 us at a divisor of 500, 80 at 765, 99 at 1 400; 25 000 and more: flat.  A partitioned instance's slices span one
    // partition's groups each, so the same divisor means fewer workgroups: config E (4 partitions) is best where it was, 73 us at 1 400 against
    // 91 at 765 (tools/r06_ticket_sweep2.sh, profiles/r06_experiments.txt).
    const bool fused = in->fuseWaves && in->fastWalk && !in->walkT;
    static const long ticketDiv = labEnv("BEAGLE_MI355_CHUNK_DIV") ? atol(labEnv("BEAGLE_MI355_CHUNK_DIV")) : 0;
    // (the T32 walk since it runs three workgroups per CU — 768 places, round 6 —: config B, 20 states, evaluations/s at 40 / 56 / 76 / 100 /
    // 130 / 180 micro-operations per slice: 467 / 467 / 475 / 479 / 480 / 455; 76 is what 2 560 gives there, 102 what 1 900 does)
    // (21..64 states, k_walkT64 — 512 places of four tiles, a stage is microseconds long: config C, 61 states, evaluations/s at 8 / 12 / 16 / 24 /
    // 32 / 48 / 64 micro-operations per slice: 340 / 342 / 338 / 332 / 334 / 336 / 334, one walk: 180 — tools/r06_t64_ring.sh)
    if (in->walkT && in->S > 20) return (int)std::min<long>(150, std::max<long>(8, (long)opCount * groups / 2600));
    const long div = in->walkT ? 1900 : !fused ? 2560 : !in->useTickets ? 1400 : ticketDiv > 0 ? ticketDiv : in->partitionCount > 1 ? 1400 : 765;
    return (int)std::min<long>(150, std::max<long>(24, (long)opCount * groups / div));
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
    bool allowVirtual = tuple == BEAGLE_PARTITION_OP_COUNT || parts == 1;
    // (gradient evaluations want the partials of every node the pre-order pass cannot re-evaluate itself: engine_preorder.cpp
    // runPreOperations, walkableDefinition)
    struct StepLimit { mi355::WalkPlanner& pl; ~StepLimit() { pl.stepLimit = 0; } } stepLimitGuard{in->planner};
    if (in->storeAllEvaluations > 0) {
        in->storeAllEvaluations--;
        if (in->gradientVirtual && parts == 1 && tuple == BEAGLE_OP_COUNT) in->planner.stepLimit = in->gradientVirtualSteps;
        else allowVirtual = false;
    }
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
            if (launches && timeThisCall(in)) {
                if (in->eventsUsed == in->events.size()) { hipEvent_t x, y; HIP_TRY(hipEventCreate(&x)); HIP_TRY(hipEventCreate(&y)); in->events.emplace_back(x, y); }
                a = in->events[in->eventsUsed].first; b = in->events[in->eventsUsed].second; in->eventsUsed++;
            }
            int rc = runPlan(in, *in->planner.planned, in->planner.plannedTag, a); if (rc) return rc;
            if (b) { HIP_TRY(hipEventRecord(b, live(in))); in->pendingLaunches++; }
            const double usRun = usSince(t1);
            in->hostRunUs += usRun; in->hostRunHitUs += usRun;
            if (in->hostTrace && usPlan + usRun > 40.0)
                fprintf(stderr, "[mi355] call %ld (replayed, tag %ld): planner %.1f us, run %.1f us (resolved again: %d, folds rebuilt so far %ld)\n", in->hostCalls,
                        in->planner.plannedTag, usPlan, usRun, (int)in->lastResolveMiss, in->statFoldBuilds);
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
    if (timeThisCall(in)) {
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
        double usPlanThis = 0.0;
        { const double us = usSince(t1); usPlanThis = us; in->hostPlanUs += us; if (hit) { in->hostPlanHitUs += us; in->hostHits++; } }
        t1 = Clock::now();
        // with the kernel timer on, ONE HIP-event pair brackets the walk launches of the call (the program upload and the
        // snapshot copies are outside: the events time the pruning kernel, which is what the roofline is about)
        if (!in->planner.planned->prog.empty()) {
            rc = runPlan(in, *in->planner.planned, in->planner.plannedTag, launches == 0 ? e0 : nullptr); if (rc) return rc;
            launches++;
        } else { rc = runPlan(in, *in->planner.planned, in->planner.plannedTag); if (rc) return rc; }
        { const double us = usSince(t1); in->hostRunUs += us; if (hit) in->hostRunHitUs += us;
          if (in->hostTrace && usPlanThis + us > 40.0)
              fprintf(stderr, "[mi355] call %ld (%d ops from %d, cache %s, tag %ld): planner %.1f us, run %.1f us (resolved again: %d, folds rebuilt so far %ld)\n", in->hostCalls, n, begin,
                      hit ? "hit" : "miss", in->planner.plannedTag, usPlanThis, us, (int)in->lastResolveMiss, in->statFoldBuilds); }
        begin += n;
    }
    if (e1 && launches > 0) { HIP_TRY(hipEventRecord(e1, live(in))); in->pendingLaunches += launches; }
    else if (e1) { in->eventsUsed--; in->timedCalls--; }       // nothing was launched: give the (unrecorded) event pair back
    return foldCumulative(in, ops, count, tuple, globalCum);
}


}  // namespace eng
}  // namespace mi355
