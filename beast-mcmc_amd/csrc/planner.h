// planner.h — the walk planner: turns an updatePartials operation list into the micro-operation program the 4-state
// pattern-walk kernel executes (kernels_walk4.hip), and keeps the definitions of "virtual" partials buffers.
//
// Pure host logic: no HIP types, no device pointers — everything is expressed in buffer / matrix / scale INDICES; the
// engine resolves them to addresses (engine_walk.cpp).  That keeps the planner testable without a GPU: tests/native/ holds
// an index-level interpreter of the micro-operations and checks planner output against list-order evaluation.
//
// Reference behaviour this has to preserve (all paths relative to /root/reference):
//   * an operation list is any dependency-ordered list of 7-int tuples {dest, writeScale, readScale, child1, matrix1,
//     child2, matrix2} (src/dr/evomodel/treelikelihood/BeagleTreeLikelihood.java:1266-1299), post-order or the reverse
//     level order of src/dr/evomodel/treedatalikelihood/LikelihoodTreeTraversal.java:133-205; 9-int tuples add
//     {partition, cumulativeScale} (MultiPartitionDataLikelihoodDelegate.java:972-997);
//   * every buffer index is independent storage that keeps the value its last operation gave it
//     (src/dr/evomodel/treedatalikelihood/BufferIndexHelper.java:71-106 flips between two indices per node and expects
//     the unflipped one to survive a rejected proposal).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mi355 {

constexpr int PLAN_MAX_STEPS = 32;       // capacity of a virtual definition (snapshot slots per buffer = 2 * this)
constexpr int PLAN_NONE = -1;

// kinds / scale modes: numerically identical to WK_* / WS_* of kernels.h (static_assert in engine_internal.h)
enum { PK_MEM = 0, PK_TIPS = 1, PK_ACC = 2, PK_H0 = 3, PK_H1 = 4, PK_H2 = 5 };
enum { PS_NONE = 0, PS_READ = 1, PS_WRITE = 2 };

struct MicroOp {
    int storeBuf;        // partials buffer that receives the result, or PLAN_NONE
    int k1, a1;          // first child: kind, and the partials / tip buffer index for PK_MEM / PK_TIPS
    int k2, a2;          // second child (PK_MEM, PK_TIPS or PK_ACC)
    int mat1, mat2;      // matrix slot of each child's branch (caller's index, or a snapshot slot >= matrixCount)
    int scaleIdx, smode; // scale buffer and PS_*
    int hold;            // 0, or 1 + hold slot that ALSO receives the result
};

// A program slice: every pattern group of `partition` walks it.  Slices of the same `wave` are independent of each other
// (one launch); a slice may read what slices of EARLIER waves stored.
// depStart / depCount: the slices whose stored results this one reads (indices into Plan::segs, all of earlier waves), as a
// range of Plan::deps.  tail: micro-operations from the start of this slice to the end of the last slice that (transitively)
// waits for it — the critical path behind it.  A single launch may run ALL slices side by side when every workgroup first
// waits for the slices in its dependency list (same pattern group) and the slices are dispatched by descending tail — an
// order in which every slice comes after the ones it reads (tail(child) > tail(parent)); Plan::launchOrder is that order.
// next: the ONE slice that waits for this one (-1: none — a root of the forest; meaningful when Plan::leaves > 0).  A slice's stored
// result normally has one reader, so the slices form a forest and a launch needs no waiting at all: only the slices without
// dependencies ("leaves") get workgroups, a workgroup that finishes slice s counts itself in at s.next for its pattern group, and
// the one that arrives LAST there carries on with that slice itself — everything it reads was stored by workgroups that counted
// in before it (kernels_walk4.hip: "tickets").
struct PlanSeg { int progStart, progCount, partition, wave, depStart, depCount, tail, next; };

struct Plan {
    std::vector<MicroOp> prog;
    std::vector<PlanSeg> segs;       // sorted by wave
    std::vector<int> deps;           // PlanSeg::depStart / depCount
    std::vector<int> launchOrder;    // permutation of segs: descending tail (critical path first), dependencies before dependants
    std::vector<int> snapPairs;      // (source matrix slot, destination snapshot slot) pairs to copy BEFORE the program runs
    int leaves = 0;                  // > 0: the slices form a forest (every slice has at most one dependant, none is empty) and this many of
                                     // them wait for nothing — the program can run on tickets (PlanSeg::next); 0: dependency flags only
    void clear() { prog.clear(); segs.clear(); deps.clear(); launchOrder.clear(); snapPairs.clear(); leaves = 0; }
};

// Read-mode programs: where the reciprocal scale factors are applied.  A result that is not stored is seen by nobody but the
// micro-operation that consumes it, and a partial is linear in each of its children — so an unstored result may pass the
// factors it owes (its own and those its unstored operands passed on) to its consumer, and only a result that IS stored, that
// ends its slice, or that has gathered `maxMembers` of them pays: one multiplication by the product of the members'
// reciprocals.  foldScaleFactors says, per micro-operation of plan.prog, which scale buffers it pays for:
// members[payStart[i] .. payStart[i + 1]) (none: it multiplies by nothing; one: that buffer's own reciprocals; several: a fold the
// engine builds, engine_walk.cpp).  false: the program rescales in write mode somewhere (nothing is folded then).
struct FoldMap { std::vector<int> payStart, members; };
bool foldScaleFactors(const Plan& plan, int maxMembers, FoldMap& out);

// Definition of a virtual buffer: one step per internal node of a small all-compact-tip subtree, in post-order (a step's
// sub-steps precede it; the last step is the buffer's own node).  An operand of a step is a compact tip or an earlier
// step; `need` = hold slots its evaluation takes (Sethi-Ullman: two internal operands -> the costlier one first, parked
// in a hold slot while the other one is evaluated), at most one less than the walk has, so that a parent can still park
// the definition's own result.
enum { VT_CHERRY = 1, VT_EXTEND = 2, VT_JOIN = 3 };
struct VirtStep {
    int type;               // VT_CHERRY: tipA, tipB;  VT_EXTEND: subA, tipB;  VT_JOIN: subA, subB
    int tipA, tipB;         // leaf buffers (or -1): compact tip states, or (memA / memB) partials the caller uploaded for a tip
    bool memA, memB;        // (no initialisers: a definition's unused steps are never looked at, and never copied — VirtDef below)
    int subA, subB;         // earlier steps of the same definition (or -1)
    int scaleIdx;           // this node's scale buffer, or PLAN_NONE
    int originA, originB;   // matrix slots the snapshots of operand A's / B's branch were last copied FROM
    int need;               // hold slots the evaluation of this step takes
};
// (1.5 KB with all 32 steps; a list of 1000 operations makes, keeps and copies a thousand of them, most of one to three steps:
// copies take the steps in use only — the planner's time on a list it has not seen: 87 -> 52 us with that, profiles/r05_experiments.txt 12)
struct VirtDef {
    bool on = false;
    bool chainOnly = true;  // evaluates without a hold slot (need of the last step == 0)
    int nSteps = 0;
    int stamp = -1;         // planner stamp of the list that created or last re-confirmed it
    // the op that defined it, for the steady-state path (an MCMC chain re-issues the same op on the same buffers every
    // other evaluation): a definition whose op and children are unchanged is re-confirmed, not rebuilt
    int version = 0;
    int sigC1 = -1, sigM1 = -1, sigC2 = -1, sigM2 = -1, sigScale = -2;
    bool sigTip1 = false, sigTip2 = false, sigMem1 = false, sigMem2 = false, fresh1 = false, fresh2 = false;   // sigTip: the child is a leaf
    int childVer1 = -1, childVer2 = -1;
    long cacheTag = 0;      // plan-cache entry that last wrote or confirmed this definition (0: none) — see WalkPlanner::replay
    VirtStep steps[PLAN_MAX_STEPS];      // (copies take steps[0 .. nSteps) only)
    VirtDef() {}
    VirtDef(const VirtDef& o) { copyFrom(o); }
    VirtDef& operator=(const VirtDef& o) { if (this != &o) copyFrom(o); return *this; }
private:
    void copyFrom(const VirtDef& o);
};

inline void VirtDef::copyFrom(const VirtDef& o) {
    on = o.on; chainOnly = o.chainOnly; nSteps = o.nSteps; stamp = o.stamp; version = o.version;
    sigC1 = o.sigC1; sigM1 = o.sigM1; sigC2 = o.sigC2; sigM2 = o.sigM2; sigScale = o.sigScale;
    sigTip1 = o.sigTip1; sigTip2 = o.sigTip2; sigMem1 = o.sigMem1; sigMem2 = o.sigMem2; fresh1 = o.fresh1; fresh2 = o.fresh2;
    childVer1 = o.childVer1; childVer2 = o.childVer2; cacheTag = o.cacheTag;
    for (int s = 0; s < o.nSteps; s++) steps[s] = o.steps[s];
}

class WalkPlanner {
public:
    // holdSlots: per-thread slots a value can wait in while its sibling's subtree is evaluated (2 or 3; kernels.h)
    void init(int partialsCount, int tipCount, int matrixCount, int scaleCount, int maxVirtSteps, bool virtualEnabled, int holdSlots = 2);

    // maintained by the engine through setCompactTip: buffer holds compact tip states (index < tipCount and setTipStates was
    // the last setter).  compactEpoch counts the changes (a cached plan is only valid for the flags it was made with).
    std::vector<char> compactTip;
    long compactEpoch = 0;
    void setCompactTip(int buf, bool on) { const char v = on ? 1 : 0; if (compactTip[buf] != v) { compactTip[buf] = v; compactEpoch++; } }
    // ... or holds PARTIALS the caller uploaded for a tip (setTipPartials / setPartials on a tip index: ambiguity codes as
    // partials, sequence-error models — BeagleTreeLikelihood.java:497-509): data no operation computes, so a definition may
    // read it as a leaf too (32 C bytes per pattern from memory instead of a state byte)
    std::vector<char> leafPartials;
    void setLeafPartials(int buf, bool on) { const char v = on ? 1 : 0; if (leafPartials[buf] != v) { leafPartials[buf] = v; compactEpoch++; } }

    // A definition belongs to a (buffer, partition) pair — a partitioned instance updates the pattern ranges of one buffer
    // independently (MultiPartitionDataLikelihoodDelegate.java:972-997: every partition flips its own buffer indices) — and
    // is identified by its KEY = buffer * partitionCount + partition; with one partition the key is the buffer index.
    void setPartitionCount(int parts);       // forgets every definition (the engine materialises them first)
    int partitionCount() const { return keyParts_; }
    int key(int buf, int part) const { return buf * keyParts_ + part; }
    int bufferOf(int key) const { return key / keyParts_; }
    int partitionOf(int key) const { return key % keyParts_; }
    bool isVirtualKey(int key) const { return virt_[key].on; }
    bool isVirtual(int buf) const { for (int k = 0; k < keyParts_; k++) if (virt_[(std::size_t)buf * keyParts_ + k].on) return true; return false; }
    void keysOf(int buf, std::vector<int>& out) const { for (int k = 0; k < keyParts_; k++) if (virt_[(std::size_t)buf * keyParts_ + k].on) out.push_back(buf * keyParts_ + k); }
    const VirtDef& definition(int key) const { return virt_[key]; }
    bool virtualEnabled() const { return enabled_; }
    int snapSlot(int key, int step, int which) const { return matrixCount_ + key * 2 * maxSteps_ + 2 * step + which; }
    int matrixSlots() const { return matrixCount_ + (enabled_ ? partialsCount_ * keyParts_ * 2 * maxSteps_ : 0); }

    const std::vector<int>& tipUsers(int tip) const { return tipUsers_[tip]; }       // KEYS of the definitions that read this tip
    const std::vector<int>& scaleUsers(int idx) const { return scaleUsers_[idx]; }   // ... this scale buffer
    void clearVirtualKey(int key);       // forget the definition (that range of the buffer is about to get real data)
    void clearVirtual(int buf) { for (int k = 0; k < keyParts_; k++) clearVirtualKey(buf * keyParts_ + k); }
    // Define `buf` as the cherry node(tipA over matrix mA, tipB over matrix mB) [read-mode scale buffer scaleIdx or PLAN_NONE]
    // (the level-scheduled T32 path, engine_levels.cpp runOperationsLevels: one partition); appends the matrix snapshot copies to snapPairs.
    bool defineCherry(int buf, int tipA, int mA, int tipB, int mB, int scaleIdx, std::vector<int>& snapPairs);

    // Length of the longest prefix of ops[begin..count) that can run as one walk: no buffer (or scale buffer) is written
    // twice, written after an earlier op of the prefix read it, or read through a scale index another op writes.
    int hazardFreePrefix(const int* ops, int begin, int count, int tuple, int partitionCount);

    // Definitions (KEYS) that must get real data before this (hazard-free) list runs: those that read a scale buffer the list
    // rewrites (unless the list redefines them anyway), and a virtual destination that is its own child.
    void mustMaterializeBefore(const int* ops, int count, int tuple, std::vector<int>& out);

    // Plan a hazard-free list.  Indices must have been range-checked by the caller; partitionCount must be the planner's.
    // `allowVirtual`: destinations may become virtual.  Returns 0, or a BEAGLE error code.
    // `chunkOps` > 0 (few pattern groups, so a launch cannot fill the chip with one walk per group): the forest is cut
    // into independent subtrees of about that many micro-operations, run side by side, wave after wave.
    int plan(const int* ops, int count, int tuple, int partitionCount, bool allowVirtual, Plan& out, int chunkOps = 0);

    // The steady state of a chain: the SAME closed list as one planned before (byte for byte, same compact-tip flags).  Returns
    // true and leaves the planner as plan() would — `planned` is the kept program — after one memcmp and one tag comparison per
    // operation.  `simple` (out): the list has no rescaling operation, no in-place update and no tip as a destination, so
    // nothing has to be materialised before it and nothing accumulated after it: the engine may skip its per-operation
    // checks too (they passed when the entry was made).  With `simple` given, a list that is not simple is left alone (false).
    bool replayCached(const int* ops, int count, int tuple, int partitionCount, bool allowVirtual, int chunkOps, bool* simple);

    // Program that gives every definition of `keys` its real partials (one slice per partition); the definitions are dropped.
    void planMaterialize(const std::vector<int>& keys, Plan& out);

    // statistics of the last plan() (bench / tests)
    int lastStored = 0, lastMemReads = 0, lastHolds = 0, lastWaves = 0;
    // chunkTopOps > 0: every wave above the first peels subtrees of at most this many micro-operations (the engine sets it once,
    // when ALL slices of a program run in one launch and a wave is not a launch: near the root few subtrees are left side by
    // side, so long slices there are a long serial tail on a few CUs; short ones keep more of them in flight)
    int chunkTopOps = 0;
    // How many slices the chip runs side by side (workgroup slots / pattern groups of a slice; the engine sets it): linkSlices
    // orders a one-launch program by simulating a greedy critical-path schedule on that many machines, so that a slice is
    // dispatched about when the slices it waits for are done — dispatched earlier it would only hold its workgroup slots
    // while it polls.  <= 0: plain descending-tail order.
    double launchMachines = 0.0;
    // stepLimit > 0: definitions made by the next plan() hold at most this many steps (a gradient chain: the pre-order walk
    // re-evaluates an unstored operand from its tips every time it meets it, kernels_preorder4.hip — tip-tip nodes and a tip-tip node
    // under one more tip, nothing longer); part of a cached plan's identity.  The snapshot slots keep the instance's own spacing.
    int stepLimit = 0;
    int stepCap() const { return stepLimit > 0 && stepLimit < maxSteps_ ? stepLimit : maxSteps_; }
    long cacheHits = 0;                  // plans served from the cache below
    long replayInPlace = 0;              // definitions a replay took over from another entry with their leaves unchanged (user lists edited in place)
    bool cacheEnabled = true;
    // what the last plan() produced: `out`, or the cache's copy (no copy is made on a hit).  plannedTag identifies the
    // cache entry (0: not cached) so that the engine can keep what it derives from the plan.
    const Plan* planned = nullptr;
    long plannedTag = 0;

private:
    struct OpInfo {
        int dest, wS, rS, c1, m1, c2, m2, part;
        bool tip1, tip2, leaf1, leaf2, virtDest;     // tip: compact states;  leaf: that, or uploaded tip partials
        int need;           // hold slots the evaluation needs (-1: not computed yet)
        int size;           // real micro-ops below (ordering heuristic)
        bool emitted;
    };
    bool buildVirtual(int X, int c1, bool leaf1, bool mem1, int m1, int c2, bool leaf2, bool mem2, int m2, int scaleIdx, std::vector<int>& snapPairs);   // X, and a non-leaf child: KEYS
    void registerVirtual(int X);
    // emission
    void emitReal(int root, unsigned freeMask, Plan& out);
    void emitVirtualStep(int buf, int idx, unsigned freeMask, bool writeMode, Plan& out);
    void emitVirtual(int buf, unsigned freeMask, bool writeMode, Plan& out);
    int virtNeed(int buf) const { return virt_[buf].nSteps ? virt_[buf].steps[virt_[buf].nSteps - 1].need : 0; }
    void linkSlices(Plan& out);          // PlanSeg::dep*, tail and Plan::launchOrder
    std::vector<int> storedBy_, storedStamp_;          // per (buffer, partition): the slice of the current plan that stores it
    int linkStamp_ = 0;

    int partialsCount_ = 0, tipCount_ = 0, matrixCount_ = 0, scaleCount_ = 0, maxSteps_ = 6, keyParts_ = 1;
    bool enabled_ = false;
    std::vector<VirtDef> virt_;
    std::vector<int> pairScratch_;                     // buildVirtual's snapshot pairs before they are known to be wanted
    std::vector<long> tagOf_;                          // per key: -1 not virtual, 0 virtual, > 0 virtual and written by that cache entry
    long tagEpoch_ = 0;                                // counts the writes to tagOf_: an entry replayed at this count and asked for again at
                                                       // the same count finds every key as it left it (CacheEntry::cleanAtEpoch) — the chain's
                                                       // steady state, two lists alternating, costs no pass over the list at all
                                                       // (a compact mirror of on / cacheTag: replaying a plan touches 8 bytes per op)
    std::vector<std::vector<int>> tipUsers_, scaleUsers_;
    int virtVersion_ = 0;
    int stamp_ = 0;
    // per-list scratch (stamped, so nothing is cleared between calls)
    std::vector<int> wStamp_, rStamp_, wOp_;          // per (buffer, partition)
    std::vector<int> sWStamp_, sRStamp_, sDone_;      // per scale buffer: written / read in this list, write emitted
    std::vector<OpInfo> info_;
    std::vector<int> prod1_, prod2_;                   // op of this list that produced each child (or -1)
    unsigned allSlots_ = 3u;                           // mask of the hold slots
    int parts_ = 1;

    // Plans of CLOSED lists (every child is a compact tip or the destination of an earlier operation of the same list:
    // a full evaluation, what BEAST issues whenever a model parameter changes — MarkovChain.java:207-263 with all nodes
    // dirty) depend on nothing but the list itself and the tips' compact flags, so they are kept and replayed: the chain
    // alternates between two such lists (BufferIndexHelper.java:71-106).  Partial updates are planned every time.
    struct CacheEntry {
        bool valid = false;
        long tag = 0;
        int count = 0, tuple = 0, parts = 0, chunkOps = 0, stepLimit = 0;
        bool allowVirtual = false;
        std::vector<int> ops;
        long tipEpoch = -1;                            // compactEpoch the plan was made under
        bool simple = false;                           // see replayCached
        Plan plan;
        std::vector<VirtDef> defs;                     // per op: the definition its destination ends up with (on = false: real)
        std::vector<char> defOn;                       // defs[k].on
        int stored = 0, memReads = 0, holds = 0, waves = 0;
        mutable long cleanAtEpoch = -1;                // tagEpoch_ right after the last replay of this entry (-1: never replayed since it was filled)
    };
    // (8: a chain under DYNAMIC rescaling cycles through 2 buffer-flip states x 2 scale-buffer sets of its read-mode list and the
    // same four of the list that recomputes the factors every 100th evaluation — with 4 ways every such evaluation evicted a
    // read-mode entry and the next evaluations planned, resolved and uploaded from scratch: 1.4 ms each on the 12 500-pattern shard)
    static constexpr int CACHE_WAYS = 8;
    CacheEntry cache_[CACHE_WAYS];
    int cacheNext_ = 0;
    long cacheTagNext_ = 0;
    void replay(const CacheEntry& e, const int* ops);
    CacheEntry* findCached(const int* ops, int count, int tuple, int parts, bool allowVirtual, int chunkOps);
};

}  // namespace mi355
