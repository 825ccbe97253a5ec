// planner.cpp — see planner.h.  Pure host logic (compiled into the engine by hipcc and into the planner test by g++).
#include "planner.h"

#include <algorithm>
#include <cstring>

namespace mi355 {

namespace {
constexpr int OP_NONE = -1;                 // BEAGLE_OP_NONE
enum { CL_TIPS = 0, CL_MEM = 1, CL_VIRT = 2, CL_REAL = 3 };
inline int popcount2(unsigned m) { return (int)(m & 1u) + (int)((m >> 1) & 1u) + (int)((m >> 2) & 1u); }      // free hold slots (up to 3)
inline int lowestSlot(unsigned m) { return (m & 1u) ? 0 : (m & 2u) ? 1 : 2; }
}  // namespace

void WalkPlanner::init(int partialsCount, int tipCount, int matrixCount, int scaleCount, int maxVirtSteps, bool virtualEnabled, int holdSlots) {
    allSlots_ = (1u << std::max(0, std::min(3, holdSlots))) - 1u;          // (0: no hold slots at all — the 21..64-state walk)
    partialsCount_ = partialsCount; tipCount_ = tipCount; matrixCount_ = matrixCount; scaleCount_ = std::max(1, scaleCount);
    maxSteps_ = std::max(1, std::min(maxVirtSteps, PLAN_MAX_STEPS));
    enabled_ = virtualEnabled;
    keyParts_ = 1;
    virt_.assign(partialsCount, VirtDef());
    tagOf_.assign(partialsCount, -1); tagEpoch_++;
    tipUsers_.assign(partialsCount, std::vector<int>());
    scaleUsers_.assign(scaleCount_, std::vector<int>());
    compactTip.assign(partialsCount, 0);
    leafPartials.assign(partialsCount, 0);
    wStamp_.assign(partialsCount, 0); rStamp_.assign(partialsCount, 0); wOp_.assign(partialsCount, 0);
    sWStamp_.assign(scaleCount_, 0); sRStamp_.assign(scaleCount_, 0); sDone_.assign(scaleCount_, 0);
    stamp_ = 0; virtVersion_ = 0;
}

void WalkPlanner::setPartitionCount(int parts) {
    keyParts_ = std::max(1, parts);
    virt_.assign((size_t)partialsCount_ * keyParts_, VirtDef());
    tagOf_.assign((size_t)partialsCount_ * keyParts_, -1); tagEpoch_++;
    for (auto& u : tipUsers_) u.clear();
    for (auto& u : scaleUsers_) u.clear();
    for (CacheEntry& e : cache_) e.valid = false;
}

void WalkPlanner::clearVirtualKey(int X) {
    VirtDef& v = virt_[X];
    if (!v.on) return;
    auto drop = [X](std::vector<int>& u) { u.erase(std::remove(u.begin(), u.end(), X), u.end()); };
    for (int s = 0; s < v.nSteps; s++) {
        const VirtStep& h = v.steps[s];
        if (h.tipA >= 0) drop(tipUsers_[h.tipA]);
        if (h.tipB >= 0) drop(tipUsers_[h.tipB]);
        if (h.scaleIdx >= 0) drop(scaleUsers_[h.scaleIdx]);
    }
    v.on = false;
    tagOf_[X] = -1; tagEpoch_++;
}

void WalkPlanner::registerVirtual(int X) {
    const VirtDef& v = virt_[X];
    auto add = [X](std::vector<int>& u) { if (std::find(u.begin(), u.end(), X) == u.end()) u.push_back(X); };
    for (int s = 0; s < v.nSteps; s++) {
        const VirtStep& h = v.steps[s];
        if (h.tipA >= 0) add(tipUsers_[h.tipA]);
        if (h.tipB >= 0) add(tipUsers_[h.tipB]);
        if (h.scaleIdx >= 0) add(scaleUsers_[h.scaleIdx]);
    }
}

// Try to define buffer X = node(child1 over matrix m1, child2 over matrix m2, scale).  Children are compact tips or
// virtual buffers.  Appends (source, destination) matrix-copy pairs.  false: too many steps, or its evaluation would need
// more hold slots than a definition may take.
bool WalkPlanner::buildVirtual(int X, int c1, bool tip1, bool mem1, int m1, int c2, bool tip2, bool mem2, int m2, int scaleIdx, std::vector<int>& snapPairs) {
    // (tip1 / tip2: the child is a LEAF — compact states, or with mem1 / mem2 uploaded tip partials)
    VirtDef nv;
    nv.on = true; nv.stamp = stamp_; nv.nSteps = 0; nv.chainOnly = true;
    std::vector<int>& pairs = pairScratch_;
    pairs.clear();
    const int maxNeed = popcount2(allSlots_) - 1;
    const int cap = stepCap();
    // copies the steps of srcBuf's definition behind what nv holds; returns the index of its last step, -1: no room
    auto append = [&](int srcBuf) -> int {
        const VirtDef& src = virt_[srcBuf];
        const int base = nv.nSteps;
        for (int s = 0; s < src.nSteps; s++) {
            if (nv.nSteps >= cap) return -1;
            VirtStep h = src.steps[s];
            // a child defined in THIS list has its slots written by the same snapshot launch: copy from its origins
            const int fromA = src.stamp == stamp_ ? h.originA : snapSlot(srcBuf, s, 0);
            const int fromB = src.stamp == stamp_ ? h.originB : snapSlot(srcBuf, s, 1);
            h.originA = fromA; h.originB = fromB;
            if (h.subA >= 0) h.subA += base;
            if (h.subB >= 0) h.subB += base;
            pairs.push_back(fromA); pairs.push_back(snapSlot(X, nv.nSteps, 0));
            pairs.push_back(fromB); pairs.push_back(snapSlot(X, nv.nSteps, 1));
            nv.steps[nv.nSteps++] = h;
        }
        return nv.nSteps - 1;
    };
    VirtStep last;
    last.scaleIdx = scaleIdx; last.tipA = -1; last.tipB = -1; last.subA = -1; last.subB = -1; last.need = 0; last.memA = last.memB = false;
    if (tip1 && tip2) {
        last.type = VT_CHERRY; last.tipA = c1; last.tipB = c2; last.originA = m1; last.originB = m2; last.memA = mem1; last.memB = mem2;
    } else if (tip1 != tip2) {
        const int vb = tip1 ? c2 : c1, t = tip1 ? c1 : c2, mv = tip1 ? m2 : m1, mt = tip1 ? m1 : m2;
        if (!virt_[vb].on) return false;
        const int r = append(vb);
        if (r < 0) return false;
        last.type = VT_EXTEND; last.subA = r; last.tipB = t; last.originA = mv; last.originB = mt; last.memB = tip1 ? mem1 : mem2;
        last.need = nv.steps[r].need;
    } else {
        if (!virt_[c1].on || !virt_[c2].on) return false;
        const int ra = append(c1);
        if (ra < 0) return false;
        const int rb = append(c2);
        if (rb < 0) return false;
        last.type = VT_JOIN; last.subA = ra; last.subB = rb; last.originA = m1; last.originB = m2;
        const int na = nv.steps[ra].need, nb = nv.steps[rb].need;
        last.need = std::min(std::max(na, 1 + nb), std::max(nb, 1 + na));
        if (last.need > maxNeed) return false;
    }
    if (nv.nSteps >= cap) return false;
    pairs.push_back(last.originA); pairs.push_back(snapSlot(X, nv.nSteps, 0));
    pairs.push_back(last.originB); pairs.push_back(snapSlot(X, nv.nSteps, 1));
    nv.steps[nv.nSteps++] = last;
    nv.chainOnly = last.need == 0;
    virt_[X] = nv;
    tagOf_[X] = 0; tagEpoch_++;
    registerVirtual(X);
    snapPairs.insert(snapPairs.end(), pairs.begin(), pairs.end());
    return true;
}

bool WalkPlanner::defineCherry(int X, int tipA, int mA, int tipB, int mB, int scaleIdx, std::vector<int>& snapPairs) {
    if (!enabled_ || !compactTip[tipA] || !compactTip[tipB]) return false;
    stamp_++;
    if (virt_[X].on) clearVirtualKey(X);
    if (!buildVirtual(X, tipA, true, false, mA, tipB, true, false, mB, scaleIdx, snapPairs)) return false;
    VirtDef& nv = virt_[X];
    nv.version = ++virtVersion_;
    nv.sigC1 = tipA; nv.sigM1 = mA; nv.sigC2 = tipB; nv.sigM2 = mB; nv.sigScale = scaleIdx; nv.sigTip1 = nv.sigTip2 = true;
    return true;
}

int WalkPlanner::hazardFreePrefix(const int* ops, int begin, int count, int tuple, int parts) {
    const size_t nKeys = (size_t)partialsCount_ * parts, nS = (size_t)scaleCount_ * parts;
    if (wStamp_.size() < nKeys) { wStamp_.assign(nKeys, 0); rStamp_.assign(nKeys, 0); wOp_.assign(nKeys, 0); }
    if (sWStamp_.size() < nS) { sWStamp_.assign(nS, 0); sRStamp_.assign(nS, 0); sDone_.assign(nS, 0); }
    stamp_++;
    int k = begin;
    for (; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        const int part = tuple > 7 ? op[7] : 0;
        const size_t kd = (size_t)op[0] * parts + part, k1 = (size_t)op[3] * parts + part, k2 = (size_t)op[5] * parts + part;
        bool hazard = wStamp_[kd] == stamp_ || rStamp_[kd] == stamp_;
        if (op[1] != OP_NONE) { const size_t s = (size_t)op[1] * parts + part; hazard = hazard || sWStamp_[s] == stamp_ || sRStamp_[s] == stamp_; }
        if (op[2] != OP_NONE && op[1] == OP_NONE) { const size_t s = (size_t)op[2] * parts + part; hazard = hazard || sWStamp_[s] == stamp_; }
        if (hazard && k > begin) break;
        rStamp_[k1] = stamp_; rStamp_[k2] = stamp_; wStamp_[kd] = stamp_;
        if (op[1] != OP_NONE) sWStamp_[(size_t)op[1] * parts + part] = stamp_;
        else if (op[2] != OP_NONE) sRStamp_[(size_t)op[2] * parts + part] = stamp_;
    }
    return k - begin;
}

void WalkPlanner::mustMaterializeBefore(const int* ops, int count, int tuple, std::vector<int>& out) {
    if (!enabled_) return;
    stamp_++;
    const size_t nKeys = (size_t)partialsCount_ * keyParts_;
    if (wStamp_.size() < nKeys) { wStamp_.assign(nKeys, 0); rStamp_.assign(nKeys, 0); wOp_.assign(nKeys, 0); }
    auto keyOf = [&](const int* op) { return (size_t)op[0] * keyParts_ + (tuple > 7 ? op[7] : 0); };
    for (int k = 0; k < count; k++) wStamp_[keyOf(ops + (size_t)k * tuple)] = stamp_;
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        if ((op[3] == op[0] || op[5] == op[0]) && virt_[keyOf(op)].on) out.push_back((int)keyOf(op));     // in-place update of a virtual buffer
        const int wS = op[1];
        if (wS == OP_NONE || wS < 0 || wS >= scaleCount_) continue;
        for (int u : scaleUsers_[wS])
            if (wStamp_[u] != stamp_) out.push_back(u);          // a destination of this list is redefined anyway
    }
}

// ---- emission ----------------------------------------------------------------------------------------------------
namespace {
struct Child { int cls, buf, mat, prod, need, size, vkey; };      // vkey: definition key of (buf, the op's partition)
inline MicroOp blankOp() {
    MicroOp m; m.storeBuf = PLAN_NONE; m.k1 = PK_MEM; m.a1 = 0; m.k2 = PK_MEM; m.a2 = 0; m.mat1 = 0; m.mat2 = 0;
    m.scaleIdx = PLAN_NONE; m.smode = PS_NONE; m.hold = 0;
    return m;
}
inline void setLeaf(MicroOp& m, int which, const Child& c) {
    const int kind = c.cls == CL_TIPS ? PK_TIPS : PK_MEM;
    if (which == 0) { m.k1 = kind; m.a1 = c.buf; m.mat1 = c.mat; } else { m.k2 = kind; m.a2 = c.buf; m.mat2 = c.mat; }
}
}  // namespace

void WalkPlanner::emitVirtualStep(int buf, int idx, unsigned freeMask, bool writeMode, Plan& out) {
    const VirtDef& v = virt_[buf];
    const VirtStep& st = v.steps[idx];
    MicroOp m = blankOp();
    if (st.type == VT_CHERRY) {
        // a leaf read from memory goes first (the kernels prefetch the first child's partials; kernels.h)
        const bool swap = st.memB && !st.memA;
        const int tA = swap ? st.tipB : st.tipA, tB = swap ? st.tipA : st.tipB;
        const bool mA = swap ? st.memB : st.memA, mB = swap ? st.memA : st.memB;
        m.k1 = mA ? PK_MEM : PK_TIPS; m.a1 = tA; m.mat1 = snapSlot(buf, idx, swap ? 1 : 0);
        m.k2 = mB ? PK_MEM : PK_TIPS; m.a2 = tB; m.mat2 = snapSlot(buf, idx, swap ? 0 : 1);
        lastMemReads += (mA ? 1 : 0) + (mB ? 1 : 0);
    } else if (st.type == VT_EXTEND) {
        emitVirtualStep(buf, st.subA, freeMask, writeMode, out);
        m.k1 = st.memB ? PK_MEM : PK_TIPS; m.a1 = st.tipB; m.mat1 = snapSlot(buf, idx, 1);
        m.k2 = PK_ACC; m.mat2 = snapSlot(buf, idx, 0);
        if (st.memB) lastMemReads++;
    } else {   // VT_JOIN: the operand that needs more hold slots first, parked while the other one is evaluated
        const int F = popcount2(freeMask);
        const int na = v.steps[st.subA].need, nb = v.steps[st.subB].need;
        const bool aOK = na <= F && 1 + nb <= F, bOK = nb <= F && 1 + na <= F;
        const bool aFirst = aOK && (!bOK || na >= nb);            // (one of them holds: need <= F by construction)
        const int first = aFirst ? st.subA : st.subB, second = aFirst ? st.subB : st.subA;
        emitVirtualStep(buf, first, freeMask, writeMode, out);
        const int h = lowestSlot(freeMask);
        out.prog.back().hold = h + 1;
        lastHolds++;
        emitVirtualStep(buf, second, freeMask & ~(1u << h), writeMode, out);
        m.k1 = PK_H0 + h; m.mat1 = snapSlot(buf, idx, aFirst ? 0 : 1);
        m.k2 = PK_ACC; m.mat2 = snapSlot(buf, idx, aFirst ? 1 : 0);
    }
    if (st.scaleIdx >= 0) {
        m.scaleIdx = st.scaleIdx;
        const size_t sk = (size_t)st.scaleIdx * parts_ + partitionOf(buf);      // scale-buffer use is tracked per partition
        const bool w = writeMode && sWStamp_[sk] == stamp_;
        m.smode = w ? PS_WRITE : PS_READ;
        if (w) sDone_[sk] = stamp_;
    }
    out.prog.push_back(m);
}

void WalkPlanner::emitVirtual(int buf, unsigned freeMask, bool writeMode, Plan& out) {
    emitVirtualStep(buf, virt_[buf].nSteps - 1, freeMask, writeMode, out);
}

// Emission of the real op `root` and everything below it that this walk evaluates.  Iterative (explicit frame stack on the
// heap): the dependency depth of an operation list is unbounded — a 5000-tip ladder tree is 4999 frames deep — and under
// BEAST this runs on a JVM thread whose native stack is 1 MiB or less (-Xss).
void WalkPlanner::emitReal(int root, unsigned rootMask, Plan& out) {
    struct Frame { int j; unsigned freeMask; int phase; Child ch[2]; int first; bool hold; MicroOp m; };
    std::vector<Frame> st;
    auto push = [&](int j, unsigned fm) { Frame f; f.j = j; f.freeMask = fm; f.phase = 0; f.first = 0; f.hold = false; f.m = blankOp(); st.push_back(f); };
    push(root, rootMask);
    while (!st.empty()) {
        const size_t top = st.size() - 1;                 // (a push invalidates references: every path that pushes `continue`s)
        OpInfo& o = info_[st[top].j];
        if (st[top].phase == 0) {
            Frame& f = st[top];
            for (int w = 0; w < 2; w++) {
                Child& c = f.ch[w];
                c.buf = w ? o.c2 : o.c1; c.mat = w ? o.m2 : o.m1;
                const bool tip = w ? o.tip2 : o.tip1;
                c.prod = -1; c.need = 0; c.size = 0; c.vkey = key(c.buf, o.part);
                if (tip) { c.cls = CL_TIPS; continue; }
                const int prod = w ? prod2_[f.j] : prod1_[f.j];
                if (prod >= 0) {
                    const OpInfo& p = info_[prod];
                    if (p.virtDest) { c.cls = CL_VIRT; c.need = virtNeed(c.vkey); c.size = 0; }
                    else if (p.emitted) c.cls = CL_MEM;
                    else { c.cls = CL_REAL; c.prod = prod; c.need = p.need; c.size = p.size; }
                } else if (virt_[c.vkey].on) { c.cls = CL_VIRT; c.need = virtNeed(c.vkey); }
                else c.cls = CL_MEM;
            }
            const bool e0 = f.ch[0].cls >= CL_VIRT, e1 = f.ch[1].cls >= CL_VIRT;
            if (!e0 && !e1) {
                const int firstLeaf = (f.ch[0].cls == CL_TIPS && f.ch[1].cls == CL_MEM) ? 1 : 0;      // a child in memory goes first
                setLeaf(f.m, 0, f.ch[firstLeaf]); setLeaf(f.m, 1, f.ch[1 - firstLeaf]);
                if (f.ch[0].cls == CL_MEM) lastMemReads++;
                if (f.ch[1].cls == CL_MEM) lastMemReads++;
                f.phase = 9;
            } else if (e0 != e1) {
                const Child e = e0 ? f.ch[0] : f.ch[1];
                const Child l = e0 ? f.ch[1] : f.ch[0];
                setLeaf(f.m, 0, l);
                if (l.cls == CL_MEM) lastMemReads++;
                f.m.k2 = PK_ACC; f.m.mat2 = e.mat;
                f.phase = 9;
                if (e.cls == CL_VIRT) emitVirtual(e.vkey, f.freeMask, true, out);
                else { push(e.prod, f.freeMask); continue; }
            } else {
                const int F = popcount2(f.freeMask);
                auto holdOK = [&](const Child& a, const Child& b) { return a.need <= F && 1 + b.need <= F; };
                auto plainOK = [&](const Child& a, const Child& b) { return a.cls == CL_REAL && a.need <= F && b.need <= F; };
                const Child* ch = f.ch;
                const bool h01 = holdOK(ch[0], ch[1]), h10 = holdOK(ch[1], ch[0]);
                if (h01 || h10) {
                    f.hold = true;
                    if (h01 && h10) f.first = (ch[1].need > ch[0].need || (ch[1].need == ch[0].need && ch[1].size > ch[0].size)) ? 1 : 0;
                    else f.first = h01 ? 0 : 1;
                } else {
                    const bool p01 = plainOK(ch[0], ch[1]), p10 = plainOK(ch[1], ch[0]);
                    if (p01 && p10) f.first = ch[1].size > ch[0].size ? 1 : 0;
                    else if (p01 || p10) f.first = p01 ? 0 : 1;      // by construction one of them holds (planner.h: need <= 2) ...
                    // ... unless the walk has no hold slots (21..64 states): a real child first if there is one (it is stored anyway),
                    // otherwise a definition — evaluated and STORED for once (it stays a definition: the data is what it evaluates to)
                    else f.first = ch[0].cls == CL_REAL ? 0 : ch[1].cls == CL_REAL ? 1 : 0;
                }
                f.phase = 1;
                const Child a = ch[f.first];
                if (a.cls == CL_VIRT) {
                    emitVirtual(a.vkey, f.freeMask, true, out);
                    if (!f.hold) { out.prog.back().storeBuf = a.buf; lastStored++; }      // (read back as PK_MEM below)
                }
                else { push(a.prod, f.freeMask); continue; }
            }
        }
        if (st[top].phase == 1) {                         // the first of two evaluated children is done
            Frame& f = st[top];
            const Child a = f.ch[f.first], b = f.ch[1 - f.first];
            unsigned mask2 = f.freeMask;
            if (f.hold) {
                const int h = lowestSlot(f.freeMask);
                out.prog.back().hold = h + 1;
                lastHolds++;
                mask2 = f.freeMask & ~(1u << h);
                f.m.k1 = PK_H0 + h; f.m.mat1 = a.mat;
            } else {
                f.m.k1 = PK_MEM; f.m.a1 = a.buf; f.m.mat1 = a.mat;
                lastMemReads++;
            }
            f.m.k2 = PK_ACC; f.m.mat2 = b.mat;
            f.phase = 9;
            if (b.cls == CL_VIRT) emitVirtual(b.vkey, mask2, true, out);
            else { push(b.prod, mask2); continue; }
        }
        {                                                 // phase 9: the node itself
            Frame& f = st[top];
            MicroOp& m = f.m;
            m.storeBuf = o.dest;
            if (o.wS != OP_NONE) { m.scaleIdx = o.wS; m.smode = PS_WRITE; sDone_[(size_t)o.wS * parts_ + o.part] = stamp_; }
            else if (o.rS != OP_NONE) { m.scaleIdx = o.rS; m.smode = PS_READ; }
            out.prog.push_back(m);
            o.emitted = true;
            lastStored++;
            st.pop_back();
        }
    }
}

int WalkPlanner::plan(const int* ops, int count, int tuple, int parts, bool allowVirtual, Plan& out, int chunkOps) {
    out.clear();
    planned = &out; plannedTag = 0;
    lastStored = lastMemReads = lastHolds = 0;
    if (count <= 0) return 0;
    parts_ = parts;
    const size_t nKeys = (size_t)partialsCount_ * parts, nS = (size_t)scaleCount_ * parts;
    if (wStamp_.size() < nKeys) { wStamp_.assign(nKeys, 0); rStamp_.assign(nKeys, 0); wOp_.assign(nKeys, 0); }
    if (sWStamp_.size() < nS) { sWStamp_.assign(nS, 0); sRStamp_.assign(nS, 0); sDone_.assign(nS, 0); }
    stamp_++;
    if (parts != keyParts_) return -1;                 // (BEAGLE_ERROR_GENERAL) the engine sets the partition count first
    allowVirtual = allowVirtual && enabled_;

    // ---- closed list?  then the plan may be in the cache (planner.h)
    CacheEntry* fill = nullptr;
    if (cacheEnabled && count >= 16) {
        if (CacheEntry* hit = findCached(ops, count, tuple, parts, allowVirtual, chunkOps)) {
            stamp_++;
            replay(*hit, ops);
            cacheHits++;
            return 0;
        }
        bool closed = true, simple = true;
        for (int k = 0; k < count && closed; k++) {
            const int* op = ops + (size_t)k * tuple;
            const int part = tuple > 7 ? op[7] : 0;
            if (!compactTip[op[3]] && wStamp_[(size_t)op[3] * parts + part] != stamp_) closed = false;
            if (!compactTip[op[5]] && wStamp_[(size_t)op[5] * parts + part] != stamp_) closed = false;
            wStamp_[(size_t)op[0] * parts + part] = stamp_;
            if (op[1] != OP_NONE || op[0] == op[3] || op[0] == op[5] || (compactTip[op[0]] || leafPartials[op[0]]) ||
                (tuple > 7 && op[8] != OP_NONE)) simple = false;
        }
        stamp_++;                                      // the marks above must not look like producers to pass 1
        if (closed) {
            fill = &cache_[cacheNext_];
            cacheNext_ = (cacheNext_ + 1) % CACHE_WAYS;
            fill->valid = false; fill->cleanAtEpoch = -1; fill->tag = ++cacheTagNext_; fill->count = count; fill->tuple = tuple; fill->parts = parts; fill->chunkOps = chunkOps;
            fill->allowVirtual = allowVirtual; fill->stepLimit = allowVirtual ? stepLimit : 0; fill->tipEpoch = compactEpoch; fill->simple = simple;
            fill->ops.assign(ops, ops + (size_t)count * tuple);
        }
    }
    info_.assign(count, OpInfo());
    prod1_.assign(count, -1); prod2_.assign(count, -1);
    std::vector<char> consumed(count, 0);

    // ---- pass 1, list order: virtual definitions, producers, hold needs
    for (int k = 0; k < count; k++) {
        const int* op = ops + (size_t)k * tuple;
        if (k + 6 < count) {                               // (a definition is 1.5 KB: the destination's two lines that matter, a few operations ahead)
            const int* ahead = op + (size_t)6 * tuple;
            const char* a = reinterpret_cast<const char*>(&virt_[(size_t)ahead[0] * parts + (tuple > 7 ? ahead[7] : 0)]);
            __builtin_prefetch(a, 1); __builtin_prefetch(a + 64, 1);
        }
        OpInfo& o = info_[k];
        o.dest = op[0]; o.wS = op[1]; o.rS = op[2]; o.c1 = op[3]; o.m1 = op[4]; o.c2 = op[5]; o.m2 = op[6];
        o.part = tuple > 7 ? op[7] : 0;
        o.tip1 = compactTip[o.c1] != 0; o.tip2 = compactTip[o.c2] != 0;
        o.leaf1 = o.tip1 || leafPartials[o.c1] != 0; o.leaf2 = o.tip2 || leafPartials[o.c2] != 0;
        o.virtDest = false; o.emitted = false; o.need = 0; o.size = 1;
        const size_t kd = (size_t)o.dest * parts + o.part, kc1 = (size_t)o.c1 * parts + o.part, kc2 = (size_t)o.c2 * parts + o.part;
        if (!o.tip1 && wStamp_[kc1] == stamp_) { prod1_[k] = wOp_[kc1]; consumed[prod1_[k]] = 1; }
        if (!o.tip2 && wStamp_[kc2] == stamp_) { prod2_[k] = wOp_[kc2]; consumed[prod2_[k]] = 1; }
        if (o.wS != OP_NONE) sWStamp_[(size_t)o.wS * parts + o.part] = stamp_;

        const bool v1 = !o.leaf1 && virt_[kc1].on, v2 = !o.leaf2 && virt_[kc2].on;
        const int ownScale = o.wS != OP_NONE ? o.wS : o.rS;
        bool makeVirtual = false;
        if (allowVirtual && (o.leaf1 || v1) && (o.leaf2 || v2) && o.c1 != o.dest && o.c2 != o.dest) {
            VirtDef& ev = virt_[kd];
            // Steady state: the same op on the same buffers as when `dest` was last defined, its virtual children unchanged
            // (same definition version) and re-confirmed in this list exactly as they were fresh then -> the definition
            // stands; only its matrix snapshots are refreshed.  Anything else rebuilds it.
            auto childSame = [&](size_t c, bool tip, bool sigTip, int ver, bool fresh) {      // c: the child's key
                if (tip != sigTip) return false;
                if (tip) return true;
                const VirtDef& cv = virt_[c];
                return fresh && cv.stamp == stamp_ && cv.version == ver;
            };
            if (ev.on && ev.nSteps <= stepCap() && ev.sigC1 == o.c1 && ev.sigM1 == o.m1 && ev.sigC2 == o.c2 && ev.sigM2 == o.m2 && ev.sigScale == ownScale &&
                ev.sigMem1 == (o.leaf1 && !o.tip1) && ev.sigMem2 == (o.leaf2 && !o.tip2) &&
                childSame(kc1, o.leaf1, ev.sigTip1, ev.childVer1, ev.fresh1) && childSame(kc2, o.leaf2, ev.sigTip2, ev.childVer2, ev.fresh2)) {
                for (int st = 0; st < ev.nSteps; st++) {
                    out.snapPairs.push_back(ev.steps[st].originA); out.snapPairs.push_back(snapSlot((int)kd, st, 0));
                    out.snapPairs.push_back(ev.steps[st].originB); out.snapPairs.push_back(snapSlot((int)kd, st, 1));
                }
                ev.stamp = stamp_;
                makeVirtual = true;
            } else {
                VirtDef saved = ev;
                if (saved.on) clearVirtualKey((int)kd);
                makeVirtual = buildVirtual((int)kd, o.leaf1 ? o.c1 : (int)kc1, o.leaf1, o.leaf1 && !o.tip1, o.m1,
                                           o.leaf2 ? o.c2 : (int)kc2, o.leaf2, o.leaf2 && !o.tip2, o.m2, ownScale, out.snapPairs);
                if (!makeVirtual && saved.on) { virt_[kd] = saved; tagOf_[kd] = saved.cacheTag; tagEpoch_++; registerVirtual((int)kd); }
                if (makeVirtual) {
                    VirtDef& nv = virt_[kd];
                    nv.version = ++virtVersion_;
                    nv.sigC1 = o.c1; nv.sigM1 = o.m1; nv.sigC2 = o.c2; nv.sigM2 = o.m2; nv.sigScale = ownScale;
                    nv.sigTip1 = o.leaf1; nv.sigTip2 = o.leaf2; nv.sigMem1 = o.leaf1 && !o.tip1; nv.sigMem2 = o.leaf2 && !o.tip2;
                    nv.fresh1 = !o.leaf1 && virt_[kc1].stamp == stamp_; nv.fresh2 = !o.leaf2 && virt_[kc2].stamp == stamp_;
                    nv.childVer1 = o.leaf1 ? -1 : virt_[kc1].version; nv.childVer2 = o.leaf2 ? -1 : virt_[kc2].version;
                }
            }
        }
        if (!makeVirtual && virt_[kd].on) clearVirtualKey((int)kd);      // whatever it was, this op replaces it
        o.virtDest = makeVirtual;
        wStamp_[kd] = stamp_; wOp_[kd] = k;

        // hold slots the evaluation of this (real) node needs, and its size — children come earlier in the list
        if (!makeVirtual) {
            int need[2] = {0, 0}, size[2] = {0, 0}; bool eval[2] = {false, false}, real[2] = {false, false};
            for (int w = 0; w < 2; w++) {
                const bool tip = w ? o.tip2 : o.tip1;
                const int c = w ? o.c2 : o.c1, prod = w ? prod2_[k] : prod1_[k];
                if (tip) continue;
                if (prod >= 0 && !info_[prod].virtDest) { eval[w] = real[w] = true; need[w] = info_[prod].need; size[w] = info_[prod].size; }
                else if (virt_[key(c, o.part)].on) { eval[w] = true; need[w] = virtNeed(key(c, o.part)); }
            }
            o.size = 1 + size[0] + size[1];
            if (eval[0] && eval[1]) {
                auto cost = [&](int a, int b) { return real[a] ? std::max(need[a], need[b]) : std::max(need[a], 1 + need[b]); };
                o.need = std::min(cost(0, 1), cost(1, 0));
            } else o.need = eval[0] ? need[0] : eval[1] ? need[1] : 0;
            if (allSlots_ == 0u) o.need = 0;            // no hold slots: the first of two evaluated children goes through memory
        }
    }

    // ---- pass 2: walk order.  One slice per partition; inside it, every unconsumed real op is a root of the forest.
    // With chunkOps > 0 the forest is peeled in waves: a wave = the maximal subtrees of at most chunkOps micro-operations
    // among what is left, one slice each; what is above them reads their (stored) roots in a later wave.
    std::vector<int> partsSeen;
    {
        std::vector<char> seen(parts, 0);
        for (int k = 0; k < count; k++) if (!seen[info_[k].part]) { seen[info_[k].part] = 1; partsSeen.push_back(info_[k].part); }
    }
    lastWaves = 0;
    std::vector<int> weight, chunkRoots, stack;
    for (int part : partsSeen) {
        int wave = 0;
        for (;;) {
            bool chunked = false;
            const int chunk = (wave == 0 || chunkTopOps <= 0) ? chunkOps : std::min(chunkOps, chunkTopOps);
            if (chunk > 0) {
                // micro-operations below every op that is still to be emitted (children precede parents in the list)
                weight.assign(count, 0);
                long total = 0;
                for (int k = 0; k < count; k++) {
                    const OpInfo& o = info_[k];
                    if (o.part != part || o.virtDest || o.emitted) continue;
                    int w = 1;
                    for (int c = 0; c < 2; c++) {
                        const bool tip = c ? o.tip2 : o.tip1;
                        const int buf = c ? o.c2 : o.c1, prod = c ? prod2_[k] : prod1_[k];
                        if (tip) continue;
                        if (prod >= 0 && !info_[prod].virtDest) { if (!info_[prod].emitted) w += weight[prod]; }
                        else if (virt_[key(buf, o.part)].on) w += virt_[key(buf, o.part)].nSteps;
                    }
                    weight[k] = w;
                    if (!consumed[k]) total += w;
                }
                if (total > chunk + chunk / 2) {
                    chunkRoots.clear(); stack.clear();
                    for (int k = count - 1; k >= 0; k--) {
                        const OpInfo& o = info_[k];
                        if (o.part == part && !o.virtDest && !o.emitted && !consumed[k]) stack.push_back(k);
                    }
                    while (!stack.empty()) {
                        const int k = stack.back(); stack.pop_back();
                        if (weight[k] <= chunk) { chunkRoots.push_back(k); continue; }
                        bool descended = false;
                        for (int c = 1; c >= 0; c--) {
                            const int prod = c ? prod2_[k] : prod1_[k];
                            if (prod >= 0 && !info_[prod].virtDest && !info_[prod].emitted) { stack.push_back(prod); descended = true; }
                        }
                        if (!descended) chunkRoots.push_back(k);      // heavier than a chunk but nothing below is left to peel
                    }
                    // a DAG may reach the same op twice: emit once
                    std::sort(chunkRoots.begin(), chunkRoots.end());
                    chunkRoots.erase(std::unique(chunkRoots.begin(), chunkRoots.end()), chunkRoots.end());
                    // only worth a wave of its own when it really runs side by side
                    if (chunkRoots.size() >= 2) {
                        for (int k : chunkRoots) {
                            if (info_[k].emitted) continue;
                            PlanSeg seg; seg.progStart = (int)out.prog.size(); seg.partition = part; seg.wave = wave;
                            emitReal(k, allSlots_, out);
                            seg.progCount = (int)out.prog.size() - seg.progStart;
                            out.segs.push_back(seg);
                        }
                        chunked = true;
                        wave++;
                    }
                }
            }
            if (chunked) continue;
            PlanSeg seg; seg.progStart = (int)out.prog.size(); seg.partition = part; seg.wave = wave;
            for (int k = 0; k < count; k++) {
                OpInfo& o = info_[k];
                if (o.part != part || o.virtDest || o.emitted) continue;
                if (!consumed[k]) emitReal(k, allSlots_, out);
            }
            // virtual nodes of a rescaling evaluation still owe their scale factors if nothing above evaluated them
            for (int k = 0; k < count; k++) {
                const OpInfo& o = info_[k];
                if (o.part != part || !o.virtDest || o.wS == OP_NONE) continue;
                if (sDone_[(size_t)o.wS * parts + o.part] != stamp_) emitVirtual(key(o.dest, o.part), allSlots_, true, out);
            }
            seg.progCount = (int)out.prog.size() - seg.progStart;
            if (seg.progCount > 0) { out.segs.push_back(seg); wave++; }
            break;
        }
        lastWaves = std::max(lastWaves, wave);
    }
    // launches run wave by wave: order the slices that way (stable: partitions keep their order inside a wave)
    std::stable_sort(out.segs.begin(), out.segs.end(), [](const PlanSeg& a, const PlanSeg& b) { return a.wave < b.wave; });
    linkSlices(out);
    if (fill) {
        // (the entry TAKES the program — `out` is left with whatever the entry held — and `planned` names the entry's copy, as on
        // a hit; the definitions of unstored destinations are copied, steps in use only, over whatever the way held before)
        fill->plan.clear();
        fill->plan.prog.swap(out.prog); fill->plan.segs.swap(out.segs); fill->plan.deps.swap(out.deps);
        fill->plan.launchOrder.swap(out.launchOrder); fill->plan.snapPairs.swap(out.snapPairs); fill->plan.leaves = out.leaves;
        planned = &fill->plan;
        if ((int)fill->defs.size() < count) fill->defs.resize(count);
        fill->defOn.assign(count, 0);
        for (int k = 0; k < count; k++)
            if (info_[k].virtDest) { const int kk = key(info_[k].dest, info_[k].part); virt_[kk].cacheTag = fill->tag; tagOf_[kk] = fill->tag; tagEpoch_++; fill->defs[k] = virt_[kk]; fill->defOn[k] = 1; }
        plannedTag = fill->tag;
        fill->stored = lastStored; fill->memReads = lastMemReads; fill->holds = lastHolds; fill->waves = lastWaves;
        fill->valid = true;
    }
    return 0;
}

// Which slices read what other slices of the same plan store (planner.h PlanSeg): a slice reads a stored result only as a
// PK_MEM operand, and only of a slice of an EARLIER wave of the same partition (tests/native/plan_check.cpp checks both).
void WalkPlanner::linkSlices(Plan& out) {
    const size_t nKeys = (size_t)partialsCount_ * keyParts_;
    if (storedBy_.size() < nKeys) { storedBy_.assign(nKeys, -1); storedStamp_.assign(nKeys, 0); }
    linkStamp_++;
    const int n = (int)out.segs.size();
    for (int s = 0; s < n; s++) {
        const PlanSeg& sg = out.segs[s];
        for (int k = sg.progStart; k < sg.progStart + sg.progCount; k++) {
            const int st = out.prog[k].storeBuf;
            if (st >= 0) { const size_t kk = (size_t)st * keyParts_ + sg.partition; storedBy_[kk] = s; storedStamp_[kk] = linkStamp_; }
        }
    }
    out.deps.clear();
    std::vector<int> mark(n, -1);
    for (int s = 0; s < n; s++) {
        PlanSeg& sg = out.segs[s];
        sg.depStart = (int)out.deps.size();
        auto reads = [&](int kind, int buf) {
            if (kind != PK_MEM) return;
            const size_t kk = (size_t)buf * keyParts_ + sg.partition;
            if (kk >= nKeys || storedStamp_[kk] != linkStamp_) return;
            const int by = storedBy_[kk];
            if (by == s || mark[by] == s) return;
            mark[by] = s;
            out.deps.push_back(by);
        };
        for (int k = sg.progStart; k < sg.progStart + sg.progCount; k++) { reads(out.prog[k].k1, out.prog[k].a1); reads(out.prog[k].k2, out.prog[k].a2); }
        sg.depCount = (int)out.deps.size() - sg.depStart;
        sg.tail = 0;
    }
    // tickets (planner.h PlanSeg::next): does every slice have at most one dependant?
    {
        std::vector<int> dependants(n, 0);
        for (int s = 0; s < n; s++) out.segs[s].next = -1;
        for (int s = 0; s < n; s++)
            for (int d = out.segs[s].depStart; d < out.segs[s].depStart + out.segs[s].depCount; d++) { dependants[out.deps[d]]++; out.segs[out.deps[d]].next = s; }
        bool forest = n > 0;
        int leaves = 0;
        for (int s = 0; s < n; s++) {
            if (dependants[s] > 1 || out.segs[s].progCount <= 0) forest = false;
            if (out.segs[s].depCount == 0) leaves++;
        }
        out.leaves = forest ? leaves : 0;
    }
    // tail: own length + the longest tail among the slices that wait for this one.  Slices are sorted by wave and read only
    // earlier waves, so one backward sweep settles every slice before the ones it reads.
    for (int s = n - 1; s >= 0; s--) {
        PlanSeg& sg = out.segs[s];
        sg.tail += sg.progCount;
        for (int d = sg.depStart; d < sg.depStart + sg.depCount; d++) {
            PlanSeg& child = out.segs[out.deps[d]];
            child.tail = std::max(child.tail, sg.tail);
        }
    }
    out.launchOrder.resize(n);
    for (int s = 0; s < n; s++) out.launchOrder[s] = s;
    std::stable_sort(out.launchOrder.begin(), out.launchOrder.end(), [&](int a, int b) { return out.segs[a].tail > out.segs[b].tail; });
    if (launchMachines <= 0.0 || n < 3) return;
    // List scheduling on `machines` identical machines, a slice taking its length: whenever a machine is free, the ready slice
    // (all of its dependencies finished) with the longest tail starts; the order of the starts is the launch order.
    const int machines = std::max(1, (int)(launchMachines + 0.5));
    std::vector<int> waiting(n, 0);                       // unfinished dependencies
    std::vector<std::vector<int>> dependants(n);
    for (int s = 0; s < n; s++) {
        waiting[s] = out.segs[s].depCount;
        for (int d = out.segs[s].depStart; d < out.segs[s].depStart + out.segs[s].depCount; d++) dependants[out.deps[d]].push_back(s);
    }
    std::vector<long> freeAt(machines, 0);
    std::vector<std::pair<long, int>> running;            // (finish time, slice)
    std::vector<int> ready;
    for (int s : out.launchOrder) if (waiting[s] == 0) ready.push_back(s);      // (kept in descending-tail order)
    std::vector<int> order;
    order.reserve(n);
    auto retire = [&](long now) {
        for (size_t i = 0; i < running.size();) {
            if (running[i].first > now) { i++; continue; }
            for (int c : dependants[running[i].second])
                if (--waiting[c] == 0) {
                    const auto at = std::upper_bound(ready.begin(), ready.end(), c, [&](int a, int b) { return out.segs[a].tail > out.segs[b].tail; });
                    ready.insert(at, c);
                }
            running[i] = running.back(); running.pop_back();
        }
    };
    while ((int)order.size() < n) {
        const size_t m = std::min_element(freeAt.begin(), freeAt.end()) - freeAt.begin();
        long now = freeAt[m];
        retire(now);
        if (ready.empty()) {                              // nothing can start: the machine idles until the next slice finishes
            if (running.empty()) break;                   // (cannot happen: the dependencies are acyclic)
            now = std::min_element(running.begin(), running.end())->first;
            retire(now);
            for (long& f : freeAt) f = std::max(f, now);
            continue;
        }
        const int s = ready.front();
        ready.erase(ready.begin());
        order.push_back(s);
        freeAt[m] = now + out.segs[s].progCount;
        running.emplace_back(freeAt[m], s);
    }
    if ((int)order.size() == n) out.launchOrder = order;
}

WalkPlanner::CacheEntry* WalkPlanner::findCached(const int* ops, int count, int tuple, int parts, bool allowVirtual, int chunkOps) {
    for (CacheEntry& e : cache_)
        if (e.valid && e.count == count && e.tuple == tuple && e.parts == parts && e.chunkOps == chunkOps && e.allowVirtual == allowVirtual &&
            e.stepLimit == (allowVirtual ? stepLimit : 0) &&
            e.tipEpoch == compactEpoch && memcmp(e.ops.data(), ops, (size_t)count * tuple * sizeof(int)) == 0)
            return &e;
    return nullptr;
}

bool WalkPlanner::replayCached(const int* ops, int count, int tuple, int parts, bool allowVirtual, int chunkOps, bool* simple) {
    if (!cacheEnabled || count < 16 || parts != keyParts_) return false;
    allowVirtual = allowVirtual && enabled_;
    CacheEntry* e = findCached(ops, count, tuple, parts, allowVirtual, chunkOps);
    if (!e || (simple && !e->simple)) return false;    // (asked for a simple list only: nothing has been touched)
    parts_ = parts;
    stamp_ += 2;                                       // a list of its own, as in plan()
    replay(*e, ops);
    cacheHits++;
    if (simple) *simple = e->simple;
    return true;
}

// Put the planner into the state planning the (closed) list again would leave it in; the kept program is handed out
// through `planned`.  A definition still tagged with this entry is the one the entry wrote (every other writer of a
// definition resets the tag), so the steady state costs one comparison per operation.
void WalkPlanner::replay(const CacheEntry& e, const int* ops) {
    // (nobody has written a tag since this entry was last replayed: every key is as that replay left it)
    for (int k = 0; e.cleanAtEpoch != tagEpoch_ && k < e.count; k++) {
        const int part = e.tuple > 7 ? ops[(size_t)k * e.tuple + 7] : 0;
        const int dest = key(ops[(size_t)k * e.tuple], part);
        if (e.defOn[k] ? tagOf_[dest] == e.tag : tagOf_[dest] < 0) continue;        // already what the entry leaves behind
        if (k + 6 < e.count) {                                   // (a definition is 1.5 KB: both sides of the copy below miss the caches otherwise)
            const int aheadPart = e.tuple > 7 ? ops[(size_t)(k + 6) * e.tuple + 7] : 0;
            const char* a = reinterpret_cast<const char*>(&e.defs[(size_t)k + 6]);
            const char* b = reinterpret_cast<const char*>(&virt_[key(ops[(size_t)(k + 6) * e.tuple], aheadPart)]);
            __builtin_prefetch(a); __builtin_prefetch(a + 64); __builtin_prefetch(a + 128); __builtin_prefetch(a + 192);
            __builtin_prefetch(b, 1); __builtin_prefetch(b + 64, 1); __builtin_prefetch(b + 128, 1); __builtin_prefetch(b + 192, 1);
        }
        const VirtDef& want = e.defs[k];
        VirtDef& cur = virt_[dest];
        // The usual reason to be here: the same node over the same tips, defined by the entry of the OTHER scale-buffer set (the evaluations
        // right behind a rescaling one) — the definition's leaves stand in the tips' user lists already, only its scale buffers change.
        // Taking the key out of every list and putting it back cost more than planning the list from scratch (tests/native/plan_check.cpp
        // bench-replay: 303 us against 218 for a 1000-taxon list on the build container).
        bool sameLeaves = cur.on && e.defOn[k] && cur.nSteps == want.nSteps;
        for (int st = 0; sameLeaves && st < want.nSteps; st++)
            sameLeaves = cur.steps[st].tipA == want.steps[st].tipA && cur.steps[st].tipB == want.steps[st].tipB;
        if (sameLeaves) {
            replayInPlace++;
            for (int st = 0; st < want.nSteps; st++) {
                const int was = cur.steps[st].scaleIdx;
                if (was < 0 || was == want.steps[st].scaleIdx) continue;
                bool still = false;
                for (int q = 0; q < want.nSteps && !still; q++) still = want.steps[q].scaleIdx == was;
                if (!still) { std::vector<int>& u = scaleUsers_[was]; u.erase(std::remove(u.begin(), u.end(), dest), u.end()); }
            }
        } else {
            if (cur.on) clearVirtualKey(dest);
            if (!e.defOn[k]) continue;                            // (defs[k] may be a leftover of an earlier list in this way)
        }
        cur = want;                                               // (tagged with e.tag when the entry was filled)
        tagOf_[dest] = e.tag; tagEpoch_++;
        cur.stamp = stamp_;
        cur.version = ++virtVersion_;
        cur.childVer1 = cur.sigTip1 ? -1 : virt_[key(cur.sigC1, part)].version;
        cur.childVer2 = cur.sigTip2 ? -1 : virt_[key(cur.sigC2, part)].version;
        if (sameLeaves) {
            for (int st = 0; st < cur.nSteps; st++) {
                const int sc = cur.steps[st].scaleIdx;
                if (sc < 0) continue;
                std::vector<int>& u = scaleUsers_[sc];
                if (std::find(u.begin(), u.end(), dest) == u.end()) u.push_back(dest);
            }
        } else registerVirtual(dest);
    }
    e.cleanAtEpoch = tagEpoch_;
    planned = &e.plan; plannedTag = e.tag;
    lastStored = e.stored; lastMemReads = e.memReads; lastHolds = e.holds; lastWaves = e.waves;
}

void WalkPlanner::planMaterialize(const std::vector<int>& keys, Plan& out) {
    out.clear();
    parts_ = keyParts_;
    for (int part = 0; part < keyParts_; part++) {      // one slice per partition that has something to materialise
        PlanSeg seg; seg.progStart = (int)out.prog.size(); seg.partition = part; seg.wave = 0;
        for (int X : keys) {
            if (partitionOf(X) != part || !virt_[X].on) continue;
            emitVirtual(X, allSlots_, false, out);
            out.prog.back().storeBuf = bufferOf(X);
            clearVirtualKey(X);
        }
        seg.progCount = (int)out.prog.size() - seg.progStart;
        if (seg.progCount > 0) out.segs.push_back(seg);
    }
    linkSlices(out);
}

bool foldScaleFactors(const Plan& plan, int maxMembers, FoldMap& out) {
    const size_t n = plan.prog.size();
    out.payStart.assign(n + 1, 0);
    out.members.clear();
    for (const MicroOp& m : plan.prog) if (m.smode == PS_WRITE) return false;
    // first the count of members every micro-operation pays for (slices tile plan.prog in any order), then the lists themselves
    std::vector<std::vector<int>> pays(n);
    std::vector<int> acc, hold[3], cur;
    for (const PlanSeg& sg : plan.segs) {
        acc.clear();
        for (std::vector<int>& h : hold) h.clear();          // (no hold slot is in use at the start of a slice)
        for (int i = sg.progStart; i < sg.progStart + sg.progCount; i++) {
            const MicroOp& m = plan.prog[(size_t)i];
            cur.clear();
            if (m.k1 >= PK_H0) { const std::vector<int>& h = hold[m.k1 - PK_H0]; cur.insert(cur.end(), h.begin(), h.end()); }
            if (m.k2 == PK_ACC) cur.insert(cur.end(), acc.begin(), acc.end());
            if (m.smode == PS_READ) cur.push_back(m.scaleIdx);
            if (m.storeBuf >= 0 || i == sg.progStart + sg.progCount - 1 || (int)cur.size() >= maxMembers) { pays[(size_t)i] = cur; cur.clear(); }
            if (m.hold) hold[m.hold - 1] = cur;
            acc.swap(cur);
        }
    }
    for (size_t i = 0; i < n; i++) {
        out.payStart[i] = (int)out.members.size();
        out.members.insert(out.members.end(), pays[i].begin(), pays[i].end());
    }
    out.payStart[n] = (int)out.members.size();
    return true;
}

}  // namespace mi355
