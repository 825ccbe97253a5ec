// plan_check.cpp — CPU check of the walk planner (beast-mcmc_amd/csrc/planner.cpp).  TEST INFRASTRUCTURE, not product.
//
// Two worlds are driven by the same sequence of BEAGLE-style calls:
//   truth   every buffer index is real storage; an operation list is evaluated in list order, one op after the other
//           (the semantics of lib/beagle.jar!beagle/GeneralBeagleImpl#updatePartials)
//   planned the planner's micro-operation programs are executed by an index-level interpreter of the walk kernel's
//           register model (ACC + two hold slots per (pattern, category)); virtual buffers hold no data
// After every list all real buffers and scale buffers must agree BITWISE; virtual buffers are compared when they are
// materialised (at random, and all of them at the end).  Scenarios: full evaluations in both traversal orders, rescaling
// write / read mode, MCMC-style partial updates with BufferIndexHelper flips and rejections
// (src/dr/evomodel/treedatalikelihood/BufferIndexHelper.java:71-106), tip-state changes, partitioned 9-int lists,
// lists with hazards, a 5000-tip caterpillar (4999 dependency levels: the emission must not recurse on the native stack).
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "../../beast-mcmc_amd/csrc/planner.h"

using namespace mi355;

static int P = 7, C = 2;
static long foldedPlans = 0, foldedPays = 0, foldedMembers = 0, unfoldedReads = 0;
static const char* g_where = "";
static long g_list = 0;
#define CHECK(cond, m, k) do { if (!(cond)) { fprintf(stderr, "FAILED %s [%s, list %ld] micro-op %d: store %d k1 %d a1 %d k2 %d a2 %d mat %d %d scale %d mode %d hold %d\n", #cond, g_where, g_list, k, (m).storeBuf, (m).k1, (m).a1, (m).k2, (m).a2, (m).mat1, (m).mat2, (m).scaleIdx, (m).smode, (m).hold); exit(1); } } while (0)
struct V4 { double v[4]; };
static inline V4 matvec(const double* m, const V4& x) {
    V4 y;
    for (int i = 0; i < 4; i++) y.v[i] = std::fma(m[4 * i + 3], x.v[3], std::fma(m[4 * i + 2], x.v[2], std::fma(m[4 * i + 1], x.v[1], m[4 * i] * x.v[0])));
    return y;
}
static inline V4 column(const double* m, int s) { V4 y; for (int i = 0; i < 4; i++) y.v[i] = s < 4 ? m[4 * i + s] : 1.0; return y; }

struct World {
    std::vector<std::vector<double>> partials;     // [buf] -> C*P*4 or empty
    std::vector<std::vector<uint8_t>> tips;        // [buf] -> P or empty
    std::vector<std::vector<double>> mats;         // [slot] -> C*16
    std::vector<std::vector<double>> scale;        // [idx] -> P raw factors (empty: never written)
};

static V4 childFactor(const World& w, int buf, bool tip, int mat, int c, int p) {
    const double* m = &w.mats[mat][(size_t)c * 16];
    if (tip) return column(m, w.tips[buf][p]);
    assert(!w.partials[buf].empty());
    V4 x; memcpy(x.v, &w.partials[buf][((size_t)c * P + p) * 4], 32);
    return matvec(m, x);
}

// list-order evaluation of one op (all patterns of [p0, p1))
static void truthOp(World& w, const std::vector<char>& compact, const int* op, int p0, int p1) {
    const int dest = op[0], wS = op[1], rS = op[2], c1 = op[3], m1 = op[4], c2 = op[5], m2 = op[6];
    std::vector<double> out = w.partials[dest];
    if (out.empty()) out.assign((size_t)C * P * 4, 0.0);
    for (int p = p0; p < p1; p++) {
        V4 r[8];
        for (int c = 0; c < C; c++) {
            const V4 a = childFactor(w, c1, compact[c1], m1, c, p), b = childFactor(w, c2, compact[c2], m2, c, p);
            for (int i = 0; i < 4; i++) r[c].v[i] = a.v[i] * b.v[i];
        }
        if (wS >= 0) {
            double m = 0.0;
            for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) m = std::fmax(m, r[c].v[i]);
            if (!(m > 0.0)) m = 1.0;
            if (w.scale[wS].empty()) w.scale[wS].assign(P, 0.0);
            w.scale[wS][p] = m;
            const double im = 1.0 / m;
            for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) r[c].v[i] *= im;
        } else if (rS >= 0) {
            const double im = 1.0 / w.scale[rS][p];
            for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) r[c].v[i] *= im;
        }
        for (int c = 0; c < C; c++) memcpy(&out[((size_t)c * P + p) * 4], r[c].v, 32);
    }
    w.partials[dest] = out;
}

static long g_ticketRuns = 0;          // programs executed on tickets (runPlan below)
// the walk kernel's register model, index level
// fold != nullptr: the engine's read-mode folding (planner.h FoldMap) — a micro-operation multiplies by the product of the reciprocals
// of the scale buffers it pays for (nothing where it pays for none) instead of by its own node's
static void runPlan(World& w, const Plan& plan, const std::vector<int>& partStart, const std::vector<int>& partEnd, const FoldMap* fold = nullptr) {
    // snapshot copies: one parallel launch (all sources are read before any destination is written)
    std::vector<std::vector<double>> src;
    for (size_t i = 0; i + 1 < plan.snapPairs.size(); i += 2) src.push_back(w.mats[plan.snapPairs[i]]);
    for (size_t i = 0; i + 1 < plan.snapPairs.size(); i += 2) w.mats[plan.snapPairs[i + 1]] = src[i / 2];
    // The kernels are software-pipelined two micro-operations deep: the partials a micro-operation reads as its FIRST child
    // (PK_MEM in k1) are requested at the start of the PREVIOUS micro-operation's stage, before that one stores its result
    // (kernels_walk4.hip WALK_STAGE, tools/gen_walk4_fast.py stage()).  A program must therefore never read in k1 what the
    // micro-operation right before it stores; and slices of one wave run side by side, so none may read what another stores.
    for (size_t a = 0; a < plan.segs.size(); a++) {
        const PlanSeg& sg = plan.segs[a];
        for (int k = sg.progStart + 1; k < sg.progStart + sg.progCount; k++) {
            const MicroOp& m = plan.prog[k];
            CHECK(!(m.k1 == PK_MEM && plan.prog[k - 1].storeBuf == m.a1), m, k);
        }
        for (size_t b = 0; b < plan.segs.size(); b++) {
            const PlanSeg& ob = plan.segs[b];
            if (a == b || ob.wave != sg.wave || ob.partition != sg.partition) continue;
            for (int k = sg.progStart; k < sg.progStart + sg.progCount; k++) {
                const MicroOp& m = plan.prog[k];
                for (int q = ob.progStart; q < ob.progStart + ob.progCount; q++) {
                    const int st = plan.prog[q].storeBuf;
                    if (st < 0) continue;
                    CHECK(!(m.k1 == PK_MEM && m.a1 == st), m, k);
                    CHECK(!(m.k2 == PK_MEM && m.a2 == st), m, k);
                }
            }
        }
    }
    // One launch may run every slice of the plan side by side, each workgroup first waiting for the slices in its dependency
    // list (planner.h PlanSeg): every stored result a slice reads must come from a slice it waits for, of an earlier wave, and
    // the launch order must put every slice behind the ones it waits for.
    {
        CHECK(plan.launchOrder.size() == plan.segs.size(), plan.prog[0], 0);
        std::vector<int> pos(plan.segs.size(), -1);
        for (size_t i = 0; i < plan.launchOrder.size(); i++) { CHECK(pos[plan.launchOrder[i]] < 0, plan.prog[0], (int)i); pos[plan.launchOrder[i]] = (int)i; }
        for (size_t a = 0; a < plan.segs.size(); a++) {
            const PlanSeg& sg = plan.segs[a];
            for (int d = sg.depStart; d < sg.depStart + sg.depCount; d++) {
                const int b = plan.deps[d];
                CHECK(plan.segs[b].wave < sg.wave && plan.segs[b].partition == sg.partition, plan.prog[sg.progStart], (int)a);
                CHECK(pos[b] < pos[a], plan.prog[sg.progStart], (int)a);
                CHECK(plan.segs[b].tail >= plan.segs[b].progCount + sg.tail, plan.prog[sg.progStart], (int)a);
            }
            for (size_t b = 0; b < plan.segs.size(); b++) {
                const PlanSeg& ob = plan.segs[b];
                if (a == b || ob.partition != sg.partition) continue;
                bool waits = false;
                for (int d = sg.depStart; d < sg.depStart + sg.depCount; d++) waits = waits || plan.deps[d] == (int)b;
                if (waits) continue;
                for (int k = sg.progStart; k < sg.progStart + sg.progCount; k++) {
                    const MicroOp& m = plan.prog[k];
                    for (int q = ob.progStart; q < ob.progStart + ob.progCount; q++) {
                        const int st = plan.prog[q].storeBuf;
                        if (st < 0) continue;
                        CHECK(!(m.k1 == PK_MEM && m.a1 == st), m, k);
                        CHECK(!(m.k2 == PK_MEM && m.a2 == st), m, k);
                    }
                }
            }
        }
    }
    auto runSlice = [&](int si) {
        const PlanSeg& sg = plan.segs[si];
        for (int p = partStart[sg.partition]; p < partEnd[sg.partition]; p++) {
            V4 ACC[8], H[3][8];
            for (int k = sg.progStart; k < sg.progStart + sg.progCount; k++) {
                const MicroOp& m = plan.prog[k];
                V4 r[8];
                for (int c = 0; c < C; c++) {
                    V4 f1, f2;
                    const double* M1 = &w.mats[m.mat1][(size_t)c * 16];
                    const double* M2 = &w.mats[m.mat2][(size_t)c * 16];
                    if (m.k1 == PK_TIPS) f1 = column(M1, w.tips[m.a1][p]);
                    else if (m.k1 == PK_MEM) { CHECK(!w.partials[m.a1].empty(), m, k); V4 x; memcpy(x.v, &w.partials[m.a1][((size_t)c * P + p) * 4], 32); f1 = matvec(M1, x); }
                    else { assert(m.k1 >= PK_H0 && m.k1 <= PK_H2); f1 = matvec(M1, H[m.k1 - PK_H0][c]); }
                    if (m.k2 == PK_TIPS) f2 = column(M2, w.tips[m.a2][p]);
                    else if (m.k2 == PK_MEM) { CHECK(!w.partials[m.a2].empty(), m, k); V4 x; memcpy(x.v, &w.partials[m.a2][((size_t)c * P + p) * 4], 32); f2 = matvec(M2, x); }
                    else { assert(m.k2 == PK_ACC); f2 = matvec(M2, ACC[c]); }
                    for (int i = 0; i < 4; i++) r[c].v[i] = f1.v[i] * f2.v[i];
                }
                if (m.smode == PS_WRITE) {
                    double mx = 0.0;
                    for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) mx = std::fmax(mx, r[c].v[i]);
                    if (!(mx > 0.0)) mx = 1.0;
                    if (w.scale[m.scaleIdx].empty()) w.scale[m.scaleIdx].assign(P, 0.0);
                    w.scale[m.scaleIdx][p] = mx;
                    const double im = 1.0 / mx;
                    for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) r[c].v[i] *= im;
                } else if (fold) {
                    double im = 1.0;
                    for (int q = fold->payStart[k]; q < fold->payStart[k + 1]; q++) { CHECK(!w.scale[fold->members[q]].empty(), m, k); im *= 1.0 / w.scale[fold->members[q]][p]; }
                    if (fold->payStart[k + 1] > fold->payStart[k]) for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) r[c].v[i] *= im;
                } else if (m.smode == PS_READ) {
                    CHECK(!w.scale[m.scaleIdx].empty(), m, k);
                    const double im = 1.0 / w.scale[m.scaleIdx][p];
                    for (int c = 0; c < C; c++) for (int i = 0; i < 4; i++) r[c].v[i] *= im;
                }
                if (m.storeBuf >= 0) {
                    if (w.partials[m.storeBuf].empty()) w.partials[m.storeBuf].assign((size_t)C * P * 4, 0.0);
                    for (int c = 0; c < C; c++) memcpy(&w.partials[m.storeBuf][((size_t)c * P + p) * 4], r[c].v, 32);
                }
                for (int c = 0; c < C; c++) { ACC[c] = r[c]; if (m.hold) H[m.hold - 1][c] = r[c]; }
            }
        }
    };
    // Two ways to run a one-launch program (kernels_walk4.hip).  FLAGS: every slice has its own workgroups, which wait for the slices
    // they read from — any order that puts a slice behind its dependencies, here the launch order.  TICKETS (plan.leaves > 0: the
    // slices form a forest): only the slices without dependencies start; whoever finishes slice s counts itself in at s.next and the
    // LAST to arrive there carries on with that slice.  Every other program with a forest is run that way here, the leaves in the
    // REVERSE of the launch order (the order must not matter), and every slice must have run exactly once at the end.
    static unsigned long ticketToggle = 0;
    if (plan.leaves > 0) {
        const int n = (int)plan.segs.size();
        int leaves = 0;
        for (int a = 0; a < n; a++) {
            const PlanSeg& sg = plan.segs[a];
            CHECK(sg.progCount > 0, plan.prog[0], a);
            if (sg.depCount == 0) leaves++;
            for (int d = sg.depStart; d < sg.depStart + sg.depCount; d++) CHECK(plan.segs[plan.deps[d]].next == a, plan.prog[sg.progStart], a);
            if (sg.next >= 0) {
                bool listed = false;
                const PlanSeg& nx = plan.segs[sg.next];
                for (int d = nx.depStart; d < nx.depStart + nx.depCount; d++) listed = listed || plan.deps[d] == a;
                CHECK(listed, plan.prog[sg.progStart], a);
            }
        }
        CHECK(leaves == plan.leaves, plan.prog[0], leaves);
    }
    if (plan.leaves > 0 && (ticketToggle++ & 1)) {
        const int n = (int)plan.segs.size();
        std::vector<int> arrived(n, 0), ran(n, 0);
        for (int i = n - 1; i >= 0; i--) {
            int s = plan.launchOrder[i];
            if (plan.segs[s].depCount != 0) continue;
            for (;;) {
                runSlice(s); ran[s]++;
                const int nxt = plan.segs[s].next;
                if (nxt < 0 || ++arrived[nxt] != plan.segs[nxt].depCount) break;
                s = nxt;
            }
        }
        for (int a = 0; a < n; a++) CHECK(ran[a] == 1, plan.prog[plan.segs[a].progStart], a);
        g_ticketRuns++;
    } else
        for (int si : plan.launchOrder) runSlice(si);
}

// The planner's user lists against its definitions: a definition's key stands in the list of every leaf and every scale buffer its steps
// name, and in no other (replay() edits the lists in place when a definition keeps its leaves: planner.cpp).
static void checkUserLists(const WalkPlanner& pl, int nBuf, int nScale, const char* where) {
    const int parts = pl.partitionCount();
    std::vector<std::vector<int>> wantTip((size_t)nBuf), wantScale((size_t)nScale);
    for (int k = 0; k < nBuf * parts; k++) {
        if (!pl.isVirtualKey(k)) continue;
        const VirtDef& v = pl.definition(k);
        for (int s = 0; s < v.nSteps; s++) {
            if (v.steps[s].tipA >= 0) wantTip[(size_t)v.steps[s].tipA].push_back(k);
            if (v.steps[s].tipB >= 0) wantTip[(size_t)v.steps[s].tipB].push_back(k);
            if (v.steps[s].scaleIdx >= 0) wantScale[(size_t)v.steps[s].scaleIdx].push_back(k);
        }
    }
    auto same = [](std::vector<int> a, std::vector<int> b) {
        std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end());
        std::sort(b.begin(), b.end());
        return a == b;                               // (b unsorted-unique on purpose: a key listed twice is an error too)
    };
    for (int t = 0; t < nBuf; t++) if (!same(wantTip[(size_t)t], pl.tipUsers(t))) { fprintf(stderr, "USER LISTS: leaf %d [%s]\n", t, where); exit(1); }
    for (int x = 0; x < nScale; x++) if (!same(wantScale[(size_t)x], pl.scaleUsers(x))) { fprintf(stderr, "USER LISTS: scale buffer %d [%s]\n", x, where); exit(1); }
}

static long g_replayInPlace = 0;
struct Harness {
    ~Harness() { g_replayInPlace += pl.replayInPlace; }
    int T, nBuf, nMat, nScale;
    World truth, plan;
    WalkPlanner pl;
    std::vector<char> compact;
    std::vector<int> partStart{0}, partEnd;
    int parts = 1;
    std::mt19937 rng;
    long lists = 0, micro = 0, holds = 0, memReads = 0, stored = 0, materialised = 0, waves = 0, segsTotal = 0;
    long fastReplays = 0;
    int fixedChunk = -1;          // >= 0: every list is planned with this chunk size (the engine's choice depends on the instance only)
    bool mixStepLimit = false;    // some lists are planned under a step limit (scenarioMcmc)

    void init(int tips, int nBuffers, int nMatrices, int nScales, bool virt, unsigned seed) {
        T = tips; nBuf = nBuffers; nMat = nMatrices; nScale = nScales; rng.seed(seed);
        { static const int caps[6] = {6, 8, 12, 16, 24, 32}; static const int holdChoice[3] = {2, 3, 0}; pl.init(nBuf, T, nMat, nScale, caps[seed % 6], virt, holdChoice[(seed / 6) % 3]); }     // (0 hold slots: the 21..64-state walk, kernels_mfma.hip k_walkT64)
        // one-launch programs: the simulated schedule's machine count and the smaller slices above the first wave, varied
        { static const double mach[4] = {0.0, 1.0, 3.0, 10.4}; pl.launchMachines = mach[(seed / 2) % 4]; }
        { static const int top[3] = {0, 8, 16}; pl.chunkTopOps = top[(seed / 3) % 3]; }
        const int slots = pl.matrixSlots();
        for (World* w : {&truth, &plan}) {
            w->partials.assign(nBuf, {}); w->tips.assign(nBuf, {}); w->mats.assign(slots, std::vector<double>((size_t)C * 16, 0.0));
            w->scale.assign(nScale, {});
        }
        compact.assign(nBuf, 0);
        partEnd.assign(1, P);
    }
    void setTipStates(int tip) {
        materialiseKeys(pl.tipUsers(tip));
        std::vector<uint8_t> s(P);
        for (int p = 0; p < P; p++) s[p] = (uint8_t)(rng() % 5);
        truth.tips[tip] = s; plan.tips[tip] = s;
        compact[tip] = 1; pl.setCompactTip(tip, true); pl.setLeafPartials(tip, false);
    }
    void setTipPartials(int tip) {
        materialiseKeys(pl.tipUsers(tip));
        pl.clearVirtual(tip);
        std::vector<double> x((size_t)C * P * 4);
        std::uniform_real_distribution<double> u(0.05, 1.0);
        for (double& v : x) v = u(rng);
        truth.partials[tip] = x; plan.partials[tip] = x;
        compact[tip] = 0; pl.setCompactTip(tip, false); pl.setLeafPartials(tip, true);      // uploaded partials: a leaf definitions may read
    }
    void setMatrix(int slot) {
        std::uniform_real_distribution<double> u(0.01, 1.0);
        std::vector<double> m((size_t)C * 16);
        for (double& v : m) v = u(rng) * 1e-3;     // small entries: products shrink, rescaling matters
        truth.mats[slot] = m; plan.mats[slot] = m;
    }
    void materialise(const std::vector<int>& bufs) {       // buffers -> the keys of their (virtual) partitions
        std::vector<int> keys;
        for (int b : bufs) pl.keysOf(b, keys);
        materialiseKeys(keys);
    }
    void materialiseKeys(std::vector<int> keys) {
        if (keys.empty()) return;
        Plan mp;
        std::vector<int> which;
        for (int x : keys) if (pl.isVirtualKey(x)) which.push_back(x);
        pl.planMaterialize(keys, mp);
        runPlan(plan, mp, partStart, partEnd);
        for (int x : which) { compareRange(pl.bufferOf(x), partStart[pl.partitionOf(x)], partEnd[pl.partitionOf(x)], "materialised"); materialised++; }
    }
    void compareRange(int b, int p0, int p1, const char* what) {
        if (truth.partials[b].empty()) return;
        bool ok = plan.partials[b].size() == truth.partials[b].size();
        for (int c = 0; ok && c < C; c++)
            ok = memcmp(&plan.partials[b][((size_t)c * P + p0) * 4], &truth.partials[b][((size_t)c * P + p0) * 4], (size_t)(p1 - p0) * 32) == 0;
        if (!ok) { fprintf(stderr, "MISMATCH (%s) buffer %d patterns [%d, %d) after %ld lists [%s]\n", what, b, p0, p1, lists, g_where); exit(1); }
    }
    void compareAll() {
        for (int b = 0; b < nBuf; b++) {
            if (compact[b]) continue;
            for (int k = 0; k < parts; k++) if (!pl.isVirtualKey(pl.key(b, k))) compareRange(b, partStart[k], partEnd[k], "real");
        }
        for (int s = 0; s < nScale; s++) {
            if (truth.scale[s].empty()) continue;
            if (plan.scale[s].size() != truth.scale[s].size() || memcmp(plan.scale[s].data(), truth.scale[s].data(), (size_t)P * 8) != 0) {
                fprintf(stderr, "MISMATCH scale %d after %ld lists\n", s, lists); exit(1);
            }
        }
    }
    // one updatePartials call, the way the engine drives the planner
    void update(const std::vector<int>& ops, int tuple) {
        const int count = (int)(ops.size() / tuple);
        for (int k = 0; k < count; k++) {
            const int* op = &ops[(size_t)k * tuple];
            const int part = tuple > 7 ? op[7] : 0;
            truthOp(truth, compact, op, partStart[part], partEnd[part]);
        }
        // a gradient chain's post-order lists are planned with short definitions only (planner.h stepLimit): mixed in at random
        pl.stepLimit = mixStepLimit && (lists % 5 == 3 || lists % 7 == 5) ? 1 + (int)(lists % 2) : 0;
        int begin = 0;
        {   // the engine's fast path for a repeated closed list (engine_walk.cpp runOperationsWalk): no checks, no planning
            bool simple = false;
            if (fixedChunk >= 0 && pl.replayCached(ops.data(), count, tuple, parts, true, fixedChunk, &simple)) {
                assert(simple);
                checkFolded(*pl.planned);
                runPlan(plan, *pl.planned, partStart, partEnd);
                fastReplays++;
                begin = count;
            }
        }
        while (begin < count) {
            const int n = pl.hazardFreePrefix(ops.data(), begin, count, tuple, parts);
            assert(n >= 1);
            const int* sub = ops.data() + (size_t)begin * tuple;
            std::vector<int> need;                 // (keys)
            pl.mustMaterializeBefore(sub, n, tuple, need);
            materialiseNoCompare(need);
            Plan scratch;
            static const int chunkChoices[6] = {0, 0, 3, 8, 20, 64};
            const int rc = pl.plan(sub, n, tuple, parts, true, scratch, fixedChunk >= 0 ? fixedChunk : chunkChoices[rng() % 6]);
            const Plan& p = *pl.planned;           // `scratch`, or the planner's cached copy
            for (size_t q = 1; q < p.segs.size(); q++) assert(p.segs[q].wave >= p.segs[q - 1].wave);
            waves += pl.lastWaves; segsTotal += (long)p.segs.size();
            assert(rc == 0);
            // every MEM child must be real data, every destination must end up real or virtual
            if (begin == 0 && n == count) checkFolded(p);         // (the truth world holds the values after the WHOLE list)
            runPlan(plan, p, partStart, partEnd);
            micro += (long)p.prog.size(); holds += pl.lastHolds; memReads += pl.lastMemReads; stored += pl.lastStored;
            begin += n;
        }
        if (pl.stepLimit > 0)
            for (int k = 0; k < count; k++) {
                const int* op = &ops[(size_t)k * tuple];
                const int kk = pl.key(op[0], tuple > 7 ? op[7] : 0);
                if (pl.isVirtualKey(kk) && pl.definition(kk).nSteps > pl.stepLimit) { fprintf(stderr, "a definition of %d steps under a limit of %d [%s, list %ld]\n", pl.definition(kk).nSteps, pl.stepLimit, g_where, lists); exit(1); }
            }
        pl.stepLimit = 0;
        lists++; g_list = lists;
        compareAll();
        if (nBuf <= 1200) checkUserLists(pl, nBuf, nScale, g_where);
    }
    // The engine's read-mode folding of reciprocal scale factors (planner.h foldScaleFactors): the same program with the factors of
    // unstored results applied where the planner says must give every STORED result the value list-order evaluation gives it — to
    // rounding (the factors are the same numbers multiplied in another order); a factor applied twice or not at all would be off
    // by orders of magnitude.  Run on a copy of the planned world as it is before the program.
    void checkFolded(const Plan& p) {
        if (nBuf > 1200) return;                      // (the copy: not for the 5000-tip ladder)
        FoldMap fm;
        static const int caps[3] = {32, 4, 2};
        if (!foldScaleFactors(p, caps[lists % 3], fm)) return;
        bool anyRead = false;
        for (const MicroOp& m : p.prog) anyRead = anyRead || m.smode == PS_READ;
        if (!anyRead) return;
        // every factor is paid exactly once: the members over the whole program are the program's read-mode scale indices
        {
            std::vector<int> want, got(fm.members);
            for (const MicroOp& m : p.prog) if (m.smode == PS_READ) want.push_back(m.scaleIdx);
            std::sort(want.begin(), want.end()); std::sort(got.begin(), got.end());
            if (want != got) { fprintf(stderr, "FOLD: members differ from the program's read-mode factors [%s, list %ld]\n", g_where, lists); exit(1); }
        }
        World shadow = plan;
        runPlan(shadow, p, partStart, partEnd, &fm);
        for (const PlanSeg& sg : p.segs)
            for (int k = sg.progStart; k < sg.progStart + sg.progCount; k++) {
                const int b = p.prog[k].storeBuf;
                if (b < 0) continue;
                for (int c = 0; c < C; c++)
                    for (int q = partStart[sg.partition]; q < partEnd[sg.partition]; q++)
                        for (int i = 0; i < 4; i++) {
                            const double x = shadow.partials[b][((size_t)c * P + q) * 4 + i], t = truth.partials[b][((size_t)c * P + q) * 4 + i];
                            if (!(std::fabs(x - t) <= 1e-12 * std::fabs(t))) {
                                fprintf(stderr, "FOLD MISMATCH buffer %d pattern %d: %.17g against %.17g [%s, list %ld]\n", b, q, x, t, g_where, lists); exit(1);
                            }
                        }
            }
        foldedPlans++;
        for (size_t k = 0; k < p.prog.size(); k++) { if (fm.payStart[k + 1] > fm.payStart[k]) foldedPays++; if (p.prog[k].smode == PS_READ) unfoldedReads++; }
        foldedMembers += (long)fm.members.size();
    }
    // materialise without comparing: the truth world already holds the values AFTER the list (used for buffers whose
    // old definition is about to be invalidated by a scale rewrite — their truth value is the OLD one only if the list
    // does not rewrite them; they are checked by compareAll afterwards)
    void materialiseNoCompare(const std::vector<int>& xs) {
        if (xs.empty()) return;
        Plan mp;
        pl.planMaterialize(xs, mp);
        runPlan(plan, mp, partStart, partEnd);
    }
};

// ---- a tree and BEAST's buffer-index protocol -------------------------------------------------------------------
struct Tree {
    int T; std::vector<int> left, right, parent;   // internal nodes T..2T-2; root = 2T-2
    void random(int tips, std::mt19937& rng, bool caterpillar) {
        T = tips; const int N = 2 * T - 1;
        left.assign(N, -1); right.assign(N, -1); parent.assign(N, -1);
        std::vector<int> roots;
        for (int i = 0; i < T; i++) roots.push_back(i);
        for (int n = T; n < N; n++) {
            int a, b;
            if (caterpillar) { a = roots.size() - 1; b = 0; if (a == b) b = 1; }
            else { a = rng() % roots.size(); do { b = rng() % roots.size(); } while (b == a); }
            left[n] = roots[a]; right[n] = roots[b]; parent[roots[a]] = n; parent[roots[b]] = n;
            if (a < b) std::swap(a, b);
            roots.erase(roots.begin() + a); roots.erase(roots.begin() + b);
            roots.push_back(n);
        }
    }
    void postOrder(int root, std::vector<int>& out) const {      // iterative: the ladder tree is 4999 levels deep
        std::vector<std::pair<int, int>> st{{root, 0}};
        while (!st.empty()) {
            const int n = st.back().first, phase = st.back().second++;
            if (n < T) { st.pop_back(); continue; }
            if (phase == 0) st.push_back({left[n], 0});
            else if (phase == 1) st.push_back({right[n], 0});
            else { out.push_back(n); st.pop_back(); }
        }
    }
    std::vector<int> levelOrder() const {   // reverse level order: deepest internal nodes first
        const int N = 2 * T - 1;
        std::vector<int> depth(N, 0), order;
        for (int n = N - 2; n >= 0; n--) {}
        std::vector<int> stack{N - 1};
        std::vector<std::pair<int, int>> byDepth;
        while (!stack.empty()) { int n = stack.back(); stack.pop_back(); if (n < T) continue; byDepth.push_back({depth[n], n});
            depth[left[n]] = depth[right[n]] = depth[n] + 1; stack.push_back(left[n]); stack.push_back(right[n]); }
        std::stable_sort(byDepth.begin(), byDepth.end(), [](auto& a, auto& b) { return a.first > b.first; });
        for (auto& d : byDepth) order.push_back(d.second);
        return order;
    }
};

// partials buffers: tips 0..T-1, internal node n -> T + 2(n-T) + flip;  matrices: node n -> 2n + flip;  scale: node n-T -> 2(n-T)+flip
struct Protocol {
    const Tree& t; int T;
    std::vector<int> pFlip, mFlip, sFlip, pSaved, mSaved, sSaved;
    explicit Protocol(const Tree& tr) : t(tr), T(tr.T) { const int N = 2 * T - 1; pFlip.assign(N, 0); mFlip.assign(N, 0); sFlip.assign(N, 0); }
    int pBuf(int n) const { return n < T ? n : T + 2 * (n - T) + pFlip[n]; }
    int mBuf(int n) const { return 2 * n + mFlip[n]; }
    int sBuf(int n) const { return 2 * (n - T) + sFlip[n]; }
    void store() { pSaved = pFlip; mSaved = mFlip; sSaved = sFlip; }
    void restore() { pFlip = pSaved; mFlip = mSaved; sFlip = sSaved; }
    void emit(const std::vector<int>& nodes, int scaleMode /*0 none 1 write 2 read*/, std::vector<int>& ops) {
        for (int n : nodes) {
            if (scaleMode == 1) sFlip[n] ^= 1;
            ops.insert(ops.end(), {pBuf(n), scaleMode == 1 ? sBuf(n) : -1, scaleMode == 2 ? sBuf(n) : -1, pBuf(t.left[n]), mBuf(t.left[n]), pBuf(t.right[n]), mBuf(t.right[n])});
        }
    }
};

static void scenarioMcmc(int T, bool virt, bool caterpillar, unsigned seed, int steps, bool someTipPartials) {
    std::mt19937 rng(seed);
    static char where[128]; snprintf(where, sizeof where, "mcmc T=%d virt=%d cat=%d seed=%u tipPartials=%d", T, (int)virt, (int)caterpillar, seed, (int)someTipPartials); g_where = where; g_list = 0;
    Tree tree; tree.random(T, rng, caterpillar);
    const int N = 2 * T - 1;
    Harness h; h.init(T, T + 2 * (T - 1), 2 * N, 2 * (T - 1), virt, seed + 1);
    h.mixStepLimit = true;
    for (int i = 0; i < T; i++) { if (someTipPartials && rng() % 7 == 0) h.setTipPartials(i); else h.setTipStates(i); }
    for (int s = 0; s < 2 * N; s++) h.setMatrix(s);
    Protocol pr(tree);
    std::vector<int> all; tree.postOrder(N - 1, all);
    std::vector<int> lvl = tree.levelOrder();
    // full evaluation, no scaling, post-order; then level order with rescaling (write), then read mode
    { std::vector<int> ops; for (int n : all) pr.pFlip[n] ^= 1; pr.emit(all, 0, ops); h.update(ops, 7); }
    { std::vector<int> ops; for (int n : lvl) pr.pFlip[n] ^= 1; pr.emit(lvl, 1, ops); h.update(ops, 7); }
    { std::vector<int> ops; for (int n : lvl) pr.pFlip[n] ^= 1; pr.emit(lvl, 2, ops); h.update(ops, 7); }
    for (int step = 0; step < steps; step++) {
        pr.store();
        // a proposal: change one or two branches -> their matrices and the path(s) to the root
        std::vector<char> dirty(N, 0);
        const int nChanges = 1 + rng() % 2;
        for (int q = 0; q < nChanges; q++) {
            const int n = rng() % (N - 1);
            pr.mFlip[n] ^= 1; h.setMatrix(pr.mBuf(n));
            for (int a = tree.parent[n]; a >= 0; a = tree.parent[a]) dirty[a] = 1;
        }
        std::vector<int> nodes;
        const bool level = rng() % 2;
        for (int n : (level ? lvl : all)) if (dirty[n]) nodes.push_back(n);
        const int mode = step % 9 == 4 ? 1 : 2;      // every ninth proposal recomputes the scale factors of ITS nodes
        if (mode == 1) {                               // BEAST recomputes all of them: full traversal in write mode
            nodes = level ? lvl : all;
        }
        for (int n : nodes) pr.pFlip[n] ^= 1;
        std::vector<int> ops; pr.emit(nodes, mode, ops);
        h.update(ops, 7);
        if (rng() % 3 == 0) pr.restore();              // rejected: the unflipped buffers must still hold their values
        if (rng() % 4 == 0) { std::vector<int> xs; for (int q = 0; q < 3; q++) xs.push_back(T + rng() % (2 * (T - 1))); h.materialise(xs); }
        if (rng() % 11 == 0) { const int tip = rng() % T; if (h.compact[tip]) {
            // new tip data invalidates everything above it: BEAST would re-evaluate; here only the planner hooks are exercised
            h.setTipStates(tip);
            std::vector<int> nodes2; std::vector<char> d2(N, 0);
            for (int a = tree.parent[tip]; a >= 0; a = tree.parent[a]) d2[a] = 1;
            for (int n : all) if (d2[n]) nodes2.push_back(n);
            pr.store(); for (int n : nodes2) pr.pFlip[n] ^= 1;
            std::vector<int> ops2; pr.emit(nodes2, 2, ops2); h.update(ops2, 7);
            // the OTHER flip of those nodes is now stale in BEAST too (it never reads it again before rewriting it)
        } }
    }
    std::vector<int> every; for (int b = T; b < h.nBuf; b++) every.push_back(b);
    h.materialise(every);
    h.compareAll();
    printf("  mcmc T=%d virt=%d cat=%d: %ld lists, %ld micro-ops, %ld stored, %ld holds, %ld memory reads, %ld materialised, %ld slices in %ld waves\n",
           T, (int)virt, (int)caterpillar, h.lists, h.micro, h.stored, h.holds, h.memReads, h.materialised, h.segsTotal, h.waves);
}

// The chain's steady state when a model parameter changes every iteration: the SAME two full-evaluation lists alternate
// (all nodes flip between their two buffers), with partial updates and rejected proposals in between.  The planner
// serves the repeats from its plan cache; every evaluation is still checked against list-order evaluation.
static void scenarioSteady(int T, bool level, unsigned seed) {
    std::mt19937 rng(seed);
    static char where[128]; snprintf(where, sizeof where, "steady T=%d level=%d seed=%u", T, (int)level, seed); g_where = where; g_list = 0;
    Tree tree; tree.random(T, rng, false);
    const int N = 2 * T - 1;
    Harness h; h.init(T, T + 2 * (T - 1), 2 * N, 2 * (T - 1), true, seed + 1);
    { static const int chunks[4] = {0, 8, 20, 64}; h.fixedChunk = chunks[seed % 4]; }
    for (int i = 0; i < T; i++) h.setTipStates(i);
    for (int s = 0; s < 2 * N; s++) h.setMatrix(s);
    Protocol pr(tree);
    std::vector<int> all; tree.postOrder(N - 1, all);
    std::vector<int> lvl = tree.levelOrder();
    const std::vector<int>& order = level ? lvl : all;
    for (int it = 0; it < 14; it++) {
        for (int n : order) pr.pFlip[n] ^= 1;
        for (int n = 0; n < N - 1; n++) { pr.mFlip[n] ^= 1; h.setMatrix(pr.mBuf(n)); }        // new branch matrices every time
        std::vector<int> ops; pr.emit(order, it % 5 == 0 ? 1 : 2, ops);
        h.update(ops, 7);
        h.compareAll();
        if (it % 3 == 1) {                                   // a proposal on one branch, accepted or not
            pr.store();
            std::vector<char> dirty(N, 0);
            const int n = rng() % (N - 1);
            pr.mFlip[n] ^= 1; h.setMatrix(pr.mBuf(n));
            for (int a = tree.parent[n]; a >= 0; a = tree.parent[a]) dirty[a] = 1;
            std::vector<int> nodes;
            for (int x : order) if (dirty[x]) nodes.push_back(x);
            for (int x : nodes) pr.pFlip[x] ^= 1;
            std::vector<int> ops2; pr.emit(nodes, 2, ops2);
            h.update(ops2, 7);
            if (it != 10) pr.restore();                      // rejected (the accepted one changes the lists that follow)
            h.compareAll();
        }
        if (it == 7) { std::vector<int> xs; for (int q = 0; q < 4; q++) xs.push_back(T + rng() % (2 * (T - 1))); h.materialise(xs); }
        if (it == 9) h.setTipStates(rng() % T);                // new tip data: the next evaluation recomputes everything anyway
    }
    std::vector<int> every; for (int b = T; b < h.nBuf; b++) every.push_back(b);
    h.materialise(every);
    h.compareAll();
    if (T >= 16 && (h.pl.cacheHits < 4 || h.fastReplays < 3)) { fprintf(stderr, "steady T=%d: only %ld plan-cache hits, %ld fast replays\n", T, h.pl.cacheHits, h.fastReplays); exit(1); }
    printf("  steady T=%d level=%d: %ld lists, %ld plan-cache hits, %ld micro-ops, %ld stored\n", T, (int)level, h.lists, h.pl.cacheHits, h.micro, h.stored);
}

static void scenarioPartitions(unsigned seed, int T) {
    g_where = "partitions"; g_list = 0;
    std::mt19937 rng(seed);
    Tree tree; tree.random(T, rng, false);
    const int N = 2 * T - 1;
    Harness h; h.init(T, T + 2 * (T - 1), 2 * N * 3, 2 * (T - 1) * 3, true, seed);
    h.parts = 3; h.partStart = {0, 2, 5}; h.partEnd = {2, 5, P};
    h.pl.setPartitionCount(3);               // definitions per (buffer, partition); more snapshot slots
    for (World* w : {&h.truth, &h.plan}) w->mats.assign(h.pl.matrixSlots(), std::vector<double>((size_t)C * 16, 0.0));
    for (int i = 0; i < T; i++) h.setTipStates(i);
    for (int s = 0; s < 2 * N * 3; s++) h.setMatrix(s);
    std::vector<int> all; tree.postOrder(N - 1, all);
    for (int rep = 0; rep < 6; rep++) {
        std::vector<int> ops;
        const int mode = rep == 1 ? 1 : rep == 0 ? 0 : 2;
        for (int part = 0; part < 3; part++)
            for (int n : all) {
                if (rep > 2 && rng() % 3 == 0 && n != N - 1) continue;   // partial lists (a node above a skipped one re-reads it)
                auto pb = [&](int x) { return x < T ? x : T + 2 * (x - T) + (rep & 1); };
                auto pbOld = [&](int x) { return x < T ? x : T + 2 * (x - T) + ((rep & 1) ^ 0); };
                (void)pbOld;
                const int s = (2 * (n - T)) * 3 + part;
                ops.insert(ops.end(), {pb(n), mode == 1 ? s : -1, mode == 2 ? s : -1, pb(tree.left[n]), (2 * tree.left[n]) * 3 + part,
                                       pb(tree.right[n]), (2 * tree.right[n]) * 3 + part, part, -1});
            }
        if (rep > 2) {   // skipped nodes must exist in this flip: run a full list first in that case
            std::vector<int> full;
            for (int part = 0; part < 3; part++) for (int n : all) {
                auto pb = [&](int x) { return x < T ? x : T + 2 * (x - T) + (rep & 1); };
                const int s = (2 * (n - T)) * 3 + part;
                full.insert(full.end(), {pb(n), -1, 2 > 1 ? s : -1, pb(tree.left[n]), (2 * tree.left[n]) * 3 + part, pb(tree.right[n]), (2 * tree.right[n]) * 3 + part, part, -1});
            }
            h.update(full, 9);
        }
        h.update(ops, 9);
    }
    // per-partition buffer flips, as MultiPartitionDataLikelihoodDelegate's partialBufferHelper[i] does: partial updates of
    // ONE partition at a time on the path to the root, with rejections; definitions of the other partitions must survive
    {
        std::vector<std::vector<int>> flip(3, std::vector<int>(N, 1));       // the last full lists above wrote flip (5 & 1) = 1
        auto pbk = [&](int part, int x) { return x < T ? x : T + 2 * (x - T) + flip[part][x]; };
        for (int step = 0; step < 30; step++) {
            const int part = rng() % 3;
            std::vector<char> dirty(N, 0);
            const int n0 = rng() % (N - 1);
            h.setMatrix((2 * n0) * 3 + part);
            for (int a = tree.parent[n0]; a >= 0; a = tree.parent[a]) dirty[a] = 1;
            std::vector<int> saved = flip[part], ops;
            for (int n : all) if (dirty[n]) flip[part][n] ^= 1;
            for (int n : all) {
                if (!dirty[n]) continue;
                const int sc = (2 * (n - T)) * 3 + part;
                ops.insert(ops.end(), {pbk(part, n), -1, sc, pbk(part, tree.left[n]), (2 * tree.left[n]) * 3 + part,
                                       pbk(part, tree.right[n]), (2 * tree.right[n]) * 3 + part, part, -1});
            }
            h.update(ops, 9);
            if (rng() % 3 == 0) flip[part] = saved;            // rejected
            if (rng() % 5 == 0) { std::vector<int> xs; for (int q = 0; q < 3; q++) xs.push_back(T + rng() % (2 * (T - 1))); h.materialise(xs); }
        }
    }
    std::vector<int> every; for (int b = T; b < h.nBuf; b++) every.push_back(b);
    h.materialise(every);
    h.compareAll();
    printf("  partitions T=%d: %ld lists, %ld micro-ops, %ld stored, %ld materialised\n", T, h.lists, h.micro, h.stored, h.materialised);
}

static void scenarioHazards(unsigned seed) {
    g_where = "hazards"; g_list = 0;
    Harness h; h.init(4, 12, 16, 8, true, seed);
    for (int i = 0; i < 4; i++) h.setTipStates(i);
    for (int s = 0; s < 16; s++) h.setMatrix(s);
    // X(4) = f(0,1); Y(5) = g(X, 2); X(4) = h(2,3) [WAW + WAR]; Z(6) = k(X, Y); Z(6) = k'(Z, 3) [in place]
    std::vector<int> ops = {4, -1, -1, 0, 0, 1, 1,   5, -1, -1, 4, 2, 2, 3,   4, -1, -1, 2, 4, 3, 5,   6, -1, -1, 4, 6, 5, 7,   6, -1, -1, 6, 8, 3, 9};
    h.update(ops, 7);
    // the same scale buffer written by one op and read by a later one of the same list
    std::vector<int> ops2 = {7, 0, -1, 0, 0, 1, 1,   8, -1, 0, 7, 2, 2, 3,   9, 1, -1, 8, 4, 7, 5};
    h.update(ops2, 7);
    std::vector<int> every; for (int b = 4; b < 12; b++) every.push_back(b);
    h.materialise(every);
    printf("  hazards: %ld lists, %ld micro-ops\n", h.lists, h.micro);
}

// A DYNAMIC chain's rescaling cycles (BeagleTreeLikelihood.java:1059-1113): full evaluations in read mode alternating between the two
// buffer parities, every few of them one in WRITE mode into the other scale-buffer set, after which the read-mode lists name that set —
// closed lists throughout, so from the second cycle on every plan comes out of the cache, and right behind a rescaling evaluation it is
// replayed over definitions another entry left behind (same leaves, other scale buffers: WalkPlanner::replay edits the user lists in place).
static void scenarioRescaleCycles(int T, unsigned seed, int cycles, int chunk) {
    std::mt19937 rng(seed);
    static char where[96]; snprintf(where, sizeof where, "rescale cycles T=%d seed=%u chunk=%d", T, seed, chunk); g_where = where; g_list = 0;
    Tree tree; tree.random(T, rng, false);
    const int N = 2 * T - 1;
    Harness h; h.init(T, T + 2 * (T - 1), 2 * N, 2 * (T - 1), true, seed + 1);
    h.fixedChunk = chunk;
    for (int i = 0; i < T; i++) h.setTipStates(i);
    for (int s = 0; s < 2 * N; s++) h.setMatrix(s);
    Protocol pr(tree);
    std::vector<int> lvl = tree.levelOrder();
    auto evaluation = [&](int mode) {
        for (int n : lvl) pr.pFlip[n] ^= 1;
        for (int n = 0; n < N - 1; n++) { pr.mFlip[n] ^= 1; h.setMatrix(pr.mBuf(n)); }
        std::vector<int> ops; pr.emit(lvl, mode, ops); h.update(ops, 7);
    };
    for (int c = 0; c < cycles; c++) {
        evaluation(1);                                   // the rescaling evaluation: new factors into the other set
        for (int k = 0; k < 5; k++) evaluation(2);
        if (c % 2) {                                     // ... and a branch move with its rejection in between, as a chain has them
            pr.store();
            const int n = rng() % (N - 1);
            pr.mFlip[n] ^= 1; h.setMatrix(pr.mBuf(n));
            std::vector<int> path;
            for (int x : lvl) { bool on = false; for (int a = tree.parent[n]; a >= 0; a = tree.parent[a]) on = on || a == x; if (on) path.push_back(x); }
            for (int x : path) pr.pFlip[x] ^= 1;
            std::vector<int> ops; pr.emit(path, 2, ops); h.update(ops, 7);
            pr.restore();
        }
    }
}

// `plan_check bench`: what the planner costs the host on a list it has not seen — config A's shape (1000 taxa, full evaluation, read
// mode, the engine's settings for 1e5 patterns) with a few accepted branch moves between the lists, as a chain produces them
// (bench.py partial_update.full_evaluation_on_a_new_list; profiles/r05_experiments.txt 12).  Microseconds per plan() call.
static void benchPlanner() {
    std::mt19937 rng(7);
    const int T = 1000, N = 2 * T - 1;
    Tree tree; tree.random(T, rng, false);
    WalkPlanner pl;
    pl.init(T + 2 * (T - 1), T, 2 * N, 2 * (T - 1), 24, true, 3);
    pl.launchMachines = 1024.0 / 782.0; pl.chunkTopOps = 16;
    for (int i = 0; i < T; i++) pl.setCompactTip(i, true);
    Protocol pr(tree);
    std::vector<int> lvl = tree.levelOrder();
    Plan out;
    { std::vector<int> ops; for (int n : lvl) pr.pFlip[n] ^= 1; pr.emit(lvl, 1, ops); pl.plan(ops.data(), (int)ops.size() / 7, 7, 1, true, out, 150); }
    double total = 0.0; long micro = 0; const int reps = 200;
    for (int it = 0; it < reps; it++) {
        for (int q = 0; q < 3; q++) {                               // three accepted branch moves: their paths flip
            const int n = rng() % (N - 1);
            pr.mFlip[n] ^= 1;
            std::vector<int> nodes;
            for (int a = tree.parent[n]; a >= 0; a = tree.parent[a]) nodes.push_back(a);
            std::reverse(nodes.begin(), nodes.end());
            std::vector<int> path;                                  // (list order: children before parents)
            for (int x : lvl) if (std::find(nodes.begin(), nodes.end(), x) != nodes.end()) path.push_back(x);
            for (int x : path) pr.pFlip[x] ^= 1;
            std::vector<int> ops2; pr.emit(path, 2, ops2);
            pl.plan(ops2.data(), (int)ops2.size() / 7, 7, 1, true, out, 0);
        }
        for (int n : lvl) pr.pFlip[n] ^= 1;
        for (int n = 0; n < N - 1; n++) pr.mFlip[n] ^= 1;
        std::vector<int> ops; pr.emit(lvl, 2, ops);
        const long hits = pl.cacheHits;
        const auto t0 = std::chrono::steady_clock::now();
        pl.plan(ops.data(), (int)ops.size() / 7, 7, 1, true, out, 150);
        total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (pl.cacheHits != hits) { fprintf(stderr, "bench: the list was found in the cache\n"); exit(1); }
        micro += (long)pl.planned->prog.size();
    }
    printf("planner, %d-taxon full evaluation on a new list: %.1f us per plan() (%ld micro-operations each)\n", T, total / reps, micro / reps);
}

// `plan_check bench-replay`: what a plan out of the cache costs when the definitions its list leaves behind are another entry's — the
// evaluations of a DYNAMIC chain right behind a rescaling evaluation (the scale-buffer set has flipped: every definition of the tree is
// registered again under the other entry; profiles/r06_experiments.txt 23).  Four lists: two buffer parities x two scale sets, read mode.
static void benchReplay() {
    std::mt19937 rng(7);
    const int T = 1000, N = 2 * T - 1;
    Tree tree; tree.random(T, rng, false);
    WalkPlanner pl;
    pl.init(T + 2 * (T - 1), T, 2 * N, 2 * (T - 1), 24, true, 3);
    pl.launchMachines = 1024.0 / 782.0; pl.chunkTopOps = 16;
    for (int i = 0; i < T; i++) pl.setCompactTip(i, true);
    Protocol pr(tree);
    std::vector<int> lvl = tree.levelOrder();
    std::vector<std::vector<int>> lists;
    for (int sset = 0; sset < 2; sset++) {
        for (int n : lvl) pr.sFlip[n] = sset;
        for (int parity = 0; parity < 2; parity++) {
            for (int n : lvl) pr.pFlip[n] = parity;
            for (int n = 0; n < N - 1; n++) pr.mFlip[n] = parity;
            std::vector<int> ops; pr.emit(lvl, 2, ops); lists.push_back(ops);
        }
    }
    Plan out;
    for (auto& ops : lists) pl.plan(ops.data(), (int)ops.size() / 7, 7, 1, true, out, 150);
    double same = 0.0, other = 0.0; const int reps = 300;
    for (int it = 0; it < reps; it++) {
        const int sset = it & 1;
        for (int k = 0; k < 4; k++) {            // parity 0, 1 (both behind the flip: the other entry's definitions), then 0, 1 again (their own)
            std::vector<int>& ops = lists[(size_t)(2 * sset + (k & 1))];
            const long hits = pl.cacheHits;
            const auto t0 = std::chrono::steady_clock::now();
            pl.plan(ops.data(), (int)ops.size() / 7, 7, 1, true, out, 150);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (pl.cacheHits == hits) { fprintf(stderr, "bench-replay: the list was planned again\n"); exit(1); }
            (k < 2 ? other : same) += us;
            if (it < 3) checkUserLists(pl, T + 2 * (T - 1), 2 * (T - 1), "bench-replay");
        }
    }
    printf("planner, %d-taxon full evaluation out of the cache: %.1f us behind a flipped scale-buffer set, %.1f us in steady state\n", T, other / (2 * reps), same / (2 * reps));
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "bench")) { benchPlanner(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "bench-replay")) { benchReplay(); return 0; }
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    for (int r = 0; r < reps; r++) {
        for (int T : {2, 3, 5, 8, 13, 40, 150}) {
            scenarioMcmc(T, true, false, 1000 * r + T, 60, false);
            scenarioMcmc(T, true, false, 2000 * r + T, 40, true);
            scenarioMcmc(T, false, false, 3000 * r + T, 20, false);
            scenarioMcmc(T, true, true, 4000 * r + T, 30, false);
        }
        for (int T : {9, 40, 150, 600}) { scenarioSteady(T, false, 6000 * r + T); scenarioSteady(T, true, 7000 * r + T); }
        scenarioPartitions(77 + r, 9); scenarioPartitions(177 + r, 40);
        scenarioHazards(5 + r);
    }
    P = 2; C = 1;
    scenarioMcmc(5000, true, true, 9, 3, false);     // 4999 dependency levels
    scenarioMcmc(5000, false, true, 11, 2, false);
    scenarioMcmc(3000, true, false, 10, 5, false);
    printf("read-mode folding: %ld programs, %ld factor reads became %ld (%ld members)\n", foldedPlans, unfoldedReads, foldedPays, foldedMembers);
    if (foldedPlans < 100 || foldedPays * 3 > unfoldedReads * 2) { fprintf(stderr, "read-mode folding was hardly exercised\n"); return 1; }
    for (int T : {17, 40, 150}) { scenarioRescaleCycles(T, 77 + T, 6, 8); scenarioRescaleCycles(T, 99 + T, 4, 0); }
    printf("definitions taken over in place by a replayed plan (same leaves, the other scale-buffer set): %ld\n", g_replayInPlace);
    if (g_replayInPlace < 500) { fprintf(stderr, "the in-place replay was hardly exercised\n"); return 1; }
    printf("programs executed on tickets (slices as a forest, leaves in reverse launch order): %ld\n", g_ticketRuns);
    if (g_ticketRuns < 100) { fprintf(stderr, "the ticket form of the one-launch walk was hardly exercised\n"); return 1; }
    printf("plan_check: OK\n");
    return 0;
}
