// tree_likelihood.cpp — host-side mirror of the reference's BEAGLE caller, in C++ (no JDK in the
// build image), written against the engine's C ABI (include/beagle_mi355.h) only.
//
// It reproduces the bookkeeping BEAST does ABOVE the beagle.Beagle surface, so that tests and the
// benchmark drive the engine with exactly the call sequence the reference issues:
//
//   BufferIndexHelper         src/dr/evomodel/treedatalikelihood/BufferIndexHelper.java:40-116
//   calculateLogLikelihood    src/dr/evomodel/treelikelihood/BeagleTreeLikelihood.java:863-1130
//                             (rescaling policy :883-910, call order :944-1050, underflow retry :1059-1113)
//   traverse (post-order)     BeagleTreeLikelihood.java:1202-1322
//   reverse level order       src/dr/evomodel/treedatalikelihood/LikelihoodTreeTraversal.java:133-205
//   store / restore           BeagleTreeLikelihood.java:816-851 (index flips only, no data copies)
//   instance shape            BeagleTreeLikelihood.java:193-203, 420-433
//   branch length             rate * (parent height - node height), BeagleTreeLikelihood.java:1221-1231
//   rescaling schemes         src/dr/evomodel/treelikelihood/PartialsRescalingScheme.java:34-42
//
// It holds no likelihood arithmetic: every O(patterns) operation is a call through BeagleApi.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/beagle_mi355.h"

namespace {

class BufferIndexHelper {
public:
    BufferIndexHelper() : minIndex(0), dbl(0) {}
    BufferIndexHelper(int maxIndexValue, int minIndexValue)
        : minIndex(minIndexValue), dbl(maxIndexValue - minIndexValue),
          offsets(dbl, 0), stored(dbl, 0), flipped(dbl, 0) {}
    int getBufferCount() const { return 2 * dbl + minIndex; }
    void flipOffset(int i) {
        int k = i - minIndex;
        if (!flipped[k]) { offsets[k] = dbl - offsets[k]; flipped[k] = 1; }  // only once before accept/reject
    }
    int getOffsetIndex(int i) const { return i < minIndex ? i : offsets[i - minIndex] + i; }
    void storeState() { std::fill(flipped.begin(), flipped.end(), 0); stored = offsets; }
    void restoreState() { offsets.swap(stored); std::fill(flipped.begin(), flipped.end(), 0); }
private:
    int minIndex, dbl;
    std::vector<int> offsets, stored;
    std::vector<char> flipped;
};

enum Scheme { SCHEME_NONE = 0, SCHEME_ALWAYS = 1, SCHEME_DYNAMIC = 2, SCHEME_DELAYED = 3 };
enum Traversal { POST_ORDER = 0, REVERSE_LEVEL_ORDER = 1 };

struct TreeLikelihood {
    const BeagleApi* api = nullptr;
    int inst = -1;
    int tipCount = 0, nodeCount = 0, internalNodeCount = 0, S = 0, P = 0, C = 0;
    int root = -1;
    std::vector<int> parent, left, right;
    std::vector<double> height, branchRate;
    std::vector<char> updateNode;
    bool updateSubstitutionModel = true, updateSiteModel = true;
    bool likelihoodKnown = false;
    double logLikelihood = 0.0, storedLogLikelihood = 0.0;
    bool storedLikelihoodKnown = false;

    BufferIndexHelper partialBufferHelper, scaleBufferHelper, matrixBufferHelper, eigenBufferHelper;
    std::vector<int> scaleBufferIndices, storedScaleBufferIndices;

    int scheme = SCHEME_DYNAMIC;
    bool delayRescalingUntilUnderflow = true;
    int rescalingFrequency = 100;          // beagle.rescale default, BeagleTreeLikelihood.java:116
    static const int RESCALE_TIMES = 1;    // :117
    int rescalingCount = 0, rescalingCountInner = 0;
    bool everUnderflowed = false, useScaleFactors = false, recomputeScaleFactors = false;
    int traversal = POST_ORDER;

    std::vector<double> U, Uinv, lambda, freqs, catRates, catWeights;

    std::vector<int> branchUpdateIndices, operations, probIdx;
    std::vector<double> branchLengths;
    int branchUpdateCount = 0, operationCount = 0;
    std::vector<std::vector<int>> levelOps;     // level -> flat op tuples (reverse level order); kept across evaluations (no allocation in the steady state)
    int levelsUsed = 0;

    long totalOperationCount = 0, totalMatrixUpdateCount = 0, totalEvaluations = 0, totalRescaleRetries = 0;
    int lastError = 0;
    // development (BTL_TIMING=1): host wall clock per phase of an evaluation, microseconds summed over the evaluations
    // since the last btlTimings(reset): traversal, model uploads, updateTransitionMatrices, updatePartials, scale-factor
    // calls, weights + frequencies, root (enqueue + wait)
    bool timing = getenv("BTL_TIMING") != nullptr;
    double phaseUs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef std::chrono::steady_clock Clock;
    Clock::time_point mark;
    void tick() { if (timing) mark = Clock::now(); }
    void tock(int k) { if (timing) { const Clock::time_point n = Clock::now(); phaseUs[k] += std::chrono::duration<double, std::micro>(n - mark).count(); mark = n; } }

    void updateAllNodes() { std::fill(updateNode.begin(), updateNode.end(), 1); likelihoodKnown = false; }

    // BeagleTreeLikelihood.traverse :1202-1322 (`levels`: per-depth lists as LikelihoodTreeTraversal.traverseLevelOrder :151-195
    // collects them), as two flat passes over orders fixed by the topology (setTree) instead of a recursion per evaluation: the
    // recursion records a node's branch BEFORE it descends and its operation AFTER both children — pre-order and post-order of
    // the same walk (left child first) — so the lists come out entry for entry as the reference builds them.
    std::vector<int> preOrder, postOrder, depth;
    std::vector<char> subtreeUpdated;
    void buildOrders() {
        preOrder.clear(); postOrder.clear();
        depth.assign(nodeCount, 0);
        std::vector<std::pair<int, int>> st;          // (node, next child to visit)
        st.emplace_back(root, 0);
        preOrder.push_back(root);
        while (!st.empty()) {
            const int node = st.back().first, k = st.back().second;
            if (node < tipCount || k == 2) { postOrder.push_back(node); st.pop_back(); continue; }
            st.back().second = k + 1;
            const int c = k == 0 ? left[node] : right[node];
            depth[c] = depth[node] + 1;
            preOrder.push_back(c);
            st.emplace_back(c, 0);
        }
        subtreeUpdated.assign(nodeCount, 0);
        int maxDepth = 0;
        for (int d : depth) maxDepth = std::max(maxDepth, d);
        levelOps.resize(maxDepth + 1);
    }

    void runTraversal(bool flip) {
        branchUpdateCount = 0;
        operationCount = 0;
        const bool levels = traversal != POST_ORDER;
        if (levels) { for (int l = 0; l < levelsUsed; l++) levelOps[l].clear(); levelsUsed = 0; }
        for (int node : preOrder) {
            subtreeUpdated[node] = 0;
            if (parent[node] >= 0 && updateNode[node]) {
                const double branchLength = branchRate[node] * (height[parent[node]] - height[node]);
                if (branchLength < 0.0) { lastError = BEAGLE_ERROR_OUT_OF_RANGE; }
                if (flip) matrixBufferHelper.flipOffset(node);
                branchUpdateIndices[branchUpdateCount] = node;
                branchLengths[branchUpdateCount] = branchLength;
                branchUpdateCount++;
                subtreeUpdated[node] = 1;
            }
        }
        for (int node : postOrder) {
            if (node < tipCount) continue;
            const int c1 = left[node], c2 = right[node];
            if (!(subtreeUpdated[c1] || subtreeUpdated[c2])) continue;
            subtreeUpdated[node] = 1;
            if (flip) partialBufferHelper.flipOffset(node);
            int* op;
            if (levels) {
                const int level = depth[node];
                if (level >= levelsUsed) levelsUsed = level + 1;
                std::vector<int>& v = levelOps[level];
                v.resize(v.size() + BEAGLE_OP_COUNT);
                op = &v[v.size() - BEAGLE_OP_COUNT];
            } else {
                op = &operations[(size_t)operationCount * BEAGLE_OP_COUNT];
                operationCount++;
            }
            op[0] = partialBufferHelper.getOffsetIndex(node);
            if (useScaleFactors) {
                const int n = node - tipCount;
                if (recomputeScaleFactors) {
                    scaleBufferHelper.flipOffset(n);
                    scaleBufferIndices[n] = scaleBufferHelper.getOffsetIndex(n);
                    op[1] = scaleBufferIndices[n];     // write new scale factors
                    op[2] = BEAGLE_OP_NONE;
                } else {
                    op[1] = BEAGLE_OP_NONE;
                    op[2] = scaleBufferIndices[n];     // read existing scale factors
                }
            } else {
                op[1] = BEAGLE_OP_NONE;
                op[2] = BEAGLE_OP_NONE;
            }
            op[3] = partialBufferHelper.getOffsetIndex(c1);
            op[4] = matrixBufferHelper.getOffsetIndex(c1);
            op[5] = partialBufferHelper.getOffsetIndex(c2);
            op[6] = matrixBufferHelper.getOffsetIndex(c2);
        }
        if (levels)
            for (int l = levelsUsed - 1; l >= 0; l--) {                          // deepest level first
                const std::vector<int>& v = levelOps[l];
                if (v.empty()) continue;
                std::memcpy(&operations[(size_t)operationCount * BEAGLE_OP_COUNT], v.data(), v.size() * sizeof(int));
                operationCount += (int)(v.size() / BEAGLE_OP_COUNT);
            }
    }

    // calculateLogLikelihood (BeagleTreeLikelihood.java:863-1130) in three phases, so that a pattern-sharded
    // multi-GPU host can put ONE all-reduce between attempt() and finish():
    //   prepare()  rescaling policy :883-910, traverse :944, model/rate/matrix updates :953-974
    //   attempt()  updatePartials :1003, scale-factor accumulation :1013-1026, weights/frequencies :1029-1030,
    //              root integration :1038  (deviceOut != null: the sum stays on the device)
    //   finish()   NaN/Inf handling and the rescale-and-retry decision :1059-1113
    bool firstRescaleAttempt = true;

    int prepare() {
        recomputeScaleFactors = false;
        if (!delayRescalingUntilUnderflow || everUnderflowed) {
            if (scheme == SCHEME_ALWAYS || scheme == SCHEME_DELAYED) {
                useScaleFactors = true;
                recomputeScaleFactors = true;
            } else if (scheme == SCHEME_DYNAMIC) {
                useScaleFactors = true;
                if (rescalingCount > rescalingFrequency) { rescalingCount = 0; rescalingCountInner = 0; }
                if (rescalingCountInner < RESCALE_TIMES) {
                    recomputeScaleFactors = true;
                    updateAllNodes();
                    rescalingCountInner++;
                }
                rescalingCount++;
            }
        }
        if (scheme == SCHEME_NONE) { useScaleFactors = false; recomputeScaleFactors = false; }

        tick();
        runTraversal(true);
        tock(0);

        int rc;
        if (updateSubstitutionModel) {
            eigenBufferHelper.flipOffset(0);
            rc = api->setEigenDecomposition(inst, eigenBufferHelper.getOffsetIndex(0), U.data(), Uinv.data(), lambda.data());
            if (rc) return lastError = rc;
        }
        if (updateSiteModel) {
            rc = api->setCategoryRates(inst, catRates.data());
            if (rc) return lastError = rc;
        }
        tock(1);
        if (branchUpdateCount > 0) {
            probIdx.resize(branchUpdateCount);
            for (int i = 0; i < branchUpdateCount; i++) probIdx[i] = matrixBufferHelper.getOffsetIndex(branchUpdateIndices[i]);
            rc = api->updateTransitionMatrices(inst, eigenBufferHelper.getOffsetIndex(0), probIdx.data(), nullptr, nullptr,
                                               branchLengths.data(), branchUpdateCount);
            if (rc) return lastError = rc;
            totalMatrixUpdateCount += branchUpdateCount;
        }
        tock(2);
        firstRescaleAttempt = true;
        return 0;
    }

    // pattern-sharded job, one process per GPU: the root sum is all-reduced INSIDE the engine (include/beagle_mi355.h
    // beagleMi355CalculateRootLogLikelihoodsAllReduce) and every rank gets the global value, so the whole evaluation —
    // including the rescale-and-retry decision, taken on that global value by every rank alike — is one call here
    bool engineCollective = false;

    int attempt(void* deviceOut, double* hostOut) {
        tick();
        int rc = api->updatePartials(inst, operations.data(), operationCount, BEAGLE_OP_NONE);
        if (rc) return lastError = rc;
        totalOperationCount += operationCount;
        tock(3);

        const int rootIndex = partialBufferHelper.getOffsetIndex(root);
        int cumulateScaleBufferIndex = BEAGLE_OP_NONE;
        if (useScaleFactors) {
            if (recomputeScaleFactors) {
                scaleBufferHelper.flipOffset(internalNodeCount);
                cumulateScaleBufferIndex = scaleBufferHelper.getOffsetIndex(internalNodeCount);
                rc = api->resetScaleFactors(inst, cumulateScaleBufferIndex);
                if (rc) return lastError = rc;
                rc = api->accumulateScaleFactors(inst, scaleBufferIndices.data(), internalNodeCount, cumulateScaleBufferIndex);
                if (rc) return lastError = rc;
            } else {
                cumulateScaleBufferIndex = scaleBufferHelper.getOffsetIndex(internalNodeCount);
            }
        }
        tock(4);
        // "these could be set only when they change but store/restore would need to be considered" (:1028)
        rc = api->setCategoryWeights(inst, 0, catWeights.data());
        if (rc) return lastError = rc;
        rc = api->setStateFrequencies(inst, 0, freqs.data());
        if (rc) return lastError = rc;
        tock(5);

        const int zero = 0;
        if (deviceOut) {
            if (!api->calculateRootLogLikelihoodsDevice) return lastError = BEAGLE_ERROR_NO_IMPLEMENTATION;
            rc = api->calculateRootLogLikelihoodsDevice(inst, rootIndex, 0, 0, cumulateScaleBufferIndex, deviceOut);
            if (rc) return lastError = rc;
        } else if (engineCollective) {
            if (!api->calculateRootLogLikelihoodsAllReduce) return lastError = BEAGLE_ERROR_NO_IMPLEMENTATION;
            double sum = 0.0;
            rc = api->calculateRootLogLikelihoodsAllReduce(inst, rootIndex, 0, 0, cumulateScaleBufferIndex, &sum);
            if (rc != 0 && rc != BEAGLE_ERROR_FLOATING_POINT) return lastError = rc;
            *hostOut = sum;
        } else {
            double sum = 0.0;
            rc = api->calculateRootLogLikelihoods(inst, &rootIndex, &zero, &zero, &cumulateScaleBufferIndex, 1, &sum);
            if (rc != 0 && rc != BEAGLE_ERROR_FLOATING_POINT) return lastError = rc;   // BeagleJNIImpl tolerates -8
            *hostOut = sum;
        }
        tock(6);
        totalEvaluations++;
        return 0;
    }

    // returns true when the evaluation is complete, false when attempt() must run again (rescaling retry)
    bool finish(double& logL) {
        if (std::isnan(logL) || std::isinf(logL)) {
            everUnderflowed = true;
            logL = -INFINITY;
            if (firstRescaleAttempt && (delayRescalingUntilUnderflow || scheme == SCHEME_DELAYED) && scheme != SCHEME_NONE) {
                useScaleFactors = true;
                recomputeScaleFactors = true;
                updateAllNodes();
                // traverse again without flipping the partials (overwrite the failed attempt);
                // scale buffer indices are flipped because they are being recomputed (:1094-1099)
                runTraversal(false);
                firstRescaleAttempt = false;
                totalRescaleRetries++;
                return false;
            }
        }
        std::fill(updateNode.begin(), updateNode.end(), 0);
        updateSubstitutionModel = false;
        updateSiteModel = false;
        return true;
    }

    double calculateLogLikelihood() {
        if (prepare()) return NAN;
        double logL = NAN;
        do {
            if (attempt(nullptr, &logL)) return NAN;
        } while (!finish(logL));
        return logL;
    }

    double getLogLikelihood() {
        if (!likelihoodKnown) { logLikelihood = calculateLogLikelihood(); likelihoodKnown = true; }
        return logLikelihood;
    }

    void storeState() {
        partialBufferHelper.storeState();
        matrixBufferHelper.storeState();
        eigenBufferHelper.storeState();
        if (useScaleFactors) {
            scaleBufferHelper.storeState();
            storedScaleBufferIndices = scaleBufferIndices;
        }
        storedLikelihoodKnown = likelihoodKnown;
        storedLogLikelihood = logLikelihood;
    }
    void restoreState() {
        updateSiteModel = true;   // :834
        partialBufferHelper.restoreState();
        matrixBufferHelper.restoreState();
        eigenBufferHelper.restoreState();
        if (useScaleFactors) {
            scaleBufferHelper.restoreState();
            scaleBufferIndices.swap(storedScaleBufferIndices);
        }
        likelihoodKnown = storedLikelihoodKnown;
        logLikelihood = storedLogLikelihood;
    }
};

}  // namespace

extern "C" {

// Create the host object and its engine instance with the reference's buffer counts
// (BeagleTreeLikelihood.java:193-203): partials = tips + 2*internal, compact = tips,
// eigen = 2, matrices = 2*nodes, scale buffers = 2*(internal+1).
void* btlCreate(const BeagleApi* api, int tipCount, int stateCount, int patternCount, int categoryCount,
                int rescalingScheme, int delayRescaling, int traversal,
                const int* resourceList, int resourceCount, long preferenceFlags, long requirementFlags) {
    if (!api || tipCount < 2) return nullptr;
    TreeLikelihood* t = new TreeLikelihood();
    t->api = api;
    t->tipCount = tipCount; t->nodeCount = 2 * tipCount - 1; t->internalNodeCount = tipCount - 1;
    t->S = stateCount; t->P = patternCount; t->C = categoryCount;
    t->scheme = rescalingScheme; t->delayRescalingUntilUnderflow = delayRescaling != 0; t->traversal = traversal;
    t->partialBufferHelper = BufferIndexHelper(t->nodeCount, tipCount);
    t->scaleBufferHelper = BufferIndexHelper(t->internalNodeCount + 1, 0);
    t->matrixBufferHelper = BufferIndexHelper(t->nodeCount, 0);
    t->eigenBufferHelper = BufferIndexHelper(1, 0);
    t->scaleBufferIndices.assign(t->internalNodeCount, 0);
    t->storedScaleBufferIndices.assign(t->internalNodeCount, 0);
    t->parent.assign(t->nodeCount, -1); t->left.assign(t->nodeCount, -1); t->right.assign(t->nodeCount, -1);
    t->height.assign(t->nodeCount, 0.0); t->branchRate.assign(t->nodeCount, 1.0);
    t->updateNode.assign(t->nodeCount, 1);
    t->branchUpdateIndices.assign(t->nodeCount, 0); t->branchLengths.assign(t->nodeCount, 0.0);
    t->operations.assign((size_t)t->internalNodeCount * BEAGLE_OP_COUNT, 0);
    t->U.assign((size_t)stateCount * stateCount, 0.0); t->Uinv = t->U; t->lambda.assign((size_t)stateCount * ((requirementFlags & (1L << 5)) ? 2 : 1), 0.0);   // EIGEN_COMPLEX: S real parts, then S imaginary parts (ComplexSubstitutionModel.java:121-135)
    t->freqs.assign(stateCount, 1.0 / stateCount);
    t->catRates.assign(categoryCount, 1.0); t->catWeights.assign(categoryCount, 1.0 / categoryCount);
    BeagleInstanceDetails details;
    std::memset(&details, 0, sizeof(details));
    t->inst = api->createInstance(tipCount, t->partialBufferHelper.getBufferCount(), tipCount, stateCount, patternCount,
                                  t->eigenBufferHelper.getBufferCount(), t->matrixBufferHelper.getBufferCount(), categoryCount,
                                  t->scaleBufferHelper.getBufferCount(), resourceList, resourceCount,
                                  preferenceFlags, requirementFlags, &details);
    if (t->inst < 0) { delete t; return nullptr; }
    return t;
}

void btlDestroy(void* h) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    if (!t) return;
    if (t->inst >= 0) t->api->finalizeInstance(t->inst);
    delete t;
}

int btlInstance(void* h) { return ((TreeLikelihood*)h)->inst; }

// Tree as arrays over node numbers: tips 0..T-1, internal T..2T-2 (BeagleTreeLikelihood.java:484-487).
int btlSetTree(void* h, const int* left, const int* right, const double* heights, int root) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    std::fill(t->parent.begin(), t->parent.end(), -1);
    for (int n = 0; n < t->nodeCount; n++) {
        t->left[n] = left[n]; t->right[n] = right[n]; t->height[n] = heights[n];
        if (n >= t->tipCount) {
            if (left[n] < 0 || left[n] >= t->nodeCount || right[n] < 0 || right[n] >= t->nodeCount) return BEAGLE_ERROR_OUT_OF_RANGE;
            t->parent[left[n]] = n; t->parent[right[n]] = n;
        }
    }
    if (root < t->tipCount || root >= t->nodeCount || t->parent[root] != -1) return BEAGLE_ERROR_OUT_OF_RANGE;
    t->root = root;
    t->buildOrders();
    t->updateAllNodes();
    return 0;
}

int btlSetTipStates(void* h, int tip, const int* states) {
    TreeLikelihood* t = (TreeLikelihood*)h; t->likelihoodKnown = false;
    return t->api->setTipStates(t->inst, tip, states);
}
int btlSetTipPartials(void* h, int tip, const double* partials) {
    TreeLikelihood* t = (TreeLikelihood*)h; t->likelihoodKnown = false;
    return t->api->setTipPartials(t->inst, tip, partials);
}
int btlSetPatternWeights(void* h, const double* w) {
    TreeLikelihood* t = (TreeLikelihood*)h; t->likelihoodKnown = false;
    return t->api->setPatternWeights(t->inst, w);
}

// model changed -> all matrices dirty (AbstractTreeLikelihood.handleModelChangedEvent: updateAllNodes)
int btlSetSubstitutionModel(void* h, const double* U, const double* Uinv, const double* lambda, const double* freqs) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    std::memcpy(t->U.data(), U, t->U.size() * sizeof(double));
    std::memcpy(t->Uinv.data(), Uinv, t->Uinv.size() * sizeof(double));
    std::memcpy(t->lambda.data(), lambda, t->lambda.size() * sizeof(double));
    std::memcpy(t->freqs.data(), freqs, t->freqs.size() * sizeof(double));
    t->updateSubstitutionModel = true;
    t->updateAllNodes();
    return 0;
}
int btlSetSiteModel(void* h, const double* rates, const double* weights) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    std::memcpy(t->catRates.data(), rates, t->C * sizeof(double));
    std::memcpy(t->catWeights.data(), weights, t->C * sizeof(double));
    t->updateSiteModel = true;
    t->updateAllNodes();
    return 0;
}
int btlSetBranchRates(void* h, const double* ratePerNode) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    std::memcpy(t->branchRate.data(), ratePerNode, t->nodeCount * sizeof(double));
    t->updateAllNodes();
    return 0;
}
// A node-height move dirties the node and both children (TreeChangedEvent handling,
// AbstractTreeLikelihood.updateNodeAndChildren).
int btlSetNodeHeight(void* h, int node, double height) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    if (node < 0 || node >= t->nodeCount) return BEAGLE_ERROR_OUT_OF_RANGE;
    t->height[node] = height;
    t->updateNode[node] = 1;
    if (node >= t->tipCount) { t->updateNode[t->left[node]] = 1; t->updateNode[t->right[node]] = 1; }
    t->likelihoodKnown = false;
    return 0;
}
// The tree model's own restore after a rejected height move (TreeModel.restoreState: the likelihood's restoreState has put the
// buffer indices back; the heights are the tree's): no node becomes dirty.
int btlRestoreNodeHeight(void* h, int node, double height) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    if (node < 0 || node >= t->nodeCount) return BEAGLE_ERROR_OUT_OF_RANGE;
    t->height[node] = height;
    return 0;
}
int btlMakeDirty(void* h) { ((TreeLikelihood*)h)->updateAllNodes(); return 0; }
int btlSetRescalingFrequency(void* h, int f) { ((TreeLikelihood*)h)->rescalingFrequency = f; return 0; }

double btlGetLogLikelihood(void* h) { return ((TreeLikelihood*)h)->getLogLikelihood(); }
// Phased evaluation for the pattern-sharded multi-GPU path: btlPrepare, then repeat
// { btlAttemptDevice(deviceDouble); <all-reduce the device double, read it>; } until btlFinish(global) == 1.
int btlPrepare(void* h) { return ((TreeLikelihood*)h)->prepare(); }
int btlAttemptDevice(void* h, void* deviceOut) { return ((TreeLikelihood*)h)->attempt(deviceOut, nullptr); }
int btlAttemptHost(void* h, double* localLogL) { return ((TreeLikelihood*)h)->attempt(nullptr, localLogL); }
int btlFinish(void* h, double globalLogL) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    double v = globalLogL;
    if (!t->finish(v)) return 0;
    t->logLikelihood = v; t->likelihoodKnown = true;
    return 1;
}
// 1: getLogLikelihood's root sum is the engine's all-reduce over the ranks of the instance's communicator (beagleMi355CommInit)
int btlSetEngineCollective(void* h, int on) { ((TreeLikelihood*)h)->engineCollective = on != 0; return 0; }
int btlStoreState(void* h) { ((TreeLikelihood*)h)->storeState(); return 0; }
int btlRestoreState(void* h) { ((TreeLikelihood*)h)->restoreState(); return 0; }
int btlGetSiteLogLikelihoods(void* h, double* out) {
    TreeLikelihood* t = (TreeLikelihood*)h; return t->api->getSiteLogLikelihoods(t->inst, out);
}
int btlLastError(void* h) { return ((TreeLikelihood*)h)->lastError; }
int btlRootBufferIndex(void* h) { TreeLikelihood* t = (TreeLikelihood*)h; return t->partialBufferHelper.getOffsetIndex(t->root); }
int btlNodeBufferIndex(void* h, int node) { return ((TreeLikelihood*)h)->partialBufferHelper.getOffsetIndex(node); }
int btlNodeScaleIndex(void* h, int node) { TreeLikelihood* t = (TreeLikelihood*)h; return t->scaleBufferIndices[node - t->tipCount]; }
int btlCumulativeScaleIndex(void* h) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    return t->useScaleFactors ? t->scaleBufferHelper.getOffsetIndex(t->internalNodeCount) : BEAGLE_OP_NONE;
}
// counters: {operations, matrix updates, evaluations, rescale retries, last op count, last branch count,
//            useScaleFactors, recomputeScaleFactors(last)}
int btlCounters(void* h, long* out8) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    out8[0] = t->totalOperationCount; out8[1] = t->totalMatrixUpdateCount; out8[2] = t->totalEvaluations;
    out8[3] = t->totalRescaleRetries; out8[4] = t->operationCount; out8[5] = t->branchUpdateCount;
    out8[6] = t->useScaleFactors; out8[7] = t->everUnderflowed;
    return 0;
}
// development: the per-phase host times (TreeLikelihood::phaseUs), then reset
int btlTimings(void* h, double* out8) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    for (int k = 0; k < 8; k++) { out8[k] = t->phaseUs[k]; t->phaseUs[k] = 0.0; }
    return t->timing ? 1 : 0;
}
// ---- the multi-partition caller's per-evaluation call sequence ------------------------------------------------------------------
// What MultiPartitionDataLikelihoodDelegate.calculateLikelihood issues for ONE full evaluation, call for call and in its order
// (src/dr/evomodel/treedatalikelihood/MultiPartitionDataLikelihoodDelegate.java): models flagged as changed (:800-840), all branch
// matrices of all partitions in one updateTransitionMatricesWithMultipleModels (:880-887), the operation list (:972-997, 9-int tuples
// through updatePartialsByPartition), scale factors per partition (:1016-1017), weights and root frequencies of every partition
// (:1027-1035), the root integration by partition (:1074-1083).  The bookkeeping that decides WHAT is sent — buffer flips, the operation
// tables of both flips, the model flags — stays in beast_mcmc_amd/multipartition.py, which fills this record; the reference's caller is
// compiled code, and twenty calls through an interpreter's FFI cost a small partitioned evaluation a fifth of its time.
struct MpEvaluation {
    int K, always_rescale, n_matrices, n_ops9, n_ops7, n_scale, cum, n_branches;
    const int* update_substitution_model; const int* update_site_rate_model;         // [K] flags
    const double* const* U; const double* const* Uinv; const double* const* lambda;  // [K]
    const double* const* rates; const double* const* weights; const double* const* freqs;
    const int* eig_idx; const int* mat_idx; const double* lens0; const int* branch; double* lens;   // lens[k * n_branches + j] = lens0[j] * branch_rates[branch[j]]
    const int* ops9; const int* ops7; const int* scale_idx;
    const int* roots; const int* range; const int* cum_idx;
    double* by_part; double* total;
};
// returns 0, or the failing call's return code with *failed = its position in the sequence (1 setEigenDecomposition, 2 setCategoryRates-
// WithIndex, 3 updateTransitionMatrices..., 4 updatePartials..., 5 reset / 6 accumulateScaleFactors..., 7 setCategoryWeights,
// 8 setStateFrequencies, 9 the root call); the root call's -8 (a NaN) is returned with *failed = 0: the caller's to judge
int mpEvaluate(const BeagleApi* api, const BeaglePartitionApi* papi, int inst, const MpEvaluation* e, const double* branchRates, int* failed) {
    int rc = 0;
    *failed = 0;
    const int K = e->K;
    for (int k = 0; k < K; k++) {
        if (e->update_substitution_model[k] && (rc = api->setEigenDecomposition(inst, k, e->U[k], e->Uinv[k], e->lambda[k]))) { *failed = 1; return rc; }
        if (e->update_site_rate_model[k] && (rc = papi->setCategoryRatesWithIndex(inst, k, e->rates[k]))) { *failed = 2; return rc; }
    }
    const int nb = e->n_branches;
    for (int j = 0; j < nb; j++) e->lens[j] = e->lens0[j] * branchRates[e->branch[j]];
    for (int k = 1; k < K; k++) std::memcpy(e->lens + (size_t)k * nb, e->lens, (size_t)nb * sizeof(double));
    if ((rc = papi->updateTransitionMatricesWithMultipleModels(inst, e->eig_idx, e->eig_idx, e->mat_idx, nullptr, nullptr, e->lens, e->n_matrices))) { *failed = 3; return rc; }
    if (K > 1) rc = papi->updatePartialsByPartition(inst, e->ops9, e->n_ops9);
    else rc = api->updatePartials(inst, e->ops7, e->n_ops7, BEAGLE_OP_NONE);
    if (rc) { *failed = 4; return rc; }
    if (e->always_rescale)
        for (int k = 0; k < K; k++) {
            if (K > 1) {
                if ((rc = papi->resetScaleFactorsByPartition(inst, e->cum, k))) { *failed = 5; return rc; }
                if ((rc = papi->accumulateScaleFactorsByPartition(inst, e->scale_idx, e->n_scale, e->cum, k))) { *failed = 6; return rc; }
            } else {
                if ((rc = api->resetScaleFactors(inst, e->cum))) { *failed = 5; return rc; }
                if ((rc = api->accumulateScaleFactors(inst, e->scale_idx, e->n_scale, e->cum))) { *failed = 6; return rc; }
            }
        }
    for (int k = 0; k < K; k++) {
        if ((rc = api->setCategoryWeights(inst, k, e->weights[k]))) { *failed = 7; return rc; }
        if ((rc = api->setStateFrequencies(inst, k, e->freqs[k]))) { *failed = 8; return rc; }
    }
    if (K > 1) rc = papi->calculateRootLogLikelihoodsByPartition(inst, e->roots, e->range, e->range, e->cum_idx, e->range, K, 1, e->by_part, e->total);
    else { rc = api->calculateRootLogLikelihoods(inst, e->roots, e->range, e->range, e->cum_idx, 1, e->total); e->by_part[0] = e->total[0]; }
    if (rc && rc != -8) *failed = 9;
    return rc;
}

// last operation list (7 ints per op), for tests that assert the call protocol
int btlLastOperations(void* h, int* out, int maxOps) {
    TreeLikelihood* t = (TreeLikelihood*)h;
    int n = t->operationCount < maxOps ? t->operationCount : maxOps;
    std::memcpy(out, t->operations.data(), (size_t)n * BEAGLE_OP_COUNT * sizeof(int));
    return t->operationCount;
}

}  // extern "C"
