// root_site4.h — the root integration of a 4-state walk instance, shared by its two executors so that they produce the same bits:
// k_rootSite4W (kernels.hip: a launch of its own, the root's partials from memory) and the epilogue of k_walk4_fast's root slice
// (kernels_walk4.hip: the partials still in the waves' registers — no launch, no read-back; engine_walk.cpp PendingWalk).
//   site(p) = log( sum_c w_c sum_i pi_i L_root[c][p][i] ) (+ the cumulative scale factors)      TreeDataLikelihood / BeagleTreeLikelihood
//   -> Beagle.calculateRootLogLikelihoods (beagle.jar); the arithmetic of GeneralLikelihoodCore.java:358-406 with fixed fused
//   multiply-adds.  A wave owns 128 patterns with the assembly loop's lane map (lane 2q + r: patterns q + 32r and 64 + q + 32r),
//   adds its lanes' two weighted site values and folds them in a fixed shuffle tree; the last wave to arrive adds the groups' sums
//   in index order, 64 interleaved partial sums folded by the same tree.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace mi355 {

__device__ __forceinline__ double rootDot4(const double* __restrict__ f, double x, double y, double z, double w) {
    return __builtin_fma(f[3], w, __builtin_fma(f[2], z, __builtin_fma(f[1], y, f[0] * x)));
}
// the lane's weighted contribution: site values of its two patterns from their category-summed likelihoods
__device__ __forceinline__ double rootFinishPair(double sumA, double sumB, int pa, int pb, int pEnd, const double* __restrict__ cum, int cumIsRaw,
                                                 const double* __restrict__ patternWeights, double* __restrict__ siteLogL) {
    double ca = 0.0, cb = 0.0;
    if (pa < pEnd) {
        double site = log(sumA);
        if (cum) site += cumIsRaw ? log(cum[pa]) : cum[pa];
        siteLogL[pa] = site;
        ca = site * patternWeights[pa];
    }
    if (pb < pEnd) {
        double site = log(sumB);
        if (cum) site += cumIsRaw ? log(cum[pb]) : cum[pb];
        siteLogL[pb] = site;
        cb = site * patternWeights[pb];
    }
    return ca + cb;
}
__device__ __forceinline__ double rootWaveSum(double v) {              // lane 0 <- the wave's sum, fixed shape
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// One wave (all 64 lanes, `lane` = its lane index) has its group's sum in lane 0: publish it; the LAST group to arrive adds all of
// them up and writes the result (and the host's sequence word).  `counter` returns to 0 for the next evaluation.
__device__ __forceinline__ void rootPublishGroup(double groupSum, int lane, int group, int groups, double* __restrict__ blockSums, unsigned* counter,
                                                 double* __restrict__ out, unsigned long long* flag, unsigned long long seq) {
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(blockSums) + group, (unsigned long long)__double_as_longlong(groupSum),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                                       // my group's sum before my ticket
        last = atomicAdd(counter, 1u) == (unsigned)(groups - 1);
    }
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();
    double v = 0.0;
    for (int k = lane; k < groups; k += 64)
        v += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(blockSums) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    v = rootWaveSum(v);
    if (lane == 0) {
        out[0] = v;
        *counter = 0u;                                                         // ready for the next evaluation (same stream: ordered)
        if (flag) { __threadfence_system(); __atomic_store_n(flag, seq, __ATOMIC_RELEASE); }
    }
}

// ... the same for several partitions in one launch: group `slot` = blockOff(partition) + its index within the partition; the LAST of all
// `total` groups adds every partition's sums up in index order (64 interleaved partial sums folded by the same tree) and writes out[k].
template <class Parts, class GroupsOf, class OffOf>
__device__ __forceinline__ void rootPublishGroupParts(double groupSum, int lane, int slot, int total, int n, const Parts& parts, GroupsOf groupsOf, OffOf offOf,
                                                      double* __restrict__ blockSums, unsigned* counter, double* __restrict__ out,
                                                      unsigned long long* flag, unsigned long long seq) {
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(blockSums) + slot, (unsigned long long)__double_as_longlong(groupSum),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                                       // my group's sum before my ticket
        last = atomicAdd(counter, 1u) == (unsigned)(total - 1);
    }
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();
    for (int i = 0; i < n; i++) {
        const int groups = groupsOf(parts, i), off = offOf(parts, i);
        double v = 0.0;
        for (int k = lane; k < groups; k += 64)
            v += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(blockSums) + off + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        v = rootWaveSum(v);
        if (lane == 0) out[i] = v;
    }
    if (lane == 0) {
        *counter = 0u;                                                         // ready for the next evaluation (same stream: ordered)
        if (flag) { __threadfence_system(); __atomic_store_n(flag, seq, __ATOMIC_RELEASE); }
    }
}

}  // namespace mi355
