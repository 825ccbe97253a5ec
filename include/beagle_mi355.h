/*
 * beagle_mi355.h — C ABI of the MI355X-native tree-likelihood engine.
 *
 * This is the drop-in boundary.  Every entry point below is the plain-C shape of one
 * `native` method of the reference's JNI binding class
 *     /root/reference/lib/beagle.jar!beagle/BeagleJNIWrapper.class
 * (descriptors listed next to each function; the leading `I` of every descriptor is the
 * instance handle).  The Java-side caller of each method is cited as file:line relative
 * to /root/reference/.  Names follow BEAGLE's public C API (`beagle*`), which is what the
 * reference's `libhmsbeagle-jni.so` forwards to; the JNI symbols
 * `Java_beagle_BeagleJNIWrapper_<name>` live in csrc/jni_shim.cpp and call these functions
 * one-to-one.
 *
 * Conventions (lib/beagle.jar!beagle/BeagleErrorCode, BeagleJNIImpl):
 *   - every function returns an int error code: 0 = success, <0 = BEAGLE_ERROR_*;
 *     beagleCreateInstance returns the instance handle (>=0) or an error (<0);
 *   - arrays are borrowed for the duration of the call and may be LONGER than `count`
 *     (BeagleDataLikelihoodDelegate.java:179-183): exactly `count` (or 7*count / 9*count)
 *     entries are read;
 *   - index arrays documented "may be NULL" are accepted as NULL
 *     (HomogenousSubstitutionModelDelegate.java:260-261 passes null derivative indices);
 *   - BEAGLE_OP_NONE (-1) marks an unused scale/cumulative index (beagle.Beagle.NONE).
 *
 * Layouts at the boundary (lib/beagle.jar!beagle/GeneralBeagleImpl; BeagleTreeLikelihood.java:625-658):
 *   partials   double[C][P][S]      index c*P*S + p*S + i
 *   matrices   double[C][S][S]      row = parent state i, column = child state j
 *   eigen      U[i*S+k], Uinv[k*S+j] row-major S x S, lambda[S]
 *   tip states int[P], value >= S means missing/ambiguous (all-ones partial)
 * The layout in HBM is the engine's own.
 */
#ifndef BEAGLE_MI355_H
#define BEAGLE_MI355_H

#ifdef __cplusplus
extern "C" {
#endif
/* the engine is built with -fvisibility=hidden: what this header declares is what libhmsbeagle-jni.so exports (with the JNI natives) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- error codes: lib/beagle.jar!beagle/BeagleErrorCode#<clinit> ------------------- */
#define BEAGLE_SUCCESS                      0
#define BEAGLE_ERROR_GENERAL               -1
#define BEAGLE_ERROR_OUT_OF_MEMORY         -2
#define BEAGLE_ERROR_UNIDENTIFIED_EXCEPTION -3
#define BEAGLE_ERROR_UNINITIALIZED_INSTANCE -4
#define BEAGLE_ERROR_OUT_OF_RANGE          -5
#define BEAGLE_ERROR_NO_RESOURCE           -6
#define BEAGLE_ERROR_NO_IMPLEMENTATION     -7
#define BEAGLE_ERROR_FLOATING_POINT        -8

#define BEAGLE_OP_NONE                     -1
#define BEAGLE_OP_COUNT                     7   /* beagle.Beagle.OPERATION_TUPLE_SIZE */
#define BEAGLE_PARTITION_OP_COUNT           9   /* MultiPartitionDataLikelihoodDelegate.java:972-997 */

/* ---- flag bits: lib/beagle.jar!beagle/BeagleFlag#<clinit> --------------------------- */
#define BEAGLE_FLAG_PRECISION_SINGLE    (1L << 0)
#define BEAGLE_FLAG_PRECISION_DOUBLE    (1L << 1)
#define BEAGLE_FLAG_COMPUTATION_SYNCH   (1L << 2)
#define BEAGLE_FLAG_COMPUTATION_ASYNCH  (1L << 3)
#define BEAGLE_FLAG_EIGEN_REAL          (1L << 4)
#define BEAGLE_FLAG_EIGEN_COMPLEX       (1L << 5)
#define BEAGLE_FLAG_SCALING_MANUAL      (1L << 6)
#define BEAGLE_FLAG_SCALING_AUTO        (1L << 7)
#define BEAGLE_FLAG_SCALING_ALWAYS      (1L << 8)
#define BEAGLE_FLAG_SCALERS_RAW         (1L << 9)
#define BEAGLE_FLAG_SCALERS_LOG         (1L << 10)
#define BEAGLE_FLAG_VECTOR_SSE          (1L << 11)
#define BEAGLE_FLAG_VECTOR_NONE         (1L << 12)
#define BEAGLE_FLAG_THREADING_OPENMP    (1L << 13)
#define BEAGLE_FLAG_THREADING_NONE      (1L << 14)
#define BEAGLE_FLAG_PROCESSOR_CPU       (1L << 15)
#define BEAGLE_FLAG_PROCESSOR_GPU       (1L << 16)
#define BEAGLE_FLAG_SCALING_DYNAMIC     (1L << 19)
#define BEAGLE_FLAG_FRAMEWORK_CUDA      (1L << 22)
#define BEAGLE_FLAG_FRAMEWORK_OPENCL    (1L << 23)
#define BEAGLE_FLAG_FRAMEWORK_CPU       (1L << 27)
#define BEAGLE_FLAG_PARALLELOPS_STREAMS (1L << 28)
#define BEAGLE_FLAG_PARALLELOPS_GRID    (1L << 29)
#define BEAGLE_FLAG_THREADING_CPP       (1L << 30)

/* Filled by beagleCreateInstance; mirrors beagle.InstanceDetails (setResourceNumber,
 * setFlags, setResourceName, setImplementationName — BeagleDataLikelihoodDelegate.java:454-480). */
typedef struct {
    int   resourceNumber;
    char* resourceName;
    char* implName;
    char* implDescription;
    long  flags;
} BeagleInstanceDetails;

/* One entry of beagleGetResourceList; mirrors beagle.ResourceDetails. */
typedef struct {
    char* name;
    char* description;
    long  supportFlags;
    long  requiredFlags;
} BeagleResource;

typedef struct {
    BeagleResource* list;
    int length;
} BeagleResourceList;

/* getVersion ()Ljava/lang/String;  — must match (\d+)\.(\d+)\.(\d+).* (beagle.jar!BeagleInfo#getVersionNumbers;
 * gates in treedatalikelihood/BeagleFunctionality.java:40-70). */
const char* beagleGetVersion(void);
/* getCitation ()Ljava/lang/String; */
const char* beagleGetCitation(void);
/* getResourceList ()[Lbeagle/ResourceDetails;  — resource 0 is the host by BEAST convention
 * (BeagleTreeLikelihood.java:90-92); resources 1..G are the visible MI355X devices. */
BeagleResourceList* beagleGetResourceList(void);

/* One entry of beagleGetBenchmarkedResourceList; mirrors beagle.BenchmarkedResourceDetails (lib/beagle.jar: ctor (I),
 * setResourceNumber, setName, setDescription, setSupportFlags, setRequiredFlags, setReturnCode, setImplName,
 * setBenchedFlags, setBenchmarkResult, setPerformanceRatio). */
typedef struct {
    int    number;             /* resource number (what the caller passes back to createInstance) */
    char*  name;
    char*  description;
    long   supportFlags;
    long   requiredFlags;
    int    returnCode;         /* of the benchmark's createInstance / evaluation on that resource */
    char*  implName;
    long   benchedFlags;
    double benchmarkResult;    /* milliseconds per full-tree evaluation of the benchmark workload */
    double performanceRatio;   /* relative to the fastest resource (1.0 = fastest) */
} BeagleBenchmarkedResource;

typedef struct {
    BeagleBenchmarkedResource* list;
    int length;
} BeagleBenchmarkedResourceList;

#define BEAGLE_BENCHFLAG_SCALING_NONE    (1L << 0)
#define BEAGLE_BENCHFLAG_SCALING_ALWAYS  (1L << 1)
#define BEAGLE_BENCHFLAG_SCALING_DYNAMIC (1L << 2)

/* getBenchmarkedResourceList (IIIII[IIJJIIIJ)[Lbeagle/BenchmarkedResourceDetails;  — BEAST's -beagle_auto
 * (BeagleTreeLikelihood.java:392-414, BeagleDataLikelihoodDelegate.java:413-433): times a full-tree evaluation of a
 * synthetic alignment of the caller's shape (tips, states, patterns, categories) on every candidate resource
 * (resourceList, or all GPU resources incl. the pattern-sharded one when it is NULL) and returns them fastest first.
 * The list is owned by the library and valid until the next call. */
BeagleBenchmarkedResourceList* beagleGetBenchmarkedResourceList(int tipCount, int compactBufferCount, int stateCount, int patternCount,
                                      int categoryCount, const int* resourceList, int resourceCount, long preferenceFlags,
                                      long requirementFlags, int eigenModelCount, int partitionCount, int calculateDerivatives,
                                      long benchmarkFlags);

/* createInstance (IIIIIIIII[IIJJLbeagle/InstanceDetails;)I
 * callers: BeagleTreeLikelihood.java:420-433, BeagleDataLikelihoodDelegate.java:439-452
 * 2..64 states; more returns BEAGLE_ERROR_NO_IMPLEMENTATION (no kernel of this engine is built for it). */
int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount,
                         int stateCount, int patternCount, int eigenBufferCount,
                         int matrixBufferCount, int categoryCount, int scaleBufferCount,
                         const int* resourceList, int resourceCount,
                         long preferenceFlags, long requirementFlags,
                         BeagleInstanceDetails* returnInfo);
/* finalize (I)I — BeagleDataLikelihoodDelegate.java:1234-1238 */
int beagleFinalizeInstance(int instance);
/* setCPUThreadCount (II)I — BeagleTreeLikelihood.java:461-467 (must return 0 on a GPU instance) */
int beagleSetCPUThreadCount(int instance, int threadCount);

/* setPatternWeights (I[D)I — BeagleTreeLikelihood.java:533 */
int beagleSetPatternWeights(int instance, const double* inPatternWeights);
/* setPatternPartitions (II[I)I — MultiPartitionDataLikelihoodDelegate.java:553 */
int beagleSetPatternPartitions(int instance, int partitionCount, const int* inPatternPartitions);
/* setTipStates (II[I)I — BeagleTreeLikelihood.java:696-710 */
int beagleSetTipStates(int instance, int tipIndex, const int* inStates);
/* getTipStates (II[I)I */
int beagleGetTipStates(int instance, int tipIndex, int* outStates);
/* setTipPartials (II[D)I — double[P*S], replicated over categories by the library
 * (beagle.jar!GeneralBeagleImpl#setTipPartials) */
int beagleSetTipPartials(int instance, int tipIndex, const double* inPartials);
/* setPartials (II[D)I — double[C*P*S]; BeagleTreeLikelihood.java:621-661 */
int beagleSetPartials(int instance, int bufferIndex, const double* inPartials);
/* getPartials (III[D)I — BeagleTreeLikelihood.java:1132-1136; scaleIndex != NONE un-scales */
int beagleGetPartials(int instance, int bufferIndex, int scaleIndex, double* outPartials);
/* getLogScaleFactors (II[D)I */
int beagleGetLogScaleFactors(int instance, int scaleIndex, double* outScaleFactors);

/* setEigenDecomposition (II[D[D[D)I — HomogenousSubstitutionModelDelegate.java:228-240 */
int beagleSetEigenDecomposition(int instance, int eigenIndex, const double* inEigenVectors,
                                const double* inInverseEigenVectors, const double* inEigenValues);
/* setStateFrequencies (II[D)I — BeagleTreeLikelihood.java:1030 */
int beagleSetStateFrequencies(int instance, int stateFrequenciesIndex, const double* inStateFrequencies);
/* setCategoryWeights (II[D)I — BeagleTreeLikelihood.java:1029 */
int beagleSetCategoryWeights(int instance, int categoryWeightsIndex, const double* inCategoryWeights);
/* setCategoryRates (I[D)I — BeagleTreeLikelihood.java:970 */
int beagleSetCategoryRates(int instance, const double* inCategoryRates);
/* setCategoryRatesWithIndex (II[D)I — MultiPartitionDataLikelihoodDelegate.java:835 */
int beagleSetCategoryRatesWithIndex(int instance, int categoryRatesIndex, const double* inCategoryRates);
/* setTransitionMatrix (II[DD)I — double[C*S*S] */
int beagleSetTransitionMatrix(int instance, int matrixIndex, const double* inMatrix, double paddedValue);
/* getTransitionMatrix (II[D)I — AncestralStateBeagleTreeLikelihood.java:331 */
int beagleGetTransitionMatrix(int instance, int matrixIndex, double* outMatrix);
/* convolveTransitionMatrices (I[I[I[II)I — treelikelihood/SubstitutionModelDelegate.java:382-405 */
int beagleConvolveTransitionMatrices(int instance, const int* firstIndices, const int* secondIndices,
                                     const int* resultIndices, int matrixCount);
/* updateTransitionMatrices (II[I[I[I[DI)I — HomogenousSubstitutionModelDelegate.java:247-266;
 * derivative index arrays may be NULL */
int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices,
                                   const int* firstDerivativeIndices, const int* secondDerivativeIndices,
                                   const double* edgeLengths, int count);
/* updateTransitionMatricesWithMultipleModels (I[I[I[I[I[I[DI)I — MultiPartitionDataLikelihoodDelegate.java:880-887 */
int beagleUpdateTransitionMatricesWithMultipleModels(int instance, const int* eigenIndices,
                                   const int* categoryRateIndices, const int* probabilityIndices,
                                   const int* firstDerivativeIndices, const int* secondDerivativeIndices,
                                   const double* edgeLengths, int count);

/* updatePartials (I[III)I — BeagleTreeLikelihood.java:1003, BeagleDataLikelihoodDelegate.java:904.
 * operations = int[7*count]: {dest, writeScale, readScale, child1, matrix1, child2, matrix2}
 * (tuple built at BeagleTreeLikelihood.java:1266-1299).  The list must be dependency ordered;
 * it need not be level ordered (the engine levelises it). */
int beagleUpdatePartials(int instance, const int* operations, int operationCount, int cumulativeScaleIndex);
/* updatePartialsByPartition (I[II)I — int[9*count]:
 * {dest, writeScale, readScale, child1, matrix1, child2, matrix2, partition, cumulativeScale} */
int beagleUpdatePartialsByPartition(int instance, const int* operations, int operationCount);
/* waitForPartials (I[II)I */
int beagleWaitForPartials(int instance, const int* destinationPartials, int destinationPartialsCount);

/* accumulateScaleFactors (I[III)I — BeagleTreeLikelihood.java:1016-1023 */
int beagleAccumulateScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex);
/* accumulateScaleFactorsByPartition (I[IIII)I */
int beagleAccumulateScaleFactorsByPartition(int instance, const int* scaleIndices, int count,
                                            int cumulativeScaleIndex, int partitionIndex);
/* removeScaleFactors (I[III)I */
int beagleRemoveScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex);
/* removeScaleFactorsByPartition (I[IIII)I */
int beagleRemoveScaleFactorsByPartition(int instance, const int* scaleIndices, int count,
                                        int cumulativeScaleIndex, int partitionIndex);
/* resetScaleFactors (II)I */
int beagleResetScaleFactors(int instance, int cumulativeScaleIndex);
/* resetScaleFactorsByPartition (III)I */
int beagleResetScaleFactorsByPartition(int instance, int cumulativeScaleIndex, int partitionIndex);
/* copyScaleFactors (III)I */
int beagleCopyScaleFactors(int instance, int destScalingIndex, int srcScalingIndex);

/* calculateRootLogLikelihoods (I[I[I[I[II[D)I — BeagleTreeLikelihood.java:1038-1039,
 * BeagleDataLikelihoodDelegate.java:934-935.  Returns BEAGLE_ERROR_FLOATING_POINT (-8) when
 * the sum is NaN; outSumLogLikelihood is still written (BeagleJNIImpl tolerates -8). */
int beagleCalculateRootLogLikelihoods(int instance, const int* bufferIndices,
                                      const int* categoryWeightsIndices, const int* stateFrequenciesIndices,
                                      const int* cumulativeScaleIndices, int count,
                                      double* outSumLogLikelihood);
/* calculateRootLogLikelihoodsByPartition (I[I[I[I[I[III[D[D)I — MultiPartitionDataLikelihoodDelegate.java:1074-1083 */
int beagleCalculateRootLogLikelihoodsByPartition(int instance, const int* bufferIndices,
                                      const int* categoryWeightsIndices, const int* stateFrequenciesIndices,
                                      const int* cumulativeScaleIndices, const int* partitionIndices,
                                      int partitionCount, int count,
                                      double* outSumLogLikelihoodByPartition, double* outSumLogLikelihood);
/* getSiteLogLikelihoods (I[D)I — BeagleTreeLikelihood.java:1050 */
int beagleGetSiteLogLikelihoods(int instance, double* outLogLikelihoods);

/* ---- pre-order partials and branch gradients ---------------------------
 * Semantics from the callers (the implementing library is not in the reference tree):
 * src/dr/evomodel/treedatalikelihood/preorder/AbstractBeagleGradientDelegate.java:115-151, 207-221 and
 * AbstractBeagleBranchGradientDelegate.java:52-95 (+ the arithmetic spelled out at :103-140). */

/* setRootPrePartials (I[I[II)I — pre-order partial of a root = its state frequencies, replicated over
 * patterns and categories (what AbstractBeagleGradientDelegate.java:142-151 does through setPartials). */
int beagleSetRootPrePartials(int instance, const int* bufferIndices, const int* stateFrequenciesIndices, int count);
/* setDifferentialMatrix (II[D)I — HomogenousSubstitutionModelDelegate.java:179-193: stateCount^2 * categoryCount
 * doubles into a matrix buffer (the infinitesimal matrix scaled per category rate,
 * discrete/DiscreteTraitBranchRateDelegate.java:49-89). */
int beagleSetDifferentialMatrix(int instance, int matrixIndex, const double* inMatrix);
/* transposeTransitionMatrices (I[I[II)I — AbstractBeagleGradientDelegate.java:93-105: result[n] = input[n]^T per category */
int beagleTransposeTransitionMatrices(int instance, const int* inputIndices, const int* resultIndices, int matrixCount);
/* updatePrePartials (I[III)I — AbstractBeagleGradientDelegate.java:120.  7-int tuples
 * {pre(child) dest, writeScale, readScale, pre(parent), matrix(child), post(sibling), matrix(sibling)} (:211-217), matrices
 * untransposed:  dest[j] = sum_i P_child[i][j] * ( pre(parent)[i] * sum_k P_sib[i][k] post(sib)[k] ).
 * The list is in pre-order (a parent's op before its children's); the engine levelises it. */
int beagleUpdatePrePartials(int instance, const int* operations, int operationCount, int cumulativeScaleIndex);
/* calculateEdgeDifferentials (I[I[I[I[II[D[D[D)I — AbstractBeagleBranchGradientDelegate.java:82-92.  Per edge e and
 * pattern p: num = sum_c w_c sum_j pre[c,p,j] sum_k D[c][j][k] post[c,p,k], den = sum_c w_c sum_j pre[c,p,j] post[c,p,j];
 * outSumDerivatives[e] = sum_p weight_p num/den, outSumSquaredDerivatives[e] = sum_p weight_p (num/den)^2,
 * outDerivatives[e*P + p] = num/den.  Any of the three outputs may be NULL (BEAST passes null for outDerivatives). */
int beagleCalculateEdgeDifferentials(int instance, const int* postBufferIndices, const int* preBufferIndices,
                                     const int* derivativeMatrixIndices, const int* categoryWeightsIndices, int count,
                                     double* outDerivatives, double* outSumDerivatives, double* outSumSquaredDerivatives);

/* calculateCrossProductDifferentials (I[I[I[I[I[DI[D[D)I — discrete/SubstitutionModelCrossProductDelegate.java:153-178.
 * The arithmetic lives only in the absent beagle-lib; the semantics follow the call site and its consumer, which names what it
 * expects: "a first-order" approximation of d lnL / d Q_ij (AbstractLogAdditiveSubstitutionModelGradient.java:74-93; its
 * AFFINE_CORRECTED mode adds a term computed in Java from the same numbers).  First-order form, pinned by finite differences of
 * the golden-pinned lnL along random generator perturbations (tests/test_oracle_golden.py: the error falls in proportion to
 * the branch lengths) and exactly in the scaling direction:
 *   outSumDerivatives[i*S+j] += sum_e edgeLengths[e] sum_p weight_p (sum_c w_c r_c pre_e[c,p,i] post_e[c,p,j]) / (sum_c w_c pre_e . post_e)
 * outSumSquaredDerivatives must be NULL (BEAST passes null); otherwise BEAGLE_ERROR_NO_IMPLEMENTATION. */
int beagleCalculateCrossProductDifferentials(int instance, const int* postBufferIndices, const int* preBufferIndices,
                                             const int* categoryRateIndices, const int* categoryWeightsIndices,
                                             const double* edgeLengths, int count,
                                             double* outSumDerivatives, double* outSumSquaredDerivatives);

/* ---- entry points that exist in the binding but are not built yet: exported so the JNI shim links, and
 * return BEAGLE_ERROR_NO_IMPLEMENTATION (-7). ------------------------------------------------------ */
int beagleAddTransitionMatrices(int instance, const int* firstIndices, const int* secondIndices,
                                const int* resultIndices, int matrixCount);
int beagleUpdatePrePartialsByPartition(int instance, const int* operations, int operationCount);

/* ---- MI355X extensions (not part of the reference binding) -------------------------- */

/* Make every subsequent call of this instance enqueue on `hipStream` (a hipStream_t) instead
 * of the instance's own stream — lets a host that owns streams (torch, a JVM-side pool)
 * order the engine against its collectives. */
int beagleMi355SetStream(int instance, void* hipStream);
/* As beagleCalculateRootLogLikelihoods with count == 1, but the weighted sum stays on the
 * device: it is written to `deviceOut` (a device pointer to one double) on the instance's
 * stream and nothing is synchronised.  The pattern-sharded multi-GPU path all-reduces that
 * double over RCCL (DESIGN.md, row e). */
int beagleMi355CalculateRootLogLikelihoodsDevice(int instance, int bufferIndex, int categoryWeightsIndex,
                                      int stateFrequenciesIndex, int cumulativeScaleIndex, void* deviceOut);
/* One process per GPU, unique site patterns sharded over the processes (the job BEAST runs as -beagle_instances G,
 * TreeDataLikelihoodParser.java:205-278, with its Java-side sum): the sum over the shards as ONE all-reduce inside the engine.
 * beagleMi355GetCommUniqueId fills 128 bytes on one rank; the host hands them to every rank (any channel) and each calls
 * beagleMi355CommInit(instance, id, rank, rankCount) — RCCL over xGMI, one communicator per instance.  Then
 * beagleMi355CalculateRootLogLikelihoodsAllReduce is beagleCalculateRootLogLikelihoods with count == 1 whose result is the
 * sum over ALL ranks: reduction kernel, ncclAllReduce of one double and the hand-over to the host are enqueued back to back on
 * the instance's stream; every rank gets the same value (BEAGLE_ERROR_FLOATING_POINT when it is NaN, on every rank alike). */
int beagleMi355GetCommUniqueId(void* out128);
int beagleMi355CommInit(int instance, const void* uniqueId128, int rank, int rankCount);
int beagleMi355CalculateRootLogLikelihoodsAllReduce(int instance, int bufferIndex, int categoryWeightsIndex,
                                      int stateFrequenciesIndex, int cumulativeScaleIndex, double* outGlobalSum);
/* How many ranks the instance's communicator has — RCCL's own count (ncclCommCount), 0 without a communicator; on the
 * pattern-sharded handle (resource G+1) the ranks of its in-library communicator (0: host-side sum, no RCCL).  What a multi-GPU
 * benchmark line quotes as proof that the collective really spanned N GPUs. */
int beagleMi355CommInfo(int instance, int* outRanks);
/* getPartials for `count` buffers in one call: out = [count][C][P][S] (API layout), scale factors folded in where
 * scaleIndices[k] != BEAGLE_OP_NONE (scaleIndices may be NULL).  One batched materialisation of virtual buffers, device-side
 * layout conversion, pinned copies, one synchronisation per 256 MiB — for hosts that read many nodes per sample
 * (AncestralStateBeagleTreeLikelihood.java:414-542); the per-buffer beagleGetPartials takes the same path with count 1. */
int beagleMi355GetPartialsBatch(int instance, const int* bufferIndices, const int* scaleIndices, int count, double* outPartials);
/* For the JNI shim: getPartials / getSiteLogLikelihoods whose result STAYS in the engine's pinned host buffer — *outPinned,
 * *outCount doubles, valid until the next call on the instance — so that it reaches the Java array with one copy
 * (Set<Type>ArrayRegion) instead of two.  BEAGLE_ERROR_NO_IMPLEMENTATION on the sharded instance: use the ordinary call. */
int beagleMi355GetPartialsPinned(int instance, int bufferIndex, int scaleIndex, const double** outPinned, long* outCount);
int beagleMi355GetSiteLogLikelihoodsPinned(int instance, const double** outPinned, long* outCount);
/* Block until everything enqueued for the instance has completed. */
int beagleMi355Synchronize(int instance);
/* Engine-side timing of the hot kernel: HIP events recorded on the instance's stream around
 * the pruning launches of every updatePartials call while enabled; returns accumulated milliseconds and the
 * number of launches since the last reset. */
int beagleMi355KernelTimer(int instance, int enable, double* outMillis, long* outLaunches);
/* enable = N > 1 brackets every N-th updatePartials call only (an event pair costs the stream two barrier packets, ~12 us of
 * an evaluation): milliseconds and launches then cover those calls; beagleMi355KernelTimerCalls returns how many updatePartials
 * calls were bracketed since it was last asked (and resets the count) — the divisor for "kernel time per evaluation". */
int beagleMi355KernelTimerCalls(int instance, long* outCalls);
/* How many calculateRootLogLikelihoods calls of this instance (shard 0 of a sharded handle) were answered by the pattern walk itself —
 * a one-launch walk of an unpartitioned 4-state instance is held back until the next call, and when that call asks for the root
 * log-likelihood of the walk's last result the slice that computes it finishes the evaluation (no root kernel, no read-back of
 * the root's partials).  BEAGLE_MI355_NO_ROOT_FUSION=1 switches the holding off.  Since instance creation. */
int beagleMi355RootFusedCount(int instance, long* outCount);
/* beagleGetSiteLogLikelihoods (and its pinned form below) calls since instance creation that found the site values ALREADY ON THE HOST: a
 * caller that reads them after every whole-alignment root sum — BeagleTreeLikelihood.java:1050 does — has them sent to a pinned buffer right
 * behind the root's kernel, before it asks (two such reads in a row switch it on, a sum nobody reads the values of switches it off;
 * the values are those of the stream-ordered download, bit for bit).  BEAGLE_MI355_NO_SITE_PREFETCH=1 at creation: never. */
int beagleMi355SitePrefetchCount(int instance, long* outCount);
/* Forget what the (enabled) kernel timer and the walk counters have gathered so far — no synchronisation, no allocation: for a
 * caller that has just synchronised the stream itself and wants the measurement to start here (bench.py: between its warm-up
 * and its timed steps, without giving the device an idle gap to drop its clocks in). */
int beagleMi355KernelTimerRestart(int instance);
/* Traffic counters of the 4-state pattern walk since the last beagleMi355KernelTimer call: out[0] micro-operations,
 * [1] partials buffers stored, [2] partials buffers read from memory, [3] tip-state vectors read, [4] scale-factor
 * vectors read, [5] walk launches, [6] scale-factor vectors written, [7] walk launches that ran the assembly loop.  bench.py turns them into the
 * bytes the design has to move (roofline.achieved). */
int beagleMi355WalkStats(int instance, long* out8);
/* The one-launch pattern walk (4 states) since instance creation.  out[0]: workgroups whose wait for the slices they read from ran
 * out and that computed those slices themselves (kernels_walk4.hip: the launch cannot deadlock whatever order the hardware
 * dispatches workgroups in; on gfx950 the count stays 0); out[1]: that wait's limit in microseconds (BEAGLE_MI355_WALK_SPIN_US at
 * creation, default 20 000); out[2]: folded reciprocal vectors in use by read-mode programs (one per stored node instead of one
 * per node: DESIGN.md 4.1); out[3]: how many times such vectors were (re)built from the per-node factors. */
int beagleMi355WalkHealth(int instance, long* out4);
/* How the one-launch walks of a 4-state instance were run since its creation: out[0] launches on TICKETS (the program's slices form
 * a forest: only the slices without dependencies get workgroups, the workgroup that arrives last at a slice above runs it — nobody
 * waits; the default), out[1] launches on dependency FLAGS (every slice its own workgroups, which poll: programs whose slices do not
 * form a forest, or BEAGLE_MI355_NO_WALK_TICKETS=1), out[2] / out[3]: slices with workgroups of their own / slices in all, last launch;
 * out[4]: micro-operations that were not part of a device program because their consumer evaluated them inside its own stage (nodes over
 * two compact tips: DESIGN.md 4.1 "fused cherries"; BEAGLE_MI355_NO_CHERRY_FUSION=1: none), out[5]: micro-operations planned, both since
 * the last beagleMi355KernelTimer call; out[6]: accumulateScaleFactors calls since creation that were answered from the per-slice products of
 * factors a write-mode walk had just left behind (BEAGLE_MI355_NO_SLICE_SUMS=1: none); out[7]: calculateRootLogLikelihoodsByPartition
 * calls since creation that the top slices of the partitions finished inside the walk's own launch (4 states, up to eight partitions;
 * BEAGLE_MI355_NO_ROOT_PARTS_FUSION=1: none).  out8 holds eight values. */
int beagleMi355WalkLaunchInfo(int instance, long* out8);
/* The gradient pass (4 states) since instance creation.  A pre-order list without scale indices is held back until a call needs
 * what it writes (or changes what it reads): out[0] lists that ran together with the edge derivatives that followed them (one
 * sweep per tree level; sums and sums of squares), out[1] lists that ran operation by operation, out[2] edge-derivative calls
 * answered from a held list WITHOUT writing a pre-order partial (sums only; the list stays held), out[3] held lists that had
 * to run after all because a later call touched their buffers. */
int beagleMi355GradientStats(int instance, long* out4);
/* The instance's dimensions, for wrappers that have to size what a whole-array output defines (the JNI shim):
 * out8 = {tipCount, partialsBufferCount, stateCount, patternCount, categoryCount, matrixBufferCount, scaleBufferCount, partitionCount}. */
int beagleMi355GetDimensions(int instance, int* out8);
/* Bytes of HBM currently allocated by the instance. */
long beagleMi355DeviceBytes(int instance);

/* Function table: lets a host driver (host/tree_likelihood.cpp) or a test drive any engine
 * that implements this ABI (the HIP engine, or the CPU oracle under oracle/) through one type. */
typedef struct BeagleApi {
    const char* (*getVersion)(void);
    int (*createInstance)(int, int, int, int, int, int, int, int, int, const int*, int, long, long, BeagleInstanceDetails*);
    int (*finalizeInstance)(int);
    int (*setPatternWeights)(int, const double*);
    int (*setTipStates)(int, int, const int*);
    int (*setTipPartials)(int, int, const double*);
    int (*setPartials)(int, int, const double*);
    int (*getPartials)(int, int, int, double*);
    int (*getLogScaleFactors)(int, int, double*);
    int (*setEigenDecomposition)(int, int, const double*, const double*, const double*);
    int (*setStateFrequencies)(int, int, const double*);
    int (*setCategoryWeights)(int, int, const double*);
    int (*setCategoryRates)(int, const double*);
    int (*setTransitionMatrix)(int, int, const double*, double);
    int (*getTransitionMatrix)(int, int, double*);
    int (*updateTransitionMatrices)(int, int, const int*, const int*, const int*, const double*, int);
    int (*updatePartials)(int, const int*, int, int);
    int (*accumulateScaleFactors)(int, const int*, int, int);
    int (*removeScaleFactors)(int, const int*, int, int);
    int (*resetScaleFactors)(int, int);
    int (*copyScaleFactors)(int, int, int);
    int (*calculateRootLogLikelihoods)(int, const int*, const int*, const int*, const int*, int, double*);
    int (*getSiteLogLikelihoods)(int, double*);
    /* optional (NULL when the engine has no device memory): beagleMi355CalculateRootLogLikelihoodsDevice */
    int (*calculateRootLogLikelihoodsDevice)(int, int, int, int, int, void*);
    /* optional (NULL without a communicator layer): beagleMi355CalculateRootLogLikelihoodsAllReduce */
    int (*calculateRootLogLikelihoodsAllReduce)(int, int, int, int, int, double*);
} BeagleApi;

const BeagleApi* beagleGetApiTable(void);

/* The calls of a PARTITIONED instance's evaluation (MultiPartitionDataLikelihoodDelegate.java:800-1083), for a native caller that is
 * handed a table instead of linking the library (tools/host: the multi-partition call sequence in C++, as the single-partition one). */
typedef struct BeaglePartitionApi {
    int (*setCategoryRatesWithIndex)(int, int, const double*);
    int (*updateTransitionMatricesWithMultipleModels)(int, const int*, const int*, const int*, const int*, const int*, const double*, int);
    int (*updatePartialsByPartition)(int, const int*, int);
    int (*resetScaleFactorsByPartition)(int, int, int);
    int (*accumulateScaleFactorsByPartition)(int, const int*, int, int, int);
    int (*calculateRootLogLikelihoodsByPartition)(int, const int*, const int*, const int*, const int*, const int*, int, int, double*, double*);
} BeaglePartitionApi;
const BeaglePartitionApi* beagleGetPartitionApiTable(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BEAGLE_MI355_H */
