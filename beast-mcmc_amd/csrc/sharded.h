// sharded.h — the multi-GPU (pattern-sharded) instance behind one handle; see sharded.cpp.
#pragma once
#include <functional>

#include "../../include/beagle_mi355.h"

namespace mi355 {

constexpr int SHARD_HANDLE_BASE = 1 << 20;      // handles >= this are sharded instances
// While the library creates the shards of a sharded instance: the pattern count of the WHOLE alignment.  The size-dependent choices an
// instance makes for itself (how long a virtual definition may be: engine_abi.cpp) are then made as the single-GPU instance of the same
// alignment would make them — same programs per pattern, hence the same site values bit for bit (tests/test_gpu_sharded_instance.py).
extern thread_local int tlsWholePatternCount;
inline bool isShardedHandle(int h) { return h >= SHARD_HANDLE_BASE; }

int shardedDeviceCountOverride();               // BEAGLE_MI355_SHARDS (tests), 0 = one shard per GPU
int shardedCreate(int gpuCount, int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount, int patternCount,
                  int eigenBufferCount, int matrixBufferCount, int categoryCount, int scaleBufferCount, long preferenceFlags,
                  long requirementFlags, BeagleInstanceDetails* returnInfo);
int shardedFinalize(int handle);
int shardedShardCount(int handle);
int shardedCommRanks(int handle);           // ranks of the in-library communicator as RCCL counts them (0: host-side sum)
int shardedPatternCount(int handle);
int shardedStates(int handle);
int shardedCategories(int handle);
bool shardedEigenComplex(int handle);        // created with BEAGLE_FLAG_EIGEN_COMPLEX: eigenvalue arrays have 2 S entries
void shardedBounds(int handle, int shard, int* pStart, int* pEnd);
void shardedBoundsOfHandle(int handle, int shardHandle, int* pStart, int* pEnd);

// the same call on every shard (each on its own host thread); first error wins
int shardedBroadcast(int handle, const std::function<int(int shardHandle)>& call);
// the same, WITHOUT waiting (calls that return nothing to the caller): `call` must own everything it reads — the caller's
// arrays may be reused as soon as this returns; errors surface at the next waiting call
int shardedPost(int handle, std::function<int(int shardHandle)> call);
// arrays indexed by pattern: v is [planes][P][perPattern]; every shard sees its own block
int shardedSetPerPatternInts(int handle, const int* v, const std::function<int(int, const int*)>& call);
int shardedSetPerPatternDoubles(int handle, const double* v, int perPattern, int planes, const std::function<int(int, const double*)>& call);
int shardedGetPerPatternInts(int handle, int* out, const std::function<int(int, int*)>& call);
int shardedGetPerPatternDoubles(int handle, double* out, int perPattern, int planes, const std::function<int(int, double*)>& call);
// per-shard device-side root sums (count doubles each) -> ONE all-reduce (RCCL) -> outValues on the host
int shardedRootReduce(int handle, int count, const std::function<int(int shardHandle, double* deviceOut)>& enqueue, double* outValues);
// per-shard host vectors of `len` doubles that add up
int shardedSumDoubles(int handle, int len, const std::function<int(int shardHandle, double* out)>& call, double* outSum);


// engine_abi.cpp, for the reduction above: `count` doubles at device address dValues of (single-GPU) instance `instance` go to
// the host through the instance's mapped result words — one small kernel behind whatever is on the instance's stream, then a
// poll — instead of a device-to-host copy and a stream synchronisation.  count <= 480.
int publishAndWait(int instance, const double* dValues, int count, double* out);
// the first error a deferred operation of (single-GPU) instance `instance` left behind, cleared (0: none); no synchronisation
int takeAsyncError(int instance);

}  // namespace mi355
