#!/bin/bash
# A/B builds of the LAB engine that differ only in the generated assembly loop:  tools/build_variant.sh <name> [VAR=value ...]
# -> build/variants/<name>/libhmsbeagle-jni.so (travels to the GPU box; select with BEAGLE_MI355_ENGINE_LIB=<path>).
# Every variant is a LAB build (-DBEAGLE_MI355_LAB: csrc/kernels.h labEnv — BEAGLE_MI355_ABLATE, _WALK_LDS_PAD, _CHUNK, _SCHED ... are
# connected to the environment only there; the product library of beast-mcmc_amd/build.py ignores them).
# The tracked walk4_fast_loop.inc is left untouched.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants/$NAME; mkdir -p "$OUT/inc"
( cd "$ROOT" && python beast-mcmc_amd/build.py --lab > /dev/null )
cp "$ROOT"/beast-mcmc_amd/csrc/*.h "$ROOT"/beast-mcmc_amd/csrc/kernels_walk4.hip "$OUT/inc/"
( cd "$ROOT" && env "$@" WALK4_OUT="$OUT/inc/walk4_fast_loop.inc" python tools/gen_walk4_fast.py )
sed -i 's#"../../include/beagle_mi355.h"#"'"$ROOT"'/include/beagle_mi355.h"#' "$OUT"/inc/*.h 2>/dev/null || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBEAGLE_MI355_BUILD -DBEAGLE_MI355_LAB -w -c -x hip "$OUT/inc/kernels_walk4.hip" -o "$OUT/kernels_walk4.o"
OBJ=$ROOT/beast-mcmc_amd/lib/lab/obj
hipcc --offload-arch=gfx950 -fPIC -shared $(ls $OBJ/*.o | grep -v kernels_walk4.hip.o) "$OUT/kernels_walk4.o" -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o "$OUT/libhmsbeagle-jni.so"
rm -rf "$OUT/inc" "$OUT/kernels_walk4.o"
echo "built $OUT/libhmsbeagle-jni.so ($*)"
