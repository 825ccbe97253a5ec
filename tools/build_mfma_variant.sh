#!/bin/bash
# A/B builds of the LAB engine that differ only in kernels_mfma.hip's experiment macros:  tools/build_mfma_variant.sh <name> -DMI355_EXP_...
# -> build/variants/<name>/libhmsbeagle-jni.so (travels to the GPU box; select with BEAGLE_MI355_ENGINE_LIB=<path>).
# (the MI355_EXP_* macros only exist under -DBEAGLE_MI355_LAB, which the product build never sets)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants/$NAME; mkdir -p "$OUT"
( cd "$ROOT" && python beast-mcmc_amd/build.py --lab > /dev/null )
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBEAGLE_MI355_BUILD -DBEAGLE_MI355_LAB -w "$@" -c -x hip "$ROOT/beast-mcmc_amd/csrc/kernels_mfma.hip" -o "$OUT/kernels_mfma.o"
OBJ=$ROOT/beast-mcmc_amd/lib/lab/obj
hipcc --offload-arch=gfx950 -fPIC -shared $(ls $OBJ/*.o | grep -v kernels_mfma.hip.o) "$OUT/kernels_mfma.o" -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o "$OUT/libhmsbeagle-jni.so"
rm -f "$OUT/kernels_mfma.o"
echo "built $OUT/libhmsbeagle-jni.so ($*)"
