#!/bin/bash
# JVM-side validation pack.  NOT runnable in the build container (no JDK, no beast.jar): run it on a
# machine with a JDK, a built BEAST X (ant dist -> build/dist/beast.jar or a release) and an MI355X.
#
# It loads THIS repo's libhmsbeagle-jni.so under the unmodified BEAST X and checks, against the stock BEAGLE CPU library
# when one is installed (BEAGLE_CPU_LIB_DIR), that
#   1. `beast -beagle_info` lists the MI355X resources (resource 0 = placeholder, 1..G = GPUs);
#   2. the reference's own likelihood regression XMLs report the same log-likelihoods:
#        examples/TestXML/testLikelihood.xml, tests/TestXML/testBranchSpecificSubstitutionModel.xml,
#        tests/TestXML/testEpochConvolutionOrder.xml  (their expected values are asserted inside the XML: a wrong
#        likelihood makes BEAST exit non-zero);
#   3. a short run of examples/Benchmarks/benchmark1.xml and benchmark2.xml gives the same posterior trace for the same
#        seed on both libraries (first 1000 states; fp64 on both sides, so lnL columns agree to ~1e-10 relative).
#
# usage: BEAST_HOME=/path/to/beast-mcmc  [BEAGLE_CPU_LIB_DIR=/usr/local/lib]  tools/jvm_validation.sh
set -euo pipefail
REPO="$(cd "$(dirname "$0")/.." && pwd)"
: "${BEAST_HOME:?set BEAST_HOME to a BEAST X checkout with build/dist/beast.jar (or lib/beast.jar)}"
LIBDIR="$REPO/beast-mcmc_amd/lib"
[ -f "$LIBDIR/libhmsbeagle-jni.so" ] || { echo "build the engine first: python -c 'import __graft_entry__ as g; g.build()'"; exit 2; }
JAR=""
for c in "$BEAST_HOME/build/dist/beast.jar" "$BEAST_HOME/lib/beast.jar"; do [ -f "$c" ] && JAR="$c" && break; done
[ -n "$JAR" ] || { echo "beast.jar not found under $BEAST_HOME"; exit 2; }
CP="$JAR:$BEAST_HOME/lib/beagle.jar"
OUT="${OUT:-/tmp/mi355_jvm_validation}"; mkdir -p "$OUT"

beast() {   # $1 = java.library.path, rest = BEAST arguments
    local libpath="$1"; shift
    java -Xmx8g -Djava.library.path="$libpath" -cp "$CP" dr.app.beast.BeastMain -overwrite -working "$@"
}

echo "== 1. resources =="
beast "$LIBDIR" -beagle_info | tee "$OUT/beagle_info.txt"
grep -q "MI355" "$OUT/beagle_info.txt" || { echo "FAIL: no MI355X resource listed"; exit 1; }

echo "== 2. regression XMLs with built-in expected likelihoods (resource 1 = first MI355X) =="
for xml in examples/TestXML/testLikelihood.xml tests/TestXML/testBranchSpecificSubstitutionModel.xml tests/TestXML/testEpochConvolutionOrder.xml; do
    [ -f "$BEAST_HOME/$xml" ] || { echo "skip $xml (not in this checkout)"; continue; }
    cp "$BEAST_HOME/$xml" "$OUT/"
    ( cd "$OUT" && beast "$LIBDIR" -beagle_order 1 -beagle_double -seed 666 "$(basename "$xml")" > "$(basename "$xml").mi355.log" 2>&1 ) \
        && echo "PASS $xml" || { echo "FAIL $xml (see $OUT/$(basename "$xml").mi355.log)"; exit 1; }
done

echo "== 3. benchmark traces: MI355X engine vs stock BEAGLE CPU =="
if [ -n "${BEAGLE_CPU_LIB_DIR:-}" ]; then
    for xml in examples/Benchmarks/benchmark1.xml examples/Benchmarks/benchmark2.xml; do
        [ -f "$BEAST_HOME/$xml" ] || continue
        b="$(basename "$xml" .xml)"
        for side in mi355 cpu; do
            mkdir -p "$OUT/$b.$side"; cp "$BEAST_HOME/$xml" "$OUT/$b.$side/"
            # shorten the chain: 1000 states are enough to compare likelihood columns
            sed -i -E 's/chainLength="[0-9]+"/chainLength="1000"/; s/logEvery="[0-9]+"/logEvery="100"/g' "$OUT/$b.$side/$b.xml"
            if [ "$side" = mi355 ]; then lp="$LIBDIR"; ord="-beagle_order 1"; else lp="$BEAGLE_CPU_LIB_DIR"; ord="-beagle_CPU"; fi
            ( cd "$OUT/$b.$side" && beast "$lp" $ord -beagle_double -seed 666 "$b.xml" > run.log 2>&1 )
        done
        python3 - "$OUT/$b.mi355" "$OUT/$b.cpu" <<'PY'
import glob, sys
def col(d):
    f = sorted(glob.glob(d + "/*.log"))
    f = [x for x in f if not x.endswith("run.log")][0]
    rows = [l.split("\t") for l in open(f) if l.strip() and not l.startswith("#")]
    i = rows[0].index("likelihood") if "likelihood" in rows[0] else 1
    return [float(r[i]) for r in rows[1:]]
a, b = col(sys.argv[1]), col(sys.argv[2])
worst = max(abs(x - y) / max(abs(y), 1e-300) for x, y in zip(a, b))
print("states compared:", len(a), "max relative lnL difference:", worst)
sys.exit(0 if len(a) == len(b) and worst <= 1e-8 else 1)
PY
        echo "PASS $b trace comparison"
    done
else
    echo "BEAGLE_CPU_LIB_DIR not set: skipping the side-by-side traces (steps 1-2 already exercise the JNI binding)"
fi
echo "all JVM-side checks passed; artefacts in $OUT"
