"""The JNI shim (beast-mcmc_amd/csrc/jni_shim.cpp) executed end to end WITHOUT a JVM.

tests/native/fake_jvm.cpp supplies a JNIEnv (the 229-slot function table of the JNI specification over plain C++
objects; slots the shim must not touch abort) and calls Java_beagle_BeagleJNIWrapper_* the way beagle.BeagleJNIWrapper's
Java callers do: getVersion, getResourceList, the beagle.jar smoke test (lib/beagle.jar!beagle/BeagleFactory#main,
"PAUP logL = -1574.63623") through createInstance ... calculateRootLogLikelihoods — with BEAST's habits: arrays longer than
`count`, null derivative-index arrays — and getBenchmarkedResourceList (-beagle_auto, BeagleTreeLikelihood.java:392-414).
The Java objects the shim builds are checked against the method tables of the jar's classes."""
import os
import re
import subprocess

import pytest

import helpers
from beast_mcmc_amd.inputs import patterns
from test_oracle_golden import fmt5

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jni_symbols_drive_the_engine_through_a_fake_jnienv(tmp_path):
    g = helpers.golden("jar_smoke.json")
    exe = str(tmp_path / "fake_jvm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "native", "fake_jvm.cpp"), "-ldl", "-o", exe])
    rows = []
    for seq in g["sequences"]:
        st = patterns.nucleotide_states(seq).copy()
        st[st > 3] = 4
        rows.append(st)
    fixture = tmp_path / "smoke.txt"
    with open(fixture, "w") as fh:
        fh.write("%d\n" % len(rows[0]))
        for r in rows:
            fh.write(" ".join(str(int(x)) for x in r) + "\n")
        for key in ("evec", "ivec", "eval", "edge_lengths"):
            fh.write(" ".join(repr(float(x)) for x in g[key]) + "\n")
    out = subprocess.run([exe, os.path.join(ROOT, "beast-mcmc_amd", "lib", "libhmsbeagle-jni.so"), str(fixture)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    txt = out.stdout
    assert re.search(r"^version=4\.\d+\.\d+", txt, re.M)
    # resources: [0] the CPU placeholder, GPUs, the pattern-sharded "all GPUs" entry; ResourceDetails(int) + three setters each
    n_res = int(re.search(r"^resources=(\d+) elementClass=beagle/ResourceDetails", txt, re.M).group(1))
    assert n_res >= 3
    res = re.findall(r"^resource\[(\d+)\] class=beagle/ResourceDetails ctor=(\d+) (.*)$", txt, re.M)
    assert [int(a) for a, _, _ in res] == [int(b) for _, b, _ in res] == list(range(n_res))
    assert all("setName=<" in c and "setDescription=<" in c and "setFlags=<" in c for _, _, c in res)
    assert "sharded" in res[-1][2]
    # InstanceDetails filled through its four setters
    det = re.search(r"^details\[0\] class=beagle/InstanceDetails .*$", txt, re.M).group(0)
    assert "setResourceNumber=<1>" in det and "setImplementationName=<HIP-gfx950-fp64>" in det and "setResourceName=<" in det
    flags = int(re.search(r"setFlags=<(\d+)>", det).group(1))
    assert flags & (1 << 16) and not flags & (1 << 27)          # PROCESSOR_GPU, not FRAMEWORK_CPU: BEAST sends level-ordered ops
    m = re.search(r"^rc=(\d+) rootRc=(-?\d+) lnL=(-?[\d.]+)", txt, re.M)
    assert m.group(1) == "0" and m.group(2) == "0"
    assert fmt5(float(m.group(3))) == fmt5(g["lnL"])
    m2 = re.search(r"^siteRc=0 siteSum=(-?[\d.]+)", txt, re.M)
    assert abs(float(m2.group(1)) - float(m.group(3))) < 1e-6   # unit pattern weights: the site values add up to lnL
    m3 = re.search(r"^getTipStatesRc=0 matching=(\d+) of (\d+)", txt, re.M)
    assert m3.group(1) == m3.group(2)                           # an OUTPUT array is copied back (SetIntArrayRegion)
    assert "finalizeRc=0" in txt
    # output arrays (round-3 advisor finding): untouched when the call fails, refused (-5) when shorter than what the call defines,
    # tail preserved when longer
    me = re.search(r"^outputOnError rc=(-?\d+) untouched=(\d+) of (\d+) \| shortArray rc=(-?\d+) untouched=(\d+) of (\d+) \| longArray rc=(-?\d+) tail=(-?\d+),(-?\d+) first=(\d)", txt, re.M)
    assert me, txt[-2000:]
    assert int(me.group(1)) != 0 and me.group(2) == me.group(3)
    assert int(me.group(4)) == -5 and me.group(5) == me.group(6)
    assert int(me.group(7)) == 0 and me.group(8) == "-7" and me.group(9) == "-7" and me.group(10) == "1"
    # what the shim copies per call: count-derived lengths in (the arrays are longer: 9 edge lengths for 4 branches, 21
    # operation ints for 2 operations), outputs out only — never a whole output array in before it is overwritten
    n_sites = len(rows[0])
    traffic = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"^traffic (\w+) in=(\d+) out=(\d+)", txt, re.M)}
    assert traffic["updateTransitionMatrices"] == (4 * 4 + 4 * 8, 0)            # 4 matrix indices + 4 lengths; null derivative arrays
    assert traffic["updatePartials"] == (2 * 7 * 4, 0)
    assert traffic["calculateRootLogLikelihoods"] == (4 * 4, 8)                 # four 1-entry index arrays in, one double out
    assert traffic["getSiteLogLikelihoods"] == (0, 8 * n_sites)
    assert traffic["getPartials"] == (0, 8 * 4 * n_sites)
    assert traffic["getTipStates"] == (0, 4 * n_sites)
    mp = re.search(r"^getPartialsRc=0 lnLfromPartials=(-?[\d.]+)", txt, re.M)
    assert abs(float(mp.group(1)) - float(m.group(3))) < 1e-6                   # the root partials that came back are the right ones
    # -beagle_auto: every GPU resource benchmarked, fastest first, all ten setters of BenchmarkedResourceDetails used
    n_b = int(re.search(r"^benchmarked=(\d+) elementClass=beagle/BenchmarkedResourceDetails", txt, re.M).group(1))
    assert n_b == n_res - 1
    bench = re.findall(r"^benchmarked\[\d+\] class=beagle/BenchmarkedResourceDetails ctor=\d+ (.*)$", txt, re.M)
    times = []
    for c in bench:
        for setter in ("setResourceNumber", "setName", "setDescription", "setSupportFlags", "setRequiredFlags", "setReturnCode",
                       "setImplName", "setBenchedFlags", "setBenchmarkResult", "setPerformanceRatio"):
            assert setter + "=<" in c, (setter, c)
        assert "setReturnCode=<0>" in c
        times.append(float(re.search(r"setBenchmarkResult=<([^>]+)>", c).group(1)))
    assert all(t > 0 for t in times) and times == sorted(times)
    assert "exceptionsRaised=0 pending=0" in txt
