"""In-tree build of the native pieces (explicit hipcc / g++ command lines, no build system):

  lib/libhmsbeagle-jni.so   HIP engine + C ABI + JNI shim, gfx950 only   (csrc/*.hip, csrc/*.cpp)
  lib/libbeast_host.so      the caller stand-in: BeagleTreeLikelihood's call protocol in C++  (tools/host/tree_likelihood.cpp;
                            harness for tests and bench.py — in production the caller is BEAST's Java)
  lib/lab/libhmsbeagle-jni.so   (``--lab`` only) the same engine compiled with -DBEAGLE_MI355_LAB: the tuning knobs and timing
                            experiments of csrc/kernels.h labEnv() — some of which give wrong results by construction — are
                            connected to the environment ONLY in this build; select it with BEAGLE_MI355_ENGINE_LIB=<path>

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the built
``.so`` files travel to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(ROOT, "tools", "host")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+ " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_engine(force=False, lab=False):
    """Each source is compiled to its own object (in parallel, only when it or a header changed) and the objects are
    linked into the one shared library; the 4-state kernel file alone takes minutes, the rest seconds."""
    from concurrent.futures import ThreadPoolExecutor
    lib_dir = os.path.join(LIB, "lab") if lab else LIB
    os.makedirs(lib_dir, exist_ok=True)
    obj_dir = os.path.join(lib_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(lib_dir, "libhmsbeagle-jni.so")
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
        [os.path.join(ROOT, "include", "beagle_mi355.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBEAGLE_MI355_BUILD", "-Wall",
             "-Wno-unused-result", "-Wno-unused-value", "-Wno-cuda-compat", "-fvisibility=hidden"] + (["-DBEAGLE_MI355_LAB"] if lab else [])
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc()] + flags + ["-c", "-x", "hip", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
            list(pool.map(_run, jobs))
    # what the library exports is its interface and nothing else: the BEAGLE C API (beagle*, the engine's own beagleMi355*
    # entry points included) and the 47 JNI natives of beagle.BeagleJNIWrapper — a JNI library lives in a JVM next to other
    # natives, so planner / launcher / device-stub symbols stay local (csrc/exports.map)
    exports = os.path.join(CSRC, "exports.map")
    if jobs or force or _newer(out, objs + [exports]):
        _run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib",
             "-Wl,--version-script=" + exports, "-o", out])
    return out


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libbeast_host.so")
    src = os.path.join(HOST, "tree_likelihood.cpp")
    if force or _newer(out, [src, os.path.join(ROOT, "include", "beagle_mi355.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", src, "-o", out])
    return out


def build_all(force=False):
    return [build_engine(force), build_host(force)]


if __name__ == "__main__":
    if "--lab" in sys.argv:
        print(build_engine(force="--force" in sys.argv, lab=True))
    else:
        build_all(force="--force" in sys.argv)
