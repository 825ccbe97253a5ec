"""The walk planner (beast-mcmc_amd/csrc/planner.cpp) against list-order evaluation, on the CPU.

tests/native/plan_check.cpp drives the planner exactly as the engine does (hazard-free prefixes, materialise-before,
plan, materialise-on-demand) and executes its micro-operation programs with an index-level interpreter of the walk
kernel's register model; every real buffer and scale buffer must agree bitwise with evaluating the same operation lists
one op after the other (lib/beagle.jar!beagle/GeneralBeagleImpl#updatePartials semantics), through MCMC-style partial
updates with BufferIndexHelper flips and rejections (src/dr/evomodel/treedatalikelihood/BufferIndexHelper.java:71-106)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_programs_equal_list_order_evaluation(tmp_path):
    exe = str(tmp_path / "plan_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-Wall", "-fsanitize=address,undefined",
                           os.path.join(ROOT, "tests", "native", "plan_check.cpp"),
                           os.path.join(ROOT, "beast-mcmc_amd", "csrc", "planner.cpp"), "-o", exe])
    out = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "plan_check: OK" in out.stdout


def test_walk_kernel_isa_keeps_in_flight_registers_untouched():
    """tools/check_walk_isa.py: the hand-pipelined kernel's in-flight load destinations are not read or written before
    their s_waitcnt, no scratch, <= 72 VGPRs (cross-compiles gfx950 on the CPU box)."""
    out = subprocess.run(["python3", os.path.join(ROOT, "tools", "check_walk_isa.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
