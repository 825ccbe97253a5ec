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
    their s_waitcnt, no scratch, <= 128 VGPRs (cross-compiles gfx950 on the CPU box)."""
    out = subprocess.run(["python3", os.path.join(ROOT, "tools", "check_walk_isa.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_generated_assembly_loop_is_up_to_date_and_balanced():
    """csrc/walk4_fast_loop.inc is what tools/gen_walk4_fast.py emits now, and the stream is structurally sound: every
    out-of-line block returns, every label that is branched to exists exactly once, the fetch stage issues the four small
    loads the host's wait codes assume (kernels.h WF_WAIT8 / WF_WAIT12), and nothing above v123 / s81 is named (124 vector
    registers: four waves per SIMD)."""
    import re
    env = dict(os.environ, WALK4_CHECK_ONLY="1")
    env.pop("WALK4_EXPERIMENT", None)
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_walk4_fast.py")], env=env, capture_output=True, text=True)
    assert r.returncode == 0, "walk4_fast_loop.inc is stale: run python tools/gen_walk4_fast.py"
    text = open(os.path.join(ROOT, "beast-mcmc_amd", "csrc", "walk4_fast_loop.inc")).read()
    lines = re.findall(r'^\s+"(.*?)\\n(?:\\t)?" \\$', text, flags=re.M)
    labels = [l[:-1] for l in lines if l.endswith(":")]
    assert len(labels) == len(set(labels))
    targets = set(re.findall(r"s_c?branch\w* (\.LW4\w+_%=)", "\n".join(lines)))
    assert targets <= set(labels), targets - set(labels)
    # per stage: one LDS-DMA, two tip-pair loads, one reciprocal-pair load (prologue + two stages = 3 of each group)
    assert sum("global_load_lds_dwordx4" in l for l in lines) == 3
    assert sum(l.startswith("global_load_ushort") for l in lines) == 6
    regs = [int(x) for x in re.findall(r"\bv\[?(\d+)", "\n".join(lines))]
    assert max(regs) <= 123
    sregs = [int(x) for x in re.findall(r"\bs\[?(\d+)", "\n".join(lines))]
    assert max(sregs) <= 81 and not set(sregs) & {32, 33, 34, 35}
    assert lines[-1].startswith("s_waitcnt vmcnt(0)")
