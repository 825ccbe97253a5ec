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


def test_isa_check_sees_a_copy_made_before_the_wait():
    """The checker of hand-waited assembly loads (tools/check_walk_isa.py check_async) on a listing with exactly the defect it
    exists for — the register allocator copying a load's destination BEFORE the wait that retires it (what a tied asm operand
    produced in an early build of k_walkT32) — and on the repaired order."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_walk_isa", os.path.join(ROOT, "tools", "check_walk_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    bad = """.LBB0_1:                               ; =>This Inner Loop Header: Depth=1
	;;#ASMSTART
	global_load_ushort v167, v160, s[36:37]
	global_load_dwordx4 v[14:17], v164, s[4:5]
	;;#ASMEND
	v_add_u32_e32 v1, v2, v3
	v_mov_b32_e32 v152, v167
	;;#ASMSTART
	s_waitcnt vmcnt(5) ; retires v167 v[14:17]
	;;#ASMEND
	v_mov_b32_e32 v130, v14
	s_cbranch_scc1 .LBB0_1""".split("\n")
    problems = chk.check_async("bad", bad)
    assert len(problems) >= 1 and "v_mov_b32_e32 v152, v167" in problems[0]
    good = [l for l in bad if "v152, v167" not in l]
    good.insert(good.index("\tv_mov_b32_e32 v130, v14"), "\tv_mov_b32_e32 v152, v167")
    assert chk.check_async("good", good) == []



def test_generated_assembly_loop_is_up_to_date_and_balanced():
    """csrc/walk4_fast_loop.inc is what tools/gen_walk4_fast.py emits now, and the stream is structurally sound: every
    out-of-line block returns, every label that is branched to exists exactly once, every fetch issues the loads the host's
    wait codes assume (kernels.h walkWaitCode, engine_walk.cpp fetchLoads: the matrix table, two tip-state pairs — each behind its
    skip test —, the reciprocal pair out of line, a fused cherry's table half and two tip-state pairs out of line), the wait
    table holds every count from 1 to 16 once, and nothing above v125 / s86 is named (126 vector registers: four waves per SIMD)."""
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
    # per fetch (two in the prologue + three stages = 5): one LDS-DMA and one more for a fused cherry, two tip-pair loads and two more
    # for a fused cherry, one reciprocal-pair load
    assert sum("global_load_lds_dwordx4" in l for l in lines) == 10
    assert sum(l.startswith("global_load_ushort") for l in lines) == 20
    for tag in "pqabc":                               # each tip-pair load sits right behind its own skip test
        for n, bit in (("n1", 16), ("n2", 17)):
            k = lines.index(".LW4%s%s_%%=:" % (n, tag))
            assert lines[k - 1].startswith("global_load_ushort") and lines[k - 2] == "s_cbranch_scc1 .LW4%s%s_%%=" % (n, tag) and lines[k - 3].endswith(", %d" % bit)
    waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", "\n".join(lines))]
    assert sorted(set(waits) - {0}) == list(range(1, 17))
    assert sum(l.startswith("global_load_dwordx4 v[22:25]") or l.startswith("global_load_dwordx4 v[26:29]") or l.startswith("global_load_dwordx4 v[30:33]") for l in lines) == 5
    regs = [int(x) for x in re.findall(r"\bv\[?(\d+)", "\n".join(lines))]
    assert max(regs) <= 125
    sregs = [int(x) for x in re.findall(r"\bs\[?(\d+)", "\n".join(lines))]
    assert max(sregs) <= 87 and not set(sregs) & {32, 33, 34, 35}
    assert lines[-1].startswith("s_waitcnt vmcnt(0)")
