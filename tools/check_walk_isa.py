#!/usr/bin/env python3
"""Static check of the hand-pipelined pattern-walk kernel (beast-mcmc_amd/csrc/kernels_walk4.hip).

The kernel keeps the loads of micro-operation k+1 in flight while k computes; the loads are inline assembly, so the
compiler's own s_waitcnt bookkeeping does not know their destination registers are pending.  That is only correct if
NOTHING reads or writes those registers between a fetch block and the `s_waitcnt vmcnt(9)` that retires it (one stage
later).  This script compiles the kernel to gfx950 assembly and verifies exactly that for every instantiation, plus the
resource figures the design depends on (no scratch, <= 128 VGPRs so that 4 waves fit a SIMD).
Exit code 0 = ok.  Runs without a GPU (hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "beast-mcmc_amd", "csrc", "kernels_walk4.hip")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def all_regs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs(tok)
    return out


def check_function(name, lines):
    problems = []
    want = 24                       # 16 partials + 4 tip states + 2 reciprocal scale factors (4 registers); the matrices go to LDS
    waits = [i for i, l in enumerate(lines) if "s_setpc_b64" in l]          # the jump into the s_waitcnt table = the stage's wait
    blocks = []
    i = 0
    while i < len(lines):
        if re.match(r"\s*s_bitcmp1_b32 s\d+, 0$", lines[i]) and i + 1 < len(lines) and ".Lfx" in lines[i + 1]:
            j, dst = i, set()
            while j < len(lines):
                t = lines[j].strip()
                if t.startswith(".Lfi"):             # the label after the last group ends the block
                    j += 1
                    break
                if t.startswith("global_load"):
                    dst |= regs(t.split()[1].rstrip(","))
                elif not (t.startswith("s_bitcmp1") or t.startswith("s_cbranch_scc0") or t.startswith(".Lf")):
                    break
                j += 1
            blocks.append((i, j, dst))
            i = j
        else:
            i += 1
    labels = [i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)]
    loop_blocks = [b for b in blocks if labels and b[0] > labels[0]]
    if len(blocks) != 3 or len(loop_blocks) != 2 or len(waits) != 2:
        problems.append("%s: expected 1 prologue + 2 loop fetch blocks and 2 waits, found %d / %d / %d"
                        % (name, len(blocks) - len(loop_blocks), len(loop_blocks), len(waits)))
        return problems
    for (a, b, dst) in blocks:
        if len(dst) != want:
            problems.append("%s: fetch block at line %d writes %d registers, expected %d" % (name, a, len(dst), want))
    loop_start = labels[0]
    for (a, b, dst) in loop_blocks:
        later = [w for w in waits if w >= b]
        if len(later) >= 2:
            span = list(range(b, later[1]))
        else:        # wraps around the loop: to the end of the function text, then from the loop start
            span = list(range(b, len(lines))) + list(range(loop_start, [w for w in waits if w > loop_start][1 - len(later)]))
        for k in span:
            if "s_waitcnt" not in lines[k] and all_regs(lines[k]) & dst:
                problems.append("%s: line %d touches in-flight registers of the fetch at line %d: %s" % (name, k, a, lines[k].strip()))
    # the prologue fetch must land in the registers the second loop block refills (no copies on the back edge)
    if blocks[0][2] != loop_blocks[1][2]:
        problems.append("%s: prologue fetch registers differ from the loop's second fetch block" % name)
    # every vmcnt wait is ours: a table entry (followed by its branch / the table end) or a full drain
    for k, l in enumerate(lines):
        m = re.match(r"\s*s_waitcnt vmcnt\((\d+)\)\s*$", l)
        if m and int(m.group(1)) != 0:
            nxt = lines[k + 1].strip() if k + 1 < len(lines) else ""
            if not (nxt.startswith("s_branch .Lwd") or nxt.startswith(".Lwd")):
                problems.append("%s: unexpected compiler-inserted wait at line %d: %s" % (name, k, l.strip()))
    return problems


def check_async(name, text_lines):
    """Kernels whose loads are inline assembly with their waits written by hand (k_walkT32, k_preWalk4): the destination
    registers of an assembly load block are written asynchronously, so between the block and the hand-written wait that
    names them ("; retires ...") NO instruction may read or write them — a register-allocator copy there would read a
    value that has not arrived.  Linear scan of the listing, the loop body a second time for the back edge."""
    problems = []
    header = None
    lines = []
    for l in text_lines:
        if header is None and re.match(r"\.LBB\d+_\d+:", l) and "Loop Header" in l:
            header = len(lines)
        if not l.strip().startswith(";") or "ASMSTART" in l or "ASMEND" in l:
            lines.append(l)
    order = list(range(len(lines))) + (list(range(header, len(lines))) if header is not None else [])
    inflight, in_asm, issues, retires = {}, False, 0, 0
    for idx in order:
        t = lines[idx].strip()
        if "ASMSTART" in t:
            in_asm = True
            continue
        if "ASMEND" in t:
            in_asm = False
            continue
        if in_asm and t.startswith("global_load"):
            for r in regs(t.split()[1].rstrip(",")):
                inflight[r] = idx
            issues += 1
            continue
        if "retires" in t:
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", t.split("retires", 1)[1]):
                for r in regs(tok):
                    inflight.pop(r, None)
            retires += 1
            continue
        if re.match(r"s_waitcnt vmcnt\(0\)", t):
            inflight.clear()
            continue
        if t.startswith("s_waitcnt") or t.startswith("s_") or t.startswith("."):
            continue
        hit = all_regs(t) & set(inflight)
        if hit:
            problems.append("%s: line %d touches registers %s of the assembly load at line %d before their wait: %s"
                            % (name, idx, sorted(hit)[:4], inflight[sorted(hit)[0]], t))
    if issues == 0 or retires == 0:
        problems.append("%s: no assembly load blocks / no 'retires' waits found (%d / %d)" % (name, issues, retires))
    return problems


ASYNC_KERNELS = (("kernels_mfma.hip", r"_ZN5mi3559k_walkT32", 168), ("kernels_mfma.hip", r"_ZN5mi35511k_walkT32W1", 168),
                 ("kernels_preorder4.hip", r"_ZN5mi35510k_preWalk4", 128))


def check_async_kernels(hipcc):
    problems, summary = [], []
    for src, prefix, max_vgpr in ASYNC_KERNELS:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip",
                                os.path.join(ROOT, "beast-mcmc_amd", "csrc", src), "-o", out], capture_output=True, text=True)
            if r.returncode:
                return [r.stderr], summary
            text = open(out).read()
        funcs = re.findall(r"^(%s[^:\n]*):\s*;.*?\n(.*?)s_endpgm" % prefix, text, flags=re.S | re.M)
        if not funcs:
            problems.append("%s: no %s instantiation found" % (src, prefix))
        for name, body in funcs:
            problems += check_async(name[:48], body.split("\n"))
            meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), text, flags=re.S)
            vg = int(re.search(r"\.vgpr_count:\s+(\d+)", meta.group(1)).group(1)) if meta else -1
            sc = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta.group(1)).group(1)) if meta else -1
            summary.append("%s %d VGPRs" % (name[9:24], vg))
            if vg > max_vgpr or vg < 0:
                problems.append("%s: %d VGPRs (limit %d)" % (name[:48], vg, max_vgpr))
            if sc != 0:
                problems.append("%s: %d bytes of scratch per lane — a spilled in-flight register cannot work" % (name[:48], sc))
    return problems, summary


def main():
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "walk4.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip", SRC, "-o", out,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr)
            return 2
        text = open(out).read()
        remarks = r.stderr
    problems = []
    vg = [int(x) for x in re.findall(r"VGPRs: (\d+)", remarks)]
    sc = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", remarks)]
    if not vg or max(vg) > 128:
        problems.append("VGPRs %s: more than 128 (4 waves per SIMD are needed to keep a 1e5-pattern alignment resident)" % vg)
    if any(sc):
        problems.append("scratch in use: %s" % sc)
    funcs = re.findall(r"^(_ZN5mi3557k_walk4[^:\n]*):\s*;.*?\n(.*?)s_endpgm", text, flags=re.S | re.M)
    if len(funcs) != 3:
        problems.append("found %d kernel instantiations, expected 3" % len(funcs))
    for name, body in funcs:
        lines = [l for l in body.split("\n") if not l.strip().startswith(";")]
        problems += check_function(name[:40], lines)
    more, summary = check_async_kernels(hipcc)
    problems += more
    for p in problems:
        print("PROBLEM:", p)
    print("walk kernel ISA check: %s (VGPRs %s, %d instantiations; %s)" % ("FAILED" if problems else "ok", vg, len(funcs), ", ".join(summary)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
