#!/usr/bin/env python3
"""Dynamic instruction counts of the assembly walk loop (beast-mcmc_amd/csrc/walk4_fast_loop.inc) per micro-operation KIND, and
the issue-cycle model of DESIGN.md 4.1 built on them.

The loop is straight-line code with out-of-line blocks selected by bit tests on the micro-operation's flags, so the stream a
wave executes for a given (current, next) pair of micro-operations is found by interpreting the generated text itself: this
script follows labels and s_bitcmp1 / s_cbranch pairs with the flag values of the micro-operations at hand and classifies
every instruction it passes.  Read-mode programs only (write-mode rescaling has barriers: no per-wave stream).

    python tools/walk_cycle_model.py                      # the kinds, per-kind counts
    python tools/walk_cycle_model.py --plan <file>        # weighted by a program dumped with BEAGLE_MI355_DUMP_PLAN=2
                                                          #   (lines "[mi355]   k: k1 a k2 b hold h scale s store t")
"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "beast-mcmc_amd", "csrc", "walk4_fast_loop.inc")

# kernels.h
WK_MEM, WK_TIPS, WK_ACC, WK_H0, WK_H1, WK_H2 = 0, 1, 2, 3, 4, 5
WS_NONE, WS_READ, WS_WRITE = 0, 1, 2
WF_X, WF_T1, WF_T2, WF_INV, WF_STORE = 1, 2, 4, 8, 16
WF_HREAD, WF_HREAD1, WF_MEM2, WF_HWRITE, WF_HREAD2 = 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 30
DFL, FLS, ST = 44, (48, 56, 49), 29          # tools/gen_walk4_fast.py: the descriptor's flags, their stashes of the three pipeline slots


def walk_flags(k1, k2, hold, smode, store):
    f = (k1 << 5) | (k2 << 8) | (hold << 11) | (smode << 13)
    if k1 == WK_MEM: f |= WF_X
    if k1 == WK_TIPS: f |= WF_T1
    if k2 == WK_TIPS: f |= WF_T2
    if smode == WS_READ: f |= WF_INV
    if store: f |= WF_STORE
    if k1 >= WK_H0: f |= WF_HREAD | (WF_HREAD1 if k1 == WK_H1 else 0) | (WF_HREAD2 if k1 == WK_H2 else 0)
    if k2 == WK_MEM: f |= WF_MEM2
    if hold: f |= WF_HWRITE
    return f


def load_stream():
    text = open(INC).read()
    lines = re.findall(r'^\s+"(.*?)\\n(?:\\t)?" \\$', text, flags=re.M)
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    return lines, labels


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "dpp" in op or "row_newbcast" in ins or "quad_perm" in ins:
            return "valu_dpp"
        return "valu64" if ("f64" in op or "b64" in op) else "valu32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_"):
        return "vmem"
    if op.startswith("s_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op == "s_nop":
        return "nop"
    return "salu"


def run_iteration(lines, labels, flags):
    """One pass through the loop body = three stages (micro-operations 0, 1, 2 of `flags`; flags[2..5]: the ones being fetched
    behind them: the loop is three deep).  -> Counter per stage."""
    regs = {FLS[0]: flags[0], FLS[1]: flags[1], FLS[2]: 0, DFL: flags[2], ST: 0}
    pending = list(flags[3:6])               # what the descriptor loads deliver, stage by stage
    pc = labels[".LW4top_%="] + 1
    counts = [collections.Counter(), collections.Counter(), collections.Counter()]
    stage, scc, steps = 0, 0, 0
    while True:
        steps += 1
        if steps > 8000:
            raise RuntimeError("runaway")
        ins = lines[pc]
        pc += 1
        if ins.endswith(":"):
            if ins in (".LW4hwba_%=:", ".LW4hwbb_%=:"):       # behind a stage's hold-slot write (its last out-of-line block): the next stage
                stage += 1
                regs[DFL] = pending.pop(0) if pending else 0
            continue
        op = ins.split()[0]
        args = [a.strip() for a in ins[len(op):].split(",")]
        cls = classify(ins)
        counts[stage][cls] += 1
        if op == "s_bitcmp1_b32":
            r = int(args[0][1:])
            scc = (regs.get(r, 0) >> int(args[1])) & 1
        elif op == "s_and_b32" and args[1].startswith("s") and args[1][1:].isdigit() and not args[2].startswith("s"):
            regs[int(args[0][1:])] = regs.get(int(args[1][1:]), 0) & int(args[2], 0)
            scc = 1 if regs[int(args[0][1:])] else 0
        elif op == "s_bfe_u32":
            dst, src, spec = int(args[0][1:]), int(args[1][1:]), int(args[2], 16)
            regs[dst] = (regs.get(src, 0) >> (spec & 31)) & ((1 << (spec >> 16)) - 1)
        elif op == "s_cmp_eq_u32":
            scc = 1 if regs.get(int(args[0][1:]), 0) == int(args[1], 0) else 0
        elif op == "s_sub_u32" and args[0] == "s24":     # the loop counter: never the last micro-operation here
            scc = 0
        elif op == "s_cbranch_scc0" and args[0] == ".LW4top_%=":     # one iteration is what is counted
            return counts
        elif op == "s_mov_b32" and args[0].startswith("s") and args[1].startswith("s") and args[1][1:].isdigit():
            regs[int(args[0][1:])] = regs.get(int(args[1][1:]), 0)
        elif (op == "s_cbranch_scc1" and scc) or (op == "s_cbranch_scc0" and not scc) or op == "s_branch":
            pc = labels[args[0]]                   # (the label line itself is passed next: stage bookkeeping above)


KINDS = {
    "tip x tip": (WK_TIPS, WK_TIPS),
    "tip x ACC": (WK_TIPS, WK_ACC),
    "hold x ACC": (WK_H0, WK_ACC),
    "hold(reg) x ACC": (WK_H2, WK_ACC),
    "mem x ACC": (WK_MEM, WK_ACC),
    "mem x tip": (WK_MEM, WK_TIPS),
}
# issue cycles per instruction of a wave64 on one SIMD (CDNA3/4: 16 lanes per cycle; fp64 FMA and DPP at full rate — tools/walk_probe.hip)
CYCLES = {"valu64": 4, "valu32": 4, "valu_dpp": 4, "lds": 4, "vmem": 4, "smem": 1, "salu": 1, "branch": 1, "wait": 1, "nop": 1}


def stage_counts(lines, labels, cur, nxt, far=None):
    """counts of the stage that computes micro-operation `cur` while `nxt` and `far` are in flight behind it."""
    far = nxt if far is None else far
    c = run_iteration(lines, labels, [cur, nxt, far, far, far, far])
    return c[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", default="")
    args = ap.parse_args()
    lines, labels = load_stream()
    typical_next = walk_flags(WK_TIPS, WK_ACC, 0, WS_READ, False)
    rows = {}
    for name, (k1, k2) in KINDS.items():
        for extra, (hold, store) in (("", (0, False)), (" +hold", (1, False)), (" +store", (0, True))):
            f = walk_flags(k1, k2, hold, WS_READ, store)
            c = stage_counts(lines, labels, f, typical_next)
            rows[name + extra] = c
    print("%-24s %6s %6s %6s %5s %5s %5s %5s %6s %6s | issue cycles" % ("micro-operation", "valu64", "dpp", "valu32", "lds", "vmem", "smem", "salu", "branch", "wait"))
    for name, c in rows.items():
        cyc = sum(CYCLES[k] * v for k, v in c.items())
        print("%-24s %6d %6d %6d %5d %5d %5d %5d %6d %6d | %d (VALU %d)" % (name, c["valu64"], c["valu_dpp"], c["valu32"], c["lds"], c["vmem"], c["smem"], c["salu"],
                                                                              c["branch"], c["wait"], cyc, 4 * (c["valu64"] + c["valu_dpp"] + c["valu32"])))
    if args.plan:
        prog = []
        for ln in open(args.plan):
            m = re.search(r"\[mi355\]\s+(\d+): k1 (\d+) k2 (\d+) hold (\d+) scale (\d+) store (\d+)", ln)
            if m:
                prog.append(tuple(int(x) for x in m.groups()[1:]))
        if not prog:
            raise SystemExit("no micro-operations in %s" % args.plan)
        total = collections.Counter()
        for i, (k1, k2, hold, sm, store) in enumerate(prog):
            nk = prog[i + 1] if i + 1 < len(prog) else prog[i]
            fk = prog[i + 2] if i + 2 < len(prog) else nk
            cur = walk_flags(k1, k2, hold, sm, bool(store))
            nxt = walk_flags(nk[0], nk[1], nk[2], nk[3], bool(nk[4]))
            far = walk_flags(fk[0], fk[1], fk[2], fk[3], bool(fk[4]))
            total.update(stage_counts(lines, labels, cur, nxt, far))
        n = len(prog)
        print("\nprogram of %d micro-operations (%s), per micro-operation and wave:" % (n, args.plan))
        for k in ("valu64", "valu_dpp", "valu32", "lds", "vmem", "smem", "salu", "branch", "wait", "nop"):
            print("  %-9s %7.1f instructions  %7.1f issue cycles" % (k, total[k] / n, CYCLES[k] * total[k] / n))
        valu = 4 * (total["valu64"] + total["valu_dpp"] + total["valu32"]) / n
        allc = sum(CYCLES[k] * v for k, v in total.items()) / n
        print("  VALU cycles %.0f, all issue cycles %.0f per micro-operation and wave" % (valu, allc))


if __name__ == "__main__":
    main()
