#!/usr/bin/env python3
"""Generates beast-mcmc_amd/csrc/walk4_fast_loop.inc: the main loop of k_walk4_fast (kernels_walk4.hip) as ONE block of
gfx950 assembly with explicit registers.

Why assembly: the pattern walk executes ~100 instructions per micro-operation and wave here against ~200 from the C++ kernel
(45 % of those scalar / branch glue the compiler's control-flow structurizer adds around the hand-pipelined loads), with 2-3
taken branches per micro-operation: everything rare (partials loads, hold-slot traffic, stores, a second child in memory) is
out of line, the small loads are unconditional (the host points unused operands at dummy buffers: all-missing tip states,
all-one scale factors), and what the compute half of a stage needs from a descriptor is stashed in scalar registers when
the fetch half has used it, so one descriptor register set serves the software pipeline.

The pipeline is THREE micro-operations deep (round 4; two before): while micro-operation k computes, the small loads of k + 1
AND k + 2 — matrix table by LDS-DMA, two tip-state pairs, the reciprocal scale factors — are in flight, in three rotating
register / LDS-table slots, and descriptor k + 3 is on its way.  A wave's stage used to be as long as the round trip of the
loads issued at its start (481 ns per micro-operation for a wave alone on its SIMD; with the memory system loaded by the other
waves' traffic that round trip grows: profiles/r04_experiments.txt 12), now two stages cover it.  The registers for the third
slot come from the first-child partials buffers: a first child in memory (2 % of the micro-operations) or in an LDS hold slot
is requested in the MIDDLE of the stage before its use, into one buffer instead of two.

It covers every micro-operation, write-mode rescaling included (rescale_block: LDS atomics across the category waves, one
barrier, a true division).  What it requires is that every segment starts at a multiple of 128 patterns — the engine pads
partitions internally to arrange that — and three readable descriptors behind a program (engine_walk.cpp runPlan; any length:
the loop can leave behind each of its three stages); the C++ kernel k_walk4 computes the same values bit for bit and is the reference implementation
(BEAGLE_MI355_NO_FAST_WALK=1, tests/test_gpu_walk_kernels.py).

Run: python tools/gen_walk4_fast.py   (rewrites the .inc; tests/test_planner_native.py checks it is up to date)"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("WALK4_OUT") or os.path.join(ROOT, "beast-mcmc_amd", "csrc", "walk4_fast_loop.inc")

# ---- register map -----------------------------------------------------------------------------------------------------
# vector
XB = 0                        # a first child that comes from memory or from an LDS hold slot: ONE buffer (pattern a = +0..7, b = +8..15),
#                               requested in the middle of the stage before its use; between uses a scratch area of the second mat-vec
T1S, T2S = (16, 17, 18), (19, 20, 21)    # tip-state pairs (a | b << 8) of the two children: three pipeline slots
INVS = (22, 26, 30)           # reciprocal scale factors {a, b}: three pipeline slots
ACC = 44                      # the previous micro-operation's result: a = +0..7, b = +8..15
F, G = 60, 76                 # the two children's contributions
SP = 92                       # lane l: entry (l & 15) of the branch matrix at hand
T0, T1 = 94, 95
PA, PB, TIP, SCALE, OM, HOLD, LANE, VST = 96, 97, 98, 99, 100, 101, 104, 105
SPS = (102, 103, 34)          # the lane's LDS address inside the three table buffers
H2 = 106                      # the third hold slot lives in registers (LDS holds two: 32 KiB of the 40 a workgroup may use)
TCAS, TCBS = (36, 37, 38), (39, 40, 41)  # tip-state pairs of a FUSED CHERRY's two tips (B_CHERRY): three pipeline slots
# ... and, in programs that rescale in write mode (which have no fused cherries: engine_walk.cpp runPlan), the running PRODUCT of the
# factors the slice has written for the lane's two patterns, as mantissa in [0.5, 1) and binary exponent (rescale_block, category 0's wave)
PMA, PMB, PEA, PEB = 36, 38, 40, 41
TBVS = (124, 125, 35)         # the LDS addresses of the three table buffers, in every lane (broadcast reads of a matrix's first column)
NV = 126
# scalar (s32..s35 are left to the compiler)
DP, STRM, CNT, HSTRIDE, STEP, ST, LAST, CM0 = 20, 22, 24, 27, 28, 29, 30, 31
TBLS = (25, 26, 57)           # LDS addresses of the three table buffers of this wave
D, DFL, DW = 36, 44, 46             # descriptor (kernels.h WalkOp): src1 D+0, src2 D+2, store D+4, scale D+6 | flags s44, (pad s45), the scale buffer a rescaling operation writes s46:47 — two loads: x8 at 0, x4 at 0x20
CM160 = 83
TBLBS = (84, 85, 86)          # LDS addresses of the cherry halves of the three table buffers of this wave (TBLS + 3 tblStep)
CHOFF = 87                    # byte offset of the stream's cherry region from its main region (the same entry index in both)
S_LAST = 87
FLS, SCALEWS = (48, 56, 49), (54, 62, 58)      # what a compute stage needs of its descriptor, stashed by the fetch: three pipeline slots
SSTORE, SSRC2, SX = 50, 52, 60                 # addresses the rare blocks read again from the descriptor (store, second child, first child)
MASK = 64                     # MASK + 2 j: lanes whose piece of store instruction j lies inside the pattern range (4 pairs)
VALA, VALB = 72, 74           # lanes whose first / second pattern lies inside the range (scale-factor stores)
DIVS, SCNT, EXCH, NCAT, ROFF, RB = 76, 78, 79, 80, 81, 82    # RB: byte offset of the rescaling maximum buffer the next node uses
S_FIRST = 20

# flag bits (kernels.h)
B_X, B_T1, B_T2, B_INV, B_STORE, B_HSLOT1, B_READ, B_WRITE = 0, 1, 2, 3, 4, 12, 13, 14     # (B_READ / B_WRITE: the scale mode, WS_READ = 1, WS_WRITE = 2 at bit 13)
# B_CHERRY (kernels.h WF_CHERRY2, round 6): the second child is a node over two compact tips that is evaluated INSIDE this stage — its
# own micro-operation is not in the program (engine_walk.cpp runPlan fuses it away: a third of a tree's nodes are such cherries, and a
# stage costs ~60 instructions of overhead whatever it computes).  The descriptor's src2 / scale fields then carry the two tips' state
# arrays, the stream entry's second half the cherry's two branch-matrix tables; the value is column(A) * column(B), formed in ACC,
# and the stage goes on as if the previous micro-operation had left it there.  Same arithmetic, same order, same bits.
B_CHERRY = 15
B_HREAD, B_HREAD1, B_MEM2, B_HWRITE, B_HREAD2 = 24, 25, 26, 27, 31
# bits 16..23 belong to the kernel that runs the program (k_walk4: its wait-table jump); here: the two tip-state loads of a fetch are
# SKIPPED under B_NOLOAD1 / B_NOLOAD2 (round 6: the host sets them, in programs that do not rescale in write mode, for a child that is
# no compact tip — half the children of a tree; a vector-memory instruction occupies the CU's address unit for ~16 cycles whatever it
# loads, and that unit is one of the three pipes the loop shares between its sixteen waves), and the stage's wait as a 4-bit code
B_NOLOAD1, B_NOLOAD2, B_WAIT0 = 16, 17, 18
# the stage's wait as a 3-bit code at B_WAIT0 (kernels.h walkWaitCode): vmcnt(N) with N = WAIT_N[code]; code 0 is the common one
# (a fetch is THREE small loads — matrix table, two tip-state pairs — and a fourth, the reciprocal scale factors, only for a
# micro-operation that multiplies by them: since round 5 a read-mode program applies the factors of unstored results once, at the
# stored result above them, so nine micro-operations in ten fetch three)
# (round 6: a fused cherry adds three loads to a fetch — its table half, two more tip-state pairs —, so 9 = a plain and a fused fetch
# in flight is the second most common count and gets the second test)
WAIT_N = (6, 9, 7, 8, 10, 12, 3, 4)
# round 6, second step: with the tip-state loads skipped where a child is no tip a fetch is 1..6 loads and every count from 1 to 16 occurs:
# a 4-bit code, every count exact; code 0 (inline) = 4, codes 1 and 2 (a test each) = 3 and 5, the rest through the jump table
WAIT_N = (4, 3, 5, 2, 6, 7, 8, 9, 10, 11, 12, 1, 13, 14, 15, 16)

# Cache policy of the result stores and of the loads that read stored results back (a first child in memory, a second child
# in memory).  sc1 = device scope: the store is written through to memory before it is acknowledged, the load does not take a
# line another XCD's L2 may hold newer — what lets the slices of a program run in ONE launch and hand results over through
# flags (kernels_walk4.hip k_walk4_fast) without writing back or invalidating a whole L2 per workgroup (buffer_wbl2 /
# buffer_inv per workgroup: 672 against 126 us on the 12 500-pattern shard, profiles/r04_experiments.txt).  Both kinds of
# access stream through HBM once anyway.
STORE_POLICY = os.environ.get("WALK4_STORE_POLICY", " sc1 nt")
LOAD_POLICY = os.environ.get("WALK4_LOAD_POLICY", " sc1")
# TIMING EXPERIMENTS ONLY (wrong results; tools/walk_floor.sh, profiles/r02_experiments.txt): comma-separated parts to leave out
EXPERIMENT = set(x for x in os.environ.get("WALK4_EXPERIMENT", "").split(",") if x)
# TIMING EXPERIMENTS ONLY (same results): WALK4_PAD=<kind>:<n> adds n do-nothing instructions to every stage — snop (s_nop 0: 4 bytes,
# scalar), slit (s_mov_b32 with a 32-bit literal into a scratch register: 8 bytes, scalar), vmov (v_mov_b32 of a scratch register
# onto itself: 4 bytes, vector), vlit (v_mov_b32 of a literal into a scratch register: 8 bytes, vector) — to tell what the loop
# is bound by: instruction count, instruction bytes or the vector pipe (profiles/r04_experiments.txt)
PAD = os.environ.get("WALK4_PAD", "")
COL0 = os.environ.get("WALK4_COL0", "1") != "0"            # a mat-vec's first column from broadcast LDS reads (matvec): A/B switch
lines = []


def pad_block():
    if not PAD:
        return
    kind, n = PAD.split(":")
    for _ in range(int(n)):
        e({"snop": "s_nop 0", "slit": "s_mov_b32 %s, 0x12345678" % s(ST), "vmov": "v_mov_b32_e32 %s, %s" % (v(T1), v(T1)),
           "vlit": "v_mov_b32_e32 %s, 0x12345678" % v(T1)}[kind])


def e(s):
    lines.append(s)


def v(n, w=1):
    return "v%d" % n if w == 1 else "v[%d:%d]" % (n, n + w - 1)


def s(n, w=1):
    return "s%d" % n if w == 1 else "s[%d:%d]" % (n, n + w - 1)


def L(name):
    return ".LW4%s_%%=" % name


def matvec(dst, x, sp=None, col0=None, col0v=None):
    """dst (16 regs: a rows 0-3, b rows 0-3) = M . x for the two patterns; M spread over the lanes of `sp`.  The rounding
    sequence is y_i = fma(m_i3, x3, fma(m_i2, x2, fma(m_i1, x1, m_i0 * x0))) (= fma(m_i0, x0, 0) bit for bit: both
    factors are never negative).
    col0v: eight vector registers that hold M[0..3][0] in every lane (two broadcast ds_read_b128 of the table's first column,
    issued by the caller together with the read of `sp`): the first term is then a plain multiply and nothing has to be zeroed
    — v_mul_f64 has no DPP form, so without them the accumulators are cleared by eight v_mov_b64 and the first term is a
    fourth DPP multiply-add (40 instead of 32 vector instructions; the loop is bound by instruction issue).
    col0: the same from SGPRs (an experiment of round 3)."""
    sp = SP if sp is None else sp
    if not COL0:
        col0v = None
    if col0v is not None:
        for half in range(2):
            for i in range(4):
                e("v_mul_f64 %s, %s, %s" % (v(dst + 8 * half + 2 * i, 2), v(col0v + 2 * i, 2), v(x + 8 * half, 2)))
    elif col0 is None:
        for i in range(8):
            e("v_mov_b64 %s, 0" % v(dst + 2 * i, 2))
    else:
        for half in range(2):
            for i in range(4):
                e("v_mul_f64 %s, %s, %s" % (v(dst + 8 * half + 2 * i, 2), s(col0 + 2 * i, 2), v(x + 8 * half, 2)))
    for j in range(0 if "nofma" in EXPERIMENT else 4):
        if j == 0 and (col0 is not None or col0v is not None):
            continue
        for half in range(2):
            for i in range(4):
                e("v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf"
                  % (v(dst + 8 * half + 2 * i, 2), v(sp, 2), v(x + 8 * half + 2 * j, 2), 4 * i + j))


def col0_reads(tmp, tbv, off):
    """the table's first column (M[0..3][0]: its first 32 bytes, kernels_walk4.hip k_gatherMatrices) into tmp..tmp+7, every lane"""
    if not COL0:
        return
    e("ds_read_b128 %s, %s offset:%d" % (v(tmp, 4), v(tbv), off))
    e("ds_read_b128 %s, %s offset:%d" % (v(tmp + 4, 4), v(tbv), off + 16))


def tip_columns(dst, t, tbl, off):
    e("v_and_b32 %s, 0xff, %s" % (v(T0), v(t)))
    e("v_lshrrev_b32 %s, 8, %s" % (v(T1), v(t)))
    e("v_lshl_add_u32 %s, %s, 5, %s" % (v(T0), v(T0), s(tbl)))
    e("v_lshl_add_u32 %s, %s, 5, %s" % (v(T1), v(T1), s(tbl)))
    if "notipread" in EXPERIMENT:
        return
    e("ds_read_b128 %s, %s offset:%d" % (v(dst, 4), v(T0), off))
    e("ds_read_b128 %s, %s offset:%d" % (v(dst + 4, 4), v(T0), off + 16))
    e("ds_read_b128 %s, %s offset:%d" % (v(dst + 8, 4), v(T1), off))
    e("ds_read_b128 %s, %s offset:%d" % (v(dst + 12, 4), v(T1), off + 16))


outofline = []      # (label, [lines]) blocks placed after the loop


def fdiv_one(out, den, d0, r, t, n0, q):
    """out = 1.0 / den, correctly rounded: the sequence the compiler emits for a double division (k_walk4 has the same)."""
    return ["v_div_scale_f64 %s, %s, %s, %s, 1.0" % (v(d0, 2), s(DIVS, 2), v(den, 2), v(den, 2)),
            "v_rcp_f64_e32 %s, %s" % (v(r, 2), v(d0, 2)),
            "s_nop 3",                              # a transcendental result must not be read by the very next vector instruction
            "v_fma_f64 %s, -%s, %s, 1.0" % (v(t, 2), v(d0, 2), v(r, 2)),
            "v_fmac_f64_e32 %s, %s, %s" % (v(r, 2), v(r, 2), v(t, 2)),
            "v_fma_f64 %s, -%s, %s, 1.0" % (v(t, 2), v(d0, 2), v(r, 2)),
            "v_fmac_f64_e32 %s, %s, %s" % (v(r, 2), v(r, 2), v(t, 2)),
            "v_div_scale_f64 %s, vcc, 1.0, %s, 1.0" % (v(n0, 2), v(den, 2)),
            "v_mul_f64 %s, %s, %s" % (v(q, 2), v(n0, 2), v(r, 2)),
            "v_fma_f64 %s, -%s, %s, %s" % (v(t, 2), v(d0, 2), v(q, 2), v(n0, 2)),
            "s_nop 3",
            "v_div_fmas_f64 %s, %s, %s, %s" % (v(t, 2), v(t, 2), v(r, 2), v(q, 2)),
            "v_div_fixup_f64 %s, %s, %s, 1.0" % (v(out, 2), v(t, 2), v(den, 2))]


def rescale_block(tag, SSCALEW):
    """Write-mode rescaling of the result in ACC (AbstractLikelihoodCore.java:406-440 applied unconditionally, as k_walk4):
    per pattern the largest entry over the 4 states and ALL rate categories, a factor that is not positive becomes 1, the
    result is multiplied by the reciprocal (a true division, so that the bits equal k_walk4's); the wave of category 0 stores
    the factor (plain layout) and the reciprocal (pair-interleaved, ROFF bytes further on) of the lane's two patterns.
    F and G are free here.

    The maximum over the categories — one category per wave — is formed by LDS ATOMICS (ds_max_f64 of every wave's own maximum
    into the lane's two words of a 1 KiB buffer that starts at zero), then ONE barrier, then every wave reads the result.  Three
    such buffers rotate (RB): the one a node uses was zeroed by the category-0 wave two nodes earlier, behind that node's barrier
    — by then every wave had read it for the last time (it read it before arriving at that barrier), and nobody adds to it
    before the barrier after, which the zeroing wave only reaches with its LDS writes done.  Round 3 exchanged the maxima through
    a [category][lane] array between TWO barriers and drained the memory queue behind the factor stores (A with ALWAYS
    rescaling: 1 228 us against 622 in read mode); nothing needs the drain: the stage waits count loads only, and a wait that
    sees more outstanding stores than it expects only waits longer (engine_walk.cpp runPlan)."""
    MA, MB, RD, A2, B2, IA, IB = F, F + 2, F + 4, G, G + 2, G + 4, G + 6
    d0, r, t, n0, q = F + 8, F + 10, F + 12, F + 14, G + 8
    b = [L("wr" + tag) + ":"]
    for m, base in ((MA, ACC), (MB, ACC + 8)):
        b += ["v_max_f64 %s, 0, %s" % (v(m, 2), v(base, 2)),
              "v_max_f64 %s, %s, %s" % (v(RD, 2), v(base + 2, 2), v(base + 4, 2)),
              "v_max_f64 %s, %s, %s" % (v(m, 2), v(m, 2), v(RD, 2)),
              "v_max_f64 %s, %s, %s" % (v(m, 2), v(m, 2), v(base + 6, 2))]
    b += ["v_lshlrev_b32_e32 %s, 4, %s" % (v(T0), v(LANE)),
          "v_add_u32_e32 %s, %s, %s" % (v(T0), s(EXCH), v(T0)),           # the lane's 16 bytes of buffer 0
          "v_add_u32_e32 %s, %s, %s" % (v(T1), s(RB), v(T0)),              # ... of this node's buffer
          "ds_max_f64 %s, %s" % (v(T1), v(MA, 2)),
          "ds_max_f64 %s, %s offset:8" % (v(T1), v(MB, 2)),
          "s_waitcnt lgkmcnt(0)",
          "s_barrier",
          "ds_read_b128 %s, %s" % (v(A2, 4), v(T1)),
          # the buffer two nodes on = the one used one node ago: every wave is past its reads of it; category 0 zeroes it
          "s_add_u32 %s, %s, 0x800" % (s(ST), s(RB)),
          "s_cmp_ge_u32 %s, 0xc00" % s(ST),
          "s_cbranch_scc0 %s" % L("wrz" + tag),
          "s_sub_u32 %s, %s, 0xc00" % (s(ST), s(ST)),
          L("wrz" + tag) + ":",
          "s_cmp_lg_u32 %s, 0" % s(SCNT),
          "s_cbranch_scc1 %s" % L("wrn" + tag),
          "v_add_u32_e32 %s, %s, %s" % (v(T0), s(ST), v(T0)),
          "v_mov_b64 %s, 0" % v(RD, 2), "v_mov_b64 %s, 0" % v(RD + 2, 2),
          "ds_write_b128 %s, %s" % (v(T0), v(RD, 4)),
          L("wrn" + tag) + ":",
          "s_add_u32 %s, %s, 0x400" % (s(RB), s(RB)),                       # next node: next buffer
          "s_cmp_ge_u32 %s, 0xc00" % s(RB),
          "s_cbranch_scc0 %s" % L("wrr" + tag),
          "s_mov_b32 %s, 0" % s(RB),
          L("wrr" + tag) + ":",
          "s_waitcnt lgkmcnt(0)"]
    b.append("v_mov_b32_e32 %s, 0x3ff00000" % v(T1))                        # (a literal and VCC together exceed the constant bus)
    for m in (A2, B2):                                                      # if (!(m > 0)) m = 1
        b += ["v_cmp_lt_f64_e32 vcc, 0, %s" % v(m, 2),
              "s_nop 1",
              "v_cndmask_b32_e32 %s, %s, %s, vcc" % (v(m + 1), v(T1), v(m + 1)),
              "v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(m), v(m))]
    b += fdiv_one(IA, A2, d0, r, t, n0, q)
    b += fdiv_one(IB, B2, d0, r, t, n0, q)
    for i in range(8):
        b.append("v_mul_f64 %s, %s, %s" % (v(ACC + 2 * i, 2), v(ACC + 2 * i, 2), v(IA if i < 4 else IB, 2)))
    # category 0 stores: factor at 8 p (PA = 32 p for that wave), reciprocal at ROFF + 8 * (pair position)
    b += ["s_cmp_lg_u32 %s, 0" % s(SCNT),
          "s_cbranch_scc1 %s" % L("wrb" + tag),
          "v_lshrrev_b32_e32 %s, 2, %s" % (v(T0), v(PA)),
          "v_lshrrev_b32_e32 %s, 2, %s" % (v(T1), v(PB)),
          "v_add_u32_e32 %s, %s, %s" % (v(RD), s(ROFF), v(SCALE)),
          "s_mov_b64 exec, %s" % s(VALA, 2),
          "global_store_dwordx2 %s, %s, %s" % (v(T0), v(A2, 2), s(SSCALEW, 2)),
          "global_store_dwordx2 %s, %s, %s" % (v(RD), v(IA, 2), s(SSCALEW, 2)),
          "s_mov_b64 exec, %s" % s(VALB, 2),
          "global_store_dwordx2 %s, %s, %s" % (v(T1), v(B2, 2), s(SSCALEW, 2)),
          "global_store_dwordx2 %s, %s, %s offset:8" % (v(RD), v(IB, 2), s(SSCALEW, 2)),
          "s_mov_b64 exec, -1",
          # the slice's product of factors (round 6): what accumulateScaleFactors adds up afterwards is ONE pair of numbers per slice and
          # pattern instead of one factor per node — the product kept as (mantissa, exponent) so that it cannot leave the range
          "v_mul_f64 %s, %s, %s" % (v(PMA, 2), v(PMA, 2), v(A2, 2)),
          "v_mul_f64 %s, %s, %s" % (v(PMB, 2), v(PMB, 2), v(B2, 2)),
          "v_frexp_exp_i32_f64_e32 %s, %s" % (v(T0), v(PMA, 2)),
          "v_frexp_exp_i32_f64_e32 %s, %s" % (v(T1), v(PMB, 2)),
          "v_frexp_mant_f64_e32 %s, %s" % (v(PMA, 2), v(PMA, 2)),
          "v_frexp_mant_f64_e32 %s, %s" % (v(PMB, 2), v(PMB, 2)),
          "v_add_u32_e32 %s, %s, %s" % (v(PEA), v(PEA), v(T0)),
          "v_add_u32_e32 %s, %s, %s" % (v(PEB), v(PEB), v(T1)),
          "s_branch %s" % L("wrb" + tag)]
    return b


def fetch(tag, slot):
    """Issue the small loads of the micro-operation described by D into pipeline slot `slot` — its matrix table (LDS-DMA into
    table buffer `slot`), the two tip-state pairs, the reciprocal scale factors: always these FOUR vector-memory instructions —,
    stash what its compute stage needs (the flags and the scale buffer a rescaling operation writes: the fields rare blocks need
    — the store address, a child in memory — are read again from the descriptor where they are used), advance the stream."""
    # (an experiment that drops a load replaces it by a cheap one to the same register so that the waits still balance)
    if "nodma" in EXPERIMENT:
        e("global_load_ubyte %s, %s, %s" % (v(T1), v(TIP), s(D + 6, 2)))
    else:
        e("s_mov_b32 m0, %s" % s(TBLS[slot]))
        e("s_mov_b64 exec, 0xfffff")
        e("global_load_lds_dwordx4 %s, %s" % (v(OM), s(STRM, 2)))
        e("s_mov_b64 exec, -1")
    e("s_bitcmp1_b32 %s, %d" % (s(DFL), B_NOLOAD1))
    e("s_cbranch_scc1 %s" % L("n1" + tag))
    e("global_load_ushort %s, %s, %s" % (v(T1S[slot]), v(TIP), s(D, 2)))
    e(L("n1" + tag) + ":")
    e("s_bitcmp1_b32 %s, %d" % (s(DFL), B_NOLOAD2))
    e("s_cbranch_scc1 %s" % L("n2" + tag))
    e("global_load_ushort %s, %s, %s" % (v(T2S[slot]), v(TIP), s(D + 2, 2)))
    e(L("n2" + tag) + ":")
    # a fused cherry (B_CHERRY): the same entry of the stream's cherry region — the cherry's two matrix tables, CHOFF bytes further on —
    # into the cherry half of the table buffer, and its two tips' state pairs (descriptor fields src2 and scale): out of line
    e("s_bitcmp1_b32 %s, %d" % (s(DFL), B_CHERRY))
    e("s_cbranch_scc1 %s" % L("ch" + tag))
    e(L("chb" + tag) + ":")
    outofline.append([L("ch" + tag) + ":",
                      "v_add_u32_e32 %s, %s, %s" % (v(T1), s(CHOFF), v(OM)),
                      "s_mov_b32 m0, %s" % s(TBLBS[slot]),
                      "s_mov_b64 exec, 0xfffff",
                      "global_load_lds_dwordx4 %s, %s" % (v(T1), s(STRM, 2)),
                      "s_mov_b64 exec, -1",
                      "global_load_ushort %s, %s, %s" % (v(TCAS[slot]), v(TIP), s(D + 2, 2)),
                      "global_load_ushort %s, %s, %s" % (v(TCBS[slot]), v(TIP), s(D + 6, 2)),
                      "s_branch %s" % L("chb" + tag)])
    e("v_add_u32_e32 %s, %s, %s" % (v(OM), s(STEP), v(OM)))       # the next table of the matrix stream (a 32-bit lane offset: < 4 GiB of stream)
    e("s_mov_b32 %s, %s" % (s(FLS[slot]), s(DFL)))
    e("s_mov_b64 %s, %s" % (s(SCALEWS[slot], 2), s(DW, 2)))
    # the reciprocal scale factors: only where the micro-operation multiplies by them (WF_INV) — out of line, so that the common
    # case is a test that falls through
    e("s_bitcmp1_b32 %s, %d" % (s(DFL), B_INV))
    e("s_cbranch_scc1 %s" % L("iv" + tag))
    e(L("ivb" + tag) + ":")
    if "noinv" in EXPERIMENT:
        outofline.append([L("iv" + tag) + ":", "global_load_ubyte %s, %s, %s" % (v(T1), v(TIP), s(D + 6, 2)), "s_branch %s" % L("ivb" + tag)])
    else:
        outofline.append([L("iv" + tag) + ":", "global_load_dwordx4 %s, %s, %s" % (v(INVS[slot], 4), v(SCALE), s(D + 6, 2)), "s_branch %s" % L("ivb" + tag)])


def first_child_fetch(tag, SFLn, off_src1):
    """The first child of the FOLLOWING micro-operation (flags in SFLn), if it waits in an LDS hold slot or in memory, into XB —
    behind the last use of XB by the micro-operation at hand; ONE test on the common path.  off_src1: where DP finds that
    descriptor's src1 now."""
    e("s_and_b32 %s, %s, 0x%x" % (s(ST), s(SFLn), (1 << B_HREAD) | (1 << B_X)))
    e("s_cbranch_scc1 %s" % L("fr" + tag))
    e(L("frb" + tag) + ":")
    blk = [L("fr" + tag) + ":",
           "s_bitcmp1_b32 %s, %d" % (s(SFLn), B_HREAD),
           "s_cbranch_scc0 %s" % L("x" + tag),
           "s_bitcmp1_b32 %s, %d" % (s(SFLn), B_HREAD2),          # slot 2 is a register set: nothing to fetch
           "s_cbranch_scc1 %s" % L("frb" + tag),
           "s_bitcmp1_b32 %s, %d" % (s(SFLn), B_HREAD1),
           "s_cselect_b32 %s, %s, 0" % (s(ST), s(HSTRIDE)),
           "v_add_u32_e32 %s, %s, %s" % (v(T0), s(ST), v(HOLD))]
    for q in range(4):
        blk.append("ds_read_b128 %s, %s offset:%d" % (v(XB + 4 * q, 4), v(T0), 1024 * q))
    blk.append("s_branch %s" % L("frb" + tag))
    blk += [L("x" + tag) + ":",
            "s_load_dwordx2 %s, %s, %d" % (s(SX, 2), s(DP, 2), off_src1),
            "s_waitcnt lgkmcnt(0)",
            "global_load_dwordx4 %s, %s, %s%s" % (v(XB, 4), v(PA), s(SX, 2), LOAD_POLICY),
            "global_load_dwordx4 %s, %s, %s offset:16%s" % (v(XB + 4, 4), v(PA), s(SX, 2), LOAD_POLICY),
            "global_load_dwordx4 %s, %s, %s%s" % (v(XB + 8, 4), v(PB), s(SX, 2), LOAD_POLICY),
            "global_load_dwordx4 %s, %s, %s offset:16%s" % (v(XB + 12, 4), v(PB), s(SX, 2), LOAD_POLICY),
            "s_branch %s" % L("frb" + tag)]
    outofline.append(blk)


def stage(tag, cur):
    """One micro-operation, k: its small operands are in pipeline slot `cur` = k mod 3, its table in LDS buffer `cur`; micro-operation
    k + 2 is fetched into slot cur + 2, the first child of k + 1 — when it has to be — into XB.  DP points at descriptor k + 3
    until the stage's descriptor load, at k + 4 behind it (kernels.h WalkOp: src1 at 0, src2 at 8, store at 16)."""
    nxt, far = (cur + 1) % 3, (cur + 2) % 3
    Tt1, Tt2, INV, SFL, SSCALEW = T1S[cur], T2S[cur], INVS[cur], FLS[cur], SCALEWS[cur]
    tblCur, spCur, tbvCur = TBLS[cur], SPS[cur], TBVS[cur]
    if "notopwait" not in EXPERIMENT:
        e("s_waitcnt lgkmcnt(0)")                   # descriptor k + 2 (and LDS writes) have landed
    fetch(tag, far)
    # wait for this micro-operation's loads: N = what was issued after them and may stay outstanding — the fetches of k + 1 and
    # k + 2 (8), a first child of k - 1 from memory (+4), stores where the engine counts them (engine_walk.cpp runPlan);
    # 4 when this micro-operation's own first child comes from memory (requested a stage ago, behind the fetch of k + 1)
    e("s_and_b32 %s, %s, 0x%x" % (s(ST), s(SFL), 15 << B_WAIT0))
    e("s_cbranch_scc1 %s" % L("ws" + tag))
    novm = "novmwait" in EXPERIMENT
    e("s_nop 0" if novm else "s_waitcnt vmcnt(%d)" % WAIT_N[0])
    e(L("wd" + tag) + ":")
    # the other codes, out of line: code 1 (one of the two following micro-operations pays: the neighbours of every payment in a
    # folded program) and code 2 (both fetch four: programs that pay at every node — partial updates, write-mode lists) by a test
    # each, the rest through a jump table of (wait, branch) pairs
    blk = [L("ws" + tag) + ":",
           "s_bfe_u32 %s, %s, 0x%x" % (s(ST), s(SFL), (4 << 16) | B_WAIT0),
           "s_cmp_eq_u32 %s, 1" % s(ST),
           "s_cbranch_scc0 %s" % L("wu" + tag),
           "s_nop 0" if novm else "s_waitcnt vmcnt(%d)" % WAIT_N[1],
           "s_branch %s" % L("wd" + tag),
           L("wu" + tag) + ":",
           "s_cmp_eq_u32 %s, 2" % s(ST),
           "s_cbranch_scc0 %s" % L("wt" + tag),
           "s_nop 0" if novm else "s_waitcnt vmcnt(%d)" % WAIT_N[2],
           "s_branch %s" % L("wd" + tag),
           L("wt" + tag) + ":",
           "s_lshl_b32 %s, %s, 3" % (s(ST), s(ST)),
           "s_getpc_b64 %s" % s(SX, 2),                       # (SX: the scratch pair of the first-child block, free here)
           "s_add_u32 %s, %s, %s" % (s(SX), s(SX), s(ST)),
           "s_addc_u32 %s, %s, 0" % (s(SX + 1), s(SX + 1)),
           "s_add_u32 %s, %s, 12" % (s(SX), s(SX)),          # table entry `code` starts 20 + 8 code bytes behind s_getpc's successor ...
           "s_addc_u32 %s, %s, 0" % (s(SX + 1), s(SX + 1)),
           "s_setpc_b64 %s" % s(SX, 2)]
    # (... five 4-byte instructions follow s_getpc_b64 up to and including s_setpc_b64: entry 0 would sit at + 20; entries are 8 bytes
    # and entry 0 is never taken, so the table proper starts at entry 1 = + 20 + 8 - 8: the constant above is 20 - 8 = 12)
    for code in range(1, 16):
        blk += ["s_nop 0" if novm else "s_waitcnt vmcnt(%d)" % WAIT_N[code], "s_branch %s" % L("wd" + tag)]
    outofline.append(blk)
    off_src2 = -3 * 64 + 8
    off_src1_next = -2 * 64
    off_store = -4 * 64 + 16
    m2blk = [L("m2" + tag) + ":",                    # both children in memory: the second one is loaded into ACC, synchronously
             "s_load_dwordx2 %s, %s, %d" % (s(SSRC2, 2), s(DP, 2), off_src2),
             "s_waitcnt lgkmcnt(0)",
             "global_load_dwordx4 %s, %s, %s%s" % (v(ACC, 4), v(PA), s(SSRC2, 2), LOAD_POLICY),
             "global_load_dwordx4 %s, %s, %s offset:16%s" % (v(ACC + 4, 4), v(PA), s(SSRC2, 2), LOAD_POLICY),
             "global_load_dwordx4 %s, %s, %s%s" % (v(ACC + 8, 4), v(PB), s(SSRC2, 2), LOAD_POLICY),
             "global_load_dwordx4 %s, %s, %s offset:16%s" % (v(ACC + 12, 4), v(PB), s(SSRC2, 2), LOAD_POLICY),
             "s_waitcnt vmcnt(0)",
             "s_branch %s" % L("m2b" + tag)]
    # first child
    e("s_bitcmp1_b32 %s, %d" % (s(SFL), B_T1))
    e("s_cbranch_scc0 %s" % L("fm" + tag))
    tip_columns(F, Tt1, tblCur, 0)
    e("s_branch %s" % L("g" + tag))
    e(L("fm" + tag) + ":")
    e("ds_read_b64 %s, %s" % (v(SP, 2), v(spCur)))
    col0_reads(G, tbvCur, 0)                     # (G is free until the second child's contribution is formed)
    e("s_bitcmp1_b32 %s, %d" % (s(SFL), B_HREAD2))
    e("s_cbranch_scc1 %s" % L("fh2" + tag))
    ldsw = "s_nop 0" if "noldswait" in EXPERIMENT else "s_waitcnt lgkmcnt(0)"
    e(ldsw)
    matvec(F, XB, SP, None, col0v=G)
    save = lines[:]
    del lines[:]
    e(L("fh2" + tag) + ":")                      # first child waits in the register hold slot
    e(ldsw)
    matvec(F, H2, SP, None, col0v=G)
    e("s_branch %s" % L("g" + tag))
    outofline.append(lines[:])
    del lines[:]
    lines.extend(save)
    # second child
    e(L("g" + tag) + ":")
    e("s_bitcmp1_b32 %s, %d" % (s(SFL), B_T2))
    e("s_cbranch_scc0 %s" % L("ga" + tag))
    tip_columns(G, Tt2, tblCur, 160)
    e(ldsw)
    e("s_branch %s" % L("mul" + tag))
    e(L("ga" + tag) + ":")
    e("s_and_b32 %s, %s, 0x%x" % (s(ST), s(SFL), (1 << B_MEM2) | (1 << B_CHERRY)))       # the two rare forms of a second child: one test
    e("s_cbranch_scc1 %s" % L("m2" + tag))
    e(L("m2b" + tag) + ":")
    # a fused cherry: ACC = column(tip A) * column(tip B) from the cherry half of this stage's table buffer (G and the first-child
    # buffer are free here; what ACC held is dead: the micro-operation before a cherry's consumer parked or stored its result)
    tblB = TBLBS[cur]
    save2 = lines[:]
    del lines[:]
    e(L("cc" + tag) + ":")
    tip_columns(G, TCAS[cur], tblB, 0)
    tip_columns(XB, TCBS[cur], tblB, 160)
    e("s_waitcnt lgkmcnt(0)")
    for i in range(8):
        e("v_mul_f64 %s, %s, %s" % (v(ACC + 2 * i, 2), v(G + 2 * i, 2), v(XB + 2 * i, 2)))
    e("s_branch %s" % L("m2b" + tag))
    ccblk = lines[:]
    del lines[:]
    lines.extend(save2)
    m2blk[1:1] = ["s_bitcmp1_b32 %s, %d" % (s(SFL), B_CHERRY), "s_cbranch_scc1 %s" % L("cc" + tag)]
    outofline.append(m2blk)
    outofline.append(ccblk)
    e("ds_read_b64 %s, %s offset:160" % (v(SP, 2), v(spCur)))
    col0_reads(XB, tbvCur, 160)                  # (the first-child buffer: consumed by the first mat-vec, or not in use)
    e(ldsw)
    matvec(G, ACC, SP, None, col0v=XB)
    e(L("mul" + tag) + ":")
    # the first child of micro-operation k + 1, where it has to be fetched (XB is free from here on)
    first_child_fetch(tag, FLS[nxt], off_src1_next)
    # descriptor k + 3: behind every LDS wait of the stage (scalar loads share the counter with LDS and return out of order)
    e("s_load_dwordx8 %s, %s, 0x0" % (s(D, 8), s(DP, 2)))
    e("s_load_dwordx4 %s, %s, 0x20" % (s(DFL, 4), s(DP, 2)))
    e("s_add_u32 %s, %s, 64" % (s(DP), s(DP)))
    e("s_addc_u32 %s, %s, 0" % (s(DP + 1), s(DP + 1)))
    for i in range(8):
        e("v_mul_f64 %s, %s, %s" % (v(ACC + 2 * i, 2), v(F + 2 * i, 2), v(G + 2 * i, 2)))
    pad_block()
    e("s_and_b32 %s, %s, 0x%x" % (s(ST), s(SFL), (1 << B_WRITE) | (1 << B_STORE) | (1 << B_READ)))
    e("s_cbranch_scc1 %s" % L("tl" + tag))
    e(L("tlb" + tag) + ":")
    # the rare tail — a result that pays scale factors (read mode: one multiplication by the reciprocals fetched with it), write-mode
    # rescaling, a result that is stored — out of line: one test each with its block, then back; a result that is parked in a hold
    # slot is every third micro-operation's case and keeps its own test (behind the tail: what is parked is the scaled value)
    mulblk = [L("ml" + tag) + ":"]
    for i in range(8):
        mulblk.append("v_mul_f64 %s, %s, %s" % (v(ACC + 2 * i, 2), v(ACC + 2 * i, 2), v(INV + (0 if i < 4 else 2), 2)))
    mulblk.append("s_branch %s" % L("mlb" + tag))
    outofline.append(mulblk)
    outofline.append([L("tl" + tag) + ":",
                      # (WF_INV alone — a fetch padded to four loads behind write-mode rescaling, engine_walk.cpp runPlan — multiplies nothing)
                      "s_bitcmp1_b32 %s, %d" % (s(SFL), B_READ), "s_cbranch_scc1 %s" % L("ml" + tag), L("mlb" + tag) + ":",
                      "s_bitcmp1_b32 %s, %d" % (s(SFL), B_WRITE), "s_cbranch_scc1 %s" % L("wr" + tag), L("wrb" + tag) + ":",
                      "s_bitcmp1_b32 %s, %d" % (s(SFL), B_STORE), "s_cbranch_scc1 %s" % L("st" + tag), L("stb" + tag) + ":",
                      "s_branch %s" % L("tlb" + tag)])
    outofline.append(rescale_block(tag, SSCALEW))
    # The result leaves in FOUR FULLY CONTIGUOUS 1 KiB stores (tools/hbm_write_probe.hip: half-line non-temporal stores
    # sustain 2.0 TB/s, full lines 4.9-5.2): lane 2 q + r owns patterns q + 32 r and 64 + q + 32 r of the workgroup's 128,
    # store instruction j covers patterns 32 j .. 32 j + 31, lane 2 q + r writing half r (16 bytes) of pattern 32 j + q —
    # its own data or its neighbour's, exchanged with v_cndmask_b32_dpp quad_perm:[1,0,3,2] (no LDS).
    blk = [L("st" + tag) + ":", "s_load_dwordx2 %s, %s, %d" % (s(SSTORE, 2), s(DP, 2), off_store),
           "s_mov_b32 vcc_lo, 0x55555555", "s_mov_b32 vcc_hi, 0x55555555", "s_nop 1"]
    for d in range(4):      # j = 0 (pattern q, owner r = 0): even lanes own half 0; odd lanes take the neighbour's half 1
        blk.append("v_cndmask_b32_dpp %s, %s, %s, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (v(F + d), v(ACC + 4 + d), v(ACC + d)))
    for d in range(4):      # j = 2 (pattern 64 + q): the same on the second pattern of the lanes
        blk.append("v_cndmask_b32_dpp %s, %s, %s, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (v(F + 8 + d), v(ACC + 12 + d), v(ACC + 8 + d)))
    blk += ["s_mov_b32 vcc_lo, 0xaaaaaaaa", "s_mov_b32 vcc_hi, 0xaaaaaaaa", "s_nop 1"]
    for d in range(4):      # j = 1 (pattern 32 + q, owner r = 1): odd lanes own half 1; even lanes take the neighbour's half 0
        blk.append("v_cndmask_b32_dpp %s, %s, %s, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (v(F + 4 + d), v(ACC + d), v(ACC + 4 + d)))
    for d in range(4):      # j = 3 (pattern 96 + q)
        blk.append("v_cndmask_b32_dpp %s, %s, %s, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (v(F + 12 + d), v(ACC + 8 + d), v(ACC + 12 + d)))
    blk.append("s_waitcnt lgkmcnt(0)")
    for j in range(4):
        blk.append("s_mov_b64 exec, %s" % s(MASK + 2 * j, 2))
        blk.append("global_store_dwordx4 %s, %s, %s offset:%d%s" % (v(VST), v(F + 4 * j, 4), s(SSTORE, 2), 1024 * j, STORE_POLICY))
    blk += ["s_mov_b64 exec, -1", "s_nop 0", "s_branch %s" % L("stb" + tag)]
    outofline.append(blk)
    e("s_bitcmp1_b32 %s, %d" % (s(SFL), B_HWRITE))
    e("s_cbranch_scc1 %s" % L("hw" + tag))
    e(L("hwb" + tag) + ":")
    blk = [L("hw" + tag) + ":",
           "s_bfe_u32 %s, %s, 0x2000b" % (s(ST), s(SFL)),           # hold = 1 + slot (2 bits at 11)
           "s_cmp_eq_u32 %s, 3" % s(ST),
           "s_cbranch_scc1 %s" % L("hw2" + tag),
           "s_add_i32 %s, %s, -1" % (s(ST), s(ST)),
           "s_mul_i32 %s, %s, %s" % (s(ST), s(ST), s(HSTRIDE)),
           "v_add_u32_e32 %s, %s, %s" % (v(T0), s(ST), v(HOLD))]
    for q in range(4):
        blk.append("ds_write_b128 %s, %s offset:%d" % (v(T0), v(ACC + 4 * q, 4), 1024 * q))
    blk.append("s_branch %s" % L("hwb" + tag))
    blk.append(L("hw2" + tag) + ":")
    for i in range(8):
        blk.append("v_mov_b64 %s, %s" % (v(H2 + 2 * i, 2), v(ACC + 2 * i, 2)))
    blk.append("s_branch %s" % L("hwb" + tag))
    outofline.append(blk)


def build():
    # ---- inputs -> fixed registers
    e("s_mov_b64 %s, %%[dp]" % s(DP, 2))
    e("s_mov_b64 %s, %%[strm]" % s(STRM, 2))
    e("s_add_i32 %s, %%[cnt], -1" % s(CNT))
    e("s_mov_b32 %s, %%[tbl]" % s(TBLS[0]))
    e("s_add_u32 %s, %%[tbl], %%[tblStep]" % s(TBLS[1]))
    e("s_add_u32 %s, %s, %%[tblStep]" % (s(TBLS[2]), s(TBLS[1])))
    e("s_mov_b32 %s, %%[choff]" % s(CHOFF))
    e("s_mul_i32 %s, %%[tblStep], 3" % s(ST))                           # the cherry halves follow the three table buffers of all waves
    for j in range(3):
        e("s_add_u32 %s, %s, %s" % (s(TBLBS[j]), s(TBLS[j]), s(ST)))
    e("s_mov_b32 %s, %%[holdStride]" % s(HSTRIDE))
    e("s_mov_b32 %s, %%[strmStep]" % s(STEP))
    e("s_add_i32 %s, %%[pEnd], -1" % s(LAST))
    # ---- lane offsets
    e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(LANE))
    e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(LANE), v(LANE)))
    # lane 2 q + r owns patterns p0 + q + 32 r and p0 + 64 + q + 32 r (see the store block)
    e("v_lshrrev_b32_e32 %s, 1, %s" % (v(T1), v(LANE)))                  # q
    e("v_add_u32_e32 %s, %%[p0], %s" % (v(T1), v(T1)))                   # p0 + q: first pattern of store instruction 0's piece
    for j in range(4):
        if j:
            e("v_add_u32_e32 %s, 32, %s" % (v(T1), v(T1)))
        e("v_cmp_gt_i32_e32 vcc, %%[pEnd], %s" % v(T1))
        e("s_nop 3")
        e("s_mov_b64 %s, vcc" % s(MASK + 2 * j, 2))
    e("s_mov_b32 %s, %%[cM]" % s(CM0))
    e("s_add_u32 %s, %%[cM], 160" % s(CM160))
    e("s_mov_b32 %s, %%[exch]" % s(EXCH))
    e("s_mov_b32 %s, %%[ncat]" % s(NCAT))
    e("s_mov_b32 %s, %%[roff]" % s(ROFF))
    e("s_mov_b32 %s, %%[cat]" % s(SCNT))
    # the three maximum buffers of write-mode rescaling (rescale_block) start at zero: every wave clears all of them (3 KiB)
    e("s_mov_b32 %s, 0" % s(RB))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (v(T0), v(LANE)))
    e("v_add_u32_e32 %s, %%[exch], %s" % (v(T0), v(T0)))
    e("v_mov_b64 %s, 0" % v(F, 2))
    e("v_mov_b64 %s, 0" % v(F + 2, 2))
    for k in range(3):
        e("ds_write_b128 %s, %s offset:%d" % (v(T0), v(F, 4), 1024 * k))
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e("v_lshlrev_b32_e32 %s, 4, %s" % (v(VST), v(LANE)))                 # store instruction j: 1 KiB j + 16 lane from the group's first pattern
    e("s_lshl_b32 %s, %%[p0], 5" % s(ST))
    e("s_add_u32 %s, %s, %%[cP32]" % (s(ST), s(ST)))
    e("v_add_u32_e32 %s, %s, %s" % (v(VST), s(ST), v(VST)))
    e("v_and_b32_e32 %s, 1, %s" % (v(T0), v(LANE)))                      # r
    e("v_lshlrev_b32_e32 %s, 5, %s" % (v(T0), v(T0)))
    e("v_lshrrev_b32_e32 %s, 1, %s" % (v(T1), v(LANE)))
    e("v_add3_u32 %s, %s, %s, %%[p0]" % (v(T0), v(T0), v(T1)))           # first pattern of the lane
    e("v_add_u32_e32 %s, 64, %s" % (v(T1), v(T0)))                      # second
    e("v_cmp_gt_i32_e32 vcc, %%[pEnd], %s" % v(T0))
    e("s_nop 3")
    e("s_mov_b64 %s, vcc" % s(VALA, 2))
    e("v_cmp_gt_i32_e32 vcc, %%[pEnd], %s" % v(T1))
    e("s_nop 3")
    e("s_mov_b64 %s, vcc" % s(VALB, 2))
    e("v_min_i32_e32 %s, %s, %s" % (v(T0), s(LAST), v(T0)))             # lanes past the end recompute the last pattern
    e("v_min_i32_e32 %s, %s, %s" % (v(T1), s(LAST), v(T1)))
    e("v_lshlrev_b32_e32 %s, 5, %s" % (v(PA), v(T0)))
    e("v_lshlrev_b32_e32 %s, 5, %s" % (v(PB), v(T1)))
    e("v_add_u32_e32 %s, %%[cP32], %s" % (v(PA), v(PA)))
    e("v_add_u32_e32 %s, %%[cP32], %s" % (v(PB), v(PB)))
    e("v_lshlrev_b32_e32 %s, 1, %s" % (v(T0), v(LANE)))
    e("v_add_u32_e32 %s, %%[t0], %s" % (v(TIP), v(T0)))                  # position of the pair in the interleaved layouts (t0: kernels.h WalkSeg)
    e("v_lshlrev_b32_e32 %s, 3, %s" % (v(SCALE), v(TIP)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (v(OM), v(LANE)))
    e("v_add_u32_e32 %s, %%[cM], %s" % (v(OM), v(OM)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (v(HOLD), v(LANE)))
    e("v_add_u32_e32 %s, %%[hold], %s" % (v(HOLD), v(HOLD)))
    e("v_and_b32_e32 %s, 3, %s" % (v(T0), v(LANE)))                      # matrix entry 4 i + k of lane l & 15 is T[k][i]
    e("v_lshlrev_b32_e32 %s, 5, %s" % (v(T0), v(T0)))
    e("v_bfe_u32 %s, %s, 2, 2" % (v(T1), v(LANE)))
    e("v_lshl_add_u32 %s, %s, 3, %s" % (v(T0), v(T1), v(T0)))
    for j in range(3):
        e("v_add_u32_e32 %s, %s, %s" % (v(SPS[j]), s(TBLS[j]), v(T0)))
        e("v_mov_b32_e32 %s, %s" % (v(TBVS[j]), s(TBLS[j])))
    for i in range(8):
        e("v_mov_b64 %s, 1.0" % v(ACC + 2 * i, 2))
    e("v_mov_b64 %s, 1.0" % v(PMA, 2))                                  # the slice's product of written factors: 1 = 0.5 x 2^1
    e("v_mov_b64 %s, 1.0" % v(PMB, 2))
    e("v_mov_b32_e32 %s, 0" % v(PEA))
    e("v_mov_b32_e32 %s, 0" % v(PEB))
    # ---- prologue: fetch micro-operations 0 and 1 into slots 0 and 1, the first child of 0 if it is in memory (no hold slot is
    # in use at the start of a program), descriptor 2 into D; DP -> descriptor 3
    e("s_load_dwordx8 %s, %s, 0x0" % (s(D, 8), s(DP, 2)))
    e("s_load_dwordx4 %s, %s, 0x20" % (s(DFL, 4), s(DP, 2)))
    e("s_waitcnt lgkmcnt(0)")
    fetch("p", 0)
    e("s_load_dwordx8 %s, %s, 0x40" % (s(D, 8), s(DP, 2)))
    e("s_load_dwordx4 %s, %s, 0x60" % (s(DFL, 4), s(DP, 2)))
    e("s_waitcnt lgkmcnt(0)")
    fetch("q", 1)
    first_child_fetch("p", FLS[0], 0)
    e("s_load_dwordx8 %s, %s, 0x80" % (s(D, 8), s(DP, 2)))
    e("s_load_dwordx4 %s, %s, 0xa0" % (s(DFL, 4), s(DP, 2)))
    e("s_add_u32 %s, %s, 0xc0" % (s(DP), s(DP)))
    e("s_addc_u32 %s, %s, 0" % (s(DP + 1), s(DP + 1)))
    # ---- the loop: three stages
    # (CNT = micro-operations left behind the one at hand: the loop leaves behind ANY stage, so a program needs no padding)
    e(L("top") + ":")
    stage("a", 0)
    e("s_sub_u32 %s, %s, 1" % (s(CNT), s(CNT)))
    e("s_cbranch_scc1 %s" % L("end"))
    stage("b", 1)
    e("s_sub_u32 %s, %s, 1" % (s(CNT), s(CNT)))
    e("s_cbranch_scc1 %s" % L("end"))
    stage("c", 2)
    e("s_sub_u32 %s, %s, 1" % (s(CNT), s(CNT)))
    e("s_cbranch_scc0 %s" % L("top"))
    e("s_branch %s" % L("end"))
    for blk in outofline:
        for l in blk:
            e(l)
    e(L("end") + ":")
    # the last result also goes to LDS hold slot 0 (nothing is parked at the end of a program): the root slice's epilogue finishes
    # the evaluation from there (kernels_walk4.hip, root_site4.h) instead of a launch that reads the root's partials back
    for q in range(4):
        e("ds_write_b128 %s, %s offset:%d" % (v(HOLD), v(ACC + 4 * q, 4), 1024 * q))
    # A slice that rescaled in write mode leaves the product of its factors behind: the third no-op behind the program (DP points one
    # descriptor past it: every stage has advanced it by one) carries the two vectors that receive it — mantissas (double, the lane's
    # pair at its position in the pair-interleaved layout) in its scaleW field, exponents (int) in its store field; null: nothing to leave
    e("s_load_dwordx2 %s, %s, %d" % (s(SSRC2, 2), s(DP, 2), -64 + 40))
    e("s_load_dwordx2 %s, %s, %d" % (s(SSTORE, 2), s(DP, 2), -64 + 16))
    e("s_waitcnt lgkmcnt(0)")
    e("s_cmp_eq_u64 %s, 0" % s(SSRC2, 2))
    e("s_cbranch_scc1 %s" % L("done"))
    e("s_cmp_lg_u32 %s, 0" % s(SCNT))
    e("s_cbranch_scc1 %s" % L("done"))
    e("v_lshrrev_b32_e32 %s, 1, %s" % (v(T0), v(SCALE)))
    e("global_store_dwordx4 %s, %s, %s" % (v(SCALE), v(PMA, 4), s(SSRC2, 2)))
    e("global_store_dwordx2 %s, %s, %s" % (v(T0), v(PEA, 2), s(SSTORE, 2)))
    e(L("done") + ":")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")


def main():
    build()
    text = ["// GENERATED by tools/gen_walk4_fast.py — do not edit; see that file for the register map and the design.",
            "#define WALK4_FAST_ASM \\"]
    for l in lines:
        sep = "\\n" if l.endswith(":") else "\\n\\t"
        text.append('    "%s%s" \\' % (l, sep))
    text.append('    ""')
    clob = ['"v%d"' % i for i in range(NV)] + ['"s%d"' % i for i in range(S_FIRST, S_LAST + 1) if i not in (32, 33, 34, 35)]
    clob += ['"vcc"', '"scc"', '"memory"']
    text.append("#define WALK4_FAST_CLOBBERS " + ", ".join(clob))
    text.append("#define WALK4_FAST_VGPRS %d" % NV)
    body = "\n".join(text) + "\n"
    if os.environ.get("WALK4_CHECK_ONLY"):
        raise SystemExit(0 if os.path.exists(OUT) and open(OUT).read() == body else 1)
    open(OUT, "w").write(body)
    print("wrote %s: %d instructions" % (OUT, sum(1 for l in lines if not l.endswith(":"))))


if __name__ == "__main__":
    main()
