"""Secondary measurement: full branch-length gradient evaluations per second.

One evaluation = post-order pass + root lnL + pre-order pass (2T-2 ops) + edge derivatives for all 2T-2 branches, driven
exactly as beast-mcmc_amd/gradient.py mirrors the reference's gradient delegates, buffers alternating between two sets as
BufferIndexHelper makes them.  Not the headline metric (bench.py).  A/B switches (read at instance creation):
BEAGLE_MI355_NO_PRE_WALK=1 (pre-order partials always written, one sweep per tree level), BEAGLE_MI355_NO_FUSED_GRADIENT=1
(operation by operation).

    python tools/gradient_bench.py --config A --patterns 20000 --steps 5
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="A", choices=["A", "B", "C"])
    ap.add_argument("--patterns", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3,
                    help="untimed gradients first: the first two of an instance allocate both buffer sets (hundreds of hipMalloc calls: "
                         "64 instead of 6.3 ms per gradient at 1e5 patterns when they fall into the timed steps)")
    ap.add_argument("--cache", default="/tmp/beagle_mi355_cache")
    ap.add_argument("--rescale", action="store_true", help="the post-order pass rescales every node in write mode, every evaluation")
    args = ap.parse_args()
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.gradient import BranchGradient
    from beast_mcmc_amd.inputs import synth
    maker = {"A": synth.config_a, "B": synth.config_b, "C": synth.config_c}[args.config]
    os.makedirs(args.cache, exist_ok=True)
    wl = synth.cached(os.path.join(args.cache, "config_%s.pkl" % args.config), maker)
    if args.patterns:
        wl = wl.shard(0, min(args.patterns, wl.pattern_count))
    g = BranchGradient(wl, double_buffer=True, rescale=args.rescale)       # the reference's buffer plan: two sets, alternating (BufferIndexHelper)
    for _ in range(args.warmup):
        lnl, grad = g.gradient()
    g.b.synchronize()
    if wl.state_count == 4:
        g.b.kernelTimerRestart()            # (resets the walk's counters — without touching the held list: stored nodes per post-order pass below)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnl, grad = g.gradient()
    g.b.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    stored_per_pass = g.b.walkStats()["stored"] / args.steps if wl.state_count == 4 else None
    # the reference's buffer plan flips between two sets, so over a chain of gradients no held pre-order list ever has to run
    # after all; (the plain likelihood evaluations below re-use the sets and do make the last one run)
    assert g.b.gradientStats()["late"] == 0, g.b.gradientStats()
    for _ in range(20):                    # (let the engine's "a gradient chain wants every node stored" hint run out: the plain likelihood)
        g.log_likelihood()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.log_likelihood()
    dl = (time.perf_counter() - t0) / args.steps
    buf = wl.pattern_count * wl.state_count * wl.category_count * 8
    internal = wl.tree.node_count - wl.tip_count
    how = g.b.gradientStats()
    # what a gradient has to move on top of a likelihood when the sums are answered from the held list (how["walked"]): every
    # internal node's post-order partial written once by the post-order pass and read once by the pre-order walk; otherwise
    # (pre-order partials written) also pre(parent) read and pre(child) written per operation
    walked = how["walked"] >= args.steps
    extra_bytes = 2 * internal * buf if walked else 2 * internal * buf + (internal + 2 * internal) * buf
    # roofline of the gradient evaluation as THIS engine runs it (bytes the design moves / wall time; 8 TB/s HBM):
    #   4 states, sums answered from the held list: likelihood-side bytes (every internal node stored: written once) + every
    #   internal post-order partial read once by the pre-order walk;
    #   16..64 states: a pre-order operation reads pre(parent) and the sibling's post-order partial (nothing for a tip sibling: compact
    #   states) and writes pre(child) (kernels_mfma.hip k_preOpTiled); an edge's derivative reads pre and post of the node below it
    #   once (post: nothing for a tip) (k_edgeTiled); the post-order pass stores every node and reads every internal child.
    #   (BEAGLE_MI355_PRE_TWO_PASS=1 / BEAGLE_MI355_EDGE_TWO_STEP=1, rounds 2-3: 5 and 4-6 transfers instead.)
    nodes = wl.tree.node_count
    int_children = sum(1 for n in range(wl.tip_count, nodes) for ch in (int(wl.tree.left[n]), int(wl.tree.right[n])) if ch >= wl.tip_count)
    if wl.state_count == 4:
        # (round 5: a gradient chain's post-order passes leave tip-tip nodes, and those under one more tip, unstored — the pre-order
        # walk re-evaluates them from the tips: `stored_per_pass` nodes are written once and read once)
        moved = int(2 * stored_per_pass * buf) if walked else (internal + int_children) * buf + 5 * (nodes - 1) * buf // 2
    else:
        two_pass = os.environ.get("BEAGLE_MI355_PRE_TWO_PASS", "0") not in ("", "0")
        two_step = os.environ.get("BEAGLE_MI355_EDGE_TWO_STEP", "0") == "1"
        tips = wl.tip_count
        pre_ops = 5 * (nodes - 1) if two_pass else 2 * (nodes - 1) + int_children          # (a node's sibling is internal: one more read)
        edges = (tips + 6 * (nodes - 1 - tips)) if two_step else (nodes - 1) + (nodes - 1 - tips)
        moved = (internal + int_children) * buf + pre_ops * buf + edges * buf
    roofline = {"bound": "hbm", "bytes_moved_by_design": int(moved), "achieved": round(moved / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(moved / dt / 1e9 / 8000.0, 4),
                "note": "bytes this implementation's passes move per gradient / wall time per gradient (host included)"}
    print(json.dumps({"metric": "branch-gradient evals/sec (secondary)", "roofline": roofline, "value": round(1.0 / dt, 3), "ms_per_gradient": round(dt * 1e3, 2),
                      "ms_per_likelihood_same_driver": round(dl * 1e3, 2),
                      "rescale": bool(args.rescale), "workload": "%s: %d taxa x %d patterns, %d states, %d categories" % (wl.name, wl.tip_count, wl.pattern_count,
                                                                                          wl.state_count, wl.category_count),
                      "gradient_over_likelihood": round(dt / dl, 2),
                      "post_order_nodes_stored_per_gradient": stored_per_pass, "internal_nodes": internal,
                      "extra_algorithmic_GB": round(extra_bytes / 1e9, 2), "extra_GBs": round(extra_bytes / max(dt - dl, 1e-9) / 1e9, 1),
                      "lnL": lnl, "grad_norm": float((grad ** 2).sum() ** 0.5), "how": how,
                      "hbm_bytes_resident": int(g.b.deviceBytes())}))
    g.close()


if __name__ == "__main__":
    main()
