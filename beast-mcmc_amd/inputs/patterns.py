"""Alignment -> unique site patterns + weights, and the contiguous-block pattern sharding.

Mirrors the input semantics the engine relies on:

* nucleotide state codes     src/dr/evolution/datatype/Nucleotides.java:51-96 (A,C,G,T/U = 0..3; every other
                             symbol gets a code >= 4, which the likelihood treats as "unknown": all-ones tip partial)
* unique-pattern compression src/dr/evolution/alignment/SitePatterns.java:226-335 (first occurrence keeps the
                             column, later identical columns add to its weight)
* empirical frequencies      src/dr/evolution/alignment/PatternList.java:138-214 — for data whose only
                             non-ACGT symbol is the gap, the estimator's fixed point equals plain counting
* shard block sizes          src/dr/evolution/alignment/Patterns.java:142-167 (div = P / N, the first P % N
                             shards get one more) — the multi-GPU partition (DESIGN.md row e)
"""
import numpy as np

_NUC = np.full(128, 17, dtype=np.int32)
_NUC[ord("?")] = 16
for _ch, _code in (("A", 0), ("B", 11), ("C", 1), ("D", 12), ("G", 2), ("H", 13), ("K", 10), ("M", 7),
                   ("N", 15), ("R", 5), ("S", 9), ("T", 3), ("U", 3), ("V", 14), ("W", 8), ("Y", 6)):
    _NUC[ord(_ch)] = _code
    _NUC[ord(_ch.lower())] = _code
for _ch in "EFIJLOPQXZ":
    _NUC[ord(_ch)] = 16
    _NUC[ord(_ch.lower())] = 16


def nucleotide_states(seq):
    """Sequence string -> int32 state codes (Nucleotides.NUCLEOTIDE_STATES)."""
    b = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    return _NUC[b]


def site_patterns(state_rows, unique=True):
    """``state_rows``: int array [taxa][sites].  Returns (patterns [taxa][P], weights [P])."""
    a = np.asarray(state_rows, dtype=np.int32)
    if not unique:
        return a.copy(), np.ones(a.shape[1])
    seen = {}
    order = []
    weights = []
    for s in range(a.shape[1]):
        key = a[:, s].tobytes()
        idx = seen.get(key)
        if idx is None:
            seen[key] = len(order)
            order.append(s)
            weights.append(1.0)
        else:
            weights[idx] += 1.0
    return np.ascontiguousarray(a[:, order]), np.asarray(weights, dtype=np.float64)


def empirical_frequencies(state_rows, state_count=4):
    a = np.asarray(state_rows)
    counts = np.array([(a == s).sum() for s in range(state_count)], dtype=np.float64)
    return counts / counts.sum()


def shard_bounds(pattern_count, shard_count):
    """[(start, stop)] per shard, block sizes as Patterns.subSetPatterns."""
    div, rem = divmod(pattern_count, shard_count)
    out = []
    start = 0
    for i in range(shard_count):
        n = div + (1 if i < rem else 0)
        out.append((start, start + n))
        start += n
    return out
