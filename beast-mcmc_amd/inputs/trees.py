"""Rooted binary time-trees as flat arrays, numbered the way the reference's callers number nodes:
tips 0..T-1 (tip i <-> taxon i), internal nodes T..2T-2
(src/dr/evomodel/treelikelihood/BeagleTreeLikelihood.java:484-487, :1270).  The root is whichever
internal node has no parent — not necessarily 2T-2.
"""
import numpy as np


class Tree:
    def __init__(self, left, right, height, root):
        self.left = np.asarray(left, dtype=np.int32)
        self.right = np.asarray(right, dtype=np.int32)
        self.height = np.asarray(height, dtype=np.float64)
        self.root = int(root)
        self.node_count = len(self.left)
        self.tip_count = (self.node_count + 1) // 2
        self.parent = np.full(self.node_count, -1, dtype=np.int32)
        for n in range(self.tip_count, self.node_count):
            self.parent[self.left[n]] = n
            self.parent[self.right[n]] = n

    def branch_length(self, node, rate=1.0):
        return rate * (self.height[self.parent[node]] - self.height[node])

    def depth(self):
        """Number of internal nodes on the longest root-to-tip path (= number of dependency levels)."""
        lvl = np.zeros(self.node_count, dtype=np.int64)
        for n in self.postorder():
            if n >= self.tip_count:
                lvl[n] = 1 + max(lvl[self.left[n]], lvl[self.right[n]])
        return int(lvl[self.root])

    def postorder(self):
        out = []
        stack = [(self.root, False)]
        while stack:
            n, seen = stack.pop()
            if n < self.tip_count:
                out.append(n)
            elif seen:
                out.append(n)
            else:
                stack.append((n, True))
                stack.append((int(self.right[n]), False))
                stack.append((int(self.left[n]), False))
        return out


def from_nested(nested, tip_count):
    """``nested`` = tip index, or (left, right, height).  Internal nodes are numbered in post-order."""
    left = [-1] * (2 * tip_count - 1)
    right = [-1] * (2 * tip_count - 1)
    height = [0.0] * (2 * tip_count - 1)
    counter = [tip_count]

    def rec(x):
        if isinstance(x, int):
            return x
        a = rec(x[0])
        b = rec(x[1])
        n = counter[0]
        counter[0] += 1
        left[n], right[n], height[n] = a, b, float(x[2])
        return n

    root = rec(nested)
    assert counter[0] == 2 * tip_count - 1
    return Tree(left, right, height, root)


def coalescent_tree(tip_count, rng, root_height=None):
    """Kingman-coalescent genealogy, contemporaneous tips.  Internal node T+i is the i-th
    coalescence, so the root is 2T-2.  Heights are rescaled to ``root_height`` if given."""
    n = tip_count
    left = [-1] * (2 * n - 1)
    right = [-1] * (2 * n - 1)
    height = [0.0] * (2 * n - 1)
    active = list(range(n))
    t = 0.0
    nxt = n
    while len(active) > 1:
        k = len(active)
        t += rng.exponential(2.0 / (k * (k - 1)))
        i, j = rng.choice(k, size=2, replace=False)
        a, b = active[i], active[j]
        left[nxt], right[nxt], height[nxt] = a, b, t
        for idx in sorted((i, j), reverse=True):
            active.pop(idx)
        active.append(nxt)
        nxt += 1
    tree = Tree(left, right, height, 2 * n - 2)
    if root_height is not None:
        tree.height *= root_height / tree.height[tree.root]
    return tree


def yule_tree(tip_count, rng, root_height=None):
    """Pure-birth (Yule) tree: the coalescent with rate k instead of k(k-1)/2 — much more balanced,
    so far fewer dependency levels than a Kingman tree of the same size."""
    n = tip_count
    left = [-1] * (2 * n - 1)
    right = [-1] * (2 * n - 1)
    height = [0.0] * (2 * n - 1)
    active = list(range(n))
    t = 0.0
    nxt = n
    while len(active) > 1:
        k = len(active)
        t += rng.exponential(1.0 / k)
        i, j = rng.choice(k, size=2, replace=False)
        a, b = active[i], active[j]
        left[nxt], right[nxt], height[nxt] = a, b, t
        for idx in sorted((i, j), reverse=True):
            active.pop(idx)
        active.append(nxt)
        nxt += 1
    tree = Tree(left, right, height, 2 * n - 2)
    if root_height is not None:
        tree.height *= root_height / tree.height[tree.root]
    return tree


def caterpillar_tree(tip_count, root_height=1.0):
    """Maximally unbalanced tree: T-1 dependency levels (worst case for level batching)."""
    n = tip_count
    left = [-1] * (2 * n - 1)
    right = [-1] * (2 * n - 1)
    height = [0.0] * (2 * n - 1)
    prev = 0
    for i in range(1, n):
        node = n + i - 1
        left[node], right[node] = prev, i
        height[node] = root_height * i / (n - 1)
        prev = node
    return Tree(left, right, height, 2 * n - 2)
