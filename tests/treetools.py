"""Test helpers: layout-independent views of a descriptor array (include/map/Octree.h:89-94)."""
import sys

import numpy as np


def canonical(desc: np.ndarray, root: int, dim: int):
    """Nested signature of the tree a descriptor array encodes, independent of where nodes are stored: two arrays
    describe the same octree iff their signatures are equal.  Also returns (nodes, far pointers) visited."""
    sys.setrecursionlimit(10000)
    d = [int(v) for v in desc]
    stats = [0, 0]

    def node(index, size):
        v = d[index]
        stats[0] += 1
        valid, leaf = (v >> 16) & 0xff, (v >> 24) & 0xff
        if size == 2:
            return (valid, leaf)
        at = index + (v & 0x7fff)
        if v & 0x8000:
            stats[1] += 1
            at = d[at]
        kids, k = [], 0
        for i in range(8):
            if valid >> i & 1:
                if not (leaf >> i & 1):
                    kids.append(node(at + k, size // 2))
                k += 1
        return (valid, leaf, tuple(kids))

    return node(root, dim), tuple(stats)
