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


def sparse_octree(voxels, depth: int):
    """Descriptor array (reference format, include/map/Octree.h:89-94) of a tree of any depth that holds just the given
    solid voxels: root first, every node's kept children in one block in slot order (i = x | y<<1 | z<<2), blocks in
    breadth-first order, near pointers only.  Bottom level: valid = occupancy, leaf = 0xFF; above: valid for a kept
    child, leaf for an empty one.  Returns (descriptors, root_index)."""
    dim = 1 << depth

    def build(vs, size):                                  # nested dict: slot -> subtree (or occupancy byte at size 2)
        half = size // 2
        buckets = {}
        for (x, y, z) in vs:
            i = (x >= half) | ((y >= half) << 1) | ((z >= half) << 2)
            buckets.setdefault(int(i), []).append((x % half, y % half, z % half))
        if size == 2:
            return sum(1 << i for i in buckets)
        return {i: build(b, half) for i, b in buckets.items()}

    for v in voxels:
        assert all(0 <= c < dim for c in v)
    root = build([tuple(int(c) for c in v) for v in voxels], dim)
    out, queue = [0], [(0, root, dim)]                    # slot 0 = root; breadth-first
    while queue:
        index, node, size = queue.pop(0)
        if size == 2:
            out[index] = (node << 16) | (0xff << 24)
            continue
        valid = sum(1 << i for i in node)
        first = len(out)
        assert first - index < 0x8000, "sparse_octree: near pointers only"
        for i in sorted(node):
            queue.append((len(out), node[i], size // 2))
            out.append(0)
        out[index] = (first - index) | (valid << 16) | ((~valid & 0xff) << 24)
    return np.array(out, dtype=np.uint64), 0


def box_decode(c):
    """Extent code of the empty boxes (include/vrc.h vrc_read_empty_boxes): c (c < 4), (4 | c & 3) << (c / 4 - 1) otherwise."""
    c = int(c)
    return c if c < 4 else (4 | (c & 3)) << ((c >> 2) - 1)


def empty_children(desc: np.ndarray, root: int, dim: int):
    """Every EMPTY child slot of every descriptor the root reaches: yields (descriptor index, slot, (x, y, z) of the child's
    cube, its size) -- an independent walk of the array on the host (format of include/map/Octree.h:89-94)."""
    d = [int(v) for v in desc]
    stack = [(int(root), 0, 0, 0, dim)]
    while stack:
        index, x, y, z, size = stack.pop()
        v = d[index]
        valid, leaf = (v >> 16) & 0xff, (v >> 24) & 0xff
        half = size // 2
        at = index + (v & 0x7fff)
        if v & 0x8000:
            at = d[at]
        rank = 0
        for k in range(8):
            cx, cy, cz = x + (half if k & 1 else 0), y + (half if k & 2 else 0), z + (half if k & 4 else 0)
            if not (valid >> k & 1):
                yield index, k, (cx, cy, cz), half
                continue
            if half > 1 and not (leaf >> k & 1):
                stack.append((at + rank, cx, cy, cz, half))
            rank += 1
