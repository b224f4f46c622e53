"""Seeded random graphs in the flat layout of include/rwr.h (raw, un-normalised,
list order).  Exercises everything the reference's loader can produce: multi-edges
with different types (DataLoader.cs:64-70), UNDEFINED-relabelled links
(Experiment.cs:84-101), one-directional non-unit MENTION weights (DataLoader.cs:431-432),
dangling nodes, ETC users, non-monotone ids."""
from __future__ import annotations

import numpy as np

NODE_UNDEFINED, NODE_USER, NODE_ITEM, NODE_ETC = range(4)
(EDGE_UNDEFINED, EDGE_LIKE, EDGE_FRIENDSHIP, EDGE_FOLLOW, EDGE_MENTION,
 EDGE_AUTHORSHIP, EDGE_PURCHASE, EDGE_ETC) = range(8)


def random_graph(seed: int, n_users: int, n_items: int, n_likes: int, *, n_etc: int = 0,
                 p_undefined: float = 0.05, n_friend: int = 0, n_mention: int = 0,
                 n_author: int = 0, uniform: bool = False, shuffle_lists: bool = True):
    """Returns dict(node_id, node_type, rowptr, dst, etype, w).  Users first, then
    items, then ETC users (the loader's order, DataLoader.cs:229-231,258)."""
    rng = np.random.default_rng(seed)
    n = n_users + n_items + n_etc
    node_type = np.array([NODE_USER] * n_users + [NODE_ITEM] * n_items + [NODE_ETC] * n_etc, dtype=np.uint8)
    node_id = rng.permutation(np.arange(1000, 1000 + 7 * n, 7, dtype=np.int64))  # unique, unordered
    lists = [[] for _ in range(n)]

    def add(src, tgt, ty, wt):
        # DataLoader.addLink de-duplicates on (target, type): DataLoader.cs:64-70
        for (t, y, _) in lists[src]:
            if t == tgt and y == ty:
                return
        lists[src].append((tgt, ty, wt))

    if n_users and n_items:
        # skewed: product of uniforms
        us = (rng.random(n_likes) * rng.random(n_likes) * n_users).astype(np.int64)
        vs = (rng.random(n_likes) * rng.random(n_likes) * rng.random(n_likes) * n_items).astype(np.int64)
        for u, v in zip(us, vs):
            add(int(u), n_users + int(v), EDGE_LIKE, 1.0)
            add(n_users + int(v), int(u), EDGE_LIKE, 1.0)
    for _ in range(n_friend):
        a, b = rng.integers(0, n_users, 2)
        if a != b:
            add(int(a), int(b), EDGE_FRIENDSHIP, 1.0)
            add(int(b), int(a), EDGE_FRIENDSHIP, 1.0)
    for _ in range(n_author):
        a = int(rng.integers(0, n_users)); t = n_users + int(rng.integers(0, n_items))
        add(a, t, EDGE_AUTHORSHIP, 1.0)
        add(t, a, EDGE_AUTHORSHIP, 1.0)
    pool = n_users + n_items
    for _ in range(n_mention):
        a = int(rng.integers(0, n_users))
        b = int(rng.integers(0, n_users)) if n_etc == 0 or rng.random() < 0.5 else pool + int(rng.integers(0, n_etc))
        if a != b:
            add(a, b, EDGE_MENTION, float(rng.integers(1, 9)) * float(np.log(rng.integers(2, 40))) / 7.3)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    dst, etype, w = [], [], []
    for i in range(n):
        L = lists[i]
        if shuffle_lists and len(L) > 1:
            L = [L[j] for j in rng.permutation(len(L))]
        for (t, y, wt) in L:
            if (not uniform) and rng.random() < p_undefined:
                y = EDGE_UNDEFINED      # relabelled, stays in the raw list (Experiment.cs:96)
            dst.append(t); etype.append(y); w.append(wt)
        rowptr[i + 1] = len(dst)
    return dict(node_id=node_id, node_type=node_type, rowptr=rowptr,
                dst=np.array(dst, dtype=np.int32), etype=np.array(etype, dtype=np.uint8),
                w=np.array(w, dtype=np.float64))


def kat1():
    """bipartite path, no dangling node."""
    node_id = np.array([10, 11, 12, 13], dtype=np.int64)
    node_type = np.array([NODE_USER, NODE_ITEM, NODE_USER, NODE_ITEM], dtype=np.uint8)
    lists = {0: [1], 1: [0, 2], 2: [1, 3], 3: [2]}
    return _from_lists(node_id, node_type, lists)


def kat2():
    """node 1 is dangling."""
    node_id = np.array([7, 9], dtype=np.int64)
    node_type = np.array([NODE_USER, NODE_ITEM], dtype=np.uint8)
    return _from_lists(node_id, node_type, {0: [1], 1: []})


def _from_lists(node_id, node_type, lists):
    n = len(node_id)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    dst = []
    for i in range(n):
        dst += lists[i]
        rowptr[i + 1] = len(dst)
    return dict(node_id=node_id, node_type=node_type, rowptr=rowptr, dst=np.array(dst, dtype=np.int32),
                etype=np.full(len(dst), EDGE_LIKE, dtype=np.uint8), w=np.ones(len(dst), dtype=np.float64))
