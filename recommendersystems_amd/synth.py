"""Deterministic synthetic bipartite like-graphs (SURVEY.md section 8d / BASELINE.md section 3).

Counter-based and integer-only (splitmix64 + 64x64->high-64 multiplies), so the same graph
comes out on any host.  Users are indexed first, items after (the loader's order,
DataLoader.cs:229-231); every like becomes two directed LIKE links of raw weight 1
(DataLoader.cs:293-294); (user, item) pairs are de-duplicated as DataLoader.addLink would
(DataLoader.cs:64-70); out-links of a source are in ascending target order.
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)
SEED_BASE = 0x5EED000000000000

CONFIGS = {
    # name: (config#, users, items, likes, seeds per GPU)
    "tiny": (0, 2_000, 10_000, 100_000, 64),
    "C2": (2, 100_000, 500_000, 10_000_000, 1000),
    "C3": (3, 162_000, 62_000, 25_000_000, 8192),
    "C4": (4, 1_000_000, 5_000_000, 100_000_000, 1024),
    # config 5's graph (row-partitioned over 8 GPUs in BASELINE.md); its 2 G links also fit ONE MI355X (about 110 GB during the build)
    "C5": (5, 10_000_000, 50_000_000, 1_000_000_000, 8),
}


def splitmix64(x: np.ndarray) -> np.ndarray:
    z = x + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def mulhi(a: np.ndarray, b) -> np.ndarray:
    """High 64 bits of the 128-bit product of unsigned 64-bit operands."""
    b = np.asarray(b, dtype=np.uint64)
    a_lo, a_hi = a & M32, a >> np.uint64(32)
    b_lo, b_hi = b & M32, b >> np.uint64(32)
    lo_lo = a_lo * b_lo
    hi_lo = a_hi * b_lo
    lo_hi = a_lo * b_hi
    hi_hi = a_hi * b_hi
    cross = (lo_lo >> np.uint64(32)) + (hi_lo & M32) + lo_hi
    return hi_hi + (hi_lo >> np.uint64(32)) + (cross >> np.uint64(32))


def like_keys(cfg_no: int, U: int, I: int, E: int, chunk: int = 1 << 21) -> np.ndarray:
    """key = u*I + v for edge counters 0..E-1 (duplicates still in)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty(E, dtype=np.uint64)
    seed = np.uint64(SEED_BASE + cfg_no)

    def work(s):
        with np.errstate(over="ignore"):
            e = np.arange(s, min(E, s + chunk), dtype=np.uint64)
            base = seed + np.uint64(5) * e
            h = [splitmix64(base + np.uint64(m)) for m in range(5)]
            u = mulhi(mulhi(h[0], h[1]), np.uint64(U))                 # product of 2 uniforms: active users
            v = mulhi(mulhi(mulhi(h[2], h[3]), h[4]), np.uint64(I))   # product of 3: popular items
            out[s:s + e.shape[0]] = u * np.uint64(I) + v

    # numpy releases the GIL inside ufuncs; chunks are independent (counter-based generator)
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(work, range(0, E, chunk)))
    return out


def _sort_unique(keys: np.ndarray) -> np.ndarray:
    try:
        import torch
        if torch.cuda.is_available():
            t = torch.from_numpy(keys.view(np.int64)).cuda()     # keys < 2^63: order preserved
            t = torch.unique(t, sorted=True)
            return t.cpu().numpy().view(np.uint64)
    except Exception:
        pass
    return np.unique(keys)


def _sort(keys: np.ndarray) -> np.ndarray:
    try:
        import torch
        if torch.cuda.is_available():
            t = torch.from_numpy(keys.view(np.int64)).cuda()
            return torch.sort(t).values.cpu().numpy().view(np.uint64)
    except Exception:
        pass
    return np.sort(keys)


def bipartite(cfg_no: int, U: int, I: int, E: int) -> dict:
    """Flat graph in the layout of include/rwr.h: node_id, node_type, rowptr, dst, etype, w."""
    keys = _sort_unique(like_keys(cfg_no, U, I, E))        # sorted by (u, v)
    Ed = int(keys.shape[0])
    u = (keys // np.uint64(I)).astype(np.int64)
    v = (keys % np.uint64(I)).astype(np.int64)
    n = U + I
    rowptr = np.zeros(n + 1, dtype=np.int64)
    dst = np.empty(2 * Ed, dtype=np.int32)
    # user rows: targets U+v ascending (keys are sorted by (u, v))
    ucount = np.bincount(u, minlength=U)
    dst[:Ed] = (v + U).astype(np.int32)
    # item rows: sources u ascending -> sort by (v, u)
    k2 = _sort((v.astype(np.uint64) * np.uint64(U) + u.astype(np.uint64)))
    vcount = np.bincount(v, minlength=I)
    dst[Ed:] = (k2 % np.uint64(U)).astype(np.int32)
    np.cumsum(np.concatenate([ucount, vcount]), out=rowptr[1:])
    node_type = np.empty(n, dtype=np.uint8)
    node_type[:U] = 1   # NodeType.USER
    node_type[U:] = 2   # NodeType.ITEM
    return dict(node_id=np.arange(n, dtype=np.int64), node_type=node_type, rowptr=rowptr, dst=dst,
                etype=np.ones(2 * Ed, dtype=np.uint8), w=np.ones(2 * Ed, dtype=np.float64),
                likes=Ed, users=U, items=I)


def config(name: str) -> dict:
    no, U, I, E, K = CONFIGS[name]
    g = bipartite(no, U, I, E)
    g["name"] = name
    g["seeds_per_gpu"] = K
    return g


def seeds_for(U: int, K_total: int, first: int, count: int) -> np.ndarray:
    """seed_k = floor(k*U/K_total) for k in [first, first+count): distinct users, heavy and light."""
    k = np.arange(first, first + count, dtype=np.int64)
    return ((k * U) // K_total).astype(np.int32)


def algorithmic_bytes_per_step(n: int, nnz: int, K: int, v: int = 8) -> int:
    """compulsory HBM bytes of ONE power-iteration step over K seeds:
    matrix once (4-byte index + v-byte value per entry), row offsets, dangling flags,
    read X once + write Y once, seeds + restart scalars."""
    return nnz * (4 + v) + (n + 1) * 8 + n + 2 * n * K * 8 + K * 12
