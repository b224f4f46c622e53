"""numpy stand-in for the HIP slab backend (recommendersystems_amd.partitioned.HipSlabBackend): same interface,
same maths, CPU only.  TEST CODE: lets the multi-rank host logic (slab partition, all-reduce exchange, owner-only
ranking, result merge) run under gloo without a GPU.  Normalisation comes from the oracle's buildGraph."""
import numpy as np
import torch

from oracle.c_oracle import FlatGraph


class NumpySlabBackend:
    def __init__(self, local_flat, lo, hi, **opts):
        self.lo, self.hi = lo, hi
        self.F = FlatGraph(**local_flat)          # Graph.buildGraph on the slab's links
        self.n = self.F.n
        rp, dst = self.F.rowptr, self.F.dst
        self.src = np.repeat(np.arange(self.n), np.diff(rp))
        self.expl = self.F.etype != 0

    def begin(self, seeds, d):
        self.seeds, self.c1 = np.asarray(seeds), 1 - d
        K = len(seeds)
        x = torch.zeros(self.n * K, dtype=torch.float64)
        x.view(self.n, K)[self.seeds, np.arange(K)] = float(self.n)
        return x, torch.zeros_like(x), torch.zeros(K, dtype=torch.float64)

    def local_step(self, x, y, r):
        K = len(self.seeds)
        X = x.view(self.n, K).numpy()
        Y = np.zeros_like(X)
        m = self.expl
        np.add.at(Y, self.F.dst[m], (self.c1 * X[self.src[m]]) * self.F.w_norm[m][:, None])
        y.view(self.n, K)[:] = torch.from_numpy(Y)
        sl = slice(self.lo, self.hi)
        dang = self.F.dangling[sl].astype(bool)[:, None]
        rr = np.where(dang, X[sl], X[sl] - self.c1 * X[sl])
        r[:] = torch.from_numpy(rr.sum(axis=0))

    def step(self, x, y):
        # partial y over all rows + this slab's restart mass at the seeds' rows (rwr_part_step)
        r = torch.zeros(len(self.seeds), dtype=torch.float64)
        self.local_step(x, y, r)
        self.finish_step(y, r)

    def finish_step(self, y, r):
        K = len(self.seeds)
        y.view(self.n, K)[self.seeds, np.arange(K)] += r

    def rank(self, x, top_n):
        K = len(self.seeds)
        X = x.view(self.n, K).numpy()
        ids = np.zeros((K, top_n), dtype=np.int64)
        sc = np.zeros((K, top_n), dtype=np.float64)
        cnt = np.full(K, -1, dtype=np.int32)
        for k, s in enumerate(self.seeds):
            if not (self.lo <= s < self.hi):
                continue
            e0, e1 = self.F.rowptr[s], self.F.rowptr[s + 1]
            excl = set(self.F.dst[e0:e1][self.F.etype[e0:e1] == 1].tolist())
            cand = [(-X[j, k], -int(self.F.node_id[j])) for j in range(self.n)
                    if self.F.node_type[j] == 2 and j not in excl]
            cand.sort()
            c = min(top_n, len(cand))
            cnt[k] = c
            ids[k, :c] = [-v[1] for v in cand[:c]]
            sc[k, :c] = [-v[0] for v in cand[:c]]
        return ids, sc, cnt

    def to_exchange(self, t):
        return t

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))
