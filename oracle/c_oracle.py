"""ctypes binding of oracle/librwr_oracle.so (the flat-array C restatement).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/rwr_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "librwr_oracle.so")
    src = os.path.join(_HERE, "rwr_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librwr_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librwr_oracle.so")
        if not os.path.exists(so):
            so = build()
        _LIB = C.CDLL(so)
        _LIB.rwr_oracle_model_run.restype = C.c_int64
        _LIB.rwr_oracle_recommend.restype = C.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class FlatGraph:
    """Raw (un-normalised) out-CSR in list order + Graph.buildGraph() applied by the oracle."""

    def __init__(self, node_id, node_type, rowptr, dst, etype, w):
        self.node_id = np.ascontiguousarray(node_id, dtype=np.int64)
        self.node_type = np.ascontiguousarray(node_type, dtype=np.uint8)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.dst = np.ascontiguousarray(dst, dtype=np.int32)
        self.etype = np.ascontiguousarray(etype, dtype=np.uint8)
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self.n = int(self.node_id.shape[0])
        self.nnz = int(self.dst.shape[0])
        assert self.rowptr.shape[0] == self.n + 1 and int(self.rowptr[-1]) == self.nnz
        self.w_norm = np.zeros(self.nnz, dtype=np.float64)
        self.dangling = np.zeros(self.n, dtype=np.uint8)
        lib().rwr_oracle_build(C.c_int32(self.n), _p(self.rowptr, C.c_int64), _p(self.dst, C.c_int32),
                               _p(self.etype, C.c_uint8), _p(self.w, C.c_double),
                               _p(self.w_norm, C.c_double), _p(self.dangling, C.c_uint8))

    def _g(self):
        return (C.c_int32(self.n), _p(self.rowptr, C.c_int64), _p(self.dst, C.c_int32),
                _p(self.etype, C.c_uint8), _p(self.w_norm, C.c_double), _p(self.dangling, C.c_uint8))

    def model_run(self, d: float, seed: int = -1, mode: int = 0, value: float = 0.0,
                  dense: bool = False, max_iter: int = 0):
        rank = np.zeros(self.n, dtype=np.float64)
        it = lib().rwr_oracle_model_run(*self._g(), C.c_double(d), C.c_int32(seed), C.c_int32(mode),
                                        C.c_double(value), C.c_int32(1 if dense else 0),
                                        C.c_int64(max_iter), _p(rank, C.c_double))
        if it < 0:
            raise ValueError("rwr_oracle_model_run failed")
        return rank, int(it)

    def recommend(self, seed: int, d: float, n_iter: int, top_n: int = 0, want_rank: bool = False):
        cap = self.n
        ids = np.zeros(cap, dtype=np.int64)
        sc = np.zeros(cap, dtype=np.float64)
        rank = np.zeros(self.n, dtype=np.float64) if want_rank else None
        cnt = lib().rwr_oracle_recommend(
            C.c_int32(self.n), _p(self.node_id, C.c_int64), _p(self.node_type, C.c_uint8),
            _p(self.rowptr, C.c_int64), _p(self.dst, C.c_int32), _p(self.etype, C.c_uint8),
            _p(self.w_norm, C.c_double), _p(self.dangling, C.c_uint8),
            C.c_int32(seed), C.c_float(d), C.c_int32(n_iter), C.c_int64(top_n),
            _p(ids, C.c_int64), _p(sc, C.c_double),
            _p(rank, C.c_double) if want_rank else None)
        if cnt < 0:
            raise ValueError("rwr_oracle_recommend failed")
        if want_rank:
            return ids[:cnt].copy(), sc[:cnt].copy(), rank
        return ids[:cnt].copy(), sc[:cnt].copy()

    def recommend_batch(self, seeds, d: float, n_iter: int, top_n: int, n_threads: int = 0):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        K = int(seeds.shape[0])
        ids = np.zeros((K, top_n), dtype=np.int64)
        sc = np.zeros((K, top_n), dtype=np.float64)
        counts = np.zeros(K, dtype=np.int32)
        rc = lib().rwr_oracle_recommend_batch(
            C.c_int32(self.n), _p(self.node_id, C.c_int64), _p(self.node_type, C.c_uint8),
            _p(self.rowptr, C.c_int64), _p(self.dst, C.c_int32), _p(self.etype, C.c_uint8),
            _p(self.w_norm, C.c_double), _p(self.dangling, C.c_uint8),
            _p(seeds, C.c_int32), C.c_int32(K), C.c_float(d), C.c_int32(n_iter), C.c_int32(top_n),
            C.c_int32(n_threads), _p(ids, C.c_int64), _p(sc, C.c_double), _p(counts, C.c_int32))
        if rc != 0:
            raise ValueError("rwr_oracle_recommend_batch failed")
        return ids, sc, counts


def evaluate(ranked_ids, test_ids):
    """Experiment.cs:121-128 -> (nHits, sumPrecision)."""
    r = np.ascontiguousarray(ranked_ids, dtype=np.int64)
    t = np.ascontiguousarray(test_ids, dtype=np.int64)
    hits = C.c_int64(0)
    sp = C.c_double(0.0)
    rc = lib().rwr_oracle_evaluate(_p(r, C.c_int64), C.c_int64(len(r)), _p(t, C.c_int64), C.c_int64(len(t)),
                                   C.byref(hits), C.byref(sp))
    if rc != 0:
        raise ValueError("rwr_oracle_evaluate failed")
    return int(hits.value), float(sp.value)


def max_threads() -> int:
    return int(lib().rwr_oracle_max_threads())
