"""ctypes binding of librwr.so -- the C-ABI declared in include/rwr.h.

There is no CPU fallback: if the HIP library is missing or no gfx950 device is usable
the calls raise.  Nothing here imports the parity oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librwr.so")

RWR_OK, RWR_E_INVALID, RWR_E_RANGE, RWR_E_NO_DEVICE, RWR_E_HIP, RWR_E_NOMEM, RWR_E_CAPACITY, RWR_E_UNSUPPORTED = range(8)
RWR_MODE_EXACT = 0
RWR_RUN_ITERATIONS, RWR_RUN_THRESHOLD, RWR_RUN_DEFAULT_THRESHOLD = 0, 1, 2

# every symbol include/rwr.h declares (tests check that the library exports all of them)
EXPORTS = [
    "rwr_version", "rwr_device_count", "rwr_last_error",
    "rwr_graph_create", "rwr_graph_update_links", "rwr_graph_destroy", "rwr_graph_size", "rwr_graph_get_normalized",
    "rwr_recommend", "rwr_recommend_eval", "rwr_recommend_eval_batch", "rwr_eval_graphs", "rwr_recommend_batch", "rwr_model_run", "rwr_model_deliver",
    "rwr_part_begin", "rwr_part_step", "rwr_part_local_step", "rwr_part_finish_step", "rwr_part_rank",
    "rwr_get_stats", "rwr_reset_stats",
]


class rwr_opts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("mode", C.c_int32),
                ("tile_seeds", C.c_int32), ("tile_group", C.c_int32), ("profile", C.c_int32),
                ("workspace_bytes", C.c_int64), ("seed_row_kernel", C.c_int32), ("reserved0", C.c_int32)]


class rwr_graph_desc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("reserved0", C.c_int32), ("node_id", C.POINTER(C.c_int64)),
                ("node_type", C.POINTER(C.c_uint8)), ("rowptr", C.POINTER(C.c_int64)), ("dst", C.POINTER(C.c_int32)),
                ("etype", C.POINTER(C.c_uint8)), ("w", C.POINTER(C.c_double))]


class rwr_stats(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("n", C.c_int32), ("nnz_raw", C.c_int64), ("nnz", C.c_int64),
                ("uniform", C.c_int32), ("tile_seeds", C.c_int32), ("tile_group", C.c_int32), ("mode", C.c_int32),
                ("build_ms", C.c_double), ("spmm_ms", C.c_double), ("spmm_launches", C.c_int64),
                ("spmm_seed_steps", C.c_int64), ("chain_ms", C.c_double), ("chain_launches", C.c_int64),
                ("rank_ms", C.c_double), ("iterate_wall_ms", C.c_double), ("total_wall_ms", C.c_double),
                ("seeds_done", C.c_int64), ("chain_redo_blocks", C.c_int64),
                ("spmm_dense_ms", C.c_double), ("spmm_dense_launches", C.c_int64), ("spmm_dense_seed_steps", C.c_int64),
                ("uniform_path", C.c_int32), ("reserved1", C.c_int32)]


class RwrError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"librwr status {status}: {message}")
        self.status = status


_lib = None


def load():
    """Loads librwr.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    p = C.POINTER
    lib.rwr_version.restype = C.c_char_p
    lib.rwr_last_error.restype = C.c_char_p
    lib.rwr_device_count.restype = C.c_int32
    lib.rwr_graph_create.restype = C.c_int32
    lib.rwr_graph_create.argtypes = [C.c_int32, p(C.c_int64), p(C.c_uint8), p(C.c_int64), p(C.c_int32),
                                     p(C.c_uint8), p(C.c_double), p(rwr_opts), p(C.c_void_p)]
    lib.rwr_graph_update_links.restype = C.c_int32
    lib.rwr_graph_update_links.argtypes = [C.c_void_p, C.c_int64, p(C.c_int64), p(C.c_uint8), p(C.c_double)]
    lib.rwr_graph_destroy.restype = C.c_int32
    lib.rwr_graph_destroy.argtypes = [C.c_void_p]
    lib.rwr_graph_size.restype = C.c_int32
    lib.rwr_graph_size.argtypes = [C.c_void_p, p(C.c_int32), p(C.c_int64), p(C.c_int64)]
    lib.rwr_graph_get_normalized.restype = C.c_int32
    lib.rwr_graph_get_normalized.argtypes = [C.c_void_p, p(C.c_double), p(C.c_uint8)]
    lib.rwr_recommend.restype = C.c_int32
    lib.rwr_recommend.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32, p(C.c_int64),
                                  p(C.c_double), p(C.c_int64)]
    lib.rwr_recommend_eval.restype = C.c_int32
    lib.rwr_recommend_eval.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, p(C.c_int64), C.c_int64,
                                       p(C.c_int64), p(C.c_double), p(C.c_int64)]
    lib.rwr_recommend_eval_batch.restype = C.c_int32
    lib.rwr_recommend_eval_batch.argtypes = [C.c_void_p, p(C.c_int32), C.c_int32, C.c_float, C.c_int32, p(C.c_int64),
                                             p(C.c_int64), p(C.c_int64), p(C.c_double), p(C.c_int64)]
    lib.rwr_eval_graphs.restype = C.c_int32
    lib.rwr_eval_graphs.argtypes = [C.c_int32, p(rwr_graph_desc), p(C.c_int32), C.c_float, C.c_int32, p(C.c_int64), p(C.c_int64),
                                    p(rwr_opts), p(C.c_int64), p(C.c_double), p(C.c_int64)]
    lib.rwr_recommend_batch.restype = C.c_int32
    lib.rwr_recommend_batch.argtypes = [C.c_void_p, p(C.c_int32), C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                        p(C.c_int64), p(C.c_double), p(C.c_int32)]
    lib.rwr_model_run.restype = C.c_int32
    lib.rwr_model_run.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_double, p(C.c_double),
                                  p(C.c_int64)]
    lib.rwr_model_deliver.restype = C.c_int32
    lib.rwr_model_deliver.argtypes = [C.c_void_p, C.c_int32, C.c_double, p(C.c_double), p(C.c_double)]
    lib.rwr_part_begin.restype = C.c_int32
    lib.rwr_part_begin.argtypes = [C.c_void_p, C.c_int32, C.c_int32, p(C.c_int32), C.c_int32, C.c_double, C.c_void_p,
                                   p(C.c_int32)]
    lib.rwr_part_local_step.restype = C.c_int32
    lib.rwr_part_local_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rwr_part_step.restype = C.c_int32
    lib.rwr_part_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rwr_part_finish_step.restype = C.c_int32
    lib.rwr_part_finish_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rwr_part_rank.restype = C.c_int32
    lib.rwr_part_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, p(C.c_int64), p(C.c_double), p(C.c_int32)]
    lib.rwr_get_stats.restype = C.c_int32
    lib.rwr_get_stats.argtypes = [C.c_void_p, p(rwr_stats)]
    lib.rwr_reset_stats.restype = C.c_int32
    lib.rwr_reset_stats.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def check(status: int):
    if status != RWR_OK:
        raise RwrError(status, load().rwr_last_error().decode("utf-8", "replace"))
