"""ctypes binding of libpq_rccl.so (include/pq_rccl.h): RCCL communicator bootstrap, the column all-gather and the
row reduce-scatter.
No fallback: a missing library raises."""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (first: libpq_rccl.so shares the librccl.so.1 / libamdhip64.so.7 torch loaded)

from ._lib import PQError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpq_rccl.so")
UNIQUE_ID_BYTES = 128
EXPORTS = ("pq_rccl_last_error", "pq_comm_unique_id", "pq_comm_init_rank", "pq_comm_destroy",
           "pq_allgather_cols_workspace_bytes", "pq_allgather_cols", "pq_unstack_cols",
           "pq_reduce_scatter_rows_workspace_bytes", "pq_reduce_scatter_rows", "pq_allgather_cols_v_workspace_bytes",
           "pq_allgather_cols_v", "pq_allgather_cols_rows_async", "pq_comm_join", "pq_allgather_rows_t", "pq_comm_count", "pq_unstack_cols_v",
           "pq_allreduce_max_u32", "pq_allgather_bytes")
_lib = None
i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PQError(f"{LIB_PATH} not found: build it with `make -C protoquant_amd/csrc`")
    L = ctypes.CDLL(LIB_PATH)
    L.pq_rccl_last_error.restype = ctypes.c_char_p
    L.pq_comm_unique_id.restype = i32
    L.pq_comm_unique_id.argtypes = [vp]
    L.pq_comm_init_rank.restype = i32
    L.pq_comm_init_rank.argtypes = [ctypes.POINTER(vp), i32, vp, i32]
    L.pq_comm_destroy.restype = i32
    L.pq_comm_destroy.argtypes = [vp]
    L.pq_allgather_cols_workspace_bytes.restype = sz
    L.pq_allgather_cols_workspace_bytes.argtypes = [i32, i64, i64, i32]
    L.pq_allgather_cols.restype = i32
    L.pq_allgather_cols.argtypes = [vp, i32, vp, vp, i64, i64, i32, vp, sz, vp]
    L.pq_unstack_cols.restype = i32
    L.pq_unstack_cols.argtypes = [vp, vp, i32, i64, i64, i32, vp]
    L.pq_reduce_scatter_rows_workspace_bytes.restype = sz
    L.pq_reduce_scatter_rows_workspace_bytes.argtypes = [i32, i64, i64, i32]
    L.pq_reduce_scatter_rows.restype = i32
    L.pq_reduce_scatter_rows.argtypes = [vp, i32, vp, vp, i64, i64, i32, vp, sz, vp]
    L.pq_allgather_cols_v_workspace_bytes.restype = sz
    L.pq_allgather_cols_v_workspace_bytes.argtypes = [i32, i64, i64, i32]
    L.pq_allgather_cols_v.restype = i32
    L.pq_allgather_cols_v.argtypes = [vp, vp, i64, vp, i64, i64, i64, i32, vp, sz, vp]
    L.pq_allgather_cols_rows_async.restype = i32
    L.pq_allgather_cols_rows_async.argtypes = [vp, vp, i64, vp, i64, i64, i64, i64, i64, i32, vp, sz, vp]
    L.pq_comm_join.restype = i32
    L.pq_comm_join.argtypes = [vp, vp]
    L.pq_allgather_rows_t.restype = i32
    L.pq_allgather_rows_t.argtypes = [vp, vp, vp, i64, i64, i32, vp]
    L.pq_unstack_cols_v.restype = i32
    L.pq_unstack_cols_v.argtypes = [vp, vp, i64, i32, i64, i64, i32, vp]
    L.pq_allreduce_max_u32.restype = i32
    L.pq_allreduce_max_u32.argtypes = [vp, vp, i64, vp]
    L.pq_allgather_bytes.restype = i32
    L.pq_allgather_bytes.argtypes = [vp, vp, vp, i64, vp]
    L.pq_comm_count.restype = i32
    L.pq_comm_count.argtypes = [vp, ctypes.POINTER(i32)]
    _lib = L
    return L


def check(status: int, what: str):
    if status != 0:
        raise PQError(f"{what} failed (status {status}): {lib().pq_rccl_last_error().decode()}")
