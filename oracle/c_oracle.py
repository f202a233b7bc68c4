"""TEST INFRASTRUCTURE — ctypes front end of oracle/liboracle.so (the plain-C QSPEC restatement,
oracle/qspec_oracle.c).  numpy in, numpy out; half tensors are uint16 bit patterns + a dtype code
(0 bf16, 1 fp16, 2 f32).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None
i64, vp = ctypes.c_int64, ctypes.c_void_p


def build():
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return vp(a.ctypes.data) if a is not None else vp(0)


def _store(dtype):
    return np.float32 if dtype == 2 else np.uint16


def quant_rowwise(x: np.ndarray, dtype: int):
    x = np.ascontiguousarray(x)
    r, c = x.shape
    q = np.zeros((r, c), np.int8)
    s = np.zeros(r, np.float32)
    lib().oq_quant_rowwise(_p(x), dtype, i64(r), i64(c), i64(c), _p(q), i64(c), _p(s))
    return q, s


def quant_colwise(x: np.ndarray, dtype: int):
    x = np.ascontiguousarray(x)
    r, c = x.shape
    q = np.zeros((r, c), np.int8)
    s = np.zeros(c, np.float32)
    lib().oq_quant_colwise(_p(x), dtype, i64(r), i64(c), i64(c), _p(q), i64(c), _p(s))
    return q, s


def dequant(q: np.ndarray, scale: np.ndarray, axis: int, out_dtype: int):
    q = np.ascontiguousarray(q)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    r, c = q.shape
    out = np.zeros((r, c), _store(out_dtype))
    lib().oq_dequant(_p(q), i64(c), _p(scale), axis, i64(r), i64(c), _p(out), i64(c), out_dtype)
    return out


def gemm_s8s8s32(a: np.ndarray, b: np.ndarray):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    M, K = a.shape
    N = b.shape[0]
    c = np.zeros((M, N), np.int32)
    lib().oq_gemm_s8s8s32(_p(a), i64(K), _p(b), i64(K), _p(c), i64(N), i64(M), i64(N), i64(K))
    return c


def qlinear_s8(a, a_scale, b, b_scale, bias, out_dtype: int):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    a_scale = np.ascontiguousarray(a_scale, dtype=np.float32)
    b_scale = np.ascontiguousarray(b_scale, dtype=np.float32)
    if bias is not None:
        bias = np.ascontiguousarray(bias)
    M, K = a.shape
    N = b.shape[0]
    y = np.zeros((M, N), _store(out_dtype))
    lib().oq_qlinear_s8(_p(a), i64(K), _p(a_scale), _p(b), i64(K), _p(b_scale), _p(bias), _p(y), i64(N),
                        out_dtype, i64(M), i64(N), i64(K))
    return y


def exp_spec(t: np.ndarray) -> np.ndarray:
    L = lib()
    L.oq_exp_spec.restype = ctypes.c_float
    L.oq_exp_spec.argtypes = [ctypes.c_float]
    t = np.asarray(t, np.float32)
    return np.array([L.oq_exp_spec(float(v)) for v in t.ravel()], np.float32).reshape(t.shape)


def silu_mul_quant_rowwise(g: np.ndarray, u: np.ndarray, dtype: int, want_h: bool = True):
    """QSPEC S1-S6.  g, u: [rows, cols] stored dtype.  Returns (q, scale, h or None)."""
    g = np.ascontiguousarray(g); u = np.ascontiguousarray(u)
    r, c = g.shape
    q = np.zeros((r, c), np.int8)
    s = np.zeros(r, np.float32)
    h = np.zeros((r, c), _store(dtype)) if want_h else None
    lib().oq_silu_mul_quant_rowwise(_p(g), i64(c), _p(u), i64(c), dtype, i64(r), i64(c), _p(q), i64(c), _p(s), _p(h), i64(c))
    return q, s, h


def rmsnorm_quant_rowwise(x: np.ndarray, weight: np.ndarray, eps: float, dtype: int, want_h: bool = True):
    """QSPEC N1-N6.  x: [rows, cols], weight: [cols], stored dtype.  Returns (q, scale, h or None, rs)."""
    x = np.ascontiguousarray(x); weight = np.ascontiguousarray(weight)
    r, c = x.shape
    q = np.zeros((r, c), np.int8)
    s = np.zeros(r, np.float32)
    rs = np.zeros(r, np.float32)
    h = np.zeros((r, c), _store(dtype)) if want_h else None
    lib().oq_rmsnorm_quant_rowwise(_p(x), i64(c), _p(weight), ctypes.c_float(eps), dtype, i64(r), i64(c), _p(q), i64(c), _p(s),
                                   _p(h), i64(c), _p(rs))
    return q, s, h, rs
