"""TEST INFRASTRUCTURE — torch-free numpy restatement of QSPEC v2.

Parity status: *parity unpinned by the reference* (``/root/reference`` has no source for this path;
only ``/root/reference/CODE_OF_CONDUCT.md:1-80`` exists).  Pinned instead against
``tests/golden/*.npz`` (produced by ``oracle/torch_ref.py`` around ``torch._int_mm``).

Tensors are numpy arrays.  Half types are carried as ``uint16`` bit patterns with a dtype tag:
``"bf16"``, ``"fp16"``, ``"f32"`` (codes 0/1/2 — the same codes ``include/pq_hip.h`` uses).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

DT_BF16, DT_FP16, DT_F32 = 0, 1, 2
_NAMES = {"bf16": DT_BF16, "fp16": DT_FP16, "f32": DT_F32}


def dt(code):
    return _NAMES[code] if isinstance(code, str) else int(code)


# ---------------------------------------------------------------- dtype plumbing
def to_f32(a: np.ndarray, dtype) -> np.ndarray:
    """Exact up-conversion of a stored tensor to float32."""
    d = dt(dtype)
    if d == DT_F32:
        return np.asarray(a, dtype=np.float32)
    a = np.asarray(a, dtype=np.uint16)
    if d == DT_FP16:
        return a.view(np.float16).astype(np.float32)
    return (a.astype(np.uint32) << np.uint32(16)).view(np.float32)


def from_f32(t: np.ndarray, dtype) -> np.ndarray:
    """Round-to-nearest-even down-conversion; bf16 keeps NaN a (quiet) NaN like v_cvt_pk_bf16_f32."""
    d = dt(dtype)
    t = np.asarray(t, dtype=np.float32)
    if d == DT_F32:
        return t.copy()
    if d == DT_FP16:
        with np.errstate(over="ignore"):
            return t.astype(np.float16).view(np.uint16)
    u = t.view(np.uint32)
    rounded = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    nan = np.isnan(t)
    quiet = ((u >> np.uint32(16)) | np.uint32(0x0040)).astype(np.uint16)
    return np.where(nan, quiet, rounded)


# ---------------------------------------------------------------- QSPEC stages
_CANON_NAN = np.array([0x7FC00000], np.uint32).view(np.float32)[0]


def quantize(x: np.ndarray, dtype, reduce_axis: int):
    """QSPEC Q1-Q6. reduce_axis = 1 (per-token rows) or 0 (per-channel columns)."""
    xf = to_f32(x, dtype)
    assert xf.ndim == 2
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        ax = np.abs(xf)
        if xf.shape[reduce_axis] == 0:
            amax = np.zeros(xf.shape[1 - reduce_axis], np.float32)
        else:
            amax = np.maximum.reduce(ax, axis=reduce_axis).astype(np.float32)   # Q2: NaN-PROPAGATING max (np.maximum, like torch.amax)
        scale = (amax / np.float32(127.0)).astype(np.float32)    # Q3 true division
        scale = np.where(amax == 0, np.float32(1.0), scale).astype(np.float32)
        scale = np.where(np.isnan(scale), _CANON_NAN, scale).astype(np.float32)   # Q3: a NaN scale is the canonical quiet NaN 0x7FC00000
        s = np.expand_dims(scale, reduce_axis)
        t = np.rint((xf / s).astype(np.float32))                 # Q4 true division + RNE
        t = np.where(np.isnan(t), np.float32(0), t)              # Q5 NaN -> 0
        q = np.clip(t, -128.0, 127.0).astype(np.int8)            # Q6
    return q, scale


def row_amax_bits(x: np.ndarray, dtype) -> np.ndarray:
    """Q2 per token as f32 BIT PATTERNS (uint32 [rows]): non-negative floats and NaNs (above +Inf) order as unsigned integers, so an integer max of these
    over column blocks is the exact, NaN-propagating row amax — what the column-sharded gated MLP all-reduces (pq_silu_mul_rowamax)."""
    xf = to_f32(x, dtype)
    if xf.shape[1] == 0:
        return np.zeros(xf.shape[0], np.uint32)
    return (np.abs(xf).view(np.uint32) & np.uint32(0x7FFFFFFF)).max(axis=1).astype(np.uint32)


def quantize_rows_with_amax(x: np.ndarray, dtype, amax_bits: np.ndarray):
    """Q3-Q6 per token against a GIVEN row amax (f32 bit patterns): the column block of quantize(x_full, 1) when amax_bits is the max over all blocks."""
    xf = to_f32(x, dtype)
    amax = np.asarray(amax_bits, np.uint32).view(np.float32)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        scale = (amax / np.float32(127.0)).astype(np.float32)
        scale = np.where(amax == 0, np.float32(1.0), scale).astype(np.float32)
        scale = np.where(np.isnan(scale), _CANON_NAN, scale).astype(np.float32)
        t = np.rint((xf / scale[:, None]).astype(np.float32))
        t = np.where(np.isnan(t), np.float32(0), t)
        q = np.clip(t, -128.0, 127.0).astype(np.int8)
    return q, scale


def dequantize(q: np.ndarray, scale: np.ndarray, reduce_axis: int, out_dtype):
    with np.errstate(invalid="ignore", over="ignore"):
        t = q.astype(np.float32) * np.expand_dims(scale.astype(np.float32), reduce_axis)
    return from_f32(t.astype(np.float32), out_dtype)


def gemm_s8s8s32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """acc[m,n] = sum_k a[m,k]*b[n,k], exact (int32 wraps like the hardware would; never reached
    for K <= 131071)."""
    if a.shape[1] == 0:
        return np.zeros((a.shape[0], b.shape[0]), np.int32)
    # float64 BLAS is exact here: |acc| <= 128*128*K < 2^53 for any practical K.
    acc = a.astype(np.float64) @ b.astype(np.float64).T
    return acc.astype(np.int64).astype(np.int32)


def epilogue(acc: np.ndarray, xs: np.ndarray, ws: np.ndarray, bias, out_dtype):
    """QSPEC E1-E4: (f32(acc)*xs[m])*ws[n] (+f32(bias[n])), each op rounded separately."""
    with np.errstate(invalid="ignore", over="ignore"):
        t = acc.astype(np.float32)                               # RNE int32 -> f32
        t = (t * xs.astype(np.float32)[:, None]).astype(np.float32)
        t = (t * ws.astype(np.float32)[None, :]).astype(np.float32)
        if bias is not None:
            t = (t + to_f32(bias, out_dtype)[None, :]).astype(np.float32)
    return from_f32(t, out_dtype)


def qlinear(x, dtype, wq, ws, bias=None):
    xq, xs = quantize(x, dtype, 1)
    acc = gemm_s8s8s32(xq, wq)
    y = epilogue(acc, xs, ws, bias, dtype)
    return y, xq, xs, acc


# ---------------------------------------------------------------- QSPEC S1-S6: silu(g)*u -> per-token quantisation
def _bits(u: int) -> np.float32:
    return np.array([u], np.uint32).view(np.float32)[0]


def fma32(a, b, c) -> np.ndarray:
    """Correctly rounded binary32 fma(a, b, c) without a hardware fma: the product is exact in binary64, the sum is
    rounded to 53 bits with its error recovered by TwoSum, and the one case where rounding 53 -> 24 bits could go wrong
    (the 53-bit sum sits exactly on a binary32 tie while the true sum does not) is decided by the sign of that error."""
    a = np.asarray(a, np.float32).astype(np.float64); b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        p = a * b
        s = p + c
        bb = s - p
        err = (p - (s - bb)) + (c - bb)
        lo = s.astype(np.float32)
        lo64 = lo.astype(np.float64)
        toward = np.where(s > lo64, np.float32(np.inf), np.float32(-np.inf))
        nb = np.nextafter(lo, toward)
        tie = np.isfinite(s) & (s != lo64) & (np.abs(s - lo64) == np.abs(s - nb.astype(np.float64))) & (err != 0)
        pick_nb = tie & (np.sign(err) == np.sign(nb.astype(np.float64) - lo64))
    return np.where(pick_nb, nb, lo).astype(np.float32)


def exp_spec(t) -> np.ndarray:
    """QSPEC S1-S4 (oracle/qspec_oracle.c::oq_exp_spec)."""
    t = np.asarray(t, np.float32)
    with np.errstate(invalid="ignore", over="ignore", under="ignore"):
        tc = np.where(t < np.float32(-30), np.float32(-30), t)
        tc = np.where(tc > np.float32(100), np.float32(100), tc).astype(np.float32)
        n = np.rint(tc * _bits(0x3FB8AA3B)).astype(np.float32)
        r = fma32(n, -_bits(0x3F317200), tc)
        r = fma32(n, -_bits(0x35BFBE8E), r)
        p = np.full(t.shape, _bits(0x39500D01), np.float32)
        for cbits in (0x3AB60B61, 0x3C088889, 0x3D2AAAAB, 0x3E2AAAAB, 0x3F000000, 0x3F800000, 0x3F800000):
            p = fma32(p, r, np.full(t.shape, _bits(cbits), np.float32))
        ni = np.where(np.isnan(n), 0, n).astype(np.int32)
        n1 = ni >> 1
        n2 = ni - n1
        s1 = ((n1 + 127).astype(np.uint32) << np.uint32(23)).view(np.float32)
        s2 = ((n2 + 127).astype(np.uint32) << np.uint32(23)).view(np.float32)
        return ((p * s1).astype(np.float32) * s2).astype(np.float32)


def silu_mul(g: np.ndarray, u: np.ndarray, dtype) -> np.ndarray:
    """QSPEC S5: h = cast(f32(cast(g / (1 + exp_spec(-g)))) * f32(u)), stored dtype out."""
    gf, uf = to_f32(g, dtype), to_f32(u, dtype)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        d = (np.float32(1) + exp_spec(-gf)).astype(np.float32)
        sg = to_f32(from_f32((gf / d).astype(np.float32), dtype), dtype)
        return from_f32((sg * uf).astype(np.float32), dtype)


def silu_mul_quantize(g: np.ndarray, u: np.ndarray, dtype):
    """QSPEC S6: per-token quantisation of silu_mul(g, u).  Returns (q int8, scale f32, h stored dtype)."""
    h = silu_mul(g, u, dtype)
    q, s = quantize(h, dtype, 1)
    return q, s, h


# ---------------------------------------------------------------- QSPEC N1-N6: RMSNorm -> per-token quantisation
def rms_sumsq(xf: np.ndarray, epv: int) -> np.ndarray:
    """N1-N3: the pinned-order sum of squares of each row of xf (float32 [rows, cols])."""
    rows, cols = xf.shape
    nvec = (cols + epv - 1) // epv
    slots = (nvec + 255) // 256                       # vectors per lane
    pad = np.zeros((rows, slots * 256 * epv), np.float32)
    pad[:, :cols] = xf                                # zero padding adds fma(0, 0, acc) = acc: no effect
    v = pad.reshape(rows, slots, 256, epv)            # [row, i, lane, e]: vector i*256 + lane
    acc = np.zeros((rows, 256), np.float32)
    for i in range(slots):
        for e in range(epv):
            acc = fma32(v[:, i, :, e], v[:, i, :, e], acc)
    s = acc.reshape(rows, 4, 64)
    lanes = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        s = (s + s[:, :, lanes ^ off]).astype(np.float32)
    g = s[:, :, 0]
    return (((g[:, 0] + g[:, 1]).astype(np.float32) + g[:, 2]).astype(np.float32) + g[:, 3]).astype(np.float32)


def rmsnorm_quantize(x: np.ndarray, weight: np.ndarray, eps: float, dtype):
    """QSPEC N1-N6.  Returns (q int8, scale f32, h stored dtype, rs f32)."""
    d = dt(dtype)
    xf, wf = to_f32(x, d), to_f32(weight, d)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        ss = rms_sumsq(xf, 4 if d == DT_F32 else 8)
        var = (ss / np.float32(xf.shape[1])).astype(np.float32)
        rs = (np.float32(1) / np.sqrt((var + np.float32(eps)).astype(np.float32)).astype(np.float32)).astype(np.float32)
        xn = to_f32(from_f32((xf * rs[:, None]).astype(np.float32), d), d)
        h = from_f32((wf[None, :] * xn).astype(np.float32), d)
    q, s = quantize(h, d, 1)
    return q, s, h, rs
