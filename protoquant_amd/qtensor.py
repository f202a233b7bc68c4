"""QTensor + quantize()/dequantize(): the Python surface BASELINE.json's north_star names.

The reference's own definitions are absent from the mount (/root/reference holds only
CODE_OF_CONDUCT.md:1-80), so signatures are build-defined and numerics follow
QSPEC v2.  All arithmetic happens in libpq_hip.so (HIP, gfx950)."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib as L


@dataclass
class QTensor:
    """Dynamic symmetric int8 tensor: `int_data` (int8, original shape), `scale` (fp32, one per
    kept index), `axis` = the reduced (quantised-over) axis of the 2-D view, `orig_dtype`."""
    int_data: torch.Tensor
    scale: torch.Tensor
    axis: int
    orig_dtype: torch.dtype
    shape: torch.Size

    def dequantize(self, dtype: torch.dtype | None = None) -> torch.Tensor:
        return dequantize(self, dtype)

    @property
    def device(self):
        return self.int_data.device

    def to(self, device) -> "QTensor":
        return QTensor(self.int_data.to(device), self.scale.to(device), self.axis, self.orig_dtype, self.shape)

    def __repr__(self):
        return (f"QTensor(shape={tuple(self.shape)}, axis={self.axis}, orig_dtype={self.orig_dtype}, "
                f"device={self.int_data.device})")


def _as_2d(x: torch.Tensor, axis: int):
    """Collapse x to the 2-D matrix whose reduced axis is `axis` (last -> rows x cols reduce over
    cols; first of a 2-D tensor -> reduce over rows)."""
    nd = x.dim()
    if nd == 0:
        raise ValueError("quantize() needs at least 1 dimension")
    a = axis % nd
    if a == nd - 1:
        lead = 1
        for d in x.shape[:-1]:
            lead *= d
        return x.reshape(lead, x.shape[-1]), 1
    if nd == 2 and a == 0:
        return x, 0
    raise ValueError("quantize(): axis must be the last axis, or axis 0 of a 2-D tensor")


def quantize(x: torch.Tensor, axis: int = -1) -> QTensor:
    """Dynamic symmetric int8 quantisation. axis=-1: per-token (one scale per row of the flattened
    [..., C] tensor, kernel K1); axis=0 on a 2-D tensor: per-channel along the strided axis (K2)."""
    L.require_gpu(x, "quantize(x)")
    code = L.dtype_code(x.dtype)
    x2, red = _as_2d(x, axis)
    x2 = L.row_major_2d(x2)
    rows, cols = x2.shape
    q = torch.empty((rows, cols), dtype=torch.int8, device=x.device)
    scale = torch.empty((rows if red == 1 else cols,), dtype=torch.float32, device=x.device)
    fn = L.lib().pq_quant_rowwise if red == 1 else L.lib().pq_quant_colwise
    with torch.cuda.device(x.device):
        L.check(fn(x2.data_ptr(), code, rows, cols, L.ld(x2), q.data_ptr(), max(cols, 1), scale.data_ptr(),
                   L.stream_ptr(x)), "quantize")
    return QTensor(q.reshape(x.shape), scale, red, x.dtype, x.shape)


def _rows_view(t: torch.Tensor) -> torch.Tensor:
    """[..., C] -> a 2-D [rows, C] view with unit column stride (column slices of a wider matrix stay views)."""
    lead = 1
    for d in t.shape[:-1]:
        lead *= d
    t2 = t if t.dim() == 2 else t.reshape(lead, t.shape[-1])
    return L.row_major_2d(t2)


def silu_mul_quantize(g: torch.Tensor, u: torch.Tensor, return_h: bool = False):
    """quantize(F.silu(g) * u, axis=-1) in ONE pass (kernel K1 fused into its producer): the activation of a gated
    MLP's down projection is computed, reduced and encoded in registers and never goes to HBM in bf16.
    g and u may be column slices of one fused gate+up output.  Numerics: QSPEC S1-S6 then Q1-Q6.
    return_h=True also returns h = silu(g)*u in the input dtype (stored by the same kernel)."""
    L.require_gpu(g, "silu_mul_quantize(g)")
    L.require_gpu(u, "silu_mul_quantize(u)")
    if g.shape != u.shape or g.dtype != u.dtype or g.device != u.device or g.dim() < 1:
        raise ValueError(f"silu_mul_quantize: g {tuple(g.shape)} {g.dtype} and u {tuple(u.shape)} {u.dtype} must match")
    code = L.dtype_code(g.dtype)
    g2, u2 = _rows_view(g), _rows_view(u)
    rows, cols = g2.shape
    q = torch.empty((rows, cols), dtype=torch.int8, device=g.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=g.device)
    h = torch.empty((rows, cols), dtype=g.dtype, device=g.device) if return_h else None
    with torch.cuda.device(g.device):
        L.check(L.lib().pq_silu_mul_quant_rowwise(g2.data_ptr(), L.ld(g2), u2.data_ptr(), L.ld(u2), code, rows, cols,
                                                  q.data_ptr(), max(cols, 1), scale.data_ptr(),
                                                  h.data_ptr() if return_h else None, max(cols, 1), L.stream_ptr(g)),
                "silu_mul_quantize")
    qt = QTensor(q.reshape(g.shape), scale, 1, g.dtype, g.shape)
    return (qt, h.reshape(g.shape)) if return_h else qt


def _silu_pair(g, u, what):
    L.require_gpu(g, f"{what}(g)")
    L.require_gpu(u, f"{what}(u)")
    if g.shape != u.shape or g.dtype != u.dtype or g.device != u.device or g.dim() < 1:
        raise ValueError(f"{what}: g {tuple(g.shape)} {g.dtype} and u {tuple(u.shape)} {u.dtype} must match")
    return L.dtype_code(g.dtype), _rows_view(g), _rows_view(u)


def silu_mul_rowamax(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """First half of silu_mul_quantize for a column-sharded intermediate: per token, the f32 BIT PATTERN (int32 tensor [rows]) of max |silu(g)*u| over the
    columns this rank holds.  Non-negative floats (and NaNs, above +Inf) order as integers: an integer MAX over the ranks is the exact row amax."""
    code, g2, u2 = _silu_pair(g, u, "silu_mul_rowamax")
    rows, cols = g2.shape
    amax = torch.empty((rows,), dtype=torch.int32, device=g.device)
    with torch.cuda.device(g.device):
        L.check(L.lib().pq_silu_mul_rowamax(g2.data_ptr(), L.ld(g2), u2.data_ptr(), L.ld(u2), code, rows, cols, amax.data_ptr(), L.stream_ptr(g)), "silu_mul_rowamax")
    return amax


def silu_mul_quantize_with_amax(g: torch.Tensor, u: torch.Tensor, amax_bits: torch.Tensor, out: torch.Tensor | None = None) -> QTensor:
    """Second half: the int8 codes of THESE columns against the row amax given as f32 bit patterns (int32 [rows], the max over every rank's columns), and the
    row scales amax / 127.  Column block and scale vector of the unsharded silu_mul_quantize, bit for bit.  `out`: optional int8 [rows, cols] destination."""
    code, g2, u2 = _silu_pair(g, u, "silu_mul_quantize_with_amax")
    rows, cols = g2.shape
    if amax_bits.dtype != torch.int32 or amax_bits.shape != (rows,) or amax_bits.device != g.device or not amax_bits.is_contiguous():
        raise ValueError(f"silu_mul_quantize_with_amax: amax_bits must be a contiguous int32 [{rows}] tensor on {g.device}")
    if out is None:
        q = torch.empty((rows, cols), dtype=torch.int8, device=g.device)
    else:
        q = out
        if q.dtype != torch.int8 or q.shape != (rows, cols) or q.device != g.device or (cols > 1 and q.stride(1) != 1):
            raise ValueError(f"silu_mul_quantize_with_amax: out must be an int8 [{rows}, {cols}] tensor with contiguous rows")
    scale = torch.empty((rows,), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        L.check(L.lib().pq_silu_mul_quant_rowwise_amax(g2.data_ptr(), L.ld(g2), u2.data_ptr(), L.ld(u2), code, rows, cols, amax_bits.data_ptr(),
                                                       q.data_ptr(), L.ld(q) if rows > 1 else max(cols, 1), scale.data_ptr(), L.stream_ptr(g)), "silu_mul_quantize_with_amax")
    return QTensor(q if out is not None else q.reshape(g.shape), scale, 1, g.dtype, g.shape)


def rowamax(x: torch.Tensor) -> torch.Tensor:
    """First half of quantize(x, axis=-1) for an activation whose COLUMNS are sharded over ranks: per token, the f32 bit pattern (int32 [rows]) of max |x| over
    the columns this rank holds; an integer MAX over the ranks is the exact, NaN-propagating row amax."""
    L.require_gpu(x, "rowamax(x)")
    code = L.dtype_code(x.dtype)
    x2 = _rows_view(x)
    rows, cols = x2.shape
    amax = torch.empty((rows,), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().pq_quant_rowamax(x2.data_ptr(), code, rows, cols, L.ld(x2), amax.data_ptr(), L.stream_ptr(x)), "rowamax")
    return amax


def quantize_with_amax(x: torch.Tensor, amax_bits: torch.Tensor, out: torch.Tensor | None = None) -> QTensor:
    """Second half: the int8 codes of THESE columns against the row amax given as f32 bit patterns (int32 [rows]: the max over every rank's columns) and the
    row scales — the column block and the scale vector of quantize(x_whole, axis=-1), bit for bit."""
    L.require_gpu(x, "quantize_with_amax(x)")
    code = L.dtype_code(x.dtype)
    x2 = _rows_view(x)
    rows, cols = x2.shape
    if amax_bits.dtype != torch.int32 or amax_bits.shape != (rows,) or amax_bits.device != x.device or not amax_bits.is_contiguous():
        raise ValueError(f"quantize_with_amax: amax_bits must be a contiguous int32 [{rows}] tensor on {x.device}")
    if out is None:
        q = torch.empty((rows, cols), dtype=torch.int8, device=x.device)
    else:
        q = out
        if q.dtype != torch.int8 or q.shape != (rows, cols) or q.device != x.device or (cols > 1 and q.stride(1) != 1):
            raise ValueError(f"quantize_with_amax: out must be an int8 [{rows}, {cols}] tensor with contiguous rows")
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().pq_quant_rowwise_amax(x2.data_ptr(), code, rows, cols, L.ld(x2), amax_bits.data_ptr(), q.data_ptr(), L.ld(q) if rows > 1 else max(cols, 1),
                                              scale.data_ptr(), L.stream_ptr(x)), "quantize_with_amax")
    return QTensor(q if out is not None else q.reshape(x.shape), scale, 1, x.dtype, x.shape)


def rmsnorm_quantize(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, return_h: bool = False):
    """quantize(weight * (x.float() * rsqrt(mean(x.float()**2, -1) + eps)).to(x.dtype), axis=-1) in ONE pass (kernel K1
    fused into RMSNorm): the normalised activation feeding q/k/v or gate/up is reduced, scaled and encoded in registers.
    Numerics: QSPEC N1-N6 (pinned reduction order) then Q1-Q6.  return_h=True also returns the normalised activation."""
    L.require_gpu(x, "rmsnorm_quantize(x)")
    L.require_gpu(weight, "rmsnorm_quantize(weight)")
    if x.dim() < 1 or weight.dim() != 1 or weight.shape[0] != x.shape[-1] or weight.dtype != x.dtype or weight.device != x.device:
        raise ValueError(f"rmsnorm_quantize: x {tuple(x.shape)} {x.dtype} needs a weight of shape ({x.shape[-1] if x.dim() else '?'},) "
                         f"and the same dtype/device, got {tuple(weight.shape)} {weight.dtype}")
    code = L.dtype_code(x.dtype)
    x2 = _rows_view(x)
    w = weight.contiguous()
    rows, cols = x2.shape
    q = torch.empty((rows, cols), dtype=torch.int8, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    h = torch.empty((rows, cols), dtype=x.dtype, device=x.device) if return_h else None
    with torch.cuda.device(x.device):
        L.check(L.lib().pq_rmsnorm_quant_rowwise(x2.data_ptr(), L.ld(x2), w.data_ptr(), float(eps), code, rows, cols, q.data_ptr(),
                                                 max(cols, 1), scale.data_ptr(), h.data_ptr() if return_h else None, max(cols, 1),
                                                 L.stream_ptr(x)), "rmsnorm_quantize")
    qt = QTensor(q.reshape(x.shape), scale, 1, x.dtype, x.shape)
    return (qt, h.reshape(x.shape)) if return_h else qt


def dequantize(q: QTensor, dtype: torch.dtype | None = None) -> torch.Tensor:
    """cast_rne(f32(int_data) * scale) along the kept axis -> `dtype` (default: the original dtype)."""
    dtype = dtype or q.orig_dtype
    L.require_gpu(q.int_data, "dequantize(q)")
    code = L.dtype_code(dtype)
    if q.axis == 1:
        lead = 1
        for d in q.shape[:-1]:
            lead *= d
        d2 = q.int_data.reshape(lead, q.shape[-1])
    else:
        d2 = q.int_data
    d2 = L.row_major_2d(d2)
    rows, cols = d2.shape
    out = torch.empty((rows, cols), dtype=dtype, device=d2.device)
    with torch.cuda.device(d2.device):
        L.check(L.lib().pq_dequant(d2.data_ptr(), L.ld(d2), q.scale.data_ptr(), q.axis, rows, cols, out.data_ptr(),
                                   max(cols, 1), code, L.stream_ptr(d2)), "dequantize")
    return out.reshape(q.shape)
