"""TEST INFRASTRUCTURE — QSPEC v2 as plain torch CPU ops around ``torch._int_mm``.

Parity status: *parity unpinned by the reference* — ``/root/reference`` contains no source for
this path (only ``/root/reference/CODE_OF_CONDUCT.md:1-80``), so nothing here can cite a reference
``file:line``. What is followed instead:

* the integer GEMM is ``torch._int_mm`` itself (``aten::_int_mm(Tensor self, Tensor mat2)``), the
  primitive ``BASELINE.json`` → ``north_star`` names as the CPU oracle;
* every float stage follows QSPEC v2 (``DESIGN.md`` §2), op for op.

This module is what "protoquant's own CPU path" means in ``bench.py``'s ``cpu_baseline`` leg and what
``oracle/gen_golden.py`` runs to produce ``tests/golden/*.npz``.  It must only ever be imported from
``tests/``, ``bench.py`` (cpu_baseline) and ``oracle/gen_golden.py`` — never from ``protoquant_amd``.
"""
from __future__ import annotations

import torch

QMAX = 127.0


def quantize_ref(x: torch.Tensor, reduce_dim: int):
    """QSPEC quantize. ``reduce_dim`` is the axis the amax is taken over (-1/1: per-token rows,
    0: per-channel columns of a row-major matrix).  Returns (int8 codes, fp32 scale vector).

    NaN / Inf (QSPEC v2, Q2/Q3/Q5 — PROPAGATE, the behaviour of the plain ops below): ``amax`` propagates a NaN, so a
    row (column) that holds one gets scale = NaN and all-zero codes, and qlinear's output row is NaN, as it would be for
    the unquantised ``F.linear``; an Inf without a NaN gives scale = Inf and all-zero codes (x/Inf = 0, Inf/Inf = NaN -> 0).
    The two ``where`` lines pin what the plain ops leave open: the NaN scale's payload (canonical quiet NaN 0x7FC00000)
    and the NaN -> int8 cast (undefined in C++; 0 here)."""
    assert x.dim() == 2 and x.device.type == "cpu"
    xf = x.to(torch.float32)                                   # Q1 exact up-conversion
    amax = xf.abs().amax(dim=reduce_dim)                       # Q2 exact; NaN propagates (torch.amax)
    scale = amax / torch.tensor(QMAX, dtype=torch.float32)     # Q3 true fp32 division
    scale = torch.where(amax == 0, torch.ones_like(scale), scale)   # zero guard (QSPEC Q3)
    scale = torch.where(scale != scale, torch.full_like(scale, float("nan")), scale)   # Q3: a NaN scale is THE canonical quiet NaN
    s = scale.unsqueeze(reduce_dim)
    q = torch.round(xf / s)                                    # Q4 true division, half-to-even
    q = torch.where(q != q, torch.zeros_like(q), q)            # Q5 NaN -> 0 (x/NaN, Inf/Inf)
    q = torch.clamp(q, -128.0, 127.0).to(torch.int8)           # Q6
    return q, scale


def dequantize_ref(q: torch.Tensor, scale: torch.Tensor, reduce_dim: int, dtype: torch.dtype):
    """QSPEC dequantize: cast_rne(f32(q) * scale broadcast on the kept axis)."""
    return (q.to(torch.float32) * scale.unsqueeze(reduce_dim)).to(dtype)


def int_gemm_ref(xq: torch.Tensor, wq: torch.Tensor) -> torch.Tensor:
    """acc[m,n] = sum_k xq[m,k] * wq[n,k]  ==  torch._int_mm(xq, wq.t())."""
    return torch._int_mm(xq, wq.t())


def epilogue_ref(acc: torch.Tensor, xs: torch.Tensor, ws: torch.Tensor, bias, dtype: torch.dtype):
    """QSPEC epilogue: (f32(acc) * xs[m]) * ws[n] (+ f32(bias[n])) -> cast_rne(dtype)."""
    t = (acc.to(torch.float32) * xs[:, None]) * ws[None, :]
    if bias is not None:
        t = t + bias.to(torch.float32)[None, :]
    return t.to(dtype)


def qlinear_ref(x: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias=None):
    """Full dynamic-int8 linear for a 2-D activation. Returns (y, xq, xs, acc)."""
    xq, xs = quantize_ref(x, 1)
    acc = int_gemm_ref(xq, wq)
    y = epilogue_ref(acc, xs, ws, bias, x.dtype)
    return y, xq, xs, acc


def silu_mul_ref(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """The eager producer of the gated MLP as a model writes it: ``F.silu(g) * u`` (two ops, two storage roundings).
    For bf16 / fp16 tensors this IS QSPEC S1-S5, bit for bit, on every input: S1-S5's silu equals torch's CPU ``F.silu`` on all
    65 536 patterns of either 16-bit type (tests/test_oracle.py::test_silu_spec_is_torch_eager_on_every_16bit_pattern), and the
    product of two 16-bit floats is exact in binary32, so ``* u`` is one deterministic rounding in both.  For fp32 tensors torch's
    own exp carries its ulp and the stored h may differ in the last bits (bounded in tests/test_oracle.py); fp32 activations are
    outside BASELINE.json's configurations."""
    return torch.nn.functional.silu(g) * u


def silu_mul_quantize_ref(g: torch.Tensor, u: torch.Tensor):
    """quantize(F.silu(g) * u) per token: the torch form of QSPEC S1-S6 for 16-bit activations.  Returns (codes, scales, h)."""
    h = silu_mul_ref(g, u)
    q, s = quantize_ref(h, 1)
    return q, s, h


def rmsnorm_eager_ref(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """HF ``LlamaRMSNorm.forward``, op for op (the eager chain a swapped model would otherwise run before its q/k/v and gate/up
    linears).  NOT the spec: QSPEC N1-N6 pins one summation order for the mean of squares, torch's ``mean`` has its own, so the
    stored activation can differ in the last bit on a small share of elements — measured by oracle/measure_rmsnorm_vs_eager.py
    (profiles/r05_rmsnorm_vs_eager.txt) and bounded in tests/test_oracle.py."""
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    return weight * (xf * torch.rsqrt(var + eps)).to(x.dtype)
