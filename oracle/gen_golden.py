"""TEST INFRASTRUCTURE — generates tests/golden/*.npz with oracle/torch_ref.py (QSPEC around
``torch._int_mm``, CPU).  Run in the builder container:  ``python oracle/gen_golden.py``.

The reference mount has no code to import (``/root/reference/CODE_OF_CONDUCT.md:1-80`` only), so these
fixtures are outputs of the *contract-named primitive* (``torch._int_mm``) plus QSPEC float stages,
not of the reference itself — "parity unpinned" (oracle/README.md).  Fixtures are data only: inputs
and expected outputs; half tensors are stored as uint16 bit patterns.

Cases: ragged 5x7x3 and 17x16x8, BASELINE config-1 shape 32x512x512 (fp32 and
bf16 inputs), one 256x384x512 bf16 with bias, an fp16 case, an outlier case, a zero-row case, and four
special-value cases (NaN, +-Inf, signalling NaN, subnormals, -0.0, all-zero and all-NaN rows; NaN / Inf weight channels).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_ref as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
TORCH_DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    return t.contiguous().numpy().copy()


def plant_specials(x: torch.Tensor, dtype: str):
    """QSPEC v2 special values, planted into a random [M >= 12, K >= 8] matrix (rows named by what they hold):
    1 one NaN · 2 one +Inf · 3 all zero · 4 -Inf and NaN · 5 all subnormal · 6 a signalling NaN · 7 all NaN ·
    8 -0.0 and the largest finite value · 9 one subnormal among zeros · column 7 holds a NaN in row 10 only (so the
    per-channel quantisation has NaN columns 3, 1, 2 (row 6's), 7 and all of row 7's)."""
    td = TORCH_DT[dtype]
    tiny = {"bf16": 1e-40, "fp16": 6e-8, "f32": 1e-41}[dtype]
    big = {"bf16": 3.3895e38, "fp16": 65504.0, "f32": 3.4028234e38}[dtype]
    x[1, 3] = float("nan")
    x[2, 5] = float("inf")
    x[3, :] = 0
    x[4, 0] = float("-inf"); x[4, 1] = float("nan")
    x[5, :] = tiny
    x[7, :] = float("nan")
    x[8, 0] = -0.0; x[8, 4] = big
    x[9, :] = 0; x[9, 6] = tiny
    x[10, 7] = float("nan")
    if td is torch.float32:
        x.view(torch.int32)[6, 2] = 0x7F800001                       # signalling NaN bit patterns
    else:
        x.view(torch.int16)[6, 2] = 0x7F81 if td is torch.bfloat16 else 0x7C01
    return x


def make_case(name, M, N, K, dtype, seed, bias=False, outliers=False, zero_rows=False, wscale=0.02, specials=False):
    g = torch.Generator().manual_seed(seed)
    td = TORCH_DT[dtype]
    x = torch.randn(M, K, generator=g).to(td)
    w = (torch.randn(N, K, generator=g) * wscale).to(td)
    b = (torch.randn(N, generator=g) * 0.01).to(td) if bias else None
    if specials:
        x = plant_specials(x, dtype)
        w[N - 2, 1] = float("nan")                # a NaN weight channel: its scale is NaN, its output column NaN
        w[N - 3, 0] = float("inf")
    if outliers:                                  # 1 % of the x columns x20
        cols = torch.randperm(K, generator=g)[: max(1, K // 100)]
        x[:, cols] = (x[:, cols].float() * 20).to(td)
    if zero_rows:
        x[0] = 0
        x[M // 2] = 0
        w[N - 1] = 0
    wq, ws = R.quantize_ref(w, 1)                 # per-channel weight: one scale per output row of W[N,K]
    y, xq, xs, acc = R.qlinear_ref(x, wq, ws, b)
    # column-wise quantisation of x as an [M,K] matrix (reduce over rows) for the K2 kernel
    cq, cs = R.quantize_ref(x, 0)
    # exactness cross-check of the contract primitive against int64 matmul (SURVEY PROBE-4)
    acc64 = xq.to(torch.int64) @ wq.to(torch.int64).t()
    assert torch.equal(acc64, acc.to(torch.int64)), name
    d = dict(
        dtype=np.array(dtype), seed=np.array(seed), M=np.array(M), N=np.array(N), K=np.array(K),
        x=bits(x), w=bits(w), xq=xq.numpy(), xs=xs.numpy(), wq=wq.numpy(), ws=ws.numpy(),
        acc=acc.numpy(), y=bits(y), x_colq=cq.numpy(), x_cols=cs.numpy(),
        x_deq=bits(R.dequantize_ref(xq, xs, 1, td)), x_coldeq=bits(R.dequantize_ref(cq, cs, 0, td)),
    )
    if b is not None:
        d["bias"] = bits(b)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name}: M={M} N={N} K={K} {dtype} acc[{int(acc.min())},{int(acc.max())}]")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    make_case("ragged_5x7x3_f32", 5, 7, 3, "f32", 11)
    make_case("ragged_17x16x8_bf16", 17, 16, 8, "bf16", 12, bias=True)
    make_case("cfg1_32x512x512_f32", 32, 512, 512, "f32", 1234)
    make_case("cfg1_32x512x512_bf16", 32, 512, 512, "bf16", 1234)
    make_case("mid_256x384x512_bf16_bias", 256, 384, 512, "bf16", 13, bias=True)
    make_case("mid_64x128x256_fp16_bias", 64, 128, 256, "fp16", 14, bias=True)
    make_case("outlier_48x80x200_bf16", 48, 80, 200, "bf16", 15, outliers=True)
    make_case("zerorow_33x65x129_bf16", 33, 65, 129, "bf16", 16, zero_rows=True, bias=True)
    # QSPEC v2 NaN / Inf / sNaN / subnormal / all-zero policy, generated by the torch form (float outputs hold NaNs: tests
    # compare those as a class, everything else bit for bit)
    make_case("special_12x24x40_bf16", 12, 24, 40, "bf16", 21, bias=True, specials=True)
    make_case("special_12x24x40_fp16", 12, 24, 40, "fp16", 22, specials=True)
    make_case("special_13x9x37_f32", 13, 9, 37, "f32", 23, bias=True, specials=True)
    make_case("special_16x128x256_bf16", 16, 128, 256, "bf16", 24, specials=True)     # vector-path widths


if __name__ == "__main__":
    main()
