"""TEST INFRASTRUCTURE — generates tests/golden/producer/*.npz: fixtures for the producer-fused quantisations
silu(g)*u -> per-token int8 (QSPEC S1-S6) and RMSNorm(x; weight) -> per-token int8 (QSPEC N1-N6).  Run in the builder container:  ``python oracle/gen_golden_producer.py``.

Expected outputs come from oracle/qspec_numpy.py (binary32 arithmetic with an exactly emulated fma), which shares no code
with oracle/qspec_oracle.c or the HIP kernel; `h_torch` is torch's own eager ``F.silu(g) * u`` on CPU for the same
inputs, kept to bound how far the SPECIFIED exponential is from the stock op (tests allow 1 storage-ulp on a small
fraction of elements).  The reference mount has no code for this path: "parity unpinned" (oracle/README.md).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qspec_numpy as Q  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "producer")
TD = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}
NAMES = {0: "bf16", 1: "fp16", 2: "f32"}


def to_torch(a, code):
    return torch.from_numpy(a.copy()) if code == 2 else torch.from_numpy(a.view(np.int16).copy()).view(TD[code])


def bits(t):
    return t.numpy().copy() if t.dtype == torch.float32 else t.view(torch.int16).numpy().view(np.uint16).copy()


def make(code, rows, cols, seed, gscale):
    rng = np.random.default_rng(seed)
    g = (rng.standard_normal((rows, cols)) * gscale).astype(np.float32)
    u = rng.standard_normal((rows, cols)).astype(np.float32)
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.0, -88.0, 100.0, -100.0, 17.4, -17.4, 30.5, -30.5, 1e-40, -1e-40,
                   6e4, -6e4, 3e38, -3e38], np.float32)
    g[0, :sp.size] = sp                       # special gate values against ordinary u
    u[1, :sp.size] = sp                       # ordinary g against special u
    g[2, :sp.size] = sp; u[2, :sp.size] = sp[::-1]
    g[3] = 0.0                                # an all-zero row: scale 1, codes 0
    g[4] = -60.0                              # silu underflows toward -0: tiny amax
    gs, us = Q.from_f32(g, code), Q.from_f32(u, code)
    q, s, h = Q.silu_mul_quantize(gs, us, code)
    with torch.no_grad():
        ht = bits(torch.nn.functional.silu(to_torch(gs, code)) * to_torch(us, code))
    name = f"silu_mul_{rows}x{cols}_{NAMES[code]}"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), code=np.array(code), g=gs, u=us, q=q, scale=s, h=h, h_torch=ht)
    hf, tf = Q.to_f32(h, code), Q.to_f32(ht, code)
    ok = np.isfinite(hf) & np.isfinite(tf)
    print(f"{name}: {np.count_nonzero(hf[ok] != tf[ok])} of {ok.sum()} finite elements differ from torch eager")


def make_rms(code, rows, cols, seed, eps):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, cols)) * rng.uniform(0.05, 30.0, (rows, 1))).astype(np.float32)
    w = (1.0 + 0.2 * rng.standard_normal(cols)).astype(np.float32)
    x[0] = 0.0                                   # zero row: rs = 1/sqrt(eps), h = 0, scale 1
    if cols >= 8:
        x[1, 3] = np.nan                         # NaN poisons the statistic: every h NaN -> codes 0, scale 1
        x[2, 5] = np.inf
        x[3, :] *= 1e-3
        x[4, 0] = 500.0                          # one outlier dominates the row
    xs, ws = Q.from_f32(x, code), Q.from_f32(w, code)
    q, s, h, rs = Q.rmsnorm_quantize(xs, ws, eps, code)
    xt, wt = to_torch(xs, code), to_torch(ws, code)
    with torch.no_grad():                        # the eager op chain as HF's LlamaRMSNorm writes it
        xf = xt.float()
        ht = bits(wt * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(TD[code]))
    name = f"rmsnorm_{rows}x{cols}_{NAMES[code]}"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), code=np.array(code), eps=np.array(eps, np.float32), x=xs, w=ws, q=q, scale=s, h=h,
                        rs=rs, h_torch=ht)
    hf, tf = Q.to_f32(h, code), Q.to_f32(ht, code)
    ok = np.isfinite(hf) & np.isfinite(tf)
    print(f"{name}: {np.count_nonzero(hf[ok] != tf[ok])} of {ok.sum()} finite elements differ from torch eager")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    make(0, 40, 1000, 21, 3.0)
    make(1, 24, 520, 22, 4.0)
    make(2, 16, 260, 23, 5.0)
    make(0, 9, 11008, 24, 2.0)      # BASELINE config 3's intermediate width
    make_rms(0, 24, 4096, 31, 1e-5)  # Llama hidden size
    make_rms(1, 12, 1000, 32, 1e-6)
    make_rms(2, 10, 333, 33, 1e-6)   # ragged width: generic path
    make_rms(0, 6, 8192, 34, 1e-5)


if __name__ == "__main__":
    main()
