"""TEST INFRASTRUCTURE — how far is QSPEC N1-N6 (RMSNorm fused into the activation quantisation) from the eager chain a model runs?

HF's LlamaRMSNorm.forward is   xf = x.float(); var = xf.pow(2).mean(-1, keepdim=True); xn = (xf * rsqrt(var + eps)).to(dtype); h = weight * xn
and the next linear then quantises h per token.  N1-N6 pin ONE summation order for the mean of squares (a float sum has no value without an order)
and IEEE 1/sqrt; torch's `mean` uses its own (vectorised, build- and device-dependent) order, so var can differ in the last bit, rs with it, and a
stored bf16/fp16 h flips where f32(x)*rs sits within that distance of a rounding boundary.  This script measures the rate on >= 10^8 elements per
configuration (C oracle, oracle/qspec_oracle.c, against torch CPU eager in this container) and prints a small table; the numbers are quoted in
INTEGRATION.md §4 and DESIGN.md §2.  Run:  python oracle/measure_rmsnorm_vs_eager.py  [--elements 1e8]  > profiles/r05_rmsnorm_vs_eager.txt"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import c_oracle as C  # noqa: E402
from oracle import torch_ref as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elements", type=float, default=1e8)
    ap.add_argument("--block-rows", type=int, default=2048)
    args = ap.parse_args()
    print(f"# oracle/measure_rmsnorm_vs_eager.py  torch {torch.__version__} CPU ({torch.get_num_threads()} threads), eps 1e-5, weight = 1 + 0.1 N(0,1), x = s * N(0,1) per row with s log-uniform in [0.05, 20]")
    print("# columns: dtype H rows elements | var differs (rows) | h elements differing (count, rate) | rows with any differing h | codes differing (count, rate) | scales differing (rows) | max |dh| in storage ulps")
    for code, td, name in ((0, torch.bfloat16, "bf16"), (1, torch.float16, "fp16")):
        for H in (4096, 8192):
            rows_total = int(-(-args.elements // H))
            g = torch.Generator().manual_seed(100 + H + code)
            w = (1 + 0.1 * torch.randn(H, generator=g)).to(td)
            wb = w.view(torch.int16).numpy().view(np.uint16)
            n = dh = dq = ds = drow = dvar = 0
            max_ulp = 0
            t0 = time.time()
            done = 0
            while done < rows_total:
                r = min(args.block_rows, rows_total - done)
                scale = torch.exp(torch.empty(r, 1).uniform_(np.log(0.05), np.log(20.0), generator=g))
                x = (torch.randn(r, H, generator=g) * scale).to(td)
                xb = x.view(torch.int16).numpy().view(np.uint16)
                q, s, h, rs = C.rmsnorm_quant_rowwise(xb, wb, 1e-5, code)
                # the eager chain (HF LlamaRMSNorm), then QSPEC's quantize on its output
                xf = x.float()
                var = xf.pow(2).mean(-1, keepdim=True)
                rs_t = torch.rsqrt(var + 1e-5)
                h_t = w * (xf * rs_t).to(td)
                q_t, s_t = R.quantize_ref(h_t, 1)
                hb_t = h_t.view(torch.int16).numpy()
                hb = h.view(np.int16)
                diff = hb != hb_t
                dh += int(diff.sum()); drow += int(diff.any(axis=1).sum())
                dq += int((q != q_t.numpy()).sum()); ds += int((s.view(np.uint32) != s_t.numpy().view(np.uint32)).sum())
                dvar += int((rs.view(np.uint32) != rs_t[:, 0].numpy().view(np.uint32)).sum())
                if diff.any():       # same-sign neighbours in a 16-bit float format differ by 1 in the bit pattern per ulp
                    max_ulp = max(max_ulp, int(np.abs(hb[diff].astype(np.int32) - hb_t[diff].astype(np.int32)).max()))
                n += r * H; done += r
            print(f"{name} {H} {rows_total} {n} | {dvar} | {dh} {dh / n:.3e} | {drow} | {dq} {dq / n:.3e} | {ds} | {max_ulp}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
