"""Dev tool: time every kernel variant at the headline shape, check the fast variants against the
generic one.  Usage (GPU box): python tools/quick_bench.py [M N K]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import protoquant_amd as pq  # noqa: E402
from protoquant_amd import _lib as _pqlib  # noqa: E402


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 4096)
    torch.manual_seed(1234)
    x = torch.randn(M, K).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K) * 0.02).to(torch.bfloat16).cuda()
    qw = pq.quantize(w)
    qx = pq.quantize(x)
    # warm the clocks
    t0 = time.time()
    while time.time() - t0 < 1.0:
        pq.int_mm(qx.int_data, qw.int_data)
    torch.cuda.synchronize()
    med, mn = timeit(lambda: pq.quantize(x))
    byts = M * K * 3 + 4 * M
    print(f"K1 rowquant {M}x{K} bf16: median {med:.1f} us min {mn:.1f} us -> {byts / med / 1e6:.2f} TB/s (min {byts / mn / 1e6:.2f})")
    med, mn = timeit(lambda: pq.quantize(x, axis=0))
    print(f"K2 colquant {M}x{K} bf16: median {med:.1f} us min {mn:.1f} us -> {byts / med / 1e6:.2f} TB/s algorithmic")
    med, mn = timeit(lambda: pq.dequantize(qx))
    print(f"dequant {M}x{K}->bf16: median {med:.1f} us -> {(M * K * 3 + 4 * M) / med / 1e6:.2f} TB/s")
    ops = 2.0 * M * N * K
    _pqlib.set_option("PQ_FORCE_VARIANT", "generic")
    ref_acc = pq.int_mm(qx.int_data, qw.int_data)
    ref_y = pq.qlinear_s8(qx.int_data, qx.scale, qw.int_data, qw.scale, None, torch.bfloat16)
    for v in ("generic", "sp256_16"):
        _pqlib.set_option("PQ_FORCE_VARIANT", v)
        acc = pq.int_mm(qx.int_data, qw.int_data)
        y = pq.qlinear_s8(qx.int_data, qx.scale, qw.int_data, qw.scale, None, torch.bfloat16)
        ok = torch.equal(acc, ref_acc) and torch.equal(y.view(torch.int16), ref_y.view(torch.int16))
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        med, mn = timeit(lambda: pq.qlinear_s8(qx.int_data, qx.scale, qw.int_data, qw.scale, None, torch.bfloat16, out=out),
                         iters=20 if v == "generic" else 100)
        print(f"GEMM+epi {v:9s}: ok={ok} median {med:.1f} us min {mn:.1f} us -> {ops / med / 1e6:.1f} TOPS "
              f"({ops / med / 1e6 / 5033 * 100:.1f}% of 5033) min-> {ops / mn / 1e6:.1f}")
        med, mn = timeit(lambda: pq.int_mm(qx.int_data, qw.int_data), iters=20 if v == "generic" else 50)
        print(f"GEMM s32 {v:9s}: median {med:.1f} us -> {ops / med / 1e6:.1f} TOPS")
    _pqlib.set_option("PQ_FORCE_VARIANT", "")
    lin = pq.qlinear.from_qtensor(qw)
    med, mn = timeit(lambda: lin(x))
    print(f"qlinear fwd (K1+K3/K4 via python): median {med:.1f} us -> {ops / med / 1e6:.1f} TOPS")


if __name__ == "__main__":
    main()
