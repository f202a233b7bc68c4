"""Dev tool (GPU box): the vendor library on the same box — `torch._int_mm` (hipBLASLt / rocBLAS int8 GEMM, int32 out) and a bf16
`torch.mm` — next to this library's int32-out twin (pq_gemm_s8s8s32) and fused bf16-out qlinear, hipGraph replays, interleaved.
usage: python tools/vendor_gemm.py [--shapes MxNxK,...]"""
import argparse
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import protoquant_amd as pq  # noqa: E402
from protoquant_amd import qlinear as Q  # noqa: E402


def graph_of(fn, n):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096x4096,4096x14336x4096,4096x4096x14336,8192x8192x8192,4096x1024x4096,2048x4096x11008")
    ap.add_argument("--rounds", type=int, default=11)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import sys as _s
    qmod = _s.modules["protoquant_amd.qlinear"]
    for shp in a.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        torch.manual_seed(0)
        xq = (torch.randn(M, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        wq = (torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        xs = torch.rand(M, device=dev) * 1e-2 + 1e-3
        ws = torch.rand(N, device=dev) * 1e-2 + 1e-3
        wt = wq.t()                       # [K, N] view, column-major: what torch._int_mm(x, w.t()) gets in the reference's call
        wkn = wq.t().contiguous()         # [K, N] row-major
        xb, wb = xq.to(torch.bfloat16), wq.to(torch.bfloat16)
        cands = {}
        cands["pq int32-out (pq_gemm_s8s8s32)"] = lambda: qmod.int_mm(xq, wq)
        cands["pq fused bf16-out (pq_qlinear_s8)"] = lambda: qmod.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16)
        try:
            torch._int_mm(xq, wt); cands["torch._int_mm(x, w.t())  [vendor int8, TN]"] = lambda: torch._int_mm(xq, wt)
        except Exception as e:  # noqa: BLE001
            print("torch._int_mm TN unavailable:", str(e)[:200])
        try:
            torch._int_mm(xq, wkn); cands["torch._int_mm(x, w_kn)   [vendor int8, NN]"] = lambda: torch._int_mm(xq, wkn)
        except Exception as e:  # noqa: BLE001
            print("torch._int_mm NN unavailable:", str(e)[:200])
        cands["torch.mm bf16 (x, w.t())   [vendor bf16]"] = lambda: torch.mm(xb, wb.t())
        ref = qmod.int_mm(xq, wq)
        for k in list(cands):
            if "_int_mm" in k:
                assert torch.equal(cands[k](), ref), k
        per = 10
        graphs = {k: graph_of(f, per) for k, f in cands.items()}
        t0 = time.time()
        while time.time() - t0 < 1.0:
            for g in graphs.values():
                g.replay()
        torch.cuda.synchronize()
        times = {k: [] for k in cands}
        for _ in range(a.rounds):
            for k, g in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / (2 * per))
        ops = 2.0 * M * N * K
        for k, t in times.items():
            t = sorted(t); med = t[len(t) // 2]
            peak = 2516.5 if "bf16 (" in k else 5033.0
            print(f"{shp:>18s}  {k:46s} median {med:8.2f} us  {ops / med / 1e6:7.1f} T(FL)OPS  {100 * ops / med / 1e6 / peak:5.1f} % of {peak:.0f}", flush=True)


if __name__ == "__main__":
    main()
