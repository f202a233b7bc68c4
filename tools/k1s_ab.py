"""Dev tool (GPU box): K1s (pq_silu_mul_quant_rowwise) on wide rows, 256 vs 512 threads per row (pq_set_option PQ_SILU_TPR), interleaved
rounds of hipGraph replays; checks the codes and scales are identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoquant_amd import _lib as L
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for (R, C) in ((4096, 14336), (2048, 11008), (4096, 16384), (4096, 8192 + 8), (1024, 14336)):
    gu = (torch.randn(R, 2 * C) * 2).to(torch.bfloat16).cuda()
    outs, graphs = {}, {}
    for tpr in ("256", ""):
        L.set_option("PQ_SILU_TPR", tpr)
        q = torch.empty((R, C), dtype=torch.int8, device="cuda"); s = torch.empty(R, device="cuda")
        f = lambda: L.check(lib.pq_silu_mul_quant_rowwise(gu.data_ptr(), 2 * C, gu.data_ptr() + 2 * C, 2 * C, 0, R, C, q.data_ptr(), C, s.data_ptr(), None, C, st()), "k1s")
        f(); torch.cuda.synchronize()
        outs[tpr] = (q.clone(), s.clone())
        s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            f()
        torch.cuda.current_stream().wait_stream(s2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                f()
        graphs[tpr] = g
    same = torch.equal(outs["256"][0], outs[""][0]) and torch.equal(outs["256"][1], outs[""][1])
    for g in graphs.values():
        for _ in range(10):
            g.replay()
    torch.cuda.synchronize()
    t = {"256": [], "": []}
    for r in range(15):
        for tpr in ("256", ""):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graphs[tpr].replay(); b.record(); b.synchronize()
            t[tpr].append(a.elapsed_time(b) * 1e3 / 10)
    by = 5 * R * C + 4 * R
    for tpr in ("256", ""):
        v = sorted(t[tpr]); med = v[len(v) // 2]
        print(f"K1s {R}x{C} bf16 threads/row={tpr or '512 (auto)'}: same={same} median {med:7.2f} us min {v[0]:7.2f} us  {by / med / 1e6:.2f} TB/s algorithmic", flush=True)
L.set_option("PQ_SILU_TPR", "")
