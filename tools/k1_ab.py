"""Dev tool (GPU box): K1 (pq_quant_rowwise) with one vs two rows per wave (pq_set_option PQ_K1_RPW), interleaved rounds of
hipGraph replays; checks the codes and scales are identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoquant_amd import _lib as L
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for (R, C) in ((4096, 4096), (4096, 8192), (16384, 4096), (2048, 4096), (8192, 2048), (4096, 1024)):
    x = torch.randn(R, C).to(torch.bfloat16).cuda()
    outs, graphs = {}, {}
    for rpw in ("1", "2"):
        L.set_option("PQ_K1_RPW", rpw)
        q = torch.empty((R, C), dtype=torch.int8, device="cuda"); s = torch.empty(R, device="cuda")
        f = lambda: L.check(lib.pq_quant_rowwise(x.data_ptr(), 0, R, C, C, q.data_ptr(), C, s.data_ptr(), st()), "k1")
        f(); torch.cuda.synchronize()
        outs[rpw] = (q.clone(), s.clone())
        s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            f()
        torch.cuda.current_stream().wait_stream(s2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                f()
        graphs[rpw] = g
    same = torch.equal(outs["1"][0], outs["2"][0]) and torch.equal(outs["1"][1], outs["2"][1])
    for g in graphs.values():
        for _ in range(20):
            g.replay()
    torch.cuda.synchronize()
    t = {"1": [], "2": []}
    for r in range(15):
        for rpw in ("1", "2"):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graphs[rpw].replay(); graphs[rpw].replay(); b.record(); b.synchronize()
            t[rpw].append(a.elapsed_time(b) * 1e3 / 40)
    by = 3 * R * C + 4 * R
    for rpw in ("1", "2"):
        v = sorted(t[rpw]); med = v[len(v) // 2]
        print(f"K1 {R}x{C} bf16 rows/wave={rpw}: same={same} median {med:7.2f} us min {v[0]:7.2f} us  {by / med / 1e6:.2f} TB/s ({by / med / 1e6 / 8 * 100:.1f} % of 8 TB/s)", flush=True)
L.set_option("PQ_K1_RPW", "")
