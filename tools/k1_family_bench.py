"""Dev tool: K1 / K1n / K1s kernel times at model shapes, replayed from a hipGraph (no host gaps), with algorithmic TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoquant_amd import _lib as L
lib = L.lib()


def graph_time(fn, per=20, reps=30):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per):
            fn()
    for _ in range(5):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / (per * reps)


st = lambda: torch.cuda.current_stream().cuda_stream
for M, C in ((4096, 4096), (4096, 8192), (2048, 11008), (4096, 14336), (16384, 4096)):
    x = torch.randn(M, 2 * C, device="cuda").to(torch.bfloat16)
    w = torch.ones(C, device="cuda", dtype=torch.bfloat16)
    q = torch.empty((M, C), dtype=torch.int8, device="cuda"); s = torch.empty(M, device="cuda")
    t1 = graph_time(lambda: lib.pq_quant_rowwise(x.data_ptr(), 0, M, C, 2 * C, q.data_ptr(), C, s.data_ptr(), st()))
    tn = graph_time(lambda: lib.pq_rmsnorm_quant_rowwise(x.data_ptr(), 2 * C, w.data_ptr(), 1e-5, 0, M, C, q.data_ptr(), C, s.data_ptr(), None, 0, st()))
    ts = graph_time(lambda: lib.pq_silu_mul_quant_rowwise(x.data_ptr(), 2 * C, x.data_ptr() + 2 * C, 2 * C, 0, M, C, q.data_ptr(), C, s.data_ptr(), None, 0, st()))
    b1, b2 = 3 * M * C + 4 * M, 5 * M * C + 4 * M
    print(f"{M:6d} x {C:6d} bf16   K1 {t1:6.1f} us ({b1 / t1 / 1e6:4.2f} TB/s)   K1n {tn:6.1f} us ({b1 / tn / 1e6:4.2f} TB/s)   K1s {ts:6.1f} us ({b2 / ts / 1e6:4.2f} TB/s)")
