"""Dev tool (GPU box): K2 (pq_quant_colwise) from hipGraph replays, by shape, optionally sweeping the workgroup-count targets of its two passes
(PQ_K2_BLOCKS_A / PQ_K2_BLOCKS_E).  usage: python tools/k2_bench.py [--sweep]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import protoquant_amd as pq  # noqa: F401
from protoquant_amd import _lib as L
lib = L.lib()


def ev(fn, n=20, reps=9):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


st = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
sweep = [(0, 0)] + ([(a, e) for a in (256, 512, 1024, 2048) for e in (512, 1024, 2048, 4096)] if "--sweep" in sys.argv else [])
for rows, cols in ((4096, 4096), (11008, 4096), (4096, 14336), (28672, 4096), (8192, 1024)):
    x = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    q = torch.empty((rows, cols), dtype=torch.int8, device="cuda"); s = torch.empty(cols, device="cuda"); sr = torch.empty(rows, device="cuda")
    t1 = ev(lambda: L.check(lib.pq_quant_rowwise(x.data_ptr(), 0, rows, cols, cols, q.data_ptr(), cols, sr.data_ptr(), st()), "k1"))
    best = None
    for a, e in sweep:
        L.set_option("PQ_K2_BLOCKS_A", str(a) if a else ""); L.set_option("PQ_K2_BLOCKS_E", str(e) if e else "")
        t = ev(lambda: L.check(lib.pq_quant_colwise(x.data_ptr(), 0, rows, cols, cols, q.data_ptr(), cols, s.data_ptr(), st()), "k2"))
        if a == 0:
            print(f"K2 {rows}x{cols}: {t:6.1f} us ({rows * cols * 5 / t / 1e6:.2f} TB/s over 2 reads + 1 write)   K1 of the same matrix: {t1:.1f} us", flush=True)
        elif best is None or t < best[0]:
            best = (t, a, e)
        if a and "--all" in sys.argv:
            print(f"      A={a} E={e}: {t:.1f}")
    if best:
        print(f"      best of the sweep: {best[0]:.1f} us at PQ_K2_BLOCKS_A={best[1]} PQ_K2_BLOCKS_E={best[2]}")
