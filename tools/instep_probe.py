"""Dev tool (GPU box): why does K1 take ~9 us right after a GEMM and ~6.4 us replayed alone?  Sequences replayed gap-free from hipGraphs
(10 repetitions per graph), interleaved, medians; the cost of a kernel in a position = difference between two sequences.
  G            : GEMM
  G K(a)       : the step
  G K(a) K(b)  : a second K1 on ANOTHER, equally stale input right behind the first
  G K(a) K(a)  : a second K1 on the SAME input
  K(a) K(b)    : K1 alone, two alternating inputs (96 MB working set)
  G T K(a)     : a tiny kernel (one block) between the GEMM and K1
  S K(a)       : K1 behind a low-power streaming kernel of the GEMM's duration class (the dequant kernel on 64 MB)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protoquant_amd import _lib as LL
L = LL.lib()
M = N = K = 4096
dev = "cuda"
xa = torch.randn(M, K).to(torch.bfloat16).to(dev); xb = torch.randn(M, K).to(torch.bfloat16).to(dev)
qa = torch.empty((M, K), dtype=torch.int8, device=dev); sa = torch.empty(M, device=dev)
qb = torch.empty((M, K), dtype=torch.int8, device=dev); sb = torch.empty(M, device=dev)
wq = (torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8); ws = torch.rand(N, device=dev) * 1e-3
y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
big_q = torch.randint(-127, 127, (8192, 4096), dtype=torch.int8, device=dev); big_s = torch.rand(8192, device=dev); big_o = torch.empty((8192, 4096), dtype=torch.bfloat16, device=dev)
tiny = torch.zeros(64, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
Ka = lambda: LL.check(L.pq_quant_rowwise(xa.data_ptr(), 0, M, K, K, qa.data_ptr(), K, sa.data_ptr(), st()), "k1")
Kb = lambda: LL.check(L.pq_quant_rowwise(xb.data_ptr(), 0, M, K, K, qb.data_ptr(), K, sb.data_ptr(), st()), "k1")
G = lambda: LL.check(L.pq_qlinear_s8(qa.data_ptr(), K, sa.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, None, 0, st()), "gemm")
T = lambda: tiny.add_(1.0)
S = lambda: LL.check(L.pq_dequant(big_q.data_ptr(), 4096, big_s.data_ptr(), 1, 8192, 4096, big_o.data_ptr(), 4096, 0, st()), "dequant")
Ka(); Kb(); torch.cuda.synchronize()
def graph(fns, n=10):
    s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(s2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            for f in fns: f()
    return g
seqs = {"G": [G], "G Ka": [G, Ka], "G Ka Kb": [G, Ka, Kb], "G Ka Ka": [G, Ka, Ka], "Ka": [Ka], "Ka Kb": [Ka, Kb], "G T Ka": [G, T, Ka], "G T": [G, T],
        "S": [S], "S Ka": [S, Ka], "G G": [G, G], "G Ka G Kb": [G, Ka, G, Kb]}
gs = {k: graph(v) for k, v in seqs.items()}
import time
t0 = time.time()
while time.time() - t0 < 1.5:
    gs["G Ka"].replay()
torch.cuda.synchronize()
ts = {k: [] for k in gs}
for r in range(15):
    for k, g in gs.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); g.replay(); b.record(); b.synchronize()
        ts[k].append(a.elapsed_time(b) * 1e3 / 20)
m = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
for k, v in m.items():
    print(f"{k:12s} {v:8.2f} us per repetition")
print(f"K1 right after a GEMM                    : {m['G Ka'] - m['G']:.2f} us")
print(f"second K1, other input, behind the first : {m['G Ka Kb'] - m['G Ka']:.2f} us")
print(f"second K1, same input                    : {m['G Ka Ka'] - m['G Ka']:.2f} us")
print(f"K1 alone / alternating two inputs        : {m['Ka']:.2f} / {m['Ka Kb'] / 2:.2f} us")
print(f"tiny kernel after GEMM, K1 after tiny    : {m['G T'] - m['G']:.2f} / {m['G T Ka'] - m['G T']:.2f} us")
print(f"K1 after a streaming (low-power) kernel  : {m['S Ka'] - m['S']:.2f} us")
print(f"GEMM after GEMM                          : {m['G G'] - m['G']:.2f} us;  two steps on alternating inputs: {m['G Ka G Kb'] / 2:.2f} us per step")
