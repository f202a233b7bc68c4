"""Dev tool: the shipped GEMM+epilogue at 4096^3 on zero / gaussian-code / uniform-byte operands (DVFS / power)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import protoquant_amd as pq
from protoquant_amd import _lib as L
lib = L.lib()
M = N = K = 4096
xs = torch.rand(M, device="cuda") * 0.01; ws = torch.rand(N, device="cuda") * 0.01
y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
gens = {"all zero": lambda r, c: torch.zeros((r, c), dtype=torch.int8, device="cuda"),
        "gaussian codes sigma 28": lambda r, c: (torch.randn(r, c, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8),
        "uniform bytes": lambda r, c: torch.randint(-128, 128, (r, c), dtype=torch.int8, device="cuda")}
for name, g in gens.items():
    a, b = g(M, K), g(N, K)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.pq_qlinear_s8(a.data_ptr(), K, xs.data_ptr(), b.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, None, 0, st)
    t0 = time.time()
    while time.time() - t0 < 1.0: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 300
    print(f"{name:26s} {us:6.1f} us  {2.0*M*N*K/us/1e6:7.1f} TOPS  {2.0*M*N*K/us/1e6/50.33:5.1f} % of 5033")
