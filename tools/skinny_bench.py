"""Dev tool: small-M (decode-like) qlinear shapes: K1 + GEMM through pq_qlinear_dyn, HIP events over graph replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import protoquant_amd as pq
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


for N, K in ((4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (128256, 4096)):
    wq = (torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
    ws = torch.rand(N, device="cuda") * 1e-3
    for M in ((1, 16, 32) if os.environ.get('SK_QUICK') else (1, 8, 16, 32, 64, 128, 256)):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        xq = torch.empty((M, K), dtype=torch.int8, device="cuda"); xs = torch.empty(M, device="cuda")
        y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        st = lambda: torch.cuda.current_stream().cuda_stream
        f_k1 = lambda: lib.pq_quant_rowwise(x.data_ptr(), 0, M, K, K, xq.data_ptr(), K, xs.data_ptr(), st())
        f_g = lambda: lib.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, None, 0, st())
        t1, t2 = graph_time(f_k1), graph_time(f_g)
        wbytes = N * K + M * K + 2 * M * N
        print(f"M={M:4d} N={N:6d} K={K:5d}  K1 {t1:6.2f} us  GEMM {t2:8.2f} us  ({wbytes / t2 / 1e6:5.2f} TB/s of operand bytes; floor at 5 TB/s {wbytes / 5e6:7.2f} us)  "
              f"[{lib.pq_gemm_variant_name(M, N, K, K, K).decode()}]")
