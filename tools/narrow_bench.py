"""Dev tool: narrow-N / small-MN GEMM shapes under each forced tile variant (PQ_FORCE_VARIANT), no split-K workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoquant_amd import _lib as L
lib = L.lib()
SHAPES = [(4096, 1024, 4096, "8B k,v"), (4096, 1024, 8192, "70B q/o shard"), (4096, 1024, 28672, "70B down shard"), (512, 4096, 4096, "M=512"),
          (2048, 4096, 11008, "cfg3 down"), (4096, 3584, 8192, "70B gate/up shard"), (1024, 1024, 8192, "1k x 1k"), (4096, 512, 8192, "N=512")]


def t(fn, it=100):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3: fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / it


for M, N, K, name in SHAPES:
    xq = (torch.randn(M, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
    wq = (torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
    xs = torch.rand(M, device="cuda"); ws = torch.rand(N, device="cuda"); y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ops = 2.0 * M * N * K
    row = f"{name:18s} {M:5d} x {N:5d} x {K:5d} "
    for v in ("sp128_16", "sp128x128", "ring128", ""):
        L.set_option("PQ_FORCE_VARIANT", v)
        wb = lib.pq_qlinear_workspace_bytes(M, N, K) if v == "" else 0
        wsp = torch.empty(max(wb, 16), dtype=torch.uint8, device="cuda")
        f = lambda: lib.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, wsp.data_ptr() if wb else None, wb, st)
        us = t(f)
        row += f" | {v or 'auto':9s} {us:7.1f} us {ops / us / 1e6 / 50.33:5.1f} %"
    print(row + f"  [{lib.pq_gemm_variant_name(M, N, K, K, K).decode()}]")
