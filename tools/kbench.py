"""Dev tool: per-kernel timings through the C-ABI with preallocated buffers (no allocator / wrapper overhead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import protoquant_amd as pq
from protoquant_amd import _lib as L
lib = L.lib()
R, C = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (4096, 4096)
x = torch.randn(R, C).to(torch.bfloat16).cuda()
q = torch.empty((R, C), dtype=torch.int8, device="cuda"); sr = torch.empty(R, device="cuda"); sc = torch.empty(C, device="cuda")
o = torch.empty((R, C), dtype=torch.bfloat16, device="cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
def t(fn, it=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / it
by = 3 * R * C
u = t(lambda: lib.pq_quant_rowwise(x.data_ptr(), 0, R, C, C, q.data_ptr(), C, sr.data_ptr(), st())); print(f"K1 rowwise {R}x{C}: {u:.2f} us  {by/u/1e6:.2f} TB/s")
u = t(lambda: lib.pq_quant_colwise(x.data_ptr(), 0, R, C, C, q.data_ptr(), C, sc.data_ptr(), st())); print(f"K2 colwise {R}x{C}: {u:.2f} us  {by/u/1e6:.2f} TB/s algorithmic (3 B/elem)")
lib.pq_quant_rowwise(x.data_ptr(), 0, R, C, C, q.data_ptr(), C, sr.data_ptr(), st())
u = t(lambda: lib.pq_dequant(q.data_ptr(), C, sr.data_ptr(), 1, R, C, o.data_ptr(), C, 0, st())); print(f"dequant rows {R}x{C}->bf16: {u:.2f} us  {by/u/1e6:.2f} TB/s")
u = t(lambda: lib.pq_dequant(q.data_ptr(), C, sc.data_ptr(), 0, R, C, o.data_ptr(), C, 0, st())); print(f"dequant cols {R}x{C}->bf16: {u:.2f} us  {by/u/1e6:.2f} TB/s")
