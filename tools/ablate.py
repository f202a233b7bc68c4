"""Dev tool: time the fast GEMM under ablation flags (PQ_GEMM_DBG bits: 1 no DMA, 2 no LDS reads, 4 no MFMA, 8 no epilogue)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import protoquant_amd as pq
from tools.quick_bench import timeit
M = N = K = 4096
torch.manual_seed(0)
xq = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
wq = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
xs = torch.rand(M, device="cuda") * 0.01; ws = torch.rand(N, device="cuda") * 0.01
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
os.environ["PQ_FORCE_VARIANT"] = "sp256_16"
import time
t0 = time.time()
while time.time() - t0 < 1.0:
    pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=out)
from protoquant_amd import _lib
stamps = torch.zeros(512, dtype=torch.int64, device="cuda")
_lib.lib().pq_dev_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
for flags in [int(a) for a in sys.argv[1:]] or [0, 8, 1, 2, 3, 4, 12, 9, 10, 11, 13, 14, 15, 0]:
    os.environ["PQ_GEMM_DBG"] = str(flags)
    med, mn = timeit(lambda: pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=out), iters=100)
    names = [n for b, n in ((1, "noDMA"), (2, "noLDS"), (4, "noMFMA"), (8, "noEPI"), (32, "noVMWAIT"), (64, "noBARRIER"), (128, "noSTORE"), (256, "constSCALE")) if flags & b]
    torch.cuda.synchronize(); st = stamps.cpu().numpy().reshape(-1, 2)
    clk = (st[:, 0].sum() / max(st[:, 1].sum(), 1)) * 0.1 if flags else float("nan")
    cyc = st[:, 0].mean() if flags else float("nan")
    print(f"flags={flags:4d} {'+'.join(names) or 'full':28s} median {med:7.1f} us  min {mn:7.1f} us   loop clock {clk:.3f} GHz  loop cycles {cyc:9.0f}")
