"""Dev tool (GPU box, `make ABLATION=1` build): time the fast GEMM under ablation flags (PQ_GEMM_DBG bits: 1 no DMA, 2 no LDS
reads, 4 no MFMA, 8 no epilogue, 32 no vmcnt waits, 64 no barriers, 1024 none = stamps only) and print the in-kernel timeline
from the per-wave stamps: launch ramp, prologue, K-loop, epilogue (100 MHz chip-wide counter) and the in-loop clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import protoquant_amd as pq
from protoquant_amd import _lib as _pqlib  # noqa: E402
if os.environ.get("PQ_ABL_LIB"):          # a dev build kept beside the product library (make ABLATION=1, copied to tools/libpq_hip_abl.so)
    _pqlib.LIB_PATH = os.path.abspath(os.environ["PQ_ABL_LIB"])
from tools.quick_bench import timeit
M = N = K = 4096
if os.environ.get("PQ_ABL_SHAPE"):          # e.g. PQ_ABL_SHAPE=4096x28672x4096 (multi-round grids: stamps are per hardware block id)
    M, N, K = (int(v) for v in os.environ["PQ_ABL_SHAPE"].split("x"))
NT = ((M + 255) // 256) * ((N + 255) // 256)
torch.manual_seed(0)
if os.environ.get("PQ_ABL_UNIFORM"):
    xq = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda"); wq = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
else:   # gaussian codes, what per-token quantisation of N(0,1) data produces
    xq = (torch.randn(M, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8); wq = (torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
xs = torch.rand(M, device="cuda") * 0.01; ws = torch.rand(N, device="cuda") * 0.01
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
_pqlib.set_option("PQ_FORCE_VARIANT", "sp256_16")
for kv in filter(None, os.environ.get("PQ_ABL_OPTS", "").split(",")):      # e.g. PQ_ABL_OPTS=PQ_SP256_P3=0
    _pqlib.set_option(*kv.split("="))
import time
t0 = time.time()
while time.time() - t0 < 1.5:
    pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=out)
stamps = torch.zeros(NT * 8 * 4 * 2, dtype=torch.int64, device="cuda")
_pqlib.lib().pq_dev_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
for flags in [int(a) for a in sys.argv[1:]] or [0, 1024, 8, 1, 2, 3, 4, 12, 32, 64]:
    os.environ["PQ_GEMM_DBG"] = str(flags)
    stamps.zero_()
    med, mn = timeit(lambda: pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=out), iters=100)
    names = [n for b, n in ((1, "noDMA"), (2, "noLDS"), (4, "noMFMA"), (8, "noEPI"), (32, "noVMWAIT"), (64, "noBARRIER"), (2048, "halfEPI"), (1024, "stamps")) if flags & b]
    torch.cuda.synchronize()
    line = f"flags={flags:4d} {'+'.join(names) or 'product':22s} median {med:6.1f} us  min {mn:6.1f} us"
    if flags:
        st = stamps.cpu().numpy().reshape(NT, 8, 4, 2).astype(np.float64)
        rt, cy = st[..., 0] * 0.01, st[..., 1]            # us (100 MHz), shader cycles
        t00 = rt[:, :, 0].min()
        has3 = (st[:, :, 3, 0] > 0).all()
        pr = lambda a: f"{np.median(a):6.2f} [{a.min():6.2f} .. {a.max():6.2f}]"
        line += (f"\n      entry after first wave {pr(rt[:, :, 0] - t00)} us | prologue {pr(rt[:, :, 1] - rt[:, :, 0])} us | K-loop {pr(rt[:, :, 2] - rt[:, :, 1])} us"
                 + (f" | epilogue issue {pr(rt[:, :, 3] - rt[:, :, 2])} us | last stamp at {(rt[:, :, 3].max() - t00):6.2f} us" if has3 else f" | loop end at {(rt[:, :, 2].max() - t00):6.2f} us")
                 + f"\n      K-loop cycles {np.median(cy[:, :, 2] - cy[:, :, 1]):9.0f}  in-loop clock {np.median((cy[:, :, 2] - cy[:, :, 1]) / np.maximum(rt[:, :, 2] - rt[:, :, 1], 1e-9)) / 1e3:.3f} GHz"
                 + f"  prologue cycles {np.median(cy[:, :, 1] - cy[:, :, 0]):7.0f}" + (f"  epilogue cycles {np.median(cy[:, :, 3] - cy[:, :, 2]):7.0f}" if has3 else ""))
        if NT > 256 and has3:        # multi-round grid: how the rounds line up (entry times of the blocks, per-block total)
            ent = np.sort(rt[:, 0, 0] - t00)
            tot = rt[:, :, 3].max(axis=1) - rt[:, :, 0].min(axis=1)
            line += (f"\n      blocks {NT}: entry times of block #256/#512/#768/last: " + " / ".join(f"{ent[min(i, NT - 1)]:.1f}" for i in (256, 512, 768, NT - 1))
                     + f" us;  per-block entry->epilogue issued: median {np.median(tot):.2f} us [{tot.min():.2f} .. {tot.max():.2f}]")
    print(line, flush=True)
