"""Dev tool (GPU box): K1n (pq_rmsnorm_quant_rowwise) at hidden 4096 / 8192, wave-per-row layout vs 256-thread block per row
(pq_set_option PQ_RMS_WAVE_MAX), interleaved hipGraph replays; identical outputs required."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoquant_amd import _lib as L
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
import ctypes
for (R, C) in ((4096, 4096), (16384, 4096), (4096, 2048), (2048, 4096)):
    x = (torch.randn(R, C) * 2).to(torch.bfloat16).cuda(); w = (1 + 0.1 * torch.randn(C)).to(torch.bfloat16).cuda()
    outs, graphs = {}, {}
    for wm in ("512", "0"):
        L.set_option("PQ_RMS_WAVE_MAX", wm)
        q = torch.empty((R, C), dtype=torch.int8, device="cuda"); s = torch.empty(R, device="cuda")
        f = lambda: L.check(lib.pq_rmsnorm_quant_rowwise(x.data_ptr(), C, w.data_ptr(), ctypes.c_float(1e-5), 0, R, C, q.data_ptr(), C, s.data_ptr(), None, C, st()), "k1n")
        f(); torch.cuda.synchronize()
        outs[wm] = (q.clone(), s.clone())
        s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            f()
        torch.cuda.current_stream().wait_stream(s2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                f()
        graphs[wm] = g
    same = torch.equal(outs["512"][0], outs["0"][0]) and torch.equal(outs["512"][1], outs["0"][1])
    for g in graphs.values():
        for _ in range(10):
            g.replay()
    torch.cuda.synchronize()
    t = {"512": [], "0": []}
    for r in range(15):
        for wm in ("512", "0"):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); graphs[wm].replay(); b.record(); b.synchronize()
            t[wm].append(a.elapsed_time(b) * 1e3 / 10)
    by = 3 * R * C + 4 * R
    for wm in ("512", "0"):
        v = sorted(t[wm]); med = v[len(v) // 2]
        print(f"K1n {R}x{C} bf16 {'wave per row' if wm == '512' else '256-thread block per row'}: same={same} median {med:7.2f} us min {v[0]:7.2f} us  {by / med / 1e6:.2f} TB/s algorithmic", flush=True)
L.set_option("PQ_RMS_WAVE_MAX", "")      # back to the default (256)
