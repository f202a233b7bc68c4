"""Dev tool (GPU box, `make ABLATION=1` build copied to tools/libpq_hip_abl.so): the 128 x 128 ring tile (gemm_s8_ring128, loader / consumer form) under timing-only
ablations — PQ_GEMM_DBG bits: 1 no LDS-DMA inside the K-loop, 2 no fragment reads, 4 no MFMAs, 8 no barriers inside the loop (results are wrong for every flag != 0).  Whole-launch times from hipGraph replays,
every flag in every round (interleaved).  Usage: PQ_ABL_LIB=tools/libpq_hip_abl.so python tools/ablate_ring.py [MxNxK ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import protoquant_amd as pq
from protoquant_amd import _lib as _pqlib  # noqa: E402
if os.environ.get("PQ_ABL_LIB"):
    _pqlib.LIB_PATH = os.path.abspath(os.environ["PQ_ABL_LIB"])
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 1024, 8192), (4096, 1024, 28672), (4096, 1024, 4096)]
FLAGS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 14]
NAMES = {0: "product", 1: "noDMA", 2: "noFRAG", 3: "noDMA+noFRAG (MFMA stream)", 4: "noMFMA", 5: "noDMA+noMFMA (fragment reads)", 6: "noFRAG+noMFMA (DMA stream)", 7: "none of the three", 8: "noBARRIER (streams decoupled)", 9: "noBARRIER+noDMA", 10: "noBARRIER+noFRAG", 11: "noBARRIER: MFMA stream", 14: "noBARRIER: DMA stream"}
_pqlib.set_option("PQ_FORCE_VARIANT", "ring128_16")
REP = 20
for M, N, K in shapes:
    torch.manual_seed(0)
    NW = max(2, min(12, (600 << 20) // (N * K)))       # distinct weight sets: HBM-fed
    xq = (torch.randn(M, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
    wqs = [(torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8) for _ in range(NW)]
    xs = torch.rand(M, device="cuda") * 0.01; ws = torch.rand(N, device="cuda") * 0.01
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    graphs = {}
    for f in FLAGS:
        os.environ["PQ_GEMM_DBG"] = str(f)
        for i in range(3): pq.qlinear_s8(xq, xs, wqs[i % NW], ws, None, torch.bfloat16, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(REP): pq.qlinear_s8(xq, xs, wqs[i % NW], ws, None, torch.bfloat16, out=out)
        graphs[f] = g
    res = {f: [] for f in FLAGS}
    for r in range(12):
        for f in FLAGS:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[f].replay(); a.record(); graphs[f].replay(); b.record(); torch.cuda.synchronize()
            res[f].append(a.elapsed_time(b) * 1e3 / REP)
    nt = K // 128
    # in-kernel stamps (flags | 16): the consumers' K-loop in shader cycles and its clock
    import ctypes
    nblk = ((M + 127) // 128) * ((N + 127) // 128)
    stamps = torch.zeros(nblk * 4 * 2 * 2, dtype=torch.int64, device="cuda")
    _pqlib.lib().pq_dev_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
    clk = {}
    for f in (0, 1, 2, 3, 4, 6, 8):
        os.environ["PQ_GEMM_DBG"] = str(f | 16)
        for i in range(40): pq.qlinear_s8(xq, xs, wqs[i % NW], ws, None, torch.bfloat16, out=out)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().reshape(nblk, 4, 2, 2).astype(np.float64)
        us = (st[:, :, 1, 0] - st[:, :, 0, 0]) * 0.01; cy = st[:, :, 1, 1] - st[:, :, 0, 1]
        clk[f] = (float(np.median(cy)), float(np.median(us)), float(np.median(cy / np.maximum(us, 1e-9)) / 1e3))
    _pqlib.lib().pq_dev_set_stamp_buffer(ctypes.c_void_p(0))
    print(f"== {M}x{N}x{K}  ({NW} weight sets, {nt} K-tiles; MFMA floor at 2.4 GHz: {nt * 512 / 2400:.1f} us)")
    for f in FLAGS:
        med = float(np.median(res[f][2:]))
        extra = ""
        if f in clk:
            cy, us, ghz = clk[f]
            extra = f"   | consumers' K-loop {cy / nt:6.0f} cycles per K-tile at {ghz:.2f} GHz = {us:6.2f} us (eager launches)"
        print(f"   flags={f} {NAMES[f]:34s} {med:7.2f} us   per K-tile {med * 1e3 / nt:6.1f} ns{extra}", flush=True)
