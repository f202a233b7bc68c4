"""Dev tool (GPU box): the 64 < M <= 512 regime (round-3 verdict item 4).  For M in {128, 256, 512} x the weight shapes of Llama-3-8B
(q/o 4096 x 4096, down 4096 x 14336, fused gate+up 28672 x 4096): pq_qlinear_s8 timed from hipGraph replays (these kernels take 5 - 50 us:
eager launches from Python would time the host), one weight matrix replayed (warm) and a rotation over > 600 MB of them (every launch
streams its weights from HBM, as a layer inside a model does); next to each time the floor the verdict names —
max(weight bytes / 6.29 TB/s, ops / 5.033 POPS) — and the ratio to it.
usage: python tools/midm_bench.py [--lib path.so] [--opts PQ_X=v,...] [--shapes MxNxK,...]"""
import argparse
import ctypes
import os
import sys

import torch

i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "protoquant_amd", "libpq_hip.so"))
    ap.add_argument("--opts", default="")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--rounds", type=int, default=11)
    ap.add_argument("--preset", default="", help="model: the GEMM shapes of BASELINE configs 2-5 (the per-shape table of DESIGN.md section 4)")
    a = ap.parse_args()
    L = ctypes.CDLL(os.path.abspath(a.lib))
    L.pq_qlinear_s8.restype = i32
    L.pq_qlinear_s8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    L.pq_qlinear_workspace_bytes.restype = sz
    L.pq_qlinear_workspace_bytes.argtypes = [i64, i64, i64]
    L.pq_gemm_variant_name.restype = ctypes.c_char_p
    L.pq_gemm_variant_name.argtypes = [i64, i64, i64, i64, i64]
    L.pq_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    for o in filter(None, a.opts.split(",")):
        k, v = o.split("=")
        assert L.pq_set_option(k.encode(), v.encode()) == 0, o
    MODEL = [(4096, 4096, 4096, "cfg2 / 8B q,o"), (2048, 11008, 4096, "cfg3 gate/up"), (2048, 22016, 4096, "cfg3 fused gate+up"), (2048, 4096, 11008, "cfg3 down"),
             (4096, 1024, 4096, "8B k,v"), (4096, 6144, 4096, "8B fused qkv"), (4096, 14336, 4096, "8B gate/up"), (4096, 28672, 4096, "8B fused gate+up"),
             (4096, 4096, 14336, "8B down"), (4096, 128256, 4096, "lm_head"), (4096, 1024, 8192, "70B q/o shard"), (4096, 1280, 8192, "70B fused qkv shard"),
             (4096, 3584, 8192, "70B gate/up shard"), (4096, 7168, 8192, "70B fused gate+up shard"), (4096, 1024, 28672, "70B down shard"),
             (4096, 8192, 1024, "70B o, row-sharded"), (4096, 8192, 3584, "70B down, row-sharded"), (8192, 8192, 8192, "8k cube"), (512, 4096, 4096, "M=512"), (32, 512, 512, "cfg1")]
    names = {}
    if a.preset == "model":
        shapes = [m[:3] for m in MODEL]
        names = {m[:3]: m[3] for m in MODEL}
    else:
        shapes = ([tuple(int(v) for v in s.split("x")) for s in a.shapes.split(",")] if a.shapes else
                  [(M, N, K) for (N, K) in ((4096, 4096), (4096, 14336), (28672, 4096)) for M in (128, 256, 512)])
    dev = torch.device("cuda:0")
    print(f"# lib {a.lib} opts [{a.opts}]")
    print(f"# {'shape':>18s} {'warm us':>9s} {'HBM-fed us':>11s} {'floor us':>9s} {'warm/floor':>10s} {'hbm/floor':>10s}  dispatch")
    for (M, N, K) in shapes:
        torch.manual_seed(1)
        xq = (torch.randn(M, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        nrot = max(2, min(48, -(-640 * 2**20 // (N * K))))
        if N * K > 400 * 2**20:
            nrot = 2
        wrot = [(torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8) for _ in range(nrot)]
        xs = torch.rand(M, device=dev) * 1e-2 + 1e-3
        ws = torch.rand(N, device=dev) * 1e-2 + 1e-3
        y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        wb = L.pq_qlinear_workspace_bytes(M, N, K)
        wsp = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)

        def launch(w):
            st = torch.cuda.current_stream().cuda_stream
            rc = L.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), w.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K,
                                 wsp.data_ptr() if wb else None, wb, st)
            assert rc == 0

        def graph(fn):
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g
        PG = 24 if 2.0 * M * N * K < 4e11 else 6
        g_warm = graph(lambda: [launch(wrot[0]) for _ in range(PG)])
        g_rot = graph(lambda: [launch(w) for w in wrot])

        def ev(g, n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        for _ in range(3):
            g_warm.replay(); g_rot.replay()
        torch.cuda.synchronize()
        tw, tr = [], []
        for _ in range(a.rounds):
            tw.append(ev(g_warm, PG)); tr.append(ev(g_rot, nrot))
        med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
        floor = max(N * K / 6.29e6, 2.0 * M * N * K / 5.033e9)
        name = L.pq_gemm_variant_name(M, N, K, K, K).decode() + (" + workspace" if wb else "")
        pk = lambda us: 2.0 * M * N * K / us / 1e6 / 50.33      # noqa: E731   % of 5.033 POPS
        print(f"  {M:5d}x{N:5d}x{K:5d} {med(tw):9.2f} {med(tr):11.2f} {floor:9.2f} {med(tw) / floor:10.2f} {med(tr) / floor:10.2f}  {pk(med(tw)):5.1f} % / {pk(med(tr)):5.1f} % of peak  [{name}]"
              + (f"  {names[(M, N, K)]}" if (M, N, K) in names else ""), flush=True)
        del wrot, g_warm, g_rot


if __name__ == "__main__":
    main()
