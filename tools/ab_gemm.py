"""Dev tool (GPU box): A/B several builds of libpq_hip.so in ONE process, interleaved rounds (guide rule 24).

usage: python tools/ab_gemm.py name=path.so [name=path.so ...] [--shapes MxNxK,...] [--dtype bf16|f32] [--rounds R] [--bias]
Each build runs the fused GEMM+epilogue (pq_qlinear_s8) on the same gaussian int8 codes; outputs are compared bit for bit
against the first build; timing = hipGraph of G back-to-back launches, median and min over the rounds."""
import argparse
import ctypes
import os
import sys
import time

import torch

i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    L.pq_qlinear_s8.restype = i32
    L.pq_qlinear_s8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    L.pq_qlinear_workspace_bytes.restype = sz
    L.pq_qlinear_workspace_bytes.argtypes = [i64, i64, i64]
    L.pq_gemm_variant_name.restype = ctypes.c_char_p
    L.pq_gemm_variant_name.argtypes = [i64, i64, i64, i64, i64]
    L.pq_last_error.restype = ctypes.c_char_p
    if hasattr(L, "pq_qlinear_kslabs_workspace_bytes_for"):
        L.pq_qlinear_kslabs_workspace_bytes_for.restype = sz
        L.pq_qlinear_kslabs_workspace_bytes_for.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, i64]
        L.pq_kslabs_way_name.restype = ctypes.c_char_p
        L.pq_kslabs_way_name.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, i64, sz]
        L.pq_qlinear_s8_kslabs.restype = i32
        L.pq_qlinear_s8_kslabs.argtypes = [vp, i64, i64, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--shapes", default="4096x4096x4096")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--per-graph", type=int, default=20)
    ap.add_argument("--bias", action="store_true")
    ap.add_argument("--eager", action="store_true", help="eager back-to-back launches instead of graph replays")
    ap.add_argument("--rotate-weights", type=int, default=1, help="cycle through this many distinct weight matrices (enough of them: every launch reads its weights from HBM, as inside a model)")
    a = ap.parse_args()
    libs = []
    for spec in a.libs:            # name=path[@OPT=VAL,...]: options go through pq_set_option (load a COPY of the .so for a second setting)
        name, rest = spec.split("=", 1)
        path, _, opts = rest.partition("@")
        Lh = load(path)
        Lh.stacked = 0
        for o in filter(None, opts.split(",")):
            k, v = o.split("=")
            if k == "STACKED":          # pseudo-option: the activation codes as G stacked K-slabs through pq_qlinear_s8_kslabs (round 6)
                Lh.stacked = int(v)
                continue
            Lh.pq_set_option.restype = i32
            Lh.pq_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
            assert Lh.pq_set_option(k.encode(), v.encode()) == 0, (k, v)
        libs.append((name, Lh))
    dt = {"bf16": (torch.bfloat16, 0), "fp16": (torch.float16, 1), "f32": (torch.float32, 2)}[a.dtype]
    dev = torch.device("cuda:0")
    for shp in a.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        torch.manual_seed(1)
        xq = (torch.randn(M, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        wq = (torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
        wrot = [wq] + [(torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8) for _ in range(a.rotate_weights - 1)]
        xs = torch.rand(M, device=dev) * 1e-2 + 1e-3
        ws = torch.rand(N, device=dev) * 1e-2 + 1e-3
        bias = (torch.randn(N, device=dev) * 0.1).to(dt[0]) if a.bias else None
        outs, graphs, fns = [], [], []
        st = torch.cuda.current_stream().cuda_stream
        for name, L in libs:
            y = torch.zeros((M, N), dtype=dt[0], device=dev)
            wb = L.pq_qlinear_workspace_bytes(M, N, K)
            Gs = getattr(L, "stacked", 0)
            stk = None
            if Gs:
                kps = K // Gs
                stk = xq.reshape(M, Gs, kps).permute(1, 0, 2).contiguous()
                wb = L.pq_qlinear_kslabs_workspace_bytes_for(stk.data_ptr(), kps, M * kps, kps, wq.data_ptr(), K, M, N, K)
                print(f"   [{name}: {L.pq_kslabs_way_name(stk.data_ptr(), kps, M * kps, kps, wq.data_ptr(), K, M, N, K, wb).decode()}, workspace {wb} B]", flush=True)
            wsp = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)

            cnt = [0]

            def f(L=L, y=y, wsp=wsp, wb=wb, cnt=cnt, stk=stk, Gs=Gs):
                wq = wrot[cnt[0] % len(wrot)]; cnt[0] += 1
                if stk is not None:
                    rc = L.pq_qlinear_s8_kslabs(stk.data_ptr(), K // Gs, M * (K // Gs), K // Gs, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(),
                                                bias.data_ptr() if bias is not None else None, y.data_ptr(), N, dt[1], M, N, K,
                                                wsp.data_ptr() if wb else None, wb, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, L.pq_last_error()
                    return
                rc = L.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(),
                                     bias.data_ptr() if bias is not None else None, y.data_ptr(), N, dt[1], M, N, K,
                                     wsp.data_ptr() if wb else None, wb, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, L.pq_last_error()
            f()
            torch.cuda.synchronize()
            cnt[0] = 0
            outs.append(y.clone())
            fns.append(f)
            if not a.eager:
                s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    f()
                torch.cuda.current_stream().wait_stream(s2)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(a.per_graph):
                        f()
                graphs.append(g)
        same = [bool(torch.equal(outs[0].view(torch.uint8), o.view(torch.uint8))) for o in outs]
        # warm clocks
        t0 = time.time()
        while time.time() - t0 < 1.5:
            for i in range(len(libs)):
                graphs[i].replay() if graphs else fns[i]()
        torch.cuda.synchronize()
        times = [[] for _ in libs]
        for r in range(a.rounds):
            for i in range(len(libs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if graphs:
                    for _ in range(3):
                        graphs[i].replay()
                    n = 3 * a.per_graph
                else:
                    for _ in range(60):
                        fns[i]()
                    n = 60
                e1.record(); e1.synchronize()
                times[i].append(e0.elapsed_time(e1) * 1e3 / n)
        ops = 2.0 * M * N * K
        for (name, L), t, ok in zip(libs, times, same):
            t = sorted(t)
            med, mn = t[len(t) // 2], t[0]
            print(f"{shp:>18s} {a.dtype}{'+bias' if a.bias else ''} {name:10s} same={ok} median {med:8.2f} us  min {mn:8.2f} us  {ops / med / 1e6:7.1f} TOPS  "
                  f"{ops / med / 1e6 / 50.33:5.1f} %  [{L.pq_gemm_variant_name(M, N, K, K, K).decode()}]", flush=True)


if __name__ == "__main__":
    main()
