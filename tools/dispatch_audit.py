"""Dev tool (GPU box): audit of the dispatcher.  Over a grid of (M, N, K) it times pq_qlinear_s8 as dispatched ("auto") and with every tile variant forced, all from hipGraph
replays over a rotation of weight matrices (HBM-fed, what a layer inside a model sees), interleaved round by round, and lists the shapes where the dispatch is more than
5 % slower than the best forced variant.  usage: python tools/dispatch_audit.py [--quick] [--small] [--all-times]"""
import ctypes, os, sys
import torch
i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "protoquant_amd", "libpq_hip.so"))
L.pq_qlinear_s8.restype = i32
L.pq_qlinear_s8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
L.pq_qlinear_workspace_bytes.restype = sz; L.pq_qlinear_workspace_bytes.argtypes = [i64, i64, i64]
L.pq_gemm_variant_name.restype = ctypes.c_char_p; L.pq_gemm_variant_name.argtypes = [i64, i64, i64, i64, i64]
L.pq_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
VARS = ["", "sp256_16", "sp128_16", "ring128", "ring64x128", "ring64x64", "ring128x160", "skinny"]
quick = "--quick" in sys.argv
Ms = [48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192]
if "--small" in sys.argv:            # the decode-like end: the weight-streaming kernel against the 64-row ring tiles
    Ms = [1, 8, 16, 17, 24, 32, 33, 40, 48, 64]
Ns = [512, 1024, 2048, 4096, 6144, 8192, 14336, 28672]
Ks = [1024, 4096, 8192] if not quick else [4096]
for a_ in sys.argv:                 # --ks=14336,28672  --ns=4096,8192
    if a_.startswith("--ks="): Ks = [int(v) for v in a_[5:].split(",")]
    if a_.startswith("--ns="): Ns = [int(v) for v in a_[5:].split(",")]
    if a_.startswith("--ms="): Ms = [int(v) for v in a_[5:].split(",")]
dev = torch.device("cuda:0")
bad = []
for K in Ks:
    for N in Ns:
        nrot = max(2, min(24, -(-320 * 2**20 // (N * K))))
        wrot = [(torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8) for _ in range(nrot)]
        ws = torch.rand(N, device=dev) * 1e-2 + 1e-3
        for M in Ms:
            if 2.0 * M * N * K > 3e12:
                continue
            xq = (torch.randn(M, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
            xs = torch.rand(M, device=dev) * 1e-2 + 1e-3
            y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            graphs = {}
            for v in VARS:
                if v == "skinny" and M > 64:
                    continue
                L.pq_set_option(b"PQ_FORCE_VARIANT", v.encode())
                wb = L.pq_qlinear_workspace_bytes(M, N, K) if v == "" else 0
                wsp = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)
                def launch(w, wb=wb, wsp=wsp):
                    rc = L.pq_qlinear_s8(xq.data_ptr(), K, xs.data_ptr(), w.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K,
                                         wsp.data_ptr() if wb else None, wb, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0
                s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    launch(wrot[0])
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for w in wrot:
                        launch(w)
                graphs[v] = (g, wsp, L.pq_gemm_variant_name(M, N, K, K, K).decode() + (" +ws" if wb else ""))
            L.pq_set_option(b"PQ_FORCE_VARIANT", b"")
            ts = {v: [] for v in graphs}
            for g, _, _ in graphs.values():
                g.replay()
            torch.cuda.synchronize()
            for _ in range(5):
                for v, (g, _, _) in graphs.items():
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); g.replay(); b.record(); b.synchronize()
                    ts[v].append(a.elapsed_time(b) * 1e3 / nrot)
            med = {v: sorted(t)[len(t) // 2] for v, t in ts.items()}
            best = min((t, v) for v, t in med.items() if v)
            flag = med[""] > 1.05 * best[0]
            extra = ("  | " + " ".join(f"{v}:{t:.1f}" for v, t in med.items() if v)) if "--all-times" in sys.argv else ""
            line = f"{M:5d}x{N:5d}x{K:5d} auto {med['']:8.2f} us [{graphs[''][2]:28s}] best forced {best[0]:8.2f} [{best[1]}]{extra}" + (f"   <-- {100 * (med[''] / best[0] - 1):.0f} % slower" if flag else "")
            print(line, flush=True)
            if flag:
                bad.append(line)
        del wrot
print(f"\n{len(bad)} shapes where the dispatch is > 5 % behind the best forced variant:")
for l in bad:
    print(l)
