"""Dev tool (GPU box): the qlinear step of several library builds in one process, interleaved: K1 alone and the GEMM alone (gap-free
hipGraph replays) and the step K1 -> GEMM, every kernel from the build under test; identical outputs required.  usage: name=path.so ..."""
import ctypes, os, sys
import torch
i32, i64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
libs = []
for spec in sys.argv[1:]:
    n, pth = spec.split("=")
    L = ctypes.CDLL(os.path.abspath(pth))
    L.pq_quant_rowwise.restype = i32; L.pq_quant_rowwise.argtypes = [vp, i32, i64, i64, i64, vp, i64, vp, vp]
    L.pq_qlinear_s8.restype = i32; L.pq_qlinear_s8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i64, i64, i64, vp, sz, vp]
    libs.append((n, L))
M = N = K = 4096
x = torch.randn(M, K).to(torch.bfloat16).cuda()
wq = (torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8); ws = torch.rand(N, device="cuda") * 1e-3
y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
def graph(fn, n):
    s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        fn()
    torch.cuda.current_stream().wait_stream(s2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g
outs, g_alone, g_step, g_gemm = [], [], [], []
L0 = libs[0][1]
for n, L in libs:
    q = torch.empty((M, K), dtype=torch.int8, device="cuda"); s = torch.empty(M, device="cuda")
    k1 = lambda L=L, q=q, s=s: L.pq_quant_rowwise(x.data_ptr(), 0, M, K, K, q.data_ptr(), K, s.data_ptr(), st())
    gm = lambda L=L, q=q, s=s: L.pq_qlinear_s8(q.data_ptr(), K, s.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, None, 0, st())
    k1(); torch.cuda.synchronize(); outs.append((q.clone(), s.clone()))
    gm(); torch.cuda.synchronize(); outs[-1] = outs[-1] + (y.clone(),)
    g_alone.append(graph(k1, 20)); g_step.append(graph(lambda: (k1(), gm()), 10)); g_gemm.append(graph(gm, 20))
same = [torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2].view(torch.int16), o[2].view(torch.int16)) for o in outs]
for g in g_alone + g_step + g_gemm:
    for _ in range(5):
        g.replay()
torch.cuda.synchronize()
ta, ts, tg = [[] for _ in libs], [[] for _ in libs], [[] for _ in libs]
for r in range(21):
    for i in range(len(libs)):
        for tgt, g, n in ((ta, g_alone[i], 20), (ts, g_step[i], 10), (tg, g_gemm[i], 20)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); b.synchronize()
            tgt[i].append(a.elapsed_time(b) * 1e3 / n)
for (n, _), a, s, gg, ok in zip(libs, ta, ts, tg, same):
    a.sort(); s.sort(); gg.sort()
    print(f"{n:8s} same={ok}  K1 alone median {a[len(a)//2]:6.2f} us (min {a[0]:6.2f})   GEMM alone median {gg[len(gg)//2]:6.2f} us (min {gg[0]:6.2f})   step K1+GEMM median {s[len(s)//2]:6.2f} us (min {s[0]:6.2f})", flush=True)
