"""Dev tool (GPU box): K1 (4096 x 4096 bf16 by default) under different FEEDS and launch options, interleaved hipGraph replays:
  l2   : one input replayed (served by the L2s)           mall : two alternating inputs (96 MB working set, Infinity Cache)
  hbm  : 13 rotating input/output pairs (624 MB)          step : (GEMM, K1) - (GEMM)   = K1 right behind the 4096^3 GEMM
usage: python tools/k1_feed_probe.py [--rows R --cols C] OPT=VAL[,OPT=VAL] ...   (each argument is one configuration; '-' = defaults)"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protoquant_amd import _lib as LL
ap = argparse.ArgumentParser()
ap.add_argument("cfgs", nargs="*", default=["-"])
ap.add_argument("--rows", type=int, default=4096)
ap.add_argument("--cols", type=int, default=4096)
ap.add_argument("--lib", default=None, help="another build of libpq_hip.so (A/B across processes)")
a = ap.parse_args()
if a.lib:
    LL.LIB_PATH = os.path.abspath(a.lib)
L = LL.lib()
M, K, N = a.rows, a.cols, 4096
dev = "cuda"
nrot = max(2, -(-624 * 2**20 // (3 * M * K)))
xs_ = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nrot)]
qs_ = [torch.empty((M, K), dtype=torch.int8, device=dev) for _ in range(nrot)]
ss_ = [torch.empty(M, device=dev) for _ in range(nrot)]
wq = (torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8); ws = torch.rand(N, device=dev) * 1e-3
y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
def K1(i):
    return lambda: LL.check(L.pq_quant_rowwise(xs_[i].data_ptr(), 0, M, K, K, qs_[i].data_ptr(), K, ss_[i].data_ptr(), st()), "k1")
G = lambda: LL.check(L.pq_qlinear_s8(qs_[0].data_ptr(), K, ss_[0].data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, None, 0, st()), "gemm")
def graph(fns, n):
    s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(s2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            for f in fns: f()
    return g
def setopts(cfg):
    for k in ("PQ_K1_RPW", "PQ_K1_LDS", "PQ_K1_ST16"):
        LL.set_option(k, "0")
    if cfg != "-":
        for kv in cfg.split(","):
            k, v = kv.split("="); LL.set_option(k, v)
ref = None
graphs = {}
K1(0)(); torch.cuda.synchronize()
for cfg in a.cfgs:
    setopts(cfg)
    K1(0)(); torch.cuda.synchronize()
    if ref is None: ref = (qs_[0].clone(), ss_[0].clone())
    same = torch.equal(ref[0], qs_[0]) and torch.equal(ref[1], ss_[0])
    graphs[cfg] = {"l2": (graph([K1(0)], 20), 20), "mall": (graph([K1(0), K1(1)], 10), 20), "hbm": (graph([K1(i) for i in range(nrot)], 2), 2 * nrot),
                   "GK": (graph([G, K1(0)], 10), 10), "same": same}
setopts("-")
gG = graph([G], 10)
t0 = time.time()
while time.time() - t0 < 1.0:
    gG.replay()
torch.cuda.synchronize()
def ev(g, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
res = {c: {k: [] for k in ("l2", "mall", "hbm", "GK")} for c in a.cfgs}
tG = []
for r in range(15):
    tG.append(ev(gG, 10))
    for c in a.cfgs:
        for k in ("l2", "mall", "hbm", "GK"):
            g, n = graphs[c][k]
            res[c][k].append(ev(g, n))
med = lambda v: sorted(v)[len(v) // 2]
g0 = med(tG)
byt = 3 * M * K + 4 * M
print(f"K1 {M} x {K} bf16, {byt / 1e6:.1f} MB algorithmic; GEMM alone {g0:.2f} us")
for c in a.cfgs:
    r_ = res[c]
    print(f"{c:28s} same={graphs[c]['same']}  l2 {med(r_['l2']):6.2f} us   mall {med(r_['mall']):6.2f} us ({byt / med(r_['mall']) / 1e6:4.2f} TB/s)   "
          f"hbm {med(r_['hbm']):6.2f} us ({byt / med(r_['hbm']) / 1e6:4.2f} TB/s)   behind the GEMM {med(r_['GK']) - g0:6.2f} us", flush=True)
