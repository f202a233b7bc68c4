"""Worker of tests/test_gpu_two_graphs.py (run as a subprocess under a timeout: a regression here is a HANG).  Captures a hipGraph around one fused split-K qlinear (a launch that
zeroes its tile tickets first), THEN captures other graphs without such a launch, and replays them in alternation, synchronising after every replay; prints OK when every replay
finished and every output kept its bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import protoquant_amd as pq  # noqa: E402
from protoquant_amd import _lib  # noqa: E402

M, N, K = 1536, 3200, 11008          # the shape of the find: 78 tiles of 256 x 256, two ticket slices where the fused split-K is planned
torch.manual_seed(3)
dev = torch.device("cuda:0")
xq = (torch.randn(M, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
wq = (torch.randn(N, K, device=dev) * 28).round().clamp(-127, 127).to(torch.int8)
xs, ws = torch.rand(M, device=dev) * 1e-2 + 1e-3, torch.rand(N, device=dev) * 1e-2 + 1e-3
x16 = torch.randn(M, K, device=dev).to(torch.bfloat16)


def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g


_lib.set_option("PQ_FSK", "2")                       # two ticket slices, whatever the planner of this build would pick for the shape
assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) > 0
y_fsk = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
ref = pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16).clone()
g_fsk = capture(lambda: pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=y_fsk))
_lib.set_option("PQ_FSK", "0")                       # every graph captured from here on is free of the ticket launch
y_b = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
g_b = capture(lambda: pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=y_b))
q_out = torch.empty((M, K), dtype=torch.int8, device=dev)
g_c = capture(lambda: pq.quantize(x16))              # a third graph: K1 alone
_lib.set_option("PQ_FSK", "")
for it in range(30):
    if it % 3 == 0:
        y_fsk.zero_(); y_b.zero_()
    for g in (g_fsk, g_b, g_c):
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(y_fsk.view(torch.int16), ref.view(torch.int16)) and torch.equal(y_b.view(torch.int16), ref.view(torch.int16)), it
print("OK two graphs")
