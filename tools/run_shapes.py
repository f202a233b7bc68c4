"""Dev tool (GPU box): launch pq_qlinear_s8 `--reps` times on each of the given shapes (for rocprofv3 --kernel-trace / --pmc runs: tools/pmc_shapes.sh).
usage: python3 tools/run_shapes.py --shapes 128x4096x4096,512x4096x4096 [--reps 200] [--opts PQ_X=v,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import protoquant_amd as pq  # noqa: E402
from protoquant_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", required=True)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--opts", default="")
a = ap.parse_args()
for o in filter(None, a.opts.split(",")):
    _lib.set_option(*o.split("="))
for shp in a.shapes.split(","):
    M, N, K = (int(v) for v in shp.split("x"))
    torch.manual_seed(1)
    xq = (torch.randn(M, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
    wq = (torch.randn(N, K, device="cuda") * 28).round().clamp(-127, 127).to(torch.int8)
    xs = torch.rand(M, device="cuda") * 1e-2 + 1e-3
    ws = torch.rand(N, device="cuda") * 1e-2 + 1e-3
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    for _ in range(a.reps):
        pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=y)
    torch.cuda.synchronize()
    print(shp, _lib.lib().pq_gemm_variant_name(M, N, K, K, K).decode(), flush=True)
