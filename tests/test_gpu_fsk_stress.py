"""GPU: stress of the fused split-K hand-over (gemm_s8_sp256<..., FSK>: the K-slices of a tile hand their partial sums over INSIDE the kernel with
write-through stores + vmcnt(0) + barrier + ticket, agent-scope loads on the reading side — a hand-rolled sequence whose visibility assumption is
checked empirically, so it is checked on every GPU run).  Many back-to-back launches alternate operand sets on ONE workspace (every launch overwrites the
slabs the previous one read: a stale cached slab line, a ticket seen too early or a missed re-zeroing shows up as a mismatch), also replayed from a
hipGraph, in the default ticket form and the opt-in symmetric forms; and two streams run fused split-K GEMMs CONCURRENTLY (separate workspaces) in the
ticket form, which must neither hang nor differ (the symmetric forms are not allowed to: include/pq_hip.h)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NPOOL = 3


def _pool(pq, _lib, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    _lib.set_option("PQ_FSK", "0")
    pool = []
    for _ in range(NPOOL):
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
        b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
        xs = torch.rand(M, device="cuda", generator=g) * 0.1
        ws = torch.rand(N, device="cuda", generator=g) * 0.01
        pool.append((a, xs, b, ws, pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone()))
    return pool


@pytest.mark.parametrize("M,N,K,fsk,sym", [(2048, 4096, 11008, "", False), (2048, 4096, 11008, "", True), (4096, 1024, 28672, "4", False),
                                           (4096, 1024, 28672, "4", True), (4096, 4096, 4096, "2", False), (1000, 3000, 2560, "2", True),
                                           (4096, 1024, 8192, "4", True), (300, 520, 1920, "3", False), (2048, 4096, 11520, "3", False)])
def test_alternating_operands_on_one_workspace(M, N, K, fsk, sym, pq_opt):
    import protoquant_amd as pq
    from protoquant_amd import _lib
    pool = _pool(pq, _lib, M, N, K, M + N + K)
    pq_opt("PQ_FSK", fsk)
    pq_opt("PQ_FSK_SYMMETRIC", "1" if sym else "")
    pq_opt("PQ_FSK_FENCED", "1" if (not sym and fsk == "3") else "")       # one case through the fenced fallback (buffer_wbl2 / buffer_inv as well): same bits
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) > 0
    nbad = 0
    for i in range(60):
        a, xs, b, ws, ref = pool[i % NPOOL]
        y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16)
        if i % 4 == 3 or i >= 60 - NPOOL:
            nbad += int((y.view(torch.int16) != ref.view(torch.int16)).sum().item())
    # the same alternation replayed from a hipGraph (the launcher's ticket memset is a graph node)
    outs = [torch.empty_like(pool[0][4]) for _ in range(NPOOL)]
    s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        for p in range(NPOOL):
            pq.qlinear_s8(*pool[p][:4], None, torch.bfloat16, out=outs[p])
    torch.cuda.current_stream().wait_stream(s2)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(3):
            for p in range(NPOOL):
                pq.qlinear_s8(*pool[p][:4], None, torch.bfloat16, out=outs[p])
    for _ in range(8):
        gr.replay()
    torch.cuda.synchronize()
    for p in range(NPOOL):
        nbad += int((outs[p].view(torch.int16) != pool[p][4].view(torch.int16)).sum().item())
    assert nbad == 0


_TWO_STREAMS = r"""
import sys, torch
sys.path.insert(0, %r)
import protoquant_amd as pq
from protoquant_amd import _lib
shapes = [(2048, 4096, 11008), (2048, 4096, 14336)]
g = torch.Generator(device="cuda").manual_seed(7)
jobs = []
_lib.set_option("PQ_FSK", "0")
for (M, N, K) in shapes:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda", generator=g); b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.1; ws = torch.rand(N, device="cuda", generator=g) * 0.01
    jobs.append((a, xs, b, ws, pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone(), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")))
_lib.set_option("PQ_FSK", "")
assert all(_lib.lib().pq_qlinear_workspace_bytes(*s) > 0 for s in shapes)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
bad = 0
for rnd in range(6):
    for it in range(25):                      # both streams hold a queue of fused split-K GEMMs (each its own workspace: out= keeps one per shape)
        for s, j in zip(streams, jobs if rnd %% 2 == 0 else jobs[::-1]):
            with torch.cuda.stream(s):
                pq.qlinear_s8(*j[:4], None, torch.bfloat16, out=j[5])
    torch.cuda.synchronize()
    for j in jobs:
        bad += int((j[5].view(torch.int16) != j[4].view(torch.int16)).sum().item())
print("TWO_STREAMS", "CLEAN" if bad == 0 else "BAD %%d" %% bad)
"""


def test_two_streams_of_fused_splitk_gemms_make_progress_and_agree():
    """ADVICE (round 3): two fused split-K GEMMs on different streams with separate workspaces are allowed by pq_hip.h; with the symmetric exchange they
    could fill the CUs with workgroups that wait for partners which cannot be scheduled.  The default (ticket) form never waits for a workgroup that is
    not running.  Runs in a child process under a timeout, so a regression shows up as a failure, not as a stuck test session."""
    r = subprocess.run([sys.executable, "-c", _TWO_STREAMS % ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TWO_STREAMS CLEAN" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
