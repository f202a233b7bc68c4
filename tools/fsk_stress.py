"""Dev tool (GPU box): stress of the fused split-K hand-over (gemm_s8_sp256<..., FSK>).  For each shape a pool of operand pairs; references with PQ_FSK=0;
then many back-to-back launches that alternate the pairs on ONE workspace — every launch overwrites the slabs the previous one read, so a stale cached slab
line, a flag seen too early or a missed re-zeroing of the flags shows up as a mismatch — with PQ_FSK = the plan / 2 / 4, also inside a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import protoquant_amd as pq
from protoquant_amd import _lib

SHAPES = [(2048, 4096, 11008, ""), (2048, 4096, 10240, "4"), (4096, 2048, 16384, ""), (4096, 1024, 28672, "4"), (4096, 4096, 4096, "2"), (1000, 3000, 2560, "2"),
          (4096, 1024, 8192, "4"), (8192, 8192, 2560, "2"), (300, 520, 1280, "2"), (4096, 28672, 1280, "2")]
NPOOL, REPS = 4, 120
bad_total = 0
for M, N, K, mode in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    pool = []
    _lib.set_option("PQ_FSK", "0")
    for p in range(NPOOL):
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda", generator=g); b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
        xs = torch.rand(M, device="cuda", generator=g) * 0.1; ws = torch.rand(N, device="cuda", generator=g) * 0.01
        pool.append((a, xs, b, ws, pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone()))
    _lib.set_option("PQ_FSK", mode)
    _lib.set_option("PQ_FSK_SYMMETRIC", os.environ.get("FSK_STRESS_SYMMETRIC", ""))     # (the default form is the ticket form; tests/test_gpu_fsk_stress.py runs both in the gpu suite)
    if _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == 0:
        print(f"{M}x{N}x{K} PQ_FSK={mode or 'plan'}: refused by the planner (grid > CUs in the symmetric form)"); continue
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) > 0, (M, N, K, mode)
    nbad = 0
    for i in range(REPS):
        a, xs, b, ws, ref = pool[i % NPOOL]
        y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16)
        if i % 4 == 3 or i >= REPS - NPOOL:
            nbad += int((y.view(torch.int16) != ref.view(torch.int16)).sum().item())
    # the same alternation replayed from a hipGraph (the launcher's flag memset is a graph node)
    outs = [torch.empty_like(pool[0][4]) for _ in range(NPOOL)]
    s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        for p in range(NPOOL): pq.qlinear_s8(*pool[p][:4], None, torch.bfloat16, out=outs[p])
    torch.cuda.current_stream().wait_stream(s2)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for rep in range(3):
            for p in range(NPOOL): pq.qlinear_s8(*pool[p][:4], None, torch.bfloat16, out=outs[p])
    for rep in range(10):
        gr.replay()
    torch.cuda.synchronize()
    for p in range(NPOOL):
        nbad += int((outs[p].view(torch.int16) != pool[p][4].view(torch.int16)).sum().item())
    bad_total += nbad
    print(f"{M}x{N}x{K} PQ_FSK={mode or 'plan'}: {REPS} alternating launches + 120 graph launches, mismatching elements: {nbad}", flush=True)
    _lib.set_option("PQ_FSK", "")
print("FSK STRESS", "CLEAN" if bad_total == 0 else f"FAILED ({bad_total})")
