"""Dev tool: repeat the GEMM many times per shape under load and compare every result bitwise (race screen)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import protoquant_amd as pq
SHAPES = [(4096, 4096, 4096, 300), (4096, 1024, 8192, 300), (2048, 4096, 11008, 200), (1000, 3000, 1024, 400), (257, 511, 256, 800),
          (512, 4096, 4096, 400), (8192, 8192, 1024, 100), (4096, 4096, 128, 500), (4096, 4096, 256, 500), (4096, 4096, 384, 500), (4096, 4096, 512, 400), (4096, 8192, 640, 300),
          (4096, 1024, 4096, 300), (4096, 1024, 28672, 60), (300, 1000, 384, 500), (16, 4096, 4096, 800), (64, 4096, 14336, 400),
          (1, 128256, 4096, 100), (33, 777, 512, 800), (2048, 11008, 4096, 200), (1024, 1024, 8192, 300),
          # round 4: the 64-row ring tiles (rotated K walk, one barrier per two K-tiles) and the rotated ring128 / 128 x 256 loaders
          (128, 4096, 4096, 600), (256, 4096, 14336, 300), (500, 4000, 1152, 600), (64, 6144, 4096, 600), (200, 1000, 2048, 800), (1024, 1024, 4096, 500),
          (4096, 1280, 8192, 200), (2048, 1024, 8192, 300), (384, 4096, 128, 800), (70, 130, 384, 800)]
bad_total = 0
for M, N, K, reps in SHAPES:
    torch.manual_seed(M + N + K)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda"); b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
    xs = torch.rand(M, device="cuda"); ws = torch.rand(N, device="cuda")
    ref_acc = pq.int_mm(a, b).clone()
    rows = torch.randperm(M, device="cuda")[:64]
    want = (a[rows].to(torch.int64).cpu().numpy() @ b.to(torch.int64).cpu().numpy().T)
    assert np.array_equal(ref_acc[rows].cpu().numpy().astype(np.int64), want), "reference rows differ from int64 matmul"
    ref_y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone()
    nbad = 0
    for i in range(reps):
        acc = pq.int_mm(a, b)
        y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16)
        if i % 10 == 9 or i == reps - 1:
            nbad += int((acc != ref_acc).sum().item()) + int((y.view(torch.int16) != ref_y.view(torch.int16)).sum().item())
    bad_total += nbad
    print(f"{M}x{N}x{K}: {reps} reps, mismatching elements: {nbad}")
# round 5: the GEMM on STACKED activation codes (ring tiles walking K-slabs in place; the 4-slice fused split-K plan of the 70B `down` shard runs in the loop above)
for M, N, K, G, reps in [(4096, 1024, 28672, 8, 60), (4096, 1024, 8192, 8, 300), (1024, 1024, 8192, 8, 400), (512, 4096, 4096, 4, 400), (380, 484, 512, 4, 800), (2048, 1024, 2048, 8, 400)]:
    torch.manual_seed(M + N + K + G)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda"); b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
    xs = torch.rand(M, device="cuda"); ws = torch.rand(N, device="cuda")
    stk = a.reshape(M, G, K // G).permute(1, 0, 2).contiguous()
    ref_y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone()
    nbad = 0
    for i in range(reps):
        y = pq.qlinear_s8_kslabs(stk, xs, b, ws, None, torch.bfloat16)
        if i % 10 == 9 or i == reps - 1:
            nbad += int((y.view(torch.int16) != ref_y.view(torch.int16)).sum().item())
    bad_total += nbad
    print(f"{M}x{N}x{K} stacked in {G} K-slabs: {reps} reps, mismatching elements: {nbad}")
# round 6: (a) the fused split-K of the 256 x 256 tile walking STACKED blocks in place (kloop_p3_asm<5>: the activation cursor jumps at the slab boundaries) — planned and forced
# slice counts, alternating operand pairs on ONE workspace (every launch overwrites the slabs the previous one read); (b) the 128 x 160 ring tile (forced) and the staged
# epilogue's padded row stride; (c) the half-swap epilogue swizzle runs in every loop of this file
from protoquant_amd import _lib as _L
for M, N, K, G, fsk, reps in [(4096, 1024, 28672, 8, "", 80), (2048, 4096, 11008, 2, "", 150), (1000, 520, 8192, 8, "4", 300), (300, 300, 4096, 4, "2", 500), (513, 257, 8192, 8, "8", 300),
                              (2048, 1024, 28672, 4, "4", 100)]:
    torch.manual_seed(M + N + K + G + 6)
    pool = []
    for p_ in range(3):
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda"); b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
        xs = torch.rand(M, device="cuda"); ws = torch.rand(N, device="cuda")
        _L.set_option("PQ_FSK", "0")
        ref = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone()
        pool.append((a.reshape(M, G, K // G).permute(1, 0, 2).contiguous(), xs, b, ws, ref))
    _L.set_option("PQ_FSK", fsk)
    way = _L.lib().pq_kslabs_way_name(pool[0][0].data_ptr(), K // G, M * (K // G), K // G, pool[0][2].data_ptr(), K, M, N, K, 1 << 40).decode()
    nbad = 0
    for i in range(reps):
        stk, xs, b, ws, ref = pool[i % 3]
        y = pq.qlinear_s8_kslabs(stk, xs, b, ws, None, torch.bfloat16)
        if i % 6 >= 3 or i >= reps - 3:
            nbad += int((y.view(torch.int16) != ref.view(torch.int16)).sum().item())
    _L.set_option("PQ_FSK", "")
    bad_total += nbad
    print(f"{M}x{N}x{K} stacked in {G} K-slabs, PQ_FSK={fsk or 'plan'} [{way}], 3 alternating operand sets: {reps} reps, mismatching elements: {nbad}")
for M, N, K, reps in [(4096, 1280, 8192, 200), (700, 500, 1024, 600), (4096, 2560, 4096, 200), (130, 170, 384, 800)]:
    torch.manual_seed(M + N + K + 60)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda"); b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
    xs = torch.rand(M, device="cuda"); ws = torch.rand(N, device="cuda")
    ref_y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16).clone(); ref_f = pq.qlinear_s8(a, xs, b, ws, None, torch.float32).clone()
    _L.set_option("PQ_FORCE_VARIANT", "ring128x160")
    nbad = 0
    for i in range(reps):
        y = pq.qlinear_s8(a, xs, b, ws, None, torch.bfloat16); yf = pq.qlinear_s8(a, xs, b, ws, None, torch.float32)
        if i % 10 == 9 or i == reps - 1:
            nbad += int((y.view(torch.int16) != ref_y.view(torch.int16)).sum().item()) + int((yf.view(torch.int32) != ref_f.view(torch.int32)).sum().item())
    _L.set_option("PQ_FORCE_VARIANT", "")
    bad_total += nbad
    print(f"{M}x{N}x{K} forced 128 x 160 ring tile (bf16 and f32 out): {reps} reps, mismatching elements: {nbad}")
print("RACE SCREEN", "CLEAN" if bad_total == 0 else f"FAILED ({bad_total})")
