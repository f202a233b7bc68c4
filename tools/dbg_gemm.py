import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import protoquant_amd as pq
os.environ["PQ_FORCE_VARIANT"] = "sp256_16"
for (M, N, K) in ((256, 256, 128), (256, 256, 256), (256, 256, 512), (512, 512, 1024)):
    rng = np.random.default_rng(0)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    want = a.astype(np.int64) @ b.astype(np.int64).T
    got = pq.int_mm(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy().astype(np.int64)
    bad = got != want
    print(M, N, K, "bad:", bad.sum(), "of", bad.size, "max abs err", np.abs(got - want).max())
    if bad.any():
        idx = np.argwhere(bad)
        print("  bad rows(m) mod 64 hist:", np.bincount(idx[:, 0] % 64, minlength=64).tolist())
        print("  bad cols(n) mod 128 hist (16-bins):", np.bincount((idx[:, 1] % 128) // 16, minlength=8).tolist())
        print("  sample:", idx[:4].tolist(), got[bad][:4], want[bad][:4])
    y = pq.qlinear_s8(torch.from_numpy(a).cuda(), torch.ones(M, device="cuda"), torch.from_numpy(b).cuda(), torch.ones(N, device="cuda"), None, torch.float32).cpu().numpy()
    print("   f32-out bad:", (y != want.astype(np.float32)).sum())
M = N = 256; K = 128
a = np.ones((M, K), np.int8); b = np.ones((N, K), np.int8)
xs = (1 + np.arange(M)).astype(np.float32); ws = (1000 * (1 + np.arange(N))).astype(np.float32)
y = pq.qlinear_s8(torch.from_numpy(a).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws).cuda(), None, torch.float32).cpu().numpy()
r = y / 128.0
print("row 0, first 12 (expect 1000*(n+1)):", r[0, :12])
print("row 5, first 8 (expect 6*1000*(n+1)):", r[5, :8])
print("col 0, rows 0..8 (expect (m+1)*1000):", r[:9, 0])
for K in (128, 256, 512):
    a = np.ones((256, K), np.int8); b = np.ones((256, K), np.int8)
    y = pq.qlinear_s8(torch.from_numpy(a).cuda(), torch.ones(256, device="cuda"), torch.from_numpy(b).cuda(), torch.ones(256, device="cuda"), None, torch.float32).cpu().numpy()
    v, c = np.unique(y, return_counts=True)
    print("K", K, "values:", dict(zip(v.tolist(), c.tolist())))
    yb = pq.qlinear_s8(torch.from_numpy(a).cuda(), torch.ones(256, device="cuda"), torch.from_numpy(b).cuda(), torch.ones(256, device="cuda"), None, torch.bfloat16).float().cpu().numpy()
    v, c = np.unique(yb, return_counts=True)
    print("   bf16 values:", dict(zip(v.tolist(), c.tolist())))
