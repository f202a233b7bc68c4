"""Dev tool: differential fuzz of the GEMM variants.  Random shapes / leading dimensions / dtypes / bias; every variant
(auto dispatch incl. tail split, split-K with workspace, ring, skinny, ...) must reproduce the generic kernel bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import protoquant_amd as pq
from protoquant_amd import _lib as _pqlib  # noqa: E402

VARIANTS = ["", "sp256_16", "sp128_16", "sp128x128", "ring128", "ring64x128", "ring64x64", "ring128x160", "skinny"]
DT = [torch.bfloat16, torch.float16, torch.float32]
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
budget = float(os.environ.get("FUZZ_SECONDS", "120"))
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = rng.integers(0, 4)
    if kind == 0: M, N = int(rng.integers(1, 70)), int(rng.integers(1, 9000))             # decode-like
    elif kind == 1: M, N = int(rng.integers(1, 1200)), int(rng.integers(1, 1200))          # small
    elif kind == 2: M, N = int(rng.integers(256, 3000)), int(rng.integers(256, 6000))      # multi-tile, ragged
    else: M, N = int(rng.integers(1, 5) * 1024 + rng.integers(-3, 4)), int(rng.integers(1, 9) * 512 + rng.integers(-3, 4))
    K = int(rng.integers(1, 24)) * 128 if rng.random() < 0.8 else int(rng.integers(24, 130)) * 128
    pad_a, pad_b = int(rng.integers(0, 3)) * 16, int(rng.integers(0, 3)) * 16
    a = torch.randint(-128, 128, (M, K + pad_a), dtype=torch.int8, device="cuda")[:, :K]
    b = torch.randint(-128, 128, (N, K + pad_b), dtype=torch.int8, device="cuda")[:, :K]
    xs = torch.rand(M, device="cuda") + 0.01; ws = torch.rand(N, device="cuda") * 0.01 + 1e-4
    dt = DT[int(rng.integers(0, 3))]
    bias = torch.randn(N, device="cuda").to(dt) if rng.random() < 0.5 else None
    _pqlib.set_option("PQ_FORCE_VARIANT", "generic")
    ref_y = pq.qlinear_s8(a, xs, b, ws, bias, dt).clone(); ref_acc = pq.int_mm(a, b).clone()
    pad_y, off_y = int(rng.choice([0, 0, 8, 24, 3])), int(rng.choice([0, 0, 8, 1]))
    ybig = torch.full((M, N + pad_y + off_y), 7.0, dtype=dt, device="cuda")
    for v in VARIANTS:
        _pqlib.set_option("PQ_FORCE_VARIANT", v)
        yv = ybig[:, off_y:off_y + N]                  # the output as a window of a wider matrix: ld > N, maybe unaligned
        pq.qlinear_s8(a, xs, b, ws, bias, dt, out=yv)
        y = yv.contiguous(); acc = pq.int_mm(a, b)
        yt = pq.qlinear_s8_t(a, xs, b, ws, bias, dt)            # the transposed-output form must hold the same bits
        if pad_y + off_y:
            outside = torch.cat([ybig[:, :off_y], ybig[:, off_y + N:]], dim=1)
            if not bool((outside == 7.0).all()):
                bad += 1; print(f"WROTE OUTSIDE variant={v or 'auto'} M={M} N={N} K={K} pad_y={pad_y} off_y={off_y}")
        iv = torch.int32 if dt == torch.float32 else torch.int16
        d = int((y.view(iv) != ref_y.view(iv)).sum()) + int((acc != ref_acc).sum()) + int((yt.t().contiguous().view(iv) != ref_y.view(iv)).sum())
        if d:
            bad += 1
            print(f"MISMATCH variant={v or 'auto'} M={M} N={N} K={K} pads=({pad_a},{pad_b}) dtype={dt} bias={bias is not None}: {d} elements")
    n += 1
_pqlib.set_option("PQ_FORCE_VARIANT", "")
print(f"fuzz: {n} problems x {len(VARIANTS)} variants in {time.time() - t0:.0f} s, mismatching runs: {bad}")
print("FUZZ", "CLEAN" if bad == 0 else "FAILED")
