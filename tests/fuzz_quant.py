"""Fuzz harness (test infrastructure; run by tests/test_gpu_fuzz.py, or directly: `FUZZ_SECONDS=150 python -m tests.fuzz_quant`):
random shapes / strides / alignments / special values through quantize (both axes), dequantize, silu_mul_quantize,
rmsnorm_quantize, the split quantisation halves of the int8-code exchange (random column blocks), the fused GEMM epilogue and the GEMM on stacked
code blocks, each compared bit for bit with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import protoquant_amd as pq
from protoquant_amd import _lib as _pqlib  # noqa: E402
from oracle import c_oracle as C, qspec_numpy as Q
from tests.gpu_util import TD, bits, to_gpu

rng = None
SPECIAL = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, -1e-30, 3e38, -3e38, 65504.0, 88.0, -88.0, 1e-40], np.float32)


def rand(rows, cols, scale):
    a = (rng.standard_normal((rows, cols)) * scale).astype(np.float32)
    if rng.random() < 0.3 and a.size:
        k = int(rng.integers(1, 6))
        a.flat[rng.integers(0, a.size, k)] = rng.choice(SPECIAL, k)
    if rng.random() < 0.1 and rows:
        a[int(rng.integers(0, rows))] = 0
    return a


def eqb(got, want, what, ctx, nan_ok=None):
    g = bits(got); w = np.asarray(want); w = w.view(np.uint32) if w.dtype == np.float32 else w
    m = np.ones(g.shape, bool) if nan_ok is None else ~nan_ok
    if g.shape != w.shape or np.count_nonzero(g[m] != w[m]):
        print("MISMATCH", what, ctx); return 1
    return 0


def strided(a_np, code, pad, off):
    """GPU tensor equal to a_np but living inside a wider buffer (ld = cols + pad, column offset off)."""
    r, c = a_np.shape
    t = to_gpu(a_np, code)
    big = torch.zeros((r, c + pad + off), dtype=TD[code], device="cuda")
    big[:, off:off + c] = t
    return big[:, off:off + c]


def run(budget, seed):
  global rng
  rng = np.random.default_rng(seed)
  t0, n, bad = time.time(), 0, 0
  while time.time() - t0 < budget:
      code = int(rng.integers(0, 3))
      rows = int(rng.integers(1, 200)); cols = int(rng.choice([rng.integers(1, 64), rng.integers(1, 3000), rng.integers(1, 9) * 1024 + rng.integers(-8, 9), 11008, 14336]))
      cols = max(cols, 1)
      pad = int(rng.choice([0, 0, 8, 16, 3])); off = int(rng.choice([0, 0, 8, 1]))
      ctx = f"code={code} rows={rows} cols={cols} pad={pad} off={off}"
      x = Q.from_f32(rand(rows, cols, float(rng.choice([0.01, 1.0, 30.0]))), code)
      xt = strided(x, code, pad, off)
      q = pq.quantize(xt); wq, ws = C.quant_rowwise(x, code)
      bad += eqb(q.int_data, wq, "K1 q", ctx) + eqb(q.scale, ws, "K1 s", ctx)
      nan = np.isnan(Q.to_f32(C.dequant(wq, ws, 1, code), code))
      bad += eqb(pq.dequantize(q), C.dequant(wq, ws, 1, code), "dequant", ctx, nan)
      if rows * cols < 400000:
          qc = pq.quantize(xt, axis=0); cq, cs = C.quant_colwise(x, code)
          bad += eqb(qc.int_data, cq, "K2 q", ctx) + eqb(qc.scale, cs, "K2 s", ctx)
      u = Q.from_f32(rand(rows, cols, 1.0), code)
      g = Q.from_f32(rand(rows, cols, float(rng.choice([1.0, 4.0, 40.0]))), code)
      sq, ss, sh = C.silu_mul_quant_rowwise(g, u, code)
      qt, h = pq.silu_mul_quantize(strided(g, code, pad, off), strided(u, code, int(rng.choice([0, 16])), 0), return_h=True)
      bad += eqb(qt.int_data, sq, "K1s q", ctx) + eqb(qt.scale, ss, "K1s s", ctx) + eqb(h, sh, "K1s h", ctx, np.isnan(Q.to_f32(sh, code)))
      # round 5: the two halves of K1 / K1s over a random column split (strided, ragged, unaligned blocks): integer max of the block amaxes, encode per block
      parts = int(rng.integers(1, 5))
      cuts = sorted(set([0, cols] + [int(c) for c in rng.integers(0, cols + 1, parts - 1)]))
      blocks = [(a_, b_) for a_, b_ in zip(cuts, cuts[1:]) if b_ > a_]
      gt_, ut_ = strided(g, code, pad, off), strided(u, code, 0, 0)
      am_x = torch.stack([pq.rowamax(xt[:, a_:b_]) for a_, b_ in blocks]).max(dim=0).values
      am_h = torch.stack([pq.silu_mul_rowamax(gt_[:, a_:b_], ut_[:, a_:b_]) for a_, b_ in blocks]).max(dim=0).values
      qx2 = torch.empty((rows, cols), dtype=torch.int8, device="cuda"); qh2 = torch.empty_like(qx2)
      for a_, b_ in blocks:
          sx2 = pq.quantize_with_amax(xt[:, a_:b_], am_x, out=qx2[:, a_:b_]).scale
          sh2 = pq.silu_mul_quantize_with_amax(gt_[:, a_:b_], ut_[:, a_:b_], am_h, out=qh2[:, a_:b_]).scale
      bad += eqb(qx2, wq, "K1 split q", ctx) + eqb(sx2, ws, "K1 split s", ctx) + eqb(qh2, sq, "K1s split q", ctx) + eqb(sh2, ss, "K1s split s", ctx)
      w = Q.from_f32((1 + 0.2 * rng.standard_normal(cols)).astype(np.float32), code)
      eps = float(rng.choice([1e-6, 1e-5, 0.0]))
      nq, ns, nh, _ = C.rmsnorm_quant_rowwise(x, w, eps, code)
      qt, h = pq.rmsnorm_quantize(xt, to_gpu(w, code), eps, return_h=True)
      bad += eqb(qt.int_data, nq, "K1n q", ctx) + eqb(qt.scale, ns, "K1n s", ctx) + eqb(h, nh, "K1n h", ctx, np.isnan(Q.to_f32(nh, code)))
      # the fused epilogue (E1-E4) of every GEMM variant the dispatcher may pick at this size, against the numpy oracle
      M2, N2, K2 = int(rng.integers(1, 400)), int(rng.integers(1, 700)), int(rng.integers(1, 9)) * 128
      a = rng.integers(-128, 128, (M2, K2), dtype=np.int8); b = rng.integers(-128, 128, (N2, K2), dtype=np.int8)
      acc = (a.astype(np.int32) @ b.astype(np.int32).T)
      xs2 = (rng.random(M2).astype(np.float32) + 1e-3) * float(rng.choice([1e-3, 1.0, 50.0])); ws2 = rng.random(N2).astype(np.float32) * 0.02 + 1e-5
      bv = Q.from_f32(rng.standard_normal(N2).astype(np.float32), code) if rng.random() < 0.5 else None
      want = Q.epilogue(acc, xs2, ws2, bv, code)
      for v in ("", "generic", "ring128", "sp256_16", "ring128x160"):
          _pqlib.set_option("PQ_FORCE_VARIANT", v)
          got = pq.qlinear_s8(torch.from_numpy(a).cuda(), torch.from_numpy(xs2).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws2).cuda(),
                              to_gpu(bv, code) if bv is not None else None, TD[code])
          bad += eqb(got, want, f"epilogue[{v or 'auto'}]", f"code={code} M={M2} N={N2} K={K2} bias={bv is not None}")
      _pqlib.set_option("PQ_FORCE_VARIANT", "")
      # round 5: the same product with the activation codes STACKED in G K-slabs (ring tiles walk them in place, everything else takes the layout pass)
      G2 = int(rng.choice([g_ for g_ in (1, 2, 4, 8) if (K2 // 128) % g_ == 0]))
      stk = torch.from_numpy(np.ascontiguousarray(a.reshape(M2, G2, K2 // G2).transpose(1, 0, 2))).cuda()
      got = pq.qlinear_s8_kslabs(stk, torch.from_numpy(xs2).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws2).cuda(), to_gpu(bv, code) if bv is not None else None, TD[code])
      bad += eqb(got, want, "epilogue[kslabs]", f"code={code} M={M2} N={N2} K={K2} G={G2} bias={bv is not None}")
      # round 6: a FORCED fused split-K (f ticket slices of the 256 x 256 tile) on stacked blocks — slab counts 1 .. 8, slabs of 4 .. 9 K-tiles, slices that cover 1 / 2 / 4 slabs
      # or a share of one: the asm K-loop's activation cursor jumps at the slab boundaries.  Against the int32 matmul + E1-E4 of the numpy oracle.
      if n % 3 == 0:
          f3 = int(rng.choice([2, 4, 8])); G3 = int(rng.choice([1, 2, 4, 8])); tps3 = int(rng.integers(4, 10))
          while G3 * tps3 % f3 or G3 * tps3 // f3 < 5:
              tps3 += 1
          K3 = G3 * tps3 * 128
          M3, N3 = int(rng.integers(65, 600)), int(rng.integers(129, 600))
          a3 = rng.integers(-128, 128, (M3, K3), dtype=np.int8); b3 = rng.integers(-128, 128, (N3, K3), dtype=np.int8)
          xs3 = rng.random(M3).astype(np.float32) + 1e-3; ws3 = rng.random(N3).astype(np.float32) * 0.02 + 1e-5
          want3 = Q.epilogue(a3.astype(np.int32) @ b3.astype(np.int32).T, xs3, ws3, None, code)
          stk3 = torch.from_numpy(np.ascontiguousarray(a3.reshape(M3, G3, K3 // G3).transpose(1, 0, 2))).cuda()
          _pqlib.set_option("PQ_FSK", str(f3))
          try:
              got3 = pq.qlinear_s8_kslabs(stk3, torch.from_numpy(xs3).cuda(), torch.from_numpy(b3).cuda(), torch.from_numpy(ws3).cuda(), None, TD[code])
          finally:
              _pqlib.set_option("PQ_FSK", "")
          bad += eqb(got3, want3, "epilogue[kslabs, forced fsk]", f"code={code} M={M3} N={N3} K={K3} G={G3} f={f3}")
      n += 1
  print(f"fuzz_quant: {n} problems in {time.time() - t0:.0f} s, mismatches: {bad}")
  return n, bad


if __name__ == "__main__":
    n_, bad_ = run(float(os.environ.get("FUZZ_SECONDS", "90")), int(os.environ.get("FUZZ_SEED", "2")))
    print("FUZZ", "CLEAN" if bad_ == 0 else "FAILED")
    sys.exit(1 if bad_ else 0)
