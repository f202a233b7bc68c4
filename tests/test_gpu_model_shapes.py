"""-m gpu: parity at the REAL shapes of BASELINE.json configs[3] (Llama-3-8B prefill, seq 4096) and configs[4]
(Llama-3-70B column shards, 8 GPUs): the shapes where the dispatch changes (tail split, ring tiles, split-K on/off).

Per shape (sampled-row pattern of test_full_size_cfg2_properties):
  * K1: xq / xs of ALL 4096 token rows bit-exact vs the C oracle;
  * K3: the int32 accumulator of 64 sampled rows x ALL N columns == torch._int_mm on the host (the primitive the contract
    names) — and == an int64 matmul on a column sample; the full column checksum sum_m acc[m, :] == (sum_m xq[m, :]) . wq^T;
  * K4: y of those rows bit-identical to the QSPEC epilogue (numpy oracle), with a bias on one shape per config;
  * the variant string the library reports for the shape.
Weights are synthetic int8 codes (gaussian, sigma 28, generated on the GPU and copied to the host for the oracle)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import qspec_numpy as Q
from tests.gpu_util import bits, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pq():
    import protoquant_amd
    from protoquant_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return protoquant_amd


_X = {}


def _activation(pq, K):
    """x[4096, K] bf16 (seeded on the host), its GPU quantisation checked in full against the C oracle — once per K."""
    if K not in _X:
        g = torch.Generator().manual_seed(4242 + K)
        x = torch.randn(4096, K, generator=g).to(torch.bfloat16)
        q = pq.quantize(x.cuda())
        xq, xs = C.quant_rowwise(bits(x), 0)
        same(q.int_data, xq, f"xq K={K}"); same(q.scale, xs, f"xs K={K}")
        _X.clear()                                   # keep one activation alive at a time
        _X[K] = (q, xq, xs)
    return _X[K]


def _check(pq, M, N, K, want_variant, bias=False, nrows=64):
    from protoquant_amd import _lib
    q, xq, xs = _activation(pq, K)
    gen = torch.Generator(device="cuda").manual_seed(N * 7 + K)
    wq_t = (torch.randn(N, K, device="cuda", generator=gen) * 28).round().clamp(-127, 127).to(torch.int8)
    ws_t = torch.rand(N, device="cuda", generator=gen) * 1e-2 + 1e-4
    b_t = (torch.randn(N, device="cuda", generator=gen) * 0.05).to(torch.bfloat16) if bias else None
    name = _lib.lib().pq_gemm_variant_name(M, N, K, K, K).decode()
    assert want_variant in name, (name, want_variant)
    y = pq.qlinear_s8(q.int_data, q.scale, wq_t, ws_t, b_t, torch.bfloat16)
    acc = pq.int_mm(q.int_data, wq_t)
    torch.cuda.synchronize()
    wq = wq_t.cpu()
    rows = np.sort(np.random.default_rng(N + K).choice(M, nrows, replace=False))
    rt = torch.from_numpy(rows).cuda()
    # a3: the contract's primitive on the host, all N columns of the sampled rows
    acc_want = torch._int_mm(torch.from_numpy(np.ascontiguousarray(xq[rows])), wq.t()).numpy()
    cols = np.random.default_rng(K).choice(N, min(N, 257), replace=False)
    assert np.array_equal(acc_want[:, cols].astype(np.int64), xq[rows].astype(np.int64) @ wq.numpy()[cols].astype(np.int64).T)
    same(acc[rt].contiguous(), acc_want, f"acc rows {M}x{N}x{K}")
    # a4: QSPEC epilogue on those accumulators
    y_want = Q.epilogue(acc_want, xs[rows], ws_t.cpu().numpy(), bits(b_t) if bias else None, 0)
    same(y[rt].contiguous(), y_want, f"y rows {M}x{N}x{K}")
    # checksum of checksums over ALL rows (exact in float64: every partial sum is an integer < 2^53)
    colsum = acc.sum(dim=0, dtype=torch.int64).cpu().numpy()
    want = (xq.astype(np.float64).sum(axis=0) @ wq.numpy().astype(np.float64).T).astype(np.int64)
    assert np.array_equal(colsum, want), f"column checksum {M}x{N}x{K}"
    return y, acc


# BASELINE configs[3]: Llama-3-8B, bs 1, seq 4096 -> M = 4096
@pytest.mark.parametrize("N,K,variant,bias", [
    (6144, 4096, "sp256_16x16x64 + sp128 tail (N)", True),      # fused qkv: 384 tiles = 1.5 rounds -> tail split
    (28672, 4096, "sp256_16x16x64", False),                     # fused gate+up: 1792 tiles = 7 full rounds
    (4096, 14336, "sp256_16x16x64", False),                     # down: one round, K = 14336
    (128256, 4096, "sp256_16x16x64 + sp128 tail (N)", False),   # lm_head: 8016 tiles + tail split
    (1024, 4096, "ring128", False),                             # unfused k / v projection
])
def test_llama8b_prefill_shapes(pq, N, K, variant, bias):
    _check(pq, 4096, N, K, variant, bias=bias)


# BASELINE configs[4]: Llama-3-70B, W column-sharded over 8 GPUs: per-GPU shard shapes at M = 4096
@pytest.mark.parametrize("N,K,variant,bias", [
    (1024, 8192, "ring128", True),            # q / o shard
    (3584, 8192, "sp256_16x16x64", False),    # gate / up shard
    (1024, 28672, "ring128", False),          # down shard (column-sharded: full K)
    (16032, 8192, "sp256_16x16x64", False),   # lm_head shard: 128256 / 8 = 16032 = 62.6 tile columns (ragged last tile)
    (128, 8192, "", False),                   # k / v shard (8 KV heads x 128 / 8 GPUs)
    (1280, 8192, "ring128x160", True),        # FUSED q+k+v shard, what bench.py --workload llama70b-shard runs: (8192 + 2 x 1024) / 8 — round 6: 32 x 8 = 256 tiles of 128 x 160
    (7168, 8192, "sp256_16x16x64", False),    # FUSED gate+up shard: 2 x 28672 / 8 (28 tile columns x 16 = 448 tiles: 1.75 rounds)
])
def test_llama70b_shard_shapes(pq, N, K, variant, bias, pq_opt):
    y, acc = _check(pq, 4096, N, K, variant, bias=bias)
    # the same shard through the other split-K setting: bit-identical (qlinear_s8 hands the library the workspace it
    # asks for; PQ_NO_SPLITK forces the single-pass kernels)
    from protoquant_amd import _lib
    used_splitk = _lib.lib().pq_qlinear_workspace_bytes(4096, N, K) > 0
    pq_opt("PQ_NO_SPLITK", "1")
    assert _lib.lib().pq_qlinear_workspace_bytes(4096, N, K) == 0
    y2, _ = _check(pq, 4096, N, K, variant, bias=bias)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), f"split-K {'on' if used_splitk else 'off'} vs forced off"


def test_70b_shard_splitk_forced_on_matches(pq, pq_opt):
    """The slab path at a 70B shard shape where the planner would not pick it by itself: 4096 x 512 x 8192 (a quarter-filled
    grid, K >= 8192 -> 4 slices) against the single-pass result, bit for bit, plus the oracle on sampled rows."""
    from protoquant_amd import _lib
    M, N, K = 4096, 512, 8192
    assert _lib.lib().pq_gemm_variant_name(M, N, K, K, K) == b"ring64x128_16x16x64" and _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == 0   # the default since round 4
    y0, _ = _check(pq, M, N, K, "ring64x128")
    pq_opt("PQ_NO_MIDM", "1")                         # the round-3 plan: 128 x 128 ring tiles' grid is a quarter of the chip -> split-K through the workspace
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) > 0
    y, _ = _check(pq, M, N, K, "")
    pq_opt("PQ_NO_SPLITK", "1")
    y2, _ = _check(pq, M, N, K, "")
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)) and torch.equal(y.view(torch.int16), y0.view(torch.int16))


@pytest.mark.parametrize("H,Kfull,name", [(8192, 8192, "o"), (8192, 28672, "down")])
def test_llama70b_row_sharded_real_shapes(pq, H, Kfull, name):
    """BASELINE configs[4], the row-sharded pairing of bench.py --workload llama70b-shard: `o` as 4096 x 8192 x 1024 and
    `down` as 4096 x 8192 x 3584 per rank.  Two of the eight ranks are built offline on one GPU (RowShardedQLinear.from_linear with
    world = 8), each computes its f32 partial from its own K-slice of the activation; partial(rank 0) + partial(rank 1), in f32, must
    equal the oracle's restatement (full-row weight scales, per-slice activation scales, bias on rank 0) on 64 sampled token rows."""
    from protoquant_amd.sharded import shard_bounds
    M, world = 4096, 8
    g = torch.Generator().manual_seed(70 + Kfull)
    lin = torch.nn.Linear(Kfull, H, bias=True, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_((torch.randn(H, Kfull, generator=g) * 0.02).to(torch.bfloat16))
        lin.bias.copy_((torch.randn(H, generator=g) * 0.05).to(torch.bfloat16))
    lin_g = torch.nn.Linear(Kfull, H, bias=True, dtype=torch.bfloat16, device="cuda")
    lin_g.load_state_dict(lin.state_dict())
    wq, ws = C.quant_rowwise(bits(lin.weight), 0)
    rows = np.sort(np.random.default_rng(Kfull).choice(M, 64, replace=False))
    total, want = None, None
    for r in range(2):
        k0, k1 = shard_bounds(Kfull, world, r)
        assert k1 - k0 == Kfull // world
        x = torch.randn(M, k1 - k0, generator=g).to(torch.bfloat16)          # this rank's local activation slice
        layer = pq.RowShardedQLinear.from_linear(lin_g, world=world, rank=r)
        same(layer.local.wq, np.ascontiguousarray(wq[:, k0:k1]), f"{name}: weight codes of rank {r}")
        same(layer.local.ws, ws, f"{name}: full-row weight scales")
        p = layer.partial(x.cuda())
        assert p.dtype == torch.float32 and p.shape == (M, H)
        total = p if total is None else total + p
        xq, xs = C.quant_rowwise(np.ascontiguousarray(bits(x)[rows]), 0)
        b = lin.bias.detach().float().numpy() if r == 0 else None
        pw = C.qlinear_s8(xq, xs, np.ascontiguousarray(wq[:, k0:k1]), ws, b, 2)
        same(p[torch.from_numpy(rows).cuda()].contiguous(), pw, f"{name}: f32 partial of rank {r}")
        want = pw if want is None else (want + pw).astype(np.float32)
    torch.cuda.synchronize()
    same(total[torch.from_numpy(rows).cuda()].contiguous(), want, f"{name}: sum of two ranks' partials")
    name_v = __import__("protoquant_amd")._lib.lib().pq_gemm_variant_name(M, H, Kfull // world, Kfull // world, Kfull // world).decode()
    assert "sp256" in name_v, name_v


def test_persistent_multi_round_kernel_matches(pq, pq_opt):
    """gemm_s8_p3_persist (opt-in, PQ_SP256_PERSIST=1: one workgroup per CU walks its tiles, the next tile's first K-tile prefetched under
    the epilogue, the asm K-loop entered at the ring phase the previous tile left) against the default one-workgroup-per-tile launch: a
    7-round grid, a ragged one (edge tiles through the direct epilogue) with a bias, and the shortest K the form accepts — bit for bit,
    and the default itself against the oracle on sampled rows."""
    from protoquant_amd import _lib
    for (M, N, K, bias) in ((4096, 28672, 4096, False), (4000, 9000, 1024, True), (4352, 4608, 640, False)):
        y0, acc0 = _check(pq, M, N, K, "sp256", bias=bias, nrows=16) if M == 4096 else (None, None)
        q = None
        g = torch.Generator(device="cuda").manual_seed(N + K)
        xq = (torch.randn(M, K, device="cuda", generator=g) * 28).round().clamp(-127, 127).to(torch.int8)
        wq = (torch.randn(N, K, device="cuda", generator=g) * 28).round().clamp(-127, 127).to(torch.int8)
        xs = torch.rand(M, device="cuda", generator=g) * 1e-2 + 1e-4
        ws = torch.rand(N, device="cuda", generator=g) * 1e-2 + 1e-4
        b = (torch.randn(N, device="cuda", generator=g) * 0.05).to(torch.bfloat16) if bias else None
        ya = pq.qlinear_s8(xq, xs, wq, ws, b, torch.bfloat16)
        fa = pq.qlinear_s8(xq, xs, wq, ws, b.float() if bias else None, torch.float32)
        pq_opt("PQ_SP256_PERSIST", "1")
        yb = pq.qlinear_s8(xq, xs, wq, ws, b, torch.bfloat16)
        fb = pq.qlinear_s8(xq, xs, wq, ws, b.float() if bias else None, torch.float32)
        ib = pq.int_mm(xq, wq)
        pq_opt("PQ_SP256_PERSIST", "0")
        ia = pq.int_mm(xq, wq)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int16), yb.view(torch.int16)), (M, N, K, "bf16")
        assert torch.equal(fa.view(torch.int32), fb.view(torch.int32)), (M, N, K, "f32")
        assert torch.equal(ia, ib), (M, N, K, "int32")


@pytest.mark.parametrize("shape", [(4096, 4096, 4096), (2048, 22016, 4096), (4096, 1024, 28672), (4096, 1280, 8192), (512, 4096, 4096), (64, 6144, 4096), (16, 28672, 4096)])
def test_whole_accumulator_equals_torch_int_mm_on_this_gpu(shape):
    """EVERY int32 accumulator of a full-size problem against an implementation that shares no code with this library: torch._int_mm on the same GPU
    (hipBLASLt's int8 GEMM).  Integer sums have one right answer; the sampled-row checks against the HOST torch._int_mm above pin the contract's named
    primitive, this pins the rest of the output to it.  Full-range operands (-128 .. 127)."""
    import protoquant_amd as pq
    M, N, K = shape
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    got = pq.int_mm(a, b)
    if M <= 16:                       # torch._int_mm wants more than 16 rows: pad the activation
        ap = torch.zeros((32, K), dtype=torch.int8, device="cuda"); ap[:M] = a
        want = torch._int_mm(ap, b.t())[:M]
    else:
        want = torch._int_mm(a, b.t())
    assert want.dtype == torch.int32 and torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(4096, 4096, 4096), (2048, 4096, 11008), (4096, 1024, 8192), (256, 4096, 4096), (48, 4096, 14336)])
def test_whole_output_equals_the_epilogue_in_stock_torch_ops(shape, dtype):
    """EVERY output element of full-size problems, all three output types, with bias: the fused epilogue against QSPEC E1-E4 written in stock torch ops on the GPU
    over torch._int_mm's accumulator — two f32 multiplies, one add, one RNE cast: nothing there that torch-ROCm rounds differently from the CPU (its division does;
    the epilogue has none).  The hashes of tests/golden/fullsize_hashes.json pin bf16 outputs to the CPU pipeline; this extends the whole-output check to fp16 / f32."""
    import protoquant_amd as pq
    M, N, K = shape
    g = torch.Generator(device="cuda"); g.manual_seed(M * 3 + N + K)
    a = (torch.randn((M, K), device="cuda", generator=g) * 30).round().clamp(-127, 127).to(torch.int8)
    b = (torch.randn((N, K), device="cuda", generator=g) * 30).round().clamp(-127, 127).to(torch.int8)
    xs = torch.rand(M, device="cuda", generator=g) * 0.02 + 1e-3
    ws = torch.rand(N, device="cuda", generator=g) * 0.002 + 1e-4
    bias = torch.randn(N, device="cuda", generator=g).to(dtype)
    y = pq.qlinear_s8(a, xs, b, ws, bias, dtype)
    ap = a
    if M <= 16:
        ap = torch.zeros((32, K), dtype=torch.int8, device="cuda"); ap[:M] = a
    acc = torch._int_mm(ap, b.t())[:M]
    ref = ((acc.float() * xs[:, None]) * ws[None, :] + bias.float()[None, :]).to(dtype)
    view = torch.int16 if dtype != torch.float32 else torch.int32
    assert torch.equal(y.view(view), ref.view(view))
