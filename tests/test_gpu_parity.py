"""GPU parity (run with -m gpu on MI355X): the HIP path, called through the C-ABI, vs the committed
golden vectors (torch._int_mm pipeline) and vs the oracle on seeded inputs.  Bar: bit-exact int8
codes / fp32 scales / int32 accumulators, bit-identical bf16/fp16/f32 outputs (=> 1e-5 rel)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import qspec_numpy as Q
from tests.gpu_util import TD, bits, same, same_f, to_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pq():
    import protoquant_amd
    from protoquant_amd import _lib
    _lib.lib()            # fails loudly if libpq_hip.so is missing
    assert torch.cuda.is_available()
    return protoquant_amd


def test_golden_quantize(pq, golden):
    g = golden
    x = to_gpu(g["x"], g["code"])
    q = pq.quantize(x, axis=-1)
    same(q.int_data, g["xq"], "xq"); same(q.scale, g["xs"], "xs")
    w = to_gpu(g["w"], g["code"])
    qw = pq.quantize(w, axis=-1)
    same(qw.int_data, g["wq"], "wq"); same(qw.scale, g["ws"], "ws")
    qc = pq.quantize(x, axis=0)
    same(qc.int_data, g["x_colq"], "x_colq"); same(qc.scale, g["x_cols"], "x_cols")
    same_f(pq.dequantize(q), g["x_deq"], g["code"], "x_deq")
    same_f(pq.dequantize(qc), g["x_coldeq"], g["code"], "x_coldeq")


@pytest.mark.parametrize("variant", ["auto", "generic", "sp256_16", "sp128_16", "sp128x128", "ring128", "ring64x128", "ring64x64", "ring128x160"])
def test_golden_gemm_and_qlinear(pq, golden, variant, pq_opt):
    g = golden
    pq_opt("PQ_FORCE_VARIANT", "" if variant == "auto" else variant)
    xq = torch.from_numpy(g["xq"]).cuda(); wq = torch.from_numpy(g["wq"]).cuda()
    xs = torch.from_numpy(g["xs"]).cuda(); ws = torch.from_numpy(g["ws"]).cuda()
    same(pq.int_mm(xq, wq), g["acc"], "acc")
    bias = to_gpu(g["bias"], g["code"]) if g["bias"] is not None else None
    same_f(pq.qlinear_s8(xq, xs, wq, ws, bias, TD[g["code"]]), g["y"], g["code"], "y")


def test_golden_qlinear_module(pq, golden):
    g = golden
    lin = torch.nn.Linear(int(g["K"]), int(g["N"]), bias=g["bias"] is not None, device="cuda", dtype=TD[g["code"]])
    with torch.no_grad():
        lin.weight.copy_(to_gpu(g["w"], g["code"]))
        if g["bias"] is not None:
            lin.bias.copy_(to_gpu(g["bias"], g["code"]))
    m = pq.qlinear.from_linear(lin)
    same(m.wq, g["wq"], "module wq"); same(m.ws, g["ws"], "module ws")
    x = to_gpu(g["x"], g["code"])
    same_f(m(x), g["y"], g["code"], "module y")
    # [..., K] inputs flatten to [M, K]
    if x.shape[0] % 2 == 0:
        y3 = m(x.reshape(2, -1, x.shape[1]))
        same_f(y3.reshape(-1, y3.shape[-1]), g["y"], g["code"], "module y 3-D")


SHAPES = [(1, 1, 1), (3, 5, 7), (64, 64, 64), (100, 200, 300), (255, 257, 128), (256, 256, 128),
          (512, 256, 384), (257, 300, 256), (1, 4096, 4096), (333, 1024, 1024), (1024, 768, 2048)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("variant", ["auto", "generic", "sp256_16", "sp128_16", "sp128x128", "ring128", "ring64x128", "ring64x64", "ring128x160"])
def test_int_gemm_exact_full_range(pq, M, N, K, variant, pq_opt):
    """Full-range int8 operands (incl. -128) and an asymmetric B: exact int32 vs int64 matmul."""
    pq_opt("PQ_FORCE_VARIANT", "" if variant == "auto" else variant)
    rng = np.random.default_rng(M * 1000003 + N * 1009 + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8)
    b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    want = (a.astype(np.int64) @ b.astype(np.int64).T).astype(np.int32)
    got = pq.int_mm(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    same(got, want, f"acc {M}x{N}x{K} {variant}")


@pytest.mark.parametrize("variant", ["sp256_16", "sp128_16", "sp128x128", "ring128", "ring64x128", "ring64x64", "ring128x160", "generic"])
def test_gemm_identity_asymmetric(pq, variant, pq_opt):
    """A = I with an asymmetric B catches a swapped C layout (cdna guide §3)."""
    pq_opt("PQ_FORCE_VARIANT", variant)
    n = 256
    a = np.eye(n, dtype=np.int8)
    b = (np.arange(n * n, dtype=np.int64).reshape(n, n) % 251 - 125).astype(np.int8)
    got = pq.int_mm(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    same(got, b.T.astype(np.int32), "identity")


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 13), (33, 1000), (64, 4096), (5, 11008), (3, 28672), (2, 40000),
                                        (129, 520), (0, 16), (4, 0)])
def test_quant_vs_oracle(pq, code, rows, cols):
    rng = np.random.default_rng(rows * 7919 + cols + code)
    xf = (rng.standard_normal((rows, cols)) * rng.uniform(0.01, 30)).astype(np.float32)
    if rows > 2 and cols > 2:
        xf[1] = 0
        xf[2, 0] = 1e30 if code != 1 else 60000
    x = Q.from_f32(xf, code)
    xg = to_gpu(x, code)
    q = pq.quantize(xg, axis=-1)
    wq, wsc = C.quant_rowwise(x, code)
    same(q.int_data, wq, "rowwise codes"); same(q.scale, wsc, "rowwise scale")
    for dt in (0, 1, 2):
        same(pq.dequantize(q, TD[dt]), C.dequant(wq, wsc, 1, dt), f"dequant->{dt}")
    if rows > 0 and cols > 0:
        qc = pq.quantize(xg, axis=0)
        cq, cs = C.quant_colwise(x, code)
        same(qc.int_data, cq, "colwise codes"); same(qc.scale, cs, "colwise scale")
        same(pq.dequantize(qc), C.dequant(cq, cs, 0, code), "col dequant")


@pytest.mark.parametrize("code", [0, 1, 2])
def test_quant_special_values(pq, code):
    """NaN / Inf / signalling NaN / subnormal policy (QSPEC v2 Q2, Q3, Q5: a NaN PROPAGATES into its row's / column's scale)."""
    rng = np.random.default_rng(5)
    xf = rng.standard_normal((9, 64)).astype(np.float32)
    xf[1, 3] = np.nan; xf[2, 5] = np.inf; xf[3, :] = 0; xf[4, 0] = -np.inf; xf[4, 1] = np.nan
    xf[5, :] = 1e-41
    x = Q.from_f32(xf, code)
    if code != 2:
        x[6, 2] = 0x7F81 if code == 0 else 0x7C01
    else:
        x.view(np.uint32)[6, 2] = 0x7F800001
    xg = to_gpu(x, code)
    q = pq.quantize(xg, axis=-1)
    wq, wsc = C.quant_rowwise(x, code)
    same(q.int_data, wq, "codes"); same(q.scale, wsc, "scale")
    qc = pq.quantize(xg, axis=0)
    cq, cs = C.quant_colwise(x, code)
    same(qc.int_data, cq, "col codes"); same(qc.scale, cs, "col scale")


def test_many_rows_dequant_and_colquant(pq):
    """More than 65 535 x 4 rows: row blocks ride on grid.x (a grid.y of that size would fail to launch)."""
    rows, cols = 300000, 16
    x = (torch.randn(rows, cols, device="cuda") * 3).to(torch.bfloat16)
    xb = bits(x)
    q = pq.quantize(x, axis=-1)
    wq, ws = C.quant_rowwise(xb, 0)
    same(q.int_data, wq, "rows>262k codes"); same(q.scale, ws, "rows>262k scale")
    same(pq.dequantize(q), C.dequant(wq, ws, 1, 0), "rows>262k dequant")
    qc = pq.quantize(x, axis=0)
    cq, cs = C.quant_colwise(xb, 0)
    same(qc.int_data, cq, "rows>262k col codes"); same(qc.scale, cs, "rows>262k col scale")


def test_quant_strided_and_unaligned(pq):
    """Leading dimension > cols and an odd element offset take the generic paths."""
    rng = np.random.default_rng(11)
    big = torch.from_numpy(rng.standard_normal((40, 1030)).astype(np.float32)).cuda().to(torch.bfloat16)
    for view in (big[:, :1024], big[:, 1:1025], big[:, 3:1003], big[5:, 8:520]):
        x = view
        xb = bits(x.contiguous())
        q = pq.quantize(x, axis=-1)
        wq, wsc = C.quant_rowwise(xb, 0)
        same(q.int_data, wq, "strided codes"); same(q.scale, wsc, "strided scale")
        qc = pq.quantize(x, axis=0)
        cq, cs = C.quant_colwise(xb, 0)
        same(qc.int_data, cq, "strided col codes"); same(qc.scale, cs, "strided col scale")


@pytest.mark.parametrize("M,N,K,code,bias", [(300, 520, 640, 0, True), (256, 512, 1024, 1, True), (77, 130, 384, 2, False),
                                              (512, 1024, 512, 0, False)])
@pytest.mark.parametrize("variant", ["auto", "generic", "sp256_16", "sp128_16", "sp128x128", "ring128", "ring64x128", "ring64x64", "ring128x160"])
def test_qlinear_vs_oracle(pq, M, N, K, code, bias, variant, pq_opt):
    pq_opt("PQ_FORCE_VARIANT", "" if variant == "auto" else variant)
    rng = np.random.default_rng(M + N + K + code)
    x = Q.from_f32(rng.standard_normal((M, K)).astype(np.float32), code)
    w = Q.from_f32((rng.standard_normal((N, K)) * 0.02).astype(np.float32), code)
    b = Q.from_f32((rng.standard_normal(N) * 0.01).astype(np.float32), code) if bias else None
    wq, ws = C.quant_rowwise(w, code)
    y_want, xq_want, xs_want, acc_want = Q.qlinear(x, code, wq, ws, b)
    lin = torch.nn.Linear(K, N, bias=bias, device="cuda", dtype=TD[code])
    with torch.no_grad():
        lin.weight.copy_(to_gpu(w, code))
        if bias:
            lin.bias.copy_(to_gpu(b, code))
    m = pq.qlinear.from_linear(lin)
    same(m(to_gpu(x, code)), y_want, "qlinear y")
    same(pq.int_mm(torch.from_numpy(xq_want).cuda(), torch.from_numpy(wq).cuda()), acc_want, "acc")


def test_qlinear_unaligned_scales_and_output(pq):
    """Scale vectors / outputs that are only 4-byte aligned take the direct epilogue: same bits."""
    rng = np.random.default_rng(21)
    M, N, K = 256, 512, 256
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M + 1).astype(np.float32); ws = rng.random(N + 1).astype(np.float32)
    want = Q.epilogue((a.astype(np.int64) @ b.astype(np.int64).T).astype(np.int32), xs[1:], ws[1:], None, 0)
    xs_t = torch.from_numpy(xs).cuda()[1:]; ws_t = torch.from_numpy(ws).cuda()[1:]          # +4 bytes: unaligned
    got = pq.qlinear_s8(torch.from_numpy(a).cuda(), xs_t, torch.from_numpy(b).cuda(), ws_t, None, torch.bfloat16)
    same(got, want, "unaligned scales")
    big = torch.empty((M, N + 8), dtype=torch.bfloat16, device="cuda")
    out = big[:, 1:N + 1]                                                                   # 2-byte-aligned rows, ld = N + 8
    pq.qlinear_s8(torch.from_numpy(a).cuda(), xs_t, torch.from_numpy(b).cuda(), ws_t, None, torch.bfloat16, out=out)
    same(out.contiguous(), want, "unaligned output")


@pytest.mark.parametrize("M,N,K,code,bias", [(300, 260, 8192, 0, True), (512, 1024, 8192, 0, False), (130, 517, 8192, 2, True),
                                              (1024, 1024, 8192, 1, True), (2048, 1024, 8192, 0, False), (1500, 1000, 16384, 0, True), (128, 1024, 16384, 0, True),
                                              (200, 2048, 14336, 1, False)])
def test_splitk_bit_identical(pq, M, N, K, code, bias, pq_opt):
    """Small M*N / long K: the workspace-based split-K path (exact integer slab reduction) == the oracle, and
    == the single-pass kernel (PQ_NO_SPLITK).  (Small grids: since round 4 the dispatcher prefers the single-pass 64-row ring tiles there; PQ_NO_MIDM=1 keeps
    the two-pass path covered at those shapes too.)"""
    from protoquant_amd import _lib
    if _lib.lib().pq_gemm_variant_name(M, N, K, K, K).startswith(b"ring64"):
        assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == 0
        pq_opt("PQ_NO_MIDM", "1")
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) > 0, "shape should be planned as split-K"
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if bias else None
    acc = (a.astype(np.float64) @ b.astype(np.float64).T).astype(np.int32)      # exact: |acc| < 2^53 (BLAS, fast)
    want = Q.epilogue(acc, xs, ws, bv, code)
    args = (torch.from_numpy(a).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws).cuda(),
            to_gpu(bv, code) if bias else None, TD[code])
    same(pq.qlinear_s8(*args), want, "split-K y")
    pq_opt("PQ_NO_SPLITK", "1")
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == 0
    same(pq.qlinear_s8(*args), want, "single-pass y")


@pytest.mark.parametrize("M,N,S,NT,code,bias", [(300, 520, 2, 5, 0, True), (257, 256, 3, 7, 2, False), (512, 300, 4, 6, 1, True), (256, 512, 2, 8, 0, False),
                                                 (640, 256, 2, 9, 0, True), (256, 777, 5, 10, 0, False), (130, 130, 2, 11, 2, True), (512, 512, 2, 4, 0, True)])
def test_forced_splitk_slices_on_the_asm_loop(pq, M, N, S, NT, code, bias, pq_opt):
    """PQ_FORCE_SPLITK: S slices of NT K-tiles each on the 256 x 256 split-ring tile — NT >= 5 runs the asm K-loop with int32 slab output (every ring
    phase at the exit: NT = 5 .. 11), NT = 4 the 2-deep HIP ring — + the reduction pass, ragged M and N: == the oracle and == the single-pass kernel."""
    from protoquant_amd import _lib
    K = S * NT * 128
    pq_opt("PQ_FORCE_SPLITK", str(S))
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == S * M * N * 4
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if bias else None
    acc = (a.astype(np.float64) @ b.astype(np.float64).T).astype(np.int32)
    want = Q.epilogue(acc, xs, ws, bv, code)
    args = (torch.from_numpy(a).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws).cuda(),
            to_gpu(bv, code) if bias else None, TD[code])
    same(pq.qlinear_s8(*args), want, "forced split-K y")
    pq_opt("PQ_FORCE_SPLITK", "")
    same(pq.qlinear_s8(*args), want, "default dispatch y")


@pytest.mark.parametrize("M,N,S,NT,code,bias", [(300, 520, 2, 5, 0, True), (257, 256, 3, 7, 2, False), (512, 300, 4, 6, 1, True), (256, 512, 2, 8, 0, False),
                                                 (640, 256, 2, 9, 0, True), (256, 777, 5, 10, 0, False), (130, 130, 2, 11, 2, True), (2048, 4096, 2, 43, 0, False),
                                                 (1024, 1024, 4, 16, 1, True), (2048, 4096, 2, 43, 0, True)])
def test_fused_splitk_matches(pq, M, N, S, NT, code, bias, pq_opt):
    """PQ_FSK=S (forced here; the plan picks S = 2 for half-filled grids with a long K): the K-slices of a 256 x 256 tile hand their partial sums over INSIDE the GEMM kernel — the ticket form (the last
    workgroup of a tile to arrive adds the others' slabs: the default, placement-independent) and, with PQ_FSK_SYMMETRIC=1, S = 2: the symmetric exchange between
    workgroups 2 p and 2 p + 1 (each finishes one column half), S = 4: the four-way symmetric exchange (a quarter each);
    every ring phase at the exit (NT = 5 .. 11), ragged M and N (edge tiles through the direct epilogue), a full-size half-filled grid (the cfg-3 `down`
    GEMM, all 256 CUs in the exchange at once), repeated calls on one workspace (the launcher re-zeroes the flags): == the oracle and == the default dispatch."""
    from protoquant_amd import _lib
    K = S * NT * 128
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if bias else None
    args = (torch.from_numpy(a).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws).cuda(),
            to_gpu(bv, code) if bias else None, TD[code])
    pq_opt("PQ_FSK", "0")
    y_def = pq.qlinear_s8(*args)
    if M * N <= 1 << 20:
        acc = (a.astype(np.float64) @ b.astype(np.float64).T).astype(np.int32)
        same(y_def, Q.epilogue(acc, xs, ws, bv, code), "default dispatch y")
    pq_opt("PQ_FSK", str(S))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == ((tiles * 4 * (4 if S == 4 else 2) + 255) // 256) * 256 + tiles * (S - 1) * 256 * 256 * 4
    for rep in range(6):
        pq_opt("PQ_FSK_SYMMETRIC", "1" if rep >= 3 else "")          # three launches in the ticket form, three in the symmetric form (S = 2 / 4; other S: ticket again)
        y = pq.qlinear_s8(*args)
        assert torch.equal(y.view(torch.uint8), y_def.view(torch.uint8)), f"fused split-K y (rep {rep})"
    # round 5: the symmetric kernels under a COOPERATIVE launch (PQ_FSK_COOP=1: co-residency guaranteed by the runtime, ticket form if it refuses) — eagerly and
    # captured into a hipGraph; measured slower than the ticket form (profiles/r05_ab_fsk_coop.txt), kept opt-in and bit-exact
    pq_opt("PQ_FSK_SYMMETRIC", "")
    pq_opt("PQ_FSK_COOP", "1")
    for rep in range(2):
        assert torch.equal(pq.qlinear_s8(*args).view(torch.uint8), y_def.view(torch.uint8)), f"cooperative fused split-K y (rep {rep})"
    out = torch.empty_like(y_def)
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        pq.qlinear_s8(*args, out=out)
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=s_):
            pq.qlinear_s8(*args, out=out)
        out.zero_(); g_.replay(); torch.cuda.synchronize()
    assert torch.equal(out.view(torch.uint8), y_def.view(torch.uint8)), "cooperative fused split-K under capture"


@pytest.mark.parametrize("M,N,S,NT", [(300, 300, 3, 17), (520, 260, 3, 64), (260, 300, 5, 27), (4096, 1280, 3, 64), (300, 520, 7, 36)])
def test_fused_splitk_with_uneven_slices(pq, pq_opt, M, N, S, NT):
    """round 6: the ticket form deals the K-tiles as evenly as they go — NT K-tiles over S slices, the first NT % S slices one more (3 slices of 64: 22 / 21 / 21; 5 of 27: 6 / 6 / 5 / 5 / 5,
    the minimum of five per slice) — where rounds 3-5 wanted K % (128 S) == 0.  Bit-identical to the single-pass dispatch and (small shapes) to the oracle, repeated on one workspace."""
    from protoquant_amd import _lib
    K = NT * 128
    assert K % (128 * S) != 0
    rng = np.random.default_rng(M + N + K + S)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), 0)
    args = (torch.from_numpy(a).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(ws).cuda(), to_gpu(bv, 0), torch.bfloat16)
    pq_opt("PQ_FSK", "0")
    y_def = pq.qlinear_s8(*args)
    if M * N <= 1 << 18:
        acc = (a.astype(np.float64) @ b.astype(np.float64).T).astype(np.int32)
        same(y_def, Q.epilogue(acc, xs, ws, bv, 0), "default dispatch y")
    pq_opt("PQ_FSK", str(S))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == ((tiles * 4 * (4 if S == 4 else 2) + 255) // 256) * 256 + tiles * (S - 1) * 256 * 256 * 4
    for rep in range(4):
        y = pq.qlinear_s8(*args)
        assert torch.equal(y.view(torch.uint8), y_def.view(torch.uint8)), f"uneven fused split-K y (rep {rep})"
    pq_opt("PQ_FSK_SYMMETRIC", "1")                      # the symmetric forms keep equal slices: an uneven K is refused by that plan (the planner's other choices run): same bits
    assert torch.equal(pq.qlinear_s8(*args).view(torch.uint8), y_def.view(torch.uint8))


@pytest.mark.parametrize("cus", [32, 64, 128, 256])
def test_plans_made_for_fewer_cus_stay_bit_exact(pq, pq_opt, cus):
    """PQ_FAKE_CUS (a partitioned / CU-masked device as the planner would see it, with its XCD count: one per 32 CUs in the tile remaps): every plan the smaller
    device gets — other tile kinds, tail splits at other places, fused split-K on other grids — gives the bits of the default plan and of torch._int_mm."""
    from protoquant_amd import _lib
    seen = set()
    for (M, N, K) in ((2048, 2816, 1024), (1024, 2048, 11264), (1000, 1100, 512), (512, 4096, 1024), (4096, 1280, 1024), (3000, 520, 640), (1024, 1024, 4096)):
        g = torch.Generator().manual_seed(M + N + K)
        a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
        b = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
        xs, ws = torch.rand(M, generator=g).cuda() * 0.1, torch.rand(N, generator=g).cuda() * 0.01
        ag, bg = a.cuda(), b.cuda()
        pq_opt("PQ_FAKE_CUS", "")
        y_def, acc_def = pq.qlinear_s8(ag, xs, bg, ws, None, torch.bfloat16), pq.int_mm(ag, bg)
        assert torch.equal(acc_def.cpu(), torch._int_mm(a, b.t()))
        pq_opt("PQ_FAKE_CUS", str(cus))
        seen.add((_lib.lib().pq_gemm_variant_name(M, N, K, K, K), _lib.lib().pq_qlinear_workspace_bytes(M, N, K) > 0))
        assert torch.equal(pq.qlinear_s8(ag, xs, bg, ws, None, torch.bfloat16).view(torch.int16), y_def.view(torch.int16)), (cus, M, N, K)
        assert torch.equal(pq.int_mm(ag, bg), acc_def), (cus, M, N, K)
        yt = pq.qlinear_s8_t(ag, xs, bg, ws, None, torch.bfloat16)
        assert torch.equal(yt.t().contiguous().view(torch.int16), y_def.view(torch.int16)), (cus, "transposed", M, N, K)
    assert len(seen) >= 3          # the shapes exercise several plans at every CU count


@pytest.mark.parametrize("M", [1, 2, 7, 16, 17, 32, 33, 48, 64])
@pytest.mark.parametrize("N,K", [(16, 128), (100, 256), (512, 1024), (4096, 4096), (1000, 2048), (37, 8192), (8192, 1024)])
def test_skinny_gemm_exact(pq, M, N, K, pq_opt):
    """Decode-like shapes (M <= 64) run the weight-streaming kernel: full-range int8 operands, exact int32 accumulators
    (int_mm) and the fused epilogue with bias, all dtypes of output, ragged N and M."""
    from protoquant_amd import _lib
    if M > 32 and N >= 6144 and K % 128 == 0:      # 33 .. 64 tokens against wide matrices: the 64-row ring tiles since round 4; the weight-streaming kernel stays covered by force
        assert not _lib.lib().pq_gemm_variant_name(M, N, K, K, K).startswith(b"skinny")
        pq_opt("PQ_FORCE_VARIANT", "skinny")
    assert _lib.lib().pq_gemm_variant_name(M, N, K, K, K) == b"skinny_16x16x64"
    rng = np.random.default_rng(M * 7919 + N + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    acc = (a.astype(np.float64) @ b.astype(np.float64).T).astype(np.int32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    same(pq.int_mm(ta, tb), acc, "skinny acc")
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    code = (M + N) % 3
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if (M % 2) else None
    want = Q.epilogue(acc, xs, ws, bv, code)
    got = pq.qlinear_s8(ta, torch.from_numpy(xs).cuda(), tb, torch.from_numpy(ws).cuda(), to_gpu(bv, code) if bv is not None else None, TD[code])
    same(got, want, "skinny y")


def test_skinny_strided_operands_and_qlinear_module(pq):
    """Row strides larger than K on both operands, an output view with ld > N, and the module path (K1 + skinny GEMM)."""
    rng = np.random.default_rng(31)
    M, N, K = 24, 777, 512
    abig = rng.integers(-128, 128, (M, K + 128), dtype=np.int8); bbig = rng.integers(-128, 128, (N, K + 256), dtype=np.int8)
    a, b = abig[:, :K], bbig[:, :K]
    acc = (a.astype(np.int64) @ b.astype(np.int64).T).astype(np.int32)
    same(pq.int_mm(torch.from_numpy(abig).cuda()[:, :K], torch.from_numpy(bbig).cuda()[:, :K]), acc, "strided skinny acc")
    xs = rng.random(M).astype(np.float32); ws = rng.random(N).astype(np.float32)
    big = torch.empty((M, N + 24), dtype=torch.bfloat16, device="cuda")
    out = big[:, 8:N + 8]
    pq.qlinear_s8(torch.from_numpy(abig).cuda()[:, :K], torch.from_numpy(xs).cuda(), torch.from_numpy(bbig).cuda()[:, :K],
                  torch.from_numpy(ws).cuda(), None, torch.bfloat16, out=out)
    same(out.contiguous(), Q.epilogue(acc, xs, ws, None, 0), "strided skinny y")
    torch.manual_seed(7)
    lin = torch.nn.Linear(1024, 640, bias=True, dtype=torch.bfloat16)
    x = torch.randn(5, 1024).to(torch.bfloat16)
    import copy
    y = pq.qlinear.from_linear(copy.deepcopy(lin).cuda())(x.cuda())
    wq, wsc = C.quant_rowwise(bits(lin.weight), 0)
    y_want, _, _, _ = Q.qlinear(bits(x), 0, wq, wsc, bits(lin.bias))
    same(y, y_want, "qlinear module on a decode batch")


@pytest.mark.parametrize("variant", ["auto", "generic", "sp256_16", "sp128_16", "sp128x128", "ring128", "skinny"])
def test_extreme_accumulation_near_int32_limit(pq, variant, pq_opt):
    """K = 130944 with every code at -128 on both sides: |acc| = 128^2 * K = 2 145 386 496, 2.1 M short of 2^31 — the
    accumulators must neither saturate nor wrap, in any variant (skinny: M = 16; the others: M = 192)."""
    pq_opt("PQ_FORCE_VARIANT", "" if variant == "auto" else variant)
    K, N = 130944, 320
    for M in ((16,) if variant == "skinny" else (192,)):
        a = torch.full((M, K), -128, dtype=torch.int8, device="cuda"); b = torch.full((N, K), -128, dtype=torch.int8, device="cuda")
        b[1::2] = 127                                   # alternate rows: large negative sums too
        acc = pq.int_mm(a, b).cpu().numpy()
        want = np.empty((M, N), np.int64); want[:, 0::2] = 128 * 128 * K; want[:, 1::2] = -128 * 127 * K
        assert np.array_equal(acc.astype(np.int64), want)


@pytest.mark.parametrize("M,N,K,code,bias", [(2048, 11008, 128, 0, True), (4096, 4352, 128, 1, False), (11008, 2048, 128, 0, True),
                                              (2050, 10990, 256, 2, True)])
def test_tail_split_bit_identical(pq, M, N, K, code, bias, pq_opt):
    """Grids a little over a whole number of rounds: the trailing tile columns/rows run as a second launch of 128-row
    tiles.  Result == the oracle, == the single launch (PQ_NO_TAILSPLIT), and the int32 twin stays exact."""
    from protoquant_amd import _lib
    assert b"tail" in _lib.lib().pq_gemm_variant_name(M, N, K, K, K), "shape should be planned with a tail launch"
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if bias else None
    acc = (a.astype(np.int32) @ b.astype(np.int32).T)
    want = Q.epilogue(acc, xs, ws, bv, code)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    args = (ta, torch.from_numpy(xs).cuda(), tb, torch.from_numpy(ws).cuda(), to_gpu(bv, code) if bias else None, TD[code])
    same(pq.qlinear_s8(*args), want, "tail-split y")
    same(pq.int_mm(ta, tb), acc, "tail-split acc")
    pq_opt("PQ_NO_TAILSPLIT", "1")
    assert b"tail" not in _lib.lib().pq_gemm_variant_name(M, N, K, K, K)
    same(pq.qlinear_s8(*args), want, "single-launch y")


@pytest.mark.parametrize("M,N,K,code,bias", [(512, 512, 128, 0, True), (512, 768, 256, 1, False), (300, 520, 384, 0, True), (1000, 777, 512, 2, True),
                                              (768, 1024, 640, 0, False), (1024, 512, 4096, 0, True), (257, 255, 1152, 1, True)])
def test_split_rings_bit_identical(pq, M, N, K, code, bias, pq_opt):
    """The 256 x 256 tile with split LDS rings (weights three slots deep, activations two: the default) against the same tile with the
    2-deep ring of whole K-tiles (PQ_SP256_P3=0), rings of 1 to 32 K-tiles, ragged edges, every output dtype: both == the oracle."""
    rng = np.random.default_rng(M * 3 + N * 5 + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if bias else None
    acc = (a.astype(np.int32) @ b.astype(np.int32).T)
    want = Q.epilogue(acc, xs, ws, bv, code)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    args = (ta, torch.from_numpy(xs).cuda(), tb, torch.from_numpy(ws).cuda(), to_gpu(bv, code) if bias else None, TD[code])
    pq_opt("PQ_FORCE_VARIANT", "sp256_16")
    same(pq.qlinear_s8(*args), want, "split rings y")
    same(pq.int_mm(ta, tb), acc, "split rings acc")
    same(pq.qlinear_s8_t(*args).t().contiguous(), want, "split rings y^T")
    pq_opt("PQ_SP256_P3", "0")
    same(pq.qlinear_s8(*args), want, "2-deep ring y")
    same(pq.int_mm(ta, tb), acc, "2-deep ring acc")


def test_splitk_workspace_too_small_is_an_error(pq, pq_opt):
    from protoquant_amd import _lib
    L = _lib.lib()
    M, N, K = 1024, 1024, 8192          # (a quarter-filled grid with a long K: the round-3 plan is split-K; the default since round 4 is the single-pass 64 x 64 ring tile)
    pq_opt("PQ_NO_MIDM", "1")
    need = L.pq_qlinear_workspace_bytes(M, N, K)
    assert need > 0
    a = torch.zeros((M, K), dtype=torch.int8, device="cuda"); b = torch.zeros((N, K), dtype=torch.int8, device="cuda")
    s1 = torch.ones(M, device="cuda"); s2 = torch.ones(N, device="cuda"); y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    wsp = torch.empty(need // 2, dtype=torch.uint8, device="cuda")
    st = L.pq_qlinear_s8(a.data_ptr(), K, s1.data_ptr(), b.data_ptr(), K, s2.data_ptr(), None, y.data_ptr(), N, 0, M, N, K,
                         wsp.data_ptr(), wsp.numel(), None)
    assert st == 5 and b"workspace" in L.pq_last_error()
    # NULL workspace: allowed, single-pass path
    st = L.pq_qlinear_s8(a.data_ptr(), K, s1.data_ptr(), b.data_ptr(), K, s2.data_ptr(), None, y.data_ptr(), N, 0, M, N, K, None, 0, None)
    torch.cuda.synchronize()
    assert st == 0 and float(y.float().abs().max()) == 0.0


def test_full_size_cfg2_properties(pq):
    """BASELINE config 2 (M=N=K=4096, bf16): size-independent checks + sampled exact parity.
    (a) xq/xs bit-exact vs the C oracle on all rows; (b) int32 accumulator exact vs int64 matmul on
    256 sampled rows x all columns; (c) y bit-identical to the oracle epilogue on those rows;
    (d) linearity of the integer GEMM: acc(a, b1) + acc(a, b2) == acc(a, b1 + b2) for small codes."""
    M = N = K = 4096
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    xb, wb = bits(x), bits(w)
    lin = torch.nn.Linear(K, N, bias=False, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w.cuda())
    m = pq.qlinear.from_linear(lin)
    wq, ws = C.quant_rowwise(wb, 0)
    same(m.wq, wq, "wq 4096"); same(m.ws, ws, "ws 4096")
    xq_t = pq.quantize(x.cuda())
    xq, xs = C.quant_rowwise(xb, 0)
    same(xq_t.int_data, xq, "xq 4096"); same(xq_t.scale, xs, "xs 4096")
    acc = pq.int_mm(xq_t.int_data, m.wq)
    y = m(x.cuda())
    rows = np.random.default_rng(0).choice(M, 256, replace=False)
    acc_want = (xq[rows].astype(np.int64) @ wq.astype(np.int64).T).astype(np.int32)
    same(acc[torch.from_numpy(rows).cuda()], acc_want, "acc rows 4096")
    y_want = Q.epilogue(acc_want, xs[rows], ws, None, 0)
    same(y[torch.from_numpy(rows).cuda()], y_want, "y rows 4096")
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda")
    b1 = torch.randint(-60, 60, (N, K), dtype=torch.int8, device="cuda")
    b2 = torch.randint(-60, 60, (N, K), dtype=torch.int8, device="cuda")
    lhs = pq.int_mm(a, b1) + pq.int_mm(a, b2)
    rhs = pq.int_mm(a, (b1 + b2))
    assert torch.equal(lhs, rhs)
    # checksum of checksums: column sums of acc == (sum_m a[m,:]) . b^T computed in int64 on the host
    colsum = acc.sum(dim=0, dtype=torch.int64).cpu().numpy()
    want = xq.astype(np.int64).sum(axis=0) @ wq.astype(np.int64).T
    assert np.array_equal(colsum, want)


def test_fast_quotient_bruteforce(pq):
    """The division-free quotient of K1 (two FMA residual corrections on r = RN(1/s)) equals the IEEE quotient,
    hence the same int8 code, on 2^28 random (x, s) bit patterns: uniform bits, bf16-like x (the tie-heavy
    case), and scales near powers of two."""
    from protoquant_amd import _lib
    L = _lib.lib()
    n = 1 << 26
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(99)
    for mode in range(4):
        xb = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
        sb = torch.randint(0, (1 << 31) - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
        if mode == 1:
            xb = xb & ~0xFFFF                                # bf16-representable x: exact ties are frequent
        if mode == 2:
            sb = (sb & 0x7F800000) | (sb & 0x7)              # scales a few ulps above a power of two
        if mode == 3:
            sb = (sb & 0x7F800000) | 0x7FFFF8 | (sb & 0x7)   # scales just below a power of two (incl. all-ones)
            xb = xb & ~0xFFFF
        _lib.check(L.pq_selftest_fast_quotient(xb.data_ptr(), sb.data_ptr(), n, out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "selftest")
    torch.cuda.synchronize()
    assert out.tolist() == [0, 0], f"fast quotient mismatches (codes, quotients): {out.tolist()}"


@pytest.mark.parametrize("dtype_code,name,min_pairs", [(0, "bf16", 5.0e8), (1, "fp16", 1.0e9)])
def test_half_encode_whole_domain(pq, dtype_code, name, min_pairs):
    """16-bit rows take ONE residual correction (quant_device.h: quotient_fast1).  The GPU enumerates that path's whole domain with its
    own fma — every amax pattern whose scale is on the fast path x every magnitude <= amax x both signs — against true division + rintf
    (the host repeats the enumeration in C: tests/test_half_quotient_identity.py)."""
    from protoquant_amd import _lib
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().pq_selftest_half_encode(dtype_code, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "selftest")
    torch.cuda.synchronize()
    pairs, bad = out.tolist()
    assert pairs > min_pairs and bad == 0, f"{name}: {pairs} pairs, {bad} mismatches"


@pytest.mark.parametrize("dtype_code,name,min_patterns", [(0, "bf16", 30000), (1, "fp16", 40000)])
def test_silu_short_division_whole_domain(pq, dtype_code, name, min_patterns):
    """bf16 / fp16 rows of silu(g)*u take ONE residual correction on the raw reciprocal (producer_kernels.hip: SHORT).  The stored silu(g)
    is a function of the 16-bit g alone: every pattern of the fast-division domain (0 < |g| <= 86) is enumerated on the GPU through the
    short form, the two-correction form and true division — all three must store the same value."""
    from protoquant_amd import _lib
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().pq_selftest_silu_short(dtype_code, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "selftest")
    torch.cuda.synchronize()
    n, bad_short, bad_two = out.tolist()
    assert n > min_patterns and bad_short == 0 and bad_two == 0, f"{name}: {n} patterns, {bad_short} / {bad_two} mismatches"


def test_column_sharded_world1_matches_unsharded(pq):
    """ColumnShardedQLinear over RCCL (backend nccl) with a 1-rank group == plain qlinear, bit for bit."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        torch.manual_seed(3)
        lin = torch.nn.Linear(512, 768, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(6, 50, 512, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        y1 = pq.ColumnShardedQLinear.from_linear(lin)(x)
        assert y1.shape == y0.shape and torch.equal(y0.view(torch.int16), y1.view(torch.int16))
        for chunks in (2, 3, 7):        # row blocks, each block's all-gather issued asynchronously behind its GEMM
            y2 = pq.ColumnShardedQLinear.from_linear(lin, overlap_chunks=chunks)(x)
            torch.cuda.synchronize()
            assert y2.shape == y0.shape and torch.equal(y0.view(torch.int16), y2.view(torch.int16)), chunks
    finally:
        if created:
            dist.destroy_process_group()


def test_native_rccl_gather(pq):
    """libpq_rccl.so: the layout-fix kernel on a synthetic 3-rank stacked buffer, and a real 1-rank RCCL
    communicator (bootstrap, ncclAllGather, unstack) behind ColumnShardedQLinear — bit-identical to plain qlinear."""
    import ctypes
    import torch.distributed as dist
    from protoquant_amd import _rccl
    R = _rccl.lib()
    for dt, code in ((torch.bfloat16, 0), (torch.float32, 2)):
        for (G, M, n) in ((3, 37, 24), (2, 5, 7), (8, 64, 512)):
            st = torch.randn(G, M, n, device="cuda").to(dt)
            out = torch.empty((M, G * n), dtype=dt, device="cuda")
            _rccl.check(R.pq_unstack_cols(st.data_ptr(), out.data_ptr(), G, M, n, code, torch.cuda.current_stream().cuda_stream), "unstack")
            assert torch.equal(out, st.permute(1, 0, 2).reshape(M, G * n))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        gather = pq.RcclColumnGather()
        torch.manual_seed(4)
        lin = torch.nn.Linear(256, 384, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(70, 256, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        y1 = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather)(x)
        torch.cuda.synchronize()
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
        gather.close()
    finally:
        if created:
            dist.destroy_process_group()


def _oracle_row_sharded(x_bits, lin, world, code=0):
    """QSPEC restatement of the row-sharded linear: full-row weight scales, per-slice activation scales, f32 partials
    (bias on rank 0) summed in rank order, one cast."""
    from protoquant_amd.sharded import shard_bounds
    wq, ws = C.quant_rowwise(bits(lin.weight), code)
    total = None
    for r in range(world):
        k0, k1 = shard_bounds(lin.in_features, world, r)
        xq, xs = C.quant_rowwise(np.ascontiguousarray(x_bits[:, k0:k1]), code)
        b = lin.bias.detach().float().cpu().numpy() if (lin.bias is not None and r == 0) else None
        p = C.qlinear_s8(xq, xs, np.ascontiguousarray(wq[:, k0:k1]), ws, b, 2)
        total = p if total is None else (total + p).astype(np.float32)
    return Q.from_f32(total, code)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_row_sharded_partials_match_oracle(pq, world):
    """Every rank's shard built offline (shard_of), partial() on one GPU, f32 sum in rank order, one cast == the oracle
    restatement, bit for bit; world 1 == the plain qlinear."""
    torch.manual_seed(11)
    lin = torch.nn.Linear(640, 384, bias=True, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.mul_(3)
    x = torch.randn(70, 640).to(torch.bfloat16)
    from protoquant_amd.sharded import shard_bounds
    import copy
    lin_g = copy.deepcopy(lin).cuda()
    total = None
    for r in range(world):
        k0, k1 = shard_bounds(640, world, r)
        layer = pq.RowShardedQLinear.from_linear(lin_g, world=world, rank=r)
        p = layer.partial(x[:, k0:k1].contiguous().cuda())
        assert p.dtype == torch.float32 and p.shape == (70, 384)
        total = p if total is None else total + p
    y = total.to(torch.bfloat16)
    same(y, _oracle_row_sharded(bits(x), lin, world), f"row-sharded world {world}")
    if world == 1:
        same(pq.qlinear.from_linear(lin_g)(x.cuda()), bits(y), "world 1 == qlinear")
    ref = lin.float()(x.float())
    assert float((y.float().cpu() - ref).norm() / ref.norm()) < 0.02


def test_sharded_gated_mlp_partials(pq):
    """ShardedGatedMLP shards built offline for 2 ranks: sum of the ranks' down partials == the oracle pipeline (per-rank
    gate/up shard GEMMs, local silu*mul quantisation, K-split down), and close to the float MLP."""
    M, H, I, world = 64, 256, 512, 2
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(M, H, generator=gen).to(torch.bfloat16)
    lins = {n: torch.nn.Linear(i, o, bias=False, dtype=torch.bfloat16) for n, (o, i) in (("gate", (I, H)), ("up", (I, H)), ("down", (H, I)))}
    with torch.no_grad():
        for l in lins.values():
            l.weight.copy_((torch.randn(l.weight.shape, generator=gen) * 0.05).to(torch.bfloat16))
    import copy
    from protoquant_amd.sharded import shard_bounds
    g = {n: copy.deepcopy(l).cuda() for n, l in lins.items()}
    xq, xs = C.quant_rowwise(bits(x), 0)
    dq, ds = C.quant_rowwise(bits(lins["down"].weight), 0)
    total, want = None, None
    for r in range(world):
        mlp = pq.ShardedGatedMLP.from_linears(g["gate"], g["up"], g["down"], world=world, rank=r)
        gt, up = mlp.gate_up(x.cuda())
        p = mlp.down.partial(pq.silu_mul_quantize(gt, up))
        total = p if total is None else total + p
        lo, hi = shard_bounds(I, world, r)
        wg = C.quant_rowwise(bits(lins["gate"].weight[lo:hi]), 0); wu = C.quant_rowwise(bits(lins["up"].weight[lo:hi]), 0)
        hq, hs, _ = C.silu_mul_quant_rowwise(C.qlinear_s8(xq, xs, *wg, None, 0), C.qlinear_s8(xq, xs, *wu, None, 0), 0)
        pw = C.qlinear_s8(hq, hs, np.ascontiguousarray(dq[:, lo:hi]), ds, None, 2)
        want = pw if want is None else (want + pw).astype(np.float32)
    same(total.to(torch.bfloat16), Q.from_f32(want, 0), "sharded gated MLP")
    ref = lins["down"].float()(torch.nn.functional.silu(lins["gate"].float()(x.float())) * lins["up"].float()(x.float()))
    assert float((total.cpu() - ref).norm() / ref.norm()) < 0.03


def test_native_reduce_scatter_world1(pq):
    """libpq_rccl.so pq_reduce_scatter_rows on a real 1-rank communicator: f32 partials -> bf16 / f32 rows, and
    RowShardedQLinear(native=...) == plain qlinear."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29535")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        gather = pq.RcclColumnGather()
        rs = pq.RcclRowReduceScatter(gather)
        p = torch.randn(48, 200, device="cuda")
        assert torch.equal(rs(p, torch.bfloat16), p.to(torch.bfloat16)) and torch.equal(rs(p, torch.float32), p)
        assert torch.equal(rs(p, torch.float16), p.to(torch.float16))
        torch.manual_seed(5)
        lin = torch.nn.Linear(256, 384, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(64, 256, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        y1 = pq.RowShardedQLinear.from_linear(lin, native=rs)(x)
        y2 = pq.RowShardedQLinear.from_linear(lin, scatter=False)(x)
        torch.cuda.synchronize()
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)) and torch.equal(y0.view(torch.int16), y2.view(torch.int16))
        gather.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_swap_linears_on_mlp(pq):
    """Llama-style MLP block (gate/up/down) with every nn.Linear swapped: each projection bit-exact vs the oracle."""
    torch.manual_seed(5)
    H, I, M = 256, 640, 96

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gate, self.up, self.down = (torch.nn.Linear(H, I, bias=False), torch.nn.Linear(H, I, bias=False),
                                             torch.nn.Linear(I, H, bias=False))

        def forward(self, x):
            return self.down(torch.nn.functional.silu(self.gate(x)) * self.up(x))

    mlp = MLP().to("cuda", torch.bfloat16)
    ws = {n: bits(getattr(mlp, n).weight.detach()) for n in ("gate", "up", "down")}
    pq.swap_linears(mlp)
    assert all(isinstance(getattr(mlp, n), pq.qlinear) for n in ("gate", "up", "down"))
    x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
    g = mlp.gate(x); u = mlp.up(x)
    for name, out in (("gate", g), ("up", u)):
        wq, wsc = C.quant_rowwise(ws[name], 0)
        want, *_ = Q.qlinear(bits(x), 0, wq, wsc, None)
        same(out, want, name)
    h = torch.nn.functional.silu(g) * u
    wq, wsc = C.quant_rowwise(ws["down"], 0)
    want, *_ = Q.qlinear(bits(h), 0, wq, wsc, None)
    same(mlp.down(h), want, "down")
    assert mlp(x).shape == (M, H)


def test_fused_qkv_matches_separate(pq):
    """Horizontal fusion (q/k/v share the activation): outputs bit-identical to three separate qlinears."""
    torch.manual_seed(8)
    H = 512
    lq, lk, lv = (torch.nn.Linear(H, n, bias=True, device="cuda", dtype=torch.bfloat16) for n in (512, 128, 128))
    x = torch.randn(3, 100, H, device="cuda", dtype=torch.bfloat16)
    fused = pq.FusedQLinear.from_linears(lq, lk, lv)
    outs = fused(x)
    for lin, got in zip((lq, lk, lv), outs):
        want = pq.qlinear.from_linear(lin)(x)
        assert got.shape == want.shape and torch.equal(got.contiguous().view(torch.int16), want.view(torch.int16))


def test_full_size_cfg3_mlp_block(pq):
    """BASELINE config 3: Llama MLP block 4096 -> 11008 -> 4096 at seq 2048, all three projections as qlinear.
    Every projection is checked on 64 sampled token rows x all output channels, bit-exact vs the oracle
    (int64 accumulators + QSPEC epilogue), plus full xq/xs parity of both activation quantisations."""
    M, H, I = 2048, 4096, 11008
    g = torch.Generator().manual_seed(77)
    x = torch.randn(M, H, generator=g).to(torch.bfloat16)
    ws = {n: (torch.randn(o, i, generator=g) * 0.02).to(torch.bfloat16) for n, (o, i) in
          (("gate", (I, H)), ("up", (I, H)), ("down", (H, I)))}
    mods = {}
    for n, w in ws.items():
        lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device="cuda", dtype=torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_(w.cuda())
        mods[n] = pq.qlinear.from_linear(lin)
    fused = pq.FusedQLinear([mods["gate"], mods["up"]])
    xg = x.cuda()
    gate, up = fused(xg)
    h = (torch.nn.functional.silu(gate.float()) * up.float()).to(torch.bfloat16)      # stock torch-ROCm elementwise
    y = mods["down"](h)
    rows = np.random.default_rng(1).choice(M, 64, replace=False)
    rt = torch.from_numpy(rows).cuda()

    def check(name, inp_bits, out, out_name):
        wq, wsc = C.quant_rowwise(bits(ws[name]), 0)
        same(mods[name].wq, wq, name + " wq")
        xq, xs = C.quant_rowwise(inp_bits, 0)
        acc = (xq[rows].astype(np.int64) @ wq.astype(np.int64).T).astype(np.int32)
        same(out[rt].contiguous(), Q.epilogue(acc, xs[rows], wsc, None, 0), out_name)
        return xq, xs

    xq, xs = check("gate", bits(x), gate, "gate rows")
    check("up", bits(x), up, "up rows")
    q = pq.quantize(xg)
    same(q.int_data, xq, "xq cfg3"); same(q.scale, xs, "xs cfg3")
    hq, hs = check("down", bits(h), y, "down rows")
    qh = pq.quantize(h)
    same(qh.int_data, hq, "hq cfg3"); same(qh.scale, hs, "hs cfg3")


# ---------------------------------------------------------------- producer-fused quantisation (QSPEC S1-S6)
def _same_h(got, want, code, what):
    """h parity: NaNs as a class (payloads are not specified), everything else bit for bit"""
    gb = bits(got); wb = np.asarray(want)
    wb = wb.view(np.uint32) if wb.dtype == np.float32 else wb
    nan = np.isnan(Q.to_f32(want, code))
    assert np.array_equal(np.isnan(got.float().cpu().numpy()), nan), what + ": NaN positions"
    bad = int(np.count_nonzero(gb[~nan] != wb[~nan]))
    assert bad == 0, f"{what}: {bad} elements differ"


def test_golden_silu_mul_quantize(pq, producer_golden):
    g = producer_golden
    code = g["code"]
    qt, h = pq.silu_mul_quantize(to_gpu(g["g"], code), to_gpu(g["u"], code), return_h=True)
    same(qt.int_data, g["q"], "silu q"); same(qt.scale, g["scale"], "silu scale")
    _same_h(h, g["h"], code, "silu h")
    qt2 = pq.silu_mul_quantize(to_gpu(g["g"], code), to_gpu(g["u"], code))
    same(qt2.int_data, g["q"], "silu q (no h)"); same(qt2.scale, g["scale"], "silu scale (no h)")


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 13), (33, 1000), (64, 4096), (5, 11008), (3, 28672), (2, 40000), (300, 512),
                                       (1, 8), (130, 2048), (2, 65536 + 8)])
def test_silu_mul_quant_vs_oracle(pq, code, rows, cols):
    """Vector path (16-byte aligned widths up to 4096 vectors) and generic path (ragged / very wide), g and u as the two
    column halves of ONE [rows, 2*cols] matrix (how a fused gate+up GEMM hands them over) and as separate tensors."""
    rng = np.random.default_rng(rows * 131 + cols + code)
    gu = Q.from_f32((rng.standard_normal((rows, 2 * cols)) * 2.5).astype(np.float32), code)
    gu_t = to_gpu(gu, code)
    want_q, want_s, want_h = C.silu_mul_quant_rowwise(gu[:, :cols], gu[:, cols:], code)
    qt, h = pq.silu_mul_quantize(gu_t[:, :cols], gu_t[:, cols:], return_h=True)
    same(qt.int_data, want_q, "q (halves)"); same(qt.scale, want_s, "scale (halves)"); _same_h(h, want_h, code, "h (halves)")
    qt = pq.silu_mul_quantize(gu_t[:, :cols].contiguous(), gu_t[:, cols:].contiguous())
    same(qt.int_data, want_q, "q (separate)"); same(qt.scale, want_s, "scale (separate)")
    # == the two-step product path on the same h: quantize(h)
    q2 = pq.quantize(h)
    same(q2.int_data, want_q, "quantize(h)"); same(q2.scale, want_s, "quantize(h) scale")


@pytest.mark.parametrize("code", [0, 1, 2])
def test_silu_fast_division_equals_ieee_division(pq, code):
    """The vector path divides without v_div_scale/fixup when every |g| of a wave is in (0, 86]; the generic path (taken
    here by shifting the same data one element, which breaks the 16-byte alignment) always uses `/`.  Same h, bit for bit,
    on 8M random gate values spread over the whole safe range, incl. its edges and sub-2^-25 magnitudes."""
    rng = np.random.default_rng(77 + code)
    rows, cols = 2048, 4096
    mag = np.exp(rng.uniform(np.log(1e-30), np.log(86.0), (rows, cols)))
    mag[:, :16] = np.array([86.0, 85.99, 1e-38, 2.0**-25, 2.0**-24, 1.0, 17.3, 60.0, 6e-8 if code == 1 else 1e-30, 0.5, 3.0, 10.0, 44.0, 80.0, 87.9, 2.0**-100])
    g = (mag * rng.choice([-1.0, 1.0], (rows, cols))).astype(np.float32)
    u = rng.standard_normal((rows, cols)).astype(np.float32)
    gs, us = to_gpu(Q.from_f32(g, code), code), to_gpu(Q.from_f32(u, code), code)
    pad = torch.zeros((rows, cols + 8), dtype=gs.dtype, device="cuda"); pad2 = torch.zeros_like(pad)
    pad[:, 1:cols + 1] = gs; pad2[:, 1:cols + 1] = us
    qa, ha = pq.silu_mul_quantize(gs, us, return_h=True)                                    # vector path
    qb, hb = pq.silu_mul_quantize(pad[:, 1:cols + 1], pad2[:, 1:cols + 1], return_h=True)  # generic path
    assert torch.equal(ha.view(torch.int32 if code == 2 else torch.int16), hb.view(torch.int32 if code == 2 else torch.int16))
    assert torch.equal(qa.int_data, qb.int_data) and torch.equal(qa.scale, qb.scale)
    # and a sample of rows against the oracle
    want_q, want_s, want_h = C.silu_mul_quant_rowwise(bits(gs[:64]), bits(us[:64]), code)
    same(qa.int_data[:64], want_q, "q"); same(qa.scale[:64], want_s, "scale"); _same_h(ha[:64], want_h, code, "h")


def test_silu_mul_quantize_3d_and_errors(pq):
    g = torch.randn(2, 5, 64, device="cuda", dtype=torch.bfloat16); u = torch.randn(2, 5, 64, device="cuda", dtype=torch.bfloat16)
    qt = pq.silu_mul_quantize(g, u)
    assert qt.int_data.shape == (2, 5, 64) and qt.scale.shape == (10,) and qt.axis == 1
    want_q, want_s, _ = C.silu_mul_quant_rowwise(bits(g).reshape(10, 64), bits(u).reshape(10, 64), 0)
    same(qt.int_data.reshape(10, 64), want_q, "3d q"); same(qt.scale, want_s, "3d scale")
    with pytest.raises(ValueError):
        pq.silu_mul_quantize(g, u[:, :, :32])
    with pytest.raises(Exception):
        pq.silu_mul_quantize(g.cpu(), u.cpu())
    from protoquant_amd import _lib
    L = _lib.lib()
    st = L.pq_silu_mul_quant_rowwise(g.data_ptr(), 32, u.data_ptr(), 64, 0, 10, 64, qt.int_data.data_ptr(), 64, qt.scale.data_ptr(), None, 0, None)
    assert st == 1 and b"pq_silu_mul_quant_rowwise" in L.pq_last_error()


def test_gated_mlp_matches_oracle_pipeline(pq):
    """GatedMLP (fused gate+up GEMM -> silu_mul_quantize -> down GEMM) vs the oracle pipeline, bit for bit, and within
    int8 noise of the float block."""
    M, H, I = 192, 256, 640
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(M, H, generator=gen).to(torch.bfloat16)
    lins = {n: torch.nn.Linear(i, o, bias=False, dtype=torch.bfloat16) for n, (o, i) in (("gate", (I, H)), ("up", (I, H)), ("down", (H, I)))}
    with torch.no_grad():
        for l in lins.values():
            l.weight.copy_((torch.randn(l.weight.shape, generator=gen) * 0.05).to(torch.bfloat16))
    import copy
    mlp = pq.GatedMLP.from_linears(*(copy.deepcopy(lins[n]).cuda() for n in ("gate", "up", "down")))
    y = mlp(x.cuda())
    wq = {n: C.quant_rowwise(bits(l.weight), 0) for n, l in lins.items()}
    xq, xs = C.quant_rowwise(bits(x), 0)
    gate = C.qlinear_s8(xq, xs, *wq["gate"], None, 0); up = C.qlinear_s8(xq, xs, *wq["up"], None, 0)
    hq, hs, _ = C.silu_mul_quant_rowwise(gate, up, 0)
    same(y, C.qlinear_s8(hq, hs, *wq["down"], None, 0), "GatedMLP y")
    ref = lins["down"].float()(torch.nn.functional.silu(lins["gate"].float()(x.float())) * lins["up"].float()(x.float()))
    rel = float((y.float().cpu() - ref).norm() / ref.norm())
    assert rel < 0.03, rel


# ---------------------------------------------------------------- RMSNorm -> quantisation (QSPEC N1-N6)
def test_golden_rmsnorm_quantize(pq, rms_golden):
    g = rms_golden
    code, eps = g["code"], float(g["eps"])
    qt, h = pq.rmsnorm_quantize(to_gpu(g["x"], code), to_gpu(g["w"], code), eps, return_h=True)
    same(qt.int_data, g["q"], "rms q"); same(qt.scale, g["scale"], "rms scale")
    _same_h(h, g["h"], code, "rms h")
    qt2 = pq.rmsnorm_quantize(to_gpu(g["x"], code), to_gpu(g["w"], code), eps)
    same(qt2.int_data, g["q"], "rms q (no h)"); same(qt2.scale, g["scale"], "rms scale (no h)")


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 13), (33, 1000), (64, 4096), (5, 11008), (3, 28672), (2, 40000), (300, 512),
                                       (1, 8), (130, 2048), (4, 0), (17, 8192), (2, 65536 + 8)])
def test_rmsnorm_quant_vs_oracle(pq, code, rows, cols):
    """Vector path (up to 4096 16-byte vectors per row) and generic path (ragged / wider), rows with very different
    magnitudes, contiguous and strided x."""
    rng = np.random.default_rng(rows * 977 + cols + code)
    x = Q.from_f32((rng.standard_normal((rows, cols)) * rng.uniform(0.01, 50.0, (rows, 1))).astype(np.float32), code)
    w = Q.from_f32((1.0 + 0.3 * rng.standard_normal(cols)).astype(np.float32), code)
    want_q, want_s, want_h, _ = C.rmsnorm_quant_rowwise(x, w, 1e-5, code)
    qt, h = pq.rmsnorm_quantize(to_gpu(x, code), to_gpu(w, code), 1e-5, return_h=True)
    same(qt.int_data, want_q, "q"); same(qt.scale, want_s, "scale"); _same_h(h, want_h, code, "h")
    if cols:
        wide = torch.zeros((rows, cols + 16), dtype=TD[code], device="cuda")
        wide[:, :cols] = to_gpu(x, code)
        qt = pq.rmsnorm_quantize(wide[:, :cols], to_gpu(w, code), 1e-5)
        same(qt.int_data, want_q, "q (strided)"); same(qt.scale, want_s, "scale (strided)")


@pytest.mark.parametrize("wave_max", ["512", "0"])
@pytest.mark.parametrize("code,cols", [(0, 4096), (0, 3072), (1, 2304), (2, 2048), (2, 1100 * 4 // 4 * 4)])
def test_rmsnorm_both_layouts_same_bits(pq, pq_opt, wave_max, code, cols):
    """The wave-per-row layout (forced up to 512 vectors per row: round 1's choice) and the 256-thread-block layout (forced everywhere)
    both reproduce the oracle: QSPEC N1-N3 pins the order of the sum of squares, so the layout cannot change a bit."""
    pq_opt("PQ_RMS_WAVE_MAX", wave_max)
    rng = np.random.default_rng(cols + code)
    x = Q.from_f32((rng.standard_normal((70, cols)) * rng.uniform(0.01, 50.0, (70, 1))).astype(np.float32), code)
    w = Q.from_f32((1.0 + 0.3 * rng.standard_normal(cols)).astype(np.float32), code)
    want_q, want_s, want_h, _ = C.rmsnorm_quant_rowwise(x, w, 1e-5, code)
    qt, h = pq.rmsnorm_quantize(to_gpu(x, code), to_gpu(w, code), 1e-5, return_h=True)
    same(qt.int_data, want_q, "q"); same(qt.scale, want_s, "scale"); _same_h(h, want_h, code, "h")


@pytest.mark.parametrize("cols", [1025, 2043, 4102])
def test_rmsnorm_fp16_rounds_to_f32_before_fp16(pq, cols):
    """QSPEC N5 rounds x*rs to binary32 and THEN to the storage dtype.  hipcc used to fold the multiply and the fp16 conversion
    into v_fma_mixlo_f16 (one rounding of the exact product) in the ragged-width kernel: ~1 element in 10^4 differed by an ulp
    where the f32 product sits on an fp16 tie (found by tests/fuzz_quant.py; Elem<PQ_FP16>::from_f32 now pins the f32 value)."""
    rng = np.random.default_rng(cols)
    x = Q.from_f32((rng.standard_normal((384, cols)) * rng.choice([0.01, 1.0, 30.0], (384, 1))).astype(np.float32), 1)
    w = Q.from_f32((1 + 0.2 * rng.standard_normal(cols)).astype(np.float32), 1)
    want_q, want_s, want_h, _ = C.rmsnorm_quant_rowwise(x, w, 1e-6, 1)
    qt, h = pq.rmsnorm_quantize(to_gpu(x, 1), to_gpu(w, 1), 1e-6, return_h=True)
    same(qt.int_data, want_q, "q"); same(qt.scale, want_s, "scale"); _same_h(h, want_h, 1, "h (fp16, ragged width)")


def test_rmsnorm_many_rows_pins_sqrt_and_division(pq):
    """65536 rows of 8 values with variances spread over 60 binades: every row exercises the IEEE 1/sqrt(var + eps)."""
    rng = np.random.default_rng(8)
    rows, cols = 65536, 8
    x = (rng.standard_normal((rows, cols)) * np.exp2(rng.uniform(-30, 30, (rows, 1)))).astype(np.float32)
    w = np.ones(cols, np.float32)
    want_q, want_s, want_h, _ = C.rmsnorm_quant_rowwise(x, w, 1e-12, 2)
    qt, h = pq.rmsnorm_quantize(to_gpu(x, 2), to_gpu(w, 2), 1e-12, return_h=True)
    same(qt.int_data, want_q, "q"); same(qt.scale, want_s, "scale"); _same_h(h, want_h, 2, "h")


def test_rmsnorm_quantize_feeds_fused_qkv(pq):
    """rmsnorm_quantize -> FusedQLinear(q, k, v) on the QTensor == the oracle pipeline, bit for bit."""
    M, H = 96, 512
    gen = torch.Generator().manual_seed(9)
    x = (torch.randn(M, H, generator=gen) * 3).to(torch.bfloat16)
    wn = (1 + 0.1 * torch.randn(H, generator=gen)).to(torch.bfloat16)
    lins = [torch.nn.Linear(H, n, bias=False, dtype=torch.bfloat16) for n in (512, 128, 128)]
    with torch.no_grad():
        for l in lins:
            l.weight.copy_((torch.randn(l.weight.shape, generator=gen) * 0.04).to(torch.bfloat16))
    import copy
    fused = pq.FusedQLinear.from_linears(*(copy.deepcopy(l).cuda() for l in lins))
    outs = fused(pq.rmsnorm_quantize(x.cuda(), wn.cuda(), 1e-5))
    hq, hs, _, _ = C.rmsnorm_quant_rowwise(bits(x), bits(wn), 1e-5, 0)
    for l, o in zip(lins, outs):
        wq, ws = C.quant_rowwise(bits(l.weight), 0)
        same(o.contiguous(), C.qlinear_s8(hq, hs, wq, ws, None, 0), "fused qkv on rmsnorm_quantize")
    with pytest.raises(ValueError):
        pq.rmsnorm_quantize(x.cuda(), wn.cuda()[:100])


def test_full_size_cfg3_gated_mlp_fused(pq):
    """BASELINE config 3 at full size through GatedMLP (fused gate+up GEMM -> silu_mul_quantize -> down): the fused
    quantisation of the whole 2048 x 11008 intermediate equals the C oracle on the GPU-produced gate/up, bit for bit, and the
    block output equals the oracle's down projection on 64 sampled token rows."""
    M, H, I = 2048, 4096, 11008
    g = torch.Generator().manual_seed(78)
    x = torch.randn(M, H, generator=g).to(torch.bfloat16)
    ws = {n: (torch.randn(o, i, generator=g) * 0.02).to(torch.bfloat16) for n, (o, i) in
          (("gate", (I, H)), ("up", (I, H)), ("down", (H, I)))}
    lins = {}
    for n, w in ws.items():
        lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device="cuda", dtype=torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_(w.cuda())
        lins[n] = lin
    mlp = pq.GatedMLP.from_linears(lins["gate"], lins["up"], lins["down"])
    gate, up = mlp.gate_up(x.cuda())
    hq = pq.silu_mul_quantize(gate, up)
    y = mlp.down(hq)
    same(y, bits(mlp(x.cuda())), "GatedMLP.forward == its parts")
    want_q, want_s, _ = C.silu_mul_quant_rowwise(bits(gate), bits(up), 0, want_h=False)
    same(hq.int_data, want_q, "cfg3 fused silu codes"); same(hq.scale, want_s, "cfg3 fused silu scales")
    rows = np.random.default_rng(2).choice(M, 64, replace=False)
    dq, ds = C.quant_rowwise(bits(ws["down"]), 0)
    acc = (want_q[rows].astype(np.int64) @ dq.astype(np.int64).T).astype(np.int32)
    same(y[torch.from_numpy(rows).cuda()].contiguous(), Q.epilogue(acc, want_s[rows], ds, None, 0), "cfg3 down rows")


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("rows,cols,pad_q,off_q", [(37, 4096, 64, 0), (9, 1000, 8, 8), (5, 777, 3, 1), (64, 11008, 16, 16), (3, 256, 0, 4)])
def test_c_abi_strided_code_outputs(pq, code, rows, cols, pad_q, off_q):
    """The C-ABI's ld_q / ld_out arguments: codes written into a wider, offset buffer (vector and generic paths) by K1, K2,
    K1s and K1n; dequant reading them back from there.  Bytes outside the [rows, cols] window stay untouched."""
    from protoquant_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(rows * 31 + cols + code)
    x = Q.from_f32((rng.standard_normal((rows, cols)) * 2).astype(np.float32), code)
    u = Q.from_f32(rng.standard_normal((rows, cols)).astype(np.float32), code)
    w = Q.from_f32((1 + 0.1 * rng.standard_normal(cols)).astype(np.float32), code)
    xt, ut, wt = to_gpu(x, code), to_gpu(u, code), to_gpu(w, code)
    ldq = cols + pad_q + off_q
    st = torch.cuda.current_stream().cuda_stream

    def fresh():
        return torch.full((rows, ldq), 77, dtype=torch.int8, device="cuda"), torch.empty(max(rows, cols), dtype=torch.float32, device="cuda")

    def check(qbuf, want_q, what):
        got = qbuf.cpu().numpy()
        assert np.array_equal(got[:, off_q:off_q + cols], want_q), what
        mask = np.ones(got.shape, bool); mask[:, off_q:off_q + cols] = False
        assert np.all(got[mask] == 77), what + ": wrote outside its window"

    qb, sc = fresh()
    _lib.check(L.pq_quant_rowwise(xt.data_ptr(), code, rows, cols, cols, qb.data_ptr() + off_q, ldq, sc.data_ptr(), st), "k1")
    wq, ws = C.quant_rowwise(x, code); check(qb, wq, "K1"); same(sc[:rows], ws, "K1 scale")
    out = torch.empty((rows, cols + 8), dtype=TD[code], device="cuda")
    _lib.check(L.pq_dequant(qb.data_ptr() + off_q, ldq, sc.data_ptr(), 1, rows, cols, out.data_ptr(), cols + 8, code, st), "dequant")
    _same_h(out[:, :cols].contiguous(), C.dequant(wq, ws, 1, code), code, "dequant from a strided code buffer")
    qb, sc = fresh()
    _lib.check(L.pq_quant_colwise(xt.data_ptr(), code, rows, cols, cols, qb.data_ptr() + off_q, ldq, sc.data_ptr(), st), "k2")
    cq, cs = C.quant_colwise(x, code); check(qb, cq, "K2"); same(sc[:cols], cs, "K2 scale")
    qb, sc = fresh()
    _lib.check(L.pq_silu_mul_quant_rowwise(xt.data_ptr(), cols, ut.data_ptr(), cols, code, rows, cols, qb.data_ptr() + off_q, ldq, sc.data_ptr(), None, 0, st), "k1s")
    sq, ss, _ = C.silu_mul_quant_rowwise(x, u, code); check(qb, sq, "K1s"); same(sc[:rows], ss, "K1s scale")
    qb, sc = fresh()
    _lib.check(L.pq_rmsnorm_quant_rowwise(xt.data_ptr(), cols, wt.data_ptr(), 1e-5, code, rows, cols, qb.data_ptr() + off_q, ldq, sc.data_ptr(), None, 0, st), "k1n")
    nq, ns, _, _ = C.rmsnorm_quant_rowwise(x, w, 1e-5, code); check(qb, nq, "K1n"); same(sc[:rows], ns, "K1n scale")


def test_swap_linears_on_a_transformers_llama(pq):
    """A tiny randomly initialised transformers LlamaForCausalLM: swap_linears(fuse_gated_mlp=True) turns every projection into
    qlinear and every LlamaMLP into GatedMLP; the logits stay close to the bf16 model's (int8 noise only), and the fused MLP
    block equals the oracle chain bit for bit."""
    tr = pytest.importorskip("transformers")
    torch.manual_seed(0)
    cfg = tr.LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=2, max_position_embeddings=128)
    model = tr.LlamaForCausalLM(cfg).to(torch.bfloat16).cuda().eval()
    ids = torch.randint(0, 512, (2, 48), device="cuda")
    with torch.no_grad():
        ref = model(ids).logits.float()
        mlp0 = model.model.layers[0].mlp
        wts = {n: getattr(mlp0, n).weight.detach().cpu().clone() for n in ("gate_proj", "up_proj", "down_proj")}
        pq.swap_linears(model, predicate=lambda n, m: n != "lm_head", fuse_gated_mlp=True)
        assert isinstance(model.model.layers[0].mlp, pq.GatedMLP) and isinstance(model.model.layers[1].self_attn.q_proj, pq.qlinear)
        assert isinstance(model.lm_head, torch.nn.Linear)
        got = model(ids).logits.float()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0)
        assert float(cos) > 0.995, float(cos)
        x = torch.randn(40, 256, device="cuda").to(torch.bfloat16)
        y = model.model.layers[0].mlp(x)
    xq, xs = C.quant_rowwise(bits(x), 0)
    w = {n: C.quant_rowwise(bits(t), 0) for n, t in wts.items()}
    hq, hs, _ = C.silu_mul_quant_rowwise(C.qlinear_s8(xq, xs, *w["gate_proj"], None, 0), C.qlinear_s8(xq, xs, *w["up_proj"], None, 0), 0)
    same(y, C.qlinear_s8(hq, hs, *w["down_proj"], None, 0), "swapped LlamaMLP == oracle chain")


def test_randomized_shape_sweep(pq):
    """Seeded sweep: 60 random (M, N, K, dtype, bias) problems — ragged tiles on the MFMA fast path (K % 128 == 0),
    arbitrary K on the generic path — qlinear bits and int32 accumulators vs the oracle."""
    rng = np.random.default_rng(2026)
    for case in range(60):
        M = int(rng.integers(1, 700)); N = int(rng.integers(1, 700))
        K = int(rng.integers(1, 9)) * 128 if case % 2 == 0 else int(rng.integers(1, 600))
        code = int(rng.integers(0, 3)); bias = bool(rng.integers(0, 2))
        x = Q.from_f32((rng.standard_normal((M, K)) * rng.uniform(0.1, 10)).astype(np.float32), code)
        w = Q.from_f32((rng.standard_normal((N, K)) * 0.05).astype(np.float32), code)
        b = Q.from_f32((rng.standard_normal(N) * 0.1).astype(np.float32), code) if bias else None
        wq, wsc = C.quant_rowwise(w, code)
        y_want, xq_want, xs_want, acc_want = Q.qlinear(x, code, wq, wsc, b)
        tag = f"case {case}: {M}x{N}x{K} dt{code} bias={bias}"
        q = pq.quantize(to_gpu(x, code))
        same(q.int_data, xq_want, tag + " xq"); same(q.scale, xs_want, tag + " xs")
        wq_t, ws_t = torch.from_numpy(wq).cuda(), torch.from_numpy(wsc).cuda()
        same(pq.int_mm(q.int_data, wq_t), acc_want, tag + " acc")
        y = pq.qlinear_s8(q.int_data, q.scale, wq_t, ws_t, to_gpu(b, code) if bias else None, TD[code])
        same(y, y_want, tag + " y")


def test_qlinear_state_dict_roundtrip(pq):
    """The pre-quantised weight (wq int8, ws f32, bias) is plain module state: save -> load -> identical outputs."""
    import io
    torch.manual_seed(12)
    lin = torch.nn.Linear(384, 256, bias=True, device="cuda", dtype=torch.bfloat16)
    m = pq.qlinear.from_linear(lin)
    buf = io.BytesIO(); torch.save(m.state_dict(), buf); buf.seek(0)
    m2 = pq.qlinear(384, 256, bias=True, device="cuda", dtype=torch.bfloat16)
    m2.load_state_dict(torch.load(buf))
    x = torch.randn(33, 384, device="cuda", dtype=torch.bfloat16)
    assert torch.equal(m(x).view(torch.int16), m2(x).view(torch.int16))
    assert sorted(m.state_dict()) == ["bias", "wq", "ws"]


def test_qlinear_dyn_matches_two_call_path(pq):
    """pq_qlinear_dyn (one call, workspace scratch) == quantize() + qlinear_s8(), incl. a split-K shape and [..., K] input."""
    torch.manual_seed(31)
    for (lead, K, N, dt, bias) in (((7, 33), 512, 384, torch.bfloat16, True), ((512,), 4096, 1024, torch.bfloat16, False),
                                   ((50,), 200, 130, torch.float32, True), ((300,), 256, 256, torch.float16, False)):
        x = torch.randn(*lead, K, device="cuda").to(dt)
        lin = torch.nn.Linear(K, N, bias=bias, device="cuda", dtype=dt)
        m = pq.qlinear.from_linear(lin)
        y1 = m(x)                                                       # forward() uses the one-call path
        q = pq.quantize(x)
        y2 = pq.qlinear_s8(q.int_data.reshape(-1, K), q.scale, m.wq, m.ws, m.bias, dt).reshape(*lead, N)
        assert y1.shape == y2.shape and torch.equal(y1.contiguous().view(torch.uint8), y2.contiguous().view(torch.uint8))


def test_qlinear_is_graph_capturable(pq):
    """The C-ABI neither allocates nor synchronises: a qlinear forward (incl. a split-K shape) can be captured in a
    hipGraph and replayed on new inputs with the same bits as the eager call."""
    torch.manual_seed(41)
    for (M, K, N) in ((256, 512, 384), (512, 4096, 1024)):
        m = pq.qlinear.from_linear(torch.nn.Linear(K, N, bias=True, device="cuda", dtype=torch.bfloat16))
        x_static = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        m(x_static)                                        # warm up: workspace allocated outside the capture
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x_static)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y_static = m(x_static)
        for _ in range(3):
            x_new = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            x_static.copy_(x_new)
            g.replay()
            want = m(x_new)
            torch.cuda.synchronize()
            assert torch.equal(y_static.view(torch.int16), want.view(torch.int16))


def test_errors_are_loud(pq):
    from protoquant_amd import _lib
    with pytest.raises(_lib.PQError):
        pq.quantize(torch.randn(4, 4))                       # CPU tensor: no fallback
    with pytest.raises(TypeError):
        pq.quantize(torch.zeros(4, 4, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        pq.int_mm(torch.zeros(4, 8, dtype=torch.int8, device="cuda"), torch.zeros(4, 16, dtype=torch.int8, device="cuda"))
    L = _lib.lib()
    st = L.pq_quant_rowwise(None, 0, 4, 4, 2, None, 4, None, None)
    assert st == 1 and b"pq_quant_rowwise" in L.pq_last_error()


@pytest.mark.parametrize("M,N,K,want", [(48, 6144, 4096, "ring64x64"), (64, 28672, 4096, "ring64x128"), (24, 28672, 4096, "skinny"), (64, 5120, 11008, "ring64x64"), (48, 5120, 4096, "ring64x64"), (32, 5120, 11008, "skinny"), (16, 28672, 8192, "ring64x128"), (16, 14336, 8192, "ring64x64"), (8, 28672, 8192, "skinny"), (16, 28672, 4096, "skinny"), (24, 28672, 8192, "ring64x128"), (25, 28672, 4096, "ring64x128"), (24, 14336, 4096, "ring64x64"), (64, 4096, 4096, "skinny"), (65, 4096, 4096, "ring64x64"), (128, 4096, 4096, "ring64x64"), (200, 1000, 2048, "ring64x64"), (256, 4096, 4096, "ring64x64"),
                                        (384, 4096, 14336, "ring64x128"), (512, 4096, 4096, "ring64x128"), (512, 4096, 14336, "ring64x128"), (500, 4000, 1152, "ring64x128"),
                                        (128, 28672, 4096, "ring128"), (512, 28672, 4096, "sp256"), (1024, 1024, 8192, "ring64x64"), (2048, 1024, 8192, "ring64x128"),
                                        (640, 2048, 4096, "ring64x128"), (4096, 1024, 8192, "ring128")])
def test_mid_m_regime_exact(pq, M, N, K, want, pq_opt):
    """64 < M <= 512 (round-3 verdict item 4) and other small grids: the 64-row ring tiles of gemm_s8_ring.hip, chosen by the dispatcher when the 128 x 128 ring tiles
    would fill well under the chip.  Full-range int8 operands: int32 accumulator == an exact float64 matmul, y == the QSPEC epilogue (bias, every output dtype), ==
    the round-3 dispatch (PQ_NO_MIDM=1) bit for bit, ragged M / N included."""
    from protoquant_amd import _lib
    name = _lib.lib().pq_gemm_variant_name(M, N, K, K, K).decode()
    assert name.startswith(want), name
    assert _lib.lib().pq_qlinear_workspace_bytes(M, N, K) == 0 or not want.startswith("ring64")
    rng = np.random.default_rng(M * 31 + N * 7 + K)
    a = rng.integers(-128, 128, (M, K), dtype=np.int8); b = rng.integers(-128, 128, (N, K), dtype=np.int8)
    acc = (a.astype(np.float64) @ b.astype(np.float64).T).astype(np.int32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    same(pq.int_mm(ta, tb), acc, f"acc [{name}]")
    xs = rng.random(M).astype(np.float32) * 0.1; ws = rng.random(N).astype(np.float32) * 0.01
    txs, tws = torch.from_numpy(xs).cuda(), torch.from_numpy(ws).cuda()
    for code in (0, 1, 2):
        bv = Q.from_f32(rng.standard_normal(N).astype(np.float32), code) if code != 1 else None
        y = pq.qlinear_s8(ta, txs, tb, tws, to_gpu(bv, code) if bv is not None else None, TD[code])
        same(y, Q.epilogue(acc, xs, ws, bv, code), f"y dt{code} [{name}]")
        if code == 0:
            pq_opt("PQ_NO_MIDM", "1")
            assert not _lib.lib().pq_gemm_variant_name(M, N, K, K, K).decode().startswith("ring64")
            y3 = pq.qlinear_s8(ta, txs, tb, tws, to_gpu(bv, code), TD[code])
            pq_opt("PQ_NO_MIDM", "")
            assert torch.equal(y.view(torch.int16), y3.view(torch.int16)), "mid-M tiles vs the round-3 dispatch"


@pytest.mark.parametrize("M,N,K", [(0, 64, 128), (33, 0, 128), (0, 0, 128), (17, 40, 0), (0, 40, 0), (300, 520, 0)])
def test_empty_problems(pq, M, N, K):
    """Empty inputs: no rows, no output channels, or an empty reduction (K = 0: every accumulator is 0, y = 0 * xs * ws (+ bias) like the oracle's) — through the int32
    twin, the fused qlinear (both orientations), the one-call dynamic form and the module with an empty batch."""
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
    xs = torch.rand(M, device="cuda") + 0.5; ws = torch.rand(N, device="cuda") + 0.5
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    acc = pq.int_mm(a, b)
    assert acc.shape == (M, N) and acc.dtype == torch.int32 and not bool(acc.any())
    want = Q.epilogue(np.zeros((M, N), np.int32), xs.cpu().numpy(), ws.cpu().numpy(), bits(bias), 0)
    y = pq.qlinear_s8(a, xs, b, ws, bias, torch.bfloat16)
    assert y.shape == (M, N)
    same(y, want, "y")
    yt = pq.qlinear_s8_t(a, xs, b, ws, bias, torch.bfloat16)
    assert yt.shape == (N, M)
    same(yt.t().contiguous(), want, "y^T")
    if K > 0:
        lin = torch.nn.Linear(K, max(N, 1), bias=True, device="cuda", dtype=torch.bfloat16)
        m = pq.qlinear.from_linear(lin)
        assert m(torch.empty((0, K), dtype=torch.bfloat16, device="cuda")).shape == (0, max(N, 1))
        assert m(torch.empty((2, 0, K), dtype=torch.bfloat16, device="cuda")).shape == (2, 0, max(N, 1))
        x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        qw = pq.quantize(torch.randn((max(N, 1), K), device="cuda").to(torch.bfloat16))
        yd = pq.qlinear_dyn(x, qw.int_data, qw.scale, None)
        assert yd.shape == (M, max(N, 1))


@pytest.mark.parametrize("K", [4000, 1600, 4104, 200, 72, 129])
@pytest.mark.parametrize("M", [1, 64, 300])
def test_modules_pad_k_to_the_mfma_k_tile(pq, K, M):
    """in_features that is not a multiple of 128: the modules run the MFMA tiles over a zero-padded K (weights padded once, activation codes quantised into a
    zero-tailed buffer) instead of the generic kernel — and every output bit is the unpadded problem's: against the oracle, for float and QTensor inputs, through
    qlinear and FusedQLinear, and again after load_state_dict replaced the weights (the padded copy must follow)."""
    N = 384
    torch.manual_seed(K + M)
    lin = torch.nn.Linear(K, N, bias=True, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(2, M, K, device="cuda", dtype=torch.bfloat16)
    m = pq.qlinear.from_linear(lin)
    assert tuple(m.wq.shape) == (N, K)                                   # the state-dict weight stays unpadded
    wq, ws = C.quant_rowwise(bits(lin.weight), 0)
    want, _, _, _ = Q.qlinear(bits(x.reshape(-1, K)), 0, wq, ws, bits(lin.bias))
    y = m(x)
    assert tuple(y.shape) == (2, M, N)
    same(y.reshape(-1, N), want, "padded-K qlinear")
    same(m(pq.quantize(x)).reshape(-1, N), want, "padded-K qlinear, QTensor input")
    f = pq.FusedQLinear([m, pq.qlinear.from_linear(lin)])
    a, b = f(x)
    same(a.reshape(-1, N), want, "padded-K fused, first"); same(b.reshape(-1, N), want, "padded-K fused, second")
    a, b = f(pq.quantize(x))
    same(b.reshape(-1, N), want, "padded-K fused, QTensor input")
    # new weights through load_state_dict: the cached padded copy is rebuilt
    lin2 = torch.nn.Linear(K, N, bias=True, device="cuda", dtype=torch.bfloat16)
    m.load_state_dict(pq.qlinear.from_linear(lin2).state_dict())
    wq2, ws2 = C.quant_rowwise(bits(lin2.weight), 0)
    want2, _, _, _ = Q.qlinear(bits(x.reshape(-1, K)), 0, wq2, ws2, bits(lin2.bias))
    same(m(x).reshape(-1, N), want2, "padded-K qlinear after load_state_dict")


def test_padded_k_runs_the_mfma_tiles(pq):
    """... and it is the point of the exercise: 4096 x 4096 x 4000 through the module takes about what K = 4096 takes, not the generic kernel's 7x."""
    def t_us(m, x):
        for _ in range(3):
            m(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            m(x)
        b.record(); b.synchronize()
        return a.elapsed_time(b) * 100
    t = {}
    for K in (4096, 4000):
        lin = torch.nn.Linear(K, 4096, bias=False, device="cuda", dtype=torch.bfloat16)
        t[K] = t_us(pq.qlinear.from_linear(lin), torch.randn(4096, K, device="cuda", dtype=torch.bfloat16))
    assert t[4000] < 2.0 * t[4096], t


def test_sharded_modules_pad_k_too(pq):
    """The sharded forms with an in_features (or a K slice per rank: 11008 / 8 = 1376) that is not a multiple of 128: rows, row chunks, transposed shards and the
    row-sharded partial all run the padded operands (gemm_operands) and keep the bits of the plain module / the oracle partial."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        torch.manual_seed(5)
        K, N = 1376, 640
        lin = torch.nn.Linear(K, N, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(3, 70, K, device="cuda", dtype=torch.bfloat16)
        wq, ws = C.quant_rowwise(bits(lin.weight), 0)
        want, _, _, _ = Q.qlinear(bits(x.reshape(-1, K)), 0, wq, ws, bits(lin.bias))
        y0 = pq.qlinear.from_linear(lin)(x)
        same(y0.reshape(-1, N), want, "plain module")
        for kw in ({}, {"overlap_chunks": 3}, {"layout": "transposed"}):
            y1 = pq.ColumnShardedQLinear.from_linear(lin, **kw)(x)
            torch.cuda.synchronize()
            assert torch.equal(y0.view(torch.int16), y1.contiguous().view(torch.int16)), kw
        # row-sharded: rank 1 of 8 over K = 11008 owns 1376 input features
        big = torch.nn.Linear(11008, 256, bias=False, device="cuda", dtype=torch.bfloat16)
        xb = torch.randn(40, 11008, device="cuda", dtype=torch.bfloat16)
        lay = pq.RowShardedQLinear.from_linear(big, world=8, rank=1)
        assert lay.local.in_features == 1376
        k0, k1 = 1376, 2752
        p = lay.partial(xb[:, k0:k1].contiguous())
        wq, ws = C.quant_rowwise(bits(big.weight), 0)                       # full-row weight scales, then the K slice (RowShardedQLinear.shard_of)
        xq, xs = C.quant_rowwise(bits(xb[:, k0:k1].contiguous()), 0)
        same(p, Q.epilogue(Q.gemm_s8s8s32(xq, np.ascontiguousarray(wq[:, k0:k1])), xs, ws, None, 2), "row-sharded partial")
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("K", [4000, 1608, 1001])
def test_functional_entry_points_pad_k_when_it_pays(pq, K):
    """int_mm / qlinear_s8 / qlinear_s8_t / qlinear_dyn with a K that is not a multiple of 128 on a problem large enough (>= 2^31 multiply-adds): both operands are
    copied into zero-tailed buffers and the MFMA tiles run — every bit as from the unpadded operands (torch._int_mm on this GPU; QSPEC E1-E4 in stock torch ops)."""
    M, N = 1536, 2048
    g = torch.Generator(device="cuda"); g.manual_seed(K)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.02 + 1e-3
    ws = torch.rand(N, device="cuda", generator=g) * 0.002 + 1e-4
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    kp = -(-K // 8) * 8                                            # torch._int_mm wants K % 8 == 0: its own zero padding
    ap = torch.zeros((M, kp), dtype=torch.int8, device="cuda"); ap[:, :K] = a
    bp = torch.zeros((N, kp), dtype=torch.int8, device="cuda"); bp[:, :K] = b
    acc = torch._int_mm(ap, bp.t())
    assert torch.equal(pq.int_mm(a, b), acc)
    ref = ((acc.float() * xs[:, None]) * ws[None, :] + bias.float()[None, :]).to(torch.bfloat16)
    assert torch.equal(pq.qlinear_s8(a, xs, b, ws, bias, torch.bfloat16).view(torch.int16), ref.view(torch.int16))
    assert torch.equal(pq.qlinear_s8_t(a, xs, b, ws, bias, torch.bfloat16).t().contiguous().view(torch.int16), ref.view(torch.int16))
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    q = pq.quantize(x)
    y = pq.qlinear_dyn(x, b, ws, bias)
    assert torch.equal(y.view(torch.int16), pq.qlinear_s8(q.int_data, q.scale, b, ws, bias, torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("setting", ["", "0"])
@pytest.mark.parametrize("M,N,K", [(512, 4001, 512), (300, 777, 640), (4096, 1025, 1024), (128, 50257, 256)])
def test_odd_leading_dimensions_of_y(pq, pq_opt, setting, M, N, K):
    """Output rows that are not 16-byte aligned (odd N: a 50257-wide vocabulary): by default the staged epilogue stores its 16-byte pieces at element-aligned addresses
    (gfx950 compute queues run with unaligned access enabled; -26 % at 2048 x 50257 x 4096); PQ_EPI_ANY_ALIGN=0 keeps the round-3 behaviour (guarded direct stores).
    Same bits either way, for every output type."""
    pq_opt("PQ_EPI_ANY_ALIGN", setting)
    g = torch.Generator(device="cuda"); g.manual_seed(N)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.02 + 1e-3
    ws = torch.rand(N, device="cuda", generator=g) * 0.002 + 1e-4
    acc = pq.int_mm(a, b)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        bias = torch.randn(N, device="cuda", generator=g).to(dtype)
        ref = ((acc.float() * xs[:, None]) * ws[None, :] + bias.float()[None, :]).to(dtype)
        y = pq.qlinear_s8(a, xs, b, ws, bias, dtype)
        view = torch.int16 if dtype != torch.float32 else torch.int32
        assert torch.equal(y.view(view), ref.view(view)), dtype
        # a padded, offset output buffer: nothing outside [M, N] is touched
        big = torch.full((M + 2, N + 5), 7.0, dtype=dtype, device="cuda")
        out = big[1:M + 1, 3:N + 3]
        pq.qlinear_s8(a, xs, b, ws, bias, dtype, out=out)
        assert torch.equal(out.contiguous().view(view), ref.view(view))
        big[1:M + 1, 3:N + 3] = 7.0
        assert bool((big == 7.0).all()), "stores outside the output"


def test_kn_stored_weights_and_a_transformers_gpt2(pq):
    """Weights stored [in_features, out_features] (Hugging Face GPT-2's Conv1D): qlinear.from_kn_weight quantises along the strided axis (K2) and is, bit for bit,
    from_linear on the transposed weight; swap_linears turns every Conv1D of a transformers GPT2LMHeadModel — and its 50257-wide lm_head (odd N: the unaligned staged
    epilogue) — into qlinear, and the swapped model's logits equal those of the same model with its Conv1D layers rewritten as nn.Linear and swapped."""
    torch.manual_seed(11)
    w_kn = torch.randn(640, 392, device="cuda").to(torch.bfloat16)
    b = torch.randn(392, device="cuda").to(torch.bfloat16)
    m1 = pq.qlinear.from_kn_weight(w_kn, b)
    lin = torch.nn.Linear(640, 392, bias=True, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w_kn.t()); lin.bias.copy_(b)
    m2 = pq.qlinear.from_linear(lin)
    assert torch.equal(m1.wq, m2.wq) and torch.equal(m1.ws.view(torch.int32), m2.ws.view(torch.int32))
    x = torch.randn(3, 17, 640, device="cuda").to(torch.bfloat16)
    assert torch.equal(m1(x).view(torch.int16), m2(x).view(torch.int16))

    tr = pytest.importorskip("transformers")
    cfg = tr.GPT2Config(n_embd=256, n_layer=2, n_head=4, n_positions=64, vocab_size=50257)
    model = tr.GPT2LMHeadModel(cfg).to(torch.bfloat16).cuda().eval()
    import copy
    twin = copy.deepcopy(model)
    n_conv = sum(1 for mod in model.modules() if type(mod).__name__ == "Conv1D")
    assert n_conv == 8                                              # c_attn, c_proj, c_fc, c_proj per layer
    for parent in list(twin.modules()):                             # the twin: every Conv1D rewritten as the nn.Linear it is
        for name, child in list(parent.named_children()):
            if type(child).__name__ == "Conv1D":
                l = torch.nn.Linear(child.weight.shape[0], child.nf, bias=True, device="cuda", dtype=torch.bfloat16)
                with torch.no_grad():
                    l.weight.copy_(child.weight.t()); l.bias.copy_(child.bias)
                setattr(parent, name, l)
    pq.swap_linears(model); pq.swap_linears(twin)
    assert sum(1 for mod in model.modules() if isinstance(mod, pq.qlinear)) == n_conv + 1      # + lm_head
    ids = torch.randint(0, 50257, (2, 48), device="cuda")
    with torch.no_grad():
        a = model(ids).logits; bb = twin(ids).logits
    assert a.shape == (2, 48, 50257) and torch.equal(a.view(torch.int16), bb.view(torch.int16))


def test_padded_k_module_is_graph_capturable(pq):
    """The padded-K forward (fresh code buffer, zeroed tail, K1 with a wider ld_q, the GEMM over the padded weight copy) under hipGraph capture and replay."""
    torch.manual_seed(2)
    lin = torch.nn.Linear(1000, 512, bias=True, device="cuda", dtype=torch.bfloat16)
    m = pq.qlinear.from_linear(lin)
    x = torch.randn(64, 1000, device="cuda", dtype=torch.bfloat16)
    want = m(x).clone()                                             # warm-up: builds the padded weight copy outside the capture
    out = torch.empty_like(want)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out.copy_(m(x))
        for _ in range(3):
            out.zero_(); g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    x.copy_(torch.randn(64, 1000, device="cuda").to(torch.bfloat16))
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), m(x).view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (16, 4096, 4096), (16, 777, 512), (40, 4100, 1024), (64, 4096, 14336), (24, 28672, 4096), (33, 6144, 4096)])
def test_transposed_output_with_few_tokens(pq, M, N, K):
    """pq_qlinear_s8_t with few tokens: the weight-streaming kernel computes the product in its normal orientation and stores it transposed (the swapped form would be a
    16-column problem for the tile kernels: 16 x 4096 x 4096 took 23 us, now 5.8) — yt[n][m] == y[m][n] bit for bit, every output type, with bias, into a padded buffer."""
    g = torch.Generator(device="cuda"); g.manual_seed(M + N)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    xs = torch.rand(M, device="cuda", generator=g) * 0.02 + 1e-3
    ws = torch.rand(N, device="cuda", generator=g) * 0.002 + 1e-4
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for bias in (None, torch.randn(N, device="cuda", generator=g).to(dtype)):
            y = pq.qlinear_s8(a, xs, b, ws, bias, dtype)
            yt = pq.qlinear_s8_t(a, xs, b, ws, bias, dtype)
            assert yt.shape == (N, M) and torch.equal(yt.t().contiguous(), y), (dtype, bias is not None)
            big = torch.full((N + 2, M + 3), 7.0, dtype=dtype, device="cuda")
            pq.qlinear_s8_t(a, xs, b, ws, bias, dtype, out=big[1:N + 1, 2:M + 2])
            assert torch.equal(big[1:N + 1, 2:M + 2].t().contiguous(), y)
            big[1:N + 1, 2:M + 2] = 7.0
            assert bool((big == 7.0).all())


@pytest.mark.parametrize("rows,cols,code", [(64, 32768, 0), (40, 40960, 0), (33, 53248, 0), (16, 65536, 1), (8, 65544, 0), (24, 28672, 2), (12, 32768, 2), (5, 32776, 2)])
def test_rowwise_quant_of_very_wide_rows(pq, rows, cols, code):
    """Rows past the round-3 register-resident limit (32 768 bf16 / 16 384 f32 columns): up to 65 536 / 32 768 columns the row still stays in registers (32 vectors per
    lane: 4096 x 53248 bf16 352 -> 118 us), beyond that the generic kernel takes over — same codes and scales as the oracle either way, incl. a NaN row and an all-zero row."""
    g = torch.Generator(device="cuda"); g.manual_seed(cols)
    x = (torch.randn(rows, cols, device="cuda", generator=g) * 3).to(TD[code])
    x[1] = 0
    x[2, cols - 1] = float("nan")
    x[3, cols // 2] = 1e4
    q = pq.quantize(x)
    wq, ws = Q.quantize(x.cpu().numpy() if code == 2 else bits(x), code, 1)
    assert np.array_equal(q.int_data.cpu().numpy(), wq)
    assert np.array_equal(bits(q.scale), ws.view(np.uint32))
