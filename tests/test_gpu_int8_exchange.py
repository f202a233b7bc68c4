"""-m gpu: the int8-code exchange of the column-sharded gated MLP (BASELINE config 5, gate/up -> down both column-sharded): the two halves of K1s
(pq_silu_mul_rowamax, pq_silu_mul_quant_rowwise_amax), the GEMM on stacked code blocks (pq_qlinear_s8_kslabs) and ColumnShardedGatedMLP — G ranks
played offline on one GPU (as test_llama70b_row_sharded_real_shapes does for the row-sharded pairing), against the C / numpy oracle and against the
unsharded GatedMLP, bit for bit; a 1-rank RCCL communicator under hipGraph capture.  Real ranks: tests/rccl_rank_worker.py."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import qspec_numpy as Q
from tests.gpu_util import TD, bits, same, to_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pq():
    import protoquant_amd
    from protoquant_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return protoquant_amd


def _split(n, parts):
    q, r = divmod(n, parts)
    out, lo = [], 0
    for i in range(parts):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi)); lo = hi
    return out


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("rows,cols,parts", [(64, 4096, 8), (37, 3584, 1), (5, 1000, 3), (130, 11008, 8), (3, 77, 2), (9, 28672, 8), (16, 40000, 2)])
def test_split_silu_quant_equals_the_fused_kernel_and_the_oracle(pq, code, rows, cols, parts):
    """column blocks: local amax per block (vector and generic layouts, ragged and unaligned widths), integer max over the blocks, encode per block == the oracle's
    S1-S6 on the whole row and == the one-pass kernel; special values (NaN, Inf, zero rows) propagate exactly as in K1s."""
    rng = np.random.default_rng(rows * 7 + cols + code)
    g = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
    u = rng.standard_normal((rows, cols)).astype(np.float32)
    if rows >= 5:
        g[1, cols // 2] = np.nan; u[2, 3] = np.inf; g[3, :] = 0; u[4, :] = 0; g[0, 0] = 200.0; g[0, 1] = -200.0
    gs, us = Q.from_f32(g, code), Q.from_f32(u, code)
    want_q, want_s, _ = C.silu_mul_quant_rowwise(gs, us, code, want_h=False)
    gt, ut = to_gpu(gs, code), to_gpu(us, code)
    blocks = _split(cols, parts)
    am = [pq.silu_mul_rowamax(gt[:, a:b], ut[:, a:b]) for a, b in blocks]
    for (a, b), t in zip(blocks, am):           # each block's amax is the oracle's (NaN as a class: above +Inf)
        want = Q.row_amax_bits(Q.silu_mul(gs[:, a:b], us[:, a:b], code), code)
        got = t.cpu().numpy().view(np.uint32)
        nan = want > 0x7F800000
        assert np.array_equal(got > 0x7F800000, nan) and np.array_equal(got[~nan], want[~nan])
    glob = torch.stack(am).max(dim=0).values
    q = torch.empty((rows, cols), dtype=torch.int8, device="cuda")
    scales = []
    for a, b in blocks:
        scales.append(pq.silu_mul_quantize_with_amax(gt[:, a:b], ut[:, a:b], glob, out=q[:, a:b]).scale)
    same(q, want_q, "codes of the blocks")
    for s in scales:
        same(s, want_s, "row scales")
    one = pq.silu_mul_quantize(gt, ut)
    assert torch.equal(one.int_data, q) and torch.equal(one.scale.view(torch.int32), scales[0].view(torch.int32))


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("rows,cols,parts", [(64, 8192, 8), (37, 1024, 1), (5, 1000, 3), (130, 4096, 4), (3, 77, 2), (16, 40000, 2)])
def test_split_plain_quant_equals_k1_and_the_oracle(pq, code, rows, cols, parts):
    """the same two halves for a PLAIN activation (pq_quant_rowamax / pq_quant_rowwise_amax: a rank's heads of the attention output): block amax, integer max over the
    blocks, encode per block == the oracle's Q1-Q6 on the whole row == K1; NaN / Inf / zero rows included."""
    rng = np.random.default_rng(rows * 13 + cols + code)
    x = (rng.standard_normal((rows, cols)) * 2).astype(np.float32)
    if rows >= 5:
        x[1, cols // 2] = np.nan; x[2, 3] = np.inf; x[3, :] = 0; x[4, -1] = -np.inf
    xs = Q.from_f32(x, code)
    want_q, want_s = C.quant_rowwise(xs, code)
    xt = to_gpu(xs, code)
    blocks = _split(cols, parts)
    am = [pq.rowamax(xt[:, a:b]) for a, b in blocks]
    glob = torch.stack(am).max(dim=0).values
    q = torch.empty((rows, cols), dtype=torch.int8, device="cuda")
    scales = [pq.quantize_with_amax(xt[:, a:b], glob, out=q[:, a:b]).scale for a, b in blocks]
    same(q, want_q, "codes of the blocks")
    for s in scales:
        same(s, want_s, "row scales")
    one = pq.quantize(xt)
    assert torch.equal(one.int_data, q) and torch.equal(one.scale.view(torch.int32), scales[0].view(torch.int32))


@pytest.mark.parametrize("M,N,K,G,bias", [(300, 384, 1024, 2, True), (129, 1000, 2048, 8, False), (4096, 8192, 8192, 8, False), (50, 96, 768, 3, True)])
def test_column_sharded_qlinear_on_a_column_sharded_input(pq, M, N, K, G, bias):
    """ColumnShardedQLinear.forward_sharded_input with G ranks played offline (the 70B `o` projection at its real shape among them: every rank holds 1024 of the 8192
    attention-output features and 1024 of the 8192 output channels): the int8-code exchange gives forward()'s result on the concatenated activation, bit for bit."""
    torch.manual_seed(M + N)
    lin = torch.nn.Linear(K, N, bias=bias, device="cuda", dtype=torch.bfloat16)
    x = (torch.randn(M, K, device="cuda") * 1.3).to(torch.bfloat16)
    y_ref = pq.qlinear.from_linear(lin)(x)
    qx = pq.quantize(x)
    mods = [pq.ColumnShardedQLinear.from_linear(lin, world=G, rank=r) for r in range(G)]
    kb = _split(K, G)
    am = [m._local_amax(x[:, a:b]) for m, (a, b) in zip(mods, kb)]
    glob = torch.stack(am).max(dim=0).values
    enc = [m._encode(x[:, a:b], glob) for m, (a, b) in zip(mods, kb)]
    stacked = torch.stack([e.int_data for e in enc]).contiguous()
    assert torch.equal(stacked.permute(1, 0, 2).reshape(M, K), qx.int_data) and all(torch.equal(e.scale.view(torch.int32), qx.scale.view(torch.int32)) for e in enc)
    ranks = range(G) if M * N * K < 1 << 32 else (0, G - 1)
    nb = _split(N, G)
    for r in ranks:
        y_r = mods[r]._local_rows_stacked(stacked, enc[r].scale, torch.bfloat16)
        assert torch.equal(y_r.view(torch.int16), y_ref[:, nb[r][0]:nb[r][1]].contiguous().view(torch.int16)), r


def _way(L, stacked, wq, M, N, K, kps, nbytes=None, lda=None, stride=None):
    lda = kps if lda is None else lda
    stride = M * kps if stride is None else stride
    need = L.pq_qlinear_kslabs_workspace_bytes_for(stacked.data_ptr(), lda, stride, kps, wq.data_ptr(), K, M, N, K)
    return L.pq_kslabs_way_name(stacked.data_ptr(), lda, stride, kps, wq.data_ptr(), K, M, N, K, need if nbytes is None else nbytes).decode(), need


@pytest.mark.parametrize("M,N,K,G,way", [(380, 484, 512, 4, "ring"), (133, 633, 256, 2, "ring"), (700, 300, 1024, 8, "ring"), (2048, 1024, 2048, 8, "ring"),      # slabs of ONE or two K-tiles (a fuzz find)
                                        (4096, 1024, 28672, 8, "fused split-K x4"), (1024, 1024, 8192, 8, "ring"), (2048, 512, 4096, 4, "ring"), (700, 1000, 1024, 2, "ring"),
                                        (256, 4096, 2048, 2, "ring"), (4096, 4096, 4096, 4, "layout"), (2048, 4096, 11008, 2, "fused split-K x2"), (2048, 4096, 11008, 4, "layout"), (16, 1024, 8192, 8, "layout"), (37, 50, 384, 3, "layout"),
                                        (300, 640, 960, 5, "layout"), (2048, 4096, 11264, 2, "fused split-K x2"), (4096, 1024, 28672, 2, "fused split-K x4"), (4096, 1024, 28672, 4, "fused split-K x4")])
def test_qlinear_on_stacked_code_blocks(pq, M, N, K, G, way):
    """pq_qlinear_s8_kslabs == pq_qlinear_s8 on the row-major codes, bit for bit: the ring tiles (128 x 128, 64 x 128, 64 x 64) walk the slabs in place (no
    workspace), the fused split-K of the 256 x 256 tile does where the planner runs it (round 6: the 70B `down` shard over 8, 4 and 2 slabs — two slabs per slice, one, half of one —
    and a cfg-3-like `down`); every other dispatch — plain 256-wide tiles, the weight-streaming kernel, the generic kernel with K / G not a multiple of 128 — takes the layout pass.
    All three output dtypes, with bias."""
    from protoquant_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    xq = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    wq = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    xs = (torch.rand(M, generator=g) * 0.01 + 1e-4).cuda()
    ws = (torch.rand(N, generator=g) * 0.01 + 1e-4).cuda()
    kps = K // G
    stacked = xq.reshape(M, G, kps).permute(1, 0, 2).contiguous()
    L_ = _lib.lib()
    name, need = _way(L_, stacked, wq, M, N, K, kps)
    assert way in name, (name, need)
    assert (need == 0) == (way == "ring") and L_.pq_qlinear_kslabs_workspace_bytes(M, N, K, kps) >= max(need, M * K)      # (the short query is always enough)
    if "split-K" in way:        # without the hand-over workspace the ring tile (or the layout pass) still gives the answer: never an error
        assert "split-K" not in _way(L_, stacked, wq, M, N, K, kps, nbytes=0)[0]
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        bias = (torch.randn(N, generator=g) * 0.1).to(dt).cuda()
        for b in (None, bias):
            want = pq.qlinear_s8(xq, xs, wq, ws, b, dt)
            got = pq.qlinear_s8_kslabs(stacked, xs, wq, ws, b, dt)
            assert torch.equal(got.view(torch.int32 if dt == torch.float32 else torch.int16), want.view(torch.int32 if dt == torch.float32 else torch.int16)), (dt, b is not None)
    if M * N * K <= 300 * 640 * 960:            # small enough for the numpy oracle
        acc = Q.gemm_s8s8s32(xq.cpu().numpy(), wq.cpu().numpy())
        same(pq.qlinear_s8_kslabs(stacked, xs, wq, ws, None, torch.bfloat16), Q.epilogue(acc, xs.cpu().numpy(), ws.cpu().numpy(), None, 0), "vs oracle")
    try:                                        # the switch: the layout pass everywhere, same bits
        _lib.set_option("PQ_NO_KSLABS", "1")
        assert _way(L_, stacked, wq, M, N, K, kps)[0] == "layout pass"
        assert torch.equal(pq.qlinear_s8_kslabs(stacked, xs, wq, ws, None, torch.bfloat16).view(torch.int16), pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16).view(torch.int16))
    finally:
        _lib.set_option("PQ_NO_KSLABS", "")


@pytest.mark.parametrize("M,N,K,G,f,pad_k,pad_m", [(300, 300, 4096, 2, 2, 0, 0), (300, 300, 4096, 4, 2, 16, 1), (130, 130, 4096, 8, 2, 0, 0), (260, 520, 4096, 2, 4, 0, 2), (513, 257, 8192, 8, 4, 128, 0),
                                                   (700, 300, 8192, 8, 8, 0, 0), (256, 256, 5120, 1, 2, 0, 0), (300, 300, 6144, 4, 2, 0, 0), (2048, 1024, 28672, 8, 4, 0, 0),
                                                   (300, 300, 3072, 8, 2, 0, 0), (300, 300, 6144, 3, 2, 0, 0), (300, 300, 7680, 4, 4, 0, 0)])
def test_forced_fused_split_k_walks_stacked_blocks_in_place(pq, pq_opt, M, N, K, G, f, pad_k, pad_m):
    """PQ_FSK=f forces f ticket slices wherever the shape admits them; on stacked blocks the asm K-loop's activation cursor jumps at the slab boundaries (kloop_p3_asm<5>):
    slices of 1, 2 and 4 slabs (slabs of 4 K-tiles — the minimum — up to 28), slabs that hold 2 and 4 slices, one slab (= pq_qlinear_s8), padded rows and padded slabs.
    Bit-identical to the contiguous form with the same forced split, to the unforced dispatch, and (small shapes) to the numpy oracle.  Shapes the in-place form does not
    admit — slabs of 3 K-tiles, a slab count that neither divides nor is divided by the slice count, slices of fewer than 5 K-tiles — take another way, same bits."""
    from protoquant_amd import _lib
    L_ = _lib.lib()
    g = torch.Generator().manual_seed(M + 7 * N + K + G)
    xq = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    xs, ws = (torch.rand(M, generator=g) * 0.01 + 1e-4).cuda(), (torch.rand(N, generator=g) * 0.01 + 1e-4).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    kps = K // G
    want = pq.qlinear_s8(xq, xs, wq, ws, bias, torch.bfloat16)               # the unforced dispatch
    pq_opt("PQ_FSK", str(f))
    forced = pq.qlinear_s8(xq, xs, wq, ws, bias, torch.bfloat16)             # f slices on the row-major codes
    buf = torch.full((G, M + pad_m, kps + pad_k), 77, dtype=torch.int8, device="cuda")
    buf[:, :M, :kps] = xq.reshape(M, G, kps).permute(1, 0, 2)
    lda, stride = kps + pad_k, (M + pad_m) * (kps + pad_k)
    name, need = _way(L_, buf, wq, M, N, K, kps, lda=lda, stride=stride)
    admits = G > 1 and kps % 128 == 0 and kps >= 512 and (G % f == 0 or f % G == 0) and K // f >= 640 and K % (128 * f) == 0 and M > 64 and M * N >= 128 * 128
    assert (f"fused split-K x{f}" in name) == admits, (name, admits)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    wsp = torch.empty((need + 256,), dtype=torch.uint8, device="cuda")
    for _ in range(2):          # (twice: the launcher zeroes its tickets itself)
        y.zero_()
        rc = L_.pq_qlinear_s8_kslabs(buf.data_ptr(), lda, stride, kps, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), bias.data_ptr(), y.data_ptr(), N, 0, M, N, K,
                                     wsp.data_ptr() if need else None, need, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, L_.pq_last_error()
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), want.view(torch.int16)) and torch.equal(forced.view(torch.int16), want.view(torch.int16))
    if M * N * K <= 300 * 300 * 6144:
        acc = Q.gemm_s8s8s32(xq.cpu().numpy(), wq.cpu().numpy())
        same(y, Q.epilogue(acc, xs.cpu().numpy(), ws.cpu().numpy(), bias.cpu().view(torch.int16).numpy().view(np.uint16), 0), "vs oracle")
    for dt in (torch.float16, torch.float32):                                # the other output types (module-level wrapper, contiguous blocks)
        stacked = xq.reshape(M, G, kps).permute(1, 0, 2).contiguous()
        got = pq.qlinear_s8_kslabs(stacked, xs, wq, ws, None, dt)
        pq_opt("PQ_FSK", "")
        ref = pq.qlinear_s8(xq, xs, wq, ws, None, dt)
        pq_opt("PQ_FSK", str(f))
        v = torch.int32 if dt == torch.float32 else torch.int16
        assert torch.equal(got.view(v), ref.view(v)), dt


@pytest.mark.parametrize("M,N,K,G,pad_k,pad_m", [(700, 300, 1024, 8, 16, 3), (4096, 1024, 8192, 8, 128, 0), (64, 96, 200, 2, 0, 0), (100, 130, 600, 3, 7, 2), (300, 4096, 2048, 2, 32, 1)])
def test_kslabs_with_padded_slabs_through_the_c_abi(pq, M, N, K, G, pad_k, pad_m):
    """the C-ABI's own generality: rows of a slab with a leading dimension wider than K / G, slabs further apart than M rows (a gather buffer with padding), widths that are
    not a multiple of 16 bytes (byte-wise layout pass + the generic GEMM) — in place where a ring tile runs, through the layout pass otherwise; always pq_qlinear_s8's bits."""
    from protoquant_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M * 3 + N + K)
    xq = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    xs, ws = (torch.rand(M, generator=g) * 0.01 + 1e-4).cuda(), (torch.rand(N, generator=g) * 0.01 + 1e-4).cuda()
    kps = K // G
    buf = torch.full((G, M + pad_m, kps + pad_k), 77, dtype=torch.int8, device="cuda")
    buf[:, :M, :kps] = xq.reshape(M, G, kps).permute(1, 0, 2)
    want = pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    lda, stride = kps + pad_k, (M + pad_m) * (kps + pad_k)
    # (the fast tiles need 16-byte aligned rows: an odd leading dimension or slab stride takes the layout pass, and the workspace query is told so through k_per_slab only —
    # a direct caller with unaligned slabs sizes the workspace for the layout pass itself)
    need = L.pq_qlinear_kslabs_workspace_bytes(M, N, K, kps)          # (the short query: always enough)
    assert need >= ((M * K + 255) // 256) * 256 + L.pq_qlinear_workspace_bytes(M, N, K)
    wsp = torch.empty((need + 256,), dtype=torch.uint8, device="cuda")
    rc = L.pq_qlinear_s8_kslabs(buf.data_ptr(), lda, stride, kps, xs.data_ptr(), wq.data_ptr(), K, ws.data_ptr(), None, y.data_ptr(), N, 0, M, N, K,
                                wsp.data_ptr(), need, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.pq_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), want.view(torch.int16))


def test_kslabs_argument_validation(pq):
    from protoquant_amd import _lib
    L = _lib.lib()
    a = torch.zeros((2, 64, 128), dtype=torch.int8, device="cuda")
    w = torch.zeros((32, 256), dtype=torch.int8, device="cuda")
    s1, s2 = torch.ones(64, device="cuda"), torch.ones(32, device="cuda")
    y = torch.empty((64, 32), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    # K not a multiple of k_per_slab; slabs that overlap; missing workspace on a shape that needs the layout pass
    assert L.pq_qlinear_s8_kslabs(a.data_ptr(), 128, 64 * 128, 100, s1.data_ptr(), w.data_ptr(), 256, s2.data_ptr(), None, y.data_ptr(), 32, 0, 64, 32, 256, None, 0, st) == 1
    assert L.pq_qlinear_s8_kslabs(a.data_ptr(), 128, 100, 128, s1.data_ptr(), w.data_ptr(), 256, s2.data_ptr(), None, y.data_ptr(), 32, 0, 64, 32, 256, None, 0, st) == 1
    need = L.pq_qlinear_kslabs_workspace_bytes(64, 32, 256, 128)
    assert need >= 64 * 256
    assert L.pq_qlinear_s8_kslabs(a.data_ptr(), 128, 64 * 128, 128, s1.data_ptr(), w.data_ptr(), 256, s2.data_ptr(), None, y.data_ptr(), 32, 0, 64, 32, 256, None, 0, st) == 5
    with pytest.raises(ValueError):
        pq.qlinear_s8_kslabs(a.permute(1, 0, 2), s1, w, s2, None, torch.bfloat16)          # not contiguous [G, M, K/G]
    with pytest.raises(ValueError):
        pq.silu_mul_quantize_with_amax(torch.zeros(4, 8, device="cuda", dtype=torch.bfloat16), torch.zeros(4, 8, device="cuda", dtype=torch.bfloat16),
                                       torch.zeros(4, device="cuda"))                        # amax must be int32 bit patterns


def _linears(H, I, seed, bias=False, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    out = []
    for (o, i) in ((I, H), (I, H), (H, I)):
        lin = torch.nn.Linear(i, o, bias=bias, device="cuda", dtype=dtype)
        with torch.no_grad():
            lin.weight.copy_((torch.randn(o, i, generator=g) * 0.03).to(dtype))
            if bias:
                lin.bias.copy_((torch.randn(o, generator=g) * 0.05).to(dtype))
        out.append(lin)
    return out


def _play_ranks(pq, lins, x, G, ranks=None):
    """every rank of a G-rank ColumnShardedGatedMLP on this one GPU: the device steps are the module's own, the three collectives are replaced by their
    definitions (integer max over the ranks, stacking of the code blocks, concatenation of the output shards).  Returns (stacked codes, scales, {rank: y shard})."""
    mods = [pq.ColumnShardedGatedMLP.from_linears(*lins, world=G, rank=r) for r in range(G)]
    x2 = x.reshape(-1, x.shape[-1])
    gu = [m._gate_up(x2) for m in mods]
    glob = torch.stack([m._local_amax(g_, u_) for m, (g_, u_) in zip(mods, gu)]).max(dim=0).values
    hq = [m._encode(g_, u_, glob) for m, (g_, u_) in zip(mods, gu)]
    stacked = torch.stack([h.int_data for h in hq]).contiguous()
    for h in hq[1:]:
        assert torch.equal(h.scale.view(torch.int32), hq[0].scale.view(torch.int32))
    ys = {r: mods[r]._down(stacked, hq[r].scale, x.dtype) for r in (ranks if ranks is not None else range(G))}
    return stacked, hq[0].scale, ys


@pytest.mark.parametrize("M,H,I,G,bias,dtype", [(300, 512, 1024, 2, True, torch.bfloat16), (129, 256, 1536, 8, False, torch.float16), (64, 384, 768, 3, True, torch.bfloat16),
                                                (1000, 1024, 4096, 4, False, torch.bfloat16)])
def test_column_sharded_gated_mlp_equals_the_unsharded_block(pq, M, H, I, G, bias, dtype):
    """G ranks played offline: the int8 codes and row scales of the intermediate, and the block output, are the unsharded GatedMLP's, bit for bit; and the
    unsharded block is the oracle chain's."""
    lins = _linears(H, I, M + H + I, bias, dtype)
    x = (torch.randn(M, H, generator=torch.Generator().manual_seed(M)) * 1.5).to(dtype).cuda()
    ref = pq.GatedMLP.from_linears(*lins)
    g, u = ref.gate_up(x)
    hq_ref = pq.silu_mul_quantize(g, u)
    y_ref = ref(x)
    stacked, scale, ys = _play_ranks(pq, lins, x, G)
    assert torch.equal(stacked.permute(1, 0, 2).reshape(M, I), hq_ref.int_data) and torch.equal(scale.view(torch.int32), hq_ref.scale.view(torch.int32))
    y = torch.cat([ys[r] for r in range(G)], dim=1)
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    # ... and the unsharded block against the oracle chain (so the sharded one is pinned to the oracle as well)
    code = 0 if dtype == torch.bfloat16 else 1
    wq = [C.quant_rowwise(bits(l.weight), code) for l in lins]
    bb = [bits(l.bias) if bias else None for l in lins]
    xq, xs = C.quant_rowwise(bits(x), code)
    sq, ss, _ = C.silu_mul_quant_rowwise(C.qlinear_s8(xq, xs, *wq[0], bb[0], code), C.qlinear_s8(xq, xs, *wq[1], bb[1], code), code)
    same(y, C.qlinear_s8(sq, ss, *wq[2], bb[2], code), "sharded block vs oracle chain")


def test_column_sharded_gated_mlp_llama70b_real_shapes(pq):
    """BASELINE config 5 at its real shapes (M = 4096 tokens, hidden 8192, intermediate 28672, 8 ranks): all eight ranks' gate/up shards and code blocks, two ranks'
    down shards (4096 x 1024 x 28672 on the stacked blocks — the ring tile walking the slabs in place) against the unsharded block."""
    M, H, I, G = 4096, 8192, 28672, 8
    lins = _linears(H, I, 70, False)
    x = torch.randn(M, H, generator=torch.Generator().manual_seed(70)).to(torch.bfloat16).cuda()
    ref = pq.GatedMLP.from_linears(*lins)
    g, u = ref.gate_up(x)
    hq_ref = pq.silu_mul_quantize(g, u)
    y_ref = ref.down(hq_ref)
    del g, u
    stacked, scale, ys = _play_ranks(pq, lins, x, G, ranks=(0, 5))
    assert torch.equal(scale.view(torch.int32), hq_ref.scale.view(torch.int32))
    assert torch.equal(stacked.permute(1, 0, 2).reshape(M, I), hq_ref.int_data)
    for r, y in ys.items():
        assert torch.equal(y.view(torch.int16), y_ref[:, r * (H // G):(r + 1) * (H // G)].contiguous().view(torch.int16)), r


def test_column_sharded_gated_mlp_world1_rccl_under_capture(pq):
    """a 1-rank RCCL communicator: forward() through libpq_rccl.so (all-reduce max, byte all-gather, output gather) eagerly and replayed from a hipGraph."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29549")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        gather = pq.RcclColumnGather()
        lins = _linears(512, 1024, 5, True)
        x = torch.randn(3, 100, 512, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
        y_ref = pq.GatedMLP.from_linears(*lins)(x)
        m = pq.ColumnShardedGatedMLP.from_linears(*lins, native=gather)
        y = m(x)
        assert y.shape == y_ref.shape and torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
        m2 = pq.ColumnShardedGatedMLP.from_linears(*lins)          # torch.distributed collectives (world 1: no-ops)
        assert torch.equal(m2(x).view(torch.int16), y_ref.view(torch.int16))
        out = torch.empty_like(y_ref)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                out.copy_(m(x))
            for _ in range(3):
                out.zero_(); gr.replay()
            torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), y_ref.view(torch.int16))
        # the same exchange in front of a plain column-sharded projection (world 1: the "block" is the whole activation)
        lin = torch.nn.Linear(512, 384, bias=True, device="cuda", dtype=torch.bfloat16)
        mq = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather)
        yq = pq.qlinear.from_linear(lin)(x)
        assert torch.equal(mq.forward_sharded_input(x).view(torch.int16), yq.view(torch.int16))
        assert torch.equal(pq.ColumnShardedQLinear.from_linear(lin).forward_sharded_input(x).view(torch.int16), yq.view(torch.int16))
        with pytest.raises(ValueError):
            mq.forward_sharded_input(x[..., :100])
        gather.close()
    finally:
        if created:
            dist.destroy_process_group()
    with pytest.raises(ValueError):
        pq.ColumnShardedGatedMLP.from_linears(*_linears(256, 1000, 3), world=3, rank=0)       # 1000 % 3 != 0
