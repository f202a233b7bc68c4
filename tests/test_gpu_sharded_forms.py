"""-m gpu: the transposed-output qlinear (pq_qlinear_s8_t) and the native exchange forms of the column-sharded configuration
(ragged shards, row-chunked overlapped, transposed) — against the C / numpy oracle and a 1-rank RCCL communicator."""
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


@pytest.mark.parametrize("variant", ["auto", "generic", "sp256_16", "sp128_16", "sp128x128", "ring128", "skinny"])
@pytest.mark.parametrize("M,N,K,code,bias", [(256, 512, 256, 0, True), (300, 130, 384, 0, True), (64, 1000, 128, 1, True), (17, 33, 65, 2, True),
                                              (512, 256, 1024, 2, False), (48, 1024, 512, 0, False), (700, 260, 256, 0, True)])
def test_transposed_qlinear_vs_oracle(pq, pq_opt, variant, M, N, K, code, bias):
    """yt[n][m] of pq_qlinear_s8_t == the oracle's y[m][n], bit for bit (token scale first although the tokens are the GEMM's
    columns; bias along rows) through every kernel variant, staged and direct epilogues, all output dtypes."""
    pq_opt("PQ_FORCE_VARIANT", "" if variant == "auto" else variant)
    rng = np.random.default_rng(M * 31 + N * 7 + K + code)
    xq = rng.integers(-127, 128, (M, K), dtype=np.int8); wq = rng.integers(-127, 128, (N, K), dtype=np.int8)
    xs = (rng.random(M) * 0.02 + 1e-3).astype(np.float32); ws = (rng.random(N) * 0.02 + 1e-3).astype(np.float32)
    b = Q.from_f32((rng.standard_normal(N) * 0.5).astype(np.float32), code) if bias else None
    want = C.qlinear_s8(xq, xs, wq, ws, b, code)
    yt = pq.qlinear_s8_t(torch.from_numpy(xq).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(wq).cuda(), torch.from_numpy(ws).cuda(),
                         to_gpu(b, code) if bias else None, TD[code])
    assert tuple(yt.shape) == (N, M)
    same(yt.t().contiguous(), want, f"yt {M}x{N}x{K} {variant}")


def test_transposed_qlinear_full_size_and_splitk(pq, pq_opt):
    """The 70B q/o shard at its real size through the transposed form (N_local = 1024 rows x 4096 token columns), and a
    quarter-filled long-K problem with and without the split-K slabs: identical to the row-major result."""
    for (M, N, K) in ((4096, 1024, 8192), (512, 4096, 8192)):
        g = torch.Generator(device="cuda").manual_seed(M + N)
        xq = (torch.randn(M, K, device="cuda", generator=g) * 28).round().clamp(-127, 127).to(torch.int8)
        wq = (torch.randn(N, K, device="cuda", generator=g) * 28).round().clamp(-127, 127).to(torch.int8)
        xs = torch.rand(M, device="cuda", generator=g) * 1e-2 + 1e-4; ws = torch.rand(N, device="cuda", generator=g) * 1e-2 + 1e-4
        bias = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
        y = pq.qlinear_s8(xq, xs, wq, ws, bias, torch.bfloat16)
        yt = pq.qlinear_s8_t(xq, xs, wq, ws, bias, torch.bfloat16)
        assert torch.equal(yt.t().contiguous().view(torch.int16), y.view(torch.int16))
        pq_opt("PQ_NO_SPLITK", "1")
        yt2 = pq.qlinear_s8_t(xq, xs, wq, ws, bias, torch.bfloat16)
        pq_opt("PQ_NO_SPLITK", "")
        assert torch.equal(yt2.view(torch.int16), yt.view(torch.int16))


def test_ragged_layout_kernel(pq):
    """pq_unstack_cols_v on synthetic multi-rank stacked buffers with ragged shard widths and a padded output."""
    from protoquant_amd import _rccl
    from protoquant_amd.sharded import shard_bounds
    R = _rccl.lib()
    for dt, code in ((torch.bfloat16, 0), (torch.float32, 2)):
        for (G, M, n_total, pad) in ((3, 37, 70, 0), (8, 5, 61, 3), (2, 64, 4097, 0), (8, 16, 128256 // 8 + 5, 8)):
            n_max = -(-n_total // G)
            st = torch.randn(G, M, n_max, device="cuda").to(dt)
            out = torch.zeros((M, n_total + pad), dtype=dt, device="cuda")
            _rccl.check(R.pq_unstack_cols_v(st.data_ptr(), out.data_ptr(), n_total + pad, G, M, n_total, code, torch.cuda.current_stream().cuda_stream), "unstack_v")
            want = torch.cat([st[r, :, : shard_bounds(n_total, G, r)[1] - shard_bounds(n_total, G, r)[0]] for r in range(G)], dim=1)
            assert torch.equal(out[:, :n_total], want)
            assert pad == 0 or bool((out[:, n_total:] == 0).all())


def test_native_gather_forms_world1(pq):
    """A real 1-rank RCCL communicator behind every exchange form: whole (contiguous and STRIDED shards: the pack kernel and the
    strided layout kernel), row-chunked on the side stream, transposed — each bit-identical to the plain qlinear — and RCCL's
    own rank count."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        gather = pq.RcclColumnGather()
        assert gather.comm_ranks() == 1
        torch.manual_seed(4)
        lin = torch.nn.Linear(256, 384, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(700, 256, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        for kw in ({}, {"overlap_chunks": 3}, {"layout": "transposed"}):
            m = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather, **kw)
            y1 = m(x)
            torch.cuda.synchronize()
            assert tuple(y1.shape) == (700, 384) and torch.equal(y0.view(torch.int16), y1.contiguous().view(torch.int16)), kw
            y3 = m(x.reshape(7, 100, 256))
            assert tuple(y3.shape) == (7, 100, 384) and torch.equal(y3.reshape(700, 384).contiguous().view(torch.int16), y0.view(torch.int16)), kw
        # strided shard (a column slice of a wider buffer) into a strided destination
        wide = torch.zeros((700, 400), dtype=torch.bfloat16, device="cuda"); wide[:, 8:392] = y0
        dst = torch.zeros((700, 512), dtype=torch.bfloat16, device="cuda")
        gather.gather_into(wide[:, 8:392], dst[:, :384], 384)
        torch.cuda.synchronize()
        assert torch.equal(dst[:, :384], y0) and bool((dst[:, 384:] == 0).all())
        # the torch.distributed forms of the same module
        for kw in ({}, {"overlap_chunks": 2}, {"layout": "transposed"}):
            y2 = pq.ColumnShardedQLinear.from_linear(lin, **kw)(x)
            assert torch.equal(y0.view(torch.int16), y2.contiguous().view(torch.int16)), kw
        gather.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_operand_validation(pq):
    """CPU or mistyped operands must raise, not hand host pointers to the GPU."""
    from protoquant_amd._lib import PQError
    m = pq.qlinear(64, 32)                                   # buffers on the CPU
    x = torch.randn(4, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(PQError):
        m(x)
    xq = torch.zeros((4, 64), dtype=torch.int8, device="cuda"); wq = torch.zeros((32, 64), dtype=torch.int8, device="cuda")
    xs = torch.ones(4, device="cuda"); ws = torch.ones(32, device="cuda")
    with pytest.raises(TypeError):
        pq.qlinear_s8(xq, xs.double(), wq, ws, None, torch.bfloat16)
    with pytest.raises(ValueError):
        pq.qlinear_s8(xq, xs, wq, torch.ones(64, device="cuda")[::2], None, torch.bfloat16)
    with pytest.raises(ValueError):
        pq.qlinear_s8(xq, xs, wq, ws, None, torch.bfloat16, out=torch.empty((4, 31), dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(PQError):
        pq.qlinear_s8(xq, xs, wq, ws.cpu(), None, torch.bfloat16)


def test_workspace_is_bounded(pq):
    """rising M must not pin every outgrown workspace (geometric growth, outgrown buffers dropped)."""
    import sys
    QL = sys.modules["protoquant_amd.qlinear"]
    pq.clear_workspaces()
    lin = pq.qlinear.from_linear(torch.nn.Linear(512, 256, bias=False, device="cuda", dtype=torch.bfloat16))
    sizes = set()
    for M in range(1, 600, 7):
        lin(torch.randn(M, 512, device="cuda", dtype=torch.bfloat16))
        sizes.add(next(iter(QL._WORKSPACES.values()))[0].numel())
    assert len(QL._RETIRED) == 0 and len(QL._WORKSPACES) == 1 and len(sizes) <= 12


def test_workspace_used_by_a_graph_survives_eager_growth(pq):
    """a workspace allocated EAGERLY (the warm-up) and then used inside a hipGraph captured on the same stream is pinned by
    that graph: a later eager call that outgrows it must park it, not free it — otherwise the caching allocator hands the block to other
    tensors and a replay writes its scratch (here: the one-call path's codes and scales) into foreign memory.
    Sequence: warm up, capture on the same stream, grow eagerly, allocate over the freed space, replay, compare bits."""
    import sys
    QL = sys.modules["protoquant_amd.qlinear"]
    pq.clear_workspaces()
    torch.manual_seed(3)
    lin = pq.qlinear.from_linear(torch.nn.Linear(512, 256, bias=False, device="cuda", dtype=torch.bfloat16))
    x = torch.randn(64, 512, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y_ref = lin(x).clone()                      # eager warm-up on stream s: allocates the workspace (not capturing)
        g = torch.cuda.CUDAGraph()
        out = torch.empty_like(y_ref)
        with torch.cuda.graph(g, stream=s):         # the capture re-uses that buffer without growing it
            out.copy_(lin(x))
        ent = QL._WORKSPACES[("cuda", x.device.index, s.cuda_stream)]
        assert ent[1], "a buffer handed out under capture must be marked graph-pinned"
        old_ptr = ent[0].data_ptr()
        lin(torch.randn(4096, 512, device="cuda", dtype=torch.bfloat16))      # eager, larger M: outgrows the pinned buffer
        assert any(b.data_ptr() == old_ptr for b in QL._RETIRED), "the outgrown graph-pinned buffer was dropped"
        junk = [torch.full((ent[0].numel(),), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(8)]   # would land on a freed block
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), y_ref.view(torch.int16))
        assert all(bool((j == 0x5A).all()) for j in junk), "the replay wrote into memory it no longer owned"
    torch.cuda.current_stream().wait_stream(s)
    pq.clear_workspaces()
