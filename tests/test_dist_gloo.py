"""CPU, world_size 2 over gloo: the N>1 host logic (shard bounds, the one collective, the layout fix).
No HIP calls: shards are synthetic tensors whose value encodes (row, global column)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, M, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from protoquant_amd.sharded import gather_columns, shard_bounds
        lo, hi = shard_bounds(n_total, world, rank)
        rows = torch.arange(M, dtype=torch.float32)[:, None]
        cols = torch.arange(lo, hi, dtype=torch.float32)[None, :]
        y_local = (rows * 10000 + cols).to(torch.bfloat16 if n_total < 200 else torch.float32)
        y = gather_columns(y_local, n_total)
        want = (rows * 10000 + torch.arange(n_total, dtype=torch.float32)[None, :]).to(y_local.dtype)
        ok = y.shape == (M, n_total) and torch.equal(y, want)
        st = gather_columns(y_local, n_total, stacked=True)
        ok = ok and st.shape[0] == world and torch.equal(st[rank, :, : hi - lo], y_local)
        # weak-scaling bench reduction: max over ranks of a per-rank time
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,M", [(8, 4), (7, 3), (4096, 16)])
def test_gather_columns_world2(n_total, M):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_total, M, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_bounds_cover_exactly():
    from protoquant_amd.sharded import shard_bounds
    for n in (1, 7, 8, 4096, 128256, 28672):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _rr_worker(rank, world, port, M, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from protoquant_amd.sharded import reduce_rows, shard_bounds
        g = torch.Generator().manual_seed(100 + M + N)
        parts = [torch.randn(M, N, generator=g) * (r + 1) for r in range(world)]     # every rank can rebuild all partials
        total = parts[0] + parts[1]
        lo, hi = shard_bounds(M, world, rank)
        ok = True
        for dt in (torch.float32, torch.bfloat16):
            mine = reduce_rows(parts[rank].clone(), dt, scatter=True)
            ok = ok and mine.shape == (hi - lo, N) and torch.equal(mine, total[lo:hi].to(dt))
            full = reduce_rows(parts[rank].clone(), dt, scatter=False)
            ok = ok and full.shape == (M, N) and torch.equal(full, total.to(dt))
        try:
            reduce_rows(parts[rank].to(torch.bfloat16), torch.bfloat16)
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,N", [(8, 16), (7, 5), (1, 3), (256, 64)])
def test_reduce_rows_world2(M, N):
    """Row-sharded qlinear's one collective: f32 partial outputs summed over ranks, scattered by balanced row blocks
    (ragged M padded for the collective) or all-reduced; cast once."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rr_worker, args=(r, world, port, M, N, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _ov_worker(rank, world, port, M, n_total, chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from protoquant_amd.sharded import gather_columns_overlapped, shard_bounds
        lo, hi = shard_bounds(n_total, world, rank)
        full = torch.arange(M, dtype=torch.float32)[:, None] * 10000 + torch.arange(n_total, dtype=torch.float32)[None, :]
        calls = []

        def rows(m0, m1):
            calls.append((m0, m1))
            return full[m0:m1, lo:hi].clone()
        y = gather_columns_overlapped(rows, M, n_total, chunks, torch.float32, torch.device("cpu"))
        ok = torch.equal(y, full) and sum(b - a for a, b in calls) == M and all(b > a for a, b in calls)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,n_total,chunks", [(16, 8, 4), (7, 7, 3), (5, 6, 8), (64, 4096, 2)])
def test_gather_columns_overlapped_world2(M, n_total, chunks):
    """Row-chunked, asynchronously issued all-gathers (ragged shards and ragged row blocks, more chunks than rows)
    rebuild exactly the matrix the one-shot gather does."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ov_worker, args=(r, world, port, M, n_total, chunks, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


# ---------------------------------------------------------------- ColumnShardedQLinear.forward at world 2 (host logic of the module)
def _module_worker(rank, world, port, N, M, K, layout, chunks, q):
    """The module's forward with its three device steps replaced by CPU computations from the numpy oracle (the HIP kernels
    are covered by the -m gpu tests): what runs here is the real forward() — shard bounds, the choice of exchange, the
    collectives over gloo, the layout handling — and the result must equal the oracle's unsharded qlinear, bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from oracle import qspec_numpy as Q
        from protoquant_amd.qtensor import QTensor
        from protoquant_amd.sharded import ColumnShardedQLinear, shard_bounds

        rng = np.random.default_rng(5)
        x = Q.from_f32(rng.standard_normal((M, K)).astype(np.float32), 0)
        w = Q.from_f32((rng.standard_normal((N, K)) * 0.05).astype(np.float32), 0)
        bias = Q.from_f32((rng.standard_normal(N) * 0.1).astype(np.float32), 0)
        wq, ws = Q.quantize(w, 0, 1)
        y_want, xq, xs, _ = Q.qlinear(x, 0, wq, ws, bias)
        lo, hi = shard_bounds(N, world, rank)
        bf = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)     # noqa: E731

        import types
        local = types.SimpleNamespace(in_features=K, out_features=hi - lo, wq=torch.from_numpy(wq[lo:hi]),      # the rank's shard
                                      ws=torch.from_numpy(ws[lo:hi]), bias=bf(bias[lo:hi]))

        class Stub(ColumnShardedQLinear):
            def _quantize(self, xt):
                return QTensor(torch.from_numpy(xq), torch.from_numpy(xs), 1, torch.bfloat16, torch.Size((M, K)))

            def _ref(self, codes, scales):
                acc = Q.gemm_s8s8s32(codes.numpy(), wq[lo:hi])
                return Q.epilogue(acc, scales.numpy(), ws[lo:hi], bias[lo:hi], 0)

            def _local_rows(self, codes, scales, dtype, out=None):
                y = bf(self._ref(codes, scales))
                if out is not None:
                    out.copy_(y)
                    return out
                return y

            def _local_t(self, codes, scales, dtype):
                return bf(self._ref(codes, scales)).t().contiguous()

        m = Stub.__new__(Stub)
        torch.nn.Module.__init__(m)
        m.local, m.out_features, m.group, m.in_features, m.native_gather, m.overlap_chunks, m.layout = local, N, None, K, None, chunks, layout
        m.transposed_view = False
        y = m(bf(x))
        ok = tuple(y.shape) == (M, N) and torch.equal(y.contiguous().view(torch.int16), bf(y_want).view(torch.int16))
        if layout == "transposed":
            ok = ok and y.is_contiguous()               # drop-in: stock consumers .view() the result
            yt = m.forward_t(bf(x))                     # the contiguous y^T itself: no layout pass, no copy
            ok = ok and tuple(yt.shape) == (N, M) and yt.is_contiguous() and torch.equal(yt.t().contiguous().view(torch.int16), bf(y_want).view(torch.int16))
            m.transposed_view = True
            ok = ok and m(bf(x)).stride() == (1, M)     # opt-in: a view of y^T
            m.transposed_view = False
        y3 = m(bf(x).reshape(2, M // 2, K))             # [..., K] inputs
        ok = ok and tuple(y3.shape) == (2, M // 2, N) and torch.equal(y3.reshape(M, N).contiguous().view(torch.int16), bf(y_want).view(torch.int16))
        q.put((rank, bool(ok)))
    except Exception as e:                              # fail fast instead of letting the parent wait for its timeout
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,layout,chunks", [(64, "rows", 1), (63, "rows", 1), (64, "rows", 3), (63, "rows", 2), (64, "transposed", 1), (63, "transposed", 1)])
def test_column_sharded_module_forward_world2(N, layout, chunks):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_module_worker, args=(r, world, port, N, 12, 40, layout, chunks, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


# ---------------------------------------------------------------- ColumnShardedGatedMLP.forward at world 2 (host logic + the three collectives over gloo)
def _cgm_worker(rank, world, port, M, H, I, q):
    """forward() of the int8-code exchange with its four device steps replaced by the numpy oracle (the HIP kernels: tests/test_gpu_int8_exchange.py): what runs here
    is the real host logic — shard bounds, the integer all-reduce(max) of the amax bit patterns, the all-gather of the int8 blocks into the stacked layout, the gather
    of the output shards — and the result must equal the oracle's UNSHARDED block (quantize(silu(g)*u) on whole rows, one GEMM over the whole K), bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from oracle import qspec_numpy as Q
        from protoquant_amd.qtensor import QTensor
        from protoquant_amd.sharded import ColumnShardedGatedMLP, shard_bounds

        rng = np.random.default_rng(11)
        f = lambda *shape, s=1.0: Q.from_f32((rng.standard_normal(shape) * s).astype(np.float32), 0)      # noqa: E731
        x, wg, wu, wd = f(M, H), f(I, H, s=0.08), f(I, H, s=0.08), f(H, I, s=0.05)
        x[1, 3] = 0x7FC0                                    # a NaN token: its amax must propagate through the integer max
        (gq, gs), (uq, us), (dq, ds) = (Q.quantize(w, 0, 1) for w in (wg, wu, wd))
        g_full = Q.qlinear(x, 0, gq, gs)[0]
        u_full = Q.qlinear(x, 0, uq, us)[0]
        hq_want, hs_want, _ = Q.silu_mul_quantize(g_full, u_full, 0)
        y_want = Q.epilogue(Q.gemm_s8s8s32(hq_want, dq), hs_want, ds, None, 0)
        ilo, ihi = shard_bounds(I, world, rank)
        hlo, hhi = shard_bounds(H, world, rank)
        bf = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)     # noqa: E731
        nb = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)                          # noqa: E731

        class Stub(ColumnShardedGatedMLP):
            def _gate_up(self, xt):
                return bf(g_full[:, ilo:ihi]), bf(u_full[:, ilo:ihi])

            def _local_amax(self, g, u):
                return torch.from_numpy(Q.row_amax_bits(Q.silu_mul(nb(g), nb(u), 0), 0).view(np.int32).copy())

            def _encode(self, g, u, amax_bits):
                qq, ss = Q.quantize_rows_with_amax(Q.silu_mul(nb(g), nb(u), 0), 0, amax_bits.numpy().view(np.uint32))
                return QTensor(torch.from_numpy(qq), torch.from_numpy(ss), 1, torch.bfloat16, torch.Size(qq.shape))

            def _down(self, stacked, scale, dtype):
                G, Mm, kps = stacked.shape
                codes = stacked.permute(1, 0, 2).reshape(Mm, G * kps).numpy()            # what the slab walk computes: the row-major product
                return bf(Q.epilogue(Q.gemm_s8s8s32(codes, dq[hlo:hhi]), scale.numpy(), ds[hlo:hhi], None, 0))

        m = Stub.__new__(Stub)
        torch.nn.Module.__init__(m)
        m.gate_up = m.down = None
        m.hidden, m.intermediate, m.group, m.native, m.world, m.rank = H, I, None, None, world, rank
        y = m(bf(x))
        stacked, scale = m.hidden_codes(bf(x))
        ok = tuple(y.shape) == (M, H) and torch.equal(y.contiguous().view(torch.int16), bf(y_want).view(torch.int16))
        ok = ok and tuple(stacked.shape) == (world, M, I // world) and np.array_equal(stacked.permute(1, 0, 2).reshape(M, I).numpy(), hq_want)
        nan = np.isnan(hs_want)
        ok = ok and np.array_equal(np.isnan(scale.numpy()), nan) and np.array_equal(scale.numpy()[~nan].view(np.uint32), hs_want[~nan].view(np.uint32)) and bool(nan[1])
        y3 = m(bf(x).reshape(2, M // 2, H))
        ok = ok and tuple(y3.shape) == (2, M // 2, H)
        q.put((rank, bool(ok)))
    except Exception as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,H,I", [(12, 64, 96), (6, 40, 128)])
def test_column_sharded_gated_mlp_forward_world2(M, H, I):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_cgm_worker, args=(r, world, port, M, H, I, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _csi_worker(rank, world, port, M, N, K, q):
    """ColumnShardedQLinear.forward_sharded_input over gloo with the device steps replaced by the numpy oracle: the exchange (integer all-reduce(max), all-gather of the
    int8 blocks into the stacked layout, gather of the output shards) must reproduce the oracle's unsharded qlinear on the concatenated activation, bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types

        import numpy as np
        from oracle import qspec_numpy as Q
        from protoquant_amd.qtensor import QTensor
        from protoquant_amd.sharded import ColumnShardedQLinear, shard_bounds

        rng = np.random.default_rng(21)
        x = Q.from_f32(rng.standard_normal((M, K)).astype(np.float32), 0)
        x[2, K - 1] = 0x7F80                                     # +Inf in the LAST rank's block: its row amax must reach every rank
        w = Q.from_f32((rng.standard_normal((N, K)) * 0.05).astype(np.float32), 0)
        wq, ws = Q.quantize(w, 0, 1)
        y_want, xq_want, xs_want, _ = Q.qlinear(x, 0, wq, ws, None)
        lo, hi = shard_bounds(N, world, rank)
        k0, k1 = shard_bounds(K, world, rank)
        bf = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)     # noqa: E731
        nb = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)                          # noqa: E731

        class Stub(ColumnShardedQLinear):
            def _local_amax(self, x2):
                return torch.from_numpy(Q.row_amax_bits(nb(x2), 0).view(np.int32).copy())

            def _encode(self, x2, amax_bits):
                qq, ss = Q.quantize_rows_with_amax(nb(x2), 0, amax_bits.numpy().view(np.uint32))
                return QTensor(torch.from_numpy(qq), torch.from_numpy(ss), 1, torch.bfloat16, torch.Size(qq.shape))

            def _local_rows_stacked(self, stacked, scales, dtype):
                G, Mm, kps = stacked.shape
                codes = stacked.permute(1, 0, 2).reshape(Mm, G * kps).numpy()
                return bf(Q.epilogue(Q.gemm_s8s8s32(codes, wq[lo:hi]), scales.numpy(), ws[lo:hi], None, 0))

        m = Stub.__new__(Stub)
        torch.nn.Module.__init__(m)
        m.local = types.SimpleNamespace(in_features=K, out_features=hi - lo)
        m.out_features, m.group, m.in_features, m.native_gather, m.overlap_chunks, m.layout, m.transposed_view = N, None, K, None, 1, "rows", False
        y = m.forward_sharded_input(bf(x[:, k0:k1]))
        nan = np.isnan(Q.to_f32(y_want, 0))
        got = nb(y)
        ok = tuple(y.shape) == (M, N) and np.array_equal(np.isnan(y.float().numpy()), nan) and np.array_equal(got[~nan], y_want[~nan]) and bool(nan[2].all())
        y3 = m.forward_sharded_input(bf(x[:, k0:k1]).reshape(2, M // 2, k1 - k0))
        ok = ok and tuple(y3.shape) == (2, M // 2, N)
        q.put((rank, bool(ok)))
    except Exception as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,N,K", [(12, 64, 96), (6, 63, 128)])
def test_column_sharded_qlinear_sharded_input_world2(M, N, K):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_csi_worker, args=(r, world, port, M, N, K, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
