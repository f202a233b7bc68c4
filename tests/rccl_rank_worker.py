"""Child process of tests/test_zz_gpu_real_ranks.py: ONE rank of a multi-GPU job (one process per GPU, RANK / WORLD_SIZE / MASTER_* from the
environment, gloo for the bootstrap, libpq_rccl.so for the data path).  Every rank holds the full weights, so each result is checked
locally against the unsharded qlinear — bit for bit for the gathers, and for the reduce-scatter at two ranks (a two-term f32 sum has
one order).  Prints 'OK <rank>' on success; any assertion kills the job (the parent checks every rank's output)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)       # (two ranks on ONE GPU are not an option: ncclCommInitRank refuses with "Duplicate GPU detected", tried on the pool's 1-GPU box)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import protoquant_amd as pq
    from protoquant_amd.sharded import shard_bounds
    gather = pq.RcclColumnGather()
    assert gather.comm_ranks() == world
    bits = lambda t: t.contiguous().view(torch.int16)          # noqa: E731
    for (M, N, K, bias) in ((700, 384, 256, True), (513, 391, 384, False), (4096, 4096, 1024, False)):     # equal, ragged, full-size shards
        torch.manual_seed(7 + N)                                                                           # same weights on every rank
        lin = torch.nn.Linear(K, N, bias=bias, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        for kw in ({}, {"overlap_chunks": 3}, {"layout": "transposed"}, {"layout": "transposed", "transposed_view": True}):
            m = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather, **kw)
            y1 = m(x)
            torch.cuda.synchronize()
            assert tuple(y1.shape) == (M, N) and torch.equal(bits(y0), bits(y1)), (rank, M, N, K, kw)
        yt = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather, layout="transposed").forward_t(x)
        assert yt.is_contiguous() and torch.equal(bits(yt.t()), bits(y0))
        # the whole sharded step replayed from a hipGraph: GEMM + RCCL exchange captured (what bench.py --gpus N times)
        m = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather)
        m(x); torch.cuda.synchronize()
        out = torch.empty_like(y0)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out.copy_(m(x))
            for _ in range(3):
                out.zero_(); g.replay()
            torch.cuda.synchronize()
        assert torch.equal(bits(out), bits(y0)), (rank, "graph replay", M, N, K)
        # row-sharded pairing: K split over the ranks, f32 partials, native reduce-scatter of row blocks (equal blocks only)
        if M % world:
            continue
        rs = pq.RcclRowReduceScatter(gather)
        lay = pq.RowShardedQLinear.from_linear(lin, native=rs)
        k0, k1 = shard_bounds(K, world, rank)
        yr = lay(x[:, k0:k1].contiguous())
        total = None
        for r in range(world):                       # every rank's partial rebuilt locally, summed in rank order
            a, b = shard_bounds(K, world, r)
            p = pq.RowShardedQLinear.from_linear(lin, world=world, rank=r).partial(x[:, a:b].contiguous())
            total = p if total is None else total + p
        m0, m1 = shard_bounds(M, world, rank)
        torch.cuda.synchronize()
        if world == 2:
            assert torch.equal(bits(yr), bits(total[m0:m1].to(torch.bfloat16))), (rank, "reduce-scatter", M, N, K)
        else:
            assert torch.allclose(yr.float(), total[m0:m1], rtol=2e-2, atol=1e-3)
    # the int8-code exchange of the column-sharded gated MLP: all-reduce(max) of the row-amax bit patterns, all-gather of the int8 code blocks, the down shard's GEMM on
    # the stacked blocks, gather of the output shards — against the unsharded GatedMLP on this rank, bit for bit; eagerly and replayed from a hipGraph
    for (M, H, I, bias) in ((300, 512, 1024, True), (4096, 1024, 8 * 896, False)):
        if I % world or H % world:
            continue
        torch.manual_seed(11 + I)
        lins = [torch.nn.Linear(i, o, bias=bias, device="cuda", dtype=torch.bfloat16) for (o, i) in ((I, H), (I, H), (H, I))]
        x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16)
        x[1, 5] = float("nan")                                                       # a NaN token: must come through the integer max as a NaN row
        y0 = pq.GatedMLP.from_linears(*lins)(x)
        m = pq.ColumnShardedGatedMLP.from_linears(*lins, native=gather)
        y1 = m(x)
        torch.cuda.synchronize()
        nan0 = torch.isnan(y0)
        assert torch.equal(torch.isnan(y1), nan0) and bool(nan0[1].all()) and torch.equal(bits(y1)[~nan0], bits(y0)[~nan0]), (rank, "int8-code exchange", M, H, I)
        out = torch.empty_like(y0)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out.copy_(m(x))
            for _ in range(3):
                out.zero_(); g.replay()
            torch.cuda.synchronize()
        assert torch.equal(bits(out)[~nan0], bits(y0)[~nan0]), (rank, "int8-code exchange, graph replay", M, H, I)
    # ... and in front of a plain column-sharded projection: every rank holds K / world input features (its heads of an attention output) and N / world output channels
    for (M, N, K) in ((300, 384, 1024), (4096, 1024, 2048)):
        if K % world:
            continue
        torch.manual_seed(3 + K)
        lin = torch.nn.Linear(K, N, bias=True, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        y0 = pq.qlinear.from_linear(lin)(x)
        k0, k1 = shard_bounds(K, world, rank)
        m = pq.ColumnShardedQLinear.from_linear(lin, native_gather=gather)
        y1 = m.forward_sharded_input(x[:, k0:k1])
        torch.cuda.synchronize()
        assert torch.equal(bits(y1), bits(y0)), (rank, "sharded input", M, N, K)
    # a whole (small) Llama sharded over the ranks with the native exchange (llama.shard_llama_layers): logits against the unsharded int8 model on this rank
    try:
        import transformers as tr
    except ImportError:
        tr = None
    if tr is not None:
        import copy

        from protoquant_amd.llama import fuse_llama_layers, shard_llama_layers
        torch.manual_seed(0)
        cfg = tr.LlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=8,
                             max_position_embeddings=256, attn_implementation="eager")
        base = tr.LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
        ref = copy.deepcopy(base).cuda()
        pq.swap_linears(ref, fuse_gated_mlp=True)
        fuse_llama_layers(ref)
        sh = copy.deepcopy(base)
        assert shard_llama_layers(sh, native=gather) == 2
        sh = sh.cuda()
        ids = torch.randint(0, 512, (2, 64), generator=torch.Generator().manual_seed(5)).cuda()
        with torch.no_grad():
            a, b = ref(ids).logits, sh(ids).logits
        torch.cuda.synchronize()
        diff = float((a.float() - b.float()).abs().max())
        assert a.shape == b.shape and diff <= 0.02 * float(a.float().abs().max()), (rank, "sharded llama", diff)
    dist.barrier()
    gather.close()
    dist.destroy_process_group()
    print(f"OK {rank}", flush=True)


if __name__ == "__main__":
    main()
