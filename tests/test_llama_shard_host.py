"""CPU (-m "not gpu"): the host logic of llama.shard_llama_layers — which rows of which projection every rank takes — on a small transformers Llama whose linears are
already int8 (qlinear modules built from random codes on the CPU: slicing needs no kernel).  Over all ranks the shards must tile every weight exactly once: q by heads,
k / v by KV heads, o / down / lm_head by output channels, gate / up by intermediate channels (the same block for both), scales and biases with their rows."""
import copy

import pytest
import torch

tr = pytest.importorskip("transformers")


def _int8_model():
    from protoquant_amd.qlinear import qlinear
    from protoquant_amd.qtensor import QTensor
    torch.manual_seed(0)
    cfg = tr.LlamaConfig(vocab_size=96, hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=64)
    model = tr.LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    g = torch.Generator().manual_seed(1)

    def fake(lin):
        n, k = lin.out_features, lin.in_features
        wq = torch.randint(-127, 128, (n, k), generator=g, dtype=torch.int8)
        return qlinear.from_qtensor(QTensor(wq, torch.rand(n, generator=g) + 0.1, 1, torch.bfloat16, wq.shape), None)
    for layer in model.model.layers:
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(layer.self_attn, name, fake(getattr(layer.self_attn, name)))
        for name in ("gate_proj", "up_proj", "down_proj"):
            setattr(layer.mlp, name, fake(getattr(layer.mlp, name)))
    model.lm_head = fake(model.lm_head)
    return model


@pytest.mark.parametrize("world", [1, 2, 4])
def test_shards_tile_every_weight_exactly_once(world):
    from protoquant_amd.llama import RMSNormQuant, shard_llama_layers
    from protoquant_amd.sharded import ColumnShardedGatedMLP, ColumnShardedQLinear
    full = _int8_model()
    shards = []
    for r in range(world):
        m = copy.deepcopy(full)
        assert shard_llama_layers(m, world=world, rank=r, device="cpu") == 2
        shards.append(m)
    hd = 64 // 8
    for li in range(2):
        fa, fm = full.model.layers[li].self_attn, full.model.layers[li].mlp
        parts = [s.model.layers[li] for s in shards]
        assert all(isinstance(p.mlp, ColumnShardedGatedMLP) and isinstance(p.input_layernorm, RMSNormQuant) and isinstance(p.post_attention_layernorm, RMSNormQuant) for p in parts)
        # fused local q | k | v: the rank's heads, then its KV heads (twice)
        nq, nkv = 8 // world * hd, 4 // world * hd
        for r, p in enumerate(parts):
            f = p.self_attn.qkv_fused.fused
            assert f.splits == [nq, nkv, nkv] and f.in_features == 64
            want = torch.cat([fa.q_proj.wq[r * nq:(r + 1) * nq], fa.k_proj.wq[r * nkv:(r + 1) * nkv], fa.v_proj.wq[r * nkv:(r + 1) * nkv]])
            assert torch.equal(f.wq, want)
            assert torch.equal(f.ws, torch.cat([fa.q_proj.ws[r * nq:(r + 1) * nq], fa.k_proj.ws[r * nkv:(r + 1) * nkv], fa.v_proj.ws[r * nkv:(r + 1) * nkv]]))
        # o, down: output channels; gate / up: the same intermediate block for both
        assert torch.equal(torch.cat([p.self_attn.o_proj.sharded.local.wq for p in parts]), fa.o_proj.wq)
        assert all(isinstance(p.self_attn.o_proj.sharded, ColumnShardedQLinear) and p.self_attn.o_proj.sharded.out_features == 64 for p in parts)
        assert torch.equal(torch.cat([p.mlp.down.wq for p in parts]), fm.down_proj.wq) and torch.equal(torch.cat([p.mlp.down.ws for p in parts]), fm.down_proj.ws)
        ib = 160 // world
        for r, p in enumerate(parts):
            assert p.mlp.gate_up.splits == [ib, ib] and (p.mlp.world, p.mlp.rank, p.mlp.hidden, p.mlp.intermediate) == (world, r, 64, 160)
            assert torch.equal(p.mlp.gate_up.wq, torch.cat([fm.gate_proj.wq[r * ib:(r + 1) * ib], fm.up_proj.wq[r * ib:(r + 1) * ib]]))
    assert torch.equal(torch.cat([s.lm_head.local.wq for s in shards]), full.lm_head.wq) and all(s.lm_head.out_features == 96 for s in shards)


def test_refuses_splits_that_cut_a_head():
    from protoquant_amd.llama import shard_llama_layers
    with pytest.raises(ValueError):
        shard_llama_layers(_int8_model(), world=8, rank=0, device="cpu")        # 4 KV heads over 8 ranks
    with pytest.raises(ValueError):
        shard_llama_layers(_int8_model(), world=2, device="cpu")                # world without rank
