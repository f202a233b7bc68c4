"""-m gpu: the serialised int8 weight format (protoquant_amd/serialize.py): bf16 checkpoint -> converted
state_dict -> freshly constructed int8 modules, checked against the C oracle bit for bit."""
import os

import numpy as np
import pytest
import torch
from torch import nn

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


class MLP(nn.Module):
    def __init__(self, H, I, bias=False):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = nn.Linear(H, I, bias=bias), nn.Linear(H, I, bias=bias), nn.Linear(I, H, bias=bias)
        self.act_fn = nn.SiLU()

    def forward(self, x):
        return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))


class Attn(nn.Module):
    def __init__(self, H, KV):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = nn.Linear(H, H, bias=True), nn.Linear(H, KV, bias=True), nn.Linear(H, KV, bias=True), nn.Linear(H, H, bias=False)

    def forward(self, x):                         # (not attention: only the projections matter here)
        q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        return self.o_proj(q + torch.cat([k, v] * (q.shape[-1] // (2 * k.shape[-1])), dim=-1))


class Block(nn.Module):
    def __init__(self, H, I, KV):
        super().__init__()
        self.input_layernorm = nn.LayerNorm(H)
        self.self_attn, self.mlp = Attn(H, KV), MLP(H, I)

    def forward(self, x):
        x = x + self.self_attn(self.input_layernorm(x))
        return x + self.mlp(x)


class Tiny(nn.Module):
    def __init__(self, H=256, I=640, KV=64, L=2, V=300):
        super().__init__()
        self.embed_tokens = nn.Embedding(V, H)
        self.layers = nn.ModuleList([Block(H, I, KV) for _ in range(L)])
        self.lm_head = nn.Linear(H, V, bias=False)

    def forward(self, ids):
        x = self.embed_tokens(ids)
        for l in self.layers:
            x = l(x)
        return self.lm_head(x)


def test_convert_load_matches_swap_and_oracle(pq, tmp_path):
    from protoquant_amd import serialize as S
    torch.manual_seed(11)
    ref = Tiny().to(torch.bfloat16)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    conv = S.convert_checkpoint(sd, gated_mlp=[r"re:layers\.\d+\.mlp"])
    # every linear's codes and scales == the C oracle's quantisation of the checkpoint weight
    for k, w in sd.items():
        if k.endswith("o_proj.weight") or k.endswith("lm_head.weight") or k.endswith("q_proj.weight"):
            p = k[: -len(".weight")]
            wq, ws = C.quant_rowwise(bits(w), 0)
            same(conv[p + ".wq"], wq, p + ".wq"); same(conv[p + ".ws"], ws, p + ".ws")
    g, u = sd["layers.1.mlp.gate_proj.weight"], sd["layers.1.mlp.up_proj.weight"]
    wq = np.concatenate([C.quant_rowwise(bits(g), 0)[0], C.quant_rowwise(bits(u), 0)[0]])
    same(conv["layers.1.mlp.gate_up.wq"], wq, "gate_up.wq")
    assert "layers.0.mlp.gate_proj.weight" not in conv and "embed_tokens.weight" in conv and "layers.0.input_layernorm.weight" in conv
    assert conv["layers.0.self_attn.q_proj.bias"].dtype == torch.bfloat16
    # file round trip
    path = os.path.join(tmp_path, "tiny.int8.safetensors")
    S.save_quantized(conv, path)
    back = S.load_quantized(path)
    assert set(back) == set(conv) and all(torch.equal(back[k], conv[k]) for k in conv)
    # a freshly constructed model (meta device: no float weights ever exist) receives it
    with torch.device("meta"):
        fresh = Tiny().to(torch.bfloat16)
    S.prepare_for_int8(fresh, fuse_gated_mlp=True)
    fresh = fresh.to_empty(device="cpu") if any(p.device.type == "meta" for p in fresh.parameters()) else fresh
    missing, unexpected = fresh.load_state_dict(back, strict=True)
    assert not missing and not unexpected
    fresh = fresh.cuda()
    swapped = pq.swap_linears(ref.cuda(), fuse_gated_mlp=True)
    ids = torch.randint(0, 300, (3, 50), device="cuda")
    y0, y1 = swapped(ids), fresh(ids)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    # one projection end to end against the oracle pipeline
    x = torch.randn(70, 256, device="cuda", dtype=torch.bfloat16)
    wq, ws = C.quant_rowwise(bits(sd["layers.0.self_attn.q_proj.weight"]), 0)
    want, *_ = Q.qlinear(bits(x), 0, wq, ws, bits(sd["layers.0.self_attn.q_proj.bias"]))
    same(fresh.layers[0].self_attn.q_proj(x), want, "loaded q_proj vs oracle")
    # the gated MLP against the oracle chain: fused gate+up GEMM, silu*mul fused into the quantisation, down
    gq, gs = C.quant_rowwise(bits(sd["layers.0.mlp.gate_proj.weight"]), 0); uq, us = C.quant_rowwise(bits(sd["layers.0.mlp.up_proj.weight"]), 0)
    dq, ds = C.quant_rowwise(bits(sd["layers.0.mlp.down_proj.weight"]), 0)
    gt, xq, xs, _ = Q.qlinear(bits(x), 0, gq, gs, None); ut, *_ = Q.qlinear(bits(x), 0, uq, us, None)
    hq, hs, _h = C.silu_mul_quant_rowwise(gt, ut, 0)
    want = C.qlinear_s8(hq, hs, dq, ds, None, 0)
    same(fresh.layers[0].mlp(x), want, "loaded GatedMLP vs oracle chain")


def test_fused_qkv_entry(pq):
    from protoquant_amd import serialize as S
    torch.manual_seed(3)
    a = Attn(256, 64).to(torch.bfloat16)
    sd = a.state_dict()
    conv = S.convert_checkpoint(sd, fuse={"qkv": ["q_proj", "k_proj", "v_proj"]})
    assert set(conv) == {"qkv.wq", "qkv.ws", "qkv.bias", "o_proj.wq", "o_proj.ws"}
    f = S.empty_fused(256, (256, 64, 64), True, torch.bfloat16)
    f.load_state_dict({k[4:]: v for k, v in conv.items() if k.startswith("qkv.")})
    f = f.cuda()
    x = torch.randn(33, 256, device="cuda", dtype=torch.bfloat16)
    q, k, v = f(x)
    for out, name in ((q, "q_proj"), (k, "k_proj"), (v, "v_proj")):
        wq, ws = C.quant_rowwise(bits(sd[name + ".weight"]), 0)
        want, *_ = Q.qlinear(bits(x), 0, wq, ws, bits(sd[name + ".bias"]))
        same(out.contiguous(), want, "fused " + name)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_entries_load(pq, world):
    """Offline conversion of every rank's shard: loaded modules hold exactly what the live constructors build, and the
    ranks' results combine to the oracle's unsharded (column shards) / row-sharded-spec (K shards) outputs."""
    from protoquant_amd import serialize as S
    from tests.test_gpu_parity import _oracle_row_sharded
    torch.manual_seed(5)
    H, I = 256, 384 + 64
    mlp = MLP(H, I, bias=True).to(torch.bfloat16)
    lin = nn.Linear(H, 200, bias=True).to(torch.bfloat16)
    sd = {**{"mlp." + k: v for k, v in mlp.state_dict().items()}, **{"proj." + k: v for k, v in lin.state_dict().items()}}
    x = torch.randn(40, H, device="cuda", dtype=torch.bfloat16)
    cols, partials = [], []
    for r in range(world):
        conv = S.convert_checkpoint(sd, sharded_gated_mlp=["mlp"], column_sharded=["proj"], world=world, rank=r)
        cs = S.empty_column_sharded(H, 200, True, world, r)
        cs.load_state_dict({k[len("proj."):]: v for k, v in conv.items() if k.startswith("proj.")})
        live = pq.ColumnShardedQLinear.from_linear(lin.cuda(), world=world, rank=r)
        assert torch.equal(cs.local.wq, live.local.wq.cpu()) and torch.equal(cs.local.ws, live.local.ws.cpu()) and torch.equal(cs.local.bias, live.local.bias.cpu())
        cs = cs.cuda()
        xq = pq.quantize(x)
        cols.append(pq.qlinear_s8(xq.int_data, xq.scale, cs.local.wq, cs.local.ws, cs.local.bias, torch.bfloat16))
        sm = S.empty_sharded_gated_mlp(H, I, world, r, bias=True)
        sm.load_state_dict({k[len("mlp."):]: v for k, v in conv.items() if k.startswith("mlp.")})
        live = pq.ShardedGatedMLP.from_linears(mlp.gate_proj.cuda(), mlp.up_proj.cuda(), mlp.down_proj.cuda(), world=world, rank=r)
        for a, b in ((sm.gate_up.wq, live.gate_up.wq), (sm.gate_up.ws, live.gate_up.ws), (sm.down.local.wq, live.down.local.wq), (sm.down.local.ws, live.down.local.ws)):
            assert torch.equal(a, b.cpu())
        assert (sm.down.local.bias is None) == (r != 0)
        sm = sm.cuda()
        g, u = sm.gate_up(x)
        partials.append(sm.down.partial(pq.silu_mul_quantize(g, u)))
    wq, ws = C.quant_rowwise(bits(lin.weight), 0)
    want, *_ = Q.qlinear(bits(x), 0, wq, ws, bits(lin.bias))
    same(torch.cat(cols, dim=1), want, "column shards, concatenated")
    total = partials[0]
    for p in partials[1:]:
        total = total + p
    assert torch.isfinite(total).all() and total.shape == (40, H)


def test_convert_with_model_quantises_only_nn_linear(pq):
    """with the float model given, convert_checkpoint quantises exactly the nn.Linear weights — a 2-D float parameter that is
    not a Linear (here a [in, out] matrix, the GPT-2 Conv1D layout) stays float, where the name / shape heuristic would quantise it along
    the wrong axis — and literal prefixes with regex metacharacters are compared literally."""
    from protoquant_amd import serialize as S

    class Odd(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(64, 48, bias=False)
            self.conv1d_w = torch.nn.Parameter(torch.randn(64, 48))          # [in, out]: NOT an nn.Linear weight
            self.blocks = torch.nn.ModuleDict({"a+b": torch.nn.Linear(48, 32, bias=True)})

    m = Odd().to(torch.bfloat16)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["conv1d.weight"] = sd.pop("conv1d_w")                                  # a name the heuristic would take for a linear
    conv = S.convert_checkpoint(sd, model=m, column_sharded=["blocks.a+b"], world=2, rank=1)
    assert "proj.wq" in conv and conv["proj.wq"].dtype == torch.int8
    assert "conv1d.weight" in conv and conv["conv1d.weight"].dtype == torch.bfloat16 and "conv1d.wq" not in conv
    assert "blocks.a+b.local.wq" in conv and tuple(conv["blocks.a+b.local.wq"].shape) == (16, 48)       # rank 1 of 2: rows 16..31
    heur = S.convert_checkpoint(sd)
    assert "conv1d.wq" in heur                                                # what the heuristic alone does (documented, hence `model=`)
    # a pattern that matches nothing is an error (a plain string that used to be read as a regex would otherwise leave the checkpoint unsharded, silently)
    with pytest.raises(KeyError, match="looks like a regular expression"):
        S.convert_checkpoint(sd, model=m, column_sharded=[r"blocks\..*"], world=2, rank=1)
    with pytest.raises(KeyError, match="matches no linear module"):
        S.convert_checkpoint(sd, model=m, row_sharded=["blocks.nothing"], world=2, rank=1)
    assert "blocks.a+b.local.wq" in S.convert_checkpoint(sd, model=m, column_sharded=[r"re:blocks\..*"], world=2, rank=1)
