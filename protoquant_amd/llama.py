"""Call-site integration for Llama-family decoder layers (transformers' LlamaDecoderLayer shape: input_layernorm -> self_attn with
q_proj / k_proj / v_proj / o_proj -> post_attention_layernorm -> mlp), SURVEY.md §8(f)1-2 applied to a whole model:

* both RMSNorms of a layer are fused into the activation quantisation of the projections they feed (``rmsnorm_quantize``: the
  normalised activation never reaches HBM, the norm costs no traffic beyond the quantisation that had to happen anyway);
* q / k / v share that one quantisation and ONE GEMM launch (``FusedQLinear``, N = hidden + 2 * kv);
* gate / up / down are a ``GatedMLP`` (fused gate+up GEMM, silu*mul fused into the quantisation of down's input).

The stock attention code keeps calling ``self.q_proj(h)``, ``self.k_proj(h)``, ``self.v_proj(h)``: ``h`` is now a per-token
``QTensor`` (it only needs ``.shape`` besides being handed to the projections), the first call runs the fused GEMM and the other
two return their column slices of its output.  Attention itself (rope, SDPA), the residual adds, the final norm and the embedding
stay stock torch-ROCm ops.  ``swap_linears(model, fuse_gated_mlp=True)`` must have run first."""
from __future__ import annotations

import torch
from torch import nn

from .qlinear import FusedQLinear, GatedMLP, qlinear
from .qtensor import QTensor, rmsnorm_quantize


class RMSNormQuant(nn.Module):
    """RMSNorm whose output is the per-token int8 quantisation of the normalised activation (QSPEC N1-N6 then Q1-Q6)."""

    def __init__(self, weight: torch.Tensor, eps: float):
        super().__init__()
        self.weight = nn.Parameter(weight.detach().clone(), requires_grad=False)
        self.variance_epsilon = float(eps)

    def forward(self, x: torch.Tensor) -> QTensor:
        return rmsnorm_quantize(x, self.weight, self.variance_epsilon)

    def extra_repr(self):
        return f"{tuple(self.weight.shape)}, eps={self.variance_epsilon} -> int8 per-token QTensor"


class _FusedSlice(nn.Module):
    """Projection number `index` of a FusedQLinear shared by sibling slices.  The attention module's forward pre-hook (installed by
    fuse_llama_layers) runs the fused GEMM ONCE per forward on the hidden states it is called with; the slices, called with that
    same object by the stock attention code, return their parts of it.  Called with anything else (outside the attention forward, a
    different tensor), a slice computes its result from its own input — correct, just not shared."""

    def __init__(self, shared: "_SharedFused", index: int):
        super().__init__()
        self._shared = [shared]            # (a list: the shared module is registered once, on its owner)
        self.index = index

    def forward(self, x):
        return self._shared[0].part(x, self.index)


class _SharedFused(nn.Module):
    """The fused q/k/v GEMM of one attention module and its per-forward result: begin() at the attention's entry, end() at its exit
    (always: exceptions included), so nothing outlives the forward, a recomputed forward (activation checkpointing) recomputes, and
    an input changed in place between two forwards is never served stale."""

    def __init__(self, fused: FusedQLinear):
        super().__init__()
        self.fused = fused
        self._key, self._outs = None, None

    def begin(self, x):
        self._key, self._outs = x, self.fused(x)

    def end(self):
        self._key, self._outs = None, None

    def part(self, x, index):
        if self._outs is not None and self._key is x:
            return self._outs[index]
        return self.fused(x)[index]


def _qkv_pre_hook(mod, args, kwargs):
    x = kwargs.get("hidden_states", args[0] if args else None)
    if x is not None:
        mod.qkv_fused.begin(x)          # (looked up on the module, not captured: a deep copy of the model shares nothing with its original)


def _qkv_post_hook(mod, args, kwargs, out):
    mod.qkv_fused.end()


def _install_qkv_hooks(attn: nn.Module) -> None:
    attn.register_forward_pre_hook(_qkv_pre_hook, with_kwargs=True)
    attn.register_forward_hook(_qkv_post_hook, with_kwargs=True, always_call=True)


def _is_rmsnorm(m) -> bool:
    return hasattr(m, "weight") and hasattr(m, "variance_epsilon") and isinstance(getattr(m, "weight"), torch.Tensor) and m.weight.dim() == 1


def fuse_llama_layers(model: nn.Module, fuse_norms: bool = True, fuse_qkv: bool = True) -> int:
    """Apply the fusions above to every decoder layer found in `model` (in place); returns the number of layers changed."""
    n = 0
    for layer in model.modules():
        attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
        if attn is None or mlp is None or not hasattr(layer, "input_layernorm") or not hasattr(layer, "post_attention_layernorm"):
            continue
        q, k, v = (getattr(attn, p, None) for p in ("q_proj", "k_proj", "v_proj"))
        if not all(isinstance(p, qlinear) for p in (q, k, v)):
            continue
        if fuse_qkv:
            attn.qkv_fused = _SharedFused(FusedQLinear([q, k, v]))
            attn.q_proj, attn.k_proj, attn.v_proj = (_FusedSlice(attn.qkv_fused, i) for i in range(3))
            _install_qkv_hooks(attn)
        if fuse_norms and _is_rmsnorm(layer.input_layernorm):
            layer.input_layernorm = RMSNormQuant(layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
        mlp_ok = isinstance(mlp, GatedMLP) or all(isinstance(getattr(mlp, p, None), qlinear) for p in ("gate_proj", "up_proj"))
        if fuse_norms and _is_rmsnorm(layer.post_attention_layernorm) and mlp_ok:
            layer.post_attention_layernorm = RMSNormQuant(layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon)
        n += 1
    return n


# ---------------------------------------------------------------- BASELINE config 5: the same layers column-sharded over the ranks of one node
class _ShardedInputProj(nn.Module):
    """o_proj of a head-sharded attention: called by the stock attention code with THIS rank's heads of the attention output [..., H / G]; the int8-code exchange of
    ColumnShardedQLinear.forward_sharded_input rebuilds the whole projection (this rank's output channels, then the all-gather of the output shards)."""

    def __init__(self, sharded):
        super().__init__()
        self.sharded = sharded

    def forward(self, x):
        return self.sharded.forward_sharded_input(x)


def _rows_of(lin: nn.Linear, lo: int, hi: int, device) -> nn.Linear:
    sub = nn.Linear(lin.in_features, hi - lo, bias=lin.bias is not None, device=device, dtype=lin.weight.dtype)
    with torch.no_grad():
        sub.weight.copy_(lin.weight[lo:hi])
        if lin.bias is not None:
            sub.bias.copy_(lin.bias[lo:hi])
    return sub


def _q_rows(mod, lo: int, hi: int, device) -> qlinear:
    """rows [lo, hi) of a projection as a qlinear on `device`: an nn.Linear's slice is quantised there; a qlinear's int8 codes, scales and bias are SLICED (per-channel
    quantisation is row-local: the slice of the codes is the codes of the slice) — an int8 checkpoint is sharded without its bf16 weights ever existing."""
    if isinstance(mod, qlinear):
        bias = mod.bias[lo:hi].to(device).contiguous() if mod.bias is not None else None
        dt = mod.bias.dtype if mod.bias is not None else torch.bfloat16
        qt = QTensor(mod.wq[lo:hi].to(device).contiguous(), mod.ws[lo:hi].to(device).contiguous(), 1, dt, torch.Size((hi - lo, mod.in_features)))
        return qlinear.from_qtensor(qt, bias)
    return qlinear.from_linear(_rows_of(mod, lo, hi, device))


def _is_proj(m) -> bool:
    return isinstance(m, (nn.Linear, qlinear))


def shard_llama_layers(model: nn.Module, world: int | None = None, rank: int | None = None, group=None, native=None, device=None,
                       shard_lm_head: bool = True) -> int:
    """BASELINE config 5 as a call site: every linear of every Llama-family decoder layer of `model` — nn.Linear (call this INSTEAD of swap_linears) or already a
    qlinear / GatedMLP (a model loaded from an int8 checkpoint, or after swap_linears: codes and scales are sliced, nothing is re-quantised) — becomes this
    rank's COLUMN shard of the int8 layer — north_star's scheme — and the stock attention / residual code keeps running unchanged, on this rank's heads:

    * q / k / v: the rank's heads (q rows [r H/G, (r+1) H/G), k / v rows of its KV heads) as ONE fused local GEMM on the replicated, RMSNorm-fused int8 input; their
      outputs are consumed by the rank's own attention heads — nothing is gathered (HF's attention infers the head count from the projection's width);
    * o: the rank's output channels; its input — the rank's heads of the attention output — is exchanged as int8 codes (forward_sharded_input), its output shards are
      all-gathered (the residual stream stays replicated);
    * gate / up / down: ColumnShardedGatedMLP (int8-code exchange between them, all-gather of down's output shards);
    * lm_head (shard_lm_head): column-sharded over the vocabulary, logits all-gathered.
    The weights may live anywhere (CPU included): each layer's slices are copied to `device` (default: the current CUDA device) and quantised there, so a model
    larger than one GPU's share can be sharded layer by layer.  native: an RcclColumnGather (libpq_rccl.so) — otherwise torch.distributed collectives over `group`.
    Needs heads % G == 0, kv_heads % G == 0, intermediate % G == 0.  Returns the number of layers changed."""
    import torch.distributed as dist

    from .sharded import ColumnShardedGatedMLP, ColumnShardedQLinear, shard_bounds
    if (world is None) != (rank is None):
        raise ValueError("shard_llama_layers: pass world and rank together (or neither: the communicator's / process group's)")
    if world is None:
        world, rank = (native.world, native.rank) if native is not None else (dist.get_world_size(group), dist.get_rank(group))
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = 0
    for layer in model.modules():
        attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
        if attn is None or mlp is None or not hasattr(layer, "input_layernorm") or not hasattr(layer, "post_attention_layernorm"):
            continue
        q, k, v, o = (getattr(attn, p, None) for p in ("q_proj", "k_proj", "v_proj", "o_proj"))
        if isinstance(mlp, GatedMLP):          # swap_linears(fuse_gated_mlp=True): gate and up are the two row blocks of one fused weight
            I_ = mlp.gate_up.splits[0]
            gu_w = mlp.gate_up
            mk = lambda a, b: qlinear.from_qtensor(QTensor(gu_w.wq[a:b], gu_w.ws[a:b], 1, torch.bfloat16, torch.Size((b - a, gu_w.in_features))),      # noqa: E731
                                                   gu_w.bias[a:b] if gu_w.bias is not None else None)
            g, u, d = mk(0, I_), mk(I_, 2 * I_), mlp.down
        else:
            g, u, d = (getattr(mlp, p, None) for p in ("gate_proj", "up_proj", "down_proj"))
        if not all(_is_proj(p) for p in (q, k, v, o, g, u, d)):
            continue
        hd = int(attn.head_dim)
        nq, nkv = q.out_features // hd, k.out_features // hd
        if nq % world or nkv % world or g.out_features % world or o.in_features != q.out_features:
            raise ValueError(f"shard_llama_layers: {nq} heads / {nkv} KV heads / intermediate {g.out_features} do not split over {world} ranks")
        ql, qh = rank * (nq // world) * hd, (rank + 1) * (nq // world) * hd
        kl, kh = rank * (nkv // world) * hd, (rank + 1) * (nkv // world) * hd
        attn.qkv_fused = _SharedFused(FusedQLinear([_q_rows(q, ql, qh, device), _q_rows(k, kl, kh, device), _q_rows(v, kl, kh, device)]))
        attn.q_proj, attn.k_proj, attn.v_proj = (_FusedSlice(attn.qkv_fused, i) for i in range(3))
        _install_qkv_hooks(attn)
        ol, oh = shard_bounds(o.out_features, world, rank)
        attn.o_proj = _ShardedInputProj(ColumnShardedQLinear(_q_rows(o, ol, oh, device), o.out_features, group, native))
        il, ih = shard_bounds(g.out_features, world, rank)
        hl, hh = shard_bounds(d.out_features, world, rank)
        layer.mlp = ColumnShardedGatedMLP(FusedQLinear([_q_rows(g, il, ih, device), _q_rows(u, il, ih, device)]), _q_rows(d, hl, hh, device),
                                          d.out_features, g.out_features, group, native, world, rank)
        for name in ("input_layernorm", "post_attention_layernorm"):
            nm = getattr(layer, name)
            if _is_rmsnorm(nm):
                setattr(layer, name, RMSNormQuant(nm.weight.to(device), nm.variance_epsilon))
        n += 1
    head = getattr(model, "lm_head", None)
    if shard_lm_head and n and _is_proj(head):
        lo, hi = shard_bounds(head.out_features, world, rank)
        model.lm_head = ColumnShardedQLinear(_q_rows(head, lo, hi, device), head.out_features, group, native)
    return n
