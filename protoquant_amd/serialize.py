"""Serialised int8 weight format: a converter from a bf16/fp16/f32 checkpoint (a plain ``state_dict``) to the
``wq`` / ``ws`` / ``bias`` tensors the int8 modules hold, and the structural swap that lets a freshly constructed model load it.

Format ``protoquant_amd.int8.v1`` — a flat ``state_dict``:

=========================  =========================================================================================
``<mod>.wq``               int8 ``[N, K]`` row-major, K contiguous — exactly what the GEMM's LDS-DMA reads (no pre-shuffle:
                           the swizzle is applied on the source address)
``<mod>.ws``               fp32 ``[N]``, one scale per output channel (QSPEC Q1-Q6 over the rows of W)
``<mod>.bias``             the checkpoint's bias, unchanged (absent when the linear has none)
fused projections          ``<parent>.<name>.wq`` = the members' codes concatenated along N (scales likewise): per-row
                           quantisation makes the concatenation exact
gated MLP                  ``<mlp>.gate_up.{wq,ws}`` (gate then up) and ``<mlp>.down.{wq,ws}`` — the GatedMLP layout
column shard (rank r of G) rows ``shard_bounds(N, G, r)`` of ``wq`` / ``ws`` / ``bias`` under ``<mod>.local.*``
row shard (rank r of G)    columns ``shard_bounds(K, G, r)`` of ``wq``, the FULL-row ``ws``, the bias on rank 0 only, under
                           ``<mod>.local.*``
everything else            copied through (embeddings, norms, ...)
=========================  =========================================================================================

The quantisation itself runs on the GPU through the product path (``quantize`` -> ``pq_quant_rowwise``): there is no CPU
fallback.  ``save_quantized`` / ``load_quantized`` store the dict as one safetensors file with the format tag in its metadata.
The reference's own serialisation is not observable (/root/reference holds only CODE_OF_CONDUCT.md:1-80): build-defined."""
from __future__ import annotations

import re
from typing import Callable, Iterable

import torch
from torch import nn

from .qlinear import FusedQLinear, GatedMLP, _is_silu, qlinear
from .qtensor import quantize
from .sharded import ColumnShardedQLinear, RowShardedQLinear, ShardedGatedMLP, shard_bounds

FORMAT = "protoquant_amd.int8.v1"


def default_is_linear(name: str, weight: torch.Tensor) -> bool:
    """Which ``<name>.weight`` entries of a checkpoint are nn.Linear weights: 2-D floating tensors that are not embeddings
    or norms (a state_dict does not say; pass your own predicate for other naming schemes)."""
    return weight.dim() == 2 and weight.is_floating_point() and not re.search(r"(embed|wte|wpe|norm|ln_)", name)


def _quant_rows(w: torch.Tensor, device):
    q = quantize(w.to(device), axis=-1)
    return q.int_data, q.scale


def convert_checkpoint(state_dict: dict, *, device="cuda", is_linear: Callable[[str, torch.Tensor], bool] = default_is_linear,
                       fuse: dict[str, Iterable[str]] | None = None, gated_mlp: Iterable[str] = (),
                       column_sharded: Iterable[str] = (), row_sharded: Iterable[str] = (), sharded_gated_mlp: Iterable[str] = (),
                       world: int = 1, rank: int = 0, out_device="cpu", model: nn.Module | None = None) -> dict:
    """state_dict of float weights -> state_dict in FORMAT.

    model: the FLOAT model the checkpoint belongs to (a meta-device instance is enough).  When given, exactly the weights of its
      nn.Linear modules are quantised — the same criterion prepare_for_int8() / swap_linears() use on the loading side — instead of
      the name / shape heuristic `is_linear` (which takes any 2-D float tensor that is not an embedding or a norm: a transposed
      [in, out] weight such as GPT-2's Conv1D would be quantised along the wrong axis).

    fuse: ``{"layers.0.attn.qkv": ["layers.0.attn.q_proj", "layers.0.attn.k_proj", "layers.0.attn.v_proj"]}`` — the members
      are replaced by one FusedQLinear entry.  gated_mlp: prefixes of modules with gate_proj / up_proj / down_proj children
      (-> ``<p>.gate_up.*``, ``<p>.down.*``).  column_sharded / row_sharded: linear prefixes stored as this rank's shard
      (``<p>.local.*``).  sharded_gated_mlp: MLP prefixes stored as ShardedGatedMLP (gate/up column shards fused, down row
      shard with full-row scales).  A pattern is a module prefix, compared LITERALLY ("model.layers.0.mlp": its dots are dots, a name
      with brackets or a plus sign is fine), or a regular expression given explicitly: a compiled ``re.Pattern`` or a string that
      starts with ``re:`` (matched with re.fullmatch).  A pattern that matches no module raises KeyError (a plain string that looks like a
      regular expression gets a hint): the checkpoint is never written silently unfused / unsharded."""
    gated_mlp, column_sharded, row_sharded, sharded_gated_mlp = (list(v) for v in (gated_mlp, column_sharded, row_sharded, sharded_gated_mlp))
    hits: dict[int, int] = {}          # id(pattern) -> modules it matched: a pattern that matches NOTHING is an error, not a silently unsharded checkpoint

    def match(patterns, name):
        found = False
        for p in patterns:
            if isinstance(p, re.Pattern):
                ok = p.fullmatch(name) is not None
            elif isinstance(p, str) and p.startswith("re:"):
                ok = re.fullmatch(p[3:], name) is not None
            else:
                ok = p == name
            if ok:
                hits[id(p)] = hits.get(id(p), 0) + 1
                found = True
        return found

    if model is not None:
        linear_names = {n for n, m in model.named_modules() if isinstance(m, nn.Linear)}
        is_linear = lambda name, w: name in linear_names      # noqa: E731
    sd = dict(state_dict)
    out = {}
    lin = {k[:-len(".weight")]: v for k, v in sd.items() if k.endswith(".weight") and is_linear(k[:-len(".weight")], v)}
    used = set()

    def codes(prefix):
        return _quant_rows(lin[prefix], device)

    def put(prefix, wq, ws, bias):
        out[prefix + ".wq"], out[prefix + ".ws"] = wq.to(out_device), ws.to(out_device)
        if bias is not None:
            out[prefix + ".bias"] = bias.to(out_device)

    def col_shard(wq, ws, bias):
        lo, hi = shard_bounds(wq.shape[0], world, rank)
        return wq[lo:hi].contiguous(), ws[lo:hi].contiguous(), (bias[lo:hi].contiguous() if bias is not None else None)

    def row_shard(wq, ws, bias):
        k0, k1 = shard_bounds(wq.shape[1], world, rank)
        return wq[:, k0:k1].contiguous(), ws, (bias if (bias is not None and rank == 0) else None)

    mlps = sorted({p[: -len(".gate_proj")] for p in lin if p.endswith(".gate_proj")})
    for mp in mlps:
        sharded = match(sharded_gated_mlp, mp)
        if not (sharded or match(gated_mlp, mp)):
            continue
        members = [mp + ".gate_proj", mp + ".up_proj", mp + ".down_proj"]
        if not all(m in lin for m in members):
            raise KeyError(f"{mp}: a gated MLP needs gate_proj, up_proj and down_proj weights")
        (gq, gs), (uq, us), (dq, ds) = (codes(m) for m in members)
        gb, ub, db = (sd.get(m + ".bias") for m in members)
        if (gb is None) != (ub is None):
            raise ValueError(f"{mp}: gate and up must both have a bias or neither")
        if sharded:
            gq, gs, gb = col_shard(gq, gs, gb.to(device) if gb is not None else None)
            uq, us, ub = col_shard(uq, us, ub.to(device) if ub is not None else None)
            dq, ds, db = row_shard(dq, ds, db)
            put(mp + ".gate_up", torch.cat([gq, uq]), torch.cat([gs, us]), torch.cat([gb, ub]) if gb is not None else None)
            put(mp + ".down.local", dq, ds, db)
        else:
            put(mp + ".gate_up", torch.cat([gq, uq]), torch.cat([gs, us]), torch.cat([gb.to(device), ub.to(device)]) if gb is not None else None)
            put(mp + ".down", dq, ds, db)
        used.update(members)
    for fused_name, members in (fuse or {}).items():
        members = list(members)
        if any(m not in lin for m in members):
            raise KeyError(f"{fused_name}: members {members} must all be linear weights of the checkpoint")
        qs = [codes(m) for m in members]
        bs = [sd.get(m + ".bias") for m in members]
        if any((b is None) != (bs[0] is None) for b in bs):
            raise ValueError(f"{fused_name}: either all or none of the fused projections may have a bias")
        put(fused_name, torch.cat([q for q, _ in qs]), torch.cat([s for _, s in qs]), torch.cat([b.to(device) for b in bs]) if bs[0] is not None else None)
        used.update(members)
    for p in lin:
        if p in used:
            continue
        wq, ws = codes(p)
        bias = sd.get(p + ".bias")
        if match(column_sharded, p):
            put(p + ".local", *col_shard(wq, ws, bias.to(device) if bias is not None else None))
        elif match(row_sharded, p):
            put(p + ".local", *row_shard(wq, ws, bias))
        else:
            put(p, wq, ws, bias)
        used.add(p)
    for kind, patterns in (("gated_mlp", gated_mlp), ("sharded_gated_mlp", sharded_gated_mlp), ("column_sharded", column_sharded), ("row_sharded", row_sharded)):
        for p in patterns:
            if hits.get(id(p), 0) == 0:
                shown = p.pattern if isinstance(p, re.Pattern) else p
                hint = ""
                if isinstance(p, str) and not p.startswith("re:") and re.search(r"[\\*+?\[\](){}|^$]", p):
                    hint = ' — it looks like a regular expression: plain strings are compared LITERALLY, write "re:' + p + '" or pass a compiled re.Pattern'
                raise KeyError(f"convert_checkpoint: {kind} pattern {shown!r} matches no linear module of the checkpoint{hint}")
    for k, v in sd.items():
        base = k.rsplit(".", 1)[0]
        if base in used and (k.endswith(".weight") or k.endswith(".bias")):
            continue
        out[k] = v.to(out_device) if isinstance(v, torch.Tensor) else v
    return out


# ---------------------------------------------------------------- constructing the receiving modules without float weights
def empty_qlinear(in_features: int, out_features: int, bias: bool, dtype=torch.bfloat16, device=None) -> qlinear:
    return qlinear(in_features, out_features, bias=bias, device=device, dtype=dtype)


def empty_fused(in_features: int, splits: Iterable[int], bias: bool, dtype=torch.bfloat16, device=None) -> FusedQLinear:
    return FusedQLinear([empty_qlinear(in_features, n, bias, dtype, device) for n in splits])


def empty_gated_mlp(hidden: int, intermediate: int, bias: bool = False, dtype=torch.bfloat16, device=None) -> GatedMLP:
    return GatedMLP(empty_fused(hidden, (intermediate, intermediate), bias, dtype, device), empty_qlinear(intermediate, hidden, bias, dtype, device))


def empty_column_sharded(in_features: int, out_features: int, bias: bool, world: int, rank: int, dtype=torch.bfloat16, device=None,
                         group=None, **kw) -> ColumnShardedQLinear:
    lo, hi = shard_bounds(out_features, world, rank)
    return ColumnShardedQLinear(empty_qlinear(in_features, hi - lo, bias, dtype, device), out_features, group, **kw)


def empty_row_sharded(in_features: int, out_features: int, bias: bool, world: int, rank: int, dtype=torch.bfloat16, device=None,
                      group=None, scatter: bool = True, native=None) -> RowShardedQLinear:
    k0, k1 = shard_bounds(in_features, world, rank)
    return RowShardedQLinear(empty_qlinear(k1 - k0, out_features, bias and rank == 0, dtype, device), in_features, group, scatter, native)


def empty_sharded_gated_mlp(hidden: int, intermediate: int, world: int, rank: int, bias: bool = False, dtype=torch.bfloat16, device=None,
                            group=None, scatter: bool = True, native=None) -> ShardedGatedMLP:
    lo, hi = shard_bounds(intermediate, world, rank)
    gate_up = empty_fused(hidden, (hi - lo, hi - lo), bias, dtype, device)
    down = RowShardedQLinear(empty_qlinear(hi - lo, hidden, bias and rank == 0, dtype, device), intermediate, group, scatter, native)
    return ShardedGatedMLP(gate_up, down)


def prepare_for_int8(model: nn.Module, predicate=None, fuse_gated_mlp: bool = False) -> nn.Module:
    """The structural half of swap_linears(): replace every nn.Linear (and, with fuse_gated_mlp, every Llama-style MLP) by an
    EMPTY int8 module of the same shape — no quantisation, no float weights needed (works on meta-device models) — so that
    ``model.load_state_dict(convert_checkpoint(...))`` fills it.  Empty buffers live on the linear's device unless it is meta,
    then on the CPU; move the model to the GPU after loading."""
    for name, child in list(model.named_children()):
        if fuse_gated_mlp and (predicate is None or predicate(name, child)):
            g, u, d = (getattr(child, n, None) for n in ("gate_proj", "up_proj", "down_proj"))
            if all(isinstance(l, nn.Linear) for l in (g, u, d)) and _is_silu(getattr(child, "act_fn", None)) and \
                    g.in_features == u.in_features and g.out_features == u.out_features == d.in_features and (g.bias is None) == (u.bias is None):
                dev = None if g.weight.device.type == "meta" else g.weight.device
                mlp = GatedMLP(empty_fused(g.in_features, (g.out_features, u.out_features), g.bias is not None, g.weight.dtype, dev),
                               empty_qlinear(d.in_features, d.out_features, d.bias is not None, d.weight.dtype, dev))
                setattr(model, name, mlp)
                continue
        if isinstance(child, nn.Linear) and (predicate is None or predicate(name, child)):
            dev = None if child.weight.device.type == "meta" else child.weight.device
            setattr(model, name, empty_qlinear(child.in_features, child.out_features, child.bias is not None, child.weight.dtype, dev))
        else:
            prepare_for_int8(child, predicate, fuse_gated_mlp)
    return model


def save_quantized(state_dict: dict, path: str) -> None:
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in state_dict.items()}, path, metadata={"format": FORMAT, "qspec": "v1"})


def load_quantized(path: str, device="cpu") -> dict:
    from safetensors import safe_open
    out = {}
    with safe_open(path, framework="pt", device=str(device)) as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise ValueError(f"{path}: format tag {meta.get('format')!r}, expected {FORMAT!r}")
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out
