"""Sharded qlinear for one node of MI355X GPUs.  Second half of this file: the row-sharded (K-split) pairing with a
reduce-scatter.  First half — column-sharded qlinear (BASELINE config 5): the int8 weight W[N,K] is split
along N (output channels) into `world` contiguous blocks, one per rank (one process per GPU); every rank
quantises the replicated activation itself (K1, 9 us — cheaper than a broadcast) and computes y[:, n0:n1]
with the fused kernel; ONE collective — an all-gather of the bf16/fp16 output shards over RCCL/xGMI —
rebuilds y[M,N].  (torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" runs the same host logic on CPU.)

Layout trap: an all-gather concatenates rank buffers along the OUTERMOST axis, so gathering [M, N/G]
shards yields [G, M, N/G], not [M, N].  `gather_columns` therefore gathers the stacked buffer and
permutes once; `stacked=True` returns the [G, M, N/G] view without the extra pass for consumers that
can index shards directly.  The host logic (shard bounds, collective, layout) contains no HIP calls and
is covered by world_size-2 gloo tests on CPU."""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from .qlinear import FusedQLinear, qlinear, qlinear_s8
from .qtensor import QTensor, quantize, silu_mul_quantize


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced split of n output channels: the first n % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_columns(y_local: torch.Tensor, n_total: int, group=None, stacked: bool = False) -> torch.Tensor:
    """All-gather column shards y_local[M, n_r] -> y[M, n_total] (or the stacked [G, M, n_max] buffer).
    Ragged shards (n_total % world != 0) are padded to the largest shard for the collective."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    M = y_local.shape[0]
    n_max = -(-n_total // world)
    lo, hi = shard_bounds(n_total, world, rank)
    if y_local.shape[1] != hi - lo:
        raise ValueError(f"rank {rank}: shard has {y_local.shape[1]} columns, expected {hi - lo}")
    if y_local.shape[1] != n_max:
        pad = y_local.new_zeros((M, n_max))
        pad[:, : hi - lo] = y_local
        y_local = pad
    buf = y_local.new_empty((world, M, n_max))
    dist.all_gather_into_tensor(buf.view(-1), y_local.contiguous().view(-1), group=group)
    if stacked:
        return buf
    if n_total % world == 0:
        return buf.permute(1, 0, 2).reshape(M, n_total)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        parts.append(buf[r, :, : b - a])
    return torch.cat(parts, dim=1)


def gather_columns_overlapped(produce_rows, M: int, n_total: int, chunks: int, dtype, device, group=None) -> torch.Tensor:
    """The same exchange as gather_columns, pipelined against the compute that feeds it: the M rows are cut into `chunks`
    balanced blocks; `produce_rows(m0, m1)` launches the local GEMM for rows [m0, m1) and returns its [m1 - m0, n_r] shard;
    each block's all-gather is issued asynchronously right behind its GEMM, so on RCCL (its own stream) the transfer of block i
    runs while block i+1 computes — at 70B shapes the gather of a whole layer (~190 us at xGMI link rate) is longer than its
    GEMM, and only the last block's transfer stays exposed.  Equal results: every row block is gathered exactly as before."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_total, world, rank)
    n_max = -(-n_total // world)
    y = torch.empty((M, n_total), dtype=dtype, device=device)
    pending = []
    for c in range(chunks):
        m0, m1 = shard_bounds(M, chunks, c)
        if m1 == m0:
            continue
        part = produce_rows(m0, m1)
        if part.shape != (m1 - m0, hi - lo):
            raise ValueError(f"rank {rank}: produce_rows({m0}, {m1}) returned {tuple(part.shape)}, expected {(m1 - m0, hi - lo)}")
        if hi - lo != n_max:
            pad = part.new_zeros((m1 - m0, n_max)); pad[:, : hi - lo] = part; part = pad
        buf = part.new_empty((world, m1 - m0, n_max))
        work = dist.all_gather_into_tensor(buf.view(-1), part.contiguous().view(-1), group=group, async_op=True)
        pending.append((work, buf, part, m0, m1))
    for work, buf, _part, m0, m1 in pending:
        work.wait()
        if n_total % world == 0:
            y[m0:m1] = buf.permute(1, 0, 2).reshape(m1 - m0, n_total)
        else:
            for r in range(world):
                a, b = shard_bounds(n_total, world, r)
                y[m0:m1, a:b] = buf[r, :, : b - a]
    return y


class RcclColumnGather:
    """The same exchange through the native C-ABI (include/pq_rccl.h): a dedicated RCCL communicator, one
    ncclAllGather into a stacked workspace and the layout-fix kernel, all stream-ordered on torch's current
    stream.  The 128-byte unique id is created on rank 0 and distributed with torch.distributed (any backend).
    Requires equal shard widths (n_total % world == 0)."""

    def __init__(self, group=None):
        from . import _rccl
        import ctypes
        self._R, self._ct = _rccl, ctypes
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        ident = [None]
        if self.rank == 0:
            buf = ctypes.create_string_buffer(_rccl.UNIQUE_ID_BYTES)
            _rccl.check(_rccl.lib().pq_comm_unique_id(buf), "pq_comm_unique_id")
            ident = [bytes(buf.raw)]
        dist.broadcast_object_list(ident, src=0, group=group)
        self._comm = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(ident[0], _rccl.UNIQUE_ID_BYTES)
        _rccl.check(_rccl.lib().pq_comm_init_rank(ctypes.byref(self._comm), self.world, idbuf, self.rank), "pq_comm_init_rank")
        self._ws = None

    def __call__(self, y_local: torch.Tensor, n_total: int) -> torch.Tensor:
        from . import _lib as L
        if n_total % self.world or y_local.shape[1] * self.world != n_total:
            raise ValueError("RcclColumnGather needs equal shards (n_total % world == 0)")
        y_local = y_local.contiguous()
        M, n = y_local.shape
        code = L.dtype_code(y_local.dtype)
        need = self._R.lib().pq_allgather_cols_workspace_bytes(self.world, M, n, code)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=y_local.device)
        out = torch.empty((M, n_total), dtype=y_local.dtype, device=y_local.device)
        with torch.cuda.device(y_local.device):
            self._R.check(self._R.lib().pq_allgather_cols(self._comm, self.world, y_local.data_ptr(), out.data_ptr(), M, n, code,
                                                          self._ws.data_ptr(), self._ws.numel(), L.stream_ptr(y_local)),
                          "pq_allgather_cols")
        return out

    def close(self):
        if self._comm:
            self._R.check(self._R.lib().pq_comm_destroy(self._comm), "pq_comm_destroy")
            self._comm = self._ct.c_void_p()


class ColumnShardedQLinear(nn.Module):
    """qlinear whose int8 weight rows [n0:n1) live on this rank; forward returns the full y[..., N]."""

    def __init__(self, local: qlinear, out_features: int, group=None, native_gather: "RcclColumnGather | None" = None,
                 overlap_chunks: int = 1):
        super().__init__()
        self.local, self.out_features, self.group = local, out_features, group
        self.in_features = local.in_features
        self.native_gather = native_gather          # optional: exchange through libpq_rccl.so instead of torch.distributed
        self.overlap_chunks = overlap_chunks        # > 1: row blocks, each block's gather overlapping the next block's GEMM

    @classmethod
    def from_linear(cls, lin: nn.Linear, group=None, native_gather=None, overlap_chunks: int = 1) -> "ColumnShardedQLinear":
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_bounds(lin.out_features, world, rank)
        sub = nn.Linear(lin.in_features, hi - lo, bias=lin.bias is not None, device=lin.weight.device, dtype=lin.weight.dtype)
        with torch.no_grad():
            sub.weight.copy_(lin.weight[lo:hi])
            if lin.bias is not None:
                sub.bias.copy_(lin.bias[lo:hi])
        return cls(qlinear.from_linear(sub), lin.out_features, group, native_gather, overlap_chunks)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xq = quantize(x, axis=-1)                                   # replicated activation: every rank runs K1 itself
        if self.overlap_chunks > 1 and self.native_gather is None:
            codes = xq.int_data.reshape(-1, self.in_features)

            def rows(m0, m1):
                return qlinear_s8(codes[m0:m1], xq.scale[m0:m1], self.local.wq, self.local.ws, self.local.bias, x.dtype)
            y = gather_columns_overlapped(rows, codes.shape[0], self.out_features, self.overlap_chunks, x.dtype, x.device, self.group)
            return y.reshape(*x.shape[:-1], self.out_features)
        y_local = qlinear_s8(xq.int_data.reshape(-1, self.in_features), xq.scale, self.local.wq, self.local.ws,
                             self.local.bias, x.dtype)
        if self.native_gather is not None:
            y = self.native_gather(y_local, self.out_features)
        else:
            y = gather_columns(y_local, self.out_features, self.group)
        return y.reshape(*x.shape[:-1], self.out_features)


# ---------------------------------------------------------------- row-sharded (K-split) qlinear + reduce-scatter
def reduce_rows(partial: torch.Tensor, out_dtype: torch.dtype, group=None, scatter: bool = True) -> torch.Tensor:
    """Sum the ranks' partial outputs partial[M, N] (f32: each rank's K-slice contribution to the WHOLE output).
    scatter=True: reduce-scatter over contiguous, balanced row blocks — rank r gets rows shard_bounds(M, world, r) of the
    sum (ragged M is padded to the largest block for the collective); scatter=False: all-reduce, every rank gets [M, N].
    The sum is formed in f32 and cast once (RNE).  Order across ranks is the backend's: exact for world <= 2."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if partial.dtype != torch.float32 or partial.dim() != 2:
        raise ValueError("reduce_rows: partial outputs must be 2-D float32")
    M, N = partial.shape
    if not scatter:
        total = partial.contiguous().clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        return total.to(out_dtype)
    m_max = -(-M // world)
    lo, hi = shard_bounds(M, world, rank)
    if M % world == 0:
        src = partial.contiguous()
    else:                                          # pad every row block to m_max rows
        src = partial.new_zeros((world, m_max, N))
        for r in range(world):
            a, b = shard_bounds(M, world, r)
            src[r, : b - a] = partial[a:b]
    out = partial.new_empty((m_max, N))
    dist.reduce_scatter_tensor(out.view(-1), src.view(-1), op=dist.ReduceOp.SUM, group=group)
    return out[: hi - lo].to(out_dtype)


class RcclRowReduceScatter:
    """reduce_rows(scatter=True) through the native C-ABI (pq_reduce_scatter_rows): ncclReduceScatter in f32 and the
    cast kernel, stream-ordered on torch's current stream.  Shares the communicator bootstrap with RcclColumnGather.
    Requires M % world == 0."""

    def __init__(self, gather: "RcclColumnGather"):
        self._g = gather
        self._ws = None

    def __call__(self, partial: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
        from . import _lib as L
        g = self._g
        M, N = partial.shape
        if partial.dtype != torch.float32 or M % g.world:
            raise ValueError("RcclRowReduceScatter needs float32 partials with M % world == 0")
        partial = partial.contiguous()
        m = M // g.world
        code = L.dtype_code(out_dtype)
        need = g._R.lib().pq_reduce_scatter_rows_workspace_bytes(g.world, m, N, code)
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = torch.empty((need,), dtype=torch.uint8, device=partial.device)
        out = torch.empty((m, N), dtype=out_dtype, device=partial.device)
        with torch.cuda.device(partial.device):
            g._R.check(g._R.lib().pq_reduce_scatter_rows(g._comm, g.world, partial.data_ptr(), out.data_ptr(), m, N, code,
                                                         self._ws.data_ptr() if need else None, need, L.stream_ptr(partial)),
                       "pq_reduce_scatter_rows")
        return out


class RowShardedQLinear(nn.Module):
    """qlinear whose int8 weight COLUMNS [k0:k1) (input features) live on this rank — the pairing of a column-sharded
    producer: the producer's local output shard is this layer's local input, so no all-gather happens between them.
    Per-channel weight scales are those of the FULL row (every rank's partial accumulators share them); the activation
    slice is quantised per token locally (its own row scales).  forward() returns the rank's ROW block of the output
    (scatter=True, reduce-scatter) or the whole output (scatter=False, all-reduce).  The bias is added on rank 0's partial.

    Numerics: y = cast_rne( sum_r ((f32(acc_r) * xs_r[m]) * ws[n]) (+ bias) ), partial sums in f32 — not the bits of the
    unsharded qlinear (each K-slice has its own activation scale; the result is closer to the float linear, not further)."""

    def __init__(self, local: qlinear, in_features: int, group=None, scatter: bool = True, native: "RcclRowReduceScatter | None" = None):
        super().__init__()
        self.local, self.in_features, self.group, self.scatter, self.native = local, in_features, group, scatter, native
        self.out_features = local.out_features

    @staticmethod
    def shard_of(lin: nn.Linear, world: int, rank: int) -> qlinear:
        """The rank's qlinear: full-row weight quantisation, then the column slice (bias only on rank 0)."""
        from .qtensor import quantize as _q
        qw = _q(lin.weight.detach(), axis=-1)
        k0, k1 = shard_bounds(lin.in_features, world, rank)
        sub = QTensor(qw.int_data[:, k0:k1].contiguous(), qw.scale, 1, qw.orig_dtype, torch.Size((lin.out_features, k1 - k0)))
        return qlinear.from_qtensor(sub, lin.bias.detach().clone() if (lin.bias is not None and rank == 0) else None)

    @classmethod
    def from_linear(cls, lin: nn.Linear, group=None, scatter: bool = True, native=None, world=None, rank=None) -> "RowShardedQLinear":
        """world/rank default to the process group's; passing them builds a given rank's shard offline."""
        if world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        return cls(cls.shard_of(lin, world, rank), lin.in_features, group, scatter, native)

    def partial(self, x) -> torch.Tensor:
        """This rank's f32 contribution to the whole output, [M, N]; x: the local activation slice [..., K_r] or its QTensor."""
        xq = x if isinstance(x, QTensor) else quantize(x, axis=-1)
        if xq.shape[-1] != self.local.in_features:
            raise ValueError(f"RowShardedQLinear: local input has {xq.shape[-1]} features, this rank owns {self.local.in_features}")
        b = self.local.bias.float() if self.local.bias is not None else None
        return qlinear_s8(xq.int_data.reshape(-1, self.local.in_features), xq.scale, self.local.wq, self.local.ws, b, torch.float32)

    def forward(self, x) -> torch.Tensor:
        dtype = x.orig_dtype if isinstance(x, QTensor) else x.dtype
        p = self.partial(x)
        if self.native is not None and self.scatter:
            return self.native(p, dtype)
        return reduce_rows(p, dtype, self.group, self.scatter)


class ShardedGatedMLP(nn.Module):
    """The gated MLP under tensor parallelism without an all-gather: gate and up column-sharded (this rank's I/G
    intermediate channels, one fused GEMM on the replicated input), silu*mul fused into the LOCAL quantisation, down
    row-sharded over the same channels, ONE reduce-scatter (or all-reduce) of [M, H] f32 partials.  Against the
    column-shard + all-gather scheme this moves M*H*4 bytes instead of M*2I*2: 7x less at Llama-70B's shapes."""

    def __init__(self, gate_up_local: FusedQLinear, down: RowShardedQLinear):
        super().__init__()
        self.gate_up, self.down = gate_up_local, down

    @classmethod
    def from_linears(cls, gate: nn.Linear, up: nn.Linear, down: nn.Linear, group=None, scatter: bool = True, native=None,
                     world=None, rank=None):
        if world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_bounds(gate.out_features, world, rank)

        def rows(lin):
            sub = nn.Linear(lin.in_features, hi - lo, bias=lin.bias is not None, device=lin.weight.device, dtype=lin.weight.dtype)
            with torch.no_grad():
                sub.weight.copy_(lin.weight[lo:hi])
                if lin.bias is not None:
                    sub.bias.copy_(lin.bias[lo:hi])
            return sub
        return cls(FusedQLinear.from_linears(rows(gate), rows(up)),
                   RowShardedQLinear.from_linear(down, group, scatter, native, world, rank))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        g, u = self.gate_up(x)
        return self.down(silu_mul_quantize(g, u))
