"""Column-sharded qlinear for one node of MI355X GPUs (BASELINE config 5): the int8 weight W[N,K] is split
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

from .qlinear import qlinear, qlinear_s8
from .qtensor import quantize


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

    def __init__(self, local: qlinear, out_features: int, group=None, native_gather: "RcclColumnGather | None" = None):
        super().__init__()
        self.local, self.out_features, self.group = local, out_features, group
        self.in_features = local.in_features
        self.native_gather = native_gather          # optional: exchange through libpq_rccl.so instead of torch.distributed

    @classmethod
    def from_linear(cls, lin: nn.Linear, group=None, native_gather=None) -> "ColumnShardedQLinear":
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_bounds(lin.out_features, world, rank)
        sub = nn.Linear(lin.in_features, hi - lo, bias=lin.bias is not None, device=lin.weight.device, dtype=lin.weight.dtype)
        with torch.no_grad():
            sub.weight.copy_(lin.weight[lo:hi])
            if lin.bias is not None:
                sub.bias.copy_(lin.bias[lo:hi])
        return cls(qlinear.from_linear(sub), lin.out_features, group, native_gather)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xq = quantize(x, axis=-1)                                   # replicated activation: every rank runs K1 itself
        y_local = qlinear_s8(xq.int_data.reshape(-1, self.in_features), xq.scale, self.local.wq, self.local.ws,
                             self.local.bias, x.dtype)
        if self.native_gather is not None:
            y = self.native_gather(y_local, self.out_features)
        else:
            y = gather_columns(y_local, self.out_features, self.group)
        return y.reshape(*x.shape[:-1], self.out_features)
