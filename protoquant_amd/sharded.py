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

from .qlinear import FusedQLinear, gemm_operands, qlinear, qlinear_s8, qlinear_s8_kslabs, qlinear_s8_t
from .qtensor import QTensor, quantize, quantize_with_amax, rowamax, silu_mul_quantize, silu_mul_quantize_with_amax, silu_mul_rowamax


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
    """The same exchange through the native C-ABI (include/pq_rccl.h): a dedicated RCCL communicator, ncclAllGather into a
    stacked workspace and the layout kernel (pq_allgather_cols_v: equal OR ragged shards), all stream-ordered on torch's
    current stream; a row-chunked overlapped form (pq_allgather_cols_rows_async + pq_comm_join: the exchange of a row block
    runs on the communicator's side stream while the next block's GEMM runs on torch's stream — no allocation per chunk); and
    the contiguous gather of TRANSPOSED shards (pq_allgather_rows_t: no staging, no layout kernel).
    The 128-byte unique id is created on rank 0 and distributed with torch.distributed (any backend)."""

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

    def comm_ranks(self) -> int:
        """The rank count RCCL itself reports for the communicator (ncclCommCount)."""
        n = self._ct.c_int32(0)
        self._R.check(self._R.lib().pq_comm_count(self._comm, self._ct.byref(n)), "pq_comm_count")
        return int(n.value)

    def _workspace(self, M, n_total, code, device):
        need = self._R.lib().pq_allgather_cols_v_workspace_bytes(self.world, M, n_total, code)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=device)
        return self._ws

    def _check(self, y_local, n_total):
        lo, hi = shard_bounds(n_total, self.world, self.rank)
        if y_local.dim() != 2 or y_local.shape[1] != hi - lo:
            raise ValueError(f"rank {self.rank}: shard has shape {tuple(y_local.shape)}, expected [M, {hi - lo}]")
        if y_local.shape[1] > 1 and y_local.stride(1) != 1:
            raise ValueError("shard rows must be contiguous")

    def gather_into(self, y_local: torch.Tensor, out: torch.Tensor, n_total: int) -> torch.Tensor:
        """y_local[M, width_r] (any leading dimension) -> out[M, n_total] on every rank (out preallocated)."""
        from . import _lib as L
        self._check(y_local, n_total)
        M = y_local.shape[0]
        if out.shape != (M, n_total) or out.dtype != y_local.dtype or out.device != y_local.device or (n_total > 1 and out.stride(1) != 1):
            raise ValueError(f"out must be a row-major [{M}, {n_total}] {y_local.dtype} tensor on {y_local.device}")
        code = L.dtype_code(y_local.dtype)
        ws = self._workspace(M, n_total, code, y_local.device)
        with torch.cuda.device(y_local.device):
            self._R.check(self._R.lib().pq_allgather_cols_v(self._comm, y_local.data_ptr(), L.ld(y_local), out.data_ptr(), L.ld(out), M, n_total,
                                                            code, ws.data_ptr(), ws.numel(), L.stream_ptr(y_local)), "pq_allgather_cols_v")
        return out

    def __call__(self, y_local: torch.Tensor, n_total: int) -> torch.Tensor:
        out = torch.empty((y_local.shape[0], n_total), dtype=y_local.dtype, device=y_local.device)
        return self.gather_into(y_local, out, n_total)

    def gather_rows_async(self, y_local: torch.Tensor, out: torch.Tensor, m0: int, m1: int, n_total: int) -> None:
        """Exchange rows [m0, m1) on the communicator's side stream, behind everything already enqueued on torch's current
        stream; the caller keeps launching on the current stream.  join() before anything reads `out`."""
        from . import _lib as L
        self._check(y_local, n_total)
        M = y_local.shape[0]
        code = L.dtype_code(y_local.dtype)
        ws = self._workspace(M, n_total, code, y_local.device)
        with torch.cuda.device(y_local.device):
            self._R.check(self._R.lib().pq_allgather_cols_rows_async(self._comm, y_local.data_ptr(), L.ld(y_local), out.data_ptr(), L.ld(out), M, m0, m1,
                                                                     n_total, code, ws.data_ptr(), ws.numel(), L.stream_ptr(y_local)),
                          "pq_allgather_cols_rows_async")

    def join(self, device=None) -> None:
        """torch's current stream waits for every exchange issued so far."""
        stream = torch.cuda.current_stream(device).cuda_stream
        self._R.check(self._R.lib().pq_comm_join(self._comm, stream), "pq_comm_join")

    def gather_t(self, yt_local: torch.Tensor, n_total: int, out: torch.Tensor | None = None) -> torch.Tensor:
        """Transposed shards yt_local[width_r, M] (contiguous) -> yt[n_total, M]: one contiguous collective, no layout pass."""
        from . import _lib as L
        lo, hi = shard_bounds(n_total, self.world, self.rank)
        if yt_local.dim() != 2 or yt_local.shape[0] != hi - lo or not yt_local.is_contiguous():
            raise ValueError(f"rank {self.rank}: transposed shard must be a contiguous [{hi - lo}, M] tensor, got {tuple(yt_local.shape)}")
        M = yt_local.shape[1]
        if out is None:
            out = torch.empty((n_total, M), dtype=yt_local.dtype, device=yt_local.device)
        elif out.shape != (n_total, M) or not out.is_contiguous() or out.dtype != yt_local.dtype or out.device != yt_local.device:
            raise ValueError(f"out must be a contiguous [{n_total}, {M}] tensor")
        with torch.cuda.device(yt_local.device):
            self._R.check(self._R.lib().pq_allgather_rows_t(self._comm, yt_local.data_ptr(), out.data_ptr(), n_total, M, L.dtype_code(yt_local.dtype),
                                                            L.stream_ptr(yt_local)), "pq_allgather_rows_t")
        return out

    def allreduce_max_(self, bits: torch.Tensor) -> torch.Tensor:
        """In-place exact MAX over the ranks of an int32 tensor of non-negative-float bit patterns (pq_allreduce_max_u32: ncclUint32 / ncclMax)."""
        from . import _lib as L
        if bits.dtype != torch.int32 or not bits.is_contiguous():
            raise ValueError("allreduce_max_: a contiguous int32 tensor of f32 bit patterns is expected")
        with torch.cuda.device(bits.device):
            self._R.check(self._R.lib().pq_allreduce_max_u32(self._comm, bits.data_ptr(), bits.numel(), L.stream_ptr(bits)), "pq_allreduce_max_u32")
        return bits

    def gather_stacked(self, shard: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """Contiguous all-gather of equal blocks: shard[...] (any dtype, contiguous) -> out[world, ...] (pq_allgather_bytes: no staging, no layout pass)."""
        from . import _lib as L
        if not shard.is_contiguous():
            raise ValueError("gather_stacked: the shard must be contiguous")
        if out is None:
            out = torch.empty((self.world, *shard.shape), dtype=shard.dtype, device=shard.device)
        elif out.shape != (self.world, *shard.shape) or out.dtype != shard.dtype or out.device != shard.device or not out.is_contiguous():
            raise ValueError(f"gather_stacked: out must be a contiguous {(self.world, *shard.shape)} {shard.dtype} tensor")
        with torch.cuda.device(shard.device):
            self._R.check(self._R.lib().pq_allgather_bytes(self._comm, shard.data_ptr(), out.data_ptr(), shard.numel() * shard.element_size(), L.stream_ptr(shard)),
                          "pq_allgather_bytes")
        return out

    def close(self):
        if self._comm:
            self._R.check(self._R.lib().pq_comm_destroy(self._comm), "pq_comm_destroy")
            self._comm = self._ct.c_void_p()


def gather_rows_t(yt_local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """torch.distributed form of RcclColumnGather.gather_t: transposed shards [width_r, M] -> yt[n_total, M].  Row blocks are
    contiguous, so equal shards are ONE all_gather_into_tensor straight into the result; ragged shards use one broadcast per rank
    into its row block of the result (no padding, no layout pass)."""
    world = dist.get_world_size(group)
    M = yt_local.shape[1]
    out = yt_local.new_empty((n_total, M))
    if n_total % world == 0:
        dist.all_gather_into_tensor(out.view(-1), yt_local.contiguous().view(-1), group=group)
        return out
    rank = dist.get_rank(group)
    for r in range(world):                  # an all-gather-v as one broadcast per rank, each landing in place (as the native form does)
        a, b = shard_bounds(n_total, world, r)
        if r == rank:
            out[a:b].copy_(yt_local)
        if b > a:
            dist.broadcast(out[a:b], src=dist.get_global_rank(group, r) if group is not None else r, group=group)
    return out


def allreduce_max_bits(bits: torch.Tensor, world: int, native: "RcclColumnGather | None" = None, group=None) -> torch.Tensor:
    """In-place exact MAX over the ranks of int32 f32-bit-patterns of non-negative values (sign bit clear: the signed and the unsigned order agree; NaNs sort above +Inf)."""
    if world == 1:
        return bits
    if native is not None:
        return native.allreduce_max_(bits)
    dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=group)
    return bits


def gather_stacked(block: torch.Tensor, world: int, native: "RcclColumnGather | None" = None, group=None) -> torch.Tensor:
    """Contiguous all-gather of equal blocks -> [world, *block.shape] (the layout an all-gather leaves; consumers walk it in place: qlinear_s8_kslabs)."""
    if native is not None:
        return native.gather_stacked(block.contiguous())
    out = block.new_empty((world, *block.shape))
    if world == 1:
        out[0].copy_(block)
    else:
        dist.all_gather_into_tensor(out.view(-1), block.contiguous().view(-1), group=group)
    return out


class ColumnShardedQLinear(nn.Module):
    """qlinear whose int8 weight rows [n0:n1) live on this rank; forward returns the full y[..., N].

    layout="rows" (default): the local GEMM writes y[:, n0:n1] shards; the all-gather lands them stacked and one layout pass
    builds row-major y[M, N].  layout="transposed" (SURVEY.md §8(e) option 1): the local GEMM writes the TRANSPOSED shard
    yt[n0:n1, :] (pq_qlinear_s8_t, same bits), whose all-gather is contiguous — no staging buffer, no layout kernel.  forward_t()
    hands out yt[N, M] itself (for a consumer that is fed transposed or indexes logically); forward() stays a drop-in for
    nn.Linear and returns ROW-MAJOR memory (one transpose copy: stock consumers such as `self.q_proj(h).view(...)` raise on a
    strided view) unless the module was built with transposed_view=True, in which case it returns yt.t(), strides (1, M).
    overlap_chunks > 1 (layout="rows"): the rows are cut into blocks and each block's exchange overlaps the next block's GEMM —
    natively on the communicator's side stream (native_gather) or through torch.distributed's async collectives."""

    def __init__(self, local: qlinear, out_features: int, group=None, native_gather: "RcclColumnGather | None" = None,
                 overlap_chunks: int = 1, layout: str = "rows", transposed_view: bool = False):
        super().__init__()
        if layout not in ("rows", "transposed"):
            raise ValueError("layout must be 'rows' or 'transposed'")
        self.transposed_view = transposed_view      # layout="transposed" only: forward() may return the strided view yt.t()
        self.local, self.out_features, self.group = local, out_features, group
        self.in_features = local.in_features
        self.native_gather = native_gather          # optional: exchange through libpq_rccl.so instead of torch.distributed
        self.overlap_chunks = overlap_chunks        # > 1: row blocks, each block's gather overlapping the next block's GEMM
        self.layout = layout

    @classmethod
    def from_linear(cls, lin: nn.Linear, group=None, native_gather=None, overlap_chunks: int = 1, layout: str = "rows",
                    world=None, rank=None, transposed_view: bool = False) -> "ColumnShardedQLinear":
        if (world is None) != (rank is None):
            raise ValueError("ColumnShardedQLinear.from_linear: pass world and rank together (or neither: the process group's)")
        if world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_bounds(lin.out_features, world, rank)
        sub = nn.Linear(lin.in_features, hi - lo, bias=lin.bias is not None, device=lin.weight.device, dtype=lin.weight.dtype)
        with torch.no_grad():
            sub.weight.copy_(lin.weight[lo:hi])
            if lin.bias is not None:
                sub.bias.copy_(lin.bias[lo:hi])
        return cls(qlinear.from_linear(sub), lin.out_features, group, native_gather, overlap_chunks, layout, transposed_view)

    # the three device steps, separate so that host logic can be tested with them stubbed (tests/test_dist_gloo.py)
    def _quantize(self, x):
        return quantize(x, axis=-1)                                  # replicated activation: every rank runs K1 itself

    def _local_rows(self, codes, scales, dtype, out=None):
        codes, wq = gemm_operands(self.local, codes)                 # (K not a multiple of 128: both zero-padded, same bits)
        return qlinear_s8(codes, scales, wq, self.local.ws, self.local.bias, dtype, out=out)

    def _local_t(self, codes, scales, dtype):
        codes, wq = gemm_operands(self.local, codes)
        return qlinear_s8_t(codes, scales, wq, self.local.ws, self.local.bias, dtype)

    def forward_t(self, x: torch.Tensor) -> torch.Tensor:
        """The transposed result yt[N, M] (M = all leading dimensions of x flattened), contiguous: the local GEMM writes transposed
        shards and the gather lands them in place.  Any layout setting; no layout pass, no copy."""
        xq = self._quantize(x)
        codes = xq.int_data.reshape(-1, self.in_features)
        yt_local = self._local_t(codes, xq.scale, x.dtype)
        return self.native_gather.gather_t(yt_local, self.out_features) if self.native_gather is not None else \
            gather_rows_t(yt_local, self.out_features, self.group)

    # device steps of forward_sharded_input (stubbed in tests/test_dist_gloo.py)
    def _local_amax(self, x2):
        return rowamax(x2)

    def _encode(self, x2, amax_bits):
        return quantize_with_amax(x2, amax_bits)

    def _local_rows_stacked(self, stacked, scales, dtype):
        return qlinear_s8_kslabs(stacked, scales, self.local.wq, self.local.ws, self.local.bias, dtype)

    def forward_sharded_input(self, x_local: torch.Tensor) -> torch.Tensor:
        """The input activation is itself COLUMN-sharded: x_local[..., K/G] is this rank's block of the K input features (its heads of the attention output in front of a
        column-sharded `o` projection).  Instead of gathering the bf16 blocks and quantising the whole [M, K] activation on every rank, the int8 CODES travel (as in
        ColumnShardedGatedMLP): local row amax -> ONE all-reduce(max) of M 32-bit patterns -> local encode against the global amax -> all-gather of the int8 blocks
        (1 byte per element instead of 2, each column quantised once) -> this rank's weight rows against the stacked blocks, walked in place.  The result is forward()'s
        on the concatenated activation, bit for bit (max is exact and order-free).  Needs in_features % world == 0; layout "rows" only."""
        world, rank = (self.native_gather.world, self.native_gather.rank) if self.native_gather is not None else (dist.get_world_size(self.group), dist.get_rank(self.group))
        if self.in_features % world or x_local.shape[-1] != self.in_features // world:
            raise ValueError(f"forward_sharded_input: expected [..., {self.in_features}/{world}] input features, got {x_local.shape[-1]}")
        x2 = x_local.reshape(-1, x_local.shape[-1])
        amax = allreduce_max_bits(self._local_amax(x2), world, self.native_gather, self.group)
        xq = self._encode(x2, amax)
        stacked = gather_stacked(xq.int_data.reshape(x2.shape), world, self.native_gather, self.group)
        y_local = self._local_rows_stacked(stacked, xq.scale, x_local.dtype)
        if self.native_gather is not None:
            y = self.native_gather(y_local, self.out_features)
        else:
            y = y_local if world == 1 else gather_columns(y_local, self.out_features, self.group)
        return y.reshape(*x_local.shape[:-1], self.out_features)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.layout == "transposed":
            y = self.forward_t(x).t()                                # [M, N] view, strides (1, M)
            if not self.transposed_view:
                y = y.contiguous()
            return y if x.dim() == 2 else y.unflatten(0, x.shape[:-1])
        xq = self._quantize(x)
        codes = xq.int_data.reshape(-1, self.in_features)
        M = codes.shape[0]
        if self.overlap_chunks > 1 and self.native_gather is not None:
            lo, hi = shard_bounds(self.out_features, self.native_gather.world, self.native_gather.rank)
            y_local = torch.empty((M, hi - lo), dtype=x.dtype, device=x.device)
            y = torch.empty((M, self.out_features), dtype=x.dtype, device=x.device)
            for c in range(self.overlap_chunks):
                m0, m1 = shard_bounds(M, self.overlap_chunks, c)
                if m1 > m0:
                    self._local_rows(codes[m0:m1], xq.scale[m0:m1], x.dtype, out=y_local[m0:m1])
                    self.native_gather.gather_rows_async(y_local, y, m0, m1, self.out_features)
            self.native_gather.join(x.device)       # the side stream's reads of y_local are ordered before anything enqueued from here on
            return y.reshape(*x.shape[:-1], self.out_features)
        if self.overlap_chunks > 1:
            def rows(m0, m1):
                return self._local_rows(codes[m0:m1], xq.scale[m0:m1], x.dtype)
            y = gather_columns_overlapped(rows, M, self.out_features, self.overlap_chunks, x.dtype, x.device, self.group)
            return y.reshape(*x.shape[:-1], self.out_features)
        y_local = self._local_rows(codes, xq.scale, x.dtype)
        if self.native_gather is not None:
            y = self.native_gather(y_local, self.out_features)
        else:
            y = gather_columns(y_local, self.out_features, self.group)
        return y.reshape(*x.shape[:-1], self.out_features)


class ColumnShardedGatedMLP(nn.Module):
    """The gated MLP with gate, up AND down column-sharded (north_star's scheme for BASELINE config 5) and an INT8-CODE exchange between them.  The plain composition —
    gather the bf16 gate and up shards, silu*mul and quantise the gathered [M, I] activation on every rank — moves 4 bytes per intermediate element and
    repeats the quantisation G times.  Here, per token: silu*mul is column-local, so every rank computes the amax of ITS I/G columns (pq_silu_mul_rowamax); the row amax
    is an exact max, so ONE all-reduce(max) of M 32-bit patterns makes it global; every rank encodes its own columns against it (pq_silu_mul_quant_rowwise_amax) and
    the int8 blocks [M, I/G] are all-gathered — 1 byte per element, each column quantised once.  The gather leaves the blocks STACKED [G, M, I/G]; the down shard's GEMM
    walks them in place (pq_qlinear_s8_kslabs: an integer sum has no order), so no layout pass follows.  Codes, scales and output are the unsharded GatedMLP's, bit
    for bit (max is exact and order-free; S1-S5 are per element).  The output shards [M, H/G] are gathered like any ColumnShardedQLinear's.
    Needs intermediate % world == 0.  native: an RcclColumnGather (libpq_rccl.so) — otherwise torch.distributed collectives."""

    def __init__(self, gate_up_local: FusedQLinear, down_local: qlinear, hidden: int, intermediate: int, group=None, native: "RcclColumnGather | None" = None,
                 world: int | None = None, rank: int | None = None):
        super().__init__()
        self.gate_up, self.down = gate_up_local, down_local
        self.hidden, self.intermediate, self.group, self.native = hidden, intermediate, group, native
        if native is not None:
            world, rank = native.world, native.rank
        elif world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.world, self.rank = world, rank
        if intermediate % world:
            raise ValueError(f"ColumnShardedGatedMLP: intermediate size {intermediate} is not a multiple of the world size {world} (the stacked code blocks must be equal)")

    @classmethod
    def from_linears(cls, gate: nn.Linear, up: nn.Linear, down: nn.Linear, group=None, native=None, world=None, rank=None) -> "ColumnShardedGatedMLP":
        if (world is None) != (rank is None):
            raise ValueError("ColumnShardedGatedMLP.from_linears: pass world and rank together (or neither: the process group's)")
        if world is None:
            world, rank = (native.world, native.rank) if native is not None else (dist.get_world_size(group), dist.get_rank(group))

        def rows(lin, lo, hi):
            sub = nn.Linear(lin.in_features, hi - lo, bias=lin.bias is not None, device=lin.weight.device, dtype=lin.weight.dtype)
            with torch.no_grad():
                sub.weight.copy_(lin.weight[lo:hi])
                if lin.bias is not None:
                    sub.bias.copy_(lin.bias[lo:hi])
            return sub
        ilo, ihi = shard_bounds(gate.out_features, world, rank)
        hlo, hhi = shard_bounds(down.out_features, world, rank)
        return cls(FusedQLinear.from_linears(rows(gate, ilo, ihi), rows(up, ilo, ihi)), qlinear.from_linear(rows(down, hlo, hhi)),
                   down.out_features, gate.out_features, group, native, world, rank)

    # ---- device steps (separate so that the host logic can be tested with them stubbed: tests/test_dist_gloo.py)
    def _gate_up(self, x):
        return self.gate_up(x)                                       # (g, u): this rank's I/G intermediate channels, one fused GEMM on the replicated input

    def _local_amax(self, g, u):
        return silu_mul_rowamax(g, u)

    def _encode(self, g, u, amax_bits):
        return silu_mul_quantize_with_amax(g, u, amax_bits)

    def _down(self, stacked, scale, dtype):
        return qlinear_s8_kslabs(stacked, scale, self.down.wq, self.down.ws, self.down.bias, dtype)

    # ---- exchange steps
    def _amax_allreduce(self, bits):
        return allreduce_max_bits(bits, self.world, self.native, self.group)

    def _gather_codes(self, codes):
        return gather_stacked(codes, self.world, self.native, self.group)

    def _gather_out(self, y_local):
        if self.native is not None:
            return self.native(y_local, self.hidden)
        if self.world == 1:
            return y_local
        return gather_columns(y_local, self.hidden, self.group)

    def hidden_codes(self, x: torch.Tensor):
        """(stacked int8 codes [G, M, I/G], row scales [M]) of the quantised silu(gate(x)) * up(x): the exchange half of forward()"""
        g, u = self._gate_up(x)
        g2, u2 = g.reshape(-1, g.shape[-1]), u.reshape(-1, u.shape[-1])
        amax = self._amax_allreduce(self._local_amax(g2, u2))
        hq = self._encode(g2, u2, amax)
        return self._gather_codes(hq.int_data.reshape(g2.shape)), hq.scale

    def forward(self, x) -> torch.Tensor:
        """x: the replicated activation [..., H], or its per-token QTensor (e.g. from rmsnorm_quantize)"""
        stacked, scale = self.hidden_codes(x)
        dtype = x.orig_dtype if isinstance(x, QTensor) else x.dtype
        y = self._gather_out(self._down(stacked, scale, dtype))
        return y.reshape(*x.shape[:-1], self.hidden)


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
        if (world is None) != (rank is None):
            raise ValueError("RowShardedQLinear.from_linear: pass world and rank together (or neither: the process group's)")
        if world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        return cls(cls.shard_of(lin, world, rank), lin.in_features, group, scatter, native)

    def partial(self, x) -> torch.Tensor:
        """This rank's f32 contribution to the whole output, [M, N]; x: the local activation slice [..., K_r] or its QTensor."""
        xq = x if isinstance(x, QTensor) else quantize(x, axis=-1)
        if xq.shape[-1] != self.local.in_features:
            raise ValueError(f"RowShardedQLinear: local input has {xq.shape[-1]} features, this rank owns {self.local.in_features}")
        b = self.local.bias.float() if self.local.bias is not None else None
        codes, wq = gemm_operands(self.local, xq.int_data.reshape(-1, self.local.in_features))      # (K_r not a multiple of 128, e.g. 11008 / 8: zero-padded, same bits)
        return qlinear_s8(codes, xq.scale, wq, self.local.ws, b, torch.float32)

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
        if (world is None) != (rank is None):
            raise ValueError("ShardedGatedMLP.from_linears: pass world and rank together (or neither: the process group's)")
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
