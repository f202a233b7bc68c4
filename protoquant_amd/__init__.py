"""protoquant_amd — MI355X-native dynamic-int8 linear path behind the protoquant Python surface
(QTensor, quantize(), dequantize(), qlinear).  Hot path = hand-written HIP (gfx950) in
libpq_hip.so reached through the C-ABI in include/pq_hip.h; there is no CPU/eager fallback."""
from .qtensor import QTensor, quantize, dequantize, silu_mul_quantize, rmsnorm_quantize, silu_mul_rowamax, silu_mul_quantize_with_amax, rowamax, quantize_with_amax
from .qlinear import qlinear, qlinear_s8, qlinear_s8_t, qlinear_s8_kslabs, qlinear_dyn, int_mm, clear_workspaces, swap_linears, FusedQLinear, GatedMLP
from .sharded import (ColumnShardedGatedMLP, ColumnShardedQLinear, RcclColumnGather, RcclRowReduceScatter, RowShardedQLinear, ShardedGatedMLP,
                      gather_columns, gather_columns_overlapped, gather_rows_t, reduce_rows, shard_bounds)

__all__ = ["QTensor", "quantize", "dequantize", "qlinear", "qlinear_s8", "qlinear_dyn", "int_mm", "swap_linears", "FusedQLinear", "GatedMLP", "silu_mul_quantize", "rmsnorm_quantize",
           "ColumnShardedQLinear", "RcclColumnGather", "gather_columns", "shard_bounds", "RowShardedQLinear", "ShardedGatedMLP",
           "RcclRowReduceScatter", "reduce_rows", "gather_columns_overlapped", "gather_rows_t", "qlinear_s8_t", "clear_workspaces",
           "ColumnShardedGatedMLP", "qlinear_s8_kslabs", "silu_mul_rowamax", "silu_mul_quantize_with_amax", "rowamax", "quantize_with_amax"]
