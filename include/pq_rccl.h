/* pq_rccl.h — C-ABI of libpq_rccl.so: the one exchange step of the column-sharded configuration
 * (BASELINE config 5): an RCCL all-gather of the per-rank output shards over xGMI plus the layout fix.
 *
 * Reference side: BASELINE.json's north_star ("shard the weight matrix column-wise across the 8 GPUs of one
 * node with an RCCL all-gather over xGMI"); no reference source exists in the mount
 * (/root/reference/CODE_OF_CONDUCT.md:1-80 only).  One process per GPU; the communicator is bootstrapped from a
 * 128-byte unique id that rank 0 creates and the host distributes by any means (the Python side uses
 * torch.distributed.broadcast_object_list).
 *
 * Conventions are those of pq_hip.h: caller-owned device buffers, stream-ordered, status codes (0 = OK,
 * 1 bad argument, 5 workspace too small, 6 RCCL error), message via pq_rccl_last_error().
 */
#ifndef PQ_RCCL_H
#define PQ_RCCL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PQ_RCCL_UNIQUE_ID_BYTES 128

const char* pq_rccl_last_error(void);
/* rank 0: fill id[128] (ncclGetUniqueId). */
int32_t pq_comm_unique_id(void* id);
/* every rank, after cudaSetDevice: *comm = ncclCommInitRank(nranks, id, rank). Blocks until all ranks arrive. */
int32_t pq_comm_init_rank(void** comm, int32_t nranks, const void* id, int32_t rank);
int32_t pq_comm_destroy(void* comm);

/* y_shard[M, n_shard] (row-major, contiguous) on every rank  ->  y_full[M, nranks * n_shard] on every rank.
 * An all-gather concatenates along the OUTERMOST axis, so the shards first land stacked [nranks, M, n_shard] in
 * `workspace` (ncclAllGather), then one coalesced kernel interleaves them into y_full's columns.
 * dtype: 0 bf16, 1 fp16, 2 f32.  workspace >= pq_allgather_cols_workspace_bytes(). */
size_t pq_allgather_cols_workspace_bytes(int32_t nranks, int64_t M, int64_t n_shard, int32_t dtype);
int32_t pq_allgather_cols(void* comm, int32_t nranks, const void* y_shard, void* y_full, int64_t M, int64_t n_shard,
                          int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The same exchange for ANY shard split and leading dimensions: rank r owns the columns shard_bounds(n_total, nranks, r) =
 * the balanced contiguous split whose first n_total % nranks ranks get one extra column (protoquant_amd/sharded.py);
 * y_shard[M, width_r] with leading dimension ld_shard, y_full[M, n_total] with leading dimension ld_full.  Ragged or strided
 * shards are packed to [M, ceil(n_total / nranks)] first (one extra kernel; the pad is never read).
 * workspace >= pq_allgather_cols_v_workspace_bytes() = (nranks + 1) * M * ceil(n_total / nranks) * sizeof(dtype). */
size_t pq_allgather_cols_v_workspace_bytes(int32_t nranks, int64_t M, int64_t n_total, int32_t dtype);
int32_t pq_allgather_cols_v(void* comm, const void* y_shard, int64_t ld_shard, void* y_full, int64_t ld_full, int64_t M,
                            int64_t n_total, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Overlapped form: the exchange of the row block [m0, m1) runs on the communicator's own side stream, ordered behind
 * everything already enqueued on compute_stream (the block's GEMM), while the caller goes on launching the next block's
 * GEMM on compute_stream.  No allocation per call: the side stream and its events are created with the communicator,
 * every block uses its own slice of the ONE workspace sized for all M rows.  pq_comm_join() makes compute_stream wait for
 * every exchange issued so far.  xGMI is point-to-point (7 links x ~153 GB/s): at 70B shapes a layer's gather (~190 us) is
 * longer than its GEMM, so only the last block's transfer stays exposed. */
int32_t pq_allgather_cols_rows_async(void* comm, const void* y_shard, int64_t ld_shard, void* y_full, int64_t ld_full,
                                     int64_t M, int64_t m0, int64_t m1, int64_t n_total, int32_t dtype, void* workspace,
                                     size_t workspace_bytes, void* compute_stream);
int32_t pq_comm_join(void* comm, void* compute_stream);
/* Transposed shards (SURVEY.md §8(e) option 1): every rank computed yT_shard[width_r, M] = its rows of y^T (pq_qlinear_s8_t);
 * row blocks are contiguous, so ONE ncclAllGather (equal shards) or one group of ncclBroadcasts (ragged) builds
 * yT_full[n_total, M] in place — no staging, no layout kernel. */
int32_t pq_allgather_rows_t(void* comm, const void* yt_shard, void* yt_full, int64_t n_total, int64_t M, int32_t dtype, void* stream);
/* The int8-code exchange of the column-sharded gated MLP (BASELINE config 5, gate/up -> down; pq_hip.h: pq_silu_mul_rowamax / pq_silu_mul_quant_rowwise_amax /
 * pq_qlinear_s8_kslabs): an exact MAX of the ranks' row-amax bit patterns, in place (ncclAllReduce, ncclUint32, ncclMax — non-negative floats and NaNs order as
 * unsigned integers), and a plain contiguous all-gather of `bytes` bytes per rank into stacked[nranks][bytes] (the ranks' int8 code blocks [M, I / G]; the
 * consumer walks the stacked blocks in place, so there is no layout pass and no staging buffer). */
int32_t pq_allreduce_max_u32(void* comm, uint32_t* buf, int64_t count, void* stream);
int32_t pq_allgather_bytes(void* comm, const void* shard, void* stacked, int64_t bytes, void* stream);
/* ranks of the communicator as RCCL reports them (ncclCommCount) */
int32_t pq_comm_count(void* comm, int32_t* nranks);

/* the layout-fix kernel alone (stacked[nranks, M, n_shard] -> y_full[M, nranks*n_shard]); exposed for tests */
int32_t pq_unstack_cols(const void* stacked, void* y_full, int32_t nranks, int64_t M, int64_t n_shard, int32_t dtype,
                        void* stream);

/* the ragged layout kernel alone: stacked[nranks, M, ceil(n_total / nranks)] (rank r's block holds its shard_bounds() width,
 * the rest is padding) -> y_full[M, n_total] with leading dimension ld_full; exposed for tests */
int32_t pq_unstack_cols_v(const void* stacked, void* y_full, int64_t ld_full, int32_t nranks, int64_t M, int64_t n_total,
                          int32_t dtype, void* stream);

/* Row-sharded (K-split) qlinear, the Megatron pairing of a column-sharded producer: every rank holds
 * partial[nranks * m_shard, N] in f32 — its K-slice's contribution to the whole output — and receives the SUM over ranks of
 * row block `rank`: y_rows[m_shard, N], cast (RNE) to out_dtype.  One ncclReduceScatter (f32, sum) — row blocks are
 * contiguous, so there is no layout pass — plus a cast kernel when out_dtype is 16-bit (the f32 sum lands in `workspace`).
 * Summation order across ranks is RCCL's: exact for nranks <= 2, within (nranks - 1) f32 ulps otherwise. */
size_t pq_reduce_scatter_rows_workspace_bytes(int32_t nranks, int64_t m_shard, int64_t N, int32_t out_dtype);
int32_t pq_reduce_scatter_rows(void* comm, int32_t nranks, const float* partial, void* y_rows, int64_t m_shard, int64_t N,
                               int32_t out_dtype, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
