// pq_rccl.hip — libpq_rccl.so (include/pq_rccl.h): RCCL all-gather of column shards + layout fix, and the
// reduce-scatter of row-sharded partial outputs + cast.
// Kept apart from libpq_hip.so so the compute library has no communication dependency.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/pq_rccl.h"

namespace {
thread_local char g_err[512] = "";
int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
int elem_bytes(int32_t dtype) { return dtype == 2 ? 4 : 2; }

// stacked[r][m][c] -> out[m][r * row_bytes + c], all in bytes; VEC = 16-byte moves when row_bytes % 16 == 0.
template <typename V>
__global__ __launch_bounds__(256) void unstack_kernel(const V* __restrict__ stacked, V* __restrict__ out, int nranks,
                                                      int64_t M, int64_t row_v) {   // row_v: shard row length in V units
    const int64_t total = (int64_t)nranks * M * row_v;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % row_v, m = (i / row_v) % M, r = i / (row_v * M);
        out[(m * nranks + r) * row_v + c] = stacked[i];
    }
}
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

int32_t launch_unstack(const void* stacked, void* y_full, int32_t nranks, int64_t M, int64_t n_shard, int32_t dtype, hipStream_t st) {
    const int64_t row_bytes = n_shard * elem_bytes(dtype);
    const int64_t total_bytes = (int64_t)nranks * M * row_bytes;
    if (total_bytes == 0) return 0;
    const bool vec = (row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(stacked) | reinterpret_cast<uintptr_t>(y_full)) % 16 == 0);
    const int64_t units = vec ? total_bytes / 16 : total_bytes / 2;
    int64_t blocks = (units + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (vec) unstack_kernel<v4u><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(reinterpret_cast<const v4u*>(stacked), reinterpret_cast<v4u*>(y_full), nranks, M, row_bytes / 16);
    else unstack_kernel<unsigned short><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(reinterpret_cast<const unsigned short*>(stacked), reinterpret_cast<unsigned short*>(y_full), nranks, M, row_bytes / 2);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(3, "unstack launch: %s", hipGetErrorString(e));
}
// f32 -> bf16 / fp16 / f32 (RNE) of a contiguous block; the cast after the f32 sum of row-sharded partial outputs
template <int DT>
__global__ __launch_bounds__(256) void cast_from_f32(const float* __restrict__ in, void* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = in[i];
        if constexpr (DT == 0) reinterpret_cast<__bf16*>(out)[i] = (__bf16)v;
        else if constexpr (DT == 1) reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
        else reinterpret_cast<float*>(out)[i] = v;
    }
}
}  // namespace

extern "C" {

const char* pq_rccl_last_error(void) { return g_err; }

int32_t pq_comm_unique_id(void* id) {
    if (!id) return fail(1, "pq_comm_unique_id: null id");
    ncclUniqueId u;
    const ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail(6, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    static_assert(sizeof(u) == PQ_RCCL_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof(u));
    return 0;
}

int32_t pq_comm_init_rank(void** comm, int32_t nranks, const void* id, int32_t rank) {
    if (!comm || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(1, "pq_comm_init_rank: bad arguments (nranks=%d rank=%d)", nranks, rank);
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, nranks, u, rank);
    if (r != ncclSuccess) return fail(6, "ncclCommInitRank: %s", ncclGetErrorString(r));
    *comm = c;
    return 0;
}

int32_t pq_comm_destroy(void* comm) {
    if (!comm) return 0;
    const ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(comm));
    return r == ncclSuccess ? 0 : fail(6, "ncclCommDestroy: %s", ncclGetErrorString(r));
}

size_t pq_allgather_cols_workspace_bytes(int32_t nranks, int64_t M, int64_t n_shard, int32_t dtype) {
    if (nranks < 1 || M < 0 || n_shard < 0 || dtype < 0 || dtype > 2) return 0;
    return (size_t)nranks * (size_t)M * (size_t)n_shard * (size_t)elem_bytes(dtype);
}

int32_t pq_unstack_cols(const void* stacked, void* y_full, int32_t nranks, int64_t M, int64_t n_shard, int32_t dtype, void* stream) {
    if (nranks < 1 || M < 0 || n_shard < 0 || dtype < 0 || dtype > 2 || ((M > 0 && n_shard > 0) && (!stacked || !y_full)))
        return fail(1, "pq_unstack_cols: bad arguments");
    return launch_unstack(stacked, y_full, nranks, M, n_shard, dtype, static_cast<hipStream_t>(stream));
}

int32_t pq_allgather_cols(void* comm, int32_t nranks, const void* y_shard, void* y_full, int64_t M, int64_t n_shard,
                          int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!comm || nranks < 1 || M < 0 || n_shard < 0 || dtype < 0 || dtype > 2) return fail(1, "pq_allgather_cols: bad arguments");
    const size_t need = pq_allgather_cols_workspace_bytes(nranks, M, n_shard, dtype);
    if (need == 0) return 0;
    if (!y_shard || !y_full || !workspace) return fail(1, "pq_allgather_cols: null buffer");
    if (workspace_bytes < need) return fail(5, "pq_allgather_cols: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // bytes on the wire: dtype-agnostic (ncclInt8)
    const ncclResult_t r = ncclAllGather(y_shard, workspace, need / (size_t)nranks, ncclInt8, static_cast<ncclComm_t>(comm), st);
    if (r != ncclSuccess) return fail(6, "ncclAllGather: %s", ncclGetErrorString(r));
    return launch_unstack(workspace, y_full, nranks, M, n_shard, dtype, st);
}

size_t pq_reduce_scatter_rows_workspace_bytes(int32_t nranks, int64_t m_shard, int64_t N, int32_t out_dtype) {
    if (nranks < 1 || m_shard < 0 || N < 0 || out_dtype < 0 || out_dtype > 2) return 0;
    return out_dtype == 2 ? 0 : (size_t)m_shard * (size_t)N * sizeof(float);
}

int32_t pq_reduce_scatter_rows(void* comm, int32_t nranks, const float* partial, void* y_rows, int64_t m_shard, int64_t N,
                               int32_t out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!comm || nranks < 1 || m_shard < 0 || N < 0 || out_dtype < 0 || out_dtype > 2) return fail(1, "pq_reduce_scatter_rows: bad arguments");
    const size_t count = (size_t)m_shard * (size_t)N;
    if (count == 0) return 0;
    if (!partial || !y_rows) return fail(1, "pq_reduce_scatter_rows: null buffer");
    const size_t need = pq_reduce_scatter_rows_workspace_bytes(nranks, m_shard, N, out_dtype);
    if (need && (!workspace || workspace_bytes < need)) return fail(5, "pq_reduce_scatter_rows: workspace %zu < %zu bytes", workspace ? workspace_bytes : (size_t)0, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* sum = out_dtype == 2 ? static_cast<float*>(y_rows) : static_cast<float*>(workspace);
    const ncclResult_t r = ncclReduceScatter(partial, sum, count, ncclFloat, ncclSum, static_cast<ncclComm_t>(comm), st);
    if (r != ncclSuccess) return fail(6, "ncclReduceScatter: %s", ncclGetErrorString(r));
    if (out_dtype != 2) {
        int64_t blocks = ((int64_t)count + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (out_dtype == 0) cast_from_f32<0><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(sum, y_rows, (int64_t)count);
        else cast_from_f32<1><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(sum, y_rows, (int64_t)count);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(3, "cast launch: %s", hipGetErrorString(e));
    }
    return 0;
}

}  // extern "C"
