// pq_rccl.hip — libpq_rccl.so (include/pq_rccl.h): RCCL all-gather of column shards + layout fix (equal or ragged shards,
// whole or row-chunked on a side stream so a chunk's transfer overlaps the next chunk's GEMM), the contiguous all-gather of
// TRANSPOSED shards (no layout pass), and the reduce-scatter of row-sharded partial outputs + cast.
// Kept apart from libpq_hip.so so the compute library has no communication dependency.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>

#include "../../include/pq_rccl.h"

namespace {
thread_local char g_err[512] = "";
int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
int elem_bytes(int32_t dtype) { return dtype == 2 ? 4 : 2; }

// PQ_ROCTX=1: roctx ranges around the exchange entry points (as in libpq_hip.so); read once
struct Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const Roctx& roctx() {
    static Roctx r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("PQ_ROCTX");
        if (!e || !*e || *e == '0') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        r.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!r.push || !r.pop) r = Roctx{};
    });
    return r;
}
struct Range {
    bool on;
    explicit Range(const char* n) : on(roctx().push != nullptr) { if (on) roctx().push(n); }
    ~Range() { if (on) roctx().pop(); }
};

// The handle behind the void* of the ABI: the RCCL communicator plus what the overlapped exchange needs — a side stream
// and two events, created once at init so that no call allocates.
struct PqComm {
    ncclComm_t c = nullptr;
    int nranks = 0, rank = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_compute = nullptr, ev_side = nullptr;
};
PqComm* H(void* p) { return static_cast<PqComm*>(p); }

// contiguous, balanced split of n over `world` parts: the first n % world parts get one extra (sharded.py: shard_bounds)
__host__ __device__ inline void bounds(int64_t n, int world, int r, int64_t* lo, int64_t* hi) {
    const int64_t q = n / world, rem = n % world;
    *lo = r * q + (r < rem ? r : rem);
    *hi = *lo + q + (r < rem ? 1 : 0);
}

// ragged forms, 2-byte units (every dtype is a multiple): pack[m][c] = shard[m * ld + c] for c < width (the pad is never read);
// out[m * ld_out + lo_r + c] = stacked[(r * rows + m) * n_max + c] for c < width_r.
__global__ __launch_bounds__(256) void pack_rows_kernel(const unsigned short* __restrict__ shard, int64_t ld, unsigned short* __restrict__ pack,
                                                        int64_t rows, int64_t width, int64_t n_max) {
    const int64_t total = rows * width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % width, m = i / width;
        pack[m * n_max + c] = shard[m * ld + c];
    }
}
__global__ __launch_bounds__(256) void unstack_ragged_kernel(const unsigned short* __restrict__ stacked, unsigned short* __restrict__ out, int nranks,
                                                             int64_t rows, int64_t n_total_u, int64_t n_max_u, int64_t ld_out_u, int unit) {
    // n_total_u etc. in 2-byte units; shard bounds are computed in ELEMENTS (unit = 2-byte units per element) so that they
    // match the host's split
    const int64_t total = (int64_t)nranks * rows * n_max_u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % n_max_u, m = (i / n_max_u) % rows;
        const int r = (int)(i / (n_max_u * rows));
        int64_t lo, hi;
        bounds(n_total_u / unit, nranks, r, &lo, &hi);
        if (c < (hi - lo) * unit) out[m * ld_out_u + lo * unit + c] = stacked[i];
    }
}

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

namespace {
int64_t grid_for(int64_t units) {
    int64_t blocks = (units + 255) / 256;
    return blocks > 256 * 8 ? 256 * 8 : (blocks < 1 ? 1 : blocks);
}
bool launch_ok(const char* what, int32_t* rc) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return true;
    *rc = fail(3, "%s: %s", what, hipGetErrorString(e));
    return false;
}

// One exchange of rows [0, rows) of the column shards on stream st: (pack if ragged / strided) -> ncclAllGather into the
// stacked receive region -> layout kernel into y_full's columns.  ws: (nranks + 1) * rows * n_max * eb bytes.
int32_t gather_cols_on(PqComm* h, const void* y_shard, int64_t ld_shard, void* y_full, int64_t ld_full, int64_t rows,
                       int64_t n_total, int32_t dtype, void* ws, hipStream_t st) {
    const int eb = elem_bytes(dtype), G = h->nranks;
    const int64_t n_max = (n_total + G - 1) / G;
    int64_t lo, hi;
    bounds(n_total, G, h->rank, &lo, &hi);
    const int64_t width = hi - lo;
    if (rows == 0 || n_total == 0) return 0;
    unsigned char* send = static_cast<unsigned char*>(ws);
    unsigned char* recv = send + (size_t)rows * (size_t)n_max * eb;
    const void* src = y_shard;
    int32_t rc = 0;
    if (width != n_max || ld_shard != width) {           // ragged or strided shard: pack it to [rows, n_max]
        const int unit = eb / 2;
        pack_rows_kernel<<<dim3((unsigned)grid_for(rows * width * unit)), dim3(256), 0, st>>>(
            static_cast<const unsigned short*>(y_shard), ld_shard * unit, reinterpret_cast<unsigned short*>(send), rows, width * unit, n_max * unit);
        if (!launch_ok("pack launch", &rc)) return rc;
        src = send;
    }
    const size_t per_rank = (size_t)rows * (size_t)n_max * eb;
    const ncclResult_t r = ncclAllGather(src, recv, per_rank, ncclInt8, h->c, st);     // bytes on the wire: dtype-agnostic
    if (r != ncclSuccess) return fail(6, "ncclAllGather: %s", ncclGetErrorString(r));
    if (n_total % G == 0 && ld_full == n_total) return launch_unstack(recv, y_full, G, rows, n_max, dtype, st);
    const int unit = eb / 2;
    unstack_ragged_kernel<<<dim3((unsigned)grid_for((int64_t)G * rows * n_max * unit)), dim3(256), 0, st>>>(
        reinterpret_cast<const unsigned short*>(recv), static_cast<unsigned short*>(y_full), G, rows, n_total * unit, n_max * unit, ld_full * unit, unit);
    launch_ok("unstack launch", &rc);
    return rc;
}
size_t gather_ws_bytes(int nranks, int64_t rows, int64_t n_total, int32_t dtype) {
    const int64_t n_max = (n_total + nranks - 1) / nranks;
    return (size_t)(nranks + 1) * (size_t)rows * (size_t)n_max * (size_t)elem_bytes(dtype);
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
    PqComm* h = new PqComm();
    const ncclResult_t r = ncclCommInitRank(&h->c, nranks, u, rank);
    if (r != ncclSuccess) { delete h; return fail(6, "ncclCommInitRank: %s", ncclGetErrorString(r)); }
    h->nranks = nranks; h->rank = rank;
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_compute, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming) != hipSuccess) {
        ncclCommDestroy(h->c); delete h;
        return fail(3, "pq_comm_init_rank: stream / event creation failed");
    }
    *comm = h;
    return 0;
}

int32_t pq_comm_count(void* comm, int32_t* nranks) {
    if (!comm || !nranks) return fail(1, "pq_comm_count: null argument");
    int n = 0;
    const ncclResult_t r = ncclCommCount(H(comm)->c, &n);
    if (r != ncclSuccess) return fail(6, "ncclCommCount: %s", ncclGetErrorString(r));
    *nranks = n;
    return 0;
}

int32_t pq_comm_destroy(void* comm) {
    if (!comm) return 0;
    PqComm* h = H(comm);
    const ncclResult_t r = ncclCommDestroy(h->c);
    if (h->ev_compute) (void)hipEventDestroy(h->ev_compute);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    if (h->side) (void)hipStreamDestroy(h->side);
    delete h;
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

int32_t pq_unstack_cols_v(const void* stacked, void* y_full, int64_t ld_full, int32_t nranks, int64_t M, int64_t n_total, int32_t dtype, void* stream) {
    if (nranks < 1 || M < 0 || n_total < 0 || dtype < 0 || dtype > 2 || ld_full < n_total || ((M > 0 && n_total > 0) && (!stacked || !y_full)))
        return fail(1, "pq_unstack_cols_v: bad arguments");
    if (M == 0 || n_total == 0) return 0;
    const int unit = elem_bytes(dtype) / 2;
    const int64_t n_max = (n_total + nranks - 1) / nranks;
    unstack_ragged_kernel<<<dim3((unsigned)grid_for((int64_t)nranks * M * n_max * unit)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const unsigned short*>(stacked), static_cast<unsigned short*>(y_full), nranks, M, n_total * unit, n_max * unit, ld_full * unit, unit);
    int32_t rc = 0;
    launch_ok("unstack launch", &rc);
    return rc;
}

int32_t pq_allgather_cols(void* comm, int32_t nranks, const void* y_shard, void* y_full, int64_t M, int64_t n_shard,
                          int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    Range range_("pq:allgather_cols");
    if (!comm || nranks < 1 || M < 0 || n_shard < 0 || dtype < 0 || dtype > 2) return fail(1, "pq_allgather_cols: bad arguments");
    if (nranks != H(comm)->nranks) return fail(1, "pq_allgather_cols: nranks %d != communicator's %d", nranks, H(comm)->nranks);
    const size_t need = pq_allgather_cols_workspace_bytes(nranks, M, n_shard, dtype);
    if (need == 0) return 0;
    if (!y_shard || !y_full || !workspace) return fail(1, "pq_allgather_cols: null buffer");
    if (workspace_bytes < need) return fail(5, "pq_allgather_cols: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const ncclResult_t r = ncclAllGather(y_shard, workspace, need / (size_t)nranks, ncclInt8, H(comm)->c, st);
    if (r != ncclSuccess) return fail(6, "ncclAllGather: %s", ncclGetErrorString(r));
    return launch_unstack(workspace, y_full, nranks, M, n_shard, dtype, st);
}

size_t pq_allgather_cols_v_workspace_bytes(int32_t nranks, int64_t M, int64_t n_total, int32_t dtype) {
    if (nranks < 1 || M < 0 || n_total < 0 || dtype < 0 || dtype > 2) return 0;
    return gather_ws_bytes(nranks, M, n_total, dtype);
}

int32_t pq_allgather_cols_v(void* comm, const void* y_shard, int64_t ld_shard, void* y_full, int64_t ld_full, int64_t M,
                            int64_t n_total, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
    Range range_("pq:allgather_cols_v");
    if (!comm || M < 0 || n_total < 0 || dtype < 0 || dtype > 2 || ld_full < n_total) return fail(1, "pq_allgather_cols_v: bad arguments");
    PqComm* h = H(comm);
    if (M == 0 || n_total == 0) return 0;
    if (!y_shard || !y_full || !workspace) return fail(1, "pq_allgather_cols_v: null buffer");
    const size_t need = gather_ws_bytes(h->nranks, M, n_total, dtype);
    if (workspace_bytes < need) return fail(5, "pq_allgather_cols_v: workspace %zu < %zu bytes", workspace_bytes, need);
    return gather_cols_on(h, y_shard, ld_shard, y_full, ld_full, M, n_total, dtype, workspace, static_cast<hipStream_t>(stream));
}

int32_t pq_allgather_cols_rows_async(void* comm, const void* y_shard, int64_t ld_shard, void* y_full, int64_t ld_full,
                                     int64_t M, int64_t m0, int64_t m1, int64_t n_total, int32_t dtype, void* workspace,
                                     size_t workspace_bytes, void* compute_stream) {
    if (!comm || M < 0 || m0 < 0 || m1 < m0 || m1 > M || n_total < 0 || dtype < 0 || dtype > 2 || ld_full < n_total)
        return fail(1, "pq_allgather_cols_rows_async: bad arguments");
    PqComm* h = H(comm);
    if (m1 == m0 || n_total == 0) return 0;
    if (!y_shard || !y_full || !workspace) return fail(1, "pq_allgather_cols_rows_async: null buffer");
    if (workspace_bytes < gather_ws_bytes(h->nranks, M, n_total, dtype)) return fail(5, "pq_allgather_cols_rows_async: workspace too small");
    const int eb = elem_bytes(dtype);
    // the chunk's slice of the workspace: chunks of one exchange never overlap, so any number may be in flight
    unsigned char* ws = static_cast<unsigned char*>(workspace) + gather_ws_bytes(h->nranks, m0, n_total, dtype);
    hipStream_t cs = static_cast<hipStream_t>(compute_stream);
    // side stream runs behind everything the compute stream has been given so far (this chunk's GEMM included) ...
    if (hipEventRecord(h->ev_compute, cs) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_compute, 0) != hipSuccess)
        return fail(3, "pq_allgather_cols_rows_async: event record / wait failed");
    const unsigned char* src = static_cast<const unsigned char*>(y_shard) + (size_t)m0 * (size_t)ld_shard * eb;
    unsigned char* dst = static_cast<unsigned char*>(y_full) + (size_t)m0 * (size_t)ld_full * eb;
    // ... while the caller's next launches on the compute stream run beside it
    return gather_cols_on(h, src, ld_shard, dst, ld_full, m1 - m0, n_total, dtype, ws, h->side);
}

int32_t pq_comm_join(void* comm, void* compute_stream) {
    if (!comm) return fail(1, "pq_comm_join: null communicator");
    PqComm* h = H(comm);
    if (hipEventRecord(h->ev_side, h->side) != hipSuccess ||
        hipStreamWaitEvent(static_cast<hipStream_t>(compute_stream), h->ev_side, 0) != hipSuccess)
        return fail(3, "pq_comm_join: event record / wait failed");
    return 0;
}

int32_t pq_allgather_rows_t(void* comm, const void* yt_shard, void* yt_full, int64_t n_total, int64_t M, int32_t dtype, void* stream) {
    Range range_("pq:allgather_rows_t");
    if (!comm || n_total < 0 || M < 0 || dtype < 0 || dtype > 2) return fail(1, "pq_allgather_rows_t: bad arguments");
    PqComm* h = H(comm);
    if (n_total == 0 || M == 0) return 0;
    if (!yt_shard || !yt_full) return fail(1, "pq_allgather_rows_t: null buffer");
    const int eb = elem_bytes(dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_total % h->nranks == 0) {      // equal shards: one all-gather, the result IS y^T[N, M], contiguous
        const ncclResult_t r = ncclAllGather(yt_shard, yt_full, (size_t)(n_total / h->nranks) * (size_t)M * eb, ncclInt8, h->c, st);
        return r == ncclSuccess ? 0 : fail(6, "ncclAllGather: %s", ncclGetErrorString(r));
    }
    // ragged: an all-gather-v as one group of broadcasts (row blocks of y^T are contiguous, so every rank's block lands in place)
    ncclResult_t r = ncclGroupStart();
    for (int root = 0; root < h->nranks && r == ncclSuccess; ++root) {
        int64_t lo, hi;
        bounds(n_total, h->nranks, root, &lo, &hi);
        unsigned char* dst = static_cast<unsigned char*>(yt_full) + (size_t)lo * (size_t)M * eb;
        r = ncclBroadcast(root == h->rank ? yt_shard : dst, dst, (size_t)(hi - lo) * (size_t)M * eb, ncclInt8, root, h->c, st);
    }
    const ncclResult_t e = ncclGroupEnd();
    if (r != ncclSuccess || e != ncclSuccess) return fail(6, "ncclBroadcast group: %s", ncclGetErrorString(r != ncclSuccess ? r : e));
    return 0;
}

int32_t pq_allreduce_max_u32(void* comm, uint32_t* buf, int64_t count, void* stream) {
    Range range_("pq:allreduce_max_u32");
    if (!comm || count < 0) return fail(1, "pq_allreduce_max_u32: bad arguments");
    if (count == 0) return 0;
    if (!buf) return fail(1, "pq_allreduce_max_u32: null buffer");
    const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclUint32, ncclMax, H(comm)->c, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : fail(6, "ncclAllReduce: %s", ncclGetErrorString(r));
}

int32_t pq_allgather_bytes(void* comm, const void* shard, void* stacked, int64_t bytes, void* stream) {
    Range range_("pq:allgather_bytes");
    if (!comm || bytes < 0) return fail(1, "pq_allgather_bytes: bad arguments");
    if (bytes == 0) return 0;
    if (!shard || !stacked) return fail(1, "pq_allgather_bytes: null buffer");
    const ncclResult_t r = ncclAllGather(shard, stacked, (size_t)bytes, ncclInt8, H(comm)->c, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : fail(6, "ncclAllGather: %s", ncclGetErrorString(r));
}

size_t pq_reduce_scatter_rows_workspace_bytes(int32_t nranks, int64_t m_shard, int64_t N, int32_t out_dtype) {
    if (nranks < 1 || m_shard < 0 || N < 0 || out_dtype < 0 || out_dtype > 2) return 0;
    return out_dtype == 2 ? 0 : (size_t)m_shard * (size_t)N * sizeof(float);
}

int32_t pq_reduce_scatter_rows(void* comm, int32_t nranks, const float* partial, void* y_rows, int64_t m_shard, int64_t N,
                               int32_t out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    Range range_("pq:reduce_scatter_rows");
    if (!comm || nranks < 1 || m_shard < 0 || N < 0 || out_dtype < 0 || out_dtype > 2) return fail(1, "pq_reduce_scatter_rows: bad arguments");
    const size_t count = (size_t)m_shard * (size_t)N;
    if (count == 0) return 0;
    if (!partial || !y_rows) return fail(1, "pq_reduce_scatter_rows: null buffer");
    const size_t need = pq_reduce_scatter_rows_workspace_bytes(nranks, m_shard, N, out_dtype);
    if (need && (!workspace || workspace_bytes < need)) return fail(5, "pq_reduce_scatter_rows: workspace %zu < %zu bytes", workspace ? workspace_bytes : (size_t)0, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* sum = out_dtype == 2 ? static_cast<float*>(y_rows) : static_cast<float*>(workspace);
    const ncclResult_t r = ncclReduceScatter(partial, sum, count, ncclFloat, ncclSum, H(comm)->c, st);
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
