// Dev microbenchmark: per-CU ingest rate from L2-resident data: LDS-DMA (global_load_lds_dwordx4) vs
// plain global_load_dwordx4 -> VGPR.  Every CU streams the same SHARED bytes (L2/MALL hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int MODE>   // 0: LDS-DMA, 1: VGPR loads, 2: VGPR loads fragment-shaped (16 rows x 64 B per instr)
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ src, size_t bytes, int iters, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // each block streams a window that depends on blockIdx%8-group so that an XCD's CUs share lines
    const size_t base = (size_t)(blockIdx.x / 8 % 4) * 65536;
    v4u acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const size_t off = (base + (size_t)it * 262144) % bytes;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + off + (w * 8 + p) * 1024 + lane * 16), (lptr_t)(lds + (w * 8 + p) * 1024), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0070);
        } else if constexpr (MODE == 1) {
            v4u v[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const v4u*>(src + off + (w * 8 + p) * 1024 + lane * 16);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc ^= v[p];
        } else {
            v4u v[8];   // rows of 128 B; lane -> row (lane&15), 16-B chunk (lane>>4) + 4*(p&1): two instrs complete a line
#pragma unroll
            for (int p = 0; p < 8; ++p)
                v[p] = *reinterpret_cast<const v4u*>(src + off + (w * 8) * 1024 + ((p >> 1) * 16 + (lane & 15)) * 128 + ((lane >> 4) + 4 * (p & 1)) * 16);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc ^= v[p];
        }
    }
    if (MODE == 0) { __syncthreads(); acc[0] = lds[tid]; }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) out[tid] = acc[0];
}

template <int MODE> void run(const char* name, const unsigned char* d, size_t bytes, unsigned* out) {
    const int iters = 2000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) k<MODE><<<256, 512>>>(d, bytes, iters, out);
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k<MODE><<<256, 512>>>(d, bytes, iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double by = 256.0 * 65536.0 * iters * 10;
    printf("%-44s %.2f TB/s  (%.1f GB/s per CU)\n", name, by / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 256 / 1e9);
}
int main() {
    const size_t bytes = 8u << 20;   // 8 MiB shared window: L2 + Infinity-Cache resident
    unsigned char* d; unsigned* out; (void)hipMalloc(&d, bytes + (1 << 20)); (void)hipMalloc(&out, 4096);
    (void)hipMemset(d, 1, bytes + (1 << 20));
    run<0>("LDS-DMA global_load_lds_dwordx4", d, bytes, out);
    run<1>("global_load_dwordx4 -> VGPR (1 KiB/instr)", d, bytes, out);
    run<2>("global_load_dwordx4 -> VGPR (16 rows x 64 B)", d, bytes, out);
    return 0;
}
