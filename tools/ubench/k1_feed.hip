// Dev microbenchmark: what a 50-MB read-2-write-1 transfer costs on this chip depending on WHERE its bytes are (the ceiling K1 is held to).
// Feeds: l2 = one input replayed; mall = two alternating inputs (96 MB working set); hbm = 13 rotating input/output pairs (624 MB).
// Kernels: copy-like (no reduction, no arithmetic; default and write-through stores; grid-stride and one-row-per-wave shapes) and the product K1.
#include "../../protoquant_amd/csrc/quant_kernels.hip"
#include <cstdio>
#include <vector>
using namespace pq;

template <bool WT>
__global__ __launch_bounds__(256) void copy_like(const v4u* __restrict__ x, v2u* __restrict__ q, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const v4u v = x[i];
        const v2u o = v2u{__builtin_amdgcn_perm(v[1], v[0], 0x07050301u), __builtin_amdgcn_perm(v[3], v[2], 0x07050301u)};
        if constexpr (WT) store_wt_b64(q + i, o); else q[i] = o;
    }
}
// K1's access shape without its arithmetic: one wave per 8-KiB row, 8 loads up front, then 8 stores
__global__ __launch_bounds__(256) void copy_rows(const uint8_t* __restrict__ x, uint8_t* __restrict__ q, int64_t rows) {
    const int t = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    v4u v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const v4u*>(x + row * 8192 + (int64_t)(i * 64 + t) * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        store_wt_b64(q + row * 4096 + (int64_t)(i * 64 + t) * 8, v2u{__builtin_amdgcn_perm(v[i][1], v[i][0], 0x07050301u), __builtin_amdgcn_perm(v[i][3], v[i][2], 0x07050301u)});
}

// the same shape with 16-BYTE stores: adjacent lanes swap halves (DPP quad_perm [1,0,3,2]) so that an even lane holds 16 consecutive codes of
// segment i and its odd neighbour 16 consecutive codes of segment i+1; loads stay dense (1 KiB per wave-instruction)
__global__ __launch_bounds__(256) void copy_rows16(const uint8_t* __restrict__ x, uint8_t* __restrict__ q, int64_t rows) {
    const int t = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    v4u v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const v4u*>(x + row * 8192 + (int64_t)(i * 64 + t) * 16);
    const bool odd = t & 1;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const uint32_t a0 = __builtin_amdgcn_perm(v[i][1], v[i][0], 0x07050301u), a1 = __builtin_amdgcn_perm(v[i][3], v[i][2], 0x07050301u);
        const uint32_t b0 = __builtin_amdgcn_perm(v[i + 1][1], v[i + 1][0], 0x07050301u), b1 = __builtin_amdgcn_perm(v[i + 1][3], v[i + 1][2], 0x07050301u);
        const uint32_t na0 = __builtin_amdgcn_update_dpp(0u, a0, 0xB1, 0xF, 0xF, false), na1 = __builtin_amdgcn_update_dpp(0u, a1, 0xB1, 0xF, 0xF, false);
        const uint32_t nb0 = __builtin_amdgcn_update_dpp(0u, b0, 0xB1, 0xF, 0xF, false), nb1 = __builtin_amdgcn_update_dpp(0u, b1, 0xB1, 0xF, 0xF, false);
        const v4u o = odd ? v4u{nb0, nb1, b0, b1} : v4u{a0, a1, na0, na1};
        const int64_t seg = odd ? (i + 1) * 64 + (t - 1) : i * 64 + t;
        store_wt_b128(q + row * 4096 + seg * 8, o);
    }
}

int main() {
    const int64_t R = 4096, C = 4096;
    const int NB = 13;
    std::vector<uint16_t*> x(NB); std::vector<int8_t*> q(NB); std::vector<float*> sc(NB);
    std::vector<uint16_t> h(R * C);
    for (int b = 0; b < NB; ++b) {
        (void)hipMalloc(&x[b], R * C * 2); (void)hipMalloc(&q[b], R * C); (void)hipMalloc(&sc[b], R * 4);
        for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
        (void)hipMemcpy(x[b], h.data(), R * C * 2, hipMemcpyHostToDevice);
    }
    const double bytes = 3.0 * R * C + 4.0 * R;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto&& launch) {
        printf("%-44s", name);
        for (int nb : {1, 2, NB}) {
            for (int i = 0; i < 3 * NB; ++i) launch(i % nb);
            const int iters = 20 * NB;
            (void)hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) launch(i % nb);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters;
            printf("  %s %6.2f us %5.2f TB/s", nb == 1 ? "l2  " : (nb == 2 ? "mall" : "hbm "), us, bytes / us / 1e6);
        }
        printf("\n");
    };
    const int nvec = C / 8;
    for (int g : {1024, 2048, 4096}) {
        char nm[64]; snprintf(nm, sizeof nm, "copy-like grid-stride, grid %d", g);
        run(nm, [&](int b) { copy_like<false><<<g, 256>>>((const v4u*)x[b], (v2u*)q[b], R * C / 8); });
        snprintf(nm, sizeof nm, "copy-like grid-stride, grid %d, sc1 stores", g);
        run(nm, [&](int b) { copy_like<true><<<g, 256>>>((const v4u*)x[b], (v2u*)q[b], R * C / 8); });
    }
    run("copy, one wave per row (K1's shape), sc1", [&](int b) { copy_rows<<<R / 4, 256>>>((const uint8_t*)x[b], (uint8_t*)q[b], R); });
    run("copy, one wave per row, 16-B sc1 stores", [&](int b) { copy_rows16<<<R / 4, 256>>>((const uint8_t*)x[b], (uint8_t*)q[b], R); });
    run("K1 product (TPR 64, VPT 8)", [&](int b) { quant_rowwise_vec<PQ_BF16, 8, 64><<<R / 4, 256>>>((const uint8_t*)x[b], R, nvec, C * 2, q[b], C, sc[b]); });
    run("K1 256 threads per row (VPT 2)", [&](int b) { quant_rowwise_vec<PQ_BF16, 2, 256><<<R, 256>>>((const uint8_t*)x[b], R, nvec, C * 2, q[b], C, sc[b]); });
    return 0;
}
