// Dev microbenchmark (round 5, SURVEY.md:219 "amax-only pass + quantise-in-prologue"): what the amax-only pre-pass of that variant would cost — K1's access shape
// (one wave per 8-KiB bf16 row, every load issued up front) with the reduction and ONE 4-byte store per row instead of the encode and the 4096 code bytes —
// beside the product K1 in the same rounds, by where the bytes are (l2: one input replayed; mall: two alternating; hbm: 13 rotating inputs, 624 MB with K1's outputs).
#include "../../protoquant_amd/csrc/quant_kernels.hip"
#include <cstdio>
#include <vector>
namespace pq { const Options& opt() { static Options o; return o; } }      // (the library's switches: defaults)
using namespace pq;

__global__ __launch_bounds__(256) void amax_rows(const uint8_t* __restrict__ x, uint32_t* __restrict__ amax, int64_t rows) {
    const int t = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    v4u v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const v4u*>(x + row * 8192 + (int64_t)(i * 64 + t) * 16);
    uint32_t ab = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) ab = vec_amax_bits<PQ_BF16>(v[i], ab);
    ab = wave_max_u32(amax_acc_finish<PQ_BF16>(ab));
    if (t == 0) amax[row] = ab << 16;
}
// grid-stride form: 2048 blocks, each wave walks rows (more bytes in flight per CU at the start, no tail of short-lived workgroups)
__global__ __launch_bounds__(256) void amax_rows_gs(const uint8_t* __restrict__ x, uint32_t* __restrict__ amax, int64_t rows) {
    const int t = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        v4u v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const v4u*>(x + row * 8192 + (int64_t)(i * 64 + t) * 16);
        uint32_t ab = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) ab = vec_amax_bits<PQ_BF16>(v[i], ab);
        ab = wave_max_u32(amax_acc_finish<PQ_BF16>(ab));
        if (t == 0) amax[row] = ab << 16;
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
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, double bytes, auto&& launch) {
        printf("%-52s", name);
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
    for (int rep = 0; rep < 2; ++rep) {
        run("amax-only pass, one wave per row (reads 32 MiB)", 2.0 * R * C + 4.0 * R, [&](int b) { amax_rows<<<R / 4, 256>>>((const uint8_t*)x[b], (uint32_t*)sc[b], R); });
        run("amax-only pass, grid-stride over 512 blocks", 2.0 * R * C + 4.0 * R, [&](int b) { amax_rows_gs<<<512, 256>>>((const uint8_t*)x[b], (uint32_t*)sc[b], R); });
        run("K1 product (reads 32 MiB, writes 16 MiB)", 3.0 * R * C + 4.0 * R, [&](int b) { quant_rowwise_vec<PQ_BF16, 8, 64><<<R / 4, 256>>>((const uint8_t*)x[b], R, nvec, C * 2, q[b], C, sc[b]); });
    }
    return 0;
}
