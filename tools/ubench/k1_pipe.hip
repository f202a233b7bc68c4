// Dev microbenchmark (round 5): K1 with the read and write phases INTERLEAVED over time.  The product K1 runs one wave per row with every row of a 4096-row
// activation resident at once: the chip reads 32 MB, then writes 16 MB.  Here each wave walks several rows (grid-stride) and requests row i+1 before it encodes and
// stores row i (two register sets), so that loads and stores of different rows overlap at the memory side the way a grid-stride copy's do.
// Same arithmetic as the product (fast exact encode; NaN / extreme-scale rows are not handled here: timing + bit check on ordinary data only).
#include "../../protoquant_amd/csrc/quant_kernels.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace pq { const Options& opt() { static Options o; return o; } }
using namespace pq;

template <int DEPTH>   // rows in flight per wave: 2 = double-buffered
__global__ __launch_bounds__(256) void k1_pipe(const uint8_t* __restrict__ x, int64_t rows, int8_t* __restrict__ q, float* __restrict__ scale) {
    const int t = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    v4u cur[8], nxt[8];
    int64_t row = wave;
    if (row >= rows) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = *reinterpret_cast<const v4u*>(x + row * 8192 + (int64_t)(i * 64 + t) * 16);
    for (; row < rows; row += nwaves) {
        const int64_t nrow = row + nwaves < rows ? row + nwaves : row;      // (the last iteration re-reads its own row: harmless)
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = *reinterpret_cast<const v4u*>(x + nrow * 8192 + (int64_t)(i * 64 + t) * 16);
        uint32_t ab = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) ab = vec_amax_bits<PQ_BF16>(cur[i], ab);
        ab = wave_max_u32(amax_acc_finish<PQ_BF16>(ab));
        const float s = scale_of(amax_bits_to_f32<PQ_BF16>(ab));
        if (t == 0) scale[row] = s;
        const float r = 1.0f / s;
        int8_t* qr = q + row * 4096;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f[8];
            Unpack<PQ_BF16, 8>::run(cur[i], f);
            uint32_t pk[2];
            fast_encode<8, kQuotientSteps<PQ_BF16>>(f, s, r, pk);
            store_wt_b64(qr + (int64_t)(i * 64 + t) * 8, v2u{pk[0], pk[1]});
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
}

int main() {
    const int64_t R = 4096, C = 4096;
    const int NB = 13;
    std::vector<uint16_t*> x(NB); std::vector<int8_t*> q(NB), q2(NB); std::vector<float*> sc(NB), sc2(NB);
    std::vector<uint16_t> h(R * C);
    for (int b = 0; b < NB; ++b) {
        (void)hipMalloc(&x[b], R * C * 2); (void)hipMalloc(&q[b], R * C); (void)hipMalloc(&sc[b], R * 4); (void)hipMalloc(&q2[b], R * C); (void)hipMalloc(&sc2[b], R * 4);
        for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
        (void)hipMemcpy(x[b], h.data(), R * C * 2, hipMemcpyHostToDevice);
    }
    const int nvec = C / 8;
    // bit check against the product kernel
    quant_rowwise_vec<PQ_BF16, 8, 64><<<R / 4, 256>>>((const uint8_t*)x[0], R, nvec, C * 2, q[0], C, sc[0]);
    k1_pipe<2><<<512, 256>>>((const uint8_t*)x[0], R, q2[0], sc2[0]);
    (void)hipDeviceSynchronize();
    std::vector<int8_t> a(R * C), b(R * C); std::vector<float> sa(R), sb(R);
    (void)hipMemcpy(a.data(), q[0], R * C, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), q2[0], R * C, hipMemcpyDeviceToHost);
    (void)hipMemcpy(sa.data(), sc[0], R * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(sb.data(), sc2[0], R * 4, hipMemcpyDeviceToHost);
    printf("pipelined == product: codes %s, scales %s\n", memcmp(a.data(), b.data(), R * C) ? "DIFFER" : "same", memcmp(sa.data(), sb.data(), R * 4) ? "DIFFER" : "same");
    const double bytes = 3.0 * R * C + 4.0 * R;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto&& launch) {
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
    for (int rep = 0; rep < 2; ++rep) {
        run("K1 product (one wave per row, 1024 blocks)", [&](int b) { quant_rowwise_vec<PQ_BF16, 8, 64><<<R / 4, 256>>>((const uint8_t*)x[b], R, nvec, C * 2, q[b], C, sc[b]); });
        run("K1 product, 256 threads per row", [&](int b) { quant_rowwise_vec<PQ_BF16, 2, 256><<<R, 256>>>((const uint8_t*)x[b], R, nvec, C * 2, q[b], C, sc[b]); });
        for (int g : {256, 512, 768, 1024}) {
            char nm[80]; snprintf(nm, sizeof nm, "K1 pipelined over rows, %d blocks (%d rows per wave)", g, (int)(R / (g * 4)));
            run(nm, [&](int b) { k1_pipe<2><<<g, 256>>>((const uint8_t*)x[b], R, q[b], sc[b]); });
        }
    }
    return 0;
}
