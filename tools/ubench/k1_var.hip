// Dev microbenchmark: K1 configurations vs a copy-like ceiling (read 2 B/elem, write 1 B/elem, no row reduction).
#include "../../protoquant_amd/csrc/quant_kernels.hip"
#include <cstdio>
#include <vector>
using namespace pq;

__global__ __launch_bounds__(256) void copy_like(const v4u* __restrict__ x, v2u* __restrict__ q, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const v4u v = x[i];
        q[i] = v2u{__builtin_amdgcn_perm(v[1], v[0], 0x07050301u), __builtin_amdgcn_perm(v[3], v[2], 0x07050301u)};
    }
}

template <typename F> float time_us(F&& f, int iters = 200) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms * 1e3f / iters;
}
int main() {
    const int64_t R = 4096, C = 4096;
    uint16_t* x; int8_t* q; float* sc;
    (void)hipMalloc(&x, R * C * 2); (void)hipMalloc(&q, R * C); (void)hipMalloc(&sc, R * 4);
    std::vector<uint16_t> h(R * C); for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    (void)hipMemcpy(x, h.data(), R * C * 2, hipMemcpyHostToDevice);
    const double bytes = 3.0 * R * C + 4.0 * R;
    const uint8_t* xb = (const uint8_t*)x;
    for (int g : {512, 1024, 2048, 4096, 8192}) {
        float t = time_us([&] { copy_like<<<g, 256>>>((const v4u*)x, (v2u*)q, R * C / 8); });
        printf("copy-like grid=%5d: %.2f us  %.2f TB/s\n", g, t, bytes / t / 1e6);
    }
    const int nvec = C / 8;
#define PRD(V,T) { float t = time_us([&] { quant_rowwise_vec<PQ_BF16, V, T><<<R / (256 / T), 256>>>(xb, R, nvec, C * 2, q, C, sc); }); printf("K1 product TPR=%d VPT=%d: %.2f us %.2f TB/s\n", T, V, t, bytes / t / 1e6); }
    PRD(8,64) PRD(2,256)
    return 0;
}
