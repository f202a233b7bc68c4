// Dev microbenchmark: int8 MFMA issue ceiling on gfx950. No memory traffic in the loop.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512, 2) void k(const v4i* __restrict__ in, int* __restrict__ out, int iters, unsigned long long* clk) {
    using acc_t = typename std::conditional<SHAPE == 16, v4i, v16i>::type;
    acc_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < (SHAPE == 16 ? 4 : 16); ++r) acc[a][r] = 0;
    v4i fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = in[(threadIdx.x * 4 + i) % 4096]; fb[i] = in[(threadIdx.x * 4 + i + 77) % 4096]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if constexpr (SHAPE == 16) acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a & 3], fb[(a >> 2) & 3], acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a & 3], fb[(a >> 2) & 3], acc[a], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < (SHAPE == 16 ? 4 : 16); ++r) s ^= acc[a][r];
    if (s == 0x7fffffff) out[threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int NACC>
void run(const char* name, const v4i* din, int* dout, unsigned long long* dclk, int threads, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) k<SHAPE, NACC><<<256, threads>>>(din, dout, iters, dclk);
    hipDeviceSynchronize();
    // warm clocks ~1 s
    for (int w = 0; w < 200; ++w) k<SHAPE, NACC><<<256, threads>>>(din, dout, iters, dclk);
    hipEventRecord(a);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) k<SHAPE, NACC><<<256, threads>>>(din, dout, iters, dclk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(512); hipMemcpy(c.data(), dclk, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
    const double ops_per_mfma = (SHAPE == 16) ? 2.0 * 16 * 16 * 64 : 2.0 * 32 * 32 * 32;
    const double nm = 256.0 * (threads / 64) * iters * NACC;
    const double us = ms * 1e3 / reps;
    printf("%-34s threads=%d: %.1f us/launch  %.0f TOPS  in-kernel clock %.3f GHz  cycles/MFMA/SIMD %.2f\n", name, threads, us,
           nm * ops_per_mfma / us / 1e6, (cyc / rt) * 0.1, (cyc / 256) / (double(iters) * NACC * (threads / 256.0)));
}

int main() {
    v4i* din; int* dout; unsigned long long* dclk;
    hipMalloc(&din, 4096 * 16); hipMalloc(&dout, 4096); hipMalloc(&dclk, 512 * 8);
    std::vector<int> h(4096 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& v : h) v = mode ? rand() ^ (rand() << 16) : 0;
        hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        printf("== operands: %s\n", mode ? "random" : "zero");
        run<16, 16>("16x16x64 i8, 16 acc", din, dout, dclk, 512, 4000);
        run<16, 16>("16x16x64 i8, 16 acc", din, dout, dclk, 256, 8000);
        run<32, 4>("32x32x32 i8, 4 acc", din, dout, dclk, 512, 8000);
        run<32, 4>("32x32x32 i8, 4 acc", din, dout, dclk, 256, 16000);
    }
    return 0;
}
