// Dev microbenchmark: MFMA (16x16x64 i8) fed from LDS at the two candidate ratios, random data, no global traffic:
//   A) 8 waves/CU, wave tile 128x64:  64 MFMA + 24 ds_read_b128 per K-step (the shipped kernel)
//   B) 4 waves/CU, wave tile 128x128: 128 MFMA + 32 ds_read_b128 per K-step (candidate)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NP, int NQ, int THREADS, bool AGPR = false>   // NP x NQ 16x16 tiles per wave, 2 k-steps of 64
__global__ __launch_bounds__(THREADS) void k(const v4i* __restrict__ in, int* __restrict__ out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += THREADS) reinterpret_cast<v4i*>(lds)[i] = in[i];
    __syncthreads();
    v4i acc[NP][NQ];
    for (int i = 0; i < NP; ++i) for (int j = 0; j < NQ; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    const unsigned char* base = lds + ((lane & 15) * 128 + (((lane >> 4)) ^ ((lane & 15) >> 1)) * 16) + (tid >> 6) * 2048;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned char* b = base + (it & 7) * 4096;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i fp[NP], fq[NQ];
#pragma unroll
            for (int i = 0; i < NP; ++i) fp[i] = *reinterpret_cast<const v4i*>(b + ((i * 2048 + ks * 64) & 32767));
#pragma unroll
            for (int j = 0; j < NQ; ++j) fq[j] = *reinterpret_cast<const v4i*>(b + 32768 + ((j * 2048 + (ks * 64 ^ 64)) & 28671));
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if constexpr (AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fp[i]), "v"(fq[j]));
                    else acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fp[i], fq[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    if constexpr (AGPR) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < NP; ++i) for (int j = 0; j < NQ; ++j) s ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
    if (s == 0x7fffffff) out[tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int NP, int NQ, int THREADS, bool AGPR = false> void run(const char* name, const v4i* din, int* dout, unsigned long long* dclk, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 300; ++w) k<NP, NQ, THREADS, AGPR><<<256, THREADS>>>(din, dout, iters, dclk);
    (void)hipEventRecord(a);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) k<NP, NQ, THREADS, AGPR><<<256, THREADS>>>(din, dout, iters, dclk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(512); (void)hipMemcpy(c.data(), dclk, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
    const double nm = 256.0 * (THREADS / 64) * iters * 2.0 * NP * NQ, us = ms * 1e3 / reps;
    printf("%-52s %.1f us  %.0f TOPS (%.1f%% of 5033)  clock %.3f GHz\n", name, us, nm * 32768.0 / us / 1e6, nm * 32768.0 / us / 1e6 / 50.33, (cyc / rt) * 0.1);
}
int main() {
    v4i* din; int* dout; unsigned long long* dclk;
    (void)hipMalloc(&din, 65536); (void)hipMalloc(&dout, 4096); (void)hipMalloc(&dclk, 512 * 8);
    std::vector<int> h(16384);
    for (int mode = 0; mode < 3; ++mode) {
        for (auto& v : h) {
            if (mode == 0) v = 0;
            else if (mode == 1) v = rand() ^ (rand() << 16);
            else { int q = 0; for (int b = 0; b < 4; ++b) { double g = 0; for (int t = 0; t < 12; ++t) g += rand() / (double)RAND_MAX; int c = (int)lrint((g - 6.0) * 28.0); c = c > 127 ? 127 : (c < -127 ? -127 : c); q |= (c & 0xff) << (8 * b); } v = q; }
        }
        (void)hipMemcpy(din, h.data(), 65536, hipMemcpyHostToDevice);
        printf("== operands: %s\n", mode == 0 ? "zero" : mode == 1 ? "uniform random bytes" : "gaussian int8 codes (sigma 28)");
        run<8, 4, 512>("A) 8 waves, 128x64/wave: 64 MFMA + 24 ds_read", din, dout, dclk, 1500);
        run<8, 8, 256>("B) 4 waves, 128x128/wave: 128 MFMA + 32 ds_read", din, dout, dclk, 1500);
        run<8, 8, 256, true>("B2) same, asm MFMA with AGPR accumulators", din, dout, dclk, 1500);
        run<8, 4, 512, true>("A2) 8 waves, asm MFMA with AGPR accumulators", din, dout, dclk, 1500);
    }
    return 0;
}
