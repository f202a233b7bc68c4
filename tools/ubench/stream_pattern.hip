// Dev microbenchmark (round 4): HBM streaming rate of a weight matrix [N, K] (row stride K bytes) by the shape of one load instruction:
//   mode 0: 16 rows x 64 B  (the v_mfma_i32_16x16x64_i8 operand layout: what gemm_s8_skinny issues)
//   mode 1:  8 rows x 128 B (whole 128-byte lines)
//   mode 2:  4 rows x 256 B
//   mode 3:  1 row  x 1 KiB
// Every wave owns a block of rows and walks K; 16 waves per workgroup split K like the skinny kernel's; 8 loads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(const unsigned char* __restrict__ w, long N, long K, unsigned* out) {
    constexpr int ROWS = MODE == 0 ? 16 : (MODE == 1 ? 8 : (MODE == 2 ? 4 : 1));
    constexpr int SPAN = 1024 / ROWS;                         // bytes per row per instruction
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, KS = blockDim.x >> 6;
    const int r = lane % ROWS, c = lane / ROWS;               // row of the block, 16-byte chunk of the span
    const long n0 = (long)blockIdx.x * 16;                    // every workgroup covers 16 rows (in 16 / ROWS passes of its instruction shape)
    const long k0 = K * wv / KS, k1 = K * (wv + 1) / KS;
    v4u acc = {0, 0, 0, 0};
    for (int pass = 0; pass < 16 / ROWS; ++pass) {
        const unsigned char* p = w + (n0 + pass * ROWS + r) * K + c * 16;
        for (long kk = k0; kk + 8 * SPAN <= k1; kk += 8 * SPAN) {
            v4u v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const v4u*>(p + kk + u * SPAN);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) out[threadIdx.x] = acc[0];
}

template <int MODE> void run(const char* name, const unsigned char* d, long N, long K, int ks, unsigned* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<MODE><<<(unsigned)(N / 16), ks * 64>>>(d, N, K, out);
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k<MODE><<<(unsigned)(N / 16), ks * 64>>>(d, N, K, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("  N=%6ld K=%6ld waves/WG=%2d  %-18s %6.2f TB/s  (%.1f us per pass)\n", N, K, ks, name, (double)N * K * 10 / (ms * 1e-3) / 1e12, ms * 100);
}
int main() {
    unsigned char* d; unsigned* out;
    const size_t bytes = (size_t)1 << 30;                      // 1 GiB: every pass streams from HBM
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&out, 8192); (void)hipMemset(d, 1, bytes);
    // (K slices per wave must hold at least 8 instructions' worth of every shape: K / waves >= 8 KiB for the 1-KiB shape)
    const long shapes[][3] = {{65536, 16384, 2}, {16384, 65536, 4}};
    for (auto& s : shapes) {
        run<0>("16 rows x 64 B", d, s[0], s[1], (int)s[2], out);
        run<1>("8 rows x 128 B", d, s[0], s[1], (int)s[2], out);
        run<2>("4 rows x 256 B", d, s[0], s[1], (int)s[2], out);
        run<3>("1 row x 1 KiB", d, s[0], s[1], (int)s[2], out);
    }
    return 0;
}
