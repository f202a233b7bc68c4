// Dev microbenchmark: issue rate of the VALU ops the producer-fused quantisation kernel is made of
// (v_fma_f32, v_pk_fma_f32, v_rcp_f32, the IEEE division expansion, v_cvt_pk_bf16_f32), in lane-results per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x[CH]; v2f p[CH];
    for (int i = 0; i < CH; ++i) { x[i] = a + i + threadIdx.x; p[i] = v2f{x[i], x[i] + 0.5f}; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if constexpr (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if constexpr (OP == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{a, a}, v2f{b, b});
            if constexpr (OP == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if constexpr (OP == 3) x[i] = b / x[i];
            if constexpr (OP == 4) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); p[i] = p[i] + __builtin_convertvector(__builtin_convertvector(p[i], b2), v2f); }
            if constexpr (OP == 5) p[i] = p[i] * v2f{a, a};
            if constexpr (OP == 6) x[i] = __builtin_ldexpf(x[i], (int)a);
            if constexpr (OP == 7) x[i] = __builtin_rintf(x[i] * a);
        }
    }
    float s = 0; for (int i = 0; i < CH; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double results_per_op, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;
    k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = 5.0 * blocks * 256 * (double)ITERS * CH;
    printf("%-34s %8.1f G wave-lane-ops/s  = %6.1f lane-ops/clk/CU @2.4GHz  (%.1f results/clk/CU)\n", name, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.4e9, ops * results_per_op / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 1, out);
    run<1>("v_pk_fma_f32", 2, out);
    run<5>("v_pk_mul_f32", 2, out);
    run<2>("v_rcp_f32", 1, out);
    run<3>("IEEE division b / x", 1, out);
    run<4>("v_cvt_pk_bf16_f32 + unpack + pk_add", 2, out);
    run<6>("v_ldexp_f32", 1, out);
    run<7>("v_mul + v_rndne", 1, out);
    return 0;
}
