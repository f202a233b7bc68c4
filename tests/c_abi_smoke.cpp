// A host with no Python and no torch: plain C-style calls through include/pq_hip.h (the drop-in boundary), device memory from
// the HIP runtime, results compared bit for bit with the plain-C oracle (oracle/liboracle.so — test infrastructure).
// Built and run by tests/test_c_abi_host.py on the GPU box.  Exit code 0 = every comparison passed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pq_hip.h"

extern "C" {
void oq_quant_rowwise(const void*, int, int64_t, int64_t, int64_t, int8_t*, int64_t, float*);
void oq_qlinear_s8(const int8_t*, int64_t, const float*, const int8_t*, int64_t, const float*, const void*, void*, int64_t, int,
                   int64_t, int64_t, int64_t);
void oq_silu_mul_quant_rowwise(const void*, int64_t, const void*, int64_t, int, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define PQ(x) do { int32_t s_ = (x); if (s_ != PQ_OK) { fprintf(stderr, "pq error %d at line %d: %s\n", s_, __LINE__, pq_last_error()); return 3; } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
static float frand(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return ((float)((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f); }

static int run(int64_t M, int64_t N, int64_t K) {
    uint32_t seed = (uint32_t)(M * 131 + N * 7 + K);
    uint16_t* x = (uint16_t*)malloc(M * K * 2); uint16_t* w = (uint16_t*)malloc(N * K * 2); uint16_t* u = (uint16_t*)malloc(M * K * 2);
    for (int64_t i = 0; i < M * K; ++i) { x[i] = f2bf(3.0f * frand(&seed)); u[i] = f2bf(frand(&seed)); }
    for (int64_t i = 0; i < N * K; ++i) w[i] = f2bf(0.05f * frand(&seed));
    // oracle: weight and activation quantisation, fused qlinear, fused silu*mul quantisation
    int8_t* wq = (int8_t*)malloc(N * K); float* ws = (float*)malloc(N * 4); int8_t* xq = (int8_t*)malloc(M * K); float* xs = (float*)malloc(M * 4);
    uint16_t* y = (uint16_t*)malloc(M * N * 2); int8_t* hq = (int8_t*)malloc(M * K); float* hs = (float*)malloc(M * 4);
    oq_quant_rowwise(w, PQ_BF16, N, K, K, wq, K, ws);
    oq_quant_rowwise(x, PQ_BF16, M, K, K, xq, K, xs);
    oq_qlinear_s8(xq, K, xs, wq, K, ws, NULL, y, N, PQ_BF16, M, N, K);
    oq_silu_mul_quant_rowwise(x, K, u, K, PQ_BF16, M, K, hq, K, hs, NULL, 0);
    // device side, through the C-ABI only
    void *dx, *dw, *du, *dwq, *dws, *dy, *dwork, *dhq, *dhs;
    CK(hipMalloc(&dx, M * K * 2)); CK(hipMalloc(&dw, N * K * 2)); CK(hipMalloc(&du, M * K * 2)); CK(hipMalloc(&dwq, N * K)); CK(hipMalloc(&dws, N * 4));
    CK(hipMalloc(&dy, M * N * 2)); CK(hipMalloc(&dhq, M * K)); CK(hipMalloc(&dhs, M * 4));
    const size_t wb = pq_qlinear_dyn_workspace_bytes(M, N, K);
    CK(hipMalloc(&dwork, wb));
    CK(hipMemcpy(dx, x, M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w, N * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(du, u, M * K * 2, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    PQ(pq_quant_rowwise(dw, PQ_BF16, N, K, K, (int8_t*)dwq, K, (float*)dws, st));
    PQ(pq_qlinear_dyn(dx, PQ_BF16, K, (const int8_t*)dwq, K, (const float*)dws, NULL, dy, N, M, N, K, dwork, wb, st));
    PQ(pq_silu_mul_quant_rowwise(dx, K, du, K, PQ_BF16, M, K, (int8_t*)dhq, K, (float*)dhs, NULL, 0, st));
    CK(hipStreamSynchronize(st));
    int8_t* g_wq = (int8_t*)malloc(N * K); uint16_t* g_y = (uint16_t*)malloc(M * N * 2); int8_t* g_hq = (int8_t*)malloc(M * K); float* g_hs = (float*)malloc(M * 4);
    CK(hipMemcpy(g_wq, dwq, N * K, hipMemcpyDeviceToHost)); CK(hipMemcpy(g_y, dy, M * N * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g_hq, dhq, M * K, hipMemcpyDeviceToHost)); CK(hipMemcpy(g_hs, dhs, M * 4, hipMemcpyDeviceToHost));
    const int bad = (memcmp(g_wq, wq, N * K) != 0) + (memcmp(g_y, y, M * N * 2) != 0) + (memcmp(g_hq, hq, M * K) != 0) + (memcmp(g_hs, hs, M * 4) != 0);
    printf("M=%lld N=%lld K=%lld [%s]: %s\n", (long long)M, (long long)N, (long long)K, pq_gemm_variant_name(M, N, K, K, K), bad ? "MISMATCH" : "bit-identical to the oracle");
    return bad;
}

int main(void) {
    if (pq_version() != PQ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 4; }
    // an argument error comes back as a status + message, never as an exception
    if (pq_quant_rowwise(NULL, 0, 4, 4, 4, NULL, 4, NULL, NULL) != PQ_ERR_BAD_ARG || strlen(pq_last_error()) == 0) { fprintf(stderr, "expected PQ_ERR_BAD_ARG\n"); return 5; }
    int bad = 0;
    bad += run(32, 512, 512);          // BASELINE config 1 (weight-streaming kernel)
    bad += run(300, 640, 1024);        // ragged rows, ring tiles
    bad += run(2048, 2560, 256);       // 256 x 256 tiles
    return bad ? 1 : 0;
}
