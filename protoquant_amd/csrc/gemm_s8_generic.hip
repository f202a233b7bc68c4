// gemm_s8_generic.hip — K3/K4 correctness-first variant: any M, N, K, leading dimension, alignment.
// 64x64 output tile, BK = 64, 4 waves (2x2), v_mfma_i32_16x16x64_i8, guarded zero-filled staging
// through LDS.  Used for ragged shapes and as the in-tree cross-check of the fast variant.
//
// MFMA roles are swapped on purpose (first operand = weight rows n, second = activation rows m), so a
// lane's 4 accumulator registers are 4 consecutive n of one output row m: the epilogue stores them as
// one contiguous piece of y[m, :].
#include "gemm_epilogue.h"

namespace pq {

constexpr int GT = 64;        // tile edge
constexpr int GBK = 64;       // K bytes per step
constexpr int GROW = GBK + 16;  // padded LDS row (keeps 16-B alignment)

__device__ __forceinline__ v4u guarded_load16(const int8_t* base, int64_t row, int64_t nrows, int64_t ld, int64_t k, int64_t K) {
    v4u v = {0, 0, 0, 0};
    if (row >= nrows) return v;
    const int8_t* p = base + row * ld + k;
    if (k + 16 <= K && (reinterpret_cast<uintptr_t>(p) & 15) == 0) return *reinterpret_cast<const v4u*>(p);
    uint8_t* b = reinterpret_cast<uint8_t*>(&v);
    for (int i = 0; i < 16; ++i) b[i] = (k + i < K) ? (uint8_t)p[i] : (uint8_t)0;
    return v;
}

template <int OUT>
__global__ __launch_bounds__(256) void gemm_s8_generic(const int8_t* __restrict__ A, int64_t lda,
                                                       const int8_t* __restrict__ B, int64_t ldb, EpiArgs epi,
                                                       int64_t M, int64_t N, int64_t K) {
    __shared__ __attribute__((aligned(16))) uint8_t sm[2 * GT * GROW];
    uint8_t* sA = sm;                 // activation rows (m)
    uint8_t* sB = sm + GT * GROW;     // weight rows (n)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GT, n0 = (int64_t)blockIdx.x * GT;
    const int lrow = tid >> 2, lchunk = tid & 3;

    v4i acc[2][2];   // [n-tile i][m-tile j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    for (int64_t k0 = 0; k0 < K; k0 += GBK) {
        const v4u va = guarded_load16(A, m0 + lrow, M, lda, k0 + lchunk * 16, K);
        const v4u vb = guarded_load16(B, n0 + lrow, N, ldb, k0 + lchunk * 16, K);
        __syncthreads();   // previous step's fragment reads are done
        *reinterpret_cast<v4u*>(sA + lrow * GROW + lchunk * 16) = va;
        *reinterpret_cast<v4u*>(sB + lrow * GROW + lchunk * 16) = vb;
        __syncthreads();
        v4i fn[2], fm[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            fn[i] = *reinterpret_cast<const v4i*>(sB + (wn * 32 + i * 16 + (lane & 15)) * GROW + (lane >> 4) * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            fm[j] = *reinterpret_cast<const v4i*>(sA + (wm * 32 + j * 16 + (lane & 15)) * GROW + (lane >> 4) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fn[i], fm[j], acc[i][j], 0, 0, 0);
    }

    // D[row = 4*(lane>>4)+r  <-> n][col = lane&15 <-> m]
    using O = typename OutElem<OUT>::type;
    O* y = reinterpret_cast<O*>(epi.y);
    const bool has_bias = epi.bias != nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t m = m0 + wm * 32 + j * 16 + (lane & 15);
        if (m >= M) continue;
        float as = 1.0f;
        if constexpr (OUT != OUT_I32) as = epi.a_scale[m];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = n0 + wn * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (n >= N) continue;
                float bs = 1.0f, bf = 0.0f;
                if constexpr (OUT != OUT_I32) {
                    bs = epi.b_scale[n];
                    if (has_bias) bf = load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : n);
                }
                y[m * epi.ldy + n] = epi_convert<OUT>(acc[i][j][r], as, bs, bf, has_bias, epi.flags & EPI_COL_FIRST);
            }
        }
    }
}

template <int OUT>
void launch_gemm_generic(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi,
                         int64_t M, int64_t N, int64_t K, hipStream_t st) {
    const dim3 grid((unsigned)((N + GT - 1) / GT), (unsigned)((M + GT - 1) / GT)), block(256);
    gemm_s8_generic<OUT><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, M, N, K);
}

template void launch_gemm_generic<PQ_BF16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template void launch_gemm_generic<PQ_FP16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template void launch_gemm_generic<PQ_F32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template void launch_gemm_generic<OUT_I32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);

}  // namespace pq
