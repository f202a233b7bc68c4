// gemm_epilogue.h — K4: the fused dequant epilogue shared by every GEMM variant.
// OUT codes: PQ_BF16 / PQ_FP16 / PQ_F32 = QSPEC E1-E4 output; OUT_I32 = raw accumulator (parity twin).
#pragma once
#include "pq_common.h"

namespace pq {

constexpr int OUT_I32 = 3;

struct EpiArgs {
    const float* a_scale;   // [M] row scales   (unused for OUT_I32)
    const float* b_scale;   // [N] column scales
    const void* bias;       // [N] in output dtype, nullable
    void* y;                // [M, ldy]
    int64_t ldy;
};

template <int OUT> struct OutElem { using type = typename Elem<OUT>::store_t; };
template <> struct OutElem<OUT_I32> { using type = int32_t; };

// one accumulator -> one output element value (still in registers)
template <int OUT>
__device__ __forceinline__ typename OutElem<OUT>::type epi_convert(int acc, float as, float bs, float bias_f, bool has_bias) {
    if constexpr (OUT == OUT_I32) {
        return acc;
    } else {
        float t = epilogue_val(acc, as, bs);
        if (has_bias) t = t + bias_f;          // separate rounded add (built with -ffp-contract=off)
        return Elem<OUT>::from_f32(t);
    }
}

template <int OUT>
__device__ __forceinline__ float load_bias(const void* bias, int64_t n) {
    if constexpr (OUT == OUT_I32) return 0.0f;
    else return Elem<OUT>::to_f32(reinterpret_cast<const typename Elem<OUT>::store_t*>(bias)[n]);
}

}  // namespace pq
