// quant_device.h — device helpers shared by the quantising kernels (K1, K2, the producer-fused K1 variants):
// 16-byte vector unpacking, the division-free exact encode (QSPEC Q4-Q6), amax on raw bit patterns.
#pragma once
#include "pq_common.h"

namespace pq {

template <int DT, int N> struct Unpack;
template <> struct Unpack<PQ_BF16, 8> {
    __device__ static __forceinline__ void run(const v4u& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
            f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xFFFF0000u);
        }
    }
};
template <> struct Unpack<PQ_FP16, 8> {
    __device__ static __forceinline__ void run(const v4u& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = Elem<PQ_FP16>::to_f32((uint16_t)(v[i] & 0xFFFFu));
            f[2 * i + 1] = Elem<PQ_FP16>::to_f32((uint16_t)(v[i] >> 16));
        }
    }
};
template <> struct Unpack<PQ_F32, 4> {
    __device__ static __forceinline__ void run(const v4u& v, float (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u = v[i];   // copy first: bit_cast of a vector-element lvalue reads element 0
            f[i] = __builtin_bit_cast(float, u);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Exact-result fast encode (QSPEC Q4-Q6 without a division per element).
//   r = RN(1/s): ONE IEEE division per scale.  Per element the correctly rounded quotient Q = RN(x/s) is
//   rebuilt with two FMA residual corrections (Markstein: with y = RN(1/b) and q within 1 ulp of a/b,
//   r = a - b*q is exact and RN(q + r*y) = RN(a/b), unless b's significand is all ones):
//       q = x*r;  e = fma(-q, s, x);  q = fma(e, r, q);  e = fma(-q, s, x);  q = fma(e, r, q)   ==  x / s
//   then m = q + 1.5*2^23 rounds q to the nearest-even integer k (= rintf) and leaves k's two's complement
//   in the low mantissa bits, so the code byte is the low byte of m.  |x| <= amax gives |k| <= 127: no clamp.
// Valid when 2^-60 < s < 2^60, s's significand is not all ones and the data hold no NaN/Inf
// (scale_fast_ok + the amax bit test); everything else takes the exact-division path.  The identity is
// checked by brute force on the GPU in tests/test_gpu_parity.py::test_fast_quotient_bruteforce.
constexpr float kMagic = 12582912.0f;          // 1.5 * 2^23

__device__ __forceinline__ bool scale_fast_ok(float s) {
    const uint32_t b = __builtin_bit_cast(uint32_t, s);
    const uint32_t e = b >> 23;                 // s > 0
    return e >= 67u && e <= 187u && (b & 0x7FFFFFu) != 0x7FFFFFu;
}
__device__ __forceinline__ float quotient_fast(float x, float s, float r) {
    float q = x * r;
    float e = __builtin_fmaf(-q, s, x);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-q, s, x);
    q = __builtin_fmaf(e, r, q);
    return q;
}
// 16-bit inputs need only ONE correction step for the CODE (not the quotient) to be exact: the row's amax is itself a bf16 / fp16
// value, so s = amax / 127 takes one of 2^15 values and x one of the 2^15 magnitudes below amax — a domain small enough to enumerate.
// Every (amax, x) pair of both formats whose scale passes scale_fast_ok was checked: 0 mismatches against rintf(x / s) with one step
// (2 904 / 8 366 with none).  tests/test_half_quotient_identity.py repeats the enumeration on the host in C,
// pq_selftest_half_encode on the GPU.  fp32 inputs keep the two-step form (Markstein's theorem, 2^28 sampled patterns).
__device__ __forceinline__ float quotient_fast1(float x, float s, float r) {
    float q = x * r;
    const float e = __builtin_fmaf(-q, s, x);
    return __builtin_fmaf(e, r, q);
}
template <int DT> constexpr int kQuotientSteps = DT == PQ_F32 ? 2 : 1;
// encodes N elements; returns the N code bytes packed little-endian
template <int N, int STEPS = 2>
__device__ __forceinline__ void fast_encode(const float (&f)[N], float s, float r, uint32_t (&packed)[N / 4]) {
    uint32_t mb[N];
#pragma unroll
    for (int j = 0; j < N; ++j) mb[j] = __builtin_bit_cast(uint32_t, (STEPS == 1 ? quotient_fast1(f[j], s, r) : quotient_fast(f[j], s, r)) + kMagic);
#pragma unroll
    for (int g = 0; g < N / 4; ++g)
        packed[g] = __builtin_amdgcn_perm(mb[4 * g + 1], mb[4 * g], 0x0c0c0400u) | __builtin_amdgcn_perm(mb[4 * g + 3], mb[4 * g + 2], 0x04000c0cu);
}

// amax of one 16-byte vector on the raw bit patterns (integer max; NaN patterns sort above Inf and are
// detected afterwards).  Returns max |x| bits widened to f32 bit patterns.
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {          // v_pk_max_u16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}
// `cur` is an ACCUMULATOR: for 16-bit formats it holds two running maxima (one per half-word: and + v_pk_max_u16 per packed pair);
// amax_acc_finish() folds it into the widened bit pattern the rest of the kernels use.
template <int DT>
__device__ __forceinline__ uint32_t vec_amax_bits(const v4u& v, uint32_t cur) {
    if constexpr (DT == PQ_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t a = v[i] & 0x7FFFFFFFu; cur = a > cur ? a : cur; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) cur = pk_max_u16(cur, v[i] & 0x7FFF7FFFu);
    }
    return cur;
}
template <int DT> __device__ __forceinline__ uint32_t amax_acc_finish(uint32_t acc) {
    if constexpr (DT == PQ_F32) return acc;
    else { const uint32_t lo = acc & 0xFFFFu, hi = acc >> 16; return lo > hi ? lo : hi; }
}
template <int DT> __device__ __forceinline__ bool amax_bits_has_nan(uint32_t b) {
    if constexpr (DT == PQ_F32) return b > 0x7F800000u;
    else if constexpr (DT == PQ_BF16) return b > 0x7F80u;
    else return b > 0x7C00u;
}
template <int DT> __device__ __forceinline__ float amax_bits_to_f32(uint32_t b) {
    if constexpr (DT == PQ_F32) return __builtin_bit_cast(float, b);
    else return Elem<DT>::to_f32((uint16_t)b);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

}  // namespace pq
