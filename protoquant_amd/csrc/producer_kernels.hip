// producer_kernels.hip — K1 fused into the op that produces the activation:
//   silu(g) * u           ->  per-token int8 codes + row scales   (K1s: the `down` input of a gated MLP)
//   RMSNorm(x; weight)    ->  per-token int8 codes + row scales   (K1n: the q/k/v and gate/up input of a decoder layer)
// without the bf16 activation ever going to HBM.
// Same skeleton as K1 (quant_kernels.hip): TPR threads own a row, the row of h lives in registers between the amax
// reduction and the encode; here it is COMPUTED from one 16-byte vector of g and one of u per slot instead of loaded.
// Arithmetic follows QSPEC S1-S6: a specified exponential (Cody-Waite + degree-7 Horner with fma),
// IEEE division, storage-dtype rounding after silu and after the product — bit-identical to oracle/qspec_oracle.c.
// Algorithmic traffic: read 2 x elem bytes, write 1 B/elem + 4 B/row (+ elem bytes when h is also requested).
#include <type_traits>

#include "quant_device.h"

namespace pq {

typedef float v2f __attribute__((ext_vector_type(2)));

// Two elements at a time: gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at the rate of their scalar forms,
// and one v_cvt_pk_bf16_f32 rounds both.  This kernel is VALU-bound before it is HBM-bound (about 50 scalar VALU ops
// per element against 5 bytes), so the pairing is what moves it.
__device__ __forceinline__ v2f splat(float v) { return v2f{v, v}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

template <int DT> struct Pair;            // one 32-bit word of storage <-> two floats
template <> struct Pair<PQ_BF16> {
    typedef __bf16 st2 __attribute__((ext_vector_type(2)));
    __device__ static __forceinline__ v2f unpack(uint32_t w) { return v2f{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)}; }
    __device__ static __forceinline__ uint32_t pack(v2f f) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, st2)); }
};
template <> struct Pair<PQ_FP16> {
    typedef _Float16 st2 __attribute__((ext_vector_type(2)));
    __device__ static __forceinline__ v2f unpack(uint32_t w) { return __builtin_convertvector(__builtin_bit_cast(st2, w), v2f); }
    __device__ static __forceinline__ uint32_t pack(v2f f) {
        asm volatile("" : "+v"(f));     // a materialised f32 pair: no v_fma_mixlo_f16 folding (see Elem<PQ_FP16>::from_f32)
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, st2));
    }
};

// QSPEC S1-S5 on NP pairs at once, written stage by stage so that NP independent instructions follow each other: one
// wave's dependent v_pk_fma chain leaves the VALU idle most of the time (measured: 2x off the issue rate), NP chains do
// not.  Notes on the forms used:
//  - the clamp is v_med3_f32 (a NaN comes out finite): the only consumer divides g by 1 + exp(-g), so a NaN g still
//    yields NaN, exactly as the specification's pass-through does;
//  - ldexp(p, n) equals the specification's two exact power-of-two multiplications for every n in [-43, 144];
//  - FASTDIV: the IEEE quotient g / d without v_div_scale / v_div_fmas / v_div_fixup, which serialise on VCC and run at
//    ~6 results/clk/CU against ~100 for an fma (tools/ubench/valu_rate).  It is the arithmetic core of the hardware's own
//    correctly rounded sequence — rcp, one Newton step, the quotient and two residual corrections — without the operand
//    scaling, which is only needed when an intermediate can overflow or lose bits to underflow.  For 0 < |g| <= 86 none
//    can: d lies in [1, 2^125), the residuals g - d*q are exact (for |g| < 2^-25, d is exactly 2 and every step is an exact
//    scaling).  Waves holding a zero (whose sign the residual steps would lose), |g| > 86, Inf or NaN take the `/` path
//    (silu_fast_div_ok, decided once per wave).
// Returns the products BEFORE their storage rounding.
//  - SHORT (16-bit storage only): silu(g) is rounded to the storage format before the product, and g itself is a 16-bit value, so
//    the stored silu(g) is a function of 65 536 inputs.  On ALL of the fast-division domain ONE residual correction on the raw rcp
//    (q = g*y0; e = fma(-d, q, g); q = fma(e, y0, q)) rounds to the same stored value as the correctly rounded quotient — enumerated on
//    the GPU (pq_selftest_silu_short, tests/test_gpu_parity.py::test_silu_short_division_whole_domain); the Newton step and the second
//    correction (4 of ~40 VALU results per element) are dropped for bf16 / fp16 rows.
template <int DT, bool FASTDIV, int NP, bool SHORT = false>
__device__ __forceinline__ void silu_mul_stage(const v2f (&g)[NP], const v2f (&u)[NP], v2f (&h)[NP]) {
    v2f tc[NP], n[NP], r[NP], p[NP], d[NP], sg[NP];
    if constexpr (FASTDIV && SHORT && DT == PQ_BF16) {
        // bf16 rows on the fast-division domain: 1 + exp(-g) from the hardware's exp2 (v_exp_f32, ~1 ulp).  The STORED silu(g) is a function of
        // the 16-bit g alone, and on every one of the 34 136 patterns of the domain this sequence stores the value the specified polynomial
        // exponential + correctly rounded quotient store (tools/ubench/silu_variants enumerates the candidates; pq_selftest_silu_short
        // re-checks the shipped one on the GPU it runs on).  Fourteen VALU results per element fewer.  (fp16 keeps the polynomial: with its
        // 11-bit significand two patterns differ.)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const v2f a = g[k] * splat(-__builtin_bit_cast(float, 0x3FB8AA3Bu));          // -g * log2(e)
            d[k] = splat(1.0f) + v2f{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        }
    } else {
#pragma unroll
    for (int k = 0; k < NP; ++k) tc[k] = v2f{__builtin_amdgcn_fmed3f(-g[k].x, -30.0f, 100.0f), __builtin_amdgcn_fmed3f(-g[k].y, -30.0f, 100.0f)};
#pragma unroll
    for (int k = 0; k < NP; ++k) n[k] = tc[k] * splat(__builtin_bit_cast(float, 0x3FB8AA3Bu));
#pragma unroll
    for (int k = 0; k < NP; ++k) n[k] = v2f{__builtin_rintf(n[k].x), __builtin_rintf(n[k].y)};
#pragma unroll
    for (int k = 0; k < NP; ++k) r[k] = pk_fma(n[k], splat(-__builtin_bit_cast(float, 0x3F317200u)), tc[k]);
#pragma unroll
    for (int k = 0; k < NP; ++k) r[k] = pk_fma(n[k], splat(-__builtin_bit_cast(float, 0x35BFBE8Eu)), r[k]);
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = pk_fma(splat(__builtin_bit_cast(float, 0x39500D01u)), r[k], splat(__builtin_bit_cast(float, 0x3AB60B61u)));
    constexpr uint32_t kC[6] = {0x3C088889u, 0x3D2AAAABu, 0x3E2AAAABu, 0x3F000000u, 0x3F800000u, 0x3F800000u};
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int k = 0; k < NP; ++k) p[k] = pk_fma(p[k], r[k], splat(__builtin_bit_cast(float, kC[c])));
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) d[k] = splat(1.0f) + v2f{__builtin_ldexpf(p[k].x, (int)n[k].x), __builtin_ldexpf(p[k].y, (int)n[k].y)};
    }
    if constexpr (FASTDIV && SHORT) {
        v2f y0[NP], q[NP], e[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) y0[k] = v2f{__builtin_amdgcn_rcpf(d[k].x), __builtin_amdgcn_rcpf(d[k].y)};
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = g[k] * y0[k];
#pragma unroll
        for (int k = 0; k < NP; ++k) e[k] = pk_fma(-d[k], q[k], g[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) sg[k] = pk_fma(e[k], y0[k], q[k]);
    } else if constexpr (FASTDIV) {
        v2f y0[NP], y[NP], q[NP], e[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) y0[k] = v2f{__builtin_amdgcn_rcpf(d[k].x), __builtin_amdgcn_rcpf(d[k].y)};
#pragma unroll
        for (int k = 0; k < NP; ++k) e[k] = pk_fma(-d[k], y0[k], splat(1.0f));
#pragma unroll
        for (int k = 0; k < NP; ++k) y[k] = pk_fma(e[k], y0[k], y0[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = g[k] * y[k];
#pragma unroll
        for (int k = 0; k < NP; ++k) e[k] = pk_fma(-d[k], q[k], g[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = pk_fma(e[k], y[k], q[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) e[k] = pk_fma(-d[k], q[k], g[k]);
#pragma unroll
        for (int k = 0; k < NP; ++k) sg[k] = pk_fma(e[k], y[k], q[k]);
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) sg[k] = v2f{g[k].x / d[k].x, g[k].y / d[k].y};
    }
    if constexpr (DT != PQ_F32) {
#pragma unroll
        for (int k = 0; k < NP; ++k) sg[k] = Pair<DT>::unpack(Pair<DT>::pack(sg[k]));
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) h[k] = sg[k] * u[k];
}
template <int DT>
__device__ __forceinline__ float silu_mul_spec(float g, float u) {
    const v2f ga[1] = {v2f{g, g}}, ua[1] = {v2f{u, u}};
    v2f h[1];
    silu_mul_stage<DT, false, 1>(ga, ua, h);
    return h[0].x;
}

// min / max of |g| over one 16-byte vector, on raw bit patterns (the same ordering trick as vec_amax_bits)
template <int DT>
__device__ __forceinline__ void vec_absminmax_bits(const v4u& v, uint32_t& mn, uint32_t& mx) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (DT == PQ_F32) {
            const uint32_t a = v[i] & 0x7FFFFFFFu;
            mn = a < mn ? a : mn;
            mx = a > mx ? a : mx;
        } else {
            const uint32_t a = v[i] & 0x7FFF7FFFu, lo = a & 0xFFFFu, hi = a >> 16;
            mn = min(mn, min(lo, hi));
            mx = max(mx, max(lo, hi));
        }
    }
}
template <int DT> __device__ __forceinline__ bool silu_fast_div_ok(uint32_t mn, uint32_t mx) {
    constexpr uint32_t k86 = DT == PQ_F32 ? 0x42AC0000u : (DT == PQ_BF16 ? 0x42ACu : 0x5560u);   // 86.0
    return mn != 0u && mx <= k86;
}

// one 16-byte vector of g and of u -> one 16-byte vector of h in the storage dtype
template <int DT, bool FASTDIV, bool SHORT = (DT != PQ_F32)>
__device__ __forceinline__ v4u silu_mul_vec(const v4u& gv, const v4u& uv) {
    constexpr int NP = DT == PQ_F32 ? 2 : 4;
    v2f g[NP], u[NP], h[NP];
    v4u out;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if constexpr (DT == PQ_F32) {
            const uint32_t g0 = gv[2 * j], g1 = gv[2 * j + 1], u0 = uv[2 * j], u1 = uv[2 * j + 1];   // copies first (hipcc quirk)
            g[j] = v2f{__builtin_bit_cast(float, g0), __builtin_bit_cast(float, g1)};
            u[j] = v2f{__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1)};
        } else {
            const uint32_t gw = gv[j], uw = uv[j];
            g[j] = Pair<DT>::unpack(gw);
            u[j] = Pair<DT>::unpack(uw);
        }
    }
    silu_mul_stage<DT, FASTDIV, NP, SHORT && DT != PQ_F32>(g, u, h);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if constexpr (DT == PQ_F32) {
            const float hx = h[j].x, hy = h[j].y;   // copies first: bit_cast of a vector-element lvalue reads element 0
            out[2 * j] = __builtin_bit_cast(uint32_t, hx);
            out[2 * j + 1] = __builtin_bit_cast(uint32_t, hy);
        } else {
            out[j] = Pair<DT>::pack(h[j]);
        }
    }
    return out;
}

// division-free exact encode (quant_device.h) of one 16-byte vector of h, two elements per instruction
template <int DT>
__device__ __forceinline__ void fast_encode_vec(const v4u& hv, float s, float r, uint32_t (&pk)[(16 / Elem<DT>::kBytes) / 4]) {
    const v2f vs = splat(s), vr = splat(r);
    uint32_t mb[16 / Elem<DT>::kBytes];
#pragma unroll
    for (int j = 0; j < (16 / Elem<DT>::kBytes) / 2; ++j) {
        v2f x;
        if constexpr (DT == PQ_F32) {
            const uint32_t a = hv[2 * j], b = hv[2 * j + 1];
            x = v2f{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
        } else {
            const uint32_t w = hv[j];
            x = Pair<DT>::unpack(w);
        }
        v2f q = x * vr;
        v2f e = pk_fma(-q, vs, x);
        q = pk_fma(e, vr, q);
        if constexpr (kQuotientSteps<DT> == 2) {          // 16-bit h: one step is exact for the code (quant_device.h)
            e = pk_fma(-q, vs, x);
            q = pk_fma(e, vr, q);
        }
        const v2f m = q + splat(kMagic);
        const float mx = m.x, my = m.y;       // copies first (same hipcc quirk)
        mb[2 * j] = __builtin_bit_cast(uint32_t, mx);
        mb[2 * j + 1] = __builtin_bit_cast(uint32_t, my);
    }
#pragma unroll
    for (int k = 0; k < (16 / Elem<DT>::kBytes) / 4; ++k)
        pk[k] = __builtin_amdgcn_perm(mb[4 * k + 1], mb[4 * k], 0x0c0c0400u) | __builtin_amdgcn_perm(mb[4 * k + 3], mb[4 * k + 2], 0x04000c0cu);
}

// The second half of K1, shared by the producer-fused kernels, in two steps: the row amax of the h vectors held in registers (bit-pattern
// max: a NaN propagates into the scale) as an f32 bit pattern, then scale + the division-free exact encode (or the true-division
// path for NaN/Inf data and extreme scales) for a GIVEN row amax.  TPR threads own the row; t = thread's index in the row.
// The split is what the column-sharded gated MLP needs (pq_silu_mul_rowamax / pq_silu_mul_quant_rowwise_amax): a rank holds only I/G of a
// token's intermediate channels, the row amax is an exact max over the ranks (an integer max of these bit patterns), and the encode
// then runs locally against the GLOBAL amax — the codes are the unsharded kernel's, bit for bit.
template <int DT, int TPR>
__device__ __forceinline__ uint32_t row_amax_f32_bits(uint32_t ab) {           // `ab` arrives as vec_amax_bits' accumulator
    ab = wave_max_u32(amax_acc_finish<DT>(ab));
    constexpr int NW = (TPR > 256 ? TPR : 256) / kWave;      // waves of the block (a row group wider than a wave is the whole block)
    __shared__ uint32_t part[NW];
    if constexpr (TPR > kWave) {
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = ab;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) ab = part[w] > ab ? part[w] : ab;
    }
    // widened to the f32 pattern of the same value: non-negative floats (and NaNs, which sort above +Inf) order as unsigned integers in every format
    return __builtin_bit_cast(uint32_t, amax_bits_to_f32<DT>(ab)) & 0x7FFFFFFFu;
}
template <int DT, int VPT, int TPR>
__device__ __forceinline__ void encode_with_amax(const v4u (&hv)[VPT], uint32_t amax_f32_bits, int t, int nvec, bool active, int64_t row,
                                                 int8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const bool has_nan = amax_f32_bits > 0x7F800000u;        // QSPEC v2: a NaN propagates (scale = canonical NaN, codes 0 by the true-division path)
    const float s = scale_of(__builtin_bit_cast(float, amax_f32_bits));
    if (!active) return;
    if (t == 0) scale[row] = s;
    int8_t* qr = q + row * ldq;
    auto store_vec = [&](int idx, const uint32_t (&pk)[EPV / 4]) {          // write-through (pq_common.h)
        if constexpr (EPV == 8) store_wt_b64(qr + (int64_t)idx * 8, v2u{pk[0], pk[1]});
        else store_wt_b32(qr + (int64_t)idx * 4, pk[0]);
    };
    if (!has_nan && scale_fast_ok(s)) {
        const float r = 1.0f / s;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = i * TPR + t;
            uint32_t pk[EPV / 4];
            fast_encode_vec<DT>(hv[i], s, r, pk);
            if (idx < nvec) store_vec(idx, pk);
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = i * TPR + t;
            float f[EPV];
            Unpack<DT, EPV>::run(hv[i], f);
            uint32_t pk[EPV / 4];
#pragma unroll
            for (int k = 0; k < EPV / 4; ++k)
                pk[k] = pack4(code_of(f[4 * k], s), code_of(f[4 * k + 1], s), code_of(f[4 * k + 2], s), code_of(f[4 * k + 3], s));
            if (idx < nvec) store_vec(idx, pk);
        }
    }
}
template <int DT, int VPT, int TPR>
__device__ __forceinline__ void reduce_and_encode(const v4u (&hv)[VPT], uint32_t ab, int t, int nvec, bool active, int64_t row,
                                                  int8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
    encode_with_amax<DT, VPT, TPR>(hv, row_amax_f32_bits<DT, TPR>(ab), t, nvec, active, row, q, ldq, scale);
}

// TPR = 512 (a 512-thread block per row, wide rows): round 1 held such rows with 256 threads x 8 vectors of g and of u —
// 143 VGPRs, 3 waves per SIMD in a kernel that is VALU-bound before it is HBM-bound; 512 threads x 3-4 vectors need < 100.
// MODE 0: K1s (amax + encode).  MODE 1: the row amax only (amax_io[row] = f32 bit pattern of max |h| over these columns; nothing else is written).
// MODE 2: encode against the row amax GIVEN in amax_io (the max over every rank's columns): no reduction; writes codes and the scale.
// IDENT (MODE 1 / 2 only): h = g itself — the same two halves for a PLAIN column-sharded activation (a rank's heads of the attention output feeding a
// column-sharded `o` projection: pq_quant_rowamax / pq_quant_rowwise_amax); u is not read.
template <int DT, int VPT, int TPR, bool WRITE_H, int MODE = 0, bool IDENT = false>
__global__ __launch_bounds__(TPR > 256 ? TPR : 256) void silu_mul_quant_vec(const uint8_t* __restrict__ g, int64_t ldg_bytes,
                                                          const uint8_t* __restrict__ u, int64_t ldu_bytes, int64_t rows,
                                                          int nvec, int8_t* __restrict__ q, int64_t ldq,
                                                          float* __restrict__ scale, uint8_t* __restrict__ h_out,
                                                          int64_t ldh_bytes, uint32_t* __restrict__ amax_io = nullptr) {
    constexpr int BS = TPR > 256 ? TPR : 256;
    constexpr int RPB = BS / TPR;
    const int t = threadIdx.x % TPR;
    int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
    const bool active = row < rows;
    row = active ? row : rows - 1;
    const uint8_t* gr = g + row * ldg_bytes;
    const uint8_t* ur = u + row * ldu_bytes;

    // every load is issued before the first use (clamped addresses: a duplicate of the tail vector changes no max)
    v4u gv[VPT], uv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = i * TPR + t;
        const int64_t off = (int64_t)(idx < nvec ? idx : nvec - 1) * 16;
        gv[i] = *reinterpret_cast<const v4u*>(gr + off);
        if constexpr (!IDENT) uv[i] = *reinterpret_cast<const v4u*>(ur + off);
    }
    v4u hv[VPT];
    uint32_t ab = 0;
    uint32_t gmn = 0xFFFFFFFFu, gmx = 0u;
    if constexpr (!IDENT) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) vec_absminmax_bits<DT>(gv[i], gmn, gmx);
    }
    const bool fast_div = IDENT || __builtin_amdgcn_ballot_w64(!silu_fast_div_ok<DT>(gmn, gmx)) == 0ull;   // wave-uniform
    auto produce = [&](auto fast) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = i * TPR + t;
            // slots past the row's end (whole waves of them when the width is not VPT * TPR vectors) skip the arithmetic
            if constexpr (IDENT) hv[i] = idx < nvec ? gv[i] : v4u{0u, 0u, 0u, 0u};
            else hv[i] = idx < nvec ? silu_mul_vec<DT, decltype(fast)::value>(gv[i], uv[i]) : v4u{0u, 0u, 0u, 0u};
            if constexpr (MODE != 2) ab = vec_amax_bits<DT>(hv[i], ab);
            if constexpr (WRITE_H) {
                if (active && idx < nvec) store_wt_b128(h_out + row * ldh_bytes + (int64_t)idx * 16, hv[i]);
            }
        }
    };
    if (fast_div) produce(std::true_type{});
    else produce(std::false_type{});
    if constexpr (MODE == 1) {
        const uint32_t fb = row_amax_f32_bits<DT, TPR>(ab);
        if (active && t == 0) amax_io[row] = fb;
    } else if constexpr (MODE == 2) {
        encode_with_amax<DT, VPT, TPR>(hv, amax_io[row], t, nvec, active, row, q, ldq, scale);
    } else {
        reduce_and_encode<DT, VPT, TPR>(hv, ab, t, nvec, active, row, q, ldq, scale);
    }
}

// generic path: any cols / leading dimensions / alignment.  One block per row; h is recomputed in the second pass.
template <int DT, int MODE = 0, bool IDENT = false>
__global__ __launch_bounds__(256) void silu_mul_quant_generic(const void* __restrict__ g, int64_t ldg, const void* __restrict__ u,
                                                              int64_t ldu, int64_t cols, int8_t* __restrict__ q, int64_t ldq,
                                                              float* __restrict__ scale, void* __restrict__ h_out, int64_t ldh,
                                                              uint32_t* __restrict__ amax_io = nullptr) {
    using S = typename Elem<DT>::store_t;
    const int64_t row = blockIdx.x;
    const S* gr = reinterpret_cast<const S*>(g) + row * ldg;
    const S* ur = reinterpret_cast<const S*>(u) + row * ldu;
    auto h_at = [&](int64_t c) -> S {
        if constexpr (IDENT) return gr[c];
        else return Elem<DT>::from_f32(silu_mul_spec<DT>(Elem<DT>::to_f32(gr[c]), Elem<DT>::to_f32(ur[c])));
    };
    float amax = 0.0f;
    if constexpr (MODE == 2) {
        amax = __builtin_bit_cast(float, amax_io[row]);
    } else {
        for (int64_t c = threadIdx.x; c < cols; c += 256) {
            const S h = h_at(c);
            if (h_out) reinterpret_cast<S*>(h_out)[row * ldh + c] = h;
            amax = amax_step(amax, Elem<DT>::to_f32(h));
        }
        amax = wave_max(amax);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) amax = amax_merge(amax, part[w]);
    }
    if constexpr (MODE == 1) {
        if (threadIdx.x == 0) amax_io[row] = __builtin_bit_cast(uint32_t, amax);
        return;
    }
    const float s = scale_of(amax);
    if (threadIdx.x == 0) scale[row] = s;
    int8_t* qr = q + row * ldq;
    for (int64_t c = threadIdx.x; c < cols; c += 256) qr[c] = (int8_t)code_of(Elem<DT>::to_f32(h_at(c)), s);
}

// ------------------------------------------------------------------------------------------------
// K1n: RMSNorm fused into the per-token quantisation (QSPEC N1-N6).  One row per 256-thread block (the reduction order N1-N3
// IS this layout: vector v on lane v mod 256, xor butterfly per 64 lanes, the four wave sums left to right).  Reads x once
// (2 B/elem) and the weight vector from L2, writes 1 B/elem + 4 B/row: the normalised bf16 activation never goes to HBM.
__device__ __forceinline__ float rms_block_sum(float acc) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off, 64);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    return ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}
__device__ __forceinline__ float rms_rs(float ss, int cols, float eps) {
    const float var = ss / (float)cols;
    return 1.0f / __builtin_sqrtf(var + eps);
}
template <int DT>
__device__ __forceinline__ float rms_h(float x, float w, float rs) {
    const float xn = Elem<DT>::to_f32(Elem<DT>::from_f32(x * rs));
    return w * xn;       // the caller rounds to the storage dtype
}

// one 16-byte vector of x and of the weight -> one 16-byte vector of h (QSPEC N5), two elements per instruction
template <int DT>
__device__ __forceinline__ v4u rms_h_vec(const v4u& xv, const v4u& wv, float rs) {
    v4u out;
    if constexpr (DT == PQ_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t xb = xv[j], wb = wv[j];      // copies first (hipcc quirk with vector-element lvalues)
            out[j] = __builtin_bit_cast(uint32_t, rms_h<DT>(__builtin_bit_cast(float, xb), __builtin_bit_cast(float, wb), rs));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t xw = xv[j], ww = wv[j];
            const v2f xn = Pair<DT>::unpack(Pair<DT>::pack(Pair<DT>::unpack(xw) * splat(rs)));
            out[j] = Pair<DT>::pack(Pair<DT>::unpack(ww) * xn);
        }
    }
    return out;
}

template <int DT, int VPT, bool WRITE_H>
__global__ __launch_bounds__(256) void rmsnorm_quant_vec(const uint8_t* __restrict__ x, int64_t ldx_bytes, const uint8_t* __restrict__ wgt,
                                                         float eps, int cols, int nvec, int8_t* __restrict__ q, int64_t ldq,
                                                         float* __restrict__ scale, uint8_t* __restrict__ h_out, int64_t ldh_bytes) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const int t = threadIdx.x;
    const int64_t row = blockIdx.x;
    const uint8_t* xr = x + row * ldx_bytes;
    v4u xv[VPT], wv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = i * 256 + t;
        const int64_t off = (int64_t)(idx < nvec ? idx : nvec - 1) * 16;
        xv[i] = *reinterpret_cast<const v4u*>(xr + off);
        wv[i] = *reinterpret_cast<const v4u*>(wgt + off);
    }
    float acc = 0.0f;                       // N2: this lane's vectors in increasing v, elements in order
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (i * 256 + t >= nvec) xv[i] = v4u{0u, 0u, 0u, 0u};      // past the row: fma(0, 0, acc) = acc
        float f[EPV];
        Unpack<DT, EPV>::run(xv[i], f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc = __builtin_fmaf(f[j], f[j], acc);
    }
    const float rs = rms_rs(rms_block_sum(acc), cols, eps);        // N3, N4
    v4u hv[VPT];
    uint32_t ab = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        hv[i] = rms_h_vec<DT>(xv[i], wv[i], rs);
        ab = vec_amax_bits<DT>(hv[i], ab);
        if constexpr (WRITE_H) {
            const int idx = i * 256 + t;
            if (idx < nvec) store_wt_b128(h_out + row * ldh_bytes + (int64_t)idx * 16, hv[i]);
        }
    }
    reduce_and_encode<DT, VPT, 256>(hv, ab, t, nvec, true, row, q, ldq, scale);
}

// Short rows (at most 512 vectors, e.g. a 4096-wide bf16 hidden state): one WAVE per row, four rows per block and no block
// barrier, like K1.  The wave plays all four 64-lane groups of the specification: physical lane l holds virtual lanes
// l, l+64, l+128, l+192 (vector v = i*64 + l belongs to virtual lane v mod 256 = (i mod 4)*64 + l), keeps one accumulator
// per group, runs the xor butterfly on each and adds the four sums left to right — the same float operations in the same
// order as the 256-thread layout, so the same bits.
template <int DT, int VPT, bool WRITE_H>
__global__ __launch_bounds__(256) void rmsnorm_quant_wave(const uint8_t* __restrict__ x, int64_t ldx_bytes, const uint8_t* __restrict__ wgt,
                                                          float eps, int cols, int nvec, int64_t rows, int8_t* __restrict__ q, int64_t ldq,
                                                          float* __restrict__ scale, uint8_t* __restrict__ h_out, int64_t ldh_bytes) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const int t = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool active = row < rows;
    row = active ? row : rows - 1;
    const uint8_t* xr = x + row * ldx_bytes;
    v4u xv[VPT], wv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = i * 64 + t;
        const int64_t off = (int64_t)(idx < nvec ? idx : nvec - 1) * 16;
        xv[i] = *reinterpret_cast<const v4u*>(xr + off);
        wv[i] = *reinterpret_cast<const v4u*>(wgt + off);
    }
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (i * 64 + t >= nvec) xv[i] = v4u{0u, 0u, 0u, 0u};
        float f[EPV];
        Unpack<DT, EPV>::run(xv[i], f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[i & 3] = __builtin_fmaf(f[j], f[j], acc[i & 3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) acc[gi] = acc[gi] + __shfl_xor(acc[gi], off, 64);
    }
    const float rs = rms_rs(((acc[0] + acc[1]) + acc[2]) + acc[3], cols, eps);
    v4u hv[VPT];
    uint32_t ab = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        hv[i] = rms_h_vec<DT>(xv[i], wv[i], rs);
        ab = vec_amax_bits<DT>(hv[i], ab);
        if constexpr (WRITE_H) {
            const int idx = i * 64 + t;
            if (active && idx < nvec) store_wt_b128(h_out + row * ldh_bytes + (int64_t)idx * 16, hv[i]);
        }
    }
    reduce_and_encode<DT, VPT, 64>(hv, ab, t, nvec, active, row, q, ldq, scale);
}

// generic path (ragged widths, unaligned pointers): the same lane layout walked element by element.
template <int DT>
__global__ __launch_bounds__(256) void rmsnorm_quant_generic(const void* __restrict__ x, int64_t ldx, const void* __restrict__ wgt, float eps,
                                                             int64_t cols, int8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale,
                                                             void* __restrict__ h_out, int64_t ldh) {
    using S = typename Elem<DT>::store_t;
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const int64_t row = blockIdx.x;
    const S* xr = reinterpret_cast<const S*>(x) + row * ldx;
    const S* wr = reinterpret_cast<const S*>(wgt);
    const int64_t nvec = (cols + EPV - 1) / EPV;
    float acc = 0.0f;
    for (int64_t v = threadIdx.x; v < nvec; v += 256)
        for (int e = 0; e < EPV && v * EPV + e < cols; ++e) {
            const float f = Elem<DT>::to_f32(xr[v * EPV + e]);
            acc = __builtin_fmaf(f, f, acc);
        }
    const float rs = rms_rs(rms_block_sum(acc), (int)cols, eps);
    auto h_at = [&](int64_t c) -> S { return Elem<DT>::from_f32(rms_h<DT>(Elem<DT>::to_f32(xr[c]), Elem<DT>::to_f32(wr[c]), rs)); };
    float amax = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        const S h = h_at(c);
        if (h_out) reinterpret_cast<S*>(h_out)[row * ldh + c] = h;
        amax = amax_step(amax, Elem<DT>::to_f32(h));
    }
    amax = wave_max(amax);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) amax = amax_merge(amax, part[w]);
    const float s = scale_of(amax);
    if (threadIdx.x == 0) scale[row] = s;
    int8_t* qr = q + row * ldq;
    for (int64_t c = threadIdx.x; c < cols; c += 256) qr[c] = (int8_t)code_of(Elem<DT>::to_f32(h_at(c)), s);
}

// dev/test kernel: every 16-bit pattern g of the fast-division domain (0 < |g| <= 86) through the three division forms, u = 1 (so h is the
// stored silu(g)).  out[0] += patterns in the domain, out[1] += patterns whose SHORT result differs from the two-correction form,
// out[2] += patterns whose two-correction form differs from true division.
template <int DT>
__global__ __launch_bounds__(256) void silu_short_check(unsigned long long* __restrict__ out) {
    const uint32_t pat = blockIdx.x * 256u + threadIdx.x;            // 256 blocks x 256 threads = all 65 536 patterns
    const uint32_t mag = pat & 0x7FFFu;
    if (!silu_fast_div_ok<DT>(mag, mag)) return;
    const uint32_t one = DT == PQ_BF16 ? 0x3F80u : 0x3C00u;
    const v4u gv = v4u{pat | (pat << 16), pat | (pat << 16), pat | (pat << 16), pat | (pat << 16)};
    const v4u uv = v4u{one | (one << 16), one | (one << 16), one | (one << 16), one | (one << 16)};
    const v4u a = silu_mul_vec<DT, true, true>(gv, uv), b = silu_mul_vec<DT, true, false>(gv, uv), c = silu_mul_vec<DT, false, false>(gv, uv);
    atomicAdd(&out[0], 1ull);
    if (a[0] != b[0]) atomicAdd(&out[1], 1ull);
    if (b[0] != c[0]) atomicAdd(&out[2], 1ull);
}
void launch_silu_short_check(int dtype, unsigned long long* out, hipStream_t st) {
    if (dtype == PQ_BF16) silu_short_check<PQ_BF16><<<dim3(256), dim3(256), 0, st>>>(out);
    else silu_short_check<PQ_FP16><<<dim3(256), dim3(256), 0, st>>>(out);
}

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// widest row (in 16-byte vectors) of the wave-per-row RMSNorm layout (pq_set_option("PQ_RMS_WAVE_MAX"), 0..512).  Round 1 used it up to
// 512 vectors (a 4096-wide bf16 hidden state): 118 VGPRs, 4 waves per SIMD.  Measured in round 2 (profiles/r02_k1n_layout.txt): at 512
// vectors the 256-thread block per row (2 vectors per thread, ~44 VGPRs) is 10 % faster at 4096 rows (15.7 -> 14.2 us) and equal at 16384;
// at 256 vectors the wave layout wins (8.7 vs 9.6 us).  Same bits either way (QSPEC N1-N3 pins the order of the sum).

template <int DT, int TPR, bool WRITE_H, int MODE = 0, bool IDENT = false>
static void launch_silu_mul_vec(int vpt, const uint8_t* g, int64_t ldg_b, const uint8_t* u, int64_t ldu_b, int64_t rows, int nvec,
                                int8_t* q, int64_t ldq, float* scale, uint8_t* h, int64_t ldh_b, hipStream_t st, uint32_t* amax_io = nullptr) {
    constexpr int BS = TPR > 256 ? TPR : 256, RPB = BS / TPR;
    const dim3 grid((unsigned)((rows + RPB - 1) / RPB)), block(BS);
    switch (vpt) {
        case 1: silu_mul_quant_vec<DT, 1, TPR, WRITE_H, MODE, IDENT><<<grid, block, 0, st>>>(g, ldg_b, u, ldu_b, rows, nvec, q, ldq, scale, h, ldh_b, amax_io); break;
        case 2: silu_mul_quant_vec<DT, 2, TPR, WRITE_H, MODE, IDENT><<<grid, block, 0, st>>>(g, ldg_b, u, ldu_b, rows, nvec, q, ldq, scale, h, ldh_b, amax_io); break;
        case 3:
            if constexpr (TPR == 512) silu_mul_quant_vec<DT, 3, TPR, WRITE_H, MODE, IDENT><<<grid, block, 0, st>>>(g, ldg_b, u, ldu_b, rows, nvec, q, ldq, scale, h, ldh_b, amax_io);
            break;
        case 4: silu_mul_quant_vec<DT, 4, TPR, WRITE_H, MODE, IDENT><<<grid, block, 0, st>>>(g, ldg_b, u, ldu_b, rows, nvec, q, ldq, scale, h, ldh_b, amax_io); break;
        case 8:
            if constexpr (TPR != 512) silu_mul_quant_vec<DT, 8, TPR, WRITE_H, MODE, IDENT><<<grid, block, 0, st>>>(g, ldg_b, u, ldu_b, rows, nvec, q, ldq, scale, h, ldh_b, amax_io);
            break;
        default:
            if constexpr (TPR == 256) silu_mul_quant_vec<DT, 16, TPR, WRITE_H, MODE, IDENT><<<grid, block, 0, st>>>(g, ldg_b, u, ldu_b, rows, nvec, q, ldq, scale, h, ldh_b, amax_io);
            break;
    }
}

// the two halves of K1s for a column-sharded intermediate (MODE 1: row amax of these columns -> amax_io; MODE 2: encode against the amax in amax_io).
// Same layouts as the fused kernel (the h vectors are recomputed in MODE 2 — g and u are read twice, from the Infinity Cache the second time — instead of
// parking a 16-bit h in HBM between the two passes: the same 9 bytes per element either way, and no extra buffer).
template <int DT, int MODE, bool IDENT>
void silu_mul_split_dispatch(const void* g, int64_t ldg, const void* u, int64_t ldu, int64_t rows, int64_t cols, uint32_t* amax_io, int8_t* q,
                             int64_t ldq, float* scale, hipStream_t st) {
    static_assert(MODE == 1 || MODE == 2, "split modes");
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    if constexpr (IDENT) { u = g; ldu = ldg; }          // (never read)
    const bool vec_ok = cols > 0 && (cols % EPV == 0) && (ldg % EPV == 0) && (ldu % EPV == 0) && aligned_to(g, 16) && aligned_to(u, 16) &&
                        cols / EPV <= 256 * 16 && (MODE == 1 || ((ldq % EPV == 0) && aligned_to(q, EPV)));
    if (!vec_ok) {
        silu_mul_quant_generic<DT, MODE, IDENT><<<dim3((unsigned)rows), dim3(256), 0, st>>>(g, ldg, u, ldu, cols, q, ldq, scale, nullptr, 0, amax_io);
        return;
    }
    const int nvec = (int)(cols / EPV);
    auto pow2 = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
    const uint8_t* gb = reinterpret_cast<const uint8_t*>(g);
    const uint8_t* ub = reinterpret_cast<const uint8_t*>(u);
    const int64_t kb = Elem<DT>::kBytes;
    if (nvec <= 64 * 4) launch_silu_mul_vec<DT, 64, false, MODE, IDENT>(pow2((nvec + 63) / 64), gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, nullptr, 0, st, amax_io);
    else if (!IDENT && nvec > 1024 && nvec <= 1536 && opt().silu_tpr != 256) launch_silu_mul_vec<DT, 512, false, MODE, IDENT>((nvec + 511) / 512, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, nullptr, 0, st, amax_io);
    else launch_silu_mul_vec<DT, 256, false, MODE, IDENT>(pow2((nvec + 255) / 256), gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, nullptr, 0, st, amax_io);
}

template <int DT>
void silu_mul_quant_dispatch(const void* g, int64_t ldg, const void* u, int64_t ldu, int64_t rows, int64_t cols, int8_t* q,
                             int64_t ldq, float* scale, void* h_out, int64_t ldh, hipStream_t st) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const bool vec_ok = cols > 0 && (cols % EPV == 0) && (ldg % EPV == 0) && (ldu % EPV == 0) && aligned_to(g, 16) && aligned_to(u, 16) &&
                        (ldq % EPV == 0) && aligned_to(q, EPV) && cols / EPV <= 256 * 16 &&
                        (!h_out || ((ldh % EPV == 0) && aligned_to(h_out, 16)));
    if (!vec_ok) {
        silu_mul_quant_generic<DT><<<dim3((unsigned)rows), dim3(256), 0, st>>>(g, ldg, u, ldu, cols, q, ldq, scale, h_out, ldh);
        return;
    }
    const int nvec = (int)(cols / EPV);
    auto pow2 = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
    const uint8_t* gb = reinterpret_cast<const uint8_t*>(g);
    const uint8_t* ub = reinterpret_cast<const uint8_t*>(u);
    uint8_t* hb = reinterpret_cast<uint8_t*>(h_out);
    const int64_t kb = Elem<DT>::kBytes;
    if (nvec <= 64 * 4) {
        const int vpt = pow2((nvec + 63) / 64);
        if (h_out) launch_silu_mul_vec<DT, 64, true>(vpt, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, hb, ldh * kb, st);
        else launch_silu_mul_vec<DT, 64, false>(vpt, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, hb, 0, st);
    } else if (nvec > 1024 && nvec <= 1536 && opt().silu_tpr != 256) {
        // rows of 1025..1536 vectors (e.g. 11008 columns): 512 threads x 3 vectors fill 90 % of their slots where 256 threads x 8
        // fill 67 % (the kernel is VALU-bound, idle slots are idle lanes): 2048 x 11008 30.4 -> 28.1 us.  At 1537..2048 vectors
        // (14336, 16384 columns) both layouts have the same slots and measured the same (profiles/r02_k1s_threads_per_row.txt).
        const int vpt = (nvec + 511) / 512;        // 3
        if (h_out) launch_silu_mul_vec<DT, 512, true>(vpt, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, hb, ldh * kb, st);
        else launch_silu_mul_vec<DT, 512, false>(vpt, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, hb, 0, st);
    } else {
        const int vpt = pow2((nvec + 255) / 256);
        if (h_out) launch_silu_mul_vec<DT, 256, true>(vpt, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, hb, ldh * kb, st);
        else launch_silu_mul_vec<DT, 256, false>(vpt, gb, ldg * kb, ub, ldu * kb, rows, nvec, q, ldq, scale, hb, 0, st);
    }
}

template <int DT>
void rmsnorm_quant_dispatch(const void* x, int64_t ldx, const void* wgt, float eps, int64_t rows, int64_t cols, int8_t* q, int64_t ldq,
                            float* scale, void* h_out, int64_t ldh, hipStream_t st) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const bool vec_ok = (cols % EPV == 0) && (ldx % EPV == 0) && aligned_to(x, 16) && aligned_to(wgt, 16) && (ldq % EPV == 0) &&
                        aligned_to(q, EPV) && cols / EPV <= 256 * 16 && (!h_out || ((ldh % EPV == 0) && aligned_to(h_out, 16)));
    const dim3 grid((unsigned)rows), block(256);
    if (!vec_ok) {
        rmsnorm_quant_generic<DT><<<grid, block, 0, st>>>(x, ldx, wgt, eps, cols, q, ldq, scale, h_out, ldh);
        return;
    }
    const int nvec = (int)(cols / EPV);
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(wgt);
    uint8_t* hb = reinterpret_cast<uint8_t*>(h_out);
    const int64_t kb = Elem<DT>::kBytes;
    if (nvec <= opt().rms_wave_max) {         // one wave per row (rmsnorm_quant_wave): VPT in {4, 8} keeps i & 3 meaningful
        const dim3 wgrid((unsigned)((rows + 3) / 4));
#define PQ_RMSW_LAUNCH(V)                                                                                                                 \
    do {                                                                                                                              \
        if (h_out) rmsnorm_quant_wave<DT, V, true><<<wgrid, block, 0, st>>>(xb, ldx * kb, wb, eps, (int)cols, nvec, rows, q, ldq, scale, hb, ldh * kb); \
        else rmsnorm_quant_wave<DT, V, false><<<wgrid, block, 0, st>>>(xb, ldx * kb, wb, eps, (int)cols, nvec, rows, q, ldq, scale, hb, 0);              \
    } while (0)
        if (nvec <= 64) PQ_RMSW_LAUNCH(1);
        else if (nvec <= 128) PQ_RMSW_LAUNCH(2);
        else if (nvec <= 256) PQ_RMSW_LAUNCH(4);
        else PQ_RMSW_LAUNCH(8);
#undef PQ_RMSW_LAUNCH
        return;
    }
    int vpt = 1;
    while (vpt * 256 < nvec) vpt <<= 1;
#define PQ_RMS_LAUNCH(V)                                                                                                              \
    do {                                                                                                                              \
        if (h_out) rmsnorm_quant_vec<DT, V, true><<<grid, block, 0, st>>>(xb, ldx * kb, wb, eps, (int)cols, nvec, q, ldq, scale, hb, ldh * kb); \
        else rmsnorm_quant_vec<DT, V, false><<<grid, block, 0, st>>>(xb, ldx * kb, wb, eps, (int)cols, nvec, q, ldq, scale, hb, 0);              \
    } while (0)
    switch (vpt) {
        case 1: PQ_RMS_LAUNCH(1); break;
        case 2: PQ_RMS_LAUNCH(2); break;
        case 4: PQ_RMS_LAUNCH(4); break;
        case 8: PQ_RMS_LAUNCH(8); break;
        default: PQ_RMS_LAUNCH(16); break;
    }
#undef PQ_RMS_LAUNCH
}

template void rmsnorm_quant_dispatch<PQ_BF16>(const void*, int64_t, const void*, float, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);
template void rmsnorm_quant_dispatch<PQ_FP16>(const void*, int64_t, const void*, float, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);
template void rmsnorm_quant_dispatch<PQ_F32>(const void*, int64_t, const void*, float, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);

#define PQ_SPLIT_INST(DT) \
    template void silu_mul_split_dispatch<DT, 1, false>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, uint32_t*, int8_t*, int64_t, float*, hipStream_t); \
    template void silu_mul_split_dispatch<DT, 2, false>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, uint32_t*, int8_t*, int64_t, float*, hipStream_t); \
    template void silu_mul_split_dispatch<DT, 1, true>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, uint32_t*, int8_t*, int64_t, float*, hipStream_t); \
    template void silu_mul_split_dispatch<DT, 2, true>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, uint32_t*, int8_t*, int64_t, float*, hipStream_t);
PQ_SPLIT_INST(PQ_BF16) PQ_SPLIT_INST(PQ_FP16) PQ_SPLIT_INST(PQ_F32)
#undef PQ_SPLIT_INST
template void silu_mul_quant_dispatch<PQ_BF16>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);
template void silu_mul_quant_dispatch<PQ_FP16>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);
template void silu_mul_quant_dispatch<PQ_F32>(const void*, int64_t, const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);

}  // namespace pq
