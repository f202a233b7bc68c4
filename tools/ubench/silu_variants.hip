// Dev microbenchmark / explorer: for bf16 and fp16 rows the STORED silu(g) is a function of the 16-bit g alone, so any cheaper instruction sequence
// that stores the same value for every pattern of the fast-division domain (0 < |g| <= 86) is as exact as the specification's.  Enumerates
// candidate sequences against the product path (two-correction division on the specified exp) and prints the mismatching patterns.
#include "../../protoquant_amd/csrc/producer_kernels.hip"
#include <cstdio>
#include <vector>
using namespace pq;

template <int DT> __device__ uint32_t stored_bits(float sg) {      // round to storage, return the 16-bit pattern
    const v2f v = v2f{sg, sg};
    return Pair<DT>::pack(v) & 0xFFFFu;
}
__device__ __forceinline__ float exp_spec(float tc, int terms, bool two_step) {   // the specification's exp(tc) with fewer polynomial terms / one reduction step
    float n = __builtin_rintf(tc * __builtin_bit_cast(float, 0x3FB8AA3Bu));
    float r = __builtin_fmaf(n, -__builtin_bit_cast(float, 0x3F317200u), tc);
    if (two_step) r = __builtin_fmaf(n, -__builtin_bit_cast(float, 0x35BFBE8Eu), r);
    const uint32_t kC[8] = {0x39500D01u, 0x3AB60B61u, 0x3C088889u, 0x3D2AAAABu, 0x3E2AAAABu, 0x3F000000u, 0x3F800000u, 0x3F800000u};
    float p = __builtin_bit_cast(float, kC[8 - terms]);
    for (int c = 8 - terms + 1; c < 8; ++c) p = __builtin_fmaf(p, r, __builtin_bit_cast(float, kC[c]));
    return __builtin_ldexpf(p, (int)n);
}
template <int DT>
__global__ void explore(int variant, unsigned long long* out, uint32_t* bad_list) {
    const uint32_t pat = blockIdx.x * 256u + threadIdx.x;
    const uint32_t mag = pat & 0x7FFFu;
    if (!silu_fast_div_ok<DT>(mag, mag)) return;
    const uint32_t one = DT == PQ_BF16 ? 0x3F80u : 0x3C00u;
    const v4u gv = v4u{pat | (pat << 16), 0, 0, 0}, uv = v4u{one | (one << 16), 0, 0, 0};
    const uint32_t want = silu_mul_vec<DT, true, false>(gv, uv)[0] & 0xFFFFu;
    const float g = Elem<DT>::to_f32((uint16_t)pat);
    float tc = -g, e, sg;
    switch (variant) {
        default:
        case 0: tc = __builtin_amdgcn_fmed3f(-g, -30.0f, 100.0f); e = exp_spec(tc, 8, true); break;      // the specification (sanity: 0 mismatches)
        case 1: e = exp_spec(tc, 8, true); break;                                                          // no clamp
        case 2: e = exp_spec(tc, 8, false); break;                                                         // + one reduction step
        case 3: e = exp_spec(tc, 7, true); break;                                                          // no clamp, one polynomial term less
        case 4: e = exp_spec(tc, 6, true); break;
        case 5: e = exp_spec(tc, 5, true); break;
        case 6: case 7: case 8: e = __builtin_amdgcn_exp2f(tc * __builtin_bit_cast(float, 0x3FB8AA3Bu)); break;   // hardware exp2
        case 9: e = exp_spec(tc, 8, true); break;
    }
    const float d = 1.0f + e;
    const float y0 = __builtin_amdgcn_rcpf(d);
    float q = g * y0;
    if (variant != 7 && variant != 9) { const float r = __builtin_fmaf(-d, q, g); q = __builtin_fmaf(r, y0, q); }   // one correction (7, 9: none)
    if (variant == 8) { const float r = __builtin_fmaf(-d, q, g); q = __builtin_fmaf(r, y0, q); }                     // 8: two corrections on the hardware exp
    sg = q;
    const uint32_t got = stored_bits<DT>(sg);
    atomicAdd(&out[0], 1ull);
    if (got != want) { const unsigned long long k = atomicAdd(&out[1], 1ull); if (k < 16) bad_list[k] = pat; }
}
int main() {
    unsigned long long* out; uint32_t* bad;
    (void)hipMalloc(&out, 16); (void)hipMalloc(&bad, 64);
    const char* names[] = {"specification sequence, one correction (shipped)", "no clamp", "no clamp, ONE reduction step", "no clamp, 7 polynomial terms", "no clamp, 6 terms",
                           "no clamp, 5 terms", "hardware exp2 (v_exp_f32), one correction", "hardware exp2, NO correction", "hardware exp2, two corrections", "specification exp, NO correction"};
    for (int dt = 0; dt < 2; ++dt)
        for (int v = 0; v < 10; ++v) {
            (void)hipMemset(out, 0, 16); (void)hipMemset(bad, 0, 64);
            if (dt == 0) explore<PQ_BF16><<<256, 256>>>(v, out, bad); else explore<PQ_FP16><<<256, 256>>>(v, out, bad);
            unsigned long long h[2]; uint32_t hb[16];
            (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost);
            printf("%s  variant %d %-52s patterns %llu  mismatches %llu ", dt == 0 ? "bf16" : "fp16", v, names[v], h[0], h[1]);
            for (unsigned long long k = 0; k < h[1] && k < 8; ++k) printf(" 0x%04x", hb[k]);
            printf("\n");
        }
    return 0;
}
