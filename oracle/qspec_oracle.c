/* TEST INFRASTRUCTURE — plain-C restatement of QSPEC v2.
 *
 * Parity status: PARITY UNPINNED BY THE REFERENCE.  /root/reference holds no source, tests or
 * golden vectors for the dynamic-int8 linear path (only /root/reference/CODE_OF_CONDUCT.md:1-80),
 * so no function below can cite a reference file:line.  Each function instead cites the QSPEC
 * clause it follows; the integer GEMM restates torch._int_mm (aten::_int_mm, third-party PyTorch
 * 2.10.0, git 449b1768), which is exact integer arithmetic.  Pinned against tests/golden/ (made
 * by oracle/torch_ref.py around torch._int_mm) in tests/test_oracle.py.
 *
 * Build: oracle/Makefile -> oracle/liboracle.so   (gcc -O2 -ffp-contract=off -fopenmp; NO -ffast-math)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { OQ_BF16 = 0, OQ_FP16 = 1, OQ_F32 = 2 };

/* ---- dtype plumbing (exact up-conversion, RNE down-conversion) ---- */
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f;
}
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if (f != f) return (uint16_t)((u >> 16) | 0x0040u);           /* keep NaN a quiet NaN */
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);    /* round-to-nearest-even */
}
/* IEEE binary16 <-> binary32 by hand (gcc 11 has no _Float16 on x86-64) */
static inline float fp16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3FFu; u = sign | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
static inline uint16_t f32_to_fp16(float f) {        /* round-to-nearest-even, overflow -> inf */
    uint32_t u; memcpy(&u, &f, 4);
    uint16_t sign = (uint16_t)((u >> 16) & 0x8000u); uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((a >> 13) & 0x3FFu));   /* NaN */
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                          /* >= 65520 -> inf */
    if (a < 0x33000001u) return sign;                                                 /* <= 2^-25 -> 0 */
    int32_t e = (int32_t)(a >> 23) - 127; uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                                    /* subnormal: extra shift */
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    uint32_t he = (e < -14) ? 0u : (uint32_t)(e + 15);
    /* q holds the implicit bit for normals: adding (he-1)<<10 lets a mantissa carry bump the exponent */
    uint32_t out = (e < -14) ? q : (((he - 1u) << 10) + q);
    return (uint16_t)(sign | out);
}

static inline float load_f32(const void* p, int dtype, int64_t i) {
    switch (dtype) {
        case OQ_BF16: return bf16_to_f32(((const uint16_t*)p)[i]);
        case OQ_FP16: return fp16_to_f32(((const uint16_t*)p)[i]);
        default:      return ((const float*)p)[i];
    }
}
static inline void store_f32(void* p, int dtype, int64_t i, float v) {
    switch (dtype) {
        case OQ_BF16: ((uint16_t*)p)[i] = f32_to_bf16(v); break;
        case OQ_FP16: ((uint16_t*)p)[i] = f32_to_fp16(v); break;
        default:      ((float*)p)[i] = v; break;
    }
}

/* QSPEC Q2: running max of |x| that PROPAGATES a NaN (as torch.amax does): once NaN, always NaN */
static inline float amax_step(float amax, float v) { float a = fabsf(v); return (a > amax || a != a) ? a : amax; }
/* QSPEC Q3: scale = amax/127 (true division), 1.0 when amax == 0; a NaN scale is the canonical quiet NaN 0x7FC00000 */
static inline float scale_of(float amax) {
    if (amax != amax) { const uint32_t u = 0x7FC00000u; float f; memcpy(&f, &u, 4); return f; }
    return amax == 0.0f ? 1.0f : amax / 127.0f;
}
/* QSPEC Q4-Q6: q = clamp(rne(x/scale)), NaN -> 0 */
static inline int8_t code_of(float x, float scale) {
    float t = rintf(x / scale);            /* default rounding mode: half-to-even */
    if (t != t) return 0;
    if (t > 127.0f) t = 127.0f;
    if (t < -128.0f) t = -128.0f;
    return (int8_t)t;
}

/* QSPEC quantize, reduce over columns (per-token). x[rows, cols] with leading dimension ldx. */
void oq_quant_rowwise(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ldx,
                      int8_t* q, int64_t ldq, float* scale) {
    #pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float amax = 0.0f;
        for (int64_t c = 0; c < cols; ++c) amax = amax_step(amax, load_f32(x, dtype, r * ldx + c));
        float s = scale_of(amax);
        scale[r] = s;
        for (int64_t c = 0; c < cols; ++c) q[r * ldq + c] = code_of(load_f32(x, dtype, r * ldx + c), s);
    }
}

/* QSPEC quantize, reduce over rows (per-channel of a row-major [rows, cols] matrix). */
void oq_quant_colwise(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ldx,
                      int8_t* q, int64_t ldq, float* scale) {
    #pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < cols; ++c) {
        float amax = 0.0f;
        for (int64_t r = 0; r < rows; ++r) amax = amax_step(amax, load_f32(x, dtype, r * ldx + c));
        scale[c] = scale_of(amax);
    }
    #pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) q[r * ldq + c] = code_of(load_f32(x, dtype, r * ldx + c), scale[c]);
}

/* QSPEC dequantize: out = cast_rne(f32(q) * scale[kept axis]); axis = reduced axis (1 rows-scale, 0 cols-scale) */
void oq_dequant(const int8_t* q, int64_t ldq, const float* scale, int axis, int64_t rows, int64_t cols,
                void* out, int64_t ldo, int out_dtype) {
    #pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c)
            store_f32(out, out_dtype, r * ldo + c, (float)q[r * ldq + c] * (axis == 0 ? scale[c] : scale[r]));
}

/* restates torch._int_mm(a, b.t()): c[m,n] = sum_k a[m,k]*b[n,k], exact int32 */
void oq_gemm_s8s8s32(const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb,
                     int32_t* c, int64_t ldc, int64_t M, int64_t N, int64_t K) {
    #pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            const int8_t* ar = a + m * lda; const int8_t* br = b + n * ldb;
            int32_t acc = 0;
            for (int64_t k = 0; k < K; ++k) acc += (int32_t)ar[k] * (int32_t)br[k];
            c[m * ldc + n] = acc;
        }
}

/* QSPEC epilogue E1-E4 on top of the exact accumulator */
void oq_qlinear_s8(const int8_t* a, int64_t lda, const float* a_scale,
                   const int8_t* b, int64_t ldb, const float* b_scale,
                   const void* bias, void* y, int64_t ldy, int out_dtype,
                   int64_t M, int64_t N, int64_t K) {
    #pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            const int8_t* ar = a + m * lda; const int8_t* br = b + n * ldb;
            int32_t acc = 0;
            for (int64_t k = 0; k < K; ++k) acc += (int32_t)ar[k] * (int32_t)br[k];
            float t = (float)acc;              /* E1: RNE int32 -> f32 */
            t = t * a_scale[m];                /* E2: row scale first  */
            t = t * b_scale[n];                /* E3: then column scale */
            if (bias) t = t + load_f32(bias, out_dtype, n);   /* E4: separate rounded add */
            store_f32(y, out_dtype, m * ldy + n, t);
        }
}

/* ---- QSPEC S1-S6: silu(g)*u -> per-token quantisation, the producer-fused form of
 *      quantize(F.silu(g) * u)   (BASELINE config 3 names the silu*mul between up and down).
 * The exponential is SPECIFIED (not "libm's exp"), so every implementation produces the same bits:
 * Cody-Waite reduction + degree-7 Taylor polynomial, all in binary32 with correctly rounded fmaf. */
static inline float f32_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float oq_exp_spec(float t) {
    /* S1: clamp (a NaN fails both compares and passes through) */
    float tc = t < -30.0f ? -30.0f : t;
    tc = tc > 100.0f ? 100.0f : tc;
    /* S2: n = rne(tc * log2(e));  r = tc - n*ln2 in two fma steps (ln2 = hi + lo, hi has 9 trailing zero bits) */
    const float n = rintf(tc * f32_bits(0x3FB8AA3Bu));
    float r = fmaf(n, -f32_bits(0x3F317200u), tc);
    r = fmaf(n, -f32_bits(0x35BFBE8Eu), r);
    /* S3: e^r = 1 + r(1 + r(1/2 + r(1/6 + r(1/24 + r(1/120 + r(1/720 + r/5040)))))), Horner with fmaf */
    float p = f32_bits(0x39500D01u);
    p = fmaf(p, r, f32_bits(0x3AB60B61u));
    p = fmaf(p, r, f32_bits(0x3C088889u));
    p = fmaf(p, r, f32_bits(0x3D2AAAABu));
    p = fmaf(p, r, f32_bits(0x3E2AAAABu));
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    /* S4: scale by 2^n in two exact steps (n in [-43, 144]; the second may overflow to +inf, as exp does) */
    const int32_t ni = (n == n) ? (int32_t)n : 0;
    const int32_t n1 = ni >> 1, n2 = ni - n1;          /* arithmetic shift: floor(n/2) */
    return (p * f32_bits((uint32_t)(n1 + 127) << 23)) * f32_bits((uint32_t)(n2 + 127) << 23);
}

/* S5: h = cast(f32(cast(g / (1 + exp_spec(-g)))) * f32(u)) — the storage-dtype rounding after silu and after the
 * product mirrors the two eager ops F.silu(g) and (...) * u. */
static inline float silu_mul_spec(float g, float u, int dtype) {
    const float d = 1.0f + oq_exp_spec(-g);
    float sg = g / d;
    if (dtype == OQ_BF16) sg = bf16_to_f32(f32_to_bf16(sg));
    else if (dtype == OQ_FP16) sg = fp16_to_f32(f32_to_fp16(sg));
    return sg * u;    /* the caller rounds the product to the storage dtype */
}

/* S6: Q1-Q6 on the rows of h.  g[rows, cols] (ld ldg), u[rows, cols] (ld ldu), same dtype; h_out nullable. */
void oq_silu_mul_quant_rowwise(const void* g, int64_t ldg, const void* u, int64_t ldu, int dtype, int64_t rows,
                               int64_t cols, int8_t* q, int64_t ldq, float* scale, void* h_out, int64_t ldh) {
    #pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float amax = 0.0f;
        for (int pass = 0; pass < 2; ++pass) {
            const float s = scale_of(amax);
            if (pass == 1) scale[r] = s;
            for (int64_t c = 0; c < cols; ++c) {
                float h = silu_mul_spec(load_f32(g, dtype, r * ldg + c), load_f32(u, dtype, r * ldu + c), dtype);
                if (dtype == OQ_BF16) h = bf16_to_f32(f32_to_bf16(h));
                else if (dtype == OQ_FP16) h = fp16_to_f32(f32_to_fp16(h));
                if (pass == 0) {
                    amax = amax_step(amax, h);
                    if (h_out) store_f32(h_out, dtype, r * ldh + c, h);
                } else q[r * ldq + c] = code_of(h, s);
            }
        }
    }
}

/* ---- QSPEC N1-N6: RMSNorm(x; weight, eps) -> per-token quantisation, the producer-fused form of
 *      quantize(weight * (x.float() * rsqrt(mean(x.float()^2) + eps)).to(dtype))     
 * The sum of squares is a float reduction, so its ORDER is part of the specification (any fixed order is as accurate
 * as any other; pinning one makes every implementation agree bit for bit):
 *   N1  the row is cut into 16-byte vectors of EPV = 16/sizeof(dtype) elements; vector v belongs to lane v mod 256
 *   N2  each lane accumulates its vectors in increasing v, elements in increasing index:  acc = fma(x, x, acc)
 *   N3  lanes combine inside each group of 64 by an xor butterfly (offsets 32,16,8,4,2,1: s = s + s[lane ^ off]),
 *       then the four group sums left to right:  ss = ((s0 + s1) + s2) + s3
 *   N4  var = ss / f32(C);   rs = 1 / sqrt(var + eps)          (IEEE sqrt and division)
 *   N5  xn = cast(f32(x) * rs);   h = cast(f32(weight) * f32(xn))   (storage rounding after each, as the eager ops do)
 *   N6  Q1-Q6 on the rows of h */
static float rms_sumsq_spec(const void* x, int dtype, int64_t base, int64_t cols) {
    const int epv = dtype == OQ_F32 ? 4 : 8;
    float lane[256];
    for (int l = 0; l < 256; ++l) lane[l] = 0.0f;
    const int64_t nvec = (cols + epv - 1) / epv;
    for (int64_t v = 0; v < nvec; ++v) {
        float acc = lane[v & 255];
        for (int e = 0; e < epv && v * epv + e < cols; ++e) {
            const float xv = load_f32(x, dtype, base + v * epv + e);
            acc = fmaf(xv, xv, acc);
        }
        lane[v & 255] = acc;
    }
    float grp[4];
    for (int gidx = 0; gidx < 4; ++gidx) {
        float s[64], t[64];
        for (int l = 0; l < 64; ++l) s[l] = lane[gidx * 64 + l];
        for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ off];
            for (int l = 0; l < 64; ++l) s[l] = t[l];
        }
        grp[gidx] = s[0];
    }
    return ((grp[0] + grp[1]) + grp[2]) + grp[3];
}

static inline float round_store(float v, int dtype) {
    if (dtype == OQ_BF16) return bf16_to_f32(f32_to_bf16(v));
    if (dtype == OQ_FP16) return fp16_to_f32(f32_to_fp16(v));
    return v;
}

/* x[rows, cols] (ld ldx), weight[cols], both of `dtype`; h_out nullable; rs_out nullable (rows floats, for tests). */
void oq_rmsnorm_quant_rowwise(const void* x, int64_t ldx, const void* weight, float eps, int dtype, int64_t rows, int64_t cols,
                              int8_t* q, int64_t ldq, float* scale, void* h_out, int64_t ldh, float* rs_out) {
    #pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float ss = rms_sumsq_spec(x, dtype, r * ldx, cols);
        const float var = ss / (float)cols;
        const float rs = 1.0f / sqrtf(var + eps);
        if (rs_out) rs_out[r] = rs;
        float amax = 0.0f;
        for (int pass = 0; pass < 2; ++pass) {
            const float s = scale_of(amax);
            if (pass == 1) scale[r] = s;
            for (int64_t c = 0; c < cols; ++c) {
                const float xn = round_store(load_f32(x, dtype, r * ldx + c) * rs, dtype);
                const float h = round_store(load_f32(weight, dtype, c) * xn, dtype);
                if (pass == 0) {
                    amax = amax_step(amax, h);
                    if (h_out) store_f32(h_out, dtype, r * ldh + c, h);
                } else q[r * ldq + c] = code_of(h, s);
            }
        }
    }
}
