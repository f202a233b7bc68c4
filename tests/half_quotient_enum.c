/* Test infrastructure (host, plain C): enumerates the WHOLE domain of the one-step division-free encode that 16-bit inputs take in the
 * quantising kernels (protoquant_amd/csrc/quant_device.h: quotient_fast1) and counts the pairs whose code differs from QSPEC Q4-Q6
 * (true fp32 division, rintf).  Domain: the row's amax is a bf16 / fp16 value, so s = amax / 127 takes one of 2^15 values, and every
 * element is one of the magnitudes <= amax.  Prints "<pairs> <mismatch_0_steps> <mismatch_1_step> <mismatch_2_steps>".
 * usage: half_quotient_enum bf16|fp16      (compile with -O2 -mfma -ffp-contract=off; fmaf must be the correctly rounded one) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bf16_value(uint16_t b) { return from_bits((uint32_t)b << 16); }
static float fp16_value(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u;
    uint32_t m = h & 1023u;
    if (e == 31u) return from_bits(sign | 0x7F800000u | (m << 13));
    if (e != 0u) return from_bits(sign | ((e + 112u) << 23) | (m << 13));
    if (m == 0u) return from_bits(sign);
    int shift = 0;                                   /* subnormal: renormalise */
    while (!(m & 1024u)) { m <<= 1; ++shift; }
    return from_bits(sign | ((uint32_t)(113 - shift) << 23) | ((m & 1023u) << 13));
}

int main(int argc, char** argv) {
    const int fp16 = argc > 1 && strcmp(argv[1], "fp16") == 0;
    const float magic = 12582912.0f; /* 1.5 * 2^23 */
    const int top = fp16 ? 0x7C00 : 0x7F80; /* first non-finite pattern */
    unsigned long long pairs = 0, bad[3] = {0, 0, 0};
    for (int a = 1; a < top; ++a) {
        const float amax = fp16 ? fp16_value((uint16_t)a) : bf16_value((uint16_t)a);
        volatile float sv = amax / 127.0f;
        const float s = sv;
        const uint32_t sb = to_bits(s), e = sb >> 23;
        if (!(e >= 67u && e <= 187u && (sb & 0x7FFFFFu) != 0x7FFFFFu)) continue; /* scale_fast_ok: other scales take true division */
        volatile float rv = 1.0f / s;
        const float r = rv;
        for (int xb = 0; xb <= a; ++xb) {
            for (int sign = 0; sign < 2; ++sign) {
                const uint16_t pat = (uint16_t)(xb | (sign << 15));
                const float x = fp16 ? fp16_value(pat) : bf16_value(pat);
                volatile float qd = x / s;
                const int want = (int)rintf(qd);
                float q[3];
                q[0] = x * r;
                q[1] = fmaf(fmaf(-q[0], s, x), r, q[0]);
                q[2] = fmaf(fmaf(-q[1], s, x), r, q[1]);
                ++pairs;
                for (int k = 0; k < 3; ++k)
                    if ((int)(int8_t)(to_bits(q[k] + magic) & 255u) != want) ++bad[k];
            }
        }
    }
    printf("%llu %llu %llu %llu\n", pairs, bad[0], bad[1], bad[2]);
    return 0;
}
