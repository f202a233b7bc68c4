// quant_kernels.hip — K1 per-token row quant, K2 per-channel column quant, dequant (gfx950).
// All three are HBM-bound byte movers: 16-byte coalesced loads, one read of x for K1 (the row lives
// in registers between the amax reduction and the encode), wavefront shuffles + one LDS hop for the
// reductions.  Arithmetic follows QSPEC v2 exactly (true fp32 division, RNE, no contraction).
#include <type_traits>
#include "quant_device.h"

namespace pq {

// ------------------------------------------------------------------------------------------------
// K1 vector path.  TPR threads own one row; thread t holds 16-byte vectors t, t+TPR, ... (VPT of
// them) so every wave-instruction reads 1 KiB contiguous.  The row stays in registers between the amax
// reduction and the encode: ONE HBM read.  Algorithmic traffic: read once, write 1 B/elem + 4 B/row.
// RPW (TPR = 64 only): rows per wave; 2 issues both rows' loads up front (an experiment that measured slower: see
// launch_rowwise_vec).
// ST16 (16-bit inputs, TPR = 64, even nvec, 16-byte aligned code rows): 16-byte stores — adjacent lanes swap halves (DPP quad_perm [1,0,3,2]) so that an
// even lane holds the 16 consecutive codes of vector pair (idx, idx + 1) of segment i and its odd neighbour those of segment i + 1; loads stay dense.
template <int DT, int VPT, int TPR, int RPW = 1, bool ST16 = false>
__global__ __launch_bounds__(256) void quant_rowwise_vec(const uint8_t* __restrict__ x, int64_t rows, int nvec,
                                                         int64_t ldx_bytes, int8_t* __restrict__ q, int64_t ldq,
                                                         float* __restrict__ scale) {
    static_assert(RPW == 1 || TPR == kWave, "several rows per wave only in the wave-per-row layout");
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    constexpr int RPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t row0 = ((int64_t)blockIdx.x * RPB + threadIdx.x / TPR) * RPW;

    // Loads are UNCONDITIONAL (clamped address): a per-element "load or zero" select makes hipcc branch
    // around every load and wait vmcnt(0) each time.  Duplicates of the clamped tail vector do not change a max.
    v4u v[RPW][VPT];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int64_t row = row0 + rr < rows ? row0 + rr : rows - 1;
        const uint8_t* xr = x + row * ldx_bytes;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = i * TPR + t;
            v[rr][i] = *reinterpret_cast<const v4u*>(xr + (int64_t)(idx < nvec ? idx : nvec - 1) * 16);
        }
    }
    __shared__ uint32_t part[256 / kWave];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const bool active = row0 + rr < rows;           // only the last block can hold inactive rows
        const int64_t row = active ? row0 + rr : rows - 1;
        const uint8_t* xr = x + row * ldx_bytes;
        // ---- amax (Q2) on bit patterns
        uint32_t ab = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) ab = vec_amax_bits<DT>(v[rr][i], ab);
        ab = wave_max_u32(amax_acc_finish<DT>(ab));
        if constexpr (TPR > kWave) {
            if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = ab;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 256 / kWave; ++w) ab = part[w] > ab ? part[w] : ab;
        }
        const bool has_nan = amax_bits_has_nan<DT>(ab);      // QSPEC v2: a NaN propagates (scale = canonical NaN, codes 0 by the true-division path)
        const float s = scale_of(amax_bits_to_f32<DT>(ab));
        if (!active) { if constexpr (RPW == 1) return; else continue; }
        if (t == 0) scale[row] = s;
        int8_t* qr = q + row * ldq;
        auto store_vec = [&](int idx, const uint32_t (&pk)[EPV / 4]) {
            if constexpr (EPV == 8) store_wt_b64(qr + (int64_t)idx * 8, v2u{pk[0], pk[1]});
            else store_wt_b32(qr + (int64_t)idx * 4, pk[0]);
        };
        if (!has_nan && scale_fast_ok(s)) {       // the hot path: no division per element, results identical to x / s
            const float r = 1.0f / s;
            if constexpr (ST16 && EPV == 8 && TPR == kWave && (VPT % 2) == 0) {
                const bool odd = t & 1;
#pragma unroll
                for (int i = 0; i < VPT; i += 2) {
                    float fa[EPV], fb[EPV];
                    Unpack<DT, EPV>::run(v[rr][i], fa);
                    Unpack<DT, EPV>::run(v[rr][i + 1], fb);
                    uint32_t a[2], b[2];
                    fast_encode<EPV, kQuotientSteps<DT>>(fa, s, r, a);
                    fast_encode<EPV, kQuotientSteps<DT>>(fb, s, r, b);
                    const uint32_t na0 = __builtin_amdgcn_update_dpp(0u, a[0], 0xB1, 0xF, 0xF, false), na1 = __builtin_amdgcn_update_dpp(0u, a[1], 0xB1, 0xF, 0xF, false);
                    const uint32_t nb0 = __builtin_amdgcn_update_dpp(0u, b[0], 0xB1, 0xF, 0xF, false), nb1 = __builtin_amdgcn_update_dpp(0u, b[1], 0xB1, 0xF, 0xF, false);
                    const v4u o = odd ? v4u{nb0, nb1, b[0], b[1]} : v4u{a[0], a[1], na0, na1};
                    const int first = odd ? (i + 1) * TPR + t - 1 : i * TPR + t;        // first vector of this lane's pair (even: nvec is even, so the pair is in or out as a whole)
                    if (first < nvec) store_wt_b128(qr + (int64_t)first * 8, o);
                }
            } else {
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int idx = i * TPR + t;
                float f[EPV];
                Unpack<DT, EPV>::run(v[rr][i], f);
                uint32_t pk[EPV / 4];
                fast_encode<EPV, kQuotientSteps<DT>>(f, s, r, pk);
                if (idx < nvec) store_vec(idx, pk);
            }
            }
        } else {                                  // uniform per row group: true division (NaN/Inf data, extreme scales)
#pragma unroll 1
            for (int i = 0; i < VPT; ++i) {
                const int idx = i * TPR + t;
                if (idx < nvec) {
                    float f[EPV];
                    Unpack<DT, EPV>::run(*reinterpret_cast<const v4u*>(xr + (int64_t)idx * 16), f);
                    uint32_t pk[EPV / 4];
#pragma unroll
                    for (int g = 0; g < EPV / 4; ++g)
                        pk[g] = pack4(code_of(f[4 * g], s), code_of(f[4 * g + 1], s), code_of(f[4 * g + 2], s), code_of(f[4 * g + 3], s));
                    store_vec(idx, pk);
                }
            }
        }
    }
}

// dev/test kernel: compares the fast quotient + magic rounding against rintf(x / s) on raw bit patterns.
// out[0] += number of (x, s) pairs, among the n tested per thread, whose codes differ.
__global__ void fast_quotient_check(const uint32_t* __restrict__ xbits, const uint32_t* __restrict__ sbits, int64_t n,
                                    unsigned long long* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = __builtin_bit_cast(float, sbits[i]);
    float x = __builtin_bit_cast(float, xbits[i]);
    if (!(s > 0.0f) || !scale_fast_ok(s) || x != x) return;
    const float lim = 127.0f * s;                      // QSPEC guarantees |x| <= amax = about 127*s
    x = x > lim ? lim : (x < -lim ? -lim : x);
    const float r = 1.0f / s;
    const uint32_t fast = __builtin_bit_cast(uint32_t, quotient_fast(x, s, r) + kMagic) & 0xFFu;
    const uint32_t want = (uint32_t)code_of(x, s) & 0xFFu;
    const bool qdiff = __builtin_bit_cast(uint32_t, quotient_fast(x, s, r)) != __builtin_bit_cast(uint32_t, x / s) &&
                       __builtin_fabsf(x / s) >= 0x1p-40f;
    if (fast != want) atomicAdd(&out[0], 1ull);
    if (qdiff) atomicAdd(&out[1], 1ull);
}
void launch_fast_quotient_check(const uint32_t* xb, const uint32_t* sb, int64_t n, unsigned long long* out, hipStream_t st) {
    fast_quotient_check<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(xb, sb, n, out);
}

// dev/test kernel: the WHOLE domain of the one-step encode of a 16-bit format (quant_device.h: quotient_fast1).  Block b owns the amax
// bit pattern b + 1; its threads walk every magnitude pattern <= amax with both signs.  out[0] += pairs on the fast path,
// out[1] += pairs whose code differs from code_of(x, s) (true division + rintf).
template <int DT>
__global__ __launch_bounds__(256) void half_encode_check(unsigned long long* __restrict__ out) {
    const uint32_t a = blockIdx.x + 1u;
    const float s = scale_of(Elem<DT>::to_f32((uint16_t)a));
    if (!scale_fast_ok(s)) return;
    const float r = 1.0f / s;
    unsigned long long n = 0, bad = 0;
    for (uint32_t xb = threadIdx.x; xb <= a; xb += 256u) {
#pragma unroll
        for (uint32_t sign = 0; sign < 2; ++sign) {
            const float x = Elem<DT>::to_f32((uint16_t)(xb | (sign << 15)));
            const uint32_t fast = __builtin_bit_cast(uint32_t, quotient_fast1(x, s, r) + kMagic) & 0xFFu;
            const uint32_t want = (uint32_t)code_of(x, s) & 0xFFu;
            ++n;
            bad += fast != want ? 1u : 0u;
        }
    }
    atomicAdd(&out[0], n);
    if (bad) atomicAdd(&out[1], bad);
}
void launch_half_encode_check(int dtype, unsigned long long* out, hipStream_t st) {
    if (dtype == PQ_BF16) half_encode_check<PQ_BF16><<<dim3(0x7F7Fu), dim3(256), 0, st>>>(out);        // every finite positive pattern
    else half_encode_check<PQ_FP16><<<dim3(0x7BFFu), dim3(256), 0, st>>>(out);
}

// K1 generic path: any cols / leading dimension / alignment.  One block per row, two passes over the
// row (the second is served by L2).
template <int DT>
__global__ __launch_bounds__(256) void quant_rowwise_generic(const void* __restrict__ x, int64_t rows, int64_t cols,
                                                             int64_t ldx, int8_t* __restrict__ q, int64_t ldq,
                                                             float* __restrict__ scale) {
    using S = typename Elem<DT>::store_t;
    const int64_t row = blockIdx.x;
    const S* xr = reinterpret_cast<const S*>(x) + row * ldx;
    float amax = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) amax = amax_step(amax, Elem<DT>::to_f32(xr[c]));
    amax = wave_max(amax);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) amax = amax_merge(amax, part[w]);
    const float s = scale_of(amax);
    if (threadIdx.x == 0) scale[row] = s;
    int8_t* qr = q + row * ldq;
    for (int64_t c = threadIdx.x; c < cols; c += 256) qr[c] = (int8_t)code_of(Elem<DT>::to_f32(xr[c]), s);
}

// ------------------------------------------------------------------------------------------------
// K2: reduction along the strided axis, two passes over x (the second one finds a weight-sized matrix in the Infinity Cache).
//   pass 1 (col_amax): per-column max |x| on the raw bit patterns (integer max: two columns per v_pk_max_u16 for 16-bit types; a NaN pattern sorts above Inf,
//     so it propagates), merged across the block's four waves in LDS and across blocks by atomicMax on the f32 bit pattern.  `scale` is the scratch: while it
//     holds an AMAX the word carries the sign bit (0x80000000 | bits — the unsigned order is unchanged; the memset writes 0x80000000 = "amax 0").
//   pass 2 (col_encode): every block decodes the word — sign set: s = scale_of(amax); sign clear: it already is the final scale, written by the row-block-0
//     workgroup of that column strip while this one was still reading (both give the same s: the race is benign by construction) — and encodes with the per-column
//     exact quotient (division-free when every column of the wave allows it); the blockIdx.x == 0 workgroups store the final scales.  No third launch.
// Loads are unrolled kColUnroll rows deep (independent, clamped addresses): bytes in flight are what an HBM-bound column walk is made of.
constexpr int kColUnroll = 8;      // encode: independent row loads per thread per iteration
constexpr int kAmaxUnroll = 8;     // amax: raw 16-byte vectors only; 16 waves x 8 KiB in flight per workgroup
constexpr uint32_t kAmaxTag = 0x80000000u;

__device__ __forceinline__ float col_scale_of_word(uint32_t u) {
    return (u & kAmaxTag) ? scale_of(__builtin_bit_cast(float, u & 0x7FFFFFFFu)) : __builtin_bit_cast(float, u);
}

// pass 1.  VEC: 16 waves per workgroup; a wave-instruction reads EIGHT rows of an 8-vector column strip (128 bytes: one line per row).  Narrow strips and big
// workgroups mean few row splits per column for the same bytes in flight, and every row split of a column is one more device-scope atomic on the same word — they
// serialise at the memory side (with 64-lane strips and 64 row splits the atomics alone took ~10 us of a 28-us K2 at 4096 x 4096).  grid = (row splits, 8-vector strips).
constexpr int kAmaxWaves = 16;
template <int DT, bool VEC>
__global__ __launch_bounds__(64 * kAmaxWaves) void col_amax(const uint8_t* __restrict__ x, int64_t rows, int64_t ncolv,
                                                            int64_t ldx_bytes, uint32_t* __restrict__ amax_bits, int rows_per_block) {
    constexpr int EPV = VEC ? 16 / Elem<DT>::kBytes : 1;
    using S = typename Elem<DT>::store_t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // VEC: strips ride on grid.x (2^31 workgroups: any width), the few row splits on grid.y; scalar fallback: rows on grid.x, 64-column strips on grid.y
    const int64_t r0 = (int64_t)(VEC ? blockIdx.y : blockIdx.x) * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    if constexpr (VEC) {
        const int sub = lane & 7, rsub = lane >> 3;
        const int64_t cvr = (int64_t)blockIdx.x * 8 + sub;
        const int64_t cv = cvr < ncolv ? cvr : ncolv - 1;        // clamped: duplicates do not change a max
        const uint8_t* col = x + cv * 16;
        v4u m = {0u, 0u, 0u, 0u};                                 // per 32-bit word: one f32 magnitude, or two 16-bit magnitudes
        for (int64_t r = r0 + w * 8 + rsub; r < r1; r += 8 * kAmaxWaves * kAmaxUnroll) {
            v4u raw[kAmaxUnroll];
#pragma unroll
            for (int u = 0; u < kAmaxUnroll; ++u) {
                const int64_t rr = r + 8 * kAmaxWaves * u < r1 ? r + 8 * kAmaxWaves * u : r1 - 1;
                raw[u] = *reinterpret_cast<const v4u*>(col + rr * ldx_bytes);
            }
#pragma unroll
            for (int u = 0; u < kAmaxUnroll; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (DT == PQ_F32) { const uint32_t a = raw[u][i] & 0x7FFFFFFFu; m[i] = a > m[i] ? a : m[i]; }
                    else m[i] = pk_max_u16(m[i], raw[u][i] & 0x7FFF7FFFu);
                }
        }
        auto merge = [&](const v4u& o) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (DT == PQ_F32) m[i] = o[i] > m[i] ? o[i] : m[i];
                else m[i] = pk_max_u16(m[i], o[i]);
            }
        };
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {                 // the eight row groups of the wave
            v4u o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (uint32_t)__shfl_xor((int)m[i], off, 64);
            merge(o);
        }
        __shared__ v4u part[kAmaxWaves][8];
        if (rsub == 0) part[w][sub] = m;
        __syncthreads();
        if (w == 0 && rsub == 0 && cvr < ncolv) {
#pragma unroll
            for (int ww = 1; ww < kAmaxWaves; ++ww) merge(part[ww][sub]);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                uint32_t fb;                                     // the column's max |x| as an f32 bit pattern (a NaN stays a NaN: it sorts above Inf)
                if constexpr (DT == PQ_F32) fb = m[j];
                else fb = __builtin_bit_cast(uint32_t, amax_bits_to_f32<DT>((m[j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
                atomicMax(&amax_bits[cv * EPV + j], kAmaxTag | fb);
            }
        }
    } else {
        const int64_t cvr = (int64_t)blockIdx.y * 64 + lane;
        const int64_t cv = cvr < ncolv ? cvr : ncolv - 1;
        const uint8_t* col = x + cv * Elem<DT>::kBytes;
        float m = 0.0f;
        for (int64_t r = r0 + w; r < r1; r += 4 * kColUnroll) {
            float f[kColUnroll];
#pragma unroll
            for (int u = 0; u < kColUnroll; ++u) {
                const int64_t rr = r + 4 * u < r1 ? r + 4 * u : r1 - 1;
                f[u] = Elem<DT>::to_f32(*reinterpret_cast<const S*>(col + rr * ldx_bytes));
            }
#pragma unroll
            for (int u = 0; u < kColUnroll; ++u) m = amax_step(m, f[u]);
        }
        __shared__ float part[4][64];
        part[w][lane] = m;
        __syncthreads();
        if (w == 0 && cvr < ncolv) {
            float a = m;
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) a = amax_merge(a, part[ww][lane]);
            atomicMax(&amax_bits[cv], kAmaxTag | __builtin_bit_cast(uint32_t, a));
        }
    }
}

template <int DT, bool VEC>
__global__ __launch_bounds__(256) void col_encode(const uint8_t* __restrict__ x, int64_t rows, int64_t ncolv,
                                                  int64_t ldx_bytes, uint32_t* scale_io,
                                                  int8_t* __restrict__ q, int64_t ldq, int rows_per_block) {
    constexpr int EPV = VEC ? 16 / Elem<DT>::kBytes : 1;
    using S = typename Elem<DT>::store_t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t cvr = (int64_t)blockIdx.y * 64 + lane;
    const bool live = cvr < ncolv;
    const int64_t cv = live ? cvr : ncolv - 1;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const uint8_t* col = x + cv * (VEC ? 16 : Elem<DT>::kBytes);
    // the first batch of rows is requested BEFORE the scale words are read (two dependent trips to memory would otherwise open every workgroup, and a
    // workgroup of a weight-sized matrix lives for only two or three batches); from then on the next batch is in flight while the current one is encoded
    using Raw = typename std::conditional<VEC, v4u, S>::type;
    auto load_batch = [&](int64_t r, Raw (&raw)[kColUnroll]) {
#pragma unroll
        for (int u = 0; u < kColUnroll; ++u) {
            const int64_t rr = r + 4 * u < r1 ? r + 4 * u : r1 - 1;
            raw[u] = *reinterpret_cast<const Raw*>(col + rr * ldx_bytes);
        }
    };
    Raw bufA[kColUnroll], bufB[kColUnroll];
    int64_t r = r0 + w;
    if (r < r1) load_batch(r, bufA);

    float s[EPV], rcp[EPV];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        // (amax word or, already, the final scale: same s either way.  Relaxed agent-scope atomics: the word may be replaced by another workgroup while this one reads)
        s[j] = col_scale_of_word(__hip_atomic_load(&scale_io[cv * EPV + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        rcp[j] = 1.0f / s[j];
        ok = ok && scale_fast_ok(s[j]);
    }
    const bool fast = __builtin_amdgcn_ballot_w64(!ok) == 0;      // wave-uniform
    __syncthreads();                                              // every wave of THIS block has read its words before wave 0 replaces them
    if (blockIdx.x == 0 && w == 0 && live) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) __hip_atomic_store(&scale_io[cv * EPV + j], __builtin_bit_cast(uint32_t, s[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    auto encode_batch = [&](int64_t rb, const Raw (&raw)[kColUnroll]) {
#pragma unroll
        for (int u = 0; u < kColUnroll; ++u) {
            const int64_t rr = rb + 4 * u;
            float f[EPV];
            if constexpr (VEC) Unpack<DT, EPV>::run(raw[u], f);
            else f[0] = Elem<DT>::to_f32(raw[u]);
            if (live && rr < r1) {
                int8_t* o = q + rr * ldq + cv * EPV;
                if (fast) {
                    // every column of the wave passes scale_fast_ok: no NaN / Inf anywhere in these columns (they would be in the amax), |x| <= amax: the division-free
                    // exact code of K1 (quant_device.h: one correction step for 16-bit inputs, two for fp32), packed four codes per v_perm pair
                    uint32_t mb[EPV];
#pragma unroll
                    for (int j = 0; j < EPV; ++j)
                        mb[j] = __builtin_bit_cast(uint32_t, (kQuotientSteps<DT> == 1 ? quotient_fast1(f[j], s[j], rcp[j]) : quotient_fast(f[j], s[j], rcp[j])) + kMagic);
                    if constexpr (EPV >= 4) {
                        uint32_t pk[EPV / 4];
#pragma unroll
                        for (int g = 0; g < EPV / 4; ++g)
                            pk[g] = __builtin_amdgcn_perm(mb[4 * g + 1], mb[4 * g], 0x0c0c0400u) | __builtin_amdgcn_perm(mb[4 * g + 3], mb[4 * g + 2], 0x04000c0cu);
                        if constexpr (EPV == 8) store_wt_b64(o, v2u{pk[0], pk[1]});
                        else store_wt_b32(o, pk[0]);
                    } else {
                        *o = (int8_t)(mb[0] & 0xFFu);
                    }
                } else {
                    int c[EPV];
#pragma unroll
                    for (int j = 0; j < EPV; ++j) c[j] = code_of(f[j], s[j]);
                    if constexpr (EPV == 8) store_wt_b64(o, v2u{pack4(c[0], c[1], c[2], c[3]), pack4(c[4], c[5], c[6], c[7])});
                    else if constexpr (EPV == 4) store_wt_b32(o, pack4(c[0], c[1], c[2], c[3]));
                    else *o = (int8_t)c[0];
                }
            }
        }
    };
    constexpr int64_t STEP = 4 * kColUnroll;
    while (r < r1) {
        if (r + STEP < r1) load_batch(r + STEP, bufB);
        encode_batch(r, bufA);
        r += STEP;
        if (r >= r1) break;
        if (r + STEP < r1) load_batch(r + STEP, bufA);
        encode_batch(r, bufB);
        r += STEP;
    }
}

// (rows == 0 only: no encode launch — the words, all "amax 0", become scale 1.0)
__global__ void col_finalize(uint32_t* __restrict__ scale, int64_t cols) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) scale[c] = __builtin_bit_cast(uint32_t, col_scale_of_word(scale[c]));
}

// ------------------------------------------------------------------------------------------------
// dequant: out = cast_rne(f32(q) * scale[kept axis]).  Vector path: every lane STORES 16 contiguous bytes
// (8 half / 4 f32 results), so a wave-instruction writes 1 KiB of whole lines and reads 512/256 B contiguous.
template <int ODT, bool VEC>
__global__ __launch_bounds__(256) void dequant_kernel(const int8_t* __restrict__ q, int64_t ldq,
                                                      const float* __restrict__ scale, int axis, int64_t rows,
                                                      int64_t ncolv, void* __restrict__ out, int64_t ldo) {
    using S = typename Elem<ODT>::store_t;
    constexpr int EPT = VEC ? 16 / (int)sizeof(S) : 1;      // 8 (half) or 4 (f32) codes per thread
    const int64_t cvr = (int64_t)blockIdx.y * 64 + (threadIdx.x & 63);   // rows ride on grid.x (2^31 blocks)
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cvr >= ncolv || r >= rows) return;
    const int8_t* p = q + r * ldq + cvr * EPT;
    S* o = reinterpret_cast<S*>(out) + r * ldo + cvr * EPT;
    if constexpr (VEC) {
        uint32_t raw[EPT / 4];
        if constexpr (EPT == 8) { const v2u t = *reinterpret_cast<const v2u*>(p); raw[0] = t[0]; raw[1] = t[1]; }
        else raw[0] = *reinterpret_cast<const uint32_t*>(p);
        float scv[EPT];
        if (axis == 0) {                       // per-column scales: EPT consecutive floats, 16-byte vector loads
#pragma unroll
            for (int g = 0; g < EPT / 4; ++g) {
                const v4f t4 = *reinterpret_cast<const v4f*>(scale + cvr * EPT + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) scv[4 * g + j] = t4[j];
            }
        } else {
            const float sr = scale[r];
#pragma unroll
            for (int j = 0; j < EPT; ++j) scv[j] = sr;
        }
        S res[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int code = (int)(int8_t)((raw[j >> 2] >> (8 * (j & 3))) & 0xFF);
            res[j] = Elem<ODT>::from_f32((float)code * scv[j]);
        }
        store_wt_b128(o, *reinterpret_cast<const v4u*>(res));
    } else {
        const float sc = axis == 0 ? scale[cvr] : scale[r];
        *o = Elem<ODT>::from_f32((float)(*p) * sc);
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from pq_api.hip)
static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

template <int DT, int TPR>
static void launch_rowwise_vec(int vpt, const void* x, int64_t rows, int nvec, int64_t ldx_bytes, int8_t* q,
                               int64_t ldq, float* scale, hipStream_t st) {
    constexpr int RPB = 256 / TPR;
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
    const size_t k1_lds = (size_t)opt().k1_lds;
    if constexpr (TPR == kWave) {
        // two rows per wave: built to overlap one row's stores with the other's loads, and measured SLOWER on every shape
        // (profiles/r02_k1_rows_per_wave.txt: 4096 x 4096 9.56 -> 10.49 us, 16384 x 4096 31.6 -> 32.5 us): the one-row kernel
        // already runs at 76-80 % of 8 TB/s once the problem is large enough (4096 x 8192, 16384 x 4096); what keeps
        // 4096 x 4096 at 66 % is ~1.5 us of launch ramp and tail on an 8 us transfer, not the phase structure.  Kept
        // selectable (pq_set_option("PQ_K1_RPW", "2")) so the measurement can be repeated.
        const Options& o = opt();
        const bool two = o.k1_rpw == 2;
        if (two && vpt <= 8) {
            const dim3 grid2((unsigned)((rows + 2 * RPB - 1) / (2 * RPB))), block(256);
            switch (vpt) {
                case 1: quant_rowwise_vec<DT, 1, TPR, 2><<<grid2, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
                case 2: quant_rowwise_vec<DT, 2, TPR, 2><<<grid2, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
                case 4: quant_rowwise_vec<DT, 4, TPR, 2><<<grid2, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
                default: quant_rowwise_vec<DT, 8, TPR, 2><<<grid2, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
            }
        }
    }
    const dim3 grid((unsigned)((rows + RPB - 1) / RPB)), block(256);
    if constexpr (TPR == kWave && Elem<DT>::kBytes == 2) {
        if (opt().k1_st16 && (nvec & 1) == 0 && (ldq & 15) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0 && vpt >= 2 && vpt <= 8) {
            switch (vpt) {
                case 2: quant_rowwise_vec<DT, 2, TPR, 1, true><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
                case 4: quant_rowwise_vec<DT, 4, TPR, 1, true><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
                default: quant_rowwise_vec<DT, 8, TPR, 1, true><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); return;
            }
        }
    }
    switch (vpt) {
        case 1: quant_rowwise_vec<DT, 1, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 2: quant_rowwise_vec<DT, 2, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 4: quant_rowwise_vec<DT, 4, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 8: quant_rowwise_vec<DT, 8, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 16: quant_rowwise_vec<DT, 16, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        default:      // 32 vectors per lane (256 lanes per row only): rows up to 65 536 bf16 / 32 768 f32 columns stay in registers (128 VGPRs of raw data)
            if constexpr (TPR == 256) quant_rowwise_vec<DT, 32, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale);
            else quant_rowwise_vec<DT, 16, TPR><<<grid, block, k1_lds, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale);
            break;
    }
}

template <int DT>
void quant_rowwise_dispatch(const void* x, int64_t rows, int64_t cols, int64_t ldx, int8_t* q, int64_t ldq,
                            float* scale, hipStream_t st) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    // (cols == 0 goes to the generic kernel: it touches no element and writes scale = 1)
    const bool vec_ok = cols > 0 && (cols % EPV == 0) && (ldx % EPV == 0) && aligned(x, 16) && (ldq % EPV == 0) &&
                        aligned(q, EPV) && cols / EPV <= 256 * 32;
    if (vec_ok) {
        const int nvec = (int)(cols / EPV);
        auto pow2 = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
        if (nvec <= 64 * 8) {
            launch_rowwise_vec<DT, 64>(pow2((nvec + 63) / 64), x, rows, nvec, ldx * Elem<DT>::kBytes, q, ldq, scale, st);
        } else {
            launch_rowwise_vec<DT, 256>(pow2((nvec + 255) / 256), x, rows, nvec, ldx * Elem<DT>::kBytes, q, ldq, scale, st);
        }
    } else {
        quant_rowwise_generic<DT><<<dim3((unsigned)rows), dim3(256), 0, st>>>(x, rows, cols, ldx, q, ldq, scale);
    }
}

__global__ void fill_words(uint32_t* p, uint32_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
template <int DT>
hipError_t quant_colwise_dispatch(const void* x, int64_t rows, int64_t cols, int64_t ldx, int8_t* q, int64_t ldq,
                                  float* scale, hipStream_t st) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const bool vec_ok = (cols % EPV == 0) && (ldx % EPV == 0) && aligned(x, 16) && (ldq % EPV == 0) && aligned(q, EPV);
    const int64_t ncolv = vec_ok ? cols / EPV : cols;
    // rows per block: enough blocks to keep ~48 KiB per CU in flight (6 TB/s x ~2 us of latency over 256 CUs) without drowning the amax pass in per-block LDS
    // merges and atomics: ~2 blocks per CU for the amax pass (16 KiB of raw vectors in flight per block), ~4 per CU for the encode pass (8 rows deep, no atomics);
    // PQ_K2_BLOCKS_A / PQ_K2_BLOCKS_E (experiments) set the targets.  Multiples of one unrolled batch.
    const int64_t strips = (ncolv + 63) / 64;                                        // encode: 64-vector strips (and the scalar fallback of both passes)
    const int64_t strips_a = vec_ok ? (ncolv + 7) / 8 : strips;                      // amax: 8-vector strips, eight rows per wave-instruction, 16 waves per workgroup
    auto plan = [&](int64_t target_blocks, int64_t nstrips, int batch) {
        int64_t r = (rows * nstrips + target_blocks - 1) / target_blocks;
        r = (r + batch - 1) / batch * batch;
        return (int)(r < batch ? batch : (r > (1 << 20) ? (1 << 20) : r));
    };
    const int rpb_a = plan(opt().k2_blocks_a > 0 ? opt().k2_blocks_a : 256, strips_a, vec_ok ? 8 * kAmaxWaves * kAmaxUnroll : 4 * kColUnroll);
    const int rpb_e = plan(opt().k2_blocks_e > 0 ? opt().k2_blocks_e : 512, strips, 4 * kColUnroll);
    const dim3 grid_a = vec_ok ? dim3((unsigned)strips_a, (unsigned)((rows + rpb_a - 1) / rpb_a)) : dim3((unsigned)((rows + rpb_a - 1) / rpb_a), (unsigned)strips_a);
    const dim3 grid_e((unsigned)((rows + rpb_e - 1) / rpb_e), (unsigned)strips), block(256);
    const dim3 block_a(vec_ok ? 64 * kAmaxWaves : 256);
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
    const int64_t ldb = ldx * Elem<DT>::kBytes;
    uint32_t* words = reinterpret_cast<uint32_t*>(scale);
    // "amax 0" (the atomicMax identity) by a kernel of our own, not a memset node (round 6: gemm_s8_fast.hip, fsk_zero_counters — memset nodes in a hipGraph misbehaved)
    fill_words<<<dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st>>>(words, kAmaxTag, cols);
    const hipError_t me = hipGetLastError();
    if (me != hipSuccess) return me;                                                      // never launch on an un-initialised scratch
    if (rows > 0) {
        if (vec_ok) {
            col_amax<DT, true><<<grid_a, block_a, 0, st>>>(xb, rows, ncolv, ldb, words, rpb_a);
            col_encode<DT, true><<<grid_e, block, 0, st>>>(xb, rows, ncolv, ldb, words, q, ldq, rpb_e);
        } else {
            col_amax<DT, false><<<grid_a, block_a, 0, st>>>(xb, rows, ncolv, ldb, words, rpb_a);
            col_encode<DT, false><<<grid_e, block, 0, st>>>(xb, rows, ncolv, ldb, words, q, ldq, rpb_e);
        }
    } else if (cols > 0) {
        col_finalize<<<dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st>>>(words, cols);
    }
    return hipSuccess;
}

template <int ODT>
void dequant_dispatch(const int8_t* q, int64_t ldq, const float* scale, int axis, int64_t rows, int64_t cols,
                      void* out, int64_t ldo, hipStream_t st) {
    constexpr int EPT = 16 / Elem<ODT>::kBytes;
    const bool vec_ok = (cols % EPT == 0) && (ldq % EPT == 0) && aligned(q, EPT) && aligned(out, 16) &&
                        ((ldo * Elem<ODT>::kBytes) % 16 == 0) && (axis != 0 || aligned(scale, 16));
    const int64_t ncolv = vec_ok ? cols / EPT : cols;
    const dim3 grid((unsigned)((rows + 3) / 4), (unsigned)((ncolv + 63) / 64)), block(256);
    if (vec_ok) dequant_kernel<ODT, true><<<grid, block, 0, st>>>(q, ldq, scale, axis, rows, ncolv, out, ldo);
    else dequant_kernel<ODT, false><<<grid, block, 0, st>>>(q, ldq, scale, axis, rows, ncolv, out, ldo);
}

// explicit instantiations used by pq_api.hip
template void quant_rowwise_dispatch<PQ_BF16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_rowwise_dispatch<PQ_FP16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_rowwise_dispatch<PQ_F32>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template hipError_t quant_colwise_dispatch<PQ_BF16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template hipError_t quant_colwise_dispatch<PQ_FP16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template hipError_t quant_colwise_dispatch<PQ_F32>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void dequant_dispatch<PQ_BF16>(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);
template void dequant_dispatch<PQ_FP16>(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);
template void dequant_dispatch<PQ_F32>(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);

}  // namespace pq
