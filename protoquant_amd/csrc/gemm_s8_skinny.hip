// gemm_s8_skinny.hip — K3/K4 for decode-like problems (M <= 64 tokens): y[M,N] = dequant(xq[M,K] . wq[N,K]^T).
// With so few tokens the GEMM is a streaming read of the int8 weight matrix — HBM-bound, N*K bytes — and the tiled kernels
// (one 128-row tile per CU, half of every staged K-tile spent on duplicated activation rows, N/128 CUs busy) reach < 1 TB/s.
// Here every wave streams 16 weight rows STRAIGHT INTO MFMA FRAGMENTS: lane (r = lane & 15, c = lane >> 4) loads the 16 bytes
// k0 + 16c .. +15 of row n0 + r — exactly the A operand of v_mfma_i32_16x16x64_i8 — so the weights never touch LDS; the
// few activation rows come the same way from L2.  A workgroup = KS waves that split K between them for one 16-row block
// (enough waves in flight to cover HBM latency: KS * U KiB per workgroup), reduce their exact int32 partial tiles through
// LDS and apply QSPEC E1-E4.  Results are bit-identical to every other variant.
#include <cstdlib>

#include "gemm_epilogue.h"

namespace pq {

// k-steps (64 bytes of K each) per batch, sized so that 16 waves per workgroup stay within 128 VGPRs.  One weight block per wave (RB = 1): TWO batches of weights are in
// registers — the next one is requested before the current one is multiplied, so a wave always has weight bytes in flight (HBM-fed: 1 x 4096 x 4096 7.1 -> 6.7 us,
// 16 x 4096 x 14336 20.5 -> 18.7).  Two blocks per wave (RB = 2): one batch at a time, as in rounds 1-3 — the pipelined form was 7 % SLOWER there (1 x 28672 x 4096).
constexpr int sk_batch(int mt, int rb) { return rb == 1 ? (mt == 1 ? 4 : 2) : (mt * rb == 2 ? 4 : 2); }

// RB: 16-row weight blocks per wave.  The direct-to-fragment access pattern (16 segments of 64 B per instruction) tops out
// near 9.8 TB/s over the whole chip (tools/ubench/l2_ingest), and the activation fragments travel the same path from L2:
// with RB = 1 and one token tile they take half of it.  RB = 2 reuses every activation fragment for two weight blocks.
// STAGE (round 4): the activation fragments no longer come straight from L2 in the MFMA operand layout (16 rows x 64 B per load instruction — the same 16-segment request the
// weight fragments need, on the same vector-memory pipeline: with one weight block per wave HALF of that pipeline's requests were activation re-reads, and 16 tokens ran 15 - 40 %
// slower than 1 token, whose 16 "rows" are one clamped row).  A wave now loads its K-batch of the token rows with row-contiguous requests (2 - 4 rows x 256 - 512 B per
// instruction), parks them in a wave-private LDS region (row stride padded by 16 B: conflict-free both ways) and reads the fragments back with ds_read_b128.  Same values, same
// MFMAs: bit-identical.
template <int OUT, int MT, int RB, bool STAGE>       // MT: 16-row tiles of activations (M <= 16 * MT)
__global__ __launch_bounds__(1024) void gemm_s8_skinny(const int8_t* __restrict__ X, int64_t ldx, const int8_t* __restrict__ W,
                                                       int64_t ldw, EpiArgs epi, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sk_smem[];     // STAGE: [KS][MT * 16 rows][SK_U * 64 + 16 B] staging, reused as [KS][MT][64 lanes] v4i for the reduction
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), KS = blockDim.x >> 6;
    const int r = lane & 15, c = lane >> 4;
    const int n0 = blockIdx.x * (16 * RB);
    constexpr int SK_U = sk_batch(MT, RB);

    // this wave's k-steps: a balanced slice of the K / 64 steps
    const int steps = K >> 6, s0 = (int)((int64_t)steps * w / KS), s1 = (int)((int64_t)steps * (w + 1) / KS);
    const int8_t* wp[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int nrow = n0 + b * 16 + r < N ? n0 + b * 16 + r : N - 1;    // clamp: rows past the edge re-read a valid row
        wp[b] = W + (int64_t)nrow * ldw + c * 16;
    }
    const int8_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + r < M ? t * 16 + r : M - 1;
        xp[t] = X + (int64_t)m * ldx + c * 16;
    }
    v4i acc[RB][MT];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[b][t] = v4i{0, 0, 0, 0};

    // STAGE: row-contiguous activation loads.  A batch is SK_U k-steps = UPR = 4 SK_U 16-byte units per row; a load instruction covers 64 / UPR rows.
    constexpr int UPR = 4 * SK_U, RPI = 64 / UPR, NLD = MT * 16 / RPI;          // units per row, rows per instruction, instructions per batch
    constexpr int RSTR = SK_U * 64 + 16;                                        // staged row stride (bytes)
    uint8_t* const stg = sk_smem + (size_t)w * (MT * 16 * RSTR);
    const int lrow = lane / UPR, lunit = lane % UPR;
    const int8_t* xrow[STAGE ? NLD : 1];
    if constexpr (STAGE) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = i * RPI + lrow < M ? i * RPI + lrow : M - 1;          // (rows past M: a valid row again; their products land in accumulator columns nobody stores)
            xrow[i] = X + (int64_t)m * ldx + lunit * 16;
        }
    }
    // one batch: SK_U k-steps.  RB = 1: the weight fragments of batch s + SK_U are requested BEFORE the MFMAs of batch s (two register sets).
    auto load_w = [&](v4i (&fw)[SK_U][RB], int sb) {
#pragma unroll
        for (int u = 0; u < SK_U; ++u)
#pragma unroll
            for (int b = 0; b < RB; ++b) fw[u][b] = *reinterpret_cast<const v4i*>(wp[b] + (int64_t)(sb + u) * 64);
    };
    auto compute = [&](const v4i (&fw)[SK_U][RB], int sb) {
        v4i fx[SK_U][MT];
        if constexpr (!STAGE) {
#pragma unroll
            for (int u = 0; u < SK_U; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t) fx[u][t] = *reinterpret_cast<const v4i*>(xp[t] + (int64_t)(sb + u) * 64);
        } else {
            v4i xr[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) xr[i] = *reinterpret_cast<const v4i*>(xrow[i] + (int64_t)sb * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the previous batch's fragment reads have returned: its region may be overwritten
#pragma unroll
            for (int i = 0; i < NLD; ++i) *reinterpret_cast<v4i*>(stg + (i * RPI + lrow) * RSTR + lunit * 16) = xr[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // wave-private region, LDS operations of one wave execute in order: written -> readable
#pragma unroll
            for (int u = 0; u < SK_U; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t) fx[u][t] = *reinterpret_cast<const v4i*>(stg + (t * 16 + r) * RSTR + (u * 4 + c) * 16);
        }
#pragma unroll
        for (int u = 0; u < SK_U; ++u)
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[b][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[u][b], fx[u][t], acc[b][t], 0, 0, 0);
    };
    int s = s0;
    if constexpr (RB != 1) {            // (rounds 1-3's loop, weight and activation requests interleaved k-step by k-step when the activations come straight from L2)
        for (; s + SK_U <= s1; s += SK_U) {
            v4i fw[SK_U][RB];
            if constexpr (STAGE) {
                load_w(fw, s);
                compute(fw, s);
            } else {
                v4i fx[SK_U][MT];
#pragma unroll
                for (int u = 0; u < SK_U; ++u) {
#pragma unroll
                    for (int b = 0; b < RB; ++b) fw[u][b] = *reinterpret_cast<const v4i*>(wp[b] + (int64_t)(s + u) * 64);
#pragma unroll
                    for (int t = 0; t < MT; ++t) fx[u][t] = *reinterpret_cast<const v4i*>(xp[t] + (int64_t)(s + u) * 64);
                }
#pragma unroll
                for (int u = 0; u < SK_U; ++u)
#pragma unroll
                    for (int b = 0; b < RB; ++b)
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[b][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[u][b], fx[u][t], acc[b][t], 0, 0, 0);
            }
        }
    } else if (s + SK_U <= s1) {
        v4i fw0[SK_U][RB], fw1[SK_U][RB];
        load_w(fw0, s);
        for (;;) {
            const bool more1 = s + 2 * SK_U <= s1;
            if (more1) load_w(fw1, s + SK_U);
            compute(fw0, s); s += SK_U;
            if (!more1) break;
            const bool more2 = s + 2 * SK_U <= s1;
            if (more2) load_w(fw0, s + SK_U);
            compute(fw1, s); s += SK_U;
            if (!more2) break;
        }
    }
    for (; s < s1; ++s) {
        v4i fx[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) fx[t] = *reinterpret_cast<const v4i*>(xp[t] + (int64_t)s * 64);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const v4i fw = *reinterpret_cast<const v4i*>(wp[b] + (int64_t)s * 64);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[b][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw, fx[t], acc[b][t], 0, 0, 0);
        }
    }

    // ---- exact reduction of the KS partial tiles through LDS, then E1-E4.  D[row <-> n][col <-> m]: the lane holds
    // n = n0 + 4c .. 4c+3 of token m = 16t + r.
    v4i* red = reinterpret_cast<v4i*>(sk_smem);
    constexpr int NTL = RB * MT;                          // output tiles of this workgroup: tile q = b * MT + t
    if (KS > 1) {
        if constexpr (STAGE) __syncthreads();            // the reduction buffer overlays the staging regions of ALL waves: everybody is done with theirs
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int t = 0; t < MT; ++t) red[(w * NTL + b * MT + t) * 64 + lane] = acc[b][t];
        __syncthreads();
    }
    using O = typename OutElem<OUT>::type;
    O* y = reinterpret_cast<O*>(epi.y);
    const bool has_bias = (OUT != OUT_I32) && epi.bias != nullptr;
    for (int q = w; q < NTL; q += KS) {                   // wave w finishes tiles w, w + KS, ...
        const int b = q / MT, t = q - b * MT;
        v4i sum = acc[0][0];
        if (KS > 1) {
            sum = v4i{0, 0, 0, 0};
            for (int k = 0; k < KS; ++k) sum += red[(k * NTL + q) * 64 + lane];
        } else {
#pragma unroll
            for (int bb = 0; bb < RB; ++bb)
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) if (bb * MT + tt == q) sum = acc[bb][tt];
        }
        const int m = t * 16 + r, nb = n0 + b * 16 + c * 4;
        if (m >= M || nb >= N) continue;
        float as = 1.0f;
        if constexpr (OUT != OUT_I32) as = epi.a_scale[m];
        O o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb + j < N ? nb + j : N - 1;
            float bs = 1.0f, bf = 0.0f;
            if constexpr (OUT != OUT_I32) {
                bs = epi.b_scale[n];
                if (has_bias) bf = load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : n);
            }
            o[j] = epi_convert<OUT>(sum[j], as, bs, bf, has_bias, epi.flags & EPI_COL_FIRST);
        }
        if (epi.flags & EPI_STORE_T) {                    // y^T[n][m]: four rows of the transposed output, one element each (the whole output is M x N <= 64 x N elements)
#pragma unroll
            for (int j = 0; j < 4; ++j) if (nb + j < N) y[(int64_t)(nb + j) * epi.ldy + m] = o[j];
            continue;
        }
        O* dst = y + (int64_t)m * epi.ldy + nb;
        const bool vec = (nb + 3 < N) && ((reinterpret_cast<uintptr_t>(dst) & (4 * sizeof(O) - 1)) == 0);
        if (vec) {
            if constexpr (sizeof(O) == 2) *reinterpret_cast<v2u*>(dst) = *reinterpret_cast<const v2u*>(o);
            else *reinterpret_cast<v4u*>(dst) = *reinterpret_cast<const v4u*>(o);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (nb + j < N) dst[j] = o[j];
        }
    }
}

// (RB, KS) of a launch: weight blocks per wave and the K split (waves per workgroup).  Round 4 swept both over 19 decode-like shapes, weights from HBM
// (profiles/r04_skinny_sweep.txt); the rules below are that table's arg-min, within 3 % on every shape (the round-3 rule — two blocks per wave from N = 6144 on,
// ~4096 waves per launch — was 5 - 21 % off on eleven of them):
//   * one token: one block per wave, at least four waves per workgroup (the waves of a workgroup walk the SAME 16 weight rows at different K offsets: with one
//     wave per workgroup thousands of rows are open at once, 64 bytes at a time — lm_head 115 us against 93), ~4096 waves per launch;
//   * 2 .. 32 tokens (the activation fragments are real traffic): narrow matrices (N <= 4096) one block per wave and ~2048 waves; 4096 < N < 16384 two blocks per
//     wave (every activation fragment reused twice) and only ~768 waves; from N = 16384 on one block per wave again, split in 4 (up to 8 tokens) or 2;
//   * 33 .. 64 tokens (three or four token tiles; only against narrow matrices): one block per wave, ~4096 waves — unchanged.
// Always at least one k-step batch per wave, at most 16 waves per workgroup.  PQ_SKINNY_RB / PQ_SKINNY_KS force either.
static void skinny_plan(int64_t M, int64_t N, int64_t K, int* rb_out, int* ks_out) {
    const int mt = (int)((M + 15) / 16);
    int rb = 1, min_ks = 1;
    int64_t target = 4096;
    if (mt <= 2) {
        if (M == 1) min_ks = 4;
        else if (N <= 4096) target = 2048, min_ks = 2;
        else if (N < 16384) rb = 2, target = 768, min_ks = 2;     // (from N = 4097 on: 16-row blocks of a 5120-wide matrix are 320 workgroups, a round and a quarter of the chip)
        else target = 1, min_ks = M <= 8 ? 4 : 2;
    }
    if (const int f = opt().skinny_rb; f && mt <= 2) rb = f;
    const int64_t blocks = (N + 16 * rb - 1) / (16 * rb), steps = K / 64;
    int ks = 1;
    while (ks < 16 && (blocks * ks < target || ks < min_ks) && steps / (ks * 2) >= sk_batch(mt, rb)) ks <<= 1;
    if (const int f = opt().skinny_ks; f > 0) {            // PQ_SKINNY_KS (experiments): a forced power of two, clamped to what the K-steps allow
        ks = 1;
        while (ks < f && ks < 16 && steps / (ks * 2) >= 1) ks <<= 1;
    }
    *rb_out = rb; *ks_out = ks;
}

template <int OUT>
void launch_gemm_skinny(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi, int64_t M, int64_t N,
                        int64_t K, hipStream_t st) {
    const int mt = (int)((M + 15) / 16);
    int rb = 1, ks = 1;                                    // (3-4 token tiles x 2 blocks would not fit the register budget: skinny_plan keeps rb = 1 there)
    skinny_plan(M, N, K, &rb, &ks);
    const int64_t blocks = (N + 16 * rb - 1) / (16 * rb);
    const dim3 grid((unsigned)blocks), block((unsigned)(ks * 64));
    const bool stage = opt().skinny_stage && M > 1;       // (one token: its 16 "rows" are one clamped row — the direct loads are already cheap)
    const size_t lds_red = ks > 1 ? (size_t)ks * mt * rb * 64 * sizeof(v4i) : 0;
    const size_t lds_stg = stage ? (size_t)ks * mt * 16 * (sk_batch(mt, rb) * 64 + 16) : 0;
    const size_t lds = lds_red > lds_stg ? lds_red : lds_stg;
    // (the staging regions of 16 waves need up to 147 KiB of dynamic LDS: above the 64-KiB default the limit is raised — per launch, a host-side attribute of the
    // function on the CURRENT device, so a process that drives several devices gets it on each)
#define PQ_SK(MTv, RBv) do { if (stage) { if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_s8_skinny<OUT, MTv, RBv, true>), \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                  gemm_s8_skinny<OUT, MTv, RBv, true><<<grid, block, lds, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K); } \
                             else gemm_s8_skinny<OUT, MTv, RBv, false><<<grid, block, lds, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K); } while (0)
    if (rb == 2) { if (mt == 1) PQ_SK(1, 2); else PQ_SK(2, 2); }
    else {
        switch (mt) {
            case 1: PQ_SK(1, 1); break;
            case 2: PQ_SK(2, 1); break;
            case 3: PQ_SK(3, 1); break;
            default: PQ_SK(4, 1); break;
        }
    }
#undef PQ_SK
}
template void launch_gemm_skinny<PQ_BF16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template void launch_gemm_skinny<PQ_FP16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template void launch_gemm_skinny<PQ_F32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template void launch_gemm_skinny<OUT_I32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);

}  // namespace pq
