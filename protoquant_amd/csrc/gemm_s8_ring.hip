// gemm_s8_ring.hip — K3/K4 for SMALL GRIDS: 33 .. 512 tokens (short prefill chunks, speculative / batched decode) and any other problem whose 128 x 128 ring tiles
// (gemm_s8_ring128) would fill well under the 256 CUs: y[M, N] = dequant(xq[M, K] . wq[N, K]^T) on 64(m) x 128(n) or 64 x 64 tiles, one per CU on every CU.
//
// What bounds the regime (profiles/r04_ablate_ring.txt, r04_pmc_ring_tiles.txt): neither the matrix pipe nor HBM but the L2 -> CU path.  A tile of tm x tn
// rows ingests (tm + tn) x 128 bytes per K-tile by LDS-DMA for tm x tn x 128 multiply-adds — 16 KiB per 128 MFMA-cycles for a 64 x 64 tile, four times the bytes per operation
// of the 256 x 256 tile — and a CU ingests ~50 B/clk (26 TB/s over the chip: profiles/r01_ubench_l2_ingest.txt): ~330 cycles per K-tile for 64 x 64 (measured ~400), ~490 for
// 64 x 128 (~680); the 128 x 128 ring tile runs its DMA stream alone at 21 TB/s and the whole loop at a power-limited 1.56 GHz.  M = 512 x N = 4096 makes only 128 tiles of
// 128 x 128: half the chip at that rate.
// Splitting K over workgroups (the fused hand-over of the big tile) costs ~3 - 4 us of store -> ticket -> load latency, as much as it saves on a 10-us launch.  So this kernel
// goes the other way: smaller tiles on EVERY CU, with the loader / consumer structure of the ring tile, rings of 3 slots x 2 K-tiles x 24 KiB / 4 x 2 x 16 KiB (one barrier per
// slot) and a K walk rotated between the workgroups that share a weight panel (below).  The dispatcher (pq_api.hip: pick_variant) chooses between the three ring tiles by
// rounds of 256 CUs x the measured time of one tile.
//   * 8 waves: waves 0-3 consume (2 x 2 over the tile: wave tile (TN/2) n x (TM/2) m, fragments of the next K-tile read in the shadows of the
//     current one's MFMAs, double-buffered in registers), waves 4-7 issue the LDS-DMA pieces (8 rows x 128 B each) of tile kt + NB and do the
//     counted vmcnt waits; ONE s_barrier per ring slot (two K-tiles) joins the roles.
//   * LDS image as everywhere: [rows][128 B], 16-byte chunk c of row r at c ^ ((r >> 1) & 7) — swizzle on the DMA's per-lane SOURCE address and
//     on the ds_read_b128 address.
//   * epilogue: gemm_epilogue.h (QSPEC E1-E4 in registers, wave-private transpose through a free ring slot, write-through 16-byte stores);
//     ragged edges through guarded direct stores.  Same MFMA, same integer sums, same epilogue arithmetic as every other variant: bit-identical.
#include "gemm_tile_common.h"

namespace pq {

// NB ring slots of KT consecutive K-tiles each: ONE s_barrier per slot (a barrier round trip costs ~150 cycles on this chip — more than the 128 cycles of MFMA a
// 64 x 64 tile has per K-tile)
template <int OUT, int TM, int TN, int NB, int KT>
__global__ __launch_bounds__(512, 2) void gemm_s8_ringt(const int8_t* __restrict__ X, int64_t ldx, const int8_t* __restrict__ W, int64_t ldw,
                                                        EpiArgs epi, int M, int N, int K, int tiles_m, int tiles_n, int ct, int rot_div, KSlabs xs = KSlabs{}) {
    constexpr int P_OPER = TN * FBK, Q_OPER = TM * FBK, BUF = P_OPER + Q_OPER;
    constexpr int PPW = TN / 32, QPW = TM / 32, PPT = PPW + QPW;      // 1-KiB DMA pieces per loader wave per K-tile: P side, Q side, both
    constexpr int NPI = TN / 32, NQJ = TM / 32;                       // 16 x 16 tiles of a consumer's wave block: along n, along m
    constexpr int NMF = 2 * NPI * NQJ, NRD = 2 * (NPI + NQJ);         // MFMAs and fragment reads per K-tile and consumer wave
    constexpr int RPS = (2 * NRD + NMF - 1) / NMF;                    // fragment reads per MFMA shadow: all of them behind the first half of the MFMAs
    constexpr int SLOT = KT * BUF;
    static_assert(TM % 32 == 0 && TN % 32 == 0 && NB >= 3 && NB * SLOT <= 160 * 1024 && (NB - 1) * KT * PPT <= 63, "ring tile shape");
    __shared__ __attribute__((aligned(16))) uint8_t smem[NB * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int w = wave & 3, wp = w >> 1, wq = w & 1;

    // tile assignment: XCD remap (blocks of one XCD get a contiguous run of tiles), then bands of up to 8 m-tiles with m fastest inside the band (as gemm_s8_ring128): the
    // tiles that stream one weight panel are neighbours on one XCD — the panel comes from HBM once per XCD, the others hit its L2 — and an XCD's 32 tiles touch 8 activation
    // + 4 weight panels instead of 32 + 1
    const int t = xcd_remap((int)blockIdx.x, tiles_m * tiles_n, epi.nxcd);
    constexpr int GM = 8;
    const int band = t / (GM * tiles_n);
    const int gm = (tiles_m - band * GM) < GM ? (tiles_m - band * GM) : GM;
    const int tin = t - band * GM * tiles_n;
    const int tm = band * GM + tin % gm, tn = tin / gm;
    const int m0 = tm * TM, n0 = tn * TN;

    uint32_t offP[PPW], offQ[QPW];
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) {
        const int piece = w * PPW + jj, r = piece * 8 + (lane >> 3);
        const int src_chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        const int nl = (n0 + r < N) ? r : (N - 1 - n0);       // clamp: rows past the edge re-read a valid row
        offP[jj] = (uint32_t)nl * (uint32_t)ldw + src_chunk * 16;
    }
#pragma unroll
    for (int jj = 0; jj < QPW; ++jj) {
        const int piece = w * QPW + jj, r = piece * 8 + (lane >> 3);
        const int src_chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        const int ml = (m0 + r < M) ? r : (M - 1 - m0);
        offQ[jj] = (uint32_t)ml * (uint32_t)ldx + src_chunk * 16;
    }
    const int NT = K / FBK;
    // K ROTATION in chunks: the up-to-8 workgroups of a band that stream one weight panel (neighbours on one XCD, started together) walk each chunk of `ct` K-tiles
    // from DIFFERENT starting points, len / rot_div apart (rot_div = min(tiles_m, 8), from the launcher), wrapping inside the chunk.  In lockstep they would all ask for the same weight bytes at
    // the same time: the panel's unique bytes in flight — what HBM bandwidth is made of — would be ONE workgroup's ring however many workgroups there are
    // (measured, weights from HBM: 512 x 4096 x 14336 at 1.15 TB/s).  Rotated, every workgroup's ring holds different bytes; each byte comes from HBM once
    // and the others find it in the XCD's L2 — as long as the chunk of all the panels an XCD works on fits that L2 (4 MiB): a rotation over the whole of a
    // long K turned every re-read into an Infinity-Cache read and made the launch up to 2x slower (512 x 4096 x 14336: 37 -> 73 us).  `ct` is sized by the
    // launcher for ~2 MiB per XCD (ct >= NT: one chunk; rot_div = 0: no rotation).  Integer sums do not depend on the order of the K-tiles: same bits.
    auto rot_of = [&](int len) { return rot_div > 0 ? (int)(((int64_t)((tin % gm) % rot_div) * len) / rot_div) : 0; };
    int cbase = 0, clen = ct < NT ? ct : NT;
    int cpos = rot_of(clen), cleft = clen;
    const int8_t* const gP0 = W + (int64_t)n0 * ldw;
    const int8_t* const gQ0 = X + (int64_t)m0 * ldx;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lptr_t)smem;
    auto stage1 = [&](uint32_t la) {                           // this loader wave's pieces of its next K-tile, then the K walk moves on
        const int ktu = __builtin_amdgcn_readfirstlane(cbase + cpos);      // (the rotation's division is vector code; the DMA's base operand must be provably wave-uniform)
        const int8_t* gP = gP0 + ktu * FBK;
        // X stacked in K-slabs (pq_qlinear_s8_kslabs, as in gemm_s8_ring128): K-tile kt lives in slab kt / xs.tiles; the division by a host-computed reciprocal, all scalar
        int64_t xo = (int64_t)ktu * FBK;
        if (xs.tiles == 1) {
            xo = (int64_t)ktu * xs.stride;                               // one K-tile per slab (its reciprocal, 2^32, does not fit the 32-bit magic)
        } else if (xs.tiles > 1) {
            const int sl = (int)(((uint64_t)(uint32_t)ktu * (uint64_t)xs.magic) >> 32);
            xo = (int64_t)sl * xs.stride + (int64_t)(ktu - sl * xs.tiles) * FBK;
        }
        const int8_t* gQ = gQ0 + xo;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) glds16_sbase(gP, offP[jj], la + (uint32_t)(w * PPW + jj) * 1024u);
#pragma unroll
        for (int jj = 0; jj < QPW; ++jj) glds16_sbase(gQ, offQ[jj], la + P_OPER + (uint32_t)(w * QPW + jj) * 1024u);
        if (++cpos == clen) cpos = 0;
        if (--cleft == 0) {                                    // next chunk
            cbase += clen;
            clen = NT - cbase < ct ? NT - cbase : ct;
            cleft = clen;
            cpos = clen > 0 ? rot_of(clen) : 0;
        }
    };
    const int NS = (NT + KT - 1) / KT;                         // ring slots' worth of K-tiles (the last one may be partly filled)
    auto stage = [&](int s) {                                  // slot s: its KT K-tiles (those that exist)
#pragma unroll
        for (int u = 0; u < KT; ++u)
            if (s * KT + u < NT) stage1(smem_base + (uint32_t)(s % NB) * SLOT + (uint32_t)u * BUF);
    };

    if (loader) {
        // prologue: up to NB slots in flight; then, per slot s: slot s + 1 must have landed (slots s + 2 .. s + NB - 1 may stay in flight: counted in the
        // pieces they really hold), the barrier the consumers share, and the pieces of slot s + NB into the ring slot that slot s has just vacated
        auto fly = [&](int s) {                                // this wave's pieces of the slots behind slot s + 1 that have been issued
            const int t0 = (s + 2) * KT, t1 = (s + NB) * KT < NT ? (s + NB) * KT : NT;
            return t1 > t0 ? (t1 - t0) * PPT : 0;
        };
#pragma unroll 1
        for (int b = 0; b < NB && b < NS; ++b) stage(b);
        {                                                      // slot 0 has landed; slots 1 .. NB - 1 may stay in flight
            const int t1 = NB * KT < NT ? NB * KT : NT;
            wait_vmcnt_lgkm0(t1 > KT ? (t1 - KT) * PPT : 0);
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int sl = 0; sl < NS; ++sl) {
            if (sl + 1 < NS) {
                wait_vmcnt_lgkm0(fly(sl));
                __builtin_amdgcn_s_barrier();
            }
            if (sl + NB < NS) stage(sl + NB);
        }
        return;
    }

    // ---- consumers.  Fragments: P tile i = rows wp * (TN/2) + 16 i .. + 15, Q tile j = rows wq * (TM/2) + 16 j .. + 15; k-step ks = 64 bytes
    const int frow = lane & 15, fchunk = lane >> 4, fkey = (frow >> 1) & 7;
    uint32_t lP[2], lQ[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fchunk;
        lP[ks] = (uint32_t)((wp * (TN / 2) + frow) * 128 + ((c ^ fkey) * 16));
        lQ[ks] = (uint32_t)((wq * (TM / 2) + frow) * 128 + ((c ^ fkey) * 16)) + P_OPER;
    }
    // item it < 2 NPI: P tile it % NPI of k-step it / NPI; the others: Q tile (it - 2 NPI) % NQJ of k-step (it - 2 NPI) / NQJ
    v4i fa[NRD], fb[NRD];
    auto read_item = [&](int bufoff, v4i (&f)[NRD], auto ic) {
        constexpr int it = decltype(ic)::value;
        if constexpr (it < 2 * NPI) f[it] = *reinterpret_cast<const v4i*>(smem + bufoff + lP[it / NPI] + (it % NPI) * 16 * 128);
        else f[it] = *reinterpret_cast<const v4i*>(smem + bufoff + lQ[(it - 2 * NPI) / NQJ] + ((it - 2 * NPI) % NQJ) * 16 * 128);
    };
    v4i acc[NPI][NQJ];
#pragma unroll
    for (int i = 0; i < NPI; ++i)
#pragma unroll
        for (int j = 0; j < NQJ; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    __builtin_amdgcn_s_barrier();                              // tile 0 has landed (the loaders waited for it)
    static_for<NRD>([&](auto ic) { read_item(0, fa, ic); });

    auto tile = [&](int kt, v4i (&cur)[NRD], v4i (&nxt)[NRD]) {
        // the next K-tile opens a new slot: that slot must have landed (the loaders waited), and this wave is done with the slot it leaves
        if (kt + 1 < NT && (kt + 1) % KT == 0) {
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));    // this wave's fragment reads so far (the loaders wait for the DMA)
            __builtin_amdgcn_s_barrier();
        }
        const int nbuf = (((kt + 1) / KT) % NB) * SLOT + ((kt + 1) % KT) * BUF;
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMF>([&](auto xc) {
            constexpr int x = decltype(xc)::value, ks = x / (NPI * NQJ), i = (x / NQJ) % NPI, j = x % NQJ;
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur[ks * NPI + i], cur[2 * NPI + ks * NQJ + j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            static_for<RPS>([&](auto rc) {                     // (last tile: reads a stale slot, values unused)
                constexpr int it = x * RPS + decltype(rc)::value;
                if constexpr (it < NRD) read_item(nbuf, nxt, std::integral_constant<int, it>{});
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int kt = 0;
    for (; kt + 1 < NT; kt += 2) { tile(kt, fa, fb); tile(kt + 1, fb, fa); }
    if (kt < NT) tile(kt, fa, fb);

    // ---- epilogue: D[row <-> n][col <-> m]; lane holds 4 consecutive n of one m per accumulator
    using O = typename OutElem<OUT>::type;
    constexpr int OB = (int)sizeof(O);
    O* y = reinterpret_cast<O*>(epi.y);
    const bool has_bias = (OUT != OUT_I32) && epi.bias != nullptr;
    const int dcol = lane & 15, drow4 = (lane >> 4) * 4;
    const int wm0 = m0 + wq * (TM / 2), wn0 = n0 + wp * (TN / 2);
    const bool staged = (wm0 + TM / 2 <= M) && (wn0 + TN / 2 <= N) && epi_rows_storable(epi, y, OB) &&
                        (OUT == OUT_I32 || (reinterpret_cast<uintptr_t>(epi.b_scale) & 15) == 0) &&
                        (!has_bias || (reinterpret_cast<uintptr_t>(epi.bias) & (4 * OB - 1)) == 0);
    if (staged) {
        // staging: this wave's quarter of the first K-tile of the ring slot BEHIND the last one: no slot NS exists, so no DMA targets it, its previous
        // tenant (slot NS - NB) was read out long ago, and the last tile's prefetch of "tile NT" reads values nobody uses
        const uint32_t sw_off = (uint32_t)((NS % NB) * SLOT + w * (BUF / 4));
        constexpr int PT_PASS = (NPI * 16 * OB > 256 && NPI % 2 == 0) ? NPI / 2 : NPI;       // (five column tiles — the 128 x 160 tile — stage in one pass of 160 / 320 used bytes per row)
        constexpr int RSTRIDE = PT_PASS * 16 * OB <= 64 ? 64 : PT_PASS * 16 * OB <= 128 ? 128 : PT_PASS * 16 * OB <= 256 ? 256 : 512;      // gemm_epilogue.h: RBY
        constexpr int QT_MAX = (BUF / 4) / (16 * RSTRIDE);
        constexpr int QT_PASS = QT_MAX >= NQJ ? NQJ : 1;
        static_assert(QT_PASS >= 1 && QT_PASS * 16 * RSTRIDE <= BUF / 4, "epilogue staging region");
        auto acc_of = [&](int pt, int qt) -> const v4i& { return acc[pt][qt]; };
        auto as_of = [&](int qt) { return epi.a_scale[wm0 + qt * 16 + dcol]; };
        auto bs_of = [&](int pt) { return *reinterpret_cast<const v4f*>(epi.b_scale + wn0 + pt * 16 + drow4); };
        uint8_t* y_blk = reinterpret_cast<uint8_t*>(y + (int64_t)wm0 * epi.ldy + wn0);
        const void* bias_blk = has_bias ? static_cast<const void*>(reinterpret_cast<const O*>(epi.bias) + ((epi.flags & EPI_BIAS_ROWS) ? wm0 : wn0)) : nullptr;
        PQ_EPI_STAGED_DISPATCH(OUT, NPI, NQJ, QT_PASS, PT_PASS, has_bias, epi.flags, acc_of, as_of, bs_of, bias_blk, smem, sw_off, y_blk, epi.ldy * OB, lane);
        return;
    }
    // direct path (edge tiles / unaligned y): guarded stores from registers, 4 consecutive n at a time when aligned
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) & (4 * OB - 1)) == 0) && ((epi.ldy & 3) == 0);
#pragma unroll
    for (int j = 0; j < NQJ; ++j) {
        const int m = wm0 + j * 16 + dcol;
        if (m >= M) continue;
        float as = 1.0f;
        if constexpr (OUT != OUT_I32) as = epi.a_scale[m];
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int nb = wn0 + i * 16 + drow4;
            if (nb >= N) continue;
            O o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + r < N ? nb + r : N - 1;
                float bs = 1.0f, bf = 0.0f;
                if constexpr (OUT != OUT_I32) {
                    bs = epi.b_scale[n];
                    if (has_bias) bf = load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : n);
                }
                o[r] = epi_convert<OUT>(acc[i][j][r], as, bs, bf, has_bias, epi.flags & EPI_COL_FIRST);
            }
            O* dst = y + (int64_t)m * epi.ldy + nb;
            if (nb + 3 < N && vec_ok) {
                if constexpr (OB == 2) *reinterpret_cast<v2u*>(dst) = *reinterpret_cast<const v2u*>(o);
                else *reinterpret_cast<v4u*>(dst) = *reinterpret_cast<const v4u*>(o);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < N) dst[r] = o[r];
            }
        }
    }
}

// PQ_MIDM_CT (experiments): 0 = rot_chunk_ktiles(), 1 = no rotation, n > 1 = n K-tiles per chunk.
static void rot_plan(int tiles_m, int tn, int* ct, int* rot_div) {
    const int force = opt().midm_ct;
    *ct = force > 1 ? force : rot_chunk_ktiles(tiles_m, tn);
    *rot_div = force == 1 ? 0 : tiles_m;
}

// tile: 0 = 64(m) x 128(n), 3 slots of 2 K-tiles (144 KiB); 1 = 64 x 64, 4 slots of 2 K-tiles (128 KiB); 2 (round 6) = 128(m) x 160(n), 4 slots of 1 K-tile (144 KiB): the
// tile that turns a 1280-wide output (the Llama-3-70B fused-qkv shard: 4096 x 1280) into exactly 32 x 8 = 256 workgroups — the 128 x 256 tile fills 160 of the 256 CUs
template <int OUT>
void launch_gemm_ringt(int tile, const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi, int64_t M, int64_t N, int64_t K, hipStream_t st,
                       int64_t a_slab_stride, int64_t a_k_per_slab) {
    int ct = 0, rd = 0;
    KSlabs xs{};
    if (a_k_per_slab > 0) {          // stacked activation operand (the caller checked: k_per_slab % 128 == 0, K / 128 < 2^16)
        xs.tiles = (int)(a_k_per_slab / FBK);
        xs.magic = (uint32_t)(((1ull << 32) + (uint64_t)xs.tiles - 1) / (uint64_t)xs.tiles);
        xs.stride = a_slab_stride;
    }
    if (tile == 2) {
        const int tiles_m = (int)((M + 127) / 128), tiles_n = (int)((N + 159) / 160);
        rot_plan(tiles_m < 8 ? tiles_m : 8, 160, &ct, &rd);
        if (N * K < (6 << 20)) rd = 0;          // (weight streams that matter, as for the 128 x 128 ring tile)
        gemm_s8_ringt<OUT, 128, 160, 4, 1><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, ct, rd, xs);
    } else if (tile == 0) {
        const int tiles_m = (int)((M + 63) / 64), tiles_n = (int)((N + 127) / 128);
        rot_plan(tiles_m < 8 ? tiles_m : 8, 128, &ct, &rd);
        gemm_s8_ringt<OUT, 64, 128, 3, 2><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, ct, rd, xs);
    } else {
        const int tiles_m = (int)((M + 63) / 64), tiles_n = (int)((N + 63) / 64);
        rot_plan(tiles_m < 8 ? tiles_m : 8, 64, &ct, &rd);
        gemm_s8_ringt<OUT, 64, 64, 4, 2><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, ct, rd, xs);
    }
}
template void launch_gemm_ringt<PQ_BF16>(int, const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);
template void launch_gemm_ringt<PQ_FP16>(int, const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);
template void launch_gemm_ringt<PQ_F32>(int, const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);
template void launch_gemm_ringt<OUT_I32>(int, const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);

}  // namespace pq
