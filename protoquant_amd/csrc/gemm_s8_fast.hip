// gemm_s8_fast.hip — K3/K4 hot variant for gfx950: s8 x s8 -> s32 on v_mfma_i32_16x16x64_i8
// with the fused row-scale x col-scale dequant epilogue.
//
// Structure (designed for CDNA4):
//   * 256 x 256 output tile per workgroup, K step = 128 bytes, 512 threads = 8 waves = 2 per SIMD.
//   * MFMA roles: first operand P = weight rows (n), second operand Q = activation rows (m); a lane's
//     accumulator registers are then consecutive n of ONE output row m.
//   * wave (wp, wq) = (w>>2, w&3) owns n-range wp*128+[0,128) x m-range wq*64+[0,64), split in
//     halves hP (64 n) x hQ (32 m): four quadrants of 16 MFMAs.  (A 32x32x32 form of the same kernel held a
//     lower clock, 1.74 vs 2.07 GHz in a bare MFMA loop, and ran 13 % slower: removed in round 2.)
//   * LDS: the 256 x 256 tile keeps SPLIT rings — three 32-KiB slots of weight (P) half-tiles, two of activation (Q) half-tiles,
//     160 KiB (template flag P3; a weight piece, which comes from HBM inside a model, gets two K-tiles to land) — the other tiles a
//     ring of whole K-tiles {P half0, P half1, Q half0, Q half1}: 3 x 48 KiB for the 128-row tile (2 x 64 KiB: the 256-row tile
//     with P3 = false).  One __shared__ array.  A half-tile is [rows][128 B]; 16-byte chunk c of row r sits at chunk c ^ ((r>>1)&7)
//     (measured conflict-free for ds_read_b128 with both MFMA shapes).  Staging is
//     global_load_lds_dwordx4: the LDS image is lane-linear, the XOR goes on the per-lane SOURCE address.
//   * ONE s_barrier per K-tile (a barrier round trip costs ~130-150 cycles on this chip, measured; an
//     8-barrier ping-pong spent 40 % of the loop in them).  Each wave pipelines itself: fragments are
//     double-buffered in registers and the LDS reads of quadrant q+1 are issued before the MFMAs of
//     quadrant q.  The barrier sits mid-tile (between quadrants 1 and 2): by then every wave has
//     issued+retired its last read of tile t (so tile t's buffer can be refilled with tile t+2) and
//     has waited for its own DMA pieces of tile t+1 (so tile t+1 is visible to all after the barrier).
//     Every DMA therefore has a whole K-tile of MFMA time to land (the P pieces of the split rings: two).
//   * Epilogue (gemm_epilogue.h, epi_staged_block): the tile's scales are DMA'd to LDS (prologue; split rings: tile NT-3); QSPEC E1-E4 in
//     registers; the tile is transposed through a wave-private, XOR-swizzled region of the ring slot that the last K-tile
//     does NOT occupy (no barrier: every wave is past the barrier that freed it) and written with 16-byte stores as
//     256-byte row segments.
//   * The first K-tile is entered as soon as its first half (P0, Q0: 32 KiB) has landed; Q1 and P1 are waited for
//     just before the quadrants that read them (two extra barriers, first tile only).
//   * Dev builds (make ABLATION=1) add compile-time ablated instantiations + cycle/clock stamps (tools/ablate.py).
#include <cstdlib>
#include <type_traits>

#include "gemm_tile_common.h"
#include "kloop_p3_asm.inc"   // generated: tools/gen_kloop_asm.py

namespace pq {

// fused split-K workspace: bytes of the two per-tile counter arrays in front of the slabs
__host__ __device__ constexpr size_t fsk_counter_bytes(int ntiles, int kslices) { return ((size_t)ntiles * 4 * (kslices == 4 ? 4 : 2) + 255) & ~(size_t)255; }   // ticket form: {ticket, ready} per tile; symmetric forms: one flag per slice

// TM: rows (m) of the output tile, 256 or 128.  TM = 128 halves the Q (activation) side — 16 rows per wave-half, one
// 16x16 MFMA tile — for problems whose 256x256 grid would leave most CUs idle; it needs 48 KiB of DMA per
// 1024 MFMA-cycles, so it runs ingest-bound (~3/4 of the 256-row tile's rate per CU).
// TN: columns (n) of the output tile, 256 or 128 (with TM = 128 only): a 128 x 128 tile for 1024-wide shards whose grid would
// otherwise fill half the chip or need split-K.  It needs 32 KiB of DMA and 96 KiB of LDS fragment reads per 512
// MFMA-cycles, so it runs LDS-read/ingest-bound — but with every CU busy and no slab traffic.
// LC (TM = 128, TN = 256 only): the loader / consumer wave split of the ring tile below applied to this kernel — waves 0-3 are
// consumers with the 256-row kernel's wave tile (128 n x 64 m, four 16-MFMA quadrants, 2 x 2 waves over the 256 n x 128 m tile) and
// issue no DMA; waves 4-7 are loaders that issue all 48 pieces of a K-tile (12 each) and do the vmcnt waits; both roles meet at
// the same barriers.  The 8-wave form of this tile runs ~1900 cycles per K-tile against 1024 of MFMA because all 8 waves stall in
// their DMA issues at the same point of the tile; here a SIMD's consumer never issues a DMA.  It also reads 96 instead of 160 KiB
// of LDS fragments per K-tile (the wave tile is twice as tall).
// P3 (TM = TN = 256 only): SPLIT rings — the weight (P) half-tiles in a ring of THREE 32-KiB slots, the activation (Q) half-tiles
// in a ring of two; 160 KiB, all of the LDS.  The weights of a model layer are read once per pass, from HBM: measured, a launch
// whose weights are not in the Infinity Cache runs its K-loop in 92 K instead of 76 K cycles with the 2-deep ring (every K-tile
// waits ~500 cycles for its first-touch pieces: one K-tile of lead is ~1 000 cycles, HBM under load answers in ~1 700), 5-9 % of
// the launch.  Here a weight piece is issued TWO K-tiles before it is needed.  The activations were written by the previous
// kernel and come from the Infinity Cache: one K-tile of lead is enough for them.  No LDS of their own is left for the scale vectors:
// they are DMA'd, two K-tiles before the epilogue, into the P slot that no later tile needs.
// ASMV (P3 only): K-tiles 1 .. NT-4 run in the hand-allocated asm statement kloop_p3_asm<ASMV> (kloop_p3_asm.inc) instead of the HIP loop.
// NCW (LC only): consumer waves, 4 or 8.  8: a 12-wave workgroup, three waves per SIMD (168 registers each) — the eight consumers take
// 64 n x 64 m wave tiles (four 8-MFMA quadrants; 64 accumulator registers), two per SIMD, so one covers the other's operand and
// barrier stalls the way the 256-row tile's two waves per SIMD do; the four loaders are unchanged.
// FSK (P3 + ASMV only): FUSED split-K.  kslices workgroups share one output tile, each over K / kslices; the first kslices - 1 to finish their
// K-loop (an agent-scope ticket counter per tile) dump their 128 accumulator registers to a slab of the workspace (16 B per lane,
// coalesced: the layout is the register file's, only the partner reads it) and leave; the last one waits until those slabs are complete
// — their writers are past their K-loops, so the wait is bounded by a 256-KiB store, never by a workgroup that has not been scheduled —
// adds them to its registers (integer sums commute: the bits do not depend on who is last) and runs the ordinary epilogue.  The hand-over is the
// asm statement behind the K-loop's (fsk_tail_asm, registers pinned: tools/gen_kloop_asm.py gen_fsk_tail).  No second
// pass over int32 slabs of the whole output, and half (or a quarter) of the slab traffic of the two-pass form.  `stamps` carries the
// workspace: [tickets: ntiles u32][ready: ntiles u32] (zeroed by the launcher), padded to 256 B, then the slabs.
// FSK = 2 / 4: the two- / four-slice SYMMETRIC exchange — workgroups S p .. S p + S - 1 share tile p; each keeps one part of the accumulators (S = 2: a column
// half of every wave block; S = 4: a quarter = column half x row half), stores the other parts, waits for the partners' flags, adds their contributions to the
// part it kept and runs the epilogue of that part (fsk_sym2_asm / fsk_sym4_asm): no idle CU, 1 / S of an epilogue each.  The wait is for workgroups that may NOT
// be running (another fused split-K launch on a second stream can hold the CUs their partners need): these forms are opt-in (PQ_FSK_SYMMETRIC=1, pq_hip.h); the default
// is the ticket form above (FSK = 1), which never waits for a workgroup that has not finished its K-loop.
// KSL (FSK = 1 only; round 6): the activation operand arrives STACKED as an all-gather of int8 column blocks leaves it, [G][M][K / G] (pq_qlinear_s8_kslabs) — K-tile kt of
// the whole K lives in slab kt / q_tps at q_slab_stride bytes per slab, ldx = the slab's row length.  A slice's K range starts inside one slab and may run over several:
// the asm K-loop's activation cursor jumps at every slab boundary (kloop_p3_asm<5>: slab_step in tools/gen_kloop_asm.py).  The launcher guarantees that the slices
// cover whole slabs or that a slab holds whole slices, and at least four K-tiles per slab (the HIP-coded prologue and tile 0 request the slice's first three).
// Integer sums have no order: the bits of the row-major GEMM, without the layout pass.
template <int OUT, int ABL, int TM = 256, int TN = 256, bool LC = false, bool P3 = false, int ASMV = 0, int NCW = 4, int FSK = 0, bool KSL = false>   // ABL: compile-time ablation (0 = product)
__global__ __launch_bounds__((LC && NCW == 8) ? 768 : 512, (LC && NCW == 8) ? 3 : 2) void gemm_s8_sp256(const int8_t* __restrict__ X, int64_t ldx,
                                                        const int8_t* __restrict__ W, int64_t ldw, EpiArgs epi,
                                                        int M, int N, int K, int tiles_m, int tiles_n, int dbg, unsigned long long* stamps, int kslices,
                                                        int64_t q_slab_stride = 0, int q_tps = 0) {
    // ablation bits (dev builds only): 1 skip DMA, 2 skip LDS reads, 4 skip MFMA, 8 skip epilogue, 16 direct epilogue
    constexpr bool DBG = ABL != 0;
    constexpr bool no_dma = ABL & 1, no_lds = ABL & 2, no_mma = ABL & 4, no_epi = ABL & 8, direct_epi = ABL & 16;
    constexpr bool no_vmwait = ABL & 32, no_barrier = ABL & 64;   // timing-only: results are wrong
    constexpr bool half_epi = ABL & 2048;                         // timing-only (round 4): the epilogue of ONE column half only — 16 of the 32 MB: a bound on what an early epilogue of half the accumulators can return
    constexpr int SHAPE = 16;                                     // v_mfma_i32_16x16x64_i8
    (void)dbg;
    // dev builds: every wave stamps the chip-wide 100 MHz counter (s_memrealtime) and its shader-cycle counter (s_memtime)
    // at four points — entry, first MFMA possible, K-loop done, epilogue issued — into a buffer of its own (never an output):
    // stamps[((block * 8 + wave) * 4 + point) * 2 + {0: realtime, 1: cycles}]
    auto stamp = [&](int point) {
        if constexpr (DBG) {
            if (stamps != nullptr) {
                const unsigned long long r = __builtin_amdgcn_s_memrealtime(), c = __builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 63) == 0 && threadIdx.x < 512) {
                    unsigned long long* d = stamps + (((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + point) * 2;
                    d[0] = r; d[1] = c;
                }
            }
        }
    };
    stamp(0);
    // LDS ring: a K-tile is {P half0, P half1, Q half0, Q half1}; the 128-row tile's 48-KiB K-tiles fit three deep (two in
    // flight while one is read: 2.5 K-tiles of MFMA time for every DMA to land instead of 1.5 — its K-tile is only 1024
    // MFMA-cycles long, and operands that come from HBM rather than a warm MALL need more than that)
    constexpr int PHB = (TN / 2) * FBK, QHB = (TM / 2) * FBK;   // bytes of a P / Q half-tile
    constexpr int BUFB = 2 * PHB + 2 * QHB;                      // 64, 48 or 32 KiB
    constexpr int NBUF = (TM == 128 && TN == 256) ? 3 : 2;
    constexpr int PB = 2 * PHB, QB = 2 * QHB;                    // bytes of one K-tile's P / Q side
    constexpr int NPB = P3 ? 3 : NBUF, NQB = P3 ? 2 : NBUF;      // ring depths of the two sides (equal, and interleaved slot by slot, unless P3)
    constexpr int QBASE = P3 ? NPB * PB : 0;                     // P3: [P slot 0..2][Q slot 0..1]; otherwise [slot: P h0, P h1, Q h0, Q h1]...
    constexpr int SCALE_OFF = NBUF * BUFB;                                                 // ring (and, in a free slot, the epilogue staging) below, scales above
    __shared__ __attribute__((aligned(16))) uint8_t smem[P3 ? NPB * PB + NQB * QB : SCALE_OFF + 2048];   // (not P3) + 1 KiB row scales + 1 KiB column scales

    static_assert(!LC || (TM == 128 && TN == 256), "loader / consumer split: 128 x 256 tile only");
    static_assert(!P3 || (TM == 256 && TN == 256 && !LC), "split rings: 256 x 256 tile only");
    static_assert(ASMV == 0 || (P3 && (ABL & ~(1024 | 8 | 2048)) == 0), "asm K-loop: split-ring tile only (dev builds: with stamps, and the epilogue ablations)");
    static_assert(FSK == 0 || (P3 && ASMV != 0 && ABL == 0 && OUT != OUT_I32), "fused split-K: the asm split-ring kernel with a dequantising epilogue");
    static_assert(!KSL || FSK == 1, "K-slab walk: the ticket form of the fused split-K only");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(NCW == 4 || (LC && NCW == 8), "consumer waves");
    const bool loader = LC && wave >= NCW;
    const int w = LC ? (loader ? wave - NCW : wave) : wave;   // index inside the role (issuer index for DMA pieces, consumer index for tiles)
    constexpr int NWQ = LC ? 2 : 4;                      // consumer waves along m
    constexpr int NWP = LC ? NCW / 2 : 2;                // ... along n
    constexpr int NISS = LC ? 4 : 8;                     // waves that issue DMA pieces
    const int wp = w / NWQ, wq = w % NWQ;

    // ---- tile assignment: XCD remap, then grouped order (GM m-tiles per band) for L2 reuse
    // split-K (int32 output only): blocks [s*ntiles, (s+1)*ntiles) own K-slice s and write their exact partial
    // accumulators to slab s of the output buffer; a separate kernel sums the slabs and applies the epilogue.
    const int ntiles_all = tiles_m * tiles_n;
    const int kslice = FSK >= 2 ? ((int)blockIdx.x & (FSK - 1)) : (int)blockIdx.x / ntiles_all;
    // K-tiles of this slice: the K / FBK tiles are dealt as evenly as they go (round 6: the first (K / FBK) % kslices slices take one more — three slices of a 64-tile K)
    const int nt_base = (K / FBK) / kslices, nt_rem = (K / FBK) - nt_base * kslices;
    const int kt_start = kslice * nt_base + (kslice < nt_rem ? kslice : nt_rem);
    const int Ks = (nt_base + (kslice < nt_rem ? 1 : 0)) * FBK;      // bytes of K of this slice (a multiple of FBK)
    int t;
    if constexpr (FSK >= 2) {      // partners S p .. S p + S - 1 (neighbouring XCDs); every (8 / S)-th partner set shares those XCDs: a contiguous run of tiles for them
        constexpr int NGX = 8 / FSK;
        const int p = (int)blockIdx.x / FSK, qg = ntiles_all / NGX, rg = ntiles_all % NGX, grp = p % NGX, idx = p / NGX;
        t = (grp < rg ? grp * (qg + 1) : rg * (qg + 1) + (grp - rg) * qg) + idx;
    } else {
        t = xcd_remap((int)blockIdx.x - kslice * ntiles_all, ntiles_all, epi.nxcd);
    }
    constexpr int GM = 4;
    const int band = t / (GM * tiles_n);
    const int gm = (tiles_m - band * GM) < GM ? (tiles_m - band * GM) : GM;
    const int tin = t - band * GM * tiles_n;
    const int tm = band * GM + tin % gm, tn = tin / gm;
    const int m0 = tm * TM, n0 = tn * TN;

    static_assert(TM == 256 || TM == 128, "tile rows");
    static_assert(TN == 256 || (TN == 128 && TM == 128), "TN = 128 comes with TM = 128");
    constexpr int PWH = (TN / 2) / NWP;           // P rows per wave per half-tile: 64 or 32
    constexpr int PPW = (TN / 16) / NISS;         // DMA pieces per issuing wave per P half-tile (a P half-tile is TN/2 rows = TN/16 pieces)
    constexpr int QW = (TM / 2) / NWQ;            // Q rows per wave per half-tile: 32 or 16
    constexpr int QPW = (TM / 16) / NISS;         // DMA pieces per issuing wave per Q half-tile (a Q half-tile is TM/2 rows)
    // ---- staging source offsets: wave w issues pieces (w*2+jj), jj = 0,1, of every half-tile.
    // LDS row r = w*16 + jj*8 + (lane>>3) of the half-tile; physical chunk lane&7.
    // P half h, LDS row r <-> n_local = (r/PWH)*2PWH + h*PWH + (r%PWH);  Q: m_local = (r/QW)*2QW + h*QW + (r%QW).
    uint32_t offP[2][PPW], offQ[2][QPW];   // [half][jj] byte offset from the tile's first row, k = 0
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int piece = w * PPW + jj;               // LDS rows piece*8 .. +7 of the P half-tile
            const int r = piece * 8 + (lane >> 3);
            const int src_chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
            int nl = (r / PWH) * (2 * PWH) + h * PWH + (r % PWH);
            nl = (n0 + nl < N) ? nl : (N - 1 - n0);       // clamp: rows past the edge re-read a valid row
            offP[h][jj] = (uint32_t)nl * (uint32_t)ldw + src_chunk * 16;
        }
#pragma unroll
        for (int jj = 0; jj < QPW; ++jj) {
            const int piece = w * QPW + jj;               // LDS rows piece*8 .. +7 of the Q half-tile
            const int r = piece * 8 + (lane >> 3);
            const int src_chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
            int ml = (r / QW) * (2 * QW) + h * QW + (r % QW);
            ml = (m0 + ml < M) ? ml : (M - 1 - m0);
            offQ[h][jj] = (uint32_t)ml * (uint32_t)ldx + src_chunk * 16;
        }
    }
    const int8_t* gP = W + (int64_t)n0 * ldw + (int64_t)kt_start * FBK;   // uniform; advanced by FBK per staged K-tile
    const int8_t* gQ = X + (int64_t)m0 * ldx + (int64_t)kt_start * FBK;
    uint32_t qs_cnt = 0x7fffffffu;          // KSL: K-tiles whose activation pieces are still to be requested from the current slab, counted from the asm loop's first tile, minus one
    if constexpr (KSL) {
        const int kt0 = kt_start, sl0 = kt0 / q_tps, in0 = kt0 - sl0 * q_tps;
        gQ = X + (int64_t)m0 * ldx + (int64_t)sl0 * q_slab_stride + (int64_t)in0 * FBK;
        if (Ks / FBK > q_tps - in0) qs_cnt = (uint32_t)__builtin_amdgcn_readfirstlane(q_tps - in0 - 4);      // (three tiles are requested before the statement: launcher, q_tps - in0 >= 4)
    }
    // K walk of the loader / consumer form (round 4; `dbg` carries the chunk length in K-tiles for that form, 0 = the plain walk): the up-to-4 workgroups of a band
    // that stream one weight panel (neighbours on one XCD, started together) walk each chunk of K-tiles from different starting points and wrap inside the chunk —
    // gemm_s8_ring.hip, "K ROTATION in chunks".  Integer sums: same bits.
    const bool rot_on = LC && !P3 && dbg > 0;
    const int rot_ct = rot_on ? dbg : (1 << 30);
    const int8_t* const gP0 = gP;
    const int8_t* const gQ0 = gQ;
    auto rot_of = [&](int len) { return rot_on ? (int)(((int64_t)(tin % gm) * len) / gm) : 0; };
    int cbase = 0, clen = rot_ct < Ks / FBK ? rot_ct : Ks / FBK;
    int cpos = rot_of(clen), cleft = clen;
    if constexpr (LC && !P3) {
        const int koff0 = __builtin_amdgcn_readfirstlane(cpos * FBK);
        gP += koff0; gQ += koff0;
    }
    const int piece_off = w * PPW * 1024;       // this wave's PPW pieces of a P half-tile
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lptr_t)smem;   // LDS byte address of the array

    // ---- fragment read addresses (lane part)
    constexpr int NPI = PWH / 16;  // P tiles per wave-half (PWH rows)
    constexpr int NQJ = QW / 16;   // Q tiles per wave-half (QW rows)
    constexpr int NKS = 2;         // MFMA k-steps per 128-byte row
    constexpr int NACC = 4;        // accumulator registers per tile
    const int frow = lane & 15;
    const int fchunk = lane >> 4;
    const int fkey = (frow >> 1) & 7;
    uint32_t lP[NKS], lQ[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = ks * (8 / NKS) + fchunk;
        lP[ks] = (uint32_t)((wp * PWH + frow) * 128 + ((c ^ fkey) * 16));
        lQ[ks] = (uint32_t)((wq * QW + frow) * 128 + ((c ^ fkey) * 16));
    }

    using acc_t = v4i;
    acc_t acc[2][2][NPI][NQJ];          // zeroed behind the prologue's DMA issue (below): 128 v_mov in front of the first piece were ~300 cycles of every tile's start

    v4i fPa[NPI][NKS], fPb[NPI][NKS], fQa[NQJ][NKS], fQb[NQJ][NKS];
    if constexpr (DBG) {   // defined operands when LDS reads are ablated
#pragma unroll
        for (int i = 0; i < NPI; ++i)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) { fPa[i][ks] = v4i{lane, i, ks, 1}; fPb[i][ks] = v4i{lane, i, ks, 4}; }
#pragma unroll
        for (int j = 0; j < NQJ; ++j)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) { fQa[j][ks] = v4i{lane, j, ks, 2}; fQb[j][ks] = v4i{lane, j, ks, 3}; }
    }

    // ---- issue items.  Everything below is hand-interleaved: ONE item in the shadow of each MFMA.
    // DMA piece g of a K-tile, in need order P0 (PPW pieces) | Q0 (QPW pieces) | Q1 (QPW pieces) | P1 (PPW pieces)
    constexpr int NDMA = 2 * PPW + 2 * QPW;
    // LDS byte offset of slot ps of the P ring / slot qs of the Q ring
    auto p_off = [&](int ps) { return P3 ? ps * PB : ps * BUFB; };
    auto q_off = [&](int qs) { return P3 ? QBASE + qs * QB : qs * BUFB + PB; };
    auto dma_piece = [&](auto isq_c, auto h_c, auto jj_c, int off) {   // one 1-KiB piece of half h of the side whose slot starts at `off`
        constexpr bool isQ = decltype(isq_c)::value;
        constexpr int h = decltype(h_c)::value, jj = decltype(jj_c)::value;
        if (!no_dma) {
            if constexpr (isQ) glds16_sbase(gQ, offQ[h][jj], smem_base + off + h * QHB + (w * QPW + jj) * 1024);
            else glds16_sbase(gP, offP[h][jj], smem_base + off + h * PHB + piece_off + jj * 1024);
        }
    };
    auto dma_item = [&](int buf, auto gc) {       // both sides of one K-tile into slot `buf` of each ring, need order P0 | Q0 | Q1 | P1
        constexpr int g = decltype(gc)::value;
        if constexpr (g < NDMA) {
            constexpr bool isQ = (g >= PPW && g < PPW + 2 * QPW);
            constexpr int h = isQ ? (g - PPW) / QPW : (g >= PPW ? 1 : 0);
            constexpr int jj = isQ ? (g - PPW) % QPW : (g < PPW ? g : g - PPW - 2 * QPW);
            dma_piece(std::integral_constant<bool, isQ>{}, std::integral_constant<int, h>{}, std::integral_constant<int, jj>{}, isQ ? q_off(buf) : p_off(buf));
            if constexpr (g == NDMA - 1) {
                if constexpr (LC && !P3) {
                    if (++cpos == clen) cpos = 0;
                    if (--cleft == 0) {
                        cbase += clen;
                        clen = Ks / FBK - cbase < rot_ct ? Ks / FBK - cbase : rot_ct;
                        cleft = clen;
                        cpos = clen > 0 ? rot_of(clen) : 0;
                    }
                    const int koff = __builtin_amdgcn_readfirstlane((cbase + cpos) * FBK);
                    gP = gP0 + koff; gQ = gQ0 + koff;
                } else {
                    gP += FBK; gQ += FBK;
                }
            }
        }
    };
    // P3 steady state: the Q side of tile kt+2 FIRST (it is needed one K-tile earlier: the mid-tile wait may leave the P pieces behind
    // it in flight), then — if tile kt+3 exists — its P side.  gQ and gP walk independently (gQ is one tile behind gP).
    auto dma_item3 = [&](int ps, int qs, auto gc, auto with_p) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < 2 * QPW) {
            dma_piece(std::true_type{}, std::integral_constant<int, g / QPW>{}, std::integral_constant<int, g % QPW>{}, q_off(qs));
            if constexpr (g == 2 * QPW - 1) gQ += FBK;
        } else if constexpr (g < NDMA && decltype(with_p)::value) {
            constexpr int gp = g - 2 * QPW;
            dma_piece(std::false_type{}, std::integral_constant<int, gp / PPW>{}, std::integral_constant<int, gp % PPW>{}, p_off(ps));
            if constexpr (g == NDMA - 1) gP += FBK;
        }
    };
    auto stage_tile = [&](int buf) {   // whole tile at once (prologue only)
        static_for<NDMA>([&](auto gc) { dma_item(buf, gc); });
    };
    auto stage_q = [&](int qs) { static_for<2 * QPW>([&](auto gc) { dma_item3(0, qs, gc, std::false_type{}); }); };                               // P3 prologue
    auto stage_p = [&](int ps) { static_for<2 * PPW>([&](auto gc) { dma_item3(ps, 0, std::integral_constant<int, 2 * QPW + decltype(gc)::value>{}, std::true_type{}); }); };
    // fragment item it: P: i = it % NPI, ks = it / NPI (NPR items); Q: j = it % NQJ, ks = it / NQJ (NQR items)
    auto readP_item = [&](int poff, int h, v4i (&f)[NPI][NKS], auto ic) {        // poff = p_off(slot)
        constexpr int it = decltype(ic)::value, i = it % NPI, ks = it / NPI;
        if (!no_lds) f[i][ks] = *reinterpret_cast<const v4i*>(smem + poff + lP[ks] + h * PHB + i * SHAPE * 128);
    };
    auto readQ_item = [&](int qoff, int h, v4i (&f)[NQJ][NKS], auto ic) {        // qoff = q_off(slot)
        constexpr int it = decltype(ic)::value, j = it % NQJ, ks = it / NQJ;
        if (!no_lds) f[j][ks] = *reinterpret_cast<const v4i*>(smem + qoff + lQ[ks] + h * QHB + j * SHAPE * 128);
    };
    constexpr int NPR = NPI * NKS;               // P fragment reads per half: 8 or 4
    auto readP = [&](int poff, int h, v4i (&f)[NPI][NKS]) { static_for<NPR>([&](auto ic) { readP_item(poff, h, f, ic); }); };
    constexpr int NQR = NQJ * NKS;               // Q fragment reads per half: 4 or 2
    auto readQ = [&](int qoff, int h, v4i (&f)[NQJ][NKS]) { static_for<NQR>([&](auto ic) { readQ_item(qoff, h, f, ic); }); };

    constexpr int NM = NKS * NPI * NQJ;          // MFMAs per quadrant: 16, 8 or 4
    constexpr int PPS = NPR / (NM / 2);          // P reads per slot (half a quadrant's slots): 1 or 2
    constexpr int DPS = (NDMA + NM / 2 - 1) / (NM / 2);   // DMA pieces per slot: 1 or 2
    static_assert(PPS >= 1 && PPS * (NM / 2) == NPR && NQR <= NM, "slot plan");
    // One quadrant: MFMA idx, then slot(idx) in its shadow.  The first MFMA needs this quadrant's own
    // operands (all issued during the previous quadrant) -> hipcc's wait there is an exact lgkmcnt(0).
    auto mma = [&](acc_t (&c)[NPI][NQJ], v4i (&fp)[NPI][NKS], v4i (&fq)[NQJ][NKS], auto&& slot) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<NM>([&](auto xc) {
            // MFMA order inside a quadrant: the P fragment is shared by NQJ consecutive MFMAs.  (Q-fragment-stationary and
            // accumulator-stationary orders measured 0.7-1.3 % slower: profiles/r02_ab_experiments.txt)
            constexpr int x = decltype(xc)::value, ks = x / (NPI * NQJ), i = (x / NQJ) % NPI, j = x % NQJ;
            if (!no_mma) {
                c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fp[i][ks], fq[j][ks], c[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            slot(xc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    const int NT = Ks / FBK;

    // this wave's WM(m) x WN(n) block of the output tile (epilogue coordinates; D[row <-> n][col <-> m]: a lane holds 4 consecutive n)
    constexpr int WM = 2 * QW;                                              // rows (m) of this wave's block: 64 or 32
    constexpr int WN = 2 * PWH;                                             // columns (n) of this wave's block: 128 or 64
    const int wm0 = m0 + wq * WM, wn0 = n0 + wp * WN;
    // the staged epilogue reads the scale vectors 16 bytes at a time from an LDS image filled by DMA (scale_dma below)
    const bool scales_ok = (OUT == OUT_I32) ||
        ((((reinterpret_cast<uintptr_t>(epi.a_scale) | reinterpret_cast<uintptr_t>(epi.b_scale)) & 15) == 0) && M >= 4 && N >= 4);
    // Where the tile's 256 row scales and 256 column scales sit in LDS for the epilogue (1 KiB each, by DMA).  P3 has no LDS of its
    // own left for them: they go to the P slot that tile NT-3 vacates — no later tile's DMA targets it — issued in that tile (or, for
    // NT < 3, in the prologue: that slot is never used then), i.e. two K-tiles before the epilogue needs them.
    const int scale_off = P3 ? p_off((Ks / FBK) % 3) : SCALE_OFF;
    auto scale_dma = [&]() {       // waves 0 and 1 (of the issuing role): 4 floats per lane, clamped at the matrix edge
        const int base = w == 0 ? m0 : n0, lim = w == 0 ? M : N;
        int e0 = base + lane * 4;
        e0 = e0 + 3 < lim ? e0 : (lim >= 4 ? lim - 4 : 0);     // edge tiles: any valid address (values unused there)
        const float* src = (w == 0 ? epi.a_scale : epi.b_scale) + e0;
        glds16_vaddr(src, smem_base + scale_off + w * 1024);
    };

    // One K-tile, branch-free inside (flags are compile-time).  Quadrant order (P0,Q0) (P0,Q1) (P1,Q0)
    // (P1,Q1): every register set returns to the same role each tile.  Entry: fPa = P0[kt], fQa = Q0[kt].
    //   q0: MFMA acc[0][0] (fPa, fQa)   | slots: read Q1[kt] -> fQb
    //   q1: MFMA acc[0][1] (fPa, fQb)   | slots: read P1[kt] -> fPb            <- last LDS read of tile kt
    //   --- vmcnt(0) + lgkmcnt(0) + s_barrier: tile kt+1 visible to all, tile kt's buffer free ---
    //   q2: MFMA acc[1][0] (fPb, fQa)   | slots: read P0[kt+1] -> fPa, then DMA tile kt+2
    //   q3: MFMA acc[1][1] (fPb, fQb)   | slots: read Q0[kt+1] -> fQa (free after q2)
    // flags: next = tile kt+1 exists, next2 = tile kt+2 exists, dma = tile kt+NBUF exists (its DMA is issued here, into the
    // slot tile kt vacates).  slot = kt % NBUF is carried by the caller.
    // Tile 0 only (a wave-uniform branch on kt — a peeled copy of the body made hipcc fold the accumulators' zero into the
    // first MFMAs' C operand and spill): the prologue waited for P0 and Q0 alone; Q1 and P1 are waited for (and published
    // with a barrier) right before the quadrants whose shadows read them.  VMA = DMA pieces issued after tile 0's in the
    // prologue, a compile-time function of the flavour.
    // P3: `slot` is the P ring's slot (kt % 3), `qslot` the Q ring's (kt % 2); has_dma = tile kt+3 exists (its P side is issued here; the Q
    // side of tile kt+2 whenever that tile exists).  Otherwise qslot == slot.
    auto tile = [&](int kt, int slot, int qslot, auto has_next, auto has_next2, auto has_dma) {
        const int poff = p_off(slot), qoff = q_off(qslot);
        const int poffn = p_off(slot + 1 == NPB ? 0 : slot + 1), qoffn = q_off(qslot + 1 == NQB ? 0 : qslot + 1);
        constexpr bool next = decltype(has_next)::value, next2 = decltype(has_next2)::value, dma = decltype(has_dma)::value;
        // VMA: DMA pieces younger than tile 0's when tile 0 waits for its Q1 / P1.  P3: tile 1 from the prologue; the P side of tile 2 is
        // issued in tile 0's first quadrant (below) — in the prologue it cost every wave four more issue stalls before the first MFMA
        constexpr int VMA = !next ? 0 : (P3 ? NDMA : ((NBUF == 3 && next2) ? 2 * NDMA : NDMA));
        constexpr int VMA1 = VMA + ((P3 && next2) ? 2 * PPW : 0);
        // (LC: the loaders do the vmcnt waits; a consumer only retires its own LDS reads before each barrier)
        if (!no_vmwait && kt == 0) { __builtin_amdgcn_s_waitcnt(waitcnt_imm(LC ? 63 : VMA + PPW, 0)); if constexpr (!no_barrier) __builtin_amdgcn_s_barrier(); }   // Q1 of tile 0 visible
        mma(acc[0][0], fPa, fQa, [&](auto xc) {
            constexpr int x = decltype(xc)::value;
            if constexpr (x < NQR) readQ_item(qoff, 1, fQb, xc);
            if constexpr (P3 && next2 && x >= NQR && x < NQR + 2 * PPW) {
                if (kt == 0) dma_item3(2, 0, std::integral_constant<int, 2 * QPW + (x - NQR)>{}, std::true_type{});       // P side of tile 2 -> P slot 2
            }
        });
        if (!no_vmwait && kt == 0) { __builtin_amdgcn_s_waitcnt(waitcnt_imm(LC ? 63 : VMA1, 0)); if constexpr (!no_barrier) __builtin_amdgcn_s_barrier(); }        // P1 of tile 0 visible
        mma(acc[0][1], fPa, fQb, [&](auto xc) {
            constexpr int x = decltype(xc)::value;
            if constexpr (x < NM / 2) static_for<PPS>([&](auto pc) { readP_item(poff, 1, fPb, std::integral_constant<int, x * PPS + decltype(pc)::value>{}); });
        });
        if constexpr (next) {
            // tile kt+1 must have landed; with a 3-deep ring tile kt+2 (NDMA pieces per wave) may stay in flight; with split rings
            // the P side of tile kt+2 (2 * PPW pieces, issued behind the Q side of tile kt+1) may
            if constexpr (no_vmwait || LC) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));   // lgkmcnt(0) only
            else if constexpr (P3 && next2) __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * PPW, 0));
            else if constexpr (NBUF == 3 && next2) __builtin_amdgcn_s_waitcnt(waitcnt_imm(NDMA, 0));
            else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));   // vmcnt(0) lgkmcnt(0): builtin form, so hipcc's scoreboard knows
            if constexpr (!no_barrier) __builtin_amdgcn_s_barrier();
        }
        mma(acc[1][0], fPb, fQa, [&](auto xc) {
            constexpr int x = decltype(xc)::value;
            if constexpr (next && x < NM / 2) static_for<PPS>([&](auto pc) { readP_item(poffn, 0, fPa, std::integral_constant<int, x * PPS + decltype(pc)::value>{}); });
            if constexpr (P3) {
                if constexpr (next2 && x >= NM / 2)
                    static_for<DPS>([&](auto pc) { dma_item3(slot, qslot, std::integral_constant<int, (x - NM / 2) * DPS + decltype(pc)::value>{}, has_dma); });
                // tile NT-3 (the one flavour with a tile kt+2 but no tile kt+3): the scale vectors follow the Q pieces into the P slot
                // this tile has just vacated; the vmcnt(0) of tile NT-2's mid-tile wait covers them, its barrier publishes them
                if constexpr (next2 && !dma && OUT != OUT_I32 && x == NM / 2 + 2 * QPW) {
                    if (scales_ok && w < 2) scale_dma();
                }
            } else {
                if constexpr (dma && !LC && x >= NM / 2) static_for<DPS>([&](auto pc) { dma_item(slot, std::integral_constant<int, (x - NM / 2) * DPS + decltype(pc)::value>{}); });
            }
        });
        mma(acc[1][1], fPb, fQb, [&](auto xc) {
            constexpr int x = decltype(xc)::value;
            if constexpr (next && x < NQR) readQ_item(qoffn, 0, fQa, xc);
        });
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};

    // ---- prologue: the tile's 256 row scales and 256 column scales go to LDS by DMA (waves 0 and 1, 4 floats per
    // lane, clamped at the matrix edge) so the epilogue never waits on a global load; then tiles 0 and 1.
    // (only when both scale vectors are 16-byte aligned and at least 4 long; otherwise the direct epilogue is used)
    const bool scales_in_lds = scales_ok;
    if constexpr (OUT != OUT_I32) {
        if (scales_in_lds && w < 2 && (!LC || loader) && (!P3 || Ks / FBK < 3))
            scale_dma();                                             // (one more piece ahead of this wave's tile pieces: waited for with them)
    }
    // DMA pieces (per issuing wave) behind tile 0's
    const int vm_after = P3 ? (NT > 1 ? NDMA : 0) : ((NT < NBUF ? NT : NBUF) - 1) * NDMA;
    if (!LC || loader) {
        stage_tile(0);
        if constexpr (P3) {            // tile 1, Q side first (the P side of tile 2 follows in tile 0's first quadrant): gQ and gP end at tile 2
            if (NT > 1) { stage_q(1); stage_p(1); }
        } else {
            if (NT > 1) stage_tile(1);
            if (NBUF == 3 && NT > 2) stage_tile(2);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NPI; ++i)
#pragma unroll
                for (int j = 0; j < NQJ; ++j) {
                    acc[a][b][i][j] = v4i{0, 0, 0, 0};
                    // (asm K-loop kernels: keep the zeros in registers — folded into the C operand of the peeled tile 0 they cost that tile 64 more live registers: spills)
                    if constexpr (ASMV != 0) asm volatile("" : "+v"(acc[a][b][i][j]));
                }
    __builtin_amdgcn_sched_barrier(0);
    if (!LC || loader) {
        // wait for the first half of tile 0 only (P0, Q0 — pieces are issued in need order P0 | Q0 | Q1 | P1): the rest of
        // tile 0 and the other staged tiles stay in flight
        if constexpr (no_vmwait) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
        else wait_vmcnt_lgkm0(vm_after + NDMA - PPW - QPW);
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (LC) {
        if (loader) {
            // The loader's K-loop mirrors the consumers' barriers one for one.  Tile 0: Q1, then P1, each behind its wait;
            // then per K-tile: wait until this wave's pieces of tile kt+1 have landed (tile kt+2 may stay in flight), the
            // mid-tile barrier, and the 12 pieces of tile kt+3 into the slot tile kt vacates.
            wait_vmcnt_lgkm0(vm_after + PPW); __builtin_amdgcn_s_barrier();
            wait_vmcnt_lgkm0(vm_after); __builtin_amdgcn_s_barrier();
            int lslot = 0;
            for (int lk = 0; lk < NT; ++lk) {
                if (lk + 1 < NT) {
                    if (lk + 2 < NT) __builtin_amdgcn_s_waitcnt(waitcnt_imm(NDMA, 15));
                    else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
                    __builtin_amdgcn_s_barrier();
                }
                if (lk + NBUF < NT) stage_tile(lslot);
                lslot = (lslot + 1 == NBUF) ? 0 : lslot + 1;
            }
            return;
        }
    }
    stamp(1);
    readP(p_off(0), 0, fPa);
    readQ(q_off(0), 0, fQa);

    int kt = 0, slot = 0, qslot = 0;
    constexpr int DEPTH = P3 ? 3 : NBUF;      // tiles ahead of the current one whose (P-side) DMA is issued in it
    auto adv = [&]() { ++kt; slot = (slot + 1 == NPB) ? 0 : slot + 1; qslot = (qslot + 1 == NQB) ? 0 : qslot + 1; };
    if constexpr (ASMV != 0) {
        // Tile 0 (first-half prologue) in HIP, then tiles 1 .. NT-1 in the asm statement: it continues the HIP tile's protocol exactly
        // (fPa / fQa hold tile 1's first fragments, gQ points at tile 3, gP at tile 4, the per-wave DMA issue order and the vmcnt
        // counts are the HIP loop's), including the scale-vector DMA of tile NT-3.  The launcher sends only K >= 5 * 128 here.
        {
            tile(0, 0, 0, yes, yes, yes);
            const uint32_t bp[2] = {smem_base + lP[0], smem_base + lP[1]};
            const uint32_t bph[2] = {bp[0] + 2 * PB, bp[1] + 2 * PB};
            const uint32_t bq[2] = {smem_base + QBASE + lQ[0], smem_base + QBASE + lQ[1]};
            const int sbase = w == 0 ? m0 : n0, slim = w == 0 ? M : N;          // (scale_dma's source address)
            int e0 = sbase + lane * 4;
            e0 = e0 + 3 < slim ? e0 : (slim >= 4 ? slim - 4 : 0);
            const float* ssrc = (w == 0 ? epi.a_scale : epi.b_scale) + e0;
            const uint32_t do_scales = (uint32_t)__builtin_amdgcn_readfirstlane((int)(OUT != OUT_I32 && scales_ok && w < 2));   // ("s" operands must be provably uniform)
            if constexpr (ASMV == 11) {      // dev builds, timing only: the Q operand straight from L2 — the per-lane offsets of the MFMA operand layout (16 rows x 64 B per load)
                uint32_t offQd[2][QPW];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int jj = 0; jj < QPW; ++jj) {
                        int r = wq * WM + h * (WM / 2) + jj * 16 + (lane & 15);
                        r = m0 + r < M ? r : M - 1 - m0;
                        offQd[h][jj] = (uint32_t)r * (uint32_t)ldx + (uint32_t)(lane >> 4) * 16u;
                    }
                kloop_p3_asm<ASMV>(acc, fPa, fPb, fQa, fQb, bp, bph, bq, offP, offQd, gP, gQ, (uint32_t)(NT - 4), smem_base + (uint32_t)piece_off,
                                   0u, ssrc, smem_base + (uint32_t)w * 1024u, do_scales, (uint32_t)(wave >> 2));
            } else if constexpr (KSL) {
                const int64_t qdelta = q_slab_stride - (int64_t)q_tps * FBK;
                kloop_p3_asm<5>(acc, fPa, fPb, fQa, fQb, bp, bph, bq, offP, offQ, gP, gQ, (uint32_t)(NT - 4), smem_base + (uint32_t)piece_off,
                                0u, ssrc, smem_base + (uint32_t)w * 1024u, do_scales, (uint32_t)(wave >> 2),
                                qs_cnt, (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qdelta), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(qdelta >> 32)),
                                (uint32_t)__builtin_amdgcn_readfirstlane(q_tps - 1));
            } else
            kloop_p3_asm<FSK != 0 ? 4 : ASMV>(acc, fPa, fPb, fQa, fQb, bp, bph, bq, offP, offQ, gP, gQ, (uint32_t)(NT - 4), smem_base + (uint32_t)piece_off,
                               0u, ssrc, smem_base + (uint32_t)w * 1024u, do_scales, (uint32_t)(wave >> 2));
            if constexpr (FSK == 1) {
                // the hand-over: the first kslices - 1 workgroups of a tile to arrive END inside this statement, the last leaves it with the tile's sums
                unsigned* const ctr = reinterpret_cast<unsigned*>(stamps);
                const uint8_t* const slab0 = reinterpret_cast<const uint8_t*>(stamps) + fsk_counter_bytes(ntiles_all, kslices) + (size_t)t * (size_t)(kslices - 1) * (256 * 256 * 4);
                fsk_tail_asm(acc, ctr + t, ctr + ntiles_all + t, slab0, (uint32_t)(kslices - 1), smem_base + (uint32_t)scale_off + 2048u, (uint32_t)wave,
                             (uint32_t)__builtin_amdgcn_readfirstlane(dbg));          // (`dbg` of this kernel form: PQ_FSK_FENCED)
            } else if constexpr (FSK >= 2) {
                // the exchange: this workgroup leaves the statement with the tile's sums in accumulator part `kslice`
                const unsigned* const flags = reinterpret_cast<const unsigned*>(stamps) + FSK * t;
                const uint8_t* const slabs = reinterpret_cast<const uint8_t*>(stamps) + fsk_counter_bytes(ntiles_all, FSK) + (size_t)t * (size_t)(FSK - 1) * (256 * 256 * 4);
                if constexpr (FSK == 2) fsk_sym2_asm(acc, flags, slabs, (uint32_t)wave, (uint32_t)kslice);
                else fsk_sym4_asm(acc, flags, slabs, (uint32_t)wave, (uint32_t)kslice);
            }
        }
    } else {
    while (kt + DEPTH < NT) { tile(kt, slot, qslot, yes, yes, yes); adv(); }
    if constexpr (DEPTH == 3) {
        if (kt + 2 < NT) { tile(kt, slot, qslot, yes, yes, no); adv(); }
    }
    if (kt + 1 < NT) { tile(kt, slot, qslot, yes, no, no); adv(); }
    tile(kt, slot, qslot, no, no, no);
    }

    stamp(2);
    // epilogue lane coordinates.  (FSK: from the exec mask, behind an opaque barrier — that kernel has no register left to carry a
    // lane-derived value across the asm K-loop: hipcc spilled them, and 16 accumulators with them)
    int lane_e = lane;
    if constexpr (FSK != 0) {
        lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_e));
    }
    const int dcol = lane_e & 15;                                           // m inside a Q tile
    const int drow4 = (lane_e >> 4) * 4;                                    // first of 4 consecutive n
    // ---- K4 epilogue: D[row <-> n][col <-> m]; lane holds 4 consecutive n per register group.
    if (no_epi) {   // keep the accumulators live, write (almost) nothing
        int sink = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < NPI; ++i)
#pragma unroll
                    for (int j = 0; j < NQJ; ++j)
#pragma unroll
                        for (int r = 0; r < NACC; ++r) sink ^= acc[a][b][i][j][r];
        if (sink == 0x12345678) reinterpret_cast<int*>(epi.y)[tid] = sink;
        return;
    }

    using O = typename OutElem<OUT>::type;
    constexpr int OB = (int)sizeof(O);
    O* y = reinterpret_cast<O*>(epi.y) + (FSK != 0 ? 0 : (int64_t)kslice * M * epi.ldy);   // slab of this K-slice (kslices == 1 or fused split-K: the output itself)
    const bool has_bias = (OUT != OUT_I32) && epi.bias != nullptr;
    constexpr int NG = 1;

    // (HALF = -1: all columns of the wave block; 0 / 1: one column half of it; HALFQ likewise for its rows — the symmetric fused split-K forms finish
    // acc[HALF][..] (two slices) or acc[HALF][HALFQ][..] (four) only)
    auto epilogue = [&](auto half_c, auto halfq_c) {
        constexpr int HALF = decltype(half_c)::value, HALFQ = decltype(halfq_c)::value;
        constexpr int COL0 = HALF < 0 ? 0 : HALF * PWH;             // first column / row of the part inside the wave block
        constexpr int ROW0 = HALFQ < 0 ? 0 : HALFQ * QW;
        // staged path: whole block in range, 16-byte aligned rows
        const bool staged = !direct_epi && scales_in_lds && (wm0 + WM <= M) && (wn0 + WN <= N) &&
                            epi_rows_storable(epi, y, OB) &&
                            (!has_bias || (reinterpret_cast<uintptr_t>(epi.bias) & (4 * OB - 1)) == 0);
        if (staged) {
            // Staging region: this wave's eighth of the ring slot AFTER the last K-tile's.  Nobody reads that slot any more
            // (its tile was consumed before a barrier every wave has passed) and no DMA targets it (the last NBUF tiles issue
            // none), so a wave that finishes early starts its epilogue under the MFMAs of the slower ones: no barrier.
            constexpr int WREG = BUFB / (LC ? NCW : 8);                         // 8, 6 or 4 KiB (LC: 12 KiB for each of 4 consumers, 6 KiB for each of 8)
            // (P3: the P slot and the Q slot of tile NT-2 — read for the last time before a barrier every wave has passed, and no DMA
            // targets them again — four waves each; the third free slot, tile NT-3's P slot, holds the scales)
            const int last_slot = (NT - 1) % NBUF;
            const uint32_t sw_off = P3 ? (uint32_t)((w < 4 ? p_off((NT + 1) % 3) : q_off(NT % 2)) + (w & 3) * WREG)
                                       : (uint32_t)((last_slot + 1 == NBUF ? 0 : last_slot + 1) * BUFB + w * WREG);
            constexpr int NPT = (HALF < 0 ? 2 : 1) * NPI, NQT = (HALFQ < 0 ? 2 : 1) * NQJ;   // column / row tiles of the wave block (of the part)
            constexpr int PT_PASS = (NPT * 16 * OB > 256) ? NPT / 2 : NPT;      // staged rows of at most 256 bytes
            constexpr int QT_PASS_MAX = WREG / (16 * PT_PASS * 16 * OB);
            constexpr int QT_PASS = QT_PASS_MAX >= NQT ? NQT : (QT_PASS_MAX >= 2 ? 2 : 1);
            static_assert(QT_PASS >= 1 && QT_PASS * 16 * PT_PASS * 16 * OB <= WREG, "epilogue staging region");
            auto acc_of = [&](int pt, int qt) -> const v4i& { return acc[HALF < 0 ? pt / NPI : HALF][HALFQ < 0 ? qt / NQJ : HALFQ][pt % NPI][qt % NQJ]; };
            auto as_of = [&](int qt) { return reinterpret_cast<const float*>(smem + scale_off)[wq * WM + ROW0 + qt * 16 + dcol]; };
            auto bs_of = [&](int pt) { return *reinterpret_cast<const v4f*>(smem + scale_off + 1024 + (wp * WN + COL0 + pt * 16 + drow4) * 4); };
            uint8_t* y_blk = reinterpret_cast<uint8_t*>(y + (int64_t)(wm0 + ROW0) * epi.ldy + wn0 + COL0);
            const void* bias_blk = has_bias ? static_cast<const void*>(reinterpret_cast<const O*>(epi.bias) + ((epi.flags & EPI_BIAS_ROWS) ? wm0 + ROW0 : wn0 + COL0)) : nullptr;
            PQ_EPI_STAGED_DISPATCH(OUT, NPT, NQT, QT_PASS, PT_PASS, has_bias, epi.flags, acc_of, as_of, bs_of, bias_blk, smem, sw_off, y_blk, epi.ldy * OB, lane_e);
            stamp(3);
            return;
        }

        // direct path (edge tiles / unaligned y): guarded stores straight from registers
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) & (4 * OB - 1)) == 0) && ((epi.ldy & 3) == 0);
#pragma unroll
        for (int hQ = (HALFQ < 0 ? 0 : HALFQ); hQ < (HALFQ < 0 ? 2 : HALFQ + 1); ++hQ)
#pragma unroll
            for (int j = 0; j < NQJ; ++j) {
                const int m = wm0 + hQ * QW + j * SHAPE + dcol;
                const bool mok = m < M;
                float as = 1.0f;
                if constexpr (OUT != OUT_I32) as = mok ? epi.a_scale[m] : 0.0f;
#pragma unroll
                for (int hP = (HALF < 0 ? 0 : HALF); hP < (HALF < 0 ? 2 : HALF + 1); ++hP)
#pragma unroll
                    for (int i = 0; i < NPI; ++i)
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const int n = wn0 + hP * PWH + i * SHAPE + drow4 + 8 * g;
                            if (!mok || n >= N) continue;
                            const acc_t& c = acc[hP][hQ][i][j];
                            O* dst = y + (int64_t)m * epi.ldy + n;
                            if (n + 3 < N && vec_ok) {
                                O o[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float bs = 1.0f, bf = 0.0f;
                                    if constexpr (OUT != OUT_I32) {
                                        bs = epi.b_scale[n + r];
                                        if (has_bias) bf = load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : n + r);
                                    }
                                    o[r] = epi_convert<OUT>(c[g * 4 + r], as, bs, bf, has_bias, epi.flags & EPI_COL_FIRST);
                                }
                                if constexpr (OB == 2) *reinterpret_cast<v2u*>(dst) = *reinterpret_cast<const v2u*>(o);
                                else *reinterpret_cast<v4u*>(dst) = *reinterpret_cast<const v4u*>(o);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (n + r >= N) continue;
                                    float bs = 1.0f, bf = 0.0f;
                                    if constexpr (OUT != OUT_I32) {
                                        bs = epi.b_scale[n + r];
                                        if (has_bias) bf = load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : n + r);
                                    }
                                    dst[r] = epi_convert<OUT>(c[g * 4 + r], as, bs, bf, has_bias, epi.flags & EPI_COL_FIRST);
                                }
                            }
                        }
            }
    };
    constexpr std::integral_constant<int, -1> all{};
    constexpr std::integral_constant<int, 0> h0{};
    constexpr std::integral_constant<int, 1> h1{};
    if constexpr (FSK == 2) {
        if (kslice == 0) epilogue(h0, all);
        else epilogue(h1, all);
    } else if constexpr (FSK == 4) {
        if (kslice == 0) epilogue(h0, h0);
        else if (kslice == 1) epilogue(h0, h1);
        else if (kslice == 2) epilogue(h1, h0);
        else epilogue(h1, h1);
    } else if constexpr (half_epi) {
        epilogue(h0, all);
    } else {
        epilogue(all, all);
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_s8_p3_persist — the split-ring 256 x 256 tile for grids of MORE than one round (gate+up: 7 rounds, lm_head: 31): one
// workgroup per CU walks its output tiles (ids b, b + G, b + 2G, ...: the same XCD every time), so that what a fresh workgroup
// pays per round — launch, address set-up, the DMA issue burst and the first-touch latency of its first K-tile (weights come from
// HBM inside a model: 2.6 us of the 47 us a round takes, measured with the in-kernel stamps) — overlaps the previous tile's epilogue:
//   * behind the last K-tile of tile i (one barrier: every wave has read its last fragments) the first K-tile of tile i+1 is DMA'd
//     into the ring slots the last K-tile has just vacated — the epilogue stages through the slots of K-tile NT-2 and keeps the
//     scales in K-tile NT-3's weight slot, so all three P slots and both Q slots are spoken for and nothing collides;
//   * the epilogue of tile i runs (its 16 - 32 stores per wave are issued behind those 8 DMA pieces: a counted vmcnt leaves exactly
//     the stores in flight), one barrier, then K-tiles 1 and 2 are requested and the asm statement runs ALL K-tiles of tile i+1,
//     entered at the ring phase the previous tile left (kloop_p3_asm: `phase`).
// Every output tile is computed by the same instruction sequence as in gemm_s8_sp256<..., P3, ASMV = 1> apart from its first K-tile
// (here a regular tile of the statement instead of the HIP code's first-half form): integer sums, same epilogue -> same bits.
template <int OUT>
__global__ __launch_bounds__(512, 2) void gemm_s8_p3_persist(const int8_t* __restrict__ X, int64_t ldx, const int8_t* __restrict__ W, int64_t ldw,
                                                             EpiArgs epi, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int PHB = 128 * FBK, QHB = 128 * FBK, PB = 2 * PHB, QB = 2 * QHB, QBASE = 3 * PB;
    __shared__ __attribute__((aligned(16))) uint8_t smem[3 * PB + 2 * QB];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wp = w >> 2, wq = w & 3;
    const int NT = K / FBK;                                  // >= 4 (launcher)
    const int ntiles = tiles_m * tiles_n;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t sbw = smem_base + (uint32_t)w * 2048u, sbs = smem_base + (uint32_t)w * 1024u;
    constexpr int WM = 64, WN = 128;
    // Register budget: 128 accumulators + 96 fragment registers + the statement's 15 address registers leave ~15 VGPRs.  Everything that
    // depends on the lane is therefore recomputed where it is used, from a copy of the lane id that the compiler cannot see through —
    // hoisted out of the tile loop, those values stay live across the statement AND the epilogue and spill.
    auto lane_now = [&]() { int l = (int)threadIdx.x & 63; asm volatile("" : "+v"(l)); return l; };

    struct Src { uint32_t offP[2][2], offQ[2][2]; const int8_t *gP, *gQ; int m0, n0; };
    // coordinates of output tile `tix` and the per-lane source offsets of this wave's DMA pieces (as in gemm_s8_sp256)
    auto locate = [&](int tix, Src& o) {
        const int lane = lane_now();
        const int t = xcd_remap(tix, ntiles, epi.nxcd);
        constexpr int GM = 4;
        const int band = t / (GM * tiles_n);
        const int gm = (tiles_m - band * GM) < GM ? (tiles_m - band * GM) : GM;
        const int tin = t - band * GM * tiles_n;
        o.m0 = (band * GM + tin % gm) * 256; o.n0 = (tin / gm) * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int piece = w * 2 + jj, r = piece * 8 + (lane >> 3);
                const int src_chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
                int nl = (r / 64) * 128 + h * 64 + (r % 64);
                nl = (o.n0 + nl < N) ? nl : (N - 1 - o.n0);
                o.offP[h][jj] = (uint32_t)nl * (uint32_t)ldw + src_chunk * 16;
                int ml = (r / 32) * 64 + h * 32 + (r % 32);
                ml = (o.m0 + ml < M) ? ml : (M - 1 - o.m0);
                o.offQ[h][jj] = (uint32_t)ml * (uint32_t)ldx + src_chunk * 16;
            }
        o.gP = W + (int64_t)o.n0 * ldw; o.gQ = X + (int64_t)o.m0 * ldx;
    };
    // one K-tile side of this wave: 4 pieces into ring slot `slot`, from the tile's first K byte + koff
    auto dma_p = [&](const Src& o, int slot, int koff) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) glds16_sbase(o.gP + koff, o.offP[h][jj], sbw + slot * PB + h * PHB + jj * 1024);
    };
    auto dma_q = [&](const Src& o, int slot, int koff) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) glds16_sbase(o.gQ + koff, o.offQ[h][jj], sbw + QBASE + slot * QB + h * QHB + jj * 1024);
    };
    auto inc3 = [](int v, int d) { return (v + d) % 3; };

    using acc_t = v4i;
    acc_t acc[2][2][4][2];
    int sp = 0, sq = 0;                                      // ring slots of the current output tile's K-tile 0
    int tix = (int)blockIdx.x;
    {
        Src s0;
        locate(tix, s0);
        dma_p(s0, 0, 0); dma_q(s0, 0, 0);                    // first tile: its K-tile 0, waited for in full (once per workgroup)
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
    }
    for (;;) {
        int cm0, cn0;
        {
            // K-tile 0 sits in (sp, sq), visible to all.  Request K-tile 1 (weights first: the statement's issue order) and the weights of K-tile 2.
            Src cur;
            locate(tix, cur);
            cm0 = cur.m0; cn0 = cur.n0;
            dma_p(cur, inc3(sp, 1), FBK); dma_q(cur, sq ^ 1, FBK); dma_p(cur, inc3(sp, 2), 2 * FBK);
            const int8_t* cP = cur.gP + 3 * FBK;             // the statement's cursors: weights of K-tile 3, activations of K-tile 2
            const int8_t* cQ = cur.gQ + 2 * FBK;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) { acc[a][b][i][j] = v4i{0, 0, 0, 0}; asm volatile("" : "+v"(acc[a][b][i][j])); }
            const int lane = lane_now();
            const int frow = lane & 15, fchunk = lane >> 4, fkey = (frow >> 1) & 7;
            uint32_t bp[2], bph[2], bq[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c = ks * 4 + fchunk;
                bp[ks] = smem_base + (uint32_t)((wp * 64 + frow) * 128 + ((c ^ fkey) * 16));
                bph[ks] = bp[ks] + 2 * PB;
                bq[ks] = smem_base + QBASE + (uint32_t)((wq * 32 + frow) * 128 + ((c ^ fkey) * 16));
            }
            v4i fPa[4][2], fPb[4][2], fQa[2][2], fQb[2][2];
            {
                const uint32_t po = (uint32_t)(sp * PB), qo = (uint32_t)(sq * QB);
#pragma unroll
                for (int it = 0; it < 8; ++it) fPa[it % 4][it / 4] = *reinterpret_cast<const v4i*>(smem + (bp[it / 4] - smem_base) + po + (it % 4) * 2048);
#pragma unroll
                for (int it = 0; it < 4; ++it) fQa[it % 2][it / 2] = *reinterpret_cast<const v4i*>(smem + (bq[it / 2] - smem_base) + qo + (it % 2) * 2048);
#pragma unroll
                for (int it = 0; it < 8; ++it) fPb[it % 4][it / 4] = v4i{0, 0, 0, 0};      // (written by the statement before it reads them)
#pragma unroll
                for (int it = 0; it < 4; ++it) fQb[it % 2][it / 2] = v4i{0, 0, 0, 0};
            }
            // position of K-tile 0 in the statement's turn: position p holds (P slot (p + 1) % 3, Q slot (p + 1) % 2)  [CRT: p + 1 = 4 sp + 3 sq mod 6]
            const int phase = __builtin_amdgcn_readfirstlane((4 * sp + 3 * sq + 5) % 6);
            const bool scales_ok_ = (OUT == OUT_I32) ||
                ((((reinterpret_cast<uintptr_t>(epi.a_scale) | reinterpret_cast<uintptr_t>(epi.b_scale)) & 15) == 0) && M >= 4 && N >= 4);
            const int sbase = w == 0 ? cm0 : cn0, slim = w == 0 ? M : N;
            int e0 = sbase + lane * 4;
            e0 = e0 + 3 < slim ? e0 : (slim >= 4 ? slim - 4 : 0);
            const float* ssrc = (w == 0 ? epi.a_scale : epi.b_scale) + e0;
            const uint32_t do_scales = (uint32_t)__builtin_amdgcn_readfirstlane((int)(OUT != OUT_I32 && scales_ok_ && w < 2));
            cP -= phase * FBK; cQ -= phase * FBK;            // immediate-offset form: position `phase` of the first turn addresses this tile
            kloop_p3_asm<1>(acc, fPa, fPb, fQa, fQb, bp, bph, bq, cur.offP, cur.offQ, cP, cQ, (uint32_t)(NT - 3), sbw, (uint32_t)phase, ssrc, sbs, do_scales, (uint32_t)(w >> 2));
        }
        // ---- behind the last K-tile: its slots take the next output tile's first K-tile
        const int spl = inc3(sp, (NT - 1) % 3), sql = (sq + NT - 1) & 1;      // slots of K-tile NT-1
        const int nxt = tix + (int)gridDim.x;
        const bool more = nxt < ntiles;
        if (more) {
            __builtin_amdgcn_s_barrier();                    // every wave has read its last fragments
            Src nx;
            locate(nxt, nx);
            dma_p(nx, spl, 0); dma_q(nx, sql, 0);
        }
        // ---- K4 epilogue of the tile just finished (as in gemm_s8_sp256; ring slots relative to this tile's phase)
        using O = typename OutElem<OUT>::type;
        constexpr int OB = (int)sizeof(O);
        O* y = reinterpret_cast<O*>(epi.y);
        const bool has_bias = (OUT != OUT_I32) && epi.bias != nullptr;
        const bool scales_ok = (OUT == OUT_I32) ||
            ((((reinterpret_cast<uintptr_t>(epi.a_scale) | reinterpret_cast<uintptr_t>(epi.b_scale)) & 15) == 0) && M >= 4 && N >= 4);
        const int lane = lane_now();
        const int dcol = lane & 15, drow4 = (lane >> 4) * 4;
        const int wm0 = cm0 + wq * WM, wn0 = cn0 + wp * WN;
        const int scale_off = inc3(sp, NT % 3) * PB;                                    // K-tile NT-3's weight slot
        const bool staged = scales_ok && (wm0 + WM <= M) && (wn0 + WN <= N) && epi_rows_storable(epi, y, OB) &&
                            (!has_bias || (reinterpret_cast<uintptr_t>(epi.bias) & (4 * OB - 1)) == 0);
        constexpr int NSTORE = (OB == 2) ? 16 : 32;          // global stores per wave of the staged epilogue
        if (staged) {
            constexpr int WREG = 8192;
            const uint32_t sw_off = (uint32_t)((w < 4 ? inc3(sp, (NT + 1) % 3) * PB : QBASE + ((sq + NT) & 1) * QB) + (w & 3) * WREG);   // K-tile NT-2's slots
            constexpr int NPT = 8, NQT = 4;
            constexpr int PT_PASS = (NPT * 16 * OB > 256) ? NPT / 2 : NPT;
            constexpr int QT_PASS_MAX = WREG / (16 * PT_PASS * 16 * OB);
            constexpr int QT_PASS = QT_PASS_MAX >= NQT ? NQT : (QT_PASS_MAX >= 2 ? 2 : 1);
            static_assert((NQT / QT_PASS) * (NPT / PT_PASS) * (QT_PASS * 16 / (64 / (PT_PASS * 16 * OB / 16))) == NSTORE, "store count of the staged epilogue");
            auto acc_of = [&](int pt, int qt) -> const v4i& { return acc[pt / 4][qt / 2][pt % 4][qt % 2]; };
            auto as_of = [&](int qt) { return reinterpret_cast<const float*>(smem + scale_off)[wq * WM + qt * 16 + dcol]; };
            auto bs_of = [&](int pt) { return *reinterpret_cast<const v4f*>(smem + scale_off + 1024 + (wp * WN + pt * 16 + drow4) * 4); };
            uint8_t* y_blk = reinterpret_cast<uint8_t*>(y + (int64_t)wm0 * epi.ldy + wn0);
            const void* bias_blk = has_bias ? static_cast<const void*>(reinterpret_cast<const O*>(epi.bias) + ((epi.flags & EPI_BIAS_ROWS) ? wm0 : wn0)) : nullptr;
            PQ_EPI_STAGED_DISPATCH(OUT, NPT, NQT, QT_PASS, PT_PASS, has_bias, epi.flags, acc_of, as_of, bs_of, bias_blk, smem, sw_off, y_blk, epi.ldy * OB, lane);
        } else {
            const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) & (4 * OB - 1)) == 0) && ((epi.ldy & 3) == 0);
#pragma unroll
            for (int hQ = 0; hQ < 2; ++hQ)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int m = wm0 + hQ * 32 + j * 16 + dcol;
                    const bool mok = m < M;
                    float as = 1.0f;
                    if constexpr (OUT != OUT_I32) as = mok ? epi.a_scale[m] : 0.0f;
#pragma unroll
                    for (int hP = 0; hP < 2; ++hP)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int n = wn0 + hP * 64 + i * 16 + drow4;
                            if (!mok || n >= N) continue;
                            const acc_t& c = acc[hP][hQ][i][j];
                            O* dst = y + (int64_t)m * epi.ldy + n;
                            O o[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int nn = n + r < N ? n + r : N - 1;
                                float bs = 1.0f, bf = 0.0f;
                                if constexpr (OUT != OUT_I32) {
                                    bs = epi.b_scale[nn];
                                    if (has_bias) bf = load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : nn);
                                }
                                o[r] = epi_convert<OUT>(c[r], as, bs, bf, has_bias, epi.flags & EPI_COL_FIRST);
                            }
                            if (n + 3 < N && vec_ok) {
                                if constexpr (OB == 2) *reinterpret_cast<v2u*>(dst) = *reinterpret_cast<const v2u*>(o);
                                else *reinterpret_cast<v4u*>(dst) = *reinterpret_cast<const v4u*>(o);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) if (n + r < N) dst[r] = o[r];
                            }
                        }
                }
        }
        if (!more) break;
        // the next tile's first K-tile must have landed: its 8 pieces are OLDER than this epilogue's stores (vmcnt retires in order), so
        // the staged path leaves exactly its stores in flight; the direct path (edge tiles, compiler-counted stores and loads) drains
        if (staged) __builtin_amdgcn_s_waitcnt(waitcnt_imm(NSTORE, 0));
        else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();                        // visible to all; the staging slots and the scale slot are free again
        tix = nxt; sp = spl; sq = sql;
    }
}

template <int OUT>
void launch_gemm_p3_persist(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    const int nt = tiles_m * tiles_n;
    gemm_s8_p3_persist<OUT><<<dim3((unsigned)(nt < 256 ? nt : 256)), dim3(512), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n);
}

bool gemm_fast_eligible(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return M >= 1 && N >= 1 && K >= FBK && (K % FBK) == 0 && (lda % 16) == 0 && (ldb % 16) == 0 && al(A) && al(B) &&
           M < (1 << 30) && N < (1 << 30) && lda < (1 << 23) && ldb < (1 << 23);
}

unsigned long long* g_stamps = nullptr;   // dev builds only: set through pq_dev_set_stamp_buffer
void set_stamp_buffer(unsigned long long* p) { g_stamps = p; }
int gemm_debug_flags() { const char* e = getenv("PQ_GEMM_DBG"); return e ? atoi(e) : 0; }

                            // the 8-wave form (dev builds: 2 = 8 consumers + 4 loaders, 12 waves)
// multi-round grids of the 256 x 256 tile through gemm_s8_p3_persist: OFF by default — bit-identical, race-screened, and measured 0 .. 1.7 % SLOWER than one workgroup
// per tile (profiles/r03_ab_persistent.txt: the hardware dispatcher already overlaps a finished workgroup's store drain with its successor's prologue and
// balances the tiles dynamically; the persistent form can prefetch only ONE K-tile under the epilogue — the rings are full — and then waits for the second)

template <int OUT, int TM, int TN>
void launch_gemm_fast(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi,
                      int64_t M, int64_t N, int64_t K, hipStream_t st) {
    const int tiles_m = (int)((M + TM - 1) / TM), tiles_n = (int)((N + TN - 1) / TN);
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(512);
    if constexpr (TM == 128 && TN == 256) {
#ifdef PQ_ABLATION_BUILD   // 12-wave form (8 consumers): measured 1-3 % slower warm, +-1 % HBM-fed (profiles/r03_ab_lc12.txt) — the tile is ingest-bound, not issue-bound
        if (opt().sp128_lc == 2) {
            gemm_s8_sp256<OUT, 0, TM, TN, true, false, 0, 8><<<grid, dim3(768), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1);
            return;
        }
#endif
        if (opt().sp128_lc) {
            // K rotation of the loaders (`dbg` = K-tiles per chunk): for weight streams that matter (>= 6 MiB: profiles/r04_rotation.txt), chunks of 8 K-tiles
            const int ct = (opt().ring_rot && N * K >= (6 << 20)) ? (opt().ring_rot > 1 ? opt().ring_rot : rot_chunk_ktiles(tiles_m < 4 ? tiles_m : 4, 256)) : 0;
            gemm_s8_sp256<OUT, 0, TM, TN, true><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, ct, nullptr, 1);
            return;
        }
    }
#ifdef PQ_ABLATION_BUILD
    if constexpr (OUT == PQ_BF16 && TM == 256 && TN == 256) {
        if (opt().sp256_p3 && gemm_debug_flags() == 1024) {      // stamps only, around an asm K-loop variant
            switch (K >= 5 * FBK ? opt().sp256_asm : 0) {
#define PQ_ASMS(n) case n: gemm_s8_sp256<OUT, 1024, TM, TN, false, true, n><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 1024, g_stamps, 1); return;
                PQ_ASMS(1) PQ_ASMS(2) PQ_ASMS(3) PQ_ASMS(6) PQ_ASMS(7) PQ_ASMS(8) PQ_ASMS(9) PQ_ASMS(10) PQ_ASMS(11)
#undef PQ_ASMS
                default: break;
            }
        }
        if (opt().sp256_p3 && K >= 5 * FBK && opt().sp256_asm == 1) {      // round 4: the product asm loop + stamps with NO epilogue (1032) / HALF the epilogue (3072): timing only
            if (gemm_debug_flags() == 1032) { gemm_s8_sp256<OUT, 1032, TM, TN, false, true, 1><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 1032, g_stamps, 1); return; }
            if (gemm_debug_flags() == 3072) { gemm_s8_sp256<OUT, 3072, TM, TN, false, true, 1><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 3072, g_stamps, 1); return; }
        }
        if (opt().sp256_p3) {
            switch (gemm_debug_flags()) {
#define PQ_ABL3(n) case n: gemm_s8_sp256<OUT, n, TM, TN, false, true><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, n, g_stamps, 1); return;
                PQ_ABL3(1) PQ_ABL3(2) PQ_ABL3(3) PQ_ABL3(4) PQ_ABL3(8) PQ_ABL3(1024)
#undef PQ_ABL3
                default: break;
            }
        }
        switch (gemm_debug_flags()) {
#define PQ_ABL(n) case n: gemm_s8_sp256<OUT, n><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, n, g_stamps, 1); return;
            PQ_ABL(1) PQ_ABL(2) PQ_ABL(3) PQ_ABL(4) PQ_ABL(8) PQ_ABL(9) PQ_ABL(10) PQ_ABL(11) PQ_ABL(12) PQ_ABL(13) PQ_ABL(14) PQ_ABL(15) PQ_ABL(16) PQ_ABL(40) PQ_ABL(72) PQ_ABL(104) PQ_ABL(1024)
#undef PQ_ABL
            default: break;
        }
    }
#endif
    if constexpr (TM == 256 && TN == 256) {
        if (opt().sp256_p3) {
            // K-tiles 1 .. NT-1 in the hand-allocated asm statement (kloop_p3_asm.inc, variant 1) whenever there are at least five K-tiles;
            // PQ_SP256_ASM=0 keeps the HIP loop (same bits); dev builds: 2 / 3 / 6-9 are the A/B and timing-only variants (bf16 output only)
            const int av = K >= 5 * FBK ? opt().sp256_asm : 0;
            if (av == 1 && opt().sp256_persist && tiles_m * tiles_n > 256) {       // more than one round: one workgroup per CU walks its tiles
                launch_gemm_p3_persist<OUT>(A, lda, B, ldb, epi, M, N, K, st);
                return;
            }
            if (av == 1) {
                gemm_s8_sp256<OUT, 0, TM, TN, false, true, 1><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1);
                return;
            }
            if constexpr (OUT == PQ_BF16) {
#ifdef PQ_ABLATION_BUILD      // dev builds only: the A/B placement variant (2) and the timing-only variants, whose results are WRONG (3: no waits / barriers; 6-9: ablations)
                if (av == 2) { gemm_s8_sp256<OUT, 0, TM, TN, false, true, 2><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1); return; }
                if (av == 3) { gemm_s8_sp256<OUT, 0, TM, TN, false, true, 3><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1); return; }
                if (av == 6) { gemm_s8_sp256<OUT, 0, TM, TN, false, true, 6><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1); return; }
                if (av == 7) { gemm_s8_sp256<OUT, 0, TM, TN, false, true, 7><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1); return; }
                if (av == 8) { gemm_s8_sp256<OUT, 0, TM, TN, false, true, 8><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1); return; }
                if (av == 9) { gemm_s8_sp256<OUT, 0, TM, TN, false, true, 9><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1); return; }
#endif
            }
            gemm_s8_sp256<OUT, 0, TM, TN, false, true><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1);
            return;
        }
    }
    gemm_s8_sp256<OUT, 0, TM, TN><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, 1);
}

// ---- fused split-K (gemm_s8_sp256<..., FSK>): kslices workgroups per 256 x 256 tile, partial sums handed over inside the kernel
size_t fsk_workspace_bytes(int64_t M, int64_t N, int kslices) {
    const int64_t ntiles = ((M + 255) / 256) * ((N + 255) / 256);
    return fsk_counter_bytes((int)ntiles, kslices) + (size_t)ntiles * (size_t)(kslices - 1) * (size_t)(256 * 256 * 4);
}
// (returns false, WITHOUT launching, when the flags could not be zeroed: the kernel's waits would never end on flags that hold garbage)
// can the ticket form walk this stacked activation operand in place?  (slices cover whole slabs, or a slab holds whole slices; >= 4 K-tiles per slab; the ticket form)
bool fsk_kslabs_ok(int64_t K, int64_t k_per_slab, int kslices) {
    if (k_per_slab <= 0 || k_per_slab % FBK != 0 || K % k_per_slab != 0 || k_per_slab < 4 * FBK || kslices < 2 || K % ((int64_t)kslices * FBK) != 0) return false;
    if (opt().fsk_symmetric || opt().fsk_coop) return false;
    const int64_t nslabs = K / k_per_slab;
    return nslabs % kslices == 0 || kslices % nslabs == 0;
}
// the tickets and ready counts / flags of a launch are zeroed by a KERNEL of our own, not by hipMemsetAsync (round 6): a hipGraph that holds [memset node, fused split-K
// kernel] hung on its next replay as soon as ANY other graph without such a node had been captured after it (round-5 and round-6 libraries alike, ROCm 7.0 / torch 2.10:
// profiles/r06_hipgraph_memset_hang.txt) — the kernel's last arriver polls a `ready` count that the memset node did not (or not in order) reset.  A kernel node has no such problem.
__global__ void fsk_zero_counters(uint32_t* p, int n) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i < n) p[i] = 0u;
}
template <int OUT>
bool launch_gemm_fsk(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi, int64_t M, int64_t N, int64_t K,
                     int kslices, void* workspace, hipStream_t st, int64_t a_slab_stride, int64_t a_k_per_slab) {
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    {
        const int nwords = (int)(fsk_counter_bytes(tiles_m * tiles_n, kslices) / 4);
        fsk_zero_counters<<<dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st>>>(static_cast<uint32_t*>(workspace), nwords);   // tickets and ready counts / flags
        if (hipGetLastError() != hipSuccess) return false;
    }
    const dim3 grid((unsigned)(tiles_m * tiles_n * kslices)), block(512);
    unsigned long long* const ws = static_cast<unsigned long long*>(workspace);
    if (a_k_per_slab > 0 && a_k_per_slab < K) {      // stacked activation codes walked in place (the caller checked fsk_kslabs_ok)
        gemm_s8_sp256<OUT, 0, 256, 256, false, true, 1, 4, 1, true><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, opt().fsk_fenced ? 1 : 0, ws, kslices,
                                                                                     a_slab_stride, (int)(a_k_per_slab / FBK));
        return true;
    }
    // the SYMMETRIC exchanges WAIT for partner workgroups: they need every workgroup of the launch resident at once.  A COOPERATIVE launch (round 5) is the runtime's own
    // guarantee of exactly that — hipLaunchCooperativeKernel either places the whole grid together (and serialises cooperative launches of the device against each other) or
    // returns an error — so with PQ_FSK_COOP=1 the symmetric kernels go out that way, and any error (grid too large for the CUs this queue may use, a capture
    // that does not take cooperative nodes, an old runtime) falls through to the ticket form, which never waits for a workgroup that may not be running.
    // MEASURED (profiles/r05_ab_fsk_coop.txt): it works, eagerly and under hipGraph capture, bit-identical — and costs 21-24 us per launch (the runtime's cooperative
    // queue hand-off): 2048 x 4096 x 11008 97.9 us against 75.9 ticket / 74.0 symmetric, 4096 x 1024 x 28672 119.0 against 99.5 / 94.0.  The safe route to the
    // symmetric exchange costs ten times what the exchange saves: OFF by default.  PQ_FSK_SYMMETRIC=1 remains the caller's-promise route (plain launch).
    if ((kslices == 2 || kslices == 4) && opt().fsk_coop && !opt().fsk_symmetric) {
        int M_ = (int)M, N_ = (int)N, K_ = (int)K, tm_ = tiles_m, tn_ = tiles_n, zero = 0, ks_ = kslices;
        int64_t lda_ = lda, ldb_ = ldb;
        EpiArgs epi_ = epi;
        unsigned long long* ws_ = ws;
        int64_t no_stride = 0;
        int no_tps = 0;          // (every parameter of the kernel: the trailing K-slab pair too — round 6)
        void* args[] = {(void*)&A, (void*)&lda_, (void*)&B, (void*)&ldb_, (void*)&epi_, (void*)&M_, (void*)&N_, (void*)&K_, (void*)&tm_, (void*)&tn_, (void*)&zero, (void*)&ws_, (void*)&ks_,
                        (void*)&no_stride, (void*)&no_tps};
        const void* fn = kslices == 2 ? reinterpret_cast<const void*>(&gemm_s8_sp256<OUT, 0, 256, 256, false, true, 1, 4, 2>)
                                      : reinterpret_cast<const void*>(&gemm_s8_sp256<OUT, 0, 256, 256, false, true, 1, 4, 4>);
        if (hipLaunchCooperativeKernel(fn, grid, block, args, 0, st) == hipSuccess) return true;
        (void)hipGetLastError();      // not placed: the ticket form below
    }
    const bool sym = opt().fsk_symmetric;
    if (kslices == 2 && sym)
        gemm_s8_sp256<OUT, 0, 256, 256, false, true, 1, 4, 2><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, ws, 2);
    else if (kslices == 4 && sym)
        gemm_s8_sp256<OUT, 0, 256, 256, false, true, 1, 4, 4><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, ws, 4);
    else
        gemm_s8_sp256<OUT, 0, 256, 256, false, true, 1, 4, 1><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, opt().fsk_fenced ? 1 : 0, ws, kslices);
    return true;
}
template bool launch_gemm_fsk<PQ_BF16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, int, void*, hipStream_t, int64_t, int64_t);
template bool launch_gemm_fsk<PQ_FP16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, int, void*, hipStream_t, int64_t, int64_t);
template bool launch_gemm_fsk<PQ_F32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, int, void*, hipStream_t, int64_t, int64_t);

// ---- split-K: S K-slices of the int32 GEMM into S slabs of `slabs` (each [M, N], ld = N), then one pass that sums
// the slabs (exact) and applies QSPEC E1-E4.  Doubles/quadruples the busy CUs for small-MN / long-K problems.
template <int TM>
void launch_gemm_splitk_i32(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, int32_t* slabs,
                            int64_t M, int64_t N, int64_t K, int kslices, hipStream_t st, int nxcd) {
    const int tiles_m = (int)((M + TM - 1) / TM), tiles_n = (int)((N + FT - 1) / FT);
    const dim3 grid((unsigned)(tiles_m * tiles_n * kslices)), block(512);
    EpiArgs epi{nullptr, nullptr, nullptr, slabs, N, 0};
    epi.nxcd = nxcd;                  // (the caller's: the XCD count of the device the launch goes to — ADVICE r5)
    if constexpr (TM == 256) {      // slices of at least five K-tiles: the split-ring tile with the asm K-loop
        if (opt().sp256_p3 && opt().sp256_asm == 1 && K / kslices >= 5 * FBK) {
            gemm_s8_sp256<OUT_I32, 0, 256, 256, false, true, 1><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, kslices);
            return;
        }
    }
    gemm_s8_sp256<OUT_I32, 0, TM><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 0, nullptr, kslices);
}
template void launch_gemm_splitk_i32<256>(const int8_t*, int64_t, const int8_t*, int64_t, int32_t*, int64_t, int64_t, int64_t, int, hipStream_t, int);
template void launch_gemm_splitk_i32<128>(const int8_t*, int64_t, const int8_t*, int64_t, int32_t*, int64_t, int64_t, int64_t, int, hipStream_t, int);

template <int OUT>
__global__ __launch_bounds__(256) void splitk_reduce_epilogue(const int32_t* __restrict__ slabs, int kslices, int64_t M, int64_t N,
                                                              EpiArgs epi) {
    using O = typename OutElem<OUT>::type;
    const int64_t nvec = (N + 3) / 4;                                  // 4 consecutive n per thread
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nvec) return;
    const int64_t m = i / nvec, n = (i % nvec) * 4;
    const bool full = (n + 3 < N) && ((N & 3) == 0);
    int acc[4] = {0, 0, 0, 0};
    for (int s = 0; s < kslices; ++s) {
        const int32_t* p = slabs + ((int64_t)s * M + m) * N + n;
        if (full) { const v4i v = *reinterpret_cast<const v4i*>(p); acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
        else { for (int r = 0; r < 4; ++r) if (n + r < N) acc[r] += p[r]; }
    }
    const float as = epi.a_scale[m];
    const bool has_bias = epi.bias != nullptr;
    O* dst = reinterpret_cast<O*>(epi.y) + m * epi.ldy + n;
    O o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (n + r < N) {
            const float bf = has_bias ? load_bias<OUT>(epi.bias, (epi.flags & EPI_BIAS_ROWS) ? m : n + r) : 0.0f;
            o[r] = epi_convert<OUT>(acc[r], as, epi.b_scale[n + r], bf, has_bias, epi.flags & EPI_COL_FIRST);
        } else o[r] = O{};
    }
    const bool vec_st = full && ((reinterpret_cast<uintptr_t>(dst) & (4 * sizeof(O) - 1)) == 0);
    if (vec_st) {
        if constexpr (sizeof(O) == 2) store_wt_b64(dst, *reinterpret_cast<const v2u*>(o));
        else store_wt_b128(dst, *reinterpret_cast<const v4u*>(o));
    } else {
        for (int r = 0; r < 4; ++r) if (n + r < N) dst[r] = o[r];
    }
}
template <int OUT>
void launch_splitk_reduce(const int32_t* slabs, int kslices, int64_t M, int64_t N, const EpiArgs& epi, hipStream_t st) {
    const int64_t work = M * ((N + 3) / 4);
    splitk_reduce_epilogue<OUT><<<dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st>>>(slabs, kslices, M, N, epi);
}
template void launch_splitk_reduce<PQ_BF16>(const int32_t*, int, int64_t, int64_t, const EpiArgs&, hipStream_t);
template void launch_splitk_reduce<PQ_FP16>(const int32_t*, int, int64_t, int64_t, const EpiArgs&, hipStream_t);
template void launch_splitk_reduce<PQ_F32>(const int32_t*, int, int64_t, int64_t, const EpiArgs&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// gemm_s8_ring128 — 128 x 128 output tile, 4 waves (one per SIMD), wave tile 64(n) x 64(m): the variant for grids that fill
// at most half the chip with the big tiles (1024-wide shards, M <= 512).  A tile this small needs 64 B/clk/CU of operand
// ingest to keep its MFMAs fed; at ~2000 cycles of L2 latency that is 128 KiB in flight per CU, so the design is a 4-deep ring of 32-KiB K-tiles
// (three in flight while one is read).  What it reaches is set by the L2 -> LDS path and the board's power (round 4, profiles/r04_ablate_ring.txt:
// the DMA stream alone 82 GB/s per CU = 21 TB/s over the chip; the product loop 707 cycles per K-tile at 1.56 GHz; barriers 4.5 %, a fifth slot -4 %).
// Each wave keeps the fragments of the current K-tile in registers and reads ALL fragments of the next one (16 x
// ds_read_b128) in the shadows of the current tile's 32 MFMAs; the DMA pieces of tile kt+4 follow in the next shadows.
// One s_barrier per K-tile.  Same LDS image as the big kernel: [128 rows][128 B], 16-byte chunk c of row r at c ^ ((r>>1)&7).
constexpr int R_TILE = 128, R_NBUF = 4, R_OPER = 128 * FBK /* 16 KiB */, R_BUF = 2 * R_OPER, R_LDS = R_NBUF * R_BUF;

// LC (loader / consumer split, 8 waves): measured with in-kernel stamps, the 4-wave form spends ~1080 cycles per K-tile against
// 512 of MFMA because every LDS-DMA piece holds the issuing wave for 60-180 cycles (the texture path takes ~22 cycles per 1-KiB
// piece per CU and the wave issues in order) — with ONE wave per SIMD the matrix pipe idles through each of them.  LC puts a
// second wave on every SIMD whose only job is the DMA stream: waves 0-3 (consumers) run the pure MFMA + ds_read stream, waves
// 4-7 (loaders) issue all 32 pieces of a K-tile (8 each) and wait for them; one s_barrier per K-tile joins the two roles
// exactly where the 4-wave form has its barrier, so the ring protocol (and every result bit) is unchanged.
template <int OUT, bool LC = false, int ABL = 0>   // ABL (dev builds, timing only): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMAs, 8 no barriers in the loop, 16 stamps around the consumers' K-loop
__global__ __launch_bounds__(LC ? 512 : 256, LC ? 2 : 1) void gemm_s8_ring128(const int8_t* __restrict__ X, int64_t ldx, const int8_t* __restrict__ W,
                                                       int64_t ldw, EpiArgs epi, int M, int N, int K, int tiles_m, int tiles_n, int ct, int rot_div,
                                                       unsigned long long* stamps, KSlabs xs = KSlabs{}) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[R_LDS];
    auto stamp = [&](int point) {     // dev builds (ABL & 16): consumer waves stamp {100 MHz counter, shader cycles} around their K-loop
        if constexpr ((ABL & 16) != 0) {
            if (stamps != nullptr) {
                const unsigned long long r = __builtin_amdgcn_s_memrealtime(), c = __builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) {
                    unsigned long long* d = stamps + (((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + point) * 2;
                    d[0] = r; d[1] = c;
                }
            }
        }
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LC && wave >= 4;                                 // (4-wave form: every wave loads and computes)
    const int w = wave & 3;                                            // index inside the role
    const int wp = w >> 1, wq = w & 1;

    int t = xcd_remap((int)blockIdx.x, tiles_m * tiles_n, epi.nxcd);
    constexpr int GM = 8;
    const int band = t / (GM * tiles_n);
    const int gm = (tiles_m - band * GM) < GM ? (tiles_m - band * GM) : GM;
    const int tin = t - band * GM * tiles_n;
    const int tm = band * GM + tin % gm, tn = tin / gm;
    const int m0 = tm * R_TILE, n0 = tn * R_TILE;

    // ---- staging: operand tile = 16 pieces of 8 rows x 128 B; wave w issues pieces w*4 .. w*4+3 of P and of Q
    uint32_t offP[4], offQ[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int piece = w * 4 + jj;
        const int r = piece * 8 + (lane >> 3);
        const int src_chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        const int nl = (n0 + r < N) ? r : (N - 1 - n0);       // clamp: rows past the edge re-read a valid row
        const int ml = (m0 + r < M) ? r : (M - 1 - m0);
        offP[jj] = (uint32_t)nl * (uint32_t)ldw + src_chunk * 16;
        offQ[jj] = (uint32_t)ml * (uint32_t)ldx + src_chunk * 16;
    }
    // K walk (round 4, LC form): the up-to-8 workgroups of a band that stream one weight panel (neighbours on one XCD, started together) walk each chunk of
    // `ct` K-tiles from different starting points and wrap inside the chunk (gemm_s8_ring.hip: "K ROTATION in chunks") — in lockstep the panel's unique bytes in
    // flight are one workgroup's ring, which is what an HBM-fed launch of these tiles lacks.  rot_div = 0 (and the 4-wave form): the plain walk.  Same bits.
    const int NTw = K / FBK;
    auto rot_of = [&](int len) { return rot_div > 0 ? (int)(((int64_t)((tin % gm) % rot_div) * len) / rot_div) : 0; };
    int cbase = 0, clen = (LC && ct < NTw) ? ct : NTw;
    int cpos = LC ? rot_of(clen) : 0, cleft = clen;
    const int8_t* const gP0 = W + (int64_t)n0 * ldw;
    const int8_t* const gQ0 = X + (int64_t)m0 * ldx;
    // (the walk's offset goes through readfirstlane: the rotation's division is vector code, and the DMA's base operand must be provably wave-uniform)
    // K-SLAB form of the X operand (pq_qlinear_s8_kslabs; LC form only): the activation codes arrive STACKED as an all-gather leaves them, [G][M][K / G] — K-tile kt lives in
    // slab kt / xs.tiles at xs.stride bytes per slab, ldx = the slab's row length.  An integer sum has no order, so walking the slabs in place is the row-major GEMM's bits
    // without the layout pass.  kt / tiles by a host-computed reciprocal (exact for kt < 2^16), all scalar.
    auto x_off = [&](int ktu) -> int64_t {
        if (xs.tiles <= 0) return (int64_t)ktu * FBK;
        if (xs.tiles == 1) return (int64_t)ktu * xs.stride;          // one K-tile per slab (its reciprocal, 2^32, does not fit the 32-bit magic)
        const int sl = (int)(((uint64_t)(uint32_t)ktu * (uint64_t)xs.magic) >> 32);
        return (int64_t)sl * xs.stride + (int64_t)(ktu - sl * xs.tiles) * FBK;
    };
    const int kt0u = __builtin_amdgcn_readfirstlane(cpos);
    const int8_t* gP = gP0 + kt0u * FBK;
    const int8_t* gQ = gQ0 + x_off(kt0u);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lptr_t)smem;
    auto dma_item = [&](int buf, auto gc) {      // piece g of the next K-tile: 0..3 P, 4..7 Q; the 8th moves the K walk on
        constexpr int g = decltype(gc)::value;
        const uint32_t la = smem_base + buf * R_BUF + (g >= 4 ? R_OPER : 0) + (w * 4 + (g & 3)) * 1024;
        if constexpr (g < 4) glds16_sbase(gP, offP[g], la);
        else glds16_sbase(gQ, offQ[g - 4], la);
        if constexpr (g == 7) {
            if constexpr (!LC) { gP += FBK; gQ += FBK; }
            else {
                if (++cpos == clen) cpos = 0;
                if (--cleft == 0) {
                    cbase += clen;
                    clen = NTw - cbase < ct ? NTw - cbase : ct;
                    cleft = clen;
                    cpos = clen > 0 ? rot_of(clen) : 0;
                }
                const int ktu = __builtin_amdgcn_readfirstlane(cbase + cpos);
                gP = gP0 + ktu * FBK;
                gQ = gQ0 + x_off(ktu);
            }
        }
    };

    // ---- fragments: P tile i = rows wp*64 + i*16 .. +15, Q tile j = rows wq*64 + j*16 .. +15; k-step ks = 64 bytes
    const int frow = lane & 15, fchunk = lane >> 4, fkey = (frow >> 1) & 7;
    uint32_t lP[2], lQ[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fchunk;
        lP[ks] = (uint32_t)((wp * 64 + frow) * 128 + ((c ^ fkey) * 16));
        lQ[ks] = (uint32_t)((wq * 64 + frow) * 128 + ((c ^ fkey) * 16)) + R_OPER;
    }
    v4i fa[16], fb[16];              // item it: it < 8: P tile it & 3, ks = it >> 2; it >= 8: Q tile it & 3, ks = (it >> 2) & 1
    auto read_item = [&](int bufoff, v4i (&f)[16], auto ic) {
        constexpr int it = decltype(ic)::value, ti = it & 3, ks = (it >> 2) & 1;
        f[it] = *reinterpret_cast<const v4i*>(smem + bufoff + (it < 8 ? lP[ks] : lQ[ks]) + ti * 16 * 128);
    };
    v4i acc[4][4];                       // zeroed behind the prologue's DMA issue

    const int NT = K / FBK;
    // ---- prologue: up to 4 tiles in flight, wait for tile 0, read its fragments
    if (!LC || loader) {
#pragma unroll
        for (int b = 0; b < R_NBUF; ++b)
            if (b < NT) static_for<8>([&](auto gc) { dma_item(b, gc); });
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    __builtin_amdgcn_sched_barrier(0);
    if (!LC || loader) {
        if (NT >= 4) __builtin_amdgcn_s_waitcnt(0x4078);        // vmcnt(24): 3 tiles may still be in flight
        else if (NT == 3) __builtin_amdgcn_s_waitcnt(0x4070);   // vmcnt(16)
        else if (NT == 2) __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8)
        else __builtin_amdgcn_s_waitcnt(0x0070);
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (LC) {
        if (loader) {
            // the loader's K-loop: the same waits and the same barrier as the consumers' tiles, then the 8 pieces of tile kt+4
            for (int kt = 0; kt < NT; ++kt) {
                const int rem = NT - 1 - kt;
                if (rem > 0) {
                    if (rem >= 3) __builtin_amdgcn_s_waitcnt(waitcnt_imm(16, 15));
                    else if (rem == 2) __builtin_amdgcn_s_waitcnt(waitcnt_imm(8, 15));
                    else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
                    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
                }
                if constexpr (!(ABL & 1)) if (rem >= R_NBUF) static_for<8>([&](auto gc) { dma_item(kt & 3, gc); });
            }
            return;
        }
    }
    static_for<16>([&](auto ic) { read_item(0, fa, ic); });
    stamp(0);

    // one K-tile: 32 MFMAs on `cur`; in their shadows the 16 fragment reads of tile kt+1 into `nxt`, then the 8 DMA pieces
    // of tile kt+4 into the ring slot tile kt just vacated (its fragments are in registers; every wave passed the barrier
    // after its own lgkmcnt(0)).
    auto tile = [&](int kt, v4i (&cur)[16], v4i (&nxt)[16]) {
        const int rem = NT - 1 - kt;                          // tiles after this one
        if (rem > 0) {
            // tile kt+1 must have landed: tiles kt+2, kt+3 (8 pieces each per wave) may stay in flight
            if constexpr (LC) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));   // consumer: its fragment reads only (the loaders wait for the DMA)
            else if (rem >= 3) __builtin_amdgcn_s_waitcnt(0x4070);      // vmcnt(16) lgkmcnt(0)
            else if (rem == 2) __builtin_amdgcn_s_waitcnt(0x0078); // vmcnt(8)  lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0)  lgkmcnt(0)
            if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        }
        const int nbuf = ((kt + 1) & 3) * R_BUF;
        const bool more = rem >= R_NBUF;                      // tile kt+4 exists
        __builtin_amdgcn_sched_barrier(0);
        static_for<32>([&](auto xc) {
            constexpr int x = decltype(xc)::value, ks = x >> 4, i = (x >> 2) & 3, j = x & 3;
            // accumulators pinned in AGPRs through the asm form: with one wave per SIMD hipcc otherwise splits them between
            // the two register files and pays 4 v_accvgpr_write + s_nop in front of every other MFMA (measured 44 % -> see DESIGN)
            // (LC: two waves per SIMD share a 256-register budget, which hipcc halves as soon as a kernel names AGPRs — the builtin
            // keeps all of it as arch VGPRs, as in the big kernel)
            if constexpr (ABL & 4) { if constexpr (x == 0) asm volatile("" : "+v"(cur[0]), "+v"(cur[8])); }
            else if constexpr (LC) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur[ks * 4 + i], cur[8 + ks * 4 + j], acc[i][j], 0, 0, 0);
            else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(cur[ks * 4 + i]), "v"(cur[8 + ks * 4 + j]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (x < 16 && !(ABL & 2)) read_item(nbuf, nxt, xc);      // (last tile: reads a stale slot, values unused)
            else if constexpr (x < 24 && !LC) { if (more) dma_item(kt & 3, std::integral_constant<int, x - 16>{}); }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int kt = 0;
    for (; kt + 1 < NT; kt += 2) { tile(kt, fa, fb); tile(kt + 1, fb, fa); }
    if (kt < NT) tile(kt, fa, fb);
    stamp(1);

    // the asm MFMAs are invisible to hipcc's hazard tracking: drain the pipe, then pass every accumulator through an empty
    // asm so that no v_accvgpr_read can be scheduled above the drain
    if constexpr (!LC) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
    }

    // ---- epilogue: D[row <-> n][col <-> m]; lane holds 4 consecutive n of one m per accumulator
    using O = typename OutElem<OUT>::type;
    constexpr int OB = (int)sizeof(O);
    O* y = reinterpret_cast<O*>(epi.y);
    const bool has_bias = (OUT != OUT_I32) && epi.bias != nullptr;
    const int dcol = lane & 15, drow4 = (lane >> 4) * 4;
    const int wm0 = m0 + wq * 64, wn0 = n0 + wp * 64;
    const bool staged = (wm0 + 64 <= M) && (wn0 + 64 <= N) && epi_rows_storable(epi, y, OB) &&
                        (OUT == OUT_I32 || (reinterpret_cast<uintptr_t>(epi.b_scale) & 15) == 0) &&
                        (!has_bias || (reinterpret_cast<uintptr_t>(epi.bias) & (4 * OB - 1)) == 0);
    if (staged) {
        // staging: this wave's quarter (8 KiB) of the ring slot after the last K-tile's.  The last barrier of the loop is
        // at the start of tile NT-2; after it only slot (NT-1) & 3 is still read for real (the last tile's prefetch of slot
        // NT & 3 reads values nobody uses), and no DMA is in flight: no barrier needed.
        const uint32_t sw_off = (uint32_t)((NT & 3) * R_BUF + w * (R_BUF / 4));
        constexpr int QT_PASS = (OB == 2) ? 4 : 2;                          // 64 rows x 128 B, or 32 rows x 256 B
        auto acc_of = [&](int pt, int qt) -> const v4i& { return acc[pt][qt]; };
        auto as_of = [&](int qt) { return epi.a_scale[wm0 + qt * 16 + dcol]; };
        auto bs_of = [&](int pt) { return *reinterpret_cast<const v4f*>(epi.b_scale + wn0 + pt * 16 + drow4); };
        uint8_t* y_blk = reinterpret_cast<uint8_t*>(y + (int64_t)wm0 * epi.ldy + wn0);
        const void* bias_blk = has_bias ? static_cast<const void*>(reinterpret_cast<const O*>(epi.bias) + ((epi.flags & EPI_BIAS_ROWS) ? wm0 : wn0)) : nullptr;
        PQ_EPI_STAGED_DISPATCH(OUT, 4, 4, QT_PASS, 4, has_bias, epi.flags, acc_of, as_of, bs_of, bias_blk, smem, sw_off, y_blk, epi.ldy * OB, lane);
        return;
    }
    // direct path (edge tiles / unaligned y): guarded stores from registers, 4 consecutive n at a time when aligned
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) & (4 * OB - 1)) == 0) && ((epi.ldy & 3) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = wm0 + j * 16 + dcol;
        if (m >= M) continue;
        float as = 1.0f;
        if constexpr (OUT != OUT_I32) as = epi.a_scale[m];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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


template <int OUT>
void launch_gemm_ring128(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi, int64_t M, int64_t N,
                         int64_t K, hipStream_t st, int64_t a_slab_stride, int64_t a_k_per_slab) {
    KSlabs xs{};
    if (a_k_per_slab > 0) {          // stacked activation operand (the caller checked: LC form, k_per_slab % 128 == 0, K / 128 < 2^16)
        xs.tiles = (int)(a_k_per_slab / FBK);
        xs.magic = (uint32_t)(((1ull << 32) + (uint64_t)xs.tiles - 1) / (uint64_t)xs.tiles);
        xs.stride = a_slab_stride;
    }
    const int tiles_m = (int)((M + R_TILE - 1) / R_TILE), tiles_n = (int)((N + R_TILE - 1) / R_TILE);
    // K rotation (LC form), chunks of rot_chunk_ktiles() K-tiles: HBM-fed launches lose their lockstep penalty — 4096 x 1024 x 8192 37.5 -> 33.9 us, 2048 x 1024 x 8192
    // 34.4 -> 26.8, 4096 x 1024 x 28672 127.6 -> 102.4 — for 4 - 5 % on cache-warm replays; PQ_RING_ROT=0 switches it off (A/B: profiles/r04_rotation.txt)
    const int gmx = tiles_m < 8 ? tiles_m : 8;
    const int rot_div = (opt().ring_rot && N * K >= (6 << 20)) ? gmx : 0;      // weight streams that matter: below ~6 MiB the rotation only costs (4096 x 512 x 8192: -5 %)
    const int ct = opt().ring_rot > 1 ? opt().ring_rot : rot_chunk_ktiles(gmx, 128);
#ifdef PQ_ABLATION_BUILD
    if constexpr (OUT == PQ_BF16) {
        switch (gemm_debug_flags()) {
#define PQ_RABL(n) case n: gemm_s8_ring128<OUT, true, n><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, ct, rot_div, g_stamps); return;
            PQ_RABL(1) PQ_RABL(2) PQ_RABL(3) PQ_RABL(4) PQ_RABL(5) PQ_RABL(6) PQ_RABL(7) PQ_RABL(8) PQ_RABL(9) PQ_RABL(10) PQ_RABL(11) PQ_RABL(14)
            PQ_RABL(16) PQ_RABL(17) PQ_RABL(18) PQ_RABL(19) PQ_RABL(20) PQ_RABL(22) PQ_RABL(24)
#undef PQ_RABL
            default: break;
        }
    }
#endif
    if (opt().ring_lc || xs.tiles > 0) gemm_s8_ring128<OUT, true><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, ct, rot_div, nullptr, xs);
    else gemm_s8_ring128<OUT, false><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n, 1 << 30, 0, nullptr);
}
template void launch_gemm_ring128<PQ_BF16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);
template void launch_gemm_ring128<PQ_FP16>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);
template void launch_gemm_ring128<PQ_F32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);
template void launch_gemm_ring128<OUT_I32>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t, int64_t);

#define PQ_INST(OUT, TM, TN) \
    template void launch_gemm_fast<OUT, TM, TN>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
PQ_INST(PQ_BF16, 256, 256) PQ_INST(PQ_FP16, 256, 256) PQ_INST(PQ_F32, 256, 256) PQ_INST(OUT_I32, 256, 256)
PQ_INST(PQ_BF16, 128, 256) PQ_INST(PQ_FP16, 128, 256) PQ_INST(PQ_F32, 128, 256) PQ_INST(OUT_I32, 128, 256)
PQ_INST(PQ_BF16, 128, 128) PQ_INST(PQ_FP16, 128, 128) PQ_INST(PQ_F32, 128, 128) PQ_INST(OUT_I32, 128, 128)
#undef PQ_INST

}  // namespace pq
