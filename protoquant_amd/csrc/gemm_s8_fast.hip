// gemm_s8_fast.hip — K3/K4 hot variant for gfx950: s8 x s8 -> s32 on v_mfma_i32_16x16x64_i8 /
// v_mfma_i32_32x32x32_i8 with the fused row-scale x col-scale dequant epilogue.
//
// Structure (designed for CDNA4):
//   * 256 x 256 output tile per workgroup, K step = 128 bytes, 512 threads = 8 waves = 2 per SIMD.
//   * MFMA roles: first operand P = weight rows (n), second operand Q = activation rows (m); a lane's
//     accumulator registers are then consecutive n of ONE output row m -> contiguous y stores.
//   * wave (wp, wq) = (w>>2, w&3) owns n-range wp*128+[0,128) x m-range wq*64+[0,64), split in
//     halves hP (64 n) x hQ (32 m): four quadrants of 16 (16x16x64) or 8 (32x32x32) MFMAs.
//   * LDS: 2 buffers x {P half0, P half1, Q half0, Q half1} x 16 KiB = 128 KiB, one __shared__
//     array.  A half-tile is [128 rows][128 B]; 16-byte chunk c of row r sits at chunk c ^ ((r>>1)&7)
//     (conflict-free ds_read_b128 for both MFMA shapes).  Staging is global_load_lds_dwordx4: the LDS
//     image is lane-linear, the XOR goes on the per-lane SOURCE address.
//   * Ping-pong: waves 0-3 and 4-7 (SIMD partners) run one barrier apart, so on every SIMD one wave
//     issues MFMAs while its partner issues LDS reads + the next half-tile's DMA.
//   * One K-tile = 4 phases; phase p stages one half-tile of tile t+1, so every DMA has >= 2 phases
//     (~1000 cycles) to land.  Waits are counted (vmcnt(4)), never 0 in the loop; barriers are raw
//     s_barrier.  RAW: a half-tile is read one phase after the vmcnt that retires it (+ barrier).
//     WAR: a slot is restaged >= 2 phases after its last ds_read.
#include <type_traits>

#include "gemm_epilogue.h"

namespace pq {

constexpr int FT = 256;          // tile edge (both m and n)
constexpr int FBK = 128;         // K bytes per tile step
constexpr int HALF_BYTES = 128 * FBK;        // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;    // 64 KiB
constexpr int LDS_BYTES = 2 * BUF_BYTES;     // 128 KiB

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ __forceinline__ void glds16(const int8_t* g, uint8_t* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

#define PQ_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// XCD-aware bijective remap: blocks that share an XCD (equal bid % 8) get a contiguous run of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int OUT, int SHAPE>   // SHAPE: 16 -> 16x16x64, 32 -> 32x32x32
__global__ __launch_bounds__(512, 2) void gemm_s8_pp256(const int8_t* __restrict__ X, int64_t ldx,
                                                        const int8_t* __restrict__ W, int64_t ldw, EpiArgs epi,
                                                        int M, int N, int K, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w >> 2, wq = w & 3;

    // ---- tile assignment: XCD remap, then grouped order (GM m-tiles per band) for L2 reuse
    int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    constexpr int GM = 4;
    const int band = t / (GM * tiles_n);
    const int gm = (tiles_m - band * GM) < GM ? (tiles_m - band * GM) : GM;
    const int tin = t - band * GM * tiles_n;
    const int tm = band * GM + tin % gm, tn = tin / gm;
    const int m0 = tm * FT, n0 = tn * FT;

    // ---- staging source offsets: wave w issues pieces (w*2+jj), jj = 0,1, of every half-tile.
    // LDS row r = w*16 + jj*8 + (lane>>3) of the half-tile; physical chunk lane&7.
    // P half h, LDS row r <-> n_local = (r>>6)*128 + h*64 + (r&63);  Q: m_local = (r>>5)*64 + h*32 + (r&31).
    uint32_t offP[2][2], offQ[2][2];   // [half][jj] byte offset from the tile's first row, k = 0
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = w * 16 + jj * 8 + (lane >> 3);
            const int src_chunk = (lane & 7) ^ (jj * 4 + (lane >> 4));
            int nl = (r >> 6) * 128 + h * 64 + (r & 63);
            int ml = (r >> 5) * 64 + h * 32 + (r & 31);
            nl = (n0 + nl < N) ? nl : (N - 1 - n0);       // clamp: rows past the edge re-read a valid row
            ml = (m0 + ml < M) ? ml : (M - 1 - m0);
            offP[h][jj] = (uint32_t)nl * (uint32_t)ldw + src_chunk * 16;
            offQ[h][jj] = (uint32_t)ml * (uint32_t)ldx + src_chunk * 16;
        }
    const int8_t* gP = W + (int64_t)n0 * ldw;   // uniform; advanced by FBK per K-tile
    const int8_t* gQ = X + (int64_t)m0 * ldx;

    // LDS destinations (wave-uniform): half-tile base + piece*1024
    auto lds_half = [&](int buf, int isQ, int h) -> uint8_t* { return smem + buf * BUF_BYTES + isQ * 2 * HALF_BYTES + h * HALF_BYTES; };
    const int piece_off = w * 2048;   // pieces w*2 and w*2+1

    // ---- fragment read addresses (lane part), ks selects the 64-byte half of the 128-byte row
    constexpr int NPI = (SHAPE == 16) ? 4 : 2;     // P tiles per half (64 rows)
    constexpr int NQJ = (SHAPE == 16) ? 2 : 1;     // Q tiles per half (32 rows)
    constexpr int NKS = (SHAPE == 16) ? 2 : 4;     // MFMA k-steps per 128-byte row
    constexpr int NACC = (SHAPE == 16) ? 4 : 16;   // accumulator registers per tile
    const int frow = (SHAPE == 16) ? (lane & 15) : (lane & 31);
    const int fchunk = (SHAPE == 16) ? (lane >> 4) : (lane >> 5);
    const int fkey = (frow >> 1) & 7;
    // byte address of (row frow + rowbase, logical chunk c) = row*128 + ((c ^ key) * 16); c = ks*(8/NKS) + fchunk
    uint32_t lP[NKS], lQ[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = ks * (8 / NKS) + fchunk;
        lP[ks] = (uint32_t)((wp * 64 + frow) * 128 + ((c ^ fkey) * 16));
        lQ[ks] = (uint32_t)((wq * 32 + frow) * 128 + ((c ^ fkey) * 16)) + 2 * HALF_BYTES;
    }

    using acc_t = typename std::conditional<SHAPE == 16, v4i, v16i>::type;
    acc_t acc[2][2][NPI][NQJ];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NPI; ++i)
#pragma unroll
                for (int j = 0; j < NQJ; ++j)
#pragma unroll
                    for (int r = 0; r < NACC; ++r) acc[a][b][i][j][r] = 0;

    v4i fP[NPI][NKS], fQ0[NQJ][NKS], fQ1[NQJ][NKS];

    auto stage = [&](int buf, int isQ, int h) {   // 2 x global_load_lds_dwordx4 per wave
        uint8_t* l = lds_half(buf, isQ, h) + piece_off;
        const int8_t* g = isQ ? gQ : gP;
        const uint32_t o0 = isQ ? offQ[h][0] : offP[h][0], o1 = isQ ? offQ[h][1] : offP[h][1];
        glds16(g + o0, l);
        glds16(g + o1, l + 1024);
    };
    auto readP = [&](int bufoff, int h) {
#pragma unroll
        for (int i = 0; i < NPI; ++i)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                fP[i][ks] = *reinterpret_cast<const v4i*>(smem + bufoff + lP[ks] + h * HALF_BYTES + i * SHAPE * 128);
    };
    auto readQ = [&](int bufoff, int h, v4i (&f)[NQJ][NKS]) {
#pragma unroll
        for (int j = 0; j < NQJ; ++j)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                f[j][ks] = *reinterpret_cast<const v4i*>(smem + bufoff + lQ[ks] + h * HALF_BYTES + j * SHAPE * 128);
    };
    auto mma = [&](acc_t (&c)[NPI][NQJ], v4i (&fq)[NQJ][NKS]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < NPI; ++i)
#pragma unroll
                for (int j = 0; j < NQJ; ++j) {
                    if constexpr (SHAPE == 16) c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fP[i][ks], fq[j][ks], c[i][j], 0, 0, 0);
                    else c[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fP[i][ks], fq[j][ks], c[i][j], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    const int NT = K / FBK;

    // ---- prologue: all of tile 0, in need order P0, Q0, Q1, P1
    stage(0, 0, 0); stage(0, 1, 0); stage(0, 1, 1); stage(0, 0, 1);
    gP += FBK; gQ += FBK;
    PQ_WAIT_VMCNT(4);                       // P0, Q0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wp == 1) __builtin_amdgcn_s_barrier();   // stagger: waves 4-7 run one barrier behind

    int bufoff = 0;
    for (int kt = 0; kt < NT; ++kt) {
        const bool more = (kt + 1 < NT);
        const int nb = (kt + 1) & 1;
        // -------- phase 0: quadrant (hP0, hQ0)
        readQ(bufoff, 0, fQ0);
        __builtin_amdgcn_sched_barrier(0);
        readP(bufoff, 0);
        if (more) { stage(nb, 0, 0); PQ_WAIT_VMCNT(4); } else { PQ_WAIT_VMCNT(2); }   // Q1(t) landed
        __builtin_amdgcn_s_barrier();
        mma(acc[0][0], fQ0);
        __builtin_amdgcn_s_barrier();
        // -------- phase 1: quadrant (hP0, hQ1)
        readQ(bufoff, 1, fQ1);
        if (more) { stage(nb, 1, 0); PQ_WAIT_VMCNT(4); } else { PQ_WAIT_VMCNT(0); }   // P1(t) landed
        __builtin_amdgcn_s_barrier();
        mma(acc[0][1], fQ1);
        __builtin_amdgcn_s_barrier();
        // -------- phase 2: quadrant (hP1, hQ1)
        readP(bufoff, 1);
        if (more) { stage(nb, 1, 1); }
        __builtin_amdgcn_s_barrier();
        mma(acc[1][1], fQ1);
        __builtin_amdgcn_s_barrier();
        // -------- phase 3: quadrant (hP1, hQ0) — Q0 fragments kept in registers
        if (more) { stage(nb, 0, 1); PQ_WAIT_VMCNT(4); }                                // P0, Q0 (t+1) landed
        __builtin_amdgcn_s_barrier();
        mma(acc[1][0], fQ0);
        __builtin_amdgcn_s_barrier();
        gP += FBK; gQ += FBK;
        bufoff ^= BUF_BYTES;
    }
    if (wp == 0) __builtin_amdgcn_s_barrier();   // matches the stagger barrier of waves 4-7

    // ---- K4 epilogue: D[row <-> n][col <-> m]; lane holds 4 consecutive n per register group.
    using O = typename OutElem<OUT>::type;
    O* y = reinterpret_cast<O*>(epi.y);
    const bool has_bias = (OUT != OUT_I32) && epi.bias != nullptr;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) & (4 * sizeof(O) - 1)) == 0) && ((epi.ldy & 3) == 0);
    // column (m) index of this lane inside a Q tile, row (n) group inside a P tile
    const int dcol = (SHAPE == 16) ? (lane & 15) : (lane & 31);
    const int drow4 = (SHAPE == 16) ? (lane >> 4) * 4 : (lane >> 5) * 4;   // + 8*g for 32x32 groups
    constexpr int NG = (SHAPE == 16) ? 1 : 4;      // groups of 4 consecutive n per tile register file
#pragma unroll
    for (int hQ = 0; hQ < 2; ++hQ)
#pragma unroll
        for (int j = 0; j < NQJ; ++j) {
            const int m = m0 + wq * 64 + hQ * 32 + j * SHAPE + dcol;
            const bool mok = m < M;
            float as = 1.0f;
            if constexpr (OUT != OUT_I32) as = mok ? epi.a_scale[m] : 0.0f;
#pragma unroll
            for (int hP = 0; hP < 2; ++hP)
#pragma unroll
                for (int i = 0; i < NPI; ++i)
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const int n = n0 + wp * 128 + hP * 64 + i * SHAPE + drow4 + 8 * g;
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
                                    if (has_bias) bf = load_bias<OUT>(epi.bias, n + r);
                                }
                                o[r] = epi_convert<OUT>(c[g * 4 + r], as, bs, bf, has_bias);
                            }
                            if constexpr (sizeof(O) == 2) *reinterpret_cast<v2u*>(dst) = *reinterpret_cast<const v2u*>(o);
                            else *reinterpret_cast<v4u*>(dst) = *reinterpret_cast<const v4u*>(o);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (n + r >= N) continue;
                                float bs = 1.0f, bf = 0.0f;
                                if constexpr (OUT != OUT_I32) {
                                    bs = epi.b_scale[n + r];
                                    if (has_bias) bf = load_bias<OUT>(epi.bias, n + r);
                                }
                                dst[r] = epi_convert<OUT>(c[g * 4 + r], as, bs, bf, has_bias);
                            }
                        }
                    }
        }
}

bool gemm_fast_eligible(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return M >= 1 && N >= 1 && K >= FBK && (K % FBK) == 0 && (lda % 16) == 0 && (ldb % 16) == 0 && al(A) && al(B) &&
           M < (1 << 30) && N < (1 << 30) && lda < (1 << 23) && ldb < (1 << 23);
}

template <int OUT, int SHAPE>
void launch_gemm_fast(const int8_t* A, int64_t lda, const int8_t* B, int64_t ldb, const EpiArgs& epi,
                      int64_t M, int64_t N, int64_t K, hipStream_t st) {
    const int tiles_m = (int)((M + FT - 1) / FT), tiles_n = (int)((N + FT - 1) / FT);
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(512);
    gemm_s8_pp256<OUT, SHAPE><<<grid, block, 0, st>>>(A, lda, B, ldb, epi, (int)M, (int)N, (int)K, tiles_m, tiles_n);
}

#define PQ_INST(OUT, SHAPE) \
    template void launch_gemm_fast<OUT, SHAPE>(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
PQ_INST(PQ_BF16, 16) PQ_INST(PQ_FP16, 16) PQ_INST(PQ_F32, 16) PQ_INST(OUT_I32, 16)
PQ_INST(PQ_BF16, 32) PQ_INST(OUT_I32, 32)
#undef PQ_INST

}  // namespace pq
