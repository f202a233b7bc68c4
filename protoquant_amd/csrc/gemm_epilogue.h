// gemm_epilogue.h — K4: the fused dequant epilogue shared by every GEMM variant.
// OUT codes: PQ_BF16 / PQ_FP16 / PQ_F32 = QSPEC E1-E4 output; OUT_I32 = raw accumulator (parity twin).
#pragma once
#include "pq_common.h"

namespace pq {

constexpr int OUT_I32 = 3;

struct EpiArgs {
    const float* a_scale;   // [M] row scales   (unused for OUT_I32)
    const float* b_scale;   // [N] column scales
    const void* bias;       // [N] in output dtype, nullable
    void* y;                // [M, ldy]
    int64_t ldy;
    int32_t flags;          // EPI_* bits; 0 = QSPEC as written: (f32(acc) * a_scale[row]) * b_scale[col] (+ bias[col])
    int32_t nxcd = 8;          // XCDs of the device the launch goes to (the C-ABI entry points set it from hipDeviceAttributeNumberOfXccs): the tile remaps group tiles per L2
    int32_t y_any_align = 0;   // 1: the staged epilogue may store its 16-byte pieces at element-aligned addresses (odd leading dimensions, e.g. a 50257-wide vocabulary)
};
// rows of y fit the staged epilogue's 16-byte write-through stores: 16-byte aligned rows, or (y_any_align) any element-aligned address — gfx950 compute queues run
// with unaligned access enabled, and the stores are inline asm: no compiler alignment assumption is involved
__device__ __forceinline__ bool epi_rows_storable(const EpiArgs& e, const void* y, int ob) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(y);
    return e.y_any_align ? (a & (uintptr_t)(ob - 1)) == 0 : ((a & 15) == 0 && ((e.ldy * ob) & 15) == 0);
}
// The transposed product y^T = Wq . xq^T (pq_qlinear_s8_t: the GEMM's rows are output channels n, its columns tokens m) must
// still round as QSPEC E2-E3 say — the TOKEN scale first — so the column scale is applied first and the bias runs along rows.
constexpr int32_t EPI_COL_FIRST = 1, EPI_BIAS_ROWS = 2;
// EPI_STORE_T (the weight-streaming kernel only, alone): the product is computed in its normal orientation — tokens as the narrow side, QSPEC as written — and STORED
// transposed, y[n * ldy + m]: pq_qlinear_s8_t with few tokens (its swapped form would hand the tile kernels a 16-column problem).
constexpr int32_t EPI_STORE_T = 4;
// The two bits only ever travel together: 0 (y = x.W^T: row scale first, bias along columns) or the transposed form (column scale
// first; a bias, if any, along rows).  The staged epilogue dispatches on that pairing (PQ_EPI_STAGED_DISPATCH), the direct / split-K /
// skinny / generic epilogues read the bits one by one: any other combination would round differently on interior and edge tiles, so
// every launcher checks epi_flags_valid() (pq_api.hip: run_gemm) and nothing else constructs flags.
constexpr bool epi_flags_valid(int32_t flags, bool has_bias) {
    return flags == 0 || flags == EPI_STORE_T || (flags == EPI_COL_FIRST && !has_bias) || (flags == (EPI_COL_FIRST | EPI_BIAS_ROWS) && has_bias);
}

template <int OUT> struct OutElem { using type = typename Elem<OUT>::store_t; };
template <> struct OutElem<OUT_I32> { using type = int32_t; };

// one accumulator -> one output element value (still in registers)
template <int OUT>
__device__ __forceinline__ typename OutElem<OUT>::type epi_convert(int acc, float as, float bs, float bias_f, bool has_bias, bool col_first = false) {
    if constexpr (OUT == OUT_I32) {
        return acc;
    } else {
        float t = epilogue_val(acc, col_first ? bs : as, col_first ? as : bs);
        if (has_bias) t = t + bias_f;          // separate rounded add (built with -ffp-contract=off)
        return Elem<OUT>::from_f32(t);
    }
}

template <int OUT>
__device__ __forceinline__ float load_bias(const void* bias, int64_t n) {
    if constexpr (OUT == OUT_I32) return 0.0f;
    else return Elem<OUT>::to_f32(reinterpret_cast<const typename Elem<OUT>::store_t*>(bias)[n]);
}

// ------------------------------------------------------------------------------------------------
// Dispatch on the run-time (wave-uniform) bias / order flags to the compile-time specialisations.
#define PQ_EPI_STAGED_DISPATCH(OUT, NPT, NQT, QTP, PTP, has_bias, flags, ...)                                                          \
    do {                                                                                                                           \
        const int bmode_ = !(has_bias) ? 0 : (((flags) & EPI_BIAS_ROWS) ? 2 : 1);                                                   \
        if (!((flags) & EPI_COL_FIRST)) {                                                                                          \
            if (bmode_ == 0) epi_staged_block<OUT, NPT, NQT, QTP, PTP, 0, false>(__VA_ARGS__);                                      \
            else epi_staged_block<OUT, NPT, NQT, QTP, PTP, 1, false>(__VA_ARGS__);                                                  \
        } else {                                                                                                                   \
            if (bmode_ == 0) epi_staged_block<OUT, NPT, NQT, QTP, PTP, 0, true>(__VA_ARGS__);                                       \
            else epi_staged_block<OUT, NPT, NQT, QTP, PTP, 2, true>(__VA_ARGS__);                                                   \
        }                                                                                                                          \
    } while (0)

// Staged epilogue of ONE wave's block of 16x16 accumulator tiles (v_mfma_i32_16x16x64_i8 with the weight rows as the first
// operand): tile (pt, qt) holds, in lane l = 16 q + c, the four consecutive columns n = 16 pt + 4 q .. + 3 of output row
// m = 16 qt + c.  QSPEC E1-E4 in registers, then a transpose through a wave-private LDS region (so that a store
// instruction writes whole row segments of up to 256 bytes instead of 8 bytes per lane) and 16-byte global stores.
//
// Cost model (measured round 1: the epilogue was ISSUE-bound, ~3000 instructions per wave, not store-bound): everything that
// does not depend on the tile is hoisted — the column scales and the bias of the wave's NPT column tiles live in registers,
// the LDS write address is one register per column tile (+ an immediate per row tile), the bias is a template flag.
// Per 4 outputs: 4 v_cvt_f32_i32, 4 v_pk_mul_f32, (2 v_pk_add_f32,) 2 v_cvt_pk, 1 ds_write.
//
// The block is staged in passes of QT_PASS row tiles x PT_PASS column tiles that fit the wave's region
// (QT_PASS * 16 rows of PT_PASS * 16 * sizeof(out) bytes); LDS operations of one wave execute in order, so a pass may
// overwrite the region as soon as its reads are issued, and only the write -> read turn waits (lgkmcnt(0)).
// Swizzle: 16-byte chunk c of staged row r sits at chunk c ^ (r & KM): conflict-free ds_read_b128.
// 16-bit outputs (ds_write_b64: four groups of 16 contiguous lanes — the 16 rows of one tile at one q — on 32 banks, MI355X_MICROARCH.md section LDS): with the chunk key alone
// rows r and r + 8 of a 128- or 256-byte staged row meet on one bank pair — the 2-way conflict rounds 1-5 documented here and round 5's PMC pass counted (2^18 extra LDS cycles
// per 4096^3 launch = 4 per write: profiles/r06_lds_bank_conflicts.txt).  Round 6 (HSWZ): the key is 3 bits (r & 7) and the two 8-byte HALVES of a chunk trade places in rows
// with bit 3 set; the reader's row is it * RPI + lrow with lrow < RPI <= 8, so bit 3 is a compile-time property of the read and the halves are put back by naming the
// registers the other way round: no instruction added, SQ_LDS_BANK_CONFLICT = 0.  (64-byte staged rows — the 64 x 64 ring tile — keep the old form: there bit 3 of the row
// depends on the lane.)
typedef float v2f __attribute__((ext_vector_type(2)));

// BIAS: 0 none, 1 along columns (n = 16 pt + 4 q + r), 2 along rows (m = 16 qt + c; EPI_BIAS_ROWS).  SWAP: EPI_COL_FIRST.
template <int OUT, int NPT, int NQT, int QT_PASS, int PT_PASS, int BIAS, bool SWAP, typename AccFn, typename AsFn, typename BsFn>
__device__ __forceinline__ void epi_staged_block(AccFn&& acc_of, AsFn&& a_scale_of, BsFn&& b_scale_of, const void* bias_blk,
                                                 uint8_t* smem, uint32_t sw_off, uint8_t* y_blk, int64_t ldy_bytes, int lane) {
    using O = typename OutElem<OUT>::type;
    constexpr int OB = (int)sizeof(O);
    constexpr int RBU = PT_PASS * 16 * OB;                // bytes of a staged row that are USED: 64 (two half-precision column tiles: the 64 x 64 ring tile), 128, 160 (five: the 128 x 160 ring tile), 256, 320 or 512
    constexpr int RBY = RBU <= 64 ? 64 : RBU <= 128 ? 128 : RBU <= 256 ? 256 : 512;      // ... and its stride in the staging region: the next power of two (round 6; the swizzle wants one)
    constexpr int CPR = RBY / 16;                         // 16-byte chunks per staged row (stride)
    constexpr int CPU = RBU / 16;                         // ... of which are used
    constexpr bool HSWZ = (OB == 2) && RBY >= 128;        // 16-bit outputs: half-swap swizzle (above)
    constexpr int KM = HSWZ ? 7 : ((CPR < 16 ? CPR : 16) - 1);         // swizzle key mask
    constexpr int RPI = 64 / CPR;                         // rows per ds_read_b128 / global store instruction: 8, 4 or 2
    constexpr int CSH = (OB == 2) ? 1 : 2;                // chunk of column tile p (inside a pass) = (p << CSH) | b
    static_assert(NQT % QT_PASS == 0 && NPT % PT_PASS == 0 && RBU <= 512 && RBU % 16 == 0, "epilogue pass shape");
    const int dcol = lane & 15, q = lane >> 4;
    const int b = (OB == 2) ? (q >> 1) : q;
    const uint32_t low = (OB == 2) ? (uint32_t)((q & 1) ^ (HSWZ ? ((dcol >> 3) & 1) : 0)) * 8u : 0u;
    const uint32_t k2 = (uint32_t)(b ^ (dcol & KM));
    const uint32_t wbase = sw_off + (uint32_t)dcol * RBY + low;
    // hoisted per column tile: scales (and bias) of n = 16 pt + 4 q .. + 3
    v4f bs[NPT], bf[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        bs[pt] = v4f{1.f, 1.f, 1.f, 1.f};
        bf[pt] = v4f{0.f, 0.f, 0.f, 0.f};
        if constexpr (OUT != OUT_I32) {
            bs[pt] = b_scale_of(pt);
            if constexpr (BIAS == 1) {
                const O* bp = reinterpret_cast<const O*>(bias_blk) + pt * 16 + q * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) bf[pt][r] = Elem<OUT>::to_f32(bp[r]);
            }
        }
    }
    const int lrow = lane / CPR, ch = lane % CPR;
    const uint32_t rbase = sw_off + (uint32_t)lrow * RBY;
    const uint32_t vlane = (uint32_t)lrow * (uint32_t)ldy_bytes + (uint32_t)ch * 16u;
#pragma unroll
    for (int qp = 0; qp < NQT / QT_PASS; ++qp) {
#pragma unroll
        for (int pp = 0; pp < NPT / PT_PASS; ++pp) {
#pragma unroll
            for (int ql = 0; ql < QT_PASS; ++ql) {
                const int qt = qp * QT_PASS + ql;
                float as = 1.0f, brow = 0.0f;
                if constexpr (OUT != OUT_I32) {
                    as = a_scale_of(qt);
                    if constexpr (BIAS == 2) brow = Elem<OUT>::to_f32(reinterpret_cast<const O*>(bias_blk)[qt * 16 + dcol]);
                }
#pragma unroll
                for (int pl = 0; pl < PT_PASS; ++pl) {
                    const int pt = pp * PT_PASS + pl;
                    const v4i c = acc_of(pt, qt);
                    uint8_t* d = smem + (wbase + ((((uint32_t)pl << CSH) ^ k2) << 4)) + ql * 16 * RBY;
                    if constexpr (OUT == OUT_I32) {
                        *reinterpret_cast<v4i*>(d) = c;
                    } else {
                        v2f t0 = {(float)c[0], (float)c[1]}, t1 = {(float)c[2], (float)c[3]};   // E1
                        if constexpr (!SWAP) {
                            t0 = t0 * v2f{as, as};                                               // E2 (each lane-op rounds separately)
                            t1 = t1 * v2f{as, as};
                            t0 = t0 * v2f{bs[pt][0], bs[pt][1]};                                 // E3
                            t1 = t1 * v2f{bs[pt][2], bs[pt][3]};
                        } else {                                                                 // transposed product: the columns are the tokens
                            t0 = t0 * v2f{bs[pt][0], bs[pt][1]};
                            t1 = t1 * v2f{bs[pt][2], bs[pt][3]};
                            t0 = t0 * v2f{as, as};
                            t1 = t1 * v2f{as, as};
                        }
                        if constexpr (BIAS == 1) {                                               // E4: a separate rounded add
                            t0 = t0 + v2f{bf[pt][0], bf[pt][1]};
                            t1 = t1 + v2f{bf[pt][2], bf[pt][3]};
                        } else if constexpr (BIAS == 2) {
                            t0 = t0 + v2f{brow, brow};
                            t1 = t1 + v2f{brow, brow};
                        }
                        O o[4] = {Elem<OUT>::from_f32(t0[0]), Elem<OUT>::from_f32(t0[1]), Elem<OUT>::from_f32(t1[0]), Elem<OUT>::from_f32(t1[1])};
                        if constexpr (OB == 2) *reinterpret_cast<v2u*>(d) = *reinterpret_cast<const v2u*>(o);
                        else *reinterpret_cast<v4u*>(d) = *reinterpret_cast<const v4u*>(o);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's writes have retired (wave-private region)
#pragma unroll
            for (int it = 0; it < QT_PASS * 16 / RPI; ++it) {
                const int r = it * RPI + lrow;                                  // staged row of this lane
                if (CPU < CPR && ch >= CPU) continue;                            // (a padded stride: the lanes of the unused chunks sit the store out)
                v4u v = *reinterpret_cast<const v4u*>(smem + rbase + it * RPI * RBY + (((uint32_t)(ch ^ (r & KM))) << 4));
                if constexpr (HSWZ) {
                    static_assert(RPI <= 8, "bit 3 of the staged row must not depend on the lane");
                    if ((it * RPI) & 8) v = v4u{v[2], v[3], v[0], v[1]};         // (compile-time: `it` is unrolled)
                }
                uint8_t* rowp = y_blk + (int64_t)(qp * QT_PASS * 16 + it * RPI) * ldy_bytes + pp * RBU;   // wave-uniform
                store_wt_b128(rowp + vlane, v);                                  // write-through (pq_common.h)
            }
        }
    }
}

}  // namespace pq
