// gemm_tile_common.h — helpers shared by the tiled GEMM translation units (gemm_s8_fast.hip, gemm_s8_ring.hip): s_waitcnt immediates,
// the LDS-DMA forms, the XCD-aware block remap, a compile-time loop.
#pragma once
#include <type_traits>

#include "gemm_epilogue.h"

namespace pq {

constexpr int FT = 256;          // tile edge (both m and n)
constexpr int FBK = 128;         // K bytes per tile step

// s_waitcnt immediate (gfx9 encoding): vmcnt[3:0] | expcnt(7) << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | 0x70 | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }
// vmcnt(n) for a run-time n (prologue only): the builtin wants a literal
__device__ __forceinline__ void wait_vmcnt_lgkm0(int n) {
    switch (n) {
#define PQ_W(k) case k: __builtin_amdgcn_s_waitcnt(waitcnt_imm(k, 0)); break;
        PQ_W(0) PQ_W(1) PQ_W(2) PQ_W(3) PQ_W(4) PQ_W(5) PQ_W(6) PQ_W(7) PQ_W(8) PQ_W(9) PQ_W(10) PQ_W(11) PQ_W(12) PQ_W(13) PQ_W(14) PQ_W(15)
        PQ_W(16) PQ_W(17) PQ_W(18) PQ_W(19) PQ_W(20) PQ_W(21) PQ_W(22) PQ_W(23) PQ_W(24) PQ_W(25) PQ_W(26) PQ_W(27) PQ_W(28) PQ_W(29) PQ_W(30) PQ_W(31)
        PQ_W(32) PQ_W(33) PQ_W(34) PQ_W(35) PQ_W(36) PQ_W(37) PQ_W(38) PQ_W(39) PQ_W(40) PQ_W(41) PQ_W(42) PQ_W(43) PQ_W(44) PQ_W(45) PQ_W(46) PQ_W(47)
        PQ_W(48) PQ_W(49) PQ_W(50) PQ_W(51) PQ_W(52) PQ_W(53) PQ_W(54) PQ_W(55) PQ_W(56) PQ_W(57) PQ_W(58) PQ_W(59) PQ_W(60) PQ_W(61) PQ_W(62) PQ_W(63)
#undef PQ_W
        default: __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0)); break;
    }
}

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// LDS-DMA in its SGPR-base + 32-bit-VGPR-offset form: no 64-bit VALU address math beside the MFMAs.
// base must be wave-uniform, lds_addr a wave-uniform LDS byte address.  M0 is written in the same statement that reads it
// and NOT restored: these kernels contain no compiler-generated user of M0 (every LDS-DMA goes through these helpers;
// `make asm` + grep m0 confirms) — round 1 saved and restored it around every piece, 16 extra SALU per K-tile per wave.
// hipcc does not count this load: the K-loop waits with explicit vmcnt.
__device__ __forceinline__ void glds16_sbase(const int8_t* base, uint32_t voff, uint32_t lds_addr) {
    // (cache-policy bits on this load — sc1 / sc0, which bypass the vector L1 — measured within 0.2 % of none on every tile kind:
    // profiles/r02_ab_experiments.txt)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// the same with a per-lane 64-bit source address (scale vectors in the prologue)
__device__ __forceinline__ void glds16_vaddr(const void* src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds_addr) : "memory");
}

// XCD-aware bijective remap: workgroups are dealt to the XCDs round-robin, so blocks that share an XCD (equal bid % nx) get a contiguous run of tiles.  nx = the XCDs of
// the device (8 on a whole MI355X; a partitioned device has fewer — 1 makes this the identity); any nx >= 1 gives a bijection.
__device__ __forceinline__ int xcd_remap(int bid, int nwg, int nx = 8) {
    int q, r, xcd, idx;
    if (nx == 8) { q = nwg >> 3; r = nwg & 7; xcd = bid & 7; idx = bid >> 3; }
    else { q = nwg / nx; r = nwg - q * nx; idx = bid / nx; xcd = bid - idx * nx; }
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// K-tiles per rotation chunk of the loader / consumer tiles ("K ROTATION in chunks", gemm_s8_ring.hip): the chunk of every weight panel an XCD streams at one time
// (32 CUs -> 32 / sharers panels of tn rows each) within ~2 MiB of its 4-MiB L2, at most 16.  Measured (profiles/r04_rotation.txt, r04_midm_ab_raw.txt): 8 .. 32 K-tiles
// within 3 % where they fit; a chunk that does not fit (128 x 256 tiles at 16 K-tiles: 4 MiB) is 25 % SLOWER than no rotation at all.
inline int rot_chunk_ktiles(int sharers, int tn) {
    const int panels = (32 + sharers - 1) / sharers;
    int ct = (int)((2 << 20) / ((int64_t)panels * tn * FBK));
    return ct < 4 ? 4 : (ct > 16 ? 16 : ct);
}

// the activation operand as an all-gather leaves int8 codes: K-slab s = [M][k_per_slab] at `stride` bytes per slab (pq_qlinear_s8_kslabs); tiles = 0: plain row-major
struct KSlabs {
    int tiles = 0;            // K-tiles (128 B) per slab
    uint32_t magic = 0;       // ceil(2^32 / tiles): kt / tiles == (kt * magic) >> 32 for kt < 2^16 (tiles >= 2; tiles == 1 is handled apart: 2^32 does not fit)
    int64_t stride = 0;       // bytes between slabs
};

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

}  // namespace pq
