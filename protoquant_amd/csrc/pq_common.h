// pq_common.h — shared device helpers for the gfx950 dynamic-int8 linear path (QSPEC v2).
// gfx950 only: 64-wide wavefronts are hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pq_hip.h"

namespace pq {

constexpr int kWave = 64;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- behaviour switches (host side)
// Every switch of the library lives in ONE immutable snapshot.  pq_set_option() (and the one-time pass over the environment) builds a modified
// copy and publishes it with one atomic pointer swap; every C-ABI entry point pins the snapshot that is live when it is entered (thread-local,
// pq_api.hip: CallScope) and the planners and launchers of that call read nothing else — so a call plans and launches under ONE consistent set of
// switches whatever other host threads set meanwhile, and concurrent launches from several host threads are defined.  Snapshots are never freed
// (a few hundred bytes per pq_set_option call; tests and experiments only).
struct Options {
    int variant = 0;                 // PQ_FORCE_VARIANT (pq_api.hip: enum Variant; 0 = auto)
    bool no_tailsplit = false, no_splitk = false;
    int force_splitk = 0;
    int fsk = -1;                    // fused split-K: -1 = by plan (fsk_plan), 0 = never, S > 1 = S slices whenever the shape admits them (experiments)
    bool fsk_symmetric = false;      // PQ_FSK_SYMMETRIC=1: the symmetric exchange for 2 / 4 slices (waits for partner workgroups: see pq_hip.h); default: the ticket form
    int midm_ct = 0;                 // PQ_MIDM_CT: K-tiles per rotation chunk of the mid-M ring tiles (0 = by rule, 1 = no rotation)
    bool no_kslabs = false;          // PQ_NO_KSLABS=1: pq_qlinear_s8_kslabs always takes the layout pass (never walks the slabs in place)
    bool no_ring160 = false;        // PQ_NO_RING160=1: never plan the 128 x 160 ring tile (round 6)
    bool no_midm = false;            // PQ_NO_MIDM=1: no 64-row ring tiles for 64 < M <= 512 (the round-3 dispatch)
    bool fsk_coop = false;           // PQ_FSK_COOP=1: launch the symmetric fused split-K kernels COOPERATIVELY (co-residency guaranteed by the runtime, ticket form on error).
                                     // Off by default: measured +21-24 us per launch on ROCm 7.2 / gfx950, against 2-5 us the symmetric exchange saves (profiles/r05_ab_fsk_coop.txt)
    bool fsk_fenced = false;         // PQ_FSK_FENCED=1: the ticket hand-over with the documented agent-scope release / acquire (buffer_wbl2 sc1 / buffer_inv sc1) as well
    int fake_cus = 0;                // PQ_FAKE_CUS=n: plan as if the device had n CUs (tests of the residency guard)
    bool skinny_stage = true;        // PQ_SKINNY_STAGE=0: the weight-streaming kernel with its activation fragments straight from L2 (rounds 1-3)
    int skinny_rb = 0;               // 0 auto, 1 / 2 forced
    int k1_rpw = 0;                  // 0 auto, 1 / 2 forced
    bool k1_st16 = false;            // 16-byte code stores in K1: A/B in profiles/r03_k1_st16.txt
    int skinny_ks = 0;               // PQ_SKINNY_KS (experiments): forced K-split of the weight-streaming kernel (waves per workgroup)
    bool epi_any_align = true;       // PQ_EPI_ANY_ALIGN=0: the staged epilogue only for 16-byte aligned rows of y (round 4 default: any element-aligned row — odd leading dimensions, a 50257-wide vocabulary: 498 -> 370 us)
    int k2_blocks_a = 0, k2_blocks_e = 0;   // PQ_K2_BLOCKS_A / _E (experiments): target workgroup counts of K2's amax / encode passes (0 = default)
    int k1_lds = 0;                  // experiment: bytes of unused dynamic LDS per block = a cap on resident blocks per CU
    int ring_rot = 1;                // PQ_RING_ROT: K rotation of the 128 x 128 ring tile's loaders (0 = off, 1 = on with the rule's chunk, n > 1 = n K-tiles per chunk)
    bool ring_lc = true;             // loader / consumer split of the 128 x 128 ring tile
    int sp128_lc = 1;                // loader / consumer split of the 128 x 256 tile: 1 = 4 consumers + 4 loaders, 0 = 8 symmetric waves, 2 = 12 waves (dev builds)
    bool sp256_p3 = true;            // split rings of the 256 x 256 tile (weights 3 slots deep)
    int sp256_asm = 1;               // K-loop variant of kloop_p3_asm.inc; 0 = the HIP loop
    bool sp256_persist = false;
    int silu_tpr = 0;                // 0 auto; 256 forces the 256-thread layout on wide rows
    int rms_wave_max = 256;
    int (*roctx_push)(const char*) = nullptr;      // PQ_ROCTX=1 (environment only)
    int (*roctx_pop)() = nullptr;
};
// the snapshot pinned by the C-ABI call in progress on this thread (outside a call: the live one)
const Options& opt();

// ---------------------------------------------------------------- write-through global stores
// Output streams (codes, dequantised tiles, y) are stored with the sc1 cache policy: the line is written through to the memory side
// and dropped from the XCD's L2 instead of staying there dirty.  A kernel that leaves B dirty bytes in L2 pays ~B / 6 TB/s at its
// end for the write-back (MI355X_MICROARCH.md, price-list row "boundary"): measured on this path, K1 at 4096 x 4096 9.3 -> 6.4 us
// (16 MB of codes), the 4096^3 GEMM 50.9 -> 49.7 us (32 MB of y), the qlinear step 61.4 -> 59.6 us — same bits
// (profiles/r02_write_through_stores.txt).  The next kernel reads these buffers from the Infinity Cache either way (the XCDs' L2s are
// not coherent, and the consumer's tiles rarely sit on the producer's XCD).  Non-temporal stores (nt) were tried too: K1 alone is as
// fast, but the consumer then misses the Infinity Cache as well and the step gets slower.
// hipcc does not count an asm store in its vmcnt bookkeeping; that only makes its own counted waits stricter (vmcnt retires in order).
__device__ __forceinline__ void store_wt_b128(void* p, const v4u& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");   // (s_nop 1: >64-bit store data hazard)
}
__device__ __forceinline__ void store_wt_b64(void* p, const v2u& v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_wt_b32(void* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }

// ---------------------------------------------------------------- element types
// DT: 0 bf16, 1 fp16 (both stored as 16-bit patterns), 2 f32.
template <int DT> struct Elem;
template <> struct Elem<PQ_BF16> {
    using store_t = uint16_t;
    static constexpr int kBytes = 2;
    __device__ static __forceinline__ float to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
    // plain cast: v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN stays NaN)
    __device__ static __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
};
template <> struct Elem<PQ_FP16> {
    using store_t = uint16_t;
    static constexpr int kBytes = 2;
    __device__ static __forceinline__ float to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    // The empty asm pins f as a rounded binary32 value: without it hipcc folds "f32 multiply, then convert" into
    // v_fma_mixlo_f16, which rounds the exact product ONCE — a different result whenever the f32 product lands on an fp16 tie
    // (QSPEC rounds to f32 first).  Found by tests/fuzz_quant.py in the RMSNorm kernel; gfx950 has no bf16 counterpart.
    __device__ static __forceinline__ uint16_t from_f32(float f) {
        asm volatile("" : "+v"(f));
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
};
template <> struct Elem<PQ_F32> {
    using store_t = float;
    static constexpr int kBytes = 4;
    __device__ static __forceinline__ float to_f32(float f) { return f; }
    __device__ static __forceinline__ float from_f32(float f) { return f; }
};

// ---------------------------------------------------------------- QSPEC scalar rules
// Q2 (QSPEC v2): running max of |x| that PROPAGATES a NaN, as torch.amax does.  On the bit patterns of sign-cleared floats every NaN
// sorts above +Inf, so an unsigned integer max is the float max with NaN propagation (and costs the same v_max).
__device__ __forceinline__ float amax_merge(float a, float b) {      // a, b: non-negative, or NaN with the sign bit cleared
    const uint32_t x = __builtin_bit_cast(uint32_t, a), y = __builtin_bit_cast(uint32_t, b);
    return __builtin_bit_cast(float, x > y ? x : y);
}
__device__ __forceinline__ float amax_step(float amax, float v) {
    return amax_merge(amax, __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu));
}
// Q3: scale = amax / 127 (IEEE division), 1 when amax == 0; a NaN scale is THE canonical quiet NaN 0x7FC00000.
__device__ __forceinline__ float scale_of(float amax) {
    if (amax != amax) return __builtin_bit_cast(float, 0x7FC00000u);
    return amax == 0.0f ? 1.0f : amax / 127.0f;
}
// Q4-Q6: code = clamp(rne(x / scale), -128, 127), NaN -> 0.   True division: never x * (1/scale).
__device__ __forceinline__ int code_of(float x, float scale) {
    float t = __builtin_rintf(x / scale);
    t = (t != t) ? 0.0f : t;
    t = t > 127.0f ? 127.0f : t;
    t = t < -128.0f ? -128.0f : t;
    return (int)t;
}
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)d << 24);
}

// wave-wide NaN-propagating max of sign-cleared floats (all 64 lanes get the result)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = amax_merge(v, __shfl_xor(v, off, 64));
    return v;
}

// E1-E4 epilogue value for one accumulator.
__device__ __forceinline__ float epilogue_val(int acc, float a_scale, float b_scale) {
    float t = (float)acc;      // v_cvt_f32_i32: RNE
    t = t * a_scale;           // row scale first
    t = t * b_scale;           // then column scale
    return t;
}

}  // namespace pq
