/* pq_hip.h — C-ABI of libpq_hip.so: the MI355X (gfx950) dynamic-int8 linear hot path.
 *
 * This is the drop-in boundary.  The reference's operator interface for this path
 * is the Python API BASELINE.json names (QTensor, quantize(), dequantize(), qlinear); its source is
 * absent from the mount (/root/reference holds only CODE_OF_CONDUCT.md:1-80), so each entry point
 * cites the contract clause / primitive it replaces instead of a reference file:line.
 *
 * Conventions
 *  - plain pointers + sizes; every buffer is CALLER-OWNED DEVICE memory on the current HIP device;
 *    the library never allocates, frees, copies to host or synchronises.
 *  - stream-ordered and asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*;
 *    NULL = the null stream).  Safe to capture into a hipGraph.
 *  - every function returns a pq_status; pq_last_error() gives the thread-local message of the last
 *    failing call on this thread.  Nothing throws across the ABI.
 *  - matrices are row-major with explicit leading dimensions in ELEMENTS.
 *  - dtype codes: 0 = bf16, 1 = fp16, 2 = f32.
 *  - numeric contract: QSPEC v2.  int32 accumulators are exact; float stages are
 *    IEEE binary32, round-to-nearest-even, no contraction, true division.  A NaN in a token row (weight channel)
 *    PROPAGATES: that row's scale is the canonical quiet NaN 0x7FC00000, its codes are 0 and its qlinear output is NaN —
 *    what the unquantised linear would give; an Inf gives scale = Inf, codes 0 and a NaN output row.
 */
#ifndef PQ_HIP_H
#define PQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQ_ABI_VERSION 1

typedef enum {
    PQ_OK = 0,
    PQ_ERR_BAD_ARG = 1,        /* null pointer, negative size, ld too small, unknown dtype/axis */
    PQ_ERR_BAD_ALIGN = 2,      /* reserved: every alignment is currently served by a generic path */
    PQ_ERR_LAUNCH = 3,         /* hipGetLastError() after a launch was not hipSuccess */
    PQ_ERR_UNIMPLEMENTED = 4,
    PQ_ERR_WORKSPACE = 5,      /* caller-provided workspace smaller than pq_*_workspace_bytes() */
    PQ_ERR_COMM = 6            /* RCCL error (multi-GPU entry points) */
} pq_status;

enum { PQ_BF16 = 0, PQ_FP16 = 1, PQ_F32 = 2 };

/* ABI version of the loaded library (== PQ_ABI_VERSION it was built with). */
int32_t pq_version(void);
/* Message of the last failing call on the calling thread ("" if none). Valid until the next call. */
const char* pq_last_error(void);
/* Behaviour switches for tests and experiments: PQ_FORCE_VARIANT (generic | sp256_16 | sp128_16 | sp128x128 | ring128 |
 * ring64x128 | ring64x64 | ring128x160 | skinny | "" = auto), PQ_NO_RING160 (1 = never plan the 128 x 160 ring tile: the round-5 dispatch), PQ_NO_KSLABS (1 = stacked codes always take
 * the layout pass), PQ_NO_TAILSPLIT, PQ_NO_SPLITK, PQ_FORCE_SPLITK (slice count: experiments), PQ_FSK (0 = no fused
 * split-K, S = S slices), PQ_FSK_SYMMETRIC and PQ_FSK_FENCED (see pq_qlinear_s8), PQ_NO_MIDM (no 64-row ring tiles), PQ_RING_ROT (0 = no K rotation), PQ_FAKE_CUS (plan as if the device had n CUs),
 * PQ_SKINNY_RB ("" = off / auto), PQ_EPI_ANY_ALIGN (0 = the staged epilogue only for 16-byte aligned output rows; default: any element-aligned row),
 * PQ_K2_BLOCKS_A / PQ_K2_BLOCKS_E (workgroup-count targets of K2's two passes).  The environment variables of the same names are read ONCE, at the first call into the
 * library; this call changes a switch afterwards.
 * Threading: the switches live in an immutable snapshot; pq_set_option publishes a modified copy with one atomic pointer
 * swap, and every other entry point pins the snapshot that is live when it is ENTERED and plans and launches under that
 * one — so pq_set_option may race with launches on other host threads (each call sees the old or the new set, never a
 * mixture), and launches from several host threads at once are defined.  There is no other mutable global state in the
 * library (one-time init aside).  A captured hipGraph keeps the choice that was live at capture time. */
int32_t pq_set_option(const char* name, const char* value);

/* K1 — per-token dynamic symmetric int8 quantisation: replaces quantize(x) of the contract for an
 * activation x[rows, cols] (amax over cols).  q[rows, cols] int8, scale[rows] f32.   QSPEC Q1-Q6. */
int32_t pq_quant_rowwise(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x,
                         int8_t* q, int64_t ld_q, float* scale, void* stream);

/* K2 — per-channel quantisation of a row-major matrix along its strided axis (amax over rows):
 * replaces quantize(W) for a [K, N]-stored weight.  scale[cols] f32.  Three stream-ordered graph nodes
 * (a fill kernel, the amax pass, the encode pass: three kernel nodes under capture); `scale` doubles as the amax scratch — no workspace.   QSPEC Q1-Q6. */
int32_t pq_quant_colwise(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x,
                         int8_t* q, int64_t ld_q, float* scale, void* stream);

/* K1 fused into its producer: quantize(F.silu(g) * u) per token in one pass — the activation of the
 * `down` projection of a gated MLP (BASELINE config 3) — so the bf16 product never goes to HBM.  g, u: [rows, cols] of
 * `dtype` with their own leading dimensions (the two column halves of a fused gate+up output qualify: ld = 2*cols).
 * h_out (nullable, ld_h): also store h = silu(g)*u in `dtype`.  Numerics: QSPEC S1-S6 (specified exponential, IEEE
 * division, storage rounding after silu and after the product), then Q1-Q6 on the rows of h. */
int32_t pq_silu_mul_quant_rowwise(const void* g, int64_t ld_g, const void* u, int64_t ld_u, int32_t dtype,
                                  int64_t rows, int64_t cols, int8_t* q, int64_t ld_q, float* scale,
                                  void* h_out, int64_t ld_h, void* stream);

/* The two halves of pq_silu_mul_quant_rowwise for an intermediate whose COLUMNS are sharded over ranks (BASELINE config 5, gate/up and down both
 * column-sharded): a rank holds g, u[rows, cols_local].  The row amax of h = silu(g)*u is an exact max, so
 *   pq_silu_mul_rowamax            -> amax_bits[rows]: the f32 bit pattern of max |h| over the LOCAL columns (non-negative floats and NaNs order as
 *                                     unsigned integers: a NaN propagates, as Q2 says);
 *   all-reduce(max) of amax_bits as uint32 over the ranks (pq_allreduce_max_u32 in pq_rccl.h, or any other exact max);
 *   pq_silu_mul_quant_rowwise_amax -> q[rows, cols_local] and scale[rows] against the GLOBAL amax (h is recomputed: S1-S5 are deterministic)
 * give every rank the column block of the unsharded kernel's codes and the unsharded scale vector, bit for bit; the int8 blocks (1 byte per
 * element instead of 2 x 2 bytes of bf16 g and u) are what travels.  Same numerics, layouts and argument meaning as pq_silu_mul_quant_rowwise.
 * Precondition of the encode half: amax_bits[r] >= the row's local amax (true of any max that includes this block) — the exact encode does not clamp, a smaller
 * amax makes codes wrap. */
int32_t pq_silu_mul_rowamax(const void* g, int64_t ld_g, const void* u, int64_t ld_u, int32_t dtype, int64_t rows, int64_t cols,
                            uint32_t* amax_bits, void* stream);
int32_t pq_silu_mul_quant_rowwise_amax(const void* g, int64_t ld_g, const void* u, int64_t ld_u, int32_t dtype, int64_t rows, int64_t cols,
                                       const uint32_t* amax_bits, int8_t* q, int64_t ld_q, float* scale, void* stream);

/* The same two halves for a PLAIN activation whose columns are sharded over ranks (a rank's heads of the attention output feeding a column-sharded `o`
 * projection): pq_quant_rowamax -> all-reduce(max) -> pq_quant_rowwise_amax give the column block of pq_quant_rowwise's codes on the whole row and its scale
 * vector, bit for bit (QSPEC Q1-Q6; wide rows past 32 768 / 16 384 columns per rank and unaligned operands take the generic kernels). */
int32_t pq_quant_rowamax(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x, uint32_t* amax_bits, void* stream);
int32_t pq_quant_rowwise_amax(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x, const uint32_t* amax_bits, int8_t* q, int64_t ld_q,
                              float* scale, void* stream);

/* K1 fused into RMSNorm: quantize(weight * (x.float() * rsqrt(mean(x.float()^2) + eps)).to(dtype)) per
 * token in one pass — the input of the q/k/v and gate/up projections of a decoder layer.  x: [rows, cols], weight: [cols],
 * both of `dtype`; h_out (nullable, ld_h): also store the normalised activation.  Numerics: QSPEC N1-N6 — the sum of squares
 * is reduced in a pinned order (16-byte vectors dealt to 256 lanes, xor butterfly per 64 lanes, four partial sums left to
 * right), IEEE sqrt and division, storage rounding after x*rs and after the weight product — then Q1-Q6.  cols < 2^24. */
int32_t pq_rmsnorm_quant_rowwise(const void* x, int64_t ld_x, const void* weight, float eps, int32_t dtype,
                                 int64_t rows, int64_t cols, int8_t* q, int64_t ld_q, float* scale,
                                 void* h_out, int64_t ld_h, void* stream);

/* dequantize(): out[r,c] = cast_rne(f32(q[r,c]) * scale[axis==0 ? c : r]).  `axis` is the axis the
 * scale was reduced over (1: one scale per row, 0: one scale per column).   QSPEC D1. */
int32_t pq_dequant(const int8_t* q, int64_t ld_q, const float* scale, int32_t axis,
                   int64_t rows, int64_t cols, void* out, int64_t ld_out, int32_t out_dtype, void* stream);

/* K3 debug/parity twin — c[M,N] = sum_k a[M,k] * b[N,k], exact int32: the drop-in for
 * torch._int_mm(a, b.t()) (aten::_int_mm), which BASELINE.json names as the CPU oracle. */
int32_t pq_gemm_s8s8s32(const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb,
                        int32_t* c, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream);

/* K3+K4 — int8 GEMM on v_mfma_i32_*_i8 with the fused dequant epilogue:
 *   y[m,n] = cast_rne_out((f32(acc[m,n]) * a_scale[m]) * b_scale[n] (+ f32(bias[n])))   QSPEC E1-E4.
 * bias is nullable and has the output dtype.
 * workspace: optional.  pq_qlinear_workspace_bytes(M,N,K) > 0 marks problems (small M*N, long K) for which a
 * 16-byte aligned device workspace of that size enables split-K — partial int32 sums either in slabs of the whole
 * output + an exact integer reduction pass, or handed over between the workgroups of a tile inside the GEMM kernel
 * (tickets + per-tile slabs; the call re-initialises the tickets itself): results are bit-identical.  The contents need
 * not be initialised or preserved; one workspace must not serve two calls that may run concurrently.  With
 * workspace == NULL the single-pass kernel runs instead.
 * Liveness of the in-kernel hand-over: the default (ticket) form never waits for a workgroup that may not be running —
 * the workgroups of a tile that finish first store their partial sums and leave, the last one adds them — so it is safe
 * under any placement: several such GEMMs on concurrent streams, CU-masked queues, partitioned devices, a co-running
 * persistent kernel.  It is planned only when tiles x slices <= the CUs the current device reports (a performance rule).
 * Visibility of the handed-over sums rests on write-through (sc1) stores whose acknowledgement (vmcnt) precedes the ticket,
 * and agent-scope (sc1) loads behind a poll + barrier: a sequence measured valid on gfx950 / ROCm 7.2, not
 * an architectural guarantee; PQ_FSK_FENCED=1 adds the documented release / acquire (buffer_wbl2 sc1 / buffer_inv sc1) as
 * a fallback, at ~35 us per launch.
 * PQ_FSK_SYMMETRIC=1 opts into the symmetric exchange for 2 / 4 slices (each workgroup keeps a part of the tile and
 * WAITS for its partners' contributions: 2-5 % faster): the caller then guarantees that every workgroup of the launch
 * can be resident at once — no second fused split-K GEMM in flight on another stream, no CU mask — and the planner
 * additionally refuses it when tiles x slices exceeds the device's CU count.  PQ_FSK_COOP=1 launches those symmetric kernels
 * COOPERATIVELY instead (hipLaunchCooperativeKernel: the runtime guarantees co-residency or refuses, then the ticket form runs) — correct, also under
 * hipGraph capture, but measured 21-24 us slower per launch than the ticket form on ROCm 7.2 (profiles/r05_ab_fsk_coop.txt): opt-in.
 * Output rows that are not 16-byte aligned (an odd ldy, e.g. a 50257-wide vocabulary): the staged epilogue stores its 16-byte pieces at element-aligned addresses,
 * which needs the queue's unaligned-access mode (SH_MEM_CONFIG alignment mode "unaligned" — the default of ROCm compute queues on gfx9, but a platform setting, not an
 * architectural guarantee: some virtual functions and debug configurations run strict).  On such a platform set PQ_EPI_ANY_ALIGN=0 (environment or pq_set_option): those
 * rows then take the slower guarded direct stores; everything else is unaffected. */
int32_t pq_qlinear_s8(const int8_t* a, int64_t lda, const float* a_scale,
                      const int8_t* b, int64_t ldb, const float* b_scale,
                      const void* bias, void* y, int64_t ldy, int32_t out_dtype,
                      int64_t M, int64_t N, int64_t K,
                      void* workspace, size_t workspace_bytes, void* stream);
size_t pq_qlinear_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* The same qlinear with the output TRANSPOSED: yt[N, M] (leading dimension ldyt), yt[n][m] bit-identical to pq_qlinear_s8's
 * y[m][n] (the epilogue keeps QSPEC's order — token scale first — although the tokens are now the GEMM's columns; the bias
 * runs along rows).  For the column-sharded configuration (SURVEY.md §8(e) option 1): the ranks' yt shards [N/G, M] are row
 * blocks of yt, so the all-gather is contiguous and needs no layout pass (pq_allgather_rows_t in pq_rccl.h).
 * Arguments as pq_qlinear_s8 (a, a_scale: activation codes [M, K] and token scales; b, b_scale: weight codes [N, K] and
 * channel scales); workspace per pq_qlinear_t_workspace_bytes(M, N, K).  With few tokens (where pq_qlinear_s8 would run its
 * weight-streaming kernel) the product is computed in the normal orientation and only STORED transposed. */
size_t pq_qlinear_t_workspace_bytes(int64_t M, int64_t N, int64_t K);
int32_t pq_qlinear_s8_t(const int8_t* a, int64_t lda, const float* a_scale, const int8_t* b, int64_t ldb,
                        const float* b_scale, const void* bias, void* yt, int64_t ldyt, int32_t out_dtype, int64_t M,
                        int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);

/* pq_qlinear_s8 on STACKED activation codes: K-slab s — the columns [s * k_per_slab, (s + 1) * k_per_slab) of the logical a[M, K] — is the row-major
 * block a + s * slab_stride with leading dimension lda (what an all-gather of the ranks' int8 column blocks [M, K / G] leaves: slab_stride = M * K / G).
 * An integer sum has no order: the result is pq_qlinear_s8's on the row-major matrix, bit for bit.  Three ways, in this order:
 *   (1) where pq_qlinear_s8 would run the fused split-K of the 256 x 256 tile (the Llama-70B `down` shard 4096 x 1024 x 28672: four K-slices) AND a workspace of
 *       the hand-over's size is passed AND the slices cover whole slabs (or a slab holds whole slices; >= 512 columns per slab): that kernel walks the slabs in
 *       place — its K-loop's activation cursor jumps at the slab boundaries (round 6);
 *   (2) where the planner picks a ring tile (128 x 128, 64 x 128, 64 x 64: the Llama-70B `o` shard, and the `down` shard when no workspace comes): its loaders walk
 *       the slabs in place — no layout pass, no workspace;
 *   (3) any other shape: ONE layout pass into `workspace` (M * K bytes read and written), then pq_qlinear_s8.
 * K % k_per_slab == 0.  Workspace (256-byte aligned): pq_qlinear_kslabs_workspace_bytes is the size that is ALWAYS enough (way 3's: it knows neither base nor
 * strides); pq_qlinear_kslabs_workspace_bytes_for decides on the very operands of the call (only their alignment is looked at, nothing is read): way 1's hand-over
 * slabs, 0 for way 2, way 3's otherwise.  A workspace smaller than way 1 needs is not an error: ways 2 / 3 follow. */
size_t pq_qlinear_kslabs_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t k_per_slab);
size_t pq_qlinear_kslabs_workspace_bytes_for(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t k_per_slab, const int8_t* b, int64_t ldb,
                                             int64_t M, int64_t N, int64_t K);
/* which way a call with these operands and a workspace of workspace_bytes takes: "in place: fused split-K x4", "in place: ring128", "layout pass", ... (static string) */
const char* pq_kslabs_way_name(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t k_per_slab, const int8_t* b, int64_t ldb,
                               int64_t M, int64_t N, int64_t K, size_t workspace_bytes);
int32_t pq_qlinear_s8_kslabs(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t k_per_slab, const float* a_scale, const int8_t* b, int64_t ldb,
                             const float* b_scale, const void* bias, void* y, int64_t ldy, int32_t out_dtype, int64_t M, int64_t N, int64_t K,
                             void* workspace, size_t workspace_bytes, void* stream);

/* qlinear.forward in ONE call: y[M,N] = qlinear(x[M,K]) with dynamic per-token quantisation of x (K1), the int8 MFMA GEMM
 * and the fused dequant epilogue, output dtype = input dtype.  Scratch (xq, xs, optional split-K slabs) is carved from
 * `workspace` (>= pq_qlinear_dyn_workspace_bytes(M,N,K), 256-byte aligned, caller-owned, reusable across calls on one
 * stream).  Same results, bit for bit, as pq_quant_rowwise followed by pq_qlinear_s8. */
size_t pq_qlinear_dyn_workspace_bytes(int64_t M, int64_t N, int64_t K);
int32_t pq_qlinear_dyn(const void* x, int32_t dtype, int64_t ld_x, const int8_t* w, int64_t ldw, const float* w_scale,
                       const void* bias, void* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Self-test hook: for n (x, s) fp32 bit-pattern pairs, counts in mismatches[0] the pairs whose division-free
 * code (K1/K2 hot path) differs from clamp(rne(x / s)) and in mismatches[1] the pairs whose rebuilt quotient
 * differs from the IEEE quotient.  mismatches[2] must be zeroed by the caller.  QSPEC Q4. */
int32_t pq_selftest_fast_quotient(const uint32_t* x_bits, const uint32_t* s_bits, int64_t n,
                                  unsigned long long* mismatches, void* stream);

/* Self-test hook: enumerates on the GPU the WHOLE domain of the one-step division-free encode that 16-bit inputs take (dtype 0 = bf16,
 * 1 = fp16): every amax bit pattern whose scale takes the fast path x every magnitude pattern <= amax x both signs.  counts[0] += pairs
 * checked, counts[1] += pairs whose code differs from clamp(rne(x / s)).  counts[2] must be zeroed by the caller.  QSPEC Q4. */
int32_t pq_selftest_half_encode(int32_t dtype, unsigned long long* counts, void* stream);

/* Self-test hook: every 16-bit pattern g with 0 < |g| <= 86 (the domain of the division-free silu of bf16 / fp16 rows) through the
 * one-correction division the producer kernel uses, the two-correction form and true division.  counts[0] += patterns, counts[1] +=
 * patterns whose stored silu(g) differs between the first two, counts[2] += between the last two.  counts[3] zeroed by the caller.  QSPEC S4. */
int32_t pq_selftest_silu_short(int32_t dtype, unsigned long long* counts, void* stream);

/* Name of the single-pass GEMM kernel variant the dispatcher would pick for this problem (static string); with a workspace
 * (pq_qlinear_workspace_bytes > 0) pq_qlinear_s8 runs a split-K form of the 256 x 256 (or 128 x 256) tile instead. */
const char* pq_gemm_variant_name(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb);

#ifdef __cplusplus
}
#endif
#endif /* PQ_HIP_H */
