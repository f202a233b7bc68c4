// pq_api.hip — the C-ABI of libpq_hip.so (include/pq_hip.h): argument validation, variant selection,
// launches, status codes.  No allocation, no synchronisation, no host copies: graph-capturable.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <atomic>
#include <mutex>

#include "gemm_epilogue.h"

namespace pq {
template <int DT> void quant_rowwise_dispatch(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template <int DT> hipError_t quant_colwise_dispatch(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template <int DT> void silu_mul_quant_dispatch(const void*, int64_t, const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);
template <int DT> void rmsnorm_quant_dispatch(const void*, int64_t, const void*, float, int64_t, int64_t, int8_t*, int64_t, float*, void*, int64_t, hipStream_t);
template <int ODT> void dequant_dispatch(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);
template <int OUT> void launch_gemm_generic(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template <int OUT, int TM, int TN> void launch_gemm_fast(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
template <int DT, int MODE, bool IDENT> void silu_mul_split_dispatch(const void*, int64_t, const void*, int64_t, int64_t, int64_t, uint32_t*, int8_t*, int64_t, float*, hipStream_t);
template <int OUT> void launch_gemm_ring128(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t a_slab_stride = 0, int64_t a_k_per_slab = 0);
template <int OUT> void launch_gemm_ringt(int, const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t, int64_t a_slab_stride = 0, int64_t a_k_per_slab = 0);
template <int OUT> void launch_gemm_skinny(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, hipStream_t);
bool gemm_fast_eligible(const int8_t*, int64_t, const int8_t*, int64_t, int64_t, int64_t, int64_t);
template <int TM> void launch_gemm_splitk_i32(const int8_t*, int64_t, const int8_t*, int64_t, int32_t*, int64_t, int64_t, int64_t, int, hipStream_t, int nxcd);
template <int OUT> void launch_splitk_reduce(const int32_t*, int, int64_t, int64_t, const EpiArgs&, hipStream_t);
template <int OUT> bool launch_gemm_fsk(const int8_t*, int64_t, const int8_t*, int64_t, const EpiArgs&, int64_t, int64_t, int64_t, int, void*, hipStream_t, int64_t a_slab_stride = 0, int64_t a_k_per_slab = 0);
bool fsk_kslabs_ok(int64_t K, int64_t k_per_slab, int kslices);
size_t fsk_workspace_bytes(int64_t, int64_t, int);
void set_stamp_buffer(unsigned long long*);
void launch_fast_quotient_check(const uint32_t*, const uint32_t*, int64_t, unsigned long long*, hipStream_t);
void launch_half_encode_check(int, unsigned long long*, hipStream_t);
void launch_silu_short_check(int, unsigned long long*, hipStream_t);
}  // namespace pq

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int32_t check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return PQ_OK;
}

enum Variant { V_AUTO = 0, V_GENERIC, V_SP256_16, V_SP128_16, V_SP128X128, V_RING128, V_SKINNY, V_RING64X128, V_RING64X64, V_RING128X160 };

Variant parse_variant(const char* e) {
    if (!e || !*e) return V_AUTO;
    if (!strcmp(e, "generic")) return V_GENERIC;
    if (!strcmp(e, "sp256_16")) return V_SP256_16;
    if (!strcmp(e, "sp128_16")) return V_SP128_16;
    if (!strcmp(e, "sp128x128")) return V_SP128X128;
    if (!strcmp(e, "ring128")) return V_RING128;
    if (!strcmp(e, "skinny")) return V_SKINNY;
    if (!strcmp(e, "ring64x128")) return V_RING64X128;
    if (!strcmp(e, "ring64x64")) return V_RING64X64;
    if (!strcmp(e, "ring128x160")) return V_RING128X160;
    return V_AUTO;
}

// Behaviour switches (tests / experiments): pq::Options (pq_common.h), one immutable snapshot per state.  The environment is read ONCE, at the
// first call into the library (a getenv per launch is host time on the hot path, and a switch that a captured hipGraph freezes must not look
// live); pq_set_option() publishes a modified copy by one atomic pointer swap.  Every C-ABI entry point pins the live snapshot for its own
// duration (CallScope): planning and launching happen under ONE set of switches, and concurrent calls from several host threads are defined.
std::atomic<const pq::Options*> g_live{nullptr};
std::mutex g_opt_mu;                       // writers only (pq_set_option, the environment pass)
std::once_flag g_opt_once;
thread_local const pq::Options* tl_opt = nullptr;      // the snapshot pinned by the C-ABI call in progress on this thread
thread_local int tl_depth = 0;

// every behaviour switch, by name: the ONE place both the environment pass (once) and pq_set_option go through
const char* const kOptionNames[] = {"PQ_FORCE_VARIANT", "PQ_NO_TAILSPLIT", "PQ_NO_SPLITK", "PQ_FORCE_SPLITK", "PQ_FSK", "PQ_FSK_SYMMETRIC", "PQ_FSK_FENCED", "PQ_FSK_COOP", "PQ_FAKE_CUS", "PQ_NO_MIDM", "PQ_NO_KSLABS", "PQ_NO_RING160", "PQ_MIDM_CT", "PQ_RMS_WAVE_MAX", "PQ_SILU_TPR", "PQ_SP128_LC",
                                    "PQ_SP256_P3", "PQ_SP256_ASM", "PQ_SP256_PERSIST", "PQ_RING_LC", "PQ_RING_ROT", "PQ_K1_LDS", "PQ_EPI_ANY_ALIGN", "PQ_K2_BLOCKS_A", "PQ_K2_BLOCKS_E", "PQ_K1_RPW", "PQ_K1_ST16", "PQ_SKINNY_RB", "PQ_SKINNY_STAGE", "PQ_SKINNY_KS"};
bool apply_option(pq::Options& o, const char* name, const char* value) {
    const bool set = value && *value;
    const int iv = set ? atoi(value) : 0;
    if (!strcmp(name, "PQ_FORCE_VARIANT")) o.variant = parse_variant(value);
    else if (!strcmp(name, "PQ_NO_TAILSPLIT")) o.no_tailsplit = set;
    else if (!strcmp(name, "PQ_NO_SPLITK")) o.no_splitk = set;
    else if (!strcmp(name, "PQ_FORCE_SPLITK")) o.force_splitk = iv;
    else if (!strcmp(name, "PQ_FSK")) o.fsk = set ? iv : -1;
    else if (!strcmp(name, "PQ_FSK_SYMMETRIC")) o.fsk_symmetric = set && *value == '1';
    else if (!strcmp(name, "PQ_FSK_FENCED")) o.fsk_fenced = set && *value == '1';
    else if (!strcmp(name, "PQ_FSK_COOP")) o.fsk_coop = set && *value == '1';
    else if (!strcmp(name, "PQ_FAKE_CUS")) o.fake_cus = iv > 0 ? iv : 0;
    else if (!strcmp(name, "PQ_NO_MIDM")) o.no_midm = set;
    else if (!strcmp(name, "PQ_NO_KSLABS")) o.no_kslabs = set;
    else if (!strcmp(name, "PQ_NO_RING160")) o.no_ring160 = set;
    else if (!strcmp(name, "PQ_MIDM_CT")) o.midm_ct = iv > 0 ? iv : 0;
    else if (!strcmp(name, "PQ_RMS_WAVE_MAX")) o.rms_wave_max = !set || iv < 0 ? 256 : (iv > 512 ? 512 : iv);
    else if (!strcmp(name, "PQ_SILU_TPR")) o.silu_tpr = set && !strcmp(value, "256") ? 256 : 0;
    else if (!strcmp(name, "PQ_SP128_LC")) o.sp128_lc = !set || iv < 0 || iv > 2 ? 1 : iv;
    else if (!strcmp(name, "PQ_SP256_P3")) o.sp256_p3 = !(set && *value == '0');
    else if (!strcmp(name, "PQ_SP256_ASM")) o.sp256_asm = !set || iv < 0 ? 1 : iv;     // "" = default (1)
    else if (!strcmp(name, "PQ_SP256_PERSIST")) o.sp256_persist = set && *value == '1';
    else if (!strcmp(name, "PQ_RING_LC")) o.ring_lc = !(set && *value == '0');
    else if (!strcmp(name, "PQ_RING_ROT")) o.ring_rot = !set ? 1 : (iv < 0 ? 0 : iv);
    else if (!strcmp(name, "PQ_K1_LDS")) o.k1_lds = iv < 0 ? 0 : (iv > 65536 ? 65536 : iv);
    else if (!strcmp(name, "PQ_EPI_ANY_ALIGN")) o.epi_any_align = !(set && *value == '0');
    else if (!strcmp(name, "PQ_K2_BLOCKS_A")) o.k2_blocks_a = iv > 0 ? iv : 0;
    else if (!strcmp(name, "PQ_K2_BLOCKS_E")) o.k2_blocks_e = iv > 0 ? iv : 0;
    else if (!strcmp(name, "PQ_K1_ST16")) o.k1_st16 = set && *value == '1';
    else if (!strcmp(name, "PQ_K1_RPW")) o.k1_rpw = set && *value == '2' ? 2 : (set && *value == '1' ? 1 : 0);
    else if (!strcmp(name, "PQ_SKINNY_STAGE")) o.skinny_stage = !(set && *value == '0');
    else if (!strcmp(name, "PQ_SKINNY_KS")) o.skinny_ks = iv > 0 ? iv : 0;
    else if (!strcmp(name, "PQ_SKINNY_RB")) o.skinny_rb = set && *value == '2' ? 2 : (set && *value == '1' ? 1 : 0);
    else return false;
    return true;
}
const pq::Options* live_options() {
    std::call_once(g_opt_once, [] {
        auto* o = new pq::Options();
        for (const char* n : kOptionNames)
            if (const char* v = getenv(n)) apply_option(*o, n, v);
        if (const char* r = getenv("PQ_ROCTX")) {
            // PQ_ROCTX=1: every C-ABI entry point pushes / pops a roctx range (quant / gemm / ...), so a rocprofv3 --marker-trace
            // timeline shows the path's stages by name.  The marker library is dlopen'ed on first use; absent library = no ranges.
            if (*r && *r != '0') {
                void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    o->roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                    o->roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                    if (!o->roctx_push || !o->roctx_pop) o->roctx_push = nullptr, o->roctx_pop = nullptr;
                }
            }
        }
        g_live.store(o, std::memory_order_release);
    });
    return g_live.load(std::memory_order_acquire);
}
const pq::Options& options() { return tl_opt ? *tl_opt : *live_options(); }

// pins the live snapshot for one C-ABI call on this thread (re-entrant: pq_qlinear_dyn calls pq_quant_rowwise and pq_qlinear_s8 under ITS snapshot)
struct CallScope {
    CallScope() { if (tl_depth++ == 0) tl_opt = live_options(); }
    ~CallScope() { if (--tl_depth == 0) tl_opt = nullptr; }
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
};

// CUs of the current device (hipDeviceGetAttribute, cached per device ordinal; a CU-masked or partitioned device reports fewer); PQ_FAKE_CUS overrides
int device_cus() {
    if (options().fake_cus > 0) return options().fake_cus;
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// XCDs (L2 domains) of the current device for the tile remaps (hipDeviceAttributeNumberOfXccs, cached; a partitioned device reports fewer); with PQ_FAKE_CUS: one per 32 CUs
int device_xcds() {
    if (options().fake_cus > 0) return options().fake_cus >= 32 ? options().fake_cus / 32 : 1;
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 8; }
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n <= 0 || n > 64) { (void)hipGetLastError(); n = 8; }
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
Variant forced_variant() { return static_cast<Variant>(options().variant); }

struct Range {      // one C-ABI call: pins the option snapshot (CallScope) and, with PQ_ROCTX, a roctx range (host side: it brackets the asynchronous launches)
    CallScope scope_;
    bool on;
    explicit Range(const char* name) : on(options().roctx_push != nullptr) { if (on) options().roctx_push(name); }
    ~Range() { if (on) options().roctx_pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

Variant pick_variant(const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    const bool ok = pq::gemm_fast_eligible(a, lda, b, ldb, M, N, K);
    const Variant f = forced_variant();
    if (f == V_GENERIC || !ok) return V_GENERIC;
    if (f == V_SKINNY) return M <= 64 ? V_SKINNY : V_RING128;      // the skinny kernel holds at most 4 token tiles
    if (f != V_AUTO) return f;
    // decode-like: stream the weights straight into MFMA fragments (HBM-bound).  One 16-token tile: always (4096x4096 6 us vs 18 us tiled); two token tiles while the tiled
    // grid cannot fill the chip (N <= 8192); three and four token tiles (33 .. 64 tokens) only against narrow matrices (N <= 4096: one round of 16-row blocks; a 5120-wide matrix is a round and a quarter — 64 x 5120 x 11008 43.7 us streaming, 30.7 on 64 x 64 ring tiles) — beyond, the
    // 64-row ring tiles of round 4 have enough tiles and win (HBM-fed: 64 x 6144 x 4096 17.9 -> 13.2 us, 64 x 28672 x 4096 46.7 -> 31.6; but 64 x 4096 x 4096 10.4 against
    // 12.7 and 64 x 4096 x 14336 26.3 against 36.3 stay here: profiles/r04_midm_decode.txt).  PQ_NO_MIDM=1: the round-3 split (<= 64 tokens, N <= 8192).
    // (round 4 audit, tools/dispatch_audit.py --small, two boxes: 17 .. 24 tokens against the widest matrices with K <= 4096 stay with the streaming kernel — 17 x 28672 x 4096
    // 32.2 us against 34.8 for the 64 x 128 ring tile)
    // ... and 9 .. 16 tokens against wide matrices with a LONG K (K >= 8192, N >= 14336: Llama-70B's gate / up at batch 16) go to the ring tiles: 16 x 28672 x 8192 55.5 - 57 us
    // against 59.7 streaming, 16 x 14336 x 8192 28.4 against 30.0 (8 tokens: the streaming kernel still wins at N = 28672, 49 against 55)
    const bool long_k_ring = !options().no_midm && M > 8 && M <= 16 && N >= 14336 && K >= 8192;
    if ((M <= 16 && !long_k_ring) || (M <= 32 && N <= 8192) || (M <= 64 && N < (options().no_midm ? 8193 : 4097)) || (!options().no_midm && M <= 24 && N >= 16384 && K <= 4096)) return V_SKINNY;
    if (M * N < 128 * 128) return V_GENERIC;   // a 256^2 tile would be mostly padding
    // 256x256 tiles unless they fill well under one round of the 256 CUs: then 128(m) x 256(n) tiles double the
    // blocks at ~3/4 of the per-CU rate (ingest-bound) — worth it when they keep everything in one round.
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), t128 = ((M + 127) / 128) * ((N + 255) / 256);
    const int64_t t128sq = ((M + 127) / 128) * ((N + 127) / 128);
    const int64_t cus = device_cus();          // one workgroup of these tiles per CU: every "fills the chip" threshold below is a share of the CUs THIS device reports (round 5; 256 on a whole MI355X)
    // even 128-row tiles fill at most half the chip: 128 x 128 tiles from a 4-deep DMA ring (gemm_s8_ring128; latency-bound
    // on operand ingest, ~2/3 of the 128 x 256 tile's rate per CU, but twice the blocks and no slab traffic).  Measured:
    // k/v 4096x1024x4096 31 -> 24 us, 70B q/o shard 47 (split-K) -> 40 us, 70B down shard 124 (split-K) -> 119 us.
    if (t128 <= cus / 2 && t128sq > t128) {
        // 64 < M <= 512 (gemm_s8_ring.hip): when the 128 x 128 ring tiles fill well under the chip, SMALLER tiles on every CU — the regime is bound by the L2 -> CU path
        // (profiles/r04_ablate_ring.txt), and a K split over workgroups costs more hand-over than it saves on launches this short (measured: profiles/r04_midm_fsk.txt).  Rounds of the 256 CUs x the measured time of one tile relative to the 128 x 128 ring tile (K = 4096: 14.0 / 13.4 / 9.2 us; profiles/r04_midm.txt);
        // PQ_NO_MIDM=1 restores the round-3 dispatch.
        if (!options().no_midm && options().force_splitk <= 1 && options().fsk <= 1) {      // (a forced slice count — experiments, tests — means the split-K forms)
            auto rounds = [cus](int64_t tiles) { return (double)((tiles + cus - 1) / cus); };
            const double c128 = rounds(t128sq) * 1.00, c64x128 = rounds(((M + 63) / 64) * ((N + 127) / 128)) * 0.95, c64x64 = rounds(((M + 63) / 64) * ((N + 63) / 64)) * 0.66;
            if (c64x64 < c128 && c64x64 <= c64x128) return V_RING64X64;
            if (c64x128 < c128) return V_RING64X128;
        }
        return V_RING128;
    }
    if (t256 <= cus * 5 / 8 && t128 > t256 && t128 <= cus) {
        // round 6: where the 128 x 256 tiles fill at most two thirds of the chip but 128 x 160 tiles make (almost) exactly one round of it — the Llama-3-70B fused-qkv shard
        // 4096 x 1280: 160 against 256 workgroups — the 128 x 160 ring tile (gemm_s8_ringt<128, 160>).  Measured over that class, weights from HBM, two boxes
        // (profiles/r06_dispatch_audit_tile160.txt, r06_ab_tile160.txt): 8 - 14 % ahead on 11 of 12 shapes with K >= 4096 (4096 x 1280 x 4096: +-3 %); everywhere outside the
        // class it loses 8 - 90 %.  PQ_NO_RING160=1 restores the round-5 choice.
        const int64_t t160 = ((M + 127) / 128) * ((N + 159) / 160);
        if (!options().no_ring160 && options().force_splitk <= 1 && options().fsk <= 1 &&      // (a forced slice count — experiments, tests — means the split-K forms, as for the mid-M tiles)
            K >= 4096 && t160 <= cus && t160 * 10 >= cus * 9 && t128 * 3 <= cus * 2) return V_RING128X160;
        return V_SP128_16;
    }
    return V_SP256_16;
}

template <int OUT>
void run_gemm(Variant v, const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb, const pq::EpiArgs& epi,
              int64_t M, int64_t N, int64_t K, hipStream_t st) {
    if (!pq::epi_flags_valid(epi.flags, epi.bias != nullptr)) abort();      // internal invariant (gemm_epilogue.h): never reachable from the C-ABI
    if (v == V_SP256_16) pq::launch_gemm_fast<OUT, 256, 256>(a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_SP128_16) pq::launch_gemm_fast<OUT, 128, 256>(a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_SP128X128) pq::launch_gemm_fast<OUT, 128, 128>(a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_RING128) pq::launch_gemm_ring128<OUT>(a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_SKINNY) pq::launch_gemm_skinny<OUT>(a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_RING64X128) pq::launch_gemm_ringt<OUT>(0, a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_RING64X64) pq::launch_gemm_ringt<OUT>(1, a, lda, b, ldb, epi, M, N, K, st);
    else if (v == V_RING128X160) pq::launch_gemm_ringt<OUT>(2, a, lda, b, ldb, epi, M, N, K, st);
    else pq::launch_gemm_generic<OUT>(a, lda, b, ldb, epi, M, N, K, st);
}

// Tail split: a grid of T > 256 tiles runs ceil(T/256) rounds of one 256x256 tile per CU, and the last round is as long as
// the others however few tiles it holds.  When that round is poorly filled, the trailing tile columns (or rows) go to a
// second launch of 128(m) x 256(n) tiles instead — twice the blocks, each ~0.65 of a full tile's time (measured) — so
// e.g. 344 tiles cost 1 + 0.65 rounds instead of 2.  Both launches are plain sub-problems (pointer offsets), results are
// unchanged bit for bit.  Returns the split axis (0 none, 1 along N, 2 along M) and the extent of the leading part.
constexpr double kHalfTileCost = 0.58, kSecondLaunchCost = 0.06;   // (round 2: the loader/consumer form of the 128-row tile: 29 vs 51 us per tile at K = 4096)

int tail_split_plan(int64_t M, int64_t N, int64_t* lead) {
    if (options().no_tailsplit) return 0;
    const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256, tiles = tm * tn, cus = device_cus();
    if (tiles <= cus) return 0;
    auto rounds = [cus](int64_t blocks) { return (double)((blocks + cus - 1) / cus); };
    double best = rounds(tiles) - 0.12;   // a split must save at least ~1/8 of a round to be worth a second launch
    int axis = 0;
    const int64_t hm = (M + 127) / 128;
    for (int64_t c = 1; c < tn; ++c) {    // trailing c tile columns, all rows, as 128-row tiles
        if (hm * c > 2 * cus) break;
        const double cost = rounds(tm * (tn - c)) + rounds(hm * c) * kHalfTileCost + kSecondLaunchCost;
        if (cost < best) { best = cost; axis = 1; *lead = (tn - c) * 256; }
    }
    for (int64_t r = 1; r < tm; ++r) {    // trailing r tile rows, all columns
        const int64_t tail_h = (M - (tm - r) * 256 + 127) / 128;
        if (tail_h * tn > 2 * cus) break;
        const double cost = rounds((tm - r) * tn) + rounds(tail_h * tn) * kHalfTileCost + kSecondLaunchCost;
        if (cost < best) { best = cost; axis = 2; *lead = (tm - r) * 256; }
    }
    return axis;
}

template <int OUT>
void run_gemm_auto(Variant v, const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb, const pq::EpiArgs& epi,
                   int64_t M, int64_t N, int64_t K, hipStream_t st) {
    int64_t lead = 0;
    const int axis = (v == V_SP256_16 && forced_variant() == V_AUTO) ? tail_split_plan(M, N, &lead) : 0;
    if (axis == 0) return run_gemm<OUT>(v, a, lda, b, ldb, epi, M, N, K, st);
    using elem_t = typename pq::OutElem<OUT>::type;
    pq::EpiArgs tail = epi;
    if (axis == 1) {
        tail.b_scale = epi.b_scale ? epi.b_scale + lead : nullptr;
        tail.bias = (epi.bias && !(epi.flags & pq::EPI_BIAS_ROWS)) ? static_cast<const elem_t*>(epi.bias) + lead : epi.bias;
        tail.y = static_cast<elem_t*>(epi.y) + lead;
        run_gemm<OUT>(V_SP256_16, a, lda, b, ldb, epi, M, lead, K, st);
        run_gemm<OUT>(V_SP128_16, a, lda, b + lead * ldb, ldb, tail, M, N - lead, K, st);
    } else {
        tail.a_scale = epi.a_scale ? epi.a_scale + lead : nullptr;
        tail.bias = (epi.bias && (epi.flags & pq::EPI_BIAS_ROWS)) ? static_cast<const elem_t*>(epi.bias) + lead : epi.bias;
        tail.y = static_cast<elem_t*>(epi.y) + lead * epi.ldy;
        run_gemm<OUT>(V_SP256_16, a, lda, b, ldb, epi, lead, N, K, st);
        run_gemm<OUT>(V_SP128_16, a + lead * lda, lda, b, ldb, tail, M - lead, N, K, st);
    }
}

bool bad_mat(const void* p, int64_t rows, int64_t cols, int64_t ld) {
    return rows < 0 || cols < 0 || ld < cols || (rows > 0 && cols > 0 && p == nullptr);
}

}  // namespace

// ---- qlinear on STACKED activation codes (what an all-gather of the ranks' int8 column blocks leaves: [G][M][K / G])
namespace {
typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
// stacked[s][m][c] (slab stride in bytes, row length lda) -> out[m][s * kps + c], 16 bytes at a time (kps % 16 == 0, aligned bases) or byte by byte
template <typename V>
__global__ __launch_bounds__(256) void unstack_kslabs_kernel(const uint8_t* __restrict__ a, int64_t lda, int64_t slab_stride, uint8_t* __restrict__ out, int64_t ldo,
                                                             int64_t M, int64_t kps_v, int nslabs) {
    const int64_t total = (int64_t)nslabs * M * kps_v;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % kps_v, m = (i / kps_v) % M, sl = i / (kps_v * M);
        *reinterpret_cast<V*>(out + m * ldo + (sl * kps_v + c) * (int64_t)sizeof(V)) = *reinterpret_cast<const V*>(a + sl * slab_stride + m * lda + c * (int64_t)sizeof(V));
    }
}
// the stacked operand can be walked in place by the loaders of the ring tiles (gemm_s8_ring128, gemm_s8_ringt: KSlabs) when one of them is what the planner picks for the
// shape and runs it single-pass; returns that variant, or V_GENERIC for "take the layout pass"
Variant kslabs_in_place(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t kps, const int8_t* b, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    if (options().no_kslabs || forced_variant() != V_AUTO || kps % 128 != 0 || K / 128 >= 65536 || (slab_stride & 15) != 0) return V_GENERIC;
    // (eligibility of the fast path is decided on the slab's own leading dimension and base; K itself is the whole K)
    const Variant v = pick_variant(a, lda, b, ldb, M, N, K);
    return (v == V_RING128 || v == V_RING64X128 || v == V_RING64X64 || v == V_RING128X160) ? v : V_GENERIC;
}
}  // namespace


namespace pq {
const Options& opt() { return options(); }
}  // namespace pq

extern "C" {

int32_t pq_version(void) { return PQ_ABI_VERSION; }

int32_t pq_set_option(const char* name, const char* value) {
    if (!name) return fail(PQ_ERR_BAD_ARG, "pq_set_option: null name");
    std::lock_guard<std::mutex> lock(g_opt_mu);
    auto* o = new pq::Options(*live_options());  // (the environment is consumed first, so a later call_once cannot undo this)
    if (!apply_option(*o, name, value)) { delete o; return fail(PQ_ERR_BAD_ARG, "pq_set_option: unknown option %s", name); }
    g_live.store(o, std::memory_order_release);  // calls already in flight keep the snapshot they pinned; the old one is never freed
    return PQ_OK;
}
const char* pq_last_error(void) { return g_err; }

int32_t pq_quant_rowwise(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x, int8_t* q,
                         int64_t ld_q, float* scale, void* stream) {
    Range range_("pq:quant_rowwise (K1)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_quant_rowwise: unknown dtype %d", dtype);
    if (bad_mat(x, rows, cols, ld_x) || bad_mat(q, rows, cols, ld_q) || (rows > 0 && !scale))
        return fail(PQ_ERR_BAD_ARG, "pq_quant_rowwise: bad matrix (rows=%lld cols=%lld ld_x=%lld ld_q=%lld)", (long long)rows, (long long)cols, (long long)ld_x, (long long)ld_q);
    if (rows == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PQ_BF16: pq::quant_rowwise_dispatch<PQ_BF16>(x, rows, cols, ld_x, q, ld_q, scale, st); break;
        case PQ_FP16: pq::quant_rowwise_dispatch<PQ_FP16>(x, rows, cols, ld_x, q, ld_q, scale, st); break;
        default: pq::quant_rowwise_dispatch<PQ_F32>(x, rows, cols, ld_x, q, ld_q, scale, st); break;
    }
    return check_launch("pq_quant_rowwise");
}

int32_t pq_quant_colwise(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x, int8_t* q,
                         int64_t ld_q, float* scale, void* stream) {
    Range range_("pq:quant_colwise (K2)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_quant_colwise: unknown dtype %d", dtype);
    if (bad_mat(x, rows, cols, ld_x) || bad_mat(q, rows, cols, ld_q) || (cols > 0 && !scale))
        return fail(PQ_ERR_BAD_ARG, "pq_quant_colwise: bad matrix (rows=%lld cols=%lld ld_x=%lld ld_q=%lld)", (long long)rows, (long long)cols, (long long)ld_x, (long long)ld_q);
    if (cols == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e;
    switch (dtype) {
        case PQ_BF16: e = pq::quant_colwise_dispatch<PQ_BF16>(x, rows, cols, ld_x, q, ld_q, scale, st); break;
        case PQ_FP16: e = pq::quant_colwise_dispatch<PQ_FP16>(x, rows, cols, ld_x, q, ld_q, scale, st); break;
        default: e = pq::quant_colwise_dispatch<PQ_F32>(x, rows, cols, ld_x, q, ld_q, scale, st); break;
    }
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "pq_quant_colwise: initialising the amax scratch: %s", hipGetErrorString(e));
    return check_launch("pq_quant_colwise");
}

int32_t pq_silu_mul_quant_rowwise(const void* g, int64_t ld_g, const void* u, int64_t ld_u, int32_t dtype, int64_t rows,
                                  int64_t cols, int8_t* q, int64_t ld_q, float* scale, void* h_out, int64_t ld_h, void* stream) {
    Range range_("pq:silu_mul_quant (K1s)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_silu_mul_quant_rowwise: unknown dtype %d", dtype);
    if (bad_mat(g, rows, cols, ld_g) || bad_mat(u, rows, cols, ld_u) || bad_mat(q, rows, cols, ld_q) || (rows > 0 && !scale) ||
        (h_out && ld_h < cols))
        return fail(PQ_ERR_BAD_ARG, "pq_silu_mul_quant_rowwise: bad matrix (rows=%lld cols=%lld ld_g=%lld ld_u=%lld ld_q=%lld ld_h=%lld)", (long long)rows, (long long)cols, (long long)ld_g, (long long)ld_u, (long long)ld_q, (long long)ld_h);
    if (rows == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PQ_BF16: pq::silu_mul_quant_dispatch<PQ_BF16>(g, ld_g, u, ld_u, rows, cols, q, ld_q, scale, h_out, ld_h, st); break;
        case PQ_FP16: pq::silu_mul_quant_dispatch<PQ_FP16>(g, ld_g, u, ld_u, rows, cols, q, ld_q, scale, h_out, ld_h, st); break;
        default: pq::silu_mul_quant_dispatch<PQ_F32>(g, ld_g, u, ld_u, rows, cols, q, ld_q, scale, h_out, ld_h, st); break;
    }
    return check_launch("pq_silu_mul_quant_rowwise");
}

int32_t pq_silu_mul_rowamax(const void* g, int64_t ld_g, const void* u, int64_t ld_u, int32_t dtype, int64_t rows, int64_t cols,
                            uint32_t* amax_bits, void* stream) {
    Range range_("pq:silu_mul_rowamax (K1s, amax half)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_silu_mul_rowamax: unknown dtype %d", dtype);
    if (bad_mat(g, rows, cols, ld_g) || bad_mat(u, rows, cols, ld_u) || (rows > 0 && !amax_bits))
        return fail(PQ_ERR_BAD_ARG, "pq_silu_mul_rowamax: bad matrix (rows=%lld cols=%lld ld_g=%lld ld_u=%lld)", (long long)rows, (long long)cols, (long long)ld_g, (long long)ld_u);
    if (rows == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PQ_BF16: pq::silu_mul_split_dispatch<PQ_BF16, 1, false>(g, ld_g, u, ld_u, rows, cols, amax_bits, nullptr, 0, nullptr, st); break;
        case PQ_FP16: pq::silu_mul_split_dispatch<PQ_FP16, 1, false>(g, ld_g, u, ld_u, rows, cols, amax_bits, nullptr, 0, nullptr, st); break;
        default: pq::silu_mul_split_dispatch<PQ_F32, 1, false>(g, ld_g, u, ld_u, rows, cols, amax_bits, nullptr, 0, nullptr, st); break;
    }
    return check_launch("pq_silu_mul_rowamax");
}

int32_t pq_silu_mul_quant_rowwise_amax(const void* g, int64_t ld_g, const void* u, int64_t ld_u, int32_t dtype, int64_t rows, int64_t cols,
                                       const uint32_t* amax_bits, int8_t* q, int64_t ld_q, float* scale, void* stream) {
    Range range_("pq:silu_mul_quant_amax (K1s, encode half)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_silu_mul_quant_rowwise_amax: unknown dtype %d", dtype);
    if (bad_mat(g, rows, cols, ld_g) || bad_mat(u, rows, cols, ld_u) || bad_mat(q, rows, cols, ld_q) || (rows > 0 && (!scale || !amax_bits)))
        return fail(PQ_ERR_BAD_ARG, "pq_silu_mul_quant_rowwise_amax: bad matrix (rows=%lld cols=%lld ld_g=%lld ld_u=%lld ld_q=%lld)", (long long)rows, (long long)cols, (long long)ld_g, (long long)ld_u, (long long)ld_q);
    if (rows == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t* ab = const_cast<uint32_t*>(amax_bits);          // (read-only in this mode)
    switch (dtype) {
        case PQ_BF16: pq::silu_mul_split_dispatch<PQ_BF16, 2, false>(g, ld_g, u, ld_u, rows, cols, ab, q, ld_q, scale, st); break;
        case PQ_FP16: pq::silu_mul_split_dispatch<PQ_FP16, 2, false>(g, ld_g, u, ld_u, rows, cols, ab, q, ld_q, scale, st); break;
        default: pq::silu_mul_split_dispatch<PQ_F32, 2, false>(g, ld_g, u, ld_u, rows, cols, ab, q, ld_q, scale, st); break;
    }
    return check_launch("pq_silu_mul_quant_rowwise_amax");
}

int32_t pq_quant_rowamax(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x, uint32_t* amax_bits, void* stream) {
    Range range_("pq:quant_rowamax (K1, amax half)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_quant_rowamax: unknown dtype %d", dtype);
    if (bad_mat(x, rows, cols, ld_x) || (rows > 0 && !amax_bits))
        return fail(PQ_ERR_BAD_ARG, "pq_quant_rowamax: bad matrix (rows=%lld cols=%lld ld_x=%lld)", (long long)rows, (long long)cols, (long long)ld_x);
    if (rows == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PQ_BF16: pq::silu_mul_split_dispatch<PQ_BF16, 1, true>(x, ld_x, nullptr, 0, rows, cols, amax_bits, nullptr, 0, nullptr, st); break;
        case PQ_FP16: pq::silu_mul_split_dispatch<PQ_FP16, 1, true>(x, ld_x, nullptr, 0, rows, cols, amax_bits, nullptr, 0, nullptr, st); break;
        default: pq::silu_mul_split_dispatch<PQ_F32, 1, true>(x, ld_x, nullptr, 0, rows, cols, amax_bits, nullptr, 0, nullptr, st); break;
    }
    return check_launch("pq_quant_rowamax");
}

int32_t pq_quant_rowwise_amax(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld_x, const uint32_t* amax_bits, int8_t* q, int64_t ld_q,
                              float* scale, void* stream) {
    Range range_("pq:quant_rowwise_amax (K1, encode half)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_quant_rowwise_amax: unknown dtype %d", dtype);
    if (bad_mat(x, rows, cols, ld_x) || bad_mat(q, rows, cols, ld_q) || (rows > 0 && (!scale || !amax_bits)))
        return fail(PQ_ERR_BAD_ARG, "pq_quant_rowwise_amax: bad matrix (rows=%lld cols=%lld ld_x=%lld ld_q=%lld)", (long long)rows, (long long)cols, (long long)ld_x, (long long)ld_q);
    if (rows == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t* ab = const_cast<uint32_t*>(amax_bits);          // (read-only in this mode)
    switch (dtype) {
        case PQ_BF16: pq::silu_mul_split_dispatch<PQ_BF16, 2, true>(x, ld_x, nullptr, 0, rows, cols, ab, q, ld_q, scale, st); break;
        case PQ_FP16: pq::silu_mul_split_dispatch<PQ_FP16, 2, true>(x, ld_x, nullptr, 0, rows, cols, ab, q, ld_q, scale, st); break;
        default: pq::silu_mul_split_dispatch<PQ_F32, 2, true>(x, ld_x, nullptr, 0, rows, cols, ab, q, ld_q, scale, st); break;
    }
    return check_launch("pq_quant_rowwise_amax");
}

int32_t pq_rmsnorm_quant_rowwise(const void* x, int64_t ld_x, const void* weight, float eps, int32_t dtype, int64_t rows, int64_t cols,
                                 int8_t* q, int64_t ld_q, float* scale, void* h_out, int64_t ld_h, void* stream) {
    Range range_("pq:rmsnorm_quant (K1n)");
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_rmsnorm_quant_rowwise: unknown dtype %d", dtype);
    if (bad_mat(x, rows, cols, ld_x) || bad_mat(q, rows, cols, ld_q) || (rows > 0 && !scale) || (rows > 0 && cols > 0 && !weight) ||
        (h_out && ld_h < cols) || cols >= (1 << 24) || !(eps >= 0.0f))
        return fail(PQ_ERR_BAD_ARG, "pq_rmsnorm_quant_rowwise: bad arguments (rows=%lld cols=%lld ld_x=%lld ld_q=%lld ld_h=%lld eps=%g)", (long long)rows, (long long)cols, (long long)ld_x, (long long)ld_q, (long long)ld_h, (double)eps);
    if (rows == 0) return PQ_OK;
    if (cols == 0) {       // nothing to normalise: scale = 1 (QSPEC Q3), no codes
        hipStream_t st0 = static_cast<hipStream_t>(stream);
        pq::quant_rowwise_dispatch<PQ_F32>(x, rows, 0, 1, q, 1, scale, st0);
        return check_launch("pq_rmsnorm_quant_rowwise");
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PQ_BF16: pq::rmsnorm_quant_dispatch<PQ_BF16>(x, ld_x, weight, eps, rows, cols, q, ld_q, scale, h_out, ld_h, st); break;
        case PQ_FP16: pq::rmsnorm_quant_dispatch<PQ_FP16>(x, ld_x, weight, eps, rows, cols, q, ld_q, scale, h_out, ld_h, st); break;
        default: pq::rmsnorm_quant_dispatch<PQ_F32>(x, ld_x, weight, eps, rows, cols, q, ld_q, scale, h_out, ld_h, st); break;
    }
    return check_launch("pq_rmsnorm_quant_rowwise");
}

int32_t pq_dequant(const int8_t* q, int64_t ld_q, const float* scale, int32_t axis, int64_t rows, int64_t cols,
                   void* out, int64_t ld_out, int32_t out_dtype, void* stream) {
    Range range_("pq:dequant");
    if (out_dtype < 0 || out_dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_dequant: unknown dtype %d", out_dtype);
    if (axis != 0 && axis != 1) return fail(PQ_ERR_BAD_ARG, "pq_dequant: axis must be 0 or 1, got %d", axis);
    if (bad_mat(q, rows, cols, ld_q) || bad_mat(out, rows, cols, ld_out) || (rows > 0 && cols > 0 && !scale))
        return fail(PQ_ERR_BAD_ARG, "pq_dequant: bad matrix (rows=%lld cols=%lld)", (long long)rows, (long long)cols);
    if (rows == 0 || cols == 0) return PQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (out_dtype) {
        case PQ_BF16: pq::dequant_dispatch<PQ_BF16>(q, ld_q, scale, axis, rows, cols, out, ld_out, st); break;
        case PQ_FP16: pq::dequant_dispatch<PQ_FP16>(q, ld_q, scale, axis, rows, cols, out, ld_out, st); break;
        default: pq::dequant_dispatch<PQ_F32>(q, ld_q, scale, axis, rows, cols, out, ld_out, st); break;
    }
    return check_launch("pq_dequant");
}

int32_t pq_gemm_s8s8s32(const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb, int32_t* c, int64_t ldc,
                        int64_t M, int64_t N, int64_t K, void* stream) {
    Range range_("pq:gemm_s8s8s32 (K3)");
    if (M < 0 || N < 0 || K < 0 || bad_mat(a, M, K, lda) || bad_mat(b, N, K, ldb) || bad_mat(c, M, N, ldc))
        return fail(PQ_ERR_BAD_ARG, "pq_gemm_s8s8s32: bad arguments (M=%lld N=%lld K=%lld lda=%lld ldb=%lld ldc=%lld)", (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldb, (long long)ldc);
    if (M == 0 || N == 0) return PQ_OK;
    pq::EpiArgs epi{nullptr, nullptr, nullptr, c, ldc, 0};
    epi.nxcd = device_xcds();
    run_gemm_auto<pq::OUT_I32>(pick_variant(a, lda, b, ldb, M, N, K), a, lda, b, ldb, epi, M, N, K, static_cast<hipStream_t>(stream));
    return check_launch("pq_gemm_s8s8s32");
}

// split-K plan: how many K-slices (1 = none) and which tile height.  Only when the tile grid fills at most half of the
// 256 CUs even with 128-row tiles, K is long enough to amortise the extra pass, and the slices stay multiples of 128.
static int splitk_plan(int64_t M, int64_t N, int64_t K, int* tm_out) {
    if (options().no_splitk) return 1;
    if (options().force_splitk > 1 && M > 64 && K % (128 * options().force_splitk) == 0) { *tm_out = 256; return options().force_splitk; }   // (experiments)
    if (M <= 64 || N < 1 || K < 2048) return 1;    // (M <= 64: the skinny kernel splits K inside the workgroup)
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), t128 = ((M + 127) / 128) * ((N + 255) / 256);
    // (round 3 split the quarter-filled 256 x 256 grid with a very long K — the Llama-70B `down` shard, 4096 x 1024 x 28672 — four ways here, 99 us against 128 for the
    // ring tile fed from HBM; since round 4 the ring tile's loaders rotate their K walk and it runs 102 us from HBM in ONE launch without a workspace: profiles/r04_rotation.txt)
    const int64_t cus = device_cus();
    const int tm = (t256 <= cus * 5 / 8 && t128 > t256 && t128 <= cus) ? 128 : 256;
    const int64_t tiles = tm == 128 ? t128 : t256;
    if (tiles > cus / 2) return 1;
    // the slab reduction costs ~15 us, and the single-pass alternative for these grids is the 128 x 128 ring tile: split-K
    // only pays when the grid fills at most a quarter of the chip and K is long (measured: 4096x512x8192 32 vs 35 us,
    // 1024x1024x8192 25 vs 34 us; at half-filled grids the ring tile wins at every K)
    if (tiles > cus / 4 || K < 8192) return 1;
    int s = (tiles <= cus / 8 && K >= 12288) ? 8 : 4;       // (128 x 4096 x 14336: 40 us with 4 slices, 28 us with 8)
    while (s > 1 && (K % (128 * s) != 0 || K / s < 1024)) s >>= 1;
    *tm_out = tm;
    return s;
}

// fused split-K (gemm_s8_sp256<..., FSK>: the partial sums of a tile's K-slices are handed over inside the GEMM kernel).  Planned for the half-filled 256 x 256
// grid with a long K (cfg-3 `down`, 2048 x 4096 x 11008): two workgroups per tile, TICKET hand-over (placement-independent: pq_hip.h).  Measured, weights from HBM — what a
// layer inside a model sees (profiles/r04_ab_fsk_forms.txt, r04_rotation.txt): 78.3 us against 86.0 for the 128 x 256 tile (79.2 with its rotated K walk); the symmetric
// exchange of round 3, 76.8, is opt-in (PQ_FSK_SYMMETRIC=1).  PQ_FSK=0 turns the plan off, PQ_FSK=S (experiments) forces S slices wherever the shape admits them.
static int fsk_plan(int64_t M, int64_t N, int64_t K) {
    const int f = options().fsk;
    if (f == 0 || options().no_splitk || options().force_splitk > 1 || M <= 64 || f > 8) return 0;
    // (experiments: any grid in the ticket form, which never waits for a workgroup that is not running; the symmetric forms only when every workgroup is resident)
    // (the ticket form deals the K-tiles unevenly where they do not divide — round 6; the symmetric forms keep equal slices)
    if (f > 1) return (K % 128 == 0 && (K / 128) / f >= 5 && (!options().fsk_symmetric || (K % (128 * f) == 0 && f * (((M + 255) / 256) * ((N + 255) / 256)) <= device_cus()))) ? f : 0;
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
    // residency guard: the slices of a tile hand over inside the kernel, one workgroup per CU (160 KiB of LDS): plan it only when the whole grid fits the CUs
    // this device reports (a CU-masked or partitioned device reports fewer) — otherwise the two-pass split-K or the single-pass tile runs
    const int cus = device_cus();
    if (t256 > cus / 4 && t256 <= cus / 2 && K >= 10240 && K % 256 == 0) return 2;          // (2 * t256 <= cus: the whole grid resident)
    // the quarter-filled grid with a very long K — the Llama-70B `down` shard, 4096 x 1024 x 28672 — in four slices (4 * t256 <= cus).  Round 3 ran it with the symmetric
    // exchange (91 us), round 4 dropped the plan when the symmetric form became opt-in (ticket 104 us against 102 for the rotated 128 x 128 ring tile in one pass).
    // Round 5, three runs on two boxes, weights from HBM (profiles/r05_ab_down_shard_forms.txt): ticket 99.1 - 99.5 us, ring tile 106.0 - 114.6, symmetric 93.4 — the
    // placement-independent ticket form is 6 - 13 % ahead, so it is planned again (with the caller's workspace; without one the ring tile runs).  Smaller grids
    // (2048 x 1024 x 28672: 85 us either way) stay with the ring tiles.
    // (the K threshold from tools/dispatch_audit.py on the round-5 build, profiles/r05_dispatch_audit_longk.txt: at K = 16384 the four slices LOSE 10 - 15 % to the ring
    // tile on every such grid — 4096 x 1024 x 16384 68.8 against 62.8 us — at K = 28672 they win 2 - 6 %)
    if (t256 > cus / 8 && t256 <= cus / 4 && K >= 24576 && K % 512 == 0) return 4;
    return 0;
}
// ... and (round 6) by the TICKET form of the fused split-K of the 256 x 256 tile, where that is what the planner runs on the row-major operand (the Llama-70B `down` shard,
// 4096 x 1024 x 28672: four slices over eight slabs, 99 us against 106 - 115 for the ring tile: profiles/r05_ab_down_shard_forms.txt): each slice's K range covers whole
// slabs (or sits inside one) and the asm K-loop's activation cursor jumps at the slab boundaries (gemm_s8_sp256<..., KSL>).  Returns the slice count, 0 = not this way.
static int kslabs_fsk_in_place(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t kps, const int8_t* b, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    if (options().no_kslabs || forced_variant() != V_AUTO || (slab_stride & 15) != 0 || slab_stride >= ((int64_t)1 << 40)) return 0;
    const Variant v = pick_variant(a, lda, b, ldb, M, N, K);
    if (!(v == V_SP256_16 || v == V_SP128_16 || v == V_RING128)) return 0;        // (qlinear_core's `tiled`)
    const int f = fsk_plan(M, N, K);
    return (f > 1 && pq::fsk_kslabs_ok(K, kps, f)) ? f : 0;
}

size_t pq_qlinear_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    CallScope scope_;
    {   // (alignment of the operands is unknown here: assume the fast path, as pq_gemm_variant_name does)
        const Variant v = pick_variant(reinterpret_cast<const int8_t*>(16), K, reinterpret_cast<const int8_t*>(16), K, M, N, K);
        if (v == V_RING64X128 || v == V_RING64X64 || v == V_RING128X160) return 0;      // the mid-M tiles and the 128 x 160 tile run single-pass
    }
    if (const int f = fsk_plan(M, N, K)) return pq::fsk_workspace_bytes(M, N, f);
    int tm = 256;
    const int s = splitk_plan(M, N, K, &tm);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(int32_t) : 0;
}

// the GEMM + epilogue of one (M x N x K, A rows x B rows) problem whose EpiArgs are already in the kernel's orientation
static int32_t qlinear_core(const char* what, const int8_t* a, int64_t lda, const int8_t* b, int64_t ldb, const pq::EpiArgs& epi_in,
                            int32_t out_dtype, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
    Range range_("pq:qlinear_s8 (K3+K4)");
    pq::EpiArgs epi = epi_in;
    epi.y_any_align = options().epi_any_align ? 1 : 0;
    epi.nxcd = device_xcds();
    const Variant v = pick_variant(a, lda, b, ldb, M, N, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // split-K needs the caller's workspace (pq_qlinear_workspace_bytes); without it the single-pass path runs.
    const bool tiled = (v == V_SP256_16 || v == V_SP128_16 || v == V_RING128) && forced_variant() == V_AUTO;
    if (const int f = tiled ? fsk_plan(M, N, K) : 0; f > 1 && workspace != nullptr) {
        const size_t need = pq::fsk_workspace_bytes(M, N, f);
        if (workspace_bytes < need) return fail(PQ_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, workspace_bytes, need);
        if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return fail(PQ_ERR_BAD_ALIGN, "%s: workspace must be 16-byte aligned", what);
        if (!pq::epi_flags_valid(epi.flags, epi.bias != nullptr)) abort();
        bool launched;
        switch (out_dtype) {
            case PQ_BF16: launched = pq::launch_gemm_fsk<PQ_BF16>(a, lda, b, ldb, epi, M, N, K, f, workspace, st); break;
            case PQ_FP16: launched = pq::launch_gemm_fsk<PQ_FP16>(a, lda, b, ldb, epi, M, N, K, f, workspace, st); break;
            default: launched = pq::launch_gemm_fsk<PQ_F32>(a, lda, b, ldb, epi, M, N, K, f, workspace, st); break;
        }
        if (!launched) { (void)hipGetLastError(); return fail(PQ_ERR_WORKSPACE, "%s: the split-K workspace could not be initialised (the zeroing launch failed)", what); }
        return check_launch(what);
    }
    int tm = 256;
    const int ks = tiled && fsk_plan(M, N, K) == 0 ? splitk_plan(M, N, K, &tm) : 1;
    if (ks > 1 && workspace != nullptr) {
        const size_t need = (size_t)ks * (size_t)M * (size_t)N * sizeof(int32_t);
        if (workspace_bytes < need) return fail(PQ_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, workspace_bytes, need);
        if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return fail(PQ_ERR_BAD_ALIGN, "%s: workspace must be 16-byte aligned", what);
        int32_t* slabs = static_cast<int32_t*>(workspace);
        if (tm == 128) pq::launch_gemm_splitk_i32<128>(a, lda, b, ldb, slabs, M, N, K, ks, st, epi.nxcd);
        else pq::launch_gemm_splitk_i32<256>(a, lda, b, ldb, slabs, M, N, K, ks, st, epi.nxcd);
        switch (out_dtype) {
            case PQ_BF16: pq::launch_splitk_reduce<PQ_BF16>(slabs, ks, M, N, epi, st); break;
            case PQ_FP16: pq::launch_splitk_reduce<PQ_FP16>(slabs, ks, M, N, epi, st); break;
            default: pq::launch_splitk_reduce<PQ_F32>(slabs, ks, M, N, epi, st); break;
        }
        return check_launch(what);
    }
    switch (out_dtype) {
        case PQ_BF16: run_gemm_auto<PQ_BF16>(v, a, lda, b, ldb, epi, M, N, K, st); break;
        case PQ_FP16: run_gemm_auto<PQ_FP16>(v, a, lda, b, ldb, epi, M, N, K, st); break;
        default: run_gemm_auto<PQ_F32>(v, a, lda, b, ldb, epi, M, N, K, st); break;
    }
    return check_launch(what);
}

int32_t pq_qlinear_s8(const int8_t* a, int64_t lda, const float* a_scale, const int8_t* b, int64_t ldb,
                      const float* b_scale, const void* bias, void* y, int64_t ldy, int32_t out_dtype, int64_t M,
                      int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
    if (out_dtype < 0 || out_dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8: unknown dtype %d", out_dtype);
    if (M < 0 || N < 0 || K < 0 || bad_mat(a, M, K, lda) || bad_mat(b, N, K, ldb) || bad_mat(y, M, N, ldy) ||
        (M > 0 && !a_scale) || (N > 0 && !b_scale))
        return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8: bad arguments (M=%lld N=%lld K=%lld lda=%lld ldb=%lld ldy=%lld)", (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldb, (long long)ldy);
    if (M == 0 || N == 0) return PQ_OK;
    pq::EpiArgs epi{a_scale, b_scale, bias, y, ldy, 0};
    return qlinear_core("pq_qlinear_s8", a, lda, b, ldb, epi, out_dtype, M, N, K, workspace, workspace_bytes, stream);
}

// y^T[N, M] of the same qlinear: the kernel's rows are the weight rows (n), its columns the tokens (m); the epilogue applies
// the token (column) scale first and runs the bias along rows, so every value has the bits of pq_qlinear_s8's y[m][n].
size_t pq_qlinear_t_workspace_bytes(int64_t M, int64_t N, int64_t K) { return pq_qlinear_workspace_bytes(N, M, K); }

int32_t pq_qlinear_s8_t(const int8_t* a, int64_t lda, const float* a_scale, const int8_t* b, int64_t ldb,
                        const float* b_scale, const void* bias, void* yt, int64_t ldyt, int32_t out_dtype, int64_t M,
                        int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
    if (out_dtype < 0 || out_dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8_t: unknown dtype %d", out_dtype);
    if (M < 0 || N < 0 || K < 0 || bad_mat(a, M, K, lda) || bad_mat(b, N, K, ldb) || bad_mat(yt, N, M, ldyt) ||
        (M > 0 && !a_scale) || (N > 0 && !b_scale))
        return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8_t: bad arguments (M=%lld N=%lld K=%lld lda=%lld ldb=%lld ldyt=%lld)", (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldb, (long long)ldyt);
    if (M == 0 || N == 0) return PQ_OK;
    // the workspace arguments are validated the same way on every path (the streaming path below does not use them; error behaviour must not depend on M)
    if (workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return fail(PQ_ERR_BAD_ALIGN, "pq_qlinear_s8_t: workspace must be 16-byte aligned");
    if (workspace == nullptr && workspace_bytes != 0) return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8_t: workspace_bytes without a workspace");
    {   // few tokens: the swapped form below would be a 16-column problem for the tile kernels (16 x 4096 x 4096: 23 us); the weight-streaming kernel computes the
        // product in its normal orientation and stores it transposed (5.8 us).  Same arithmetic as pq_qlinear_s8: same bits.
        CallScope scope_;
        const Variant f = forced_variant();
        if ((f == V_AUTO || f == V_SKINNY) && pick_variant(a, lda, b, ldb, M, N, K) == V_SKINNY) {
            Range range_("pq:qlinear_s8_t (K3+K4, streaming)");         // (named only where the streaming kernel really runs: marker traces)
            pq::EpiArgs e{a_scale, b_scale, bias, yt, ldyt, pq::EPI_STORE_T};
            e.nxcd = device_xcds();
            hipStream_t st = static_cast<hipStream_t>(stream);
            switch (out_dtype) {
                case PQ_BF16: run_gemm<PQ_BF16>(V_SKINNY, a, lda, b, ldb, e, M, N, K, st); break;
                case PQ_FP16: run_gemm<PQ_FP16>(V_SKINNY, a, lda, b, ldb, e, M, N, K, st); break;
                default: run_gemm<PQ_F32>(V_SKINNY, a, lda, b, ldb, e, M, N, K, st); break;
            }
            return check_launch("pq_qlinear_s8_t");
        }
    }
    pq::EpiArgs epi{b_scale, a_scale, bias, yt, ldyt, pq::EPI_COL_FIRST | (bias ? pq::EPI_BIAS_ROWS : 0)};
    return qlinear_core("pq_qlinear_s8_t", b, ldb, a, lda, epi, out_dtype, N, M, K, workspace, workspace_bytes, stream);
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- qlinear on STACKED activation codes (helpers above, before the C-ABI block)
// (the short form knows neither the operand's base nor its strides: it returns what the layout pass needs — always enough, whichever way the call goes)
size_t pq_qlinear_kslabs_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t k_per_slab) {
    CallScope scope_;
    if (M <= 0 || N <= 0 || K <= 0 || k_per_slab <= 0 || K % k_per_slab != 0) return 0;
    if (K == k_per_slab) return pq_qlinear_workspace_bytes(M, N, K);
    return align256((size_t)M * (size_t)K) + pq_qlinear_workspace_bytes(M, N, K);
}
// the exact form: decides on the operand the call will see (only the ALIGNMENT of `a` and `b` is looked at, nothing is read)
size_t pq_qlinear_kslabs_workspace_bytes_for(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t k_per_slab, const int8_t* b, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    CallScope scope_;
    if (M <= 0 || N <= 0 || K <= 0 || k_per_slab <= 0 || K % k_per_slab != 0) return 0;
    if (K == k_per_slab) return pq_qlinear_workspace_bytes(M, N, K);
    if (const int f = kslabs_fsk_in_place(a, lda, slab_stride, k_per_slab, b, ldb, M, N, K)) return pq::fsk_workspace_bytes(M, N, f);
    if (kslabs_in_place(a, lda, slab_stride, k_per_slab, b, ldb, M, N, K) != V_GENERIC) return 0;
    return align256((size_t)M * (size_t)K) + pq_qlinear_workspace_bytes(M, N, K);
}

// which of the three ways (pq_hip.h) a pq_qlinear_s8_kslabs call with these operands and a workspace of `workspace_bytes` takes (dispatch audits, tests)
const char* pq_kslabs_way_name(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t k_per_slab, const int8_t* b, int64_t ldb, int64_t M, int64_t N, int64_t K,
                               size_t workspace_bytes) {
    CallScope scope_;
    if (M <= 0 || N <= 0 || K <= 0 || k_per_slab <= 0 || K % k_per_slab != 0) return "invalid";
    if (K == k_per_slab) return "one slab: pq_qlinear_s8";
    if (const int f = kslabs_fsk_in_place(a, lda, slab_stride, k_per_slab, b, ldb, M, N, K); f > 1 && workspace_bytes >= pq::fsk_workspace_bytes(M, N, f))
        return f == 2 ? "in place: fused split-K x2" : f == 4 ? "in place: fused split-K x4" : f == 8 ? "in place: fused split-K x8" : "in place: fused split-K";
    switch (kslabs_in_place(a, lda, slab_stride, k_per_slab, b, ldb, M, N, K)) {
        case V_RING128: return "in place: ring128";
        case V_RING64X128: return "in place: ring64x128";
        case V_RING64X64: return "in place: ring64x64";
        case V_RING128X160: return "in place: ring128x160";
        default: return "layout pass";
    }
}

int32_t pq_qlinear_s8_kslabs(const int8_t* a, int64_t lda, int64_t slab_stride, int64_t k_per_slab, const float* a_scale, const int8_t* b, int64_t ldb,
                             const float* b_scale, const void* bias, void* y, int64_t ldy, int32_t out_dtype, int64_t M, int64_t N, int64_t K,
                             void* workspace, size_t workspace_bytes, void* stream) {
    CallScope scope_;
    if (out_dtype < 0 || out_dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8_kslabs: unknown dtype %d", out_dtype);
    if (M < 0 || N < 0 || K < 0 || k_per_slab <= 0 || K % k_per_slab != 0 || bad_mat(a, M, k_per_slab, lda) || bad_mat(b, N, K, ldb) || bad_mat(y, M, N, ldy) ||
        (M > 0 && !a_scale) || (N > 0 && !b_scale) || (K > k_per_slab && slab_stride < (M - 1) * lda + k_per_slab))
        return fail(PQ_ERR_BAD_ARG, "pq_qlinear_s8_kslabs: bad arguments (M=%lld N=%lld K=%lld k_per_slab=%lld lda=%lld slab_stride=%lld ldb=%lld ldy=%lld)", (long long)M, (long long)N,
                    (long long)K, (long long)k_per_slab, (long long)lda, (long long)slab_stride, (long long)ldb, (long long)ldy);
    if (M == 0 || N == 0) return PQ_OK;
    if (K == k_per_slab) return pq_qlinear_s8(a, lda, a_scale, b, ldb, b_scale, bias, y, ldy, out_dtype, M, N, K, workspace, workspace_bytes, stream);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (const int f = kslabs_fsk_in_place(a, lda, slab_stride, k_per_slab, b, ldb, M, N, K);
        f > 1 && workspace != nullptr && workspace_bytes >= pq::fsk_workspace_bytes(M, N, f) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
        // (a workspace too small for the hand-over is not an error here: the ring tile below walks the slabs without one)
        Range range_("pq:qlinear_s8_kslabs (K3+K4, fused split-K, slabs walked in place)");
        pq::EpiArgs epi{a_scale, b_scale, bias, y, ldy, 0};
        epi.y_any_align = options().epi_any_align ? 1 : 0;
        epi.nxcd = device_xcds();
        if (!pq::epi_flags_valid(epi.flags, epi.bias != nullptr)) abort();
        bool launched;
        switch (out_dtype) {
            case PQ_BF16: launched = pq::launch_gemm_fsk<PQ_BF16>(a, lda, b, ldb, epi, M, N, K, f, workspace, st, slab_stride, k_per_slab); break;
            case PQ_FP16: launched = pq::launch_gemm_fsk<PQ_FP16>(a, lda, b, ldb, epi, M, N, K, f, workspace, st, slab_stride, k_per_slab); break;
            default: launched = pq::launch_gemm_fsk<PQ_F32>(a, lda, b, ldb, epi, M, N, K, f, workspace, st, slab_stride, k_per_slab); break;
        }
        if (!launched) { (void)hipGetLastError(); return fail(PQ_ERR_WORKSPACE, "pq_qlinear_s8_kslabs: the split-K workspace could not be initialised (the zeroing launch failed)"); }
        return check_launch("pq_qlinear_s8_kslabs");
    }
    if (const Variant v = kslabs_in_place(a, lda, slab_stride, k_per_slab, b, ldb, M, N, K); v != V_GENERIC) {
        Range range_("pq:qlinear_s8_kslabs (K3+K4, slabs walked in place)");
        pq::EpiArgs epi{a_scale, b_scale, bias, y, ldy, 0};
        epi.y_any_align = options().epi_any_align ? 1 : 0;
        epi.nxcd = device_xcds();
        if (!pq::epi_flags_valid(epi.flags, epi.bias != nullptr)) abort();
        auto go = [&](auto oc) {
            constexpr int OUT = decltype(oc)::value;
            if (v == V_RING128) pq::launch_gemm_ring128<OUT>(a, lda, b, ldb, epi, M, N, K, st, slab_stride, k_per_slab);
            else pq::launch_gemm_ringt<OUT>(v == V_RING64X128 ? 0 : (v == V_RING64X64 ? 1 : 2), a, lda, b, ldb, epi, M, N, K, st, slab_stride, k_per_slab);
        };
        switch (out_dtype) {
            case PQ_BF16: go(std::integral_constant<int, PQ_BF16>{}); break;
            case PQ_FP16: go(std::integral_constant<int, PQ_FP16>{}); break;
            default: go(std::integral_constant<int, PQ_F32>{}); break;
        }
        return check_launch("pq_qlinear_s8_kslabs");
    }
    // any other shape: one layout pass into the caller's workspace (reads + writes M * K bytes), then the planner's own choice on row-major codes
    const size_t need_a = align256((size_t)M * (size_t)K), need = need_a + pq_qlinear_workspace_bytes(M, N, K);
    if (!workspace || workspace_bytes < need) return fail(PQ_ERR_WORKSPACE, "pq_qlinear_s8_kslabs: workspace %zu < %zu bytes", workspace ? workspace_bytes : (size_t)0, need);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(PQ_ERR_BAD_ALIGN, "pq_qlinear_s8_kslabs: workspace must be 256-byte aligned");
    uint8_t* flat = static_cast<uint8_t*>(workspace);
    {
        Range range_("pq:unstack_kslabs");
        const int nslabs = (int)(K / k_per_slab);
        const bool vec = (k_per_slab % 16 == 0) && (lda % 16 == 0) && (slab_stride % 16 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
        const int64_t units = (int64_t)M * K / (vec ? 16 : 1);
        int64_t blocks = (units + 255) / 256;
        blocks = blocks > 256 * 16 ? 256 * 16 : (blocks < 1 ? 1 : blocks);
        if (vec) unstack_kslabs_kernel<v4u_><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(reinterpret_cast<const uint8_t*>(a), lda, slab_stride, flat, K, M, k_per_slab / 16, nslabs);
        else unstack_kslabs_kernel<uint8_t><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(reinterpret_cast<const uint8_t*>(a), lda, slab_stride, flat, K, M, k_per_slab, nslabs);
        const int32_t rc = check_launch("pq_qlinear_s8_kslabs (layout pass)");
        if (rc != PQ_OK) return rc;
    }
    const size_t rest = workspace_bytes - need_a;
    return pq_qlinear_s8(reinterpret_cast<const int8_t*>(flat), K, a_scale, b, ldb, b_scale, bias, y, ldy, out_dtype, M, N, K, rest ? flat + need_a : nullptr, rest, stream);
}

// ---- one-call dynamic qlinear: K1 (x -> xq, xs in the workspace) then pq_qlinear_s8 (with split-K slabs if planned).

size_t pq_qlinear_dyn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    CallScope scope_;
    if (M < 0 || N < 0 || K < 0) return 0;
    return align256((size_t)M * (size_t)K) + align256((size_t)M * sizeof(float)) + pq_qlinear_workspace_bytes(M, N, K);
}

int32_t pq_qlinear_dyn(const void* x, int32_t dtype, int64_t ld_x, const int8_t* w, int64_t ldw, const float* w_scale,
                       const void* bias, void* y, int64_t ldy, int64_t M, int64_t N, int64_t K, void* workspace,
                       size_t workspace_bytes, void* stream) {
    CallScope scope_;           // K1 and the GEMM of this call plan and launch under ONE snapshot
    if (dtype < 0 || dtype > 2) return fail(PQ_ERR_BAD_ARG, "pq_qlinear_dyn: unknown dtype %d", dtype);
    if (M < 0 || N < 0 || K < 0) return fail(PQ_ERR_BAD_ARG, "pq_qlinear_dyn: negative size");
    if (M == 0 || N == 0) return PQ_OK;
    const size_t need = pq_qlinear_dyn_workspace_bytes(M, N, K);
    if (!workspace || workspace_bytes < need) return fail(PQ_ERR_WORKSPACE, "pq_qlinear_dyn: workspace %zu < %zu bytes", workspace ? workspace_bytes : (size_t)0, need);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(PQ_ERR_BAD_ALIGN, "pq_qlinear_dyn: workspace must be 256-byte aligned");
    uint8_t* base = static_cast<uint8_t*>(workspace);
    int8_t* xq = reinterpret_cast<int8_t*>(base);
    float* xs = reinterpret_cast<float*>(base + align256((size_t)M * (size_t)K));
    uint8_t* slabs = base + align256((size_t)M * (size_t)K) + align256((size_t)M * sizeof(float));
    const size_t slab_bytes = pq_qlinear_workspace_bytes(M, N, K);
    int32_t st = pq_quant_rowwise(x, dtype, M, K, ld_x, xq, K, xs, stream);
    if (st != PQ_OK) return st;
    return pq_qlinear_s8(xq, K, xs, w, ldw, w_scale, bias, y, ldy, dtype, M, N, K, slab_bytes ? slabs : nullptr, slab_bytes, stream);
}

int32_t pq_selftest_fast_quotient(const uint32_t* x_bits, const uint32_t* s_bits, int64_t n, unsigned long long* mismatches, void* stream) {
    if (!x_bits || !s_bits || !mismatches || n < 0) return fail(PQ_ERR_BAD_ARG, "pq_selftest_fast_quotient: bad arguments");
    pq::launch_fast_quotient_check(x_bits, s_bits, n, mismatches, static_cast<hipStream_t>(stream));
    return check_launch("pq_selftest_fast_quotient");
}

int32_t pq_selftest_half_encode(int32_t dtype, unsigned long long* counts, void* stream) {
    if (!counts || (dtype != PQ_BF16 && dtype != PQ_FP16)) return fail(PQ_ERR_BAD_ARG, "pq_selftest_half_encode: dtype must be bf16 or fp16, counts non-null");
    pq::launch_half_encode_check(dtype, counts, static_cast<hipStream_t>(stream));
    return check_launch("pq_selftest_half_encode");
}

int32_t pq_selftest_silu_short(int32_t dtype, unsigned long long* counts, void* stream) {
    if (!counts || (dtype != PQ_BF16 && dtype != PQ_FP16)) return fail(PQ_ERR_BAD_ARG, "pq_selftest_silu_short: dtype must be bf16 or fp16, counts non-null");
    pq::launch_silu_short_check(dtype, counts, static_cast<hipStream_t>(stream));
    return check_launch("pq_selftest_silu_short");
}

#ifdef PQ_ABLATION_BUILD
void pq_dev_set_stamp_buffer(unsigned long long* p) { pq::set_stamp_buffer(p); }
#endif

const char* pq_gemm_variant_name(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
    CallScope scope_;
    // alignment of the pointers is unknown here: assume 16-byte aligned bases
    switch (pick_variant(reinterpret_cast<const int8_t*>(16), lda, reinterpret_cast<const int8_t*>(16), ldb, M, N, K)) {
        case V_SP256_16: {
            int64_t lead = 0;
            const int axis = forced_variant() == V_AUTO ? tail_split_plan(M, N, &lead) : 0;
            return axis == 1 ? "sp256_16x16x64 + sp128 tail (N)" : axis == 2 ? "sp256_16x16x64 + sp128 tail (M)" : "sp256_16x16x64";
        }
        case V_SP128_16: return "sp128x256_16x16x64";
        case V_SP128X128: return "sp128x128_16x16x64";
        case V_RING128: return "ring128_16x16x64";
        case V_SKINNY: return "skinny_16x16x64";
        case V_RING64X128: return "ring64x128_16x16x64";
        case V_RING64X64: return "ring64x64_16x16x64";
        case V_RING128X160: return "ring128x160_16x16x64";
        default: return "generic64";
    }
}

}  // extern "C"
