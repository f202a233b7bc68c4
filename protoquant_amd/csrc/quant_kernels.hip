// quant_kernels.hip — K1 per-token row quant, K2 per-channel column quant, dequant (gfx950).
// All three are HBM-bound byte movers: 16-byte coalesced loads, one read of x for K1 (the row lives
// in registers between the amax reduction and the encode), wavefront shuffles + one LDS hop for the
// reductions.  Arithmetic follows QSPEC v1 exactly (true fp32 division, RNE, no contraction).
#include "pq_common.h"

namespace pq {

template <int DT, int N> struct Unpack;
template <> struct Unpack<PQ_BF16, 8> {
    __device__ static __forceinline__ void run(const v4u& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
            f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xFFFF0000u);
        }
    }
};
template <> struct Unpack<PQ_FP16, 8> {
    __device__ static __forceinline__ void run(const v4u& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = Elem<PQ_FP16>::to_f32((uint16_t)(v[i] & 0xFFFFu));
            f[2 * i + 1] = Elem<PQ_FP16>::to_f32((uint16_t)(v[i] >> 16));
        }
    }
};
template <> struct Unpack<PQ_F32, 4> {
    __device__ static __forceinline__ void run(const v4u& v, float (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u = v[i];   // copy first: bit_cast of a vector-element lvalue reads element 0
            f[i] = __builtin_bit_cast(float, u);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// K1 vector path.  TPR threads own one row; thread t holds 16-byte vectors t, t+TPR, ... (VPT of
// them) so every wave-instruction reads 1 KiB contiguous.  Algorithmic traffic: read once, write
// 1 B/elem + 4 B/row.
template <int DT, int VPT, int TPR>
__global__ __launch_bounds__(256) void quant_rowwise_vec(const uint8_t* __restrict__ x, int64_t rows, int nvec,
                                                         int64_t ldx_bytes, int8_t* __restrict__ q, int64_t ldq,
                                                         float* __restrict__ scale) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    constexpr int RPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
    const bool active = row < rows;
    const uint8_t* xr = x + (active ? row : 0) * ldx_bytes;

    v4u v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = i * TPR + t;
        v[i] = (active && idx < nvec) ? *reinterpret_cast<const v4u*>(xr + (int64_t)idx * 16) : v4u{0, 0, 0, 0};
    }
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        float f[EPV];
        Unpack<DT, EPV>::run(v[i], f);
#pragma unroll
        for (int j = 0; j < EPV; ++j) amax = amax_step(amax, f[j]);
    }
    amax = wave_max(amax);
    if constexpr (TPR > kWave) {
        __shared__ float part[256 / kWave];
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = amax;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 256 / kWave; ++w) amax = part[w] > amax ? part[w] : amax;
    }
    const float s = scale_of(amax);
    if (active && t == 0) scale[row] = s;
    if (!active) return;
    int8_t* qr = q + row * ldq;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = i * TPR + t;
        if (idx < nvec) {
            float f[EPV];
            Unpack<DT, EPV>::run(v[i], f);
            int c[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) c[j] = code_of(f[j], s);
            if constexpr (EPV == 8) {
                v2u o = {pack4(c[0], c[1], c[2], c[3]), pack4(c[4], c[5], c[6], c[7])};
                *reinterpret_cast<v2u*>(qr + (int64_t)idx * 8) = o;
            } else {
                *reinterpret_cast<uint32_t*>(qr + (int64_t)idx * 4) = pack4(c[0], c[1], c[2], c[3]);
            }
        }
    }
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
    for (int w = 0; w < 4; ++w) amax = part[w] > amax ? part[w] : amax;
    const float s = scale_of(amax);
    if (threadIdx.x == 0) scale[row] = s;
    int8_t* qr = q + row * ldq;
    for (int64_t c = threadIdx.x; c < cols; c += 256) qr[c] = (int8_t)code_of(Elem<DT>::to_f32(xr[c]), s);
}

// ------------------------------------------------------------------------------------------------
// K2: reduction along the strided axis.  Launch 1: column amax (bit patterns of non-negative floats
// order like unsigned ints -> atomicMax on uint32) into `scale`; launch 2: encode with scale_of()
// computed per thread once for its columns; launch 3: scale[c] = scale_of(amax[c]) in place.
template <int DT, bool VEC>
__global__ __launch_bounds__(256) void col_amax(const uint8_t* __restrict__ x, int64_t rows, int64_t ncolv,
                                                int64_t ldx_bytes, uint32_t* __restrict__ amax_bits, int rows_per_block) {
    constexpr int EPV = VEC ? 16 / Elem<DT>::kBytes : 1;
    using S = typename Elem<DT>::store_t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t cv = (int64_t)blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float m[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) m[j] = 0.0f;
    if (cv < ncolv) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            const uint8_t* p = x + r * ldx_bytes + cv * (VEC ? 16 : Elem<DT>::kBytes);
            if constexpr (VEC) {
                float f[EPV];
                Unpack<DT, EPV>::run(*reinterpret_cast<const v4u*>(p), f);
#pragma unroll
                for (int j = 0; j < EPV; ++j) m[j] = amax_step(m[j], f[j]);
            } else {
                m[0] = amax_step(m[0], Elem<DT>::to_f32(*reinterpret_cast<const S*>(p)));
            }
        }
    }
    __shared__ float part[4][64][EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) part[w][lane][j] = m[j];
    __syncthreads();
    if (w == 0 && cv < ncolv) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            float a = m[j];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) a = part[ww][lane][j] > a ? part[ww][lane][j] : a;
            atomicMax(&amax_bits[cv * EPV + j], __builtin_bit_cast(uint32_t, a));
        }
    }
}

template <int DT, bool VEC>
__global__ __launch_bounds__(256) void col_encode(const uint8_t* __restrict__ x, int64_t rows, int64_t ncolv,
                                                  int64_t ldx_bytes, const float* __restrict__ amax,
                                                  int8_t* __restrict__ q, int64_t ldq, int rows_per_block) {
    constexpr int EPV = VEC ? 16 / Elem<DT>::kBytes : 1;
    using S = typename Elem<DT>::store_t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t cv = (int64_t)blockIdx.x * 64 + lane;
    if (cv >= ncolv) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s[j] = scale_of(amax[cv * EPV + j]);
    for (int64_t r = r0 + w; r < r1; r += 4) {
        const uint8_t* p = x + r * ldx_bytes + cv * (VEC ? 16 : Elem<DT>::kBytes);
        int8_t* o = q + r * ldq + cv * EPV;
        if constexpr (VEC) {
            float f[EPV];
            Unpack<DT, EPV>::run(*reinterpret_cast<const v4u*>(p), f);
            int c[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) c[j] = code_of(f[j], s[j]);
            if constexpr (EPV == 8) {
                v2u ov = {pack4(c[0], c[1], c[2], c[3]), pack4(c[4], c[5], c[6], c[7])};
                *reinterpret_cast<v2u*>(o) = ov;
            } else {
                *reinterpret_cast<uint32_t*>(o) = pack4(c[0], c[1], c[2], c[3]);
            }
        } else {
            *o = (int8_t)code_of(Elem<DT>::to_f32(*reinterpret_cast<const S*>(p)), s[0]);
        }
    }
}

__global__ void col_finalize(float* __restrict__ scale, int64_t cols) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) scale[c] = scale_of(scale[c]);
}

// ------------------------------------------------------------------------------------------------
// dequant: out = cast_rne(f32(q) * scale[kept axis]).  Vector path: 16 codes per thread.
template <int ODT, bool VEC>
__global__ __launch_bounds__(256) void dequant_kernel(const int8_t* __restrict__ q, int64_t ldq,
                                                      const float* __restrict__ scale, int axis, int64_t rows,
                                                      int64_t ncolv, void* __restrict__ out, int64_t ldo) {
    using S = typename Elem<ODT>::store_t;
    constexpr int EPV = VEC ? 16 : 1;
    const int64_t cv = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t r = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cv >= ncolv || r >= rows) return;
    const int8_t* p = q + r * ldq + cv * EPV;
    S* o = reinterpret_cast<S*>(out) + r * ldo + cv * EPV;
    if constexpr (VEC) {
        const v4u v = *reinterpret_cast<const v4u*>(p);
        const float sr = axis == 0 ? 0.0f : scale[r];
        S res[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int code = (int)(int8_t)((v[j >> 2] >> (8 * (j & 3))) & 0xFF);
            const float s = axis == 0 ? scale[cv * 16 + j] : sr;
            res[j] = Elem<ODT>::from_f32((float)code * s);
        }
        constexpr int NV = 16 * (int)sizeof(S) / 16;
        const v4u* rv = reinterpret_cast<const v4u*>(res);
#pragma unroll
        for (int k = 0; k < NV; ++k) reinterpret_cast<v4u*>(o)[k] = rv[k];
    } else {
        const float s = axis == 0 ? scale[cv] : scale[r];
        *o = Elem<ODT>::from_f32((float)(*p) * s);
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from pq_api.hip)
static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

template <int DT, int TPR>
static void launch_rowwise_vec(int vpt, const void* x, int64_t rows, int nvec, int64_t ldx_bytes, int8_t* q,
                               int64_t ldq, float* scale, hipStream_t st) {
    constexpr int RPB = 256 / TPR;
    const dim3 grid((unsigned)((rows + RPB - 1) / RPB)), block(256);
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
    switch (vpt) {
        case 1: quant_rowwise_vec<DT, 1, TPR><<<grid, block, 0, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 2: quant_rowwise_vec<DT, 2, TPR><<<grid, block, 0, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 4: quant_rowwise_vec<DT, 4, TPR><<<grid, block, 0, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        case 8: quant_rowwise_vec<DT, 8, TPR><<<grid, block, 0, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
        default: quant_rowwise_vec<DT, 16, TPR><<<grid, block, 0, st>>>(xb, rows, nvec, ldx_bytes, q, ldq, scale); break;
    }
}

template <int DT>
void quant_rowwise_dispatch(const void* x, int64_t rows, int64_t cols, int64_t ldx, int8_t* q, int64_t ldq,
                            float* scale, hipStream_t st) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const bool vec_ok = (cols % EPV == 0) && (ldx % EPV == 0) && aligned(x, 16) && (ldq % EPV == 0) &&
                        aligned(q, EPV) && cols / EPV <= 256 * 16;
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

template <int DT>
void quant_colwise_dispatch(const void* x, int64_t rows, int64_t cols, int64_t ldx, int8_t* q, int64_t ldq,
                            float* scale, hipStream_t st) {
    constexpr int EPV = 16 / Elem<DT>::kBytes;
    const bool vec_ok = (cols % EPV == 0) && (ldx % EPV == 0) && aligned(x, 16) && (ldq % EPV == 0) && aligned(q, EPV);
    const int64_t ncolv = vec_ok ? cols / EPV : cols;
    const int rpb = 64;
    const dim3 grid((unsigned)((ncolv + 63) / 64), (unsigned)((rows + rpb - 1) / rpb)), block(256);
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
    const int64_t ldb = ldx * Elem<DT>::kBytes;
    (void)hipMemsetAsync(scale, 0, (size_t)cols * sizeof(float), st);
    if (rows > 0) {
        if (vec_ok) {
            col_amax<DT, true><<<grid, block, 0, st>>>(xb, rows, ncolv, ldb, reinterpret_cast<uint32_t*>(scale), rpb);
            col_encode<DT, true><<<grid, block, 0, st>>>(xb, rows, ncolv, ldb, scale, q, ldq, rpb);
        } else {
            col_amax<DT, false><<<grid, block, 0, st>>>(xb, rows, ncolv, ldb, reinterpret_cast<uint32_t*>(scale), rpb);
            col_encode<DT, false><<<grid, block, 0, st>>>(xb, rows, ncolv, ldb, scale, q, ldq, rpb);
        }
    }
    col_finalize<<<dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st>>>(scale, cols);
}

template <int ODT>
void dequant_dispatch(const int8_t* q, int64_t ldq, const float* scale, int axis, int64_t rows, int64_t cols,
                      void* out, int64_t ldo, hipStream_t st) {
    const bool vec_ok = (cols % 16 == 0) && (ldq % 16 == 0) && aligned(q, 16) && aligned(out, 16) &&
                        ((ldo * Elem<ODT>::kBytes) % 16 == 0);
    const int64_t ncolv = vec_ok ? cols / 16 : cols;
    const dim3 grid((unsigned)((ncolv + 63) / 64), (unsigned)((rows + 3) / 4)), block(256);
    if (vec_ok) dequant_kernel<ODT, true><<<grid, block, 0, st>>>(q, ldq, scale, axis, rows, ncolv, out, ldo);
    else dequant_kernel<ODT, false><<<grid, block, 0, st>>>(q, ldq, scale, axis, rows, ncolv, out, ldo);
}

// explicit instantiations used by pq_api.hip
template void quant_rowwise_dispatch<PQ_BF16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_rowwise_dispatch<PQ_FP16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_rowwise_dispatch<PQ_F32>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_colwise_dispatch<PQ_BF16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_colwise_dispatch<PQ_FP16>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void quant_colwise_dispatch<PQ_F32>(const void*, int64_t, int64_t, int64_t, int8_t*, int64_t, float*, hipStream_t);
template void dequant_dispatch<PQ_BF16>(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);
template void dequant_dispatch<PQ_FP16>(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);
template void dequant_dispatch<PQ_F32>(const int8_t*, int64_t, const float*, int, int64_t, int64_t, void*, int64_t, hipStream_t);

}  // namespace pq
