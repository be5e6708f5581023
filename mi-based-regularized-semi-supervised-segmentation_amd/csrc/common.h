// Shared helpers for the gfx950 kernels behind include/miseg_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>

// ---- the fp16 build -------------------------------------------------------------------------------------------------------
// bn / conv / heads / heads_var / mi_global are compiled TWICE: as written (16-bit storage type = bf16) and with
// -DMISEG_F16_BUILD, where the same sources get IEEE half as their 16-bit type: `bf16` below becomes __half, the conversion helpers
// and the MFMA builtins switch to their f16 forms, the namespace becomes miseg_f16 and every C entry point is renamed f16_miseg_*
// (f16_rename.h, generated from the header by the Makefile).  The primary entry points forward dt == MISEG_F16 calls to those
// (MISEG_F16_DISPATCH_ON).  The data movement (LDS tiles, ds_read_b64_tr_b16, packed 16-bit stores) is type-agnostic.
#ifdef MISEG_F16_BUILD
#include <hip/hip_fp16.h>
#include "f16_rename.h"
#endif
#include "../../include/miseg_hip.h"
#include "tape.h"
#ifndef MISEG_F16_BUILD
#include "f16_protos.h"
#define MISEG_F16_DISPATCH_ON(var, fn, ...)          \
    do {                                            \
        if (var == MISEG_F16) return f16_##fn(__VA_ARGS__); \
    } while (0)
#else
#define MISEG_F16_DISPATCH_ON(var, fn, ...) do { } while (0)
#endif

namespace miseg_core {
// ---- error plumbing (thread-local message, C return codes) -------------------------------
char* last_error_buf();
int fail(int code, const char* fmt, ...);
}  // namespace miseg_core

#ifdef MISEG_F16_BUILD
#define miseg miseg_f16
#define __bf16 _Float16
#define __float2bfloat16 __float2half
#define __bfloat162float __half2float
#endif

namespace miseg {
using miseg_core::fail;
using miseg_core::last_error_buf;

#define MISEG_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return ::miseg::fail(MISEG_E_INVALID, __VA_ARGS__); \
    } while (0)

#define MISEG_LAUNCH_CHECK(what)                                                                   \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return ::miseg::fail(MISEG_E_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- element types ------------------------------------------------------------------------
#ifdef MISEG_F16_BUILD
typedef __half bf16;            // the 16-bit storage / operand type of this build (the name stays: the sources are shared)
#else
typedef __hip_bfloat16 bf16;
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return __float2bfloat16(v); }

#ifdef MISEG_F16_BUILD
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __half2float(__ushort_as_half(b)); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) { return __half_as_ushort(__float2half(v)); }
// the bf16 MFMA builtins of the shared sources -> their f16 forms (same shapes, same rates; operands re-typed bit for bit)
typedef _Float16 miseg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 miseg_h4 __attribute__((ext_vector_type(4)));
template <typename A, typename B> __device__ __forceinline__ f32x4 f16_mfma_16x16x32(A a, B b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(miseg_h8, a), __builtin_bit_cast(miseg_h8, b), c, 0, 0, 0);
}
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <typename A, typename B> __device__ __forceinline__ f32x16_t f16_mfma_32x32x16(A a, B b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(miseg_h8, a), __builtin_bit_cast(miseg_h8, b), c, 0, 0, 0);
}
template <typename A, typename B> __device__ __forceinline__ f32x4 f16_mfma_16x16x16(A a, B b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(miseg_h4, a), __builtin_bit_cast(miseg_h4, b), c, 0, 0, 0);
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) ::miseg_f16::f16_mfma_16x16x32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) ::miseg_f16::f16_mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, x, y, z) ::miseg_f16::f16_mfma_16x16x16(a, b, c)
#else
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) {
    bf16 h = __float2bfloat16(v);
    return *reinterpret_cast<unsigned short*>(&h);
}
#endif

// ---- 16-byte vectors of the NHWC activation tensors: V elements of T <-> V floats
template <typename T> struct VT;
template <> struct VT<float> {
    static constexpr int V = 4;
    typedef float4 Raw;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
    static __device__ __forceinline__ Raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct VT<bf16> {
    static constexpr int V = 8;
    typedef uint4 Raw;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
        const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_bits_to_f32((unsigned short)(u[i] & 0xffffu)); f[2 * i + 1] = bf16_bits_to_f32((unsigned short)(u[i] >> 16)); }
    }
    static __device__ __forceinline__ Raw pack(const float* f) {
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = (unsigned)f32_to_bf16_bits(f[2 * i]) | ((unsigned)f32_to_bf16_bits(f[2 * i + 1]) << 16);
        return make_uint4(u[0], u[1], u[2], u[3]);
    }
};


// y = T(relu(x)) is > 0 exactly when x exceeds the largest fp32 that T rounds to zero: the ReLU mask of the BatchNorm backward is
// recomputed from the raw convolution output (bn.hip, conv.hip) and must agree with the activation as it was STORED.
template <typename T> __device__ __forceinline__ float relu_keep_threshold();
template <> __device__ __forceinline__ float relu_keep_threshold<float>() { return 0.f; }
#ifdef MISEG_F16_BUILD
template <> __device__ __forceinline__ float relu_keep_threshold<bf16>() { return 0x1p-25f; }    // half: ties-to-even at half the smallest subnormal
#else
template <> __device__ __forceinline__ float relu_keep_threshold<bf16>() { return 0x1p-134f; }
#endif

// BatchNorm backward folded into the convolutions that consume it (conv.hip; coefficients written by bn_bwd_finalize_kernel):
//   graw = m * A * gy + P + Q * (raw - mean),   m = [T(relu(raw * scale + shift)) > 0]
// with A = gamma * invstd, P = -A * mean(dz), Q = -A * invstd * mean(dz * xhat) (training; P = Q = 0 in eval mode): the tensor
// bn_bwd_apply_kernel would have written, formed in the loader of the data- / weight-gradient kernel instead.
constexpr int kBwdCoefRows = 6;      // coef[6][C]: scale | shift | mean | A | P | Q
struct BnLoad {
    const void* gy;                  // gradient of the layer's activation (NHWC T), same shape as the raw tensor; null: plain source
    const float* coef;
};
// ... and the statistics pass of the PRODUCER of the tensor whose gradient a data-gradient launch writes, taken in its epilogue:
// parts[block][2][C] = per-block sum(dz), sum(dz * xhat) with dz = [y > 0] * (the gradient as stored), for bn_bwd_finalize_kernel.
struct BnRed {
    const void* raw;                 // the producer's raw convolution output, shape of this launch's output; null: off
    const float* saved;              // its saved[4][C]: mean | invstd | scale | shift
    float* parts;
};
template <typename T>
__device__ __forceinline__ typename VT<T>::Raw bn_graw_vec(const typename VT<T>::Raw& rraw, const typename VT<T>::Raw& rgy, const float* sc,
                                                           const float* sh, const float* mu, const float* ca, const float* cp, const float* cq) {
    constexpr int V = VT<T>::V;
    float fr[V], fg[V];
    VT<T>::unpack(rraw, fr);
    VT<T>::unpack(rgy, fg);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float yv = fr[i] * sc[i] + sh[i];
        const float lin = cp[i] + cq[i] * (fr[i] - mu[i]);
        fg[i] = yv > relu_keep_threshold<T>() ? ca[i] * fg[i] + lin : lin;
    }
    return VT<T>::pack(fg);
}

// ---- wave64 / block reductions --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over a block of up to 1024 threads; result valid in every thread.  `red` = >=17 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < nw; ++i) t += red[i];  // fixed order: deterministic
        red[16] = t;
    }
    __syncthreads();
    return red[16];
}
// ---- "last block finishes" (one launch instead of a reduce kernel + a finalize kernel) -------------------------------------------
// Every block of a grid writes one row of a [nparts][ncols] fp32 matrix of partial sums, then calls last_block_arrives(): exactly
// one block -- the last to arrive -- gets true, after which every row is visible to it (agent-scope release by the writers, acquire
// by the reader: the per-XCD L2s are written back / invalidated by the fences).  `counter` must be 0 at launch; the caller's last
// block sets it back to 0 (last_block_done), so one counter serves launch after launch on a stream.
__device__ __forceinline__ bool last_block_arrives(unsigned int* counter, unsigned int total_blocks) {
    __shared__ int s_last;
    __syncthreads();                 // the block's partial-sum stores happen-before thread 0's release
    if (threadIdx.x == 0) {
        // ONE release per block, and a release only: __threadfence() in every thread is a release + ACQUIRE per wave, and the acquire
        // half (buffer_inv sc1) drops the XCD's L2 lines under the blocks that are still streaming -- measured +20 % on the whole step
#ifndef MISEG_PARTS_WRITE_THROUGH
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total_blocks - 1u;
    }
    __syncthreads();
    if (!s_last) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the other blocks' rows (their release -> our ticket -> this acquire)
    return true;
}
// A partial-sum store of a launch that ends in last_block_arrives().  With MISEG_PARTS_WRITE_THROUGH the rows go through the L2 as
// agent-scope relaxed atomic stores (sc1: written through) and the ticket needs no release fence (no buffer_wbl2 per block).
__device__ __forceinline__ void store_part(float* p, float v) {
#ifdef MISEG_PARTS_WRITE_THROUGH
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void last_block_done(unsigned int* counter) {
    if (threadIdx.x == 0) *counter = 0u;
}
// Column sums of parts[nparts][ncols] (ncols % 4 == 0, ncols <= 1024) by ONE 256-thread block, into lds[1024 .. 1024 + ncols); lds
// holds >= 1024 + ncols floats.  16-byte loads, 256 / (ncols / 4) row groups in flight, combined in a fixed order: deterministic.
__device__ __forceinline__ float* block_column_sums(const float* __restrict__ parts, int nparts, int ncols, float* lds) {
    const int tid = threadIdx.x, nc4 = ncols >> 2;
    float4* l4 = reinterpret_cast<float4*>(lds);
    float* out = lds + 1024;
    for (int c0 = 0; c0 < nc4; c0 += 256) {                    // one pass unless ncols > 1024
        const int w = min(nc4 - c0, 256), G = 256 / w, grp = tid / w, c4 = c0 + tid - grp * w;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp < G)
            for (int q = grp; q < nparts; q += G) {
                const float4 v = *reinterpret_cast<const float4*>(parts + (size_t)q * ncols + 4 * c4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        __syncthreads();
        l4[tid] = acc;
        __syncthreads();
        if (tid < w) {
            float4 s = l4[tid];
            for (int g2 = 1; g2 < G; ++g2) { const float4 v = l4[g2 * w + tid]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            out[4 * (c0 + tid) + 0] = s.x; out[4 * (c0 + tid) + 1] = s.y; out[4 * (c0 + tid) + 2] = s.z; out[4 * (c0 + tid) + 3] = s.w;
        }
    }
    __syncthreads();
    return out;
}

// BatchNorm statistics -> coefficients by the last block of the producing convolution (conv.hip) -- the body of bn_finalize_kernel.
struct BnFinish {
    unsigned int* counter;           // null: the caller launches bn_finalize_kernel itself
    const float* gamma; const float* beta; float* rmean; float* rvar; long long* nbt; float* saved;
    float count, eps, momentum;
    int accumulate_out;              // POOL epilogues of the conv kernels: add to the output tensor instead of storing (a gradient join)
    void* out1; int split;           // full-resolution epilogues: output channels [split, Cout) go to out1 ([.., Cout - split]), [0, split) to
                                     // out ([.., split]) -- the data gradient of a concat convolution, one launch for both sources
    unsigned long long* acc;         // non-null: the statistics leave as fixed-point atomic adds into acc[2 Cout + 1] (bn_acc_add) instead of a row per block
};

// ---- BatchNorm batch statistics without a finalize launch.  Every block of the producing convolution adds its per-channel sum and sum of
// squares to ONE accumulator as 64-bit fixed point (2^-20 units): integer addition is associative, so the totals do not depend on the
// order the blocks arrive in (a float atomic would) -- the step stays bit-reproducible, and the totals are exacter than a float tree
// (each block rounds once, to 1e-6 absolute).  The consumer (bn_relu_fwd with ACC) turns the two integers of a channel into mean /
// variance in its prologue; the accumulator is zero at the start of an iteration (the step block's upload, StepIO.acc64).
// Layout: acc[0 .. 2C) the sums, acc[2C] a count of contributions that did not fit.  Range: |block sum| < 2^30 (1.07e9: the squares of
// 6 000 activations of magnitude 400), at most 2^11 blocks -> |total| < 2^61, no wrap; a larger or non-finite block sum is counted in
// acc[2C] instead and the consumer then reports NaN statistics for the layer, as the float path would for a non-finite sum.
constexpr float kBnAccScale = 1048576.f;
constexpr int kBnAccMaxBlocks = 2048;
__device__ __forceinline__ void bn_acc_add(unsigned long long* acc, unsigned long long* misfit, float v) {
    if (fabsf(v) < 1.0737e9f) __hip_atomic_fetch_add(acc, (unsigned long long)__float2ll_rn(v * kBnAccScale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(misfit, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double bn_acc_value(unsigned long long raw, unsigned long long misfit) {
    return misfit ? __builtin_nan("") : (double)(long long)raw * (1.0 / (double)kBnAccScale);
}
// The same for the BACKWARD's sums (sum dz, sum dz * xhat): gradients span many decades and, in the fp16 mode, carry the loss scale, so
// one scale does not do.  Two tiers: 2^-40 units for block sums below 2^10 (gradients as they come), 2^-12 units for block sums below
// 2^38 (loss-scaled ones); a block adds to every tier its sum fits and counts a misfit in the others; the consumer takes the finest
// tier without a misfit.  Layout: acc[0 .. 2C) fine sums, acc[2C .. 4C) coarse sums, acc[4C], acc[4C + 1] the two misfit counts.
constexpr float kBnAccFine = 1099511627776.f, kBnAccCoarse = 4096.f;
__device__ __forceinline__ void bn_acc2_add(unsigned long long* acc, int idx, int C, float v) {
    if (fabsf(v) < 1024.f) __hip_atomic_fetch_add(acc + idx, (unsigned long long)__float2ll_rn(v * kBnAccFine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(acc + 4 * C, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fabsf(v) < 2.7487e11f) __hip_atomic_fetch_add(acc + 2 * C + idx, (unsigned long long)__float2ll_rn(v * kBnAccCoarse), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(acc + 4 * C + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float bn_acc2_value(const unsigned long long* acc, int idx, int C) {
    if (acc[4 * C] == 0) return (float)((double)(long long)acc[idx] * (1.0 / (double)kBnAccFine));
    if (acc[4 * C + 1] == 0) return (float)((double)(long long)acc[2 * C + idx] * (1.0 / (double)kBnAccCoarse));
    return __builtin_nanf("");
}
__device__ __forceinline__ void bn_finish_block(const float* __restrict__ parts, int nparts, int C, const BnFinish& f, float* lds) {
    const float* sums = block_column_sums(parts, nparts, 2 * C, lds);      // [sum | sum of squares] per channel
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mean = sums[c] / f.count;
        const float var = fmaxf(sums[C + c] / f.count - mean * mean, 0.f);  // biased batch variance
        const float invstd = rsqrtf(var + f.eps), sc = f.gamma[c] * invstd;
        f.saved[c] = mean; f.saved[C + c] = invstd; f.saved[2 * C + c] = sc; f.saved[3 * C + c] = f.beta[c] - mean * sc;
        if (f.rmean) {
            const float unb = f.count > 1.f ? var * f.count / (f.count - 1.f) : var;
            f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mean;
            f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * unb;
            if (c == 0 && f.nbt) f.nbt[0] += 1;
        }
    }
}

__device__ __forceinline__ float block_min(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_min(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
        for (int i = 1; i < nw; ++i) t = fminf(t, red[i]);
        red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// Deterministic sum over `nparts` partial vectors: out[e] = sum_q parts[q*stride + map(e)].
// 64 outputs x 4 split-groups per 256-thread block; each group sums q = g, g+4, ... serially, the 4 group
// sums are added in fixed order -> bitwise reproducible, 4x the memory-level parallelism of one thread per output.
// reduce_partials_block: outputs per block depend on the number of parts (host side: reduce_grid)
constexpr int kWideReduceParts = 64, kFlatReduceParts = 16;
static inline unsigned reduce_grid(int64_t total, int64_t nparts) {
    if (nparts <= kFlatReduceParts) return (unsigned)((total + 255) / 256);
    return (unsigned)((total + (nparts >= kWideReduceParts ? 15 : 63)) / (nparts >= kWideReduceParts ? 16 : 64));
}

// `map(e)`: offset of reduction element e inside one part; `omap(e)`: where its sum goes in `out` (negative: nowhere).  Keep
// consecutive e contiguous in the PARTS (that is where the bytes are); let the output index take the permutation.
// `out`: anything indexable that yields a float lvalue -- a float*, or Split2Out below (one reduced vector, two destinations).
struct Split2Out {          // elements [0, n0) -> d0, the rest -> d1 (stacked weight | bias gradients into their own buffers)
    float* d0;
    int n0;
    float* d1;
    __device__ __forceinline__ float& operator[](int i) const { return i < n0 ? d0[i] : d1[i - n0]; }
};
template <typename OutT, typename MapFn, typename OutFn>
__device__ __forceinline__ void reduce_partials_block(const float* __restrict__ parts, int nparts, size_t stride, int total,
                                                      OutT out, MapFn map, OutFn omap) {
    __shared__ float rp_sm[16][64];
    if (nparts <= kFlatReduceParts) {
        // few parts, many outputs: one thread per output, the parts summed in order from registers (no LDS, no barrier)
        const int e = blockIdx.x * 256 + threadIdx.x;
        if (e < total && omap(e) >= 0) {
            const float* p = parts + map(e);
            float v[kFlatReduceParts];
#pragma unroll
            for (int q = 0; q < kFlatReduceParts; ++q) v[q] = q < nparts ? p[(size_t)q * stride] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < kFlatReduceParts; ++q) s += v[q];
            const int oi = omap(e);
            if (oi >= 0) out[oi] = s;
        }
        return;
    }
    if (nparts >= kWideReduceParts) {
        // many parts, (usually) few outputs: 16 outputs x 16 part-groups per block, 4 independent chains per thread,
        // then a fixed-order tree over the groups -- same result for any grid, ~16x the loads in flight of the narrow form
        const int o = threadIdx.x & 15, grp = threadIdx.x >> 4, e = blockIdx.x * 16 + o;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (e < total && omap(e) >= 0) {      // elements without a destination (tile padding) are not read at all
            const float* p = parts + map(e);
            int q = grp;
            for (; q + 48 < nparts; q += 64) {
                s0 += p[(size_t)q * stride];
                s1 += p[(size_t)(q + 16) * stride];
                s2 += p[(size_t)(q + 32) * stride];
                s3 += p[(size_t)(q + 48) * stride];
            }
            for (; q < nparts; q += 16) s0 += p[(size_t)q * stride];
        }
        rp_sm[grp][o] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (grp == 0 && e < total) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += rp_sm[g][o];
            const int oi = omap(e);
            if (oi >= 0) out[oi] = s;
        }
        return;
    }
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s = 0.f;
    if (e < total && omap(e) >= 0) {
        const size_t off = map(e);
        for (int q = grp; q < nparts; q += 4) s += parts[(size_t)q * stride + off];
    }
    rp_sm[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && e < total) {
        const int oi = omap(e);
        if (oi >= 0) out[oi] = ((rp_sm[0][threadIdx.x] + rp_sm[1][threadIdx.x]) + rp_sm[2][threadIdx.x]) + rp_sm[3][threadIdx.x];
    }
}

template <typename OutT, typename MapFn>
__device__ __forceinline__ void reduce_partials_block(const float* __restrict__ parts, int nparts, size_t stride, int total,
                                                      OutT out, MapFn map) {
    reduce_partials_block(parts, nparts, stride, total, out, map, [](int e) { return e; });
}

// Flip-aware source coordinate: bit0 = flip H, bit1 = flip W.
__device__ __forceinline__ int flip_h(int h, int H, int f) { return (f & 1) ? (H - 1 - h) : h; }
__device__ __forceinline__ int flip_w(int w, int W, int f) { return (f & 2) ? (W - 1 - w) : w; }

}  // namespace miseg
