// LocalClusterHead: S x (1x1 conv C->K + bias -> channel softmax / T), fused with the flip replay and
// the cat([flip(f_u), f_tf]) of the caller (ref: contrastyou/trainer/_utils.py:137-168,
// semi_seg/epocher.py:258-273).  Feature NHWC (dt) in, probabilities fp32 NCHW out (the layout the
// local-MI kernels consume).  One thread = one output pixel; logits staged per-thread in LDS columns
// (conflict-free), weights come through the scalar cache (wave-uniform indices).
#include "common.h"

namespace miseg {

typedef __bf16 bf16x8h_t __attribute__((ext_vector_type(8)));
#define HLDS_S16X4(ptr) ((__attribute__((address_space(3))) s16x4*)(ptr))


constexpr int kHT = 256;

template <typename T> struct Vec4;
template <> struct Vec4<float> { float4 v; __device__ float get(int i) const { return (&v.x)[i]; } };
template <> struct Vec4<bf16> { ushort4 v; __device__ float get(int i) const { return bf16_bits_to_f32((&v.x)[i]); } };

template <typename T>
__global__ __launch_bounds__(kHT) void head_local_fwd_kernel(const T* __restrict__ feat, int H, int W, int C,
                                                             const int32_t* __restrict__ src, const int32_t* __restrict__ flips,
                                                             int M, const float* __restrict__ w, const float* __restrict__ b, int S,
                                                             int K, float invT, float* __restrict__ prob) {
    extern __shared__ float zs[];  // [K][kHT]
    const int tid = threadIdx.x, HW = H * W;
    const int m = blockIdx.y;
    const int pix = blockIdx.x * kHT + tid;
    const bool live = pix < HW;
    const int h = live ? pix / W : 0, wq = live ? pix % W : 0;
    const int f = flips ? flips[m] : 0;
    const T* fp = feat + ((size_t)src[m] * HW + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
    for (int s = 0; s < S; ++s) {
        const float* ws = w + (size_t)s * K * C;
        for (int k = 0; k < K; ++k) zs[k * kHT + tid] = b[s * K + k];
        for (int c0 = 0; c0 < C; c0 += 4) {
            Vec4<T> fv = *reinterpret_cast<const Vec4<T>*>(fp + c0);
            const float f0 = fv.get(0), f1 = fv.get(1), f2 = fv.get(2), f3 = fv.get(3);
            for (int k = 0; k < K; ++k) {
                const float* wr = ws + (size_t)k * C + c0;
                zs[k * kHT + tid] += wr[0] * f0 + wr[1] * f1 + wr[2] * f2 + wr[3] * f3;
            }
        }
        float mx = -3.4e38f;
        for (int k = 0; k < K; ++k) {
            float z = zs[k * kHT + tid] * invT;
            zs[k * kHT + tid] = z;
            mx = fmaxf(mx, z);
        }
        float sum = 0.f;
        for (int k = 0; k < K; ++k) {
            float e = expf(zs[k * kHT + tid] - mx);
            zs[k * kHT + tid] = e;
            sum += e;
        }
        if (live) {
            float* out = prob + (((size_t)s * M + m) * K) * HW + pix;
            for (int k = 0; k < K; ++k) out[(size_t)k * HW] = zs[k * kHT + tid] / sum;
        }
    }
}

// Register-resident variant for K <= 32 (KP = K rounded up to a multiple of 4): the K logits of a pixel live in VGPRs,
// the head weights arrive as scalar operands (wave-uniform addresses -> scalar cache), so a sub-head costs K*C FMAs per
// pixel and no LDS traffic at all (the LDS-column kernel above spends ~15 LDS operations per logit update).
// Logits accumulate through explicit FMAs (channel order) and the softmax multiplies by one reciprocal per pixel: results
// differ from head_local_fwd_kernel by an ulp or two, inside the parity tolerance of tests/test_gpu_mi.py.
template <typename T, int KP, int PPT, bool EXACT>   // EXACT: K == KP, no per-class guards (K = 20 ships)
__global__ __launch_bounds__(kHT) void head_local_fwd_reg_kernel(const T* __restrict__ feat, int H, int W, int C,
                                                                 const int32_t* __restrict__ src, const int32_t* __restrict__ flips,
                                                                 int M, const float* __restrict__ w, const float* __restrict__ b, int S,
                                                                 int K, float invT, float* __restrict__ prob, float tol,
                                                                 int32_t* __restrict__ viol) {
    // PPT consecutive pixels per thread (PPT = 4 needs HW % 4 == 0): every (s,k) plane is then written in 16-byte pieces,
    // 4 KB contiguous per block, instead of 100 interleaved 1 KB streams
    const int tid = threadIdx.x, HW = H * W;
    const int m = blockIdx.y;
    const int pix0 = (blockIdx.x * kHT + tid) * PPT;
    const bool live = pix0 < HW;
    const int f = flips ? flips[m] : 0;
    // all S*K*C weights staged once per block; the inner loop reads them as broadcast ds_read_b128 (every lane the same
    // address), which the compiler keeps a dozen deep in flight -- as scalar loads each batch of weights was a serial
    // scalar-cache round trip in front of its FMAs
    extern __shared__ __attribute__((aligned(16))) float wl[];
    for (int i = tid; i < S * K * C; i += kHT) wl[i] = w[i];
    __syncthreads();
    const T* fp[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int pix = live ? pix0 + j : 0, h = pix / W, wq = pix % W;
        fp[j] = feat + ((size_t)src[m] * HW + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
    }
    int nbad = 0;
    for (int s = 0; s < S; ++s) {
        const float* ws = wl + (size_t)s * K * C;
        float z[PPT][KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float bv = b[s * K + (EXACT ? k : min(k, K - 1))];
#pragma unroll
            for (int j = 0; j < PPT; ++j) z[j][k] = bv;
        }
        for (int c0 = 0; c0 < C; c0 += 4) {
            float fr[PPT][4];
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                Vec4<T> fv = *reinterpret_cast<const Vec4<T>*>(fp[j] + c0);
#pragma unroll
                for (int i = 0; i < 4; ++i) fr[j][i] = fv.get(i);
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float4 wq4 = *reinterpret_cast<const float4*>(ws + (size_t)(EXACT ? k : min(k, K - 1)) * C + c0);
                const float w0 = wq4.x, w1 = wq4.y, w2 = wq4.z, w3 = wq4.w;
#pragma unroll
                for (int j = 0; j < PPT; ++j) z[j][k] = fmaf(w3, fr[j][3], fmaf(w2, fr[j][2], fmaf(w1, fr[j][1], fmaf(w0, fr[j][0], z[j][k]))));
            }
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            // softmax(z / T): exp((z - max) / T) as ONE fma + v_exp_f32 (exp2 of a pre-scaled argument).  The library expf spends
            // ~12 instructions per value on range handling this argument (<= 0, finite) never needs -- it was 40 % of the kernel's
            // instructions -- and v_exp_f32 is accurate to ~1 ulp, inside the 1e-5 parity bound on the probabilities.
            float mx = -3.4e38f;
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (EXACT || k < K) mx = fmaxf(mx, z[j][k]);
            const float sc2 = invT * 1.4426950408889634f, off2 = -mx * sc2;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                z[j][k] = __builtin_amdgcn_exp2f(fmaf(z[j][k], sc2, off2));
                if (EXACT || k < K) sum += z[j][k];
            }
            const float inv = 1.0f / sum;
            float ps = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                z[j][k] *= inv;
                if (EXACT || k < K) ps += z[j][k];
            }
            nbad += !(fabsf(ps - 1.f) <= tol);   // the caller's simplex assertion, evaluated while the values are in registers
        }
        if (live) {
            float* out = prob + (((size_t)s * M + m) * K) * HW + pix0;
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (EXACT || k < K) {
                    if (PPT == 4) *reinterpret_cast<float4*>(out + (size_t)k * HW) = make_float4(z[0][k], z[1][k], z[2][k], z[3][k]);
                    else out[(size_t)k * HW] = z[0][k];
                }
        }
    }
    if (viol && live && nbad) atomicAdd(viol, nbad);
}

// MFMA form of the kernel above for the shipped taps (bf16 features, K = 20, C = 16 or 32).  The register kernel spends
// K*C FMAs per pixel and sub-head on the VALU and is bound by them, not by its 4 bytes-per-probability of output.  Here a wave
// computes logits[class][pixel] = W x feat^T on the matrix pipe: the fp32 weights are split once per block into three bf16
// planes (w = h1 + h2 + h3 to within 2^-25 relative, the features are bf16 already, the products exact and the accumulation
// fp32), so the logits agree with the FMA chain to rounding order.  The D tiles (4 classes x 1 pixel per lane) go through a
// per-wave LDS transpose, 128 pixels at a time, into the softmax layout of the register kernel (4 consecutive pixels x 20
// classes per lane), and the epilogue -- exp2 softmax, simplex check, one 16-byte store per class plane -- is that kernel's.
typedef __bf16 bf16x4h_t __attribute__((ext_vector_type(4)));
template <int C> struct HeadFrag;
template <> struct HeadFrag<32> {
    typedef bf16x8h_t type;
    static __device__ __forceinline__ f32x4 mma(type a, type b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct HeadFrag<16> {
    typedef bf16x4h_t type;
    static __device__ __forceinline__ f32x4 mma(type a, type b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};
constexpr int kHeadZS = 132;   // floats per class row of a wave's transpose buffer (128 pixels + 4: conflict-free D writes)
template <int C> constexpr int head_mfma_cp() { return C + 8; }
template <int C>
__global__ __launch_bounds__(kHT) void head_local_fwd_mfma_kernel(const bf16* __restrict__ feat, int H, int W,
                                                                  const int32_t* __restrict__ src, const int32_t* __restrict__ flips,
                                                                  int M, const float* __restrict__ w, const float* __restrict__ b, int S,
                                                                  float invT, float* __restrict__ prob, float tol,
                                                                  int32_t* __restrict__ viol) {
    constexpr int K = 20, ZS = kHeadZS, CP = head_mfma_cp<C>(), CQ = C / 4;
    typedef typename HeadFrag<C>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char hl[];
    float* ztw = reinterpret_cast<float*>(hl) + (size_t)(threadIdx.x >> 6) * K * ZS;   // this wave's [K][ZS]
    bf16* wp = reinterpret_cast<bf16*>(hl + (size_t)(kHT / 64) * K * ZS * 4);            // [S][3][K][CP]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int HW = H * W, m = blockIdx.y;
    for (int i = tid; i < S * K * C; i += kHT) {
        const float v = w[i];
        const bf16 h1 = __float2bfloat16(v);
        const float r1 = v - __bfloat162float(h1);
        const bf16 h2 = __float2bfloat16(r1);
        const bf16 h3 = __float2bfloat16(r1 - __bfloat162float(h2));
        const int s = i / (K * C), rem = i - s * K * C, k = rem / C, c = rem - k * C;
        bf16* d = wp + ((size_t)(s * 3) * K + k) * CP + c;
        d[0] = h1, d[(size_t)K * CP] = h2, d[(size_t)2 * K * CP] = h3;
    }
    __syncthreads();
    const int f = flips ? flips[m] : 0;
    const int wbase = (blockIdx.x * (kHT / 64) + (tid >> 6)) * 256;   // first pixel of this wave
    const bf16* fs = feat + (size_t)src[m] * HW * C + q * CQ;
    int foff[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int p = min(wbase + t * 16 + l15, HW - 1), h = p / W, wq = p - h * W;
        foff[t] = (flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
    }
    const int pix0 = wbase + lane * 4;
    const bool live = pix0 < HW;
    int nbad = 0;
    for (int s = 0; s < S; ++s) {
        frag_t wa[2][3];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int cls = min(ct * 16 + l15, K - 1);
                frag_t v = *reinterpret_cast<const frag_t*>(wp + ((size_t)(s * 3 + p) * K + cls) * CP + q * CQ);
                if (ct * 16 + l15 >= K)
#pragma unroll
                    for (int e = 0; e < (int)(sizeof(frag_t) / 2); ++e) v[e] = (__bf16)0.f;
                wa[ct][p] = v;
            }
        float z[4][K];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const frag_t fb = *reinterpret_cast<const frag_t*>(fs + foff[half * 8 + t8]);
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 2; p >= 0; --p) {   // smallest plane first
                    a0 = HeadFrag<C>::mma(wa[0][p], fb, a0);
                    a1 = HeadFrag<C>::mma(wa[1][p], fb, a1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) ztw[(4 * q + r) * ZS + t8 * 16 + l15] = a0[r];
                if (q == 0)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ztw[(16 + r) * ZS + t8 * 16 + l15] = a1[r];
            }
            __builtin_amdgcn_wave_barrier();   // a wave's LDS traffic is in order: no workgroup barrier, only no reordering
            if ((lane >> 5) == half) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(ztw + k * ZS + 4 * (lane & 31));
                    z[0][k] = v.x, z[1][k] = v.y, z[2][k] = v.z, z[3][k] = v.w;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float bv = b[s * K + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j][k] += bv;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mx = -3.4e38f;
#pragma unroll
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, z[j][k]);
            const float sc2 = invT * 1.4426950408889634f, off2 = -mx * sc2;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                z[j][k] = __builtin_amdgcn_exp2f(fmaf(z[j][k], sc2, off2));
                sum += z[j][k];
            }
            const float inv = 1.0f / sum;
            float ps = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                z[j][k] *= inv;
                ps += z[j][k];
            }
            nbad += !(fabsf(ps - 1.f) <= tol);
        }
        if (live) {
            float* out = prob + (((size_t)s * M + m) * K) * HW + pix0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // non-temporal: 839 MB that nothing re-reads from a cache
                const f32x4 v4 = {z[0][k], z[1][k], z[2][k], z[3][k]};
                __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(out + (size_t)k * HW));
            }
        }
    }
    if (viol && live && nbad) atomicAdd(viol, nbad);
}

// Fused backward of the local head (one pass over prob / gprob: no dz tensor in memory).  One block = one 64-pixel chunk at a time (persistent, strided):
//   phase 1 (VALU, streaming): dz[(s,k)][px] = p*(g - <g,p>)/T from prob/gprob (coalesced along pixels) -> LDS
//   phase 2 (fp32 MFMA):       gfeat[px][c] = sum_(s,k) dz[(s,k)][px] * W[(s,k)][c]     (M=px, N=c, Kred=S*K)
//   phase 3 (fp32 MFMA):       gw[(s,k)][c] += sum_px dz[(s,k)][px] * f[px][c]          (M=(s,k), N=c, Kred=px)
// gw / gb accumulate in registers across the block's chunks; one deterministic partial per block at the end.
// BF (bf16 features, K20 only): phases 2 and 3 run on v_mfma_f32_16x16x32_bf16 instead of the fp32 16x16x4 form (5x the MACs per
// matrix-pipe cycle).  dz is kept in LDS as two bf16 planes (hi + lo, 2^-16 relative), the weights as two transposed bf16 planes,
// the features are bf16 already (exact): gw = (dz_hi + dz_lo) . f, gfeat = W_hi.dz_hi + W_hi.dz_lo + W_lo.dz_hi (then rounded
// to bf16 anyway).  The fp32 build keeps the exact fp32 MFMA path.
template <typename T, int CTM, int RW, bool K20, bool BF>   // K20: K == 20 and S*K <= 100 -> a row's sub-head is i / 5, a compile-time index
__global__ __launch_bounds__(256, (CTM == 1 ? 3 : 2)) void head_local_bwd_fused_kernel(const T* __restrict__ feat, int H, int W, int C,
                                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ flips,
                                                                   int M, const float* __restrict__ w, int S, int K, float invT,
                                                                   const float* __restrict__ prob, const float* __restrict__ gprob,
                                                                   T* __restrict__ gfeat, float* __restrict__ partials, int nblk, int accumulate) {
    extern __shared__ float sm[];
    const int R = S * K, RT = (R + 15) / 16, RP = RT * 16, CT = (C + 15) / 16, HW = H * W;
    constexpr int DZS = 65;
    constexpr int FPT = CTM * 4;              // feature values per thread: 64 px * (CTM*16) channels / 256 threads
    const int FS = C + 1, WS = C + 1;
    constexpr int RPB = 128, DZB = 72, WTB = 136, CP = CTM * 16;   // BF layout: rows padded to 4 k-steps of 32; strides conflict-free for b128
    float* dzs = sm;                          // [RP][DZS]   p*g, then dz; rows >= R zero
    float* fs = dzs + (size_t)RP * DZS;       // [64][FS]    feature chunk (fp32)
    float* wsm = fs + (size_t)64 * FS;        // [RP][WS]    head weights, rows >= R zero
    float* dots = wsm + (size_t)RP * WS;      // [4 waves][S][64]  partial <g,p> per (wave, sub-head, pixel); generic path: [S][64]
    unsigned short* dzb = reinterpret_cast<unsigned short*>(sm);      // BF: [2][RPB][DZB]  dz hi | lo (bf16 bits), rows >= R zero
    unsigned short* fsT = dzb + 2 * RPB * DZB;                         // BF: [CP][DZB]      features transposed [c][px]
    unsigned short* wT = fsT + CP * DZB;                               // BF: [2][CP][WTB]   W^T hi | lo: [c][r], r >= R zero
    if (BF) dots = reinterpret_cast<float*>(wT + 2 * CP * WTB);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    if (BF) {
        for (int idx = tid; idx < 2 * RPB * DZB; idx += 256) dzb[idx] = 0;
        for (int idx = tid; idx < CP * DZB; idx += 256) fsT[idx] = 0;
        for (int idx = tid; idx < CP * WTB; idx += 256) {
            const int c = idx / WTB, r = idx - c * WTB;
            const float v = (r < R && c < C) ? w[(size_t)r * C + c] : 0.f;
            const unsigned short hi = f32_to_bf16_bits(v);
            wT[idx] = hi;
            wT[CP * WTB + idx] = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
        }
    } else {
        for (int idx = tid; idx < RP * C; idx += 256) {
            const int c = idx % C, r = idx / C;
            wsm[r * WS + c] = r < R ? w[(size_t)r * C + c] : 0.f;
        }
        for (int idx = tid; idx < (RP - R) * 64; idx += 256) dzs[(R + idx / 64) * DZS + (idx & 63)] = 0.f;
    }
    constexpr int NA = K20 ? 2 : 4;   // row tiles per wave: 7 tiles (R <= 112) over 4 waves with K20, up to 16 otherwise
    f32x4 accw[NA][CTM];   // gw: row tiles rt = wv + 4a, column tiles c
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int c = 0; c < CTM; ++c) accw[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gbacc = 0.f;
    float gbr[K20 ? 25 : 1];                  // K20: per-lane partial sums of dz per owned row, reduced across the wave at the end
#pragma unroll
    for (int i = 0; i < (K20 ? 25 : 1); ++i) gbr[i] = 0.f;
    int fq[FPT], fc[FPT];                     // chunk-invariant (pixel, channel) of this thread's feature slots
#pragma unroll
    for (int j = 0; j < FPT; ++j) { const int idx = tid + 256 * j; fq[j] = idx / C; fc[j] = idx - fq[j] * C; }
    const int chunksPerM = (HW + 63) / 64;
    const int64_t nchunks = (int64_t)M * chunksPerM;
    const uint32_t kmagic = (65536u + (uint32_t)K - 1u) / (uint32_t)K;   // r / K == (r * kmagic) >> 16 for r*K < 65536
    const int px = lane;
    // Rows r = wv + 4i of (s,k) belong to wave wv: every wave streams ~R/4 coalesced 256-byte row segments of prob and
    // gprob per chunk.  The chunk's values live in registers and are fetched one chunk ahead (issued before the MFMA
    // phases of the current chunk), so the HBM latency hides behind phases 2/3.
    // NOTHING in fetch() may read a loaded value: a select or a conversion right after the load makes the compiler wait for it,
    // and 50 loads then cost 50 serial memory latencies (measured: 28 k of a chunk's 52 k cycles).  Dead pixels get an
    // out-of-range buffer offset (the hardware returns 0); features stay raw T until phase 1.
    float pr[RW], gr[RW];
    T fr[FPT];
    const int wvu = __builtin_amdgcn_readfirstlane(wv);   // wave-uniform: row offsets below stay in SGPRs
    const size_t tot = (size_t)S * M * K * HW;            // floats in prob / gprob (host checks tot*4 < 2^32)
    auto fetch = [&](int64_t ch) {
        const int m = ch / chunksPerM, p0 = (ch % chunksPerM) * 64;
        const bool live = p0 + px < HW;
        const int voff = live ? px * 4 : (int)0x80000000;   // past num_records for any chunk
        // one buffer resource per chunk (uniform base), row offsets as scalar soffsets, px as the only VGPR offset
        const size_t boff = (size_t)m * K * HW + p0;
        const uint32_t rem = (uint32_t)((tot - boff) * 4);
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(prob + boff), 0, (int)rem, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gprob + boff), 0, (int)rem, 0x00020000);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int r = min(wvu + 4 * i, R - 1);   // rows past R re-read row R-1 (branch-free issue); never stored
            const int sidx = (int)(((uint32_t)r * kmagic) >> 16);
            const uint32_t so = ((uint32_t)r + (uint32_t)sidx * (uint32_t)(M - 1) * (uint32_t)K) * (uint32_t)HW * 4u;
            pr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, voff, (int)so, 0));
            gr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, voff, (int)so, 0));
        }
        const int f = flips ? flips[m] : 0;
        const size_t fb = (size_t)src[m] * HW;
#pragma unroll
        for (int j = 0; j < FPT; ++j) {
            const int idx = tid + 256 * j, c = fc[j], q = fq[j];
            const bool ok = idx < 64 * C && p0 + q < HW;
            const int pq = ok ? p0 + q : 0, h = pq / W, wq = pq - h * W;
            fr[j] = feat[ok ? (fb + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C + c : 0];   // dead slots re-read element 0; masked at use
        }
    };
    if ((int64_t)blockIdx.x < nchunks) fetch(blockIdx.x);
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += nblk) {
        const int m = ch / chunksPerM, p0 = (ch % chunksPerM) * 64;
        const int f = flips ? flips[m] : 0;
        __syncthreads();   // previous chunk's MFMA phases are done with dzs / fs
        // ---- phase 1: dz = p*(g - <g,p>)/T
        if (K20) {
            // <g,p> per sub-head from the wave's own registers: rows r = wv + 4i, sub-head i / 5 (5 rows of each sub-head per
            // wave); the four waves' partials meet in LDS.  No p*g round trip through LDS, the work is the same on every wave.
            float part[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 25; ++i) part[i / 5] = fmaf(pr[i], gr[i], part[i / 5]);
#pragma unroll
            for (int sh = 0; sh < 5; ++sh) dots[(wvu * 5 + sh) * 64 + px] = part[sh];
        } else {
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const int r = wvu + 4 * i;
                if (r < R) dzs[r * DZS + px] = pr[i] * gr[i];
            }
        }
#pragma unroll
        for (int j = 0; j < FPT; ++j)
            if (tid + 256 * j < 64 * C) {
                if (BF) fsT[fc[j] * DZB + fq[j]] = (p0 + fq[j] < HW) ? *reinterpret_cast<const unsigned short*>(&fr[j]) : (unsigned short)0;
                else fs[fq[j] * FS + fc[j]] = (p0 + fq[j] < HW) ? to_f32(fr[j]) : 0.f;
            }
        __syncthreads();
        if (K20) {
            float dot[5];
#pragma unroll
            for (int sh = 0; sh < 5; ++sh)
                dot[sh] = (sh < S) ? ((dots[(0 * 5 + sh) * 64 + px] + dots[(1 * 5 + sh) * 64 + px]) + (dots[(2 * 5 + sh) * 64 + px] + dots[(3 * 5 + sh) * 64 + px])) : 0.f;
#pragma unroll
            for (int i = 0; i < 25; ++i) {
                const int r = wvu + 4 * i;
                const float dz = pr[i] * (gr[i] - dot[i / 5]) * invT;
                if (r < R) {
                    if (BF) {
                        const unsigned short hi = f32_to_bf16_bits(dz);
                        dzb[r * DZB + px] = hi;
                        dzb[(RPB + r) * DZB + px] = f32_to_bf16_bits(dz - bf16_bits_to_f32(hi));
                    } else {
                        dzs[r * DZS + px] = dz;
                    }
                    gbr[i] += dz;
                }
            }
            if (ch + nblk < nchunks) fetch(ch + nblk);
            __syncthreads();
        } else {
            for (int sidx = wv; sidx < S; sidx += 4) {
                float dot = 0.f;
#pragma unroll 4
                for (int k = 0; k < K; ++k) dot += dzs[(sidx * K + k) * DZS + px];
                dots[sidx * 64 + px] = dot;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const int r = wvu + 4 * i;
                const int sidx = (int)(((uint32_t)r * kmagic) >> 16);
                if (r < R) dzs[r * DZS + px] = pr[i] * (gr[i] - dots[sidx * 64 + px]) * invT;
            }
            if (ch + nblk < nchunks) fetch(ch + nblk);
            __syncthreads();
            if (tid < R) {
                float a = 0.f;
#pragma unroll 8
                for (int q = 0; q < 64; ++q) a += dzs[tid * DZS + q];
                gbacc += a;
            }
        }
        // ---- phase 2: gfeat tile [64 px][C]: wave wv owns pixel tile wv (16 px).  Operands swapped so that D^T comes out:
        // a lane holds 4 consecutive channels of one pixel -> one 8-byte (bf16) / 16-byte (fp32) store.  Plain stores: every
        // gfeat element has exactly one writer (distinct src per window, see the entry point) and the buffer arrives zeroed.
        if (gfeat) {
            f32x4 accf[CTM];
#pragma unroll
            for (int c = 0; c < CTM; ++c) accf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (BF) {
                // D[c][px] = sum_r W^T[c][r] dz[r][px]: A = W^T rows (b128), B = dz columns through the transposing LDS read
                // (lane 4*qq+pq of a 16-lane group reads 4 pixels of row k0+qq and receives its own pixel's 4 rows)
                const int qq = l15 >> 2, pq = l15 & 3;
#pragma unroll
                for (int ks = 0; ks < RPB / 32; ++ks) {
                    bf16x8h_t bz[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const unsigned short* b0 = dzb + (size_t)(pl * RPB + ks * 32 + 8 * kq + qq) * DZB + wv * 16 + 4 * pq;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(HLDS_S16X4(b0));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(HLDS_S16X4(b0 + 4 * DZB));
                        const s16x8 fv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        bz[pl] = __builtin_bit_cast(bf16x8h_t, fv);
                    }
#pragma unroll
                    for (int c = 0; c < CTM; ++c) {
                        const unsigned short* a0 = wT + (size_t)(c * 16 + l15) * WTB + ks * 32 + 8 * kq;
                        const bf16x8h_t whi = *reinterpret_cast<const bf16x8h_t*>(a0);
                        const bf16x8h_t wlo = *reinterpret_cast<const bf16x8h_t*>(a0 + CP * WTB);
                        accf[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, bz[0], accf[c], 0, 0, 0);
                        accf[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, bz[1], accf[c], 0, 0, 0);
                        accf[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, bz[0], accf[c], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll 7
                for (int ks = 0; ks < RP; ks += 4) {
                    const float av = dzs[(ks + kq) * DZS + wv * 16 + l15];
#pragma unroll
                    for (int c = 0; c < CTM; ++c)
                        if (c < CT) {
                            const float bv = wsm[(ks + kq) * WS + min(c * 16 + l15, C - 1)];
                            accf[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, accf[c], 0, 0, 0);
                        }
                }
            }
            // D^T[row = channel c*16 + kq*4 + r][col = pixel l15]
            const int pq = p0 + wv * 16 + l15;
            if (pq < HW) {
                const int h = pq / W, wq = pq % W;
                T* gp = gfeat + ((size_t)src[m] * HW + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
#pragma unroll
                for (int c = 0; c < CTM; ++c) {
                    const int cc = c * 16 + kq * 4;
                    if (c < CT && accumulate) {      // gfeat holds another consumer's gradient of the feature (ops._GradJoin): add, round once
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (cc + r < C) gp[cc + r] = from_f32<T>(accf[c][r] + to_f32(gp[cc + r]));
                    } else if (c < CT && cc + 3 < C) {
                        T pk[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) pk[r] = from_f32<T>(accf[c][r]);
                        if (sizeof(T) == 2) *reinterpret_cast<uint2*>(gp + cc) = *reinterpret_cast<const uint2*>(pk);
                        else *reinterpret_cast<uint4*>(gp + cc) = *reinterpret_cast<const uint4*>(pk);
                    } else if (c < CT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (cc + r < C) gp[cc + r] = from_f32<T>(accf[c][r]);
                    }
                }
            }
        }
        // ---- phase 3: gw += dz * f
        if (BF) {
            // D[r][c] = sum_px dz[r][px] f[px][c]: A = dz rows (hi, lo planes), B = transposed features, both plain b128 reads
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8h_t bf_[CTM];
#pragma unroll
                for (int c = 0; c < CTM; ++c) bf_[c] = *reinterpret_cast<const bf16x8h_t*>(fsT + (size_t)(c * 16 + l15) * DZB + ks * 32 + 8 * kq);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int rt = wv + 4 * a;           // 7 row tiles over 4 waves
                    if (rt < RT) {
                        const unsigned short* a0 = dzb + (size_t)(rt * 16 + l15) * DZB + ks * 32 + 8 * kq;
                        const bf16x8h_t dhi = *reinterpret_cast<const bf16x8h_t*>(a0);
                        const bf16x8h_t dlo = *reinterpret_cast<const bf16x8h_t*>(a0 + (size_t)RPB * DZB);
#pragma unroll
                        for (int c = 0; c < CTM; ++c) {
                            accw[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dlo, bf_[c], accw[a][c], 0, 0, 0);
                            accw[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dhi, bf_[c], accw[a][c], 0, 0, 0);
                        }
                    }
                }
            }
        } else
#pragma unroll 4
        for (int ks = 0; ks < 64; ks += 4) {
            float bfr[CTM];
#pragma unroll
            for (int c = 0; c < CTM; ++c) bfr[c] = (c < CT) ? fs[(ks + kq) * FS + min(c * 16 + l15, C - 1)] : 0.f;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int rt = wv + 4 * a;
                if (rt < RT) {
                    const float av = dzs[(rt * 16 + l15) * DZS + ks + kq];
#pragma unroll
                    for (int c = 0; c < CTM; ++c)
                        if (c < CT) accw[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfr[c], accw[a][c], 0, 0, 0);
                }
            }
        }
    }
    float* out = partials + (size_t)blockIdx.x * (R * C + R);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int rt = wv + 4 * a;
        if (rt < RT) {
#pragma unroll
            for (int c = 0; c < CTM; ++c)
                if (c < CT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = rt * 16 + kq * 4 + r, col = c * 16 + l15;
                        if (row < R && col < C) out[row * C + col] = accw[a][c][r];
                    }
                }
        }
    }
    if (K20) {
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            const float v = wave_sum(gbr[i]);
            if (lane == 0 && wvu + 4 * i < R) out[R * C + wvu + 4 * i] = v;
        }
    } else if (tid < R) out[R * C + tid] = gbacc;
}

#ifndef MISEG_HB_ABL
#define MISEG_HB_ABL 0   // timing ablations of head_local_bwd_wave_kernel (scratch builds only): 1 no gw/gb, 2 no gfeat, 3 no MFMA phase
#endif
// Wave-local backward of the local head for the shipped 16-channel tap (bf16 features, K = 20, S <= 5).  The fused kernel above
// works a 64-pixel chunk per BLOCK through three phases with workgroup barriers between them; here every WAVE owns its chunks
// outright and nothing but its own LDS slice is shared, so there is no barrier in the loop (a wave's LDS traffic is in order):
//   per sub-head: p, g of the lane's pixel (coalesced along pixels) -> dz = p (g - <g,p>) / T in registers -> hi + lo bf16 planes,
//                 written to the wave's LDS as [px][k] rows: read straight as the operand of gfeat, through the transposing LDS read
//                 (ds_read_b64_tr_b16) as the [k][px] operand of gw, gb;
//   gfeat^T[c][px] += W^T[c][k] dz^T[k][px]   3 x mfma_16x16x32_bf16 per 16-pixel tile (W_lo dz_hi + W_hi dz_lo + W_hi dz_hi)
//   gw[k][c]       += dz^T[k][px] f[px][c]    2 per (32-pixel k-chunk, class tile): dz_lo f + dz_hi f (features are exact bf16)
//   gb[k]          += dz^T[k][px] 1           the same A fragments against a constant all-ones B fragment
// gw / gb stay in accumulators over all the wave's chunks; one deterministic partial per block at the end (same workspace and
// final reduction as the fused kernel).  Precision is the fused BF variant's: hi + lo = 2^-16 relative per term.
// RECOMP: the probabilities are not read back but computed again from the features, the way the forward kernel computes them
// (W in three bf16 planes x features on the MFMA, bias, exp2 softmax: the same operations in the same order, so p is the
// forward's p bit for bit); `prob` is unused and the kernel reads S*K*4 instead of 2*S*K*4 bytes per pixel.  The logits cross from the
// D layout (4 classes x 1 pixel per lane) to one pixel per lane through the [k][px] dz rows' LDS, which are free at that point.
template <int C, bool RECOMP>
__global__ __launch_bounds__(256, 2) void head_local_bwd_wave_kernel(const bf16* __restrict__ feat, int H, int W,
                                                                     const int32_t* __restrict__ src, const int32_t* __restrict__ flips,
                                                                     int M, const float* __restrict__ w, const float* __restrict__ bias, int S,
                                                                     float invT, const float* __restrict__ prob, const float* __restrict__ gprob,
                                                                     bf16* __restrict__ gfeat, float* __restrict__ partials, int accumulate) {
    constexpr int K = 20, SM = 5, CT = C / 16, AR = 24, BR = 72, ZS = 68, CQ = C / 4;
    constexpr int ZT = RECOMP ? K * ZS * 2 : 0;                          // u16 units of the logits' transpose buffer
    constexpr int WAVE_LDS = (2 * 64 * AR + ZT + C * BR) * 2;            // bytes per wave
    static_assert(!RECOMP || C == 16, "the recomputing form is the 16-channel tap's");
    typedef typename HeadFrag<C>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char hb[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    unsigned short* dzA = reinterpret_cast<unsigned short*>(hb + (size_t)wv * WAVE_LDS);   // [2][64][AR]  hi | lo, row = pixel
    float* ztw = reinterpret_cast<float*>(dzA + 2 * 64 * AR);                              // RECOMP: [K][ZS] logits of the chunk
    unsigned short* fT = dzA + 2 * 64 * AR + ZT;                                           // [C][BR]      features, row = channel
    unsigned short* wT = reinterpret_cast<unsigned short*>(hb + (size_t)4 * WAVE_LDS);     // [S][2][C][32] W^T hi | lo, classes >= K zero
    bf16* wp = reinterpret_cast<bf16*>(wT + (size_t)SM * 2 * C * 32);                       // RECOMP: [S][3][K][C] W in three planes
    float* bs = reinterpret_cast<float*>(wp + (size_t)SM * 3 * K * C);                      // RECOMP: [S][K] bias
    const int HW = H * W, R = S * K;
    if constexpr (RECOMP) {
        for (int i = tid; i < S * K * C; i += 256) {      // the forward kernel's split, value for value
            const float v = w[i];
            const bf16 h1 = __float2bfloat16(v);
            const float r1 = v - __bfloat162float(h1);
            const bf16 h2 = __float2bfloat16(r1);
            const bf16 h3 = __float2bfloat16(r1 - __bfloat162float(h2));
            const int s = i / (K * C), rem = i - s * K * C;
            bf16* d = wp + (size_t)(s * 3) * K * C + rem;
            d[0] = h1, d[(size_t)K * C] = h2, d[(size_t)2 * K * C] = h3;
        }
        for (int i = tid; i < S * K; i += 256) bs[i] = bias[i];
    }
    for (int i = lane; i < 2 * 64 * AR / 2; i += 64) reinterpret_cast<unsigned*>(dzA)[i] = 0u;   // the pad columns 20..23 stay zero
    for (int i = tid; i < S * 2 * C * 32; i += 256) {
        const int k = i & 31, c = (i >> 5) % C, pl = (i / (32 * C)) & 1, s = i / (64 * C);
        const float v = k < K ? w[((size_t)s * K + k) * C + c] : 0.f;
        const unsigned short hi = f32_to_bf16_bits(v);
        wT[i] = pl ? f32_to_bf16_bits(v - bf16_bits_to_f32(hi)) : hi;
    }
    __syncthreads();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 agw[SM][2][CT], agb[SM][2];
#pragma unroll
    for (int s = 0; s < SM; ++s)
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
            agb[s][cl] = zero4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) agw[s][cl][ct] = zero4;
        }
    bf16x8h_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const int cps = (HW + 63) / 64, total = M * cps, nw = gridDim.x * 4;
    // p, g of the NEXT (chunk, sub-head) are fetched while the current one is worked on: P/G[s & 1] is sub-head s's buffer
    // (S = 5 is odd, so the buffer a chunk ends on is copied down once per chunk)
    float P[2][K], G[2][K];
    {
        const int g0 = min(blockIdx.x * 4 + wv, total - 1), m0 = g0 / cps, pc0 = min((g0 - m0 * cps) * 64 + lane, HW - 1);
        const size_t o = ((size_t)m0 * K) * HW + pc0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (!RECOMP) P[0][k] = prob[o + (size_t)k * HW];
            G[0][k] = gprob[o + (size_t)k * HW];
        }
    }
    for (int g = blockIdx.x * 4 + wv; g < total; g += nw) {
        const int m = g / cps, px0 = (g - m * cps) * 64;
        const int gn = min(g + nw, total - 1), mn = gn / cps, pcn = min((gn - mn * cps) * 64 + lane, HW - 1);
        const int f = flips ? flips[m] : 0;
        const int px = px0 + lane, pc = min(px, HW - 1);
        const float invT_l = px < HW ? invT : 0.f;      // pixels past the end of the sample: dz = 0
        const size_t sbase = (size_t)src[m] * HW;
        {
            const int h = pc / W, wq = pc - h * W;
            const bf16* fp = feat + (sbase + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
#pragma unroll
            for (int c0 = 0; c0 < C; c0 += 8) {
                const s16x8 v = *reinterpret_cast<const s16x8*>(fp + c0);
#pragma unroll
                for (int e = 0; e < 8; ++e) fT[(c0 + e) * BR + lane] = (unsigned short)v[e];
            }
        }
        frag_t fb[4];      // RECOMP: the features as MFMA B operands, pixel t*16 + l15, channels 4q..4q+3
        if constexpr (RECOMP) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pt = min(px0 + t * 16 + l15, HW - 1), h = pt / W, wq = pt - h * W;
                fb[t] = *reinterpret_cast<const frag_t*>(feat + (sbase + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C + q * CQ);
            }
        }
        f32x4 agf[4][CT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) agf[t][ct] = zero4;
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            {   // S == SM: the launcher takes this kernel for five sub-heads only, so the loop is straight-line code
                {
                    const size_t o = s + 1 < SM ? (((size_t)(s + 1) * M + m) * K) * HW + pc : ((size_t)mn * K) * HW + pcn;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if constexpr (!RECOMP) P[(s + 1) & 1][k] = prob[o + (size_t)k * HW];
                        G[(s + 1) & 1][k] = gprob[o + (size_t)k * HW];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RECOMP) {
                    frag_t wa[2][3];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const int cls = min(ct * 16 + l15, K - 1);      // class rows >= K: a copy of row K-1; their D rows are never read
                            wa[ct][pl] = *reinterpret_cast<const frag_t*>(wp + ((size_t)(s * 3 + pl) * K + cls) * C + q * CQ);
                        }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4 a0 = zero4, a1 = zero4;
#pragma unroll
                        for (int pl = 2; pl >= 0; --pl) {   // smallest plane first, as in the forward
                            a0 = HeadFrag<C>::mma(wa[0][pl], fb[t], a0);
                            a1 = HeadFrag<C>::mma(wa[1][pl], fb[t], a1);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) ztw[(4 * q + r) * ZS + t * 16 + l15] = a0[r];
                        if (q == 0)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ztw[(16 + r) * ZS + t * 16 + l15] = a1[r];
                    }
                    __builtin_amdgcn_wave_barrier();
                    float* z = P[s & 1];
#pragma unroll
                    for (int k4 = 0; k4 < K; k4 += 4) {
                        const float4 bv = *reinterpret_cast<const float4*>(bs + s * K + k4);      // one address for the wave: a broadcast
                        z[k4] = ztw[k4 * ZS + lane] + bv.x, z[k4 + 1] = ztw[(k4 + 1) * ZS + lane] + bv.y;
                        z[k4 + 2] = ztw[(k4 + 2) * ZS + lane] + bv.z, z[k4 + 3] = ztw[(k4 + 3) * ZS + lane] + bv.w;
                    }
                    __builtin_amdgcn_wave_barrier();
                    float mx = -3.4e38f;
#pragma unroll
                    for (int k = 0; k < K; ++k) mx = fmaxf(mx, z[k]);
                    const float sc2 = invT * 1.4426950408889634f, off2 = -mx * sc2;
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        z[k] = __builtin_amdgcn_exp2f(fmaf(z[k], sc2, off2));
                        sum += z[k];
                    }
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int k = 0; k < K; ++k) z[k] *= inv;
                }
                const float* p = P[s & 1];
                const float* gg = G[s & 1];
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) dot = fmaf(gg[k], p[k], dot);
                unsigned hw[K / 2], lw[K / 2];   // bf16 pairs (classes 2i, 2i+1): hi plane, lo plane
#pragma unroll
                for (int k = 0; k < K; k += 2) {
                    unsigned short h2[2], l2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float dz = p[k + e] * (gg[k + e] - dot) * invT_l;
                        h2[e] = f32_to_bf16_bits(dz);
                        l2[e] = f32_to_bf16_bits(dz - bf16_bits_to_f32(h2[e]));
                    }
                    hw[k / 2] = (unsigned)h2[0] | ((unsigned)h2[1] << 16);
                    lw[k / 2] = (unsigned)l2[0] | ((unsigned)l2[1] << 16);
                }
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const unsigned* v = pl ? lw : hw;
                    unsigned* row = reinterpret_cast<unsigned*>(dzA + (pl * 64 + lane) * AR);
                    *reinterpret_cast<uint4*>(row) = make_uint4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<uint4*>(row + 4) = make_uint4(v[4], v[5], v[6], v[7]);
                    *reinterpret_cast<uint2*>(row + 8) = make_uint2(v[8], v[9]);
                }
                __builtin_amdgcn_wave_barrier();
                // gfeat^T tiles: A = W^T [c][k], B = dz^T [k][px]
#if MISEG_HB_ABL != 2 && MISEG_HB_ABL != 3
                bf16x8h_t wh[CT], wl[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    wh[ct] = *reinterpret_cast<const bf16x8h_t*>(wT + ((size_t)(s * 2 + 0) * C + ct * 16 + l15) * 32 + 8 * q);
                    wl[ct] = *reinterpret_cast<const bf16x8h_t*>(wT + ((size_t)(s * 2 + 1) * C + ct * 16 + l15) * 32 + 8 * q);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // classes 24..31 (q == 3): no such columns in the row -- any finite values do, W^T is zero there
                    const bf16x8h_t bh = *reinterpret_cast<const bf16x8h_t*>(dzA + (t * 16 + l15) * AR + 8 * min(q, 2));
                    const bf16x8h_t bl = *reinterpret_cast<const bf16x8h_t*>(dzA + (64 + t * 16 + l15) * AR + 8 * min(q, 2));
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        agf[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ct], bh, agf[t][ct], 0, 0, 0);
                        agf[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ct], bl, agf[t][ct], 0, 0, 0);
                        agf[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ct], bh, agf[t][ct], 0, 0, 0);
                    }
                }
#endif
#if MISEG_HB_ABL != 1 && MISEG_HB_ABL != 3
                // gw, gb: A = dz^T [k][px], B = f [px][c] (or ones)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int cl = 0; cl < 2; ++cl) {
                        // class l15 of tile cl, pixels kc*32 + 8q .. +7: out of the [px][k] rows by the transposing read (lane 4qq+pq of a
                        // 16-lane group supplies 4 classes of pixel qq and receives its own class's 4 pixels).  Tile 1 reads columns
                        // 16..31 of 24-column rows: classes 24..31 are the next row's first values -- their D rows are never stored.
                        const unsigned short* a0 = dzA + (kc * 32 + 8 * q + (l15 >> 2)) * AR + cl * 16 + 4 * (l15 & 3);
                        bf16x8h_t ah, al;
                        {
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(HLDS_S16X4(a0));
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(HLDS_S16X4(a0 + 4 * AR));
                            const s16x8 fv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            ah = __builtin_bit_cast(bf16x8h_t, fv);
                        }
                        {
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(HLDS_S16X4(a0 + 64 * AR));
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(HLDS_S16X4(a0 + 64 * AR + 4 * AR));
                            const s16x8 fv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            al = __builtin_bit_cast(bf16x8h_t, fv);
                        }
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            const bf16x8h_t fb = *reinterpret_cast<const bf16x8h_t*>(fT + (ct * 16 + l15) * BR + kc * 32 + 8 * q);
                            agw[s][cl][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, fb, agw[s][cl][ct], 0, 0, 0);
                            agw[s][cl][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, fb, agw[s][cl][ct], 0, 0, 0);
                        }
                        agb[s][cl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ones, agb[s][cl], 0, 0, 0);
                        agb[s][cl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ones, agb[s][cl], 0, 0, 0);
                    }
#endif
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (gfeat) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pt = px0 + t * 16 + l15;
                if (pt < HW) {
                    const int h = pt / W, wq = pt - h * W;
                    bf16* gp = gfeat + (sbase + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        s16x4 o;
                        if (accumulate) {      // gfeat already holds another consumer's gradient of the same feature: add, round once
                            const s16x4 old = *reinterpret_cast<const s16x4*>(gp + ct * 16 + 4 * q);
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (short)f32_to_bf16_bits(agf[t][ct][r] + bf16_bits_to_f32((unsigned short)old[r]));
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (short)f32_to_bf16_bits(agf[t][ct][r]);
                        }
                        *reinterpret_cast<s16x4*>(gp + ct * 16 + 4 * q) = o;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (!RECOMP) P[0][k] = P[SM & 1][k];
            G[0][k] = G[SM & 1][k];
        }
    }
    // one partial per block: the four waves' accumulators through LDS, summed in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(hb + (size_t)wv * WAVE_LDS);
    const int len = R * C + R;
    for (int i = lane; i < len; i += 64) red[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < SM; ++s)
#pragma unroll
        for (int cl = 0; cl < 2; ++cl)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = cl * 16 + 4 * q + r;
                if (k < K) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) red[(s * K + k) * C + ct * 16 + l15] = agw[s][cl][ct][r];
                    if (l15 == 0) red[R * C + s * K + k] = agb[s][cl][r];
                }
            }
    __syncthreads();
    for (int i = tid; i < len; i += 256) {
        float a = 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) a += reinterpret_cast<const float*>(hb + (size_t)x * WAVE_LDS)[i];
        partials[(size_t)blockIdx.x * len + i] = a;
    }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partials, int nparts, int len, int n0, float* __restrict__ d0,
                                                           float* __restrict__ d1) {
    reduce_partials_block(partials, nparts, (size_t)len, len, Split2Out{d0, n0, d1}, [](int e) { return (size_t)e; });
}

}  // namespace miseg

using namespace miseg;

template <int C> static size_t head_mfma_lds(int64_t S) { return (size_t)(kHT / 64) * 20 * kHeadZS * 4 + (size_t)S * 3 * 20 * head_mfma_cp<C>() * 2; }
extern "C" int miseg_head_local_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                    const int32_t* src, const int32_t* flips, int64_t M, const float* w, const float* b, int64_t S,
                                    int64_t K, float T, float* prob, float simplex_tol, int32_t* simplex_violations) {
    MISEG_TAPE(miseg_head_local_fwd, stream, dt, feat, B, H, W, C, src, flips, M, w, b, S, K, T, prob, simplex_tol, simplex_violations);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_local_fwd, stream, MISEG_BF16, feat, B, H, W, C, src, flips, M, w, b, S, K, T, prob, simplex_tol, simplex_violations);
    MISEG_REQUIRE(feat && src && w && b && prob, "head_local_fwd: null pointer");
    MISEG_REQUIRE(C > 0 && C % 4 == 0 && K > 0 && K <= 64 && M > 0 && S > 0 && H > 0 && W > 0, "head_local_fwd: need C%%4==0, K<=64");
    MISEG_REQUIRE(simplex_violations == nullptr || K <= 32, "head_local_fwd: the fused simplex check needs K <= 32");
    dim3 grid((unsigned)cdiv(H * W, kHT), (unsigned)M);
    size_t ldsb = (size_t)K * kHT * 4;
    hipStream_t st = as_stream(stream);
    if (dt == MISEG_BF16 && K == 20 && (C == 16 || C == 32) && (H * W) % 4 == 0 && S * K * C <= 3200) {
        const dim3 gridm((unsigned)cdiv(H * W, kHT * 4), (unsigned)M);
        if (C == 32)
            hipLaunchKernelGGL(head_local_fwd_mfma_kernel<32>, gridm, dim3(kHT), head_mfma_lds<32>(S), st, (const bf16*)feat, (int)H, (int)W,
                               src, flips, (int)M, w, b, (int)S, 1.0f / T, prob, simplex_tol, simplex_violations);
        else
            hipLaunchKernelGGL(head_local_fwd_mfma_kernel<16>, gridm, dim3(kHT), head_mfma_lds<16>(S), st, (const bf16*)feat, (int)H, (int)W,
                               src, flips, (int)M, w, b, (int)S, 1.0f / T, prob, simplex_tol, simplex_violations);
    } else
    if (K <= 32 && (dt == MISEG_F32 || dt == MISEG_BF16)) {
        const bool quad = (H * W) % 4 == 0 && W % 4 == 0;
        const dim3 gridq((unsigned)cdiv(H * W, kHT * 4), (unsigned)M);
#define HLF(TT, KPP)                                                                                                                   \
    {                                                                                                                                 \
        if (quad && K == KPP) hipLaunchKernelGGL((head_local_fwd_reg_kernel<TT, KPP, 4, true>), gridq, dim3(kHT), (size_t)(S * K * C * 4), st, (const TT*)feat, (int)H, (int)W, (int)C, src, flips, (int)M, w, b, (int)S, (int)K, 1.0f / T, prob, simplex_tol, simplex_violations); \
        else if (quad) hipLaunchKernelGGL((head_local_fwd_reg_kernel<TT, KPP, 4, false>), gridq, dim3(kHT), (size_t)(S * K * C * 4), st, (const TT*)feat, (int)H, (int)W, (int)C, src, flips, (int)M, w, b, (int)S, (int)K, 1.0f / T, prob, simplex_tol, simplex_violations); \
        else hipLaunchKernelGGL((head_local_fwd_reg_kernel<TT, KPP, 1, false>), grid, dim3(kHT), (size_t)(S * K * C * 4), st, (const TT*)feat, (int)H, (int)W, (int)C, src, flips, (int)M, w, b, (int)S, (int)K, 1.0f / T, prob, simplex_tol, simplex_violations); \
    }
#define HLF_K(TT) switch ((K + 3) / 4) { case 1: HLF(TT, 4); break; case 2: HLF(TT, 8); break; case 3: HLF(TT, 12); break; case 4: HLF(TT, 16); break; \
                                         case 5: HLF(TT, 20); break; case 6: HLF(TT, 24); break; case 7: HLF(TT, 28); break; default: HLF(TT, 32); break; }
        if (dt == MISEG_F32) HLF_K(float) else HLF_K(bf16)
#undef HLF_K
#undef HLF
    } else
    if (dt == MISEG_F32)
        hipLaunchKernelGGL(head_local_fwd_kernel<float>, grid, dim3(kHT), ldsb, st, (const float*)feat, (int)H, (int)W, (int)C, src,
                           flips, (int)M, w, b, (int)S, (int)K, 1.0f / T, prob);
    else if (dt == MISEG_BF16)
        hipLaunchKernelGGL(head_local_fwd_kernel<bf16>, grid, dim3(kHT), ldsb, st, (const bf16*)feat, (int)H, (int)W, (int)C, src,
                           flips, (int)M, w, b, (int)S, (int)K, 1.0f / T, prob);
    else
        return fail(MISEG_E_INVALID, "head_local_fwd: bad dtype %d", dt);
    MISEG_LAUNCH_CHECK("head_local_fwd_kernel");
    return MISEG_OK;
}

static int head_w_blocks(int64_t M, int64_t HW) { return (int)std::min<int64_t>(M * cdiv(HW, 64), 768); }

extern "C" int64_t miseg_head_local_bwd_ws_bytes(int64_t M, int64_t H, int64_t W, int64_t C, int64_t S, int64_t K) {
    return (((int64_t)head_w_blocks(M, H * W) + 1) * (S * K * C + S * K)) * 4;
}

static bool head_bwd_wave_shape(int dt, int64_t C, int64_t S, int64_t K) { return dt == MISEG_BF16 && K == 20 && C == 16 && S == 5; }

extern "C" int64_t miseg_head_local_bwd_acc_supported(int dt, int64_t C, int64_t S, int64_t K) {
    (void)dt; (void)S; (void)K;
    return C % 4 == 0;      // every kernel behind miseg_head_local_bwd has the accumulating store (each gfeat element has one writer)
}

static int head_local_bwd_impl(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src,
                               const int32_t* flips, int64_t M, const float* w, const float* bias, int64_t S, int64_t K, float T,
                               const float* prob, const float* gprob, void* gfeat, float* gw, float* gb, void* ws, int64_t ws_bytes, int accumulate);

extern "C" int miseg_head_local_bwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                    const int32_t* src, const int32_t* flips, int64_t M, const float* w, int64_t S, int64_t K,
                                    float T, const float* prob, const float* gprob, void* gfeat, float* gw, float* gb, void* ws,
                                    int64_t ws_bytes) {
    MISEG_TAPE(miseg_head_local_bwd, stream, dt, feat, B, H, W, C, src, flips, M, w, S, K, T, prob, gprob, gfeat, gw, gb, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_local_bwd, stream, MISEG_BF16, feat, B, H, W, C, src, flips, M, w, S, K, T, prob, gprob, gfeat, gw, gb, ws, ws_bytes);
    return head_local_bwd_impl(stream, dt, feat, B, H, W, C, src, flips, M, w, nullptr, S, K, T, prob, gprob, gfeat, gw, gb, ws, ws_bytes, 0);
}

// gfeat_rows holds rows [row0, row0 + rows) of the batch only (src[] must lie inside): see miseg_hip.h
extern "C" int miseg_head_local_bwd_rows(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                         const int32_t* src, const int32_t* flips, int64_t M, const float* w, int64_t S, int64_t K,
                                         float T, const float* prob, const float* gprob, void* gfeat_rows, int64_t row0, float* gw, float* gb,
                                         void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_head_local_bwd_rows, stream, dt, feat, B, H, W, C, src, flips, M, w, S, K, T, prob, gprob, gfeat_rows, row0, gw, gb, ws, ws_bytes);
    MISEG_REQUIRE(row0 >= 0 && row0 < B, "head_local_bwd_rows: bad first row");
    const int64_t es = dt == MISEG_F32 ? 4 : 2;
    void* base = gfeat_rows ? static_cast<char*>(gfeat_rows) - row0 * H * W * C * es : nullptr;      // row src[m] of the batch = row src[m] - row0 here
    return miseg_head_local_bwd(stream, dt, feat, B, H, W, C, src, flips, M, w, S, K, T, prob, gprob, base, gw, gb, ws, ws_bytes);
}

extern "C" int64_t miseg_head_local_bwd_recompute_supported(int dt, int64_t C, int64_t S, int64_t K) {
    return head_bwd_wave_shape(dt == MISEG_F16 ? (int)MISEG_BF16 : dt, C, S, K);
}

// miseg_head_local_bwd_rows without the probabilities: the kernel computes them again from feat, w, bias (see miseg_hip.h)
extern "C" int miseg_head_local_bwd_recompute(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                              const int32_t* src, const int32_t* flips, int64_t M, const float* w, const float* bias, int64_t S,
                                              int64_t K, float T, const float* gprob, void* gfeat_rows, int64_t row0, float* gw, float* gb,
                                              void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_head_local_bwd_recompute, stream, dt, feat, B, H, W, C, src, flips, M, w, bias, S, K, T, gprob, gfeat_rows, row0, gw, gb, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_local_bwd_recompute, stream, MISEG_BF16, feat, B, H, W, C, src, flips, M, w, bias, S, K, T, gprob, gfeat_rows, row0, gw, gb, ws, ws_bytes);
    MISEG_REQUIRE(bias, "head_local_bwd_recompute: null bias");
    MISEG_REQUIRE(row0 >= 0 && row0 < B, "head_local_bwd_recompute: bad first row");
    void* base = gfeat_rows ? static_cast<char*>(gfeat_rows) - row0 * H * W * C * 2 : nullptr;
    return head_local_bwd_impl(stream, dt, feat, B, H, W, C, src, flips, M, w, bias, S, K, T, nullptr, gprob, base, gw, gb, ws, ws_bytes, 0);
}

extern "C" int miseg_head_local_bwd_acc(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                        const int32_t* src, const int32_t* flips, int64_t M, const float* w, int64_t S, int64_t K,
                                        float T, const float* prob, const float* gprob, void* gfeat_inout, float* gw, float* gb, void* ws,
                                        int64_t ws_bytes) {
    MISEG_TAPE(miseg_head_local_bwd_acc, stream, dt, feat, B, H, W, C, src, flips, M, w, S, K, T, prob, gprob, gfeat_inout, gw, gb, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_local_bwd_acc, stream, MISEG_BF16, feat, B, H, W, C, src, flips, M, w, S, K, T, prob, gprob, gfeat_inout, gw, gb, ws, ws_bytes);
    MISEG_REQUIRE(gfeat_inout, "head_local_bwd_acc: null gradient tensor");
    return head_local_bwd_impl(stream, dt, feat, B, H, W, C, src, flips, M, w, nullptr, S, K, T, prob, gprob, gfeat_inout, gw, gb, ws, ws_bytes, 1);
}

static int head_local_bwd_impl(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src,
                               const int32_t* flips, int64_t M, const float* w, const float* bias, int64_t S, int64_t K, float T,
                               const float* prob, const float* gprob, void* gfeat, float* gw, float* gb, void* ws, int64_t ws_bytes, int accumulate) {
    MISEG_REQUIRE(feat && src && w && (prob || bias) && gprob && gw && gb && ws, "head_local_bwd: null pointer");
    MISEG_REQUIRE(!bias || head_bwd_wave_shape(dt, C, S, K), "head_local_bwd_recompute: bf16 / f16 features, C = 16, S = 5, K = 20 only");
    MISEG_REQUIRE(C > 0 && C % 4 == 0 && C <= 128 && K > 0 && K <= 64 && S * K <= 256 && M > 0, "head_local_bwd: need C%%4==0, C<=128, S*K<=256");
    MISEG_REQUIRE(ws_bytes >= miseg_head_local_bwd_ws_bytes(M, H, W, C, S, K), "head_local_bwd: workspace too small");
    MISEG_REQUIRE(S * M * K * H * W * 4 < ((int64_t)1 << 32), "head_local_bwd: prob larger than 4 GiB (32-bit buffer offsets)");
    hipStream_t st = as_stream(stream);
    float* partials = (float*)ws;
    const int nblk = head_w_blocks(M, H * W), R = (int)(S * K), RT = (R + 15) / 16, RP = RT * 16;
    int nused = nblk;
    size_t lds = ((size_t)RP * 65 + (size_t)64 * (C + 1) + (size_t)RP * (C + 1) + (size_t)4 * std::max<int64_t>(S, 5) * 64) * 4;
    {   // the bf16-MFMA variant's layout: dz planes [2][128][72], features^T [CP][72], W^T planes [2][CP][136] (bf16) + dots
        const size_t cp = (size_t)((C + 15) / 16) * 16;
        lds = std::max(lds, ((size_t)2 * 128 * 72 + cp * 72 + 2 * cp * 136) * 2 + (size_t)4 * 5 * 64 * 4);
    }
    MISEG_REQUIRE(lds <= 150 * 1024, "head_local_bwd: S*K*C too large for LDS");
    if (head_bwd_wave_shape(dt, C, S, K)) {
        constexpr int wave_lds = (2 * 64 * 24 + 16 * 72) * 2, wave_lds_r = wave_lds + 20 * 68 * 4;      // the kernel's WAVE_LDS
        const size_t wl = (size_t)4 * wave_lds + (size_t)S * 2 * 16 * 32 * 2;
        const size_t wlr = (size_t)4 * wave_lds_r + (size_t)S * 2 * 16 * 32 * 2 + (size_t)S * 3 * 20 * 16 * 2 + (size_t)S * 20 * 4;   // + W in three planes, bias
        hipFuncSetAttribute(bias ? (const void*)head_local_bwd_wave_kernel<16, true> : (const void*)head_local_bwd_wave_kernel<16, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bias ? wlr : wl));
        // one block per CU: alone the kernel runs as fast with 256 blocks as with 768 (it is bound by the p / g read pattern, 256 B
        // per plane and wave, not by occupancy -- DESIGN.md section 7), and the smaller footprint leaves LDS and registers to the
        // main-stream kernels this side-stream kernel runs next to.
        // The recomputing form moves half the bytes and issues a third more instructions per chunk: two blocks per CU (what its LDS allows).
        static const int nb_cap = [] { const char* e = getenv("MISEG_HEAD_BWD_BLOCKS"); return e ? std::max(1, atoi(e)) : 0; }();   // diagnostic
        const int nb = std::min(nb_cap ? nb_cap : bias ? 512 : 256, nblk);
        nused = nb;   // the final reduction reads this kernel's nb partials only
        if (bias)
            hipLaunchKernelGGL((head_local_bwd_wave_kernel<16, true>), dim3(nb), dim3(256), wlr, st, (const bf16*)feat, (int)H, (int)W, src, flips,
                               (int)M, w, bias, (int)S, 1.0f / T, prob, gprob, (bf16*)gfeat, partials, accumulate);
        else
            hipLaunchKernelGGL((head_local_bwd_wave_kernel<16, false>), dim3(nb), dim3(256), wl, st, (const bf16*)feat, (int)H, (int)W, src, flips,
                               (int)M, w, bias, (int)S, 1.0f / T, prob, gprob, (bf16*)gfeat, partials, accumulate);
    } else
#define HLB2(TT, CTM, RW, K20V, BFV)                                                                                                         \
    {                                                                                                                             \
        hipFuncSetAttribute((const void*)head_local_bwd_fused_kernel<TT, CTM, RW, K20V, BFV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((head_local_bwd_fused_kernel<TT, CTM, RW, K20V, BFV>), dim3(nblk), dim3(256), lds, st, (const TT*)feat, (int)H, (int)W, (int)C, \
                           src, flips, (int)M, w, (int)S, (int)K, 1.0f / T, prob, gprob, (TT*)gfeat, partials, nblk, accumulate); \
    }
#define HLB(TT, CTM) { if (K == 20 && R <= 100 && sizeof(TT) == 2 && CTM == 2 && C == 32) HLB2(TT, CTM, 25, true, true)   /* 16 channels: HBM-bound either way and the bf16 variant does not fit 3 blocks per CU */ else if (K == 20 && R <= 100) HLB2(TT, CTM, 25, true, false) else if (R <= 112) HLB2(TT, CTM, 28, false, false) else HLB2(TT, CTM, 64, false, false) }
#define HLB_C(TT) { if (C <= 16) HLB(TT, 1) else if (C <= 32) HLB(TT, 2) else if (C <= 64) HLB(TT, 4) else HLB(TT, 8) }
    if (dt == MISEG_F32) HLB_C(float)
    else if (dt == MISEG_BF16) HLB_C(bf16)
    else return fail(MISEG_E_INVALID, "head_local_bwd: bad dtype %d", dt);
#undef HLB_C
#undef HLB
#undef HLB2
    MISEG_LAUNCH_CHECK("head_local_bwd_fused_kernel");
    const int len = R * (int)C + R;
    // the blocks' [gw | gb] partial vectors, summed straight into the two gradients
    hipLaunchKernelGGL(sum_partials_kernel, dim3(reduce_grid(len, nused)), dim3(256), 0, st, partials, nused, len, R * (int)C, gw, gb);
    MISEG_LAUNCH_CHECK("sum_partials_kernel");
    return MISEG_OK;
}
