// Cluster-head VARIANTS the shipped config does not use (config/semi.yaml:45-55 ships head_type=linear, normalize=false) but the
// reference implements (ref: contrastyou/trainer/_utils.py:96-168):
//   head_type = "mlp" : Linear/1x1-conv(C -> HID) -> LeakyReLU(0.01) -> Linear/1x1-conv(HID -> K)     (HID = 128 global, 64 local)
//   normalize = true  : logits L2-normalised over the class axis (F.normalize, eps 1e-12) before softmax(. / T)
// for both the pooled ("global", ClusterHead) and the per-pixel ("local", LocalClusterHead) head, forward and backward, with
// the same fused sample gather / flip replay as the linear kernels of heads.hip / mi_global.hip.  Generic in C, HID, K (no
// template specialisation per shape: these are not on the bench path); fp32 arithmetic in channel order; parameter gradients
// as deterministic per-block partials + a fixed-order reduce.  HID == 0 selects the single-layer head (w1 = [S][K][C]).
#include "common.h"

namespace miseg {

constexpr int kVT = 64;            // one wave = one 64-pixel chunk: its LDS columns need no workgroup barrier
constexpr float kLeaky = 0.01f;    // nn.LeakyReLU(0.01)
constexpr float kNormEps = 1e-12f; // F.normalize default eps

// ---- shared per-pixel math (column `t` of the LDS arrays) ------------------------------------------------------------------
// fs[C][kVT] features, hs[HID][kVT] hidden activations (post LeakyReLU), zs[K][kVT] logits -> (normalised) -> probabilities.
// Returns the L2 norm of the raw logits (normalize) so the backward can differentiate through it.
__device__ __forceinline__ float head_var_forward_column(const float* fs, float* hs, float* zs, int t, int C, int HID, int K,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float invT,
                                                         int normalize, float* zn_out /* [K][kVT] or null */) {
    if (HID > 0) {
        for (int j = 0; j < HID; ++j) {
            const float* wr = w1 + (size_t)j * C;
            float a = b1[j];
            for (int c = 0; c < C; ++c) a = fmaf(wr[c], fs[c * kVT + t], a);
            hs[j * kVT + t] = a > 0.f ? a : kLeaky * a;
        }
        for (int k = 0; k < K; ++k) {
            const float* wr = w2 + (size_t)k * HID;
            float a = b2[k];
            for (int j = 0; j < HID; ++j) a = fmaf(wr[j], hs[j * kVT + t], a);
            zs[k * kVT + t] = a;
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const float* wr = w1 + (size_t)k * C;
            float a = b1[k];
            for (int c = 0; c < C; ++c) a = fmaf(wr[c], fs[c * kVT + t], a);
            zs[k * kVT + t] = a;
        }
    }
    float nrm = 1.f;
    if (normalize) {
        float ss = 0.f;
        for (int k = 0; k < K; ++k) ss = fmaf(zs[k * kVT + t], zs[k * kVT + t], ss);
        nrm = sqrtf(ss);
        const float inv = 1.f / fmaxf(nrm, kNormEps);
        for (int k = 0; k < K; ++k) {
            const float v = zs[k * kVT + t] * inv;
            zs[k * kVT + t] = v;
            if (zn_out) zn_out[k * kVT + t] = v;
        }
    }
    float mx = -3.4e38f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, zs[k * kVT + t] * invT);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
        const float e = expf(zs[k * kVT + t] * invT - mx);
        zs[k * kVT + t] = e;
        sum += e;
    }
    const float inv = 1.f / sum;
    for (int k = 0; k < K; ++k) zs[k * kVT + t] *= inv;
    return nrm;
}

// dL/dlogits (pre-normalisation) from dL/dprob, in place of the probabilities in zs; zn = normalised logits (normalize only).
__device__ __forceinline__ void head_var_backward_column(float* zs, const float* zn, const float* g /* [K] strided by gs */, size_t gs,
                                                         int t, int K, float invT, int normalize, float nrm) {
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot = fmaf(zs[k * kVT + t], g[(size_t)k * gs], dot);
    for (int k = 0; k < K; ++k) zs[k * kVT + t] = zs[k * kVT + t] * (g[(size_t)k * gs] - dot) * invT;     // d/d(zn or z)
    if (normalize) {
        if (nrm > kNormEps) {
            float proj = 0.f;
            for (int k = 0; k < K; ++k) proj = fmaf(zs[k * kVT + t], zn[k * kVT + t], proj);
            const float inv = 1.f / nrm;
            for (int k = 0; k < K; ++k) zs[k * kVT + t] = (zs[k * kVT + t] - zn[k * kVT + t] * proj) * inv;
        } else {
            for (int k = 0; k < K; ++k) zs[k * kVT + t] *= (1.f / kNormEps);
        }
    }
}

template <typename T>
__device__ __forceinline__ void load_feature_column(const T* __restrict__ fp, float* fs, int t, int C, bool live) {
    for (int c = 0; c < C; ++c) fs[c * kVT + t] = live ? to_f32(fp[c]) : 0.f;
}

// ============================================================================================================ local head
template <typename T>
__global__ __launch_bounds__(kVT) void head_local_var_fwd_kernel(const T* __restrict__ feat, int H, int W, int C,
                                                                 const int32_t* __restrict__ src, const int32_t* __restrict__ flips, int M,
                                                                 const float* __restrict__ w1, const float* __restrict__ b1, int HID,
                                                                 const float* __restrict__ w2, const float* __restrict__ b2, int S, int K,
                                                                 float invT, int normalize, float* __restrict__ prob) {
    extern __shared__ float sm[];
    float* fs = sm;                          // [C][kVT]
    float* hs = fs + (size_t)C * kVT;        // [HID][kVT]
    float* zs = hs + (size_t)HID * kVT;      // [K][kVT]
    const int t = threadIdx.x, HW = H * W, m = blockIdx.y;
    const int pix = blockIdx.x * kVT + t;
    const bool live = pix < HW;
    const int h = live ? pix / W : 0, wq = live ? pix % W : 0;
    const int f = flips ? flips[m] : 0;
    load_feature_column(feat + ((size_t)src[m] * HW + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C, fs, t, C, live);
    const int R1 = HID > 0 ? HID : K;
    for (int s = 0; s < S; ++s) {
        head_var_forward_column(fs, hs, zs, t, C, HID, K, w1 + (size_t)s * R1 * C, b1 + (size_t)s * R1,
                                HID > 0 ? w2 + (size_t)s * K * HID : nullptr, HID > 0 ? b2 + (size_t)s * K : nullptr, invT, normalize, nullptr);
        if (live) {
            float* out = prob + (((size_t)s * M + m) * K) * HW + pix;
            for (int k = 0; k < K; ++k) out[(size_t)k * HW] = zs[k * kVT + t];
        }
    }
}

// per-block partial of the parameter gradients of ONE sub-head: [gw1 (R1*C) | gb1 (R1) | gw2 (K*HID) | gb2 (K)]
__host__ __device__ inline size_t head_var_part_floats(int C, int HID, int K) {
    const int R1 = HID > 0 ? HID : K;
    return (size_t)R1 * C + R1 + (HID > 0 ? (size_t)K * HID + K : 0);
}

template <typename T>
__global__ __launch_bounds__(kVT) void head_local_var_bwd_kernel(const T* __restrict__ feat, int H, int W, int C,
                                                                 const int32_t* __restrict__ src, const int32_t* __restrict__ flips, int M,
                                                                 const float* __restrict__ w1, const float* __restrict__ b1, int HID,
                                                                 const float* __restrict__ w2, const float* __restrict__ b2, int S, int K,
                                                                 float invT, int normalize, const float* __restrict__ gprob,
                                                                 T* __restrict__ gfeat, float* __restrict__ partials, int nblk) {
    extern __shared__ float sm[];
    float* fs = sm;                          // [C][kVT]   features
    float* gs = fs + (size_t)C * kVT;        // [C][kVT]   gradient w.r.t. the features, summed over the sub-heads
    float* hs = gs + (size_t)C * kVT;        // [HID][kVT] hidden activations
    float* ds = hs + (size_t)HID * kVT;      // [HID][kVT] gradient w.r.t. the hidden pre-activations
    float* zs = ds + (size_t)HID * kVT;      // [K][kVT]   probabilities, then dL/dlogits
    float* zn = zs + (size_t)K * kVT;        // [K][kVT]   normalised logits (normalize only)
    const int t = threadIdx.x, HW = H * W;
    const int R1 = HID > 0 ? HID : K;
    const size_t PF = head_var_part_floats(C, HID, K);
    float* mine = partials + (size_t)blockIdx.x * S * PF;     // zero-filled by the host wrapper; every entry has ONE owner thread
    const int chunksPerM = (HW + kVT - 1) / kVT;
    const int64_t nchunks = (int64_t)M * chunksPerM;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += nblk) {
        const int m = (int)(ch / chunksPerM), pix = (int)(ch % chunksPerM) * kVT + t;
        const bool live = pix < HW;
        const int h = live ? pix / W : 0, wq = live ? pix % W : 0;
        const int f = flips ? flips[m] : 0;
        const size_t foff = ((size_t)src[m] * HW + (size_t)flip_h(h, H, f) * W + flip_w(wq, W, f)) * C;
        load_feature_column(feat + foff, fs, t, C, live);
        for (int c = 0; c < C; ++c) gs[c * kVT + t] = 0.f;
        for (int s = 0; s < S; ++s) {
            const float* w1s = w1 + (size_t)s * R1 * C;
            const float* w2s = HID > 0 ? w2 + (size_t)s * K * HID : nullptr;
            const float nrm = head_var_forward_column(fs, hs, zs, t, C, HID, K, w1s, b1 + (size_t)s * R1, w2s,
                                                      HID > 0 ? b2 + (size_t)s * K : nullptr, invT, normalize, zn);
            if (live) {
                head_var_backward_column(zs, zn, gprob + (((size_t)s * M + m) * K) * HW + pix, (size_t)HW, t, K, invT, normalize, nrm);
            } else {
                for (int k = 0; k < K; ++k) zs[k * kVT + t] = 0.f;      // a dead pixel contributes nothing below
            }
            if (HID > 0) {
                for (int j = 0; j < HID; ++j) {
                    float a = 0.f;
                    for (int k = 0; k < K; ++k) a = fmaf(w2s[(size_t)k * HID + j], zs[k * kVT + t], a);
                    ds[j * kVT + t] = a * (hs[j * kVT + t] > 0.f ? 1.f : kLeaky);
                }
            }
            const float* dl = HID > 0 ? ds : zs;       // gradient w.r.t. the rows of w1
            for (int c = 0; c < C; ++c) {
                float a = gs[c * kVT + t];
                for (int r = 0; r < R1; ++r) a = fmaf(w1s[(size_t)r * C + c], dl[r * kVT + t], a);
                gs[c * kVT + t] = a;
            }
            __builtin_amdgcn_wave_barrier();             // one wave per block: LDS columns written above are read across lanes below
            float* part = mine + (size_t)s * PF;
            for (int e = t; e < R1 * C; e += kVT) {      // gw1[r][c] += sum_px dl[r][px] * f[c][px]
                const int r = e / C, c = e - r * C;
                float a = 0.f;
                for (int p = 0; p < kVT; ++p) a = fmaf(dl[r * kVT + p], fs[c * kVT + p], a);
                part[e] += a;
            }
            for (int r = t; r < R1; r += kVT) {
                float a = 0.f;
                for (int p = 0; p < kVT; ++p) a += dl[r * kVT + p];
                part[(size_t)R1 * C + r] += a;
            }
            if (HID > 0) {
                float* p2 = part + (size_t)R1 * C + R1;
                for (int e = t; e < K * HID; e += kVT) {
                    const int k = e / HID, j = e - k * HID;
                    float a = 0.f;
                    for (int p = 0; p < kVT; ++p) a = fmaf(zs[k * kVT + p], hs[j * kVT + p], a);
                    p2[e] += a;
                }
                for (int k = t; k < K; k += kVT) {
                    float a = 0.f;
                    for (int p = 0; p < kVT; ++p) a += zs[k * kVT + p];
                    p2[(size_t)K * HID + k] += a;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (gfeat && live) {
            T* gp = gfeat + foff;                        // src rows are pairwise distinct and each pixel has one owner: plain stores
            for (int c = 0; c < C; ++c) gp[c] = from_f32<T>(gs[c * kVT + t]);
        }
    }
}

// out[e] = sum over blocks of partials[blk][e], fixed order (deterministic); grid = ceil(n / 256)
__global__ __launch_bounds__(256) void head_var_reduce_kernel(const float* __restrict__ partials, int nblk, size_t stride, int n,
                                                              float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float a = 0.f;
    for (int q = 0; q < nblk; ++q) a += partials[(size_t)q * stride + e];
    out[e] = a;
}

// scatter the reduced [S][PF] vector into the four stacked gradient tensors
__global__ __launch_bounds__(256) void head_var_scatter_kernel(const float* __restrict__ red, int S, int C, int HID, int K,
                                                               float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2,
                                                               float* __restrict__ gb2) {
    const int R1 = HID > 0 ? HID : K;
    const size_t PF = head_var_part_floats(C, HID, K);
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)S * PF) return;
    const int s = (int)(e / PF);
    size_t o = e - (size_t)s * PF;
    if (o < (size_t)R1 * C) { gw1[(size_t)s * R1 * C + o] = red[e]; return; }
    o -= (size_t)R1 * C;
    if (o < (size_t)R1) { gb1[(size_t)s * R1 + o] = red[e]; return; }
    o -= R1;
    if (o < (size_t)K * HID) { gw2[(size_t)s * K * HID + o] = red[e]; return; }
    o -= (size_t)K * HID;
    gb2[(size_t)s * K + o] = red[e];
}

// ============================================================================================================ global head
template <typename T>
__global__ __launch_bounds__(256) void head_var_pool_kernel(const T* __restrict__ feat, int HW, int C, const int32_t* __restrict__ src,
                                                            float* __restrict__ pooled) {
    // grid (M, ceil(C/64)): 4 pixel-strided partial sums per channel, fixed-order combine
    __shared__ float part[4][64];
    const int m = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C) {
        const T* f = feat + (size_t)src[m] * HW * C + c;
        for (int p = q; p < HW; p += 4) a += to_f32(f[(size_t)p * C]);
    }
    part[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && c < C) pooled[(size_t)m * C + c] = (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]) / (float)HW;
}

// grid (ceil(M/64), S): thread = one pooled sample; the same column routines as the local head
__global__ __launch_bounds__(kVT) void head_global_var_fwd_kernel(const float* __restrict__ pooled, int M, int C,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1, int HID,
                                                                  const float* __restrict__ w2, const float* __restrict__ b2, int K,
                                                                  float invT, int normalize, float* __restrict__ prob) {
    extern __shared__ float sm[];
    float* fs = sm;
    float* hs = fs + (size_t)C * kVT;
    float* zs = hs + (size_t)HID * kVT;
    const int t = threadIdx.x, m = blockIdx.x * kVT + t, s = blockIdx.y;
    const bool live = m < M;
    load_feature_column(pooled + (size_t)(live ? m : 0) * C, fs, t, C, live);
    const int R1 = HID > 0 ? HID : K;
    head_var_forward_column(fs, hs, zs, t, C, HID, K, w1 + (size_t)s * R1 * C, b1 + (size_t)s * R1,
                            HID > 0 ? w2 + (size_t)s * K * HID : nullptr, HID > 0 ? b2 + (size_t)s * K : nullptr, invT, normalize, nullptr);
    if (live)
        for (int k = 0; k < K; ++k) prob[((size_t)s * M + m) * K + k] = zs[k * kVT + t];
}

// grid (S): one wave per sub-head walks the samples 64 at a time; parameter gradients written directly (one owner per entry),
// dpool[s][m][c] for the feature pass.
__global__ __launch_bounds__(kVT) void head_global_var_bwd_kernel(const float* __restrict__ pooled, int M, int C,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1, int HID,
                                                                  const float* __restrict__ w2, const float* __restrict__ b2, int S, int K,
                                                                  float invT, int normalize, const float* __restrict__ gprob,
                                                                  float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2,
                                                                  float* __restrict__ gb2, float* __restrict__ dpool) {
    extern __shared__ float sm[];
    float* fs = sm;
    float* hs = fs + (size_t)C * kVT;
    float* ds = hs + (size_t)HID * kVT;
    float* zs = ds + (size_t)HID * kVT;
    float* zn = zs + (size_t)K * kVT;
    const int t = threadIdx.x, s = blockIdx.x;
    const int R1 = HID > 0 ? HID : K;
    const float* w1s = w1 + (size_t)s * R1 * C;
    const float* w2s = HID > 0 ? w2 + (size_t)s * K * HID : nullptr;
    float* gw1s = gw1 + (size_t)s * R1 * C;
    float* gb1s = gb1 + (size_t)s * R1;
    float* gw2s = HID > 0 ? gw2 + (size_t)s * K * HID : nullptr;
    float* gb2s = HID > 0 ? gb2 + (size_t)s * K : nullptr;
    for (int e = t; e < R1 * C; e += kVT) gw1s[e] = 0.f;
    for (int e = t; e < R1; e += kVT) gb1s[e] = 0.f;
    if (HID > 0) {
        for (int e = t; e < K * HID; e += kVT) gw2s[e] = 0.f;
        for (int e = t; e < K; e += kVT) gb2s[e] = 0.f;
    }
    for (int m0 = 0; m0 < M; m0 += kVT) {
        const int m = m0 + t;
        const bool live = m < M;
        load_feature_column(pooled + (size_t)(live ? m : 0) * C, fs, t, C, live);
        const float nrm = head_var_forward_column(fs, hs, zs, t, C, HID, K, w1s, b1 + (size_t)s * R1, w2s, HID > 0 ? b2 + (size_t)s * K : nullptr,
                                                  invT, normalize, zn);
        if (live) head_var_backward_column(zs, zn, gprob + ((size_t)s * M + m) * K, 1, t, K, invT, normalize, nrm);
        else
            for (int k = 0; k < K; ++k) zs[k * kVT + t] = 0.f;
        if (HID > 0)
            for (int j = 0; j < HID; ++j) {
                float a = 0.f;
                for (int k = 0; k < K; ++k) a = fmaf(w2s[(size_t)k * HID + j], zs[k * kVT + t], a);
                ds[j * kVT + t] = a * (hs[j * kVT + t] > 0.f ? 1.f : kLeaky);
            }
        const float* dl = HID > 0 ? ds : zs;
        if (live)
            for (int c = 0; c < C; ++c) {
                float a = 0.f;
                for (int r = 0; r < R1; ++r) a = fmaf(w1s[(size_t)r * C + c], dl[r * kVT + t], a);
                dpool[((size_t)s * M + m) * C + c] = a;
            }
        __builtin_amdgcn_wave_barrier();
        for (int e = t; e < R1 * C; e += kVT) {
            const int r = e / C, c = e - r * C;
            float a = 0.f;
            for (int p = 0; p < kVT; ++p) a = fmaf(dl[r * kVT + p], fs[c * kVT + p], a);
            gw1s[e] += a;
        }
        for (int r = t; r < R1; r += kVT) {
            float a = 0.f;
            for (int p = 0; p < kVT; ++p) a += dl[r * kVT + p];
            gb1s[r] += a;
        }
        if (HID > 0) {
            for (int e = t; e < K * HID; e += kVT) {
                const int k = e / HID, j = e - k * HID;
                float a = 0.f;
                for (int p = 0; p < kVT; ++p) a = fmaf(zs[k * kVT + p], hs[j * kVT + p], a);
                gw2s[e] += a;
            }
            for (int k = t; k < K; k += kVT) {
                float a = 0.f;
                for (int p = 0; p < kVT; ++p) a += zs[k * kVT + p];
                gb2s[k] += a;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// gfeat[src[m]][px][c] = (sum_s dpool[s][m][c]) / HW   (avg-pool backward; rows of src are distinct, gfeat pre-zeroed elsewhere)
template <typename T>
__global__ __launch_bounds__(256) void head_global_var_feat_kernel(const float* __restrict__ dpool, int S, int M, int HW, int C,
                                                                   const int32_t* __restrict__ src, T* __restrict__ gfeat) {
    extern __shared__ float gp[];   // [C]
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += dpool[((size_t)s * M + m) * C + c];
        gp[c] = a / (float)HW;
    }
    __syncthreads();
    T* g = gfeat + (size_t)src[m] * HW * C;
    const size_t n = (size_t)HW * C;
    for (size_t e = (size_t)blockIdx.y * 256 + threadIdx.x; e < n; e += (size_t)gridDim.y * 256) g[e] = from_f32<T>(gp[e % C]);
}

static inline size_t local_fwd_lds(int64_t C, int64_t HID, int64_t K) { return (size_t)(C + HID + K) * kVT * 4; }
static inline size_t local_bwd_lds(int64_t C, int64_t HID, int64_t K) { return (size_t)(2 * C + 2 * HID + 2 * K) * kVT * 4; }
static inline int local_bwd_blocks(int64_t M, int64_t HW) { return (int)std::min<int64_t>(M * cdiv(HW, kVT), 1024); }

}  // namespace miseg

using namespace miseg;

#define VAR_COMMON_CHECKS(name)                                                                                                   \
    MISEG_REQUIRE(S > 0 && K > 1 && K <= 64 && C > 0 && M > 0 && HID >= 0 && HID <= 256, name ": bad shape (K <= 64, HID <= 256)"); \
    MISEG_REQUIRE(w1 && b1 && (HID == 0 || (w2 && b2)), name ": null weight pointer");                                             \
    MISEG_REQUIRE(T > 0.f, name ": temperature must be positive");                                                                 \
    MISEG_REQUIRE(dt == MISEG_F32 || dt == MISEG_BF16, name ": bad dtype")

extern "C" int miseg_head_local_var_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                        const int32_t* src, const int32_t* flips, int64_t M, const float* w1, const float* b1,
                                        int64_t HID, const float* w2, const float* b2, int64_t S, int64_t K, float T, int normalize,
                                        float* prob) {
    MISEG_TAPE(miseg_head_local_var_fwd, stream, dt, feat, B, H, W, C, src, flips, M, w1, b1, HID, w2, b2, S, K, T, normalize, prob);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_local_var_fwd, stream, MISEG_BF16, feat, B, H, W, C, src, flips, M, w1, b1, HID, w2, b2, S, K, T, normalize, prob);
    VAR_COMMON_CHECKS("head_local_var_fwd");
    MISEG_REQUIRE(feat && src && prob && B > 0 && H > 0 && W > 0, "head_local_var_fwd: null pointer / bad shape");
    const size_t lds = local_fwd_lds(C, HID, K);
    MISEG_REQUIRE(lds <= 150 * 1024, "head_local_var_fwd: C + HID + K too large for LDS (%zu bytes)", lds);
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)cdiv(H * W, kVT), (unsigned)M);
#define GO(TT)                                                                                                                      \
    {                                                                                                                               \
        hipFuncSetAttribute((const void*)head_local_var_fwd_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL(head_local_var_fwd_kernel<TT>, grid, dim3(kVT), lds, st, (const TT*)feat, (int)H, (int)W, (int)C, src, flips, \
                           (int)M, w1, b1, (int)HID, w2, b2, (int)S, (int)K, 1.f / T, normalize, prob);                             \
    }
    if (dt == MISEG_F32) GO(float) else GO(bf16)
#undef GO
    MISEG_LAUNCH_CHECK("head_local_var_fwd_kernel");
    return MISEG_OK;
}

extern "C" int64_t miseg_head_local_var_bwd_ws_bytes(int64_t M, int64_t H, int64_t W, int64_t C, int64_t HID, int64_t S, int64_t K) {
    if (M <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || K <= 0 || HID < 0) return -1;
    const size_t pf = head_var_part_floats((int)C, (int)HID, (int)K);
    return (int64_t)(((size_t)local_bwd_blocks(M, H * W) + 1) * S * pf * 4);
}

extern "C" int miseg_head_local_var_bwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                        const int32_t* src, const int32_t* flips, int64_t M, const float* w1, const float* b1,
                                        int64_t HID, const float* w2, const float* b2, int64_t S, int64_t K, float T, int normalize,
                                        const float* gprob, void* gfeat, float* gw1, float* gb1, float* gw2, float* gb2, void* ws,
                                        int64_t ws_bytes) {
    MISEG_TAPE(miseg_head_local_var_bwd, stream, dt, feat, B, H, W, C, src, flips, M, w1, b1, HID, w2, b2, S, K, T, normalize, gprob, gfeat, gw1, gb1, gw2, gb2, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_local_var_bwd, stream, MISEG_BF16, feat, B, H, W, C, src, flips, M, w1, b1, HID, w2, b2, S, K, T, normalize, gprob, gfeat, gw1, gb1, gw2, gb2, ws, ws_bytes);
    VAR_COMMON_CHECKS("head_local_var_bwd");
    MISEG_REQUIRE(feat && src && gprob && gw1 && gb1 && (HID == 0 || (gw2 && gb2)) && ws, "head_local_var_bwd: null pointer");
    MISEG_REQUIRE(ws_bytes >= miseg_head_local_var_bwd_ws_bytes(M, H, W, C, HID, S, K), "head_local_var_bwd: workspace too small");
    const size_t lds = local_bwd_lds(C, HID, K);
    MISEG_REQUIRE(lds <= 150 * 1024, "head_local_var_bwd: C + HID + K too large for LDS (%zu bytes)", lds);
    hipStream_t st = as_stream(stream);
    const int nblk = local_bwd_blocks(M, H * W);
    const size_t pf = head_var_part_floats((int)C, (int)HID, (int)K), stride = (size_t)S * pf;
    float* partials = (float*)ws;
    float* reduced = partials + (size_t)nblk * stride;
    if (hipMemsetAsync(partials, 0, (size_t)nblk * stride * 4, st) != hipSuccess) return fail(MISEG_E_LAUNCH, "head_local_var_bwd: memset failed");
#define GO(TT)                                                                                                                      \
    {                                                                                                                               \
        hipFuncSetAttribute((const void*)head_local_var_bwd_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL(head_local_var_bwd_kernel<TT>, dim3((unsigned)nblk), dim3(kVT), lds, st, (const TT*)feat, (int)H, (int)W, (int)C, \
                           src, flips, (int)M, w1, b1, (int)HID, w2, b2, (int)S, (int)K, 1.f / T, normalize, gprob, (TT*)gfeat, partials, nblk); \
    }
    if (dt == MISEG_F32) GO(float) else GO(bf16)
#undef GO
    MISEG_LAUNCH_CHECK("head_local_var_bwd_kernel");
    hipLaunchKernelGGL(head_var_reduce_kernel, dim3((unsigned)cdiv((int64_t)stride, 256)), dim3(256), 0, st, partials, nblk, stride, (int)stride, reduced);
    hipLaunchKernelGGL(head_var_scatter_kernel, dim3((unsigned)cdiv((int64_t)stride, 256)), dim3(256), 0, st, reduced, (int)S, (int)C, (int)HID, (int)K,
                       gw1, gb1, gw2, gb2);
    MISEG_LAUNCH_CHECK("head_var_reduce_kernel");
    return MISEG_OK;
}

extern "C" int miseg_head_global_var_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                         const int32_t* src, int64_t M, const float* w1, const float* b1, int64_t HID, const float* w2,
                                         const float* b2, int64_t S, int64_t K, float T, int normalize, float* pooled, float* prob) {
    MISEG_TAPE(miseg_head_global_var_fwd, stream, dt, feat, B, H, W, C, src, M, w1, b1, HID, w2, b2, S, K, T, normalize, pooled, prob);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_global_var_fwd, stream, MISEG_BF16, feat, B, H, W, C, src, M, w1, b1, HID, w2, b2, S, K, T, normalize, pooled, prob);
    VAR_COMMON_CHECKS("head_global_var_fwd");
    MISEG_REQUIRE(feat && src && pooled && prob && B > 0 && H > 0 && W > 0, "head_global_var_fwd: null pointer / bad shape");
    const size_t lds = local_fwd_lds(C, HID, K);
    MISEG_REQUIRE(lds <= 150 * 1024, "head_global_var_fwd: C + HID + K too large for LDS (%zu bytes)", lds);
    hipStream_t st = as_stream(stream);
    if (dt == MISEG_F32)
        hipLaunchKernelGGL(head_var_pool_kernel<float>, dim3((unsigned)M, (unsigned)cdiv(C, 64)), dim3(256), 0, st, (const float*)feat, (int)(H * W), (int)C, src, pooled);
    else
        hipLaunchKernelGGL(head_var_pool_kernel<bf16>, dim3((unsigned)M, (unsigned)cdiv(C, 64)), dim3(256), 0, st, (const bf16*)feat, (int)(H * W), (int)C, src, pooled);
    MISEG_LAUNCH_CHECK("head_var_pool_kernel");
    hipFuncSetAttribute((const void*)head_global_var_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(head_global_var_fwd_kernel, dim3((unsigned)cdiv(M, kVT), (unsigned)S), dim3(kVT), lds, st, pooled, (int)M, (int)C, w1, b1,
                       (int)HID, w2, b2, (int)K, 1.f / T, normalize, prob);
    MISEG_LAUNCH_CHECK("head_global_var_fwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_head_global_var_bwd(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src, int64_t M,
                                         const float* w1, const float* b1, int64_t HID, const float* w2, const float* b2, int64_t S,
                                         int64_t K, float T, int normalize, const float* pooled, const float* gprob, void* gfeat,
                                         float* gw1, float* gb1, float* gw2, float* gb2, float* dpool_ws) {
    MISEG_TAPE(miseg_head_global_var_bwd, stream, dt, B, H, W, C, src, M, w1, b1, HID, w2, b2, S, K, T, normalize, pooled, gprob, gfeat, gw1, gb1, gw2, gb2, dpool_ws);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_global_var_bwd, stream, MISEG_BF16, B, H, W, C, src, M, w1, b1, HID, w2, b2, S, K, T, normalize, pooled, gprob, gfeat, gw1, gb1, gw2, gb2, dpool_ws);
    VAR_COMMON_CHECKS("head_global_var_bwd");
    MISEG_REQUIRE(src && pooled && gprob && gw1 && gb1 && (HID == 0 || (gw2 && gb2)) && dpool_ws, "head_global_var_bwd: null pointer");
    const size_t lds = local_bwd_lds(C, HID, K) - (size_t)C * kVT * 4;    // no feature-gradient columns here
    MISEG_REQUIRE(lds <= 150 * 1024, "head_global_var_bwd: C + HID + K too large for LDS (%zu bytes)", lds);
    hipStream_t st = as_stream(stream);
    hipFuncSetAttribute((const void*)head_global_var_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(head_global_var_bwd_kernel, dim3((unsigned)S), dim3(kVT), lds, st, pooled, (int)M, (int)C, w1, b1, (int)HID, w2, b2, (int)S,
                       (int)K, 1.f / T, normalize, gprob, gw1, gb1, gw2, gb2, dpool_ws);
    MISEG_LAUNCH_CHECK("head_global_var_bwd_kernel");
    if (gfeat) {
        const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(32, cdiv(H * W * C, 256 * 8)));
        if (dt == MISEG_F32)
            hipLaunchKernelGGL(head_global_var_feat_kernel<float>, dim3((unsigned)M, chunks), dim3(256), (size_t)C * 4, st, dpool_ws, (int)S, (int)M,
                               (int)(H * W), (int)C, src, (float*)gfeat);
        else
            hipLaunchKernelGGL(head_global_var_feat_kernel<bf16>, dim3((unsigned)M, chunks), dim3(256), (size_t)C * 4, st, dpool_ws, (int)S, (int)M,
                               (int)(H * W), (int)C, src, (bf16*)gfeat);
        MISEG_LAUNCH_CHECK("head_global_var_feat_kernel");
    }
    return MISEG_OK;
}
