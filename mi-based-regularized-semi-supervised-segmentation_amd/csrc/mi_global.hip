// Global IIC mutual information (IIDLoss) and the encoder ClusterHead, all sub-heads per launch.
// ref: contrastyou/losses/iic_loss.py:43-94, contrastyou/trainer/_utils.py:96-134.
// These are latency-only ops (N<=64, K<=64): one small block per sub-head, fp32, fixed order.
#include "common.h"

namespace miseg {

constexpr int kMaxK = 64;

// joint P (symmetrised, normalised) into LDS; returns via Ps[K*K], rowv[K], colv[K]
__device__ void global_joint(const float* x, const float* y, int N, int K, float* Ps, float* rowv, float* colv, float* red) {
    const int KK = K * K;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        int i = e / K, j = e % K;
        float s = 0.f, t = 0.f;
        for (int n = 0; n < N; ++n) {
            s += x[n * K + i] * y[n * K + j];
            t += x[n * K + j] * y[n * K + i];
        }
        Ps[e] = (s + t) / 2.0f;  // (P + P^T)/2, iic_loss.py:90-91
    }
    __syncthreads();
    float z = 0.f;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) z += Ps[e];
    z = block_sum(z, red);
    for (int e = threadIdx.x; e < KK; e += blockDim.x) Ps[e] /= z;  // iic_loss.py:92
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        float rs = 0.f, cs = 0.f;
        for (int t = 0; t < K; ++t) { rs += Ps[c * K + t]; cs += Ps[t * K + c]; }
        rowv[c] = rs;  // p_i = P.sum(dim=1)  (iic_loss.py:56-58)
        colv[c] = cs;  // p_j = P.sum(dim=0)  (iic_loss.py:59)
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void iic_global_fwd_kernel(const float* __restrict__ xs, const float* __restrict__ ys, int N,
                                                             int K, float lamb, float* __restrict__ loss,
                                                             float* __restrict__ loss_nl, float* __restrict__ joint, long long hs) {
    __shared__ float Ps[kMaxK * kMaxK];
    __shared__ float rowv[kMaxK], colv[kMaxK], red[17];
    const int s = blockIdx.x, KK = K * K;
    global_joint(xs + (size_t)s * hs, ys + (size_t)s * hs, N, K, Ps, rowv, colv, red);      // hs: elements between sub-heads
    const float eps = 1e-10f;
    float a = 0.f, b = 0.f;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        int i = e / K, j = e % K;
        float p = Ps[e], lp = logf(p + eps), lj = logf(colv[j] + eps), li = logf(rowv[i] + eps);
        a += -p * (lp - lamb * lj - lamb * li);
        b += -p * (lp - lj - li);
        joint[(size_t)s * KK + e] = p;
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) { loss[s] = a; loss_nl[s] = b; }
}

// d loss / d x[n,i] = sum_j Gu[i][j] y[n,j], d loss / d y[n,j] = sum_i Gu[i][j] x[n,i], where
// Gu = d loss / d (unsymmetrised, unnormalised joint) = (Gsym - <Gsym, P>) / Z, Gsym = (Gp + Gp^T)/2.
__global__ __launch_bounds__(256) void iic_global_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ ys, int N,
                                                             int K, float lamb, const float* __restrict__ upstream,
                                                             float* __restrict__ gxs, float* __restrict__ gys, long long hs) {
    __shared__ float Ps[kMaxK * kMaxK];
    __shared__ float Gp[kMaxK * kMaxK];
    __shared__ float rowv[kMaxK], colv[kMaxK], red[17];
    const int s = blockIdx.x, KK = K * K;
    const float* x = xs + (size_t)s * hs;
    const float* y = ys + (size_t)s * hs;
    global_joint(x, y, N, K, Ps, rowv, colv, red);
    // recover Z (sum of the symmetrised raw joint) = sum_n (sum_i x)(sum_j y)
    float z = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float sx = 0.f, sy = 0.f;
        for (int k = 0; k < K; ++k) { sx += x[n * K + k]; sy += y[n * K + k]; }
        z += sx * sy;
    }
    z = block_sum(z, red);
    const float eps = 1e-10f;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        int i = e / K, j = e % K;
        float p = Ps[e], cj = colv[j], ri = rowv[i];
        Gp[e] = -(logf(p + eps) + p / (p + eps) - lamb * (logf(cj + eps) + cj / (cj + eps)) -
                  lamb * (logf(ri + eps) + ri / (ri + eps)));
    }
    __syncthreads();
    float dot = 0.f;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        int i = e / K, j = e % K;
        dot += (Gp[e] + Gp[j * K + i]) / 2.0f * Ps[e];
    }
    dot = block_sum(dot, red);
    const float up = upstream ? upstream[s] : 1.f;
    __syncthreads();
    // Gu overwrites Ps (symmetric)
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        int i = e / K, j = e % K;
        Ps[e] = up * (((Gp[e] + Gp[j * K + i]) / 2.0f - dot) / z);
    }
    __syncthreads();
    // raw joint J = sum_n x_n y_n^T enters as (J + J^T)/2: dL/dJ = (Gu + Gu^T)/2 = Gu (symmetric)
    for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
        int n = e / K, k = e % K;
        float ax = 0.f, ay = 0.f;
        for (int t = 0; t < K; ++t) {
            ax += Ps[k * K + t] * y[n * K + t];
            ay += Ps[t * K + k] * x[n * K + t];
        }
        gxs[(size_t)s * hs + e] = ax;
        gys[(size_t)s * hs + e] = ay;
    }
}

// ---------------------------------------------------------------- encoder ClusterHead (linear)
// pooled[m][c] = mean_{h,w} feat[src[m]][h][w][c];  prob[s][m][:] = softmax((W_s pooled + b_s)/T)
// grid (M, ceil(C/32)): a block owns 32 channels of one sample; its 8 thread rows split the pixels (8 interleaved streams of
// 64-byte coalesced reads instead of one thread walking all HW pixels), combined through LDS in fixed order
template <typename T>
__global__ __launch_bounds__(256) void head_pool_kernel(const T* __restrict__ feat, int HW, int C, const int32_t* __restrict__ src,
                                                        float* __restrict__ pooled) {
    __shared__ float part[8][32];
    const int m = blockIdx.x, c = blockIdx.y * 32 + (threadIdx.x & 31), row = threadIdx.x >> 5;
    const T* f = feat + (size_t)src[m] * HW * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int p = row;
        for (; p + 24 < HW; p += 32) {
            s0 += to_f32(f[(size_t)p * C + c]);
            s1 += to_f32(f[(size_t)(p + 8) * C + c]);
            s2 += to_f32(f[(size_t)(p + 16) * C + c]);
            s3 += to_f32(f[(size_t)(p + 24) * C + c]);
        }
        for (; p < HW; p += 8) s0 += to_f32(f[(size_t)p * C + c]);
    }
    part[row][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (row == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += part[r][threadIdx.x];
        pooled[(size_t)m * C + c] = s / (float)HW;
    }
}

__global__ __launch_bounds__(64) void head_global_fwd_kernel(const float* __restrict__ pooled, int M, int C, const float* __restrict__ w,
                                                             const float* __restrict__ b, int K, float T, float* __restrict__ prob) {
    const int m = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    float z = -3.4e38f;
    if (lane < K) {
        const float* wr = w + ((size_t)s * K + lane) * C;
        float a = b[s * K + lane];
        for (int c = 0; c < C; ++c) a += wr[c] * pooled[(size_t)m * C + c];
        z = a / T;
    }
    float mx = z;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = lane < K ? expf(z - mx) : 0.f;
    float sum = wave_sum(e);
    if (lane < K) prob[((size_t)s * M + m) * K + lane] = e / sum;
}

// grid (S, ceil(K*C/256)): dz = p*(g - <g,p>)/T of sub-head s is recomputed per block into LDS (M*K values); every thread
// then owns one gw element (k, c) = sum_m dz[m][k] * pooled[m][c].  Block y == 0 also stores dz for the feature pass and gb.
__global__ __launch_bounds__(256) void head_global_bwd_kernel(const float* __restrict__ pooled, int M, int C, const float* __restrict__ w,
                                                              int S, int K, float T, const float* __restrict__ prob,
                                                              const float* __restrict__ gprob, float* __restrict__ dz_all,
                                                              float* __restrict__ gw, float* __restrict__ gb) {
    extern __shared__ float dzs[];   // [M][K]
    const int s = blockIdx.x;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float* p = prob + ((size_t)s * M + m) * K;
        const float* g = gprob + ((size_t)s * M + m) * K;
        float dot = 0.f;
        for (int k = 0; k < K; ++k) dot += g[k] * p[k];
        for (int k = 0; k < K; ++k) dzs[m * K + k] = p[k] * (g[k] - dot) / T;
    }
    __syncthreads();
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e < K * C) {
        const int k = e / C, c = e % C;
        float a = 0.f;
        for (int m = 0; m < M; ++m) a += dzs[m * K + k] * pooled[(size_t)m * C + c];
        gw[(size_t)s * K * C + e] = a;
    }
    if (blockIdx.y == 0) {
        float* dz = dz_all + (size_t)s * M * K;
        for (int i = threadIdx.x; i < M * K; i += blockDim.x) dz[i] = dzs[i];
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float a = 0.f;
            for (int m = 0; m < M; ++m) a += dzs[m * K + k];
            gb[s * K + k] = a;
        }
    }
}

// gfeat[src[m]][h][w][c] = (sum_{s,k} W[s][k][c] dz[s][m][k]) / HW      (avg-pool backward)
// src[] is pairwise distinct and the caller hands in a zeroed gfeat (like the local head): every row has one writer, so
// the rows are written with plain 16-byte stores.  grid (M, nchunk): each block recomputes the C-vector gp (S*K*C FMAs)
// and stores its share of the HW pixels.
template <typename T>
__global__ __launch_bounds__(256) void head_global_bwd_feat_kernel(const float* __restrict__ dz_all, int M, int HW, int C,
                                                                   const int32_t* __restrict__ src, const float* __restrict__ w,
                                                                   int S, int K, T* __restrict__ gfeat) {
    extern __shared__ float gp[];  // [C]
    constexpr int V = 16 / (int)sizeof(T);
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < K; ++k) a += w[((size_t)s * K + k) * C + c] * dz_all[((size_t)s * M + m) * K + k];
        gp[c] = a / (float)HW;
    }
    __syncthreads();
    T* g = gfeat + (size_t)src[m] * HW * C;
    const int nvec = HW * C / V, cv = C / V;                 // host guarantees C % V == 0
    for (int e = blockIdx.y * 256 + threadIdx.x; e < nvec; e += gridDim.y * 256) {
        T pk[V];
        const int c0 = (e % cv) * V;
#pragma unroll
        for (int i = 0; i < V; ++i) pk[i] = from_f32<T>(gp[c0 + i]);
        reinterpret_cast<uint4*>(g)[e] = *reinterpret_cast<const uint4*>(pk);
    }
}

// compute_joint (ref iic_loss.py:74-94) on its own, with the reference's `symmetric` switch: P = sum_n x_n (x) y_n, optionally
// (P + P^T) / 2, divided by its sum.  IIDLoss always symmetrises (the fused kernels above); this pair serves direct callers.
__global__ __launch_bounds__(256) void iic_joint_fwd_kernel(const float* __restrict__ xs, const float* __restrict__ ys, int N, int K,
                                                            int symmetric, float* __restrict__ joint) {
    __shared__ float Ps[kMaxK * kMaxK];
    __shared__ float red[17];
    const int s = blockIdx.x, KK = K * K;
    const float* x = xs + (size_t)s * N * K;
    const float* y = ys + (size_t)s * N * K;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        const int i = e / K, j = e % K;
        float a = 0.f, t = 0.f;
        for (int n = 0; n < N; ++n) {
            a += x[n * K + i] * y[n * K + j];
            t += x[n * K + j] * y[n * K + i];
        }
        Ps[e] = symmetric ? (a + t) / 2.0f : a;
    }
    __syncthreads();
    float z = 0.f;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) z += Ps[e];
    z = block_sum(z, red);
    for (int e = threadIdx.x; e < KK; e += blockDim.x) joint[(size_t)s * KK + e] = Ps[e] / z;
}

// given G = dL/dP:  dL/dQ = (G - <G, P>) / Z with Q the (symmetrised) raw joint; the raw outer-product sum J enters Q as
// (J + J^T)/2 when symmetric -> dL/dJ = (D + D^T)/2, else D;  gx[n,i] = sum_j dJ[i][j] y[n,j],  gy[n,j] = sum_i dJ[i][j] x[n,i].
__global__ __launch_bounds__(256) void iic_joint_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ ys, int N, int K,
                                                            int symmetric, const float* __restrict__ joint, const float* __restrict__ gjoint,
                                                            float* __restrict__ gxs, float* __restrict__ gys) {
    __shared__ float D[kMaxK * kMaxK];
    __shared__ float red[17];
    const int s = blockIdx.x, KK = K * K;
    const float* x = xs + (size_t)s * N * K;
    const float* y = ys + (size_t)s * N * K;
    const float* P = joint + (size_t)s * KK;
    const float* G = gjoint + (size_t)s * KK;
    float z = 0.f;                                   // Z = sum_n (sum_i x)(sum_j y), symmetric or not
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float sx = 0.f, sy = 0.f;
        for (int k = 0; k < K; ++k) { sx += x[n * K + k]; sy += y[n * K + k]; }
        z += sx * sy;
    }
    z = block_sum(z, red);
    float dot = 0.f;
    for (int e = threadIdx.x; e < KK; e += blockDim.x) dot += G[e] * P[e];
    dot = block_sum(dot, red);
    for (int e = threadIdx.x; e < KK; e += blockDim.x) {
        const int i = e / K, j = e % K;
        const float d = (G[e] - dot) / z, dt = (G[j * K + i] - dot) / z;
        D[e] = symmetric ? (d + dt) / 2.0f : d;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
        const int n = e / K, k = e % K;
        float ax = 0.f, ay = 0.f;
        for (int t = 0; t < K; ++t) {
            ax += D[k * K + t] * y[n * K + t];
            ay += D[t * K + k] * x[n * K + t];
        }
        gxs[(size_t)s * N * K + e] = ax;
        gys[(size_t)s * N * K + e] = ay;
    }
}

}  // namespace miseg

using namespace miseg;

extern "C" int miseg_iic_global_joint_fwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K, int symmetric,
                                          float* joint) {
    MISEG_TAPE(miseg_iic_global_joint_fwd, stream, x, y, S, N, K, symmetric, joint);
    MISEG_REQUIRE(x && y && joint, "iic_global_joint_fwd: null pointer");
    MISEG_REQUIRE(S > 0 && N > 0 && K > 0 && K <= kMaxK, "iic_global_joint_fwd: need 0<K<=%d", kMaxK);
    hipLaunchKernelGGL(iic_joint_fwd_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), x, y, (int)N, (int)K, symmetric, joint);
    MISEG_LAUNCH_CHECK("iic_joint_fwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_global_joint_bwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K, int symmetric,
                                          const float* joint, const float* gjoint, float* gx, float* gy) {
    MISEG_TAPE(miseg_iic_global_joint_bwd, stream, x, y, S, N, K, symmetric, joint, gjoint, gx, gy);
    MISEG_REQUIRE(x && y && joint && gjoint && gx && gy, "iic_global_joint_bwd: null pointer");
    MISEG_REQUIRE(S > 0 && N > 0 && K > 0 && K <= kMaxK, "iic_global_joint_bwd: need 0<K<=%d", kMaxK);
    hipLaunchKernelGGL(iic_joint_bwd_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), x, y, (int)N, (int)K, symmetric, joint, gjoint,
                       gx, gy);
    MISEG_LAUNCH_CHECK("iic_joint_bwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_global_fwd_pair(void* stream, const float* prob, int64_t S, int64_t N, int64_t K, float lamb, float* loss,
                                         float* loss_no_lamb, float* joint) {
    MISEG_TAPE(miseg_iic_global_fwd_pair, stream, prob, S, N, K, lamb, loss, loss_no_lamb, joint);
    MISEG_REQUIRE(prob && loss && loss_no_lamb && joint, "iic_global_fwd: null pointer");
    MISEG_REQUIRE(S > 0 && N > 0 && K > 0 && K <= kMaxK, "iic_global_fwd: need 0<K<=%d", kMaxK);
    hipLaunchKernelGGL(iic_global_fwd_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), prob, prob + N * K, (int)N, (int)K, lamb, loss,
                       loss_no_lamb, joint, (long long)(2 * N * K));
    MISEG_LAUNCH_CHECK("iic_global_fwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_global_bwd_pair(void* stream, const float* prob, int64_t S, int64_t N, int64_t K, float lamb, const float* upstream,
                                         float* gprob) {
    MISEG_TAPE(miseg_iic_global_bwd_pair, stream, prob, S, N, K, lamb, upstream, gprob);
    MISEG_REQUIRE(prob && gprob, "iic_global_bwd: null pointer");
    MISEG_REQUIRE(S > 0 && N > 0 && K > 0 && K <= kMaxK, "iic_global_bwd: need 0<K<=%d", kMaxK);
    hipLaunchKernelGGL(iic_global_bwd_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), prob, prob + N * K, (int)N, (int)K, lamb,
                       upstream, gprob, gprob + N * K, (long long)(2 * N * K));
    MISEG_LAUNCH_CHECK("iic_global_bwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_global_fwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K, float lamb,
                                    float* loss, float* loss_no_lamb, float* joint) {
    MISEG_TAPE(miseg_iic_global_fwd, stream, x, y, S, N, K, lamb, loss, loss_no_lamb, joint);
    MISEG_REQUIRE(x && y && loss && loss_no_lamb && joint, "iic_global_fwd: null pointer");
    MISEG_REQUIRE(S > 0 && N > 0 && K > 0 && K <= kMaxK, "iic_global_fwd: need 0<K<=%d", kMaxK);
    hipLaunchKernelGGL(iic_global_fwd_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), x, y, (int)N, (int)K, lamb, loss,
                       loss_no_lamb, joint, (long long)(N * K));
    MISEG_LAUNCH_CHECK("iic_global_fwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_global_bwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K, float lamb,
                                    const float* upstream, float* gx, float* gy) {
    MISEG_TAPE(miseg_iic_global_bwd, stream, x, y, S, N, K, lamb, upstream, gx, gy);
    MISEG_REQUIRE(x && y && gx && gy, "iic_global_bwd: null pointer");
    MISEG_REQUIRE(S > 0 && N > 0 && K > 0 && K <= kMaxK, "iic_global_bwd: need 0<K<=%d", kMaxK);
    hipLaunchKernelGGL(iic_global_bwd_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(stream), x, y, (int)N, (int)K, lamb,
                       upstream, gx, gy, (long long)(N * K));
    MISEG_LAUNCH_CHECK("iic_global_bwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_head_global_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                     const int32_t* src, int64_t M, const float* w, const float* b, int64_t S, int64_t K, float T,
                                     float* pooled, float* prob) {
    MISEG_TAPE(miseg_head_global_fwd, stream, dt, feat, B, H, W, C, src, M, w, b, S, K, T, pooled, prob);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_global_fwd, stream, MISEG_BF16, feat, B, H, W, C, src, M, w, b, S, K, T, pooled, prob);
    MISEG_REQUIRE(feat && src && w && b && pooled && prob, "head_global_fwd: null pointer");
    MISEG_REQUIRE(K > 0 && K <= 64 && M > 0 && S > 0 && C > 0, "head_global_fwd: bad shape (K<=64)");
    hipStream_t st = as_stream(stream);
    if (dt == MISEG_F32)
        hipLaunchKernelGGL(head_pool_kernel<float>, dim3((unsigned)M, (unsigned)cdiv(C, 32)), dim3(256), 0, st, (const float*)feat, (int)(H * W), (int)C, src, pooled);
    else
        hipLaunchKernelGGL(head_pool_kernel<bf16>, dim3((unsigned)M, (unsigned)cdiv(C, 32)), dim3(256), 0, st, (const bf16*)feat, (int)(H * W), (int)C, src, pooled);
    MISEG_LAUNCH_CHECK("head_pool_kernel");
    hipLaunchKernelGGL(head_global_fwd_kernel, dim3((unsigned)M, (unsigned)S), dim3(64), 0, st, pooled, (int)M, (int)C, w, b, (int)K, T, prob);
    MISEG_LAUNCH_CHECK("head_global_fwd_kernel");
    return MISEG_OK;
}

extern "C" int miseg_head_global_bwd(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src, int64_t M,
                                     const float* w, int64_t S, int64_t K, float T, const float* pooled, const float* prob,
                                     const float* gprob, void* gfeat, float* gw, float* gb, float* dz_ws);
extern "C" int miseg_head_global_bwd_rows(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src, int64_t M,
                                          const float* w, int64_t S, int64_t K, float T, const float* pooled, const float* prob,
                                          const float* gprob, void* gfeat_rows, int64_t row0, float* gw, float* gb, float* dz_ws) {
    MISEG_TAPE(miseg_head_global_bwd_rows, stream, dt, B, H, W, C, src, M, w, S, K, T, pooled, prob, gprob, gfeat_rows, row0, gw, gb, dz_ws);
    MISEG_REQUIRE(row0 >= 0 && row0 < B, "head_global_bwd_rows: bad first row");
    const int64_t es = dt == MISEG_F32 ? 4 : 2;
    void* base = gfeat_rows ? static_cast<char*>(gfeat_rows) - row0 * H * W * C * es : nullptr;
    return miseg_head_global_bwd(stream, dt, B, H, W, C, src, M, w, S, K, T, pooled, prob, gprob, base, gw, gb, dz_ws);
}

extern "C" int miseg_head_global_bwd(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src, int64_t M,
                                     const float* w, int64_t S, int64_t K, float T, const float* pooled, const float* prob,
                                     const float* gprob, void* gfeat, float* gw, float* gb, float* dz_ws) {
    MISEG_TAPE(miseg_head_global_bwd, stream, dt, B, H, W, C, src, M, w, S, K, T, pooled, prob, gprob, gfeat, gw, gb, dz_ws);
    MISEG_F16_DISPATCH_ON(dt, miseg_head_global_bwd, stream, MISEG_BF16, B, H, W, C, src, M, w, S, K, T, pooled, prob, gprob, gfeat, gw, gb, dz_ws);
    MISEG_REQUIRE(src && w && pooled && prob && gprob && gw && gb && dz_ws, "head_global_bwd: null pointer");
    MISEG_REQUIRE(K > 0 && K <= 64 && M > 0 && S > 0 && C > 0, "head_global_bwd: bad shape");
    hipStream_t st = as_stream(stream);
    float* dz = dz_ws;
    MISEG_REQUIRE(M * K * 4 <= 64 * 1024, "head_global_bwd: M*K too large for the LDS copy of dz");
    hipLaunchKernelGGL(head_global_bwd_kernel, dim3((unsigned)S, (unsigned)cdiv(K * C, 256)), dim3(256), (size_t)(M * K * 4), st, pooled, (int)M, (int)C, w, (int)S, (int)K, T, prob,
                       gprob, dz, gw, gb);
    MISEG_LAUNCH_CHECK("head_global_bwd_kernel");
    if (gfeat) {
        MISEG_REQUIRE(C % (dt == MISEG_BF16 ? 8 : 4) == 0, "head_global_bwd: C must be a multiple of the 16-byte vector");
        const unsigned fchunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(16, cdiv(H * W * C / (dt == MISEG_BF16 ? 8 : 4), 256)));
        if (dt == MISEG_F32)
            hipLaunchKernelGGL(head_global_bwd_feat_kernel<float>, dim3((unsigned)M, fchunks), dim3(256), (size_t)C * 4, st, dz, (int)M,
                               (int)(H * W), (int)C, src, w, (int)S, (int)K, (float*)gfeat);
        else
            hipLaunchKernelGGL(head_global_bwd_feat_kernel<bf16>, dim3((unsigned)M, fchunks), dim3(256), (size_t)C * 4, st, dz, (int)M,
                               (int)(H * W), (int)C, src, w, (int)S, (int)K, (bf16*)gfeat);
        MISEG_LAUNCH_CHECK("head_global_bwd_feat_kernel");
    }
    return MISEG_OK;
}
