// BatchNorm2d(train/eval) + ReLU (+ fused 2x2 max-pool), their backward, and the small NHWC data
// movers (upsample backward, gradient joins, casts) plus fused Adam.
// ref: contrastyou/arch/unet.py:16-17,19-20,34-35 (BN+ReLU), :61-64 (MaxPool2d(2,2)), :32 (Upsample x2);
//      optimiser: semi_seg/trainer.py:67-72,179-184 (torch.optim.Adam with L2 weight decay).
// All HBM-bound: 16-byte vectors per thread, grid-stride, deterministic two-pass channel sums.
#include "common.h"

namespace miseg {

// ---- statistics -> coefficients.  saved = [mean | invstd | scale | shift] (4*C floats)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ parts, int nparts, int C, float count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                          long long* __restrict__ nbt, float* __restrict__ saved) {
    __shared__ float red[17];
    const int c = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int q = threadIdx.x; q < nparts; q += 256) {
        s1 += parts[((size_t)q * 2 + 0) * C + c];
        s2 += parts[((size_t)q * 2 + 1) * C + c];
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        const float mean = s1 / count;
        float var = s2 / count - mean * mean;  // biased batch variance
        var = fmaxf(var, 0.f);
        const float invstd = rsqrtf(var + eps);
        const float sc = gamma[c] * invstd;
        saved[c] = mean; saved[C + c] = invstd; saved[2 * C + c] = sc; saved[3 * C + c] = beta[c] - mean * sc;
        if (rmean) {
            const float unb = count > 1.f ? var * count / (count - 1.f) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
            if (c == 0 && nbt) nbt[0] += 1;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float* __restrict__ saved) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float invstd = rsqrtf(rvar[c] + eps), sc = gamma[c] * invstd;
        saved[c] = rmean[c]; saved[C + c] = invstd; saved[2 * C + c] = sc; saved[3 * C + c] = beta[c] - rmean[c] * sc;
    }
}

// ---- ACC forms of the apply kernels: the batch statistics arrive as the two fixed-point totals per channel the producing convolution
// accumulated (common.h bn_acc_add); every block works the coefficients out itself (C <= 256 threads, a few double operations each)
// and keeps them in LDS, block 0 also writes `saved` for the backward and moves the running statistics -- bn_finalize_kernel's
// formulas with the variance taken in double.
struct BnAccArgs {
    const unsigned long long* acc;
    const float* gamma; const float* beta; float* rmean; float* rvar; long long* nbt; float* saved;
    float count, eps, momentum;
};
constexpr int kBnAccMaxC = 256;
__device__ __forceinline__ void bn_acc_coeffs(const BnAccArgs& a, int C, float* sc_sh /* LDS [2][kBnAccMaxC] */) {
    for (int c = threadIdx.x; c < C; c += 256) {
        const unsigned long long misfit = a.acc[2 * C];
        const double m = bn_acc_value(a.acc[c], misfit) / (double)a.count, e2 = bn_acc_value(a.acc[C + c], misfit) / (double)a.count;
        const float mean = (float)m, var = fmaxf((float)(e2 - m * m), 0.f);         // biased batch variance (NaN stays NaN: fmaxf(NaN, 0) = 0 would hide a poisoned sum)
        const float varn = (m == m && e2 == e2) ? var : __builtin_nanf("");
        const float invstd = rsqrtf(varn + a.eps), sc = a.gamma[c] * invstd, sh = a.beta[c] - mean * sc;
        sc_sh[c] = sc;
        sc_sh[kBnAccMaxC + c] = sh;
        if (blockIdx.x == 0) {
            a.saved[c] = mean; a.saved[C + c] = invstd; a.saved[2 * C + c] = sc; a.saved[3 * C + c] = sh;
            if (a.rmean) {
                const float unb = a.count > 1.f ? varn * a.count / (a.count - 1.f) : varn;
                a.rmean[c] = (1.f - a.momentum) * a.rmean[c] + a.momentum * mean;
                a.rvar[c] = (1.f - a.momentum) * a.rvar[c] + a.momentum * unb;
                if (c == 0 && a.nbt) a.nbt[0] += 1;
            }
        }
    }
    __syncthreads();
}

// ---- y = relu(raw*scale+shift); optional 2x2 max-pool of y (one thread = one 2x2 window x V channels)
template <typename T, bool POOL, bool ACC = false>
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const T* __restrict__ raw, int N, int H, int W, int C,
                                                          const float* __restrict__ saved, T* __restrict__ y, T* __restrict__ pooled,
                                                          BnAccArgs acc = BnAccArgs{}) {
    constexpr int V = VT<T>::V;
    typedef typename VT<T>::Raw Raw;
    const int CV = C / V;
    __shared__ float cf_lds[ACC ? 2 * kBnAccMaxC : 1];
    if (ACC) bn_acc_coeffs(acc, C, cf_lds);
    const float* scale = ACC ? cf_lds : saved + 2 * C;
    const float* shift = ACC ? cf_lds + kBnAccMaxC : saved + 3 * C;
    if (!POOL) {
        const int64_t total = (int64_t)N * H * W * CV;
        for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int cv = e % CV;
            float f[V];
            VT<T>::unpack(reinterpret_cast<const Raw*>(raw)[e], f);
#pragma unroll
            for (int i = 0; i < V; ++i) f[i] = fmaxf(f[i] * scale[cv * V + i] + shift[cv * V + i], 0.f);
            reinterpret_cast<Raw*>(y)[e] = VT<T>::pack(f);
        }
    } else {
        const int Hp = H / 2, Wp = W / 2;
        const int64_t total = (int64_t)N * Hp * Wp * CV;
        for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int cv = e % CV, wp = (e / CV) % Wp, hp = (e / ((int64_t)CV * Wp)) % Hp, n = e / ((int64_t)CV * Wp * Hp);
            float mx[V];
#pragma unroll
            for (int i = 0; i < V; ++i) mx[i] = -3.4e38f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t o = ((((int64_t)n * H + 2 * hp + (q >> 1)) * W + 2 * wp + (q & 1)) * CV + cv);
                float f[V];
                VT<T>::unpack(reinterpret_cast<const Raw*>(raw)[o], f);
#pragma unroll
                for (int i = 0; i < V; ++i) { f[i] = fmaxf(f[i] * scale[cv * V + i] + shift[cv * V + i], 0.f); }
                // the max is taken on the values as STORED (rounded to T), like torch pooling the bf16 tensor
                Raw pk = VT<T>::pack(f);
                reinterpret_cast<Raw*>(y)[o] = pk;
                VT<T>::unpack(pk, f);
#pragma unroll
                for (int i = 0; i < V; ++i) mx[i] = fmaxf(mx[i], f[i]);
            }
            reinterpret_cast<Raw*>(pooled)[e] = VT<T>::pack(mx);
        }
    }
}

// The pooled form with every load and store instruction contiguous across the wave: a lane owns ONE column pixel of a 2x2 window (both
// rows) for V channels -- lanes (.., wp, column parity c, cv) in that order, so lane L of a row reads vector L of the row --, takes
// the vertical max itself and the horizontal one from its partner lane L ^ CV; the c = 0 lanes store the pooled vector.  (In
// bn_relu_fwd_kernel<T, true> a lane owns the whole window: its two loads of a row are CV vectors apart, every load instruction
// touches half of each cache line it fetches: 4.7 TB/s against the 6.7 of the unpooled kernel.)  Needs 2 CV <= 64 lanes.
template <typename T, bool ACC = false>
__global__ __launch_bounds__(256) void bn_relu_fwd_pool_kernel(const T* __restrict__ raw, int N, int H, int W, int C,
                                                               const float* __restrict__ saved, T* __restrict__ y, T* __restrict__ pooled,
                                                               BnAccArgs acc = BnAccArgs{}) {
    constexpr int V = VT<T>::V;
    typedef typename VT<T>::Raw Raw;
    const int CV = C / V, Hp = H / 2, Wp = W / 2, RV = W * CV;           // RV: vectors per image row
    __shared__ float cf_lds[ACC ? 2 * kBnAccMaxC : 1];
    if (ACC) bn_acc_coeffs(acc, C, cf_lds);
    const float* scale = ACC ? cf_lds : saved + 2 * C;
    const float* shift = ACC ? cf_lds + kBnAccMaxC : saved + 3 * C;
    const int64_t total = (int64_t)N * Hp * RV;                           // one lane per (row pair, column pixel, channel vector)
    const int64_t span = ((total + 255) / 256) * 256;                     // whole blocks run the loop: the shuffles need every lane
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < span; e += (int64_t)gridDim.x * 256) {
        const bool live = e < total;
        const int64_t ee = live ? e : total - 1;
        const int rv = (int)(ee % RV), cv = rv % CV, wcol = rv / CV;
        const int64_t rp = ee / RV;                                       // row pair index = n * Hp + hp
        const int64_t o0 = (rp * 2) * RV + rv, o1 = o0 + RV;
        float f0[V], f1[V], mx[V];
        VT<T>::unpack(reinterpret_cast<const Raw*>(raw)[o0], f0);
        VT<T>::unpack(reinterpret_cast<const Raw*>(raw)[o1], f1);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            f0[i] = fmaxf(f0[i] * scale[cv * V + i] + shift[cv * V + i], 0.f);
            f1[i] = fmaxf(f1[i] * scale[cv * V + i] + shift[cv * V + i], 0.f);
        }
        const Raw p0 = VT<T>::pack(f0), p1 = VT<T>::pack(f1);
        if (live) { reinterpret_cast<Raw*>(y)[o0] = p0; reinterpret_cast<Raw*>(y)[o1] = p1; }
        VT<T>::unpack(p0, f0);                                            // the max is taken on the values as STORED
        VT<T>::unpack(p1, f1);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float v = fmaxf(f0[i], f1[i]);
            mx[i] = fmaxf(v, __shfl_xor(v, CV, 64));                      // the other column of the window (W even: same row pair)
        }
        if (live && !(wcol & 1)) reinterpret_cast<Raw*>(pooled)[(rp * Wp + (wcol >> 1)) * CV + cv] = VT<T>::pack(mx);
    }
}

// ---- backward.  dz = (gy [+ routed gpool]) * (y > 0) is never materialised and y is never read: both passes recompute
//      y = T(relu(raw*scale+shift)) exactly as bn_relu_fwd_kernel stored it (same fp32 expression, same rounding to T), which
//      gives the ReLU mask and, for the pooled layers, the first-max routing.  Pass 1 reads (raw, gy[, gpool]) and reduces
//      sum(dz), sum(dz*xhat) per block; pass 2 re-reads them and writes graw = a*(dz - b - xhat*c).
//      One thread = V channels of one pixel (POOL: of one 2x2 window; H, W even).
// A second gradient of the activation for a contiguous run of samples (pixels [lo, hi) of the batch), added in fp32 in the loader:
// the tapped feature maps feed the logits layer AND a cluster head (ref semi_seg/epocher.py:258-273), autograd would add the two
// gradients with an elementwise kernel over the whole batch (and the head would zero-fill the samples it does not touch).
template <typename T> struct BnGy2 {
    const T* g;          // null: none; else [hi - lo][C], pixel p of the batch at g + (p - lo) * C
    int64_t lo, hi;
};
template <typename T, bool POOL, typename F>
__device__ __forceinline__ void bn_dz_foreach(const T* __restrict__ raw, const T* __restrict__ gy, const T* __restrict__ gpool, int N, int H,
                                              int W, int C, const float* __restrict__ saved, int64_t first, int64_t step, int cv, BnGy2<T> g2, F&& f) {
    constexpr int V = VT<T>::V;
    typedef typename VT<T>::Raw Raw;
    const int CV = C / V;
    float sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { sc[i] = saved[2 * C + cv * V + i]; sh[i] = saved[3 * C + cv * V + i]; }
    if (!POOL) {
        const int64_t npix = (int64_t)N * H * W;
        // U pixels per trip: 2*U 16-byte loads in flight per thread (one pair per trip leaves the loop latency-bound:
        // 16 waves x 64 lanes x 32 B = 32 KB outstanding per CU)
        constexpr int U = 4;
        for (int64_t px = first; px < npix; px += U * step) {
            Raw rr[U], gg[U], g2v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t pu = px + u * step;
                const int64_t pc = pu < npix ? pu : px;
                const int64_t o = pc * CV + cv;
                rr[u] = reinterpret_cast<const Raw*>(raw)[o];
                gg[u] = reinterpret_cast<const Raw*>(gy)[o];
                if (g2.g && pc >= g2.lo && pc < g2.hi) g2v[u] = reinterpret_cast<const Raw*>(g2.g)[(pc - g2.lo) * CV + cv];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t pu = px + u * step;
                if (pu < npix) {
                    float fr[V], fg[V], fy[V];
                    VT<T>::unpack(rr[u], fr);
                    VT<T>::unpack(gg[u], fg);
                    if (g2.g && pu >= g2.lo && pu < g2.hi) {
                        float f2[V];
                        VT<T>::unpack(g2v[u], f2);
#pragma unroll
                        for (int i = 0; i < V; ++i) fg[i] += f2[i];
                    }
#pragma unroll
                    for (int i = 0; i < V; ++i) fy[i] = fmaxf(fr[i] * sc[i] + sh[i], 0.f);
                    VT<T>::unpack(VT<T>::pack(fy), fy);
#pragma unroll
                    for (int i = 0; i < V; ++i) fg[i] = fy[i] > 0.f ? fg[i] : 0.f;
                    f(pu * CV + cv, fr, fg);
                }
            }
        }
    } else {
        const int Hp = H / 2, Wp = W / 2;
        const int64_t nwin = (int64_t)N * Hp * Wp;
        for (int64_t wi = first; wi < nwin; wi += step) {
            const int wp = wi % Wp, hp = (wi / Wp) % Hp, n = wi / ((int64_t)Wp * Hp);
            float gp[V], fr[4][V], ys[4][V], best[V];
            int arg[V];
            VT<T>::unpack(reinterpret_cast<const Raw*>(gpool)[wi * CV + cv], gp);
            int64_t offs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                offs[q] = ((((int64_t)n * H + 2 * hp + (q >> 1)) * W + 2 * wp + (q & 1)) * CV + cv);
                VT<T>::unpack(reinterpret_cast<const Raw*>(raw)[offs[q]], fr[q]);
#pragma unroll
                for (int i = 0; i < V; ++i) ys[q][i] = fmaxf(fr[q][i] * sc[i] + sh[i], 0.f);
                VT<T>::unpack(VT<T>::pack(ys[q]), ys[q]);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                best[i] = ys[0][i]; arg[i] = 0;
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    if (ys[q][i] > best[i]) { best[i] = ys[q][i]; arg[i] = q; }  // first max in scan order
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float fg[V];
                if (gy) VT<T>::unpack(reinterpret_cast<const Raw*>(gy)[offs[q]], fg);
                else {
#pragma unroll
                    for (int i = 0; i < V; ++i) fg[i] = 0.f;
                }
                {
                    const int64_t pq = offs[q] / CV;          // pixel index of this corner of the window
                    if (g2.g && pq >= g2.lo && pq < g2.hi) {
                        float f2[V];
                        VT<T>::unpack(reinterpret_cast<const Raw*>(g2.g)[(pq - g2.lo) * CV + cv], f2);
#pragma unroll
                        for (int i = 0; i < V; ++i) fg[i] += f2[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float g = fg[i] + (arg[i] == q ? gp[i] : 0.f);
                    fg[i] = ys[q][i] > 0.f ? g : 0.f;
                }
                f(offs[q], fr[q], fg);
            }
        }
    }
}

struct BnBwdFinish {
    unsigned int* counter;           // null: bn_bwd_finalize_kernel is launched separately
    const float* gamma; float* coeffs; float* ggamma; float* gbeta;
    float count; int training;
    float* bwd_coef;                 // null or [6][C] (common.h, BnLoad): the coefficients of the loader-fused backward
    unsigned long long* acc;         // non-null: the sums leave as two-tier fixed-point atomic adds (common.h bn_acc2_add), no partial rows
};
// what one channel's sums become: coeffs[3][C] for bn_bwd_apply_kernel and / or bwd_coef[6][C] for the fused loaders
__device__ __forceinline__ void bn_bwd_coeffs(int c, int C, float s1, float s2, float count, int training, const float* __restrict__ gamma,
                                              const float* __restrict__ saved, float* __restrict__ coeffs, float* __restrict__ bwd_coef,
                                              float* __restrict__ ggamma, float* __restrict__ gbeta) {
    gbeta[c] = s1;
    ggamma[c] = s2;
    const float a = gamma[c] * saved[C + c], b = training ? s1 / count : 0.f, cc = training ? s2 / count : 0.f;
    if (coeffs) { coeffs[c] = a; coeffs[C + c] = b; coeffs[2 * C + c] = cc; }
    if (bwd_coef) {
        bwd_coef[c] = saved[2 * C + c]; bwd_coef[C + c] = saved[3 * C + c]; bwd_coef[2 * C + c] = saved[c];
        bwd_coef[3 * C + c] = a; bwd_coef[4 * C + c] = -(a * b); bwd_coef[5 * C + c] = -(a * saved[C + c] * cc);
    }
}

template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const T* __restrict__ raw, const T* __restrict__ gy,
                                                                 const T* __restrict__ gpool, int N, int H, int W, int C,
                                                                 const float* __restrict__ saved, float* __restrict__ parts, BnBwdFinish fin,
                                                                 BnGy2<T> g2) {
    constexpr int V = VT<T>::V;
    extern __shared__ float sacc[];  // [256][2*V] transposed reduce
    const int CV = C / V;
    // threads keep a fixed channel vector and stride over pixels (CV | 256, so gid % CV is the same for every step)
    const int64_t gthreads = (int64_t)gridDim.x * 256, gid = blockIdx.x * 256LL + threadIdx.x;
    const int cv = gid % CV;
    float mean[V], invstd[V], a1[V], a2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a1[i] = a2[i] = 0.f; mean[i] = saved[cv * V + i]; invstd[i] = saved[C + cv * V + i]; }
    bn_dz_foreach<T, POOL>(raw, gy, gpool, N, H, W, C, saved, gid / CV, gthreads / CV, cv, g2, [&](int64_t, const float* fr, const float* dz) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            a1[i] += dz[i];
            a2[i] += dz[i] * (fr[i] - mean[i]) * invstd[i];
        }
    });
    // block reduce over the 256/CV threads that share a channel vector (CV | 256, block start is a multiple of CV):
    // sacc[tid][2V]; output (cvv, j) = sum over rows t = cvv (mod CV).  All 256 threads take part: 2C outputs x nparts
    // partial sums first (the old form left this to CV threads -- 2 of 256 for a 16-channel layer), then a fixed-order
    // combine: deterministic, independent of the grid.
#pragma unroll
    for (int i = 0; i < V; ++i) { sacc[threadIdx.x * 2 * V + i] = a1[i]; sacc[threadIdx.x * 2 * V + V + i] = a2[i]; }
    __syncthreads();
    const int nout = 2 * C, rows = 256 / CV;                 // outputs of this block; rows per output
    const int nparts = nout >= 256 ? 1 : 256 / nout;         // threads cooperating on one output
    __shared__ float spart[256];
    for (int o0 = 0; o0 < nout; o0 += 256) {
        const int o = o0 + (nparts == 1 ? (int)threadIdx.x : (int)threadIdx.x % nout), part = nparts == 1 ? 0 : (int)threadIdx.x / nout;
        float sum = 0.f;
        if (o < nout && part < nparts) {
            const int cvv = o / (2 * V), j = o % (2 * V);
            for (int m = part; m < rows; m += nparts) sum += sacc[(cvv + CV * m) * 2 * V + j];
        }
        if (o0) __syncthreads();
        spart[threadIdx.x] = sum;
        __syncthreads();
        if ((int)threadIdx.x < nout - o0 && (int)threadIdx.x < 256 && (nparts == 1 || (int)threadIdx.x < nout)) {
            const int oo = o0 + (int)threadIdx.x;
            float tot = 0.f;
            for (int q = 0; q < nparts; ++q) tot += spart[nparts == 1 ? (int)threadIdx.x : q * nout + (int)threadIdx.x];
            const int cvv = oo / (2 * V), j = oo % (2 * V);
            if (fin.acc) bn_acc2_add(fin.acc, (j >= V) * C + cvv * V + (j % V), C, tot);
            else store_part(&parts[((size_t)blockIdx.x * 2 + (j >= V)) * C + cvv * V + (j % V)], tot);
        }
    }
    if (fin.counter && last_block_arrives(fin.counter, gridDim.x)) {      // the body of bn_bwd_finalize_kernel, by the last block
        const float* sums = block_column_sums(parts, (int)gridDim.x, 2 * C, sacc);     // sacc: 256 * 2V >= 1024 + 2C floats (host-checked)
        for (int c = threadIdx.x; c < C; c += 256)
            bn_bwd_coeffs(c, C, sums[c], sums[C + c], fin.count, fin.training, fin.gamma, saved, fin.coeffs, fin.bwd_coef, fin.ggamma, fin.gbeta);
        last_block_done(fin.counter);
    }
}

// coeffs[3][C]: a = gamma*invstd, b = mean(dz), c = mean(dz*xhat) (training) or b = c = 0 (eval); also ggamma, gbeta
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ parts, int nparts, int C, float count,
                                                              const float* __restrict__ gamma, const float* __restrict__ saved,
                                                              int training, float* __restrict__ coeffs, float* __restrict__ ggamma,
                                                              float* __restrict__ gbeta, float* __restrict__ bwd_coef) {
    __shared__ float red[17];
    const int c = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int q = threadIdx.x; q < nparts; q += 256) {
        s1 += parts[((size_t)q * 2 + 0) * C + c];
        s2 += parts[((size_t)q * 2 + 1) * C + c];
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) bn_bwd_coeffs(c, C, s1, s2, count, training, gamma, saved, coeffs, bwd_coef, ggamma, gbeta);
}

// graw = a*(dz - b - xhat*c), dz recomputed from (raw, gy[, gpool]).  ACC: the coefficients come from the reduce kernel's fixed-point
// accumulator (every block works them out in a prologue, block 0 also writes ggamma / gbeta): no finalize launch.
struct BnBwdAccArgs {
    const unsigned long long* acc;
    const float* gamma; float* ggamma; float* gbeta;
    float count; int training;
};
template <typename T, bool POOL, bool ACC = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ raw, const T* __restrict__ gy, const T* __restrict__ gpool,
                                                           int N, int H, int W, int C, const float* __restrict__ saved,
                                                           const float* __restrict__ coeffs, T* __restrict__ graw, BnGy2<T> g2,
                                                           BnBwdAccArgs ba = BnBwdAccArgs{}) {
    constexpr int V = VT<T>::V;
    typedef typename VT<T>::Raw Raw;
    const int CV = C / V;
    const int64_t gthreads = (int64_t)gridDim.x * 256, gid = blockIdx.x * 256LL + threadIdx.x;
    const int cv = gid % CV;
    __shared__ float cf_lds[ACC ? 3 * kBnAccMaxC : 1];
    if (ACC) {
        for (int c = threadIdx.x; c < C; c += 256) {
            const float s1 = bn_acc2_value(ba.acc, c, C), s2 = bn_acc2_value(ba.acc, C + c, C);
            cf_lds[c] = ba.gamma[c] * saved[C + c];
            cf_lds[kBnAccMaxC + c] = ba.training ? s1 / ba.count : 0.f;
            cf_lds[2 * kBnAccMaxC + c] = ba.training ? s2 / ba.count : 0.f;
            if (blockIdx.x == 0) { ba.gbeta[c] = s1; ba.ggamma[c] = s2; }
        }
        __syncthreads();
    }
    float mean[V], invstd[V], ca[V], cb[V], cc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = cv * V + i;
        mean[i] = saved[c]; invstd[i] = saved[C + c];
        if (ACC) { ca[i] = cf_lds[c]; cb[i] = cf_lds[kBnAccMaxC + c]; cc[i] = cf_lds[2 * kBnAccMaxC + c]; }
        else { ca[i] = coeffs[c]; cb[i] = coeffs[C + c]; cc[i] = coeffs[2 * C + c]; }
    }
    bn_dz_foreach<T, POOL>(raw, gy, gpool, N, H, W, C, saved, gid / CV, gthreads / CV, cv, g2, [&](int64_t o, const float* fr, const float* dz) {
        float fd[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float xhat = (fr[i] - mean[i]) * invstd[i];
            fd[i] = ca[i] * (dz[i] - cb[i] - xhat * cc[i]);
        }
        reinterpret_cast<Raw*>(graw)[o] = VT<T>::pack(fd);
    });
}

template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* __restrict__ in, int N, int H, int W, int C, T* __restrict__ out, int accumulate) {
    constexpr int V = VT<T>::V;
    typedef typename VT<T>::Raw Raw;
    const int CV = C / V, Hp = H / 2, Wp = W / 2;
    const int64_t total = (int64_t)N * Hp * Wp * CV;
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int cv = e % CV, wp = (e / CV) % Wp, hp = (e / ((int64_t)CV * Wp)) % Hp, n = e / ((int64_t)CV * Wp * Hp);
        float s[V];
        if (accumulate) VT<T>::unpack(reinterpret_cast<const Raw*>(out)[e], s);
        else {
#pragma unroll
            for (int i = 0; i < V; ++i) s[i] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f[V];
            VT<T>::unpack(reinterpret_cast<const Raw*>(in)[((((int64_t)n * H + 2 * hp + (q >> 1)) * W + 2 * wp + (q & 1)) * CV + cv)], f);
#pragma unroll
            for (int i = 0; i < V; ++i) s[i] += f[i];
        }
        reinterpret_cast<Raw*>(out)[e] = VT<T>::pack(s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t nvec) {
    constexpr int V = VT<T>::V;
    typedef typename VT<T>::Raw Raw;
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < nvec; e += (int64_t)gridDim.x * 256) {
        float a[V], b[V];
        VT<T>::unpack(reinterpret_cast<const Raw*>(src)[e], a);
        VT<T>::unpack(reinterpret_cast<const Raw*>(dst)[e], b);
#pragma unroll
        for (int i = 0; i < V; ++i) b[i] += a[i];
        reinterpret_cast<Raw*>(dst)[e] = VT<T>::pack(b);
    }
}

// image [npix] (1 channel) -> [npix][CP] with channel 0 = value, rest 0 (stem input padded to a full vector)
template <typename TI, typename TO>
__global__ void cast_pad_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t npix, int Cin, int CP) {
    const int64_t total = npix * CP;
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = e % CP;
        out[e] = from_f32<TO>(c < Cin ? to_f32(in[(e / CP) * Cin + c]) : 0.f);
    }
}

// the shipped case: one channel -> one 16-byte vector of a 16-bit type; a lane reads its pixel's float and stores the whole vector
// (the generic kernel above stores 2 bytes per lane: 30 us for this 50 MB tensor at cfg2, 2 TB/s)
template <typename TO>
__global__ __launch_bounds__(256) void cast_pad_c1_kernel(const float* __restrict__ in, TO* __restrict__ out, int64_t npix) {
    static_assert(sizeof(TO) == 2, "one channel padded to 8 x 16 bit");
    for (int64_t p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const TO v = from_f32<TO>(in[p]);
        reinterpret_cast<uint4*>(out)[p] = make_uint4((unsigned)*reinterpret_cast<const unsigned short*>(&v), 0u, 0u, 0u);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float b1, float b2, const float* __restrict__ hyper,
                                                   float inv_scale, const float* __restrict__ guard, int nguard) {
    // guard: the iteration's deferred-check flags (non-zero or NaN = a check failed).  A failed check leaves parameters and moments
    // untouched, so the exception the host raises one iteration later finds the state the reference -- which raises before
    // backward (iic_loss.py:147-148) -- would have left.
    for (int i = 0; i < nguard; ++i)
        if (!(guard[i] == 0.f)) return;
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1], eps = hyper[2], wd = hyper[3];
    if (inv_scale < 0.f) inv_scale = hyper[4];       // the loss scale lives on the device (dynamic scale under a replayed launch tape)
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * inv_scale + wd * p[i];
        float mi = m[i] * b1 + (1.f - b1) * gi;
        float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

static inline int ew_blocks(int64_t n) {
    static const int64_t cap = [] { const char* e = getenv("MISEG_EW_BLOCKS"); return e ? atoll(e) : 8192LL; }();      // grid-stride elementwise kernels
    return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), cap));
}
static const int64_t kFinishFloats = [] { const char* e = getenv("MISEG_FINISH_FLOATS"); return e ? atoll(e) : 65536LL; }();
static inline int red_blocks(int64_t npix, int CV) {
    (void)CV;
    static const int cap = [] { const char* e = getenv("MISEG_RED_BLOCKS"); return e ? atoi(e) : 512; }();
    return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(npix, 64), cap));
}

}  // namespace miseg

using namespace miseg;

extern "C" int miseg_bn_finalize(void* stream, const float* parts, int64_t nparts, int64_t C, int64_t count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* rmean, float* rvar, int64_t* nbt, float* saved) {
    MISEG_TAPE(miseg_bn_finalize, stream, parts, nparts, C, count, gamma, beta, eps, momentum, rmean, rvar, nbt, saved);
    MISEG_REQUIRE(parts && gamma && beta && saved && C > 0 && nparts > 0 && count > 0, "bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(256), 0, as_stream(stream), parts, (int)nparts, (int)C, (float)count, gamma,
                       beta, eps, momentum, rmean, rvar, (long long*)nbt, saved);
    MISEG_LAUNCH_CHECK("bn_finalize_kernel");
    return MISEG_OK;
}

extern "C" int miseg_bn_eval_coeffs(void* stream, int64_t C, const float* gamma, const float* beta, float eps, const float* rmean,
                                    const float* rvar, float* saved) {
    MISEG_TAPE(miseg_bn_eval_coeffs, stream, C, gamma, beta, eps, rmean, rvar, saved);
    MISEG_REQUIRE(gamma && beta && rmean && rvar && saved && C > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, as_stream(stream), (int)C, gamma, beta, eps, rmean, rvar, saved);
    MISEG_LAUNCH_CHECK("bn_eval_coeffs_kernel");
    return MISEG_OK;
}

extern "C" int miseg_bn_relu_fwd(void* stream, int dt, const void* raw, int64_t N, int64_t H, int64_t W, int64_t C, const float* saved,
                                 void* y, void* pooled) {
    MISEG_TAPE(miseg_bn_relu_fwd, stream, dt, raw, N, H, W, C, saved, y, pooled);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_fwd, stream, MISEG_BF16, raw, N, H, W, C, saved, y, pooled);
    MISEG_REQUIRE(raw && saved && y, "bn_relu_fwd: null pointer");
    const int V = dt == MISEG_BF16 ? 8 : 4;
    MISEG_REQUIRE(C % V == 0, "bn_relu_fwd: C must be a multiple of %d", V);
    MISEG_REQUIRE(!pooled || (H % 2 == 0 && W % 2 == 0), "bn_relu_fwd: pooling needs even H, W");
    hipStream_t st = as_stream(stream);
    const int64_t total = pooled ? N * (H / 2) * (W / 2) * (C / V) : N * H * W * (C / V);
    const int nb = ew_blocks(total);
    const int CV = (int)(C / V);
    if (pooled && CV <= 32 && (CV & (CV - 1)) == 0) {     // the coalesced pooled form: the partner lane L ^ CV is in the same wave
        const int nbp = ew_blocks(N * (H / 2) * W * CV);
        if (dt == MISEG_F32) hipLaunchKernelGGL(bn_relu_fwd_pool_kernel<float>, dim3(nbp), dim3(256), 0, st, (const float*)raw, (int)N, (int)H, (int)W, (int)C, saved, (float*)y, (float*)pooled);
        else if (dt == MISEG_BF16) hipLaunchKernelGGL(bn_relu_fwd_pool_kernel<bf16>, dim3(nbp), dim3(256), 0, st, (const bf16*)raw, (int)N, (int)H, (int)W, (int)C, saved, (bf16*)y, (bf16*)pooled);
        else return fail(MISEG_E_INVALID, "bn_relu_fwd: bad dtype");
        MISEG_LAUNCH_CHECK("bn_relu_fwd_pool_kernel");
        return MISEG_OK;
    }
    if (dt == MISEG_F32) {
        if (pooled) hipLaunchKernelGGL((bn_relu_fwd_kernel<float, true>), dim3(nb), dim3(256), 0, st, (const float*)raw, (int)N, (int)H, (int)W, (int)C, saved, (float*)y, (float*)pooled);
        else hipLaunchKernelGGL((bn_relu_fwd_kernel<float, false>), dim3(nb), dim3(256), 0, st, (const float*)raw, (int)N, (int)H, (int)W, (int)C, saved, (float*)y, (float*)nullptr);
    } else if (dt == MISEG_BF16) {
        if (pooled) hipLaunchKernelGGL((bn_relu_fwd_kernel<bf16, true>), dim3(nb), dim3(256), 0, st, (const bf16*)raw, (int)N, (int)H, (int)W, (int)C, saved, (bf16*)y, (bf16*)pooled);
        else hipLaunchKernelGGL((bn_relu_fwd_kernel<bf16, false>), dim3(nb), dim3(256), 0, st, (const bf16*)raw, (int)N, (int)H, (int)W, (int)C, saved, (bf16*)y, (bf16*)nullptr);
    } else return fail(MISEG_E_INVALID, "bn_relu_fwd: bad dtype");
    MISEG_LAUNCH_CHECK("bn_relu_fwd_kernel");
    return MISEG_OK;
}

// miseg_bn_finalize + miseg_bn_relu_fwd in one launch, the statistics read from the accumulator miseg_conv3x3_fwd_acc filled
extern "C" int miseg_bn_relu_fwd_acc(void* stream, int dt, const void* raw, int64_t N, int64_t H, int64_t W, int64_t C, const void* acc,
                                     const float* gamma, const float* beta, float eps, float momentum, float* rmean, float* rvar, int64_t* nbt,
                                     float* saved, void* y, void* pooled) {
    MISEG_TAPE(miseg_bn_relu_fwd_acc, stream, dt, raw, N, H, W, C, acc, gamma, beta, eps, momentum, rmean, rvar, nbt, saved, y, pooled);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_fwd_acc, stream, MISEG_BF16, raw, N, H, W, C, acc, gamma, beta, eps, momentum, rmean, rvar, nbt, saved, y, pooled);
    MISEG_REQUIRE(raw && acc && gamma && beta && saved && y, "bn_relu_fwd_acc: null pointer");
    const int V = dt == MISEG_BF16 ? 8 : 4;
    MISEG_REQUIRE(C % V == 0 && C <= kBnAccMaxC, "bn_relu_fwd_acc: C must be a multiple of %d and <= %d", V, kBnAccMaxC);
    MISEG_REQUIRE(!pooled || (H % 2 == 0 && W % 2 == 0), "bn_relu_fwd_acc: pooling needs even H, W");
    MISEG_REQUIRE((rmean == nullptr) == (rvar == nullptr), "bn_relu_fwd_acc: running mean and variance come together");
    hipStream_t st = as_stream(stream);
    const BnAccArgs a{static_cast<const unsigned long long*>(acc), gamma, beta, rmean, rvar, (long long*)nbt, saved, (float)(N * H * W), eps, momentum};
    const int64_t total = pooled ? N * (H / 2) * (W / 2) * (C / V) : N * H * W * (C / V);
    const int nb = ew_blocks(total);
    const int CV = (int)(C / V);
    if (pooled && CV <= 32 && (CV & (CV - 1)) == 0) {
        const int nbp = ew_blocks(N * (H / 2) * W * CV);
        if (dt == MISEG_F32) hipLaunchKernelGGL((bn_relu_fwd_pool_kernel<float, true>), dim3(nbp), dim3(256), 0, st, (const float*)raw, (int)N, (int)H, (int)W, (int)C, saved, (float*)y, (float*)pooled, a);
        else if (dt == MISEG_BF16) hipLaunchKernelGGL((bn_relu_fwd_pool_kernel<bf16, true>), dim3(nbp), dim3(256), 0, st, (const bf16*)raw, (int)N, (int)H, (int)W, (int)C, saved, (bf16*)y, (bf16*)pooled, a);
        else return fail(MISEG_E_INVALID, "bn_relu_fwd_acc: bad dtype");
        MISEG_LAUNCH_CHECK("bn_relu_fwd_pool_kernel");
        return MISEG_OK;
    }
    if (dt == MISEG_F32) {
        if (pooled) hipLaunchKernelGGL((bn_relu_fwd_kernel<float, true, true>), dim3(nb), dim3(256), 0, st, (const float*)raw, (int)N, (int)H, (int)W, (int)C, saved, (float*)y, (float*)pooled, a);
        else hipLaunchKernelGGL((bn_relu_fwd_kernel<float, false, true>), dim3(nb), dim3(256), 0, st, (const float*)raw, (int)N, (int)H, (int)W, (int)C, saved, (float*)y, (float*)nullptr, a);
    } else if (dt == MISEG_BF16) {
        if (pooled) hipLaunchKernelGGL((bn_relu_fwd_kernel<bf16, true, true>), dim3(nb), dim3(256), 0, st, (const bf16*)raw, (int)N, (int)H, (int)W, (int)C, saved, (bf16*)y, (bf16*)pooled, a);
        else hipLaunchKernelGGL((bn_relu_fwd_kernel<bf16, false, true>), dim3(nb), dim3(256), 0, st, (const bf16*)raw, (int)N, (int)H, (int)W, (int)C, saved, (bf16*)y, (bf16*)nullptr, a);
    } else return fail(MISEG_E_INVALID, "bn_relu_fwd_acc: bad dtype");
    MISEG_LAUNCH_CHECK("bn_relu_fwd_kernel");
    return MISEG_OK;
}

extern "C" int64_t miseg_bn_bwd_ws_bytes(int64_t N, int64_t H, int64_t W, int64_t C) {
    return ((int64_t)red_blocks(N * H * W, 1) * 2 * C + 3 * C) * 4;
}

// reduce (+ finalize) (+ apply): graw == null stops after the statistics, which then leave as bwd_coef for the fused loaders of conv.hip
static int bn_relu_bwd_impl(void* stream, int dt, const void* raw, const void* gy, const void* gpool, int64_t N, int64_t H, int64_t W, int64_t C,
                            const float* gamma, const float* saved, int training, void* graw, float* ggamma, float* gbeta, void* ws,
                            int64_t ws_bytes, int32_t* sync_counter, float* bwd_coef, const void* gy2 = nullptr, int64_t n2_begin = 0,
                            int64_t n2_end = 0, void* acc = nullptr) {
    MISEG_REQUIRE(raw && (gy || gpool) && gamma && saved && (graw || bwd_coef) && ggamma && gbeta && ws, "bn_relu_bwd: null pointer");
    MISEG_REQUIRE(!acc || (graw && !bwd_coef && !sync_counter && C <= kBnAccMaxC && ((uintptr_t)acc & 7) == 0), "bn_relu_bwd_acc: needs the apply pass, C <= 256 and an 8-byte aligned accumulator");
    MISEG_REQUIRE(!gy2 || (0 <= n2_begin && n2_begin < n2_end && n2_end <= N && (gy || gpool)), "bn_relu_bwd: bad sample range of the second gradient");
    const int V = dt == MISEG_BF16 ? 8 : 4;
    const int CV = (int)(C / V);
    MISEG_REQUIRE(C % V == 0 && 256 % CV == 0, "bn_relu_bwd: C/%d must divide 256", V);
    MISEG_REQUIRE(!gpool || (H % 2 == 0 && W % 2 == 0), "bn_relu_bwd: pooling needs even H, W");
    MISEG_REQUIRE(ws_bytes >= miseg_bn_bwd_ws_bytes(N, H, W, C), "bn_relu_bwd: workspace too small");
    hipStream_t st = as_stream(stream);
    const int64_t npix = N * H * W;
    int nb = red_blocks(npix, CV);
    if (sync_counter && C % 4 == 0 && C <= 256) nb = (int)std::max<int64_t>(1, std::min<int64_t>(nb, kFinishFloats / (2 * C)));   // wide layers are small
    float* parts = (float*)ws;
    float* coeffs = parts + (size_t)nb * 2 * C;
    // the last reduce block finishes the statistics itself when the caller lends a counter and the partial matrix is small enough for
    // one block (common.h, "last block finishes"); its scratch is the reduce's own LDS, widened to 1024 + 2C floats if need be
    const bool finish = sync_counter && C % 4 == 0 && C <= 256 && (int64_t)nb * 2 * C <= kFinishFloats;
    const size_t lb = std::max<size_t>((size_t)256 * 2 * V * 4, finish ? (size_t)(1024 + 2 * C) * 4 : 0);
    BnBwdFinish fin{finish ? reinterpret_cast<unsigned int*>(sync_counter) : nullptr, gamma, coeffs, ggamma, gbeta, (float)npix, training, bwd_coef,
                    static_cast<unsigned long long*>(acc)};
    MISEG_REQUIRE(!acc || nb <= kBnAccMaxBlocks, "bn_relu_bwd_acc: %d reduce blocks exceed the accumulator's no-wrap bound", nb);
    const int64_t p2lo = n2_begin * H * W, p2hi = n2_end * H * W;
#define RED(TT, POOL) hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<TT, POOL>), dim3(nb), dim3(256), lb, st, (const TT*)raw, (const TT*)gy, (const TT*)gpool, (int)N, (int)H, (int)W, (int)C, saved, parts, fin, BnGy2<TT>{(const TT*)gy2, p2lo, p2hi})
    if (dt == MISEG_F32) { if (gpool) RED(float, true); else RED(float, false); }
    else if (dt == MISEG_BF16) { if (gpool) RED(bf16, true); else RED(bf16, false); }
    else return fail(MISEG_E_INVALID, "bn_relu_bwd: bad dtype");
#undef RED
    MISEG_LAUNCH_CHECK("bn_relu_bwd_reduce_kernel");
    if (!finish && !acc) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(256), 0, st, parts, nb, (int)C, (float)npix, gamma, saved, training, coeffs, ggamma, gbeta, bwd_coef);
        MISEG_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    }
    if (!graw) return MISEG_OK;
    // elementwise pass: many more blocks than the reduce (no partials to bound), same thread -> channel-vector mapping
    const int na = ew_blocks((gpool ? npix / 4 : npix) * CV);
    const BnBwdAccArgs ba{static_cast<const unsigned long long*>(acc), gamma, ggamma, gbeta, (float)npix, training};
#define APP(TT, POOL, ACCV) hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, POOL, ACCV>), dim3(na), dim3(256), 0, st, (const TT*)raw, (const TT*)gy, (const TT*)gpool, (int)N, (int)H, (int)W, (int)C, saved, coeffs, (TT*)graw, BnGy2<TT>{(const TT*)gy2, p2lo, p2hi}, ba)
    if (acc) {
        if (dt == MISEG_F32) { if (gpool) APP(float, true, true); else APP(float, false, true); }
        else { if (gpool) APP(bf16, true, true); else APP(bf16, false, true); }
    } else {
        if (dt == MISEG_F32) { if (gpool) APP(float, true, false); else APP(float, false, false); }
        else { if (gpool) APP(bf16, true, false); else APP(bf16, false, false); }
    }
#undef APP
    MISEG_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return MISEG_OK;
}

extern "C" int miseg_bn_relu_bwd_sync(void* stream, int dt, const void* raw, const void* y, const void* gy, const void* gpool, int64_t N,
                                      int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved, int training, void* graw,
                                      float* ggamma, float* gbeta, void* ws, int64_t ws_bytes, int32_t* sync_counter) {
    MISEG_TAPE(miseg_bn_relu_bwd_sync, stream, dt, raw, y, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes, sync_counter);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_bwd_sync, stream, MISEG_BF16, raw, y, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws,
                          ws_bytes, sync_counter);
    (void)y;   // kept in the signature; the ReLU mask and pool routing are recomputed from raw (see bn_dz_foreach)
    MISEG_REQUIRE(graw, "bn_relu_bwd: null pointer");
    return bn_relu_bwd_impl(stream, dt, raw, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes, sync_counter, nullptr);
}

extern "C" int miseg_bn_relu_bwd_dual(void* stream, int dt, const void* raw, const void* gy, const void* gpool, const void* gy2, int64_t n2_begin,
                                      int64_t n2_end, int64_t N, int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved, int training,
                                      void* graw, float* ggamma, float* gbeta, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_bn_relu_bwd_dual, stream, dt, raw, gy, gpool, gy2, n2_begin, n2_end, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_bwd_dual, stream, MISEG_BF16, raw, gy, gpool, gy2, n2_begin, n2_end, N, H, W, C, gamma, saved, training, graw,
                          ggamma, gbeta, ws, ws_bytes);
    MISEG_REQUIRE(graw, "bn_relu_bwd_dual: null pointer");
    return bn_relu_bwd_impl(stream, dt, raw, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes, nullptr, nullptr, gy2, n2_begin, n2_end);
}

// miseg_bn_relu_bwd_dual in two launches: the reduce kernel's blocks add their sums into acc (uint64[4 C + 2], zero before the launch; two
// fixed-point tiers, common.h bn_acc2_add), the apply kernel turns them into coefficients itself -- no partial rows, no finalize launch
extern "C" int miseg_bn_relu_bwd_dual_acc(void* stream, int dt, const void* raw, const void* gy, const void* gpool, const void* gy2, int64_t n2_begin,
                                          int64_t n2_end, int64_t N, int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved,
                                          int training, void* graw, float* ggamma, float* gbeta, void* ws, int64_t ws_bytes, void* acc) {
    MISEG_TAPE(miseg_bn_relu_bwd_dual_acc, stream, dt, raw, gy, gpool, gy2, n2_begin, n2_end, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes, acc);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_bwd_dual_acc, stream, MISEG_BF16, raw, gy, gpool, gy2, n2_begin, n2_end, N, H, W, C, gamma, saved, training, graw,
                          ggamma, gbeta, ws, ws_bytes, acc);
    MISEG_REQUIRE(graw && acc, "bn_relu_bwd_dual_acc: null pointer");
    return bn_relu_bwd_impl(stream, dt, raw, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes, nullptr, nullptr, gy2, n2_begin, n2_end, acc);
}

extern "C" int64_t miseg_bn_relu_bwd_acc_supported(int dt, int64_t N, int64_t H, int64_t W, int64_t C) {
    const int V = (dt == MISEG_F32) ? 4 : 8;
    if (C % V || C > kBnAccMaxC || 256 % (C / V)) return 0;
    return red_blocks(N * H * W, (int)(C / V)) <= kBnAccMaxBlocks;
}

// finalize + apply only: the per-block sums (sum dz, sum dz * xhat) were taken by the epilogue of the data-gradient convolution that
// wrote gy (miseg_conv3x3_dgrad_bn with red_*): the reduce pass over (raw, gy) is not run.  ws: 3 C floats.
extern "C" int miseg_bn_relu_bwd_ext(void* stream, int dt, const void* raw, const void* gy, int64_t N, int64_t H, int64_t W, int64_t C,
                                     const float* gamma, const float* saved, int training, void* graw, float* ggamma, float* gbeta,
                                     const float* ext_parts, int64_t ext_nparts, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_bn_relu_bwd_ext, stream, dt, raw, gy, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ext_parts, ext_nparts, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_bwd_ext, stream, MISEG_BF16, raw, gy, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ext_parts,
                          ext_nparts, ws, ws_bytes);
    MISEG_REQUIRE(raw && gy && gamma && saved && graw && ggamma && gbeta && ext_parts && ext_nparts > 0 && ws && ws_bytes >= 3 * C * 4, "bn_relu_bwd_ext: bad args");
    const int V = dt == MISEG_BF16 ? 8 : 4;
    const int CV = (int)(C / V);
    MISEG_REQUIRE(dt != MISEG_F16 && C % V == 0 && 256 % CV == 0, "bn_relu_bwd_ext: C/%d must divide 256", V);
    hipStream_t st = as_stream(stream);
    const int64_t npix = N * H * W;
    float* coeffs = (float*)ws;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(256), 0, st, ext_parts, (int)ext_nparts, (int)C, (float)npix, gamma, saved, training,
                       coeffs, ggamma, gbeta, (float*)nullptr);
    MISEG_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    const int na = ew_blocks(npix * CV);
    if (dt == MISEG_F32)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), dim3(na), dim3(256), 0, st, (const float*)raw, (const float*)gy, (const float*)nullptr, (int)N,
                           (int)H, (int)W, (int)C, saved, coeffs, (float*)graw, BnGy2<float>{nullptr, 0, 0});
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, false>), dim3(na), dim3(256), 0, st, (const bf16*)raw, (const bf16*)gy, (const bf16*)nullptr, (int)N,
                           (int)H, (int)W, (int)C, saved, coeffs, (bf16*)graw, BnGy2<bf16>{nullptr, 0, 0});
    MISEG_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return MISEG_OK;
}

extern "C" int miseg_bn_relu_bwd_stats(void* stream, int dt, const void* raw, const void* gy, int64_t N, int64_t H, int64_t W, int64_t C,
                                       const float* gamma, const float* saved, int training, float* bwd_coef, float* ggamma, float* gbeta,
                                       const float* ext_parts, int64_t ext_nparts, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_bn_relu_bwd_stats, stream, dt, raw, gy, N, H, W, C, gamma, saved, training, bwd_coef, ggamma, gbeta, ext_parts, ext_nparts, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_bn_relu_bwd_stats, stream, MISEG_BF16, raw, gy, N, H, W, C, gamma, saved, training, bwd_coef, ggamma, gbeta, ext_parts,
                          ext_nparts, ws, ws_bytes);
    MISEG_REQUIRE(bwd_coef, "bn_relu_bwd_stats: null pointer");
    if (ext_parts) {      // the sums were taken by the epilogue of the convolution that wrote gy (miseg_conv3x3_dgrad_bn)
        MISEG_REQUIRE(gamma && saved && ggamma && gbeta && ext_nparts > 0 && C > 0, "bn_relu_bwd_stats: bad args");
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(256), 0, as_stream(stream), ext_parts, (int)ext_nparts, (int)C,
                           (float)(N * H * W), gamma, saved, training, (float*)nullptr, ggamma, gbeta, bwd_coef);
        MISEG_LAUNCH_CHECK("bn_bwd_finalize_kernel");
        return MISEG_OK;
    }
    return bn_relu_bwd_impl(stream, dt, raw, gy, nullptr, N, H, W, C, gamma, saved, training, nullptr, ggamma, gbeta, ws, ws_bytes, nullptr, bwd_coef);
}

extern "C" int miseg_bn_relu_bwd(void* stream, int dt, const void* raw, const void* y, const void* gy, const void* gpool, int64_t N,
                                 int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved, int training, void* graw,
                                 float* ggamma, float* gbeta, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_bn_relu_bwd, stream, dt, raw, y, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes);
    return miseg_bn_relu_bwd_sync(stream, dt, raw, y, gy, gpool, N, H, W, C, gamma, saved, training, graw, ggamma, gbeta, ws, ws_bytes, nullptr);
}

extern "C" int miseg_sumpool2x2(void* stream, int dt, const void* in, int64_t N, int64_t H, int64_t W, int64_t C, void* out, int accumulate) {
    MISEG_TAPE(miseg_sumpool2x2, stream, dt, in, N, H, W, C, out, accumulate);
    MISEG_F16_DISPATCH_ON(dt, miseg_sumpool2x2, stream, MISEG_BF16, in, N, H, W, C, out, accumulate);
    MISEG_REQUIRE(in && out && H % 2 == 0 && W % 2 == 0, "sumpool2x2: bad args");
    const int V = dt == MISEG_BF16 ? 8 : 4;
    MISEG_REQUIRE(C % V == 0, "sumpool2x2: C must be a multiple of %d", V);
    const int64_t total = N * (H / 2) * (W / 2) * (C / V);
    hipStream_t st = as_stream(stream);
    if (dt == MISEG_F32) hipLaunchKernelGGL(sumpool2x2_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, st, (const float*)in, (int)N, (int)H, (int)W, (int)C, (float*)out, accumulate);
    else if (dt == MISEG_BF16) hipLaunchKernelGGL(sumpool2x2_kernel<bf16>, dim3(ew_blocks(total)), dim3(256), 0, st, (const bf16*)in, (int)N, (int)H, (int)W, (int)C, (bf16*)out, accumulate);
    else return fail(MISEG_E_INVALID, "sumpool2x2: bad dtype");
    MISEG_LAUNCH_CHECK("sumpool2x2_kernel");
    return MISEG_OK;
}

extern "C" int miseg_axpy(void* stream, int dt, const void* src, void* dst, int64_t numel) {
    MISEG_TAPE(miseg_axpy, stream, dt, src, dst, numel);
    MISEG_F16_DISPATCH_ON(dt, miseg_axpy, stream, MISEG_BF16, src, dst, numel);
    MISEG_REQUIRE(src && dst, "axpy: null pointer");
    const int V = dt == MISEG_BF16 ? 8 : 4;
    MISEG_REQUIRE(numel % V == 0, "axpy: numel must be a multiple of %d", V);
    hipStream_t st = as_stream(stream);
    if (dt == MISEG_F32) hipLaunchKernelGGL(axpy_kernel<float>, dim3(ew_blocks(numel / V)), dim3(256), 0, st, (const float*)src, (float*)dst, numel / V);
    else if (dt == MISEG_BF16) hipLaunchKernelGGL(axpy_kernel<bf16>, dim3(ew_blocks(numel / V)), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, numel / V);
    else return fail(MISEG_E_INVALID, "axpy: bad dtype");
    MISEG_LAUNCH_CHECK("axpy_kernel");
    return MISEG_OK;
}

extern "C" int miseg_cast_pad(void* stream, const float* in, int64_t npix, int64_t Cin, int dt_out, void* out, int64_t CP) {
    MISEG_TAPE(miseg_cast_pad, stream, in, npix, Cin, dt_out, out, CP);
    MISEG_F16_DISPATCH_ON(dt_out, miseg_cast_pad, stream, in, npix, Cin, MISEG_BF16, out, CP);
    MISEG_REQUIRE(in && out && Cin > 0 && CP >= Cin, "cast_pad: bad args");
    hipStream_t st = as_stream(stream);
    const int nb = ew_blocks(npix * CP);
    if (dt_out == MISEG_F32) hipLaunchKernelGGL((cast_pad_kernel<float, float>), dim3(nb), dim3(256), 0, st, in, (float*)out, npix, (int)Cin, (int)CP);
    else if (dt_out == MISEG_BF16 && Cin == 1 && CP == 8) hipLaunchKernelGGL(cast_pad_c1_kernel<bf16>, dim3(ew_blocks(npix)), dim3(256), 0, st, in, (bf16*)out, npix);
    else if (dt_out == MISEG_BF16) hipLaunchKernelGGL((cast_pad_kernel<float, bf16>), dim3(nb), dim3(256), 0, st, in, (bf16*)out, npix, (int)Cin, (int)CP);
    else return fail(MISEG_E_INVALID, "cast_pad: bad dtype");
    MISEG_LAUNCH_CHECK("cast_pad_kernel");
    return MISEG_OK;
}

extern "C" int miseg_adam_step_guarded(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                                       float beta1, float beta2, const float* hyper, float grad_scale, const float* guard, int64_t nguard) {
    MISEG_TAPE(miseg_adam_step_guarded, stream, param, grad, exp_avg, exp_avg_sq, numel, beta1, beta2, hyper, grad_scale, guard, nguard);
    MISEG_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper && numel > 0, "adam_step: bad args");
    MISEG_REQUIRE((grad_scale > 0.f || grad_scale == -1.f) && std::isfinite(grad_scale), "adam_step: grad_scale must be a positive finite number (or -1: read 1 / scale from hyper[4])");
    MISEG_REQUIRE(nguard >= 0 && nguard <= 1024 && (nguard == 0 || guard), "adam_step: bad guard");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(numel)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, numel, beta1, beta2,
                       hyper, grad_scale > 0.f ? 1.f / grad_scale : -1.f, guard, (int)nguard);
    MISEG_LAUNCH_CHECK("adam_kernel");
    return MISEG_OK;
}
namespace miseg {
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ count) {
    int bad = 0;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad += !(fabsf(g[i]) <= 3.4028234e38f);   // inf or NaN
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, (float)bad);       // integers below 2^24: exact in any order
}
}  // namespace miseg

extern "C" int miseg_count_nonfinite(void* stream, const float* grad, int64_t numel, float* count) {
    MISEG_TAPE(miseg_count_nonfinite, stream, grad, numel, count);
    MISEG_REQUIRE(grad && count && numel > 0, "count_nonfinite: bad args");
    hipMemsetAsync(count, 0, sizeof(float), as_stream(stream));
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)std::min<int64_t>(cdiv(numel, 1024), 1024)), dim3(256), 0, as_stream(stream), grad, numel, count);
    MISEG_LAUNCH_CHECK("count_nonfinite_kernel");
    return MISEG_OK;
}

extern "C" int miseg_adam_step_scaled(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                                      float beta1, float beta2, const float* hyper, float grad_scale) {
    MISEG_TAPE(miseg_adam_step_scaled, stream, param, grad, exp_avg, exp_avg_sq, numel, beta1, beta2, hyper, grad_scale);
    return miseg_adam_step_guarded(stream, param, grad, exp_avg, exp_avg_sq, numel, beta1, beta2, hyper, grad_scale, nullptr, 0);
}
extern "C" int miseg_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                               float beta1, float beta2, const float* hyper) {
    MISEG_TAPE(miseg_adam_step, stream, param, grad, exp_avg, exp_avg_sq, numel, beta1, beta2, hyper);
    return miseg_adam_step_scaled(stream, param, grad, exp_avg, exp_avg_sq, numel, beta1, beta2, hyper, 1.f);
}
