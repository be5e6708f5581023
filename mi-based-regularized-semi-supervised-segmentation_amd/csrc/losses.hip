// Pixel-wise losses on fp32 NHWC logits, the per-sample flip, argmax + Dice counts.
// All HBM-bound streaming kernels: one pixel per thread, channel vector in registers, two-pass
// deterministic sums (per-block partial -> single-block finish).
#include <hip/hip_fp16.h>

#include "common.h"

namespace miseg {

constexpr int kLT = 256;
constexpr int kMaxC = 32;

__device__ __forceinline__ void softmax_c(const float* __restrict__ z, int C, float* p) {
    float mx = -3.4e38f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) { p[c] = expf(z[c] - mx); s += p[c]; }
    for (int c = 0; c < C; ++c) p[c] = p[c] / s;
}

// KL_div(softmax(logits), onehot(labels)), mean over pixels (whl:loss/kl_losses.py:113-126)
template <int C>
__global__ __launch_bounds__(kLT) void softmax_kl_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         int64_t npix, const float* __restrict__ upstream,
                                                         float* __restrict__ partials, float* __restrict__ glogits,
                                                         int32_t* __restrict__ bad) {
    __shared__ float red[17];
    const float eps = 1e-16f;
    const float up = (upstream ? upstream[0] : 1.f) / (float)npix;
    float part = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)kLT + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kLT) {
        float z[C], p[C];
#pragma unroll
        for (int c = 0; c < C; ++c) z[c] = logits[i * C + c];
        softmax_c(z, C, p);
        const int64_t t = labels[i];
        if (t < 0 || t >= C) { *bad = 1; continue; }
        float kl = 0.f, pt = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float tc = (c == t) ? 1.f : 0.f;
            kl += -tc * logf((p[c] + eps) / (tc + eps));
            if (c == t) pt = p[c];
        }
        part += kl;
        if (glogits) {
            // d/dp_t = -1/(p_t+eps); through the softmax Jacobian: dz_c = p_c*(g_c - sum_k g_k p_k)
            const float gt = -1.f / (pt + eps) * up;
#pragma unroll
            for (int c = 0; c < C; ++c) glogits[i * C + c] = p[c] * (((c == t) ? gt : 0.f) - gt * pt);
        }
    }
    part = block_sum(part, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = part;
}

// mean((softmax(a) - softmax(flip(b)))^2) over N*C*H*W elements (semi_seg/epocher.py:221-224)
template <int C>
__global__ __launch_bounds__(kLT) void softmax_mse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const int32_t* __restrict__ flips, int H, int W, int64_t npix,
                                                          const float* __restrict__ upstream, float* __restrict__ partials,
                                                          float* __restrict__ ga) {
    __shared__ float red[17];
    const float up = (upstream ? upstream[0] : 1.f) * 2.f / ((float)npix * (float)C);
    const int64_t HW = (int64_t)H * W;
    float part = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)kLT + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kLT) {
        const int n = i / HW, rem = i % HW, h = rem / W, w = rem % W;
        const int f = flips ? flips[n] : 0;
        const int64_t j = (int64_t)n * HW + (int64_t)flip_h(h, H, f) * W + flip_w(w, W, f);
        float za[C], zb[C], pa[C], pb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { za[c] = a[i * C + c]; zb[c] = b[j * C + c]; }
        softmax_c(za, C, pa);
        softmax_c(zb, C, pb);
        float dot = 0.f, sq = 0.f;
        float d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { d[c] = pa[c] - pb[c]; sq += d[c] * d[c]; dot += d[c] * pa[c]; }
        part += sq;
        if (ga) {
#pragma unroll
            for (int c = 0; c < C; ++c) ga[i * C + c] = up * pa[c] * (d[c] - dot);
        }
    }
    part = block_sum(part, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = part;
}

// KL variant of the consistency term (ref semi_seg/trainer.py:137,194 `UDARegCriterion.name: kl`, semi_seg/epocher.py:221-224 with
// whl:deepclustering2/loss/kl_losses.py:107-126):  mean_pix sum_c -t_c log((p_c + eps) / (t_c + eps)),  p = softmax(a),
// t = softmax(flip(b)) detached, eps = 1e-16.  d/da_c = p_c (g_c - <g, p>) with g_c = -t_c / (p_c + eps).
template <int C>
__global__ __launch_bounds__(kLT) void softmax_klcons_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const int32_t* __restrict__ flips, int H, int W, int64_t npix,
                                                             const float* __restrict__ upstream, float* __restrict__ partials,
                                                             float* __restrict__ ga) {
    __shared__ float red[17];
    const float up = (upstream ? upstream[0] : 1.f) / (float)npix;
    const float eps = 1e-16f;
    const int64_t HW = (int64_t)H * W;
    float part = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)kLT + threadIdx.x; i < npix; i += (int64_t)gridDim.x * kLT) {
        const int n = i / HW, rem = i % HW, h = rem / W, w = rem % W;
        const int f = flips ? flips[n] : 0;
        const int64_t j = (int64_t)n * HW + (int64_t)flip_h(h, H, f) * W + flip_w(w, W, f);
        float za[C], zb[C], pa[C], pb[C], g[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { za[c] = a[i * C + c]; zb[c] = b[j * C + c]; }
        softmax_c(za, C, pa);
        softmax_c(zb, C, pb);
        float kl = 0.f, dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            kl += -pb[c] * logf((pa[c] + eps) / (pb[c] + eps));
            g[c] = -pb[c] / (pa[c] + eps);
            dot += g[c] * pa[c];
        }
        part += kl;
        if (ga) {
#pragma unroll
            for (int c = 0; c < C; ++c) ga[i * C + c] = up * pa[c] * (g[c] - dot);
        }
    }
    part = block_sum(part, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = part;
}

__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out) {
    __shared__ float red[17];
    // independent loads first (a dependent load-add chain over 8 partials per thread was ~50 us of pure latency), fixed order
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = threadIdx.x + 256 * j; v[j] = i < n ? partials[i] : 0.f; }
    float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    for (int i = threadIdx.x + 2048; i < n; i += 256) s += partials[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

template <typename E>
__global__ void flip_kernel(const E* __restrict__ in, E* __restrict__ out, int N, int C, int H, int W, int64_t is0, int64_t is1,
                            int64_t is2, int64_t is3, int64_t os0, int64_t os1, int64_t os2, int64_t os3,
                            const int32_t* __restrict__ flips) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int w = e % W, h = (e / W) % H, c = (e / ((int64_t)W * H)) % C, n = e / ((int64_t)W * H * C);
        const int f = flips[n];
        out[n * os0 + c * os1 + h * os2 + w * os3] = in[n * is0 + c * is1 + flip_h(h, H, f) * is2 + flip_w(w, W, f) * is3];
    }
}

// argmax over channels (first maximum, as torch.max(1)[1]) + per-sample per-class intersection / union
template <int C>
__global__ __launch_bounds__(256) void argmax_dice_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int HW, int64_t* __restrict__ pred, unsigned long long* __restrict__ inter,
                                                          unsigned long long* __restrict__ uni) {
    __shared__ unsigned int si[C], su[C];
    const int n = blockIdx.y;
    if (threadIdx.x < C) { si[threadIdx.x] = 0; su[threadIdx.x] = 0; }
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float* z = logits + ((size_t)n * HW + i) * C;
        int best = 0;
        float bv = z[0];
#pragma unroll
        for (int c = 1; c < C; ++c)
            if (z[c] > bv) { bv = z[c]; best = c; }
        if (pred) pred[(size_t)n * HW + i] = best;
        if (labels) {
            const int t = (int)labels[(size_t)n * HW + i];
            atomicAdd(&su[best], 1u);
            if (t >= 0 && t < C) {
                atomicAdd(&su[t], 1u);
                if (t == best) atomicAdd(&si[best], 1u);
            }
        }
    }
    __syncthreads();
    if (labels && threadIdx.x < C) {
        atomicAdd(&inter[n * C + threadIdx.x], (unsigned long long)si[threadIdx.x]);
        atomicAdd(&uni[n * C + threadIdx.x], (unsigned long long)su[threadIdx.x]);
    }
}

static int loss_blocks(int64_t npix) { return (int)std::min<int64_t>(cdiv(npix, kLT), 2048); }

// one thread per (outer, inner) position, channel stride = inner: coalesced along inner
__global__ __launch_bounds__(256) void simplex_violations_kernel(const float* __restrict__ x, int64_t outer, int C, int64_t inner,
                                                                 float tol, int32_t* __restrict__ count) {
    const int64_t total = outer * inner;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t o = i / inner, r = i - o * inner;
        const float* p = x + o * C * inner + r;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += p[(int64_t)c * inner];
        bad += !(fabsf(s - 1.f) <= tol);
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}


// The iteration's host report in ONE launch (was ~20 one-element torch kernels at the head of every step: isfinite / where / two
// mat-vecs / isnan-sum-cast per check / casts / cats).  flat = the distinct device scalars (and the vectors that are NaN-tested), one
// torch.cat; row r of coeff = the linear combination a reported value is (miseg_amd.lazy.LinearLoss).  out[0 .. R) = the values --
// a non-finite entry poisons exactly the rows that use it (0 * NaN must not reach the others) --, out[R + j] = check flag j:
// desc[j] = (kind, a, b): 0 = flat[a] as is, 1 = number of NaNs in flat[a .. a + b), 2 = (float) iflat[a].
__global__ __launch_bounds__(256) void report_scalars_kernel(const float* __restrict__ flat, const int32_t* __restrict__ iflat,
                                                             const float* __restrict__ coeff, int R, int C,
                                                             const int32_t* __restrict__ desc, int ncheck, float* __restrict__ out) {
    for (int r = threadIdx.x; r < R; r += 256) {
        float acc = 0.f;
        bool bad = false;
        for (int c = 0; c < C; ++c) {
            const float k = coeff[(size_t)r * C + c], v = flat[c];
            if (k != 0.f) {
                if (isfinite(v)) acc += k * v;
                else bad = true;
            }
        }
        out[r] = bad ? __int_as_float(0x7FC00000) : acc;
    }
    for (int j = threadIdx.x; j < ncheck; j += 256) {
        const int kind = desc[3 * j], a = desc[3 * j + 1], b = desc[3 * j + 2];
        float f;
        if (kind == 0) f = flat[a];
        else if (kind == 2) f = (float)iflat[a];
        else {
            int n = 0;
            for (int e = 0; e < b; ++e) n += isnan(flat[a + e]) ? 1 : 0;
            f = (float)n;
        }
        out[R + j] = f;
    }
}

}  // namespace miseg

using namespace miseg;

extern "C" int miseg_report_scalars(void* stream, const float* flat, const int32_t* iflat, const float* coeff, int64_t R, int64_t C,
                                    const int32_t* desc, int64_t ncheck, float* out) {
    MISEG_TAPE(miseg_report_scalars, stream, flat, iflat, coeff, R, C, desc, ncheck, out);
    MISEG_REQUIRE(out && R >= 0 && C >= 0 && ncheck >= 0 && R + ncheck > 0, "report_scalars: bad sizes");
    MISEG_REQUIRE((R == 0 || (coeff && (C == 0 || flat))) && (ncheck == 0 || desc), "report_scalars: null pointer");
    hipLaunchKernelGGL(report_scalars_kernel, dim3(1), dim3(256), 0, as_stream(stream), flat, iflat, coeff, (int)R, (int)C, desc, (int)ncheck, out);
    MISEG_LAUNCH_CHECK("report_scalars_kernel");
    return MISEG_OK;
}

extern "C" int miseg_simplex_violations(void* stream, const float* x, int64_t outer, int64_t C, int64_t inner, float tol,
                                        int32_t* count) {
    MISEG_TAPE(miseg_simplex_violations, stream, x, outer, C, inner, tol, count);
    MISEG_REQUIRE(x && count, "simplex_violations: null pointer");
    MISEG_REQUIRE(outer > 0 && C > 0 && inner > 0 && C < (1 << 20), "simplex_violations: bad shape");
    const int64_t total = outer * inner;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 4096);
    hipLaunchKernelGGL(simplex_violations_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, outer, (int)C, inner, tol, count);
    MISEG_LAUNCH_CHECK("simplex_violations_kernel");
    return MISEG_OK;
}

extern "C" int64_t miseg_loss_ws_bytes(int64_t N, int64_t H, int64_t W) { return (int64_t)loss_blocks(N * H * W) * 4; }

#define MISEG_DISPATCH_C(C, MACRO)                                                            \
    switch (C) {                                                                              \
        case 2: MACRO(2); break;                                                              \
        case 3: MACRO(3); break;                                                              \
        case 4: MACRO(4); break;                                                              \
        case 5: MACRO(5); break;                                                              \
        case 6: MACRO(6); break;                                                              \
        case 8: MACRO(8); break;                                                              \
        case 10: MACRO(10); break;                                                            \
        case 16: MACRO(16); break;                                                            \
        default: return fail(MISEG_E_INVALID, "unsupported class count %ld (2-6,8,10,16)", (long)C); \
    }

extern "C" int miseg_softmax_kl(void* stream, const float* logits, const int64_t* labels, int64_t N, int64_t H, int64_t W, int64_t C,
                                const float* upstream, float* loss, float* glogits, int32_t* bad_label, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_softmax_kl, stream, logits, labels, N, H, W, C, upstream, loss, glogits, bad_label, ws, ws_bytes);
    MISEG_REQUIRE(logits && labels && loss && bad_label && ws, "softmax_kl: null pointer");
    const int64_t npix = N * H * W;
    MISEG_REQUIRE(npix > 0 && ws_bytes >= miseg_loss_ws_bytes(N, H, W), "softmax_kl: bad shape / workspace");
    const int nb = loss_blocks(npix);
    hipStream_t st = as_stream(stream);
#define L(CC) hipLaunchKernelGGL(softmax_kl_kernel<CC>, dim3(nb), dim3(kLT), 0, st, logits, labels, npix, upstream, (float*)ws, glogits, bad_label)
    MISEG_DISPATCH_C(C, L)
#undef L
    MISEG_LAUNCH_CHECK("softmax_kl_kernel");
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nb, 1.0f / (float)npix, loss);
    MISEG_LAUNCH_CHECK("finish_sum_kernel");
    return MISEG_OK;
}

extern "C" int miseg_softmax_mse(void* stream, const float* a, const float* b, const int32_t* flips, int64_t N, int64_t H, int64_t W,
                                 int64_t C, const float* upstream, float* loss, float* ga, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_softmax_mse, stream, a, b, flips, N, H, W, C, upstream, loss, ga, ws, ws_bytes);
    MISEG_REQUIRE(a && b && loss && ws, "softmax_mse: null pointer");
    const int64_t npix = N * H * W;
    MISEG_REQUIRE(npix > 0 && ws_bytes >= miseg_loss_ws_bytes(N, H, W), "softmax_mse: bad shape / workspace");
    const int nb = loss_blocks(npix);
    hipStream_t st = as_stream(stream);
#define L(CC) hipLaunchKernelGGL(softmax_mse_kernel<CC>, dim3(nb), dim3(kLT), 0, st, a, b, flips, (int)H, (int)W, npix, upstream, (float*)ws, ga)
    MISEG_DISPATCH_C(C, L)
#undef L
    MISEG_LAUNCH_CHECK("softmax_mse_kernel");
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nb, 1.0f / ((float)npix * (float)C), loss);
    MISEG_LAUNCH_CHECK("finish_sum_kernel");
    return MISEG_OK;
}

extern "C" int miseg_softmax_klcons(void* stream, const float* a, const float* b, const int32_t* flips, int64_t N, int64_t H, int64_t W,
                                    int64_t C, const float* upstream, float* loss, float* ga, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_softmax_klcons, stream, a, b, flips, N, H, W, C, upstream, loss, ga, ws, ws_bytes);
    MISEG_REQUIRE(a && b && loss && ws, "softmax_klcons: null pointer");
    const int64_t npix = N * H * W;
    MISEG_REQUIRE(npix > 0 && ws_bytes >= miseg_loss_ws_bytes(N, H, W), "softmax_klcons: bad shape / workspace");
    const int nb = loss_blocks(npix);
    hipStream_t st = as_stream(stream);
#define L(CC) hipLaunchKernelGGL(softmax_klcons_kernel<CC>, dim3(nb), dim3(kLT), 0, st, a, b, flips, (int)H, (int)W, npix, upstream, (float*)ws, ga)
    MISEG_DISPATCH_C(C, L)
#undef L
    MISEG_LAUNCH_CHECK("softmax_klcons_kernel");
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nb, 1.0f / (float)npix, loss);
    MISEG_LAUNCH_CHECK("finish_sum_kernel");
    return MISEG_OK;
}

extern "C" int miseg_flip(void* stream, const void* in, void* out, int64_t N, int64_t C, int64_t H, int64_t W,
                          const int64_t* is, const int64_t* os, int elem_bytes, const int32_t* flips) {
    // the stride arrays are HOST memory read during this call: a recorded call keeps its own copy of them
    ::miseg_core::TapeScope miseg_tape_scope_;
#ifndef MISEG_F16_BUILD
    if (::miseg_core::tape_depth == 1 && ::miseg_core::tape_recording() && is && os) {
        auto* op = new ::miseg_core::TapeFnOp();
        struct Call { void* stream; const void* in; void* out; int64_t n, c, h, w, is[4], os[4]; int eb; const int32_t* flips; };
        auto call = std::make_shared<Call>(Call{stream, in, out, N, C, H, W, {is[0], is[1], is[2], is[3]}, {os[0], os[1], os[2], os[3]}, elem_bytes, flips});
        op->name = "miseg_flip";
        op->stream = stream;
        op->fn = [call]() { return miseg_flip(call->stream, call->in, call->out, call->n, call->c, call->h, call->w, call->is, call->os, call->eb, call->flips); };
        op->ptrs = {&call->in, const_cast<const void**>(reinterpret_cast<void**>(&call->out)), reinterpret_cast<const void**>(&call->flips)};
        ::miseg_core::tape_push(op);
    }
#endif
    MISEG_REQUIRE(in && out && is && os && flips, "flip: null pointer");
    MISEG_REQUIRE(in != out, "flip: in-place not supported");
    const int64_t total = N * C * H * W;
    if (total == 0) return MISEG_OK;
    const int nb = (int)std::min<int64_t>(cdiv(total, 256), 4096);
    hipStream_t st = as_stream(stream);
#define F(E) hipLaunchKernelGGL(flip_kernel<E>, dim3(nb), dim3(256), 0, st, (const E*)in, (E*)out, (int)N, (int)C, (int)H, (int)W, is[0], is[1], is[2], is[3], os[0], os[1], os[2], os[3], flips)
    if (elem_bytes == 2) F(uint16_t);
    else if (elem_bytes == 4) F(uint32_t);
    else if (elem_bytes == 8) F(uint64_t);
    else return fail(MISEG_E_INVALID, "flip: elem_bytes must be 2, 4 or 8");
#undef F
    MISEG_LAUNCH_CHECK("flip_kernel");
    return MISEG_OK;
}

// out = [a | b | flip(b)] along the batch dimension (contiguous [*, C, H, W], 4-byte elements): the network's input batch
// (ref semi_seg/epocher.py:148-153: stack of per-sample flips, then torch.cat) in one pass.
namespace miseg {
__global__ __launch_bounds__(256) void cat_flip_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, int Na,
                                                      int Nb, int C, int H, int W, const int32_t* __restrict__ flips) {
    const int64_t per = (int64_t)C * H * W, total = (int64_t)(Na + 2 * Nb) * per;
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e / per, r = e - n * per;
        uint32_t v;
        if (n < Na) v = a[e];
        else if (n < Na + Nb) v = b[e - (int64_t)Na * per];
        else {
            const int m = (int)(n - Na - Nb), f = flips[m];
            const int w = r % W, h = (r / W) % H, c = r / ((int64_t)W * H);
            const int hs = (f & 1) ? H - 1 - h : h, wsrc = (f & 2) ? W - 1 - w : w;
            v = b[(int64_t)m * per + ((int64_t)c * H + hs) * W + wsrc];
        }
        out[e] = v;
    }
}
}  // namespace miseg
// ... and, for one-channel images, the stem's operand in the same pass: pad[n][h][w][0] = 16-bit value, channels 1..7 zero (what
// miseg_cast_pad makes of `out`), so the network's first convolution starts from this kernel's output
namespace miseg {
template <bool HALF>
__global__ __launch_bounds__(256) void cat_flip_pad_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                          uint4* __restrict__ pad, int Na, int Nb, int H, int W, const int32_t* __restrict__ flips) {
    const int64_t per = (int64_t)H * W, total = (int64_t)(Na + 2 * Nb) * per;
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e / per, r = e - n * per;
        float v;
        if (n < Na) v = a[e];
        else if (n < Na + Nb) v = b[e - (int64_t)Na * per];
        else {
            const int m = (int)(n - Na - Nb), f = flips[m];
            const int w = r % W, h = r / W;
            const int hs = (f & 1) ? H - 1 - h : h, wsrc = (f & 2) ? W - 1 - w : w;
            v = b[(int64_t)m * per + (int64_t)hs * W + wsrc];
        }
        out[e] = v;
        unsigned bits;
        if (HALF) bits = (unsigned)__half_as_ushort(__float2half(v));
        else { const __hip_bfloat16 hb = __float2bfloat16(v); bits = (unsigned)*reinterpret_cast<const unsigned short*>(&hb); }
        pad[e] = make_uint4(bits, 0u, 0u, 0u);
    }
}
}  // namespace miseg
extern "C" int miseg_cat_flip_pad(void* stream, const float* a, int64_t Na, const float* b, int64_t Nb, int64_t H, int64_t W, const int32_t* flips,
                                  float* out, int dt_pad, void* pad8) {
    MISEG_TAPE(miseg_cat_flip_pad, stream, a, Na, b, Nb, H, W, flips, out, dt_pad, pad8);
    MISEG_REQUIRE(out && pad8 && flips && (a || Na == 0) && b && Na >= 0 && Nb > 0 && H > 0 && W > 0, "cat_flip_pad: bad args");
    MISEG_REQUIRE(dt_pad == MISEG_BF16 || dt_pad == MISEG_F16, "cat_flip_pad: the padded copy is a 16-bit type");
    const int64_t total = (Na + 2 * Nb) * H * W;
    const dim3 grid((unsigned)std::min<int64_t>(cdiv(total, 256), 8192));
    if (dt_pad == MISEG_F16)
        hipLaunchKernelGGL(cat_flip_pad_kernel<true>, grid, dim3(256), 0, as_stream(stream), a, b, out, (uint4*)pad8, (int)Na, (int)Nb, (int)H, (int)W, flips);
    else
        hipLaunchKernelGGL(cat_flip_pad_kernel<false>, grid, dim3(256), 0, as_stream(stream), a, b, out, (uint4*)pad8, (int)Na, (int)Nb, (int)H, (int)W, flips);
    MISEG_LAUNCH_CHECK("cat_flip_pad_kernel");
    return MISEG_OK;
}

extern "C" int miseg_cat_flip(void* stream, const void* a, int64_t Na, const void* b, int64_t Nb, int64_t C, int64_t H, int64_t W,
                              const int32_t* flips, void* out) {
    MISEG_TAPE(miseg_cat_flip, stream, a, Na, b, Nb, C, H, W, flips, out);
    MISEG_REQUIRE(out && flips && (a || Na == 0) && b && Na >= 0 && Nb > 0 && C > 0 && H > 0 && W > 0, "cat_flip: bad args");
    const int64_t total = (Na + 2 * Nb) * C * H * W;
    hipLaunchKernelGGL(cat_flip_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 8192)), dim3(256), 0, as_stream(stream), (const uint32_t*)a,
                       (const uint32_t*)b, (uint32_t*)out, (int)Na, (int)Nb, (int)C, (int)H, (int)W, flips);
    MISEG_LAUNCH_CHECK("cat_flip_kernel");
    return MISEG_OK;
}

extern "C" int miseg_argmax_dice(void* stream, const float* logits, const int64_t* labels, int64_t N, int64_t H, int64_t W, int64_t C,
                                 int64_t* pred, int64_t* inter, int64_t* uni) {
    MISEG_TAPE(miseg_argmax_dice, stream, logits, labels, N, H, W, C, pred, inter, uni);
    MISEG_REQUIRE(logits && (pred || labels), "argmax_dice: null pointer");
    MISEG_REQUIRE(!labels || (inter && uni), "argmax_dice: labels need inter/uni outputs");
    hipStream_t st = as_stream(stream);
    if (labels) {
        if (uni == inter + N * C) hipMemsetAsync(inter, 0, (size_t)N * C * 16, st);      // adjacent (the iteration's read-back block): one fill
        else {
            hipMemsetAsync(inter, 0, (size_t)N * C * 8, st);
            hipMemsetAsync(uni, 0, (size_t)N * C * 8, st);
        }
    }
    dim3 grid((unsigned)std::min<int64_t>(cdiv(H * W, 256), 64), (unsigned)N);
#define L(CC) hipLaunchKernelGGL(argmax_dice_kernel<CC>, grid, dim3(256), 0, st, logits, labels, (int)(H * W), pred, (unsigned long long*)inter, (unsigned long long*)uni)
    MISEG_DISPATCH_C(C, L)
#undef L
    MISEG_LAUNCH_CHECK("argmax_dice_kernel");
    return MISEG_OK;
}
