// Local (displacement-window) IIC mutual information on gfx950.
//
// Replaces the reference's IIDSegmentationLoss (contrastyou/losses/iic_loss.py:107-149):
//   permute+contiguous x2, F.conv2d(x[K,N,H,W], weight=y[K,N,H,W], padding=p)  -> joint_fwd
//   min-shift / normalise / symmetrise / MI                                       -> loss_fwd
//   autograd through all of it                                                    -> bwd
//
// Formulation (DESIGN.md "local MI"): for one image row r of X the displacement joint is a GEMM
//   D[(dx,i),(dy,j)] += sum_w X[i][r][w+dx] * Y[j][r-dy][w]
// with M = N = T*K (T = 2*pad+1; 140 for K=20,pad=3) and the reduction running over pixels, so
// both displacement axes are stacked into the MFMA tile dims (padding waste (140/144)^2, not the
// (20/32)^2 of a per-displacement 20x20 tile) and no im2col copy is ever materialised: operand
// fragments are gathered straight from haloed LDS tiles of X and Y.
// This file is the exact-fp32 path: v_mfma_f32_16x16x4_f32 (k-ordered fp32 fma chain).
#include "mi_local.h"

namespace miseg {

// D[(dx,i),(dy,j)] accumulators: MT x NT tiles of 16x16 (4 fp32 per lane each).  A full 9x9 D is 324
// registers per lane -- more than the 256-entry accumulator file -- so the D tile set is split between
// the two waves of a PAIR (role 0 / role 1, 41 + 40 tiles for 9x9) that sweep the same image rows;
// a block is 4 pairs = 8 waves (2 per SIMD) sharing one staged X/Y tile.
template <int MT, int NT, int ROLE>
__device__ __forceinline__ void joint_fwd_body(const float* __restrict__ x, const float* __restrict__ y,
                                               const float* __restrict__ mask, const JointGeom& g,
                                               const int32_t* __restrict__ win, float* __restrict__ partials, float* lds) {
    typedef TileSet<MT, NT, ROLE> TS;
    float* Xs = lds;
    float* Ys = lds + (size_t)g.K * g.planeX;
    const int tid = threadIdx.x, lane = tid & 63, pair = tid >> 7;
    const int l15 = lane & 15, kq = lane >> 4;
    const int slot = blockIdx.y;                 // (p*sb + sm)*sb + sn
    const int sn = slot % g.sb, sm = (slot / g.sb) % g.sb, p = slot / (g.sb * g.sb);
    const int mtu = min(g.tps, g.tilesM - sm * g.tps), ntu = min(g.tps, g.tilesM - sn * g.tps);
    const int h0 = win[p * 4 + 0], h1 = win[p * 4 + 1], w0 = win[p * 4 + 2], w1 = win[p * 4 + 3];
    const int tr = (h1 - h0 + g.RB - 1) / g.RB, tc = (w1 - w0 + g.WB - 1) / g.WB;
    const int nItems = g.N * tr * tc;

    // per-lane gather bases: row m=(dxi,i) of A reads X plane i shifted by dxi columns;
    // column c=(dyi,j) of B reads Y plane j, tile row (rx + 2*pad - dyi).
    int aoff[MT], boff[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = min((sm * g.tps + mt) * 16 + l15, g.Mdim - 1);
        aoff[mt] = (m % g.K) * g.planeX + (m / g.K) + kq;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int c = min((sn * g.tps + nt) * 16 + l15, g.Mdim - 1);
        boff[nt] = (c % g.K) * g.planeY + (2 * g.pad - c / g.K) * g.WB + kq;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t plane = (size_t)g.H * g.W;
    for (int it = blockIdx.x; it < nItems; it += g.G) {
        const int ct = it % tc, rt = (it / tc) % tr, n = it / (tc * tr);
        const int row0 = h0 + rt * g.RB, col0 = w0 + ct * g.WB;
        __syncthreads();  // previous tile fully consumed
        // ---- stage X tile: rows [row0,row0+RB), cols [col0-pad, col0+WB+pad), zero outside the window
        const int nx = g.K * g.RB * g.WX;
        for (int idx = tid; idx < nx; idx += kJT) {
            int cx = idx % g.WX, r = (idx / g.WX) % g.RB, ch = idx / (g.WX * g.RB);
            int row = row0 + r, col = col0 - g.pad + cx;
            float v = 0.f;
            if (row < h1 && col >= w0 && col < w1) {
                size_t o = (size_t)row * g.W + col;
                v = x[((size_t)n * g.K + ch) * plane + o];
                if (mask) v *= mask[(size_t)n * plane + o];
            }
            Xs[ch * g.planeX + r * g.WX + cx] = v;
        }
        // ---- stage Y tile: rows [row0-pad,row0+RB+pad), cols [col0,col0+WB)
        const int ny = g.K * g.RBY * g.WB;
        for (int idx = tid; idx < ny; idx += kJT) {
            int cy = idx % g.WB, r = (idx / g.WB) % g.RBY, ch = idx / (g.WB * g.RBY);
            int row = row0 - g.pad + r, col = col0 + cy;
            float v = 0.f;
            if (row >= h0 && row < h1 && col < w1) {
                size_t o = (size_t)row * g.W + col;
                v = y[((size_t)n * g.K + ch) * plane + o];
                if (mask) v *= mask[(size_t)n * plane + o];
            }
            Ys[ch * g.planeY + r * g.WB + cy] = v;
        }
        __syncthreads();
        // ---- MFMA: each pair sweeps its RW rows; 4 pixels of reduction per k-step
        for (int rr = 0; rr < g.RW; ++rr) {
            const int rx = pair * g.RW + rr;
            const float* xa = Xs + rx * g.WX;
            const float* yb = Ys + rx * g.WB;
#pragma unroll 2
            for (int s = 0; s < g.WB; s += 4) {
                float a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (TS::row_used(mt)) a[mt] = xa[aoff[mt] + s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (TS::col_used(nt)) b[nt] = yb[boff[nt] + s];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (TS::row_used(mt) && mt < mtu) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            if (TS::mine(mt, nt) && nt < ntu)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- reduce the 4 pairs' accumulators through LDS in fixed order, then one partial per block
    const int Dn = NT * 16;
    float* Ds = lds;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (pair == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (TS::mine(mt, nt)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int idx = (mt * 16 + kq * 4 + r) * Dn + nt * 16 + l15;
                            if (w == 0) Ds[idx] = acc[mt][nt][r];
                            else Ds[idx] += acc[mt][nt][r];
                        }
                    }
        }
    }
    __syncthreads();
    float* out = partials + ((size_t)slot * g.G + blockIdx.x) * (MT * 16 * Dn);
    for (int e = tid; e < MT * 16 * Dn; e += kJT) out[e] = Ds[e];
}

template <int MT, int NT>
__global__ __launch_bounds__(kJT, 2) void joint_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ mask, JointGeom g,
                                                             const int32_t* __restrict__ win, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // role is wave-uniform; both bodies execute the same barrier sequence
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) joint_fwd_body<MT, NT, 1>(x, y, mask, g, win, partials, lds);
    else joint_fwd_body<MT, NT, 0>(x, y, mask, g, win, partials, lds);
}

// raw[p][a][b][i][j] = sum_g partial[(p,sm,sn)][g][m = b*K+i][c = a*K+j]   (fixed order => deterministic)
__global__ __launch_bounds__(256) void joint_reduce_kernel(const float* __restrict__ partials, JointGeom g, int MTmax, float* __restrict__ raw) {
    const int TT = g.T * g.T, KK = g.K * g.K;
    const int Dn = MTmax * 16, Dsz = Dn * Dn;
    reduce_partials_block(partials, g.G, (size_t)Dsz, g.P * TT * KK, raw, [=](int e) {
        const int j = e % g.K, i = (e / g.K) % g.K, b = (e / KK) % g.T, a = (e / (KK * g.T)) % g.T, p = e / (KK * TT);
        const int m = b * g.K + i, c = a * g.K + j;
        const int sm = m / (g.tps * 16), sn = c / (g.tps * 16);
        return ((size_t)((p * g.sb + sm) * g.sb + sn) * g.G) * Dsz + (size_t)(m - sm * g.tps * 16) * Dn + (c - sn * g.tps * 16);
    });
}

// -------------------------------------------------------------------------------------------
// Epilogue: one block per window.  fp32 throughout, same operation order as iic_loss.py:124-146.
// -------------------------------------------------------------------------------------------
// One displacement d of one window, worked by ONE wave (both loss kernels call this, so their numbers are identical): normalise
// (R - mn + eps), symmetrise, marginals, the displacement's loss term (returned) and d loss / d raw[d] (written to grad_d).
__device__ __forceinline__ float loss_displacement(const float* __restrict__ Rg, float mn, int K, int TT, float lamda, float* Pw,
                                                   float* Gw, float* colv, float* rowv, float* __restrict__ grad_d, int lane) {
    const int KK = K * K;
    const float eps = 1e-16f;
    // the displacement's K x K joint is read three times, twice transposed: one coalesced pass into the wave's LDS slice first
    // (independent loads, one L2 latency) instead of a dependent L2 round trip per later access
    float* Rw = rowv + 3 * K;
    for (int e = lane; e < KK; e += 64) Rw[e] = Rg[e];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float* R = Rw;
    float z = 0.f;
    for (int e = lane; e < KK; e += 64) z += (R[e] - mn) + eps;
    z = wave_sum(z);
    for (int e = lane; e < KK; e += 64) {
        int i = e / K, j = e % K;
        float q = ((R[e] - mn) + eps) / z, qt = ((R[j * K + i] - mn) + eps) / z;
        Pw[e] = (q + qt) / 2.0f;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // the marginals' logarithms and ratios depend on one index only: evaluated once per class (2K logf instead of 2K^2;
    // same operands, same results), lanes [0,K) the column side, lanes [32,32+K) the row side when K <= 32
    if (K <= 32) {
        const int c = lane & 31;
        if (c < K) {
            float v = 0.f;
            if (lane < 32) { for (int t = 0; t < K; ++t) v += Pw[t * K + c]; }   // p_i_mat: sum over dim i, a function of j (iic_loss.py:135)
            else { for (int t = 0; t < K; ++t) v += Pw[c * K + t]; }             // p_j_mat: sum over dim j, a function of i (iic_loss.py:136)
            float* o = lane < 32 ? colv : rowv;
            o[c] = logf(v + eps);
            o[2 * K + c] = v / (v + eps);
        }
    } else {
        for (int c = lane; c < K; c += 64) {
            float cs = 0.f, rs = 0.f;
            for (int t = 0; t < K; ++t) { cs += Pw[t * K + c]; rs += Pw[c * K + t]; }
            colv[c] = logf(cs + eps), colv[2 * K + c] = cs / (cs + eps);
            rowv[c] = logf(rs + eps), rowv[2 * K + c] = rs / (rs + eps);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float part = 0.f;
    for (int e = lane; e < KK; e += 64) {
        int i = e / K, j = e % K;
        float ps = Pw[e];
        float lp = logf(ps + eps), lc = colv[j], lr = rowv[i];
        part += ps * (lp - lamda * lc - lamda * lr);
        Gw[e] = -(lp + ps / (ps + eps) - lamda * (lc + colv[2 * K + j]) - lamda * (lr + rowv[2 * K + i])) / (float)TT;
    }
    const float term = -wave_sum(part);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float sdot = 0.f;
    for (int e = lane; e < KK; e += 64) {
        int i = e / K, j = e % K;
        float gq = (Gw[e] + Gw[j * K + i]) / 2.0f;
        float q = ((R[e] - mn) + eps) / z;
        sdot += gq * q;
    }
    sdot = wave_sum(sdot);
    for (int e = lane; e < KK; e += 64) {
        int i = e / K, j = e % K;
        float gq = (Gw[e] + Gw[j * K + i]) / 2.0f;
        grad_d[e] = (gq - sdot) / z;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    return term;
}

__global__ __launch_bounds__(1024) void local_loss_kernel(const float* __restrict__ raw_all, int K, int T, float lamda,
                                                          float* __restrict__ loss, float* __restrict__ grad_all) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[17];
    const int KK = K * K, TT = T * T, nw = blockDim.x >> 6, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* raw = raw_all + (size_t)blockIdx.x * TT * KK;
    float* grad = grad_all + (size_t)blockIdx.x * TT * KK;
    float* Pw = sm + (size_t)wid * (3 * KK + 4 * K);  // per wave: Ps[KK], Gs[KK], col {log, ratio}[K, .., K], row {log, ratio}, R[KK]
    float* Gw = Pw + KK;
    float* colv = Gw + KK;       // [0,K) log(colsum + eps), [2K,3K) colsum / (colsum + eps)
    float* rowv = colv + K;      // [K,2K) and [3K,4K): the same for the row sums
    float mn = 3.4e38f;
    for (int e = threadIdx.x; e < TT * KK; e += blockDim.x) mn = fminf(mn, raw[e]);
    mn = block_min(mn, red);

    float wave_loss = 0.f;
    for (int d = wid; d < TT; d += nw) {
        wave_loss += loss_displacement(raw + (size_t)d * KK, mn, K, TT, lamda, Pw, Gw, colv, rowv, grad + (size_t)d * KK, lane);
    }
    float tot = block_sum(lane == 0 ? wave_loss : 0.f, red);
    if (threadIdx.x == 0) loss[blockIdx.x] = tot / (float)TT;
}


// The same epilogue with one block per (window, displacement): the single-block form walks its T^2 displacements in
// ceil(T^2 / 16) rounds of ~12 us each on ONE CU while the IIC chain -- the step's critical path -- waits for it.  Every block
// recomputes the window's global minimum (T^2 K^2 floats, L2-resident), wave 0 runs loss_displacement, the displacement's
// loss term goes to parts[window][d]; local_loss_finish_kernel adds them up in d order.
__global__ __launch_bounds__(1024) void local_loss_disp_kernel(const float* __restrict__ raw_all, int K, int T, float lamda,
                                                              float* __restrict__ parts, float* __restrict__ grad_all) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[17];
    const int KK = K * K, TT = T * T, wid = threadIdx.x >> 6, lane = threadIdx.x & 63, d = blockIdx.y;
    const float* raw = raw_all + (size_t)blockIdx.x * TT * KK;
    float mn = 3.4e38f;
    const int n = TT * KK;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0) {   // 16-byte pieces: ~5 independent loads per thread
        for (int e = threadIdx.x; e < n / 4; e += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(raw)[e];
            mn = fminf(fminf(mn, v.x), fminf(fminf(v.y, v.z), v.w));
        }
    } else {
        for (int e = threadIdx.x; e < n; e += blockDim.x) mn = fminf(mn, raw[e]);
    }
    mn = block_min(mn, red);
    if (wid != 0) return;
    float* Pw = sm;
    float* Gw = Pw + KK;
    float* colv = Gw + KK;
    float* rowv = colv + K;
    const float term = loss_displacement(raw + (size_t)d * KK, mn, K, TT, lamda, Pw, Gw, colv, rowv,
                                         grad_all + ((size_t)blockIdx.x * TT + d) * KK, lane);
    if (lane == 0) parts[(size_t)blockIdx.x * TT + d] = term;
}

__global__ __launch_bounds__(64) void local_loss_finish_kernel(const float* __restrict__ parts, int TT, float* __restrict__ loss) {
    const float* p = parts + (size_t)blockIdx.x * TT;
    const int lane = threadIdx.x;
    float tot = 0.f;
    for (int d0 = 0; d0 < TT; d0 += 64) {   // one coalesced load per 64 terms, then the d-ordered sum out of registers
        const float v = d0 + lane < TT ? p[d0 + lane] : 0.f;
        const int m = min(64, TT - d0);
        for (int d = 0; d < m; ++d) tot += __shfl(v, d, 64);
    }
    if (lane == 0) loss[blockIdx.x] = tot / (float)TT;
}

// -------------------------------------------------------------------------------------------
// Backward through the joint: out[n,o,h,w] += scale[p] * sum_{a,b,c} Gm[(a,b,c)][o] * src[n,c,h+s(a-pad),w+s(b-pad)]
//   dir 0: out = gx (o=i, c=j), src = y, s = -1, Gm = G[a,b,o,c]
//   dir 1: out = gy (o=j, c=i), src = x, s = +1, Gm = G[a,b,c,o]
// GEMM per output row: M = o (K padded to 32), N = 64 pixels, reduction = (a,b,c) = T*T*K.
// -------------------------------------------------------------------------------------------
struct BwdGeom {
    int N, K, Kc, H, W, pad, T, P;
    int WB, WS, RS, planeS;   // tile cols, src tile width/rows, plane stride
    int gInLds, G;
};

template <int NTP>  // pixel tiles of 16 per wave row (WB = 16*NTP)
__global__ __launch_bounds__(kThreads, 1) void local_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ mask, BwdGeom g,
                                                                  const int32_t* __restrict__ win,
                                                                  const float* __restrict__ grad_raw,
                                                                  const float* __restrict__ scale, float* __restrict__ gx,
                                                                  float* __restrict__ gy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ss = lds;                                   // src tile [Kc][RS][WS]
    float* Gs = lds + (size_t)g.Kc * g.planeS;         // Gm [(ab*Kc + c)][K] when it fits
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int TT = g.T * g.T, KK = g.K * g.K;
    const size_t plane = (size_t)g.H * g.W;

    // enumerate work: dir-major, then window, then (n, row tile, col tile)
    int64_t total = 0;
    for (int p = 0; p < g.P; ++p) {
        int tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4, tc = (win[p * 4 + 3] - win[p * 4 + 2] + g.WB - 1) / g.WB;
        total += (int64_t)g.N * tr * tc;
    }
    int curKey = -1;
    for (int64_t it = blockIdx.x; it < 2 * total; it += g.G) {
        const int dir = it >= total;
        int64_t rem = it - (dir ? total : 0);
        int p = 0, tr = 0, tc = 0;
        for (; p < g.P; ++p) {
            tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4;
            tc = (win[p * 4 + 3] - win[p * 4 + 2] + g.WB - 1) / g.WB;
            int64_t cnt = (int64_t)g.N * tr * tc;
            if (rem < cnt) break;
            rem -= cnt;
        }
        const int h0 = win[p * 4 + 0], h1 = win[p * 4 + 1], w0 = win[p * 4 + 2], w1 = win[p * 4 + 3];
        const int ct = rem % tc, rt = (rem / tc) % tr, n = rem / ((int64_t)tc * tr);
        const int row0 = h0 + rt * 4, col0 = w0 + ct * g.WB;
        const float* src = dir ? x : y;
        float* out = dir ? gy : gx;
        const int sgn = dir ? 1 : -1;
        const float* G = grad_raw + (size_t)p * TT * KK;

        __syncthreads();
        if (g.gInLds && curKey != p * 2 + dir) {
            curKey = p * 2 + dir;
            const int ng = TT * g.Kc * g.K;
            for (int idx = tid; idx < ng; idx += kThreads) {
                int o = idx % g.K, c = (idx / g.K) % g.Kc, ab = idx / (g.K * g.Kc);
                float v = 0.f;
                if (c < g.K) v = dir ? G[(size_t)ab * KK + c * g.K + o] : G[(size_t)ab * KK + o * g.K + c];
                Gs[idx] = v;
            }
        }
        // src tile rows [row0-pad, row0+4+pad), cols [col0-pad, col0+WB+pad)
        const int ns = g.Kc * g.RS * g.WS;
        for (int idx = tid; idx < ns; idx += kThreads) {
            int cx = idx % g.WS, r = (idx / g.WS) % g.RS, ch = idx / (g.WS * g.RS);
            int row = row0 - g.pad + r, col = col0 - g.pad + cx;
            float v = 0.f;
            if (ch < g.K && row >= h0 && row < h1 && col >= w0 && col < w1) {
                size_t o = (size_t)row * g.W + col;
                v = src[((size_t)n * g.K + ch) * plane + o];
                if (mask) v *= mask[(size_t)n * plane + o];
            }
            Ss[ch * g.planeS + r * g.WS + cx] = v;
        }
        __syncthreads();

        f32x4 acc[2][NTP];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int o0 = min(l15, g.K - 1), o1 = min(16 + l15, g.K - 1);
        const int ksteps = g.Kc / 4;
        for (int ab = 0; ab < TT; ++ab) {
            const int a = ab / g.T, b = ab % g.T;
            const float* sp = Ss + (wv + g.pad + sgn * (a - g.pad)) * g.WS + g.pad + sgn * (b - g.pad) + l15;
            for (int cs = 0; cs < ksteps; ++cs) {
                const int c = 4 * cs + kq;
                float a0, a1;
                if (g.gInLds) {
                    const float* gp = Gs + (size_t)(ab * g.Kc + c) * g.K;
                    a0 = gp[o0]; a1 = gp[o1];
                } else {
                    const int cc = min(c, g.K - 1);
                    const float z = c < g.K ? 1.f : 0.f;
                    a0 = z * (dir ? G[(size_t)ab * KK + cc * g.K + o0] : G[(size_t)ab * KK + o0 * g.K + cc]);
                    a1 = z * (dir ? G[(size_t)ab * KK + cc * g.K + o1] : G[(size_t)ab * KK + o1 * g.K + cc]);
                }
                const float* spc = sp + c * g.planeS;
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    float bv = spc[nt * 16];
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][nt], 0, 0, 0);
                }
            }
        }
        // D[row=o][col=pixel]: lane holds o = mt*16 + kq*4 + r at pixel nt*16 + l15
        const int row = row0 + wv;
        if (row < h1) {
            const float sc = scale[p];
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) {
                const int col = col0 + nt * 16 + l15;
                if (col < w1) {
                    const size_t po = (size_t)row * g.W + col;
                    const float mk = mask ? mask[(size_t)n * plane + po] : 1.f;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = mt * 16 + kq * 4 + r;
                            if (o < g.K) out[((size_t)n * g.K + o) * plane + po] += sc * mk * acc[mt][nt][r];
                        }
                }
            }
        }
    }
}


// -------------------------------------------------------------------------------------------
// Backward, transposed-GEMM form (the fast path).  For one output row h of direction "gx":
//   dA[(b,i), w] = sum_{(a,j)} G[a,b,i,j] * Y[j][h-(a-p)][w]          GEMM: M=(b,i)=T*K, Kred=(a,j)=T*K, N=pixels
//   gx[i][h][w'] = sum_b dA[(b,i), w'-(b-p)]                           "col2im": 7 shifted adds
// (gy: swap roles, mirror the shifts).  M is padded 140->144 only (97 %), N tiles of 16 columns carry a 2p halo
// (58 of 64 useful for p=3), Kred is exact -- versus 20->32 rows (62.5 %) in the direct form above.
// The shifted adds go through ds_add_f32 into a per-wave [K][64] LDS row buffer: lanes of one accumulator
// register never collide (a 16-row tile holds each channel i at most once) and program order fixes the
// summation order, so the result stays deterministic.
// -------------------------------------------------------------------------------------------
struct Bwd2Geom {
    int N, K, Kc, H, W, pad, T, P;
    int Mdim, Kred, ksteps;     // T*K, T*Kc, Kred/4
    int WB, RS, planeS, gstride;  // useful output columns per tile (64-2p), src rows (4+2p), plane stride, G row stride
    int G;
    int plain_rmw;   // col2im may use plain LDS read-modify-write (no two lanes of one instruction share an address)
    int accumulate;  // 0: plain stores (every output pixel is covered exactly once by this call), 1: +=
    int ablate;  // profiling only (MISEG_ABLATE): 1 no src loads, 2 no MFMA loop, 4 no col2im, 8 no output write
};

template <int MT>
__global__ __launch_bounds__(kThreads, 1) void local_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ mask, Bwd2Geom g,
                                                                   const int32_t* __restrict__ win,
                                                                   const float* __restrict__ grad_raw,
                                                                   const float* __restrict__ scale, float* __restrict__ gx,
                                                                   float* __restrict__ gy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 4, WT = 64;                       // 4 pixel tiles = 64 columns (incl. 2p halo)
    constexpr int OWS = 68;                              // Ow row stride: 4*OWS = 16 (mod 32) -> the two kq groups of a
                                                         // 32-lane half land on disjoint bank ranges (conflict-free ds_add)
    float* Gs = lds;                                     // [MT*16][gstride]   A operand: Gm[(b,o)][(a,c)]
    float* Ss = Gs + (size_t)MT * 16 * g.gstride;        // [Kc][RS][WT] src tile (cols col0-p .. col0-p+63)
    float* Os = Ss + (size_t)g.Kc * g.planeS;            // [4 waves][K][OWS] output row buffers
    int* Mtab = reinterpret_cast<int*>(Os + (size_t)4 * g.K * OWS);  // [MT*16] row m -> (shift << 16) | o, -1 for padding
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int TT = g.T * g.T, KK = g.K * g.K;
    const size_t plane = (size_t)g.H * g.W;
    float* Ow = Os + (size_t)wv * g.K * OWS;

    int64_t total = 0;
    for (int p = 0; p < g.P; ++p) {
        int tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4, tc = (win[p * 4 + 3] - win[p * 4 + 2] + g.WB - 1) / g.WB;
        total += (int64_t)g.N * tr * tc;
    }
    int arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = min(mt * 16 + l15, g.Mdim - 1) * g.gstride + kq;

    // ---- work decode + register-staged prefetch of the src tile (loads of item k+1 fly during the MFMAs of item k)
    constexpr int PF = 10;   // rows per (wave, channel) batch: RS = 4 + 2*pad <= 10 on this path (pad <= 3, checked on host)
    constexpr int CHW = 5;   // channels per wave: Kc <= 20
    struct Item { int dir, p, n, row0, col0, h0, h1, w0, w1; };
    auto decode = [&](int64_t it, Item& q) {
        q.dir = it >= total;
        int64_t rem = it - (q.dir ? total : 0);
        int p = 0, tr = 0, tc = 0;
        for (; p < g.P; ++p) {
            tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4;
            tc = (win[p * 4 + 3] - win[p * 4 + 2] + g.WB - 1) / g.WB;
            int64_t cnt = (int64_t)g.N * tr * tc;
            if (rem < cnt) break;
            rem -= cnt;
        }
        q.p = p;
        q.h0 = win[p * 4 + 0]; q.h1 = win[p * 4 + 1]; q.w0 = win[p * 4 + 2]; q.w1 = win[p * 4 + 3];
        const int ct = rem % tc, rt = (rem / tc) % tr;
        q.n = rem / ((int64_t)tc * tr);
        q.row0 = q.h0 + rt * 4; q.col0 = q.w0 + ct * g.WB;
    };
    float pf[CHW][PF];
    auto prefetch = [&](const Item& q) {
        const float* src = q.dir ? x : y;
        const int col = q.col0 - g.pad + lane;
        const bool colok = col >= q.w0 && col < q.w1 && !(g.ablate & 1);
#pragma unroll
        for (int cw = 0; cw < CHW; ++cw) {
            const int ch = wv + 4 * cw;
#pragma unroll
            for (int r = 0; r < PF; ++r) {
                const int row = q.row0 - g.pad + r;
                float v = 0.f;
                if (r < g.RS && colok && ch < g.K && row >= q.h0 && row < q.h1) {
                    const size_t o = (size_t)row * g.W + col;
                    v = src[((size_t)q.n * g.K + ch) * plane + o];
                    if (mask) v *= mask[(size_t)q.n * plane + o];
                }
                pf[cw][r] = v;
            }
        }
    };
    auto commit = [&]() {  // registers -> LDS tile
#pragma unroll
        for (int cw = 0; cw < CHW; ++cw) {
            const int ch = wv + 4 * cw;
            if (ch < g.Kc) {
#pragma unroll
                for (int r = 0; r < PF; ++r)
                    if (r < g.RS) Ss[ch * g.planeS + r * WT + lane] = pf[cw][r];
            }
        }
    };

    int curKey = -1;
    Item cur, nxt;
    if ((int64_t)blockIdx.x < 2 * total) { decode(blockIdx.x, cur); prefetch(cur); }
    for (int64_t it = blockIdx.x; it < 2 * total; it += g.G) {
        const int dir = cur.dir, p = cur.p, n = cur.n, row0 = cur.row0, col0 = cur.col0, h1 = cur.h1, w1 = cur.w1;
        float* out = dir ? gy : gx;
        const int sgn = dir ? 1 : -1;
        const float* G = grad_raw + (size_t)p * TT * KK;

        __syncthreads();   // previous item's tile and G fully consumed
        if (curKey != p * 2 + dir) {
            curKey = p * 2 + dir;
            // Gs[m=(b,o)][kred=(a,c)] = dir ? G[a,b,c,o] : G[a,b,o,c]   (zero for padded c); one row m per wave-iteration
            for (int m = wv; m < g.Mdim; m += 4) {
                const int b = m / g.K, o = m % g.K;
                for (int kr = lane; kr < g.Kred; kr += 64) {
                    const int a = kr / g.Kc, c = kr % g.Kc;
                    float v = 0.f;
                    if (c < g.K) v = dir ? G[((size_t)(a * g.T + b)) * KK + c * g.K + o] : G[((size_t)(a * g.T + b)) * KK + o * g.K + c];
                    Gs[m * g.gstride + kr] = v;
                }
            }
            for (int m = tid; m < MT * 16; m += kThreads)
                Mtab[m] = m < g.Mdim ? ((((-sgn * (m / g.K - g.pad)) + 64) << 16) | (m % g.K)) : -1;
        }
        commit();
        for (int o = 0; o < g.K; ++o) Ow[o * OWS + lane] = 0.f;
        __syncthreads();
        const bool more = it + g.G < 2 * total;
        if (more) { decode(it + g.G, nxt); prefetch(nxt); }   // global loads in flight across the MFMA phase

        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // reduction index kred = a*Kc + c; 4 consecutive c per MFMA k-step
        const int cper = g.Kc / 4;
        for (int a = 0; a < ((g.ablate & 2) ? 0 : g.T); ++a) {
            // src row for output row (row0+wv): h + sgn*(a-pad)  -> tile row wv + pad + sgn*(a-pad)
            const float* sp = Ss + (wv + g.pad + sgn * (a - g.pad)) * WT + l15 + kq * g.planeS;
            const int kbase = a * g.Kc;
            for (int cs = 0; cs < cper; ++cs) {
                float av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = Gs[arow[mt] + kbase + 4 * cs];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = sp[(4 * cs) * g.planeS + nt * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        // col2im: D[m=(b,o)][tile col w] -> Ow[o][w + shift(b)]:  gx: w' = w + (b-p); gy: w' = w - (b-p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = Mtab[mt * 16 + kq * 4 + r];
                if (e >= 0 && !(g.ablate & 4)) {
                    const int o = e & 0xffff, shift = (e >> 16) - 64;
                    float* orow = Ow + o * OWS + shift + l15;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int wc = nt * 16 + l15 + shift;
                        if (wc >= 0 && wc < WT) {
                            // rows kq*4+r of one instruction hold distinct channels unless K divides 4, 8 or 12
                            if (g.plain_rmw) orow[nt * 16] += acc[mt][nt][r];
                            else atomicAdd(orow + nt * 16, acc[mt][nt][r]);
                        }
                    }
                }
            }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        // write the valid columns [pad, pad+WB) of this wave's row: one coalesced row segment per channel
        const int row = row0 + wv;
        const int col = col0 + lane;
        if (row < h1 && lane < g.WB && col < w1 && !(g.ablate & 8)) {
            const float sc = scale[p];
            const size_t po = (size_t)row * g.W + col;
            const float mk = sc * (mask ? mask[(size_t)n * plane + po] : 1.f);
            float* op = out + (size_t)n * g.K * plane + po;
            if (g.accumulate) {
                for (int o = 0; o < g.K; ++o) op[(size_t)o * plane] += mk * Ow[o * OWS + g.pad + lane];
            } else {
                for (int o = 0; o < g.K; ++o) op[(size_t)o * plane] = mk * Ow[o * OWS + g.pad + lane];
            }
        }
        cur = nxt;
    }
}

}  // namespace miseg

using namespace miseg;

// precision (include/miseg_hip.h) -> products per fp32-class product: 1 = bf16x3, 2 = plain bf16, 3 = f16 + fp8 cross terms
static inline int nterms_of(int precision) { return precision == 1 ? 3 : precision == 3 ? 2 : 1; }

extern "C" int64_t miseg_iic_local_planes_bytes(int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W, int64_t pad);

extern "C" int64_t miseg_iic_local_joint_ws_bytes(int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad, int64_t P) {
    JointGeom g;
    if (N <= 0 || K <= 0 || P <= 0 || pad < 0 || !plan_joint(g, N, K, H, W, pad, P)) return -1;
    int cap = g.tilesM <= 4 ? 4 : 9;
    return (int64_t)g.P * g.sb * g.sb * g.G * (cap * 16) * (cap * 16) * 4;
}

extern "C" int miseg_iic_local_joint_fwd(void* stream, const float* x, const float* y, const float* mask, int64_t N,
                                         int64_t K, int64_t H, int64_t W, int64_t pad, const int32_t* win, int64_t P,
                                         float* raw, void* ws, int64_t ws_bytes, int precision) {
    MISEG_TAPE(miseg_iic_local_joint_fwd, stream, x, y, mask, N, K, H, W, pad, win, P, raw, ws, ws_bytes, precision);
    MISEG_REQUIRE(x && y && win && raw && ws, "iic_local_joint_fwd: null pointer");
    MISEG_REQUIRE(N > 0 && K > 0 && H > 0 && W > 0 && pad >= 0 && P > 0, "iic_local_joint_fwd: bad shape");
    JointGeom g;
    MISEG_REQUIRE(plan_joint(g, N, K, H, W, pad, P), "iic_local_joint_fwd: K=%ld pad=%ld does not fit LDS", (long)K, (long)pad);
    int64_t need = miseg_iic_local_joint_ws_bytes(N, K, H, W, pad, P);
    MISEG_REQUIRE(ws_bytes >= need, "iic_local_joint_fwd: workspace %ld < %ld", (long)ws_bytes, (long)need);
    const size_t ldsb = joint_lds_bytes(g);
    dim3 grid(g.G, g.P * g.sb * g.sb), block(kJT);
    hipStream_t st = as_stream(stream);
    const int cap = g.tilesM <= 4 ? 4 : 9;
    if (precision != 0 && mask == nullptr && joint_fwd_bf16_supported(g)) {   // 1: bf16 x3 split (fp32-class), 2: plain bf16
        launch_joint_fwd_px(st, x, y, g, win, (float*)ws, nterms_of(precision));
    } else if (cap == 4) {
        hipFuncSetAttribute((const void*)joint_fwd_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipLaunchKernelGGL((joint_fwd_kernel<4, 4>), grid, block, ldsb, st, x, y, mask, g, win, (float*)ws);
    } else {
        hipFuncSetAttribute((const void*)joint_fwd_kernel<9, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipLaunchKernelGGL((joint_fwd_kernel<9, 9>), grid, block, ldsb, st, x, y, mask, g, win, (float*)ws);
    }
    MISEG_LAUNCH_CHECK("joint_fwd_kernel");
    int64_t total = (int64_t)P * g.T * g.T * K * K;
    hipLaunchKernelGGL(joint_reduce_kernel, dim3(reduce_grid(total, g.G)), dim3(256), 0, st, (const float*)ws, g, cap, raw);
    MISEG_LAUNCH_CHECK("joint_reduce_kernel");
    return MISEG_OK;
}

// ---- all S sub-heads of a tap in one launch: probs[S][2*UB][K][H][W], x_s = probs[s][:UB], y_s = probs[s][UB:]
extern "C" int miseg_iic_local_joint_fwd_heads(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W,
                                               int64_t pad, const int32_t* win, int64_t P, float* raw, void* ws, int64_t ws_bytes,
                                               int precision) {
    MISEG_TAPE(miseg_iic_local_joint_fwd_heads, stream, probs, S, UB, K, H, W, pad, win, P, raw, ws, ws_bytes, precision);
    MISEG_REQUIRE(probs && win && raw && ws, "iic_local_joint_fwd_heads: null pointer");
    MISEG_REQUIRE(S > 0 && UB > 0 && K > 0 && H > 0 && W > 0 && pad >= 0 && P > 0, "iic_local_joint_fwd_heads: bad shape");
    const int64_t hs = 2 * UB * K * H * W, TT = (2 * pad + 1) * (2 * pad + 1);
    JointGeom g;
    MISEG_REQUIRE(plan_joint(g, UB, K, H, W, pad, P * S), "iic_local_joint_fwd_heads: K=%ld pad=%ld does not fit LDS", (long)K, (long)pad);
    if (precision != 0 && joint_fwd_bf16_supported(g)) {
        MISEG_REQUIRE(ws_bytes >= miseg_iic_local_joint_ws_bytes(UB, K, H, W, pad, P * S), "iic_local_joint_fwd_heads: workspace too small");
        hipStream_t st = as_stream(stream);
        const int cap = g.tilesM <= 4 ? 4 : 9;
        JointGeom gh = g;                    // planned for P*S slots (sets G); the kernel sees P windows x S heads
        gh.P = (int)P; gh.S = (int)S; gh.hs = hs;
        launch_joint_fwd_px(st, probs, probs + UB * K * H * W, gh, win, (float*)ws, nterms_of(precision));
        MISEG_LAUNCH_CHECK("joint_fwd_px_kernel");
        const int64_t total = P * S * TT * K * K;
        hipLaunchKernelGGL(joint_reduce_kernel, dim3(reduce_grid(total, g.G)), dim3(256), 0, st, (const float*)ws, g, cap, raw);
        MISEG_LAUNCH_CHECK("joint_reduce_kernel");
        return MISEG_OK;
    }
    for (int64_t s = 0; s < S; ++s) {        // other shapes / exact fp32: one launch per sub-head
        const int rc = miseg_iic_local_joint_fwd(stream, probs + s * hs, probs + s * hs + UB * K * H * W, nullptr, UB, K, H, W, pad, win, P,
                                                 raw + s * P * TT * K * K, ws, ws_bytes, precision);
        if (rc != MISEG_OK) return rc;
    }
    return MISEG_OK;
}

// the same at precision 3 (forward: bf16 hi / lo), and every probability it stages also leaves as the backward's operand planes
extern "C" int miseg_iic_local_joint_fwd_heads_planes(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W,
                                                      int64_t pad, const int32_t* win, int64_t P, float* raw, void* ws, int64_t ws_bytes,
                                                      void* planes, int64_t planes_bytes) {
    MISEG_TAPE(miseg_iic_local_joint_fwd_heads_planes, stream, probs, S, UB, K, H, W, pad, win, P, raw, ws, ws_bytes, planes, planes_bytes);
    MISEG_REQUIRE(probs && win && raw && ws && planes, "iic_local_joint_fwd_heads_planes: null pointer");
    MISEG_REQUIRE(P > 0, "iic_local_joint_fwd_heads_planes: bad shape");
    const int64_t need = miseg_iic_local_planes_bytes(S, UB, K, H, W, pad);
    MISEG_REQUIRE(need > 0, "iic_local_joint_fwd_heads_planes: K = 20, pad = 3 only");
    MISEG_REQUIRE(planes_bytes >= need, "iic_local_joint_fwd_heads_planes: plane buffer %ld < %ld", (long)planes_bytes, (long)need);
    const int64_t hs = 2 * UB * K * H * W, TT = (2 * pad + 1) * (2 * pad + 1);
    JointGeom g;
    MISEG_REQUIRE(plan_joint(g, UB, K, H, W, pad, P * S) && joint_fwd_bf16_supported(g), "iic_local_joint_fwd_heads_planes: shape not supported");
    MISEG_REQUIRE(ws_bytes >= miseg_iic_local_joint_ws_bytes(UB, K, H, W, pad, P * S), "iic_local_joint_fwd_heads_planes: workspace too small");
    hipStream_t st = as_stream(stream);
    const int cap = g.tilesM <= 4 ? 4 : 9;
    JointGeom gh = g;
    gh.P = (int)P; gh.S = (int)S; gh.hs = hs;
    MISEG_REQUIRE(launch_joint_fwd_px(st, probs, probs + UB * K * H * W, gh, win, (float*)ws, 3, static_cast<unsigned char*>(planes)) == 0,
                  "iic_local_joint_fwd_heads_planes: no kernel for this shape");
    MISEG_LAUNCH_CHECK("joint_fwd_px_kernel");
    const int64_t total = P * S * TT * K * K;
    hipLaunchKernelGGL(joint_reduce_kernel, dim3(reduce_grid(total, g.G)), dim3(256), 0, st, (const float*)ws, g, cap, raw);
    MISEG_LAUNCH_CHECK("joint_reduce_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_local_bwd_heads(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W,
                                         int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale,
                                         float* gprob, int accumulate, int precision, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_iic_local_bwd_heads, stream, probs, S, UB, K, H, W, pad, win, P, grad_raw, scale, gprob, accumulate, precision, ws, ws_bytes);
    MISEG_REQUIRE(probs && win && grad_raw && scale && gprob, "iic_local_bwd_heads: null pointer");
    MISEG_REQUIRE(S > 0 && UB > 0 && K > 0 && K <= 32 && H > 0 && W > 0 && pad >= 0 && P > 0, "iic_local_bwd_heads: bad shape (K<=32)");
    const int64_t hs = 2 * UB * K * H * W, half = UB * K * H * W, TT = (2 * pad + 1) * (2 * pad + 1);
    if (precision != 0 && local_bwd_bf16_supported(UB, K, H, W, pad)) {
        MISEG_REQUIRE(ws && ws_bytes >= (int64_t)local_bwd_bf16_ws_bytes(K, pad, P * S), "iic_local_bwd_heads: workspace too small");
        launch_local_bwd_rows(as_stream(stream), probs, probs + half, S, hs, UB, K, H, W, pad, win, P, grad_raw, scale, gprob, gprob + half,
                              accumulate, ws, nterms_of(precision));
        MISEG_LAUNCH_CHECK("local_bwd_rows_kernel");
        return MISEG_OK;
    }
    for (int64_t s = 0; s < S; ++s) {
        const int rc = miseg_iic_local_bwd(stream, probs + s * hs, probs + s * hs + half, nullptr, UB, K, H, W, pad, win, P,
                                           grad_raw + s * P * TT * K * K, scale + s * P, gprob + s * hs, gprob + s * hs + half, accumulate,
                                           precision, ws, ws_bytes);
        if (rc != MISEG_OK) return rc;
    }
    return MISEG_OK;
}

// ---- operand planes of the f16 + fp8 backward (see miseg_hip.h)
extern "C" int64_t miseg_iic_local_planes_bytes(int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W, int64_t pad) {
    if (S <= 0 || UB <= 0 || H <= 0 || W <= 0 || !local_bwd_f8_supported(K, pad) || !local_bwd_bf16_supported(UB, K, H, W, pad)) return 0;
    if (S * 2 * UB * H * W * 40 >= ((int64_t)1 << 32)) return 0;
    return mi_planes_bytes(S * 2 * UB, H * W);
}

extern "C" int miseg_iic_local_make_planes(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W, int64_t pad,
                                           void* planes, int64_t planes_bytes) {
    MISEG_TAPE(miseg_iic_local_make_planes, stream, probs, S, UB, K, H, W, pad, planes, planes_bytes);
    MISEG_REQUIRE(probs && planes, "iic_local_make_planes: null pointer");
    const int64_t need = miseg_iic_local_planes_bytes(S, UB, K, H, W, pad);
    MISEG_REQUIRE(need > 0, "iic_local_make_planes: K = 20, pad = 3 only");
    MISEG_REQUIRE(planes_bytes >= need, "iic_local_make_planes: buffer %ld < %ld", (long)planes_bytes, (long)need);
    launch_make_planes(as_stream(stream), probs, S * 2 * UB, H * W, static_cast<unsigned char*>(planes));
    MISEG_LAUNCH_CHECK("make_planes_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_local_bwd_heads_planes(void* stream, const void* planes, int64_t planes_bytes, int64_t S, int64_t UB, int64_t K, int64_t H,
                                                int64_t W, int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale,
                                                float* gprob, int accumulate, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_iic_local_bwd_heads_planes, stream, planes, planes_bytes, S, UB, K, H, W, pad, win, P, grad_raw, scale, gprob, accumulate, ws, ws_bytes);
    MISEG_REQUIRE(planes && win && grad_raw && scale && gprob && ws, "iic_local_bwd_heads_planes: null pointer");
    MISEG_REQUIRE(P > 0, "iic_local_bwd_heads_planes: bad shape");
    const int64_t need = miseg_iic_local_planes_bytes(S, UB, K, H, W, pad);
    MISEG_REQUIRE(need > 0, "iic_local_bwd_heads_planes: K = 20, pad = 3 only");
    MISEG_REQUIRE(planes_bytes >= need, "iic_local_bwd_heads_planes: plane buffer %ld < %ld", (long)planes_bytes, (long)need);
    MISEG_REQUIRE(ws_bytes >= (int64_t)local_bwd_bf16_ws_bytes(K, pad, P * S), "iic_local_bwd_heads_planes: workspace too small");
    const int64_t hs = 2 * UB * K * H * W, half = UB * K * H * W;
    launch_local_bwd_rows(as_stream(stream), nullptr, nullptr, S, hs, UB, K, H, W, pad, win, P, grad_raw, scale, gprob, gprob + half, accumulate, ws, 2,
                          static_cast<const unsigned char*>(planes));
    MISEG_LAUNCH_CHECK("local_bwd_f8_kernel");
    return MISEG_OK;
}

extern "C" int64_t miseg_iic_local_bwd_ws_bytes(int64_t K, int64_t pad, int64_t P) {
    return (int64_t)local_bwd_bf16_ws_bytes(K, pad, P) + 16;
}

extern "C" int64_t miseg_iic_local_loss_ws_bytes(int64_t pad, int64_t P) { return P * (2 * pad + 1) * (2 * pad + 1) * 4; }

extern "C" int miseg_iic_local_loss_fwd_ws(void* stream, const float* raw, int64_t K, int64_t pad, int64_t P, float lamda,
                                           float* loss, float* grad_raw, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_iic_local_loss_fwd_ws, stream, raw, K, pad, P, lamda, loss, grad_raw, ws, ws_bytes);
    MISEG_REQUIRE(raw && loss && grad_raw && ws, "iic_local_loss_fwd_ws: null pointer");
    MISEG_REQUIRE(K > 0 && K <= 64 && pad >= 0 && P > 0, "iic_local_loss_fwd_ws: bad shape");
    MISEG_REQUIRE(ws_bytes >= miseg_iic_local_loss_ws_bytes(pad, P), "iic_local_loss_fwd_ws: workspace too small");
    const int T = 2 * (int)pad + 1;
    const size_t ldsb = (size_t)(3 * K * K + 4 * K) * 4;
    hipLaunchKernelGGL(local_loss_disp_kernel, dim3((unsigned)P, (unsigned)(T * T)), dim3(1024), ldsb, as_stream(stream), raw, (int)K, T,
                       lamda, (float*)ws, grad_raw);
    MISEG_LAUNCH_CHECK("local_loss_disp_kernel");
    hipLaunchKernelGGL(local_loss_finish_kernel, dim3((unsigned)P), dim3(64), 0, as_stream(stream), (const float*)ws, T * T, loss);
    MISEG_LAUNCH_CHECK("local_loss_finish_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_local_loss_fwd(void* stream, const float* raw, int64_t K, int64_t pad, int64_t P, float lamda,
                                        float* loss, float* grad_raw) {
    MISEG_TAPE(miseg_iic_local_loss_fwd, stream, raw, K, pad, P, lamda, loss, grad_raw);
    MISEG_REQUIRE(raw && loss && grad_raw, "iic_local_loss_fwd: null pointer");
    MISEG_REQUIRE(K > 0 && K <= 64 && pad >= 0 && P > 0, "iic_local_loss_fwd: bad shape");
    const int T = 2 * (int)pad + 1;
    const size_t ldsb = (size_t)16 * (3 * K * K + 4 * K) * 4;
    MISEG_REQUIRE(ldsb <= (size_t)kLdsBudget, "iic_local_loss_fwd: K too large");
    hipFuncSetAttribute((const void*)local_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL(local_loss_kernel, dim3((unsigned)P), dim3(1024), ldsb, as_stream(stream), raw, (int)K, T, lamda, loss,
                       grad_raw);
    MISEG_LAUNCH_CHECK("local_loss_kernel");
    return MISEG_OK;
}

extern "C" int miseg_iic_local_bwd(void* stream, const float* x, const float* y, const float* mask, int64_t N, int64_t K,
                                   int64_t H, int64_t W, int64_t pad, const int32_t* win, int64_t P, const float* grad_raw,
                                   const float* scale, float* gx, float* gy, int accumulate, int precision,
                                   void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_iic_local_bwd, stream, x, y, mask, N, K, H, W, pad, win, P, grad_raw, scale, gx, gy, accumulate, precision, ws, ws_bytes);
    MISEG_REQUIRE(x && y && win && grad_raw && scale && gx && gy, "iic_local_bwd: null pointer");
    MISEG_REQUIRE(N > 0 && K > 0 && K <= 32 && H > 0 && W > 0 && pad >= 0 && P > 0, "iic_local_bwd: bad shape (K<=32)");
    hipStream_t st = as_stream(stream);
    if (precision != 0 && mask == nullptr && local_bwd_bf16_supported(N, K, H, W, pad)) {   // bf16 MFMA, hi/lo split (1) or plain (2)
        MISEG_REQUIRE(ws && ws_bytes >= (int64_t)local_bwd_bf16_ws_bytes(K, pad, P), "iic_local_bwd: workspace too small for the bf16 path");
        launch_local_bwd_rows(st, x, y, 1, 0, N, K, H, W, pad, win, P, grad_raw, scale, gx, gy, accumulate, ws, nterms_of(precision));
        MISEG_LAUNCH_CHECK("local_bwd_rows_kernel");
        return MISEG_OK;
    }
    {   // fast path: transposed GEMM + LDS col2im (needs T*K <= 144 rows and the G matrix + tiles in LDS)
        Bwd2Geom b2;
        b2.N = (int)N; b2.K = (int)K; b2.Kc = ((int)K + 3) & ~3; b2.H = (int)H; b2.W = (int)W; b2.pad = (int)pad; b2.T = 2 * (int)pad + 1;
        b2.P = (int)P; b2.Mdim = b2.T * b2.K; b2.Kred = b2.T * b2.Kc; b2.ksteps = b2.Kred / 4;
        b2.WB = 64 - 2 * b2.pad; b2.RS = 4 + 2 * b2.pad; b2.planeS = (b2.RS * 64) | 1; b2.gstride = b2.Kred | 1; b2.G = 256;
        b2.ablate = 0;
        const int mt = (b2.Mdim + 15) / 16;
        const int mtc = mt <= 4 ? 4 : 9;
        const size_t lds2 = ((size_t)mtc * 16 * b2.gstride + (size_t)b2.Kc * b2.planeS + (size_t)4 * b2.K * 68 + (size_t)mtc * 16) * 4;
        b2.accumulate = accumulate;
        b2.plain_rmw = (4 % b2.K != 0) && (8 % b2.K != 0) && (12 % b2.K != 0);
        if (mt <= 9 && b2.WB >= 16 && b2.RS <= 10 && b2.Kc <= 20 && lds2 <= (size_t)kLdsBudget) {
            if (mtc == 4) {
                hipFuncSetAttribute((const void*)local_bwd2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                hipLaunchKernelGGL((local_bwd2_kernel<4>), dim3(b2.G), dim3(kThreads), lds2, st, x, y, mask, b2, win, grad_raw, scale, gx, gy);
            } else {
                hipFuncSetAttribute((const void*)local_bwd2_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                hipLaunchKernelGGL((local_bwd2_kernel<9>), dim3(b2.G), dim3(kThreads), lds2, st, x, y, mask, b2, win, grad_raw, scale, gx, gy);
            }
            MISEG_LAUNCH_CHECK("local_bwd2_kernel");
            return MISEG_OK;
        }
    }
    if (!accumulate) {  // the direct-form fallback below always accumulates
        hipMemsetAsync(gx, 0, (size_t)N * K * H * W * sizeof(float), st);
        hipMemsetAsync(gy, 0, (size_t)N * K * H * W * sizeof(float), st);
    }
    BwdGeom g;
    g.N = (int)N; g.K = (int)K; g.Kc = ((int)K + 3) & ~3; g.H = (int)H; g.W = (int)W; g.pad = (int)pad; g.T = 2 * (int)pad + 1;
    g.P = (int)P; g.RS = 4 + 2 * g.pad;
    int ntp = 0;
    size_t ldsb = 0;
    for (int cand : {4, 2, 1}) {
        g.WB = 16 * cand; g.WS = g.WB + 2 * g.pad; g.planeS = (g.RS * g.WS) | 1;
        size_t srcb = (size_t)g.Kc * g.planeS * 4, gb = (size_t)g.T * g.T * g.Kc * g.K * 4;
        if (srcb + gb <= (size_t)kLdsBudget) { g.gInLds = 1; ntp = cand; ldsb = srcb + gb; break; }
    }
    if (!ntp) {
        for (int cand : {4, 2, 1}) {
            g.WB = 16 * cand; g.WS = g.WB + 2 * g.pad; g.planeS = (g.RS * g.WS) | 1;
            size_t srcb = (size_t)g.Kc * g.planeS * 4;
            if (srcb <= (size_t)kLdsBudget) { g.gInLds = 0; ntp = cand; ldsb = srcb; break; }
        }
    }
    MISEG_REQUIRE(ntp, "iic_local_bwd: K=%ld pad=%ld does not fit LDS", (long)K, (long)pad);
    g.G = 256;
#define MISEG_BWD_LAUNCH(NTP)                                                                                         \
    hipFuncSetAttribute((const void*)local_bwd_kernel<NTP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);   \
    hipLaunchKernelGGL((local_bwd_kernel<NTP>), dim3(g.G), dim3(kThreads), ldsb, st, x, y, mask, g, win, grad_raw, scale, gx, gy)
    if (ntp == 4) { MISEG_BWD_LAUNCH(4); } else if (ntp == 2) { MISEG_BWD_LAUNCH(2); } else { MISEG_BWD_LAUNCH(1); }
#undef MISEG_BWD_LAUNCH
    MISEG_LAUNCH_CHECK("local_bwd_kernel");
    return MISEG_OK;
}
