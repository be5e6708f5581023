// Local-MI displacement joint on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), fp32-class accuracy via operand splitting.
//
// Same GEMM as mi_local.hip:  D[(dx,i),(dy,j)] += sum_w X[i][r][w+dx] * Y[j][r-dy][w]   (M = N = T*K, reduction over pixels)
// but the bf16 MFMA wants 8 CONSECUTIVE reduction elements per lane:
//   * B = Y[j][r-dy][w..w+7]: a displacement in y is a row select -> fragments are always 16-byte aligned;
//   * A = X[i][r][w+dx..w+dx+7]: a displacement in x mis-aligns the fragment by dx elements.  Each wave pair therefore
//     materialises the T x-shifted copies of its current row chunk in LDS ("As", 18 KB per 32-pixel k-step for T=7):
//     a lane reads a 24-element window of the raw row once and emits the T copies with register funnel shifts
//     (v_alignbit for odd shifts, plain register selection for even ones) as aligned ds_write_b128.
// NTERMS = 3: x = hi + lo (two bf16), D += Ahi*Bhi + Ahi*Blo + Alo*Bhi: products carry ~2^-17 relative error, unbiased, and
// average out over the 1e6-pixel reduction -> the joint matches the fp32 kernel to ~1e-6 (tests), at 3/16 of its MFMA time.
// NTERMS = 1: plain bf16 operands (fast mode).
#include "mi_local.h"

namespace miseg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BRB = 4;          // X rows per block tile: one per wave pair
constexpr int BKW = 64;         // pixels per tile row (2 k-steps of 32)
constexpr int BXW = BKW + 16;   // raw X row: columns col0-8 .. col0+BKW+8
constexpr int BYW = BKW + 8;    // Y row stride (144 B: conflict-free b128 fragment reads)
constexpr int BAW = 32;         // As row: 64 B = four 16-B slots, XOR-swizzled by (row>>2)&3 -> conflict-free b128 reads, no padding

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
}
__device__ __forceinline__ void split_store(float v, unsigned short* hi, unsigned short* lo) {
    const unsigned short h = f32_to_bf16_bits(v);
    *hi = h;
    if (lo) *lo = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
}

constexpr int kBT = 256;        // 4 waves = 2 pairs, one wave per SIMD (512-register budget: D tiles + prefetch registers)

template <int MT, int NT, int PAD, int NTERMS, int ROLE>
__device__ __forceinline__ void joint_bf16_body(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ mask,
                                                const JointGeom& g, const int32_t* __restrict__ win, float* __restrict__ partials,
                                                unsigned char* lds) {
    typedef TileSet<MT, NT, ROLE> TS;
    constexpr int T = 2 * PAD + 1, NP = NTERMS == 1 ? 1 : 2, RBY = BRB + 2 * PAD;
    const int K = g.K;
    // LDS carve (bf16 = unsigned short)
    unsigned short* Xr = reinterpret_cast<unsigned short*>(lds);                 // [NP][K][BRB][BXW]
    unsigned short* Ys = Xr + (size_t)NP * K * BRB * BXW;                        // [NP][K][RBY][BYW]
    unsigned short* Asb = Ys + (size_t)NP * K * RBY * BYW;                       // [2 pairs][NP][T][K][BAW]
    const size_t xPlane = (size_t)K * BRB * BXW, yPlane = (size_t)K * RBY * BYW, aPlane = (size_t)T * K * BAW;
    const int tid = threadIdx.x, lane = tid & 63, pair = tid >> 7, ptid = tid & 127, wv = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    unsigned short* As = Asb + (size_t)pair * NP * aPlane;

    const int slot = blockIdx.y;                          // one sub-block only on this path (T*K <= 144): slot = sub-head * P + window
    const int shead = slot / g.P, p = slot - shead * g.P;
    x += (size_t)shead * g.hs;
    y += (size_t)shead * g.hs;
    const int h0 = win[p * 4 + 0], h1 = win[p * 4 + 1], w0 = win[p * 4 + 2], w1 = win[p * 4 + 3];
    const int tr = (h1 - h0 + BRB - 1) / BRB, tc = (w1 - w0 + BKW - 1) / BKW;
    const int nItems = g.N * tr * tc;
    const int mtu = g.tilesM, ntu = g.tilesM;

    int aoff[MT], boff[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = min(mt * 16 + l15, g.Mdim - 1);
        aoff[mt] = m * BAW + 8 * (q ^ ((m >> 2) & 3));    // m = dx*K + i is exactly the As row index; swizzled slot
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = min(nt * 16 + l15, g.Mdim - 1);
        boff[nt] = ((c % K) * RBY + (2 * PAD - c / K)) * BYW + 8 * q;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t plane = (size_t)g.H * g.W;
    // ---- register-staged prefetch: each wave owns (ch,row) pairs pr = wv, wv+4, ...  (K*BRB/4 <= 20 X rows, K*RBY/4 <= 50 Y rows)
    constexpr int XB = 20, YB = 50;
    float xv[XB], xe[XB], yv[YB];
    const int nxr = K * BRB, nyr = K * RBY;
    // Buffer loads: one descriptor per tensor, a 32-bit byte offset per load (row part wave-uniform, column part
    // per lane) and hardware range checking: an out-of-window row/column is given an out-of-range offset and reads 0.
    // (64-bit flat addresses for ~90 in-flight loads cost 180 address registers and spilled.)
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)tbytes, 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    auto prefetch = [&](int it) {
        const int ct = it % tc, rt = (it / tc) % tr, n = it / (tc * tr);
        const int row0 = h0 + rt * BRB, col0 = w0 + ct * BKW;
        const int colx = col0 - 8 + lane, cole = colx + 64, coly = col0 + lane;
        const unsigned vx = (colx >= w0 && colx < w1) ? (unsigned)colx * 4u : OOB;
        const unsigned ve = (lane < 16 && cole >= w0 && cole < w1) ? (unsigned)cole * 4u : OOB;
        const unsigned vy = (coly < w1) ? (unsigned)coly * 4u : OOB;
#pragma unroll
        for (int bi = 0; bi < XB; ++bi) {
            const int pr = wvu + 4 * bi;
            const int ch = pr / BRB, r = pr - ch * BRB, row = row0 + r;
            const bool ok = pr < nxr && row < h1;
            const unsigned so = ok ? (unsigned)((((size_t)n * K + ch) * plane + (size_t)row * g.W) * 4) : OOB;
            xv[bi] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx_, (int)(so + vx), 0, 0));
            xe[bi] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx_, (int)(so + ve), 0, 0));
        }
#pragma unroll
        for (int bi = 0; bi < YB; ++bi) {
            const int pr = wvu + 4 * bi;
            const int ch = pr / RBY, r = pr - ch * RBY, row = row0 - PAD + r;
            const bool ok = pr < nyr && row >= h0 && row < h1;
            const unsigned so = ok ? (unsigned)((((size_t)n * K + ch) * plane + (size_t)row * g.W) * 4) : OOB;
            yv[bi] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry_, (int)(so + vy), 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int bi = 0; bi < XB; ++bi) {
            const int pr = wv + 4 * bi;
            if (pr < nxr) {
                const int ch = pr / BRB, r = pr - ch * BRB;
                unsigned short* d = Xr + ((size_t)ch * BRB + r) * BXW + lane;
                split_store(xv[bi], d, NP == 2 ? d + xPlane : nullptr);
                if (lane < 16) split_store(xe[bi], d + 64, NP == 2 ? d + 64 + xPlane : nullptr);
            }
        }
#pragma unroll
        for (int bi = 0; bi < YB; ++bi) {
            const int pr = wv + 4 * bi;
            if (pr < nyr) {
                const int ch = pr / RBY, r = pr - ch * RBY;
                unsigned short* d = Ys + ((size_t)ch * RBY + r) * BYW + lane;
                split_store(yv[bi], d, NP == 2 ? d + yPlane : nullptr);
            }
        }
    };

    if ((int)blockIdx.x < nItems) prefetch(blockIdx.x);
    for (int it = blockIdx.x; it < nItems; it += g.G) {
        __syncthreads();                 // previous tile fully consumed
        commit();
        __syncthreads();
        if (it + g.G < nItems) prefetch(it + g.G);   // in flight during the MFMA phases below
#pragma unroll 1
        for (int step = 0; step < (BRB / 2) * (BKW / 32); ++step) {
            const int rx = pair + 2 * (step / (BKW / 32)), ks = step % (BKW / 32);
            // ---- materialise the T shifted copies of row rx, pixels [32ks, 32ks+32): As[pl][dx][i][k] = Xr[pl][i][rx][8+32ks+k+dx-PAD]
            for (int task = ptid; task < NP * K * 4; task += 128) {
                const int c8 = task & 3, i = (task >> 2) % K, pl = task / (4 * K);
                const unsigned* wsrc = reinterpret_cast<const unsigned*>(Xr + pl * xPlane + ((size_t)i * BRB + rx) * BXW + 32 * ks + 8 * c8);
                unsigned dwin[12];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(wsrc + 4 * t);
                    dwin[4 * t + 0] = v[0]; dwin[4 * t + 1] = v[1]; dwin[4 * t + 2] = v[2]; dwin[4 * t + 3] = v[3];
                }
#pragma unroll
                for (int dx = 0; dx < T; ++dx) {
                    constexpr int base = 8 - PAD;
                    const int s = base + dx;
                    u32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int d0 = s / 2 + t;
                        o[t] = (s & 1) ? __builtin_amdgcn_alignbit(dwin[d0 + 1], dwin[d0], 16) : dwin[d0];
                    }
                    const int m = dx * K + i;
                    *reinterpret_cast<u32x4*>(As + pl * aPlane + (size_t)m * BAW + 8 * (c8 ^ ((m >> 2) & 3))) = o;
                }
            }
            __syncthreads();
            bf16x8_t af[NP][MT];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (TS::row_used(mt)) af[pl][mt] = *reinterpret_cast<const bf16x8_t*>(As + pl * aPlane + aoff[mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (TS::col_used(nt) && nt < ntu) {
                    bf16x8_t bfr[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        bfr[pl] = *reinterpret_cast<const bf16x8_t*>(Ys + pl * yPlane + boff[nt] + rx * BYW + 32 * ks);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (TS::mine(mt, nt) && mt < mtu) {
                            if (NTERMS == 3) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1][mt], bfr[0], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bfr[NP - 1], acc[mt][nt], 0, 0, 0);
                            }
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bfr[0], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();   // As is rewritten by the next step
        }
    }
    // ---- reduce the 2 pairs' accumulators through LDS in fixed order, then one partial per block
    const int Dn = NT * 16;
    float* Ds = reinterpret_cast<float*>(lds);
    for (int w = 0; w < 2; ++w) {
        __syncthreads();
        if (pair == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (TS::mine(mt, nt)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = (mt * 16 + q * 4 + r) * Dn + nt * 16 + l15;
                            if (w == 0) Ds[idx] = acc[mt][nt][r];
                            else Ds[idx] += acc[mt][nt][r];
                        }
                    }
        }
    }
    __syncthreads();
    float* out = partials + ((size_t)slot * g.G + blockIdx.x) * (MT * 16 * Dn);
    for (int e = tid; e < MT * 16 * Dn; e += kBT) out[e] = Ds[e];
}

template <int MT, int NT, int PAD, int NTERMS>
__global__ __launch_bounds__(kBT, 1) void joint_fwd_bf16_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ mask, JointGeom g,
                                                                  const int32_t* __restrict__ win, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) joint_bf16_body<MT, NT, PAD, NTERMS, 1>(x, y, mask, g, win, partials, ldsb);
    else joint_bf16_body<MT, NT, PAD, NTERMS, 0>(x, y, mask, g, win, partials, ldsb);
}

static size_t bf16_lds_bytes(const JointGeom& g, int nterms) {
    const int np = nterms == 1 ? 1 : 2, T = g.T, rby = BRB + 2 * g.pad;
    const size_t tiles = ((size_t)np * g.K * BRB * BXW + (size_t)np * g.K * rby * BYW + (size_t)2 * np * T * g.K * BAW) * 2;
    const int cap = g.tilesM <= 4 ? 4 : 9;
    const size_t dred = (size_t)(cap * 16) * (cap * 16) * 4;
    return tiles > dred ? tiles : dred;
}

bool joint_fwd_bf16_supported(const JointGeom& g) {
    if (g.sb != 1) return false;
    if (!((g.pad == 3 && g.tilesM > 4 && g.tilesM <= 9) || (g.pad == 1 && g.tilesM <= 4))) return false;
    if ((size_t)g.N * g.K * g.H * g.W * 4 >= 0x40000000ull) return false;   // 32-bit buffer offsets with an out-of-range marker
    if (g.K * BRB > 80 || g.K * (BRB + 2 * g.pad) > 200) return false;   // prefetch register batches (XB, YB rows per wave)
    return bf16_lds_bytes(g, 3) <= (size_t)kLdsBudget;
}

int launch_joint_fwd_bf16(hipStream_t st, const float* x, const float* y, const float* mask, const JointGeom& g, const int32_t* win,
                          float* partials, int nterms) {
    const size_t ldsb = bf16_lds_bytes(g, nterms);
    dim3 grid(g.G, g.P * g.S), block(kBT);
#define JB(MT, NT, PAD, NTERMS)                                                                                                  \
    {                                                                                                                            \
        hipFuncSetAttribute((const void*)joint_fwd_bf16_kernel<MT, NT, PAD, NTERMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
        hipLaunchKernelGGL((joint_fwd_bf16_kernel<MT, NT, PAD, NTERMS>), grid, block, ldsb, st, x, y, mask, g, win, partials);   \
    }
    if (g.pad == 3) { if (nterms == 1) JB(9, 9, 3, 1) else JB(9, 9, 3, 3) }
    else { if (nterms == 1) JB(4, 4, 1, 1) else JB(4, 4, 1, 3) }
#undef JB
    return 0;
}

}  // namespace miseg
