// Backward of the local-MI joint on the bf16 matrix cores, hi/lo operand split (fp32-class accuracy).
//
// Same transposed GEMM as local_bwd2_kernel (mi_local.hip), per output row h and direction:
//   dA[(b,o), w] = sum_{kred=(a,c)} Gm[(b,o)][(a,c)] * S[c][h + s(a-p)][w]      M = T*K (140 -> 144), N = 64 px, Kred = T*K (140 -> 160)
//   out[o][h][w'] = sum_b dA[(b,o), w' -+ (b-p)]                                 "col2im"
// What is different on bf16 MFMA (v_mfma_f32_16x16x32_bf16):
//   * A (the gradient matrix Gm) is packed once per call into bf16 hi/lo k-step slices [dir][ks][plane][144][32] (row-swizzled),
//     and streamed L2 -> registers -> LDS one 18 KB slice ahead of the MFMAs (it no longer fits LDS whole);
//   * B needs 8 consecutive reduction elements (= 8 channel planes of one pixel) per lane while the tile is planar [c][row][w]:
//     ds_read_b64_tr_b16 transposes 4 planes x 16 pixels on the fly, so the tile keeps its coalescing-friendly layout;
//   * col2im is a gather: each wave parks 3 M-tiles of D at a time in LDS and every lane (= one output column) sums the rows it
//     needs into K registers -- independent LDS reads instead of the read-modify-write chain of the fp32 kernel.
// K and PAD are template parameters (K=20; PAD=3 and 1: the shipped taps); other shapes use the fp32 kernels.
#include "mi_local.h"

namespace miseg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_S16X4(ptr) ((__attribute__((address_space(3))) s16x4*)(ptr))

struct Bwd3Geom {
    int N, H, W, P, G, accumulate;
};

template <int K, int PAD>
struct B3 {
    static constexpr int T = 2 * PAD + 1, MD = T * K, MT = (MD + 15) / 16, MP = MT * 16, KRED = T * K, KS = (KRED + 31) / 32;
    static constexpr int RS = 4 + 2 * PAD, WT = 64, WB = WT - 2 * PAD, SW = 72;
    static constexpr int MG = MT >= 9 ? 3 : 2, NG = (MT + MG - 1) / MG;
};

__device__ __forceinline__ unsigned short bf16_hi(float v) { return f32_to_bf16_bits(v); }

// gpack[p][dir][ks][pl][m][32]: value(m=(b,o), kred=(a,c)) = dir ? G[a,b,c,o] : G[a,b,o,c]; 16-byte slot s of row m stored at s ^ ((m>>2)&3)
template <int K, int PAD, int NP>
__global__ void pack_g_bf16_kernel(const float* __restrict__ grad_raw, int P, unsigned short* __restrict__ gpack) {
    typedef B3<K, PAD> C;
    const int total = P * 2 * C::KS * C::MP * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int kk = e & 31, m = (e >> 5) % C::MP, ks = (e / (32 * C::MP)) % C::KS, dir = (e / (32 * C::MP * C::KS)) & 1,
                  p = e / (32 * C::MP * C::KS * 2);
        const int kred = ks * 32 + kk;
        float v = 0.f;
        if (m < C::MD && kred < C::KRED) {
            const int b = m / K, o = m % K, a = kred / K, c = kred % K;
            const float* G = grad_raw + (size_t)p * C::T * C::T * K * K + (size_t)(a * C::T + b) * K * K;
            v = dir ? G[c * K + o] : G[o * K + c];
        }
        const unsigned short hi = bf16_hi(v);
        const size_t base = ((((size_t)(p * 2 + dir) * C::KS + ks) * NP) * C::MP + m) * 32 + 8 * ((kk >> 3) ^ ((m >> 2) & 3)) + (kk & 7);
        gpack[base] = hi;
        if (NP == 2) gpack[base + (size_t)C::MP * 32] = bf16_hi(v - bf16_bits_to_f32(hi));
    }
}

template <int K, int PAD, int NTERMS>
__global__ __launch_bounds__(256, 1) void local_bwd_bf16_kernel(const float* __restrict__ x, const float* __restrict__ y, Bwd3Geom g,
                                                                const int32_t* __restrict__ win,
                                                                const unsigned short* __restrict__ gpack,
                                                                const float* __restrict__ scale, float* __restrict__ gx,
                                                                float* __restrict__ gy) {
    typedef B3<K, PAD> C;
    constexpr int NP = NTERMS == 1 ? 1 : 2, NT = 4;
    constexpr int T = C::T, MT = C::MT, MP = C::MP, KS = C::KS, RS = C::RS, WT = C::WT, SW = C::SW, MG = C::MG, NG = C::NG;
    constexpr int SLICE = NP * MP * 32;                  // bf16 elements per k-step slice
    constexpr int SPLANE = K * RS * SW, DROW = 65;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned short* Gsl = reinterpret_cast<unsigned short*>(ldsb);               // [2][NP][MP][32]
    unsigned short* Ss = Gsl + 2 * SLICE;                                        // [NP][K][RS][SW]
    float* Dst = reinterpret_cast<float*>(Ss + NP * SPLANE);                     // [4 waves][MG*16][DROW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    float* Dw = Dst + (size_t)wv * MG * 16 * DROW;
    const size_t plane = (size_t)g.H * g.W;

    int64_t total = 0;
    for (int p = 0; p < g.P; ++p) {
        int tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4, tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WB - 1) / C::WB;
        total += (int64_t)g.N * tr * tc;
    }
    struct Item { int dir, p, n, row0, col0, h0, h1, w0, w1; };
    auto decode = [&](int64_t it, Item& o) {
        o.dir = it >= total;
        int64_t rem = it - (o.dir ? total : 0);
        int p = 0, tr = 0, tc = 0;
        for (; p < g.P; ++p) {
            tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4;
            tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WB - 1) / C::WB;
            int64_t cnt = (int64_t)g.N * tr * tc;
            if (rem < cnt) break;
            rem -= cnt;
        }
        o.p = p;
        o.h0 = win[p * 4 + 0]; o.h1 = win[p * 4 + 1]; o.w0 = win[p * 4 + 2]; o.w1 = win[p * 4 + 3];
        const int ct = rem % tc, rt = (rem / tc) % tr;
        o.n = rem / ((int64_t)tc * tr);
        o.row0 = o.h0 + rt * 4; o.col0 = o.w0 + ct * C::WB;
    };
    // src tile prefetch (fp32 values in registers, split into bf16 planes at commit): (ch,row) pairs pr = wv + 4*bi
    constexpr int PFN = (K * RS + 3) / 4;
    float pf[PFN];
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)tbytes, 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;
    auto prefetch = [&](const Item& o) {
        const int col = o.col0 - PAD + lane;
        const unsigned vo = (col >= o.w0 && col < o.w1) ? (unsigned)col * 4u : OOB;
#pragma unroll
        for (int bi = 0; bi < PFN; ++bi) {
            const int pr = wvu + 4 * bi;
            const int ch = pr / RS, r = pr - ch * RS, row = o.row0 - PAD + r;
            const bool ok = pr < K * RS && row >= o.h0 && row < o.h1;
            const unsigned so = ok ? (unsigned)((((size_t)o.n * K + ch) * plane + (size_t)row * g.W) * 4) : OOB;
            pf[bi] = __uint_as_float(o.dir ? __builtin_amdgcn_raw_buffer_load_b32(rsx, (int)(so + vo), 0, 0)
                                           : __builtin_amdgcn_raw_buffer_load_b32(rsy, (int)(so + vo), 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int bi = 0; bi < PFN; ++bi) {
            const int pr = wv + 4 * bi;
            if (pr < K * RS) {
                const int ch = pr / RS, r = pr - ch * RS;
                unsigned short* d = Ss + (ch * RS + r) * SW + lane;
                const unsigned short hi = bf16_hi(pf[bi]);
                d[0] = hi;
                if (NP == 2) d[SPLANE] = bf16_hi(pf[bi] - bf16_bits_to_f32(hi));
            }
        }
    };
    // G slice streaming: SLICE*2 bytes = SLICE/8 uint4 per slice, <= 5 per thread
    constexpr int GQ = (SLICE / 8 + 255) / 256;
    u32x4 gq[GQ];
    auto gload = [&](const unsigned short* slice) {
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
            const int idx = tid + 256 * i;
            if (idx < SLICE / 8) gq[i] = *reinterpret_cast<const u32x4*>(slice + (size_t)idx * 8);
        }
    };
    auto gstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
            const int idx = tid + 256 * i;
            if (idx < SLICE / 8) *reinterpret_cast<u32x4*>(Gsl + (size_t)buf * SLICE + (size_t)idx * 8) = gq[i];
        }
    };

    Item cur, nxt;
    if ((int64_t)blockIdx.x < 2 * total) { decode(blockIdx.x, cur); prefetch(cur); }
    for (int64_t it = blockIdx.x; it < 2 * total; it += g.G) {
        const int dir = cur.dir, p = cur.p, n = cur.n, row0 = cur.row0, col0 = cur.col0, h1 = cur.h1, w1 = cur.w1;
        float* out = dir ? gy : gx;
        const int sgn = dir ? 1 : -1;
        const unsigned short* gbase = gpack + (size_t)(p * 2 + dir) * KS * SLICE;
        __syncthreads();                         // previous item done with Ss / Gsl / Dst
        commit();
        gload(gbase);
        gstore(0);
        __syncthreads();
        const bool more = it + g.G < 2 * total;
        if (more) { decode(it + g.G, nxt); prefetch(nxt); }

        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) gload(gbase + (size_t)(ks + 1) * SLICE);
            const unsigned short* Gb = Gsl + (size_t)(ks & 1) * SLICE;
            // B fragments: kred block of 8 = two 4-blocks kb, kb+4; 4-block -> (a, c0..c0+3) never straddles a (K % 4 == 0)
            bf16x8_t bfr[NP][NT];
            {
                const int qq = l15 >> 2, pp = l15 & 3;
                int soff[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int kb = min(ks * 32 + 8 * q + 4 * hf, C::KRED - 4);
                    const int a = kb / K, c0 = kb - a * K;
                    soff[hf] = ((c0 + qq) * RS + (wv + PAD + sgn * (a - PAD))) * SW + 4 * pp;
                }
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(Ss + pl * SPLANE + soff[0] + 16 * nt));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(Ss + pl * SPLANE + soff[1] + 16 * nt));
                        const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        bfr[pl][nt] = __builtin_bit_cast(bf16x8_t, f);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + l15;
                bf16x8_t af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[pl] = *reinterpret_cast<const bf16x8_t*>(Gb + ((size_t)pl * MP + m) * 32 + 8 * (q ^ ((m >> 2) & 3)));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (NTERMS == 3) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1], bfr[0][nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[NP - 1][nt], acc[mt][nt], 0, 0, 0);
                    }
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[0][nt], acc[mt][nt], 0, 0, 0);
                }
            }
            if (ks + 1 < KS) gstore((ks + 1) & 1);
            __syncthreads();
        }
        // ---- col2im gather: lane = output tile column wc; out[o] += D[(b,o)][wc - shift(b)], shift = -sgn*(b-PAD)
        float outv[K];
#pragma unroll
        for (int o = 0; o < K; ++o) outv[o] = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
            for (int ml = 0; ml < MG; ++ml) {
                const int mt = gi * MG + ml;
                if (mt < MT) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Dw[(ml * 16 + q * 4 + r) * DROW + nt * 16 + l15] = acc[mt][nt][r];
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
#pragma unroll
            for (int rr = 0; rr < MG * 16; ++rr) {
                const int m = gi * MG * 16 + rr;          // compile-time after unrolling
                if (m < C::MD) {
                    const int b = m / K, o = m % K;
                    const int src_col = lane + sgn * (b - PAD);   // wc = w + shift  =>  w = wc - shift = wc + sgn*(b-PAD)
                    const float v = (src_col >= 0 && src_col < WT) ? Dw[rr * DROW + src_col] : 0.f;
                    outv[o] += v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
        const int row = row0 + wv, col = col0 + lane - PAD;
        if (row < h1 && lane >= PAD && lane < PAD + C::WB && col < w1) {
            const float sc = scale[p];
            float* op = out + (size_t)n * K * plane + (size_t)row * g.W + col;
            if (g.accumulate) {
#pragma unroll
                for (int o = 0; o < K; ++o) op[(size_t)o * plane] += sc * outv[o];
            } else {
#pragma unroll
                for (int o = 0; o < K; ++o) op[(size_t)o * plane] = sc * outv[o];
            }
        }
        cur = nxt;
    }
}

template <int K, int PAD>
static size_t bwd3_lds(int nterms) {
    typedef B3<K, PAD> C;
    const int np = nterms == 1 ? 1 : 2;
    return (size_t)2 * np * C::MP * 32 * 2 + (size_t)np * K * C::RS * C::SW * 2 + (size_t)4 * C::MG * 16 * 65 * 4;
}

bool local_bwd_bf16_supported(int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad) {
    return K == 20 && (pad == 3 || pad == 1) && (size_t)N * K * H * W * 4 < 0x40000000ull;
}

size_t local_bwd_bf16_ws_bytes(int64_t K, int64_t pad, int64_t P) {
    const int T = 2 * (int)pad + 1, MP = ((T * (int)K + 15) / 16) * 16, KS = (T * (int)K + 31) / 32;
    return (size_t)P * 2 * KS * 2 * MP * 32 * 2;
}

template <int K, int PAD>
static int launch_bwd3(hipStream_t st, const float* x, const float* y, Bwd3Geom g, const int32_t* win, const float* grad_raw,
                       const float* scale, float* gx, float* gy, void* ws, int nterms) {
    typedef B3<K, PAD> C;
    unsigned short* gpack = reinterpret_cast<unsigned short*>(ws);
    const int total = g.P * 2 * C::KS * C::MP * 32;
    if (nterms == 1) hipLaunchKernelGGL((pack_g_bf16_kernel<K, PAD, 1>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, g.P, gpack);
    else hipLaunchKernelGGL((pack_g_bf16_kernel<K, PAD, 2>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, g.P, gpack);
    const size_t lds = bwd3_lds<K, PAD>(nterms);
    if (nterms == 1) {
        hipFuncSetAttribute((const void*)local_bwd_bf16_kernel<K, PAD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((local_bwd_bf16_kernel<K, PAD, 1>), dim3(g.G), dim3(256), lds, st, x, y, g, win, gpack, scale, gx, gy);
    } else {
        hipFuncSetAttribute((const void*)local_bwd_bf16_kernel<K, PAD, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((local_bwd_bf16_kernel<K, PAD, 3>), dim3(g.G), dim3(256), lds, st, x, y, g, win, gpack, scale, gx, gy);
    }
    return 0;
}

int launch_local_bwd_bf16(hipStream_t st, const float* x, const float* y, int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad,
                          const int32_t* win, int64_t P, const float* grad_raw, const float* scale, float* gx, float* gy, int accumulate,
                          void* ws, int nterms) {
    Bwd3Geom g{(int)N, (int)H, (int)W, (int)P, 256, accumulate};
    if (pad == 3) return launch_bwd3<20, 3>(st, x, y, g, win, grad_raw, scale, gx, gy, ws, nterms);
    return launch_bwd3<20, 1>(st, x, y, g, win, grad_raw, scale, gx, gy, ws, nterms);
}

}  // namespace miseg
