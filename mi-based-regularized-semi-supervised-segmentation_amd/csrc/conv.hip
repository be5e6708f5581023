// U-Net convolutions on gfx950 MFMA (ref: contrastyou/arch/unet.py:10-40, 66-84, 86-133).
//
// conv3x3 (stride 1, pad 1, no bias) as an implicit GEMM over NHWC activations:
//   M = 16-pixel row segments, N = output channels, reduction = 9 taps x input channels.
// The A operand is gathered straight from a haloed LDS tile of the input (a tap shift is a whole
// pixel = a whole channel vector, so fragments stay 16-byte aligned); nearest-x2 upsampling and
// the skip-connection concat are index math in the tile loader, never materialised.
// dt = bf16: v_mfma_f32_16x16x32_bf16; dt = f32: v_mfma_f32_16x16x4_f32 (exact fp32 parity mode).
// Weight gradients use the fp32 MFMA for both dtypes (deterministic split-K over pixels).
#include "common.h"

namespace miseg {

constexpr int kCT = 256;  // threads per block (4 waves)
constexpr int TH = 16;    // tile rows: 4 per wave
constexpr int CK = 32;    // input channels per LDS chunk

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int CKP = 40;   // padded channel stride (elements): 80 B rows, 16-B aligned (2-way on ds_read_b128's lane groups)
    static constexpr int KP = 48;    // 96 B rows: the 16-lane groups of ds_read_b128 hit 64 distinct banks (conv3x3_kernel)
    static constexpr int VEC = 8;    // elements per 16-byte vector
    typedef s16x8 Frag;
    static __device__ __forceinline__ Frag load(const bf16* base, int kq) { return *reinterpret_cast<const Frag*>(base + 8 * kq); }
    static __device__ __forceinline__ void mma_chunk(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int CKP = 34;   // 2m+kq bank pattern: conflict-free ds_read_b32 gathers
    static constexpr int KP = 34;
    static constexpr int VEC = 4;
    struct Frag { float v[8]; };
    static __device__ __forceinline__ Frag load(const float* base, int kq) {
        Frag f;
#pragma unroll
        for (int s = 0; s < 8; ++s) f.v[s] = base[4 * s + kq];
        return f;
    }
    static __device__ __forceinline__ void mma_chunk(const Frag& a, const Frag& b, f32x4& c) {
#pragma unroll
        for (int s = 0; s < 8; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], c, 0, 0, 0);
    }
};

struct ConvSrc {
    const void* p0; const void* p1;
    int C0, C1, ups0, ups1;
};

template <typename T>
__device__ __forceinline__ void zero_vec(T* dst) {
#pragma unroll
    for (int i = 0; i < Mma<T>::VEC; ++i) dst[i] = from_f32<T>(0.f);
}

// COT: output channels per block; TW: tile width (32 or 16).  grid = (N*tilesR*tilesC, ceil(Cout/COT)).
// POOL: as in conv3x3_stream_kernel below -- the output leaves 2x2 sum-pooled (H, W even; no statistics).
// BNL: the (single, full-resolution) source is a BatchNorm layer's raw output and bl.gy the gradient of its activation -- the loader
// forms graw (common.h, bn_graw_vec); out-of-image halo pixels stay zero.  RED: the epilogue also takes the BatchNorm-backward sums of
// the layer that produced this launch's output tensor (BnRed; non-pooled output, Cout % 4 == 0, no forward statistics).
// NW waves per block (4, or 8 for the 16-row tiles of the deep layers: the same LDS tile -- one block per CU either way -- worked by two
// waves per SIMD that cover each other's LDS round trips, each with half the rows).
template <typename T, int COT, int TW, int THT, bool POOL = false, bool BNL = false, bool RED = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void conv3x3_kernel(ConvSrc src, int N, int H, int W, const T* __restrict__ wpk, int Cout,
                                                        T* __restrict__ out, float* __restrict__ stats, BnFinish fin, BnLoad bl, BnRed br) {
    static_assert(!(RED && POOL), "the epilogue reduce exists for full-resolution outputs");
    typedef Mma<T> MM;
    constexpr int kCT = 64 * NW;                            // shadows the file-scope block size inside this kernel
    static_assert(THT % NW == 0, "whole rows per wave");
    constexpr int VEC = MM::VEC, NT = COT / 16, MTR = TW / 16, RW = THT / NW, MTW = RW * MTR;   // RW rows of the tile per wave
    constexpr int KP = MM::KP;                              // LDS stride of one pixel / one weight row (elements)
    constexpr int IW = TW + 2, IH = THT + 2;
    constexpr int NIS = (IH * IW * (CK / VEC) + kCT - 1) / kCT, NWS = (9 * COT * (CK / VEC) + kCT - 1) / kCT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Is = reinterpret_cast<T*>(smem);                    // [IH][IW][KP]
    T* Ws = Is + IH * IW * KP;                             // [9][COT][KP]
    const int Cin = src.C0 + src.C1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int tilesC = (W + TW - 1) / TW, tilesR = (H + THT - 1) / THT;
    const int tc = blockIdx.x % tilesC, tr = (blockIdx.x / tilesC) % tilesR, n = blockIdx.x / (tilesC * tilesR);
    const int h0 = tr * THT, w0 = tc * TW, co0 = blockIdx.y * COT;

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One 32-channel chunk of the haloed input tile and of the weights is in flight in registers while the previous
    // chunk is on the matrix cores (the block is alone on its CU: 1 wave per SIMD, so nothing else hides the latency).
    uint4 pin[NIS], pwt[NWS];
    uint4 pgy[BNL ? NIS : 1];                    // BNL: the activation gradient beside the raw output, the slot's coefficients,
    float cfr[BNL ? kBwdCoefRows * VEC : 1];     // and which slots lie inside the image (the others stay zero, not P - Q * mean)
    unsigned okm = 0;
    static_assert(NIS <= 32 && kCT % (CK / VEC) == 0, "slot mask / per-thread channel vector");
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    auto fetch = [&](int c0) __attribute__((always_inline)) {
        if (BNL) {
            okm = 0;
            const int c = c0 + (tid % (CK / VEC)) * VEC;         // the same channel vector in every slot of this thread
#pragma unroll
            for (int r = 0; r < kBwdCoefRows; ++r)
#pragma unroll
                for (int i = 0; i < VEC; i += 4) {
                    const float4 q = c < Cin ? *reinterpret_cast<const float4*>(bl.coef + (size_t)r * Cin + c + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                    cfr[r * VEC + i] = q.x; cfr[r * VEC + i + 1] = q.y; cfr[r * VEC + i + 2] = q.z; cfr[r * VEC + i + 3] = q.w;
                }
        }
#pragma unroll
        for (int j = 0; j < NIS; ++j) {
            const int idx = tid + kCT * j;
            const int v = idx % (CK / VEC), px = idx / (CK / VEC), ix = px % IW, iy = px / IW;
            const int h = h0 - 1 + iy, w = w0 - 1 + ix, c = c0 + v * VEC;
            uint4 val = zero4;
            if (idx < IH * IW * (CK / VEC) && h >= 0 && h < H && w >= 0 && w < W && c < Cin) {
                const T* sp;
                if (BNL) {
                    const size_t o = (((size_t)n * H + h) * W + w) * src.C0 + c;
                    sp = reinterpret_cast<const T*>(src.p0) + o;
                    pgy[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(bl.gy) + o);
                    okm |= 1u << j;
                } else if (c < src.C0) {
                    const int hs = H >> src.ups0, wsz = W >> src.ups0;
                    sp = reinterpret_cast<const T*>(src.p0) + (((size_t)n * hs + (h >> src.ups0)) * wsz + (w >> src.ups0)) * src.C0 + c;
                } else {
                    const int hs = H >> src.ups1, wsz = W >> src.ups1;
                    sp = reinterpret_cast<const T*>(src.p1) + (((size_t)n * hs + (h >> src.ups1)) * wsz + (w >> src.ups1)) * src.C1 + (c - src.C0);
                }
                val = *reinterpret_cast<const uint4*>(sp);
            }
            pin[j] = val;
        }
#pragma unroll
        for (int j = 0; j < NWS; ++j) {
            const int idx = tid + kCT * j;
            const int v = idx % (CK / VEC), co = (idx / (CK / VEC)) % COT, tap = idx / ((CK / VEC) * COT);
            const int c = c0 + v * VEC;
            uint4 val = zero4;
            if (idx < 9 * COT * (CK / VEC) && co0 + co < Cout && c < Cin)
                val = *reinterpret_cast<const uint4*>(wpk + ((size_t)tap * Cout + co0 + co) * Cin + c);
            pwt[j] = val;
        }
    };
    // every fragment read below is one of these two bases plus a compile-time offset (an instruction immediate)
    const T* Ib = Is + ((wv * RW) * IW + l15) * KP;
    const T* Wb = Ws + l15 * KP;
    fetch(0);
    auto chunk = [&](int c0) __attribute__((always_inline)) {
        __syncthreads();                       // previous chunk's MFMAs are done with Is / Ws
#pragma unroll
        for (int j = 0; j < NIS; ++j) {
            const int idx = tid + kCT * j;
            if (idx < IH * IW * (CK / VEC)) {
                uint4 val = pin[j];
                if (BNL) {
                    typedef typename VT<T>::Raw Raw;
                    if ((okm >> j) & 1u) {
                        const Raw g = bn_graw_vec<T>(*reinterpret_cast<const Raw*>(&pin[j]), *reinterpret_cast<const Raw*>(&pgy[j]), cfr, cfr + VEC,
                                                     cfr + 2 * VEC, cfr + 3 * VEC, cfr + 4 * VEC, cfr + 5 * VEC);
                        val = *reinterpret_cast<const uint4*>(&g);
                    } else val = zero4;
                }
                *reinterpret_cast<uint4*>(Is + (idx / (CK / VEC)) * KP + (idx % (CK / VEC)) * VEC) = val;
            }
        }
#pragma unroll
        for (int j = 0; j < NWS; ++j) {
            const int idx = tid + kCT * j;
            if (idx < 9 * COT * (CK / VEC)) *reinterpret_cast<uint4*>(Ws + (idx / (CK / VEC)) * KP + (idx % (CK / VEC)) * VEC) = pwt[j];
        }
        __syncthreads();
        if (c0 + CK < Cin) fetch(c0 + CK);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            typename MM::Frag bf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = MM::load(Wb + (tap * COT + t * 16) * KP, kq);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                typename MM::Frag af = MM::load(Ib + ((m / MTR + ky) * IW + (m % MTR) * 16 + kx) * KP, kq);
#pragma unroll
                for (int t = 0; t < NT; ++t) MM::mma_chunk(bf[t], af, acc[m][t]);   // D^T: rows = channels, cols = pixels
            }
        }
    };
    // RED: the producer's raw output at this block's output positions (a lane's 4 channels of a pixel per (m, t)) and its coefficients
    // are requested before the LAST chunk goes to the matrix cores, so that they have arrived when the epilogue wants them
    struct RedQuad { float v[4]; };
    RedQuad rp[RED ? MTW : 1][RED ? NT : 1];
    float rsc[RED ? NT : 1][4], rsh[RED ? NT : 1][4], rmu[RED ? NT : 1][4];
    if (RED) {
        int c0 = 0;
        for (; c0 + CK < Cin; c0 += CK) chunk(c0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = co0 + t * 16 + kq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = co + r < Cout;
                rmu[t][r] = in ? br.saved[co + r] : 0.f;
                rsc[t][r] = in ? br.saved[2 * Cout + co + r] : 0.f;
                rsh[t][r] = in ? br.saved[3 * Cout + co + r] : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
            const T* rq = reinterpret_cast<const T*>(br.raw) + (((size_t)n * H + h) * W + w) * Cout;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = co0 + t * 16 + kq * 4;
                T q[4] = {from_f32<T>(0.f), from_f32<T>(0.f), from_f32<T>(0.f), from_f32<T>(0.f)};
                if (h < H && w < W && co + 3 < Cout) {
                    if (sizeof(T) == 2) *reinterpret_cast<uint2*>(q) = *reinterpret_cast<const uint2*>(rq + co);
                    else *reinterpret_cast<uint4*>(q) = *reinterpret_cast<const uint4*>(rq + co);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) rp[m][t].v[r] = to_f32(q[r]);
            }
        }
        chunk(c0);
    } else {
        for (int c0 = 0; c0 < Cin; c0 += CK) chunk(c0);
    }
    // ---- epilogue: D^T[row = channel t*16 + kq*4 + r][col = pixel l15]: a lane owns 4 consecutive channels of one pixel, so
    // the 16 lanes x 4 kq of a wave store whole NHWC pixel vectors (8-byte pieces, contiguous across kq)
    if (POOL) {      // rows 2 pr, 2 pr + 1 of the wave (two registers of the lane) + the column neighbour's sum; even lanes store
        static_assert(!POOL || RW % 2 == 0, "pooled epilogue: a wave owns whole row pairs");
        const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
        for (int mp = 0; mp < MTW / 2; ++mp) {
            const int m0 = (2 * (mp / MTR)) * MTR + mp % MTR, m1 = m0 + MTR;
            const int hp = ((h0 + wv * RW) >> 1) + mp / MTR, wp = ((w0 + (mp % MTR) * 16) >> 1) + (l15 >> 1);
            const bool mine = !(l15 & 1) && hp < Hp && wp < Wp;
            T* op = out + (((size_t)n * Hp + hp) * Wp + wp) * Cout;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = co0 + t * 16 + kq * 4;
                T pk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[m0][t][r] + acc[m1][t][r];
                    v += __shfl_xor(v, 1, 64);
                    pk[r] = from_f32<T>(v);
                }
                if (!mine) continue;
                if (co + 3 < Cout) {
                    if (sizeof(T) == 2) *reinterpret_cast<uint2*>(op + co) = *reinterpret_cast<const uint2*>(pk);
                    else *reinterpret_cast<uint4*>(op + co) = *reinterpret_cast<const uint4*>(pk);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < Cout) op[co + r] = pk[r];
                }
            }
        }
        return;
    }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[t][r] = s2[t][r] = 0.f;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
        const bool ok = h < H && w < W;
        const size_t px = ((size_t)n * H + h) * W + w;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = co0 + t * 16 + kq * 4;
            // `op + co` is the lane's 4 channels; two destinations: a lane never straddles the split (a multiple of 4)
            T* op = out + px * Cout;
            if (fin.out1) op = co < fin.split ? out + px * fin.split : reinterpret_cast<T*>(fin.out1) + px * (Cout - fin.split) - fin.split;
            if (ok && co + 3 < Cout) {
                T pk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[m][t][r];
                    pk[r] = from_f32<T>(v);
                    if (RED) {      // dz = [y > 0] * (the gradient as the producer's backward will read it); sum dz, sum dz * (raw - mean)
                        const float x = rp[m][t].v[r];
                        const float dz = x * rsc[t][r] + rsh[t][r] > relu_keep_threshold<T>() ? to_f32(pk[r]) : 0.f;
                        s1[t][r] += dz;
                        s2[t][r] += dz * (x - rmu[t][r]);
                    } else {
                        s1[t][r] += v;
                        s2[t][r] += v * v;
                    }
                }
                if (sizeof(T) == 2) *reinterpret_cast<uint2*>(op + co) = *reinterpret_cast<const uint2*>(pk);
                else *reinterpret_cast<uint4*>(op + co) = *reinterpret_cast<const uint4*>(pk);
            } else if (ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < Cout) {
                        const float v = acc[m][t][r];
                        op[co + r] = from_f32<T>(v);
                        s1[t][r] += v;
                        s2[t][r] += v * v;
                    }
            }
        }
    }
    if (RED) stats = br.parts;
    if (stats) {
        __shared__ float sred[NW][2][COT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[t][r], b = s2[t][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    a += __shfl_xor(a, off, 64);
                    b += __shfl_xor(b, off, 64);
                }
                if (l15 == 0) { sred[wv][0][t * 16 + kq * 4 + r] = a; sred[wv][1][t * 16 + kq * 4 + r] = b; }
            }
        __syncthreads();
        if (tid < COT && co0 + tid < Cout) {
            float a = sred[0][0][tid] + sred[1][0][tid] + sred[2][0][tid] + sred[3][0][tid];
            float b = sred[0][1][tid] + sred[1][1][tid] + sred[2][1][tid] + sred[3][1][tid];
#pragma unroll
            for (int q = 4; q < NW; ++q) { a += sred[q][0][tid]; b += sred[q][1][tid]; }
            if (RED) b *= br.saved[Cout + co0 + tid];          // sum dz * (raw - mean) -> sum dz * xhat
            if (!RED && fin.acc) {
                bn_acc_add(fin.acc + co0 + tid, fin.acc + 2 * Cout, a);
                bn_acc_add(fin.acc + Cout + co0 + tid, fin.acc + 2 * Cout, b);
            } else {
                store_part(&stats[((size_t)blockIdx.x * 2 + 0) * Cout + co0 + tid], a);
                store_part(&stats[((size_t)blockIdx.x * 2 + 1) * Cout + co0 + tid], b);
            }
        }
        if (!RED && fin.counter && last_block_arrives(fin.counter, gridDim.x * gridDim.y)) {   // the tile memory is dead: scratch for the sums
            bn_finish_block(stats, (int)gridDim.x, Cout, fin, reinterpret_cast<float*>(smem));
            last_block_done(fin.counter);
        }
    }
}

// Persistent form of conv3x3_kernel for 16-bit storage (plain forms: no BatchNorm loader / epilogue reduce / last-block finish).
// conv3x3_kernel spends most of a block's life outside the matrix cores: a 64^2 layer with 32 input channels is ONE chunk per tile, so
// the first (and only) global fetch, the two barriers, the epilogue stores and the 2 x NT x 4 x 4-shuffle statistics reduction are all
// exposed, once per tile, three to six block generations per CU (profiles/r04a_bench_bf16_kernel_stats.csv: 1.06 ms per step at 15-26 %
// of the matrix peak).  Here a block owns a CONTIGUOUS run of tiles of one output-channel slice and software-pipelines across them:
//   * while the LAST chunk of tile i is on the matrix cores, chunk 0 of tile i + 1 (haloed input + weights) is already in flight in
//     registers, so the epilogue of tile i (stores, statistics) runs under that fetch instead of in front of it;
//   * BatchNorm statistics are kept per lane across the block's tiles and reduced once per block (one partial row per block:
//     parts = gridDim.x, which also shrinks what bn_finalize has to read);
//   * consecutive tiles of a block are spatial neighbours (their halos hit the XCD's L2 while still warm) and share the weight slice.
// Arithmetic, tile shape, LDS layout and epilogues are those of conv3x3_kernel; only the summation order of the statistics differs
// (per block over its tiles instead of per tile).
template <typename T, int COT, int TW, int THT, bool POOL, int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW, 1) void conv3x3_pt_kernel(ConvSrc src, int N, int H, int W, const T* __restrict__ wpk, int Cout,
                                                           T* __restrict__ out, float* __restrict__ stats, BnFinish fin, int ntiles, int per) {
    static_assert(sizeof(T) == 2, "persistent tiled form: 16-bit storage types");
    static_assert(DEPTH == 1 || DEPTH == 2, "stages in flight ahead of the one on the matrix cores");
    typedef Mma<T> MM;
    constexpr int kCT = 64 * NW;
    static_assert(THT % NW == 0, "whole rows per wave");
    constexpr int VEC = MM::VEC, NT = COT / 16, MTR = TW / 16, RW = THT / NW, MTW = RW * MTR;
    constexpr int KP = MM::KP;
    constexpr int IW = TW + 2, IH = THT + 2;
    constexpr int NIS = (IH * IW * (CK / VEC) + kCT - 1) / kCT, NWS = (9 * COT * (CK / VEC) + kCT - 1) / kCT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Is = reinterpret_cast<T*>(smem);                    // [IH][IW][KP]
    T* Ws = Is + IH * IW * KP;                             // [9][COT][KP]
    const int Cin = src.C0 + src.C1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int tilesC = (W + TW - 1) / TW, tilesR = (H + THT - 1) / THT;
    const int co0 = blockIdx.y * COT;
    const int t_begin = blockIdx.x * per, t_end = min(t_begin + per, ntiles);
    const int nck = (Cin + CK - 1) / CK;

    // A STAGE = one 32-channel chunk of one tile (haloed input + weights).  DEPTH stages are in flight in registers ahead of the one
    // on the matrix cores, across tile boundaries: set A / set B alternate (static register names: the stage loop is unrolled by two).
    uint4 pinA[NIS], pwtA[NWS], pinB[DEPTH == 2 ? NIS : 1], pwtB[DEPTH == 2 ? NWS : 1];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // A thread's slots of the haloed input tile and of the weight slice never change: pixel, channel vector, tap, output channel.  Their
    // geometry is worked out once; a stage adds the tile's corner and the chunk's first channel (both block-uniform) and tests ranges.
    // (Derived per stage, the index math was most of this kernel's 6.4 vector instructions per MFMA: profiles/r04_pmc.json.)
    constexpr int CV4 = CK / VEC;
    int i_iy[NIS], i_ix[NIS], i_cv[NIS], i_rel0[NIS], i_rel1[NIS];       // row - 1, column - 1, channel offset in the chunk (< 0: no slot), offsets inside source 0 / 1
    int w_cv[NWS], w_rel[NWS];                                            // channel offset in the chunk (< 0: no slot), offset of (tap, co) in the packed weights
    {
        const int ws0 = W >> src.ups0, ws1 = W >> src.ups1;
#pragma unroll
        for (int j = 0; j < NIS; ++j) {
            const int idx = tid + kCT * j, v = idx % CV4, px = idx / CV4;
            i_ix[j] = px % IW - 1, i_iy[j] = px / IW - 1;
            i_cv[j] = idx < IH * IW * CV4 ? v * VEC : -1;
            // (h0 + d) >> ups = (h0 >> ups) + (d >> ups): tile corners are multiples of THT x TW, the shift is arithmetic (d = -1 stays outside)
            i_rel0[j] = ((i_iy[j] >> src.ups0) * ws0 + (i_ix[j] >> src.ups0)) * src.C0 + v * VEC;
            i_rel1[j] = ((i_iy[j] >> src.ups1) * ws1 + (i_ix[j] >> src.ups1)) * src.C1 + v * VEC - src.C0;
        }
#pragma unroll
        for (int j = 0; j < NWS; ++j) {
            const int idx = tid + kCT * j, v = idx % CV4, co = (idx / CV4) % COT, tap = idx / (CV4 * COT);
            w_cv[j] = (idx < 9 * COT * CV4 && co0 + co < Cout) ? v * VEC : -1;
            w_rel[j] = (tap * Cout + co0 + co) * Cin + v * VEC;
        }
    }
    auto fetch = [&](uint4 (&pin)[NIS], uint4 (&pwt)[NWS], int n, int h0, int w0, int c0) __attribute__((always_inline)) {
        const T* s0 = reinterpret_cast<const T*>(src.p0) + (((size_t)n * (H >> src.ups0) + (h0 >> src.ups0)) * (W >> src.ups0) + (w0 >> src.ups0)) * src.C0 + c0;
        const T* s1 = reinterpret_cast<const T*>(src.p1) + (((size_t)n * (H >> src.ups1) + (h0 >> src.ups1)) * (W >> src.ups1) + (w0 >> src.ups1)) * src.C1 + c0;
#pragma unroll
        for (int j = 0; j < NIS; ++j) {
            const int c = c0 + i_cv[j];
            uint4 val = zero4;
            if (i_cv[j] >= 0 && c < Cin && (unsigned)(h0 + i_iy[j]) < (unsigned)H && (unsigned)(w0 + i_ix[j]) < (unsigned)W)
                val = *reinterpret_cast<const uint4*>(c < src.C0 ? s0 + i_rel0[j] : s1 + i_rel1[j]);
            pin[j] = val;
        }
        const T* wc = wpk + c0;
#pragma unroll
        for (int j = 0; j < NWS; ++j) {
            uint4 val = zero4;
            if (w_cv[j] >= 0 && c0 + w_cv[j] < Cin) val = *reinterpret_cast<const uint4*>(wc + w_rel[j]);
            pwt[j] = val;
        }
    };
    const T* Ib = Is + ((wv * RW) * IW + l15) * KP;
    const T* Wb = Ws + l15 * KP;
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[t][r] = s2[t][r] = 0.f;
    f32x4 acc[MTW][NT];

    // two cursors over the block's stages, both block-uniform (scalar unit): f = the next stage to FETCH, p = the stage to PROCESS
    struct Cur { int tile, ck, n, h0, w0; };
    auto decode = [&](Cur& c) __attribute__((always_inline)) {
        c.n = c.tile / (tilesC * tilesR); c.h0 = ((c.tile / tilesC) % tilesR) * THT; c.w0 = (c.tile % tilesC) * TW;
    };
    auto advance = [&](Cur& c) __attribute__((always_inline)) {
        if (++c.ck == nck) { c.ck = 0; ++c.tile; if (c.tile < t_end) decode(c); }
    };
    Cur f{t_begin, 0, 0, 0, 0}, p{t_begin, 0, 0, 0, 0};
    if (t_begin < t_end) { decode(f); p = f; }
    if (f.tile < t_end) { fetch(pinA, pwtA, f.n, f.h0, f.w0, 0); advance(f); }
    if constexpr (DEPTH == 2) {
        if (f.tile < t_end) { fetch(pinB, pwtB, f.n, f.h0, f.w0, f.ck * CK); advance(f); }
    }
    auto stage = [&](uint4 (&pin)[NIS], uint4 (&pwt)[NWS]) __attribute__((always_inline)) {
        if (p.ck == 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                       // the previous stage's MFMAs are done with Is / Ws
#pragma unroll
        for (int j = 0; j < NIS; ++j) {
            const int idx = tid + kCT * j;
            if (idx < IH * IW * (CK / VEC)) *reinterpret_cast<uint4*>(Is + (idx / (CK / VEC)) * KP + (idx % (CK / VEC)) * VEC) = pin[j];
        }
#pragma unroll
        for (int j = 0; j < NWS; ++j) {
            const int idx = tid + kCT * j;
            if (idx < 9 * COT * (CK / VEC)) *reinterpret_cast<uint4*>(Ws + (idx / (CK / VEC)) * KP + (idx % (CK / VEC)) * VEC) = pwt[j];
        }
        __syncthreads();
        if (f.tile < t_end) { fetch(pin, pwt, f.n, f.h0, f.w0, f.ck * CK); advance(f); }      // this register set is free again: DEPTH stages ahead
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            typename MM::Frag bf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = MM::load(Wb + (tap * COT + t * 16) * KP, kq);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                typename MM::Frag af = MM::load(Ib + ((m / MTR + ky) * IW + (m % MTR) * 16 + kx) * KP, kq);
#pragma unroll
                for (int t = 0; t < NT; ++t) MM::mma_chunk(bf[t], af, acc[m][t]);   // D^T: rows = channels, cols = pixels
            }
        }
        if (p.ck == nck - 1) {
            // ---- epilogue of this tile (conv3x3_kernel's): a lane owns 4 consecutive channels of one pixel
            const int n = p.n, h0 = p.h0, w0 = p.w0;
            if (POOL) {
                static_assert(!POOL || RW % 2 == 0, "pooled epilogue: a wave owns whole row pairs");
                const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
                for (int mp = 0; mp < MTW / 2; ++mp) {
                    const int m0 = (2 * (mp / MTR)) * MTR + mp % MTR, m1 = m0 + MTR;
                    const int hp = ((h0 + wv * RW) >> 1) + mp / MTR, wp = ((w0 + (mp % MTR) * 16) >> 1) + (l15 >> 1);
                    const bool mine = !(l15 & 1) && hp < Hp && wp < Wp;
                    T* op = out + (((size_t)n * Hp + hp) * Wp + wp) * Cout;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int co = co0 + t * 16 + kq * 4;
                        T pk[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[m0][t][r] + acc[m1][t][r];
                            v += __shfl_xor(v, 1, 64);
                            pk[r] = from_f32<T>(v);
                        }
                        if (!mine) continue;
                        if (co + 3 < Cout) *reinterpret_cast<uint2*>(op + co) = *reinterpret_cast<const uint2*>(pk);
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co + r < Cout) op[co + r] = pk[r];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
                    const bool ok = h < H && w < W;
                    const size_t px = ((size_t)n * H + h) * W + w;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int co = co0 + t * 16 + kq * 4;
                        T* op = out + px * Cout;
                        if (fin.out1) op = co < fin.split ? out + px * fin.split : reinterpret_cast<T*>(fin.out1) + px * (Cout - fin.split) - fin.split;
                        if (ok && co + 3 < Cout) {
                            T pk[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[m][t][r];
                                pk[r] = from_f32<T>(v);
                                s1[t][r] += v;
                                s2[t][r] += v * v;
                            }
                            *reinterpret_cast<uint2*>(op + co) = *reinterpret_cast<const uint2*>(pk);
                        } else if (ok) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co + r < Cout) {
                                    const float v = acc[m][t][r];
                                    op[co + r] = from_f32<T>(v);
                                    s1[t][r] += v;
                                    s2[t][r] += v * v;
                                }
                        }
                    }
                }
            }
        }
        advance(p);
    };
    while (p.tile < t_end) {
        stage(pinA, pwtA);
        if constexpr (DEPTH == 2) {
            if (p.tile >= t_end) break;
            stage(pinB, pwtB);
        }
    }
    if (!POOL && stats) {          // one partial row per block
        __shared__ float sred[NW][2][COT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[t][r], b = s2[t][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    a += __shfl_xor(a, off, 64);
                    b += __shfl_xor(b, off, 64);
                }
                if (l15 == 0) { sred[wv][0][t * 16 + kq * 4 + r] = a; sred[wv][1][t * 16 + kq * 4 + r] = b; }
            }
        __syncthreads();
        if (tid < COT && co0 + tid < Cout) {
            float a = sred[0][0][tid] + sred[1][0][tid] + sred[2][0][tid] + sred[3][0][tid];
            float b = sred[0][1][tid] + sred[1][1][tid] + sred[2][1][tid] + sred[3][1][tid];
#pragma unroll
            for (int q = 4; q < NW; ++q) { a += sred[q][0][tid]; b += sred[q][1][tid]; }
            if (fin.acc) {
                bn_acc_add(fin.acc + co0 + tid, fin.acc + 2 * Cout, a);
                bn_acc_add(fin.acc + Cout + co0 + tid, fin.acc + 2 * Cout, b);
            } else {
                stats[((size_t)blockIdx.x * 2 + 0) * Cout + co0 + tid] = a;
                stats[((size_t)blockIdx.x * 2 + 1) * Cout + co0 + tid] = b;
            }
        }
    }
}

// Streaming variant for the HBM-bound layers (Cin <= 32, bf16): persistent blocks walk tiles; the next tile's haloed input
// (NVEC 16-byte channel vectors per pixel) is fetched into registers while the current tile is on the matrix cores, and the
// weight chunk is staged once per block.  Same arithmetic, tile shape and output as conv3x3_kernel.  These layers are
// instruction-issue bound, so everything tile-invariant is hoisted: per-slot byte offsets relative to a per-tile base (buffer
// loads with a scalar base; out-of-image lanes get an out-of-range offset and read 0), a branch-free epilogue for interior
// tiles, and BN statistics accumulated over all tiles of the block (one stats part per block: parts = gridDim.x).
// Logical block id is XCD-major (hardware deals consecutive workgroup ids round-robin to the 8 XCDs), so the tiles one
// XCD works on in a sweep are contiguous and their halos hit that XCD's L2.
// DUAL (concat of two sources): slots are vector-major so that one load instruction reads one source.
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// POOL: the output leaves 2x2 sum-pooled ([N, H/2, W/2, Cout]; H, W even) -- the backward of a convolution whose input was read
// through a nearest x2 upsample: its data gradient at full resolution (4x the bytes of what is wanted) is never written; the four
// fp32 accumulators of a 2x2 block are added (two registers of the lane, then its neighbour's sum) and rounded once.
// BNL / RED: as in conv3x3_kernel.  The loader keeps the raw output and the activation gradient of the next tile in registers and
// forms graw when the tile is committed to LDS (coefficients staged in LDS once per block); the epilogue reduce reads the producer's
// raw output at the tile's output positions after the tile has left the matrix cores and adds it up one iteration late, from the
// packed gradient registers, just before they are stored.
template <int COT, int TW, int NVEC, bool DUAL, bool POOL = false, bool BNL = false, bool RED = false, int RW = 4>
__global__ __launch_bounds__(kCT, 2) void conv3x3_stream_kernel(ConvSrc src, int N, int H, int W, const bf16* __restrict__ wpk, int Cout,
                                                               bf16* __restrict__ out, float* __restrict__ stats, int ntiles, BnFinish fin,
                                                               BnLoad bl, BnRed br) {
    typedef bf16 T;
    typedef Mma<T> MM;
    static_assert(!BNL || (!DUAL && (NVEC == 1 || NVEC == 2 || NVEC == 4)), "BN loader: one source, a thread keeps one channel vector");
    static_assert(!(RED && POOL), "the epilogue reduce exists for full-resolution outputs");
    // RW rows of a tile per wave: 4 (16-row tiles), or 2 for the fused forms with 32 output or 32 input channels, whose
    // accumulators, packed outputs and second operand stream do not fit 256 registers at 4
    constexpr int THS = 4 * RW;
    static_assert(RW == 2 || RW == 4, "rows per wave");
    constexpr int CKP = MM::CKP, VEC = MM::VEC, NT = COT / 16, MTR = TW / 16, MTW = RW * MTR;
    // pixel stride of the input tile: 48 elements (24 dwords) makes the 16-lane groups of ds_read_b128 conflict-free
    // (40 is 2-way); taken where the larger tile still leaves two blocks per CU
    constexpr int IKP = COT == 16 ? 48 : CKP;
    constexpr int IW = TW + 2, IH = THS + 2, NPIX = IH * IW;
    constexpr int NPF = DUAL ? NVEC * ((NPIX + kCT - 1) / kCT) : (NPIX * NVEC + kCT - 1) / kCT;
    constexpr unsigned OOB = 0xFFFFFF00u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Is = reinterpret_cast<T*>(smem);                    // [IH][IW][IKP]
    T* Ws = Is + IH * IW * IKP;                            // [9][COT][CKP]
    float* Cf = reinterpret_cast<float*>(Ws + 9 * COT * CKP);   // BNL: [6][CK] loader coefficients; RED: + [3][COT] mean | scale | shift
    float* Rf = Cf + (BNL ? kBwdCoefRows * CK : 0);
    const int Cin = src.C0 + src.C1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int tilesC = (W + TW - 1) / TW, tilesR = (H + THS - 1) / THS;
    const int co0 = blockIdx.y * COT;
    const int G = gridDim.x;
    const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (BNL)
        for (int idx = tid; idx < kBwdCoefRows * CK; idx += kCT) {
            const int r = idx / CK, c = idx % CK;
            Cf[idx] = c < Cin ? bl.coef[(size_t)r * Cin + c] : 0.f;
        }
    if (RED)
        for (int idx = tid; idx < 3 * COT; idx += kCT) {
            const int r = idx / COT, c = co0 + idx % COT;
            Rf[idx] = c < Cout ? br.saved[(size_t)(r == 0 ? 0 : r + 1) * Cout + c] : 0.f;
        }

    // channels [Cin, CK) of the tile stay zero for the whole kernel; the weight chunk is loaded once
    for (int idx = tid; idx < NPIX * (CK / VEC); idx += kCT) {
        const int v = idx % (CK / VEC), px = idx / (CK / VEC);
        if (v >= NVEC) zero_vec<T>(Is + px * IKP + v * VEC);
    }
    for (int idx = tid; idx < 9 * COT * (CK / VEC); idx += kCT) {
        const int v = idx % (CK / VEC), co = (idx / (CK / VEC)) % COT, tap = idx / ((CK / VEC) * COT);
        const int c = v * VEC;
        T* dst = Ws + (tap * COT + co) * CKP + v * VEC;
        if (co0 + co >= Cout || c >= Cin) { zero_vec<T>(dst); continue; }
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(wpk + ((size_t)tap * Cout + co0 + co) * Cin + c);
    }

    // ---- tile-invariant slot tables: rel = byte offset from the tile base (pixel (h0-1, w0-1) of the source, after its x2
    // upsampling shift), yx = iy | ix << 8 (0xFFFF for an unused slot), ldso = element offset into Is
    const int hs0 = H >> src.ups0, ws0 = W >> src.ups0, hs1 = H >> src.ups1, ws1 = W >> src.ups1;
    unsigned rel[NPF], yx[NPF];          // yx also carries ldso in its upper half (one register per slot instead of two)
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        int v, px;
        if (DUAL) { v = j % NVEC; px = tid + kCT * (j / NVEC); }
        else { const int idx = tid + kCT * j; v = idx % NVEC; px = idx / NVEC; }
        const int ix = px % IW, iy = px / IW, c = v * VEC;
        const bool used = px < NPIX;
        const bool from1 = c >= src.C0;
        const int ups = from1 ? src.ups1 : src.ups0, wsz = from1 ? ws1 : ws0, Cs = from1 ? src.C1 : src.C0, cs = from1 ? c - src.C0 : c;
        const int ry = ((iy - 1) >> ups) + 1, rx = ((ix - 1) >> ups) + 1;     // >= 0
        rel[j] = (unsigned)(((ry * wsz + rx) * Cs + cs) * 2);
        yx[j] = (used ? (unsigned)(iy | (ix << 8)) : 0xFFFFu) | ((unsigned)(used ? px * IKP + v * VEC : 0) << 16);
        static_assert(NPIX * IKP < 65536, "ldso must fit 16 bits");
    }

    if (BNL || RED) __syncthreads();                       // Cf / Rf are read before the loop's first barrier
    u32x4v pf[NPF];
    u32x4v pg[BNL ? NPF : 1];
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR, n = tile / (tilesC * tilesR);
        const int h0 = tr * THS, w0 = tc * TW;
        const int ylo = h0 == 0 ? 1 : 0, yhi = min(IH, H - h0 + 1), xlo = w0 == 0 ? 1 : 0, xhi = min(IW, W - w0 + 1);
        const long long b0 = ((((long long)n * hs0 + (h0 >> src.ups0) - 1) * ws0 + (w0 >> src.ups0) - 1) * src.C0) * 2;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src.p0 + b0), 0, 0x7FFFFFFF, 0x00020000);
        __amdgpu_buffer_rsrc_t r1 = r0;
        if (DUAL) {
            const long long b1 = ((((long long)n * hs1 + (h0 >> src.ups1) - 1) * ws1 + (w0 >> src.ups1) - 1) * src.C1) * 2;
            r1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src.p1 + b1), 0, 0x7FFFFFFF, 0x00020000);
        }
        if (BNL) r1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)bl.gy + b0), 0, 0x7FFFFFFF, 0x00020000);   // same shape as the source
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int iy = yx[j] & 0xFF, ix = (yx[j] >> 8) & 0xFF;
            const bool ok = iy >= ylo && iy < yhi && ix >= xlo && ix < xhi;     // an unused slot has ix = 255
            const unsigned off = ok ? rel[j] : OOB;
            if (DUAL && (j % NVEC) * VEC >= src.C0) pf[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)off, 0, 0);
            else pf[j] = __builtin_amdgcn_raw_buffer_load_b128(r0, (int)off, 0, 0);
            if (BNL) pg[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)off, 0, 0);
        }
    };
    // BNL: graw of the tile's slots (zero outside the image) from (pf, pg) and this thread's channel vector of coefficients
    auto to_graw = [&](int tile) __attribute__((always_inline)) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR;
        const int h0 = tr * THS, w0 = tc * TW;
        const int ylo = h0 == 0 ? 1 : 0, yhi = min(IH, H - h0 + 1), xlo = w0 == 0 ? 1 : 0, xhi = min(IW, W - w0 + 1);
        // two passes of four channels: 24 coefficient registers live instead of 48 (read from LDS here, every tile -- held across
        // the MFMA phase they would spill); the result replaces the raw output's half of pf in place
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            asm volatile("" ::: "memory");
            float4 cf[kBwdCoefRows];
#pragma unroll
            for (int r = 0; r < kBwdCoefRows; ++r) cf[r] = *reinterpret_cast<const float4*>(Cf + r * CK + (tid % NVEC) * VEC + 4 * hf);
            const float sc[4] = {cf[0].x, cf[0].y, cf[0].z, cf[0].w}, sh[4] = {cf[1].x, cf[1].y, cf[1].z, cf[1].w},
                        mu[4] = {cf[2].x, cf[2].y, cf[2].z, cf[2].w}, ca[4] = {cf[3].x, cf[3].y, cf[3].z, cf[3].w},
                        cp[4] = {cf[4].x, cf[4].y, cf[4].z, cf[4].w}, cq[4] = {cf[5].x, cf[5].y, cf[5].z, cf[5].w};
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int iy = yx[j] & 0xFF, ix = (yx[j] >> 8) & 0xFF;
                const bool ok = iy >= ylo && iy < yhi && ix >= xlo && ix < xhi;
                unsigned o2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned ur = pf[j][2 * hf + i], ug = pg[j][2 * hf + i];
                    float res[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float fr = bf16_bits_to_f32((unsigned short)(e ? ur >> 16 : ur & 0xffffu));
                        const float fg = bf16_bits_to_f32((unsigned short)(e ? ug >> 16 : ug & 0xffffu));
                        const int c = 2 * i + e;
                        const float lin = cp[c] + cq[c] * (fr - mu[c]);
                        res[e] = fr * sc[c] + sh[c] > relu_keep_threshold<T>() ? ca[c] * fg + lin : lin;
                    }
                    o2[i] = (unsigned)f32_to_bf16_bits(res[0]) | ((unsigned)f32_to_bf16_bits(res[1]) << 16);
                }
                pf[j][2 * hf] = ok ? o2[0] : 0u;
                pf[j][2 * hf + 1] = ok ? o2[1] : 0u;
            }
        }
    };

    // The epilogue of tile i is issued one iteration late (from packed registers), in front of the next loads: the
    // s_waitcnt vmcnt(0) guarding the register->LDS commit then only sees memory operations one MFMA phase old.
    constexpr int MPK = POOL ? MTW / 2 : MTW;            // POOL: (pooled row 0 / 1 of the wave's four rows) x (16-column block)
    uint2 pk[MPK][NT];
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[t][r] = s2[t][r] = 0.f;
    int ptile = -1;
    uint2 rp[RED ? MTW : 1][RED ? NT : 1];       // RED: the producer's raw output at the previous tile's output positions
    auto load_red = [&](int tile) __attribute__((always_inline)) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR, n = tile / (tilesC * tilesR);
        const int h0 = tr * THS, w0 = tc * TW;
        const T* rb = reinterpret_cast<const T*>(br.raw) + (((size_t)n * H + h0 + wv * RW) * W + w0 + l15) * Cout + co0 + kq * 4;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint2 q = make_uint2(0u, 0u);
                if (h < H && w < W && co0 + t * 16 + kq * 4 + 4 <= Cout)
                    q = *reinterpret_cast<const uint2*>(rb + ((size_t)(m / MTR) * W + (m % MTR) * 16) * Cout + t * 16);
                rp[m][t] = q;
            }
        }
    };
    auto store_prev = [&]() __attribute__((always_inline)) {
        const int tc = ptile % tilesC, tr = (ptile / tilesC) % tilesR, n = ptile / (tilesC * tilesR);
        const int h0 = tr * THS, w0 = tc * TW;
        const bool full = h0 + THS <= H && w0 + TW <= W && co0 + COT <= Cout;
        if (RED) {      // sums of dz = [y > 0] * (gradient as stored) and dz * (raw - mean) over the previous tile's valid outputs
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 mu = *reinterpret_cast<const float4*>(Rf + t * 16 + kq * 4);
                const float4 sc = *reinterpret_cast<const float4*>(Rf + COT + t * 16 + kq * 4);
                const float4 sh = *reinterpret_cast<const float4*>(Rf + 2 * COT + t * 16 + kq * 4);
                const float mua[4] = {mu.x, mu.y, mu.z, mu.w}, sca[4] = {sc.x, sc.y, sc.z, sc.w}, sha[4] = {sh.x, sh.y, sh.z, sh.w};
                const bool cok = co0 + t * 16 + kq * 4 + 4 <= Cout;
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
                    const bool ok = full || (h < H && w < W && cok);
                    const T* ge = reinterpret_cast<const T*>(&pk[m][t]);
                    const T* xe = reinterpret_cast<const T*>(&rp[m][t]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = to_f32(xe[r]);
                        const float dz = (ok && x * sca[r] + sha[r] > relu_keep_threshold<T>()) ? to_f32(ge[r]) : 0.f;
                        s1[t][r] += dz;
                        s2[t][r] += dz * (x - mua[r]);
                    }
                }
            }
        }
        if (POOL) {     // even lanes hold the sums of their 2x2 block: pooled pixel (h0/2 + 2 wv + pr, w0/2 + 8 mc + l15/2)
            const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
            for (int mp = 0; mp < MPK; ++mp) {
                const int hp = (h0 >> 1) + wv * (RW / 2) + mp / MTR, wp = (w0 >> 1) + (mp % MTR) * 8 + (l15 >> 1);
                if ((l15 & 1) || hp >= Hp || wp >= Wp) continue;
                T* op = out + (((size_t)n * Hp + hp) * Wp + wp) * Cout + co0 + kq * 4;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (fin.accumulate_out) {   // the output already holds another consumer's gradient of this tensor (ops._GradJoin)
                        const T* e = reinterpret_cast<const T*>(&pk[mp][t]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + t * 16 + kq * 4 + r < Cout) op[t * 16 + r] = from_f32<T>(to_f32(op[t * 16 + r]) + to_f32(e[r]));
                    } else if (co0 + t * 16 + kq * 4 + 4 <= Cout) *reinterpret_cast<uint2*>(op + t * 16) = pk[mp][t];
                    else {
                        const T* e = reinterpret_cast<const T*>(&pk[mp][t]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + t * 16 + kq * 4 + r < Cout) op[t * 16 + r] = e[r];
                    }
                }
            }
            return;
        }
        // per 16-channel tile t: where its lanes' 4 channels go and the pixel stride there (two destinations: fin.out1, see BnFinish;
        // the split is a multiple of 16, so a tile never straddles it)
        const size_t px0 = ((size_t)n * H + h0 + wv * RW) * W + w0 + l15;
        T* ot[NT];
        int ost[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = co0 + t * 16 + kq * 4;
            if (!fin.out1) { ot[t] = out + px0 * Cout + c; ost[t] = Cout; }
            else if (c < fin.split) { ot[t] = out + px0 * fin.split + c; ost[t] = fin.split; }
            else { ot[t] = reinterpret_cast<T*>(fin.out1) + px0 * (Cout - fin.split) + c - fin.split; ost[t] = Cout - fin.split; }
        }
        if (full) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<uint2*>(ot[t] + ((size_t)(m / MTR) * W + (m % MTR) * 16) * ost[t]) = pk[m][t];
        } else {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
                if (h < H && w < W) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const T* e = reinterpret_cast<const T*>(&pk[m][t]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + t * 16 + kq * 4 + r < Cout) ot[t][((size_t)(m / MTR) * W + (m % MTR) * 16) * ost[t] + r] = e[r];
                    }
                }
            }
        }
    };
    if (lb < ntiles) fetch(lb);
    for (int tile = lb; tile < ntiles; tile += G) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR;
        const int h0 = tr * THS, w0 = tc * TW;
        if (BNL) to_graw(tile);                // in registers, before the barrier: overlaps the other waves' last MFMAs
        __syncthreads();                       // previous tile's MFMAs are done with Is
#pragma unroll
        for (int j = 0; j < NPF; ++j)
            if ((yx[j] & 0xFFFFu) != 0xFFFFu) *reinterpret_cast<u32x4v*>(Is + (yx[j] >> 16)) = pf[j];
        __syncthreads();
        if (ptile >= 0) store_prev();
        if (tile + G < ntiles) fetch(tile + G);

        f32x4 acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* Ib = Is + ((wv * RW) * IW + l15) * IKP + 8 * kq;   // all 72 A-fragment reads are this base + an immediate
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            typename MM::Frag bf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = MM::load(Ws + (tap * COT + t * 16 + l15) * CKP, kq);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                typename MM::Frag af = *reinterpret_cast<const typename MM::Frag*>(Ib + ((m / MTR + ky) * IW + (m % MTR) * 16 + kx) * IKP);
#pragma unroll
                for (int t = 0; t < NT; ++t) MM::mma_chunk(bf[t], af, acc[m][t]);
            }
        }
        // D^T[row = channel t*16 + kq*4 + r][col = pixel l15]: pack to bf16, accumulate the BN statistics of valid outputs
        const bool full = h0 + THS <= H && w0 + TW <= W && co0 + COT <= Cout;
        if (POOL) {
#pragma unroll
            for (int mp = 0; mp < MPK; ++mp) {
                const int m0 = (2 * (mp / MTR)) * MTR + mp % MTR, m1 = m0 + MTR;       // rows 2 pr and 2 pr + 1, same column block
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    T e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[m0][t][r] + acc[m1][t][r];
                        v += __shfl_xor(v, 1, 64);                                    // the column neighbour (lane l15 ^ 1)
                        e[r] = from_f32<T>(v);
                    }
                    pk[mp][t] = *reinterpret_cast<const uint2*>(e);
                }
            }
            ptile = tile;
            continue;
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int h = h0 + wv * RW + m / MTR, w = w0 + (m % MTR) * 16 + l15;
            const bool ok = full || (h < H && w < W);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                T e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[m][t][r];
                    e[r] = from_f32<T>(v);
                    if (!RED) {
                        const float vv = (full || (ok && co0 + t * 16 + kq * 4 + r < Cout)) ? v : 0.f;
                        s1[t][r] += vv;
                        s2[t][r] += vv * vv;
                    }
                }
                pk[m][t] = *reinterpret_cast<const uint2*>(e);
            }
        }
        ptile = tile;
        if (RED) load_red(tile);               // in flight across the next commit, consumed by store_prev
    }
    if (ptile >= 0) store_prev();
    if (RED) stats = br.parts;
    if (stats) {   // one part per block (zero for a block without tiles)
        __shared__ float sred[4][2][COT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[t][r], b = s2[t][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    a += __shfl_xor(a, off, 64);
                    b += __shfl_xor(b, off, 64);
                }
                if (l15 == 0) { sred[wv][0][t * 16 + kq * 4 + r] = a; sred[wv][1][t * 16 + kq * 4 + r] = b; }
            }
        __syncthreads();
        if (tid < COT && co0 + tid < Cout) {
            float b = sred[0][1][tid] + sred[1][1][tid] + sred[2][1][tid] + sred[3][1][tid];
            if (RED) b *= br.saved[Cout + co0 + tid];          // sum dz * (raw - mean) -> sum dz * xhat
            const float a = sred[0][0][tid] + sred[1][0][tid] + sred[2][0][tid] + sred[3][0][tid];
            if (!RED && fin.acc) {
                bn_acc_add(fin.acc + co0 + tid, fin.acc + 2 * Cout, a);
                bn_acc_add(fin.acc + Cout + co0 + tid, fin.acc + 2 * Cout, b);
            } else {
                store_part(&stats[((size_t)blockIdx.x * 2 + 0) * Cout + co0 + tid], a);
                store_part(&stats[((size_t)blockIdx.x * 2 + 1) * Cout + co0 + tid], b);
            }
        }
        if (!RED && fin.counter && last_block_arrives(fin.counter, gridDim.x * gridDim.y)) {
            bn_finish_block(stats, (int)gridDim.x, Cout, fin, reinterpret_cast<float*>(smem));
            last_block_done(fin.counter);
        }
    }
}

// OIHW fp32 -> packed [tap][n][k] of T.  kind 0 (forward): n = o, k = i, tap = ky*3+kx.
// kind 1 (dgrad): n = i - ci_begin (ci_count of them), k = o, tap mirrored (2-ky, 2-kx).
template <typename T>
__global__ void pack_w_kernel(const float* __restrict__ w, int Cout, int Cin, int kind, int ci_begin, int ci_count, T* __restrict__ pk) {
    // kind >> 8 = input channels the SOURCE tensor really has (0: Cin): the stem's [Cout, 1, 3, 3] weight packed as Cin = one channel
    // vector, the extra channels zero (forward layout only) -- the host no longer pads the weight with torch ops every step
    const int csrc = (kind >> 8) ? (kind >> 8) : Cin;
    kind &= 0xff;
    const int Nn = kind ? ci_count : Cout, Kk = kind ? Cout : Cin;
    const int total = 9 * Nn * Kk;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k = e % Kk, nn = (e / Kk) % Nn, tap = e / (Kk * Nn);
        const int ky = tap / 3, kx = tap % 3;
        float v;
        if (!kind) v = k < csrc ? w[(((size_t)nn * csrc + k) * 3 + ky) * 3 + kx] : 0.f;
        else v = w[(((size_t)k * Cin + ci_begin + nn) * 3 + (2 - ky)) * 3 + (2 - kx)];
        pk[e] = from_f32<T>(v);
    }
}

// All conv weights of a step in one launch: job j packs like pack_w_kernel; blocks [first_block[j], first_block[j+1]) belong to it.
struct PackJob {            // mirrors miseg_pack_job (include/miseg_hip.h)
    const float* w;
    void* packed;
    int32_t Cout, Cin, kind, ci_begin, ci_count, first_block;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_w_multi_kernel(const PackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                       // last job whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    PackJob jb = jobs[lo];
    const int csrc = (jb.kind >> 8) ? (jb.kind >> 8) : jb.Cin;      // see pack_w_kernel
    jb.kind &= 0xff;
    const int Nn = jb.kind ? jb.ci_count : jb.Cout, Kk = jb.kind ? jb.Cout : jb.Cin, total = 9 * Nn * Kk;
    const int e = ((int)blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (e >= total) return;
    const int k = e % Kk, nn = (e / Kk) % Nn, tap = e / (Kk * Nn), ky = tap / 3, kx = tap % 3;
    const float v = !jb.kind ? (k < csrc ? jb.w[(((size_t)nn * csrc + k) * 3 + ky) * 3 + kx] : 0.f)
                             : jb.w[(((size_t)k * jb.Cin + jb.ci_begin + nn) * 3 + (2 - ky)) * 3 + (2 - kx)];
    reinterpret_cast<T*>(jb.packed)[e] = from_f32<T>(v);
}

// ------------------------------------------------------------------------------------------ wgrad
// gw[co][ci][ky][kx] = sum_{n,h,w} gout[n,h,w,co] * in[n,h+ky-1,w+kx-1,ci]   (fp32 MFMA, split over pixels)
// block = (co tile 32, ci chunk 32, split); wave = 2 of the tile's 8 rows; acc 2 x 18 tiles.
constexpr int WG_TH = 8, WG_TW = 32, WG_P = 34;  // tile rows / cols, LDS channel stride (32 + 2)

// BNL (all three weight-gradient kernels): `gout` is the layer's raw convolution output and bl.gy the gradient of its activation; the
// loader forms graw (common.h, bn_graw_vec) -- positions outside the image stay zero.
template <typename T, bool BNL = false>
__global__ __launch_bounds__(kCT, 1) void conv3x3_wgrad_kernel(ConvSrc src, int N, int H, int W, const T* __restrict__ gout, int Cout,
                                                              int nsplit, float* __restrict__ partials, BnLoad bl) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* Gs = wsm;                                   // [WG_TH*WG_TW][WG_P]
    float* Is = wsm + WG_TH * WG_TW * WG_P;            // [(WG_TH+2)*(WG_TW+2)][WG_P]
    const int Cin = src.C0 + src.C1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int tilesC = (W + WG_TW - 1) / WG_TW, tilesR = (H + WG_TH - 1) / WG_TH;
    const int ntiles = N * tilesR * tilesC;
    constexpr int IW = WG_TW + 2;
    f32x4 acc[2][18];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 18; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bc[kBwdCoefRows] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // BNL: this thread's channel (tid & 31) in every element it loads
    if (BNL && co0 + (tid & 31) < Cout)
#pragma unroll
        for (int r = 0; r < kBwdCoefRows; ++r) bc[r] = bl.coef[(size_t)r * Cout + co0 + (tid & 31)];

    for (int tile = split; tile < ntiles; tile += nsplit) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR, n = tile / (tilesC * tilesR);
        const int h0 = tr * WG_TH, w0 = tc * WG_TW;
        __syncthreads();
        for (int idx = tid; idx < WG_TH * WG_TW * 32; idx += kCT) {
            const int c = idx & 31, px = idx >> 5, ix = px % WG_TW, iy = px / WG_TW;
            const int h = h0 + iy, w = w0 + ix;
            float v = 0.f;
            if (h < H && w < W && co0 + c < Cout) {
                const size_t o = (((size_t)n * H + h) * W + w) * Cout + co0 + c;
                v = to_f32(gout[o]);
                if (BNL) {
                    const float g = to_f32(reinterpret_cast<const T*>(bl.gy)[o]);
                    const float lin = bc[4] + bc[5] * (v - bc[2]);
                    v = v * bc[0] + bc[1] > relu_keep_threshold<T>() ? bc[3] * g + lin : lin;
                }
            }
            Gs[px * WG_P + c] = v;
        }
        for (int idx = tid; idx < (WG_TH + 2) * IW * 32; idx += kCT) {
            const int c = idx & 31, px = idx >> 5, ix = px % IW, iy = px / IW;
            const int h = h0 - 1 + iy, w = w0 - 1 + ix, cc = ci0 + c;
            float v = 0.f;
            if (h >= 0 && h < H && w >= 0 && w < W && cc < Cin) {
                if (cc < src.C0) {
                    const int hs = H >> src.ups0, wsz = W >> src.ups0;
                    v = to_f32(reinterpret_cast<const T*>(src.p0)[(((size_t)n * hs + (h >> src.ups0)) * wsz + (w >> src.ups0)) * src.C0 + cc]);
                } else {
                    const int hs = H >> src.ups1, wsz = W >> src.ups1;
                    v = to_f32(reinterpret_cast<const T*>(src.p1)[(((size_t)n * hs + (h >> src.ups1)) * wsz + (w >> src.ups1)) * src.C1 + cc - src.C0]);
                }
            }
            Is[px * WG_P + c] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int row = wv * 2 + rr;
#pragma unroll 1
            for (int cs = 0; cs < WG_TW; cs += 4) {
                const float a0 = Gs[(row * WG_TW + cs + kq) * WG_P + l15];
                const float a1 = Gs[(row * WG_TW + cs + kq) * WG_P + 16 + l15];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    const float* ip = Is + ((row + ky) * IW + cs + kq + kx) * WG_P + l15;
                    const float b0 = ip[0], b1 = ip[16];
                    acc[0][tap * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][tap * 2 + 0], 0, 0, 0);
                    acc[0][tap * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][tap * 2 + 1], 0, 0, 0);
                    acc[1][tap * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][tap * 2 + 0], 0, 0, 0);
                    acc[1][tap * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][tap * 2 + 1], 0, 0, 0);
                }
            }
        }
    }
    // reduce the 4 waves in LDS (fixed order), write partial [split][co 32][tap 9][ci 32]
    float* Ds = wsm;  // 32 x 288 floats
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 18; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = a * 16 + kq * 4 + r, col = (b / 2) * 32 + (b % 2) * 16 + l15;
                        if (w == 0) Ds[co * 288 + col] = acc[a][b][r];
                        else Ds[co * 288 + col] += acc[a][b][r];
                    }
        }
    }
    __syncthreads();
    float* outp = partials + (((size_t)split * gridDim.z + blockIdx.z) * gridDim.y + blockIdx.y) * (32 * 288);
    for (int e = tid; e < 32 * 288; e += kCT) outp[e] = Ds[e];
}

// bf16 variant: both operands need 8 consecutive PIXELS per lane for a fixed channel, while NHWC keeps channels
// contiguous -> the LDS tiles stay in their natural [pixel][16 channels] shape (32-byte rows: the 8 rows a 32-lane
// half touches tile the 64 banks exactly) and fragments come out of ds_read_b64_tr_b16 (hardware 4x16 transpose):
// lane 4q+p of a 16-lane group supplies (pixel 8g+q, channels 4p..4p+3) and receives its channel's 4 pixels.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_S16X4(ptr) ((__attribute__((address_space(3))) s16x4*)(ptr))

// The same for a 32-pixel k-step of v_mfma_f32_16x16x32_bf16 with the k <-> pixel map  k = 8 kq + e  <->  pixel 16 (e >> 2) + 4 kq + (e & 3)
// (any bijection does, A and B share it): per transpose read the two lane groups of a half-wave (kq = 0, 1 or 2, 3) cover 8
// CONSECUTIVE pixels x 32 B = every bank once.  With pixel = 8 kq + e they read pixels 0-3 and 8-11, 256 B apart: a 2-way conflict
// on every read (38 % of conv3x3_wgrad_bf16_c16_kernel's LDS cycles, profiles/r02_pmc.json).
__device__ __forceinline__ bf16x8_t tr_frag32(const short* row_px0, int kq, int q, int p) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row_px0 + (4 * kq + q) * 16 + 4 * p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row_px0 + (16 + 4 * kq + q) * 16 + 4 * p));
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, f);
}

__device__ __forceinline__ bf16x8_t tr_frag(const short* blk_px0, int q, int p) {
    // blk_px0: first pixel (of this lane group's 8) of a [pixel][16] sub-tile
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(blk_px0 + q * 16 + 4 * p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(blk_px0 + (4 + q) * 16 + 4 * p));
    s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, f);
}

template <bool BNL>
__global__ __launch_bounds__(kCT, 2) void conv3x3_wgrad_bf16_kernel(ConvSrc src, int N, int H, int W, const bf16* __restrict__ gout,
                                                                   int Cout, int nsplit, float* __restrict__ partials, BnLoad bl) {
    constexpr int IW = WG_TW + 2, NPX = WG_TH * WG_TW, NIPX = (WG_TH + 2) * IW;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm_raw[];
    // The two 16-channel sub-tiles of a half-wave's transpose read (lanes 0-15 | 16-31) must land in different 128-byte bank windows:
    // NPX * 32 B = 8 KB is a multiple of the 256-byte bank period, so the second sub-tile of Gs starts 128 B late (GSUB); NIPX * 32 B
    // = 10 880 B already is 128 (mod 256).  Unpadded, 31 % of this kernel's LDS cycles were 2-way conflicts (profiles/r02_pmc.json).
    constexpr int GSUB = NPX * 16 + 64;
    short* Gs = reinterpret_cast<short*>(wsm_raw);            // [2][GSUB]: [NPX][16] twice
    short* Is = Gs + 2 * GSUB;                                // [2][NIPX][16]
    static_assert((NIPX * 32) % 256 == 128, "Is sub-tiles: re-check the bank offset for this tile shape");
    const int Cin = src.C0 + src.C1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int g8 = 8 * kq, q = (lane >> 2) & 3, p4 = lane & 3;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int tilesC = (W + WG_TW - 1) / WG_TW, tilesR = (H + WG_TH - 1) / WG_TH;
    const int ntiles = N * tilesR * tilesC;
    // v_mfma_f32_32x32x16_bf16: one 32 co x 32 ci tile per tap (16 fp32 per lane), k = 16 pixels per step.  Lane map:
    // operand row/col = lane & 31, k-half (8 pixels) = lane >> 5; the 16-lane groups of ds_read_b64_tr_b16 are therefore
    // (channels 0-15 | 16-31) x (pixels 0-7 | 8-15) = (g16 & 1, g16 >> 1).  Half the LDS bytes per flop of 16x16x32.
    // Work split over the 4 waves BY TAP: wave w owns taps 2w and 2w+1 for all 8 rows of a tile, and the two rows 2w, 2w+1 of
    // tap 8.  A tap's accumulator then lives in exactly one wave: taps 0-7 leave through one parallel LDS write, only tap 8
    // (1/9 of the tile) is summed over the waves.  With rows split over the waves instead, all 9 x 16 accumulator registers of
    // every wave went through four serial read-modify-write rounds per block -- a quarter of the kernel's time (measured).
    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int tap0 = 2 * wvu, tap1 = 2 * wvu + 1;
    const int off0 = ((tap0 / 3) * IW + tap0 % 3) * 16, off1 = ((tap1 / 3) * IW + tap1 % 3) * 16, off8 = (2 * IW + 2) * 16;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int g16 = lane >> 4, chh = g16 & 1, pxh = g16 >> 1;

    // next tile's operands travel global -> registers while the current tile is on the matrix cores
    constexpr int NG = NPX * 4 / kCT, NI = (NIPX * 4 + kCT - 1) / kCT;
    uint4 pg[NG], pi[NI];
    uint4 pr[BNL ? NG : 1];                                 // BNL: pg = activation gradient, pr = raw output, okg = slots inside the image
    unsigned okg = 0;
    float* Cf = reinterpret_cast<float*>(Is + 2 * NIPX * 16);   // BNL: [6][32] loader coefficients behind the operand tiles
    if (BNL) {
        for (int idx = tid; idx < kBwdCoefRows * 32; idx += kCT) Cf[idx] = co0 + (idx & 31) < Cout ? bl.coef[(size_t)(idx >> 5) * Cout + co0 + (idx & 31)] : 0.f;
        __syncthreads();
    }
    // A thread's slots of the two operand tiles never change: which pixel of the tile, which 8 channels, which source tensor.  Worked out
    // once -- per tile only the tile's corner moves (scalar) and two range tests per slot remain.  (Derived per tile, the index math was
    // ~10 vector instructions per MFMA: profiles/r04_pmc.json, SQ_INSTS_VALU / SQ_INSTS_MFMA of this kernel.)
    int g_iy[NG], g_ix[NG], g_rel[NG];                    // activation-gradient tile: row, column, element offset from the tile's corner (< 0: no such channel)
    int i_iy[NI], i_ix[NI], i_rel[NI], i_src[NI];         // input tile with halo: row - 1, column - 1, offset inside its source, source (0 | 1; -1: no slot)
    {
        const int hs0 = H >> src.ups0, ws0 = W >> src.ups0, ws1 = W >> src.ups1;
        (void)hs0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + kCT * j, v = idx & 3, px = idx >> 2, c = co0 + v * 8;
            g_ix[j] = px % WG_TW, g_iy[j] = px / WG_TW;
            g_rel[j] = c < Cout ? (g_iy[j] * W + g_ix[j]) * Cout + c : -1;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + kCT * j, v = idx & 3, px = idx >> 2, cc = ci0 + v * 8;
            i_ix[j] = px % IW - 1, i_iy[j] = px / IW - 1;
            const bool slot = idx < NIPX * (3 + 1) && cc < Cin, first = cc < src.C0;
            i_src[j] = slot ? (first ? 0 : 1) : -1;
            // (h0 + d) >> ups = (h0 >> ups) + (d >> ups): tile corners are multiples of 8 x 32, the shift is arithmetic (d = -1 stays outside)
            i_rel[j] = first ? ((i_iy[j] >> src.ups0) * ws0 + (i_ix[j] >> src.ups0)) * src.C0 + cc
                             : ((i_iy[j] >> src.ups1) * ws1 + (i_ix[j] >> src.ups1)) * src.C1 + cc - src.C0;
        }
    }
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR, n = tile / (tilesC * tilesR);
        const int h0 = tr * WG_TH, w0 = tc * WG_TW;
        const bf16* gcorner = gout + (((size_t)n * H + h0) * W + w0) * Cout;
        const bf16* ycorner = BNL ? reinterpret_cast<const bf16*>(bl.gy) + (((size_t)n * H + h0) * W + w0) * Cout : nullptr;
        const bf16* s0 = reinterpret_cast<const bf16*>(src.p0) + (((size_t)n * (H >> src.ups0) + (h0 >> src.ups0)) * (W >> src.ups0) + (w0 >> src.ups0)) * src.C0;
        const bf16* s1 = reinterpret_cast<const bf16*>(src.p1) + (((size_t)n * (H >> src.ups1) + (h0 >> src.ups1)) * (W >> src.ups1) + (w0 >> src.ups1)) * src.C1;
        okg = 0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            uint4 val = zero4;
            if (g_rel[j] >= 0 && g_iy[j] < H - h0 && g_ix[j] < W - w0) {
                if (BNL) { pr[j] = *reinterpret_cast<const uint4*>(gcorner + g_rel[j]); val = *reinterpret_cast<const uint4*>(ycorner + g_rel[j]); okg |= 1u << j; }
                else val = *reinterpret_cast<const uint4*>(gcorner + g_rel[j]);
            }
            pg[j] = val;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            uint4 val = zero4;
            if (i_src[j] >= 0 && (unsigned)(h0 + i_iy[j]) < (unsigned)H && (unsigned)(w0 + i_ix[j]) < (unsigned)W)
                val = *reinterpret_cast<const uint4*>((i_src[j] ? s1 : s0) + i_rel[j]);
            pi[j] = val;
        }
    };
    if (split < ntiles) fetch(split);
    for (int tile = split; tile < ntiles; tile += nsplit) {
        __syncthreads();
        if (BNL) {
            asm volatile("" ::: "memory");
            float cf[kBwdCoefRows][8];
#pragma unroll
            for (int r = 0; r < kBwdCoefRows; ++r)
#pragma unroll
                for (int i = 0; i < 8; i += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(Cf + r * 32 + (tid & 3) * 8 + i);
                    cf[r][i] = q.x; cf[r][i + 1] = q.y; cf[r][i + 2] = q.z; cf[r][i + 3] = q.w;
                }
#pragma unroll
            for (int j = 0; j < NG; ++j)
                pg[j] = (okg >> j) & 1u ? bn_graw_vec<bf16>(pr[j], pg[j], cf[0], cf[1], cf[2], cf[3], cf[4], cf[5]) : zero4;
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + kCT * j, v = idx & 3, px = idx >> 2;
            *reinterpret_cast<uint4*>(Gs + (v >> 1) * GSUB + px * 16 + (v & 1) * 8) = pg[j];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + kCT * j, v = idx & 3, px = idx >> 2;
            if (idx < NIPX * 4) *reinterpret_cast<uint4*>(Is + ((v >> 1) * NIPX + px) * 16 + (v & 1) * 8) = pi[j];
        }
        __syncthreads();
        if (tile + nsplit < ntiles) fetch(tile + nsplit);
#pragma unroll 2
        for (int row = 0; row < WG_TH; ++row) {
            const short* gb = Gs + chh * GSUB + (row * WG_TW + 8 * pxh) * 16;
            const bf16x8_t a0 = tr_frag(gb, q, p4), a1 = tr_frag(gb + 16 * 16, q, p4);     // pixels 0-15 | 16-31 of the row
            const short* ib = Is + (chh * NIPX + row * IW + 8 * pxh) * 16;
            {
                const bf16x8_t b0 = tr_frag(ib + off0, q, p4), b1 = tr_frag(ib + off0 + 16 * 16, q, p4);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[0], 0, 0, 0);
            }
            {
                const bf16x8_t b0 = tr_frag(ib + off1, q, p4), b1 = tr_frag(ib + off1 + 16 * 16, q, p4);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1], 0, 0, 0);
            }
            if ((row >> 1) == wvu) {
                const bf16x8_t b0 = tr_frag(ib + off8, q, p4), b1 = tr_frag(ib + off8 + 16 * 16, q, p4);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[2], 0, 0, 0);
            }
        }
    }
    // partial [split][co 32][tap 9][ci 32] through LDS.  D[row = co = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = ci = lane & 31]
    float* Ds = reinterpret_cast<float*>(wsm_raw);  // [32][288]; then the four waves' tap-8 planes [4][32][32]
    float* D8 = Ds + 32 * 288;
    __syncthreads();                                // the operand tiles are dead
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = lane & 31;
        Ds[co * 288 + tap0 * 32 + ci] = acc[0][r];
        Ds[co * 288 + tap1 * 32 + ci] = acc[1][r];
        D8[wvu * 1024 + co * 32 + ci] = acc[2][r];
    }
    __syncthreads();
    float* outp = partials + (((size_t)split * gridDim.z + blockIdx.z) * gridDim.y + blockIdx.y) * (32 * 288);
    for (int e = tid; e < 32 * 288; e += kCT) {
        const int co = e / 288, col = e - co * 288;
        float v = Ds[e];
        if (col >= 256) {
            const int i8 = co * 32 + col - 256;
            v = (D8[i8] + D8[1024 + i8]) + (D8[2048 + i8] + D8[3072 + i8]);
        }
        outp[e] = v;
    }
}

// Narrow-layer variant (the 256^2 / 128^2 layers that are HBM-bound): a block owns 16 output channels (grid.z) x one
// 16-channel slice of the input (grid.y), v_mfma_f32_16x16x32_bf16 (k = the 32 pixels of a tile row), 36 accumulator
// registers instead of 144 and 19 KB of LDS instead of 38 -> ~5 blocks per CU keep enough loads in flight to stream.
// Writes the same partial layout as conv3x3_wgrad_bf16_kernel (its 16 x 9 x 16 corner of the 32 x 288 tile).
template <bool BNL>
__global__ __launch_bounds__(kCT, BNL ? 3 : 4) void conv3x3_wgrad_bf16_c16_kernel(ConvSrc src, int N, int H, int W, const bf16* __restrict__ gout,
                                                                       int Cout, int nsplit, float* __restrict__ partials, BnLoad bl) {
    constexpr int IW = WG_TW + 2, NPX = WG_TH * WG_TW, NIPX = (WG_TH + 2) * IW;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm_raw[];
    short* Gs = reinterpret_cast<short*>(wsm_raw);            // [NPX][16]
    short* Is = Gs + NPX * 16;                                // [NIPX][16]
    const int Cin = src.C0 + src.C1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int q = (lane >> 2) & 3, p4 = lane & 3;
    const int split = blockIdx.x, ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16;
    const int tilesC = (W + WG_TW - 1) / WG_TW, tilesR = (H + WG_TH - 1) / WG_TH;
    const int ntiles = N * tilesR * tilesC;
    f32x4 acc[9];
#pragma unroll
    for (int b = 0; b < 9; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    constexpr int NG = NPX * 2 / kCT, NI = (NIPX * 2 + kCT - 1) / kCT;
    uint4 pg[NG], pi[NI];
    uint4 pr[BNL ? NG : 1];
    unsigned okg = 0;
    float* Cf = reinterpret_cast<float*>(Is + NIPX * 16);       // BNL: [6][16] loader coefficients behind the operand tiles
    if (BNL) {
        for (int idx = tid; idx < kBwdCoefRows * 16; idx += kCT) Cf[idx] = co0 + (idx & 15) < Cout ? bl.coef[(size_t)(idx >> 4) * Cout + co0 + (idx & 15)] : 0.f;
        __syncthreads();
    }
    // tile-invariant slot geometry, as in conv3x3_wgrad_bf16_kernel
    int g_iy[NG], g_ix[NG], g_rel[NG];                    // activation-gradient tile: row, column, element offset from the tile's corner (< 0: no such channel)
    int i_iy[NI], i_ix[NI], i_rel[NI], i_src[NI];         // input tile with halo: row - 1, column - 1, offset inside its source, source (0 | 1; -1: no slot)
    {
        const int hs0 = H >> src.ups0, ws0 = W >> src.ups0, ws1 = W >> src.ups1;
        (void)hs0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + kCT * j, v = idx & 1, px = idx >> 1, c = co0 + v * 8;
            g_ix[j] = px % WG_TW, g_iy[j] = px / WG_TW;
            g_rel[j] = c < Cout ? (g_iy[j] * W + g_ix[j]) * Cout + c : -1;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + kCT * j, v = idx & 1, px = idx >> 1, cc = ci0 + v * 8;
            i_ix[j] = px % IW - 1, i_iy[j] = px / IW - 1;
            const bool slot = idx < NIPX * (1 + 1) && cc < Cin, first = cc < src.C0;
            i_src[j] = slot ? (first ? 0 : 1) : -1;
            // (h0 + d) >> ups = (h0 >> ups) + (d >> ups): tile corners are multiples of 8 x 32, the shift is arithmetic (d = -1 stays outside)
            i_rel[j] = first ? ((i_iy[j] >> src.ups0) * ws0 + (i_ix[j] >> src.ups0)) * src.C0 + cc
                             : ((i_iy[j] >> src.ups1) * ws1 + (i_ix[j] >> src.ups1)) * src.C1 + cc - src.C0;
        }
    }
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int tc = tile % tilesC, tr = (tile / tilesC) % tilesR, n = tile / (tilesC * tilesR);
        const int h0 = tr * WG_TH, w0 = tc * WG_TW;
        const bf16* gcorner = gout + (((size_t)n * H + h0) * W + w0) * Cout;
        const bf16* ycorner = BNL ? reinterpret_cast<const bf16*>(bl.gy) + (((size_t)n * H + h0) * W + w0) * Cout : nullptr;
        const bf16* s0 = reinterpret_cast<const bf16*>(src.p0) + (((size_t)n * (H >> src.ups0) + (h0 >> src.ups0)) * (W >> src.ups0) + (w0 >> src.ups0)) * src.C0;
        const bf16* s1 = reinterpret_cast<const bf16*>(src.p1) + (((size_t)n * (H >> src.ups1) + (h0 >> src.ups1)) * (W >> src.ups1) + (w0 >> src.ups1)) * src.C1;
        okg = 0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            uint4 val = zero4;
            if (g_rel[j] >= 0 && g_iy[j] < H - h0 && g_ix[j] < W - w0) {
                if (BNL) { pr[j] = *reinterpret_cast<const uint4*>(gcorner + g_rel[j]); val = *reinterpret_cast<const uint4*>(ycorner + g_rel[j]); okg |= 1u << j; }
                else val = *reinterpret_cast<const uint4*>(gcorner + g_rel[j]);
            }
            pg[j] = val;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            uint4 val = zero4;
            if (i_src[j] >= 0 && (unsigned)(h0 + i_iy[j]) < (unsigned)H && (unsigned)(w0 + i_ix[j]) < (unsigned)W)
                val = *reinterpret_cast<const uint4*>((i_src[j] ? s1 : s0) + i_rel[j]);
            pi[j] = val;
        }
    };
    if (split < ntiles) fetch(split);
    for (int tile = split; tile < ntiles; tile += nsplit) {
        __syncthreads();
        if (BNL) {
            asm volatile("" ::: "memory");
            float cf[kBwdCoefRows][8];
#pragma unroll
            for (int r = 0; r < kBwdCoefRows; ++r)
#pragma unroll
                for (int i = 0; i < 8; i += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(Cf + r * 16 + (tid & 1) * 8 + i);
                    cf[r][i] = q.x; cf[r][i + 1] = q.y; cf[r][i + 2] = q.z; cf[r][i + 3] = q.w;
                }
#pragma unroll
            for (int j = 0; j < NG; ++j)
                pg[j] = (okg >> j) & 1u ? bn_graw_vec<bf16>(pr[j], pg[j], cf[0], cf[1], cf[2], cf[3], cf[4], cf[5]) : zero4;
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + kCT * j;
            *reinterpret_cast<uint4*>(Gs + (idx >> 1) * 16 + (idx & 1) * 8) = pg[j];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + kCT * j;
            if (idx < NIPX * 2) *reinterpret_cast<uint4*>(Is + (idx >> 1) * 16 + (idx & 1) * 8) = pi[j];
        }
        __syncthreads();
        if (tile + nsplit < ntiles) fetch(tile + nsplit);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = wv * 2 + rr;
            const bf16x8_t a0 = tr_frag32(Gs + row * WG_TW * 16, kq, q, p4);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const bf16x8_t b0 = tr_frag32(Is + ((row + ky) * IW + kx) * 16, kq, q, p4);
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[tap], 0, 0, 0);
            }
        }
    }
    // reduce the 4 waves in LDS (fixed order): D[row = co = kq*4 + r][col = ci = l15] -> Ds[co][tap*16 + ci]
    float* Ds = reinterpret_cast<float*>(wsm_raw);   // 16 x 144 floats = 9.2 KB
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (kq * 4 + r) * 144 + tap * 16 + l15;
                    if (w == 0) Ds[o] = acc[tap][r];
                    else Ds[o] += acc[tap][r];
                }
        }
    }
    __syncthreads();
    const int nci32 = (Cin + 31) / 32, nco32 = (Cout + 31) / 32;
    float* outp = partials + (((size_t)split * nco32 + blockIdx.z / 2) * nci32 + blockIdx.y / 2) * (32 * 288) + (blockIdx.y & 1) * 16;
    for (int e = tid; e < 16 * 144; e += kCT) {
        const int co = (blockIdx.z & 1) * 16 + e / 144, tap = (e % 144) / 16, ci = e % 16;
        outp[co * 288 + tap * 32 + ci] = Ds[e];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, int nsplit, int Cout, int Cin, int nco,
                                                           int nci, float* __restrict__ gw) {
    // walk the partial tiles in THEIR order ([co tile][ci tile][co 32][tap 9][ci 32]: consecutive threads read consecutive
    // floats of every part); the weight-gradient index takes the permutation (read many parts, write once)
    const size_t stride = (size_t)nco * nci * (32 * 288);
    reduce_partials_block(partials, nsplit, stride, nco * nci * 32 * 288, gw, [](int e) { return (size_t)e; }, [=](int e) {
        const int ci32 = e & 31, tap = (e >> 5) % 9, co32 = (e / 288) & 31, tile = e / (32 * 288);
        const int co = (tile / nci) * 32 + co32, ci = (tile % nci) * 32 + ci32;
        return (co < Cout && ci < Cin) ? (co * Cin + ci) * 9 + tap : -1;
    });
}

// ------------------------------------------------------------------------------------------ 1x1 logits head
template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const T* __restrict__ in, int64_t npix, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
    // the pixel's CI channels arrive as 16-byte vectors and (CO == 4) the logits leave as one float4: per-element 2-byte loads
    // and 4-byte stores ran this 150 MB stream at 0.9 TB/s
    float wr[CO * CI], br[CO];
#pragma unroll
    for (int e = 0; e < CO * CI; ++e) wr[e] = w[e];
#pragma unroll
    for (int o = 0; o < CO; ++o) br[o] = bias[o];
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        constexpr int NV = CI * (int)sizeof(T) / 16;
        static_assert(CI * sizeof(T) % 16 == 0, "conv1x1_fwd: the channel vector must be a multiple of 16 bytes");
        uint4 vin[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) vin[v] = reinterpret_cast<const uint4*>(in + i * CI)[v];
        const T* fe = reinterpret_cast<const T*>(vin);
        float f[CI], a[CO];
#pragma unroll
        for (int c = 0; c < CI; ++c) f[c] = to_f32(fe[c]);
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            a[o] = br[o];
#pragma unroll
            for (int c = 0; c < CI; ++c) a[o] += wr[o * CI + c] * f[c];
        }
        if (CO == 4) *reinterpret_cast<float4*>(out + i * 4) = make_float4(a[0], a[1 % CO], a[2 % CO], a[3 % CO]);
        else {
#pragma unroll
            for (int o = 0; o < CO; ++o) out[i * CO + o] = a[o];
        }
    }
}

template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv1x1_bwd_kernel(const T* __restrict__ in, const float* __restrict__ gout, int64_t npix,
                                                          const float* __restrict__ w, T* __restrict__ gin, float* __restrict__ partials) {
    __shared__ float red[4][CO * CI + CO];
    float gwacc[CO * CI], gbacc[CO];
#pragma unroll
    for (int e = 0; e < CO * CI; ++e) gwacc[e] = 0.f;
#pragma unroll
    for (int o = 0; o < CO; ++o) gbacc[o] = 0.f;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        // the pixel's CI channels travel as 16-byte vectors both ways (2-byte scalar stores were the bottleneck)
        constexpr int NV = CI * (int)sizeof(T) / 16;
        static_assert(CI * sizeof(T) % 16 == 0, "conv1x1_bwd: the channel vector must be a multiple of 16 bytes");
        uint4 vin[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) vin[v] = reinterpret_cast<const uint4*>(in + i * CI)[v];
        const T* fe = reinterpret_cast<const T*>(vin);
        float f[CI], g[CO];
#pragma unroll
        for (int c = 0; c < CI; ++c) f[c] = to_f32(fe[c]);
        if (CO == 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(gout + i * 4);
            g[0] = g4.x; g[1 % CO] = g4.y; g[2 % CO] = g4.z; g[3 % CO] = g4.w;
        } else {
#pragma unroll
            for (int o = 0; o < CO; ++o) g[o] = gout[i * CO + o];
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) gbacc[o] += g[o];
        uint4 vout[NV];
        T* oe = reinterpret_cast<T*>(vout);
#pragma unroll
        for (int c = 0; c < CI; ++c) {
            float a = 0.f;
#pragma unroll
            for (int o = 0; o < CO; ++o) { a += w[o * CI + c] * g[o]; gwacc[o * CI + c] += g[o] * f[c]; }
            oe[c] = from_f32<T>(a);
        }
        if (gin) {
#pragma unroll
            for (int v = 0; v < NV; ++v) reinterpret_cast<uint4*>(gin + i * CI)[v] = vout[v];
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < CO * CI; ++e) {
        float v = wave_sum(gwacc[e]);
        if (lane == 0) red[wv][e] = v;
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        float v = wave_sum(gbacc[o]);
        if (lane == 0) red[wv][CO * CI + o] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < CO * CI + CO; e += 256)
        partials[(size_t)blockIdx.x * (CO * CI + CO) + e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
}

__global__ __launch_bounds__(256) void sum_parts2_kernel(const float* __restrict__ partials, int nparts, int len, int n0, float* __restrict__ d0,
                                                         float* __restrict__ d1) {
    reduce_partials_block(partials, nparts, (size_t)len, len, Split2Out{d0, n0, d1}, [](int e) { return (size_t)e; });
}

// ---- The stem (one image channel in) without matrix cores (MISEG_STEM_KERNELS=0: the MFMA path).
// `conv3x3_stream_kernel` treats the padded channel vector as 8 (of 32) input channels: 31 of 32 multiplies are by zero, the launch takes
// 50 us at 256 x 256 for a 100 MB store, and the padded operand has to be written first (15 us); its weight gradient (59 + 10 us) is the
// LAST kernel of the backward pass, alone on the GPU, i.e. all of it is step tail.  These kernels read the fp32 image itself and take
// 50 / 52 + 7 us (rows through a prefetched LDS ring; staging three rows per output row without prefetch: 56 / 57; nine direct loads per
// thread: 122 / 140; on the padded operand, 16-byte stride: 148 / 156).
// A thread owns one pixel and four output channels: nine bf16 x bf16 products per output (exact in fp32, accumulated by FMA in tap order), 8-byte stores that
// are contiguous across the lanes of a pixel; statistics from the fp32 accumulators as everywhere.  x: [N][H][W][CP] (channel 0 is the
// image), w: the fp32 master [Cout][Cw][3][3] (Cw >= 1: only input channel 0 is read), rounded to the storage type as the packed
// weights of the MFMA path are.
// The input rows of a block's run of output rows, as a ring of four LDS rows (+ one row of zeros): an output row reads rows r - 1, r,
// r + 1; row r + 2 is committed from registers behind its compute and row r + 3 is already in flight (one barrier per row, the
// global round trip behind a whole row's arithmetic).  Rows of the neighbouring image / outside the image read the zero row.
template <typename T, typename TX, int NPRE>
struct StemRing {
    unsigned short* buf;      // [5][RB]: slots 0..3 the ring, slot 4 zeros
    const TX* x;
    int RB, W, CP, rows;
    unsigned short pre[NPRE];
    __device__ __forceinline__ void fetch(int rin) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int c = (int)threadIdx.x + 256 * k - 1;
            T v = from_f32<T>(0.f);
            if ((unsigned)rin < (unsigned)rows && (unsigned)c < (unsigned)W) v = from_f32<T>(to_f32(x[((size_t)rin * W + c) * CP]));    // an fp32 image is rounded here
            pre[k] = *reinterpret_cast<const unsigned short*>(&v);
        }
    }
    __device__ __forceinline__ void commit(int rin) {
        unsigned short* d = buf + ((rin + 1) & 3) * RB;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = (int)threadIdx.x + 256 * k;
            if (i < RB) d[i] = pre[k];
        }
    }
    __device__ __forceinline__ const unsigned short* row(int rin, bool inside) const { return inside ? buf + ((rin + 1) & 3) * RB : buf + 4 * RB; }
    // rows r0 - 1, r0, r0 + 1 staged, r0 + 2 in flight
    __device__ __forceinline__ void start(int r0) {
        for (int i = threadIdx.x; i < RB; i += 256) buf[4 * RB + i] = 0;
        for (int d = -1; d <= 1; ++d) { fetch(r0 + d); commit(r0 + d); }
        fetch(r0 + 2);
        __syncthreads();
    }
    // after the compute of row r: row r + 2 becomes readable, row r + 3 takes off
    __device__ __forceinline__ void advance(int r) {
        commit(r + 2);
        fetch(r + 3);
        __syncthreads();
    }
};
constexpr int kStemPre = 4;       // W + 2 <= 1024

template <typename T, typename TX>
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const TX* __restrict__ x, int CP, int N, int H, int W, const float* __restrict__ w, int Cw,
                                                            int Cout, T* __restrict__ out, unsigned long long* __restrict__ acc) {
    const int G = Cout / 4, g = threadIdx.x % G, ppb = 256 / G;
    float wr[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[tap][r] = to_f32(from_f32<T>(w[((size_t)(g * 4 + r) * Cw) * 9 + tap]));
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // a block walks a contiguous run of image rows (row = n * H + h: block-uniform, so is h), its threads the row's pixels
    extern __shared__ unsigned short stem_rows[];        // [5][W + 2]
    const int rows = N * H, c0 = threadIdx.x / G, per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r_begin = blockIdx.x * per, r_end = min(rows, r_begin + per);
    StemRing<T, TX, kStemPre> ring{stem_rows, x, W + 2, W, CP, rows, {}};
    if (r_begin < r_end) ring.start(r_begin);
    for (int row = r_begin; row < r_end; ++row) {
      const int h = row % H;
      const unsigned short* rp[3] = {ring.row(row - 1, h > 0), ring.row(row, true), ring.row(row + 1, h < H - 1)};
      for (int wq = c0; wq < W; wq += ppb) {
        const size_t p = (size_t)row * W + wq;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = bf16_bits_to_f32(rp[tap / 3][wq + tap % 3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaf(v, wr[tap][r], a[r]);
        }
        T o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { o4[r] = from_f32<T>(a[r]); s1[r] += a[r]; s2[r] += a[r] * a[r]; }
        *reinterpret_cast<uint2*>(out + p * Cout + g * 4) = *reinterpret_cast<const uint2*>(o4);
      }
      ring.advance(row);
    }
    if (acc) {       // block sums per channel (fixed order), then the fixed-point accumulator (common.h bn_acc_add)
        __shared__ float sred[256][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { sred[threadIdx.x][r] = s1[r]; sred[threadIdx.x][4 + r] = s2[r]; }
        __syncthreads();
        if ((int)threadIdx.x < 2 * Cout) {
            const int which = threadIdx.x / Cout, c = threadIdx.x % Cout, gg = c / 4, r = c % 4;
            float t = 0.f;
            for (int m = 0; m < ppb; ++m) t += sred[m * G + gg][which * 4 + r];
            bn_acc_add(acc + which * Cout + c, acc + 2 * Cout, t);
        }
    }
}

// gw[co][tap] = sum_px graw[px][co] * x[px + tap]: the same thread map and row ring, 36 accumulators per thread over the block's pixels,
// one partial vector [Cout * 9] per block (summed by stem_wgrad_sum_kernel in fixed order: deterministic)
template <typename T, typename TX>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const TX* __restrict__ x, int CP, int N, int H, int W, const T* __restrict__ graw, int Cout,
                                                         float* __restrict__ partials) {
    const int G = Cout / 4, g = threadIdx.x % G, ppb = 256 / G;
    float a[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[tap][r] = 0.f;
    extern __shared__ float swr[];      // [256][37]: the threads' 36 sums (odd stride: conflict-free column reads); behind it the row ring
    const int rows = N * H, c0 = threadIdx.x / G, per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r_begin = blockIdx.x * per, r_end = min(rows, r_begin + per);
    StemRing<T, TX, kStemPre> ring{reinterpret_cast<unsigned short*>(swr + 256 * 37), x, W + 2, W, CP, rows, {}};
    if (r_begin < r_end) ring.start(r_begin);
    for (int row = r_begin; row < r_end; ++row) {
      const int h = row % H;
      const unsigned short* rp[3] = {ring.row(row - 1, h > 0), ring.row(row, true), ring.row(row + 1, h < H - 1)};
      for (int wq = c0; wq < W; wq += ppb) {
        const size_t p = (size_t)row * W + wq;
        const uint2 gq = *reinterpret_cast<const uint2*>(graw + p * Cout + g * 4);
        T g4[4];
        *reinterpret_cast<uint2*>(g4) = gq;
        const float gv[4] = {to_f32(g4[0]), to_f32(g4[1]), to_f32(g4[2]), to_f32(g4[3])};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = bf16_bits_to_f32(rp[tap / 3][wq + tap % 3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[tap][r] = fmaf(gv[r], v, a[tap][r]);
        }
      }
      ring.advance(row);
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r) swr[threadIdx.x * 37 + tap * 4 + r] = a[tap][r];
    __syncthreads();
    for (int o = threadIdx.x; o < Cout * 9; o += 256) {
        const int co = o / 9, tap = o % 9, gg = co / 4, r = co % 4;
        float t = 0.f;
        for (int m = 0; m < ppb; ++m) t += swr[(m * G + gg) * 37 + tap * 4 + r];
        partials[(size_t)blockIdx.x * Cout * 9 + o] = t;
    }
}
__global__ __launch_bounds__(256) void stem_wgrad_sum_kernel(const float* __restrict__ partials, int nparts, int len, float* __restrict__ gw) {
    reduce_partials_block(partials, nparts, (size_t)len, len, gw, [](int e) { return (size_t)e; });
}
constexpr int kStemBlocksMax = 4096;
static const int kStemBlocks = [] { const char* e = getenv("MISEG_STEM_BLOCKS"); return e ? std::min(kStemBlocksMax, std::max(1, atoi(e))) : 1024; }();      // (ws sized for kStemBlocksMax)

static inline int tile_w(int64_t W) { return W >= 32 ? 32 : 16; }

}  // namespace miseg

using namespace miseg;

extern "C" int miseg_pack_conv3x3_weights(void* stream, int dt, const float* w, int64_t Cout, int64_t Cin, int kind, int64_t ci_begin,
                                          int64_t ci_count, void* packed) {
    MISEG_TAPE(miseg_pack_conv3x3_weights, stream, dt, w, Cout, Cin, kind, ci_begin, ci_count, packed);
    MISEG_F16_DISPATCH_ON(dt, miseg_pack_conv3x3_weights, stream, MISEG_BF16, w, Cout, Cin, kind, ci_begin, ci_count, packed);
    MISEG_REQUIRE(w && packed && Cout > 0 && Cin > 0, "pack_conv3x3_weights: bad args");
    MISEG_REQUIRE(kind >= 0 && (kind & 0xff) <= 1 && (kind >> 8) <= Cin && (!(kind >> 8) || !(kind & 0xff)), "pack_conv3x3_weights: bad kind");
    if (!(kind & 0xff)) { ci_begin = 0; ci_count = Cin; }
    MISEG_REQUIRE(ci_begin >= 0 && ci_count > 0 && ci_begin + ci_count <= Cin, "pack_conv3x3_weights: bad channel slice");
    const int total = (int)(9 * ((kind & 0xff) ? ci_count * Cout : Cout * Cin));
    const int nb = std::min((total + 255) / 256, 1024);
    if (dt == MISEG_F32)
        hipLaunchKernelGGL(pack_w_kernel<float>, dim3(nb), dim3(256), 0, as_stream(stream), w, (int)Cout, (int)Cin, kind, (int)ci_begin, (int)ci_count, (float*)packed);
    else if (dt == MISEG_BF16)
        hipLaunchKernelGGL(pack_w_kernel<bf16>, dim3(nb), dim3(256), 0, as_stream(stream), w, (int)Cout, (int)Cin, kind, (int)ci_begin, (int)ci_count, (bf16*)packed);
    else return fail(MISEG_E_INVALID, "pack_conv3x3_weights: bad dtype");
    MISEG_LAUNCH_CHECK("pack_w_kernel");
    return MISEG_OK;
}

extern "C" int miseg_pack_conv3x3_weights_multi(void* stream, int dt, const void* jobs_dev, int64_t njobs, int64_t total_blocks) {
    MISEG_TAPE(miseg_pack_conv3x3_weights_multi, stream, dt, jobs_dev, njobs, total_blocks);
    MISEG_F16_DISPATCH_ON(dt, miseg_pack_conv3x3_weights_multi, stream, MISEG_BF16, jobs_dev, njobs, total_blocks);
    MISEG_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "pack_conv3x3_weights_multi: bad args");
    static_assert(sizeof(PackJob) == 40, "PackJob layout must match miseg_pack_job");
    if (dt == MISEG_F32)
        hipLaunchKernelGGL(pack_w_multi_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), (const PackJob*)jobs_dev, (int)njobs);
    else if (dt == MISEG_BF16)
        hipLaunchKernelGGL(pack_w_multi_kernel<bf16>, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), (const PackJob*)jobs_dev, (int)njobs);
    else return fail(MISEG_E_INVALID, "pack_conv3x3_weights_multi: bad dtype");
    MISEG_LAUNCH_CHECK("pack_w_multi_kernel");
    return MISEG_OK;
}

// gw[Cout][Cin][9] <- gw_pad[Cout][Cin_pad][9] (first Cin input channels): the stem's weight gradient, computed for one whole channel
// vector of (zero) input channels, into the parameter's own gradient (ref contrastyou/arch/unet.py:15: Conv2d(input_dim, 16, 3)).
namespace miseg {
__global__ __launch_bounds__(256) void slice_cin_kernel(const float* __restrict__ src, int Cout, int Cin_pad, int Cin, float* __restrict__ dst) {
    const int total = Cout * Cin * 9;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int t = e % 9, ci = (e / 9) % Cin, co = e / (9 * Cin);
        dst[e] = src[((size_t)co * Cin_pad + ci) * 9 + t];
    }
}
}  // namespace miseg
extern "C" int miseg_conv3x3_wgrad_slice(void* stream, const float* gw_padded, int64_t Cout, int64_t Cin_pad, int64_t Cin, float* gw) {
    MISEG_TAPE(miseg_conv3x3_wgrad_slice, stream, gw_padded, Cout, Cin_pad, Cin, gw);
    MISEG_REQUIRE(gw_padded && gw && Cout > 0 && Cin > 0 && Cin <= Cin_pad, "conv3x3_wgrad_slice: bad args");
    const int total = (int)(Cout * Cin * 9);
    hipLaunchKernelGGL(slice_cin_kernel, dim3(std::min((total + 255) / 256, 256)), dim3(256), 0, as_stream(stream), gw_padded, (int)Cout, (int)Cin_pad,
                       (int)Cin, gw);
    MISEG_LAUNCH_CHECK("slice_cin_kernel");
    return MISEG_OK;
}

// the persistent streaming kernel serves the HBM-bound bf16 shapes (one channel chunk, enough tiles to fill the chip)
static bool conv_streams(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W) {
    return dt == MISEG_BF16 && Cin <= CK && tile_w(W) == 32 && N * cdiv(H, TH) * cdiv(W, 32) >= 512;
}
static int64_t stream_blocks(int64_t N, int64_t H, int64_t W) {
    static const int64_t cap = [] { const char* e = getenv("MISEG_STREAM_BLOCKS"); return e ? atoll(e) : 512LL; }();    // persistent blocks (2 per CU)
    return std::min<int64_t>(N * cdiv(H, TH) * cdiv(W, 32), cap);
}

// Grid of the persistent tiled kernel: `blocks` x-blocks of `per` consecutive tiles each (none empty), sized so that blocks.x * slices
// fills the chip once: two blocks per CU for the 8-row tiles (60 KB of LDS each), one for the 16-row tiles.
struct PtGrid { int blocks, per; };
static inline PtGrid pt_grid(int ntiles, int slices, int th) {
    static const int per_cu = [] { const char* e = getenv("MISEG_PT_BLOCKS_PER_CU"); return e ? std::max(1, atoi(e)) : 2; }();     // diagnostic: 1, 3, 4 measured slower
    const int target = std::max(1, (256 * (th == 8 ? per_cu : 1)) / std::max(1, slices));
    const int per = std::max(1, (ntiles + target - 1) / target);
    return PtGrid{(ntiles + per - 1) / per, per};
}
static inline bool tiled_persistent() {
    static const bool on = [] { const char* e = getenv("MISEG_CONV_PERSISTENT"); return !e || atoi(e) != 0; }();
    return on;
}

// tile height of the generic kernel (bf16): 8 rows from 64^2 upwards (smaller LDS tile -> two blocks per CU; measured
// 128^2 64->32: 78 -> 64 us), 16 rows for the small deep layers (32^2: 8-row tiles cost 46 -> 55 us) and for exact fp32
static int generic_tile_h(int dt, int64_t H, int64_t W) {
    if (dt != MISEG_BF16) return 16;
    return H * W >= 64 * 64 ? 8 : 16;
}

extern "C" int64_t miseg_conv3x3_stats_parts(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W) {
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_stats_parts, MISEG_BF16, Cin, N, H, W);
    if (conv_streams(dt, Cin, N, H, W)) return stream_blocks(N, H, W);
    const int tw = tile_w(W);
    return N * cdiv(H, generic_tile_h(dt, H, W)) * cdiv(W, tw);
}

// Rows of the statistics matrix miseg_conv3x3_fwd writes for this layer (what bn_finalize must be told): one per block of the kernel
// that serves the shape -- the persistent tiled kernel has far fewer blocks than tiles.
extern "C" int64_t miseg_conv3x3_fwd_parts(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout) {
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_fwd_parts, MISEG_BF16, Cin, N, H, W, Cout);
    const int64_t ntiles = miseg_conv3x3_stats_parts(dt, Cin, N, H, W);
    if (conv_streams(dt, Cin, N, H, W) || dt != MISEG_BF16 || !tiled_persistent()) return ntiles;
    const int th = generic_tile_h(dt, H, W);
    if (th != 8) return ntiles;                                                          // (launch_tiled: the persistent form serves 8-row tiles)
    const int64_t cot = Cout <= 16 ? 16 : 32;                                           // conv3x3_fwd_impl's slice width for 8-row tiles
    return pt_grid((int)ntiles, (int)cdiv(Cout, cot), th).blocks;
}

// pooled-output forms exist for the streaming shapes and, in the tiled kernel, for 16-bit storage with more than 32 output channels
static bool sumpool_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout) {
    if (H % 2 || W % 2) return false;
    return conv_streams(dt, Cin, N, H, W) || (dt == MISEG_BF16 && Cout > 32);
}

// One tile shape of the tiled kernel / one slot layout of the streaming kernel in the forms a call can ask for: plain, BN loader,
// epilogue reduce, both (the pooled-output forms: plain and BN loader).
template <typename TT, int COT, int TWW, int THH, bool POOL>
static void launch_tiled(dim3 grid, hipStream_t st, const ConvSrc& s, int N, int H, int W, const void* wpk, int Cout, void* out, float* stats,
                         const BnFinish& fin, const BnLoad& bl, const BnRed& br) {
    const size_t lb = ((size_t)(THH + 2) * (TWW + 2) + 9 * COT) * Mma<TT>::KP * sizeof(TT);
    if constexpr (sizeof(TT) == 2) {
        // the plain forms of the 8-row tiles (maps from 64^2 up: several tiles per block): persistent blocks, pipelined across tiles.
        // Measured per launch, same box (gpurun_out/prof_pt*): 32-channel slices 48.1 -> 38.7 us, pooled 64-channel 75.5 -> 67.6; the
        // 16-row tiles of the deep layers have one tile per block either way and ran 4-29 % slower in this form -> they keep the other
        if (THH == 8 && !bl.gy && !br.raw && !fin.counter && tiled_persistent()) {
            const int ntiles = (int)grid.x;
            const PtGrid g = pt_grid(ntiles, (int)grid.y, THH);
            constexpr int NWW = (THH == 16) ? 8 : 4;
            // stages in flight ahead of the matrix cores: ONE.  Two (201 instead of 155 registers in the 32-channel-slice form) measured
            // 6.72 / 6.73 against 6.66 / 6.63 ms per step, same box, two alternations (gpurun_out/ab_depth_*.json); MISEG_CONV_DEPTH=2 selects it
            static const int depth = [] { const char* e = getenv("MISEG_CONV_DEPTH"); return e ? atoi(e) : 1; }();
            if (depth == 2 && !POOL) {      // (the pooled 64-channel form with two stages in flight needs 256 registers: one wave per SIMD)
                hipFuncSetAttribute((const void*)conv3x3_pt_kernel<TT, COT, TWW, THH, POOL, NWW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL((conv3x3_pt_kernel<TT, COT, TWW, THH, POOL, NWW, 2>), dim3((unsigned)g.blocks, grid.y), dim3(64 * NWW), lb, st, s, N, H, W,
                                   (const TT*)wpk, Cout, (TT*)out, stats, fin, ntiles, g.per);
            } else {
                hipFuncSetAttribute((const void*)conv3x3_pt_kernel<TT, COT, TWW, THH, POOL, NWW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL((conv3x3_pt_kernel<TT, COT, TWW, THH, POOL, NWW, 1>), dim3((unsigned)g.blocks, grid.y), dim3(64 * NWW), lb, st, s, N, H, W,
                                   (const TT*)wpk, Cout, (TT*)out, stats, fin, ntiles, g.per);
            }
            return;
        }
    }
#define GO(BNL, RED, NWW)                                                                                                            \
    {                                                                                                                               \
        hipFuncSetAttribute((const void*)conv3x3_kernel<TT, COT, TWW, THH, POOL, BNL, RED, NWW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); \
        hipLaunchKernelGGL((conv3x3_kernel<TT, COT, TWW, THH, POOL, BNL, RED, NWW>), grid, dim3(64 * NWW), lb, st, s, N, H, W, (const TT*)wpk, \
                           Cout, (TT*)out, stats, fin, bl, br);                                                                     \
    }
    // 16-row tiles of 16-bit layers: 8 waves (plain forms; the fused forms keep 4).  Same box, forward, us: 32^2 64->128 21.3 -> 20.6,
    // 128->128 29.4 -> 26.5, 256->128 45.4 -> 41.9; 16^2 128->256 19.0 -> 17.8, 256->256 27.4 -> 26.3 (gpurun_out/sweep_nw.log)
    if constexpr (THH == 16 && sizeof(TT) == 2) {
        if (!bl.gy && !br.raw) { GO(false, false, 8) return; }
    }
    if constexpr (POOL) { if (bl.gy) GO(true, false, 4) else GO(false, false, 4) }
    else { if (bl.gy && br.raw) GO(true, true, 4) else if (bl.gy) GO(true, false, 4) else if (br.raw) GO(false, true, 4) else GO(false, false, 4) }
#undef GO
}
// rows per wave of the streaming kernel's fused forms (see the kernel): 2 unless 16 gradient channels meet 16 outputs or a pooled output
template <int COT, int NV, bool POOL> constexpr int fused_rw() { return (NV == 2 && (COT == 16 || POOL)) ? 4 : 2; }
template <int COT, int NV, bool DU, bool POOL>
static int launch_stream(unsigned blocks, hipStream_t st, const ConvSrc& s, int N, int H, int W, const void* wpk, int Cout, void* out, float* stats,
                         const BnFinish& fin, const BnLoad& bl, const BnRed& br) {
    const dim3 grid(blocks, (unsigned)cdiv(Cout, COT));
#define GO(BNL, RED, RWW)                                                                                                            \
    {                                                                                                                               \
        const size_t lb = ((size_t)(4 * RWW + 2) * (32 + 2) * (COT == 16 ? 48 : Mma<bf16>::CKP) + 9 * COT * Mma<bf16>::CKP) * sizeof(bf16) + \
                          (BNL ? kBwdCoefRows * CK * 4 : 0) + (RED ? 3 * COT * 4 : 0);                                               \
        const int ntiles = (int)(N * cdiv(H, 4 * RWW) * cdiv(W, 32));                                                                \
        hipFuncSetAttribute((const void*)conv3x3_stream_kernel<COT, 32, NV, DU, POOL, BNL, RED, RWW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb); \
        hipLaunchKernelGGL((conv3x3_stream_kernel<COT, 32, NV, DU, POOL, BNL, RED, RWW>), grid, dim3(kCT), lb, st, s, N, H, W,          \
                           (const bf16*)wpk, Cout, (bf16*)out, stats, ntiles, fin, bl, br);                                          \
    }
    constexpr bool kFusable = !DU && (NV == 2 || NV == 4);      // the data-gradient shapes of the U-Net: 16 or 32 gradient channels
    if constexpr (kFusable) {
        constexpr int FRW = fused_rw<COT, NV, POOL>();
        if constexpr (POOL) { if (bl.gy) GO(true, false, FRW) else GO(false, false, 4) }
        else { if (bl.gy && br.raw) GO(true, true, FRW) else if (bl.gy) GO(true, false, FRW) else if (br.raw) GO(false, true, FRW) else GO(false, false, 4) }
    } else {
        if (bl.gy || br.raw) return fail(MISEG_E_INVALID, "conv3x3: BatchNorm loader / reduce not built for this streaming shape");
        GO(false, false, 4)
    }
#undef GO
    return MISEG_OK;
}

static int conv3x3_fwd_impl(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                            int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out, float* stats, BnFinish fin,
                            bool pool_out = false, BnLoad bl = BnLoad{nullptr, nullptr}, BnRed br = BnRed{nullptr, nullptr, nullptr}) {
    MISEG_REQUIRE(in0 && packed_w && out, "conv3x3_fwd: null pointer");
    MISEG_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && C0 > 0 && C1 >= 0, "conv3x3_fwd: bad shape");
    MISEG_REQUIRE(C1 == 0 || in1, "conv3x3_fwd: second source missing");
    MISEG_REQUIRE((ups0 == 0 || ups0 == 1) && (ups1 == 0 || ups1 == 1), "conv3x3_fwd: ups must be 0/1");
    MISEG_REQUIRE((!ups0 || (H % 2 == 0 && W % 2 == 0)) && (!ups1 || (H % 2 == 0 && W % 2 == 0)), "conv3x3_fwd: upsampled size must be even");
    const int vec = dt == MISEG_BF16 ? 8 : 4;
    MISEG_REQUIRE(C0 % vec == 0 && C1 % vec == 0, "conv3x3_fwd: channel counts must be multiples of %d (use conv_small for the stem)", vec);
    ConvSrc s{in0, in1, (int)C0, (int)C1, ups0, ups1};
    const int tw = tile_w(W);
    const int th = conv_streams(dt, C0 + C1, N, H, W) ? TH : generic_tile_h(dt, H, W);
    const unsigned gx = (unsigned)(N * cdiv(H, th) * cdiv(W, tw));
    hipStream_t st = as_stream(stream);
    MISEG_REQUIRE(!bl.gy || (bl.coef && C1 == 0 && ups0 == 0), "conv3x3: the BatchNorm loader reads one full-resolution source");
    MISEG_REQUIRE(!br.raw || (br.saved && br.parts && !pool_out && !stats && Cout % 4 == 0), "conv3x3: epilogue reduce needs a full-resolution "
                  "output with a multiple of 4 channels and no forward statistics");
#define LAUNCH_TH(TT, COT, TWW, THH)                                                                                       \
    launch_tiled<TT, COT, TWW, THH, false>(dim3(gx, (unsigned)cdiv(Cout, COT)), st, s, (int)N, (int)H, (int)W, packed_w, (int)Cout, out, stats, fin, bl, br);
#define LAUNCH(TT, COT, TWW) { if (th == 8) LAUNCH_TH(TT, COT, TWW, 8) else LAUNCH_TH(TT, COT, TWW, 16) }
    MISEG_REQUIRE(!pool_out || (sumpool_supported(dt, C0 + C1, N, H, W, Cout) && C1 == 0 && !stats),
                  "conv3x3_fwd_sumpool: shape not supported (ask miseg_conv3x3_fwd_sumpool_supported)");
    if (conv_streams(dt, C0 + C1, N, H, W)) {
        int rc = MISEG_OK;
#define SLAUNCH(COT, NV, DU)                                                                                               \
    {                                                                                                                     \
        const unsigned g = (unsigned)stream_blocks(N, H, W);                                                               \
        if (pool_out && !DU) rc = launch_stream<COT, NV, false, true>(g, st, s, (int)N, (int)H, (int)W, packed_w, (int)Cout, out, stats, fin, bl, br); \
        else rc = launch_stream<COT, NV, DU, false>(g, st, s, (int)N, (int)H, (int)W, packed_w, (int)Cout, out, stats, fin, bl, br); \
    }
#define SLAUNCH_NV(COT)                                                                                                    \
    {                                                                                                                     \
        if (C1 > 0) { switch ((C0 + C1) / 8) { case 2: SLAUNCH(COT, 2, true) break; case 3: SLAUNCH(COT, 3, true) break; default: SLAUNCH(COT, 4, true) break; } } \
        else { switch (C0 / 8) { case 1: SLAUNCH(COT, 1, false) break; case 2: SLAUNCH(COT, 2, false) break; case 3: SLAUNCH(COT, 3, false) break; default: SLAUNCH(COT, 4, false) break; } } \
    }
        // wider outputs run as 32-channel slices (grid.y): a 64-channel block would need 97 KB of LDS and the whole
        // register file, i.e. one block per CU; re-reading the (narrow) input per slice is cheaper
        if (Cout <= 16) SLAUNCH_NV(16)
        else SLAUNCH_NV(32)
#undef SLAUNCH_NV
#undef SLAUNCH
        if (rc != MISEG_OK) return rc;
    } else if (dt == MISEG_BF16 && pool_out) {
#define LAUNCH_P(TWW, THH)                                                                                                 \
    launch_tiled<bf16, 64, TWW, THH, true>(dim3(gx, (unsigned)cdiv(Cout, 64)), st, s, (int)N, (int)H, (int)W, packed_w, (int)Cout, out, stats, fin, bl, br);
        if (tw == 32) { if (th == 8) LAUNCH_P(32, 8) else LAUNCH_P(32, 16) }
        else { if (th == 8) LAUNCH_P(16, 8) else LAUNCH_P(16, 16) }
#undef LAUNCH_P
    } else if (dt == MISEG_BF16) {
        if (Cout <= 16) { if (tw == 32) LAUNCH(bf16, 16, 32) else LAUNCH(bf16, 16, 16) }
        // 8-row tiles (maps from 64^2 up) take 32 output channels per block: with 64 the weight chunk makes the block's LDS 88 KB -- one
        // block per CU --, with 32 it is 60 KB and two fit (64^2, same box: 32->64 31.8 -> 26.7 us, 64->64 39.2 -> 36.3, 128->64 55.5 -> 49.9;
        // the 16-row tiles of the deep layers are one block per CU either way and lose 20 % with the narrower slice)
        else if (Cout <= 32 || th == 8) { if (tw == 32) LAUNCH(bf16, 32, 32) else LAUNCH(bf16, 32, 16) }
        else { if (tw == 32) LAUNCH(bf16, 64, 32) else LAUNCH(bf16, 64, 16) }
    } else if (dt == MISEG_F32) {
        if (Cout <= 16) { if (tw == 32) LAUNCH(float, 16, 32) else LAUNCH(float, 16, 16) }
        else { if (tw == 32) LAUNCH(float, 32, 32) else LAUNCH(float, 32, 16) }
    } else return fail(MISEG_E_INVALID, "conv3x3_fwd: bad dtype");
#undef LAUNCH
#undef LAUNCH_TH
    MISEG_LAUNCH_CHECK("conv3x3_kernel");
    return MISEG_OK;
}

extern "C" int miseg_conv3x3_fwd(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                                 int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out, float* stats) {
    MISEG_TAPE(miseg_conv3x3_fwd, stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, stats);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_fwd, stream, MISEG_BF16, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, stats);
    return conv3x3_fwd_impl(stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, stats, BnFinish{});
}

extern "C" int64_t miseg_conv3x3_fwd_sumpool_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout) {
    return sumpool_supported(dt == MISEG_F16 ? MISEG_BF16 : dt, Cin, N, H, W, Cout);
}

extern "C" int miseg_conv3x3_fwd_sumpool(void* stream, int dt, const void* in, int64_t Cin, int64_t N, int64_t H, int64_t W,
                                         const void* packed_w, int64_t Cout, void* out_pooled) {
    MISEG_TAPE(miseg_conv3x3_fwd_sumpool, stream, dt, in, Cin, N, H, W, packed_w, Cout, out_pooled);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_fwd_sumpool, stream, MISEG_BF16, in, Cin, N, H, W, packed_w, Cout, out_pooled);
    return conv3x3_fwd_impl(stream, dt, in, Cin, 0, nullptr, 0, 0, N, H, W, packed_w, Cout, out_pooled, nullptr, BnFinish{}, true);
}

extern "C" int64_t miseg_conv3x3_fwd_sumpool_acc_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W) {
    return conv_streams(dt == MISEG_F16 ? MISEG_BF16 : dt, Cin, N, H, W) && H % 2 == 0 && W % 2 == 0;
}

extern "C" int miseg_conv3x3_fwd_sumpool_acc(void* stream, int dt, const void* in, int64_t Cin, int64_t N, int64_t H, int64_t W,
                                             const void* packed_w, int64_t Cout, void* inout_pooled) {
    MISEG_TAPE(miseg_conv3x3_fwd_sumpool_acc, stream, dt, in, Cin, N, H, W, packed_w, Cout, inout_pooled);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_fwd_sumpool_acc, stream, MISEG_BF16, in, Cin, N, H, W, packed_w, Cout, inout_pooled);
    MISEG_REQUIRE(miseg_conv3x3_fwd_sumpool_acc_supported(dt, Cin, N, H, W), "conv3x3_fwd_sumpool_acc: streaming shapes only");
    BnFinish opt{};
    opt.accumulate_out = 1;
    return conv3x3_fwd_impl(stream, dt, in, Cin, 0, nullptr, 0, 0, N, H, W, packed_w, Cout, inout_pooled, nullptr, opt, true);
}

// ---- data gradient of a convolution over the channel concat of two full-resolution sources: ONE launch, two destinations
extern "C" int miseg_conv3x3_dgrad_dual(void* stream, int dt, const void* graw, int64_t K, int64_t N, int64_t H, int64_t W, const void* packed_w,
                                        int64_t C0, void* out0, int64_t C1, void* out1) {
    MISEG_TAPE(miseg_conv3x3_dgrad_dual, stream, dt, graw, K, N, H, W, packed_w, C0, out0, C1, out1);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_dgrad_dual, stream, MISEG_BF16, graw, K, N, H, W, packed_w, C0, out0, C1, out1);
    MISEG_REQUIRE(out0 && out1 && C0 > 0 && C1 > 0 && C0 % 16 == 0 && C1 % 4 == 0, "conv3x3_dgrad_dual: the first source must have a multiple of "
                  "16 channels, the second a multiple of 4");
    BnFinish opt{};
    opt.out1 = out1;
    opt.split = (int)C0;
    return conv3x3_fwd_impl(stream, dt, graw, K, 0, nullptr, 0, 0, N, H, W, packed_w, C0 + C1, out0, nullptr, opt);
}

// ---- data gradient with the BatchNorm backward folded in (unet_ops._ConvBNReLU.backward)
static bool dgrad_bn_stream_ok(int dt, int64_t K, int64_t N, int64_t H, int64_t W) {
    return !conv_streams(dt, K, N, H, W) || K == 16 || K == 32;
}
extern "C" int64_t miseg_conv3x3_dgrad_bn_supported(int dt, int64_t K, int64_t N, int64_t H, int64_t W, int64_t Cs, int pool_out) {
    if (dt == MISEG_F16) dt = MISEG_BF16;
    const int vec = dt == MISEG_BF16 ? 8 : 4;
    if (K % vec || !dgrad_bn_stream_ok(dt, K, N, H, W)) return 0;
    return pool_out ? sumpool_supported(dt, K, N, H, W, Cs) : 1;
}
extern "C" int64_t miseg_conv3x3_dgrad_red_parts(int dt, int64_t K, int64_t N, int64_t H, int64_t W, int64_t Cs) {
    if (dt == MISEG_F16) dt = MISEG_BF16;
    if (Cs % 4 || !dgrad_bn_stream_ok(dt, K, N, H, W)) return 0;
    return miseg_conv3x3_stats_parts(dt, K, N, H, W);
}
extern "C" int miseg_conv3x3_dgrad_bn(void* stream, int dt, const void* raw_or_graw, const void* gy, const float* bwd_coef, int64_t K,
                                      int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cs, void* out, int pool_out,
                                      const void* red_raw, const float* red_saved, float* red_parts) {
    MISEG_TAPE(miseg_conv3x3_dgrad_bn, stream, dt, raw_or_graw, gy, bwd_coef, K, N, H, W, packed_w, Cs, out, pool_out, red_raw, red_saved, red_parts);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_dgrad_bn, stream, MISEG_BF16, raw_or_graw, gy, bwd_coef, K, N, H, W, packed_w, Cs, out, pool_out, red_raw,
                          red_saved, red_parts);
    MISEG_REQUIRE(!gy || bwd_coef, "conv3x3_dgrad_bn: coefficients missing");
    MISEG_REQUIRE(!red_raw || (!pool_out && red_saved && red_parts && miseg_conv3x3_dgrad_red_parts(dt, K, N, H, W, Cs) > 0),
                  "conv3x3_dgrad_bn: epilogue reduce not available for this call (ask miseg_conv3x3_dgrad_red_parts)");
    MISEG_REQUIRE(!gy || miseg_conv3x3_dgrad_bn_supported(dt, K, N, H, W, Cs, pool_out), "conv3x3_dgrad_bn: shape not supported (ask _supported)");
    return conv3x3_fwd_impl(stream, dt, raw_or_graw, K, 0, nullptr, 0, 0, N, H, W, packed_w, Cs, out, nullptr, BnFinish{}, pool_out != 0,
                            BnLoad{gy, bwd_coef}, BnRed{red_raw, red_saved, red_parts});
}

// The partial-sum matrix one finishing block reads: [parts][2 Cout] floats.  Above this the separate, C-block bn_finalize is faster.
static const int64_t kBnFinishMaxFloats = [] { const char* e = getenv("MISEG_FINISH_FLOATS"); return e ? atoll(e) : 65536LL; }();

// ---- the stem without matrix cores (one real input channel; see stem_conv_fwd_kernel)
extern "C" int64_t miseg_conv3x3_stem_supported(int dt, int64_t Cin_weight, int64_t CP, int64_t Cout) {
    return (dt == MISEG_BF16 || dt == MISEG_F16) && Cin_weight == 1 && CP >= 1 && Cout >= 4 && Cout <= 64 && Cout % 4 == 0 && 256 % (Cout / 4) == 0;     // (W <= 4096: LDS rows, checked per call)
}

extern "C" int miseg_conv3x3_stem_fwd(void* stream, int dt, const void* x, int x_f32, int64_t CP, int64_t N, int64_t H, int64_t W, const float* w,
                                      int64_t Cin_weight, int64_t Cout, void* out, void* acc_or_null) {
    MISEG_TAPE(miseg_conv3x3_stem_fwd, stream, dt, x, x_f32, CP, N, H, W, w, Cin_weight, Cout, out, acc_or_null);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_stem_fwd, stream, MISEG_BF16, x, x_f32, CP, N, H, W, w, Cin_weight, Cout, out, acc_or_null);
    MISEG_REQUIRE(x && w && out && N > 0 && H > 0 && W > 0, "conv3x3_stem_fwd: bad args");
    MISEG_REQUIRE(miseg_conv3x3_stem_supported(dt, Cin_weight, CP, Cout), "conv3x3_stem_fwd: 16-bit storage, one weight input channel, Cout in {4, 8, 16, 32, 64}");
    MISEG_REQUIRE(!acc_or_null || ((uintptr_t)acc_or_null & 7) == 0, "conv3x3_stem_fwd: the accumulator must be 8-byte aligned");
    MISEG_REQUIRE(W + 2 <= 256 * kStemPre, "conv3x3_stem_fwd: rows of at most %d pixels", 256 * kStemPre - 2);
    const unsigned nb = (unsigned)std::min<int64_t>(N * H, (int64_t)kStemBlocks);
    if (x_f32)
        hipLaunchKernelGGL((stem_conv_fwd_kernel<bf16, float>), dim3(nb), dim3(256), (size_t)5 * (W + 2) * 2, as_stream(stream), (const float*)x, (int)CP, (int)N, (int)H, (int)W, w,
                           (int)Cin_weight, (int)Cout, (bf16*)out, static_cast<unsigned long long*>(acc_or_null));
    else
        hipLaunchKernelGGL((stem_conv_fwd_kernel<bf16, bf16>), dim3(nb), dim3(256), (size_t)5 * (W + 2) * 2, as_stream(stream), (const bf16*)x, (int)CP, (int)N, (int)H, (int)W, w,
                           (int)Cin_weight, (int)Cout, (bf16*)out, static_cast<unsigned long long*>(acc_or_null));
    MISEG_LAUNCH_CHECK("stem_conv_fwd_kernel");
    return MISEG_OK;
}

extern "C" int64_t miseg_conv3x3_stem_wgrad_ws_bytes(int64_t Cout) { return (int64_t)kStemBlocksMax * Cout * 9 * 4; }

extern "C" int miseg_conv3x3_stem_wgrad(void* stream, int dt, const void* x, int x_f32, int64_t CP, int64_t N, int64_t H, int64_t W, const void* graw,
                                        int64_t Cout, float* gw, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_conv3x3_stem_wgrad, stream, dt, x, x_f32, CP, N, H, W, graw, Cout, gw, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_stem_wgrad, stream, MISEG_BF16, x, x_f32, CP, N, H, W, graw, Cout, gw, ws, ws_bytes);
    MISEG_REQUIRE(x && graw && gw && ws && N > 0 && H > 0 && W > 0, "conv3x3_stem_wgrad: bad args");
    MISEG_REQUIRE(miseg_conv3x3_stem_supported(dt, 1, CP, Cout), "conv3x3_stem_wgrad: 16-bit storage, Cout in {4, 8, 16, 32, 64}");
    MISEG_REQUIRE(ws_bytes >= miseg_conv3x3_stem_wgrad_ws_bytes(Cout), "conv3x3_stem_wgrad: workspace too small");
    MISEG_REQUIRE(W + 2 <= 256 * kStemPre, "conv3x3_stem_wgrad: rows of at most %d pixels", 256 * kStemPre - 2);
    const unsigned nb = (unsigned)std::min<int64_t>(N * H, (int64_t)kStemBlocks);
    hipStream_t st = as_stream(stream);
    if (x_f32)
        hipLaunchKernelGGL((stem_wgrad_kernel<bf16, float>), dim3(nb), dim3(256), (size_t)256 * 37 * 4 + (size_t)5 * (W + 2) * 2, st, (const float*)x, (int)CP, (int)N, (int)H, (int)W,
                           (const bf16*)graw, (int)Cout, (float*)ws);
    else
        hipLaunchKernelGGL((stem_wgrad_kernel<bf16, bf16>), dim3(nb), dim3(256), (size_t)256 * 37 * 4 + (size_t)5 * (W + 2) * 2, st, (const bf16*)x, (int)CP, (int)N, (int)H, (int)W,
                           (const bf16*)graw, (int)Cout, (float*)ws);
    MISEG_LAUNCH_CHECK("stem_wgrad_kernel");
    const int len = (int)(Cout * 9);
    hipLaunchKernelGGL(stem_wgrad_sum_kernel, dim3(reduce_grid(len, nb)), dim3(256), 0, st, (const float*)ws, (int)nb, len, gw);
    MISEG_LAUNCH_CHECK("stem_wgrad_sum_kernel");
    return MISEG_OK;
}

extern "C" int64_t miseg_conv3x3_fwd_acc_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout) {
    return Cout <= 256 && miseg_conv3x3_fwd_parts(dt, Cin, N, H, W, Cout) <= kBnAccMaxBlocks;       // the fixed point's no-wrap bound
}

// miseg_conv3x3_fwd with the BatchNorm statistics added into acc[2 Cout + 1] (fixed point, common.h bn_acc_add) instead of written as one row
// per block: no finalize launch -- miseg_bn_relu_fwd_acc reads the two totals of a channel itself
extern "C" int miseg_conv3x3_fwd_acc(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                                     int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out, void* acc) {
    MISEG_TAPE(miseg_conv3x3_fwd_acc, stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, acc);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_fwd_acc, stream, MISEG_BF16, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, acc);
    MISEG_REQUIRE(acc && ((uintptr_t)acc & 7) == 0, "conv3x3_fwd_acc: the accumulator must be an 8-byte aligned [2 Cout + 1] array");
    MISEG_REQUIRE(miseg_conv3x3_fwd_acc_supported(dt, C0 + C1, N, H, W, Cout), "conv3x3_fwd_acc: shape not supported (ask miseg_conv3x3_fwd_acc_supported)");
    BnFinish fin{};
    fin.acc = static_cast<unsigned long long*>(acc);
    return conv3x3_fwd_impl(stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, reinterpret_cast<float*>(acc), fin);
}

extern "C" int64_t miseg_conv3x3_bn_fwd_fusable(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout) {
    const int64_t parts = miseg_conv3x3_stats_parts(dt == MISEG_F16 ? MISEG_BF16 : dt, Cin, N, H, W);
    return Cout % 4 == 0 && Cout <= 256 && parts * 2 * Cout <= kBnFinishMaxFloats;
}

extern "C" int miseg_conv3x3_bn_fwd(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                                    int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out, float* stats,
                                    const float* gamma, const float* beta, float eps, float momentum, float* rmean, float* rvar,
                                    int64_t* nbt, float* saved, int32_t* sync_counter) {
    MISEG_TAPE(miseg_conv3x3_bn_fwd, stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, stats, gamma, beta, eps, momentum, rmean, rvar, nbt, saved, sync_counter);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_bn_fwd, stream, MISEG_BF16, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, stats, gamma, beta,
                          eps, momentum, rmean, rvar, nbt, saved, sync_counter);
    MISEG_REQUIRE(stats && gamma && beta && saved && sync_counter, "conv3x3_bn_fwd: null pointer");
    MISEG_REQUIRE(miseg_conv3x3_bn_fwd_fusable(dt, C0 + C1, N, H, W, Cout), "conv3x3_bn_fwd: shape not fusable (ask miseg_conv3x3_bn_fwd_fusable)");
    BnFinish fin{reinterpret_cast<unsigned int*>(sync_counter), gamma, beta, rmean, rvar, (long long*)nbt, saved, (float)(N * H * W), eps, momentum};
    return conv3x3_fwd_impl(stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, packed_w, Cout, out, stats, fin);
}

static int wgrad_splits(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout) {
    const int64_t ntiles = N * cdiv(H, WG_TH) * cdiv(W, WG_TW);
    const int64_t per = cdiv(Cin, 32) * cdiv(Cout, 32);
    // Blocks a weight-gradient launch is split into.  The launches run on their own stream beside the main stream's backward: with 512
    // the wgrad kernels themselves are fastest (1.86 ms per step in total), with 256 they take 1.96 ms but the STEP is 0.14 ms
    // shorter (7.13 vs 7.27 ms, same box, two alternations; 192: 7.18-7.35, 128: 7.31-7.47) -- one block per CU leaves the other
    // half of every CU's wave slots and LDS to the critical-path kernels.
    static const int target = [] { const char* e = getenv("MISEG_WGRAD_BLOCKS"); return e ? atoi(e) : 256; }();
    int64_t s = std::max<int64_t>(1, target / per);
    return (int)std::min<int64_t>(s, ntiles);
}

extern "C" int64_t miseg_conv3x3_wgrad_ws_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout) {
    return (int64_t)wgrad_splits(N, H, W, Cin, Cout) * cdiv(Cin, 32) * cdiv(Cout, 32) * 32 * 288 * 4;
}

static int conv3x3_wgrad_impl(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1, int64_t N,
                              int64_t H, int64_t W, const void* gout, int64_t Cout, float* gw, void* ws, int64_t ws_bytes, BnLoad bl) {
    MISEG_REQUIRE(in0 && gout && gw && ws, "conv3x3_wgrad: null pointer");
    MISEG_REQUIRE(C1 == 0 || in1, "conv3x3_wgrad: second source missing");
    const int64_t Cin = C0 + C1;
    MISEG_REQUIRE(ws_bytes >= miseg_conv3x3_wgrad_ws_bytes(N, H, W, Cin, Cout), "conv3x3_wgrad: workspace too small");
    ConvSrc s{in0, in1, (int)C0, (int)C1, ups0, ups1};
    const int ns = wgrad_splits(N, H, W, Cin, Cout), nci = (int)cdiv(Cin, 32), nco = (int)cdiv(Cout, 32);
    const size_t lb = ((size_t)WG_TH * WG_TW + (WG_TH + 2) * (WG_TW + 2)) * WG_P * 4;
    hipStream_t st = as_stream(stream);
    dim3 grid(ns, nci, nco);
    if (dt == MISEG_F32) {
        if (bl.gy) {
            hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<float, true>), grid, dim3(kCT), lb, st, s, (int)N, (int)H, (int)W, (const float*)gout, (int)Cout, ns, (float*)ws, bl);
        } else {
            hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            hipLaunchKernelGGL((conv3x3_wgrad_kernel<float, false>), grid, dim3(kCT), lb, st, s, (int)N, (int)H, (int)W, (const float*)gout, (int)Cout, ns, (float*)ws, bl);
        }
    } else if (dt == MISEG_BF16) {
        MISEG_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0 && Cout % 8 == 0, "conv3x3_wgrad: bf16 needs channel counts that are multiples of 8");
        const size_t lbb = std::max<size_t>(((size_t)2 * WG_TH * WG_TW + 2 * (WG_TH + 2) * (WG_TW + 2)) * 16 * 2 + 2 * 64 * 2, (size_t)(32 * 288 + 4 * 1024) * 4);
        // narrow layers (measured per shape, scratch/time_conv.py): the lean 16x16-tile kernel wins when one side has <= 16
        // channels (256^2 16->16: 125 -> 52 us, 32->16: 129 -> 83, 128^2 16->32: 85 -> 48) and for 32->64 (71 -> 53);
        // from 32->32 up the 32x32-tile kernel's operand reuse wins
        const bool narrow = Cout % 16 == 0 && (std::min(Cin, Cout) <= 16 || (Cin == 32 && Cout == 64));
        if (narrow || Cout <= 16) {
            const size_t lbc = (size_t)(WG_TH * WG_TW + (WG_TH + 2) * (WG_TW + 2)) * 16 * 2 + (bl.gy ? kBwdCoefRows * 16 * 4 : 0);
            dim3 gridc(ns, (unsigned)cdiv(Cin, 16), (unsigned)cdiv(Cout, 16));
            if (bl.gy) hipLaunchKernelGGL(conv3x3_wgrad_bf16_c16_kernel<true>, gridc, dim3(kCT), lbc, st, s, (int)N, (int)H, (int)W, (const bf16*)gout, (int)Cout, ns, (float*)ws, bl);
            else hipLaunchKernelGGL(conv3x3_wgrad_bf16_c16_kernel<false>, gridc, dim3(kCT), lbc, st, s, (int)N, (int)H, (int)W, (const bf16*)gout, (int)Cout, ns, (float*)ws, bl);
        } else if (bl.gy) {
            hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lbb);
            hipLaunchKernelGGL(conv3x3_wgrad_bf16_kernel<true>, grid, dim3(kCT), lbb, st, s, (int)N, (int)H, (int)W, (const bf16*)gout, (int)Cout, ns, (float*)ws, bl);
        } else {
            hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lbb);
            hipLaunchKernelGGL(conv3x3_wgrad_bf16_kernel<false>, grid, dim3(kCT), lbb, st, s, (int)N, (int)H, (int)W, (const bf16*)gout, (int)Cout, ns, (float*)ws, bl);
        }
    } else return fail(MISEG_E_INVALID, "conv3x3_wgrad: bad dtype");
    MISEG_LAUNCH_CHECK("conv3x3_wgrad_kernel");
    const int total = nco * nci * 32 * 288;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(total, ns)), dim3(256), 0, st, (const float*)ws, ns, (int)Cout, (int)Cin, nco, nci, gw);
    MISEG_LAUNCH_CHECK("wgrad_reduce_kernel");
    return MISEG_OK;
}

extern "C" int miseg_conv3x3_wgrad(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                                   int64_t N, int64_t H, int64_t W, const void* gout, int64_t Cout, float* gw, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_conv3x3_wgrad, stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, gout, Cout, gw, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_wgrad, stream, MISEG_BF16, in0, C0, ups0, in1, C1, ups1, N, H, W, gout, Cout, gw, ws, ws_bytes);
    return conv3x3_wgrad_impl(stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, gout, Cout, gw, ws, ws_bytes, BnLoad{nullptr, nullptr});
}

extern "C" int miseg_conv3x3_wgrad_bn(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                                      int64_t N, int64_t H, int64_t W, const void* raw, const void* gy, const float* bwd_coef, int64_t Cout,
                                      float* gw, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_conv3x3_wgrad_bn, stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, raw, gy, bwd_coef, Cout, gw, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv3x3_wgrad_bn, stream, MISEG_BF16, in0, C0, ups0, in1, C1, ups1, N, H, W, raw, gy, bwd_coef, Cout, gw, ws, ws_bytes);
    MISEG_REQUIRE(gy && bwd_coef, "conv3x3_wgrad_bn: null pointer");
    MISEG_REQUIRE(Cout % (dt == MISEG_F32 ? 4 : 8) == 0, "conv3x3_wgrad_bn: Cout must be a whole number of 16-byte vectors");
    return conv3x3_wgrad_impl(stream, dt, in0, C0, ups0, in1, C1, ups1, N, H, W, raw, Cout, gw, ws, ws_bytes, BnLoad{gy, bwd_coef});
}

extern "C" int miseg_conv1x1_fwd(void* stream, int dt, const void* in, int64_t N, int64_t H, int64_t W, int64_t Cin, const float* w,
                                 const float* bias, int64_t Cout, float* out) {
    MISEG_TAPE(miseg_conv1x1_fwd, stream, dt, in, N, H, W, Cin, w, bias, Cout, out);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv1x1_fwd, stream, MISEG_BF16, in, N, H, W, Cin, w, bias, Cout, out);
    MISEG_REQUIRE(in && w && bias && out, "conv1x1_fwd: null pointer");
    MISEG_REQUIRE(Cin == 16, "conv1x1_fwd: Cin must be 16 (unet.py:84)");
    const int64_t npix = N * H * W;
    const int nb = (int)std::min<int64_t>(cdiv(npix, 256), 4096);
    hipStream_t st = as_stream(stream);
#define L(CO)                                                                                                                      \
    if (dt == MISEG_F32) hipLaunchKernelGGL((conv1x1_fwd_kernel<float, 16, CO>), dim3(nb), dim3(256), 0, st, (const float*)in, npix, w, bias, out); \
    else hipLaunchKernelGGL((conv1x1_fwd_kernel<bf16, 16, CO>), dim3(nb), dim3(256), 0, st, (const bf16*)in, npix, w, bias, out)
    switch (Cout) {
        case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break;
        case 5: L(5); break; case 8: L(8); break;
        default: return fail(MISEG_E_INVALID, "conv1x1_fwd: unsupported Cout %ld", (long)Cout);
    }
#undef L
    MISEG_LAUNCH_CHECK("conv1x1_fwd_kernel");
    return MISEG_OK;
}

// 2 blocks per CU: the per-block epilogue (68 wave reductions) costs as much as ~20 pixels of the main loop
static int c1_blocks(int64_t npix) { return (int)std::min<int64_t>(cdiv(npix, 256), 512); }
extern "C" int64_t miseg_conv1x1_bwd_ws_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout) {
    return ((int64_t)c1_blocks(N * H * W) + 1) * (Cout * Cin + Cout) * 4;
}

extern "C" int miseg_conv1x1_bwd(void* stream, int dt, const void* in, const float* gout, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                 const float* w, int64_t Cout, void* gin, float* gw, float* gbias, void* ws, int64_t ws_bytes) {
    MISEG_TAPE(miseg_conv1x1_bwd, stream, dt, in, gout, N, H, W, Cin, w, Cout, gin, gw, gbias, ws, ws_bytes);
    MISEG_F16_DISPATCH_ON(dt, miseg_conv1x1_bwd, stream, MISEG_BF16, in, gout, N, H, W, Cin, w, Cout, gin, gw, gbias, ws, ws_bytes);
    MISEG_REQUIRE(in && gout && w && gw && gbias && ws, "conv1x1_bwd: null pointer");
    MISEG_REQUIRE(Cin == 16, "conv1x1_bwd: Cin must be 16");
    MISEG_REQUIRE(ws_bytes >= miseg_conv1x1_bwd_ws_bytes(N, H, W, Cin, Cout), "conv1x1_bwd: workspace too small");
    const int64_t npix = N * H * W;
    const int nb = c1_blocks(npix);
    hipStream_t st = as_stream(stream);
#define L(CO)                                                                                                                                  \
    if (dt == MISEG_F32) hipLaunchKernelGGL((conv1x1_bwd_kernel<float, 16, CO>), dim3(nb), dim3(256), 0, st, (const float*)in, gout, npix, w, (float*)gin, (float*)ws); \
    else hipLaunchKernelGGL((conv1x1_bwd_kernel<bf16, 16, CO>), dim3(nb), dim3(256), 0, st, (const bf16*)in, gout, npix, w, (bf16*)gin, (float*)ws)
    switch (Cout) {
        case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break;
        case 5: L(5); break; case 8: L(8); break;
        default: return fail(MISEG_E_INVALID, "conv1x1_bwd: unsupported Cout %ld", (long)Cout);
    }
#undef L
    MISEG_LAUNCH_CHECK("conv1x1_bwd_kernel");
    const int len = (int)(Cout * Cin + Cout);
    // the blocks' [gw | gbias] partial vectors, summed straight into the two gradients
    hipLaunchKernelGGL(sum_parts2_kernel, dim3(reduce_grid(len, nb)), dim3(256), 0, st, (const float*)ws, nb, len, (int)(Cout * Cin), gw, gbias);
    MISEG_LAUNCH_CHECK("sum_parts2_kernel");
    return MISEG_OK;
}
