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

constexpr int kBT = 512;        // 8 waves = 2 quads, two waves per SIMD: each SIMD hosts one wave of each quad

template <int PAD> constexpr int T_of() { return 2 * PAD + 1; }

template <int MT, int NT, int PAD, int NTERMS, int ROLE>
__device__ __forceinline__ void joint_bf16_body(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ mask,
                                                const JointGeom& g, const int32_t* __restrict__ win, float* __restrict__ partials,
                                                unsigned char* lds) {
    typedef QuadTiles<MT, NT, ROLE> TS;
    constexpr int T = 2 * PAD + 1, NP = NTERMS == 1 ? 1 : 2, RBY = BRB + 2 * PAD;
    constexpr int K = 20;                                  // the shipped cluster count (host check): divisions by K fold to multiplies
    // LDS carve (bf16 = unsigned short)
    unsigned short* Xr = reinterpret_cast<unsigned short*>(lds);                 // [NP][K][BRB][BXW]
    unsigned short* Ys = Xr + (size_t)NP * K * BRB * BXW;                        // [NP][K][RBY][BYW]
    unsigned short* Asb = Ys + (size_t)NP * K * RBY * BYW;                       // [2 quads][2 buffers][NP][T][K][BAW]
    const size_t xPlane = (size_t)K * BRB * BXW, yPlane = (size_t)K * RBY * BYW, aPlane = (size_t)T * K * BAW;
    const int tid = threadIdx.x, lane = tid & 63, pair = tid >> 8, ptid = tid & 255, wv = tid >> 6;   // pair = quad index
    const int l15 = lane & 15, q = lane >> 4;
    unsigned short* As = Asb + (size_t)pair * 2 * NP * aPlane;                   // this quad's two buffers

    const int slot = blockIdx.y;                          // one sub-block only on this path (T*K <= 144): slot = sub-head * P + window
    const int shead = slot / g.P, p = slot - shead * g.P;
    x += (size_t)shead * g.hs;
    y += (size_t)shead * g.hs;
    const int h0 = win[p * 4 + 0], h1 = win[p * 4 + 1], w0 = win[p * 4 + 2], w1 = win[p * 4 + 3];
    const int tr = (h1 - h0 + BRB - 1) / BRB, tc = (w1 - w0 + BKW - 1) / BKW;
    const int nItems = g.N * tr * tc;
    static_assert(MT == (T_of<PAD>() * 20 + 15) / 16 && NT == MT, "tile counts are tied to K = 20");

    auto aoff = [&](int mt) {                             // m = dx*K + i is exactly the As row index; swizzled 16-B slot
        const int m = min(mt * 16 + l15, g.Mdim - 1);
        return m * BAW + 8 * (q ^ ((m >> 2) & 3));
    };
    auto boff = [&](int nt) {
        const int c = min(nt * 16 + l15, g.Mdim - 1);
        return ((c % K) * RBY + (2 * PAD - c / K)) * BYW + 8 * q;
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t plane = (size_t)g.H * g.W;
    // ---- register-staged prefetch.  Every load is a b64 (two adjacent pixels per lane) and every LDS store a packed bf16 pair:
    // half the VMEM and DS instructions of a pixel-per-lane scheme (the commit was LDS-store bound: 720 ds_write_b16 per item).
    //   X main  : unit U = wave + 8*bi covers rows pr = 2U, 2U+1 (lane half) x 64 px; (ch, r) = (U / 2 + ..., 2 (U & 1) + half)
    //   X extra : unit E covers 8 rows x 16 px (8 lanes each): E = wave (and 8 + wave for waves 0, 1)
    //   Y       : unit U = wave + 8*bi covers rows 2U, 2U+1 of the [K][RBY] row list (RBY even: both in one channel)
    // Row patterns repeat with a fixed channel stride (see YPER below), so addresses are running sums; one descriptor per sample
    // makes channels >= K read 0, an invalid row gets the out-of-range marker as its base.
    static_assert(BRB == 4 && RBY % 2 == 0, "row pairing assumes 4 X rows and an even number of Y rows per channel");
    constexpr int XB = 5, XE = 2, YH = RBY / 2;             // X units per wave (K*4/2/8 <= 5), extra units, Y row pairs per channel
    constexpr int YPER = YH % 8 == 0 ? 1 : YH % 4 == 0 ? 2 : YH % 2 == 0 ? 4 : YH;   // lcm(8, YH) / 8
    constexpr int YCST = 8 * YPER / YH;                     // channels advanced per period
    constexpr int YB = (20 * YH + 7) / 8;                   // Y units per wave for K <= 20
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    f32x2 xv[XB], xe[XE], yv[YB];
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    (void)tbytes;
    constexpr unsigned OOB = 0xC0000000u;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned pl4 = (unsigned)(plane * 4);
    const int half = lane >> 5, l31 = lane & 31;
    int pcol0 = 0;                                          // column origin of the prefetched item (for the edge masks at commit)
    auto ld2 = [&](__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vo, (int)so, 0);
        return f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
    };
    auto prefetch = [&](int it) {
        const int ct = it % tc, rt = (it / tc) % tr, n = it / (tc * tr);
        const int row0 = h0 + rt * BRB, col0 = w0 + ct * BKW;
        pcol0 = col0;
        const unsigned sbytes = (unsigned)K * pl4;
        const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * K * plane), 0, (int)sbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * K * plane), 0, (int)sbytes, 0x00020000);
        const unsigned wb = (unsigned)g.W * 4u;
        {   // X main: U = wvu + 8*bi: ch = U >> 1 = (wvu >> 1) + 4*bi, r = 2 * (wvu & 1) + half
            const int r = 2 * (wvu & 1) + half, row = row0 + r, col = col0 - 8 + 2 * l31;
            const bool ok = row < h1 && col + 1 >= w0 && col < w1 && col >= 0;
            const unsigned vo = ok ? (unsigned)row * wb + (unsigned)col * 4u : OOB;
            unsigned so = (unsigned)(wvu >> 1) * pl4;
#pragma unroll
            for (int bi = 0; bi < XB; ++bi) {
                xv[bi] = ld2(rx_, vo, so);
                so += 4 * pl4;
            }
        }
#pragma unroll
        for (int e = 0; e < XE; ++e) {      // X extra: pr = 8E + (lane >> 3): ch = 2E + (lane >> 5), r = (lane >> 3) & 3; px 64 + 2*(lane & 7)
            const int E = wvu + 8 * e, ch = 2 * E + half, r = (lane >> 3) & 3, row = row0 + r, col = col0 + 56 + 2 * (lane & 7);
            const bool ok = E < 10 && ch < K && row < h1 && col + 1 >= w0 && col < w1;
            const unsigned vo = ok ? (unsigned)ch * pl4 + (unsigned)row * wb + (unsigned)col * 4u : OOB;
            xe[e] = ld2(rx_, vo, 0);
        }
        {   // Y: U = wvu + 8*bi, bi = YPER*u + v: channel c_v + YCST*u, row pair q_v
            const int col = col0 + 2 * l31;
            const unsigned vc = col < w1 ? (unsigned)col * 4u : OOB;
#pragma unroll
            for (int v = 0; v < YPER; ++v) {
                const int U = wvu + 8 * v, ch = U / YH, r = 2 * (U - ch * YH) + half, row = row0 - PAD + r;
                const unsigned vo = (vc != OOB && row >= h0 && row < h1) ? (unsigned)row * wb + vc : OOB;
                unsigned so = (unsigned)ch * pl4;
#pragma unroll
                for (int u = 0; u * YPER + v < YB; ++u) {
                    yv[u * YPER + v] = ld2(ry_, vo, so);
                    so += YCST * pl4;
                }
            }
        }
    };
    // split a pixel pair into packed (hi, hi) and (lo, lo) bf16 words
    auto store2 = [&](f32x2 v, bool k0, bool k1, unsigned short* d, size_t planeStride) {
        const float a = k0 ? v[0] : 0.f, b = k1 ? v[1] : 0.f;
        const unsigned short ha = f32_to_bf16_bits(a), hb = f32_to_bf16_bits(b);
        *reinterpret_cast<unsigned*>(d) = (unsigned)ha | ((unsigned)hb << 16);
        if (NP == 2)
            *reinterpret_cast<unsigned*>(d + planeStride) = (unsigned)f32_to_bf16_bits(a - bf16_bits_to_f32(ha)) |
                                                            ((unsigned)f32_to_bf16_bits(b - bf16_bits_to_f32(hb)) << 16);
    };
    auto commit = [&]() {
        {
            const int r = 2 * (wvu & 1) + half, col = pcol0 - 8 + 2 * l31;
            const bool k0 = col >= w0 && col < w1, k1 = col + 1 >= w0 && col + 1 < w1;
#pragma unroll
            for (int bi = 0; bi < XB; ++bi) {
                const int ch = (wvu >> 1) + 4 * bi;
                if (ch < K) store2(xv[bi], k0, k1, Xr + ((size_t)ch * BRB + r) * BXW + 2 * l31, xPlane);
            }
        }
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int E = wvu + 8 * e, ch = 2 * E + half, r = (lane >> 3) & 3, col = pcol0 + 56 + 2 * (lane & 7);
            const bool k0 = col >= w0 && col < w1, k1 = col + 1 >= w0 && col + 1 < w1;
            if (E < 10 && ch < K) store2(xe[e], k0, k1, Xr + ((size_t)ch * BRB + r) * BXW + 64 + 2 * (lane & 7), xPlane);
        }
        {
            const int col = pcol0 + 2 * l31;
            const bool k0 = col < w1, k1 = col + 1 < w1;
#pragma unroll
            for (int bi = 0; bi < YB; ++bi) {
                const int u = bi / YPER, v = bi % YPER;
                const int U = wvu + 8 * v, c0 = U / YH, r = 2 * (U - c0 * YH) + half, ch = c0 + YCST * u;
                if (ch < K) store2(yv[bi], k0, k1, Ys + ((size_t)ch * RBY + r) * BYW + 2 * l31, yPlane);
            }
        }
    };

    if ((int)blockIdx.x < nItems) prefetch(blockIdx.x);
    for (int it = blockIdx.x; it < nItems; it += g.G) {
        __syncthreads();                 // previous tile fully consumed
        commit();
        __syncthreads();
        const bool more_items = it + g.G < nItems;
        // The next item's loads are issued by quad 0 before its first MFMA phase and by quad 1 after it (same staggering as the
        // shifted copies below): ~3 k cycles of address arithmetic and VMEM issue per wave that would otherwise idle the matrix pipe.
        if (more_items && pair == 0) prefetch(it + g.G);
        // ---- materialise the T shifted copies of row rx, pixels [32ks, 32ks+32) into As buffer `buf`:
        //      As[pl][dx][i][k] = Xr[pl][i][rx][8+32ks+k+dx-PAD]
        constexpr int NSTEP = (BRB / 2) * (BKW / 32);
        auto materialise = [&](int step, int buf) {
            const int rx = pair + 2 * (step / (BKW / 32)), ks = step % (BKW / 32);
            unsigned short* Ab = As + (size_t)buf * NP * aPlane;
            for (int task = ptid; task < NP * K * 4; task += 256) {
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
                    *reinterpret_cast<u32x4*>(Ab + pl * aPlane + (size_t)m * BAW + 8 * (c8 ^ ((m >> 2) & 3))) = o;
                }
            }
        };
        // As is double-buffered: step s multiplies out of buffer s & 1 while the copies for step s+1 are written into the other
        // one -- by quad 0 BEFORE its MFMAs and by quad 1 AFTER them.  A SIMD hosts one wave of each quad, so in each half of a
        // step one of its waves is on the matrix pipe and the other on the LDS; one barrier per step.
        materialise(0, 0);
        __syncthreads();
#pragma unroll 1
        for (int step = 0; step < NSTEP; ++step) {
            const int rx = pair + 2 * (step / (BKW / 32)), ks = step % (BKW / 32);
            const unsigned short* Ab = As + (size_t)(step & 1) * NP * aPlane;
            if (pair == 0 && step + 1 < NSTEP) materialise(step + 1, (step + 1) & 1);
            bf16x8_t af[NP][MT];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (TS::row_used(mt)) af[pl][mt] = *reinterpret_cast<const bf16x8_t*>(Ab + pl * aPlane + aoff(mt));
            // B fragments one tile column ahead of the MFMAs (two register slots): a column's 2-3 accumulator tiles are only
            // ~100 cycles of matrix work, less than an LDS read's latency.
            constexpr int BQ = NTERMS == 3 ? 2 : 4;                           // register slots: BQ - 1 columns in flight (x3: register budget)
            bf16x8_t bq[BQ][NP];
            auto load_b = [&](int nt, bf16x8_t* dst) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    dst[pl] = *reinterpret_cast<const bf16x8_t*>(Ys + pl * yPlane + boff(nt) + rx * BYW + 32 * ks);
            };
#pragma unroll
            for (int j = 0; j < BQ - 1; ++j)
                if (TS::nth_col(j) < NT) load_b(TS::nth_col(j), bq[j]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (TS::col_used(nt)) {
                    constexpr int dummy = 0; (void)dummy;
                    const int rank = TS::col_rank(nt), slot = rank % BQ;       // compile-time after unrolling
                    if (TS::nth_col(rank + BQ - 1) < NT) load_b(TS::nth_col(rank + BQ - 1), bq[(rank + BQ - 1) % BQ]);
                    __builtin_amdgcn_sched_barrier(0);     // or the scheduler sinks the read next to its use
                    {   // MT = NT = ceil(T*K/16) exactly (K compile-time, checked on the host): no runtime tile guards -> straight-line MFMAs
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (TS::mine(mt, nt)) {
                                if (NTERMS == 3) {
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1][mt], bq[slot][0], acc[mt][nt], 0, 0, 0);
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bq[slot][NP - 1], acc[mt][nt], 0, 0, 0);
                                }
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bq[slot][0], acc[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (pair == 1 && step + 1 < NSTEP) materialise(step + 1, (step + 1) & 1);
            if (step == 0 && more_items && pair == 1) prefetch(it + g.G);
            __syncthreads();   // step s+1's copies complete; buffer s & 1 free for step s+2
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
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3) {
        case 0: joint_bf16_body<MT, NT, PAD, NTERMS, 0>(x, y, mask, g, win, partials, ldsb); break;
        case 1: joint_bf16_body<MT, NT, PAD, NTERMS, 1>(x, y, mask, g, win, partials, ldsb); break;
        case 2: joint_bf16_body<MT, NT, PAD, NTERMS, 2>(x, y, mask, g, win, partials, ldsb); break;
        default: joint_bf16_body<MT, NT, PAD, NTERMS, 3>(x, y, mask, g, win, partials, ldsb); break;
    }
}

static size_t bf16_lds_bytes(const JointGeom& g, int nterms) {
    const int np = nterms == 1 ? 1 : 2, T = g.T, rby = BRB + 2 * g.pad;
    const size_t tiles = ((size_t)np * g.K * BRB * BXW + (size_t)np * g.K * rby * BYW + (size_t)2 * 2 * np * T * g.K * BAW) * 2;
    const int cap = g.tilesM <= 4 ? 4 : 9;
    const size_t dred = (size_t)(cap * 16) * (cap * 16) * 4;
    return tiles > dred ? tiles : dred;
}

bool joint_fwd_bf16_supported(const JointGeom& g) {
    if (g.sb != 1 || g.K != 20) return false;
    if (!((g.pad == 3 && g.tilesM > 4 && g.tilesM <= 9) || (g.pad == 1 && g.tilesM <= 4))) return false;
    if ((size_t)g.N * g.K * g.H * g.W * 4 >= 0x40000000ull) return false;   // 32-bit buffer offsets with an out-of-range marker
    if (g.K * BRB > 80 || g.K * (BRB + 2 * g.pad) > 200) return false;   // prefetch register batches (XB, YB rows per wave, 8 waves)
    return bf16_lds_bytes(g, 3) <= (size_t)kLdsBudget;
}

int launch_joint_fwd_bf16(hipStream_t st, const float* x, const float* y, const float* mask, const JointGeom& g, const int32_t* win,
                          float* partials, int nterms) {
    // MISEG_FWD_KERNEL=copies: this file's kernel (x-shifted copies in LDS); default: pixel-major operands (mi_local_fwd_px.hip)
    static const bool copies = [] { const char* e = getenv("MISEG_FWD_KERNEL"); return e && !strcmp(e, "copies"); }();
    if (!copies) return launch_joint_fwd_px(st, x, y, g, win, partials, nterms);
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
