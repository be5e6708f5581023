// Local-MI displacement joint on the bf16 matrix cores, pixel-major operands (no shifted copies).
//
// Same GEMM and the same partial layout as mi_local_bf16.hip,
//     D[(dx,i),(dy,j)] += sum_{n,r,w} X[n,i,r,w + dx - p] * Y[n,j,r + p - dy,w]           M = N = T K (140), reduction over pixels
// (ref contrastyou/losses/iic_loss.py:120-123: the conv2d of the permuted maps), hi/lo operand split for fp32-class accuracy.
// That kernel keeps X planar ([channel][pixel]) and therefore has to MATERIALISE the T x-shifted copies of every row chunk in LDS
// (funnel shifts in registers, 18 KB of LDS stores and a workgroup barrier per 32-pixel k-step): its matrix pipe is 34 % busy
// (profiles/r02_pmc.json), with 3.2 vector instructions per MFMA and a third of its LDS cycles lost to bank conflicts.
// Here the rows sit PIXEL-MAJOR in LDS ([pixel][20 channels], bf16 hi / lo planes) and every fragment is two
// ds_read_b64_tr_b16: a source lane reads 4 consecutive CHANNELS of one pixel, the transpose hands lane m its 4 consecutive
// PIXELS -- so a displacement in x is just a different pixel ADDRESS ((k + dx) * 40 B, always 8-byte aligned) and a displacement
// in y a different row of the Y ring.  No copies, no funnel shifts, one barrier per PAIR of image rows.
//   * K = 20 = 16 + 4: tiles 0..T-1 of M are (dx, i < 16), the 4 remaining channels of all T displacements share ceil(4T/16)
//     tiles whose lane group (the transpose's 4-lane source group) is the displacement -- per-lane addresses, nothing else; same
//     on the N side with dy.  9 x 9 tiles for pad 3 (the same count as the (dx*K + i) stacking), written out in the old
//     (dx*K + i, dy*K + j) order so joint_reduce_kernel is shared.
//   * a block walks DOWN a 64-pixel strip: quad 0 (waves 0-3) takes image row r, quad 1 row r+1; both read one ring of Y rows
//     (r - p .. r + 1 + p live, the next two being filled); the tiles of D are split over the four waves of a quad in 2 x 2 blocks
//     (BlockTiles below): two waves per SIMD, one of each quad.
//   * the next row pair travels global -> registers during the MFMAs (4 channels of a pixel per lane, so the LDS store is one
//     ds_write_b64 per plane) and is committed after them.
#include "mi_local.h"

namespace miseg {

typedef __bf16 pbf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int pu32x2 __attribute__((ext_vector_type(2)));
#define PX_LDS_S16X4(ptr) ((__attribute__((address_space(3))) s16x4*)(ptr))

constexpr int kPT = 512;   // 8 waves: quad = wave >> 2, role = wave & 3 (quad 0) or 3 - (wave & 3) (quad 1)

// The MT x NT tiles of D cut into 2 x 2 near-square blocks, one per role: a wave reads the A fragments of ~MT/2 tile rows and the B
// fragments of ~NT/2 tile columns per k-step (row-major runs, QuadTiles: ~MT/4 + 1 rows but ALL NT columns -- a quarter more
// LDS reads, and the reads are what bounds this kernel).  The blocks are unequal (25 / 20 / 20 / 16 tiles at 9 x 9); quad 1 deals
// them to its waves in reverse order, so the two waves that share a SIMD carry 41 / 40 / 40 / 41 tiles together.
template <int MT, int NT, int ROLE>
struct BlockTiles {
    static constexpr int RS = (MT + 1) / 2, CS = (NT + 1) / 2;
    static constexpr bool mine(int m, int n) { return ((m >= RS ? 2 : 0) + (n >= CS ? 1 : 0)) == ROLE; }
    static constexpr bool row_used(int m) { return (m >= RS) == ((ROLE & 2) != 0); }
    static constexpr bool col_used(int n) { return (n >= CS) == ((ROLE & 1) != 0); }
    static constexpr int nth_col(int j) {       // j-th used column, NT if there are fewer
        int r = 0;
        for (int k = 0; k < NT; ++k)
            if (col_used(k)) {
                if (r == j) return k;
                ++r;
            }
        return NT;
    }
    static constexpr int col_rank(int n) {      // how many used columns precede n
        int r = 0;
        for (int k = 0; k < n; ++k) r += col_used(k) ? 1 : 0;
        return r;
    }
};

template <int PAD>
struct PX {
    static constexpr int K = 20, T = 2 * PAD + 1, RT = (T * 4 + 15) / 16, MT = T + RT, DN = MT * 16;
    static constexpr int WT = 64, KS = WT / 32, XW = WT + 2 * PAD, XWP = (XW + 7) / 8 * 8, RING = 2 * PAD + 4;
    // a row is TWO pixel-major arrays: classes 0..15 (32 B per pixel) and classes 16..19 (8 B per pixel).  With 32 B per pixel and
    // the k <-> pixel map below, the 32 lanes ds_read_b64_tr_b16 serves per LDS cycle read 8 consecutive pixels = all 64 banks once.
    // The 4-class array of Y is read with the source lane group pq selecting the RING ROW (displacement 4 t + pq): its row stride is
    // padded by 64 B so that the four groups of a half-wave land in four different 64-byte bank windows (unpadded, 512 B apart, they
    // collided 4-way: 39 % of this kernel's LDS cycles were bank conflicts, profiles/r02_pmc.json at library version 201).
    static constexpr int CM = 16, CR = 4;
    static constexpr int YROWM = WT * CM, YROWR = WT * CR + 32, XROWM = XWP * CM, XROWR = XWP * CR;     // bf16 elements per row buffer
};

struct PxGeom {
    int N, H, W, P, S, G;
    long long hs;
};

// PLW: every probability this kernel stages is also written out as the backward's operand planes (miseg_hip.h, "local-MI operand
// planes"; mi_local.h split_quad): Y rows by the Y tasks, the strip's own 64 columns of the X rows by the X tasks -- each element of
// the window at least once (rows on a segment boundary twice, with the same bytes).  Maps: x = s * 2N + n, y = s * 2N + N + n.
template <int PAD, int NTERMS, int ROLE, bool PLW>
__device__ __forceinline__ void joint_px_body(const float* __restrict__ x, const float* __restrict__ y, const PxGeom& g,
                                              const int32_t* __restrict__ win, float* __restrict__ partials, unsigned char* lds,
                                              const MiPlanes pln) {
    typedef PX<PAD> C;
    constexpr int K = C::K, T = C::T, RT = C::RT, MT = C::MT, NT = C::MT, NP = NTERMS == 1 ? 1 : 2, WT = C::WT, RING = C::RING, KS = C::KS;
    constexpr int CM = C::CM, CR = C::CR;
    typedef BlockTiles<MT, NT, ROLE> TS;
    constexpr size_t ymPlane = (size_t)RING * C::YROWM, yrPlane = (size_t)RING * C::YROWR, xmPlane = (size_t)4 * C::XROWM, xrPlane = (size_t)4 * C::XROWR;
    unsigned short* Ym = reinterpret_cast<unsigned short*>(lds);                     // [NP][RING][WT][16]
    unsigned short* Yr = Ym + (size_t)NP * ymPlane;                                   // [NP][RING][WT][4]
    unsigned short* Xm = Yr + (size_t)NP * yrPlane;                                   // [NP][4][XWP][16]   (buffer = 2 * parity + quad)
    unsigned short* Xr = Xm + (size_t)NP * xmPlane;                                   // [NP][4][XWP][4]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4, qq = l15 >> 2, pq = l15 & 3;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), quad = wv >> 2;

    const int slot = blockIdx.y, shead = slot / g.P, p = slot - shead * g.P;
    x += (size_t)shead * g.hs;
    y += (size_t)shead * g.hs;
    const int h0 = win[p * 4 + 0], h1 = win[p * 4 + 1], w0 = win[p * 4 + 2], w1 = win[p * 4 + 3];
    // Work of a slot = the row PAIRS of its N * tc column strips, in (image, strip, row) order; the G blocks of the slot take equal
    // contiguous shares of that list (to within one pair), so a block walks down a strip for as long as its share lasts and warms the
    // Y ring only where it starts or changes strip (2-3 times, not once per fixed-size segment) -- and no block has a segment more
    // than the others (fixed 32-row segments dealt round-robin: 11 for some blocks, 10 for the rest = 9 % of the launch idle).
    const int tc = (w1 - w0 + WT - 1) / WT, pps = (h1 - h0 + 1) / 2;
    const long long totalPairs = (long long)g.N * tc * pps;
    const int qlo = (int)(totalPairs * blockIdx.x / g.G), qhi = (int)(totalPairs * (blockIdx.x + 1) / g.G);
    const size_t plane = (size_t)g.H * g.W;
    const unsigned pl4 = (unsigned)(plane * 4), wb = (unsigned)g.W * 4u;
    constexpr unsigned OOB = 0xC0000000u;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- global -> registers -> LDS.  Task t of a row pair: (row kind, channel quad c4 = 4 consecutive channels); a lane is a pixel
    // and loads its 4 channels from 4 planes (each load coalesced across the wave), so the pixel-major store is one 8-byte word per
    // plane.  Kinds: 0 / 1 = the two new Y rows, 2 / 3 = the two X rows (64 pixels from column col0 - PAD), 4 = the X rows' tails
    // (lane = (c4, pixel 64 ..)).  Tasks are dealt round-robin to the 8 waves: 20 + 2 tasks -> 3 per wave at most.
    // t = wave + 8 i never changes, so everything that depends on the task alone -- which array, which word of a row buffer, the
    // channel offset -- is worked out ONCE here; a segment adds the column part, a row pair only `row * W`.  (Deriving all of it per
    // row pair from t cost ~800 mostly scalar, branchy instructions per wave and pair: the matrix pipe sat idle for half the kernel.)
    constexpr int NTASK = 22, TPW = (NTASK + 7) / 8;
    float pfA[TPW][4], pfB[TPW][4];                  // two register sets: the loads of pair i + 2 fly while those of pair i + 1 are committed
    bool act[TPW], isY[TPW], tail[TPW];                  // wave-uniform
    int odd[TPW];                                        // second row of the pair (wave-uniform)
    unsigned chan_so[TPW];                               // bytes: first channel of the task (scalar part of the load address)
    unsigned chan_vo[TPW];                               // bytes: lane-dependent channel offset (tails only)
    unsigned dst0[TPW], lo_off[TPW], rstride[TPW];       // LDS bytes: the lane's word in row buffer 0 of its array (hi plane), hi -> lo, buffer stride
    bool lane_st[TPW];                                   // this lane has a word to store
    const int tl_c4 = lane / (2 * PAD), tl_px = 64 + lane % (2 * PAD);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wv + 8 * i;
        act[i] = t < NTASK;
        tail[i] = t >= 20;
        const int kind = t < 20 ? t / 5 : 4, c4u = t < 20 ? t % 5 : 0;
        isY[i] = kind < 2;
        odd[i] = t < 20 ? (kind & 1) : t - 20;
        const int c4v = tail[i] ? tl_c4 : c4u, px = tail[i] ? tl_px : lane;
        const bool mainArr = c4v < 4;
        chan_so[i] = tail[i] ? 0u : (unsigned)(c4u * 4) * pl4;
        chan_vo[i] = tail[i] ? (unsigned)(tl_c4 * 4) * pl4 : 0u;
        const unsigned short* base = isY[i] ? (mainArr ? Ym : Yr) : (mainArr ? Xm : Xr);
        dst0[i] = (unsigned)((base - reinterpret_cast<unsigned short*>(lds)) + (mainArr ? px * CM + c4v * 4 : px * CR)) * 2u;
        lo_off[i] = (unsigned)(isY[i] ? (mainArr ? ymPlane : yrPlane) : (mainArr ? xmPlane : xrPlane)) * 2u;
        rstride[i] = (unsigned)(isY[i] ? (mainArr ? C::YROWM : C::YROWR) : (mainArr ? C::XROWM : C::XROWR)) * 2u;
        lane_st[i] = act[i] && (!tail[i] || lane < 5 * 2 * PAD);
    }
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    // two descriptors, chosen by a wave-uniform branch: a descriptor picked through a per-task pointer made the compiler wrap every
    // load in a waterfall loop (readfirstlane / saveexec / branch per load)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)tbytes, 0x00020000);
    unsigned loff[TPW], seg_so[TPW];                     // per segment: the lane's column (bytes, OOB if it has nothing to load); image + channel
    auto seg_setup1 = [&](int n, int col0) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int col = tail[i] ? col0 - PAD + tl_px : (isY[i] ? col0 + lane : col0 - PAD + lane);
            const bool okc = lane_st[i] && col < w1 && (isY[i] || col >= w0);
            loff[i] = okc ? (unsigned)col * 4u + chan_vo[i] : OOB;
            seg_so[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)n * (unsigned)K * pl4 + chan_so[i]));
        }
    };
    // rows ybase, ybase + 1 of Y and xbase, xbase + 1 of X (rows outside the window read zeros: OOB + anything stays out of range)
    auto prefetch1 = [&](float (&pf)[TPW][4], int ybase, int xbase) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int row = __builtin_amdgcn_readfirstlane((isY[i] ? ybase : xbase) + odd[i]);
            const unsigned radd = (row >= h0 && row < h1) ? (unsigned)row * wb : OOB;
            const unsigned vo = act[i] ? loff[i] + radd : OOB;
            const unsigned so = seg_so[i];
#ifdef MISEG_PX_NOLOAD
            for (int c = 0; c < 4; ++c) pf[i][c] = __uint_as_float(vo + so + c);
            continue;
#endif
            if (isY[i]) {
#pragma unroll
                for (int c = 0; c < 4; ++c) pf[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsY, (int)vo, (int)(so + (unsigned)c * pl4), 0));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) pf[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, (int)vo, (int)(so + (unsigned)c * pl4), 0));
            }
        }
    };
    auto commit1 = [&](float (&pf)[TPW][4], int ybase, int xpar) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (!act[i]) continue;                        // wave-uniform
            const int buf = __builtin_amdgcn_readfirstlane(isY[i] ? (ybase + odd[i] + 4 * RING) % RING : 2 * xpar + odd[i]);
            unsigned char* dst = lds + dst0[i] + (unsigned)buf * rstride[i];
            unsigned hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hi[c] = f32_to_bf16_bits(pf[i][c]);
                lo[c] = f32_to_bf16_bits(pf[i][c] - bf16_bits_to_f32((unsigned short)hi[c]));
            }
            if (lane_st[i]) {                             // divergent for the tails only
                *reinterpret_cast<pu32x2*>(dst) = pu32x2{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
                if (NP == 2) *reinterpret_cast<pu32x2*>(dst + lo_off[i]) = pu32x2{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
            }
        }
    };

    // ---- PLW: the same rows through a second loader, dealt so that the plane stores are contiguous.  Wave w stages row kind w >> 1 (Y,
    // Y + 1, X, X + 1), pixels 32 (w & 1) .. + 31; a lane is (pixel, half of its 20 channels): half 0 loads classes 0..11 (quads 0, 1, 2),
    // half 1 classes 12..19 (quads 3, 4) -- and, in the first wave of an X row, quad ti % 5 of tail pixel 64 + ti / 5 in its spare four
    // registers (ti = lane >> 1 < 10 PAD).  Loads stay coalesced (32 consecutive pixels per plane and half-wave); a pixel's 40 + 24 + 24
    // plane bytes leave from two neighbouring lanes, 1 280 + 2 x 768 contiguous bytes per wave.  (With the task loader above the planes
    // left as 8-byte pieces 40 bytes apart: +170 us on the cfg2 launch instead of +70.  Without the plane stores this loader is the
    // slower one by 4 %: it is used for PLW only.)
    static_assert(!PLW || NP == 2, "the plane by-product belongs to the three-term forward");
    const int kind2 = wv >> 1, odd2 = kind2 & 1, half = lane & 1, pxl = (wv & 1) * 32 + (lane >> 1), ti = lane >> 1;
    const bool isY2 = kind2 < 2, tailw = wv >= 4 && !(wv & 1), tlane = tailw && half && ti < 5 * 2 * PAD;
    const int tp = 64 + ti / 5, tq = ti % 5;
    const unsigned ldsYm = 0u, ldsYr = (unsigned)(NP * ymPlane) * 2u, ldsXm = ldsYr + (unsigned)(NP * yrPlane) * 2u, ldsXr = ldsXm + (unsigned)(NP * xmPlane) * 2u;
    const unsigned bM = isY2 ? ldsYm : ldsXm, bR = isY2 ? ldsYr : ldsXr;
    const unsigned strideM = (unsigned)(isY2 ? C::YROWM : C::XROWM) * 2u, strideR = (unsigned)(isY2 ? C::YROWR : C::XROWR) * 2u;
    const unsigned loM = (unsigned)(isY2 ? ymPlane : xmPlane) * 2u, loR = (unsigned)(isY2 ? yrPlane : xrPlane) * 2u;
    // quads A (0 | 3), B (1 | 4), C (2 | a tail quad): LDS byte address in row buffer 0 (hi plane), hi -> lo, buffer stride
    const unsigned dA = bM + (unsigned)(pxl * CM + (half ? 12 : 0)) * 2u;
    const unsigned dB = half ? bR + (unsigned)(pxl * CR) * 2u : bM + (unsigned)(pxl * CM + 4) * 2u;
    const unsigned dC = half ? (tq < 4 ? ldsXm + (unsigned)(tp * CM + tq * 4) * 2u : ldsXr + (unsigned)(tp * CR) * 2u) : bM + (unsigned)(pxl * CM + 8) * 2u;
    const unsigned loB = half ? loR : loM, sB = half ? strideR : strideM;
    const unsigned loC = half ? (tq < 4 ? (unsigned)xmPlane * 2u : (unsigned)xrPlane * 2u) : loM;
    const unsigned sC = half ? (tq < 4 ? (unsigned)C::XROWM * 2u : (unsigned)C::XROWR * 2u) : strideM;
    const bool hasC = !half || tlane;
    const unsigned nmaps = (unsigned)g.S * 2u * (unsigned)g.N;
    const __amdgpu_buffer_rsrc_t rsP16 = __builtin_amdgcn_make_buffer_rsrc((void*)pln.p16, 0, PLW ? (int)(nmaps * (unsigned)plane * 40u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP8L = __builtin_amdgcn_make_buffer_rsrc((void*)pln.p8l, 0, PLW ? (int)(nmaps * (unsigned)plane * 24u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP8H = __builtin_amdgcn_make_buffer_rsrc((void*)pln.p8h, 0, PLW ? (int)(nmaps * (unsigned)plane * 24u) : 0, 0x00020000);
    unsigned voA = OOB, voC = OOB, seg_so2 = 0u, pmap2 = 0u;                  // per segment: load offsets (bytes), image, first row of the map's planes
    unsigned pv16 = OOB, pvC16 = OOB, pv8 = OOB, pvC8 = OOB, pvZ8 = OOB, pvZT = OOB;   // plane byte offsets inside a plane row (OOB: not this strip's pixel)
    auto seg_setup2 = [&](int n, int col0) {
        const int col = isY2 ? col0 + pxl : col0 - PAD + pxl, colT = col0 - PAD + tp;
        const bool okc = col < w1 && (isY2 || col >= w0), okT = tlane && colT < w1 && colT >= w0;
        const bool own = okc && col >= col0 && col < col0 + WT, ownT = okT && colT < col0 + WT;
        voA = okc ? (unsigned)col * 4u + (half ? 12u : 0u) * pl4 : OOB;
        voC = half ? (okT ? (unsigned)colT * 4u + (unsigned)(tq * 4) * pl4 : OOB) : (okc ? (unsigned)col * 4u + 8u * pl4 : OOB);
        seg_so2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)n * (unsigned)K * pl4));
        pmap2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((((unsigned)shead * 2u + (isY2 ? 1u : 0u)) * (unsigned)g.N + (unsigned)n) * (unsigned)g.H));
        pv16 = own ? (unsigned)col * 40u + (half ? 24u : 0u) : OOB;
        pvC16 = half ? (ownT ? (unsigned)colT * 40u + (unsigned)tq * 8u : OOB) : (own ? (unsigned)col * 40u + 16u : OOB);
        pv8 = own ? (unsigned)col * 24u + (half ? 12u : 0u) : OOB;
        pvC8 = half ? (ownT ? (unsigned)colT * 24u + (unsigned)tq * 4u : OOB) : (own ? (unsigned)col * 24u + 8u : OOB);
        pvZ8 = half && own ? (unsigned)col * 24u + 20u : OOB;
        pvZT = ownT && tq == 4 ? (unsigned)colT * 24u + 20u : OOB;
    };
    auto prefetch2 = [&](float (&pf)[TPW][4], int ybase, int xbase) {
        const int row = __builtin_amdgcn_readfirstlane((isY2 ? ybase : xbase) + odd2);
        const unsigned radd = (row >= h0 && row < h1) ? (unsigned)row * wb : OOB;
        const unsigned va = voA + radd, vc = voC + radd;
#ifdef MISEG_PX_NOLOAD
        for (int j = 0; j < 12; ++j) pf[j >> 2][j & 3] = __uint_as_float(va + vc + j);
        return;
#endif
        if (isY2) {
#pragma unroll
            for (int j = 0; j < 12; ++j)
                pf[j >> 2][j & 3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsY, (int)(j < 8 ? va : vc), (int)(seg_so2 + (unsigned)(j & 7) * pl4), 0));
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j)
                pf[j >> 2][j & 3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, (int)(j < 8 ? va : vc), (int)(seg_so2 + (unsigned)(j & 7) * pl4), 0));
        }
    };
    auto commit2 = [&](float (&pf)[TPW][4], int ybase, int xpar, int xbase, bool storeX) {
        const int row = __builtin_amdgcn_readfirstlane((isY2 ? ybase : xbase) + odd2);
        const unsigned buf = (unsigned)__builtin_amdgcn_readfirstlane(isY2 ? (ybase + odd2 + 4 * RING) % RING : 2 * xpar + odd2);
        if (row >= h0 && row < h1 && (isY2 || storeX)) {                      // wave-uniform: the row's operand planes
            qu32x2 h16[3];
            unsigned l8[3], h8[3];
#pragma unroll
            for (int qd = 0; qd < 3; ++qd) split_quad(pf[qd], h16[qd], l8[qd], h8[qd]);
            const unsigned prow = (pmap2 + (unsigned)row) * (unsigned)g.W;
            __builtin_amdgcn_raw_buffer_store_b128(qu32x4{h16[0][0], h16[0][1], h16[1][0], h16[1][1]}, rsP16, (int)pv16, (int)(prow * 40u), 0);
            __builtin_amdgcn_raw_buffer_store_b64(h16[2], rsP16, (int)pvC16, (int)(prow * 40u), 0);
            __builtin_amdgcn_raw_buffer_store_b64(qu32x2{l8[0], l8[1]}, rsP8L, (int)pv8, (int)(prow * 24u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(l8[2], rsP8L, (int)pvC8, (int)(prow * 24u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(0u, rsP8L, (int)pvZ8, (int)(prow * 24u), 0);
            __builtin_amdgcn_raw_buffer_store_b64(qu32x2{h8[0], h8[1]}, rsP8H, (int)pv8, (int)(prow * 24u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(h8[2], rsP8H, (int)pvC8, (int)(prow * 24u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(0u, rsP8H, (int)pvZ8, (int)(prow * 24u), 0);
            if (tailw) {                                                      // wave-uniform: tail pixels that hold classes 16..19 -> their pad bytes
                __builtin_amdgcn_raw_buffer_store_b32(0u, rsP8L, (int)pvZT, (int)(prow * 24u), 0);
                __builtin_amdgcn_raw_buffer_store_b32(0u, rsP8H, (int)pvZT, (int)(prow * 24u), 0);
            }
        }
        unsigned hi[12], lo[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            hi[j] = f32_to_bf16_bits(pf[j >> 2][j & 3]);
            lo[j] = f32_to_bf16_bits(pf[j >> 2][j & 3] - bf16_bits_to_f32((unsigned short)hi[j]));
        }
        unsigned char* a = lds + dA + buf * strideM;
        unsigned char* b = lds + dB + buf * sB;
        unsigned char* c = lds + dC + buf * sC;
        *reinterpret_cast<pu32x2*>(a) = pu32x2{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
        if (NP == 2) *reinterpret_cast<pu32x2*>(a + loM) = pu32x2{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
        *reinterpret_cast<pu32x2*>(b) = pu32x2{hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16)};
        if (NP == 2) *reinterpret_cast<pu32x2*>(b + loB) = pu32x2{lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16)};
        if (hasC) {
            *reinterpret_cast<pu32x2*>(c) = pu32x2{hi[8] | (hi[9] << 16), hi[10] | (hi[11] << 16)};
            if (NP == 2) *reinterpret_cast<pu32x2*>(c + loC) = pu32x2{lo[8] | (lo[9] << 16), lo[10] | (lo[11] << 16)};
        }
    };
    auto seg_setup = [&](int n, int col0) { if constexpr (PLW) seg_setup2(n, col0); else seg_setup1(n, col0); };
    auto prefetch = [&](float (&pf)[TPW][4], int ybase, int xbase) { if constexpr (PLW) prefetch2(pf, ybase, xbase); else prefetch1(pf, ybase, xbase); };
    auto commit = [&](float (&pf)[TPW][4], int ybase, int xpar, int xbase, bool storeX) {
        if constexpr (PLW) commit2(pf, ybase, xpar, xbase, storeX); else commit1(pf, ybase, xpar);
    };

#pragma unroll 1
    for (int q0 = qlo; q0 < qhi;) {
        const int strip = q0 / pps, pr = q0 - strip * pps, n = strip / tc, ct = strip - n * tc;
        const int npairs = min(qhi - q0, pps - pr);
        const int col0 = w0 + ct * WT, r0 = h0 + 2 * pr, r1 = min(h1, r0 + 2 * npairs);
        q0 += npairs;
        // ---- warm the ring: Y rows r0 - PAD .. r0 + PAD + 1 and X rows r0, r0 + 1 through the same task machinery, two Y rows at a time
        seg_setup(n, col0);
        __syncthreads();                                  // the previous segment's rows are consumed
        static_assert((PAD + 1) % 2 == 0, "the ring is warmed two rounds at a time");
        for (int w2 = 0; w2 < PAD + 1; w2 += 2) {         // both register sets in flight: half as many exposed round trips
            prefetch(pfA, r0 - PAD + 2 * w2, r0);
            prefetch(pfB, r0 - PAD + 2 * w2 + 2, r0);
            commit(pfA, r0 - PAD + 2 * w2, 0, r0, w2 == 0);
            commit(pfB, r0 - PAD + 2 * w2 + 2, 0, r0, false);
        }
        __syncthreads();
        // One pair of image rows.  Its loads are for the pair AFTER next (global -> registers takes longer than one pair's MFMAs: with
        // a single register set 0.16 of the launch's 0.64 ms was the commit waiting for its loads); it commits the next pair's rows,
        // fetched one pair earlier into the other register set.
        auto do_pair = [&](int r, float (&pfIssue)[TPW][4], float (&pfCommit)[TPW][4]) {
            const int xpar = ((r - r0) >> 1) & 1;
            const bool more = r + 2 < r1;
            if (r + 4 < r1) prefetch(pfIssue, r + 4 + PAD, r + 4);
            const int rx = r + quad;                      // this quad's image row; an odd tail row (rx >= r1) multiplies zeros only if ...
            const unsigned short* Xbm = Xm + (size_t)(2 * xpar + quad) * C::XROWM;
            const unsigned short* Xbr = Xr + (size_t)(2 * xpar + quad) * C::XROWR;
            if (rx < r1) {                                // wave-uniform (per quad)
                // k <-> pixel: the MFMA's k = 8 q + 4 hf + e is pixel 16 hf + 4 q + e of the 32-pixel step (any bijection will do, A
                // and B use the same one): per transpose read (hf fixed) the lanes of one LDS cycle (q = 0, 1 or q = 2, 3; source lane
                // (qq, pq) = pixel 4 q + qq, classes 4 pq ..) then cover 8 CONSECUTIVE pixels x 32 B = every bank exactly once, with or
                // without a displacement -- and the whole lane-dependent part of a fragment address is ONE register.
                const int lane_m = (4 * q + qq) * CM + 4 * pq, lane_r = (4 * q + qq) * CR;
                int ybm[T], ybr[RT], xar[RT];             // per-row bases (bf16 elements): Y ring slot of displacement dy = rx + PAD - dy
#pragma unroll
                for (int dy = 0; dy < T; ++dy) ybm[dy] = ((rx + PAD - dy + 4 * RING) % RING) * C::YROWM + lane_m;
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int dl = min(4 * t + pq, T - 1);      // remainder tiles: the source lane group is the displacement
                    ybr[t] = ((rx + PAD - dl + 4 * RING) % RING) * C::YROWR + lane_r;
                    xar[t] = lane_r + dl * CR;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    auto tr2 = [&](const unsigned short* p0, int step_elems) {          // one fragment = two transpose reads (hf = 0, 1)
#ifdef MISEG_PX_NOLDS
                        { const short v = (short)(size_t)p0; const s16x8 f = {v, v, v, v, (short)step_elems, v, v, v}; return __builtin_bit_cast(pbf16x8_t, f); }
#endif
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(PX_LDS_S16X4(p0));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(PX_LDS_S16X4(p0 + step_elems));
                        const s16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        return __builtin_bit_cast(pbf16x8_t, f);
                    };
                    // A fragments of the tile rows this role uses
                    pbf16x8_t af[NP][MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (!TS::row_used(mt)) continue;
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            af[pl][mt] = mt < T ? tr2(Xbm + pl * xmPlane + lane_m + (ks * 32 + mt) * CM, 16 * CM)                       // dx = mt
                                                : tr2(Xbr + pl * xrPlane + xar[mt >= T ? mt - T : 0] + ks * 32 * CR, 16 * CR);
                    }
#ifndef MISEG_PX_BQ
#define MISEG_PX_BQ 2
#endif
                    constexpr int BQ = MISEG_PX_BQ;        // B fragments BQ - 1 tile columns ahead of the MFMAs
                    pbf16x8_t bq[BQ][NP];
                    auto load_b = [&](int nt, pbf16x8_t* dst) {
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            dst[pl] = nt < T ? tr2(Ym + pl * ymPlane + ybm[nt < T ? nt : 0] + ks * 32 * CM, 16 * CM)
                                             : tr2(Yr + pl * yrPlane + ybr[nt >= T ? nt - T : 0] + ks * 32 * CR, 16 * CR);
                    };
#pragma unroll
                    for (int j = 0; j < BQ - 1; ++j)
                        if (TS::nth_col(j) < NT) load_b(TS::nth_col(j), bq[j]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (!TS::col_used(nt)) continue;
                        const int rank = TS::col_rank(nt), sl = rank % BQ;          // compile-time after unrolling
                        if (TS::nth_col(rank + BQ - 1) < NT) load_b(TS::nth_col(rank + BQ - 1), bq[(rank + BQ - 1) % BQ]);
                        __builtin_amdgcn_sched_barrier(0);    // keep the read-ahead where it is
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (!TS::mine(mt, nt)) continue;
#ifdef MISEG_PX_NOMFMA
                            acc[mt][nt][0] += (float)af[0][mt][0] * (float)bq[sl][0][0] + (float)af[NP - 1][mt][1] * (float)bq[sl][NP - 1][1];
                            continue;
#endif
                            if (NTERMS == 3) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1][mt], bq[sl][0], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bq[sl][NP - 1], acc[mt][nt], 0, 0, 0);
                            }
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bq[sl][0], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
            }
            if (more) commit(pfCommit, r + 2 + PAD, xpar ^ 1, r + 2, true);
            __syncthreads();                              // next pair's rows visible; this pair's X buffers / oldest ring rows reusable
        };
        if (r0 + 2 < r1) prefetch(pfB, r0 + 2 + PAD, r0 + 2);
#pragma unroll 1
        for (int r = r0; r < r1; r += 4) {
            do_pair(r, pfA, pfB);
            if (r + 2 < r1) do_pair(r + 2, pfB, pfA);
        }
    }
    // ---- the two quads' accumulators meet in LDS (fixed order), in the (dx*K + i, dy*K + j) layout joint_reduce_kernel reads
    constexpr int DN = C::DN;
    float* Ds = reinterpret_cast<float*>(lds);
    __syncthreads();
    for (int e = tid; e < DN * DN; e += kPT) Ds[e] = 0.f;
    for (int w = 0; w < 2; ++w) {
        __syncthreads();
        if (quad == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (TS::mine(mt, nt)) {
                        // tile coordinates -> (displacement, class): main tiles hold classes 0..15 of one displacement (row 4q + r, column
                        // l15), remainder tiles classes 16..19 of four displacements (row: displacement 4t + q, class 16 + r)
                        const int dyc = nt < T ? nt : 4 * (nt - T) + (l15 >> 2), jc = nt < T ? l15 : 16 + (l15 & 3);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int dxr = mt < T ? mt : 4 * (mt - T) + q, ir = mt < T ? 4 * q + r : 16 + r;
                            if (dxr < T && dyc < T) {
                                const int idx = (dxr * K + ir) * DN + dyc * K + jc;
                                Ds[idx] += acc[mt][nt][r];
                            }
                        }
                    }
        }
    }
    __syncthreads();
    float* out = partials + ((size_t)slot * g.G + blockIdx.x) * (DN * DN);
    for (int e = tid; e < DN * DN; e += kPT) out[e] = Ds[e];
}

template <int PAD, int NTERMS, bool PLW>
__global__ __launch_bounds__(kPT, 1) void joint_fwd_px_kernel(const float* __restrict__ x, const float* __restrict__ y, PxGeom g,
                                                              const int32_t* __restrict__ win, float* __restrict__ partials, MiPlanes pln) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch ((wave & 4) ? 3 - (wave & 3) : (wave & 3)) {
        case 0: joint_px_body<PAD, NTERMS, 0, PLW>(x, y, g, win, partials, ldsb, pln); break;
        case 1: joint_px_body<PAD, NTERMS, 1, PLW>(x, y, g, win, partials, ldsb, pln); break;
        case 2: joint_px_body<PAD, NTERMS, 2, PLW>(x, y, g, win, partials, ldsb, pln); break;
        default: joint_px_body<PAD, NTERMS, 3, PLW>(x, y, g, win, partials, ldsb, pln); break;
    }
}

template <int PAD>
static size_t px_lds(int nterms) {
    typedef PX<PAD> C;
    const int np = nterms == 1 ? 1 : 2;
    const size_t rows = ((size_t)np * C::RING * (C::YROWM + C::YROWR) + (size_t)np * 4 * (C::XROWM + C::XROWR)) * 2, dred = (size_t)C::DN * C::DN * 4;
    return rows > dred ? rows : dred;
}

bool joint_fwd_bf16_supported(const JointGeom& g) {
    if (g.sb != 1 || g.K != 20 || (g.pad != 3 && g.pad != 1)) return false;
    return (size_t)g.N * g.K * g.H * g.W * 4 < 0x40000000ull;                  // 32-bit buffer offsets with an out-of-range marker
}

int launch_joint_fwd_px(hipStream_t st, const float* x, const float* y, const JointGeom& jg, const int32_t* win, float* partials, int nterms,
                        unsigned char* planes) {
    if (nterms == 2) nterms = 3;     // the f16 + fp8 split exists for the backward only (mi_local_bwd_f8.hip)
    PxGeom g{jg.N, jg.H, jg.W, jg.P, jg.S, jg.G, jg.hs};
    dim3 grid(g.G, g.P * g.S), block(kPT);
    const MiPlanes pln = mi_planes(planes, (int64_t)g.S * 2 * g.N, (int64_t)g.H * g.W);     // planes: the heads layout (x = probs, y = probs + N K H W)
#define PXL(PADV, NT_, PLW_)                                                                                                           \
    {                                                                                                                                  \
        const size_t lb = px_lds<PADV>(nterms);                                                                                         \
        hipFuncSetAttribute((const void*)joint_fwd_px_kernel<PADV, NT_, PLW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);    \
        hipLaunchKernelGGL((joint_fwd_px_kernel<PADV, NT_, PLW_>), grid, block, lb, st, x, y, g, win, partials, pln);                   \
    }
    if (planes) {
        if (jg.pad != 3 || nterms != 3) return -1;
        PXL(3, 3, true)
    } else if (jg.pad == 3) { if (nterms == 1) PXL(3, 1, false) else PXL(3, 3, false) }
    else { if (nterms == 1) PXL(1, 1, false) else PXL(1, 3, false) }
#undef PXL
    return 0;
}

}  // namespace miseg
