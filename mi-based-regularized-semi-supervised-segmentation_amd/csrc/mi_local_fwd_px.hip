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
    int N, H, W, P, S, G, L;      // L: image rows per segment (even)
    long long hs;
};

template <int PAD, int NTERMS, int ROLE>
__device__ __forceinline__ void joint_px_body(const float* __restrict__ x, const float* __restrict__ y, const PxGeom& g,
                                              const int32_t* __restrict__ win, float* __restrict__ partials, unsigned char* lds) {
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
    const int tc = (w1 - w0 + WT - 1) / WT, ns = (h1 - h0 + g.L - 1) / g.L;
    const int nSeg = g.N * tc * ns;
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
    // (lane = (row, c4, pixel 64 ..)).  Tasks are dealt round-robin to the 8 waves: 20 + 2 tasks -> 3 per wave at most.
    constexpr int NTASK = 22, TPW = (NTASK + 7) / 8;
    float pf[TPW][4];
    struct RowSet { int n, col0, ya, yb, xa, xb; };      // image rows of the four row loads (ya, yb: Y; xa, xb: X), may be outside the window
    auto task_addr = [&](const RowSet& rs, int t, unsigned& vo, unsigned& so) {
        // the per-lane byte offset (OOB if the lane has nothing to load) and the scalar offset of channel c4*4; t < 10 reads y, else x
        int kind, c4, row, col;
        if (t < 20) { kind = t / 5; c4 = t % 5; }
        else { kind = 4; c4 = 0; }
        bool ok;
        if (kind < 2) {
            row = kind == 0 ? rs.ya : rs.yb; col = rs.col0 + lane;
            ok = row >= h0 && row < h1 && col < w1;
        } else if (kind < 4) {
            row = kind == 2 ? rs.xa : rs.xb; col = rs.col0 - PAD + lane;
            ok = row >= h0 && row < h1 && col >= w0 && col < w1;
        } else {   // tails: task 20 -> row xa, task 21 -> row xb; lane = c4 * (2 PAD) + pixel
            row = t == 20 ? rs.xa : rs.xb; c4 = lane / (2 * PAD); col = rs.col0 - PAD + 64 + lane % (2 * PAD);
            ok = lane < 5 * 2 * PAD && row >= h0 && row < h1 && col >= w0 && col < w1;
        }
        vo = ok ? (unsigned)row * wb + (unsigned)col * 4u + (kind == 4 ? (unsigned)(c4 * 4) * pl4 : 0u) : OOB;
        so = (unsigned)rs.n * (unsigned)K * pl4 + (kind == 4 ? 0u : (unsigned)(c4 * 4) * pl4);
    };
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    // two descriptors, chosen by a wave-uniform branch: a descriptor picked through a per-task pointer made the compiler wrap every
    // load in a waterfall loop (readfirstlane / saveexec / branch per load)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)tbytes, 0x00020000);
    auto prefetch = [&](const RowSet& rs) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wv + 8 * i;                     // wave-uniform
            unsigned vo = OOB, so = 0;
            if (t < NTASK) task_addr(rs, t, vo, so);
            so = (unsigned)__builtin_amdgcn_readfirstlane((int)so);      // wave-uniform by construction: say so (else: a waterfall loop per load)
            if (t < 10) {
#pragma unroll
                for (int c = 0; c < 4; ++c) pf[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsY, (int)vo, (int)(so + (unsigned)c * pl4), 0));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) pf[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, (int)vo, (int)(so + (unsigned)c * pl4), 0));
            }
        }
    };
    auto commit = [&](const RowSet& rs, int xpar) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wv + 8 * i;
            if (t >= NTASK) continue;                     // wave-uniform
            unsigned short* dst;
            size_t lo_plane;                              // distance to the lo plane of the array written
            if (t < 20) {
                const int c4 = t % 5, kind = t / 5;       // kinds 0 / 1: the Y rows, 2 / 3: the X rows
                if (kind < 2) {
                    const int sl = ((kind == 0 ? rs.ya : rs.yb) + 4 * RING) % RING;
                    dst = c4 < 4 ? Ym + (size_t)sl * C::YROWM + (size_t)lane * CM + c4 * 4 : Yr + (size_t)sl * C::YROWR + (size_t)lane * CR;
                    lo_plane = c4 < 4 ? ymPlane : yrPlane;
                } else {
                    const int xb = 2 * xpar + (kind - 2);
                    dst = c4 < 4 ? Xm + (size_t)xb * C::XROWM + (size_t)lane * CM + c4 * 4 : Xr + (size_t)xb * C::XROWR + (size_t)lane * CR;
                    lo_plane = c4 < 4 ? xmPlane : xrPlane;
                }
            } else {
                const int c4 = lane / (2 * PAD), px = 64 + lane % (2 * PAD), xb = 2 * xpar + (t - 20);
                dst = c4 < 4 ? Xm + (size_t)xb * C::XROWM + (size_t)px * CM + c4 * 4 : Xr + (size_t)xb * C::XROWR + (size_t)px * CR;
                lo_plane = c4 < 4 ? xmPlane : xrPlane;
                if (lane >= 5 * 2 * PAD) continue;        // divergent: only the tail lanes store
            }
            unsigned hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hi[c] = f32_to_bf16_bits(pf[i][c]);
                lo[c] = f32_to_bf16_bits(pf[i][c] - bf16_bits_to_f32((unsigned short)hi[c]));
            }
            *reinterpret_cast<pu32x2*>(dst) = pu32x2{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
            if (NP == 2) *reinterpret_cast<pu32x2*>(dst + lo_plane) = pu32x2{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
        }
    };

#pragma unroll 1
    for (int seg = blockIdx.x; seg < nSeg; seg += g.G) {
        const int sg = seg % ns, ct = (seg / ns) % tc, n = seg / (ns * tc);
        const int col0 = w0 + ct * WT, r0 = h0 + sg * g.L, r1 = min(h1, r0 + g.L);
        // ---- warm the ring: Y rows r0 - PAD .. r0 + PAD + 1 and X rows r0, r0 + 1 through the same task machinery, two Y rows at a time
        __syncthreads();                                  // the previous segment's rows are consumed
        for (int w2 = 0; w2 < PAD + 1; ++w2) {
            RowSet rs{n, col0, r0 - PAD + 2 * w2, r0 - PAD + 2 * w2 + 1, r0, r0 + 1};
            prefetch(rs);
            commit(rs, 0);
        }
        __syncthreads();
#pragma unroll 1
        for (int r = r0; r < r1; r += 2) {
            const int xpar = ((r - r0) >> 1) & 1;
            const bool more = r + 2 < r1;
            // rows of the NEXT pair: its two new Y rows and its X rows (zeros outside the window / segment tail)
            RowSet nx{n, col0, r + 2 + PAD, r + 3 + PAD, r + 2, r + 3};
            if (more) prefetch(nx);
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
                    constexpr int BQ = 2;                  // B fragments one tile column ahead of the MFMAs
                    pbf16x8_t bq[BQ][NP];
                    auto load_b = [&](int nt, pbf16x8_t* dst) {
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            dst[pl] = nt < T ? tr2(Ym + pl * ymPlane + ybm[nt < T ? nt : 0] + ks * 32 * CM, 16 * CM)
                                             : tr2(Yr + pl * yrPlane + ybr[nt >= T ? nt - T : 0] + ks * 32 * CR, 16 * CR);
                    };
                    if (TS::nth_col(0) < NT) load_b(TS::nth_col(0), bq[0]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (!TS::col_used(nt)) continue;
                        const int rank = TS::col_rank(nt), sl = rank % BQ;          // compile-time after unrolling
                        if (TS::nth_col(rank + 1) < NT) load_b(TS::nth_col(rank + 1), bq[(rank + 1) % BQ]);
                        __builtin_amdgcn_sched_barrier(0);    // keep the read-ahead where it is
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (!TS::mine(mt, nt)) continue;
                            if (NTERMS == 3) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1][mt], bq[sl][0], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bq[sl][NP - 1], acc[mt][nt], 0, 0, 0);
                            }
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mt], bq[sl][0], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
            }
            if (more) commit(nx, xpar ^ 1);
            __syncthreads();                              // next pair's rows visible; this pair's X buffers / oldest ring rows reusable
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

template <int PAD, int NTERMS>
__global__ __launch_bounds__(kPT, 1) void joint_fwd_px_kernel(const float* __restrict__ x, const float* __restrict__ y, PxGeom g,
                                                              const int32_t* __restrict__ win, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch ((wave & 4) ? 3 - (wave & 3) : (wave & 3)) {
        case 0: joint_px_body<PAD, NTERMS, 0>(x, y, g, win, partials, ldsb); break;
        case 1: joint_px_body<PAD, NTERMS, 1>(x, y, g, win, partials, ldsb); break;
        case 2: joint_px_body<PAD, NTERMS, 2>(x, y, g, win, partials, ldsb); break;
        default: joint_px_body<PAD, NTERMS, 3>(x, y, g, win, partials, ldsb); break;
    }
}

template <int PAD>
static size_t px_lds(int nterms) {
    typedef PX<PAD> C;
    const int np = nterms == 1 ? 1 : 2;
    const size_t rows = ((size_t)np * C::RING * (C::YROWM + C::YROWR) + (size_t)np * 4 * (C::XROWM + C::XROWR)) * 2, dred = (size_t)C::DN * C::DN * 4;
    return rows > dred ? rows : dred;
}

int launch_joint_fwd_px(hipStream_t st, const float* x, const float* y, const JointGeom& jg, const int32_t* win, float* partials, int nterms) {
    PxGeom g{jg.N, jg.H, jg.W, jg.P, jg.S, jg.G, 32, jg.hs};
    { const char* e = getenv("MISEG_FWD_PX_L"); if (e && atoi(e) > 0) g.L = atoi(e) & ~1; }
    dim3 grid(g.G, g.P * g.S), block(kPT);
#define PXL(PADV, NT_)                                                                                                           \
    {                                                                                                                            \
        const size_t lb = px_lds<PADV>(nterms);                                                                                   \
        hipFuncSetAttribute((const void*)joint_fwd_px_kernel<PADV, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);    \
        hipLaunchKernelGGL((joint_fwd_px_kernel<PADV, NT_>), grid, block, lb, st, x, y, g, win, partials);                        \
    }
    if (jg.pad == 3) { if (nterms == 1) PXL(3, 1) else PXL(3, 3) }
    else { if (nterms == 1) PXL(1, 1) else PXL(1, 3) }
#undef PXL
    return 0;
}

}  // namespace miseg
