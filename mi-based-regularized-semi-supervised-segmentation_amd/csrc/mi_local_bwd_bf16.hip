// Backward of the local-MI joint on the bf16 matrix cores, hi/lo operand split (fp32-class accuracy).
//
// Same transposed GEMM as local_bwd2_kernel (mi_local.hip), per output row h and direction:
//   dA[(b,o), w] = sum_{kred=(a,c)} Gm[(b,o)][(a,c)] * S[c][h + s(a-p)][w]      M = T*K (140 -> 144), N = 64 px, Kred = T*K (140 -> 160)
//   out[o][h][w'] = sum_b dA[(b,o), w' -+ (b-p)]                                 "col2im"
// Block = 512 threads = 8 waves = 2 waves per SIMD, so one wave's LDS / col2im / global phases overlap its SIMD partner's
// MFMAs.  Wave w: output row (w & 3) of the item's 4 rows, M half (w >> 2) -- the two halves of a row sit on the same SIMD.
//   * A (the gradient matrix Gm) is packed once per call into bf16 hi/lo k-step slices [dir][ks][plane][144][32]; every wave
//     loads its own A fragments straight from L2/L1 into registers (1 KiB coalesced buffer loads, one k-step ahead of the
//     MFMAs, the first k-step of the next item during the last of this one).  Nothing in LDS changes during the k-loop, so the
//     loop has NO workgroup barrier and the two waves of a SIMD drift apart: one's LDS / col2im phases hide under the other's
//     MFMAs (an earlier version staged A through a double-buffered LDS slice with a barrier per k-step: all eight waves then
//     moved in lockstep, MFMA and LDS phases serialised, 22 k cycles per item for 8.6 k cycles of MFMA);
//   * B: the source tile stays planar [c][row][w] in LDS; ds_read_b64_tr_b16 transposes 4 planes x 16 pixels on the fly into
//     the 8-consecutive-k lane layout.  Rows live in a 12-row ring (3 groups of 4): a block walks a segment of consecutive
//     4-row items down one 58-column strip and fetches only the 4 new rows per item (registers, one item ahead);
//     The first two k-steps of A (37 KB) are kept in LDS for a whole segment -- every item of a segment multiplies by the same
//     G -- and only the other k-steps are loaded per wave from L2; the half-1 waves park their partial sums in their own col2im
//     staging tile (read by the row's half-0 wave between the next item's two barriers), which is what frees that LDS;
//   * col2im is a gather: a wave parks one M-tile of D at a time in LDS as [col][16 rows] and every lane (= one output
//     column) reads 4 rows per ds_read_b128 at its shifted column; the two M halves are summed through LDS.
// K and PAD are template parameters (K=20; PAD=3 and 1: the shipped taps); other shapes use the fp32 kernels.
#include "mi_local.h"

namespace miseg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_S16X4(ptr) ((__attribute__((address_space(3))) s16x4*)(ptr))

// Profiling builds only (-DMISEG_BWD_ABLATE=bits: 1 no src tile, 2 no MFMA, 4 no col2im, 8 no output write, 16 no A loads).  Compile-time
// so that the shipped kernel's item body is straight-line code: any branch inside it makes the compiler fall back to vmcnt(0)
// at the next wait, which drains the A-fragment prefetches the loop keeps in flight.
#ifndef MISEG_BWD_ABLATE
#define MISEG_BWD_ABLATE 0
#endif
constexpr int kAbl = MISEG_BWD_ABLATE;

struct Bwd3Geom {
    int N, H, W, P, G, accumulate;
    int L;        // items (4-row tiles) per segment
    int S;        // sub-heads in this launch: operands / outputs of head s at + s * hs elements, G and scale rows s * P + p
    long long hs;
};

template <int K, int PAD>
struct B3 {
    static constexpr int T = 2 * PAD + 1, MD = T * K, MT = (MD + 15) / 16, MP = MT * 16, KRED = T * K, KS = (KRED + 31) / 32;
    static constexpr int WT = 64, WB = WT - 2 * PAD, SW = 72, RING = 12;
    static constexpr int MH = (MT + 1) / 2;   // M tiles of half 0; half 1 owns MT - MH
    static constexpr int DL = 20;             // floats per column of the col2im staging tile (16 rows + 4 pad; >= K)
    static_assert(PAD <= 4 && K % 4 == 0 && K <= DL && (K * 4) % 8 == 0, "B3: unsupported shape");
};

__device__ __forceinline__ unsigned short bf16_hi(float v) { return f32_to_bf16_bits(v); }

// workgroup barrier that publishes LDS writes but leaves global loads in flight (__syncthreads also drains vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gpack[p][dir][ks][pl][m][32]: value(m=(b,o), kred=(a,c)) = dir ? G[a,b,c,o] : G[a,b,o,c]; 16-byte slot s of row m stored at
// s ^ (-(m>>2) & 3): with ds_read_b128's lane groups {0-3,12-15,20-27},{4-11,16-19,28-31},.. the 16 lanes of a group then hit 64 distinct banks
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
        const size_t base = ((((size_t)(p * 2 + dir) * C::KS + ks) * NP) * C::MP + m) * 32 + 8 * ((kk >> 3) ^ ((0 - (m >> 2)) & 3)) + (kk & 7);
        gpack[base] = hi;
        if (NP == 2) gpack[base + (size_t)C::MP * 32] = bf16_hi(v - bf16_bits_to_f32(hi));
    }
}

template <int K, int PAD, int NTERMS, bool ACC>
__global__ __launch_bounds__(512, 1) void local_bwd_bf16_kernel(const float* __restrict__ x, const float* __restrict__ y, Bwd3Geom g,
                                                                const int32_t* __restrict__ win,
                                                                const unsigned short* __restrict__ gpack,
                                                                const float* __restrict__ scale, float* __restrict__ gx,
                                                                float* __restrict__ gy) {
    typedef B3<K, PAD> C;
    constexpr int NP = NTERMS == 1 ? 1 : 2, NT = 4;
    constexpr int MT = C::MT, MP = C::MP, MH = C::MH, KS = C::KS, WT = C::WT, SW = C::SW, RING = C::RING, DL = C::DL;
    constexpr int SLICE = NP * MP * 32;                  // bf16 elements per k-step slice
    constexpr int SPLANE = K * RING * SW;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned short* Ss = reinterpret_cast<unsigned short*>(ldsb);                // [NP][K][RING][SW]
    float* Dst = reinterpret_cast<float*>(Ss + NP * SPLANE);                     // [8 waves][64 cols][DL]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r4 = wvu & 3, half = wvu >> 2;
    float* Dw = Dst + (size_t)wvu * WT * DL;
    // the first CK k-steps of A stay in LDS for a whole segment (same G for all its items): filled once per segment, read by
    // every wave at LDS bandwidth; only the remaining k-steps travel L2 -> registers per wave
    constexpr int CK = KS < 2 ? KS : 2;
    unsigned short* Acache = reinterpret_cast<unsigned short*>(Dst + (size_t)8 * WT * DL);   // [CK][SLICE]
    const size_t plane = (size_t)g.H * g.W;

    // ---- segments: (dir, window p, sample n, column strip ct, run of <= L consecutive 4-row items)
    int64_t per_head = 0;
    for (int p = 0; p < g.P; ++p) {
        const int tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4, tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WB - 1) / C::WB;
        per_head += (int64_t)g.N * tc * ((tr + g.L - 1) / g.L);
    }
    const int64_t total = per_head * g.S;
    struct Seg { int dir, s, p, n, col0, h0, h1, w0, w1, rt0, rt1; };
    auto decode = [&](int64_t sid, Seg& o) {
        o.dir = sid >= total;
        int64_t rem = sid - (o.dir ? total : 0);
        o.s = (int)(rem / per_head);
        rem -= (int64_t)o.s * per_head;
        int p = 0, tr = 0, tc = 0, ns = 0;
        for (; p < g.P; ++p) {
            tr = (win[p * 4 + 1] - win[p * 4 + 0] + 3) / 4;
            tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WB - 1) / C::WB;
            ns = (tr + g.L - 1) / g.L;
            const int64_t cnt = (int64_t)g.N * tc * ns;
            if (rem < cnt) break;
            rem -= cnt;
        }
        o.p = p;
        o.h0 = win[p * 4 + 0]; o.h1 = win[p * 4 + 1]; o.w0 = win[p * 4 + 2]; o.w1 = win[p * 4 + 3];
        const int sg = rem % ns, ct = (rem / ns) % tc;
        o.n = rem / ((int64_t)ns * tc);
        o.col0 = o.w0 + ct * C::WB;
        o.rt0 = sg * g.L;
        o.rt1 = min(tr, o.rt0 + g.L);
    };

    // ---- source rows: group j = rows h0+4j .. h0+4j+3 of all K planes; wave w owns (ch,row) pairs pr = w + 8*i; ring slot (j+1) % 3
    constexpr int PFN = (K * 4) / 8;
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    constexpr unsigned OOB = 0xC0000000u;
    auto fetch_group = [&](const Seg& o, int j, float* dst, bool live = true) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((o.dir ? x : y) + (size_t)o.s * g.hs), 0, (int)tbytes, 0x00020000);
        const int col = o.col0 - PAD + lane;
        const unsigned vo = (col >= o.w0 && col < o.w1) ? (unsigned)col * 4u : OOB;
#pragma unroll
        for (int i = 0; i < PFN; ++i) {
            const int pr = wvu + 8 * i, ch = pr >> 2, r = pr & 3, row = o.h0 + 4 * j + r;
            const bool ok = live && row >= o.h0 && row < o.h1;
            // 32-bit arithmetic (tensor < 1 GiB, checked on the host), computed unconditionally so that `ok` is a scalar select, not a branch
            const unsigned lin = ((unsigned)(o.n * K + ch) * (unsigned)plane + (unsigned)row * (unsigned)g.W) * 4u;
            const unsigned so = ok ? lin : OOB;
            dst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(so + vo), 0, 0));
        }
    };
    auto commit_group = [&](int slot, const float* src) {
#pragma unroll
        for (int i = 0; i < PFN; ++i) {
            const int pr = wvu + 8 * i, ch = pr >> 2, r = pr & 3;
            unsigned short* d = Ss + (ch * RING + slot * 4 + r) * SW + lane;
            const unsigned short hi = bf16_hi(src[i]);
            d[0] = hi;
            if (NP == 2) d[SPLANE] = bf16_hi(src[i] - bf16_bits_to_f32(hi));
        }
    };
    auto slices_of = [&](const Seg& o) { return gpack + (size_t)((o.s * g.P + o.p) * 2 + o.dir) * KS * SLICE; };

    // pending output of the previous item (kept in registers until the partner half's partial is visible)
    float outp[K];
    int prow = 0, pcol0 = 0, pn = 0, pdir = 0, ph1 = 0, pw1 = 0, pp = 0, ps = 0;
    bool pending = false;
    // Branch-free: a lane without an output (tile halo, past the window, nothing pending yet) stores to an out-of-range buffer
    // offset, which the hardware drops.
    auto finish_output = [&]() {
        const float* Dpartner = Dst + (size_t)(wvu + 4) * WT * DL;   // the row's half-1 wave parked its partial sums in its own staging tile
#pragma unroll
        for (int o4 = 0; o4 < K; o4 += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Dpartner + lane * DL + o4);
            outp[o4] += v[0]; outp[o4 + 1] += v[1]; outp[o4 + 2] += v[2]; outp[o4 + 3] += v[3];
        }
        const int col = pcol0 + lane - PAD;
        const bool live = pending && prow < ph1 && lane >= PAD && lane < PAD + C::WB && col < pw1 && !(kAbl & 8);
        const float sc = scale[ps * g.P + pp];
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)((pdir ? gy : gx) + (size_t)ps * g.hs), 0, (int)tbytes, 0x00020000);
        const unsigned vo = live ? (unsigned)(((size_t)pn * K * plane + (size_t)prow * g.W + col) * 4) : OOB;
        const unsigned pl4 = (unsigned)(plane * 4);
        if (ACC) {
            float old[K];
#pragma unroll
            for (int o = 0; o < K; ++o) old[o] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rso, (int)vo, (int)(o * pl4), 0));
#pragma unroll
            for (int o = 0; o < K; ++o) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old[o] + sc * outp[o]), rso, (int)vo, (int)(o * pl4), 0);
        } else {
#pragma unroll
            for (int o = 0; o < K; ++o) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc * outp[o]), rso, (int)vo, (int)(o * pl4), 0);
        }
    };

    const int64_t nseg = 2 * total;
    Seg sg, sgn_;
    // A fragments of the uncached k-steps: "step" = (k-step - CK, M tile) in execution order; AHEAD steps in flight in a ring of
    // AHEAD + 1 register slots, issued from the start of the item (the cached k-steps run first and cover the latency).
    constexpr int AHEAD = 3;
    const int aoff0 = ((l15 * 32) + 8 * (q ^ ((0 - (l15 >> 2)) & 3))) * 2;      // byte offset of this lane's 16 B inside an M tile
    auto rsrc_of = [&](const unsigned short* slices) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)slices, 0, KS * SLICE * 2, 0x00020000);
    };
    auto load_step = [&](__amdgpu_buffer_rsrc_t rs, int mt0, int mtn, int step, u32x4* dst) {
        const int ks = CK + step / mtn, ml = step - (step / mtn) * mtn;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            dst[pl] = (kAbl & 16) ? u32x4{0u, 0u, 0u, 0u}
                                      : __builtin_amdgcn_raw_buffer_load_b128(rs, aoff0, (ks * SLICE + (pl * MP + (mt0 + ml) * 16) * 32) * 2, 0);
    };
    if ((int64_t)blockIdx.x < nseg) decode(blockIdx.x, sg);
#pragma unroll 1
    for (int64_t sid = blockIdx.x; sid < nseg; sid += g.G) {
        const bool more_seg = sid + g.G < nseg;
        if (more_seg) decode(sid + g.G, sgn_);
        const int dir = sg.dir, sgn = dir ? 1 : -1;
        const unsigned short* gbase = slices_of(sg);
        lds_barrier();                           // previous item: every wave is done with the ring, partials are parked
        if (!(kAbl & 1)) {                   // warm the ring: groups rt0-1, rt0, rt0+1
            float w3[3][PFN];
#pragma unroll
            for (int u = 0; u < 3; ++u) fetch_group(sg, sg.rt0 - 1 + u, w3[u]);
#pragma unroll
            for (int u = 0; u < 3; ++u) commit_group((sg.rt0 + u) % 3, w3[u]);
        }
        for (int idx = tid; idx < CK * SLICE / 8; idx += 512)     // this segment's first CK slices (published by the item's second barrier)
            *reinterpret_cast<u32x4*>(Acache + (size_t)idx * 8) = *reinterpret_cast<const u32x4*>(gbase + (size_t)idx * 8);
        int base3 = sg.rt0 % 3;                  // ring slot of group rt-1
        float pf[PFN];
#pragma unroll 1
        for (int rt = sg.rt0; rt < sg.rt1; ++rt) {
            if (rt > sg.rt0) {
                lds_barrier();
                if (!(kAbl & 1)) commit_group((base3 + 2) % 3, pf);       // group rt+1 replaces group rt-2
            }
            // Retire the previous item here (half-0 waves: partner partials + global stores): every wave is between the two
            // barriers anyway, the stores get a head start of a whole B-fragment read burst on the first A-fragment wait (vmcnt
            // retires in order), and the 20 output registers are dead during the k-loop.
            if (half == 0) finish_output();
            lds_barrier();                       // ring commits visible
            // From here to the next item's barrier the waves run free (no workgroup barrier inside the k-loop).
            const bool more_rt = rt + 1 < sg.rt1;
            // The rows of item rt+1 are fetched at the start of the LAST k-step: vmcnt retires in order, so loads issued before the
            // k-loop would sit in front of every A-fragment wait; issued here only the next item's first fragments queue behind them.
            auto late_work = [&]() {
                if (!(kAbl & 1)) fetch_group(sg, rt + 2, pf, more_rt);
            };
            auto body = [&](auto HC) {
                constexpr int HALF = decltype(HC)::value;
                constexpr int MT0 = HALF ? MH : 0, MTN = HALF ? MT - MH : MH;
                f32x4 acc[MTN][NT];
#pragma unroll
                for (int ml = 0; ml < MTN; ++ml)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[ml][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const __amdgpu_buffer_rsrc_t rsA = rsrc_of(gbase);
                constexpr int NS = (KS - CK) * MTN;       // uncached steps
                u32x4 ring[AHEAD + 1][NP];
#pragma unroll
                for (int i = 0; i < AHEAD; ++i)
                    if (i < NS) load_step(rsA, MT0, MTN, i, ring[i]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks == KS - 1) late_work();
                    // B fragments: kred block of 8 = two 4-blocks kb, kb+4; 4-block -> (a, c0..c0+3) never straddles a (K % 4 == 0)
                    bf16x8_t bfr[NP][NT];
                    {
                        const int qq = l15 >> 2, pq = l15 & 3;
                        int soff[2];
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const int kb = min(ks * 32 + 8 * q + 4 * hf, C::KRED - 4);
                            const int a = kb / K, c0 = kb - a * K;
                            const int rr = r4 + sgn * (a - PAD);                 // source row relative to the item's first row: [-PAD, 3+PAD]
                            int slot = base3 + 1 + ((rr + 4) >> 2) - 1;          // group rt + floor(rr/4) -> slot (group + 1) % 3
                            slot = slot >= 3 ? slot - 3 : slot;
                            soff[hf] = ((c0 + qq) * RING + slot * 4 + (rr & 3)) * SW + 4 * pq;
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
                    if (!(kAbl & 2))
#pragma unroll
                        for (int ml = 0; ml < MTN; ++ml) {
                            bf16x8_t af[NP];
                            if (ks < CK) {                                     // compile-time after unrolling
                                __builtin_amdgcn_sched_barrier(0);             // one tile's fragments at a time
                                const int m = (MT0 + ml) * 16 + l15;
#pragma unroll
                                for (int pl = 0; pl < NP; ++pl)
                                    af[pl] = *reinterpret_cast<const bf16x8_t*>(Acache + (size_t)ks * SLICE + ((size_t)pl * MP + m) * 32 + 8 * (q ^ ((0 - (m >> 2)) & 3)));
                            } else {
                                const int step = (ks - CK) * MTN + ml;
                                if (step + AHEAD < NS) load_step(rsA, MT0, MTN, step + AHEAD, ring[(step + AHEAD) % (AHEAD + 1)]);
                                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch where it is: the scheduler would sink it next to its use
#pragma unroll
                                for (int pl = 0; pl < NP; ++pl) af[pl] = __builtin_bit_cast(bf16x8_t, ring[step % (AHEAD + 1)][pl]);
                            }
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if (NTERMS == 3) {
                                    acc[ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1], bfr[0][nt], acc[ml][nt], 0, 0, 0);
                                    acc[ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[NP - 1][nt], acc[ml][nt], 0, 0, 0);
                                }
                                acc[ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[0][nt], acc[ml][nt], 0, 0, 0);
                            }
                        }
                }
                // ---- col2im gather: lane = output tile column wc; out[o] += D[(b,o)][wc + sgn*(b-PAD)]
#pragma unroll
                for (int o = 0; o < K; ++o) outp[o] = 0.f;
                if (kAbl & 4) {
#pragma unroll
                    for (int ml = 0; ml < MTN; ++ml)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) outp[(ml * NT + nt) % K] += acc[ml][nt][0] + acc[ml][nt][1] + acc[ml][nt][2] + acc[ml][nt][3];
                } else {
#pragma unroll
                    for (int ml = 0; ml < MTN; ++ml) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(Dw + (nt * 16 + l15) * DL + 4 * q) = acc[ml][nt];
                        // same-wave LDS traffic is serviced in issue order: a compiler barrier is all the staging needs
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            constexpr int dummy = 0; (void)dummy;
                            const int m = (MT0 + ml) * 16 + 4 * rq;           // compile-time after unrolling
                            if (m < C::MD) {
                                const int b = m / K, o = m % K;
                                const int src_col = lane + sgn * (b - PAD);
                                const bool in = src_col >= 0 && src_col < WT;
                                const f32x4 v = *reinterpret_cast<const f32x4*>(Dw + (in ? src_col : lane) * DL + 4 * rq);
                                outp[o] += in ? v[0] : 0.f; outp[o + 1] += in ? v[1] : 0.f;
                                outp[o + 2] += in ? v[2] : 0.f; outp[o + 3] += in ? v[3] : 0.f;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("" ::: "memory");
                    }
                }
                if (HALF == 1) {                 // park the partial sums for the row's half-0 wave
                    float* pk = Dw;              // read by the half-0 wave between the next item's two barriers, before this tile is staged into again
#pragma unroll
                    for (int o4 = 0; o4 < K; o4 += 4)
                        *reinterpret_cast<f32x4*>(pk + lane * DL + o4) = f32x4{outp[o4], outp[o4 + 1], outp[o4 + 2], outp[o4 + 3]};
                }
            };
            if (half == 0) body(std::integral_constant<int, 0>{});
            else body(std::integral_constant<int, 1>{});
            prow = sg.h0 + 4 * rt + r4; pcol0 = sg.col0; pn = sg.n; pdir = dir; ph1 = sg.h1; pw1 = sg.w1; pp = sg.p; ps = sg.s;
            pending = true;
            base3 = base3 == 2 ? 0 : base3 + 1;
        }
        sg = sgn_;
    }
    __syncthreads();
    if (half == 0) finish_output();
}

template <int K, int PAD>
static size_t bwd3_lds(int nterms) {
    typedef B3<K, PAD> C;
    const int np = nterms == 1 ? 1 : 2;
    const int ck = C::KS < 2 ? C::KS : 2;
    return (size_t)np * K * C::RING * C::SW * 2 + (size_t)8 * C::WT * C::DL * 4 + (size_t)ck * np * C::MP * 32 * 2;
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
    const int total = g.P * g.S * 2 * C::KS * C::MP * 32;
    if (nterms == 1) hipLaunchKernelGGL((pack_g_bf16_kernel<K, PAD, 1>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, g.P * g.S, gpack);
    else hipLaunchKernelGGL((pack_g_bf16_kernel<K, PAD, 2>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, g.P * g.S, gpack);
    const size_t lds = bwd3_lds<K, PAD>(nterms);
    auto go = [&](auto kernel) {
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(g.G), dim3(512), lds, st, x, y, g, win, gpack, scale, gx, gy);
    };
    if (nterms == 1) g.accumulate ? go(local_bwd_bf16_kernel<K, PAD, 1, true>) : go(local_bwd_bf16_kernel<K, PAD, 1, false>);
    else g.accumulate ? go(local_bwd_bf16_kernel<K, PAD, 3, true>) : go(local_bwd_bf16_kernel<K, PAD, 3, false>);
    return 0;
}

int launch_local_bwd_bf16(hipStream_t st, const float* x, const float* y, int64_t S, int64_t hs, int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad,
                          const int32_t* win, int64_t P, const float* grad_raw, const float* scale, float* gx, float* gy, int accumulate,
                          void* ws, int nterms) {
    // MISEG_BWD_KERNEL=tiles: the round-1 stacked-(b,o) + col2im kernel of this file; default: the row-streaming kernel
    // (mi_local_bwd_rows.hip), same operands, same workspace size
    static const bool tiles = [] { const char* e = getenv("MISEG_BWD_KERNEL"); return e && !strcmp(e, "tiles"); }();
    if (!tiles) return launch_local_bwd_rows(st, x, y, S, hs, N, K, H, W, pad, win, P, grad_raw, scale, gx, gy, accumulate, ws, nterms);
    Bwd3Geom g{(int)N, (int)H, (int)W, (int)P, 256, accumulate, 8, (int)S, (long long)hs};
    if (P == 1) {   // one whole-image window (the shipped configuration): pick the segment length with the best block balance
        const int wb = 64 - 2 * (int)pad, tr = ((int)H + 3) / 4, tc = ((int)W + wb - 1) / wb;
        double best = 1e30;
        for (int L = 4; L <= 64; L *= 2) {
            const int64_t segs = 2 * S * N * tc * ((tr + L - 1) / L);
            const double cost = (double)((segs + g.G - 1) / g.G) * (std::min(L, tr) + 1.0);   // +1: ring warm-up per segment
            if (cost < best) { best = cost; g.L = L; }
        }
    }
    if (pad == 3) return launch_bwd3<20, 3>(st, x, y, g, win, grad_raw, scale, gx, gy, ws, nterms);
    return launch_bwd3<20, 1>(st, x, y, g, win, grad_raw, scale, gx, gy, ws, nterms);
}

}  // namespace miseg
