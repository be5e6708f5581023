// Backward of the local-MI joint, row-streaming form (bf16 matrix cores, hi/lo operand split = fp32-class accuracy).
//
// Per sub-head, direction and sample the backward is a T x T "convolution" with K x K channel mixing (ref
// contrastyou/losses/iic_loss.py:120-123 differentiated):
//     out[o][h][w] = sum_{a,b,c} Gsel[a][b][o,c] * S[c][h + s(a-p)][w + s(b-p)]          s = +1 (gY from X) / -1 (gX from Y)
// The earlier kernel (mi_local_bwd_bf16.hip) stacks (b,o) into the GEMM's M dimension and undoes the column shift b after the MFMAs
// with a gather through LDS ("col2im"): 64-column tiles then deliver 58 outputs, every wave re-reads the gradient matrix from L2 and
// the eight waves meet at two barriers per 4-row item -- its matrix pipe is ~45 % busy.  Here the ROW offset is stacked instead:
//     M = (tau, o),  tau = row slot:  source row hs contributes block tau to OUTPUT row  h = hs + p - tau
//     K = (beta, c), beta = column slot:  B[(beta,c)][w] = S[c][hs][w + beta - p]        (shifted views of ONE source row in LDS)
// and a wave walks DOWN a 64-column strip one source row at a time.  The T open output rows live in the MFMA accumulators; after a
// source row the row with tau = T-1 is finished and stored.
//   * no column halo: a 256-wide row is exactly 4 strips of 64 (was 5 tiles of 58 useful columns), no col2im, no LDS staging of D;
//   * the packed gradient matrix A (92 KB for K=20, pad 3, hi+lo) stays in LDS for as long as the block works on one (sub-head,
//     direction, window); fragments come from LDS (conflict-free ds_read_b128), never from L2 inside the loop;
//   * every wave owns its strip and its two private source-row buffers: the row loop has NO workgroup barrier at all; one wave per
//     SIMD (256 threads, up to 512 registers) software-pipelines the next row's global loads / hi-lo split / LDS commit against the
//     current row's MFMAs.
// The accumulators themselves never move: register slot j keeps the open output row h = j (mod T) from the source row that opens it
// to the one that finishes it; what rotates is the block of A a slot multiplies by -- tau_j = (hs + p - j) mod T -- i.e. an LDS
// address (all T blocks of A are resident), not a register index.  (Rotating the sums instead, through the MFMA's C operand, made the
// compiler shuttle 350 accumulator registers per source row between the two register files.)  The finished slot is selected by a
// wave-uniform switch, stored, and cleared: it is the slot the next source row opens its new output row in.
// K = 20 splits into 16 + 4 output channels: slots for o < 16 are T full 16-row tiles; the 4 remaining channels of all T slots share
// ceil(4T/16) tiles whose lane-row is the slot (per-lane A row addresses).  M = 16 T + 16 ceil(T/4) rows (144 for pad 3: the same
// tile count as the stacked-(b,o) form), Kred = T K (140 -> 160).
#include "mi_local.h"

#ifndef MISEG_ROWS_EDGE
#define MISEG_ROWS_EDGE 1       // 0: every slot every row; 1: skip slots whose output row is outside the unit (one code instance); 2: two instances
#endif
#ifndef MISEG_ROWS_SCHED
#define MISEG_ROWS_SCHED 0      // 0: a full scheduling barrier per tile (shipped); 1: none (A/B builds)
#endif

namespace miseg {

typedef __bf16 rbf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ru32x2 __attribute__((ext_vector_type(2)));

template <int K, int PAD, int NT_ = 4>
struct R3 {
    static_assert(K == 20, "R3: the 16 + 4 channel split is written for K = 20");
    static constexpr int T = 2 * PAD + 1, REM = K - 16, RT = (T * REM + 15) / 16, MT = T + RT, MP = MT * 16;
    static constexpr int KRED = T * K, KS = (KRED + 31) / 32;
    static constexpr int NT = NT_, WT = 16 * NT_, WS = WT + 2 * PAD, WSP = (WS + 7) / 8 * 8, CS = K;   // source row: WS pixels x CS channels (bf16), pixel-major
    static_assert(WT <= 64, "one lane per strip column");
    static constexpr int QD = ((T - 1) * REM % 16) / 4, TD = (T - 1) * REM / 16;     // lane-row / tile of the finished remainder slot
};

struct RowsGeom {
    int N, H, W, P, S, accumulate, G;
    long long hs;
};

// gpack[sdp][ks][pl][m][32], same swizzle as pack_g_bf16_kernel (conflict-free ds_read_b128): value(m = (tau,o), k = (beta,c)) =
// dir ? G[a][b][c][o] : G[a][b][o][c]  with  a = dir ? tau : T-1-tau,  b = dir ? beta : T-1-beta.
template <int K, int PAD, int NP>
__global__ void pack_g_rows_kernel(const float* __restrict__ grad_raw, int PS, unsigned short* __restrict__ gpack) {
    typedef R3<K, PAD> C;
    const int total = PS * 2 * C::KS * C::MP * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int kk = e & 31, m = (e >> 5) % C::MP, ks = (e / (32 * C::MP)) % C::KS, dir = (e / (32 * C::MP * C::KS)) & 1,
                  p = e / (32 * C::MP * C::KS * 2);
        const int kred = ks * 32 + kk;
        int tau, o;
        if (m < C::T * 16) { tau = m >> 4; o = m & 15; }
        else { const int rr = m - C::T * 16; tau = rr / C::REM; o = 16 + rr % C::REM; }
        float v = 0.f;
        if (tau < C::T && kred < C::KRED) {
            const int beta = kred / K, c = kred % K;
            const int a = dir ? tau : C::T - 1 - tau, b = dir ? beta : C::T - 1 - beta;
            const float* G = grad_raw + (size_t)p * C::T * C::T * K * K + (size_t)(a * C::T + b) * K * K;
            v = dir ? G[c * K + o] : G[o * K + c];
        }
        const unsigned short hi = f32_to_bf16_bits(v);
        const size_t base = ((((size_t)(p * 2 + dir) * C::KS + ks) * NP) * C::MP + m) * 32 + 8 * ((kk >> 3) ^ ((0 - (m >> 2)) & 3)) + (kk & 7);
        gpack[base] = hi;
        if (NP == 2) gpack[base + (size_t)C::MP * 32] = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
    }
}

template <int K, int PAD, int NTERMS, bool ACC, int NTW, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void local_bwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ y, RowsGeom g,
                                                                const int32_t* __restrict__ win,
                                                                const unsigned short* __restrict__ gpack,
                                                                const float* __restrict__ scale, float* __restrict__ gx,
                                                                float* __restrict__ gy) {
    typedef R3<K, PAD, NTW> C;
    constexpr int NP = NTERMS == 1 ? 1 : 2, T = C::T, RT = C::RT, NT = C::NT, KS = C::KS, MP = C::MP, CS = C::CS, WSP = C::WSP;
    constexpr int ASLICE = NP * MP * 32;                 // bf16 elements of A per k-step
    constexpr int BPLANE = WSP * CS;                     // bf16 elements of one source-row plane
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned short* Asm = reinterpret_cast<unsigned short*>(ldsb);                                  // [KS][NP][MP][32]
    unsigned short* Brows = Asm + (size_t)KS * ASLICE;                                              // [WAVES][2][NP][WSP][CS]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NBUF = (WAVES > 8 || (WAVES == 8 && NTW == 4)) ? 1 : 2;            // two waves per SIMD: the partner's MFMAs cover a wave's fetch, no in-wave prefetch (and its 40 registers)
    unsigned short* Bw = Brows + (size_t)wv * NBUF * NP * BPLANE;
    const size_t plane = (size_t)g.H * g.W;
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    constexpr unsigned OOB = 0xC0000000u;

    // ---- work = output rows.  All output rows of the launch in (direction, sub-head, window, sample, column strip, row) order form
    // one list; the G blocks take equal contiguous shares of it and, inside a block, the waves equal contiguous shares of the part of
    // the block's share that lies in one sdp = (sub-head, window, direction) -- the A image in LDS is per sdp, so a block changes it
    // at most once or twice, all waves together.  A wave walks down a strip for as long as its share lasts (a "unit": rows [r0, r1) of
    // one strip; it pays the 2 PAD warm-up source rows where it starts or changes strip).  Fixed 128-row units dealt round-robin
    // left 512 of the 3072 waves -- 36 whole CUs -- without work at the cfg2 shape.
    int64_t rows_head = 0;                                // output rows of one (direction, sub-head): sum over windows of N * strips * rows
    for (int p = 0; p < g.P; ++p) {
        const int tr = win[p * 4 + 1] - win[p * 4 + 0], tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WT - 1) / C::WT;
        rows_head += (int64_t)g.N * tc * tr;
    }
    const int64_t total_rows = rows_head * g.S * 2;
    const int64_t blk_lo = total_rows * blockIdx.x / g.G, blk_hi = total_rows * (blockIdx.x + 1) / g.G;
#pragma unroll 1
    for (int64_t cur = blk_lo; cur < blk_hi;) {
        // decode: cur -> (dir, s, p, row index inside the window's list)
        int64_t left = cur;
        const int dir = (int)(left / (rows_head * g.S));
        left -= (int64_t)dir * rows_head * g.S;
        const int s = (int)(left / rows_head);
        left -= (int64_t)s * rows_head;
        int p = 0, tr = 0, tc = 0;
        int64_t rp = 0;
        for (; p < g.P; ++p) {
            tr = win[p * 4 + 1] - win[p * 4 + 0];
            tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WT - 1) / C::WT;
            rp = (int64_t)g.N * tc * tr;
            if (left < rp) break;
            left -= rp;
        }
        const int64_t phase = min(blk_hi - cur, rp - left);                 // rows of this block's share inside (dir, s, p)
        cur += phase;
        const int h0w = win[p * 4 + 0], h1w = win[p * 4 + 1], w0w = win[p * 4 + 2], w1w = win[p * 4 + 3];
        const int sdp = ((s * g.P + p) * 2 + dir);
        {                                                // block-uniform: every phase is a different sdp
            __syncthreads();                             // every wave is done with the previous A image
            const unsigned short* src = gpack + (size_t)sdp * KS * ASLICE;
            for (int idx = tid; idx < KS * ASLICE / 8; idx += 64 * WAVES)
                *reinterpret_cast<ru32x4*>(Asm + (size_t)idx * 8) = *reinterpret_cast<const ru32x4*>(src + (size_t)idx * 8);
            __syncthreads();
        }
        const float* srcp = (dir ? x : y) + (size_t)s * g.hs;
        float* dstp = (dir ? gy : gx) + (size_t)s * g.hs;
        const float sc = scale[s * g.P + p];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, (int)tbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)dstp, 0, (int)tbytes, 0x00020000);
        int64_t wa = left + phase * wv / WAVES;
        const int64_t wb = left + phase * (wv + 1) / WAVES;
#pragma unroll 1
      while (wa < wb) {                                  // units of this wave (no barrier inside)
        const int strip = (int)(wa / tr), rr = (int)(wa - (int64_t)strip * tr);
        const int len = (int)min((int64_t)(tr - rr), wb - wa);
        wa += len;
        const int n = strip / tc, ct = strip - n * tc;
        const int col0 = w0w + ct * C::WT;
        const int r0 = h0w + rr, r1 = r0 + len;          // output rows of this unit

        // ---- source row hs -> registers: lane = pixel col0 - PAD + lane (+ 64 for the 2 PAD tail pixels), 20 channel planes
        constexpr bool TAIL = C::WS > 64;                  // a 64-column strip needs 2 PAD more pixels than a wave has lanes
        float pfa[K], pfb[TAIL ? K : 1];
        auto fetch_row = [&](int hsr) {
            const bool rok = hsr >= h0w && hsr < h1w;
            const int ca = col0 - PAD + lane, cb = ca + 64;
            const unsigned va = (rok && lane < C::WS && ca >= w0w && ca < w1w) ? (unsigned)ca * 4u : OOB;
            const unsigned vb = (rok && lane < C::WS - 64 && cb >= w0w && cb < w1w) ? (unsigned)cb * 4u : OOB;
            // per-lane byte offset of the column in voffset (out of range = dropped), the wave-uniform row / channel part in soffset:
            // no vector add per load
            // wave-uniform, and SAID so: left to itself the compiler folds `rok` into the lanes' column tests, keeps this offset in a VGPR
            // and wraps every load below in a waterfall loop (readfirstlane / compare / saveexec / branch per load)
            const unsigned rowoff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(n * K) * (unsigned)plane + (unsigned)(rok ? hsr : 0) * (unsigned)g.W) * 4u));
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const unsigned so = rowoff + (unsigned)c * (unsigned)plane * 4u;
                pfa[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)va, (int)so, 0));
                if (TAIL) pfb[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vb, (int)so, 0));
            }
        };
        // registers -> bf16 hi (+ lo) planes of row buffer `buf`, pixel-major [pixel][CS]: 4 channels per ds_write_b64
        auto commit_row = [&](int buf) {
            unsigned short* B0 = Bw + (size_t)buf * NP * BPLANE;
            auto put = [&](const float* v, int pix) {
#pragma unroll
                for (int c4 = 0; c4 < K; c4 += 4) {
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hi[j] = f32_to_bf16_bits(v[c4 + j]);
                        lo[j] = f32_to_bf16_bits(v[c4 + j] - bf16_bits_to_f32((unsigned short)hi[j]));
                    }
                    *reinterpret_cast<ru32x2*>(B0 + (size_t)pix * CS + c4) = ru32x2{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
                    if (NP == 2)
                        *reinterpret_cast<ru32x2*>(B0 + BPLANE + (size_t)pix * CS + c4) = ru32x2{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16)};
                }
            };
            if (C::WS >= 64 || lane < C::WS) put(pfa, lane);
            if (TAIL && lane < C::WS - 64) put(pfb, 64 + lane);
        };

        // Accumulators never move: slot j holds the open output row h = j (mod T).  What rotates is WHICH block of A a slot multiplies
        // by: at source row hs slot j takes block tau_j = (hs + p - j) mod T -- an LDS address, not a register index.
        f32x4 acc[T][NT], rem[RT][NT];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rem[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int hs_first = r0 - PAD, hs_last = r1 - 1 + PAD;
        fetch_row(hs_first);
        commit_row(0);
        const int aoff = (l15 * 32 + 8 * (q ^ ((0 - (l15 >> 2)) & 3)));       // this lane's 16 B inside a main M tile (bf16 elements)
        // B fragments: lane (n = l15, k-group q) holds k = ks*32 + 8q .. +7 = two 4-blocks (beta, c0..c0+3) that never straddle a beta
        auto boff_of = [&](int ks, int hf) {
            const int kb = min(ks * 32 + 8 * q + 4 * hf, C::KRED - 4);                  // k >= KRED: A is zero there, B only has to be finite
            const int beta = kb / K, c0 = kb - beta * K;
            return (l15 + beta) * CS + c0;
        };
        int boff[NBUF == 2 ? KS : 1][2];                   // kept in registers only where there are registers to spare (one wave per SIMD)
        if (NBUF == 2) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { boff[ks][0] = boff_of(ks, 0); boff[ks][1] = boff_of(ks, 1); }
        }
#pragma unroll 1
        for (int hsr = hs_first; hsr <= hs_last; ++hsr) {
            const int buf = NBUF == 2 ? (hsr - hs_first) & 1 : 0;
            if (NBUF == 2 && hsr < hs_last) fetch_row(hsr + 1);        // in flight during this row's MFMAs
            const unsigned short* Bb = Bw + (size_t)buf * NP * BPLANE;
            // ---- this row's A blocks per slot
            const int ph = hsr + PAD + T;                 // >= T > j
            int abase[T], arem[RT];
#pragma unroll
            for (int j = 0; j < T; ++j) abase[j] = aoff + ((ph - j) % T) * 16 * 32;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int jr = t * 4 + (l15 >> 2);
                const int taur = jr < T ? (ph - jr) % T : T;                            // unused lane-rows read the zero rows behind slot T-1
                const int mrow = T * 16 + taur * C::REM + (l15 & 3);
                arem[t] = mrow * 32 + 8 * (q ^ ((0 - (mrow >> 2)) & 3));
            }
            auto loadA = [&](int idx, rbf16x8_t* af) {    // idx = ks * MT + tile, tile < T: main slot, else remainder tile
                const int ks = idx / C::MT, i = idx - ks * C::MT;
                const int o = i < T ? abase[i < T ? i : 0] : arem[i >= T ? i - T : 0];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[pl] = *reinterpret_cast<const rbf16x8_t*>(Asm + (size_t)ks * ASLICE + (size_t)pl * MP * 32 + o);
            };
            auto loadB = [&](int ks, rbf16x8_t (*bf)[NT]) {
                const int b0 = NBUF == 2 ? boff[NBUF == 2 ? ks : 0][0] : boff_of(ks, 0), b1 = NBUF == 2 ? boff[NBUF == 2 ? ks : 0][1] : boff_of(ks, 1);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const ru32x2 lo = *reinterpret_cast<const ru32x2*>(Bb + pl * BPLANE + b0 + nt * 16 * CS);
                        const ru32x2 hi = *reinterpret_cast<const ru32x2*>(Bb + pl * BPLANE + b1 + nt * 16 * CS);
                        const ru32x4 f = {lo[0], lo[1], hi[0], hi[1]};
                        bf[pl][nt] = __builtin_bit_cast(rbf16x8_t, f);
                    }
            };
            const int jdone = (ph + 1) % T;                // slot with tau = T-1: finishes output row hsr - PAD (= the slot opened next row)
            const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
            const int jopen = ph % T;                      // slot with tau = 0: opens output row hsr + PAD
#pragma unroll
            for (int t = 0; t < RT; ++t) {                 // remainder tiles: the opening slot is a lane-row -> a select per register
                const bool opens = t * 4 + q == jopen;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rem[t][nt][r] = opens ? 0.f : rem[t][nt][r];
            }
            // Edge rows of a unit (the first and last 2 PAD source rows) feed output rows outside [r0, r1) through some slots: those
            // slots' MFMAs are skipped (16 % of a 32-row unit's matrix work).  Interior rows run the branch-free instance.
            bool open_row[T];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int hj = hsr + PAD - (ph - j) % T;
                open_row[j] = hj >= r0 && hj < r1;
            }
            auto mma_row = [&](auto EDGE_T) {
                constexpr bool EDGE = decltype(EDGE_T)::value;
            constexpr int BD = NBUF;                       // B-fragment sets in registers: double-buffered across k-steps only with one wave per SIMD
            rbf16x8_t afq[2][NP], bq[BD][NP][NT];
                loadB(0, bq[0]);
                loadA(0, afq[0]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int i = 0; i < C::MT; ++i) {
                        const int idx = ks * C::MT + i;
                        if (idx + 1 < KS * C::MT) loadA(idx + 1, afq[(idx + 1) & 1]);       // one tile ahead of the MFMAs that consume it
                        if (BD == 2 && i == 0 && ks + 1 < KS) loadB(ks + 1, bq[(ks + 1) & 1]);         // the next k-step's B fragments ride along
                        if (BD == 1 && i == 0 && ks > 0) loadB(ks, bq[0]);
#if MISEG_ROWS_SCHED == 0
                        __builtin_amdgcn_sched_barrier(0);    // keep the prefetches here: the scheduler sinks loads next to their use
#endif
                        const rbf16x8_t* af = afq[idx & 1];
                        if (EDGE && i < T && !open_row[i < T ? i : 0]) continue;      // this slot's output row lies outside the unit: skip its MFMAs
                        f32x4* d = i < T ? acc[i < T ? i : 0] : rem[i >= T ? i - T : 0];
                        // three products per tile, the four column tiles interleaved: an accumulator is touched every 4th MFMA
                        // The slot that opens a new output row (tau = 0) must start from zero, not from the row it finished one source row
                        // ago.  No register is cleared: its first MFMA group takes a zero C operand, behind a wave-uniform branch.
                        const rbf16x8_t a1 = NTERMS == 3 ? af[NP - 1] : af[0];
                        if (ks == 0 && i < T && i == jopen) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq[ks & (BD - 1)][0][nt], zero4, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq[ks & (BD - 1)][0][nt], d[nt], 0, 0, 0);
                        }
                        if (NTERMS == 3) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[ks & (BD - 1)][NP - 1][nt], d[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[ks & (BD - 1)][0][nt], d[nt], 0, 0, 0);
                        }
                    }
                }
            };
#if MISEG_ROWS_EDGE == 2
            if (hsr - PAD >= r0 && hsr + PAD < r1) mma_row(std::integral_constant<bool, false>{});
            else mma_row(std::integral_constant<bool, true>{});
#elif MISEG_ROWS_EDGE == 1
            mma_row(std::integral_constant<bool, true>{});      // one instance, a wave-uniform branch per slot and k-step
#else
            mma_row(std::integral_constant<bool, false>{});
#endif
            // ---- the slot with tau = T-1 now holds output row h = hsr - PAD: copy it out of the accumulators (wave-uniform switch: the
            // array needs a compile-time index; nothing is written back) and store it if it belongs to this unit
            const int h = hsr - PAD;
            const bool keep = h >= r0;                     // h < r1 by construction of hs_last
            const unsigned rowo = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(n * K) * (unsigned)plane + (unsigned)(keep ? h : 0) * (unsigned)g.W + (unsigned)col0) * 4u));   // wave-uniform
            f32x4 done[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) done[nt] = zero4;
            switch (jdone) {
#define MISEG_ROWS_CASE(J) case J: if (J < T) { _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) done[nt] = acc[J < T ? J : 0][nt]; } break;
                MISEG_ROWS_CASE(0) MISEG_ROWS_CASE(1) MISEG_ROWS_CASE(2) MISEG_ROWS_CASE(3) MISEG_ROWS_CASE(4)
                MISEG_ROWS_CASE(5) MISEG_ROWS_CASE(6) MISEG_ROWS_CASE(7) MISEG_ROWS_CASE(8)
#undef MISEG_ROWS_CASE
                default: break;
            }
            // voffset: lane-dependent part (column, channel 4q + r of a main tile / validity), soffset: the wave-uniform row offset
            // Store (or, for windows that share pixels with windows of earlier launches, add into) the finished row.  The accumulate form
            // issues ALL its loads before the first store: a load / add / store chain per element serialises on the memory round trip
            // (the compiler may not move a load above a store to the same buffer) -- 48 round trips per row and wave made the 49-window
            // launches of BASELINE configs[3] take 6.3 ms instead of 3.
            const unsigned qplane = (unsigned)(4 * q) * (unsigned)plane * 4u;
            unsigned vo_m[NT], vo_r[RT][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool cok = keep && col0 + nt * 16 + l15 < w1w;
                vo_m[nt] = cok ? qplane + (unsigned)(nt * 16 + l15) * 4u : OOB;
#pragma unroll
                for (int t = 0; t < RT; ++t) vo_r[t][nt] = (cok && t * 4 + q == jdone) ? (unsigned)(nt * 16 + l15) * 4u : OOB;
            }
            float old_m[NT][4], old_r[RT][NT][4];
            if (ACC) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        old_m[nt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rso, (int)vo_m[nt], (int)(rowo + (unsigned)r * (unsigned)plane * 4u), 0));
#pragma unroll
                        for (int t = 0; t < RT; ++t)
                            old_r[t][nt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rso, (int)vo_r[t][nt], (int)(rowo + (unsigned)(16 + r) * (unsigned)plane * 4u), 0));
                    }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = sc * done[nt][r] + (ACC ? old_m[nt][r] : 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rso, (int)vo_m[nt], (int)(rowo + (unsigned)r * (unsigned)plane * 4u), 0);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const float vr = sc * rem[t][nt][r] + (ACC ? old_r[t][nt][r] : 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vr), rso, (int)vo_r[t][nt], (int)(rowo + (unsigned)(16 + r) * (unsigned)plane * 4u), 0);
                    }
                }
            if (hsr < hs_last) {
                if (NBUF == 1) fetch_row(hsr + 1);         // this row's B reads are done (same wave: program order)
                commit_row(NBUF == 2 ? buf ^ 1 : 0);
            }
        }
      }
    }
}

template <int K, int PAD, int NTW, int WAVES>
static size_t rows_lds(int nterms) {
    typedef R3<K, PAD, NTW> C;
    const int np = nterms == 1 ? 1 : 2;
    return (size_t)C::KS * np * C::MP * 32 * 2 + (size_t)WAVES * ((WAVES > 8 || (WAVES == 8 && NTW == 4)) ? 1 : 2) * np * C::WSP * C::CS * 2;
}

bool local_bwd_bf16_supported(int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad) {
    return K == 20 && (pad == 3 || pad == 1) && (size_t)N * K * H * W * 4 < 0x40000000ull;       // 32-bit buffer offsets with an out-of-range marker
}

// packed gradient matrices of all (sub-head, window, direction) triples: the larger of the two kernels' images
size_t local_bwd_bf16_ws_bytes(int64_t K, int64_t pad, int64_t P) {
    const int T = 2 * (int)pad + 1, MP = (T + (T * 4 + 15) / 16) * 16, KS = (T * (int)K + 31) / 32;
    return std::max((size_t)P * 2 * KS * 2 * MP * 32 * 2, local_bwd_f8_ws_bytes(K, pad, P));
}

template <int K, int PAD, int NTW, int WAVES>
static int launch_rows(hipStream_t st, const float* x, const float* y, RowsGeom g, const int32_t* win, const float* grad_raw, const float* scale,
                       float* gx, float* gy, void* ws, int nterms) {
    typedef R3<K, PAD, NTW> C;
    unsigned short* gpack = reinterpret_cast<unsigned short*>(ws);
    const int total = g.P * g.S * 2 * C::KS * C::MP * 32;
    if (nterms == 1) hipLaunchKernelGGL((pack_g_rows_kernel<K, PAD, 1>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, g.P * g.S, gpack);
    else hipLaunchKernelGGL((pack_g_rows_kernel<K, PAD, 2>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, g.P * g.S, gpack);
    const size_t lds = rows_lds<K, PAD, NTW, WAVES>(nterms);
    auto go = [&](auto kernel) {
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(g.G), dim3(64 * WAVES), lds, st, x, y, g, win, gpack, scale, gx, gy);
    };
    if (nterms == 1) g.accumulate ? go(local_bwd_rows_kernel<K, PAD, 1, true, NTW, WAVES>) : go(local_bwd_rows_kernel<K, PAD, 1, false, NTW, WAVES>);
    else g.accumulate ? go(local_bwd_rows_kernel<K, PAD, 3, true, NTW, WAVES>) : go(local_bwd_rows_kernel<K, PAD, 3, false, NTW, WAVES>);
    return 0;
}

int launch_local_bwd_rows(hipStream_t st, const float* x, const float* y, int64_t S, int64_t hs, int64_t N, int64_t K, int64_t H, int64_t W,
                          int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale, float* gx, float* gy,
                          int accumulate, void* ws, int nterms, const unsigned char* planes) {
    if (nterms == 2) {        // f16 hi x hi + fp8 cross terms where that kernel exists (pad 3), the bf16 split elsewhere
        if (local_bwd_f8_supported(K, pad)) return launch_local_bwd_f8(st, x, y, S, hs, N, K, H, W, pad, win, P, grad_raw, scale, gx, gy, accumulate, ws, planes);
        nterms = 3;
    }
    RowsGeom g{(int)N, (int)H, (int)W, (int)P, (int)S, accumulate, 256, (long long)hs};
    // Shape of a block = 8 waves x 64-column strips.  Measured on the cfg2 launch (S=5, N=16, 256^2, pad 3; round 2): with the work dealt
    // in equal contiguous shares the shapes 12x32 / 8x64 / 8x32 / 4x64 are within 2 % of each other as kernels (1.06-1.12 ms; 4x64, one
    // wave per SIMD, 1.29-1.31), but the STEP is 0.05-0.06 ms shorter with 8x64 than with 12x32 (eight waves leave more of the CU to
    // the kernels of the other streams).
    if (pad == 3) return launch_rows<20, 3, 4, 8>(st, x, y, g, win, grad_raw, scale, gx, gy, ws, nterms);
    return launch_rows<20, 1, 4, 8>(st, x, y, g, win, grad_raw, scale, gx, gy, ws, nterms);
}

}  // namespace miseg
