// Backward of the local-MI joint, row-streaming form, with the fp32-class product split as  f16 hi x hi  +  fp8 cross terms.
//
// Same GEMM, work split and accumulator rotation as mi_local_bwd_rows.hip (ref contrastyou/losses/iic_loss.py:120-123
// differentiated: M = (row slot tau, class o), K = (column shift beta, class c), a wave streams source rows down a 64-column strip).
// What changes is the arithmetic of one fp32-class product g * s.  bf16x3 issues three bf16 MFMAs for it (hi*hi, hi*lo, lo*hi).
// Here both operands are split as  v = h + l,  h = f16(v) (11 significant bits),  l = v - h (|l| <= 2^-12 |v|), and
//     g * s  =  gh * sh                    one v_mfma_f32_16x16x32_f16 (16 cycles per 32 k)
//            +  gh * sl  +  gl * sh        both cross terms K-CONCATENATED into one block-scaled fp8 MFMA
//                                          (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands, 32 cycles per 128 k)
// The cross terms are 2^-12 of the product, so the 4 significant bits of e4m3 leave a 2^-16 relative error -- the class of the
// dropped lo*lo term; CPU emulation of the whole op on the shapes of tests/test_gpu_mi.py: loss error <= 5e-8 absolute on
// near-uniform heads, 2e-6 relative on peaked ones, gradients 3e-5 of their scale (bf16x3: 1e-8 / 3e-7 / 1e-5; bounds 5e-7 / 1e-5 /
// 1e-4).  Matrix-pipe time per source row and 16-column tile: 9 tiles x (5 x 16 + 3 x 32) = 1 584 cycles against 9 x 15 x 16 = 2 160;
// profiles/microbench/r03_mfma_dtypes.txt: this instruction mix runs at 0.64 of bf16x3's wall time per product block on
// random operand bits, continuous and in 1 ms bursts (per-clock rate and per-watt rate agree here).
//
// Operand images (layouts measured in profiles/microbench/r03_mx_layout_probe.txt: lane (l15, q) of the scaled MFMA holds bytes
// k = 32 q .. 32 q + 31 of row / column l15; one E8M0 scale per operand, the same for every lane):
//   * f16 planes exactly as the bf16 hi plane of the rows kernel: A [ks][m][32] swizzled, source rows [pixel][20] (40 B per pixel);
//   * 8-bit source planes [pixel][24 B] (20 classes + 4 zero bytes): k' = 24 beta + c is then an ADDRESS offset from the lane's own
//     pixel, a fragment = 32 contiguous bytes = four 8-byte-aligned ds_read_b64.  One term spans 7 x 24 = 168 -> 192 bytes = 6
//     chunks of 32; chunks 0-5 pair (gh8, sl8), chunks 6-11 pair (gl8, sh8): 12 chunks = 3 instructions, no lane straddles a term;
//   * 8-bit A [instr][16-row block][16-byte chunk 0..7][16 rows][16 B]: a fragment = two ds_read_b128, conflict-free (the 16 lanes of
//     a b128 lane group read 16 different 16-byte slots of 256-byte lines); a block is 2 048 B = 4 x the f16 image's 512 elements, so
//     the rotating block address of a slot is ONE register for both images (v_lshl_add).
// Scaling: G is pre-scaled per (sub-head, window) by 2^-e, e = floor(log2 max|G|), so that f16 holds it (the output scale takes
// 2^e back); gh8 = e4m3(2^7 gh), gl8 = e4m3(2^19 gl), sh8 = e4m3(2^8 s), sl8 = e4m3(2^20 sl): every operand <= 256 < 448 = e4m3 max,
// both cross terms carry 2^27, undone by the instruction's E8M0 scales (2^-13 x 2^-14).  Precondition: |s| <= 1.74 (probabilities).
//
// Shape of the kernel (round 3, measured at the cfg2 launch: S = 5, N = 16, 256 x 256, pad 3; bf16x3 rows kernel 1.09-1.11 ms):
//   * shipped: 8 waves x 64-column strips, every wave fetches, splits and multiplies (two waves per SIMD cover each other): 0.96-0.99
//     ms, matrix pipe 1 076 M busy cycles instead of 1 446 M.  What made it work: NO spill inside the row loop.  A lone register
//     reload from scratch costs ~450 cycles there (in-kernel s_memtime stamps), so every choice below is the one the register
//     allocator answers with zero reloads between a row's first and last MFMA: phase 1 keeps its branch per edge slot (straight-line
//     code: 17 reloads per row), phase 2 has none (a branch per tile there: 48 accumulator registers spilled), the f16 B fragment
//     address is one lane register plus immediates, both A images share the rotating block address.
//   * one loader wave + one MFMA wave per SIMD, hand-over through LDS sequence counters (built and measured, not kept): 1.21 ms.
//     The MFMA wave has nobody to hide its stalls behind and 256 registers do not hold 144 accumulators, 56 fragment registers and
//     the addresses without reloads in the loop.  (A counter store guarded by `lane == 0` made the allocator spill ALL accumulators
//     around it: 336 spilled registers against 28 with the store executed by every lane.)
//   * one wave per SIMD with 512 registers, next row's loads / split / LDS stores and the previous row's output stores placed by
//     hand between the MFMAs (built and measured, not kept): zero spills, 1.11 ms.  A lone wave issues ~1 instruction per 8
//     cycles and the filler instructions do NOT disappear behind 16- or 32-cycle MFMAs of these shapes: 180 + 108 MFMAs alone 4 050 +
//     3 770 cycles per row (ideal 2 880 + 3 456), with the ~400 filler instructions 6 200 + 6 500.
//   So this kernel is bound by instruction issue and LDS round trips around the MFMAs, not by the matrix pipe (~52 % busy): the
//   third of the MFMA cycles the split saves shows up as 11 %.  Where a wave's time goes in the shipped form (s_memtime stamps,
//   -DMISEG_F8_STAMP=1, cycles per source row and wave): fetch + split + commit 4 900, f16 phase 5 900, fp8 phase 4 900, output
//   switch + stores 1 700 = 17 400 for 2 x 6 336 cycles of MFMA per SIMD (73 % inside the loop); a wave walks 86-98 source rows for its
//   80 output rows (2 PAD warm-up rows per unit, one or two units per wave) and the launch ends with its slowest wave.  Raising the
//   issue priority of the non-MFMA phases (s_setprio) changed nothing.
#include "mi_local.h"

#ifndef MISEG_F8_D1
#define MISEG_F8_D1 1          // A fragments of phase 1 in flight ahead of the MFMAs (2, 3: no faster; a second B fragment set: slower, it spills)
#endif
#ifndef MISEG_F8_BQ
#define MISEG_F8_BQ 1
#endif
#ifndef MISEG_F8_EDGE1
#define MISEG_F8_EDGE1 1
#endif
#ifndef MISEG_F8_STAMP
#define MISEG_F8_STAMP 0     // diagnostic build: per-phase s_memtime sums of every wave -> the tail of the workspace (scratch/f8_stamps.py)
#endif
#ifndef MISEG_F8_STORE_AUX
#define MISEG_F8_STORE_AUX 0     // cache policy of the output stores (2 = non-temporal: measured)
#endif
#ifndef MISEG_F8_ABL
#define MISEG_F8_ABL 0       // ablation builds (scratch): 1 = no per-row fetch / commit, 2 = no output stores
#endif

namespace miseg {

template <int K, int PAD, int NT_ = 4>
struct Q3 {
    static_assert(K == 20, "Q3: the 16 + 4 channel split is written for K = 20");
    static constexpr int T = 2 * PAD + 1, REM = K - 16, RT = (T * REM + 15) / 16, MT = T + RT, MP = MT * 16;
    static constexpr int KRED = T * K, KS = (KRED + 31) / 32;
    static constexpr int CS8 = 24, CH = (T * CS8 + 31) / 32, NI = (2 * CH + 3) / 4;           // bytes per pixel, chunks per term, scaled MFMAs per tile
    static constexpr int NT = NT_, WT = 16 * NT_, WS = WT + 2 * PAD, WSP = (WS + 7) / 8 * 8, CS = K;
    static constexpr int B16P = WSP * CS * 2;                                                   // bytes of the f16 source-row plane
    static constexpr int B8P = (((WT - 1) * CS8 + CH * 32 > WS * CS8 ? (WT - 1) * CS8 + CH * 32 : WS * CS8) + 15) / 16 * 16;   // one 8-bit plane, read overhang included
    static constexpr int A16B = KS * MP * 64, A8B = NI * 8 * MP * 16;                           // bytes of the A images of one sdp
    static_assert(MP % 16 == 0, "A8 bank pattern");
};

struct Rows8Geom {
    int N, H, W, P, S, accumulate, G;
    long long hs;
};

// 2^floor(log2 max|G|) per (sub-head, window): the exponent f16 needs taken out of G
__global__ __launch_bounds__(1024) void gexp_kernel(const float* __restrict__ grad_raw, int per, float* __restrict__ gexp) {
    const float* G = grad_raw + (size_t)blockIdx.x * per;
    float m = 0.f;
    for (int e = threadIdx.x; e < per; e += 1024) m = fmaxf(m, fabsf(G[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
        gexp[blockIdx.x] = (m > 0.f && m < 3.0e38f) ? ldexpf(1.f, ilogbf(m)) : 1.f;
    }
}

// ---- operand planes in memory (miseg_hip.h: "local-MI operand planes"): per map and pixel the three images a source row of the backward
// is made of, pixel-major, so that a row of the kernel's LDS buffer is a plain copy of 70 consecutive pixels of each plane.
//   p16 [map][H][W][20] f16      p8l, p8h [map][H][W][24] e4m3 (classes 20..23 zero)
__global__ __launch_bounds__(256) void make_planes_kernel(const float* __restrict__ probs, int maps, int HW, unsigned char* __restrict__ p16,
                                                          unsigned char* __restrict__ p8l, unsigned char* __restrict__ p8h) {
    constexpr int K = 20;
    const int pix = blockIdx.x * 256 + threadIdx.x, map = blockIdx.y;
    if (pix >= HW || map >= maps) return;
    const float* src = probs + (size_t)map * K * HW + pix;
    const size_t at = (size_t)map * HW + pix;
    unsigned h[10], l[6], g[6];
#pragma unroll
    for (int c4 = 0; c4 < K; c4 += 4) {
        const float v[4] = {src[(size_t)c4 * HW], src[(size_t)(c4 + 1) * HW], src[(size_t)(c4 + 2) * HW], src[(size_t)(c4 + 3) * HW]};
        qu32x2 hh;
        split_quad(v, hh, l[c4 / 4], g[c4 / 4]);
        h[c4 / 2] = hh[0], h[c4 / 2 + 1] = hh[1];
    }
    l[5] = 0u, g[5] = 0u;
    qu32x2* o16 = reinterpret_cast<qu32x2*>(p16 + at * 40);
#pragma unroll
    for (int i = 0; i < 5; ++i) o16[i] = qu32x2{h[2 * i], h[2 * i + 1]};
    qu32x2* ol = reinterpret_cast<qu32x2*>(p8l + at * 24);
    qu32x2* oh = reinterpret_cast<qu32x2*>(p8h + at * 24);
#pragma unroll
    for (int i = 0; i < 3; ++i) ol[i] = qu32x2{l[2 * i], l[2 * i + 1]}, oh[i] = qu32x2{g[2 * i], g[2 * i + 1]};
}

// 16 bytes per lane, memory -> LDS without a register in between: lane i's bytes land at lds_dst + 16 i (lds_dst wave-uniform).  The
// compiler does not count this load: the caller waits for it (s_waitcnt vmcnt) before the wave reads the bytes.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// value(m = (tau,o), beta, c) of direction dir, pre-scaled
template <int K, int PAD>
__device__ __forceinline__ float g_value(const float* __restrict__ grad_raw, int p, int dir, int m, int beta, int c, float inv) {
    typedef Q3<K, PAD> C;
    int tau, o;
    if (m < C::T * 16) { tau = m >> 4; o = m & 15; }
    else { const int rr = m - C::T * 16; tau = rr / C::REM; o = 16 + rr % C::REM; }
    if (tau >= C::T || beta >= C::T || c >= K) return 0.f;
    const int a = dir ? tau : C::T - 1 - tau, b = dir ? beta : C::T - 1 - beta;
    const float* G = grad_raw + (size_t)p * C::T * C::T * K * K + (size_t)(a * C::T + b) * K * K;
    return (dir ? G[c * K + o] : G[o * K + c]) * inv;
}

// gpack[sdp] = [A16: [ks][m][32] f16, the rows kernel's swizzle][A8: [instr][m / 16][chunk][m % 16][16 B]]
template <int K, int PAD>
__global__ void pack_g_f8_kernel(const float* __restrict__ grad_raw, const float* __restrict__ gexp, int PS, unsigned char* __restrict__ gpack) {
    typedef Q3<K, PAD> C;
    constexpr int N16 = C::KS * C::MP * 32, N8 = C::NI * 8 * C::MP * 4;       // f16 elements / dwords of 8-bit data per sdp
    const int total = PS * 2 * (N16 + N8);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int sdp = e / (N16 + N8), r = e - sdp * (N16 + N8), dir = sdp & 1, p = sdp >> 1;
        const float inv = 1.f / gexp[p];
        unsigned char* base = gpack + (size_t)sdp * (C::A16B + C::A8B);
        if (r < N16) {
            const int kk = r & 31, m = (r >> 5) % C::MP, ks = r / (32 * C::MP);
            const int kred = ks * 32 + kk;
            const float v = kred < C::KRED ? g_value<K, PAD>(grad_raw, p, dir, m, kred / K, kred % K, inv) : 0.f;
            const _Float16 h = (_Float16)v;
            const size_t at = ((size_t)ks * C::MP + m) * 32 + 8 * ((kk >> 3) ^ ((0 - (m >> 2)) & 3)) + (kk & 7);
            reinterpret_cast<_Float16*>(base)[at] = h;
        } else {
            const int d = r - N16, b4 = d & 3, chunk = (d >> 6) & 7, m = ((d >> 9) % C::MT) * 16 + ((d >> 2) & 15), t = d / (512 * C::MT);
            const int j = 4 * t + (chunk >> 1), term = j / C::CH;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int byte = (chunk & 1) * 16 + b4 * 4 + i, kk = (j % C::CH) * 32 + byte;
                float g = j < 2 * C::CH ? g_value<K, PAD>(grad_raw, p, dir, m, kk / C::CS8, kk % C::CS8, inv) : 0.f;
                const float h = (float)(_Float16)g;
                v[i] = term == 0 ? h * 128.f : (g - h) * 524288.f;          // gh8 = e4m3(2^7 gh), gl8 = e4m3(2^19 gl)
            }
            reinterpret_cast<unsigned*>(base + C::A16B)[d] = pack_fp8x4(v[0], v[1], v[2], v[3]);
        }
    }
}

// PL: the source rows come from the operand planes (x, y unused): a row is seven 16-byte-per-lane copies memory -> LDS, issued as soon
// as the phase that reads the plane's previous row is over and waited for where the next row's phase begins -- no registers, no
// conversions, no exposed round trip (the register loader has to fetch AFTER the row's MFMAs: 40 live registers do not fit beside
// 180 accumulators).  Pixels outside the window are zeroed in LDS once the copy has landed (strips on a window edge only).
template <int K, int PAD, bool ACC, int NTW, int WAVES, bool PL>
__global__ __launch_bounds__(64 * WAVES, 1) void local_bwd_f8_kernel(const float* __restrict__ x, const float* __restrict__ y, Rows8Geom g,
                                                              const int32_t* __restrict__ win,
                                                              const unsigned char* __restrict__ gpack, const float* __restrict__ gexp,
                                                              const float* __restrict__ scale, float* __restrict__ gx,
                                                              float* __restrict__ gy, unsigned long long* __restrict__ stamps,
                                                              const unsigned char* __restrict__ pl16, const unsigned char* __restrict__ pl8l,
                                                              const unsigned char* __restrict__ pl8h) {
    typedef Q3<K, PAD, NTW> C;
    constexpr int T = C::T, RT = C::RT, NT = C::NT, KS = C::KS, NI = C::NI, MP = C::MP, CS = C::CS, CS8 = C::CS8;
    constexpr int BWB = C::B16P + 2 * C::B8P;            // bytes of one wave's source-row buffer: f16 plane, 8-bit lo plane, 8-bit hi plane
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const _Float16* Asm16 = reinterpret_cast<const _Float16*>(ldsb);                                 // [KS][MP][32]
    const unsigned char* Asm8 = ldsb + C::A16B;                                                      // [NI][MT][8][16][16]
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* Bw = ldsb + C::A16B + C::A8B + (size_t)wv * BWB;
    const size_t plane = (size_t)g.H * g.W;
    const unsigned tbytes = (unsigned)((size_t)g.N * K * plane * 4);
    constexpr unsigned OOB = 0xC0000000u;
    const int sa = 127 - 13, sb = 127 - 14;               // E8M0: 2^-13 x 2^-14 undo the operands' 2^27

    // the 8-bit planes' pad bytes (classes 20..23 of a pixel, pixels >= WS) meet zero rows of A: they only have to be finite, i.e.
    // never 0x7F / 0xFF -- cleared once; commit_row rewrites whole pixels (24 bytes) afterwards
    for (int i = lane; i < BWB / 4; i += 64) reinterpret_cast<unsigned*>(Bw)[i] = 0u;

#if MISEG_F8_STAMP
    unsigned long long st_fetch = 0, st_p1 = 0, st_p2 = 0, st_out = 0, st_t = __builtin_amdgcn_s_memtime(), st_rows = 0;
#define F8_STAMP(acc_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - st_t; st_t = now_; }
#else
#define F8_STAMP(acc_)
#endif
    // ---- work = output rows, dealt as in local_bwd_rows_kernel: equal contiguous shares per block, per wave inside one sdp
    int64_t rows_head = 0;
    for (int p = 0; p < g.P; ++p) {
        const int tr = win[p * 4 + 1] - win[p * 4 + 0], tc = (win[p * 4 + 3] - win[p * 4 + 2] + C::WT - 1) / C::WT;
        rows_head += (int64_t)g.N * tc * tr;
    }
    const int64_t total_rows = rows_head * g.S * 2;
    const int64_t blk_lo = total_rows * blockIdx.x / g.G, blk_hi = total_rows * (blockIdx.x + 1) / g.G;
#pragma unroll 1
    for (int64_t cur = blk_lo; cur < blk_hi;) {
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
        const int64_t phase = min(blk_hi - cur, rp - left);
        cur += phase;
        const int h0w = win[p * 4 + 0], h1w = win[p * 4 + 1], w0w = win[p * 4 + 2], w1w = win[p * 4 + 3];
        const int sdp = ((s * g.P + p) * 2 + dir);
        {
            __syncthreads();
            const unsigned char* src = gpack + (size_t)sdp * (C::A16B + C::A8B);
            for (int idx = tid; idx < (C::A16B + C::A8B) / 16; idx += 64 * WAVES)
                *reinterpret_cast<qu32x4*>(ldsb + (size_t)idx * 16) = *reinterpret_cast<const qu32x4*>(src + (size_t)idx * 16);
            __syncthreads();
        }
        const float* srcp = (dir ? x : y) + (size_t)s * g.hs;
        float* dstp = (dir ? gy : gx) + (size_t)s * g.hs;
        const float sc = scale[s * g.P + p] * gexp[s * g.P + p];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, (int)tbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)dstp, 0, (int)tbytes, 0x00020000);
        int64_t wa = left + phase * wv / WAVES;
        const int64_t wb = left + phase * (wv + 1) / WAVES;
#pragma unroll 1
      while (wa < wb) {
        const int strip = (int)(wa / tr), rr = (int)(wa - (int64_t)strip * tr);
        const int len = (int)min((int64_t)(tr - rr), wb - wa);
        wa += len;
        const int n = strip / tc, ct = strip - n * tc;
        const int col0 = w0w + ct * C::WT;
        const int r0 = h0w + rr, r1 = r0 + len;

        constexpr bool TAIL = C::WS > 64;
        float pfa[K], pfb[TAIL ? K : 1];
        auto fetch_row = [&](int hsr) {
            const bool rok = hsr >= h0w && hsr < h1w;
            const int ca = col0 - PAD + lane, cb = ca + 64;
            const unsigned va = (rok && lane < C::WS && ca >= w0w && ca < w1w) ? (unsigned)ca * 4u : OOB;
            const unsigned vb = (rok && lane < C::WS - 64 && cb >= w0w && cb < w1w) ? (unsigned)cb * 4u : OOB;
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
        // registers -> the three planes of the wave's row buffer, pixel-major: f16 [pixel][20], sl8 [pixel][24], sh8 [pixel][24]
        auto commit_row = [&]() {
            auto put = [&](const float* v, int pix) {
                unsigned char* p16 = Bw + (size_t)pix * (CS * 2);
                unsigned char* p8l = Bw + C::B16P + (size_t)pix * CS8;
                unsigned char* p8h = p8l + C::B8P;
#pragma unroll
                for (int c4 = 0; c4 < K; c4 += 4) {
                    qu32x2 h16;
                    unsigned l8, h8;
                    split_quad(v + c4, h16, l8, h8);
                    *reinterpret_cast<qu32x2*>(p16 + c4 * 2) = h16;
                    *reinterpret_cast<unsigned*>(p8l + c4) = l8;
                    *reinterpret_cast<unsigned*>(p8h + c4) = h8;
                }
            };
            if (C::WS >= 64 || lane < C::WS) put(pfa, lane);
            if (TAIL && lane < C::WS - 64) put(pfb, 64 + lane);
        };

        // ---- PL: the planes' rows.  Row hsr of map (s, dir ? x : y, n), pixels col0 - PAD .. + 69: 2 800 bytes of p16 = 175 lanes x 16 B,
        // 1 680 bytes of each 8-bit plane = 105.  A row outside the window is zeros, written by the wave itself.
        const unsigned ldsBw = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bw;
        const long long pix00 = ((long long)((long long)s * 2 * g.N + (dir ? 0 : g.N) + n) * g.H) * g.W + (col0 - PAD);
        const bool edge = col0 - PAD < w0w || col0 - PAD + C::WS > w1w;        // wave-uniform: some of the row's pixels lie outside the window
        auto row_in = [&](int hsr) { return hsr >= h0w && hsr < h1w; };
        auto zero_span = [&](unsigned char* dst, int bytes) {
            for (int i = lane * 16; i < bytes; i += 64 * 16) *reinterpret_cast<qu32x4*>(dst + i) = qu32x4{0u, 0u, 0u, 0u};
        };
        auto issue16 = [&](int hsr) {
            static_assert(C::WS * CS * 2 == 2800 && C::B16P >= 2800 && C::B16P % 16 == 0, "row geometry of the f16 plane copy");
            if (row_in(hsr)) {
                const unsigned char* src = pl16 + (pix00 + (long long)hsr * g.W) * (CS * 2) + lane * 16;
                glds16(src, ldsBw);
                glds16(src + 1024, ldsBw + 1024);
                if (lane < 175 - 128) glds16(src + 2048, ldsBw + 2048);
            } else zero_span(Bw, C::B16P);
        };
        auto issue8 = [&](int hsr) {
            static_assert(C::WS * CS8 == 1680 && C::B8P >= 1680 && C::B8P % 16 == 0, "row geometry of the 8-bit plane copies");
            if (row_in(hsr)) {
                const long long at = (pix00 + (long long)hsr * g.W) * CS8 + lane * 16;
                glds16(pl8l + at, ldsBw + C::B16P);
                if (lane < 105 - 64) glds16(pl8l + at + 1024, ldsBw + C::B16P + 1024);
                glds16(pl8h + at, ldsBw + C::B16P + C::B8P);
                if (lane < 105 - 64) glds16(pl8h + at + 1024, ldsBw + C::B16P + C::B8P + 1024);
            } else zero_span(Bw + C::B16P, 2 * C::B8P);
        };
        // after the copy has landed: pixels outside the window's columns -> zeros (what the register loader's OOB loads return)
        auto fix16 = [&]() {
            if (!edge) return;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int pix = half * 64 + lane, col = col0 - PAD + pix;
                if (pix < C::WS && (col < w0w || col >= w1w)) {
#pragma unroll
                    for (int i = 0; i < CS * 2; i += 8) *reinterpret_cast<qu32x2*>(Bw + pix * (CS * 2) + i) = qu32x2{0u, 0u};
                }
            }
        };
        auto fix8 = [&]() {
            if (!edge) return;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int pix = half * 64 + lane, col = col0 - PAD + pix;
                if (pix < C::WS && (col < w0w || col >= w1w)) {
#pragma unroll
                    for (int i = 0; i < CS8; i += 8) {
                        *reinterpret_cast<qu32x2*>(Bw + C::B16P + pix * CS8 + i) = qu32x2{0u, 0u};
                        *reinterpret_cast<qu32x2*>(Bw + C::B16P + C::B8P + pix * CS8 + i) = qu32x2{0u, 0u};
                    }
                }
            }
        };

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
        if constexpr (PL) {
            issue16(hs_first);
            issue8(hs_first);
            wait_vm<0>();
            fix16();
            fix8();
        } else {
            fetch_row(hs_first);
            commit_row();
        }
        const int aoff = (l15 * 32 + 8 * (q ^ ((0 - (l15 >> 2)) & 3)));       // this lane's 16 B inside a main M tile of the f16 image (elements)
        // f16 B fragment of k-step ks = two 4-blocks k = 32 ks + 8 q + 4 hf .. + 3 of (beta, c) = (k / 20, k % 20) at pixel l15 + beta:
        // pixel-major rows of exactly 20 classes make that the byte address 40 l15 + 2 k -- one lane register plus immediates.  Steps past
        // k = 140 read on into the following pixels (finite data or the zeroed pad pixels; A is zero there).
        const int b16lane = l15 * (CS * 2) + 16 * q;
        static_assert(NI == 3 && C::CH == 6, "chunk -> plane map below is written out for pad 3");
        // t = 0: chunks 0-3 (term 0, 32 q); t = 1: chunks 4, 5 (term 0) | 6, 7 (term 1, chunk 0, 1); t = 2: chunks 8-11 = term 1 at 64 + 32 q
        const int b8off0 = C::B16P + 32 * q + l15 * CS8, b8off1 = b8off0 + (q < 2 ? 128 : C::B8P - 64);
        constexpr int B8OFF2 = C::B8P + 64;
        const int a8lane = 512 * q + 16 * l15 - 4 * aoff;
#pragma unroll 1
        for (int hsr = hs_first; hsr <= hs_last; ++hsr) {
            const int ph = hsr + PAD + T;
            int abase[T], arem[RT], a8rem[RT];
#pragma unroll
            for (int j = 0; j < T; ++j) abase[j] = aoff + ((ph - j) % T) * 16 * 32;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int jr = t * 4 + (l15 >> 2);
                const int taur = jr < T ? (ph - jr) % T : T;
                const int mrow = T * 16 + taur * C::REM + (l15 & 3);
                arem[t] = mrow * 32 + 8 * (q ^ ((0 - (mrow >> 2)) & 3));
                a8rem[t] = (mrow >> 4) * 2048 + 512 * q + (mrow & 15) * 16;
            }
            auto loadA16 = [&](int idx, qh8_t& af) {      // idx = ks * MT + tile
                const int ks = idx / C::MT, i = idx - ks * C::MT;
                const int o = i < T ? abase[i < T ? i : 0] : arem[i >= T ? i - T : 0];
                af = *reinterpret_cast<const qh8_t*>(Asm16 + (size_t)ks * MP * 32 + o);
            };
            auto loadA8 = [&](int idx, qi8_t& af) {       // idx = t * MT + tile
                const int t = idx / C::MT, i = idx - t * C::MT;
                const int o = (i < T ? 4 * abase[i < T ? i : 0] + a8lane : a8rem[i >= T ? i - T : 0]) + t * C::MT * 2048;
                const qu32x4 lo = *reinterpret_cast<const qu32x4*>(Asm8 + o), hi = *reinterpret_cast<const qu32x4*>(Asm8 + o + 256);
                af = qi8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            };
            auto loadB16 = [&](int ks, qh8_t* bf) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const qu32x2 lo = *reinterpret_cast<const qu32x2*>(Bw + b16lane + ks * 64 + nt * 16 * CS * 2);
                    const qu32x2 hi = *reinterpret_cast<const qu32x2*>(Bw + b16lane + ks * 64 + 8 + nt * 16 * CS * 2);
                    const qu32x4 f = {lo[0], lo[1], hi[0], hi[1]};
                    bf[nt] = __builtin_bit_cast(qh8_t, f);
                }
            };
            auto loadB16n = [&](int ks, int nt, qh8_t& bf) {
                const qu32x2 lo = *reinterpret_cast<const qu32x2*>(Bw + b16lane + ks * 64 + nt * 16 * CS * 2);
                const qu32x2 hi = *reinterpret_cast<const qu32x2*>(Bw + b16lane + ks * 64 + 8 + nt * 16 * CS * 2);
                const qu32x4 f = {lo[0], lo[1], hi[0], hi[1]};
                bf = __builtin_bit_cast(qh8_t, f);
            };
            auto loadB8 = [&](int t, qi8_t* bf) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const unsigned char* b = Bw + (t == 0 ? b8off0 : t == 1 ? b8off1 : b8off0 + B8OFF2) + nt * 16 * CS8;
                    const qu32x2 r0 = *reinterpret_cast<const qu32x2*>(b), r1 = *reinterpret_cast<const qu32x2*>(b + 8),
                                 r2 = *reinterpret_cast<const qu32x2*>(b + 16), r3 = *reinterpret_cast<const qu32x2*>(b + 24);
                    bf[nt] = qi8_t{(int)r0[0], (int)r0[1], (int)r1[0], (int)r1[1], (int)r2[0], (int)r2[1], (int)r3[0], (int)r3[1]};
                }
            };
            const int jdone = (ph + 1) % T;
            const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
            const int jopen = ph % T;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const bool opens = t * 4 + q == jopen;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rem[t][nt][r] = opens ? 0.f : rem[t][nt][r];
            }
            bool open_row[T];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int hj = hsr + PAD - (ph - j) % T;
                open_row[j] = hj >= r0 && hj < r1;
            }
            if constexpr (PL) {
                if (hsr > hs_first) {          // this row's f16 plane, issued after the previous row's phase 1; its four 8-bit copies may still fly
                    wait_vm<4>();
                    fix16();
                }
            }
            F8_STAMP(st_fetch)
            {
                // ---- phase 1: hi x hi on the f16 pipe; edge slots (output row outside the unit) skipped behind a wave-uniform branch per tile
                constexpr int D1 = MISEG_F8_D1, NBQ = MISEG_F8_BQ;
                qh8_t aq[D1 + 1], bq[NBQ][NT];
                loadB16(0, bq[0]);
#pragma unroll
                for (int d = 0; d < D1; ++d) loadA16(d, aq[d]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int i = 0; i < C::MT; ++i) {
                        const int idx = ks * C::MT + i;
                        if (idx + D1 < KS * C::MT) loadA16(idx + D1, aq[(idx + D1) % (D1 + 1)]);
                        if (NBQ == 1 && i == 0 && ks > 0) loadB16(ks, bq[0]);
                        if (NBQ == 2 && i < NT && ks + 1 < KS) loadB16n(ks + 1, i, bq[(ks + 1) & 1][i]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (MISEG_F8_EDGE1 && i < T && !open_row[i < T ? i : 0]) continue;
                        f32x4* d = i < T ? acc[i < T ? i : 0] : rem[i >= T ? i - T : 0];
                        const qh8_t a = aq[idx % (D1 + 1)];
                        if (ks == 0 && i < T && i == jopen) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq[ks & (NBQ - 1)][nt], zero4, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq[ks & (NBQ - 1)][nt], d[nt], 0, 0, 0);
                        }
                    }
                }
            }
            F8_STAMP(st_p1)
            __builtin_amdgcn_sched_barrier(0);            // phase 2's first fragment loads stay behind phase 1's last MFMAs (they would not fit beside its registers)
            if constexpr (PL) {
                const bool more = hsr < hs_last;
                if (more) issue16(hsr + 1);               // the f16 plane is free: the next row's copy flies during phase 2
                if (hsr > hs_first) {                     // this row's 8-bit planes, issued after the previous row's stores
                    if (more && row_in(hsr + 1)) wait_vm<3>(); else wait_vm<0>();
                    fix8();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                // ---- phase 2: both cross terms, K-concatenated, on the block-scaled fp8 pipe
                qi8_t a8q[2], b8[NT];
                loadB8(0, b8);
                loadA8(0, a8q[0]);
#pragma unroll
                for (int t = 0; t < NI; ++t) {
#pragma unroll
                    for (int i = 0; i < C::MT; ++i) {
                        const int idx = t * C::MT + i;
                        if (idx + 1 < NI * C::MT) loadA8(idx + 1, a8q[(idx + 1) & 1]);
                        if (i == 0 && t > 0) loadB8(t, b8);
                        __builtin_amdgcn_sched_barrier(0);
                        // no edge-slot skipping here: a branch around every tile of this phase too made the register allocator spill 48
                        // accumulator registers inside the row loop (117 spilled registers against 32, none of them accumulators); a closed
                        // slot's sums are never stored, so the extra products (~3 % of a wave's work) are harmless
                        f32x4* d = i < T ? acc[i < T ? i : 0] : rem[i >= T ? i - T : 0];
                        const qi8_t a = a8q[idx & 1];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b8[nt], d[nt], 0, 0, 0, sa, 0, sb);
                    }
                }
            }
            F8_STAMP(st_p2)
            // ---- the slot with tau = T-1 now holds output row h = hsr - PAD
            const int h = hsr - PAD;
            const bool keep = h >= r0;
            const unsigned rowo = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(n * K) * (unsigned)plane + (unsigned)(keep ? h : 0) * (unsigned)g.W + (unsigned)col0) * 4u));
            f32x4 done[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) done[nt] = zero4;
            switch (jdone) {
#define MISEG_ROWS8_CASE(J) case J: if (J < T) { _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) done[nt] = acc[J < T ? J : 0][nt]; } break;
                MISEG_ROWS8_CASE(0) MISEG_ROWS8_CASE(1) MISEG_ROWS8_CASE(2) MISEG_ROWS8_CASE(3) MISEG_ROWS8_CASE(4)
                MISEG_ROWS8_CASE(5) MISEG_ROWS8_CASE(6) MISEG_ROWS8_CASE(7) MISEG_ROWS8_CASE(8)
#undef MISEG_ROWS8_CASE
                default: break;
            }
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
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rso, (int)vo_m[nt], (int)(rowo + (unsigned)r * (unsigned)plane * 4u), MISEG_F8_STORE_AUX);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const float vr = sc * rem[t][nt][r] + (ACC ? old_r[t][nt][r] : 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vr), rso, (int)vo_r[t][nt], (int)(rowo + (unsigned)(16 + r) * (unsigned)plane * 4u), MISEG_F8_STORE_AUX);
                    }
                }
            F8_STAMP(st_out)
#if MISEG_F8_STAMP
            ++st_rows;
#endif
#if !(MISEG_F8_ABL & 1)
            if (hsr < hs_last) {
                if constexpr (PL) issue8(hsr + 1);         // behind the stores: `vmcnt(4)` at the top of the next row then covers stores + f16 copies
                else {
                    fetch_row(hsr + 1);                    // this row's B reads are done (same wave: program order)
                    commit_row();
                }
            }
#endif
        }
      }
    }
#if MISEG_F8_STAMP
    if (lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * WAVES + wv) * 8;
        o[0] = st_fetch; o[1] = st_p1; o[2] = st_p2; o[3] = st_out; o[4] = st_rows;
    }
#endif
}

size_t local_bwd_f8_ws_bytes(int64_t K, int64_t pad, int64_t P) {
    if (K != 20 || pad != 3) return 0;
    typedef Q3<20, 3> C;
    return (size_t)256 + (size_t)((P * 4 + 255) / 256 * 256) + (size_t)P * 2 * (C::A16B + C::A8B) + (MISEG_F8_STAMP ? 256 * 8 * 64 : 0);
}

bool local_bwd_f8_supported(int64_t K, int64_t pad) { return K == 20 && pad == 3; }

int launch_local_bwd_f8(hipStream_t st, const float* x, const float* y, int64_t S, int64_t hs, int64_t N, int64_t K, int64_t H, int64_t W,
                        int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale, float* gx, float* gy,
                        int accumulate, void* ws, const unsigned char* planes) {
    typedef Q3<20, 3, 4> C;
    constexpr int WAVES = 8;
    Rows8Geom g{(int)N, (int)H, (int)W, (int)P, (int)S, accumulate, 256, (long long)hs};
    const int PS = (int)(P * S);
    float* gexp = reinterpret_cast<float*>(ws);
    unsigned char* gpack = reinterpret_cast<unsigned char*>(ws) + (size_t)((PS * 4 + 255) / 256 * 256);
    hipLaunchKernelGGL(gexp_kernel, dim3(PS), dim3(1024), 0, st, grad_raw, C::T * C::T * 20 * 20, gexp);
    const int total = PS * 2 * (C::KS * C::MP * 32 + C::NI * 8 * C::MP * 4);
    hipLaunchKernelGGL((pack_g_f8_kernel<20, 3>), dim3((total + 255) / 256), dim3(256), 0, st, grad_raw, gexp, PS, gpack);
    const size_t lds = (size_t)C::A16B + C::A8B + (size_t)WAVES * (C::B16P + 2 * C::B8P);
    const MiPlanes pl = mi_planes(const_cast<unsigned char*>(planes), S * 2 * N, H * W);
    auto go = [&](auto kernel) {
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(g.G), dim3(64 * WAVES), lds, st, x, y, g, win, gpack, gexp, scale, gx, gy,
                           reinterpret_cast<unsigned long long*>(gpack + (size_t)PS * 2 * (C::A16B + C::A8B)), pl.p16, pl.p8l, pl.p8h);
    };
    if (planes) {
        if (g.accumulate) go(local_bwd_f8_kernel<20, 3, true, 4, WAVES, true>);
        else go(local_bwd_f8_kernel<20, 3, false, 4, WAVES, true>);
    } else {
        if (g.accumulate) go(local_bwd_f8_kernel<20, 3, true, 4, WAVES, false>);
        else go(local_bwd_f8_kernel<20, 3, false, 4, WAVES, false>);
    }
    return 0;
}

int launch_make_planes(hipStream_t st, const float* probs, int64_t maps, int64_t HW, unsigned char* planes) {
    const MiPlanes pl = mi_planes(planes, maps, HW);
    hipLaunchKernelGGL(make_planes_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)maps), dim3(256), 0, st, probs, (int)maps, (int)HW, pl.p16, pl.p8l,
                       pl.p8h);
    return 0;
}

}  // namespace miseg
