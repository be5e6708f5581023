// Shared geometry of the local-MI joint kernels (fp32 and bf16 paths).
#pragma once
#include "common.h"

namespace miseg {

constexpr int kThreads = 256;         // 4 waves, one per SIMD: each wave owns a full D accumulator
constexpr int kLdsBudget = 156 * 1024;

struct JointGeom {
    int N, K, H, W, pad, T, Mdim;
    int RB, WB, RW;          // block tile rows / cols, rows per wave
    int WX, RBY;             // X tile width (WB+2pad), Y tile rows (RB+2pad)
    int planeX, planeY;      // LDS floats per channel plane (odd => conflict-poor gathers)
    int P, G;                // windows, persistent blocks per (window, sub-block)
    int tilesM, sb, tps;     // 16-row tiles of D, sub-blocks per dim, tiles per sub-block
    int S;                   // sub-heads batched into one launch (bf16 kernels): slot = s * P + p, operands at + s * hs
    long long hs;            // elements between consecutive sub-heads of x / y
};

constexpr int kJT = 512;

template <int MT, int NT, int ROLE>
struct TileSet {
    static constexpr int SPLIT = (MT * NT + 1) / 2;
    static constexpr bool mine(int m, int n) { return ((m * NT + n) < SPLIT) == (ROLE == 0); }
    static constexpr bool row_used(int m) {
        for (int n = 0; n < NT; ++n)
            if (mine(m, n)) return true;
        return false;
    }
    static constexpr bool col_used(int n) {
        for (int m = 0; m < MT; ++m)
            if (mine(m, n)) return true;
        return false;
    }
};

static inline bool plan_joint(JointGeom& g, int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad, int64_t P) {
    g.N = (int)N; g.K = (int)K; g.H = (int)H; g.W = (int)W; g.pad = (int)pad; g.T = 2 * (int)pad + 1;
    g.Mdim = g.T * g.K; g.P = (int)P; g.S = 1; g.hs = 0;
    g.tilesM = (g.Mdim + 15) / 16;
    const int cap = g.tilesM <= 4 ? 4 : 9;
    g.sb = (g.tilesM + cap - 1) / cap;
    g.tps = (g.tilesM + g.sb - 1) / g.sb;
    static const int cand[][2] = {{8, 64}, {4, 64}, {4, 32}, {4, 16}};
    bool ok = false;
    for (auto& c : cand) {
        g.RB = c[0]; g.WB = c[1]; g.RW = g.RB / 4;
        g.WX = g.WB + 2 * g.pad; g.RBY = g.RB + 2 * g.pad;
        g.planeX = (g.RB * g.WX) | 1; g.planeY = (g.RBY * g.WB) | 1;
        size_t tiles = (size_t)g.K * (g.planeX + g.planeY) * 4, dred = (size_t)(cap * 16) * (cap * 16) * 4;
        if (tiles <= (size_t)kLdsBudget && dred <= (size_t)kLdsBudget) { ok = true; break; }
    }
    int slots = g.P * g.sb * g.sb;
    g.G = 256 / slots;
    if (g.G < 1) g.G = 1;
    return ok;
}
static inline size_t joint_lds_bytes(const JointGeom& g) {
    int cap = g.tilesM <= 4 ? 4 : 9;
    size_t tiles = (size_t)g.K * (g.planeX + g.planeY) * 4, dred = (size_t)(cap * 16) * (cap * 16) * 4;
    return tiles > dred ? tiles : dred;
}


// bf16 / f16 matrix-core paths of the joint (mi_local_fwd_px.hip) and of its backward (mi_local_bwd_rows.hip, mi_local_bwd_f8.hip).
// nterms: 1 = plain bf16 operands, 3 = bf16 hi/lo split (three products), 2 = f16 hi x hi + fp8 cross terms (backward, pad 3)
bool joint_fwd_bf16_supported(const JointGeom& g);
int launch_joint_fwd_px(hipStream_t st, const float* x, const float* y, const JointGeom& g, const int32_t* win, float* partials, int nterms,
                        unsigned char* planes = nullptr);
bool local_bwd_bf16_supported(int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad);
size_t local_bwd_bf16_ws_bytes(int64_t K, int64_t pad, int64_t P);
bool local_bwd_f8_supported(int64_t K, int64_t pad);
typedef _Float16 qh8_t __attribute__((ext_vector_type(8)));
typedef _Float16 qh2_t __attribute__((ext_vector_type(2)));
typedef int qi8_t __attribute__((ext_vector_type(8)));
typedef unsigned int qu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int qu32x2 __attribute__((ext_vector_type(2)));


#ifdef __HIPCC__
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// Four probabilities -> their operand images: f16 hi (4 x 2 B), e4m3(2^20 (v - hi)) and e4m3(2^8 v) (4 x 1 B each).  ONE definition for
// the in-kernel split, the stand-alone plane writer below and the joint forward's by-product, so the three agree bit for bit.
__device__ __forceinline__ void split_quad(const float* v, qu32x2& h16, unsigned& l8, unsigned& h8) {
    const qh2_t h01 = {(_Float16)v[0], (_Float16)v[1]}, h23 = {(_Float16)v[2], (_Float16)v[3]};
    h16 = qu32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    l8 = pack_fp8x4((v[0] - (float)h01[0]) * 1048576.f, (v[1] - (float)h01[1]) * 1048576.f, (v[2] - (float)h23[0]) * 1048576.f,
                    (v[3] - (float)h23[1]) * 1048576.f);
    h8 = pack_fp8x4(v[0] * 256.f, v[1] * 256.f, v[2] * 256.f, v[3] * 256.f);
}

#endif

// Operand planes of the f16 + fp8 backward (K = 20), one buffer: [slack][p16: maps x HW x 40 B][slack][p8l: maps x HW x 24 B][slack][p8h: same]
// [slack] -- the kernel copies whole 70-pixel row segments, which start up to 3 pixels before a row and end up to 69 after it.
struct MiPlanes {
    unsigned char *p16, *p8l, *p8h;
};
constexpr int64_t kPlaneSlack = 4096;
static inline int64_t mi_planes_bytes(int64_t maps, int64_t HW) { return maps * HW * 88 + 4 * kPlaneSlack; }
static inline MiPlanes mi_planes(unsigned char* base, int64_t maps, int64_t HW) {
    if (!base) return MiPlanes{nullptr, nullptr, nullptr};
    unsigned char* p16 = base + kPlaneSlack;
    unsigned char* p8l = p16 + maps * HW * 40 + kPlaneSlack;
    return MiPlanes{p16, p8l, p8l + maps * HW * 24 + kPlaneSlack};
}
int launch_make_planes(hipStream_t st, const float* probs, int64_t maps, int64_t HW, unsigned char* planes);
size_t local_bwd_f8_ws_bytes(int64_t K, int64_t pad, int64_t P);
int launch_local_bwd_f8(hipStream_t st, const float* x, const float* y, int64_t S, int64_t hs, int64_t N, int64_t K, int64_t H, int64_t W,
                        int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale, float* gx, float* gy,
                        int accumulate, void* ws, const unsigned char* planes);

int launch_local_bwd_rows(hipStream_t st, const float* x, const float* y, int64_t S, int64_t hs, int64_t N, int64_t K, int64_t H, int64_t W,
                          int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale, float* gx, float* gy,
                          int accumulate, void* ws, int nterms, const unsigned char* planes = nullptr);

}  // namespace miseg
