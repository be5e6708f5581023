// Chip-wide matrix-core rate per MFMA data type on random operand bits, register-resident operands, no memory traffic --
// continuous (back-to-back launches, ~0.3 s) and in BURST form (one ~1 ms launch every 7 ms, the duty cycle of the local-MI kernels
// inside the train step).  Per-clock rate is not per-watt rate: the board holds its clock down under matrix load, so what decides
// between operand splits (bf16x3 vs f16 + fp8 cross terms ...) is the wall-clock rate of each instruction mix, printed here as
//   * TOP/s of the instruction itself (2*M*N*K per instruction), the in-kernel shader clock (s_memtime / s_memrealtime) and the
//     cycles per instruction that follow from the two;
//   * for the candidate splits: ns per "fp32-class 16x16x64 product block" on one SIMD.
// hipcc -O3 --offload-arch=gfx950 mfma_dtypes.hip -o mfma_dtypes && ./mfma_dtypes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum Kind { BF16 = 0, F16, FP8, SF8, SF6, SF4, I8, BF16_32, MIX_BF16X3, MIX_F16_SF8, MIX_F16_SF6, MIX_F16_I8, NKIND };
static const char* kname[NKIND] = {"bf16 16x16x32", "f16 16x16x32", "fp8 16x16x32 (non-scaled)", "scaled e4m3 16x16x128", "scaled e2m3 (fp6) 16x16x128",
                                   "scaled e2m1 (fp4) 16x16x128", "i8 16x16x64", "bf16 32x32x16", "mix: 6 x bf16 (bf16x3 per 64 k)",
                                   "mix: 2 x f16 + 1 x scaled e4m3 (per 64 k)", "mix: 4 x f16 + 1 x scaled fp6 (per 128 k)", "mix: 2 x f16 + 2 x i8 (per 64 k)"};
// operations (2*M*N*K) per loop iteration of the kernel below
static const double kops[NKIND] = {8 * 16384.0, 8 * 16384.0, 8 * 16384.0, 8 * 65536.0, 8 * 65536.0, 8 * 65536.0, 8 * 32768.0, 4 * 32768.0,
                                   12 * 16384.0, 8 * 16384.0 + 4 * 65536.0, 8 * 16384.0 + 2 * 65536.0, 8 * 16384.0 + 8 * 32768.0};
// fp32-class 16x16x64 product blocks per iteration (mixes only)
static const double kblocks[NKIND] = {0, 0, 0, 0, 0, 0, 0, 0, 2, 4, 4, 4};

// One kernel per instruction mix.  Every mix keeps its accumulators in its own registers and its MFMAs in inline asm ("+a": the
// accumulator file) -- written with the builtins on shared arrays, hipcc shuffled accumulators between the two register files inside
// the single-type loops (v_accvgpr_read / _write per iteration), which is what round 2's 1.25 PFLOP/s "sustained bf16 rate" measured.
#define MF(op, acc, x, y) asm volatile(op " %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y))
typedef int i32x6 __attribute__((ext_vector_type(6)));
#define MS(fmt, acc, x, y) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" fmt : "+a"(acc) : "v"(x), "v"(y), "v"(sc))
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const u32x4* seed, int iters, unsigned long long* clk) {
    f32x4 a[8];
    i32x4 ai[8];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 aw[2];
    for (int i = 0; i < 8; ++i) { a[i] = f32x4{0, 0, 0, 0}; ai[i] = i32x4{0, 0, 0, 0}; }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) aw[i][j] = 0.f;
    u32x4 r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed[(threadIdx.x * 8 + i) & 4095];
    // bf16 / f16 fragments: exponent forced near 1 (finite sums), random sign + mantissa
    bf16x8 xb[4]; f16x8 xh[4];
    for (int i = 0; i < 4; ++i) {
        u32x4 t = r[i];
        for (int j = 0; j < 4; ++j) t[j] = (t[j] & 0x80FF80FFu) | 0x3F003F00u;
        xb[i] = __builtin_bit_cast(bf16x8, t);
        u32x4 u = r[i];
        for (int j = 0; j < 4; ++j) u[j] = (u[j] & 0x83FF83FFu) | 0x38003800u;
        xh[i] = __builtin_bit_cast(f16x8, u);
    }
    // 8-bit operands: random bytes with the top exponent bit cleared (no NaN / inf in e4m3 / e5m2; small magnitudes); i8 as is
    i32x8 x8[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) x8[i][j] = (int)(r[4 + i * 2 + (j >> 2)][j & 3] & 0xBFBFBFBFu);
    long l8[2] = {(long)(unsigned)x8[0][0] | ((long)x8[0][1] << 32), (long)(unsigned)x8[1][0] | ((long)x8[1][1] << 32)};
    i32x4 q8[2] = {i32x4{x8[0][0], x8[0][1], x8[0][2], x8[0][3]}, i32x4{x8[1][0], x8[1][1], x8[1][2], x8[1][3]}};
    const int sc = 0x7F7F7F7F;      // E8M0 1.0
    // fp6 operands are 6 registers (32 x 6 bit), fp4 operands 4
    i32x6 x6[2] = {i32x6{x8[0][0], x8[0][1], x8[0][2], x8[0][3], x8[0][4], x8[0][5]}, i32x6{x8[1][0], x8[1][1], x8[1][2], x8[1][3], x8[1][4], x8[1][5]}};
    unsigned long long t0 = 0, r0 = 0;
    if (clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (KIND == BF16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_f32_16x16x32_bf16", a[j], xb[j & 3], xb[(j >> 1) & 3]);
        } else if (KIND == F16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_f32_16x16x32_f16", a[j], xh[j & 3], xh[(j >> 1) & 3]);
        } else if (KIND == FP8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_f32_16x16x32_fp8_fp8", a[j], l8[j & 1], l8[(j >> 1) & 1]);
        } else if (KIND == SF8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MS("", a[j], x8[j & 1], x8[(j >> 1) & 1]);
        } else if (KIND == SF6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MS(" cbsz:2 blgp:2", a[j], x6[j & 1], x6[(j >> 1) & 1]);
        } else if (KIND == SF4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MS(" cbsz:4 blgp:4", a[j], q8[j & 1], q8[(j >> 1) & 1]);
        } else if (KIND == I8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_i32_16x16x64_i8", ai[j], q8[j & 1], q8[(j >> 1) & 1]);
        } else if (KIND == BF16_32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) MF("v_mfma_f32_32x32x16_bf16", aw[j & 1], xb[j & 3], xb[(j + 1) & 3]);
        } else if (KIND == MIX_BF16X3) {
            // two 16x16x64 product blocks: 6 MFMAs each on one accumulator pair (4 accumulators, as the kernels interleave them)
#pragma unroll
            for (int j = 0; j < 12; ++j) MF("v_mfma_f32_16x16x32_bf16", a[j & 3], xb[j & 3], xb[(j >> 1) & 3]);
        } else if (KIND == MIX_F16_SF8) {
            // four product blocks: 2 f16 (hi*hi over 64 k) + 1 scaled fp8 over 128 k = both cross terms K-concatenated
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                MF("v_mfma_f32_16x16x32_f16", a[j], xh[j & 3], xh[(j >> 1) & 3]);
                MF("v_mfma_f32_16x16x32_f16", a[j], xh[(j + 1) & 3], xh[(j >> 1) & 3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) MS("", a[j], x8[j & 1], x8[(j >> 1) & 1]);
        } else if (KIND == MIX_F16_SF6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_f32_16x16x32_f16", a[j & 3], xh[j & 3], xh[(j >> 1) & 3]);
#pragma unroll
            for (int j = 0; j < 2; ++j) MS(" cbsz:2 blgp:2", a[j], x6[j & 1], x6[(j >> 1) & 1]);
        } else if (KIND == MIX_F16_I8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_f32_16x16x32_f16", a[j & 3], xh[j & 3], xh[(j >> 1) & 3]);
#pragma unroll
            for (int j = 0; j < 8; ++j) MF("v_mfma_i32_16x16x64_i8", ai[j & 3], q8[j & 1], q8[(j >> 1) & 1]);
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    if (clk && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][3] + (float)ai[i][0] + (float)ai[i][2];
    s += aw[0][0] + aw[1][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kern_t)(float*, const u32x4*, int, unsigned long long*);
static kern_t kerns[NKIND] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>};

static int cmp_d(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512;           // 512 x 4 waves = 2 waves per SIMD, as the MI kernels run
    float* out; u32x4* seed; unsigned long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&seed, 4096 * 16); hipMalloc(&clk, 4096 * 16);
    unsigned* h = (unsigned*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 4; ++i) h[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
    hipMemcpy(seed, h, 4096 * 16, hipMemcpyHostToDevice);
    unsigned long long* hc = (unsigned long long*)malloc(4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d blocks x 4 waves; T = 1e12 operations/s (2*M*N*K per instruction)\n", blocks);
    printf("%-46s | continuous: T/s  GHz  cyc/instr-block | burst 1 ms / 7 ms: T/s  GHz | ns per fp32-class 16x16x64 block and SIMD (cont / burst)\n", "instruction");
    for (int kind = 0; kind < NKIND; ++kind) {
        // calibrate iterations for ~1 ms
        int iters = 2000;
        float ms = 0;
        for (int pass = 0; pass < 3; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kerns[kind], dim3(blocks), dim3(256), 0, 0, out, seed, iters, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            iters = (int)(iters * 1.0 / ms);
            if (iters < 16) iters = 16;
        }
        const double ops_launch = (double)blocks * 4 * iters * kops[kind];
        // ---- continuous: 300 launches back to back, the last 100 timed; clocks from the last launch
        for (int l = 0; l < 200; ++l) hipLaunchKernelGGL(kerns[kind], dim3(blocks), dim3(256), 0, 0, out, seed, iters, nullptr);
        hipEventRecord(e0);
        for (int l = 0; l < 99; ++l) hipLaunchKernelGGL(kerns[kind], dim3(blocks), dim3(256), 0, 0, out, seed, iters, nullptr);
        hipLaunchKernelGGL(kerns[kind], dim3(blocks), dim3(256), 0, 0, out, seed, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double c_tops = ops_launch * 100 / ms / 1e9;
        hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
        double* gh = (double*)malloc(blocks * sizeof(double));
        double* cy = (double*)malloc(blocks * sizeof(double));
        for (int b = 0; b < blocks; ++b) { gh[b] = (double)hc[b * 2] / (double)hc[b * 2 + 1] * 0.1; cy[b] = (double)hc[b * 2] / iters; }
        qsort(gh, blocks, sizeof(double), cmp_d); qsort(cy, blocks, sizeof(double), cmp_d);
        const double c_ghz = gh[blocks / 2], c_cyc = cy[blocks / 2];
        const double c_ms_launch = ms / 100;
        // ---- burst: one launch, then 6 ms of idle host sleep, 40 times; per-launch event times, median
        double bt[40], bg[40];
        for (int l = 0; l < 40; ++l) {
            usleep(6000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kerns[kind], dim3(blocks), dim3(256), 0, 0, out, seed, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            bt[l] = ms;
            hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
            for (int b = 0; b < blocks; ++b) gh[b] = (double)hc[b * 2] / (double)hc[b * 2 + 1] * 0.1;
            qsort(gh, blocks, sizeof(double), cmp_d);
            bg[l] = gh[blocks / 2];
        }
        qsort(bt, 40, sizeof(double), cmp_d); qsort(bg, 40, sizeof(double), cmp_d);
        const double b_tops = ops_launch / bt[20] / 1e9;
        // ns per product block and SIMD: a SIMD holds blocks*4/1024 waves, each doing kblocks per iteration
        const double waves_per_simd = blocks * 4 / 1024.0;
        char tail[128] = "";
        if (kblocks[kind] > 0)
            snprintf(tail, sizeof tail, "%.1f / %.1f", c_ms_launch * 1e6 / (iters * kblocks[kind] * waves_per_simd), bt[20] * 1e6 / (iters * kblocks[kind] * waves_per_simd));
        printf("%-46s | %7.0f  %.2f  %7.1f | %7.0f  %.2f | %s\n", kname[kind], c_tops, c_ghz, c_cyc, b_tops, bg[20], tail);
        fflush(stdout);
        free(gh); free(cy);
    }
    return 0;
}
