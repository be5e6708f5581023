// Sustained dense bf16 MFMA rate of the whole chip (256 CUs), register-resident operands, no memory traffic:
// what v_mfma_f32_16x16x32_bf16 delivers under the board's power management, for constant and for random operand bits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, const u32x4* seed, int iters) {
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{0, 0, 0, 0};
    bf16x8 x[4], y[4];
    for (int i = 0; i < 4; ++i) {
        x[i] = __builtin_bit_cast(bf16x8, seed[(threadIdx.x * 8 + i) & 4095]);
        y[i] = __builtin_bit_cast(bf16x8, seed[(threadIdx.x * 8 + 4 + i) & 4095]);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[j & 3], y[(j >> 1) & 3], a[j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    float* out; u32x4* seed; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&seed, 4096 * 16);
    unsigned* h = (unsigned*)malloc(4096 * 16);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (int i = 0; i < 4096 * 4; ++i) {
            // mode 0: every element the bf16 value 1.0 (0x3F80); mode 1: random sign/mantissa, exponent near 1 (finite sums)
            unsigned lo = mode ? (0x3F00u | (rand() & 0x80FFu)) : 0x3F80u, hi = mode ? (0x3F00u | (rand() & 0x80FFu)) : 0x3F80u;
            h[i] = lo | (hi << 16);
        }
        hipMemcpy(seed, h, 4096 * 16, hipMemcpyHostToDevice);
        const int iters = 20000;     // 8 MFMAs per iteration
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, seed, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 20.0 * blocks * 4 * (double)iters * 8 * 16 * 16 * 32 * 2;
            printf("%s operands, %d blocks x 4 waves, pass %d: %.1f ms -> %.0f TFLOP/s\n", mode ? "random" : "constant", blocks, rep, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
