// Does it matter which register file holds the MFMA accumulator?  The same f16 16x16x32 loop with the C/D operand forced into
// ArchVGPRs ("v" constraint) or AccVGPRs ("a"), 8 independent accumulators, operands constant across the loop; 1 and 2 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 mfma_acc_file.hip -o mfma_acc_file && ./mfma_acc_file
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int FILE_, int SCALED>
__global__ __launch_bounds__(256) void k(float* out, const u32x4* seed, int iters) {
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{0, 0, 0, 0};
    f16x8 x[4];
    for (int i = 0; i < 4; ++i) {
        u32x4 u = seed[(threadIdx.x * 4 + i) & 4095];
        for (int j = 0; j < 4; ++j) u[j] = (u[j] & 0x83FF83FFu) | 0x38003800u;
        x[i] = __builtin_bit_cast(f16x8, u);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (FILE_ == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x[j & 3]), "v"(x[(j >> 1) & 3]));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(a[j]) : "v"(x[j & 3]), "v"(x[(j >> 1) & 3]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; u32x4* seed; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&seed, 4096 * 16); hipMemset(seed, 0x5A, 4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks = 256; blocks <= 512; blocks *= 2)
        for (int f = 0; f < 2; ++f) {
            const int iters = 20000;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int l = 0; l < 10; ++l) {
                    if (f == 0) hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256), 0, 0, out, seed, iters);
                    else hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), 0, 0, out, seed, iters);
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double flop = 10.0 * blocks * 4 * (double)iters * 8 * 16384;
            printf("%d waves/SIMD, accumulators in %s: %.0f TFLOP/s\n", blocks / 256, f ? "AccVGPRs" : "ArchVGPRs", flop / ms / 1e9);
        }
    return 0;
}
