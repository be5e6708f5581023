// Operand-layout probe for the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4), the 8-bit transposed LDS read
// (ds_read_b64_tr_b8) and the f32 -> fp8 conversion, on exact small-integer data.  This image has no ISA manual for them, so the
// layouts the local-MI kernels rely on are MEASURED here; the output is committed next to this file.
// hipcc -O3 --offload-arch=gfx950 mx_layout_probe.hip -o mx_layout_probe && ./mx_layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// a[64][32] / b[64][32] bytes per lane; d[64][4] floats per lane; scales per lane
__global__ void mfma_probe(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, float* d, int opsel_a, int opsel_b) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = *reinterpret_cast<const int*>(a + l * 32 + j * 4);
        bv[j] = *reinterpret_cast<const int*>(b + l * 32 + j * 4);
    }
    f32x4 c{0, 0, 0, 0};
    f32x4 r;
    if (opsel_a == 0 && opsel_b == 0) r = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
    else if (opsel_a == 1 && opsel_b == 0) r = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 1, sa[l], 0, sb[l]);
    else if (opsel_a == 0 && opsel_b == 2) r = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 2, sb[l]);
    else r = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 3, sa[l], 3, sb[l]);
    for (int j = 0; j < 4; ++j) d[l * 4 + j] = r[j];
}

__global__ void tr8_probe(const unsigned char* img, int stride, unsigned char* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = img[i];
    __syncthreads();
    typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
    const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lds + threadIdx.x * stride));
    *reinterpret_cast<i32x2*>(out + threadIdx.x * 8) = v;
}

__global__ void cvt_probe(const float* in, int n, unsigned* out) {
    const int i = threadIdx.x;
    if (i * 2 + 1 < n + 1) {
        const float x = in[i * 2], y = i * 2 + 1 < n ? in[i * 2 + 1] : 0.f;
        out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x, y, 0, false);
    }
}

static float e4m3_to_f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 0) r = m / 8.0f * (1.0f / 64);
    else if (e == 15 && m == 7) r = 0.f / 0.f;
    else { r = (1 + m / 8.0f); int ee = e - 7; while (ee > 0) { r *= 2; --ee; } while (ee < 0) { r *= 0.5f; ++ee; } }
    return s ? -r : r;
}
static unsigned char f_to_e4m3_small_int(int v) {      // exact for |v| <= 16
    for (int b = 0; b < 256; ++b) if (e4m3_to_f((unsigned char)b) == (float)v && !(b == 0x80)) return (unsigned char)b;
    return 0;
}

int main() {
    unsigned char *da, *db, *dimg, *dout; int *dsa, *dsb; float* dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
    hipMalloc(&dimg, 4096); hipMalloc(&dout, 512);
    unsigned char ha[2048], hb[2048]; int hsa[64], hsb[64]; float hd[256];
    const unsigned char one = 0x38;
    for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 0x7F7F7F7F;
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    auto run = [&](int oa, int ob) {
        hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, oa, ob);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    };
    // D layout assumed (dtype-independent per the guide): lane l, reg r -> row 4*(l>>4)+r, col l&15
    auto D = [&](int i, int j) { return hd[(j + 16 * (i >> 2)) * 4 + (i & 3)]; };

    // ---- 1. rows of A: A one-hot at (lane, byte), B all ones -> which row i lights up
    printf("== A operand: (lane, byte) -> row i  [B = all ones]\n");
    int rowA[64][32], colB[64][32], kA[64][32], kB[64][32];
    memset(hb, one, 2048);
    for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) {
        memset(ha, 0, 2048); ha[l * 32 + by] = one; run(0, 0);
        int row = -1, cnt = 0;
        for (int i = 0; i < 16; ++i) { bool all = true; for (int j = 0; j < 16; ++j) all = all && D(i, j) == 1.f; if (all) { row = i; ++cnt; } }
        rowA[l][by] = cnt == 1 ? row : -1;
    }
    bool okA = true;
    for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) okA = okA && rowA[l][by] == (l & 15);
    printf("row(lane, byte) == lane & 15 for every byte: %s\n", okA ? "YES" : "NO");
    if (!okA) for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int by = 0; by < 32; ++by) printf(" %2d", rowA[l][by]); printf("\n"); }
    // ---- 2. cols of B
    memset(ha, one, 2048);
    for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) {
        memset(hb, 0, 2048); hb[l * 32 + by] = one; run(0, 0);
        int col = -1, cnt = 0;
        for (int j = 0; j < 16; ++j) { bool all = true; for (int i = 0; i < 16; ++i) all = all && D(i, j) == 1.f; if (all) { col = j; ++cnt; } }
        colB[l][by] = cnt == 1 ? col : -1;
    }
    bool okB = true;
    for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) okB = okB && colB[l][by] == (l & 15);
    printf("== B operand: col(lane, byte) == lane & 15 for every byte: %s\n", okB ? "YES" : "NO");
    if (!okB) for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int by = 0; by < 32; ++by) printf(" %2d", colB[l][by]); printf("\n"); }
    // ---- 3. k of A relative to the labelling k_B(lane, byte) = 32*(lane>>4) + byte: B carries (k%16)+1 then (k/16)+1
    for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) kA[l][by] = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) {
            const int k = 32 * (l >> 4) + by;
            hb[l * 32 + by] = f_to_e4m3_small_int(pass ? k / 16 + 1 : k % 16 + 1);
        }
        for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) {
            memset(ha, 0, 2048); ha[l * 32 + by] = one; run(0, 0);
            const int i = l & 15;
            const float v = D(i, 0);
            bool cst = true; for (int j = 0; j < 16; ++j) cst = cst && D(i, j) == v;
            if (!cst) kA[l][by] = -100000;
            else kA[l][by] += pass ? ((int)v - 1) * 16 : (int)v - 1;
        }
    }
    bool okK = true;
    for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) okK = okK && kA[l][by] == 32 * (l >> 4) + by;
    printf("== k: A(lane, byte) pairs with B(lane', byte') iff 32*(lane>>4)+byte equal on both sides: %s\n", okK ? "YES" : "NO");
    if (!okK) for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int by = 0; by < 32; ++by) printf(" %4d", kA[l][by]); printf("\n"); }
    (void)kB;
    // ---- 4. full random small-integer GEMM under that layout
    {
        srand(3);
        int A[16][128], B[128][16];
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i][k] = rand() % 9 - 4;
        for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k][j] = rand() % 7 - 3;
        for (int l = 0; l < 64; ++l) for (int by = 0; by < 32; ++by) {
            ha[l * 32 + by] = f_to_e4m3_small_int(A[l & 15][32 * (l >> 4) + by]);
            hb[l * 32 + by] = f_to_e4m3_small_int(B[32 * (l >> 4) + by][l & 15]);
        }
        run(0, 0);
        int bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 0; for (int k = 0; k < 128; ++k) r += A[i][k] * B[k][j]; bad += D(i, j) != (float)r; }
        printf("== random integer 16x16x128 GEMM with A[i=l&15][k=32(l>>4)+byte], B[k][j=l&15], D[row=4(l>>4)+r][col=l&15]: %d mismatches\n", bad);
        // ---- 5. scales: E8M0 byte per lane, opsel picks the byte; which lanes' scale applies to which 32-element k block?
        // give lane l scale byte0 = 127 + (l>>4) for A  => block kb of row i scaled by 2^(kb) if scale(lane) applies to that lane's own 32 k's
        for (int l = 0; l < 64; ++l) { hsa[l] = (127 + (l >> 4)) | (120 << 8) | (121 << 16) | (122 << 24); hsb[l] = 0x7F7F7F7F; }
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        run(0, 0);
        bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 0; for (int k = 0; k < 128; ++k) r += (A[i][k] * B[k][j]) << (k / 32); bad += D(i, j) != (float)r; }
        printf("== A scale byte0 of lane l = 2^(l>>4) applies to that lane's own 32 k values: %d mismatches\n", bad);
        // per-row scale variation: lane l scale = 2^(l&3)
        for (int l = 0; l < 64; ++l) hsa[l] = 127 + (l & 3);
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice);
        run(0, 0);
        bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 0; for (int k = 0; k < 128; ++k) r += (A[i][k] * B[k][j]) << (i & 3); bad += D(i, j) != (float)r; }
        printf("== A scale of lane l = 2^(l&3) applies per row i = l&15: %d mismatches\n", bad);
        // opsel: byte 1 of A's scale register = 120 -> 2^-7; byte 2 of B's = 0x7F when opsel 2
        for (int l = 0; l < 64; ++l) { hsa[l] = 127 | (120 << 8) | (121 << 16) | (122 << 24); hsb[l] = 127 | (126 << 8) | (129 << 16) | (125 << 24); }
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        run(1, 0);
        { int r = 0; for (int k = 0; k < 128; ++k) r += A[3][k] * B[k][5]; printf("== opsel_a=1 (byte1 = 2^-7): D/ref = %g (expect %g)\n", D(3, 5) / r, 1.0 / 128); }
        run(0, 2);
        { int r = 0; for (int k = 0; k < 128; ++k) r += A[3][k] * B[k][5]; printf("== opsel_b=2 (byte2 = 2^2): D/ref = %g (expect 4)\n", D(3, 5) / r); }
        run(3, 3);
        { int r = 0; for (int k = 0; k < 128; ++k) r += A[3][k] * B[k][5]; printf("== opsel 3/3 (2^-5 * 2^-2): D/ref = %g (expect %g)\n", D(3, 5) / r, 1.0 / 128); }
    }
    // ---- 6. ds_read_b64_tr_b8: image byte a holds (a & 255) then (a >> 8); lane address = lane * stride
    for (int stride = 8; stride <= 16; stride += 8) {
        unsigned char img[4096], o0[512], o1[512];
        for (int a = 0; a < 4096; ++a) img[a] = a & 255;
        hipMemcpy(dimg, img, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr8_probe, dim3(1), dim3(64), 0, 0, dimg, stride, dout);
        hipMemcpy(o0, dout, 512, hipMemcpyDeviceToHost);
        for (int a = 0; a < 4096; ++a) img[a] = a >> 8;
        hipMemcpy(dimg, img, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr8_probe, dim3(1), dim3(64), 0, 0, dimg, stride, dout);
        hipMemcpy(o1, dout, 512, hipMemcpyDeviceToHost);
        printf("== ds_read_b64_tr_b8, lane address = %d * lane: result byte b of lane t comes from LDS byte address ... as (source lane, source byte)\n", stride);
        for (int t = 0; t < 64; ++t) {
            printf("lane %2d:", t);
            for (int b = 0; b < 8; ++b) { const int a = o0[t * 8 + b] | (o1[t * 8 + b] << 8); printf(" (%2d,%d)", a / stride, a % stride); }
            printf("\n");
            if (t == 17) { printf("  ... (lanes 18-63 checked against the 16-lane-group pattern below)\n"); break; }
        }
        int bad = 0;
        for (int t = 0; t < 64; ++t) for (int b = 0; b < 8; ++b) {
            const int a = o0[t * 8 + b] | (o1[t * 8 + b] << 8);
            const int t0 = t & 15, a0 = o0[t0 * 8 + b] | (o1[t0 * 8 + b] << 8);
            bad += a != a0 + (t >> 4) * 16 * stride;
        }
        printf("16-lane groups repeat the pattern of lanes 0-15 shifted by 16 lanes: %s\n", bad ? "NO" : "YES");
    }
    // ---- 7. v_cvt_pk_fp8_f32
    {
        const float vals[] = {0.f, 1.f, 0.5f, 1.0625f, 1.125f, 1.1875f, 448.f, 480.f, 1000.f, -3.f, 0.015625f, 0.001953125f, 0.0009765625f, 0.0029296875f, 1e-4f, 17.f, 18.f, 19.f, 0.3f, -0.3f};
        const int n = sizeof(vals) / sizeof(float);
        float* din; unsigned* dcv; unsigned hcv[32];
        hipMalloc(&din, 256); hipMalloc(&dcv, 256);
        hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(16), 0, 0, din, n, dcv);
        hipMemcpy(hcv, dcv, 64, hipMemcpyDeviceToHost);
        printf("== v_cvt_pk_fp8_f32 (OCP e4m3): value -> byte -> value\n");
        for (int i = 0; i < n; ++i) { const unsigned char by = (hcv[i / 2] >> (8 * (i & 1))) & 255; printf("  %12.9g -> 0x%02X -> %g\n", vals[i], by, e4m3_to_f(by)); }
    }
    return 0;
}
