// Input pipeline on the device: the ACDC slices live in HBM as u8 atlases; one launch turns a list of
// augmentation jobs into the fp32 image batch and the int64 label batch the epochers consume.
//
// Replaces, per slice, the reference's PIL chain (semi_seg/augment.py:7-52 through contrastyou/augment/sequential_wrapper.py
// and whl:deepclustering2/augment/pil_augment.py): RandomRotation (PIL Image.rotate, NEAREST, zero fill -> libImaging
// affine_fixed, 16.16 fixed point), RandomVertical/HorizontalFlip, RandomCrop / CenterCrop, torchvision-0.7 ColorJitter on
// an "L" image (ImageEnhance.Brightness / Contrast via Image.blend; Color is the identity on one band), ToTensor (u8/255)
// and ToLabel (raw value -> int64).  All arithmetic that decides a pixel is integer or a single fp32 multiply-add pair
// executed exactly as libImaging's C does, so the result is bit-identical to the PIL chain given the same parameters; the
// parameters (angle -> fixed-point matrix, flips, crop origin, colour factors and their order) are drawn on the host from
// the reference's own random stream (miseg_amd/slices.py).
//
// One block per job; a 224x224 crop is 49 pixels per thread held in registers between the gather, the block-wide integer
// mean the contrast step needs, and the store.  HBM-bound byte gather; a batch is ~100 KB in, ~10 MB out.
#include "common.h"

namespace miseg {

constexpr int kAugThreads = 1024;
constexpr int kAugMaxPerThread = 64;  // output crops up to 256 x 256

__device__ __forceinline__ unsigned blend_u8(unsigned base, unsigned v, float alpha, bool interpolate) {
    // libImaging Blend.c: (int)in1 + alpha * ((int)in2 - (int)in1) in fp32, truncated; clipped when extrapolating.
    const float t = (float)(int)base + alpha * (float)((int)v - (int)base);
    if (interpolate) return (unsigned)(int)t & 0xFFu;
    if (t <= 0.0f) return 0u;
    if (t >= 255.0f) return 255u;
    return (unsigned)(int)t;
}

__global__ __launch_bounds__(kAugThreads) void augment_slices_kernel(const uint8_t* __restrict__ atlas_img,
                                                                     const uint8_t* __restrict__ atlas_gt, int n_slices, int slice_h,
                                                                     int row_pitch, const int* __restrict__ jobs, int out_h,
                                                                     int out_w, float* __restrict__ img_out,
                                                                     long long* __restrict__ gt_out) {
    __shared__ int job[MISEG_AUG_JOB_INTS];
    __shared__ unsigned long long red[kAugThreads / 64 + 1];
    const int tid = threadIdx.x;
    if (tid < MISEG_AUG_JOB_INTS) job[tid] = jobs[(long long)blockIdx.x * MISEG_AUG_JOB_INTS + tid];
    __syncthreads();
    // a malformed job must not turn into an out-of-bounds read: clamp the counts, bound the final coordinates
    const int slice = job[0], n_geo = min(max(job[3], 0), MISEG_AUG_MAX_GEO), n_col = min(max(job[4], 0), 3);
    const bool slice_ok = slice >= 0 && slice < n_slices;
    const long long slice_pitch = (long long)slice_h * row_pitch;
    const uint8_t* src_img = atlas_img + (long long)slice * slice_pitch;
    const uint8_t* src_gt = atlas_gt ? atlas_gt + (long long)slice * slice_pitch : nullptr;
    const int npix = out_h * out_w;
    float* oimg = img_out + (long long)blockIdx.x * npix;
    long long* ogt = gt_out ? gt_out + (long long)blockIdx.x * npix : nullptr;

    unsigned packed[kAugMaxPerThread / 4];
#pragma unroll
    for (int i = 0; i < kAugMaxPerThread / 4; ++i) packed[i] = 0u;

    // ---- geometry: walk the op chain from the output back to the stored slice
#pragma unroll
    for (int k = 0; k < kAugMaxPerThread; ++k) {
        const int p = tid + k * kAugThreads;
        if (p < npix) {
            int y = p / out_w, x = p - y * out_w;
            bool ok = true;
#pragma unroll 1
            for (int g = n_geo - 1; g >= 0; --g) {
                const int* op = job + 12 + g * 9;
                const int iw = op[7], ih = op[8];  // size of this op's input image
                switch (op[0]) {
                    case MISEG_AUG_CROP:  // out(y,x) = in(y+i, x+j); PIL crop reads zeros outside
                        y += op[1];
                        x += op[2];
                        ok = ok && y >= 0 && y < ih && x >= 0 && x < iw;
                        break;
                    case MISEG_AUG_VFLIP: y = ih - 1 - y; break;
                    case MISEG_AUG_HFLIP: x = iw - 1 - x; break;
                    case MISEG_AUG_AFFINE: {  // libImaging affine_fixed: xx = a2 + y*a1 + x*a0 (int32), xin = xx >> 16
                        const int xx = op[3] + y * op[2] + x * op[1];
                        const int yy = op[6] + y * op[5] + x * op[4];
                        x = xx >> 16;
                        y = yy >> 16;
                        ok = ok && x >= 0 && x < iw && y >= 0 && y < ih;
                        break;
                    }
                    default: break;
                }
                if (!ok) break;
            }
            unsigned v = 0u, lab = 0u;
            if (ok && slice_ok && y >= 0 && y < slice_h && x >= 0 && x < row_pitch) {
                const long long o = (long long)y * row_pitch + x;
                v = src_img[o];
                if (src_gt) lab = src_gt[o];
            }
            packed[k >> 2] |= v << ((k & 3) * 8);
            if (ogt) ogt[p] = (long long)lab;
        }
    }

    // ---- colour ops in their drawn order
    for (int c = 0; c < n_col; ++c) {
        const int code = job[5 + c];
        const float alpha = __int_as_float(job[8 + c]);
        const bool interp = alpha >= 0.0f && alpha <= 1.0f;
        unsigned base = 0u;
        if (code == MISEG_AUG_CONTRAST) {  // degenerate image = int(mean + 0.5) of the image as it is now
            unsigned long long s = 0;
#pragma unroll
            for (int k = 0; k < kAugMaxPerThread; ++k)
                if (tid + k * kAugThreads < npix) s += (packed[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) {
                unsigned long long t = 0;
                for (int i = 0; i < kAugThreads / 64; ++i) t += red[i];
                red[kAugThreads / 64] = t;
            }
            __syncthreads();
            const double mean = (double)red[kAugThreads / 64] / (double)npix;
            base = (unsigned)(int)(mean + 0.5);
        } else if (code != MISEG_AUG_BRIGHTNESS) {
            continue;  // saturation: ImageEnhance.Color blends an "L" image with itself
        }
#pragma unroll
        for (int k = 0; k < kAugMaxPerThread; ++k) {
            const int sh = (k & 3) * 8;
            const unsigned v = (packed[k >> 2] >> sh) & 0xFFu;
            const unsigned r = blend_u8(base, v, alpha, interp);
            packed[k >> 2] = (packed[k >> 2] & ~(0xFFu << sh)) | (r << sh);
        }
    }

    // ---- ToTensor
#pragma unroll
    for (int k = 0; k < kAugMaxPerThread; ++k) {
        const int p = tid + k * kAugThreads;
        if (p < npix) oimg[p] = (float)((packed[k >> 2] >> ((k & 3) * 8)) & 0xFFu) / 255.0f;
    }
}

}  // namespace miseg

using namespace miseg;

extern "C" int miseg_augment_slices(void* stream, const uint8_t* atlas_img, const uint8_t* atlas_gt, int64_t n_slices,
                                    int64_t slice_h, int64_t slice_w, const int32_t* jobs_dev, int64_t njobs, int64_t out_h,
                                    int64_t out_w, float* img_out, int64_t* gt_out) {
    MISEG_TAPE(miseg_augment_slices, stream, atlas_img, atlas_gt, n_slices, slice_h, slice_w, jobs_dev, njobs, out_h, out_w, img_out, gt_out);
    MISEG_REQUIRE(atlas_img && jobs_dev && img_out, "augment_slices: null pointer");
    MISEG_REQUIRE((atlas_gt == nullptr) == (gt_out == nullptr), "augment_slices: label atlas and label output go together");
    MISEG_REQUIRE(n_slices > 0 && slice_h > 0 && slice_w > 0 && slice_h < 32768 && slice_w < 32768, "augment_slices: bad atlas shape");
    MISEG_REQUIRE(out_h > 0 && out_w > 0 && out_h * out_w <= (int64_t)kAugThreads * kAugMaxPerThread,
                  "augment_slices: output %lld x %lld exceeds %d pixels", (long long)out_h, (long long)out_w, kAugThreads * kAugMaxPerThread);
    MISEG_REQUIRE(njobs >= 0 && njobs < (1 << 20), "augment_slices: bad job count");
    if (njobs == 0) return MISEG_OK;
    hipLaunchKernelGGL(augment_slices_kernel, dim3((unsigned)njobs), dim3(kAugThreads), 0, as_stream(stream), atlas_img, atlas_gt,
                       (int)n_slices, (int)slice_h, (int)slice_w, jobs_dev, (int)out_h, (int)out_w, img_out,
                       reinterpret_cast<long long*>(gt_out));
    MISEG_LAUNCH_CHECK("augment_slices");
    return MISEG_OK;
}
