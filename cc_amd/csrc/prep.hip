// Input pipeline, device side: custom_transforms.py's ArrayToTensor + Normalize, with the crop of
// RandomScaleCrop and the mirror of RandomHorizontalFlip folded in -- HWC frames (uint8, or the float32 arrays
// datasets/sequence_folders.py:27-28 produces) -> normalised fp32 NCHW in ONE pass over the batch:
//     dst[n, c, y, x] = ((float)src[n, y + oy_n, xs, c] / 255 - mean_c) / std_c,   xs = flip_n ? W - 1 - (x + ox_n) : x + ox_n
// (the reference's arithmetic order: `.float() / 255`, then `t.sub_(m).div_(s)`; custom_transforms.py:21-30,47-57).
#include "cc_common.h"
#include "../../include/ccengine.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_frames_to_tensor(const T* __restrict__ src, float* __restrict__ dst,
                                                          const int* __restrict__ geo, int H, int W, int h, int w, float m0,
                                                          float m1, float m2, float s0, float s1, float s2) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    const int flip = geo[3 * n], oy = geo[3 * n + 1], ox = geo[3 * n + 2];
    const int ys = y + oy, xc = x + ox;
    const int xs = flip ? (W - 1 - xc) : xc;
    const T* s = src + (((size_t)n * H + ys) * W + xs) * 3;
    float* d = dst + (size_t)n * 3 * h * w + p;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; c++) d[(size_t)c * h * w] = ((float)s[c] / 255.0f - mean[c]) / stdv[c];
}

}  // namespace

extern "C" {

/* src: [N,H,W,3] frames (src_is_u8 ? uint8 : float32 in [0,255]); dst: [N,3,h,w]; geo: int32 [N,3] = (flip, off_y, off_x) per
 * frame with off_y + h <= H, off_x + w <= W (checked by the caller); mean3 / std3 by value. */
int cc_frames_to_tensor(const void* src, int src_is_u8, float* dst, const int* geo, int N, int H, int W, int h, int w, float mean0,
                        float mean1, float mean2, float std0, float std1, float std2, void* stream) {
    if (N <= 0 || h <= 0 || w <= 0 || h > H || w > W) return CC_ERR_ARG;
    dim3 g((unsigned)((h * w + 255) / 256), (unsigned)N);
    if (src_is_u8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frames_to_tensor<unsigned char>), g, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned char*)src, dst, geo, H, W, h, w, mean0, mean1, mean2, std0, std1, std2);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frames_to_tensor<float>), g, dim3(256), 0, (hipStream_t)stream, (const float*)src, dst,
                           geo, H, W, h, w, mean0, mean1, mean2, std0, std1, std2);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
