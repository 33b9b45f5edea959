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

// ---- RandomScaleCrop's resize on the device (custom_transforms.py:93-121): scipy.misc.imresize = byte-scale the float frame to
// its own min..max, then Pillow's 8-bit bilinear resampler (Resample.c): horizontal pass -> uint8 -> vertical pass, 22-bit
// fixed-point coefficients (built on the host, cc_amd/custom_transforms.py resample_table), accumulate from 1 << 21, shift, clip.
// Integer arithmetic throughout: bit-exact with the host path.  geo per frame: (flip, off_y, off_x, sh, sw, htab, vtab, -).
constexpr int GEO = 8;
constexpr int MMB = 64;          // partial min/max blocks per frame
constexpr int PREC = 22;         // Pillow PRECISION_BITS = 32 - 8 - 2

__global__ __launch_bounds__(256) void k_frames_minmax_partial(const float* __restrict__ src, float* __restrict__ ws, long per_frame) {
    const int n = blockIdx.y;
    const float* s = src + (size_t)n * per_frame;
    float lo = INFINITY, hi = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_frame; i += (long)MMB * 256) {
        const float v = s[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    __shared__ float slo[256], shi[256];
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            slo[threadIdx.x] = fminf(slo[threadIdx.x], slo[threadIdx.x + st]);
            shi[threadIdx.x] = fmaxf(shi[threadIdx.x], shi[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ws[((size_t)n * MMB + blockIdx.x) * 2] = slo[0];
        ws[((size_t)n * MMB + blockIdx.x) * 2 + 1] = shi[0];
    }
}

__global__ __launch_bounds__(MMB) void k_frames_minmax_final(const float* __restrict__ ws, float* __restrict__ minmax) {
    const int n = blockIdx.x;
    __shared__ float slo[MMB], shi[MMB];
    slo[threadIdx.x] = ws[((size_t)n * MMB + threadIdx.x) * 2];
    shi[threadIdx.x] = ws[((size_t)n * MMB + threadIdx.x) * 2 + 1];
    __syncthreads();
    for (int st = MMB / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            slo[threadIdx.x] = fminf(slo[threadIdx.x], slo[threadIdx.x + st]);
            shi[threadIdx.x] = fmaxf(shi[threadIdx.x], shi[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        minmax[2 * n] = slo[0];
        minmax[2 * n + 1] = shi[0];
    }
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// scipy 1.1 bytescale: (x - cmin) * (255 / (cmax - cmin)), clip to 0..255, + 0.5, truncate (all in float32)
__device__ __forceinline__ int bytescale(float v, float cmin, float scale) {
    float b = (v - cmin) * scale;
    b = fminf(fmaxf(b, 0.f), 255.f) + 0.5f;
    return (int)b;
}

template <typename T>
__global__ __launch_bounds__(256) void k_frames_resize_h(const T* __restrict__ src, const float* __restrict__ minmax,
                                                         unsigned char* __restrict__ tmp, const int* __restrict__ geo,
                                                         const int* __restrict__ tab, int H, int W, int tmp_w, int KT) {
    const int n = blockIdx.z, y = blockIdx.y;
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int* g = geo + GEO * n;
    const int flip = g[0], sw = g[4];
    if (xo >= sw) return;
    const int* xmin = tab + g[5];
    const int* kk = xmin + sw + (size_t)xo * KT;
    const int x0 = xmin[xo];
    float cmin = 0.f, scale = 1.f;
    if constexpr (sizeof(T) == 4) {
        cmin = minmax[2 * n];
        float cs = minmax[2 * n + 1] - cmin;
        if (cs == 0.f) cs = 1.f;
        scale = 255.0f / cs;
    }
    const T* row = src + ((size_t)n * H + y) * W * 3;
    int acc[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
    for (int i = 0; i < KT; i++) {
        const int k = kk[i];
        int xs = x0 + i;
        if (xs > W - 1) xs = W - 1;                 // padding taps carry weight 0
        if (flip) xs = W - 1 - xs;                  // RandomHorizontalFlip runs before the resize (train.py:166-170)
        const T* px = row + (size_t)xs * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int v;
            if constexpr (sizeof(T) == 4) v = bytescale((float)px[c], cmin, scale);
            else v = (int)px[c];
            acc[c] += k * v;
        }
    }
    unsigned char* d = tmp + (((size_t)n * H + y) * tmp_w + xo) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) d[c] = (unsigned char)clip8(acc[c] >> PREC);
}

__global__ __launch_bounds__(256) void k_frames_resize_v(const unsigned char* __restrict__ tmp, float* __restrict__ dst,
                                                         const int* __restrict__ geo, const int* __restrict__ tab, int H, int tmp_w,
                                                         int KT, int h, int w, float m0, float m1, float m2, float s0, float s1,
                                                         float s2) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    const int* g = geo + GEO * n;
    const int ys = y + g[1], xs = x + g[2], sh = g[3];
    const int* ymin = tab + g[6];
    const int* kk = ymin + sh + (size_t)ys * KT;
    const int y0 = ymin[ys];
    int acc[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
    for (int i = 0; i < KT; i++) {
        const int k = kk[i];
        int yy = y0 + i;
        if (yy > H - 1) yy = H - 1;
        const unsigned char* px = tmp + (((size_t)n * H + yy) * tmp_w + xs) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) acc[c] += k * (int)px[c];
    }
    float* d = dst + (size_t)n * 3 * h * w + p;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; c++) d[(size_t)c * h * w] = ((float)clip8(acc[c] >> PREC) / 255.0f - mean[c]) / stdv[c];
}

// ---- RandomRotate on the device (custom_transforms.py:75-86, composed first in train.py:178-184): scipy.misc.imrotate =
// byte-scale of the float frame to its own min..max, then Pillow's Image.rotate(angle, BILINEAR, expand=False) =
// ImagingGenericTransform + affine_transform + bilinear_filter32RGB (Geometry.c), all in double: the source position of output
// pixel (x, y) is M . (x + 0.5, y + 0.5) (M: six doubles built on the host exactly as Image.rotate builds them); outside
// [0, W) x [0, H) -> 0; else shift by -0.5, floor, blend the clamped 2x2 neighbourhood as a + (b - a) * d (the lower row
// replaced by the upper one beyond the last row), truncate to uint8.  rot per frame: (apply, a0 .. a5, -) doubles; apply == 0:
// the frame is only byte-scaled (what the resize that follows would do first).  Output: uint8 [N,H,W,3].
template <typename T>
__global__ __launch_bounds__(256) void k_frames_rotate(const T* __restrict__ src, const float* __restrict__ minmax,
                                                       unsigned char* __restrict__ dst, const double* __restrict__ rot, int H, int W) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    float cmin = 0.f, scale = 1.f;
    if constexpr (sizeof(T) == 4) {
        cmin = minmax[2 * n];
        float cs = minmax[2 * n + 1] - cmin;
        if (cs == 0.f) cs = 1.f;
        scale = 255.0f / cs;
    }
    const T* img = src + (size_t)n * H * W * 3;
    unsigned char* d = dst + ((size_t)n * H * W + p) * 3;
    auto tap = [&](int yy, int xx, int c) -> double {
        const T v = img[((size_t)yy * W + xx) * 3 + c];
        if constexpr (sizeof(T) == 4) return (double)bytescale((float)v, cmin, scale);
        else return (double)v;
    };
    const double* r = rot + 8 * n;
    if (r[0] == 0.0) {
#pragma unroll
        for (int c = 0; c < 3; c++) d[c] = (unsigned char)(int)tap(y, x, c);
        return;
    }
    const double xs = (double)x + 0.5, ys = (double)y + 0.5;
    double xin = r[1] * xs + r[2] * ys + r[3];
    double yin = r[4] * xs + r[5] * ys + r[6];
    if (xin < 0.0 || xin >= (double)W || yin < 0.0 || yin >= (double)H) {
        d[0] = d[1] = d[2] = 0;
        return;
    }
    xin -= 0.5;
    yin -= 0.5;
    const int xi = (int)floor(xin), yi = (int)floor(yin);
    const double dx = xin - (double)xi, dy = yin - (double)yi;
    const int x0 = xi < 0 ? 0 : (xi < W ? xi : W - 1);
    const int x1 = xi + 1 < 0 ? 0 : (xi + 1 < W ? xi + 1 : W - 1);
    const int y0 = yi < 0 ? 0 : (yi < H ? yi : H - 1);
    const bool low = (yi + 1 >= 0) && (yi + 1 < H);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double a = tap(y0, x0, c), b = tap(y0, x1, c);
        double v1 = a + (b - a) * dx, v2 = v1;
        if (low) {
            const double a2 = tap(yi + 1, x0, c), b2 = tap(yi + 1, x1, c);
            v2 = a2 + (b2 - a2) * dx;
        }
        v1 = v1 + (v2 - v1) * dy;
        d[c] = (unsigned char)(int)v1;
    }
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

size_t cc_frames_resize_ws_bytes(int N, int H, int tmp_w) {
    return (size_t)N * (MMB * 2 + 2) * sizeof(float) + (size_t)N * H * tmp_w * 3;
}

/* RandomHorizontalFlip + RandomScaleCrop (resize to (sh_n, sw_n), crop h x w at (off_y_n, off_x_n)) + ArrayToTensor + Normalize.
 * geo: int32 [N,8] = (flip, off_y, off_x, sh, sw, htab, vtab, 0); tab: int32 resampling tables, per distinct size `first[size]`
 * followed by `weights[size][KT]` (Pillow's 22-bit coefficients, zero padded to KT taps); htab / vtab = offsets into tab.
 * ws: cc_frames_resize_ws_bytes(N, H, tmp_w) bytes, tmp_w >= max sw. */
int cc_frames_resize_to_tensor(const void* src, int src_is_u8, float* dst, const int* geo, const int* tab, int KT, void* ws, int N,
                               int H, int W, int tmp_w, int max_sw, int h, int w, float mean0, float mean1, float mean2, float std0,
                               float std1, float std2, void* stream) {
    if (!src || !dst || !geo || !tab || !ws || N <= 0 || KT <= 0 || h <= 0 || w <= 0 || max_sw <= 0 || max_sw > tmp_w) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* mmws = (float*)ws;
    float* minmax = mmws + (size_t)N * MMB * 2;
    unsigned char* tmp = (unsigned char*)(minmax + (size_t)N * 2);
    dim3 gh((unsigned)((max_sw + 255) / 256), (unsigned)H, (unsigned)N);
    if (!src_is_u8) {
        hipLaunchKernelGGL(k_frames_minmax_partial, dim3(MMB, (unsigned)N), dim3(256), 0, s, (const float*)src, mmws, (long)H * W * 3);
        hipLaunchKernelGGL(k_frames_minmax_final, dim3((unsigned)N), dim3(MMB), 0, s, (const float*)mmws, minmax);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frames_resize_h<float>), gh, dim3(256), 0, s, (const float*)src, (const float*)minmax, tmp,
                           geo, tab, H, W, tmp_w, KT);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frames_resize_h<unsigned char>), gh, dim3(256), 0, s, (const unsigned char*)src,
                           (const float*)minmax, tmp, geo, tab, H, W, tmp_w, KT);
    }
    hipLaunchKernelGGL(k_frames_resize_v, dim3((unsigned)((h * w + 255) / 256), (unsigned)N), dim3(256), 0, s,
                       (const unsigned char*)tmp, dst, geo, tab, H, tmp_w, KT, h, w, mean0, mean1, mean2, std0, std1, std2);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

size_t cc_frames_rotate_ws_bytes(int N) { return (size_t)N * (MMB * 2 + 2) * sizeof(float); }

/* RandomRotate (custom_transforms.py:75-86) for N frames: dst_u8 [N,H,W,3] = Pillow-exact bilinear rotation of the byte-scaled
 * frames.  rot: double [N,8] = (apply, a0..a5, 0) per frame, a* = the affine coefficients Image.rotate computes (host side:
 * cc_amd/custom_transforms.py rotate_matrix); apply == 0: byte-scale only.  ws: cc_frames_rotate_ws_bytes(N). */
int cc_frames_rotate(const void* src, int src_is_u8, void* dst_u8, const double* rot, void* ws, int N, int H, int W, void* stream) {
    if (!src || !dst_u8 || !rot || !ws || N <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* mmws = (float*)ws;
    float* minmax = mmws + (size_t)N * MMB * 2;
    dim3 g((unsigned)((H * W + 255) / 256), (unsigned)N);
    if (!src_is_u8) {
        hipLaunchKernelGGL(k_frames_minmax_partial, dim3(MMB, (unsigned)N), dim3(256), 0, s, (const float*)src, mmws, (long)H * W * 3);
        hipLaunchKernelGGL(k_frames_minmax_final, dim3((unsigned)N), dim3(MMB), 0, s, (const float*)mmws, minmax);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frames_rotate<float>), g, dim3(256), 0, s, (const float*)src, (const float*)minmax,
                           (unsigned char*)dst_u8, rot, H, W);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frames_rotate<unsigned char>), g, dim3(256), 0, s, (const unsigned char*)src,
                           (const float*)minmax, (unsigned char*)dst_u8, rot, H, W);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
