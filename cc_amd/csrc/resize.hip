// x2 bilinear up-sampling of the 1-2 channel prediction maps (F.interpolate(scale_factor=2, mode='bilinear',
// align_corners=False): models/DispResNet6.py:170-186 disp*_up, models/back2future.py:196-285 up_flow / flow*_fwd) with the
// scale factor the callers multiply in afterwards (`20 * up(...)`, `-0.625 * ...`) fused, writing through batch strides so
// that the result can land directly in a channel slice of a concat buffer.  HBM-bound: 4 B read + 16 B written per input
// pixel.  ATen semantics: src = max(0.5 * (o + 0.5) - 0.5, 0), i0 = floor(src), i1 = i0 + (i0 < N-1), lambda = src - i0.
#include "cc_common.h"
#include "../../include/ccengine.h"

namespace {

__device__ __forceinline__ void up2_src(int o, int N, int& i0, int& i1, float& l1) {
    float s = 0.5f * ((float)o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < N - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// one work-item per group of VEC consecutive output pixels of a row (VEC = 4: float4 stores, even W; VEC = 2: any W)
template <int VEC>
__global__ __launch_bounds__(256) void k_upsample2x_fwd(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                                                        long x_bs, long y_bs, float scale, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;        // over [n][c][oy][ox/VEC]
    if (e >= total) return;
    const int OW = 2 * W, OH = 2 * H, q4 = OW / VEC;
    const int oxq = (int)(e % q4);
    long r = e / q4;
    const int oy = (int)(r % OH);
    r /= OH;
    const int c = (int)(r % C), n = (int)(r / C);
    int y0, y1;
    float ly;
    up2_src(oy, H, y0, y1, ly);
    const float* __restrict__ r0 = x + (long)n * x_bs + ((long)c * H + y0) * W;
    const float* __restrict__ r1 = x + (long)n * x_bs + ((long)c * H + y1) * W;
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) {
        int x0, x1;
        float lx;
        up2_src(VEC * oxq + k, W, x0, x1, lx);
        const float top = (1.f - lx) * r0[x0] + lx * r0[x1];
        const float bot = (1.f - lx) * r1[x0] + lx * r1[x1];
        o[k] = scale * ((1.f - ly) * top + ly * bot);
    }
    float* dst = y + (long)n * y_bs + ((long)c * OH + oy) * OW + VEC * oxq;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[VEC - 1]);
    } else {
        dst[0] = o[0];
        dst[1] = o[1];
    }
}

__device__ __forceinline__ float up2_w(int o, int i, int N) {
    int i0, i1;
    float l1;
    up2_src(o, N, i0, i1, l1);
    return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

// gather form of the adjoint (deterministic, no atomics): input pixel i collects from output rows/cols 2i-1 .. 2i+2
__global__ __launch_bounds__(256) void k_upsample2x_bwd(const float* __restrict__ gy, float* __restrict__ gx, int C, int H, int W,
                                                        long gy_bs, long gx_bs, float scale, int accumulate, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;        // over [n][c][iy][ix]
    if (e >= total) return;
    const int ix = (int)(e % W);
    long r = e / W;
    const int iy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), n = (int)(r / C);
    const int OW = 2 * W, OH = 2 * H;
    const float* __restrict__ g = gy + (long)n * gy_bs + (long)c * OH * OW;
    float wx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ox = 2 * ix - 1 + k;
        wx[k] = ((unsigned)ox < (unsigned)OW) ? up2_w(ox, ix, W) : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int oy = 2 * iy - 1 + j;
        if ((unsigned)oy >= (unsigned)OH) continue;
        const float wy = up2_w(oy, iy, H);
        float row = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ox = 2 * ix - 1 + k;
            if ((unsigned)ox < (unsigned)OW) row += wx[k] * g[(long)oy * OW + ox];
        }
        acc += wy * row;
    }
    float* o = gx + (long)n * gx_bs + ((long)c * H + iy) * W + ix;
    const float v = scale * acc;
    *o = accumulate ? *o + v : v;
}

}  // namespace

extern "C" {

int cc_upsample2x_fwd(const float* x, float* y, int B, int C, int H, int W, long x_bs, long y_bs, float scale, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    if (!(W & 1) && !(y_bs & 3) && !((uintptr_t)y & 15)) {
        const long total = (long)B * C * (2 * H) * (W >> 1);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_upsample2x_fwd<4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x, y, C, H, W, x_bs, y_bs, scale, total);
    } else {
        const long total = (long)B * C * (2 * H) * W;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_upsample2x_fwd<2>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x, y, C, H, W, x_bs, y_bs, scale, total);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_upsample2x_bwd(const float* gy, float* gx, int B, int C, int H, int W, long gy_bs, long gx_bs, float scale, int accumulate,
                      void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(k_upsample2x_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, C, H, W,
                       gy_bs, gx_bs, scale, accumulate, total);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
