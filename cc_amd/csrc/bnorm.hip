// Training-mode BatchNorm2d of the 13 ResNet shortcut branches of DispResNet6 (models/DispResNet6.py:45-60:
// Conv1x1(stride) + BatchNorm2d).  The full- and half-resolution ones are 16-64 channels x 53 k - 850 k values per
// channel: a one-workgroup-per-channel kernel (what the vendor library runs here) leaves the chip idle for ~350 us;
// these are HBM-bound passes that want thousands of workgroups.
//
//   forward : k_bn_stats   partial (sum, sum of squares) of (x - K_c) per (chunk, channel, image)   [K_c = x[0,c,0]: shift
//                          against cancellation]
//             k_bn_finalize per channel: mean, biased var, invstd; running stats (momentum, unbiased var); the affine
//                          y = x * scale_c + shift_c
//             k_bn_apply   y = x * scale + shift
//   backward: k_bn_bwd_stats partial (sum dy, sum dy * (x - mean)); k_bn_bwd_finalize: dbias, dweight and the three
//             coefficients of  dx = c1 * (dy - c2 - (x - mean) * c3);  k_bn_bwd_apply
// All reductions are two-stage with a fixed order (deterministic).  float4 when the plane size allows.
#include "cc_common.h"
#include "../../include/ccengine.h"

namespace {

constexpr int BN_MAXCHUNK = 64;       // partials per channel = cpp * B <= 64

template <bool VEC4, bool BWD>
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, const float* __restrict__ gy,
                                                  const float* __restrict__ mean, float* __restrict__ partial, int C, int HW,
                                                  long x_bs) {
    __shared__ float red[8];
    const int c = blockIdx.y, n = blockIdx.z, cpp = gridDim.x;
    const float* __restrict__ xp = x + (long)n * x_bs + (long)c * HW;
    const float* __restrict__ gp = BWD ? gy + (long)n * x_bs + (long)c * HW : nullptr;
    const float k = BWD ? mean[c] : x[(long)c * HW];            // forward: shift by the channel's first value
    float s[2] = {0.f, 0.f};
    if (VEC4) {
        // four iterations' loads in flight per work item, accumulated in element order
        const int nq = HW >> 2, stp = cpp * 256;
        for (int q0 = blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += 4 * stp) {
            float4 vv[4], gg[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int q = q0 + u * stp;
                vv[u] = (q < nq) ? ((const float4*)xp)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                gg[u] = (BWD && q < nq) ? ((const float4*)gp)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (q0 + u * stp >= nq) continue;
                const float4 v = vv[u];
                const float d[4] = {v.x - k, v.y - k, v.z - k, v.w - k};
                if (BWD) {
                    const float4 g = gg[u];
                    s[0] += (g.x + g.y) + (g.z + g.w);
                    s[1] += (g.x * d[0] + g.y * d[1]) + (g.z * d[2] + g.w * d[3]);
                } else {
                    s[0] += (d[0] + d[1]) + (d[2] + d[3]);
                    s[1] += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                }
            }
        }
    } else {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += cpp * 256) {
            const float d = xp[e] - k;
            if (BWD) { const float g = gp[e]; s[0] += g; s[1] += g * d; }
            else { s[0] += d; s[1] += d * d; }
        }
    }
    cc::block_sum_256<2>(s, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((long)c * (cpp * gridDim.z) + n * cpp + blockIdx.x) * 2;
        o[0] = s[0];
        o[1] = s[1];
    }
}

// one wave per channel
__global__ __launch_bounds__(64) void k_bn_finalize(const float* __restrict__ partial, int nchunk, const float* __restrict__ x,
                                                    int HW, const float* __restrict__ weight, const float* __restrict__ bias,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                    float* __restrict__ scale_shift, int C, float count, float momentum,
                                                    float eps) {
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < nchunk; i += 64) {
        s0 += partial[((long)c * nchunk + i) * 2];
        s1 += partial[((long)c * nchunk + i) * 2 + 1];
    }
    s0 = cc::wave_sum(s0);
    s1 = cc::wave_sum(s1);
    if (threadIdx.x == 0) {
        const float k = x[(long)c * HW];
        const float md = s0 / count;                       // mean of (x - k)
        const float mean = k + md;
        float var = s1 / count - md * md;                  // biased variance (shift-invariant)
        var = var < 0.f ? 0.f : var;
        const float invstd = 1.f / sqrtf(var + eps);
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) {
            const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
        const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
        scale_shift[c] = w * invstd;
        scale_shift[C + c] = b - mean * (w * invstd);
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, float* __restrict__ y,
                                                  const float* __restrict__ scale_shift, int C, int HW, long x_bs) {
    const int c = blockIdx.y, n = blockIdx.z, cpp = gridDim.x;
    const float sc = scale_shift[c], sh = scale_shift[C + c];
    const float* __restrict__ xp = x + (long)n * x_bs + (long)c * HW;
    float* __restrict__ yp = y + (long)n * x_bs + (long)c * HW;
    if (VEC4) {
        const int nq = HW >> 2;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += cpp * 256) {
            float4 v = ((const float4*)xp)[q];
            v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            ((float4*)yp)[q] = v;
        }
    } else {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += cpp * 256) yp[e] = fmaf(xp[e], sc, sh);
    }
}

__global__ __launch_bounds__(64) void k_bn_bwd_finalize(const float* __restrict__ partial, int nchunk,
                                                        const float* __restrict__ weight, const float* __restrict__ save_invstd,
                                                        float* __restrict__ gweight, float* __restrict__ gbias,
                                                        float* __restrict__ coef, int C, float count, int accumulate) {
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < nchunk; i += 64) {
        s0 += partial[((long)c * nchunk + i) * 2];
        s1 += partial[((long)c * nchunk + i) * 2 + 1];
    }
    s0 = cc::wave_sum(s0);
    s1 = cc::wave_sum(s1);
    if (threadIdx.x == 0) {
        const float invstd = save_invstd[c], w = weight ? weight[c] : 1.f;
        const float gb = s0, gw = s1 * invstd;
        if (gbias) gbias[c] = accumulate ? gbias[c] + gb : gb;
        if (gweight) gweight[c] = accumulate ? gweight[c] + gw : gw;
        coef[c] = w * invstd;                               // c1
        coef[C + c] = s0 / count;                           // c2 = mean(dy)
        coef[2 * C + c] = s1 * invstd * invstd / count;     // c3 = invstd^2 * mean(dy * (x - mean))
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ gy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ coef,
                                                      float* __restrict__ gx, int C, int HW, long x_bs) {
    const int c = blockIdx.y, n = blockIdx.z, cpp = gridDim.x;
    const float c1 = coef[c], c2 = coef[C + c], c3 = coef[2 * C + c], m = mean[c];
    const float* __restrict__ xp = x + (long)n * x_bs + (long)c * HW;
    const float* __restrict__ gp = gy + (long)n * x_bs + (long)c * HW;
    float* __restrict__ op = gx + (long)n * x_bs + (long)c * HW;
    if (VEC4) {
        const int nq = HW >> 2;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += cpp * 256) {
            const float4 v = ((const float4*)xp)[q], g = ((const float4*)gp)[q];
            float4 o;
            o.x = c1 * (g.x - c2 - (v.x - m) * c3);
            o.y = c1 * (g.y - c2 - (v.y - m) * c3);
            o.z = c1 * (g.z - c2 - (v.z - m) * c3);
            o.w = c1 * (g.w - c2 - (v.w - m) * c3);
            ((float4*)op)[q] = o;
        }
    } else {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += cpp * 256) op[e] = c1 * (gp[e] - c2 - (xp[e] - m) * c3);
    }
}

// ---- small maps (B*H*W <= BN_SMALL_MAX values per channel: the 9 deep shortcut branches, 128-512 channels on <= 32x104 maps)
// ONE launch per direction: a workgroup owns a channel, reduces its B*HW values (second pass re-reads them from L1/L2),
// finalises the statistics itself and applies them.  Same formulas / summation structure as the three-kernel path above
// (per-thread strided partial sums -> wave shuffles -> 4-wave sum), fixed order -> deterministic.
constexpr int BN_SMALL_MAX = 16384;

__global__ __launch_bounds__(256) void k_bn_small_fwd(const float* __restrict__ x, const float* __restrict__ weight,
                                                      const float* __restrict__ bias, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, float* __restrict__ y,
                                                      float* __restrict__ save_mean, float* __restrict__ save_invstd, int B,
                                                      int C, int HW, float momentum, float eps) {
    __shared__ float red[8];
    __shared__ float bc[2];
    const int c = blockIdx.x;
    const long bs = (long)C * HW;
    const float k = x[(long)c * HW];
    float s[2] = {0.f, 0.f};
    for (int n = 0; n < B; n++) {
        const float* __restrict__ xp = x + (long)n * bs + (long)c * HW;
        // four loads in flight per work item, accumulated in element order (the plain loop is one load latency per element)
        for (int e0 = threadIdx.x; e0 < HW; e0 += 1024) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = (e0 + 256 * u < HW) ? xp[e0 + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e0 + 256 * u < HW) {
                    const float d = v[u] - k;
                    s[0] += d;
                    s[1] += d * d;
                }
        }
    }
    cc::block_sum_256<2>(s, red);
    if (threadIdx.x == 0) {
        const float count = (float)((long)B * HW);
        const float md = s[0] / count;
        const float mean = k + md;
        float var = s[1] / count - md * md;
        var = var < 0.f ? 0.f : var;
        const float invstd = 1.f / sqrtf(var + eps);
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) {
            const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
        const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
        bc[0] = w * invstd;
        bc[1] = b - mean * (w * invstd);
    }
    __syncthreads();
    const float sc = bc[0], sh = bc[1];
    for (int n = 0; n < B; n++) {
        const float* __restrict__ xp = x + (long)n * bs + (long)c * HW;
        float* __restrict__ yp = y + (long)n * bs + (long)c * HW;
        for (int e0 = threadIdx.x; e0 < HW; e0 += 1024) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = (e0 + 256 * u < HW) ? xp[e0 + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e0 + 256 * u < HW) yp[e0 + 256 * u] = fmaf(v[u], sc, sh);
        }
    }
}

__global__ __launch_bounds__(256) void k_bn_small_bwd(const float* __restrict__ gy, const float* __restrict__ x,
                                                      const float* __restrict__ weight, const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_invstd, float* __restrict__ gx,
                                                      float* __restrict__ gweight, float* __restrict__ gbias, int B, int C,
                                                      int HW, int accumulate) {
    __shared__ float red[8];
    __shared__ float bc[3];
    const int c = blockIdx.x;
    const long bs = (long)C * HW;
    const float m = save_mean[c];
    float s[2] = {0.f, 0.f};
    for (int n = 0; n < B; n++) {
        const float* __restrict__ xp = x + (long)n * bs + (long)c * HW;
        const float* __restrict__ gp = gy + (long)n * bs + (long)c * HW;
        for (int e0 = threadIdx.x; e0 < HW; e0 += 1024) {          // four (gy, x) pairs in flight, accumulated in element order
            float g[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = e0 + 256 * u < HW;
                g[u] = in ? gp[e0 + 256 * u] : 0.f;
                v[u] = in ? xp[e0 + 256 * u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e0 + 256 * u < HW) {
                    s[0] += g[u];
                    s[1] += g[u] * (v[u] - m);
                }
        }
    }
    cc::block_sum_256<2>(s, red);
    if (threadIdx.x == 0) {
        const float count = (float)((long)B * HW);
        const float invstd = save_invstd[c], w = weight ? weight[c] : 1.f;
        const float gb = s[0], gw = s[1] * invstd;
        if (gbias) gbias[c] = accumulate ? gbias[c] + gb : gb;
        if (gweight) gweight[c] = accumulate ? gweight[c] + gw : gw;
        bc[0] = w * invstd;
        bc[1] = s[0] / count;
        bc[2] = s[1] * invstd * invstd / count;
    }
    __syncthreads();
    const float c1 = bc[0], c2 = bc[1], c3 = bc[2];
    for (int n = 0; n < B; n++) {
        const float* __restrict__ xp = x + (long)n * bs + (long)c * HW;
        const float* __restrict__ gp = gy + (long)n * bs + (long)c * HW;
        float* __restrict__ op = gx + (long)n * bs + (long)c * HW;
        for (int e0 = threadIdx.x; e0 < HW; e0 += 1024) {
            float g[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = e0 + 256 * u < HW;
                g[u] = in ? gp[e0 + 256 * u] : 0.f;
                v[u] = in ? xp[e0 + 256 * u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e0 + 256 * u < HW) op[e0 + 256 * u] = c1 * (g[u] - c2 - (v[u] - m) * c3);
        }
    }
}

// eval mode: scale_c = w_c / sqrt(running_var_c + eps), shift_c = b_c - running_mean_c * scale_c
__global__ __launch_bounds__(256) void k_bn_eval_coef(const float* __restrict__ weight, const float* __restrict__ bias,
                                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                      float* __restrict__ scale_shift, int C, float eps, int grad_only) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = (weight ? weight[c] : 1.f) / sqrtf(rvar[c] + eps);
    scale_shift[c] = sc;
    scale_shift[C + c] = grad_only ? 0.f : ((bias ? bias[c] : 0.f) - rmean[c] * sc);
}

inline int bn_cpp(int B, int HW) {
    int cpp = (HW + 8191) / 8192;
    const int cap = BN_MAXCHUNK / B > 0 ? BN_MAXCHUNK / B : 1;
    return cpp < 1 ? 1 : (cpp > cap ? cap : cpp);
}

}  // namespace

extern "C" {

size_t cc_bn_ws_bytes(int C) { return (size_t)C * (2 * BN_MAXCHUNK + 3) * sizeof(float); }

/* nn.BatchNorm2d forward in training mode (models/DispResNet6.py:53-56): y = (x - mean_c) / sqrt(var_c + eps) * w_c + b_c
 * with batch statistics over (B, H, W); running_mean / running_var updated with `momentum` (unbiased variance), either
 * may be null; save_mean / save_invstd [C] are kept for the backward.  x, y: [B,C,H,W] contiguous.  ws: cc_bn_ws_bytes(C). */
int cc_bn_train_fwd(const float* x, const float* weight_or_null, const float* bias_or_null, float* running_mean_or_null,
                    float* running_var_or_null, float* y, float* save_mean, float* save_invstd, float* ws, int B, int C,
                    int H, int W, float momentum, float eps, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    if ((long)B * HW <= BN_SMALL_MAX) {
        hipLaunchKernelGGL(k_bn_small_fwd, dim3(C), dim3(256), 0, s, x, weight_or_null, bias_or_null, running_mean_or_null,
                           running_var_or_null, y, save_mean, save_invstd, B, C, HW, momentum, eps);
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    if (B > BN_MAXCHUNK) return CC_ERR_ARG;
    const long bs = (long)C * HW;
    const int cpp = bn_cpp(B, HW), nchunk = cpp * B;
    float* partial = ws;
    float* scale_shift = ws + (size_t)C * 2 * BN_MAXCHUNK;
    const bool v4 = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    dim3 g(cpp, C, B);
    if (v4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_stats<true, false>), g, dim3(256), 0, s, x, (const float*)nullptr, (const float*)nullptr, partial, C, HW, bs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_stats<false, false>), g, dim3(256), 0, s, x, (const float*)nullptr, (const float*)nullptr, partial, C, HW, bs);
    hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(64), 0, s, (const float*)partial, nchunk, x, HW, weight_or_null, bias_or_null,
                       running_mean_or_null, running_var_or_null, save_mean, save_invstd, scale_shift, C, (float)((long)B * HW),
                       momentum, eps);
    if (v4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_apply<true>), g, dim3(256), 0, s, x, y, (const float*)scale_shift, C, HW, bs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_apply<false>), g, dim3(256), 0, s, x, y, (const float*)scale_shift, C, HW, bs);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* its backward: gx, gweight[c] (+)= sum gy * xhat, gbias[c] (+)= sum gy  (gweight / gbias may be null) */
int cc_bn_train_bwd(const float* gy, const float* x, const float* weight_or_null, const float* save_mean,
                    const float* save_invstd, float* gx, float* gweight_or_null, float* gbias_or_null, float* ws, int B, int C,
                    int H, int W, int accumulate_wb, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    if ((long)B * HW <= BN_SMALL_MAX) {
        hipLaunchKernelGGL(k_bn_small_bwd, dim3(C), dim3(256), 0, s, gy, x, weight_or_null, save_mean, save_invstd, gx,
                           gweight_or_null, gbias_or_null, B, C, HW, accumulate_wb);
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    if (B > BN_MAXCHUNK) return CC_ERR_ARG;
    const long bs = (long)C * HW;
    const int cpp = bn_cpp(B, HW), nchunk = cpp * B;
    float* partial = ws;
    float* coef = ws + (size_t)C * 2 * BN_MAXCHUNK;
    const bool v4 = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx) % 16 == 0);
    dim3 g(cpp, C, B);
    if (v4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_stats<true, true>), g, dim3(256), 0, s, x, gy, save_mean, partial, C, HW, bs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_stats<false, true>), g, dim3(256), 0, s, x, gy, save_mean, partial, C, HW, bs);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C), dim3(64), 0, s, (const float*)partial, nchunk, weight_or_null, save_invstd,
                       gweight_or_null, gbias_or_null, coef, C, (float)((long)B * HW), accumulate_wb);
    if (v4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_bwd_apply<true>), g, dim3(256), 0, s, gy, x, save_mean, (const float*)coef, gx, C, HW, bs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_bwd_apply<false>), g, dim3(256), 0, s, gy, x, save_mean, (const float*)coef, gx, C, HW, bs);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* nn.BatchNorm2d forward in eval mode: the affine map of the running statistics, y = (x - running_mean_c) / sqrt(running_var_c
 * + eps) * w_c + b_c (DispResNet6's shortcut BatchNorms inside the validation loops, train.py:588-777).  grad_only != 0: the
 * input gradient of that map instead, y = x * w_c / sqrt(running_var_c + eps) (x = upstream gradient).  ws: 2*C floats. */
int cc_bn_eval_fwd(const float* x, const float* weight_or_null, const float* bias_or_null, const float* running_mean,
                   const float* running_var, float* y, float* ws, int B, int C, int H, int W, float eps, int grad_only,
                   void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !running_mean || !running_var) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    hipLaunchKernelGGL(k_bn_eval_coef, dim3((C + 255) / 256), dim3(256), 0, s, weight_or_null, bias_or_null, running_mean,
                       running_var, ws, C, eps, grad_only);
    int cpp = (HW + 8191) / 8192;
    if (cpp < 1) cpp = 1;
    if (cpp > 64) cpp = 64;
    const bool v4 = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    dim3 g(cpp, C, B);
    const long bs = (long)C * HW;
    if (v4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_apply<true>), g, dim3(256), 0, s, x, y, (const float*)ws, C, HW, bs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_apply<false>), g, dim3(256), 0, s, x, y, (const float*)ws, C, HW, bs);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
