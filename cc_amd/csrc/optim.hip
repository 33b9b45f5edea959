// Fused Adam over the flat fp32 parameter bucket (torch.optim.Adam of train.py:307-310,568: betas (0.9, 0.999),
// eps 1e-8, no weight decay, no amsgrad) -- one launch for all 74.26 M parameters of the four nets instead of
// ~460 per-tensor update chains.  HBM-bound: 4 reads + 3 writes of 4 B per parameter (28 B), float4 accesses.
// The step counter lives on the device so the launch can sit inside a hipGraph.
#include "cc_common.h"
#include "../../include/ccengine.h"

namespace {

__global__ void k_adam_tick(float* step) { step[0] += 1.0f; }

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                              const float* __restrict__ step, float grad_scale) {
    const float t = step[0];
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, rs2 = 1.f / sqrtf(bc2);
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 pp = *reinterpret_cast<float4*>(p + i4);
        const float4 gg = *reinterpret_cast<const float4*>(g + i4);
        float4 mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float gr = G[k] * grad_scale;
            M[k] = b1 * M[k] + (1.f - b1) * gr;
            V[k] = b2 * V[k] + (1.f - b2) * gr * gr;
            P[k] -= step_size * (M[k] / (sqrtf(V[k]) * rs2 + eps));
        }
        *reinterpret_cast<float4*>(p + i4) = pp;
        *reinterpret_cast<float4*>(m + i4) = mm;
        *reinterpret_cast<float4*>(v + i4) = vv;
    } else {
        for (long i = i4; i < n; i++) {
            const float gr = g[i] * grad_scale;
            m[i] = b1 * m[i] + (1.f - b1) * gr;
            v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
            p[i] -= step_size * (m[i] / (sqrtf(v[i]) * rs2 + eps));
        }
    }
}

__global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, long n, float value) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) *reinterpret_cast<float4*>(p + i4) = make_float4(value, value, value, value);
    else for (long i = i4; i < n; i++) p[i] = value;
}

}  // namespace

extern "C" {

int cc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, long n, float lr,
                 float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (n <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, s, step_dev);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, (const float*)step_dev, grad_scale);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* The same update on a sub-range of the bucket (the caller passes the range's base pointers); tick = 0 leaves the step counter as
 * it is: the second and later segments of one optimizer step.  Lets the update of a segment whose gradients have arrived run
 * while the all-reduce of the next segment is still in flight. */
int cc_adam_step_segment(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, long n, float lr,
                         float beta1, float beta2, float eps, float grad_scale, int tick, void* stream) {
    if (n <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (tick) hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, s, step_dev);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, (const float*)step_dev, grad_scale);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_adam_tick(float* step_dev, void* stream) {
    if (!step_dev) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_fill(float* p, long n, float value, void* stream) {
    if (n <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, p, n, value);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
