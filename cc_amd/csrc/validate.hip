// Validation side of the training loop (reference train.py:588-777): the rigidity-mask composition of
// validate_flow_with_gt and the two out-of-bound maps it logs, one elementwise pass over the network outputs instead of the
// reference's ~25 stock-torch launches.  HBM-bound: 6 planes in, up to 11 planes out per pixel.
#include <hip/hip_runtime.h>
#include "cc_common.h"
#include "../../include/ccengine.h"

namespace {

struct RigidityOut {
    float* rigidity;        // [B,1,H,W]  train.py:676
    float* census;          // [B,H,W]    :678-681
    float* combined;        // [B,1,H,W]  :683
    float* flow_non_rigid;  // [B,2,H,W]  :685
    float* flow_rigid;      // [B,2,H,W]  :686
    float* total_flow;      // [B,2,H,W]  :687
    float* oob_rigid;       // [B,H,W]    :673 (inverse_warp.py:222-238 on flow_cam)
    float* oob_non_rigid;   // [B,H,W]    :674
};

// inverse_warp.py:230-238: X = 2*((x+u)/(w-1) - 0.5), Y likewise; oob = |X| > 1 or |Y| > 1
__device__ __forceinline__ float oob_of(float u, float v, int x, int y, float wm1, float hm1) {
    const float X = 2.f * (((float)x + u) / wm1 - 0.5f);
    const float Y = 2.f * (((float)y + v) / hm1 - 0.5f);
    return (fabsf(X) > 1.f || fabsf(Y) > 1.f) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void k_rigidity_compose(const float* __restrict__ exp_mask, int MC,
                                                          const float* __restrict__ flow_cam,
                                                          const float* __restrict__ flow_fwd, RigidityOut o, float thresh,
                                                          int B, int H, int W) {
    const int HW = H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int b = (int)(i / HW), p = (int)(i - (long)b * HW);
    const float* m = exp_mask + (long)b * MC * HW + p;
    const float m1 = m[HW], m2 = m[2 * HW];
    const long f = (long)b * 2 * HW + p;
    const float cu = flow_cam[f], cv = flow_cam[f + HW], fu = flow_fwd[f], fv = flow_fwd[f + HW];
    // :676  1 - (1-m1)*(1-m2) > 0.5   (the comparison binds last)
    const float rig = (1.f - (1.f - m1) * (1.f - m2)) > 0.5f ? 1.f : 0.f;
    // :678-681
    const float cen = ((fabsf(cu - fu) < thresh) ? 1.f : 0.f) * ((fabsf(cv - fv) < thresh) ? 1.f : 0.f);
    // :683
    const float comb = 1.f - (1.f - rig) * (1.f - cen);
    const float wn = (comb <= thresh) ? 1.f : 0.f, wr = (comb > thresh) ? 1.f : 0.f;
    const float nu = wn * fu, nv = wn * fv, ru = wr * cu, rv = wr * cv;
    if (o.rigidity) o.rigidity[i] = rig;
    if (o.census) o.census[i] = cen;
    if (o.combined) o.combined[i] = comb;
    if (o.flow_non_rigid) { o.flow_non_rigid[f] = nu; o.flow_non_rigid[f + HW] = nv; }
    if (o.flow_rigid) { o.flow_rigid[f] = ru; o.flow_rigid[f + HW] = rv; }
    if (o.total_flow) { o.total_flow[f] = ru + nu; o.total_flow[f + HW] = rv + nv; }
    if (o.oob_rigid || o.oob_non_rigid) {
        const int y = p / W, x = p - y * W;
        const float wm1 = (float)W - 1.f, hm1 = (float)H - 1.f;
        if (o.oob_rigid) o.oob_rigid[i] = oob_of(cu, cv, x, y, wm1, hm1);
        if (o.oob_non_rigid) o.oob_non_rigid[i] = oob_of(fu, fv, x, y, wm1, hm1);
    }
}

}  // namespace

extern "C" {

int cc_rigidity_compose(const float* exp_mask, int MC, const float* flow_cam, const float* flow_fwd, float* rigidity,
                        float* census, float* combined, float* flow_non_rigid, float* flow_rigid, float* total_flow,
                        float* oob_rigid, float* oob_non_rigid, float thresh, int B, int H, int W, void* stream) {
    if (!exp_mask || !flow_cam || !flow_fwd || MC < 3 || B <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    RigidityOut o = {rigidity, census, combined, flow_non_rigid, flow_rigid, total_flow, oob_rigid, oob_non_rigid};
    const long n = (long)B * H * W;
    hipLaunchKernelGGL(k_rigidity_compose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, exp_mask, MC,
                       flow_cam, flow_fwd, o, thresh, B, H, W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
