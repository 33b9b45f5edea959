// Elementwise / stencil loss kernels of the CC step (gfx950), all HBM-bound:
//   image pyramid (adaptive_avg_pool2d), rigid + flow occlusion masks, edge-aware and second-order
//   smoothness, explainability BCE, consensus (weighted) BCE -- each with its gradient produced in
//   the same pass (the loss is a scalar with constant weights, so d(loss)/d(input) is known as soon
//   as the forward value is) and with deterministic two-stage reductions (no float atomics).
// Replaces the per-scale Python loops of loss_functions.py:132-137,148-155,221-261,287-352.
#include "cc_common.h"
#include "jobs.h"
#include "../../include/ccengine.h"

namespace {

using ccjobs::JobTab;

// ------------------------------------------------------------------ pyramid
// F.adaptive_avg_pool2d(img, (h, w)) (loss_functions.py:36-37,89-90,163-165,315): window
// [floor(i*H/h), ceil((i+1)*H/h)), summed row-major in fp32 then divided by the count (ATen order).
__global__ __launch_bounds__(256) void k_adaptive_pool(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                       int h, int w, int planes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = h * w;
    if (i >= n) return;
    const int pl = blockIdx.y;
    const int oy = i / w, ox = i - oy * w;
    const int y0 = (int)(((long)oy * H) / h), y1 = (int)((((long)oy + 1) * H + h - 1) / h);
    const int x0 = (int)(((long)ox * W) / w), x1 = (int)((((long)ox + 1) * W + w - 1) / w);
    const float* src = in + (size_t)pl * H * W;
    float s = 0.f;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) s += src[y * W + x];
    out[(size_t)pl * n + i] = s / (float)((y1 - y0) * (x1 - x0));
}

// all pyramid levels of one 32x32 level-0 tile (see cc_pyramid_build)
__global__ __launch_bounds__(256) void k_pyramid_tile(const float* __restrict__ in, float* __restrict__ out, int nlevels,
                                                      int planes, int H, int W) {
    constexpr int TS = 36;                    // row stride: float4-aligned rows
    __shared__ __attribute__((aligned(16))) float tile[32 * TS];
    const int tw = W / 32;
    const int ty = blockIdx.x / tw, tx = blockIdx.x - ty * tw;
    const int pl = blockIdx.y;
    const float* src = in + (size_t)pl * H * W + (size_t)(ty * 32) * W + tx * 32;
    {
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;        // row 0..31, float4 0..7
        *(float4*)(tile + r * TS + 4 * q) = *(const float4*)(src + (size_t)r * W + 4 * q);
    }
    __syncthreads();
    size_t off = 0;
    for (int l = 1; l < nlevels; l++) {
        const int k = 1 << l, n = 32 >> l;              // window size, outputs per tile side
        const int h = H >> l, w = W >> l;
        if ((int)threadIdx.x < n * n) {
            const int oy = threadIdx.x / n, ox = threadIdx.x - oy * n;
            float s = 0.f;
            // the reference's row-major summation order; the loads of a window row are issued together (float4 / float2)
            // so that the chain waits for LDS once per row instead of once per element
            if (k >= 4) {
                for (int y = 0; y < k; y++) {
                    const float4* row = (const float4*)(tile + (oy * k + y) * TS + ox * k);
                    for (int x4 = 0; x4 < k / 4; x4 += 2) {
                        const float4 a = row[x4];
                        const float4 b = (x4 + 1 < k / 4) ? row[x4 + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
                        s += a.x; s += a.y; s += a.z; s += a.w;
                        if (x4 + 1 < k / 4) { s += b.x; s += b.y; s += b.z; s += b.w; }
                    }
                }
            } else {
                for (int y = 0; y < k; y++) {
                    const float2 a = *(const float2*)(tile + (oy * k + y) * TS + ox * k);
                    s += a.x; s += a.y;
                }
            }
            out[off + (size_t)pl * h * w + (size_t)(ty * n + oy) * w + tx * n + ox] = s / (float)(k * k);
        }
        off += (size_t)planes * h * w;
    }
}

// ... of up to PYR_MAXIMG images of one size in ONE launch (the target frame and the reference frames of a step): blockIdx.z = image
constexpr int PYR_MAXIMG = 8;
struct PyrMulti { const float* in[PYR_MAXIMG]; float* out[PYR_MAXIMG]; };
__global__ __launch_bounds__(256) void k_pyramid_tile_multi(PyrMulti t, int nlevels, int planes, int H, int W) {
    constexpr int TS = 36;
    __shared__ __attribute__((aligned(16))) float tile[32 * TS];
    const float* in = t.in[blockIdx.z];
    float* out = t.out[blockIdx.z];
    const int tw = W / 32;
    const int ty = blockIdx.x / tw, tx = blockIdx.x - ty * tw;
    const int pl = blockIdx.y;
    const float* src = in + (size_t)pl * H * W + (size_t)(ty * 32) * W + tx * 32;
    {
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
        *(float4*)(tile + r * TS + 4 * q) = *(const float4*)(src + (size_t)r * W + 4 * q);
    }
    __syncthreads();
    size_t off = 0;
    for (int l = 1; l < nlevels; l++) {           // (the summation order of k_pyramid_tile: bit-identical results)
        const int k = 1 << l, n = 32 >> l;
        const int h = H >> l, w = W >> l;
        if ((int)threadIdx.x < n * n) {
            const int oy = threadIdx.x / n, ox = threadIdx.x - oy * n;
            float s = 0.f;
            if (k >= 4) {
                for (int y = 0; y < k; y++) {
                    const float4* row = (const float4*)(tile + (oy * k + y) * TS + ox * k);
                    for (int x4 = 0; x4 < k / 4; x4 += 2) {
                        const float4 a = row[x4];
                        const float4 b = (x4 + 1 < k / 4) ? row[x4 + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
                        s += a.x; s += a.y; s += a.z; s += a.w;
                        if (x4 + 1 < k / 4) { s += b.x; s += b.y; s += b.z; s += b.w; }
                    }
                }
            } else {
                for (int y = 0; y < k; y++) {
                    const float2 a = *(const float2*)(tile + (oy * k + y) * TS + ox * k);
                    s += a.x; s += a.y;
                }
            }
            out[off + (size_t)pl * h * w + (size_t)(ty * n + oy) * w + tx * n + ox] = s / (float)(k * k);
        }
        off += (size_t)planes * h * w;
    }
}

// ------------------------------------------------------------------ occlusion masks
// loss_functions.py:343-352 occlusion_masks: occ = sum_c(f_fw + f_bw) > 0.08*(|f_fw|^2 + |f_bw|^2) + 1
// (signed sum; occ_fw == occ_bw).  Output is (1 - occ), the factor the losses multiply by.
__device__ __forceinline__ float noocc(float bu, float bv, float fu, float fv) {
    const float mag = (fu * fu + fv * fv) + (bu * bu + bv * bv);
    const float thr = 0.08f * mag + 1.0f;
    const float s = (fu + bu) + (fv + bv);
    return (s > thr) ? 0.f : 1.f;
}

__global__ __launch_bounds__(256) void k_flow_noocc(const float* __restrict__ flow_bw, const float* __restrict__ flow_fw,
                                                    float* __restrict__ out, int HW) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int b = blockIdx.y;
    const float* fb = flow_bw + (size_t)b * 2 * HW + p;
    const float* ff = flow_fw + (size_t)b * 2 * HW + p;
    out[(size_t)b * HW + p] = noocc(fb[0], fb[HW], ff[0], ff[HW]);
}

// loss_functions.py:132-137 depth_occlusion_masks: four rigid flows (pose2flow with the FULL-resolution
// K at every scale, Q4), pairs (1,2) and (0,3) -> (1 - occ) for refs 0..3, [B,4,H,W].
// flows: [4][B,2,H,W] computed by cc_pose2flow_fwd into one buffer.
__global__ __launch_bounds__(256) void k_rigid_noocc(const float* __restrict__ flows, float* __restrict__ out, int B,
                                                     int HW) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int b = blockIdx.y;
    const size_t fs = (size_t)B * 2 * HW;
    const float* f0 = flows + 0 * fs + (size_t)b * 2 * HW + p;
    const float* f1 = flows + 1 * fs + (size_t)b * 2 * HW + p;
    const float* f2 = flows + 2 * fs + (size_t)b * 2 * HW + p;
    const float* f3 = flows + 3 * fs + (size_t)b * 2 * HW + p;
    const float m12 = noocc(f1[0], f1[HW], f2[0], f2[HW]);   // occlusion_masks(flow_cam[1], flow_cam[2])
    const float m03 = noocc(f0[0], f0[HW], f3[0], f3[HW]);   // occlusion_masks(flow_cam[0], flow_cam[3])
    float* o = out + (size_t)b * 4 * HW + p;
    o[0] = m03;
    o[HW] = m12;
    o[2 * HW] = m12;
    o[3 * HW] = m03;
}

// ------------------------------------------------------------------ generic deterministic finalize
// accum[0] += coef * sum(partials[0..n))
__global__ __launch_bounds__(256) void k_reduce_add(const float* __restrict__ partials, int n, float coef,
                                                    float* __restrict__ accum) {
    __shared__ float red[4];
    float v[1] = {0.f};
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {        // eight loads in flight, added in index order
        float p8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) p8[u] = (i0 + 256 * u < n) ? partials[i0 + 256 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + 256 * u < n) v[0] += p8[u];
    }
    cc::block_sum_256<1>(v, red);
    if (threadIdx.x == 0) accum[0] += coef * v[0];
}

// ------------------------------------------------------------------ edge-aware smoothness
// loss_functions.py:287-319 (per scale): mean(|p[y]-p[y+1]| * exp(-mean_c|im[y]-im[y+1]|)) over [B,C,H-1,W]
//                                     + mean(|p[x]-p[x+1]| * exp(-mean_c|im[x]-im[x+1]|)) over [B,C,H,W-1]
// ("gradient_x" runs along H, Q7).  One thread per pixel of one (b, c) plane; the gradient is gathered.
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ float edge_w(const float* __restrict__ im, int HW, int p, int q) {
    const float a = fabsf(im[p] - im[q]) + fabsf(im[HW + p] - im[HW + q]) + fabsf(im[2 * HW + p] - im[2 * HW + q]);
    return expf(-(a / 3.f));
}

__global__ __launch_bounds__(256) void k_edge_smooth(const float* __restrict__ img, const float* __restrict__ pred,
                                                     float* __restrict__ gpred, float* __restrict__ partials, int C, int H,
                                                     int W, float inv_nx, float inv_ny, float gscale) {
    __shared__ float red[4];
    const int HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int bc = blockIdx.y, b = bc / C;
    float part[1] = {0.f};
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        const float* im = img + (size_t)b * 3 * HW;
        const float* pr = pred + (size_t)bc * HW;
        const float v = pr[p];
        float g = 0.f;
        if (y + 1 < H) {                      // term (y, x) of the H-direction sum
            const float d = v - pr[p + W];
            const float w = edge_w(im, HW, p, p + W);
            part[0] += fabsf(d) * w * inv_nx;
            g += sgn(d) * w * inv_nx;
        }
        if (y > 0) {                          // this pixel is the "+1" operand of term (y-1, x)
            const float d = pr[p - W] - v;
            g -= sgn(d) * edge_w(im, HW, p - W, p) * inv_nx;
        }
        if (x + 1 < W) {
            const float d = v - pr[p + 1];
            const float w = edge_w(im, HW, p, p + 1);
            part[0] += fabsf(d) * w * inv_ny;
            g += sgn(d) * w * inv_ny;
        }
        if (x > 0) {
            const float d = pr[p - 1] - v;
            g -= sgn(d) * edge_w(im, HW, p - 1, p) * inv_ny;
        }
        if (gpred) gpred[(size_t)bc * HW + p] = g * gscale;
    }
    cc::block_sum_256<1>(part, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = part[0];
}

// ------------------------------------------------------------------ second-order smoothness
// loss_functions.py:323-341 smooth_loss (per scale, times weight): mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2|
//   dx2(y,x)  = p(y,x+2) - 2p(y,x+1) + p(y,x)          over [H, W-2]
//   dxdy(y,x) = p(y+1,x+1) - p(y+1,x) - p(y,x+1) + p(y,x) over [H-1, W-1]   (dydx is the same expression)
//   dy2(y,x)  = p(y+2,x) - 2p(y+1,x) + p(y,x)          over [H-2, W]
struct Plane {
    const float* p; int H, W;
    __device__ __forceinline__ float at(int y, int x) const { return p[y * W + x]; }
    __device__ __forceinline__ float dx2(int y, int x) const { return at(y, x + 2) - 2.f * at(y, x + 1) + at(y, x); }
    __device__ __forceinline__ float dy2(int y, int x) const { return at(y + 2, x) - 2.f * at(y + 1, x) + at(y, x); }
    __device__ __forceinline__ float dxy(int y, int x) const { return (at(y + 1, x + 1) - at(y + 1, x)) - (at(y, x + 1) - at(y, x)); }
};

__global__ __launch_bounds__(256) void k_smooth2(const float* __restrict__ pred, float* __restrict__ gpred,
                                                 float* __restrict__ partials, int H, int W, float c_dx2, float c_dxy,
                                                 float c_dy2, float gscale) {
    __shared__ float red[4];
    const int HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float part[1] = {0.f};
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        Plane P{pred + (size_t)blockIdx.y * HW, H, W};
        float g = 0.f;
        // forward terms anchored at this pixel
        if (x + 2 < W) part[0] += fabsf(P.dx2(y, x)) * c_dx2;
        if (y + 2 < H) part[0] += fabsf(P.dy2(y, x)) * c_dy2;
        if (x + 1 < W && y + 1 < H) part[0] += fabsf(P.dxy(y, x)) * c_dxy;     // c_dxy already counts dxdy + dydx
        // gradient: every stencil this pixel takes part in
        for (int k = 0; k < 3; k++) {          // dx2 anchored at x-k, coefficient {1,-2,1}[k]
            const int xa = x - k;
            if (xa >= 0 && xa + 2 < W) g += sgn(P.dx2(y, xa)) * ((k == 1) ? -2.f : 1.f) * c_dx2;
            const int ya = y - k;
            if (ya >= 0 && ya + 2 < H) g += sgn(P.dy2(ya, x)) * ((k == 1) ? -2.f : 1.f) * c_dy2;
        }
        for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) {   // dxy anchored at (y-dy, x-dx): coefficient +1 if dy==dx else -1
                const int ya = y - dy, xa = x - dx;
                if (ya >= 0 && xa >= 0 && ya + 1 < H && xa + 1 < W)
                    g += sgn(P.dxy(ya, xa)) * ((dy == dx) ? 1.f : -1.f) * c_dxy;
            }
        if (gpred) gpred[(size_t)blockIdx.y * HW + p] = g * gscale;
    }
    cc::block_sum_256<1>(part, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = part[0];
}

// ------------------------------------------------------------------ BCE losses
// loss_functions.py:148-155 explainability_loss: F.binary_cross_entropy(mask, 1) = mean(-max(log(mask), -100))
__global__ __launch_bounds__(256) void k_bce_ones(const float* __restrict__ mask, float* __restrict__ gmask,
                                                  float* __restrict__ partials, int n, float inv_n, float gscale) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float part[1] = {0.f};
    if (i < n) {
        const float m = mask[i];
        const float lg = fmaxf(logf(m), -100.f);
        part[0] = -lg * inv_n;
        // ATen binary_cross_entropy_backward: (x - y) / max((1 - x) * x, 1e-12), y = 1
        if (gmask) gmask[i] = ((m - 1.f) / fmaxf((1.f - m) * m, 1e-12f)) * inv_n * gscale;
    }
    cc::block_sum_256<1>(part, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = part[0];
}

// loss_functions.py:221-261 consensus_depth_flow_mask, one scale (census_* = |cam_flow - flow|, train.py:475-476):
//   census_d = prod_c(census_mask_d < THRESH)  OR  exp_target_d                 (d in {bwd, fwd})
//   target   = (bwd, bwd, fwd, fwd);  loss = -mean(w1*t*log(e+eps) + w0*(1-t)*log(1-e+eps)), w=[wbce, 1-wbce]
__global__ __launch_bounds__(256) void k_consensus_bce(const float* __restrict__ exp_mask, const float* __restrict__ census_bwd,
                                                       const float* __restrict__ census_fwd, const float* __restrict__ tgt_bwd,
                                                       const float* __restrict__ tgt_fwd, float* __restrict__ gmask,
                                                       float* __restrict__ partials, int HW, float thresh, float w0,
                                                       float w1, float inv_n, float gscale) {
    __shared__ float red[4];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    float part[1] = {0.f};
    if (p < HW) {
        const size_t f = (size_t)b * 2 * HW + p;
        const float cb = (census_bwd[f] < thresh && census_bwd[f + HW] < thresh) ? 1.f : 0.f;
        const float cf = (census_fwd[f] < thresh && census_fwd[f + HW] < thresh) ? 1.f : 0.f;
        const float tb = 1.f - (1.f - cb) * (1.f - tgt_bwd[(size_t)b * HW + p]);
        const float tf = 1.f - (1.f - cf) * (1.f - tgt_fwd[(size_t)b * HW + p]);
        const float t[4] = {tb, tb, tf, tf};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const size_t o = ((size_t)b * 4 + c) * HW + p;
            const float e = exp_mask[o];
            const float a1 = e + 1e-8f, a0 = (1.f - e) + 1e-8f;
            part[0] -= (w1 * (t[c] * logf(a1)) + w0 * ((1.f - t[c]) * logf(a0))) * inv_n;
            if (gmask) gmask[o] = -(w1 * t[c] / a1 - w0 * (1.f - t[c]) / a0) * inv_n * gscale;
        }
    }
    cc::block_sum_256<1>(part, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = part[0];
}

// out = a * s   (device scalar s) -- used to apply grad_output to stashed gradients without a host sync
// out[b][e] (=, +=) src_0[b][e] + src_1[b][e] + ... (fixed order): the gradient of a tensor with several consumers, whose parts
// are channel slices of different buffers (own batch stride each), summed in ONE launch instead of n - 1 adds.
constexpr int SUMN_MAX = 8;
struct SumN { const float* src[SUMN_MAX]; long bs[SUMN_MAX]; int n; };
template <bool VEC4>
__global__ __launch_bounds__(256) void k_sum_strided(SumN t, float* __restrict__ out, long out_bs, long chw, int accumulate) {
    const int b = blockIdx.y;
    if (VEC4) {
        const long q = (long)blockIdx.x * 256 + threadIdx.x;
        if (q * 4 >= chw) return;
        float4 v[SUMN_MAX];
#pragma unroll
        for (int k = 0; k < SUMN_MAX; k++)
            v[k] = (k < t.n) ? ((const float4*)(t.src[k] + (long)b * t.bs[k]))[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 a = v[0];
#pragma unroll
        for (int k = 1; k < SUMN_MAX; k++)
            if (k < t.n) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
        float4* o = (float4*)(out + (long)b * out_bs) + q;
        if (accumulate) { const float4 g = *o; a.x = g.x + a.x; a.y = g.y + a.y; a.z = g.z + a.z; a.w = g.w + a.w; }
        *o = a;
    } else {
        const long e = (long)blockIdx.x * 256 + threadIdx.x;
        if (e >= chw) return;
        float a = t.src[0][(long)b * t.bs[0] + e];
        for (int k = 1; k < t.n; k++) a += t.src[k][(long)b * t.bs[k] + e];
        float* o = out + (long)b * out_bs + e;
        *o = accumulate ? (*o + a) : a;
    }
}

__global__ __launch_bounds__(256) void k_scale_by_scalar(const float* __restrict__ a, const float* __restrict__ s,
                                                         float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * s[0];
}

// dst_k (=, +=) src_k * s[0] for up to SA_MAX spans in one launch (cc_scale_acc_jobs): the `* grad_output` of a fused loss written
// straight into the step's per-tensor gradient accumulators -- float4 where both spans are 16-byte aligned
constexpr int SA_MAX = 32;
struct ScaleAccTab { const float* src[SA_MAX]; float* dst[SA_MAX]; int n[SA_MAX]; int acc[SA_MAX]; int blk_end[SA_MAX]; int njobs; };
__global__ __launch_bounds__(256) void k_scale_acc_jobs(ScaleAccTab t, const float* __restrict__ s) {
    int j = 0, first = 0;
#pragma unroll 1
    for (int q = 0; q + 1 < t.njobs; q++)
        if ((int)blockIdx.x >= t.blk_end[q]) { j = q + 1; first = t.blk_end[q]; }
    const float sc = s[0];
    const float* __restrict__ src = t.src[j];
    float* __restrict__ dst = t.dst[j];
    const int n = t.n[j], acc = t.acc[j];
    const int i4 = (((int)blockIdx.x - first) * 256 + (int)threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(src + i4);
        float4 o = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        if (acc) { const float4 d = *reinterpret_cast<const float4*>(dst + i4); o.x = d.x + o.x; o.y = d.y + o.y; o.z = d.z + o.z; o.w = d.w + o.w; }
        *reinterpret_cast<float4*>(dst + i4) = o;
        return;
    }
    for (int i = i4; i < n && i < i4 + 4; i++) {
        const float o = src[i] * sc;
        dst[i] = acc ? dst[i] + o : o;
    }
}

// ------------------------------------------------------------------ job-table forms (all pyramid levels in one launch, jobs.h)
#define CC_JOB_PIXEL(t, j, b, p, HW)                                        \
    int first__;                                                            \
    const int j = ccjobs::find(t, (int)blockIdx.x, first__);                \
    const int HW = t.H[j] * t.W[j], nb__ = (HW + 255) >> 8;                 \
    const int local__ = (int)blockIdx.x - first__;                          \
    const int b = local__ / nb__, p = (local__ - b * nb__) * 256 + (int)threadIdx.x;

// occlusion_masks per level: slots 0 flow_bw [B,2,H,W], 1 flow_fw, 2 out [B,1,H,W]
__global__ __launch_bounds__(256) void k_flow_noocc_jobs(JobTab t) {
    CC_JOB_PIXEL(t, j, b, p, HW)
    if (p >= HW) return;
    const float* fb = ccjobs::ptr<const float>(t, j, 0) + (size_t)b * 2 * HW + p;
    const float* ff = ccjobs::ptr<const float>(t, j, 1) + (size_t)b * 2 * HW + p;
    ccjobs::ptr<float>(t, j, 2)[(size_t)b * HW + p] = noocc(fb[0], fb[HW], ff[0], ff[HW]);
}

// consensus target per level: slots 0 err_cam_fwd, 1 err_cam_bwd, 2 err_flow_fwd, 3 valid_cam_fwd, 4 valid_cam_bwd, 5 target
__global__ __launch_bounds__(256) void k_consensus_combine_jobs(JobTab t, float wrig) {
    CC_JOB_PIXEL(t, j, b, p, HW)
    if (p >= HW) return;
    const size_t i = (size_t)b * HW + p;
    const float vcf = ccjobs::ptr<const float>(t, j, 3)[i], vcb = ccjobs::ptr<const float>(t, j, 4)[i];
    const float valid = 1.f - (1.f - vcf) * (1.f - vcb);
    const float cam_err = fminf(ccjobs::ptr<const float>(t, j, 0)[i], ccjobs::ptr<const float>(t, j, 1)[i]) * valid;
    ccjobs::ptr<float>(t, j, 5)[i] = (wrig * cam_err <= (ccjobs::ptr<const float>(t, j, 2)[i] + 1e-8f)) ? 1.f : 0.f;
}

// end of a photometric loss's gradient pass, per level: gdepth = sum over reference frames of the per-frame depth gradients
// (reference order r = 0..R-1), gmask[:, c] *= scale[c] (the per-term normalisers known only after the reduction).
// slots 0 gd_all [R][B][HW] (or 0), 1 gdepth [B][HW], 2 gmask [B][MC][HW] (or 0), 3 scales (MC floats)
__global__ __launch_bounds__(256) void k_sum_refs_scale_jobs(JobTab t, int R, int MC) {
    CC_JOB_PIXEL(t, j, b, p, HW)
    if (p >= HW) return;
    const float* gd = ccjobs::ptr<const float>(t, j, 0);
    if (gd) {
        float s = gd[(size_t)b * HW + p];
        for (int r = 1; r < R; r++) s += gd[((size_t)r * t.B + b) * HW + p];
        ccjobs::ptr<float>(t, j, 1)[(size_t)b * HW + p] = s;
    }
    float* gm = ccjobs::ptr<float>(t, j, 2);
    if (gm) {
        const float* sc = ccjobs::ptr<const float>(t, j, 3);
        for (int c = 0; c < MC; c++) gm[((size_t)b * MC + c) * HW + p] *= sc[c];
    }
}

// deterministic finalize shared by the per-level losses below: accum[0] += coef * sum(partials[0..n))  (one workgroup)
// (the partials of all levels are contiguous, level after level, block after block)

// edge-aware smoothness, all levels: slots 0 img level [B,3,H,W], 1 pred [B,C,H,W], 2 gpred (or 0), 3 partials of this job
// table B = batch * C (one (image, channel) plane per "batch item")
// the edge weights of one pyramid level, once per (image, scale): slots 0 img level [B,3,H,W], 1 out [B,2,H,W] = (w(p, p + 1),
// w(p, p + W)), 0 beyond the last column / row.  The smoothness terms of a step (depth, two flows, the 4-channel masks: 9 planes per
// image and scale) share them instead of evaluating exp(-|dI| / 3) four times per pixel and plane (same expression, same bits).
__global__ __launch_bounds__(256) void k_edge_weights_jobs(JobTab t) {
    CC_JOB_PIXEL(t, j, b, p, HW)
    if (p >= HW) return;
    const int W = t.W[j], H = t.H[j];
    const int y = p / W, x = p - y * W;
    const float* im = ccjobs::ptr<const float>(t, j, 0) + (size_t)b * 3 * HW;
    float* o = ccjobs::ptr<float>(t, j, 1) + (size_t)b * 2 * HW;
    o[p] = (x + 1 < W) ? edge_w(im, HW, p, p + 1) : 0.f;
    o[HW + p] = (y + 1 < H) ? edge_w(im, HW, p, p + W) : 0.f;
}

// C == 0: the jobs of SEVERAL terms in one launch (train.py:497-501: depth, flow_fwd, flow_bwd, exp_mask -- 1 / 2 / 2 / 4 channels);
// slot 4 then holds the job's channel count and t.B the batch size
// PPT pixels per work-item (PPT = 4 in the merged form: 24 jobs x 9 planes are 40 k workgroups of 256 pixels, and the launch was
// bound by per-workgroup fixed cost -- job search, index arithmetic, block reduction -- not by its 88 MB: 99 us)
template <int PPT>
__global__ __launch_bounds__(256) void k_edge_smooth_jobs(JobTab t, int C, float gscale) {
    __shared__ float red[4];
    int local__;
    const int j = ccjobs::find_xcd(t, (int)blockIdx.x, local__);
    const int HW = t.H[j] * t.W[j], nb1 = (HW + 255) >> 8, nb__ = (nb1 + PPT - 1) / PPT;
    const int bc = local__ / nb__, blk = local__ - bc * nb__;
    const int Cj = C ? C : (int)t.slot[j][4];
    const int planes = C ? t.B : t.B * Cj;
    const int H = t.H[j], W = t.W[j], b = bc / Cj;
    const float inv_nx = 1.f / ((float)planes * (H - 1) * W), inv_ny = 1.f / ((float)planes * H * (W - 1));
    float part[1] = {0.f};
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const int p = (blk * PPT + k) * 256 + (int)threadIdx.x;
      if (p < HW) {
        const int y = p / W, x = p - y * W;
        const float* im = ccjobs::ptr<const float>(t, j, 0) + (size_t)b * 3 * HW;
        const float* pr = ccjobs::ptr<const float>(t, j, 1) + (size_t)bc * HW;
        float* gpred = ccjobs::ptr<float>(t, j, 2);
        // slot 5 (optional): the level's edge weights [B,2,H,W] from k_edge_weights_jobs -- four loads instead of 24 + four exp()
        const float* wg = t.slot[j][5] ? ccjobs::ptr<const float>(t, j, 5) + (size_t)b * 2 * HW : nullptr;
        const float v = pr[p];
        float g = 0.f;
        if (y + 1 < H) {
            const float d = v - pr[p + W];
            const float w = wg ? wg[HW + p] : edge_w(im, HW, p, p + W);
            part[0] += fabsf(d) * w * inv_nx;
            g += sgn(d) * w * inv_nx;
        }
        if (y > 0) {
            const float d = pr[p - W] - v;
            g -= sgn(d) * (wg ? wg[HW + p - W] : edge_w(im, HW, p - W, p)) * inv_nx;
        }
        if (x + 1 < W) {
            const float d = v - pr[p + 1];
            const float w = wg ? wg[p] : edge_w(im, HW, p, p + 1);
            part[0] += fabsf(d) * w * inv_ny;
            g += sgn(d) * w * inv_ny;
        }
        if (x > 0) {
            const float d = pr[p - 1] - v;
            g -= sgn(d) * (wg ? wg[p - 1] : edge_w(im, HW, p - 1, p)) * inv_ny;
        }
        if (gpred) gpred[(size_t)bc * HW + p] = g * gscale;
      }
    }
    cc::block_sum_256<1>(part, red);
    // the partial areas keep their 256-pixel granularity (the host lays them out back to back per job and sums them all): this
    // workgroup's sum goes to the first of its PPT places, zeros to the others
    if ((int)threadIdx.x < PPT) {
        const int idx = blk * PPT + (int)threadIdx.x;
        if (idx < nb1) ccjobs::ptr<float>(t, j, 3)[(size_t)bc * nb1 + idx] = threadIdx.x == 0 ? part[0] : 0.f;
    }
}

// explainability BCE, all levels: slots 0 mask (n = B*C*H*W elements: table B = batch * C), 1 gmask (or 0), 2 partials
__global__ __launch_bounds__(256) void k_bce_ones_jobs(JobTab t, float gscale) {
    __shared__ float red[4];
    CC_JOB_PIXEL(t, j, bc, p, HW)
    const float inv_n = 1.f / ((float)t.B * HW);
    float part[1] = {0.f};
    if (p < HW) {
        const size_t i = (size_t)bc * HW + p;
        const float m = ccjobs::ptr<const float>(t, j, 0)[i];
        const float lg = fmaxf(logf(m), -100.f);
        part[0] = -lg * inv_n;
        float* gmask = ccjobs::ptr<float>(t, j, 1);
        if (gmask) gmask[i] = ((m - 1.f) / fmaxf((1.f - m) * m, 1e-12f)) * inv_n * gscale;
    }
    cc::block_sum_256<1>(part, red);
    if (threadIdx.x == 0) ccjobs::ptr<float>(t, j, 2)[local__] = part[0];
}

// consensus weighted BCE, all levels: slots 0 exp_mask [B,4,H,W], 1 census_bwd [B,2,H,W], 2 census_fwd, 3 tgt_bwd [B,1,H,W],
// 4 tgt_fwd, 5 gmask (or 0), 6 partials
__global__ __launch_bounds__(256) void k_consensus_bce_jobs(JobTab t, float thresh, float w0, float w1, float gscale) {
    __shared__ float red[4];
    CC_JOB_PIXEL(t, j, b, p, HW)
    const float inv_n = 1.f / ((float)t.B * 4 * HW);
    float part[1] = {0.f};
    if (p < HW) {
        const float* exp_mask = ccjobs::ptr<const float>(t, j, 0);
        const float* census_bwd = ccjobs::ptr<const float>(t, j, 1);
        const float* census_fwd = ccjobs::ptr<const float>(t, j, 2);
        float* gmask = ccjobs::ptr<float>(t, j, 5);
        const size_t f = (size_t)b * 2 * HW + p;
        const float cb = (census_bwd[f] < thresh && census_bwd[f + HW] < thresh) ? 1.f : 0.f;
        const float cf = (census_fwd[f] < thresh && census_fwd[f + HW] < thresh) ? 1.f : 0.f;
        const float tb = 1.f - (1.f - cb) * (1.f - ccjobs::ptr<const float>(t, j, 3)[(size_t)b * HW + p]);
        const float tf = 1.f - (1.f - cf) * (1.f - ccjobs::ptr<const float>(t, j, 4)[(size_t)b * HW + p]);
        const float tg[4] = {tb, tb, tf, tf};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const size_t o = ((size_t)b * 4 + c) * HW + p;
            const float e = exp_mask[o];
            const float a1 = e + 1e-8f, a0 = (1.f - e) + 1e-8f;
            part[0] -= (w1 * (tg[c] * logf(a1)) + w0 * ((1.f - tg[c]) * logf(a0))) * inv_n;
            if (gmask) gmask[o] = -(w1 * tg[c] / a1 - w0 * (1.f - tg[c]) / a0) * inv_n * gscale;
        }
    }
    cc::block_sum_256<1>(part, red);
    if (threadIdx.x == 0) ccjobs::ptr<float>(t, j, 6)[local__] = part[0];
}

// small element-wise glue of train.py:458,475-476,488 over all pyramid levels in one launch (table B = planes per job):
//   op 0: out = 1 / a                      slots a, out            (depth = 1 / disparity, :458)
//   op 1: ga = -g * (y * y)                slots g, y, ga          (its backward; y = the forward result)
//   op 2: out = |a - b|                    slots a, b, out         (rigidity masks |flow_cam - flow|, :475-476)
//   op 3: out[b, c] = 1 - m[b, c0 + c]     slots m, out            (flow_exp_mask = 1 - exp_mask[:, 1:3], :488; planes = B * nc)
//   op 4: gm[b, c] = -g[b, c - c0] inside [c0, c0 + nc), 0 outside   slots g, gm   (its backward; planes = B * MC)
//   op 5: out = ((a * 0.5 + 0.5) - mean_c) / std_c, ImageNet statistics, c = plane % 3   slots a, out
//         (Back2Future.normalize, models/back2future.py:118-132: three images of one step in one launch)
__global__ __launch_bounds__(256) void k_elementwise_jobs(JobTab t, int op, int c0, int nc, int MC) {
    CC_JOB_PIXEL(t, j, q, p, HW)
    if (p >= HW) return;
    const size_t i = (size_t)q * HW + p;
    if (op == 0) {
        ccjobs::ptr<float>(t, j, 1)[i] = 1.0f / ccjobs::ptr<const float>(t, j, 0)[i];
    } else if (op == 1) {
        const float y = ccjobs::ptr<const float>(t, j, 1)[i];
        ccjobs::ptr<float>(t, j, 2)[i] = -ccjobs::ptr<const float>(t, j, 0)[i] * (y * y);
    } else if (op == 2) {
        ccjobs::ptr<float>(t, j, 2)[i] = fabsf(ccjobs::ptr<const float>(t, j, 0)[i] - ccjobs::ptr<const float>(t, j, 1)[i]);
    } else if (op == 5) {
        const int c = q % 3;
        const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
        ccjobs::ptr<float>(t, j, 1)[i] = ((ccjobs::ptr<const float>(t, j, 0)[i] * 0.5f + 0.5f) - mean) / sd;
    } else if (op == 3) {
        const int b = q / nc, c = q - b * nc;
        ccjobs::ptr<float>(t, j, 1)[i] = 1.0f - ccjobs::ptr<const float>(t, j, 0)[((size_t)b * MC + c0 + c) * HW + p];
    } else {
        const int b = q / MC, c = q - b * MC;
        const bool in = (c >= c0) && (c < c0 + nc);
        ccjobs::ptr<float>(t, j, 1)[i] = in ? -ccjobs::ptr<const float>(t, j, 0)[((size_t)b * nc + (c - c0)) * HW + p] : 0.f;
    }
}

}  // namespace

extern "C" {

int cc_adaptive_avg_pool(const float* in, float* out, int planes, int H, int W, int h, int w, void* stream) {
    if (planes <= 0 || h <= 0 || w <= 0 || H < h || W < w) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_adaptive_pool, dim3((h * w + 255) / 256, planes), dim3(256), 0, (hipStream_t)stream, in, out, H, W,
                       h, w, planes);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

// levels 1..nlevels-1 of the 2^l box-mean pyramid of `planes` H x W images, each computed from level 0
// (as the reference does); out_packed holds the levels back to back: [planes, H>>1, W>>1], [planes, H>>2, W>>2] ...
// H, W multiples of 32 and <= 6 levels: ONE launch, every 32x32 tile of level 0 is read once into LDS and all its
// descendants are produced from it, each window summed in the reference's row-major order (bit-identical to the
// per-level kernel); otherwise one k_adaptive_pool launch per level.
int cc_pyramid_build(const float* level0, float* out_packed, int nlevels, int planes, int H, int W, void* stream) {
    if (nlevels < 1 || planes <= 0) return CC_ERR_ARG;
    if (nlevels >= 2 && nlevels <= 6 && H % 32 == 0 && W % 32 == 0 && ((uintptr_t)level0 % 16) == 0) {
        hipLaunchKernelGGL(k_pyramid_tile, dim3((unsigned)((H / 32) * (W / 32)), (unsigned)planes), dim3(256), 0,
                           (hipStream_t)stream, level0, out_packed, nlevels, planes, H, W);
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    size_t off = 0;
    for (int l = 1; l < nlevels; l++) {
        const int h = H >> l, w = W >> l;
        if (h < 1 || w < 1) return CC_ERR_ARG;
        hipLaunchKernelGGL(k_adaptive_pool, dim3((h * w + 255) / 256, planes), dim3(256), 0, (hipStream_t)stream, level0,
                           out_packed + off, H, W, h, w, planes);
        off += (size_t)planes * h * w;
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_pyramid_build_multi(const long* level0_host, const long* out_packed_host, int nimg, int nlevels, int planes, int H, int W,
                           void* stream) {
    if (!level0_host || !out_packed_host || nimg < 1 || nimg > PYR_MAXIMG || nlevels < 2 || nlevels > 6 || planes <= 0 || H % 32 != 0 ||
        W % 32 != 0 || H <= 0 || W <= 0)
        return CC_ERR_ARG;
    PyrMulti t = {};
    for (int i = 0; i < nimg; i++) {
        if (level0_host[i] % 16 != 0 || !out_packed_host[i]) return CC_ERR_ARG;
        t.in[i] = reinterpret_cast<const float*>(level0_host[i]);
        t.out[i] = reinterpret_cast<float*>(out_packed_host[i]);
    }
    hipLaunchKernelGGL(k_pyramid_tile_multi, dim3((unsigned)((H / 32) * (W / 32)), (unsigned)planes, (unsigned)nimg), dim3(256), 0,
                       (hipStream_t)stream, t, nlevels, planes, H, W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_flow_noocc(const float* flow_bw, const float* flow_fw, float* out, int B, int H, int W, void* stream) {
    if (B <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_flow_noocc, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, flow_bw, flow_fw, out,
                       H * W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_rigid_noocc(const float* flows4, float* out, int B, int H, int W, void* stream) {
    if (B <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_rigid_noocc, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, flows4, out, B,
                       H * W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_reduce_add(const float* partials, int n, float coef, float* accum, void* stream) {
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n, coef, accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

size_t cc_elem_num_blocks(int n) { return (n + 255) / 256; }

int cc_edge_smooth_fwd_bwd(const float* img, const float* pred, float* gpred_or_null, float* partials, float* loss_accum,
                           float gscale, int B, int C, int H, int W, void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (H * W + 255) / 256;
    const float inv_nx = 1.f / ((float)B * C * (H - 1) * W), inv_ny = 1.f / ((float)B * C * H * (W - 1));
    hipLaunchKernelGGL(k_edge_smooth, dim3(nb, B * C), dim3(256), 0, s, img, pred, gpred_or_null, partials, C, H, W, inv_nx,
                       inv_ny, gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, nb * B * C, 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_smooth2_fwd_bwd(const float* pred, float* gpred_or_null, float* partials, float* loss_accum, float weight,
                       float gscale, int planes, int H, int W, void* stream) {
    if (planes <= 0 || H < 1 || W < 1) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (H * W + 255) / 256;
    // a map thinner than 3 pixels makes one of the reference's .mean() calls run over an empty tensor -> NaN
    const bool degenerate = (H < 3 || W < 3);
    const float c_dx2 = (W > 2) ? weight / ((float)planes * H * (W - 2)) : 0.f;
    const float c_dy2 = (H > 2) ? weight / ((float)planes * (H - 2) * W) : 0.f;
    const float c_dxy = (H > 1 && W > 1) ? 2.f * weight / ((float)planes * (H - 1) * (W - 1)) : 0.f;
    hipLaunchKernelGGL(k_smooth2, dim3(nb, planes), dim3(256), 0, s, pred, gpred_or_null, partials, H, W, c_dx2, c_dxy,
                       c_dy2, gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, nb * planes,
                       degenerate ? __builtin_nanf("") : 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_bce_ones_fwd_bwd(const float* mask, float* gmask_or_null, float* partials, float* loss_accum, float gscale, int n,
                        void* stream) {
    if (n <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_bce_ones, dim3(nb), dim3(256), 0, s, mask, gmask_or_null, partials, n, 1.f / (float)n, gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, nb, 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_consensus_bce_fwd_bwd(const float* exp_mask, const float* census_bwd, const float* census_fwd,
                             const float* target_bwd, const float* target_fwd, float* gmask_or_null, float* partials,
                             float* loss_accum, float thresh, float wbce, float gscale, int B, int H, int W,
                             void* stream) {
    if (B <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W, nb = (HW + 255) / 256;
    // weights = [wbce, 1 - wbce]: weights[1] multiplies the target term, weights[0] the (1 - target) term
    hipLaunchKernelGGL(k_consensus_bce, dim3(nb, B), dim3(256), 0, s, exp_mask, census_bwd, census_fwd, target_bwd,
                       target_fwd, gmask_or_null, partials, HW, thresh, wbce, 1.f - wbce, 1.f / ((float)B * 4 * HW),
                       gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, nb * B, 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_sum_strided(int n, const long* src, const long* src_bs, float* out, long out_bs, int B, long chw, int accumulate,
                   void* stream) {
    if (n <= 0 || n > SUMN_MAX || !src || !src_bs || !out || B <= 0 || chw <= 0) return CC_ERR_ARG;
    SumN t = {};
    t.n = n;
    bool vec4 = (chw % 4 == 0) && (out_bs % 4 == 0) && (((uintptr_t)out) % 16 == 0);
    for (int k = 0; k < n; k++) {
        if (!src[k]) return CC_ERR_ARG;
        t.src[k] = (const float*)src[k];
        t.bs[k] = src_bs[k];
        vec4 = vec4 && (src_bs[k] % 4 == 0) && (((uintptr_t)src[k]) % 16 == 0);
    }
    hipStream_t s = (hipStream_t)stream;
    if (vec4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sum_strided<true>), dim3((unsigned)((chw / 4 + 255) / 256), (unsigned)B), dim3(256), 0, s, t, out, out_bs, chw, accumulate);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sum_strided<false>), dim3((unsigned)((chw + 255) / 256), (unsigned)B), dim3(256), 0, s, t, out, out_bs, chw, accumulate);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_scale_by_scalar(const float* a, const float* scalar_dev, float* out, int n, void* stream) {
    if (n <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_scale_by_scalar, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, scalar_dev, out, n);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_scale_acc_jobs(const long* jobs_host, int njobs, const float* scalar_dev, void* stream) {
    if (!jobs_host || njobs <= 0 || !scalar_dev) return CC_ERR_ARG;
    for (int j0 = 0; j0 < njobs; j0 += SA_MAX) {
        ScaleAccTab t = {};
        long blk = 0;
        const int nj = njobs - j0 < SA_MAX ? njobs - j0 : SA_MAX;
        for (int k = 0; k < nj; k++) {
            const long* d = jobs_host + 4l * (j0 + k);
            if (!d[0] || !d[1] || d[2] <= 0 || d[2] >= (1l << 31)) return CC_ERR_ARG;
            t.src[k] = (const float*)d[0]; t.dst[k] = (float*)d[1]; t.n[k] = (int)d[2]; t.acc[k] = d[3] ? 1 : 0;
            blk += (d[2] + 1023) / 1024;
            if (blk >= (1l << 31)) return CC_ERR_ARG;
            t.blk_end[k] = (int)blk;
        }
        t.njobs = nj;
        hipLaunchKernelGGL(k_scale_acc_jobs, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, t, scalar_dev);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* ---- job-table forms (jobs: HOST array of njobs x 10 longs {slot0..7, H, W}; njobs <= 24; slots per kernel above).
 * The *_fwd_bwd_jobs losses write one partial sum per block into the per-job partial areas, which the caller lays out
 * back to back (job after job) starting at `partials`; one finalize launch adds their sum to loss_accum. */
static int loss_jobs_tab(ccjobs::JobTab& t, const long* jobs, int njobs, int B) {
    if (!jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return -1;
    return ccjobs::fill(t, jobs, njobs, B, ccjobs::pix_blocks);
}

int cc_flow_noocc_jobs(const long* jobs, int njobs, int B, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_flow_noocc_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_consensus_target_jobs(const long* jobs, int njobs, int B, float wrig, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_consensus_combine_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t, wrig);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_sum_refs_scale_jobs(const long* jobs, int njobs, int B, int R, int MC, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_sum_refs_scale_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t, R, MC);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_elementwise_jobs(const long* jobs, int njobs, int planes, int op, int c0, int nc, int MC, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, planes);
    if (nblk <= 0 || op < 0 || op > 5) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_elementwise_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t, op, c0, nc, MC);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

size_t cc_loss_jobs_num_blocks(const long* jobs, int njobs, int planes) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, planes);
    return nblk > 0 ? (size_t)nblk : 0;
}

int cc_edge_smooth_fwd_bwd_jobs(const long* jobs, int njobs, int B, int C, float* partials, float* loss_accum, float gscale,
                                void* stream) {
    ccjobs::JobTab t;
    int nblk, npart = -1;
    if (C > 0) {
        nblk = loss_jobs_tab(t, jobs, njobs, B * C);
        npart = nblk;
    } else {
        // per-job channel counts (slot 4): job j owns B * C_j * ceil(H W / 256) blocks
        if (C < 0 || !jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return CC_ERR_ARG;
        nblk = ccjobs::fill(t, jobs, njobs, B, ccjobs::pix_blocks);
        int tot = 0;
        npart = 0;
        for (int j = 0; j < njobs; j++) {
            const int cj = (int)t.slot[j][4];
            if (cj <= 0) return CC_ERR_ARG;
            const int nb1 = ccjobs::pix_blocks(t.H[j], t.W[j]);
            tot += B * cj * ((nb1 + 3) / 4);          // four 256-pixel pieces per workgroup
            npart += B * cj * nb1;                    // partial sums keep the 256-pixel layout
            t.blk_end[j] = tot;
        }
        nblk = tot;
    }
    if (nblk <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (C > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_smooth_jobs<1>), dim3((unsigned)nblk), dim3(256), 0, s, t, C, gscale);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_smooth_jobs<4>), dim3((unsigned)nblk), dim3(256), 0, s, t, C, gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, npart, 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_edge_weights_jobs(const long* jobs, int njobs, int B, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_edge_weights_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_bce_ones_fwd_bwd_jobs(const long* jobs, int njobs, int planes, float* partials, float* loss_accum, float gscale, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, planes);
    if (nblk <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bce_ones_jobs, dim3((unsigned)nblk), dim3(256), 0, s, t, gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, nblk, 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_consensus_bce_fwd_bwd_jobs(const long* jobs, int njobs, int B, float* partials, float* loss_accum, float thresh, float wbce,
                                  float gscale, void* stream) {
    ccjobs::JobTab t;
    const int nblk = loss_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_consensus_bce_jobs, dim3((unsigned)nblk), dim3(256), 0, s, t, thresh, wbce, 1.f - wbce, gscale);
    hipLaunchKernelGGL(k_reduce_add, dim3(1), dim3(256), 0, s, (const float*)partials, nblk, 1.0f, loss_accum);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
