// Geometry layer of the CC hot path as fused gfx950 kernels:
//   depth -> 3-D point (K^-1) -> SE(3)+project (P = K.[R|t]) -> normalise -> OOB rule
//   -> bilinear gather, in ONE pass per pixel, plus the analytic backward to
//   depth / P (rigid) or flow, with an atomic-free block reduction for dL/dP.
// Replaces inverse_warp.py:31-79,164-220,250-283 and models/back2future.py:287-321
// (reference = a bmm + ~25 elementwise ATen ops + grid_sampler_2d per call).
//
// Coordinate arithmetic follows SURVEY.md appendix D exactly (explicit fmaf where
// the CPU reference's bmm fuses, nowhere else; build with -ffp-contract=off), so
// that tap indices are bit-identical to the reference given identical P / Kinv.
// All kernels are HBM-bound (<= ~10 flop/B): one work-item per pixel, x fastest,
// so depth/flow/out accesses are fully coalesced and the 4-tap gathers hit
// neighbouring lines (L1/L2-resident for the small motions of this workload).
#include "cc_common.h"
#include "jobs.h"
#include "../../include/ccengine.h"

namespace {

using ccjobs::JobTab;

struct Bilinear {
    float w, e, n, s;     // distances to the west/east/north/south tap (ATen cpu/GridSamplerKernel naming)
    int x0, y0;
    bool vx0, vx1, vy0, vy1;
    float gmx, gmy;       // d(ix)/d(xn), d(iy)/d(yn) incl. border-clip mask
};

template <bool AC, bool BORDER>
__device__ __forceinline__ void bilinear_setup(float xn, float yn, int W, int H, Bilinear& t) {
    float ix, iy;
    if (AC) {
        ix = ((xn + 1.f) * 0.5f) * (float)(W - 1);
        iy = ((yn + 1.f) * 0.5f) * (float)(H - 1);
        t.gmx = (float)(W - 1) * 0.5f;
        t.gmy = (float)(H - 1) * 0.5f;
    } else {
        // ATen's vectorised CPU kernel evaluates (x+1)*(W/2) - 0.5 as ONE fma (measured: 100 % of
        // 2e5 random coordinates bit-identical with this form, 77 % without the fma)
        ix = fmaf(xn + 1.f, (float)W * 0.5f, -0.5f);
        iy = fmaf(yn + 1.f, (float)H * 0.5f, -0.5f);
        t.gmx = (float)W * 0.5f;
        t.gmy = (float)H * 0.5f;
    }
    if (BORDER) {
        const float mx = (float)(W - 1), my = (float)(H - 1);
        if (!(ix > 0.f)) { ix = 0.f; t.gmx = 0.f; } else if (!(ix < mx)) { ix = mx; t.gmx = 0.f; }
        if (!(iy > 0.f)) { iy = 0.f; t.gmy = 0.f; } else if (!(iy < my)) { iy = my; t.gmy = 0.f; }
    }
    const float x0f = floorf(ix), y0f = floorf(iy);
    t.w = ix - x0f;
    t.e = 1.f - t.w;
    t.n = iy - y0f;
    t.s = 1.f - t.n;
    t.vx0 = (x0f >= 0.f) && (x0f <= (float)(W - 1));
    t.vx1 = (x0f + 1.f >= 0.f) && (x0f + 1.f <= (float)(W - 1));
    t.vy0 = (y0f >= 0.f) && (y0f <= (float)(H - 1));
    t.vy1 = (y0f + 1.f >= 0.f) && (y0f + 1.f <= (float)(H - 1));
    t.x0 = (int)fminf(fmaxf(x0f, -2.f), (float)W);
    t.y0 = (int)fminf(fmaxf(y0f, -2.f), (float)H);
}

__device__ __forceinline__ void gather4(const float* __restrict__ plane, int W, const Bilinear& t,
                                        float& nw, float& ne, float& sw, float& se) {
    const int o = t.y0 * W + t.x0;
    nw = (t.vy0 && t.vx0) ? plane[o] : 0.f;
    ne = (t.vy0 && t.vx1) ? plane[o + 1] : 0.f;
    sw = (t.vy1 && t.vx0) ? plane[o + W] : 0.f;
    se = (t.vy1 && t.vx1) ? plane[o + W + 1] : 0.f;
}

__device__ __forceinline__ float blend(const Bilinear& t, float nw, float ne, float sw, float se) {
    // same contraction as the CPU reference (bit-identical on 2e5 random samples)
    return fmaf(se, t.n * t.w, fmaf(sw, t.n * t.e, fmaf(ne, t.s * t.w, nw * (t.s * t.e))));
}

// out_c = bilinear sample of plane c for c in [0, C): up to four channels' taps in flight (one channel per iteration is a chain of
// C load latencies; the stores do not wait)
__device__ __forceinline__ void sample_channels(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int W,
                                                const Bilinear& t) {
    for (int c0 = 0; c0 < C; c0 += 4) {
        float nw[4], ne[4], sw[4], se[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            nw[u] = ne[u] = sw[u] = se[u] = 0.f;
            if (c0 + u < C) gather4(src + (size_t)(c0 + u) * HW, W, t, nw[u], ne[u], sw[u], se[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c0 + u < C) dst[(size_t)(c0 + u) * HW] = blend(t, nw[u], ne[u], sw[u], se[u]);
    }
}

struct Rigid {
    float ray[3], cam[3], p0, p1, p2, Z, xn, yn;
    bool xo, yo;   // coordinate rewritten to 2 (zeros mode): no gradient
};

// inverse_warp.py:43-45 (pixel2cam) + :60-76 (cam2pixel); SURVEY.md appendix D recipe.
__device__ __forceinline__ void rigid_project(const float* __restrict__ P, const float* __restrict__ Ki, float x,
                                              float y, float d, int W, int H, bool rewrite, Rigid& r) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        r.ray[i] = fmaf(Ki[3 * i + 2], 1.0f, fmaf(Ki[3 * i + 1], y, Ki[3 * i] * x));
        r.cam[i] = r.ray[i] * d;
    }
    r.p0 = fmaf(P[2], r.cam[2], fmaf(P[1], r.cam[1], P[0] * r.cam[0])) + P[3];
    r.p1 = fmaf(P[6], r.cam[2], fmaf(P[5], r.cam[1], P[4] * r.cam[0])) + P[7];
    r.p2 = fmaf(P[10], r.cam[2], fmaf(P[9], r.cam[1], P[8] * r.cam[0])) + P[11];
    r.Z = fmaxf(r.p2, 1e-3f);
    r.xn = (2.0f * (r.p0 / r.Z)) / (float)(W - 1) - 1.0f;
    r.yn = (2.0f * (r.p1 / r.Z)) / (float)(H - 1) - 1.0f;
    r.xo = rewrite && (r.xn > 1.f || r.xn < -1.f);
    r.yo = rewrite && (r.yn > 1.f || r.yn < -1.f);
    if (r.xo) r.xn = 2.f;
    if (r.yo) r.yn = 2.f;
}

// chain d(loss)/d(xn,yn) back to depth and to the 12 entries of P
__device__ __forceinline__ void rigid_backward(const float* __restrict__ P, const Rigid& r, float gxn, float gyn, int W,
                                               int H, float& gdepth, float (&gP)[12]) {
    if (r.xo) gxn = 0.f;
    if (r.yo) gyn = 0.f;
    const float gu = gxn * (2.0f / (float)(W - 1));
    const float gv = gyn * (2.0f / (float)(H - 1));
    const float iz = 1.0f / r.Z;
    const float gp0 = gu * iz, gp1 = gv * iz;
    const float gz = -(gu * r.p0 + gv * r.p1) * iz * iz;
    const float gp2 = (r.p2 < 1e-3f) ? 0.f : gz;
    const float gp[3] = {gp0, gp1, gp2};
    float gcam[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            gP[4 * i + j] = gp[i] * r.cam[j];
            gcam[j] += gp[i] * P[4 * i + j];
        }
        gP[4 * i + 3] = gp[i];
    }
    gdepth = gcam[0] * r.ray[0] + gcam[1] * r.ray[1] + gcam[2] * r.ray[2];
}

// ------------------------------------------------------------------ rigid (inverse_warp / pose2flow)
template <bool AC, bool BORDER>
__global__ __launch_bounds__(256) void k_inverse_warp_fwd(const float* __restrict__ img, const float* __restrict__ depth,
                                                          const float* __restrict__ P, const float* __restrict__ Kinv,
                                                          float* __restrict__ out, int C, int H, int W) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    Rigid r;
    rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, !BORDER, r);
    Bilinear t;
    bilinear_setup<AC, BORDER>(r.xn, r.yn, W, H, t);
    const float* src = img + (size_t)b * C * HW;
    float* dst = out + (size_t)b * C * HW + p;
    sample_channels(src, dst, C, HW, W, t);
}

// d(sum_c gout_c * sample_c)/d(ix, iy) (ATen grid_sampler_2d_backward, bilinear)
// fx != nullptr: the scatter goes to a 64-bit FIXED-POINT image (contribution * fx_scale rounded to an integer, fx_scale a power
// of two): integer addition is associative, so the sums -- unlike float atomics -- do not depend on the order in which the
// atomics land (config.deterministic, cc_feature_warp_bwd_det)
__device__ __forceinline__ void fx_add(unsigned long long* p, float v, float fx_scale) {
    atomicAdd(p, (unsigned long long)__float2ll_rn(v * fx_scale));
}

__device__ __forceinline__ void sample_grad(const float* __restrict__ src, const float* __restrict__ gout, int C, int HW,
                                            int W, const Bilinear& t, float* __restrict__ gimg, float& gix, float& giy,
                                            unsigned long long* __restrict__ fx = nullptr, float fx_scale = 0.f) {
    gix = 0.f;
    giy = 0.f;
    // up to four channels' loads (4 taps + the output gradient each) in flight, consumed in channel order: with one channel per
    // iteration the loop is a chain of C load latencies (the image warps have C = 3, the feature warps 16-128 per wave)
    for (int c0 = 0; c0 < C; c0 += 4) {
        float nw[4], ne[4], sw[4], se[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            nw[u] = ne[u] = sw[u] = se[u] = g[u] = 0.f;
            if (c0 + u < C) {
                gather4(src + (size_t)(c0 + u) * HW, W, t, nw[u], ne[u], sw[u], se[u]);
                g[u] = gout[(size_t)(c0 + u) * HW];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c0 + u < C) {
                gix += ((ne[u] - nw[u]) * t.s + (se[u] - sw[u]) * t.n) * g[u];
                giy += ((sw[u] - nw[u]) * t.e + (se[u] - ne[u]) * t.w) * g[u];
                if (fx) {
                    unsigned long long* gp = fx + (size_t)(c0 + u) * HW + t.y0 * W + t.x0;
                    if (t.vy0 && t.vx0) fx_add(gp, g[u] * (t.s * t.e), fx_scale);
                    if (t.vy0 && t.vx1) fx_add(gp + 1, g[u] * (t.s * t.w), fx_scale);
                    if (t.vy1 && t.vx0) fx_add(gp + W, g[u] * (t.n * t.e), fx_scale);
                    if (t.vy1 && t.vx1) fx_add(gp + W + 1, g[u] * (t.n * t.w), fx_scale);
                } else if (gimg) {
                    float* gp = gimg + (size_t)(c0 + u) * HW + t.y0 * W + t.x0;
                    if (t.vy0 && t.vx0) atomicAdd(gp, g[u] * (t.s * t.e));
                    if (t.vy0 && t.vx1) atomicAdd(gp + 1, g[u] * (t.s * t.w));
                    if (t.vy1 && t.vx0) atomicAdd(gp + W, g[u] * (t.n * t.e));
                    if (t.vy1 && t.vx1) atomicAdd(gp + W + 1, g[u] * (t.n * t.w));
                }
            }
    }
}

template <bool AC, bool BORDER>
__global__ __launch_bounds__(256) void k_inverse_warp_bwd(const float* __restrict__ gout, const float* __restrict__ img,
                                                          const float* __restrict__ depth, const float* __restrict__ P,
                                                          const float* __restrict__ Kinv, float* __restrict__ gdepth,
                                                          float* __restrict__ gP_part, float* __restrict__ gimg, int C,
                                                          int H, int W) {
    __shared__ float red[4 * 12];
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float gP[12];
#pragma unroll
    for (int i = 0; i < 12; i++) gP[i] = 0.f;
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        Rigid r;
        rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, !BORDER, r);
        Bilinear t;
        bilinear_setup<AC, BORDER>(r.xn, r.yn, W, H, t);
        float gix, giy, gd;
        sample_grad(img + (size_t)b * C * HW, gout + (size_t)b * C * HW + p, C, HW, W, t,
                    gimg ? gimg + (size_t)b * C * HW : nullptr, gix, giy);
        rigid_backward(P + 12 * b, r, gix * t.gmx, giy * t.gmy, W, H, gd, gP);
        gdepth[(size_t)b * HW + p] = gd;
    }
    cc::block_sum_256<12>(gP, red);
    if (threadIdx.x == 0) {
        float* o = gP_part + ((size_t)b * gridDim.x + blockIdx.x) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) o[i] = gP[i];
    }
}

// deterministic second stage of the dL/dP reduction: one wave per batch item
__global__ __launch_bounds__(64) void k_reduce_gP(const float* __restrict__ part, float* __restrict__ gP, int nblk,
                                                  int accumulate) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0.f;
    for (int k = lane; k < nblk; k += 64) {
        const float* s = part + ((size_t)b * nblk + k) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) acc[i] += s[i];
    }
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = cc::wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) gP[12 * b + i] = accumulate ? gP[12 * b + i] + acc[i] : acc[i];
    }
}

__global__ __launch_bounds__(256) void k_pose2flow_fwd(const float* __restrict__ depth, const float* __restrict__ P,
                                                       const float* __restrict__ Kinv, float* __restrict__ flow, int H,
                                                       int W, int rewrite) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    Rigid r;
    rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, rewrite != 0, r);
    // inverse_warp.py:217-218
    flow[((size_t)b * 2 + 0) * HW + p] = (float)(W - 1) * (r.xn / 2.0f + 0.5f) - (float)x;
    flow[((size_t)b * 2 + 1) * HW + p] = (float)(H - 1) * (r.yn / 2.0f + 0.5f) - (float)y;
}

__global__ __launch_bounds__(256) void k_pose2flow_bwd(const float* __restrict__ gflow, const float* __restrict__ depth,
                                                       const float* __restrict__ P, const float* __restrict__ Kinv,
                                                       float* __restrict__ gdepth, float* __restrict__ gP_part, int H,
                                                       int W, int rewrite) {
    __shared__ float red[4 * 12];
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float gP[12];
#pragma unroll
    for (int i = 0; i < 12; i++) gP[i] = 0.f;
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        Rigid r;
        rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, rewrite != 0, r);
        const float gxn = gflow[((size_t)b * 2 + 0) * HW + p] * ((float)(W - 1) * 0.5f);
        const float gyn = gflow[((size_t)b * 2 + 1) * HW + p] * ((float)(H - 1) * 0.5f);
        float gd;
        rigid_backward(P + 12 * b, r, gxn, gyn, W, H, gd, gP);
        gdepth[(size_t)b * HW + p] = gd;
    }
    cc::block_sum_256<12>(gP, red);
    if (threadIdx.x == 0) {
        float* o = gP_part + ((size_t)b * gridDim.x + blockIdx.x) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) o[i] = gP[i];
    }
}

// ------------------------------------------------------------------ flow-driven warps
// FEATURE=false: inverse_warp.py:185-188 flow_warp grid; FEATURE=true: back2future.py:306-307 grid
template <bool FEATURE>
__device__ __forceinline__ void flow_coords(float x, float y, float u, float v, int W, int H, float& xn, float& yn,
                                            float& dxn, float& dyn) {
    if (FEATURE) {
        const float mw = (float)(W - 1 > 1 ? W - 1 : 1), mh = (float)(H - 1 > 1 ? H - 1 : 1);
        xn = (2.0f * (x + u)) / mw - 1.0f;
        yn = (2.0f * (y + v)) / mh - 1.0f;
        dxn = 2.0f / mw;
        dyn = 2.0f / mh;
    } else {
        xn = 2.0f * ((x + u) / ((float)W - 1.0f) - 0.5f);
        yn = 2.0f * ((y + v) / ((float)H - 1.0f) - 0.5f);
        dxn = 2.0f / ((float)W - 1.0f);
        dyn = 2.0f / ((float)H - 1.0f);
    }
}

// fs: constant the flow is multiplied by first (Back2Future warps with `up_flow * 0.625 ...` / its negation, back2future.py:196-285)
template <bool AC, bool BORDER, bool FEATURE>
__global__ __launch_bounds__(256) void k_flow_warp_fwd(const float* __restrict__ img, const float* __restrict__ flow,
                                                       float* __restrict__ out, int C, int H, int W, float fs) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float xn, yn, dxn, dyn;
    flow_coords<FEATURE>((float)x, (float)y, flow[((size_t)b * 2) * HW + p] * fs, flow[((size_t)b * 2 + 1) * HW + p] * fs, W, H, xn,
                         yn, dxn, dyn);
    Bilinear t;
    bilinear_setup<AC, BORDER>(xn, yn, W, H, t);
    const float* src = img + (size_t)b * C * HW;
    float* dst = out + (size_t)b * C * HW + p;
    // gridDim.z > 1: channel groups (many-channel feature maps on small pyramid levels need more than HW work-items)
    const int cper = (C + (int)gridDim.z - 1) / (int)gridDim.z;
    const int c0 = (int)blockIdx.z * cper, c1 = (c0 + cper < C) ? c0 + cper : C;
    if (c0 < c1) sample_channels(src + (size_t)c0 * HW, dst + (size_t)c0 * HW, c1 - c0, HW, W, t);
}

template <bool AC, bool BORDER, bool FEATURE>
__global__ __launch_bounds__(256) void k_flow_warp_bwd(const float* __restrict__ gout, const float* __restrict__ img,
                                                       const float* __restrict__ flow, float* __restrict__ gflow,
                                                       float* __restrict__ gimg, int C, int H, int W, float fs) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float xn, yn, dxn, dyn;
    flow_coords<FEATURE>((float)x, (float)y, flow[((size_t)b * 2) * HW + p] * fs, flow[((size_t)b * 2 + 1) * HW + p] * fs, W, H, xn,
                         yn, dxn, dyn);
    Bilinear t;
    bilinear_setup<AC, BORDER>(xn, yn, W, H, t);
    float gix, giy;
    sample_grad(img + (size_t)b * C * HW, gout + (size_t)b * C * HW + p, C, HW, W, t,
                gimg ? gimg + (size_t)b * C * HW : nullptr, gix, giy);
    if (gflow) {
        gflow[((size_t)b * 2) * HW + p] = gix * t.gmx * dxn * fs;
        gflow[((size_t)b * 2 + 1) * HW + p] = giy * t.gmy * dyn * fs;
    }
}

// Feature-warp backward (Back2Future.warp on 32-128 channel maps): one work-item per pixel walking all C channels leaves
// the chip with < 1 workgroup per CU on the 64x208 level (146 us avg).  Here a workgroup is 64 pixels x 4 channel groups
// (wave w = channels [w*C/4, (w+1)*C/4) of 64 consecutive pixels: coalesced), the four partial flow gradients are summed
// in LDS in a fixed order; the feature gradient is the same float-atomic scatter as the reference's grid_sample backward.
template <bool AC>
__global__ __launch_bounds__(256) void k_feature_warp_bwd4(const float* __restrict__ gout, const float* __restrict__ img,
                                                           const float* __restrict__ flow, float* __restrict__ gflow,
                                                           float* __restrict__ gimg, int C, int H, int W, float fs,
                                                           unsigned long long* __restrict__ fx = nullptr,
                                                           const unsigned* __restrict__ fx_max = nullptr) {
    __shared__ float part[4][64][2];
    // deterministic form: scale = 2^(46 - exponent of max |gout|): the largest contribution uses <= 47 bits, 2^16 of them fit
    float fx_scale = 0.f;
    if (fx) {
        int e = 0;
        (void)frexpf(__uint_as_float(*fx_max), &e);
        fx_scale = ldexpf(1.0f, 46 - e);
    }
    const int b = blockIdx.y, HW = H * W;
    const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const int cper = (C + 3) >> 2;
    const int c0 = cg * cper, c1 = (c0 + cper < C) ? c0 + cper : C;
    float gix = 0.f, giy = 0.f, gmx = 0.f, gmy = 0.f, dxn = 0.f, dyn = 0.f;
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        float xn, yn;
        flow_coords<true>((float)x, (float)y, flow[((size_t)b * 2) * HW + p] * fs, flow[((size_t)b * 2 + 1) * HW + p] * fs, W, H, xn, yn,
                          dxn, dyn);
        Bilinear t;
        bilinear_setup<AC, true>(xn, yn, W, H, t);
        gmx = t.gmx;
        gmy = t.gmy;
        if (c0 < c1)
            sample_grad(img + ((size_t)b * C + c0) * HW, gout + ((size_t)b * C + c0) * HW + p, c1 - c0, HW, W, t,
                        gimg ? gimg + ((size_t)b * C + c0) * HW : nullptr, gix, giy,
                        fx ? fx + ((size_t)b * C + c0) * HW : nullptr, fx_scale);
    }
    part[cg][lane][0] = gix;
    part[cg][lane][1] = giy;
    __syncthreads();
    if (cg == 0 && p < HW && gflow) {
        const float sx = ((part[0][lane][0] + part[1][lane][0]) + part[2][lane][0]) + part[3][lane][0];
        const float sy = ((part[0][lane][1] + part[1][lane][1]) + part[2][lane][1]) + part[3][lane][1];
        gflow[((size_t)b * 2) * HW + p] = sx * gmx * dxn * fs;
        gflow[((size_t)b * 2 + 1) * HW + p] = sy * gmy * dyn * fs;
    }
}

// loss_functions.py:132-137 depth_occlusion_masks in ONE pass: the four rigid flows (pose2flow with the full-resolution K,
// inverse_warp.py:195-220, no OOB rewrite) stay in registers, then occlusion_masks (:343-352) on the pairs (1,2) and (0,3)
// -> (1 - occ) for refs 0..3, [B,4,H,W].  Replaces 4 cc_pose2flow_fwd launches + cc_rigid_noocc and the [4,B,2,H,W] buffer.
__device__ __forceinline__ float noocc_pair(float bu, float bv, float fu, float fv) {
    const float mag = (fu * fu + fv * fv) + (bu * bu + bv * bv);
    const float thr = 0.08f * mag + 1.0f;
    const float s = (fu + bu) + (fv + bv);
    return (s > thr) ? 0.f : 1.f;
}

__global__ __launch_bounds__(256) void k_rigid_noocc_fused(const float* __restrict__ depth, const float* __restrict__ P4,
                                                           const float* __restrict__ Kinv, float* __restrict__ out, int B, int H,
                                                           int W) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float d = depth[(size_t)b * HW + p];
    float u[4], v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        Rigid rg;
        rigid_project(P4 + ((size_t)r * B + b) * 12, Kinv + 9 * b, (float)x, (float)y, d, W, H, false, rg);
        u[r] = (float)(W - 1) * (rg.xn / 2.0f + 0.5f) - (float)x;
        v[r] = (float)(H - 1) * (rg.yn / 2.0f + 0.5f) - (float)y;
    }
    const float m12 = noocc_pair(u[1], v[1], u[2], v[2]);
    const float m03 = noocc_pair(u[0], v[0], u[3], v[3]);
    float* o = out + (size_t)b * 4 * HW + p;
    o[0] = m03;
    o[HW] = m12;
    o[2 * HW] = m12;
    o[3 * HW] = m03;
}

// ------------------------------------------------------------------ job-table forms (all pyramid levels x reference frames
// of one loss in ONE launch; jobs.h).  Same per-pixel arithmetic as the single-call kernels above (shared device functions).
// Round 5: WPPT pixels per work-item, 256 apart (a workgroup = WPPT consecutive 256-pixel pieces of one image).  The 24 jobs of a
// loss pass are 9-18 k workgroups of 256 pixels; job search, slot pointers, the P / Kinv rows and (backward) the block reduction are
// per-workgroup costs, and the pixels of a work-item are independent load chains (measured on the smoothness kernel: 99 -> 57 us).
constexpr int WPPT = 4;
#define CC_WARP_JOB_BLOCK(t, j, b, blk, H, W, HW, nb1)                                      \
    int local__;                                                                            \
    const int j = ccjobs::find_xcd(t, (int)blockIdx.x, local__);                            \
    const int H = t.H[j], W = t.W[j], HW = H * W, nb1 = (HW + 255) >> 8;                    \
    const int nb4__ = (nb1 + WPPT - 1) / WPPT;                                              \
    const int b = local__ / nb4__, blk = local__ - b * nb4__;

// rigid fwd slots: 0 img [B,C,H,W], 1 depth [B,H,W], 2 P [B,12], 3 Kinv [B,9], 4 out
template <bool AC, bool BORDER>
__global__ __launch_bounds__(256) void k_inverse_warp_fwd_jobs(JobTab t, int C) {
    CC_WARP_JOB_BLOCK(t, j, b, blk, H, W, HW, nb1)
    const float* __restrict__ img = ccjobs::ptr<const float>(t, j, 0);
    const float* __restrict__ depth = ccjobs::ptr<const float>(t, j, 1);
    const float* __restrict__ P = ccjobs::ptr<const float>(t, j, 2);
    const float* __restrict__ Kinv = ccjobs::ptr<const float>(t, j, 3);
    float* __restrict__ out = ccjobs::ptr<float>(t, j, 4);
    const float* src = img + (size_t)b * C * HW;
#pragma unroll
    for (int k = 0; k < WPPT; k++) {
        const int p = (blk * WPPT + k) * 256 + (int)threadIdx.x;
        if (p >= HW) break;
        const int y = p / W, x = p - y * W;
        Rigid r;
        rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, !BORDER, r);
        Bilinear bl;
        bilinear_setup<AC, BORDER>(r.xn, r.yn, W, H, bl);
        float* dst = out + (size_t)b * C * HW + p;
        sample_channels(src, dst, C, HW, W, bl);
    }
}

// rigid bwd slots: 0 gout, 1 img, 2 depth, 3 P, 4 Kinv, 5 gdepth [B,H,W], 6 gP partials [B][nb][12]
template <bool AC, bool BORDER>
__global__ __launch_bounds__(256) void k_inverse_warp_bwd_jobs(JobTab t, int C) {
    __shared__ float red[4 * 12];
    CC_WARP_JOB_BLOCK(t, j, b, blk, H, W, HW, nb)
    const float* __restrict__ gout = ccjobs::ptr<const float>(t, j, 0);
    const float* __restrict__ img = ccjobs::ptr<const float>(t, j, 1);
    const float* __restrict__ depth = ccjobs::ptr<const float>(t, j, 2);
    const float* __restrict__ P = ccjobs::ptr<const float>(t, j, 3);
    const float* __restrict__ Kinv = ccjobs::ptr<const float>(t, j, 4);
    float* __restrict__ gdepth = ccjobs::ptr<float>(t, j, 5);
    float* __restrict__ gP_part = ccjobs::ptr<float>(t, j, 6);
    float gP[12];
#pragma unroll
    for (int i = 0; i < 12; i++) gP[i] = 0.f;
#pragma unroll
    for (int k = 0; k < WPPT; k++) {
        const int p = (blk * WPPT + k) * 256 + (int)threadIdx.x;
        if (p < HW) {
            const int y = p / W, x = p - y * W;
            Rigid r;
            rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, !BORDER, r);
            Bilinear bl;
            bilinear_setup<AC, BORDER>(r.xn, r.yn, W, H, bl);
            float gix, giy, gd;
            sample_grad(img + (size_t)b * C * HW, gout + (size_t)b * C * HW + p, C, HW, W, bl, nullptr, gix, giy);
            float g1[12];                                 // (rigid_backward assigns: this pixel's contribution)
            rigid_backward(P + 12 * b, r, gix * bl.gmx, giy * bl.gmy, W, H, gd, g1);
#pragma unroll
            for (int i = 0; i < 12; i++) gP[i] += g1[i];
            gdepth[(size_t)b * HW + p] = gd;
        }
    }
    cc::block_sum_256<12>(gP, red);
    // the pose-gradient partials keep their [B][ceil(HW / 256)][12] layout (the host sizes it, k_pose_grad_jobs sums it): this
    // workgroup's sums go to the first of its WPPT rows, zeros to the others
    if ((int)threadIdx.x < WPPT) {
        const int row = blk * WPPT + (int)threadIdx.x;
        if (row < nb) {
            float* o = gP_part + ((size_t)b * nb + row) * 12;
#pragma unroll
            for (int i = 0; i < 12; i++) o[i] = threadIdx.x == 0 ? gP[i] : 0.f;
        }
    }
}

// flow fwd slots: 0 img, 1 flow [B,2,H,W], 2 out;   flow bwd slots: 0 gout, 1 img, 2 flow, 3 gflow
template <bool AC, bool BORDER>
__global__ __launch_bounds__(256) void k_flow_warp_fwd_jobs(JobTab t, int C) {
    CC_WARP_JOB_BLOCK(t, j, b, blk, H, W, HW, nb1)
    const float* __restrict__ img = ccjobs::ptr<const float>(t, j, 0);
    const float* __restrict__ flow = ccjobs::ptr<const float>(t, j, 1);
    float* __restrict__ out = ccjobs::ptr<float>(t, j, 2);
    const float* src = img + (size_t)b * C * HW;
#pragma unroll
    for (int k = 0; k < WPPT; k++) {
        const int p = (blk * WPPT + k) * 256 + (int)threadIdx.x;
        if (p >= HW) break;
        const int y = p / W, x = p - y * W;
        float xn, yn, dxn, dyn;
        flow_coords<false>((float)x, (float)y, flow[((size_t)b * 2) * HW + p], flow[((size_t)b * 2 + 1) * HW + p], W, H, xn, yn, dxn, dyn);
        Bilinear bl;
        bilinear_setup<AC, BORDER>(xn, yn, W, H, bl);
        float* dst = out + (size_t)b * C * HW + p;
        sample_channels(src, dst, C, HW, W, bl);
    }
}

template <bool AC, bool BORDER>
__global__ __launch_bounds__(256) void k_flow_warp_bwd_jobs(JobTab t, int C) {
    // (one pixel per work-item: with WPPT = 4 this kernel was slower, 26.0 vs 23.4 us -- no block reduction to amortise)
    int local;
    const int j = ccjobs::find_xcd(t, (int)blockIdx.x, local);
    const int H = t.H[j], W = t.W[j], HW = H * W, nb = (HW + 255) >> 8;
    const int b = local / nb, p = (local - b * nb) * 256 + (int)threadIdx.x;
    if (p >= HW) return;
    const float* __restrict__ gout = ccjobs::ptr<const float>(t, j, 0);
    const float* __restrict__ img = ccjobs::ptr<const float>(t, j, 1);
    const float* __restrict__ flow = ccjobs::ptr<const float>(t, j, 2);
    float* __restrict__ gflow = ccjobs::ptr<float>(t, j, 3);
    const int y = p / W, x = p - y * W;
    float xn, yn, dxn, dyn;
    flow_coords<false>((float)x, (float)y, flow[((size_t)b * 2) * HW + p], flow[((size_t)b * 2 + 1) * HW + p], W, H, xn, yn, dxn, dyn);
    Bilinear bl;
    bilinear_setup<AC, BORDER>(xn, yn, W, H, bl);
    float gix, giy;
    sample_grad(img + (size_t)b * C * HW, gout + (size_t)b * C * HW + p, C, HW, W, bl, nullptr, gix, giy);
    gflow[((size_t)b * 2) * HW + p] = gix * bl.gmx * dxn;
    gflow[((size_t)b * 2 + 1) * HW + p] = giy * bl.gmy * dyn;
}

// depth_occlusion_masks of every level: slots 0 depth [B,H,W], 1 P4 [4,B,12] (full-resolution K, Q4), 2 Kinv, 3 out [B,4,H,W]
__global__ __launch_bounds__(256) void k_rigid_noocc_jobs(JobTab t) {
    int first;
    const int j = ccjobs::find(t, (int)blockIdx.x, first);
    const int H = t.H[j], W = t.W[j], HW = H * W, nb = (HW + 255) >> 8;
    const int local = (int)blockIdx.x - first;
    const int b = local / nb, p = (local - b * nb) * 256 + (int)threadIdx.x;
    if (p >= HW) return;
    const float* __restrict__ depth = ccjobs::ptr<const float>(t, j, 0);
    const float* __restrict__ P4 = ccjobs::ptr<const float>(t, j, 1);
    const float* __restrict__ Kinv = ccjobs::ptr<const float>(t, j, 2);
    float* __restrict__ out = ccjobs::ptr<float>(t, j, 3);
    const int y = p / W, x = p - y * W;
    const float d = depth[(size_t)b * HW + p];
    float u[4], v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        Rigid rg;
        rigid_project(P4 + ((size_t)r * t.B + b) * 12, Kinv + 9 * b, (float)x, (float)y, d, W, H, false, rg);
        u[r] = (float)(W - 1) * (rg.xn / 2.0f + 0.5f) - (float)x;
        v[r] = (float)(H - 1) * (rg.yn / 2.0f + 0.5f) - (float)y;
    }
    const float m12 = noocc_pair(u[1], v[1], u[2], v[2]);
    const float m03 = noocc_pair(u[0], v[0], u[3], v[3]);
    float* o = out + (size_t)b * 4 * HW + p;
    o[0] = m03;
    o[HW] = m12;
    o[2 * HW] = m12;
    o[3 * HW] = m03;
}

// pose2flow of every level (train.py:470-471 flows_cam_fwd / _bwd): slots 0 depth, 1 P [B,12], 2 Kinv, 3 flow [B,2,H,W]
__global__ __launch_bounds__(256) void k_pose2flow_fwd_jobs(JobTab t, int rewrite) {
    CC_WARP_JOB_BLOCK(t, j, b, blk, H, W, HW, nb1)
    const float* __restrict__ depth = ccjobs::ptr<const float>(t, j, 0);
    const float* __restrict__ P = ccjobs::ptr<const float>(t, j, 1);
    const float* __restrict__ Kinv = ccjobs::ptr<const float>(t, j, 2);
    float* __restrict__ flow = ccjobs::ptr<float>(t, j, 3);
#pragma unroll
    for (int k = 0; k < WPPT; k++) {
        const int p = (blk * WPPT + k) * 256 + (int)threadIdx.x;
        if (p >= HW) break;
        const int y = p / W, x = p - y * W;
        Rigid r;
        rigid_project(P + 12 * b, Kinv + 9 * b, (float)x, (float)y, depth[(size_t)b * HW + p], W, H, rewrite != 0, r);
        flow[((size_t)b * 2 + 0) * HW + p] = (float)(W - 1) * (r.xn / 2.0f + 0.5f) - (float)x;
        flow[((size_t)b * 2 + 1) * HW + p] = (float)(H - 1) * (r.yn / 2.0f + 0.5f) - (float)y;
    }
}

// inverse_warp.py:31-45 pixel2cam and :48-79 cam2pixel as stand-alone maps (the training path uses the fused kernels above;
// these share their arithmetic -- ray = Kinv.(x, y, 1) by the same fmaf chain, cam = ray * depth; projection by the same
// chain as rigid_project -- so their composition reproduces the fused kernels' sampling coordinates bit for bit)
__global__ __launch_bounds__(256) void k_pixel2cam(const float* __restrict__ depth, const float* __restrict__ Kinv,
                                                   float* __restrict__ cam, int H, int W) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* Ki = Kinv + 9 * b;
    const float d = depth[(size_t)b * HW + p];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float ray = fmaf(Ki[3 * i + 2], 1.0f, fmaf(Ki[3 * i + 1], (float)y, Ki[3 * i] * (float)x));
        cam[((size_t)b * 3 + i) * HW + p] = ray * d;
    }
}

// P: [B,12] rows (rot | tr); has_rot / has_tr: the reference's `is not None` switches; mode: 0 none, 1 'zeros' (OOB -> 2)
__global__ __launch_bounds__(256) void k_cam2pixel(const float* __restrict__ cam, const float* __restrict__ P,
                                                   float* __restrict__ out, int H, int W, int has_rot, int has_tr, int rewrite) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float c0 = cam[((size_t)b * 3 + 0) * HW + p], c1 = cam[((size_t)b * 3 + 1) * HW + p], c2 = cam[((size_t)b * 3 + 2) * HW + p];
    const float* Pb = P + 12 * b;
    float q[3] = {c0, c1, c2};
    if (has_rot) {
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = fmaf(Pb[4 * i + 2], c2, fmaf(Pb[4 * i + 1], c1, Pb[4 * i] * c0));
    }
    if (has_tr) {
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = q[i] + Pb[4 * i + 3];
    }
    const float Z = fmaxf(q[2], 1e-3f);
    float xn = (2.0f * (q[0] / Z)) / (float)(W - 1) - 1.0f;
    float yn = (2.0f * (q[1] / Z)) / (float)(H - 1) - 1.0f;
    if (rewrite) {
        if (xn > 1.f || xn < -1.f) xn = 2.f;
        if (yn > 1.f || yn < -1.f) yn = 2.f;
    }
    out[((size_t)b * HW + p) * 2] = xn;
    out[((size_t)b * HW + p) * 2 + 1] = yn;
}

// backward of the two stand-alone halves (validation-side API of inverse_warp.py:31-79; the training step uses the fused warp).
// pixel2cam: cam_i = ray_i * d, ray = Kinv . (x, y, 1):  gdepth = sum_i g_i ray_i;  gKinv[i][j] = sum_p g_i d (x, y, 1)_j
// (per-block partials in 12-float records, reduced by k_reduce_gP: slots 0..8 = gKinv row-major).
__global__ __launch_bounds__(256) void k_pixel2cam_bwd(const float* __restrict__ g, const float* __restrict__ depth,
                                                       const float* __restrict__ Kinv, float* __restrict__ gdepth,
                                                       float* __restrict__ part, int H, int W) {
    __shared__ float red[4 * 12];
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0.f;
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        const float* Ki = Kinv + 9 * b;
        const float d = depth[(size_t)b * HW + p];
        const float pix[3] = {(float)x, (float)y, 1.0f};
        float gd = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float gi = g[((size_t)b * 3 + i) * HW + p];
            const float ray = fmaf(Ki[3 * i + 2], 1.0f, fmaf(Ki[3 * i + 1], (float)y, Ki[3 * i] * (float)x));
            gd += gi * ray;
#pragma unroll
            for (int j = 0; j < 3; j++) acc[3 * i + j] = gi * d * pix[j];
        }
        if (gdepth) gdepth[(size_t)b * HW + p] = gd;
    }
    cc::block_sum_256<12>(acc, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) o[i] = acc[i];
    }
}

// cam2pixel: q = rot . cam + tr, Z = max(q2, 1e-3), (xn, yn) = 2 (q0, q1) / Z / (W-1, H-1) - 1; rewritten OOB coordinates and a
// clamped Z carry no gradient (Q10).  gcam = rot^T gq (gq itself without rot); gP[i][0..2] = sum_p gq_i cam, gP[i][3] = sum_p gq_i.
__global__ __launch_bounds__(256) void k_cam2pixel_bwd(const float* __restrict__ ggrid, const float* __restrict__ cam,
                                                       const float* __restrict__ P, float* __restrict__ gcam,
                                                       float* __restrict__ part, int H, int W, int has_rot, int has_tr, int rewrite) {
    __shared__ float red[4 * 12];
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0.f;
    if (p < HW) {
        const float c[3] = {cam[((size_t)b * 3 + 0) * HW + p], cam[((size_t)b * 3 + 1) * HW + p], cam[((size_t)b * 3 + 2) * HW + p]};
        const float* Pb = P + 12 * b;
        float q[3] = {c[0], c[1], c[2]};
        if (has_rot) {
#pragma unroll
            for (int i = 0; i < 3; i++) q[i] = fmaf(Pb[4 * i + 2], c[2], fmaf(Pb[4 * i + 1], c[1], Pb[4 * i] * c[0]));
        }
        if (has_tr) {
#pragma unroll
            for (int i = 0; i < 3; i++) q[i] = q[i] + Pb[4 * i + 3];
        }
        const float Z = fmaxf(q[2], 1e-3f);
        const float xn = (2.0f * (q[0] / Z)) / (float)(W - 1) - 1.0f;
        const float yn = (2.0f * (q[1] / Z)) / (float)(H - 1) - 1.0f;
        float gxn = ggrid[((size_t)b * HW + p) * 2], gyn = ggrid[((size_t)b * HW + p) * 2 + 1];
        if (rewrite && (xn > 1.f || xn < -1.f)) gxn = 0.f;
        if (rewrite && (yn > 1.f || yn < -1.f)) gyn = 0.f;
        const float gu = gxn * (2.0f / (float)(W - 1)), gv = gyn * (2.0f / (float)(H - 1));
        const float iz = 1.0f / Z;
        float gq[3] = {gu * iz, gv * iz, 0.f};
        if (!(q[2] < 1e-3f)) gq[2] = -(gu * q[0] + gv * q[1]) * iz * iz;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float gc = has_rot ? (gq[0] * Pb[j] + gq[1] * Pb[4 + j] + gq[2] * Pb[8 + j]) : gq[j];
            if (gcam) gcam[((size_t)b * 3 + j) * HW + p] = gc;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[4 * i + j] = gq[i] * c[j];
            acc[4 * i + 3] = gq[i];
        }
    }
    cc::block_sum_256<12>(acc, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) o[i] = acc[i];
    }
}

inline dim3 pix_grid(int B, int H, int W) { return dim3((unsigned)((H * W + 255) / 256), (unsigned)B); }

// ---- fixed-point scatter support (cc_feature_warp_bwd_det)
__global__ __launch_bounds__(256) void k_fx_zero(unsigned long long* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n && i < (long)(blockIdx.x + 1) * 1024; i += 256) p[i] = 0ull;
}

// max |x| as the bit pattern of a non-negative float: unsigned atomicMax is order-independent
__global__ __launch_bounds__(256) void k_fx_absmax(const float* __restrict__ x, unsigned* __restrict__ out, long n) {
    __shared__ unsigned red[256];
    unsigned m = 0u;
    const long base = (long)blockIdx.x * 4096;
    for (int k = 0; k < 16; k++) {
        const long i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const unsigned b = __float_as_uint(fabsf(x[i]));
            if (b < 0x7f800000u && b > m) m = b;             // finite values only
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st && red[threadIdx.x + st] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicMax(out, red[0]);
}

__global__ __launch_bounds__(256) void k_fx_to_float(const unsigned long long* __restrict__ fx, const unsigned* __restrict__ mx,
                                                     float* __restrict__ out, long n, int accumulate) {
    int e = 0;
    (void)frexpf(__uint_as_float(*mx), &e);
    const float inv = ldexpf(1.0f, e - 46);
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n && i < (long)(blockIdx.x + 1) * 1024; i += 256) {
        const float v = (float)(long long)fx[i] * inv;
        out[i] = accumulate ? out[i] + v : v;
    }
}

}  // namespace

#define CC_DISPATCH_AC_PAD(KERN, ac, border, ...)                                                            \
    do {                                                                                                     \
        if (ac) {                                                                                            \
            if (border) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERN<true, true>), __VA_ARGS__);                  \
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(KERN<true, false>), __VA_ARGS__);                        \
        } else {                                                                                             \
            if (border) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERN<false, true>), __VA_ARGS__);                 \
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(KERN<false, false>), __VA_ARGS__);                       \
        }                                                                                                    \
    } while (0)

extern "C" {

int cc_pixel2cam(const float* depth, const float* Kinv, float* cam, int B, int H, int W, void* stream) {
    if (B <= 0 || H < 1 || W < 1) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_pixel2cam, pix_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, depth, Kinv, cam, H, W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_cam2pixel(const float* cam, const float* P, float* grid, int B, int H, int W, int has_rot, int has_tr, int rewrite_oob,
                 void* stream) {
    if (B <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_cam2pixel, pix_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, cam, P, grid, H, W, has_rot, has_tr,
                       rewrite_oob);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* backward of cc_pixel2cam: g [B,3,H,W] -> gdepth [B,H,W] (or null), gKinv12 [B,12] (slots 0..8 = d/dKinv row-major, 9..11 zero).
 * ws_partials: cc_warp_partials_bytes(B, H, W). */
int cc_pixel2cam_bwd(const float* g, const float* depth, const float* Kinv, float* gdepth_or_null, float* gKinv12, float* ws_partials,
                     int B, int H, int W, void* stream) {
    if (B <= 0 || H < 1 || W < 1 || !g || !gKinv12 || !ws_partials) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 gr = pix_grid(B, H, W);
    hipLaunchKernelGGL(k_pixel2cam_bwd, gr, dim3(256), 0, s, g, depth, Kinv, gdepth_or_null, ws_partials, H, W);
    hipLaunchKernelGGL(k_reduce_gP, dim3(B), dim3(64), 0, s, (const float*)ws_partials, gKinv12, (int)gr.x, 0);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* backward of cc_cam2pixel: ggrid [B,H,W,2] -> gcam [B,3,H,W] (or null), gP [B,12] rows (d/drot | d/dtr). */
int cc_cam2pixel_bwd(const float* ggrid, const float* cam, const float* P, float* gcam_or_null, float* gP, float* ws_partials, int B,
                     int H, int W, int has_rot, int has_tr, int rewrite_oob, void* stream) {
    if (B <= 0 || H < 2 || W < 2 || !ggrid || !gP || !ws_partials) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 gr = pix_grid(B, H, W);
    hipLaunchKernelGGL(k_cam2pixel_bwd, gr, dim3(256), 0, s, ggrid, cam, P, gcam_or_null, ws_partials, H, W, has_rot, has_tr, rewrite_oob);
    hipLaunchKernelGGL(k_reduce_gP, dim3(B), dim3(64), 0, s, (const float*)ws_partials, gP, (int)gr.x, 0);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

size_t cc_warp_partials_bytes(int B, int H, int W) { return (size_t)B * ((H * W + 255) / 256) * 12 * sizeof(float); }

int cc_inverse_warp_fwd(const float* img, const float* depth, const float* P, const float* Kinv, float* out, int B, int C,
                        int H, int W, int padding_border, int align_corners, void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    CC_DISPATCH_AC_PAD(k_inverse_warp_fwd, align_corners, padding_border, pix_grid(B, H, W), dim3(256), 0, s, img, depth,
                       P, Kinv, out, C, H, W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_inverse_warp_bwd(const float* gout, const float* img, const float* depth, const float* P, const float* Kinv,
                        float* gdepth, float* gP, float* gimg_or_null, float* ws_partials, int B, int C, int H, int W,
                        int padding_border, int align_corners, void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 g = pix_grid(B, H, W);
    CC_DISPATCH_AC_PAD(k_inverse_warp_bwd, align_corners, padding_border, g, dim3(256), 0, s, gout, img, depth, P, Kinv,
                       gdepth, ws_partials, gimg_or_null, C, H, W);
    hipLaunchKernelGGL(k_reduce_gP, dim3(B), dim3(64), 0, s, (const float*)ws_partials, gP, (int)g.x, 0);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_pose2flow_fwd(const float* depth, const float* P, const float* Kinv, float* flow, int B, int H, int W,
                     int rewrite_oob, void* stream) {
    if (B <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_pose2flow_fwd, pix_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, depth, P, Kinv, flow, H, W,
                       rewrite_oob);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_rigid_noocc_fused(const float* depth, const float* P4, const float* Kinv, float* out, int B, int H, int W, void* stream) {
    if (B <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_rigid_noocc_fused, pix_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, depth, P4, Kinv, out, B, H, W);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_pose2flow_bwd(const float* gflow, const float* depth, const float* P, const float* Kinv, float* gdepth, float* gP,
                     float* ws_partials, int B, int H, int W, int rewrite_oob, void* stream) {
    if (B <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 g = pix_grid(B, H, W);
    hipLaunchKernelGGL(k_pose2flow_bwd, g, dim3(256), 0, s, gflow, depth, P, Kinv, gdepth, ws_partials, H, W, rewrite_oob);
    hipLaunchKernelGGL(k_reduce_gP, dim3(B), dim3(64), 0, s, (const float*)ws_partials, gP, (int)g.x, 0);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_flow_warp_fwd(const float* img, const float* flow, float* out, int B, int C, int H, int W, int padding_border,
                     int align_corners, void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 g = pix_grid(B, H, W);
    if (align_corners) {
        if (padding_border) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_fwd<true, true, false>), g, dim3(256), 0, s, img, flow, out, C, H, W, 1.f);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_fwd<true, false, false>), g, dim3(256), 0, s, img, flow, out, C, H, W, 1.f);
    } else {
        if (padding_border) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_fwd<false, true, false>), g, dim3(256), 0, s, img, flow, out, C, H, W, 1.f);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_fwd<false, false, false>), g, dim3(256), 0, s, img, flow, out, C, H, W, 1.f);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_flow_warp_bwd(const float* gout, const float* img, const float* flow, float* gflow_or_null, float* gimg_or_null,
                     int B, int C, int H, int W, int padding_border, int align_corners, void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 g = pix_grid(B, H, W);
    if (align_corners) {
        if (padding_border) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_bwd<true, true, false>), g, dim3(256), 0, s, gout, img, flow, gflow_or_null, gimg_or_null, C, H, W, 1.f);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_bwd<true, false, false>), g, dim3(256), 0, s, gout, img, flow, gflow_or_null, gimg_or_null, C, H, W, 1.f);
    } else {
        if (padding_border) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_bwd<false, true, false>), g, dim3(256), 0, s, gout, img, flow, gflow_or_null, gimg_or_null, C, H, W, 1.f);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_bwd<false, false, false>), g, dim3(256), 0, s, gout, img, flow, gflow_or_null, gimg_or_null, C, H, W, 1.f);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

// models/back2future.py:287-321 Model.warp (border padding, grid = 2(x+u)/max(W-1,1) - 1)
int cc_feature_warp_fwd(const float* feat, const float* flow, float* out, int B, int C, int H, int W, int align_corners,
                        float flow_scale, void* stream) {
    if (B <= 0 || C <= 0 || H < 1 || W < 1) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 g = pix_grid(B, H, W);
    if (C >= 16) g.z = 4;
    if (align_corners) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_fwd<true, true, true>), g, dim3(256), 0, s, feat, flow, out, C, H, W, flow_scale);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_fwd<false, true, true>), g, dim3(256), 0, s, feat, flow, out, C, H, W, flow_scale);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

// gfeat must be zero-filled by the caller (scatter-add with float atomics)
int cc_feature_warp_bwd(const float* gout, const float* feat, const float* flow, float* gflow_or_null,
                        float* gfeat_or_null, int B, int C, int H, int W, int align_corners, float flow_scale, void* stream) {
    if (B <= 0 || C <= 0 || H < 1 || W < 1) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 g = pix_grid(B, H, W);
    if (C >= 16) {
        dim3 g4((unsigned)((H * W + 63) / 64), (unsigned)B);
        if (align_corners) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feature_warp_bwd4<true>), g4, dim3(256), 0, s, gout, feat, flow, gflow_or_null, gfeat_or_null, C, H, W, flow_scale);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feature_warp_bwd4<false>), g4, dim3(256), 0, s, gout, feat, flow, gflow_or_null, gfeat_or_null, C, H, W, flow_scale);
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    if (align_corners) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_bwd<true, true, true>), g, dim3(256), 0, s, gout, feat, flow, gflow_or_null, gfeat_or_null, C, H, W, flow_scale);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flow_warp_bwd<false, true, true>), g, dim3(256), 0, s, gout, feat, flow, gflow_or_null, gfeat_or_null, C, H, W, flow_scale);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* The same backward with a run-to-run REPRODUCIBLE feature gradient: contributions are accumulated as 64-bit fixed-point integers
 * (integer atomics are order-independent), scaled by a power of two taken from max |gout|, then converted:
 *   gfeat (+)= fixed / scale.   ws: cc_feature_warp_bwd_det_ws_bytes(B, C, H, W).  Four launches instead of one + the caller's
 * zero fill.  (The reference's grid_sample backward scatters with float atomics and is not reproducible either.) */
size_t cc_feature_warp_bwd_det_ws_bytes(int B, int C, int H, int W) { return ((size_t)B * C * H * W + 2) * sizeof(unsigned long long); }

int cc_feature_warp_bwd_det(const float* gout, const float* feat, const float* flow, float* gflow_or_null, float* gfeat,
                            void* ws, int B, int C, int H, int W, int align_corners, float flow_scale, int accumulate,
                            void* stream) {
    if (B <= 0 || C <= 0 || H < 1 || W < 1 || !gfeat || !ws) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)B * C * H * W;
    unsigned long long* fx = (unsigned long long*)ws + 2;
    unsigned* mx = (unsigned*)ws;
    hipLaunchKernelGGL(k_fx_zero, dim3((unsigned)((n + 2 + 1023) / 1024)), dim3(256), 0, s, (unsigned long long*)ws, n + 2);
    hipLaunchKernelGGL(k_fx_absmax, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, s, gout, mx, n);
    dim3 g4((unsigned)((H * W + 63) / 64), (unsigned)B);
    if (align_corners) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feature_warp_bwd4<true>), g4, dim3(256), 0, s, gout, feat, flow, gflow_or_null, (float*)nullptr, C, H, W, flow_scale, fx, (const unsigned*)mx);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feature_warp_bwd4<false>), g4, dim3(256), 0, s, gout, feat, flow, gflow_or_null, (float*)nullptr, C, H, W, flow_scale, fx, (const unsigned*)mx);
    hipLaunchKernelGGL(k_fx_to_float, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, (const unsigned long long*)fx, (const unsigned*)mx, gfeat, n, accumulate);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* ---- job-table forms: `jobs` = HOST array of njobs x 10 longs {slot0..slot7, H, W} (slot meaning per entry point, see
 * include/ccengine.h); every job spans B batch items; njobs <= 24. */
static int warp_jobs_tab(ccjobs::JobTab& t, const long* jobs, int njobs, int B) {
    if (!jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return -1;
    return ccjobs::fill(t, jobs, njobs, B, ccjobs::pix_blocks);
}
// ... for the kernels that give every work-item WPPT pixels (256 apart): ceil(ceil(H W / 256) / WPPT) workgroups per image
static int warp_jobs_tab4(ccjobs::JobTab& t, const long* jobs, int njobs, int B) {
    if (!jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return -1;
    return ccjobs::fill(t, jobs, njobs, B, [](int H, int W) { return (ccjobs::pix_blocks(H, W) + WPPT - 1) / WPPT; });
}

int cc_inverse_warp_fwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream) {
    ccjobs::JobTab t;
    const int nblk = warp_jobs_tab4(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    CC_DISPATCH_AC_PAD(k_inverse_warp_fwd_jobs, align_corners, padding_border, dim3((unsigned)nblk), dim3(256), 0,
                       (hipStream_t)stream, t, C);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_inverse_warp_bwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream) {
    ccjobs::JobTab t;
    const int nblk = warp_jobs_tab4(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    CC_DISPATCH_AC_PAD(k_inverse_warp_bwd_jobs, align_corners, padding_border, dim3((unsigned)nblk), dim3(256), 0,
                       (hipStream_t)stream, t, C);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_flow_warp_fwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream) {
    ccjobs::JobTab t;
    const int nblk = warp_jobs_tab4(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    CC_DISPATCH_AC_PAD(k_flow_warp_fwd_jobs, align_corners, padding_border, dim3((unsigned)nblk), dim3(256), 0,
                       (hipStream_t)stream, t, C);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_flow_warp_bwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream) {
    ccjobs::JobTab t;
    const int nblk = warp_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    CC_DISPATCH_AC_PAD(k_flow_warp_bwd_jobs, align_corners, padding_border, dim3((unsigned)nblk), dim3(256), 0,
                       (hipStream_t)stream, t, C);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_rigid_noocc_jobs(const long* jobs, int njobs, int B, void* stream) {
    ccjobs::JobTab t;
    const int nblk = warp_jobs_tab(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_rigid_noocc_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_pose2flow_fwd_jobs(const long* jobs, int njobs, int B, int rewrite_oob, void* stream) {
    ccjobs::JobTab t;
    const int nblk = warp_jobs_tab4(t, jobs, njobs, B);
    if (nblk <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_pose2flow_fwd_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t, rewrite_oob);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
