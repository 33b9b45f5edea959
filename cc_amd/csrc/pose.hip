// pose (tx,ty,tz,rx,ry,rz) -> P = K_s . [Rx.Ry.Rz | t]   (inverse_warp.py:82-119 euler2mat, :146-162 pose_vec2mat,
// :214/:278 intrinsics.bmm(pose_mat); K_s = K with rows 0,1 divided by the pyramid downscale, loss_functions.py:91)
// and its adjoint dL/dP -> dL/dpose.  Replaces ~17 tiny ATen launches (cos, sin, neg, stack, bmm x3, cat ...) per
// (scale, reference frame) -- ~700 launches per training step -- by one launch each way.  One work-item per sample.
#include "cc_common.h"
#include "jobs.h"
#include "../../include/ccengine.h"

namespace {

struct Rot { float R[9]; float cx, sx, cy, sy, cz, sz; };

__device__ __forceinline__ void euler(const float* p, Rot& r) {
    r.cx = cosf(p[3]); r.sx = sinf(p[3]);
    r.cy = cosf(p[4]); r.sy = sinf(p[4]);
    r.cz = cosf(p[5]); r.sz = sinf(p[5]);
    // (Rx.Ry).Rz
    const float a00 = r.cy, a01 = 0.f, a02 = r.sy;
    const float a10 = r.sx * r.sy, a11 = r.cx, a12 = -r.sx * r.cy;
    const float a20 = -r.cx * r.sy, a21 = r.sx, a22 = r.cx * r.cy;
    r.R[0] = a00 * r.cz + a01 * r.sz; r.R[1] = -a00 * r.sz + a01 * r.cz; r.R[2] = a02;
    r.R[3] = a10 * r.cz + a11 * r.sz; r.R[4] = -a10 * r.sz + a11 * r.cz; r.R[5] = a12;
    r.R[6] = a20 * r.cz + a21 * r.sz; r.R[7] = -a20 * r.sz + a21 * r.cz; r.R[8] = a22;
}

__global__ void k_pose_proj_fwd(const float* __restrict__ pose, long pose_stride, const float* __restrict__ K,
                                float* __restrict__ P, int N, float kdiv) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const float* p = pose + (long)n * pose_stride;
    Rot r;
    euler(p, r);
    float Ks[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Ks[i] = (i < 6) ? K[9 * n + i] / kdiv : K[9 * n + i];
    float* o = P + 12 * n;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            o[4 * i + j] = fmaf(Ks[3 * i + 2], r.R[6 + j], fmaf(Ks[3 * i + 1], r.R[3 + j], Ks[3 * i] * r.R[j]));
        o[4 * i + 3] = fmaf(Ks[3 * i + 2], p[2], fmaf(Ks[3 * i + 1], p[1], Ks[3 * i] * p[0]));
    }
}

// dL/dP (12) -> dL/d(tx,ty,tz,rx,ry,rz) for one sample: p = its pose vector, Kn = its 3x3 intrinsics, kdiv = pyramid downscale
__device__ __forceinline__ void pose_bwd_one(const float* __restrict__ gPn, const float* __restrict__ p,
                                             const float* __restrict__ Kn, float kdiv, float (&v)[6]) {
    Rot r;
    euler(p, r);
    float Ks[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Ks[i] = (i < 6) ? Kn[i] / kdiv : Kn[i];
    // gT = Ks^T . gP   (3x4)
    float gT[12];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            gT[4 * i + j] = Ks[i] * gPn[j] + Ks[3 + i] * gPn[4 + j] + Ks[6 + i] * gPn[8 + j];
    // R = A.Rz with A = Rx.Ry ;  d/drz: A.dRz ; d/dry: Rx.dRy.Rz ; d/drx: dRx.Ry.Rz
    const float cx = r.cx, sx = r.sx, cy = r.cy, sy = r.sy, cz = r.cz, sz = r.sz;
    const float A[9] = {cy, 0.f, sy, sx * sy, cx, -sx * cy, -cx * sy, sx, cx * cy};
    float grz = 0.f, gry = 0.f, grx = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // dR/drz row i = A[i,:] . dRz,  dRz = [[-sz,-cz,0],[cz,-sz,0],[0,0,0]]
        const float d0 = -A[3 * i] * sz + A[3 * i + 1] * cz, d1 = -A[3 * i] * cz - A[3 * i + 1] * sz;
        grz += gT[4 * i] * d0 + gT[4 * i + 1] * d1;
    }
    {
        // dA/dry = Rx . dRy,  dRy = [[-sy,0,cy],[0,0,0],[-cy,0,-sy]]
        const float dA[9] = {-sy, 0.f, cy, sx * cy, 0.f, sx * sy, -cx * cy, 0.f, -cx * sy};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float d0 = dA[3 * i] * cz + dA[3 * i + 1] * sz, d1 = -dA[3 * i] * sz + dA[3 * i + 1] * cz, d2 = dA[3 * i + 2];
            gry += gT[4 * i] * d0 + gT[4 * i + 1] * d1 + gT[4 * i + 2] * d2;
        }
        // dA/drx = dRx . Ry,  dRx = [[0,0,0],[0,-sx,-cx],[0,cx,-sx]]
        const float dB[9] = {0.f, 0.f, 0.f, cx * sy, -sx, -cx * cy, sx * sy, cx, -sx * cy};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float d0 = dB[3 * i] * cz + dB[3 * i + 1] * sz, d1 = -dB[3 * i] * sz + dB[3 * i + 1] * cz, d2 = dB[3 * i + 2];
            grx += gT[4 * i] * d0 + gT[4 * i + 1] * d1 + gT[4 * i + 2] * d2;
        }
    }
    v[0] = gT[3]; v[1] = gT[7]; v[2] = gT[11]; v[3] = grx; v[4] = gry; v[5] = grz;
}

__global__ void k_pose_proj_bwd(const float* __restrict__ gP, const float* __restrict__ pose, long pose_stride,
                                const float* __restrict__ K, float* __restrict__ gpose, long gpose_stride, int N, float kdiv,
                                int accumulate) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    float v[6];
    pose_bwd_one(gP + 12 * n, pose + (long)n * pose_stride, K + 9 * n, kdiv, v);
    float* g = gpose + (long)n * gpose_stride;
#pragma unroll
    for (int i = 0; i < 6; i++) g[i] = accumulate ? g[i] + v[i] : v[i];
}

// P of every (level, reference frame, batch item) in one launch: pose [B,R,6], K [B,9] -> P_all [L][R][B][12], level l using
// K rows 0,1 / kdiv[l] (loss_functions.py:91; kdiv = 1: the full-resolution K of the occlusion masks, Q4)
struct PoseLv { int L, R, B; float kdiv[8]; };

__global__ void k_pose_proj_levels(const float* __restrict__ pose, const float* __restrict__ K, float* __restrict__ P, PoseLv lv) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= lv.L * lv.R * lv.B) return;
    const int b = idx % lv.B, r = (idx / lv.B) % lv.R, l = idx / (lv.B * lv.R);
    const float* p = pose + ((long)b * lv.R + r) * 6;
    const float kdiv = lv.kdiv[l];
    Rot rt;
    euler(p, rt);
    float Ks[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Ks[i] = (i < 6) ? K[9 * b + i] / kdiv : K[9 * b + i];
    float* o = P + 12 * (long)idx;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            o[4 * i + j] = fmaf(Ks[3 * i + 2], rt.R[6 + j], fmaf(Ks[3 * i + 1], rt.R[3 + j], Ks[3 * i] * rt.R[j]));
        o[4 * i + 3] = fmaf(Ks[3 * i + 2], p[2], fmaf(Ks[3 * i + 1], p[1], Ks[3 * i] * p[0]));
    }
}

// dL/dpose[b][r] = sum over levels of pose_bwd(sum over pixel blocks of the dL/dP partials of job (level, r)); one wave per
// (r, b), levels in ascending order, fixed-order reductions -> deterministic.  Jobs are ordered level-major (j = l * R + r),
// slot 0 = partials [B][nb][12] of cc_inverse_warp_bwd_jobs, H, W give nb.
__global__ __launch_bounds__(64) void k_pose_grad_jobs(ccjobs::JobTab t, const float* __restrict__ pose,
                                                       const float* __restrict__ K, float* __restrict__ gpose, PoseLv lv) {
    const int r = blockIdx.x % lv.R, b = blockIdx.x / lv.R, lane = threadIdx.x;
    float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < lv.L; l++) {
        const int j = l * lv.R + r;
        const int nb = (t.H[j] * t.W[j] + 255) >> 8;
        const float* part = ccjobs::ptr<const float>(t, j, 0) + (size_t)b * nb * 12;
        float acc[12];
#pragma unroll
        for (int i = 0; i < 12; i++) acc[i] = 0.f;
        for (int k = lane; k < nb; k += 64) {
#pragma unroll
            for (int i = 0; i < 12; i++) acc[i] += part[(size_t)k * 12 + i];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) acc[i] = cc::wave_sum(acc[i]);
        float v[6];
        pose_bwd_one(acc, pose + ((long)b * lv.R + r) * 6, K + 9 * b, lv.kdiv[l], v);
#pragma unroll
        for (int i = 0; i < 6; i++) tot[i] += v[i];
    }
    if (lane == 0) {
        float* g = gpose + ((long)b * lv.R + r) * 6;
#pragma unroll
        for (int i = 0; i < 6; i++) g[i] = tot[i];
    }
}

}  // namespace

extern "C" {

int cc_pose_proj_fwd(const float* pose, long pose_stride, const float* K, float* P, int N, float k_div, void* stream) {
    if (N <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_pose_proj_fwd, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, pose_stride, K, P, N, k_div);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_pose_proj_bwd(const float* gP, const float* pose, long pose_stride, const float* K, float* gpose, long gpose_stride,
                     int N, float k_div, int accumulate, void* stream) {
    if (N <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_pose_proj_bwd, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, gP, pose, pose_stride, K, gpose,
                       gpose_stride, N, k_div, accumulate);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* pose [B,R,6] (contiguous), K [B,9] -> P_all [L][R][B][12]; kdiv_host: L floats (HOST) */
int cc_pose_proj_levels(const float* pose, const float* K, float* P_all, int L, int R, int B, const float* kdiv_host, void* stream) {
    if (L <= 0 || L > 8 || R <= 0 || B <= 0) return CC_ERR_ARG;
    PoseLv lv = {};
    lv.L = L; lv.R = R; lv.B = B;
    for (int l = 0; l < L; l++) lv.kdiv[l] = kdiv_host[l];
    hipLaunchKernelGGL(k_pose_proj_levels, dim3((unsigned)((L * R * B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pose, K, P_all, lv);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* gpose [B,R,6] = d loss / d pose from the dL/dP partials of the njobs = L*R inverse-warp backward jobs (level-major) */
int cc_pose_grad_jobs(const long* jobs, int njobs, int L, int R, int B, const float* pose, const float* K, float* gpose,
                      const float* kdiv_host, void* stream) {
    if (!jobs || njobs != L * R || njobs > ccjobs::MAXJOBS || L > 8 || B <= 0) return CC_ERR_ARG;
    ccjobs::JobTab t;
    ccjobs::fill(t, jobs, njobs, B, ccjobs::pix_blocks);
    PoseLv lv = {};
    lv.L = L; lv.R = R; lv.B = B;
    for (int l = 0; l < L; l++) lv.kdiv[l] = kdiv_host[l];
    hipLaunchKernelGGL(k_pose_grad_jobs, dim3((unsigned)(R * B)), dim3(64), 0, (hipStream_t)stream, t, pose, K, gpose, lv);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
