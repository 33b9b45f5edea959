// pose (tx,ty,tz,rx,ry,rz) -> P = K_s . [Rx.Ry.Rz | t]   (inverse_warp.py:82-119 euler2mat, :146-162 pose_vec2mat,
// :214/:278 intrinsics.bmm(pose_mat); K_s = K with rows 0,1 divided by the pyramid downscale, loss_functions.py:91)
// and its adjoint dL/dP -> dL/dpose.  Replaces ~17 tiny ATen launches (cos, sin, neg, stack, bmm x3, cat ...) per
// (scale, reference frame) -- ~700 launches per training step -- by one launch each way.  One work-item per sample.
#include "cc_common.h"
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

__global__ void k_pose_proj_bwd(const float* __restrict__ gP, const float* __restrict__ pose, long pose_stride,
                                const float* __restrict__ K, float* __restrict__ gpose, long gpose_stride, int N, float kdiv,
                                int accumulate) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const float* p = pose + (long)n * pose_stride;
    Rot r;
    euler(p, r);
    float Ks[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Ks[i] = (i < 6) ? K[9 * n + i] / kdiv : K[9 * n + i];
    // gT = Ks^T . gP   (3x4)
    float gT[12];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            gT[4 * i + j] = Ks[i] * gP[12 * n + j] + Ks[3 + i] * gP[12 * n + 4 + j] + Ks[6 + i] * gP[12 * n + 8 + j];
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
    float* g = gpose + (long)n * gpose_stride;
    const float v[6] = {gT[3], gT[7], gT[11], grx, gry, grz};
#pragma unroll
    for (int i = 0; i < 6; i++) g[i] = accumulate ? g[i] + v[i] : v[i];
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

}  // extern "C"
