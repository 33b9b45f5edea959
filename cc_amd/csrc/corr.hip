// 9x9-displacement cost volume of Back2Future (models/back2future.py:15-25 `correlate`):
//   vol[b, d, y, x] = (1/C) * sum_c f1[b,c,y,x] * f2[b,c,y+dy-4,x+dx-4],  d = dy*9+dx, zero outside f2
// i.e. the third-party spatial_correlation_sampler CUDA extension (requirements.txt:13, kernel_size=1,
// patch_size=9, stride=1) + the /C of :24 + the channel permutation idx_fwd/idx_bwd of :56-59,175-177
// folded into the store (chan_of_disp[d] = output channel of displacement d; null = identity).
// Gather-form forward AND backward (deterministic, no atomics).  Maps are tiny (4x13 .. 64x208) and the
// op is ~0.2 % of the step's MACs, so the kernels are kept simple: one work-item per output element,
// f2 window reads served by L1/L2.
#include "cc_common.h"
#include "../../include/ccengine.h"

namespace {

constexpr int R = 4, PATCH = 9, ND = 81;

__global__ __launch_bounds__(256) void k_corr_fwd(const float* __restrict__ f1, const float* __restrict__ f2,
                                                  float* __restrict__ out, const int* __restrict__ chan_of_disp, int C,
                                                  int H, int W, int out_cstride_total, int out_coffset) {
    // one thread per (pixel, dy): 9 dx accumulators
    const int HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int dy = blockIdx.y, b = blockIdx.z;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const int yy = y + dy - R;
    float acc[PATCH];
#pragma unroll
    for (int j = 0; j < PATCH; j++) acc[j] = 0.f;
    if (yy >= 0 && yy < H) {
        const float* a = f1 + (size_t)b * C * HW + p;
        const float* bb = f2 + (size_t)b * C * HW + yy * W;
        for (int c = 0; c < C; c++) {
            const float av = a[(size_t)c * HW];
            const float* row = bb + (size_t)c * HW;
#pragma unroll
            for (int j = 0; j < PATCH; j++) {
                const int xx = x + j - R;
                if (xx >= 0 && xx < W) acc[j] = fmaf(av, row[xx], acc[j]);
            }
        }
    }
    const float inv = 1.f / (float)C;
#pragma unroll
    for (int j = 0; j < PATCH; j++) {
        const int d = dy * PATCH + j;
        const int ch = (chan_of_disp ? chan_of_disp[d] : d) + out_coffset;
        out[((size_t)b * out_cstride_total + ch) * HW + p] = acc[j] * inv;
    }
}

// g1[b,c,y,x] = (1/C) sum_d gout[b,ch(d),y,x] * f2[b,c,y+dy-4,x+dx-4]
__global__ __launch_bounds__(256) void k_corr_bwd_f1(const float* __restrict__ gout, const float* __restrict__ f2,
                                                     float* __restrict__ g1, const int* __restrict__ chan_of_disp, int C,
                                                     int H, int W, int g_cstride_total, int g_coffset, int accumulate) {
    const int HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* g = gout + ((size_t)b * g_cstride_total + g_coffset) * HW + p;
    const float* src = f2 + ((size_t)b * C + c) * HW;
    float acc = 0.f;
    for (int dy = 0; dy < PATCH; dy++) {
        const int yy = y + dy - R;
        if (yy < 0 || yy >= H) continue;
        // the nine taps of the row are loaded before they are used (a load inside each skipped-or-not iteration is a chain of up
        // to 81 latencies per work item); taps outside the row contribute fmaf(g, 0, acc) == acc
        float gv[PATCH], sv[PATCH];
#pragma unroll
        for (int dx = 0; dx < PATCH; dx++) {
            const int xx = x + dx - R;
            const int d = dy * PATCH + dx;
            const int ch = chan_of_disp ? chan_of_disp[d] : d;
            const bool in = (xx >= 0) && (xx < W);
            gv[dx] = in ? g[(size_t)ch * HW] : 0.f;
            sv[dx] = in ? src[yy * W + xx] : 0.f;
        }
#pragma unroll
        for (int dx = 0; dx < PATCH; dx++) {
            const int xx = x + dx - R;
            if (xx >= 0 && xx < W) acc = fmaf(gv[dx], sv[dx], acc);
        }
    }
    const size_t o = ((size_t)b * C + c) * HW + p;
    const float r = acc / (float)C;
    g1[o] = accumulate ? g1[o] + r : r;
}

// g2[b,c,y',x'] = (1/C) sum_d gout[b,ch(d),y'-dy+4,x'-dx+4] * f1[b,c,y'-dy+4,x'-dx+4]
__global__ __launch_bounds__(256) void k_corr_bwd_f2(const float* __restrict__ gout, const float* __restrict__ f1,
                                                     float* __restrict__ g2, const int* __restrict__ chan_of_disp, int C,
                                                     int H, int W, int g_cstride_total, int g_coffset) {
    const int HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* g = gout + ((size_t)b * g_cstride_total + g_coffset) * HW;
    const float* src = f1 + ((size_t)b * C + c) * HW;
    float acc = 0.f;
    for (int dy = 0; dy < PATCH; dy++) {
        const int ys = y - dy + R;
        if (ys < 0 || ys >= H) continue;
        float gv[PATCH], sv[PATCH];              // see k_corr_bwd_f1
#pragma unroll
        for (int dx = 0; dx < PATCH; dx++) {
            const int xs = x - dx + R;
            const int d = dy * PATCH + dx;
            const int ch = chan_of_disp ? chan_of_disp[d] : d;
            const bool in = (xs >= 0) && (xs < W);
            const int q = ys * W + xs;
            gv[dx] = in ? g[(size_t)ch * HW + q] : 0.f;
            sv[dx] = in ? src[q] : 0.f;
        }
#pragma unroll
        for (int dx = 0; dx < PATCH; dx++) {
            const int xs = x - dx + R;
            if (xs >= 0 && xs < W) acc = fmaf(gv[dx], sv[dx], acc);
        }
    }
    g2[((size_t)b * C + c) * HW + p] = acc / (float)C;
}

// ---- small maps (W = 13, 26: the two coarsest pyramid levels, 128-192 channels): one work-item per (pixel, displacement)
// instead of per (pixel, dy) -- 9x more work-items for maps of 52-208 pixels; same channel order -> same bits.
__global__ __launch_bounds__(256) void k_corr_fwd_small(const float* __restrict__ f1, const float* __restrict__ f2,
                                                        float* __restrict__ out, const int* __restrict__ chan_of_disp, int C,
                                                        int H, int W, int out_cstride_total, int out_coffset) {
    const int HW = H * W;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= HW * ND) return;
    const int d = t / HW, p = t - d * HW;
    const int dy = d / PATCH, dx = d - dy * PATCH;
    const int y = p / W, x = p - y * W;
    const int yy = y + dy - R, xx = x + dx - R;
    float acc = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float* a = f1 + (size_t)b * C * HW + p;
        const float* bb = f2 + (size_t)b * C * HW + yy * W + xx;
        // 8 channel pairs in flight per step (the plain loop waits for every pair: 0.2 us per channel); same order, same bits
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { av[k] = a[(size_t)(c + k) * HW]; bv[k] = bb[(size_t)(c + k) * HW]; }
#pragma unroll
            for (int k = 0; k < 8; k++) acc = fmaf(av[k], bv[k], acc);
        }
        for (; c < C; c++) acc = fmaf(a[(size_t)c * HW], bb[(size_t)c * HW], acc);
    }
    const int ch = (chan_of_disp ? chan_of_disp[d] : d) + out_coffset;
    out[((size_t)b * out_cstride_total + ch) * HW + p] = acc * (1.f / (float)C);
}

// ---- W % 4 == 0 variants: every work-item owns 4 consecutive pixels (and 4 channels in the backward kernels), so the
// f2 / f1 / gout window of a row is read as three aligned float4 and reused from registers for all 9 dx (the scalar
// kernels above issue one load per FMA).  Same summation order per output element -> bit-identical results.
__device__ __forceinline__ void load12(const float* __restrict__ row, int x, int W, float (&v)[12]) {
    // row[x-4 .. x+7], zero outside [0, W); x % 4 == 0 and W % 4 == 0 -> each quad is entirely inside or outside
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int xx = x - 4 + 4 * q;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xx >= 0 && xx < W) t = *(const float4*)(row + xx);
        v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

__global__ __launch_bounds__(256) void k_corr_fwd4(const float* __restrict__ f1, const float* __restrict__ f2,
                                                   float* __restrict__ out, const int* __restrict__ chan_of_disp, int C,
                                                   int H, int W, int out_cstride_total, int out_coffset) {
    const int HW = H * W, W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int dy = blockIdx.y, b = blockIdx.z;
    if (t >= H * W4) return;
    const int y = t / W4, x = (t - y * W4) * 4;
    const int p = y * W + x;
    const int yy = y + dy - R;
    float acc[PATCH][4];
#pragma unroll
    for (int j = 0; j < PATCH; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[j][k] = 0.f;
    if (yy >= 0 && yy < H) {
        const float* a = f1 + (size_t)b * C * HW + p;
        const float* bb = f2 + (size_t)b * C * HW + yy * W;
        // two channels' operands (8 x 16 bytes) in flight per step, consumed in channel order: the small pyramid levels run
        // 1-2 waves per SIMD and a one-channel loop is then a chain of C load latencies
        int c = 0;
        for (; c + 2 <= C; c += 2) {
            const float4 av0 = *(const float4*)(a + (size_t)c * HW), av1 = *(const float4*)(a + (size_t)(c + 1) * HW);
            float v0[12], v1[12];
            load12(bb + (size_t)c * HW, x, W, v0);
            load12(bb + (size_t)(c + 1) * HW, x, W, v1);
            const float a0[4] = {av0.x, av0.y, av0.z, av0.w}, a1[4] = {av1.x, av1.y, av1.z, av1.w};
#pragma unroll
            for (int j = 0; j < PATCH; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[j][k] = fmaf(a0[k], v0[k + j], acc[j][k]);     // x+k + j-4 -> v[k+j]
#pragma unroll
            for (int j = 0; j < PATCH; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[j][k] = fmaf(a1[k], v1[k + j], acc[j][k]);
        }
        for (; c < C; c++) {
            const float4 av = *(const float4*)(a + (size_t)c * HW);
            const float a4[4] = {av.x, av.y, av.z, av.w};
            float v[12];
            load12(bb + (size_t)c * HW, x, W, v);
#pragma unroll
            for (int j = 0; j < PATCH; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[j][k] = fmaf(a4[k], v[k + j], acc[j][k]);
        }
    }
    const float inv = 1.f / (float)C;
#pragma unroll
    for (int j = 0; j < PATCH; j++) {
        const int d = dy * PATCH + j;
        const int ch = (chan_of_disp ? chan_of_disp[d] : d) + out_coffset;
        *(float4*)(out + ((size_t)b * out_cstride_total + ch) * HW + p) =
            make_float4(acc[j][0] * inv, acc[j][1] * inv, acc[j][2] * inv, acc[j][3] * inv);
    }
}

// 4 pixels x 4 channels per work-item
__global__ __launch_bounds__(256) void k_corr_bwd_f1_4(const float* __restrict__ gout, const float* __restrict__ f2,
                                                       float* __restrict__ g1, const int* __restrict__ chan_of_disp, int C,
                                                       int H, int W, int g_cstride_total, int g_coffset, int accumulate) {
    const int HW = H * W, W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * 4, b = blockIdx.z;
    if (t >= H * W4) return;
    const int y = t / W4, x = (t - y * W4) * 4;
    const int p = y * W + x;
    const float* g = gout + ((size_t)b * g_cstride_total + g_coffset) * HW + p;
    float acc[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[cc][k] = 0.f;
    for (int dy = 0; dy < PATCH; dy++) {
        const int yy = y + dy - R;
        if (yy < 0 || yy >= H) continue;
        float v[4][12];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int c = (c0 + cc < C) ? c0 + cc : C - 1;
            load12(f2 + ((size_t)b * C + c) * HW + yy * W, x, W, v[cc]);
        }
#pragma unroll
        for (int dx = 0; dx < PATCH; dx++) {
            const int d = dy * PATCH + dx;
            const int ch = chan_of_disp ? chan_of_disp[d] : d;
            const float4 gv = *(const float4*)(g + (size_t)ch * HW);
            const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
            // the scalar kernel skips taps outside the row (xx < 0 || xx >= W): load12 returns 0 there, and
            // fmaf(g, 0, acc) == acc exactly, so the chain is the same
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[cc][k] = fmaf(g4[k], v[cc][k + dx], acc[cc][k]);
        }
    }
    const float invC = (float)C;
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
        if (c0 + cc >= C) break;
        float* o = g1 + ((size_t)b * C + c0 + cc) * HW + p;
        float4 r = make_float4(acc[cc][0] / invC, acc[cc][1] / invC, acc[cc][2] / invC, acc[cc][3] / invC);
        if (accumulate) {
            const float4 old = *(const float4*)o;
            r.x = old.x + r.x; r.y = old.y + r.y; r.z = old.z + r.z; r.w = old.w + r.w;
        }
        *(float4*)o = r;
    }
}

__global__ __launch_bounds__(256) void k_corr_bwd_f2_4(const float* __restrict__ gout, const float* __restrict__ f1,
                                                       float* __restrict__ g2, const int* __restrict__ chan_of_disp, int C,
                                                       int H, int W, int g_cstride_total, int g_coffset) {
    const int HW = H * W, W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * 4, b = blockIdx.z;
    if (t >= H * W4) return;
    const int y = t / W4, x = (t - y * W4) * 4;
    const int p = y * W + x;
    const float* g = gout + ((size_t)b * g_cstride_total + g_coffset) * HW;
    float acc[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[cc][k] = 0.f;
    for (int dy = 0; dy < PATCH; dy++) {
        const int ys = y - dy + R;
        if (ys < 0 || ys >= H) continue;
        float v[4][12];                         // f1[c][ys][x-4 .. x+7]
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int c = (c0 + cc < C) ? c0 + cc : C - 1;
            load12(f1 + ((size_t)b * C + c) * HW + ys * W, x, W, v[cc]);
        }
#pragma unroll
        for (int dx = 0; dx < PATCH; dx++) {
            const int d = dy * PATCH + dx;
            const int ch = chan_of_disp ? chan_of_disp[d] : d;
            float gv[12];                       // gout[ch][ys][x-4 .. x+7]; source pixel xs = x + k - dx + 4 -> index k + 8 - dx
            load12(g + (size_t)ch * HW + ys * W, x, W, gv);
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[cc][k] = fmaf(gv[k + 8 - dx], v[cc][k + 8 - dx], acc[cc][k]);
        }
    }
    const float invC = (float)C;
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
        if (c0 + cc >= C) break;
        *(float4*)(g2 + ((size_t)b * C + c0 + cc) * HW + p) =
            make_float4(acc[cc][0] / invC, acc[cc][1] / invC, acc[cc][2] / invC, acc[cc][3] / invC);
    }
}

// ---- general P x P displacement grid with dilation D (FlowNetC6: P = 21, D = 2 on the 1/8-resolution features,
// models/FlowNetC6.py:18-30).  Off the BASELINE path: simple gather kernels, one work-item per output element,
// channel / displacement sums in index order (deterministic).
//   out[b, i*P + j, y, x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c, y + (i-r)*D, x + (j-r)*D],  r = (P-1)/2, zero outside
__global__ __launch_bounds__(256) void k_corrp_fwd(const float* __restrict__ f1, const float* __restrict__ f2,
                                                   float* __restrict__ out, int C, int H, int W, int P, int D) {
    const int HW = H * W, ND2 = P * P, r = (P - 1) / 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= (long)HW * ND2) return;
    const int d = (int)(t / HW), p = (int)(t - (long)d * HW);
    const int i = d / P, j = d - i * P;
    const int y = p / W, x = p - y * W;
    const int yy = y + (i - r) * D, xx = x + (j - r) * D;
    float acc = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float* a = f1 + (size_t)b * C * HW + p;
        const float* bb = f2 + (size_t)b * C * HW + yy * W + xx;
        for (int c = 0; c < C; c++) acc = fmaf(a[(size_t)c * HW], bb[(size_t)c * HW], acc);
    }
    out[((size_t)b * ND2 + d) * HW + p] = acc / (float)C;
}

// g1[b,c,p] = (1/C) sum_d gout[b,d,p] * f2[b,c,p + disp(d)];   g2[b,c,q] = (1/C) sum_d gout[b,d,q - disp(d)] * f1[b,c,q - disp(d)]
__global__ __launch_bounds__(256) void k_corrp_bwd(const float* __restrict__ gout, const float* __restrict__ f1,
                                                   const float* __restrict__ f2, float* __restrict__ g1, float* __restrict__ g2,
                                                   int C, int H, int W, int P, int D) {
    const int HW = H * W, ND2 = P * P, r = (P - 1) / 2;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* g = gout + (size_t)b * ND2 * HW;
    const float* s1 = f1 + ((size_t)b * C + c) * HW;
    const float* s2 = f2 + ((size_t)b * C + c) * HW;
    float a1 = 0.f, a2 = 0.f;
    for (int i = 0; i < P; i++) {
        const int dy = (i - r) * D;
        for (int j = 0; j < P; j++) {
            const int dx = (j - r) * D, d = i * P + j;
            const int yy = y + dy, xx = x + dx;              // f2 sample of output pixel p
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) a1 = fmaf(g[(size_t)d * HW + p], s2[yy * W + xx], a1);
            const int ys = y - dy, xs = x - dx;              // output pixel whose f2 sample is p
            if (ys >= 0 && ys < H && xs >= 0 && xs < W) {
                const int q = ys * W + xs;
                a2 = fmaf(g[(size_t)d * HW + q], s1[q], a2);
            }
        }
    }
    const size_t o = ((size_t)b * C + c) * HW + p;
    if (g1) g1[o] = a1 / (float)C;
    if (g2) g2[o] = a2 / (float)C;
}

}  // namespace

extern "C" {

int cc_corr9x9_fwd(const float* f1, const float* f2, float* out, const int* chan_of_disp_or_null, int B, int C, int H,
                   int W, int out_channels_total, int out_channel_offset, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || out_channel_offset + ND > out_channels_total) return CC_ERR_ARG;
    const bool v4 = (W % 4 == 0) && (((uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)out) % 16 == 0);
    if (v4)
        hipLaunchKernelGGL(k_corr_fwd4, dim3((H * (W / 4) + 255) / 256, PATCH, B), dim3(256), 0, (hipStream_t)stream, f1, f2, out,
                           chan_of_disp_or_null, C, H, W, out_channels_total, out_channel_offset);
    else if (H * W <= 4096)
        hipLaunchKernelGGL(k_corr_fwd_small, dim3((H * W * ND + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, f1, f2, out,
                           chan_of_disp_or_null, C, H, W, out_channels_total, out_channel_offset);
    else
        hipLaunchKernelGGL(k_corr_fwd, dim3((H * W + 255) / 256, PATCH, B), dim3(256), 0, (hipStream_t)stream, f1, f2, out,
                           chan_of_disp_or_null, C, H, W, out_channels_total, out_channel_offset);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_corr9x9_bwd(const float* gout, const float* f1, const float* f2, float* g1, float* g2_or_null,
                   const int* chan_of_disp_or_null, int B, int C, int H, int W, int g_channels_total,
                   int g_channel_offset, int accumulate_g1, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || g_channel_offset + ND > g_channels_total) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (W % 4 == 0) &&
                    (((uintptr_t)gout | (uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)g1 | (uintptr_t)g2_or_null) % 16 == 0);
    if (v4) {
        dim3 g4((H * (W / 4) + 255) / 256, (C + 3) / 4, B);
        hipLaunchKernelGGL(k_corr_bwd_f1_4, g4, dim3(256), 0, s, gout, f2, g1, chan_of_disp_or_null, C, H, W, g_channels_total,
                           g_channel_offset, accumulate_g1);
        if (g2_or_null)
            hipLaunchKernelGGL(k_corr_bwd_f2_4, g4, dim3(256), 0, s, gout, f1, g2_or_null, chan_of_disp_or_null, C, H, W,
                               g_channels_total, g_channel_offset);
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    dim3 g((H * W + 255) / 256, C, B);
    hipLaunchKernelGGL(k_corr_bwd_f1, g, dim3(256), 0, s, gout, f2, g1, chan_of_disp_or_null, C, H, W, g_channels_total,
                       g_channel_offset, accumulate_g1);
    if (g2_or_null)
        hipLaunchKernelGGL(k_corr_bwd_f2, g, dim3(256), 0, s, gout, f1, g2_or_null, chan_of_disp_or_null, C, H, W,
                           g_channels_total, g_channel_offset);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* spatial_correlation_sample(f1, f2, kernel_size=1, patch_size=P, dilation_patch=D) / C  -> [B, P*P, H, W]
 * (models/FlowNetC6.py:18-30: P = 21, D = 2) */
int cc_corr_patch_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int patch, int dilation,
                      void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || patch < 1 || (patch & 1) == 0 || dilation < 1) return CC_ERR_ARG;
    const long n = (long)H * W * patch * patch;
    hipLaunchKernelGGL(k_corrp_fwd, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, f1, f2, out, C, H, W,
                       patch, dilation);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_corr_patch_bwd(const float* gout, const float* f1, const float* f2, float* g1_or_null, float* g2_or_null, int B, int C,
                      int H, int W, int patch, int dilation, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || patch < 1 || (patch & 1) == 0 || dilation < 1) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_corrp_bwd, dim3((H * W + 255) / 256, C, B), dim3(256), 0, (hipStream_t)stream, gout, f1, f2, g1_or_null,
                       g2_or_null, C, H, W, patch, dilation);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
