// 3x3 / stride-1 / pad-1 convolutions with <= 4 channels on ONE side, as plain vector-ALU kernels (gfx950).
//
// The prediction heads of the networks (models/DispResNet6.py:84-89 `predict_disp`: C -> 1 + sigmoid, models/MaskNet6.py pred_mask:
// C -> 4, models/back2future.py:47 flow heads: 32 -> 2) and their data-gradients are not matrix problems: a head's data-gradient
// reduces over 1-4 channels x 9 taps, i.e. 9-36 multiply-adds per output element, and writes B x C x H x W floats -- it is bound
// by that write.  On the MFMA tile kernels (conv.hip) these layers pad the reduction to 8 channels and pay a full tile pipeline
// (weight image + patch DMA + three stages + tile epilogue): 17-35 us per launch whatever the size (tools/head_dgrad_probe.py).
//
//   k_conv_thinc<K, VEC, DS>   K <= 4 REDUCTION channels -> M outputs (a head's data-gradient; a forward layer with <= 4 inputs):
//       work-item = VEC consecutive pixels of one row; its 3 x (VEC + 2) x K input neighbourhood is loaded once into registers;
//       loop over the output channels of the block's range: 9 K VEC FMAs against weights read as wave-uniform LDS words, fused
//       epilogue (conv_tail.h: bias / residual / activation, or (sum + add) * act'(mul) for data-gradients), one VEC-wide store.
//       Algorithmic traffic: (K + M [+ M per epilogue operand]) x B H W floats; the input is re-read once per channel block from L2.
#include "cc_common.h"
#include "conv_internal.h"
#include "conv_tail.h"

namespace {

using ccint::HeadConv;

constexpr int TC_MPB = 16;          // output channels per workgroup, at most

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&o)[VEC]) {
    const typename VecT<VEC>::T v = *reinterpret_cast<const typename VecT<VEC>::T*>(p);
    if constexpr (VEC == 1) o[0] = v;
    else if constexpr (VEC == 2) { o[0] = v.x; o[1] = v.y; }
    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&o)[VEC]) {
    if constexpr (VEC == 1) *p = o[0];
    else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
    else *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}

// DS = +1: y[m][p] = sum_{c,i,j} w[m][c][i][j] x[c][p + (i - 1, j - 1)]   (forward arithmetic)
// DS = -1: y[m][p] = sum_{c,i,j} w[m][c][i][j] x[c][p - (i - 1, j - 1)]   (data-gradient arithmetic: taps run backwards)
// weight element (m, c, i, j) at w[w0 + m * w_sm + c * w_sc + 3 i + j]
template <int K, int VEC, int DS>
__global__ __launch_bounds__(256) void k_conv_thinc(HeadConv g, int Wg, long ngroups, int mpb) {
    __shared__ float wsm[TC_MPB * K * 9];
    const int tid = threadIdx.x;
    const int m_beg = (int)blockIdx.y * mpb;
    int m_end = m_beg + mpb;
    if (m_end > g.M) m_end = g.M;
    for (int e = tid; e < (m_end - m_beg) * K * 9; e += 256) {
        const int mi = e / (K * 9), r = e - mi * (K * 9);
        const int c = r / 9, t = r - c * 9;
        wsm[e] = g.w[g.w0 + (long)(m_beg + mi) * g.w_sm + (long)c * g.w_sc + t];
    }
    __syncthreads();
    const long gid = (long)blockIdx.x * 256 + tid;
    if (gid >= ngroups) return;
    const int xg = (int)(gid % Wg);
    const long rr = gid / Wg;
    const int y = (int)(rr % g.H), n = (int)(rr / g.H);
    const int x0 = xg * VEC;
    const int HW = g.H * g.W;

    // neighbourhood: rows y - DS, y, y + DS (tap rows i = 0, 1, 2), columns x0 - 1 .. x0 + VEC
    float nb[K][3][VEC + 2];
    const float* xn = g.x + (long)n * g.x_bs;
#pragma unroll
    for (int c = 0; c < K; c++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int iy = y + (i - 1) * DS;
            const bool rowin = (unsigned)iy < (unsigned)g.H;
            const float* row = xn + (long)c * HW + (long)(rowin ? iy : 0) * g.W;
            float ctr[VEC];
            if (rowin) vload<VEC>(row + x0, ctr);
            else {
#pragma unroll
                for (int v = 0; v < VEC; v++) ctr[v] = 0.f;
            }
            nb[c][i][0] = (rowin && x0 > 0) ? row[x0 - 1] : 0.f;
#pragma unroll
            for (int v = 0; v < VEC; v++) nb[c][i][1 + v] = ctr[v];
            nb[c][i][VEC + 1] = (rowin && x0 + VEC < g.W) ? row[x0 + VEC] : 0.f;
        }

    const long pix = (long)y * g.W + x0;
    const bool hr = g.res != nullptr, ha = g.add != nullptr;
    for (int m = m_beg; m < m_end; m++) {
        const float* wm = wsm + (m - m_beg) * (K * 9);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = 0.f;
#pragma unroll
        for (int c = 0; c < K; c++)
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float wv = wm[c * 9 + 3 * i + j];
#pragma unroll
                    for (int v = 0; v < VEC; v++) acc[v] = fmaf(wv, nb[c][i][v + (DS > 0 ? j : 2 - j)], acc[v]);
                }
        const long o = (long)m * HW + pix;
        const float bias = g.bias ? g.bias[m] : 0.f;
        float rv[VEC], av[VEC], out[VEC];
        if (hr) vload<VEC>(g.res + (long)n * g.res_bs + o, rv);
        if (ha) vload<VEC>(g.add + (long)n * g.add_bs + o, av);
#pragma unroll
        for (int v = 0; v < VEC; v++)
            out[v] = cctail::conv_tail(acc[v] + bias, hr, hr ? rv[v] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ha ? av[v] : 0.f);
        vstore<VEC>(g.y + (long)n * g.y_bs + o, out);
    }
}

template <int K, int VEC>
void launch_thinc(const HeadConv& g, int Wg, long ngroups, int mpb, dim3 grid, hipStream_t s) {
    if (g.dstep > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_thinc<K, VEC, 1>), grid, dim3(256), 0, s, g, Wg, ngroups, mpb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_thinc<K, VEC, -1>), grid, dim3(256), 0, s, g, Wg, ngroups, mpb);
}

template <int VEC>
void launch_thinc_k(const HeadConv& g, int Wg, long ngroups, int mpb, dim3 grid, hipStream_t s) {
    switch (g.Cin) {
        case 1: launch_thinc<1, VEC>(g, Wg, ngroups, mpb, grid, s); break;
        case 2: launch_thinc<2, VEC>(g, Wg, ngroups, mpb, grid, s); break;
        case 3: launch_thinc<3, VEC>(g, Wg, ngroups, mpb, grid, s); break;
        default: launch_thinc<4, VEC>(g, Wg, ngroups, mpb, grid, s); break;
    }
}

inline bool aligned_to(const void* p, int bytes) { return p == nullptr || ((uintptr_t)p % (uintptr_t)bytes) == 0; }

}  // namespace

namespace ccint {

int head_conv_thinc_vec(const HeadConv& g) {
    if (g.Cin < 1 || g.Cin > 4 || g.M < 1 || g.B < 1 || g.H < 1 || g.W < 1) return 0;
    if (g.dstep != 1 && g.dstep != -1) return 0;
    for (int vec = 4; vec >= 1; vec >>= 1) {
        if ((g.W % vec) != 0 || (g.x_bs % vec) != 0 || (g.y_bs % vec) != 0) continue;
        if ((g.res && (g.res_bs % vec) != 0) || (g.add && (g.add_bs % vec) != 0)) continue;
        const int by = vec * 4;
        if (!aligned_to(g.x, by) || !aligned_to(g.y, by) || !aligned_to(g.res, by) || !aligned_to(g.add, by)) continue;
        return vec;
    }
    return 0;
}

bool head_conv_thinc_launch(const HeadConv& g, hipStream_t s) {
    const int vec = head_conv_thinc_vec(g);
    if (!vec) return false;
    const int Wg = g.W / vec;
    const long ngroups = (long)g.B * g.H * Wg;
    const long nbx = (ngroups + 255) / 256;
    if (nbx >= (1l << 31)) return false;
    // channel blocks: enough workgroups to fill the chip (>= ~1024), at most TC_MPB channels each, at least 4 unless M is smaller
    long nmb = (1024 + nbx - 1) / nbx;
    if (nmb > g.M) nmb = g.M;
    if (nmb < 1) nmb = 1;
    int mpb = (int)((g.M + nmb - 1) / nmb);
    if (mpb < 4) mpb = g.M < 4 ? g.M : 4;
    if (mpb > TC_MPB) mpb = TC_MPB;
    const int nby = (g.M + mpb - 1) / mpb;
    if (nby > 65535) return false;
    dim3 grid((unsigned)nbx, (unsigned)nby);
    if (vec == 4) launch_thinc_k<4>(g, Wg, ngroups, mpb, grid, s);
    else if (vec == 2) launch_thinc_k<2>(g, Wg, ngroups, mpb, grid, s);
    else launch_thinc_k<1>(g, Wg, ngroups, mpb, grid, s);
    return true;
}

}  // namespace ccint
