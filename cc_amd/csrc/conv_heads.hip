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
//   k_conv_thinm<M, DS>        M <= 4 OUTPUT channels <- C <= 64 inputs on the large maps (the heads' forward pass): work-item = 4
//       consecutive pixels x one share of the input channels (CS shares per workgroup, one per wave group); per channel nine loads
//       (three rows: one 16-byte piece + the two halo columns) and 36 M FMAs against LDS-resident weights; the shares' partial
//       sums meet in LDS, the first share applies the epilogue.  Algorithmic traffic: (C + M) x B H W floats (the input once).
//   k_wgrad_thinm<M, CB>       weight gradient of a head (M <= 2 gradient channels; 4 channels measured slower than wgrad_thin.hip): dW[m][c][i][j] = sum_p dY[m][p - (i-1, j-1)] x[c][p].
//       work-item = 4 consecutive pixels x a strip of R rows x CB input channels; the 3 x 6 x M neighbourhood of dY slides down the
//       strip in registers (one new row per step), every input row is ONE 16-byte load per channel (no halo: the taps shift dY, not
//       x); 9 M CB accumulators per work-item, summed over the workgroup at the end of the strip and written as one partial slab
//       per workgroup in the layout of the generic weight-gradient kernel (wgrad_reduce.hip kind 0).
//       Algorithmic traffic: (C + M) x B H W floats; dY is re-read once per channel block.
#include "cc_common.h"
#include "conv_internal.h"
#include "conv_tail.h"
#include "cc_tools.h"

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
    // four output channels per round: their epilogue operands (2 x 4 vector loads) are requested before the first FMA -- a loop of
    // {load, wait, store} per channel would be a chain of memory round trips
    for (int mq = m_beg; mq < m_end; mq += 4) {
        float rv[4][VEC], av[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int m = mq + u < m_end ? mq + u : m_end - 1;
            const long o = (long)m * HW + pix;
            if (hr) vload<VEC>(g.res + (long)n * g.res_bs + o, rv[u]);
            if (ha) vload<VEC>(g.add + (long)n * g.add_bs + o, av[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int m = mq + u;
            if (m >= m_end) break;
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
            const float bias = g.bias ? g.bias[m] : 0.f;
            float out[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++)
                out[v] = cctail::conv_tail(acc[v] + bias, hr, hr ? rv[u][v] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ha ? av[u][v] : 0.f);
            vstore<VEC>(g.y + (long)n * g.y_bs + (long)m * HW + pix, out);
        }
    }
}

constexpr int TM_MAXC = 64;         // input channels, at most (weights stay in LDS: 64 x 4 x 9 floats)
constexpr int TM_MAXCS = 8;         // channel shares per workgroup, at most

// y[m][p] = tail( sum_{c,i,j} w[m][c][i][j] x[c][p + DS (i - 1, j - 1)] ),  M <= 4.  Work-item = 4 consecutive pixels x R rows
// (R = 4 for M <= 2, 2 above: the R + 2 input rows of a channel serve R output rows -- 1.5x / 2x row traffic through the L1 instead
// of 3x, and 3 (R + 2) load instructions per 4 R outputs).  Workgroup = NW waves = (NW / nshare) pixel waves x nshare channel shares:
// wave w accumulates channels [cs * cpw, (cs + 1) * cpw), cs = w % nshare, for the groups (blockIdx.x * (NW / nshare) + w / nshare)
// * 64 + lane.  NB channels per batch: all 3 NB (R + 2) loads are issued before the first FMA.
template <int M, int DS, int R>
__global__ __launch_bounds__(64 * TM_MAXCS) void k_conv_thinm(HeadConv g, int Wg, long ngroups, int cpw, int nshare, int nstrips) {
    constexpr int NB = (R >= 4) ? 1 : (R == 2 ? 2 : 4);
    constexpr int NR = R + 2;
    __shared__ __attribute__((aligned(16))) float wsm[TM_MAXC * M * 9];       // [c][m][9]
    constexpr int MAXW = (M * R >= 8) ? 4 : TM_MAXCS;                         // waves per workgroup, at most (32 KB of partial sums)
    __shared__ float red[MAXW * M * 4 * R * 64];                              // [wave][m][r][v][lane] (share 0's slots stay unused)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wid % nshare, pw = wid / nshare;
    const int pwpb = ((int)blockDim.x >> 6) / nshare;
    const long gid = ((long)blockIdx.x * pwpb + pw) * 64 + lane;
    const bool live = gid < ngroups;
    const long gq = live ? gid : 0;
    const int xg = (int)(gq % Wg);
    const long rr = gq / Wg;
    const int ys = (int)(rr % nstrips), n = (int)(rr / nstrips);
    const int x0 = xg * 4, y0 = ys * R;
    const int HW = g.H * g.W;
    const int c_beg = cs * cpw;
    int c_end = c_beg + cpw;
    if (c_end > g.Cin) c_end = g.Cin;

    float acc[M][R][4];
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[m][r][v] = 0.f;
    // input rows y0 - 1 .. y0 + R (window row q); rows / halo columns outside the image contribute zero (their loads are redirected to
    // an image row / the centre piece and the value masked)
    bool rowin[NR];
    long roff[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int iy = y0 - 1 + q;
        rowin[q] = live && (unsigned)iy < (unsigned)g.H;
        roff[q] = (long)(rowin[q] ? iy : y0) * g.W + x0;
    }
    const bool lin = x0 > 0, rin = x0 + 4 < g.W;
    const int lo = lin ? -1 : 0, ro = rin ? 4 : 3;
    const float* xc = g.x + (long)n * g.x_bs + (long)(c_beg < g.Cin ? c_beg : 0) * HW;      // (an empty share reads channel 0 and uses nothing)
    struct Batch { float4 ctr[NB][NR]; float lft[NB][NR], rgt[NB][NR]; };
    auto load = [&](const float* xp, int nch, Batch& bt) {
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const float* xu = xp + (long)(u < nch ? u : (nch > 0 ? nch - 1 : 0)) * HW;
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const float* row = xu + roff[q];
                bt.ctr[u][q] = *reinterpret_cast<const float4*>(row);
                bt.lft[u][q] = row[lo];
                bt.rgt[u][q] = row[ro];
            }
        }
    };
    auto fma_batch = [&](int c0, int nch, const Batch& bt) {
#pragma unroll
        for (int u = 0; u < NB; u++) {
            if (u >= nch) break;
            float nb[NR][6];
#pragma unroll
            for (int q = 0; q < NR; q++) {
                nb[q][0] = (rowin[q] && lin) ? bt.lft[u][q] : 0.f;
                nb[q][1] = rowin[q] ? bt.ctr[u][q].x : 0.f; nb[q][2] = rowin[q] ? bt.ctr[u][q].y : 0.f;
                nb[q][3] = rowin[q] ? bt.ctr[u][q].z : 0.f; nb[q][4] = rowin[q] ? bt.ctr[u][q].w : 0.f;
                nb[q][5] = (rowin[q] && rin) ? bt.rgt[u][q] : 0.f;
            }
            const float* wc = wsm + (c0 + u) * (M * 9);
#pragma unroll
            for (int m = 0; m < M; m++)
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const float wv = wc[m * 9 + 3 * i + j];
#pragma unroll
                        for (int r = 0; r < R; r++)
#pragma unroll
                            for (int v = 0; v < 4; v++)      // output row y0 + r reads input row y0 + r + DS (i - 1) = window row r + 1 + DS (i - 1)
                                acc[m][r][v] = fmaf(wv, nb[r + (DS > 0 ? i : 2 - i)][v + (DS > 0 ? j : 2 - j)], acc[m][r][v]);
                    }
        }
    };
    // weights -> LDS (four loads in flight per work-item) with the first input batch requested behind them: one memory round trip
    // covers both
    Batch bt;
    {
        const int nw = g.Cin * M * 9;
        bool first = true;
        for (int e0 = tid; e0 < nw || first; e0 += 4 * (int)blockDim.x) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u * (int)blockDim.x;
                const int ee = e < nw ? e : 0;
                const int c = ee / (M * 9), r = ee - c * (M * 9);
                const int m = r / 9, t = r - m * 9;
                v[u] = g.w[g.w0 + (long)m * g.w_sm + (long)c * g.w_sc + t];
            }
            if (first) {
                const int nch = c_end - c_beg;
                load(xc, nch < NB ? nch : NB, bt);
                first = false;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u * (int)blockDim.x;
                if (e < nw) wsm[e] = v[u];
            }
        }
    }
    __syncthreads();
    for (int c = c_beg; c < c_end; c += NB) {
        const int nch = c_end - c < NB ? c_end - c : NB;
        fma_batch(c, nch, bt);
        const int left = c_end - c - NB;
        if (left > 0) {
            xc += NB * (long)HW;
            load(xc, left < NB ? left : NB, bt);
        }
    }
    if (cs > 0) {
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int v = 0; v < 4; v++) red[(((wid * M + m) * R + r) * 4 + v) * 64 + lane] = acc[m][r][v];
    }
    __syncthreads();
    if (cs > 0 || !live) return;
    for (int k = 1; k < nshare; k++)
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[m][r][v] += red[((((wid + k) * M + m) * R + r) * 4 + v) * 64 + lane];
    const bool hr = g.res != nullptr, ha = g.add != nullptr;
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (y0 + r >= g.H) break;
        const long pix = (long)(y0 + r) * g.W + x0;
        float rv[M][4], av[M][4];
#pragma unroll
        for (int m = 0; m < M; m++) {
            const long o = (long)m * HW + pix;
            if (hr) vload<4>(g.res + (long)n * g.res_bs + o, rv[m]);
            if (ha) vload<4>(g.add + (long)n * g.add_bs + o, av[m]);
        }
#pragma unroll
        for (int m = 0; m < M; m++) {
            const float bias = g.bias ? g.bias[m] : 0.f;
            float out[4];
#pragma unroll
            for (int v = 0; v < 4; v++)
                out[v] = cctail::conv_tail(acc[m][r][v] + bias, hr, hr ? rv[m][v] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ha ? av[m][v] : 0.f);
            vstore<4>(g.y + (long)n * g.y_bs + (long)m * HW + pix, out);
        }
    }
}

struct HeadWgrad {
    const float* dy; const float* x; float* ws;
    int B, M, C, H, W;
    long dy_bs, x_bs;
    int R, nstrips, Wg;          // rows per strip, strips per image, 4-pixel groups per row
    long ngroups;                // B * nstrips * Wg work-items per channel block
};

template <int M, int CB>
__global__ __launch_bounds__(256) void k_wgrad_thinm(HeadWgrad g) {
    constexpr int NA = CB * M * 9;
    __shared__ float red[4 * NA];
    const int tid = threadIdx.x;
    const long gid = (long)blockIdx.x * 256 + tid;
    const bool live = gid < g.ngroups;
    const long gq = live ? gid : 0;
    const int xg = (int)(gq % g.Wg);
    const long rr = gq / g.Wg;
    const int ys = (int)(rr % g.nstrips), n = (int)(rr / g.nstrips);
    const int x0 = xg * 4, y0 = ys * g.R;
    int y1 = y0 + g.R;
    if (y1 > g.H) y1 = g.H;
    const int c0 = (int)blockIdx.y * CB;
    const int HW = g.H * g.W;
    const bool lin = x0 > 0, rin = x0 + 4 < g.W;
    const int lo = lin ? -1 : 0, ro = rin ? 4 : 3;

    float acc[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) acc[k] = 0.f;
    // dY neighbourhood: nb[m][r][u], r = 0, 1, 2 <-> rows y - 1, y, y + 1, u <-> columns x0 - 1 .. x0 + 4 (zero outside the image)
    float nb[M][3][6];
    const float* dyn = g.dy + (long)n * g.dy_bs + x0;
    auto load_row = [&](int yy, float (&o)[M][6]) {
        const bool in = live && (unsigned)yy < (unsigned)g.H;
        const float* row = dyn + (long)(in ? yy : 0) * g.W;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const float* rm = row + (long)m * HW;
            const float4 c = *reinterpret_cast<const float4*>(rm);
            const float l = rm[lo], r = rm[ro];
            o[m][0] = (in && lin) ? l : 0.f;
            o[m][1] = in ? c.x : 0.f; o[m][2] = in ? c.y : 0.f; o[m][3] = in ? c.z : 0.f; o[m][4] = in ? c.w : 0.f;
            o[m][5] = (in && rin) ? r : 0.f;
        }
    };
    {
        float t0[M][6], t1[M][6];
        load_row(y0 - 1, t0);
        load_row(y0, t1);
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int u = 0; u < 6; u++) { nb[m][0][u] = t0[m][u]; nb[m][1][u] = t1[m][u]; }
    }
    // input channels of this block (channels past the layer's last are clamped and their sums dropped at the end)
    const float* xb[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) xb[c] = g.x + (long)n * g.x_bs + (long)(c0 + c < g.C ? c0 + c : g.C - 1) * HW + x0;
    for (int y = y0; y < y1; y++) {
        float t2[M][6];
        load_row(y + 1, t2);
        float4 xv[CB];
#pragma unroll
        for (int c = 0; c < CB; c++) xv[c] = *reinterpret_cast<const float4*>(xb[c] + (long)y * g.W);
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int u = 0; u < 6; u++) nb[m][2][u] = t2[m][u];
#pragma unroll
        for (int c = 0; c < CB; c++) {
            const float xs[4] = {live ? xv[c].x : 0.f, live ? xv[c].y : 0.f, live ? xv[c].z : 0.f, live ? xv[c].w : 0.f};
#pragma unroll
            for (int m = 0; m < M; m++)
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        float a = acc[(c * M + m) * 9 + 3 * i + j];
#pragma unroll
                        for (int v = 0; v < 4; v++) a = fmaf(xs[v], nb[m][2 - i][v + 2 - j], a);
                        acc[(c * M + m) * 9 + 3 * i + j] = a;
                    }
        }
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int u = 0; u < 6; u++) { nb[m][0][u] = nb[m][1][u]; nb[m][1][u] = nb[m][2][u]; }
    }
    cc::block_sum_256<NA>(acc, red);
    if (tid == 0) {
        float* w = g.ws + (long)blockIdx.x * g.M * ((long)g.C * 9);
#pragma unroll
        for (int c = 0; c < CB; c++)
            if (c0 + c < g.C)
#pragma unroll
                for (int m = 0; m < M; m++)
#pragma unroll
                    for (int t = 0; t < 9; t++) w[(long)m * g.C * 9 + (long)(c0 + c) * 9 + t] = acc[(c * M + m) * 9 + t];
    }
}

template <int M, int R>
void launch_thinm(const HeadConv& g, int Wg, long ngroups, int cpw, int nshare, int nstrips, dim3 grid, int threads, hipStream_t s) {
    if (g.dstep > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_thinm<M, 1, R>), grid, dim3(threads), 0, s, g, Wg, ngroups, cpw, nshare, nstrips);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv_thinm<M, -1, R>), grid, dim3(threads), 0, s, g, Wg, ngroups, cpw, nshare, nstrips);
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

// few OUTPUT channels (M <= 4) from <= 64 inputs on maps of >= 32768 pixels in total
bool head_conv_thinm_ok(const HeadConv& g) {
    if (g.M < 1 || g.M > 4 || g.Cin < 1 || g.Cin > TM_MAXC || (g.dstep != 1 && g.dstep != -1)) return false;
    if ((long)g.B * g.H * g.W < cctools::env_int("CC_HEAD_MINPIX", 32768)) return false;
    if ((g.W % 4) != 0 || (g.x_bs % 4) != 0 || (g.y_bs % 4) != 0) return false;
    if ((g.res && (g.res_bs % 4) != 0) || (g.add && (g.add_bs % 4) != 0)) return false;
    return aligned_to(g.x, 16) && aligned_to(g.y, 16) && aligned_to(g.res, 16) && aligned_to(g.add, 16);
}

bool head_conv_thinm_launch(const HeadConv& g, hipStream_t s) {
    if (!head_conv_thinm_ok(g)) return false;
    const int Wg = g.W / 4;
    // rows per work-item: 4 (M <= 2) / 2 on maps of >= 200 k pixels, where enough waves remain (measured per map size, us:
    // M1 256x832 34 -> 25, 128x416 24 -> 17, but 64x208 15 -> 24; M4 128x416 36 -> 26, 64x208 20 -> 31), else 1
    const bool big = (long)g.B * g.H * g.W >= cctools::env_int("CC_HEAD_ROWS_MINPIX", 200000);
    const int R = big ? (g.M <= 2 ? 4 : 2) : 1;
    const int nstrips = (g.H + R - 1) / R;
    const long ngroups = (long)g.B * nstrips * Wg;
    const long npw = (ngroups + 63) / 64;                      // pixel waves
    // channel shares: the kernel is a chain of memory round trips per wave (weights + first batch, further batches, epilogue), so
    // short chains in many waves: >= ~8000 waves, >= 4 channels per share; workgroup = 4 waves (8 with 8 shares)
    const int maxshare = g.M * R >= 8 ? 4 : TM_MAXCS;         // (the kernel's LDS for the shares' partial sums: MAXW)
    int nshare = 1;
    while (nshare < maxshare && npw * nshare < cctools::env_int("CC_HEAD_WAVES", 8000) && g.Cin / (nshare * 2) >= 4) nshare *= 2;
    const int cpw = (g.Cin + nshare - 1) / nshare;
    const int nwaves = nshare > 4 ? nshare : 4;
    const int pwpb = nwaves / nshare;
    const long nbx = (npw + pwpb - 1) / pwpb;
    if (nbx >= (1l << 31)) return false;
    dim3 grid((unsigned)nbx);
    const int th = 64 * nwaves;
#define CC_THINM(M_)                                                                                          \
    if (R == 4) launch_thinm<M_, (M_ <= 2 ? 4 : 2)>(g, Wg, ngroups, cpw, nshare, nstrips, grid, th, s);       \
    else if (R == 2) launch_thinm<M_, 2>(g, Wg, ngroups, cpw, nshare, nstrips, grid, th, s);                  \
    else launch_thinm<M_, 1>(g, Wg, ngroups, cpw, nshare, nstrips, grid, th, s);
    switch (g.M) {
        case 1: CC_THINM(1) break;
        case 2: CC_THINM(2) break;
        case 3: CC_THINM(3) break;
        default: CC_THINM(4) break;
    }
#undef CC_THINM
    return true;
}

// ---- weight gradient of a head.  Plan: rows per strip so that >= ~1024 waves are in flight (2 <= R <= 8); channels per block by M.
static inline int hw_cb(int M) { return M == 1 ? 8 : 4; }      // (M = 3, 4 measured slower than wgrad_thin.hip at any block: not taken)
HeadWgradPlan head_wgrad_plan(int B, int M, int H, int W, int Cin) {
    HeadWgradPlan p = {};
    if (cctools::env_int("CC_NO_HEAD_KERNELS", 0) == 1 || cctools::env_int("CC_NO_HEAD_KERNELS", 0) == 4) return p;
    if (M < 1 || M > 2 || Cin < 1 || B < 1 || H < 2 || (W % 4) != 0) return p;
    if ((long)B * H * W < cctools::env_int("CC_HEAD_WGRAD_MINPIX", 16384)) return p;
    const int cb = hw_cb(M);
    const long ncb = (Cin + cb - 1) / cb;
    const int Wg = W / 4;
    int R = 8;
    while (R > 2 && ((long)B * ((H + R - 1) / R) * Wg + 63) / 64 * ncb < cctools::env_int("CC_HEAD_WGRAD_WAVES", 1024)) R >>= 1;
    p.ok = 1;
    p.R = R;
    p.nstrips = (H + R - 1) / R;
    p.nblk = (int)(((long)B * p.nstrips * Wg + 255) / 256);
    p.ws_floats = (size_t)p.nblk * M * Cin * 9;
    return p;
}

bool head_wgrad_launch(const HeadWgradPlan& p, const float* dy, const float* x, float* ws, int B, int M, int H, int W, long dy_bs,
                       int Cin, long x_bs, hipStream_t s) {
    if (!p.ok || !aligned_to(dy, 16) || !aligned_to(x, 16) || (dy_bs % 4) != 0 || (x_bs % 4) != 0) return false;
    HeadWgrad g = {dy, x, ws, B, M, Cin, H, W, dy_bs, x_bs, p.R, p.nstrips, W / 4, (long)B * p.nstrips * (W / 4)};
    const int cb = hw_cb(M);
    dim3 grid((unsigned)p.nblk, (unsigned)((Cin + cb - 1) / cb));
    switch (M) {
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_thinm<1, 8>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_thinm<2, 4>), grid, dim3(256), 0, s, g); break;
        default: return false;
    }
    return true;
}

}  // namespace ccint
