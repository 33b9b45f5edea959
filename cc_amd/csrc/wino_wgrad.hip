// Winograd F(3x3, 2x2) weight gradient of the 3x3 / stride-1 / pad-1 layers on the gfx950 fp32 matrix cores: the third pass of the
// algorithm wino.hip runs for the forward and data-gradient arithmetic (cudnn.benchmark = True picks it for the reference,
// train.py:299), 16 instead of 36 multiply-adds per 2x2 output tile and (m, c) pair.
//
//   forward:   Y = A^T [ (G g G^T) .* (B^T d B) ] A          (2x2 output tile Y, 4x4 input tile d)
//   gradient:  dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
//
// = 16 independent GEMMs (one per frequency f = 4i + j):  S_f[m][c] = sum_t A'_f[m][t] * V_f[c][t],  t = tile,  A' = A dY A^T,
// V = B^T d B, followed by one 4x4 -> 3x3 transform per (m, c).  Workgroup = 64 dY channels x 64 input channels x all 16
// frequencies, as 8 waves: wave w = (sub-tile st = w & 3: 32 m x 32 c, frequency half fh = w >> 2: rows 2 fh, 2 fh + 1 of the
// frequency matrix = 8 accumulator tiles) -- the structure of wino.hip, with BOTH operands produced by in-kernel transforms:
//   * the reduction runs over chunks of 8 consecutive tiles of one tile row (4 MFMA k-steps); a tile row is ceil(TX / 8) chunks,
//     tiles past the row end read out of range = zero, which is exactly their contribution;
//   * transform role of wave w: 16 rows (rg = w >> 1) x 4 tiles (quad q = w & 1) of BOTH operands, one dY item and one input
//     item per thread and stage.  The 2x2 dY block comes straight into registers (two 8-byte buffer loads); the 4x4 input blocks
//     of the wave's 16 channels x 4 tiles come as a wave-private patch by LDS-DMA (4 rows x 16 floats per channel: four
//     instructions), requested one stage ahead;
//   * results go to A'[f][quad][m][t4] / V[f][quad][c][t4] in LDS (double-buffered; lane-linear conflict-free writes, ds_read_b128
//     operands), one barrier per stage.
// Epilogue: each wave applies G^T . G to ITS two frequency rows (the transform is linear), the two waves of a sub-tile exchange
// half of the rows through LDS, and the 3x3 results are written as partial slabs ws[split][tap][m][c] -- the layout of
// k_wgrad3x3, summed by the same deterministic reduce table (wgrad_reduce.hip kind 1).
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "cc_common.h"
#include "cc_tools.h"
#include "conv_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_MAXP = 4;                 // problems per launch (= conv.hip MAXGRP)
constexpr int OBLK = 16 * 8 * 64;          // floats of one operand chunk: [f 16][quad 2][row 64][t4]
constexpr int XPW = 16 * 16 * 4;           // floats of one wave's input patch: [channel 16][input row 4][16 floats]
constexpr int WGT = 512;

struct WW {
    const float* ga[WG_MAXP];
    const float* gx[WG_MAXP];
    float* gws[WG_MAXP];
    int M, C, H, W, HW;
    long a_bs, x_bs;
    unsigned a_bytes, x_bytes;
    int TX, CPR, CPI, NCH, cps;            // tile columns, chunks per tile row / per image / in total, chunks per split
    int ncb, Cp;                           // 64-channel blocks of the input, padded input channels (slab pitch)
};
// one problem of a multi-geometry launch (k_wino_wgrad_multi): the fields the kernel body reads from WW, per problem
struct WW1 {
    const float* a; const float* x; float* ws;
    int M, C, H, W, HW;
    long a_bs, x_bs;
    unsigned a_bytes, x_bytes;
    int TX, CPR, CPI, NCH, cps;
    int ncb, Cp;
    int nxy;                               // 64 x 64 blocks of the problem (its grid is nxy x nsplit, x fastest)
};
constexpr int WWM_MAX = 24;                // problems per multi-geometry launch
struct WWM { WW1 p[WWM_MAX]; int blk_end[WWM_MAX]; int n; };

// bx: 64 x 64 block of the weight matrix (mb * ncb + cb), bz: split of the chunk range
template <int ABL, class D>
__device__ __forceinline__ void wino_wgrad_body(const D& g, const float* __restrict__ a_, const float* __restrict__ x_,
                                                float* __restrict__ ws_, const int bx_, const int bz_) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* As = smem;                      // [2][OBLK]
    float* Vs = smem + 2 * OBLK;           // [2][OBLK]
    float* Xs = smem + 4 * OBLK;           // [8 waves][XPW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st = wid & 3, fh = wid >> 2;
    const int wm = st >> 1, wc = st & 1;
    const int l31 = lane & 31, lk = lane >> 5;
    const int mb = bx_ / g.ncb, cb = bx_ - mb * g.ncb;
    const int c_beg = bz_ * g.cps;
    int c_end = c_beg + g.cps;
    if (c_end > g.NCH) c_end = g.NCH;

    // ---- transform role: rows rg*16 + r16 (of dY AND of the input), tiles 4 q + t4 of the chunk
    const int rg = wid >> 1, q = wid & 1;
    const int r16 = lane >> 2, t4 = lane & 3;
    const cc_buf_t ar = CC_BUF_RSRC(a_, g.a_bytes);
    const cc_buf_t xr = CC_BUF_RSRC(x_, g.x_bytes);
    float* Xw = Xs + wid * XPW;
    const int m_t = mb * 64 + rg * 16 + r16;
    const unsigned dy_row = m_t < g.M ? (unsigned)m_t * (unsigned)g.HW * 4u + (unsigned)t4 * 8u : CC_BUF_OOB;
    // input patch pieces of this lane: piece = i * 64 + lane -> (channel c16 = piece >> 4, input row a = (piece >> 2) & 3, float4 j = piece & 3)
    unsigned xp_c[4], xp_inv[4];                    // byte offset of the piece's channel plane; out-of-range marker (OR-ed in)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = cb * 64 + rg * 16 + ((i * 64 + lane) >> 4);
        xp_c[i] = c < g.C ? (unsigned)c * (unsigned)g.HW * 4u : 0u;
        xp_inv[i] = c < g.C ? 0u : CC_BUF_OOB;
    }
    const int xa = (lane >> 2) & 3, xj = lane & 3;       // (the same for the four pieces: i * 64 does not touch these bits)

    float2 dyr[2];
    // requests of chunk k: the dY block of this thread (registers) and the wave's input patch (LDS-DMA); k past the range: zeros
    auto request = [&](int k) {
        const bool kv = k < g.NCH;
        const int kk = kv ? k : 0;
        const int n = kk / g.CPI;
        const int rem = kk - n * g.CPI;
        const int ty = rem / g.CPR, kx = rem - ty * g.CPR;
        const int txq = 8 * kx + 4 * q;                                   // first tile column of this wave's quad
        if constexpr (!(ABL & 1)) {
            // input patch: rows 2 ty - 1 + a, columns 2 txq - 4 + 4 j .. + 3 (16-byte aligned: W % 4 == 0)
            const int iy = 2 * ty - 1 + xa, ix = 2 * txq - 4 + 4 * xj;
            const bool ok = kv && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const unsigned pix = ok ? (unsigned)(iy * g.W + ix) * 4u : 0u;
            const unsigned pinv = ok ? 0u : CC_BUF_OOB;
            const unsigned so = (unsigned)n * (unsigned)g.x_bs * 4u;
#pragma unroll
            for (int i = 0; i < 4; i++) CC_BUF_GLDS16(xr, (xp_c[i] + pix) | xp_inv[i] | pinv, so, Xw + i * 256);
        }
        // dY block: rows 2 ty, 2 ty + 1, columns 2 (txq + t4), + 1
        const bool tv = kv && txq + t4 < g.TX;
        const unsigned v0 = tv ? dy_row : CC_BUF_OOB;
        const unsigned v1 = (tv && 2 * ty + 1 < g.H) ? dy_row : CC_BUF_OOB;
        const unsigned so = ((unsigned)n * (unsigned)g.a_bs + (unsigned)(2 * ty * g.W + 2 * txq)) * 4u;
        dyr[0] = CC_BUF_LOAD_F32X2(ar, v0, so);
        dyr[1] = CC_BUF_LOAD_F32X2(ar, v1, so + (unsigned)g.W * 4u);
    };
    float raw[16];
    auto read_patch = [&]() {
        const float* src = Xw + r16 * 64 + 2 * t4 + 3;                     // this tile's 4 x 4 block inside the channel's [4][16] window
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) raw[4 * a + b] = (ABL & 2) ? 0.f : src[a * 16 + b];
    };
    // A' = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]: rows u0 = d0, u1 = d0 + d1, u2 = d0 - d1, u3 = -d1, then the same on the columns
    auto dy_step = [&](int half, int buf) {                                // half 0: frequency rows 0, 1; half 1: rows 2, 3
        if constexpr (ABL & 2) return;
        float* o = As + buf * OBLK + q * 256 + rg * 64 + lane;
        const float d00 = dyr[0].x, d01 = dyr[0].y, d10 = dyr[1].x, d11 = dyr[1].y;
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = 2 * half + ii;
            const float u0 = i == 0 ? d00 : (i == 1 ? d00 + d10 : (i == 2 ? d00 - d10 : 0.f - d10));
            const float u1 = i == 0 ? d01 : (i == 1 ? d01 + d11 : (i == 2 ? d01 - d11 : 0.f - d11));
            o[(4 * i + 0) * 512] = u0;
            o[(4 * i + 1) * 512] = u0 + u1;
            o[(4 * i + 2) * 512] = u0 - u1;
            o[(4 * i + 3) * 512] = 0.f - u1;
        }
    };
    // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], in 8 steps (column b of B^T d, then row i of the result)
    float tt[4][4];
    auto col_step = [&](int b) {
        if constexpr (ABL & 2) return;
        const float d0 = raw[b], d1 = raw[4 + b], d2 = raw[8 + b], d3 = raw[12 + b];
        tt[0][b] = d0 - d2;
        tt[1][b] = d1 + d2;
        tt[2][b] = d2 - d1;
        tt[3][b] = d1 - d3;
    };
    auto row_step = [&](int i, int buf) {
        if constexpr (ABL & 2) return;
        float* o = Vs + buf * OBLK + q * 256 + rg * 64 + lane;
        o[(4 * i + 0) * 512] = tt[i][0] - tt[i][2];
        o[(4 * i + 1) * 512] = tt[i][1] + tt[i][2];
        o[(4 * i + 2) * 512] = tt[i][2] - tt[i][1];
        o[(4 * i + 3) * 512] = tt[i][1] - tt[i][3];
    };

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; f++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[f][r] = 0.f;

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // One stage = one chunk of 8 tiles = 8 frequencies x 4 k-steps = 32 MFMAs per wave, straight-line and hand-placed as in
    // wino.hip (frequency pairs alternate; pieces of the NEXT chunk's preparation sit in the MFMAs' shadows):
    //   gap  14     the requests of the next chunk (issued a stage ago) have landed (the compiler's wait for the dY registers covers
    //               the patch: the DMA was issued first; memory reads return in order); read this thread's 4 x 4 input block
    //   gaps 15-16  dY transform -> the other A' buffer;  gaps 17-24  input transform -> the other V buffer
    //   gap  26     request the chunk after the next
    auto stage = [&](auto PAR, int kc) {
        constexpr int buf = decltype(PAR)::value;
        const float4* Ua = reinterpret_cast<const float4*>(As + buf * OBLK) + fh * 1024 + lk * 64 + wm * 32 + l31;
        const float4* Vb = reinterpret_cast<const float4*>(Vs + buf * OBLK) + fh * 1024 + lk * 64 + wc * 32 + l31;
        float4 a[2][2], b[2][2];
        a[0][0] = Ua[0];   b[0][0] = Vb[0];
        a[0][1] = Ua[128]; b[0][1] = Vb[128];
#pragma unroll
        for (int fp = 0; fp < 4; fp++) {
            const int cur = fp & 1, nxt = cur ^ 1;
            if (fp < 3) {
                a[nxt][0] = Ua[(2 * fp + 2) * 128]; b[nxt][0] = Vb[(2 * fp + 2) * 128];
                a[nxt][1] = Ua[(2 * fp + 3) * 128]; b[nxt][1] = Vb[(2 * fp + 3) * 128];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float av[2][4] = {{a[cur][0].x, a[cur][0].y, a[cur][0].z, a[cur][0].w}, {a[cur][1].x, a[cur][1].y, a[cur][1].z, a[cur][1].w}};
            const float bv[2][4] = {{b[cur][0].x, b[cur][0].y, b[cur][0].z, b[cur][0].w}, {b[cur][1].x, b[cur][1].y, b[cur][1].z, b[cur][1].w}};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int f = 2 * fp + (i & 1), j = i >> 1;
                if constexpr (ABL & 8) acc[f][0] = fmaf(av[i & 1][j], bv[i & 1][j], acc[f][0]);
                else acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 1][j], bv[i & 1][j], acc[f], 0, 0, 0);
                const int gap = 8 * fp + i;
                if (gap == 14) { CC_WAIT_VMCNT0_FENCE(); read_patch(); }
                else if (gap == 15 || gap == 16) dy_step(gap - 15, buf ^ 1);
                else if (gap >= 17 && gap < 25) {
                    const int s8 = gap - 17;
                    if (s8 < 4) col_step(s8);
                    else row_step(s8 - 4, buf ^ 1);
                }
                else if (gap == 26) request(kc + 2);
                if (gap >= 14 && gap < 27) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's operand writes are done (the requests stay in flight)
        __builtin_amdgcn_s_barrier();
    };

    if (c_beg < c_end) {
        const int odd = (c_end - c_beg) & 1;     // an odd number of stages starts on parity 1 (the peeled stage in front of the loop)
        request(c_beg);
        CC_WAIT_VMCNT0_FENCE();
        read_patch();
        dy_step(0, odd);
        dy_step(1, odd);
#pragma unroll
        for (int b = 0; b < 4; b++) col_step(b);
#pragma unroll
        for (int i = 0; i < 4; i++) row_step(i, odd);
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the patch has been read
        __builtin_amdgcn_wave_barrier();
        request(c_beg + 1);
        __builtin_amdgcn_s_barrier();
        int kc = c_beg;
        if (odd) {
            stage(I1(), kc);
            kc++;
        }
        for (; kc < c_end; kc += 2) {
            stage(I0(), kc);
            stage(I1(), kc + 1);
        }
    }
    // the last request (a chunk nobody multiplies) must not land in LDS after the exchange below has started
    CC_WAIT_VMCNT0_FENCE();
    __syncthreads();

    // ---- dg = G^T S G, G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]].  This lane holds column c = wc*32 + l31 (MFMA D column) and the 16
    // rows m = wm*32 + (r & 3) + 8*(r >> 2) + 4*lk of frequency rows 2 fh, 2 fh + 1: P = G^T[:, 2fh : 2fh+2] S[2fh : 2fh+2, :] (3 x 4),
    // then P G (3 x 3); the parts of the two frequency halves add up.  A wave finishes rows r in [8 fh, 8 fh + 8) and hands the
    // parts of the other eight rows to its partner through LDS (every buffer is free now).
    auto part = [&](int r, float (&o)[9]) {
        float p[3][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float s0 = acc[j][r], s1 = acc[4 + j][r];
            if (fh == 0) { const float h = 0.5f * s1; p[0][j] = s0 + h; p[1][j] = h; p[2][j] = h; }                 // rows 0, 1
            else { const float h = 0.5f * s0; p[0][j] = h; p[1][j] = 0.f - h; p[2][j] = h + s1; }                    // rows 2, 3
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float hs = 0.5f * (p[a][1] + p[a][2]), hd = 0.5f * (p[a][1] - p[a][2]);
            o[3 * a + 0] = p[a][0] + hs;
            o[3 * a + 1] = hd;
            o[3 * a + 2] = hs + p[a][3];
        }
    };
    float* Ex = smem;                                          // [sub-tile 4][sender fh 2][row 8][value 9][lane 64]
    {
        float* xo = Ex + ((st * 2 + fh) * 72) * 64 + lane;
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) {
            float o[9];
            part(8 * (fh ^ 1) + r8, o);
#pragma unroll
            for (int k = 0; k < 9; k++) xo[(r8 * 9 + k) * 64] = o[k];
        }
    }
    __syncthreads();
    const int c = cb * 64 + wc * 32 + l31;
    const float* xi = Ex + ((st * 2 + (fh ^ 1)) * 72) * 64 + lane;
    const int m_base = mb * 64 + wm * 32 + 4 * lk;
    const long tstride = (long)g.M * g.Cp;
#pragma unroll
    for (int r8 = 0; r8 < 8; r8++) {
        const int r = 8 * fh + r8;
        const int m = m_base + (r & 3) + 8 * (r >> 2);
        float o[9];
        part(r, o);
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] += xi[(r8 * 9 + k) * 64];
        if (m < g.M) {
            float* w = ws_ + (long)bz_ * 9 * tstride + (long)m * g.Cp + c;
#pragma unroll
            for (int k = 0; k < 9; k++) w[k * tstride] = o[k];
        }
    }
}

template <int ABL>
__global__ __launch_bounds__(WGT, 2) void k_wino_wgrad(WW g) {
    if constexpr (CC_XCD_MASK & 2) {
        // XCD order (cc_common.h) over the flattened grid, x fastest: the 64 x 64 blocks of one (problem, chunk range) -- they
        // transform the same slices of dY and x -- run on one XCD
        const int gx = (int)gridDim.x, gy = (int)gridDim.y;
        const int b = cc_xcd_order((int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z), gx * gy * (int)gridDim.z);
        const int r = b / gx, y = r % gy;
        wino_wgrad_body<ABL>(g, g.ga[y], g.gx[y], g.gws[y], b - r * gx, r / gy);
    } else {
        wino_wgrad_body<ABL>(g, g.ga[blockIdx.y], g.gx[blockIdx.y], g.gws[blockIdx.y], (int)blockIdx.x, (int)blockIdx.z);
    }
}

// Problems of DIFFERENT shapes in one launch (what a backward stage leaves parked until its end: the single layers and incomplete
// groups of the small pyramid levels, 8-250 workgroups each -- a launch of its own is mostly ramp-up and drain for them;
// cc_conv2d_wgrad_list).  blockIdx.x ranges over the problems' grids back to back, longest chains first.
__global__ __launch_bounds__(WGT, 2) void k_wino_wgrad_multi(WWM a) {
    int k = 0, first = 0, end = a.blk_end[0];
#pragma unroll 1
    for (int q = 0; q + 1 < a.n; q++)
        if ((int)blockIdx.x >= a.blk_end[q]) { k = q + 1; first = a.blk_end[q]; end = a.blk_end[q + 1]; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    const WW1& g = *(reinterpret_cast<const WW1*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WWM, p)) + k);
#else
    const WW1& g = a.p[k];
#endif
    const int b = (CC_XCD_MASK & 2) ? cc_xcd_order((int)blockIdx.x - first, end - first) : (int)blockIdx.x - first;      // (as k_wino_wgrad)
    const int bz = b / g.nxy;
    wino_wgrad_body<0>(g, g.a, g.x, g.ws, b - bz * g.nxy, bz);
}

}  // namespace

namespace ccint {

WinoWgradPlan wino_wgrad_plan(int B, int M, int H, int W, int Cin, int G, int minq) {
    WinoWgradPlan p = {};
    if (cctools::env_flag("CC_NO_WINO_WGRAD")) return p;
    const int TY = (H + 1) / 2, TX = (W + 1) / 2;
    // rows of both tensors as 16-byte / 8-byte pieces: W % 4 == 0; enough rows on both sides to fill 64 x 64 tiles, enough tiles to
    // reduce over (measured per layer shape: profiles/r04_wino_wgrad_layers.txt)
    if ((W % 4) != 0 || H < 2 || M < cctools::env_int("CC_WW_MINM", 48) || Cin < cctools::env_int("CC_WW_MINC", 48) ||
        (long)B * TY * TX < (minq >= 0 ? minq : cctools::env_int("CC_WW_MINQ", 256)))
        return p;
    if ((long)B * (M > Cin ? M : Cin) * H * W >= (1l << 26)) return p;
    p.ok = 1;
    p.TX = TX;
    p.CPR = (TX + 7) / 8;
    p.CPI = TY * p.CPR;
    p.NCH = B * p.CPI;
    p.nmb = (M + 63) / 64;
    p.ncb = (Cin + 63) / 64;
    p.Cp = p.ncb * 64;
    // One workgroup per CU (the kernel takes the whole LDS), so the launch runs in rounds of 256 workgroups: pick the split count that
    // minimises rounds x (chunks per split + the epilogue's cost in chunks), never more than 256-ish blocks that spill into a
    // second round (r04_ab_round4.txt: 264 blocks ran at half the rate of 252).
    const long base = (long)p.nmb * p.ncb * (G > 1 ? G : 1);
    const long cus = cctools::env_int("CC_WW_SPLIT", 256), minch = cctools::env_int("CC_WW_MINCHUNKS", 2);
    const long epi = cctools::env_int("CC_WW_EPI", 4);
    const long cap = (p.NCH + minch - 1) / minch;
    long nsplit = 1, best = -1;
    for (long ns = 1; ns <= (cap < 1 ? 1 : cap) && ns <= 256; ns++) {
        const long cps = (p.NCH + ns - 1) / ns, eff = (p.NCH + cps - 1) / cps;
        if (eff != ns) continue;
        const long cost = ((base * ns + cus - 1) / cus) * (cps + epi);
        if (best < 0 || cost < best) { best = cost; nsplit = ns; }
    }
    p.cps = (int)((p.NCH + nsplit - 1) / nsplit);
    p.nsplit = (p.NCH + p.cps - 1) / p.cps;
    p.ws_floats = 64 + (size_t)p.nsplit * 9 * M * p.Cp;
    return p;
}

// A problem that will share a multi-geometry launch (wino_wgrad_launch_parked) does not have to fill the chip on its own: fewer,
// longer splits -- the epilogue (worth ~4 chunks) weighs less and fewer partial slabs are written and reduced.  Never more splits
// than the stand-alone plan (the workspace is sized for that one).
WinoWgradPlan wino_wgrad_plan_parked(const WinoWgradPlan& alone, int M) {
    WinoWgradPlan p = alone;
    if (!p.ok) return p;
    const int target = cctools::env_int("CC_WW_PARK_CPS", 16);
    if (target <= 0 || p.cps >= target) return p;
    int cps = target < p.NCH ? target : p.NCH;
    int nsplit = (p.NCH + cps - 1) / cps;
    cps = (p.NCH + nsplit - 1) / nsplit;                 // even shares
    nsplit = (p.NCH + cps - 1) / cps;
    if (nsplit >= alone.nsplit) return p;
    p.cps = cps;
    p.nsplit = nsplit;
    p.ws_floats = 64 + (size_t)p.nsplit * 9 * M * p.Cp;
    return p;
}

static bool g_attr_multi = false;

void wino_wgrad_launch_parked(WinoWgradParked* c, hipStream_t s) {
    if (!c || c->n <= 0) return;
    // longest chains first (chunks per split): the short ones fill the tail
    int order[WINO_WGRAD_PARK_CAP];
    for (int i = 0; i < c->n; i++) order[i] = i;
    for (int i = 1; i < c->n; i++) {
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && c->d[order[j]].cps < c->d[v].cps) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    const size_t smem = (size_t)(4 * OBLK + 8 * XPW) * sizeof(float);
    int i = 0;
    while (i < c->n) {
        WWM m = {};
        long blk = 0;
        for (; i < c->n && m.n < WWM_MAX; i++) {
            const WinoWgradParked::Desc& d = c->d[order[i]];
            const long nb = (long)d.nxy * d.nsplit;
            if (blk + nb >= (1l << 31)) break;
            WW1& w = m.p[m.n];
            w.a = d.a; w.x = d.x; w.ws = d.ws;
            w.M = d.M; w.C = d.C; w.H = d.H; w.W = d.W; w.HW = d.H * d.W;
            w.a_bs = d.a_bs; w.x_bs = d.x_bs; w.a_bytes = d.a_bytes; w.x_bytes = d.x_bytes;
            w.TX = d.TX; w.CPR = d.CPR; w.CPI = d.CPI; w.NCH = d.NCH; w.cps = d.cps; w.ncb = d.ncb; w.Cp = d.Cp; w.nxy = d.nxy;
            blk += nb;
            m.blk_end[m.n] = (int)blk;
            m.n++;
        }
        if (!m.n) break;
        if (!g_attr_multi) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wino_wgrad_multi), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            g_attr_multi = true;
        }
        if (cctools::env_flag("CC_WINO_TRACE")) fprintf(stderr, "wino_wgrad_multi: %d problems, %ld workgroups\n", m.n, blk);
        hipLaunchKernelGGL(k_wino_wgrad_multi, dim3((unsigned)blk), dim3(WGT), smem, s, m);
    }
    c->n = 0;
}

bool wino_wgrad_launch(const WinoWgradPlan& p, const float* const* a, const float* const* x, float* const* ws, int G, int B, int M,
                       int H, int W, long a_bs, int Cin, long x_bs, hipStream_t s, WinoWgradParked* park) {
    const long afl = ((long)B - 1) * a_bs + (long)M * H * W, xfl = ((long)B - 1) * x_bs + (long)Cin * H * W;
    if (G < 1 || G > WG_MAXP || (a_bs % 2) != 0 || (x_bs % 4) != 0 || afl * 4 >= (long)CC_BUF_OOB || xfl * 4 >= (long)CC_BUF_OOB) return false;
    for (int k = 0; k < G; k++)
        if (((uintptr_t)a[k] % 8) != 0 || ((uintptr_t)x[k] % 16) != 0) return false;
    if (park && park->n + G <= WINO_WGRAD_PARK_CAP) {
        for (int k = 0; k < G; k++) {
            WinoWgradParked::Desc& d = park->d[park->n++];
            d.a = a[k]; d.x = x[k]; d.ws = ws[k];
            d.M = M; d.C = Cin; d.H = H; d.W = W; d.a_bs = a_bs; d.x_bs = x_bs;
            d.a_bytes = (unsigned)(afl * 4); d.x_bytes = (unsigned)(xfl * 4);
            d.TX = p.TX; d.CPR = p.CPR; d.CPI = p.CPI; d.NCH = p.NCH; d.cps = p.cps; d.ncb = p.ncb; d.Cp = p.Cp;
            d.nxy = p.nmb * p.ncb; d.nsplit = p.nsplit;
            d.gflop = 2e-9 * 16.0 * B * ((H + 1) / 2) * ((W + 1) / 2) * (double)M * Cin;
        }
        return true;
    }
    WW w = {};
    for (int k = 0; k < G; k++) { w.ga[k] = a[k]; w.gx[k] = x[k]; w.gws[k] = ws[k]; }
    w.M = M; w.C = Cin; w.H = H; w.W = W; w.HW = H * W;
    w.a_bs = a_bs; w.x_bs = x_bs; w.a_bytes = (unsigned)(afl * 4); w.x_bytes = (unsigned)(xfl * 4);
    w.TX = p.TX; w.CPR = p.CPR; w.CPI = p.CPI; w.NCH = p.NCH; w.cps = p.cps;
    w.ncb = p.ncb; w.Cp = p.Cp;
    if (cctools::env_flag("CC_WINO_TRACE"))
        fprintf(stderr, "wino_wgrad: %dx[B%d M%d C%d %dx%d] nmb %d ncb %d chunks %d nsplit %d cps %d\n", G, B, M, Cin, H, W, p.nmb, p.ncb,
                p.NCH, p.nsplit, p.cps);
    const size_t smem = (size_t)(4 * OBLK + 8 * XPW) * sizeof(float);
    dim3 grid((unsigned)(p.nmb * p.ncb), (unsigned)G, (unsigned)p.nsplit);
    auto go = [&](auto kern, bool& attr) {
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(WGT), smem, s, w);
    };
    static bool attr0 = false;
#ifdef CC_TOOLS
    static bool attr_abl[4] = {};
    switch (cctools::env_int("CC_WW_ABL", 0)) {
        case 1: go(k_wino_wgrad<1>, attr_abl[0]); return true;
        case 2: go(k_wino_wgrad<2>, attr_abl[1]); return true;
        case 3: go(k_wino_wgrad<3>, attr_abl[2]); return true;
        case 8: go(k_wino_wgrad<8>, attr_abl[3]); return true;
        default: break;
    }
#endif
    go(k_wino_wgrad<0>, attr0);
    return true;
}

}  // namespace ccint
