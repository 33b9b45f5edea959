// Winograd F(2x2, 3x3) convolution on the gfx950 fp32 matrix cores: the 3x3 / stride-1 / pad-1 layers of the four CC networks
// (forward AND data-gradient arithmetic -- the data-gradient of such a layer is the same convolution with flipped, transposed
// weights) at 4 instead of 9 multiply-adds per output pixel and channel pair.  This is the algorithm choice the reference gets
// from cudnn.benchmark = True (train.py:299); fp32 operands, fp32 accumulation, no reduced-precision emulation.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, d = its 4x4 input tile (Lavin & Gray 2016)
//
// = 16 independent GEMMs (one per "frequency" f = 4i + j):  D_f[m][t] = sum_c U_f[m][c] * V_f[c][t],  t = tile index.
// One workgroup (4 waves, one per SIMD) = 64 output channels x 64 tiles (256 output pixels) x all 16 frequencies; each wave owns a
// 32 x 32 (m x t) sub-tile of ALL 16 frequencies: 16 accumulator tiles of v_mfma_f32_32x32x2_f32 = 256 registers per lane, so
// the output transform A^T M A runs entirely in registers (the 16 values of one (m, t) lie in one lane).  Per 8-channel chunk:
//   * U (transformed weights, produced once per step by k_repack_table / per call by k_wino_weights in the staging layout
//     [f][quad][m][c4], wino_weights.h) arrives by LDS-DMA, 32 KB, double-buffered;
//   * the input transform is fused: every thread loads the 4x4 input tile of one (channel, tile) pair straight from global memory
//     (raw buffer loads: the zero padding at the image border is the hardware's out-of-range answer, no compare / select),
//     transforms it in registers (32 add/sub) and writes the 16 frequencies to V[f][quad][t][c4] in LDS (conflict-free
//     ds_write_b32, lane-linear) -- the loads of chunk k+1 are issued before the MFMAs of chunk k and consumed between them;
//   * 64 MFMAs per wave; both operands by conflict-free ds_read_b128 (one read = the operand of four k-steps).
// One barrier per chunk (4096 MFMA cycles).  Tiles are numbered linearly over (image, tile row, tile column), so maps of any
// size fill the 64-tile blocks (64x208: 52 blocks per image; 16x52: 13 blocks over 4 images).  Workgroups that share input
// tiles (the m-blocks of one tile block) are neighbours on one XCD (blockIdx swizzle) and hit its L2.
// Split-K over channel chunks for deep layers on small maps: partial sums go through the output transform first (it is
// linear) and are written as [split][n][m][Hp][Wp] slabs for conv.hip's deterministic k_splitk_epilogue*.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "cc_common.h"
#include "cc_tools.h"
#include "conv_tail.h"
#include "conv_internal.h"
#include "wino_weights.h"

namespace {

using namespace cctail;
using namespace ccwino;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WN {
    ccint::WinoProb p[ccint::WINO_MAXP];
    int total, per_prob, nmb;
    int Cin, H, W, HW;
    unsigned x_bytes;
    long x_bs;
    int M;
    long y_bs, res_bs, add_bs;
    int TX, TPI, Q, nchunk, cps;
    long part_stride;
    int Hp, Wp;
    int act;
    float act_a, act_b;
    int res_mul, vec2;
};

constexpr int VBLK = 16 * WCK * 64;       // floats of one V chunk (64 tiles)
// raw input patch of ONE wave (its 16 consecutive tiles) for one 8-channel chunk: [channel 8][input row 4][12 float4] floats
constexpr int RROW = 12 * 4, RCH = 4 * RROW, RWAVE = WCK * RCH;      // 48, 192, 1536 floats (6 KB per wave)

// epilogue forms (template parameter EPI; chosen on the host): branch-free code for what the step uses, the generic tail otherwise
//   EPI_LIN : y = act(v + bias + res),       act in {none, ReLU, LeakyReLU}  as  t > 0 ? t : slope * t   (slope 1 / 0 / s)
//   EPI_GRAD: y = (v + add) * act'(res),     act in {ReLU, LeakyReLU}        as  res > 0 ? t : slope * t
//   EPI_GEN : conv_tail() (sigmoid forms)
enum { EPI_LIN = 0, EPI_GRAD = 1, EPI_GEN = 2 };

// ABL (tools build only, CC_WINO_ABL): timing ablations that compute garbage -- 1: no raw-patch DMA, 2: no input transform / V
// writes, 4: no U DMA, 8: no MFMAs, 16: no fragment reads, 32: no epilogue
template <int SPLIT, int EPI, int ABL>
__global__ __launch_bounds__(256, 1) void k_wino_f2x3(WN g) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* Us = smem;                          // [2][UBLK]
    float* Vs = smem + 2 * UBLK;               // [2][VBLK]
    float* Rs = smem + 2 * UBLK + 2 * VBLK;    // [4 waves][RWAVE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wt = wid & 1;
    const int l31 = lane & 31, lk = lane >> 5;

    // workgroup -> (problem, tile block, m block); blocks b, b + 8, b + 16 ... run on one XCD: give them consecutive work items
    const int cnt = (int)gridDim.x >> 3;
    const int w = ((int)blockIdx.x & 7) * cnt + ((int)blockIdx.x >> 3);
    if (w >= g.total) return;
    const int prob = w / g.per_prob;
    const int wr = w - prob * g.per_prob;
    const int qb = wr / g.nmb, mb = wr - qb * g.nmb;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    // per-problem pointers straight from the kernel-argument segment (indexing the by-value struct copies it to scratch)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const ccint::WinoProb& P = *(reinterpret_cast<const ccint::WinoProb*>(ka + offsetof(WN, p)) + prob);
#else
    const ccint::WinoProb& P = g.p[prob];
#endif
    const int c_beg = (int)blockIdx.z * g.cps;
    int c_end = c_beg + g.cps;
    if (c_end > g.nchunk) c_end = g.nchunk;

    // ---- input path.  Every VMEM instruction costs the issuing wave 60-75 cycles of its in-order stream (measured: 32 dword loads
    // per thread and chunk cost 2 800 cycles per stage even when all of them were out of range, profiles/r04_wino_probe.txt), so
    // the raw input goes the way that needs the fewest: 16-byte LDS-DMA.  A wave stages the input rows of ITS 16 consecutive
    // tiles (it transforms exactly those): the tiles form one or two runs inside a tile row (TX >= 16, or TX == 8 and two whole
    // rows); per channel and input row a run is 2 len + 2 floats, fetched as the 16-byte-aligned float4s that cover it (the rows of
    // x are 16-byte aligned: W % 4 == 0) -- at most 12 float4 per (channel, row), 384 per chunk = SIX DMA instructions per wave.
    // Out-of-image float4s and channels past Cin are out-of-range offsets of the buffer resource (they move zeros).
    float* Rw = Rs + wid * RWAVE;
    const int q0 = qb * 64 + 16 * wid;
    int rn[2], riy[2], rxal[2], rlen[2], rjb[2], rxs[2];
    {
        int q = q0;
        int jb = 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const bool v = q < g.Q;
            const int qq = v ? q : 0;
            const int n = qq / g.TPI;
            const int rem = qq - n * g.TPI;
            const int ty = rem / g.TX, tx = rem - ty * g.TX;
            int len = g.TX - tx;
            const int left = 16 - (q - q0);
            if (len > left) len = left;
            if (len > g.Q - q) len = g.Q - q;
            if (!v || len < 0) len = 0;
            const int xs = 2 * tx - 1;
            const int xal = xs & ~3;                       // floor to a multiple of 4 (xs = -1 -> -4)
            rn[r] = n; riy[r] = 2 * ty - 1; rxal[r] = xal; rlen[r] = len; rjb[r] = jb; rxs[r] = xs;
            jb += len > 0 ? (xs + 2 * len + 2 - xal + 3) >> 2 : 0;
            q += len;
        }
    }
    unsigned doff[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int piece = i * 64 + lane;                   // (channel, input row, float4 slot) = (piece / 48, piece % 48 / 12, piece % 12)
        const int cl = piece / 48, rem = piece - cl * 48;
        const int a = rem / 12, j = rem - a * 12;
        const int r = (rlen[1] > 0 && j >= rjb[1]) ? 1 : 0;
        const int n = r ? rn[1] : rn[0], iy = (r ? riy[1] : riy[0]) + a, xal = r ? rxal[1] : rxal[0];
        const int len = r ? rlen[1] : rlen[0], xs = r ? rxs[1] : rxs[0], jb = r ? rjb[1] : rjb[0];
        const int x = xal + 4 * (j - jb);
        const bool ok = len > 0 && x < xs + 2 * len + 2 && (unsigned)iy < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        doff[i] = ok ? (unsigned)(((long)n * g.x_bs + (long)cl * g.HW + (long)iy * g.W + x) * 4) : CC_BUF_OOB;
    }
    const cc_buf_t xr = CC_BUF_RSRC(P.x, g.x_bytes);
    auto dma_raw = [&](int kc, int i) {
        if constexpr (ABL & 1) return;
        const int cl = (i * 64 + lane) / 48;
        const unsigned inv = (kc * WCK + cl < g.Cin) ? 0u : CC_BUF_OOB;       // OR-ed in (a select becomes a branch)
        CC_BUF_GLDS16(xr, doff[i] | inv, (unsigned)(kc * WCK) * (unsigned)g.HW * 4u, Rw + i * 256);
    };
    // transform role: thread = (tile tl16 of the wave's 16, channel c3 of a quad), two quads per chunk; this tile's 4 x 4 input block
    // starts at float roff of a channel's [4][48] image (+ 48 per input row)
    const int tl16 = lane & 15, c3 = lane >> 4;
    const int tl = wid * 16 + tl16;
    int roff;
    {
        const int r = tl16 < rlen[0] ? 0 : 1;
        const int tt = tl16 - (r ? rlen[0] : 0);
        roff = (r ? rjb[1] : rjb[0]) * 4 + (r ? rxs[1] : rxs[0]) + 2 * tt - (r ? rxal[1] : rxal[0]);
        if (tt >= (r ? rlen[1] : rlen[0])) roff = 0;          // tile past the end of the problem: any in-range address (its column is dropped)
    }
    float raw[2][16];
    auto read_raw = [&](int rr) {
        const float* src = Rw + (rr * 4 + c3) * RCH + roff;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) raw[rr][4 * a + b] = (ABL & 2) ? 0.f : src[a * RROW + b];
    };
    // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], in 8 steps per item (column b of B^T d, then row i of the
    // result) so that the main loop can spread them between its MFMAs
    float tt[2][4][4];
    auto col_step = [&](int rr, int b) {
        if constexpr (ABL & 2) return;
        const float d0 = raw[rr][b], d1 = raw[rr][4 + b], d2 = raw[rr][8 + b], d3 = raw[rr][12 + b];
        tt[rr][0][b] = d0 - d2;
        tt[rr][1][b] = d1 + d2;
        tt[rr][2][b] = d2 - d1;
        tt[rr][3][b] = d1 - d3;
    };
    auto row_step = [&](int rr, int i, int buf) {
        if constexpr (ABL & 2) return;
        float* o = Vs + buf * VBLK + rr * 256 + tl * 4 + c3;
        o[(4 * i + 0) * 512] = tt[rr][i][0] - tt[rr][i][2];
        o[(4 * i + 1) * 512] = tt[rr][i][1] + tt[rr][i][2];
        o[(4 * i + 2) * 512] = tt[rr][i][2] - tt[rr][i][1];
        o[(4 * i + 3) * 512] = tt[rr][i][1] - tt[rr][i][3];
    };
    auto dma_U = [&](int kc, int buf, int half) {
        const float* src = P.U + ((long)mb * g.nchunk + kc) * UBLK + wid * 2048 + lane * 4;
        float* dst = Us + buf * UBLK + wid * 2048;
        if constexpr (ABL & 4) return;
        if (half == 0) CC_GLDS16X4(src, dst);
        else CC_GLDS16X4(src + 1024, dst + 1024);
    };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; f++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[f][r] = 0.f;

    // the patch starts as zeros (slots that are out of range for every chunk must read as the conv's zero padding whether or not an
    // out-of-range DMA lane writes its zeros)
#pragma unroll
    for (int i = 0; i < 6; i++) *reinterpret_cast<float4*>(Rw + i * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): before the first DMA into the patch

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // One stage = one 8-channel chunk = 16 frequencies x 4 k-steps = 64 MFMAs per wave, straight-line and hand-placed: every
    // MFMA (64 cycles on the SIMD's matrix pipe) is followed by a small piece of the NEXT chunk's preparation, fenced so that
    // hipcc keeps it in that MFMA's shadow instead of clustering it in front of the matrix work:
    //   gaps 0-1    the next U block (two LDS-DMA groups of 4 KB per wave)
    //   gaps 2-7    the next chunk's raw patch of this wave (one LDS-DMA each)
    //   gap  44     wait for them (wave-private data: no barrier), gaps 44-45 read the two 4 x 4 blocks of this thread
    //   gaps 46-61  the input transform (one of 16 steps per gap) -> the other V buffer
    // and the fragments of frequency pair fp + 1 are read before the MFMAs of pair fp.  Frequencies go in pairs whose MFMAs
    // alternate (f, f+1, f, f+1 ...): two MFMAs on ONE accumulator are never adjacent, so the pieces placed between them do not
    // sit inside a dependent-accumulator pair.  The last stage prefetches a chunk that is never used (out of range / the last U
    // block again): no branch inside a stage.  PAR = parity of the stage = the LDS buffers it reads.
    auto stage = [&](auto PAR, int kc) {
        constexpr int buf = decltype(PAR)::value;
        const int kd = kc + 1 < g.nchunk ? kc + 1 : g.nchunk - 1;
        const float4* Ua = reinterpret_cast<const float4*>(Us + buf * UBLK) + lk * 64 + wm * 32 + l31;
        const float4* Vb = reinterpret_cast<const float4*>(Vs + buf * VBLK) + lk * 64 + wt * 32 + l31;
        float4 a[2][2], b[2][2];
        a[0][0] = Ua[0];   b[0][0] = Vb[0];
        a[0][1] = Ua[128]; b[0][1] = Vb[128];
#pragma unroll
        for (int fp = 0; fp < 8; fp++) {
            const int cur = fp & 1, nxt = cur ^ 1;
            if (fp < 7 && !(ABL & 16)) {
                a[nxt][0] = Ua[(2 * fp + 2) * 128]; b[nxt][0] = Vb[(2 * fp + 2) * 128];
                a[nxt][1] = Ua[(2 * fp + 3) * 128]; b[nxt][1] = Vb[(2 * fp + 3) * 128];
            } else if (fp < 7) {
                a[nxt][0] = a[cur][1]; b[nxt][0] = b[cur][1]; a[nxt][1] = a[cur][0]; b[nxt][1] = b[cur][0];
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
                if (gap < 2) dma_U(kd, buf ^ 1, gap);
                else if (gap < 8) dma_raw(kc + 1, gap - 2);
                else if (gap == 44) { CC_WAIT_VMCNT0_FENCE(); read_raw(0); }
                else if (gap == 45) read_raw(1);
                else if (gap >= 46 && gap < 62) {
                    const int st = gap - 46, rr = st >> 3, s8 = st & 7;       // steps 0-3: columns, 4-7: rows
                    if (s8 < 4) col_step(rr, s8);
                    else row_step(rr, s8 - 4, buf ^ 1);
                }
                if (gap < 8 || (gap >= 44 && gap < 62)) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // end of the stage: this wave's V writes and raw-patch reads are done (the U block and the patch were waited for at gap 44)
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };

    if (c_beg < c_end) {
        // An odd number of stages starts on parity 1 (LDS buffers 1; the peeled stage sits IN FRONT of the loop: behind it, the
        // 256 accumulators of the two paths would meet and hipcc spills them).  The prologue fills the buffers of the first parity.
        const int odd = (c_end - c_beg) & 1;
        dma_U(c_beg, odd, 0);
        dma_U(c_beg, odd, 1);
#pragma unroll
        for (int i = 0; i < 6; i++) dma_raw(c_beg, i);
        CC_WAIT_VMCNT0_FENCE();
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            read_raw(rr);
#pragma unroll
            for (int b = 0; b < 4; b++) col_step(rr, b);
#pragma unroll
            for (int i = 0; i < 4; i++) row_step(rr, i, odd);
        }
        __syncthreads();
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

    // ---- output transform Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]: this lane holds tile t = wt*32 + l31 (MFMA D column) and the
    // 16 rows m = wm*32 + (r & 3) + 8*(r >> 2) + 4*lk of every frequency
    if constexpr (ABL & 32) {
        if (acc[3][5] == 12345.f) P.y[tid] = acc[7][1] + acc[0][0];
        return;
    }
    const int q = qb * 64 + wt * 32 + l31;
    if (q >= g.Q) return;
    const int n = q / g.TPI;
    const int qr = q - n * g.TPI;
    const int ty = qr / g.TX, tx = qr - ty * g.TX;
    const int oy = 2 * ty, ox = 2 * tx;
    const int m_base = mb * WBM + wm * 32 + 4 * lk;
    const bool hr = P.res != nullptr, ha = P.add != nullptr;
    auto out_tile = [&](int r, float& y00, float& y01, float& y10, float& y11) {
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s[0][j] = (acc[j][r] + acc[4 + j][r]) + acc[8 + j][r];
            s[1][j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
        }
        y00 = (s[0][0] + s[0][1]) + s[0][2];
        y01 = (s[0][1] - s[0][2]) - s[0][3];
        y10 = (s[1][0] + s[1][1]) + s[1][2];
        y11 = (s[1][1] - s[1][2]) - s[1][3];
    };
    if constexpr (SPLIT) {
        float* pb0 = P.part + (long)blockIdx.z * g.part_stride + (((long)n * g.M) * g.Hp + oy) * g.Wp + ox;
        const long mstride = (long)g.Hp * g.Wp;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            float y00, y01, y10, y11;
            out_tile(r, y00, y01, y10, y11);
            if (m < g.M) {
                float* pb = pb0 + (long)m * mstride;
                *reinterpret_cast<float2*>(pb) = make_float2(y00, y01);
                *reinterpret_cast<float2*>(pb + g.Wp) = make_float2(y10, y11);
            }
        }
        return;
    }
    const bool row1 = oy + 1 < g.H;
    const long o0 = (long)oy * g.W + ox;
    // slope of the branch-free activation forms: t > 0 ? t : slope * t  (none 1, ReLU 0, LeakyReLU act_b, 0 -> 0.2: conv_tail.h)
    const float slope = g.act == ACT_RELU ? 0.f : (g.act == ACT_LRELU ? (g.act_b != 0.f ? g.act_b : 0.2f) : 1.f);
    auto tail = [&](float v, float r, float ad) -> float {
        if constexpr (EPI == EPI_LIN) {
            const float t = v + r;                       // r = 0 without a residual
            return t > 0.f ? t : slope * t + 0.f;        // + 0: ReLU's -0 becomes +0 (NaN stays NaN)
        } else if constexpr (EPI == EPI_GRAD) {
            const float t = v + ad;
            return r > 0.f ? t : slope * t;
        } else {
            return conv_tail(v, hr, r, g.res_mul, g.act, g.act_a, g.act_b, ad);
        }
    };
    if (g.vec2) {
        // even W, 8-byte aligned rows: both columns of the tile exist.  Every load of a half of the epilogue (bias, residual /
        // multiplier, add) is issued BEFORE that half's first store: on gfx9 one counter tracks loads and stores, so a load behind
        // a store waits for the store's round trip (two halves of 8 rows: 16 rows of operands would not fit beside the accumulators)
        float bias_r[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            const int mc = m < g.M ? m : g.M - 1;
            bias_r[r] = P.bias ? P.bias[mc] : 0.f;
        }
        float* yb = P.y + (long)n * g.y_bs + o0;
        const float* rb = hr ? P.res + (long)n * g.res_bs + o0 : nullptr;
        const float* ab = ha ? P.add + (long)n * g.add_bs + o0 : nullptr;
        const int w1 = row1 ? g.W : 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float2 res_r[8][2], add_r[8][2];
#pragma unroll
            for (int r8 = 0; r8 < 8; r8++) {
                res_r[r8][0] = res_r[r8][1] = add_r[r8][0] = add_r[r8][1] = make_float2(0.f, 0.f);
            }
            if (hr) {
#pragma unroll
                for (int r8 = 0; r8 < 8; r8++) {
                    const int r = 8 * h + r8;
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    const int mc = m < g.M ? m : g.M - 1;
                    res_r[r8][0] = *reinterpret_cast<const float2*>(rb + (long)mc * g.HW);
                    res_r[r8][1] = *reinterpret_cast<const float2*>(rb + (long)mc * g.HW + w1);
                }
            }
            if (ha) {
#pragma unroll
                for (int r8 = 0; r8 < 8; r8++) {
                    const int r = 8 * h + r8;
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    const int mc = m < g.M ? m : g.M - 1;
                    add_r[r8][0] = *reinterpret_cast<const float2*>(ab + (long)mc * g.HW);
                    add_r[r8][1] = *reinterpret_cast<const float2*>(ab + (long)mc * g.HW + w1);
                }
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; r8++) {
                const int r = 8 * h + r8;
                const int m = m_base + (r & 3) + 8 * (r >> 2);
                float y00, y01, y10, y11;
                out_tile(r, y00, y01, y10, y11);
                const float bv = bias_r[r];
                float2 o0v, o1v;
                o0v.x = tail(y00 + bv, res_r[r8][0].x, add_r[r8][0].x);
                o0v.y = tail(y01 + bv, res_r[r8][0].y, add_r[r8][0].y);
                o1v.x = tail(y10 + bv, res_r[r8][1].x, add_r[r8][1].x);
                o1v.y = tail(y11 + bv, res_r[r8][1].y, add_r[r8][1].y);
                if (m < g.M) {
                    float* yo = yb + (long)m * g.HW;
                    *reinterpret_cast<float2*>(yo) = o0v;
                    if (row1) *reinterpret_cast<float2*>(yo + g.W) = o1v;
                }
            }
        }
        return;
    }
    // odd heights / unaligned tensors: element by element (rare: not unrolled twice)
    const bool col1 = ox + 1 < g.W;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float y00, y01, y10, y11;
        out_tile(r, y00, y01, y10, y11);
        const int m = m_base + (r & 3) + 8 * (r >> 2);
        if (m >= g.M) continue;
        const float bv = P.bias ? P.bias[m] : 0.f;
        const long o = (long)m * g.HW + o0;
        float* yo = P.y + (long)n * g.y_bs + o;
        const float* ro = hr ? P.res + (long)n * g.res_bs + o : nullptr;
        const float* ao = ha ? P.add + (long)n * g.add_bs + o : nullptr;
        yo[0] = tail(y00 + bv, hr ? ro[0] : 0.f, ha ? ao[0] : 0.f);
        if (col1) yo[1] = tail(y01 + bv, hr ? ro[1] : 0.f, ha ? ao[1] : 0.f);
        if (row1) {
            yo[g.W] = tail(y10 + bv, hr ? ro[g.W] : 0.f, ha ? ao[g.W] : 0.f);
            if (col1) yo[g.W + 1] = tail(y11 + bv, hr ? ro[g.W + 1] : 0.f, ha ? ao[g.W + 1] : 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void k_wino_weights(const float* __restrict__ w, float* __restrict__ U, int M, int Cin, int Cpad,
                                                      long w_sm, long w_sc, long w0, long w_ri, long w_sj, int flip) {
    wino_weight_body(w, U, M, Cin, Cpad, w_sm, w_sc, w0, w_ri, w_sj, flip, (int)blockIdx.x);
}

}  // namespace

namespace ccint {

WinoPlan wino_plan(int B, int Cin, int H, int W, int M, int mult) {
    WinoPlan p = {};
    if (cctools::env_flag("CC_NO_WINO")) return p;
    // where it pays (measured per layer shape, profiles/r04_wino_layers.txt): enough output rows to fill half a 64-row tile, enough
    // reduction channels to amortise the workgroup's prologue / output transform, enough tiles to fill a 64-tile block
    const int TY = (H + 1) / 2, TX = (W + 1) / 2;
    const long Q = (long)B * TY * TX;
    if (M < cctools::env_int("CC_WINO_MINM", 33) || Cin < cctools::env_int("CC_WINO_MINC", 24) || Q < cctools::env_int("CC_WINO_MINQ", 48)) return p;
    // the raw-input path stages 16-byte pieces of aligned rows, one or two tile-row runs per wave (16 tiles)
    if (H < 2 || (W % 4) != 0 || !(TX >= 16 || TX == 8)) return p;
    if (Q > (1l << 30)) return p;
    p.ok = 1;
    p.Mpad = ((M + WBM - 1) / WBM) * WBM;
    p.Cpad = ((Cin + WCK - 1) / WCK) * WCK;
    p.nchunk = p.Cpad / WCK;
    p.TY = TY; p.TX = TX;
    p.nqb = (int)((Q + 63) / 64);
    p.nmb = p.Mpad / WBM;
    p.Hp = 2 * TY; p.Wp = 2 * TX;
    p.u_floats = (size_t)16 * p.Cpad * p.Mpad;
    // split-K: one workgroup per CU (256 registers of accumulators per lane), so a launch runs in ceil(workgroups / 256) rounds of
    // `chunks per workgroup` stages; pick the split that minimises rounds * (stages + fixed per-workgroup cost), the partial slabs
    // and the epilogue launch charged as a few stages
    const long blocks = (long)p.nqb * p.nmb * (mult > 1 ? mult : 1);
    p.nsplit = 1;
    p.cps = p.nchunk;
    if (blocks < cctools::env_int("CC_WINO_SPLIT_BELOW", 224) && p.nchunk >= 4) {
        const double fixed = 1.5;       // prologue + output transform, in stages
        double best = 1e30;
        int best_ns = 1;
        const int cap = p.nchunk / 2 < 16 ? p.nchunk / 2 : 16;
        for (int ns = 1; ns <= cap; ns++) {
            const int cps = (p.nchunk + ns - 1) / ns;
            const int real = (p.nchunk + cps - 1) / cps;
            const double rounds = (double)((blocks * real + 255) / 256);
            const double t = rounds * (cps + fixed) + (real > 1 ? 1.0 + 0.25 * real : 0.0);
            if (t < best - 1e-9) { best = t; best_ns = real; p.cps = cps; }
        }
        p.nsplit = best_ns;
    }
    p.part_floats = p.nsplit > 1 ? (size_t)p.nsplit * B * M * p.Hp * p.Wp : 0;
    return p;
}

void wino_weights_launch(const float* w, float* U, int M, int Cin, int Cpad, int Mpad, long w_sm, long w_sc, long w0, long w_ri,
                         long w_sj, int flip, hipStream_t s) {
    hipLaunchKernelGGL(k_wino_weights, dim3((unsigned)wino_weight_blocks(Mpad, Cpad)), dim3(256), 0, s, w, U, M, Cin, Cpad, w_sm, w_sc,
                       w0, w_ri, w_sj, flip);
}

bool wino_launch(const WinoGeom& gg, const WinoPlan& p, const WinoProb* probs, int nprob, hipStream_t s) {
    // 16-byte pieces of x: base and batch stride 16-byte aligned (dense NCHW tensors and channel slices of them are: W % 4 == 0),
    // and byte offsets below the out-of-range marker of the buffer resource
    const long xfl = ((long)gg.B - 1) * gg.x_bs + (long)gg.Cin * gg.H * gg.W;
    if ((gg.x_bs % 4) != 0 || xfl * 4 >= (long)CC_BUF_OOB) return false;
    for (int k = 0; k < nprob; k++)
        if (((uintptr_t)probs[k].x % 16) != 0) return false;
    WN a = {};
    for (int k = 0; k < nprob; k++) a.p[k] = probs[k];
    a.per_prob = p.nqb * p.nmb;
    a.total = a.per_prob * nprob;
    a.nmb = p.nmb;
    a.Cin = gg.Cin; a.H = gg.H; a.W = gg.W; a.HW = gg.H * gg.W;
    a.x_bytes = (unsigned)((((long)gg.B - 1) * gg.x_bs + (long)gg.Cin * gg.H * gg.W) * 4);
    a.x_bs = gg.x_bs;
    a.M = gg.M; a.y_bs = gg.y_bs; a.res_bs = gg.res_bs; a.add_bs = gg.add_bs;
    a.TX = p.TX; a.TPI = p.TY * p.TX; a.Q = gg.B * a.TPI; a.nchunk = p.nchunk; a.cps = p.cps;
    a.Hp = p.Hp; a.Wp = p.Wp;
    a.part_stride = (long)gg.B * gg.M * p.Hp * p.Wp;
    a.act = gg.act; a.act_a = gg.act_a; a.act_b = gg.act_b; a.res_mul = gg.res_mul;
    bool v2 = (gg.W % 2 == 0) && (gg.y_bs % 2 == 0) && (gg.res_bs % 2 == 0) && (gg.add_bs % 2 == 0);
    for (int k = 0; k < nprob; k++)
        v2 = v2 && ((((uintptr_t)probs[k].y) | (uintptr_t)probs[k].res | (uintptr_t)probs[k].add) % 8 == 0);
    a.vec2 = v2 ? 1 : 0;
    if (cctools::env_flag("CC_WINO_TRACE"))
        fprintf(stderr, "wino: %dx[B%d M%d C%d %dx%d] nqb %d nmb %d nsplit %d cps %d act %d res_mul %d vec2 %d\n", nprob, gg.B, gg.M, gg.Cin,
                gg.H, gg.W, p.nqb, p.nmb, p.nsplit, p.cps, gg.act, gg.res_mul, a.vec2);
    const size_t smem = (size_t)(2 * UBLK + 2 * VBLK + 4 * RWAVE) * sizeof(float);
    dim3 grid((unsigned)(((a.total + 7) / 8) * 8), 1, (unsigned)p.nsplit);
    const bool grad = gg.res_mul != 0;          // (every problem of a launch has the multiplier or none has: conv.hip same_problem_shape)
    const int epi = (!grad && (gg.act == ACT_NONE || gg.act == ACT_RELU || gg.act == ACT_LRELU)) ? EPI_LIN
                    : (grad && probs[0].res && (gg.act == ACT_RELU || gg.act == ACT_LRELU)) ? EPI_GRAD : EPI_GEN;
    auto go = [&](auto kern, bool& attr) {
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    };
    static bool attr_set[4] = {false, false, false, false};
#ifdef CC_TOOLS
    const int abl = cctools::env_int("CC_WINO_ABL", 0);
    static bool attr_abl[16] = {};
    if (abl && p.nsplit == 1 && epi == EPI_LIN) {
        switch (abl) {
            case 1: go(k_wino_f2x3<0, 0, 1>, attr_abl[1]); return true;
            case 2: go(k_wino_f2x3<0, 0, 2>, attr_abl[2]); return true;
            case 3: go(k_wino_f2x3<0, 0, 3>, attr_abl[3]); return true;
            case 4: go(k_wino_f2x3<0, 0, 4>, attr_abl[4]); return true;
            case 7: go(k_wino_f2x3<0, 0, 7>, attr_abl[5]); return true;
            case 8: go(k_wino_f2x3<0, 0, 8>, attr_abl[6]); return true;
            case 16: go(k_wino_f2x3<0, 0, 16>, attr_abl[7]); return true;
            case 23: go(k_wino_f2x3<0, 0, 23>, attr_abl[8]); return true;
            case 32: go(k_wino_f2x3<0, 0, 32>, attr_abl[9]); return true;
            default: break;
        }
    }
#endif
    if (p.nsplit > 1) go(k_wino_f2x3<1, 0, 0>, attr_set[3]);
    else if (epi == EPI_LIN) go(k_wino_f2x3<0, EPI_LIN, 0>, attr_set[0]);
    else if (epi == EPI_GRAD) go(k_wino_f2x3<0, EPI_GRAD, 0>, attr_set[1]);
    else go(k_wino_f2x3<0, EPI_GEN, 0>, attr_set[2]);
    return true;
}

}  // namespace ccint
