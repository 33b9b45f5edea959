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
    int mb_major;      // work-item order inside a problem: 0 = tile block, then m block (an XCD's run shares INPUT tiles); 1 = m block, then
                       // tile block (it shares WEIGHT blocks: the deep layers, whose weights are the larger operand) -- wino_launch
};

constexpr int VBLK = 16 * WCK * 64;       // floats of one V chunk (64 tiles)
// raw input patch of ONE wave (16 consecutive tiles x 4 channels = the items it transforms) for one chunk:
// [channel 4][input row 4][12 float4] floats
constexpr int RROW = 12 * 4, RCH = 4 * RROW, RWAVE = 4 * RCH;      // 48, 192, 768 floats (3 KB per wave)
constexpr int WTHREADS = 512;

// epilogue forms (template parameter EPI; chosen on the host): branch-free code for what the step uses, the generic tail otherwise
//   EPI_LIN : y = act(v + bias + res),       act in {none, ReLU, LeakyReLU}  as  t > 0 ? t : slope * t   (slope 1 / 0 / s)
//   EPI_GRAD: y = (v + add) * act'(res),     act in {ReLU, LeakyReLU}        as  res > 0 ? t : slope * t
//   EPI_GEN : conv_tail() (sigmoid forms)
enum { EPI_LIN = 0, EPI_GRAD = 1, EPI_GEN = 2 };

// Workgroup = 8 waves, two per SIMD.  Wave w = (sub-tile st = w & 3: 32 m x 32 tiles, frequency half fh = w >> 2: rows 2 fh, 2 fh + 1
// of the 4 x 4 frequency matrix = 8 accumulator tiles = 128 registers).  The two waves of a SIMD (st, 0) and (st, 1) run the same
// stream; while one of them is held up issuing a VMEM instruction (an LDS-DMA costs the issuing wave 100-180 cycles of its in-order
// stream, measured: profiles/r04_wino_probe_a.txt) the other one feeds the matrix pipe -- with ONE wave per SIMD and all 16
// frequencies in it (256 accumulator registers, the first version of this kernel) every such cycle was lost to the MFMAs.
// ABL (tools build only, CC_WINO_ABL): timing ablations that compute garbage -- 1: no raw-patch DMA, 2: no input transform / V
// writes, 4: no U DMA, 8: no MFMAs, 16: no fragment reads, 32: no epilogue
template <int SPLIT, int EPI, int ABL>
__global__ __launch_bounds__(WTHREADS, 2) void k_wino_f2x3(WN g) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* Us = smem;                          // [2][UBLK]
    float* Vs = smem + 2 * UBLK;               // [2][VBLK]
    float* Rs = smem + 2 * UBLK + 2 * VBLK;    // [8 waves][RWAVE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st = wid & 3, fh = wid >> 2;
    const int wm = st >> 1, wt = st & 1;
    const int l31 = lane & 31, lk = lane >> 5;

    // workgroup -> (problem, tile block, m block); blocks b, b + 8, b + 16 ... run on one XCD: give them consecutive work items
    const int cnt = (int)gridDim.x >> 3;
    const int w = ((int)blockIdx.x & 7) * cnt + ((int)blockIdx.x >> 3);
    if (w >= g.total) return;
    const int prob = w / g.per_prob;
    const int wr = w - prob * g.per_prob;
    int qb, mb;
    if (g.mb_major) { const int nqb = g.per_prob / g.nmb; mb = wr / nqb; qb = wr - mb * nqb; }
    else { qb = wr / g.nmb; mb = wr - qb * g.nmb; }
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
    // An odd number of stages starts on parity 1 (LDS buffers 1; the peeled stage sits IN FRONT of the loop: behind it, the
    // accumulators of the two paths would meet and hipcc copies them).  The prologue fills the buffers of the first parity.
    const int odd = (c_end - c_beg) & 1;
    auto dma_U = [&](int kc, int buf) {      // this wave's eighth (4 KB) of the 32 KB block
        if constexpr (ABL & 4) return;
        const float* src = P.U + ((long)mb * g.nchunk + kc) * UBLK + wid * 1024 + lane * 4;
        CC_GLDS16X4(src, Us + buf * UBLK + wid * 1024);
    };

    if (c_beg < c_end) dma_U(c_beg, odd);      // first request of the workgroup: in flight while the patch geometry is worked out
    // bias of this wave's 32 output rows: ONE load now (lane l31 -> row wm*32 + l31), handed to the rows' owners by lane shuffles in
    // the epilogue (eight scalar loads per lane there cost eight VMEM issues and a memory round trip per workgroup)
    float bias_w = 0.f;
    if constexpr (!SPLIT) {
        const int mr = mb * WBM + wm * 32 + l31;
        if (P.bias) bias_w = P.bias[mr < g.M ? mr : g.M - 1];
    }

    // ---- input path.  Every VMEM instruction is expensive for the wave that issues it, so the raw input goes the way that needs
    // the fewest: 16-byte LDS-DMA.  Transform role of wave w: the 16 consecutive tiles tg = w & 3 of the block x the channel quad
    // qd = w >> 2 of the chunk (one 4 x 4 input block per thread); it stages exactly the input rows of those items itself
    // (wave-private: its own vmcnt wait, no barrier).  The 16 tiles form one or two runs inside a tile row (TX >= 16, or TX == 8
    // and two whole rows); per channel and input row a run is 2 len + 2 floats, fetched as the 16-byte-aligned float4s that cover
    // it (rows of x are 16-byte aligned: W % 4 == 0) -- at most 12 float4 per (channel, row), 192 per chunk = THREE DMA
    // instructions per wave.  Out-of-image float4s and channels past Cin are out-of-range offsets of the buffer resource (zeros).
    const int tg = st, qd = fh;
    float* Rw = Rs + wid * RWAVE;
    const int q0 = qb * 64 + 16 * tg;
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
    unsigned doff[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int piece = i * 64 + lane;                   // (channel of the quad, input row, float4 slot) = (piece / 48, piece % 48 / 12, piece % 12)
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
        const int cb = kc * WCK + qd * 4;
        const unsigned inv = (cb + cl < g.Cin) ? 0u : CC_BUF_OOB;       // OR-ed in (a select becomes a branch)
        CC_BUF_GLDS16(xr, doff[i] | inv, (unsigned)cb * (unsigned)g.HW * 4u, Rw + i * 256);
    };
    // this thread's item: tile tl16 of the wave's 16, channel c3 of its quad; the 4 x 4 input block starts at float roff of the
    // channel's [4][48] image (+ 48 per input row)
    const int tl16 = lane & 15, c3 = lane >> 4;
    const int tl = tg * 16 + tl16;
    int roff;
    {
        const int r = tl16 < rlen[0] ? 0 : 1;
        const int tt = tl16 - (r ? rlen[0] : 0);
        roff = (r ? rjb[1] : rjb[0]) * 4 + (r ? rxs[1] : rxs[0]) + 2 * tt - (r ? rxal[1] : rxal[0]);
        if (tt >= (r ? rlen[1] : rlen[0])) roff = 0;          // tile past the end of the problem: any in-range address (its column is dropped)
    }
    float raw[16];
    auto read_raw = [&]() {
        const float* src = Rw + c3 * RCH + roff;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) raw[4 * a + b] = (ABL & 2) ? 0.f : src[a * RROW + b];
    };
    // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], in 8 steps (column b of B^T d, then row i of the result) so
    // that the main loop can spread them between its MFMAs
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
        float* o = Vs + buf * VBLK + qd * 256 + tl * 4 + c3;
        o[(4 * i + 0) * 512] = tt[i][0] - tt[i][2];
        o[(4 * i + 1) * 512] = tt[i][1] + tt[i][2];
        o[(4 * i + 2) * 512] = tt[i][2] - tt[i][1];
        o[(4 * i + 3) * 512] = tt[i][1] - tt[i][3];
    };
    // the patch starts as zeros (slots that are out of range for every chunk must read as the conv's zero padding whether or not an
    // out-of-range DMA lane writes its zeros)
#pragma unroll
    for (int i = 0; i < 3; i++) *reinterpret_cast<float4*>(Rw + i * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): before the first DMA into the patch
    __builtin_amdgcn_wave_barrier();
    if (c_beg < c_end) {
#pragma unroll
        for (int i = 0; i < 3; i++) dma_raw(c_beg, i);
    }

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; f++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[f][r] = 0.f;

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // One stage = one 8-channel chunk = 8 frequencies x 4 k-steps = 32 MFMAs per wave (64 per SIMD), straight-line and hand-placed:
    // every MFMA is followed by a small piece of the NEXT chunk's preparation, fenced so that hipcc keeps it in that MFMA's shadow
    // instead of clustering it in front of the matrix work:
    //   gap  0 / 2    this wave's eighth of the next U block (one LDS-DMA group of 4 x 1 KB; frequency half 0 / 1)
    //   gap  18       the next chunk's raw patch of this wave -- requested a whole stage ago -- has landed (a counted wait:
    //                 wave-private data, no barrier); read this thread's 4 x 4 block
    //   gaps 19-26    the input transform (one of 8 steps per gap) -> the other V buffer
    //   gap  27 / 29  request the patch of the chunk after the next (three LDS-DMA; the patch buffer is free again).  A request
    //                 that is consumed inside the stage it is issued in leaves ~2 000 cycles for L2 / HBM and the wave waits at
    //                 the fence most of the time (measured: DMA removed = -650 / -890 cycles per stage)
    // and the fragments of frequency pair fp + 1 are read before the MFMAs of pair fp.  Frequencies go in pairs whose MFMAs
    // alternate (f, f+1, f, f+1 ...): two MFMAs on ONE accumulator are never adjacent, so the pieces placed between them do not
    // sit inside a dependent-accumulator pair.  The last stage prefetches a chunk that is never used (out of range / the last U
    // block again): no branch inside a stage.  PAR = parity of the stage = the LDS buffers it reads.
    auto stage = [&](auto PAR, int kc) {
        constexpr int buf = decltype(PAR)::value;
        const int kd = kc + 1 < g.nchunk ? kc + 1 : g.nchunk - 1;
        const float4* Ua = reinterpret_cast<const float4*>(Us + buf * UBLK) + fh * 1024 + lk * 64 + wm * 32 + l31;
        const float4* Vb = reinterpret_cast<const float4*>(Vs + buf * VBLK) + fh * 1024 + lk * 64 + wt * 32 + l31;
        float4 a[2][2], b[2][2];
        a[0][0] = Ua[0];   b[0][0] = Vb[0];
        a[0][1] = Ua[128]; b[0][1] = Vb[128];
#pragma unroll
        for (int fp = 0; fp < 4; fp++) {
            const int cur = fp & 1, nxt = cur ^ 1;
            if (fp < 3 && !(ABL & 16)) {
                a[nxt][0] = Ua[(2 * fp + 2) * 128]; b[nxt][0] = Vb[(2 * fp + 2) * 128];
                a[nxt][1] = Ua[(2 * fp + 3) * 128]; b[nxt][1] = Vb[(2 * fp + 3) * 128];
            } else if (fp < 3) {
                a[nxt][0] = a[cur][1]; b[nxt][0] = b[cur][1]; a[nxt][1] = a[cur][0]; b[nxt][1] = b[cur][0];
            }
            if constexpr (!(ABL & 64)) __builtin_amdgcn_sched_barrier(0);
            const float av[2][4] = {{a[cur][0].x, a[cur][0].y, a[cur][0].z, a[cur][0].w}, {a[cur][1].x, a[cur][1].y, a[cur][1].z, a[cur][1].w}};
            const float bv[2][4] = {{b[cur][0].x, b[cur][0].y, b[cur][0].z, b[cur][0].w}, {b[cur][1].x, b[cur][1].y, b[cur][1].z, b[cur][1].w}};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int f = 2 * fp + (i & 1), j = i >> 1;
                if constexpr (ABL & 8) acc[f][0] = fmaf(av[i & 1][j], bv[i & 1][j], acc[f][0]);
                else acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 1][j], bv[i & 1][j], acc[f], 0, 0, 0);
                const int gap = 8 * fp + i;
                // (the two waves of a SIMD, fh = 0 / 1, issue their DMA in different gaps; uniform branches: fh is wave-uniform)
                constexpr int GU = (ABL & 128) ? 8 : 0;       // variant: U requests later in the stage
                if (gap == GU) { if (fh == 0) dma_U(kd, buf ^ 1); }
                else if (gap == GU + 2) { if (fh == 1) dma_U(kd, buf ^ 1); }
                else if (gap == 18) { CC_WAIT_VMCNT_FENCE(4); read_raw(); }       // the patch (requested one stage ago) is in; the U group may be out
                else if (gap >= 19 && gap < 27) {
                    const int s8 = gap - 19;                   // steps 0-3: columns, 4-7: rows
                    if (s8 < 4) col_step(s8);
                    else row_step(s8 - 4, buf ^ 1);
                }
                else if (gap == 27) { if (fh == 0) { dma_raw(kc + 2, 0); dma_raw(kc + 2, 1); dma_raw(kc + 2, 2); } }
                else if (gap == 29) { if (fh == 1) { dma_raw(kc + 2, 0); dma_raw(kc + 2, 1); dma_raw(kc + 2, 2); } }
                if constexpr (!(ABL & 64))
                    if (gap == GU || gap == GU + 2 || (gap >= 18 && gap < 30)) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(ABL & 64)) __builtin_amdgcn_sched_barrier(0);
        }
        // end of the stage: this wave's V writes are done and its part of the next U block has landed (it is older than the three
        // patch requests, which stay in flight across the barrier: memory reads return in issue order)
        if constexpr (ABL & 1) CC_WAIT_VMCNT0_FENCE();
        else CC_WAIT_VMCNT_FENCE(3);
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };

    if (c_beg < c_end) {
        CC_WAIT_VMCNT0_FENCE();
        read_raw();
#pragma unroll
        for (int b = 0; b < 4; b++) col_step(b);
#pragma unroll
        for (int i = 0; i < 4; i++) row_step(i, odd);
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the patch has been read
        __builtin_amdgcn_wave_barrier();         // (no instruction; the CPU emulation of the kernels runs a wave's lanes one after the other)
#pragma unroll
        for (int i = 0; i < 3; i++) dma_raw(c_beg + 1, i);       // stays in flight across the barrier
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

    // ---- output transform Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]].  This lane holds tile t = wt*32 + l31 (MFMA D column) and the 16
    // rows m = wm*32 + (r & 3) + 8*(r >> 2) + 4*lk of frequency rows 2 fh, 2 fh + 1.  A^T M is linear in the rows of M: each wave
    // forms its part P = A^T[:, 2fh : 2fh+2] M[2fh : 2fh+2, :] (2 x 4), applies the column transform (P A: 2 x 2) and the two parts
    // of a sub-tile are added: the wave finishes rows r in [8 fh, 8 fh + 8) and hands the parts of the other eight rows to its
    // partner through LDS (the U buffers are free: every wave has passed the last stage's barrier).
    if constexpr (ABL & 32) {
        if (acc[3][5] == 12345.f) P.y[tid] = acc[7][1] + acc[0][0];
        return;
    }
    auto part_tile = [&](int r, float& z00, float& z01, float& z10, float& z11) {
        float p0[4], p1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (fh == 0) { p0[j] = acc[j][r] + acc[4 + j][r]; p1[j] = acc[4 + j][r]; }           // rows 0, 1:  (m0 + m1, m1)
            else { p0[j] = acc[j][r]; p1[j] = (0.f - acc[j][r]) - acc[4 + j][r]; }                // rows 2, 3:  (m2, -m2 - m3)
        }
        z00 = (p0[0] + p0[1]) + p0[2];
        z01 = (p0[1] - p0[2]) - p0[3];
        z10 = (p1[0] + p1[1]) + p1[2];
        z11 = (p1[1] - p1[2]) - p1[3];
    };
    // output geometry of this lane.  Lanes 2k / 2k + 1 hold horizontally adjacent tiles of one tile row (TX is even), i.e. 4 x 2
    // pixels: the even lane takes row oy of the pair, the odd lane row oy + 1, and every access of the epilogue (residual /
    // multiplier, add, the result) is ONE 16-byte operation per lane and output channel instead of two 8-byte ones.  The loads are
    // issued here, in front of the exchange, so that their latency runs under it, and in front of the first store: on gfx9 one
    // counter tracks loads and stores, so a load behind a store waits for the store's round trip.
    const int q = qb * 64 + wt * 32 + l31;
    const bool qv = q < g.Q;
    const int qq = qv ? q : 0;
    const int n = qq / g.TPI;
    const int qr = qq - n * g.TPI;
    const int ty = qr / g.TX, tx = qr - ty * g.TX;
    const int oy = 2 * ty, ox = 2 * tx;
    const int m_base = mb * WBM + wm * 32 + 4 * lk;
    const bool hr = P.res != nullptr, ha = P.add != nullptr;
    const bool row1 = oy + 1 < g.H;
    const long o0 = (long)oy * g.W + ox;
    const int odd_lane = lane & 1;
    // this lane's 16-byte piece: row oy + odd_lane, columns 4 * (tx >> 1) .. + 3
    const long o4 = (long)(oy + odd_lane) * g.W + (ox - 2 * odd_lane);
    const bool row_ok = odd_lane ? row1 : true;
    float4 res_r[8], add_r[8];
    if constexpr (!SPLIT) {
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) res_r[r8] = add_r[r8] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.vec2 && qv && row_ok) {
            if (hr) {
                const float* rb = P.res + (long)n * g.res_bs + o4;
#pragma unroll
                for (int r8 = 0; r8 < 8; r8++) {
                    const int r = 8 * fh + r8;
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    res_r[r8] = *reinterpret_cast<const float4*>(rb + (long)(m < g.M ? m : g.M - 1) * g.HW);
                }
            }
            if (ha) {
                const float* ab = P.add + (long)n * g.add_bs + o4;
#pragma unroll
                for (int r8 = 0; r8 < 8; r8++) {
                    const int r = 8 * fh + r8;
                    const int m = m_base + (r & 3) + 8 * (r >> 2);
                    add_r[r8] = *reinterpret_cast<const float4*>(ab + (long)(m < g.M ? m : g.M - 1) * g.HW);
                }
            }
        }
    }
    float* Xs = smem;                                          // [sub-tile 4][sender fh 2][row 8][value 4][lane 64]
    {
        float* xo = Xs + ((st * 2 + fh) * 32) * 64 + lane;
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) {
            const int r = 8 * (fh ^ 1) + r8;                   // the partner's rows
            float z00, z01, z10, z11;
            part_tile(r, z00, z01, z10, z11);
            xo[(r8 * 4 + 0) * 64] = z00;
            xo[(r8 * 4 + 1) * 64] = z01;
            xo[(r8 * 4 + 2) * 64] = z10;
            xo[(r8 * 4 + 3) * 64] = z11;
        }
    }
    __syncthreads();
    const float* xi = Xs + ((st * 2 + (fh ^ 1)) * 32) * 64 + lane;
    auto out_tile = [&](int r8, float& y00, float& y01, float& y10, float& y11) {
        part_tile(8 * fh + r8, y00, y01, y10, y11);
        y00 += xi[(r8 * 4 + 0) * 64];
        y01 += xi[(r8 * 4 + 1) * 64];
        y10 += xi[(r8 * 4 + 2) * 64];
        y11 += xi[(r8 * 4 + 3) * 64];
    };
    // the pair exchange: the even lane hands its row oy + 1 to the odd lane and receives the odd lane's row oy -> (4 pixels of one
    // row) per lane.  Every lane of the wave takes part (lanes past the end of the problem carry garbage that is never stored).
    auto pair4 = [&](float y00, float y01, float y10, float y11) -> float4 {
        const float sx = odd_lane ? y00 : y10, sy = odd_lane ? y01 : y11;
        const float rx = __shfl_xor(sx, 1), ry = __shfl_xor(sy, 1);
        return odd_lane ? make_float4(rx, ry, y10, y11) : make_float4(y00, y01, rx, ry);
    };
    if constexpr (SPLIT) {
        float* pb0 = P.part + (long)blockIdx.z * g.part_stride + (((long)n * g.M) * g.Hp + oy + odd_lane) * g.Wp + (ox - 2 * odd_lane);
        const long mstride = (long)g.Hp * g.Wp;
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) {
            const int r = 8 * fh + r8;
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            float y00, y01, y10, y11;
            out_tile(r8, y00, y01, y10, y11);
            const float4 v = pair4(y00, y01, y10, y11);
            if (qv && m < g.M) *reinterpret_cast<float4*>(pb0 + (long)m * mstride) = v;      // (slab rows 2 TY x 2 TX: both rows exist)
        }
        return;
    }
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
        float* yb = P.y + (long)n * g.y_bs + o4;
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) {
            const int r = 8 * fh + r8;
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            float y00, y01, y10, y11;
            out_tile(r8, y00, y01, y10, y11);
            const float bv = __shfl(bias_w, 4 * lk + (r & 3) + 8 * (r >> 2));      // row m of this lane: lane m - (mb*64 + wm*32) of the wave's load
            float4 v = pair4(y00, y01, y10, y11);
            v.x = tail(v.x + bv, res_r[r8].x, add_r[r8].x);
            v.y = tail(v.y + bv, res_r[r8].y, add_r[r8].y);
            v.z = tail(v.z + bv, res_r[r8].z, add_r[r8].z);
            v.w = tail(v.w + bv, res_r[r8].w, add_r[r8].w);
            if (qv && row_ok && m < g.M) *reinterpret_cast<float4*>(yb + (long)m * g.HW) = v;
        }
        return;
    }
    if (!qv) return;
    // odd heights / unaligned tensors: element by element (rare)
    const bool col1 = ox + 1 < g.W;
#pragma unroll
    for (int r8 = 0; r8 < 8; r8++) {
        const int r = 8 * fh + r8;
        float y00, y01, y10, y11;
        out_tile(r8, y00, y01, y10, y11);
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

// ---- Small-tile instance: 32 output channels x 32 tiles (128 output pixels) x 16 frequencies per workgroup, for launches whose 64 x 64
// blocks cannot fill the chip (the <= 32x104 levels: 52-208 blocks on 256 CUs, which used to go split-K -> partial slabs -> a separate
// epilogue launch).  A quarter of the block area = 4x the workgroups at FULL reduction depth, bias / residual / activation fused, no
// slabs.  KS = 1: four waves, one per SIMD; wave fr owns frequency ROW fr (f = 4 fr + j: four accumulator tiles = 64 registers), 76 KB
// of LDS -> two workgroups per CU: one's prologue / output transform runs under the other's MFMAs, and the two waves of a SIMD are in
// unrelated phases of their stages.  KS = 2 (in-workgroup split of the reduction): eight waves = frequency row x HALF of the channel
// chunks, each half with its own U / V / patch buffers (152 KB, one workgroup per CU, two waves per SIMD); the halves meet in the
// output transform's LDS exchange -- for problems that have <= 256 blocks even at this tile size (16x52 with 256 channels: 208).
// Same stage structure as the 64 x 64 kernel (U by LDS-DMA, wave-private raw patches by bounds-checked buffer LDS-DMA one stage ahead,
// input transform between the MFMAs); U comes from the SAME weight image: the 512-byte halves of the 1 KB [f][quad] rows of a 64-row
// block (CC_GLDS16X4_S).  Output transform: columns inside the wave (M[fr][:] A), rows across the waves through LDS.
constexpr int SBM = 32, SBT = 32;
constexpr int UBLK_S = 16 * WCK * SBM;        // floats of one U chunk of a 32-row block (16 KB)
constexpr int VBLK_S = 16 * WCK * SBT;
constexpr int SHALF = 2 * UBLK_S + 2 * VBLK_S + 4 * RWAVE;      // LDS floats of one reduction half: U / V double-buffered + four patches

template <int KS, int SPLIT, int EPI>
__global__ __launch_bounds__(256 * KS, 2) void k_wino_f2x3_s(WN g) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = wid & 3, kh = wid >> 2;           // frequency row of this wave's MFMAs; reduction half (KS = 2)
    const int l31 = lane & 31, lk = lane >> 5;
    float* Us = smem + kh * SHALF;                   // [2][UBLK_S]
    float* Vs = Us + 2 * UBLK_S;                     // [2][VBLK_S]
    float* Rs = Vs + 2 * VBLK_S;                     // [4 waves][RWAVE]

    const int cnt = (int)gridDim.x >> 3;
    const int w = ((int)blockIdx.x & 7) * cnt + ((int)blockIdx.x >> 3);
    if (w >= g.total) return;
    const int prob = w / g.per_prob;
    const int wr = w - prob * g.per_prob;
    int qb, mb;
    if (g.mb_major) { const int nqb = g.per_prob / g.nmb; mb = wr / nqb; qb = wr - mb * nqb; }
    else { qb = wr / g.nmb; mb = wr - qb * g.nmb; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const ccint::WinoProb& P = *(reinterpret_cast<const ccint::WinoProb*>(ka + offsetof(WN, p)) + prob);
#else
    const ccint::WinoProb& P = g.p[prob];
#endif
    // chunk range of this launch slice (blockIdx.z: split-K across workgroups), then of this wave's half of it
    int c_beg = (int)blockIdx.z * g.cps;
    int c_end = c_beg + g.cps;
    if (c_end > g.nchunk) c_end = g.nchunk;
    if constexpr (KS == 2) {       // (the host gives KS = 2 launches an even number of chunks per slice: both halves run the same number of barriers)
        const int half = (c_end - c_beg) >> 1;
        c_beg += kh * half;
        c_end = c_beg + half;
    }
    const int odd = (c_end - c_beg) & 1;
    // U: rows f = 4 fr + k of the block, both channel quads (lanes 0-31 / 32-63), this block's 32-row half of the 64-row weight block
    unsigned uoff[4];
#pragma unroll
    for (int k = 0; k < 4; k++) uoff[k] = (unsigned)(fr * 8192 + k * 1024 + lk * 1024 + (mb & 1) * 512 + l31 * 16);
    auto dma_U = [&](int kc, int buf) {
        const float* src = P.U + ((long)(mb >> 1) * g.nchunk + kc) * UBLK;
        CC_GLDS16X4_S(src, uoff, Us + buf * UBLK_S + fr * 1024);
    };
    if (c_beg < c_end) dma_U(c_beg, odd);
    float bias_w = 0.f;
    if constexpr (!SPLIT) {
        const int mr = mb * SBM + l31;
        if (P.bias) bias_w = P.bias[mr < g.M ? mr : g.M - 1];
    }

    // ---- input path (as in the 64 x 64 kernel): transform role of wave fr = the 16 consecutive tiles tg = fr & 1 of the block x the
    // channel quad qd = fr >> 1 of the chunk
    const int tg = fr & 1, qd = fr >> 1;
    float* Rw = Rs + fr * RWAVE;
    const int q0 = qb * SBT + 16 * tg;
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
            const int xal = xs & ~3;
            rn[r] = n; riy[r] = 2 * ty - 1; rxal[r] = xal; rlen[r] = len; rjb[r] = jb; rxs[r] = xs;
            jb += len > 0 ? (xs + 2 * len + 2 - xal + 3) >> 2 : 0;
            q += len;
        }
    }
    unsigned doff[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int piece = i * 64 + lane;
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
        const int cl = (i * 64 + lane) / 48;
        const int cb = kc * WCK + qd * 4;
        // (chunks past this wave's range are never consumed: their requests move zeros / garbage into a patch nobody reads)
        const unsigned inv = (cb + cl < g.Cin) ? 0u : CC_BUF_OOB;
        CC_BUF_GLDS16(xr, doff[i] | inv, (unsigned)cb * (unsigned)g.HW * 4u, Rw + i * 256);
    };
    const int tl16 = lane & 15, c3 = lane >> 4;
    const int tl = tg * 16 + tl16;
    int roff;
    {
        const int r = tl16 < rlen[0] ? 0 : 1;
        const int tt = tl16 - (r ? rlen[0] : 0);
        roff = (r ? rjb[1] : rjb[0]) * 4 + (r ? rxs[1] : rxs[0]) + 2 * tt - (r ? rxal[1] : rxal[0]);
        if (tt >= (r ? rlen[1] : rlen[0])) roff = 0;
    }
    float raw[16];
    auto read_raw = [&]() {
        const float* src = Rw + c3 * RCH + roff;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) raw[4 * a + b] = src[a * RROW + b];
    };
    float tt[4][4];
    auto col_step = [&](int b) {
        const float d0 = raw[b], d1 = raw[4 + b], d2 = raw[8 + b], d3 = raw[12 + b];
        tt[0][b] = d0 - d2;
        tt[1][b] = d1 + d2;
        tt[2][b] = d2 - d1;
        tt[3][b] = d1 - d3;
    };
    auto row_step = [&](int i, int buf) {
        float* o = Vs + buf * VBLK_S + qd * 128 + tl * 4 + c3;
        o[(4 * i + 0) * 256] = tt[i][0] - tt[i][2];
        o[(4 * i + 1) * 256] = tt[i][1] + tt[i][2];
        o[(4 * i + 2) * 256] = tt[i][2] - tt[i][1];
        o[(4 * i + 3) * 256] = tt[i][1] - tt[i][3];
    };
#pragma unroll
    for (int i = 0; i < 3; i++) *reinterpret_cast<float4*>(Rw + i * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): before the first DMA into the patch
    __builtin_amdgcn_wave_barrier();
    if (c_beg < c_end) {
#pragma unroll
        for (int i = 0; i < 3; i++) dma_raw(c_beg, i);
    }

    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[f][r] = 0.f;

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // One stage = one 8-channel chunk = 4 frequencies x 4 k-steps = 16 MFMAs per wave; the next chunk's preparation sits in their gaps:
    //   gap 0       this wave's quarter of the next U block (4 LDS-DMA rows)
    //   gap 3       the next chunk's raw patch (requested a stage ago) has landed (counted wait); read this thread's 4 x 4 block
    //   gaps 4-11   the input transform -> the other V buffer
    //   gap 13      request the patch of the chunk after the next
    auto stage = [&](auto PAR, int kc) {
        constexpr int buf = decltype(PAR)::value;
        const int kd = kc + 1 < g.nchunk ? kc + 1 : g.nchunk - 1;
        const float4* Ua = reinterpret_cast<const float4*>(Us + buf * UBLK_S) + fr * 256 + lane;
        const float4* Vb = reinterpret_cast<const float4*>(Vs + buf * VBLK_S) + fr * 256 + lane;
        float4 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { a[j] = Ua[j * 64]; b[j] = Vb[j * 64]; }
        __builtin_amdgcn_sched_barrier(0);
        const float av[4][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}, {a[2].x, a[2].y, a[2].z, a[2].w}, {a[3].x, a[3].y, a[3].z, a[3].w}};
        const float bv[4][4] = {{b[0].x, b[0].y, b[0].z, b[0].w}, {b[1].x, b[1].y, b[1].z, b[1].w}, {b[2].x, b[2].y, b[2].z, b[2].w}, {b[3].x, b[3].y, b[3].z, b[3].w}};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int j = i & 3, ks = i >> 2;           // frequencies alternate: two MFMAs on one accumulator are four apart
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][ks], bv[j][ks], acc[j], 0, 0, 0);
            if (i == 0) dma_U(kd, buf ^ 1);
            else if (i == 3) { CC_WAIT_VMCNT_FENCE(4); read_raw(); }
            else if (i >= 4 && i < 12) {
                const int s8 = i - 4;
                if (s8 < 4) col_step(s8);
                else row_step(s8 - 4, buf ^ 1);
            }
            else if (i == 13) { dma_raw(kc + 2, 0); dma_raw(kc + 2, 1); dma_raw(kc + 2, 2); }
            if (i == 0 || (i >= 3 && i < 14)) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        CC_WAIT_VMCNT_FENCE(3);                  // this wave's part of the next U block has landed (the three patch requests stay in flight)
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };

    if (c_beg < c_end) {
        CC_WAIT_VMCNT0_FENCE();
        read_raw();
#pragma unroll
        for (int b = 0; b < 4; b++) col_step(b);
#pragma unroll
        for (int i = 0; i < 4; i++) row_step(i, odd);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 3; i++) dma_raw(c_beg + 1, i);
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
    CC_WAIT_VMCNT0_FENCE();                      // the last stage's look-ahead patch requests: landed before the patch area can be reused

    // ---- output transform Y = A^T M A.  This lane holds tile t = l31 and rows m = (r & 3) + 8 (r >> 2) + 4 lk (r = 0 .. 15) of frequency
    // row fr.  Columns first, inside the wave: P[fr][0] = M0 + M1 + M2, P[fr][1] = M1 - M2 - M3; every wave publishes its two column
    // values of all 16 rows in LDS (lane-linear: conflict-free), then wave (fr, kh) finishes rows r0 .. r0 + RPW - 1: Y[0][b] = P0 + P1 + P2,
    // Y[1][b] = P1 - P2 - P3 (summed over the reduction halves first).  No register is indexed by a wave id that way.
    constexpr int RPW = 4 / KS;                  // rows of the 16 finished per wave
    const int rbase = 4 * fr + RPW * kh;         // r = rbase + rr:  r >> 2 = fr,  r & 3 = RPW kh + rr
    const int q = qb * SBT + l31;
    const bool qv = q < g.Q;
    const int qq = qv ? q : 0;
    const int n = qq / g.TPI;
    const int qr = qq - n * g.TPI;
    const int ty = qr / g.TX, tx = qr - ty * g.TX;
    const int oy = 2 * ty, ox = 2 * tx;
    const int m_base = mb * SBM + 4 * lk + 8 * fr + RPW * kh;        // + rr
    const bool hr = P.res != nullptr, ha = P.add != nullptr;
    const bool row1 = oy + 1 < g.H;
    const long o0 = (long)oy * g.W + ox;
    const int odd_lane = lane & 1;
    const long o4 = (long)(oy + odd_lane) * g.W + (ox - 2 * odd_lane);
    const bool row_ok = odd_lane ? row1 : true;
    float4 res_r[RPW], add_r[RPW];
    if constexpr (!SPLIT) {
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) res_r[rr] = add_r[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.vec2 && qv && row_ok) {
            if (hr) {
                const float* rb = P.res + (long)n * g.res_bs + o4;
#pragma unroll
                for (int rr = 0; rr < RPW; rr++) {
                    const int m = m_base + rr;
                    res_r[rr] = *reinterpret_cast<const float4*>(rb + (long)(m < g.M ? m : g.M - 1) * g.HW);
                }
            }
            if (ha) {
                const float* ab = P.add + (long)n * g.add_bs + o4;
#pragma unroll
                for (int rr = 0; rr < RPW; rr++) {
                    const int m = m_base + rr;
                    add_r[rr] = *reinterpret_cast<const float4*>(ab + (long)(m < g.M ? m : g.M - 1) * g.HW);
                }
            }
        }
    }
    float* Xs = smem;                            // [half KS][frequency row 4][column value 2][row 16][lane 64]: 32 KB per half, over the U / V buffers
    __syncthreads();                             // (KS = 2: the other half may still be reading its last fragments out of the area)
    {
        float* xo = Xs + ((kh * 4 + fr) * 32) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            xo[r * 64] = (acc[0][r] + acc[1][r]) + acc[2][r];
            xo[(16 + r) * 64] = (acc[1][r] - acc[2][r]) - acc[3][r];
        }
    }
    __syncthreads();
    auto out_tile = [&](int rr, float& y00, float& y01, float& y10, float& y11) {
        float pv[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                float v = Xs[((i * 2 + b) * 16 + rbase + rr) * 64 + lane];
                if constexpr (KS == 2) v += Xs[(((4 + i) * 2 + b) * 16 + rbase + rr) * 64 + lane];
                pv[i][b] = v;
            }
        y00 = (pv[0][0] + pv[1][0]) + pv[2][0];
        y01 = (pv[0][1] + pv[1][1]) + pv[2][1];
        y10 = (pv[1][0] - pv[2][0]) - pv[3][0];
        y11 = (pv[1][1] - pv[2][1]) - pv[3][1];
    };
    auto pair4 = [&](float y00, float y01, float y10, float y11) -> float4 {
        const float sx = odd_lane ? y00 : y10, sy = odd_lane ? y01 : y11;
        const float rx = __shfl_xor(sx, 1), ry = __shfl_xor(sy, 1);
        return odd_lane ? make_float4(rx, ry, y10, y11) : make_float4(y00, y01, rx, ry);
    };
    if constexpr (SPLIT) {
        float* pb0 = P.part + (long)blockIdx.z * g.part_stride + (((long)n * g.M) * g.Hp + oy + odd_lane) * g.Wp + (ox - 2 * odd_lane);
        const long mstride = (long)g.Hp * g.Wp;
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) {
            const int m = m_base + rr;
            float y00, y01, y10, y11;
            out_tile(rr, y00, y01, y10, y11);
            const float4 v = pair4(y00, y01, y10, y11);
            if (qv && m < g.M) *reinterpret_cast<float4*>(pb0 + (long)m * mstride) = v;
        }
        return;
    }
    const float slope = g.act == ACT_RELU ? 0.f : (g.act == ACT_LRELU ? (g.act_b != 0.f ? g.act_b : 0.2f) : 1.f);
    auto tail = [&](float v, float r, float ad) -> float {
        if constexpr (EPI == EPI_LIN) {
            const float t = v + r;
            return t > 0.f ? t : slope * t + 0.f;
        } else if constexpr (EPI == EPI_GRAD) {
            const float t = v + ad;
            return r > 0.f ? t : slope * t;
        } else {
            return conv_tail(v, hr, r, g.res_mul, g.act, g.act_a, g.act_b, ad);
        }
    };
    if (g.vec2) {
        float* yb = P.y + (long)n * g.y_bs + o4;
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) {
            const int m = m_base + rr;
            float y00, y01, y10, y11;
            out_tile(rr, y00, y01, y10, y11);
            const float bv = __shfl(bias_w, 4 * lk + 8 * fr + RPW * kh + rr);      // row m of this lane: lane m - mb*32 of the wave's load
            float4 v = pair4(y00, y01, y10, y11);
            v.x = tail(v.x + bv, res_r[rr].x, add_r[rr].x);
            v.y = tail(v.y + bv, res_r[rr].y, add_r[rr].y);
            v.z = tail(v.z + bv, res_r[rr].z, add_r[rr].z);
            v.w = tail(v.w + bv, res_r[rr].w, add_r[rr].w);
            if (qv && row_ok && m < g.M) *reinterpret_cast<float4*>(yb + (long)m * g.HW) = v;
        }
        return;
    }
    if (!qv) return;
    const bool col1 = ox + 1 < g.W;
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        float y00, y01, y10, y11;
        out_tile(rr, y00, y01, y10, y11);
        const int m = m_base + rr;
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
    if (M < cctools::env_int("CC_WINO_MINM", 32) || Cin < cctools::env_int("CC_WINO_MINC", 24) || Q < cctools::env_int("CC_WINO_MINQ", 48)) return p;
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
    // Which instance, and how many slices of the reduction.  Cost model in units of one 8-channel stage of the 64 x 64 kernel
    // (~5 800 cycles, profiles/r04_wino_probe_b.txt); per candidate: rounds x (stages per workgroup x stage cost + fixed), the
    // partial slabs and the epilogue launch of a split charged as a few stages.
    //   64 x 64: one workgroup per CU (152 KB of LDS), stage 1.0, fixed 1.5
    //   32 x 32, four waves: two per CU; a stage costs `sst` when the CU is shared and `sal` when the workgroup has it alone
    //   32 x 32, eight waves (the reduction halved inside the workgroup): one per CU, stages / 2 at `sst` each, a longer exchange
    const long mu = mult > 1 ? mult : 1;
    const long blocks = (long)p.nqb * p.nmb * mu;
    p.nsplit = 1;
    p.cps = p.nchunk;
    p.tile = 0;
    const int small_mode = cctools::env_int("CC_WINO_SMALL", 1);        // tools: 0 = never, 2 = wherever the geometry allows
    const int nqb_s = (int)((Q + SBT - 1) / SBT), nmb_s = (M + SBM - 1) / SBM;
    const long blocks_s = (long)nqb_s * nmb_s * mu;
    double best = 1e30;
    auto split_cost = [](int real) { return real > 1 ? 1.0 + 0.25 * real : 0.0; };
    if (small_mode != 2 || blocks_s > (1l << 20)) {
        // 64 x 64 candidates (the model of round 4)
        best = (double)((blocks + 255) / 256) * (p.nchunk + 1.5);
        if (blocks < cctools::env_int("CC_WINO_SPLIT_BELOW", 224) && p.nchunk >= 4) {
            const int cap = p.nchunk / 2 < 16 ? p.nchunk / 2 : 16;
            for (int ns = 2; ns <= cap; ns++) {
                const int cps = (p.nchunk + ns - 1) / ns;
                const int real = (p.nchunk + cps - 1) / cps;
                const double rounds = (double)((blocks * real + 255) / 256);
                const double t = rounds * (cps + 1.5) + split_cost(real);
                if (t < best - 1e-9) { best = t; p.nsplit = real; p.cps = cps; }
            }
        }
    }
    if (small_mode && blocks < cctools::env_int("CC_WINO_SMALL_BELOW", 3000) && blocks_s <= (1l << 20)) {
        const double sst = 0.01 * cctools::env_int("CC_WINO_S_STAGE", 25);      // shared CU: two workgroups advance one stage each per ~3 600 cycles
        const double sal = 0.01 * cctools::env_int("CC_WINO_S_ALONE", 42);
        const double sfx = 0.01 * cctools::env_int("CC_WINO_S_FIXED", 100);
        const int only_tile = cctools::env_int("CC_WINO_S_TILE", 0);           // tools / tests: 1 = four-wave form only, 2 = eight-wave form wherever it can run
        int t_tile = 0, t_ns = 1, t_cps = p.nchunk;
        double t_best = 1e30;
        const int cap = p.nchunk >= 4 ? (p.nchunk / 2 < 16 ? p.nchunk / 2 : 16) : 1;
        for (int ns = 1; ns <= cap; ns++) {
            const int cps = (p.nchunk + ns - 1) / ns;
            const int real = (p.nchunk + cps - 1) / cps;
            const long wgs = blocks_s * real;
            // four waves: 512 slots; a launch that leaves every CU one workgroup runs the stages alone
            const double t4 = wgs <= 256 ? cps * sal + sfx : (double)((wgs + 511) / 512) * (cps * 2.0 * sst + sfx);
            if (t4 + split_cost(real) < t_best - 1e-9) { t_best = t4 + split_cost(real); t_tile = 1; t_ns = real; t_cps = cps; }
            if (real == 1 && (p.nchunk & 1) == 0 && p.nchunk >= 4 && only_tile != 1) {
                const double t8 = only_tile == 2 ? -1.0 : (double)((wgs + 255) / 256) * ((cps / 2) * 2.0 * sst + sfx + 0.3);
                if (t8 < t_best - 1e-9) { t_best = t8; t_tile = 2; t_ns = 1; t_cps = cps; }
            }
        }
        if (t_best < best || small_mode == 2) { best = t_best; p.tile = t_tile; p.nsplit = t_ns; p.cps = t_cps; }
    }
    if (p.tile) { p.nqb = nqb_s; p.nmb = nmb_s; }
    // layers with fewer than 48 output channels are admitted for the 32-row blocks only: on a 64 x 64 block half of the rows would be
    // padding (the eligibility floor dropped from 33 to 32 channels when the 32 x 32 instances arrived, round 5)
    if (!p.tile && M < 48 && cctools::env_int("CC_WINO_MINM", 32) >= 32) return WinoPlan{};
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
    // 16-byte epilogue accesses (lane pairs: 4 pixels of one row): rows and batch strides of y / res / add 16-byte aligned
    bool v2 = (gg.W % 4 == 0) && (gg.y_bs % 4 == 0) && (gg.res_bs % 4 == 0) && (gg.add_bs % 4 == 0);
    for (int k = 0; k < nprob; k++)
        v2 = v2 && ((((uintptr_t)probs[k].y) | (uintptr_t)probs[k].res | (uintptr_t)probs[k].add | (uintptr_t)probs[k].part) % 16 == 0);
    a.vec2 = v2 ? 1 : 0;
    // XCD order (the kernels deal consecutive work items to one XCD): which operand should an XCD's run of items have in common?  The
    // transformed weights of a problem are 16 M Cin floats, its input B Cin H W: on the 8x26 ... 2x7 maps of the 256-1024 channel layers
    // the weights are 2-20x the input, and with tile-block-major order every XCD streamed ALL of them from HBM (M512 C512 8x26:
    // 90 MB per launch for 17 MB of weights).
#ifndef CC_WINO_MBMAJOR
#define CC_WINO_MBMAJOR 1
#endif
    a.mb_major = (CC_WINO_MBMAJOR && 16l * gg.M > (long)gg.B * gg.H * gg.W) ? 1 : 0;
    if (cctools::env_flag("CC_WINO_TRACE"))
        fprintf(stderr, "wino: %dx[B%d M%d C%d %dx%d] tile %d nqb %d nmb %d nsplit %d cps %d act %d res_mul %d vec2 %d\n", nprob, gg.B, gg.M,
                gg.Cin, gg.H, gg.W, p.tile, p.nqb, p.nmb, p.nsplit, p.cps, gg.act, gg.res_mul, a.vec2);
    const size_t smem = p.tile ? (size_t)SHALF * p.tile * sizeof(float) : (size_t)(2 * UBLK + 2 * VBLK + 8 * RWAVE) * sizeof(float);
    dim3 grid((unsigned)(((a.total + 7) / 8) * 8), 1, (unsigned)p.nsplit);
    const bool grad = gg.res_mul != 0;          // (every problem of a launch has the multiplier or none has: conv.hip same_problem_shape)
    const int epi = (!grad && (gg.act == ACT_NONE || gg.act == ACT_RELU || gg.act == ACT_LRELU)) ? EPI_LIN
                    : (grad && probs[0].res && (gg.act == ACT_RELU || gg.act == ACT_LRELU)) ? EPI_GRAD : EPI_GEN;
    auto go = [&](auto kern, bool& attr) {
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(WTHREADS), smem, s, a);
    };
    static bool attr_set[4] = {false, false, false, false};
    if (p.tile) {        // 32 x 32 instance: four waves (p.tile 1) or eight with the reduction halved inside the workgroup (2)
        static bool attr_s[2][4] = {};
        const unsigned nth = 256u * (unsigned)p.tile;
        auto gos = [&](auto kern, bool& attr) {
            if (!attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr = true;
            }
            hipLaunchKernelGGL(kern, grid, dim3(nth), smem, s, a);
        };
        if (p.tile == 2) {
            if (p.nsplit > 1 || (p.nchunk & 1)) return false;
            if (epi == EPI_LIN) gos(k_wino_f2x3_s<2, 0, EPI_LIN>, attr_s[1][0]);
            else if (epi == EPI_GRAD) gos(k_wino_f2x3_s<2, 0, EPI_GRAD>, attr_s[1][1]);
            else gos(k_wino_f2x3_s<2, 0, EPI_GEN>, attr_s[1][2]);
        } else {
            if (p.nsplit > 1) gos(k_wino_f2x3_s<1, 1, 0>, attr_s[0][3]);
            else if (epi == EPI_LIN) gos(k_wino_f2x3_s<1, 0, EPI_LIN>, attr_s[0][0]);
            else if (epi == EPI_GRAD) gos(k_wino_f2x3_s<1, 0, EPI_GRAD>, attr_s[0][1]);
            else gos(k_wino_f2x3_s<1, 0, EPI_GEN>, attr_s[0][2]);
        }
        return true;
    }
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
            case 64: go(k_wino_f2x3<0, 0, 64>, attr_abl[10]); return true;
            case 128: go(k_wino_f2x3<0, 0, 128>, attr_abl[11]); return true;
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
