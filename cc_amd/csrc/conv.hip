// fp32 convolutions of the four CC networks on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32:
// exact fp32, bit-equal to an fmaf chain, 157 TFLOP/s peak).  Replaces cuDNN/MIOpen's conv2d,
// conv_transpose2d and convolution_backward (219 forward + 297 backward calls per step).
//
// ONE implicit-GEMM "gather-GEMM" kernel covers conv forward, conv data-gradient, transposed-conv forward
// and transposed-conv data-gradient:
//      Y[n, m, P(t)] = epilogue( sum_{c, i, j}  A[m, (c,i,j)] * X[n, c, si*ty + dy(i), si*tx + dx(j)] )
//   * output pixels t = (ty, tx) live on a lattice  P(t) = (oy0 + so*ty, ox0 + so*tx)   (so = 2 selects one
//     parity class of a stride-2 transposed conv / stride-2 data-gradient, so no MFMA work is spent on
//     structurally-zero taps);
//   * the taps form a regular Rt x St grid: dy(i) = dy0 + i*dstep, dx(j) = dx0 + j*dstep, and the weight of
//     (m, c, i, j) sits at  w[w0 + m*w_sm + c*w_sc + i*w_ri + j*w_sj]  -- strides express [K,C,R,S] weights,
//     their transpose/flip for data-gradients, and ConvTranspose2d's [Cin,Cout,R,S] layout without repacking.
// GEMM view: M = output channels, N = B*OHt*OWt lattice pixels (contiguous in NCHW -> coalesced stores, and
// MFMA D columns map to lanes = pixels), K = Cin*Rt*St gathered on the fly (im2col never materialised).
// Tile: BM x 128 pixels x 16 (K) per 256-thread workgroup, 4 waves of (BM/2 or 32) x (64 or 32) built from
// 32x32x2 MFMAs; register-staged double-buffered LDS, one barrier per K chunk.  B-tile loads are
// lane = pixel (coalesced along x) with a wave-uniform k so the (c,i,j) decode runs on the scalar unit.
// Epilogue fuses bias, residual add, ReLU / LeakyReLU(0.2) / a*sigmoid+b.
//
// Weight gradient: second kernel, M = channels of dY, N = (c,i,j), K = pixels, split over pixel ranges with a
// deterministic second-stage reduction (no atomics).
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include "cc_common.h"
#include "conv_internal.h"
#include "cc_tools.h"
#include "conv_tail.h"
#include "wino_weights.h"
#include "../../include/ccengine.h"
#include <vector>
#include <string>
#include <mutex>

// ---- per-kernel timing (measurement aid for bench.py's roofline line; off unless cc_timing_enable(1) was called on this
// process; autograd runs the backward pass on its own threads): the MAIN device kernel of every conv / weight-gradient call is bracketed with HIP events on its own stream, so the
// reported duration is the kernel's (what rocprofv3 --kernel-trace shows), not the C-ABI call's.
namespace cctiming {
#ifdef CC_TOOLS
struct Rec { std::string name; double gflop; hipEvent_t e0, e1; };
static std::vector<Rec>* recs = nullptr;
static std::mutex mtx;
struct Scope {
    hipEvent_t e1 = nullptr;
    hipStream_t s;
    Scope(const char* name, double gflop, hipStream_t st, bool active = true) : s(st) {
        if (!recs || !active) return;
        hipEvent_t e0 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        {
            std::lock_guard<std::mutex> lk(mtx);
            if (!recs) return;
            recs->push_back(Rec{name, gflop, e0, e1});
        }
        (void)hipEventRecord(e0, s);
    }
    ~Scope() { if (e1) (void)hipEventRecord(e1, s); }
};
#else
struct Scope { Scope(const char*, double, hipStream_t, bool = true) {} };      // product build: no registry, no events
#endif
}  // namespace cctiming

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// BM = 16 (layers with <= 16 output channels: the full-resolution ends of the nets, prediction heads, their data-gradients):
// v_mfma_f32_16x16x4_f32 tiles, so that no MFMA row is spent on channels that do not exist (a 32-row tile wastes half)
inline int pick_bm_fwd(int M) {
    static const int no16 = cctools::env_flag("CC_CONV_NO_BM16");
    return M > 64 ? 128 : (M > 32 ? 64 : ((M > 16 || no16) ? 32 : 16));
}

constexpr int BN = 128;   // pixels per workgroup tile
constexpr int BK = 16;    // reduction chunk

using namespace cctail;

struct GG {
    const float* x; const float* w; const float* bias; const float* res; float* y;
    int B, Cin, IH, IW; long x_bs;
    int M; long w_sm, w_sc; int w0, w_ri, w_sj;
    int Rt, St, dy0, dx0, dstep, si;
    int OHt, OWt, so, oy0, ox0, OH, OW; long y_bs, res_bs;
    int act; float act_a, act_b;
    int res_mul;
    const float* add; long add_bs;      // res_mul mode only: tensor of y's shape added before act'(res) is applied (may alias y)
};

template <int BM>
__global__ __launch_bounds__(256) void k_gather_gemm(GG g) {
    constexpr int WM = (BM >= 64) ? BM / 2 : 32;     // wave tile rows
    constexpr int WN = (BM >= 64) ? 64 : 32;         // wave tile cols
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM + 4;                        // padded A row (k-major: As[k][m])
    constexpr int AQ = BM / 16;                       // A elements per thread per chunk
    __shared__ float As[2][BK * AP];
    __shared__ float Bs[2][BK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (BM >= 64) ? (wid >> 1) : 0;
    const int wn = (BM >= 64) ? (wid & 1) : wid;
    const int m0 = blockIdx.y * BM;
    const long p0 = (long)blockIdx.x * BN;
    const int HWt = g.OHt * g.OWt;
    const long Ntot = (long)g.B * HWt;
    const int RS = g.Rt * g.St;
    const int Ktot = g.Cin * RS;
    const int x_cs = g.IH * g.IW;

    // ---- loader roles
    // B: pixel column jb, k rows kr0 + 2q
    const int jb = tid & (BN - 1);
    const int kr0 = __builtin_amdgcn_readfirstlane(tid >> 7);
    const long pb = p0 + jb;
    const bool pvalid = pb < Ntot;
    int iy0 = 0, ix0 = 0;
    long xbase = 0;
    if (pvalid) {
        const int n = (int)(pb / HWt);
        const int t = (int)(pb - (long)n * HWt);
        const int ty = t / g.OWt, tx = t - ty * g.OWt;
        iy0 = g.si * ty;
        ix0 = g.si * tx;
        xbase = (long)n * g.x_bs;
    }
    float ra[AQ], rb[8];

    auto load_chunk = [&](int kbase) {
        // A tile: element e = tid + 256q -> (m = e>>4, kk = e&15 = tid&15): one (c,i,j) decode per chunk
        {
            const int k = kbase + (tid & 15);
            const bool kv = k < Ktot;
            const int c = k / RS, rem = k - c * RS;
            const int i = rem / g.St, j = rem - i * g.St;
            const long woff = (long)g.w0 + (long)c * g.w_sc + i * g.w_ri + j * g.w_sj;
#pragma unroll
            for (int q = 0; q < AQ; q++) {
                const int m = m0 + (tid >> 4) + 16 * q;
                ra[q] = (kv && m < g.M) ? g.w[woff + (long)m * g.w_sm] : 0.f;
            }
        }
        // B tile: k wave-uniform
        int k = kbase + kr0;
        int c = k / RS, rem = k - c * RS;
        int i = rem / g.St, j = rem - i * g.St;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            float v = 0.f;
            if (pvalid && k < Ktot) {
                const int iy = iy0 + g.dy0 + i * g.dstep, ix = ix0 + g.dx0 + j * g.dstep;
                if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
                    v = g.x[xbase + (long)c * x_cs + iy * g.IW + ix];
            }
            rb[q] = v;
            k += 2;
            j += 2;
            while (j >= g.St) { j -= g.St; i++; }
            while (i >= g.Rt) { i -= g.Rt; c++; }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int q = 0; q < AQ; q++) {
            const int e = tid + 256 * q;
            As[buf][(e & 15) * AP + (e >> 4)] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) Bs[buf][(kr0 + 2 * q) * BN + jb] = rb[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int nchunks = (Ktot + BK - 1) / BK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const int l31 = lane & 31, lk = lane >> 5;
    for (int ch = 0; ch < nchunks; ch++) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ks++) {
            float af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; a++) af[a] = As[buf][(2 * ks + lk) * AP + wm * WM + a * 32 + l31];
#pragma unroll
            for (int b = 0; b < TN; b++) bf[b] = Bs[buf][(2 * ks + lk) * BN + wn * WN + b * 32 + l31];
#pragma unroll
            for (int a = 0; a < TM; a++)
#pragma unroll
                for (int b = 0; b < TN; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D col = lane&31 -> pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> channel
    const int y_cs = g.OH * g.OW;
#pragma unroll
    for (int b = 0; b < TN; b++) {
        const long p = p0 + wn * WN + b * 32 + l31;
        if (p >= Ntot) continue;
        const int n = (int)(p / HWt);
        const int t = (int)(p - (long)n * HWt);
        const int ty = t / g.OWt, tx = t - ty * g.OWt;
        const long pix = (long)(g.oy0 + g.so * ty) * g.OW + (g.ox0 + g.so * tx);
        float* yb = g.y + (long)n * g.y_bs + pix;
        const float* rbp = g.res ? g.res + (long)n * g.res_bs + pix : nullptr;
        const float* abp = g.add ? g.add + (long)n * g.add_bs + pix : nullptr;
#pragma unroll
        for (int a = 0; a < TM; a++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < g.M) {
                    float v = acc[a][b][r];
                    if (g.bias) v += g.bias[m];
                    yb[(long)m * y_cs] = conv_tail(v, rbp != nullptr, rbp ? rbp[(long)m * y_cs] : 0.f, g.res_mul, g.act, g.act_a, g.act_b,
                                                   abp ? abp[(long)m * y_cs] : 0.f);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ patch-staged implicit GEMM (main path)
// Same GEMM as k_gather_gemm, restructured so that neither operand needs per-element index math:
//   * the output tile is a 4 x 32 block of lattice pixels of ONE image; for a chunk of CK input channels the
//     input PATCH that all Rt x St taps of that tile touch is staged in LDS once (zero-filled outside the
//     image) and every tap's B fragment is a shifted ds_read of it -> global->LDS traffic / tap count;
//   * weights are repacked per call to wp[tap][c][m] (m contiguous, zero padded to CK / BM multiples), so an
//     A tile is CK contiguous rows;
//   * both are moved by LDS-DMA (global_load_lds: no staging VGPRs, no ds_write, fully asynchronous) and
//     double-buffered: A per (chunk, tap) stage, patch per chunk; one barrier per stage (= CK/2 MFMA k-steps
//     x TM x TN MFMAs per wave);
//   * deep layers on small maps get split-K over channel chunks (grid.z) with a deterministic second pass.
// Two tile shapes of 128 lattice pixels: 4 rows x 32 columns, or 8 x 16 (CP::tw16) -- whichever pads the map less
// (208 columns = 6.5 x 32 but 13 x 16; 104 = 3.25 x 32 but 6.5 x 16).  The 32 MFMA columns of a wave are one row of 32 pixels
// or two rows of 16: only the per-lane patch offset differs.
constexpr int TH = 4, TW = 32;

struct CP {
    const float* x; const float* wp; const float* zeros; const float* bias; const float* res; float* y; float* part;
    int B, Cin, IH, IW; long x_bs;
    int M, Mpad, Cpad;
    int Rt, St, si, dstep, dy_base, dx_base, ymin, xmin;
    int PH, PWr, PS;
    int tw16;              // tile = 8 rows x 16 columns instead of 4 x 32
    int ipt, phi;          // ipt > 1: maps of <= 4 lattice rows -- the 8 tile rows stack the rows of ipt consecutive images, each with
                           // its own phi patch rows (halo included) in the LDS patch
    int aligned, shift;    // aligned: patch rows start on a 16-byte boundary of x (PWr = padded row length) -> dwordx4 LDS-DMA
    int OHt, OWt, so, oy0, ox0, OH, OW; long y_bs, res_bs;
    int tiles_x, tiles_y;
    int nsplit, cps; long part_stride;
    int act; float act_a, act_b;
    int res_mul;
    const float* add; long add_bs;
    int vec4;              // fused epilogue may use 16-byte accesses: unit lattice stride, rows 16-byte aligned in y / res / add
    int wmajor;            // XCD order of the launch (conv_xcd_blocks): 0 = an XCD's run of work shares INPUT tiles, 1 = it shares WEIGHT
                           // blocks (layers whose weights are the larger operand: the 8x26 ... 2x7 maps of the 256-1024 channel layers)
};

// wp[(t*Cpad + c)*Mpad + m] = w[w0 + m*w_sm + c*w_sc + i*w_ri + j*w_sj]  (0 beyond Cin / M), t = i*St + j
__global__ __launch_bounds__(256) void k_repack_w(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ zeros,
                                                  int M, int Cin, int Mpad, int Cpad, int T, int St, long w_sm, long w_sc,
                                                  int w0, int w_ri, int w_sj) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) zeros[threadIdx.x] = 0.f;
    const long tot = (long)T * Cpad * Mpad;
    if (e >= tot) return;
    const int m = (int)(e % Mpad);
    const long r = e / Mpad;
    const int c = (int)(r % Cpad), t = (int)(r / Cpad);
    const int i = t / St, j = t - i * St;
    wp[e] = (m < M && c < Cin) ? w[(long)w0 + (long)m * w_sm + (long)c * w_sc + i * w_ri + j * w_sj] : 0.f;
}

// All weight repacks of a training step in ONE launch: desc[d] = 16 longs
//   {src, dst, M, Cin, Mpad, Cpad, T, St, w_sm, w_sc, w0, w_ri, w_sj, nfloats, first_block, nblocks}
// (the per-call k_repack_w launches were ~540 tiny kernels = 3 ms of launch latency per step).
// It is a transpose ([m][c][tap] or [c][m][tap] in memory -> [tap][c][m]) of ~600 MB per step, so it goes through an
// LDS tile: one workgroup = 64 m x CT c x all T taps, read in SOURCE memory order (tap fastest, then whichever of c / m
// has the smaller stride), written in destination order (m fastest): both sides coalesced.
__host__ __device__ inline int repack_ct(int T) { const int ct = 72 / T; return ct < 1 ? 1 : (ct > 64 ? 64 : ct); }
inline long repack_blocks(int Mpad, int Cpad, int T) {
    const int ct = repack_ct(T);
    return (long)((Mpad + 63) / 64) * ((Cpad + ct - 1) / ct);
}

// TC > 0: the tap count as a compile-time constant (the index arithmetic of the two loops is three integer divisions per
// element: with run-time divisors they -- not memory -- bound the kernel, 0.6 ms per step for 890 MB)
template <int TC>
__device__ __forceinline__ void repack_body(const long* __restrict__ d, float* tile, int bid) {
    const float* __restrict__ w = reinterpret_cast<const float*>(d[0]);
    float* __restrict__ wp = reinterpret_cast<float*>(d[1]);
    const int M = (int)d[2], Cin = (int)d[3], Mpad = (int)d[4], Cpad = (int)d[5], St = (int)d[7];
    const int T = TC > 0 ? TC : (int)d[6];
    const long w_sm = d[8], w_sc = d[9], w0 = d[10], w_ri = d[11], w_sj = d[12];
    const int CT = repack_ct(T);
    const int nmt = (Mpad + 63) / 64;
    const int m0 = (bid % nmt) * 64, c0 = (bid / nmt) * CT;
    const int CTT = CT * T, n = 64 * CTT;
    const bool m_slow = w_sm > w_sc;
    // eight loads in flight per work item (one load, wait, LDS store per iteration leaves the kernel at 1.6 TB/s: latency-bound)
    for (int e0 = threadIdx.x; e0 < n; e0 += 8 * 256) {
        float v[8];
        int li[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * 256;
            v[u] = 0.f;
            li[u] = -1;
            if (e < n) {
                int m_, c_, t;
                if (m_slow) { m_ = e / CTT; const int r = e - m_ * CTT; c_ = r / T; t = r - c_ * T; }
                else { c_ = e / (64 * T); const int r = e - c_ * 64 * T; m_ = r / T; t = r - m_ * T; }
                const int m = m0 + m_, c = c0 + c_;
                li[u] = (t * CT + c_) * 65 + m_;
                if (m < M && c < Cin) {
                    const int i = t / St, j = t - i * St;
                    v[u] = w[w0 + (long)m * w_sm + (long)c * w_sc + i * w_ri + j * w_sj];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (li[u] >= 0) tile[li[u]] = v[u];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
        const int m_ = e & 63, r = e >> 6;
        const int t = r / CT, c_ = r - t * CT;
        if (m0 + m_ < Mpad && c0 + c_ < Cpad) wp[((long)t * Cpad + c0 + c_) * Mpad + m0 + m_] = tile[r * 65 + m_];
    }
}

__global__ __launch_bounds__(256) void k_repack_table(const long* __restrict__ desc, int ndesc) {
    __shared__ float tile[72 * 65 + 64 * 65];
    // binary search the descriptor whose block range contains blockIdx.x
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[16 * mid + 14] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long* d = desc + 16 * lo;
    const int bid = (int)((long)blockIdx.x - d[14]);
    if (bid >= (int)d[15]) return;
    if ((int)d[6] == ccwino::WINO_T) {         // Winograd layers: U = G g G^T in the staging layout of wino.hip (d[7] < 0: flipped taps)
        ccwino::wino_weight_body(reinterpret_cast<const float*>(d[0]), reinterpret_cast<float*>(d[1]), (int)d[2], (int)d[3], (int)d[5],
                                 d[8], d[9], d[10], d[11], d[12], d[7] < 0 ? 1 : 0, bid);
        return;
    }
    switch ((int)d[6]) {                       // tap counts of the CC networks (3x3, 7x7, 5x5, 4x4 and their parity classes)
        case 9: repack_body<9>(d, tile, bid); break;
        case 1: repack_body<1>(d, tile, bid); break;
        case 2: repack_body<2>(d, tile, bid); break;
        case 4: repack_body<4>(d, tile, bid); break;
        case 49: repack_body<49>(d, tile, bid); break;
        case 25: repack_body<25>(d, tile, bid); break;
        case 16: repack_body<16>(d, tile, bid); break;
        case 6: repack_body<6>(d, tile, bid); break;
        case 12: repack_body<12>(d, tile, bid); break;
        default: repack_body<0>(d, tile, bid); break;
    }
}

// TPS = taps per pipeline stage: one barrier (+ DMA wait) per TPS*CK/2*TM*TN MFMAs per wave.  With TPS = 1 a stage is only
// 32 MFMAs (2 k cycles) and the LDS-read latency + barrier skew at every stage boundary costs ~10-15 %; TPS = 3 (a whole
// tap row of a 3x3) amortises it 3x for 32 KB more LDS (still two workgroups per CU).
// SPLIT: 0 = whole reduction in this workgroup (fused epilogue), 1 = split-K partial slabs, 2 = decided per class at run time
#if defined(CC_ABLATE_DMA) || defined(CC_ABLATE_LDS) || defined(CC_ABLATE_BARRIER) || defined(CC_ABLATE_MFMA)
// ablation builds compute garbage: keep it finite and tiny so that the rest of the step runs at its normal speed
__device__ __forceinline__ float abl_fix(float v) { return (v == v && fabsf(v) < 1e30f) ? 1e-6f * fminf(fmaxf(v, -1.f), 1.f) : 0.f; }
#else
__device__ __forceinline__ float abl_fix(float v) { return v; }
#endif

// STK: the tile stacks the rows of CP::ipt images (maps of <= 4 lattice rows); a separate instantiation -- the row mapping costs
// scalar registers the common kernels do not have (they sit at the SGPR limit: +10 spills and -6 % measured with it compiled in)
template <int BM, int CK, int TPS, int SPLIT, int STK>
__device__ __forceinline__ void conv_patch_body(const CP& g, const int bx_in, const int by_in, const int bz_in) {
    constexpr int WM = (BM >= 64) ? BM / 2 : 32;
    constexpr int TM = WM / 32;
    constexpr int TN = (BM >= 64) ? 2 : 1;           // lattice rows of the tile per wave
    HIP_DYNAMIC_SHARED(float, smem)
    float* As = smem;                          // [2][TPS][CK][BM]
    float* Ps = smem + 2 * TPS * CK * BM;      // [2][CK][PS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (BM >= 64) ? (wid >> 1) : 0;
    const int row0 = (BM >= 64) ? 2 * (wid & 1) : wid;       // first lattice row (0..3) of this wave
    const int l31 = lane & 31, lk = lane >> 5;

    int bx = bx_in;
    const int tile_x = bx % g.tiles_x;
    bx /= g.tiles_x;
    const int tile_y = bx % g.tiles_y;
    const int ipt = STK ? g.ipt : 1;
    const int n = (bx / g.tiles_y) * ipt;                       // (first) image of the tile
    const int n_tile = n;
    const int rowstep = g.tw16 ? 2 : 1;                         // lattice rows per MFMA column group
    const int lr = g.tw16 ? (l31 >> 4) : 0, lc = g.tw16 ? (l31 & 15) : l31;     // this lane's row / column inside the group
    const int ty0 = tile_y * (g.tw16 ? 8 : TH), tx0 = tile_x * (g.tw16 ? 16 : TW);
    const int gy0 = g.si * ty0 + g.ymin, gx0 = g.si * tx0 + g.xmin;
    const int m0 = by_in * BM;
    const int T = g.Rt * g.St;
    const int nchunk = g.Cpad / CK;
    const int c_beg = bz_in * g.cps;
    int c_end = c_beg + g.cps;
    if (c_end > nchunk) c_end = nchunk;
    const int x_cs = g.IH * g.IW;
    const int R = g.PS >> 6;
    const float* xn = g.x + (long)n * g.x_bs;
    // tile row -> (image of the tile, lattice row); rows past the stacked images are dead (their results are not stored)
    auto rowmap = [&](int tyv, int& j, int& ty) -> bool {
        if (!STK || ipt <= 1) { j = 0; ty = tyv; return true; }
        j = tyv / g.OHt;
        ty = tyv - j * g.OHt;
        return j < ipt && n + j < g.B;
    };
    // patch offset of a tile row's first pixel (dead rows read row 0: any finite address inside the patch)
    auto prow = [&](int tr) -> int {
        if (!STK || ipt <= 1) return g.si * tr * g.PWr;
        int j, ty;
        return rowmap(tr, j, ty) ? (j * g.phi + g.si * ty) * g.PWr : 0;
    };

    auto load_patch = [&](int chunk, int buf) {
        float* dst = Ps + buf * CK * g.PS;
        if (g.aligned) {
            // 16-byte LDS-DMA: lane = one 4-float chunk of a patch row (rows are 16-byte aligned in x, see plan_conv);
            // a chunk is entirely inside or outside the image because IW % 4 == 0
            const int cpr = g.PWr >> 2;                       // chunks per patch row
            const int nq = g.PS >> 2;                         // chunks per channel (multiple of 16)
            for (int q0 = 0; q0 < nq; q0 += 64) {
                const int q = q0 + lane;
                const int pyt = q / cpr, qx = q - pyt * cpr;
                const int pj = (STK && ipt > 1) ? pyt / g.phi : 0, py = pyt - pj * g.phi;       // stacked images: patch rows [image][phi]
                const int iy = gy0 + py, ix = gx0 - g.shift + 4 * qx;
                const bool ok = (q < nq) && (pyt < g.PH) && (!STK || n + pj < g.B) && ((unsigned)iy < (unsigned)g.IH) && ((unsigned)ix < (unsigned)g.IW);
                const long go = (STK ? (long)pj * g.x_bs : 0l) + (long)iy * g.IW + ix;
#pragma unroll
                for (int k = 0; k < CK / 4; k++) {
                    const int cl = wid + 4 * k;
                    const int c = chunk * CK + cl;
                    const float* src = (ok && c < g.Cin) ? xn + (long)c * x_cs + go : g.zeros;
                    __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(src), CC_LDS_PTR(dst + cl * g.PS + 4 * q0), 16, 0, 0);
                }
            }
            return;
        }
        for (int r = 0; r < R; r++) {
            const int pos = lane + 64 * r;
            const int pyt = pos / g.PWr, px = pos - pyt * g.PWr;
            const int pj = (STK && ipt > 1) ? pyt / g.phi : 0, py = pyt - pj * g.phi;
            const int iy = gy0 + py, ix = gx0 + px;
            const bool ok = (pyt < g.PH) && (!STK || n + pj < g.B) && ((unsigned)iy < (unsigned)g.IH) && ((unsigned)ix < (unsigned)g.IW);
            const long go = (STK ? (long)pj * g.x_bs : 0l) + (long)iy * g.IW + ix;
#pragma unroll
            for (int k = 0; k < CK / 4; k++) {
                const int cl = wid + 4 * k;
                const int c = chunk * CK + cl;
                const float* src = (ok && c < g.Cin) ? xn + (long)c * x_cs + go : g.zeros + lane;
                __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(src), CC_LDS_PTR(dst + cl * g.PS + 64 * r), 4, 0, 0);
            }
        }
    };
    auto load_A = [&](int chunk, int tap0, int buf) {
        // per tap: CK rows of BM contiguous floats: wp[((tap*Cpad + chunk*CK + kk) * Mpad) + m0 + mm]
        constexpr int N4 = CK * BM / 4;                        // float4 count of one tap
#pragma unroll
        for (int tt = 0; tt < TPS; tt++) {
            if (tap0 + tt < T) {
                const float* base = g.wp + ((long)(tap0 + tt) * g.Cpad + (long)chunk * CK) * g.Mpad + m0;
                float* dst = As + (buf * TPS + tt) * CK * BM;
#pragma unroll
                for (int q = 0; q < (N4 + 255) / 256; q++) {
                    const int w4 = q * 256 + wid * 64;         // first float4 of this wave (uniform)
                    if (w4 < N4 && (N4 % 64 == 0 || w4 + lane < N4)) {        // BM = 16, CK = 8: half a wave (EXEC-masked DMA)
                        const int f = 4 * (w4 + lane);
                        const int kk = f / BM, mm = f - kk * BM;
                        __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(base + (long)kk * g.Mpad + mm), CC_LDS_PTR(dst + 4 * w4), 16, 0, 0);
                    }
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // BM == 16 only

    // patch offsets of this lane's pixels (one per lattice-row group of the wave)
    int poff[2] = {0, 0};
    if constexpr (STK && BM == 16) {
        const int l15 = lane & 15;
        poff[0] = prow(rowstep * row0) + g.si * l15;
        poff[1] = g.tw16 ? prow(rowstep * row0 + 1) + g.si * l15 : poff[0] + g.si * 16;
    } else if constexpr (STK != 0) {
#pragma unroll
        for (int b = 0; b < 2; b++) poff[b] = prow(rowstep * (row0 + (b < TN ? b : 0)) + lr) + g.si * lc;
    }
    if (c_beg < c_end) {
        load_patch(c_beg, c_beg & 1);
        load_A(c_beg, 0, 0);
        CC_WAIT_VMCNT0();
        __syncthreads();
        int s = 0;
        for (int chunk = c_beg; chunk < c_end; chunk++) {
            const float* Pb = Ps + (chunk & 1) * CK * g.PS;
            int ti = 0, tj = 0;
            for (int tap0 = 0; tap0 < T; tap0 += TPS, s++) {
                // prefetch the next stage's operands (other buffers; their last readers passed the previous barrier)
#ifndef CC_ABLATE_DMA          // ablation builds (tools/ablate_conv.sh): timing only, results are garbage
                if (tap0 + TPS < T) load_A(chunk, tap0 + TPS, (s + 1) & 1);
                else if (chunk + 1 < c_end) load_A(chunk + 1, 0, (s + 1) & 1);
                if (tap0 == 0 && chunk + 1 < c_end) load_patch(chunk + 1, (chunk + 1) & 1);
#endif
#pragma unroll
                for (int tt = 0; tt < TPS; tt++) {
                    if (tap0 + tt < T) {
                        const float* Ab = As + ((s & 1) * TPS + tt) * CK * BM;
                        const int tapoff = (g.dy_base + ti * g.dstep) * g.PWr + (g.dx_base + tj * g.dstep) + g.shift;
                        if constexpr (BM == 16) {
                            // 16x16x4: lane (i = lane & 15, k = lane >> 4) feeds A[m = i][k] and B[k][pixel = i]; two 16-pixel
                            // halves of the wave's lattice row -> two independent accumulators (40-cycle dependent latency)
                            const int l15 = lane & 15, l4 = lane >> 4;
                            const float* Pl = Pb + l4 * g.PS + (STK ? 0 : (g.si * rowstep * row0) * g.PWr + g.si * l15) + tapoff;
                            const float* Al = Ab + l4 * BM + l15;
                            const int half = g.tw16 ? g.si * g.PWr : g.si * 16;      // second 16-pixel half: next row / next 16 columns
                            float af[CK / 4], bf[CK / 4][2];
#pragma unroll
                            for (int ks = 0; ks < CK / 4; ks++) {
                                af[ks] = Al[(4 * ks) * BM];
                                bf[ks][0] = Pl[(4 * ks) * g.PS + (STK ? poff[0] : 0)];
                                bf[ks][1] = Pl[(4 * ks) * g.PS + (STK ? poff[1] : half)];
                            }
#pragma unroll
                            for (int ks = 0; ks < CK / 4; ks++) {
                                acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[ks][0], acc16[0], 0, 0, 0);
                                acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[ks][1], acc16[1], 0, 0, 0);
                            }
                            if (++tj == g.St) { tj = 0; ti++; }
                            continue;
                        }
                        const float* Pl = Pb + lk * g.PS + (STK ? 0 : (g.si * (rowstep * row0 + lr)) * g.PWr + g.si * lc) + tapoff;
                        const float* Al = Ab + lk * BM + wm * WM + l31;
                        // all fragments of a tap are fetched up front (2*(TM+TN)*CK/2 VGPRs): one exposed LDS latency per
                        // tap instead of one per k-step; the MFMAs then issue back to back behind counted lgkmcnt waits
                        float af[CK / 2][TM], bf[CK / 2][TN];
#pragma unroll
                        for (int ks = 0; ks < CK / 2; ks++) {
#ifdef CC_ABLATE_LDS
#pragma unroll
                            for (int a = 0; a < TM; a++) af[ks][a] = 1e-3f * (float)((lane & 7) + a);
#pragma unroll
                            for (int b = 0; b < TN; b++) bf[ks][b] = 1e-3f * (float)((lane & 3) + b);
#else
#pragma unroll
                            for (int a = 0; a < TM; a++) af[ks][a] = Al[(2 * ks) * BM + a * 32];
#pragma unroll
                            for (int b = 0; b < TN; b++) bf[ks][b] = Pl[(2 * ks) * g.PS + (STK ? poff[b] : (g.si * rowstep * b) * g.PWr)];
#endif
                        }
#pragma unroll
                        for (int ks = 0; ks < CK / 2; ks++) {
#pragma unroll
                            for (int a = 0; a < TM; a++)
#pragma unroll
                                for (int b = 0; b < TN; b++)
#ifdef CC_ABLATE_MFMA      // one VALU multiply-add instead of the 64-cycle matrix instruction: what the step costs WITHOUT the matrix work
                                    acc[a][b][0] = fmaf(af[ks][a], bf[ks][b], acc[a][b][0]);
#else
                                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
#endif
                        }
                        if (++tj == g.St) { tj = 0; ti++; }
                    }
                }
                CC_WAIT_VMCNT0();
#ifndef CC_ABLATE_BARRIER
                __syncthreads();
#endif
            }
        }
    }

    const int y_cs = g.OH * g.OW;
    if constexpr (BM == 16) {
        // D col = lane & 15 -> pixel of the half, row = 4 * (lane >> 4) + r -> channel
        const int HWt = g.OHt * g.OWt;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int ej, ty;
            if (!rowmap(ty0 + rowstep * row0 + (g.tw16 ? h : 0), ej, ty)) continue;
            const int tx = tx0 + (g.tw16 ? 0 : 16 * h) + (lane & 15);
            if (ty >= g.OHt || tx >= g.OWt) continue;
            const int n = STK ? n_tile + ej : n_tile;                          // (shadows the tile's first image)
            const long pix = (long)(g.oy0 + g.so * ty) * g.OW + (g.ox0 + g.so * tx);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 4 * (lane >> 4) + r;
                if (m >= g.M) continue;
                if (SPLIT == 1 || (SPLIT == 2 && g.nsplit > 1)) {
                    const int Wp16 = g.tiles_x * (g.tw16 ? 16 : TW), Hp16 = g.tiles_y * (g.tw16 ? 8 : TH);      // padded slab rows
                    g.part[(long)bz_in * g.part_stride + (((long)n * g.M + m) * Hp16 + ty) * Wp16 + tx] = abl_fix(acc16[h][r]);
                } else {
                    float v = abl_fix(acc16[h][r]);
                    if (g.bias) v += g.bias[m];
                    const long o = (long)n * g.y_bs + pix + (long)m * y_cs;
                    const bool hr = g.res != nullptr;
                    g.y[o] = conv_tail(v, hr, hr ? g.res[(long)n * g.res_bs + pix + (long)m * y_cs] : 0.f, g.res_mul, g.act, g.act_a, g.act_b,
                                       g.add ? g.add[(long)n * g.add_bs + pix + (long)m * y_cs] : 0.f);
                }
            }
        }
        return;
    }
    // ---- epilogue (round 3): 16-byte stores.  The MFMA result has one PIXEL per lane and 16 channel rows per register set: storing
    // it as it lies costs 16 scalar stores per 32x32 tile (64 per wave).  Each tile goes through a wave-private 4 KB LDS block
    // (free after the main loop's last barrier) and comes back as float4 = 4 consecutive pixels of one channel row: 4 dwordx4
    // stores per tile; residual / add / mul operands are read as float4 the same way (same-box A/B: -0.22 ms/step).
    //   write: lane (pixel c = lane & 31, half lk) -> T[row][c], row = (r & 3) + 8 * (r >> 2) + 4 * lk   (32 consecutive floats per
    //          32-lane group: conflict-free);  read k = 0..3: lane -> row 8k + (lane >> 3), columns 4 * (lane & 7) .. +3.
    // Partial slabs (split-K) are padded to whole tiles, [split][n][m][tiles_y * th][tiles_x * tw]: every 16-byte store is aligned and
    // in bounds without a guard.
    float* Tt = smem + wid * 1024;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const int lr4 = g.tw16 ? (c4 >> 4) : 0, lc4 = g.tw16 ? (c4 & 15) : c4;
    const int tx = tx0 + lc4;
    const int Wp = g.tiles_x * (g.tw16 ? 16 : TW), Hp = g.tiles_y * (g.tw16 ? 8 : TH);
    const bool split = (SPLIT == 1 || (SPLIT == 2 && g.nsplit > 1));
    const bool vec = g.vec4 != 0;       // 16-byte path of the fused epilogue (decided on the host, make_cp)
    // One 32x32 accumulator tile at a time, called with literal (a, b): a loop over acc[a][b] that the compiler does not fully unroll
    // sends the accumulators to scratch memory -- in the MAIN loop as well.  Split and fused forms are separate code paths behind one
    // uniform branch, so that neither keeps the other's operands alive (the multi-problem kernel is at the SGPR limit).
    auto stage = [&](const f32x16& A) {
#pragma unroll
        for (int r = 0; r < 16; r++) Tt[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + l31] = abl_fix(A[r]);
        __builtin_amdgcn_wave_barrier();
    };
    if (split) {
        auto tile = [&](const f32x16& A, const int a, const int b) {
            stage(A);
            int ej, ty;
            const bool rowok = rowmap(ty0 + rowstep * (row0 + b) + lr4, ej, ty);
            // the slab is padded so that every store is aligned and in bounds, but the padding is never read: quads that start
            // outside the image are not written (the 2x7 / 4x13 layers would otherwise write 5-18x their partial sums)
            const bool live = rowok && ty < g.OHt && tx < g.OWt;
            float* pb = g.part + (long)bz_in * g.part_stride + (((long)(n + ej) * g.M + (m0 + wm * WM + a * 32 + rsub)) * Hp + ty) * Wp + tx;
            const long mstep = (long)8 * Hp * Wp;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 v = *reinterpret_cast<const float4*>(&Tt[(8 * k + rsub) * 32 + c4]);
                if (m0 + wm * WM + a * 32 + 8 * k + rsub < g.M && live) {
                    *reinterpret_cast<float4*>(pb + k * mstep) = v;
#ifdef CC_ABLATE_STORE
                    *reinterpret_cast<volatile float4*>(pb + k * mstep) = v;
#endif
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        tile(acc[0][0], 0, 0);
        if constexpr (TN > 1) tile(acc[0][1], 0, 1);
        if constexpr (TM > 1) {
            tile(acc[1][0], 1, 0);
            if constexpr (TN > 1) tile(acc[1][1], 1, 1);
        }
        return;
    }
    const bool hr = g.res != nullptr, ha = g.add != nullptr;
    auto tile = [&](const f32x16& A, const int a, const int b) {
        stage(A);
        int ej, ty;
        const bool rowok = rowmap(ty0 + rowstep * (row0 + b) + lr4, ej, ty);
        const bool inside = rowok && ty < g.OHt && tx < g.OWt;
        const int n = STK ? n_tile + ej : n_tile;                              // (shadows the tile's first image)
        const long pix = (long)(g.oy0 + g.so * ty) * g.OW + (g.ox0 + g.so * tx);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 v = *reinterpret_cast<const float4*>(&Tt[(8 * k + rsub) * 32 + c4]);
            const int m = m0 + wm * WM + a * 32 + 8 * k + rsub;
            if (m < g.M && inside) {
                const long o = (long)m * y_cs + pix;
                const float bias = g.bias ? g.bias[m] : 0.f;
                const float v0 = v.x + bias, v1 = v.y + bias, v2 = v.z + bias, v3 = v.w + bias;
                if (vec && tx + 3 < g.OWt) {
                    float4 rr = make_float4(0.f, 0.f, 0.f, 0.f), aa = rr;
                    if (hr) rr = *reinterpret_cast<const float4*>(g.res + (long)n * g.res_bs + o);
                    if (ha) aa = *reinterpret_cast<const float4*>(g.add + (long)n * g.add_bs + o);
                    float4 out;
                    out.x = conv_tail(v0, hr, rr.x, g.res_mul, g.act, g.act_a, g.act_b, aa.x);
                    out.y = conv_tail(v1, hr, rr.y, g.res_mul, g.act, g.act_a, g.act_b, aa.y);
                    out.z = conv_tail(v2, hr, rr.z, g.res_mul, g.act, g.act_a, g.act_b, aa.z);
                    out.w = conv_tail(v3, hr, rr.w, g.res_mul, g.act, g.act_a, g.act_b, aa.w);
                    *reinterpret_cast<float4*>(g.y + (long)n * g.y_bs + o) = out;
#ifdef CC_ABLATE_STORE
                    *reinterpret_cast<volatile float4*>(g.y + (long)n * g.y_bs + o) = out;
#endif
                } else {
                    const long st = g.so;
                    float* yo = g.y + (long)n * g.y_bs + o;
                    const float* ro = hr ? g.res + (long)n * g.res_bs + o : nullptr;
                    const float* ao = ha ? g.add + (long)n * g.add_bs + o : nullptr;
                    yo[0] = conv_tail(v0, hr, hr ? ro[0] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ao ? ao[0] : 0.f);
                    if (tx + 1 < g.OWt) yo[st] = conv_tail(v1, hr, hr ? ro[st] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ao ? ao[st] : 0.f);
                    if (tx + 2 < g.OWt) yo[2 * st] = conv_tail(v2, hr, hr ? ro[2 * st] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ao ? ao[2 * st] : 0.f);
                    if (tx + 3 < g.OWt) yo[3 * st] = conv_tail(v3, hr, hr ? ro[3 * st] : 0.f, g.res_mul, g.act, g.act_a, g.act_b, ao ? ao[3 * st] : 0.f);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    tile(acc[0][0], 0, 0);
    if constexpr (TN > 1) tile(acc[0][1], 0, 1);
    if constexpr (TM > 1) {
        tile(acc[1][0], 1, 0);
        if constexpr (TN > 1) tile(acc[1][1], 1, 1);
    }
}

// Workgroup -> (tile, channel block, split) in XCD order (cc_common.h).  Input-major (wmajor 0): the tiles in XCD order, channel block
// and split as dispatched -- an XCD's L2 sees a contiguous run of tiles (shared halos) and, whenever gridDim.x is a multiple of 8, all
// channel blocks and splits of a tile (same input patch).  Weight-major (wmajor 1): the whole grid in XCD order, tiles fastest -- an
// XCD works through ALL tiles of a contiguous range of (channel block, split) pairs, so a slice of the weight image is streamed from
// HBM by one XCD instead of by all eight (512 -> 512 3x3 on 8x26: 9.4 MB of weights against 1.7 MB of input).
__device__ __forceinline__ void conv_xcd_blocks(int wmajor, int& bx, int& by, int& bz) {
    bx = (int)blockIdx.x; by = (int)blockIdx.y; bz = (int)blockIdx.z;
    if constexpr (!(CC_XCD_MASK & 1)) return;
    if (wmajor < 0) return;
    if (wmajor) {
        const int gx = (int)gridDim.x, gy = (int)gridDim.y;
        const int w = cc_xcd_order(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        const int r = w / gx;
        bx = w - r * gx;
        bz = r / gy;
        by = r - bz * gy;
    } else {
        bx = cc_xcd_order(bx, (int)gridDim.x);
    }
}

// min 4 waves per SIMD (<= 128 registers): the accumulators then live in VGPRs (95-99 registers in total, no spill) instead of
// 64 AGPRs + 72-85 VGPRs, and four 40 KB workgroups fit a CU (CC_PATCH_LB: A/B builds, tools/)
#ifndef CC_PATCH_LB
#define CC_PATCH_LB 4
#endif
template <int BM, int CK, int TPS, int SPLIT>
__global__ __launch_bounds__(256, CC_PATCH_LB) void k_conv_patch(CP g) {
    int bx, by, bz;
    conv_xcd_blocks(g.wmajor, bx, by, bz);
    conv_patch_body<BM, CK, TPS, SPLIT, 0>(g, bx, by, bz);
}
template <int BM, int TPS, int SPLIT>          // stacked tiny maps (8-channel chunks only)
__global__ __launch_bounds__(256, CC_PATCH_LB) void k_conv_patch_stk(CP g) {
    int bx, by, bz;
    conv_xcd_blocks(g.wmajor, bx, by, bz);
    conv_patch_body<BM, 8, TPS, SPLIT, 1>(g, bx, by, bz);
}

// The (up to) four output-parity classes of a stride-2 data-gradient / transposed convolution in ONE launch:
// blockIdx.x ranges over the classes' tiles back to back (each class has its own geometry, weight image and partial
// slabs; the small-map layers are launch-bound, and four quarter-size grids in a row under-fill the chip).
// ... and, more generally, up to MAXCLS independent problems of one tile configuration in ONE launch: the G same-shaped
// convolutions of a network's parallel branches (Back2Future's decoder_fwd / decoder_bwd / decoder_occ at one pyramid level,
// its a / b / c feature streams) times their parity classes.  The deep pyramid levels are 1-4 GFLOP problems: one launch per
// group instead of one per branch triples the work per launch, needs a third of the split-K (partial-slab traffic) and
// removes two thirds of the ~5 us launch floors.
constexpr int MAXCLS = 12;
struct CPM {
    CP c[MAXCLS];
    int n;
    int bx_end[MAXCLS];
    int wmajor;            // as CP::wmajor, for the launch (the classes' weights against their inputs)
};

// (a macro, not a function taking the argument block by reference: the kernel reads its class descriptor from the kernel-argument
// segment, and the wrapper cost nine more scalar spills)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
#define CC_MULTI_DESC(a, k) (*(reinterpret_cast<const CP*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(CPM, c)) + (k)))
#else
#define CC_MULTI_DESC(a, k) ((a).c[k])
#endif
// the class descriptor is read straight from the kernel-argument segment (scalar loads at a run-time offset): indexing the
// by-value argument `a.c[k]` makes the compiler copy descriptors to scratch memory once the body is large
#define CC_MULTI_BODY(BM_, CK_, TPS_, STK_)                                                                                   \
    int bx, by, bz;                                                                                                           \
    conv_xcd_blocks(a.wmajor ? 1 : -1, bx, by, bz);      /* -1: as dispatched (the class's tiles are put in XCD order below) */  \
    int k = 0, first = 0, end = a.bx_end[0];                                                                                  \
    _Pragma("unroll") for (int q = 0; q < MAXCLS - 1; q++)                                                                    \
        if (q + 1 < a.n && bx >= a.bx_end[q]) { k = q + 1; first = a.bx_end[q]; end = a.bx_end[q + 1]; }                      \
    const CP& g = CC_MULTI_DESC(a, k);                                                                                        \
    if (bz >= g.nsplit || by * BM_ >= g.Mpad) return;     /* grid.y / grid.z are the launch's maxima */                        \
    bx -= first;                                                                                                              \
    if ((CC_XCD_MASK & 1) && !a.wmajor) bx = cc_xcd_order(bx, end - first);                                                   \
    conv_patch_body<BM_, CK_, TPS_, 2, STK_>(g, bx, by, bz);

template <int BM, int CK, int TPS>
__global__ __launch_bounds__(256, CC_PATCH_LB) void k_conv_patch_multi(CPM a) { CC_MULTI_BODY(BM, CK, TPS, 0) }
template <int BM, int TPS>                     // at least one class with stacked tiny maps (the others have ipt = 1)
__global__ __launch_bounds__(256, CC_PATCH_LB) void k_conv_patch_multi_stk(CPM a) { CC_MULTI_BODY(BM, 8, TPS, 1) }

// y[lattice pixel] = act(bias + res + sum_k part[k])  (second, deterministic stage of split-K)
__global__ __launch_bounds__(256) void k_splitk_epilogue(const float* __restrict__ part, int nsplit, long part_stride,
                                                         const float* __restrict__ bias, const float* __restrict__ res,
                                                         float* __restrict__ y, int M, int OHt, int OWt, int so, int oy0,
                                                         int ox0, int OH, int OW, long y_bs, long res_bs, long total,
                                                         int act, float act_a, float act_b, int res_mul,
                                                         const float* add, long add_bs, int Hp, int Wp) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int HWt = OHt * OWt;
    const long per = (long)M * HWt;
    const int n = (int)(e / per);
    const long r = e - (long)n * per;
    const int m = (int)(r / HWt);
    const int t = (int)(r - (long)m * HWt);
    const int ty = t / OWt, tx = t - ty * OWt;
    // the slabs are padded to whole tiles: [split][n][m][Hp][Wp]
    const long po = (((long)n * M + m) * Hp + ty) * Wp + tx;
    // eight partial loads in flight, added in split order (a one-by-one loop is a chain of nsplit dependent HBM/L2 latencies and
    // this kernel is nothing else)
    float v = 0.f;
    {
        for (int k = 0; k < nsplit; k += 8) {
            float p8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p8[u] = (k + u < nsplit) ? part[(long)(k + u) * part_stride + po] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k + u < nsplit) v += p8[u];
        }
    }
    const long o = (long)m * OH * OW + (long)(oy0 + so * ty) * OW + (ox0 + so * tx);
    if (bias) v += bias[m];
    y[(long)n * y_bs + o] = conv_tail(v, res != nullptr, res ? res[(long)n * res_bs + o] : 0.f, res_mul, act, act_a, act_b,
                                      add ? add[(long)n * add_bs + o] : 0.f);
}

// every class carries its own geometry and epilogue (round 3: the problems of one launch may come from different layers of
// different networks -- cc_conv2d_list)
struct EPC {
    const float* part; const float* bias; const float* res; const float* add; float* y;
    int nsplit; long part_stride; int OHt, OWt, oy0, ox0; long total;
    int Hp, Wp;                      // padded slab rows / pitch (whole tiles)
    int M, so, OH, OW; long y_bs, res_bs, add_bs;
    int act; float act_a, act_b;
    int res_mul;
};
struct EPM {
    EPC c[MAXCLS];
    int n;
    int bx_end[MAXCLS];
};

__global__ __launch_bounds__(256) void k_splitk_epilogue_multi(EPM a) {
    int k = 0, first = 0;
#pragma unroll
    for (int q = 0; q < MAXCLS - 1; q++)
        if (q + 1 < a.n && (int)blockIdx.x >= a.bx_end[q]) { k = q + 1; first = a.bx_end[q]; }
    const EPC& c = a.c[k];
    const long e = (long)((int)blockIdx.x - first) * 256 + threadIdx.x;
    if (e >= c.total) return;
    const int HWt = c.OHt * c.OWt;
    const long per = (long)c.M * HWt;
    const int n = (int)(e / per);
    const long r = e - (long)n * per;
    const int m = (int)(r / HWt);
    const int t = (int)(r - (long)m * HWt);
    const int ty = t / c.OWt, tx = t - ty * c.OWt;
    const long po = (((long)n * c.M + m) * c.Hp + ty) * c.Wp + tx;
    float v = 0.f;
    {
        for (int z = 0; z < c.nsplit; z += 8) {      // see k_splitk_epilogue
            float p8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p8[u] = (z + u < c.nsplit) ? c.part[(long)(z + u) * c.part_stride + po] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (z + u < c.nsplit) v += p8[u];
        }
    }
    const long o = (long)m * c.OH * c.OW + (long)(c.oy0 + c.so * ty) * c.OW + (c.ox0 + c.so * tx);
    if (c.bias) v += c.bias[m];
    c.y[(long)n * c.y_bs + o] = conv_tail(v, c.res != nullptr, c.res ? c.res[(long)n * c.res_bs + o] : 0.f, c.res_mul, c.act, c.act_a, c.act_b,
                                          c.add ? c.add[(long)n * c.add_bs + o] : 0.f);
}

static int dbg_flag_early(const char* name) { return cctools::env_flag(name); }

struct ConvPlan {
    bool use_patch;
    int wino;                  // Winograd F(2x2, 3x3) kernel (wino.hip): wn holds its plan, wp_floats the size of the U image
    ccint::WinoPlan wn;
    int Hp, Wp;                // rows / pitch of the split-K partial slabs [split][n][m][Hp][Wp]
    int tw16;
    int ipt, phi;              // stacked tiny maps (CP::ipt)
    int bm, ck, tps, Mpad, Cpad, PH, PWr, PS, ymin, xmin, tiles_x, tiles_y, nsplit, cps, aligned, shift;
    size_t smem, wp_floats, part_floats;
    int wpad;                  // Winograd over a zero-padded copy of the input (maps whose width is not a multiple of 4): its row pitch
    size_t pad_floats;         // ... and size (behind the partial slabs in the workspace)
};

static int env_int_early(const char* name, int dflt) { return cctools::env_int(name, dflt); }

// pixel tiles of a problem (grid.x of its launch): stacked tiny maps take one tile per ipt images
inline long conv_tiles(const GG& g, const ConvPlan& p) { return (long)((g.B + p.ipt - 1) / p.ipt) * p.tiles_x * p.tiles_y; }

// mult: number of same-shaped problems that share the launch (split-K only has to fill what they leave empty)
// 3x3 / stride 1 / pad 1 on the full lattice, taps forwards (conv2d) or backwards (its data-gradient)
inline bool wino_geometry(const GG& g) {
    return g.Rt == 3 && g.St == 3 && g.si == 1 && g.so == 1 && g.oy0 == 0 && g.ox0 == 0 && (g.dstep == 1 || g.dstep == -1) &&
           g.dy0 == -g.dstep && g.dx0 == -g.dstep && g.IH == g.OH && g.IW == g.OW && g.OHt == g.OH && g.OWt == g.OW && g.Cin > 0 &&
           (long)g.B * g.Cin * g.IH * g.IW < (1l << 26);
}

inline ConvPlan plan_conv(const GG& g, int mult = 1) {
    ConvPlan p = {};
    p.ipt = 1;
    if (wino_geometry(g)) {
        // the algorithm is a function of the geometry alone (the per-step weight image is laid out for it); `mult` only moves split-K
        const ccint::WinoPlan w = ccint::wino_plan(g.B, g.Cin, g.IH, g.IW, g.M, mult);
        if (w.ok) {
            p.wino = 1;
            p.wn = w;
            p.use_patch = true;
            p.bm = ccwino::WBM; p.ck = ccwino::WCK; p.tps = 1;
            p.Mpad = w.Mpad; p.Cpad = w.Cpad;
            p.nsplit = w.nsplit; p.cps = w.cps;
            p.Hp = w.Hp; p.Wp = w.Wp;
            p.wp_floats = w.u_floats;
            p.part_floats = w.part_floats;
            return p;
        }
        // Maps whose width is not a multiple of 4 (the 8x26 level of DispResNet6: 512- / 1024-channel layers, 4 GFLOP each on the direct
        // kernel + a split-K epilogue): the Winograd kernel stages aligned 16-byte row pieces, so the input is first copied into rows
        // padded with zeros to the next width it takes (k_pad_rows; zero columns ARE the convolution's padding) and the launch is
        // always split-K: the partial slabs have the padded pitch, and the deterministic epilogue kernel that sums them writes the
        // real output.  Large layers only (the copy, 23 % empty tile columns and the forced second pass have to pay), never for the
        // grouped launches of parallel branches (they share one direct launch today).
        if (!dbg_flag_early("CC_NO_WINO_PAD") && (g.IW % 4) != 0 && g.IH >= 2 && g.M >= env_int_early("CC_WINOP_MINM", 256) &&
            g.Cin >= env_int_early("CC_WINOP_MINC", 256)) {
            int wp = (g.IW + 3) & ~3;
            while (!(wp / 2 >= 16 || wp / 2 == 8)) wp += 4;
            if (4 * (wp - g.IW) <= wp && (long)g.B * ((g.IH + 1) / 2) * (wp / 2) >= env_int_early("CC_WINOP_MINQ", 64)) {
                ccint::WinoPlan wq = ccint::wino_plan(g.B, g.Cin, g.IH, wp, g.M, mult);
                if (wq.ok) {
                    if (wq.nsplit < 2) {                    // the epilogue pass is what un-pads the output
                        if (wq.tile == 2) wq.tile = 1;      // (the eight-wave instance halves the reduction itself: no slices across workgroups)
                        wq.cps = (wq.nchunk + 1) / 2;
                        wq.nsplit = (wq.nchunk + wq.cps - 1) / wq.cps;
                        wq.part_floats = (size_t)wq.nsplit * g.B * g.M * wq.Hp * wq.Wp;
                    }
                    if (wq.nsplit >= 2) {
                        p.wino = 1;
                        p.wn = wq;
                        p.use_patch = true;
                        p.bm = ccwino::WBM; p.ck = ccwino::WCK; p.tps = 1;
                        p.Mpad = wq.Mpad; p.Cpad = wq.Cpad;
                        p.nsplit = wq.nsplit; p.cps = wq.cps;
                        p.Hp = wq.Hp; p.Wp = wq.Wp;
                        p.wp_floats = wq.u_floats;
                        p.part_floats = wq.part_floats;
                        p.wpad = wp;
                        p.pad_floats = ((size_t)g.B * g.Cin * g.IH * wp + 3) & ~(size_t)3;
                        return p;
                    }
                }
            }
        }
    }
    p.bm = pick_bm_fwd(g.M);
    {   // a narrower channel tile when it saves >= 25 % of the PADDED output channels: M = 65 / 96 -> 3 x 32 instead of 128,
        // 129 -> 3 x 64 instead of 256, 260 -> 9 x 32 instead of 384 (concatenations with a 1-2 channel map, the 96-channel
        // decoder layers): -0.33 ms/step (r3s3 A/B; thresholds 12-25 % equal, 35 % loses it)
        const int thr = env_int_early("CC_CONV_BM_PADSAVE", 25);
        if (thr > 0 && p.bm > 32) {
            const int cur = ((g.M + p.bm - 1) / p.bm) * p.bm;
            for (int b2 = p.bm / 2; b2 >= 32; b2 /= 2) {
                const int m2 = ((g.M + b2 - 1) / b2) * b2;
                if ((cur - m2) * 100 >= thr * cur) { p.bm = b2; break; }
            }
        }
    }
    {   // tile shape: 4 x 32 or 8 x 16 lattice pixels, whichever covers the map with fewer padded pixels
        const long a32 = (long)((g.OWt + 31) / 32) * 32 * (((g.OHt + 3) / 4) * 4);
        const long a16 = (long)((g.OWt + 15) / 16) * 16 * (((g.OHt + 7) / 8) * 8);
        // ties (all maps of <= 16x52: both shapes pad them equally) go to 8 x 16: -0.07 ms/step (r3s3)
        p.tw16 = ((a16 < a32 || (a16 == a32 && env_int_early("CC_CONV_TW16_TIES", 1))) && !dbg_flag_early("CC_CONV_NO_TW16")) ? 1 : 0;
    }
    // Maps of <= 4 lattice rows (DispResNet6's 4x13 / 2x7 / 1x4 levels and their parity classes): one image fills 1-4 of the 8 tile
    // rows and the matrix cores multiply padding (a 2x7 map: 14 live pixels of 128).  The 8 x 16 tile then stacks the rows of
    // 8 / OHt consecutive images, each with its own halo rows in the patch.
    p.ipt = 1;
    if (g.OHt <= 4 && g.B >= 2 && env_int_early("CC_CONV_STACK", 1) && !dbg_flag_early("CC_CONV_CK16")) {
        p.tw16 = 1;
        p.ipt = 8 / g.OHt < g.B ? 8 / g.OHt : g.B;
    }
    const int th = p.tw16 ? 8 : TH, tw = p.tw16 ? 16 : TW;
    {   // few pixel tiles: shrink the channel tile (more workgroups, every one over the whole reduction) before resorting to
        // split-K (partial slabs + an epilogue launch); CC_CONV_BM_MINBLOCKS: block count below which the tile is halved
        const long tiles = (long)g.B * ((g.OWt + tw - 1) / tw) * ((g.OHt + th - 1) / th);
        const int thr = env_int_early("CC_CONV_BM64_BELOW", 0);
        if (p.bm == 128 && tiles * ((g.M + 127) / 128) < thr) p.bm = 64;
        const int minb = env_int_early("CC_CONV_BM_MINBLOCKS", 0);
        while (p.bm > 32 && tiles * ((g.M + p.bm - 1) / p.bm) < minb) p.bm /= 2;
    }
    const int ylast = g.dy0 + (g.Rt - 1) * g.dstep, xlast = g.dx0 + (g.St - 1) * g.dstep;
    p.ymin = g.dy0 < ylast ? g.dy0 : ylast;
    p.xmin = g.dx0 < xlast ? g.dx0 : xlast;
    const int ymax = g.dy0 < ylast ? ylast : g.dy0, xmax = g.dx0 < xlast ? xlast : g.dx0;
    p.PH = (th - 1) * g.si + (ymax - p.ymin) + 1;
    p.phi = 0;
    if (p.ipt > 1) {
        p.phi = (g.OHt - 1) * g.si + (ymax - p.ymin) + 1;
        p.PH = p.ipt * p.phi;
    }
    p.PWr = (tw - 1) * g.si + (xmax - p.xmin) + 1;
    // 16-byte aligned variant: start every patch row at the 4-float boundary at or below its first column
    p.aligned = (g.IW % 4 == 0) && !dbg_flag_early("CC_NO_ALIGNED_PATCH");
    p.shift = 0;
    if (p.aligned) {
        p.shift = ((p.xmin % 4) + 4) % 4;                  // si * tx0 is a multiple of 4
        p.PWr = ((p.shift + p.PWr + 3) / 4) * 4;
    }
    p.PS = ((p.PH * p.PWr + 63) / 64) * 64;
    if (p.aligned) p.PS = ((p.PH * p.PWr + 255) / 256) * 256;     // whole 64-lane x 16-byte DMA instructions per channel
    // 8-channel chunks: 20-40 KB of LDS per workgroup -> 3 (BM = 128, register-limited) to 7 workgroups per CU.  Measured
    // against 16-channel chunks (80 KB, two per CU, half as many barriers): -0.8 ms/step in total (r02g-r02i A/Bs);
    // CC_CONV_CK16=1 restores the round-1 plan (16 wherever one stage fits in 64 KB).
    p.ck = 8;
    auto smem_of = [&](int ck, int tps) { return (size_t)(2 * tps * ck * p.bm + 2 * ck * p.PS) * sizeof(float); };
    if (dbg_flag_early("CC_CONV_CK16") && smem_of(16, 1) <= 64 * 1024) p.ck = 16;
    // three taps per pipeline stage when the extra weight buffers still leave two workgroups per CU (2 x 80 KB)
    p.tps = (g.Rt * g.St >= 3 && smem_of(p.ck, 3) <= 80 * 1024 && !dbg_flag_early("CC_CONV_TPS1")) ? 3 : 1;
    p.smem = smem_of(p.ck, p.tps);
    if (p.bm >= 32 && p.smem < 16384) p.smem = 16384;        // the epilogue transposes one 32x32 tile per wave through LDS
    p.use_patch = (p.smem <= 150 * 1024) && g.Cin > 0;
    p.Mpad = ((g.M + p.bm - 1) / p.bm) * p.bm;
    p.Cpad = ((g.Cin + p.ck - 1) / p.ck) * p.ck;
    p.tiles_x = (g.OWt + tw - 1) / tw;
    p.tiles_y = (g.OHt + th - 1) / th;
    p.wp_floats = (size_t)g.Rt * g.St * p.Cpad * p.Mpad;
    const long blocks = conv_tiles(g, p) * (p.Mpad / p.bm) * (mult > 1 ? mult : 1);
    const int nchunk = p.Cpad / p.ck;
    p.nsplit = 1;
    p.cps = nchunk;
    if (blocks < env_int_early("CC_CONV_SPLIT_BELOW", 384) && nchunk >= 4 && !(g.so != 1 && dbg_flag_early("CC_DBG_NO_PARITY_SPLIT"))) {
        long want = (env_int_early("CC_CONV_SPLIT_TARGET", 512) + blocks - 1) / blocks;
        if (want > nchunk / env_int_early("CC_CONV_MINCHUNKS", 1)) want = nchunk / env_int_early("CC_CONV_MINCHUNKS", 1);
        if (want > env_int_early("CC_CONV_MAXSPLIT", 32)) want = env_int_early("CC_CONV_MAXSPLIT", 32);
        if (want >= 2) {
            p.cps = (int)((nchunk + want - 1) / want);
            p.nsplit = (nchunk + p.cps - 1) / p.cps;
        }
    }
    // Wave quantisation (round 3): a launch of 257..~1000 workgroups runs as ceil(blocks / 256) "rounds" on the 256 CUs -- the
    // MFMA pipes of a CU are saturated by one workgroup, so k co-resident workgroups take k times as long -- e.g. the 336
    // workgroups of Back2Future's level-3 decoder groups (32x104 maps, G = 3) cost two rounds for 1.3 rounds of work.  A modest
    // split-K re-balances them when the reduction is long enough to pay for the partial slabs.  Cost model (microseconds):
    //   T(ns) = ceil(blocks * ns / 256) * (stages / ns) * t_stage(BM) + 2.5  [+ 2 * ns * out_bytes / 3 TB/s + 5 when ns > 1]
    // calibrated on the split-K launches of the step (512->512 on 8x26: model 59 us, measured 63).
    if (blocks >= 256 && nchunk >= 8 && cctools::env_int("CC_CONV_BALANCE", 1)) {
        const int T = g.Rt * g.St;
        const double mfma_cyc = (p.bm == 128 ? 3072.0 : p.bm == 64 ? 1536.0 : p.bm == 32 ? 768.0 : 384.0) * (p.tps == 3 ? 1.0 : 1.0 / 3.0);
        const double t_stage = mfma_cyc / 2100.0;
        const double stages = (double)nchunk * ((T + p.tps - 1) / p.tps);
        const double out_bytes = 4.0 * g.B * g.M * g.OHt * g.OWt * (mult > 1 ? mult : 1);
        auto cost = [&](int ns) {
            const double k = (double)((blocks * ns + 255) / 256);
            double t = k * (stages / ns) * t_stage + 2.5;
            if (ns > 1) t += 2.0 * ns * out_bytes / 3.0e6 + 5.0;
            return t;
        };
        int best = 1;
        const int cap = nchunk / 4 < 8 ? nchunk / 4 : 8;
        for (int ns = 2; ns <= cap; ns++)
            if (cost(ns) < cost(best)) best = ns;
        if (best > 1 && cost(best) < 0.01 * cctools::env_int("CC_CONV_BALANCE_PCT", 97) * cost(1)) {
            p.cps = (nchunk + best - 1) / best;
            p.nsplit = (nchunk + p.cps - 1) / p.cps;
        }
    }
    // partial slabs are padded to whole tiles (16-byte stores without guards): [split][n][m][tiles_y * th][tiles_x * tw]
    p.Hp = p.tiles_y * th;
    p.Wp = p.tiles_x * tw;
    p.part_floats = p.nsplit > 1 ? (size_t)p.nsplit * g.B * g.M * p.Hp * p.Wp : 0;
    return p;
}

inline size_t conv_ws_floats(const ConvPlan& p) { return 64 + p.wp_floats + p.part_floats + p.pad_floats; }

// ------------------------------------------------------------------ weight gradient
constexpr int MAXGRP = 4;       // same-shaped weight-gradient problems per launch (parallel branches of a network)
struct WG {
    const float* a;   // "dY-like" tensor [B, M, AH, AW] (batch stride a_bs), sampled on the full lattice (ty, tx)
    const float* x;   // gathered tensor [B, Cin, IH, IW]
    float* out;       // partial tiles ws[split][M][N] (or the final gradient when nsplit == 1 -> strided store)
    const float* ga[MAXGRP]; const float* gxp[MAXGRP]; float* gout[MAXGRP];   // per-problem pointers (blockIdx.z / nsplit)
    int nsplit;
    int B, M, AH, AW; long a_bs;
    int Cin, IH, IW; long x_bs;
    int Rt, St, dy0, dx0, dstep, si;
    long o_sm, o_sc; int o_ri, o_sj;     // gradient strides (used when direct == 1)
    int direct, accum;
    int pix_per_split;
};

template <int BM>
__device__ __forceinline__ void wgrad_body(const WG& g, const int bx_, const int by_, const int bz_) {
    const int grp = bz_ / g.nsplit, zsplit = bz_ - grp * g.nsplit;
    const float* __restrict__ a_ = g.ga[grp];
    const float* __restrict__ x_ = g.gxp[grp];
    float* __restrict__ out_ = g.gout[grp];
    constexpr int WM = (BM >= 64) ? BM / 2 : 32;
    constexpr int WN = (BM >= 64) ? 64 : 32;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM + 4, BP = BN + 4;
    constexpr int AQ = BM / 16;
    __shared__ float As[2][BK * AP];   // As[pp][m]
    __shared__ float Bs[2][BK * BP];   // Bs[pp][jn]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (BM >= 64) ? (wid >> 1) : 0;
    const int wn = (BM >= 64) ? (wid & 1) : wid;
    const int m0 = by_ * BM;
    const int n0 = bx_ * BN;
    const int RS = g.Rt * g.St;
    const int Ntot = g.Cin * RS;
    const int HWa = g.AH * g.AW;
    const long Ptot = (long)g.B * HWa;
    const long pbeg = (long)zsplit * g.pix_per_split;
    long pend = pbeg + g.pix_per_split;
    if (pend > Ptot) pend = Ptot;
    const int x_cs = g.IH * g.IW;

    // loader roles: pp = tid & 15 (pixel within the chunk), row group = tid >> 4 (16 groups)
    const int pp = tid & 15, rgp = tid >> 4;
    // B columns handled by this thread: jn = rgp + 16*q  -> constant over the pixel loop
    int xoff[8], tdy[8], tdx[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int jn = n0 + rgp + 16 * q;
        if (jn < Ntot) {
            const int c = jn / RS, rem = jn - c * RS;
            const int i = rem / g.St, j = rem - i * g.St;
            tdy[q] = g.dy0 + i * g.dstep;
            tdx[q] = g.dx0 + j * g.dstep;
            xoff[q] = c * x_cs + tdy[q] * g.IW + tdx[q];
        } else {
            tdy[q] = -(1 << 28);
            tdx[q] = 0;
            xoff[q] = 0;
        }
    }
    float ra[AQ], rb[8];
    unsigned okA = 0, okB = 0;
    // branch-free loads: invalid elements read element 0 of their tensor and are zeroed by a select AT STORE TIME
    // (hipcc otherwise wraps every predicated load in its own s_cbranch_execz block; and a select placed right after
    // the load would force s_waitcnt vmcnt(0) ahead of the MFMAs of the current chunk)
    auto load_chunk = [&](long pbase) {
        okA = 0;
        okB = 0;
        const long p = pbase + pp;
        const bool pv = p < pend;
        const long ps = pv ? p : 0;
        const int n = (int)(ps / HWa);
        const int t = (int)(ps - (long)n * HWa);
        const int ty = t / g.AW, tx = t - ty * g.AW;
        const long abase = (long)n * g.a_bs + ty * g.AW + tx;
#pragma unroll
        for (int q = 0; q < AQ; q++) {
            const int m = m0 + rgp + 16 * q;
            const bool ok = pv && (m < g.M);
            ra[q] = a_[ok ? abase + (long)m * HWa : 0];
            okA |= (ok ? 1u : 0u) << q;
        }
        const int iy0 = g.si * ty, ix0 = g.si * tx;
        const long xb = (long)n * g.x_bs + (long)iy0 * g.IW + ix0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int iy = iy0 + tdy[q], ix = ix0 + tdx[q];
            const bool ok = pv && ((unsigned)iy < (unsigned)g.IH) && ((unsigned)ix < (unsigned)g.IW);
            rb[q] = x_[ok ? xb + xoff[q] : 0];
            okB |= (ok ? 1u : 0u) << q;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int q = 0; q < AQ; q++) As[buf][pp * AP + rgp + 16 * q] = ((okA >> q) & 1u) ? ra[q] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) Bs[buf][pp * BP + rgp + 16 * q] = ((okB >> q) & 1u) ? rb[q] : 0.f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int l31 = lane & 31, lk = lane >> 5;
    const long nchunks = (pend > pbeg) ? (pend - pbeg + BK - 1) / BK : 0;
    if (nchunks > 0) {
        load_chunk(pbeg);
        store_chunk(0);
    }
    __syncthreads();
    for (long ch = 0; ch < nchunks; ch++) {
        const int buf = (int)(ch & 1);
        if (ch + 1 < nchunks) load_chunk(pbeg + (ch + 1) * BK);
        {
            float af[BK / 2][TM], bf[BK / 2][TN];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ks++) {
#pragma unroll
                for (int a = 0; a < TM; a++) af[ks][a] = As[buf][(2 * ks + lk) * AP + wm * WM + a * 32 + l31];
#pragma unroll
                for (int b = 0; b < TN; b++) bf[ks][b] = Bs[buf][(2 * ks + lk) * BP + wn * WN + b * 32 + l31];
            }
#pragma unroll
            for (int ks = 0; ks < BK / 2; ks++)
#pragma unroll
                for (int a = 0; a < TM; a++)
#pragma unroll
                    for (int b = 0; b < TN; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // epilogue: D col = lane&31 -> (c,i,j) column, row -> channel m
#pragma unroll
    for (int b = 0; b < TN; b++) {
        const int jn = n0 + wn * WN + b * 32 + l31;
        if (jn >= Ntot) continue;
        long obase;
        if (g.direct) {
            const int c = jn / RS, rem = jn - c * RS;
            const int i = rem / g.St, j = rem - i * g.St;
            obase = (long)c * g.o_sc + i * g.o_ri + j * g.o_sj;
        } else {
            obase = (long)zsplit * g.M * Ntot + jn;
        }
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < g.M) {
                    float* o = out_ + obase + (g.direct ? (long)m * g.o_sm : (long)m * Ntot);
                    *o = (g.direct && g.accum) ? (*o + acc[a][b][r]) : acc[a][b][r];
                }
            }
    }
}

template <int BM>
__global__ __launch_bounds__(256) void k_wgrad(WG g) {
    if constexpr (CC_XCD_MASK & 4) {
        // XCD order (cc_common.h) over the flattened grid, x fastest: the tiles of one (problem, pixel range) -- which gather the same
        // slices of dY and x -- run on one XCD
        const int gx = (int)gridDim.x, gy = (int)gridDim.y;
        const int b = cc_xcd_order((int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z), gx * gy * (int)gridDim.z);
        const int r = b / gx;
        wgrad_body<BM>(g, b - r * gx, r % gy, r / gy);
    } else {
        wgrad_body<BM>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
    }
}

// Problems of DIFFERENT shapes in one launch (the single-layer weight gradients a backward stage leaves parked until its end -- the
// stride-2 / 1x1 / small-map layers: 15-40 us launches of 30-600 workgroups each, mostly ramp-up and drain; cc_conv2d_wgrad_list).
// blockIdx.x ranges over the classes' grids back to back; a class's grid is flattened x-fastest.
constexpr int MAXWCLS = 12;
struct WGM { WG c[MAXWCLS]; int n; int bx_end[MAXWCLS]; int gx[MAXWCLS], gy[MAXWCLS]; };
template <int BM>
__global__ __launch_bounds__(256) void k_wgrad_multi(WGM a) {
    int k = 0, first = 0, end = a.bx_end[0];
#pragma unroll
    for (int q = 0; q < MAXWCLS - 1; q++)
        if (q + 1 < a.n && (int)blockIdx.x >= a.bx_end[q]) { k = q + 1; first = a.bx_end[q]; end = a.bx_end[q + 1]; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    const WG& g = *(reinterpret_cast<const WG*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WGM, c)) + k);
#else
    const WG& g = a.c[k];
#endif
    const int b = (CC_XCD_MASK & 4) ? cc_xcd_order((int)blockIdx.x - first, end - first) : (int)blockIdx.x - first;      // (as k_wgrad)
    const int gx = a.gx[k], gy = a.gy[k];
    const int bx = b % gx, r = b / gx;
    wgrad_body<BM>(g, bx, r % gy, r / gy);
}

// ------------------------------------------------------------------ weight gradient, patch-staged (main path)
// gw[m][c][i][j] = sum_{n,ty,tx} a[n][m][ty][tx] * x[n][c][si*ty + i - pad][si*tx + j - pad] as ONE GEMM PER TAP:
//   D_t[m][c] += A[m][pixel] * X_t[pixel][c],  X_t = the input patch shifted by tap t (never materialised).
// Workgroup = (BMW rows of dY) x (32 input channels) x (a group of <= TG taps), looping over 2 x 32 pixel tiles of
// its split; per tile dY (pixel-minor, row stride 65) and the 32-channel input patch (channel stride odd) are
// LDS-DMA'd, both MFMA operands are then conflict-free strided ds_reads (lane = m resp. lane = channel, k = pixel).
// Every MFMA is useful work (no im2col padding); accumulators: one 32x32 tile per (m-tile, tap), spread over the 4 waves.
constexpr int WTH = 2;          // lattice rows per pixel tile
constexpr int WPIX = WTH * 32;  // 64 pixels = 32 MFMA k-steps
constexpr int APS = WPIX + 1;   // dY row stride in LDS (odd -> bank = (m + p) mod 32)

struct WP {
    const float* a; const float* x; const float* zeros; float* ws;
    int B, M, AH, AW; long a_bs;
    int Cin, IH, IW; long x_bs;
    int R, S, si, pad;
    int PH, PWr, PSc, npos;
    int tiles_x, tiles_y, ntiles, tiles_per_split, nsplit;
    int TG, ngroups, Cp32, nbuf;
    int dbg;   // ablation switches (CC_WGRAD_DBG): 1 = skip the per-tile LDS-DMA, 2 = skip the MFMAs
};

// Wave specialisation: waves 0-3 only read LDS and issue MFMAs, waves 4-5 only issue the LDS-DMA of the NEXT pixel
// tile (their ~20-instruction address chains would otherwise sit in front of the MFMAs of an in-order wave).
constexpr int WG_THREADS = 384;

template <int BMW, int NT>
__global__ __launch_bounds__(384) void k_wgrad_patch(WP g) {
    constexpr int MT = BMW / 32;
    HIP_DYNAMIC_SHARED(float, smem)
    const int a_sz = BMW * APS, p_sz = 32 * g.PSc;
    float* As = smem;                         // [nbuf][BMW][APS]
    float* Ps = smem + g.nbuf * a_sz;         // [nbuf][32][PSc]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lk = lane >> 5;
    // blockIdx.x -> (m-tile, c-tile, tap group)
    int bx = blockIdx.x;
    const int grp = bx % g.ngroups;
    bx /= g.ngroups;
    const int ctile = bx % (g.Cp32 / 32);
    const int mtile = bx / (g.Cp32 / 32);
    const int m0 = mtile * BMW, c0 = ctile * 32;
    const int t_first = grp * g.TG;
    const int T = g.R * g.S;
    int ntap = T - t_first;
    if (ntap > g.TG) ntap = g.TG;
    const int x_cs = g.IH * g.IW, a_cs = g.AH * g.AW;
    const int pt_beg = blockIdx.z * g.tiles_per_split;
    int pt_end = pt_beg + g.tiles_per_split;
    if (pt_end > g.ntiles) pt_end = g.ntiles;
    const int RI = (g.npos + 63) >> 6;

    const bool loader = wid >= 4;
    const int lw = wid - 4;                    // loader wave index (0/1)
    // accumulator tiles of this wave: q = wid + 4*k -> (mt = q % MT, tap = q / MT)
    int my_mt[NT], my_tap[NT];
    f32x16 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int q = (wid & 3) + 4 * k;
        my_mt[k] = q % MT;
        my_tap[k] = q / MT;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[k][r] = 0.f;
    }

    auto load_tile = [&](int pt, int buf) {
        const int tile_x = pt % g.tiles_x;
        const int r2 = pt / g.tiles_x;
        const int tile_y = r2 % g.tiles_y;
        const int n = r2 / g.tiles_y;
        const int ty0 = tile_y * WTH, tx0 = tile_x * 32;
        // dY rows: one LDS-DMA per channel m (64 pixels = 2 lattice rows of 32)
        {
            const int ty = ty0 + lk, tx = tx0 + l31;
            const bool ok = (ty < g.AH) && (tx < g.AW);
            const float* an = g.a + (long)n * g.a_bs + (long)ty * g.AW + tx;
            float* dst = As + buf * a_sz;
            for (int mm = lw; mm < BMW; mm += 2) {
                const int m = m0 + mm;
                const float* src = (ok && m < g.M) ? an + (long)m * a_cs : g.zeros + lane;
                __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(src), CC_LDS_PTR(dst + mm * APS), 4, 0, 0);
            }
        }
        // input patch of 32 channels: positions pos = py*PWr + px, py < PH
        {
            const int gy0 = g.si * ty0 - g.pad, gx0 = g.si * tx0 - g.pad;
            const float* xn = g.x + (long)n * g.x_bs;
            float* dst = Ps + buf * p_sz;
            for (int r = 0; r < RI; r++) {
                const int pos = lane + 64 * r;
                const int py = pos / g.PWr, px = pos - py * g.PWr;
                const int iy = gy0 + py, ix = gx0 + px;
                const bool inb = ((unsigned)iy < (unsigned)g.IH) && ((unsigned)ix < (unsigned)g.IW);
                const long go = (long)iy * g.IW + ix;
                if (pos < g.npos) {                 // lanes past the patch stay out of the DMA (EXEC-masked)
                    for (int cc = lw; cc < 32; cc += 2) {
                        const int c = c0 + cc;
                        const float* src = (inb && c < g.Cin) ? xn + (long)c * x_cs + go : g.zeros + lane;
                        __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(src), CC_LDS_PTR(dst + cc * g.PSc + 64 * r), 4, 0, 0);
                    }
                }
            }
        }
    };

    if (pt_beg < pt_end) {
        if (loader) {
            load_tile(pt_beg, 0);
            CC_WAIT_VMCNT0();
        }
        __syncthreads();
        for (int pt = pt_beg; pt < pt_end; pt++) {
            const int buf = (g.nbuf == 2) ? ((pt - pt_beg) & 1) : 0;
            if (loader) {
                if (g.nbuf == 2 && pt + 1 < pt_end && !(g.dbg & 1)) {
                    load_tile(pt + 1, buf ^ 1);
                    CC_WAIT_VMCNT0();
                }
            } else if (!(g.dbg & 2)) {
                const float* Ab = As + buf * a_sz + l31 * APS + lk;          // lane = m row, k = pixel
                const float* Pb = Ps + buf * p_sz + l31 * g.PSc + g.si * lk; // lane = channel
                // branch-free inner loop: every accumulator slot multiplies (slots past the tap group re-do its last
                // tap and are dropped in the epilogue), all operands of a k-step are fetched before its MFMAs
                int toff[NT];
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    const int tt = my_tap[k] < ntap ? my_tap[k] : ntap - 1;
                    const int t = t_first + tt;
                    const int i = t / g.S, j = t - i * g.S;
                    toff[k] = i * g.PWr + j;
                }
#pragma unroll
                for (int row = 0; row < WTH; row++) {
                    const float* Ar = Ab + row * 32;
                    const float* Pr = Pb + (g.si * row) * g.PWr;
#pragma unroll 4
                    for (int ks = 0; ks < 16; ks++) {
                        float af[MT], bv[NT];
#pragma unroll
                        for (int a = 0; a < MT; a++) af[a] = Ar[a * 32 * APS + 2 * ks];
#pragma unroll
                        for (int k = 0; k < NT; k++) bv[k] = Pr[(2 * g.si) * ks + toff[k]];
#pragma unroll
                        for (int k = 0; k < NT; k++) {
                            const float av = (MT == 1) ? af[0] : (my_mt[k] ? af[MT - 1] : af[0]);   // no runtime register indexing
                            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[k], acc[k], 0, 0, 0);
                        }
                    }
                }
            }
            if (g.nbuf == 1) {
                __syncthreads();
                if (loader && pt + 1 < pt_end) {
                    load_tile(pt + 1, 0);
                    CC_WAIT_VMCNT0();
                }
            }
            __syncthreads();
        }
    }
    if (loader) return;
    // partial slabs: ws[split][t][m][c]  (c contiguous: D col = lane&31 = channel -> coalesced)
#pragma unroll
    for (int k = 0; k < NT; k++) {
        if (my_tap[k] >= ntap) continue;
        const int t = t_first + my_tap[k];
        float* o = g.ws + (((long)blockIdx.z * T + t) * g.M) * g.Cp32 + c0 + l31;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + my_mt[k] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < g.M) o[(long)m * g.Cp32] = acc[k][r];
        }
    }
}

// ------------------------------------------------------------------ weight gradient of 3x3 / stride 1 / pad 1 convs (main path)
// The layers that carry ~85 % of the step's weight-gradient FLOPs.  Same per-tap GEMM as k_wgrad_patch
//   D_(i,j)[m][c] += dY[m][pixel] * X[c][pixel + (i-1, j-1)]
// but every byte moves by 16-byte LDS-DMA (4x fewer DMA instructions -- the dword form is issue-bound, measured) and
// every MFMA operand comes from a conflict-free ds_read_b128:
//   * dY tile [BM m][2 x 32 px] as 16-byte chunks, XOR-swizzled by (m & 15) through the SOURCE address of the DMA
//     (LDS image stays lane-linear); a chunk = 4 pixels = the A operand of two k-steps (lane>>5 picks the pixel);
//   * input patch [32 c][4 rows][40 cols] (cols tx0-4 .. tx0+35: 16-byte aligned in global memory), channel stride 41
//     chunks (odd -> 16 consecutive channels hit 16 different bank quads); three chunks of one patch row hold the B
//     operands of the 3 taps of that row for 4 pixels;
//   * one wave per (tap row i, 32-row m tile): 3 accumulator tiles, 4 ds_read_b128 per 6 MFMAs.
constexpr int W3_PC = 41;       // chunks per channel in the patch (40 used + 1 pad)

struct W3 {
    const float* a; const float* x; const float* zeros; float* ws;
    const float* ga[MAXGRP]; const float* gxp[MAXGRP]; float* gws[MAXGRP];      // per-problem pointers (blockIdx.y)
    int B, M, AH, AW; long a_bs;
    int Cin; long x_bs;
    int tiles_x, tiles_y, ntiles, tiles_per_split, nsplit, Cpad, dbg;
};

// Workgroup = 4 waves (a 3- or 6-wave workgroup lands 2+2+1+1 on the SIMDs and caps at 75 % of the MFMA rate:
// tools/mfma_probe.hip measures 116 vs 155 TFLOP/s).  It covers MT m-tiles x CT channel-tiles (MT*CT = 4) x 3 tap rows
// = 12 (tap row, tile) groups, three per wave = 9 accumulators; every k-step pair costs 4 ds_read_b128 per 12 MFMAs.
template <int MT, int CT>
__global__ __launch_bounds__(256, 2) void k_wgrad3x3(W3 g) {
    const float* __restrict__ a_ = g.ga[blockIdx.y];
    const float* __restrict__ x_ = g.gxp[blockIdx.y];
    float* __restrict__ ws_ = g.gws[blockIdx.y];
    constexpr int BM = 32 * MT, BC = 32 * CT;
    constexpr int A_SLOTS = BM * 16;                           // 16-byte slots of the dY tile
    constexpr int P_SLOTS = ((BC * W3_PC + 63) / 64) * 64;     // rounded up so that every DMA instruction runs all 64 lanes
    HIP_DYNAMIC_SHARED(float, smem)
    float4* As = reinterpret_cast<float4*>(smem);              // [A_SLOTS]
    float4* Ps = reinterpret_cast<float4*>(smem) + A_SLOTS;    // [P_SLOTS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lk = lane >> 5;
    const int ctiles = g.Cpad / BC;
    const int ctile = blockIdx.x % ctiles, mtile = blockIdx.x / ctiles;
    const int m0 = mtile * BM, c0 = ctile * BC;
    const int HW = g.AH * g.AW;
    const int pt_beg = blockIdx.z * g.tiles_per_split;
    int pt_end = pt_beg + g.tiles_per_split;
    if (pt_end > g.ntiles) pt_end = g.ntiles;

    // this wave's three groups: gidx = wid + 4k -> tap row gidx % 3, tile gidx / 3 -> (mt, ct)
    int g_row[3], g_mt[3], g_ct[3];
    f32x16 acc[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int gi = wid + 4 * k;
        g_row[k] = gi % 3;
        const int tl = gi / 3;
        g_mt[k] = tl % MT;
        g_ct[k] = tl / MT;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[k][j][r] = 0.f;
    }

    auto load_tile = [&](int pt) {
        const int tile_x = pt % g.tiles_x;
        const int r2 = pt / g.tiles_x;
        const int tile_y = r2 % g.tiles_y;
        const int n = r2 / g.tiles_y;
        const int ty0 = tile_y * 2, tx0 = tile_x * 32;
        // dY: LDS slot s = m*16 + sc holds pixel chunk pc = sc ^ (m & 15) of row m  (pc = row*8 + col4)
        for (int s0 = wid * 64; s0 < A_SLOTS; s0 += 256) {
            const int sl = s0 + lane;
            const int mm = sl >> 4, sc = sl & 15;
            const int pc = sc ^ (mm & 15);
            const int ty = ty0 + (pc >> 3), tx = tx0 + 4 * (pc & 7);
            const int m = m0 + mm;
            const bool ok = (m < g.M) && (ty < g.AH) && (tx < g.AW);
            const float* src = ok ? a_ + (long)n * g.a_bs + (long)m * HW + (long)ty * g.AW + tx : g.zeros;
            __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(src), CC_LDS_PTR(As + s0), 16, 0, 0);
        }
        // patch: LDS slot s = c*41 + r, r = py*10 + ch (r == 40: pad)
        for (int s0 = wid * 64; s0 < P_SLOTS; s0 += 256) {
            const int sl = s0 + lane;
            const int cc = sl / W3_PC, r = sl - cc * W3_PC;
            const int py = r / 10, ch = r - py * 10;
            const int iy = ty0 - 1 + py, ix = tx0 - 4 + 4 * ch;
            const int c = c0 + cc;
            const bool ok = (cc < BC) && (r < 40) && (c < g.Cin) && ((unsigned)iy < (unsigned)g.AH) && ((unsigned)ix < (unsigned)g.AW);
            const float* src = ok ? x_ + (long)n * g.x_bs + (long)c * HW + (long)iy * g.AW + ix : g.zeros;
            __builtin_amdgcn_global_load_lds(CC_GLOBAL_PTR(src), CC_LDS_PTR(Ps + s0), 16, 0, 0);
        }
    };

    for (int pt = pt_beg; pt < pt_end; pt++) {
        if (pt > pt_beg) __syncthreads();                                 // everyone is done reading the previous tile
        if (!(g.dbg & 1) || pt == pt_beg) load_tile(pt);
        CC_WAIT_VMCNT0();
        __syncthreads();
        // MFMA k index (lane>>5) <-> the two HALVES of a 32-pixel row: lanes 0-31 take pixel chunk pq, lanes 32-63 chunk
        // pq+4, each through its own ds_read_b128 address -> element e of every chunk feeds MFMA e directly
        if (g.dbg & 2) continue;
#pragma unroll
        for (int row = 0; row < 2; row++) {
#pragma unroll 2
            for (int pq = 0; pq < 4; pq++) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int mrow = g_mt[k] * 32 + l31;
                    float4 av = As[mrow * 16 + ((row * 8 + pq + 4 * lk) ^ (mrow & 15))];
                    const float4* Pr = Ps + (g_ct[k] * 32 + l31) * W3_PC + (row + g_row[k]) * 10 + 4 * lk + pq;
                    float4 w0 = Pr[0], w1 = Pr[1], w2 = Pr[2];
                    CC_KEEP4(av); CC_KEEP4(w0); CC_KEEP4(w1); CC_KEEP4(w2);
                    const float a[4] = {av.x, av.y, av.z, av.w};
                    const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            acc[k][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], w[3 + j + e], acc[k][j], 0, 0, 0);
                }
            }
        }
    }
    // partial slabs ws[split][t][m][c], t = tap_row*3 + j   (same layout as k_wgrad_patch -> same reduce kernel)
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float* o = ws_ + (((long)blockIdx.z * 9 + (g_row[k] * 3 + j)) * g.M) * g.Cpad + c0 + g_ct[k] * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + g_mt[k] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < g.M) o[(long)m * g.Cpad] = acc[k][j][r];
            }
        }
    }
}

__global__ void k_zero64(float* p) { p[threadIdx.x] = 0.f; }

struct RG { const float* ws[MAXGRP]; float* gw[MAXGRP]; };

struct WPlan {
    bool ok;
    int bmw, nt, TG, ngroups, Cp32, PH, PWr, PSc, npos, nbuf, tiles_x, tiles_y, ntiles, nsplit, tps;
    size_t smem, ws_floats;
};

template <int BMW, int NT>
inline void launch_wgrad_patch(const WP& w, dim3 grid, size_t smem, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_patch<BMW, NT>), grid, dim3(WG_THREADS), smem, s, w);
}

static int dbg_flag(const char* name) { return cctools::env_flag(name); }

inline WPlan plan_wgrad(int B, int M, int AH, int AW, int Cin, int R, int S, int si) {
    WPlan p = {};
    const int T = R * S;
    p.bmw = (M > 32) ? 64 : 32;
    const int MT = p.bmw / 32;
    // NT accumulator tiles per wave: cover min(T, 9) taps per group
    const int tg_target = T < 9 ? T : 9;
    p.nt = (MT * tg_target + 3) / 4;
    if (p.nt > 5) p.nt = 5;
    p.TG = (4 * p.nt) / MT;
    if (p.TG > T) p.TG = T;
    p.ngroups = (T + p.TG - 1) / p.TG;
    p.Cp32 = ((Cin + 31) / 32) * 32;
    p.PH = (WTH - 1) * si + R;
    p.PWr = 31 * si + S;
    p.npos = p.PH * p.PWr;
    p.PSc = p.npos | 1;                       // odd channel stride
    auto smem_of = [&](int nbuf) { return (size_t)nbuf * (p.bmw * APS + 32 * p.PSc) * sizeof(float); };
    p.nbuf = 2;
    if (smem_of(2) > 150 * 1024) p.nbuf = 1;
    p.smem = smem_of(p.nbuf);
    // measured on MI355X (profiles/r01_*): the im2col-style k_wgrad is still ~2-8 % faster end to end than this
    // per-tap kernel (small-channel layers pad to 32 channels here); kept selectable for tuning: CC_WGRAD_PATCH=1
    p.ok = p.smem <= 150 * 1024 && dbg_flag("CC_WGRAD_PATCH");
    p.tiles_x = (AW + 31) / 32;
    p.tiles_y = (AH + WTH - 1) / WTH;
    p.ntiles = B * p.tiles_x * p.tiles_y;
    const long base = (long)((M + p.bmw - 1) / p.bmw) * (p.Cp32 / 32) * p.ngroups;
    long nsplit = (512 + base - 1) / base;
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    if (nsplit < 1) nsplit = 1;
    p.tps = (int)((p.ntiles + nsplit - 1) / nsplit);
    p.nsplit = (p.ntiles + p.tps - 1) / p.tps;
    p.ws_floats = 64 + (size_t)p.nsplit * T * M * p.Cp32;
    return p;
}

// ------------------------------------------------------------------ activation backward + bias gradient
// geff = gy * act'(y) (in place allowed);  partial[m][n*cpp + chunk] = sum over the chunk of geff.
// grid (cpp, C, B): one (image, channel) plane chunk per workgroup -> no per-element index arithmetic, float4 accesses
// when the plane size allows (HBM-bound: 2 reads + 1 write per element).

struct AB {      // up to MAXGRP same-shaped problems per launch: blockIdx.z = problem * zper + image
    const float* gy[MAXGRP]; const float* y[MAXGRP]; float* geff[MAXGRP]; float* partial[MAXGRP]; float* gbias_direct[MAXGRP];
    int zper;
};

template <bool VEC4>
__global__ __launch_bounds__(256) void k_act_bwd(AB t, int HW, long gy_bs, long y_bs, long ge_bs, int act, float act_a,
                                                 float act_b, int nb, int accum) {
    __shared__ float red[4];
    const int grp = (int)blockIdx.z / t.zper, zimg = (int)blockIdx.z - grp * t.zper;
    const float* __restrict__ gy = t.gy[grp];
    const float* __restrict__ y = t.y[grp];
    float* __restrict__ geff = t.geff[grp];
    float* __restrict__ partial = t.partial[grp];
    float* __restrict__ gbias_direct = t.gbias_direct[grp];
    const int m = blockIdx.y, cpp = gridDim.x;
    float s[1] = {0.f};
    // nb == 1: this workgroup owns image zimg; nb == B (small maps, zper == 1): it walks all images itself and
    // writes the channel's bias gradient directly (no second-stage launch)
    for (int nn = 0; nn < nb; nn++) {
        const int n = zimg + nn;
        const float* __restrict__ gp = gy + (long)n * gy_bs + (long)m * HW;
        const float* __restrict__ yp = (act != ACT_NONE) ? y + (long)n * y_bs + (long)m * HW : nullptr;
        float* __restrict__ ep = geff ? geff + (long)n * ge_bs + (long)m * HW : nullptr;
        if (VEC4) {
            // four iterations' loads (up to 8 x 16 bytes) in flight per work item; same element order as a one-by-one loop
            const int nq = HW >> 2, stp = cpp * 256;
            for (int q0 = blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += 4 * stp) {
                float4 gg[4], vv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * stp;
                    gg[u] = (q < nq) ? ((const float4*)gp)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                    vv[u] = (q < nq && act != ACT_NONE) ? ((const float4*)yp)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * stp;
                    if (q < nq) {
                        float4 g = gg[u];
                        if (act != ACT_NONE) {
                            g.x = act_grad(g.x, vv[u].x, act, act_a, act_b);
                            g.y = act_grad(g.y, vv[u].y, act, act_a, act_b);
                            g.z = act_grad(g.z, vv[u].z, act, act_a, act_b);
                            g.w = act_grad(g.w, vv[u].w, act, act_a, act_b);
                        }
                        if (ep) ((float4*)ep)[q] = g;
                        s[0] += (g.x + g.y) + (g.z + g.w);
                    }
                }
            }
        } else {
            for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += cpp * 256) {
                float g = gp[e];
                if (act != ACT_NONE) g = act_grad(g, yp[e], act, act_a, act_b);
                if (ep) ep[e] = g;
                s[0] += g;
            }
        }
    }
    cc::block_sum_256<1>(s, red);
    if (threadIdx.x == 0) {
        if (gbias_direct) gbias_direct[m] = accum ? (gbias_direct[m] + s[0]) : s[0];
        else if (partial) partial[(long)m * (cpp * t.zper) + zimg * cpp + blockIdx.x] = s[0];
    }
}

struct BR { const float* partial[MAXGRP]; float* gbias[MAXGRP]; };

__global__ __launch_bounds__(64) void k_bias_reduce(BR t, int nchunk, int accum) {
    const float* __restrict__ partial = t.partial[blockIdx.y];
    float* __restrict__ gbias = t.gbias[blockIdx.y];
    const int m = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < nchunk; k += 64) s += partial[(long)m * nchunk + k];
    s = cc::wave_sum(s);
    if (threadIdx.x == 0) gbias[m] = accum ? (gbias[m] + s) : s;
}

inline int pick_bm(int M) { return M > 64 ? 128 : (M > 32 ? 64 : 32); }

inline void launch_gg_flat(const GG& g, hipStream_t s) {
    const long Ntot = (long)g.B * g.OHt * g.OWt;
    const int bm = pick_bm(g.M);
    dim3 grid((unsigned)((Ntot + BN - 1) / BN), (unsigned)((g.M + bm - 1) / bm));
    if (bm == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_gemm<128>), grid, dim3(256), 0, s, g);
    else if (bm == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_gemm<64>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_gemm<32>), grid, dim3(256), 0, s, g);
}

template <class K, class ARGS>
inline void launch_patch_kernel(K kern, bool& big_lds_enabled, const ARGS& c, dim3 grid, size_t smem, hipStream_t s) {
    if (smem > 64 * 1024 && !big_lds_enabled) {            // > 64 KB of dynamic LDS has to be requested once per kernel
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        big_lds_enabled = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, c);
}

template <int BM, int CK, int TPS, int SPLIT>
inline void launch_patch1(const CP& c, dim3 grid, size_t smem, hipStream_t s) {
    static bool big = false;
    launch_patch_kernel(&k_conv_patch<BM, CK, TPS, SPLIT>, big, c, grid, smem, s);
}
template <int BM, int TPS, int SPLIT>
inline void launch_patch1_stk(const CP& c, dim3 grid, size_t smem, hipStream_t s) {
    static bool big = false;
    launch_patch_kernel(&k_conv_patch_stk<BM, TPS, SPLIT>, big, c, grid, smem, s);
}

inline bool stacked(const CP& c) { return c.ipt > 1; }
inline bool stacked(const CPM& a) {
    for (int k = 0; k < a.n; k++)
        if (a.c[k].ipt > 1) return true;
    return false;
}

template <int BM, int CK, int TPS>
inline void launch_patch(const CP& c, dim3 grid, size_t smem, hipStream_t s) {
    if constexpr (CK == 8) {
        if (stacked(c)) {
            if (c.nsplit > 1) launch_patch1_stk<BM, TPS, 1>(c, grid, smem, s);
            else launch_patch1_stk<BM, TPS, 0>(c, grid, smem, s);
            return;
        }
    }
    if (c.nsplit > 1) launch_patch1<BM, CK, TPS, 1>(c, grid, smem, s);
    else launch_patch1<BM, CK, TPS, 0>(c, grid, smem, s);
}

template <int BM, int CK, int TPS>
inline void launch_patch(const CPM& c, dim3 grid, size_t smem, hipStream_t s) {
    if constexpr (CK == 8) {
        if (stacked(c)) {
            static bool big = false;
            launch_patch_kernel(&k_conv_patch_multi_stk<BM, TPS>, big, c, grid, smem, s);
            return;
        }
    }
    static bool big = false;
    launch_patch_kernel(&k_conv_patch_multi<BM, CK, TPS>, big, c, grid, smem, s);
}

template <class ARGS>
inline void dispatch_patch(int bm, int ck, int tps, const ARGS& c, dim3 grid, size_t smem, hipStream_t s) {
    if (tps == 3) {
        if (ck == 16) {
            if (bm == 128) launch_patch<128, 16, 3>(c, grid, smem, s);
            else if (bm == 64) launch_patch<64, 16, 3>(c, grid, smem, s);
            else if (bm == 32) launch_patch<32, 16, 3>(c, grid, smem, s);
            else launch_patch<16, 16, 3>(c, grid, smem, s);
        } else {
            if (bm == 128) launch_patch<128, 8, 3>(c, grid, smem, s);
            else if (bm == 64) launch_patch<64, 8, 3>(c, grid, smem, s);
            else if (bm == 32) launch_patch<32, 8, 3>(c, grid, smem, s);
            else launch_patch<16, 8, 3>(c, grid, smem, s);
        }
    } else if (ck == 16) {
        if (bm == 128) launch_patch<128, 16, 1>(c, grid, smem, s);
        else if (bm == 64) launch_patch<64, 16, 1>(c, grid, smem, s);
        else if (bm == 32) launch_patch<32, 16, 1>(c, grid, smem, s);
        else launch_patch<16, 16, 1>(c, grid, smem, s);
    } else {
        if (bm == 128) launch_patch<128, 8, 1>(c, grid, smem, s);
        else if (bm == 64) launch_patch<64, 8, 1>(c, grid, smem, s);
        else if (bm == 32) launch_patch<32, 8, 1>(c, grid, smem, s);
        else launch_patch<16, 8, 1>(c, grid, smem, s);
    }
}

inline CP make_cp(const GG& g, const ConvPlan& p, const float* zeros, const float* wp, float* part) {
    CP c = {};
    c.x = g.x; c.wp = wp; c.zeros = zeros; c.bias = g.bias; c.res = g.res; c.y = g.y; c.part = part;
    c.B = g.B; c.Cin = g.Cin; c.IH = g.IH; c.IW = g.IW; c.x_bs = g.x_bs;
    c.M = g.M; c.Mpad = p.Mpad; c.Cpad = p.Cpad;
    c.Rt = g.Rt; c.St = g.St; c.si = g.si; c.dstep = g.dstep;
    c.dy_base = g.dy0 - p.ymin; c.dx_base = g.dx0 - p.xmin; c.ymin = p.ymin; c.xmin = p.xmin;
    c.PH = p.PH; c.PWr = p.PWr; c.PS = p.PS; c.tw16 = p.tw16; c.ipt = p.ipt; c.phi = p.phi; c.aligned = p.aligned; c.shift = p.shift;
    c.OHt = g.OHt; c.OWt = g.OWt; c.so = g.so; c.oy0 = g.oy0; c.ox0 = g.ox0; c.OH = g.OH; c.OW = g.OW;
    c.y_bs = g.y_bs; c.res_bs = g.res_bs;
    c.tiles_x = p.tiles_x; c.tiles_y = p.tiles_y;
    c.nsplit = p.nsplit; c.cps = p.cps;
    c.part_stride = (long)g.B * g.M * (p.tiles_y * (p.tw16 ? 8 : TH)) * (p.tiles_x * (p.tw16 ? 16 : TW));      // padded slabs
    c.act = g.act; c.act_a = g.act_a; c.act_b = g.act_b; c.res_mul = g.res_mul;
    c.add = g.add; c.add_bs = g.add_bs;
#ifndef CC_CONV_WMAJOR
#define CC_CONV_WMAJOR 1
#endif
    c.wmajor = (CC_CONV_WMAJOR && (long)g.M * g.Rt * g.St > (long)g.B * g.IH * g.IW) ? 1 : 0;      // weights M Cin T floats, input B Cin IH IW
    c.vec4 = (g.so == 1) && ((g.OW & 3) == 0) && ((g.ox0 & 3) == 0) && ((((uintptr_t)g.y) & 15) == 0) && ((g.y_bs & 3) == 0) &&
             (!g.res || ((((uintptr_t)g.res) & 15) == 0 && (g.res_bs & 3) == 0)) &&
             (!g.add || ((((uintptr_t)g.add) & 15) == 0 && (g.add_bs & 3) == 0));
    return c;
}

// ---- Winograd path (wino.hip): nprob same-shaped problems of geometry g[0] in one launch
inline ccint::WinoGeom wino_geom(const GG& g) {
    ccint::WinoGeom w = {};
    w.B = g.B; w.Cin = g.Cin; w.H = g.IH; w.W = g.IW; w.x_bs = g.x_bs; w.M = g.M; w.y_bs = g.y_bs; w.res_bs = g.res_bs; w.add_bs = g.add_bs;
    w.act = g.act; w.act_a = g.act_a; w.act_b = g.act_b; w.res_mul = g.res_mul;
    return w;
}

inline void wino_scope_name(const GG& g, const ConvPlan& p, int nprob, char* nm, int cap) {
    int nl = p.wn.tile ? snprintf(nm, cap, "k_wino_f2x3_s<%d, %d>", p.wn.tile, p.nsplit > 1 ? 1 : 0)
                       : snprintf(nm, cap, "k_wino_f2x3<%d>", p.nsplit > 1 ? 1 : 0);
    if (cctools::env_flag("CC_TIMING_DETAIL"))
        snprintf(nm + nl, cap - nl, " %dx[B%d M%d C%d %dx%d%s t9 k%d] wg%d", nprob, g.B, g.M, g.Cin, g.OH, g.OW, p.wpad ? "(pad)" : "", p.nsplit,
                 nprob * p.wn.nqb * p.wn.nmb * p.nsplit);
}

// MFMA FLOPs the Winograd kernel executes: 16 multiply-adds per 2x2 output tile and channel pair (the direct form: 36)
inline double wino_gflop(const GG& g, const ConvPlan& p) { return 2e-9 * 16.0 * g.B * p.wn.TY * p.wn.TX * (double)g.M * g.Cin; }

inline int wino_flip(const GG& g) { return g.dstep < 0 ? 1 : 0; }

// ---- 3x3 / stride-1 layers on maps whose width is not a multiple of 4 (8x26, 4x13: the deep levels): the Winograd weight-gradient
// kernel moves rows as 16-byte pieces, so dY and the input are first copied into rows padded with zeros to a multiple of 4 (zero
// dY columns add nothing, zero input columns are the convolution's own padding) -- two 1-2 MB copies in one launch against half
// the time of the im2col kernel on these 512-channel layers (profiles/r04_ab_round4.txt).
struct PadJob { const float* src; float* dst; long bs; int rows_per_image; };        // rows of image n start at src + n * bs
struct PadTab { PadJob j[2 * MAXGRP]; int n, B, W, Wp; long row_end[2 * MAXGRP]; };     // row_end: cumulative B * rows_per_image
__global__ __launch_bounds__(256) void k_pad_rows(PadTab t) {
    const int q4 = t.Wp >> 2;                                  // float4s per padded row
    const long e = (long)blockIdx.x * 256 + threadIdx.x;       // one float4 of one padded row
    const long row = e / q4;
    const int c4 = (int)(e - row * q4) * 4;
    int k = 0;
    long first = 0;
#pragma unroll
    for (int q = 0; q < 2 * MAXGRP - 1; q++)
        if (q + 1 < t.n && row >= t.row_end[q]) { k = q + 1; first = t.row_end[q]; }
    if (row >= t.row_end[t.n - 1]) return;
    const PadJob& j = t.j[k];
    const long r = row - first;
    const int n = (int)(r / j.rows_per_image);
    const long rr = r - (long)n * j.rows_per_image;
    const float* s = j.src + (long)n * j.bs + rr * t.W;
    float4 v;
    v.x = c4 + 0 < t.W ? s[c4 + 0] : 0.f;
    v.y = c4 + 1 < t.W ? s[c4 + 1] : 0.f;
    v.z = c4 + 2 < t.W ? s[c4 + 2] : 0.f;
    v.w = c4 + 3 < t.W ? s[c4 + 3] : 0.f;
    *reinterpret_cast<float4*>(j.dst + (r * t.Wp + c4)) = v;
}

// Winograd over a zero-padded copy of the input (ConvPlan::wpad): the copy goes behind the problem's partial slabs; pr / wg are
// re-pointed at it (rows of wpad floats, dense [B][Cin][H][wpad])
inline void wino_pad_input(const GG& g, const ConvPlan& p, float* part, hipStream_t s, ccint::WinoProb& pr, ccint::WinoGeom& wg) {
    float* xpad = part + p.part_floats;
    PadTab t = {};
    t.B = g.B; t.W = g.IW; t.Wp = p.wpad; t.n = 1;
    t.j[0] = PadJob{g.x, xpad, g.x_bs, g.Cin * g.IH};
    t.row_end[0] = (long)g.B * g.Cin * g.IH;
    const long nf4 = t.row_end[0] * (p.wpad >> 2);
    hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((nf4 + 255) / 256)), dim3(256), 0, s, t);
    if (cctools::env_flag("CC_WINO_TRACE"))
        fprintf(stderr, "wino padded input: B%d M%d C%d %dx%d -> pitch %d, nsplit %d dstep %d\n", g.B, g.M, g.Cin, g.IH, g.IW, p.wpad, p.nsplit, g.dstep);
    pr.x = xpad;
    wg.W = p.wpad;
    wg.x_bs = (long)g.Cin * g.IH * p.wpad;
}

// ws: [64 zeros][repacked weights][split-K partial slabs]; sized by conv_ws_floats(plan_conv(g))
// prepacked (optional): {64 zeros, wp} produced earlier by k_repack_table -> no repack launch here
// A 3x3 / stride-1 / pad-1 problem with <= 4 channels on one side (a prediction head or its data-gradient, a layer with <= 4 inputs)
// as the conv_heads.hip description: 1 = few reduction channels (k_conv_thinc), 2 = few output channels (k_conv_thinm), 0 = neither
inline int head_kernel_of(const GG& g, ccint::HeadConv& h) {
    static const int off = cctools::env_int("CC_NO_HEAD_KERNELS", 0);       // tools: 1 = none, 2 = no thinm, 3 = no thinc
    if (off == 1 || (g.Cin > 4 && g.M > 4) || g.Cin < 1 || g.Rt != 3 || g.St != 3 || g.si != 1 || g.so != 1 || g.oy0 != 0 || g.ox0 != 0) return 0;
    if (g.OHt != g.OH || g.OWt != g.OW || g.OH != g.IH || g.OW != g.IW || g.w_ri != 3 || g.w_sj != 1) return 0;
    if (!((g.dstep == 1 && g.dy0 == -1 && g.dx0 == -1) || (g.dstep == -1 && g.dy0 == 1 && g.dx0 == 1))) return 0;
    if (g.res_mul && !g.res) return 0;
    h = ccint::HeadConv{g.x, g.w, g.bias, g.res, g.res_mul ? g.add : nullptr, g.y, g.B, g.Cin, g.IH, g.IW, g.M, g.x_bs, g.y_bs, g.res_bs,
                        g.add_bs, g.w_sm, g.w_sc, (long)g.w0, g.dstep, g.act, g.act_a, g.act_b, g.res_mul};
    if (g.Cin <= 4 && off != 3 && ccint::head_conv_thinc_vec(h) != 0) return 1;
    if (g.M <= 4 && off != 2 && ccint::head_conv_thinm_ok(h)) return 2;
    return 0;
}

inline void launch_gg(const GG& g, float* ws, hipStream_t s, const float* prepacked = nullptr, const float* pre_zeros = nullptr) {
    {
        ccint::HeadConv h;
        const int hk = head_kernel_of(g, h);
        if (hk) {
            char nm[96];
            int nl = snprintf(nm, sizeof nm, hk == 1 ? "k_conv_thinc<%d>" : "k_conv_thinm<%d>", hk == 1 ? g.Cin : g.M);
            if (cctools::env_flag("CC_TIMING_DETAIL"))
                snprintf(nm + nl, sizeof nm - nl, " B%d M%d C%d %dx%d t9", g.B, g.M, g.Cin, g.OHt, g.OWt);
            cctiming::Scope tsc(nm, 2e-9 * g.B * g.OHt * g.OWt * (double)g.M * g.Cin * 9, s);
            if (cctools::env_flag("CC_HEAD_TRACE"))
                fprintf(stderr, "head kernel %d: B%d M%d C%d %dx%d dstep %d\n", hk, g.B, g.M, g.Cin, g.OHt, g.OWt, g.dstep);
            if (hk == 1 ? ccint::head_conv_thinc_launch(h, s) : ccint::head_conv_thinm_launch(h, s)) return;
        }
    }
    const ConvPlan p = plan_conv(g);
    if (!p.use_patch || ws == nullptr) { launch_gg_flat(g, s); return; }
    const float* zeros = ws;
    const float* wp = ws + 64;
    float* part = ws + 64 + (prepacked ? 0 : p.wp_floats);
    const int T = g.Rt * g.St;
    if (p.wino) {
        if (!prepacked)
            ccint::wino_weights_launch(g.w, ws + 64, g.M, g.Cin, p.Cpad, p.Mpad, g.w_sm, g.w_sc, g.w0, g.w_ri, g.w_sj, wino_flip(g), s);
        ccint::WinoProb pr = {g.x, prepacked ? prepacked : wp, g.bias, g.res, g.add, g.y, part};
        ccint::WinoGeom wg = wino_geom(g);
        if (p.wpad) wino_pad_input(g, p, part, s, pr, wg);
        bool ok;
        {
            char nm[128];
            wino_scope_name(g, p, 1, nm, sizeof nm);
            cctiming::Scope tsc(nm, wino_gflop(g, p), s);
            ok = ccint::wino_launch(wg, p.wn, &pr, 1, s);
        }
        if (!ok) { launch_gg_flat(g, s); return; }      // x not 16-byte aligned (an odd view): the gather kernel reads the weights as they lie
        if (p.nsplit > 1) {
            const long total = (long)g.B * g.M * g.OHt * g.OWt;
            hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)part,
                               p.nsplit, (long)g.B * g.M * p.Hp * p.Wp, g.bias, g.res, g.y, g.M, g.OHt, g.OWt, g.so, g.oy0, g.ox0, g.OH,
                               g.OW, g.y_bs, g.res_bs, total, g.act, g.act_a, g.act_b, g.res_mul, g.add, g.add_bs, p.Hp, p.Wp);
        }
        return;
    }
    if (prepacked) {
        zeros = pre_zeros;
        wp = prepacked;
    } else {
        hipLaunchKernelGGL(k_repack_w, dim3((unsigned)((p.wp_floats + 255) / 256)), dim3(256), 0, s, g.w, ws + 64, ws, g.M, g.Cin,
                           p.Mpad, p.Cpad, T, g.St, g.w_sm, g.w_sc, g.w0, g.w_ri, g.w_sj);
    }
    const CP c = make_cp(g, p, zeros, wp, part);
    dim3 grid((unsigned)conv_tiles(g, p), (unsigned)(p.Mpad / p.bm), (unsigned)p.nsplit);
    {
        char nm[96];
        int nl = p.ipt > 1 ? snprintf(nm, sizeof nm, "k_conv_patch_stk<%d, %d, %d>", p.bm, p.tps, p.nsplit > 1 ? 1 : 0)
                           : snprintf(nm, sizeof nm, "k_conv_patch<%d, %d, %d, %d>", p.bm, p.ck, p.tps, p.nsplit > 1 ? 1 : 0);
        if (cctools::env_flag("CC_TIMING_DETAIL"))
            snprintf(nm + nl, sizeof nm - nl, " B%d M%d C%d %dx%d t%d k%d wg%d", g.B, g.M, g.Cin, g.OHt, g.OWt, g.Rt * g.St, p.nsplit,
                     (int)(grid.x * grid.y * grid.z));
        cctiming::Scope tsc(nm, 2e-9 * g.B * g.OHt * g.OWt * (double)g.M * g.Cin * g.Rt * g.St, s);
        dispatch_patch(p.bm, p.ck, p.tps, c, grid, p.smem, s);
    }
    if (p.nsplit > 1) {
        const long total = (long)g.B * g.M * g.OHt * g.OWt;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)part,
                           p.nsplit, c.part_stride, g.bias, g.res, g.y, g.M, g.OHt, g.OWt, g.so, g.oy0, g.ox0, g.OH, g.OW,
                           g.y_bs, g.res_bs, total, g.act, g.act_a, g.act_b, g.res_mul, g.add, g.add_bs,
                           p.Hp, p.Wp);
    }
}

// n problems in ONE conv launch (+ ONE split-K epilogue launch): the parity classes of a stride-2 data-gradient, the
// same-shaped convolutions of parallel branches, and (round 3, cc_conv2d_list) independent layers of DIFFERENT networks --
// every class carries its own geometry, channel count and epilogue; what the classes of a launch share is the tile
// configuration (BM, CK) of the kernel instance.  Needs prepacked weight images.  zeros / wp / part: the 64-float zero block,
// weight image and partial-slab area of the class.
struct ClsIn { GG g; ConvPlan p; const float* zeros; const float* wp; float* part; };

inline size_t smem_cls(const ConvPlan& p, int tps) {
    const size_t b = (size_t)(2 * tps * p.ck * p.bm + 2 * p.ck * p.PS) * sizeof(float);
    return (p.bm >= 32 && b < 16384) ? 16384 : b;            // epilogue transpose: 4 KB per wave
}

// same geometry and epilogue form (the Winograd launch shares them between its problems)
inline bool same_problem_shape(const GG& a, const GG& b) {
    return a.B == b.B && a.Cin == b.Cin && a.IH == b.IH && a.IW == b.IW && a.x_bs == b.x_bs && a.M == b.M && a.y_bs == b.y_bs &&
           a.res_bs == b.res_bs && a.add_bs == b.add_bs && a.act == b.act && a.act_a == b.act_a && a.act_b == b.act_b &&
           a.res_mul == b.res_mul && a.dstep == b.dstep;
}

inline bool launch_classes(const ClsIn* cs, int n, hipStream_t s, bool idle_taps = false) {
    if (n < 1 || n > MAXCLS) return false;
    {   // Winograd problems: all of the launch or none (a mixed list goes back to the caller, which launches one by one)
        int nw = 0;
        for (int k = 0; k < n; k++) nw += cs[k].p.wino ? 1 : 0;

        if (nw) {
            if (nw != n) return false;
            ccint::WinoProb pr[MAXCLS];
            EPM e = {};
            e.n = n;
            int ebx = 0;
            const ConvPlan& p = cs[0].p;
            for (int k = 0; k < n; k++) {
                const GG& g = cs[k].g;
                if (!cs[k].wp || !same_problem_shape(g, cs[0].g) || cs[k].p.nsplit != p.nsplit || cs[k].p.cps != p.cps) {
                    if (cctools::env_flag("CC_WINO_TRACE"))
                        fprintf(stderr, "wino classes declined: k %d wp %p same %d nsplit %d/%d cps %d/%d\n", k, (const void*)cs[k].wp,
                                (int)same_problem_shape(g, cs[0].g), cs[k].p.nsplit, p.nsplit, cs[k].p.cps, p.cps);
                    return false;
                }
                pr[k] = ccint::WinoProb{g.x, cs[k].wp, g.bias, g.res, g.add, g.y, cs[k].part};
                EPC& c = e.c[k];
                c.part = cs[k].part; c.bias = g.bias; c.res = g.res; c.add = g.add; c.y = g.y;
                c.Hp = p.Hp; c.Wp = p.Wp;
                c.part_stride = (long)g.B * g.M * c.Hp * c.Wp;
                c.nsplit = p.nsplit;
                c.OHt = g.OHt; c.OWt = g.OWt; c.oy0 = g.oy0; c.ox0 = g.ox0;
                c.total = p.nsplit > 1 ? (long)g.B * g.M * g.OHt * g.OWt : 0;
                c.M = g.M; c.so = g.so; c.OH = g.OH; c.OW = g.OW; c.y_bs = g.y_bs; c.res_bs = g.res_bs; c.add_bs = g.add_bs;
                c.act = g.act; c.act_a = g.act_a; c.act_b = g.act_b; c.res_mul = g.res_mul;
                ebx += (int)((c.total + 255) / 256);
                e.bx_end[k] = ebx;
            }
            ccint::WinoGeom wg = wino_geom(cs[0].g);
            if (p.wpad) {
                // zero-padded input copies of all n problems (same shape: same plan), 2 * MAXGRP jobs per copy launch
                for (int k = 0; k < n; k++)
                    if (cs[k].p.wpad != p.wpad || cs[k].p.part_floats != p.part_floats || !cs[k].part) return false;
                for (int k0 = 0; k0 < n; k0 += 2 * MAXGRP) {
                    PadTab t = {};
                    t.B = cs[0].g.B; t.W = cs[0].g.IW; t.Wp = p.wpad;
                    long rows = 0;
                    for (int k = k0; k < n && k < k0 + 2 * MAXGRP; k++) {
                        const GG& g = cs[k].g;
                        float* xpad = cs[k].part + p.part_floats;
                        t.j[t.n] = PadJob{g.x, xpad, g.x_bs, g.Cin * g.IH};
                        rows += (long)g.B * g.Cin * g.IH;
                        t.row_end[t.n] = rows;
                        t.n++;
                        pr[k].x = xpad;
                    }
                    const long nf4 = rows * (p.wpad >> 2);
                    hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((nf4 + 255) / 256)), dim3(256), 0, s, t);
                }
                wg.W = p.wpad;
                wg.x_bs = (long)cs[0].g.Cin * cs[0].g.IH * p.wpad;
                if (cctools::env_flag("CC_WINO_TRACE"))
                    fprintf(stderr, "wino padded input, %d problems: B%d M%d C%d %dx%d -> pitch %d, nsplit %d\n", n, cs[0].g.B, cs[0].g.M,
                            cs[0].g.Cin, cs[0].g.IH, cs[0].g.IW, p.wpad, p.nsplit);
            }
            {
                char nm[128];
                wino_scope_name(cs[0].g, p, n, nm, sizeof nm);
                cctiming::Scope tsc(nm, n * wino_gflop(cs[0].g, p), s);
                if (!ccint::wino_launch(wg, p.wn, pr, n, s)) return false;
            }
            if (p.nsplit > 1) hipLaunchKernelGGL(k_splitk_epilogue_multi, dim3((unsigned)ebx), dim3(256), 0, s, e);
            return true;
        }
    }
    size_t smem = 0;
    int tps = 3, maxsplit = 1, maxy = 1, ref = -1;
    if (cctools::env_int("CC_CONV_IDLE_TAPS", 1)) idle_taps = true;
    for (int k = 0; k < n; k++) {
        const ConvPlan& p = cs[k].p;
        // a class no tap reaches (the odd output parities of a 1x1 stride-2 data-gradient: DispResNet6's shortcut convolutions):
        // its result is the epilogue of zero -- it takes no workgroup of the conv launch, only a slot of the epilogue launch
        if (cs[k].g.Cin == 0) continue;
        if (ref < 0) ref = k;
        if (!p.use_patch || p.bm != cs[ref].p.bm || p.ck != cs[ref].p.ck || !cs[k].wp) return false;
        // three taps per stage launch-wide unless a class's patch does not leave room (a class with fewer taps idles the slots)
        if (p.tps != 3 && !(idle_taps && cs[k].g.Rt * cs[k].g.St < 3 && smem_cls(p, 3) <= 80 * 1024)) tps = 1;
        if (p.nsplit > maxsplit) maxsplit = p.nsplit;
        if (p.Mpad / p.bm > maxy) maxy = p.Mpad / p.bm;
    }
    CPM a = {};        // ~3 KB + ~1.5 KB of host stack, passed to the launches by value
    EPM e = {};
    e.n = n;
    int bx = 0, ebx = 0, epi_any = 0, nc = 0;
    double gf = 0;
    for (int k = 0; k < n; k++) {
        const GG& g = cs[k].g;
        const ConvPlan& p = cs[k].p;
        const bool empty = g.Cin == 0;
        EPC& c = e.c[k];
        c.part = empty ? nullptr : cs[k].part; c.bias = g.bias; c.res = g.res; c.add = g.add; c.y = g.y;
        c.Hp = empty ? g.OHt : p.Hp;
        c.Wp = empty ? g.OWt : p.Wp;
        c.part_stride = (long)g.B * g.M * c.Hp * c.Wp;
        c.nsplit = empty ? 0 : p.nsplit;
        c.OHt = g.OHt; c.OWt = g.OWt; c.oy0 = g.oy0; c.ox0 = g.ox0;
        c.total = (empty || p.nsplit > 1) ? (long)g.B * g.M * g.OHt * g.OWt : 0;
        c.M = g.M; c.so = g.so; c.OH = g.OH; c.OW = g.OW; c.y_bs = g.y_bs; c.res_bs = g.res_bs; c.add_bs = g.add_bs;
        c.act = g.act; c.act_a = g.act_a; c.act_b = g.act_b; c.res_mul = g.res_mul;
        ebx += (int)((c.total + 255) / 256);
        e.bx_end[k] = ebx;
        if (c.total) epi_any = 1;
        if (empty) continue;
        const size_t sm = smem_cls(p, tps);          // A buffers follow the launch-wide TPS, the patch buffers this class's PS
        if (sm > smem) smem = sm;
        a.c[nc] = make_cp(g, p, cs[k].zeros, cs[k].wp, cs[k].part);
        bx += (int)conv_tiles(g, p);
        a.bx_end[nc] = bx;
        nc++;
        gf += 2e-9 * g.B * g.OHt * g.OWt * (double)g.M * g.Cin * g.Rt * g.St;
    }
    a.n = nc;
    {   // launch-wide XCD order: weight-major when the classes' weights outweigh their inputs
        double wf = 0, xf = 0;
        for (int k = 0; k < nc; k++) {
            const CP& c = a.c[k];
            wf += (double)c.M * c.Cin * c.Rt * c.St;
            xf += (double)c.B * c.Cin * c.IH * c.IW;
        }
        a.wmajor = (CC_CONV_WMAJOR && wf > xf) ? 1 : 0;
    }
    if (smem > 80 * 1024) return false;
    if (nc > 0) {
        dim3 grid((unsigned)bx, (unsigned)maxy, (unsigned)maxsplit);
        char nm[224];
        int nl = stacked(a) ? snprintf(nm, sizeof nm, "k_conv_patch_multi_stk<%d, %d>", cs[ref].p.bm, tps)
                            : snprintf(nm, sizeof nm, "k_conv_patch_multi<%d, %d, %d>", cs[ref].p.bm, cs[ref].p.ck, tps);
        if (cctools::env_flag("CC_TIMING_DETAIL")) {
            // classes with the same geometry are counted, not repeated
            for (int k = 0; k < n && nl < (int)sizeof nm - 48; k++) {
                const GG& g = cs[k].g;
                if (g.Cin == 0) continue;
                int same = 0, first = 1;
                for (int j = 0; j < n; j++) {
                    const GG& h = cs[j].g;
                    const bool eq = h.Cin == g.Cin && h.M == g.M && h.OHt == g.OHt && h.OWt == g.OWt && h.Rt * h.St == g.Rt * g.St &&
                                    cs[j].p.nsplit == cs[k].p.nsplit;
                    if (eq) { same++; if (j < k) first = 0; }
                }
                if (!first) continue;
                nl += snprintf(nm + nl, sizeof nm - nl, " %dx[B%d M%d C%d %dx%d t%d k%d]", same, g.B, g.M, g.Cin, g.OHt, g.OWt,
                               g.Rt * g.St, cs[k].p.nsplit);
            }
            snprintf(nm + nl, sizeof nm - nl, " wg%d", (int)(grid.x * grid.y * grid.z));
        }
        cctiming::Scope tsc(nm, gf, s);
        dispatch_patch(cs[ref].p.bm, cs[ref].p.ck, tps, a, grid, smem, s);
    }
    if (epi_any) hipLaunchKernelGGL(k_splitk_epilogue_multi, dim3((unsigned)ebx), dim3(256), 0, s, e);
    return true;
}

// the G (x parity classes) same-shaped problems of the *_group entry points: split-K planned for the whole launch (mult)
inline bool launch_gg_classes(const GG* gs, int n, int mult, const float* const* zeros, const float* const* wps,
                              float* const* parts, hipStream_t s) {
    if (n < 2 || n > MAXCLS || dbg_flag_early("CC_NO_CLASS_MERGE")) return false;
    ClsIn cs[MAXCLS];
    for (int k = 0; k < n; k++) {
        cs[k].g = gs[k];
        cs[k].p = plan_conv(gs[k], mult);
        cs[k].zeros = zeros[k]; cs[k].wp = wps[k]; cs[k].part = parts[k];
    }
    return launch_classes(cs, n, s);
}

}  // namespace

extern "C" {

static GG make_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int Cin, int IH,
                   int IW, long x_bs, int Cout, int R, int S, int stride, int pad, int OH, int OW, long y_bs, long res_bs,
                   int act, float act_a, float act_b) {
    GG g = {};
    g.x = x; g.w = w; g.bias = bias; g.res = res; g.y = y;
    g.B = B; g.Cin = Cin; g.IH = IH; g.IW = IW; g.x_bs = x_bs;
    g.M = Cout; g.w_sm = (long)Cin * R * S; g.w_sc = (long)R * S; g.w0 = 0; g.w_ri = S; g.w_sj = 1;
    g.Rt = R; g.St = S; g.dy0 = -pad; g.dx0 = -pad; g.dstep = 1; g.si = stride;
    g.OHt = OH; g.OWt = OW; g.so = 1; g.oy0 = 0; g.ox0 = 0; g.OH = OH; g.OW = OW; g.y_bs = y_bs; g.res_bs = res_bs;
    g.act = act; g.act_a = act_a; g.act_b = act_b;
    return g;
}

static void fill_desc(const GG& g, const ConvPlan& p, long src, long dst, long* d) {
    if (p.wino) {        // U = G g G^T (wino_weights.h); d[6] marks the descriptor, the sign of d[7] the tap direction
        d[0] = src; d[1] = dst; d[2] = g.M; d[3] = g.Cin; d[4] = p.Mpad; d[5] = p.Cpad; d[6] = ccwino::WINO_T; d[7] = wino_flip(g) ? -3 : 3;
        d[8] = g.w_sm; d[9] = g.w_sc; d[10] = g.w0; d[11] = g.w_ri; d[12] = g.w_sj; d[13] = (long)p.wp_floats; d[14] = 0;
        d[15] = ccwino::wino_weight_blocks(p.Mpad, p.Cpad);
        return;
    }
    d[0] = src; d[1] = dst; d[2] = g.M; d[3] = g.Cin; d[4] = p.Mpad; d[5] = p.Cpad; d[6] = (long)g.Rt * g.St; d[7] = g.St;
    d[8] = g.w_sm; d[9] = g.w_sc; d[10] = g.w0; d[11] = g.w_ri; d[12] = g.w_sj; d[13] = (long)p.wp_floats; d[14] = 0;
    d[15] = repack_blocks(p.Mpad, p.Cpad, g.Rt * g.St);
}

/* Per-step weight prepack (optional fast path).  *_pack_desc fill 16-long descriptors ({src, dst, ...}; dst = where the
 * [tap][c][m] image of this layer goes: pack_base + 64 floats (+ the images of earlier parity classes for dgrad)) and
 * return the number of descriptors (0: this geometry does not use the patch kernel); cc_repack_table runs all of them in
 * one launch after the caller has filled d[14] = first block of each descriptor (cumulative sum of d[15] = its block count).
 * Buffers passed as `prepacked` to the conv entry points must start with 64 zero floats. */
size_t cc_conv2d_fwd_pack_floats(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW) {
    GG g = make_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, B, Cin, IH, IW, 0, Cout, R, S, stride, pad, OH, OW, 0, 0, 0,
                    1.f, 0.f);
    const ConvPlan p = plan_conv(g);
    return (p.use_patch && R * S <= 136) ? 64 + p.wp_floats : 0;      // 136 = tile rows of k_repack_table
}

int cc_conv2d_fwd_pack_desc(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW,
                            long src_ptr, long pack_base_ptr, long* desc_out_host) {
    GG g = make_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, B, Cin, IH, IW, 0, Cout, R, S, stride, pad, OH, OW, 0, 0, 0,
                    1.f, 0.f);
    const ConvPlan p = plan_conv(g);
    if (!p.use_patch) return 0;
    fill_desc(g, p, src_ptr, pack_base_ptr + 64 * (long)sizeof(float), desc_out_host);
    return 1;
}

size_t cc_conv2d_fwd_ws_bytes(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW) {
    GG g = make_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, B, Cin, IH, IW, 0, Cout, R, S, stride, pad, OH, OW, 0, 0, 0,
                    1.f, 0.f);
    return conv_ws_floats(plan_conv(g)) * sizeof(float);
}

/* y = act(conv2d(x, w, stride, pad) + bias + res).  x: [B,Cin,IH,IW] (batch stride x_bs), w: [Cout,Cin,R,S],
 * y: [B,Cout,OH,OW] (batch stride y_bs; may be a channel slice of a wider tensor).  ws: cc_conv2d_fwd_ws_bytes(). */
int cc_conv2d_fwd(const float* x, const float* w, const float* bias_or_null, const float* res_or_null, float* y, float* ws,
                  const float* prepacked_or_null, int B, int Cin, int IH, int IW, long x_bs, int Cout, int R, int S, int stride,
                  int pad, int OH, int OW, long y_bs, long res_bs, int act, float act_a, float act_b, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0) return CC_ERR_ARG;
    GG g = make_fwd(x, w, bias_or_null, res_or_null, y, B, Cin, IH, IW, x_bs, Cout, R, S, stride, pad, OH, OW, y_bs, res_bs,
                    act, act_a, act_b);
    launch_gg(g, ws, (hipStream_t)stream, prepacked_or_null ? prepacked_or_null + 64 : nullptr, prepacked_or_null);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* G same-shaped convolutions (parallel branches of a network) in one launch.  x / w / bias / res / y / prepacked: HOST arrays
 * of G device addresses (0 = null); ws: G consecutive areas of cc_conv2d_fwd_group_ws_bytes() / G bytes each. */
size_t cc_conv2d_fwd_group_ws_bytes(int G, int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH,
                                    int OW) {
    GG g = make_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, B, Cin, IH, IW, 0, Cout, R, S, stride, pad, OH, OW, 0, 0, 0,
                    1.f, 0.f);
    const size_t a = conv_ws_floats(plan_conv(g, G)), b = conv_ws_floats(plan_conv(g));
    return (size_t)G * (a > b ? a : b) * sizeof(float);
}

int cc_conv2d_fwd_group(int G, const long* x, const long* w, const long* bias, const long* res, const long* y, float* ws,
                        const long* prepacked, int B, int Cin, int IH, int IW, long x_bs, int Cout, int R, int S, int stride,
                        int pad, int OH, int OW, long y_bs, long res_bs, int act, float act_a, float act_b, void* stream) {
    if (G <= 0 || G > MAXCLS || B <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t stride_f = cc_conv2d_fwd_group_ws_bytes(G, B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW) / sizeof(float) / G;
    GG gs[MAXCLS];
    const float *zeros[MAXCLS], *wps[MAXCLS];
    float* parts[MAXCLS];
    bool packed = true;
    for (int k = 0; k < G; k++) {
        gs[k] = make_fwd((const float*)x[k], (const float*)w[k], bias ? (const float*)bias[k] : nullptr,
                         res ? (const float*)res[k] : nullptr, (float*)y[k], B, Cin, IH, IW, x_bs, Cout, R, S, stride, pad, OH,
                         OW, y_bs, res_bs, act, act_a, act_b);
        const float* pk = prepacked ? (const float*)prepacked[k] : nullptr;
        if (!pk) packed = false;
        zeros[k] = pk;
        wps[k] = pk ? pk + 64 : nullptr;
        parts[k] = ws + k * stride_f + 64;
    }
    if (!(G > 1 && packed && ws && launch_gg_classes(gs, G, G, zeros, wps, parts, s))) {
        for (int k = 0; k < G; k++) launch_gg(gs[k], ws ? ws + k * stride_f : nullptr, s, wps[k], zeros[k]);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* Transposed-convolution arithmetic  gx[n, c, iy, ix] = sum_{k,r,s} wT(k,c,r,s) * gy[n, k, oy, ox],  iy = oy*stride - pad + r.
 * Used for (a) the data-gradient of conv2d (w: [K,C,R,S] -> w_k_stride = C*R*S, w_c_stride = R*S) and
 * (b) ConvTranspose2d forward (w: [Cin=K, Cout=C, R, S] -> w_k_stride = C*R*S, w_c_stride = R*S as well),
 * one launch per output parity class so that no structurally-zero tap is multiplied.
 * gy: [B,K,OH,OW]; gx: [B,C,IH,IW]. */
static bool make_dgrad_class(GG& g, int py, int px, const float* gy, const float* w, const float* bias, float* gx, int B,
                             int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride, int pad, int IH, int IW,
                             long gx_bs, long w_k_stride, long w_c_stride, int act, float act_a, float act_b,
                             const float* mul = nullptr, long mul_bs = 0) {
    // taps r with (py + pad - r) % stride == 0, r ascending: r = r0 + stride*i
    const int r0 = (py + pad) % stride, s0 = (px + pad) % stride;
    const int Rt = (r0 < R) ? (R - r0 + stride - 1) / stride : 0;
    const int St = (s0 < S) ? (S - s0 + stride - 1) / stride : 0;
    const int OHt = (IH - py + stride - 1) / stride, OWt = (IW - px + stride - 1) / stride;
    if (OHt <= 0 || OWt <= 0) return false;
    g = GG();
    g.x = gy; g.w = w; g.bias = bias; g.res = mul; g.res_bs = mul_bs; g.res_mul = mul ? 1 : 0; g.y = gx;
    g.B = B; g.Cin = K; g.IH = OH; g.IW = OW; g.x_bs = gy_bs;
    g.M = C; g.w_sm = w_c_stride; g.w_sc = w_k_stride;
    g.w0 = r0 * S + s0; g.w_ri = stride * S; g.w_sj = stride;
    g.Rt = Rt > 0 ? Rt : 1; g.St = St > 0 ? St : 1;
    // oy = (iy + pad - r)/stride with iy = py + stride*ty, r = r0 + stride*i  ->  oy = ty + (py + pad - r0)/stride - i
    g.dy0 = (py + pad - r0) / stride; g.dx0 = (px + pad - s0) / stride; g.dstep = -1; g.si = 1;
    if (Rt == 0 || St == 0) g.Cin = 0;   // no tap reaches this parity class: output = act(bias)
    g.OHt = OHt; g.OWt = OWt; g.so = stride; g.oy0 = py; g.ox0 = px; g.OH = IH; g.OW = IW; g.y_bs = gx_bs;
    g.act = act; g.act_a = act_a; g.act_b = act_b;
    return true;
}

size_t cc_conv2d_dgrad_group_ws_bytes(int G, int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW);
size_t cc_conv2d_dgrad_ws_bytes(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW) {
    return cc_conv2d_dgrad_group_ws_bytes(1, B, K, OH, OW, C, R, S, stride, pad, IH, IW);
}

size_t cc_conv2d_dgrad_pack_floats(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW,
                                   long w_k_stride, long w_c_stride) {
    size_t tot = 64;
    if (R * S > 136) return 0;
    for (int py = 0; py < stride; py++)
        for (int px = 0; px < stride; px++) {
            GG g;
            if (!make_dgrad_class(g, py, px, nullptr, nullptr, nullptr, nullptr, B, K, OH, OW, 0, C, R, S, stride, pad, IH, IW, 0,
                                  w_k_stride, w_c_stride, 0, 1.f, 0.f))
                continue;
            if (g.Cin == 0) continue;                    // no tap reaches this parity class (1x1 stride 2): epilogue only, no image
            const ConvPlan p = plan_conv(g);
            if (!p.use_patch) return 0;
            tot += p.wp_floats;
        }
    return tot;
}

int cc_conv2d_dgrad_pack_desc(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW,
                              long w_k_stride, long w_c_stride, long src_ptr, long pack_base_ptr, long* desc_out_host) {
    int n = 0;
    long off = 64;
    for (int py = 0; py < stride; py++)
        for (int px = 0; px < stride; px++) {
            GG g;
            if (!make_dgrad_class(g, py, px, nullptr, nullptr, nullptr, nullptr, B, K, OH, OW, 0, C, R, S, stride, pad, IH, IW, 0,
                                  w_k_stride, w_c_stride, 0, 1.f, 0.f))
                continue;
            if (g.Cin == 0) continue;
            const ConvPlan p = plan_conv(g);
            if (!p.use_patch) return 0;
            fill_desc(g, p, src_ptr, pack_base_ptr + off * (long)sizeof(float), desc_out_host + 16 * n);
            off += (long)p.wp_floats;
            n++;
        }
    return n;
}

int cc_repack_table(const long* table_dev, int ndesc, long total_blocks, void* stream) {
    if (ndesc <= 0 || total_blocks <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_repack_table, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, ndesc);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* G same-shaped problems of the transposed-convolution arithmetic in one launch (all parity classes of all problems).
 * mul (optional, per problem): tensor of gx's shape (batch stride mul_bs); gx = act_grad(sum (+ bias), mul) = the gradient
 * w.r.t. the pre-activation of the layer whose OUTPUT `mul` is (act / act_a / act_b then describe THAT activation);
 * without it gx = act(sum + bias) (ConvTranspose2d forward).  Host pointer arrays as in cc_conv2d_fwd_group. */
size_t cc_conv2d_dgrad_group_ws_bytes(int G, int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW) {
    size_t best = 0;
    for (int mult = 1; mult <= G; mult += (G > 1 ? G - 1 : 1)) {
        size_t wmax = 0, psum = 0;        // the merged launch keeps every class's partial slabs alive at once
        for (int py = 0; py < stride; py++)
            for (int px = 0; px < stride; px++) {
                GG g;
                if (!make_dgrad_class(g, py, px, nullptr, nullptr, nullptr, nullptr, B, K, OH, OW, 0, C, R, S, stride, pad, IH, IW,
                                      0, (long)C * R * S, (long)R * S, 0, 1.f, 0.f))
                    continue;
                const ConvPlan p = plan_conv(g, mult);
                if (p.wp_floats > wmax) wmax = p.wp_floats;
                psum += p.part_floats + p.pad_floats;
            }
        const size_t t = 64 + wmax + psum;
        if (t > best) best = t;
    }
    return (size_t)G * best * sizeof(float);
}

static int dgrad_group_impl(int G, const long* gy, const long* w, const long* bias, const long* gx, const long* mul,
                            const long* add, float* ws, const long* prepacked, int B, int K, int OH, int OW, long gy_bs, int C,
                            int R, int S, int stride, int pad, int IH, int IW, long gx_bs, long mul_bs, long add_bs,
                            long w_k_stride, long w_c_stride, int act, float act_a, float act_b, void* stream) {
    if (G <= 0 || G > MAXCLS || B <= 0 || K <= 0 || C <= 0 || stride <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t stride_f = cc_conv2d_dgrad_group_ws_bytes(G, B, K, OH, OW, C, R, S, stride, pad, IH, IW) / sizeof(float) / G;
    bool packed = true;
    for (int k = 0; k < G; k++)
        if (!prepacked || !prepacked[k]) packed = false;
    const int ncls = stride * stride;
    if (packed && ws && G * ncls >= 2 && G * ncls <= MAXCLS) {
        GG gs[MAXCLS];
        const float *zeros[MAXCLS], *wps[MAXCLS];
        float* parts[MAXCLS];
        int n = 0;
        bool all = true;
        for (int k = 0; k < G && all; k++) {
            const float* pk = (const float*)prepacked[k];
            long off = 64;
            float* part = ws + k * stride_f + 64;
            for (int py = 0; py < stride && all; py++)
                for (int px = 0; px < stride; px++) {
                    if (!make_dgrad_class(gs[n], py, px, (const float*)gy[k], (const float*)w[k],
                                          bias ? (const float*)bias[k] : nullptr, (float*)gx[k], B, K, OH, OW, gy_bs, C, R, S,
                                          stride, pad, IH, IW, gx_bs, w_k_stride, w_c_stride, act, act_a, act_b,
                                          mul ? (const float*)mul[k] : nullptr, mul_bs)) { all = false; break; }
                    if (add && add[k]) {
                        if (mul && mul[k]) { gs[n].add = (const float*)add[k]; gs[n].add_bs = add_bs; }
                        else { gs[n].res = (const float*)add[k]; gs[n].res_bs = add_bs; gs[n].res_mul = 0; }      // gx = act(sum + add)
                    }
                    const ConvPlan p = plan_conv(gs[n], G);
                    zeros[n] = pk;
                    wps[n] = pk + off;
                    parts[n] = part;
                    off += (long)p.wp_floats;
                    part += p.part_floats;
                    n++;
                }
        }
        if (all && launch_gg_classes(gs, n, G, zeros, wps, parts, s)) {
            CC_CHECK_LAUNCH();
            return CC_OK;
        }
    }
    for (int k = 0; k < G; k++) {
        const float* pk = prepacked ? (const float*)prepacked[k] : nullptr;
        float* wk = ws ? ws + k * stride_f : nullptr;
        long off = 64;
        for (int py = 0; py < stride; py++) {
            for (int px = 0; px < stride; px++) {
                GG g;
                if (!make_dgrad_class(g, py, px, (const float*)gy[k], (const float*)w[k], bias ? (const float*)bias[k] : nullptr,
                                      (float*)gx[k], B, K, OH, OW, gy_bs, C, R, S, stride, pad, IH, IW, gx_bs, w_k_stride,
                                      w_c_stride, act, act_a, act_b, mul ? (const float*)mul[k] : nullptr, mul_bs))
                    continue;
                if (add && add[k]) {
                    if (mul && mul[k]) { g.add = (const float*)add[k]; g.add_bs = add_bs; }
                    else { g.res = (const float*)add[k]; g.res_bs = add_bs; g.res_mul = 0; }
                }
                if (g.Cin == 0) {                       // result = epilogue of zero
                    const long total = (long)g.B * g.M * g.OHt * g.OWt;
                    hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)nullptr,
                                       0, total, g.bias, g.res, g.y, g.M, g.OHt, g.OWt, g.so, g.oy0, g.ox0, g.OH, g.OW, g.y_bs, g.res_bs,
                                       total, g.act, g.act_a, g.act_b, g.res_mul, g.add, g.add_bs, g.OHt, g.OWt);
                } else if (pk) {
                    const ConvPlan p = plan_conv(g);
                    launch_gg(g, wk, s, pk + off, pk);
                    off += (long)p.wp_floats;
                } else {
                    launch_gg(g, wk, s);
                }
            }
        }
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_conv2d_dgrad_group(int G, const long* gy, const long* w, const long* bias, const long* gx, const long* mul, float* ws,
                          const long* prepacked, int B, int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride,
                          int pad, int IH, int IW, long gx_bs, long mul_bs, long w_k_stride, long w_c_stride, int act, float act_a,
                          float act_b, void* stream) {
    return dgrad_group_impl(G, gy, w, bias, gx, mul, nullptr, ws, prepacked, B, K, OH, OW, gy_bs, C, R, S, stride, pad, IH, IW, gx_bs,
                            mul_bs, 0, w_k_stride, w_c_stride, act, act_a, act_b, stream);
}

/* ... with `add`: gx = (sum + add) * act'(mul), or act(sum + add) without mul (act 0: plain accumulation) -- the other gradient contributions of a fan-out tensor (a residual
 * shortcut's gradient, what earlier data-gradients left in gx: add may alias gx) are summed in the epilogue. */
int cc_conv2d_dgrad_group_add(int G, const long* gy, const long* w, const long* gx, const long* mul, const long* add, float* ws,
                              const long* prepacked, int B, int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride,
                              int pad, int IH, int IW, long gx_bs, long mul_bs, long add_bs, long w_k_stride, long w_c_stride,
                              int act, float act_a, float act_b, void* stream) {
    return dgrad_group_impl(G, gy, w, nullptr, gx, mul, add, ws, prepacked, B, K, OH, OW, gy_bs, C, R, S, stride, pad, IH, IW, gx_bs,
                            mul_bs, add_bs, w_k_stride, w_c_stride, act, act_a, act_b, stream);
}

int cc_conv2d_dgrad(const float* gy, const float* w, const float* bias_or_null, float* gx, float* ws,
                    const float* prepacked_or_null, int B, int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride,
                    int pad, int IH, int IW, long gx_bs, long w_k_stride, long w_c_stride, int act, float act_a, float act_b,
                    void* stream) {
    const long gyp = (long)gy, wp = (long)w, bp = (long)bias_or_null, gxp = (long)gx, pk = (long)prepacked_or_null;
    return cc_conv2d_dgrad_group(1, &gyp, &wp, &bp, &gxp, nullptr, ws, &pk, B, K, OH, OW, gy_bs, C, R, S, stride, pad, IH, IW, gx_bs,
                                 0, w_k_stride, w_c_stride, act, act_a, act_b, stream);
}

/* ---- heterogeneous launch lists (round 3) ------------------------------------------------------------------------------
 * n independent convolution problems -- layers of different networks, or a layer's forward next to another layer's
 * data-gradient -- in as few launches as their tile configurations allow: problems whose kernel instance (BM, CK) agrees
 * share ONE k_conv_patch_multi launch (+ one split-K epilogue launch), <= 12 classes per launch; split-K is planned for
 * the launch as a whole (the chip is filled by all of its problems together).  desc_host: n records of CC_CL_LONGS longs:
 *   0 kind (0: conv2d forward arithmetic, 1: transposed arithmetic = data-gradient / ConvTranspose2d forward)
 *   1 x (kind 1: gy)  2 w  3 bias  4 res (kind 0: added before act; kind 1: `mul`, see cc_conv2d_dgrad_group)  5 y (gx)
 *   6 prepacked weight image (required)  7 add (kind 1 with mul: (sum + add) * act'(mul); may alias y)
 *   8 B  9 Cin (K)  10 IH (OH)  11 IW (OW)  12 x_bs  13 Cout (C)  14 R  15 S  16 stride  17 pad  18 OH (IH)  19 OW (IW)
 *   20 y_bs  21 res_bs  22 add_bs  23 act  24 act_a (float bits)  25 act_b (float bits)  26 w_k_stride  27 w_c_stride
 * ws: cc_conv2d_list_ws_bytes() bytes (partial slabs of every class). */
constexpr int CL_LONGS = 32;
struct ListCls { ClsIn c; int prob; };

static float bits_to_float(long v) { unsigned u = (unsigned)v; float f; memcpy(&f, &u, 4); return f; }

// -> classes of all problems in order (plans: no split yet), or -1 on a malformed record
static int list_classes(int n, const long* d, std::vector<ListCls>& out) {
    for (int i = 0; i < n; i++) {
        const long* r = d + (long)i * CL_LONGS;
        const float act_a = bits_to_float(r[24]), act_b = bits_to_float(r[25]);
        const int B = (int)r[8], Ci = (int)r[9], H0 = (int)r[10], W0 = (int)r[11], Co = (int)r[13], R = (int)r[14], S = (int)r[15];
        const int stride = (int)r[16], pad = (int)r[17], H1 = (int)r[18], W1 = (int)r[19];
        if (B <= 0 || Ci <= 0 || Co <= 0 || R <= 0 || S <= 0 || stride <= 0 || !r[6]) return -1;
        const float* pk = (const float*)r[6];
        if (r[0] == 0) {
            ListCls lc = {};
            lc.c.g = make_fwd((const float*)r[1], (const float*)r[2], (const float*)r[3], (const float*)r[4], (float*)r[5], B, Ci, H0, W0,
                              r[12], Co, R, S, stride, pad, H1, W1, r[20], r[21], (int)r[23], act_a, act_b);
            lc.c.p = plan_conv(lc.c.g);
            lc.c.zeros = pk; lc.c.wp = pk + 64; lc.prob = i;
            out.push_back(lc);
        } else {
            long off = 64;
            for (int py = 0; py < stride; py++)
                for (int px = 0; px < stride; px++) {
                    ListCls lc = {};
                    if (!make_dgrad_class(lc.c.g, py, px, (const float*)r[1], (const float*)r[2], (const float*)r[3], (float*)r[5], B, Ci, H0,
                                          W0, r[12], Co, R, S, stride, pad, H1, W1, r[20], r[26], r[27], (int)r[23], act_a, act_b,
                                          (const float*)r[4], r[21]))
                        continue;
                    lc.c.g.add = (const float*)r[7]; lc.c.g.add_bs = r[22];
                    if (lc.c.g.add && !lc.c.g.res_mul) return -1;          // raw accumulation is res (kind 0 form) or add WITH mul
                    lc.c.p = plan_conv(lc.c.g);
                    lc.c.zeros = pk; lc.c.wp = pk + off; lc.prob = i;
                    off += (long)lc.c.p.wp_floats;
                    out.push_back(lc);
                }
        }
    }
    return (int)out.size();
}

// split-K of the classes of ONE launch: fill the chip with all of them together, then cut classes whose workgroups would run
// much longer than the launch as a whole (a 512-channel layer on an 8x26 map next to a 128-channel one on 64x208)
static void list_plan_splits(ListCls** cls, int n, int target) {
    long fill = 0;
    double work = 0;
    for (int k = 0; k < n; k++) {
        const ConvPlan& p = cls[k]->c.p;
        const GG& g = cls[k]->c.g;
        const long blocks = conv_tiles(g, p) * (p.Mpad / p.bm);
        fill += blocks;
        work += (double)blocks * (p.Cpad / p.ck) * ((g.Rt * g.St + 2) / 3);
    }
    const long slots = fill < 256 ? 256 : (fill > 1024 ? 1024 : fill);
    const double t_ideal = work / (double)slots;            // stages per workgroup slot if the launch were perfectly balanced
    for (int k = 0; k < n; k++) {
        ConvPlan& p = cls[k]->c.p;
        const GG& g = cls[k]->c.g;
        const int nchunk = p.Cpad / p.ck;
        long want = 1;
        if (fill < env_int_early("CC_CONV_FILL_BELOW", 256) && nchunk >= 4) want = (target + fill - 1) / fill;
        else if (nchunk >= 8) {
            const double len = (double)nchunk * ((g.Rt * g.St + 2) / 3);
            const double cmin = env_int_early("CC_CONV_CLASS_STAGES", 24);
            const double cap = t_ideal > cmin ? t_ideal : cmin;
            if (len > 2 * cap) want = (long)(len / cap + 0.999);
        }
        if (want > nchunk / env_int_early("CC_CONV_MINCHUNKS", 1)) want = nchunk / env_int_early("CC_CONV_MINCHUNKS", 1);
        if (want > env_int_early("CC_CONV_MAXSPLIT", 32)) want = env_int_early("CC_CONV_MAXSPLIT", 32);
        p.nsplit = 1; p.cps = nchunk;
        if (want >= 2) {
            p.cps = (int)((nchunk + want - 1) / want);
            p.nsplit = (nchunk + p.cps - 1) / p.cps;
        }
        p.part_floats = p.nsplit > 1 ? (size_t)p.nsplit * g.B * g.M * p.Hp * p.Wp : 0;
    }
}

// groups the classes into launches (same (bm, ck), <= MAXCLS, list order kept inside a launch), plans the splits, assigns the
// partial-slab areas; launches when s_or_null is a stream (ws != nullptr).  -> floats of workspace needed / used, -1: error
static long list_run(int n, const long* d, float* ws, int target, bool launch, hipStream_t s) {
    std::vector<ListCls> all;
    if (list_classes(n, d, all) < 0) return -1;
    std::vector<char> done(all.size(), 0);
    long off = 64;
    for (size_t i = 0; i < all.size(); i++) {
        if (done[i]) continue;
        if (!all[i].c.p.use_patch) {
            done[i] = 1;
            if (launch) launch_gg_flat(all[i].c.g, s);
            continue;
        }
        if (all[i].c.p.wino) {       // Winograd problems keep the plan of their own geometry: one launch each
            done[i] = 1;
            all[i].c.part = ws ? ws + off : nullptr;
            off += (long)(all[i].c.p.part_floats + all[i].c.p.pad_floats);
            if (launch && !launch_classes(&all[i].c, 1, s)) return -1;
            continue;
        }
        ListCls* grp[MAXCLS];
        int m = 0;
        for (size_t j = i; j < all.size() && m < MAXCLS; j++)
            if (!done[j] && all[j].c.p.use_patch && !all[j].c.p.wino && all[j].c.p.bm == all[i].c.p.bm && all[j].c.p.ck == all[i].c.p.ck) {
                grp[m++] = &all[j];
                done[j] = 1;
            }
        list_plan_splits(grp, m, target);
        ClsIn cs[MAXCLS];
        for (int k = 0; k < m; k++) {
            grp[k]->c.part = ws ? ws + off : nullptr;
            off += (long)grp[k]->c.p.part_floats;
            cs[k] = grp[k]->c;
        }
        if (launch && !launch_classes(cs, m, s, true)) return -1;
    }
    return off;
}

size_t cc_conv2d_list_ws_bytes(int n, const long* desc_host, int split_target) {
    if (n <= 0 || !desc_host) return 0;
    const long f = list_run(n, desc_host, nullptr, split_target > 0 ? split_target : 512, false, nullptr);
    return f < 0 ? 0 : (size_t)f * sizeof(float);
}

int cc_conv2d_list(int n, const long* desc_host, float* ws, int split_target, void* stream) {
    if (n <= 0 || !desc_host || !ws) return CC_ERR_ARG;
    if (list_run(n, desc_host, ws, split_target > 0 ? split_target : 512, true, (hipStream_t)stream) < 0) return CC_ERR_ARG;
    CC_CHECK_LAUNCH();
    return CC_OK;
}

struct W3Plan { bool ok; int mt, nbuf, tiles_x, tiles_y, ntiles, nsplit, tps, Cp32; size_t smem, ws_floats; };

static int env_int(const char* name, int dflt) { return cctools::env_int(name, dflt); }

inline W3Plan plan_w3(int B, int M, int AH, int AW, int Cin, int R, int S, int si, int pad, int IH, int IW, int G = 1) {
    W3Plan p = {};
    p.ok = (R == 3 && S == 3 && si == 1 && pad == 1 && IH == AH && IW == AW && (AW % 4) == 0 && AW >= 16 && Cin >= 32 && M >= env_int("CC_W3_MINM", 64) &&
            !dbg_flag("CC_NO_WGRAD3X3"));   // measured (tools/wgrad_ablate.py): wins for M > 64 (1.2-1.45x), loses below 64;
                                            // M = 64 (<2, 2> tiles): -0.2 ms/step against the thin / generic kernels (r3o A/B)
    if (!p.ok) return p;
    p.mt = (M > 64) ? 4 : ((M > 32 && Cin > 32) ? 2 : ((Cin > 64) ? 1 : 2));
    p.nbuf = 1;
    const int BM = 32 * p.mt, BC = 32 * (4 / p.mt);
    p.Cp32 = ((Cin + BC - 1) / BC) * BC;
    p.tiles_x = (AW + 31) / 32;
    p.tiles_y = (AH + 1) / 2;
    p.ntiles = B * p.tiles_x * p.tiles_y;
    const long base = (long)((M + BM - 1) / BM) * (p.Cp32 / BC) * (G > 1 ? G : 1);
    long nsplit = (env_int("CC_W3_SPLIT", 512) + base - 1) / base;      // 512: measured -0.27 ms/step vs 256 (r02f A/B)
    const long mt_ = env_int("CC_W3_MINTILES", 3);
    const long cap = (p.ntiles + mt_ - 1) / mt_;  // >= 6 pixel tiles per split: every split writes a 9*M*Cpad partial slab
    if (nsplit > cap) nsplit = cap;
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    if (nsplit < 1) nsplit = 1;
    p.tps = (int)((p.ntiles + nsplit - 1) / nsplit);
    p.nsplit = (p.ntiles + p.tps - 1) / p.tps;
    p.smem = (size_t)(BM * 16 + ((BC * W3_PC + 63) / 64) * 64) * 16;
    p.ws_floats = 64 + (size_t)p.nsplit * 9 * M * p.Cp32;
    return p;
}

static size_t wgrad_ws_bytes_base(int B, int M, int AH, int AW, int Cin, int R, int S, int si);
static size_t wgrad_ws_bytes_rest(int B, int M, int AH, int AW, int Cin, int R, int S, int si);

struct WinoPadPlan { bool ok; int Wp; ccint::WinoWgradPlan wp; size_t pad_floats; };     // pad_floats: padded x + dY of ONE problem
inline WinoPadPlan wino_pad_plan(int B, int M, int AH, int AW, int Cin, int G) {
    WinoPadPlan p = {};
    if ((AW % 4) == 0 || AH < 2 || dbg_flag("CC_NO_WINO_WGRAD_PAD")) return p;
    // large weight matrices only: the copies and the kernel's per-workgroup epilogue have to pay (measured per shape)
    if (M < env_int_early("CC_WWP_MINM", 96) || Cin < env_int_early("CC_WWP_MINC", 96)) return p;
    p.Wp = (AW + 3) & ~3;
    p.wp = ccint::wino_wgrad_plan(B, M, AH, p.Wp, Cin, G, env_int_early("CC_WWP_MINQ", 64));
    p.ok = p.wp.ok != 0;
    p.pad_floats = ((size_t)B * (Cin + M) * AH * p.Wp + 3) & ~(size_t)3;
    return p;
}

size_t cc_conv2d_wgrad_ws_bytes(int B, int M, int AH, int AW, int Cin, int R, int S, int si) {
    // the thin path is confirmed at launch (pad, input width): size for it AND for the path it would fall back to
    size_t thin = ccint::wgrad_thin_ws_floats(B, M, AH, AW, Cin, R, S, si) * sizeof(float);
    if (R == 3 && S == 3 && si == 1 && M <= 2) {          // a head's weight gradient (conv_heads.hip)
        const ccint::HeadWgradPlan hp = ccint::head_wgrad_plan(B, M, AH, AW, Cin);
        if (hp.ok && hp.ws_floats * sizeof(float) > thin) thin = hp.ws_floats * sizeof(float);
    }
    const size_t base = wgrad_ws_bytes_base(B, M, AH, AW, Cin, R, S, si);
    return ((thin > base ? thin : base) + 15) & ~(size_t)15;        // (the areas of a group's problems follow each other: keep them 16-byte aligned)
}

static size_t wgrad_ws_bytes_base(int B, int M, int AH, int AW, int Cin, int R, int S, int si) {
    size_t wino = 0;
    if (R == 3 && S == 3 && si == 1) {       // Winograd path (pad / input size are implied by "same" convolutions: checked again at launch)
        for (int G = 1; G <= MAXGRP; G++) {       // every group size: the split search is not monotonic in G
            const ccint::WinoWgradPlan wp = ccint::wino_wgrad_plan(B, M, AH, AW, Cin, G);
            if (wp.ok && wp.ws_floats * sizeof(float) > wino) wino = wp.ws_floats * sizeof(float);
            const WinoPadPlan pp = wino_pad_plan(B, M, AH, AW, Cin, G);
            if (pp.ok && (pp.wp.ws_floats + pp.pad_floats) * sizeof(float) > wino) wino = (pp.wp.ws_floats + pp.pad_floats) * sizeof(float);
        }
    }
    const size_t rest = wgrad_ws_bytes_rest(B, M, AH, AW, Cin, R, S, si);
    return wino > rest ? wino : rest;
}

static size_t wgrad_ws_bytes_rest(int B, int M, int AH, int AW, int Cin, int R, int S, int si) {
    {   // the 3x3/s1/p1 path (pad and input size are implied by "same" convolutions: checked again at launch)
        const W3Plan q = plan_w3(B, M, AH, AW, Cin, R, S, si, 1, AH, AW);
        if (q.ok) return q.ws_floats * sizeof(float);
    }
    const WPlan p = plan_wgrad(B, M, AH, AW, Cin, R, S, si);
    if (p.ok) return p.ws_floats * sizeof(float);
    const long Ntot = (long)Cin * R * S;
    const long P = (long)B * AH * AW;
    const int bm = pick_bm(M);
    const long tiles = ((Ntot + BN - 1) / BN) * ((M + bm - 1) / bm);
    long nsplit = (env_int_early("CC_WGRAD_SPLIT_TARGET", 512) + tiles - 1) / tiles;
    const long mr = env_int_early("CC_WGRAD_MINRANGE", 32);
    const long maxsplit = (P + mr - 1) / mr;  // small maps still need >= 256 workgroups: split down to 32-pixel ranges (-0.16 ms/step against 64, r3s3)
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    return nsplit <= 1 ? 16 : (size_t)nsplit * M * Ntot * sizeof(float);
}

/* gw[m, c, r, s] (strides o_*) = sum_{n,ty,tx} a[n, m, ty, tx] * x[n, c, si*ty - pad + r, si*tx - pad + s].
 * conv2d weight-gradient: a = dY [B,Cout,OH,OW], x = input, si = stride, o strides of [Cout,Cin,R,S];
 * ConvTranspose2d weight-gradient: a = input [B,Cin,IH,IW], x = dY, si = stride, o strides of [Cin,Cout,R,S].
 * Group form: G (<= 4) same-shaped problems in one launch (+ one reduction launch); a / x / gw: HOST arrays of device
 * addresses; ws: G consecutive areas of cc_conv2d_wgrad_ws_bytes() each. */
// launches of the generic kernel collected by cc_conv2d_wgrad_list instead of issued one by one
struct WgradParked { WG g; int bm; dim3 grid; double gflop; };
struct WgradCollector { WgradParked* p; int cap, n; ccint::WinoWgradParked* wino; };      // wino: parked Winograd problems (or null)

static void launch_wgrad_parked(const WgradCollector& c, hipStream_t s) {
    for (int bm = 128; bm >= 32; bm /= 2) {
        int i = 0;
        while (i < c.n) {
            WGM m = {};
            long blk = 0;
            double gf = 0;
            for (; i < c.n && m.n < MAXWCLS; i++) {
                if (c.p[i].bm != bm) continue;
                const dim3& gr = c.p[i].grid;
                const long nb = (long)gr.x * gr.y * gr.z;
                if (m.n && blk + nb >= (1l << 31)) break;          // (a problem that large goes into a launch of its own: never dropped)
                m.c[m.n] = c.p[i].g;
                m.gx[m.n] = (int)gr.x; m.gy[m.n] = (int)gr.y;
                blk += nb;
                m.bx_end[m.n] = (int)blk;
                gf += c.p[i].gflop;
                m.n++;
            }
            if (!m.n) break;
            char nm[64];
            snprintf(nm, sizeof nm, "k_wgrad_multi<%d>%s", bm, "");
            cctiming::Scope tsc(nm, gf, s);
            if (bm == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_multi<128>), dim3((unsigned)blk), dim3(256), 0, s, m);
            else if (bm == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_multi<64>), dim3((unsigned)blk), dim3(256), 0, s, m);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_multi<32>), dim3((unsigned)blk), dim3(256), 0, s, m);
        }
    }
}

static int wgrad_group_impl(int G, const long* a, const long* x, const long* gw, float* ws, int B, int M, int AH, int AW, long a_bs,
                            int Cin, int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate,
                            void* stream, ccint::RedSink* sink, const float* zeros64 = nullptr, WgradCollector* park = nullptr) {
    if (G <= 0 || G > MAXGRP || B <= 0 || M <= 0 || Cin <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    long rd[MAXGRP][ccint::RD_LONGS];
    const size_t stride_f = cc_conv2d_wgrad_ws_bytes(B, M, AH, AW, Cin, R, S, si) / sizeof(float);
    if (R == 3 && S == 3 && si == 1 && pad == 1 && IH == AH && IW == AW) {
        // Winograd F(3x3, 2x2): 16 instead of 36 multiply-adds per 2x2 tile (wino_wgrad.hip); same slab layout / reduction as k_wgrad3x3
        ccint::WinoWgradPlan wp = ccint::wino_wgrad_plan(B, M, AH, AW, Cin, G);
        if (wp.ok && park && park->wino && park->wino->n + G <= ccint::WINO_WGRAD_PARK_CAP) wp = ccint::wino_wgrad_plan_parked(wp, M);
        if (wp.ok) {
            const float *ap[MAXGRP], *xp[MAXGRP];
            float* wsp[MAXGRP];
            for (int k = 0; k < G; k++) { ap[k] = (const float*)a[k]; xp[k] = (const float*)x[k]; wsp[k] = ws + k * stride_f + 64; }
            bool ok;
            {
                char nm[128];
                int nl = snprintf(nm, sizeof nm, "k_wino_wgrad");
                if (cctools::env_flag("CC_TIMING_DETAIL"))
                    snprintf(nm + nl, sizeof nm - nl, " G%d B%d M%d C%d %dx%d k%d wg%d", G, B, M, Cin, AH, AW, wp.nsplit,
                             wp.nmb * wp.ncb * G * wp.nsplit);
                ccint::WinoWgradParked* wpark = park ? park->wino : nullptr;
                cctiming::Scope tsc(nm, 2e-9 * 16.0 * G * B * ((AH + 1) / 2) * ((AW + 1) / 2) * (double)M * Cin, s,
                                    !(wpark && wpark->n + G <= ccint::WINO_WGRAD_PARK_CAP));
                ok = ccint::wino_wgrad_launch(wp, ap, xp, wsp, G, B, M, AH, AW, a_bs, Cin, x_bs, s, wpark);
            }
            if (ok) {
                for (int k = 0; k < G; k++) {
                    const long d[ccint::RD_LONGS] = {1, (long)wsp[k], (long)gw[k], wp.nsplit, accumulate, o_sm, o_sc, 9, M, Cin, wp.Cp};
                    for (int i = 0; i < ccint::RD_LONGS; i++) rd[k][i] = d[i];
                }
                if (ccint::wgrad_reduce_emit(sink, &rd[0][0], G, s) != CC_OK) return CC_ERR_ARG;
                CC_CHECK_LAUNCH();
                return CC_OK;
            }
        }
    }
    if (R == 3 && S == 3 && si == 1 && pad == 1 && IH == AH && IW == AW) {
        WinoPadPlan pp = wino_pad_plan(B, M, AH, AW, Cin, G);
        if (pp.ok && park && park->wino && park->wino->n + G <= ccint::WINO_WGRAD_PARK_CAP) {
            const size_t slabs = (pp.wp.ws_floats + 3) & ~(size_t)3;          // the padded copies keep their place behind the stand-alone plan's slabs
            pp.wp = ccint::wino_wgrad_plan_parked(pp.wp, M);
            pp.wp.ws_floats = slabs;
        }
        if (pp.ok) {
            const float *ap[MAXGRP], *xp[MAXGRP];
            float* wsp[MAXGRP];
            PadTab t = {};
            t.B = B; t.W = AW; t.Wp = pp.Wp;
            long rows = 0;
            for (int k = 0; k < G; k++) {
                float* area = ws + k * stride_f;
                wsp[k] = area + 64;
                float* xpad = area + ((pp.wp.ws_floats + 3) & ~(size_t)3);
                float* apad = xpad + (size_t)B * Cin * AH * pp.Wp;
                t.j[2 * k] = PadJob{(const float*)x[k], xpad, x_bs, Cin * AH};
                rows += (long)B * Cin * AH; t.row_end[2 * k] = rows;
                t.j[2 * k + 1] = PadJob{(const float*)a[k], apad, a_bs, M * AH};
                rows += (long)B * M * AH; t.row_end[2 * k + 1] = rows;
                xp[k] = xpad; ap[k] = apad;
            }
            t.n = 2 * G;
            const long nf4 = rows * (pp.Wp >> 2);
            bool ok;
            {
                char nm[128];
                int nl = snprintf(nm, sizeof nm, "k_wino_wgrad");
                if (cctools::env_flag("CC_TIMING_DETAIL"))
                    snprintf(nm + nl, sizeof nm - nl, " G%d B%d M%d C%d %dx%d(pad %d) k%d wg%d", G, B, M, Cin, AH, AW, pp.Wp, pp.wp.nsplit,
                             pp.wp.nmb * pp.wp.ncb * G * pp.wp.nsplit);
                ccint::WinoWgradParked* wpark = park ? park->wino : nullptr;
                cctiming::Scope tsc(nm, 2e-9 * 16.0 * G * B * ((AH + 1) / 2) * (pp.Wp / 2) * (double)M * Cin, s,
                                    !(wpark && wpark->n + G <= ccint::WINO_WGRAD_PARK_CAP));
                hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((nf4 + 255) / 256)), dim3(256), 0, s, t);
                ok = ccint::wino_wgrad_launch(pp.wp, ap, xp, wsp, G, B, M, AH, pp.Wp, (long)M * AH * pp.Wp, Cin, (long)Cin * AH * pp.Wp, s, wpark);
            }
            if (ok) {
                for (int k = 0; k < G; k++) {
                    const long d[ccint::RD_LONGS] = {1, (long)wsp[k], (long)gw[k], pp.wp.nsplit, accumulate, o_sm, o_sc, 9, M, Cin, pp.wp.Cp};
                    for (int i = 0; i < ccint::RD_LONGS; i++) rd[k][i] = d[i];
                }
                if (ccint::wgrad_reduce_emit(sink, &rd[0][0], G, s) != CC_OK) return CC_ERR_ARG;
                CC_CHECK_LAUNCH();
                return CC_OK;
            }
        }
    }
    if (R == 3 && S == 3 && si == 1 && pad == 1 && IH == AH && IW == AW && M <= 2) {
        // weight gradient of a prediction head: HBM-bound VALU kernel (conv_heads.hip), one launch per problem
        const ccint::HeadWgradPlan hp = ccint::head_wgrad_plan(B, M, AH, AW, Cin);
        if (hp.ok && hp.ws_floats <= stride_f) {
            bool ok = true;
            char nm[128];
            int nl = snprintf(nm, sizeof nm, "k_wgrad_thinm<%d>", M);
            if (cctools::env_flag("CC_TIMING_DETAIL"))
                snprintf(nm + nl, sizeof nm - nl, " G%d B%d M%d C%d %dx%d r3 s1 k%d", G, B, M, Cin, AH, AW, hp.nblk);
            {
                cctiming::Scope tsc(nm, 2e-9 * G * B * AH * AW * (double)M * Cin * 9, s);
                for (int k = 0; k < G && ok; k++)
                    ok = ccint::head_wgrad_launch(hp, (const float*)a[k], (const float*)x[k], ws + k * stride_f, B, M, AH, AW, a_bs, Cin, x_bs, s);
            }
            if (ok && cctools::env_flag("CC_HEAD_TRACE"))
                fprintf(stderr, "head wgrad: G%d B%d M%d C%d %dx%d R%d nblk %d\n", G, B, M, Cin, AH, AW, hp.R, hp.nblk);
            if (ok) {
                for (int k = 0; k < G; k++) {
                    const long d[ccint::RD_LONGS] = {0, (long)(ws + k * stride_f), (long)gw[k], hp.nblk, accumulate, o_sm, o_sc, M, (long)Cin * 9, 9, 3, 3, 1};
                    for (int i = 0; i < ccint::RD_LONGS; i++) rd[k][i] = d[i];
                }
                if (ccint::wgrad_reduce_emit(sink, &rd[0][0], G, s) != CC_OK) return CC_ERR_ARG;
                CC_CHECK_LAUNCH();
                return CC_OK;
            }
        }
    }
    {   // thin layers fill the chip on their own: one launch per problem
        bool thin = true;
        char nm[128];
        ccint::wgrad_thin_name(B, M, AH, AW, Cin, IH, IW, R, S, si, pad, nm, 64);
        if (nm[0] && cctools::env_flag("CC_TIMING_DETAIL")) {
            const size_t nl = strlen(nm);
            snprintf(nm + nl, sizeof nm - nl, " G%d B%d M%d C%d %dx%d r%d s%d", G, B, M, Cin, AH, AW, R, si);
        }
        // (recorded only when the thin path is taken: the name is empty otherwise; alignment can still turn a problem away, then
        // the record brackets nothing)
        cctiming::Scope tsc(nm[0] ? nm : "k_wgrad_thin<declined>", nm[0] ? 2e-9 * G * B * AH * AW * (double)M * Cin * R * S : 0.0,
                            nm[0] ? s : nullptr, nm[0] != 0 && !ccint::wgrad_thin_parking());
        for (int k = 0; k < G && thin; k++)
            thin = ccint::wgrad_thin_launch((const float*)a[k], (const float*)x[k], (float*)gw[k], ws + k * stride_f, B, M, AH, AW, a_bs,
                                            Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate, s, sink);
        // eligibility depends on the geometry (and 16-byte alignment of the pointers): all problems or none, in practice
        if (thin) {
            CC_CHECK_LAUNCH();
            return CC_OK;
        }
    }
    RG rg = {};
    const W3Plan q = plan_w3(B, M, AH, AW, Cin, R, S, si, pad, IH, IW, G);
    if (q.ok) {
        W3 w = {};
        w.zeros = zeros64 ? zeros64 : ws;
        for (int k = 0; k < G; k++) {
            w.ga[k] = (const float*)a[k]; w.gxp[k] = (const float*)x[k]; w.gws[k] = ws + k * stride_f + 64;
            rg.ws[k] = w.gws[k]; rg.gw[k] = (float*)gw[k];
        }
        w.B = B; w.M = M; w.AH = AH; w.AW = AW; w.a_bs = a_bs; w.Cin = Cin; w.x_bs = x_bs;
        w.tiles_x = q.tiles_x; w.tiles_y = q.tiles_y; w.ntiles = q.ntiles; w.tiles_per_split = q.tps; w.nsplit = q.nsplit;
        w.Cpad = q.Cp32;
        if (!zeros64) hipLaunchKernelGGL(k_zero64, dim3(1), dim3(64), 0, s, ws);
        const int BM = 32 * q.mt, BC = 32 * (4 / q.mt);
        dim3 grid((unsigned)(((M + BM - 1) / BM) * (q.Cp32 / BC)), (unsigned)G, (unsigned)q.nsplit);
        w.dbg = env_int("CC_W3_DBG", 0);
        {
            char nm[128];
            int nl = snprintf(nm, sizeof nm, "k_wgrad3x3<%d, %d>", q.mt, 4 / q.mt);
            if (cctools::env_flag("CC_TIMING_DETAIL"))
                snprintf(nm + nl, sizeof nm - nl, " G%d B%d M%d C%d %dx%d k%d wg%d", G, B, M, Cin, AH, AW, q.nsplit,
                         (int)(grid.x * grid.y * grid.z));
            cctiming::Scope tsc(nm, 2e-9 * G * B * AH * AW * (double)M * Cin * 9, s);
            if (q.mt == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad3x3<4, 1>), grid, dim3(256), q.smem, s, w);
            else if (q.mt == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad3x3<2, 2>), grid, dim3(256), q.smem, s, w);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad3x3<1, 4>), grid, dim3(256), q.smem, s, w);
        }
        for (int k = 0; k < G; k++) {
            const long d[ccint::RD_LONGS] = {1, (long)rg.ws[k], (long)rg.gw[k], q.nsplit, accumulate, o_sm, o_sc, 9, M, Cin, q.Cp32};
            for (int i = 0; i < ccint::RD_LONGS; i++) rd[k][i] = d[i];
        }
        if (ccint::wgrad_reduce_emit(sink, &rd[0][0], G, s) != CC_OK) return CC_ERR_ARG;
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    const WPlan p = plan_wgrad(B, M, AH, AW, Cin, R, S, si);
    if (p.ok) {       // experimental per-tap kernel (CC_WGRAD_PATCH=1): one problem at a time
        for (int k = 0; k < G; k++) {
            float* wsk = ws + k * stride_f;
            WP w = {};
            w.a = (const float*)a[k]; w.x = (const float*)x[k]; w.zeros = wsk; w.ws = wsk + 64;
            w.B = B; w.M = M; w.AH = AH; w.AW = AW; w.a_bs = a_bs; w.Cin = Cin; w.IH = IH; w.IW = IW; w.x_bs = x_bs;
            w.R = R; w.S = S; w.si = si; w.pad = pad; w.PH = p.PH; w.PWr = p.PWr; w.PSc = p.PSc; w.npos = p.npos;
            w.tiles_x = p.tiles_x; w.tiles_y = p.tiles_y; w.ntiles = p.ntiles; w.tiles_per_split = p.tps; w.nsplit = p.nsplit;
            w.TG = p.TG; w.ngroups = p.ngroups; w.Cp32 = p.Cp32; w.nbuf = p.nbuf;
            w.dbg = cctools::env_int("CC_WGRAD_DBG", 0);
            hipLaunchKernelGGL(k_zero64, dim3(1), dim3(64), 0, s, wsk);      // the LDS-DMA halo source (a kernel, not a memset node)
            dim3 grid((unsigned)(((M + p.bmw - 1) / p.bmw) * (p.Cp32 / 32) * p.ngroups), 1, (unsigned)p.nsplit);
            if (p.bmw == 64) {
                if (p.nt == 5) launch_wgrad_patch<64, 5>(w, grid, p.smem, s);
                else if (p.nt == 4) launch_wgrad_patch<64, 4>(w, grid, p.smem, s);
                else if (p.nt == 3) launch_wgrad_patch<64, 3>(w, grid, p.smem, s);
                else if (p.nt == 2) launch_wgrad_patch<64, 2>(w, grid, p.smem, s);
                else launch_wgrad_patch<64, 1>(w, grid, p.smem, s);
            } else {
                if (p.nt >= 3) launch_wgrad_patch<32, 3>(w, grid, p.smem, s);
                else if (p.nt == 2) launch_wgrad_patch<32, 2>(w, grid, p.smem, s);
                else launch_wgrad_patch<32, 1>(w, grid, p.smem, s);
            }
            const long d[ccint::RD_LONGS] = {1, (long)w.ws, (long)gw[k], p.nsplit, accumulate, o_sm, o_sc, R * S, M, Cin, p.Cp32};
            if (ccint::wgrad_reduce_emit(sink, d, 1, s) != CC_OK) return CC_ERR_ARG;
        }
        CC_CHECK_LAUNCH();
        return CC_OK;
    }
    const long Ntot = (long)Cin * R * S;
    const long P = (long)B * AH * AW;
    const int bm = pick_bm(M);
    const long tiles = ((Ntot + BN - 1) / BN) * ((M + bm - 1) / bm) * G;
    // a problem that will share a launch with others (cc_conv2d_wgrad_list) does not have to fill the chip alone: fewer, longer
    // pixel ranges -- fewer 64 KB partial tiles written, reduced and paid for in epilogues (never more splits than stand-alone:
    // the workspace is sized for that).  Measured (profiles/r04_ab_round4.txt): target 256 / ranges >= 64 pixels -0.13 ms against
    // the stand-alone plan; much longer chains lose again (target 64: +0.5 ms, 32: +1.6 ms -- the kernel is slow per k-step)
    const bool parked = park && park->n < park->cap;
    const long target = parked ? env_int_early("CC_WGRAD_PARK_TARGET", 256) : env_int_early("CC_WGRAD_SPLIT_TARGET", 512);
    long nsplit = (target + tiles - 1) / tiles;
    const long mr = parked ? env_int_early("CC_WGRAD_PARK_MINRANGE", 64) : env_int_early("CC_WGRAD_MINRANGE", 32);
    const long maxsplit = (P + mr - 1) / mr;  // small maps still need >= 256 workgroups: split down to 32-pixel ranges (-0.16 ms/step against 64, r3s3)
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    long pps = (P + nsplit - 1) / nsplit;
    pps = ((pps + BK - 1) / BK) * BK;
    nsplit = (P + pps - 1) / pps;
    WG g = {};
    g.B = B; g.M = M; g.AH = AH; g.AW = AW; g.a_bs = a_bs;
    g.Cin = Cin; g.IH = IH; g.IW = IW; g.x_bs = x_bs;
    g.Rt = R; g.St = S; g.dy0 = -pad; g.dx0 = -pad; g.dstep = 1; g.si = si;
    g.o_sm = o_sm; g.o_sc = o_sc; g.o_ri = S; g.o_sj = 1;
    g.direct = (nsplit == 1);
    g.accum = accumulate;
    g.nsplit = (int)nsplit;
    for (int k = 0; k < G; k++) {
        g.ga[k] = (const float*)a[k]; g.gxp[k] = (const float*)x[k];
        g.gout[k] = g.direct ? (float*)gw[k] : ws + k * stride_f;
        rg.ws[k] = ws + k * stride_f; rg.gw[k] = (float*)gw[k];
    }
    g.pix_per_split = (int)pps;
    dim3 grid((unsigned)((Ntot + BN - 1) / BN), (unsigned)((M + bm - 1) / bm), (unsigned)(nsplit * G));
    if (park && park->n < park->cap) {
        park->p[park->n++] = WgradParked{g, bm, grid, 2e-9 * G * B * AH * AW * (double)M * Cin * R * S};
    } else {
        char nm[128];
        int nl = snprintf(nm, sizeof nm, "k_wgrad<%d>", bm);
        if (cctools::env_flag("CC_TIMING_DETAIL"))
            snprintf(nm + nl, sizeof nm - nl, " G%d B%d M%d C%d %dx%d r%d s%d k%ld wg%d", G, B, M, Cin, AH, AW, R, si, (long)nsplit,
                     (int)(grid.x * grid.y * grid.z));
        cctiming::Scope tsc(nm, 2e-9 * G * B * AH * AW * (double)M * Cin * R * S, s);
        if (bm == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad<128>), grid, dim3(256), 0, s, g);
        else if (bm == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad<64>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad<32>), grid, dim3(256), 0, s, g);
    }
    if (!g.direct) {
        for (int k = 0; k < G; k++) {
            const long d[ccint::RD_LONGS] = {0, (long)rg.ws[k], (long)rg.gw[k], nsplit, accumulate, o_sm, o_sc, M, Ntot, R * S, S, S, 1};
            for (int i = 0; i < ccint::RD_LONGS; i++) rd[k][i] = d[i];
        }
        if (ccint::wgrad_reduce_emit(sink, &rd[0][0], G, s) != CC_OK) return CC_ERR_ARG;
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_conv2d_wgrad_group(int G, const long* a, const long* x, const long* gw, float* ws, int B, int M, int AH, int AW, long a_bs,
                          int Cin, int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate,
                          void* stream) {
    return wgrad_group_impl(G, a, x, gw, ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate, stream,
                            nullptr);
}

/* ... with the reductions of the partial slabs left to the caller: their descriptors (16 longs each, at most G) are written to
 * red_host[0 .. *nred_host) and ws must stay untouched until cc_wgrad_reduce_table has run on them.  zeros64_or_null: 64 zero
 * floats that outlive the launch (the LDS-DMA source of halo pixels; without it a fill launch precedes the kernel). */
int cc_conv2d_wgrad_group_defer(int G, const long* a, const long* x, const long* gw, float* ws, int B, int M, int AH, int AW,
                                long a_bs, int Cin, int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc,
                                int accumulate, const float* zeros64_or_null, long* red_host, int red_cap, int* nred_host,
                                void* stream) {
    if (!red_host || !nred_host || red_cap < G) return CC_ERR_ARG;
    ccint::RedSink sink = {red_host, red_cap, 0};
    const int r = wgrad_group_impl(G, a, x, gw, ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate,
                                   stream, &sink, zeros64_or_null);
    *nred_host = sink.n;
    return r;
}

/* n groups of DIFFERENT shapes (what a backward stage has parked at its end): desc_host = n x 32 longs
 *   {G, a[4], x[4], gw[4], ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate, 0, 0}
 * -- each group exactly as one cc_conv2d_wgrad_group_defer call (same kernels, same arithmetic, same reduce descriptors, in list
 * order), except that the groups the planner sends to the generic kernel share launches (k_wgrad_multi: up to 12 per launch). */
int cc_conv2d_wgrad_list(int n, const long* desc_host, const float* zeros64_or_null, long* red_host, int red_cap, int* nred_host,
                         void* stream) {
    if (n <= 0 || !desc_host || !red_host || !nred_host) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    constexpr int CAP = 64;
    static thread_local WgradParked parked[CAP];
    static thread_local ccint::WinoWgradParked wino_parked;
    wino_parked.n = 0;
    WgradCollector col = {parked, CAP, 0, cctools::env_flag("CC_NO_WINO_WGRAD_LIST") ? nullptr : &wino_parked};
    ccint::RedSink sink = {red_host, red_cap, 0};
    // the thin weight gradients of the list share launches too (wgrad_thin.hip: per kernel instance); launched when this call ends
    struct ThinPark {
        hipStream_t s; bool was;
        explicit ThinPark(hipStream_t st) : s(st), was(ccint::wgrad_thin_park(true)) {}
        ~ThinPark() {
            const double gf = ccint::wgrad_thin_parked_gflop();
            if (gf > 0) {
                cctiming::Scope tsc("k_wgrad_thin_multi", gf, s);
                ccint::wgrad_thin_flush(s);
            }
            ccint::wgrad_thin_park(was);
        }
    } thin_park(s);
    for (int i = 0; i < n; i++) {
        const long* d = desc_host + 32l * i;
        const int G = (int)d[0];
        // on an error in the middle of the list: what was collected is launched (its reduce descriptors describe slabs that are then
        // really written) and the descriptors emitted so far are handed back, so that the caller's state stays consistent
        auto bail = [&](int code) {
            launch_wgrad_parked(col, s);
            if (wino_parked.n > 0) ccint::wino_wgrad_launch_parked(&wino_parked, s);
            *nred_host = sink.n;
            return code;
        };
        if (G <= 0 || G > MAXGRP || sink.n + G > red_cap) return bail(CC_ERR_ARG);
        const int before = col.n;
        const int r = wgrad_group_impl(G, d + 1, d + 5, d + 9, (float*)d[13], (int)d[14], (int)d[15], (int)d[16], (int)d[17], d[18],
                                       (int)d[19], (int)d[20], (int)d[21], d[22], (int)d[23], (int)d[24], (int)d[25], (int)d[26], d[27],
                                       d[28], (int)d[29], stream, &sink, zeros64_or_null, &col);
        if (r != CC_OK) return bail(r);
        if (col.n > before && col.p[before].g.direct) {
            // a problem that writes its gradient itself (no split): it must not share a launch with an earlier one of the same target
            bool dup = false;
            for (int j = 0; j < before && !dup; j++)
                if (col.p[j].g.direct)
                    for (int u = 0; u < MAXGRP && !dup; u++)
                        for (int v = 0; v < MAXGRP; v++)
                            if (col.p[j].g.gout[u] && col.p[j].g.gout[u] == col.p[before].g.gout[v]) { dup = true; break; }
            if (dup) {
                const WgradParked keep = col.p[before];
                col.n = before;
                launch_wgrad_parked(col, s);
                col.p[0] = keep;
                col.n = 1;
            }
        }
    }
    launch_wgrad_parked(col, s);
    if (wino_parked.n > 0) {
        double gf = 0;
        for (int i = 0; i < wino_parked.n; i++) gf += wino_parked.d[i].gflop;
        cctiming::Scope tsc("k_wino_wgrad_multi", gf, s);
        ccint::wino_wgrad_launch_parked(&wino_parked, s);
    }
    *nred_host = sink.n;
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_conv2d_wgrad(const float* a, const float* x, float* gw, float* ws, int B, int M, int AH, int AW, long a_bs, int Cin,
                    int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate, void* stream) {
    const long ap = (long)a, xp = (long)x, gp = (long)gw;
    return cc_conv2d_wgrad_group(1, &ap, &xp, &gp, ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate,
                                 stream);
}

/* ---- per-kernel timing (measurement aid): cc_timing_enable(1) starts recording (process-wide), cc_timing_collect
 * waits for the recorded kernels and writes one line per device kernel "name\tlaunches\ttotal_ms\ttotal_gflop\n" into the
 * HOST buffer (returns the number of characters, stops recording). */
#ifdef CC_TOOLS
int cc_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(cctiming::mtx);
    if (on && !cctiming::recs) cctiming::recs = new std::vector<cctiming::Rec>();
    if (on) cctiming::recs->reserve(4096);
    if (!on && cctiming::recs) {
        for (auto& r : *cctiming::recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        delete cctiming::recs;
        cctiming::recs = nullptr;
    }
    return CC_OK;
}

int cc_timing_collect(void* out_host, int cap) {
    char* out = (char*)out_host;
    int len = 0;
    if (!cctiming::recs || cap <= 0) return 0;
    struct Agg { std::string name; int n; double ms, gf; };
    std::vector<Agg> agg;
    for (auto& r : *cctiming::recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        Agg* a = nullptr;
        for (auto& x : agg) if (x.name == r.name) { a = &x; break; }
        if (!a) { agg.push_back(Agg{r.name, 0, 0.0, 0.0}); a = &agg.back(); }
        a->n++; a->ms += ms; a->gf += r.gflop;
    }
    for (auto& a : agg) {
        const int k = snprintf(out + len, cap - len, "%s\t%d\t%.6f\t%.6f\n", a.name.c_str(), a.n, a.ms, a.gf);
        if (k < 0 || k >= cap - len) break;
        len += k;
    }
    cc_timing_enable(0);       // (takes the lock itself)
    return len;
}
#else
/* product build: no timing registry (the library keeps no state); the tools build records */
int cc_timing_enable(int on) { return on ? CC_ERR_ARG : CC_OK; }
int cc_timing_collect(void* out_host, int cap) { (void)out_host; (void)cap; return 0; }
#endif

/* 1 for the tools build (switches + timing compiled in), 0 for the product library */
int cc_is_tools_build(void) {
#ifdef CC_TOOLS
    return 1;
#else
    return 0;
#endif
}

/* ---- introspection (bench.py groups its per-call timings by the kernel a call dispatches to) */
static void patch_name(const ConvPlan& p, bool multi, char* out, int cap) {
    if (p.wino && p.wn.tile) { snprintf(out, cap, "k_wino_f2x3_s<%d, %d>%s", p.wn.tile, p.nsplit > 1 ? 1 : 0, p.nsplit > 1 ? "+splitk" : ""); return; }
    if (p.wino) { snprintf(out, cap, "k_wino_f2x3<%d>%s", p.nsplit > 1 ? 1 : 0, p.nsplit > 1 ? "+splitk" : ""); return; }
    if (!p.use_patch) { snprintf(out, cap, "k_gather_gemm<%d>", pick_bm(p.Mpad ? p.Mpad : 32)); return; }
    const char* sk = p.nsplit > 1 ? "+splitk" : "";
    if (multi && p.ipt > 1) snprintf(out, cap, "k_conv_patch_multi_stk<%d, %d>%s", p.bm, p.tps, sk);
    else if (multi) snprintf(out, cap, "k_conv_patch_multi<%d, %d, %d>%s", p.bm, p.ck, p.tps, sk);
    else if (p.ipt > 1) snprintf(out, cap, "k_conv_patch_stk<%d, %d, %d>%s", p.bm, p.tps, p.nsplit > 1 ? 1 : 0, sk);
    else snprintf(out, cap, "k_conv_patch<%d, %d, %d, %d>%s", p.bm, p.ck, p.tps, p.nsplit > 1 ? 1 : 0, sk);
}

int cc_conv2d_fwd_kernel(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW,
                         void* name_out_host, int cap) {
    GG g = make_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, B, Cin, IH, IW, 0, Cout, R, S, stride, pad, OH, OW, 0, 0, 0,
                    1.f, 0.f);
    ccint::HeadConv h;
    if (const int hk = head_kernel_of(g, h)) {       // (geometry only: the launch also checks the tensors' alignment)
        snprintf((char*)name_out_host, cap, hk == 1 ? "k_conv_thinc<%d>" : "k_conv_thinm<%d>", hk == 1 ? g.Cin : g.M);
        return CC_OK;
    }
    patch_name(plan_conv(g), false, (char*)name_out_host, cap);
    return CC_OK;
}

int cc_conv2d_dgrad_kernel(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW,
                           int prepacked, void* name_out_host, int cap) {
    GG gs[4];
    int n = 0;
    bool all = true;
    for (int py = 0; py < stride && all; py++)
        for (int px = 0; px < stride; px++) {
            if (n >= 4 || !make_dgrad_class(gs[n], py, px, nullptr, nullptr, nullptr, nullptr, B, K, OH, OW, 0, C, R, S, stride, pad,
                                            IH, IW, 0, (long)C * R * S, (long)R * S, 0, 1.f, 0.f)) { all = false; break; }
            n++;
        }
    if (n == 0) { ((char*)name_out_host)[0] = 0; return CC_OK; }
    if (n == 1 && stride == 1) {
        ccint::HeadConv h;
        if (const int hk = head_kernel_of(gs[0], h)) {
            snprintf((char*)name_out_host, cap, hk == 1 ? "k_conv_thinc<%d>" : "k_conv_thinm<%d>", hk == 1 ? gs[0].Cin : gs[0].M);
            return CC_OK;
        }
    }
    ConvPlan p = plan_conv(gs[0]);
    bool multi = false;
    if (stride == 2 && prepacked && all && n >= 2 && !dbg_flag_early("CC_NO_CLASS_MERGE")) {
        multi = true;
        int tps = 3, maxsplit = 1;
        for (int k = 0; k < n; k++) {
            const ConvPlan q = plan_conv(gs[k]);
            if (!q.use_patch || q.bm != p.bm || q.ck != p.ck) multi = false;
            if (q.tps != 3) tps = 1;
            if (q.nsplit > maxsplit) maxsplit = q.nsplit;
        }
        if (multi) { p.tps = tps; p.nsplit = maxsplit; }
    }
    patch_name(p, multi, (char*)name_out_host, cap);
    return CC_OK;
}

int cc_conv2d_wgrad_kernel(int B, int M, int AH, int AW, int Cin, int IH, int IW, int R, int S, int si, int pad,
                           void* name_out_host, int cap) {
    char* out = (char*)name_out_host;
    if (R == 3 && S == 3 && si == 1 && pad == 1 && IH == AH && IW == AW && ccint::wino_wgrad_plan(B, M, AH, AW, Cin, 1).ok) {
        snprintf(out, cap, "k_wino_wgrad");
        return CC_OK;
    }
    ccint::wgrad_thin_name(B, M, AH, AW, Cin, IH, IW, R, S, si, pad, out, cap);
    if (out[0]) return CC_OK;
    const W3Plan q = plan_w3(B, M, AH, AW, Cin, R, S, si, pad, IH, IW);
    if (q.ok) { snprintf(out, cap, "k_wgrad3x3<%d, %d>", q.mt, 4 / q.mt); return CC_OK; }
    const WPlan p = plan_wgrad(B, M, AH, AW, Cin, R, S, si);
    if (p.ok) { snprintf(out, cap, "k_wgrad_patch<%d, %d>", p.bmw, p.nt); return CC_OK; }
    snprintf(out, cap, "k_wgrad<%d>", pick_bm(M));
    return CC_OK;
}

// Bias gradients of a whole backward stage in ONE launch.  With the activation derivative applied in the data-gradient epilogues
// (planned backward), the per-layer pass left over is a pure reduction of the pre-activation gradient over (B, H, W): 84 launches
// of 5-8 us per step.  The trainer parks them (the gradients stay alive for the parked weight-gradient launches anyway) and
// cc_bias_grad_table sums up to NBJ layers per launch; per-chunk partials are finished by cc_wgrad_reduce_table (kind 4), small
// maps are written directly -- block decomposition and summation order are k_act_bwd's.
constexpr int NBJ = 32;
struct BJ {
    const float* gy; float* partial; float* gbias; long gy_bs;
    int B, C, HW, cpp, single, accum, vec4, blk_end;
};
struct BT { BJ j[NBJ]; int n; };

__global__ __launch_bounds__(256) void k_bias_table(BT t) {
    __shared__ float red[4];
    int k = 0, first = 0;
#pragma unroll 1
    for (int q = 0; q + 1 < t.n; q++)
        if ((int)blockIdx.x >= t.j[q].blk_end) { k = q + 1; first = t.j[q].blk_end; }
    const BJ& j = t.j[k];
    const int bid = (int)blockIdx.x - first;
    const int per_m = j.single ? 1 : j.B * j.cpp;                 // workgroups per channel
    const int m = bid / per_m, rem = bid - m * per_m;
    const int zimg = j.single ? 0 : rem / j.cpp, chunk = j.single ? 0 : rem - zimg * j.cpp;
    const int cpp = j.single ? 1 : j.cpp, nb = j.single ? j.B : 1, HW = j.HW;
    float s[1] = {0.f};
    for (int nn = 0; nn < nb; nn++) {
        const float* __restrict__ gp = j.gy + (long)(zimg + nn) * j.gy_bs + (long)m * HW;
        if (j.vec4) {
            const int nq = HW >> 2, stp = cpp * 256;
            for (int q0 = chunk * 256 + threadIdx.x; q0 < nq; q0 += 4 * stp) {
                float4 gg[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * stp;
                    gg[u] = (q < nq) ? ((const float4*)gp)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (q0 + u * stp < nq) s[0] += (gg[u].x + gg[u].y) + (gg[u].z + gg[u].w);
            }
        } else {
            for (int e = chunk * 256 + threadIdx.x; e < HW; e += cpp * 256) s[0] += gp[e];
        }
    }
    cc::block_sum_256<1>(s, red);
    if (threadIdx.x == 0) {
        if (j.single) j.gbias[m] = j.accum ? (j.gbias[m] + s[0]) : s[0];
        else j.partial[(long)m * (j.cpp * j.B) + zimg * j.cpp + chunk] = s[0];
    }
}

size_t cc_act_bwd_ws_bytes(int C) { return (size_t)C * 64 * sizeof(float); }

/* geff = gy * act'(y) (geff may alias gy or be null), gbias[c] = sum_{n,p} geff (gbias may be null).
 * Group form: G (<= 4) same-shaped problems per launch; gy / y / geff / gbias: HOST arrays of device addresses (0 = null,
 * uniformly over the group); ws: G areas of cc_act_bwd_ws_bytes(C) each. */
static int act_bwd_bias_impl(int G, const long* gy, const long* y, const long* geff, const long* gbias, float* ws, int B, int C, int H,
                             int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b, int accumulate_bias,
                             void* stream, ccint::RedSink* sink) {
    if (G <= 0 || G > MAXGRP || B <= 0 || C <= 0) return CC_ERR_ARG;
    const bool has_y = y && y[0], has_ge = geff && geff[0], has_gb = gbias && gbias[0];
    if (act != ACT_NONE && !has_y) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    int cpp = (HW + 8191) / 8192;                       // chunks per (image, channel) plane; B * cpp <= 64 partials per channel
    const int cap = 64 / B > 0 ? 64 / B : 1;
    cpp = cpp < 1 ? 1 : (cpp > cap ? cap : cpp);
    if (B > 64) return CC_ERR_ARG;
    bool vec4 = (HW % 4 == 0) && (gy_bs % 4 == 0) && (y_bs % 4 == 0) && (geff_bs % 4 == 0);
    for (int k = 0; k < G; k++)
        vec4 = vec4 && ((((uintptr_t)gy[k]) | (uintptr_t)(has_y ? y[k] : 0) | (uintptr_t)(has_ge ? geff[k] : 0)) % 16 == 0);
    // small maps with enough channels to occupy the chip: one workgroup per channel, bias gradient written in place
    const bool single = ((long)B * HW <= 32768) && ((long)C * B * HW * G <= (1l << 22) || (long)C * G >= 128);
    const int nb = single ? B : 1;
    const size_t wstride = cc_act_bwd_ws_bytes(C) / sizeof(float);
    AB t = {};
    BR r = {};
    t.zper = single ? 1 : B;
    for (int k = 0; k < G; k++) {
        t.gy[k] = (const float*)gy[k];
        t.y[k] = has_y ? (const float*)y[k] : nullptr;
        t.geff[k] = has_ge ? (float*)geff[k] : nullptr;
        t.gbias_direct[k] = (single && has_gb) ? (float*)gbias[k] : nullptr;
        t.partial[k] = (has_gb && !single) ? ws + k * wstride : nullptr;
        r.partial[k] = t.partial[k];
        r.gbias[k] = has_gb ? (float*)gbias[k] : nullptr;
    }
    dim3 grid(single ? 1 : cpp, C, (single ? 1 : B) * G);
    const int nchunk = single ? 1 : cpp * B;
    if (vec4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_act_bwd<true>), grid, dim3(256), 0, s, t, HW, gy_bs, y_bs, geff_bs, act, act_a, act_b,
                           nb, accumulate_bias);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_act_bwd<false>), grid, dim3(256), 0, s, t, HW, gy_bs, y_bs, geff_bs, act, act_a, act_b,
                           nb, accumulate_bias);
    if (has_gb && !single) {
        if (sink) {      // second stage parked: kind 4 of the reduce table (same per-channel summation as k_bias_reduce)
            for (int k = 0; k < G; k++) {
                const long d[ccint::RD_LONGS] = {4, (long)r.partial[k], (long)r.gbias[k], nchunk, accumulate_bias, 0, 0, C};
                if (ccint::wgrad_reduce_emit(sink, d, 1, s) != CC_OK) return CC_ERR_ARG;
            }
        } else {
            hipLaunchKernelGGL(k_bias_reduce, dim3(C, G), dim3(64), 0, s, r, nchunk, accumulate_bias);
        }
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_act_bwd_bias_group(int G, const long* gy, const long* y, const long* geff, const long* gbias, float* ws, int B, int C, int H,
                          int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b, int accumulate_bias,
                          void* stream) {
    return act_bwd_bias_impl(G, gy, y, geff, gbias, ws, B, C, H, W, gy_bs, y_bs, geff_bs, act, act_a, act_b, accumulate_bias, stream,
                             nullptr);
}

/* ... with the second stage of the bias gradient (sum of the per-chunk partials in ws) left to the caller: descriptors for
 * cc_wgrad_reduce_table (16 longs each, at most G, none when the kernel wrote gbias itself) go to red_host[0 .. *nred_host);
 * ws must stay untouched until that call. */
int cc_act_bwd_bias_group_defer(int G, const long* gy, const long* y, const long* geff, const long* gbias, float* ws, int B, int C,
                                int H, int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b,
                                int accumulate_bias, long* red_host, int red_cap, int* nred_host, void* stream) {
    if (!red_host || !nred_host || red_cap < G) return CC_ERR_ARG;
    ccint::RedSink sink = {red_host, red_cap, 0};
    const int rc = act_bwd_bias_impl(G, gy, y, geff, gbias, ws, B, C, H, W, gy_bs, y_bs, geff_bs, act, act_a, act_b, accumulate_bias,
                                     stream, &sink);
    *nred_host = sink.n;
    return rc;
}

/* Bias gradient gbias[c] (+)= sum_{n,h,w} gy[n,c,h,w], parked: nothing is launched.  job_host[12] receives the job for
 * cc_bias_grad_table; when the map is large enough to be summed in chunks, red_host[16] receives the descriptor of the second
 * stage for cc_wgrad_reduce_table and *nred_host = 1 (else 0).  ws: cc_act_bwd_ws_bytes(C) bytes, untouched until both ran. */
int cc_bias_grad_defer(const float* gy, float* gbias, float* ws, int B, int C, int H, int W, long gy_bs, int accumulate,
                       long* job_host, long* red_host, int* nred_host) {
    if (!gy || !gbias || !job_host || !red_host || !nred_host || B <= 0 || B > 64 || C <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    const int HW = H * W;
    int cpp = (HW + 8191) / 8192;                       // as act_bwd_bias_impl
    const int cap = 64 / B > 0 ? 64 / B : 1;
    cpp = cpp < 1 ? 1 : (cpp > cap ? cap : cpp);
    const bool vec4 = (HW % 4 == 0) && (gy_bs % 4 == 0) && (((uintptr_t)gy) % 16 == 0);
    // one workgroup per channel (no second stage) only where it walks <= 4096 elements: in a table launch the longest job
    // sets the duration (k_act_bwd's own threshold is 32768: there a second launch would cost more than the walk)
    const bool single = (long)B * HW <= 4096;
    if (!single && !ws) return CC_ERR_ARG;
    const long job[12] = {(long)gy, single ? 0 : (long)ws, (long)gbias, gy_bs, B, C, HW, cpp, single ? 1 : 0, accumulate ? 1 : 0,
                          vec4 ? 1 : 0, 0};
    for (int i = 0; i < 12; i++) job_host[i] = job[i];
    *nred_host = 0;
    if (!single) {
        const long d[ccint::RD_LONGS] = {4, (long)ws, (long)gbias, (long)cpp * B, accumulate ? 1 : 0, 0, 0, C};
        for (int i = 0; i < ccint::RD_LONGS; i++) red_host[i] = d[i];
        *nred_host = 1;
    }
    return CC_OK;
}

/* Run n parked bias-gradient jobs (12 longs each, from cc_bias_grad_defer): one launch per 32. */
int cc_bias_grad_table(const long* jobs_host, int n, void* stream) {
    if (!jobs_host || n <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int j0 = 0; j0 < n; j0 += NBJ) {
        BT t = {};
        int bx = 0;
        t.n = n - j0 < NBJ ? n - j0 : NBJ;
        for (int k = 0; k < t.n; k++) {
            const long* d = jobs_host + (long)(j0 + k) * 12;
            BJ& j = t.j[k];
            j.gy = (const float*)d[0]; j.partial = (float*)d[1]; j.gbias = (float*)d[2]; j.gy_bs = d[3];
            j.B = (int)d[4]; j.C = (int)d[5]; j.HW = (int)d[6]; j.cpp = (int)d[7]; j.single = (int)d[8]; j.accum = (int)d[9];
            j.vec4 = (int)d[10];
            if (!j.gy || !j.gbias || j.B <= 0 || j.C <= 0 || j.HW <= 0 || j.cpp <= 0 || (!j.single && !j.partial)) return CC_ERR_ARG;
            bx += j.single ? j.C : j.C * j.B * j.cpp;
            j.blk_end = bx;
        }
        hipLaunchKernelGGL(k_bias_table, dim3((unsigned)bx), dim3(256), 0, s, t);
    }
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_act_bwd_bias(const float* gy, const float* y_or_null, float* geff_or_null, float* gbias_or_null, float* ws, int B,
                    int C, int H, int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b,
                    int accumulate_bias, void* stream) {
    const long a = (long)gy, b = (long)y_or_null, c = (long)geff_or_null, d = (long)gbias_or_null;
    return cc_act_bwd_bias_group(1, &a, &b, &c, &d, ws, B, C, H, W, gy_bs, y_bs, geff_bs, act, act_a, act_b, accumulate_bias, stream);
}

}  // extern "C"
