// Weight gradient of the "thin" layers: <= 32 channels on either side but 50 k - 850 k pixels per call (the full- and
// half-resolution layers of DispResNet6 / MaskNet6 / Back2Future: first convs, last iconvs, prediction heads).
//
//      gw[m, c, r, s] = sum_{n, y, x}  a[n, m, y, x] * X[n, c, SI*y + r - PAD, SI*x + s - PAD]
//
// These calls are HBM-shaped (16-32 channel planes of up to 213 k pixels read once, 2-5 k outputs), and an
// im2col-style GEMM spends its time gathering: every X element is fetched R*S times with per-element address
// arithmetic.  Here the reduction (pixel) axis is the MFMA K axis and NOTHING is staged:
//   * v_mfma_f32_16x16x4_f32:  D[m][c] += sum_{k<4} A[m][k] * B[k][c]  with k = 4 consecutive output pixels;
//     lane (i = lane & 15, k = lane >> 4) supplies a[m = i][pixel k] and X[c = i][pixel k shifted by the tap];
//   * one "unit" = 16 consecutive output pixels of one row: lane k owns pixels x0 + 4k .. x0 + 4k + 3, so its A operand
//     for the 4 MFMA steps is ONE aligned float4 of the dY row and its B operands for all S taps of a tap row are a
//     window of NQ aligned float4 of the X row (global_load_dwordx4 straight into the MFMA source registers);
//   * every tap (r, s) has its own 16x16 accumulator (4 VGPRs): 9 taps = 36 VGPRs, the window index of (step j, tap s)
//     is a compile-time constant, so the inner loop is loads + v_cndmask (zero padding) + MFMAs only;
//   * the loads of the next unit are issued before the MFMAs of the current one (register double buffer), addresses are clamped
//     instead of predicated so that no branch sits between loads and MFMAs;
//   * round 4: the four waves of a workgroup walk DOWN a strip of four adjacent columns, so the window rows a unit shares with the
//     unit above stay in registers (4 loads per unit instead of 10 for 3x3 / stride 1: -11 % on that kernel; with the loads or
//     the MFMAs ablated it still runs at 80 % of its time -- what is left is the masks, the address arithmetic and the
//     accumulator round trips of a one-wave-per-SIMD kernel, profiles/r04_ab_round4.txt).
// Work split: blockIdx.x = a contiguous range of strip rows, blockIdx.y = (16-channel group of
// a) x (16-channel group of X) x (group of TR tap rows).  Each workgroup reduces its 4 waves in LDS (fixed order) and
// writes one partial slab; k_wgrad_thin_reduce sums the slabs in a fixed order (deterministic, no atomics).
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>
#include "cc_common.h"
#include "conv_internal.h"
#include "cc_tools.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// waves per workgroup.  Measured on the 16->16 3x3 256x832 layer (tools/gpu_thin_ablate.sh): full kernel 73 us, no-MFMA build
// 69 us, no-load build 45 us -> bound by its load path (253 MB L1->L2 requests for 109 MB of unique data: every X row is
// requested once per tap row); 8 waves per workgroup, more/smaller workgroups and the XCD swizzle below all left it unchanged.
constexpr int THIN_NW = 4;

struct WT {
    const float* a; const float* x; float* ws;
    int B, M, AH, AW; long a_bs;
    int Cin, IH, IW; long x_bs;
    int R, nxc, units, upb, npb, ngc, ngt, swz;
};

template <int S, int SI, int PAD, int TR>
struct ThinCfg {
    static constexpr int PQ = (PAD + 3) / 4;                         // float4s to the left of the unit's first pixel
    static constexpr int IMAX = SI * 3 + S - 1 - PAD + 4 * PQ;       // largest window index used
    static constexpr int NQ = IMAX / 4 + 1;                          // float4s per tap row window
    static constexpr int TS = TR * S;
};

// DBG (ablation builds of the 3x3/s1 kernel only, CC_WGRAD_THIN_DBG): 1 = no MFMAs (loads + masks), 2 = no loads after the
// first unit (MFMAs + masks on stale registers); results are wrong by construction.
template <int S, int SI, int PAD, int TR, int DBG>
__device__ __forceinline__ void wgrad_thin_body(const WT& g, const int bx_, const int by_) {
    typedef ThinCfg<S, SI, PAD, TR> C;
    constexpr int PQ = C::PQ, NQ = C::NQ, TS = C::TS;
    __shared__ float red[TS * 256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, k = lane >> 4;
    int combo = by_;
    const int tg = combo % g.ngt;
    combo /= g.ngt;
    const int cg = combo % g.ngc, mg = combo / g.ngc;
    const int r0 = tg * TR;
    const int m = mg * 16 + i, c = cg * 16 + i;
    const bool mok = m < g.M, cok = c < g.Cin;
    const float* __restrict__ abase = g.a + (long)(mok ? m : 0) * g.AH * g.AW;
    const float* __restrict__ xbase = g.x + (long)(cok ? c : 0) * g.IH * g.IW;

    f32x4 acc[TR][S];
#pragma unroll
    for (int tr = 0; tr < TR; tr++)
#pragma unroll
        for (int s = 0; s < S; s++) acc[tr][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // XCD-aware range assignment: workgroups are dealt round-robin to the 8 XCDs (private L2 each); give XCD k the k-th
    // CONTIGUOUS eighth of the unit ranges so that the halo rows two neighbouring ranges share are fetched by ONE L2
    // (npb is a multiple of 8; the range index bx_ % 8 is the XCD for every combination when the launch is this problem's alone)
    const int pb = (g.swz ? ((bx_ & 7) * (g.npb >> 3) + (bx_ >> 3)) : bx_);
    const int u0 = pb * g.upb;
    const int u1 = (u0 + g.upb < g.units) ? (u0 + g.upb) : g.units;

    // The workgroup walks DOWN a strip of four adjacent 16-pixel columns, one column per wave, u = (b * ngx + strip) * AH + y:
    // consecutive units of a wave share TR - SI of their TR window rows, which stay in registers (a rolling window) -- 1 + SI * NQ
    // loads per unit instead of 1 + TR * NQ (3x3 / stride 1: 4 instead of 10).  The kernel was bound by its load path (header: the
    // no-MFMA build ran at 95 % of the full one); every load instruction costs the issuing wave 100+ cycles whatever its latency.
    // The four waves read neighbouring 64-byte pieces of the same rows at the same time (one wave walking a column alone touched
    // half a cache line per row and ran 3.6x slower: profiles/r04_ab_round4.txt).
    constexpr int KEEP = (TR > SI) ? TR - SI : 0;            // window rows carried over to the next unit of the column
    float4 An;
    float4 Wn[TR][NQ], Wc[TR][NQ];
    // loads of unit (b, y, xc), window rows t >= t0 (clamped addresses: always in bounds, the padding mask is applied at use)
    auto issue = [&](auto T0, int b, int y, int xc, float4& A, float4 (&W)[TR][NQ]) {
        constexpr int t0 = decltype(T0)::value;
        const int x0 = xc * 16 + 4 * k;
        const int xa = (x0 < g.AW) ? x0 : 0;
        // 32-bit element offsets (the host side only takes this path for tensors below 2^31 elements)
        A = *(const float4*)(abase + (unsigned)(b * (int)g.a_bs + y * g.AW + xa));
        const unsigned xb = (unsigned)(b * (int)g.x_bs);
#pragma unroll
        for (int tr = t0; tr < TR; tr++) {
            int iy = y * SI + r0 + tr - PAD;
            iy = iy < 0 ? 0 : (iy >= g.IH ? g.IH - 1 : iy);
            const unsigned xr = xb + (unsigned)(iy * g.IW);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                int col = SI * x0 - 4 * PQ + 4 * q;
                col = (col < 0 || col >= g.IW) ? 0 : col;
                W[tr][q] = *(const float4*)(xbase + (xr + (unsigned)col));
            }
        }
    };

    const int ngx = (g.nxc + THIN_NW - 1) / THIN_NW;          // strips per image row (a strip's surplus columns multiply zeros)
    int u = u0;
    const int ue = u1;
    if (u < ue) {
        // unit position: decoded once, then advanced incrementally on the scalar unit
        int nb, ny, nxcpos;
        {
            const int colu = u / g.AH;
            ny = u - colu * g.AH;
            nb = colu / ngx;
            nxcpos = (colu - nb * ngx) * THIN_NW + wave;
        }
        typedef std::integral_constant<int, 0> ALL;
        typedef std::integral_constant<int, KEEP> NEW;
        issue(ALL(), nb, ny, nxcpos, An, Wn);
        bool fresh = true;                                    // the next-unit buffer holds a whole window (run / column start)
        for (; u < ue; u++) {
            const float4 Ac = An;
            if (fresh) {
#pragma unroll
                for (int tr = 0; tr < TR; tr++)
#pragma unroll
                    for (int q = 0; q < NQ; q++) Wc[tr][q] = Wn[tr][q];
            } else {
#pragma unroll
                for (int tr = 0; tr < TR; tr++)
#pragma unroll
                    for (int q = 0; q < NQ; q++) Wc[tr][q] = (tr < KEEP) ? Wc[tr + (TR - KEEP)][q] : Wn[tr][q];
            }
            const int y = ny, xc = nxcpos;
            if (u + 1 < ue) {
                fresh = false;
                if (++ny == g.AH) {
                    ny = 0;
                    fresh = true;
                    nxcpos += THIN_NW;
                    if (nxcpos - wave >= g.nxc) { nxcpos = wave; nb++; }
                }
                if (DBG != 2) {
                    if (fresh) issue(ALL(), nb, ny, nxcpos, An, Wn);
                    else issue(NEW(), nb, ny, nxcpos, An, Wn);
                }
            }

            const int x0 = xc * 16 + 4 * k;
            const bool aok = mok && (x0 < g.AW);
            float a4[4];
            a4[0] = aok ? Ac.x : 0.f;
            a4[1] = aok ? Ac.y : 0.f;
            a4[2] = aok ? Ac.z : 0.f;
            a4[3] = aok ? Ac.w : 0.f;
#pragma unroll
            for (int tr = 0; tr < TR; tr++) {
                // a short last tap-row group (r0 + tr >= R) multiplies zeros: no branch between the MFMAs
                const int iy = y * SI + r0 + tr - PAD;
                const bool rok = cok && (r0 + tr < g.R) && iy >= 0 && iy < g.IH;
                float w[NQ * 4];
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int col = SI * x0 - 4 * PQ + 4 * q;
                    const bool ok = rok && col >= 0 && col < g.IW;
                    w[4 * q + 0] = ok ? Wc[tr][q].x : 0.f;
                    w[4 * q + 1] = ok ? Wc[tr][q].y : 0.f;
                    w[4 * q + 2] = ok ? Wc[tr][q].z : 0.f;
                    w[4 * q + 3] = ok ? Wc[tr][q].w : 0.f;
                }
                if (DBG == 1) {
#pragma unroll
                    for (int s = 0; s < S; s++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[tr][s][j] += a4[j] * w[SI * j + s - PAD + 4 * PQ];
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int s = 0; s < S; s++)
                        acc[tr][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], w[SI * j + s - PAD + 4 * PQ], acc[tr][s], 0, 0, 0);
            }
        }
    }

    // workgroup reduction, waves in a fixed order: red[(t*4 + reg)*64 + lane]
#pragma unroll 1
    for (int wv = 0; wv < THIN_NW; wv++) {
        if (wave == wv) {
#pragma unroll
            for (int tr = 0; tr < TR; tr++)
#pragma unroll
                for (int s = 0; s < S; s++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int idx = ((tr * S + s) * 4 + r) * 64 + lane;
                        const float v = acc[tr][s][r];
                        red[idx] = (wv == 0) ? v : (red[idx] + v);
                    }
        }
        __syncthreads();
    }
    // slab [t][m16][c16]; D element (reg, lane) is m = 4*(lane>>4) + reg, c = lane & 15
    float* __restrict__ slab = g.ws + ((long)by_ * g.npb + pb) * (TS * 256);
    for (int e = threadIdx.x; e < TS * 256; e += 64 * THIN_NW) {
        const int t = e >> 8, mc = e & 255, mm = mc >> 4, cc_ = mc & 15;
        slab[e] = red[(t * 4 + (mm & 3)) * 64 + (mm >> 2) * 16 + cc_];
    }
}


template <int S, int SI, int PAD, int TR, int DBG = 0>
__global__ __launch_bounds__(64 * THIN_NW) void k_wgrad_thin(WT g) {
    wgrad_thin_body<S, SI, PAD, TR, DBG>(g, (int)blockIdx.x, (int)blockIdx.y);
}

// Problems of DIFFERENT shapes (one kernel instance: same tap geometry) in one launch: the thin weight gradients a backward stage
// leaves parked until its end (cc_conv2d_wgrad_list).  A problem's own grid is 64-256 workgroups of four waves per 16 x 16 channel
// combination -- ONE wave per SIMD for the 16 -> 16 layers -- and the kernel is bound by the latency of its load path (header): launched
// one after the other the problems leave most of the chip's wave slots empty; side by side they fill them.  blockIdx.x ranges over the
// problems' grids back to back (a problem's grid flattened range-fastest, so that the XCD order of its ranges survives whenever the
// grids in front of it are multiples of 8 -- it is a bijection of the ranges either way).
constexpr int THIN_MAXP = 24;
struct WTM { WT p[THIN_MAXP]; int blk_end[THIN_MAXP]; int n; };
template <int S, int SI, int PAD, int TR>
__global__ __launch_bounds__(64 * THIN_NW) void k_wgrad_thin_multi(WTM a) {
    int k = 0, first = 0;
#pragma unroll 1
    for (int q = 0; q + 1 < a.n; q++)
        if ((int)blockIdx.x >= a.blk_end[q]) { k = q + 1; first = a.blk_end[q]; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    const WT& g = *(reinterpret_cast<const WT*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WTM, p)) + k);
#else
    const WT& g = a.p[k];
#endif
    const int b = (int)blockIdx.x - first;
    const int by = b / g.npb;
    wgrad_thin_body<S, SI, PAD, TR, 0>(g, b - by * g.npb, by);
}


int env_int(const char* name, int dflt) { return cctools::env_int(name, dflt); }

struct ThinPlan {
    bool ok;
    int kind, TR, ngm, ngc, ngt, nxc, units, upb, npb, TS;
    size_t ws_floats;
};

// kinds: (S, SI, PAD, TR) instantiations
enum { K_3_1 = 0, K_3_2, K_7_2, K_7_1, K_1_1, K_5_2, K_4_2, K_1_2, K_NONE };

ThinPlan plan_thin(int B, int M, int AH, int AW, int Cin, int R, int S, int si) {
    ThinPlan p = {};
    p.kind = K_NONE;
    if (R != S || env_int("CC_NO_WGRAD_THIN", 0)) return p;
    if (S == 3 && si == 1) { p.kind = K_3_1; p.TR = 3; }
    else if (S == 3 && si == 2) { p.kind = K_3_2; p.TR = 3; }
    else if (S == 7 && si == 2) { p.kind = K_7_2; p.TR = 2; }
    else if (S == 7 && si == 1) { p.kind = K_7_1; p.TR = 2; }
    else if (S == 1 && si == 1) { p.kind = K_1_1; p.TR = 1; }
    else if (S == 5 && si == 2) { p.kind = K_5_2; p.TR = 3; }
    else if (S == 4 && si == 2) { p.kind = K_4_2; p.TR = 4; }
    else if (S == 1 && si == 2) { p.kind = K_1_2; p.TR = 1; }
    else return p;
    p.ngm = (M + 15) / 16;
    p.ngc = (Cin + 15) / 16;
    const long P = (long)B * AH * AW;
    // <= 15 (M, Cin) 16-channel combinations: the 64 x 64 layers (16) run faster on the 3x3 kernel's <2, 2> tiles (r3o A/B)
    if ((AW & 3) || p.ngm * p.ngc > env_int("CC_WGRAD_THIN_MAXCOMBO", 15) || P < env_int("CC_WGRAD_THIN_MINPIX", 8192) ||
        (S == 7 && Cin < 8)) return p;
    p.ngt = (R + p.TR - 1) / p.TR;
    p.TS = p.TR * S;
    p.nxc = (AW + 15) / 16;
    const long units = (long)B * AH * ((p.nxc + THIN_NW - 1) / THIN_NW);       // rows of four-column strips (one column per wave)
    if (units > (1l << 30)) return p;
    p.units = (int)units;
    long npb = units / env_int("CC_WGRAD_THIN_UPB", 8);
    const long cap = env_int("CC_WGRAD_THIN_NPB", 256);      // 256 (one round of the 256 CUs): -0.08 ms/step against 512 (r3s3 A/Bs)
    npb = npb < 1 ? 1 : (npb > cap ? cap : npb);
    p.upb = (int)((units + npb - 1) / npb);
    p.npb = (p.units + p.upb - 1) / p.upb;
    if (p.npb >= 64) p.npb = ((p.npb + 7) / 8) * 8;       // XCD swizzle needs a multiple of 8 (surplus ranges are empty)
    p.ws_floats = (size_t)p.ngm * p.ngc * p.ngt * p.npb * p.TS * 256;
    p.ok = true;
    return p;
}

template <int S, int SI, int PAD, int TR>
void launch_thin(const WT& g, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_thin<S, SI, PAD, TR>), grid, dim3(64 * THIN_NW), 0, s, g);
}

template <int S, int SI, int PAD, int TR>
void launch_thin_multi(const WTM& m, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_thin_multi<S, SI, PAD, TR>), dim3(blocks), dim3(64 * THIN_NW), 0, s, m);
}

// problems parked by wgrad_thin_launch while wgrad_thin_park(true) is in force (this thread), launched by wgrad_thin_flush
constexpr int THIN_PARK_CAP = 96;
struct ThinParkedOne { WT g; int kind, ncombo; double gflop; };
struct ThinParked { bool on; int n; ThinParkedOne d[THIN_PARK_CAP]; };
thread_local ThinParked g_thin_parked = {};

void launch_thin_kind(int kind, const WT& g, dim3 grid, hipStream_t s) {
    switch (kind) {
        case K_3_1: launch_thin<3, 1, 1, 3>(g, grid, s); break;
        case K_3_2: launch_thin<3, 2, 1, 3>(g, grid, s); break;
        case K_7_2: launch_thin<7, 2, 3, 2>(g, grid, s); break;
        case K_7_1: launch_thin<7, 1, 3, 2>(g, grid, s); break;
        case K_1_1: launch_thin<1, 1, 0, 1>(g, grid, s); break;
        case K_5_2: launch_thin<5, 2, 2, 3>(g, grid, s); break;
        case K_1_2: launch_thin<1, 2, 0, 1>(g, grid, s); break;
        default: launch_thin<4, 2, 1, 4>(g, grid, s); break;
    }
}

void launch_thin_multi_kind(int kind, const WTM& m, unsigned blocks, hipStream_t s) {
    switch (kind) {
        case K_3_1: launch_thin_multi<3, 1, 1, 3>(m, blocks, s); break;
        case K_3_2: launch_thin_multi<3, 2, 1, 3>(m, blocks, s); break;
        case K_7_2: launch_thin_multi<7, 2, 3, 2>(m, blocks, s); break;
        case K_7_1: launch_thin_multi<7, 1, 3, 2>(m, blocks, s); break;
        case K_1_1: launch_thin_multi<1, 1, 0, 1>(m, blocks, s); break;
        case K_5_2: launch_thin_multi<5, 2, 2, 3>(m, blocks, s); break;
        case K_1_2: launch_thin_multi<1, 2, 0, 1>(m, blocks, s); break;
        default: launch_thin_multi<4, 2, 1, 4>(m, blocks, s); break;
    }
}

}  // namespace

namespace ccint {

bool wgrad_thin_park(bool on) {
    const bool was = g_thin_parked.on;
    g_thin_parked.on = on && !env_int("CC_NO_WGRAD_THIN_LIST", 0);
    return was;
}

bool wgrad_thin_parking() { return g_thin_parked.on; }

double wgrad_thin_parked_gflop() {
    double gf = 0;
    for (int i = 0; i < g_thin_parked.n; i++) gf += g_thin_parked.d[i].gflop;
    return gf;
}

// launch what is parked: per kernel instance, the problems in the order they were parked, up to THIN_MAXP per launch (a single
// problem goes to the plain kernel)
void wgrad_thin_flush(hipStream_t s) {
    ThinParked& P = g_thin_parked;
    for (int kind = 0; kind < K_NONE && P.n > 0; kind++) {
        int i = 0;
        while (i < P.n) {
            WTM m = {};
            long blk = 0;
            int last = -1;
            for (; i < P.n && m.n < THIN_MAXP; i++) {
                if (P.d[i].kind != kind) continue;
                const long nb = (long)P.d[i].g.npb * P.d[i].ncombo;
                if (m.n && blk + nb >= (1l << 31)) break;
                m.p[m.n] = P.d[i].g;
                blk += nb;
                m.blk_end[m.n] = (int)blk;
                m.n++;
                last = i;
            }
            if (!m.n) break;
            if (env_int("CC_WGRAD_THIN_TRACE", 0)) fprintf(stderr, "[wgrad_thin_multi] kind %d: %d problems, %ld workgroups\n", kind, m.n, blk);
            if (m.n == 1) launch_thin_kind(kind, P.d[last].g, dim3((unsigned)P.d[last].g.npb, (unsigned)P.d[last].ncombo), s);
            else launch_thin_multi_kind(kind, m, (unsigned)blk, s);
        }
    }
    P.n = 0;
}

size_t wgrad_thin_ws_floats(int B, int M, int AH, int AW, int Cin, int R, int S, int si) {
    const ThinPlan p = plan_thin(B, M, AH, AW, Cin, R, S, si);
    return p.ok ? p.ws_floats : 0;
}

bool wgrad_thin_launch(const float* a, const float* x, float* gw, float* ws, int B, int M, int AH, int AW, long a_bs, int Cin,
                       int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate, hipStream_t s,
                       RedSink* sink) {
    const ThinPlan p = plan_thin(B, M, AH, AW, Cin, R, S, si);
    if (!p.ok || (IW & 3) || (a_bs & 3) || (x_bs & 3) || (((uintptr_t)a | (uintptr_t)x) & 15)) return false;
    if ((long)B * a_bs >= (1l << 31) || (long)B * x_bs >= (1l << 31)) return false;      // 32-bit offsets in the kernel
    static const int want_pad[] = {1, 1, 3, 3, 0, 2, 1, 0};
    if (pad != want_pad[p.kind]) return false;
    WT g = {};
    g.a = a; g.x = x; g.ws = ws;
    g.B = B; g.M = M; g.AH = AH; g.AW = AW; g.a_bs = a_bs; g.Cin = Cin; g.IH = IH; g.IW = IW; g.x_bs = x_bs;
    g.R = R; g.nxc = p.nxc; g.units = p.units; g.upb = p.upb; g.npb = p.npb; g.ngc = p.ngc; g.ngt = p.ngt;
    g.swz = (p.npb % 8 == 0 && p.npb >= 64 && !env_int("CC_WGRAD_THIN_NOSWZ", 0)) ? 1 : 0;
    const int ncombo = p.ngm * p.ngc * p.ngt;
    if (env_int("CC_WGRAD_THIN_TRACE", 0))
        fprintf(stderr, "[wgrad_thin] kind %d M %d Cin %d A %dx%d X %dx%d npb %d upb %d combos %d\n", p.kind, M, Cin, AH, AW, IH, IW,
                p.npb, p.upb, ncombo);
    dim3 grid((unsigned)p.npb, (unsigned)ncombo);
    // (parking needs a DEFERRED reduce sink: an immediate one would launch the reduction in front of the parked kernel)
    if (g_thin_parked.on && sink && g_thin_parked.n < THIN_PARK_CAP) {
        ThinParkedOne& d = g_thin_parked.d[g_thin_parked.n++];
        d.g = g; d.kind = p.kind; d.ncombo = ncombo;
        d.gflop = 2e-9 * B * AH * AW * (double)M * Cin * R * S;
        const long rdp[RD_LONGS] = {2, (long)ws, (long)gw, p.npb, accumulate, o_sm, o_sc, p.TS, S, p.TR, p.ngc, p.ngt, M, Cin, R, ncombo};
        (void)wgrad_reduce_emit(sink, rdp, 1, s);
        return true;
    }
    switch (p.kind) {
        case K_3_1: {
            const int dbg = env_int("CC_WGRAD_THIN_DBG", 0);
            if (dbg == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_thin<3, 1, 1, 3, 1>), grid, dim3(64 * THIN_NW), 0, s, g);
            else if (dbg == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_thin<3, 1, 1, 3, 2>), grid, dim3(64 * THIN_NW), 0, s, g);
            else launch_thin<3, 1, 1, 3>(g, grid, s);
            break;
        }
        case K_3_2: launch_thin<3, 2, 1, 3>(g, grid, s); break;
        case K_7_2: launch_thin<7, 2, 3, 2>(g, grid, s); break;
        case K_7_1: launch_thin<7, 1, 3, 2>(g, grid, s); break;
        case K_1_1: launch_thin<1, 1, 0, 1>(g, grid, s); break;
        case K_5_2: launch_thin<5, 2, 2, 3>(g, grid, s); break;
        case K_1_2: launch_thin<1, 2, 0, 1>(g, grid, s); break;
        default: launch_thin<4, 2, 1, 4>(g, grid, s); break;
    }
    const long rd[RD_LONGS] = {2, (long)ws, (long)gw, p.npb, accumulate, o_sm, o_sc, p.TS, S, p.TR, p.ngc, p.ngt, M, Cin, R, ncombo};
    (void)wgrad_reduce_emit(sink, rd, 1, s);
    return true;
}

void wgrad_thin_name(int B, int M, int AH, int AW, int Cin, int IH, int IW, int R, int S, int si, int pad, char* out, int cap) {
    out[0] = 0;
    const ThinPlan p = plan_thin(B, M, AH, AW, Cin, R, S, si);
    static const int want_pad[] = {1, 1, 3, 3, 0, 2, 1, 0};
    static const int tr[] = {3, 3, 2, 2, 1, 3, 4, 1};
    if (!p.ok || (IW & 3) || pad != want_pad[p.kind]) return;
    snprintf(out, cap, "k_wgrad_thin<%d, %d, %d, %d>", S, si, pad, tr[p.kind]);
}

}  // namespace ccint
