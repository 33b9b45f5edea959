// Second stage of every split weight-gradient kernel (k_wgrad, k_wgrad3x3 / k_wgrad_patch, k_wgrad_thin): the partial slabs of up
// to NRD problems are summed -- in a fixed order, no atomics -- by ONE launch.  The slabs of a layer are 4-50 MB and its
// reduction alone is a 8-15 us launch at ~1.4 TB/s (latency, not bandwidth: 9-300 workgroups); the weight gradients are not
// needed before the optimizer step, so the trainer parks the descriptors (cc_conv2d_wgrad_group_defer) and reduces a whole
// backward stage at once (cc_wgrad_reduce_table): the same bytes in a grid that fills the chip.
// Descriptor (RD_LONGS longs, host): kind, ws, gw, nsplit, accumulate, o_sm, o_sc, p0..p8
//   kind 0 (k_wgrad):       gw[m*o_sm + c*o_sc + i*p4 + j*p5] (+)= sum_z ws[z][m][(c,i,j)]        p = M, Ntot, RS, St, o_ri, o_sj
//                           (p6 is set at launch: dense destination + aligned slabs -> float4 path)
//   kind 1 (k_wgrad3x3):    gw[m*o_sm + c*o_sc + t]           (+)= sum_z ws[z][t][m][c]            p = T, M, Cin, Cp32
//   kind 2 (k_wgrad_thin):  gw[m*o_sm + c*o_sc + r*S + s]     (+)= sum_pb slab[combo][pb][t][m16][c16]
//                                                                                  p = TS, S, TR, ngc, ngt, M, Cin, R, ncombo
//   kind 4 (bias gradient, second stage of k_act_bwd): gw[m] (+)= sum_k ws[m*nsplit + k]                           p = C
#include "cc_common.h"
#include "conv_internal.h"
#include "../../include/ccengine.h"

namespace {

constexpr int NRD = 32;

struct RD {
    const float* ws; float* gw;
    long o_sm, o_sc;
    int kind, nsplit, accum;
    int p[9];
    int blk_end;
};
struct RT { RD d[NRD]; int n; };

__device__ __forceinline__ void put(float* o, float s, int accum) { *o = accum ? (*o + s) : s; }

__global__ __launch_bounds__(256) void k_wgrad_reduce_table(RT t) {
    __shared__ float4 part[16][16];
    int k = 0, first = 0;
#pragma unroll 1
    for (int q = 0; q + 1 < t.n; q++)
        if ((int)blockIdx.x >= t.d[q].blk_end) { k = q + 1; first = t.d[q].blk_end; }
    const RD& d = t.d[k];
    const int bid = (int)blockIdx.x - first;
    const float* __restrict__ ws = d.ws;
    if (d.kind == 0) {
        const int M = d.p[0], Ntot = d.p[1], RS = d.p[2], St = d.p[3];
        const long tot = (long)M * Ntot;
        if (d.p[6]) {
            // dense [M][C][R][S] destination (offset == e) and 16-byte aligned slabs: four elements per work-item
            const long e = ((long)bid * 256 + threadIdx.x) * 4;
            if (e >= tot) return;
            // eight slab loads in flight, added in slab order (the launch is as long as its longest dependent chain)
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int z = 0; z < d.nsplit; z += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    v[u] = (z + u < d.nsplit) ? *(const float4*)(ws + (long)(z + u) * tot + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (z + u < d.nsplit) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            float4* o = (float4*)(d.gw + e);
            if (d.accum) { const float4 g = *o; s.x = g.x + s.x; s.y = g.y + s.y; s.z = g.z + s.z; s.w = g.w + s.w; }
            *o = s;
            return;
        }
        const long e = (long)bid * 256 + threadIdx.x;
        if (e >= tot) return;
        float s = 0.f;
        for (int z = 0; z < d.nsplit; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (z + u < d.nsplit) ? ws[(long)(z + u) * tot + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (z + u < d.nsplit) s += v[u];
        }
        const int m = (int)(e / Ntot), jn = (int)(e - (long)m * Ntot);
        const int c = jn / RS, rem = jn - c * RS;
        const int i = rem / St, j = rem - i * St;
        put(d.gw + (long)m * d.o_sm + (long)c * d.o_sc + i * d.p[4] + j * d.p[5], s, d.accum);
    } else if (d.kind == 1) {
        // one work-item per (m, c): its T taps are T coalesced slab reads and ONE run of T consecutive floats in gw (a work-item
        // per slab element writes every 36-byte run from 9 different workgroups)
        const int T = d.p[0], M = d.p[1], Cin = d.p[2], Cp32 = d.p[3];
        const long mc = (long)bid * 256 + threadIdx.x;     // over [m][c]
        const long per = (long)M * Cp32;
        if (mc >= per) return;
        const int c = (int)(mc % Cp32), m = (int)(mc / Cp32);
        if (c >= Cin) return;
        const long tot = (long)T * per;
        float* o = d.gw + (long)m * d.o_sm + (long)c * d.o_sc;
        if (T == 9) {
            // the nine taps side by side: nine independent loads per slab instead of nine dependent passes over the slabs
            float s9[9];
#pragma unroll
            for (int tt = 0; tt < 9; tt++) s9[tt] = 0.f;
            int z = 0;
            for (; z + 2 <= d.nsplit; z += 2) {
                const float* wz = ws + (long)z * tot + mc;
                float v0[9], v1[9];
#pragma unroll
                for (int tt = 0; tt < 9; tt++) { v0[tt] = wz[(long)tt * per]; v1[tt] = wz[tot + (long)tt * per]; }
#pragma unroll
                for (int tt = 0; tt < 9; tt++) { s9[tt] += v0[tt]; s9[tt] += v1[tt]; }
            }
            for (; z < d.nsplit; z++) {
                const float* wz = ws + (long)z * tot + mc;
#pragma unroll
                for (int tt = 0; tt < 9; tt++) s9[tt] += wz[(long)tt * per];
            }
#pragma unroll
            for (int tt = 0; tt < 9; tt++) put(o + tt, s9[tt], d.accum);
        } else {
            for (int tt = 0; tt < T; tt++) {
                float s = 0.f;
                for (int z = 0; z < d.nsplit; z++) s += ws[(long)z * tot + (long)tt * per + mc];
                put(o + tt, s, d.accum);
            }
        }
    } else if (d.kind == 4) {
        // four channels per workgroup, one wave each: the summation of k_bias_reduce
        const int m = bid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (m < d.p[0]) {
            float s = 0.f;
            for (int k = lane; k < d.nsplit; k += 64) s += ws[(long)m * d.nsplit + k];
            s = cc::wave_sum(s);
            if (lane == 0) put(d.gw + m, s, d.accum);
        }
    } else {
        // one workgroup = a quarter (16 float4) of one [m16][c16] slab position; 16 sub-groups stride over the npb slabs, then
        // sub-group 0 adds the 16 partial sums in order
        const int TS = d.p[0], S = d.p[1], TR = d.p[2], ngc = d.p[3], ngt = d.p[4], M = d.p[5], Cin = d.p[6], R = d.p[7];
        const int npb = d.nsplit;
        const int quarter = bid & 3, ct = bid >> 2;
        const int combo = ct / TS, tt = ct - combo * TS;
        const int sub = threadIdx.x >> 4, q = quarter * 16 + (threadIdx.x & 15);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int pb = sub;
        for (; pb + 112 < npb; pb += 128) {        // eight slab loads in flight, added in slab order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *(const float4*)(ws + (((long)combo * npb + pb + 16 * u) * TS + tt) * 256 + 4 * q);
#pragma unroll
            for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; pb < npb; pb += 16) {
            const float4 v = *(const float4*)(ws + (((long)combo * npb + pb) * TS + tt) * 256 + 4 * q);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        part[sub][threadIdx.x & 15] = s;
        __syncthreads();
        if (sub == 0) {
            for (int z = 1; z < 16; z++) {
                const float4 v = part[z][threadIdx.x & 15];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            int cb = combo;
            const int tg = cb % ngt;
            cb /= ngt;
            const int cg = cb % ngc, mg = cb / ngc;
            const int r = tg * TR + tt / S, sc = tt % S;
            const int m = mg * 16 + (q >> 2), c0 = cg * 16 + (q & 3) * 4;
            if (m < M && r < R) {
                float* o = d.gw + (long)m * d.o_sm + (long)r * S + sc;
                const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (c0 + e < Cin) put(o + (long)(c0 + e) * d.o_sc, v[e], d.accum);
            }
        }
    }
}

long blocks_of(const long* h) {
    switch ((int)h[0]) {
        case 0: return h[13] ? (h[7] * h[8] / 4 + 255) / 256 : (h[7] * h[8] + 255) / 256;
        case 1: return (h[8] * h[10] + 255) / 256;
        case 2: return h[15] * h[7] * 4;
        case 4: return (h[7] + 3) / 4;
        default: return -1;
    }
}

}  // namespace

namespace ccint {

int wgrad_reduce_launch(const long* desc, int n, hipStream_t s) {
    for (int b = 0; b < n; b += NRD) {
        RT t = {};
        const int m = (n - b) < NRD ? (n - b) : NRD;
        long blk = 0;
        for (int k = 0; k < m; k++) {
            const long* h = desc + (long)(b + k) * RD_LONGS;
            RD& d = t.d[k];
            d.kind = (int)h[0]; d.ws = (const float*)h[1]; d.gw = (float*)h[2]; d.nsplit = (int)h[3]; d.accum = (int)h[4];
            d.o_sm = h[5]; d.o_sc = h[6];
            for (int i = 0; i < 9; i++) d.p[i] = (int)h[7 + i];
            long hv[RD_LONGS];
            for (int i = 0; i < RD_LONGS; i++) hv[i] = h[i];
            if (d.kind == 0) {      // p[6]: identity destination mapping, float4-able
                const long tot = h[7] * h[8];
                hv[13] = (d.o_sm == h[8] && d.o_sc == h[9] && h[11] == h[10] && h[12] == 1 && tot % 4 == 0 &&
                          ((uintptr_t)d.ws % 16 == 0) && ((uintptr_t)d.gw % 16 == 0)) ? 1 : 0;
                d.p[6] = (int)hv[13];
            }
            const long nb = blocks_of(hv);
            if (nb <= 0 || !d.ws || !d.gw || d.nsplit <= 0 || blk + nb >= (1l << 31)) return CC_ERR_ARG;
            blk += nb;
            d.blk_end = (int)blk;
        }
        t.n = m;
        hipLaunchKernelGGL(k_wgrad_reduce_table, dim3((unsigned)blk), dim3(256), 0, s, t);
    }
    return CC_OK;
}

int wgrad_reduce_emit(RedSink* sink, const long* desc, int n, hipStream_t s) {
    if (!sink) return wgrad_reduce_launch(desc, n, s);
    if (sink->n + n > sink->cap) return CC_ERR_ARG;
    for (long i = 0; i < (long)n * RD_LONGS; i++) sink->out[(long)sink->n * RD_LONGS + i] = desc[i];
    sink->n += n;
    return CC_OK;
}

}  // namespace ccint

extern "C" {

int cc_wgrad_reduce_table(const long* desc_host, int n, void* stream) {
    if (!desc_host || n <= 0) return CC_ERR_ARG;
    const int r = ccint::wgrad_reduce_launch(desc_host, n, (hipStream_t)stream);
    if (r != CC_OK) return r;
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
