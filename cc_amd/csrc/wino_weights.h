// Winograd F(2x2, 3x3) weight transform  U = G g G^T  (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], Lavin & Gray 2016),
// written in the layout the MFMA kernel of wino.hip stages by LDS-DMA:
//   U[mb][kc][f][quad][m64][c4]   mb = m / 64, kc = c / 8, f = 4*i + j (frequency), quad = (c % 8) / 4, m64 = m % 64, c4 = c % 4
// -> one (mb, kc) block = 8192 contiguous floats = the A operands of one 8-channel chunk of one 64-row tile; a lane's
// ds_read_b128 at [f][quad = lane>>5][m][0..3] holds the A values of four MFMA k-steps.  Zero beyond M / Cin.
// The device body is shared by the stand-alone kernel (wino.hip: per-call transform into the workspace) and by the
// per-step table launch (conv.hip k_repack_table: descriptors with tap count WINO_T).
#pragma once
#include <hip/hip_runtime.h>

namespace ccwino {

constexpr int WINO_T = 1016;          // "tap count" that marks a Winograd descriptor in the repack table
constexpr int WBM = 64, WCK = 8;      // output rows per workgroup tile / channels per chunk
constexpr int UBLK = 16 * WCK * WBM;  // floats of one (mb, kc) block

__host__ __device__ inline long wino_weight_blocks(int Mpad, int Cpad) { return (long)(Mpad / WBM) * (Cpad / WCK) * 2; }

// one workgroup (256 threads) = one (mb, kc, quad): thread = (m64 = tid >> 2, c4 = tid & 3) -> 16 coalesced 1 KB rows
// canonical filter  g[a][b] = w[w0 + m*w_sm + c*w_sc + i(a)*w_ri + i(b)*w_sj],  i(a) = a (flip == 0) or 2 - a (flip: the taps of a
// data-gradient run backwards: conv.hip make_dgrad_class, dstep = -1)
__device__ __forceinline__ void wino_weight_body(const float* __restrict__ w, float* __restrict__ U, int M, int Cin, int Cpad,
                                                 long w_sm, long w_sc, long w0, long w_ri, long w_sj, int flip, int bid) {
    const int tid = threadIdx.x;
    const int quad = bid & 1;
    const int nkc = Cpad / WCK;
    const int kc = (bid >> 1) % nkc, mb = (bid >> 1) / nkc;
    const int m = mb * WBM + (tid >> 2), c = kc * WCK + quad * 4 + (tid & 3);
    float g[3][3];
    const bool ok = m < M && c < Cin;
    const float* wp = w + w0 + (long)m * w_sm + (long)c * w_sc;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) g[a][b] = ok ? wp[(flip ? 2 - a : a) * w_ri + (flip ? 2 - b : b) * w_sj] : 0.f;
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const float s = g[0][b] + g[2][b];
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (s + g[1][b]);
        t[2][b] = 0.5f * (s - g[1][b]);
        t[3][b] = g[2][b];
    }
    float* o = U + ((long)(mb * nkc + kc)) * UBLK + quad * 256 + tid;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float s = t[i][0] + t[i][2];
        o[(4 * i + 0) * 512] = t[i][0];
        o[(4 * i + 1) * 512] = 0.5f * (s + t[i][1]);
        o[(4 * i + 2) * 512] = 0.5f * (s - t[i][1]);
        o[(4 * i + 3) * 512] = t[i][2];
    }
}

}  // namespace ccwino
