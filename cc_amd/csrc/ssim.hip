// Fused 13x13-Gaussian SSIM + robust-L1 + mask photometric kernels (gfx950).
//
// Replaces, per (scale, reference frame), ssim.py:19-36 (five depth-wise 13x13 conv2d) plus the
// ~40 elementwise ATen ops of loss_functions.py:44-58 / :100-114 / :181-193 with
//   k_ssim_tile<MODE>   one pass: LDS tile (+6 px halo) -> separable 13-tap filter of the five
//                       moments (mu_x, mu_y, E[xx], E[yy], E[xy]) -> SSIM -> valid / masks /
//                       robust-L1 -> block-reduced partial sums (+ the three adjoint maps)
//   k_ssim_adjoint      separable filter of the adjoint maps = d(loss)/d(warped image)
//   k_photo_finalize    deterministic reduction of the partial sums -> loss term, OOB normaliser
// The window is separable (outer product of the 1-D Gaussian, ssim.py:13-17), so a tile costs
// 2 x 13 taps instead of 169.  Zero padding 6 as in the reference.
//
// Tile: 32x32 outputs per 256-thread workgroup, one channel at a time through LDS
// (2 x 44x44 inputs + 5 x 44x32 row-filtered moments = 43 KB -> 3 workgroups / CU).
//   H pass: work item = (row, 4 consecutive columns): 16 inputs per image via ds_read_b128,
//           4 outputs x 13 taps x 5 moments in registers, ds_write_b128.
//   V pass: thread = (column, 4 consecutive rows): 16 ds_read_b32 per moment (conflict-free:
//           a 32-lane group reads 32 consecutive columns).
#include "cc_common.h"
#include "jobs.h"
#include "../../include/ccengine.h"

namespace {


using ccjobs::JobTab;

constexpr int TS = 32;          // output tile edge
constexpr int HALO = 6;         // window_size // 2
constexpr int TIN = TS + 2 * HALO;   // 44
constexpr float C1 = 0.0001f;   // 0.01^2  (ssim.py:31)
constexpr float C2 = 0.0009f;   // 0.03^2  (ssim.py:32)

struct Gauss13 { float g[13]; };

enum { MODE_MAP = 0, MODE_PHOTO = 1, MODE_ERR = 2, MODE_GRAD = 3 };

struct PhotoArgs {
    const float* x;        // tgt   [B,3,H,W]
    const float* y;        // warped[B,3,H,W]
    const float* mask_a;   // non-differentiable factor (1-occ), element (b, y, x) at mask_a[b*a_bs + p], or null
    const float* mask_b;   // differentiable factor (explainability), at mask_b[b*b_bs + p], or null
    int a_bs, b_bs;
    int b_complement;      // use (1 - mask_b)  (train.py:488 flow_exp_mask)
    const float* upstream; // MODE_GRAD: d(loss)/d(ssim map) [B,3,H,W]
    float* out_map;        // MODE_MAP: ssim [B,3,H,W];  MODE_ERR: err [B,1,H,W]
    float* out_valid;      // MODE_ERR: valid [B,1,H,W]
    float* partials;       // MODE_PHOTO: [nblk][4]
    float* adjA; float* adjB; float* adjC;   // MODE_PHOTO (grad): [B,3,H,W] each
    float* g0;             // MODE_PHOTO (grad): direct robust-L1 adjoint [B,3,H,W]
    float* gmask;          // MODE_PHOTO (grad): d/d mask_b (unscaled), at gmask[b*gm_bs + p], or null
    int gm_bs;
    int want_grad;
    float wssim, q;
    int H, W;
};

__device__ __forceinline__ float robust_pow(float v, float q) {
    // (x^2 + 0.01)^q  (loss_functions.py:18-25); q = 0.5 is the only value the reference trains with (v >= 0.01: cc_sqrt)
    return (q == 0.5f) ? cc_sqrt(v) : powf(v, q);
}

// one 32x32 tile of image b: (tile_x, tile_y) -> outputs; `blk` = index of this tile's partial sums (MODE_PHOTO)
template <int MODE>
__device__ __forceinline__ void ssim_tile_body(const PhotoArgs& a, const Gauss13& gw, int b, int tile_x, int tile_y, size_t blk) {
    // The horizontal pass reads its 16-pixel input windows straight from global memory (8-byte loads, 16 in flight per work
    // item) instead of staging the two 44x44 input tiles in LDS first: 28 KB of LDS instead of 43 KB, one barrier less per
    // channel, no scalar staging loop with an integer division per element.
    // four moment planes: E[x], E[y], E[xx + yy], E[xy] -- the SSIM map and its adjoints need sigma_x^2 + sigma_y^2 only as a sum
    // (ssim.py:33 `sigma1_sq + sigma2_sq + C2`), so E[xx] and E[yy] are filtered together (a fifth of the filter arithmetic less)
    __shared__ __attribute__((aligned(16))) float hb[4][TIN * TS];
    __shared__ float red[4 * 3];

    const int H = a.H, W = a.W, HW = H * W;
    const int ox0 = tile_x * TS, oy0 = tile_y * TS;
    const int tid = threadIdx.x;
    const int cx = tid & 31, rg = tid >> 5;
    const int gx = ox0 + cx;

    // per-thread pixels: column gx, rows oy0 + 4*rg + j
    float valid[4], m[4], ma[4], acc_gm[4], err_rob[4], err_ss[4];
    bool inimg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = oy0 + 4 * rg + j;
        inimg[j] = (gx < W) && (gy < H);
        valid[j] = 0.f; m[j] = 0.f; ma[j] = 1.f; acc_gm[j] = 0.f; err_rob[j] = 0.f; err_ss[j] = 0.f;
        if (inimg[j]) {
            const int p = gy * W + gx;
            if (MODE == MODE_PHOTO || MODE == MODE_ERR) {
                const float* yp = a.y + (size_t)b * 3 * HW + p;
                // loss_functions.py:45,100: 1 - prod_c(warped == 0)
                valid[j] = (yp[0] == 0.f && yp[HW] == 0.f && yp[2 * HW] == 0.f) ? 0.f : 1.f;
            }
            if (MODE == MODE_PHOTO) {
                float mm = 1.f;
                if (a.mask_a) { ma[j] = a.mask_a[(size_t)b * a.a_bs + p]; mm = ma[j]; }
                if (a.mask_b) {
                    float mb = a.mask_b[(size_t)b * a.b_bs + p];
                    if (a.b_complement) mb = 1.f - mb;
                    // reference order: diff * (1-occ) * exp  (rigid, :107-108) / diff * exp * (1-occ) (flow, :51-56);
                    // both are one product of two factors -> commutative in fp32
                    mm = mm * mb;
                }
                m[j] = mm;
            }
        }
    }

    float s_rob = 0.f, s_sl = 0.f;

    // even width + 8-byte aligned planes: rows start 8-byte aligned and a pair never straddles the image edge
    const bool pairs = !(W & 1) && ((((uintptr_t)a.x) | ((uintptr_t)a.y)) & 7) == 0;
    const bool colsafe = (ox0 >= HALO) && (ox0 + TS + HALO <= W);          // no column of the tile's input window leaves the image
    for (int c = 0; c < 3; c++) {
        const float* xp = a.x + ((size_t)b * 3 + c) * HW;
        const float* yp = a.y + ((size_t)b * 3 + c) * HW;
        if (c > 0) __syncthreads();          // everyone is done with the previous channel's V pass
        // ---- H pass.  Every load is unconditional: rows outside the image are clamped to an image row and the item's outputs zeroed
        // afterwards; columns outside the image are clamped and the loaded value zeroed (only the first / last tile column of an
        // image has any: `colsafe` tiles skip that too).  A load under `in ? *p : 0` compiles to a branch around each load (16 basic
        // blocks with ~17 instructions of mask bookkeeping each: 40 % of the pass).
        for (int it = tid; it < TIN * (TS / 4); it += 256) {
            const int r = it >> 3, cg = it & 7;
            const int yy = oy0 - HALO + r, x0 = ox0 - HALO + 4 * cg;
            const bool rowin = (yy >= 0) && (yy < H);
            const int yc = yy < 0 ? 0 : (yy < H ? yy : H - 1);
            const float* xr = xp + (long)yc * W;
            const float* yr = yp + (long)yc * W;
            float xv[16], yv[16];
            if (pairs && colsafe) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float2 vx = *reinterpret_cast<const float2*>(xr + x0 + 2 * k);
                    const float2 vy = *reinterpret_cast<const float2*>(yr + x0 + 2 * k);
                    xv[2 * k] = vx.x; xv[2 * k + 1] = vx.y;
                    yv[2 * k] = vy.x; yv[2 * k + 1] = vy.y;
                }
            } else if (pairs) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int xx = x0 + 2 * k;
                    const bool in = (xx >= 0) && (xx < W);
                    const int xc = in ? xx : 0;
                    const float2 vx = *reinterpret_cast<const float2*>(xr + xc);
                    const float2 vy = *reinterpret_cast<const float2*>(yr + xc);
                    xv[2 * k] = in ? vx.x : 0.f; xv[2 * k + 1] = in ? vx.y : 0.f;
                    yv[2 * k] = in ? vy.x : 0.f; yv[2 * k + 1] = in ? vy.y : 0.f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int xx = x0 + k;
                    const bool in = (xx >= 0) && (xx < W);
                    const int xc = in ? xx : 0;
                    const float vx = xr[xc], vy = yr[xc];
                    xv[k] = in ? vx : 0.f;
                    yv[k] = in ? vy : 0.f;
                }
            }
            float o[4][4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int j = 0; j < 4; j++) o[mi][j] = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float vx = xv[k], vy = yv[k];
                const float pq = fmaf(vy, vy, vx * vx), pxy = vx * vy;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int t = k - j;
                    if (t >= 0 && t < 13) {
                        const float g = gw.g[t];
                        o[0][j] = fmaf(g, vx, o[0][j]);
                        o[1][j] = fmaf(g, vy, o[1][j]);
                        o[2][j] = fmaf(g, pq, o[2][j]);
                        o[3][j] = fmaf(g, pxy, o[3][j]);
                    }
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                *reinterpret_cast<float4*>(&hb[mi][r * TS + 4 * cg]) =
                    rowin ? make_float4(o[mi][0], o[mi][1], o[mi][2], o[mi][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // ---- V pass
        float mo[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
#pragma unroll
            for (int j = 0; j < 4; j++) mo[mi][j] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float v = hb[mi][(4 * rg + i) * TS + cx];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int t = i - j;
                    if (t >= 0 && t < 13) mo[mi][j] = fmaf(gw.g[t], v, mo[mi][j]);
                }
            }
        }
        // ---- per-pixel SSIM (ssim.py:20-34) and loss pieces
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!inimg[j]) continue;
            const int p = (oy0 + 4 * rg + j) * W + gx;
            const float mu1 = mo[0][j], mu2 = mo[1][j];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s12 = mo[3][j] - mu12;
            const float num1 = 2.f * mu12 + C1, num2 = 2.f * s12 + C2;
            const float den1 = mu1_sq + mu2_sq + C1, den2 = ((mo[2][j] - mu1_sq) - mu2_sq) + C2;       // sigma1_sq + sigma2_sq + C2
            // ONE reciprocal per pixel and channel (den1 >= C1, den2 ~ C2: a normal number; v_rcp_f32, 1 ulp) shared by the SSIM value
            // and its adjoints: the IEEE divisions and sqrtf of this block were ~40 of its ~150 VALU instructions per pixel
            const float iD = cc_rcp(den1 * den2);
            const float S = (num1 * num2) * iD;
            if (MODE == MODE_MAP) {
                a.out_map[((size_t)b * 3 + c) * HW + p] = S;
                continue;
            }
            if (MODE == MODE_GRAD) {
                // adjoint maps of the SSIM map w.r.t. the SECOND image (call with swapped roles for the first)
                const size_t o = ((size_t)b * 3 + c) * HW + p;
                const float gS = a.upstream[o];
                // 1 / den2 = den1 / D, 1 / den1 = den2 / D
                const float id1 = den2 * iD, id2 = den1 * iD;
                a.adjC[o] = gS * (2.f * num1 * iD);
                a.adjB[o] = gS * (-S * id2);
                a.adjA[o] = gS * (2.f * mu1 * (num2 - num1) * iD - 2.f * mu2 * S * (id1 - id2));
                continue;
            }
            const float xc = xp[p], yc = yp[p];
            if (MODE == MODE_ERR) {
                // loss_functions.py:181-188: robust_l1_per_pix(tgt - warped) and (1 - ssim), channel means
                const float d = xc - yc;
                err_rob[j] += cc_sqrt(d * d + 0.01f);
                err_ss[j] += 1.f - S;
                continue;
            }
            // MODE_PHOTO
            const float vm = valid[j] * m[j];
            const float d = (xc - yc) * vm;                    // diff * valid * masks
            const float base = d * d + 0.01f;
            const float rob = robust_pow(base, a.q);
            const float sl = (1.f - S * valid[j]) * m[j];       // ssim_loss
            s_rob += rob;
            s_sl += sl;
            if (a.want_grad) {
                // d rob / d d = q * base^(q-1) * 2 d
                const float drob = (a.q == 0.5f) ? (d * cc_rcp(rob)) : (a.q * powf(base, a.q - 1.f) * 2.f * d);
                const size_t o = ((size_t)b * 3 + c) * HW + p;
                a.g0[o] = -drob * vm;                           // d/dy through diff
                const float id1 = den2 * iD, id2 = den1 * iD;
                const float gS = -vm * a.wssim;                 // d(wssim * sl)/dS
                a.adjC[o] = gS * (2.f * num1 * iD);             // dS/dE[xy]
                a.adjB[o] = gS * (-S * id2);                    // dS/dE[yy]
                a.adjA[o] = gS * (2.f * mu1 * (num2 - num1) * iD - 2.f * mu2 * S * (id1 - id2));  // dS/dmu_y
                // d/d mask_b: diff and ssim_loss are both linear in the mask product
                acc_gm[j] += drob * (xc - yc) * valid[j] * ma[j] + a.wssim * (1.f - S * valid[j]) * ma[j];
            }
        }
    }

    if (MODE == MODE_ERR) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!inimg[j]) continue;
            const int p = (oy0 + 4 * rg + j) * W + gx;
            // (1-wssim) * mean_c(robust) + wssim * mean_c(1 - ssim)   (torch mean over 3 channels = sum / 3)
            a.out_map[(size_t)b * HW + p] = (1.f - a.wssim) * (err_rob[j] / 3.f) + a.wssim * (err_ss[j] / 3.f);
            a.out_valid[(size_t)b * HW + p] = valid[j];
        }
    }
    if (MODE == MODE_PHOTO) {
        float s_valid = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!inimg[j]) continue;
            s_valid += valid[j];
            if (a.want_grad && a.gmask) {
                const int p = (oy0 + 4 * rg + j) * W + gx;
                a.gmask[(size_t)b * a.gm_bs + p] = a.b_complement ? -acc_gm[j] : acc_gm[j];
            }
        }
        float v[3] = {s_rob, s_sl, s_valid};
        __syncthreads();
        cc::block_sum_256<3>(v, red);
        if (tid == 0) {
            a.partials[blk * 4 + 0] = v[0];
            a.partials[blk * 4 + 1] = v[1];
            a.partials[blk * 4 + 2] = v[2];
            a.partials[blk * 4 + 3] = 0.f;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_ssim_tile(PhotoArgs a, Gauss13 gw) {
    ssim_tile_body<MODE>(a, gw, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y,
                         ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// job-table forms (jobs.h): every (pyramid level, reference frame) term of a loss in ONE launch; blocks = 32x32 tiles
__device__ __forceinline__ void tile_of(const JobTab& t, int& j, int& b, int& tile_x, int& tile_y, int& local) {
    int first;
    j = ccjobs::find(t, (int)blockIdx.x, first);
    const int tw = (t.W[j] + TS - 1) / TS, th = (t.H[j] + TS - 1) / TS;
    // XCD-aware order: workgroup i runs on XCD i % 8 (each with its own L2), so the job's tiles are dealt out as eight contiguous
    // runs -- neighbouring tiles, which share their 6-pixel halos, then meet in one L2 instead of re-reading them from HBM
    // (measured: k_ssim_photo_jobs 195 -> 169 us, profiles/r04_ab_round4.txt)
    {
        const int n = t.blk_end[j] - first, l = (int)blockIdx.x - first;
        const int k = l & 7, q = n >> 3, r = n & 7;
        local = k * q + (k < r ? k : r) + (l >> 3);
    }
    b = local / (tw * th);
    const int r = local - b * (tw * th);
    tile_y = r / tw;
    tile_x = r - tile_y * tw;
}

// photo slots: 0 tgt, 1 warped, 2 mask_a, 3 mask_b, 4 gmask, 5 adjoint maps (adjA, adjB, adjC, g0: 4 x [B,3,H,W]),
//              6 partials [B*tiles][4], 7 batch strides of mask_a / mask_b / gmask in units of H*W (8 bits each)
struct PhotoCommon { int b_complement, want_grad; float wssim, q; };

__global__ __launch_bounds__(256, 4) void k_ssim_photo_jobs(JobTab t, PhotoCommon c, Gauss13 gw) {
    int j, b, tile_x, tile_y, local;
    tile_of(t, j, b, tile_x, tile_y, local);
    PhotoArgs a = {};
    a.H = t.H[j]; a.W = t.W[j];
    const int HW = a.H * a.W;
    const long st = t.slot[j][7];
    a.x = ccjobs::ptr<const float>(t, j, 0);
    a.y = ccjobs::ptr<const float>(t, j, 1);
    a.mask_a = ccjobs::ptr<const float>(t, j, 2);
    a.mask_b = ccjobs::ptr<const float>(t, j, 3);
    a.gmask = ccjobs::ptr<float>(t, j, 4);
    a.a_bs = (int)(st & 255) * HW; a.b_bs = (int)((st >> 8) & 255) * HW; a.gm_bs = (int)((st >> 16) & 255) * HW;
    float* adj = ccjobs::ptr<float>(t, j, 5);
    const size_t map = (size_t)t.B * 3 * HW;
    a.adjA = adj; a.adjB = adj + map; a.adjC = adj + 2 * map; a.g0 = adj + 3 * map;
    a.partials = ccjobs::ptr<float>(t, j, 6);
    a.b_complement = c.b_complement; a.want_grad = c.want_grad; a.wssim = c.wssim; a.q = c.q;
    ssim_tile_body<MODE_PHOTO>(a, gw, b, tile_x, tile_y, (size_t)local);
}

// err slots (consensus_exp_masks): 0 tgt, 1 warped, 2 err [B,1,H,W], 3 valid [B,1,H,W]
__global__ __launch_bounds__(256) void k_ssim_err_jobs(JobTab t, float wssim, Gauss13 gw) {
    int j, b, tile_x, tile_y, local;
    tile_of(t, j, b, tile_x, tile_y, local);
    PhotoArgs a = {};
    a.H = t.H[j]; a.W = t.W[j];
    a.x = ccjobs::ptr<const float>(t, j, 0);
    a.y = ccjobs::ptr<const float>(t, j, 1);
    a.out_map = ccjobs::ptr<float>(t, j, 2);
    a.out_valid = ccjobs::ptr<float>(t, j, 3);
    a.wssim = wssim;
    ssim_tile_body<MODE_ERR>(a, gw, b, tile_x, tile_y, 0);
}

// gy = scale * (g0 + G*adjA + 2*y*(G*adjB) + x*(G*adjC)),  G* = zero-padded 13x13 Gaussian filter
__device__ __forceinline__ void ssim_adjoint_body(const float* __restrict__ adjA, const float* __restrict__ adjB,
                                                  const float* __restrict__ adjC, const float* __restrict__ g0,
                                                  const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ scale, float* __restrict__ gy, int H,
                                                  int W, int accumulate, const Gauss13& gw, int b, int tile_x, int tile_y) {
    __shared__ __attribute__((aligned(16))) float tin[3][TIN * TIN];
    __shared__ __attribute__((aligned(16))) float hb[3][TIN * TS];
    const int HW = H * W;
    const int ox0 = tile_x * TS, oy0 = tile_y * TS;
    const int tid = threadIdx.x, cx = tid & 31, rg = tid >> 5, gx = ox0 + cx;
    const float sc = scale ? scale[0] : 1.f;
    for (int c = 0; c < 3; c++) {
        const size_t plane = ((size_t)b * 3 + c) * HW;
        if (c > 0) __syncthreads();
        for (int i0 = tid; i0 < TIN * TIN; i0 += 4 * 256) {        // 12 loads in flight per work item, then the LDS stores
            float va[4], vb[4], vc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + 256 * u;
                const int r = i / TIN, col = i - r * TIN;
                const int yy = oy0 - HALO + r, xx = ox0 - HALO + col;
                const bool in = (i < TIN * TIN) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
                // unconditional loads from a clamped address, then a select (a load under `in ? *p : 0` is a branch around each load)
                const size_t o = plane + (in ? (size_t)yy * W + xx : 0);
                const float ta = adjA[o], tb = adjB[o], tc = adjC[o];
                va[u] = in ? ta : 0.f;
                vb[u] = in ? tb : 0.f;
                vc[u] = in ? tc : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + 256 * u;
                if (i < TIN * TIN) { tin[0][i] = va[u]; tin[1][i] = vb[u]; tin[2][i] = vc[u]; }
            }
        }
        __syncthreads();
        for (int it = tid; it < TIN * (TS / 4); it += 256) {
            const int r = it >> 3, cg = it & 7;
            // (scalar FMAs: on gfx950 a v_pk_fma_f32 costs more than the two v_fma_f32 it replaces -- the packed forward filter ran
            // 47 % slower with 25 % fewer VALU instructions, profiles/r04_ab_round4.txt)
            float va[16], vb[16], vc[16];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 qa = reinterpret_cast<const float4*>(&tin[0][r * TIN + 4 * cg])[k];
                const float4 qb = reinterpret_cast<const float4*>(&tin[1][r * TIN + 4 * cg])[k];
                const float4 qc = reinterpret_cast<const float4*>(&tin[2][r * TIN + 4 * cg])[k];
                va[4 * k] = qa.x; va[4 * k + 1] = qa.y; va[4 * k + 2] = qa.z; va[4 * k + 3] = qa.w;
                vb[4 * k] = qb.x; vb[4 * k + 1] = qb.y; vb[4 * k + 2] = qb.z; vb[4 * k + 3] = qb.w;
                vc[4 * k] = qc.x; vc[4 * k + 1] = qc.y; vc[4 * k + 2] = qc.z; vc[4 * k + 3] = qc.w;
            }
            float oa[4], ob[4], oc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { oa[j] = 0.f; ob[j] = 0.f; oc[j] = 0.f; }
#pragma unroll
            for (int k = 0; k < 16; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int t = k - j;
                    if (t >= 0 && t < 13) {
                        oa[j] = fmaf(gw.g[t], va[k], oa[j]);
                        ob[j] = fmaf(gw.g[t], vb[k], ob[j]);
                        oc[j] = fmaf(gw.g[t], vc[k], oc[j]);
                    }
                }
            }
            *reinterpret_cast<float4*>(&hb[0][r * TS + 4 * cg]) = make_float4(oa[0], oa[1], oa[2], oa[3]);
            *reinterpret_cast<float4*>(&hb[1][r * TS + 4 * cg]) = make_float4(ob[0], ob[1], ob[2], ob[3]);
            *reinterpret_cast<float4*>(&hb[2][r * TS + 4 * cg]) = make_float4(oc[0], oc[1], oc[2], oc[3]);
        }
        __syncthreads();
        float mo[3][4];
#pragma unroll
        for (int mi = 0; mi < 3; mi++) {
#pragma unroll
            for (int j = 0; j < 4; j++) mo[mi][j] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float v = hb[mi][(4 * rg + i) * TS + cx];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int t = i - j;
                    if (t >= 0 && t < 13) mo[mi][j] = fmaf(gw.g[t], v, mo[mi][j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gyy = oy0 + 4 * rg + j;
            if (gx < W && gyy < H) {
                const size_t o = plane + (size_t)gyy * W + gx;
                const float r = sc * ((g0 ? g0[o] : 0.f) + mo[0][j] + 2.f * y[o] * mo[1][j] + x[o] * mo[2][j]);
                gy[o] = accumulate ? gy[o] + r : r;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_ssim_adjoint(const float* __restrict__ adjA, const float* __restrict__ adjB,
                                                      const float* __restrict__ adjC, const float* __restrict__ g0,
                                                      const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ scale, float* __restrict__ gy, int H,
                                                      int W, int accumulate, Gauss13 gw) {
    ssim_adjoint_body(adjA, adjB, adjC, g0, x, y, scale, gy, H, W, accumulate, gw, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y);
}

// adjoint slots: 0 adjoint maps (adjA, adjB, adjC, g0), 1 tgt, 2 warped, 3 scale (1 float), 4 gwarped [B,3,H,W]
__global__ __launch_bounds__(256) void k_ssim_adjoint_jobs(JobTab t, Gauss13 gw) {
    int j, b, tile_x, tile_y, local;
    tile_of(t, j, b, tile_x, tile_y, local);
    const int H = t.H[j], W = t.W[j];
    const float* adj = ccjobs::ptr<const float>(t, j, 0);
    const size_t map = (size_t)t.B * 3 * H * W;
    ssim_adjoint_body(adj, adj + map, adj + 2 * map, adj + 3 * map, ccjobs::ptr<const float>(t, j, 1), ccjobs::ptr<const float>(t, j, 2),
                      ccjobs::ptr<const float>(t, j, 3), ccjobs::ptr<float>(t, j, 4), H, W, 0, gw, b, tile_x, tile_y);
}

// all terms of one photometric loss: one wave per job (16 waves), then the terms are added to the loss in job order.
// slots: 6 partials [B*tiles][4] (as written by k_ssim_photo_jobs); scale_out[j], terms in LDS; nan_flag set on any NaN term
__global__ __launch_bounds__(1024) void k_photo_finalize_jobs(JobTab t, float wssim, float q, float lambda_oob,
                                                              float* __restrict__ loss_accum, float* __restrict__ scale_out,
                                                              float* __restrict__ nan_flag) {
    __shared__ float terms[ccjobs::MAXJOBS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < t.n; j += 16) {
        const int nblk = t.B * ((t.W[j] + TS - 1) / TS) * ((t.H[j] + TS - 1) / TS);
        const float* partials = ccjobs::ptr<const float>(t, j, 6);
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        for (int i = lane; i < nblk; i += 64) {
            v0 += partials[i * 4 + 0];
            v1 += partials[i * 4 + 1];
            v2 += partials[i * 4 + 2];
        }
        v0 = cc::wave_sum(v0); v1 = cc::wave_sum(v1); v2 = cc::wave_sum(v2);
        if (lane == 0) {
            const float n1 = (float)t.B * (float)t.H[j] * (float)t.W[j], n3 = 3.f * n1;
            const float oob = n1 / v2;
            float term = (1.f - wssim) * oob * (v0 / n3 + wssim * (v1 / n3));
            if (lambda_oob != 0.f) {
                const float hi = powf(1.01f, q), lo = powf(0.01f, q);
                term += lambda_oob * (((n1 - v2) * hi + v2 * lo) / n1);
            }
            terms[j] = term;
            scale_out[j] = (1.f - wssim) * oob / n3;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = loss_accum[0];
        bool bad = false;
        for (int j = 0; j < t.n; j++) {
            acc += terms[j];
            bad = bad || !(terms[j] == terms[j]);
        }
        loss_accum[0] = acc;
        if (bad) nan_flag[0] = 1.f;
    }
}

// loss_functions.py:48,58 / :103,114: oob = N/sum(valid);  term = (1-wssim)*oob*(mean(rob) + wssim*mean(sl))
//                                     + lambda_oob * robust_l1(1 - valid)
// out[0] += term;  out[1] = (1-wssim)*oob/N3 (the common scale of every adjoint of this term);
// out[2] = 1 if the term is NaN (deferred version of the reference's `assert loss == loss`)
__global__ __launch_bounds__(256) void k_photo_finalize(const float* __restrict__ partials, int nblk, float n1, float n3,
                                                        float wssim, float q, float lambda_oob,
                                                        float* __restrict__ loss_accum, float* __restrict__ scale_out,
                                                        float* __restrict__ nan_flag) {
    __shared__ float red[4 * 3];
    float v[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblk; i += 256) {
        v[0] += partials[i * 4 + 0];
        v[1] += partials[i * 4 + 1];
        v[2] += partials[i * 4 + 2];
    }
    cc::block_sum_256<3>(v, red);
    if (threadIdx.x == 0) {
        const float oob = n1 / v[2];
        float term = (1.f - wssim) * oob * (v[0] / n3 + wssim * (v[1] / n3));
        if (lambda_oob != 0.f) {
            const float hi = powf(1.01f, q), lo = powf(0.01f, q);
            term += lambda_oob * (((n1 - v[2]) * hi + v[2] * lo) / n1);
        }
        loss_accum[0] += term;
        scale_out[0] = (1.f - wssim) * oob / n3;
        if (!(term == term)) nan_flag[0] = 1.f;
    }
}

// loss_functions.py:189-193: target = (wrig * min(err_cf, err_cb) * (valid_cf OR valid_cb) <= err_ff + 1e-8)
__global__ __launch_bounds__(256) void k_consensus_combine(const float* __restrict__ err_cf, const float* __restrict__ err_cb,
                                                           const float* __restrict__ err_ff, const float* __restrict__ v_cf,
                                                           const float* __restrict__ v_cb, float* __restrict__ target,
                                                           float wrig, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float valid = 1.f - (1.f - v_cf[i]) * (1.f - v_cb[i]);
    const float cam_err = fminf(err_cf[i], err_cb[i]) * valid;
    target[i] = (wrig * cam_err <= (err_ff[i] + 1e-8f)) ? 1.f : 0.f;
}

inline dim3 tile_grid(int B, int H, int W) { return dim3((W + TS - 1) / TS, (H + TS - 1) / TS, B); }

inline Gauss13 make_gauss(const float* g13) {
    Gauss13 g;
    for (int i = 0; i < 13; i++) g.g[i] = g13[i];
    return g;
}

}  // namespace

extern "C" {

size_t cc_ssim_num_blocks(int B, int H, int W) { return B * ((W + TS - 1) / TS) * ((H + TS - 1) / TS); }

int cc_ssim_fwd(const float* img1, const float* img2, float* out, const float* gauss13_host, int B, int H, int W,
                void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    PhotoArgs a = {};
    a.x = img1; a.y = img2; a.out_map = out; a.H = H; a.W = W;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ssim_tile<MODE_MAP>), tile_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, a,
                       make_gauss(gauss13_host));
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_ssim_photo_fwd(const float* tgt, const float* warped, const float* mask_a, int mask_a_bstride,
                      const float* mask_b, int mask_b_bstride, int mask_b_complement, float* partials, float* adjA,
                      float* adjB, float* adjC, float* g0, float* gmask, int gmask_bstride, int want_grad,
                      float wssim, float q, float lambda_oob, float* loss_accum, float* scale_out, float* nan_flag,
                      const float* gauss13_host, int B, int H, int W, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    PhotoArgs a = {};
    a.x = tgt; a.y = warped; a.mask_a = mask_a; a.a_bs = mask_a_bstride; a.mask_b = mask_b; a.b_bs = mask_b_bstride;
    a.b_complement = mask_b_complement; a.partials = partials; a.adjA = adjA; a.adjB = adjB; a.adjC = adjC; a.g0 = g0;
    a.gmask = gmask; a.gm_bs = gmask_bstride; a.want_grad = want_grad; a.wssim = wssim; a.q = q; a.H = H; a.W = W;
    dim3 g = tile_grid(B, H, W);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ssim_tile<MODE_PHOTO>), g, dim3(256), 0, s, a, make_gauss(gauss13_host));
    const float n1 = (float)B * (float)H * (float)W;
    hipLaunchKernelGGL(k_photo_finalize, dim3(1), dim3(256), 0, s, (const float*)partials, (int)(g.x * g.y * g.z), n1,
                       3.f * n1, wssim, q, lambda_oob, loss_accum, scale_out, nan_flag);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_ssim_photo_bwd(const float* adjA, const float* adjB, const float* adjC, const float* g0, const float* tgt,
                      const float* warped, const float* scale, float* gwarped, int accumulate,
                      const float* gauss13_host, int B, int H, int W, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_ssim_adjoint, tile_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, adjA, adjB, adjC, g0, tgt,
                       warped, scale, gwarped, H, W, accumulate, make_gauss(gauss13_host));
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* generic autograd of cc_ssim_fwd: given d(loss)/d(ssim) computes d(loss)/d(img2); call again with
 * img1/img2 swapped for d(loss)/d(img1) (SSIM is symmetric).  adjA/B/C: scratch [B,3,H,W] each. */
int cc_ssim_bwd(const float* img1, const float* img2, const float* gout, float* adjA, float* adjB, float* adjC,
                float* gimg2, const float* gauss13_host, int B, int H, int W, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    PhotoArgs a = {};
    a.x = img1; a.y = img2; a.upstream = gout; a.adjA = adjA; a.adjB = adjB; a.adjC = adjC; a.H = H; a.W = W;
    Gauss13 g = make_gauss(gauss13_host);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ssim_tile<MODE_GRAD>), tile_grid(B, H, W), dim3(256), 0, s, a, g);
    hipLaunchKernelGGL(k_ssim_adjoint, tile_grid(B, H, W), dim3(256), 0, s, (const float*)adjA, (const float*)adjB,
                       (const float*)adjC, (const float*)nullptr, img1, img2, (const float*)nullptr, gimg2, H, W, 0, g);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_ssim_err_fwd(const float* tgt, const float* warped, float* err, float* valid, float wssim,
                    const float* gauss13_host, int B, int H, int W, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return CC_ERR_ARG;
    PhotoArgs a = {};
    a.x = tgt; a.y = warped; a.out_map = err; a.out_valid = valid; a.wssim = wssim; a.H = H; a.W = W;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ssim_tile<MODE_ERR>), tile_grid(B, H, W), dim3(256), 0, (hipStream_t)stream, a,
                       make_gauss(gauss13_host));
    CC_CHECK_LAUNCH();
    return CC_OK;
}

/* ---- job-table forms (jobs: HOST array of njobs x 10 longs {slot0..7, H, W}; njobs <= 24; slots per kernel above) */
static int tile_blocks(int H, int W) { return ((W + TS - 1) / TS) * ((H + TS - 1) / TS); }

int cc_ssim_photo_fwd_jobs(const long* jobs, int njobs, int B, int mask_b_complement, int want_grad, float wssim, float q,
                           float lambda_oob, float* loss_accum, float* scale_out, float* nan_flag, const float* gauss13_host,
                           void* stream) {
    if (!jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return CC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    JobTab t;
    const int nblk = ccjobs::fill(t, jobs, njobs, B, tile_blocks);
    PhotoCommon c = {mask_b_complement, want_grad, wssim, q};
    hipLaunchKernelGGL(k_ssim_photo_jobs, dim3((unsigned)nblk), dim3(256), 0, s, t, c, make_gauss(gauss13_host));
    hipLaunchKernelGGL(k_photo_finalize_jobs, dim3(1), dim3(1024), 0, s, t, wssim, q, lambda_oob, loss_accum, scale_out, nan_flag);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_ssim_photo_bwd_jobs(const long* jobs, int njobs, int B, const float* gauss13_host, void* stream) {
    if (!jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return CC_ERR_ARG;
    JobTab t;
    const int nblk = ccjobs::fill(t, jobs, njobs, B, tile_blocks);
    hipLaunchKernelGGL(k_ssim_adjoint_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t, make_gauss(gauss13_host));
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_ssim_err_fwd_jobs(const long* jobs, int njobs, int B, float wssim, const float* gauss13_host, void* stream) {
    if (!jobs || njobs <= 0 || njobs > ccjobs::MAXJOBS || B <= 0) return CC_ERR_ARG;
    JobTab t;
    const int nblk = ccjobs::fill(t, jobs, njobs, B, tile_blocks);
    hipLaunchKernelGGL(k_ssim_err_jobs, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, t, wssim, make_gauss(gauss13_host));
    CC_CHECK_LAUNCH();
    return CC_OK;
}

int cc_consensus_target(const float* err_cam_fwd, const float* err_cam_bwd, const float* err_flow_fwd,
                        const float* valid_cam_fwd, const float* valid_cam_bwd, float* target, float wrig, int n,
                        void* stream) {
    if (n <= 0) return CC_ERR_ARG;
    hipLaunchKernelGGL(k_consensus_combine, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, err_cam_fwd,
                       err_cam_bwd, err_flow_fwd, valid_cam_fwd, valid_cam_bwd, target, wrig, n);
    CC_CHECK_LAUNCH();
    return CC_OK;
}

}  // extern "C"
