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

// 16 consecutive floats of an image row starting at column x0 (x0 = 2 mod 4 in the tile geometry below: ox0 - 6 + 4 cg), columns
// outside [0, W) read as zero.  VEC: width a multiple of 4 and 16-byte aligned planes -- FIVE 16-byte loads from the aligned column
// x0 - 2 (a group of four columns is then entirely inside or outside the image) instead of eight 8-byte ones: every vector-memory
// instruction costs the issuing wave its place in the in-order stream (round 4), and these kernels issued ~180 of them per work item
// and tile.  All loads are unconditional (clamped address, value selected afterwards: a load under `in ? *p : 0` is a branch).
template <bool VEC>
__device__ __forceinline__ void load_row16(const float* __restrict__ row, int x0, int W, bool colsafe, bool pairs, float (&v)[20]) {
    // -> v[2 .. 17] = columns x0 .. x0 + 15 (v[0], v[1], v[18], v[19]: the rest of the aligned groups in the VEC form, unused)
    if (VEC) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int xx = x0 - 2 + 4 * k;
            const bool in = colsafe || ((xx >= 0) && (xx < W));
            const float4 q = *reinterpret_cast<const float4*>(row + (in ? xx : 0));
            v[4 * k] = in ? q.x : 0.f; v[4 * k + 1] = in ? q.y : 0.f; v[4 * k + 2] = in ? q.z : 0.f; v[4 * k + 3] = in ? q.w : 0.f;
        }
    } else if (pairs) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int xx = x0 + 2 * k;
            const bool in = colsafe || ((xx >= 0) && (xx < W));
            const float2 q = *reinterpret_cast<const float2*>(row + (in ? xx : 0));
            v[2 + 2 * k] = in ? q.x : 0.f; v[3 + 2 * k] = in ? q.y : 0.f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int xx = x0 + k;
            const bool in = (xx >= 0) && (xx < W);
            const float q = row[in ? xx : 0];
            v[2 + k] = in ? q : 0.f;
        }
    }
}

// four consecutive floats at p (VEC: one 16-byte access; else element-wise under `ok[j]`)
template <bool VEC>
__device__ __forceinline__ void load4(const float* __restrict__ p, const bool (&ok)[4], float (&v)[4]) {
    if (VEC) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = ok[j] ? p[j] : 0.f;
    }
}
template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, const bool (&ok)[4], const float (&v)[4]) {
    if (VEC) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) if (ok[j]) p[j] = v[j];
    }
}

// one 32x32 tile of image b: (tile_x, tile_y) -> outputs; `blk` = index of this tile's partial sums (MODE_PHOTO)
// Work item of the vertical pass and the per-pixel part = (tile row, FOUR CONSECUTIVE COLUMNS) (round 6; before: one column, four
// rows): the centre pixels, masks and adjoint maps move as 16-byte accesses, the row-filtered planes are read with ds_read_b128
// (a wave reads 8 rows x 128 B: conflict-free).  Same taps in the same order per output as before: bit-identical results.
template <int MODE, bool VEC>
__device__ __forceinline__ void ssim_tile_body(const PhotoArgs& a, const Gauss13& gw, int b, int tile_x, int tile_y, size_t blk,
                                               float (*hb)[TIN * TS], float* red) {
    // The horizontal pass reads its 16-pixel input windows straight from global memory instead of staging the two 44x44 input tiles
    // in LDS first: 28 KB of LDS instead of 43 KB, one barrier less per channel, no scalar staging loop.
    // hb: four moment planes: E[x], E[y], E[xx + yy], E[xy] -- the SSIM map and its adjoints need sigma_x^2 + sigma_y^2 only as a sum
    // (ssim.py:33 `sigma1_sq + sigma2_sq + C2`), so E[xx] and E[yy] are filtered together (a fifth of the filter arithmetic less)

    const int H = a.H, W = a.W, HW = H * W;
    const int ox0 = tile_x * TS, oy0 = tile_y * TS;
    const int tid = threadIdx.x;
    const int rr = tid >> 3, cq = tid & 7;
    const int gx0 = ox0 + 4 * cq, gy = oy0 + rr;
    const int p0 = gy * W + gx0;                  // (only dereferenced where inimg says so; VEC: gx0 + 3 < W whenever gx0 < W)

    // per-thread pixels: row gy, columns gx0 + j
    float valid[4], m[4], ma[4], acc_gm[4], err_rob[4], err_ss[4];
    bool inimg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        inimg[j] = (gx0 + j < W) && (gy < H);
        valid[j] = 0.f; m[j] = 0.f; ma[j] = 1.f; acc_gm[j] = 0.f; err_rob[j] = 0.f; err_ss[j] = 0.f;
    }
    const bool any = inimg[0];
    if (any) {
        if (MODE == MODE_PHOTO || MODE == MODE_ERR) {
            const float* yp = a.y + (size_t)b * 3 * HW + p0;
            float y0[4], y1[4], y2[4];
            load4<VEC>(yp, inimg, y0); load4<VEC>(yp + HW, inimg, y1); load4<VEC>(yp + 2 * HW, inimg, y2);
#pragma unroll
            for (int j = 0; j < 4; j++)        // loss_functions.py:45,100: 1 - prod_c(warped == 0)
                valid[j] = (!inimg[j] || (y0[j] == 0.f && y1[j] == 0.f && y2[j] == 0.f)) ? 0.f : 1.f;
        }
        if (MODE == MODE_PHOTO) {
            float mb[4];
            if (a.mask_a) load4<VEC>(a.mask_a + (size_t)b * a.a_bs + p0, inimg, ma);
            if (a.mask_b) load4<VEC>(a.mask_b + (size_t)b * a.b_bs + p0, inimg, mb);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!inimg[j]) { ma[j] = 1.f; continue; }
                float mm = a.mask_a ? ma[j] : 1.f;
                if (!a.mask_a) ma[j] = 1.f;
                if (a.mask_b) {
                    const float v = a.b_complement ? 1.f - mb[j] : mb[j];
                    // reference order: diff * (1-occ) * exp  (rigid, :107-108) / diff * exp * (1-occ) (flow, :51-56);
                    // both are one product of two factors -> commutative in fp32
                    mm = mm * v;
                }
                m[j] = mm;
            }
        }
    }

    float s_rob = 0.f, s_sl = 0.f;

    // even width + 8-byte aligned planes: rows start 8-byte aligned and a pair never straddles the image edge
    const bool pairs = !(W & 1) && ((((uintptr_t)a.x) | ((uintptr_t)a.y)) & 7) == 0;
    const bool colsafe = (ox0 >= HALO + 2) && (ox0 + TS + HALO + 2 <= W);  // no column of the tile's (aligned) input window leaves the image
    for (int c = 0; c < 3; c++) {
        const float* xp = a.x + ((size_t)b * 3 + c) * HW;
        const float* yp = a.y + ((size_t)b * 3 + c) * HW;
        if (c > 0) __syncthreads();          // everyone is done with the previous channel's V pass
        // ---- H pass.  Rows outside the image are clamped to an image row and the item's outputs zeroed afterwards.
        for (int it = tid; it < TIN * (TS / 4); it += 256) {
            const int r = it >> 3, cg = it & 7;
            const int yy = oy0 - HALO + r, x0 = ox0 - HALO + 4 * cg;
            const bool rowin = (yy >= 0) && (yy < H);
            const int yc = yy < 0 ? 0 : (yy < H ? yy : H - 1);
            float xv[20], yv[20];
            load_row16<VEC>(xp + (long)yc * W, x0, W, colsafe, pairs, xv);
            load_row16<VEC>(yp + (long)yc * W, x0, W, colsafe, pairs, yv);
            float o[4][4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int j = 0; j < 4; j++) o[mi][j] = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float vx = xv[k + 2], vy = yv[k + 2];
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
        // the centre pixels of this channel (needed after the V pass): requested in front of its LDS reads
        float xc4[4] = {0.f, 0.f, 0.f, 0.f}, yc4[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE != MODE_MAP && MODE != MODE_GRAD && any) { load4<VEC>(xp + p0, inimg, xc4); load4<VEC>(yp + p0, inimg, yc4); }
        float gS4[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == MODE_GRAD && any) load4<VEC>(a.upstream + ((size_t)b * 3 + c) * HW + p0, inimg, gS4);
        // ---- V pass: output row rr = taps over the row-filtered rows rr .. rr + 12
        float mo[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int j = 0; j < 4; j++) mo[mi][j] = 0.f;
        // (a ROLLED loop over the taps: fully unrolled, the scheduler hoists all 52 16-byte reads -- 208 registers -- and spills)
#pragma unroll 1
        for (int i = 0; i < 13; i++) {
            const float g = gw.g[i];
#pragma unroll
            for (int mi = 0; mi < 4; mi++) {
                float4 v = *reinterpret_cast<const float4*>(&hb[mi][(rr + i) * TS + 4 * cq]);
                CC_KEEP4(v);
                mo[mi][0] = fmaf(g, v.x, mo[mi][0]);
                mo[mi][1] = fmaf(g, v.y, mo[mi][1]);
                mo[mi][2] = fmaf(g, v.z, mo[mi][2]);
                mo[mi][3] = fmaf(g, v.w, mo[mi][3]);
            }
        }
        // ---- per-pixel SSIM (ssim.py:20-34) and loss pieces
        if (!any) continue;
        float oA[4] = {0.f, 0.f, 0.f, 0.f}, oB[4] = {0.f, 0.f, 0.f, 0.f}, oC[4] = {0.f, 0.f, 0.f, 0.f}, oG[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!inimg[j]) continue;
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
                oA[j] = S;
                continue;
            }
            if (MODE == MODE_GRAD) {
                // adjoint maps of the SSIM map w.r.t. the SECOND image (call with swapped roles for the first)
                const float gS = gS4[j];
                // 1 / den2 = den1 / D, 1 / den1 = den2 / D
                const float id1 = den2 * iD, id2 = den1 * iD;
                oC[j] = gS * (2.f * num1 * iD);
                oB[j] = gS * (-S * id2);
                oA[j] = gS * (2.f * mu1 * (num2 - num1) * iD - 2.f * mu2 * S * (id1 - id2));
                continue;
            }
            const float xc = xc4[j], yc = yc4[j];
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
                oG[j] = -drob * vm;                             // d/dy through diff
                const float id1 = den2 * iD, id2 = den1 * iD;
                const float gS = -vm * a.wssim;                 // d(wssim * sl)/dS
                oC[j] = gS * (2.f * num1 * iD);                 // dS/dE[xy]
                oB[j] = gS * (-S * id2);                        // dS/dE[yy]
                oA[j] = gS * (2.f * mu1 * (num2 - num1) * iD - 2.f * mu2 * S * (id1 - id2));  // dS/dmu_y
                // d/d mask_b: diff and ssim_loss are both linear in the mask product
                acc_gm[j] += drob * (xc - yc) * valid[j] * ma[j] + a.wssim * (1.f - S * valid[j]) * ma[j];
            }
        }
        const size_t o = ((size_t)b * 3 + c) * HW + p0;
        if (MODE == MODE_MAP) store4<VEC>(a.out_map + o, inimg, oA);
        if (MODE == MODE_GRAD || (MODE == MODE_PHOTO && a.want_grad)) {
            store4<VEC>(a.adjA + o, inimg, oA);
            store4<VEC>(a.adjB + o, inimg, oB);
            store4<VEC>(a.adjC + o, inimg, oC);
            if (MODE == MODE_PHOTO) store4<VEC>(a.g0 + o, inimg, oG);
        }
    }

    if (MODE == MODE_ERR && any) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; j++)         // (1-wssim) * mean_c(robust) + wssim * mean_c(1 - ssim)   (torch mean over 3 channels = sum / 3)
            e[j] = (1.f - a.wssim) * (err_rob[j] / 3.f) + a.wssim * (err_ss[j] / 3.f);
        store4<VEC>(a.out_map + (size_t)b * HW + p0, inimg, e);
        store4<VEC>(a.out_valid + (size_t)b * HW + p0, inimg, valid);
    }
    if (MODE == MODE_PHOTO) {
        float s_valid = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) if (inimg[j]) s_valid += valid[j];
        if (a.want_grad && a.gmask && any) {
            float gm[4];
#pragma unroll
            for (int j = 0; j < 4; j++) gm[j] = a.b_complement ? -acc_gm[j] : acc_gm[j];
            store4<VEC>(a.gmask + (size_t)b * a.gm_bs + p0, inimg, gm);
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

// VEC path: every plane 16-byte aligned with a width that is a multiple of 4 (all pyramid levels of an 832-wide frame down to 52)
__device__ __forceinline__ bool ssim_vec_ok(const PhotoArgs& a) {
    uintptr_t u = (uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.mask_a | (uintptr_t)a.mask_b | (uintptr_t)a.upstream | (uintptr_t)a.out_map |
                  (uintptr_t)a.out_valid | (uintptr_t)a.adjA | (uintptr_t)a.adjB | (uintptr_t)a.adjC | (uintptr_t)a.g0 | (uintptr_t)a.gmask;
    return !(a.W & 3) && !(u & 15) && !((a.a_bs | a.b_bs | a.gm_bs) & 3);
}
template <int MODE>
__device__ __forceinline__ void ssim_tile_any(const PhotoArgs& a, const Gauss13& gw, int b, int tile_x, int tile_y, size_t blk) {
    __shared__ __attribute__((aligned(16))) float hb[4][TIN * TS];       // (one allocation for both instances of the body)
    __shared__ float red[4 * 3];
    if (ssim_vec_ok(a)) ssim_tile_body<MODE, true>(a, gw, b, tile_x, tile_y, blk, hb, red);
    else ssim_tile_body<MODE, false>(a, gw, b, tile_x, tile_y, blk, hb, red);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_ssim_tile(PhotoArgs a, Gauss13 gw) {
    ssim_tile_any<MODE>(a, gw, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y,
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
    ssim_tile_any<MODE_PHOTO>(a, gw, b, tile_x, tile_y, (size_t)local);
}

// err slots (consensus_exp_masks): 0 tgt, 1 warped, 2 err [B,1,H,W], 3 valid [B,1,H,W]
__global__ __launch_bounds__(256, 4) void k_ssim_err_jobs(JobTab t, float wssim, Gauss13 gw) {
    int j, b, tile_x, tile_y, local;
    tile_of(t, j, b, tile_x, tile_y, local);
    PhotoArgs a = {};
    a.H = t.H[j]; a.W = t.W[j];
    a.x = ccjobs::ptr<const float>(t, j, 0);
    a.y = ccjobs::ptr<const float>(t, j, 1);
    a.out_map = ccjobs::ptr<float>(t, j, 2);
    a.out_valid = ccjobs::ptr<float>(t, j, 3);
    a.wssim = wssim;
    ssim_tile_any<MODE_ERR>(a, gw, b, tile_x, tile_y, 0);
}

// gy = scale * (g0 + G*adjA + 2*y*(G*adjB) + x*(G*adjC)),  G* = zero-padded 13x13 Gaussian filter
// Round 6: the three adjoint tiles are staged with a pitch of 48 floats from the 16-byte aligned column ox0 - 8 (VEC: 12 x 16-byte
// loads per row instead of 44 4-byte ones), the vertical pass and the final expression work on (row, four consecutive columns) with
// 16-byte accesses to g0 / x / y / gy -- see ssim_tile_body.
constexpr int TP = 48;          // staged tile pitch: columns ox0 - 8 .. ox0 + 39
template <bool VEC>
__device__ __forceinline__ void ssim_adjoint_body(const float* __restrict__ adjA, const float* __restrict__ adjB,
                                                  const float* __restrict__ adjC, const float* __restrict__ g0,
                                                  const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ scale, float* __restrict__ gy, int H,
                                                  int W, int accumulate, const Gauss13& gw, int b, int tile_x, int tile_y,
                                                  float (*tin)[TIN * TP], float (*hb)[TIN * TS]) {
    const int HW = H * W;
    const int ox0 = tile_x * TS, oy0 = tile_y * TS;
    const int tid = threadIdx.x, rr = tid >> 3, cq = tid & 7, gx0 = ox0 + 4 * cq, gyy = oy0 + rr;
    const float sc = scale ? scale[0] : 1.f;
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ok[j] = (gx0 + j < W) && (gyy < H);
    for (int c = 0; c < 3; c++) {
        const size_t plane = ((size_t)b * 3 + c) * HW;
        if (c > 0) __syncthreads();
        if (VEC) {
            // 44 rows x 12 groups of four columns, three maps: the three 16-byte loads of a group, then its LDS stores (528 groups for
            // 256 work items: two rounds and a bit)
            for (int i = tid; i < TIN * (TP / 4); i += 256) {
                const int r = i / (TP / 4), g4 = i - r * (TP / 4);
                const int yy = oy0 - HALO + r, xx = ox0 - 8 + 4 * g4;
                const bool in = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
                const size_t o = plane + (in ? (size_t)yy * W + xx : 0);
                float4 ta = *reinterpret_cast<const float4*>(adjA + o), tb = *reinterpret_cast<const float4*>(adjB + o),
                       tc = *reinterpret_cast<const float4*>(adjC + o);
                if (!in) { ta = make_float4(0.f, 0.f, 0.f, 0.f); tb = ta; tc = ta; }
                *reinterpret_cast<float4*>(&tin[0][4 * i]) = ta;
                *reinterpret_cast<float4*>(&tin[1][4 * i]) = tb;
                *reinterpret_cast<float4*>(&tin[2][4 * i]) = tc;
            }
        } else {
            for (int i0 = tid; i0 < TIN * TP; i0 += 4 * 256) {        // 12 loads in flight per work item, then the LDS stores
                float va[4], vb[4], vc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + 256 * u;
                    const int r = i / TP, col = i - r * TP;
                    const int yy = oy0 - HALO + r, xx = ox0 - 8 + col;
                    const bool in = (i < TIN * TP) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
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
                    if (i < TIN * TP) { tin[0][i] = va[u]; tin[1][i] = vb[u]; tin[2][i] = vc[u]; }
                }
            }
        }
        // the final expression's operands of this channel: requested before the barriers
        float g04[4] = {0.f, 0.f, 0.f, 0.f}, x4[4] = {0.f, 0.f, 0.f, 0.f}, y4[4] = {0.f, 0.f, 0.f, 0.f}, old4[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t o4 = plane + (size_t)gyy * W + gx0;
        if (ok[0]) {
            if (g0) load4<VEC>(g0 + o4, ok, g04);
            load4<VEC>(x + o4, ok, x4);
            load4<VEC>(y + o4, ok, y4);
            if (accumulate) load4<VEC>(gy + o4, ok, old4);
        }
        __syncthreads();
        for (int it = tid; it < TIN * (TS / 4); it += 256) {
            const int r = it >> 3, cg = it & 7;
            // (scalar FMAs: on gfx950 a v_pk_fma_f32 costs more than the two v_fma_f32 it replaces -- the packed forward filter ran
            // 47 % slower with 25 % fewer VALU instructions, profiles/r04_ab_round4.txt)
            // input columns ox0 - 6 + 4 cg + k, k = 0..15 = staged columns 4 cg + 2 + k: five 16-byte reads from 4 cg
            float va[20], vb[20], vc[20];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                float4 qa = reinterpret_cast<const float4*>(&tin[0][r * TP + 4 * cg])[k];
                float4 qb = reinterpret_cast<const float4*>(&tin[1][r * TP + 4 * cg])[k];
                float4 qc = reinterpret_cast<const float4*>(&tin[2][r * TP + 4 * cg])[k];
                CC_KEEP4(qa); CC_KEEP4(qb); CC_KEEP4(qc);      // (whole 16-byte reads although the first / last group is half used)
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
                        oa[j] = fmaf(gw.g[t], va[k + 2], oa[j]);
                        ob[j] = fmaf(gw.g[t], vb[k + 2], ob[j]);
                        oc[j] = fmaf(gw.g[t], vc[k + 2], oc[j]);
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
        for (int mi = 0; mi < 3; mi++)
#pragma unroll
            for (int j = 0; j < 4; j++) mo[mi][j] = 0.f;
#pragma unroll 1
        for (int i = 0; i < 13; i++) {
            const float g = gw.g[i];
#pragma unroll
            for (int mi = 0; mi < 3; mi++) {
                float4 v = *reinterpret_cast<const float4*>(&hb[mi][(rr + i) * TS + 4 * cq]);
                CC_KEEP4(v);
                mo[mi][0] = fmaf(g, v.x, mo[mi][0]);
                mo[mi][1] = fmaf(g, v.y, mo[mi][1]);
                mo[mi][2] = fmaf(g, v.z, mo[mi][2]);
                mo[mi][3] = fmaf(g, v.w, mo[mi][3]);
            }
        }
        if (ok[0]) {
            float r4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float r = sc * ((g0 ? g04[j] : 0.f) + mo[0][j] + 2.f * y4[j] * mo[1][j] + x4[j] * mo[2][j]);
                r4[j] = accumulate ? old4[j] + r : r;
            }
            store4<VEC>(gy + o4, ok, r4);
        }
    }
}

__device__ __forceinline__ void ssim_adjoint_any(const float* adjA, const float* adjB, const float* adjC, const float* g0, const float* x,
                                                 const float* y, const float* scale, float* gy, int H, int W, int accumulate,
                                                 const Gauss13& gw, int b, int tile_x, int tile_y) {
    __shared__ __attribute__((aligned(16))) float tin[3][TIN * TP];
    __shared__ __attribute__((aligned(16))) float hb[3][TIN * TS];
    const uintptr_t u = (uintptr_t)adjA | (uintptr_t)adjB | (uintptr_t)adjC | (uintptr_t)g0 | (uintptr_t)x | (uintptr_t)y | (uintptr_t)gy;
    if (!(W & 3) && !(u & 15)) ssim_adjoint_body<true>(adjA, adjB, adjC, g0, x, y, scale, gy, H, W, accumulate, gw, b, tile_x, tile_y, tin, hb);
    else ssim_adjoint_body<false>(adjA, adjB, adjC, g0, x, y, scale, gy, H, W, accumulate, gw, b, tile_x, tile_y, tin, hb);
}

__global__ __launch_bounds__(256) void k_ssim_adjoint(const float* __restrict__ adjA, const float* __restrict__ adjB,
                                                      const float* __restrict__ adjC, const float* __restrict__ g0,
                                                      const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ scale, float* __restrict__ gy, int H,
                                                      int W, int accumulate, Gauss13 gw) {
    ssim_adjoint_any(adjA, adjB, adjC, g0, x, y, scale, gy, H, W, accumulate, gw, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y);
}

// adjoint slots: 0 adjoint maps (adjA, adjB, adjC, g0), 1 tgt, 2 warped, 3 scale (1 float), 4 gwarped [B,3,H,W]
__global__ __launch_bounds__(256, 4) void k_ssim_adjoint_jobs(JobTab t, Gauss13 gw) {
    int j, b, tile_x, tile_y, local;
    tile_of(t, j, b, tile_x, tile_y, local);
    const int H = t.H[j], W = t.W[j];
    const float* adj = ccjobs::ptr<const float>(t, j, 0);
    const size_t map = (size_t)t.B * 3 * H * W;
    ssim_adjoint_any(adj, adj + map, adj + 2 * map, adj + 3 * map, ccjobs::ptr<const float>(t, j, 1), ccjobs::ptr<const float>(t, j, 2),
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
