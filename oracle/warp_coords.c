/* TEST INFRASTRUCTURE ONLY (oracle) -- plain-C restatement of the reference's sampling-coordinate arithmetic
 * (inverse_warp.py:31-45 pixel2cam, :48-79 cam2pixel, the grid_sample unnormalisation of ATen's CPU kernel),
 * following the exact-rounding recipe of SURVEY.md appendix D: IEEE fp32, fmaf() exactly where the CPU reference's
 * bmm fuses and nowhere else (build with -ffp-contract=off).  Unlike the torch restatement (oracle/geometry.py),
 * whose bmm rounding depends on the host CPU's BLAS kernels, this gives the same bits on every host; it is pinned
 * to the reference by the `grid_zeros` / `pose2flow` fixtures (tests/test_oracle_golden.py::test_c_warp_coords).
 */
#include <math.h>

/* grid[b][y][x][2] (normalised, 'zeros' mode rewrites OOB to 2), flow[b][2][y][x] (pose2flow, no rewrite),
 * tap[b][y][x][2] = floor of the un-normalised source index (align_corners = ac) */
void cc_oracle_warp_coords(const float* depth, const float* P, const float* Kinv, int B, int H, int W, int ac,
                           float* grid, float* flow, int* tap) {
    for (int b = 0; b < B; b++) {
        const float* p = P + 12 * b;
        const float* k = Kinv + 9 * b;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float d = depth[((long)b * H + y) * W + x];
                float cam[3], q[3];
                for (int i = 0; i < 3; i++)
                    cam[i] = fmaf(k[3 * i + 2], 1.0f, fmaf(k[3 * i + 1], (float)y, k[3 * i] * (float)x)) * d;
                for (int i = 0; i < 3; i++)
                    q[i] = fmaf(p[4 * i + 2], cam[2], fmaf(p[4 * i + 1], cam[1], p[4 * i] * cam[0])) + p[4 * i + 3];
                const float Z = q[2] < 1e-3f ? 1e-3f : q[2];
                float xn = (2.0f * (q[0] / Z)) / (float)(W - 1) - 1.0f;
                float yn = (2.0f * (q[1] / Z)) / (float)(H - 1) - 1.0f;
                const long o = ((long)b * H + y) * W + x;
                if (flow) {
                    flow[((long)b * 2 + 0) * H * W + (long)y * W + x] = (float)(W - 1) * (xn / 2.0f + 0.5f) - (float)x;
                    flow[((long)b * 2 + 1) * H * W + (long)y * W + x] = (float)(H - 1) * (yn / 2.0f + 0.5f) - (float)y;
                }
                if (xn > 1.f || xn < -1.f) xn = 2.f;
                if (yn > 1.f || yn < -1.f) yn = 2.f;
                if (grid) { grid[2 * o] = xn; grid[2 * o + 1] = yn; }
                if (tap) {
                    const float ix = ac ? ((xn + 1.f) * 0.5f) * (float)(W - 1) : fmaf(xn + 1.f, (float)W * 0.5f, -0.5f);
                    const float iy = ac ? ((yn + 1.f) * 0.5f) * (float)(H - 1) : fmaf(yn + 1.f, (float)H * 0.5f, -0.5f);
                    tap[2 * o] = (int)floorf(ix);
                    tap[2 * o + 1] = (int)floorf(iy);
                }
            }
    }
}
