// "Job table" launches: ONE kernel launch over a list of same-kind work items of different sizes -- the (pyramid level,
// reference frame) terms of a loss (loss_functions.py loops `for scale ... for ref ...`, 24-36 terms per loss per step).
// A job = 8 pointer-sized slots + (H, W); the table travels in the kernel arguments (< 2 KB), blocks are dealt to jobs by a
// prefix table, so the six pyramid levels (212 992 ... 208 pixels per image) share one well-filled grid instead of six
// launches of which the small ones are pure launch latency.
#pragma once
#include "cc_common.h"

namespace ccjobs {

constexpr int MAXJOBS = 24;
constexpr int SLOTS = 8;
constexpr int JOB_LONGS = SLOTS + 2;        // host layout of one job: slots[8], H, W

struct JobTab {
    int n, B;
    int blk_end[MAXJOBS];       // cumulative block count (job j owns blocks [blk_end[j-1], blk_end[j]))
    int H[MAXJOBS], W[MAXJOBS];
    long slot[MAXJOBS][SLOTS];
};

// blocks per (job, batch item): pixel kernels use ceil(H*W / 256); tile kernels ceil(W/T)*ceil(H/T)
inline int pix_blocks(int H, int W) { return (H * W + 255) / 256; }

// fill a table from the host job array; blocks_per_image(H, W) -> blocks one image of that size needs; -> total blocks
template <class F>
inline int fill(JobTab& t, const long* jobs, int njobs, int B, F blocks_per_image) {
    t.n = njobs;
    t.B = B;
    int tot = 0;
    for (int j = 0; j < njobs; j++) {
        const long* q = jobs + (long)j * JOB_LONGS;
        for (int k = 0; k < SLOTS; k++) t.slot[j][k] = q[k];
        t.H[j] = (int)q[SLOTS];
        t.W[j] = (int)q[SLOTS + 1];
        tot += B * blocks_per_image(t.H[j], t.W[j]);
        t.blk_end[j] = tot;
    }
    return tot;
}

// block index -> (job, first block of the job).  n <= 24: a scalar linear scan
__device__ __forceinline__ int find(const JobTab& t, int bx, int& first) {
    int j = 0;
    first = 0;
    for (int q = 0; q + 1 < t.n; q++)
        if (bx >= t.blk_end[q]) { j = q + 1; first = t.blk_end[q]; }
    return j;
}

// ... -> (job, index of the block INSIDE the job in XCD order, cc_common.h): a job's blocks are consecutive pieces of its images, so
// every XCD works through a contiguous band of rows -- the rows a bilinear gather or a stencil shares with the piece above / below
// are then fetched into ONE L2 (the warp kernels moved 1.3-1.7x their distinct bytes with the blocks dealt round-robin)
__device__ __forceinline__ int find_xcd(const JobTab& t, int bx, int& local) {
    int first;
    const int j = find(t, bx, first);
    local = (CC_XCD_MASK & 8) ? cc_xcd_order(bx - first, t.blk_end[j] - first) : bx - first;
    return j;
}

template <class T>
__device__ __forceinline__ T* ptr(const JobTab& t, int j, int k) { return reinterpret_cast<T*>(t.slot[j][k]); }

}  // namespace ccjobs
