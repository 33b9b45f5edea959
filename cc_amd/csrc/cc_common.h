// Shared device helpers for the ccengine HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CC_OK 0
#define CC_ERR_ARG (-1)
#define CC_ERR_LAUNCH (-2)

#define CC_CHECK_LAUNCH()                                   \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return CC_ERR_LAUNCH; \
    } while (0)

// address-space casts for __builtin_amdgcn_global_load_lds (LDS-DMA)
#ifndef CC_LDS_PTR
#define CC_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CC_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#endif
// keep all four lanes of a float4 LDS read alive so that hipcc emits ONE ds_read_b128 (it otherwise narrows a partially
// used vector to ds_read_b32 / ds_read2_b32, whose 32-bank rule conflicts on layouts tuned for the b128 64-bank rule)
#ifndef CC_KEEP4
#define CC_KEEP4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#endif
// s_waitcnt vmcnt(0) with expcnt/lgkmcnt left at their maxima (gfx9 encoding)
#define CC_WAIT_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)

namespace cc {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Sum `N` per-thread values over a 256-thread workgroup; the totals land in
// out[0..N) of thread 0 (valid for threadIdx.x == 0 only).  `scratch` needs 4*N floats.
template <int N>
__device__ __forceinline__ void block_sum_256(float (&v)[N], float* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) scratch[wid * N + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = (scratch[i] + scratch[N + i]) + (scratch[2 * N + i] + scratch[3 * N + i]);
    }
    __syncthreads();
}

}  // namespace cc
