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

// Raw buffer loads (wino.hip): a 128-bit buffer resource over [ptr, ptr + bytes) lets the hardware bounds-check every lane --
// a byte offset >= `bytes` (CC_BUF_OOB) reads 0.0f, so zero padding at the image border costs no compare / select.
// voff: per-lane byte offset (VGPR), soff: wave-uniform byte offset (SGPR, not part of the range check).
#ifndef CC_BUF_RSRC
typedef __amdgpu_buffer_rsrc_t cc_buf_t;
#define CC_BUF_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define CC_BUF_LOAD_F32(rsrc, voff, soff) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rsrc), (int)(voff), (int)(soff), 0))
typedef unsigned cc_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cc_buf_load_f32x2(cc_buf_t rsrc, unsigned voff, unsigned soff) {
    const cc_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
#define CC_BUF_LOAD_F32X2(rsrc, voff, soff) cc_buf_load_f32x2((rsrc), (voff), (soff))
#endif
#define CC_BUF_OOB 0x80000000u

// One-instruction reciprocal / square root (v_rcp_f32, v_sqrt_f32: 1 ulp) for arguments known to be normal numbers -- the IEEE
// division and sqrtf sequences cost ~10 VALU instructions each (scaling, Newton steps, fix-up of denormals and specials).
__device__ __forceinline__ float cc_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
__device__ __forceinline__ float cc_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CC_HIPEMU)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

// Four 16-byte LDS-DMA transfers per lane (1 KB per wave each), rows 1 KB apart in BOTH global memory and LDS (the immediate
// offset of global_load_lds applies to both addresses), issued from inline assembly: hipcc waits vmcnt(0) before the next ds_read
// whenever a compiler-visible LDS-DMA is in flight (it cannot tell which LDS bytes the DMA writes), which serialises a
// double-buffered stage; an asm DMA is outside its bookkeeping -- the kernel waits for it itself (CC_WAIT_VMCNT0 + barrier before
// the buffer is read).  lds_dst: wave-uniform LDS byte address (M0), gsrc: this lane's source.
#ifndef CC_GLDS16X4
__device__ __forceinline__ void cc_glds16x4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
#define CC_GLDS16X4(gsrc, lds_ptr) cc_glds16x4((gsrc), __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)CC_LDS_PTR(lds_ptr)))
#endif

// The same four transfers with a wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit byte offset PER ROW: the rows still land
// 1 KB apart in LDS (the immediate offset applies to both sides), but their global sources are voff[k] + 1024 k bytes behind the
// base, i.e. wherever the caller points them (wino.hip small-tile kernel: 512-byte halves of the 1 KB rows of a 64-row weight block).
#ifndef CC_GLDS16X4_S
__device__ __forceinline__ void cc_glds16x4_s(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
        "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
        "global_load_lds_dwordx4 %4, %5 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ const void* cc_uniform_ptr(const void* p) {      // a pointer the compiler can keep in an SGPR pair
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
#define CC_GLDS16X4_S(sbase, voff, lds_ptr)                                                         \
    cc_glds16x4_s(cc_uniform_ptr(sbase), (voff)[0], (voff)[1], (voff)[2], (voff)[3],                \
                  __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)CC_LDS_PTR(lds_ptr)))
#endif

// One 16-byte LDS-DMA transfer per lane through a buffer resource (bounds-checked: lanes whose offset lies at or beyond the
// resource's size move zeros), from inline assembly for the same reason as CC_GLDS16X4.  lds_ptr: wave-uniform destination (lane l
// lands at lds_ptr + 16 l bytes); voff: per-lane byte offset; soff: wave-uniform byte offset (not range-checked).
#ifndef CC_BUF_GLDS16
__device__ __forceinline__ void cc_buf_glds16(cc_buf_t rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff)
        : "memory");
}
#define CC_BUF_GLDS16(rsrc, voff, soff, lds_ptr) \
    cc_buf_glds16((rsrc), (voff), __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)CC_LDS_PTR(lds_ptr)))
// s_waitcnt vmcnt(0) the compiler cannot move LDS reads across (for data that arrived by an asm LDS-DMA)
#define CC_WAIT_VMCNT0_FENCE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// ... all but the N most recently issued VMEM operations have completed (memory reads return in issue order)
#define CC_WAIT_VMCNT_FENCE(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#endif
// s_waitcnt vmcnt(0) with expcnt/lgkmcnt left at their maxima (gfx9 encoding)
#define CC_WAIT_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)

// XCD-aware work order.  The workgroups of a launch are dealt round-robin to the 8 XCDs of the chip (private 4 MB L2 each): workgroups
// b, b + 8, b + 16, ... share an L2.  cc_xcd_order(b, n) gives workgroup b of n the work item (b & 7) * (n / 8) + (b >> 3), so every XCD
// works through ONE CONTIGUOUS run of items: neighbouring tiles (shared halo rows), the channel blocks of one tile (same input patch)
// or the blocks of one reduction range (same operand slices) meet in one L2 instead of being fetched from HBM by up to eight.  A
// bijection of [0, n) (the n % 8 items past the last whole group of 8 keep their index); results do not depend on it.
// CC_XCD_MASK (A/B builds): bit 0 direct conv kernels, bit 1 Winograd weight gradient, bit 2 generic weight gradient, bit 3 the
// job-table warp / smoothness kernels (the SSIM tiles and the Winograd forward kernel have orders of their own).
#ifndef CC_XCD_MASK
#define CC_XCD_MASK 15
#endif
__device__ __forceinline__ int cc_xcd_order(int b, int n) {
    const int cnt = n >> 3;
    return b < (cnt << 3) ? (b & 7) * cnt + (b >> 3) : b;
}

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
