// TEST INFRASTRUCTURE ONLY -- a host-side emulation of the small HIP subset the
// kernels under cc_amd/csrc use, so that the *unchanged* kernel sources can be
// compiled for x86 (clang++ -x c++ -I tests/hipemu/shim ...) and checked against
// the oracle on a box without a GPU.  The product never sees this header: the
// shipped libccengine.so is built by hipcc against the real <hip/hip_runtime.h>
// and cc_amd/_lib.py loads nothing else.
//
// Model: one OS thread runs one workgroup at a time; every work-item is a fiber
// (hand-rolled x86-64 context switch).  __syncthreads() and the wave-level
// collectives (__shfl*, __ballot, MFMA) are generation barriers on which fibers
// spin-yield.  Workgroups of a launch are distributed over OpenMP threads.
// Wave = 64 consecutive linear thread ids.  MFMA lane layouts follow
// /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 16; return hipSuccess; }      // few "CUs": persistent kernels loop
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

namespace hipemu {

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    uint3 tid;
    unsigned lin = 0;
};

struct WaveState {
    unsigned gen = 0, arrived = 0, live = 0;
    uint64_t slot[64];
    float a[64], b[64];
};

struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    unsigned gen = 0, arrived = 0, live = 0;
    uint3 bid;
    dim3 bdim, gdim;
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    std::function<void()>* body = nullptr;
    std::vector<char> dyn;
    size_t stack_bytes = 256 * 1024;
};

inline BlockCtx*& ctxp() { static thread_local BlockCtx* p = nullptr; return p; }
inline BlockCtx& ctx() { return *ctxp(); }
inline void* dyn_smem() { return ctx().dyn.data(); }

extern "C" void hipemu_switch(void** save_sp, void* new_sp);
#ifdef HIPEMU_IMPL
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");
#endif

inline void yield() { BlockCtx& c = ctx(); hipemu_switch(&c.cur->sp, c.sched_sp); }

inline void fiber_entry() {
    BlockCtx& c = ctx();
    (*c.body)();
    Fiber* f = c.cur;
    f->done = true;
    c.live--;
    if (c.live > 0 && c.arrived == c.live) { c.arrived = 0; c.gen++; }
    WaveState& w = c.waves[f->lin / 64];
    w.live--;
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
    hipemu_switch(&f->sp, c.sched_sp);
    abort();
}

inline void block_barrier() {
    BlockCtx& c = ctx();
    unsigned g = c.gen;
    if (++c.arrived == c.live) { c.arrived = 0; c.gen++; return; }
    while (c.gen == g) yield();
}

inline void wave_barrier() {
    BlockCtx& c = ctx();
    WaveState& w = c.waves[c.cur->lin / 64];
    unsigned g = w.gen;
    if (++w.arrived == w.live) { w.arrived = 0; w.gen++; return; }
    while (w.gen == g) yield();
}

inline unsigned lane_id() { return ctx().cur->lin & 63; }
inline WaveState& wave() { BlockCtx& c = ctx(); return c.waves[c.cur->lin / 64]; }

template <typename T> inline T shfl_from(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl type");
    WaveState& w = wave();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    w.slot[lane_id()] = bits;
    wave_barrier();
    uint64_t r = w.slot[src & 63];
    wave_barrier();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}

void run_block(BlockCtx& c, std::function<void()>& body);
void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);

#ifdef HIPEMU_IMPL
static void trampoline() { fiber_entry(); }

void run_block(BlockCtx& c, std::function<void()>& body) {
    unsigned n = c.bdim.x * c.bdim.y * c.bdim.z;
    c.body = &body;
    c.gen = c.arrived = 0;
    c.live = n;
    unsigned nw = (n + 63) / 64;
    c.waves.assign(nw, WaveState());
    for (unsigned i = 0; i < n; i++) c.waves[i / 64].live++;
    if (c.fibers.size() < n) {
        size_t old = c.fibers.size();
        c.fibers.resize(n);
        for (size_t i = old; i < n; i++) c.fibers[i].stack = (char*)aligned_alloc(64, c.stack_bytes);
    }
    for (unsigned i = 0; i < n; i++) {
        Fiber& f = c.fibers[i];
        f.done = false;
        f.lin = i;
        f.tid.x = i % c.bdim.x;
        f.tid.y = (i / c.bdim.x) % c.bdim.y;
        f.tid.z = i / (c.bdim.x * c.bdim.y);
        // initial frame: 6 callee-saved regs, return address = trampoline, then alignment slot
        uintptr_t top = ((uintptr_t)(f.stack + c.stack_bytes)) & ~(uintptr_t)63;
        void** sp = (void**)top;
        *(--sp) = nullptr;                 // fake return address of trampoline (keeps rsp%16==8 at entry)
        *(--sp) = (void*)&trampoline;      // 'ret' target
        for (int k = 0; k < 6; k++) *(--sp) = nullptr;
        f.sp = sp;
    }
    while (c.live > 0) {
        for (unsigned i = 0; i < n; i++) {
            Fiber& f = c.fibers[i];
            if (f.done) continue;
            c.cur = &f;
            hipemu_switch(&c.sched_sp, f.sp);
        }
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    long nblocks = (long)grid.x * grid.y * grid.z;
    #pragma omp parallel
    {
        static thread_local BlockCtx* mine = nullptr;
        if (!mine) mine = new BlockCtx();
        BlockCtx& c = *mine;
        ctxp() = &c;
        c.bdim = block;
        c.gdim = grid;
        c.dyn.assign(shmem + 64, 0);
        #pragma omp for schedule(dynamic, 1)
        for (long b = 0; b < nblocks; b++) {
            c.bid.x = b % grid.x;
            c.bid.y = (b / grid.x) % grid.y;
            c.bid.z = b / ((long)grid.x * grid.y);
            run_block(c, body);
        }
    }
}
#endif
}  // namespace hipemu

#define threadIdx (hipemu::ctx().cur->tid)
#define blockIdx (hipemu::ctx().bid)
#define blockDim (hipemu::ctx().bdim)
#define gridDim (hipemu::ctx().gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::block_barrier(); }
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::lane_id();
    return hipemu::shfl_from(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) {
    return hipemu::shfl_from(v, hipemu::lane_id() ^ m);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::lane_id();
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::shfl_from(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = hipemu::lane_id();
    int src = lane - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::shfl_from(v, src);
}
static inline unsigned long long __ballot(int pred) {
    hipemu::WaveState& w = hipemu::wave();
    w.slot[hipemu::lane_id()] = pred ? 1 : 0;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    hipemu::BlockCtx& c = hipemu::ctx();
    unsigned base = (c.cur->lin / 64) * 64;
    unsigned n = c.bdim.x * c.bdim.y * c.bdim.z;
    for (unsigned l = 0; l < 64 && base + l < n; l++)
        if (!c.fibers[base + l].done && w.slot[l]) m |= 1ull << l;
    hipemu::wave_barrier();
    return m;
}

// ---- atomics (workgroups may run on different OS threads)
static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&nw, &f, 4);
    } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline long long __float2ll_rn(float f) { return (long long)llrintf(f); }
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- math / bit casts
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __fdividef(float a, float b) { return a / b; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __powf(a, b) powf(a, b)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline float __ldg(const float* p) { return *p; }
#ifndef __clang__
#error "hipemu needs clang++ (ext_vector_type)"
#endif

// ---- MFMA f32 (guide section 3: A[i][k] in lane i + 32k, B[k][j] in lane j + 32k;
//      D row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31; k-ordered fmaf chain)
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c) {
    hipemu::WaveState& w = hipemu::wave();
    unsigned lane = hipemu::lane_id();
    w.a[lane] = a;
    w.b[lane] = b;
    hipemu::wave_barrier();
    hipemu_f32x16 d;
    unsigned col = lane & 31;
    for (unsigned r = 0; r < 16; r++) {
        unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (unsigned k = 0; k < 2; k++) acc = fmaf(w.a[row + 32 * k], w.b[col + 32 * k], acc);
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
static inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c) {
    hipemu::WaveState& w = hipemu::wave();
    unsigned lane = hipemu::lane_id();
    w.a[lane] = a;
    w.b[lane] = b;
    hipemu::wave_barrier();
    hipemu_f32x4 d;
    unsigned col = lane & 15;
    for (unsigned r = 0; r < 4; r++) {
        unsigned row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (unsigned k = 0; k < 4; k++) acc = fmaf(w.a[row + 16 * k], w.b[col + 16 * k], acc);
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4(a, b, c)
#define __builtin_amdgcn_readfirstlane(x) (x)
// LDS-DMA: destination = wave-uniform base + lane * size (guide section 5); executes synchronously here
#define CC_LDS_PTR(p) ((void*)(p))
#define CC_KEEP4(v) ((void)0)
#define CC_GLOBAL_PTR(p) ((const void*)(p))
static inline void hipemu_glds(const void* src, void* dst, unsigned size) {
    memcpy(static_cast<char*>(dst) + (size_t)hipemu::lane_id() * size, src, size);
}
#define __builtin_amdgcn_global_load_lds(src, dst, size, off, aux) hipemu_glds(src, dst, size)
// raw buffer loads with hardware bounds check (cc_common.h): offsets at or beyond the resource's size read 0
struct cc_buf_t { const char* base; unsigned bytes; };
#define CC_BUF_RSRC(ptr, nbytes) (cc_buf_t{reinterpret_cast<const char*>(ptr), (unsigned)(nbytes)})
static inline float hipemu_buf_load(cc_buf_t r, unsigned voff, unsigned soff) {
    if (voff >= r.bytes) return 0.f;
    float v;
    memcpy(&v, r.base + (size_t)voff + soff, 4);
    return v;
}
#define CC_BUF_LOAD_F32(rsrc, voff, soff) hipemu_buf_load((rsrc), (unsigned)(voff), (unsigned)(soff))
static inline float2 hipemu_buf_load2(cc_buf_t r, unsigned voff, unsigned soff) {
    float2 v{0.f, 0.f};
    if (voff + 4 < r.bytes) memcpy(&v, r.base + (size_t)voff + soff, 8);
    return v;
}
#define CC_BUF_LOAD_F32X2(rsrc, voff, soff) hipemu_buf_load2((rsrc), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
// four LDS-DMA rows, 1 KB apart on both sides (cc_common.h CC_GLDS16X4)
static inline void hipemu_glds16x4(const void* src, void* dst) {
    for (int i = 0; i < 4; i++)
        memcpy(static_cast<char*>(dst) + 1024 * i + (size_t)hipemu::lane_id() * 16, static_cast<const char*>(src) + 1024 * i, 16);
}
#define CC_GLDS16X4(gsrc, lds_ptr) hipemu_glds16x4((gsrc), (lds_ptr))
// ... with a uniform base and one per-lane byte offset per row (cc_common.h CC_GLDS16X4_S): row k reads base + voff[k] + 1024 k
static inline void hipemu_glds16x4_s(const void* base, const unsigned* voff, void* dst) {
    for (int i = 0; i < 4; i++)
        memcpy(static_cast<char*>(dst) + 1024 * i + (size_t)hipemu::lane_id() * 16, static_cast<const char*>(base) + voff[i] + 1024 * i, 16);
}
#define CC_GLDS16X4_S(sbase, voff, lds_ptr) hipemu_glds16x4_s((sbase), (voff), (lds_ptr))
// bounds-checked 16-byte LDS-DMA (cc_common.h CC_BUF_GLDS16): out-of-range lanes move zeros
static inline void hipemu_buf_glds16(cc_buf_t r, unsigned voff, unsigned soff, void* dst) {
    char* d = static_cast<char*>(dst) + (size_t)hipemu::lane_id() * 16;
    if (voff >= r.bytes) memset(d, 0, 16);
    else memcpy(d, r.base + (size_t)voff + soff, 16);
}
#define CC_BUF_GLDS16(rsrc, voff, soff, lds_ptr) hipemu_buf_glds16((rsrc), (unsigned)(voff), (unsigned)(soff), (lds_ptr))
#define CC_WAIT_VMCNT_FENCE(N) hipemu::wave_barrier()
#define CC_WAIT_VMCNT0_FENCE() hipemu::wave_barrier()      // lanes run one after the other here: the wave must have issued its DMA
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define CC_HIPEMU 1
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
#define __builtin_amdgcn_wave_barrier() hipemu::wave_barrier()
