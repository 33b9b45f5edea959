// Probe: what fp32 MFMA rate does a 6-wave / 3-accumulator loop reach with (a) registers only, (b) + ds_read_b128 per
// 12 MFMAs (as k_wgrad3x3), (c) + barrier per 96 MFMAs?   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define KEEP4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))

template <int MODE, int NACC>
__global__ __launch_bounds__(384) void probe(float* out, int iters) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 384) lds[i] = make_float4(i * 1e-3f, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    float4 a = lds[lane], w0 = lds[lane + 64], w1 = lds[lane + 128], w2 = lds[lane + 192];
    for (int it = 0; it < iters; it++) {
        for (int pq = 0; pq < 8; pq++) {
            if (MODE >= 1) {
                const int o = ((it * 8 + pq) * 64 + lane * 41) & 1023;
                a = lds[o ^ (lane & 15)]; w0 = lds[1024 + o]; w1 = lds[1024 + ((o + 1) & 1023)]; w2 = lds[1024 + ((o + 2) & 1023)];
                KEEP4(a); KEEP4(w0); KEEP4(w1); KEEP4(w2);
            }
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], w[3 + (j % 3) + e], acc[j], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int j = 0; j < NACC; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 384 + tid] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks, int threads) {
    float* out; hipMalloc(&out, (size_t)blocks * 384 * 4);
    const int iters = 200;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(probe<MODE, NACC>), dim3(blocks), dim3(threads), 65536, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(probe<MODE, NACC>), dim3(blocks), dim3(threads), 65536, 0, out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double mf = (double)blocks * (threads / 64) * iters * 8 * 4 * NACC;
    printf("%-34s blocks %4d thr %3d  %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, ms, mf * 4096.0 / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0, 3>("regs only, 3 acc, 256 thr", 1024, 256);
    run<0, 4>("regs only, 4 acc, 384 thr", 512, 384);
    run<0, 4>("regs only, 4 acc, 384 thr 1024blk", 1024, 384);
    run<0, 6>("regs only, 6 acc, 256 thr", 1024, 256);
    run<0, 9>("regs only, 9 acc, 256 thr", 1024, 256);
    run<1, 9>("+reads, 9 acc, 256 thr", 1024, 256);
    run<0, 3>("regs only, 3 acc, 192 thr", 1024, 192);
    run<0, 2>("regs only, 2 acc, 256 thr", 1024, 256);
    run<0, 3>("regs only, 3 acc", 512, 384);
    run<0, 3>("regs only, 3 acc, 1024 blk", 1024, 384);
    run<0, 4>("regs only, 4 acc, 256 thr", 1024, 256);
    run<1, 3>("+4 b128 reads /12 mfma", 512, 384);
    run<2, 3>("+reads +barrier/96 mfma", 512, 384);
    run<1, 4>("+reads, 4 acc, 256 thr", 1024, 256);
    run<2, 4>("+reads +barrier, 4 acc 256 thr", 1024, 256);
    return 0;
}
