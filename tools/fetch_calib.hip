// Round 6: calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS PATTERN.
// /opt/skills/guides/MI355X_MICROARCH.md (section HBM) calibrates one case only -- FETCH_SIZE reports half the bytes of a 16 B/lane
// coalesced streaming read -- and says "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your
// own access pattern".  These micro-kernels move a KNOWN number of bytes in the access patterns of the step's HBM-bound kernels:
//   k_read16 / k_read8 / k_read4   coalesced streaming reads, 16 / 8 / 4 bytes per lane          (Adam, epilogues / SSIM rows / job kernels)
//   k_gather4                      four 4-byte bilinear taps per pixel at a displaced coordinate   (the warp kernels)
//   k_write16 / k_write4           coalesced streaming stores, 16 / 4 bytes per lane
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/fetch_calib.py divides the known bytes by
// the counters -> profiles/fetch_calib.json, which tools/pmc_traffic.py applies per kernel class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read16(const float4* __restrict__ in, float* __restrict__ out, long n4) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_read8(const float2* __restrict__ in, float* __restrict__ out, long n2) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) { float2 v = in[i]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_read4(const float* __restrict__ in, float* __restrict__ out, long n) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += in[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
// planes of H x W floats; every pixel reads the 2 x 2 neighbourhood at (x + dx, y + dy) with a smooth displacement of a few pixels
// (what a rigid / flow warp does); every input element is touched ~4 times through L1 / L2, the distinct bytes are the planes once
__global__ __launch_bounds__(256) void k_gather4(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W) {
    const long npx = (long)H * W;
    float acc = 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npx * planes; p += (long)gridDim.x * 256) {
        const int pl = (int)(p / npx);
        const int r = (int)(p - (long)pl * npx);
        const int y = r / W, x = r - y * W;
        int x0 = x + ((y >> 4) & 3) - 1, y0 = y + ((x >> 5) & 3) - 1;
        x0 = x0 < 0 ? 0 : (x0 > W - 2 ? W - 2 : x0);
        y0 = y0 < 0 ? 0 : (y0 > H - 2 ? H - 2 : y0);
        const float* b = in + (long)pl * npx + (long)y0 * W + x0;
        acc += b[0] + b[1] + b[W] + b[W + 1];
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_write16(float4* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_write4(float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = 1.f;
}

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 512;            // working set per kernel
    const long n = mb * 1024 * 1024 / 4;
    float *a, *o;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&o, 1 << 20));
    CK(hipMemset(a, 0, n * 4));
    const int H = 256, W = 832, planes = (int)(n / ((long)H * W));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const float4*)a, o, n / 4);
        hipLaunchKernelGGL(k_read8, dim3(grid), dim3(256), 0, 0, (const float2*)a, o, n / 2);
        hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, (const float*)a, o, n);
        hipLaunchKernelGGL(k_gather4, dim3(grid), dim3(256), 0, 0, (const float*)a, o, planes, H, W);
        hipLaunchKernelGGL(k_write16, dim3(grid), dim3(256), 0, 0, (float4*)a, n / 4);
        hipLaunchKernelGGL(k_write4, dim3(grid), dim3(256), 0, 0, a, n);
    }
    CK(hipDeviceSynchronize());
    // known bytes per launch (tools/fetch_calib.py reads this line)
    printf("CALIB bytes k_read16 %ld k_read8 %ld k_read4 %ld k_gather4 %ld k_write16 %ld k_write4 %ld\n", n * 4, n * 4, n * 4,
           (long)planes * H * W * 4, n * 4, n * 4);
    return 0;
}
