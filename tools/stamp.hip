// Diagnosis only (tools/branch_timeline.py): one-thread kernels that write the GPU's constant-rate clock (100 MHz) into a buffer,
// launched on whatever stream the caller names -- inside a stream capture they become nodes of the graph, so a replayed step leaves
// a time stamp at every point of interest without a profiler attached.
#include <hip/hip_runtime.h>
__global__ void k_stamp(unsigned long long* dst) { *dst = wall_clock64(); }
extern "C" int stamp_launch(unsigned long long* dst, void* stream) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
