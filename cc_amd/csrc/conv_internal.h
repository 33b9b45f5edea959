// Internal (non-ABI) entry points shared between the convolution translation units.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

namespace ccint {

// second stage of the split weight-gradient kernels (wgrad_reduce.hip): host descriptors of RD_LONGS longs each
constexpr int RD_LONGS = 16;
struct RedSink { long* out; int cap; int n; };      // deferred mode: descriptors are appended here instead of launched
// sink == nullptr: reduce these n problems now (one launch per 32); else append them to the sink.  -> CC_OK / CC_ERR_ARG
int wgrad_reduce_emit(RedSink* sink, const long* desc, int n, hipStream_t s);
int wgrad_reduce_launch(const long* desc, int n, hipStream_t s);

// "thin" weight-gradient (few channels, very many pixels): wgrad_thin.hip
// -> workspace floats needed, or 0 when the geometry is not eligible (pad / input size are checked at launch).
size_t wgrad_thin_ws_floats(int B, int M, int AH, int AW, int Cin, int R, int S, int si);
// -> true when it launched (same argument meaning as cc_conv2d_wgrad), false when not eligible (nothing launched).
bool wgrad_thin_launch(const float* a, const float* x, float* gw, float* ws, int B, int M, int AH, int AW, long a_bs, int Cin,
                       int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate, hipStream_t s,
                       RedSink* sink = nullptr);

// name of the kernel wgrad_thin_launch would use ("" when not eligible)
void wgrad_thin_name(int B, int M, int AH, int AW, int Cin, int IH, int IW, int R, int S, int si, int pad, char* out, int cap);

}  // namespace ccint
