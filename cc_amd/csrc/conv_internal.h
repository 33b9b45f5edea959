// Internal (non-ABI) entry points shared between the convolution translation units.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

namespace ccint {

// "thin" weight-gradient (few channels, very many pixels): wgrad_thin.hip
// -> workspace floats needed, or 0 when the geometry is not eligible (pad / input size are checked at launch).
size_t wgrad_thin_ws_floats(int B, int M, int AH, int AW, int Cin, int R, int S, int si);
// -> true when it launched (same argument meaning as cc_conv2d_wgrad), false when not eligible (nothing launched).
bool wgrad_thin_launch(const float* a, const float* x, float* gw, float* ws, int B, int M, int AH, int AW, long a_bs, int Cin,
                       int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate, hipStream_t s);

// name of the kernel wgrad_thin_launch would use ("" when not eligible)
void wgrad_thin_name(int B, int M, int AH, int AW, int Cin, int IH, int IW, int R, int S, int si, int pad, char* out, int cap);

}  // namespace ccint
