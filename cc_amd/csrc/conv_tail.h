// Activation codes and the epilogue tail shared by every convolution kernel (conv.hip, wino.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cctail {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_SIGMOID = 3 };

__device__ __forceinline__ float apply_act(float v, int act, float a, float b) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LRELU) return v > 0.f ? v : (b != 0.f ? b : 0.2f) * v;       // slope: act_b, 0 -> the 0.2 of Back2Future
    if (act == ACT_SIGMOID) return a * (1.f / (1.f + expf(-v))) + b;
    return v;
}

// d act(pre) / d pre expressed through the activation's OUTPUT v, times the upstream gradient g
__device__ __forceinline__ float act_grad(float g, float v, int act, float act_a, float act_b) {
    if (act == ACT_RELU) return v > 0.f ? g : 0.f;
    if (act == ACT_LRELU) return v > 0.f ? g : (act_b != 0.f ? act_b : 0.2f) * g;
    const float sg = (v - act_b) / act_a;
    return g * act_a * sg * (1.f - sg);
}

// epilogue tail shared by every conv kernel: res_mul == 0: act(v + res);  res_mul == 1 (data-gradient calls): the gradient
// w.r.t. the PRE-activation of the layer that produced this conv's input, v * act'(r), r = that layer's output (= this
// conv's input, same shape as the gradient) -- the producer's separate activation-backward pass disappears
// (round 3) ... and with `add`: (v + add) * act'(r) -- the other gradient contributions of a fan-out tensor (a residual
// shortcut, a skip connection, what earlier data-gradients left in the same buffer: add may alias the output) are summed
// here instead of by separate accumulation launches
__device__ __forceinline__ float conv_tail(float v, bool has_res, float r, int res_mul, int act, float a, float b, float addv = 0.f) {
    if (has_res && res_mul) return act_grad(v + addv, r, act, a, b);
    if (has_res) v += r;
    return apply_act(v, act, a, b);
}

}  // namespace cctail
