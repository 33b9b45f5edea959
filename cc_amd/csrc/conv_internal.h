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

// Parking (cc_conv2d_wgrad_list): while on (per thread), wgrad_thin_launch stores its problem instead of launching it (the reduce
// descriptor is emitted as usual) and wgrad_thin_flush launches what is stored -- problems of one kernel instance share launches
// (k_wgrad_thin_multi).  wgrad_thin_park returns the previous state.
bool wgrad_thin_park(bool on);
bool wgrad_thin_parking();
double wgrad_thin_parked_gflop();
void wgrad_thin_flush(hipStream_t s);

// name of the kernel wgrad_thin_launch would use ("" when not eligible)
void wgrad_thin_name(int B, int M, int AH, int AW, int Cin, int IH, int IW, int R, int S, int si, int pad, char* out, int cap);


// ---- Winograd F(2x2, 3x3) convolution (wino.hip): the 3x3 / stride-1 / pad-1 layers (forward and data-gradient arithmetic)
struct WinoGeom {
    int B, Cin, H, W; long x_bs;                 // input [B, Cin, H, W]; output [B, M, H, W]
    int M; long y_bs, res_bs, add_bs;
    int act; float act_a, act_b; int res_mul;    // epilogue (conv_tail.h)
};
struct WinoProb { const float* x; const float* U; const float* bias; const float* res; const float* add; float* y; float* part; };
struct WinoPlan {
    int ok;                     // geometry runs on the Winograd kernel
    int Mpad, Cpad, nchunk;     // M padded to 64, Cin padded to 8, 8-channel chunks
    int TY, TX, nqb, nmb;       // 2x2 output tiles per image (rows, columns), 64-tile blocks over all images, 64-row blocks
    int nsplit, cps;            // split-K over channel chunks (deterministic second pass: conv.hip k_splitk_epilogue*)
    int Hp, Wp;                 // partial slab rows / pitch: [split][n][m][Hp][Wp]
    size_t u_floats, part_floats;
    int tile;                   // kernel instance: 0 = 64 rows x 64 tiles (nqb / nmb count those blocks); 1 = 32 x 32, four waves, two
                                // workgroups per CU; 2 = 32 x 32, eight waves: the channel chunks halved INSIDE the workgroup
};
constexpr int WINO_MAXP = 12;   // problems per launch (= conv.hip MAXCLS)
// eligibility (geometry only: the weight image is laid out for the algorithm the plan names) + split-K for `mult` problems per launch
WinoPlan wino_plan(int B, int Cin, int H, int W, int M, int mult);
// nprob same-shaped problems in one launch; p[k].part: partial slabs of problem k when plan.nsplit > 1 (the caller runs the
// split-K epilogue).  -> false (nothing launched) when x is not 16-byte aligned / too large for 32-bit byte offsets
bool wino_launch(const WinoGeom& g, const WinoPlan& p, const WinoProb* probs, int nprob, hipStream_t s);
// U = G g G^T of one layer into `U` (plan.u_floats floats); strides / flip as in wino_weights.h
void wino_weights_launch(const float* w, float* U, int M, int Cin, int Cpad, int Mpad, long w_sm, long w_sc, long w0, long w_ri,
                         long w_sj, int flip, hipStream_t s);

// ---- Winograd F(3x3, 2x2) weight gradient (wino_wgrad.hip): 3x3 / stride-1 / pad-1 layers, dY [B, M, H, W], x [B, Cin, H, W]
struct WinoWgradPlan {
    int ok;
    int TX, CPR, CPI, NCH;      // tile columns; 8-tile chunks per tile row / per image / in total
    int nmb, ncb, Cp;           // 64-row blocks of dY / of the input, padded input channels (slab pitch)
    int nsplit, cps;            // split of the chunk range, chunks per split
    size_t ws_floats;           // 64 + partial slabs ws[split][tap 9][m][Cp] (the layout of k_wgrad3x3: reduce kind 1)
};
WinoWgradPlan wino_wgrad_plan(int B, int M, int H, int W, int Cin, int G, int minq = -1);      // minq: tile-count floor (-1: default)
WinoWgradPlan wino_wgrad_plan_parked(const WinoWgradPlan& alone, int M);      // split count for a problem that shares a multi-geometry launch
// G (<= 4) same-shaped problems in one launch; ws[k]: slab area of problem k.  -> false (nothing launched) when the tensors are not
// 8- / 16-byte aligned
// park != nullptr (and room left): the G problems are appended to the collector instead of launched; wino_wgrad_launch_parked then
// runs everything collected in multi-geometry launches (problems of different shapes share a launch, longest chains first)
constexpr int WINO_WGRAD_PARK_CAP = 96;
struct WinoWgradParked {
    struct Desc {
        const float* a; const float* x; float* ws;
        int M, C, H, W; long a_bs, x_bs; unsigned a_bytes, x_bytes;
        int TX, CPR, CPI, NCH, cps, ncb, Cp, nxy, nsplit;
        double gflop;
    };
    Desc d[WINO_WGRAD_PARK_CAP];
    int n;
};
bool wino_wgrad_launch(const WinoWgradPlan& p, const float* const* a, const float* const* x, float* const* ws, int G, int B, int M,
                       int H, int W, long a_bs, int Cin, long x_bs, hipStream_t s, WinoWgradParked* park = nullptr);
void wino_wgrad_launch_parked(WinoWgradParked* c, hipStream_t s);

// ---- 3x3 / stride-1 / pad-1 layers with <= 4 channels on one side (conv_heads.hip): the prediction heads and their data-gradients
// as HBM-bound vector-ALU kernels instead of padded MFMA tiles.  Input [B, Cin, H, W] (batch stride x_bs), output [B, M, H, W];
// weight element (m, c, i, j) at w[w0 + m * w_sm + c * w_sc + 3 i + j]; dstep +1: forward taps (input pixel p + (i - 1, j - 1)),
// -1: data-gradient taps (p - (i - 1, j - 1)); epilogue operands as in conv_tail.h (res / add: tensors of the output's shape).
struct HeadConv {
    const float* x; const float* w; const float* bias; const float* res; const float* add; float* y;
    int B, Cin, H, W, M;
    long x_bs, y_bs, res_bs, add_bs;
    long w_sm, w_sc, w0;
    int dstep;
    int act; float act_a, act_b; int res_mul;
};
// few REDUCTION channels (Cin <= 4): vector width the problem runs with (4 / 2 / 1 pixels per work-item), 0 = not eligible
int head_conv_thinc_vec(const HeadConv& g);
bool head_conv_thinc_launch(const HeadConv& g, hipStream_t s);      // -> false: not eligible, nothing launched
// few OUTPUT channels (M <= 4) from <= 64 input channels on large maps (the heads' forward pass)
bool head_conv_thinm_ok(const HeadConv& g);
bool head_conv_thinm_launch(const HeadConv& g, hipStream_t s);      // -> false: not eligible, nothing launched
// weight gradient of a head (M <= 4 channels of dY [B, M, H, W], x [B, Cin, H, W], 3x3 / stride 1 / pad 1): partial slabs
// ws[nblk][M][Cin * 9] (the generic kernel's layout: wgrad_reduce.hip kind 0 with nsplit = nblk)
struct HeadWgradPlan { int ok, R, nstrips, nblk; size_t ws_floats; };
HeadWgradPlan head_wgrad_plan(int B, int M, int H, int W, int Cin);
bool head_wgrad_launch(const HeadWgradPlan& p, const float* dy, const float* x, float* ws, int B, int M, int H, int W, long dy_bs,
                       int Cin, long x_bs, hipStream_t s);      // -> false: tensors not 16-byte aligned, nothing launched

}  // namespace ccint
