// convs.hip -- 3x3 stride-1 convolutions of the LOW-RESOLUTION stages (reference call sites: lib/quant_ops.py:39-41 through
// model_blocks.py:74-89, :196-220: the 9x16 -> 45x80 -> 90x160 blocks of NeRV-boost, 15 / 30 channels).
//
// At 45x80 a layer is 58 MFLOP: ~1 us of matrix work.  The 8x32-pixel tiles of the persistent kernels (conv.hip, convbf.hip) cut such an
// image into 18 tiles -- 36 blocks on 256 CUs, each a serial chain of [weight staging, tile staging, 270-deep K loop over 256 pixels,
// epilogue] -- and the launch takes 13..19 us whatever the arithmetic.  Here the same implicit GEMM (v_mfma_f32_16x16x4_f32, exact f32)
// runs on 4x16-pixel tiles x 16 output channels, ONE tile per block and no persistence:
//   * 4 x more blocks per image and a K loop 4 x shorter per wave (each wave owns ONE 16-pixel row = one M tile);
//   * the weights of the block's 16 output channels are one contiguous slice of the OIHW tensor (forward) or 16-row segments of it
//     (data gradient): loaded coalesced into LDS as they are, the B fragment of (tap, channel quad) is one ds_read_b32 at a
//     per-lane base + immediate -- no gather, no fragment re-layout;
//   * the haloed input tile (6 x 24 floats per channel, all channels at once: Cin <= 32) enters through raw buffer loads with the
//     affine prologue applied on the way; zero padding through out-of-range offsets;
//   * epilogues straight from the accumulators (a D fragment = 4 consecutive pixels of one output channel per lane): bias, sin / cos
//     (stride-1 or PixelShuffle(2 / 3 / 5) scatter), gelu pair, residual, plain, dGELU(saved) and dSIN with their per-channel sums.
// Scope: k = 3, Cin <= 32 (64 for the unshuffle(2) prologue: the data gradient of a PixelShuffle(2) up-conv), any Cout (16 per block),
// H * W <= 16384 (65536 for the up-convs and their data gradients), float4-aligned rows.
#include "convs_body.h"

namespace {
using namespace bnerv_conv;
using namespace bnerv_convs;

// NQ = ceil(Cin / 4) rounded to 4 or 8 (16 or 32 staged channels; 16 = 64 channels for the unshuffle prologue)
template <int IN, int EP, int NQ>
__global__ __launch_bounds__(256, 4) void conv_small_kernel(const SArgs sa) {
    conv_small_body<IN, EP, NQ>(sa, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// the same body with 96 staged channels (NQ = 24: 112 KB of LDS, ~150 registers -- one block per CU, which is all a <= 32 x 32 image fills anyway)
template <int IN, int EP>
__global__ __launch_bounds__(256, 1) void conv_small96_kernel(const SArgs sa) {
    conv_small_body<IN, EP, 24>(sa, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}
template <int IN, int EP>
int launch_small96(hipStream_t st, const SArgs& sa) {
    const bnerv_conv_desc& d = sa.d;
    const size_t lds = convs_lds_bytes<24>();
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small96_kernel<IN, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((conv_small96_kernel<IN, EP>), dim3(sa.tiles_x * sa.tiles_y, cdiv(d.Cout, 16), d.B), dim3(256), lds, st, sa);
    BNERV_LAUNCH_CHECK("conv_small96");
    return BNERV_OK;
}
// (instantiated for the TAT convs and their data gradients only: HNeRV-boost's decoder[0], 95 -> 95 at 9 x 16, model_hnerv.py:200-202)
template <int IN, int EP>
constexpr bool small96_combo() {
    return (IN == BNERV_IN_AFFINE && (EP == BNERV_EP_BIAS_GELU || EP == BNERV_EP_BIAS_RES)) ||
           (IN == BNERV_IN_PLAIN && (EP == BNERV_EP_DGELU_SAVED || EP == BNERV_EP_DSIN));
}
static bool small96_shape(const bnerv_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("BNERV_SMALL96"); return e && e[0] == '0'; }();     // A/B switch
    if (off || d.in_mode == BNERV_IN_UNSHUFFLE || d.Cin <= 32 || d.Cin > 96 || d.out_s != 1 || d.in_s != 1 || (size_t)d.H * d.W > 1024) return false;
    const int i = d.in_mode, e = d.ep_mode;
    return (i == BNERV_IN_AFFINE && (e == BNERV_EP_BIAS_GELU || e == BNERV_EP_BIAS_RES)) || (i == BNERV_IN_PLAIN && (e == BNERV_EP_DGELU_SAVED || e == BNERV_EP_DSIN));
}

template <int IN, int EP, int NQ>
int launch_small(hipStream_t st, const SArgs& sa) {
    const bnerv_conv_desc& d = sa.d;
    const size_t lds = convs_lds_bytes<NQ>();
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_kernel<IN, EP, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((conv_small_kernel<IN, EP, NQ>), dim3(sa.tiles_x * sa.tiles_y, cdiv(d.Cout, 16), d.B), dim3(256), lds, st, sa);
    BNERV_LAUNCH_CHECK("conv_small");
    return BNERV_OK;
}

template <int IN, int EP>
int launch_small_nq(hipStream_t st, const SArgs& sa) {
    if constexpr (IN == BNERV_IN_UNSHUFFLE) return sa.d.Cin <= 32 ? launch_small<IN, EP, 8>(st, sa) : launch_small<IN, EP, 16>(st, sa);
    else {
        if constexpr (small96_combo<IN, EP>()) { if (sa.d.Cin > 32) return launch_small96<IN, EP>(st, sa); }
        return sa.d.Cin <= 16 ? launch_small<IN, EP, 4>(st, sa) : launch_small<IN, EP, 8>(st, sa);
    }
}

}  // namespace

// shapes of this family: small images whose 8x32 tiling leaves the chip idle (the threshold keeps 180x320 and above on the persistent kernels)
bool bnerv_convs_shape_ok(const bnerv_conv_desc& d, int vec) {
    { const char* e = getenv("BNERV_SMALL"); if (e && e[0] == '0') return false; }      // A/B switch, read per call (tests reach the other families with it)
    // small images; an up-conv (several cout groups per tile) pays up to 180x320, where the persistent split kernel still runs one tile per block
    const bool uns = d.in_mode == BNERV_IN_UNSHUFFLE;          // the data gradient of a PixelShuffle(2) up-conv: its input is the shuffled gradient
    static const size_t px_up = [] { const char* e = getenv("BNERV_SMALL_MAXPX_UP"); return e ? (size_t)atol(e) : (size_t)65536; }();     // (A/B switches)
    static const size_t px_uns = [] { const char* e = getenv("BNERV_SMALL_MAXPX_UNS"); return e ? (size_t)atol(e) : (size_t)65536; }();
    const size_t max_px = uns ? px_uns : (d.out_s == 2 && d.Cout >= 32) ? px_up : 16384;
    // (33..96 input channels on an image of <= 1024 pixels: the 96-channel staging of conv_small96_kernel -- round 5 ran HNeRV-boost's
    //  95 -> 95 TAT convs at 9 x 16 on the generic kernel's 12 blocks: 47-53 us per launch for 0.2 GFLOP)
    const int cin_max = uns ? 64 : (small96_shape(d) ? 96 : 32);
    if (!(vec && d.k == 3 && d.Cin <= cin_max && (size_t)d.H * d.W <= max_px && d.B <= 65535 && cdiv(d.Cout, 16) <= 65535)) return false;
    if (uns) return d.in_s == 2 && (d.Cin & 3) == 0 && d.ep_mode == BNERV_EP_PLAIN && d.out_s == 1 && (size_t)d.B * d.Cin * d.H * d.W * 4 < LEAN_MAX_BYTES;
    if (d.in_s != 1) return false;
    if (d.Cin <= 12 && d.Cout <= 12) return false;                 // the 12-channel layers have their own family (conv4.hip)
    if (!(d.in_mode == BNERV_IN_PLAIN || d.in_mode == BNERV_IN_AFFINE)) return false;
    const int e = d.ep_mode;
    if (d.out_s != 1) return d.in_mode == BNERV_IN_PLAIN && (e == BNERV_EP_BIAS || e == BNERV_EP_BIAS_SIN) && (d.out_s == 2 || d.out_s == 3 || d.out_s == 5);
    if (d.in_mode == BNERV_IN_AFFINE) return e == BNERV_EP_BIAS || e == BNERV_EP_BIAS_GELU || e == BNERV_EP_BIAS_RES;
    return e == BNERV_EP_BIAS || e == BNERV_EP_BIAS_SIN || e == BNERV_EP_PLAIN || e == BNERV_EP_DGELU_SAVED || e == BNERV_EP_DSIN;
}
int bnerv_convs_tiles(int H, int W) { return cdiv(H, STH) * cdiv(W, STW); }

// 1: not this family's layer; BNERV_OK / negative BNERV_E_*: handled
int bnerv_convs_try(hipStream_t st, const bnerv_conv_desc& d, int vec, int ksplit) {
    if (ksplit > 1 || !bnerv_convs_shape_ok(d, vec)) return 1;
    SArgs sa;
    sa.d = d;
    sa.tiles_x = cdiv(d.W, STW);
    sa.tiles_y = cdiv(d.H, STH);
    const int in = d.in_mode, ep = d.ep_mode;
#define BNERV_CASE(I, E) if (in == I && ep == E) return launch_small_nq<I, E>(st, sa);
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DGELU_SAVED)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DSIN)
    BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS_GELU)
    BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS_RES)
    BNERV_CASE(BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN)
#undef BNERV_CASE
    return 1;
}
