// conv4.hip -- the 3x3 stride-1 convolutions with at most 12 input and 12 output channels (every TAT conv, stride-1 block conv and
// their data gradients of the 12-channel stages; reference call sites: model_blocks.py:74-89, :196-220 through
// lib/quant_ops.py:39-41) on v_mfma_f32_4x4x1_16b_f32.
//
// Why another MFMA shape.  On gfx950 the f32 MFMA issues at the f32 vector rate whatever its shape, so the 16x16x4 kernels of
// conv.hip pay for N = 16 output channels on a 12-channel layer (25 % of the matrix cycles are padding) and move an A AND a B
// fragment through LDS for every 4 k-steps.  The 4x4x1 shape is 16 independent 4x4 outer products per instruction:
//     D_b[i][j] += A_b[i] * B_b[j]            b = 0..15;  A_b[i] in lane 4b + i,  B_b[j] in lane 4b + j,  D_b[i][j]: lane 4b + j, reg i
// With A = the input value of the lane's OWN pixel (a wave = 64 pixels = 2 tile rows x 32 px, block b = 4 consecutive pixels)
// and B = a quad of weights (4 output channels of one (ci, tap)), one instruction is 64 px x 4 cout x 1 k:
//   * N granularity 4: 12 output channels are exactly 3 instructions per (ci, tap) -- 324 MFMAs x 8 cycles per 64 pixels against
//     108 x 32 for the 16x16x4 form;
//   * the weights never move: BLGP = 4 + g broadcasts the 16-lane group g of B to all four groups, so one register holds four
//     quads (each replicated over its group's four blocks) and the 27 quads of an input channel are 7 registers; all 12 channels
//     (84 registers) stay resident for the whole persistent block;
//   * the K loop is  { 1 ds_read_b32 ; 3 v_mfma }  x 108 with compile-time LDS offsets: no VALU, no SALU, no address arithmetic;
//   * D: lane 4b + j holds 4 consecutive pixels of output channel 4n + j = one 16-byte store per accumulator, and every lane
//     carries a real channel (the 16x16x4 epilogue spends a quarter of its lanes on the padding channels).
// Tile geometry, staging (raw buffer loads prefetched under the matrix phase, prologue applied on the way into LDS), item
// partition, epilogues and the hosted slab reductions are those of conv_lean_kernel (conv.hip).
#include "conv4_body.h"

namespace {
using namespace bnerv_conv;
using namespace bnerv_q4;

template <int IN, int EP>
__global__ __launch_bounds__(256, 3) void conv_q4_kernel(const KArgs ka, const SidePack side) {
    conv_q4_body<IN, EP>(ka, side, (int)blockIdx.x, (int)gridDim.x);
}

template <int IN, int EP>
int launch_q4(hipStream_t st, KArgs& ka) {
    q4_prepare(ka);
    const size_t lds = q4_lds_bytes();
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_q4_kernel<IN, EP>), 256, lds) != hipSuccess || nb < 1) nb = 1;
        blocks_per_cu = nb;
    }
    int grid = 256 * blocks_per_cu;                        // everything resident: the static item partition is then balanced
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(ka.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_q4_kernel<IN, EP>), dim3(grid), dim3(256), lds, st, ka, side);
    BNERV_LAUNCH_CHECK("conv_q4");
    return BNERV_OK;
}

}  // namespace

#ifdef BNERV_TRACE
extern "C" int bnerv_debug_trace4_read(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace4), sizeof(g_trace4)); }
#endif
// 1: not this family's layer (the caller goes on to the 16x16x4 kernels); BNERV_OK / negative BNERV_E_*: handled.
int bnerv_conv4_try(hipStream_t st, bnerv_conv::KArgs& ka) {
    const bnerv_conv_desc& d = ka.d;
    static const bool off = [] { const char* e = getenv("BNERV_Q4"); return e && e[0] == '0'; }();      // A/B switch (tools/kbench.py)
    if (off) return 1;
    if (!q4_shape_ok(ka)) return 1;
    const int in = d.in_mode, ep = d.ep_mode;
#define BNERV_CASE(I, E) if (in == I && ep == E) return launch_q4<I, E>(st, ka);
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DGELU_SAVED)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DSIN)
#undef BNERV_CASE
    return 1;
}
