// convbf.hip -- split-bf16 implicit-GEMM convolution for the high-resolution 12..16-channel layers (gfx950).
//
// Forward convolution and data gradient of the stride-1 3x3 / 1x1 convs with 8 < Cin <= 16 and Cout <= 16 (every TAT conv, the
// stride-1 block convs, heads and their data gradients of the NeRV-boost decoders at >= 180x320; reference call sites as in
// conv.hip: lib/quant_ops.py:39-41 via model_blocks.py:74-89, :196-220).  Same GEMM view, work-item walk, buffer-load staging and
// accumulator-direct epilogues as conv.hip's lean kernel; the contraction runs on the bf16 matrix pipe with f32 operands
// split into bf16 pieces (see below), f32 accumulation.
#include "common.h"
#include "sidejob.h"
#include "split16.h"
#include "wgrad_bfw_body.h"   // the wide weight-gradient body: one role of the paired launch at the bottom of this file
#include <stdlib.h>
#include <string.h>
#include <type_traits>

constexpr int BF_NOT_HANDLED = 1;     // "not this family's layer" (bnerv_convbf_try and the launchers below); errors stay negative

namespace {

constexpr int TH = 8, TW = 32;     // spatial tile (as conv.hip)

template <int KS> struct Geo {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int ROWS = TH + 2 * PAD;
    static constexpr int XOFF = (KS == 3) ? 4 : 0;             // left margin, multiple of 4 -> aligned float4 segments
    static constexpr int RS = TW + 2 * XOFF;                   // 40 / 32
    static constexpr int SEGS = RS / 4;
    static constexpr int T = KS * KS;
    static constexpr int COL0 = XOFF - PAD;                    // planar column of input x = x0 + px + kx - PAD is px + kx + COL0
};

struct KArgs {
    bnerv_conv_desc d;
    int tiles_x, tiles_y, total_items;
    int ksplit, cps;                       // wide kernel, EP_PLAIN: the K chunks are split over `ksplit` work items of `cps` chunks, each writing its slab of d.partial
    unsigned magic_tiles, magic_tiles_x;   // floor(2^32 / n) + 1: a / n == umulhi(a, magic) for a * n < 2^32
};

static inline unsigned div_magic(int n) { return n <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)n) + 1ull); }   // 0 encodes n == 1
__device__ __forceinline__ int fast_div(int a, unsigned magic) { return magic ? (int)__umulhi((unsigned)a, magic) : a; }

template <int IN>
__device__ __forceinline__ float xform1(float v, float sc, float sh, float aux) {
    if constexpr (IN == BNERV_IN_AFFINE) return v * sc + sh;
    if constexpr (IN == BNERV_IN_GELU_AFFINE) return gelu_f(v) * sc + sh;
    return v;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned shift_bytes, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) - shift_bytes), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff, (int)soff, 0);
    // store-data hazard of 128-bit buffer stores with an SGPR soffset (see conv.hip, bstore): pad, fenced against the scheduler
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3");
    __builtin_amdgcn_sched_barrier(0);
}

struct LItem { int b, ty, tx; };

#ifdef BNERV_TRACE_BFW   // debug variant only (tools/ktrace_bfw.py): s_memtime stamps of the wide kernel's stages
__device__ unsigned long long g_trace_bfw[256 * 4 * 8 * 8];
#define WTR(slot) do { if (lane == 0 && vb < 256 && trace_stage < 8) g_trace_bfw[((vb * 4 + wave) * 8 + trace_stage) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WTR(slot) do {} while (0)
#endif
#ifdef BNERV_TRACE   // debug variant only (tools/ktrace_bf.py): s_memtime phase stamps
__device__ unsigned long long g_trace_bf[1024 * 4 * 6 * 8];
#define TRACE(slot) do { if (lane == 0 && blockIdx.x < 1024 && trace_iter < 6) g_trace_bf[((blockIdx.x * 4 + wave) * 6 + trace_iter) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRACE(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------- split kernels
// Measured on gfx950 (tools/ubench/mfma_interleave.cpp, tools/ubench/bf16_split.cpp, profiles/r02_*): the f32-input MFMA runs at
// the f32 VECTOR rate and ordinary VALU work does not hide under it (interleaved in one wave or phased across waves), so a
// 12-channel layer on v_mfma_f32_16x16x4_f32 cannot beat  MFMA cycles + VALU cycles  -- 23.5 us of matrix work at 720x1280
// (25 % of it the 12 -> 16 padding) plus everything around it.  The 16-bit matrix pipe is 16x faster, and an f32 product can
// be rebuilt from 16-bit pieces with f32 accumulation (every partial product of two pieces is exact in f32):
//   SP_BF16X6   x = x1 + x2 + x3 exactly (bf16 pieces of 8 mantissa bits, RNE at every level);  a b ~= a1 b1 + (a1 b2 + a2 b1) +
//               (a2 b2 + a1 b3 + a3 b1): 6 MFMAs, dropped terms <= 2^-24 |a b|.  Measured 7.9e-8 * sum|a b| at K = 128 -- better than
//               the k-ordered f32 chain of the f32 MFMA itself (1.3e-7).
//   SP_F16X3    x s = h + l + O(2^-22) (f16 pieces of 11 bits after an exact power-of-two scale s that puts the TILE's largest
//               |x| into [2^14, 2^15): no overflow, and whatever falls below the f16 range is < 2^-39 of that maximum);
//               a b ~= a_h b_h + (a_h b_l + a_l b_h): 3 MFMAs, error ~2^-22 |a b|; the accumulators are rescaled exactly.
//   SP_BF16X3   two bf16 pieces, 3 MFMAs, ~2^-16 |a b| (opt-in, for comparison only).
// 6 x 16 (or 3 x 16) cycles replace 8 x 32 cycles per 32 k-values.
//
// Same scope, item walk, buffer-load staging and accumulator-direct epilogues as conv.hip's lean kernel; what changes:
//   * v_mfma_f32_16x16x32_{bf16,f16}: lane (i = l & 15, kq = l >> 4) holds 8 consecutive k.  k is ordered (tap, channel): K step s
//     covers taps 2s and 2s+1 x 16 channels, lane group kq -> tap 2s + (kq >> 1), channels 8 (kq & 1) .. +7.  A lane's A fragment
//     is 8 channels of ONE pixel = 16 bytes of a PIXEL-MAJOR tile  s_a[piece][row][pixel][half]  (16-B slots; the slot index
//     inside each group of 4 pixels is rotated by the group index, which spreads both the staging writes of 8 neighbouring
//     segments and the 16 pixels of a fragment read over the banks).
//   * staging: thread slot = (channel half, halo row, 4-px segment); waves 0-1 own channels 0..7, waves 2-3 channels 8..8+CB-1
//     (CB = 4 for Cin <= 12: the padding channels are never loaded or converted).  One buffer load per channel in the coalescing
//     pattern of the planar kernels, prologue transform, split, and 4 pixels x NS ds_write_b128.
//   * B fragments live in LDS in lane order (one ds_read_b128 per (step, piece), shared by the wave's 4 M tiles), built once per
//     block from a coalesced read of the weight tensor.  Channels >= Cin and tap 9 carry zero weights, so whatever finite value
//     the A side holds there contributes nothing.
// exact power of two that moves max_abs into [2^14, 2^15); 1 for 0 / non-finite input
__device__ __forceinline__ float pow2_scale(float max_abs) {
    const unsigned e = (__builtin_bit_cast(unsigned, max_abs) >> 23) & 0xffu;
    return (e == 0u || e == 0xffu) ? 1.0f : __builtin_bit_cast(float, (268u - e) << 23);
}
__device__ __forceinline__ float pow2_inv(float s) {                      // 1 / s for s = 2^k, exact
    return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, s));
}

template <int KS> struct BfGeo {
    using G = Geo<KS>;
    static constexpr int NP = (KS == 3) ? 34 : 32;                 // pixels per tile row kept in LDS (planar columns COL0 .. COL0 + NP - 1)
    // 16-B slots per row: 68 / 64.  (3x3: 8 full groups of 4 pixels + pixels 32, 33, whose group index 8 rotates by 0, so they use the
    // first four slots of the ninth group only -- the 4 spare slots are what lets the wide kernel keep two blocks per CU)
    static constexpr int SPR = (NP / 4) * 8 + (NP % 4) * 2;
    static constexpr int PIECE = G::ROWS * SPR * 16;               // bytes per piece: 10880 / 8192
    static constexpr int STEPS = (G::T + 1) / 2;                   // 5 / 1
    static constexpr int HSLOT = G::ROWS * G::SEGS;                // staging slots per channel half: 100 / 64
    static_assert(HSLOT <= 128, "one slot per thread of a wave pair");
};
__device__ __forceinline__ int bf_slot(int r, int p, int h, int spr) { return r * spr + 8 * (p >> 2) + ((2 * (p & 3) + h + (p >> 2)) & 7); }

template <int KS, int IN, int EP, int SP, int CB>
__global__ __launch_bounds__(256, 3) void conv_bf_kernel(const KArgs ka, const SidePack side) {
    using G = Geo<KS>;
    using BG = BfGeo<KS>;
    constexpr int NS = Split<SP>::NS;
    constexpr bool SCALED = Split<SP>::SCALED;
    constexpr bool AFF = (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE);
    constexpr bool RED = (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    constexpr int SB_BYTES = BG::STEPS * NS * 64 * 16;
    static_assert(IN != BNERV_IN_TANHGRAD && IN != BNERV_IN_UNSHUFFLE, "prologues of the split kernel");
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* s_a = reinterpret_cast<char*>(smem);                      // [NS][ROWS][SPR] 16-B slots
    char* s_b = s_a + NS * BG::PIECE;                              // [STEPS][NS][64 lanes] 16 B
    float* s_red = reinterpret_cast<float*>(s_b + SB_BYTES);       // [4 waves][2][16]
    float* s_aff = s_red + 128;                                    // [2][16]
    float* s_max = s_aff + 32;                                     // [4] wave maxima (tile / weight scale)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int nlb = (gridDim.x - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lb;
    if (itx >= r1) { side_run_hosted(side, smem); return; }
    const int step_q = fast_div(nlb, ka.magic_tiles_x), step_r = nlb - step_q * tiles_x;
    LItem it;
    {
        const int tiles = tiles_x * tiles_y;
        it.b = fast_div(itx, ka.magic_tiles);
        const int t = itx - it.b * tiles;
        it.ty = fast_div(t, ka.magic_tiles_x);
        it.tx = t - it.ty * tiles_x;
    }
    auto advance = [&](LItem a) __attribute__((always_inline)) {
        a.tx += step_r;
        a.ty += step_q;
        if (a.tx >= tiles_x) { a.tx -= tiles_x; ++a.ty; }
        while (a.ty >= tiles_y) { a.ty -= tiles_y; ++a.b; }
        return a;
    };

    // ---- staging slot of this thread: waves 0-1 <-> channels 0..7, waves 2-3 <-> channels 8..8+CB-1; (halo row, 4-px segment) with
    //      the segment fastest (the coalescing pattern of the planar kernels)
    const int s_h = wave >> 1;                                     // wave-uniform
    const int sidx = tid & 127;
    const bool has_slot = sidx < BG::HSLOT;
    const int s_sg = sidx % G::SEGS, s_r = (sidx / G::SEGS) % G::ROWS;
    const unsigned voff0 = has_slot ? (unsigned)((((8 * s_h) * H + s_r) * W + 4 * s_sg) * 4) : OOB;
    const unsigned hw4 = (unsigned)(H * W * 4);
    const unsigned shift = (unsigned)((G::PAD * W + G::XOFF) * 4);
    const unsigned in_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, in_bytes);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ro2 = make_rsrc(((EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) && d.out2) ? d.out2 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(d.aux0 ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);
    // LDS byte address of pixel j of this thread's segment (piece 0); pixels outside the kept columns are not written
    int w_addr[4];
    bool w_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 4 * s_sg + j - G::COL0;
        w_ok[j] = has_slot && p >= 0 && p < BG::NP;
        w_addr[j] = bf_slot(s_r, w_ok[j] ? p : 0, s_h, BG::SPR) * 16;
    }

    const unsigned ovoff = li < Cout ? (unsigned)(((li * H) * W + 4 * kq) * 4) : OOB;
    const float bias_l = (EP != BNERV_EP_PLAIN && !RED && d.bias && li < Cout) ? d.bias[li] : 0.f;

    auto fetch_affine = [&](int b) __attribute__((always_inline)) {
        const int c = tid & 15;
        float v = 0.f;
        if (tid < 32 && c < Cin) v = tid < 16 ? 1.0f + d.scale[b * Cin + c] : d.shift[b * Cin + c];
        return v;
    };
    auto block_max = [&](float v) __attribute__((always_inline)) {                        // max over the block (a barrier on each side of the LDS exchange)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        if (lane == 0) s_max[wave] = v;
        lds_barrier();
        const float m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        lds_barrier();
        return m;
    };
    float scl = 0.f;

    f32x4 ra[8];
    auto issue = [&](const LItem& a) __attribute__((always_inline)) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        unsigned sb = (unsigned)((((a.b * Cin) * H + ty0) * W + tx0) * 4);
        const int gy = ty0 + s_r - G::PAD, gx = tx0 + 4 * s_sg - G::XOFF;
        const unsigned vo = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? voff0 : OOB;
        if (s_h == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ra[e] = bload(rx, vo, sb); sb += hw4; }
        } else {
#pragma unroll
            for (int e = 0; e < CB; ++e) { ra[e] = bload(rx, vo, sb); sb += hw4; }
        }
    };
    // prologue transform in place (zero padding applies AFTER it); returns the largest |value| of this thread
    auto transform_half = [&](const LItem& a, auto ne_tag) __attribute__((always_inline)) {
        constexpr int NE = decltype(ne_tag)::value;
        float mx = 0.f;
        if constexpr (IN != BNERV_IN_PLAIN) {
            const int ty0 = a.ty * TH, tx0 = a.tx * TW;
            const int gy = ty0 + s_r - G::PAD, gx = tx0 + 4 * s_sg - G::XOFF;
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float sc = (AFF && inside) ? s_aff[8 * s_h + e] : 0.f, sh = (AFF && inside) ? s_aff[16 + 8 * s_h + e] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[e][j] = xform1<IN>(ra[e][j], sc, sh, 0.f);
            }
        }
        if constexpr (SCALED) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(ra[e][j]));
        }
        return mx;
    };
    auto write_half = [&](float s, auto ne_tag) __attribute__((always_inline)) {
        constexpr int NE = decltype(ne_tag)::value;
        float xs[4][8];                                    // (element-wise copies: a vector subscript that survives to codegen goes to scratch)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x4 t = e < NE ? ra[e] : f32x4{0.f, 0.f, 0.f, 0.f};
            xs[0][e] = SCALED ? t.x * s : t.x;
            xs[1][e] = SCALED ? t.y * s : t.y;
            xs[2][e] = SCALED ? t.z * s : t.z;
            xs[3][e] = SCALED ? t.w * s : t.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 pc[NS];
            split8<SP, NE>(xs[j], pc);
            if (w_ok[j]) {
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<u32x4*>(s_a + p * BG::PIECE + w_addr[j]) = pc[p];
            }
        }
    };
    // Staging of the tile whose loads are in `ra`, in two halves around a barrier the loop has anyway:
    //   stage_pre   registers only: prologue transform, and (scaled split) this wave's largest |value| -> s_max[wave]
    //   stage_post  after the barrier: block maximum -> power-of-two scale, split, LDS writes; returns 1 / scale
    auto stage_pre = [&](const LItem& a) __attribute__((always_inline)) {
        float mx;
        if (s_h == 0) mx = transform_half(a, std::integral_constant<int, 8>{});
        else mx = transform_half(a, std::integral_constant<int, CB>{});
        if constexpr (SCALED) {
            mx = has_slot ? mx : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            if (lane == 0) s_max[wave] = mx;
        }
    };
    auto stage_post = [&]() __attribute__((always_inline)) {
        float s = 1.0f;
        if constexpr (SCALED) s = pow2_scale(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])));
        if (s_h == 0) write_half(s, std::integral_constant<int, 8>{});
        else write_half(s, std::integral_constant<int, CB>{});
        return SCALED ? pow2_inv(s) : 1.0f;
    };
    auto flush_partials = [&](const LItem& a) __attribute__((always_inline)) {
        if (wave == 0 && lane < 32) {
            const int q = lane >> 4, c = lane & 15;
            const float s = ((s_red[(0 * 2 + q) * 16 + c] + s_red[(1 * 2 + q) * 16 + c]) + s_red[(2 * 2 + q) * 16 + c]) + s_red[(3 * 2 + q) * 16 + c];
            const size_t row = (size_t)(a.ty * tiles_x + a.tx) * d.B + a.b;
            if (c < Cout) d.partial[(row * 2 + q) * Cout + c] = s;
        }
    };

    // ---- prologue: first tile's loads in flight, then the B fragments.  The weight tensor is read coalesced (element i of the
    //      OIHW array by thread i mod 256); element (a, b, tap) lands at fragment (step, lane group, column, k) of every piece.
    int aff_b = -1, ep_b = -1;
    float aff_v = 0.f;
    if constexpr (AFF) { aff_v = fetch_affine(it.b); aff_b = it.b; }
    issue(it);
    float inv_b = 1.0f;
    {
        const int nw = d.wCo * d.wCi * G::T;
        constexpr int NWV = (16 * 16 * G::T + 255) / 256;
        float wv[NWV];
        float wmx = 0.f;
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            const int i = tid + k * 256;
            wv[k] = i < nw ? d.w[i] : 0.f;
            wmx = fmaxf(wmx, fabsf(wv[k]));
        }
        for (int i = tid; i < SB_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(s_b)[i] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (AFF) { if (tid < 32) s_aff[tid] = aff_v; }
        float sb_ = 1.0f;
        if constexpr (SCALED) { sb_ = pow2_scale(block_max(wmx)); inv_b = pow2_inv(sb_); }
        else lds_barrier();
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            const int i = tid + k * 256;
            const int pair = i / G::T, t = i - pair * G::T;
            const int wa_ = pair / d.wCi, wb_ = pair - wa_ * d.wCi;
            const int n = d.transposed ? wb_ : wa_, c = d.transposed ? wa_ : wb_;       // GEMM column (cout) and k-channel (cin)
            if (i < nw && n < Cout && c < Cin) {
                const int tg = d.transposed ? G::T - 1 - t : t;
                const int st = tg >> 1, q = ((tg & 1) << 1) | (c >> 3), e = c & 7;
                float r = SCALED ? wv[k] * sb_ : wv[k];
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    unsigned short bits;
                    float back;
                    if constexpr (SP == SP_F16X3) { const _Float16 hv = (_Float16)r; bits = __builtin_bit_cast(unsigned short, hv); back = (float)hv; }
                    else { const __bf16 hv = (__bf16)r; bits = __builtin_bit_cast(unsigned short, hv); back = (float)hv; }
                    *reinterpret_cast<unsigned short*>(s_b + ((st * NS + p) * 64 + q * 16 + n) * 16 + e * 2) = bits;
                    r -= back;
                }
            }
        }
    }
    stage_pre(it);                                         // (the affine table was made visible by the barrier above)
    if constexpr (SCALED) lds_barrier();
    float inv_cur = stage_post() * inv_b;

    // A-fragment byte addresses (tile-invariant): [step][x half of the M tile], rows of the M tile pair added as immediates
    int a_addr[BG::STEPS][2];
#pragma unroll
    for (int s = 0; s < BG::STEPS; ++s) {
        int t = 2 * s + (kq >> 1);
        if (t > G::T - 1) t = G::T - 1;
        const int ky = t / KS, kx = t - ky * KS;
#pragma unroll
        for (int xh = 0; xh < 2; ++xh) a_addr[s][xh] = bf_slot(2 * wave + ky, xh * 16 + li + kx, kq & 1, BG::SPR) * 16;
    }
    const int b_addr = lane * 16;

    LItem prev = it;
    bool have_prev = false;
    int trace_iter = 0; (void)trace_iter;
    for (; itx < r1; itx += nlb, ++trace_iter) {
        TRACE(0);
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool has_next = itx + nlb < r1;
        LItem nxt = it;
        if (has_next) nxt = advance(it);
        lds_barrier();                                     // (A) s_a(t), B fragments and s_red(t-1) visible
        TRACE(1);
        if (has_next) issue(nxt);
        TRACE(2);
        if constexpr (RED) { if (have_prev) flush_partials(prev); }
#pragma unroll
        for (int s = 0; s < BG::STEPS; ++s) {
            u32x4 bfr[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) bfr[p] = *reinterpret_cast<const u32x4*>(s_b + (s * NS + p) * 1024 + b_addr);
            // M tiles in groups of MG: consecutive MFMAs go to DIFFERENT accumulators (a dependent chain issues slower) while the
            // fragments of one group stay within the register budget; smallest terms first
            constexpr int MG = (NS == 3) ? 2 : 4;
#pragma unroll
            for (int m0 = 0; m0 < 4; m0 += MG) {
                u32x4 afr[MG][NS];
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        afr[m][p] = *reinterpret_cast<const u32x4*>(s_a + p * BG::PIECE + a_addr[s][(m0 + m) & 1] + ((m0 + m) >> 1) * (BG::SPR * 16));
#define BNERV_BF_PROD(pa, pb) _Pragma("unroll") for (int m = 0; m < MG; ++m) acc[m0 + m] = mfma16<SP>(afr[m][pa], bfr[pb], acc[m0 + m]);
                if constexpr (NS == 3) {
                    BNERV_BF_PROD(2, 0)
                    BNERV_BF_PROD(0, 2)
                    BNERV_BF_PROD(1, 1)
                }
                BNERV_BF_PROD(1, 0)
                BNERV_BF_PROD(0, 1)
                BNERV_BF_PROD(0, 0)
#undef BNERV_BF_PROD
            }
        }
        TRACE(3);
        if (has_next) {
            if constexpr (AFF) {
                if (nxt.b != aff_b) {                      // sample change (B > 1): reload the affine table (rare, latency exposed)
                    const float v = fetch_affine(nxt.b);
                    lds_barrier();                         // (every wave has read the old table)
                    if (tid < 32) s_aff[tid] = v;
                    lds_barrier();
                    aff_b = nxt.b;
                }
            }
            stage_pre(nxt);                                // registers + s_max only: legal before barrier (B)
        }
        lds_barrier();                                     // (B) every wave is done reading s_a(t); s_max(t+1) visible
        TRACE(4);
        float inv_next = 1.0f;
        if (has_next) inv_next = stage_post() * inv_b;
        TRACE(5);
        if constexpr (SCALED) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] *= inv_cur;
        }
        // ---- epilogue straight from the accumulators (as the lean kernel: the D fragment layout is the same)
        {
            const int ty0 = it.ty * TH, tx0 = it.tx * TW;
            const unsigned ob = (unsigned)((((it.b * Cout) * H + ty0 + 2 * wave) * W + tx0) * 4);
            const bool full = ty0 + TH <= H && tx0 + TW <= W;
            if constexpr (RED) { if (it.b != ep_b) { scl = li < Cout ? 1.0f + d.scale[it.b * Cout + li] : 0.f; ep_b = it.b; } }
            unsigned so[4], vo[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                so[m] = ob + (unsigned)(((m >> 1) * W + (m & 1) * 16) * 4);
                vo[m] = ovoff;
                if (!full) {
                    const bool oky = ty0 + 2 * wave + (m >> 1) < H;
                    const bool okx = tx0 + (m & 1) * 16 + 4 * kq < W;
                    vo[m] = (okx && oky) ? ovoff : OOB;
                    if constexpr (RED) { if (!(okx && oky)) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
            }
            if constexpr (EP == BNERV_EP_BIAS || EP == BNERV_EP_PLAIN) {
#pragma unroll
                for (int m = 0; m < 4; ++m) bstore(ro, vo[m], so[m], acc[m] + bias_l);
            } else if constexpr (EP == BNERV_EP_BIAS_SIN) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 sv, cv;
                    sincos4_f(acc[m] + bias_l, &sv, &cv);
                    bstore(ro, vo[m], so[m], sv);
                    if (d.out2) bstore(ro2, vo[m], so[m], cv);
                }
            } else if constexpr (EP == BNERV_EP_BIAS_GELU) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 hv, gv;
                    gelu_pair4_f(acc[m] + bias_l, &hv, &gv);
                    bstore(ro, vo[m], so[m], hv);
                    if (d.out2) bstore(ro2, vo[m], so[m], gv);
                }
            } else if constexpr (EP == BNERV_EP_BIAS_TANH) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = tanhf(acc[m][e] + bias_l) * 0.5f + 0.5f;
                    bstore(ro, vo[m], so[m], r);
                }
            } else if constexpr (EP == BNERV_EP_BIAS_RES) {
                f32x4 a0[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) a0[m] = bload(ra0, vo[m], so[m]);
#pragma unroll
                for (int m = 0; m < 4; ++m) bstore(ro, vo[m], so[m], acc[m] + bias_l + a0[m]);
            } else {                                       // DGELU / DSIN
                f32x4 a0[4], a1[4], a2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    a0[m] = bload(ra0, vo[m], so[m]);
                    if constexpr (EP == BNERV_EP_DGELU_SAVED) a1[m] = bload(ra1, vo[m], so[m]);
                    if constexpr (EP == BNERV_EP_DSIN) {
                        a1[m] = bload(ra1, vo[m], so[m]);
                        a2[m] = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (d.aux2) a2[m] = bload(ra2, vo[m], so[m]);
                    }
                }
                float ps = 0.f, pt = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 r;
                    const f32x4 v = acc[m];
                    if constexpr (EP == BNERV_EP_DGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl * gelu_grad_f(a0[m][e]); ps = fmaf(v[e], gelu_f(a0[m][e]), ps); pt += v[e]; }
                    } else if constexpr (EP == BNERV_EP_DGELU_SAVED) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl * a0[m][e]; ps = fmaf(v[e], a1[m][e], ps); pt += v[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = (a1[m][e] + v[e] * scl) * a2[m][e]; ps = fmaf(v[e], a0[m][e], ps); pt += v[e]; }
                    }
                    bstore(ro, vo[m], so[m], r);
                }
                ps += __shfl_xor(ps, 16, 64);
                pt += __shfl_xor(pt, 16, 64);
                ps += __shfl_xor(ps, 32, 64);
                pt += __shfl_xor(pt, 32, 64);
                if (lane < 16) { s_red[(wave * 2 + 0) * 16 + lane] = ps; s_red[(wave * 2 + 1) * 16 + lane] = pt; }
            }
        }
        TRACE(6);
        prev = it;
        have_prev = true;
        it = nxt;
        inv_cur = inv_next;
    }
    if constexpr (RED) {
        lds_barrier();
        flush_partials(prev);
    }
    side_run_hosted(side, smem);
}

// Status (round 2, profiles/r02_split_*): correct in all three modes, but for ONE cout tile the staging work (split + LDS writes),
// not the matrix pipe, sets the pace: 36.9 us (bf16x6) / 27.8 us (bf16x3) against 40.0 us for the f32 lean kernel on the
// 12 -> 12 layer at 720x1280, and no gain on the data-gradient epilogues.  It is therefore OFF unless BNERV_SPLIT selects a mode;
// the layers with several cout tiles per staged input tile are where the 16-bit pipe pays.
static int split_mode() {                                  // BNERV_SPLIT = off (default) | bf16x6 | f16x3 | bf16x3
    static const int v = [] {
        const char* e = getenv("BNERV_SPLIT");
        if (!e || !strcmp(e, "off") || !strcmp(e, "0")) return -1;
        if (!strcmp(e, "f16x3")) return (int)SP_F16X3;
        if (!strcmp(e, "bf16x3")) return (int)SP_BF16X3;
        return (int)SP_BF16X6;
    }();
    return v;
}

template <int KS, int IN, int EP, int SP, int CB>
int launch_bf(hipStream_t st, KArgs& ka) {
    using BG = BfGeo<KS>;
    constexpr int NS = Split<SP>::NS;
    const bnerv_conv_desc& d = ka.d;
    ka.total_items = d.B * ka.tiles_x * ka.tiles_y;
    ka.magic_tiles = div_magic(ka.tiles_x * ka.tiles_y);
    ka.magic_tiles_x = div_magic(ka.tiles_x);
    const size_t lds = (size_t)NS * BG::PIECE + (size_t)BG::STEPS * NS * 1024 + (128 + 32 + 4) * sizeof(float);
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf_kernel<KS, IN, EP, SP, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_bf_kernel<KS, IN, EP, SP, CB>), 256, lds) != hipSuccess || nb < 1) nb = 1;
        blocks_per_cu = nb > 3 ? 3 : nb;
    }
    int grid = 256 * blocks_per_cu;
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(ka.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_bf_kernel<KS, IN, EP, SP, CB>), dim3(grid), dim3(256), lds, st, ka, side);
    BNERV_LAUNCH_CHECK("conv_bf");
    return BNERV_OK;
}

// ---------------------------------------------------------------------------------------------------------------- wide split kernel
// The same split core for the layers with MORE than 16 input or output channels (3x3, stride-1 output: the TAT convolutions,
// stride-1 block convs, PixelShuffle(2) up-convs and every data gradient of the 22..177-channel stages of C3 / C4, C1's up-convs and
// 30-channel stage -- conv.hip's lean2 scope without the tanh-grad prologue and the x3 / x5 shuffles).  Here the staged input tile of
// one 16-channel K chunk feeds NTB cout tiles, so the staging work that sets the pace of the one-tile kernel above is amortised over
// NTB times the matrix work -- this is where the 16-bit pipe pays.  Default on (wide_mode()).
//   * work item = (cout group of NTB tiles, sample, 8x32 tile); pipeline stage = (item, K chunk): the input chunk of the next stage
//     is loaded into registers under the MFMA phase of the current one and transformed / split / written to LDS after barrier (B);
//   * B fragments come PRE-SPLIT from a scratch buffer of the stream context (bf_wprep_kernel, one small launch per call in front
//     of this kernel: the weight tensor -> [group][chunk][tile][step][piece][lane] 16-B fragments), because the fragments of all
//     chunks do not fit LDS and splitting them again per (tile, chunk) would cost as much as the chunk's MFMAs; a stage copies its
//     46 KB (NTB = 3) from there (L2-resident) to LDS after barrier (B);
//   * the MFMA phase is a fragment pipeline (see the loop): A fragments one phase ahead, B pieces reloaded behind their last product;
//   * NTB shrinks at the low-resolution stages so that the work items still cover the chip (launch_bfw_ntb);
//   * channels beyond Cin inside the last chunk are never loaded: their A slots keep finite stale data and meet zero weights.
// Modes: bf16x6 (default) and bf16x3; the scaled f16 mode is not built for this kernel.
constexpr int AFF_MAX = 512;                                // affine table: 2 x (Cin rounded up to 16) floats of LDS, sized per layer

// one 16-B fragment slot-lane (group, chunk, tile, step, lane (q, n)): gathers its 8 k-elements from the OIHW array, splits them and
// writes the NS pieces -- every slot is written (zeros where the layer has no channel / tap), so the buffer needs no clearing pass
template <int NS>
__device__ __forceinline__ void wprep_slot(const float* __restrict__ w, u32x4* __restrict__ frag, int wCi, int transposed, int Cin, int Cout,
                                           int nck, int ntb, int f) {
    const int lane = f & 63, rest = f >> 6;
    const int st = rest % 5, r2 = rest / 5;
    const int nl = r2 % ntb, gc = r2 / ntb;
    const int c = gc % nck, g = gc / nck;
    const int n = lane & 15, q = lane >> 4;
    const int co = (g * ntb + nl) * 16 + n, tg = 2 * st + (q >> 1), ci0 = 16 * c + 8 * (q & 1);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ci0 + e;
        float v = 0.f;
        if (co < Cout && ci < Cin && tg < 9) v = transposed ? w[((size_t)ci * wCi + co) * 9 + (8 - tg)] : w[((size_t)co * wCi + ci) * 9 + tg];
        x[e] = v;
    }
    u32x4 pc[NS];
    split8<NS == 3 ? SP_BF16X6 : SP_BF16X3, 8>(x, pc);
#pragma unroll
    for (int p = 0; p < NS; ++p) frag[((size_t)rest * NS + p) * 64 + lane] = pc[p];
}

template <int NS>
__global__ __launch_bounds__(256) void bf_wprep_kernel(const float* __restrict__ w, u32x4* __restrict__ frag, int wCo, int wCi, int transposed,
                                                       int Cin, int Cout, int nck, int ntb, int nfrag) {
    for (int f = blockIdx.x * 256 + threadIdx.x; f < nfrag; f += gridDim.x * 256) wprep_slot<NS>(w, frag, wCi, transposed, Cin, Cout, nck, ntb, f);
}

// ---- weight-fragment plan of a repeated step (include/bnerv.h, bnerv_ctx_wplan_*): the fragments of every wide conv call of the
// step in ONE launch at its start (the weights do not change between the first forward conv and the last data gradient), instead of
// one small launch in front of each call (C1: 12, C3 / C4: ~50 launches of 3-6 us per step, each a pipeline drain of its own)
struct WPlanEntry {
    const float* w;
    unsigned long long slot0;                              // first 16-B slot of the entry in the arena
    int wCo, wCi, transposed, Cin, Cout, nck, ntb, nfrag, ns, block0;
};
}  // namespace
struct BfWPlan {
    int state = 0;                                         // 0: none, 1: recording, 2: frozen
    bool live = false;                                     // between wplan_run and wplan_end: the arena holds the current weights' fragments
    std::vector<WPlanEntry> e;
    u32x4* arena = nullptr;
    WPlanEntry* table = nullptr;
    int* blockmap = nullptr;
    int blocks = 0;
};
namespace {

__global__ __launch_bounds__(256) void bf_wprep_plan_kernel(const WPlanEntry* __restrict__ table, const int* __restrict__ blockmap, u32x4* __restrict__ arena) {
    const WPlanEntry e = table[blockmap[blockIdx.x]];
    const int f = ((int)blockIdx.x - e.block0) * 256 + threadIdx.x;
    if (f >= e.nfrag) return;
    if (e.ns == 3) wprep_slot<3>(e.w, arena + e.slot0, e.wCi, e.transposed, e.Cin, e.Cout, e.nck, e.ntb, f);
    else wprep_slot<2>(e.w, arena + e.slot0, e.wCi, e.transposed, e.Cin, e.Cout, e.nck, e.ntb, f);
}

// the same launch with the step's frame fetch as a second block range (bnerv_ctx_wplan_run_fetch)
struct FetchArgs { const float* clip; const double* norms; const float* sel; int n_frames; size_t frame_elems; float* dst; double* dst_norm; };
__global__ __launch_bounds__(256) void bf_wprep_plan_fetch_kernel(const WPlanEntry* __restrict__ table, const int* __restrict__ blockmap, u32x4* __restrict__ arena,
                                                                  const int n_plan, const FetchArgs fa) {
    if ((int)blockIdx.x >= n_plan) {                       // block-uniform
        fetch_frame_body(fa.clip, fa.norms, fa.sel, fa.n_frames, fa.frame_elems, fa.dst, fa.dst_norm, (int)blockIdx.x - n_plan, (int)gridDim.x - n_plan);
        return;
    }
    const WPlanEntry e = table[blockmap[blockIdx.x]];
    const int f = ((int)blockIdx.x - e.block0) * 256 + threadIdx.x;
    if (f >= e.nfrag) return;
    if (e.ns == 3) wprep_slot<3>(e.w, arena + e.slot0, e.wCi, e.transposed, e.Cin, e.Cout, e.nck, e.ntb, f);
    else wprep_slot<2>(e.w, arena + e.slot0, e.wCi, e.transposed, e.Cin, e.Cout, e.nck, e.ntb, f);
}

static bool wplan_same(const WPlanEntry& a, const bnerv_conv_desc& d, int nck, int ntb, int ns) {
    return a.w == d.w && a.wCo == d.wCo && a.wCi == d.wCi && a.transposed == d.transposed && a.Cin == d.Cin && a.Cout == d.Cout && a.nck == nck &&
           a.ntb == ntb && a.ns == ns;
}
// the arena fragments of this call, or nullptr (no live plan / no matching entry)
static const u32x4* wplan_lookup(const bnerv_ctx* ctx, const bnerv_conv_desc& d, int nck, int ntb, int ns) {
    const BfWPlan* p = ctx ? ctx->wplan : nullptr;
    if (!p || p->state != 2 || !p->live) return nullptr;
    for (const WPlanEntry& a : p->e)
        if (wplan_same(a, d, nck, ntb, ns)) return p->arena + a.slot0;
    return nullptr;
}
static void wplan_note(bnerv_ctx* ctx, const bnerv_conv_desc& d, int nck, int ntb, int ns, int nfrag) {
    BfWPlan* p = ctx ? ctx->wplan : nullptr;
    if (!p || p->state != 1) return;
    for (const WPlanEntry& a : p->e)
        if (wplan_same(a, d, nck, ntb, ns)) return;
    WPlanEntry a{};
    a.w = d.w; a.wCo = d.wCo; a.wCi = d.wCi; a.transposed = d.transposed; a.Cin = d.Cin; a.Cout = d.Cout; a.nck = nck; a.ntb = ntb; a.nfrag = nfrag; a.ns = ns;
    p->e.push_back(a);
}

struct WItem { int g, b, ty, tx, sp; };

// PS2 = 2: the output goes through PixelShuffle(2) (conv channel 4c + 2i + j at (y, x) -> out[c][2y + i][2x + j]; the up-convs) with
// paired 16-B stores; PS2 = 3: PixelShuffle(s), s = out_s in {3, 5}, with four 4-B stores s columns apart per accumulator quad.
// IN_UNSHUFFLE: the input is read through the inverse map from the shuffled tensor (data gradient of an up-conv; in_s == 2).
// (A software-pipelined one-block-per-CU form of this kernel was built and measured slower in round 4 -- DESIGN section 11.1,
// profiles/r04_bfw_pipe_trace.md -- and retired in round 5; git history has it.)
template <int IN, int EP, int SP, int NTB, int PS2>
__device__ __forceinline__ void conv_bfw_body(const KArgs& ka, const u32x4* __restrict__ wfrag, const int ngroups, const SidePack& side, const int vb, const int vgrid) {
    // (vb of vgrid: this launch's block index / size, or the conv role of a paired launch)
    constexpr int KS = 3;
    constexpr bool UNSH = (IN == BNERV_IN_UNSHUFFLE);
    static_assert(PS2 == 0 || EP == BNERV_EP_BIAS || EP == BNERV_EP_BIAS_SIN, "pixel-shuffle epilogues");
    using G = Geo<KS>;
    using BG = BfGeo<KS>;
    constexpr int NS = Split<SP>::NS;
    static_assert(!Split<SP>::SCALED, "the wide kernel is built for the bf16 modes");
    constexpr bool AFF = (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE);
    constexpr bool RED = (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    constexpr int SB_SLOTS = NTB * BG::STEPS * NS * 64;             // 16-B slots of one stage's B fragments
    constexpr int NWB = (SB_SLOTS + 255) / 256;
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_BYTES = NS * BG::PIECE, B_BYTES = SB_SLOTS * 16;
    char* s_a = reinterpret_cast<char*>(smem);             // [NS pieces][PIECE]
    char* s_b = s_a + A_BYTES;                             // [SB_SLOTS] 16-B fragment slots
    float* s_red = reinterpret_cast<float*>(s_b + B_BYTES);  // [4 waves][2][NTB * 16]
    float* s_aff = s_red + 4 * 2 * NTB * 16;                       // [2][aff_n]
    const int aff_n = (ka.d.Cin + 15) & ~15;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;
    const int nck = (Cin + 15) >> 4;

    const int xcd = vb & 7, lb = vb >> 3;
    const int nlb = (vgrid - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lb;
    for (int i = tid; i < A_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(s_a)[i] = u32x4{0u, 0u, 0u, 0u};   // no NaN patterns beside zero weights
    if (itx >= r1) { side_run_hosted(side, smem, vb, vgrid); return; }
    auto decode = [&](int i) __attribute__((always_inline)) {     // item order: tile fastest, then sample, then cout group
        WItem w;
        const int tiles = tiles_x * tiles_y;
        const int rest = fast_div(i, ka.magic_tiles), t = i - rest * tiles;
        w.ty = fast_div(t, ka.magic_tiles_x);
        w.tx = t - w.ty * tiles_x;
        const int q = rest / d.B;
        w.b = rest - q * d.B;
        w.sp = q / ngroups;                                // (split slowest: the splits of one tile run far apart, their slabs meet in reduce_slabs)
        w.g = q - w.sp * ngroups;
        return w;
    };

    const int s_h = wave >> 1;
    const int sidx = tid & 127;
    const bool has_slot = sidx < BG::HSLOT;
    const int s_sg = sidx % G::SEGS, s_r = (sidx / G::SEGS) % G::ROWS;
    // plain input: thread offset inside channel 8h of the tile, one plane further per channel.  Unshuffled input (UNSH): the
    // conv-space pair (4c + 2i, 4c + 2i + 1) at 4 pixels is 8 consecutive floats of row 2y + i of du[c]; thread offset in units of
    // the shuffled tensor, the (c, i) row added per pair.
    const int W2 = 2 * W, H2 = 2 * H;
    const int sin_ = UNSH ? d.in_s : 1, Wsi = sin_ * W, Hsi = sin_ * H;      // (UNSH with in_s = 3 / 5: four 4-B loads s columns apart per channel)
    const unsigned voff0 = !has_slot ? OOB : (UNSH ? (unsigned)(((sin_ * s_r) * Wsi + 4 * sin_ * s_sg) * 4) : (unsigned)((((8 * s_h) * H + s_r) * W + 4 * s_sg) * 4));
    const unsigned hw4 = (unsigned)(H * W * 4);
    const unsigned shift = UNSH ? (unsigned)((sin_ * G::PAD * Wsi + sin_ * G::XOFF) * 4) : (unsigned)((G::PAD * W + G::XOFF) * 4);
    const unsigned in_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, in_bytes);
    const bool ksplit_on = EP == BNERV_EP_PLAIN && ka.ksplit > 1;
    const __amdgpu_buffer_rsrc_t ro = ksplit_on ? make_rsrc(d.partial, 0, out_bytes * (unsigned)ka.ksplit) : make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ro2 = make_rsrc(((EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) && d.out2) ? d.out2 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(d.aux0 ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);
    int w_addr[4];
    bool w_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 4 * s_sg + j - G::COL0;
        w_ok[j] = has_slot && p >= 0 && p < BG::NP;
        w_addr[j] = bf_slot(s_r, w_ok[j] ? p : 0, s_h, BG::SPR) * 16;
    }

    f32x4 ra[8];
    int ra_valid = 0;                                      // channels of this thread's half that the loaded chunk really has
    auto issue = [&](const WItem& a, int c) __attribute__((always_inline)) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const int gy = ty0 + s_r - G::PAD, gx = tx0 + 4 * s_sg - G::XOFF;
        const unsigned vo = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? voff0 : OOB;
        int nv = Cin - 16 * c - 8 * s_h;
        nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
        ra_valid = nv;
        if constexpr (UNSH) {
            if (sin_ != 2) {                               // x3 / x5: conv channel c s^2 + i s + j at (y, x) = du[c][s y + i][s x + j]
                const int ss = sin_ * sin_;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ci = 16 * c + 8 * s_h + e;
                    const int cs = sin_ == 3 ? ci / 9 : ci / 25, rs = ci - cs * ss, is = sin_ == 3 ? rs / 3 : rs / 5;
                    const unsigned sb = (unsigned)(((((a.b * (Cin / ss) + cs) * Hsi) + sin_ * ty0 + is) * Wsi + sin_ * tx0 + (rs - is * sin_)) * 4);
                    f32x4 l = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (e < nv) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            l[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(vo == OOB ? OOB : vo + (unsigned)(t * sin_ * 4)), (int)sb, 0));
                    }
                    ra[e] = l;
                }
                return;
            }
            const int cd0 = (16 * c + 8 * s_h) >> 2;       // first du channel of this half (two per half: (cd0, i), (cd0 + 1, i))
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const int cd = cd0 + (e >> 2), i = (e >> 1) & 1;
                const unsigned sb = (unsigned)(((((a.b * (Cin >> 2) + cd) * H2) + 2 * ty0 + i) * W2 + 2 * tx0) * 4);
                f32x4 l0 = f32x4{0.f, 0.f, 0.f, 0.f}, l1 = l0;
                if (e < nv) { l0 = bload(rx, vo, sb); l1 = bload(rx, vo == OOB ? OOB : vo + 16u, sb); }
                ra[e] = f32x4{l0.x, l0.z, l1.x, l1.z};     // j = 0
                ra[e + 1] = f32x4{l0.y, l0.w, l1.y, l1.w}; // j = 1
            }
        } else {
            unsigned sb = (unsigned)((((a.b * Cin + 16 * c) * H + ty0) * W + tx0) * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < nv) ra[e] = bload(rx, vo, sb);
                else ra[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                sb += hw4;
            }
        }
    };
    // staging of one (item, chunk) stage after its loads (issue): transform + split + LDS write, one of the thread's four pixels per call
    float aff_sc[8], aff_sh[8];
    auto commit_begin = [&](const WItem& a, int c) __attribute__((always_inline)) {
        if constexpr (IN != BNERV_IN_PLAIN && !UNSH) {
            const int ty0 = a.ty * TH, tx0 = a.tx * TW;
            const int gy = ty0 + s_r - G::PAD, gx = tx0 + 4 * s_sg - G::XOFF;
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;     // zero padding applies AFTER the prologue
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ci = 16 * c + 8 * s_h + e;
                const bool on = AFF && inside && e < ra_valid;
                const float tsc = s_aff[ci], tsh = s_aff[aff_n + ci];      // (ci < aff_n: chunks cover Cin rounded up to 16; read unconditionally)
                aff_sc[e] = on ? tsc : 0.f;
                aff_sh[e] = on ? tsh : 0.f;
            }
        }
    };
    auto commit_pixel = [&](int j, char* sa) __attribute__((always_inline)) {
        if (ra_valid > 0) {                                // (a half without channels in this chunk keeps its stale slots: zero weights)
            float xs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = ra[e][j];
                if constexpr (IN != BNERV_IN_PLAIN && !UNSH) v = xform1<IN>(v, aff_sc[e], aff_sh[e], 0.f);
                xs[e] = v;
            }
            u32x4 pc[NS];
            split8<SP, 8>(xs, pc);
            if (w_ok[j]) {
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<u32x4*>(sa + p * BG::PIECE + w_addr[j]) = pc[p];
            }
        }
    };
    // B fragments of the stage: global (L2-resident, shared by every block) -> LDS.  Two-block form: through registers, after barrier (B)
    // (prefetching them under the MFMA phase was measured no faster there, and its 48 VGPRs are worth more to the fragment pipeline).
    auto copy_b = [&](const WItem& a, int c, char* sb) __attribute__((always_inline)) {
        const u32x4* src = wfrag + (size_t)(a.g * nck + c) * SB_SLOTS;
        u32x4 rb[NWB];
#pragma unroll
        for (int k = 0; k < NWB; ++k) {
            const int i = tid + k * 256;
            rb[k] = i < SB_SLOTS ? src[i] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < NWB; ++k) {
            const int i = tid + k * 256;
            if (i < SB_SLOTS) reinterpret_cast<u32x4*>(sb)[i] = rb[k];
        }
    };
    auto commit = [&](const WItem& a, int c) __attribute__((always_inline)) {          // the whole stage at once
        commit_begin(a, c);
#pragma unroll
        for (int j = 0; j < 4; ++j) commit_pixel(j, s_a);
        copy_b(a, c, s_b);
    };
    auto load_affine = [&](int b) __attribute__((always_inline)) {                 // s_aff[c] = 1 + scale[b][c], s_aff[aff_n + c] = shift[b][c]
        for (int i = tid; i < 2 * aff_n; i += 256) {
            const int c = i < aff_n ? i : i - aff_n;
            float v = 0.f;
            if (c < Cin) v = i < aff_n ? 1.0f + d.scale[b * Cin + c] : d.shift[b * Cin + c];
            s_aff[i] = v;
        }
    };
    auto flush_partials = [&](const WItem& a) __attribute__((always_inline)) {
        if (wave == 0) {
            for (int i = lane; i < 2 * NTB * 16; i += 64) {
                const int q = i / (NTB * 16), c = i - q * (NTB * 16);
                const float sm = ((s_red[(0 * 2 + q) * NTB * 16 + c] + s_red[(1 * 2 + q) * NTB * 16 + c]) + s_red[(2 * 2 + q) * NTB * 16 + c]) + s_red[(3 * 2 + q) * NTB * 16 + c];
                const int co = a.g * NTB * 16 + c;
                const size_t row = (size_t)(a.ty * tiles_x + a.tx) * d.B + a.b;
                if (co < Cout) d.partial[(row * 2 + q) * Cout + co] = sm;
            }
        }
    };

    int a_addr[BG::STEPS][2];
#pragma unroll
    for (int s = 0; s < BG::STEPS; ++s) {
        int t = 2 * s + (kq >> 1);
        if (t > G::T - 1) t = G::T - 1;
        const int ky = t / KS, kx = t - ky * KS;
#pragma unroll
        for (int xh = 0; xh < 2; ++xh) a_addr[s][xh] = bf_slot(2 * wave + ky, xh * 16 + li + kx, kq & 1, BG::SPR) * 16;
    }
    const int b_addr = lane * 16;

    WItem it = decode(itx);
    int aff_b = -1, ep_b = -1;
    float scl[NTB];
#pragma unroll
    for (int n = 0; n < NTB; ++n) scl[n] = 0.f;
    if constexpr (AFF) { load_affine(it.b); aff_b = it.b; }
    // K chunks of an item: all of them, or -- split-K, EP_PLAIN -- the item's `cps` chunks from sp * cps on
    auto c_first = [&](const WItem& a) __attribute__((always_inline)) { return a.sp * ka.cps; };
    auto c_stop = [&](const WItem& a) __attribute__((always_inline)) { return min(nck, a.sp * ka.cps + ka.cps); };
    
    issue(it, c_first(it));
    lds_barrier();                                         // zeroed s_a and the affine table visible
    commit(it, c_first(it));
          // the first stage's B fragments have landed
    WItem prev = it;
    bool have_prev = false;
    int trace_stage = 0; (void)trace_stage;
    while (itx < r1) {
        f32x4 acc[4][NTB];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool has_next_item = itx + nlb < r1;
        WItem nxt = it;
        if (has_next_item) nxt = decode(itx + nlb);
        const int c_lo = c_first(it), c_hi = c_stop(it), c_nxt = c_first(nxt);
        for (int c = c_lo; c < c_hi; ++c) {
            const bool last_chunk = c == c_hi - 1;
            const bool more = !last_chunk || has_next_item;
            WTR(0);
            lds_barrier();                                 // (A) this stage's s_a / s_b (and s_red of the previous item) visible
            WTR(1);
            const char* sa_cur = s_a;
            const char* sb_cur = s_b;
            
            if (more) issue(last_chunk ? nxt : it, last_chunk ? c_nxt : c + 1);
            if constexpr (RED) { if (c == c_lo && have_prev) flush_partials(prev); }
            WTR(2);
            {
                // Fragment pipeline.  Phase = (K step, pair of M tiles), ten per stage.  A fragments of the next phase are read (second
                // register set) before this phase's products are issued.  B fragments are shared by the two phases of a step and stay
                // single-buffered: the products are ordered by the B piece they read (pieces 2, 1, 0), and in the step's second phase
                // each piece of the NEXT step is read as soon as the last product using the old one is issued -- by the time the next
                // phase reaches that piece it has been in flight for a dozen MFMAs.  (Order of the six products inside a step does not
                // matter numerically: the accumulator already holds the earlier steps.)
                u32x4 afr[2][2][NS], bfr[NTB][NS];
                auto load_a = [&](int ph, int buf) __attribute__((always_inline)) {
                    const int s = ph >> 1, m0 = (ph & 1) * 2;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int p = 0; p < NS; ++p)
                            afr[buf][m][p] = *reinterpret_cast<const u32x4*>(sa_cur + p * BG::PIECE + a_addr[s][(m0 + m) & 1] + ((m0 + m) >> 1) * (BG::SPR * 16));
                };
                auto load_b = [&](int s, int p) __attribute__((always_inline)) {
#pragma unroll
                    for (int n = 0; n < NTB; ++n) bfr[n][p] = *reinterpret_cast<const u32x4*>(sb_cur + ((n * BG::STEPS + s) * NS + p) * 1024 + b_addr);
                };
                {
#pragma unroll
                for (int p = NS - 1; p >= 0; --p) load_b(0, p);
                load_a(0, 0);
#pragma unroll
                for (int ph = 0; ph < 2 * BG::STEPS; ++ph) {
                    const int s = ph >> 1, m0 = (ph & 1) * 2;
                    const bool reload = (ph & 1) && s + 1 < BG::STEPS;
                    if (ph + 1 < 2 * BG::STEPS) load_a(ph + 1, (ph + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#define BNERV_BF_PROD(pa, pb) _Pragma("unroll") for (int n = 0; n < NTB; ++n) _Pragma("unroll") for (int m = 0; m < 2; ++m) \
                        acc[m0 + m][n] = mfma16<SP>(afr[ph & 1][m][pa], bfr[n][pb], acc[m0 + m][n]);
                    if constexpr (NS == 3) {
                        BNERV_BF_PROD(0, 2)
                        if (reload) { load_b(s + 1, 2); __builtin_amdgcn_sched_barrier(0); }
                        BNERV_BF_PROD(1, 1)
                    }
                    BNERV_BF_PROD(0, 1)
                    if (reload) { load_b(s + 1, 1); __builtin_amdgcn_sched_barrier(0); }
                    if constexpr (NS == 3) { BNERV_BF_PROD(2, 0) }
                    BNERV_BF_PROD(1, 0)
                    BNERV_BF_PROD(0, 0)
#undef BNERV_BF_PROD
                    if (reload) load_b(s + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            }
            WTR(3);
            {
                lds_barrier();                             // (B) every wave is done reading this stage
                if (more) {
                    if constexpr (AFF) {
                        if (last_chunk && nxt.b != aff_b) { load_affine(nxt.b); lds_barrier(); aff_b = nxt.b; }
                    }
                    commit(last_chunk ? nxt : it, last_chunk ? c_nxt : c + 1);
                }
            }
            WTR(4);
            if (!last_chunk) ++trace_stage;
        }
        // ---- epilogue straight from the accumulators, one cout tile after the other
        WTR(5);
        {
            const int ty0 = it.ty * TH, tx0 = it.tx * TW;
            const int co_base = it.g * NTB * 16;
            const bool full = ty0 + TH <= H && tx0 + TW <= W;
            unsigned so[4];
            bool okm[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                so[m] = (unsigned)((((it.b * Cout) * H + ty0 + 2 * wave + (m >> 1)) * W + tx0 + (m & 1) * 16) * 4) + (ksplit_on ? (unsigned)it.sp * out_bytes : 0u);
                okm[m] = full || (ty0 + 2 * wave + (m >> 1) < H && tx0 + (m & 1) * 16 + 4 * kq < W);
            }
            float ps[NTB], pt[NTB];
#pragma unroll
            for (int n = 0; n < NTB; ++n) {
                const int co = co_base + 16 * n + li;
                const unsigned ovoff = co < Cout ? (unsigned)(((co * H) * W + 4 * kq) * 4) : OOB;
                const float bias_l = (EP != BNERV_EP_PLAIN && !RED && d.bias && co < Cout) ? d.bias[co] : 0.f;
                if constexpr (RED) { if (it.b != ep_b) scl[n] = co < Cout ? 1.0f + d.scale[it.b * Cout + co] : 0.f; }
                ps[n] = 0.f; pt[n] = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const unsigned vo = okm[m] ? ovoff : OOB;
                    f32x4 v = acc[m][n];
                    if constexpr (RED) { if (!okm[m]) v = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    if constexpr (PS2 == 3) {
                        // PixelShuffle(s), s = 3 or 5: conv channel c s^2 + i s + j at (y, x) -> out[c][s y + i][s x + j]; the lane's four
                        // pixels land s columns apart in one output row: four 4-B stores (the L2 merges the s^2 channels of a block)
                        const int sps = d.out_s, ss = sps * sps;
                        const int cps = sps == 3 ? co / 9 : co / 25, rps = co - cps * ss, ips = sps == 3 ? rps / 3 : rps / 5, jps = rps - ips * sps;
                        const int Hs = H * sps, Ws = W * sps;
                        const unsigned pvo = (okm[m] && co < Cout) ? (unsigned)((((cps * Hs + ips) * Ws) + jps + sps * 4 * kq) * 4) : OOB;
                        const unsigned pso = (unsigned)(((((it.b * (Cout / ss)) * Hs) + sps * (ty0 + 2 * wave + (m >> 1))) * Ws + sps * (tx0 + (m & 1) * 16)) * 4);
                        f32x4 q4s = v + bias_l, q4c = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (EP == BNERV_EP_BIAS_SIN) sincos4_f(v + bias_l, &q4s, &q4c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned vo_e = pvo == OOB ? OOB : pvo + (unsigned)(e * sps * 4);
                            const float s_ = q4s[e], c_ = q4c[e];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), ro, (int)vo_e, (int)pso, 0);
                            if constexpr (EP == BNERV_EP_BIAS_SIN) { if (d.out2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, c_), ro2, (int)vo_e, (int)pso, 0); }
                        }
                    } else if constexpr (PS2 == 2) {
                        // lane pair (j = 0 / 1 = even / odd conv channel) owns 8 consecutive output columns of row 2y + i: the even
                        // lane keeps pixels 0, 1 of both and stores columns 0..3, the odd lane pixels 2, 3 and columns 4..7
                        const bool odd = li & 1;
                        const int cps = co >> 2, ips = (co >> 1) & 1;
                        const unsigned pvo = (okm[m] && co < Cout) ? (unsigned)((((cps * H2 + ips) * W2) + 8 * kq + (odd ? 4 : 0)) * 4) : OOB;
                        const unsigned pso = (unsigned)(((((it.b * (Cout >> 2)) * H2) + 2 * (ty0 + 2 * wave + (m >> 1))) * W2 + 2 * (tx0 + (m & 1) * 16)) * 4);
                        auto pair_up = [&](f32x4 q) __attribute__((always_inline)) {
                            const float s0 = odd ? q.x : q.z, s1 = odd ? q.y : q.w;          // what the partner needs
                            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
                            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
                            return odd ? f32x4{r0, q.z, r1, q.w} : f32x4{q.x, r0, q.y, r1};
                        };
                        if constexpr (EP == BNERV_EP_BIAS) {
                            bstore(ro, pvo, pso, pair_up(v + bias_l));
                        } else {
                            f32x4 sv, cv;
                            sincos4_f(v + bias_l, &sv, &cv);
                            bstore(ro, pvo, pso, pair_up(sv));
                            if (d.out2) bstore(ro2, pvo, pso, pair_up(cv));
                        }
                    } else if constexpr (EP == BNERV_EP_BIAS || EP == BNERV_EP_PLAIN) {
                        bstore(ro, vo, so[m], v + bias_l);
                    } else if constexpr (EP == BNERV_EP_BIAS_SIN) {
                        f32x4 sv, cv;
                        sincos4_f(v + bias_l, &sv, &cv);
                        bstore(ro, vo, so[m], sv);
                        if (d.out2) bstore(ro2, vo, so[m], cv);
                    } else if constexpr (EP == BNERV_EP_BIAS_GELU) {
                        f32x4 hv, gv;
                        gelu_pair4_f(v + bias_l, &hv, &gv);
                        bstore(ro, vo, so[m], hv);
                        if (d.out2) bstore(ro2, vo, so[m], gv);
                    } else if constexpr (EP == BNERV_EP_BIAS_TANH) {
                        f32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = tanhf(v[e] + bias_l) * 0.5f + 0.5f;
                        bstore(ro, vo, so[m], r);
                    } else if constexpr (EP == BNERV_EP_BIAS_RES) {
                        const f32x4 a0 = bload(ra0, vo, so[m]);
                        bstore(ro, vo, so[m], v + bias_l + a0);
                    } else {                               // DGELU / DSIN
                        const f32x4 a0 = bload(ra0, vo, so[m]);
                        f32x4 a1 = f32x4{0.f, 0.f, 0.f, 0.f}, a2 = f32x4{1.f, 1.f, 1.f, 1.f}, r;
                        if constexpr (EP == BNERV_EP_DGELU_SAVED || EP == BNERV_EP_DSIN) a1 = bload(ra1, vo, so[m]);
                        if constexpr (EP == BNERV_EP_DSIN) { if (d.aux2) a2 = bload(ra2, vo, so[m]); }
                        if constexpr (EP == BNERV_EP_DGELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl[n] * gelu_grad_f(a0[e]); ps[n] = fmaf(v[e], gelu_f(a0[e]), ps[n]); pt[n] += v[e]; }
                        } else if constexpr (EP == BNERV_EP_DGELU_SAVED) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl[n] * a0[e]; ps[n] = fmaf(v[e], a1[e], ps[n]); pt[n] += v[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = (a1[e] + v[e] * scl[n]) * a2[e]; ps[n] = fmaf(v[e], a0[e], ps[n]); pt[n] += v[e]; }
                        }
                        bstore(ro, vo, so[m], r);
                    }
                }
            }
            if constexpr (RED) {
                ep_b = it.b;
#pragma unroll
                for (int n = 0; n < NTB; ++n) {
                    float a = ps[n], b2 = pt[n];
                    a += __shfl_xor(a, 16, 64); b2 += __shfl_xor(b2, 16, 64);
                    a += __shfl_xor(a, 32, 64); b2 += __shfl_xor(b2, 32, 64);
                    if (lane < 16) { s_red[(wave * 2 + 0) * NTB * 16 + n * 16 + lane] = a; s_red[(wave * 2 + 1) * NTB * 16 + n * 16 + lane] = b2; }
                }
            }
        }
        WTR(6);
        ++trace_stage;
        prev = it;
        have_prev = true;
        it = nxt;
        itx += nlb;
    }
    if constexpr (RED) {
        lds_barrier();
        flush_partials(prev);
    }
    side_run_hosted(side, smem, vb, vgrid);
}
template <int IN, int EP, int SP, int NTB, int PS2>
__global__ __launch_bounds__(256, 2) void conv_bfw_kernel(const KArgs ka, const u32x4* __restrict__ wfrag, const int ngroups, const SidePack side) {
    conv_bfw_body<IN, EP, SP, NTB, PS2>(ka, wfrag, ngroups, side, (int)blockIdx.x, (int)gridDim.x);
}

constexpr size_t LEAN_MAX_BYTES = 0x7ff00000;            // every tensor view must stay below the OOB marker offset

template <int KS, int IN, int EP, int SP>
int launch_cb(hipStream_t st, KArgs& ka) {
    return ka.d.Cin <= 12 ? launch_bf<KS, IN, EP, SP, 4>(st, ka) : launch_bf<KS, IN, EP, SP, 8>(st, ka);
}

template <int KS, int IN, int EP>
int launch_ns(hipStream_t st, KArgs& ka) {
    const int sp = split_mode();
    if (sp == SP_F16X3) return launch_cb<KS, IN, EP, SP_F16X3>(st, ka);
    if (sp == SP_BF16X3) return launch_cb<KS, IN, EP, SP_BF16X3>(st, ka);
    return launch_cb<KS, IN, EP, SP_BF16X6>(st, ka);
}

template <int KS>
int launch_mode(hipStream_t st, KArgs& ka) {
    const int in = ka.d.in_mode, ep = ka.d.ep_mode;
#define BNERV_CASE(I, E) if (in == I && ep == E) return launch_ns<KS, I, E>(st, ka);
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_TANH)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_PLAIN)
    if constexpr (KS == 3) {
        BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS)
        BNERV_CASE(BNERV_IN_GELU_AFFINE, BNERV_EP_BIAS_RES)
        BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DGELU)
        BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DSIN)
        BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS_GELU)
        BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS_RES)
        BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DGELU_SAVED)
    }
#undef BNERV_CASE
    return BF_NOT_HANDLED;
}

// ---- wide kernel: mode switch, launch, dispatch
static int wide_mode() {                                   // BNERV_SPLIT_WIDE = bf16x6 (default) | bf16x3 | off
    static const int v = [] {
        const char* e = getenv("BNERV_SPLIT_WIDE");
        if (!e) return (int)SP_BF16X6;
        if (!strcmp(e, "off") || !strcmp(e, "0")) return -1;
        if (!strcmp(e, "bf16x3")) return (int)SP_BF16X3;
        return (int)SP_BF16X6;
    }();
    return v;
}

// the pre-split B fragments of this call: from the stream context's plan when one is live, else prepared here into the context's scratch
// (NULL: no context, or the scratch would have to grow inside a graph capture -- the caller falls back to the f32 kernels)
template <int NS>
const void* weight_fragments(hipStream_t st, const bnerv_conv_desc& d, int nck, int ntb, int ngroups) {
    using BG = BfGeo<3>;
    const size_t slots = (size_t)ngroups * nck * ntb * BG::STEPS * NS * 64;
    const int nfrag = ngroups * nck * ntb * BG::STEPS * 64;
    const void* scratch = wplan_lookup(d.ctx, d, nck, ntb, NS);      // a live plan prepared this call's fragments at the start of the step
    if (scratch) return scratch;
    void* own = bnerv_ctx_scratch(d.ctx, slots * 16, st);
    if (!own) {                                         // no context, or it would have to grow inside a graph capture: f32 kernels
        static bool warned = false;                     // (different arithmetic and speed than the eager steps had: say so, once)
        if (!warned) {
            warned = true;
            fprintf(stderr, "[bnerv] wide split conv: no scratch for the weight fragments (%zu bytes; context %s) -- this call runs on the f32 MFMA kernels; "
                            "reserve the context's scratch before capturing (bnerv_ctx_reserve)\n", slots * 16, d.ctx ? "cannot grow here" : "missing");
        }
        return nullptr;
    }
    hipLaunchKernelGGL(bf_wprep_kernel<NS>, dim3(cdiv(nfrag, 256)), dim3(256), 0, st, d.w, reinterpret_cast<u32x4*>(own),
                       d.wCo, d.wCi, d.transposed, d.Cin, d.Cout, nck, ntb, nfrag);
    if (hipGetLastError() != hipSuccess) return nullptr;
    wplan_note(d.ctx, d, nck, ntb, NS, nfrag);
    return own;
}

template <int IN, int EP, int SP, int NTB, int PS2 = 0>
int launch_bfw(hipStream_t st, KArgs& ka) {
    using BG = BfGeo<3>;
    constexpr int NS = Split<SP>::NS;
    const bnerv_conv_desc& d = ka.d;
    const int nck = cdiv(d.Cin, 16), ngroups = cdiv(cdiv(d.Cout, 16), NTB);
    const void* scratch = weight_fragments<NS>(st, d, nck, NTB, ngroups);
    if (!scratch) return BF_NOT_HANDLED;
    ka.total_items = ka.ksplit * ngroups * d.B * ka.tiles_x * ka.tiles_y;
    ka.magic_tiles = div_magic(ka.tiles_x * ka.tiles_y);
    ka.magic_tiles_x = div_magic(ka.tiles_x);
    const size_t lds = (size_t)NS * BG::PIECE + (size_t)NTB * BG::STEPS * NS * 1024 + (size_t)(4 * 2 * NTB * 16 + ((IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE) ? 2 * ((d.Cin + 15) & ~15) : 0)) * sizeof(float);   // (affine table only where there is an affine prologue)
    // LDS depends on the layer through the affine table (2 x Cin floats): occupancy is looked up per distinct size
    static size_t attr_lds = 0, occ_lds = 0;
    static int blocks_per_cu = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bfw_kernel<IN, EP, SP, NTB, PS2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    if (lds != occ_lds) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_bfw_kernel<IN, EP, SP, NTB, PS2>), 256, lds) != hipSuccess || nb < 1) nb = 1;
        blocks_per_cu = NTB == 1 ? (nb > 3 ? 3 : nb) : (nb > 2 ? 2 : nb);   // (one-tile variants fit three blocks: 49 KB LDS, <= 168 VGPRs where the compiler got there)
        occ_lds = lds;
    }
    int grid = 256 * blocks_per_cu;
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_bfw_kernel<IN, EP, SP, NTB, PS2>), dim3(grid), dim3(256), lds, st, ka, reinterpret_cast<const u32x4*>(scratch), ngroups, side);
    BNERV_LAUNCH_CHECK("conv_bfw");
    return BNERV_OK;
}

template <int IN, int EP, int SP, int PS2 = 0>
int launch_bfw_ntb(hipStream_t st, KArgs& ka) {
    const int nt = cdiv(ka.d.Cout, 16);
    int ntb = nt <= 3 ? nt : (nt == 4 ? 2 : 3);
    // low-resolution stages: fewer cout tiles per block while the work items would leave half the chip idle (each group re-stages the
    // input tile, but on CUs that had nothing to do: the launch is as long as ONE block's chain of stages)
    const int tiles = ka.d.B * ka.tiles_x * ka.tiles_y;
    int min_items = 128;
    if (const char* e = getenv("BNERV_SPLIT_WIDE_MIN_ITEMS")) min_items = atoi(e);     // tests set 1 to keep the widest blocks on small shapes
    while (ntb > 1 && ka.ksplit * cdiv(nt, ntb) * tiles < min_items) --ntb;
    if (ntb == 1) return launch_bfw<IN, EP, SP, 1, PS2>(st, ka);
    if (ntb == 2) return launch_bfw<IN, EP, SP, 2, PS2>(st, ka);
    return launch_bfw<IN, EP, SP, 3, PS2>(st, ka);
}

template <int IN, int EP, int PS2 = 0>
int launch_bfw_sp(hipStream_t st, KArgs& ka) {
    return wide_mode() == SP_BF16X3 ? launch_bfw_ntb<IN, EP, SP_BF16X3, PS2>(st, ka) : launch_bfw_ntb<IN, EP, SP_BF16X6, PS2>(st, ka);
}

// ---- paired launch, wide form: the data gradient of a wide 3x3 layer (conv_bfw body) and the weight gradient that reads the same incoming
// gradient (wgrad_bfw body, wgrad_bfw_body.h) as interleaved roles of ONE grid -- equal block counts, the role of XCD-local slot s is
// (s + s / 32) & 1 (every CU holds both roles), both roles walk their XCD's slice of the image top to bottom at the same rate, so what
// one role reads of the shared gradient the other finds in that XCD's L2 (wgrad.hip::launch_pair has the 12-channel form and the
// measurements).  Only for layers that fill the chip in both roles; everything else keeps its own launches (or the low-resolution pair).
template <int CIN, int CEP, int NTB, int WIN, int MTW, int GM2>
__global__ __launch_bounds__(256, 2) void bfw_pair_kernel(const KArgs ka, const u32x4* __restrict__ wfrag, const int ngroups, const bnerv_wb::WArgs wa,
                                                          const int slots, const int ngn, const int ngm, const int nr, const SidePack side) {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, vb = ((slot >> 1) << 3) + xcd;
    if (((slot + (slot >> 5)) & 1) == 0) {
        SidePack none;
        none.n_jobs = 0; none.n_slices = 0;
        conv_bfw_body<CIN, CEP, SP_BF16X6, NTB, 0>(ka, wfrag, ngroups, none, vb, nr);
    } else {
        bnerv_wb::wgrad_bfw_body<WIN, SP_BF16X6, MTW, GM2>(wa, slots, ngn, ngm, side, vb, nr);
    }
}

template <int CIN, int CEP, int NTB, int WIN, int MTW, int GM2>
int launch_bfw_pair(hipStream_t st, KArgs& ka, const bnerv_wb::WArgs& wa, int slots, int ngn, int ngm, int nr) {
    using BG = BfGeo<3>;
    constexpr int NS = Split<SP_BF16X6>::NS;
    const bnerv_conv_desc& d = ka.d;
    const int nck = cdiv(d.Cin, 16), ngroups = cdiv(cdiv(d.Cout, 16), NTB);
    const void* scratch = weight_fragments<NS>(st, d, nck, NTB, ngroups);
    if (!scratch) return BF_NOT_HANDLED;
    ka.total_items = ngroups * d.B * ka.tiles_x * ka.tiles_y;
    ka.magic_tiles = div_magic(ka.tiles_x * ka.tiles_y);
    ka.magic_tiles_x = div_magic(ka.tiles_x);
    size_t lds = (size_t)NS * BG::PIECE + (size_t)NTB * BG::STEPS * NS * 1024 + (size_t)(4 * 2 * NTB * 16) * sizeof(float);
    size_t lw = (size_t)NS * bnerv_wb::BW_PIECE + 2 * bnerv_wb::BW_NPL * sizeof(float);
    const size_t red = (size_t)MTW * 16 * bnerv_wb::BW_NTW * 16 * sizeof(float);
    if (lw < red) lw = red;
    if (lds < lw) lds = lw;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bfw_pair_kernel<CIN, CEP, NTB, WIN, MTW, GM2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    SidePack side;
    bnerv_side_take(d.ctx, &side, 2 * nr);
    hipLaunchKernelGGL((bfw_pair_kernel<CIN, CEP, NTB, WIN, MTW, GM2>), dim3(2 * nr), dim3(256), lds, st, ka, reinterpret_cast<const u32x4*>(scratch), ngroups, wa, slots, ngn, ngm, nr, side);
    BNERV_LAUNCH_CHECK("bfw_pair");
    return BNERV_OK;
}

int launch_wide_mode(hipStream_t st, KArgs& ka) {
    const int in = ka.d.in_mode, ep = ka.d.ep_mode;
    if (ka.d.out_s == 3 || ka.d.out_s == 5) {              // up-conv forward through PixelShuffle(3 / 5): scatter stores
        if (in == BNERV_IN_PLAIN && ep == BNERV_EP_BIAS_SIN) return launch_bfw_sp<BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN, 3>(st, ka);
        if (in == BNERV_IN_PLAIN && ep == BNERV_EP_BIAS) return launch_bfw_sp<BNERV_IN_PLAIN, BNERV_EP_BIAS, 3>(st, ka);
        return BF_NOT_HANDLED;
    }
    if (ka.d.out_s == 2) {                                 // up-conv forward: conv -> bias -> PixelShuffle(2) [-> sin, cos]
        if (in == BNERV_IN_PLAIN && ep == BNERV_EP_BIAS_SIN) return launch_bfw_sp<BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN, 2>(st, ka);
        if (in == BNERV_IN_PLAIN && ep == BNERV_EP_BIAS) return launch_bfw_sp<BNERV_IN_PLAIN, BNERV_EP_BIAS, 2>(st, ka);
        return BF_NOT_HANDLED;
    }
    if (in == BNERV_IN_UNSHUFFLE) return ep == BNERV_EP_PLAIN ? launch_bfw_sp<BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN>(st, ka) : -1;
#define BNERV_CASE(I, E) if (in == I && ep == E) return launch_bfw_sp<I, E>(st, ka);
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_TANH)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_GELU_AFFINE, BNERV_EP_BIAS_RES)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DSIN)
    BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS_GELU)
    BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS_RES)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DGELU_SAVED)
#undef BNERV_CASE
    return BF_NOT_HANDLED;
}

}  // namespace

// ---- plan API (include/bnerv.h)
static void wplan_release_device(BfWPlan* p) {
    if (p->arena) (void)hipFree(p->arena);
    if (p->table) (void)hipFree(p->table);
    if (p->blockmap) (void)hipFree(p->blockmap);
    p->arena = nullptr; p->table = nullptr; p->blockmap = nullptr; p->blocks = 0;
}
void bnerv_wplan_free(bnerv_ctx* ctx) {
    if (!ctx || !ctx->wplan) return;
    wplan_release_device(ctx->wplan);
    delete ctx->wplan;
    ctx->wplan = nullptr;
}
extern "C" int bnerv_ctx_wplan_record(bnerv_ctx* ctx) {
    BNERV_REQUIRE(ctx != nullptr, "ctx_wplan_record: null context");
    if (!ctx->wplan) ctx->wplan = new (std::nothrow) BfWPlan();
    BNERV_REQUIRE(ctx->wplan != nullptr, "ctx_wplan_record: out of memory");
    BfWPlan* p = ctx->wplan;
    if (p->arena) {                                        // a previous plan's arena may still be read by work in flight
        if (hipDeviceSynchronize() != hipSuccess) return bnerv_set_error(BNERV_E_LAUNCH, "ctx_wplan_record: device synchronize failed");
        wplan_release_device(p);
    }
    p->e.clear();
    p->live = false;
    p->state = 1;
    return BNERV_OK;
}
extern "C" int bnerv_ctx_wplan_freeze(bnerv_ctx* ctx) {
    BNERV_REQUIRE(ctx != nullptr && ctx->wplan != nullptr && ctx->wplan->state == 1, "ctx_wplan_freeze: no recording in progress");
    BfWPlan* p = ctx->wplan;
    p->state = 0;
    p->live = false;
    if (p->e.empty()) return 0;
    unsigned long long slots = 0;
    int blocks = 0;
    for (WPlanEntry& a : p->e) {
        a.slot0 = slots; a.block0 = blocks;
        slots += (unsigned long long)(a.nfrag >> 6) * a.ns * 64;
        blocks += cdiv(a.nfrag, 256);
    }
    std::vector<int> map((size_t)blocks);
    for (size_t i = 0; i < p->e.size(); ++i)
        for (int b = 0; b < cdiv(p->e[i].nfrag, 256); ++b) map[(size_t)p->e[i].block0 + b] = (int)i;
    const size_t tbytes = p->e.size() * sizeof(WPlanEntry), mbytes = map.size() * sizeof(int);
    if (hipMalloc(reinterpret_cast<void**>(&p->arena), (size_t)slots * 16) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&p->table), tbytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&p->blockmap), mbytes) != hipSuccess || hipMemcpy(p->table, p->e.data(), tbytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->blockmap, map.data(), mbytes, hipMemcpyHostToDevice) != hipSuccess) {
        wplan_release_device(p);
        p->e.clear();
        return bnerv_set_error(BNERV_E_WS, "ctx_wplan_freeze: cannot allocate the fragment arena (%llu bytes)", slots * 16ull);
    }
    p->blocks = blocks;
    p->state = 2;
    return (int)p->e.size();
}
extern "C" int bnerv_ctx_wplan_run(bnerv_ctx* ctx, void* stream) {
    BNERV_REQUIRE(ctx != nullptr, "ctx_wplan_run: null context");
    BfWPlan* p = ctx->wplan;
    if (!p || p->state != 2 || p->blocks == 0) return BNERV_OK;       // nothing planned: every call prepares its own fragments
    hipLaunchKernelGGL(bf_wprep_plan_kernel, dim3(p->blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p->table, p->blockmap, p->arena);
    BNERV_LAUNCH_CHECK("bf_wprep_plan");
    p->live = true;
    return BNERV_OK;
}
// bnerv_ctx_wplan_run + bnerv_fetch_frame as ONE launch (the two are independent and open every captured step that trains a resident clip).
// Returns 1 when there is no frozen plan with work: the caller issues bnerv_fetch_frame (and nothing else) as before.
extern "C" int bnerv_ctx_wplan_run_fetch(bnerv_ctx* ctx, void* stream, const float* clip, const double* norms, const float* sel_dev, int n_frames, size_t frame_elems,
                                         float* dst_img, double* dst_norm) {
    BNERV_REQUIRE(ctx != nullptr, "ctx_wplan_run_fetch: null context");
    BNERV_REQUIRE(clip && sel_dev && dst_img && n_frames > 0 && frame_elems > 0, "ctx_wplan_run_fetch: bad fetch args");
    BNERV_REQUIRE((reinterpret_cast<uintptr_t>(clip) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_img) & 15) == 0 && (frame_elems % 4 == 0 || n_frames == 1),
                  "ctx_wplan_run_fetch: 16-byte aligned frames required");
    BfWPlan* p = ctx->wplan;
    if (!p || p->state != 2 || p->blocks == 0) return 1;
    const size_t n4 = frame_elems / 4;
    int fb = (int)((n4 + 255) / 256);
    if (fb > 2048) fb = 2048;
    if (fb < 1) fb = 1;
    const FetchArgs fa{clip, norms, sel_dev, n_frames, frame_elems, dst_img, dst_norm};
    hipLaunchKernelGGL(bf_wprep_plan_fetch_kernel, dim3(p->blocks + fb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p->table, p->blockmap, p->arena, p->blocks, fa);
    BNERV_LAUNCH_CHECK("bf_wprep_plan_fetch");
    p->live = true;
    return BNERV_OK;
}
extern "C" int bnerv_ctx_wplan_end(bnerv_ctx* ctx) {
    if (ctx && ctx->wplan) ctx->wplan->live = false;
    return BNERV_OK;
}
extern "C" int bnerv_ctx_wplan_entries(const bnerv_ctx* ctx) { return (ctx && ctx->wplan && ctx->wplan->state == 2) ? (int)ctx->wplan->e.size() : 0; }

#ifdef BNERV_TRACE_BFW
extern "C" int bnerv_debug_trace_read_bfw(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_bfw), sizeof(g_trace_bfw)); }
#endif
#ifdef BNERV_TRACE
extern "C" int bnerv_debug_trace_read_bf(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_bf), sizeof(g_trace_bf)); }
#endif

// Called by bnerv_conv_igemm (conv.hip) after argument validation.  Returns BF_NOT_HANDLED (+1: not a BNERV_E_* value, so a real
// argument / launch error raised on this path is never mistaken for it) when the shape / mode is not this kernel's -- the caller
// then takes its f32-MFMA kernels -- otherwise the launch status.
int bnerv_convbf_try(hipStream_t st, const bnerv_conv_desc& d, int vec, int ksplit, int chunks_per_split) {
    if (!vec || d.in_mode == BNERV_IN_TANHGRAD) return BF_NOT_HANDLED;
    if (ksplit > 1 && (d.ep_mode != BNERV_EP_PLAIN || !d.partial || d.out_s != 1)) return BF_NOT_HANDLED;
    const bool shuffled = d.out_s != 1 || d.in_mode == BNERV_IN_UNSHUFFLE;                // up-conv forward / its data gradient
    if (d.out_s != 1 && !((d.out_s == 2 || d.out_s == 3 || d.out_s == 5) && d.Cout % (d.out_s * d.out_s) == 0)) return BF_NOT_HANDLED;
    if ((size_t)d.B * d.Cout * d.H * d.W * 4 >= LEAN_MAX_BYTES) return BF_NOT_HANDLED;      // (the shuffled output is addressed as one buffer)
    if (d.in_mode == BNERV_IN_UNSHUFFLE && (!(d.in_s == 2 || d.in_s == 3 || d.in_s == 5) || d.Cin % (d.in_s * d.in_s) != 0 || d.out_s != 1)) return BF_NOT_HANDLED;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    if ((size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 >= LEAN_MAX_BYTES) return BF_NOT_HANDLED;
    KArgs ka;
    ka.d = d;
    ka.tiles_x = cdiv(d.W, TW);
    ka.tiles_y = cdiv(d.H, TH);
    ka.ksplit = ksplit > 1 ? ksplit : 1;
    ka.cps = ksplit > 1 ? chunks_per_split : cdiv(d.Cin, 16);
    if ((size_t)ka.ksplit * d.B * d.Cout * d.H * d.W * 4 >= LEAN_MAX_BYTES) return BF_NOT_HANDLED;
    const bool narrow = d.Cout <= 16 && d.Cin <= 16 && !shuffled;
    if (narrow) {                                          // one cout tile, one K chunk: opt-in (see split_mode)
        if (split_mode() < 0 || ksplit > 1 || d.Cin <= 8 || d.wCo > 16 || d.wCi > 16) return BF_NOT_HANDLED;
        return d.k == 1 ? launch_mode<1>(st, ka) : launch_mode<3>(st, ka);
    }
    // several cout tiles and / or K chunks: the wide kernel, where the image is big enough to fill the chip with its work items
    if (wide_mode() < 0 || d.k != 3 || !d.ctx) return BF_NOT_HANDLED;
    const bool affine = d.in_mode == BNERV_IN_AFFINE || d.in_mode == BNERV_IN_GELU_AFFINE;
    if (affine && d.Cin > AFF_MAX) return BF_NOT_HANDLED;
    int min_tiles = 16;                                    // (measured on C1 / C3 / C4: 16 >= 32 >= 64 >= 256; below it the f32 kernels' split policies win)
    if (const char* e = getenv("BNERV_SPLIT_WIDE_MIN_TILES")) min_tiles = atoi(e);     // tests lower it to reach the kernel with small shapes
    if (d.B * ka.tiles_x * ka.tiles_y < min_tiles) return BF_NOT_HANDLED;
    return launch_wide_mode(st, ka);
}

// 1: not a pair of this form (the caller goes on).  On BNERV_OK *n_slabs is the weight gradient's slab count.
// w_mtw / ngn / ngm / nat_slots: the weight gradient's plan for its own launch (wgrad.hip::bw_plan).
int bnerv_convbf_pair_try(hipStream_t st, const bnerv_conv_desc& d, int vec, const bnerv_wb::WArgs& wa, int w_mtw, int ngn, int ngm, int nat_slots, int* n_slabs) {
    { static const bool off = [] { const char* e = getenv("BNERV_PAIR_BFW"); return e && e[0] == '0'; }(); if (off) return 1; }
    if (wide_mode() != (int)SP_BF16X6 || !vec || d.k != 3 || !d.ctx || d.out_s != 1) return 1;
    const bool uns = d.in_mode == BNERV_IN_UNSHUFFLE;
    if (!(d.in_mode == BNERV_IN_PLAIN || (uns && d.in_s == 2 && d.Cin % 4 == 0))) return 1;
    if (d.Cout <= 16 && d.Cin <= 16 && !uns) return 1;                                     // (the one-tile kernels' layer)
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    if ((size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 >= LEAN_MAX_BYTES) return 1;
    const bnerv_wgrad_desc& w = wa.d;
    KArgs ka;
    ka.d = d;
    ka.tiles_x = cdiv(d.W, TW);
    ka.tiles_y = cdiv(d.H, TH);
    ka.ksplit = 1;
    ka.cps = cdiv(d.Cin, 16);
    const int nt = cdiv(d.Cout, 16), ntb = nt <= 3 ? nt : (nt == 4 ? 2 : 3), ngroups = cdiv(nt, ntb);
    const int wg = ngn * ngm;
    const int slots = wg <= 32 ? 32 / wg : 0;              // 8 x wg x slots = the blocks per role (<= 256: two blocks per CU in all)
    if (slots < 1) return 1;
    const int nr = 8 * wg * slots;
    // both roles must fill the chip on their own: the conv has at least two items per block, the weight gradient's own plan at least twice the slots
    // both roles must fill the chip on their own: about one item per resident block or more (7/8 of the blocks busy).  Round 5: the
    // threshold was two items per block, which sent C1's 12 -> 48 up-conv backward at 180x320 (230 tiles for 256 blocks per role) to the
    // low-resolution pair -- 46.6 us there against 25.3 us here (profiles/r05_timeline_c1.md); at 90x160 (60 tiles) the low-resolution
    // pair stays faster (17.0 against 20.1 us).  BNERV_PAIR_BFW_FILL=<eighths> overrides (16 = the old rule, 0 = always).
    static const int fill = [] { const char* e = getenv("BNERV_PAIR_BFW_FILL"); return e ? atoi(e) : 7; }();
    if (ngroups * d.B * ka.tiles_x * ka.tiles_y * 8 < fill * nr || nat_slots * 8 < fill * slots) return 1;
    int rc = 1;
#define BNERV_BP(CI, CE, NT, WI, MT, G2) if (d.in_mode == CI && d.ep_mode == CE && ntb == NT && w.in_mode == WI && w_mtw == MT && gm2 == G2) \
        rc = launch_bfw_pair<CI, CE, NT, WI, MT, G2>(st, ka, wa, slots, ngn, ngm, nr);
    const int gm2 = w.g_s == 2 ? 1 : 0;
    if (w.g_s > 2 || w.g_mode == BNERV_IN_TANHGRAD) return 1;
    // an up-conv's (dW | d input) where the data gradient has ONE output tile (<= 16 channels: the 12 -> 48 up-convs of the 12-channel
    // stages): unshuffle(2) prologue | plain input, shuffled gradient.  Measured and NOT instantiated: the 38-channel TAT / block convs
    // of C3 (both roles three tiles wide, matrix-bound) ran 18.54 ms per step paired against 18.21 ms with their own launches.
    BNERV_BP(BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN, 1, BNERV_IN_PLAIN, 3, 1)
    BNERV_BP(BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN, 1, BNERV_IN_PLAIN, 2, 1)
#undef BNERV_BP
    if (rc == BNERV_OK) *n_slabs = 8 * slots;
    return rc == BF_NOT_HANDLED ? 1 : rc;
}
