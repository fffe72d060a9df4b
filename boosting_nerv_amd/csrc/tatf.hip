// tatf.hip -- the TAT residual block FORWARD (ResBlock_SFT.forward, reference model_blocks.py:74-89) of the 12-channel stages as ONE launch:
//
//     out = x0 + conv1( sft1( gelu( conv0( sft0(x0) ) ) ) ),     sft_i(a) = a (1 + scale_i[b, c]) + shift_i[b, c]      (model_blocks.py:92-105)
//
// with h = gelu(v) and gp = gelu'(v) (v = conv0(...) + b0) still written for the backward (ops._tat_backward reads both), and neither of
// them written in the decode path.  Until round 4 this was two launches (conv0 with the GELU-pair epilogue, conv1 with the residual
// epilogue): h crossed HBM twice, x0 was read twice, and each launch paid its own block prologue and tile tail.
//
// One block = 256 threads = 4 waves on ONE compute unit by itself (its LDS image is ~126 KB): one wave per SIMD, the whole register file,
// so BOTH weight sets stay resident in registers in the 4x4x1 form of conv4.hip (v_mfma_f32_4x4x1_16b_f32: lane = pixel, one register =
// four weight quads broadcast with BLGP, 84 registers per 12 x 12 x 9 weight set, no 12 -> 16 channel padding).
//
// Per 16 x 32 output tile:
//   * the x0 tile with a TWO-pixel halo (20 rows x 40 columns x 12 channels, channel-planar) enters LDS by LDS-DMA
//     (buffer_load ... lds), double-buffered: the next tile's loads are issued at the top of the tile and land under its matrix phases;
//   * phase 0: conv0 on the 18 x 34 region the second convolution needs.  Interior: every wave owns 4 tile rows = two "wave-ops" of 2 rows x
//     32 pixels; the one-pixel ring around the tile (52 four-pixel blocks: top / bottom rows, left / right columns) is a third, 13/16
//     occupied wave-op per wave.  Epilogue from the accumulators: bias, folded shift term, the GELU pair; h goes to an LDS tile
//     (channel-planar, ZERO outside the image: conv1's zero padding applies to its affine input) and, with gp, to HBM for the tile's own pixels;
//   * barrier; phase 1: conv1 from the LDS tile of h, two wave-ops per wave; epilogue: bias, folded shift term, + x0 straight from the
//     input tile in LDS (no second HBM read), one 16-byte store per accumulator;
//   * barrier (next input tile landed; h tile free).
// 5 wave-ops per wave and tile against 4 for two separate convolutions (the halo recompute), one 44 MB read and one launch less.
//
// The TAT affines are folded as in conv4_body.h: conv_w(x (1 + s) + t) = conv_{w (1 + s)}(x) + sum over the taps INSIDE the image of
// t_ci w -- the tiles hold the raw tensors, the weights are scaled once per sample, interior pixels add a per-channel constant and pixels
// next to the image border add the taps that exist.
#include "conv_common.h"

namespace {
using namespace bnerv_conv;

constexpr int FTH = 16, FTW = 32;                          // output tile
constexpr int NCH = 12, NG = 3;                            // staged channels, output-channel quads
constexpr int NQD = 9 * NG, NWR = (NQD + 3) / 4, QPAD = NWR * 4;     // weight quads per input channel (tap-major), registers, padded quads
constexpr int IN_ROWS = FTH + 4, IN_RS = 40, IN_PLANE = IN_ROWS * IN_RS;       // x0 tile: rows ty0 - 2 .., columns tx0 - 4 .. tx0 + 35
constexpr int H_ROWS = FTH + 2, H_RS = 40, H_PLANE = H_ROWS * H_RS;           // h tile: rows ty0 - 1 .., columns tx0 - 4 .. tx0 + 35
constexpr int IN_SLOTS = NCH * IN_ROWS * (IN_RS / 4);      // 16-byte DMA slots of one input tile
constexpr int NPRE = (IN_SLOTS + 255) / 256;               // DMA passes per wave
constexpr int S_IN = NPRE * 256 * 4;                       // floats per input buffer (the idle slots at its end are the overflow pad of the last plane)
constexpr int S_W = NCH * QPAD * 4;                        // floats per staged weight set
constexpr int RING = 2 * 10 + 2 * FTH;                     // four-pixel blocks of the one-pixel ring: top, bottom (10 each), left, right (FTH each)
constexpr int RING_PER_WAVE = RING / 4;                    // 13
static_assert(RING % 4 == 0 && RING_PER_WAVE <= 16, "the ring is one wave-op per wave");
constexpr int LDS_FLOATS = 4 + 2 * S_IN + NCH * H_PLANE + 8 + 2 * S_W + 64 + 2 * 9 * 16;

#ifdef BNERV_TRACE_TAT
__device__ unsigned long long g_trace_tat[256 * 4 * 4 * 16];      // [block][wave][tile < 4][stamp]
#define TTR(slot) do { if (lane == 0 && trace_tile < 4) g_trace_tat[(((int)blockIdx.x * 4 + wave) * 4 + trace_tile) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TTR(slot) do {} while (0)
#endif

struct TatArgs {
    bnerv_tat_desc d;
    int tiles_x, tiles_y, total_items;
    unsigned magic_tiles, magic_tiles_x;
};

template <int G>
__device__ __forceinline__ f32x4 mfma_q(float a, float w, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c, 0, 0, 4 + G); }
__device__ __forceinline__ f32x4 mfma_qsel(int g, float a, float w, f32x4 c) {          // g is a constant after unrolling
    switch (g) {
        case 0: return mfma_q<0>(a, w, c);
        case 1: return mfma_q<1>(a, w, c);
        case 2: return mfma_q<2>(a, w, c);
        default: return mfma_q<3>(a, w, c);
    }
}
// LDS-DMA of one 16-byte slot per lane (LDS address = m0 + 16 * lane), through inline asm so that the waits are placed by hand (conv4_body.h)
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one wave-op: 64 pixels x 12 output channels x (12 channels x 9 taps); a = the lane's own pixel at tap (0, 0) of channel 0.
// The block runs ONE wave per SIMD, and a lone wave sees the full LDS round trip (~100+ cycles for a 4-byte read against 72 cycles of
// MFMA per tap row): the A values are therefore read DEPTH tap rows ahead (3 registers per row -- the register file is this wave's alone).
template <int PLANE, int RS>
__device__ __forceinline__ void kloop(f32x4 (&acc)[NG], const float* a, const float (&wr)[NCH][NWR]) {
    constexpr int DEPTH = 4, NT = NCH * 3;                 // tap rows in flight; tap rows of the K loop (t = ci * 3 + ky)
    float av[NT + DEPTH][3];                               // (fully unrolled: every index is a constant, a value lives DEPTH rows)
#pragma unroll
    for (int n = 0; n < NG; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DEPTH; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) av[t][kx] = a[(t / 3) * PLANE + (t % 3) * RS + kx];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ci = t / 3, ky = t % 3, tn = t + DEPTH;
        if (tn < NT) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) av[tn][kx] = a[(tn / 3) * PLANE + (tn % 3) * RS + kx];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const int q = (ky * 3 + kx) * NG + n;
                acc[n] = mfma_qsel(q & 3, av[t][kx], wr[ci][q >> 2], acc[n]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the folded shift term of 4 consecutive pixels (row gy, columns gx0 ..) of output channel `ch`: all nine taps for an interior pixel
// (`full`), the taps that fall inside the image otherwise.  tab = [9 taps][16 channels].
__device__ __forceinline__ f32x4 shift_term(const float* tab, int ch, float full, bool border, int gy, int gx0, int H, int W) {
    if (!border) return f32x4{full, full, full, full};
    float col[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float t = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) t += ((unsigned)(gy + ky - 1) < (unsigned)H) ? tab[(ky * 3 + kx) * 16 + ch] : 0.f;
        col[kx] = t;
    }
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) t += ((unsigned)(gx0 + e + kx - 1) < (unsigned)W) ? col[kx] : 0.f;
        r[e] = t;
    }
    return r;
}

template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void tat_fused_kernel(const TatArgs ka) {
    const bnerv_tat_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem + 4;                                // [2][S_IN]; 16 bytes in front: the ring's leftmost column reads one float before a row
    float* s_h = s_in + 2 * S_IN;                          // [12][H_ROWS][H_RS] gelu(conv0), zero outside the image
    float* s_w = s_h + NCH * H_PLANE + 8;                  // [2][ci][quad][4] raw weight quads of conv0 / conv1
    float* s_aff = s_w + 2 * S_W;                          // [4][16]: 1 + scale0, shift0, 1 + scale1, shift1 of the current sample
    float* s_beta = s_aff + 64;                            // [2][9 taps][16]: sum_ci shift[ci] W(co, ci, tap)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 3, lb = lane >> 2;               // output channel within a quad; four-pixel block of the wave-op
    const int C = d.C, H = d.H, W = d.W;
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;

    // ---- item range: XCD x owns a contiguous slice of the tile list, its blocks take it round-robin (neighbouring tiles share an L2)
    const int vb = (int)blockIdx.x, vgrid = (int)gridDim.x;
    const int xcd = vb & 7, lbk = vb >> 3;
    const int nlb = (vgrid - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lbk;
    if (itx >= r1) return;
    const int step_q = fast_div(nlb, ka.magic_tiles_x), step_r = nlb - step_q * tiles_x;
    LItem it;
    {
        const int tiles = tiles_x * tiles_y;
        it.b = fast_div(itx, ka.magic_tiles);
        const int t = itx - it.b * tiles;
        it.ty = fast_div(t, ka.magic_tiles_x);
        it.tx = t - it.ty * tiles_x;
    }
    auto advance = [&](LItem a) {
        a.tx += step_r;
        a.ty += step_q;
        if (a.tx >= tiles_x) { a.tx -= tiles_x; ++a.ty; }
        while (a.ty >= tiles_y) { a.ty -= tiles_y; ++a.b; }
        return a;
    };

    // ---- DMA slot constants: slot = (channel, tile row, 4-px segment); thread t owns slots t, t + 256, ...; LDS home = byte 16 * slot
    auto slot_geom = [&](int k, int& c, int& r, int& sg) {
        const int sidx = tid + k * 256;
        c = sidx / (IN_ROWS * (IN_RS / 4));
        const int rem = sidx - c * (IN_ROWS * (IN_RS / 4));
        r = rem / (IN_RS / 4);
        sg = rem - r * (IN_RS / 4);
    };
    unsigned voff[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        voff[k] = (tid + k * 256 < IN_SLOTS && c < C) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;      // out of range: the DMA writes zeros
    }
    const unsigned shift = (unsigned)((2 * W + 4) * 4);    // the x0 view starts 2 rows + 4 columns early: offsets >= 0
    const unsigned t_bytes = (unsigned)((size_t)d.B * C * H * W * 4);
    i32x4 rx;
    {
        const uintptr_t base = reinterpret_cast<uintptr_t>(d.x0) - shift;
        rx[0] = (int)(unsigned)(base & 0xffffffffu);
        rx[1] = (int)(unsigned)((base >> 32) & 0xffffu);
        rx[2] = (int)(t_bytes + shift);
        rx[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.out, 0, t_bytes);
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(TRAIN ? d.h : d.out, 0, t_bytes);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(TRAIN ? d.gp : d.out, 0, t_bytes);
    const unsigned lds_in = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_in + (unsigned)wave * 1024u;

    auto issue = [&](const LItem& a, int buf) {
        const int ty0 = a.ty * FTH, tx0 = a.tx * FTW;
        const unsigned sb = (unsigned)((((a.b * C) * H + ty0) * W + tx0) * 4);
        const unsigned lbase = lds_in + (unsigned)buf * (unsigned)(S_IN * 4);
        const bool interior = ty0 >= 2 && ty0 + FTH + 2 <= H && tx0 >= 4 && tx0 + FTW + 4 <= W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rx, voff[k], sb, lbase + (unsigned)k * 4096u);
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                int c, r, sg;
                slot_geom(k, c, r, sg);
                const int gy = ty0 + r - 2, gx = tx0 + 4 * sg - 4;
                const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                dma16(rx, in ? voff[k] : OOB, sb, lbase + (unsigned)k * 4096u);
            }
        }
    };

    // ---- lane constants of the epilogues: lane (lb, lj) holds 4 consecutive pixels (row lb >> 3 of the wave-op's two rows, columns 4 (lb & 7) ..)
    // of output channels 4 n + lj
    const unsigned ovoff = (unsigned)(((lj * H + (lb >> 3)) * W + 4 * (lb & 7)) * 4);
    const unsigned nstep = (unsigned)(4 * H * W * 4);
    // ring block of this lane (the same for every tile): rb = 13 wave + lb
    int rrow = 0, rcol = 0;
    const bool ring_lane = lb < RING_PER_WAVE;
    {
        const int rb = RING_PER_WAVE * wave + (ring_lane ? lb : 0);
        if (rb < 10) { rrow = -1; rcol = 4 * rb - 4; }
        else if (rb < 20) { rrow = FTH; rcol = 4 * (rb - 10) - 4; }
        else if (rb < 20 + FTH) { rrow = rb - 20; rcol = -4; }
        else { rrow = rb - 20 - FTH; rcol = FTW; }
    }
    const int ring_a = (rrow + 1) * IN_RS + rcol + (lane & 3) + 3;           // A operand: the lane's own pixel (rrow, rcol + (lane & 3)) at tap (0, 0)
    const int ring_h = (rrow + 1) * H_RS + rcol + 4;                          // the block's home in the h tile

    float wr0[NCH][NWR], wr1[NCH][NWR];                    // resident B registers: lane l = quad 4 v + (l >> 4), element l & 3
    float bias0[NG], bias1[NG], full0[NG], full1[NG];
    auto fetch_affine = [&](int b) {                       // tid < 64: (1 + scale0 | shift0 | 1 + scale1 | shift1)[c]
        const int c = tid & 15, which = (tid >> 4) & 3;
        float v = 0.f;
        if (tid < 64 && c < C) {
            const float* p = which == 0 ? d.scale0 : which == 1 ? d.shift0 : which == 2 ? d.scale1 : d.shift1;
            v = p[b * C + c];
            if ((which & 1) == 0) v += 1.0f;
        }
        return v;
    };
    auto load_weights = [&]() {                            // after s_w and s_aff are visible; ends with the per-lane constants
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            const float f0 = s_aff[ci], f1 = s_aff[32 + ci];
#pragma unroll
            for (int v = 0; v < NWR; ++v) {
                const int e = (ci * QPAD + 4 * v + (lane >> 4)) * 4 + lj;
                wr0[ci][v] = s_w[e] * f0;
                wr1[ci][v] = s_w[S_W + e] * f1;
            }
        }
        for (int e = tid; e < 2 * 9 * 16; e += 256) {      // the folded shift terms per (conv, tap, output channel)
            const int cv = e / 144, te = e - cv * 144, tap = te >> 4, co = te & 15;
            float a_ = 0.f;
            if (co < 4 * NG) {
#pragma unroll
                for (int ci = 0; ci < NCH; ++ci) a_ = fmaf(s_aff[cv * 32 + 16 + ci], s_w[cv * S_W + (ci * QPAD + tap * NG + (co >> 2)) * 4 + (co & 3)], a_);
            }
            s_beta[e] = a_;
        }
        lds_barrier();
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) { t0 += s_beta[tap * 16 + 4 * n + lj]; t1 += s_beta[144 + tap * 16 + 4 * n + lj]; }
            full0[n] = t0; full1[n] = t1;
        }
    };

    // ---- prologue: the first tile's DMA, the raw weights of both convolutions, the sample's affine parameters: one exposed memory latency
    int cur_b = it.b;
    {
        const float av = fetch_affine(it.b);
        issue(it, 0);
        constexpr int NWV = (2 * S_W + 255) / 256;
        float wv[NWV];
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e2 = tid + u * 256;
            const int cv = e2 >= S_W ? 1 : 0, e = e2 - cv * S_W;
            const int j = e & 3, cq = e >> 2;
            const int ci = cq / QPAD, q = cq - ci * QPAD;
            const int tap = q / NG, n = q - tap * NG;
            const int co = 4 * n + j;
            float v = 0.f;
            if (e2 < 2 * S_W && q < NQD && co < C && ci < C) v = (cv ? d.w1 : d.w0)[(co * C + ci) * 9 + tap];
            wv[u] = v;
        }
        if (tid < 64) s_aff[tid] = av;
#pragma unroll
        for (int u = 0; u < NWV; ++u)
            if (tid + u * 256 < 2 * S_W) s_w[tid + u * 256] = wv[u];
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const bool ok = 4 * n + lj < C;
            bias0[n] = (ok && d.b0) ? d.b0[4 * n + lj] : 0.f;
            bias1[n] = (ok && d.b1) ? d.b1[4 * n + lj] : 0.f;
        }
        wait_vm<0>();
        lds_barrier();
        load_weights();
    }

    int buf = 0;
    int trace_tile = 0; (void)trace_tile;
    for (; itx < r1; itx += nlb, ++trace_tile) {
        TTR(0);
        const bool has_next = itx + nlb < r1;
        LItem nxt = it;
        if (has_next) { nxt = advance(it); issue(nxt, buf ^ 1); }
        const int ty0 = it.ty * FTH, tx0 = it.tx * FTW;
        const float* sin_b = s_in + buf * S_IN;
        // every pixel this tile computes (ring included) has all nine taps inside the image, for both convolutions
        const bool border = ty0 < 2 || ty0 + FTH + 2 > H || tx0 < 4 || tx0 + FTW + 2 > W;
        const unsigned ob = (unsigned)((((it.b * C) * H + ty0 + 4 * wave) * W + tx0) * 4);
        f32x4 acc[NG];
        TTR(1);

        // ---- phase 0: h = gelu(conv0) on the tile (wave-ops 0, 1) and its one-pixel ring (wave-op 2).  One code body for the three (the
        // K loop is 324 MFMAs: unrolling the wave-ops would put ~55 KB of instructions in the tile loop)
#pragma nounroll
        for (int op = 0; op < 3; ++op) {
            const bool ring = op == 2;
            kloop<IN_PLANE, IN_RS>(acc, sin_b + (ring ? ring_a : (4 * wave + 2 * op + (lane >> 5) + 1) * IN_RS + (lane & 31) + 3), wr0);
            TTR(2 + 2 * op);
            const int row = ring ? rrow : 4 * wave + 2 * op + (lb >> 3);      // tile row / column of this lane's four-pixel block
            const int colb = ring ? rcol : 4 * (lb & 7);
            const int gy = ty0 + row, gx0 = tx0 + colb;
            const bool in_img = (unsigned)gy < (unsigned)H && (unsigned)gx0 < (unsigned)W;
            const bool lane_on = ring ? ring_lane : true;
            float* hdst = s_h + (row + 1) * H_RS + colb + 4;
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const f32x4 v = acc[n] + bias0[n] + shift_term(s_beta, 4 * n + lj, full0[n], border, gy, gx0, H, W);
                f32x4 hv, gv;
                gelu_pair4_f(v, &hv, &gv);
                const bool ok = in_img && (4 * n + lj < C);
                if (!ok) hv = f32x4{0.f, 0.f, 0.f, 0.f};   // outside the image: conv1's zero padding
                if (lane_on) *reinterpret_cast<f32x4*>(hdst + (4 * n + lj) * H_PLANE) = hv;
                if constexpr (TRAIN) {                     // (always issued, dropped through the offset: the DMA wait below counts them)
                    const unsigned so = ob + (unsigned)(2 * (op & 1) * W * 4) + (unsigned)n * nstep;
                    const unsigned vo = (ok && !ring) ? ovoff : OOB;
                    bstore(rh, vo, so, hv);
                    bstore(rg, vo, so, gv);
                }
            }
            TTR(3 + 2 * op);
        }
        lds_barrier();                                     // the h tile is complete
        TTR(8);

        // ---- phase 1: out = conv1(h) + b1 + shift term + x0
#pragma nounroll
        for (int op = 0; op < 2; ++op) {
            const int row = 4 * wave + 2 * op + (lb >> 3);
            kloop<H_PLANE, H_RS>(acc, s_h + (4 * wave + 2 * op + (lane >> 5)) * H_RS + (lane & 31) + 3, wr1);
            TTR(9 + 2 * op);
            const int gy = ty0 + row, gx0 = tx0 + 4 * (lb & 7);
            const bool px_ok = gy < H && gx0 < W;
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const f32x4 x0v = *reinterpret_cast<const f32x4*>(sin_b + (4 * n + lj) * IN_PLANE + (row + 2) * IN_RS + 4 * (lb & 7) + 4);
                const f32x4 v = acc[n] + bias1[n] + shift_term(s_beta + 144, 4 * n + lj, full1[n], border, gy, gx0, H, W) + x0v;
                const bool ok = px_ok && (4 * n + lj < C);
                bstore(ro, ok ? ovoff : OOB, ob + (unsigned)(2 * op * W * 4) + (unsigned)n * nstep, v);
            }
            TTR(10 + 2 * op);
        }
        if (has_next) {
            // this tile's stores are younger than the next tile's DMA: all but them are complete
            wait_vm<(TRAIN ? 24 : 6)>();
            lds_barrier();                                 // next input tile landed in every wave; every wave is done with this tile's buffers
            if (nxt.b != cur_b) {                          // B > 1: the next sample's affine parameters (rare)
                const float av = fetch_affine(nxt.b);
                if (tid < 64) s_aff[tid] = av;
                lds_barrier();
                load_weights();
                cur_b = nxt.b;
            }
            buf ^= 1;
        }
        TTR(13);
        it = nxt;
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

#ifdef BNERV_TRACE_TAT
extern "C" int bnerv_debug_trace_tat_read(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_tat), sizeof(g_trace_tat)); }
#endif
// 1: not this kernel's block (the caller issues the two convolution calls); BNERV_OK; negative BNERV_E_*.
extern "C" int bnerv_tat_block_fwd(void* stream, const bnerv_tat_desc* dp) {
    BNERV_REQUIRE(dp != nullptr, "bnerv_tat_block_fwd: null descriptor");
    const bnerv_tat_desc& d = *dp;
    BNERV_REQUIRE(d.x0 && d.w0 && d.w1 && d.scale0 && d.shift0 && d.scale1 && d.shift1 && d.out, "bnerv_tat_block_fwd: null tensor");
    BNERV_REQUIRE((d.h == nullptr) == (d.gp == nullptr), "bnerv_tat_block_fwd: h and gp are written together (train) or not at all (decode)");
    BNERV_REQUIRE(d.B > 0 && d.C > 0 && d.H > 0 && d.W > 0, "bnerv_tat_block_fwd: bad shape %d x %d x %d x %d", d.B, d.C, d.H, d.W);
    // OPT-IN (BNERV_TATF=1; read per call: tests and tools flip it).  Measured on MI355X (profiles/r04_tatf_trace.md): 118-121 us per 720p
    // block against 89 us for the two launches it replaces -- with one wave per SIMD nothing overlaps the f32 MFMA (K loops 5 x 3.85 k
    // cycles, epilogues 6.9 k, LDS-DMA issue 1.8 k per tile), the halo recompute is a fifth wave-op per four, and 1800 tiles on 256 blocks
    // are 8 rounds for 7.03.  Kept as a tested alternative form, not dispatched by default.
    { const char* e = getenv("BNERV_TATF"); if (!(e && e[0] == '1')) return 1; }
    const size_t bytes = (size_t)d.B * d.C * d.H * d.W * 4;
    if (!(d.C > 8 && d.C <= NCH && d.W % 4 == 0 && d.H >= 3 && d.W >= 4 && bytes + (size_t)(2 * d.W + 4) * 4 < LEAN_MAX_BYTES &&
          aligned16(d.x0) && aligned16(d.out) && aligned16(d.h) && aligned16(d.gp)))
        return 1;
    TatArgs ka;
    ka.d = d;
    ka.tiles_x = cdiv(d.W, FTW);
    ka.tiles_y = cdiv(d.H, FTH);
    ka.total_items = d.B * ka.tiles_x * ka.tiles_y;
    int min_items = 64;
    { const char* e = getenv("BNERV_TATF_MIN_TILES"); if (e) min_items = atoi(e); }
    if (ka.total_items < min_items) return 1;              // a handful of tiles: the two small-image launches fill the chip better
    ka.magic_tiles = div_magic(ka.tiles_x * ka.tiles_y);
    ka.magic_tiles_x = div_magic(ka.tiles_x);
    const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int grid = 256;                                        // one block per compute unit (the LDS image admits no second one)
    if (grid > ka.total_items) grid = ka.total_items;
    static bool attr_done[2] = {false, false};
    if (d.h) {
        if (!attr_done[1]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tat_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[1] = true; }
        hipLaunchKernelGGL((tat_fused_kernel<true>), dim3(grid), dim3(256), lds, st, ka);
    } else {
        if (!attr_done[0]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tat_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[0] = true; }
        hipLaunchKernelGGL((tat_fused_kernel<false>), dim3(grid), dim3(256), lds, st, ka);
    }
    BNERV_LAUNCH_CHECK("tat_fused");
    return BNERV_OK;
}
