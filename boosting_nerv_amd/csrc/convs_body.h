// convs_body.h -- the low-resolution 3x3 convolution of convs.hip as a device function, so that a paired launch (wgrad.hip) can run it
// next to the weight gradient that reads the same incoming gradient.  See convs.hip for the design.
#pragma once
#include "common.h"
#include <type_traits>
#include "sidejob.h"
#include "conv_common.h"

#ifndef BNERV_ABLS
#define BNERV_ABLS 0   // debug ablations (never shipped; tools/ksmall.py): bit 0 no weight loads, 1 no input loads, 2 no K loop, 3 no epilogue math
#endif

namespace bnerv_convs {
using namespace bnerv_conv;

constexpr int STH = 4, STW = 16;           // tile: 4 rows x 16 px; wave w owns row w
constexpr int SROWS = STH + 2;             // haloed rows
constexpr int SXOFF = 4;                   // left margin (aligned float4 segments)
constexpr int SRS = STW + 2 * SXOFF;       // 24 floats per LDS row
constexpr int SSEGS = SRS / 4;             // 6 float4 per row
constexpr int SPLANE = SROWS * SRS + 4;    // 148: == 20 (mod 32) -> the four k-lanes of an A fragment read spread over the banks
constexpr int SCOL0 = SXOFF - 1;
constexpr int SMAXC = 64;                  // input channels staged at once (32 for the plain / affine modes, 64 for the unshuffled gradient of an up-conv)

struct SArgs {
    bnerv_conv_desc d;
    int tiles_x, tiles_y;
};

template <int EP>
__device__ __forceinline__ float s_ep(float v, float bias, float* o2) {
    if constexpr (EP == BNERV_EP_BIAS) return v + bias;
    if constexpr (EP == BNERV_EP_BIAS_SIN) { float sv, cv; sincos_f(v + bias, &sv, &cv); *o2 = cv; return sv; }
    if constexpr (EP == BNERV_EP_BIAS_GELU) { float h; gelu_pair_f(v + bias, &h, o2); return h; }
    return v;
}
template <int EP>
__device__ __forceinline__ f32x4 s_ep4(f32x4 v, float bias, f32x4* o2) {          // four elements at once: the packed forms of common.h
    if constexpr (EP == BNERV_EP_BIAS) return v + bias;
    if constexpr (EP == BNERV_EP_BIAS_SIN) { f32x4 sv; sincos4_f(v + bias, &sv, o2); return sv; }
    if constexpr (EP == BNERV_EP_BIAS_GELU) { f32x4 h; gelu_pair4_f(v + bias, &h, o2); return h; }
    return v;
}

// NQ = ceil(Cin / 4) rounded to 4 or 8 (16 or 32 staged channels)
template <int IN, int EP, int NQ>
__device__ __forceinline__ void conv_small_body(const SArgs& sa, const int tile, const int cog, const int b) {
    // (tile, cog, b: the block's pixel tile, group of 16 output channels and sample -- blockIdx of conv_small_kernel, or what a paired
    //  launch (wgrad.hip) derives from its own block index)
    constexpr int NCH = NQ * 4;
    constexpr int NSLOT = NCH * SROWS * SSEGS;                 // float4 slots of the input tile
    constexpr int NPRE = (NSLOT + 255) / 256;
    constexpr bool AFF = (IN == BNERV_IN_AFFINE);
    constexpr bool RED = (EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    const bnerv_conv_desc& d = sa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                        // [NCH][SPLANE]
    float* s_w = smem + NCH * SPLANE;                          // raw weight slice: forward [16 co][Cin * 9], transposed [Cin][16 co][9]
    float* s_red = s_w + 16 * NCH * 9;                         // [4 waves][2][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int co_base = cog * 16;
    const int ty0 = (tile / sa.tiles_x) * STH, tx0 = (tile % sa.tiles_x) * STW;

    // ---- input tile: raw buffer loads, all in flight, affine applied on the way into LDS (zero padding AFTER the affine)
    const unsigned shift = (unsigned)((W + SXOFF) * 4);
    const unsigned in_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, in_bytes);
    const unsigned sb = (unsigned)((((b * Cin) * H + ty0) * W + tx0) * 4);
    constexpr bool UNS = (IN == BNERV_IN_UNSHUFFLE);           // x is stored pixel-shuffled (x2): [B][Cin / 4][2H][2W]
    constexpr int NSLOT_U = (NCH / 2) * SROWS * SSEGS;         // unshuffle: a slot = channel PAIR (cf, i, j = 0 / 1) x row x segment, two float4 of the source row
    constexpr int NPRE_U = (NSLOT_U + 255) / 256;
    f32x4 ra[UNS ? NPRE_U : NPRE], rb[UNS ? NPRE_U : 1];
    float sc[UNS ? 1 : NPRE], sh[UNS ? 1 : NPRE];
    if constexpr (UNS) {
        const unsigned ub = (unsigned)((size_t)d.B * Cin * H * W * 4);
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(d.x, 0, ub);
        const unsigned sbu = (unsigned)((size_t)b * Cin * H * W * 4);
#pragma unroll
        for (int k = 0; k < NPRE_U; ++k) {
            const int sidx = tid + k * 256;
            const int p = sidx / (SROWS * SSEGS), rem = sidx - p * (SROWS * SSEGS), r = rem / SSEGS, sg = rem - r * SSEGS;
            const int gy = ty0 + r - 1, gx = tx0 + 4 * sg - SXOFF;
            const bool ok = sidx < NSLOT_U && 2 * p < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const unsigned off = ok ? (unsigned)((((p >> 1) * 2 * H + 2 * gy + (p & 1)) * (2 * W) + 2 * gx) * 4) : OOB;
            ra[k] = bload(ru, off, sbu);
            rb[k] = bload(ru, ok ? off + 16u : OOB, sbu);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int sidx = tid + k * 256;
            const int c = sidx / (SROWS * SSEGS), rem = sidx - c * (SROWS * SSEGS), r = rem / SSEGS, sg = rem - r * SSEGS;
            const int gy = ty0 + r - 1, gx = tx0 + 4 * sg - SXOFF;
            const bool ok = sidx < NSLOT && c < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            ra[k] = bload(rx, (ok && !((BNERV_ABLS & 2) && d.B > 0)) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB, sb);
            sc[k] = 1.f; sh[k] = 0.f;
            if constexpr (AFF) { if (ok) { sc[k] = 1.0f + d.scale[b * Cin + c]; sh[k] = d.shift[b * Cin + c]; } else sc[k] = 0.f; }
        }
    }
    // ---- weight slice of this block's 16 output channels, as it lies in memory (coalesced), zero beyond Cout / Cin.  Every load is
    //      issued before the first store (a loop of load -> store pairs is one L2 round trip per iteration: 17 of them were most of
    //      the launch).
    {
        constexpr int NWL = (16 * NCH * 9 + 255) / 256;        // dwords per thread (18 at 32 channels)
        const int ncopy = 16 * Cin * 9;
        float wv[NWL];
        if (!d.transposed) {
            // W(co, ci, t) = w[co][ci][t]: rows co_base .. co_base + 15 are contiguous
            const float* src = d.w + (size_t)co_base * Cin * 9;
            const int nvalid = min(16, Cout - co_base) * Cin * 9;
#pragma unroll
            for (int u = 0; u < NWL; ++u) { const int i = tid + u * 256; wv[u] = (i < nvalid && !((BNERV_ABLS & 1) && d.B > 0)) ? src[i] : 0.f; }
        } else {
            // W(co, ci, t) = w[ci][co][8 - t] (w is [wCo = Cin][wCi = Cout][9]): per ci a segment of 16 x 9 floats; kept as [ci][16][9]
#pragma unroll
            for (int u = 0; u < NWL; ++u) {
                const int i = tid + u * 256;
                const int ci = i / 144, rem = i - ci * 144, col = rem / 9;
                wv[u] = (i < ncopy && co_base + col < Cout && !((BNERV_ABLS & 1) && d.B > 0)) ? d.w[((size_t)ci * d.wCi + co_base) * 9 + rem] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < NWL; ++u) { const int i = tid + u * 256; if (i < ncopy) s_w[i] = wv[u]; }
        // the last channel quad may reach up to 3 channels past Cin: zero what it reads beyond the slice (forward: the 36 floats after the
        // last row; transposed: the [ci][16][9] segments of the missing channels), so that 0 (the A side) meets a finite number
        if (Cin & 3) {
            const int tail = d.transposed ? (4 - (Cin & 3)) * 144 : 36;
            if (tid < tail && ncopy + tid < 16 * NCH * 9) s_w[ncopy + tid] = 0.f;
            if (tid + 256 < tail && ncopy + tid + 256 < 16 * NCH * 9) s_w[ncopy + tid + 256] = 0.f;
        }
    }
    if constexpr (UNS) {
#pragma unroll
        for (int k = 0; k < NPRE_U; ++k) {
            const int sidx = tid + k * 256;
            if (sidx < NSLOT_U) {
                const int p = sidx / (SROWS * SSEGS), rem = sidx - p * (SROWS * SSEGS), r = rem / SSEGS, sg = rem - r * SSEGS;
                const f32x4 va = ra[k], vb = rb[k];
                float* dst = s_in + (2 * p) * SPLANE + r * SRS + 4 * sg;
                *reinterpret_cast<f32x4*>(dst) = f32x4{va[0], va[2], vb[0], vb[2]};              // j = 0: even source columns
                *reinterpret_cast<f32x4*>(dst + SPLANE) = f32x4{va[1], va[3], vb[1], vb[3]};     // j = 1: odd source columns
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int sidx = tid + k * 256;
            if (sidx < NSLOT) {
                const int c = sidx / (SROWS * SSEGS), rem = sidx - c * (SROWS * SSEGS), r = rem / SSEGS, sg = rem - r * SSEGS;
                f32x4 v = ra[k];
                if constexpr (AFF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[k] + sh[k];
                }
                *reinterpret_cast<f32x4*>(s_in + c * SPLANE + r * SRS + 4 * sg) = v;
            }
        }
    }
    __syncthreads();

    // ---- K loop: A = pixel li of the wave's row, channel 4q + kq ; B = W(co_base + li, 4q + kq, tap) from the raw slice.
    // One channel quad per iteration of a REAL loop over the quads that exist (pointer bumps; every LDS address inside the body is base +
    // immediate), two quads per trip so that quad q + 1's eighteen operands are in flight under quad q's nine MFMAs in two fixed register
    // sets -- round 5's form was a chain of per-quad uniform branches with a select per operand, ~80 cycles per MFMA at one wave per SIMD
    // (tools/ksmall.py ablation: 2.6 of a 60-block launch's 7.3 us).  Channels beyond Cin in the last quad: their A planes are zero and
    // their weights are finite (the next row of the slice, or the zero-filled tail above), so the products vanish without a select.
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = s_in + kq * SPLANE + wave * SRS + li + SCOL0;
    const float* bp = d.transposed ? s_w + (kq * 16 + li) * 9 : s_w + li * Cin * 9 + kq * 9;
    const int b_qstep = d.transposed ? 4 * 144 : 36;           // + 4 input channels
    const int nq = (BNERV_ABLS & 4) && d.B > 0 ? 0 : min(NQ, (Cin + 3) >> 2);
    auto ld = [&](const float* a, const float* b, float (&av)[9], float (&bv)[9], auto tr) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            av[t] = a[(t / 3) * SRS + (t % 3)];
            bv[t] = b[decltype(tr)::value ? 8 - t : t];
        }
    };
    auto kloop = [&](auto tr) __attribute__((always_inline)) {
        float a0[9], b0[9], a1[9], b1[9];
        if (nq > 0) ld(ap, bp, a0, b0, tr);
        int q = 0;
#pragma unroll 1
        for (; q + 1 < nq; q += 2) {
            ld(ap + 4 * SPLANE, bp + b_qstep, a1, b1, tr);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b0[t], acc, 0, 0, 0);
            ap += 8 * SPLANE; bp += 2 * b_qstep;
            if (q + 2 < nq) ld(ap, bp, a0, b0, tr);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], b1[t], acc, 0, 0, 0);
        }
        if (q < nq) {
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b0[t], acc, 0, 0, 0);
        }
    };
    if (d.transposed) kloop(std::true_type{}); else kloop(std::false_type{});

    // ---- epilogue from the accumulator: lane (li, kq) = output channel co_base + li, pixels (row wave, columns 4 kq .. 4 kq + 3)
    const int co = co_base + li, gy = ty0 + wave, gx = tx0 + 4 * kq;
    const bool ok = co < Cout && gy < H && gx < W;
    const float bias = (EP != BNERV_EP_PLAIN && !RED && d.bias && co < Cout) ? d.bias[co] : 0.f;
    const size_t o = (((size_t)b * Cout + co) * H + gy) * (size_t)W + gx;
    if constexpr (RED) {
        float ps = 0.f, pt = 0.f;
        if (ok) {
            const float scl = 1.0f + d.scale[b * Cout + co];
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(d.aux0 + o), a1 = *reinterpret_cast<const f32x4*>(d.aux1 + o);
            f32x4 a2 = {1.f, 1.f, 1.f, 1.f}, r;
            if constexpr (EP == BNERV_EP_DSIN) { if (d.aux2) a2 = *reinterpret_cast<const f32x4*>(d.aux2 + o); }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[e];
                if constexpr (EP == BNERV_EP_DGELU_SAVED) { r[e] = v * scl * a0[e]; ps = fmaf(v, a1[e], ps); }
                else { r[e] = (a1[e] + v * scl) * a2[e]; ps = fmaf(v, a0[e], ps); }
                pt += v;
            }
            *reinterpret_cast<f32x4*>(d.out + o) = r;
        }
        ps += __shfl_xor(ps, 16, 64); pt += __shfl_xor(pt, 16, 64);
        ps += __shfl_xor(ps, 32, 64); pt += __shfl_xor(pt, 32, 64);
        if (lane < 16) { s_red[(wave * 2 + 0) * 16 + lane] = ps; s_red[(wave * 2 + 1) * 16 + lane] = pt; }
        __syncthreads();
        if (tid < 32) {
            const int qq = tid >> 4, c = tid & 15;
            const float s = ((s_red[(0 * 2 + qq) * 16 + c] + s_red[(1 * 2 + qq) * 16 + c]) + s_red[(2 * 2 + qq) * 16 + c]) + s_red[(3 * 2 + qq) * 16 + c];
            if (co_base + c < Cout) d.partial[(((size_t)tile * d.B + b) * 2 + qq) * Cout + co_base + c] = s;
        }
    } else if (d.out_s == 1) {
        if (ok) {
            f32x4 r, r2;
            if constexpr (EP == BNERV_EP_BIAS_RES) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(d.aux0 + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = acc[e] + bias + a0[e];
            } else {
                r2 = f32x4{0.f, 0.f, 0.f, 0.f};
                r = s_ep4<EP>(acc, bias, &r2);
            }
            *reinterpret_cast<f32x4*>(d.out + o) = r;
            if constexpr (EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) { if (d.out2) *reinterpret_cast<f32x4*>(d.out2 + o) = r2; }
        }
    } else if (d.out_s == 2 && (Cout & 3) == 0) {
        // PixelShuffle(2): lanes li = 4 c + 2 i + j.  Lanes j = 0 / 1 (neighbours) hold the even / odd output columns of the same row:
        // they swap halves so that each stores 4 CONSECUTIVE output pixels -- lane j = 0 the columns 2 gx .. 2 gx + 3, lane j = 1 the next four.
        f32x4 c2v = {0.f, 0.f, 0.f, 0.f};
        const f32x4 rv = s_ep4<EP>(acc, bias, &c2v);
        const int j = li & 1;
        f32x4 o1, o2;
        {
            float pr[4], pc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { pr[e] = __shfl_xor(rv[e], 1, 64); pc[e] = __shfl_xor(c2v[e], 1, 64); }
            o1 = j ? f32x4{pr[2], rv[2], pr[3], rv[3]} : f32x4{rv[0], pr[0], rv[1], pr[1]};
            o2 = j ? f32x4{pc[2], c2v[2], pc[3], c2v[3]} : f32x4{c2v[0], pc[0], c2v[1], pc[1]};
        }
        if (ok) {
            const int c = co >> 2, i = (co >> 1) & 1;
            const int Cf = Cout >> 2, HF = 2 * H, WF = 2 * W;
            const size_t oo = (((size_t)b * Cf + c) * HF + (size_t)(2 * gy + i)) * (size_t)WF + (size_t)(2 * gx + 4 * j);
            *reinterpret_cast<f32x4*>(d.out + oo) = o1;
            if constexpr (EP == BNERV_EP_BIAS_SIN) { if (d.out2) *reinterpret_cast<f32x4*>(d.out2 + oo) = o2; }
        }
    } else {
        // PixelShuffle(s): conv-space channel co -> (c, i, j); pixel (gy, gx + e) -> (gy * s + i, (gx + e) * s + j)
        if (ok) {
            const int s = d.out_s, s2 = s * s, c = co / s2, rem = co - c * s2, i = rem / s, j = rem - i * s;
            const int Cf = Cout / s2, HF = H * s, WF = W * s;
            const size_t rowo = (((size_t)b * Cf + c) * HF + (size_t)(gy * s + i)) * (size_t)WF;
            f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 r = s_ep4<EP>(acc, bias, &c2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (gx + e < W) {
                    const size_t oo = rowo + (size_t)((gx + e) * s + j);
                    d.out[oo] = r[e];
                    if constexpr (EP == BNERV_EP_BIAS_SIN) { if (d.out2) d.out2[oo] = c2[e]; }
                }
            }
        }
    }
}

template <int NQ>
constexpr size_t convs_lds_bytes() { return ((size_t)NQ * 4 * SPLANE + (size_t)16 * NQ * 4 * 9 + 128) * sizeof(float); }

}  // namespace bnerv_convs
