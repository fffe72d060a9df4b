// conv4_body.h -- device body of the 4x4x1-MFMA convolution kernel (conv4.hip), shared with the paired conv + weight-gradient launch
// (wgrad.hip: bnerv_conv_wgrad_pair).  See conv4.hip for the design notes.
#pragma once
#include "common.h"
#include "sidejob.h"
#include "conv_common.h"

namespace bnerv_q4 {
using bnerv_conv::TH; using bnerv_conv::TW; using bnerv_conv::Geo; using bnerv_conv::KArgs; using bnerv_conv::LItem; using bnerv_conv::OOB;
using bnerv_conv::i32x4; using bnerv_conv::make_rsrc; using bnerv_conv::bload; using bnerv_conv::bstore; using bnerv_conv::fast_div;
using bnerv_conv::div_magic; using bnerv_conv::xform1; using bnerv_conv::LEAN_MAX_BYTES;


#ifndef BNERV_ABL4
#define BNERV_ABL4 0   // debug ablations (never shipped): 1 no MFMA, 2 no global loads / stores, 3 no epilogue math
#endif
constexpr int Q4_NCH = 12;                 // staged input channels
constexpr int Q4_NG = 3;                   // output-channel quads
constexpr int Q4_NQD = 9 * Q4_NG;          // weight quads per input channel (tap-major: q = tap * NG + n)
constexpr int Q4_NWR = (Q4_NQD + 3) / 4;   // weight registers per input channel (4 quads each)
constexpr int Q4_QPAD = Q4_NWR * 4;        // quads per channel in the LDS image (padding quads are zero)

#ifdef BNERV_TRACE
__device__ unsigned long long g_trace4[1024 * 4 * 6 * 8];
#define TRACE(slot) do { if (lane == 0 && vb < 1024 && trace_iter < 6) g_trace4[((vb * 4 + wave) * 6 + trace_iter) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRACE(slot) do {} while (0)
#endif

template <int G>
__device__ __forceinline__ f32x4 mfma_q(float a, float w, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c, 0, 0, 4 + G);
}
__device__ __forceinline__ f32x4 mfma_qsel(int g, float a, float w, f32x4 c) {          // g is a constant after unrolling
    switch (g) {
        case 0: return mfma_q<0>(a, w, c);
        case 1: return mfma_q<1>(a, w, c);
        case 2: return mfma_q<2>(a, w, c);
        default: return mfma_q<3>(a, w, c);
    }
}

// LDS-DMA of one 16-byte slot per lane: LDS address = m0 + 16 * lane.  Issued through inline asm on purpose: the compiler would
// order every later ds_read behind a builtin LDS-DMA with s_waitcnt vmcnt(0) (it cannot tell the two input buffers apart),
// which would serialise the next tile's loads with this tile's matrix phase.  The waits are placed by hand (q4_wait_dma).
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void q4_wait_dma() {            // all but the N youngest vector-memory operations of this wave are complete
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int IN, int EP>
__device__ __forceinline__ void conv_q4_body(const KArgs& ka, const SidePack& side, const int vb, const int vgrid) {
    using G = Geo<3>;
    constexpr int NSLOT = Q4_NCH * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    constexpr int S_IN = NPRE * 256 * 4;                   // floats per input buffer: every slot (idle ones included) owns 16 bytes
    constexpr int S_W = Q4_NCH * Q4_QPAD * 4;
    constexpr bool AFF = (IN == BNERV_IN_AFFINE);
    constexpr bool RED = (EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    constexpr bool TWO_OUT = (EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU);
    static_assert(IN == BNERV_IN_PLAIN || IN == BNERV_IN_AFFINE, "prologues of the 12-channel layers");
    static_assert(G::PLANE == G::PLANE_RAW && S_IN >= Q4_NCH * G::PLANE, "slot offset = 16 * slot index");
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                    // [2][S_IN] input tiles, filled by LDS-DMA
    float* s_w = smem + 2 * S_IN;                          // [ci][quad][4] compact weight quads
    float* s_red = s_w + S_W;                              // [2 sets][4 waves][2][16] per-channel partial sums (DGELU_SAVED / DSIN).  Two sets, by tile parity:
                                                           // wave 0 reads tile i's sums at the TOP of tile i + 1 while the other waves write tile i + 1's at its
                                                           // end with no barrier in between -- with one set a starved wave 0 could read sums of the wrong tile
                                                           // (seen as a last-bit difference of a TAT gradient once in ~10..100 steps); a set is rewritten only two
                                                           // tiles later, behind the barrier that closes the tile in which it was read
    float* s_aff = s_red + 256;                            // [2][16] affine prologue parameters of the current sample
    float* s_beta = s_aff + 32;                            // [9 taps][16] sum_ci shift[ci] * W(co, ci, tap)  (affine prologue, folded)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 3, lb = lane >> 2;               // output channel within a quad, pixel quad (row lb >> 3, columns 4 (lb & 7) ..)
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;
    { const int trace_iter = 5; (void)trace_iter; TRACE(0); }

    // this block's item range: XCD x owns a contiguous slice of the item list; its blocks take it round-robin
    const int xcd = vb & 7, lbk = vb >> 3;                 // (vb of vgrid: this launch's block index / size, or the conv part of a pair launch)
    const int nlb = (vgrid - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lbk;
    if (itx >= r1) { side_run_hosted(side, smem, vb, vgrid); return; }
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

    // ---- per-slot constants (slot = (channel, halo row, 4-px segment); thread t owns slots t, t+256, ...; its LDS home is byte 16 * slot)
    auto slot_geom = [&](int k, int& c, int& r, int& sg) {
        const int sidx = tid + k * 256;
        c = sidx / (G::ROWS * G::SEGS);
        const int rem = sidx - c * (G::ROWS * G::SEGS);
        r = rem / G::SEGS;
        sg = rem - r * G::SEGS;
    };
    auto slot_inside = [&](int k, int ty0, int tx0) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        const int gy = ty0 + r - G::PAD, gx = tx0 + 4 * sg - G::XOFF;
        return (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    };
    unsigned voff[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        voff[k] = (tid + k * 256 < NSLOT && c < Cin) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;       // out of range: the DMA writes zeros
    }
    const unsigned shift = (unsigned)((G::PAD * W + G::XOFF) * 4);          // the x view starts PAD rows + XOFF columns early: offsets >= 0
    const unsigned in_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    i32x4 rx;                                              // the raw buffer descriptor of make_rsrc(d.x - shift, in_bytes), as the four dwords the DMA asm takes
    {
        const uintptr_t base = reinterpret_cast<uintptr_t>(d.x) - shift;
        rx[0] = (int)(unsigned)(base & 0xffffffffu);
        rx[1] = (int)(unsigned)((base >> 32) & 0xffffu);
        rx[2] = (int)in_bytes;
        rx[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ro2 = make_rsrc((TWO_OUT && d.out2) ? d.out2 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(d.aux0 ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);
    const unsigned lds_in = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_in + (unsigned)wave * 1024u;      // this wave's 64 slots of DMA pass 0

    // epilogue lane constants: lane (lb, lj) owns output channels 4 n + lj, pixels (row lb >> 3, columns 4 (lb & 7) .. + 3) of the wave's two rows
    const unsigned ovoff = (unsigned)(((lj * H + (lb >> 3)) * W + 4 * (lb & 7)) * 4);
    const unsigned nstep = (unsigned)(4 * H * W * 4);      // + 4 output channels
    float bias_l[Q4_NG];                                   // bias (+ the folded shift term of an interior pixel)
#pragma unroll
    for (int n = 0; n < Q4_NG; ++n) bias_l[n] = (EP != BNERV_EP_PLAIN && !RED && d.bias && 4 * n + lj < Cout) ? d.bias[4 * n + lj] : 0.f;
    float scl[Q4_NG];                                      // 1 + scale[b][co] of the DGELU_SAVED / DSIN epilogues
#pragma unroll
    for (int n = 0; n < Q4_NG; ++n) scl[n] = 0.f;

    // tile (a) -> input buffer `buf`: NPRE DMA passes per wave, zeros outside the image / beyond Cin
    auto issue = [&](const LItem& a, int buf) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const unsigned sb = (unsigned)((((a.b * Cin) * H + ty0) * W + tx0) * 4);
        const unsigned lbase = lds_in + (unsigned)buf * (unsigned)(S_IN * 4);
        const bool interior = ty0 >= G::PAD && ty0 + TH + G::PAD <= H && tx0 >= G::XOFF && tx0 + TW + G::XOFF <= W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rx, (BNERV_ABL4 == 2 && d.B > 0) ? OOB : voff[k], sb, lbase + (unsigned)k * 4096u);
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rx, slot_inside(k, ty0, tx0) ? voff[k] : OOB, sb, lbase + (unsigned)k * 4096u);
        }
    };
    auto flush_partials = [&](const LItem& a, const int set) {            // wave 0: sum the 4 waves' channel sums of tile `a`, fixed order
        if (wave == 0 && lane < 32) {
            const int q = lane >> 4, c = lane & 15;
            const float* sr = s_red + set * 128;
            const float s = ((sr[(0 * 2 + q) * 16 + c] + sr[(1 * 2 + q) * 16 + c]) + sr[(2 * 2 + q) * 16 + c]) + sr[(3 * 2 + q) * 16 + c];
            const size_t row = (size_t)(a.ty * tiles_x + a.tx) * d.B + a.b;            // [tiles][B][2][Cout]
            if (c < Cout) d.partial[(row * 2 + q) * Cout + c] = s;
        }
    };

    // resident B registers: wr[ci][v], lane l = quad 4 v + (l >> 4), element l & 3 (replicated over the group's four blocks).
    // The TAT affine a = x (1 + s_ci) + t_ci in front of the convolution is folded: conv_w(a) = conv_{w (1 + s_ci)}(x) + sum of
    // t_ci w over the taps that fall INSIDE the image (zero padding applies after the affine), so the staged tile is the raw
    // tensor, the weights are scaled once per sample and the shift term joins the bias (interior pixels: all 9 taps; pixels on
    // the image border: the taps that exist -- s_beta keeps the per-tap sums).
    float wr[Q4_NCH][Q4_NWR];
    float beta_l[Q4_NG];
#pragma unroll
    for (int n = 0; n < Q4_NG; ++n) beta_l[n] = 0.f;
    auto fetch_affine = [&](int b) {                       // tid < 32 only: the value this lane contributes
        const int c = tid & 15;
        float v = 0.f;
        if (tid < 32 && c < Cin) v = tid < 16 ? 1.0f + d.scale[b * Cin + c] : d.shift[b * Cin + c];
        return v;
    };
    auto load_weights = [&]() {                            // after s_w (and s_aff) are visible
#pragma unroll
        for (int ci = 0; ci < Q4_NCH; ++ci) {
            const float f = AFF ? s_aff[ci] : 1.0f;
#pragma unroll
            for (int v = 0; v < Q4_NWR; ++v) {
                const float w = s_w[(ci * Q4_QPAD + 4 * v + (lane >> 4)) * 4 + lj];
                wr[ci][v] = AFF ? w * f : w;
            }
        }
    };
    auto fold_shift = [&]() {                              // s_beta from s_w / s_aff (both visible), then the interior sums; ends with a barrier
        if constexpr (AFF) {
            if (tid < 9 * 16) {
                const int tap = tid >> 4, co = tid & 15;
                float acc_ = 0.f;
                if (co < 4 * Q4_NG) {
#pragma unroll
                    for (int ci = 0; ci < Q4_NCH; ++ci) acc_ = fmaf(s_aff[16 + ci], s_w[(ci * Q4_QPAD + tap * Q4_NG + (co >> 2)) * 4 + (co & 3)], acc_);
                }
                s_beta[tid] = acc_;
            }
            lds_barrier();
#pragma unroll
            for (int n = 0; n < Q4_NG; ++n) {
                float t = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) t += s_beta[tap * 16 + 4 * n + lj];
                beta_l[n] = t;
            }
        }
    };

    // prologue: the first tile's DMA, then the weight loads back to back (one exposed memory latency for the lot)
    int aff_b = -1, ep_b = -1;
    float aff_v = 0.f;
    { const int trace_iter = 5; (void)trace_iter; TRACE(2); }
    if constexpr (AFF) { aff_v = fetch_affine(it.b); aff_b = it.b; }
    issue(it, 0);
    { const int trace_iter = 5; (void)trace_iter; TRACE(3); }
    {
        // s_w[(ci * QPAD + q) * 4 + j] = W(co = 4 n + j, ci, tap), q = tap * NG + n; W(co, ci, t) = w[co][ci][t] or, transposed, w[ci][co][8 - t]
        constexpr int NWV = (S_W + 255) / 256;
        float wv[NWV];
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e = tid + u * 256;
            const int j = e & 3, cq = e >> 2;
            const int ci = cq / Q4_QPAD, q = cq - ci * Q4_QPAD;
            const int tap = q / Q4_NG, n = q - tap * Q4_NG;
            const int co = 4 * n + j;
            float v = 0.f;
            if (e < S_W && q < Q4_NQD && co < Cout && ci < Cin)
                v = d.transposed ? d.w[(ci * d.wCi + co) * 9 + (8 - tap)] : d.w[(co * d.wCi + ci) * 9 + tap];
            wv[u] = v;
        }
        { const int trace_iter = 5; (void)trace_iter; TRACE(4); }
        if constexpr (AFF) { if (tid < 32) s_aff[tid] = aff_v; }
#pragma unroll
        for (int u = 0; u < NWV; ++u)
            if (tid + u * 256 < S_W) s_w[tid + u * 256] = wv[u];
    }
    q4_wait_dma<0>();                                      // this wave's share of tile 0 has landed (the weight loads are complete: their values were stored)
    lds_barrier();
    { const int trace_iter = 5; (void)trace_iter; TRACE(5); }
    load_weights();
    fold_shift();
    { const int trace_iter = 5; (void)trace_iter; TRACE(6); }
    { const int trace_iter = 5; (void)trace_iter; TRACE(1); }
    // A operand: the lane's own pixel, row 2 wave + (lane >> 5), column lane & 31
    const float* a_base = s_in + (2 * wave + (lane >> 5)) * G::RS + (lane & 31) + G::COL0;
    int buf = 0;
    LItem prev = it;
    bool have_prev = false;
    int red_set = 0;                                       // s_red set this tile's epilogue writes (the previous tile wrote red_set ^ 1)
    int trace_iter = 0; (void)trace_iter;
    for (; itx < r1; itx += nlb, ++trace_iter, red_set ^= 1) {
        TRACE(0);
        f32x4 acc[Q4_NG];
#pragma unroll
        for (int n = 0; n < Q4_NG; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool has_next = itx + nlb < r1;
        LItem nxt = it;
        if (has_next) nxt = advance(it);
        TRACE(1);
        if (has_next) issue(nxt, buf ^ 1);                 // lands under the matrix phase; the other buffer was last read before the barrier below
        TRACE(2);
        if constexpr (RED) { if (have_prev) flush_partials(prev, red_set ^ 1); }
        // the epilogue's auxiliary tensors (residual / saved activations) are fetched NOW, into registers: they land under the matrix
        // phase instead of costing one exposed memory latency per tensor in the epilogue
        constexpr int NAUX = (EP == BNERV_EP_BIAS_RES) ? 1 : (EP == BNERV_EP_DGELU_SAVED) ? 2 : (EP == BNERV_EP_DSIN) ? 3 : 0;
        f32x4 ax0[NAUX ? Q4_NG : 1], ax1[NAUX >= 2 ? Q4_NG : 1], ax2[NAUX >= 3 ? Q4_NG : 1];
        unsigned so[Q4_NG], vo[Q4_NG];
        bool ch_ok[Q4_NG];
        {
            const int ty0 = it.ty * TH, tx0 = it.tx * TW;
            const unsigned ob = (unsigned)((((it.b * Cout) * H + ty0 + 2 * wave) * W + tx0) * 4);
            const bool full = ty0 + TH <= H && tx0 + TW <= W;
            bool px_ok = true;
            if (!full) px_ok = (ty0 + 2 * wave + (lb >> 3) < H) && (tx0 + 4 * (lb & 7) < W);
#pragma unroll
            for (int n = 0; n < Q4_NG; ++n) {
                so[n] = ob + (unsigned)n * nstep;
                ch_ok[n] = px_ok && (4 * n + lj < Cout);
                vo[n] = (ch_ok[n] && !(BNERV_ABL4 == 2 && d.B > 0)) ? ovoff : OOB;
            }
            if constexpr (NAUX >= 1) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) ax0[n] = bload(ra0, vo[n], so[n]);
            }
            if constexpr (NAUX >= 2) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) ax1[n] = bload(ra1, vo[n], so[n]);
            }
            if constexpr (NAUX >= 3) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) ax2[n] = d.aux2 ? bload(ra2, vo[n], so[n]) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
        }
        // K loop: the 3 A values of the next tap row are read while the 9 MFMAs of the current one issue
        float a_cur[3], a_nxt[3];
        if (BNERV_ABL4 != 1 || d.B < 0) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) a_cur[kx] = a_base[kx];
#pragma unroll
        for (int ci = 0; ci < Q4_NCH; ++ci) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int nci = ky == 2 ? ci + 1 : ci, nky = ky == 2 ? 0 : ky + 1;
                if (nci < Q4_NCH) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) a_nxt[kx] = a_base[nci * G::PLANE + nky * G::RS + kx];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) {
                        const int q = (ky * 3 + kx) * Q4_NG + n;
                        acc[n] = mfma_qsel(q & 3, a_cur[kx], wr[ci][q >> 2], acc[n]);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) a_cur[kx] = a_nxt[kx];
            }
        }
        }
        TRACE(3);
        TRACE(4);
        TRACE(5);
        // ---- epilogue straight from the accumulators: acc[n] = 4 consecutive pixels of output channel 4 n + lj
        {
            const int ty0 = it.ty * TH, tx0 = it.tx * TW;
            if constexpr (RED) {
                if (it.b != ep_b) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) scl[n] = 4 * n + lj < Cout ? 1.0f + d.scale[it.b * Cout + 4 * n + lj] : 0.f;
                    ep_b = it.b;
                }
            }
            if constexpr (RED) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) { if (!ch_ok[n]) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            if constexpr (AFF) {
                // the folded shift term: every tap of an interior pixel, the existing taps of a pixel on the image border
                const bool border = ty0 == 0 || ty0 + TH >= H || tx0 == 0 || tx0 + TW >= W;
                if (!border) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) acc[n] += beta_l[n];
                } else {
                    const int gy = ty0 + 2 * wave + (lb >> 3), gx0 = tx0 + 4 * (lb & 7);
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) {
                        float col[3];
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            float t = 0.f;
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) t += ((unsigned)(gy + ky - 1) < (unsigned)H) ? s_beta[(ky * 3 + kx) * 16 + 4 * n + lj] : 0.f;
                            col[kx] = t;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = 0.f;
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) t += ((unsigned)(gx0 + e + kx - 1) < (unsigned)W) ? col[kx] : 0.f;
                            acc[n][e] += t;
                        }
                    }
                }
            }
            if constexpr (EP == BNERV_EP_BIAS || EP == BNERV_EP_PLAIN) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) bstore(ro, vo[n], so[n], acc[n] + bias_l[n]);
            } else if constexpr (EP == BNERV_EP_BIAS_SIN) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) {
                    f32x4 sv, cv;
                    sincos4_f(acc[n] + bias_l[n], &sv, &cv);
                    bstore(ro, vo[n], so[n], sv);
                    bstore(ro2, d.out2 ? vo[n] : OOB, so[n], cv);            // always issued (dropped when there is no second output): the DMA wait counts it
                }
            } else if constexpr (EP == BNERV_EP_BIAS_GELU) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) {
                    f32x4 hv, gv;
                    gelu_pair4_f(acc[n] + bias_l[n], &hv, &gv);
                    bstore(ro, vo[n], so[n], hv);
                    bstore(ro2, d.out2 ? vo[n] : OOB, so[n], gv);
                }
            } else if constexpr (EP == BNERV_EP_BIAS_RES) {
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) bstore(ro, vo[n], so[n], acc[n] + bias_l[n] + ax0[n]);
            } else {                                       // DGELU_SAVED / DSIN
                float ps[Q4_NG], pt[Q4_NG];
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) {
                    f32x4 r;
                    const f32x4 v = acc[n];
                    ps[n] = 0.f; pt[n] = 0.f;
                    if constexpr (EP == BNERV_EP_DGELU_SAVED) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl[n] * ax0[n][e]; ps[n] = fmaf(v[e], ax1[n][e], ps[n]); pt[n] += v[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = (ax1[n][e] + v[e] * scl[n]) * ax2[n][e]; ps[n] = fmaf(v[e], ax0[n][e], ps[n]); pt[n] += v[e]; }
                    }
                    bstore(ro, vo[n], so[n], r);
                }
                // per-channel sums: lanes with equal lj (stride 4) hold the same channel
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) {
#pragma unroll
                    for (int off = 4; off < 64; off <<= 1) {
                        ps[n] += __shfl_xor(ps[n], off, 64);
                        pt[n] += __shfl_xor(pt[n], off, 64);
                    }
                }
                if (lane < 4) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) { s_red[red_set * 128 + (wave * 2 + 0) * 16 + 4 * n + lane] = ps[n]; s_red[red_set * 128 + (wave * 2 + 1) * 16 + 4 * n + lane] = pt[n]; }
                }
            }
        }
        TRACE(6);
        if (has_next) {
            // the stores of this epilogue (Q4_NG, twice that with a second output) are younger than the DMA of the next tile
            q4_wait_dma<(TWO_OUT ? 2 : 1) * Q4_NG>();
            lds_barrier();                                 // next tile landed in every wave; every wave is done with this tile's buffer and s_red is visible
            if constexpr (AFF) {
                if (nxt.b != aff_b) {                      // B > 1: the next sample's scale / shift (rare)
                    const float v = fetch_affine(nxt.b);
                    if (tid < 32) s_aff[tid] = v;
                    lds_barrier();
                    load_weights();
                    fold_shift();
                    aff_b = nxt.b;
                }
            }
            buf ^= 1;
            a_base = s_in + buf * S_IN + (2 * wave + (lane >> 5)) * G::RS + (lane & 31) + G::COL0;
        }
        TRACE(7);
        prev = it;
        have_prev = true;
        it = nxt;
    }
    if constexpr (RED) {
        lds_barrier();
        flush_partials(prev, red_set ^ 1);                 // (the loop's increment flipped the set once more after the last tile)
    }
    side_run_hosted(side, smem, vb, vgrid);                // queued slab reductions, least-loaded blocks first (sidejob.h)
}


// dynamic LDS of the kernel (both input buffers, the weight quads, the reduction / affine / folded-shift areas)
inline size_t q4_lds_bytes() {
    using G = Geo<3>;
    constexpr int NSLOT = Q4_NCH * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    return ((size_t)2 * NPRE * 256 * 4 + (size_t)Q4_NCH * Q4_QPAD * 4 + 256 + 32 + 9 * 16) * sizeof(float);
}
// shapes this family takes (the mode is checked by the launcher)
inline bool q4_shape_ok(const KArgs& ka) {
    const bnerv_conv_desc& d = ka.d;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    return ka.vec && d.k == 3 && d.out_s == 1 && d.Cin > 8 && d.Cin <= Q4_NCH && d.Cout <= 4 * Q4_NG && ka.ksplit == 1 &&
           (size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 < LEAN_MAX_BYTES;
}
inline void q4_prepare(KArgs& ka) {                        // item bookkeeping of the persistent blocks
    ka.ngroups = 1;
    ka.total_items = ka.d.B * ka.tiles_x * ka.tiles_y;
    ka.nq_total = 3;
    ka.w_resident = 1;
    ka.magic_tiles = div_magic(ka.tiles_x * ka.tiles_y);
    ka.magic_tiles_x = div_magic(ka.tiles_x);
}

}  // namespace bnerv_q4
