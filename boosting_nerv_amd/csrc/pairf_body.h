// pairf_body.h -- the SHARED-TILE backward pair of a 12-channel 3x3 stride-1 convolution (round 5): the data gradient
//     dx[ci][p] = sum_{co, t} g[co][p - t + pad] w[co][ci][t]            (+ the DGELU_SAVED / DSIN / PLAIN epilogues of conv4_body.h)
// and the weight gradient
//     dw[co][ci][t] = sum_p g[co][p] a[ci][p + t - pad],  a = prologue(x)  (+ the bias column)
// of ONE layer in ONE block, tile by tile, from ONE staged copy of the incoming gradient g (autograd's backward of F.conv2d at
// lib/quant_ops.py:39-41 under ResBlock_SFT / NeRVBlock, model_blocks.py:74-89, :34-39).  Included by wgrad.hip inside its anonymous namespace.
//
// Why.  The interleaved pair (conv_wgrad_pair_kernel) runs the two roles in different blocks; each stages g (+ 1-px halo) for itself, so the
// gradient plane crosses L2 -> CU twice and HBM -> L2 1.3 times (profiles/r04_pmc_pair_dk2s.md: 292.6 MB against 221.2 MB algorithmic,
// and every extra 44 MB tensor of a pair costs ~9 us at 720p: the PLAIN / DGELU_SAVED / DSIN pairs differ by exactly their auxiliary planes).
// Here a block's g tile enters LDS ONCE, by LDS-DMA (no registers, no VALU), double-buffered, and feeds
//   * the data gradient's 4x4x1 K loop (conv4_body.h's: A = the lane's own pixel, weights resident in 84 VGPRs), and
//   * the weight gradient's 16x16x4 K loop as its A operand (16 couts x 4 pixels per step), read from the SAME halo-layout tile
//     (plane stride 400 == 16 (mod 32): an 8-way bank conflict on these 16 reads per wave-tile, ~230 LDS cycles, against the 4 global loads,
//     8 ds_write_b64 and their address arithmetic per thread-tile of the separate staging);
// the weight gradient's B operand (the prologue-transformed input tile with halo, ones / zeros planes for the bias column and the
// padding columns) is staged exactly as wgrad_lean_body does (register prefetch spread over the K loop, committed behind the tile's barrier).
// Two barriers per tile: (A) tile t + 1's g has landed in every wave and everyone is done with tile t's buffers -> commit x(t + 1);
// (B) between the two K loops of tile t + 1: x(t + 1) visible.  One slab per block, summed by the deferred slab reduction as before.
// LDS: 2 x 20 KB (g) + 23.5 KB (x planes) + 5.3 KB (weight quads) + 1.2 KB = 71 KB -> two blocks per CU; <= 256 VGPRs.
#pragma once
#ifdef BNERV_TRACE          // debug variant only (tools/ktrace_pairf.py): s_memtime stamps per wave, tile and phase into conv4_body.h's g_trace4
#define FTRACE(slot) do { if (lane == 0 && vb < 1024 && trace_iter < 6) bnerv_q4::g_trace4[((vb * 4 + wave) * 6 + trace_iter) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FTRACE(slot) do {} while (0)
#endif

template <int EP, int WIN>
__device__ __forceinline__ void pair_fused_body(const bnerv_conv::KArgs& ka, const WArgs& wa, const SidePack& side, const int vb, const int vgrid) {
    using namespace bnerv_q4;
    using GQ = bnerv_conv::Geo<3>;                         // g tile (the data gradient's input): plane stride 400
    using GW = Geo<3>;                                     // x tile (the weight gradient's B operand): plane stride 420
    using bnerv_conv::LItem;
    constexpr unsigned kOOB = 0x80000000u;                 // the out-of-range marker offset of the raw buffer views (conv_common.h)
    // 16-byte slots of a 12-plane halo tile, 101 per plane (100 + one idle): a plane stride of 404 floats == 20 (mod 32) makes the weight
    // gradient's A reads (16 planes x 2 k-lanes per half wave) 2-way bank conflicts; the natural 400 == 16 (mod 32) made them 8-way
    constexpr int PSLOT = GQ::ROWS * GQ::SEGS + 1, GPL = PSLOT * 4;
    constexpr int NSLOT = Q4_NCH * PSLOT;                  // 1212
    constexpr int NPRE = (NSLOT + 255) / 256;              // 5
    constexpr int S_IN = NPRE * 256 * 4;                   // floats per g buffer
    constexpr int S_W = Q4_NCH * Q4_QPAD * 4;
    constexpr bool RED = (EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    constexpr bool WAFF = (WIN == BNERV_IN_AFFINE);
    constexpr int NTW = 7, NPLL = 12;
    constexpr int NXS = NPRE;
    static_assert(GQ::PLANE == GQ::PLANE_RAW && S_IN >= Q4_NCH * GPL && GW::ROWS == GQ::ROWS && GW::SEGS == GQ::SEGS, "slot geometry");
    static_assert(EP == BNERV_EP_PLAIN || RED, "epilogues of the backward pairs");
    const bnerv_conv_desc& d = ka.d;                       // the data gradient as a convolution: d.x = g, d.Cin = couts of the layer, d.Cout = its input channels
    const bnerv_wgrad_desc& w = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g2 = smem;                                    // [2][S_IN] g tiles, filled by LDS-DMA
    float* s_w = smem + 2 * S_IN;                          // [co-as-ci][quad][4] compact weight quads of the data gradient
    float* s_red = s_w + S_W;                              // [4 waves][2][16] per-channel sums of the waves at a flush (DGELU_SAVED / DSIN)
    float* s_x = s_red + 256;                              // (NPLL + 2) planes of GW::PLANE + dump area of the idle slots
    float* s_affw = s_x + (NPLL + 2) * GW::PLANE + 64 * 4;                     // [2][16] affine prologue of the weight gradient's input (behind the dump area)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 3, lb = lane >> 2;
    const int li = lane & 15, kq = lane >> 4;
    const int Cg = d.Cin, Cx = d.Cout, H = d.H, W = d.W;   // Cg = channels of g (= w.Cout), Cx = channels of x / dx (= w.Cin)
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;
    const int nW = Cx * 9;

    // this block's item range: XCD x owns a contiguous slice of the tile list; its blocks take it round-robin (conv4_body.h)
    const int xcd = vb & 7, lbk = vb >> 3;
    const int nlb = (vgrid - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lbk;
    const bool has_work = itx < r1;
    const int step_q = bnerv_conv::fast_div(nlb, ka.magic_tiles_x), step_r = nlb - step_q * tiles_x;
    LItem it{0, 0, 0};
    if (has_work) {
        const int tiles = tiles_x * tiles_y;
        it.b = bnerv_conv::fast_div(itx, ka.magic_tiles);
        const int t = itx - it.b * tiles;
        it.ty = bnerv_conv::fast_div(t, ka.magic_tiles_x);
        it.tx = t - it.ty * tiles_x;
    }
    auto advance = [&](LItem a) {
        a.tx += step_r;
        a.ty += step_q;
        if (a.tx >= tiles_x) { a.tx -= tiles_x; ++a.ty; }
        while (a.ty >= tiles_y) { a.ty -= tiles_y; ++a.b; }
        return a;
    };

    // constant planes of the weight gradient: ones (bias column) and zeros (columns beyond the weight matrix)
    for (int i = tid; i < 2 * GW::PLANE; i += 256) s_x[NPLL * GW::PLANE + i] = i < GW::PLANE ? 1.0f : 0.0f;

    // ---- per-slot constants: slot = (channel, halo row, 4-px segment); thread t owns slots t, t + 256, ... of BOTH tiles
    auto slot_geom = [&](int k, int& c, int& r, int& sg) {     // (r == ROWS: the plane's idle 101st slot)
        const int sidx = tid + k * 256;
        c = sidx / PSLOT;
        const int rem = sidx - c * PSLOT;
        r = rem / GQ::SEGS;
        sg = rem - r * GQ::SEGS;
    };
    auto slot_inside = [&](int k, int ty0, int tx0) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        const int gy = ty0 + r - GQ::PAD, gx = tx0 + 4 * sg - GQ::XOFF;
        return (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    };
    unsigned voffg[NPRE], voffx[NXS];
    int loffx[NXS];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        const bool real = tid + k * 256 < NSLOT && r < GQ::ROWS;
        const unsigned off = (unsigned)(((c * H + r) * W + 4 * sg) * 4);
        voffg[k] = (real && c < Cg) ? off : kOOB;          // out of range: the DMA writes zeros
        voffx[k] = (real && c < Cx) ? off : kOOB;
        loffx[k] = real ? (c * GW::PLANE + r * GW::RS + 4 * sg) * 4 : ((NPLL + 2) * GW::PLANE + (tid & 63) * 4) * 4;      // idle slots: a 64-slot dump area
    }
    const unsigned shift = (unsigned)((GQ::PAD * W + GQ::XOFF) * 4);         // both views start PAD rows + XOFF columns early: offsets >= 0
    const unsigned g_bytes = (unsigned)((size_t)d.B * Cg * H * W * 4) + shift;
    const unsigned x_bytes = (unsigned)((size_t)d.B * Cx * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cx * H * W * 4);
    bnerv_conv::i32x4 rg;
    {
        const uintptr_t base = reinterpret_cast<uintptr_t>(d.x) - shift;
        rg[0] = (int)(unsigned)(base & 0xffffffffu);
        rg[1] = (int)(unsigned)((base >> 32) & 0xffffu);
        rg[2] = (int)g_bytes;
        rg[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t rx = bnerv_conv::make_rsrc(w.x, shift, x_bytes);
    const __amdgpu_buffer_rsrc_t ro = bnerv_conv::make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = bnerv_conv::make_rsrc(d.aux0 ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = bnerv_conv::make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = bnerv_conv::make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);
    const unsigned lds_g = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_g2 + (unsigned)wave * 1024u;

    // data-gradient epilogue lane constants (conv4_body.h)
    const unsigned ovoff = (unsigned)(((lj * H + (lb >> 3)) * W + 4 * (lb & 7)) * 4);
    const unsigned nstep = (unsigned)(4 * H * W * 4);
    float scl[Q4_NG];
#pragma unroll
    for (int n = 0; n < Q4_NG; ++n) scl[n] = 0.f;

    auto issue_g = [&](const LItem& a, int buf) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const unsigned sb = (unsigned)((((a.b * Cg) * H + ty0) * W + tx0) * 4);
        const unsigned lbase = lds_g + (unsigned)buf * (unsigned)(S_IN * 4);
        const bool interior = ty0 >= GQ::PAD && ty0 + TH + GQ::PAD <= H && tx0 >= GQ::XOFF && tx0 + TW + GQ::XOFF <= W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rg, voffg[k], sb, lbase + (unsigned)k * 4096u);
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rg, slot_inside(k, ty0, tx0) ? voffg[k] : kOOB, sb, lbase + (unsigned)k * 4096u);
        }
    };
    // Per-channel sums of the DGELU_SAVED / DSIN epilogues: [tiles][B][2][C] rows that a deferred slab reduction adds up.  The block
    // keeps PER-LANE running sums over its tiles and reduces them across lanes and waves ONCE per (block, sample) -- into the row of the
    // last tile it ran for that sample; the rows of its other tiles are written as zeros.  (Per tile the cross-lane reduction was 24
    // ds_bpermute + adds on the critical path of every epilogue: ~1.5 k of a tile's ~20 k cycles, tools/ktrace_pairf.py.)
    auto write_row = [&](const LItem& a, const bool real) {                 // wave 0: the row of tile `a`; real: from s_red (visible), else zeros
        if (wave == 0 && lane < 32) {
            const int q = lane >> 4, c = lane & 15;
            float s = 0.f;
            if (real) s = ((s_red[(0 * 2 + q) * 16 + c] + s_red[(1 * 2 + q) * 16 + c]) + s_red[(2 * 2 + q) * 16 + c]) + s_red[(3 * 2 + q) * 16 + c];
            const size_t row = (size_t)(a.ty * tiles_x + a.tx) * d.B + a.b;
            if (c < Cx) d.partial[(row * 2 + q) * Cx + c] = s;
        }
    };

    // ---- the weight gradient's input tile: register prefetch, prologue on the way into LDS (wgrad_lean_body)
    float sc[NXS], sh[NXS];
    auto load_affine = [&](int b) {                        // ends with the values in registers; two barriers
        const int c = tid & 15;
        float v = 0.f;
        if (tid < 32 && c < Cx) v = tid < 16 ? 1.0f + w.scale[b * Cx + c] : w.shift[b * Cx + c];
        lds_barrier();
        if (tid < 32) s_affw[tid] = v;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            int c2, r, sg;
            slot_geom(k, c2, r, sg);
            const bool ok = voffx[k] != kOOB;
            sc[k] = ok ? s_affw[c2 & 15] : 0.f;
            sh[k] = ok ? s_affw[16 + (c2 & 15)] : 0.f;
        }
    };
    f32x4 xa[NXS];
    struct Pre { unsigned sbx; int ty0, tx0; bool interior; };
    auto prep = [&](const LItem& a) {
        Pre q;
        q.ty0 = a.ty * TH; q.tx0 = a.tx * TW;
        q.sbx = (unsigned)((((a.b * Cx) * H + q.ty0) * W + q.tx0) * 4);
        q.interior = q.ty0 >= GQ::PAD && q.ty0 + TH + GQ::PAD <= H && q.tx0 >= GQ::XOFF && q.tx0 + TW + GQ::XOFF <= W;
        return q;
    };
    auto issue_x = [&](const Pre& q, int k) {              // k is a compile-time constant at every call site
        unsigned vo = voffx[k];
        if (!q.interior) vo = slot_inside(k, q.ty0, q.tx0) ? vo : kOOB;
        xa[k] = bnerv_conv::bload(rx, vo, q.sbx);
    };
    auto commit_x = [&](const LItem& a) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const bool interior = ty0 >= GQ::PAD && ty0 + TH + GQ::PAD <= H && tx0 >= GQ::XOFF && tx0 + TW + GQ::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            f32x4 v = xa[k];
            if constexpr (WAFF) {
                float s = sc[k], h = sh[k];
                if (!interior) {                           // zero padding is applied AFTER the prologue
                    const bool ok = slot_inside(k, ty0, tx0);
                    s = ok ? s : 0.f;
                    h = ok ? h : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * s + h;
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(s_x) + loffx[k]) = v;
        }
    };

    // ---- resident weight quads of the data gradient (conv4_body.h: wr[ci][v], lane l = quad 4 v + (l >> 4), element l & 3)
    float wr[Q4_NCH][Q4_NWR];
    auto load_weights = [&]() {
#pragma unroll
        for (int ci = 0; ci < Q4_NCH; ++ci)
#pragma unroll
            for (int v = 0; v < Q4_NWR; ++v) wr[ci][v] = s_w[(ci * Q4_QPAD + 4 * v + (lane >> 4)) * 4 + lj];
    };

    // ---- the weight gradient's fragment bases.  pixel of (wave, step, kq): row 2 wave + (step >> 3), column 4 (step & 7) + kq
    int bbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = nt * 16 + li;
        int off;
        if (n < nW) {
            const int ci = n / 9, tap = n - ci * 9;
            off = ci * GW::PLANE + (tap / 3) * GW::RS + (tap % 3) + GW::COL0;
        } else {
            off = (n == nW ? NPLL : NPLL + 1) * GW::PLANE;
        }
        bbase[nt] = off + (2 * wave) * GW::RS + kq;
    }
    // A operand from the g halo tile: plane li, the CENTRE of pixel (row, column) = halo (row + PAD, column + XOFF); rows 12..15 read the
    // slots behind plane 11 / the other buffer: they only reach output rows that are dropped
    const int abase = li * GPL + (2 * wave + GQ::PAD) * GQ::RS + GQ::XOFF + kq;

    f32x4 accw[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) accw[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (has_work) {
        // prologue: the first tile's DMA and x loads, then the weight loads back to back (one exposed memory latency for the lot)
        issue_g(it, 0);
        {
            const Pre q = prep(it);
#pragma unroll
            for (int k = 0; k < NXS; ++k) issue_x(q, k);
        }
        {
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
                if (e < S_W && q < Q4_NQD && co < Cx && ci < Cg)
                    v = d.transposed ? d.w[(ci * d.wCi + co) * 9 + (8 - tap)] : d.w[(co * d.wCi + ci) * 9 + tap];
                wv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < NWV; ++u)
                if (tid + u * 256 < S_W) s_w[tid + u * 256] = wv[u];
        }
        int aff_b = -1, ep_b = -1;
        if constexpr (WAFF) { load_affine(it.b); aff_b = it.b; }
        commit_x(it);
        q4_wait_dma<0>();                                  // this wave's share of g tile 0 has landed
        lds_barrier();
        if constexpr (!RED) load_weights();                // (no auxiliary registers in flight: the 84 weight registers stay resident)

        const float* a_base = s_g2 + (2 * wave + (lane >> 5)) * GQ::RS + (lane & 31) + GQ::COL0;
        const float* s_gc = s_g2;                          // the current g buffer
        int buf = 0;
        float rs[Q4_NG], rt[Q4_NG];                        // per-lane running channel sums (RED)
#pragma unroll
        for (int n = 0; n < Q4_NG; ++n) { rs[n] = 0.f; rt[n] = 0.f; }
        int trace_iter = 0; (void)trace_iter;
        for (; itx < r1; itx += nlb, ++trace_iter) {
            FTRACE(0);
            f32x4 acc[Q4_NG];
#pragma unroll
            for (int n = 0; n < Q4_NG; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool has_next = itx + nlb < r1;
            LItem nxt = it;
            if (has_next) nxt = advance(it);
            if (has_next) issue_g(nxt, buf ^ 1);           // lands under the matrix phases; the other buffer was last read before barrier (A)
            constexpr int NAUX = (EP == BNERV_EP_DGELU_SAVED) ? 2 : (EP == BNERV_EP_DSIN) ? 3 : 0;
            f32x4 ax0[NAUX ? Q4_NG : 1], ax1[NAUX >= 2 ? Q4_NG : 1], ax2[NAUX >= 3 ? Q4_NG : 1];
            unsigned so[Q4_NG], vo[Q4_NG];
            bool ch_ok[Q4_NG];
            {
                const int ty0 = it.ty * TH, tx0 = it.tx * TW;
                const unsigned ob = (unsigned)((((it.b * Cx) * H + ty0 + 2 * wave) * W + tx0) * 4);
                const bool full = ty0 + TH <= H && tx0 + TW <= W;
                bool px_ok = true;
                if (!full) px_ok = (ty0 + 2 * wave + (lb >> 3) < H) && (tx0 + 4 * (lb & 7) < W);
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) {
                    so[n] = ob + (unsigned)n * nstep;
                    ch_ok[n] = px_ok && (4 * n + lj < Cx);
                    vo[n] = ch_ok[n] ? ovoff : kOOB;
                }
            }
            // ---- data gradient: K loop over (g channel, tap row); the 3 A values of the next tap row are read while 9 MFMAs issue.
            // The 84 weight registers are re-read from LDS per tile (84 ds_read_b32 against the tile's 236): resident across the weight
            // gradient's phase they would push the DSIN form (36 auxiliary registers in flight) past 256 VGPRs into scratch
            {
                if constexpr (RED) load_weights();
#ifdef BNERV_PAIRF_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
                float a_cur[3], a_nxt[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) a_cur[kx] = a_base[kx];
#pragma unroll
                for (int ci = 0; ci < Q4_NCH; ++ci) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int nci = ky == 2 ? ci + 1 : ci, nky = ky == 2 ? 0 : ky + 1;
                        if (nci < Q4_NCH) {
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a_nxt[kx] = a_base[nci * GPL + nky * GQ::RS + kx];
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
            // the epilogue's auxiliary tensors are fetched NOW, into registers: they land under the weight gradient's K loop (fetched before the
            // data gradient's they would be live next to its 84 weight registers)
            {
                if constexpr (NAUX >= 1) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) ax0[n] = bnerv_conv::bload(ra0, vo[n], so[n]);
                }
                if constexpr (NAUX >= 2) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) ax1[n] = bnerv_conv::bload(ra1, vo[n], so[n]);
                }
                if constexpr (NAUX >= 3) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) ax2[n] = d.aux2 ? bnerv_conv::bload(ra2, vo[n], so[n]) : f32x4{1.f, 1.f, 1.f, 1.f};
                }
            }
#ifdef BNERV_PAIRF_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            FTRACE(1);
            lds_barrier();                                 // (B) this tile's x planes (committed behind barrier (A) / in the prologue) are visible
            FTRACE(2);
            // ---- weight gradient: 16 K steps of 4 pixels; the next tile's x loads are issued one at a time, spread over the steps
            {
                const Pre pre = prep(nxt);
#ifdef BNERV_PAIRF_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    if (has_next) {
#pragma unroll
                        for (int k = 0; k < NXS; ++k)
                            if (k * 16 / NXS == st) issue_x(pre, k);
                    }
                    float bf[NTW];
                    const float af = s_gc[abase + (st >> 3) * GQ::RS + (st & 7) * 4];
#pragma unroll
                    for (int n = 0; n < NTW; ++n) bf[n] = s_x[bbase[n] + ((st >> 3) * GW::RS + (st & 7) * 4)];
#pragma unroll
                    for (int n = 0; n < NTW; ++n) accw[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], accw[n], 0, 0, 0);
                }
            }
#ifdef BNERV_PAIRF_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            FTRACE(3);
            // ---- data-gradient epilogue straight from the accumulators (conv4_body.h)
            {
                if constexpr (RED) {
                    if (it.b != ep_b) {
#pragma unroll
                        for (int n = 0; n < Q4_NG; ++n) scl[n] = 4 * n + lj < Cx ? 1.0f + d.scale[it.b * Cx + 4 * n + lj] : 0.f;
                        ep_b = it.b;
                    }
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) { if (!ch_ok[n]) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
                if constexpr (EP == BNERV_EP_PLAIN) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) bnerv_conv::bstore(ro, vo[n], so[n], acc[n]);
                } else {
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
                        bnerv_conv::bstore(ro, vo[n], so[n], r);
                    }
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) { rs[n] += ps[n]; rt[n] += pt[n]; }
                    const bool flush = !has_next || nxt.b != it.b;          // block-uniform
                    if (!flush) {
                        write_row(it, false);
                    } else {
#pragma unroll
                        for (int n = 0; n < Q4_NG; ++n) {
#pragma unroll
                            for (int off = 4; off < 64; off <<= 1) {
                                rs[n] += __shfl_xor(rs[n], off, 64);
                                rt[n] += __shfl_xor(rt[n], off, 64);
                            }
                        }
                        if (lane < 4) {
#pragma unroll
                            for (int n = 0; n < Q4_NG; ++n) { s_red[(wave * 2 + 0) * 16 + 4 * n + lane] = rs[n]; s_red[(wave * 2 + 1) * 16 + 4 * n + lane] = rt[n]; }
                        }
                        lds_barrier();
                        write_row(it, true);
#pragma unroll
                        for (int n = 0; n < Q4_NG; ++n) { rs[n] = 0.f; rt[n] = 0.f; }
                    }
                }
            }
            FTRACE(4);
            if (has_next) {
                // everything but this epilogue's Q4_NG stores is complete: the next g tile (DMA) and the next x tile (registers)
                q4_wait_dma<Q4_NG>();
                FTRACE(5);
                lds_barrier();                             // (A) next g landed in every wave; every wave is done with this tile's g, x and s_red is visible
                if constexpr (WAFF) {
                    if (nxt.b != aff_b) { load_affine(nxt.b); aff_b = nxt.b; }      // B > 1: the next sample's scale / shift (rare)
                }
                FTRACE(6);
                commit_x(nxt);
                buf ^= 1;
                s_gc = s_g2 + buf * S_IN;
                a_base = s_gc + (2 * wave + (lane >> 5)) * GQ::RS + (lane & 31) + GQ::COL0;
            }
            FTRACE(7);
            it = nxt;
        }
    }

    // ---- the block's weight-gradient slab: cross-wave reduction through LDS (fixed order => deterministic)
    __syncthreads();
    float* s_sum = smem;
    constexpr int RW = NTW * 16, RSZ = 16 * RW;
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_sum[wave * RSZ + (4 * kq + r) * RW + n * 16 + li] = accw[n][r];
    __syncthreads();
    float* slab = wa.slab + (size_t)vb * Cg * wa.ncols;
    for (int idx = tid; idx < RSZ; idx += 256) {
        const int row = idx / RW, col = idx - row * RW;
        if (row < Cg && col < wa.ncols)
            slab[(size_t)row * wa.ncols + col] = (s_sum[idx] + s_sum[RSZ + idx]) + (s_sum[2 * RSZ + idx] + s_sum[3 * RSZ + idx]);
    }
    side_run_hosted(side, smem, vb, vgrid);
}

// The FOLD form of the shared-tile pair for an affine prologue on the weight gradient's input, B == 1 (pairf_body.h):
//     a = x (1 + s_ci) + t_ci inside the image, 0 outside   ==>   dW[co][ci][tap] = (1 + s_ci) R[co][ci][tap] + t_ci M[co][tap],
//     R = sum_p g[co][p] x[ci][p + tap - pad]  (x zero outside),      M[co][tap] = sum_p g[co][p] inside(p + tap - pad)
// so the input tile is staged RAW -- by LDS-DMA, like g: no registers, no fma, no ds_write -- and the 16x16x4 K loop accumulates R (108 columns)
// and M (9 more columns whose B operand reads a halo-aware MASK plane: ones inside the image, rewritten only on border tiles; its centre tap is the
// bias column) in EIGHT N tiles.  The slab reduction applies (1 + s), t (sidejob.h: fold jobs).  The reducing epilogues read their raw-input operand
// (`aux0` of DSIN, `aux1` of DGELU_SAVED = the weight gradient's x) from the tile in LDS instead of global memory.
template <int EP>
__device__ __forceinline__ void pair_fold_body(const bnerv_conv::KArgs& ka, const WArgs& wa, const SidePack& side, const int vb, const int vgrid) {
    using namespace bnerv_q4;
    using GQ = bnerv_conv::Geo<3>;                         // g tile (the data gradient's input): plane stride 400
    using GW = Geo<3>;                                     // x tile (the weight gradient's B operand): plane stride 420
    using bnerv_conv::LItem;
    constexpr unsigned kOOB = 0x80000000u;                 // the out-of-range marker offset of the raw buffer views (conv_common.h)
    // 16-byte slots of a 12-plane halo tile, 101 per plane (100 + one idle): a plane stride of 404 floats == 20 (mod 32) makes the weight
    // gradient's A reads (16 planes x 2 k-lanes per half wave) 2-way bank conflicts; the natural 400 == 16 (mod 32) made them 8-way
    constexpr int PSLOT = GQ::ROWS * GQ::SEGS + 1, GPL = PSLOT * 4;
    constexpr int NSLOT = Q4_NCH * PSLOT;                  // 1212
    constexpr int NPRE = (NSLOT + 255) / 256;              // 5
    constexpr int S_IN = NPRE * 256 * 4;                   // floats per g buffer
    constexpr int S_W = Q4_NCH * Q4_QPAD * 4;
    constexpr bool RED = (EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    constexpr int NTW = 8, NPLL = 12;
    constexpr int XSLOT = GW::PLANE / 4;                   // 105 16-byte slots per x plane (100 + 5 idle): the 420-float stride of wgrad_lean_body
    constexpr int NXSLOT = NPLL * XSLOT;                   // 1260
    constexpr int NXS = (NXSLOT + 255) / 256;              // 5 DMA passes; their 1280 slots = S_X floats
    constexpr int S_X = NXS * 256 * 4;
    constexpr int MASKP = S_X, ZEROP = S_X + GW::PLANE;    // the mask plane and the zeros plane behind the DMA area
    static_assert(EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED, "the fold form serves the TAT pairs");
    static_assert(GQ::PLANE == GQ::PLANE_RAW && S_IN >= Q4_NCH * GPL && GW::ROWS == GQ::ROWS && GW::SEGS == GQ::SEGS, "slot geometry");
    const bnerv_conv_desc& d = ka.d;                       // the data gradient as a convolution: d.x = g, d.Cin = couts of the layer, d.Cout = its input channels
    const bnerv_wgrad_desc& w = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g2 = smem;                                    // [2][S_IN] g tiles, filled by LDS-DMA
    float* s_w = smem + 2 * S_IN;                          // [co-as-ci][quad][4] compact weight quads of the data gradient
    float* s_red = s_w + S_W;                              // [4 waves][2][16] per-channel sums of the waves at a flush (DGELU_SAVED / DSIN)
    float* s_x = s_red + 256;                              // [S_X] raw x planes (LDS-DMA), then the mask plane, then the zeros plane

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 3, lb = lane >> 2;
    const int li = lane & 15, kq = lane >> 4;
    const int Cg = d.Cin, Cx = d.Cout, H = d.H, W = d.W;   // Cg = channels of g (= w.Cout), Cx = channels of x / dx (= w.Cin)
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;
    const int nW = Cx * 9;

    // this block's item range: XCD x owns a contiguous slice of the tile list; its blocks take it round-robin (conv4_body.h)
    const int xcd = vb & 7, lbk = vb >> 3;
    const int nlb = (vgrid - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lbk;
    const bool has_work = itx < r1;
    const int step_q = bnerv_conv::fast_div(nlb, ka.magic_tiles_x), step_r = nlb - step_q * tiles_x;
    LItem it{0, 0, 0};
    if (has_work) {
        const int tiles = tiles_x * tiles_y;
        it.b = bnerv_conv::fast_div(itx, ka.magic_tiles);
        const int t = itx - it.b * tiles;
        it.ty = bnerv_conv::fast_div(t, ka.magic_tiles_x);
        it.tx = t - it.ty * tiles_x;
    }
    auto advance = [&](LItem a) {
        a.tx += step_r;
        a.ty += step_q;
        if (a.tx >= tiles_x) { a.tx -= tiles_x; ++a.ty; }
        while (a.ty >= tiles_y) { a.ty -= tiles_y; ++a.b; }
        return a;
    };

    // constant planes of the weight gradient: ones (bias column) and zeros (columns beyond the weight matrix)
    for (int i = tid; i < 2 * GW::PLANE; i += 256) s_x[MASKP + i] = i < GW::PLANE ? 1.0f : 0.0f;
    bool mask_ones = true;                                 // the mask plane holds all ones (interior tiles); block-uniform

    // ---- per-slot constants: slot = (channel, halo row, 4-px segment); thread t owns slots t, t + 256, ... of BOTH tiles
    auto slot_geom = [&](int k, int& c, int& r, int& sg) {     // (r == ROWS: the plane's idle 101st slot)
        const int sidx = tid + k * 256;
        c = sidx / PSLOT;
        const int rem = sidx - c * PSLOT;
        r = rem / GQ::SEGS;
        sg = rem - r * GQ::SEGS;
    };
    auto slot_inside = [&](int k, int ty0, int tx0) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        const int gy = ty0 + r - GQ::PAD, gx = tx0 + 4 * sg - GQ::XOFF;
        return (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    };
    unsigned voffg[NPRE], voffx[NXS];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        const bool real = tid + k * 256 < NSLOT && r < GQ::ROWS;
        const unsigned off = (unsigned)(((c * H + r) * W + 4 * sg) * 4);
        voffg[k] = (real && c < Cg) ? off : kOOB;          // out of range: the DMA writes zeros
    }
    auto xslot_geom = [&](int k, int& c, int& r, int& sg) {    // x tile: 105 slots per plane (r >= ROWS: idle)
        const int sidx = tid + k * 256;
        c = sidx / XSLOT;
        const int rem = sidx - c * XSLOT;
        r = rem / GQ::SEGS;
        sg = rem - r * GQ::SEGS;
    };
    auto xslot_inside = [&](int k, int ty0, int tx0) {
        int c, r, sg;
        xslot_geom(k, c, r, sg);
        const int gy = ty0 + r - GQ::PAD, gx = tx0 + 4 * sg - GQ::XOFF;
        return (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    };
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
        int c, r, sg;
        xslot_geom(k, c, r, sg);
        const bool real = tid + k * 256 < NXSLOT && r < GQ::ROWS;
        voffx[k] = (real && c < Cx) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : kOOB;
    }
    const unsigned shift = (unsigned)((GQ::PAD * W + GQ::XOFF) * 4);         // both views start PAD rows + XOFF columns early: offsets >= 0
    const unsigned g_bytes = (unsigned)((size_t)d.B * Cg * H * W * 4) + shift;
    const unsigned x_bytes = (unsigned)((size_t)d.B * Cx * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cx * H * W * 4);
    bnerv_conv::i32x4 rg;
    {
        const uintptr_t base = reinterpret_cast<uintptr_t>(d.x) - shift;
        rg[0] = (int)(unsigned)(base & 0xffffffffu);
        rg[1] = (int)(unsigned)((base >> 32) & 0xffffu);
        rg[2] = (int)g_bytes;
        rg[3] = 0x00020000;
    }
    bnerv_conv::i32x4 rxd;                                 // raw descriptor of the x view for the DMA asm
    {
        const uintptr_t base = reinterpret_cast<uintptr_t>(w.x) - shift;
        rxd[0] = (int)(unsigned)(base & 0xffffffffu);
        rxd[1] = (int)(unsigned)((base >> 32) & 0xffffu);
        rxd[2] = (int)x_bytes;
        rxd[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t ro = bnerv_conv::make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = bnerv_conv::make_rsrc(d.aux0 ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = bnerv_conv::make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = bnerv_conv::make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);
    const unsigned lds_g = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_g2 + (unsigned)wave * 1024u;
    const unsigned lds_x = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_x + (unsigned)wave * 1024u;

    // data-gradient epilogue lane constants (conv4_body.h)
    const unsigned ovoff = (unsigned)(((lj * H + (lb >> 3)) * W + 4 * (lb & 7)) * 4);
    const unsigned nstep = (unsigned)(4 * H * W * 4);
    float scl[Q4_NG];
#pragma unroll
    for (int n = 0; n < Q4_NG; ++n) scl[n] = 0.f;

    auto issue_g = [&](const LItem& a, int buf) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const unsigned sb = (unsigned)((((a.b * Cg) * H + ty0) * W + tx0) * 4);
        const unsigned lbase = lds_g + (unsigned)buf * (unsigned)(S_IN * 4);
        const bool interior = ty0 >= GQ::PAD && ty0 + TH + GQ::PAD <= H && tx0 >= GQ::XOFF && tx0 + TW + GQ::XOFF <= W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rg, voffg[k], sb, lbase + (unsigned)k * 4096u);
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dma16(rg, slot_inside(k, ty0, tx0) ? voffg[k] : kOOB, sb, lbase + (unsigned)k * 4096u);
        }
    };
    // Per-channel sums of the DGELU_SAVED / DSIN epilogues: [tiles][B][2][C] rows that a deferred slab reduction adds up.  The block
    // keeps PER-LANE running sums over its tiles and reduces them across lanes and waves ONCE per (block, sample) -- into the row of the
    // last tile it ran for that sample; the rows of its other tiles are written as zeros.  (Per tile the cross-lane reduction was 24
    // ds_bpermute + adds on the critical path of every epilogue: ~1.5 k of a tile's ~20 k cycles, tools/ktrace_pairf.py.)
    auto write_row = [&](const LItem& a, const bool real) {                 // wave 0: the row of tile `a`; real: from s_red (visible), else zeros
        if (wave == 0 && lane < 32) {
            const int q = lane >> 4, c = lane & 15;
            float s = 0.f;
            if (real) s = ((s_red[(0 * 2 + q) * 16 + c] + s_red[(1 * 2 + q) * 16 + c]) + s_red[(2 * 2 + q) * 16 + c]) + s_red[(3 * 2 + q) * 16 + c];
            const size_t row = (size_t)(a.ty * tiles_x + a.tx) * d.B + a.b;
            if (c < Cx) d.partial[(row * 2 + q) * Cx + c] = s;
        }
    };

    // ---- the weight gradient's input tile: raw, by LDS-DMA (zeros outside the image / beyond Cx); the mask plane follows the tile
    auto issue_xd = [&](const LItem& a) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const unsigned sb = (unsigned)((((a.b * Cx) * H + ty0) * W + tx0) * 4);
        const bool interior = ty0 >= GQ::PAD && ty0 + TH + GQ::PAD <= H && tx0 >= GQ::XOFF && tx0 + TW + GQ::XOFF <= W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NXS; ++k) dma16(rxd, voffx[k], sb, lds_x + (unsigned)k * 4096u);
        } else {
#pragma unroll
            for (int k = 0; k < NXS; ++k) dma16(rxd, xslot_inside(k, ty0, tx0) ? voffx[k] : kOOB, sb, lds_x + (unsigned)k * 4096u);
        }
    };
    auto set_mask = [&](const LItem& a) {                   // block-uniform decisions; plain LDS stores, ordered by the barrier that follows
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const bool interior = ty0 >= GQ::PAD && ty0 + TH + GQ::PAD <= H && tx0 >= GQ::XOFF && tx0 + TW + GQ::XOFF <= W;
        if (interior && mask_ones) return;
        for (int i = tid; i < GQ::ROWS * GW::RS; i += 256) {
            const int r = i / GW::RS, c = i - r * GW::RS;
            const int gy = ty0 + r - GQ::PAD, gx = tx0 + c - GQ::XOFF;
            s_x[MASKP + i] = (interior || ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)) ? 1.0f : 0.0f;
        }
        mask_ones = interior;
    };

    // ---- resident weight quads of the data gradient (conv4_body.h: wr[ci][v], lane l = quad 4 v + (l >> 4), element l & 3)
    float wr[Q4_NCH][Q4_NWR];
    auto load_weights = [&]() {
#pragma unroll
        for (int ci = 0; ci < Q4_NCH; ++ci)
#pragma unroll
            for (int v = 0; v < Q4_NWR; ++v) wr[ci][v] = s_w[(ci * Q4_QPAD + 4 * v + (lane >> 4)) * 4 + lj];
    };

    // ---- the weight gradient's fragment bases.  pixel of (wave, step, kq): row 2 wave + (step >> 3), column 4 (step & 7) + kq
    int bbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = nt * 16 + li;
        int off;
        if (n < nW) {
            const int ci = n / 9, tap = n - ci * 9;
            off = ci * GW::PLANE + (tap / 3) * GW::RS + (tap % 3) + GW::COL0;
        } else if (n < nW + 9) {                           // the nine M columns follow the Cx * 9 weight columns (slab row = ncols + 8 columns)
            const int tap = n - nW;
            off = MASKP + (tap / 3) * GW::RS + (tap % 3) + GW::COL0;
        } else {
            off = ZEROP;
        }
        bbase[nt] = off + (2 * wave) * GW::RS + kq;
    }
    // A operand from the g halo tile: plane li, the CENTRE of pixel (row, column) = halo (row + PAD, column + XOFF); rows 12..15 read the
    // slots behind plane 11 / the other buffer: they only reach output rows that are dropped
    const int abase = li * GPL + (2 * wave + GQ::PAD) * GQ::RS + GQ::XOFF + kq;

    f32x4 accw[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) accw[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (has_work) {
        // prologue: the first tile's two DMAs, then the weight loads back to back (one exposed memory latency for the lot)
        issue_g(it, 0);
        set_mask(it);
        issue_xd(it);
        {
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
                if (e < S_W && q < Q4_NQD && co < Cx && ci < Cg)
                    v = d.transposed ? d.w[(ci * d.wCi + co) * 9 + (8 - tap)] : d.w[(co * d.wCi + ci) * 9 + tap];
                wv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < NWV; ++u)
                if (tid + u * 256 < S_W) s_w[tid + u * 256] = wv[u];
        }
        int ep_b = -1;
        q4_wait_dma<0>();                                  // this wave's shares of g tile 0 and x tile 0 have landed
        lds_barrier();
#ifndef BNERV_FOLD_RELOAD
        load_weights();                                    // resident: without the x prefetch registers the 84 weight registers fit (244 / 256 VGPRs)
#endif

        const float* a_base = s_g2 + (2 * wave + (lane >> 5)) * GQ::RS + (lane & 31) + GQ::COL0;
        const float* s_gc = s_g2;                          // the current g buffer
        int buf = 0;
        float rs[Q4_NG], rt[Q4_NG];                        // per-lane running channel sums (RED)
#pragma unroll
        for (int n = 0; n < Q4_NG; ++n) { rs[n] = 0.f; rt[n] = 0.f; }
        int trace_iter = 0; (void)trace_iter;
        for (; itx < r1; itx += nlb, ++trace_iter) {
            FTRACE(0);
            f32x4 acc[Q4_NG];
#pragma unroll
            for (int n = 0; n < Q4_NG; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool has_next = itx + nlb < r1;
            LItem nxt = it;
            if (has_next) nxt = advance(it);
            if (has_next) issue_g(nxt, buf ^ 1);           // lands under the matrix phases; the other buffer was last read before barrier (A)
            f32x4 ax0[Q4_NG], ax1[Q4_NG], ax2[Q4_NG];       // (DGELU_SAVED uses ax0 = gelu'; DSIN ax1, ax2)
            unsigned so[Q4_NG], vo[Q4_NG];
            bool ch_ok[Q4_NG];
            {
                const int ty0 = it.ty * TH, tx0 = it.tx * TW;
                const unsigned ob = (unsigned)((((it.b * Cx) * H + ty0 + 2 * wave) * W + tx0) * 4);
                const bool full = ty0 + TH <= H && tx0 + TW <= W;
                bool px_ok = true;
                if (!full) px_ok = (ty0 + 2 * wave + (lb >> 3) < H) && (tx0 + 4 * (lb & 7) < W);
#pragma unroll
                for (int n = 0; n < Q4_NG; ++n) {
                    so[n] = ob + (unsigned)n * nstep;
                    ch_ok[n] = px_ok && (4 * n + lj < Cx);
                    vo[n] = ch_ok[n] ? ovoff : kOOB;
                }
            }
            // ---- data gradient: K loop over (g channel, tap row); the 3 A values of the next tap row are read while 9 MFMAs issue.
            // The 84 weight registers are re-read from LDS per tile (84 ds_read_b32 against the tile's 236): resident across the weight
            // gradient's phase they would push the DSIN form (36 auxiliary registers in flight) past 256 VGPRs into scratch
            {
#ifdef BNERV_FOLD_RELOAD
                load_weights();
#endif
#ifdef BNERV_PAIRF_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
                float a_cur[3], a_nxt[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) a_cur[kx] = a_base[kx];
#pragma unroll
                for (int ci = 0; ci < Q4_NCH; ++ci) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int nci = ky == 2 ? ci + 1 : ci, nky = ky == 2 ? 0 : ky + 1;
                        if (nci < Q4_NCH) {
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a_nxt[kx] = a_base[nci * GPL + nky * GQ::RS + kx];
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
            // this tile's x planes were requested behind barrier (A) of the previous tile (prologue: before the loop): everything older than the
            // NPRE gradient DMAs issued at the top of this tile is complete
            if (has_next) q4_wait_dma<NPRE>(); else q4_wait_dma<0>();
            // the epilogue's auxiliary tensors that are NOT the weight gradient's input, into registers: they land under its K loop
            {
                if constexpr (EP == BNERV_EP_DGELU_SAVED) {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) ax0[n] = bnerv_conv::bload(ra0, vo[n], so[n]);
                } else {
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) ax1[n] = bnerv_conv::bload(ra1, vo[n], so[n]);
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) ax2[n] = d.aux2 ? bnerv_conv::bload(ra2, vo[n], so[n]) : f32x4{1.f, 1.f, 1.f, 1.f};
                }
            }
#ifdef BNERV_PAIRF_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            FTRACE(1);
            lds_barrier();                                 // (B) this tile's x planes and mask plane are in LDS for every wave
            FTRACE(2);
            // ---- weight gradient: 16 K steps of 4 pixels, eight N tiles (R and M columns)
            {
#ifdef BNERV_PAIRF_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    float bf[NTW];
                    const float af = s_gc[abase + (st >> 3) * GQ::RS + (st & 7) * 4];
#pragma unroll
                    for (int n = 0; n < NTW; ++n) bf[n] = s_x[bbase[n] + ((st >> 3) * GW::RS + (st & 7) * 4)];
#pragma unroll
                    for (int n = 0; n < NTW; ++n) accw[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], accw[n], 0, 0, 0);
                }
            }
#ifdef BNERV_PAIRF_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            FTRACE(3);
            // ---- data-gradient epilogue straight from the accumulators (conv4_body.h)
            {
                if constexpr (RED) {
                    if (it.b != ep_b) {
#pragma unroll
                        for (int n = 0; n < Q4_NG; ++n) scl[n] = 4 * n + lj < Cx ? 1.0f + d.scale[it.b * Cx + 4 * n + lj] : 0.f;
                        ep_b = it.b;
                    }
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) { if (!ch_ok[n]) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
                {
                    // the weight gradient's RAW input at this lane's pixels (channel 4 n + lj): aux0 of DSIN, aux1 of DGELU_SAVED
                    f32x4 xr[Q4_NG];
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n)
                        xr[n] = *reinterpret_cast<const f32x4*>(s_x + (4 * n + lj) * GW::PLANE + (2 * wave + (lb >> 3) + GQ::PAD) * GW::RS + GQ::XOFF + 4 * (lb & 7));
                    float ps[Q4_NG], pt[Q4_NG];
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) {
                        f32x4 r;
                        const f32x4 v = acc[n];
                        ps[n] = 0.f; pt[n] = 0.f;
                        if constexpr (EP == BNERV_EP_DGELU_SAVED) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl[n] * ax0[n][e]; ps[n] = fmaf(v[e], xr[n][e], ps[n]); pt[n] += v[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = (ax1[n][e] + v[e] * scl[n]) * ax2[n][e]; ps[n] = fmaf(v[e], xr[n][e], ps[n]); pt[n] += v[e]; }
                        }
                        bnerv_conv::bstore(ro, vo[n], so[n], r);
                    }
#pragma unroll
                    for (int n = 0; n < Q4_NG; ++n) { rs[n] += ps[n]; rt[n] += pt[n]; }
                    const bool flush = !has_next || nxt.b != it.b;          // block-uniform
                    if (!flush) {
                        write_row(it, false);
                    } else {
#pragma unroll
                        for (int n = 0; n < Q4_NG; ++n) {
#pragma unroll
                            for (int off = 4; off < 64; off <<= 1) {
                                rs[n] += __shfl_xor(rs[n], off, 64);
                                rt[n] += __shfl_xor(rt[n], off, 64);
                            }
                        }
                        if (lane < 4) {
#pragma unroll
                            for (int n = 0; n < Q4_NG; ++n) { s_red[(wave * 2 + 0) * 16 + 4 * n + lane] = rs[n]; s_red[(wave * 2 + 1) * 16 + 4 * n + lane] = rt[n]; }
                        }
                        lds_barrier();
                        write_row(it, true);
#pragma unroll
                        for (int n = 0; n < Q4_NG; ++n) { rs[n] = 0.f; rt[n] = 0.f; }
                    }
                }
            }
            FTRACE(4);
            if (has_next) {
                // everything but this epilogue's Q4_NG stores is complete: the next g tile has landed
                q4_wait_dma<Q4_NG>();
                FTRACE(5);
                lds_barrier();                             // (A) next g landed in every wave; every wave is done with this tile's g, x, mask; s_red is visible
                FTRACE(6);
                set_mask(nxt);
                issue_xd(nxt);                             // lands under the next tile's data-gradient K loop
                buf ^= 1;
                s_gc = s_g2 + buf * S_IN;
                a_base = s_gc + (2 * wave + (lane >> 5)) * GQ::RS + (lane & 31) + GQ::COL0;
            }
            FTRACE(7);
            it = nxt;
        }
    }

    // ---- the block's weight-gradient slab: cross-wave reduction through LDS (fixed order => deterministic)
    __syncthreads();
    float* s_sum = smem;
    constexpr int RW = NTW * 16, RSZ = 16 * RW;
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_sum[wave * RSZ + (4 * kq + r) * RW + n * 16 + li] = accw[n][r];
    __syncthreads();
    const int fold_nc = nW + 9;                            // slab row: Cx * 9 R columns + 9 M columns (= wa.ncols + 8)
    float* slab = wa.slab + (size_t)vb * Cg * fold_nc;
    for (int idx = tid; idx < RSZ; idx += 256) {
        const int row = idx / RW, col = idx - row * RW;
        if (row < Cg && col < fold_nc)
            slab[(size_t)row * fold_nc + col] = (s_sum[idx] + s_sum[RSZ + idx]) + (s_sum[2 * RSZ + idx] + s_sum[3 * RSZ + idx]);
    }
    side_run_hosted(side, smem, vb, vgrid);
}

inline size_t pair_fold_lds_bytes() {
    using GQ = bnerv_conv::Geo<3>;
    using GW = Geo<3>;
    constexpr int NSLOT = bnerv_q4::Q4_NCH * (GQ::ROWS * GQ::SEGS + 1);
    constexpr int NPRE = (NSLOT + 255) / 256;
    constexpr int NXS = (12 * (GW::PLANE / 4) + 255) / 256;
    return ((size_t)2 * NPRE * 256 * 4 + (size_t)bnerv_q4::Q4_NCH * bnerv_q4::Q4_QPAD * 4 + 256 + (size_t)NXS * 256 * 4 + 2 * GW::PLANE + 64) * sizeof(float);
}
inline size_t pair_fused_lds_bytes() {
    using GQ = bnerv_conv::Geo<3>;
    using GW = Geo<3>;
    constexpr int NSLOT = bnerv_q4::Q4_NCH * (GQ::ROWS * GQ::SEGS + 1);
    constexpr int NPRE = (NSLOT + 255) / 256;
    return ((size_t)2 * NPRE * 256 * 4 + (size_t)bnerv_q4::Q4_NCH * bnerv_q4::Q4_QPAD * 4 + 256 + (size_t)(12 + 2) * GW::PLANE + 64 * 4 + 32 + 64) * sizeof(float);
}
