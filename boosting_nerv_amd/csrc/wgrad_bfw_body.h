// wgrad_bfw_body.h -- the split-precision wide weight-gradient kernel of wgrad.hip as a device function over a virtual block index, so
// that paired launches (wgrad.hip: next to convs.hip's data gradient; convbf.hip: next to the wide split conv's) can run it as one role
// of their grid.  The design notes are below, where they always were.
#pragma once
#include "common.h"
#include "sidejob.h"
#include "split16.h"
#include <stdlib.h>
#include <string.h>
#ifndef BTRACE
#define BTRACE(it_, slot) do {} while (0)
#endif

namespace bnerv_wb {

struct WArgs {
    bnerv_wgrad_desc d;
    float* slab;
    int tiles_x, tiles_y, n_mgroups, n_ngroups, ncols;   // ncols = Cin*T + 1
    int vec;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned shift_bytes, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) - shift_bytes), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

struct LTile { int b, ty, tx; };

// ---------------------------------------------------------------------------------------------------------------- split wide kernel
// The weight gradient of the 3x3 stride-1 layers with more than 16 output or 12 input channels (plain gradient: the TAT convs and the
// stride-1 block convs of the 22..95-channel stages) on the 16-bit matrix pipe with f32 operands split into bf16 pieces (split16.h;
// convbf.hip explains the scheme: bf16x6 is exact to below the f32 MFMA's own rounding).  K = pixels, 32 per MFMA:
//   * tile = 4 rows x 32 px; wave w owns row w (one K step per tile); block = MTW cout tiles x 8 column tiles (128 (ci, tap) columns);
//   * A = the gradient: lane (cout, kq) holds 8 pixels of ONE channel -- loaded straight from global into registers (two 16-B loads
//     of the 128-B row segment shared by the 4 kq lanes), split there: no LDS for g, no staging redundancy across waves;
//   * B = the input window: 8 consecutive pixels shifted by (ky - 1, kx - 1).  A 16-bit vector read must be 16-B aligned, so LDS holds
//     THREE copies of the (prologue-transformed, split) input tile, one per kx, each stored pre-shifted: copy kx at position x is
//     a[x + kx - 1]; the ky shift is a whole row.  [piece][kx][channel][6 rows], padded against bank conflicts (BW_* below);
//   * the bias gradient is the column after the last weight column: its B operand reads a row of ones; columns beyond read zeros;
//   * accumulators stay in registers over the block's tiles; the waves add theirs one after the other into ONE LDS area, the block
//     writes its 128 columns of one slab -- slabs, finish and deferred reduction exactly as the f32 wide kernel.
constexpr int BW_TH = 4, BW_NTW = 8, BW_NPL = (BW_NTW * 16 + 7) / 9 + 1;      // 16 channels span 128 consecutive columns
// LDS image of one piece (bytes): rows of 32 px x 2 B, 6 rows + 32 B per channel plane, the three kx copies 16 planes + 192 B apart.
// With these strides the 16 lanes of a ds_read_b128 service group (16 consecutive (ci, ky, kx) columns, two kq) spread over the 16
// 16-B slots of the bank row almost evenly (exhaustive search over paddings <= this size: 1.9 accesses per slot-cycle on average
// against 3.0 for the unpadded image; the only conflict-free layout needs 800-B planes, i.e. one block per CU).
constexpr int BW_ROW = 64, BW_PLANE = 6 * BW_ROW + 32, BW_COPY = BW_NPL * BW_PLANE + 192, BW_CONST = 4 * BW_ROW;
constexpr int BW_PIECE = 3 * BW_COPY + 2 * BW_CONST;                          // + a plane of ones (piece 0 only) and a plane of zeros

// GM2: 0 = g as is, 1 = g is the pixel-shuffled (x2) gradient of an up-conv (conv channel 4c + 2i + j at (y, x) = du[c][2y + i][2x + j]),
// 3 = shuffled by s = g_s in {3, 5} (conv channel c s^2 + i s + j at (y, x) = du[c][s y + i][s x + j]; strided 4-B loads)
template <int IN, int SP, int MTW, int GM2>
__device__ __forceinline__ void wgrad_bfw_body(const WArgs& wa, const int slots, const int ngroups_n, const int ngroups_m, const SidePack& side, const int vb, const int vgrid) {
    constexpr int NS = Split<SP>::NS;
    constexpr bool AFF = (IN == BNERV_IN_AFFINE);
    constexpr int PIECE = BW_PIECE;
    const bnerv_wgrad_desc& d = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* s_a = reinterpret_cast<char*>(smem);                                // [NS][PIECE]
    float* s_aff = reinterpret_cast<float*>(s_a + NS * PIECE);                // [2][BW_NPL]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int nW = Cin * 9;
    const int tiles_x = (W + 31) >> 5, tiles_y = (H + BW_TH - 1) / BW_TH;

    const int ngroups = ngroups_n * ngroups_m;
    const int xcd = vb & 7, q = vb >> 3;
    const int slot = q / ngroups, grp = q - slot * ngroups;
    const int mg = grp / ngroups_n;
    const int co_base = mg * MTW * 16;
    const int n_base = (grp - mg * ngroups_n) * BW_NTW * 16;
    const int ci_lo = min(n_base, nW - 1) / 9, ci_hi = min(n_base + BW_NTW * 16 - 1, nW - 1) / 9;
    const int npl = ci_hi - ci_lo + 1;
    const int total = d.B * tiles_x * tiles_y;
    const int per = total >> 3, extra = total & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);

    // constant planes: ones (bf16 1.0 in piece 0, zero in the other pieces) and zeros
    for (int i = tid; i < NS * 2 * BW_CONST / 4; i += 256) {
        const int p = i / (2 * BW_CONST / 4), w = i - p * (2 * BW_CONST / 4);
        reinterpret_cast<unsigned*>(s_a + p * PIECE + 3 * BW_COPY)[w] = (p == 0 && w < BW_CONST / 4) ? 0x3f803f80u : 0u;
    }

    // staging slots: (channel, row, 4-px segment) -> the aligned float4 at x = tx0 + 4 sg plus its two neighbours x - 1 and x + 4.
    // The six values are split ONCE as the pairs (v0, v1) (v2, v3) (L, R): copy kx = 1 is the first two pairs as they are, copies
    // kx = 0 (a[x - 1]) and kx = 2 (a[x + 1]) share the middle pair (v1, v2) = one alignbit and take one byte-permute each.
    // Addresses: per-lane offsets are constants of the block, the tile enters through the scalar offset of the buffer loads; only the
    // tiles on the image border (a few per cent) compute per-slot validity.
    constexpr int NXS = BW_NPL * 6 * 8 / 256;                                 // = 3 slots per thread, no idle ones
    static_assert(BW_NPL * 6 * 8 == NXS * 256, "staging slots must fill the block");
    const unsigned x_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4);
    const unsigned g_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, 0, x_bytes);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(d.g, 0, g_bytes);
    int xs_c[NXS], xs_r[NXS], xs_lds[NXS];
    unsigned xs_off[NXS], gs_off[MTW];
    const int xs_sg = tid & 7;
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
        const int sidx = tid + k * 256;
        xs_r[k] = (sidx >> 3) % 6;
        xs_c[k] = (sidx >> 3) / 6;
        xs_lds[k] = xs_c[k] * BW_PLANE + xs_r[k] * BW_ROW + (xs_sg & 3) * 16 + (xs_sg >> 2) * 8;   // K order of the fragments, see gs_off
        xs_off[k] = xs_c[k] < npl ? (unsigned)(((xs_c[k] * H + xs_r[k]) * W + 4 * xs_sg) * 4) : OOB;
    }
    // K order inside a 32-px step: lane kq holds pixels 4 kq .. 4 kq + 3 and 16 + 4 kq .. 16 + 4 kq + 3, so that each of the two loads
    // of a fragment reads 64 contiguous bytes per channel row (the LDS rows of the input are stored in the same order).
    // Shuffled gradient (GM2): the lane pair (j = 0, 1) of conv channels 4c + 2i + j shares 2 x 8 consecutive floats of row 2y + i of
    // du[c] -- lane j loads the block of pixel group j (32 B), keeps its own parity and swaps the other one with its partner (DPP).
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int cl = 16 * m + li;
        if constexpr (GM2 == 1)
            gs_off[m] = co_base + cl < Cout ? (unsigned)(((((cl >> 2) * 2 * H) + 2 * wave + ((cl >> 1) & 1)) * 2 * W + 8 * kq + 32 * (cl & 1)) * 4) : OOB;
        else if constexpr (GM2 == 3) {                     // x3 / x5 shuffle: absolute shuffled channel, row s (y) + i, column s (4 kq) + j
            const int sg_ = d.g_s, co = co_base + cl, cs = sg_ == 3 ? co / 9 : co / 25, rs = co - cs * sg_ * sg_, is = sg_ == 3 ? rs / 3 : rs / 5;
            gs_off[m] = co < Cout ? (unsigned)((((cs * sg_ * H) + sg_ * wave + is) * sg_ * W + (rs - is * sg_) + sg_ * 4 * kq) * 4) : OOB;
        } else
            gs_off[m] = co_base + cl < Cout ? (unsigned)(((cl * H + wave) * W + 4 * kq) * 4) : OOB;
    }
    f32x4 xv[NXS];
    float xl[NXS], xr[NXS];
    f32x4 ga[MTW][2];                                                         // gradient fragments of this wave's row: lane (cout co_base + 16 m + li,
                                                                              // pixels 8 kq .. 8 kq + 7)
    unsigned xmask = 0;                                                       // border tiles: bit 3k: slot live, 3k + 1: left neighbour, 3k + 2: right
    auto tile_interior = [&](const LTile& a) __attribute__((always_inline)) {
        return a.ty > 0 && a.ty * BW_TH + BW_TH + 1 <= H && a.tx > 0 && a.tx * 32 + 32 < W;
    };
    auto issue_loads = [&](const LTile& a) __attribute__((always_inline)) {
        const int ty0 = a.ty * BW_TH, tx0 = a.tx * 32;
        if (tile_interior(a)) {
            const unsigned sx = (unsigned)((((a.b * Cin + ci_lo) * H + ty0 - 1) * W + tx0) * 4);
            if constexpr (GM2 == 3) {                          // four 4-B loads s columns apart per pixel group
                const int sg_ = d.g_s;
                const unsigned sgb = (unsigned)(((((a.b * (Cout / (sg_ * sg_))) * sg_ * H) + sg_ * ty0) * sg_ * W + sg_ * tx0) * 4);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            ga[m][h][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (int)gs_off[m], (int)(sgb + (unsigned)(sg_ * (16 * h + t) * 4)), 0));
            } else {
            const unsigned sg = GM2 == 1 ? (unsigned)(((((a.b * (Cout >> 2) + (co_base >> 2)) * 2 * H) + 2 * ty0) * 2 * W + 2 * tx0) * 4)
                                         : (unsigned)((((a.b * Cout + co_base) * H + ty0) * W + tx0) * 4);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                ga[m][0] = bload(rg, gs_off[m], sg);
                ga[m][1] = bload(rg, gs_off[m], sg + (GM2 == 1 ? 16u : 64u));
            }
            }
#pragma unroll
            for (int k = 0; k < NXS; ++k) {
                xv[k] = bload(rx, xs_off[k], sx);
                xl[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)xs_off[k], (int)(sx - 4u), 0));
                xr[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)xs_off[k], (int)(sx + 16u), 0));
            }
            return;
        }
        {
            const int gy = ty0 + wave, gx = tx0 + 4 * kq;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int co = co_base + 16 * m + li;
                const bool ok = co < Cout && gy < H;
                if constexpr (GM2 == 3) {
                    const int sg_ = d.g_s, cs = sg_ == 3 ? co / 9 : co / 25, rs = co - cs * sg_ * sg_, is = sg_ == 3 ? rs / 3 : rs / 5;
                    const unsigned base = (unsigned)(((((a.b * (Cout / (sg_ * sg_)) + cs) * sg_ * H) + sg_ * gy + is) * sg_ * W + sg_ * gx + (rs - is * sg_)) * 4);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            ga[m][h][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (int)((ok && gx + 16 * h < W) ? base + (unsigned)(sg_ * (16 * h + t) * 4) : OOB), 0, 0));
                } else if constexpr (GM2 == 1) {
                    const int px0 = gx + 16 * (co & 1);
                    const unsigned base = (unsigned)(((((a.b * (Cout >> 2) + (co >> 2)) * 2 * H) + 2 * gy + ((co >> 1) & 1)) * 2 * W + 2 * px0) * 4);
                    ga[m][0] = bload(rg, (ok && px0 < W) ? base : OOB, 0u);
                    ga[m][1] = bload(rg, (ok && px0 < W) ? base + 16u : OOB, 0u);
                } else {
                    const unsigned base = (unsigned)((((a.b * Cout + co) * H + gy) * W + gx) * 4);
                    ga[m][0] = bload(rg, (ok && gx < W) ? base : OOB, 0u);
                    ga[m][1] = bload(rg, (ok && gx + 16 < W) ? base + 64u : OOB, 0u);
                }
            }
        }
        const int gx = tx0 + 4 * xs_sg;
        xmask = 0;
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            const int gy = ty0 + xs_r[k] - 1;
            const bool live = xs_c[k] < npl && (unsigned)gy < (unsigned)H && gx < W;
            const bool hl = live && gx > 0, hr = live && gx + 4 < W;
            const unsigned vo = (unsigned)((((a.b * Cin + ci_lo + xs_c[k]) * H + gy) * W + gx) * 4);
            xv[k] = bload(rx, live ? vo : OOB, 0u);
            xl[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(hl ? vo - 4u : OOB), 0, 0));
            xr[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(hr ? vo + 16u : OOB), 0, 0));
            xmask |= (live ? 1u : 0u) << (3 * k) | (hl ? 2u : 0u) << (3 * k) | (hr ? 4u : 0u) << (3 * k);
        }
    };
    float a_sc[NXS], a_sh[NXS];                                               // this thread's slots' affine parameters (per batch item)
    auto fetch_affine = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NXS; ++k) { a_sc[k] = s_aff[xs_c[k]]; a_sh[k] = s_aff[BW_NPL + xs_c[k]]; }
    };
    auto store_x = [&](const bool interior) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            float x[8] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w, xl[k], xr[k], 0.f, 0.f};
            if constexpr (AFF) {                                              // zero padding applies AFTER the prologue; dead channels carry
                const float sc = a_sc[k], sh = a_sh[k];                       // scale = shift = 0, so an interior tile needs no mask at all
#pragma unroll
                for (int e = 0; e < 6; ++e) x[e] = x[e] * sc + sh;
                if (!interior) {
                    const bool live = (xmask >> (3 * k)) & 1u, hl = (xmask >> (3 * k)) & 2u, hr = (xmask >> (3 * k)) & 4u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = live ? x[e] : 0.f;
                    x[4] = hl ? x[4] : 0.f;
                    x[5] = hr ? x[5] : 0.f;
                }
            }
            u32x4 pc[NS];
            split8<SP, 6>(x, pc);
            char* dst = s_a + xs_lds[k];
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                const unsigned mid = __builtin_amdgcn_alignbit(pc[p][1], pc[p][0], 16);                  // (v1, v2)
                const unsigned lft = __builtin_amdgcn_perm(pc[p][0], pc[p][2], 0x05040100u);            // (L, v0)
                const unsigned rgt = __builtin_amdgcn_perm(pc[p][2], pc[p][1], 0x07060302u);            // (v3, R)
                *reinterpret_cast<uint2*>(dst + p * PIECE) = uint2{lft, mid};
                *reinterpret_cast<uint2*>(dst + p * PIECE + BW_COPY) = uint2{pc[p][0], pc[p][1]};
                *reinterpret_cast<uint2*>(dst + p * PIECE + 2 * BW_COPY) = uint2{mid, rgt};
            }
        }
    };
    auto load_affine = [&](int b) __attribute__((always_inline)) {
        if (tid < 2 * BW_NPL) {
            const int c = tid % BW_NPL;
            float v = 0.f;
            if (c < npl) v = tid < BW_NPL ? 1.0f + d.scale[b * Cin + ci_lo + c] : d.shift[b * Cin + ci_lo + c];
            s_aff[tid] = v;
        }
    };

    int bbase[BW_NTW];
#pragma unroll
    for (int nt = 0; nt < BW_NTW; ++nt) {
        const int n = n_base + nt * 16 + li;
        int off;
        if (n < nW) {
            const int ci = n / 9, tap = n - ci * 9, ky = tap / 3, kx = tap - ky * 3;
            off = kx * BW_COPY + (ci - ci_lo) * BW_PLANE + ky * BW_ROW;
        } else {
            off = 3 * BW_COPY + (n == nW ? 0 : BW_CONST);                     // ones / zeros
        }
        bbase[nt] = off + wave * BW_ROW + kq * 16;
    }

    f32x4 acc[MTW][BW_NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < BW_NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int itx = r0 + slot;
    LTile it{0, 0, 0};
    auto decode = [&](int i) __attribute__((always_inline)) {
        LTile t;
        const int tiles = tiles_x * tiles_y;
        t.b = i / tiles;
        const int rem = i - t.b * tiles;
        t.ty = rem / tiles_x;
        t.tx = rem - t.ty * tiles_x;
        return t;
    };
    int aff_b = -1;
    if (itx < r1) {
        it = decode(itx);
        issue_loads(it);
        if constexpr (AFF) { load_affine(it.b); aff_b = it.b; }
        lds_barrier();
        if constexpr (AFF) fetch_affine();
        store_x(tile_interior(it));
    }
    int bt = -3;                                          // (trace: tiles 3..9 of the block)
    for (; itx < r1; itx += slots, ++bt) {
        BTRACE(bt, 0);
        const bool has_next = itx + slots < r1;
        LTile nxt = it;
        if (has_next) nxt = decode(itx + slots);
        // split this wave's gradient row (registers only), then prefetch the next tile's gradient and input
        u32x4 afr[MTW][NS];
        auto split_g = [&](auto mc) __attribute__((always_inline)) {          // (explicit per-m instances: a loop with DPP moves is not unrolled)
            constexpr int m = decltype(mc)::value;
            if constexpr (m < MTW) {
                float x[8] = {ga[m][0].x, ga[m][0].y, ga[m][0].z, ga[m][0].w, ga[m][1].x, ga[m][1].y, ga[m][1].z, ga[m][1].w};
                if constexpr (GM2 == 1) {
                    // own block f0..f7: even floats belong to channel j = 0, odd ones to j = 1.  The j = 0 lane holds pixel group 0, the
                    // j = 1 lane group 1; each keeps its parity of its own block and gets its parity of the partner's block.
                    const bool odd = li & 1;
                    const float k0 = odd ? x[1] : x[0], k1 = odd ? x[3] : x[2], k2 = odd ? x[5] : x[4], k3 = odd ? x[7] : x[6];
                    const float s0 = odd ? x[0] : x[1], s1 = odd ? x[2] : x[3], s2 = odd ? x[4] : x[5], s3 = odd ? x[6] : x[7];
                    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
                    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
                    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0xB1, 0xF, 0xF, false));
                    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s3), 0xB1, 0xF, 0xF, false));
                    x[0] = odd ? r0 : k0; x[1] = odd ? r1 : k1; x[2] = odd ? r2 : k2; x[3] = odd ? r3 : k3;
                    x[4] = odd ? k0 : r0; x[5] = odd ? k1 : r1; x[6] = odd ? k2 : r2; x[7] = odd ? k3 : r3;
                }
                split8<SP, 8>(x, afr[m]);
            }
        };
        split_g(std::integral_constant<int, 0>{});
        split_g(std::integral_constant<int, 1>{});
        split_g(std::integral_constant<int, 2>{});
        BTRACE(bt, 1);
        if (has_next) issue_loads(nxt);
        BTRACE(bt, 2);
        lds_barrier();                                     // (A) the input copies of this tile are in LDS
        BTRACE(bt, 3);
        // (reading the B fragments one column tile ahead through a second register set was measured: no gain -- the other resident
        //  block covers the LDS latency -- and it costs the MTW = 3 variants their last registers)
#pragma unroll
        for (int nt = 0; nt < BW_NTW; ++nt) {
            u32x4 bfr[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) bfr[p] = *reinterpret_cast<const u32x4*>(s_a + p * PIECE + bbase[nt]);
#define BNERV_BW_PROD(pa, pb) _Pragma("unroll") for (int m = 0; m < MTW; ++m) acc[m][nt] = mfma16<SP>(afr[m][pa], bfr[pb], acc[m][nt]);
            if constexpr (NS == 3) {
                BNERV_BW_PROD(2, 0)
                BNERV_BW_PROD(0, 2)
                BNERV_BW_PROD(1, 1)
            }
            BNERV_BW_PROD(1, 0)
            BNERV_BW_PROD(0, 1)
            BNERV_BW_PROD(0, 0)
#undef BNERV_BW_PROD
        }
        BTRACE(bt, 4);
        lds_barrier();                                     // (B) everyone done reading
        BTRACE(bt, 5);
        if (has_next) {
            if constexpr (AFF) { if (nxt.b != aff_b) { load_affine(nxt.b); lds_barrier(); fetch_affine(); aff_b = nxt.b; } }
            if (tile_interior(nxt)) store_x(true);
            else store_x(false);
        }
        BTRACE(bt, 6);
        it = nxt;
    }

    // cross-wave reduction, one wave after the other into one area (fixed order), then this block's columns of the slot's slab
    __syncthreads();
    float* s_red = smem;
    constexpr int RW = BW_NTW * 16, RSZ = MTW * 16 * RW;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < BW_NTW; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // D fragment: column (N) = li, row (M) = 4 kq + r  ->  cout 16 m + 4 kq + r, column 16 n + li
                        float* qd = s_red + (m * 16 + 4 * kq + r) * RW + n * 16 + li;
                        *qd = (w == 0) ? acc[m][n][r] : *qd + acc[m][n][r];
                    }
        }
        __syncthreads();
    }
    float* slab = wa.slab + (size_t)(xcd * slots + slot) * Cout * wa.ncols;
    for (int idx = tid; idx < RSZ; idx += 256) {
        const int row = idx / RW, colq = idx - row * RW;
        const int col = n_base + colq;
        if (co_base + row < Cout && col < wa.ncols) slab[(size_t)(co_base + row) * wa.ncols + col] = s_red[idx];
    }
    side_run_hosted(side, smem, vb, vgrid);
}

struct BwPlan { int mtw, ngroups_n, ngroups_m, slots; };
static int bw_mode() {                                     // BNERV_SPLIT_WIDE = bf16x6 (default) | bf16x3 | off   (shared with convbf.hip)
    static const int v = [] {
        const char* e = getenv("BNERV_SPLIT_WIDE");
        if (!e) return (int)SP_BF16X6;
        if (!strcmp(e, "off") || !strcmp(e, "0")) return -1;
        if (!strcmp(e, "bf16x3")) return (int)SP_BF16X3;
        return (int)SP_BF16X6;
    }();
    return v;
}

}  // namespace bnerv_wb
