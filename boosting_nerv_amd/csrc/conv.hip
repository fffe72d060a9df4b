// conv.hip -- fp32 implicit-GEMM convolution on MFMA 16x16x4 (gfx950) with fused prologues / epilogues.
//
// One kernel family serves the forward convolution and the data gradient of every conv on the Boosting-NeRV decoder
// path (reference call sites: lib/quant_ops.py:39-41 via model_blocks.py:74-89, :196-220, model_enerv.py:73-102,
// model_nerv.py:41,56).  GEMM view, per sample:
//     D[pixel][cout] = sum_{tap, cin} A[pixel][(tap,cin)] * Wt[(tap,cin)][cout]
// M = pixels (16 consecutive x per MFMA tile), N = cout (16 per tile), K = taps*Cin walked 4 at a time.
//
// PERSISTENT blocks (256 threads = 4 waves) walk work items (cout-group, sample, 8x32 spatial tile); each XCD owns a
// contiguous range of items so neighbouring tiles (shared halo rows) meet in one L2.
//   LDS  s_in  [16 ch][ROWS][RS]   channel-planar halo tile.  RS = 40 (3x3): columns cover x0-4 .. x0+35 so that every row is
//                                  10 ALIGNED float4 segments (one global_load_dwordx4 + one ds_write_b128 each); plane stride
//                                  400 == 16 (mod 32) -> the four k-lanes of an A-fragment ds_read_b32 hit disjoint bank halves.
//        s_w   [tap][q][n][64]     B fragments in lane order; RESIDENT for the whole block when the layer's weights fit
//                                  (every C<=16 layer, i.e. all the high-resolution stages), restaged per K-chunk otherwise.
//        s_out [16 cout][8x32 +4]  one cout-tile of accumulators, re-laid so the copy-out walks the FINAL layout with float4
//                                  stores / float4 aux reads (pixel-shuffle x2 included).
// Software pipeline (single K-chunk layers): the global loads of tile t+1 are issued into registers BEFORE the MFMA phase
// of tile t and written to LDS after it, so HBM/L2 latency hides under the matrix pipe.
// Each wave owns 2 rows x 32 px = 4 M-tiles; acc[4][NTB] (f32x4) stays in registers for the whole K loop.
#include "common.h"
#include "sidejob.h"
#include "conv_common.h"

int bnerv_convbf_try(hipStream_t st, const bnerv_conv_desc& d, int vec, int ksplit, int chunks_per_split);   // convbf.hip
namespace bnerv_conv { struct KArgs; }
int bnerv_conv4_try(hipStream_t st, bnerv_conv::KArgs& ka);   // conv4.hip: 1 = not that family's layer
int bnerv_convs_try(hipStream_t st, const bnerv_conv_desc& d, int vec, int ksplit);   // convs.hip (low-resolution stages): 1 = not that family's layer
int bnerv_head3_try(hipStream_t st, const bnerv_conv_desc& d);            // head3.hip (3x3 head with 3 outputs: forward + tanh, data gradient): 1 = not that layer
bool bnerv_convs_shape_ok(const bnerv_conv_desc& d, int vec);
int bnerv_stem_dgrad_try(hipStream_t st, const bnerv_conv_desc& d);       // stem.hip (images of <= 256 pixels, long K): 1 = not that layer
size_t bnerv_stem_dgrad_ws_bytes(const bnerv_conv_desc& d);
int bnerv_convs_tiles(int H, int W);

namespace {
using namespace bnerv_conv;

#ifndef BNERV_ABL
#define BNERV_ABL 0   // debug ablations of conv_fast_kernel (never shipped): 1 no MFMA, 2 no global ld/st, 3 loads only, 4 stores only
#endif
constexpr bool ABL_NO_MFMA = BNERV_ABL == 1 || BNERV_ABL == 3 || BNERV_ABL == 4;
constexpr bool ABL_NO_LOAD = BNERV_ABL == 2 || BNERV_ABL == 4;
constexpr bool ABL_NO_STORE = BNERV_ABL == 2 || BNERV_ABL == 3;
struct Item { int g, b, ty0, tx0, tile, ks; };

__device__ __forceinline__ Item decode_item(const KArgs& ka, int it) {
    const int tiles = ka.tiles_x * ka.tiles_y;
    Item r;
    r.ks = 0;
    if (ka.ksplit > 1) { r.ks = it % ka.ksplit; it /= ka.ksplit; }
    r.tile = it % tiles;
    const int rest = it / tiles;
    r.b = rest % ka.d.B;
    r.g = rest / ka.d.B;
    r.ty0 = (r.tile / ka.tiles_x) * TH;
    r.tx0 = (r.tile % ka.tiles_x) * TW;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------- staging
template <int IN>
__device__ __forceinline__ float load_in_scalar(const bnerv_conv_desc& d, int b, int ci, int gy, int gx) {
    if constexpr (IN == BNERV_IN_UNSHUFFLE) {
        const int s = d.in_s, s2 = s * s;
        const int c = ci / s2, rem = ci - c * s2, i = rem / s, j = rem - i * s;
        return d.x[(((size_t)b * (d.Cin / s2) + c) * (size_t)(d.H * s) + (size_t)(gy * s + i)) * (size_t)(d.W * s) + (size_t)(gx * s + j)];
    } else {
        const size_t idx = (((size_t)b * d.Cin + ci) * d.H + gy) * (size_t)d.W + gx;
        float sc = 1.f, sh = 0.f, aux = 0.f;
        if constexpr (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE) { sc = 1.0f + d.scale[b * d.Cin + ci]; sh = d.shift[b * d.Cin + ci]; }
        if constexpr (IN == BNERV_IN_TANHGRAD) aux = d.aux0[idx];
        return xform1<IN>(d.x[idx], sc, sh, aux);
    }
}

// generic (any W, any shuffle factor): one float per step, zero outside the image / beyond Cin
template <int KS, int IN>
__device__ __forceinline__ void stage_scalar(const bnerv_conv_desc& d, float* s_in, int b, int c0, int nch, int ty0, int tx0) {
    using G = Geo<KS>;
    // loads are issued in batches of 8 before the LDS stores, otherwise every element costs a full memory latency
    const int n_el = nch * G::PLANE_RAW;
    for (int i0 = threadIdx.x; i0 < n_el; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + u * 256;
            const int c = idx / G::PLANE_RAW;
            const int rem = idx - c * G::PLANE_RAW;
            const int r = rem / G::RS, col = rem - r * G::RS;
            const int gy = ty0 + r - G::PAD, gx = tx0 + col - G::XOFF, ci = c0 + c;
            v[u] = 0.f;
            if (idx < n_el && ci < d.Cin && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W) v[u] = load_in_scalar<IN>(d, b, ci, gy, gx);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + u * 256;
            const int c = idx / G::PLANE_RAW;
            const int rem = idx - c * G::PLANE_RAW;
            const int r = rem / G::RS, col = rem - r * G::RS;
            if (idx < n_el) s_in[c * G::PLANE + r * G::RS + col] = v[u];
        }
    }
}

// vector staging, split in two so the loads of the NEXT tile can fly under the MFMA phase of the current one.
// PLAIN/AFFINE/GELU_AFFINE/TANHGRAD: slot = (channel, row, 4-px segment): one float4 (two for TANHGRAD).
// UNSHUFFLE with in_s == 2: slot = (channel PAIR (c,i,j=0/1), row, segment): two float4 of the shuffled source row.
template <int KS, int IN>
struct VecStage {
    using G = Geo<KS>;
    static constexpr bool PAIR = (IN == BNERV_IN_UNSHUFFLE);
    static constexpr bool TWO = PAIR || (IN == BNERV_IN_TANHGRAD);
    static constexpr int NPRE = PAIR ? (G::SLOTS / 2 + 255) / 256 : G::NPRE;
    f32x4 ra[NPRE], rb[TWO ? NPRE : 1];

    __device__ __forceinline__ int nslots(int nch) const { return (PAIR ? nch / 2 : nch) * G::ROWS * G::SEGS; }

    __device__ __forceinline__ void issue(const bnerv_conv_desc& d, int b, int c0, int nch, int ty0, int tx0) {
        const int ns = nslots(nch);
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int sidx = threadIdx.x + k * 256;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (sidx < ns) {
                const int c = sidx / (G::ROWS * G::SEGS);
                const int rem = sidx - c * (G::ROWS * G::SEGS);
                const int r = rem / G::SEGS, sg = rem - r * G::SEGS;
                const int gy = ty0 + r - G::PAD, gx = tx0 - G::XOFF + 4 * sg;
                if (gy >= 0 && gy < d.H && gx >= 0 && gx < d.W) {
                    if constexpr (PAIR) {
                        const int ci = c0 + 2 * c;                 // conv-space channel of the j = 0 plane
                        if (ci < d.Cin) {
                            const int cf = ci >> 2, i = (ci >> 1) & 1;
                            const float* src = d.x + (((size_t)b * (d.Cin >> 2) + cf) * (size_t)(2 * d.H) + (size_t)(2 * gy + i)) * (size_t)(2 * d.W) + (size_t)(2 * gx);
                            va = *reinterpret_cast<const f32x4*>(src);
                            vb = *reinterpret_cast<const f32x4*>(src + 4);
                        }
                    } else {
                        const int ci = c0 + c;
                        if (ci < d.Cin) {
                            const size_t idx = (((size_t)b * d.Cin + ci) * d.H + gy) * (size_t)d.W + gx;
                            va = *reinterpret_cast<const f32x4*>(d.x + idx);
                            if constexpr (IN == BNERV_IN_TANHGRAD) vb = *reinterpret_cast<const f32x4*>(d.aux0 + idx);
                        }
                    }
                }
            }
            ra[k] = va;
            if constexpr (TWO) rb[k] = vb;
        }
    }

    __device__ __forceinline__ void commit(const bnerv_conv_desc& d, float* s_in, int b, int c0, int nch, int ty0, int tx0) {
        const int ns = nslots(nch);
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int sidx = threadIdx.x + k * 256;
            if (sidx < ns) {
                const int c = sidx / (G::ROWS * G::SEGS);
                const int rem = sidx - c * (G::ROWS * G::SEGS);
                const int r = rem / G::SEGS, sg = rem - r * G::SEGS;
                if constexpr (PAIR) {
                    const f32x4 a = ra[k], bq = rb[k];
                    float* dst = s_in + (2 * c) * G::PLANE + r * G::RS + 4 * sg;
                    *reinterpret_cast<f32x4*>(dst) = f32x4{a[0], a[2], bq[0], bq[2]};
                    *reinterpret_cast<f32x4*>(dst + G::PLANE) = f32x4{a[1], a[3], bq[1], bq[3]};
                } else {
                    f32x4 v = ra[k];
                    if constexpr (IN != BNERV_IN_PLAIN) {
                        const int gy = ty0 + r - G::PAD, gx = tx0 - G::XOFF + 4 * sg, ci = c0 + c;
                        const bool inside = ci < d.Cin && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;     // padding is applied AFTER the prologue
                        if (inside) {
                            float sc = 1.f, sh = 0.f;
                            if constexpr (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE) { sc = 1.0f + d.scale[b * d.Cin + ci]; sh = d.shift[b * d.Cin + ci]; }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = xform1<IN>(v[e], sc, sh, TWO ? rb[k][e] : 0.f);
                        }
                    }
                    *reinterpret_cast<f32x4*>(s_in + c * G::PLANE + r * G::RS + 4 * sg) = v;
                }
            }
        }
    }
};

// B fragments: lane (j = l&15, kq = l>>4) <- W(co_base+16n+j, 4*(q0+q)+kq, tap)
// A fragment row r = (tap, q, n) is 64 consecutive elements = one wave's worth, so its decomposition is WAVE-UNIFORM: it is
// done once per row in scalar registers (the layers whose weights do not stay resident restage per K chunk per tile, and with
// per-element runtime divisions this cost as many instructions as the chunk's MFMAs).  8 gathers in flight per thread.
template <int KS, int NTB>
__device__ __forceinline__ void stage_weights(const bnerv_conv_desc& d, float* s_w, int co_base, int q0, int nq, int qstride) {
    using G = Geo<KS>;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int n_rows = G::T * nq * NTB;
    const float inv_nq = 1.0f / (float)nq;
    // per-lane element offset of (co = co_base + li, ci = 4 q0 + kq) at tap 0; the row adds (16 n, 4 q, tap)
    const int co_l = co_base + li, ci_l = q0 * 4 + kq;
    const int lane_off = d.transposed ? (ci_l * d.wCi + co_l) * G::T : (co_l * d.wCi + ci_l) * G::T;
    const int co_step = d.transposed ? G::T : d.wCi * G::T;          // element stride of +1 cout
    const int ci_step = d.transposed ? d.wCi * G::T : G::T;          // element stride of +1 cin
    for (int r0 = wave; r0 < n_rows; r0 += 4 * 8) {
        float v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;                                // uniform
            const int tq = r / NTB, n = r - tq * NTB;                // NTB is a compile-time constant
            const int tap = (int)(((float)tq + 0.5f) * inv_nq), q = tq - tap * nq;
            const int co = co_l + 16 * n, ci = ci_l + 4 * q;
            v[u] = 0.f;
            if (r < n_rows && co < d.Cout && ci < d.Cin)
                v[u] = d.w[lane_off + 16 * n * co_step + 4 * q * ci_step + (d.transposed ? G::T - 1 - tap : tap)];
            dst[u] = ((tap * qstride + q) * NTB + n) * 64 + lane;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 4 * u < n_rows) s_w[dst[u]] = v[u];
    }
}

// ---------------------------------------------------------------------------------------------------------------- epilogue
template <int EP>
__device__ __forceinline__ float ep_value(const bnerv_conv_desc& d, float v, float bias, size_t o, float* out2v) {
    if constexpr (EP == BNERV_EP_BIAS) return v + bias;
    if constexpr (EP == BNERV_EP_BIAS_SIN) { float sv, cv; sincos_f(v + bias, &sv, &cv); *out2v = cv; return sv; }
    if constexpr (EP == BNERV_EP_BIAS_GELU) { float h; gelu_pair_f(v + bias, &h, out2v); return h; }
    if constexpr (EP == BNERV_EP_BIAS_RES) return v + bias + d.aux0[o];
    if constexpr (EP == BNERV_EP_BIAS_TANH) return tanhf(v + bias) * 0.5f + 0.5f;
    return v;
}

// copy one cout-tile (16 conv-space channels starting at co0) from s_out to global, final layout.
template <int EP>
__device__ __forceinline__ void copy_out_tile(const KArgs& ka, const float* s_out, const Item& it, int co0) {
    const bnerv_conv_desc& d = ka.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W, Cout = d.Cout, b = it.b, ty0 = it.ty0, tx0 = it.tx0;
    const int s = d.out_s;
    float* outp = d.out;                                  // split-K partial results go to this item's slab instead
    if constexpr (EP == BNERV_EP_PLAIN) { if (ka.ksplit > 1) outp = d.partial + (size_t)it.ks * d.B * Cout * H * W; }
    if constexpr (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED) {
        // stride-1 only.  Pass k: wave w handles channel cl = 4k + w: 64 lanes = 8 rows x 8 float4 (vec) or 4 x 64 px (scalar);
        // per-channel (ds, dt) partial sums need only a wave reduction.
        for (int k = 0; k < 4; ++k) {
            const int cl = k * 4 + wave, co = co0 + cl;
            if (co >= Cout) continue;                       // wave-uniform
            const float sc = 1.0f + d.scale[b * Cout + co];
            float ps = 0.f, pt = 0.f;
            if (ka.vec) {
                const int py = lane >> 3, px = (lane & 7) * 4;
                const int gy = ty0 + py, gx = tx0 + px;
                if (gy < H && gx < W) {
                    const size_t o = (((size_t)b * Cout + co) * H + gy) * (size_t)W + gx;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + cl * CS + py * TW + px);
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(d.aux0 + o);
                    f32x4 r;
                    if constexpr (EP == BNERV_EP_DGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = v[e] * sc * gelu_grad_f(a0[e]); ps = fmaf(v[e], gelu_f(a0[e]), ps); pt += v[e]; }
                    } else if constexpr (EP == BNERV_EP_DGELU_SAVED) {
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(d.aux1 + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = v[e] * sc * a0[e]; ps = fmaf(v[e], a1[e], ps); pt += v[e]; }
                    } else {
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(d.aux1 + o);
                        f32x4 a2 = {1.f, 1.f, 1.f, 1.f};
                        if (d.aux2) a2 = *reinterpret_cast<const f32x4*>(d.aux2 + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r[e] = (a1[e] + v[e] * sc) * a2[e]; ps = fmaf(v[e], a0[e], ps); pt += v[e]; }
                    }
                    *reinterpret_cast<f32x4*>(d.out + o) = r;
                }
            } else {
                for (int p = lane; p < TH * TW; p += 64) {
                    const int py = p >> 5, px = p & 31, gy = ty0 + py, gx = tx0 + px;
                    if (gy >= H || gx >= W) continue;
                    const size_t o = (((size_t)b * Cout + co) * H + gy) * (size_t)W + gx;
                    const float v = s_out[cl * CS + p];
                    if constexpr (EP == BNERV_EP_DGELU) {
                        const float pre = d.aux0[o];
                        d.out[o] = v * sc * gelu_grad_f(pre);
                        ps = fmaf(v, gelu_f(pre), ps);
                    } else if constexpr (EP == BNERV_EP_DGELU_SAVED) {
                        d.out[o] = v * sc * d.aux0[o];
                        ps = fmaf(v, d.aux1[o], ps);
                    } else {
                        d.out[o] = (d.aux1[o] + v * sc) * (d.aux2 ? d.aux2[o] : 1.0f);
                        ps = fmaf(v, d.aux0[o], ps);
                    }
                    pt += v;
                }
            }
            ps = wave_sum(ps);
            pt = wave_sum(pt);
            if (lane == 0) {
                const size_t row = (size_t)it.tile * d.B + b;              // [tiles][B][2][Cout]
                d.partial[(row * 2 + 0) * Cout + co] = ps;
                d.partial[(row * 2 + 1) * Cout + co] = pt;
            }
        }
    } else if (s == 1 && ka.vec) {
        for (int idx = tid; idx < 16 * TH * (TW / 4); idx += 256) {
            const int cl = idx >> 6, rem = idx & 63, py = rem >> 3, px = (rem & 7) * 4;
            const int co = co0 + cl, gy = ty0 + py, gx = tx0 + px;
            if (co >= Cout || gy >= H || gx >= W) continue;
            if (ABL_NO_STORE && d.B > 0) continue;
            const size_t o = (((size_t)b * Cout + co) * H + gy) * (size_t)W + gx;
            const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + cl * CS + py * TW + px);
            const float bias = (EP != BNERV_EP_PLAIN && d.bias) ? d.bias[co] : 0.f;
            f32x4 r, r2;
            if constexpr (EP == BNERV_EP_BIAS_RES) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(d.aux0 + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = v[e] + bias + a0[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float c2 = 0.f; r[e] = ep_value<EP>(d, v[e], bias, o + e, &c2); r2[e] = c2; }
            }
            *reinterpret_cast<f32x4*>(outp + o) = r;
            if constexpr (EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) { if (d.out2) *reinterpret_cast<f32x4*>(d.out2 + o) = r2; }
        }
    } else if (s == 2 && ka.vec) {
        // PixelShuffle(2): final channel cf = co/4; output float4 = (px, j=0), (px, j=1), (px+1, j=0), (px+1, j=1) of row 2*py+i
        const int Cf = Cout >> 2, HF = 2 * H, WF = 2 * W;
        for (int idx = tid; idx < 4 * (2 * TH) * (TW / 2); idx += 256) {          // 4 final channels x 16 rows x 16 float4
            const int cfl = idx >> 8, rem = idx & 255, orow = rem >> 4, q4 = rem & 15;
            const int py = orow >> 1, i = orow & 1, px = q4 * 2;
            const int cl0 = cfl * 4 + i * 2;
            const int cf = (co0 >> 2) + cfl;
            const int gy = 2 * ty0 + orow, gx = 2 * tx0 + 4 * q4;
            if (cf >= Cf || gy >= HF || gx >= WF) continue;
            const float* p0 = s_out + cl0 * CS + py * TW + px;
            const float v00 = p0[0], v10 = p0[1], v01 = p0[CS], v11 = p0[CS + 1];
            const float b0 = (EP != BNERV_EP_PLAIN && d.bias) ? d.bias[co0 + cl0] : 0.f;
            const float b1 = (EP != BNERV_EP_PLAIN && d.bias) ? d.bias[co0 + cl0 + 1] : 0.f;
            const size_t o = (((size_t)b * Cf + cf) * HF + gy) * (size_t)WF + gx;
            f32x4 r, r2;
            const float in4[4] = {v00, v01, v10, v11};
            const float bs4[4] = {b0, b1, b0, b1};
#pragma unroll
            for (int e = 0; e < 4; ++e) { float c2 = 0.f; r[e] = ep_value<EP>(d, in4[e], bs4[e], o + e, &c2); r2[e] = c2; }
            *reinterpret_cast<f32x4*>(d.out + o) = r;
            if constexpr (EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) { if (d.out2) *reinterpret_cast<f32x4*>(d.out2 + o) = r2; }
        }
    } else {
        // generic pixel-shuffle scatter (s = 3, 5, or unaligned widths): low-resolution stages only
        const int s2 = s * s, Cf = Cout / s2, HF = H * s, WF = W * s;
        for (int e = tid; e < 16 * TH * TW; e += 256) {
            const int cl = e >> 8, p = e & 255, py = p >> 5, px = p & 31;
            const int co = co0 + cl;
            if (co >= Cout || ty0 + py >= H || tx0 + px >= W) continue;
            const int c = co / s2, rem = co - c * s2, i = rem / s, j = rem - i * s;
            const size_t o = (((size_t)b * Cf + c) * HF + (size_t)((ty0 + py) * s + i)) * (size_t)WF + (size_t)((tx0 + px) * s + j);
            const float bias = (EP != BNERV_EP_PLAIN && d.bias) ? d.bias[co] : 0.f;
            float c2 = 0.f;
            outp[o] = ep_value<EP>(d, s_out[cl * CS + py * TW + px], bias, o, &c2);
            if constexpr (EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) { if (d.out2) d.out2[o] = c2; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- kernel
template <int KS, int IN, int EP, int NTB>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const KArgs ka, const SidePack side) {
    using G = Geo<KS>;
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                   // CC * PLANE
    float* s_out = smem + CC * G::PLANE;                  // 16 * CS
    float* s_w = s_out + 16 * CS;                         // resident: T*nq_total*NTB*64 ; else T*NQ*NTB*64

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin;

    // this block's item range: XCD x owns a contiguous slice of the item list; its blocks take it round-robin
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int nlb = (gridDim.x - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);

    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = kq * G::PLANE + (2 * wave + (m >> 1)) * G::RS + (m & 1) * 16 + li + G::COL0;

    const int qstride = ka.w_resident ? ka.nq_total : NQ;
    const bool vec_in = ka.vec && (IN != BNERV_IN_UNSHUFFLE || d.in_s == 2);
    // Software pipeline over (item, K chunk) stages whenever the float4 staging applies: the global loads of stage s+1 are
    // issued into registers before the MFMA phase of stage s and committed to LDS after it (one s_in buffer, two LDS-only
    // barriers per stage), so neither the chunks of one tile nor consecutive tiles expose their load latency.
    const bool piped = vec_in;
    auto chunk_begin = [&](const Item& i) { return ka.ksplit > 1 ? i.ks * ka.chunks_per_split * CC : 0; };
    auto chunk_end = [&](const Item& i) { return ka.ksplit > 1 ? min(Cin, chunk_begin(i) + ka.chunks_per_split * CC) : Cin; };
    auto chunk_nch = [&](int c0, int c_end) { return ((min(CC, c_end - c0) + 3) >> 2) * 4; };
    VecStage<KS, IN> vs;
    int cur_g = -1;

    int itx = r0 + lb;
    if (itx < r1 && piped) {
        const Item it = decode_item(ka, itx);
        const int c0 = chunk_begin(it);
        vs.issue(d, it.b, c0, chunk_nch(c0, chunk_end(it)), it.ty0, it.tx0);
        vs.commit(d, s_in, it.b, c0, chunk_nch(c0, chunk_end(it)), it.ty0, it.tx0);
    }
    for (; itx < r1; itx += nlb) {
        const Item it = decode_item(ka, itx);
        const int co_base = it.g * NTB * 16;
        f32x4 acc[4][NTB];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (ka.w_resident && cur_g != it.g) {
            lds_barrier();                                 // nobody still reads the previous group's fragments
            stage_weights<KS, NTB>(d, s_w, co_base, 0, ka.nq_total, qstride);
            cur_g = it.g;
        }
        const bool has_next = itx + nlb < r1;
        Item nxt = it;
        if (has_next) nxt = decode_item(ka, itx + nlb);

        const int c_begin = chunk_begin(it), c_end = chunk_end(it);
        for (int c0 = c_begin; c0 < c_end; c0 += CC) {
            const int cc = min(CC, Cin - c0);
            const int nq = (cc + 3) >> 2;
            const int q0 = c0 >> 2;
            const bool last_chunk = c0 + CC >= c_end;
            if (!piped) {
                __syncthreads();                           // previous chunk / tile fully consumed
                if (vec_in) { vs.issue(d, it.b, c0, nq * 4, it.ty0, it.tx0); vs.commit(d, s_in, it.b, c0, nq * 4, it.ty0, it.tx0); }
                else stage_scalar<KS, IN>(d, s_in, it.b, c0, nq * 4, it.ty0, it.tx0);
            }
            if (!ka.w_resident) stage_weights<KS, NTB>(d, s_w, co_base, q0, nq, qstride);   // (after the previous stage's barrier B)
            lds_barrier();                                 // (A) this stage's s_in / s_w visible
            if (piped) {                                   // next stage's loads fly under the MFMA phase
                if (!last_chunk) vs.issue(d, it.b, c0 + CC, chunk_nch(c0 + CC, c_end), it.ty0, it.tx0);
                else if (has_next) vs.issue(d, nxt.b, chunk_begin(nxt), chunk_nch(chunk_begin(nxt), chunk_end(nxt)), nxt.ty0, nxt.tx0);
            }
            const int qb = ka.w_resident ? q0 : 0;
#pragma unroll
            for (int tap = 0; tap < G::T; ++tap) {
                const int toff = (tap / KS) * G::RS + (tap % KS);
                for (int q = 0; q < nq; ++q) {
                    float bf[NTB], af[4];
#pragma unroll
                    for (int n = 0; n < NTB; ++n) bf[n] = s_w[((tap * qstride + qb + q) * NTB + n) * 64 + lane];
#pragma unroll
                    for (int m = 0; m < 4; ++m) af[m] = s_in[abase[m] + q * 4 * G::PLANE + toff];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < NTB; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
                }
            }
            if (piped && !last_chunk) {
                lds_barrier();                             // (B) every wave is done reading this chunk
                vs.commit(d, s_in, it.b, c0 + CC, chunk_nch(c0 + CC, c_end), it.ty0, it.tx0);
            }
        }

        // ---- epilogue, one cout-tile at a time through s_out (D layout: lane holds pixels 4*kq..4*kq+3 of cout li) ----
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            if (co_base + n * 16 < d.Cout) {
                if (n > 0) lds_barrier();                  // previous cout-tile copied out
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int py = 2 * wave + (m >> 1), px = (m & 1) * 16 + 4 * kq;
                    *reinterpret_cast<f32x4*>(&s_out[li * CS + py * TW + px]) = acc[m][n];
                }
                lds_barrier();                             // s_out complete; every wave is also done reading s_in / s_w(chunk)
                if (n == 0 && piped && has_next)
                    vs.commit(d, s_in, nxt.b, chunk_begin(nxt), chunk_nch(chunk_begin(nxt), chunk_end(nxt)), nxt.ty0, nxt.tx0);
                copy_out_tile<EP>(ka, s_out, it, co_base + n * 16);
            }
        }
        lds_barrier();                                     // s_out free for the next item; s_in(next) visible
    }
    side_run_hosted(side, smem);                           // queued slab reductions (sidejob.h)
}

// ---------------------------------------------------------------------------------------------------------------- fast kernel
// Single-K-chunk layers (Cin <= 16: every high-resolution stage of the NeRV-style decoders), float4-aligned rows.
// Compared with the generic kernel above:
//   * the K loop is FULLY unrolled (NQ1 = ceil(Cin/4) is a template constant): every ds_read carries an immediate offset,
//     the loop body is ds_read_b32 x5 + MFMA x4 with no VALU at all;
//   * the (channel,row,segment) geometry of each thread's staging slots is tile-invariant and computed ONCE per block
//     (LDS offset, global offset relative to the tile origin); interior tiles skip every bounds check;
//   * LDS is carved to the layer's real channel counts, so a 12->12 layer needs 38.6 KB -> 4 blocks per CU.
#ifdef BNERV_TRACE
__device__ unsigned long long g_trace[1024 * 4 * 6 * 8];
#define TRACE(slot) do { if (lane == 0 && blockIdx.x < 1024 && trace_iter < 6) g_trace[((blockIdx.x * 4 + wave) * 6 + trace_iter) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRACE(slot) do {} while (0)
#endif
template <int KS, int IN, int EP, int NTB, int NQ1>
__global__ __launch_bounds__(256, (NTB == 1 ? (NQ1 <= 3 ? 4 : 3) : 2)) void conv_fast_kernel(const KArgs ka, const int ncs /* s_out channels */, const SidePack side) {
    using G = Geo<KS>;
    constexpr int NCH = NQ1 * 4;
    constexpr int NSLOT = NCH * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    constexpr bool TWO = (IN == BNERV_IN_TANHGRAD);
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                   // NCH * PLANE
    float* s_out = smem + NCH * G::PLANE;                 // ncs * CS
    float* s_w = s_out + ncs * CS;                        // T * NQ1 * NTB * 64

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, H = d.H, W = d.W;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int nlb = (gridDim.x - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);

    // tile-invariant slot geometry
    int lds_off[NPRE], goff[NPRE], rsc[NPRE];            // rsc: row | seg << 8 | channel << 16 | valid << 31... (valid kept in sign)
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int sidx = tid + k * 256;
        const int c = sidx / (G::ROWS * G::SEGS);
        const int rem = sidx - c * (G::ROWS * G::SEGS);
        const int r = rem / G::SEGS, sg = rem - r * G::SEGS;
        const bool ok = sidx < NSLOT;
        lds_off[k] = ok ? c * G::PLANE + r * G::RS + 4 * sg : -1;
        goff[k] = (c * H + (r - G::PAD)) * W + 4 * sg - G::XOFF;
        rsc[k] = r | (sg << 8) | (c << 16);
    }
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = kq * G::PLANE + (2 * wave + (m >> 1)) * G::RS + (m & 1) * 16 + li + G::COL0;

    f32x4 ra[NPRE], rb[TWO ? NPRE : 1];
    auto issue = [&](const Item& it) {
        const size_t tbase = ((size_t)it.b * Cin * H + it.ty0) * (size_t)W + it.tx0;
        const bool interior = it.ty0 >= G::PAD && it.ty0 + TH + G::PAD <= H && it.tx0 >= G::XOFF && it.tx0 + TW + G::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            bool ok = lds_off[k] >= 0 && (rsc[k] >> 16) < Cin;
            if (ABL_NO_LOAD) ok = ok && d.B < 0;
            if (!interior) {
                const int gy = it.ty0 + (rsc[k] & 255) - G::PAD, gx = it.tx0 - G::XOFF + 4 * ((rsc[k] >> 8) & 255);
                ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
            }
            if (ok) {
                va = *reinterpret_cast<const f32x4*>(d.x + tbase + goff[k]);
                if constexpr (TWO) vb = *reinterpret_cast<const f32x4*>(d.aux0 + tbase + goff[k]);
            }
            ra[k] = va;
            if constexpr (TWO) rb[k] = vb;
        }
    };
    auto commit = [&](const Item& it) {
        const bool interior = it.ty0 >= G::PAD && it.ty0 + TH + G::PAD <= H && it.tx0 >= G::XOFF && it.tx0 + TW + G::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            if (lds_off[k] < 0) continue;
            f32x4 v = ra[k];
            if constexpr (IN != BNERV_IN_PLAIN) {
                const int ci = rsc[k] >> 16;
                bool ok = ci < Cin;
                if (!interior) {
                    const int gy = it.ty0 + (rsc[k] & 255) - G::PAD, gx = it.tx0 - G::XOFF + 4 * ((rsc[k] >> 8) & 255);
                    ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
                }
                if (ok) {                                  // padding is applied AFTER the prologue: outside stays 0
                    float sc = 1.f, sh = 0.f;
                    if constexpr (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE) { sc = 1.0f + d.scale[it.b * Cin + ci]; sh = d.shift[it.b * Cin + ci]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xform1<IN>(v[e], sc, sh, TWO ? rb[k][e] : 0.f);
                }
            }
            *reinterpret_cast<f32x4*>(s_in + lds_off[k]) = v;
        }
    };

    // Barriers here guard LDS hazards only, so they wait for LDS traffic (lgkmcnt) and NOT for the global stores of the
    // previous copy-out or the prefetch loads in flight (a __syncthreads() would drain both: vmcnt(0)).
    // Per tile (NTB = 1): (A) s_in(t) + weights visible, s_out free   (B) s_out(t) complete, s_in free.
    int cur_g = -1;
    int itx = r0 + lb;
    Item it = decode_item(ka, itx < r1 ? itx : 0);
    if (itx < r1) {
        issue(it);
        commit(it);
    }
    int trace_iter = 0; (void)trace_iter;
    for (; itx < r1; itx += nlb, ++trace_iter) {
        TRACE(0);
        const int co_base = it.g * NTB * 16;
        f32x4 acc[4][NTB];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (cur_g != it.g) {
            lds_barrier();
            stage_weights<KS, NTB>(d, s_w, co_base, 0, NQ1, NQ1);
            cur_g = it.g;
        }
        const bool has_next = itx + nlb < r1;
        Item nxt = it;
        if (has_next) nxt = decode_item(ka, itx + nlb);
        lds_barrier();                                     // (A)
        TRACE(1);
        if (has_next) issue(nxt);                          // flies under the MFMA phase
        TRACE(2);
#if !(BNERV_ABL == 1 || BNERV_ABL == 3 || BNERV_ABL == 4)
#pragma unroll
        for (int tap = 0; tap < G::T; ++tap) {
#pragma unroll
            for (int q = 0; q < NQ1; ++q) {
                float bf[NTB], af[4];
#pragma unroll
                for (int n = 0; n < NTB; ++n) bf[n] = s_w[((tap * NQ1 + q) * NTB + n) * 64 + lane];
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = s_in[abase[m] + (q * 4 * G::PLANE + (tap / KS) * G::RS + (tap % KS))];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
        }
#endif
        TRACE(3);
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            if (co_base + n * 16 < d.Cout) {
                if (n > 0) lds_barrier();
                if (li < ncs) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int py = 2 * wave + (m >> 1), px = (m & 1) * 16 + 4 * kq;
                        *reinterpret_cast<f32x4*>(&s_out[li * CS + py * TW + px]) = acc[m][n];
                    }
                }
                TRACE(4);
                lds_barrier();                             // (B)
                TRACE(5);
                if (n == 0 && has_next) commit(nxt);
                TRACE(6);
                copy_out_tile<EP>(ka, s_out, it, co_base + n * 16);
                TRACE(7);
            }
        }
        it = nxt;                                          // barrier (A) of the next iteration also frees s_out
    }
    side_run_hosted(side, smem);                           // queued slab reductions (sidejob.h)
}

template <int KS, int IN, int EP, int NTB, int NQ1>
int launch_fast(hipStream_t st, KArgs& ka) {
    using G = Geo<KS>;
    const bnerv_conv_desc& d = ka.d;
    ka.ngroups = cdiv(cdiv(d.Cout, 16), NTB);
    ka.total_items = ka.ngroups * d.B * ka.tiles_x * ka.tiles_y;
    ka.nq_total = NQ1;
    ka.w_resident = 1;
    const int s2 = d.out_s * d.out_s;
    int ncs = d.Cout >= 16 ? 16 : ((d.Cout + 3) / 4) * 4;          // multiple of 4: float4 / PixelShuffle(2) groups stay whole
    if (s2 > 4) ncs = 16;
    const size_t lds = ((size_t)NQ1 * 4 * G::PLANE + (size_t)ncs * CS + (size_t)G::T * NQ1 * NTB * 64) * sizeof(float);
    static size_t attr_lds = 0;
    static int blocks_per_cu = 0;
    if (lds > attr_lds || blocks_per_cu == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fast_kernel<KS, IN, EP, NTB, NQ1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_fast_kernel<KS, IN, EP, NTB, NQ1>), 256, lds) != hipSuccess || nb < 1) nb = 1;
        blocks_per_cu = nb > 4 ? 4 : nb;
    }
    int grid = 256 * blocks_per_cu;                       // everything resident: the static item partition is then balanced
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(ka.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_fast_kernel<KS, IN, EP, NTB, NQ1>), dim3(grid), dim3(256), lds, st, ka, ncs, side);
    BNERV_LAUNCH_CHECK("conv_fast");
    return BNERV_OK;
}

// ---------------------------------------------------------------------------------------------------------------- lean kernel
// Measured on gfx950 (tools/ubench/mfma_valu_mix.cpp, tools/ktrace.py): ordinary VALU / SALU instructions do NOT hide under
// the MFMAs of the other waves of a SIMD -- each costs ~2.6 / ~1.7 cycles of SIMD time on top of the 32 cycles per
// MFMA 16x16x4 f32.  The per-tile work around the K loop therefore has to be counted in instructions:
//   * global traffic goes through raw buffer loads / stores: per-tile SGPR base (soffset) + per-slot VGPR offsets that are
//     computed ONCE per block; out-of-image / idle slots carry an offset beyond num_records, which the buffer unit turns into
//     "load 0 / drop store" without any branch.  Interior tiles (uniform test) run a path with no per-slot arithmetic at all.
//   * the epilogue works straight from the accumulators: an MFMA 16x16x4 D fragment is 4 consecutive pixels of one output
//     channel per lane = one 16-byte store.  No s_out staging, no LDS round trip, no copy-out address math.
//   * the item (sample, tile-row, tile-col) is advanced incrementally instead of decoded by integer division.
// Scope: Cout <= 16 (one N tile), stride-1 output, 8 < Cin <= 16, float4-aligned rows, tensors < 2 GiB.

template <int KS, int IN, int EP, int NQ1>
__global__ __launch_bounds__(256, (NQ1 <= 3 ? 4 : 3)) void conv_lean_kernel(const KArgs ka, const SidePack side) {
    using G = Geo<KS>;
    constexpr int NCH = NQ1 * 4;
    constexpr int NSLOT = NCH * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    constexpr int S_IN = NCH * G::PLANE + (NPRE * 256 - NSLOT) * 4;        // floats; the tail is a dump area for idle slots
    constexpr bool TWO = (IN == BNERV_IN_TANHGRAD);
    constexpr bool AFF = (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE);
    constexpr bool RED = (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + S_IN;                              // [T][NQ1][64] B fragments
    float* s_red = s_w + G::T * NQ1 * 64;                  // [4 waves][2][16] per-channel partial sums (DGELU / DSIN)
    float* s_aff = s_red + 128;                            // [2][16] affine prologue parameters of the current sample

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int tiles_x = ka.tiles_x, tiles_y = ka.tiles_y;
    { const int trace_iter = 5; (void)trace_iter; TRACE(0); }

    // this block's item range: XCD x owns a contiguous slice of the item list; its blocks take it round-robin
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int nlb = (gridDim.x - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lb;
    if (itx >= r1) { side_run_hosted(side, smem); return; }
    const int step_q = fast_div(nlb, ka.magic_tiles_x), step_r = nlb - step_q * tiles_x;      // uniform: two SALU per division
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

    // ---- per-slot constants (slot = (channel, halo row, 4-px segment); thread t owns slots t, t+256, ...)
    // (channel, row, segment) of slot k -- cheap constant divisions, recomputed on the rare paths that need them
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
    // LDS byte offset of slot k.  Unpadded planes (3x3: PLANE == ROWS * RS) make it 16 * slot index for real and idle slots
    // alike -> one VGPR + immediates; padded planes (1x1) need the general form.
    auto loff = [&](int k) {
        const int sidx = tid + k * 256;
        if constexpr (G::PLANE == G::PLANE_RAW) return sidx * 16;
        int c, r, sg;
        slot_geom(k, c, r, sg);
        return sidx < NSLOT ? (c * G::PLANE + r * G::RS + 4 * sg) * 4 : (NCH * G::PLANE + (sidx - NSLOT) * 4) * 4;
    };
    unsigned voff[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        voff[k] = (tid + k * 256 < NSLOT && c < Cin) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;
    }
    // the x view starts PAD rows + XOFF columns before the tensor, so every slot offset is >= 0
    const unsigned shift = (unsigned)((G::PAD * W + G::XOFF) * 4);
    const unsigned in_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, in_bytes);
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(TWO ? d.aux0 : d.x, shift, in_bytes);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ro2 = make_rsrc(((EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) && d.out2) ? d.out2 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc((!TWO && d.aux0) ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);

    // epilogue lane constants: lane (li, kq) owns output channel li, pixels 4*kq .. 4*kq+3 of each 16-px M tile
    const unsigned ovoff = li < Cout ? (unsigned)(((li * H) * W + 4 * kq) * 4) : OOB;
    const float bias_l = (EP != BNERV_EP_PLAIN && !RED && d.bias && li < Cout) ? d.bias[li] : 0.f;

    // Affine prologue parameters go through LDS: ONE 32-lane load per block (s_aff[c] = 1 + scale[b][c], s_aff[16 + c] =
    // shift[b][c]), then each thread picks the values of its slots.  (A wave64 global load occupies the address unit for ~16
    // cycles whatever its width: 2 * NPRE per-lane loads in every wave of every block were ~2.5 k cycles of every launch.)
    float sc[NPRE], sh[NPRE];
    auto fetch_affine = [&](int b) {                       // tid < 32 only: the value this lane contributes
        const int c = tid & 15;
        float v = 0.f;
        if (tid < 32 && c < Cin) v = tid < 16 ? 1.0f + d.scale[b * Cin + c] : d.shift[b * Cin + c];
        return v;
    };
    auto pick_affine = [&]() {                             // after s_aff is visible
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            int c, r, sg;
            slot_geom(k, c, r, sg);
            const bool ok = voff[k] != OOB;
            sc[k] = ok ? s_aff[c & 15] : 0.f;
            sh[k] = ok ? s_aff[16 + (c & 15)] : 0.f;
        }
    };
    auto load_affine = [&](int b) {                        // mid-loop reload when the sample changes (B > 1; latency exposed, rare)
        const float v = fetch_affine(b);
        lds_barrier();
        if (tid < 32) s_aff[tid] = v;
        lds_barrier();
        pick_affine();
    };
    float scl = 0.f;                                       // 1 + scale[b][co] of the DGELU / DSIN epilogues

    f32x4 ra[NPRE], rb[TWO ? NPRE : 1];
    auto issue = [&](const LItem& a) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const unsigned sb = (unsigned)((((a.b * Cin) * H + ty0) * W + tx0) * 4);
        const bool interior = ty0 >= G::PAD && ty0 + TH + G::PAD <= H && tx0 >= G::XOFF && tx0 + TW + G::XOFF <= W;
        if (interior) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                ra[k] = bload(rx, voff[k], sb);
                if constexpr (TWO) rb[k] = bload(rx2, voff[k], sb);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const unsigned vo = slot_inside(k, ty0, tx0) ? voff[k] : OOB;
                ra[k] = bload(rx, vo, sb);
                if constexpr (TWO) rb[k] = bload(rx2, vo, sb);
            }
        }
    };
    auto commit = [&](const LItem& a) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const bool interior = ty0 >= G::PAD && ty0 + TH + G::PAD <= H && tx0 >= G::XOFF && tx0 + TW + G::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            f32x4 v = ra[k];
            if constexpr (IN != BNERV_IN_PLAIN) {
                float s = AFF ? sc[k] : 0.f, h = AFF ? sh[k] : 0.f;
                if constexpr (AFF) {
                    if (!interior) {                       // zero padding is applied AFTER the prologue: outside stays 0
                        const bool ok = slot_inside(k, ty0, tx0);
                        s = ok ? s : 0.f;
                        h = ok ? h : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xform1<IN>(v[e], s, h, TWO ? rb[k][e] : 0.f);
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(s_in) + loff(k)) = v;
        }
    };
    auto flush_partials = [&](const LItem& a) {            // wave 0: sum the 4 waves' channel sums of tile `a`, fixed order
        if (wave == 0 && lane < 32) {
            const int q = lane >> 4, c = lane & 15;
            const float s = ((s_red[(0 * 2 + q) * 16 + c] + s_red[(1 * 2 + q) * 16 + c]) + s_red[(2 * 2 + q) * 16 + c]) + s_red[(3 * 2 + q) * 16 + c];
            const size_t row = (size_t)(a.ty * tiles_x + a.tx) * d.B + a.b;            // [tiles][B][2][Cout]
            if (c < Cout) d.partial[(row * 2 + q) * Cout + c] = s;
        }
    };

    // prologue: the first tile's loads, then ALL weight loads back to back (one exposed memory latency for the lot)
    int aff_b = -1, ep_b = -1;
    { const int trace_iter = 5; (void)trace_iter; TRACE(2); }
    float aff_v = 0.f;
    if constexpr (AFF) { aff_v = fetch_affine(it.b); aff_b = it.b; }      // oldest load in flight: first to come back
    issue(it);
    { const int trace_iter = 5; (void)trace_iter; TRACE(3); }
    {
        constexpr int NWV = (G::T * NQ1 * 64 + 255) / 256;
        float wv[NWV];
        // B fragment (tap, q) lane l <- W(co = l & 15, ci = 4 q + (l >> 4), tap).  (tap, q) is wave-uniform (tq = wave + 4 j): the
        // per-lane part of the element index is one constant, the rest is scalar arithmetic.
        const int lane_w = d.transposed ? ((kq * d.wCi + li) * G::T) : ((li * d.wCi + kq) * G::T);
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int tq = wave + 4 * j;
            const int tap = tq / NQ1, q = tq - tap * NQ1;
            const int sc_off = d.transposed ? (4 * q * d.wCi * G::T + (G::T - 1 - tap)) : (4 * q * G::T + tap);
            float v = 0.f;
            if (tq < G::T * NQ1 && li < Cout && q * 4 + kq < Cin) v = d.w[lane_w + sc_off];
            wv[j] = v;
        }
        { const int trace_iter = 5; (void)trace_iter; TRACE(4); }
        if constexpr (AFF) { if (tid < 32) s_aff[tid] = aff_v; }
#pragma unroll
        for (int j = 0; j < NWV; ++j)
            if (tid + j * 256 < G::T * NQ1 * 64) s_w[tid + j * 256] = wv[j];
        { const int trace_iter = 5; (void)trace_iter; TRACE(5); }
    }
    if constexpr (AFF) { lds_barrier(); pick_affine(); }
    commit(it);
    { const int trace_iter = 5; (void)trace_iter; TRACE(1); }
    const int abase = kq * G::PLANE + (2 * wave) * G::RS + li + G::COL0;
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
        lds_barrier();                                     // (A) s_in(t), weights and s_red(t-1) visible
        TRACE(1);
        if (has_next) issue(nxt);                          // flies under the MFMA phase
        TRACE(2);
        if constexpr (RED) { if (have_prev) flush_partials(prev); }
        // K loop: a REAL loop over the tap rows (KS x NQ1 steps unrolled inside) -- fully unrolled, the scheduler hoists LDS
        // reads until it runs out of registers and spills.
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
            const float* a_row = s_in + abase + ky * G::RS;
            const float* b_row = s_w + ky * (KS * NQ1 * 64) + lane;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int q = 0; q < NQ1; ++q) {
                    float af[4];
                    const float bf = b_row[(kx * NQ1 + q) * 64];
#pragma unroll
                    for (int m = 0; m < 4; ++m) af[m] = a_row[q * 4 * G::PLANE + (m >> 1) * G::RS + (m & 1) * 16 + kx];
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf, acc[m], 0, 0, 0);
                }
            }
        }
        TRACE(3);
        TRACE(4);
        lds_barrier();                                     // (B) every wave is done reading s_in(t)
        TRACE(5);
        if (has_next) {
            if constexpr (AFF) { if (nxt.b != aff_b) { load_affine(nxt.b); aff_b = nxt.b; } }
            commit(nxt);
        }
        TRACE(6);
        // ---- epilogue straight from the accumulators
        {
            const int ty0 = it.ty * TH, tx0 = it.tx * TW;
            const unsigned ob = (unsigned)((((it.b * Cout) * H + ty0 + 2 * wave) * W + tx0) * 4);
            const bool full = ty0 + TH <= H && tx0 + TW <= W;
            if constexpr (RED) { if (it.b != ep_b) { scl = li < Cout ? 1.0f + d.scale[it.b * Cout + li] : 0.f; ep_b = it.b; } }
            unsigned so[4], vo[4];
            bool row_ok[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                so[m] = ob + (unsigned)(((m >> 1) * W + (m & 1) * 16) * 4);
                vo[m] = ovoff;
                row_ok[m] = true;
                if (!full) {
                    row_ok[m] = ty0 + 2 * wave + (m >> 1) < H;
                    const bool okx = tx0 + (m & 1) * 16 + 4 * kq < W;
                    vo[m] = (okx && row_ok[m]) ? ovoff : OOB;
                    if constexpr (RED) { if (!(okx && row_ok[m])) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
            (void)row_ok;
        }
        TRACE(7);
        prev = it;
        have_prev = true;
        it = nxt;
    }
    if constexpr (RED) {
        lds_barrier();
        flush_partials(prev);
    }
    side_run_hosted(side, smem);                           // queued slab reductions, least-loaded blocks first (sidejob.h)
}


template <int KS, int IN, int EP, int NQ1>
int launch_lean(hipStream_t st, KArgs& ka) {
    using G = Geo<KS>;
    const bnerv_conv_desc& d = ka.d;
    constexpr int NCH = NQ1 * 4;
    constexpr int NSLOT = NCH * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    ka.ngroups = 1;
    ka.total_items = d.B * ka.tiles_x * ka.tiles_y;
    ka.nq_total = NQ1;
    ka.w_resident = 1;
    ka.magic_tiles = div_magic(ka.tiles_x * ka.tiles_y);
    ka.magic_tiles_x = div_magic(ka.tiles_x);
    const size_t lds = ((size_t)NCH * G::PLANE + (size_t)(NPRE * 256 - NSLOT) * 4 + (size_t)G::T * NQ1 * 64 + 128 + 32) * sizeof(float);
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_lean_kernel<KS, IN, EP, NQ1>), 256, lds) != hipSuccess || nb < 1) nb = 1;
        blocks_per_cu = nb;
    }
    int grid = 256 * blocks_per_cu;                       // everything resident: the static item partition is then balanced
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(ka.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_lean_kernel<KS, IN, EP, NQ1>), dim3(grid), dim3(256), lds, st, ka, side);
    BNERV_LAUNCH_CHECK("conv_lean");
    return BNERV_OK;
}

// ---------------------------------------------------------------------------------------------------------------- lean2 kernel
// The lean recipe for the layers the first lean kernel does not take: Cin > 16 (K walked in chunks of 16 channels, the next
// (tile, chunk) stage prefetched into registers under the MFMA phase of the current one) and/or Cout > 16 (NTB <= 4 cout
// tiles per block, the input tile staged once for all of them).  Stride-1 outputs only: the epilogue works straight from the
// accumulators for every cout tile.  These are the TAT convolutions, heads and their data gradients of the 3M models
// (22..55 channels) and of C1's 30-channel stage; the generic kernel keeps PixelShuffle outputs and the unshuffle prologue.
template <int KS, int IN, int EP, int NTB>
__global__ __launch_bounds__(256, (NTB == 1 ? 3 : 2)) void conv_lean2_kernel(const KArgs ka, const SidePack side) {
    using G = Geo<KS>;
    constexpr int NQ1 = 4, NCH = 16;
    // IN_UNSHUFFLE (in_s == 2, the data gradient of an up-conv feeding PixelShuffle(2)): conv channel 4c + 2i + j at (y, x) is
    // du[c][2y + i][2x + j]; a slot is (c, i, row, 4-px segment) = 8 consecutive floats of one du row, split into the j = 0 / 1 planes
    constexpr bool UNSH = (IN == BNERV_IN_UNSHUFFLE);
    constexpr int NSLOT = (UNSH ? NCH / 2 : NCH) * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    constexpr int S_IN = NCH * G::PLANE + (UNSH ? 0 : (NPRE * 256 - NSLOT) * 4);
    constexpr bool TWO = (IN == BNERV_IN_TANHGRAD);
    constexpr bool AFF = (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE);
    constexpr bool RED = (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN || EP == BNERV_EP_DGELU_SAVED);
    constexpr int AFF_MAX = 128;                           // input channels whose affine parameters fit the LDS table
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_red = smem + S_IN;                            // [4 waves][2][NTB*16]
    float* s_aff = s_red + 4 * 2 * NTB * 16;               // [2][AFF_MAX]
    float* s_w = s_aff + 2 * AFF_MAX;                      // resident: T*nq_total*NTB*64 ; else T*NQ1*NTB*64

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int nchunks = (Cin + NCH - 1) / NCH;
    const int qstride = ka.w_resident ? ka.nq_total : NQ1;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int nlb = (gridDim.x - xcd + 7) >> 3;
    const int per = ka.total_items >> 3, extra = ka.total_items & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lb;
    if (itx >= r1) { side_run_hosted(side, smem); return; }

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
    auto loff = [&](int k) {
        const int sidx = tid + k * 256;
        int c, r, sg;
        slot_geom(k, c, r, sg);
        if constexpr (UNSH) return ((4 * (c >> 1) + 2 * (c & 1)) * G::PLANE + r * G::RS + 4 * sg) * 4;   // the j = 0 plane; j = 1 follows
        if constexpr (G::PLANE == G::PLANE_RAW) return sidx * 16;
        return sidx < NSLOT ? (c * G::PLANE + r * G::RS + 4 * sg) * 4 : (NCH * G::PLANE + (sidx - NSLOT) * 4) * 4;
    };
    unsigned voff[NPRE];                                   // chunk-local: channel c of the chunk
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c, r, sg;
        slot_geom(k, c, r, sg);
        if constexpr (UNSH) voff[k] = (tid + k * 256 < NSLOT) ? (unsigned)(((((c >> 1) * 2 * H) + 2 * r + (c & 1)) * (2 * W) + 8 * sg) * 4) : OOB;
        else voff[k] = (tid + k * 256 < NSLOT) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;
    }
    const unsigned shift = UNSH ? (unsigned)((2 * G::PAD * 2 * W + 2 * G::XOFF) * 4) : (unsigned)((G::PAD * W + G::XOFF) * 4);
    const unsigned in_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned out_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, in_bytes);
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(TWO ? d.aux0 : d.x, shift, in_bytes);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ro2 = make_rsrc(((EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU) && d.out2) ? d.out2 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra0 = make_rsrc((!TWO && d.aux0) ? d.aux0 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(d.aux1 ? d.aux1 : d.out, 0, out_bytes);
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(d.aux2 ? d.aux2 : d.out, 0, out_bytes);

    auto load_affine = [&](int b) {                        // s_aff[c] = 1 + scale[b][c], s_aff[AFF_MAX + c] = shift[b][c]; 0 beyond Cin
        lds_barrier();
        for (int i = tid; i < 2 * AFF_MAX; i += 256) {
            const int c = i < AFF_MAX ? i : i - AFF_MAX;
            float v = 0.f;
            if (c < Cin) v = i < AFF_MAX ? 1.0f + d.scale[b * Cin + c] : d.shift[b * Cin + c];
            s_aff[i] = v;
        }
        lds_barrier();
    };

    f32x4 ra[NPRE], rb[(TWO || UNSH) ? NPRE : 1];
    // stage = (item, chunk); issue its loads / commit them to s_in
    auto issue = [&](const Item& a, int c0) {
        const unsigned sb = UNSH ? (unsigned)((((a.b * (Cin >> 2) + (c0 >> 2)) * 2 * H + 2 * a.ty0) * (2 * W) + 2 * a.tx0) * 4)
                                 : (unsigned)((((a.b * Cin + c0) * H + a.ty0) * W + a.tx0) * 4);
        const bool interior = a.ty0 >= G::PAD && a.ty0 + TH + G::PAD <= H && a.tx0 >= G::XOFF && a.tx0 + TW + G::XOFF <= W;
        const int nch = min(NCH, Cin - c0);
        const bool plain = interior && nch == NCH;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            unsigned vo = voff[k];
            if (!plain) {
                int c, r, sg;
                slot_geom(k, c, r, sg);
                if ((UNSH ? 2 * c : c) >= nch || (!interior && !slot_inside(k, a.ty0, a.tx0))) vo = OOB;
            }
            ra[k] = bload(rx, vo, sb);
            if constexpr (TWO) rb[k] = bload(rx2, vo, sb);
            if constexpr (UNSH) rb[k] = bload(rx, vo == OOB ? OOB : vo + 16u, sb);
        }
    };
    auto commit = [&](const Item& a, int c0) {
        const bool interior = a.ty0 >= G::PAD && a.ty0 + TH + G::PAD <= H && a.tx0 >= G::XOFF && a.tx0 + TW + G::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            if constexpr (UNSH) {
                if (k == NPRE - 1 && tid + k * 256 >= NSLOT) continue;                 // idle slots of the last round
                const f32x4 a0 = ra[k], a1 = rb[k];
                char* dst = reinterpret_cast<char*>(s_in) + loff(k);
                *reinterpret_cast<f32x4*>(dst) = f32x4{a0[0], a0[2], a1[0], a1[2]};
                *reinterpret_cast<f32x4*>(dst + G::PLANE * 4) = f32x4{a0[1], a0[3], a1[1], a1[3]};
                continue;
            }
            f32x4 v = ra[k];
            if constexpr (IN != BNERV_IN_PLAIN) {
                float sc_ = 0.f, sh_ = 0.f;
                if constexpr (AFF) {
                    int c, r, sg;
                    slot_geom(k, c, r, sg);
                    const int ci = min(c0 + c, AFF_MAX - 1);          // channels >= Cin hold 0 (loads returned 0 -> stays 0)
                    sc_ = s_aff[ci];
                    sh_ = s_aff[AFF_MAX + ci];
                    if (!interior) {
                        const bool ok = slot_inside(k, a.ty0, a.tx0);
                        sc_ = ok ? sc_ : 0.f;
                        sh_ = ok ? sh_ : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xform1<IN>(v[e], sc_, sh_, TWO ? rb[k][e] : 0.f);
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(s_in) + loff(k)) = v;
        }
    };
    auto flush_partials = [&](const Item& a, int co_base) {
        if (wave == 0) {
            for (int idx = lane; idx < 2 * NTB * 16; idx += 64) {
                const int q = idx / (NTB * 16), c = idx - q * (NTB * 16);
                const float sum = ((s_red[(0 * 2 + q) * NTB * 16 + c] + s_red[(1 * 2 + q) * NTB * 16 + c]) + s_red[(2 * 2 + q) * NTB * 16 + c]) + s_red[(3 * 2 + q) * NTB * 16 + c];
                const size_t row = (size_t)a.tile * d.B + a.b;
                if (co_base + c < Cout) d.partial[(row * 2 + q) * Cout + co_base + c] = sum;
            }
        }
    };

    // Weights that do not stay resident (Cin * NTB too large for the LDS budget) are restaged per (tile, chunk) stage; like the
    // input tile they are prefetched into registers under the MFMA phase of the previous stage.  Row r = (tap, q, n) of the
    // stage's B fragments is wave-uniform (wave w owns rows w, w + 4, ...): its decomposition is scalar arithmetic.
    // Wave w owns channel quad q = w of the chunk: row j of its set is (tap, n) = (j / NTB, j % NTB) -- compile-time, so the
    // prefetch needs no index arithmetic at all (a per-row runtime decomposition cost ~100 spilled VGPRs here).
    constexpr bool WPRE = !TWO && NTB >= 2;                // (single-tile layers and the tanh-grad prologue: resident weights only)
    constexpr int NWR = WPRE ? G::T * NTB : 1;
    float wr[NWR];
    const int w_co_step = d.transposed ? G::T : d.wCi * G::T;
    const int w_ci_step = d.transposed ? d.wCi * G::T : G::T;
    const int w_tap0 = d.transposed ? G::T - 1 : 0, w_tapstep = d.transposed ? -1 : 1;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(d.w, 0, (unsigned)((size_t)d.wCo * d.wCi * G::T * 4));
    auto w_issue = [&](int co_b, int q0, int nq) {         // one VGPR offset per lane, the (tap, n) part in the scalar offset
        const int co_l = co_b + li, ci_l = (q0 + wave) * 4 + kq;
        const bool ok = wave < nq && ci_l < Cin;
        const unsigned vo = (unsigned)((d.transposed ? (ci_l * d.wCi + co_l) * G::T : (co_l * d.wCi + ci_l) * G::T) * 4);
#pragma unroll
        for (int j = 0; j < NWR; ++j) {
            const int tap = j / NTB, n = j % NTB;
            const unsigned so = (unsigned)((16 * n * w_co_step + w_tap0 + tap * w_tapstep) * 4);
            wr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, (int)((ok && co_l + 16 * n < Cout) ? vo : OOB), (int)so, 0));
        }
    };
    auto w_commit = [&](int nq) {
        if (wave < nq) {
            float* dst = s_w + wave * NTB * 64 + lane;
#pragma unroll
            for (int j = 0; j < NWR; ++j) dst[((j / NTB) * NQ1 * NTB + (j % NTB)) * 64] = wr[j];
        }
    };
    auto chunk_nq = [&](int c0) { return (min(NCH, Cin - c0) + 3) >> 2; };

    Item it = decode_item(ka, itx);
    int cur_g = -1, aff_b = -1;
    if constexpr (AFF) { load_affine(it.b); aff_b = it.b; }
    issue(it, 0);
    if (WPRE && !ka.w_resident) { w_issue(it.g * NTB * 16, 0, chunk_nq(0)); w_commit(chunk_nq(0)); }
    commit(it, 0);
    const int abase = kq * G::PLANE + (2 * wave) * G::RS + li + G::COL0;
    Item prev = it;
    bool have_prev = false;
    float scl[NTB];
#pragma unroll
    for (int n = 0; n < NTB; ++n) scl[n] = 0.f;

    for (; itx < r1; itx += nlb) {
        const int co_base = it.g * NTB * 16;
        f32x4 acc[4][NTB];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ka.w_resident && cur_g != it.g) {
            lds_barrier();
            stage_weights<KS, NTB>(d, s_w, co_base, 0, ka.nq_total, qstride);
            cur_g = it.g;
        }
        const bool has_next = itx + nlb < r1;
        Item nxt = it;
        if (has_next) nxt = decode_item(ka, itx + nlb);

        for (int ch = 0; ch < nchunks; ++ch) {
            const int c0 = ch * NCH;
            const int nq = (min(NCH, Cin - c0) + 3) >> 2;
            const bool last_chunk = ch == nchunks - 1;
            lds_barrier();                                 // (A) this stage's s_in / s_w (and s_red of the previous item) visible
            if (!last_chunk) {                             // next stage's loads fly under the MFMA phase
                issue(it, c0 + NCH);
                if (WPRE && !ka.w_resident) w_issue(co_base, (c0 + NCH) >> 2, chunk_nq(c0 + NCH));
            } else if (has_next) {
                issue(nxt, 0);
                if (WPRE && !ka.w_resident) w_issue(nxt.g * NTB * 16, 0, chunk_nq(0));
            }
            if constexpr (RED) { if (ch == 0 && have_prev) flush_partials(prev, prev.g * NTB * 16); }
            const int qb = ka.w_resident ? (c0 >> 2) : 0;
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky) {
                const float* a_row = s_in + abase + ky * G::RS;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float* b_row = s_w + ((ky * KS + kx) * qstride + qb) * NTB * 64 + lane;
#pragma unroll
                    for (int q = 0; q < NQ1; ++q) {
                        if (q < nq) {
                            float af[4], bf[NTB];
#pragma unroll
                            for (int n = 0; n < NTB; ++n) bf[n] = b_row[(q * NTB + n) * 64];
#pragma unroll
                            for (int m = 0; m < 4; ++m) af[m] = a_row[q * 4 * G::PLANE + (m >> 1) * G::RS + (m & 1) * 16 + kx];
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int n = 0; n < NTB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
                        }
                    }
                }
            }
            lds_barrier();                                 // (B) every wave is done reading this stage
            if (!last_chunk) {
                if (WPRE && !ka.w_resident) w_commit(chunk_nq(c0 + NCH));
                commit(it, c0 + NCH);
            } else if (has_next) {
                if constexpr (AFF) { if (nxt.b != aff_b) { load_affine(nxt.b); aff_b = nxt.b; } }
                if (WPRE && !ka.w_resident) w_commit(chunk_nq(0));
                commit(nxt, 0);
            }
        }

        // ---- epilogue straight from the accumulators, one cout tile after the other
        {
            const int ty0 = it.ty0, tx0 = it.tx0;
            const bool full = ty0 + TH <= H && tx0 + TW <= W;
            if constexpr (RED) {
                {
#pragma unroll
                    for (int n = 0; n < NTB; ++n) { const int co = co_base + n * 16 + li; scl[n] = co < Cout ? 1.0f + d.scale[it.b * Cout + co] : 0.f; }
                }
            }
#pragma unroll
            for (int n = 0; n < NTB; ++n) {
                const int co = co_base + n * 16 + li;
                if (co_base + n * 16 >= Cout) continue;    // uniform
                const unsigned ovoff = co < Cout ? (unsigned)(((co * H) * W + 4 * kq) * 4) : OOB;
                const float bias_l = (EP != BNERV_EP_PLAIN && !RED && d.bias && co < Cout) ? d.bias[co] : 0.f;
                const unsigned ob = (unsigned)((((it.b * Cout) * H + ty0 + 2 * wave) * W + tx0) * 4);
                unsigned so[4], vo[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    so[m] = ob + (unsigned)(((m >> 1) * W + (m & 1) * 16) * 4);
                    vo[m] = ovoff;
                    if (!full) {
                        const bool ok = ty0 + 2 * wave + (m >> 1) < H && tx0 + (m & 1) * 16 + 4 * kq < W;
                        vo[m] = ok ? ovoff : OOB;
                        if constexpr (RED) { if (!ok) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    }
                }
                if constexpr (EP == BNERV_EP_BIAS || EP == BNERV_EP_PLAIN) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) bstore(ro, vo[m], so[m], acc[m][n] + bias_l);
                } else if constexpr (EP == BNERV_EP_BIAS_GELU) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x4 hv, gv;
                        gelu_pair4_f(acc[m][n] + bias_l, &hv, &gv);
                        bstore(ro, vo[m], so[m], hv);
                        if (d.out2) bstore(ro2, vo[m], so[m], gv);
                    }
                } else if constexpr (EP == BNERV_EP_BIAS_SIN) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x4 sv, cv;
                        sincos4_f(acc[m][n] + bias_l, &sv, &cv);
                        bstore(ro, vo[m], so[m], sv);
                        if (d.out2) bstore(ro2, vo[m], so[m], cv);
                    }
                } else if constexpr (EP == BNERV_EP_BIAS_TANH) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = tanhf(acc[m][n][e] + bias_l) * 0.5f + 0.5f;
                        bstore(ro, vo[m], so[m], r);
                    }
                } else if constexpr (EP == BNERV_EP_BIAS_RES) {
                    f32x4 a0[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) a0[m] = bload(ra0, vo[m], so[m]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) bstore(ro, vo[m], so[m], acc[m][n] + bias_l + a0[m]);
                } else {                                   // DGELU / DGELU_SAVED / DSIN
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
                        const f32x4 v = acc[m][n];
                        if constexpr (EP == BNERV_EP_DGELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl[n] * gelu_grad_f(a0[m][e]); ps = fmaf(v[e], gelu_f(a0[m][e]), ps); pt += v[e]; }
                        } else if constexpr (EP == BNERV_EP_DGELU_SAVED) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = v[e] * scl[n] * a0[m][e]; ps = fmaf(v[e], a1[m][e], ps); pt += v[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[e] = (a1[m][e] + v[e] * scl[n]) * a2[m][e]; ps = fmaf(v[e], a0[m][e], ps); pt += v[e]; }
                        }
                        bstore(ro, vo[m], so[m], r);
                    }
                    ps += __shfl_xor(ps, 16, 64);
                    pt += __shfl_xor(pt, 16, 64);
                    ps += __shfl_xor(ps, 32, 64);
                    pt += __shfl_xor(pt, 32, 64);
                    if (lane < 16) { s_red[(wave * 2 + 0) * NTB * 16 + n * 16 + lane] = ps; s_red[(wave * 2 + 1) * NTB * 16 + n * 16 + lane] = pt; }
                }
            }
        }
        prev = it;
        have_prev = true;
        it = nxt;
    }
    if constexpr (RED) {
        lds_barrier();
        flush_partials(prev, prev.g * NTB * 16);
    }
    side_run_hosted(side, smem);
}

template <int KS, int IN, int EP, int NTB>
int launch_lean2(hipStream_t st, KArgs& ka) {
    using G = Geo<KS>;
    const bnerv_conv_desc& d = ka.d;
    constexpr int NSLOT = 16 * G::ROWS * G::SEGS;
    constexpr int NPRE = (NSLOT + 255) / 256;
    const int nt = cdiv(d.Cout, 16);
    ka.ngroups = cdiv(nt, NTB);
    ka.total_items = ka.ngroups * d.B * ka.tiles_x * ka.tiles_y;
    ka.nq_total = cdiv(d.Cin, 16) * 4;                     // chunk-aligned: chunk ch owns q = 4 ch .. 4 ch + 3
    const size_t wres = (size_t)G::T * ka.nq_total * NTB * 64;
    ka.w_resident = wres <= (size_t)W_RESIDENT_MAX ? 1 : 0;
    if ((IN == BNERV_IN_TANHGRAD || NTB == 1) && !ka.w_resident) return -1;       // caller falls back to the generic kernel
    const size_t wfl = ka.w_resident ? wres : (size_t)G::T * 4 * NTB * 64;
    const size_t lds = ((size_t)16 * G::PLANE + (size_t)(NPRE * 256 - NSLOT) * 4 + (size_t)4 * 2 * NTB * 16 + 2 * 128 + wfl) * sizeof(float);   // (dump area unused by the unshuffle path)
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lean2_kernel<KS, IN, EP, NTB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_lean2_kernel<KS, IN, EP, NTB>), 256, lds) != hipSuccess || nb < 1) nb = 1;
    int grid = 256 * (nb > 4 ? 4 : nb);
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(ka.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_lean2_kernel<KS, IN, EP, NTB>), dim3(grid), dim3(256), lds, st, ka, side);
    BNERV_LAUNCH_CHECK("conv_lean2");
    return BNERV_OK;
}

static bool lean2_ok(const KArgs& ka) {
    const bnerv_conv_desc& d = ka.d;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    static const bool off = getenv("BNERV_NO_LEAN2") != nullptr;          // A/B switch for tools/kbench.py
    if (off) return false;
    if (d.in_mode == BNERV_IN_UNSHUFFLE && (d.in_s != 2 || d.Cin % 4 != 0)) return false;
    const bool affine = d.in_mode == BNERV_IN_AFFINE || d.in_mode == BNERV_IN_GELU_AFFINE;   // (the LDS table of scale/shift holds 128 channels)
    return ka.vec && d.out_s == 1 && d.k == 3 && (d.Cin > 16 || d.Cout > 16) && (d.Cin <= 128 || !affine) && ka.ksplit == 1 &&
           (size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 < LEAN_MAX_BYTES;
}

static bool lean_ok(const KArgs& ka) {
    const bnerv_conv_desc& d = ka.d;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    return ka.vec && d.out_s == 1 && d.Cout <= 16 && d.Cin <= CC && d.Cin > 8 &&
           (size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 < LEAN_MAX_BYTES;
}

template <int KS, int IN, int EP, int NTB>
int launch_one(hipStream_t st, KArgs& ka) {
    if constexpr (KS == 3 && (NTB <= 3 || (NTB == 4 && EP != BNERV_EP_BIAS_SIN)) && (IN != BNERV_IN_UNSHUFFLE || EP == BNERV_EP_PLAIN) &&   // (4 tiles + sincos spills)
                  (EP == BNERV_EP_BIAS || EP == BNERV_EP_BIAS_SIN || EP == BNERV_EP_BIAS_GELU || EP == BNERV_EP_BIAS_RES || EP == BNERV_EP_BIAS_TANH || EP == BNERV_EP_PLAIN ||
                   EP == BNERV_EP_DGELU_SAVED || EP == BNERV_EP_DSIN)) {
        if (lean2_ok(ka)) { const int rc = launch_lean2<KS, IN, EP, NTB>(st, ka); if (rc != -1) return rc; }
    }
    if constexpr (IN != BNERV_IN_UNSHUFFLE && NTB == 1) {
        if (lean_ok(ka)) {
            if (ka.d.Cin <= 12) return launch_lean<KS, IN, EP, 3>(st, ka);
            return launch_lean<KS, IN, EP, 4>(st, ka);
        }
    }
    // fast path: instantiated for the shapes the decoders actually have at high resolution (Cin 9..16; Cout <= 16 or 33..48)
    if constexpr (IN != BNERV_IN_UNSHUFFLE && (NTB == 1 || NTB == 3)) {
        if (ka.vec && ka.d.Cin <= CC && ka.d.Cin > 8) {
            if (ka.d.Cin <= 12) return launch_fast<KS, IN, EP, NTB, 3>(st, ka);
            return launch_fast<KS, IN, EP, NTB, 4>(st, ka);
        }
    }
    using G = Geo<KS>;
    const bnerv_conv_desc& d = ka.d;
    const int nt = cdiv(d.Cout, 16);
    ka.ngroups = cdiv(nt, NTB);
    ka.total_items = ka.ngroups * d.B * ka.tiles_x * ka.tiles_y * ka.ksplit;
    ka.nq_total = cdiv(d.Cin, 4);
    const size_t wres = (size_t)G::T * ka.nq_total * NTB * 64;
    ka.w_resident = wres <= (size_t)W_RESIDENT_MAX ? 1 : 0;
    const size_t wfl = ka.w_resident ? wres : (size_t)G::T * NQ * NTB * 64;
    const size_t lds = ((size_t)CC * G::PLANE + (size_t)16 * CS + wfl) * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<KS, IN, EP, NTB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    const int per_cu = (int)((size_t)160 * 1024 / lds);
    int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    if (grid > ka.total_items) grid = ka.total_items;
    SidePack side;
    bnerv_side_take(ka.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((conv_igemm_kernel<KS, IN, EP, NTB>), dim3(grid), dim3(256), lds, st, ka, side);
    BNERV_LAUNCH_CHECK("conv_igemm");
    return BNERV_OK;
}

template <int KS, int IN, int EP>
int launch_ntb(hipStream_t st, KArgs& ka) {
    const int nt = cdiv(ka.d.Cout, 16);
    // cout tiles per block: as many as fit (<= 4, the input tile is staged once for all of them) -- unless that leaves most of
    // the 256 CUs idle (the low-resolution stages: a handful of spatial tiles), where more, smaller blocks cut the latency.
    int ntb = nt < 4 ? nt : 4;
    const int spatial = ka.d.B * ka.tiles_x * ka.tiles_y * ka.ksplit;
    while (ntb > 1 && cdiv(nt, ntb) * spatial < 256) --ntb;
    if (ntb == 1) return launch_one<KS, IN, EP, 1>(st, ka);
    if (ntb == 2) return launch_one<KS, IN, EP, 2>(st, ka);
    if (ntb == 3) return launch_one<KS, IN, EP, 3>(st, ka);
    return launch_one<KS, IN, EP, 4>(st, ka);
}

template <int KS>
int launch_mode(hipStream_t st, KArgs& ka) {
    const int in = ka.d.in_mode, ep = ka.d.ep_mode;
#define BNERV_CASE(I, E) if (in == I && ep == E) return launch_ntb<KS, I, E>(st, ka);
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_TANH)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_TANHGRAD, BNERV_EP_PLAIN)
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
    return bnerv_set_error(BNERV_E_ARG, "conv_igemm: unsupported (k=%d, in_mode=%d, ep_mode=%d)", KS, in, ep);
}

// Split-K policy: layers with a long K loop and almost no spatial parallelism (the low-resolution data gradients:
// Cin = 750 or 1975 at 9x16 = 2 tiles) spread the input-channel chunks over work items.  Only for EP_PLAIN, out_s == 1.
struct SplitPlan { int ksplit, chunks_per_split; };
SplitPlan plan_split(const bnerv_conv_desc& d) {
    SplitPlan p{1, 0};
    if (d.ep_mode != BNERV_EP_PLAIN || d.out_s != 1) return p;
    const int nchunks = cdiv(d.Cin, CC);
    const int nt = cdiv(d.Cout, 16);
    const int ngroups = cdiv(nt, nt >= 4 ? 4 : nt);
    const int items = ngroups * d.B * cdiv(d.H, TH) * cdiv(d.W, TW);
    if (nchunks < 8 || items >= 128) return p;
    int ks = 512 / items;
    if (ks > nchunks) ks = nchunks;
    if (ks < 2) return p;
    p.chunks_per_split = cdiv(nchunks, ks);
    p.ksplit = cdiv(nchunks, p.chunks_per_split);
    return p;
}

}  // namespace

#ifdef BNERV_TRACE
extern "C" int bnerv_debug_trace_read(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(g_trace)); }
#endif
extern "C" int bnerv_conv_tiles(int H, int W) { return cdiv(H, TH) * cdiv(W, TW); }

static int conv_vec_ok(const bnerv_conv_desc& d) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return ((d.W % 4 == 0) && al(d.x) && al(d.out) && al(d.out2) && al(d.aux0) && al(d.aux1) && al(d.aux2)) ? 1 : 0;
}

extern "C" int bnerv_conv_partial_rows(const bnerv_conv_desc* dp) {
    if (!dp || dp->H <= 0 || dp->W <= 0) return 0;
    bnerv_conv_desc d = *dp;
    if (d.in_mode == BNERV_IN_UNSHUFFLE && d.in_s == 1) d.in_mode = BNERV_IN_PLAIN;
    if (bnerv_convs_shape_ok(d, conv_vec_ok(d))) return bnerv_convs_tiles(d.H, d.W);      // the low-resolution family (convs.hip): 4x16 tiles
    return cdiv(d.H, TH) * cdiv(d.W, TW);     // every other kernel of this build uses 8x32 tiles; callers must still ask
}

// Data gradient of a 1x1 head (reference: head_layer 1x1 C->3 + OutImg tanh, model_nerv.py:41,56-57): 3 -> C channels with the
// tanh-grad prologue is 9*C FMAs per pixel against 6 loaded and C stored floats -- a streaming kernel, not a GEMM (the MFMA
// path spends 29 us on it at 720x1280 where the 66 MB of traffic need ~13).  One thread = 4 consecutive pixels.
namespace {
constexpr int HEAD_KMAX = 4, HEAD_CMAX = 16;
__global__ __launch_bounds__(256) void head1x1_dgrad_kernel(const bnerv_conv_desc d, const int hw4) {
    __shared__ float s_w[HEAD_KMAX * HEAD_CMAX];
    const int K = d.Cin, C = d.Cout;
    if ((int)threadIdx.x < K * C) {                        // W(k, c): transposed weights are stored [k][c], direct ones [c][k]
        const int k = threadIdx.x / C, c = threadIdx.x - k * C;
        s_w[k * HEAD_CMAX + c] = d.transposed ? d.w[k * C + c] : d.w[c * K + k];
    }
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (q >= hw4) return;
    const size_t HW = (size_t)d.H * d.W;
    f32x4 g[HEAD_KMAX];
#pragma unroll
    for (int k = 0; k < HEAD_KMAX; ++k) {
        g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < K) {
            const size_t o = ((size_t)b * K + k) * HW + (size_t)q * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(d.x + o), a = *reinterpret_cast<const f32x4*>(d.aux0 + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[k][e] = xform1<BNERV_IN_TANHGRAD>(v[e], 0.f, 0.f, a[e]);
        }
    }
    for (int c = 0; c < C; ++c) {
        f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < HEAD_KMAX; ++k)
            if (k < K) {
                const float wv = s_w[k * HEAD_CMAX + c];
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaf(g[k][e], wv, r[e]);
            }
        *reinterpret_cast<f32x4*>(d.out + ((size_t)b * C + c) * HW + (size_t)q * 4) = r;
    }
}
// The 1x1 output head's forward (head_layer 1x1 C -> K <= 4 + OutImg tanh: img = tanh(W x + b) * 0.5 + 0.5, model_nerv.py:56-57,
// model_blocks.py:57-63) as the same kind of streaming kernel: 12 loaded and 3 stored floats per pixel against 36 fma -- the implicit-GEMM
// kernel spent 16.2 us at 720x1280 on 55 MB of traffic.  One thread = 4 consecutive pixels, every load in flight before the first use.
template <int K, int C>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const bnerv_conv_desc d, const int hw4) {
    __shared__ float s_w[HEAD_KMAX * HEAD_CMAX + HEAD_KMAX];
    if ((int)threadIdx.x < K * C) s_w[threadIdx.x] = d.w[threadIdx.x];                      // [k][c] as stored (OIHW, 1x1)
    if ((int)threadIdx.x < K) s_w[K * C + threadIdx.x] = d.bias ? d.bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (q >= hw4) return;
    const size_t HW = (size_t)d.H * d.W;
    f32x4 xv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = *reinterpret_cast<const f32x4*>(d.x + ((size_t)b * C + c) * HW + (size_t)q * 4);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float wv = s_w[k * C + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaf(xv[c][e], wv, r[e]);
        }
        const float bv = s_w[K * C + k];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = tanhf(r[e] + bv) * 0.5f + 0.5f;
        *reinterpret_cast<f32x4*>(d.out + ((size_t)b * K + k) * HW + (size_t)q * 4) = r;
    }
}
}  // namespace

extern "C" size_t bnerv_conv_splitk_ws_bytes(const bnerv_conv_desc* dp) {
    if (!dp || dp->B <= 0 || dp->Cin <= 0 || dp->Cout <= 0 || dp->H <= 0 || dp->W <= 0) return 0;
    const SplitPlan p = plan_split(*dp);
    const size_t a = p.ksplit > 1 ? (size_t)p.ksplit * dp->B * dp->Cout * dp->H * dp->W * sizeof(float) : 0;
    const size_t b = bnerv_stem_dgrad_ws_bytes(*dp);       // the stem stage's data gradient (stem.hip) keeps one slab per 8 input channels
    return a > b ? a : b;
}

extern "C" int bnerv_conv_igemm(void* stream, const bnerv_conv_desc* dp) {
    BNERV_REQUIRE(dp != nullptr, "conv_igemm: null descriptor");
    KArgs ka;
    ka.d = *dp;
    bnerv_conv_desc& d = ka.d;
    BNERV_REQUIRE(d.k == 1 || d.k == 3, "conv_igemm: k must be 1 or 3 (got %d)", d.k);
    BNERV_REQUIRE(d.B > 0 && d.Cin > 0 && d.Cout > 0 && d.H > 0 && d.W > 0, "conv_igemm: bad dims");
    BNERV_REQUIRE(d.x && d.w && d.out, "conv_igemm: null tensor");
    BNERV_REQUIRE(d.in_s >= 1 && d.out_s >= 1, "conv_igemm: shuffle factors must be >= 1");
    BNERV_REQUIRE(d.Cout % (d.out_s * d.out_s) == 0, "conv_igemm: Cout %d not divisible by out_s^2", d.Cout);
    if (d.in_mode == BNERV_IN_UNSHUFFLE) BNERV_REQUIRE(d.Cin % (d.in_s * d.in_s) == 0, "conv_igemm: Cin %d not divisible by in_s^2", d.Cin);
    if (d.transposed) BNERV_REQUIRE(d.Cout == d.wCi && d.Cin == d.wCo, "conv_igemm: transposed dims mismatch");
    else BNERV_REQUIRE(d.Cout == d.wCo && d.Cin == d.wCi, "conv_igemm: weight dims mismatch");
    if (d.in_mode == BNERV_IN_AFFINE || d.in_mode == BNERV_IN_GELU_AFFINE) BNERV_REQUIRE(d.scale && d.shift, "conv_igemm: affine prologue needs scale/shift");
    if (d.in_mode == BNERV_IN_TANHGRAD) BNERV_REQUIRE(d.aux0, "conv_igemm: tanh-grad prologue needs aux0");
    if (d.ep_mode == BNERV_EP_BIAS_RES) BNERV_REQUIRE(d.aux0, "conv_igemm: residual epilogue needs aux0");
    if (d.ep_mode == BNERV_EP_DGELU) BNERV_REQUIRE(d.aux0 && d.scale && d.partial && d.out_s == 1, "conv_igemm: DGELU epilogue args");
    if (d.ep_mode == BNERV_EP_DSIN) BNERV_REQUIRE(d.aux0 && d.aux1 && d.scale && d.partial && d.out_s == 1, "conv_igemm: DSIN epilogue args");
    if (d.ep_mode == BNERV_EP_DGELU_SAVED) BNERV_REQUIRE(d.aux0 && d.aux1 && d.scale && d.partial && d.out_s == 1, "conv_igemm: DGELU_SAVED epilogue args");
    if (d.ep_mode == BNERV_EP_BIAS_GELU) BNERV_REQUIRE(d.out_s == 1, "conv_igemm: BIAS_GELU epilogue needs a stride-1 output");   // out2 NULL: gelu only
    if (d.in_mode == BNERV_IN_UNSHUFFLE && d.in_s == 1) d.in_mode = BNERV_IN_PLAIN;       // same gather, faster staging
    ka.tiles_x = cdiv(d.W, TW);
    ka.tiles_y = cdiv(d.H, TH);
    // float4 paths need 16-B aligned rows: W % 4 == 0 and 16-B aligned base pointers (NULL counts as aligned)
    ka.vec = conv_vec_ok(d);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ka.ksplit = 1;
    ka.chunks_per_split = 0;
    ka.magic_tiles = ka.magic_tiles_x = 0;
    if (d.ep_mode == BNERV_EP_PLAIN && d.partial != nullptr) {            // caller supplied a split-K workspace
        const SplitPlan p = plan_split(d);
        ka.ksplit = p.ksplit;
        ka.chunks_per_split = p.chunks_per_split;
    }
    {
        const char* hf = getenv("BNERV_HEAD_FWD");         // A/B switch, read per call
        if (!(hf && hf[0] == '0') && d.k == 1 && d.in_mode == BNERV_IN_PLAIN && d.ep_mode == BNERV_EP_BIAS_TANH && d.out_s == 1 && ka.vec && !d.transposed &&
            d.Cin == 12 && d.Cout == 3 && d.wCo == 3 && d.wCi == 12 && ((size_t)d.H * d.W) % 4 == 0 && d.B <= 65535) {
            const int hw4 = (int)(((size_t)d.H * d.W) / 4);
            hipLaunchKernelGGL((head1x1_fwd_kernel<3, 12>), dim3(cdiv(hw4, 256), d.B), dim3(256), 0, st, d, hw4);
            BNERV_LAUNCH_CHECK("head1x1_fwd");
            return BNERV_OK;
        }
    }
    if (d.k == 1 && d.in_mode == BNERV_IN_TANHGRAD && d.ep_mode == BNERV_EP_PLAIN && d.out_s == 1 && ka.vec && ka.ksplit == 1 &&
        d.Cin <= HEAD_KMAX && d.Cout <= HEAD_CMAX && ((size_t)d.H * d.W) % 4 == 0 && d.B <= 65535) {
        const int hw4 = (int)(((size_t)d.H * d.W) / 4);
        hipLaunchKernelGGL(head1x1_dgrad_kernel, dim3(cdiv(hw4, 256), d.B), dim3(256), 0, st, d, hw4);
        BNERV_LAUNCH_CHECK("head1x1_dgrad");
        return BNERV_OK;
    }
    if (d.k == 3 && (d.Cout == 3 || d.Cin == 3) && ka.ksplit == 1) {      // HNeRV-boost's 3x3 head and its data gradient: streaming VALU kernels
        const int rh = bnerv_head3_try(st, d);
        if (rh != 1) return rh;
    }
    if (d.ep_mode == BNERV_EP_PLAIN && d.partial != nullptr) {            // tiny image, long K (the stem up-conv's data gradient)
        const int rs = bnerv_stem_dgrad_try(st, d);
        if (rs != 1) return rs;
    }
    {   // the low-resolution stages (convs.hip) first: small images, <= 32 input channels
        const int rs = bnerv_convs_try(st, d, ka.vec, ka.ksplit);
        if (rs != 1) return rs;
    }
    // split-bf16 kernels (convbf.hip) next: the wide layers (with the same split-K plan: its slabs are reduced below), opt-in 12-channel ones
    int rc = bnerv_convbf_try(st, d, ka.vec, ka.ksplit, ka.chunks_per_split);
    if (rc == 1) {                                         // +1: not the split kernels' layer (negative values are real errors)
        ka.magic_tiles = ka.magic_tiles_x = 0;
        rc = bnerv_conv4_try(st, ka);                      // <= 12-channel 3x3 layers on the 4x4x1 MFMA family
        if (rc == 1) rc = d.k == 1 ? launch_mode<1>(st, ka) : launch_mode<3>(st, ka);
    }
    if (rc != BNERV_OK || ka.ksplit == 1) return rc;
    return bnerv_reduce_slabs(stream, d.partial, ka.ksplit, d.B * d.Cout * d.H * d.W, d.out);
}
