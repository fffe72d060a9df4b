// wgrad.hip -- weight + bias gradient of the decoder convolutions as an MFMA 16x16x4 GEMM with K = pixels.
//
//   dw[co][n] = sum_{b, p} g[b][co][p] * a[b][ci(n)][p + tap(n) - pad],   n = ci*T + tap   (the OIHW flattening)
//   db[co]    = the extra column n = Cin*T, whose B operand reads a plane of ones
// (autograd's backward of F.conv2d wrt weight/bias at lib/quant_ops.py:39-41; call sites as in conv.hip.)
//
// GEMM view: M = cout (MTW tiles of 16), N = (ci,tap)+bias column (NTW tiles of 16), K = pixels of an 8x32 spatial tile
// (4 per MFMA: lane k-slot kq <-> pixel 4*step + kq).  PERSISTENT blocks walk spatial tiles with the accumulators in
// registers; the 4 waves of a block split each tile's 64 K-steps; at the end the waves are summed through LDS and the block
// writes ONE slab; wgrad_finish sums the slabs in a fixed order (deterministic, no atomics, no memset).
//   LDS  s_g  [MTW*16][256 + 2]    gradient tile, cout-major; stride == 2 (mod 32) -> conflict-free A reads
//        s_in [planes][ROWS][RS]   halo tile of a = prologue(x), RS = 40 (aligned float4 segments, as conv.hip); plane stride
//                                  == 4 (mod 32) so the 16 (ci,tap) columns x 2 k-lanes of a B read fall on distinct banks;
//                                  + a plane of ones (bias column) + a plane of zeros (columns beyond the weight matrix)
// The K loop is fully unrolled: every ds_read has an immediate offset, the body is (MTW + NTW) ds_read_b32 + MTW*NTW MFMA.
// Staging is float4 from global; for the single-cout-tile shapes (all the 12-channel layers) the loads of tile t+1 are issued
// before the MFMA phase of tile t and committed to LDS after it.
#include "common.h"
#include "sidejob.h"
#include "split16.h"
#include "conv4_body.h"     // the 4x4x1 conv body: paired conv + weight-gradient launches (bnerv_conv_wgrad_pair, bottom of this file)
#include "convs_body.h"     // the low-resolution conv body, for the same
#if defined(BNERV_TRACE) || defined(BNERV_TRACE_BW)   // debug variants only (tools/ktrace_w.py, ktrace_bw.py): s_memtime phase stamps
static __device__ unsigned long long g_trace_w[1024 * 4 * 8 * 8];
#endif
#ifdef BNERV_TRACE_BW
#define BTRACE(it_, slot) do { if (lane == 0 && blockIdx.x < 1024 && (unsigned)(it_) < 8u) g_trace_w[((blockIdx.x * 4 + wave) * 8 + (it_)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "wgrad_bfw_body.h" // WArgs, the buffer-load helpers and the split wide weight-gradient body (also one role of convbf.hip's paired launch)
int bnerv_convbf_pair_try(hipStream_t st, const bnerv_conv_desc& d, int vec, const bnerv_wb::WArgs& wa, int w_mtw, int ngn, int ngm, int nat_slots, int* n_slabs);   // convbf.hip
bool bnerv_convs_shape_ok(const bnerv_conv_desc& d, int vec);     // convs.hip
int bnerv_stem_wgrad_try(hipStream_t st, const bnerv_wgrad_desc& d);   // stem.hip (images of <= 256 pixels): 1 = not that layer
int bnerv_stem_pair_try(hipStream_t st, const bnerv_conv_desc& c, const bnerv_wgrad_desc& d, int* n_slabs);   // stem.hip: the stem stage's (dW | d input) as one launch
int bnerv_wgrad1x1_slabs(const bnerv_wgrad_desc& d);                                   // wgrad1.hip: pointwise (k = 1) layers as a plain GEMM over the pixels
int bnerv_wgrad1x1_try(hipStream_t st, const bnerv_wgrad_desc& d, int* n_slabs);
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace {
using namespace bnerv_wb;

constexpr int TH = 8, TW = 32;
constexpr int CSG = TH * TW + 2;

template <int KS> struct Geo {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int ROWS = TH + 2 * PAD;
    static constexpr int XOFF = (KS == 3) ? 4 : 0;
    static constexpr int RS = TW + 2 * XOFF;                   // 40 / 32
    static constexpr int SEGS = RS / 4;
    static constexpr int PLANE_RAW = ROWS * RS;                // 400 / 256
    // 3x3: == 4 (mod 32): columns (ci, ky, kx)+kq map to banks 4*ci' + 8*ky + kx + kq
    // 1x1: == 2 (mod 32): 16 consecutive channels x 2 k-lanes -> 32 distinct banks
    static constexpr int PLANE = (KS == 3) ? 420 : 258;
    static constexpr int T = KS * KS;
    static constexpr int COL0 = XOFF - PAD;
    static constexpr int NPL = (KS == 3) ? 14 : 16;            // data planes per block; + 1 plane of ones + 1 of zeros
};


template <int IN>
__device__ __forceinline__ float xf_in(float v, float sc, float sh) {
    if constexpr (IN == BNERV_IN_AFFINE) return v * sc + sh;
    if constexpr (IN == BNERV_IN_GELU_AFFINE) return gelu_f(v) * sc + sh;
    return v;
}

template <int GM>
__device__ __forceinline__ float load_g_scalar(const bnerv_wgrad_desc& d, int b, int co, int gy, int gx) {
    if constexpr (GM == BNERV_IN_TANHGRAD) {
        const size_t idx = (((size_t)b * d.Cout + co) * d.H + gy) * (size_t)d.W + gx;
        const float t = 2.0f * d.gaux[idx] - 1.0f;
        return d.g[idx] * 0.5f * (1.0f - t * t);
    } else {
        const int s = d.g_s;
        if (s == 1) return d.g[(((size_t)b * d.Cout + co) * d.H + gy) * (size_t)d.W + gx];
        const int s2 = s * s, c = co / s2, rem = co - c * s2, i = rem / s, j = rem - i * s;
        return d.g[(((size_t)b * (d.Cout / s2) + c) * (size_t)(d.H * s) + (size_t)(gy * s + i)) * (size_t)(d.W * s) + (size_t)(gx * s + j)];
    }
}

// input planes spanned by NTW*16 consecutive (ci, tap) columns starting anywhere
template <int KS, int NTW> constexpr int wgrad_npl() { return KS == 3 ? (NTW * 16 + 7) / 9 + 1 : NTW * 16; }

template <int KS, int IN, int GM, int MTW, int NTW>
__global__ __launch_bounds__(256, (MTW * NTW <= 7 ? 3 : 2)) void conv_wgrad_kernel(const WArgs wa) {
    using G = Geo<KS>;
    constexpr bool PIPE = (MTW == 1);                          // register-prefetch the next tile under the MFMA phase
    constexpr int NPL = wgrad_npl<KS, NTW>();                  // data planes (input channels) a block's NTW*16 columns can touch
    constexpr bool GTWO = true;                                // GM is UNSHUFFLE or TANHGRAD: both may need a second float4
    const bnerv_wgrad_desc& d = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                                  // g_rows * CSG: rows beyond Cout are not kept -- the A reads of those rows see
                                                        // s_in data, which only reaches output rows that are dropped

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int mg = blockIdx.y / wa.n_ngroups, ngp = blockIdx.y % wa.n_ngroups;
    const int co_base = mg * MTW * 16;
    const int n_base = ngp * NTW * 16;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int g_rows = min(MTW * 16, Cout - co_base);
    float* s_in = smem + g_rows * CSG;                  // (NPL + 2) * PLANE
    const int nW = Cin * G::T;                          // weight columns; column nW is the bias column
    const int ci_lo = min(n_base, nW - 1) / G::T;
    const int ci_hi = min(n_base + NTW * 16 - 1, nW - 1) / G::T;
    const int npl = ci_hi - ci_lo + 1;                  // <= NPL by construction of NTW
    const bool vec_x = wa.vec != 0;
    const bool vec_g = wa.vec != 0 && (GM == BNERV_IN_TANHGRAD || d.g_s <= 2);
    const bool pair = (GM == BNERV_IN_UNSHUFFLE) && d.g_s == 2;
    const int ng_slots = (pair ? MTW * 8 : MTW * 16) * TH * (TW / 4);
    const int nx_slots = npl * G::ROWS * G::SEGS;

    // constant planes: ones (bias column) and zeros (columns beyond the matrix)
    for (int i = tid; i < 2 * G::PLANE; i += 256) s_in[NPL * G::PLANE + i] = i < G::PLANE ? 1.0f : 0.0f;

    // per-lane fragment bases.  pixel of (wave, step, kq): p = wave*64 + step*4 + kq
    int bbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = n_base + nt * 16 + li;
        int off;
        if (n < nW) {
            const int ci = n / G::T, tap = n - ci * G::T;
            off = (ci - ci_lo) * G::PLANE + (tap / KS) * G::RS + (tap % KS) + G::COL0;
        } else {
            off = (n == nW ? NPL : NPL + 1) * G::PLANE;
        }
        bbase[nt] = off + (2 * wave) * G::RS + kq;
    }
    int abase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) abase[m] = (m * 16 + li) * CSG + wave * 64 + kq;

    // ---- slot-level staging primitives (float4) ----
    // g: slot = (cout, row, 4-px segment)   [pair: slot = (cout PAIR (c,i; j=0,1), row, segment): two float4 of the shuffled row]
    auto g_load = [&](int sidx, int b, int ty0, int tx0, f32x4& va, f32x4& vb) {
        va = f32x4{0.f, 0.f, 0.f, 0.f}; vb = va;
        if (sidx >= ng_slots) return;
        const int c = sidx >> 6, rem = sidx & 63, r = rem >> 3, sg = rem & 7;
        const int gy = ty0 + r, gx = tx0 + 4 * sg;
        if (gy >= H || gx >= W) return;
        if (pair) {
            const int co = co_base + 2 * c;
            if (co < Cout) {
                const int cf = co >> 2, i = (co >> 1) & 1;
                const float* src = d.g + (((size_t)b * (Cout >> 2) + cf) * (size_t)(2 * H) + (size_t)(2 * gy + i)) * (size_t)(2 * W) + (size_t)(2 * gx);
                va = *reinterpret_cast<const f32x4*>(src);
                vb = *reinterpret_cast<const f32x4*>(src + 4);
            }
        } else {
            const int co = co_base + c;
            if (co < Cout) {
                const size_t idx = (((size_t)b * Cout + co) * H + gy) * (size_t)W + gx;
                va = *reinterpret_cast<const f32x4*>(d.g + idx);
                if constexpr (GM == BNERV_IN_TANHGRAD) vb = *reinterpret_cast<const f32x4*>(d.gaux + idx);
            }
        }
    };
    auto g_store = [&](int sidx, f32x4 a, f32x4 bq) {
        if (sidx >= ng_slots) return;
        const int c = sidx >> 6, rem = sidx & 63, r = rem >> 3, sg = rem & 7;
        if ((pair ? 2 * c : c) >= g_rows) return;          // (Cout is even on the pair path)
        if (pair) {
            float* dst = s_g + (2 * c) * CSG + r * TW + 4 * sg;                 // 8-B aligned (CSG*4 = 1032)
            *reinterpret_cast<float2*>(dst) = float2{a[0], a[2]};
            *reinterpret_cast<float2*>(dst + 2) = float2{bq[0], bq[2]};
            *reinterpret_cast<float2*>(dst + CSG) = float2{a[1], a[3]};
            *reinterpret_cast<float2*>(dst + CSG + 2) = float2{bq[1], bq[3]};
        } else {
            if constexpr (GM == BNERV_IN_TANHGRAD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = 2.0f * bq[e] - 1.0f; a[e] = a[e] * 0.5f * (1.0f - t * t); }
            }
            float* dst = s_g + c * CSG + r * TW + 4 * sg;
            *reinterpret_cast<float2*>(dst) = float2{a[0], a[1]};
            *reinterpret_cast<float2*>(dst + 2) = float2{a[2], a[3]};
        }
    };
    auto x_load = [&](int sidx, int b, int ty0, int tx0, f32x4& v) {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sidx >= nx_slots) return;
        const int c = sidx / (G::ROWS * G::SEGS);
        const int rem = sidx - c * (G::ROWS * G::SEGS);
        const int r = rem / G::SEGS, sg = rem - r * G::SEGS;
        const int gy = ty0 + r - G::PAD, gx = tx0 - G::XOFF + 4 * sg;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = *reinterpret_cast<const f32x4*>(d.x + (((size_t)b * Cin + ci_lo + c) * H + gy) * (size_t)W + gx);
    };
    auto x_store = [&](int sidx, int b, int ty0, int tx0, f32x4 v) {
        if (sidx >= nx_slots) return;
        const int c = sidx / (G::ROWS * G::SEGS);
        const int rem = sidx - c * (G::ROWS * G::SEGS);
        const int r = rem / G::SEGS, sg = rem - r * G::SEGS;
        if constexpr (IN != BNERV_IN_PLAIN) {
            const int gy = ty0 + r - G::PAD, gx = tx0 - G::XOFF + 4 * sg;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {                      // padding AFTER the prologue
                const float sc = 1.0f + d.scale[b * Cin + ci_lo + c], sh = d.shift[b * Cin + ci_lo + c];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xf_in<IN>(v[e], sc, sh);
            }
        }
        float* dst = s_in + c * G::PLANE + r * G::RS + 4 * sg;                  // PLANE*4 = 1680 / 1032 bytes: 8-B aligned
        *reinterpret_cast<float2*>(dst) = float2{v[0], v[1]};
        *reinterpret_cast<float2*>(dst + 2) = float2{v[2], v[3]};
    };
    // scalar fallbacks (W % 4 != 0, or pixel-unshuffle factors 3 / 5): low-resolution layers only
    // (scalar paths: loads are issued in batches of 8 before the LDS stores, otherwise every element costs a full memory latency)
    auto g_stage_scalar = [&](int b, int ty0, int tx0) {
        for (int i0 = tid; i0 < MTW * 16 * TH * TW; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i0 + u * 256;
                const int cl = idx >> 8, p = idx & 255;
                const int co = co_base + cl, gy = ty0 + (p >> 5), gx = tx0 + (p & 31);
                v[u] = 0.f;
                if (idx < MTW * 16 * TH * TW && co < Cout && gy < H && gx < W) v[u] = load_g_scalar<GM>(d, b, co, gy, gx);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i0 + u * 256;
                if (idx < MTW * 16 * TH * TW && (idx >> 8) < g_rows) s_g[(idx >> 8) * CSG + (idx & 255)] = v[u];
            }
        }
    };
    auto x_stage_scalar = [&](int b, int ty0, int tx0) {
        const int n_el = npl * G::PLANE_RAW;
        for (int i0 = tid; i0 < n_el; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i0 + u * 256;
                const int c = idx / G::PLANE_RAW;
                const int rem = idx - c * G::PLANE_RAW;
                const int r = rem / G::RS, col = rem - r * G::RS;
                const int gy = ty0 + r - G::PAD, gx = tx0 + col - G::XOFF;
                v[u] = 0.f;
                if (idx < n_el && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    v[u] = d.x[(((size_t)b * Cin + ci_lo + c) * H + gy) * (size_t)W + gx];
                    if constexpr (IN != BNERV_IN_PLAIN) v[u] = xf_in<IN>(v[u], 1.0f + d.scale[b * Cin + ci_lo + c], d.shift[b * Cin + ci_lo + c]);
                }
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
    };

    constexpr int NGS = MTW * 16 * TH * (TW / 4) / 256;           // g slots per thread (pair path uses half of them)
    constexpr int NXS = (NPL * G::ROWS * G::SEGS + 255) / 256;
    f32x4 ga[PIPE ? NGS : 1], gb[PIPE && GTWO ? NGS : 1], xa[PIPE ? NXS : 1];

    auto issue = [&](int b, int ty0, int tx0) {                  // PIPE only
        if (vec_g) {
#pragma unroll
            for (int k = 0; k < NGS; ++k) g_load(tid + k * 256, b, ty0, tx0, ga[PIPE ? k : 0], gb[PIPE ? k : 0]);
        }
        if (vec_x) {
#pragma unroll
            for (int k = 0; k < NXS; ++k) x_load(tid + k * 256, b, ty0, tx0, xa[PIPE ? k : 0]);
        }
    };
    auto commit = [&](int b, int ty0, int tx0) {                 // PIPE only
        if (vec_g) {
#pragma unroll
            for (int k = 0; k < NGS; ++k) g_store(tid + k * 256, ga[PIPE ? k : 0], gb[PIPE ? k : 0]);
        } else g_stage_scalar(b, ty0, tx0);
        if (vec_x) {
#pragma unroll
            for (int k = 0; k < NXS; ++k) x_store(tid + k * 256, b, ty0, tx0, xa[PIPE ? k : 0]);
        } else x_stage_scalar(b, ty0, tx0);
    };
    auto stage_direct = [&](int b, int ty0, int tx0) {           // !PIPE: load + store slot by slot (registers reused)
        if (vec_g) {                                           // batches of 4 slots: loads first, then the LDS stores
            for (int s0 = tid; s0 < ng_slots; s0 += 256 * 4) {
                f32x4 a[4], bq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g_load(s0 + u * 256, b, ty0, tx0, a[u], bq[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) g_store(s0 + u * 256, a[u], bq[u]);
            }
        } else g_stage_scalar(b, ty0, tx0);
        if (vec_x) {
            for (int s0 = tid; s0 < nx_slots; s0 += 256 * 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x_load(s0 + u * 256, b, ty0, tx0, v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) x_store(s0 + u * 256, b, ty0, tx0, v[u]);
            }
        } else x_stage_scalar(b, ty0, tx0);
    };

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tiles = wa.tiles_x * wa.tiles_y;
    const int total = d.B * tiles;
    int t = blockIdx.x;
    if (PIPE && t < total) {
        const int b = t / tiles, tile = t - b * tiles;
        issue(b, (tile / wa.tiles_x) * TH, (tile % wa.tiles_x) * TW);
        commit(b, (tile / wa.tiles_x) * TH, (tile % wa.tiles_x) * TW);
    }
    for (; t < total; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool has_next = tn < total;
        const int bn = has_next ? tn / tiles : 0, tilen = has_next ? tn - bn * tiles : 0;
        const int ty0n = (tilen / wa.tiles_x) * TH, tx0n = (tilen % wa.tiles_x) * TW;
        if (!PIPE) {
            const int b = t / tiles, tile = t - b * tiles;
            lds_barrier();                                 // previous tile fully consumed
            stage_direct(b, (tile / wa.tiles_x) * TH, (tile % wa.tiles_x) * TW);
        }
        lds_barrier();                                     // tile t staged by everyone
        if (PIPE && has_next) issue(bn, ty0n, tx0n);       // flies under the MFMA phase
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            float af[MTW], bf[NTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) af[m] = s_g[abase[m] + st * 4];
#pragma unroll
            for (int n = 0; n < NTW; ++n) bf[n] = s_in[bbase[n] + ((st >> 3) * G::RS + (st & 7) * 4)];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
        if (PIPE) {
            lds_barrier();                                 // everyone done reading tile t
            if (has_next) commit(bn, ty0n, tx0n);
        }
    }

    // cross-wave reduction through LDS, then ONE slab per block: slab[block][co][n]; D layout: lane holds rows (cout)
    // 4*kq..4*kq+3 of column (n) li.  The waves add their accumulators ONE AFTER THE OTHER into a single [MTW*16][NTW*16] area
    // (fixed order 0,1,2,3 => deterministic): four separate areas would need more LDS than the staging buffers (86 KB for the
    // 3x7 shape) and cost the second resident block per CU.
    __syncthreads();
    float* s_red = smem;
    constexpr int RW = NTW * 16, RSZ = MTW * 16 * RW;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q = s_red + (m * 16 + 4 * kq + r) * RW + n * 16 + li;
                        *q = (w == 0) ? acc[m][n][r] : *q + acc[m][n][r];
                    }
        }
        __syncthreads();
    }
    float* slab = wa.slab + (size_t)blockIdx.x * Cout * wa.ncols;
    for (int idx = tid; idx < RSZ; idx += 256) {
        const int row = idx / RW, colq = idx - row * RW;
        const int co = co_base + row, col = n_base + colq;
        if (co < Cout && col < wa.ncols) slab[(size_t)co * wa.ncols + col] = s_red[idx];
    }
}

// finish: dw[co][n] = sum_slabs, db[co] = column nW.  32 consecutive elements x 8 slab lanes per block: coalesced 128-B rows,
// 8-way parallel over slabs, fixed combination order (deterministic).
__global__ __launch_bounds__(1024) void wgrad_finish_kernel(const float* __restrict__ slab, int n_slabs, int Cout, int ncols, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[32][33];
    const int e = threadIdx.x & 31, lane = threadIdx.x >> 5;        // 32 elements x 32 slab lanes: latency-bound, so go wide
    const int i = blockIdx.x * 32 + e;
    const int count = Cout * ncols;
    float s0 = 0.f, s1 = 0.f;
    if (i < count) {
        int k = lane;
        for (; k + 32 < n_slabs; k += 64) { s0 += slab[(size_t)k * count + i]; s1 += slab[(size_t)(k + 32) * count + i]; }
        if (k < n_slabs) s0 += slab[(size_t)k * count + i];
    }
    red[lane][e] = s0 + s1;
    __syncthreads();
    if (lane == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][e];
        const int co = i / ncols, n = i - co * ncols;
        if (n < ncols - 1) dw[(size_t)co * (ncols - 1) + n] = t;
        else if (db) db[co] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------- lean kernel
// Same GEMM, for the shapes with ONE cout tile (every 12-channel layer and the 1x1 head), rebuilt around the measured cost
// model of gfx950 (see conv.hip, "lean kernel"): VALU/SALU work is not hidden by the MFMAs of the other waves, so staging is
// raw buffer loads with per-block slot offsets + a per-tile SGPR base, out-of-image slots are switched off through the
// offset (no branches), interior tiles take a path without any per-slot arithmetic, tiles are walked incrementally, and each
// XCD works on a contiguous slice of tiles (halo rows meet in one L2).  LDS is trimmed so that 4 blocks fit a CU.

#ifdef BNERV_TRACE
#define WTRACE(it_, slot) do { if (lane == 0 && blockIdx.x < 1024 && (it_) < 8) g_trace_w[((blockIdx.x * 4 + wave) * 8 + (it_)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WTRACE(it_, slot) do {} while (0)
#endif
// GM2: 0 = g as is, 1 = g is the pixel-shuffled (x2) gradient (two float4 per channel PAIR), 2 = tanh-grad (g, gaux)
// (A v_mfma_f32_4x4x1 form of this body -- no 12 -> 16 row padding -- was built and measured slower in round 4: DESIGN section 11.1;
// retired in round 5, git history has it.)
template <int KS, int IN, int GM2>
__device__ __forceinline__ void wgrad_lean_body(const WArgs& wa, const int n_grows /* s_g rows kept */, const SidePack& side, const int vb, const int vgrid) {
    // (vb of vgrid: this launch's block index / size, or the weight-gradient part of a paired launch)
    using G = Geo<KS>;
    constexpr int NTW = (KS == 3) ? 7 : 1;
    constexpr int NPLL = (KS == 3) ? 12 : 15;                                // data planes (input channels) of this kernel
    constexpr int NXSLOT = NPLL * G::ROWS * G::SEGS;
    constexpr int NXS = (NXSLOT + 255) / 256;
    constexpr int NGS = (GM2 == 1) ? 2 : 4;                                  // g slots per thread (16 couts, or 8 cout pairs)
    constexpr bool GTWO = (GM2 != 0);
    constexpr bool AFF = (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE);
    constexpr bool B128 = (G::PLANE % 4 == 0);                               // plane stride 16-B aligned -> ds_write_b128
    const bnerv_wgrad_desc& d = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                                                       // n_grows rows of CSG.  Rows beyond Cout are never written: the
                                                                             // A reads of rows 12..15 then see s_in data, which only
                                                                             // reaches output rows that are dropped.
    float* s_in = smem + n_grows * CSG;                                      // (NPLL + 2) planes + dump area for idle slots (16-B aligned)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int nW = Cin * G::T;
    const int tiles_x = wa.tiles_x, tiles_y = wa.tiles_y;
    WTRACE(7, 0);

    // XCD x owns a contiguous slice of the tile list; its blocks take it round-robin
    const int total = d.B * tiles_x * tiles_y;
    const int nx = min(8, vgrid);                                            // (a grid of fewer than 8 blocks has fewer slices)
    const int xcd = vb % nx, lb = vb / nx;
    const int nlb = (vgrid - xcd + nx - 1) / nx;
    const int per = total / nx, extra = total % nx;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + lb;
    const bool has_work = itx < r1;                                          // (a block without tiles still writes its zero slab)
    const int step_q = nlb / tiles_x, step_r = nlb - step_q * tiles_x;
    LTile it{0, 0, 0};
    if (has_work) {
        const int tiles = tiles_x * tiles_y;
        it.b = itx / tiles;
        const int t = itx - it.b * tiles;
        it.ty = t / tiles_x;
        it.tx = t - it.ty * tiles_x;
    }
    auto advance = [&](LTile a) {
        a.tx += step_r;
        a.ty += step_q;
        if (a.tx >= tiles_x) { a.tx -= tiles_x; ++a.ty; }
        while (a.ty >= tiles_y) { a.ty -= tiles_y; ++a.b; }
        return a;
    };

    // constant planes: ones (bias column) and zeros (columns beyond the weight matrix)
    for (int i = tid; i < 2 * G::PLANE; i += 256) s_in[NPLL * G::PLANE + i] = i < G::PLANE ? 1.0f : 0.0f;

    // ---- per-slot constants
    auto xslot = [&](int k, int& c, int& r, int& sg) {
        const int sidx = tid + k * 256;
        c = sidx / (G::ROWS * G::SEGS);
        const int rem = sidx - c * (G::ROWS * G::SEGS);
        r = rem / G::SEGS;
        sg = rem - r * G::SEGS;
    };
    auto x_inside = [&](int k, int ty0, int tx0) {
        int c, r, sg;
        xslot(k, c, r, sg);
        const int gy = ty0 + r - G::PAD, gx = tx0 + 4 * sg - G::XOFF;
        return (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    };
    unsigned voffx[NXS], voffg[NGS];
    int loffx[NXS], loffg[NGS];
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
        int c, r, sg;
        xslot(k, c, r, sg);
        const bool real = tid + k * 256 < NXSLOT;
        voffx[k] = (real && c < Cin) ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;
        loffx[k] = real ? (c * G::PLANE + r * G::RS + 4 * sg) * 4 : ((NPLL + 2) * G::PLANE + (tid + k * 256 - NXSLOT) * 4) * 4;
    }
#pragma unroll
    for (int k = 0; k < NGS; ++k) {
        const int sidx = tid + k * 256;
        const int c = sidx >> 6, r = (sidx >> 3) & 7, sg = sidx & 7;
        if constexpr (GM2 == 1) {                                            // c = cout pair: couts 2c (j = 0) and 2c + 1 (j = 1)
            const int co = 2 * c, cf = co >> 2, i = (co >> 1) & 1;
            voffg[k] = co < Cout ? (unsigned)((((cf * 2 * H) + 2 * r + i) * (2 * W) + 8 * sg) * 4) : OOB;
            loffg[k] = (2 * c * CSG + r * TW + 4 * sg) * 4;
        } else {
            voffg[k] = c < Cout ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;
            loffg[k] = (c * CSG + r * TW + 4 * sg) * 4;
        }
    }
    const unsigned shift = (unsigned)((G::PAD * W + G::XOFF) * 4);
    const unsigned x_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned g_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, x_bytes);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(d.g, 0, g_bytes);
    const __amdgpu_buffer_rsrc_t rg2 = make_rsrc(GM2 == 2 ? d.gaux : d.g, 0, g_bytes);

    // affine parameters through LDS: one 32-lane load per block, then every thread picks its slots' values (see conv.hip)
    float sc[NXS], sh[NXS];
    float* s_aff = s_in + (NPLL + 2) * G::PLANE + (NXS * 256 - NXSLOT) * 4;   // [2][16], behind the dump area
    auto fetch_affine = [&](int b) {
        const int c = tid & 15;
        float v = 0.f;
        if (tid < 32 && c < Cin) v = tid < 16 ? 1.0f + d.scale[b * Cin + c] : d.shift[b * Cin + c];
        return v;
    };
    auto load_affine = [&](int b) {
        const float v = fetch_affine(b);
        lds_barrier();
        if (tid < 32) s_aff[tid] = v;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            int c, r, sg;
            xslot(k, c, r, sg);
            const bool ok = voffx[k] != OOB;
            sc[k] = ok ? s_aff[c & 15] : 0.f;
            sh[k] = ok ? s_aff[16 + (c & 15)] : 0.f;
        }
    };

    f32x4 xa[NXS], ga[NGS], gb[GTWO ? NGS : 1];
    // The prefetch of the next tile is issued ONE LOAD AT A TIME, spread over the K loop of the current tile: after a barrier
    // every wave of the CU would otherwise fire its 8-9 loads at once and sit in the vector-memory issue queue for 2-4 k cycles
    // per tile before its first MFMA (tools/ktrace_w.py).  part p < NGS: g slot p; otherwise x slot p - NGS.
    constexpr int NPART = NGS + NXS;
    struct Pre { unsigned sbx, sbg; int ty0, tx0; bool interior, gfull; };
    auto prep = [&](const LTile& a) {
        Pre q;
        q.ty0 = a.ty * TH; q.tx0 = a.tx * TW;
        q.sbx = (unsigned)((((a.b * Cin) * H + q.ty0) * W + q.tx0) * 4);
        q.sbg = GM2 == 1 ? (unsigned)(((((a.b * (Cout >> 2)) * 2 * H) + 2 * q.ty0) * (2 * W) + 2 * q.tx0) * 4)
                         : (unsigned)((((a.b * Cout) * H + q.ty0) * W + q.tx0) * 4);
        q.interior = q.ty0 >= G::PAD && q.ty0 + TH + G::PAD <= H && q.tx0 >= G::XOFF && q.tx0 + TW + G::XOFF <= W;
        q.gfull = q.ty0 + TH <= H && q.tx0 + TW <= W;
        return q;
    };
    auto issue_part = [&](const Pre& q, int p) {           // p is a compile-time constant at every call site
        if (p < NGS) {
            const int k = p;
            unsigned vo = voffg[k];
            if (!q.gfull) {
                const int sidx = tid + k * 256;
                const int r = (sidx >> 3) & 7, sg = sidx & 7;
                vo = (q.ty0 + r < H && q.tx0 + 4 * sg < W) ? vo : OOB;
            }
            ga[k] = bload(rg, vo, q.sbg);
            if constexpr (GM2 == 1) gb[k] = bload(rg, vo == OOB ? OOB : vo + 16u, q.sbg);
            if constexpr (GM2 == 2) gb[k] = bload(rg2, vo, q.sbg);
        } else {
            const int k = p - NGS;
            unsigned vo = voffx[k];
            if (!q.interior) vo = x_inside(k, q.ty0, q.tx0) ? vo : OOB;
            xa[k] = bload(rx, vo, q.sbx);
        }
    };
    auto issue = [&](const LTile& a) {                     // all parts at once (first tile of a block)
        const Pre q = prep(a);
#pragma unroll
        for (int p = 0; p < NPART; ++p) issue_part(q, p);
    };
    auto commit = [&](const LTile& a) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const bool interior = ty0 >= G::PAD && ty0 + TH + G::PAD <= H && tx0 >= G::XOFF && tx0 + TW + G::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NGS; ++k) {
            char* dst = reinterpret_cast<char*>(s_g) + loffg[k];
            if constexpr (GM2 == 1) {
                const f32x4 a0 = ga[k], a1 = gb[k];
                if (voffg[k] != OOB) {                                       // rows beyond Cout are not kept in LDS
                    *reinterpret_cast<float2*>(dst) = float2{a0[0], a0[2]};
                    *reinterpret_cast<float2*>(dst + 8) = float2{a1[0], a1[2]};
                    *reinterpret_cast<float2*>(dst + CSG * 4) = float2{a0[1], a0[3]};
                    *reinterpret_cast<float2*>(dst + CSG * 4 + 8) = float2{a1[1], a1[3]};
                }
            } else {
                f32x4 v = ga[k];
                if constexpr (GM2 == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = 2.0f * gb[k][e] - 1.0f; v[e] = v[e] * 0.5f * (1.0f - t * t); }
                }
                if (voffg[k] != OOB) {                                       // rows beyond Cout are not kept in LDS
                    *reinterpret_cast<float2*>(dst) = float2{v[0], v[1]};
                    *reinterpret_cast<float2*>(dst + 8) = float2{v[2], v[3]};
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            f32x4 v = xa[k];
            if constexpr (IN != BNERV_IN_PLAIN) {
                float s = AFF ? sc[k] : 0.f, h = AFF ? sh[k] : 0.f;
                if (!interior) {                                             // zero padding is applied AFTER the prologue
                    const bool ok = x_inside(k, ty0, tx0);
                    s = ok ? s : 0.f;
                    h = ok ? h : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xf_in<IN>(v[e], s, h);
            }
            char* dst = reinterpret_cast<char*>(s_in) + loffx[k];
            if constexpr (B128) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                *reinterpret_cast<float2*>(dst) = float2{v[0], v[1]};
                *reinterpret_cast<float2*>(dst + 8) = float2{v[2], v[3]};
            }
        }
    };

    // per-lane fragment bases.  pixel of (wave, step, kq): p = wave*64 + step*4 + kq
    int bbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = nt * 16 + li;
        int off;
        if (n < nW) {
            const int ci = n / G::T, tap = n - ci * G::T;
            off = ci * G::PLANE + (tap / KS) * G::RS + (tap % KS) + G::COL0;
        } else {
            off = (n == nW ? NPLL : NPLL + 1) * G::PLANE;
        }
        bbase[nt] = off + (2 * wave) * G::RS + kq;
    }
    const int abase = li * CSG + wave * 64 + kq;

    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int aff_b = -1;
    if (has_work) {
        issue(it);
        if constexpr (AFF) { load_affine(it.b); aff_b = it.b; }
        commit(it);
    }
    WTRACE(7, 1);
    int wt = 0; (void)wt;
    for (; itx < r1; itx += nlb, ++wt) {
        WTRACE(wt, 0);
        const bool has_next = itx + nlb < r1;
        LTile nxt = it;
        if (has_next) nxt = advance(it);
        lds_barrier();                                     // tile t staged by everyone
        WTRACE(wt, 1);
        const Pre pre = prep(nxt);
        WTRACE(wt, 2);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if (has_next) {                                // the next tile's loads, spread over the K steps
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    if (p * 16 / NPART == st) issue_part(pre, p);
            }
            float bf[NTW];
            const float af = s_g[abase + st * 4];
#pragma unroll
            for (int n = 0; n < NTW; ++n) bf[n] = s_in[bbase[n] + ((st >> 3) * G::RS + (st & 7) * 4)];
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], acc[n], 0, 0, 0);
        }
        WTRACE(wt, 3);
        lds_barrier();                                     // everyone done reading tile t
        WTRACE(wt, 5);
        if (has_next) {
            if constexpr (AFF) { if (nxt.b != aff_b) { load_affine(nxt.b); aff_b = nxt.b; } }
            commit(nxt);
        }
        WTRACE(wt, 6);
        it = nxt;
    }
    WTRACE(7, 2);

    // cross-wave reduction through LDS (fixed order => deterministic), then ONE slab per block
    __syncthreads();
    float* s_red = smem;
    constexpr int RW = NTW * 16, RSZ = 16 * RW;
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[wave * RSZ + (4 * kq + r) * RW + n * 16 + li] = acc[n][r];
    __syncthreads();
    float* slab = wa.slab + (size_t)vb * Cout * wa.ncols;
    for (int idx = tid; idx < RSZ; idx += 256) {
        const int row = idx / RW, col = idx - row * RW;
        if (row < Cout && col < wa.ncols)
            slab[(size_t)row * wa.ncols + col] = (s_red[idx] + s_red[RSZ + idx]) + (s_red[2 * RSZ + idx] + s_red[3 * RSZ + idx]);
    }
    WTRACE(7, 3);
    side_run_hosted(side, smem, vb, vgrid);                // queued slab reductions of EARLIER launches (sidejob.h)
    WTRACE(7, 4);
}
template <int KS, int IN, int GM2>
__global__ __launch_bounds__(256, (KS == 3 ? 3 : 4)) void wgrad_lean_kernel(const WArgs wa, const int n_grows, const SidePack side) {
    wgrad_lean_body<KS, IN, GM2>(wa, n_grows, side, (int)blockIdx.x, (int)gridDim.x);
}
// ---- paired launch: the data gradient of a 12-channel 3x3 conv (conv4_body.h) and a weight gradient that does not depend on it, in
// ONE grid -- blocks [0, n_conv) run the conv, the rest the weight gradient (small layers), or the two roles interleaved with equal
// block counts and the same tile walk (large layers: n_conv < 0, see launch_pair).  Inside a TAT block's backward the pairs are
// (dW1 | dconv1), (dW0 | dconv0), (dW_block | dconv_block): each pair reads the same incoming gradient, and neither half fills the
// chip through its prologue and tail (at 180x320 each is one tile per block: two latency-bound launches become one).
template <int EP, int WIN>
__global__ __launch_bounds__(256, 3) void conv_wgrad_pair_kernel(const bnerv_conv::KArgs ka, const WArgs wa, const int n_grows, const int n_conv, const SidePack side, const int pat) {
    if (n_conv < 0) {
        // interleaved roles (n_conv = -(blocks per role)): XCD-local slots alternate conv / weight gradient, and block k of either role
        // walks the same tile list on the same XCD -- what one reads of the shared gradient (and of the conv's aux = the weight
        // gradient's input) the other finds in that XCD's L2
        const int nr = -n_conv, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, vb = ((slot >> 1) << 3) + xcd;
        const int role = pat == 0 ? (slot & 1) : ((slot + (slot >> pat)) & 1);
        if (role == 0) {
            SidePack none;
            none.n_jobs = 0; none.n_slices = 0;
            bnerv_q4::conv_q4_body<BNERV_IN_PLAIN, EP>(ka, none, vb, nr);
        } else {
            wgrad_lean_body<3, WIN, 0>(wa, n_grows, side, vb, nr);
        }
        return;
    }
    if ((int)blockIdx.x < n_conv) {                        // block-uniform
        SidePack none;
        none.n_jobs = 0; none.n_slices = 0;
        bnerv_q4::conv_q4_body<BNERV_IN_PLAIN, EP>(ka, none, (int)blockIdx.x, n_conv);
    } else {
        wgrad_lean_body<3, WIN, 0>(wa, n_grows, side, (int)blockIdx.x - n_conv, (int)gridDim.x - n_conv);
    }
}

#include "pairf_body.h"      // the shared-tile form of the same pair (round 5)
template <int EP, int WIN>
__global__ __launch_bounds__(256, 2) void pair_fused_kernel(const bnerv_conv::KArgs ka, const WArgs wa, const SidePack side) {
    pair_fused_body<EP, WIN>(ka, wa, side, (int)blockIdx.x, (int)gridDim.x);
}
template <int EP>
__global__ __launch_bounds__(256, 2) void pair_fold_kernel(const bnerv_conv::KArgs ka, const WArgs wa, const SidePack side) {
    pair_fold_body<EP>(ka, wa, side, (int)blockIdx.x, (int)gridDim.x);
}

constexpr size_t WLEAN_MAX_BYTES = 0x7ff00000;

static bool wlean_ok(const WArgs& wa) {
    const bnerv_wgrad_desc& d = wa.d;
    if (!wa.vec || d.Cout > 16) return false;
    if (d.k == 3 && d.Cin > 12) return false;
    if (d.k == 1 && d.Cin > 15) return false;
    if (d.g_mode == BNERV_IN_UNSHUFFLE && d.g_s > 2) return false;
    if (d.g_mode == BNERV_IN_UNSHUFFLE && d.g_s == 2 && (d.Cout % 4 != 0)) return false;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    return (size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 < WLEAN_MAX_BYTES;
}

static int wlean_blocks(const bnerv_wgrad_desc& d) {
    const int total_tiles = d.B * cdiv(d.H, TH) * cdiv(d.W, TW);
    const int want = total_tiles < 1 ? 1 : total_tiles;   // fill the machine first: small layers are latency-bound
    const int target = 256 * (d.k == 3 ? 3 : 4);
    return want < target ? want : target;
}

template <int KS>
static size_t wlean_lds_bytes(int n_grows) {
    using G = Geo<KS>;
    constexpr int NTW = (KS == 3) ? 7 : 1;
    constexpr int NPLL = (KS == 3) ? 12 : 15;
    constexpr int NXSLOT = NPLL * G::ROWS * G::SEGS;
    constexpr int NXS = (NXSLOT + 255) / 256;
    const size_t lds_main = (size_t)(NPLL + 2) * G::PLANE + (size_t)(NXS * 256 - NXSLOT) * 4 + (size_t)n_grows * CSG + 64 + 32;
    const size_t lds_red = (size_t)4 * 16 * NTW * 16;
    return (lds_main > lds_red ? lds_main : lds_red) * sizeof(float);
}

template <int KS, int IN, int GM2>
int launch_wlean(hipStream_t st, const WArgs& wa) {
    const int n_grows = wa.d.Cout <= 12 ? 12 : 16;
    const size_t lds = wlean_lds_bytes<KS>(n_grows);
    SidePack side;
    bnerv_side_take(wa.d.ctx, &side, 2 * wlean_blocks(wa.d));
    hipLaunchKernelGGL((wgrad_lean_kernel<KS, IN, GM2>), dim3(wlean_blocks(wa.d)), dim3(256), lds, st, wa, n_grows, side);
    BNERV_LAUNCH_CHECK("wgrad_lean");
    return BNERV_OK;
}

template <int KS>
int launch_wlean_modes(hipStream_t st, const WArgs& wa) {
    const int in = wa.d.in_mode, gm = wa.d.g_mode;
    if (gm == BNERV_IN_TANHGRAD && in == BNERV_IN_PLAIN) return launch_wlean<KS, BNERV_IN_PLAIN, 2>(st, wa);
    if (gm == BNERV_IN_PLAIN || gm == BNERV_IN_UNSHUFFLE) {
        const bool pair = gm == BNERV_IN_UNSHUFFLE && wa.d.g_s == 2;
        if (in == BNERV_IN_PLAIN) return pair ? launch_wlean<KS, BNERV_IN_PLAIN, 1>(st, wa) : launch_wlean<KS, BNERV_IN_PLAIN, 0>(st, wa);
        if constexpr (KS == 3) {
            if (!pair && in == BNERV_IN_AFFINE) return launch_wlean<KS, BNERV_IN_AFFINE, 0>(st, wa);
            if (!pair && in == BNERV_IN_GELU_AFFINE) return launch_wlean<KS, BNERV_IN_GELU_AFFINE, 0>(st, wa);
        }
    }
    return -1;                                            // not covered: the caller falls back to the general kernel
}

// ---------------------------------------------------------------------------------------------------------------- wide kernel
// The lean recipe for the stride-1 3x3 layers with more than 16 output channels or more than 12 input channels (TAT convs,
// stride-1 up-convs and heads of the 3M models: 22..64 channels).  One block = MTW*16 couts x NTW*16 (ci, tap) columns
// (column group `grp`); the 4 waves split the 64 K-steps of a tile; the next tile's buffer loads are issued one at a time under
// the MFMA phase of the current one.  The column groups of one tile slot sit next to each other in the SAME XCD, so the gradient
// tile they all re-read comes from that XCD's L2.  Shapes (MTW, NTW) are picked per layer to minimise the padded columns
// (38 ch: 3x6 -> 4 groups of 96 for 343 columns; 46 ch: 3x7 -> 4 groups of 112 for 415), within the 256-VGPR budget.
// GM2: 0 = g as is, 1 = g is the pixel-shuffled (x2) gradient (two float4 per cout PAIR), 2 = tanh-grad (g, gaux),
//      3 = pixel-shuffled by g_s (3, 5): one dword per pixel, g_s apart
template <int IN, int GM2, int MTW, int NTW>
__device__ __forceinline__ void wgrad_wide_body(const WArgs& wa, const int slots, const SidePack& side, const int vb, const int vgrid) {
    // (vb of vgrid: this launch's block index / size, or the weight-gradient part of a paired launch)
    using G = Geo<3>;
    constexpr int NPL = wgrad_npl<3, NTW>();
    constexpr int NXSLOT = NPL * G::ROWS * G::SEGS;
    constexpr int NXS = (NXSLOT + 255) / 256;
    constexpr int NGS = (GM2 == 1) ? MTW * 2 : MTW * 4;                      // g slot k of wave w: cout row (pair) w + 4 k (wave-uniform)
    constexpr bool GTWO = (GM2 == 1 || GM2 == 2);
    constexpr bool AFF = (IN == BNERV_IN_AFFINE || IN == BNERV_IN_GELU_AFFINE);
    constexpr int AFFN = 32;                                                 // >= NPL
    static_assert(NPL <= AFFN, "affine table too small");
    const bnerv_wgrad_desc& d = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int co_base = (int)(((vb >> 3) % (unsigned)(wa.n_ngroups * wa.n_mgroups)) / (unsigned)wa.n_ngroups) * MTW * 16;
    const int g_rows = min(MTW * 16, Cout - co_base);
    float* s_g = smem;                                                       // g_rows rows of CSG (rows beyond Cout: see conv_wgrad_kernel)
    float* s_in = smem + g_rows * CSG;                                       // (NPL + 2) planes
    float* s_aff = s_in + (NPL + 2) * G::PLANE;                              // [2][AFFN]
    const int nW = Cin * G::T;
    const int tiles_x = wa.tiles_x, tiles_y = wa.tiles_y;

    // block -> (xcd, slot, column group); the XCD owns a contiguous slice of the tile list, its slots take it round-robin
    const int ngroups = wa.n_ngroups * wa.n_mgroups;
    const int xcd = vb & 7, q = vb >> 3;
    const int slot = q / ngroups, grp = q - slot * ngroups;
    const int total = d.B * tiles_x * tiles_y;
    const int per = total >> 3, extra = total & 7;
    const int r0 = xcd * per + min(xcd, extra), r1 = r0 + per + (xcd < extra ? 1 : 0);
    int itx = r0 + slot;
    const bool has_work = itx < r1;                                          // (a block without tiles still writes its zero slab part)
    const int step_q = slots / tiles_x, step_r = slots - step_q * tiles_x;
    LTile it{0, 0, 0};
    if (has_work) {
        const int tiles = tiles_x * tiles_y;
        it.b = itx / tiles;
        const int t = itx - it.b * tiles;
        it.ty = t / tiles_x;
        it.tx = t - it.ty * tiles_x;
    }
    auto advance = [&](LTile a) {
        a.tx += step_r;
        a.ty += step_q;
        if (a.tx >= tiles_x) { a.tx -= tiles_x; ++a.ty; }
        while (a.ty >= tiles_y) { a.ty -= tiles_y; ++a.b; }
        return a;
    };
    const int mg = grp / wa.n_ngroups;
    const int n_base = (grp - mg * wa.n_ngroups) * NTW * 16;
    const int ci_lo = min(n_base, nW - 1) / G::T;
    const int ci_hi = min(n_base + NTW * 16 - 1, nW - 1) / G::T;
    const int npl = ci_hi - ci_lo + 1;                                       // <= NPL

    for (int i = tid; i < 2 * G::PLANE; i += 256) s_in[NPL * G::PLANE + i] = i < G::PLANE ? 1.0f : 0.0f;

    auto xslot = [&](int k, int& c, int& r, int& sg) {
        const int sidx = tid + k * 256;
        c = sidx / (G::ROWS * G::SEGS);
        const int rem = sidx - c * (G::ROWS * G::SEGS);
        r = rem / G::SEGS;
        sg = rem - r * G::SEGS;
    };
    auto x_inside = [&](int k, int ty0, int tx0) {
        int c, r, sg;
        xslot(k, c, r, sg);
        const int gy = ty0 + r - G::PAD, gx = tx0 + 4 * sg - G::XOFF;
        return (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    };
    unsigned voffx[NXS];
    int loffx[NXS];
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
        int c, r, sg;
        xslot(k, c, r, sg);
        const bool real = tid + k * 256 < NXSLOT && c < npl;
        voffx[k] = real ? (unsigned)(((c * H + r) * W + 4 * sg) * 4) : OOB;
        loffx[k] = (c * G::PLANE + r * G::RS + 4 * sg) * 4;
    }
    // g slot k of this thread: cout row wave + 4 k, tile row (tid >> 3) & 7, segment tid & 7
    const int g_r = (tid >> 3) & 7, g_sg = tid & 7;
    const int gs = d.g_s;                                                    // GM2 == 3: du[c][y*s + i][x*s + j], a stride-s gather along x
    const unsigned voffg0 = GM2 == 1 ? (unsigned)((2 * g_r * 2 * W + 8 * g_sg) * 4)
                          : GM2 == 3 ? (unsigned)((g_r * gs * W * gs + 4 * g_sg * gs) * 4) : (unsigned)((g_r * W + 4 * g_sg) * 4);
    const int loffg0 = ((GM2 == 1 ? 2 * wave : wave) * CSG + g_r * TW + 4 * g_sg) * 4;
    const unsigned hw4 = (unsigned)(H * W * 4);
    const unsigned shift = (unsigned)((G::PAD * W + G::XOFF) * 4);
    const unsigned x_bytes = (unsigned)((size_t)d.B * Cin * H * W * 4) + shift;
    const unsigned g_bytes = (unsigned)((size_t)d.B * Cout * H * W * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, shift, x_bytes);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(d.g, 0, g_bytes);
    const __amdgpu_buffer_rsrc_t rg2 = make_rsrc(GM2 == 2 ? d.gaux : d.g, 0, g_bytes);

    auto load_affine = [&](int b) {                        // s_aff[c] = 1 + scale[b][ci_lo + c], s_aff[AFFN + c] = shift; 0 beyond npl
        float v = 0.f;
        if (tid < 2 * AFFN) {
            const int c = tid & (AFFN - 1);
            if (c < npl) v = tid < AFFN ? 1.0f + d.scale[b * Cin + ci_lo + c] : d.shift[b * Cin + ci_lo + c];
        }
        lds_barrier();
        if (tid < 2 * AFFN) s_aff[tid] = v;
        lds_barrier();
    };

    unsigned goff[GM2 == 3 ? NGS : 1];                                       // block constants: the (plane, i, j) part of each cout row's offset
    if constexpr (GM2 == 3) {
#pragma unroll
        for (int k = 0; k < NGS; ++k) {
            const int co = co_base + wave + 4 * k, s2 = gs * gs;
            const int cf = co / s2, rem = co - cf * s2, i = rem / gs, j = rem - i * gs;
            goff[k] = (unsigned)((((cf * gs * H) + i) * (gs * W) + j) * 4);
        }
    }
    f32x4 xa[NXS], ga[NGS], gb[GTWO ? NGS : 1];
    constexpr int NPART = NGS + NXS;
    struct Pre { unsigned sbx, sbg, vog; int ty0, tx0; bool interior; };
    auto prep = [&](const LTile& a) {
        Pre p;
        p.ty0 = a.ty * TH; p.tx0 = a.tx * TW;
        p.sbx = (unsigned)((((a.b * Cin + ci_lo) * H + p.ty0) * W + p.tx0) * 4);
        if constexpr (GM2 == 1) p.sbg = (unsigned)(((a.b * (Cout >> 2) * 2 * H + 2 * p.ty0) * 2 * W + 2 * p.tx0) * 4);   // + the pair's plane / row parity
        else if constexpr (GM2 == 3) p.sbg = (unsigned)(((a.b * (Cout / (gs * gs)) * gs * H + gs * p.ty0) * gs * W + gs * p.tx0) * 4);   // + plane / (i, j)
        else p.sbg = (unsigned)((((a.b * Cout + co_base + wave) * H + p.ty0) * W + p.tx0) * 4);
        p.interior = p.ty0 >= G::PAD && p.ty0 + TH + G::PAD <= H && p.tx0 >= G::XOFF && p.tx0 + TW + G::XOFF <= W;
        p.vog = (p.ty0 + g_r < H && p.tx0 + 4 * g_sg < W) ? voffg0 : OOB;
        return p;
    };
    auto issue_part = [&](const Pre& p, int part) {        // `part` is a compile-time constant at every call site
        if (part < NGS) {
            const int k = part;
            if constexpr (GM2 == 1) {
                if (2 * (wave + 4 * k) < g_rows) {         // wave-uniform: cout pair co, co + 1 = (cf, i, j = 0 / 1)
                    const int co = co_base + 2 * (wave + 4 * k), cf = co >> 2, i = (co >> 1) & 1;
                    const unsigned so = p.sbg + (unsigned)(((cf * 2 * H + i) * 2 * W) * 4);
                    ga[k] = bload(rg, p.vog, so);
                    gb[k] = bload(rg, p.vog == OOB ? OOB : p.vog + 16u, so);
                }
            } else if constexpr (GM2 == 3) {
                if (wave + 4 * k < g_rows) {               // wave-uniform: co = cf*s*s + i*s + j; 4 pixels = 4 dwords s apart
                    const unsigned so = p.sbg + goff[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ga[k][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (int)(p.vog == OOB ? OOB : p.vog + (unsigned)(e * gs * 4)), (int)so, 0));
                }
            } else if (wave + 4 * k < g_rows) {            // wave-uniform
                ga[k] = bload(rg, p.vog, p.sbg + (unsigned)(4 * k) * hw4);
                if constexpr (GM2 == 2) gb[k] = bload(rg2, p.vog, p.sbg + (unsigned)(4 * k) * hw4);
            }
        } else {
            const int k = part - NGS;
            unsigned vo = voffx[k];
            if (!p.interior) vo = x_inside(k, p.ty0, p.tx0) ? vo : OOB;
            xa[k] = bload(rx, vo, p.sbx);
        }
    };
    auto commit = [&](const LTile& a) {
        const int ty0 = a.ty * TH, tx0 = a.tx * TW;
        const bool interior = ty0 >= G::PAD && ty0 + TH + G::PAD <= H && tx0 >= G::XOFF && tx0 + TW + G::XOFF <= W;
#pragma unroll
        for (int k = 0; k < NGS; ++k) {
            if constexpr (GM2 == 1) {
                if (2 * (wave + 4 * k) < g_rows) {
                    const f32x4 a0 = ga[k], a1 = gb[k];
                    char* dst = reinterpret_cast<char*>(s_g) + loffg0 + k * (8 * CSG * 4);
                    *reinterpret_cast<float2*>(dst) = float2{a0[0], a0[2]};
                    *reinterpret_cast<float2*>(dst + 8) = float2{a1[0], a1[2]};
                    *reinterpret_cast<float2*>(dst + CSG * 4) = float2{a0[1], a0[3]};
                    *reinterpret_cast<float2*>(dst + CSG * 4 + 8) = float2{a1[1], a1[3]};
                }
            } else if (wave + 4 * k < g_rows) {
                f32x4 v = ga[k];
                if constexpr (GM2 == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = 2.0f * gb[k][e] - 1.0f; v[e] = v[e] * 0.5f * (1.0f - t * t); }
                }
                char* dst = reinterpret_cast<char*>(s_g) + loffg0 + k * (4 * CSG * 4);
                *reinterpret_cast<float2*>(dst) = float2{v[0], v[1]};
                *reinterpret_cast<float2*>(dst + 8) = float2{v[2], v[3]};
            }
        }
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            if (k == NXS - 1 && tid + k * 256 >= NXSLOT) continue;          // idle slots of the last round
            f32x4 v = xa[k];
            if constexpr (IN != BNERV_IN_PLAIN) {
                int c, r, sg;
                xslot(k, c, r, sg);
                float sc_ = s_aff[c & (AFFN - 1)], sh_ = s_aff[AFFN + (c & (AFFN - 1))];
                if (!interior) {                                             // zero padding is applied AFTER the prologue
                    const bool ok = x_inside(k, ty0, tx0);
                    sc_ = ok ? sc_ : 0.f;
                    sh_ = ok ? sh_ : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xf_in<IN>(v[e], sc_, sh_);
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(s_in) + loffx[k]) = v;
        }
    };

    int bbase[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = n_base + nt * 16 + li;
        int off;
        if (n < nW) {
            const int ci = n / G::T, tap = n - ci * G::T;
            off = (ci - ci_lo) * G::PLANE + (tap / 3) * G::RS + (tap % 3) + G::COL0;
        } else {
            off = (n == nW ? NPL : NPL + 1) * G::PLANE;
        }
        bbase[nt] = off + (2 * wave) * G::RS + kq;
    }
    const int abase = li * CSG + wave * 64 + kq;

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int aff_b = -1;
    if (has_work) {
        const Pre p0 = prep(it);
#pragma unroll
        for (int part = 0; part < NPART; ++part) issue_part(p0, part);
        if constexpr (AFF) { load_affine(it.b); aff_b = it.b; }
        commit(it);
    }
    for (; itx < r1; itx += slots) {
        const bool has_next = itx + slots < r1;
        LTile nxt = it;
        if (has_next) nxt = advance(it);
        lds_barrier();                                     // tile t staged by everyone
        const Pre pre = prep(nxt);
        // fragments of step st + 1 are read while the MFMAs of step st run; the scheduler fence per step keeps the compiler from
        // hoisting all 16 steps' LDS reads (which costs more registers than the kernel has)
        float af[2][MTW], bf[2][NTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) af[0][m] = s_g[abase + m * 16 * CSG];
#pragma unroll
        for (int n = 0; n < NTW; ++n) bf[0][n] = s_in[bbase[n]];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if (has_next) {                                // the next tile's loads, spread over the K steps
#pragma unroll
                for (int part = 0; part < NPART; ++part)
                    if (part * 16 / NPART == st) issue_part(pre, part);
            }
            if (st < 15) {
#pragma unroll
                for (int m = 0; m < MTW; ++m) af[(st + 1) & 1][m] = s_g[abase + m * 16 * CSG + (st + 1) * 4];
#pragma unroll
                for (int n = 0; n < NTW; ++n) bf[(st + 1) & 1][n] = s_in[bbase[n] + (((st + 1) >> 3) * G::RS + ((st + 1) & 7) * 4)];
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st & 1][m], bf[st & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                     // everyone done reading tile t
        if (has_next) {
            if constexpr (AFF) { if (nxt.b != aff_b) { load_affine(nxt.b); aff_b = nxt.b; } }
            commit(nxt);
        }
        it = nxt;
    }

    // cross-wave reduction, one wave after the other into one area (fixed order), then this block's columns of the slot's slab
    __syncthreads();
    float* s_red = smem;
    constexpr int RW = NTW * 16, RSZ = MTW * 16 * RW;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
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

struct WidePlan { int mtw, ntw, ngroups, mgroups, slots; };

static WidePlan wide_plan(const bnerv_wgrad_desc& d) {
    WidePlan p{0, 0, 0, 1, 0};
    const int ncols = d.Cin * 9 + 1, nt = cdiv(ncols, 16), mt = cdiv(d.Cout, 16);
    auto best = [&](int a, int b) { return cdiv(nt, a) * a < cdiv(nt, b) * b ? a : b; };   // fewer padded columns; ties -> b
    if (mt == 1) { p.mtw = 1; p.ntw = 8; }
    else if (mt == 2) { p.mtw = 2; p.ntw = best(7, 9); }
    else if (mt == 3) { p.mtw = 3; p.ntw = best(6, 7); }
    else if (mt == 4) { p.mtw = 4; p.ntw = 4; }
    else { p.mtw = 3; p.ntw = best(6, 7); p.mgroups = cdiv(mt, 3); }      // (64 full gradient rows would leave one block per CU)
    p.ngroups = cdiv(nt, p.ntw);
    const int per_cu = p.mtw * p.ntw <= 8 ? 3 : 2;
    const int total_tiles = d.B * cdiv(d.H, TH) * cdiv(d.W, TW);
    int s = (256 * per_cu) / (8 * p.ngroups * p.mgroups);  // slots per XCD with every block resident
    const int want = cdiv(total_tiles, 8);
    if (s > want) s = want;
    if (s < 1) s = 1;
    p.slots = s;
    return p;
}
template <int IN, int GM2, int MTW, int NTW>
__global__ __launch_bounds__(256, (MTW * NTW <= 8 ? 3 : 2)) void wgrad_wide_kernel(const WArgs wa, const int slots, const SidePack side) {
    wgrad_wide_body<IN, GM2, MTW, NTW>(wa, slots, side, (int)blockIdx.x, (int)gridDim.x);
}

static bool wide_ok(const WArgs& wa) {
    const bnerv_wgrad_desc& d = wa.d;
    static const bool off = getenv("BNERV_NO_WIDE") != nullptr;            // A/B switch for tools/kwide.py
    if (off || !wa.vec || d.k != 3) return false;
    if (d.g_s > 2 && (d.g_mode != BNERV_IN_UNSHUFFLE || d.in_mode != BNERV_IN_PLAIN || d.Cout % (d.g_s * d.g_s) != 0 || d.Cout <= 16)) return false;
    if (d.g_s == 2 && (d.g_mode != BNERV_IN_UNSHUFFLE || d.in_mode != BNERV_IN_PLAIN || d.Cout % 4 != 0 || d.Cout <= 16)) return false;
    if (d.in_mode != BNERV_IN_PLAIN && d.in_mode != BNERV_IN_AFFINE) return false;
    if (d.g_mode == BNERV_IN_TANHGRAD && (d.in_mode != BNERV_IN_PLAIN || d.Cout > 16)) return false;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    return (size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 4) * 4 < WLEAN_MAX_BYTES;
}

template <int IN, int GM2, int MTW, int NTW>
int launch_wide(hipStream_t st, const WArgs& wa, const WidePlan& p) {
    using G = Geo<3>;
    constexpr int NPL = wgrad_npl<3, NTW>();
    const int g_rows = wa.d.Cout < MTW * 16 ? wa.d.Cout : MTW * 16;
    size_t lds_fl = (size_t)g_rows * CSG + (size_t)(NPL + 2) * G::PLANE + 64;
    if (lds_fl < (size_t)MTW * 16 * CSG) lds_fl = (size_t)MTW * 16 * CSG;
    if (lds_fl < (size_t)MTW * 16 * NTW * 16) lds_fl = (size_t)MTW * 16 * NTW * 16;
    const size_t lds = lds_fl * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_kernel<IN, GM2, MTW, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    const int grid = 8 * p.slots * p.ngroups * p.mgroups;
    SidePack side;
    bnerv_side_take(wa.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((wgrad_wide_kernel<IN, GM2, MTW, NTW>), dim3(grid), dim3(256), lds, st, wa, p.slots, side);
    BNERV_LAUNCH_CHECK("wgrad_wide");
    return BNERV_OK;
}

template <int IN, int GM2>
int launch_wide_shape(hipStream_t st, const WArgs& wa, const WidePlan& p) {
    if constexpr (GM2 == 1 || GM2 == 3) {                  // up-convs feeding PixelShuffle(2 / 3 / 5): Cout = s^2 * channels, 32 and up here
        if (p.mtw == 2) return p.ntw == 9 ? launch_wide<IN, GM2, 2, 9>(st, wa, p) : launch_wide<IN, GM2, 2, 7>(st, wa, p);
        if (p.mtw == 3) return p.ntw == 7 ? launch_wide<IN, GM2, 3, 7>(st, wa, p) : launch_wide<IN, GM2, 3, 6>(st, wa, p);
        if (p.mtw == 4) return launch_wide<IN, GM2, 4, 4>(st, wa, p);
        return -1;
    }
    if (p.mtw == 1) return launch_wide<IN, GM2, 1, 8>(st, wa, p);
    if constexpr (GM2 == 0) {
        if (p.mtw == 2) return p.ntw == 9 ? launch_wide<IN, GM2, 2, 9>(st, wa, p) : launch_wide<IN, GM2, 2, 7>(st, wa, p);
        if (p.mtw == 3) return p.ntw == 7 ? launch_wide<IN, GM2, 3, 7>(st, wa, p) : launch_wide<IN, GM2, 3, 6>(st, wa, p);
        if (p.mtw == 4) return launch_wide<IN, GM2, 4, 4>(st, wa, p);
    }
    return -1;
}

static int launch_wide_modes(hipStream_t st, const WArgs& wa, const WidePlan& p) {
    if (wa.d.g_mode == BNERV_IN_TANHGRAD) return launch_wide_shape<BNERV_IN_PLAIN, 2>(st, wa, p);
    if (wa.d.g_s == 2) return launch_wide_shape<BNERV_IN_PLAIN, 1>(st, wa, p);
    if (wa.d.g_s > 2) return launch_wide_shape<BNERV_IN_PLAIN, 3>(st, wa, p);
    if (wa.d.in_mode == BNERV_IN_PLAIN) return launch_wide_shape<BNERV_IN_PLAIN, 0>(st, wa, p);
    return launch_wide_shape<BNERV_IN_AFFINE, 0>(st, wa, p);
}

template <int IN, int SP, int MTW, int GM2>
__global__ __launch_bounds__(256, 2) void wgrad_bfw_kernel(const WArgs wa, const int slots, const int ngroups_n, const int ngroups_m, const SidePack side) {
    wgrad_bfw_body<IN, SP, MTW, GM2>(wa, slots, ngroups_n, ngroups_m, side, (int)blockIdx.x, (int)gridDim.x);
}

static bool bw_ok(const WArgs& wa) {
    const bnerv_wgrad_desc& d = wa.d;
    if (bw_mode() < 0 || !wa.vec || d.k != 3 || (d.g_s > 3 && d.g_s != 5) || d.g_mode == BNERV_IN_TANHGRAD) return false;
    if (d.in_mode != BNERV_IN_PLAIN && d.in_mode != BNERV_IN_AFFINE) return false;
    if (d.g_s >= 2 && d.in_mode != BNERV_IN_PLAIN) return false;
    if (d.Cout <= 16) return false;                        // (one cout tile: the f32 kernels are as fast or faster -- 64 -> 16 @540x960: 108 vs 115 us)
    int min_tiles = 16;
    if (const char* e = getenv("BNERV_SPLIT_WIDE_MIN_TILES")) min_tiles = atoi(e);
    if (d.B * cdiv(d.H, TH) * cdiv(d.W, TW) < min_tiles) return false;
    const size_t cmax = (size_t)(d.Cin > d.Cout ? d.Cin : d.Cout);
    return (size_t)d.B * cmax * d.H * d.W * 4 + (size_t)(d.W + 8) * 4 < WLEAN_MAX_BYTES;
}
static BwPlan bw_plan(const bnerv_wgrad_desc& d) {
    BwPlan p;
    const int mt = cdiv(d.Cout, 16);
    p.mtw = mt <= 3 ? mt : (mt == 4 ? 2 : 3);
    p.ngroups_m = cdiv(mt, p.mtw);
    p.ngroups_n = cdiv(d.Cin * 9 + 1, BW_NTW * 16);
    const int total = d.B * cdiv(d.H, BW_TH) * cdiv(d.W, 32);
    int s = (256 * 2) / (8 * p.ngroups_n * p.ngroups_m);   // slots per XCD with every block resident (2 per CU)
    const int want = cdiv(total, 8);
    if (s > want) s = want;
    if (s < 1) s = 1;
    p.slots = s;
    return p;
}
template <int IN, int SP, int MTW, int GM2>
int launch_bw(hipStream_t st, const WArgs& wa, const BwPlan& p) {
    constexpr int NS = Split<SP>::NS;
    size_t lds = (size_t)NS * BW_PIECE + 2 * BW_NPL * sizeof(float);
    const size_t red = (size_t)MTW * 16 * BW_NTW * 16 * sizeof(float);
    if (lds < red) lds = red;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bfw_kernel<IN, SP, MTW, GM2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int grid = 8 * p.slots * p.ngroups_n * p.ngroups_m;
    SidePack side;
    bnerv_side_take(wa.d.ctx, &side, 2 * grid);
    hipLaunchKernelGGL((wgrad_bfw_kernel<IN, SP, MTW, GM2>), dim3(grid), dim3(256), lds, st, wa, p.slots, p.ngroups_n, p.ngroups_m, side);
    BNERV_LAUNCH_CHECK("wgrad_bfw");
    return BNERV_OK;
}
template <int IN, int SP, int GM2>
int launch_bw_m(hipStream_t st, const WArgs& wa, const BwPlan& p) {
    if (p.mtw == 1) return launch_bw<IN, SP, 1, GM2>(st, wa, p);
    if (p.mtw == 2) return launch_bw<IN, SP, 2, GM2>(st, wa, p);
    return launch_bw<IN, SP, 3, GM2>(st, wa, p);
}
static int launch_bw_modes(hipStream_t st, const WArgs& wa, const BwPlan& p) {
    const bool x3 = bw_mode() == SP_BF16X3;
    if (wa.d.g_s == 3 || wa.d.g_s == 5)                    // (x3 / x5 up-convs: strided 4-B gradient loads)
        return x3 ? launch_bw_m<BNERV_IN_PLAIN, SP_BF16X3, 3>(st, wa, p) : launch_bw_m<BNERV_IN_PLAIN, SP_BF16X6, 3>(st, wa, p);
    if (wa.d.g_s == 2)                                     // (the up-convs: plain input, shuffled gradient)
        return x3 ? launch_bw_m<BNERV_IN_PLAIN, SP_BF16X3, 1>(st, wa, p) : launch_bw_m<BNERV_IN_PLAIN, SP_BF16X6, 1>(st, wa, p);
    if (wa.d.in_mode == BNERV_IN_AFFINE) return x3 ? launch_bw_m<BNERV_IN_AFFINE, SP_BF16X3, 0>(st, wa, p) : launch_bw_m<BNERV_IN_AFFINE, SP_BF16X6, 0>(st, wa, p);
    return x3 ? launch_bw_m<BNERV_IN_PLAIN, SP_BF16X3, 0>(st, wa, p) : launch_bw_m<BNERV_IN_PLAIN, SP_BF16X6, 0>(st, wa, p);
}

struct Plan { int mtw, ntw, n_mgroups, n_ngroups, nsplit; };

Plan make_plan(int B, int Cin, int Cout, int H, int W, int k) {
    Plan p;
    const int T = k * k, ncols = Cin * T + 1;
    const int mt = cdiv(Cout, 16), nt = cdiv(ncols, 16);
    if (k == 1) { p.mtw = 4; p.ntw = 1; }
    else if (mt == 1) { p.mtw = 1; p.ntw = 7; }
    else if (mt <= 3) { p.mtw = 3; p.ntw = 7; }
    else { p.mtw = 4; p.ntw = 4; }
    p.n_mgroups = cdiv(mt, p.mtw);
    p.n_ngroups = cdiv(nt, p.ntw);
    const int total_tiles = B * cdiv(H, TH) * cdiv(W, TW);
    const int groups = p.n_mgroups * p.n_ngroups;
    // everything resident at once (balanced static partition): ~3 blocks per CU for the small-register shapes, 2 otherwise
    const int per_cu = (p.mtw * p.ntw <= 7) ? 3 : 2;
    int target = (256 * per_cu) / groups;
    if (target < 1) target = 1;
    // fill the machine first (one tile per block until every slot is taken): the small layers are latency-bound, and the
    // per-block fixed cost (constant planes, wave reduction, one slab) is only a few microseconds
    // (beyond that, ~4 tiles per block amortise the fixed cost better than more, shorter blocks)
    int want = total_tiles <= target ? total_tiles : (total_tiles + 3) / 4;
    if (want < 1) want = 1;
    p.nsplit = want < target ? want : target;
    return p;
}

template <int KS, int IN, int GM, int MTW, int NTW>
int launch_w(hipStream_t st, const WArgs& wa, const Plan& p) {
    using G = Geo<KS>;
    constexpr int NPL = wgrad_npl<KS, NTW>();
    const int g_rows = wa.d.Cout < MTW * 16 ? wa.d.Cout : MTW * 16;
    size_t lds_fl = (size_t)g_rows * CSG + (size_t)(NPL + 2) * G::PLANE;
    if (lds_fl < (size_t)MTW * 16 * CSG) lds_fl = (size_t)MTW * 16 * CSG;         // the A reads of the dropped rows stay inside the block's LDS
    if (lds_fl < (size_t)MTW * 16 * NTW * 16) lds_fl = (size_t)MTW * 16 * NTW * 16;
    const size_t lds = lds_fl * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<KS, IN, GM, MTW, NTW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    dim3 grid(p.nsplit, p.n_mgroups * p.n_ngroups);
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, IN, GM, MTW, NTW>), grid, dim3(256), lds, st, wa);
    BNERV_LAUNCH_CHECK("conv_wgrad");
    return BNERV_OK;
}

template <int KS, int IN, int GM>
int launch_shape(hipStream_t st, const WArgs& wa, const Plan& p) {
    if constexpr (KS == 1) {
        return launch_w<1, IN, GM, 4, 1>(st, wa, p);
    } else {
        if (p.mtw == 1) return launch_w<3, IN, GM, 1, 7>(st, wa, p);
        if (p.mtw == 3) return launch_w<3, IN, GM, 3, 7>(st, wa, p);
        return launch_w<3, IN, GM, 4, 4>(st, wa, p);
    }
}

template <int KS>
int launch_modes(hipStream_t st, const WArgs& wa, const Plan& p) {
    const int in = wa.d.in_mode, gm = wa.d.g_mode;
    if (gm == BNERV_IN_TANHGRAD && in == BNERV_IN_PLAIN) return launch_shape<KS, BNERV_IN_PLAIN, BNERV_IN_TANHGRAD>(st, wa, p);
    if (gm == BNERV_IN_PLAIN || gm == BNERV_IN_UNSHUFFLE) {
        if (in == BNERV_IN_PLAIN) return launch_shape<KS, BNERV_IN_PLAIN, BNERV_IN_UNSHUFFLE>(st, wa, p);
        if constexpr (KS == 3) {
            if (in == BNERV_IN_AFFINE) return launch_shape<KS, BNERV_IN_AFFINE, BNERV_IN_UNSHUFFLE>(st, wa, p);
            if (in == BNERV_IN_GELU_AFFINE) return launch_shape<KS, BNERV_IN_GELU_AFFINE, BNERV_IN_UNSHUFFLE>(st, wa, p);
        }
    }
    return bnerv_set_error(BNERV_E_ARG, "conv_wgrad: unsupported (k=%d, in_mode=%d, g_mode=%d)", KS, in, gm);
}

}  // namespace

#if defined(BNERV_TRACE) || defined(BNERV_TRACE_BW)
extern "C" int bnerv_debug_trace_read_w(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_w), sizeof(g_trace_w)); }
#ifdef BNERV_TRACE
extern "C" int bnerv_debug_trace_read_pairf(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bnerv_q4::g_trace4), sizeof(bnerv_q4::g_trace4)); }   // this translation unit's copy (pairf_body.h stamps)
#endif
#endif
extern "C" size_t bnerv_conv_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int k) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (k != 1 && k != 3)) return 0;
    const Plan p = make_plan(B, Cin, Cout, H, W, k);
    bnerv_wgrad_desc t{};
    t.B = B; t.H = H; t.W = W; t.k = k;
    int nb = wlean_blocks(t) > p.nsplit ? wlean_blocks(t) : p.nsplit;           // covers whichever kernel the launcher picks
    if (k == 1) {
        bnerv_wgrad_desc t1{};
        t1.B = B; t1.Cin = Cin; t1.Cout = Cout; t1.H = H; t1.W = W; t1.k = 1; t1.g_s = 1;
        const int n1 = bnerv_wgrad1x1_slabs(t1);
        if (n1 > nb) nb = n1;
    }
    if (k == 3) {
        t.Cin = Cin; t.Cout = Cout;
        const WidePlan wp = wide_plan(t);
        if (8 * wp.slots > nb) nb = 8 * wp.slots;
        const BwPlan bp = bw_plan(t);
        if (8 * bp.slots > nb) nb = 8 * bp.slots;
    }
    return (size_t)nb * Cout * (Cin * k * k + 1) * sizeof(float);
}

extern "C" int bnerv_conv_wgrad(void* stream, const bnerv_wgrad_desc* dp) {
    BNERV_REQUIRE(dp != nullptr, "conv_wgrad: null descriptor");
    WArgs wa;
    wa.d = *dp;
    const bnerv_wgrad_desc& d = wa.d;
    BNERV_REQUIRE(d.k == 1 || d.k == 3, "conv_wgrad: k must be 1 or 3 (got %d)", d.k);
    BNERV_REQUIRE(d.B > 0 && d.Cin > 0 && d.Cout > 0 && d.H > 0 && d.W > 0, "conv_wgrad: bad dims");
    BNERV_REQUIRE(d.x && d.g && d.dw && d.ws, "conv_wgrad: null tensor");
    BNERV_REQUIRE(d.g_s >= 1 && d.Cout % (d.g_s * d.g_s) == 0, "conv_wgrad: bad g_s");
    if (d.in_mode == BNERV_IN_AFFINE || d.in_mode == BNERV_IN_GELU_AFFINE) BNERV_REQUIRE(d.scale && d.shift, "conv_wgrad: affine prologue needs scale/shift");
    if (d.g_mode == BNERV_IN_TANHGRAD) BNERV_REQUIRE(d.gaux && d.g_s == 1, "conv_wgrad: tanh-grad needs gaux");
    const Plan p = make_plan(d.B, d.Cin, d.Cout, d.H, d.W, d.k);
    const size_t need = bnerv_conv_wgrad_ws_bytes(d.B, d.Cin, d.Cout, d.H, d.W, d.k);
    if (d.ws_bytes < need) return bnerv_set_error(BNERV_E_WS, "conv_wgrad: workspace %zu < %zu", d.ws_bytes, need);
    wa.slab = reinterpret_cast<float*>(d.ws);
    wa.tiles_x = cdiv(d.W, TW);
    wa.tiles_y = cdiv(d.H, TH);
    wa.n_mgroups = p.n_mgroups;
    wa.n_ngroups = p.n_ngroups;
    wa.ncols = d.Cin * d.k * d.k + 1;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    wa.vec = ((d.W % 4 == 0) && al(d.x) && al(d.g) && al(d.gaux)) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {   // the stem stage (an image of <= 256 pixels, many output channels): written directly, no slabs
        const int rs = bnerv_stem_wgrad_try(st, d);
        if (rs != 1) return rs;
    }
    int rc = -1, n_slabs = p.nsplit;
    if (d.k == 1) {                                       // pointwise layers with 16 or more channels on both sides: the plain GEMM over pixels (wgrad1.hip)
        int n1 = 0;
        const int r1 = bnerv_wgrad1x1_try(st, d, &n1);
        if (r1 < 0) return r1;
        if (r1 == BNERV_OK) { rc = BNERV_OK; n_slabs = n1; }
    }
    if (rc == -1 && wlean_ok(wa)) {
        rc = d.k == 1 ? launch_wlean_modes<1>(st, wa) : launch_wlean_modes<3>(st, wa);
        if (rc == BNERV_OK) n_slabs = wlean_blocks(d);
    }
    if (rc == -1 && bw_ok(wa)) {                          // split 16-bit kernel for the multi-tile plain gradients
        const BwPlan bp = bw_plan(d);
        rc = launch_bw_modes(st, wa, bp);
        if (rc == BNERV_OK) n_slabs = 8 * bp.slots;
    }
    if (rc == -1 && wide_ok(wa)) {
        const WidePlan wp = wide_plan(d);
        if (wp.mtw) {
            wa.n_ngroups = wp.ngroups;
            wa.n_mgroups = wp.mgroups;
            rc = launch_wide_modes(st, wa, wp);
            if (rc == BNERV_OK) n_slabs = 8 * wp.slots;
            else { wa.n_ngroups = p.n_ngroups; wa.n_mgroups = p.n_mgroups; }
        }
    }
    if (rc == -1) rc = d.k == 1 ? launch_modes<1>(st, wa, p) : launch_modes<3>(st, wa, p);
    if (rc != BNERV_OK) return rc;
    const int count = d.Cout * wa.ncols;
    if (d.defer_finish && d.ctx) {                                  // (queued AFTER the launch: this launch may host older jobs, never its own)
        bnerv_side_push(d.ctx, st, wa.slab, n_slabs, count, wa.ncols, d.dw, d.db);
        return BNERV_OK;
    }
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(cdiv(count, 32)), dim3(1024), 0, st, wa.slab, n_slabs, d.Cout, wa.ncols, d.dw, d.db);
    BNERV_LAUNCH_CHECK("wgrad_finish");
    return BNERV_OK;
}

// ------------------------------------------------------------------------------------------------------------------ paired launch
namespace {
template <int EP, int WIN>
int launch_pair(hipStream_t st, bnerv_conv::KArgs& ka, const WArgs& wa, int* n_w_out) {
    bnerv_q4::q4_prepare(ka);
    {
        // shared-tile form (pairf_body.h): one block runs both roles on its tile from ONE staged copy of the gradient.  Default: the pairs
        // with a reducing epilogue (TAT convs: DGELU_SAVED / DSIN) from 1024 tiles on (two tiles per block at 512 blocks).  Measured on
        // MI355X, C1 (profiles/r05_pair_fused.md): 720p DSIN pair 81.0 against 83.5 us, DGELU_SAVED 76.5 against 74.5, PLAIN 68.5 against
        // 66.7 (kept on the interleaved pair); at 360x640 (900 tiles) the interleaved pair's three blocks per CU win (27 against 29 us).
        // HBM traffic of the DSIN pair: see the same file.  BNERV_PAIR_FUSED=<tiles>: every pair from that many tiles on; 0: off.
        const char* fe = getenv("BNERV_PAIR_FUSED");         // (read per call: the parity tests switch forms inside one process)
        const int fused_env = fe ? atoi(fe) : -1;
        constexpr bool red_ep = EP == BNERV_EP_DGELU_SAVED || EP == BNERV_EP_DSIN;
        const int fused_min = fused_env >= 0 ? fused_env : (red_ep ? 1024 : 0);
        const bnerv_conv_desc& c = ka.d;
        const bnerv_wgrad_desc& w = wa.d;
        if (fused_min > 0 && ka.total_items >= fused_min && ka.total_items >= 8 && c.x == w.g && c.Cin == w.Cout && c.Cout == w.Cin && c.B == w.B &&
            w.g_s == 1 && (w.g_mode == BNERV_IN_PLAIN || w.g_mode == BNERV_IN_UNSHUFFLE)) {
            const size_t ldsf = pair_fused_lds_bytes();
            static bool attrf = false;
            if (!attrf) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_fused_kernel<EP, WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
                attrf = true;
            }
            int grid = ka.total_items < 512 ? (ka.total_items & ~7) : 512;
            if (n_w_out) *n_w_out = grid;
            SidePack side;
            bnerv_side_take(w.ctx, &side, 2 * grid);
            if constexpr (red_ep && WIN == BNERV_IN_AFFINE) {
                // fold form (pairf_body.h pair_fold_body): the input tile staged raw by LDS-DMA, the affine applied by the slab reduction.  One sample
                // per launch (the fold is per sample), the reducing epilogue's raw-input operand must BE the weight gradient's input, and the
                // workspace must hold `grid` slabs of Cout x (ncols + 8).  BNERV_PAIR_FOLD=0: the transforming form above.
                const char* ff = getenv("BNERV_PAIR_FOLD");
                const float* raw_aux = EP == BNERV_EP_DSIN ? c.aux0 : c.aux1;
                if (!(ff && ff[0] == '0') && c.B == 1 && raw_aux == w.x && w.scale && w.shift && (reinterpret_cast<uintptr_t>(w.x) & 15) == 0 &&
                    (size_t)grid * w.Cout * (wa.ncols + 8) * sizeof(float) <= w.ws_bytes) {
                    const size_t ldsd = pair_fold_lds_bytes();
                    static bool attrd = false;
                    if (!attrd) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_fold_kernel<EP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd);
                        attrd = true;
                    }
                    hipLaunchKernelGGL((pair_fold_kernel<EP>), dim3(grid), dim3(256), ldsd, st, ka, wa, side);
                    BNERV_LAUNCH_CHECK("pair_fold");
                    if (n_w_out) *n_w_out = -grid;         // negative: the caller queues a FOLD slab reduction
                    return BNERV_OK;
                }
            }
            hipLaunchKernelGGL((pair_fused_kernel<EP, WIN>), dim3(grid), dim3(256), ldsf, st, ka, wa, side);
            BNERV_LAUNCH_CHECK("pair_fused");
            return BNERV_OK;
        }
    }
    const int n_grows = wa.d.Cout <= 12 ? 12 : 16;
    size_t lds = bnerv_q4::q4_lds_bytes();
    const size_t lw = wlean_lds_bytes<3>(n_grows);
    if (lw > lds) lds = lw;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_pair_kernel<EP, WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    int n_conv = ka.total_items < 768 ? ka.total_items : 768;          // 3 conv blocks per CU when the layer is large
    n_conv = (n_conv + 7) & ~7;                                        // multiple of 8: block b runs on XCD b % 8 for BOTH halves' slices
    int n_w = wlean_blocks(wa.d);
    int grid = n_conv + n_w;
    if (n_w_out) *n_w_out = n_w;
    // BNERV_PAIR_MIX: blocks per role of the interleaved form (0: conv blocks first, then the weight gradient's); BNERV_PAIR_PAT: the
    // role of XCD-local slot s is (s + (s >> PAT)) & 1 (0: s & 1).  Measured on C1 (1.745 ms with the roles one after the other):
    // 384 blocks per role 1.680 (PAT 0), 1.642 (PAT 5: the parity flips every 32 slots = the CUs of an XCD, so every CU holds both
    // roles), 1.664 (6), 1.68 (3, 4, 7); 376 / 368 per role as 384, 512 per role (not all resident) 1.783.
    static const int mix3 = [] { const char* e = getenv("BNERV_PAIR_MIX"); return e ? atoi(e) : 384; }();
    static const int pat = [] { const char* e = getenv("BNERV_PAIR_PAT"); return e ? atoi(e) : 5; }();
    const int mix = mix3;
    if (mix > 0 && ka.total_items >= 2 * mix && n_w >= mix) {          // large layer: mix blocks per role, all resident, roles interleaved
        n_w = mix; n_conv = -mix; grid = 2 * mix;
        if (n_w_out) *n_w_out = n_w;
    }
    SidePack side;
    bnerv_side_take(wa.d.ctx, &side, 2 * n_w);
    hipLaunchKernelGGL((conv_wgrad_pair_kernel<EP, WIN>), dim3(grid), dim3(256), lds, st, ka, wa, n_grows, n_conv, side, pat);
    BNERV_LAUNCH_CHECK("conv_wgrad_pair");
    return BNERV_OK;
}

// ---- paired launch, low-resolution form: the data gradient of convs.hip's family (one 4x16 tile x 16 couts per block) next to a wide
// weight gradient (wgrad_wide / wgrad_bfw bodies) that reads the same incoming gradient.  Below 180x320 each of the two is a launch of
// 60..240 blocks that lasts 7..12 us whatever its arithmetic (staging latency, one K loop, one epilogue); together they are one such
// launch.  Blocks [0, n_c8) are the conv's (n_c8 = its block count rounded up to 8, so that the weight gradient's logical XCD of block
// vb is the physical one), the rest the weight gradient's.
struct WideRoleArgs { int slots, ngn, ngm; };
template <int IN, int GM2, int MTW, int NTW>
struct WideRole {
    static __device__ __forceinline__ void run(const WArgs& wa, const WideRoleArgs& r, const SidePack& side, int vb, int vgrid) { wgrad_wide_body<IN, GM2, MTW, NTW>(wa, r.slots, side, vb, vgrid); }
    static size_t lds(const WArgs& wa) {
        using G = Geo<3>;
        constexpr int NPL = wgrad_npl<3, NTW>();
        const int g_rows = wa.d.Cout < MTW * 16 ? wa.d.Cout : MTW * 16;
        size_t lds_fl = (size_t)g_rows * CSG + (size_t)(NPL + 2) * G::PLANE + 64;
        if (lds_fl < (size_t)MTW * 16 * CSG) lds_fl = (size_t)MTW * 16 * CSG;
        if (lds_fl < (size_t)MTW * 16 * NTW * 16) lds_fl = (size_t)MTW * 16 * NTW * 16;
        return lds_fl * sizeof(float);
    }
};
template <int IN, int SP, int MTW, int GM2>
struct BfwRole {
    static __device__ __forceinline__ void run(const WArgs& wa, const WideRoleArgs& r, const SidePack& side, int vb, int vgrid) { wgrad_bfw_body<IN, SP, MTW, GM2>(wa, r.slots, r.ngn, r.ngm, side, vb, vgrid); }
    static size_t lds(const WArgs&) {
        constexpr int NS = Split<SP>::NS;
        size_t lds = (size_t)NS * BW_PIECE + 2 * BW_NPL * sizeof(float);
        const size_t red = (size_t)MTW * 16 * BW_NTW * 16 * sizeof(float);
        return lds < red ? red : lds;
    }
};
template <class WR, int CIN, int CEP, int CNQ>
__global__ __launch_bounds__(256, 2) void small_pair_kernel(const bnerv_convs::SArgs sa, const WArgs wa, const WideRoleArgs r, const int n_c, const int n_c8,
                                                            const int c_tiles, const int c_groups, const SidePack side) {
    const int b = (int)blockIdx.x;
    if (b < n_c8) {
        if (b >= n_c) return;                              // (padding block)
        const int q = b / c_tiles;
        bnerv_convs::conv_small_body<CIN, CEP, CNQ>(sa, b - q * c_tiles, q % c_groups, q / c_groups);
    } else {
        WR::run(wa, r, side, b - n_c8, (int)gridDim.x - n_c8);
    }
}
template <class WR, int CIN, int CEP, int CNQ>
int launch_small_pair(hipStream_t st, const bnerv_convs::SArgs& sa, const WArgs& wa, const WideRoleArgs& r, int n_w) {
    size_t lds = bnerv_convs::convs_lds_bytes<CNQ>();
    const size_t lw = WR::lds(wa);
    if (lw > lds) lds = lw;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_pair_kernel<WR, CIN, CEP, CNQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    const int c_tiles = sa.tiles_x * sa.tiles_y, c_groups = cdiv(sa.d.Cout, 16);
    const int n_c = c_tiles * c_groups * sa.d.B, n_c8 = (n_c + 7) & ~7;
    SidePack side;
    bnerv_side_take(wa.d.ctx, &side, 2 * n_w);
    hipLaunchKernelGGL((small_pair_kernel<WR, CIN, CEP, CNQ>), dim3(n_c8 + n_w), dim3(256), lds, st, sa, wa, r, n_c, n_c8, c_tiles, c_groups, side);
    BNERV_LAUNCH_CHECK("small_pair");
    return BNERV_OK;
}

// 1: not a pair of this form.  On BNERV_OK *n_slabs is the weight gradient's slab count (the caller queues its reduction).
static int small_pair_try(hipStream_t st, const bnerv_conv_desc& c, WArgs& wa, int* n_slabs) {
    const bnerv_wgrad_desc& w = wa.d;
    { static const bool off = [] { const char* e = getenv("BNERV_PAIR_SMALL"); return e && e[0] == '0'; }(); if (off) return 1; }
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int cvec = ((c.W % 4 == 0) && al(c.x) && al(c.out) && al(c.out2) && al(c.aux0) && al(c.aux1) && al(c.aux2)) ? 1 : 0;
    if (!bnerv_convs_shape_ok(c, cvec)) return 1;
    if (c.in_mode != BNERV_IN_UNSHUFFLE && c.Cin > 32) return 1;                           // (convs.hip's 96-channel form has no paired instantiation)
    if (c.ep_mode == BNERV_EP_PLAIN && bnerv_conv_splitk_ws_bytes(&c) != 0) return 1;     // (a split-K layer: its own launches)
    if (!(w.k == 3 && w.B == c.B && w.H == c.H && w.W == c.W && w.defer_finish && w.ctx && w.ctx == c.ctx)) return 1;
    bnerv_convs::SArgs sa;
    sa.d = c;
    sa.tiles_x = cdiv(c.W, bnerv_convs::STW);
    sa.tiles_y = cdiv(c.H, bnerv_convs::STH);
    const int cnq = c.in_mode == BNERV_IN_UNSHUFFLE ? (c.Cin <= 32 ? 8 : 16) : (c.Cin <= 16 ? 4 : 8);
    WideRoleArgs r{0, 0, 0};
    int rc = 1;
    if (wlean_ok(wa)) return 1;                            // (the lean weight gradient pairs with conv4.hip's family)
#define BNERV_SP(WR, I, E, Q) if (c.in_mode == I && c.ep_mode == E && cnq == Q) rc = launch_small_pair<WR, I, E, Q>(st, sa, wa, r, n_w);
    if (bw_ok(wa)) {
        if (bw_mode() != (int)SP_BF16X6) return 1;
        const BwPlan bp = bw_plan(w);
        r.slots = bp.slots; r.ngn = bp.ngroups_n; r.ngm = bp.ngroups_m;
        const int n_w = 8 * bp.slots * bp.ngroups_n * bp.ngroups_m;
        *n_slabs = 8 * bp.slots;
        if (w.g_s == 2 && w.in_mode == BNERV_IN_PLAIN && w.g_mode == BNERV_IN_UNSHUFFLE) {          // an up-conv's (dW | d input)
            if (bp.mtw == 2) { using WR = BfwRole<BNERV_IN_PLAIN, SP_BF16X6, 2, 1>; BNERV_SP(WR, BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN, 8) BNERV_SP(WR, BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN, 16) }
            if (bp.mtw == 3) { using WR = BfwRole<BNERV_IN_PLAIN, SP_BF16X6, 3, 1>; BNERV_SP(WR, BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN, 16) }
        } else if (w.g_s == 1 && w.in_mode == BNERV_IN_AFFINE && w.g_mode != BNERV_IN_TANHGRAD) {  // a TAT conv's (dW | d input), 17..32 channels
            if (bp.mtw == 2) { using WR = BfwRole<BNERV_IN_AFFINE, SP_BF16X6, 2, 0>; BNERV_SP(WR, BNERV_IN_PLAIN, BNERV_EP_DGELU_SAVED, 8) BNERV_SP(WR, BNERV_IN_PLAIN, BNERV_EP_DSIN, 8) }
        }
        return rc;
    }
    if (wide_ok(wa)) {
        const WidePlan wp = wide_plan(w);
        if (!(wp.mtw == 1 && wp.ntw == 8)) return 1;
        const int ng0 = wa.n_ngroups, mg0 = wa.n_mgroups;
        wa.n_ngroups = wp.ngroups;
        wa.n_mgroups = wp.mgroups;
        r.slots = wp.slots;
        const int n_w = 8 * wp.slots * wp.ngroups * wp.mgroups;
        *n_slabs = 8 * wp.slots;
        if (w.g_s == 1 && w.in_mode == BNERV_IN_AFFINE && w.g_mode != BNERV_IN_TANHGRAD) {         // a TAT conv's (dW | d input), 13..16 channels
            using WR = WideRole<BNERV_IN_AFFINE, 0, 1, 8>;
            BNERV_SP(WR, BNERV_IN_PLAIN, BNERV_EP_DGELU_SAVED, 4) BNERV_SP(WR, BNERV_IN_PLAIN, BNERV_EP_DSIN, 4)
        }
        if (rc == 1) { wa.n_ngroups = ng0; wa.n_mgroups = mg0; }
        return rc;
    }
#undef BNERV_SP
    return 1;
}
}  // namespace

// ---- the 1x1 output head's backward as ONE streaming pass (head_layer 1x1 C -> K <= 4 + OutImg tanh, model_nerv.py:41, :56-57, model_blocks.py:57-63):
//     dy[k][p] = g[k][p] * 0.5 (1 - (2 img[k][p] - 1)^2);   dx[c][p] = sum_k W[k][c] dy[k][p];   dW[k][c] = sum_p dy[k][p] x[c][p];   db[k] = sum_p dy[k][p]
// The two halves were a streaming data-gradient kernel (13 us at 720p) and the 1x1 lean weight gradient (15.8 us): each read g and img, one
// wrote dx, the other read x -- 132 MB for 110 MB of distinct tensors and two launches for a pass that is 75 fma per pixel.  Here a thread takes
// 4 consecutive pixels per step of a grid-stride loop, keeps its K (C + 1) partial sums in registers and the block writes ONE slab [K][C + 1]
// (the weight gradient's slab layout: the deferred slab reduction finishes it as for every other layer).
constexpr int HB_KMAX = 4, HB_CMAX = 16;
struct HeadBwdArgs { const float* g; const float* img; const float* x; const float* w; float* dx; float* slab; int B, K, C, transposed; size_t HW; int nq; };
template <int K, int C>
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(const HeadBwdArgs a) {
    __shared__ float s_w[HB_KMAX * HB_CMAX];
    __shared__ float s_red[4][HB_KMAX * (HB_CMAX + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < K * C) {                                     // W(k, c): the conv descriptor's transposed weights are stored [k][c], direct ones [c][k]
        const int k = tid / C, c = tid - k * C;
        s_w[k * HB_CMAX + c] = a.transposed ? a.w[k * C + c] : a.w[c * K + k];
    }
    if (lane < K * (C + 1)) s_red[wave][(lane / (C + 1)) * (HB_CMAX + 1) + lane % (C + 1)] = 0.f;      // this wave's running sums (only its lane 0 adds to them)
    __syncthreads();
    const int total = a.B * a.nq;                          // pixel quads of all samples
    const int rounds = (total + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);       // block-uniform: the wave sums need every lane
#pragma unroll 1
    for (int it = 0; it < rounds; ++it) {
        const int q = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + tid;
        const bool live = q < total;
        const int b = live ? q / a.nq : 0, qq = live ? q - b * a.nq : 0;
        f32x4 dy[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t o = ((size_t)b * K + k) * a.HW + (size_t)qq * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.g + o), im = *reinterpret_cast<const f32x4*>(a.img + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = 2.0f * im[e] - 1.0f; dy[k][e] = live ? v[e] * 0.5f * (1.0f - t * t) : 0.f; }
        }
        f32x4 xv[C];                                       // every load of the thread in flight before the first use
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = *reinterpret_cast<const f32x4*>(a.x + ((size_t)b * C + c) * a.HW + (size_t)qq * 4);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float sm = wave_sum((dy[k][0] + dy[k][1]) + (dy[k][2] + dy[k][3]));
            if (lane == 0) s_red[wave][k * (HB_CMAX + 1) + C] += sm;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float wv = s_w[k * HB_CMAX + c];
                float pw = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = fmaf(dy[k][e], wv, r[e]); pw = fmaf(dy[k][e], xv[c][e], pw); }
                pw = wave_sum(pw);
                if (lane == 0) s_red[wave][k * (HB_CMAX + 1) + c] += pw;
            }
            if (a.dx && live) *reinterpret_cast<f32x4*>(a.dx + ((size_t)b * C + c) * a.HW + (size_t)qq * 4) = r;
            __builtin_amdgcn_sched_barrier(0);             // (one channel's three wave sums at a time: interleaving all 36 costs 256 registers)
        }
    }
    __syncthreads();
    if (tid < K * (C + 1)) {                               // the four waves in a fixed order
        const int k = tid / (C + 1), c = tid - k * (C + 1);
        const int i = k * (HB_CMAX + 1) + c;
        a.slab[(size_t)blockIdx.x * K * (C + 1) + tid] = (s_red[0][i] + s_red[1][i]) + (s_red[2][i] + s_red[3][i]);
    }
}

// 1: not the head's pair; BNERV_OK: launched, *n_slabs slabs of K x (C + 1) floats in w.ws
static int head_pair_try(hipStream_t st, const bnerv_conv_desc& c, const bnerv_wgrad_desc& w, int* n_slabs) {
    { const char* e = getenv("BNERV_PAIR_HEAD"); if (e && e[0] == '0') return 1; }      // A/B switch, read per call
    if (!(c.k == 1 && w.k == 1 && c.in_mode == BNERV_IN_TANHGRAD && c.ep_mode == BNERV_EP_PLAIN && c.out_s == 1 && w.g_mode == BNERV_IN_TANHGRAD && w.in_mode == BNERV_IN_PLAIN)) return 1;
    if (!(c.x && c.aux0 && c.w && c.out && w.x && w.g && w.gaux && w.dw && w.ws && w.defer_finish && w.ctx && w.ctx == c.ctx)) return 1;
    if (!(c.x == w.g && c.aux0 == w.gaux && c.Cin == w.Cout && c.Cout == w.Cin && c.B == w.B && c.H == w.H && c.W == w.W)) return 1;     // one head, one gradient
    const int K = c.Cin, C = c.Cout;
    if (K != 3 || C != 12 || ((size_t)c.H * c.W) % 4 != 0) return 1;            // (instantiated for the reference's heads: 12 -> 3; HNeRV's heads are 3x3)
    if (c.transposed ? !(c.wCo == K && c.wCi == C) : !(c.wCo == C && c.wCi == K)) return 1;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!(al(c.x) && al(c.aux0) && al(c.out) && al(w.x))) return 1;
    const size_t HW = (size_t)c.H * c.W;
    const int nq = (int)(HW / 4);
    // one quad per thread up to 1024 blocks (every load of a thread is in flight at once: the pass is one memory round trip per wave; a
    // first version with four quads per thread on 225 blocks ran 36.8 us -- four serialised round trips on less than one wave per SIMD)
    int blocks = cdiv(c.B * nq, 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if ((size_t)blocks * K * (C + 1) * sizeof(float) > w.ws_bytes) return 1;
    HeadBwdArgs a{c.x, c.aux0, w.x, c.w, c.out, reinterpret_cast<float*>(w.ws), c.B, K, C, c.transposed, HW, nq};
    hipLaunchKernelGGL((head1x1_bwd_kernel<3, 12>), dim3(blocks), dim3(256), 0, st, a);
    BNERV_LAUNCH_CHECK("head1x1_bwd");
    *n_slabs = blocks;
    return BNERV_OK;
}

// Returns BNERV_OK when both were launched together, 1 when the pair is not one this launch takes (the caller then issues
// bnerv_conv_wgrad and bnerv_conv_igemm separately, in that order), a negative BNERV_E_* on error.
extern "C" int bnerv_conv_wgrad_pair(void* stream, const bnerv_conv_desc* cdp, const bnerv_wgrad_desc* wdp) {
    BNERV_REQUIRE(cdp != nullptr && wdp != nullptr, "conv_wgrad_pair: null descriptor");
    static const bool off = [] { const char* e = getenv("BNERV_PAIR"); return e && e[0] == '0'; }();
    if (off) return 1;
    bnerv_conv::KArgs ka;
    ka.d = *cdp;
    bnerv_conv_desc& c = ka.d;
    WArgs wa;
    wa.d = *wdp;
    const bnerv_wgrad_desc& w = wa.d;
    // the conv half: a 3x3 stride-1 data gradient (plain input, or the unshuffle(2) prologue of an up-conv's), epilogue PLAIN / DGELU_SAVED / DSIN
    if (c.in_mode == BNERV_IN_UNSHUFFLE && c.in_s == 1) c.in_mode = BNERV_IN_PLAIN;
    if (c.k == 1) {                                        // form H: the 1x1 output head (tanh-grad prologue), one streaming pass
        int ns = 0;
        const int rh = head_pair_try(reinterpret_cast<hipStream_t>(stream), c, w, &ns);
        if (rh != BNERV_OK) return rh;
        bnerv_side_push(w.ctx, reinterpret_cast<hipStream_t>(stream), w.ws, ns, w.Cout * (w.Cin + 1), w.Cin + 1, w.dw, w.db);
        return BNERV_OK;
    }
    if (!(c.k == 3 && c.out_s == 1 && c.x && c.w && c.out && c.B > 0)) return 1;
    // form 0: the stem stage (an image of <= 256 pixels: stem.hip) -- dW / db are written directly, the data gradient's K-slice slabs
    // (c.partial, bnerv_conv_splitk_ws_bytes) are summed by a deferred reduction on the same context
    if (c.ep_mode == BNERV_EP_PLAIN && c.partial && w.k == 3 && w.x && w.g && w.dw && w.ctx && w.ctx == c.ctx) {
        int ns = 0;
        const int rs = bnerv_stem_pair_try(reinterpret_cast<hipStream_t>(stream), c, w, &ns);
        if (rs < 0) return rs;
        if (rs == BNERV_OK) {
            bnerv_side_push(c.ctx, reinterpret_cast<hipStream_t>(stream), c.partial, ns, c.B * c.Cout * c.H * c.W, 0, c.out, nullptr);
            return BNERV_OK;
        }
    }
    if (!((c.in_mode == BNERV_IN_PLAIN && c.in_s == 1) || (c.in_mode == BNERV_IN_UNSHUFFLE && c.in_s == 2))) return 1;
    if (!(c.ep_mode == BNERV_EP_DGELU_SAVED || c.ep_mode == BNERV_EP_DSIN || c.ep_mode == BNERV_EP_PLAIN)) return 1;
    if (c.ep_mode != BNERV_EP_PLAIN && !(c.aux0 && c.aux1 && c.scale && c.partial)) return 1;
    if (c.ep_mode == BNERV_EP_DSIN && !c.aux2) return 1;
    if (c.transposed ? !(c.Cout == c.wCi && c.Cin == c.wCo) : !(c.Cout == c.wCo && c.Cin == c.wCi)) return 1;
    // the weight-gradient half: a deferred 3x3 weight gradient on the same image size and context
    if (!(w.k == 3 && w.x && w.g && w.dw && w.ws && w.B == c.B && w.H == c.H && w.W == c.W && w.defer_finish && w.ctx && w.ctx == c.ctx)) return 1;
    if (!(w.in_mode == BNERV_IN_PLAIN || w.in_mode == BNERV_IN_AFFINE) || !(w.g_mode == BNERV_IN_PLAIN || w.g_mode == BNERV_IN_UNSHUFFLE)) return 1;
    if (w.in_mode == BNERV_IN_AFFINE && !(w.scale && w.shift)) return 1;
    if (!(w.g_s == 1 || (w.g_s == 2 && w.g_mode == BNERV_IN_UNSHUFFLE && w.Cout % 4 == 0))) return 1;
    if (w.ws_bytes < bnerv_conv_wgrad_ws_bytes(w.B, w.Cin, w.Cout, w.H, w.W, w.k)) return bnerv_set_error(BNERV_E_WS, "conv_wgrad_pair: weight-gradient workspace too small");
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const Plan p = make_plan(w.B, w.Cin, w.Cout, w.H, w.W, w.k);
    wa.slab = reinterpret_cast<float*>(w.ws);
    wa.tiles_x = cdiv(w.W, TW);
    wa.tiles_y = cdiv(w.H, TH);
    wa.n_mgroups = p.n_mgroups;
    wa.n_ngroups = p.n_ngroups;
    wa.ncols = w.Cin * 9 + 1;
    wa.vec = ((w.W % 4 == 0) && al(w.x) && al(w.g)) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = 1, n_slabs = 0;

    // form 1: conv4.hip's 12-channel family next to the lean weight gradient (roles interleaved per XCD on large layers)
    ka.tiles_x = cdiv(c.W, bnerv_conv::TW);
    ka.tiles_y = cdiv(c.H, bnerv_conv::TH);
    ka.vec = ((c.W % 4 == 0) && al(c.x) && al(c.out) && al(c.aux0) && al(c.aux1) && al(c.aux2)) ? 1 : 0;
    ka.ksplit = 1;
    ka.chunks_per_split = 0;
    ka.magic_tiles = ka.magic_tiles_x = 0;
    static const bool q4off = [] { const char* e = getenv("BNERV_Q4"); return e && e[0] == '0'; }();
    if (!q4off && c.in_mode == BNERV_IN_PLAIN && w.g_s == 1 && bnerv_q4::q4_shape_ok(ka) && wlean_ok(wa)) {
#define BNERV_PAIR_CASE(E, I) if (c.ep_mode == E && w.in_mode == I) rc = launch_pair<E, I>(st, ka, wa, &n_slabs);
        BNERV_PAIR_CASE(BNERV_EP_DGELU_SAVED, BNERV_IN_AFFINE)
        BNERV_PAIR_CASE(BNERV_EP_DSIN, BNERV_IN_AFFINE)
        BNERV_PAIR_CASE(BNERV_EP_PLAIN, BNERV_IN_PLAIN)
#undef BNERV_PAIR_CASE
    }
    // form 3: the wide split conv next to the wide split weight gradient, roles interleaved (layers that fill the chip in both roles)
    if (rc == 1 && !wlean_ok(wa) && bw_ok(wa) && bw_mode() == (int)SP_BF16X6) {
        const BwPlan bp = bw_plan(w);
        rc = bnerv_convbf_pair_try(st, c, ka.vec, wa, bp.mtw, bp.ngroups_n, bp.ngroups_m, bp.slots, &n_slabs);
    }
    // form 2: convs.hip's low-resolution family next to a wide weight gradient
    if (rc == 1) rc = small_pair_try(st, c, wa, &n_slabs);
    if (rc != BNERV_OK) return rc;
    if (n_slabs < 0)                                       // the shared-tile pair's fold form: slabs of ncols + 8 columns, the affine applied by the reduction
        bnerv_side_push(w.ctx, st, wa.slab, -n_slabs, w.Cout * wa.ncols, wa.ncols, w.dw, w.db, w.scale, w.shift);
    else
        bnerv_side_push(w.ctx, st, wa.slab, n_slabs, w.Cout * wa.ncols, wa.ncols, w.dw, w.db);     // the slab reduction rides on a later launch
    return BNERV_OK;
}
