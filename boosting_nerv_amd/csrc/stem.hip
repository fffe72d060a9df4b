// stem.hip -- the 3x3 up-conv of the STEM stage in backward: weight gradient and data gradient on an image of at most 256 pixels
// (reference call sites: model_nerv.py:45-61 / model_blocks.py:196-220, the first NeRVBlock: fc_dim channels at 9x16 -> fc_dim * 25
// through PixelShuffle(5); 2x9x16 latents of the HNeRV decoders).
//
// The tiled kernels are built for many pixels and few channels: a 9x16 image is 2 half-empty 8x32 tiles, so the wide weight-gradient
// kernel keeps 2 of 8 XCDs busy and still writes (and later reduces) 8 slabs of the 750 x 271 result (27 us), and the data gradient
// (K = 750 * 9 = 6750 against 144 x 30 outputs) runs as a split-K of the general implicit GEMM (22 + 5 us).  Here the whole image
// lives in LDS and the GEMMs are laid out the other way round:
//   * wgrad_tiny: dW[co][(ci, tap)] = sum over pixels of g[co][p] * x[ci][p + tap]: M = 16 output channels per block, N = (ci, tap)
//     columns (8 tiles of 16 per block, 2 per wave), K = the pixels of every sample -- nothing is split over K, so the block stores
//     its part of dW (and db) directly: no slabs, no reduction;
//   * dgrad_tiny: dx[n][p] = sum over (c, tap) of g[c][p + tap] * w[c][n][flipped tap]: M = pixels, N = the (<= 96) output channels,
//     K split over blocks in slices of 8 input channels (K = 72 per block), one slab per block, reduced by bnerv_reduce_slabs.
// Both read the gradient through the pixel-unshuffle gather (g stored shuffled by s in {1, 2, 3, 5}) with 4-byte loads -- the image
// is tiny, the loads are few.  v_mfma_f32_16x16x4_f32, exact f32 products like every f32 kernel of this build.
#include "common.h"
#include <stdlib.h>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TINY_MAX_PX = 256;
constexpr int WT_NTW = 2, WT_NTG = 4 * WT_NTW;            // weight gradient: N tiles per wave / per block
constexpr int WT_MAX_XFLOATS = 30000;                    // padded input image in LDS (120 KB)
constexpr int DT_CS = 8;                                 // data gradient: input channels per block (K = 72 = 18 steps)
constexpr int DT_KSTEPS = DT_CS * 9 / 4;

__host__ __device__ inline int row_stride4(int n) {       // >= n, == 4 (mod 32): 16 rows x 4 k-lanes spread over the banks
    int v = (n + 3) & ~3;
    v += ((4 - v) % 32 + 32) % 32;
    return v;
}

// n independent (load, LDS store) pairs spread over the block, U loads in flight per thread before the first store (a plain loop waits for
// each load before its store: 20+ serialised round trips per block)
template <int U, class LD, class ST>
__device__ __forceinline__ void staged(const int n, LD ld, ST st) {
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + u * 256; v[u] = i < n ? ld(i) : 0.f; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + u * 256; if (i < n) st(i, v[u]); }
    }
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct WTArgs { const float* x; const float* g; float* dw; float* db; int B, Cin, Cout, H, W, s; };

// base of conv-space channel co in a gradient stored pixel-shuffled by s ([B][C / s^2][H s][W s]); pixel (y, x) adds (y s)(W s) + x s
__device__ __forceinline__ int g_base(int b, int co, int C, int H, int W, int s) {
    const int ss = s * s, cf = co / ss, r = co - cf * ss, i = r / s, j = r - i * s;
    return ((b * (C / ss) + cf) * (H * s) + i) * (W * s) + j;
}

__device__ __forceinline__ void wgrad_tiny_body(const WTArgs& a, const int bx, const int by) {     // (bx, by): the block's cout tile / column group
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, W = a.W, HW = H * W, RS = W + 2, PL = (H + 2) * RS;
    const int HWp = (HW + 3) & ~3, HWs = row_stride4(HW);
    float* s_x = smem;                                   // [Cin][PL] zero border
    float* s_g = s_x + ((a.Cin * PL + 3) & ~3);          // [16][HWs]
    int* s_px = reinterpret_cast<int*>(s_g + 16 * HWs);  // [HWp]: pixel -> offset in a padded plane (of its interior origin)
    int* s_go = s_px + HWp;                              // [HWp]: pixel -> offset in the shuffled gradient plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int co_base = bx * 16, nW = a.Cin * 9;
    int coloff[WT_NTW];
#pragma unroll
    for (int t = 0; t < WT_NTW; ++t) {
        const int col = (by * WT_NTG + wave + 4 * t) * 16 + li;
        const int ci = col / 9, tap = col - ci * 9, dy = tap / 3, dx = tap - dy * 3;
        coloff[t] = col < nW ? ci * PL + dy * RS + dx : 0;                       // (columns beyond the matrix: any valid address, never stored)
    }
    for (int i = tid; i < HWp; i += 256) {
        const int y = i / W, x = i - y * W;
        s_px[i] = i < HW ? y * RS + x : 0;
        s_go[i] = i < HW ? (y * a.s) * (W * a.s) + x * a.s : 0;
    }
    for (int i = tid; i < a.Cin * PL; i += 256) s_x[i] = 0.f;                    // the borders stay zero for every sample
    for (int i = tid; i < 16 * HWs; i += 256) s_g[i] = 0.f;                      // ... and so do the padding pixels / rows beyond Cout
    __syncthreads();
    f32x4 acc[WT_NTW];
#pragma unroll
    for (int t = 0; t < WT_NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs = 0.f;
    const unsigned hw_magic = (unsigned)(((1ull << 32) + HW - 1) / HW);          // n / HW for n < 2^16 (HW <= 256)
    const int gm = tid >> 4, gp0 = tid & 15, gco = co_base + gm;                 // gradient staging: thread -> (row, pixels gp0 + 16 j)
    for (int b = 0; b < a.B; ++b) {
        if (b) __syncthreads();
        const float* xb = a.x + (size_t)b * a.Cin * HW;
        float gv[16];                                        // the block's 16 gradient rows: every load in flight before the input image's
        const float* gr = a.g + (gco < a.Cout ? g_base(b, gco, a.Cout, H, W, a.s) : 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int p = gp0 + 16 * u; gv[u] = (gco < a.Cout && p < HW) ? gr[s_go[p]] : 0.f; }
        staged<20>(a.Cin * HW, [&](int i) { return xb[i]; },
                   [&](int i, float v) { const int c = (int)__umulhi((unsigned)i, hw_magic), p = i - c * HW; s_x[c * PL + RS + 1 + s_px[p]] = v; });
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int p = gp0 + 16 * u; if (p < HW) s_g[gm * HWs + p] = gv[u]; }
        __syncthreads();
        {   // K loop: the operands of step st + 1 (and the pixel offset of st + 2) are loaded under the products of step st
            const int nst = HWp / 4;
            const float* ga = s_g + li * HWs + kq;
            const int* pq = s_px + kq;
            float av = ga[0], bv[WT_NTW];
            int po = pq[0];
#pragma unroll
            for (int t = 0; t < WT_NTW; ++t) bv[t] = s_x[coloff[t] + po];
            po = pq[nst > 1 ? 4 : 0];
            for (int st = 0; st < nst; ++st) {
                const int s1 = st + 1 < nst ? st + 1 : st, s2 = st + 2 < nst ? st + 2 : nst - 1;
                const float av_n = ga[4 * s1];
                float bn[WT_NTW];
#pragma unroll
                for (int t = 0; t < WT_NTW; ++t) bn[t] = s_x[coloff[t] + po];
                const int po_n = pq[4 * s2];
#pragma unroll
                for (int t = 0; t < WT_NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
                av = av_n; po = po_n;
#pragma unroll
                for (int t = 0; t < WT_NTW; ++t) bv[t] = bn[t];
            }
        }
        if (by == 0) {                               // bias gradient: the row sums of g (wave w: rows 4w .. 4w + 3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sm = 0.f;
                for (int p = lane; p < HW; p += 64) sm += s_g[(wave * 4 + r) * HWs + p];
                sm = wave_sum_f(sm);
                if (lane == r) dbs += sm;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < WT_NTW; ++t) {
        const int col = (by * WT_NTG + wave + 4 * t) * 16 + li;
        if (col >= nW) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co_base + 4 * kq + e;
            if (co < a.Cout) a.dw[(size_t)co * nW + col] = acc[t][e];
        }
    }
    if (by == 0 && lane < 4 && a.db && co_base + wave * 4 + lane < a.Cout) a.db[co_base + wave * 4 + lane] = dbs;
}
__global__ __launch_bounds__(256) void wgrad_tiny_kernel(const WTArgs a) { wgrad_tiny_body(a, (int)blockIdx.x, (int)blockIdx.y); }

struct DTArgs { const float* g; const float* w; float* slab; int B, Cin, Cout, H, W, s; };   // conv-space: Cin = channels of g, Cout = channels of dx

// NTN: N tiles (16 output channels each) the block computes; wave w owns the M tiles w, w + 4, w + 8, w + 12 (<= 256 pixels)
template <int NTN>
__device__ __forceinline__ void dgrad_tiny_body(const DTArgs& a, const int bx) {                     // bx: the block's slice of 8 gradient channels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, W = a.W, HW = H * W, RS = W + 2, PL = (H + 2) * RS, NW = a.Cout * 9;
    const int MT = (HW + 15) >> 4;
    float* s_g = smem;                                   // [DT_CS][PL] zero border
    float* s_w = s_g + ((DT_CS * PL + 3) & ~3);          // [DT_CS][Cout][9] as stored (w[c][n][tap]); rows beyond Cin zero
    int* s_px = reinterpret_cast<int*>(s_w + ((DT_CS * NW + 3) & ~3));      // [MT * 16]
    int* s_go = s_px + MT * 16;                          // [MT * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int c0 = bx * DT_CS;
    for (int i = tid; i < MT * 16; i += 256) {
        const int y = i / W, x = i - y * W;
        s_px[i] = i < HW ? y * RS + x : 0;
        s_go[i] = i < HW ? (y * a.s) * (W * a.s) + x * a.s : 0;
    }
    for (int i = tid; i < DT_CS * PL; i += 256) s_g[i] = 0.f;
    {
        const int nvalid = (min(a.Cin, c0 + DT_CS) - c0) * NW;          // rows beyond Cin stay zero
        const float* wb = a.w + (size_t)c0 * NW;
        staged<12>(DT_CS * NW, [&](int i) { return i < nvalid ? wb[i] : 0.f; }, [&](int i, float v) { s_w[i] = v; });
    }
    // per-lane K offsets: k = 4 st + kq = (c_local, tap); A reads g[c_local][p + tap], B reads w[c_local][n][8 - tap]
    int ka[DT_KSTEPS], kb[DT_KSTEPS];
#pragma unroll
    for (int st = 0; st < DT_KSTEPS; ++st) {
        const int k = 4 * st + kq, cl = k / 9, tap = k - cl * 9, dy = tap / 3, dx = tap - dy * 3;
        ka[st] = cl * PL + dy * RS + dx;
        kb[st] = cl * NW + (8 - tap);
    }
    int nb[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) nb[n] = min(n * 16 + li, a.Cout - 1) * 9;        // (channels beyond Cout: a valid address, never stored)
    const int gc = tid >> 5, gp0 = tid & 31;                                      // gradient staging: thread -> (local channel, pixels gp0 + 32 j)
    for (int b = 0; b < a.B; ++b) {
        __syncthreads();
        if (c0 + gc < a.Cin) {
            const float* gr = a.g + g_base(b, c0 + gc, a.Cin, H, W, a.s);
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int p = gp0 + 32 * u; v[u] = p < HW ? gr[s_go[p]] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int p = gp0 + 32 * u; if (p < HW) s_g[gc * PL + RS + 1 + s_px[p]] = v[u]; }
        }
        __syncthreads();
        float* out = a.slab + ((size_t)bx * a.B + b) * a.Cout * HW;
        for (int mt = wave; mt < MT; mt += 4) {
            f32x4 acc[NTN];
#pragma unroll
            for (int n = 0; n < NTN; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int po = s_px[mt * 16 + li];
#pragma unroll
            for (int st = 0; st < DT_KSTEPS; ++st) {
                const float av = s_g[ka[st] + po];
#pragma unroll
                for (int n = 0; n < NTN; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s_w[kb[st] + nb[n]], acc[n], 0, 0, 0);
            }
#pragma unroll
            for (int n = 0; n < NTN; ++n) {
                const int co = n * 16 + li;
                if (co >= a.Cout) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int p = mt * 16 + 4 * kq + e;
                    if (p < HW) out[(size_t)co * HW + p] = acc[n][e];
                }
            }
        }
    }
}
template <int NTN>
__global__ __launch_bounds__(256) void dgrad_tiny_kernel(const DTArgs a) { dgrad_tiny_body<NTN>(a, (int)blockIdx.x); }

// The stem stage's backward as ONE launch: the weight gradient's blocks first, the data gradient's K slices behind them.  The two read the
// same gradient and do not depend on each other; each is a 13 / 8 us launch of 47..141 / 94 blocks on 256 CUs whatever its arithmetic
// (staging latency, one K loop, one store phase), and the data gradient's slab reduction rides on the deferred-reduction queue.
template <int NTN>
__global__ __launch_bounds__(256) void stem_pair_kernel(const WTArgs wa, const DTArgs da, const int wx, const int n_w) {
    const int b = (int)blockIdx.x;
    if (b < n_w) wgrad_tiny_body(wa, b % wx, b / wx);
    else dgrad_tiny_body<NTN>(da, b - n_w);
}

bool stem_switch() {                                      // BNERV_STEM=0: the tiled kernels (A/B switch, read per call)
    const char* e = getenv("BNERV_STEM");
    return !(e && e[0] == '0');
}
bool tiny_s_ok(int s, int C) { return (s == 1 || s == 2 || s == 3 || s == 5) && C % (s * s) == 0; }

}  // namespace

// ---- weight gradient.  1: not this kernel's layer; BNERV_OK: dw / db written (nothing deferred, no workspace used)
int bnerv_stem_wgrad_try(hipStream_t st, const bnerv_wgrad_desc& d) {
    if (!stem_switch() || d.k != 3 || d.in_mode != BNERV_IN_PLAIN || d.g_mode == BNERV_IN_TANHGRAD) return 1;
    if ((size_t)d.H * d.W > TINY_MAX_PX || !tiny_s_ok(d.g_s, d.Cout)) return 1;
    const int PL = (d.H + 2) * (d.W + 2);
    if ((size_t)d.Cin * PL > WT_MAX_XFLOATS || d.Cout < 64) return 1;          // (few output channels: the tiled kernels' split over pixels is as good)
    if ((size_t)d.B * (d.Cin > d.Cout ? d.Cin : d.Cout) * d.H * d.W >= (size_t)1 << 30) return 1;
    const int HW = d.H * d.W, HWp = (HW + 3) & ~3;
    const size_t lds = ((size_t)((d.Cin * PL + 3) & ~3) + 16 * (size_t)row_stride4(HW) + 2 * HWp) * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tiny_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    WTArgs a{d.x, d.g, d.dw, d.db, d.B, d.Cin, d.Cout, d.H, d.W, d.g_s};
    const int nt = cdiv(d.Cin * 9, 16);
    hipLaunchKernelGGL(wgrad_tiny_kernel, dim3(cdiv(d.Cout, 16), cdiv(nt, WT_NTG)), dim3(256), lds, st, a);
    BNERV_LAUNCH_CHECK("wgrad_tiny");
    return BNERV_OK;
}

// ---- data gradient.  The slabs go to d.partial (bnerv_conv_splitk_ws_bytes covers them), reduced into d.out.
static bool stem_dgrad_shape(const bnerv_conv_desc& d) {
    if (d.k != 3 || d.ep_mode != BNERV_EP_PLAIN || d.out_s != 1 || !d.transposed) return false;
    if (!(d.in_mode == BNERV_IN_PLAIN || d.in_mode == BNERV_IN_UNSHUFFLE)) return false;
    const int s = d.in_mode == BNERV_IN_UNSHUFFLE ? d.in_s : 1;
    if ((size_t)d.H * d.W > TINY_MAX_PX || !tiny_s_ok(s, d.Cin)) return false;
    if (d.Cin < 128 || d.Cout > 96) return false;          // a long K against few outputs; otherwise the tiled kernels
    if (!(d.wCo == d.Cin && d.wCi == d.Cout)) return false;
    return (size_t)d.B * d.Cin * d.H * d.W < (size_t)1 << 30;
}
size_t bnerv_stem_dgrad_ws_bytes(const bnerv_conv_desc& d) {
    if (!stem_switch() || !stem_dgrad_shape(d)) return 0;
    return (size_t)cdiv(d.Cin, DT_CS) * d.B * d.Cout * d.H * d.W * sizeof(float);
}
int bnerv_stem_dgrad_try(hipStream_t st, const bnerv_conv_desc& d) {
    if (!stem_switch() || !stem_dgrad_shape(d) || !d.partial) return 1;
    const int PL = (d.H + 2) * (d.W + 2), NW = d.Cout * 9, MT = cdiv(d.H * d.W, 16);
    const size_t lds = ((size_t)((DT_CS * PL + 3) & ~3) + (size_t)((DT_CS * NW + 3) & ~3) + 2 * MT * 16) * sizeof(float);
    DTArgs a{d.x, d.w, d.partial, d.B, d.Cin, d.Cout, d.H, d.W, d.in_mode == BNERV_IN_UNSHUFFLE ? d.in_s : 1};
    const int nblk = cdiv(d.Cin, DT_CS);
    if (d.Cout <= 32) hipLaunchKernelGGL(dgrad_tiny_kernel<2>, dim3(nblk), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(dgrad_tiny_kernel<6>, dim3(nblk), dim3(256), lds, st, a);
    BNERV_LAUNCH_CHECK("dgrad_tiny");
    return bnerv_reduce_slabs(st, d.partial, nblk, d.B * d.Cout * d.H * d.W, d.out);
}

// ---- both at once (bnerv_conv_wgrad_pair, form 0).  1: not this pair; BNERV_OK: dw / db written, dx's slabs in c.partial and *n_slabs > 0
// slabs of c.B * c.Cout * c.H * c.W floats for the caller to reduce into c.out (deferred or at once).
int bnerv_stem_pair_try(hipStream_t st, const bnerv_conv_desc& c, const bnerv_wgrad_desc& d, int* n_slabs) {
    { const char* e = getenv("BNERV_PAIR_STEM"); if (e && e[0] == '0') return 1; }      // A/B switch, read per call
    if (!stem_switch() || !stem_dgrad_shape(c) || !c.partial) return 1;
    if (d.k != 3 || d.in_mode != BNERV_IN_PLAIN || d.g_mode == BNERV_IN_TANHGRAD) return 1;
    if ((size_t)d.H * d.W > TINY_MAX_PX || !tiny_s_ok(d.g_s, d.Cout)) return 1;
    const int PL = (d.H + 2) * (d.W + 2);
    if ((size_t)d.Cin * PL > WT_MAX_XFLOATS || d.Cout < 64) return 1;
    if ((size_t)d.B * (d.Cin > d.Cout ? d.Cin : d.Cout) * d.H * d.W >= (size_t)1 << 30) return 1;
    const int cs = c.in_mode == BNERV_IN_UNSHUFFLE ? c.in_s : 1;
    if (!(c.x == d.g && c.Cin == d.Cout && c.Cout == d.Cin && c.B == d.B && c.H == d.H && c.W == d.W && cs == d.g_s)) return 1;   // one layer, one gradient
    const int HW = d.H * d.W, HWp = (HW + 3) & ~3;
    const size_t lds_w = ((size_t)((d.Cin * PL + 3) & ~3) + 16 * (size_t)row_stride4(HW) + 2 * HWp) * sizeof(float);
    const int NW = c.Cout * 9, MT = cdiv(HW, 16);
    const size_t lds_d = ((size_t)((DT_CS * PL + 3) & ~3) + (size_t)((DT_CS * NW + 3) & ~3) + 2 * MT * 16) * sizeof(float);
    const size_t lds = lds_w > lds_d ? lds_w : lds_d;
    static size_t attr2 = 0, attr6 = 0;
    const bool small = c.Cout <= 32;
    size_t& attr = small ? attr2 : attr6;
    if (lds > attr) {
        if (small) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pair_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pair_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    WTArgs wa{d.x, d.g, d.dw, d.db, d.B, d.Cin, d.Cout, d.H, d.W, d.g_s};
    DTArgs da{c.x, c.w, c.partial, c.B, c.Cin, c.Cout, c.H, c.W, cs};
    const int wx = cdiv(d.Cout, 16), wy = cdiv(cdiv(d.Cin * 9, 16), WT_NTG), n_w = wx * wy, n_d = cdiv(c.Cin, DT_CS);
    if (small) hipLaunchKernelGGL(stem_pair_kernel<2>, dim3(n_w + n_d), dim3(256), lds, st, wa, da, wx, n_w);
    else hipLaunchKernelGGL(stem_pair_kernel<6>, dim3(n_w + n_d), dim3(256), lds, st, wa, da, wx, n_w);
    BNERV_LAUNCH_CHECK("stem_pair");
    *n_slabs = n_d;
    return BNERV_OK;
}

