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

__device__ __forceinline__ int g_index(int b, int co, int y, int x, int Cout, int H, int W, int s) {
    // conv-space channel co at (y, x) of a gradient stored pixel-shuffled by s: [B][Cout / s^2][H s][W s]
    const int ss = s * s, cf = co / ss, r = co - cf * ss, i = r / s, j = r - i * s;
    return (((b * (Cout / ss) + cf) * (H * s)) + y * s + i) * (W * s) + x * s + j;
}
__host__ __device__ inline int row_stride4(int n) {       // >= n, == 4 (mod 32): 16 rows x 4 k-lanes spread over the banks
    int v = (n + 3) & ~3;
    v += ((4 - v) % 32 + 32) % 32;
    return v;
}

struct WTArgs { const float* x; const float* g; float* dw; float* db; int B, Cin, Cout, H, W, s; };

__global__ __launch_bounds__(256) void wgrad_tiny_kernel(const WTArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, W = a.W, HW = H * W, RS = W + 2, PL = (H + 2) * RS;
    const int HWp = (HW + 3) & ~3, HWs = row_stride4(HW);
    float* s_x = smem;                                   // [Cin][PL] zero border
    float* s_g = s_x + ((a.Cin * PL + 3) & ~3);          // [16][HWs]
    int* s_px = reinterpret_cast<int*>(s_g + 16 * HWs);  // [HWp]: pixel -> offset in a padded plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int co_base = blockIdx.x * 16, nW = a.Cin * 9;
    int coloff[WT_NTW];
#pragma unroll
    for (int t = 0; t < WT_NTW; ++t) {
        const int col = (blockIdx.y * WT_NTG + wave + 4 * t) * 16 + li;
        const int ci = col / 9, tap = col - ci * 9, dy = tap / 3, dx = tap - dy * 3;
        coloff[t] = col < nW ? ci * PL + dy * RS + dx : 0;                       // (columns beyond the matrix: any valid address, never stored)
    }
    for (int i = tid; i < HWp; i += 256) s_px[i] = i < HW ? (i / W) * RS + (i % W) : 0;
    f32x4 acc[WT_NTW];
#pragma unroll
    for (int t = 0; t < WT_NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs = 0.f;
    for (int b = 0; b < a.B; ++b) {
        if (b) __syncthreads();
        for (int i = tid; i < a.Cin * PL; i += 256) {
            const int c = i / PL, r = i - c * PL, yy = r / RS - 1, xx = r % RS - 1;
            s_x[i] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? a.x[((size_t)(b * a.Cin + c) * H + yy) * W + xx] : 0.f;
        }
        for (int i = tid; i < 16 * HWs; i += 256) {
            const int m = i / HWs, p = i - m * HWs, co = co_base + m;
            s_g[i] = (p < HW && co < a.Cout) ? a.g[g_index(b, co, p / W, p % W, a.Cout, H, W, a.s)] : 0.f;
        }
        __syncthreads();
        for (int st = 0; st < HWp / 4; ++st) {
            const float av = s_g[li * HWs + 4 * st + kq];
            const int po = s_px[4 * st + kq];
#pragma unroll
            for (int t = 0; t < WT_NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s_x[coloff[t] + po], acc[t], 0, 0, 0);
        }
        if (blockIdx.y == 0 && tid < 16) {                   // bias gradient: the row sums of g, pixels in order
            float s = 0.f;
            for (int p = 0; p < HW; ++p) s += s_g[tid * HWs + p];
            dbs += s;
        }
    }
#pragma unroll
    for (int t = 0; t < WT_NTW; ++t) {
        const int col = (blockIdx.y * WT_NTG + wave + 4 * t) * 16 + li;
        if (col >= nW) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co_base + 4 * kq + e;
            if (co < a.Cout) a.dw[(size_t)co * nW + col] = acc[t][e];
        }
    }
    if (blockIdx.y == 0 && tid < 16 && a.db && co_base + tid < a.Cout) a.db[co_base + tid] = dbs;
}

struct DTArgs { const float* g; const float* w; float* slab; int B, Cin, Cout, H, W, s; };   // conv-space: Cin = channels of g, Cout = channels of dx

// NTN: N tiles (16 output channels each) the block computes; wave w owns the M tiles w, w + 4, w + 8, w + 12 (<= 256 pixels)
template <int NTN>
__global__ __launch_bounds__(256) void dgrad_tiny_kernel(const DTArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, W = a.W, HW = H * W, RS = W + 2, PL = (H + 2) * RS, NW = a.Cout * 9;
    const int MT = (HW + 15) >> 4;
    float* s_g = smem;                                   // [DT_CS][PL] zero border
    float* s_w = s_g + ((DT_CS * PL + 3) & ~3);          // [DT_CS][Cout][9] as stored (w[c][n][tap]); rows beyond Cin zero
    int* s_px = reinterpret_cast<int*>(s_w + ((DT_CS * NW + 3) & ~3));      // [MT * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int c0 = blockIdx.x * DT_CS;
    for (int i = tid; i < MT * 16; i += 256) s_px[i] = i < HW ? (i / W) * RS + (i % W) : 0;
    for (int i = tid; i < DT_CS * NW; i += 256) {
        const int c = c0 + i / NW;
        s_w[i] = c < a.Cin ? a.w[(size_t)c0 * NW + i] : 0.f;
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
    for (int b = 0; b < a.B; ++b) {
        __syncthreads();
        for (int i = tid; i < DT_CS * PL; i += 256) {
            const int cl = i / PL, r = i - cl * PL, yy = r / RS - 1, xx = r % RS - 1, c = c0 + cl;
            s_g[i] = (c < a.Cin && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? a.g[g_index(b, c, yy, xx, a.Cin, H, W, a.s)] : 0.f;
        }
        __syncthreads();
        float* out = a.slab + ((size_t)blockIdx.x * a.B + b) * a.Cout * HW;
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
    const size_t lds = ((size_t)((d.Cin * PL + 3) & ~3) + 16 * (size_t)row_stride4(HW) + HWp) * sizeof(float);
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
    const size_t lds = ((size_t)((DT_CS * PL + 3) & ~3) + (size_t)((DT_CS * NW + 3) & ~3) + MT * 16) * sizeof(float);
    DTArgs a{d.x, d.w, d.partial, d.B, d.Cin, d.Cout, d.H, d.W, d.in_mode == BNERV_IN_UNSHUFFLE ? d.in_s : 1};
    const int nblk = cdiv(d.Cin, DT_CS);
    if (d.Cout <= 32) hipLaunchKernelGGL(dgrad_tiny_kernel<2>, dim3(nblk), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(dgrad_tiny_kernel<6>, dim3(nblk), dim3(256), lds, st, a);
    BNERV_LAUNCH_CHECK("dgrad_tiny");
    return bnerv_reduce_slabs(st, d.partial, nblk, d.B * d.Cout * d.H * d.W, d.out);
}
