// loss.hip -- the high-frequency reconstruction loss of Boosting-NeRV, value AND gradient in one call.
//
// Replaces hnerv_utils.loss_fn (hnerv_utils.py:335-397) + its autograd backward, psnr_fn_single (:400-403) and the
// MS-SSIM metric (:410-412):
//     loss_b = c_l1*mean|d| + c_l2*mean d^2 + c_ms*(1 - ms_ssim_b) + c_fft*mean(|Re F(d)| + |Im F(d)|)/1 ,  d = pred - target
// (the reference takes FFT2(pred) - FFT2(target); the DFT is linear, so F(d) is the same quantity with ONE transform).
//
// Kernels (all HBM-streaming; no MFMA -- none of this is a GEMM):
//   diff_stats      sum|d|, sum d^2 per sample                                     (L1, L2, PSNR)
//   avgpool2        the MS-SSIM pyramid (2x2 mean, padding = size%2), pred and target together
//   ssim_fwd        per level: 11-tap separable Gaussian statistics in LDS -> per-tile sums of cs (levels 0-3) / ssim (4)
//   ms_coef         ms_ssim per (b,c) and d(loss)/d(level statistic)
//   ssim_bwd        per level, coarse -> fine: recompute the statistics for a haloed tile, differentiate, apply the adjoint
//                   Gaussian, add the upsampled coarser-level gradient; level 0 also adds the L1/L2 terms and writes `grad`
//   fft_rows_fwd    in-place mixed-radix DIF FFT of every row in LDS (digit-reversed output order -- irrelevant, see below)
//   fft_cols        column DIF FFT, sum(|Re|+|Im|), S = sign(F), then the ADJOINT transform of S, all in LDS
//   fft_rows_adj    adjoint row transform, real part, accumulated into `grad`
// The forward FFT leaves its output digit-reversed; the L1 norm does not care about order and the adjoint network is the
// exact transpose-conjugate of the forward network, so no reordering pass exists anywhere.
//
// MS-SSIM follows pytorch_msssim 0.2.1 (third-party; PARITY UNPINNED).
#include "common.h"
#include <vector>
#include <math.h>
#include <string.h>
#include <map>
#include <mutex>

namespace {

// =====================================================================================================================
// L1 / L2 statistics
// =====================================================================================================================
constexpr int NSB = 512;  // partial blocks per sample

__device__ __forceinline__ void diff_stats_body(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ part, int n_per_sample, const int bx, const int b) {
    const float* pp = p + (size_t)b * n_per_sample;
    const float* tt = t + (size_t)b * n_per_sample;
    float s1 = 0.f, s2 = 0.f;
    // eight loads in flight per thread; the sums take the elements in the same order as one by one
    constexpr int STR = NSB * 256;
    int i = bx * 256 + threadIdx.x;
    for (; i + 3 * STR < n_per_sample; i += 4 * STR) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = pp[i + u * STR] - tt[i + u * STR];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += fabsf(d[u]); s2 = fmaf(d[u], d[u], s2); }
    }
    for (; i < n_per_sample; i += STR) {
        const float d = pp[i] - tt[i];
        s1 += fabsf(d);
        s2 = fmaf(d, d, s2);
    }
    __shared__ float red[2][4];
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) {
        part[((size_t)b * NSB + bx) * 2 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    }
}
__global__ __launch_bounds__(256) void diff_stats_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ part, int n_per_sample) {
    diff_stats_body(p, t, part, n_per_sample, blockIdx.x, blockIdx.y);
}

// one wave per sample: lane-strided partial sums, then a wave reduction in fp64
__global__ __launch_bounds__(64) void psnr_final_kernel(const float* __restrict__ part, float* __restrict__ psnr, int B, int n_per_sample) {
    const int b = blockIdx.x;
    double s2 = 0.0;
    for (int k = threadIdx.x; k < NSB; k += 64) s2 += (double)part[((size_t)b * NSB + k) * 2 + 1];
    s2 = wave_sum_d(s2);
    if (threadIdx.x == 0) {
        const float mse = (float)(s2 / (double)n_per_sample);
        psnr[b] = -10.0f * log10f(mse + 1e-9f);
    }
}

// grad = k1*sign(d) + k2*d     (losses without the MS-SSIM term)
__global__ void grad_l1l2_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ g, size_t n, float k1, float k2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = p[i] - t[i];
        const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        g[i] = k1 * sg + k2 * d;
    }
}

// =====================================================================================================================
// MS-SSIM
// =====================================================================================================================
constexpr int LV = BNERV_MSSSIM_LEVELS;
constexpr int WS_ = 11, HW_ = 10;           // window size, window size - 1
constexpr int STH = 16, STW = 32;           // ssim tile
__device__ __forceinline__ int cdiv_d(int a, int b) { return (a + b - 1) / b; }

struct Win { float g[WS_]; };

struct Pyr { int H[LV], W[LV]; };

static Pyr make_pyr(int H, int W) {
    Pyr p;
    p.H[0] = H; p.W[0] = W;
    for (int l = 1; l < LV; ++l) {
        const int ph = p.H[l - 1] % 2, pw = p.W[l - 1] % 2;
        p.H[l] = (p.H[l - 1] + 2 * ph - 2) / 2 + 1;
        p.W[l] = (p.W[l - 1] + 2 * pw - 2) / 2 + 1;
    }
    return p;
}

static Win make_win() {
    // exactly pytorch_msssim._fspecial_gauss_1d in fp32: g = exp(-(x-5)^2 / (2*1.5^2)); g /= g.sum()
    Win w;
    float s = 0.f;
    for (int i = 0; i < WS_; ++i) {
        const float c = (float)(i - WS_ / 2);
        w.g[i] = expf(-(c * c) / (2.0f * 1.5f * 1.5f));
        s += w.g[i];
    }
    for (int i = 0; i < WS_; ++i) w.g[i] /= s;
    return w;
}

// 2x2 mean pool with zero padding (ph, pw), count_include_pad=True; two images per launch (blockIdx.z selects)
__global__ void avgpool2_kernel(const float* __restrict__ x0, const float* __restrict__ x1, float* __restrict__ y0, float* __restrict__ y1,
                                int planes, int H, int W, int Ho, int Wo, int ph, int pw) {
    const float* x = blockIdx.z ? x1 : x0;
    float* y = blockIdx.z ? y1 : y0;
    const size_t n = (size_t)planes * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const size_t r = i / Wo;
        const int oy = (int)(r % Ho);
        const size_t pl = r / Ho;
        const int iy = 2 * oy - ph, ix = 2 * ox - pw;
        const float* xp = x + pl * H * W;
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = iy + dy, xx = ix + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += xp[(size_t)yy * W + xx];
            }
        y[i] = 0.25f * s;
    }
}

// All four coarser levels of BOTH images in one launch, for frames whose sides stay even down to level 3 (720x1280: 360, 180, 90 |
// 640, 320, 160): a block owns a 32x32 patch of level 0 = 16x16 of level 1 = ... = 2x2 of level 4, the intermediate levels pass
// through LDS.  Same arithmetic as avgpool2_kernel level by level (0.25 * (((a + b) + c) + d)), so the pyramid is bit-identical.
struct PyrArgs { const float* src[2]; float* dst[2][LV]; int H[LV], W[LV]; int planes; };
__device__ __forceinline__ void pyramid_body(const PyrArgs& a, const int bx, const int by, const int bz) {
    __shared__ float s1[16][17], s2[8][9], s3[4][5];
    const int img = bz & 1, pl = bz >> 1;
    const float* x = a.src[img] + (size_t)pl * a.H[0] * a.W[0];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    {
        const int oy = by * 16 + ty, ox = bx * 16 + tx;
        float v = 0.f;
        if (oy < a.H[1] && ox < a.W[1]) {
            const float2 r0 = *reinterpret_cast<const float2*>(x + (size_t)(2 * oy) * a.W[0] + 2 * ox);
            const float2 r1 = *reinterpret_cast<const float2*>(x + (size_t)(2 * oy + 1) * a.W[0] + 2 * ox);
            v = 0.25f * (((r0.x + r0.y) + r1.x) + r1.y);
            a.dst[img][1][((size_t)pl * a.H[1] + oy) * a.W[1] + ox] = v;
        }
        s1[ty][tx] = v;
    }
    __syncthreads();
    if (tid < 64) {
        const int y = tid >> 3, xx = tid & 7, oy = by * 8 + y, ox = bx * 8 + xx;
        const float v = 0.25f * (((s1[2 * y][2 * xx] + s1[2 * y][2 * xx + 1]) + s1[2 * y + 1][2 * xx]) + s1[2 * y + 1][2 * xx + 1]);
        if (oy < a.H[2] && ox < a.W[2]) a.dst[img][2][((size_t)pl * a.H[2] + oy) * a.W[2] + ox] = v;
        s2[y][xx] = v;
    }
    __syncthreads();
    if (tid < 16) {
        const int y = tid >> 2, xx = tid & 3, oy = by * 4 + y, ox = bx * 4 + xx;
        const float v = 0.25f * (((s2[2 * y][2 * xx] + s2[2 * y][2 * xx + 1]) + s2[2 * y + 1][2 * xx]) + s2[2 * y + 1][2 * xx + 1]);
        if (oy < a.H[3] && ox < a.W[3]) a.dst[img][3][((size_t)pl * a.H[3] + oy) * a.W[3] + ox] = v;
        s3[y][xx] = v;
    }
    __syncthreads();
    if (tid < 4) {
        const int y = tid >> 1, xx = tid & 1, oy = by * 2 + y, ox = bx * 2 + xx;
        const float v = 0.25f * (((s3[2 * y][2 * xx] + s3[2 * y][2 * xx + 1]) + s3[2 * y + 1][2 * xx]) + s3[2 * y + 1][2 * xx + 1]);
        if (oy < a.H[4] && ox < a.W[4]) a.dst[img][4][((size_t)pl * a.H[4] + oy) * a.W[4] + ox] = v;
    }
}
__global__ __launch_bounds__(256) void pyramid_kernel(const PyrArgs a) { pyramid_body(a, blockIdx.x, blockIdx.y, blockIdx.z); }

struct SsimArgs {
    const float* X; const float* Y;
    float* partial;            // fwd: [BC][tiles]
    const float* coef;         // bwd: [BC] for this level (already includes 1/Nvalid and the loss chain)
    const float* dcoarse;      // bwd: [BC][Hc][Wc] gradient wrt the next (coarser) level's pooled image, or NULL
    float* dX;                 // bwd: [BC][H][W] written
    float* G;                  // fwd writes / bwd reads: [3][BC][H][W] UNSCALED statistic gradients (d mu1-ish, d E[xx], d E[xy]) at
                               // the valid window positions; NULL = value only.  The chain coefficient (known only after every
                               // level's forward) multiplies the backward linearly, so it is applied after the adjoint filter.
    int H, W, Hc, Wc, ph, pw;  // ph/pw: padding used when pooling THIS level into the coarser one
    int tiles_x, tiles_y;
    float C1, C2;
    float k_l1, k_l2;          // level 0 only: extra terms k_l1*sign(d) + k_l2*d
    int acc;                   // level 0 only: dX += (1) instead of dX = (0) -- the spectral gradient is already there (see loss_coarse_kernel)
    Win win;
};

// ---- forward: per-tile sum of cs (LAST=false) or ssim (LAST=true) over the valid region ----
// The filter passes run on PACKED pairs of maps (v_pk_fma_f32: two fused multiply-adds per issued instruction; these kernels are bound
// by instruction issue): (x, y) live interleaved in LDS, so a tap is one 8-byte read and pk_fma(g, (x, y)), pk_fma(g, (x x, y y)),
// fma(g, x y) -- five instructions for the eight of the scalar form; the second pass reads (v0, v1), (v2, v3), v4 likewise.  Every
// lane of a packed operation is the same IEEE fma in the same tap order as before: the numbers keep their bits.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void ssim_fwd_body(const SsimArgs& a, const bool LAST, const int bx, const int by, const int bc, const int nbc) {
#pragma clang fp contract(off)                           // the SSIM algebra as written (products rounded, then added -- as the reference's tensor ops do), identical in every instantiation; the filter taps are explicit fmas
    constexpr int WH = STH + HW_, WW = STW + HW_;        // 26 x 42 input window
    __shared__ pk2 sXY[WH][WW];
    __shared__ pk2 sV01[STH][WW], sV23[STH][WW];
    __shared__ float sV4[STH][WW];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int oy0 = by * STH, ox0 = bx * STW;
    const float* X = a.X + (size_t)bc * a.H * a.W;
    const float* Y = a.Y + (size_t)bc * a.H * a.W;
    {   // the whole window in flight (10 loads per thread), then the LDS stores: one memory round trip per block instead of five
        constexpr int NLD = (WH * WW + 255) / 256;
        pk2 ld[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + u * 256;
            const int r = i / WW, c = i - r * WW;
            const int y = oy0 + r, x = ox0 + c;
            const bool in = i < WH * WW && y < a.H && x < a.W;
            ld[u] = pk2{in ? X[(size_t)y * a.W + x] : 0.f, in ? Y[(size_t)y * a.W + x] : 0.f};
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + u * 256;
            if (i < WH * WW) (&sXY[0][0])[i] = ld[u];
        }
    }
    __syncthreads();
    // vertical (along H) first, as the reference filters dim 2 then dim 3
    for (int i = tid; i < STH * WW; i += 256) {
        const int r = i / WW, c = i - r * WW;
        pk2 v01 = {0.f, 0.f}, v23 = {0.f, 0.f};
        float v4 = 0.f;
#pragma unroll
        for (int k = 0; k < WS_; ++k) {
            const float g = a.win.g[k];
            const pk2 gg = {g, g}, xy = sXY[r + k][c];
            v01 = pk_fma(gg, xy, v01); v23 = pk_fma(gg, xy * xy, v23); v4 = fmaf(g, xy.x * xy.y, v4);
        }
        sV01[r][c] = v01; sV23[r][c] = v23; sV4[r][c] = v4;
    }
    __syncthreads();
    float acc = 0.f;
    const int Hv = a.H - HW_, Wv = a.W - HW_;
    for (int i = tid; i < STH * STW; i += 256) {
        const int r = i / STW, c = i - r * STW;
        pk2 m = {0.f, 0.f}, e = {0.f, 0.f};
        float exy = 0.f;
#pragma unroll
        for (int k = 0; k < WS_; ++k) {
            const float g = a.win.g[k];
            const pk2 gg = {g, g};
            m = pk_fma(gg, sV01[r][c + k], m); e = pk_fma(gg, sV23[r][c + k], e); exy = fmaf(g, sV4[r][c + k], exy);
        }
        const float m1 = m.x, m2 = m.y, exx = e.x, eyy = e.y;
        if (oy0 + r < Hv && ox0 + c < Wv) {
            const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            const float s1 = exx - m11, s2 = eyy - m22, s12 = exy - m12;
            const float B2 = s1 + s2 + a.C2;
            const float cs = (2.f * s12 + a.C2) / B2;
            float lum = 1.f;
            if (LAST) { lum = (2.f * m12 + a.C1) / (m11 + m22 + a.C1); acc += lum * cs; }
            else acc += cs;
            if (a.G) {
                // the statistic gradients take 1 / B2 (and 1 / B1) through v_rcp_f32 (1 ulp) -- three IEEE divisions were a quarter of this
                // pass's instructions; the VALUE path (cs, lum) keeps the exact quotient
                const float dxy0 = 2.f * __builtin_amdgcn_rcpf(B2);
                float dm = dxy0 * (m1 * cs - m2), dxx = -0.5f * cs * dxy0, dxy = dxy0;
                if (LAST) {
                    const float B1 = m11 + m22 + a.C1;
                    dm = (2.f * __builtin_amdgcn_rcpf(B1)) * (m2 - m1 * lum) * cs + lum * dm;
                    dxx *= lum; dxy *= lum;
                }
                const size_t plane = (size_t)a.H * a.W, o = (size_t)bc * plane + (size_t)(oy0 + r) * a.W + (ox0 + c);
                const size_t mstride = (size_t)nbc * plane;
                a.G[o] = dm; a.G[mstride + o] = dxx; a.G[2 * mstride + o] = dxy;
            }
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) a.partial[(size_t)bc * (a.tiles_x * a.tiles_y) + by * a.tiles_x + bx] = red[0] + red[1] + red[2] + red[3];
}
// XCD-contiguous tile order of a (tiles_x, tiles_y, planes) grid: the dispatcher deals linear block ids round-robin over the 8 XCDs;
// with this remap each XCD works on a contiguous run of tiles, whose shared window rows / columns it finds in its own L2
struct Tile3 { int x, y, z; };
__device__ __forceinline__ Tile3 xcd_tile3() {
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    const int lb = xcd_remap((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)(gridDim.x * gridDim.y * gridDim.z));
    const int z = lb / (gx * gy), r = lb - z * (gx * gy);
    return Tile3{r % gx, r / gx, z};
}
template <bool LAST>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const SsimArgs a) { const Tile3 t = xcd_tile3(); ssim_fwd_body(a, LAST, t.x, t.y, t.z, gridDim.z); }

// every level's statistics in ONE launch (the levels only depend on the pyramid): block -> (level, tile) through the running tile counts
struct SsimAllArgs { SsimArgs lv[LV]; int first[LV + 1]; };
__global__ __launch_bounds__(256) void ssim_fwd_all_kernel(const SsimAllArgs a) {
    // neighbouring tiles share 10 of their 26 x 42 window rows / columns: each XCD (linear block ids b, b + 8, ...) takes a contiguous run of tiles
    const int lb = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int gbx = lb % (int)gridDim.x, gby = lb / (int)gridDim.x;
    int l = 0;
#pragma unroll
    for (int k = 1; k < LV; ++k) if (gbx >= a.first[k]) l = k;
    const int t = gbx - a.first[l];
    const int bx = t % a.lv[l].tiles_x, by = t / a.lv[l].tiles_x;
    ssim_fwd_body(a.lv[l], l == LV - 1, bx, by, gby, gridDim.y);      // (one body: its LDS arrays exist once)
}

// ---- ms_ssim per (b,c) and the chain coefficients; one block (5 waves) per (b,c) ----
struct CoefArgs {
    const float* partial[LV];   // [BC][tiles_l]
    int tiles[LV];
    float inv_nvalid[LV];
    float weights[LV];
    float* msval;               // [BC]
    float* coef;                // [LV][BC]
    int BC;
    float chain;                // -c_ms / (B*C)
};
// one wave per level (a block of 5 waves) or, inside a 256-thread launch, waves 0..3 with wave 0 taking level 4 as well: the per-level
// sums never mix, so both forms add the same numbers in the same order
__device__ __forceinline__ void ms_coef_body(const CoefArgs& a, const int bc) {
    const int lane = threadIdx.x & 63, nw = (int)blockDim.x >> 6;
    __shared__ float stat[LV];
    for (int l = threadIdx.x >> 6; l < LV; l += nw) {
        double s = 0.0;
        // 8 loads in flight per lane (a level-0 row of 1800 partials was 29 serialised L2 round trips: 12 us for 3 blocks), added in
        // the same order as one by one
        const float* part = a.partial[l] + (size_t)bc * a.tiles[l];
        const int nt = a.tiles[l];
        for (int i0 = lane; i0 < nt; i0 += 8 * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64; v[u] = i < nt ? part[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * 64 < nt) s += (double)v[u];
        }
        s = wave_sum_d(s);
        if (lane == 0) stat[l] = (float)(s * (double)a.inv_nvalid[l]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float P = 1.f;
        for (int k = 0; k < LV; ++k) P *= powf(fmaxf(stat[k], 0.f), a.weights[k]);
        a.msval[bc] = P;
        for (int k = 0; k < LV; ++k)
            a.coef[(size_t)k * a.BC + bc] = stat[k] > 0.f ? a.chain * a.weights[k] * (P / stat[k]) * a.inv_nvalid[k] : 0.f;
    }
}
__global__ __launch_bounds__(320) void ms_coef_kernel(const CoefArgs a) { ms_coef_body(a, blockIdx.x); }

// ---- backward from the stored statistic gradients: adjoint of the separable "valid" filter over the 3 maps, then
//      dX = coef * (A0 + 2 x A1 + y A2) [+ 0.25 * d(coarser level)] [+ L1/L2 terms at level 0].  ~6x less arithmetic than
//      recomputing the statistics on a 36x52 window per tile.
// level-0 form: the coarser levels' OWN terms (n of them), combined here; ph / pw[k]: the zero padding level k was pooled into level k + 1 with
// (0 on an even pyramid; 1080 -> 540 -> 270 -> 135 -> 68 pads 135), so pixel (y, x) of level k lies in cell ((y + ph) / 2, (x + pw) / 2)
struct CoarseChain { const float* own[LV]; int H[LV], W[LV], ph[LV], pw[LV]; int n; };
constexpr int SSIM_BWD_LDS = 3 * (2 * STH + HW_) * (STW + HW_);     // floats: a packed pair and a single map, each (26 + 16) x 42
// `lds`: SSIM_BWD_LDS floats of the caller's LDS (8-byte aligned) -- a pointer so that a kernel whose blocks run EITHER this body or an FFT
// body (loss_coarse_kernel) can give both the same allocation: two static arrays would add up and halve the blocks per CU
template <bool LEVEL0>
__device__ __forceinline__ void ssim_bwd_body(const SsimArgs& a, const CoarseChain* cc, float* lds, const int bx, const int by, const int bc, const int nbc) {
    constexpr int GH = STH + HW_, GW = STW + HW_;        // 26 x 42 statistic-gradient region
    pk2 (*sG01)[GW] = reinterpret_cast<pk2 (*)[GW]>(lds);                                   // packed pairs of maps, as in the forward pass
    pk2 (*sA01)[GW] = reinterpret_cast<pk2 (*)[GW]>(lds + 2 * GH * GW);
    float (*sG2)[GW] = reinterpret_cast<float (*)[GW]>(lds + 2 * GH * GW + 2 * STH * GW);
    float (*sA2)[GW] = reinterpret_cast<float (*)[GW]>(lds + 3 * GH * GW + 2 * STH * GW);
    const int tid = threadIdx.x;
    const int py0 = by * STH, px0 = bx * STW;
    const int wy0 = py0 - HW_, wx0 = px0 - HW_;
    const int Hv = a.H - HW_, Wv = a.W - HW_;
    const size_t plane = (size_t)a.H * a.W, mstride = (size_t)nbc * plane;
    const float* Gp = a.G + (size_t)bc * plane;
    {   // the whole 26 x 42 region of the three maps in flight (15 loads per thread), then the LDS stores
        constexpr int NLD = (GH * GW + 255) / 256;
        float l0[NLD], l1[NLD], l2[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + u * 256;
            const int r = i / GW, c = i - r * GW;
            const int oy = wy0 + r, ox = wx0 + c;
            l0[u] = 0.f; l1[u] = 0.f; l2[u] = 0.f;
            if (i < GH * GW && oy >= 0 && oy < Hv && ox >= 0 && ox < Wv) {
                const size_t o = (size_t)oy * a.W + ox;
                l0[u] = Gp[o]; l1[u] = Gp[mstride + o]; l2[u] = Gp[2 * mstride + o];
            }
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + u * 256;
            if (i < GH * GW) { (&sG01[0][0])[i] = pk2{l0[u], l1[u]}; (&sG2[0][0])[i] = l2[u]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < STH * GW; i += 256) {
        const int r = i / GW, c = i - r * GW;
        pk2 a01 = {0.f, 0.f};
        float a2 = 0.f;
#pragma unroll
        for (int k = 0; k < WS_; ++k) {
            const float g = a.win.g[k];
            a01 = pk_fma(pk2{g, g}, sG01[r + HW_ - k][c], a01);
            a2 = fmaf(g, sG2[r + HW_ - k][c], a2);
        }
        sA01[r][c] = a01; sA2[r][c] = a2;
    }
    __syncthreads();
    const float coef = a.coef[bc];
    const float* X = a.X + (size_t)bc * plane;
    const float* Y = a.Y + (size_t)bc * plane;
    for (int i = tid; i < STH * STW; i += 256) {
        const int r = i / STW, c = i - r * STW;
        const int y = py0 + r, x = px0 + c;
        if (y >= a.H || x >= a.W) continue;
        pk2 a01 = {0.f, 0.f};
        float a2 = 0.f;
#pragma unroll
        for (int k = 0; k < WS_; ++k) {
            const float g = a.win.g[k];
            a01 = pk_fma(pk2{g, g}, sA01[r][c + HW_ - k], a01);
            a2 = fmaf(g, sA2[r][c + HW_ - k], a2);
        }
        const float a0 = a01.x, a1 = a01.y;
        const float xv = X[(size_t)y * a.W + x], yv = Y[(size_t)y * a.W + x];
        float d = coef * (a0 + 2.f * xv * a1 + yv * a2);
        if (a.dcoarse) d += 0.25f * a.dcoarse[((size_t)bc * a.Hc + (y + a.ph) / 2) * a.Wc + (x + a.pw) / 2];
        if (LEVEL0 && cc != nullptr && cc->n > 0) {
            // even pyramid: the coarser levels stored only their OWN terms; d_k = own_k + 0.25 d_{k+1} is evaluated here, innermost first,
            // exactly as the level-by-level launches did (0.25 * is exact), so the gradient keeps its bits
            float dc = 0.f;
            int yk[LV], xk[LV];
            yk[0] = y; xk[0] = x;
#pragma unroll
            for (int k = 1; k < LV; ++k) { yk[k] = (yk[k - 1] + cc->ph[k - 1]) >> 1; xk[k] = (xk[k - 1] + cc->pw[k - 1]) >> 1; }
#pragma unroll
            for (int k = LV - 1; k >= 1; --k) {
                if (k > cc->n) continue;
                const float o = cc->own[k][((size_t)bc * cc->H[k] + yk[k]) * cc->W[k] + xk[k]];
                dc = (k == cc->n) ? o : o + 0.25f * dc;
            }
            d += 0.25f * dc;
        }
        if (LEVEL0) {
            const float df = xv - yv;
            const float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
            d += a.k_l1 * sg + a.k_l2 * df;
        }
        float* dst = a.dX + ((size_t)bc * a.H + y) * a.W + x;
        *dst = (LEVEL0 && a.acc) ? d + *dst : d;
    }
}
template <bool LEVEL0>
__global__ __launch_bounds__(256) void ssim_bwd_from_g_kernel(const SsimArgs a) {
    __shared__ __attribute__((aligned(16))) float sl[SSIM_BWD_LDS];
    const Tile3 t = xcd_tile3(); ssim_bwd_body<LEVEL0>(a, nullptr, sl, t.x, t.y, t.z, gridDim.z);
}
__global__ __launch_bounds__(256) void ssim_bwd_level0_chain_kernel(const SsimArgs a, const CoarseChain cc) {
    __shared__ __attribute__((aligned(16))) float sl[SSIM_BWD_LDS];
    const Tile3 t = xcd_tile3(); ssim_bwd_body<true>(a, &cc, sl, t.x, t.y, t.z, gridDim.z);
}
// levels 1 .. LV-1 in one launch, each writing only its OWN term (no coarser contribution: the level-0 launch combines them)
__device__ __forceinline__ void ssim_bwd_coarse_body(const SsimAllArgs& a, float* lds, const int lin, const int gx, const int nbc) {
    const int lb = xcd_remap(lin, gx * nbc);
    const int gbx = lb % gx, gby = lb / gx;
    int l = 1;
#pragma unroll
    for (int k = 2; k < LV; ++k) if (gbx >= a.first[k]) l = k;
    const int t = gbx - a.first[l];
    const int tx = cdiv_d(a.lv[l].W, STW);
    ssim_bwd_body<false>(a.lv[l], nullptr, lds, t % tx, t / tx, gby, nbc);
}
__global__ __launch_bounds__(256) void ssim_bwd_coarse_all_kernel(const SsimAllArgs a) {
    __shared__ __attribute__((aligned(16))) float sl[SSIM_BWD_LDS];
    ssim_bwd_coarse_body(a, sl, (int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, (int)gridDim.y);
}

// =====================================================================================================================
// mixed-radix FFT in LDS
// =====================================================================================================================
constexpr int MAXRAD = 16;
struct FftPlan { int N, nrad; int rad[MAXRAD]; const float2* tw; const int* pos; };   // tw[k] = exp(-2*pi*i*k/N)
// pos[f] = buffer position that holds frequency f after the in-place DIF forward (mixed-radix digit reversal)

std::mutex g_tw_mutex;
std::map<int, float2*> g_tw;

static const float2* get_twiddles(int N) {
    std::lock_guard<std::mutex> lk(g_tw_mutex);
    auto it = g_tw.find(N);
    if (it != g_tw.end()) return it->second;
    float2* h = (float2*)malloc(sizeof(float2) * N);
    for (int k = 0; k < N; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)N;
        h[k].x = (float)cos(ang);
        h[k].y = (float)sin(ang);
    }
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * N) != hipSuccess) { free(h); return nullptr; }
    if (hipMemcpy(d, h, sizeof(float2) * N, hipMemcpyHostToDevice) != hipSuccess) { free(h); return nullptr; }
    free(h);
    g_tw[N] = d;
    return d;
}

std::map<int, int*> g_pos;
// forward DIF with radices [R, rest] on size Ns: frequency k = q + R*k' ends up in sub-block q (size M = Ns/R) at the position
// the rest of the plan gives k'  =>  pos(k) = (k % R) * M + pos_rest(k / R)
static const int* get_positions(const FftPlan& pl) {
    std::lock_guard<std::mutex> lk(g_tw_mutex);
    auto it = g_pos.find(pl.N);
    if (it != g_pos.end()) return it->second;
    std::vector<int> h(pl.N);
    for (int f = 0; f < pl.N; ++f) {
        int k = f, Ns = pl.N, p = 0;
        for (int st = 0; st < pl.nrad; ++st) {
            const int R = pl.rad[st], M = Ns / R;
            p += (k % R) * M;
            k /= R;
            Ns = M;
        }
        h[f] = p;
    }
    int* d = nullptr;
    if (hipMalloc(&d, sizeof(int) * pl.N) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(int) * pl.N, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_pos[pl.N] = d;
    return d;
}

static bool make_plan(int N, FftPlan* p) {
    p->N = N; p->nrad = 0;
    int n = N;
    auto push = [&](int r) { if (p->nrad < MAXRAD) p->rad[p->nrad++] = r; };
    while (n % 4 == 0) { push(4); n /= 4; }
    while (n % 2 == 0) { push(2); n /= 2; }
    while (n % 3 == 0) { push(3); n /= 3; }
    while (n % 5 == 0) { push(5); n /= 5; }
    for (int r = 7; r <= BNERV_FFT_MAX_RADIX && n > 1; r += 2)
        while (n % r == 0) { push(r); n /= r; }
    if (n != 1 || p->nrad >= MAXRAD) return false;
    if (N == 1) { p->nrad = 0; }
    p->tw = get_twiddles(N);
    p->pos = get_positions(*p);
    return p->tw != nullptr && p->pos != nullptr;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return float2{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a * conj(b)

// One butterfly of radix R at sub-transform size Ns (M = Ns/R) on the in-place buffer.
//  forward (DIF):  y_q = w_Ns^{jq} * sum_m x_m w_R^{mq}             x_m = buf[base+m*M], y_q -> buf[base+q*M]
//  adjoint      :  x_m = sum_q conj(w_R^{mq}) conj(w_Ns^{jq}) y_q   (exact conjugate transpose of the forward stage)
// Radix 2/3/4/5 cores use the closed forms (adds, +-i swaps, two or four real constants): the kernels are instruction-bound, and
// a table-driven core costs ~300 instructions per radix-4 butterfly against ~50 here.  SGN = -1 forward, +1 adjoint: the
// adjoint core is the forward core with i -> -i, i.e. the exact conjugate transpose with the same constants.
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
template <int SGN> __device__ __forceinline__ float2 mul_i(float2 a) { return SGN > 0 ? float2{-a.y, a.x} : float2{a.y, -a.x}; }   // (SGN*i) * a

template <int R, int SGN>
__device__ __forceinline__ void dft_core(const float2 (&v)[R], float2 (&o)[R]) {
    if constexpr (R == 2) {
        o[0] = cadd(v[0], v[1]);
        o[1] = csub(v[0], v[1]);
    } else if constexpr (R == 4) {
        const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), dd = mul_i<SGN>(csub(v[1], v[3]));
        o[0] = cadd(a, c); o[2] = csub(a, c); o[1] = cadd(b, dd); o[3] = csub(b, dd);
    } else if constexpr (R == 3) {
        constexpr float C = 0.86602540378443864676f;                       // sin(2 pi / 3)
        const float2 sum = cadd(v[1], v[2]), t = mul_i<SGN>(csub(v[1], v[2]));
        const float2 h = float2{v[0].x - 0.5f * sum.x, v[0].y - 0.5f * sum.y};
        o[0] = cadd(v[0], sum);
        o[1] = float2{h.x + C * t.x, h.y + C * t.y};
        o[2] = float2{h.x - C * t.x, h.y - C * t.y};
    } else {                                                               // R == 5
        constexpr float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;   // cos(2 pi / 5), cos(4 pi / 5)
        constexpr float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;    // sin(2 pi / 5), sin(4 pi / 5)
        const float2 s1 = cadd(v[1], v[4]), s2 = cadd(v[2], v[3]), d1 = csub(v[1], v[4]), d2 = csub(v[2], v[3]);
        const float2 a1 = float2{v[0].x + C1 * s1.x + C2 * s2.x, v[0].y + C1 * s1.y + C2 * s2.y};
        const float2 a2 = float2{v[0].x + C2 * s1.x + C1 * s2.x, v[0].y + C2 * s1.y + C1 * s2.y};
        const float2 b1 = mul_i<SGN>(float2{S1 * d1.x + S2 * d2.x, S1 * d1.y + S2 * d2.y});
        const float2 b2 = mul_i<SGN>(float2{S2 * d1.x - S1 * d2.x, S2 * d1.y - S1 * d2.y});
        o[0] = float2{v[0].x + s1.x + s2.x, v[0].y + s1.y + s2.y};
        o[1] = cadd(a1, b1); o[4] = csub(a1, b1); o[2] = cadd(a2, b2); o[3] = csub(a2, b2);
    }
}

template <int R, bool ADJ>
__device__ __forceinline__ void butterfly(float2* buf, int base, int M, int j, int tstride /* N/Ns */, const FftPlan& pl, const float2* tw) {
    float2 v[R], o[R];
    (void)pl;
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = buf[base + __mul24(m, M)];
    const int t1 = __mul24(tstride, j);                   // j < Ns/R: t1 * q < N for q < R, no wrap
    if (ADJ) {
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmulc(v[q], tw[t1 * q]);
        dft_core<R, +1>(v, o);
    } else {
        dft_core<R, -1>(v, o);
#pragma unroll
        for (int q = 1; q < R; ++q) o[q] = cmul(o[q], tw[t1 * q]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) buf[base + __mul24(q, M)] = o[q];
}

// generic radix (primes 7..31): O(R^2) with the table, operands staged in registers one output at a time
template <bool ADJ>
__device__ void butterfly_generic(float2* buf, int base, int M, int j, int tstride, int R, const FftPlan& pl, const float2* tw) {
    float2 v[BNERV_FFT_MAX_RADIX], o[BNERV_FFT_MAX_RADIX];
    const int rstep = pl.N / R;
    for (int m = 0; m < R; ++m) {
        v[m] = buf[base + m * M];
        if (ADJ && m) v[m] = cmulc(v[m], tw[tstride * j * m]);
    }
    for (int q = 0; q < R; ++q) {
        float2 s = v[0];
        for (int m = 1; m < R; ++m) {
            const float2 w = tw[rstep * ((m * q) % R)];
            const float2 t = ADJ ? cmulc(v[m], w) : cmul(v[m], w);
            s.x += t.x; s.y += t.y;
        }
        if (!ADJ && q) s = cmul(s, tw[tstride * j * q]);
        o[q] = s;
    }
    for (int q = 0; q < R; ++q) buf[base + q * M] = o[q];
}

// all butterflies of one stage, radix R known at compile time.  A thread's butterflies (2-3 per stage at 720 / 1280 points) are
// independent: the loop is unrolled by UNR so that their LDS reads are in flight together instead of one round trip per butterfly.
template <int R, bool ADJ>
__device__ __forceinline__ void fft_stage_r(float2* buf, int nlines, int lstride, int Ns, const FftPlan& pl, const float2* tw) {
    const int N = pl.N, M = Ns / R, per_line = N / R, tstride = N / Ns;
    // index split by reciprocal multiplication (operands < 2^20, quotients < 2^11: the +0.5 margin dwarfs the rounding error) and
    // 24-bit multiplies: a runtime integer division costs ~40 instructions and a 32-bit multiply issues at quarter rate, and this
    // loop is instruction-latency bound (a few butterflies per thread per stage)
    const float inv_pl = 1.0f / (float)per_line, inv_M = 1.0f / (float)M;
    const int total = nlines * per_line;
    constexpr int UNR = ADJ ? 2 : 3;                       // (measured: the forward row pass 26 us at 3 / 35 at 2, the adjoint row pass 21 at 3 / 18 at 2; columns indifferent)
    for (int bf0 = threadIdx.x; bf0 < total; bf0 += UNR * blockDim.x) {
        float2 v[UNR][R], o[UNR][R];
        int base[UNR], t1[UNR];
        bool on[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int bf = bf0 + u * blockDim.x;
            on[u] = bf < total;
            const int bfc = on[u] ? bf : 0;
            const int line = (int)(((float)bfc + 0.5f) * inv_pl), rem = bfc - __mul24(line, per_line);
            const int blk = (int)(((float)rem + 0.5f) * inv_M), j = rem - __mul24(blk, M);
            base[u] = __mul24(line, lstride) + __mul24(blk, Ns) + j;
            t1[u] = __mul24(tstride, j);                   // j < Ns/R: t1 * q < N for q < R, no wrap
#pragma unroll
            for (int m = 0; m < R; ++m) v[u][m] = buf[base[u] + __mul24(m, M)];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (ADJ) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[u][q] = cmulc(v[u][q], tw[t1[u] * q]);
                dft_core<R, +1>(v[u], o[u]);
            } else {
                dft_core<R, -1>(v[u], o[u]);
#pragma unroll
                for (int q = 1; q < R; ++q) o[u][q] = cmul(o[u][q], tw[t1[u] * q]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (on[u]) {
#pragma unroll
                for (int q = 0; q < R; ++q) buf[base[u] + __mul24(q, M)] = o[u][q];
            }
        }
    }
    __syncthreads();
}

template <bool ADJ>
__device__ void fft_stage(float2* buf, int nlines, int lstride, int Ns, int R, const FftPlan& pl, const float2* tw) {
    switch (R) {
        case 2: fft_stage_r<2, ADJ>(buf, nlines, lstride, Ns, pl, tw); return;
        case 3: fft_stage_r<3, ADJ>(buf, nlines, lstride, Ns, pl, tw); return;
        case 4: fft_stage_r<4, ADJ>(buf, nlines, lstride, Ns, pl, tw); return;
        case 5: fft_stage_r<5, ADJ>(buf, nlines, lstride, Ns, pl, tw); return;
        default: break;
    }
    const int N = pl.N, M = Ns / R, per_line = N / R, tstride = N / Ns;
    const float inv_pl = 1.0f / (float)per_line, inv_M = 1.0f / (float)M;
    for (int bf = threadIdx.x; bf < nlines * per_line; bf += blockDim.x) {
        const int line = (int)(((float)bf + 0.5f) * inv_pl), rem = bf - __mul24(line, per_line);
        const int blk = (int)(((float)rem + 0.5f) * inv_M), j = rem - __mul24(blk, M);
        const int base = __mul24(line, lstride) + __mul24(blk, Ns) + j;
        butterfly_generic<ADJ>(buf, base, M, j, tstride, R, pl, tw);
    }
    __syncthreads();
}

// `tw`: the plan's twiddle table (and, for the row kernels, the frequency -> buffer position table) copied to LDS.  The copy is split
// into an issue half (global loads into registers, up to TAB_U per thread) and a commit half (LDS stores), so that a kernel can put its
// own input loads between the two: ONE memory round trip for tables and data instead of one per loop iteration (these blocks are a
// single latency chain each -- there is about one transform line per SIMD on the chip -- so every round trip shows in the launch).
constexpr int TAB_U = 8;
struct TabRegs { float2 t[TAB_U]; int p[TAB_U]; };
template <bool POS>
__device__ __forceinline__ void tables_issue(TabRegs& r, const FftPlan& pl) {
#pragma unroll
    for (int u = 0; u < TAB_U; ++u) {
        const int i = threadIdx.x + u * blockDim.x;
        r.t[u] = i < pl.N ? pl.tw[i] : float2{0.f, 0.f};
        if (POS) r.p[u] = i < pl.N ? pl.pos[i] : 0;
    }
}
template <bool POS>
__device__ __forceinline__ void tables_commit(const TabRegs& r, float2* tw, int* lpos, const FftPlan& pl) {
#pragma unroll
    for (int u = 0; u < TAB_U; ++u) {
        const int i = threadIdx.x + u * blockDim.x;
        if (i < pl.N) { tw[i] = r.t[u]; if (POS) lpos[i] = r.p[u]; }
    }
    for (int i = threadIdx.x + TAB_U * blockDim.x; i < pl.N; i += blockDim.x) { tw[i] = pl.tw[i]; if (POS) lpos[i] = pl.pos[i]; }   // (N > 2048)
}
__device__ void fft_forward(float2* buf, int nlines, int lstride, const FftPlan& pl, const float2* tw) {
    int Ns = pl.N;
    for (int s = 0; s < pl.nrad; ++s) { fft_stage<false>(buf, nlines, lstride, Ns, pl.rad[s], pl, tw); Ns /= pl.rad[s]; }
}
__device__ void fft_adjoint(float2* buf, int nlines, int lstride, const FftPlan& pl, const float2* tw) {
    int Ns = 1;
    for (int s = pl.nrad - 1; s >= 0; --s) { Ns *= pl.rad[s]; fft_stage<true>(buf, nlines, lstride, Ns, pl.rad[s], pl, tw); }
}

// The row transforms work on PAIRS of real rows: z = a + i b is ONE complex transform, and the two real rows' spectra are its
// Hermitian and anti-Hermitian parts, A[f] = (Z[f] + conj Z[W - f]) / 2, B[f] = (Z[f] - conj Z[W - f]) / (2 i).  The adjoint pass is the
// same idea backwards: a row's gradient is Re(F^H G) of its mirrored spectrum G, which only sees G's Hermitian part (the entries
// f = 0 and f = W / 2 enter with their real parts), so F^H (G_a + i G_b) = grad_a + i grad_b.  Half the butterflies of the
// row-by-row form (these kernels are instruction-bound) for the same HBM traffic.
#ifndef BNERV_FFT_LINES
#define BNERV_FFT_LINES 1
#endif
constexpr int LINES_PER_BLOCK = BNERV_FFT_LINES;            // complex lines (row pairs) per block
constexpr int ROWS_PER_BLOCK = 2 * LINES_PER_BLOCK;
static_assert(LINES_PER_BLOCK == 1 || LINES_PER_BLOCK == 2, "the staging loops of the row kernels split their index into at most two lines");
constexpr int ROW_U = LINES_PER_BLOCK == 1 ? 5 : 4;         // input elements per thread and batch (1280 points on 256 threads: one batch)
constexpr int COLS_PER_BLOCK = 4;

struct FftArgs {
    const float* pred; const float* target;
    float2* T;            // [BC][H][Wh] complex workspace: the input is real, so only the Wh = W/2 + 1 non-redundant columns of the
                          // row transform are kept (natural frequency order); the others are their conjugate mirrors
    int Wh;
    float* partial;       // [BC][ncolblk]
    float* grad;          // [BC][H][W]
    int BC, H, W;
    float gscale;         // c_fft / (B*C*H*W*2)
    int accumulate;       // rows_adj: grad += (1) or grad = (0)
    FftPlan prow, pcol;
};

__device__ __forceinline__ void fft_rows_fwd_body(const FftArgs& a, const int bx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float2* buf = reinterpret_cast<float2*>(sm);
    const int W = a.W;
    const size_t row0 = (size_t)bx * ROWS_PER_BLOCK;                    // global row index over BC*H
    const size_t nrows = (size_t)a.BC * a.H;
    const int nl = (int)min((size_t)ROWS_PER_BLOCK, nrows - row0);
    const int nlines = (nl + 1) >> 1;                                    // complex lines: rows (2 l, 2 l + 1) -> real / imaginary part
    float2* tw = buf + LINES_PER_BLOCK * W;
    int* lpos = reinterpret_cast<int*>(tw + W);
    TabRegs tr;
    tables_issue<true>(tr, a.prow);
    bool tabs_done = false;
    for (int i0 = threadIdx.x; i0 < nlines * W; i0 += blockDim.x * ROW_U) {   // 4 ROW_U loads in flight per thread (+ the tables), then the LDS stores
        float pa[ROW_U], ta[ROW_U], pb[ROW_U], tb[ROW_U];
#pragma unroll
        for (int u = 0; u < ROW_U; ++u) {
            const int i = i0 + u * blockDim.x;
            const int line = (LINES_PER_BLOCK > 1 && i >= W) ? 1 : 0, x = i - line * W;
            const bool oka = i < nlines * W, okb = oka && 2 * line + 1 < nl;
            const size_t o = (row0 + 2 * line) * W + x;
            pa[u] = oka ? a.pred[o] : 0.f;     ta[u] = oka ? a.target[o] : 0.f;
            pb[u] = okb ? a.pred[o + W] : 0.f; tb[u] = okb ? a.target[o + W] : 0.f;
        }
        if (!tabs_done) { tables_commit<true>(tr, tw, lpos, a.prow); tabs_done = true; }
#pragma unroll
        for (int u = 0; u < ROW_U; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < nlines * W) buf[i] = float2{pa[u] - ta[u], pb[u] - tb[u]};
        }
    }
    if (!tabs_done) tables_commit<true>(tr, tw, lpos, a.prow);            // (a thread without input elements still owns table entries)
    __syncthreads();
    fft_forward(buf, nlines, W, a.prow, tw);
    const int Wh = a.Wh;
    for (int i = threadIdx.x; i < nl * Wh; i += blockDim.x) {
        const int row = i / Wh, f = i - row * Wh;
        const float2* ln = buf + (row >> 1) * W;
        const float2 z = ln[lpos[f]], m = ln[lpos[f == 0 ? 0 : W - f]];
        // even row: (Z[f] + conj Z[W - f]) / 2;  odd row: (Z[f] - conj Z[W - f]) / (2 i)
        a.T[(row0 + row) * Wh + f] = (row & 1) ? float2{0.5f * (z.y + m.y), 0.5f * (m.x - z.x)} : float2{0.5f * (z.x + m.x), 0.5f * (z.y - m.y)};
    }
}
__global__ __launch_bounds__(256) void fft_rows_fwd_kernel(const FftArgs a) { fft_rows_fwd_body(a, blockIdx.x); }

__device__ __forceinline__ void fft_cols_body(const FftArgs& a, const int bx, const int bc, const int ncolblk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float2* buf = reinterpret_cast<float2*>(sm);                          // [COLS_PER_BLOCK][H]
    __shared__ float red[4];
    const int H = a.H, W = a.Wh;                                           // W: kept columns (half spectrum)
    const int v0 = bx * COLS_PER_BLOCK;
    const int nc = min(COLS_PER_BLOCK, W - v0);
    float2* T = a.T + (size_t)bc * H * W;
    float2* tw = buf + COLS_PER_BLOCK * H;
    TabRegs tr;
    tables_issue<false>(tr, a.pcol);
    bool tabs_done = false;
    for (int i0 = threadIdx.x; i0 < H * nc; i0 += blockDim.x * 8) {       // column gather: 8 loads in flight per thread (+ the twiddles)
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            const int y = i / nc, c = i - y * nc;
            v[u] = i < H * nc ? T[(size_t)y * W + v0 + c] : float2{0.f, 0.f};
        }
        if (!tabs_done) { tables_commit<false>(tr, tw, nullptr, a.pcol); tabs_done = true; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            const int y = i / nc, c = i - y * nc;
            if (i < H * nc) buf[c * H + y] = v[u];
        }
    }
    if (!tabs_done) tables_commit<false>(tr, tw, nullptr, a.pcol);
    __syncthreads();
    fft_forward(buf, nc, H, a.pcol, tw);
    float acc = 0.f;
    for (int i = threadIdx.x; i < H * nc; i += blockDim.x) {
        const float2 f = buf[i];           // lines are contiguous: nc*H elements
        const int col = v0 + i / H;        // a kept column stands for itself and for its mirror W_full - col, unless it is its own mirror
        const float wgt = (col == 0 || 2 * col == a.W) ? 1.f : 2.f;
        acc += wgt * (fabsf(f.x) + fabsf(f.y));
        buf[i] = float2{(f.x > 0.f) ? 1.f : ((f.x < 0.f) ? -1.f : 0.f), (f.y > 0.f) ? 1.f : ((f.y < 0.f) ? -1.f : 0.f)};
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.partial[(size_t)bc * ncolblk + bx] = red[0] + red[1] + red[2] + red[3];
    if (a.grad == nullptr) return;
    fft_adjoint(buf, nc, H, a.pcol, tw);
    for (int i = threadIdx.x; i < H * nc; i += blockDim.x) {
        const int y = i / nc, c = i - y * nc;
        T[(size_t)y * W + v0 + c] = buf[c * H + y];
    }
}
__global__ __launch_bounds__(256) void fft_cols_kernel(const FftArgs a) {      // (XCD-contiguous panels: see loss_mid_kernel)
    const int lb = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    fft_cols_body(a, lb % (int)gridDim.x, lb / (int)gridDim.x, gridDim.x);
}

__device__ __forceinline__ void fft_rows_adj_body(const FftArgs& a, const int bx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float2* buf = reinterpret_cast<float2*>(sm);
    const int W = a.W;
    const size_t row0 = (size_t)bx * ROWS_PER_BLOCK;
    const size_t nrows = (size_t)a.BC * a.H;
    const int nl = (int)min((size_t)ROWS_PER_BLOCK, nrows - row0);
    const int nlines = (nl + 1) >> 1;
    float2* tw = buf + LINES_PER_BLOCK * W;
    int* lpos = reinterpret_cast<int*>(tw + W);
    const int Wh = a.Wh;
    TabRegs tr;
    tables_issue<true>(tr, a.prow);
    constexpr int AU = 4;
    // P = G_a + i G_b of the row pair, G = kept columns + their conjugate mirrors (self-mirrored columns with their real parts), into DIF order
    for (int i0 = threadIdx.x, first = 1; first || i0 < nlines * Wh; i0 += blockDim.x * AU, first = 0) {   // (every thread runs the first batch: it holds the barrier)
        float2 va[AU], vb[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int i = i0 + u * blockDim.x;
            const int line = (LINES_PER_BLOCK > 1 && i >= Wh) ? 1 : 0, f = i - line * Wh;
            const bool oka = i < nlines * Wh, okb = oka && 2 * line + 1 < nl;
            const size_t o = (row0 + 2 * line) * Wh + f;
            va[u] = oka ? a.T[o] : float2{0.f, 0.f};
            vb[u] = okb ? a.T[o + Wh] : float2{0.f, 0.f};
        }
        if (first) { tables_commit<true>(tr, tw, lpos, a.prow); __syncthreads(); }     // (the position table is read below)
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < nlines * Wh) {
                const int line = (LINES_PER_BLOCK > 1 && i >= Wh) ? 1 : 0, f = i - line * Wh;
                float2 ga = va[u], gb = vb[u];
                const bool self = f == 0 || 2 * f == W;
                if (self) { ga.y = 0.f; gb.y = 0.f; }
                buf[line * W + lpos[f]] = float2{ga.x - gb.y, ga.y + gb.x};                       // G_a[f] + i G_b[f]
                if (!self) buf[line * W + lpos[W - f]] = float2{ga.x + gb.y, gb.x - ga.y};         // conj G_a[f] + i conj G_b[f]
            }
        }
    }
    // the gradient this launch adds to: loaded under the transform
    float ga[2 * ROW_U], gb[2 * ROW_U];
    const bool one_batch = nlines * W <= 2 * ROW_U * (int)blockDim.x;
    if (one_batch) {
#pragma unroll
        for (int u = 0; u < 2 * ROW_U; ++u) {
            const int i = threadIdx.x + u * blockDim.x;
            const int line = (LINES_PER_BLOCK > 1 && i >= W) ? 1 : 0, x = i - line * W;
            const bool oka = i < nlines * W, okb = oka && 2 * line + 1 < nl;
            const size_t o = (row0 + 2 * line) * W + x;
            ga[u] = (a.accumulate && oka) ? a.grad[o] : 0.f;
            gb[u] = (a.accumulate && okb) ? a.grad[o + W] : 0.f;
        }
    }
    __syncthreads();
    fft_adjoint(buf, nlines, W, a.prow, tw);
    for (int i0 = threadIdx.x; i0 < nlines * W; i0 += blockDim.x * 2 * ROW_U) {
        if (!one_batch) {
#pragma unroll
            for (int u = 0; u < 2 * ROW_U; ++u) {
                const int i = i0 + u * blockDim.x;
                const int line = (LINES_PER_BLOCK > 1 && i >= W) ? 1 : 0, x = i - line * W;
                const bool oka = i < nlines * W, okb = oka && 2 * line + 1 < nl;
                const size_t o = (row0 + 2 * line) * W + x;
                ga[u] = (a.accumulate && oka) ? a.grad[o] : 0.f;
                gb[u] = (a.accumulate && okb) ? a.grad[o + W] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * ROW_U; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < nlines * W) {
                const int line = (LINES_PER_BLOCK > 1 && i >= W) ? 1 : 0, x = i - line * W;
                const size_t o = (row0 + 2 * line) * W + x;
                const float2 r = buf[i];
                a.grad[o] = ga[u] + a.gscale * r.x;
                if (2 * line + 1 < nl) a.grad[o + W] = gb[u] + a.gscale * r.y;
            }
        }
    }
}
__global__ __launch_bounds__(256) void fft_rows_adj_kernel(const FftArgs a) { fft_rows_adj_body(a, blockIdx.x); }

// =====================================================================================================================
// final combine: per-sample loss, batch mean, stats
// =====================================================================================================================
struct FinalArgs {
    const float* stats_part;   // [B][NSB][2]
    const float* msval;        // [BC] or NULL
    const float* fft_part;     // [BC][ncolblk] or NULL
    float* loss_out; float* stats_out;
    int B, C, n_per_sample, ncolblk;
    float c_l1, c_l2, c_ms, c_fft;
};
// one wave: lane-strided sums over the partial arrays (independent loads in flight), wave reductions in fp64
__device__ __forceinline__ void loss_final_body(const FinalArgs& a) {      // ONE wave: threadIdx.x < 64
    const int lane = threadIdx.x;
    double total = 0.0;
    for (int b = 0; b < a.B; ++b) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = lane; k < NSB; k += 64) { s1 += (double)a.stats_part[((size_t)b * NSB + k) * 2]; s2 += (double)a.stats_part[((size_t)b * NSB + k) * 2 + 1]; }
        s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
        double l = (double)a.c_l1 * (s1 / a.n_per_sample) + (double)a.c_l2 * (s2 / a.n_per_sample);
        double ms = 0.0;
        if (a.msval) {
            for (int c = 0; c < a.C; ++c) ms += (double)a.msval[b * a.C + c];
            ms /= a.C;
            l += (double)a.c_ms * (1.0 - ms);
        }
        if (a.fft_part) {
            double f = 0.0;
            const int nf = a.C * a.ncolblk;
            for (int k = lane; k < nf; k += 64) f += (double)a.fft_part[(size_t)b * nf + k];
            f = wave_sum_d(f);
            l += (double)a.c_fft * f / (2.0 * (double)a.n_per_sample);
        }
        if (lane == 0) {
            float* so = a.stats_out + b * BNERV_LOSS_STATS;
            so[0] = (float)l; so[1] = (float)s1; so[2] = (float)s2; so[3] = (float)ms;
            // psnr_fn_single (hnerv_utils.py:400-403) on the same sums, exactly as psnr_final_kernel computes it
            const float mse = (float)(s2 / (double)a.n_per_sample);
            so[4] = -10.0f * log10f(mse + 1e-9f);
        }
        total += l;
    }
    if (lane == 0) a.loss_out[0] = (float)(total / a.B);
}
__global__ __launch_bounds__(64) void loss_final_kernel(const FinalArgs a) { loss_final_body(a); }

// ---- merged launches of the Fusion losses on an even pyramid (MS-SSIM + spectral term + gradient): the 10 launches of the branch
// sequence become 6.  What is merged are launches that do not depend on each other, as block ranges of one grid (the bodies are
// unchanged, every number is computed by the same instructions in the same order):
//   head:  row FFTs of pred - target  |  the 4-level pyramid of both images  |  the L1 / L2 partial sums
//   mid:   column FFTs (+ spectral partials, + adjoint columns)  |  the MS-SSIM coefficients (needs the SSIM launch before it)
//   tail:  level-0 SSIM gradient with the 0.25-chain  |  loss_final (needs only the partials of head and mid)
struct LossHeadArgs { FftArgs f; PyrArgs p; const float* pred; const float* target; float* stats_part; int nps, n_fft, n_pyr, pyr_gx, pyr_gy; };
__global__ __launch_bounds__(256) void loss_head_kernel(const LossHeadArgs a) {
    int b = blockIdx.x;
    if (b < a.n_fft) { fft_rows_fwd_body(a.f, b); return; }               // the long blocks first
    b -= a.n_fft;
    if (b < a.n_pyr) { const int bx = b % a.pyr_gx, r = b / a.pyr_gx; pyramid_body(a.p, bx, r % a.pyr_gy, r / a.pyr_gy); return; }
    b -= a.n_pyr;
    diff_stats_body(a.pred, a.target, a.stats_part, a.nps, b % NSB, b / NSB);
}
struct LossMidArgs { FftArgs f; CoefArgs c; int ncolblk, n_cols; };
__global__ __launch_bounds__(256) void loss_mid_kernel(const LossMidArgs a) {
    const int b = blockIdx.x;
    if (b < a.n_cols) {
        // neighbouring column panels share the 128-byte lines of every row of T: give each XCD (blocks b, b + 8, ...) a CONTIGUOUS run of
        // panels, so that a line is fetched into one L2 instead of four
        const int xcd = b & 7, k = b >> 3, per = a.n_cols >> 3, extra = a.n_cols & 7;
        const int lb = xcd * per + min(xcd, extra) + k;
        fft_cols_body(a.f, lb % a.ncolblk, lb / a.ncolblk, a.ncolblk);
    } else ms_coef_body(a.c, b - a.n_cols);
}
struct LossTailArgs { SsimArgs s; CoarseChain cc; FinalArgs fin; int gx, gy, n0, BC; };
__global__ __launch_bounds__(256) void loss_tail_kernel(const LossTailArgs a) {
    __shared__ __attribute__((aligned(16))) float sl[SSIM_BWD_LDS];
    if ((int)blockIdx.x < a.n0) { const int b = xcd_remap((int)blockIdx.x, a.n0); const int bx = b % a.gx, r = b / a.gx; ssim_bwd_body<true>(a.s, &a.cc, sl, bx, r % a.gy, r / a.gy, a.BC); }
    else if (threadIdx.x < 64) loss_final_body(a.fin);
}
// coarse:  adjoint row FFTs (the spectral gradient, WRITTEN to grad)  |  the coarser levels' SSIM gradient terms.  Both only need what `mid`
// left behind and neither fills the chip (1080 one-line blocks of 17 us, 619 short tiles at 720p); the level-0 launch then ADDS its
// gradient to the spectral one (SsimArgs::acc) instead of a sixth launch accumulating onto it.  One dynamic LDS allocation serves both bodies.
// (Slot-bound at 720p: 84 VGPRs -> 5 blocks per CU = 1280 slots for 1080 x ~14 us + 1857 x ~5.5 us of block time = 23.8 us against 12.7 + 17.9
// as two launches; forcing 80 VGPRs for a sixth block spills 22 registers in the butterflies and measured 29 us.)
struct LossCoarseArgs { FftArgs f; SsimAllArgs s; int n_adj, gx, BC; };
__global__ __launch_bounds__(256) void loss_coarse_kernel(const LossCoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x;
    if (b < a.n_adj) { fft_rows_adj_body(a.f, b); return; }              // the long blocks first
    ssim_bwd_coarse_body(a.s, sm, b - a.n_adj, a.gx, a.BC);
}
__global__ void msssim_final_kernel(const float* __restrict__ msval, float* __restrict__ out, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += msval[b * C + c];
    out[b] = s / C;
}

// =====================================================================================================================
// workspace layout
// =====================================================================================================================
struct WsLayout {
    size_t stats_part, pyrX[LV], pyrY[LV], dXl[LV], Gl[LV], ssim_part[LV], msval, coef, T, fft_part, total;
    int tiles[LV];
    Pyr pyr;
    int ncolblk;
};
static size_t align64(size_t x) { return (x + 63) & ~size_t(63); }
static WsLayout make_layout(int B, int C, int H, int W, bool use_ms, bool use_fft) {
    WsLayout L{};
    size_t off = 0;                                   // in floats
    auto take = [&](size_t n) { size_t o = off; off = align64(off + n); return o; };
    const size_t BC = (size_t)B * C;
    L.stats_part = take((size_t)B * NSB * 2);
    L.pyr = make_pyr(H, W);
    if (use_ms) {
        for (int l = 0; l < LV; ++l) {
            const size_t n = BC * L.pyr.H[l] * L.pyr.W[l];
            if (l > 0) { L.pyrX[l] = take(n); L.pyrY[l] = take(n); L.dXl[l] = take(n); }
            L.Gl[l] = take(3 * n);
            L.tiles[l] = cdiv(L.pyr.H[l] - HW_, STH) * cdiv(L.pyr.W[l] - HW_, STW);
            L.ssim_part[l] = take(BC * L.tiles[l]);
        }
        L.msval = take(BC);
        L.coef = take(BC * LV);
    }
    if (use_fft) {
        L.ncolblk = cdiv(W / 2 + 1, COLS_PER_BLOCK);
        L.T = take(BC * H * (size_t)(W / 2 + 1) * 2);
        L.fft_part = take(BC * L.ncolblk);
    }
    L.total = off;
    return L;
}

static bool loss_fused() {                              // BNERV_LOSS_FUSED=0: the level-by-level MS-SSIM launches (A/B switch, read per call: tests compare the forms)
    const char* e = getenv("BNERV_LOSS_FUSED");
    return !(e && e[0] == '0');
}
static bool even_pyramid(const WsLayout& L) {         // every pooled level has even sides: no padding anywhere, aligned 2x2 cells
    if (!loss_fused()) return false;
    for (int l = 0; l < LV - 1; ++l) if ((L.pyr.H[l] & 1) || (L.pyr.W[l] & 1)) return false;
    return true;
}
static bool adj_late() {                                // BNERV_LOSS_ADJ=late: round 4's order -- the adjoint row pass as the LAST launch, accumulating
    const char* e = getenv("BNERV_LOSS_ADJ");           // onto the finished SSIM gradient (A/B switch, read per call; the gradient differs in the last bit:
    return e && !strcmp(e, "late");                     // fma(k, r, d) there, d + k r here)
}
// the 2x2 means of an ODD pyramid, level by level (zero padding where a side is odd, count_include_pad)
static int launch_pools(hipStream_t st, const float* X, const float* Y, float* ws, const WsLayout& L, int BC) {
    const float* Xl = X; const float* Yl = Y;
    for (int l = 0; l < LV - 1; ++l) {
        const int Hl = L.pyr.H[l], Wl = L.pyr.W[l], Ho = L.pyr.H[l + 1], Wo = L.pyr.W[l + 1];
        const size_t n = (size_t)BC * Ho * Wo;
        int gx = (int)((n + 255) / 256); if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(avgpool2_kernel, dim3(gx, 1, 2), dim3(256), 0, st, Xl, Yl, ws + L.pyrX[l + 1], ws + L.pyrY[l + 1], BC, Hl, Wl, Ho, Wo, Hl % 2, Wl % 2);
        BNERV_LAUNCH_CHECK("avgpool2");
        Xl = ws + L.pyrX[l + 1]; Yl = ws + L.pyrY[l + 1];
    }
    return BNERV_OK;
}

static bool loss_merged() {                             // BNERV_LOSS_MERGED=0: every launch of the even-pyramid form on its own (A/B switch, read per call)
    const char* e = getenv("BNERV_LOSS_MERGED");
    return !(e && e[0] == '0');
}

// arguments of the even-pyramid forward launches (pyramid, every level's statistics, coefficients)
static int fill_even_forward(const float* X, const float* Y, float* ws, const WsLayout& L, int BC, float chain, bool want_g, PyrArgs& pa, SsimAllArgs& sa, CoefArgs& ca) {
    static const float wts[LV] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    const Win win = make_win();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    pa.src[0] = X; pa.src[1] = Y; pa.planes = BC;
    for (int l = 0; l < LV; ++l) {
        pa.H[l] = L.pyr.H[l]; pa.W[l] = L.pyr.W[l];
        if (L.pyr.H[l] <= HW_ || L.pyr.W[l] <= HW_) return bnerv_set_error(BNERV_E_ARG, "ms_ssim: level %d is %dx%d, needs > %d on both sides", l, L.pyr.H[l], L.pyr.W[l], HW_);
        if (l > 0) { pa.dst[0][l] = ws + L.pyrX[l]; pa.dst[1][l] = ws + L.pyrY[l]; }
    }
    int nblk = 0;
    for (int l = 0; l < LV; ++l) {
        SsimArgs& a = sa.lv[l];
        a.X = l ? ws + L.pyrX[l] : X; a.Y = l ? ws + L.pyrY[l] : Y; a.partial = ws + L.ssim_part[l]; a.H = L.pyr.H[l]; a.W = L.pyr.W[l];
        a.G = want_g ? ws + L.Gl[l] : nullptr;
        a.tiles_x = cdiv(a.W - HW_, STW); a.tiles_y = cdiv(a.H - HW_, STH); a.C1 = C1; a.C2 = C2; a.win = win;
        sa.first[l] = nblk;
        nblk += a.tiles_x * a.tiles_y;
        ca.partial[l] = ws + L.ssim_part[l]; ca.tiles[l] = L.tiles[l];
        ca.inv_nvalid[l] = 1.0f / ((float)(a.H - HW_) * (float)(a.W - HW_));
        ca.weights[l] = wts[l];
    }
    sa.first[LV] = nblk;
    ca.msval = ws + L.msval; ca.coef = ws + L.coef; ca.BC = BC; ca.chain = chain;
    return BNERV_OK;
}
// ... and of the even-pyramid backward launches (the coarser levels' own terms; level 0 with the 0.25-chain over them)
static void fill_even_backward(const float* X, const float* Y, float* grad, float* ws, const WsLayout& L, int BC, float k_l1, float k_l2, int acc, SsimAllArgs& sa, CoarseChain& cc) {
    const Win win = make_win();
    int nblk = 0;
    for (int l = 0; l < LV; ++l) {
        SsimArgs& a = sa.lv[l];
        a.X = l ? ws + L.pyrX[l] : X; a.Y = l ? ws + L.pyrY[l] : Y; a.coef = ws + L.coef + (size_t)l * BC;
        a.dX = l ? ws + L.dXl[l] : grad; a.H = L.pyr.H[l]; a.W = L.pyr.W[l]; a.C1 = 0.01f * 0.01f; a.C2 = 0.03f * 0.03f; a.win = win;
        a.k_l1 = k_l1; a.k_l2 = k_l2; a.G = ws + L.Gl[l];
        sa.first[l] = nblk;
        if (l >= 1) nblk += cdiv(a.W, STW) * cdiv(a.H, STH);
        cc.own[l] = l ? ws + L.dXl[l] : nullptr; cc.H[l] = a.H; cc.W[l] = a.W; cc.ph[l] = a.H % 2; cc.pw[l] = a.W % 2;
    }
    sa.lv[0].acc = acc;
    sa.first[LV] = nblk;
    cc.n = LV - 1;
}

static int run_ms_forward(hipStream_t st, const float* X, const float* Y, float* ws, const WsLayout& L, int B, int C, float chain, bool want_g) {
    const int BC = B * C;
    const Win win = make_win();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    if (even_pyramid(L)) {
        // 3 launches instead of 10: the pyramid, every level's statistics, the coefficients
        PyrArgs pa{}; SsimAllArgs sa{}; CoefArgs ca{};
        int rc = fill_even_forward(X, Y, ws, L, BC, chain, want_g, pa, sa, ca);
        if (rc) return rc;
        hipLaunchKernelGGL(pyramid_kernel, dim3(cdiv(L.pyr.W[1], 16), cdiv(L.pyr.H[1], 16), 2 * BC), dim3(256), 0, st, pa);
        BNERV_LAUNCH_CHECK("pyramid");
        hipLaunchKernelGGL(ssim_fwd_all_kernel, dim3(sa.first[LV], BC), dim3(256), 0, st, sa);
        BNERV_LAUNCH_CHECK("ssim_fwd_all");
        hipLaunchKernelGGL(ms_coef_kernel, dim3(BC), dim3(320), 0, st, ca);
        BNERV_LAUNCH_CHECK("ms_coef");
        return BNERV_OK;
    }
    {
        // odd pyramid (1080 -> 540 -> 270 -> 135 -> 68: the 1080p configs): the padded 2x2 means level by level, then EVERY level's statistics in
        // ONE launch (the levels only depend on the pyramid; the same body per tile as the level-by-level launches, so the same bits) and the
        // coefficients: 6 launches instead of 10.  BNERV_LOSS_FUSED=0 keeps the level-by-level form.
        const char* e = getenv("BNERV_LOSS_FUSED");
        if (!(e && e[0] == '0')) {
            PyrArgs pa{}; SsimAllArgs sa{}; CoefArgs ca{};
            int rc = fill_even_forward(X, Y, ws, L, BC, chain, want_g, pa, sa, ca);
            if (rc) return rc;
            rc = launch_pools(st, X, Y, ws, L, BC);
            if (rc) return rc;
            hipLaunchKernelGGL(ssim_fwd_all_kernel, dim3(sa.first[LV], BC), dim3(256), 0, st, sa);
            BNERV_LAUNCH_CHECK("ssim_fwd_all");
            hipLaunchKernelGGL(ms_coef_kernel, dim3(BC), dim3(320), 0, st, ca);
            BNERV_LAUNCH_CHECK("ms_coef");
            return BNERV_OK;
        }
    }
    const float* Xl = X; const float* Yl = Y;
    CoefArgs ca{};
    static const float weights[LV] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    for (int l = 0; l < LV; ++l) {
        const int Hl = L.pyr.H[l], Wl = L.pyr.W[l];
        if (Hl <= HW_ || Wl <= HW_) return bnerv_set_error(BNERV_E_ARG, "ms_ssim: level %d is %dx%d, needs > %d on both sides", l, Hl, Wl, HW_);
        SsimArgs a{};
        a.X = Xl; a.Y = Yl; a.partial = ws + L.ssim_part[l]; a.H = Hl; a.W = Wl; a.G = want_g ? ws + L.Gl[l] : nullptr;
        a.tiles_x = cdiv(Wl - HW_, STW); a.tiles_y = cdiv(Hl - HW_, STH); a.C1 = C1; a.C2 = C2; a.win = win;
        dim3 grid(a.tiles_x, a.tiles_y, BC);
        if (l == LV - 1) hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(256), 0, st, a);
        BNERV_LAUNCH_CHECK("ssim_fwd");
        ca.partial[l] = ws + L.ssim_part[l]; ca.tiles[l] = L.tiles[l];
        ca.inv_nvalid[l] = 1.0f / ((float)(Hl - HW_) * (float)(Wl - HW_));
        ca.weights[l] = weights[l];
        if (l < LV - 1) {
            const int Ho = L.pyr.H[l + 1], Wo = L.pyr.W[l + 1];
            const size_t n = (size_t)BC * Ho * Wo;
            int gx = (int)((n + 255) / 256); if (gx > 4096) gx = 4096;
            hipLaunchKernelGGL(avgpool2_kernel, dim3(gx, 1, 2), dim3(256), 0, st, Xl, Yl, ws + L.pyrX[l + 1], ws + L.pyrY[l + 1], BC, Hl, Wl, Ho, Wo, Hl % 2, Wl % 2);
            BNERV_LAUNCH_CHECK("avgpool2");
            Xl = ws + L.pyrX[l + 1]; Yl = ws + L.pyrY[l + 1];
        }
    }
    ca.msval = ws + L.msval; ca.coef = ws + L.coef; ca.BC = BC; ca.chain = chain;
    hipLaunchKernelGGL(ms_coef_kernel, dim3(BC), dim3(320), 0, st, ca);
    BNERV_LAUNCH_CHECK("ms_coef");
    return BNERV_OK;
}

// acc: the level-0 launch adds to `grad` (the spectral gradient is already there) instead of writing it
static int run_ms_backward(hipStream_t st, const float* X, const float* Y, float* grad, float* ws, const WsLayout& L, int B, int C, float k_l1, float k_l2, int acc) {
    const int BC = B * C;
    const Win win = make_win();
    if (loss_fused()) {
        // 2 launches instead of 5: the coarser levels' own terms together, then level 0 with the 0.25-chain over them (odd pyramids too:
        // the chain walks the padded cells, CoarseChain::ph / pw)
        SsimAllArgs sa{};
        CoarseChain cc{};
        fill_even_backward(X, Y, grad, ws, L, BC, k_l1, k_l2, acc, sa, cc);
        hipLaunchKernelGGL(ssim_bwd_coarse_all_kernel, dim3(sa.first[LV], BC), dim3(256), 0, st, sa);
        BNERV_LAUNCH_CHECK("ssim_bwd_coarse_all");
        hipLaunchKernelGGL(ssim_bwd_level0_chain_kernel, dim3(cdiv(L.pyr.W[0], STW), cdiv(L.pyr.H[0], STH), BC), dim3(256), 0, st, sa.lv[0], cc);
        BNERV_LAUNCH_CHECK("ssim_bwd_level0");
        return BNERV_OK;
    }
    for (int l = LV - 1; l >= 0; --l) {
        const int Hl = L.pyr.H[l], Wl = L.pyr.W[l];
        SsimArgs a{};
        a.X = l ? ws + L.pyrX[l] : X; a.Y = l ? ws + L.pyrY[l] : Y;
        a.coef = ws + L.coef + (size_t)l * BC;
        a.dX = l ? ws + L.dXl[l] : grad;
        a.H = Hl; a.W = Wl; a.C1 = 0.01f * 0.01f; a.C2 = 0.03f * 0.03f; a.win = win;
        if (l < LV - 1) { a.dcoarse = ws + L.dXl[l + 1]; a.Hc = L.pyr.H[l + 1]; a.Wc = L.pyr.W[l + 1]; a.ph = Hl % 2; a.pw = Wl % 2; }
        a.k_l1 = k_l1; a.k_l2 = k_l2; a.acc = l == 0 ? acc : 0;
        a.G = ws + L.Gl[l];
        dim3 grid(cdiv(Wl, STW), cdiv(Hl, STH), BC);
        if (l == 0) hipLaunchKernelGGL((ssim_bwd_from_g_kernel<true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((ssim_bwd_from_g_kernel<false>), grid, dim3(256), 0, st, a);
        BNERV_LAUNCH_CHECK("ssim_bwd");
    }
    return BNERV_OK;
}

static int launch_stats(hipStream_t st, const float* p, const float* t, float* part, int B, int n_per_sample) {
    hipLaunchKernelGGL(diff_stats_kernel, dim3(NSB, B), dim3(256), 0, st, p, t, part, n_per_sample);
    BNERV_LAUNCH_CHECK("diff_stats");
    return BNERV_OK;
}

}  // namespace

extern "C" size_t bnerv_loss_ws_bytes(int B, int C, int H, int W, int use_ms, int use_fft) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return make_layout(B, C, H, W, use_ms != 0, use_fft != 0).total * sizeof(float);
}

// (A second stream for the spectral branch was measured and dropped: forked with an event pair the loss section of the trace shrinks
// from 207 to 194 us, but the column FFT doubles next to the SSIM maps, the fork / join edges idle the chip for 6 + 10 us and a
// two-branch graph replays slower than a linear one: C1 1.789 ms forked against 1.759 ms.  Independent launches are merged as block
// ranges of one grid instead -- loss_head / loss_mid / loss_tail above.)
extern "C" int bnerv_loss_fwd_bwd(void* stream, const bnerv_loss_desc* dp) {
    BNERV_REQUIRE(dp != nullptr, "loss: null descriptor");
    const bnerv_loss_desc d = *dp;
    BNERV_REQUIRE(d.pred && d.target && d.loss_out && d.stats_out && d.ws, "loss: null tensor");
    BNERV_REQUIRE(d.B > 0 && d.C > 0 && d.H > 0 && d.W > 0 && d.B <= 65535, "loss: bad dims");
    BNERV_REQUIRE((size_t)d.C * d.H * d.W < (size_t)1 << 31, "loss: sample too large");
    const bool use_ms = d.c_ms != 0.f, use_fft = d.c_fft != 0.f;
    const WsLayout L = make_layout(d.B, d.C, d.H, d.W, use_ms, use_fft);
    if (d.ws_bytes < L.total * sizeof(float)) return bnerv_set_error(BNERV_E_WS, "loss: workspace %zu < %zu", d.ws_bytes, L.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* ws = reinterpret_cast<float*>(d.ws);
    const int nps = d.C * d.H * d.W, BC = d.B * d.C;
    const float k_l1 = d.c_l1 / ((float)d.B * (float)nps), k_l2 = 2.0f * d.c_l2 / ((float)d.B * (float)nps);
    if (use_ms && (d.H <= 160 || d.W <= 160)) return bnerv_set_error(BNERV_E_ARG, "loss: MS-SSIM needs min(H,W) > 160 (got %dx%d)", d.H, d.W);
    FftArgs a{};
    size_t lds_row = 0, lds_col = 0;
    int nrowblk = 0;
    if (use_fft) {
        if (!make_plan(d.W, &a.prow) || !make_plan(d.H, &a.pcol))
            return bnerv_set_error(BNERV_E_ARG, "loss: FFT size %dx%d has a prime factor > %d", d.H, d.W, BNERV_FFT_MAX_RADIX);
        a.pred = d.pred; a.target = d.target; a.T = reinterpret_cast<float2*>(ws + L.T); a.Wh = d.W / 2 + 1; a.partial = ws + L.fft_part; a.grad = d.grad;
        a.BC = BC; a.H = d.H; a.W = d.W; a.gscale = d.c_fft / ((float)d.B * (float)nps * 2.0f); a.accumulate = 1;
        lds_row = (size_t)(LINES_PER_BLOCK + 1) * d.W * sizeof(float2) + (size_t)d.W * sizeof(int); lds_col = (size_t)(COLS_PER_BLOCK + 1) * d.H * sizeof(float2);   // + twiddle table (+ position table)
        BNERV_REQUIRE(lds_row <= 160 * 1024 && lds_col <= 160 * 1024, "loss: frame %dx%d too large for the LDS FFT", d.H, d.W);
        nrowblk = cdiv(BC * d.H, ROWS_PER_BLOCK);
    }
    FinalArgs f{};
    f.stats_part = ws + L.stats_part; f.msval = use_ms ? ws + L.msval : nullptr; f.fft_part = use_fft ? ws + L.fft_part : nullptr;
    f.loss_out = d.loss_out; f.stats_out = d.stats_out; f.B = d.B; f.C = d.C; f.n_per_sample = nps; f.ncolblk = L.ncolblk;
    f.c_l1 = d.c_l1; f.c_l2 = d.c_l2; f.c_ms = d.c_ms; f.c_fft = d.c_fft;

    const bool late = adj_late();
    if (use_ms && use_fft && d.grad && loss_fused() && loss_merged()) {
        // 5 launches on an even pyramid (9 on an odd one, whose four pooled levels are a launch each): head (row FFTs | pyramid | L1 / L2 sums),
        // SSIM statistics, mid (column FFTs | coefficients), coarse (adjoint row FFTs -> the spectral gradient | coarse SSIM gradients),
        // tail (level-0 gradient, added to the spectral one | loss_final).  BNERV_LOSS_ADJ=late: the adjoint rows as a sixth launch behind the tail.
        const bool even = even_pyramid(L);
        LossHeadArgs ha{}; LossMidArgs ma{}; LossTailArgs ta{};
        SsimAllArgs sf{};
        int rc = fill_even_forward(d.pred, d.target, ws, L, BC, -d.c_ms / (float)BC, true, ha.p, sf, ma.c);
        if (rc) return rc;
        ha.f = a; ha.pred = d.pred; ha.target = d.target; ha.stats_part = ws + L.stats_part; ha.nps = nps;
        ha.n_fft = nrowblk; ha.pyr_gx = cdiv(L.pyr.W[1], 16); ha.pyr_gy = cdiv(L.pyr.H[1], 16); ha.n_pyr = even ? ha.pyr_gx * ha.pyr_gy * 2 * BC : 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row);
        hipLaunchKernelGGL(loss_head_kernel, dim3(ha.n_fft + ha.n_pyr + NSB * d.B), dim3(256), lds_row, st, ha);
        BNERV_LAUNCH_CHECK("loss_head");
        if (!even) { rc = launch_pools(st, d.pred, d.target, ws, L, BC); if (rc) return rc; }
        hipLaunchKernelGGL(ssim_fwd_all_kernel, dim3(sf.first[LV], BC), dim3(256), 0, st, sf);
        BNERV_LAUNCH_CHECK("ssim_fwd_all");
        ma.f = a; ma.ncolblk = L.ncolblk; ma.n_cols = L.ncolblk * BC;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_col);
        hipLaunchKernelGGL(loss_mid_kernel, dim3(ma.n_cols + BC), dim3(256), lds_col, st, ma);
        BNERV_LAUNCH_CHECK("loss_mid");
        SsimAllArgs sb{};
        fill_even_backward(d.pred, d.target, d.grad, ws, L, BC, k_l1, k_l2, late ? 0 : 1, sb, ta.cc);
        if (late) {
            hipLaunchKernelGGL(ssim_bwd_coarse_all_kernel, dim3(sb.first[LV], BC), dim3(256), 0, st, sb);
            BNERV_LAUNCH_CHECK("ssim_bwd_coarse_all");
        } else {
            LossCoarseArgs ca{};
            ca.f = a; ca.f.accumulate = 0; ca.s = sb; ca.n_adj = nrowblk; ca.gx = sb.first[LV]; ca.BC = BC;
            const size_t lds_c = lds_row > SSIM_BWD_LDS * sizeof(float) ? lds_row : SSIM_BWD_LDS * sizeof(float);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_coarse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
            hipLaunchKernelGGL(loss_coarse_kernel, dim3(ca.n_adj + ca.gx * BC), dim3(256), lds_c, st, ca);
            BNERV_LAUNCH_CHECK("loss_coarse");
        }
        ta.s = sb.lv[0]; ta.fin = f; ta.gx = cdiv(L.pyr.W[0], STW); ta.gy = cdiv(L.pyr.H[0], STH); ta.BC = BC; ta.n0 = ta.gx * ta.gy * BC;
        hipLaunchKernelGGL(loss_tail_kernel, dim3(ta.n0 + 1), dim3(256), 0, st, ta);
        BNERV_LAUNCH_CHECK("loss_tail");
        if (late) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_adj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row);
            hipLaunchKernelGGL(fft_rows_adj_kernel, dim3(nrowblk), dim3(256), lds_row, st, a);
            BNERV_LAUNCH_CHECK("fft_rows_adj");
        }
        return BNERV_OK;
    }

    // one launch per kernel.  With both an MS-SSIM and a spectral term the spectral gradient is written FIRST and the level-0 SSIM launch adds
    // to it -- the arithmetic of the merged form above, so the two forms agree bit for bit (BNERV_LOSS_ADJ=late: the other order, in both forms)
    const bool fft_first = use_ms && use_fft && d.grad && !late;
    int rc = launch_stats(st, d.pred, d.target, ws + L.stats_part, d.B, nps);
    if (rc) return rc;
    auto run_fft = [&](int accumulate) -> int {
        FftArgs fa = a; fa.accumulate = accumulate;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_adj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_col);
        hipLaunchKernelGGL(fft_rows_fwd_kernel, dim3(nrowblk), dim3(256), lds_row, st, fa);
        BNERV_LAUNCH_CHECK("fft_rows_fwd");
        hipLaunchKernelGGL(fft_cols_kernel, dim3(L.ncolblk, BC), dim3(256), lds_col, st, fa);
        BNERV_LAUNCH_CHECK("fft_cols");
        if (d.grad) {
            hipLaunchKernelGGL(fft_rows_adj_kernel, dim3(nrowblk), dim3(256), lds_row, st, fa);
            BNERV_LAUNCH_CHECK("fft_rows_adj");
        }
        return BNERV_OK;
    };
    if (use_ms) {
        rc = run_ms_forward(st, d.pred, d.target, ws, L, d.B, d.C, -d.c_ms / (float)BC, d.grad != nullptr);
        if (rc) return rc;
        if (fft_first) { rc = run_fft(0); if (rc) return rc; }
        if (d.grad) { rc = run_ms_backward(st, d.pred, d.target, d.grad, ws, L, d.B, d.C, k_l1, k_l2, fft_first ? 1 : 0); if (rc) return rc; }
    } else if (d.grad) {
        const size_t n = (size_t)d.B * nps;
        int gx = (int)((n + 1023) / 1024); if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(grad_l1l2_kernel, dim3(gx), dim3(256), 0, st, d.pred, d.target, d.grad, n, k_l1, k_l2);
        BNERV_LAUNCH_CHECK("grad_l1l2");
    }
    if (use_fft && !fft_first) { rc = run_fft(1); if (rc) return rc; }
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, f);
    BNERV_LAUNCH_CHECK("loss_final");
    return BNERV_OK;
}

extern "C" int bnerv_msssim(void* stream, const float* x, const float* y, float* out, void* wsv, size_t ws_bytes, int B, int C, int H, int W) {
    BNERV_REQUIRE(x && y && out && wsv && B > 0 && C > 0, "msssim: bad args");
    if (H <= 160 || W <= 160) return bnerv_set_error(BNERV_E_ARG, "msssim: needs min(H,W) > 160 (got %dx%d)", H, W);
    const WsLayout L = make_layout(B, C, H, W, true, false);
    if (ws_bytes < L.total * sizeof(float)) return bnerv_set_error(BNERV_E_WS, "msssim: workspace %zu < %zu", ws_bytes, L.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* ws = reinterpret_cast<float*>(wsv);
    int rc = run_ms_forward(st, x, y, ws, L, B, C, 0.f, false);
    if (rc) return rc;
    hipLaunchKernelGGL(msssim_final_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, ws + L.msval, out, B, C);
    BNERV_LAUNCH_CHECK("msssim_final");
    return BNERV_OK;
}

extern "C" size_t bnerv_psnr_ws_bytes(int B, int C, int H, int W) { (void)C; (void)H; (void)W; return B > 0 ? (size_t)B * NSB * 2 * sizeof(float) : 0; }

extern "C" int bnerv_psnr(void* stream, const float* o, const float* gt, float* psnr, void* ws, size_t ws_bytes, int B, int C, int H, int W) {
    BNERV_REQUIRE(o && gt && psnr && ws && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0, "psnr: bad args");
    BNERV_REQUIRE((size_t)C * H * W < (size_t)1 << 31, "psnr: sample too large");
    if (ws_bytes < bnerv_psnr_ws_bytes(B, C, H, W)) return bnerv_set_error(BNERV_E_WS, "psnr: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_stats(st, o, gt, (float*)ws, B, C * H * W);
    if (rc) return rc;
    hipLaunchKernelGGL(psnr_final_kernel, dim3(B), dim3(64), 0, st, (const float*)ws, psnr, B, C * H * W);
    BNERV_LAUNCH_CHECK("psnr_final");
    return BNERV_OK;
}

// Create the FFT twiddle tables for an H x W frame ahead of time (they are otherwise created on first use; creation
// allocates and copies synchronously, which is illegal inside a stream capture).
extern "C" int bnerv_fft_prepare(int H, int W) {
    FftPlan p;
    if (!make_plan(H, &p) || !make_plan(W, &p)) return bnerv_set_error(BNERV_E_ARG, "fft_prepare: %dx%d unsupported", H, W);
    return BNERV_OK;
}
