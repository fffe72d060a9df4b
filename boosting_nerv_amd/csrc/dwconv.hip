// dwconv.hip -- depthwise KxK convolution (K odd, <= 7; stride 1, padding K/2) for the ConvNeXt encoder block of HNeRV_Boost
// (reference: model_blocks.py:223-247, `self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)`), forward,
// data gradient and weight/bias gradient.  One channel plane has no reduction over channels, so this is a streaming stencil:
// HBM-bound (read x once + halo, write y once), 49 FMAs per output from an LDS tile.  Row N3 of SURVEY 8(f): stock
// PyTorch-ROCm falls back to MIOpen's naive fp32 kernels for these shapes (6.4 ms per data gradient at 64x216x384).
//   fwd       y[b,c,p]   = bias[c] + sum_t w[c,t] * x[b,c,p + t - K/2]
//   bwd data  dx[b,c,p]  = sum_t w[c,T-1-t] * g[b,c,p + t - K/2]                    (same kernel, flipped taps, no bias)
//   bwd wgt   dw[c,t]    = sum_{b,p} g[b,c,p] * x[b,c,p + t - K/2],  db[c] = sum g   (per-block slabs, deterministic finish)
#include "common.h"
#include "sidejob.h"

namespace {

constexpr int DTH = 16, DTW = 64;                    // output tile per block (256 threads: 4 outputs each, consecutive in x)
constexpr int KMAX = 7, PMAX = KMAX / 2;

template <bool FLIP>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ y, int C, int H, int W, int K) {
    __shared__ float s_x[DTH + 2 * PMAX][DTW + 2 * PMAX + 2];
    __shared__ float s_w[KMAX * KMAX];
    const int P = K / 2, T = K * K;
    const int bc = blockIdx.z, c = bc % C;
    const int y0 = blockIdx.y * DTH, x0 = blockIdx.x * DTW;
    const float* xp = x + (size_t)bc * H * W;
    const int RH = DTH + 2 * P, RW = DTW + 2 * P;
    for (int i = threadIdx.x; i < RH * RW; i += 256) {
        const int r = i / RW, q = i - r * RW;
        const int gy = y0 + r - P, gx = x0 + q - P;
        s_x[r][q] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xp[(size_t)gy * W + gx] : 0.f;
    }
    if ((int)threadIdx.x < T) s_w[threadIdx.x] = w[(size_t)c * T + (FLIP ? T - 1 - (int)threadIdx.x : (int)threadIdx.x)];
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = (threadIdx.x & 15) * 4;          // 16 rows x 16 quads
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ky = 0; ky < K; ++ky) {
        float row[4 + KMAX - 1];
#pragma unroll
        for (int i = 0; i < 4 + KMAX - 1; ++i) row[i] = i < 4 + K - 1 ? s_x[ty + ky][tx + i] : 0.f;
#pragma unroll
        for (int kx = 0; kx < KMAX; ++kx) {
            if (kx < K) {
                const float wv = s_w[ky * K + kx];
                a0 = fmaf(wv, row[kx], a0); a1 = fmaf(wv, row[kx + 1], a1); a2 = fmaf(wv, row[kx + 2], a2); a3 = fmaf(wv, row[kx + 3], a3);
            }
        }
    }
    const float bv = (!FLIP && bias) ? bias[c] : 0.f;
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < H) {
        float* yp = y + (size_t)bc * H * W + (size_t)gy * W + gx;
        if (gx + 3 < W && (W & 3) == 0) {
            *reinterpret_cast<f32x4*>(yp) = f32x4{a0 + bv, a1 + bv, a2 + bv, a3 + bv};
        } else {
            if (gx < W) yp[0] = a0 + bv;
            if (gx + 1 < W) yp[1] = a1 + bv;
            if (gx + 2 < W) yp[2] = a2 + bv;
            if (gx + 3 < W) yp[3] = a3 + bv;
        }
    }
}

// weight/bias gradient: block = (tile column, group of tile rows, b*C + c).  Every thread keeps its K*K + 1 partial sums IN REGISTERS
// over the tiles of its group (4 pixels per tile) and the block reduces ONCE at the end: slab[b][column][group][c][T + 1].  (Round 5's
// form reduced per tile -- 49 wave reductions for 49 x 4 fma per thread: 93 us at 64 x 216 x 384 against ~10 us of traffic.)
// Finished by the deferred / immediate slab reduction (sum over blocks and batch).
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ slab,
                                                           int C, int H, int W, int K, int tiles_y, int rows_per_group) {
    __shared__ float s_x[DTH + 2 * PMAX][DTW + 2 * PMAX + 2];
    __shared__ float s_red[4][KMAX * KMAX + 1];
    const int P = K / 2, T = K * K;
    const int bc = blockIdx.z, c = bc % C, b = bc / C;
    const int x0 = blockIdx.x * DTW;
    const float* xp = x + (size_t)bc * H * W;
    const int RH = DTH + 2 * P, RW = DTW + 2 * P;
    const int ty = threadIdx.x >> 4, tx = (threadIdx.x & 15) * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[KMAX][KMAX], accb = 0.f;
#pragma unroll
    for (int ky = 0; ky < KMAX; ++ky)
#pragma unroll
        for (int kx = 0; kx < KMAX; ++kx) acc[ky][kx] = 0.f;
    const int t0 = blockIdx.y * rows_per_group, t1 = min(t0 + rows_per_group, tiles_y);
    for (int tr = t0; tr < t1; ++tr) {
        const int y0 = tr * DTH;
        if (tr > t0) __syncthreads();                            // the previous tile's window reads are done
        for (int i = threadIdx.x; i < RH * RW; i += 256) {
            const int r = i / RW, q = i - r * RW;
            const int gy = y0 + r - P, gx = x0 + q - P;
            s_x[r][q] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xp[(size_t)gy * W + gx] : 0.f;
        }
        const int gy = y0 + ty, gx = x0 + tx;
        float gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gv[i] = (gy < H && gx + i < W) ? g[(size_t)bc * H * W + (size_t)gy * W + gx + i] : 0.f;
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < KMAX; ++ky) {
            if (ky < K) {
                float row[4 + KMAX - 1];
#pragma unroll
                for (int i = 0; i < 4 + KMAX - 1; ++i) row[i] = i < 4 + K - 1 ? s_x[ty + ky][tx + i] : 0.f;
#pragma unroll
                for (int kx = 0; kx < KMAX; ++kx) {
                    if (kx < K) {
                        float pp = acc[ky][kx];
                        pp = fmaf(gv[0], row[kx], pp); pp = fmaf(gv[1], row[kx + 1], pp); pp = fmaf(gv[2], row[kx + 2], pp); pp = fmaf(gv[3], row[kx + 3], pp);
                        acc[ky][kx] = pp;
                    }
                }
            }
        }
        accb += (gv[0] + gv[1]) + (gv[2] + gv[3]);
    }
#pragma unroll
    for (int ky = 0; ky < KMAX; ++ky)
#pragma unroll
        for (int kx = 0; kx < KMAX; ++kx) {
            if (ky < K && kx < K) {
                const float pp = wave_sum(acc[ky][kx]);
                if (lane == 0) s_red[wave][ky * K + kx] = pp;
            }
        }
    const float sb = wave_sum(accb);
    if (lane == 0) s_red[wave][T] = sb;
    __syncthreads();
    if ((int)threadIdx.x <= T) {
        const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
        const size_t blk = ((size_t)b * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y;
        slab[(blk * C + c) * (T + 1) + threadIdx.x] = v;
    }
}

// groups of tile rows per tile column: enough blocks to fill the chip, as few reductions as that allows
static int dw_groups(int B, int C, int H, int W) {
    const int tx = cdiv(W, DTW), ty = cdiv(H, DTH);
    int gr = cdiv(1024, tx * B * C);
    gr = gr < 1 ? 1 : (gr > ty ? ty : gr);
    const int rows = cdiv(ty, gr);
    return cdiv(ty, rows);
}

}  // namespace

extern "C" int bnerv_dwconv_fwd(void* stream, const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int K, int flip) {
    BNERV_REQUIRE(x && w && y && B > 0 && C > 0 && H > 0 && W > 0, "dwconv_fwd: bad args");
    BNERV_REQUIRE(K >= 1 && K <= KMAX && (K & 1) == 1, "dwconv_fwd: K must be odd and <= %d (got %d)", KMAX, K);
    BNERV_REQUIRE((size_t)B * C <= 65535, "dwconv_fwd: B*C too large");
    dim3 grid(cdiv(W, DTW), cdiv(H, DTH), B * C);
    if (flip) hipLaunchKernelGGL(dwconv_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, C, H, W, K);
    else hipLaunchKernelGGL(dwconv_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, C, H, W, K);
    BNERV_LAUNCH_CHECK("dwconv");
    return BNERV_OK;
}

extern "C" size_t bnerv_dwconv_wgrad_ws_bytes(int B, int C, int H, int W, int K) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K < 1 || K > KMAX) return 0;
    return (size_t)B * cdiv(W, DTW) * cdiv(H, DTH) * C * (K * K + 1) * sizeof(float);
}

// dwb: [C][K*K + 1] (weight gradient rows with the bias gradient as the last column); `defer`: queue the slab reduction
extern "C" int bnerv_dwconv_wgrad(void* stream, const float* x, const float* g, float* dwb, void* ws, size_t ws_bytes, int B, int C, int H, int W, int K, bnerv_ctx* defer_ctx) {
    BNERV_REQUIRE(x && g && dwb && ws && B > 0 && C > 0 && H > 0 && W > 0, "dwconv_wgrad: bad args");
    BNERV_REQUIRE(K >= 1 && K <= KMAX && (K & 1) == 1, "dwconv_wgrad: K must be odd and <= %d (got %d)", KMAX, K);
    BNERV_REQUIRE((size_t)B * C <= 65535, "dwconv_wgrad: B*C too large");
    if (ws_bytes < bnerv_dwconv_wgrad_ws_bytes(B, C, H, W, K)) return bnerv_set_error(BNERV_E_WS, "dwconv_wgrad: workspace too small");
    const int tiles_x = cdiv(W, DTW), tiles_y = cdiv(H, DTH);
    const int groups = dw_groups(B, C, H, W), rows = cdiv(tiles_y, groups);
    const int n_slabs = B * tiles_x * groups;                   // (<= B * tiles: bnerv_dwconv_wgrad_ws_bytes covers it)
    dim3 grid(tiles_x, groups, B * C);
    hipLaunchKernelGGL(dwconv_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, g, (float*)ws, C, H, W, K, tiles_y, rows);
    BNERV_LAUNCH_CHECK("dwconv_wgrad");
    if (defer_ctx) {
        bnerv_side_push(defer_ctx, reinterpret_cast<hipStream_t>(stream), ws, n_slabs, C * (K * K + 1), 0, dwb, nullptr);
        return BNERV_OK;
    }
    return bnerv_reduce_slabs(stream, (const float*)ws, n_slabs, C * (K * K + 1), dwb);
}
