// lnorm.hip -- LayerNorm over the channel axis of an NCHW tensor ("channels_first"), forward and backward, for the ConvNeXt
// encoder of HNeRV_Boost (reference: model_blocks.py:250-270 `LayerNorm`, both data formats compute
//   u = mean_c x;  s = mean_c (x - u)^2;  y = w[c] * (x - u) / sqrt(s + eps) + b[c];
// the channels_last form is F.layer_norm on the permuted tensor).  Row N3 of SURVEY 8(f): stock PyTorch runs the channels_first
// form as ~8 elementwise/reduce launches forward and ~20 backward, and the channels_last form behind a strided permute copy.
// Streaming, HBM-bound: one thread owns one pixel, its C (<= 64) channel values stay in registers, loads are coalesced along
// the pixel axis (128-B rows per channel plane).
//   backward:  g = dy * w;  dx = rstd * (g - mean_c g - xhat * mean_c (g * xhat));  dw[c] = sum_p dy * xhat;  db[c] = sum_p dy
//   (dw, db: one [2][C] slab per block, finished by bnerv_reduce_slabs -- fixed order, no atomics)
#include "common.h"

namespace {

constexpr int LN_CMAX = 64;

__global__ __launch_bounds__(256) void lncf_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ y, int C, int HW, float eps) {
    const int p = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (p >= HW) return;
    const float* xp = x + (size_t)n * C * HW + p;
    float v[LN_CMAX];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_CMAX; ++c) { v[c] = c < C ? xp[(size_t)c * HW] : 0.f; sum += v[c]; }
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_CMAX; ++c) { const float d = c < C ? v[c] - mean : 0.f; sq = fmaf(d, d, sq); }
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    float* yp = y + (size_t)n * C * HW + p;
#pragma unroll
    for (int c = 0; c < LN_CMAX; ++c)
        if (c < C) yp[(size_t)c * HW] = fmaf((v[c] - mean) * rstd, w[c], b[c]);
}

__global__ __launch_bounds__(256) void lncf_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                       float* __restrict__ dx, float* __restrict__ slab, int C, int HW, float eps) {
    __shared__ float s_red[4][2 * LN_CMAX];
    const int p = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    const bool live = p < HW;
    const size_t base = (size_t)n * C * HW + (live ? p : 0);
    float v[LN_CMAX];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_CMAX; ++c) { v[c] = (live && c < C) ? x[base + (size_t)c * HW] : 0.f; sum += v[c]; }
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_CMAX; ++c) { const float d = c < C ? v[c] - mean : 0.f; sq = fmaf(d, d, sq); }
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < LN_CMAX; ++c) {
        if (c < C) {                                         // uniform
            const float xh = (v[c] - mean) * rstd;
            const float g = live ? dy[base + (size_t)c * HW] : 0.f;
            const float gw = g * w[c];
            m1 += gw;
            m2 = fmaf(gw, xh, m2);
            const float sw = wave_sum(g * xh), sb = wave_sum(g);
            if (lane == 0) { s_red[wave][c] = sw; s_red[wave][LN_CMAX + c] = sb; }
            v[c] = gw;                                       // keep g*w; xhat is recomputed below from x (re-read: L2 hit)
        }
    }
    m1 /= (float)C;
    m2 /= (float)C;
    if (live) {
#pragma unroll
        for (int c = 0; c < LN_CMAX; ++c)
            if (c < C) {
                const float xh = (x[base + (size_t)c * HW] - mean) * rstd;
                dx[base + (size_t)c * HW] = rstd * (v[c] - m1 - xh * m2);
            }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * C) {
        const int q = threadIdx.x / C, c = threadIdx.x - q * C;
        const int i = q * LN_CMAX + c;
        slab[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + q) * C + c] = (s_red[0][i] + s_red[1][i]) + (s_red[2][i] + s_red[3][i]);
    }
}

}  // namespace

extern "C" int bnerv_lncf_fwd(void* stream, const float* x, const float* w, const float* b, float* y, int B, int C, int HW, float eps) {
    BNERV_REQUIRE(x && w && b && y && B > 0 && HW > 0, "lncf_fwd: bad args");
    BNERV_REQUIRE(C >= 1 && C <= LN_CMAX, "lncf_fwd: C must be in [1, %d] (got %d)", LN_CMAX, C);
    BNERV_REQUIRE(B <= 65535, "lncf_fwd: B too large");
    hipLaunchKernelGGL(lncf_fwd_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, x, w, b, y, C, HW, eps);
    BNERV_LAUNCH_CHECK("lncf_fwd");
    return BNERV_OK;
}

extern "C" size_t bnerv_lncf_bwd_ws_bytes(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)B * cdiv(HW, 256) * 2 * C * sizeof(float);
}

// dwb: [2][C] = (dw, db)
extern "C" int bnerv_lncf_bwd(void* stream, const float* x, const float* w, const float* dy, float* dx, float* dwb, void* ws, size_t ws_bytes,
                              int B, int C, int HW, float eps) {
    BNERV_REQUIRE(x && w && dy && dx && dwb && ws && B > 0 && HW > 0, "lncf_bwd: bad args");
    BNERV_REQUIRE(C >= 1 && C <= LN_CMAX, "lncf_bwd: C must be in [1, %d] (got %d)", LN_CMAX, C);
    BNERV_REQUIRE(B <= 65535, "lncf_bwd: B too large");
    if (ws_bytes < bnerv_lncf_bwd_ws_bytes(B, C, HW)) return bnerv_set_error(BNERV_E_WS, "lncf_bwd: workspace too small");
    const int nb = cdiv(HW, 256);
    hipLaunchKernelGGL(lncf_bwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, x, w, dy, dx, (float*)ws, C, HW, eps);
    BNERV_LAUNCH_CHECK("lncf_bwd");
    return bnerv_reduce_slabs(stream, (const float*)ws, B * nb, 2 * C, dwb);
}
