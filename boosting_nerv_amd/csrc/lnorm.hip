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

// Backward: block = 64 pixels x 4 channel quarters (wave q owns channels [q CQ, (q + 1) CQ)): every load / store of a wave is one
// 256-B row segment of a channel plane, 4x the blocks of a thread-per-pixel layout keep enough loads in flight, x stays in
// registers for the dx pass (no re-read), the per-pixel statistics cross the four waves through LDS and the per-channel sums over
// pixels are plain wave reductions (a channel belongs to one wave).
constexpr int LNB_PX = 64;
__global__ __launch_bounds__(256) void lncf_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                       float* __restrict__ dx, float* __restrict__ slab, int C, int HW, float eps) {
    constexpr int CQM = LN_CMAX / 4;
    __shared__ float s_st[2][4][LNB_PX];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p = blockIdx.x * LNB_PX + lane, n = blockIdx.y;
    const bool live = p < HW;
    const int CQ = (C + 3) >> 2, c0 = q * CQ;
    const size_t base = (size_t)n * C * HW + (live ? p : 0);
    float v[CQM], gw[CQM];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < CQM; ++j) {
        const int c = c0 + j;
        v[j] = (live && j < CQ && c < C) ? x[base + (size_t)c * HW] : 0.f;
        sum += v[j];
    }
    s_st[0][q][lane] = sum;
    __syncthreads();
    const float mean = ((s_st[0][0][lane] + s_st[0][1][lane]) + (s_st[0][2][lane] + s_st[0][3][lane])) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < CQM; ++j) { const float d = (j < CQ && c0 + j < C) ? v[j] - mean : 0.f; sq = fmaf(d, d, sq); }
    s_st[1][q][lane] = sq;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((s_st[1][0][lane] + s_st[1][1][lane]) + (s_st[1][2][lane] + s_st[1][3][lane])) / (float)C + eps);
    __syncthreads();                                       // (s_st is reused below)
    float m1 = 0.f, m2 = 0.f;
    float* sl = slab + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
    for (int j = 0; j < CQM; ++j) {
        const int c = c0 + j;
        if (j < CQ && c < C) {                               // wave-uniform
            const float xh = (v[j] - mean) * rstd;
            const float g = live ? dy[base + (size_t)c * HW] : 0.f;
            gw[j] = g * w[c];
            m1 += gw[j];
            m2 = fmaf(gw[j], xh, m2);
            const float sw = wave_sum(g * xh), sb = wave_sum(g);
            if (lane == 0) { sl[c] = sw; sl[C + c] = sb; }
            v[j] = xh;
        } else {
            gw[j] = 0.f;
        }
    }
    s_st[0][q][lane] = m1;
    s_st[1][q][lane] = m2;
    __syncthreads();
    m1 = ((s_st[0][0][lane] + s_st[0][1][lane]) + (s_st[0][2][lane] + s_st[0][3][lane])) / (float)C;
    m2 = ((s_st[1][0][lane] + s_st[1][1][lane]) + (s_st[1][2][lane] + s_st[1][3][lane])) / (float)C;
    if (live) {
#pragma unroll
        for (int j = 0; j < CQM; ++j) {
            const int c = c0 + j;
            if (j < CQ && c < C) dx[base + (size_t)c * HW] = rstd * (gw[j] - m1 - v[j] * m2);
        }
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
    return (size_t)B * cdiv(HW, LNB_PX) * 2 * C * sizeof(float);
}

// dwb: [2][C] = (dw, db)
extern "C" int bnerv_lncf_bwd(void* stream, const float* x, const float* w, const float* dy, float* dx, float* dwb, void* ws, size_t ws_bytes,
                              int B, int C, int HW, float eps) {
    BNERV_REQUIRE(x && w && dy && dx && dwb && ws && B > 0 && HW > 0, "lncf_bwd: bad args");
    BNERV_REQUIRE(C >= 1 && C <= LN_CMAX, "lncf_bwd: C must be in [1, %d] (got %d)", LN_CMAX, C);
    BNERV_REQUIRE(B <= 65535, "lncf_bwd: B too large");
    if (ws_bytes < bnerv_lncf_bwd_ws_bytes(B, C, HW)) return bnerv_set_error(BNERV_E_WS, "lncf_bwd: workspace too small");
    const int nb = cdiv(HW, LNB_PX);
    hipLaunchKernelGGL(lncf_bwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, x, w, dy, dx, (float*)ws, C, HW, eps);
    BNERV_LAUNCH_CHECK("lncf_bwd");
    return bnerv_reduce_slabs(stream, (const float*)ws, B * nb, 2 * C, dwb);
}
