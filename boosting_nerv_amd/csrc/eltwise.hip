// eltwise.hip -- streaming helpers of the output head's backward.
//
// bnerv_tanh_grad: gt = g * d/dv [tanh(v) * 0.5 + 0.5] written out once, with its per-channel sums (the head's bias gradient) as per-block
// partials.  OutImg (reference model_blocks.py:57-63) maps the head conv's output v to img = tanh(v) * 0.5 + 0.5, so with t = 2 img - 1 the
// factor is 0.5 (1 - t^2) -- the same expression as the conv kernels' IN_TANHGRAD prologue (conv_common.h xform1).  HNeRV_Boost's 3x3
// head (38 -> 3, model_hnerv.py:214) takes its weight gradient with the roles of input and gradient SWAPPED (ops._HeadTanh.backward: M = the
// 38 input channels, N = 3 couts x 9 taps instead of 3 of 16 MFMA rows), which needs gt as a plain tensor.
#include "common.h"

namespace {
constexpr int TG_PER_BLOCK = 8192;      // elements of one plane per block (256 threads x 8 float4)

__global__ __launch_bounds__(256) void tanh_grad_kernel(const float* __restrict__ g, const float* __restrict__ img, float* __restrict__ gt,
                                                        float* __restrict__ part, int C, int HW, int nblk) {
    __shared__ float s_red[4];
    const int bc = blockIdx.y, blk = blockIdx.x;
    const size_t base = (size_t)bc * HW;
    const int e0 = blk * TG_PER_BLOCK;
    float acc = 0.f;
    const bool vec = (HW & 3) == 0;
    if (vec) {
#pragma unroll
        for (int u = 0; u < TG_PER_BLOCK / 1024; ++u) {
            const int e = e0 + (u * 256 + (int)threadIdx.x) * 4;
            if (e < HW) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + base + e), iv = *reinterpret_cast<const f32x4*>(img + base + e);
                f32x4 r;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float t = 2.0f * iv[k] - 1.0f; r[k] = gv[k] * 0.5f * (1.0f - t * t); }
                *reinterpret_cast<f32x4*>(gt + base + e) = r;
                acc += (r[0] + r[1]) + (r[2] + r[3]);
            }
        }
    } else {
        for (int e = e0 + (int)threadIdx.x; e < min(HW, e0 + TG_PER_BLOCK); e += 256) {
            const float t = 2.0f * img[base + e] - 1.0f;
            const float r = g[base + e] * 0.5f * (1.0f - t * t);
            gt[base + e] = r;
            acc += r;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int b = bc / C, c = bc - b * C;
        part[((size_t)b * nblk + blk) * C + c] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}
}  // namespace

extern "C" int bnerv_tanh_grad_blocks(int HW) { return HW > 0 ? cdiv(HW, TG_PER_BLOCK) : 0; }

// gt [B, C, HW] = g * 0.5 (1 - (2 img - 1)^2); part [B * bnerv_tanh_grad_blocks(HW)][C]: per-block channel sums of gt, to be summed over
// their first index (bnerv_reduce_slabs / _deferred with n_slabs = B * blocks, count = C) into the bias gradient
extern "C" int bnerv_tanh_grad(void* stream, const float* g, const float* img, float* gt, float* part, int B, int C, int HW) {
    BNERV_REQUIRE(g && img && gt && part && B > 0 && C > 0 && HW > 0, "tanh_grad: bad args");
    BNERV_REQUIRE((size_t)B * C <= 65535, "tanh_grad: B * C too large");
    if ((HW & 3) == 0) BNERV_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(gt)) & 15) == 0, "tanh_grad: tensors must be 16-byte aligned");
    const int nblk = cdiv(HW, TG_PER_BLOCK);
    hipLaunchKernelGGL(tanh_grad_kernel, dim3(nblk, B * C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, img, gt, part, C, HW, nblk);
    BNERV_LAUNCH_CHECK("tanh_grad");
    return BNERV_OK;
}
