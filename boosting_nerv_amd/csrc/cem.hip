// cem.hip -- the per-step quantise + rate term of the CEM compression path, fused over all weight / bias tensors of the model
// (reference: model.cal_params(entropy_model), model_hnerv.py:292-303 = for every CustomConv2d / CustomLinear
//   Scale_T.forward  (lib/transform_ops.py:239-251)        code = w / s;  quant = ste(round(code));  dequant = quant * s
//   DiffEntropyModel.cal_global_bitrate (lib/entropy_model.py:21-43)   mu = mean(code), sigma = std(code) (unbiased), x = code + U(-.5,.5)
//        while training / x = quant otherwise;  bits = sum max(-log2(Phi((x+.5-mu)/sigma) - Phi((x-.5-mu)/sigma) + 1e-5), 0)
// and its backward, including the paths through mu and sigma and the reference's LowerBound gate).  Stock ops need ~40 launches
// per tensor per step (about 8000 for a 200-tensor model); here ONE block per tensor does each pass, 48 tensors per launch.
#include "common.h"

namespace {

constexpr float INV_LN2 = 1.44269504088896340736f;
constexpr float INV_SQRT2 = 0.70710678118654752440f;
constexpr float INV_SQRT2PI = 0.39894228040143267794f;

__device__ __forceinline__ double block_sum_d(double v, double* red) {        // 1024 threads = 16 waves
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    return t;
}

__device__ __forceinline__ float ncdf(float u) { return 0.5f * (1.0f + erff(u * INV_SQRT2)); }
__device__ __forceinline__ float npdf(float u) { return INV_SQRT2PI * expf(-0.5f * u * u); }

__global__ __launch_bounds__(1024) void cem_fwd_kernel(const bnerv_cem_chunk ck, float* __restrict__ stats) {
    __shared__ double red[16];
    const bnerv_cem_item it = ck.it[blockIdx.x];
    const int n = it.n;
    const float inv_s = 1.0f / it.scale[0], s = it.scale[0];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float c = it.w[i] * inv_s;
        s1 += (double)c;
        s2 += (double)c * (double)c;
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    const double mean_d = s1 / n;
    const double var_d = n > 1 ? fmax((s2 - (double)n * mean_d * mean_d) / (double)(n - 1), 0.0) : 0.0;
    const float mu = (float)mean_d, sigma = (float)sqrt(var_d);
    const float sg = fminf(fmaxf(sigma, 1e-5f), 1e10f), inv_sg = 1.0f / sg;
    double bits = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float c = it.w[i] * inv_s;
        const float q = rintf(c);                          // torch.round: half to even
        if (it.dequant) it.dequant[i] = q * s;
        const float x = ck.training ? c + it.noise[i] : q;
        const float p = ncdf((x + 0.5f - mu) * inv_sg) - ncdf((x - 0.5f - mu) * inv_sg);
        bits += (double)fmaxf(-logf(p + 1e-5f) * INV_LN2, 0.0f);
    }
    bits = block_sum_d(bits, red);
    if (threadIdx.x == 0) {
        float* o = stats + (size_t)(ck.first + blockIdx.x) * 4;
        o[0] = (float)bits; o[1] = mu; o[2] = sigma; o[3] = (float)n;
    }
}

// d bits / d code and d dequant / d {w, scale}:   upstream g = d L / d bits[item],  dd = d L / d dequant (may be NULL)
__global__ __launch_bounds__(1024) void cem_bwd_kernel(const bnerv_cem_chunk_bwd ck, const float* __restrict__ stats, const float* __restrict__ d_bits,
                                                       float* __restrict__ dscale) {
    __shared__ double red[16];
    const bnerv_cem_item_bwd it = ck.it[blockIdx.x];
    const int n = it.n, item = ck.first + blockIdx.x;
    const float s = it.scale[0], inv_s = 1.0f / s;
    const float mu = stats[(size_t)item * 4 + 1], sigma = stats[(size_t)item * 4 + 2];
    const bool sigma_free = sigma >= 1e-5f && sigma <= 1e10f;              // clamp passes the gradient only inside its range
    const float sg = fminf(fmaxf(sigma, 1e-5f), 1e10f), inv_sg = 1.0f / sg;
    const float g = d_bits ? d_bits[item] : 0.0f;
    double gmu = 0.0, gsg = 0.0;
    if (g != 0.0f) {
        for (int i = threadIdx.x; i < n; i += 1024) {
            const float c = it.w[i] * inv_s;
            const float x = ck.training ? c + it.noise[i] : rintf(c);
            const float up = (x + 0.5f - mu) * inv_sg, um = (x - 0.5f - mu) * inv_sg;
            const float p = ncdf(up) - ncdf(um);
            const float raw = -logf(p + 1e-5f) * INV_LN2;
            if (raw >= 0.0f || g < 0.0f) {                                 // LowerBound.backward (lib/entropy_model.py:108-114)
                const float pp = npdf(up), pm = npdf(um), k = -INV_LN2 * inv_sg / (p + 1e-5f);
                gmu += (double)(-k * (pp - pm));                           // d bits / d mu = - d bits / d x
                gsg += (double)(k * (-up * pp + um * pm));                 // d bits / d sigma
            }
        }
        gmu = block_sum_d(gmu, red);
        gsg = block_sum_d(gsg, red);
    }
    const float c_mu = (float)(gmu / n);
    const float c_sg = (sigma_free && n > 1 && sigma > 0.f) ? (float)(gsg / ((double)(n - 1) * (double)sigma)) : 0.0f;
    double ds = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float w = it.w[i], c = w * inv_s, q = rintf(c);
        float dc = 0.0f;
        if (g != 0.0f) {
            const float x = ck.training ? c + it.noise[i] : q;
            const float up = (x + 0.5f - mu) * inv_sg, um = (x - 0.5f - mu) * inv_sg;
            const float p = ncdf(up) - ncdf(um);
            const float raw = -logf(p + 1e-5f) * INV_LN2;
            float dbx = 0.0f;
            if (raw >= 0.0f || g < 0.0f) dbx = -INV_LN2 * inv_sg / (p + 1e-5f) * (npdf(up) - npdf(um));
            dc = g * (dbx + c_mu + c_sg * (c - mu));
        }
        const float dd = it.d_dequant ? it.d_dequant[i] : 0.0f;
        if (it.dw) it.dw[i] = dc * inv_s + dd;                             // d code / d w = 1/s;  d dequant / d w = 1 (STE)
        ds += (double)(dc * (-c * inv_s)) + (double)(dd * (q - c));        // d code / d s = -c/s; d dequant / d s = round(c) - c
    }
    ds = block_sum_d(ds, red);
    if (threadIdx.x == 0) dscale[item] = (float)ds;
}

}  // namespace

extern "C" int bnerv_cem_scale_fwd(void* stream, const bnerv_cem_chunk* ck, float* stats) {
    BNERV_REQUIRE(ck && stats && ck->n_items > 0 && ck->n_items <= BNERV_CEM_MAX_TENSORS, "cem_scale_fwd: bad chunk");
    for (int i = 0; i < ck->n_items; ++i)
        BNERV_REQUIRE(ck->it[i].w && ck->it[i].scale && ck->it[i].n > 0 && (!ck->training || ck->it[i].noise), "cem_scale_fwd: item %d incomplete", i);
    hipLaunchKernelGGL(cem_fwd_kernel, dim3(ck->n_items), dim3(1024), 0, (hipStream_t)stream, *ck, stats);
    BNERV_LAUNCH_CHECK("cem_fwd");
    return BNERV_OK;
}

extern "C" int bnerv_cem_scale_bwd(void* stream, const bnerv_cem_chunk_bwd* ck, const float* stats, const float* d_bits, float* dscale) {
    BNERV_REQUIRE(ck && stats && dscale && ck->n_items > 0 && ck->n_items <= BNERV_CEM_MAX_TENSORS, "cem_scale_bwd: bad chunk");
    for (int i = 0; i < ck->n_items; ++i)
        BNERV_REQUIRE(ck->it[i].w && ck->it[i].scale && ck->it[i].n > 0 && (!ck->training || ck->it[i].noise), "cem_scale_bwd: item %d incomplete", i);
    hipLaunchKernelGGL(cem_bwd_kernel, dim3(ck->n_items), dim3(1024), 0, (hipStream_t)stream, *ck, stats, d_bits, dscale);
    BNERV_LAUNCH_CHECK("cem_bwd");
    return BNERV_OK;
}
