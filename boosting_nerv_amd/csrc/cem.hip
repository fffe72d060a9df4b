// cem.hip -- the per-step quantise + rate term of the CEM compression path, fused over all weight / bias tensors of the model
// (reference: model.cal_params(entropy_model), model_hnerv.py:292-303 = for every CustomConv2d / CustomLinear
//   Scale_T.forward  (lib/transform_ops.py:239-251)        code = w / s;  quant = ste(round(code));  dequant = quant * s
//   DiffEntropyModel.cal_global_bitrate (lib/entropy_model.py:21-43)   mu = mean(code), sigma = std(code) (unbiased), x = code + U(-.5,.5)
//        while training / x = quant otherwise;  bits = sum max(-log2(Phi((x+.5-mu)/sigma) - Phi((x-.5-mu)/sigma) + 1e-5), 0)
// and its backward, including the paths through mu and sigma and the reference's LowerBound gate).  Stock ops need ~40 launches
// per tensor per step (about 8000 for a 200-tensor model); here 48 tensors share a launch and every tensor is cut into chunks of
// 8192 elements, one 256-thread block each (grid = chunks of the largest tensor x tensors; blocks beyond a tensor's end return), so
// the 1.1 M-element stem matrix is spread over 135 blocks instead of one.  Each direction is three launches: chunk partial sums
// (f64) into a caller-provided workspace, the element pass (every block re-adds its tensor's partials in chunk order, so all blocks
// of a tensor see bit-identical statistics), and a per-tensor finalize -- fixed summation order, no atomics.
#include "common.h"

namespace {

constexpr float INV_LN2 = 1.44269504088896340736f;
constexpr float INV_SQRT2 = 0.70710678118654752440f;
constexpr float INV_SQRT2PI = 0.39894228040143267794f;

constexpr int CEM_CHUNK = 8192, CEM_THREADS = 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {        // 256 threads = 4 waves
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// sum over the chunks of one tensor of partial k (all threads get the same bits: lane l adds chunks l, l + 64, ..., then a wave tree)
__device__ __forceinline__ double chunk_total(const double* part, int nchunks, int k, double* red) {
    double t = 0.0;
    if (threadIdx.x < 64) {
        for (int c = threadIdx.x; c < nchunks; c += 64) t += part[(size_t)c * 3 + k];
        t = wave_sum_d(t);
        if (threadIdx.x == 0) red[4] = t;
    }
    __syncthreads();
    t = red[4];
    __syncthreads();
    return t;
}

__device__ __forceinline__ float ncdf(float u) { return 0.5f * (1.0f + erff(u * INV_SQRT2)); }
__device__ __forceinline__ float npdf(float u) { return INV_SQRT2PI * expf(-0.5f * u * u); }

// ws: [item][max_chunks][3] doubles.  forward: {sum code, sum code^2, bits};  backward: {d bits / d mu, d bits / d sigma, d scale}
__global__ __launch_bounds__(CEM_THREADS) void cem_fwd_a_kernel(const bnerv_cem_chunk ck, double* __restrict__ ws, int max_chunks) {
    __shared__ double red[5];
    const bnerv_cem_item it = ck.it[blockIdx.y];
    const int i0 = blockIdx.x * CEM_CHUNK, i1 = min(it.n, i0 + CEM_CHUNK);
    if (i0 >= it.n) return;
    const float s_ = it.scale[0];                             // true division below: code = w / s must round exactly as torch's (round() sits on it)
    double s1 = 0.0, s2 = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += CEM_THREADS) {
        const float c = it.w[i] / s_;
        s1 += (double)c;
        s2 += (double)c * (double)c;
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        double* o = ws + ((size_t)blockIdx.y * max_chunks + blockIdx.x) * 3;
        o[0] = s1; o[1] = s2;
    }
}

__device__ __forceinline__ void cem_moments(const double* part, int nchunks, int n, double* red, float& mu, float& sigma) {
    const double s1 = chunk_total(part, nchunks, 0, red), s2 = chunk_total(part, nchunks, 1, red);
    const double mean_d = s1 / n;
    const double var_d = n > 1 ? fmax((s2 - (double)n * mean_d * mean_d) / (double)(n - 1), 0.0) : 0.0;
    mu = (float)mean_d;
    sigma = (float)sqrt(var_d);
}

__global__ __launch_bounds__(CEM_THREADS) void cem_fwd_b_kernel(const bnerv_cem_chunk ck, double* __restrict__ ws, int max_chunks, float* __restrict__ stats) {
    __shared__ double red[5];
    const bnerv_cem_item it = ck.it[blockIdx.y];
    const int n = it.n, i0 = blockIdx.x * CEM_CHUNK, i1 = min(n, i0 + CEM_CHUNK);
    if (i0 >= n) return;
    double* part = ws + (size_t)blockIdx.y * max_chunks * 3;
    float mu, sigma;
    cem_moments(part, (n + CEM_CHUNK - 1) / CEM_CHUNK, n, red, mu, sigma);
    const float s = it.scale[0];
    const float sg = fminf(fmaxf(sigma, 1e-5f), 1e10f), inv_sg = 1.0f / sg;
    double bits = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += CEM_THREADS) {
        const float c = it.w[i] / s;
        const float q = rintf(c);                          // torch.round: half to even
        if (it.dequant) it.dequant[i] = q * s;
        const float x = ck.training ? c + it.noise[i] : q;
        const float p = ncdf((x + 0.5f - mu) * inv_sg) - ncdf((x - 0.5f - mu) * inv_sg);
        bits += (double)fmaxf(-logf(p + 1e-5f) * INV_LN2, 0.0f);
    }
    bits = block_sum_d(bits, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * 3 + 2] = bits;
        if (blockIdx.x == 0) {
            float* o = stats + (size_t)(ck.first + blockIdx.y) * 4;
            o[1] = mu; o[2] = sigma; o[3] = (float)n;
        }
    }
}

// out[item * stride] = sum over the tensor's chunks of partial 2 (forward: bits -> stats[item][0]; backward: d scale -> dscale[item])
__global__ __launch_bounds__(64) void cem_finalize_kernel(const double* __restrict__ ws, int max_chunks, bnerv_cem_chunk ck_sizes, float* __restrict__ out,
                                                         int first, int stride) {
    const int item = blockIdx.x;
    const int nchunks = (ck_sizes.it[item].n + CEM_CHUNK - 1) / CEM_CHUNK;
    const double* part = ws + (size_t)item * max_chunks * 3;
    double t = 0.0;
    for (int c = threadIdx.x; c < nchunks; c += 64) t += part[(size_t)c * 3 + 2];
    t = wave_sum_d(t);
    if (threadIdx.x == 0) out[(size_t)(first + item) * stride] = (float)t;
}

// d bits / d code and d dequant / d {w, scale}:   upstream g = d L / d bits[item],  dd = d L / d dequant (may be NULL)
__global__ __launch_bounds__(CEM_THREADS) void cem_bwd_a_kernel(const bnerv_cem_chunk_bwd ck, const float* __restrict__ stats, const float* __restrict__ d_bits,
                                                                double* __restrict__ ws, int max_chunks) {
    __shared__ double red[5];
    const bnerv_cem_item_bwd it = ck.it[blockIdx.y];
    const int n = it.n, item = ck.first + blockIdx.y, i0 = blockIdx.x * CEM_CHUNK, i1 = min(n, i0 + CEM_CHUNK);
    if (i0 >= n) return;
    const float s_ = it.scale[0];
    const float mu = stats[(size_t)item * 4 + 1], sigma = stats[(size_t)item * 4 + 2];
    const float sg = fminf(fmaxf(sigma, 1e-5f), 1e10f), inv_sg = 1.0f / sg;
    const float g = d_bits ? d_bits[item] : 0.0f;
    double gmu = 0.0, gsg = 0.0;
    if (g != 0.0f) {                                        // block-uniform
        for (int i = i0 + threadIdx.x; i < i1; i += CEM_THREADS) {
            const float c = it.w[i] / s_;
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
    if (threadIdx.x == 0) {
        double* o = ws + ((size_t)blockIdx.y * max_chunks + blockIdx.x) * 3;
        o[0] = gmu; o[1] = gsg;
    }
}

__global__ __launch_bounds__(CEM_THREADS) void cem_bwd_b_kernel(const bnerv_cem_chunk_bwd ck, const float* __restrict__ stats, const float* __restrict__ d_bits,
                                                                double* __restrict__ ws, int max_chunks) {
    __shared__ double red[5];
    const bnerv_cem_item_bwd it = ck.it[blockIdx.y];
    const int n = it.n, item = ck.first + blockIdx.y, i0 = blockIdx.x * CEM_CHUNK, i1 = min(n, i0 + CEM_CHUNK);
    if (i0 >= n) return;
    double* part = ws + (size_t)blockIdx.y * max_chunks * 3;
    const int nchunks = (n + CEM_CHUNK - 1) / CEM_CHUNK;
    const double gmu = chunk_total(part, nchunks, 0, red), gsg = chunk_total(part, nchunks, 1, red);
    const float s = it.scale[0];
    const float mu = stats[(size_t)item * 4 + 1], sigma = stats[(size_t)item * 4 + 2];
    const bool sigma_free = sigma >= 1e-5f && sigma <= 1e10f;              // clamp passes the gradient only inside its range
    const float sg = fminf(fmaxf(sigma, 1e-5f), 1e10f), inv_sg = 1.0f / sg;
    const float g = d_bits ? d_bits[item] : 0.0f;
    const float c_mu = (float)(gmu / n);
    const float c_sg = (sigma_free && n > 1 && sigma > 0.f) ? (float)(gsg / ((double)(n - 1) * (double)sigma)) : 0.0f;
    double ds = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += CEM_THREADS) {
        const float w = it.w[i], c = w / s, q = rintf(c);
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
        if (it.dw) it.dw[i] = dc / s + dd;                             // d code / d w = 1/s;  d dequant / d w = 1 (STE)
        ds += (double)(dc * (-c / s)) + (double)(dd * (q - c));        // d code / d s = -c/s; d dequant / d s = round(c) - c
    }
    ds = block_sum_d(ds, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * 3 + 2] = ds;
}

}  // namespace

static int cem_max_chunks(const int* ns, int n_items) {
    int m = 1;
    for (int i = 0; i < n_items; ++i) m = ns[i] > m ? ns[i] : m;
    return (m + CEM_CHUNK - 1) / CEM_CHUNK;
}

// workspace of one launch group: n_items tensors, the largest of max_n elements
extern "C" size_t bnerv_cem_ws_bytes(int n_items, int max_n) {
    if (n_items <= 0 || max_n <= 0) return 0;
    return (size_t)n_items * ((max_n + CEM_CHUNK - 1) / CEM_CHUNK) * 3 * sizeof(double);
}

extern "C" int bnerv_cem_scale_fwd(void* stream, const bnerv_cem_chunk* ck, float* stats, void* ws, size_t ws_bytes) {
    BNERV_REQUIRE(ck && stats && ws && ck->n_items > 0 && ck->n_items <= BNERV_CEM_MAX_TENSORS, "cem_scale_fwd: bad chunk");
    int ns[BNERV_CEM_MAX_TENSORS];
    for (int i = 0; i < ck->n_items; ++i) {
        BNERV_REQUIRE(ck->it[i].w && ck->it[i].scale && ck->it[i].n > 0 && (!ck->training || ck->it[i].noise), "cem_scale_fwd: item %d incomplete", i);
        ns[i] = ck->it[i].n;
    }
    const int mc = cem_max_chunks(ns, ck->n_items);
    const size_t need = (size_t)ck->n_items * mc * 3 * sizeof(double);
    if (ws_bytes < need) return bnerv_set_error(BNERV_E_WS, "cem_scale_fwd: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    double* wd = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(cem_fwd_a_kernel, dim3(mc, ck->n_items), dim3(CEM_THREADS), 0, st, *ck, wd, mc);
    hipLaunchKernelGGL(cem_fwd_b_kernel, dim3(mc, ck->n_items), dim3(CEM_THREADS), 0, st, *ck, wd, mc, stats);
    hipLaunchKernelGGL(cem_finalize_kernel, dim3(ck->n_items), dim3(64), 0, st, wd, mc, *ck, stats, ck->first, 4);
    BNERV_LAUNCH_CHECK("cem_fwd");
    return BNERV_OK;
}

extern "C" int bnerv_cem_scale_bwd(void* stream, const bnerv_cem_chunk_bwd* ck, const float* stats, const float* d_bits, float* dscale,
                                   void* ws, size_t ws_bytes) {
    BNERV_REQUIRE(ck && stats && dscale && ws && ck->n_items > 0 && ck->n_items <= BNERV_CEM_MAX_TENSORS, "cem_scale_bwd: bad chunk");
    int ns[BNERV_CEM_MAX_TENSORS];
    bnerv_cem_chunk sizes{};                               // (the finalize kernel only reads the element counts)
    for (int i = 0; i < ck->n_items; ++i) {
        BNERV_REQUIRE(ck->it[i].w && ck->it[i].scale && ck->it[i].n > 0 && (!ck->training || ck->it[i].noise), "cem_scale_bwd: item %d incomplete", i);
        ns[i] = ck->it[i].n;
        sizes.it[i].n = ck->it[i].n;
    }
    sizes.n_items = ck->n_items;
    const int mc = cem_max_chunks(ns, ck->n_items);
    const size_t need = (size_t)ck->n_items * mc * 3 * sizeof(double);
    if (ws_bytes < need) return bnerv_set_error(BNERV_E_WS, "cem_scale_bwd: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    double* wd = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(cem_bwd_a_kernel, dim3(mc, ck->n_items), dim3(CEM_THREADS), 0, st, *ck, stats, d_bits, wd, mc);
    hipLaunchKernelGGL(cem_bwd_b_kernel, dim3(mc, ck->n_items), dim3(CEM_THREADS), 0, st, *ck, stats, d_bits, wd, mc);
    hipLaunchKernelGGL(cem_finalize_kernel, dim3(ck->n_items), dim3(64), 0, st, wd, mc, sizes, dscale, ck->first, 1);
    BNERV_LAUNCH_CHECK("cem_bwd");
    return BNERV_OK;
}
