// optim.hip -- fused multi-tensor Adan step and the flat-bucket gather/scatter used by the data-parallel exchange.
//
// Adan semantics follow the reference's default path exactly (optimizer.py:296-362 `_multi_tensor_adan`, called from
// Adan.step optimizer.py:125-235 with betas (0.98, 0.92, 0.99), eps 1e-8, weight_decay 0, no clipping), one pass over
// p, g, exp_avg, exp_avg_sq, exp_avg_diff, neg_pre_grad = 6 reads + 5 writes of 4 B per parameter (44 B/param),
// instead of ~17 torch._foreach launches.  The operation ORDER of the reference is kept so the result matches the
// foreach path to fp32 rounding:
//     g *= clip;  t = neg_pre_grad + g (= g - g_prev);  m = m*b1 + (1-b1) g;  d = d*b2 + (1-b2) t;
//     t = t*b2 + g;  n = n*b3 + (1-b3) t*t;  denom = sqrt(n)/sqrt(bc3) + eps;
//     p -= (lr/bc1) m/denom;  p -= (lr*b2/bc2) d/denom;  p /= (1 + lr*wd);  neg_pre_grad = -g
#include "common.h"
#include "sidejob.h"

namespace {
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// bstart[t] = first block of tensor t in the flat grid (tensor t owns min(ceil(n_t / 1024), 1024) blocks): a (blocks of the largest
// tensor) x (tensors) grid launched 48 x 1024 blocks for the chunk that holds the 1.1 M-element stem matrix, ~1500 of them with work
struct AdanArgs { bnerv_adan_chunk c; bnerv_adan_hyper h; int bstart[BNERV_ADAN_MAX_TENSORS + 1]; };

__global__ __launch_bounds__(256) void adan_kernel(const AdanArgs a) {
    int t = 0;
    for (int hi = a.c.n_tensors; hi - t > 1;) {            // (uniform: scalar loads from the kernel arguments)
        const int mid = (t + hi) >> 1;
        if ((int)blockIdx.x >= a.bstart[mid]) t = mid; else hi = mid;
    }
    const int bx = (int)blockIdx.x - a.bstart[t], gx = a.bstart[t + 1] - a.bstart[t];
    const int n = a.c.n[t];
    float* __restrict__ p = a.c.p[t];
    const float* __restrict__ g = a.c.g[t];
    float* __restrict__ m = a.c.exp_avg[t];
    float* __restrict__ v = a.c.exp_avg_sq[t];
    float* __restrict__ d = a.c.exp_avg_diff[t];
    float* __restrict__ ng = a.c.neg_pre_grad[t];
    const float lr = a.h.sched_dev[0], bc1 = a.h.sched_dev[1], bc2 = a.h.sched_dev[2], bc3s = a.h.sched_dev[3];
    const bool first = a.h.sched_dev[4] != 0.f;
    const float b1 = a.h.beta1, b2 = a.h.beta2, b3 = a.h.beta3, eps = a.h.eps, wd = a.h.weight_decay, clip = a.h.clip_global_grad_norm;
    const float step_size = lr / bc1, step_size_diff = lr * b2 / bc2;
    for (int i = bx * 256 + threadIdx.x; i < n; i += gx * 256) {
        const float gi = g[i] * clip;
        float t0 = (first ? -gi : ng[i]) + gi;                       // g - g_prev   (0 on the first step)
        const float mi = m[i] * b1 + (1.0f - b1) * gi;
        const float di = d[i] * b2 + (1.0f - b2) * t0;
        t0 = t0 * b2 + gi;
        const float vi = v[i] * b3 + (1.0f - b3) * (t0 * t0);
        const float denom = sqrtf(vi) / bc3s + eps;
        float pi = p[i];
        if (a.h.no_prox) {
            pi *= (1.0f - lr * wd);
            pi += -step_size * (mi / denom);
            pi += -step_size_diff * (di / denom);
        } else {
            pi += -step_size * (mi / denom);
            pi += -step_size_diff * (di / denom);
            pi /= (1.0f + lr * wd);
        }
        p[i] = pi; m[i] = mi; v[i] = vi; d[i] = di; ng[i] = -gi;
    }
}

__device__ __forceinline__ void adan_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                            float* __restrict__ d, float* __restrict__ ng, int n, int bx, int gx, const bnerv_adan_hyper& h) {
    const float lr = h.sched_dev[0], bc1 = h.sched_dev[1], bc2 = h.sched_dev[2], bc3s = h.sched_dev[3];
    const bool first = h.sched_dev[4] != 0.f;
    const float b1 = h.beta1, b2 = h.beta2, b3 = h.beta3, eps = h.eps, wd = h.weight_decay, clip = h.clip_global_grad_norm;
    const float step_size = lr / bc1, step_size_diff = lr * b2 / bc2;
    for (int i = bx * 256 + threadIdx.x; i < n; i += gx * 256) {
        const float gi = g[i] * clip;
        float t0 = (first ? -gi : ng[i]) + gi;                       // g - g_prev   (0 on the first step)
        const float mi = m[i] * b1 + (1.0f - b1) * gi;
        const float di = d[i] * b2 + (1.0f - b2) * t0;
        t0 = t0 * b2 + gi;
        const float vi = v[i] * b3 + (1.0f - b3) * (t0 * t0);
        const float denom = sqrtf(vi) / bc3s + eps;
        float pi = p[i];
        if (h.no_prox) {
            pi *= (1.0f - lr * wd);
            pi += -step_size * (mi / denom);
            pi += -step_size_diff * (di / denom);
        } else {
            pi += -step_size * (mi / denom);
            pi += -step_size_diff * (di / denom);
            pi /= (1.0f + lr * wd);
        }
        p[i] = pi; m[i] = mi; v[i] = vi; d[i] = di; ng[i] = -gi;
    }
}

// table form: the tensor of a block is found by bisection over bstart in device memory (uniform addresses: scalar loads)
__global__ __launch_bounds__(256) void adan_table_kernel(const bnerv_adan_entry* __restrict__ tab, const int n_tensors, const bnerv_adan_hyper h) {
    int t = 0;
    for (int hi = n_tensors; hi - t > 1;) {
        const int mid = (t + hi) >> 1;
        if ((int)blockIdx.x >= tab[mid].bstart) t = mid; else hi = mid;
    }
    const bnerv_adan_entry e = tab[t];
    const int gx = min(cdiv_dev(e.n, 256 * 4), 1024);
    adan_update(e.p, e.g, e.exp_avg, e.exp_avg_sq, e.exp_avg_diff, e.neg_pre_grad, e.n, (int)blockIdx.x - e.bstart, gx, h);
}

__global__ __launch_bounds__(256) void fetch_frame_kernel(const float* __restrict__ clip, const double* __restrict__ norms, const float* __restrict__ sel,
                                                          const int n_frames, const size_t frame_elems, float* __restrict__ dst, double* __restrict__ dst_norm) {
    fetch_frame_body(clip, norms, sel, n_frames, frame_elems, dst, dst_norm, (int)blockIdx.x, (int)gridDim.x);
}

struct BucketArgs { bnerv_bucket_chunk c; float* bucket; float scale; int to_bucket; int bstart[BNERV_ADAN_MAX_TENSORS * 2 + 1]; };

__global__ __launch_bounds__(256) void bucket_kernel(const BucketArgs a) {
    int t = 0;                                             // flat grid, as adan_kernel
    for (int hi = a.c.n_tensors; hi - t > 1;) {
        const int mid = (t + hi) >> 1;
        if ((int)blockIdx.x >= a.bstart[mid]) t = mid; else hi = mid;
    }
    const int bx = (int)blockIdx.x - a.bstart[t], gx = a.bstart[t + 1] - a.bstart[t];
    const int n = a.c.n[t];
    float* __restrict__ x = a.c.t[t];
    float* __restrict__ bk = a.bucket + a.c.off[t];
    for (int i = bx * 256 + threadIdx.x; i < n; i += gx * 256) {
        if (a.to_bucket) bk[i] = x[i] * a.scale;
        else x[i] = bk[i] * a.scale;
    }
}

}  // namespace

extern "C" int bnerv_adan_multi_tensor(void* stream, const bnerv_adan_chunk* chunk, const bnerv_adan_hyper* h) {
    BNERV_REQUIRE(chunk && h && h->sched_dev, "adan_multi_tensor: null args");
    BNERV_REQUIRE(chunk->n_tensors > 0 && chunk->n_tensors <= BNERV_ADAN_MAX_TENSORS, "adan_multi_tensor: n_tensors=%d", chunk->n_tensors);
    AdanArgs a;
    a.c = *chunk;
    a.h = *h;
    int blocks = 0;
    for (int i = 0; i < chunk->n_tensors; ++i) {
        BNERV_REQUIRE(chunk->p[i] && chunk->g[i] && chunk->exp_avg[i] && chunk->exp_avg_sq[i] && chunk->exp_avg_diff[i] && chunk->neg_pre_grad[i] && chunk->n[i] > 0,
                      "adan_multi_tensor: bad tensor %d", i);
        int gx = cdiv(chunk->n[i], 256 * 4);
        if (gx > 1024) gx = 1024;
        a.bstart[i] = blocks;
        blocks += gx;
    }
    for (int i = chunk->n_tensors; i <= BNERV_ADAN_MAX_TENSORS; ++i) a.bstart[i] = blocks;
    hipLaunchKernelGGL(adan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    BNERV_LAUNCH_CHECK("adan");
    return BNERV_OK;
}

extern "C" int bnerv_adan_table_blocks(int n) {
    if (n <= 0) return 0;
    const int gx = cdiv(n, 256 * 4);
    return gx > 1024 ? 1024 : gx;
}

extern "C" int bnerv_adan_table(void* stream, const bnerv_adan_entry* table_dev, int n_tensors, int total_blocks, const bnerv_adan_hyper* h) {
    BNERV_REQUIRE(table_dev && h && h->sched_dev, "adan_table: null args");
    BNERV_REQUIRE(n_tensors > 0 && total_blocks > 0, "adan_table: n_tensors=%d total_blocks=%d", n_tensors, total_blocks);
    hipLaunchKernelGGL(adan_table_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n_tensors, *h);
    BNERV_LAUNCH_CHECK("adan_table");
    return BNERV_OK;
}

extern "C" int bnerv_fetch_frame(void* stream, const float* clip, const double* norms, const float* sel_dev, int n_frames, size_t frame_elems,
                                 float* dst_img, double* dst_norm) {
    BNERV_REQUIRE(clip && sel_dev && dst_img && n_frames > 0 && frame_elems > 0, "fetch_frame: bad args");
    BNERV_REQUIRE((reinterpret_cast<uintptr_t>(clip) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_img) & 15) == 0 && (frame_elems % 4 == 0 || n_frames == 1),
                  "fetch_frame: 16-byte aligned frames required");
    const size_t n4 = frame_elems / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fetch_frame_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, clip, norms, sel_dev, n_frames, frame_elems, dst_img, dst_norm);
    BNERV_LAUNCH_CHECK("fetch_frame");
    return BNERV_OK;
}

static int bucket_launch(void* stream, const bnerv_bucket_chunk* c, float* bucket, float scale, int to_bucket) {
    BNERV_REQUIRE(c && bucket && c->n_tensors > 0 && c->n_tensors <= BNERV_ADAN_MAX_TENSORS * 2, "bucket: bad args");
    BucketArgs a;
    a.c = *c; a.bucket = bucket; a.scale = scale; a.to_bucket = to_bucket;
    int blocks = 0;
    for (int i = 0; i < c->n_tensors; ++i) {
        BNERV_REQUIRE(c->t[i] && c->n[i] > 0 && c->off[i] >= 0, "bucket: bad tensor %d", i);
        int gx = cdiv(c->n[i], 256 * 4);
        if (gx > 1024) gx = 1024;
        a.bstart[i] = blocks;
        blocks += gx;
    }
    for (int i = c->n_tensors; i <= BNERV_ADAN_MAX_TENSORS * 2; ++i) a.bstart[i] = blocks;
    hipLaunchKernelGGL(bucket_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    BNERV_LAUNCH_CHECK("bucket");
    return BNERV_OK;
}

extern "C" int bnerv_bucket_gather(void* stream, const bnerv_bucket_chunk* c, float* bucket, float scale) {
    return bucket_launch(stream, c, bucket, scale, 1);
}
extern "C" int bnerv_bucket_scatter(void* stream, const bnerv_bucket_chunk* c, const float* bucket, float scale) {
    return bucket_launch(stream, c, const_cast<float*>(bucket), scale, 0);
}
