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
