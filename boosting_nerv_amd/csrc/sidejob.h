// sidejob.h -- deferred slab reductions ("side jobs").
//
// Measured on MI355X (tools/kconc.py): a tiny dependent kernel between two big ones costs ~10 us of pipeline (drain, launch,
// ramp), far more than its own work, and parallel graph branches do not overlap here.  The ~45 slab reductions of a train
// step (weight-gradient slabs -> dw/db; per-tile epilogue sums -> [B][2][C]) are therefore not launched: they are QUEUED, and
// the next lean conv / weight-gradient launch executes them as slices at the END of its blocks, least-loaded blocks first
// (with a static tile partition about half of the blocks own one tile less and would idle at the tail).  bnerv_flush_deferred
// runs whatever is still queued in one standalone launch; every consumer-side entry point of the library flushes first.
//
// One job:  out[i] = sum_{k < n_slabs} src[k * count + i]   (fixed order: `lanes` interleaved partial sums, then a sequential
// combine -- deterministic and independent of who executes the slice).  ncols > 0 selects the weight-gradient output split:
// element i = co * ncols + n goes to out[co * (ncols - 1) + n] for n < ncols - 1 and to out2[co] for the bias column.
#pragma once
#include "common.h"

constexpr int SIDE_MAX_JOBS = 4;

struct SideJob {
    const float* src;
    float* out;
    float* out2;
    int n_slabs, count, ncols;
    int epb;        // consecutive elements per slice (power of two <= 256); lanes = 256 / epb
    int slices;     // ceil(count / epb)
    // FOLD job (fs != NULL; the shared-tile pair's fold form, pairf_body.h): the slabs hold rows of ncols + 8 columns -- (ncols - 1) raw
    // products R, then 9 tap-masked gradient sums M -- and output element (co, n) is (1 + fs[n / 9]) R + ft[n / 9] M[n % 9] for a weight
    // column, M[4] for the bias column: the affine prologue of the weight gradient's input applied after the sum instead of per element
    const float* fs;
    const float* ft;
};

struct SidePack {
    SideJob j[SIDE_MAX_JOBS];
    int n_jobs, n_slices;
};
constexpr int SIDE_FLUSH_JOBS = 32;    // the stand-alone flush takes many more jobs per launch (it is all the launch does)
struct SideFlushPack {
    SideJob j[SIDE_FLUSH_JOBS];
    int n_jobs, n_slices;
    int start[SIDE_FLUSH_JOBS];          // first slice of job k (prefix sums of j[].slices): a block finds its job in ONE round trip
};

// the caller-owned context of one stream (include/bnerv.h): the deferred-reduction queue and a scratch buffer in device memory
#include <vector>
struct bnerv_ctx {
    std::vector<SideJob> queue;          // deferred slab reductions of ONE stream, in issue order
    void* scratch = nullptr;             // pre-split weight fragments of the wide split conv kernels (reused call after call:
    size_t scratch_bytes = 0;            //  stream order keeps a call's fragments alive until its kernel has read them)
    struct BfWPlan* wplan = nullptr;     // weight-fragment plan of a captured step (convbf.hip; bnerv_ctx_wplan_* in include/bnerv.h)
};
void bnerv_wplan_free(bnerv_ctx* ctx);   // convbf.hip: releases the plan and its device buffers (called by bnerv_ctx_destroy)
// >= `bytes` of device scratch owned by the context, or nullptr when there is no context / the buffer would have to grow while
// `st` is being captured into a graph (allocation is illegal there: the first, eager calls size it)
void* bnerv_ctx_scratch(bnerv_ctx* ctx, size_t bytes, hipStream_t st);

// host side (abi.hip).  The queue belongs to the caller's bnerv_ctx; ctx == NULL means "no queue": nothing to take, and a push
// runs the reduction at once on `st`.
void bnerv_side_push(bnerv_ctx* ctx, hipStream_t st, const void* src, int n_slabs, int count, int ncols, float* out, float* out2,
                     const float* fold_scale = nullptr, const float* fold_shift = nullptr);
void bnerv_side_take(bnerv_ctx* ctx, SidePack* sp, int max_slices);   // moves up to SIDE_MAX_JOBS queued jobs (while their slices fit
                                                         // max_slices) into *sp (n_jobs = 0 if none): a small grid must not host a big reduction
int bnerv_side_flush(bnerv_ctx* ctx, hipStream_t st);    // standalone launch(es) for everything still queued
int bnerv_side_pending(const bnerv_ctx* ctx);

// one slice (256 threads, `red` = 256 floats of LDS the caller no longer needs; caller guarantees a barrier before)
// FOLD: the instantiation of the stand-alone flush kernels (abi.hip), which alone execute fold jobs (bnerv_side_take never hands one to a hosting
// launch: the hosted copy of this routine inside the hot kernels keeps its register footprint)
// which job slice s belongs to; *s becomes the slice index inside that job
template <class Pack>
__device__ __forceinline__ int side_find_job(const Pack& sp, int* s) {
    int j = 0;
    while (j < sp.n_jobs - 1 && *s >= sp.j[j].slices) { *s -= sp.j[j].slices; ++j; }
    return j;
}
// the stand-alone flush carries up to 32 jobs: the walk above is a chain of up to 31 dependent scalar loads per BLOCK (measured: the 44 000
// blocks of C3's final flush spent most of their 142 us in it).  Here lane k compares against start[k]: one load, one ballot.
template <>
__device__ __forceinline__ int side_find_job<SideFlushPack>(const SideFlushPack& sp, int* s) {
    const int k = (int)(threadIdx.x & 63);
    const int st = k < sp.n_jobs ? sp.start[k < SIDE_FLUSH_JOBS ? k : 0] : 0x7fffffff;
    const unsigned long long m = __ballot(st <= *s);
    const int j = __popcll(m) - 1;                         // start[0] == 0: at least one bit
    *s -= __shfl(st, j, 64);
    return __builtin_amdgcn_readfirstlane(j);
}

template <class Pack, bool FOLD = false>
__device__ __forceinline__ void side_slice(const Pack& sp, int s, float* red) {
    const int j = side_find_job(sp, &s);
    const SideJob& job = sp.j[j];
    const int epb = job.epb, lanes = 256 / epb;
    const int e = threadIdx.x % epb, lane = threadIdx.x / epb;
    const int i = s * epb + e;
    // 16 loads in flight per thread (one memory latency per 16 slabs), combined in a fixed tree
    const bool fold = FOLD && job.fs != nullptr;
    float acc = 0.f, acc2 = 0.f;
    int fci = 0;
    bool fbias = false;
    if (i < job.count) {
        size_t cnt = (size_t)job.count;
        size_t i0 = (size_t)i, i1 = 0;
        if (fold) {
            const int co = i / job.ncols, n = i - co * job.ncols, rowl = job.ncols + 8;
            cnt = (size_t)(job.count / job.ncols) * rowl;
            fbias = n == job.ncols - 1;
            fci = fbias ? 0 : n / 9;
            i1 = (size_t)co * rowl + (job.ncols - 1) + (fbias ? 4 : n - fci * 9);
            i0 = fbias ? i1 : (size_t)co * rowl + n;
        }
        const float* p = job.src + i0;
        const float* p2 = job.src + i1;
        for (int k0 = lane; k0 < job.n_slabs; k0 += 16 * lanes) {
            float v[16], u2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u * lanes;
                v[u] = k < job.n_slabs ? p[(size_t)k * cnt] : 0.f;
                u2[u] = (fold && !fbias && k < job.n_slabs) ? p2[(size_t)k * cnt] : 0.f;
            }
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) { v[u] += v[u + w]; u2[u] += u2[u + w]; }
            acc += v[0];
            acc2 += u2[0];
        }
    }
    red[lane * epb + e] = acc;
    __syncthreads();
    float t = 0.f;
    if (lane == 0 && i < job.count)
        for (int l = 0; l < lanes; ++l) t += red[l * epb + e];
    if (fold) {                                            // (block-uniform: a slice belongs to one job) the second column sum through the same area
        __syncthreads();
        red[lane * epb + e] = acc2;
        __syncthreads();
        if (lane == 0 && i < job.count && !fbias) {
            float t2 = 0.f;
            for (int l = 0; l < lanes; ++l) t2 += red[l * epb + e];
            t = (1.0f + job.fs[fci]) * t + job.ft[fci] * t2;
        }
    }
    if (lane == 0 && i < job.count) {
        if (job.ncols > 0) {
            const int co = i / job.ncols, n = i - co * job.ncols;
            if (n < job.ncols - 1) job.out[(size_t)co * (job.ncols - 1) + n] = t;
            else if (job.out2) job.out2[co] = t;
        } else {
            job.out[i] = t;
        }
    }
    __syncthreads();
}

// the slices a hosting block executes: blocks are ranked from the END of the grid (they own the fewest tiles)
__device__ __forceinline__ void side_run_hosted(const SidePack& sp, float* red, int vb, int vgrid) {     // (vb of vgrid: a launch that runs two kernels' blocks)
    if (sp.n_slices == 0) return;
    __syncthreads();
    for (int s = vgrid - 1 - vb; s < sp.n_slices; s += vgrid) side_slice(sp, s, red);
}
__device__ __forceinline__ void side_run_hosted(const SidePack& sp, float* red) { side_run_hosted(sp, red, (int)blockIdx.x, (int)gridDim.x); }
