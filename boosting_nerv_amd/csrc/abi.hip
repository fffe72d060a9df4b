// abi.hip -- error reporting and version entry points of the C-ABI.
#include "common.h"
#include <string.h>

namespace { thread_local char g_err[512] = ""; }

int bnerv_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int bnerv_abi_version(void) { return BNERV_ABI_VERSION; }
extern "C" const char* bnerv_last_error(void) { return g_err; }
extern "C" const char* bnerv_build_arch(void) { return "gfx950"; }

// ---------------------------------------------------------------------------------------------------------------- side jobs
#include "sidejob.h"
#include <new>
#include <vector>

namespace {
__global__ __launch_bounds__(256) void side_flush_kernel(const SidePack sp) {
    __shared__ float red[256];
    side_slice<SidePack, true>(sp, blockIdx.x, red);
}
__global__ __launch_bounds__(256) void side_flush_many_kernel(const SideFlushPack sp) {
    __shared__ float red[256];
    side_slice<SideFlushPack, true>(sp, blockIdx.x, red);
}

SideJob make_job(const void* src, int n_slabs, int count, int ncols, float* out, float* out2, const float* fs = nullptr, const float* ft = nullptr) {
    SideJob j;
    j.fs = fs;
    j.ft = ft;
    j.src = reinterpret_cast<const float*>(src);
    j.out = out;
    j.out2 = out2;
    j.n_slabs = n_slabs;
    j.count = count;
    j.ncols = ncols;
    // lanes >= n_slabs / 16 (one batch of 16 loads per thread when possible), power of two in [8, 128]
    int lanes = 8;
    while (lanes < 128 && lanes * 16 < n_slabs) lanes *= 2;
    j.epb = 256 / lanes;
    j.slices = (count + j.epb - 1) / j.epb;
    return j;
}
}  // namespace

void bnerv_side_push(bnerv_ctx* ctx, hipStream_t st, const void* src, int n_slabs, int count, int ncols, float* out, float* out2, const float* fold_scale, const float* fold_shift) {
    const SideJob j = make_job(src, n_slabs, count, ncols, out, out2, fold_scale, fold_shift);
    if (ctx) { ctx->queue.push_back(j); return; }
    SidePack sp;                                          // no context: run it now
    sp.j[0] = j;
    sp.n_jobs = 1;
    sp.n_slices = j.slices;
    hipLaunchKernelGGL(side_flush_kernel, dim3(sp.n_slices), dim3(256), 0, st, sp);
}

void bnerv_side_take(bnerv_ctx* ctx, SidePack* sp, int max_slices) {
    sp->n_jobs = 0;
    sp->n_slices = 0;
    if (!ctx) return;
    // hosts are launches with an idle tail (several rounds of tiles per block); a one-tile-per-block launch (the <= 180x320 stages: every
    // caller passes max_slices = 2 x its grid) would pay the hosted slices on top of its only round
    static const int min_grid = [] { const char* e = getenv("BNERV_SIDE_MIN_GRID"); return e ? atoi(e) : 0; }();
    if (max_slices < 2 * min_grid) return;
    std::vector<SideJob>& q = ctx->queue;
    // queued jobs in issue order; fold jobs (sidejob.h) stay behind for the flush kernels and are stepped over -- the jobs are independent of
    // one another (distinct outputs), the order only keeps the oldest slabs moving first
    int n = 0;
    size_t i = 0;
    while (i < q.size() && n < SIDE_MAX_JOBS) {
        if (q[i].fs) { ++i; continue; }
        if (sp->n_slices + q[i].slices > max_slices) break;
        sp->j[n] = q[i];
        sp->n_slices += q[i].slices;
        ++n;
        q.erase(q.begin() + (long)i);
    }
    sp->n_jobs = n;
}

int bnerv_side_pending(const bnerv_ctx* ctx) { return ctx ? (int)ctx->queue.size() : 0; }

int bnerv_side_flush(bnerv_ctx* ctx, hipStream_t st) {
    while (ctx && !ctx->queue.empty()) {                   // up to SIDE_FLUSH_JOBS queued reductions per launch, in issue order
        SideFlushPack sp;
        std::vector<SideJob>& q = ctx->queue;
        int n = 0;
        sp.n_slices = 0;
        while (n < (int)q.size() && n < SIDE_FLUSH_JOBS) {
            sp.j[n] = q[n];
            sp.start[n] = sp.n_slices;
            sp.n_slices += q[n].slices;
            ++n;
        }
        for (int k = n; k < SIDE_FLUSH_JOBS; ++k) sp.start[k] = 0x7fffffff;
        sp.n_jobs = n;
        q.erase(q.begin(), q.begin() + n);
        hipLaunchKernelGGL(side_flush_many_kernel, dim3(sp.n_slices), dim3(256), 0, st, sp);
        BNERV_LAUNCH_CHECK("side_flush");
    }
    return BNERV_OK;
}

extern "C" int bnerv_ctx_create(bnerv_ctx** out) {
    BNERV_REQUIRE(out != nullptr, "ctx_create: null output");
    *out = new (std::nothrow) bnerv_ctx();
    return *out ? BNERV_OK : bnerv_set_error(BNERV_E_ARG, "ctx_create: out of memory");
}
extern "C" void bnerv_ctx_destroy(bnerv_ctx* ctx) {
    if (ctx && ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx) bnerv_wplan_free(ctx);
    delete ctx;
}

extern "C" size_t bnerv_ctx_scratch_bytes(const bnerv_ctx* ctx) { return ctx ? ctx->scratch_bytes : 0; }

extern "C" int bnerv_ctx_reserve(bnerv_ctx* ctx, size_t bytes) {
    BNERV_REQUIRE(ctx != nullptr, "ctx_reserve: null context");
    if (ctx->scratch_bytes >= bytes) return BNERV_OK;
    if (hipDeviceSynchronize() != hipSuccess) return bnerv_set_error(BNERV_E_LAUNCH, "ctx_reserve: device synchronize failed");
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    if (hipMalloc(&ctx->scratch, bytes) != hipSuccess) { ctx->scratch = nullptr; return bnerv_set_error(BNERV_E_WS, "ctx_reserve: cannot allocate %zu bytes", bytes); }
    ctx->scratch_bytes = bytes;
    return BNERV_OK;
}

void* bnerv_ctx_scratch(bnerv_ctx* ctx, size_t bytes, hipStream_t st) {
    if (!ctx) return nullptr;
    if (ctx->scratch_bytes >= bytes) return ctx->scratch;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    // grow (rare: the largest layer of the model sizes it once).  Work already queued on the stream may still read the old buffer.
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    const size_t want = bytes + bytes / 2;
    if (hipMalloc(&ctx->scratch, want) != hipSuccess) { ctx->scratch = nullptr; return nullptr; }
    ctx->scratch_bytes = want;
    return ctx->scratch;
}

extern "C" int bnerv_reduce_slabs_deferred(bnerv_ctx* ctx, void* stream, const float* slabs, int n_slabs, int count, float* out) {
    BNERV_REQUIRE(slabs && out && n_slabs > 0 && count > 0, "reduce_slabs_deferred: bad args");
    bnerv_side_push(ctx, reinterpret_cast<hipStream_t>(stream), slabs, n_slabs, count, 0, out, nullptr);
    if (!ctx) BNERV_LAUNCH_CHECK("reduce_slabs_deferred(immediate)");
    return BNERV_OK;
}
extern "C" int bnerv_flush_deferred(bnerv_ctx* ctx, void* stream) { return bnerv_side_flush(ctx, reinterpret_cast<hipStream_t>(stream)); }
extern "C" int bnerv_deferred_pending(const bnerv_ctx* ctx) { return bnerv_side_pending(ctx); }
