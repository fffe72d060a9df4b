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
#include <vector>

namespace {
std::vector<SideJob> g_side_queue;       // one process drives one GPU with one launching thread (torch's autograd thread order)

__global__ __launch_bounds__(256) void side_flush_kernel(const SidePack sp) {
    __shared__ float red[256];
    side_slice(sp, blockIdx.x, red);
}
}  // namespace

void bnerv_side_push(const void* src, int n_slabs, int count, int ncols, float* out, float* out2) {
    SideJob j;
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
    g_side_queue.push_back(j);
}

void bnerv_side_take(SidePack* sp, int max_slices) {
    sp->n_jobs = 0;
    sp->n_slices = 0;
    int n = 0;
    while (n < (int)g_side_queue.size() && n < SIDE_MAX_JOBS && sp->n_slices + g_side_queue[n].slices <= max_slices) {
        sp->j[n] = g_side_queue[n];
        sp->n_slices += g_side_queue[n].slices;
        ++n;
    }
    sp->n_jobs = n;
    g_side_queue.erase(g_side_queue.begin(), g_side_queue.begin() + n);
}

int bnerv_side_pending() { return (int)g_side_queue.size(); }

int bnerv_side_flush(hipStream_t st) {
    while (!g_side_queue.empty()) {
        SidePack sp;
        bnerv_side_take(&sp, 0x7fffffff);
        hipLaunchKernelGGL(side_flush_kernel, dim3(sp.n_slices), dim3(256), 0, st, sp);
        BNERV_LAUNCH_CHECK("side_flush");
    }
    return BNERV_OK;
}

extern "C" int bnerv_reduce_slabs_deferred(const float* slabs, int n_slabs, int count, float* out) {
    BNERV_REQUIRE(slabs && out && n_slabs > 0 && count > 0, "reduce_slabs_deferred: bad args");
    bnerv_side_push(slabs, n_slabs, count, 0, out, nullptr);
    return BNERV_OK;
}
extern "C" int bnerv_flush_deferred(void* stream) { return bnerv_side_flush(reinterpret_cast<hipStream_t>(stream)); }
extern "C" int bnerv_deferred_pending(void) { return bnerv_side_pending(); }
