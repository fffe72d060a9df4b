// Micro-benchmark (not part of the product): what a device-wide barrier inside ONE launch costs on gfx950 against the launch boundary it
// replaces -- N dependent phases as (a) N kernel nodes of a replayed hipGraph, (b) one kernel with N - 1 grid barriers (release fence,
// one atomic ticket per block, spin, acquire fence).  Each phase reads what its neighbour block wrote in the previous phase (so the
// fences are really needed: the check at the end fails without them) and writes `kb` KB per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// phase p: block b reads block (b + 1) % G's slice of buf[(p + 1) & 1], adds 1, writes its slice of buf[p & 1]
__device__ __forceinline__ void phase(float* b0, float* b1, int p, int nf) {
    const float* src = (p & 1) ? b0 : b1;
    float* dst = (p & 1) ? b1 : b0;
    const int nb = (blockIdx.x + 1) % gridDim.x;
    for (int i = threadIdx.x; i < nf; i += blockDim.x) dst[(size_t)blockIdx.x * nf + i] = src[(size_t)nb * nf + i] + 1.0f;
}
__global__ __launch_bounds__(256) void k_phase(float* b0, float* b1, int p, int nf) { phase(b0, b1, p, nf); }
__global__ __launch_bounds__(256) void k_chain(float* b0, float* b1, int np, int nf, unsigned* ctr, unsigned base) {
    for (int p = 0; p < np; ++p) {
        phase(b0, b1, p, nf);
        if (p + 1 < np) grid_barrier(ctr, base + (unsigned)(p + 1) * gridDim.x);
    }
}

int main() {
    const int NP = 32;
    hipStream_t st; hipStreamCreate(&st);
    unsigned* ctr; hipMalloc(&ctr, 4);
    for (int G : {60, 230, 256, 512, 768})
        for (int kb : {1, 16, 64}) {
            const int nf = kb * 256;
            float *b0, *b1; hipMalloc(&b0, (size_t)G * nf * 4); hipMalloc(&b1, (size_t)G * nf * 4);
            hipMemset(b0, 0, (size_t)G * nf * 4); hipMemset(b1, 0, (size_t)G * nf * 4);
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int p = 0; p < NP; ++p) k_phase<<<G, 256, 0, st>>>(b0, b1, p, nf);
            hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st); for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms_g; hipEventElapsedTime(&ms_g, e0, e1);
            // chained
            hipMemset(b0, 0, (size_t)G * nf * 4); hipMemset(b1, 0, (size_t)G * nf * 4); hipMemset(ctr, 0, 4);
            unsigned base = 0;
            hipGraph_t g2; hipGraphExec_t ge2;
            // (the ticket base advances per launch: one captured launch per graph would need a device-side epoch; here plain launches)
            for (int i = 0; i < 3; ++i) { k_chain<<<G, 256, 0, st>>>(b0, b1, NP, nf, ctr, base); base += (unsigned)(NP - 1) * G; }
            hipStreamSynchronize(st);
            hipMemset(b0, 0, (size_t)G * nf * 4); hipMemset(b1, 0, (size_t)G * nf * 4);
            hipEventRecord(e0, st);
            for (int i = 0; i < 20; ++i) { k_chain<<<G, 256, 0, st>>>(b0, b1, NP, nf, ctr, base); base += (unsigned)(NP - 1) * G; }
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms_c; hipEventElapsedTime(&ms_c, e0, e1);
            std::vector<float> h((size_t)G * nf);
            hipMemcpy(h.data(), b1, h.size() * 4, hipMemcpyDeviceToHost);     // last phase (NP - 1 odd) wrote b1: value = 20 * NP
            int bad = 0; for (float v : h) bad += v != 20.0f * NP;
            printf("G %4d  %3d KB/block/phase:  graph of %d launches %.2f us/phase   one launch with grid barriers %.2f us/phase   (check: %d wrong)\n",
                   G, kb, NP, ms_g * 1e3 / (20 * NP), ms_c * 1e3 / (20 * NP), bad);
            (void)g2; (void)ge2;
            hipFree(b0); hipFree(b1);
        }
    return 0;
}
