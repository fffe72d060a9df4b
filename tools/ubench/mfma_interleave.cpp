// Micro-benchmark (not part of the product): does VALU work placed BETWEEN the MFMAs of the same wave hide under them, where the
// same work in a separate phase (tools/ubench/mfma_valu_mix.cpp) does not?  Each wave runs 108 MFMA 16x16x4 f32 per "tile"
// (4 independent accumulators) and NV VALU per MFMA, either interleaved (pattern 1: after every MFMA) or as one block after the
// 108 MFMAs (pattern 0).  Clock is read from s_memtime -> cycles per tile are real shader cycles, not wall time / 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VF(n) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[(n) & 3]), "+v"(v[((n) + 1) & 3]) : "v"(a0), "v"(b0));

template <int NV, int PAT, int TR>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
    float a0 = lane * 0.001f, b0 = lane * 0.002f;
    float v[4] = {a0, b0, a0 + 1, b0 + 1};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 27; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a0), "v"(b0));
                if constexpr (PAT == 1) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        if constexpr (TR) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 3]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(a0), "v"(b0));
                    }
                }
            }
        if constexpr (PAT == 0) {
#pragma unroll
            for (int j = 0; j < NV * 108; ++j) {
                if constexpr (TR) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 3]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(a0), "v"(b0));
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = v[0] + v[1] + v[2] + v[3];
    for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int PAT, int TR>
void run(float* d, long long* dc, int blocks_per_cu) {
    const int iters = 200, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, PAT, TR><<<grid, 256>>>(d, dc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, PAT, TR><<<grid, 256>>>(d, dc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 * iters * 108 * 2048.0;
    // s_memtime counts at a fixed 100 MHz on gfx9; report both
    printf("NV/MFMA %d %s %s waves/SIMD %d: %7.1f TFLOP/s  %8.1f us/tile-round  memtime ticks/tile %.1f\n", NV, PAT ? "interleaved" : "phased     ",
           TR ? "v_exp" : "v_fma", blocks_per_cu, flops / ms / 1e9, ms * 1e3 / iters, (double)c / iters);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 4);
    long long* dc; hipMalloc(&dc, 64);
    for (int w = 1; w <= 4; w *= 2) {
        if (w == 1) { run<0, 0, 0>(d, dc, 1); run<1, 0, 0>(d, dc, 1); run<1, 1, 0>(d, dc, 1); run<2, 0, 0>(d, dc, 1); run<2, 1, 0>(d, dc, 1); run<4, 0, 0>(d, dc, 1); run<4, 1, 0>(d, dc, 1); run<6, 1, 0>(d, dc, 1); run<2, 1, 1>(d, dc, 1); run<2, 0, 1>(d, dc, 1); }
        if (w == 2) { run<0, 0, 0>(d, dc, 2); run<1, 0, 0>(d, dc, 2); run<1, 1, 0>(d, dc, 2); run<2, 0, 0>(d, dc, 2); run<2, 1, 0>(d, dc, 2); run<4, 0, 0>(d, dc, 2); run<4, 1, 0>(d, dc, 2); run<6, 1, 0>(d, dc, 2); run<2, 1, 1>(d, dc, 2); run<2, 0, 1>(d, dc, 2); }
        if (w == 4) { run<0, 0, 0>(d, dc, 4); run<1, 0, 0>(d, dc, 4); run<1, 1, 0>(d, dc, 4); run<2, 0, 0>(d, dc, 4); run<2, 1, 0>(d, dc, 4); run<4, 0, 0>(d, dc, 4); run<4, 1, 0>(d, dc, 4); run<6, 1, 0>(d, dc, 4); run<2, 1, 1>(d, dc, 4); run<2, 0, 1>(d, dc, 4); }
    }
    return 0;
}
