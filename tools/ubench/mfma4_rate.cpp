// Micro-benchmark (not part of the product): issue interval of the f32 MFMA shapes on gfx950 (independent accumulators, operands in registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[NACC];
    f32x16 big[2];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 5, 0);
                else if constexpr (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
                else if constexpr (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
    }
    float r = 0.f;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE, int NACC>
void run(float* d, int bpc, double macs_per_inst) {
    const int iters = 1000, grid = 256 * bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NACC><<<grid, 256>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE, NACC><<<grid, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninst = (double)iters * 8 * NACC;                 // per wave
    const double cyc = ms * 1e-3 * 2.4e9 / (ninst * bpc);          // cycles per instruction per SIMD at 2.4 GHz
    printf("mode %d nacc %2d waves/SIMD %d: %.1f cycles/inst @2.4GHz, %.1f TFLOP/s\n", MODE, NACC, bpc, cyc, 2 * macs_per_inst * ninst * grid * 4 / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0, 1>(d, 1, 256); run<0, 2>(d, 1, 256); run<0, 3>(d, 1, 256); run<0, 4>(d, 1, 256); run<0, 6>(d, 1, 256); run<0, 3>(d, 2, 256); run<0, 12>(d, 1, 256); run<0, 12>(d, 2, 256);
    run<1, 12>(d, 1, 256); run<1, 24>(d, 1, 256);
    run<2, 12>(d, 1, 1024); run<2, 12>(d, 2, 1024);
    return 0;
}
