// Micro-benchmark (not part of the product): ceiling of the conv inner loop's instruction mix on gfx950.
//   variant 0: MFMA 16x16x4 f32 only (operands in registers)
//   variant 1: per MFMA one ds_read_b32 A fragment (+1 B read per 4 MFMAs), as conv.hip's K loop
//   variant 2: as 1 but 8 M-tiles per wave (B reuse x8), as the wave-tile experiment
// Reports TFLOP/s counting 2*16*16*4 flops per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR, int MT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float s[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 12 * 400 + 27 * 64; i += 256) s[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[MT];
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0, 0, 0, 0};
    const int abase = (lane >> 4) * 400 + (2 * wave) * 40 + (lane & 15) + 3;
    const float* sw = s + 12 * 400;
    float a0 = s[lane], b0 = s[lane + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float bf = b0, af[MT];
                if (VAR >= 1) bf = sw[(tap * 3 + q) * 64 + lane];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = VAR >= 1 ? s[abase + q * 1600 + ((m >> 1) % 2 + tap / 3) * 40 + (m & 1) * 16 + tap % 3] : a0;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf, acc[m], 0, 0, 0);
            }
    }
    float r = 0;
    for (int m = 0; m < MT; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int VAR, int MT>
void run(const char* name, int blocks_per_cu, float* d) {
    const int iters = 400, grid = 256 * blocks_per_cu;
    const size_t lds = (12 * 400 + 27 * 64) * 4 + (blocks_per_cu <= 2 ? 40000 : blocks_per_cu == 3 ? 20000 : 0);   // pad LDS to cap residency
    hipFuncSetAttribute((const void*)k<VAR, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<VAR, MT><<<grid, 256, lds>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<VAR, MT><<<grid, 256, lds>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 27.0 * MT * 2048.0;
    printf("%-44s blocks/CU %d (waves/SIMD %d): %7.1f TFLOP/s  (%.3f ms)\n", name, blocks_per_cu, blocks_per_cu, flops / ms / 1e9, ms);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 4);
    for (int b = 1; b <= 4; ++b) run<0, 4>("MFMA only, 4 acc/wave", b, d);
    for (int b = 1; b <= 4; ++b) run<1, 4>("1 ds_read_b32 per MFMA, 4 M-tiles/wave", b, d);
    for (int b = 1; b <= 4; ++b) run<2, 8>("1 ds_read_b32 per MFMA, 8 M-tiles/wave", b, d);
    return 0;
}
