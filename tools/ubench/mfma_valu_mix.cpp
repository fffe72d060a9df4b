// Micro-benchmark (not part of the product): do ordinary VALU / SALU instructions of one wave overlap with the MFMAs of the other
// waves on the same SIMD?  Every wave alternates a "K loop" (108 independent-accumulator MFMA 16x16x4 f32, operands in
// registers) with NV VALU (or NS SALU) instructions; 4 blocks of 4 waves per CU -> 4 waves per SIMD, odd blocks start with the
// VALU phase so the phases of co-resident waves are mixed.  If the pipes overlap, time stays at the MFMA bound until
// 4 * NV * 4 cycles exceeds 4 * 108 * 32 cycles; if they serialise, time grows by NV * 4 cycles per wave-tile from the start.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NS>
__global__ __launch_bounds__(256) void k(float* out, int iters, int mixed) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
    float a0 = lane * 0.001f, b0 = lane * 0.002f;
    float v0 = a0, v1 = b0, v2 = a0 + 1, v3 = b0 + 1;
    int s0 = iters, s1 = mixed;
    const bool valu_first = mixed && (blockIdx.x & 1);
    for (int it = 0; it < iters; ++it) {
        if (!valu_first) {
#pragma unroll
            for (int i = 0; i < 27; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NV / 4; ++i) {
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a0), "v"(b0));
        }
#pragma unroll
        for (int i = 0; i < NS / 4; ++i) {
            asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, 7\n s_lshl_b32 %1, %1, 1" : "+s"(s0), "+s"(s1));
        }
        if (valu_first) {
#pragma unroll
            for (int i = 0; i < 27; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[m], 0, 0, 0);
        }
    }
    float r = v0 + v1 + v2 + v3 + (float)(s0 ^ s1);
    for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NV, int NS>
void run(float* d, int blocks_per_cu, int mixed) {
    const int iters = 300, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NS><<<grid, 256>>>(d, 10, mixed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, NS><<<grid, 256>>>(d, iters, mixed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc_per_tile_round = ms * 1e-3 * 2.4e9 / iters;          // cycles per (waves/SIMD) wave-tiles on one SIMD
    const double flops = (double)grid * 4 * iters * 108 * 2048.0;
    printf("NV %4d NS %4d waves/SIMD %d mixed %d: %7.1f TFLOP/s   %8.0f cycles/round  (MFMA floor %d, VALU@4cyc %d)\n", NV, NS, blocks_per_cu, mixed,
           flops / ms / 1e9, cyc_per_tile_round, blocks_per_cu * 108 * 32, blocks_per_cu * NV * 4);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 4);
    for (int mixed = 0; mixed < 2; ++mixed) {
        run<0, 0>(d, 4, mixed);
        run<128, 0>(d, 4, mixed);
        run<256, 0>(d, 4, mixed);
        run<512, 0>(d, 4, mixed);
        run<1024, 0>(d, 4, mixed);
        run<0, 256>(d, 4, mixed);
        run<0, 512>(d, 4, mixed);
        run<0, 1024>(d, 4, mixed);
        run<512, 512>(d, 4, mixed);
    }
    run<512, 0>(d, 1, 0);
    run<512, 0>(d, 2, 1);
    return 0;
}
