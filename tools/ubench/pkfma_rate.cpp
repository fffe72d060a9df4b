// Micro-benchmark (not part of the product): issue rate of v_fmac_f32 vs v_pk_fma_f32 on gfx950 with the operand mix of a
// direct convolution: accumulator VGPR (pair), weight from an SGPR (pair), input from a VGPR (op_sel broadcast of one half).
//   build: hipcc --offload-arch=gfx950 -O3 -o pkfma_rate pkfma_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* wsrc, int iters) {
    f32x2 acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = f32x2{0.f, 0.f};
    f32x2 in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) in[i] = f32x2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
    typedef const unsigned long long __attribute__((address_space(4))) cu64;
    const unsigned long long w0 = ((cu64*)wsrc)[0], w1 = ((cu64*)wsrc)[1];
    const float w0lo = ((const float __attribute__((address_space(4)))*)wsrc)[0], w0hi = ((const float __attribute__((address_space(4)))*)wsrc)[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                if constexpr (MODE == 0) {          // 2 x v_fmac (SGPR weight)
                    asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %4, %3" : "+v"(acc[i][0]), "+v"(acc[i][1]) : "s"(w0lo), "v"(in[i & 3][0]), "s"(w0hi));
                } else if constexpr (MODE == 1) {   // v_pk_fma: acc pair += sgpr pair * broadcast(lo of vgpr pair)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"((i & 1) ? w1 : w0), "v"(in[i & 3]));
                } else if constexpr (MODE == 2) {   // v_pk_fma: broadcast(hi of vgpr pair)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "s"((i & 1) ? w1 : w0), "v"(in[i & 3]));
                } else if constexpr (MODE == 3) {   // v_pk_fma: all VGPR, natural pairs
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(in[(i + 1) & 3]), "v"(in[i & 3]));
                } else {                            // v_pk_fma: sgpr broadcast lo * vgpr natural pair (pack over pixels)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"((i & 1) ? w1 : w0), "v"(in[i & 3]));
                }
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) r += acc[i][0] + acc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(float* d, const float* w, int blocks_per_cu) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, w, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma_lane = (double)grid * 256 * iters * 144 * 2;      // lane-FMAs
    printf("mode %d waves/SIMD %d: %7.1f TFLOP/s  (%.2f ms)\n", MODE, blocks_per_cu, 2 * fma_lane / ms / 1e9, ms);
}

int main() {
    float *d, *w;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    hipMalloc(&w, 64);
    float hw[4] = {1e-3f, -2e-3f, 3e-3f, -1e-3f};
    hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
    for (int b = 1; b <= 4; b *= 2) {
        run<0>(d, w, b); run<1>(d, w, b); run<2>(d, w, b); run<3>(d, w, b); run<4>(d, w, b);
    }
    return 0;
}
