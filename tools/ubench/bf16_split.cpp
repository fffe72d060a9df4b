// Micro-benchmark (not part of the product).  Two questions behind the bf16-split conv core:
//  (1) do ordinary VALU instructions of other waves overlap with bf16 MFMAs (they do NOT with the fp32 16x16x4 MFMA:
//      tools/ubench/mfma_interleave.cpp), and
//  (2) how accurate is an fp32 dot product rebuilt from bf16 pieces on v_mfma_f32_16x16x32_bf16:
//      x3 = hi*hi + hi*lo + lo*hi (2 pieces), x6 = 3 pieces, 6 products -- against fp64 and against the fp32 MFMA chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int TR>
__global__ __launch_bounds__(256) void kmix(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(lane * 0.002f - i); }
    float a0 = lane * 0.001f, b0 = lane * 0.002f;
    float v[4] = {a0, b0, a0 + 1, b0 + 1};
    const bool valu_first = blockIdx.x & 1;
    for (int it = 0; it < iters; ++it) {
        if (!valu_first)
#pragma unroll
            for (int i = 0; i < 54; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV * 216; ++j) {
            if constexpr (TR) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 3]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(a0), "v"(b0));
        }
        if (valu_first)
#pragma unroll
            for (int i = 0; i < 54; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float r = v[0] + v[1] + v[2] + v[3];
    for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NV, int TR>
void runmix(float* d, int bpc) {
    const int iters = 200, grid = 256 * bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kmix<NV, TR><<<grid, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kmix<NV, TR><<<grid, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 216 * 16384.0;
    printf("bf16 MFMA 16x16x32, %d %s per MFMA (phased), waves/SIMD %d: %8.1f TFLOP/s  %7.2f us/round (MFMA only floor %.2f us at 2.4 GHz)\n", NV, TR ? "v_exp" : "v_fma", bpc,
           flops / ms / 1e9, ms * 1e3 / iters, bpc * 216 * 16 / 2400.0);
}

// accuracy: one wave computes D = A[16xK] * B[Kx16] three ways.  K = 128 (4 steps of 32).
__device__ inline float bf_up(__bf16 h) { return (float)h; }
__global__ void kacc(const float* A, const float* B, float* d3, float* d6, float* d32, int K) {
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    f32x4 c3{0, 0, 0, 0}, c6{0, 0, 0, 0}, c32{0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        bf16x8 a1, a2, a3, b1, b2, b3;
        for (int e = 0; e < 8; ++e) {
            const float x = A[i * K + k0 + kq * 8 + e], y = B[(k0 + kq * 8 + e) * 16 + i];
            a1[e] = (__bf16)x; float r = x - bf_up(a1[e]); a2[e] = (__bf16)r; r -= bf_up(a2[e]); a3[e] = (__bf16)r;
            b1[e] = (__bf16)y; float s = y - bf_up(b1[e]); b2[e] = (__bf16)s; s -= bf_up(b2[e]); b3[e] = (__bf16)s;
        }
        // smallest terms first
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c6, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c3, 0, 0, 0);
        for (int e = 0; e < 8; ++e)
            c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + e * 4 + kq], B[(k0 + e * 4 + kq) * 16 + i], c32, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {       // D[row = 4*kq + r][col = i]
        d3[(4 * kq + r) * 16 + i] = c3[r]; d6[(4 * kq + r) * 16 + i] = c6[r]; d32[(4 * kq + r) * 16 + i] = c32[r];
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 4);
    runmix<0, 0>(d, 4); runmix<1, 0>(d, 4); runmix<2, 0>(d, 4); runmix<4, 0>(d, 4); runmix<8, 0>(d, 4); runmix<2, 1>(d, 4);
    runmix<0, 0>(d, 2); runmix<2, 0>(d, 2); runmix<4, 0>(d, 2);
    runmix<0, 0>(d, 1); runmix<2, 0>(d, 1);
    for (int K : {128, 1024}) {
        std::mt19937 g(7); std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> A(16 * K), B(K * 16);
        for (auto& x : A) x = nd(g);
        for (auto& x : B) x = nd(g) * 0.1f;
        float *dA, *dB, *o3, *o6, *o32;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&o3, 1024); hipMalloc(&o6, 1024); hipMalloc(&o32, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        kacc<<<1, 64>>>(dA, dB, o3, o6, o32, K);
        float h3[256], h6[256], h32[256];
        hipMemcpy(h3, o3, 1024, hipMemcpyDeviceToHost); hipMemcpy(h6, o6, 1024, hipMemcpyDeviceToHost); hipMemcpy(h32, o32, 1024, hipMemcpyDeviceToHost);
        double e3 = 0, e6 = 0, e32 = 0, mx = 0, sabs = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ref = 0, sa = 0;
            for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; sa += fabs((double)A[i * K + k] * B[k * 16 + j]); }
            e3 = fmax(e3, fabs(h3[i * 16 + j] - ref) / sa); e6 = fmax(e6, fabs(h6[i * 16 + j] - ref) / sa); e32 = fmax(e32, fabs(h32[i * 16 + j] - ref) / sa);
            mx = fmax(mx, fabs(ref)); sabs = fmax(sabs, sa);
        }
        printf("K=%4d  max |err| / sum|a b|:  bf16x3 %.3e   bf16x6 %.3e   fp32 MFMA 16x16x4 %.3e   (max |ref| %.3f, max sum|ab| %.3f)\n", K, e3, e6, e32, mx, sabs);
    }
    return 0;
}
