// common.h -- shared helpers for the bnerv gfx950 kernels (error reporting, math, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/bnerv.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int bnerv_set_error(int code, const char* fmt, ...);

#define BNERV_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return bnerv_set_error(BNERV_E_ARG, __VA_ARGS__); \
    } while (0)

#define BNERV_LAUNCH_CHECK(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) return bnerv_set_error(BNERV_E_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device math (accurate forms; never the __sinf/__expf fast intrinsics) ----
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// h = gelu(x) and g = gelu'(x) together, sharing ONE exp: e = exp(-x^2/2) is both the pdf factor and the tail of
//   erf(z) = sign(z) * (1 - (a1 t + ... + a5 t^5) e^{-z^2}),  t = 1/(1 + p|z|),  z = x/sqrt(2)      (Abramowitz-Stegun 7.1.26)
// |error| <= 1.5e-7 on erf, i.e. 7.5e-8 on the normal cdf: at the level of one fp32 ulp of the results.  Used where both values
// are produced at once (the TAT conv0 epilogue); ~29 VALU per element instead of ~54 for gelu_f + gelu_grad_f.
__device__ __forceinline__ void gelu_pair_f(float x, float* h, float* g) {
    // e = exp(-x^2/2) <= 1 as one v_exp_f32 (1 ulp): the argument's rounding error |arg| * 2^-24 is relative to e itself, i.e.
    // below 2e-10 absolute everywhere -- far inside the A-S bound; expf() spends ~10 VALU on ranges that cannot occur here
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
    const float az = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);   // v_rcp_f32 (1 ulp) -- the IEEE division sequence costs ~10 VALU
    float poly = 1.061405429f;
    poly = fmaf(poly, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = 1.0f - poly * t * e;
    const float cdf = 0.5f + 0.5f * copysignf(erf_abs, x);
    *h = x * cdf;
    *g = fmaf(x, 0.39894228040143267794f * e, cdf);
}

// sin and cos together for the block activation (sin) and its saved derivative (cos): 3-term Cody-Waite reduction by pi/2 and
// the cephes single-precision minimax polynomials on [-pi/4, pi/4]; |x| <= 8192 (activations are O(10)), libm beyond.
// Measured against float64 through an identity 1x1 conv (tests/test_gpu_ops.py): max |error| < 2.4e-7 on [-8192, 8192].
// ~25 VALU per pair instead of ~60 executed for sincosf (whose code also carries the Payne-Hanek path).
__device__ __forceinline__ void sincos_f(float x, float* s, float* c) {
    if (fabsf(x) > 8192.0f) { sincosf(x, s, c); return; }
    const float k = rintf(x * 0.63661977236758134308f);            // x * 2/pi
    float r = fmaf(-k, 1.5703125f, x);                             // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    const float z = r * r;
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, z, -1.6666654611e-1f);
    const float sr = fmaf(sp * z, r, r);
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, z, 4.166664568298827e-2f);
    const float cr = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// ---- wave64 / block reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Workgroup barrier for LDS hazards only: waits for this wave's LDS traffic (lgkmcnt) and NOT for global loads / stores
// in flight.  __syncthreads() carries a global-memory fence (s_waitcnt vmcnt(0)) which would drain a register prefetch of
// the next tile and expose the latency of the epilogue stores at every barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// XCD-aware remap of a 1-D block index: blocks that the dispatcher places on the same XCD (b % 8) get a contiguous
// range of logical indices, so neighbouring tiles (which share halo rows) share one L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int orig, int n) {
    const int xcd = orig & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
