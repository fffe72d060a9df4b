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

// Epilogue math on PACKED pairs (v_pk_fma_f32 / v_pk_mul_f32: two lanes of arithmetic per issued instruction).  The conv kernels are bound
// by instruction issue, and beside the f32 MFMA the vector instructions do not hide at all: the GELU pair and the
// sin / cos pair of an epilogue were ~29 and ~25 VALU per ELEMENT; as pairs they are ~24 and ~34 per TWO elements.  Every packed lane is the
// IEEE operation of the scalar form in the same order (contraction off, every fused multiply-add written out), so the scalar entry points
// below -- the same code on one lane -- give the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// h = gelu(x) and g = gelu'(x) together, sharing ONE exp: e = exp(-x^2/2) is both the pdf factor and the tail of
//   erf(z) = sign(z) * (1 - (a1 t + ... + a5 t^5) e^{-z^2}),  t = 1/(1 + p|z|),  z = x/sqrt(2)      (Abramowitz-Stegun 7.1.26)
// |error| <= 1.5e-7 on erf, i.e. 7.5e-8 on the normal cdf: at the level of one fp32 ulp of the results.  Used where both values
// are produced at once (the TAT conv0 epilogue).
//   e = exp(-x^2/2) <= 1 as one v_exp_f32 (1 ulp): the argument's rounding error |arg| * 2^-24 is relative to e itself, i.e. below 2e-10
//   absolute everywhere -- far inside the A-S bound; expf() spends ~10 VALU on ranges that cannot occur here.  t through v_rcp_f32 (1 ulp):
//   the IEEE division sequence costs ~10 VALU.
__device__ __forceinline__ void gelu_pair2_f(f32x2 x, f32x2* h, f32x2* g) {
#pragma clang fp contract(off)
    const f32x2 arg = (x * x) * splat2(-0.72134752044448170368f);
    const f32x2 e = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    const f32x2 az = f32x2{fabsf(x.x), fabsf(x.y)} * splat2(0.70710678118654752440f);
    const f32x2 den = pk_fma2(splat2(0.3275911f), az, splat2(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 poly = splat2(1.061405429f);
    poly = pk_fma2(poly, t, splat2(-1.453152027f));
    poly = pk_fma2(poly, t, splat2(1.421413741f));
    poly = pk_fma2(poly, t, splat2(-0.284496736f));
    poly = pk_fma2(poly, t, splat2(0.254829592f));
    const f32x2 erf_abs = pk_fma2(-(poly * t), e, splat2(1.0f));
    const f32x2 erf_s = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
    const f32x2 cdf = pk_fma2(splat2(0.5f), erf_s, splat2(0.5f));
    *h = x * cdf;
    *g = pk_fma2(x, splat2(0.39894228040143267794f) * e, cdf);
}
__device__ __forceinline__ void gelu_pair4_f(f32x4 x, f32x4* h, f32x4* g) {
    f32x2 h0, g0, h1, g1;
    gelu_pair2_f(f32x2{x.x, x.y}, &h0, &g0);
    gelu_pair2_f(f32x2{x.z, x.w}, &h1, &g1);
    *h = f32x4{h0.x, h0.y, h1.x, h1.y};
    *g = f32x4{g0.x, g0.y, g1.x, g1.y};
}
__device__ __forceinline__ void gelu_pair_f(float x, float* h, float* g) {      // one element: the same operations, the same bits
    f32x2 h2, g2;
    gelu_pair2_f(splat2(x), &h2, &g2);
    *h = h2.x; *g = g2.x;
}

// sin and cos together for the block activation (sin) and its saved derivative (cos): 3-term Cody-Waite reduction by pi/2 and
// the cephes single-precision minimax polynomials on [-pi/4, pi/4]; |x| <= 8192 (activations are O(10)), libm beyond.
// Measured against float64 through an identity 1x1 conv (tests/test_gpu_ops.py): max |error| < 2.4e-7 on [-8192, 8192].
__device__ __forceinline__ void sincos2_f(f32x2 x, f32x2* s, f32x2* c) {
#pragma clang fp contract(off)
    if (fabsf(x.x) > 8192.0f || fabsf(x.y) > 8192.0f) {
        float s0, c0, s1, c1;
        sincosf(x.x, &s0, &c0); sincosf(x.y, &s1, &c1);
        *s = f32x2{s0, s1}; *c = f32x2{c0, c1};
        return;
    }
    const f32x2 kx = x * splat2(0.63661977236758134308f);            // x * 2/pi
    const f32x2 k = {rintf(kx.x), rintf(kx.y)};
    f32x2 r = pk_fma2(-k, splat2(1.5703125f), x);                     // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
    r = pk_fma2(-k, splat2(4.837512969970703125e-4f), r);
    r = pk_fma2(-k, splat2(7.54978995489188e-8f), r);
    const f32x2 z = r * r;
    f32x2 sp = pk_fma2(z, splat2(-1.9515295891e-4f), splat2(8.3321608736e-3f));
    sp = pk_fma2(sp, z, splat2(-1.6666654611e-1f));
    const f32x2 sr = pk_fma2(sp * z, r, r);
    f32x2 cp = pk_fma2(z, splat2(2.443315711809948e-5f), splat2(-1.388731625493765e-3f));
    cp = pk_fma2(cp, z, splat2(4.166664568298827e-2f));
    const f32x2 cr = pk_fma2(cp * z, z, pk_fma2(splat2(-0.5f), z, splat2(1.0f)));
    // quadrant: odd -> swap, bit 1 -> sign of sin, bit 1 of q + 1 -> sign of cos (sign flips as xor on the bits)
    const unsigned q0 = (unsigned)(int)k.x, q1 = (unsigned)(int)k.y;
    // (element values copied to floats first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 on this compiler)
    const float srx = sr.x, sry = sr.y, crx = cr.x, cry = cr.y;
    const unsigned sr0 = __builtin_bit_cast(unsigned, srx), cr0 = __builtin_bit_cast(unsigned, crx);
    const unsigned sr1 = __builtin_bit_cast(unsigned, sry), cr1 = __builtin_bit_cast(unsigned, cry);
    const unsigned ss0 = (q0 & 1u) ? cr0 : sr0, cc0 = (q0 & 1u) ? sr0 : cr0;
    const unsigned ss1 = (q1 & 1u) ? cr1 : sr1, cc1 = (q1 & 1u) ? sr1 : cr1;
    *s = f32x2{__builtin_bit_cast(float, ss0 ^ ((q0 << 30) & 0x80000000u)), __builtin_bit_cast(float, ss1 ^ ((q1 << 30) & 0x80000000u))};
    *c = f32x2{__builtin_bit_cast(float, cc0 ^ (((q0 + 1u) << 30) & 0x80000000u)), __builtin_bit_cast(float, cc1 ^ (((q1 + 1u) << 30) & 0x80000000u))};
}
__device__ __forceinline__ void sincos4_f(f32x4 x, f32x4* s, f32x4* c) {
    f32x2 s0, c0, s1, c1;
    sincos2_f(f32x2{x.x, x.y}, &s0, &c0);
    sincos2_f(f32x2{x.z, x.w}, &s1, &c1);
    *s = f32x4{s0.x, s0.y, s1.x, s1.y};
    *c = f32x4{c0.x, c0.y, c1.x, c1.y};
}
__device__ __forceinline__ void sincos_f(float x, float* s, float* c) {
    f32x2 s2, c2;
    sincos2_f(splat2(x), &s2, &c2);
    *s = s2.x; *c = c2.x;
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

// The step's frame fetch (csrc/optim.hip bnerv_fetch_frame; train_nerv_all.py:328-332 when the clip lives in HBM): frame (int)sel[0] of the clip and
// its normalised index copied to the step's static buffers.  A device function because the weight-fragment plan's launch (convbf.hip) can carry it
// as a second block range: two independent launches at the head of every captured step become one.
__device__ __forceinline__ void fetch_frame_body(const float* __restrict__ clip, const double* __restrict__ norms, const float* __restrict__ sel,
                                                 const int n_frames, const size_t frame_elems, float* __restrict__ dst, double* __restrict__ dst_norm,
                                                 const int bx, const int nbx) {
    int k = (int)sel[0];
    k = k < 0 ? 0 : (k >= n_frames ? n_frames - 1 : k);
    const f32x4* src = reinterpret_cast<const f32x4*>(clip + (size_t)k * frame_elems);
    f32x4* out = reinterpret_cast<f32x4*>(dst);
    const size_t n4 = frame_elems / 4;
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n4; i += (size_t)nbx * 256) out[i] = src[i];
    if (bx == 0 && threadIdx.x == 0) {
        for (size_t i = n4 * 4; i < frame_elems; ++i) dst[i] = clip[(size_t)k * frame_elems + i];
        if (norms && dst_norm) dst_norm[0] = norms[k];
    }
}
