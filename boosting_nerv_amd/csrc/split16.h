// split16.h -- f32 operands as 16-bit pieces for the 16-bit matrix pipe (see convbf.hip for the scheme and its measured accuracy).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { SP_BF16X6 = 0, SP_BF16X3 = 1, SP_F16X3 = 2 };
template <int SP> struct Split {
    static constexpr int NS = (SP == SP_BF16X6) ? 3 : 2;
    static constexpr bool SCALED = (SP == SP_F16X3);
};

template <int SP>
__device__ __forceinline__ unsigned pk16(float a, float b) {
    if constexpr (SP == SP_F16X3) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h2));
    } else {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, b2));
    }
}
template <int SP>
__device__ __forceinline__ f32x2 unpk16(unsigned pk) {
    if constexpr (SP == SP_F16X3) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_convertvector(__builtin_bit_cast(h2, pk), f32x2);
    } else {
        return f32x2{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
    }
}
// NE (<= 8, even) floats -> NS packed 8-element pieces, round to nearest even at every level (the residuals are exact in f32);
// elements NE..7 of every piece are zero
template <int SP, int NE>
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&out)[Split<SP>::NS]) {
    constexpr int NS = Split<SP>::NS;
    f32x2 r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = f32x2{x[2 * e], x[2 * e + 1]};
#pragma unroll
    for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (2 * e < NE) {
                const unsigned pk = pk16<SP>(r[e][0], r[e][1]);
                out[p][e] = pk;
                if (p + 1 < NS) r[e] -= unpk16<SP>(pk);
            } else {
                out[p][e] = 0u;
            }
        }
    }
}

template <int SP>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (SP == SP_F16X3) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
