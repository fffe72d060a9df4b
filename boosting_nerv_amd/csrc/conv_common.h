// conv_common.h -- tile geometry, kernel arguments and raw-buffer helpers shared by the implicit-GEMM convolution kernels
// (conv.hip: 16x16x4 f32 MFMA family; conv4.hip: 4x4x1 f32 MFMA family for the <= 12-channel layers).
#pragma once
#include "common.h"

namespace bnerv_conv {

constexpr int TH = 8, TW = 32;     // spatial tile
constexpr int CC = 16;             // input channels per K chunk
constexpr int NQ = CC / 4;
constexpr int CS = TH * TW + 4;    // s_out channel stride (floats): 16-B aligned, spreads ds_write_b128 over all banks
constexpr int W_RESIDENT_MAX = 12288;   // floats (48 KB) of B fragments kept resident per block

template <int KS> struct Geo {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int ROWS = TH + 2 * PAD;
    static constexpr int XOFF = (KS == 3) ? 4 : 0;             // left margin, multiple of 4 -> aligned float4 segments
    static constexpr int RS = TW + 2 * XOFF;                   // 40 / 32
    static constexpr int SEGS = RS / 4;
    static constexpr int PLANE_RAW = ROWS * RS;                // 400 / 256
    static constexpr int PLANE = ((PLANE_RAW - 16 + 31) / 32) * 32 + 16;   // 400 / 272 : == 16 (mod 32)
    static constexpr int T = KS * KS;
    static constexpr int COL0 = XOFF - PAD;                    // LDS column of input x = x0 + px + kx - PAD is px + kx + COL0
    static constexpr int SLOTS = CC * ROWS * SEGS;             // float4 slots of a full chunk
    static constexpr int NPRE = (SLOTS + 255) / 256;           // per-thread prefetch registers (float4)
};

struct KArgs {
    bnerv_conv_desc d;
    int tiles_x, tiles_y, ngroups, total_items;
    int w_resident;       // 1: all B fragments of one cout-group stay in LDS
    int nq_total;         // ceil(Cin/4) (resident stride)
    int vec;              // 1: W % 4 == 0 and aligned pointers -> float4 staging and epilogue
    int ksplit;           // >1: the K (input-channel chunk) range is split over `ksplit` work items; each writes a raw
    int chunks_per_split; //     partial result to its slab in d.partial ([ksplit][B][Cout][H][W]), finished by reduce_slabs
    unsigned magic_tiles, magic_tiles_x;   // floor(2^32 / n) + 1: a / n == umulhi(a, magic) for a * n < 2^32 (lean kernel's item decode)
};

static inline unsigned div_magic(int n) { return n <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)n) + 1ull); }   // 0 encodes n == 1
__device__ __forceinline__ int fast_div(int a, unsigned magic) { return magic ? (int)__umulhi((unsigned)a, magic) : a; }


template <int IN>
__device__ __forceinline__ float xform1(float v, float sc, float sh, float aux) {
    if constexpr (IN == BNERV_IN_AFFINE) return v * sc + sh;
    if constexpr (IN == BNERV_IN_GELU_AFFINE) return gelu_f(v) * sc + sh;
    if constexpr (IN == BNERV_IN_TANHGRAD) { const float t = 2.0f * aux - 1.0f; return v * 0.5f * (1.0f - t * t); }
    return v;
}


typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned shift_bytes, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) - shift_bytes), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff, (int)soff, 0);
    // Measured on gfx950 (tools/klean2.py, ROCm 7.2): a VALU write of the store's data VGPRs two instructions after a 128-bit
    // buffer store WITH AN SGPR soffset corrupts lanes 12..15 of every row of 16 (the store reads its upper data late).  LLVM's
    // hazard recognizer pads this case only when soffset is not a register, so the wait states are inserted here, fenced so that
    // the scheduler cannot move a VALU instruction in between.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3");
    __builtin_amdgcn_sched_barrier(0);
}


struct LItem { int b, ty, tx; };

constexpr size_t LEAN_MAX_BYTES = 0x7ff00000;            // every tensor view must stay below the OOB marker offset

}  // namespace bnerv_conv
