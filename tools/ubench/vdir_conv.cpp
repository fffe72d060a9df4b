// Micro-benchmark (not part of the product): a 3x3 stride-1 convolution with <= 16 channels as a DIRECT convolution on the f32
// vector ALU -- weights in SGPRs (scalar loads of a [ci][ky][kx][co] table), one lane = a 1 x P strip of pixels x ALL couts in
// registers, the input plane of ONE channel at a time staged through LDS (double-buffered).  On gfx950 the f32 MFMA issues at
// exactly the f32 vector rate, so a 12-channel layer pays 16/12 of its FLOPs on the matrix pipe plus every fragment /
// epilogue instruction; the direct form issues exactly Cin*9*Cout FMAs per pixel and almost nothing else.
//   build: hipcc --offload-arch=gfx950 -O3 -o vdir_conv vdir_conv.cpp ; run: ./vdir_conv [H W]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(4))) cfloat;

constexpr int P = 8;               // pixels per lane (1 x P strip)
constexpr int TW = 64, TH = 32;    // block tile: 4 waves x (8 rows x 64 cols)
constexpr int XOFF = 4;            // left margin (aligned float4 segments)
constexpr int RS = TW + 2 * XOFF;  // 72 floats per LDS row
constexpr int ROWS = TH + 2;
constexpr int SEGS = RS / 4;       // 18
constexpr int SLOTS = ROWS * SEGS; // 612 float4 per plane
constexpr int NPRE = (SLOTS + 255) / 256;

template <int CIN, int COUT, int WPS>
__global__ __launch_bounds__(256, WPS) void vdir_kernel(const float* __restrict__ x, const float* wt_, const float* __restrict__ bias,
                                                      float* __restrict__ out, int H, int W, int tiles_x, int tiles) {
    __shared__ __attribute__((aligned(16))) float s_in[2][ROWS * RS];
    cfloat* wt = (cfloat*)wt_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const int tile = blockIdx.x;
    if (tile >= tiles) return;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    f32x4 pre[NPRE];
    auto issue = [&](int ci) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                const int gy = ty0 + r - 1, gx = tx0 - XOFF + 4 * sg;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4*>(x + ((size_t)ci * H + gy) * W + gx);
            }
            pre[k] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                *reinterpret_cast<f32x4*>(&s_in[buf][r * RS + 4 * sg]) = pre[k];
            }
        }
    };

    float acc[COUT][P];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float b = bias[co];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[co][p] = b;
    }
    issue(0);
    commit(0);
    __syncthreads();
    const int row0 = wave * 8 + ly;           // output row within the tile; LDS row of ky = 0 is the same index
    const int col0 = XOFF - 1 + lx * P;       // LDS column of (px = 0, kx = 0)
    for (int ci = 0; ci < CIN; ++ci) {
        if (ci + 1 < CIN) issue(ci + 1);
        const float* pl = s_in[ci & 1];
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            float win[P + 2];
            const float* rp = pl + (row0 + ky) * RS + col0;
            win[0] = rp[0];
            const f32x4 a = *reinterpret_cast<const f32x4*>(rp + 1), b = *reinterpret_cast<const f32x4*>(rp + 5);
            win[1] = a[0]; win[2] = a[1]; win[3] = a[2]; win[4] = a[3];
            win[5] = b[0]; win[6] = b[1]; win[7] = b[2]; win[8] = b[3];
            win[9] = rp[9];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float w = wt[((ci * 3 + ky) * 3 + kx) * COUT + co];
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[co][p] = fmaf(w, win[p + kx], acc[co][p]);
                }
            }
        }
        if (ci + 1 < CIN) {
            commit((ci + 1) & 1);
            __syncthreads();
        }
    }
    const int gy = ty0 + row0, gx = tx0 + lx * P;
    if (gy < H && gx < W) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float* o = out + ((size_t)co * H + gy) * W + gx;
            *reinterpret_cast<f32x4*>(o) = f32x4{acc[co][0], acc[co][1], acc[co][2], acc[co][3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[co][4], acc[co][5], acc[co][6], acc[co][7]};
        }
    }
}

// V2: packed FMAs over cout PAIRS: acc pair (co, co+1) of one pixel += weight pair (SGPR pair, natural) * broadcast(input),
// the input being either half of an aligned VGPR pair (op_sel), so no register moves at all: 144 v_pk_fma_f32 per (ci, ky).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const unsigned long long __attribute__((address_space(4))) cu64;
__device__ __forceinline__ void pk_lo(f32x2& acc, unsigned long long w, f32x2 in) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w), "v"(in));
}
__device__ __forceinline__ void pk_hi(f32x2& acc, unsigned long long w, f32x2 in) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(in));
}

template <int CIN, int COUT, int WPS>
__global__ __launch_bounds__(256, WPS) void vdir2_kernel(const float* __restrict__ x, const float* wt_, const float* __restrict__ bias,
                                                       float* __restrict__ out, int H, int W, int tiles_x, int tiles) {
    static_assert(COUT % 2 == 0, "cout pairs");
    constexpr int CP = COUT / 2;
    __shared__ __attribute__((aligned(16))) float s_in[2][ROWS * RS];
    cu64* wt = (cu64*)wt_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const int tile = blockIdx.x;
    if (tile >= tiles) return;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    f32x4 pre[NPRE];
    auto issue = [&](int ci) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                const int gy = ty0 + r - 1, gx = tx0 - XOFF + 4 * sg;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4*>(x + ((size_t)ci * H + gy) * W + gx);
            }
            pre[k] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                *reinterpret_cast<f32x4*>(&s_in[buf][r * RS + 4 * sg]) = pre[k];
            }
        }
    };
    f32x2 acc[P][CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const f32x2 b = {bias[2 * c], bias[2 * c + 1]};
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][c] = b;
    }
    issue(0);
    commit(0);
    __syncthreads();
    const int row0 = wave * 8 + ly, col0 = XOFF - 1 + lx * P;
    for (int ci = 0; ci < CIN; ++ci) {
        if (ci + 1 < CIN) issue(ci + 1);
        const float* pl = s_in[ci & 1];
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const float* rp = pl + (row0 + ky) * RS + col0;
            // window values j = 0..9 (x0-1 .. x0+8) as aligned pairs: pr[0] = (junk, 0), pr[1] = (1,2), ... pr[4] = (7,8), pr[5] = (9, junk)
            f32x2 pr[6];
            pr[0] = *reinterpret_cast<const f32x2*>(rp - 1);
            const f32x4 a = *reinterpret_cast<const f32x4*>(rp + 1), b = *reinterpret_cast<const f32x4*>(rp + 5);
            pr[1] = f32x2{a[0], a[1]}; pr[2] = f32x2{a[2], a[3]}; pr[3] = f32x2{b[0], b[1]}; pr[4] = f32x2{b[2], b[3]};
            pr[5] = *reinterpret_cast<const f32x2*>(rp + 9);
            cu64* wp = wt + ((ci * 3 + ky) * 3) * CP;
            unsigned long long w[3 * CP];
#pragma unroll
            for (int i = 0; i < 3 * CP; ++i) w[i] = wp[i];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int j = p + kx;
#pragma unroll
                    for (int c = 0; c < CP; ++c) {
                        if (j & 1) pk_lo(acc[p][c], w[kx * CP + c], pr[(j + 1) / 2]);
                        else pk_hi(acc[p][c], w[kx * CP + c], pr[j / 2]);
                    }
                }
        }
        if (ci + 1 < CIN) {
            commit((ci + 1) & 1);
            __syncthreads();
        }
    }
    const int gy = ty0 + row0, gx = tx0 + lx * P;
    if (gy < H && gx < W) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* o = out + ((size_t)(2 * c + h) * H + gy) * W + gx;
                *reinterpret_cast<f32x4*>(o) = f32x4{acc[0][c][h], acc[1][c][h], acc[2][c][h], acc[3][c][h]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4][c][h], acc[5][c][h], acc[6][c][h], acc[7][c][h]};
            }
    }
}

// V3: V2 + the next step's window and weights are fetched under the current step's FMAs (two register sets; an empty asm
// "touches" the current set first so that the compiler's s_waitcnt lgkmcnt(0) lands BEFORE the next set's loads are issued).
template <int CP> struct StepRegs { f32x2 pr[6]; unsigned long long w[3 * CP]; };

template <int CP, int ABL = 0>
__device__ __forceinline__ void step_load(StepRegs<CP>& r, const float* rp, cu64* wp) {
    if constexpr (ABL >= 2) { asm volatile("" : "+v"(r.pr[0]), "+v"(r.pr[5])); return; }
    if constexpr (ABL == 1) {
#pragma unroll
        for (int i = 0; i < 3 * CP; ++i) r.w[i] = wp[i];
        return;
    }
    r.pr[0] = *reinterpret_cast<const f32x2*>(rp - 1);
    const f32x4 a = *reinterpret_cast<const f32x4*>(rp + 1), b = *reinterpret_cast<const f32x4*>(rp + 5);
    r.pr[1] = f32x2{a[0], a[1]}; r.pr[2] = f32x2{a[2], a[3]}; r.pr[3] = f32x2{b[0], b[1]}; r.pr[4] = f32x2{b[2], b[3]};
    r.pr[5] = *reinterpret_cast<const f32x2*>(rp + 9);
#pragma unroll
    for (int i = 0; i < 3 * CP; ++i) r.w[i] = wp[i];
}
template <int CP>
__device__ __forceinline__ void step_touch(const StepRegs<CP>& r) {
    asm volatile("" ::"v"(r.pr[0]), "v"(r.pr[1]), "v"(r.pr[3]), "v"(r.pr[5]), "s"(r.w[0]), "s"(r.w[3 * CP - 1]), "s"(r.w[3 * CP / 2]));
}
template <int CP>
__device__ __forceinline__ void step_fma(f32x2 (&acc)[P][CP], const StepRegs<CP>& r) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int j = p + kx;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                if (j & 1) pk_lo(acc[p][c], r.w[kx * CP + c], r.pr[(j + 1) / 2]);
                else pk_hi(acc[p][c], r.w[kx * CP + c], r.pr[j / 2]);
            }
        }
}

template <int CIN, int COUT, int WPS, int ABL>
__global__ __launch_bounds__(256, WPS) void vdir3_kernel(const float* __restrict__ x, const float* wt_, const float* __restrict__ bias,
                                                       float* __restrict__ out, int H, int W, int tiles_x, int tiles) {
    static_assert(COUT % 2 == 0 && CIN % 2 == 0, "pairs");
    constexpr int CP = COUT / 2;
    __shared__ __attribute__((aligned(16))) float s_in[2][ROWS * RS];
    cu64* wt = (cu64*)wt_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const int tile = blockIdx.x;
    if (tile >= tiles) return;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    f32x4 pre[NPRE];
    auto issue = [&](int ci) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                const int gy = ty0 + r - 1, gx = tx0 - XOFF + 4 * sg;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4*>(x + ((size_t)ci * H + gy) * W + gx);
            }
            pre[k] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                *reinterpret_cast<f32x4*>(&s_in[buf][r * RS + 4 * sg]) = pre[k];
            }
        }
    };
    f32x2 acc[P][CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const f32x2 b = {bias[2 * c], bias[2 * c + 1]};
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][c] = b;
    }
    issue(0);
    commit(0);
    __syncthreads();
    issue(1);
    const int row0 = wave * 8 + ly, col0 = XOFF - 1 + lx * P;
    const float* b0 = &s_in[0][row0 * RS + col0];
    const float* b1 = &s_in[1][row0 * RS + col0];
    StepRegs<CP> ra = {}, rb = {};
    step_load<CP, ABL>(ra, b0, wt);
#define SB() __builtin_amdgcn_sched_barrier(0)
    for (int ci = 0; ci < CIN; ci += 2) {
        cu64* wp = wt + ci * 9 * CP;
        // (ci, 0): set A ; prefetch (ci, 1) -> B
        step_touch<CP>(ra); SB(); step_load<CP, ABL>(rb, b0 + RS, wp + 3 * CP); SB(); step_fma<CP>(acc, ra); SB();
        // (ci, 1): set B ; prefetch (ci, 2) -> A
        step_touch<CP>(rb); SB(); step_load<CP, ABL>(ra, b0 + 2 * RS, wp + 6 * CP); SB(); step_fma<CP>(acc, rb); SB();
        // plane ci+1 -> buffer 1, then (ci, 2): set A ; prefetch (ci+1, 0) -> B
        if constexpr (ABL == 0) { commit(1); __syncthreads(); if (ci + 2 < CIN) issue(ci + 2); }
        step_touch<CP>(ra); SB(); step_load<CP, ABL>(rb, b1, wp + 9 * CP); SB(); step_fma<CP>(acc, ra); SB();
        // (ci+1, 0): set B ; prefetch (ci+1, 1) -> A
        step_touch<CP>(rb); SB(); step_load<CP, ABL>(ra, b1 + RS, wp + 12 * CP); SB(); step_fma<CP>(acc, rb); SB();
        // (ci+1, 1): set A ; prefetch (ci+1, 2) -> B
        step_touch<CP>(ra); SB(); step_load<CP, ABL>(rb, b1 + 2 * RS, wp + 15 * CP); SB(); step_fma<CP>(acc, ra); SB();
        // plane ci+2 -> buffer 0, then (ci+1, 2): set B ; prefetch (ci+2, 0) -> A
        if (ci + 2 < CIN) {
            if constexpr (ABL == 0) { commit(0); __syncthreads(); if (ci + 3 < CIN) issue(ci + 3); }
            step_touch<CP>(rb); SB(); step_load<CP, ABL>(ra, b0, wp + 18 * CP); SB();
        }
        step_fma<CP>(acc, rb); SB();
    }
    const int gy = ty0 + row0, gx = tx0 + lx * P;
    if (gy < H && gx < W) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* o = out + ((size_t)(2 * c + h) * H + gy) * W + gx;
                *reinterpret_cast<f32x4*>(o) = f32x4{acc[0][c][h], acc[1][c][h], acc[2][c][h], acc[3][c][h]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4][c][h], acc[5][c][h], acc[6][c][h], acc[7][c][h]};
            }
    }
}

// V4: the same tiling on v_mfma_f32_4x4x1_16b_f32 with LANE = PIXEL: A = a quad of weights (4 couts of one (ci, tap)), broadcast to
// all 16 blocks with cbsz = 4 / abid (so ONE VGPR holds 16 quads and the 108 weights of an input channel are two registers),
// B = the input value of the lane's pixel (one ds_read_b32 per (row, kx), reused by 3 ky x 3 cout groups = 9 MFMAs),
// D: lane = pixel, 4 regs = 4 couts.  N granularity 4: a 12-channel layer issues exactly 12/16 of the 16x16x4 kernel's MFMA cycles,
// and the K loop has no VALU instruction at all.
constexpr int RW = 8;    // rows per wave
template <int CIN, int COUT, int WPS>
__global__ __launch_bounds__(256, WPS) void mconv_kernel(const float* __restrict__ x, const float* __restrict__ wimg, const float* __restrict__ bias,
                                                       float* __restrict__ out, int H, int W, int tiles_x, int tiles) {
    static_assert(COUT % 4 == 0 && 9 * (COUT / 4) <= 32, "two weight registers per input channel");
    constexpr int NG = COUT / 4;
    __shared__ __attribute__((aligned(16))) float s_in[2][ROWS * RS];
    __shared__ float s_w[CIN * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    if (tile >= tiles) return;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    for (int i = tid; i < CIN * 128; i += 256) s_w[i] = wimg[i];
    f32x4 pre[NPRE];
    auto issue = [&](int ci) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                const int gy = ty0 + r - 1, gx = tx0 - XOFF + 4 * sg;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4*>(x + ((size_t)ci * H + gy) * W + gx);
            }
            pre[k] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                *reinterpret_cast<f32x4*>(&s_in[buf][r * RS + 4 * sg]) = pre[k];
            }
        }
    };
    f32x4 acc[RW][NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        const f32x4 b = {bias[4 * n], bias[4 * n + 1], bias[4 * n + 2], bias[4 * n + 3]};
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r][n] = b;
    }
    issue(0);
    commit(0);
    __syncthreads();
    const int off0 = (wave * RW) * RS + XOFF - 1 + lane;
    for (int ci = 0; ci < CIN; ++ci) {
        if (ci + 1 < CIN) issue(ci + 1);
        const float* pl = s_in[ci & 1] + off0;
        const float wa = s_w[ci * 128 + lane], wb = s_w[ci * 128 + 64 + lane];
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float b = pl[rr * RS + kx];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = rr - ky;
                    if (r < 0 || r >= RW) continue;
#pragma unroll
                    for (int n = 0; n < NG; ++n) {
                        constexpr int dummy = 0;
                        const int slot = (ky * 3 + kx) * NG + n;
                        // abid must be an immediate: enumerate
                        #define MF(S) case S: acc[r][n] = (S < 16) ? __builtin_amdgcn_mfma_f32_4x4x1f32(wa, b, acc[r][n], 4, S & 15, 0) \
                                                                  : __builtin_amdgcn_mfma_f32_4x4x1f32(wb, b, acc[r][n], 4, S & 15, 0); break;
                        switch (slot) {
                            MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(8) MF(9) MF(10) MF(11) MF(12) MF(13) MF(14) MF(15)
                            MF(16) MF(17) MF(18) MF(19) MF(20) MF(21) MF(22) MF(23) MF(24) MF(25) MF(26) MF(27) MF(28) MF(29) MF(30) MF(31)
                        }
                        #undef MF
                    }
                }
            }
        }
        if (ci + 1 < CIN) {
            commit((ci + 1) & 1);
            __syncthreads();
        }
    }
    const int gx = tx0 + lane;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int gy = ty0 + wave * RW + r;
        if (gy < H && gx < W) {
#pragma unroll
            for (int n = 0; n < NG; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) out[((size_t)(4 * n + i) * H + gy) * W + gx] = acc[r][n][i];
        }
    }
}

// V5: as V4 but A = pixels (lane = pixel), B = weights: D regs = 4 CONSECUTIVE PIXELS of cout 4n + (lane & 3) -> float4 stores.
// B has no cbsz broadcast; BLGP = 4 + g broadcasts the 16-lane group g to all four groups, so a weight register holds 4 quads
// (one per 16-lane group), each replicated over the group's four 4-lane blocks.  All 27 quads x CIN stay in registers.
template <int CIN, int COUT, int WPS>
__global__ __launch_bounds__(256, WPS) void mconv5_kernel(const float* __restrict__ x, const float* __restrict__ wq, const float* __restrict__ bias,
                                                        float* __restrict__ out, int H, int W, int tiles_x, int tiles) {
    constexpr int NG = COUT / 4, NQD = 9 * NG, NWR = (NQD + 3) / 4;     // quads per ci, weight registers per ci
    __shared__ __attribute__((aligned(16))) float s_in[2][ROWS * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    if (tile >= tiles) return;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    float wr[CIN][NWR];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int v = 0; v < NWR; ++v) {
            const int q = 4 * v + (lane >> 4);                           // quad index (tap * NG + n)
            wr[ci][v] = q < NQD ? wq[(ci * NQD + q) * 4 + (lane & 3)] : 0.f;
        }
    f32x4 pre[NPRE];
    auto issue = [&](int ci) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                const int gy = ty0 + r - 1, gx = tx0 - XOFF + 4 * sg;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4*>(x + ((size_t)ci * H + gy) * W + gx);
            }
            pre[k] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            if (s < SLOTS) {
                const int r = s / SEGS, sg = s - r * SEGS;
                *reinterpret_cast<f32x4*>(&s_in[buf][r * RS + 4 * sg]) = pre[k];
            }
        }
    };
    f32x4 acc[RW][NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        const float b = bias[4 * n + (lane & 3)];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r][n] = f32x4{b, b, b, b};
    }
    issue(0);
    commit(0);
    __syncthreads();
    const int off0 = (wave * RW) * RS + XOFF - 1 + lane;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        if (ci + 1 < CIN) issue(ci + 1);
        const float* pl = s_in[ci & 1] + off0;
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float a = pl[rr * RS + kx];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = rr - ky;
                    if (r < 0 || r >= RW) continue;
#pragma unroll
                    for (int n = 0; n < NG; ++n) {
                        const int q = (ky * 3 + kx) * NG + n;
                        const float w = wr[ci][q >> 2];
                        switch (q & 3) {
                            case 0: acc[r][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, acc[r][n], 0, 0, 4); break;
                            case 1: acc[r][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, acc[r][n], 0, 0, 5); break;
                            case 2: acc[r][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, acc[r][n], 0, 0, 6); break;
                            default: acc[r][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, acc[r][n], 0, 0, 7); break;
                        }
                    }
                }
            }
        }
        if (ci + 1 < CIN) {
            commit((ci + 1) & 1);
            __syncthreads();
        }
    }
    const int gx = tx0 + (lane >> 2) * 4;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int gy = ty0 + wave * RW + r;
        if (gy < H && gx < W) {
#pragma unroll
            for (int n = 0; n < NG; ++n) *reinterpret_cast<f32x4*>(out + ((size_t)(4 * n + (lane & 3)) * H + gy) * W + gx) = acc[r][n];
        }
    }
}

int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 720, W = argc > 2 ? atoi(argv[2]) : 1280;
    constexpr int C = 12;
    const size_t n = (size_t)C * H * W;
    std::vector<float> hx(n), hw(C * C * 9), hwt(C * 9 * C), hb(C), ho(n);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < C; ++ci)
            for (int t = 0; t < 9; ++t) hwt[(ci * 9 + t) * C + co] = hw[(co * C + ci) * 9 + t];
    std::vector<float> himg(C * 128, 0.f);
    for (int ci = 0; ci < C; ++ci)
        for (int t = 0; t < 9; ++t)
            for (int co = 0; co < C; ++co) himg[ci * 128 + (t * (C / 4) + co / 4) * 4 + (co & 3)] = hw[(co * C + ci) * 9 + t];
    std::vector<float> hq(C * 27 * 4 * (C / 12), 0.f);
    for (int ci = 0; ci < C; ++ci)
        for (int t = 0; t < 9; ++t)
            for (int co = 0; co < C; ++co) hq[((ci * 9 + t) * (C / 4) + co / 4) * 4 + (co & 3)] = hw[(co * C + ci) * 9 + t];
    float* dq; hipMalloc(&dq, hq.size() * 4); hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    float *dx, *dw, *db, *dout, *dimg;
    hipMalloc(&dimg, himg.size() * 4); hipMemcpy(dimg, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dw, hwt.size() * 4); hipMalloc(&db, C * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tiles = tiles_x * tiles_y;
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    auto launch = [&]() {
        if (variant == 0) vdir_kernel<C, C, 2><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 3) vdir_kernel<C, C, 3><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 4) vdir_kernel<C, C, 4><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 1) vdir2_kernel<C, C, 2><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 2) vdir2_kernel<C, C, 3><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 5) vdir3_kernel<C, C, 2, 0><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 6) vdir3_kernel<C, C, 3, 0><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else if (variant == 9) mconv_kernel<C, C, 2><<<tiles, 256>>>(dx, dimg, db, dout, H, W, tiles_x, tiles);
        else if (variant == 10) mconv_kernel<C, C, 3><<<tiles, 256>>>(dx, dimg, db, dout, H, W, tiles_x, tiles);
        else if (variant == 11) mconv5_kernel<C, C, 2><<<tiles, 256>>>(dx, dq, db, dout, H, W, tiles_x, tiles);
        else if (variant == 7) vdir3_kernel<C, C, 2, 1><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
        else vdir3_kernel<C, C, 2, 2><<<tiles, 256>>>(dx, dw, db, dout, H, W, tiles_x, tiles);
    };
    launch();
    hipDeviceSynchronize();
    hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int s = 0; s < 4000; ++s) {
        const int co = rand() % C, y = (s < 200) ? (s % 2 ? H - 1 : 0) : rand() % H, xx = (s < 400 && s >= 200) ? (s % 2 ? W - 1 : 0) : rand() % W;
        double r = hb[co];
        for (int ci = 0; ci < C; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xc = xx + kx - 1;
                    if (yy < 0 || yy >= H || xc < 0 || xc >= W) continue;
                    r += (double)hw[(co * C + ci) * 9 + ky * 3 + kx] * hx[((size_t)ci * H + yy) * W + xc];
                }
        maxerr = fmax(maxerr, fabs(r - ho[((size_t)co * H + y) * W + xx]));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 50;
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, flops = 2.0 * C * C * 9 * H * W;
    printf("vdir v%d %dx%d C=%d: %d blocks, %.2f us/launch, %.1f TFLOP/s (%.1f%% of 157.3), %.2f TB/s algorithmic, max|err| %.3g\n", variant, H, W, C, tiles, us,
           flops / us / 1e6, flops / us / 1e6 / 157.3 * 100, 2.0 * n * 4 / us / 1e6, maxerr);
    return 0;
}
