// wgrad.hip -- weight + bias gradient of the decoder convolutions as an MFMA 16x16x4 GEMM with K = pixels.
//
//   dw[co][n] = sum_{b, p} g[b][co][p] * a[b][ci(n)][p + tap(n) - pad],   n = ci*T + tap   (the OIHW flattening)
//   db[co]    = the extra column n = Cin*T whose B operand is the constant 1
// (autograd's backward of F.conv2d wrt weight/bias at lib/quant_ops.py:39-41; call sites as in conv.hip.)
//
// GEMM view: M = cout (MTW tiles of 16), N = (ci,tap)+bias column (NTW tiles of 16), K = pixels of an 8x32 spatial tile
// (4 per MFMA: lane k-slot kq <-> pixel 4*step + kq).  A block walks spatial tiles in a grid-stride loop with the
// accumulators in registers; its 4 waves split each tile's 64 K-steps; every wave writes ONE slab at the end and
// reduce_slabs finishes the sum deterministically (no atomics, no memset).
//   LDS: s_g  [MTW*16][256 + 2]   gradient tile, cout-major, stride == 2 (mod 32): conflict-free A reads
//        s_in [<=16 planes][halo tile] exactly as in conv.hip (same prologues: plain / affine / gelu-affine)
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;
constexpr int MAXPL = 16;
constexpr int CSG = TH * TW + 2;

template <int KS> struct Geo {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int ROWS = TH + 2 * PAD;
    static constexpr int RS = TW + 2 * PAD;
    static constexpr int PLANE_RAW = ROWS * RS;
    static constexpr int PLANE = ((PLANE_RAW - 16 + 31) / 32) * 32 + 16;
    static constexpr int T = KS * KS;
};

struct WArgs {
    bnerv_wgrad_desc d;
    float* slab;
    int tiles_x, tiles_y, n_mgroups, n_ngroups, ncols;   // ncols = Cin*T + 1
};

template <int IN>
__device__ __forceinline__ float load_x(const bnerv_wgrad_desc& d, int b, int ci, int gy, int gx) {
    const size_t idx = (((size_t)b * d.Cin + ci) * d.H + gy) * (size_t)d.W + gx;
    const float v = d.x[idx];
    if constexpr (IN == BNERV_IN_AFFINE) return v * (1.0f + d.scale[b * d.Cin + ci]) + d.shift[b * d.Cin + ci];
    if constexpr (IN == BNERV_IN_GELU_AFFINE) return gelu_f(v) * (1.0f + d.scale[b * d.Cin + ci]) + d.shift[b * d.Cin + ci];
    return v;
}

template <int GM>
__device__ __forceinline__ float load_g(const bnerv_wgrad_desc& d, int b, int co, int gy, int gx) {
    if constexpr (GM == BNERV_IN_TANHGRAD) {
        const size_t idx = (((size_t)b * d.Cout + co) * d.H + gy) * (size_t)d.W + gx;
        const float t = 2.0f * d.gaux[idx] - 1.0f;
        return d.g[idx] * 0.5f * (1.0f - t * t);
    } else {
        const int s = d.g_s;
        if (s == 1) return d.g[(((size_t)b * d.Cout + co) * d.H + gy) * (size_t)d.W + gx];
        const int s2 = s * s, c = co / s2, rem = co - c * s2, i = rem / s, j = rem - i * s;
        return d.g[(((size_t)b * (d.Cout / s2) + c) * (size_t)(d.H * s) + (size_t)(gy * s + i)) * (size_t)(d.W * s) + (size_t)(gx * s + j)];
    }
}

template <int KS, int IN, int GM, int MTW, int NTW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WArgs wa) {
    using G = Geo<KS>;
    const bnerv_wgrad_desc& d = wa.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                          // MTW*16*CSG
    float* s_in = smem + MTW * 16 * CSG;        // MAXPL*PLANE

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int mg = blockIdx.y / wa.n_ngroups, ngp = blockIdx.y % wa.n_ngroups;
    const int co_base = mg * MTW * 16;
    const int n_base = ngp * NTW * 16;
    const int Cin = d.Cin, Cout = d.Cout, H = d.H, W = d.W;
    const int nW = Cin * G::T;                  // weight columns; column nW is the bias column
    const int ci_lo = min(n_base, nW - 1) / G::T;
    const int ci_hi = min(n_base + NTW * 16 - 1, nW - 1) / G::T;
    const int npl = ci_hi - ci_lo + 1;          // <= MAXPL by construction of NTW

    // per-lane B-fragment descriptors for this block's N tiles
    int boff[NTW];
    float bconst[NTW];   // value used when the column is not an LDS read: 1 for the bias column, 0 beyond it
    bool bread[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = n_base + nt * 16 + li;
        if (n < nW) {
            const int ci = n / G::T, tap = n - ci * G::T;
            boff[nt] = (ci - ci_lo) * G::PLANE + (tap / KS) * G::RS + (tap % KS);
            bread[nt] = true; bconst[nt] = 0.f;
        } else {
            boff[nt] = 0; bread[nt] = false; bconst[nt] = (n == nW) ? 1.0f : 0.0f;
        }
    }

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tiles = wa.tiles_x * wa.tiles_y;
    const int total = d.B * tiles;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int b = t / tiles, tile = t - b * tiles;
        const int ty0 = (tile / wa.tiles_x) * TH, tx0 = (tile % wa.tiles_x) * TW;
        __syncthreads();
        for (int idx = tid; idx < MTW * 16 * TH * TW; idx += 256) {
            const int cl = idx >> 8, p = idx & 255;
            const int co = co_base + cl, gy = ty0 + (p >> 5), gx = tx0 + (p & 31);
            float v = 0.f;
            if (co < Cout && gy < H && gx < W) v = load_g<GM>(d, b, co, gy, gx);
            s_g[cl * CSG + p] = v;
        }
        for (int idx = tid; idx < npl * G::PLANE_RAW; idx += 256) {
            const int c = idx / G::PLANE_RAW;
            const int rem = idx - c * G::PLANE_RAW;
            const int r = rem / G::RS, col = rem - r * G::RS;
            const int gy = ty0 + r - G::PAD, gx = tx0 + col - G::PAD;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = load_x<IN>(d, b, ci_lo + c, gy, gx);
            s_in[c * G::PLANE + r * G::RS + col] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            const int p = (wave * 16 + st) * 4 + kq;
            const int poff = (p >> 5) * G::RS + (p & 31);
            float af[MTW], bf[NTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) af[m] = s_g[(m * 16 + li) * CSG + p];
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const float v = s_in[boff[n] + poff];
                bf[n] = bread[n] ? v : bconst[n];
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
    }
    // cross-wave reduction through LDS (fixed order => deterministic), then ONE slab per block:
    // slab[block][co][n]; D layout: lane holds rows (cout) 4*kq..4*kq+3 of column (n) li
    __syncthreads();
    float* s_red = smem;                                      // [4 waves][MTW*16][NTW*16]  (fits: <= 4*64*64 floats only for (4,4); checked on host)
    constexpr int RW = NTW * 16, RSZ = MTW * 16 * RW;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[wave * RSZ + (m * 16 + 4 * kq + r) * RW + n * 16 + li] = acc[m][n][r];
    __syncthreads();
    float* slab = wa.slab + (size_t)blockIdx.x * Cout * wa.ncols;
    for (int idx = tid; idx < RSZ; idx += 256) {
        const int row = idx / RW, colq = idx - row * RW;
        const int co = co_base + row, col = n_base + colq;
        if (co < Cout && col < wa.ncols)
            slab[(size_t)co * wa.ncols + col] = (s_red[idx] + s_red[RSZ + idx]) + (s_red[2 * RSZ + idx] + s_red[3 * RSZ + idx]);
    }
}

// finish: dw[co][n] = sum_slabs, db[co] = column nW.  32 consecutive elements x 8 slab lanes per block: coalesced 128-B rows,
// 8-way parallel over slabs, fixed combination order (deterministic).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slab, int n_slabs, int Cout, int ncols, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, lane = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const int count = Cout * ncols;
    float s = 0.f;
    if (i < count)
        for (int k = lane; k < n_slabs; k += 8) s += slab[(size_t)k * count + i];
    red[lane][e] = s;
    __syncthreads();
    if (lane == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][e];
        const int co = i / ncols, n = i - co * ncols;
        if (n < ncols - 1) dw[(size_t)co * (ncols - 1) + n] = t;
        else if (db) db[co] = t;
    }
}

struct Plan { int mtw, ntw, n_mgroups, n_ngroups, nsplit; };

Plan make_plan(int B, int Cin, int Cout, int H, int W, int k) {
    Plan p;
    const int T = k * k, ncols = Cin * T + 1;
    const int mt = cdiv(Cout, 16), nt = cdiv(ncols, 16);
    if (k == 1) { p.mtw = 4; p.ntw = 1; }
    else if (mt == 1) { p.mtw = 1; p.ntw = 7; }
    else if (mt <= 3) { p.mtw = 3; p.ntw = 7; }
    else { p.mtw = 4; p.ntw = 4; }
    p.n_mgroups = cdiv(mt, p.mtw);
    p.n_ngroups = cdiv(nt, p.ntw);
    const int total_tiles = B * cdiv(H, TH) * cdiv(W, TW);
    const int groups = p.n_mgroups * p.n_ngroups;
    int target = 512 / groups;                   // ~2 blocks per CU in flight over all groups
    if (target < 1) target = 1;
    p.nsplit = total_tiles < target ? total_tiles : target;
    return p;
}

template <int KS, int IN, int GM, int MTW, int NTW>
int launch_w(hipStream_t st, const WArgs& wa, const Plan& p) {
    using G = Geo<KS>;
    constexpr size_t lds_main = (size_t)MTW * 16 * CSG + (size_t)MAXPL * G::PLANE;
    constexpr size_t lds_red = (size_t)4 * MTW * 16 * NTW * 16;
    constexpr size_t lds = (lds_main > lds_red ? lds_main : lds_red) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<KS, IN, GM, MTW, NTW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(p.nsplit, p.n_mgroups * p.n_ngroups);
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, IN, GM, MTW, NTW>), grid, dim3(256), lds, st, wa);
    BNERV_LAUNCH_CHECK("conv_wgrad");
    return BNERV_OK;
}

template <int KS, int IN, int GM>
int launch_shape(hipStream_t st, const WArgs& wa, const Plan& p) {
    if constexpr (KS == 1) {
        return launch_w<1, IN, GM, 4, 1>(st, wa, p);
    } else {
        if (p.mtw == 1) return launch_w<3, IN, GM, 1, 7>(st, wa, p);
        if (p.mtw == 3) return launch_w<3, IN, GM, 3, 7>(st, wa, p);
        return launch_w<3, IN, GM, 4, 4>(st, wa, p);
    }
}

template <int KS>
int launch_modes(hipStream_t st, const WArgs& wa, const Plan& p) {
    const int in = wa.d.in_mode, gm = wa.d.g_mode;
    if (gm == BNERV_IN_TANHGRAD && in == BNERV_IN_PLAIN) return launch_shape<KS, BNERV_IN_PLAIN, BNERV_IN_TANHGRAD>(st, wa, p);
    if (gm == BNERV_IN_PLAIN || gm == BNERV_IN_UNSHUFFLE) {
        if (in == BNERV_IN_PLAIN) return launch_shape<KS, BNERV_IN_PLAIN, BNERV_IN_UNSHUFFLE>(st, wa, p);
        if constexpr (KS == 3) {
            if (in == BNERV_IN_AFFINE) return launch_shape<KS, BNERV_IN_AFFINE, BNERV_IN_UNSHUFFLE>(st, wa, p);
            if (in == BNERV_IN_GELU_AFFINE) return launch_shape<KS, BNERV_IN_GELU_AFFINE, BNERV_IN_UNSHUFFLE>(st, wa, p);
        }
    }
    return bnerv_set_error(BNERV_E_ARG, "conv_wgrad: unsupported (k=%d, in_mode=%d, g_mode=%d)", KS, in, gm);
}

}  // namespace

extern "C" size_t bnerv_conv_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int k) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (k != 1 && k != 3)) return 0;
    const Plan p = make_plan(B, Cin, Cout, H, W, k);
    return (size_t)p.nsplit * Cout * (Cin * k * k + 1) * sizeof(float);
}

extern "C" int bnerv_conv_wgrad(void* stream, const bnerv_wgrad_desc* dp) {
    BNERV_REQUIRE(dp != nullptr, "conv_wgrad: null descriptor");
    WArgs wa;
    wa.d = *dp;
    const bnerv_wgrad_desc& d = wa.d;
    BNERV_REQUIRE(d.k == 1 || d.k == 3, "conv_wgrad: k must be 1 or 3 (got %d)", d.k);
    BNERV_REQUIRE(d.B > 0 && d.Cin > 0 && d.Cout > 0 && d.H > 0 && d.W > 0, "conv_wgrad: bad dims");
    BNERV_REQUIRE(d.x && d.g && d.dw && d.ws, "conv_wgrad: null tensor");
    BNERV_REQUIRE(d.g_s >= 1 && d.Cout % (d.g_s * d.g_s) == 0, "conv_wgrad: bad g_s");
    if (d.in_mode == BNERV_IN_AFFINE || d.in_mode == BNERV_IN_GELU_AFFINE) BNERV_REQUIRE(d.scale && d.shift, "conv_wgrad: affine prologue needs scale/shift");
    if (d.g_mode == BNERV_IN_TANHGRAD) BNERV_REQUIRE(d.gaux && d.g_s == 1, "conv_wgrad: tanh-grad needs gaux");
    const Plan p = make_plan(d.B, d.Cin, d.Cout, d.H, d.W, d.k);
    const size_t need = bnerv_conv_wgrad_ws_bytes(d.B, d.Cin, d.Cout, d.H, d.W, d.k);
    if (d.ws_bytes < need) return bnerv_set_error(BNERV_E_WS, "conv_wgrad: workspace %zu < %zu", d.ws_bytes, need);
    wa.slab = reinterpret_cast<float*>(d.ws);
    wa.tiles_x = cdiv(d.W, TW);
    wa.tiles_y = cdiv(d.H, TH);
    wa.n_mgroups = p.n_mgroups;
    wa.n_ngroups = p.n_ngroups;
    wa.ncols = d.Cin * d.k * d.k + 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = d.k == 1 ? launch_modes<1>(st, wa, p) : launch_modes<3>(st, wa, p);
    if (rc != BNERV_OK) return rc;
    const int count = d.Cout * wa.ncols;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(cdiv(count, 32)), dim3(256), 0, st, wa.slab, p.nsplit, d.Cout, wa.ncols, d.dw, d.db);
    BNERV_LAUNCH_CHECK("wgrad_finish");
    return BNERV_OK;
}
