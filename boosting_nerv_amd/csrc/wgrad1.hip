// wgrad1.hip -- weight / bias gradient of a POINTWISE (k = 1) convolution as the plain GEMM it is:
//     dW[co][ci] = sum_{b, p} g[b][co][p] * x[b][ci][p],     db[co] = sum_{b, p} g[b][co][p]
// (reference call sites: the ConvNeXt block's pwconv1 / pwconv2 of HNeRV_Boost's encoder, model_blocks.py:245-258, and every other
// 1x1 CustomConv2d whose weight gradient lib/quant_ops.py:39-41's backward produces).  Both operands are NCHW planes: the contraction
// index (pixels) is the CONTIGUOUS one on both sides, so neither needs a tile in LDS -- a lane's float4 of four consecutive pixels is four
// K steps of its row for v_mfma_f32_16x16x4_f32 (A[m = li][k = kq] = element j of the float4 at pixel 16 s + 4 kq, for j = 0..3: the same
// pixel order on both operands).  Round 5's general kernel served these layers 16 columns per block and re-read the gradient once per
// column group (17 x for the 256-column pwconv2: 160-180 us at 64 x 216 x 384 against ~30 us of traffic).
//
// Block = 4 waves = up to 64 couts x 64 cins (4 x 4 MFMA tiles, 64 accumulator registers per wave) over ONE slab of pixels; wave w takes
// every fourth 16-pixel step of the slab; the four waves add their accumulators one after the other into one 16 KB LDS area, and the block
// writes its [rows][cols] part of slab z coalesced.  Slabs [z][Cout][Cin + 1] (bias gradient = last column) are summed by the usual
// finish / deferred reduction (wgrad.hip, sidejob.h): deterministic, no atomics.
#include "conv_common.h"
#include "sidejob.h"

namespace {
using namespace bnerv_conv;

struct W1Args {
    const float* x; const float* g; float* slab;
    int B, Cin, Cout, HW;
    int n_mg, n_ng;        // groups of 64 rows / 64 columns
    int chunk;             // pixels per slab (multiple of 64)
};

__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(const W1Args a) {
    __shared__ __attribute__((aligned(16))) float s_acc[64 * 65];
    __shared__ float s_bias[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x, mg = grp / a.n_ng, ng = grp - mg * a.n_ng, z = blockIdx.y;
    const int m0 = mg * 64, n0 = ng * 64;
    const int HW = a.HW;
    const int p_beg = z * a.chunk, p_end = min(HW, p_beg + a.chunk);
    // row validity per MFMA tile: rows beyond Cout / Cin load zeros through the out-of-range offset
    const unsigned plane = (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.g, 0, (unsigned)((size_t)a.B * a.Cout * HW * 4));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, 0, (unsigned)((size_t)a.B * a.Cin * HW * 4));
    unsigned goff[4], xoff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int co = m0 + t * 16 + li, ci = n0 + t * 16 + li;
        goff[t] = co < a.Cout ? (unsigned)co * plane + (unsigned)kq * 16u : OOB;
        xoff[t] = ci < a.Cin ? (unsigned)ci * plane + (unsigned)kq * 16u : OOB;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = ng == a.n_ng - 1;
    for (int b = 0; b < a.B; ++b) {
        const unsigned gb = (unsigned)((size_t)b * a.Cout * HW * 4), xb = (unsigned)((size_t)b * a.Cin * HW * 4);
        // the wave's steps: pixels p_beg + 16 (wave + 4 s) .. + 15; a lane's float4 starts at + 4 kq (HW % 4 == 0: a float4 is inside the plane or past p_end)
        int p = p_beg + 16 * wave;
        f32x4 ga[4], xa[4];
        auto load = [&](int pp, f32x4* gv, f32x4* xv) __attribute__((always_inline)) {
            const bool in = pp + 4 * kq < p_end;
            const unsigned po = (unsigned)pp * 4u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gv[t] = bload(rg, in ? goff[t] + po : OOB, gb);
                xv[t] = bload(rx, in ? xoff[t] + po : OOB, xb);
            }
        };
        if (p < p_end) load(p, ga, xa);
        while (p < p_end) {
            f32x4 gn[4], xn[4];
            const int pn = p + 64;
            if (pn < p_end) load(pn, gn, xn);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[mt][j], xa[nt][j], acc[mt][nt], 0, 0, 0);
            if (want_bias) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) bsum[mt] += (ga[mt][0] + ga[mt][1]) + (ga[mt][2] + ga[mt][3]);
            }
            if (pn < p_end) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { ga[t] = gn[t]; xa[t] = xn[t]; }
            }
            p = pn;
        }
    }
    // ---- the four waves' accumulators, one after the other, into one LDS tile [row 0..63][col 0..63] (row stride 65)
    //      D fragment: lane (li, kq) holds rows 4 kq + e (e = 0..3) of column li of tile (mt, nt)
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* q = s_acc + (mt * 16 + 4 * kq + e) * 65 + nt * 16 + li;
                        *q = (w == 0) ? acc[mt][nt][e] : *q + acc[mt][nt][e];
                    }
        }
        __syncthreads();
    }
    if (want_bias) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float v = bsum[mt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) s_bias[wave][mt * 16 + li] = v;
        }
        __syncthreads();
    }
    // ---- this block's part of slab z
    const int ncols = a.Cin + 1;
    float* sl = a.slab + (size_t)z * a.Cout * ncols;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (m0 + r < a.Cout && n0 + c < a.Cin) sl[(size_t)(m0 + r) * ncols + n0 + c] = s_acc[r * 65 + c];
    }
    if (want_bias && tid < 64 && m0 + tid < a.Cout)
        sl[(size_t)(m0 + tid) * ncols + a.Cin] = (s_bias[0][tid] + s_bias[1][tid]) + (s_bias[2][tid] + s_bias[3][tid]);
}

struct W1Plan { int n_mg, n_ng, nz, chunk; };

static W1Plan w1_plan(int B, int Cin, int Cout, int HW) {
    W1Plan p;
    p.n_mg = cdiv(Cout, 64);
    p.n_ng = cdiv(Cin, 64);
    const int groups = p.n_mg * p.n_ng;
    // ~768 blocks (three per CU) when the image allows it; a slab holds at least 256 pixels (four 16-pixel steps per wave)
    int nz = 768 / groups;
    if (nz < 1) nz = 1;
    int chunk = cdiv(cdiv(HW, nz), 64) * 64;
    if (chunk < 256) chunk = 256;
    p.chunk = chunk;
    p.nz = cdiv(HW, chunk);
    (void)B;
    return p;
}

static bool w1_ok(const bnerv_wgrad_desc& d) {
    static const bool off = [] { const char* e = getenv("BNERV_WGRAD1"); return e && e[0] == '0'; }();   // A/B switch
    if (off || d.k != 1 || d.g_s != 1) return false;
    if (d.in_mode != BNERV_IN_PLAIN || !(d.g_mode == BNERV_IN_PLAIN || d.g_mode == BNERV_IN_UNSHUFFLE)) return false;
    const size_t HW = (size_t)d.H * d.W;
    // (measured: 36 x 64 = 2304 pixels is faster on the general kernel, 12.6 against 18.7 us; 72 x 128 is faster here, 20 against 29)
    if ((HW & 3) || HW < 4096 || (reinterpret_cast<uintptr_t>(d.x) & 15) || (reinterpret_cast<uintptr_t>(d.g) & 15)) return false;
    if (d.Cin < 16 || d.Cout < 16) return false;               // (the lean kernel serves the few-channel layers: 1x1 heads)
    const size_t big = (size_t)d.B * (d.Cin > d.Cout ? d.Cin : d.Cout) * HW * 4;
    return big < LEAN_MAX_BYTES;
}

}  // namespace

// slabs this kernel writes for a k = 1 layer (0: not its layer) -- bnerv_conv_wgrad_ws_bytes sizes the workspace with it
int bnerv_wgrad1x1_slabs(const bnerv_wgrad_desc& d) {
    if (!w1_ok(d)) return 0;
    return w1_plan(d.B, d.Cin, d.Cout, d.H * d.W).nz;
}

// 1: not this kernel's layer; BNERV_OK: slabs written to d.ws, *n_slabs of them
int bnerv_wgrad1x1_try(hipStream_t st, const bnerv_wgrad_desc& d, int* n_slabs) {
    if (!w1_ok(d)) return 1;
    const int HW = d.H * d.W;
    const W1Plan p = w1_plan(d.B, d.Cin, d.Cout, HW);
    if ((size_t)p.nz * d.Cout * (d.Cin + 1) * sizeof(float) > d.ws_bytes) return 1;
    W1Args a{d.x, d.g, reinterpret_cast<float*>(d.ws), d.B, d.Cin, d.Cout, HW, p.n_mg, p.n_ng, p.chunk};
    hipLaunchKernelGGL(wgrad1x1_kernel, dim3(p.n_mg * p.n_ng, p.nz), dim3(256), 0, st, a);
    BNERV_LAUNCH_CHECK("wgrad1x1");
    *n_slabs = p.nz;
    return BNERV_OK;
}
