// head3.hip -- the 3x3 output head with few outputs (HNeRV_Boost: head_layer = 3x3 conv 38 -> 3 + OutImg tanh, reference model_hnerv.py:214,
// :247, model_blocks.py:57-63) as two streaming VALU kernels.
//
// With 3 output channels an implicit GEMM uses 3 of its 16 N columns (forward) or 3 of its K quads (data gradient): round 5's timelines show
// 202 us for the forward on the split-bf16 kernel and 146 us for the data gradient on the f32 kernel at 1080x1920, against ~50 us of traffic
// each (one 38-channel tensor crosses HBM).  Per pixel the layer is 1026 multiply-adds either way -- 4.2 GFLOP per launch, ~40 us of plain
// v_fma at the chip's vector rate -- so:
//   forward        lane = 4 consecutive pixels of a row x 3 outputs (12 accumulators); the input tile (8 channels at a time, 18 x 72 floats
//                  each, haloed) in LDS, next chunk prefetched into registers under the arithmetic; the weights sit in LDS as a [ci][27 (+1)] table
//                  and reach the lanes as seven uniform (broadcast) 16-byte reads per input channel; epilogue bias + tanh * 0.5 + 0.5;
//   data gradient  lane = 4 consecutive pixels; the 3-channel gradient tile (tanh-gradient prologue applied while staging when the caller
//                  passes the raw gradient + the image) in LDS, a lane's 3 x 3 x 6 window in registers once, then per input channel 27
//                  weights (the same table) x 4 pixels and one 16-byte store.
// Both are exact fp32 (fma chains in a fixed order).  The weight gradient of the same layer runs with swapped roles (ops._HeadTanh.backward).
#include "conv_common.h"

namespace {
using namespace bnerv_conv;

constexpr int HT_H = 16, HT_W = 64;                 // output tile of a block: 16 rows x 64 columns, thread = (row, 4 consecutive columns)
constexpr int HT_RS = HT_W + 8;                     // LDS row: 4 floats of margin on either side (float4-aligned segments)
constexpr int HT_ROWS = HT_H + 2;
constexpr int HT_PLANE = HT_ROWS * HT_RS;           // 1296 floats per channel
constexpr int HT_SEGS = HT_RS / 4;                  // 18 float4 per row
constexpr int HF_CC = 8;                            // input channels per chunk (forward)
constexpr int H3_MAXC = 64, H3_WS = 28;             // weight table in LDS: up to 64 input channels x (3 outputs x 9 taps, padded to 28)

struct H3Args {
    const float* x; const float* w; const float* bias; const float* img; float* out;
    int B, Cin, Cout, H, W, tiles_x;
};

// a lane's six window values of one LDS row: image columns c0 - 1 .. c0 + 4 (LDS columns c0 + 3 .. c0 + 8)
__device__ __forceinline__ void win6(const float* row, float (&v)[6]) {
    v[0] = row[3];
    const f32x4 m = *reinterpret_cast<const f32x4*>(row + 4);
    v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
    v[5] = row[8];
}

// ------------------------------------------------------------------------------------------------------------------ forward
template <int CO>
__global__ __launch_bounds__(256) void head3x3_fwd_kernel(const H3Args a) {
    __shared__ __attribute__((aligned(16))) float s_in[HF_CC * HT_PLANE];        // 41.5 KB
    __shared__ __attribute__((aligned(16))) float s_w[H3_MAXC * H3_WS];          // [ci][o][ky][kx] padded to 28 floats per input channel
    const int tid = threadIdx.x;
    for (int i = tid; i < a.Cin * CO * 9; i += 256) {                            // w[o][ci][t] -> s_w[ci][o * 9 + t]
        const int o = i / (a.Cin * 9), rem = i - o * (a.Cin * 9), ci = rem / 9, t = rem - ci * 9;
        s_w[ci * H3_WS + o * 9 + t] = a.w[i];
    }
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x, b = blockIdx.y;
    const int y0 = ty * HT_H, x0 = tx * HT_W;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const size_t HW = (size_t)H * W;
    const float* xb = a.x + (size_t)b * Cin * HW;
    constexpr int NSLOT = HF_CC * HT_ROWS * HT_SEGS;                           // float4 slots of a chunk: 2592
    constexpr int NPRE = (NSLOT + 255) / 256;                                  // 11
    f32x4 pre[NPRE];
    auto issue = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            const int c = s / (HT_ROWS * HT_SEGS), rem = s - c * (HT_ROWS * HT_SEGS), r = rem / HT_SEGS, sg = rem - r * HT_SEGS;
            const int gy = y0 + r - 1, gx = x0 + 4 * sg - 4;
            const bool ok = s < NSLOT && c0 + c < Cin && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            pre[k] = ok ? *reinterpret_cast<const f32x4*>(xb + (size_t)(c0 + c) * HW + (size_t)gy * W + gx) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int s = tid + k * 256;
            if (s < NSLOT) *reinterpret_cast<f32x4*>(s_in + s * 4) = pre[k];   // slot s = (c, r, sg) in plane order: s * 4 floats
        }
    };
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    float acc[CO][4];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;
    issue(0);
    for (int c0 = 0; c0 < Cin; c0 += HF_CC) {
        if (c0) __syncthreads();                                              // the previous chunk's window reads are done
        commit();
        __syncthreads();
        if (c0 + HF_CC < Cin) issue(c0 + HF_CC);                              // flies under the arithmetic below
        const int nc = min(HF_CC, Cin - c0);
        for (int c = 0; c < nc; ++c) {
            float wv[H3_WS];                                                  // the channel's 27 weights: seven uniform (broadcast) 16-byte LDS reads
#pragma unroll
            for (int q = 0; q < H3_WS / 4; ++q) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(s_w + (c0 + c) * H3_WS + 4 * q);
                wv[4 * q] = t4[0]; wv[4 * q + 1] = t4[1]; wv[4 * q + 2] = t4[2]; wv[4 * q + 3] = t4[3];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float v[6];
                win6(s_in + c * HT_PLANE + (r + ky) * HT_RS + c4, v);
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const float w0 = wv[o * 9 + ky * 3 + 0], w1 = wv[o * 9 + ky * 3 + 1], w2 = wv[o * 9 + ky * 3 + 2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(w2, v[e + 2], fmaf(w1, v[e + 1], fmaf(w0, v[e], acc[o][e])));
                }
            }
        }
    }
    const int gy = y0 + r, gx = x0 + c4;
    if (gy < H && gx < W) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float bv = a.bias ? a.bias[o] : 0.f;
            f32x4 res;
#pragma unroll
            for (int e = 0; e < 4; ++e) res[e] = tanhf(acc[o][e] + bv) * 0.5f + 0.5f;
            *reinterpret_cast<f32x4*>(a.out + ((size_t)b * CO + o) * HW + (size_t)gy * W + gx) = res;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ data gradient
// x = the gradient wrt the head's image [B, CO, H, W] (raw, with img given: the tanh-gradient is applied while staging; or already multiplied,
// img == NULL); w [CO][Cin][3][3]; out = d loss / d input [B, Cin, H, W]:  dx[ci][p] = sum_{o, ky, kx} w[o][ci][ky][kx] gt[o][p - (ky - 1, kx - 1)]
template <int CO>
__global__ __launch_bounds__(256) void head3x3_dgrad_kernel(const H3Args a) {
    __shared__ __attribute__((aligned(16))) float s_g[CO * HT_PLANE];           // 15.5 KB at CO = 3
    __shared__ __attribute__((aligned(16))) float s_w[H3_MAXC * H3_WS];
    const int tid = threadIdx.x;
    for (int i = tid; i < a.Cin * CO * 9; i += 256) {                            // w[o][ci][t] -> s_w[ci][o * 9 + t]
        const int o = i / (a.Cin * 9), rem = i - o * (a.Cin * 9), ci = rem / 9, t = rem - ci * 9;
        s_w[ci * H3_WS + o * 9 + t] = a.w[i];
    }
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x, b = blockIdx.y;
    const int y0 = ty * HT_H, x0 = tx * HT_W;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const size_t HW = (size_t)H * W;
    constexpr int NSLOT = CO * HT_ROWS * HT_SEGS;                              // 972 at CO = 3
    constexpr int NPRE = (NSLOT + 255) / 256;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int s = tid + k * 256;
        if (s < NSLOT) {
            const int c = s / (HT_ROWS * HT_SEGS), rem = s - c * (HT_ROWS * HT_SEGS), r = rem / HT_SEGS, sg = rem - r * HT_SEGS;
            const int gy = y0 + r - 1, gx = x0 + 4 * sg - 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const size_t o = ((size_t)b * CO + c) * HW + (size_t)gy * W + gx;
                v = *reinterpret_cast<const f32x4*>(a.x + o);
                if (a.img) {
                    const f32x4 im = *reinterpret_cast<const f32x4*>(a.img + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xform1<BNERV_IN_TANHGRAD>(v[e], 0.f, 0.f, im[e]);
                }
            }
            *reinterpret_cast<f32x4*>(s_g + s * 4) = v;
        }
    }
    __syncthreads();
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    // window of the transposed conv: tap (ky, kx) reads gt at (y - ky + 1, x - kx + 1): LDS row r + 2 - ky, window element e + 2 - kx
    float g[CO][3][6];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) win6(s_g + o * HT_PLANE + (r + rr) * HT_RS + c4, g[o][rr]);
    const int gy = y0 + r, gx = x0 + c4;
    const bool live = gy < H && gx < W;
    float* ob = a.out + (size_t)b * Cin * HW + (size_t)gy * W + gx;
    for (int ci = 0; ci < Cin; ++ci) {
        float wv[H3_WS];
#pragma unroll
        for (int q = 0; q < H3_WS / 4; ++q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(s_w + ci * H3_WS + 4 * q);
            wv[4 * q] = t4[0]; wv[4 * q + 1] = t4[1]; wv[4 * q + 2] = t4[2]; wv[4 * q + 3] = t4[3];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < CO; ++o) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wk = wv[o * 9 + ky * 3 + kx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk, g[o][2 - ky][e + 2 - kx], acc[e]);
                }
        }
        if (live) *reinterpret_cast<f32x4*>(ob + (size_t)ci * HW) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
}

static bool h3_common(const bnerv_conv_desc& d) {
    static const bool off = [] { const char* e = getenv("BNERV_HEAD3"); return e && e[0] == '0'; }();       // A/B switch
    if (off || d.k != 3 || d.out_s != 1 || d.in_s > 1 || (d.W & 3) || d.B > 65535) return false;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return al(d.x) && al(d.out) && al(d.aux0) && (size_t)d.H * d.W >= 4096 && d.Cin <= H3_MAXC && d.Cout <= H3_MAXC;
}

}  // namespace

// 1: not this kernel's layer; BNERV_OK: launched
int bnerv_head3_try(hipStream_t st, const bnerv_conv_desc& d) {
    if (!h3_common(d)) return 1;
    H3Args a{};
    a.x = d.x; a.w = d.w; a.out = d.out; a.B = d.B; a.H = d.H; a.W = d.W;
    a.tiles_x = cdiv(d.W, HT_W);
    const dim3 grid(a.tiles_x * cdiv(d.H, HT_H), d.B);
    if (!d.transposed && d.in_mode == BNERV_IN_PLAIN && d.ep_mode == BNERV_EP_BIAS_TANH && d.Cout == 3 && d.Cin >= 16 && d.wCo == 3 && d.wCi == d.Cin) {
        a.Cin = d.Cin; a.Cout = 3; a.bias = d.bias;
        hipLaunchKernelGGL(head3x3_fwd_kernel<3>, grid, dim3(256), 0, st, a);
        BNERV_LAUNCH_CHECK("head3x3_fwd");
        return BNERV_OK;
    }
    if (d.transposed && d.ep_mode == BNERV_EP_PLAIN && d.partial == nullptr && d.Cin == 3 && d.Cout >= 16 && d.wCo == 3 && d.wCi == d.Cout &&
        (d.in_mode == BNERV_IN_PLAIN || (d.in_mode == BNERV_IN_TANHGRAD && d.aux0))) {
        a.Cin = d.Cout; a.Cout = 3; a.img = d.in_mode == BNERV_IN_TANHGRAD ? d.aux0 : nullptr;
        hipLaunchKernelGGL(head3x3_dgrad_kernel<3>, grid, dim3(256), 0, st, a);
        BNERV_LAUNCH_CHECK("head3x3_dgrad");
        return BNERV_OK;
    }
    return 1;
}
