// conv.hip -- fp32 implicit-GEMM convolution on MFMA 16x16x4 (gfx950) with fused prologues / epilogues.
//
// One kernel family serves the forward convolution and the data gradient of every conv on the Boosting-NeRV decoder
// path (reference call sites: lib/quant_ops.py:39-41 via model_blocks.py:74-89, :196-220, model_enerv.py:73-102,
// model_nerv.py:41,56).  GEMM view, per sample:
//     D[pixel][cout] = sum_{tap, cin} A[pixel][(tap,cin)] * Wt[(tap,cin)][cout]
// M = pixels (16 consecutive x per MFMA tile), N = cout (16 per tile), K = taps*Cin walked 4 at a time.
//
// Block = 256 threads = 4 waves, one 8x32 spatial tile x NTB*16 output channels.
//   LDS:  s_in  [16 ch][ROWS x RS halo tile], channel-planar, plane stride == 16 (mod 32) so the four k-lanes of one
//               ds_read_b32 hit disjoint bank halves (conflict-free A-fragment reads)
//         s_w   [tap][q][n][64]  B fragments in lane order (one conflict-free ds_read_b32 per fragment)
//         s_out [cout][8x32 (+4)] accumulator tile, re-laid so the copy-out walks the FINAL layout row by row
//               (coalesced stores and coalesced aux reads for every epilogue, pixel-shuffle included).
// Each wave owns 2 rows x 32 px = 4 M-tiles; acc[4][NTB] (f32x4 each) lives in registers for the whole K loop.
// HBM traffic is the algorithmic minimum + halo: input read once per cout-group, output written once.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;     // spatial tile
constexpr int CC = 16;             // input channels per K chunk
constexpr int NQ = CC / 4;

template <int KS> struct Geo {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int ROWS = TH + 2 * PAD;
    static constexpr int RS = TW + 2 * PAD;
    static constexpr int PLANE_RAW = ROWS * RS;
    static constexpr int PLANE = ((PLANE_RAW - 16 + 31) / 32) * 32 + 16;   // >= PLANE_RAW and == 16 (mod 32)
    static constexpr int T = KS * KS;
};
constexpr int CS = TH * TW + 4;    // s_out channel stride (floats): 16-B aligned, spreads ds_write_b128 over all banks

struct KArgs {
    bnerv_conv_desc d;
    int tiles_x, tiles_y;
};

template <int IN>
__device__ __forceinline__ float load_in(const bnerv_conv_desc& d, int b, int ci, int gy, int gx) {
    if constexpr (IN == BNERV_IN_UNSHUFFLE) {
        const int s = d.in_s, s2 = s * s;
        const int c = ci / s2, rem = ci - c * s2, i = rem / s, j = rem - i * s;
        const size_t idx = (((size_t)b * (d.Cin / s2) + c) * (size_t)(d.H * s) + (size_t)(gy * s + i)) * (size_t)(d.W * s) + (size_t)(gx * s + j);
        return d.x[idx];
    } else {
        const size_t idx = (((size_t)b * d.Cin + ci) * d.H + gy) * (size_t)d.W + gx;
        const float v = d.x[idx];
        if constexpr (IN == BNERV_IN_PLAIN) return v;
        if constexpr (IN == BNERV_IN_AFFINE) return v * (1.0f + d.scale[b * d.Cin + ci]) + d.shift[b * d.Cin + ci];
        if constexpr (IN == BNERV_IN_GELU_AFFINE) return gelu_f(v) * (1.0f + d.scale[b * d.Cin + ci]) + d.shift[b * d.Cin + ci];
        if constexpr (IN == BNERV_IN_TANHGRAD) {
            const float t = 2.0f * d.aux0[idx] - 1.0f;
            return v * 0.5f * (1.0f - t * t);
        }
        return v;
    }
}

template <int KS, int IN, int EP, int NTB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const KArgs ka) {
    using G = Geo<KS>;
    const bnerv_conv_desc& d = ka.d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + CC * G::PLANE;
    float* s_out = smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ty0 = (tile / ka.tiles_x) * TH, tx0 = (tile % ka.tiles_x) * TW;
    const int ng = blockIdx.y, b = blockIdx.z;
    const int H = d.H, W = d.W, Cin = d.Cin, Cout = d.Cout;
    const int co_base = ng * NTB * 16;

    f32x4 acc[4][NTB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) abase[m] = kq * G::PLANE + (2 * wave + (m >> 1)) * G::RS + (m & 1) * 16 + li;

    for (int c0 = 0; c0 < Cin; c0 += CC) {
        const int cc = min(CC, Cin - c0);
        const int nq = (cc + 3) >> 2;
        __syncthreads();
        // ---- stage the input halo tile (prologue applied; zero outside the image = padding AFTER the prologue) ----
        for (int idx = tid; idx < nq * 4 * G::PLANE_RAW; idx += 256) {
            const int c = idx / G::PLANE_RAW;
            const int rem = idx - c * G::PLANE_RAW;
            const int r = rem / G::RS, col = rem - r * G::RS;
            const int gy = ty0 + r - G::PAD, gx = tx0 + col - G::PAD, ci = c0 + c;
            float v = 0.f;
            if (ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = load_in<IN>(d, b, ci, gy, gx);
            s_in[c * G::PLANE + r * G::RS + col] = v;
        }
        // ---- stage this chunk's weights as B fragments: lane (j = l&15, kq = l>>4) <- W(co_base+16n+j, c0+4q+kq, tap) ----
        for (int idx = tid; idx < G::T * nq * NTB * 64; idx += 256) {
            const int l = idx & 63;
            int rest = idx >> 6;
            const int n = rest % NTB; rest /= NTB;
            const int q = rest % nq;
            const int tap = rest / nq;
            const int co = co_base + n * 16 + (l & 15), ci = c0 + q * 4 + (l >> 4);
            float v = 0.f;
            if (co < Cout && ci < Cin) {
                v = d.transposed ? d.w[((size_t)ci * d.wCi + co) * G::T + (G::T - 1 - tap)]
                                 : d.w[((size_t)co * d.wCi + ci) * G::T + tap];
            }
            s_w[((tap * NQ + q) * NTB + n) * 64 + l] = v;
        }
        __syncthreads();
        // ---- MFMA main loop ----
#pragma unroll
        for (int tap = 0; tap < G::T; ++tap) {
            const int toff = (tap / KS) * G::RS + (tap % KS);
            for (int q = 0; q < nq; ++q) {
                float bf[NTB], af[4];
#pragma unroll
                for (int n = 0; n < NTB; ++n) bf[n] = s_w[((tap * NQ + q) * NTB + n) * 64 + lane];
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = s_in[abase[m] + q * 4 * G::PLANE + toff];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- accumulators -> s_out[cout_local][py*32 + px]   (D layout: lane holds pixels 4*kq..4*kq+3 of cout li) ----
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            const int py = 2 * wave + (m >> 1), px = (m & 1) * 16 + 4 * kq;
            *reinterpret_cast<f32x4*>(&s_out[(n * 16 + li) * CS + py * TW + px]) = acc[m][n];
        }
    __syncthreads();

    // ---- copy-out in the final layout ----
    const int s = d.out_s, s2 = s * s;
    if constexpr (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN) {
        // stride-1 convs only: one pixel per thread per channel, plus per-channel (ds, dt) partial sums for this tile
        float* s_red = smem + NTB * 16 * CS;      // [4 waves][NTB*16][2]
        const int py = tid >> 5, px = tid & 31;
        const int gy = ty0 + py, gx = tx0 + px;
        const bool inside = gy < H && gx < W;
        for (int cl = 0; cl < NTB * 16; ++cl) {
            const int co = co_base + cl;
            if (co >= Cout) break;
            float ps = 0.f, pt = 0.f;
            if (inside) {
                const size_t o = (((size_t)b * Cout + co) * H + gy) * (size_t)W + gx;
                const float v = s_out[cl * CS + py * TW + px];
                const float sc = 1.0f + d.scale[b * Cout + co];
                if constexpr (EP == BNERV_EP_DGELU) {
                    const float pre = d.aux0[o];
                    d.out[o] = v * sc * gelu_grad_f(pre);
                    ps = v * gelu_f(pre);
                } else {
                    const float y0 = d.aux0[o];
                    d.out[o] = (d.aux1[o] + v * sc) * (d.aux2 ? d.aux2[o] : 1.0f);
                    ps = v * y0;
                }
                pt = v;
            }
            ps = wave_sum(ps);
            pt = wave_sum(pt);
            if (lane == 0) {
                s_red[(wave * NTB * 16 + cl) * 2 + 0] = ps;
                s_red[(wave * NTB * 16 + cl) * 2 + 1] = pt;
            }
        }
        __syncthreads();
        const int ncl = min(NTB * 16, Cout - co_base);
        for (int idx = tid; idx < ncl * 2; idx += 256) {
            const int cl = idx >> 1, which = idx & 1;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += s_red[(w * NTB * 16 + cl) * 2 + which];
            const size_t row = (size_t)tile * gridDim.z + b;      // [tiles][B][2][Cout]: one reduce_slabs over tiles
            d.partial[(row * 2 + which) * Cout + co_base + cl] = v;
        }
    } else if (s <= 2 && (NTB * 16) % s2 == 0) {
        // fast path: walk final rows; consecutive threads -> consecutive output columns
        const int sh = s - 1;                     // s in {1,2}
        const int OW = TW << sh, OH = TH << sh;
        const int ncf = (NTB * 16) >> (2 * sh);
        const int Cf = Cout >> (2 * sh);
        const int HF = H << sh, WF = W << sh;
        for (int cfl = 0; cfl < ncf; ++cfl) {
            const int cf = (co_base >> (2 * sh)) + cfl;
            if (cf >= Cf) break;
            for (int e = tid; e < OH * OW; e += 256) {
                const int orow = e / OW, ocol = e - orow * OW;
                const int py = orow >> sh, i = orow & sh, px = ocol >> sh, j = ocol & sh;
                const int gy = (ty0 << sh) + orow, gx = (tx0 << sh) + ocol;
                if (gy >= HF || gx >= WF) continue;
                const int cl = (cfl << (2 * sh)) + i * s + j;
                float v = s_out[cl * CS + py * TW + px];
                const size_t o = (((size_t)b * Cf + cf) * HF + gy) * (size_t)WF + gx;
                if constexpr (EP != BNERV_EP_PLAIN) { if (d.bias) v += d.bias[co_base + cl]; }
                if constexpr (EP == BNERV_EP_BIAS || EP == BNERV_EP_PLAIN) d.out[o] = v;
                if constexpr (EP == BNERV_EP_BIAS_SIN) { float sv, cv; sincosf(v, &sv, &cv); d.out[o] = sv; if (d.out2) d.out2[o] = cv; }
                if constexpr (EP == BNERV_EP_BIAS_RES) d.out[o] = v + d.aux0[o];
                if constexpr (EP == BNERV_EP_BIAS_TANH) d.out[o] = tanhf(v) * 0.5f + 0.5f;
            }
        }
    } else {
        // generic pixel-shuffle scatter (s = 3, 5: low-resolution stages only)
        const int Cf = Cout / s2, HF = H * s, WF = W * s;
        for (int e = tid; e < NTB * 16 * TH * TW; e += 256) {
            const int cl = e >> 8, p = e & 255, py = p >> 5, px = p & 31;
            const int co = co_base + cl;
            if (co >= Cout) break;
            if (ty0 + py >= H || tx0 + px >= W) continue;
            const int c = co / s2, rem = co - c * s2, i = rem / s, j = rem - i * s;
            const size_t o = (((size_t)b * Cf + c) * HF + (size_t)((ty0 + py) * s + i)) * (size_t)WF + (size_t)((tx0 + px) * s + j);
            float v = s_out[cl * CS + py * TW + px];
            if constexpr (EP != BNERV_EP_PLAIN) { if (d.bias) v += d.bias[co]; }
            if constexpr (EP == BNERV_EP_BIAS || EP == BNERV_EP_PLAIN) d.out[o] = v;
            if constexpr (EP == BNERV_EP_BIAS_SIN) { float sv, cv; sincosf(v, &sv, &cv); d.out[o] = sv; if (d.out2) d.out2[o] = cv; }
            if constexpr (EP == BNERV_EP_BIAS_RES) d.out[o] = v + d.aux0[o];
            if constexpr (EP == BNERV_EP_BIAS_TANH) d.out[o] = tanhf(v) * 0.5f + 0.5f;
        }
    }
}

template <int KS, int NTB>
constexpr size_t conv_lds_bytes(bool reduce_ep) {
    using G = Geo<KS>;
    size_t a = (size_t)CC * G::PLANE + (size_t)G::T * NQ * NTB * 64;
    size_t o = (size_t)NTB * 16 * CS + (reduce_ep ? (size_t)4 * NTB * 16 * 2 : 0);
    return (a > o ? a : o) * sizeof(float);
}

template <int KS, int IN, int EP, int NTB>
int launch_one(hipStream_t st, const KArgs& ka, int ngroups) {
    constexpr bool red = (EP == BNERV_EP_DGELU || EP == BNERV_EP_DSIN);
    constexpr size_t lds = conv_lds_bytes<KS, NTB>(red);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<KS, IN, EP, NTB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(ka.tiles_x * ka.tiles_y, ngroups, ka.d.B);
    hipLaunchKernelGGL((conv_igemm_kernel<KS, IN, EP, NTB>), grid, dim3(256), lds, st, ka);
    BNERV_LAUNCH_CHECK("conv_igemm");
    return BNERV_OK;
}

template <int KS, int IN, int EP>
int launch_ntb(hipStream_t st, const KArgs& ka) {
    const int nt = cdiv(ka.d.Cout, 16);
    if (nt == 1) return launch_one<KS, IN, EP, 1>(st, ka, 1);
    if (nt == 2) return launch_one<KS, IN, EP, 2>(st, ka, 1);
    if (nt == 3) return launch_one<KS, IN, EP, 3>(st, ka, 1);
    return launch_one<KS, IN, EP, 4>(st, ka, cdiv(nt, 4));
}

template <int KS>
int launch_mode(hipStream_t st, const KArgs& ka) {
    const int in = ka.d.in_mode, ep = ka.d.ep_mode;
#define BNERV_CASE(I, E) if (in == I && ep == E) return launch_ntb<KS, I, E>(st, ka);
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_SIN)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_BIAS_TANH)
    BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_UNSHUFFLE, BNERV_EP_PLAIN)
    BNERV_CASE(BNERV_IN_TANHGRAD, BNERV_EP_PLAIN)
    if constexpr (KS == 3) {
        BNERV_CASE(BNERV_IN_AFFINE, BNERV_EP_BIAS)
        BNERV_CASE(BNERV_IN_GELU_AFFINE, BNERV_EP_BIAS_RES)
        BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DGELU)
        BNERV_CASE(BNERV_IN_PLAIN, BNERV_EP_DSIN)
    }
#undef BNERV_CASE
    return bnerv_set_error(BNERV_E_ARG, "conv_igemm: unsupported (k=%d, in_mode=%d, ep_mode=%d)", KS, in, ep);
}

}  // namespace

extern "C" int bnerv_conv_tiles(int H, int W) { return cdiv(H, TH) * cdiv(W, TW); }

extern "C" int bnerv_conv_igemm(void* stream, const bnerv_conv_desc* dp) {
    BNERV_REQUIRE(dp != nullptr, "conv_igemm: null descriptor");
    KArgs ka;
    ka.d = *dp;
    const bnerv_conv_desc& d = ka.d;
    BNERV_REQUIRE(d.k == 1 || d.k == 3, "conv_igemm: k must be 1 or 3 (got %d)", d.k);
    BNERV_REQUIRE(d.B > 0 && d.Cin > 0 && d.Cout > 0 && d.H > 0 && d.W > 0, "conv_igemm: bad dims");
    BNERV_REQUIRE(d.B <= 65535, "conv_igemm: batch too large");
    BNERV_REQUIRE(d.x && d.w && d.out, "conv_igemm: null tensor");
    BNERV_REQUIRE(d.in_s >= 1 && d.out_s >= 1, "conv_igemm: shuffle factors must be >= 1");
    BNERV_REQUIRE(d.Cout % (d.out_s * d.out_s) == 0, "conv_igemm: Cout %d not divisible by out_s^2", d.Cout);
    if (d.in_mode == BNERV_IN_UNSHUFFLE) BNERV_REQUIRE(d.Cin % (d.in_s * d.in_s) == 0, "conv_igemm: Cin %d not divisible by in_s^2", d.Cin);
    if (d.transposed) BNERV_REQUIRE(d.Cout == d.wCi && d.Cin == d.wCo, "conv_igemm: transposed dims mismatch");
    else BNERV_REQUIRE(d.Cout == d.wCo && d.Cin == d.wCi, "conv_igemm: weight dims mismatch");
    if (d.in_mode == BNERV_IN_AFFINE || d.in_mode == BNERV_IN_GELU_AFFINE) BNERV_REQUIRE(d.scale && d.shift, "conv_igemm: affine prologue needs scale/shift");
    if (d.in_mode == BNERV_IN_TANHGRAD) BNERV_REQUIRE(d.aux0, "conv_igemm: tanh-grad prologue needs aux0");
    if (d.ep_mode == BNERV_EP_BIAS_RES) BNERV_REQUIRE(d.aux0, "conv_igemm: residual epilogue needs aux0");
    if (d.ep_mode == BNERV_EP_DGELU) BNERV_REQUIRE(d.aux0 && d.scale && d.partial && d.out_s == 1, "conv_igemm: DGELU epilogue args");
    if (d.ep_mode == BNERV_EP_DSIN) BNERV_REQUIRE(d.aux0 && d.aux1 && d.scale && d.partial && d.out_s == 1, "conv_igemm: DSIN epilogue args");
    ka.tiles_x = cdiv(d.W, TW);
    ka.tiles_y = cdiv(d.H, TH);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return d.k == 1 ? launch_mode<1>(st, ka) : launch_mode<3>(st, ka);
}
