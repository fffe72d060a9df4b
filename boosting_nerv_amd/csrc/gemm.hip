// gemm.hip -- the two places of the path that really are GEMMs (north_star: "MFMA used only for the 1x1 / stem dense projections
// where they really are GEMMs"), on v_mfma_f32_16x16x4_f32:
//
//  (1) bnerv_dense_gemm_*: dense layers applied to 16 or more rows -- NeRV_MLP / the token MLPs of E-NeRV's transformer stem
//      (reference model_blocks.py:66-71, model_enerv.py:22-60: nn.Linear / 1x1 conv on [tokens, C]).  One strided kernel serves
//        forward   y  = act(x W^T + b)                       (epilogue: bias, relu | sin with cos saved)
//        dx        dx = dpre W                                (A-side prologue: dpre = dy * act'(.) rebuilt on load)
//        dW, db    dW = dpre^T x,  db = column sums           (same prologue; the bias gradient is a virtual column of ones)
//  (2) bnerv_cnx_mlp_*: the pointwise MLP of the ConvNeXt encoder block of HNeRV_Boost (reference model_blocks.py:245-258:
//      pwconv1 -> GELU -> pwconv2 -> gamma -> + residual), NCHW, as ONE forward and ONE backward kernel.  Per 16-pixel N tile the
//      hidden activations never leave registers: the D fragments of GEMM 1 (4 consecutive hidden units of one pixel per lane)
//      ARE the B fragments of GEMM 2 once its K index is walked in the order (hidden tile, register) -- no shuffle, no LDS.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------- dense GEMM
// C[m][n] = epi( sum_k A(m,k) * B(k,n) ),  A(m,k) = pro(a[m*sam + k*sak]),  B(k,n) = b[k*sbk + n*sbn]  (n == N-1 may be a virtual
// column of ones).  Block = 4 waves = 32 x 32 outputs (wave w: M tile w & 1, N tile w >> 1), K walked 16 at a time through LDS.
enum { PRO_NONE = 0, PRO_RELU = 1, PRO_COS = 2 };           // A-side: a * (aux > 0) | a * aux
enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_BIAS_SIN = 3, EPI_SPLIT_LAST = 4 };   // SPLIT_LAST: column N-1 goes to c2[m]

struct GemmArgs {
    const float* a; const float* a_aux; const float* b; const float* bias;
    float* c; float* c2;
    int M, N, K;
    long sam, sak, sbk, sbn;
    int ldc;              // row stride of c (EPI_SPLIT_LAST: N - 1 real columns)
    int pro, epi, ones_col;
    int kchunk;           // > 0: split K -- block z handles k in [z * kchunk, (z + 1) * kchunk) and writes slab z of c ([z][M][ldc], EPI_NONE)
};

__global__ __launch_bounds__(256) void dense_gemm_kernel(const GemmArgs g) {
    __shared__ float s_a[32][17];                          // [m][k]
    __shared__ float s_b[16][33];                          // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int mt = wave & 1, nt = wave >> 1;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    const int kbeg = g.kchunk > 0 ? (int)blockIdx.z * g.kchunk : 0;
    const int kend = g.kchunk > 0 ? min(g.K, kbeg + g.kchunk) : g.K;
    float* cbase = g.c + (g.kchunk > 0 ? (size_t)blockIdx.z * g.M * g.ldc : 0);
    // K loop, 16 at a time: the loads of step k0 + 16 are issued into registers before the MFMAs of step k0 and written to LDS after
    // them (one barrier pair per step, global latency under the previous step instead of in front of every one).
    // Staging slot e = tid, tid + 256: A 32 x 16 and B 16 x 32 (two elements of each per thread), zero outside the matrix; consecutive
    // threads walk the unit-stride index of A: k for row-major activations, m for the transposed operand of the weight gradients.
    float ra[2], rb[2];
    auto load_step = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * 256;
            const int m = g.sam == 1 ? (e & 31) : (e >> 4), k = g.sam == 1 ? (e >> 5) : (e & 15);
            float v = 0.f;
            if (m0 + m < g.M && k0 + k < kend) {
                const long idx = (long)(m0 + m) * g.sam + (long)(k0 + k) * g.sak;
                v = g.a[idx];
                if (g.pro == PRO_RELU) v = g.a_aux[idx] > 0.f ? v : 0.f;
                else if (g.pro == PRO_COS) v *= g.a_aux[idx];
            }
            ra[j] = v;
            const int kb = e >> 5, n = e & 31;
            float u = 0.f;
            if (k0 + kb < kend && n0 + n < g.N)
                u = (g.ones_col && n0 + n == g.N - 1) ? 1.0f : g.b[(long)(k0 + kb) * g.sbk + (long)(n0 + n) * g.sbn];
            rb[j] = u;
        }
    };
    auto store_step = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + j * 256;
            const int m = g.sam == 1 ? (e & 31) : (e >> 4), k = g.sam == 1 ? (e >> 5) : (e & 15);
            s_a[m][k] = ra[j];
            s_b[e >> 5][e & 31] = rb[j];
        }
    };
    if (kbeg < kend) load_step(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        store_step();
        __syncthreads();
        if (k0 + 16 < kend) load_step(k0 + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_a[mt * 16 + li][4 * s + kq], s_b[4 * s + kq][nt * 16 + li], acc, 0, 0, 0);
        __syncthreads();
    }
    // D: lane (n = li, rows 4 kq .. 4 kq + 3)
    const int n = n0 + nt * 16 + li;
    if (n >= g.N) return;
    const float bias = (g.epi >= EPI_BIAS && g.epi <= EPI_BIAS_SIN && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + 4 * kq + r;
        if (m >= g.M) continue;
        float v = acc[r] + bias;
        if (g.epi == EPI_SPLIT_LAST) {
            if (n == g.N - 1) { if (g.c2) g.c2[m] = v; }
            else cbase[(long)m * g.ldc + n] = v;
            continue;
        }
        if (g.epi == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        else if (g.epi == EPI_BIAS_SIN) {
            float sv, cv;
            sincosf(v, &sv, &cv);
            v = sv;
            if (g.c2) g.c2[(long)m * g.ldc + n] = cv;
        }
        cbase[(long)m * g.ldc + n] = v;
    }
}

// out[m][n] = sum_z slab[z][m][n] in a fixed order; column N-1 goes to c2 (bias gradient).  Block = 32 outputs x 8 slab lanes
// (lane l adds slabs l, l + 8, ...; the eight partial sums are added in lane order), so a few hundred slabs are read by
// M N / 32 blocks instead of M N / 256 threads walking all of them
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const float* slab, int nz, int M, int N, float* c, float* c2) {
    __shared__ float s_p[8][32];
    const int o = threadIdx.x & 31, zl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    float s = 0.f;
    if (i < M * N)
        for (int z = zl; z < nz; z += 8) s += slab[(size_t)z * M * N + i];
    s_p[zl][o] = s;
    __syncthreads();
    if (zl == 0 && i < M * N) {
        const float t = ((s_p[0][o] + s_p[1][o]) + (s_p[2][o] + s_p[3][o])) + ((s_p[4][o] + s_p[5][o]) + (s_p[6][o] + s_p[7][o]));
        const int m = i / N, n = i - m * N;
        if (n == N - 1) { if (c2) c2[m] = t; }
        else c[(size_t)m * (N - 1) + n] = t;
    }
}

int launch_gemm(hipStream_t st, const GemmArgs& g, int nz = 1) {
    hipLaunchKernelGGL(dense_gemm_kernel, dim3(cdiv(g.N, 32), cdiv(g.M, 32), nz), dim3(256), 0, st, g);
    BNERV_LAUNCH_CHECK("dense_gemm");
    return BNERV_OK;
}

constexpr int GEMM_SPLITK_MIN_ROWS = 512;                // weight gradients over more rows than this are split along the rows (an [O, I] output
                                                         // is a handful of 32 x 32 tiles: without the split 2304 rows ran on 18 blocks for 243 us)
// (256 rows per split: the K loop has no prefetch, so its global latency is hidden by blocks per CU -- 1024 rows per split left ~2)
int dw_splits(int B) { return B < GEMM_SPLITK_MIN_ROWS ? 1 : (cdiv(B, 256) > 512 ? 512 : cdiv(B, 256)); }

// ---------------------------------------------------------------------------------------------------------------- ConvNeXt MLP
// x, inp, out, dout, dx: [B, C, HW];  w1 [4C, C], b1 [4C], w2 [C, 4C], b2 [C], gamma [C] (may be NULL: no layer scale).
// C in {16, 32, 48, 64}.  Block = 256 threads; every wave owns NT consecutive 16-pixel tiles of one sample; the weights sit in
// LDS as MFMA A fragments ([tile][k step][lane]) for the whole block.
// pixel tiles per wave: 2 (A-fragment reuse) on images that fill the chip; 1 on small ones -- a wave's item is one serial chain of
// ~512 MFMAs + 64 erf per tile (~28 us: the round-5 C3 timeline shows 57 us for the launches at 72 x 128, 36 x 64 and 18 x 32 alike),
// so a small image wants twice the waves on half the chain

struct MlpArgs {
    const float* x; const float* inp; const float* w1; const float* b1; const float* w2; const float* b2; const float* gamma;
    float* out; float* hsave;                              // forward (hsave: [B, 4C, HW] pre-activations kept for backward, or NULL)
    const float* dout; const float* hin; float* dx; float* gbuf; float* dhbuf;   // backward
    int B, C, HW;
};

// A fragments of W1 for hidden tile t, k step s (input channels 4s .. 4s+3): lane (m = li, kq) <- w1[16 t + li][4 s + kq]
// A fragments of W2 for output tile o, k index (t, r): lane (m = li, kq) <- w2[16 o + li][16 t + 4 kq + r]
template <int C>
__device__ __forceinline__ void stage_weights(const MlpArgs& a, float* s_w1, float* s_w2, int tid) {
    constexpr int HID = 4 * C, KS1 = C / 4, HT = HID / 16;
    // Coalesced float4 loads, all of a matrix's in flight at once (16 per thread at C = 64), then the LDS stores.  A float4 of W1's row
    // 16 t + li at columns 4 s .. 4 s + 3 is the fragment element of lanes (li, kq = 0..3) of (t, s); a float4 of W2's row 16 o + li at
    // columns 16 t + 4 kq .. + 3 is elements r = 0..3 of lane (li, kq) of (o, t).  (The "load a dword, store it" loop this replaces was one
    // L2 round trip per element: 128 per block at C = 64.)
    constexpr int N4 = HID * C / 1024;                    // float4 per thread and matrix: 1 / 4 / 9 / 16 at C = 16 / 32 / 48 / 64
    f32x4 v[N4];
#pragma unroll
    for (int u = 0; u < N4; ++u) v[u] = *reinterpret_cast<const f32x4*>(a.w1 + (size_t)(tid + u * 256) * 4);
#pragma unroll
    for (int u = 0; u < N4; ++u) {
        const int e4 = tid + u * 256, row = e4 / KS1, sq = e4 - row * KS1, t = row >> 4, li = row & 15;
        float* d = s_w1 + (t * KS1 + sq) * 64 + li;
        d[0] = v[u][0]; d[16] = v[u][1]; d[32] = v[u][2]; d[48] = v[u][3];
    }
#pragma unroll
    for (int u = 0; u < N4; ++u) v[u] = *reinterpret_cast<const f32x4*>(a.w2 + (size_t)(tid + u * 256) * 4);
#pragma unroll
    for (int u = 0; u < N4; ++u) {
        const int e4 = tid + u * 256, row = e4 / (HID / 4), q4 = e4 - row * (HID / 4), t = q4 >> 2, kq = q4 & 3, o = row >> 4, li = row & 15;
        float* d = s_w2 + ((o * HT + t) * 4) * 64 + kq * 16 + li;
        d[0] = v[u][0]; d[64] = v[u][1]; d[128] = v[u][2]; d[192] = v[u][3];
    }
}

template <int C, int CNX_NT>
__global__ __launch_bounds__(256) void cnx_mlp_fwd_kernel(const MlpArgs a) {
    constexpr int HID = 4 * C, KS1 = C / 4, HT = HID / 16, OT = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w1 = smem;
    float* s_w2 = smem + HID * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    stage_weights<C>(a, s_w1, s_w2, tid);
    __syncthreads();
    const int tiles_per_b = (a.HW + 16 * CNX_NT - 1) / (16 * CNX_NT);
    const int total = a.B * tiles_per_b;
    for (int item = blockIdx.x * 4 + wave; item < total; item += gridDim.x * 4) {
        const int b = item / tiles_per_b, p0 = (item - b * tiles_per_b) * 16 * CNX_NT;
        const float* xb = a.x + (size_t)b * C * a.HW;
        // B fragments of x: lane (n = pixel li, kq) <- x[4 s + kq][pixel]
        float xf[CNX_NT][KS1];
        int px[CNX_NT];
#pragma unroll
        for (int n = 0; n < CNX_NT; ++n) {
            px[n] = p0 + 16 * n + li;
            const bool ok = px[n] < a.HW;
#pragma unroll
            for (int s = 0; s < KS1; ++s) xf[n][s] = ok ? xb[(size_t)(4 * s + kq) * a.HW + px[n]] : 0.f;
        }
        f32x4 acc2[OT][CNX_NT];
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int n = 0; n < CNX_NT; ++n) acc2[o][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < HT; ++t) {
            f32x4 h[CNX_NT];
#pragma unroll
            for (int n = 0; n < CNX_NT; ++n) h[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const float wf = s_w1[(t * KS1 + s) * 64 + lane];
#pragma unroll
                for (int n = 0; n < CNX_NT; ++n) h[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xf[n][s], h[n], 0, 0, 0);
            }
            // lane holds hidden units 16 t + 4 kq + r of pixel li: bias, GELU, then straight into GEMM 2 as B fragments
            // (GELU through common.h's packed pair form -- one exp2 + one rcp per element, |error| <= 1.5e-7 on erf, the TAT conv0 epilogue's
            //  arithmetic -- instead of erff: 64 erff per 16-pixel tile were ~5 us of a wave's ~30 us chain)
            const f32x4 bb4 = *reinterpret_cast<const f32x4*>(a.b1 + 16 * t + 4 * kq);
#pragma unroll
            for (int n = 0; n < CNX_NT; ++n) {
                const f32x4 pre = h[n] + bb4;
                if (a.hsave && px[n] < a.HW) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.hsave[((size_t)b * HID + 16 * t + 4 * kq + r) * a.HW + px[n]] = pre[r];
                }
                f32x4 gd;
                gelu_pair4_f(pre, &h[n], &gd);
            }
#pragma unroll
            for (int o = 0; o < OT; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wf = s_w2[((o * HT + t) * 4 + r) * 64 + lane];
#pragma unroll
                    for (int n = 0; n < CNX_NT; ++n) acc2[o][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, h[n][r], acc2[o][n], 0, 0, 0);
                }
        }
        // epilogue: lane (pixel li, channels 16 o + 4 kq + r): out = inp + gamma * (y + b2)
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * o + 4 * kq + r;
                const float gm = a.gamma ? a.gamma[c] : 1.0f, bb = a.b2[c];
#pragma unroll
                for (int n = 0; n < CNX_NT; ++n)
                    if (px[n] < a.HW) {
                        const size_t idx = ((size_t)b * C + c) * a.HW + px[n];
                        a.out[idx] = a.inp[idx] + gm * (acc2[o][n][r] + bb);
                    }
            }
    }
}

// backward, per pixel tile, from the pre-activations h1 the forward kept:  G = gelu(h1), G' = gelu'(h1);  dY = gamma * dout;
// dG = W2^T dY;  dH = dG * G';  dX = W1^T dH.  Writes G and dH ([B, 4C, HW]: the K = pixel operands of the two weight-gradient
// GEMMs, which go through bnerv_conv_wgrad with k = 1) and dX.  The transposed weights are staged as A fragments:
//   W2^T for hidden tile t, k step s (output channels 4s..4s+3): lane (m = li, kq) <- w2[4 s + kq][16 t + li]
//   W1^T for input tile i, k index (t, r):                         lane (m = li, kq) <- w1[16 t + 4 kq + r][16 i + li]
template <int C, int CNX_NT>
__global__ __launch_bounds__(256) void cnx_mlp_bwd_kernel(const MlpArgs a) {
    constexpr int HID = 4 * C, KS1 = C / 4, HT = HID / 16, OT = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w2t = smem;                                   // [(t * KS1 + s) * 64 + lane]
    float* s_w1t = smem + HID * C;                         // [((i * HT + t) * 4 + r) * 64 + lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    {   // coalesced float4 loads, a matrix at a time, then float4 LDS stores (see stage_weights): a float4 of W2's row 4 s + kq at columns
        // 16 t + li0 .. + 3 is lanes (li0 .. li0 + 3, kq) of (t, s); a float4 of W1's row 16 t + 4 kq + r at columns 16 i + li0 .. + 3 is
        // lanes (li0 .. + 3, kq) of (i, t, r)
        constexpr int N4 = HID * C / 1024;
        f32x4 v[N4];
#pragma unroll
        for (int u = 0; u < N4; ++u) v[u] = *reinterpret_cast<const f32x4*>(a.w2 + (size_t)(tid + u * 256) * 4);
#pragma unroll
        for (int u = 0; u < N4; ++u) {
            const int e4 = tid + u * 256, row = e4 / (HID / 4), q4 = e4 - row * (HID / 4), col0 = 4 * q4;
            const int t = col0 >> 4, li0 = col0 & 15, sq = row >> 2, kq = row & 3;
            *reinterpret_cast<f32x4*>(s_w2t + (t * KS1 + sq) * 64 + kq * 16 + li0) = v[u];
        }
#pragma unroll
        for (int u = 0; u < N4; ++u) v[u] = *reinterpret_cast<const f32x4*>(a.w1 + (size_t)(tid + u * 256) * 4);
#pragma unroll
        for (int u = 0; u < N4; ++u) {
            const int e4 = tid + u * 256, row = e4 / (C / 4), c4 = e4 - row * (C / 4), col0 = 4 * c4;
            const int i = col0 >> 4, li0 = col0 & 15, t = row >> 4, kq = (row & 15) >> 2, r = row & 3;
            *reinterpret_cast<f32x4*>(s_w1t + ((i * HT + t) * 4 + r) * 64 + kq * 16 + li0) = v[u];
        }
    }
    __syncthreads();
    const int tiles_per_b = (a.HW + 16 * CNX_NT - 1) / (16 * CNX_NT);
    const int total = a.B * tiles_per_b;
    for (int item = blockIdx.x * 4 + wave; item < total; item += gridDim.x * 4) {
        const int b = item / tiles_per_b, p0 = (item - b * tiles_per_b) * 16 * CNX_NT;
        const float* gb = a.dout + (size_t)b * C * a.HW;
        float yf[CNX_NT][KS1];                             // B fragments of dY = gamma * dout (k = channel 4 s + kq)
        int px[CNX_NT];
#pragma unroll
        for (int n = 0; n < CNX_NT; ++n) {
            px[n] = p0 + 16 * n + li;
            const bool ok = px[n] < a.HW;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const int c = 4 * s + kq;
                yf[n][s] = ok ? gb[(size_t)c * a.HW + px[n]] * (a.gamma ? a.gamma[c] : 1.0f) : 0.f;
            }
        }
        f32x4 accx[OT][CNX_NT];
#pragma unroll
        for (int i = 0; i < OT; ++i)
#pragma unroll
            for (int n = 0; n < CNX_NT; ++n) accx[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < HT; ++t) {
            f32x4 dg[CNX_NT];
            float pre[CNX_NT][4];
#pragma unroll
            for (int n = 0; n < CNX_NT; ++n) {
                dg[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pre[n][r] = px[n] < a.HW ? a.hin[((size_t)b * HID + 16 * t + 4 * kq + r) * a.HW + px[n]] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const float wt = s_w2t[(t * KS1 + s) * 64 + lane];
#pragma unroll
                for (int n = 0; n < CNX_NT; ++n) dg[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt, yf[n][s], dg[n], 0, 0, 0);
            }
            // gelu and gelu' together (common.h gelu_pair4_f: one shared exp2 + one rcp per element instead of two erff and an expf --
            // 64 such triples per 16-pixel tile were ~13 us of a wave's ~34 us chain)
#pragma unroll
            for (int n = 0; n < CNX_NT; ++n) {
                f32x4 gv, gd;
                gelu_pair4_f(f32x4{pre[n][0], pre[n][1], pre[n][2], pre[n][3]}, &gv, &gd);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dh = dg[n][r] * gd[r];
                    dg[n][r] = dh;
                    if (px[n] < a.HW) {
                        const size_t idx = ((size_t)b * HID + 16 * t + 4 * kq + r) * a.HW + px[n];
                        a.gbuf[idx] = gv[r];
                        a.dhbuf[idx] = dh;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < OT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wf = s_w1t[((i * HT + t) * 4 + r) * 64 + lane];
#pragma unroll
                    for (int n = 0; n < CNX_NT; ++n) accx[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, dg[n][r], accx[i][n], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < OT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * i + 4 * kq + r;
#pragma unroll
                for (int n = 0; n < CNX_NT; ++n)
                    if (px[n] < a.HW) a.dx[((size_t)b * C + c) * a.HW + px[n]] = accx[i][n][r];
            }
    }
}

template <int C, int NT>
int launch_mlp_nt(hipStream_t st, const MlpArgs& a, bool bwd) {
    const size_t lds = (size_t)2 * 4 * C * C * sizeof(float);
    const int total = a.B * cdiv(a.HW, 16 * NT);
    int grid = cdiv(total, 4);
    const int cap = 256 * (lds > 80 * 1024 ? 1 : 2);
    if (grid > cap) grid = cap;
    if (bwd) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnx_mlp_bwd_kernel<C, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((cnx_mlp_bwd_kernel<C, NT>), dim3(grid), dim3(256), lds, st, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnx_mlp_fwd_kernel<C, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((cnx_mlp_fwd_kernel<C, NT>), dim3(grid), dim3(256), lds, st, a);
    }
    BNERV_LAUNCH_CHECK(bwd ? "cnx_mlp_bwd" : "cnx_mlp_fwd");
    return BNERV_OK;
}

template <int C>
int launch_mlp(hipStream_t st, const MlpArgs& a, bool bwd) {
    // one 16-pixel tile per wave while that still leaves every wave at most one item (1024 wave slots at one block per CU)
    static const int nt1_max = [] { const char* e = getenv("BNERV_CNX_NT1_MAX"); return e ? atoi(e) : 16384; }();     // (A/B switch: pixels)
    if ((long)a.B * a.HW <= nt1_max) return launch_mlp_nt<C, 1>(st, a, bwd);
    return launch_mlp_nt<C, 2>(st, a, bwd);
}

int launch_mlp_c(hipStream_t st, const MlpArgs& a, bool bwd) {
    switch (a.C) {
        case 16: return launch_mlp<16>(st, a, bwd);
        case 32: return launch_mlp<32>(st, a, bwd);
        case 48: return launch_mlp<48>(st, a, bwd);
        case 64: return launch_mlp<64>(st, a, bwd);
    }
    return bnerv_set_error(BNERV_E_ARG, "cnx_mlp: C must be 16, 32, 48 or 64 (got %d)", a.C);
}

// dw2 = gamma S, db2 = gamma t, dgamma = rowsum(w2 * S) + b2 * t for one ConvNeXt block ([C x 4C] bookkeeping after the two k = 1
// weight-gradient launches): one block per output row c
__global__ __launch_bounds__(256) void cnx_param_grads_kernel(const float* __restrict__ S, const float* __restrict__ t, const float* __restrict__ w2,
                                                              const float* __restrict__ b2, const float* __restrict__ gamma, float* __restrict__ dw2,
                                                              float* __restrict__ db2, float* __restrict__ dgamma, int C) {
    __shared__ float s_part[4];
    const int c = blockIdx.x, n = 4 * C;
    const float g = gamma[c];
    float acc = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float sv = S[(size_t)c * n + j];
        dw2[(size_t)c * n + j] = g * sv;
        acc = fmaf(w2[(size_t)c * n + j], sv, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        dgamma[c] = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) + b2[c] * t[c];
        db2[c] = g * t[c];
    }
}

}  // namespace

extern "C" int bnerv_dense_gemm_fwd(void* stream, const float* x, const float* w, const float* b, float* y, float* aux, int B, int I, int O, int act) {
    BNERV_REQUIRE(x && w && y && B > 0 && I > 0 && O > 0, "dense_gemm_fwd: bad args");
    BNERV_REQUIRE(act == BNERV_ACT_NONE || act == BNERV_ACT_RELU || act == BNERV_ACT_SIN, "dense_gemm_fwd: bad activation %d", act);
    GemmArgs g{};
    g.a = x; g.b = w; g.bias = b; g.c = y; g.c2 = aux;
    g.M = B; g.N = O; g.K = I;
    g.sam = I; g.sak = 1; g.sbk = 1; g.sbn = I; g.ldc = O;
    g.pro = PRO_NONE;
    g.epi = act == BNERV_ACT_RELU ? EPI_BIAS_RELU : (act == BNERV_ACT_SIN ? EPI_BIAS_SIN : EPI_BIAS);
    return launch_gemm(reinterpret_cast<hipStream_t>(stream), g);
}

extern "C" size_t bnerv_dense_gemm_bwd_ws_bytes(int B, int I, int O) {
    if (B <= 0 || I <= 0 || O <= 0) return 0;
    const int nz = dw_splits(B);
    return nz > 1 ? (size_t)nz * O * (I + 1) * sizeof(float) : 0;
}

// dy [B, O]; y (relu mask) / aux (cos) [B, O] as the activation needs; dx [B, I] (may be NULL); dw [O, I]; db [O] (may be NULL);
// ws: bnerv_dense_gemm_bwd_ws_bytes (row-split partial weight gradients when B is large: the patchify convs have B = pixels)
extern "C" int bnerv_dense_gemm_bwd(void* stream, const float* x, const float* w, const float* y, const float* aux, const float* dy,
                                    float* dx, float* dw, float* db, void* ws, size_t ws_bytes, int B, int I, int O, int act) {
    BNERV_REQUIRE(x && w && dy && dw && B > 0 && I > 0 && O > 0, "dense_gemm_bwd: bad args");
    if (act == BNERV_ACT_RELU) BNERV_REQUIRE(y, "dense_gemm_bwd: relu needs y");
    if (act == BNERV_ACT_SIN) BNERV_REQUIRE(aux, "dense_gemm_bwd: sin needs aux");
    const int pro = act == BNERV_ACT_RELU ? PRO_RELU : (act == BNERV_ACT_SIN ? PRO_COS : PRO_NONE);
    const float* a_aux = act == BNERV_ACT_RELU ? y : aux;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // dW[o][i] (+ db[o] as the virtual last column) = sum_b dpre[b][o] * x[b][i]
    GemmArgs g{};
    g.a = dy; g.a_aux = a_aux; g.b = x; g.c = dw; g.c2 = db;
    g.M = O; g.N = I + 1; g.K = B;
    g.sam = 1; g.sak = O; g.sbk = I; g.sbn = 1; g.ldc = I;
    g.pro = pro; g.epi = EPI_SPLIT_LAST; g.ones_col = 1;
    const int nz = dw_splits(B);
    int rc;
    if (nz > 1) {
        const size_t need = bnerv_dense_gemm_bwd_ws_bytes(B, I, O);
        if (!ws || ws_bytes < need) return bnerv_set_error(BNERV_E_WS, "dense_gemm_bwd: workspace %zu < %zu", ws_bytes, need);
        g.c = reinterpret_cast<float*>(ws); g.c2 = nullptr;
        g.ldc = I + 1; g.epi = EPI_NONE;
        g.kchunk = ((cdiv(B, nz) + 15) / 16) * 16;
        rc = launch_gemm(st, g, cdiv(B, g.kchunk));
        if (rc != BNERV_OK) return rc;
        hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3(cdiv(O * (I + 1), 32)), dim3(256), 0, st, reinterpret_cast<const float*>(ws), cdiv(B, g.kchunk), O, I + 1, dw, db);
        BNERV_LAUNCH_CHECK("gemm_splitk_finish");
    } else {
        rc = launch_gemm(st, g);
    }
    if (rc != BNERV_OK || !dx) return rc;
    // dx[b][i] = sum_o dpre[b][o] * w[o][i]
    GemmArgs h{};
    h.a = dy; h.a_aux = a_aux; h.b = w; h.c = dx;
    h.M = B; h.N = I; h.K = O;
    h.sam = O; h.sak = 1; h.sbk = I; h.sbn = 1; h.ldc = I;
    h.pro = pro; h.epi = EPI_NONE;
    return launch_gemm(st, h);
}

extern "C" int bnerv_cnx_mlp_fwd(void* stream, const float* x, const float* inp, const float* w1, const float* b1, const float* w2, const float* b2,
                                 const float* gamma, float* out, float* hsave, int B, int C, int HW) {
    BNERV_REQUIRE(x && inp && w1 && b1 && w2 && b2 && out && B > 0 && HW > 0, "cnx_mlp_fwd: bad args");
    BNERV_REQUIRE(((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15) == 0, "cnx_mlp_fwd: weights must be 16-byte aligned");
    MlpArgs a{};
    a.x = x; a.inp = inp; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.gamma = gamma; a.out = out; a.hsave = hsave;
    a.B = B; a.C = C; a.HW = HW;
    return launch_mlp_c(reinterpret_cast<hipStream_t>(stream), a, false);
}

// h1: [B, 4C, HW] pre-activations the forward kept (hsave); gbuf, dhbuf: [B, 4C, HW] outputs (gelu(h1) and d loss / d h1): the
// pixel-contraction operands of the weight gradients
extern "C" int bnerv_cnx_mlp_bwd(void* stream, const float* h1, const float* dout, const float* w1, const float* w2, const float* gamma,
                                 float* dx, float* gbuf, float* dhbuf, int B, int C, int HW) {
    BNERV_REQUIRE(h1 && dout && w1 && w2 && dx && gbuf && dhbuf && B > 0 && HW > 0, "cnx_mlp_bwd: bad args");
    BNERV_REQUIRE(((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15) == 0, "cnx_mlp_bwd: weights must be 16-byte aligned");
    MlpArgs a{};
    a.hin = h1; a.dout = dout; a.w1 = w1; a.w2 = w2; a.gamma = gamma; a.dx = dx; a.gbuf = gbuf; a.dhbuf = dhbuf;
    a.B = B; a.C = C; a.HW = HW;
    return launch_mlp_c(reinterpret_cast<hipStream_t>(stream), a, true);
}

// S [C, 4C], t [C]: what bnerv_conv_wgrad (k = 1) returns for (gelu(h1), dout); outputs dw2 [C, 4C], db2 [C], dgamma [C]
extern "C" int bnerv_cnx_param_grads(void* stream, const float* S, const float* t, const float* w2, const float* b2, const float* gamma,
                                     float* dw2, float* db2, float* dgamma, int C) {
    BNERV_REQUIRE(S && t && w2 && b2 && gamma && dw2 && db2 && dgamma && C > 0, "cnx_param_grads: bad args");
    hipLaunchKernelGGL(cnx_param_grads_kernel, dim3(C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, t, w2, b2, gamma, dw2, db2, dgamma, C);
    BNERV_LAUNCH_CHECK("cnx_param_grads");
    return BNERV_OK;
}
