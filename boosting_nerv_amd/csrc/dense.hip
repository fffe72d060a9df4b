// dense.hip -- positional encoding, grouped dense (1x1-on-[B,C,1,1]) layers, slab reduction.
//
// The dense layers of the path are GEMVs at the reference's batch sizes (B = 1 per GPU): stem 160->256->fc_h*fc_w*fc_dim,
// stem_t 160->64->32, and the 4 tiny 32->32->C layers of every SFTLayer.  They are HBM/L2-bound weight reads, not MFMA
// work; what matters is launching ALL of one depth in one kernel (grouped descriptors) instead of ~70 tiny launches.
#include "common.h"
#include "sidejob.h"

namespace {

// ---------------------------------------------------------------- positional encoding (model_blocks.py:120-126)
__global__ void pe_f32_kernel(const float* __restrict__ pos, const float* __restrict__ bases, float* __restrict__ out, int N, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int n = i / L, l = i - n * L;
    // ONE IEEE fp32 multiply (no contraction possible: there is no add), then accurate sinf / cosf (full range reduction)
    const float v = __fmul_rn(pos[n], bases[l]);
    out[(size_t)n * 2 * L + l] = sinf(v);
    out[(size_t)n * 2 * L + L + l] = cosf(v);
}
// the same with the fp64 -> fp32 rounding of the position done here (`input[:, None].float()` of model_nerv.py:47 was a launch of its own)
__global__ void pe_f32_from_f64_kernel(const double* __restrict__ pos, const float* __restrict__ bases, float* __restrict__ out, int N, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int n = i / L, l = i - n * L;
    const float v = __fmul_rn((float)pos[n], bases[l]);
    out[(size_t)n * 2 * L + l] = sinf(v);
    out[(size_t)n * 2 * L + L + l] = cosf(v);
}
__global__ void pe_f64_kernel(const double* __restrict__ pos, const float* __restrict__ bases, float* __restrict__ out, int N, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int n = i / L, l = i - n * L;
    const double v = pos[n] * (double)bases[l];
    out[(size_t)n * 2 * L + l] = (float)sin(v);
    out[(size_t)n * 2 * L + L + l] = (float)cos(v);
}

// ---------------------------------------------------------------- grouped dense forward: one wave per output row
struct FwdArgs { bnerv_dense_fwd_desc g[BNERV_MAX_DENSE_GROUPS]; int B; };

__global__ __launch_bounds__(256) void dense_fwd_kernel(const FwdArgs a) {
    const FwdArgs* ap = &a;
    const bnerv_dense_fwd_desc g = a.g[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = blockIdx.x * 4 + wave;
    if (o >= g.O) return;
    const float* wrow = g.w + (size_t)o * g.I;
    const float bias = g.b ? g.b[o] : 0.f;
    for (int b = 0; b < ap->B; ++b) {
        const float* x = g.x + (size_t)b * g.I;
        float s = 0.f;
        for (int i = lane; i < g.I; i += 64) s = fmaf(wrow[i], x[i], s);
        s = wave_sum(s);
        if (lane == 0) {
            const float pre = s + bias;
            float y = pre;
            if (g.act == BNERV_ACT_RELU) y = fmaxf(pre, 0.f);
            else if (g.act == BNERV_ACT_SIN) {
                float sv, cv;
                sincosf(pre, &sv, &cv);
                y = sv;
                if (g.aux) g.aux[(size_t)b * g.O + o] = cv;
            }
            g.y[(size_t)b * g.O + o] = y;
        }
    }
}

// ---------------------------------------------------------------- grouped dense backward
struct BwdArgs { bnerv_dense_bwd_desc g[BNERV_MAX_DENSE_GROUPS]; int B; };

// phase 1: one wave per (output row o, column chunk): dpre[b][o], dw[o][chunk], db[o].  Rows with many inputs (E-NeRV: I ~ 9000) are
// split over blockIdx.z in chunks of DW_CHUNK columns -- a single wave walking the whole row is a serial chain of ~140 dependent
// load/store rounds (70 us per launch at C4); 4 independent columns per lane and round keep the loads in flight.
constexpr int DW_CHUNK = 1024;
__device__ __forceinline__ void dense_bwd_w_body(const BwdArgs& a, const int bx, const int by, const int bz) {
    const BwdArgs* ap = &a;
    const bnerv_dense_bwd_desc g = a.g[by];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = bx * 4 + wave;
    const int i0 = bz * DW_CHUNK;
    if (o >= g.O || (i0 >= g.I && bz != 0)) return;
    const int B = ap->B;
    if (bz == 0) {
        float dbsum = 0.f;
        for (int b = 0; b < B; ++b) {
            const size_t k = (size_t)b * g.O + o;
            float dp = g.dy[k];
            if (g.act == BNERV_ACT_RELU) dp = g.y[k] > 0.f ? dp : 0.f;
            else if (g.act == BNERV_ACT_SIN) dp *= g.aux[k];
            if (lane == 0) g.dpre[k] = dp;
            dbsum += dp;
        }
        if (lane == 0 && g.db) g.db[o] = dbsum;
    }
    const int i1 = min(i0 + DW_CHUNK, g.I);
    for (int ib = i0 + lane; ib < i1; ib += 256) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const size_t k = (size_t)b * g.O + o;
            float dp = g.dy[k];
            if (g.act == BNERV_ACT_RELU) dp = g.y[k] > 0.f ? dp : 0.f;
            else if (g.act == BNERV_ACT_SIN) dp *= g.aux[k];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = ib + 64 * u;
                if (i < i1) s[u] = fmaf(dp, g.x[(size_t)b * g.I + i], s[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = ib + 64 * u;
            if (i < i1) g.dw[(size_t)o * g.I + i] = s[u];
        }
    }
}

// phase 2: dx_part[chunk][b][i] = sum_{o in chunk} dpre[b][o] * w[o][i]      (threads over i: coalesced rows of w).  dpre is rebuilt
// from dy and the saved activation here (one value per row), so the two phases do not depend on each other and share ONE launch.
__device__ __forceinline__ void dense_bwd_x_body(const BwdArgs& a, const int bx, const int by, const int bz) {
    const BwdArgs* ap = &a;
    const bnerv_dense_bwd_desc g = a.g[bz];
    if (!g.dx_part) return;
    const int chunk = bx;
    const int o0 = chunk * BNERV_DENSE_DX_CHUNK;
    if (o0 >= g.O) return;
    const int o1 = min(o0 + BNERV_DENSE_DX_CHUNK, g.O);
    const int b = by;
    __shared__ float s_dp[BNERV_DENSE_DX_CHUNK];
    if ((int)threadIdx.x < BNERV_DENSE_DX_CHUNK) {
        float dp = 0.f;
        if ((int)threadIdx.x < o1 - o0) {
            const size_t k = (size_t)b * g.O + o0 + threadIdx.x;
            dp = g.dy[k];
            if (g.act == BNERV_ACT_RELU) dp = g.y[k] > 0.f ? dp : 0.f;
            else if (g.act == BNERV_ACT_SIN) dp *= g.aux[k];
        }
        s_dp[threadIdx.x] = dp;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.I; i += 256) {
        float wv[BNERV_DENSE_DX_CHUNK];                    // all rows of the chunk in flight, then the (fixed-order) dot product
#pragma unroll
        for (int r = 0; r < BNERV_DENSE_DX_CHUNK; ++r) wv[r] = o0 + r < o1 ? g.w[(size_t)(o0 + r) * g.I + i] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < BNERV_DENSE_DX_CHUNK; ++r) s = fmaf(s_dp[r], wv[r], s);
        g.dx_part[((size_t)chunk * ap->B + b) * g.I + i] = s;
    }
}

// Block -> (group, row block, column chunk) through per-group prefix sums: every group gets exactly the blocks ITS shape needs.  (Until round 5
// the grid was (largest O / 4) x groups x (largest I / chunk): fine for the equal-sized groups of a forward launch's backward, but a launch that
// mixes the stem's 4320-row layer with 32 modulation layers of 12..32 rows started 37 884 blocks of which 1 400 had work: 34 us.)
struct BwdGrid { int nw, wpre[BNERV_MAX_DENSE_GROUPS + 1], xpre[BNERV_MAX_DENSE_GROUPS + 1]; int n_groups, B; };
__global__ __launch_bounds__(256) void dense_bwd_kernel(const BwdArgs a, const BwdGrid q) {
    int t = blockIdx.x;
    const int* pre = t < q.nw ? q.wpre : q.xpre;          // block-uniform
    const bool wphase = t < q.nw;
    if (!wphase) t -= q.nw;
    // group of block t: the number of prefix entries pre[1 .. n_groups - 1] that are <= t -- one compare per lane and a ballot (a scalar walk
    // over up to 40 entries is 40 dependent scalar loads on the critical path of every block of the last group)
    const int ln = threadIdx.x & 63;
    const bool ge = ln + 1 < q.n_groups && t >= pre[ln + 1];
    const int gi = __popcll(__ballot(ge));
    t -= pre[gi];
    if (wphase) {
        const int wx = (a.g[gi].O + 3) >> 2;               // row blocks of this group; t = bz * wx + bx
        dense_bwd_w_body(a, t % wx, gi, t / wx);
    } else {
        const int xx = (a.g[gi].O + BNERV_DENSE_DX_CHUNK - 1) / BNERV_DENSE_DX_CHUNK;      // t = b * xx + chunk
        dense_bwd_x_body(a, t % xx, t / xx, gi);
    }
}

// ---------------------------------------------------------------- stand-alone TAT affine (model_blocks.py:101-105)
__global__ void sft_affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                      float* __restrict__ y, size_t n, int HW) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t bc = i / HW;
        y[i] = x[i] * (scale[bc] + 1.0f) + shift[bc];
    }
}
__global__ __launch_bounds__(256) void sft_affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ g,
                                                             float* __restrict__ dx, float* __restrict__ part, int BC, int HW) {
    const int bc = blockIdx.y, chunk = blockIdx.x;
    const float sc = scale[bc] + 1.0f;
    const size_t base = (size_t)bc * HW;
    float ps = 0.f, pt = 0.f;
    for (int i = chunk * 256 + threadIdx.x; i < HW; i += BNERV_SFT_CHUNKS * 256) {
        const float gv = g[base + i];
        dx[base + i] = gv * sc;
        ps = fmaf(gv, x[base + i], ps);
        pt += gv;
    }
    __shared__ float red[2][4];
    ps = wave_sum(ps); pt = wave_sum(pt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ps; red[1][threadIdx.x >> 6] = pt; }
    __syncthreads();
    if (threadIdx.x < 2) part[((size_t)chunk * 2 + threadIdx.x) * BC + bc] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// out[i] = sum_k slabs[k*count + i]: EPB consecutive elements x (256/EPB) slab lanes per block, fixed combination order
// (1024 threads: these reductions are latency-bound -- a few MB at most -- so the lever is loads in flight, not bytes)
template <int EPB>
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int count, float* __restrict__ out) {
    constexpr int LANES = 1024 / EPB;
    __shared__ float red[LANES][EPB + 1];
    const int e = threadIdx.x % EPB, lane = threadIdx.x / EPB;
    const int i = blockIdx.x * EPB + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < count) {
        int k = lane;
        for (; k + LANES < n_slabs; k += 2 * LANES) { s0 += slabs[(size_t)k * count + i]; s1 += slabs[(size_t)(k + LANES) * count + i]; }
        if (k < n_slabs) s0 += slabs[(size_t)k * count + i];
    }
    red[lane][e] = s0 + s1;
    __syncthreads();
    // fixed-order tree over the lanes (deterministic): 16 partial sums per element, then one thread adds those
    constexpr int G16 = LANES / 16;
    if (lane < 16 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < G16; ++k) t += red[lane * G16 + k][e];
        red[lane * G16][e] = t;
    }
    __syncthreads();
    if (lane == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k * G16][e];
        out[i] = t;
    }
}

}  // namespace

extern "C" int bnerv_pe_fwd_f32(void* stream, const float* pos, const float* bases, float* out, int N, int L) {
    BNERV_REQUIRE(pos && bases && out && N > 0 && L > 0, "pe_fwd_f32: bad args");
    hipLaunchKernelGGL(pe_f32_kernel, dim3(cdiv(N * L, 256)), dim3(256), 0, (hipStream_t)stream, pos, bases, out, N, L);
    BNERV_LAUNCH_CHECK("pe_f32");
    return BNERV_OK;
}

extern "C" int bnerv_pe_fwd_f32_from_f64(void* stream, const double* pos, const float* bases, float* out, int N, int L) {
    BNERV_REQUIRE(pos && bases && out && N > 0 && L > 0, "pe_fwd_f32_from_f64: bad args");
    hipLaunchKernelGGL(pe_f32_from_f64_kernel, dim3(cdiv(N * L, 256)), dim3(256), 0, (hipStream_t)stream, pos, bases, out, N, L);
    BNERV_LAUNCH_CHECK("pe_f32_from_f64");
    return BNERV_OK;
}

extern "C" int bnerv_pe_fwd_f64(void* stream, const double* pos, const float* bases, float* out, int N, int L) {
    BNERV_REQUIRE(pos && bases && out && N > 0 && L > 0, "pe_fwd_f64: bad args");
    hipLaunchKernelGGL(pe_f64_kernel, dim3(cdiv(N * L, 256)), dim3(256), 0, (hipStream_t)stream, pos, bases, out, N, L);
    BNERV_LAUNCH_CHECK("pe_f64");
    return BNERV_OK;
}

// ---------------------------------------------------------------- the time-embedding branch as ONE launch (round 5)
// NeRV_Boost.forward starts with a chain of five dependent tiny launches (model_nerv.py:47-51, model_blocks.py:66-71, :92-105):
//     PE(t) -> [stem layer 0 | stem_t layer 0] -> [stem layer 1 | stem_t layer 1 = z_t] -> every TAT MLP layer 0 -> every TAT MLP layer 1,
// ~4.8 us each (24 us of a 1.47 ms step) although only the stem's second layer (4.4 MB of weights) moves real data: the rest is GEMVs of a few
// thousand MACs whose launches wait on one another.  This kernel runs everything except the stem's second layer:
//   * stem blocks: PE, then a 16-row slice of stem layer 0 (sin; value and cosine written for the backward);
//   * chain blocks (8): ALL the weights the block will need -- stem_t's two layers and its share of the TAT MLPs (both layers) -- are
//     requested from memory at once, straight into LDS (one exposed memory latency instead of four: round 4's one-launch form reloaded at every
//     level and was slower than the launches it replaced); PE meanwhile; then stem_t layer 0 -> layer 1 -> the block's MLPs
//     from LDS.  Chain block 0 also writes PE and stem_t's outputs.
// The stem's second layer follows as the ordinary grouped launch: 5 launches -> 2.  Every tensor the five-launch form left behind is written,
// so the backward (bnerv_dense_grouped_bwd x 4) is unchanged.  A row here is 16 lanes x strided fma chains + 4 shuffle steps, in the grouped
// kernel a wave-strided sum: same quantities, last-bit differences.
struct TBMlp { const float* w1; const float* b1; const float* w2; const float* b2; float* hs; float* out; int C; int _pad; };
struct TBArgs {
    bnerv_time_branch_desc d;
    TBMlp m[BNERV_MAX_DENSE_GROUPS];
    int nb_stem, nb_chain, rows_per_block, mlps_per_block, cmax;      // cmax: the largest C_i, rounded up to 4
};
constexpr int TB_MAXB = 4, TB_MAXI = 256, TB_MAXH = 512, TB_MAXT = 64, TB_MAXO = 32, TB_MAXC = 128, TB_MAXM = 4;      // TB_MAXM: modulation MLPs per chain block
constexpr int TB_NT = 1024;

// pre[b * stride + r] = bias[r] + sum_i W[r * I + i] x[b][i]: 16 lanes per row, lane j takes elements j, j + 16, ... (W anywhere: LDS or global)
template <int XS>
__device__ __forceinline__ void tb_rows16(const float* __restrict__ W, const float* __restrict__ bias, const int I, const int R, const float (*x)[XS],
                                          float* pre, const int stride, const int B) {
    const int tid = threadIdx.x, g16 = tid >> 4, j = tid & 15;
    for (int r = g16; r < ((R + TB_NT / 16 - 1) / (TB_NT / 16)) * (TB_NT / 16); r += TB_NT / 16) {      // (whole passes: the shuffles need every lane)
        float v[TB_MAXB];
#pragma unroll
        for (int b = 0; b < TB_MAXB; ++b) v[b] = 0.f;
        if (r < R) {
            for (int i = j; i < I; i += 16) {
                const float wv = W[(size_t)r * I + i];
#pragma unroll
                for (int b = 0; b < TB_MAXB; ++b)
                    if (b < B) v[b] = fmaf(wv, x[b][i], v[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < TB_MAXB; ++b) {
            if (b < B) {
                float t = v[b];
                t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
                if (r < R && j == 0) pre[b * stride + r] = t + (bias ? bias[r] : 0.f);
            }
        }
    }
}

// rows of SEVERAL small matrices at once, 8 lanes per row: row = mi * rows_pad + r of matrix mi (r >= rows[mi]: idle), inputs x[mi or 0][b][.]
// in LDS, weights and biases in LDS.  Returns the pre-activation of (mi, r, b) in lane j == 0 of the row's group through `emit`.
template <class RowsOf, class WOf, class BOf, class XOf, class Emit>
__device__ __forceinline__ void tb_multi8(const int n_mat, const int rows_pad, const int I, const int B, RowsOf rows_of, WOf w_of, BOf b_of, XOf x_of, Emit emit) {
    const int tid = threadIdx.x, g8 = tid >> 3, j = tid & 7;
    const int total = n_mat * rows_pad, step = TB_NT / 8;
    for (int row0 = 0; row0 < total; row0 += step) {        // (whole passes: the shuffles need every lane)
        const int row = row0 + g8;
        const int mi = row / rows_pad, r = row - mi * rows_pad;
        const bool live = row < total && r < rows_of(mi);
        float v[TB_MAXB];
#pragma unroll
        for (int b = 0; b < TB_MAXB; ++b) v[b] = 0.f;
        if (live) {
            const float* W = w_of(mi) + r * I;
#pragma unroll 4
            for (int i = j; i < I; i += 8) {
                const float wv = W[i];
#pragma unroll
                for (int b = 0; b < TB_MAXB; ++b)
                    if (b < B) v[b] = fmaf(wv, x_of(mi, b)[i], v[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < TB_MAXB; ++b) {
            if (b < B) {
                float t = v[b];
                t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
                if (live && j == 0) emit(mi, r, b, t + b_of(mi)[r]);
            }
        }
    }
}

__global__ __launch_bounds__(TB_NT) void time_branch_kernel(const TBArgs a) {
    const bnerv_time_branch_desc& d = a.d;
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    __shared__ __attribute__((aligned(16))) float s_pe[TB_MAXB][TB_MAXI];
    __shared__ __attribute__((aligned(16))) float s_t0[TB_MAXB][TB_MAXT];
    __shared__ __attribute__((aligned(16))) float s_zt[TB_MAXB][TB_MAXT];
    __shared__ __attribute__((aligned(16))) float s_h[TB_MAXM][TB_MAXB][TB_MAXO];
    __shared__ __attribute__((aligned(16))) float s_o[TB_MAXB][TB_MAXH];
    __shared__ float s_bias[2 * TB_MAXT + TB_MAXM * (TB_MAXO + TB_MAXC)];
    const int tid = threadIdx.x;
    const int B = d.B, L = d.L, I = 2 * d.L, TO = d.TO;
    const bool stem_role = (int)blockIdx.x < a.nb_stem;
    const int cb = (int)blockIdx.x - a.nb_stem;              // chain block index
    // ---- chain blocks: request EVERY weight and bias of the block's chain NOW (the PE below overlaps the flight): one exposed latency
    float* w_t0 = tb_smem;                                   // [TH][I]
    float* w_t1 = w_t0 + d.TH * I;                           // [TO][TH]
    float* w_m = w_t1 + TO * d.TH;                           // per MLP of this block: [TO][TO] then [cmax][TO]
    const int mstride = TO * TO + a.cmax * TO;
    float* b_t0 = s_bias; float* b_t1 = s_bias + TB_MAXT; float* b_m = s_bias + 2 * TB_MAXT;     // per MLP: [TO] then [cmax]
    const int bstride = TB_MAXO + TB_MAXC;
    const int m0 = cb * a.mlps_per_block, nm = stem_role ? 0 : max(0, min(a.mlps_per_block, d.n_mlp - m0));
    const int sr0 = (int)blockIdx.x * a.rows_per_block, snr = stem_role ? min(a.rows_per_block, d.SH - sr0) : 0;
    if (stem_role) {
        if (tid < snr) s_bias[tid] = d.sb0 ? d.sb0[sr0 + tid] : 0.f;        // (visible behind the barrier that follows the PE)
    } else {
        auto stage = [&](const float* __restrict__ src, float* dst, const int n) {          // n % 4 == 0, 16-byte aligned (launcher)
            for (int i = tid * 4; i < n; i += TB_NT * 4) *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
        };
        auto stage1 = [&](const float* __restrict__ src, float* dst, const int n) {
            for (int i = tid; i < n; i += TB_NT) dst[i] = src ? src[i] : 0.f;
        };
        stage(d.tw0, w_t0, d.TH * I);
        stage(d.tw1, w_t1, TO * d.TH);
        stage1(d.tb0, b_t0, d.TH);
        stage1(d.tb1, b_t1, TO);
        for (int k = 0; k < nm; ++k) {
            const TBMlp& mm = a.m[m0 + k];
            stage(mm.w1, w_m + k * mstride, TO * TO);
            stage(mm.w2, w_m + k * mstride + TO * TO, mm.C * TO);
            stage1(mm.b1, b_m + k * bstride, TO);
            stage1(mm.b2, b_m + k * bstride + TB_MAXO, mm.C);
        }
    }
    // ---- positional encoding (model_blocks.py:120-126): ONE IEEE fp32 multiply, accurate sinf / cosf; fp64 positions rounded to fp32 first
    for (int e = tid; e < B * L; e += TB_NT) {
        const int n = e / L, l = e - n * L;
        const float v = __fmul_rn((float)d.pos[n], d.bases[l]);
        const float sv = sinf(v), cv = cosf(v);
        s_pe[n][l] = sv; s_pe[n][L + l] = cv;
        if (cb == 0) { d.pe[(size_t)n * I + l] = sv; d.pe[(size_t)n * I + L + l] = cv; }
    }
    __syncthreads();
    if (stem_role) {
        // this block's rows of stem layer 0 (weights straight from memory: each row is read by exactly one block)
        tb_rows16<TB_MAXI>(d.sw0 + (size_t)sr0 * I, nullptr, I, snr, s_pe, &s_o[0][0], TB_MAXH, B);
        __syncthreads();
        for (int e = tid; e < B * snr; e += TB_NT) {
            const int b = e / snr, r = e - b * snr;
            float sv, cv;
            sincosf(s_o[b][r] + s_bias[r], &sv, &cv);
            d.sy0[(size_t)b * d.SH + sr0 + r] = sv;
            d.saux0[(size_t)b * d.SH + sr0 + r] = cv;
        }
        return;
    }
    // ---- chain: stem_t layer 0, layer 1 (z_t), then this block's modulation MLPs -- both layers of all of them at once -- from LDS
    tb_rows16<TB_MAXI>(w_t0, b_t0, I, d.TH, s_pe, &s_t0[0][0], TB_MAXT, B);
    __syncthreads();
    for (int e = tid; e < B * d.TH; e += TB_NT) {
        const int b = e / d.TH, r = e - b * d.TH;
        float sv, cv;
        sincosf(s_t0[b][r], &sv, &cv);
        s_t0[b][r] = sv;
        if (cb == 0) { d.ty0[(size_t)b * d.TH + r] = sv; d.taux0[(size_t)b * d.TH + r] = cv; }
    }
    __syncthreads();
    tb_rows16<TB_MAXT>(w_t1, b_t1, d.TH, TO, s_t0, &s_zt[0][0], TB_MAXT, B);
    __syncthreads();
    for (int e = tid; e < B * TO; e += TB_NT) {
        const int b = e / TO, r = e - b * TO;
        float sv, cv;
        sincosf(s_zt[b][r], &sv, &cv);
        s_zt[b][r] = sv;
        if (cb == 0) { d.ty1[(size_t)b * TO + r] = sv; d.taux1[(size_t)b * TO + r] = cv; }
    }
    __syncthreads();
    tb_multi8(nm, TO, TO, B, [&](int) { return TO; }, [&](int k) { return w_m + k * mstride; }, [&](int k) { return b_m + k * bstride; },
              [&](int, int b) { return &s_zt[b][0]; },
              [&](int k, int r, int b, float v) { const float y = fmaxf(v, 0.f); s_h[k][b][r] = y; a.m[m0 + k].hs[(size_t)b * TO + r] = y; });
    __syncthreads();
    tb_multi8(nm, a.cmax, TO, B, [&](int k) { return a.m[m0 + k].C; }, [&](int k) { return w_m + k * mstride + TO * TO; }, [&](int k) { return b_m + k * bstride + TB_MAXO; },
              [&](int k, int b) { return &s_h[k][b][0]; },
              [&](int k, int r, int b, float v) { a.m[m0 + k].out[(size_t)b * a.m[m0 + k].C + r] = v; });
}

// Descriptor tables travel BY VALUE in the kernel-argument segment (<= 4 KB: BNERV_MAX_DENSE_GROUPS * 88 B), so a captured
// hipGraph replays them without touching host memory.
extern "C" int bnerv_dense_grouped_fwd(void* stream, const bnerv_dense_fwd_desc* groups, int n_groups, int B) {
    BNERV_REQUIRE(groups && n_groups > 0 && n_groups <= BNERV_MAX_DENSE_GROUPS && B > 0, "dense_grouped_fwd: bad args (n_groups=%d)", n_groups);
    FwdArgs a;
    a.B = B;
    int maxO = 0;
    for (int i = 0; i < n_groups; ++i) {
        BNERV_REQUIRE(groups[i].x && groups[i].w && groups[i].y && groups[i].I > 0 && groups[i].O > 0, "dense_grouped_fwd: bad group %d", i);
        a.g[i] = groups[i];
        if (groups[i].O > maxO) maxO = groups[i].O;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(cdiv(maxO, 4), n_groups), dim3(256), 0, st, a);
    BNERV_LAUNCH_CHECK("dense_fwd");
    return BNERV_OK;
}

extern "C" int bnerv_dense_grouped_bwd(void* stream, const bnerv_dense_bwd_desc* groups, int n_groups, int B) {
    BNERV_REQUIRE(groups && n_groups > 0 && n_groups <= BNERV_MAX_DENSE_GROUPS && B > 0 && B <= 65535, "dense_grouped_bwd: bad args (n_groups=%d)", n_groups);
    BwdArgs a;
    a.B = B;
    int maxO = 0, maxI = 0;
    bool any_dx = false;
    for (int i = 0; i < n_groups; ++i) {
        const bnerv_dense_bwd_desc& g = groups[i];
        BNERV_REQUIRE(g.x && g.w && g.dy && g.dpre && g.dw && g.I > 0 && g.O > 0, "dense_grouped_bwd: bad group %d", i);
        if (g.act == BNERV_ACT_RELU) BNERV_REQUIRE(g.y, "dense_grouped_bwd: relu group %d needs y", i);
        if (g.act == BNERV_ACT_SIN) BNERV_REQUIRE(g.aux, "dense_grouped_bwd: sin group %d needs aux", i);
        a.g[i] = g;
        if (g.O > maxO) maxO = g.O;
        if (g.I > maxI) maxI = g.I;
        any_dx |= g.dx_part != nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    BwdGrid q;
    q.n_groups = n_groups; q.B = B;
    int nw = 0, nx = 0;
    for (int i = 0; i < n_groups; ++i) {
        q.wpre[i] = nw; q.xpre[i] = nx;
        nw += cdiv(groups[i].O, 4) * cdiv(groups[i].I, DW_CHUNK);
        if (groups[i].dx_part) nx += cdiv(groups[i].O, BNERV_DENSE_DX_CHUNK) * B;
    }
    q.wpre[n_groups] = nw; q.xpre[n_groups] = nx; q.nw = nw;
    (void)maxO; (void)maxI; (void)any_dx;
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(nw + nx), dim3(256), 0, st, a, q);
    BNERV_LAUNCH_CHECK("dense_bwd");
    return BNERV_OK;
}

extern "C" int bnerv_sft_affine_fwd(void* stream, const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW) {
    BNERV_REQUIRE(x && scale && shift && y && B > 0 && C > 0 && HW > 0, "sft_affine_fwd: bad args");
    const size_t n = (size_t)B * C * HW;
    int gx = (int)((n + 1023) / 1024); if (gx > 8192) gx = 8192; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(sft_affine_fwd_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, n, HW);
    BNERV_LAUNCH_CHECK("sft_affine_fwd");
    return BNERV_OK;
}

extern "C" int bnerv_sft_affine_bwd(void* stream, const float* x, const float* scale, const float* g, float* dx, float* part, int B, int C, int HW) {
    BNERV_REQUIRE(x && scale && g && dx && part && B > 0 && C > 0 && HW > 0 && B * C <= 65535, "sft_affine_bwd: bad args");
    hipLaunchKernelGGL(sft_affine_bwd_kernel, dim3(BNERV_SFT_CHUNKS, B * C), dim3(256), 0, (hipStream_t)stream, x, scale, g, dx, part, B * C, HW);
    BNERV_LAUNCH_CHECK("sft_affine_bwd");
    return BNERV_OK;
}

extern "C" int bnerv_reduce_slabs(void* stream, const float* slabs, int n_slabs, int count, float* out) {
    BNERV_REQUIRE(slabs && out && n_slabs > 0 && count > 0, "reduce_slabs: bad args");
    if (count >= 2048 || n_slabs <= 16)
        hipLaunchKernelGGL(reduce_slabs_kernel<32>, dim3(cdiv(count, 32)), dim3(1024), 0, (hipStream_t)stream, slabs, n_slabs, count, out);
    else
        hipLaunchKernelGGL(reduce_slabs_kernel<4>, dim3(cdiv(count, 4)), dim3(1024), 0, (hipStream_t)stream, slabs, n_slabs, count, out);
    BNERV_LAUNCH_CHECK("reduce_slabs");
    return BNERV_OK;
}

// 1: not this launch's branch (the caller issues the grouped launches); BNERV_OK; negative BNERV_E_*.
extern "C" int bnerv_time_branch_fwd(void* stream, const bnerv_time_branch_desc* dp, const bnerv_time_branch_mlp* mlps) {
    BNERV_REQUIRE(dp && (mlps || dp->n_mlp == 0), "time_branch_fwd: null descriptor");
    const bnerv_time_branch_desc& d = *dp;
    BNERV_REQUIRE(d.pos && d.bases && d.pe && d.sw0 && d.sy0 && d.saux0 && d.tw0 && d.tw1 && d.ty0 && d.taux0 && d.ty1 && d.taux1, "time_branch_fwd: null tensor");
    BNERV_REQUIRE(d.B > 0 && d.L > 0 && d.SH > 0 && d.TH > 0 && d.TO > 0 && d.n_mlp >= 0, "time_branch_fwd: bad shape");
    if (d.B > TB_MAXB || 2 * d.L > TB_MAXI || d.SH > 4096 || d.TH > TB_MAXT || d.TO > TB_MAXO || d.n_mlp > BNERV_MAX_DENSE_GROUPS) return 1;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if ((2 * d.L) % 4 || d.TH % 4 || d.TO % 4 || !al16(d.tw0) || !al16(d.tw1)) return 1;                 // 16-byte staging of the chain's weights
    TBArgs a;
    a.d = d;
    a.cmax = 4;
    for (int i = 0; i < d.n_mlp; ++i) {
        BNERV_REQUIRE(mlps[i].w1 && mlps[i].w2 && mlps[i].hs && mlps[i].out && mlps[i].C > 0, "time_branch_fwd: bad modulation MLP %d", i);
        if (((mlps[i].C + 3) & ~3) > a.cmax) a.cmax = (mlps[i].C + 3) & ~3;
        if (mlps[i].C > TB_MAXC || !al16(mlps[i].w1) || !al16(mlps[i].w2)) return 1;
        a.m[i].w1 = mlps[i].w1; a.m[i].b1 = mlps[i].b1; a.m[i].w2 = mlps[i].w2; a.m[i].b2 = mlps[i].b2; a.m[i].hs = mlps[i].hs; a.m[i].out = mlps[i].out;
        a.m[i].C = mlps[i].C; a.m[i]._pad = 0;
    }
    a.rows_per_block = 16;
    a.nb_stem = cdiv(d.SH, a.rows_per_block);
    a.mlps_per_block = d.n_mlp > 32 ? TB_MAXM : 2;           // 16 chain blocks for C1's 32 MLPs; at most 10 x 4
    a.nb_chain = d.n_mlp > 0 ? cdiv(d.n_mlp, a.mlps_per_block) : 1;
    const size_t lds = ((size_t)d.TH * 2 * d.L + (size_t)d.TO * d.TH + (size_t)a.mlps_per_block * ((size_t)d.TO * d.TO + (size_t)a.cmax * d.TO)) * sizeof(float);
    if (lds > 100 * 1024) return 1;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&time_branch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL(time_branch_kernel, dim3(a.nb_stem + a.nb_chain), dim3(TB_NT), lds, (hipStream_t)stream, a);
    BNERV_LAUNCH_CHECK("time_branch");
    return BNERV_OK;
}
