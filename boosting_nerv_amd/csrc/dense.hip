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

struct BwdGrid { int wx, wy, wz, xx, xy, xz; };
__global__ __launch_bounds__(256) void dense_bwd_kernel(const BwdArgs a, const BwdGrid q) {
    const int nw = q.wx * q.wy * q.wz;
    int t = blockIdx.x;
    if (t < nw) {                                          // block-uniform
        const int bx = t % q.wx; t /= q.wx;
        dense_bwd_w_body(a, bx, t % q.wy, t / q.wy);
    } else {
        t -= nw;
        const int bx = t % q.xx; t /= q.xx;
        dense_bwd_x_body(a, bx, t % q.xy, t / q.xy);
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
    BwdGrid q{cdiv(maxO, 4), n_groups, cdiv(maxI, DW_CHUNK), 0, 0, 0};
    if (any_dx) { q.xx = cdiv(maxO, BNERV_DENSE_DX_CHUNK); q.xy = B; q.xz = n_groups; }
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(q.wx * q.wy * q.wz + q.xx * q.xy * q.xz), dim3(256), 0, st, a, q);
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
