// Micro-benchmark (not part of the product): a chain of DEPENDENT low-resolution conv-like phases (the <= 180x320 stages of the C1 step:
// one 4x16-pixel tile per block, 60..470 blocks, 6..10 us per launch of which ~1 us is arithmetic) as
//   (a) NP kernel nodes of a replayed hipGraph (what the step does today), against
//   (b) ONE launch of NP * G blocks with tile-level DATAFLOW flags: block (p, t) loads its layer's weights first (they do not depend on
//       phase p - 1), then waits for the flags of tiles t - 1, t, t + 1 of phase p - 1 (release store by the producer block, acquire
//       spin by the consumer), reads their outputs, computes, stores, raises its own flag.  No grid barrier, no contended atomic: every
//       flag has one writer.  Relies on in-order block dispatch (producers have lower block ids); a spin time-out turns a stall into
//       an error count instead of a hang.
// Each phase: 18 KB of weights + a haloed input tile of 32 channels -> LDS, 72 dependent MFMA 16x16x4 per wave fed from LDS, 16 x 64
// outputs per block.  Build: hipcc --offload-arch=gfx950 -O3 -o flagdep flagdep.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NCH = 32, TPX = 64;                    // channels, pixels per tile
constexpr int WFL = 16 * NCH * 9;                    // weights of one phase (floats)

struct Args {
    const float* w;       // [NP][WFL]
    float* act;           // [NP + 1][G][NCH][TPX]   (phase p reads act[p], writes act[p + 1]; only 16 of the 32 channels are produced: the rest stay constant)
    unsigned* flags;      // [NP][G]
    const unsigned* epoch;
    unsigned* err;
    int G, NP;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
// MODE 0: acquire loads in the spin (what round 1 of this file did); 1: relaxed spin, ONE acquire fence after it, release fence before the flag;
// 2: no cache-maintenance fences at all -- the activations travel through device-coherent (sc1) buffer loads / stores, the flag is relaxed
template <int MODE>
__device__ __forceinline__ void phase_body(const Args& a, const int p, const int t, const bool wait, const unsigned ep, float* s_in, float* s_w) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = a.G;
    // ---- weights first: independent of the previous phase
    const float* wsrc = a.w + (size_t)p * WFL;
    float wv[WFL / 256];
#pragma unroll
    for (int u = 0; u < WFL / 256; ++u) wv[u] = wsrc[tid + u * 256];
    // ---- wait for the three producer tiles
    if (wait && p > 0) {
        if (tid < 3) {
            const int tt = min(max(t + tid - 1, 0), G - 1);
            const unsigned* f = a.flags + (size_t)(p - 1) * G + tt;
            int spins = 0;
            if constexpr (MODE == 0) {
                while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ep) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { atomicAdd(a.err, 1u); break; }
                }
            } else {
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ep) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) { atomicAdd(a.err, 1u); break; }
                }
            }
        }
        __syncthreads();
        if constexpr (MODE <= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- input: own tile + one column strip of each neighbour (the halo), 32 channels
    const float* in = a.act + (size_t)p * G * NCH * TPX;
    f32x4 ra[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + k * 256;                       // float4 index within [NCH][TPX / 4]
        const int c = idx / (TPX / 4), q = idx % (TPX / 4);
        const int tt = q == 0 ? max(t - 1, 0) : (q == TPX / 4 - 1 ? min(t + 1, G - 1) : t);    // edge quads come from the neighbours
        if constexpr (MODE == 2) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((size_t)a.G * NCH * TPX * 4), 0x00020000);
            ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((size_t)tt * NCH + c) * TPX + 4 * q) * 4), 0, 16));   // aux 16 = sc1: device scope
        } else
            ra[k] = *reinterpret_cast<const f32x4*>(in + ((size_t)tt * NCH + c) * TPX + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < WFL / 256; ++u) s_w[tid + u * 256] = wv[u];
#pragma unroll
    for (int k = 0; k < 2; ++k) *reinterpret_cast<f32x4*>(s_in + (tid + k * 256) * 4) = ra[k];
    __syncthreads();
    // ---- K loop: 72 dependent MFMAs per wave, A / B from LDS
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int q = 0; q < NCH / 4; ++q)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const float av = s_in[(4 * q + kq) * TPX + ((wave * 16 + li + tp) & 63)];
            const float bv = s_w[li * NCH * 9 + (4 * q + kq) * 9 + tp];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
    // ---- epilogue: 16 channels x 64 px
    float* out = a.act + (size_t)(p + 1) * G * NCH * TPX;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __sinf(acc[e] * 1e-3f) + 1.0f;
    if constexpr (MODE == 2) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)a.G * NCH * TPX * 4), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, r), ro, (int)((((size_t)t * NCH + li) * TPX + wave * 16 + 4 * kq) * 4), 0, 16);
    } else
        *reinterpret_cast<f32x4*>(out + ((size_t)t * NCH + li) * TPX + wave * 16 + 4 * kq) = r;
    if (wait) {
        if constexpr (MODE <= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_s_waitcnt(0);                  // every sc1 store of this wave acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + (size_t)p * G + t, ep, MODE <= 1 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_phase(const Args a, const int p) {
    __shared__ __attribute__((aligned(16))) float s_in[NCH * TPX];
    __shared__ __attribute__((aligned(16))) float s_w[WFL];
    phase_body<1>(a, p, blockIdx.x, false, 0u, s_in, s_w);
}
// ablations of the plain phase (graph-of-launches form): where do its ~6.5 us go?
template <int ABL>
__global__ __launch_bounds__(256) void k_abl(const Args a, const int p) {
    constexpr int PS = (ABL & 64) ? TPX + 20 : TPX;              // plane stride: 84 == 20 (mod 32) spreads the four k-lanes of an A fragment
    constexpr int RS = (ABL & 64) ? NCH * 9 + 7 : NCH * 9;        // weight row stride: 295 == 7 (mod 32): the 16 rows of a B fragment on 16 banks
    __shared__ __attribute__((aligned(16))) float s_in[NCH * PS + 16];
    __shared__ __attribute__((aligned(16))) float s_w[16 * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = blockIdx.x, G = a.G;
    const float* wsrc = a.w + (size_t)p * WFL;
    float wv[WFL / 256];
#pragma unroll
    for (int u = 0; u < WFL / 256; ++u) wv[u] = (ABL & 1) ? (float)(tid + u) : wsrc[tid + u * 256];
    const float* in = a.act + (size_t)p * G * NCH * TPX;
    f32x4 ra[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + k * 256;
        const int c = idx / (TPX / 4), q = idx % (TPX / 4);
        const int tt = q == 0 ? max(t - 1, 0) : (q == TPX / 4 - 1 ? min(t + 1, G - 1) : t);
        if (ABL & 2) ra[k] = f32x4{(float)idx, 1.f, 2.f, 3.f};
        else ra[k] = *reinterpret_cast<const f32x4*>(in + ((size_t)tt * NCH + c) * TPX + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < WFL / 256; ++u) { const int i = tid + u * 256; s_w[(i / (NCH * 9)) * RS + i % (NCH * 9)] = wv[u]; }
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int i4 = (tid + k * 256) * 4; *reinterpret_cast<f32x4*>(s_in + (i4 / TPX) * PS + i4 % TPX) = ra[k]; }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, kq = lane >> 4;
    if (ABL & 32) {
        // clean form: every LDS address = per-lane base + compile-time offset; one q group's 18 operands are read before its 9 MFMAs,
        // the next group's reads are issued ahead of them
        const float* ab = s_in + kq * PS + wave * 16 + li;
        const float* bb = s_w + li * RS + kq * 9;
        float av[2][9], bv[2][9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) { av[0][tp] = ab[tp]; bv[0][tp] = bb[tp]; }
#pragma unroll
        for (int q = 0; q < NCH / 4; ++q) {
            if (q + 1 < NCH / 4) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) { av[(q + 1) & 1][tp] = ab[(q + 1) * 4 * PS + tp]; bv[(q + 1) & 1][tp] = bb[(q + 1) * 36 + tp]; }
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][tp], bv[q & 1][tp], acc, 0, 0, 0);
        }
    } else if (!(ABL & 4)) {
#pragma unroll
        for (int q = 0; q < ((ABL & 16) ? NCH / 8 : NCH / 4); ++q)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const float av = s_in[(4 * q + kq) * TPX + ((wave * 16 + li + tp) & 63)];
                const float bv = s_w[li * NCH * 9 + (4 * q + kq) * 9 + tp];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
    } else {
        acc[0] = s_in[tid] + s_w[tid];
    }
    float* out = a.act + (size_t)(p + 1) * G * NCH * TPX;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __sinf(acc[e] * 1e-3f) + 1.0f;
    if (ABL & 8) { if (tid == 0) out[(size_t)t * NCH * TPX] = r[0]; }
    else *reinterpret_cast<f32x4*>(out + ((size_t)t * NCH + li) * TPX + wave * 16 + 4 * kq) = r;
}
__global__ void k_empty() {}

template <int MODE>
__global__ __launch_bounds__(256) void k_chain(const Args a) {
    __shared__ __attribute__((aligned(16))) float s_in[NCH * TPX];
    __shared__ __attribute__((aligned(16))) float s_w[WFL];
    const int p = blockIdx.x / a.G, t = blockIdx.x - p * a.G;
    const unsigned ep = *a.epoch;
    phase_body<MODE>(a, p, t, true, ep, s_in, s_w);
}
__global__ void k_bump(unsigned* epoch) { if (threadIdx.x == 0) *epoch += 1u; }

int main() {
    const int NP = 16;
    hipStream_t st; hipStreamCreate(&st);
    unsigned *epoch, *err; hipMalloc(&epoch, 4); hipMalloc(&err, 4);
    for (int G : {60, 120, 230, 470}) {
        Args a;
        float *w, *act; unsigned* flags;
        hipMalloc(&w, (size_t)NP * WFL * 4); hipMalloc(&act, (size_t)(NP + 1) * G * NCH * TPX * 4); hipMalloc(&flags, (size_t)NP * G * 4);
        std::vector<float> hw((size_t)NP * WFL), ha((size_t)(NP + 1) * G * NCH * TPX);
        for (auto& v : hw) v = (rand() % 200 - 100) * 1e-3f;
        for (auto& v : ha) v = (rand() % 200 - 100) * 1e-2f;
        hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        a.w = w; a.act = act; a.flags = flags; a.epoch = epoch; a.err = err; a.G = G; a.NP = NP;
        auto reset = [&] { hipMemcpy(act, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemset(flags, 0, (size_t)NP * G * 4); hipMemset(epoch, 0, 4); hipMemset(err, 0, 4); };
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // (a) graph of NP launches
        reset();
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        k_bump<<<1, 64, 0, st>>>(epoch);
        for (int p = 0; p < NP; ++p) k_phase<<<G, 256, 0, st>>>(a, p);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms_g; hipEventElapsedTime(&ms_g, e0, e1);
        std::vector<float> ref(ha.size());
        hipMemcpy(ref.data(), act, ref.size() * 4, hipMemcpyDeviceToHost);
        // (b) one launch with dataflow flags, three synchronisation forms
        float ms_c[3]; unsigned herr = 0; size_t bad = 0; std::vector<float> got(ha.size());
        for (int mode = 0; mode < 3; ++mode) {
            reset();
            hipGraph_t g2; hipGraphExec_t ge2;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            k_bump<<<1, 64, 0, st>>>(epoch);
            if (mode == 0) k_chain<0><<<NP * G, 256, 0, st>>>(a);
            else if (mode == 1) k_chain<1><<<NP * G, 256, 0, st>>>(a);
            else k_chain<2><<<NP * G, 256, 0, st>>>(a);
            hipStreamEndCapture(st, &g2); hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
            for (int i = 0; i < 3; ++i) hipGraphLaunch(ge2, st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st); for (int i = 0; i < 50; ++i) hipGraphLaunch(ge2, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_c[mode], e0, e1);
            hipMemcpy(got.data(), act, got.size() * 4, hipMemcpyDeviceToHost);
            unsigned he = 0; hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost); herr += he;
            for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
        }
        // (d) ablations of the plain phase
        auto timeg = [&](auto launch) {
            hipGraph_t gg; hipGraphExec_t gge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            k_bump<<<1, 64, 0, st>>>(epoch);
            for (int p = 0; p < NP; ++p) launch(p);
            hipStreamEndCapture(st, &gg); hipGraphInstantiate(&gge, gg, nullptr, nullptr, 0);
            for (int i = 0; i < 3; ++i) hipGraphLaunch(gge, st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st); for (int i = 0; i < 50; ++i) hipGraphLaunch(gge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms;
        };
        const float t_empty = timeg([&](int) { k_empty<<<G, 256, 0, st>>>(); });
        const float t_full = timeg([&](int p) { k_abl<0><<<G, 256, 0, st>>>(a, p); });
        const float t_now = timeg([&](int p) { k_abl<1><<<G, 256, 0, st>>>(a, p); });
        const float t_noi = timeg([&](int p) { k_abl<2><<<G, 256, 0, st>>>(a, p); });
        const float t_nold = timeg([&](int p) { k_abl<3><<<G, 256, 0, st>>>(a, p); });
        const float t_nok = timeg([&](int p) { k_abl<4><<<G, 256, 0, st>>>(a, p); });
        const float t_halfk = timeg([&](int p) { k_abl<16><<<G, 256, 0, st>>>(a, p); });
        const float t_nost = timeg([&](int p) { k_abl<8><<<G, 256, 0, st>>>(a, p); });
        const float t_only = timeg([&](int p) { k_abl<15><<<G, 256, 0, st>>>(a, p); });
        const float t_clean = timeg([&](int p) { k_abl<32><<<G, 256, 0, st>>>(a, p); });
        const float t_cf = timeg([&](int p) { k_abl<96><<<G, 256, 0, st>>>(a, p); });
        // (c) the bump kernel alone (subtracted from both)
        hipGraph_t g3; hipGraphExec_t ge3;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        k_bump<<<1, 64, 0, st>>>(epoch);
        hipStreamEndCapture(st, &g3); hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0);
        hipGraphLaunch(ge3, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int i = 0; i < 50; ++i) hipGraphLaunch(ge3, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms_b; hipEventElapsedTime(&ms_b, e0, e1);
        printf("G %4d blocks/phase, %d phases:  graph of launches %.2f us/phase   one launch with dataflow flags: acquire spin %.2f, relaxed spin + fences %.2f, sc1 accesses without fences %.2f us/phase   (bump alone %.2f us; spin time-outs %u, mismatches %zu)\n",
               G, NP, (ms_g - ms_b) * 1e3 / (50 * NP), (ms_c[0] - ms_b) * 1e3 / (50 * NP), (ms_c[1] - ms_b) * 1e3 / (50 * NP), (ms_c[2] - ms_b) * 1e3 / (50 * NP), ms_b * 1e3 / 50, herr, bad);
        auto pp = [&](float ms) { return (ms - ms_b) * 1e3 / (50 * NP); };
        printf("        ablations (us/phase): empty kernel %.2f | full %.2f | no weight load %.2f | no input load %.2f | neither load %.2f | no K loop %.2f | half K loop %.2f | no store %.2f | LDS write + barrier only %.2f | clean pipelined K loop %.2f | the same, bank-conflict-free strides %.2f\n",
               pp(t_empty), pp(t_full), pp(t_now), pp(t_noi), pp(t_nold), pp(t_nok), pp(t_halfk), pp(t_nost), pp(t_only), pp(t_clean), pp(t_cf));
        hipFree(w); hipFree(act); hipFree(flags);
    }
    return 0;
}
