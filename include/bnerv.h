/*
 * bnerv.h -- C-ABI of libbnerv_hip.so: the MI355X (gfx950) kernels of the Boosting-NeRV conditional-decoder
 * train path.  This is the drop-in boundary.
 *
 * The reference (Xinjie-Q/Boosting-NeRV) is pure Python/PyTorch and has NO FFI: every arithmetic op on the path is an
 * ATen call made from the Python files cited below.  Each entry point here replaces the ATen call sequence of one
 * reference call site; the Python host package (boosting_nerv_amd/) binds them with ctypes and mirrors the reference's
 * module API (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and ints; no torch types.  All tensors are DEVICE pointers to contiguous fp32 NCHW
 *     (OIHW for weights) unless stated otherwise; the caller owns every buffer (no ownership transfer, no allocation
 *     here); workspaces are passed in explicitly and their sizes are given by the matching *_ws_bytes() helper.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call never synchronises.
 *   - Return 0 on success, negative BNERV_E_* on failure; bnerv_last_error() returns a thread-local message.
 *   - Thread-safe per stream; no global mutable state except read-only twiddle tables created through bnerv_fft_*.
 */
#ifndef BNERV_H
#define BNERV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNERV_ABI_VERSION 9

#define BNERV_OK 0
#define BNERV_E_ARG (-1)      /* bad argument / unsupported shape */
#define BNERV_E_LAUNCH (-2)   /* hip launch error */
#define BNERV_E_WS (-3)       /* workspace too small */

int bnerv_abi_version(void);
const char* bnerv_last_error(void);
/* name of the gfx arch the code object was built for ("gfx950") */
const char* bnerv_build_arch(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Positional encoding.  Replaces PositionEncoding.forward (model_blocks.py:120-126):
 *   v = pos * bases ; out = cat[sin v, cos v]  -> [N, 2L]
 * f32 form: pos fp32, ONE IEEE fp32 multiply, accurate sinf/cosf (model_nerv.py:47-48, model_enerv.py:286-292).
 * f64 form: pos fp64, product and sin/cos in fp64, result cast to fp32 (HNeRV_Boost, model_hnerv.py:241).
 * `bases` is the fp32 table built by the host with the reference's own torch expression (model_blocks.py:113-117).
 * No backward: pos carries no gradient.
 * ------------------------------------------------------------------------------------------------------------------ */
int bnerv_pe_fwd_f32(void* stream, const float* pos, const float* bases, float* out, int N, int L);
int bnerv_pe_fwd_f64(void* stream, const double* pos, const float* bases, float* out, int N, int L);
/* f32 form on fp64 positions: pos is rounded to fp32 first (the `input[:, None].float()` of model_nerv.py:47 / model_enerv.py:286), then as _f32 */
int bnerv_pe_fwd_f32_from_f64(void* stream, const double* pos, const float* bases, float* out, int N, int L);

/* ------------------------------------------------------------------------------------------------------------------
 * Grouped dense layers on [B, I] vectors (1x1 convs on [B,C,1,1]).  Replaces the CustomConv2d(kernel_size=1) calls of
 * NeRV_MLP (model_blocks.py:66-71: stem / stem_t / t_branch) and of SFTLayer's four 1x1 convs (model_blocks.py:92-105).
 * One launch evaluates up to BNERV_MAX_DENSE_GROUPS independent layers  y = act(W x + b).
 *   act: 0 none, 1 relu, 2 sin.   aux (sin only, may be NULL): cos(Wx+b), saved for the backward.
 * ------------------------------------------------------------------------------------------------------------------ */
#define BNERV_MAX_DENSE_GROUPS 40
#define BNERV_ACT_NONE 0
#define BNERV_ACT_RELU 1
#define BNERV_ACT_SIN 2

typedef struct {
    const float* x;   /* [B, I] */
    const float* w;   /* [O, I] */
    const float* b;   /* [O] or NULL */
    float* y;         /* [B, O] */
    float* aux;       /* [B, O] or NULL */
    int I, O, act, _pad;
} bnerv_dense_fwd_desc;

typedef struct {
    const float* x;    /* [B, I]  layer input */
    const float* w;    /* [O, I] */
    const float* y;    /* [B, O]  layer output (relu mask) */
    const float* aux;  /* [B, O]  cos(pre) for sin */
    const float* dy;   /* [B, O]  incoming gradient */
    float* dpre;       /* [B, O]  scratch: gradient wrt pre-activation (written) */
    float* dw;         /* [O, I]  written */
    float* db;         /* [O]     written (may be NULL) */
    float* dx_part;    /* [nchunk, B, I] partial input gradients (NULL: not needed); nchunk = ceil(O / BNERV_DENSE_DX_CHUNK) */
    int I, O, act, _pad;
} bnerv_dense_bwd_desc;

#define BNERV_DENSE_DX_CHUNK 64

int bnerv_dense_grouped_fwd(void* stream, const bnerv_dense_fwd_desc* groups, int n_groups, int B);
int bnerv_dense_grouped_bwd(void* stream, const bnerv_dense_bwd_desc* groups, int n_groups, int B);

/* The time-embedding branch of NeRV_Boost.forward except the stem's second layer, as ONE launch (ABI 8; csrc/dense.hip):
 *   pe = PositionEncoding(pos)                               model_nerv.py:47, model_blocks.py:120-126   (fp64 position rounded to fp32 first)
 *   stem layer 0 : sy0 = sin(sw0 pe + sb0) [SH]                                model_nerv.py:48-50, NeRV_MLP model_blocks.py:66-71
 *   stem_t       : ty0 = sin(tw0 pe + tb0) [TH];  ty1 = sin(tw1 ty0 + tb1) [TO] = z_t
 *   every TAT modulation MLP i:  hs_i = relu(w1_i z_t + b1_i) [TO];  out_i = w2_i hs_i + b2_i [C_i]          SFTLayer, model_blocks.py:92-105
 * (the stem's second layer follows as an ordinary bnerv_dense_grouped_fwd launch).  Every tensor the five-launch form leaves behind --
 * outputs and the cosines saux0 / taux* of the sin layers -- is written, so the backward is the unchanged bnerv_dense_grouped_bwd sequence.
 * Requires w1_i [TO, TO] and w2_i [C_i, TO] row-major, sw0 / tw0 [*, 2L].  Returns 1 when the shapes are not this kernel's (B > 4, 2L > 256,
 * TH > 64, TO > 32, C_i > 128, more than BNERV_MAX_DENSE_GROUPS MLPs, unaligned weights): the caller issues the grouped launches. */
typedef struct { const float* w1; const float* b1; const float* w2; const float* b2; float* hs; float* out; int C; int _pad; } bnerv_time_branch_mlp;
typedef struct {
    const double* pos;     /* [B] */
    const float* bases;    /* [L] */
    float* pe;             /* [B, 2L] */
    const float* sw0; const float* sb0;                                          /* stem layer 0: [SH, 2L], [SH] */
    float* sy0; float* saux0;                                                    /* [B, SH] x 2 */
    const float* tw0; const float* tb0; const float* tw1; const float* tb1;      /* stem_t: [TH, 2L], [TH], [TO, TH], [TO] */
    float* ty0; float* taux0; float* ty1; float* taux1;                          /* [B, TH] x 2, [B, TO] x 2 */
    int B, L, SH, TH, TO, n_mlp;
} bnerv_time_branch_desc;
int bnerv_time_branch_fwd(void* stream, const bnerv_time_branch_desc* d, const bnerv_time_branch_mlp* mlps /* [n_mlp] */);

/* Stand-alone TAT affine  y = x*(scale[b,c]+1) + shift[b,c]  (SFTLayer.forward, model_blocks.py:101-105) and its backward:
 *   dx = g*(scale+1);  part[chunk][0][b,c] = sum_chunk g*x;  part[chunk][1][b,c] = sum_chunk g
 * finish with bnerv_reduce_slabs(part, BNERV_SFT_CHUNKS, 2*B*C, out) -> out[0][b,c] = dscale, out[1][b,c] = dshift.
 * (Inside the decoder blocks the affine is fused into the conv prologues; these two exist for the module-level API.) */
#define BNERV_SFT_CHUNKS 32
int bnerv_sft_affine_fwd(void* stream, const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW);
int bnerv_sft_affine_bwd(void* stream, const float* x, const float* scale, const float* g, float* dx, float* part, int B, int C, int HW);

/* LayerNorm over the channel axis of an NCHW tensor (reference: model_blocks.py:250-270 `LayerNorm`, data_format
 * "channels_first"; the "channels_last" form of the ConvNeXt block is the same arithmetic on the permuted tensor):
 *   y[b,c,p] = w[c] * (x[b,c,p] - mean_c) / sqrt(var_c + eps) + b[c],  var = mean_c (x - mean_c)^2.   1 <= C <= 64.
 * bwd: dx as autograd's; dwb = [2][C] = (dw, db), summed over b and p in a fixed order (ws: bnerv_lncf_bwd_ws_bytes). */
int bnerv_lncf_fwd(void* stream, const float* x, const float* w, const float* b, float* y, int B, int C, int HW, float eps);
size_t bnerv_lncf_bwd_ws_bytes(int B, int C, int HW);
int bnerv_lncf_bwd(void* stream, const float* x, const float* w, const float* dy, float* dx, float* dwb, void* ws, size_t ws_bytes,
                   int B, int C, int HW, float eps);

/* out[i] = sum_{s<n_slabs} slabs[s*count + i]   (deterministic finish of every split reduction in this library) */
int bnerv_reduce_slabs(void* stream, const float* slabs, int n_slabs, int count, float* out);

/* Deferred form of the same reduction.  On MI355X a tiny dependent kernel between two big ones costs ~10 us of pipeline,
 * so the ~45 slab reductions of a train step are queued instead of launched: the next bnerv_conv_igemm / bnerv_conv_wgrad
 * launch GIVEN THE SAME CONTEXT whose kernel can host them executes the queued reductions at the end of its least-loaded
 * blocks (same fixed summation order whoever executes them).  `slabs` and `out` must stay valid, and `out` must not be read,
 * until a hosting launch or bnerv_flush_deferred() has been issued on the stream; bnerv_flush_deferred launches whatever is
 * still queued (a no-op when the queue is empty).
 *
 * The queue lives in a CALLER-OWNED context: the library keeps no mutable global state.  One context serves one stream
 * (every launch that names it must go to that stream, from one thread at a time); contexts are independent of each other, so
 * two streams / two models in one process never see each other's jobs.  A NULL context is legal everywhere: nothing is
 * hosted, and a deferral request is executed immediately instead. */
typedef struct bnerv_ctx bnerv_ctx;
int bnerv_ctx_create(bnerv_ctx** out);
void bnerv_ctx_destroy(bnerv_ctx* ctx);
/* The context also owns a device scratch buffer (pre-split weight fragments of the wide split conv kernels).  It grows on demand,
 * except while its stream is being captured into a graph (allocation is illegal there: such a call falls back to the f32 kernels).
 * A caller that is about to capture on a fresh stream reserves what the eager steps needed:  bnerv_ctx_reserve(new_ctx,
 * bnerv_ctx_scratch_bytes(old_ctx)). */
size_t bnerv_ctx_scratch_bytes(const bnerv_ctx* ctx);
int bnerv_ctx_reserve(bnerv_ctx* ctx, size_t bytes);
/* Weight-fragment plan of a repeated step (ABI 4).  Every wide split conv call (forward conv, up-conv, data gradient) is preceded
 * by a small launch that splits its weight tensor into 16-bit matrix fragments; inside one train step the weights do not change
 * between the first forward conv and the last data gradient (the optimizer runs after backward), so a step that is replayed as a
 * graph can prepare ALL of them in one launch at its start:
 *     bnerv_ctx_wplan_record(ctx);  <one forward + backward with this context, eager>;  n = bnerv_ctx_wplan_freeze(ctx);
 *     per step (normally inside the capture):  bnerv_ctx_wplan_run(ctx, stream);  <forward, backward>;  bnerv_ctx_wplan_end(ctx);
 * record: forget any previous plan and note, for every wide split conv call of this context, (weight pointer, layout, fragment
 * geometry).  freeze: allocate one arena for the distinct entries (device synchronize; not inside a capture); returns their number
 * (>= 0) or a negative error.  run: ONE launch that writes every entry's fragments from the CURRENT weights; from then until
 * wplan_end, a conv call of this context whose weight pointer / geometry matches an entry reads the arena and launches no
 * preparation of its own; calls that match nothing, and all calls outside run..end, behave as before (own preparation, scratch
 * buffer).  The caller guarantees that the planned weight tensors are not written between run and end and that their addresses
 * stay what they were at record time; the fragments are the same bits either way, so results do not depend on the plan. */
int bnerv_ctx_wplan_record(bnerv_ctx* ctx);
int bnerv_ctx_wplan_freeze(bnerv_ctx* ctx);
int bnerv_ctx_wplan_run(bnerv_ctx* ctx, void* stream);
/* bnerv_ctx_wplan_run and bnerv_fetch_frame (below) as ONE launch (ABI 8): both open a captured step on a resident clip and neither depends on the
 * other.  Returns 1 when the context has no frozen plan with work -- the caller then calls bnerv_fetch_frame alone. */
int bnerv_ctx_wplan_run_fetch(bnerv_ctx* ctx, void* stream, const float* clip, const double* norms, const float* sel_dev, int n_frames, size_t frame_elems,
                              float* dst_img, double* dst_norm);
int bnerv_ctx_wplan_end(bnerv_ctx* ctx);
int bnerv_ctx_wplan_entries(const bnerv_ctx* ctx);     /* entries of the frozen plan (0: none) */
int bnerv_reduce_slabs_deferred(bnerv_ctx* ctx, void* stream, const float* slabs, int n_slabs, int count, float* out);
int bnerv_flush_deferred(bnerv_ctx* ctx, void* stream);
int bnerv_deferred_pending(const bnerv_ctx* ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution (fp32 MFMA 16x16x4), stride 1, square kernel k in {1,3}, zero padding (k-1)/2, with fused
 * input prologue and output epilogue.  One kernel family serves the forward conv and the data gradient.
 * Replaces CustomConv2d.forward = F.conv2d (lib/quant_ops.py:39-41) as used by UpConv (model_blocks.py:196-220),
 * ResBlock_SFT (model_blocks.py:74-89), Conv_Up_Block (model_enerv.py:73-102), DownConv 'conv' (model_blocks.py:184-185)
 * and head_layer (model_nerv.py:41,56), together with the elementwise ops around it:
 *   SFTLayer affine x*(scale+1)+shift (model_blocks.py:101-105), GELU (model_blocks.py:81), PixelShuffle + Sin
 *   (model_blocks.py:213-218, :129-134), residual add (model_blocks.py:89), OutImg tanh (model_blocks.py:57-63),
 * and, for the data gradient, autograd's backward of the same.
 * ------------------------------------------------------------------------------------------------------------------ */
/* input prologues */
#define BNERV_IN_PLAIN 0         /* a = x */
#define BNERV_IN_AFFINE 1        /* a = x*(1+scale[b,c]) + shift[b,c]          (padding applies AFTER the affine) */
#define BNERV_IN_GELU_AFFINE 2   /* a = gelu(x)*(1+scale[b,c]) + shift[b,c] */
#define BNERV_IN_UNSHUFFLE 3     /* a[c*s*s+i*s+j][y][x] = x[c][y*s+i][x*s+j]  (pixel-unshuffle gather, s = in_s) */
#define BNERV_IN_TANHGRAD 4      /* a = x * 0.5*(1-(2*aux0-1)^2)               (x = dL/dimg, aux0 = img = OutImg output) */
/* output epilogues (v = conv result, o = output element) */
#define BNERV_EP_BIAS 0          /* out = v + bias                              (optionally pixel-shuffled by out_s) */
#define BNERV_EP_BIAS_SIN 1      /* u = v + bias; out = sin u; out2 = cos u     (pixel-shuffled by out_s) */
#define BNERV_EP_BIAS_RES 2      /* out = v + bias + aux0 */
#define BNERV_EP_BIAS_TANH 3     /* out = tanh(v + bias)*0.5 + 0.5 */
#define BNERV_EP_PLAIN 4         /* out = v */
#define BNERV_EP_DGELU 5         /* out = v*(1+scale[b,c])*gelu'(aux0);   partial: ds += v*gelu(aux0), dt += v */
#define BNERV_EP_DSIN 6          /* t = aux1 + v*(1+scale[b,c]); out = t*aux2 (aux2 NULL: 1); partial: ds += v*aux0, dt += v */
#define BNERV_EP_BIAS_GELU 7     /* u = v + bias; out = gelu(u); out2 = gelu'(u) (out2 may be NULL: decode)  (the TAT block saves both instead of u: the three
                                    consumers -- next conv, its weight gradient, the dGELU epilogue -- then need no erf/exp) */
#define BNERV_EP_DGELU_SAVED 8   /* out = v*(1+scale[b,c])*aux0 (aux0 = saved gelu');  partial: ds += v*aux1 (aux1 = saved gelu), dt += v */

typedef struct {
    const float* x;       /* input, conv-space [B, Cin, H, W]  (IN_UNSHUFFLE: stored as [B, Cin/s^2, H*s, W*s]) */
    const float* w;       /* weight tensor [wCo, wCi, k, k] (OIHW, as in state_dict) */
    const float* bias;    /* [Cout] or NULL */
    float* out;           /* conv-space [B, Cout, H, W], stored pixel-shuffled as [B, Cout/s^2, H*s, W*s] when out_s>1 */
    float* out2;          /* EP_BIAS_SIN: cos(u), same layout as out (may be NULL) */
    const float* aux0;    /* see epilogue / prologue tables */
    const float* aux1;
    const float* aux2;
    const float* scale;   /* [B, Cin] for IN_AFFINE / IN_GELU_AFFINE;  [B, Cout] for EP_DGELU / EP_DSIN */
    const float* shift;   /* [B, Cin] for IN_AFFINE / IN_GELU_AFFINE */
    float* partial;       /* EP_DGELU / EP_DSIN: [R][B][2][Cout] per-tile partial sums (written), R = bnerv_conv_partial_rows(d);
                             finish with bnerv_reduce_slabs(partial, R, B*2*Cout, out) -> out[b][0][c] = ds, out[b][1][c] = dt.
                             EP_PLAIN: optional split-K workspace (see bnerv_conv_splitk_ws_bytes) */
    int B, Cin, Cout, H, W;
    int k;                /* 1 or 3 */
    int in_mode, ep_mode;
    int in_s, out_s;      /* shuffle factors (1 = none) */
    int transposed;       /* 0: forward  (Cout=wCo, Cin=wCi, W(co,ci,t) = w[co][ci][t]);
                             1: data gradient (Cout=wCi, Cin=wCo, W(co,ci,t) = w[ci][co][k*k-1-t]) */
    int wCo, wCi;
    bnerv_ctx* ctx;       /* deferred-reduction context of the stream (may be NULL) */
} bnerv_conv_desc;

/* number of 8x32 spatial tiles per sample of an H x W conv-space image */
int bnerv_conv_tiles(int H, int W);
/* EP_DGELU / EP_DSIN: number of per-sample rows R this descriptor's kernel writes into `partial` ([R][B][2][Cout]); the
 * kernel is chosen from the shape and pointer alignment, so fill in every field except `partial` before asking. */
int bnerv_conv_partial_rows(const bnerv_conv_desc* d);
/* EP_PLAIN only: bytes of split-K workspace this layer wants (0 = none).  Layers with a long K loop and almost no
 * spatial parallelism (the low-resolution data gradients, e.g. Cin = 750 at 9x16) split their input-channel range over
 * work items; pass a buffer of this size in `partial` and bnerv_conv_igemm finishes with a deterministic slab reduction.
 * With partial == NULL the layer runs unsplit. */
size_t bnerv_conv_splitk_ws_bytes(const bnerv_conv_desc* d);
int bnerv_conv_igemm(void* stream, const bnerv_conv_desc* d);

/* Weight + bias gradient (autograd's backward of F.conv2d wrt weight/bias at the same call sites):
 *   dw[co][ci][t] = sum_{b,p} g[b][co][p] * a[b][ci][p + t - pad],  db[co] = sum_{b,p} g[b][co][p]
 * a = prologue(x) (IN_PLAIN / IN_AFFINE / IN_GELU_AFFINE), g gathered with pixel-unshuffle g_s or through tanh'
 * (g_mode BNERV_IN_TANHGRAD with gaux = img).  Split over spatial tiles into slabs, finished deterministically. */
typedef struct {
    const float* x;       /* [B, Cin, H, W] */
    const float* g;       /* [B, Cout, H, W] conv-space, stored shuffled [B, Cout/s^2, H*s, W*s] when g_s>1 */
    const float* gaux;    /* img for BNERV_IN_TANHGRAD */
    const float* scale;   /* [B, Cin] */
    const float* shift;   /* [B, Cin] */
    float* dw;            /* [Cout, Cin, k, k] written */
    float* db;            /* [Cout] written (may be NULL) */
    void* ws;             /* workspace of bnerv_conv_wgrad_ws_bytes() bytes */
    size_t ws_bytes;
    int B, Cin, Cout, H, W, k;
    int in_mode;          /* prologue on x */
    int g_mode;           /* BNERV_IN_PLAIN / BNERV_IN_UNSHUFFLE / BNERV_IN_TANHGRAD */
    int g_s;
    int defer_finish;     /* 1: queue the slab reduction into dw/db (see bnerv_reduce_slabs_deferred) instead of launching it */
    bnerv_ctx* ctx;       /* deferred-reduction context of the stream (NULL: nothing hosted, defer_finish ignored) */
} bnerv_wgrad_desc;

size_t bnerv_conv_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int k);
int bnerv_conv_wgrad(void* stream, const bnerv_wgrad_desc* d);

#define BNERV_LOSS_STATS 5
/* ------------------------------------------------------------------------------------------------------------------
 * CEM compression path: the per-step quantise + rate term of model.cal_params(entropy_model)
 * (model_hnerv.py:292-303) fused over tensors.  Per tensor (per-tensor `scale`, Gaussian rate model):
 *   code = w / scale;  dequant = round(code) * scale                                   Scale_T.forward, lib/transform_ops.py:239-251
 *   mu = mean(code), sigma = std(code);  x = code + noise (training) | round(code)      DiffEntropyModel, lib/entropy_model.py:21-43
 *   bits = sum max(-log2(Phi((x+.5-mu)/sigma) - Phi((x-.5-mu)/sigma) + 1e-5), 0)
 * stats[item] = {bits, mu, sigma, n}.  Backward: d_bits[item] = dL/d bits, d_dequant = dL/d dequant (NULL: none);
 * writes dw (NULL: skipped) and dscale[item], including the paths through mu, sigma and the LowerBound gate (:100-114).
 * `first` = index of the chunk's first tensor in stats / d_bits / dscale (tables travel by value, <= 48 tensors per launch). */
#define BNERV_CEM_MAX_TENSORS 48
typedef struct { const float* w; const float* scale; const float* noise; float* dequant; int n; int _pad; } bnerv_cem_item;
typedef struct { bnerv_cem_item it[BNERV_CEM_MAX_TENSORS]; int n_items; int training; int first; int _pad; } bnerv_cem_chunk;
typedef struct { const float* w; const float* scale; const float* noise; const float* d_dequant; float* dw; int n; int _pad; } bnerv_cem_item_bwd;
typedef struct { bnerv_cem_item_bwd it[BNERV_CEM_MAX_TENSORS]; int n_items; int training; int first; int _pad; } bnerv_cem_chunk_bwd;
/* Every tensor is cut into chunks of 8192 elements (one block each); the chunk partial sums (f64, fixed order) live in a caller-provided
 * workspace of bnerv_cem_ws_bytes(n_items, numel of the largest tensor of the chunk) bytes per call (ABI 3). */
size_t bnerv_cem_ws_bytes(int n_items, int max_n);
int bnerv_cem_scale_fwd(void* stream, const bnerv_cem_chunk* chunk, float* stats, void* ws, size_t ws_bytes);
int bnerv_cem_scale_bwd(void* stream, const bnerv_cem_chunk_bwd* chunk, const float* stats, const float* d_bits, float* dscale,
                        void* ws, size_t ws_bytes);

/* ------------------------------------------------------------------------------------------------------------------
 * GEMM-shaped dense work on v_mfma_f32_16x16x4_f32 (csrc/gemm.hip).
 *
 * Dense layers applied to many rows (B >= 16: the token MLPs of E-NeRV's transformer stem, NeRV_MLP at larger batches;
 * reference model_blocks.py:66-71, model_enerv.py:22-60) -- same arguments as one group of bnerv_dense_grouped_*:
 *   fwd: y [B,O] = act(x [B,I] w[O,I]^T + b), aux [B,O] = cos(pre) for BNERV_ACT_SIN (may be NULL)
 *   bwd: dw [O,I], db [O] (may be NULL), dx [B,I] (may be NULL) from dy [B,O]; y (relu) / aux (sin) as the activation needs. */
int bnerv_dense_gemm_fwd(void* stream, const float* x, const float* w, const float* b, float* y, float* aux, int B, int I, int O, int act);
size_t bnerv_dense_gemm_bwd_ws_bytes(int B, int I, int O);      /* 0 for small B; row-split partial weight gradients otherwise */
int bnerv_dense_gemm_bwd(void* stream, const float* x, const float* w, const float* y, const float* aux, const float* dy,
                         float* dx, float* dw, float* db, void* ws, size_t ws_bytes, int B, int I, int O, int act);

/* Pointwise MLP of the ConvNeXt encoder block (reference model_blocks.py:245-258, channels_last form:
 *   x = pwconv2(gelu(pwconv1(x))); x = gamma * x; return input + x ) on NCHW tensors, C in {16, 32, 48, 64}:
 *   fwd: out [B,C,HW] = inp + gamma * (w2 [C,4C] gelu(w1 [4C,C] x + b1) + b2);  hsave [B,4C,HW] (may be NULL) keeps w1 x + b1
 *   bwd: from h1 (= hsave) and dout: dx [B,C,HW] = d loss / d x, and the two pixel-contraction operands of the weight gradients,
 *        gbuf = gelu(h1), dhbuf = d loss / d h1 ([B,4C,HW] each).  The weight / bias / gamma gradients follow from two
 *        bnerv_conv_wgrad (k = 1) calls: (x, dhbuf) -> dw1, db1 and (gbuf, dout) -> S, t with dw2 = gamma S, db2 = gamma t,
 *        dgamma = rowsum(w2 * S) + b2 * t.  (d loss / d inp is dout itself.) */
int bnerv_cnx_mlp_fwd(void* stream, const float* x, const float* inp, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* gamma, float* out, float* hsave, int B, int C, int HW);
int bnerv_cnx_mlp_bwd(void* stream, const float* h1, const float* dout, const float* w1, const float* w2, const float* gamma,
                      float* dx, float* gbuf, float* dhbuf, int B, int C, int HW);
/* the [C x 4C] bookkeeping of that backward in one launch: dw2 = gamma S, db2 = gamma t, dgamma = rowsum(w2 * S) + b2 * t */
int bnerv_cnx_param_grads(void* stream, const float* S, const float* t, const float* w2, const float* b2, const float* gamma,
                          float* dw2, float* db2, float* dgamma, int C);

/* ------------------------------------------------------------------------------------------------------------------
 * Range-ANS entropy coder of the compression report (HOST buffers; csrc/ans.cpp).  Replaces constriction's AnsCoder +
 * QuantizedGaussian / Categorical models behind `real_bitrate` (lib/entropy_model.py:46-62, :65-81; consumed at
 * train_nerv_compression.py:484-512, 560-577): rANS, 64-bit state, 32-bit words, 24-bit probabilities, symbols pushed in
 * reverse so that decoding yields them in message order.
 *   *_encode_*  returns the number of 32-bit words of the compressed message (also when out == NULL or cap_words is too small:
 *               nothing is written then), or -1 on a bad argument (bnerv_last_error()).
 *   *_decode_*  the inverse; BNERV_OK or an error code.
 *   gaussian:    integer symbols in [min_sym, max_sym] under the leaky quantised Gaussian(mean, std) on that support
 *   categorical: symbols 0 .. K-1 under probs[K] (any non-negative weights; normalised internally) */
long bnerv_ans_encode_gaussian(const int32_t* symbols, size_t n, int32_t min_sym, int32_t max_sym, double mean, double std, uint32_t* out, size_t cap_words);
int bnerv_ans_decode_gaussian(const uint32_t* words, size_t n_words, size_t n, int32_t min_sym, int32_t max_sym, double mean, double std, int32_t* symbols_out);
long bnerv_ans_encode_categorical(const int32_t* symbols, size_t n, const double* probs, int K, uint32_t* out, size_t cap_words);
int bnerv_ans_decode_categorical(const uint32_t* words, size_t n_words, size_t n, const double* probs, int K, int32_t* symbols_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Depthwise KxK convolution (K odd <= 7, stride 1, padding K/2) of the ConvNeXt encoder block of HNeRV_Boost
 * (model_blocks.py:223-247: nn.Conv2d(dim, dim, 7, padding=3, groups=dim)); first kernels of SURVEY 8(f) row N3.
 *   bnerv_dwconv_fwd(flip=0): y = conv(x, w) + bias;   flip=1: the data gradient (taps flipped, bias ignored: pass g as x)
 *   bnerv_dwconv_wgrad: dwb[C][K*K+1] = weight gradient rows with the bias gradient in the last column (slabs in ws;
 *   defer_ctx != NULL queues the slab reduction there, see bnerv_reduce_slabs_deferred).  x, y, g: [B, C, H, W];  w: [C, 1, K, K]. */
int bnerv_dwconv_fwd(void* stream, const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int K, int flip);
size_t bnerv_dwconv_wgrad_ws_bytes(int B, int C, int H, int W, int K);
int bnerv_dwconv_wgrad(void* stream, const float* x, const float* g, float* dwb, void* ws, size_t ws_bytes, int B, int C, int H, int W, int K, bnerv_ctx* defer_ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * Loss and metrics.  Replaces loss_fn (hnerv_utils.py:335-397; variants L1, L2, L1_freq, Fusion10, Fusion10_freq)
 * including its autograd backward, and psnr_fn_single (hnerv_utils.py:400-403).
 *   loss_b = c_l1 * mean|d| + c_l2 * mean d^2 + c_ms * (1 - ms_ssim_b) + c_fft * mean|FFT2(pred)-FFT2(target)|_{re,im}
 * with d = pred - target; the reported loss is mean_b loss_b (batch_average=True) and `grad` = d(loss)/d(pred).
 *   L1: c_l1=1   L2: c_l2=1   L1_freq: c_l1=60,c_fft=1   Fusion10: c_l1=.7,c_ms=.3   Fusion10_freq: c_l1=42,c_ms=18,c_fft=1
 * MS-SSIM follows pytorch_msssim 0.2.1 (win 11, sigma 1.5, 5 levels) -- third-party, PARITY UNPINNED.
 * The 2-D DFT is a mixed-radix LDS FFT; H and W may have any prime factors <= BNERV_FFT_MAX_RADIX.
 * stats_out: [B, BNERV_LOSS_STATS] = {loss_b, sum|d|, sum d^2, ms_ssim_b, psnr_b (hnerv_utils.py:400-403)};  loss_out: [1].
 * ------------------------------------------------------------------------------------------------------------------ */
#define BNERV_FFT_MAX_RADIX 31
#define BNERV_MSSSIM_LEVELS 5

typedef struct {
    const float* pred;    /* [B, C, H, W] */
    const float* target;  /* [B, C, H, W] */
    float* grad;          /* [B, C, H, W] written (may be NULL: value only) */
    float* loss_out;      /* [1] */
    float* stats_out;     /* [B, BNERV_LOSS_STATS] */
    void* ws;
    size_t ws_bytes;
    int B, C, H, W;
    float c_l1, c_l2, c_ms, c_fft;
} bnerv_loss_desc;

size_t bnerv_loss_ws_bytes(int B, int C, int H, int W, int use_ms, int use_fft);
/* Build the read-only FFT twiddle tables for H x W frames ahead of time.  Tables are otherwise built on first use,
 * which allocates and copies synchronously -- not allowed while `stream` is being captured into a hipGraph. */
int bnerv_fft_prepare(int H, int W);
int bnerv_loss_fwd_bwd(void* stream, const bnerv_loss_desc* d);
/* per-sample MS-SSIM only (evaluate(): msssim_fn_single, hnerv_utils.py:410-412); out [B] */
int bnerv_msssim(void* stream, const float* x, const float* y, float* out, void* ws, size_t ws_bytes, int B, int C, int H, int W);
/* (ABI 9) The output head's tanh-gradient as a tensor: gt = g * 0.5 (1 - (2 img - 1)^2), the derivative of OutImg's tanh(v) * 0.5 + 0.5
 * (reference model_blocks.py:57-63) applied to the incoming gradient -- what the IN_TANHGRAD prologue computes on the fly -- with the
 * per-block channel sums of gt in part [B * bnerv_tanh_grad_blocks(HW)][C] (sum over the first index = the head's bias gradient:
 * bnerv_reduce_slabs(part, B * blocks, C, db)).  Used by the 3x3 head of HNeRV_Boost (model_hnerv.py:214), whose weight gradient runs
 * with input and gradient swapped (38 input channels on the MFMA M side). */
int bnerv_tanh_grad_blocks(int HW);
int bnerv_tanh_grad(void* stream, const float* g, const float* img, float* gt, float* part, int B, int C, int HW);

/* psnr[b] = -10 log10(mean_{CHW}(out-gt)^2 + 1e-9)  (hnerv_utils.py:400-403); ws: bnerv_psnr_ws_bytes() */
size_t bnerv_psnr_ws_bytes(int B, int C, int H, int W);
int bnerv_psnr(void* stream, const float* out, const float* gt, float* psnr, void* ws, size_t ws_bytes, int B, int C, int H, int W);

/* Paired launch (ABI 5): the data gradient of a 3x3 convolution with at most 12 channels on both sides (plain input; epilogues
 * EP_DGELU_SAVED / EP_DSIN / EP_PLAIN) and a weight gradient of the same image size that does not depend on it, in ONE grid.  Inside
 * the backward of ResBlock_SFT / NeRVBlock (model_blocks.py:83-89, :34-39) autograd computes, for each of the three convolutions, the
 * weight gradient and the input gradient from the SAME incoming gradient: (dW1 | d conv1), (dW0 | d conv0), (dW_block | d block conv);
 * issued as two launches each half leaves most of the chip idle through its prologue and tail (one tile per block at 180x320).
 * Both descriptors are exactly those bnerv_conv_wgrad / bnerv_conv_igemm would take (same workspaces; the weight gradient must be
 * deferred -- defer_finish with a context -- and both must name the same context).  Returns BNERV_OK when the pair was launched, 1 when
 * this pair is not one the launch takes (issue the two calls separately, weight gradient first), negative BNERV_E_* on error. */
int bnerv_conv_wgrad_pair(void* stream, const bnerv_conv_desc* conv, const bnerv_wgrad_desc* wgrad);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused multi-tensor Adan step.  Replaces Adan.step -> _multi_tensor_adan (optimizer.py:125-235, :296-362), i.e. the
 * slot the reference reserves for the external `fused_adan` extension (optimizer.py:365-395).
 * Per tensor: p, g, exp_avg, exp_avg_sq, exp_avg_diff, neg_pre_grad (all n floats).  `first_step` != 0 makes
 * neg_pre_grad := -g before the update (optimizer.py:190-192).  Scalars are those computed at optimizer.py:171-173,
 * :208-226.  lr is read from DEVICE memory (lr_dev[0]) so a captured hipGraph replays with a changing schedule.
 * ------------------------------------------------------------------------------------------------------------------ */
#define BNERV_ADAN_MAX_TENSORS 48
typedef struct {
    float* p[BNERV_ADAN_MAX_TENSORS];
    const float* g[BNERV_ADAN_MAX_TENSORS];
    float* exp_avg[BNERV_ADAN_MAX_TENSORS];
    float* exp_avg_sq[BNERV_ADAN_MAX_TENSORS];
    float* exp_avg_diff[BNERV_ADAN_MAX_TENSORS];
    float* neg_pre_grad[BNERV_ADAN_MAX_TENSORS];
    int n[BNERV_ADAN_MAX_TENSORS];
    int n_tensors;
} bnerv_adan_chunk;

typedef struct {
    float beta1, beta2, beta3;
    float eps, weight_decay;
    float clip_global_grad_norm;
    int no_prox;
    /* bias corrections depend on the step count; like lr they are read from device memory:
       sched_dev = {lr, bias_correction1, bias_correction2, sqrt(bias_correction3), first_step(0/1)} */
    const float* sched_dev;
} bnerv_adan_hyper;

int bnerv_adan_multi_tensor(void* stream, const bnerv_adan_chunk* chunk, const bnerv_adan_hyper* h);

/* Table form of the same step (ABI 5): the descriptors of ALL tensors live in DEVICE memory, so one launch serves any number of
 * tensors (the chunk form passes 48 descriptors by value: C1's 186 tensors were four dependent launches per step).  The caller
 * fills the table on the host -- entry i owns blocks [bstart, bstart + bnerv_adan_table_blocks(n)) of the grid, bstart = running
 * sum -- and uploads it once; it stays valid while the tensors keep their addresses.  Same arithmetic, same order. */
typedef struct {
    float* p;
    const float* g;
    float* exp_avg;
    float* exp_avg_sq;
    float* exp_avg_diff;
    float* neg_pre_grad;
    int n;
    int bstart;
} bnerv_adan_entry;
int bnerv_adan_table_blocks(int n);
int bnerv_adan_table(void* stream, const bnerv_adan_entry* table_dev, int n_tensors, int total_blocks, const bnerv_adan_hyper* h);

/* Frame fetch of a step whose clip is resident in device memory (train_nerv_all.py:329 moves one frame host -> device per step; with
 * the clip in HBM the step only needs to know WHICH frame): copies frame k = (int)sel_dev[0] of clip [N][frame_elems] to dst_img
 * and norm[k] (fp64, hnerv_utils.py:47) to dst_norm.  sel_dev is device memory, so a captured step replays with a moving index. */
int bnerv_fetch_frame(void* stream, const float* clip, const double* norms, const float* sel_dev, int n_frames, size_t frame_elems,
                      float* dst_img, double* dst_norm);

/* flat-bucket helpers for the data-parallel gradient exchange (replaces DDP's bucket copy, train_nerv_all.py:254):
 * gather `n_tensors` gradients into one contiguous bucket scaled by `scale`, and scatter it back. */
typedef struct {
    float* t[BNERV_ADAN_MAX_TENSORS * 2];
    int n[BNERV_ADAN_MAX_TENSORS * 2];
    int off[BNERV_ADAN_MAX_TENSORS * 2];
    int n_tensors;
} bnerv_bucket_chunk;
int bnerv_bucket_gather(void* stream, const bnerv_bucket_chunk* c, float* bucket, float scale);
int bnerv_bucket_scatter(void* stream, const bnerv_bucket_chunk* c, const float* bucket, float scale);

#ifdef __cplusplus
}
#endif
#endif /* BNERV_H */
