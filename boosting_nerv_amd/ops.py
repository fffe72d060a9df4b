"""Autograd operators of the decoder path.  Every forward/backward here is a sequence of C-ABI calls into
libbnerv_hip.so on torch's current HIP stream; torch supplies device memory, the stream and the autograd graph only.

Operator <-> reference map (the Python modules in this package call these from the same places the reference calls
ATen):
  positional_encoding      PositionEncoding.forward              model_blocks.py:120-126
  dense_grouped            NeRV_MLP / SFTLayer 1x1 convs         model_blocks.py:66-71, :92-105
  conv2d_ps                CustomConv2d (+PixelShuffle)          lib/quant_ops.py:39-41, model_blocks.py:213-218
  snerv_block              NeRVBlock.forward with TAT            model_blocks.py:34-39 + :83-89
  tat_block                ResBlock_SFT.forward                  model_blocks.py:83-89
  sft_affine               SFTLayer.forward (affine part)        model_blocks.py:101-105
  head_tanh                head_layer + OutImg('tanh')           model_nerv.py:56-57, model_blocks.py:57-63
  loss / psnr / msssim     loss_fn, psnr_fn_single, ms_ssim      hnerv_utils.py:335-403, :410-412
"""
import ctypes as C
import os

import torch
import torch.nn.functional as F

from . import _lib as L


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------------------------------------
# raw kernel wrappers
# ----------------------------------------------------------------------------------------------------------------------
def _conv(x, w, bias, out, *, B, Cin, Cout, H, W, k, in_mode, ep_mode, in_s=1, out_s=1, transposed=0, out2=None,
          aux0=None, aux1=None, aux2=None, scale=None, shift=None, partial=None, defer=False):
    d = L.ConvDesc(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(out), L.ptr(out2), L.ptr(aux0), L.ptr(aux1), L.ptr(aux2),
                   L.ptr(scale), L.ptr(shift), L.ptr(partial), B, Cin, Cout, H, W, k, in_mode, ep_mode, in_s, out_s,
                   transposed, w.shape[0], w.shape[1], L.ctx().handle)
    lib = L.load()
    ws = None
    if ep_mode in (L.EP_DGELU, L.EP_DSIN, L.EP_DGELU_SAVED):
        # per-tile (ds, dt) partial sums: the kernel (hence its tile height) is chosen from shape + alignment -> ask, allocate, reduce
        rows = lib.bnerv_conv_partial_rows(C.byref(d))
        part = torch.empty(rows, B, 2, Cout, dtype=torch.float32, device=x.device)
        d.partial = part.data_ptr()
        L.check(lib.bnerv_conv_igemm(L.stream(), C.byref(d)), "bnerv_conv_igemm")
        st = torch.empty(B, 2, Cout, dtype=torch.float32, device=x.device)
        _reduce_slabs(part, rows, B * 2 * Cout, st, defer=defer)
        return st
    if ep_mode == L.EP_PLAIN and partial is None:
        nbytes = lib.bnerv_conv_splitk_ws_bytes(C.byref(d))      # low-resolution, long-K layers want a split-K workspace
        if nbytes:
            ws = _ws(nbytes, x.device)
            d.partial = ws.data_ptr()
    L.check(lib.bnerv_conv_igemm(L.stream(), C.byref(d)), "bnerv_conv_igemm")


def _wgrad(x, g, dw, db, *, B, Cin, Cout, H, W, k, in_mode, g_mode, g_s=1, gaux=None, scale=None, shift=None, defer=False):
    lib = L.load()
    nbytes = lib.bnerv_conv_wgrad_ws_bytes(B, Cin, Cout, H, W, k)
    ws = _ws(nbytes, x.device)
    d = L.WgradDesc(L.ptr(x), L.ptr(g), L.ptr(gaux), L.ptr(scale), L.ptr(shift), L.ptr(dw), L.ptr(db), L.ptr(ws), nbytes,
                    B, Cin, Cout, H, W, k, in_mode, g_mode, g_s, 1 if defer else 0, L.ctx().handle)
    L.check(lib.bnerv_conv_wgrad(L.stream(), C.byref(d)), "bnerv_conv_wgrad")
    if defer:
        L.ctx().keep.append(ws)


def _wgrad_conv_pair(wg, cv):
    """One weight gradient and one data gradient that read the same incoming gradient and do not depend on each other, as ONE launch
    when the library takes the pair (include/bnerv.h bnerv_conv_wgrad_pair: 12-channel 3x3 layers), as the two usual launches
    otherwise.  wg: keyword arguments of _wgrad (x, g, dw, db first), always deferred; cv: keyword arguments of _conv (x, w, bias, out
    first) for an EP_DGELU_SAVED / EP_DSIN / EP_PLAIN epilogue.  Returns what _conv returns (the [B, 2, C] channel sums, or None).
    Side effect: `_pair_dx_deferred` says whether the DATA gradient itself (cv's `out`) is still a queued slab reduction when the call
    returns (the stem pair, csrc/stem.hip: its K-slice slabs are summed by a deferred job) -- every other form writes `out` directly."""
    global _pair_dx_deferred
    _pair_dx_deferred = False
    lib = L.load()
    x, g, dw, db = wg.pop("x"), wg.pop("g"), wg.pop("dw"), wg.pop("db")
    nbytes = lib.bnerv_conv_wgrad_ws_bytes(wg["B"], wg["Cin"], wg["Cout"], wg["H"], wg["W"], wg["k"])
    ws = _ws(nbytes, x.device)
    wd = L.WgradDesc(L.ptr(x), L.ptr(g), L.ptr(wg.get("gaux")), L.ptr(wg.get("scale")), L.ptr(wg.get("shift")), L.ptr(dw), L.ptr(db), L.ptr(ws), nbytes,
                     wg["B"], wg["Cin"], wg["Cout"], wg["H"], wg["W"], wg["k"], wg["in_mode"], wg["g_mode"], wg.get("g_s", 1), 1, L.ctx().handle)
    cx, cw, cb, cout = cv.pop("x"), cv.pop("w"), cv.pop("bias"), cv.pop("out")
    cd = L.ConvDesc(L.ptr(cx), L.ptr(cw), L.ptr(cb), L.ptr(cout), L.ptr(cv.get("out2")), L.ptr(cv.get("aux0")), L.ptr(cv.get("aux1")), L.ptr(cv.get("aux2")),
                    L.ptr(cv.get("scale")), L.ptr(cv.get("shift")), None, cv["B"], cv["Cin"], cv["Cout"], cv["H"], cv["W"], cv["k"], cv["in_mode"], cv["ep_mode"],
                    cv.get("in_s", 1), cv.get("out_s", 1), cv.get("transposed", 0), cw.shape[0], cw.shape[1], L.ctx().handle)
    red = cv["ep_mode"] in (L.EP_DGELU, L.EP_DSIN, L.EP_DGELU_SAVED)
    part = None
    if red:
        rows = lib.bnerv_conv_partial_rows(C.byref(cd))
        part = torch.empty(rows, cv["B"], 2, cv["Cout"], dtype=torch.float32, device=cx.device)
        cd.partial = part.data_ptr()
    skw = None
    if not red and cv["ep_mode"] == L.EP_PLAIN:            # a split-K layer (the stem stage's long-K data gradient): its slab workspace
        nb = lib.bnerv_conv_splitk_ws_bytes(C.byref(cd))
        if nb:
            skw = _ws(nb, cx.device)
            cd.partial = skw.data_ptr()
    rc = lib.bnerv_conv_wgrad_pair(L.stream(), C.byref(cd), C.byref(wd))
    if rc == 1:                                            # not a pair the launch takes: the two usual calls, weight gradient first
        L.check(lib.bnerv_conv_wgrad(L.stream(), C.byref(wd)), "bnerv_conv_wgrad")
        L.check(lib.bnerv_conv_igemm(L.stream(), C.byref(cd)), "bnerv_conv_igemm")
    else:
        L.check(rc, "bnerv_conv_wgrad_pair")
        if skw is not None:
            L.ctx().keep.append(skw)                       # (the stem pair sums its slabs in a deferred reduction: alive until the flush)
            _pair_dx_deferred = True                       # cv's `out` is complete only after the next flush
    L.ctx().keep.append(ws)
    if red:
        st = torch.empty(cv["B"], 2, cv["Cout"], dtype=torch.float32, device=cx.device)
        _reduce_slabs(part, rows, cv["B"] * 2 * cv["Cout"], st, defer=True)
        return st
    return None


# Deferred slab reductions (include/bnerv.h, bnerv_reduce_slabs_deferred): inside one backward the reductions are queued in the
# CONTEXT of the current stream (L.ctx()) and ride on the next lean conv / weight-gradient launch of that stream;
# _flush_deferred() at the end of the backward launches the leftovers, so every tensor a backward returns is complete on the
# stream.  The workspaces of queued jobs are kept alive by the context until then.
def _reduce_slabs(slabs, n_slabs, count, out, defer=False):
    if defer:
        c = L.ctx()
        L.check(L.load().bnerv_reduce_slabs_deferred(c.handle, L.stream(), L.ptr(slabs), n_slabs, count, L.ptr(out)), "bnerv_reduce_slabs_deferred")
        c.keep.append(slabs)
        return
    L.check(L.load().bnerv_reduce_slabs(L.stream(), L.ptr(slabs), n_slabs, count, L.ptr(out)), "bnerv_reduce_slabs")


_lazy_depth = 0       # > 0: inside lazy_flush() -- the per-block flushes are postponed to the first consumer of a deferred result
_lazy_dx_ok = False   # inside lazy_flush(dx_ok=True): a block may also RETURN a data gradient whose slab reduction is still queued
_pair_dx_deferred = False   # set by _wgrad_conv_pair: the data gradient of the last pair is a queued slab reduction (the stem pair)


def _flush_deferred(force=True, block_end=False, dx_deferred=False):
    """Launch whatever slab reductions are still queued on this stream's context.  `block_end`: the call that closes a block's
    backward -- inside lazy_flush() it is skipped, because the queued results (weight / bias gradients, the per-channel TAT sums) have
    no reader until the grouped dense backward or the optimizer, and both flush first: ~5 small dependent launches per step less.
    `dx_deferred`: the block returns a DATA gradient that is itself queued (the stem pair): autograd hands it to whatever produced the
    block's input -- skipped only where the model vouches that this reader is a flushing operator of this package (lazy_flush(dx_ok=True):
    NeRV_Boost, whose first block is fed by the stem MLP; E-NeRV's first up-conv is fed by a stock torch.sin, which would read the
    unreduced buffer)."""
    if block_end and _lazy_depth > 0 and (not dx_deferred or _lazy_dx_ok):
        return
    c = L.ctx()
    L.check(L.load().bnerv_flush_deferred(c.handle, L.stream()), "bnerv_flush_deferred")
    c.keep.clear()


class lazy_flush:
    """with lazy_flush(): backward()  -- postpone the end-of-block flushes of the deferred slab reductions; every consumer inside this
    package (grouped dense / dense GEMM / stand-alone affine backward) flushes on entry, and leaving the context flushes the rest.
    dx_ok: the model's promise that a block's INPUT gradient may stay queued too (see _flush_deferred)."""

    def __init__(self, dx_ok=False):
        self.dx_ok = bool(dx_ok)

    def __enter__(self):
        global _lazy_depth, _lazy_dx_ok
        _lazy_depth += 1
        self._prev = _lazy_dx_ok
        _lazy_dx_ok = self.dx_ok
        return self

    def __exit__(self, *exc):
        global _lazy_depth, _lazy_dx_ok
        _lazy_depth -= 1
        _lazy_dx_ok = self._prev
        if _lazy_depth == 0:
            _flush_deferred()
        return False


def _tiles(H, W):
    return L.load().bnerv_conv_tiles(H, W)


def _bc(t, B, Cc):
    """[B,C,1,1] / [B,C] modulation tensor -> contiguous fp32 [B,C]."""
    t = L.f32c(t).reshape(B, Cc)
    return t


# ----------------------------------------------------------------------------------------------------------------------
# positional encoding (no gradient: positions are data)
# ----------------------------------------------------------------------------------------------------------------------
def positional_encoding(pos, bases, round_to_f32=False):
    """pos [N,1] fp32 or fp64 on the device; bases fp32 [L] (built on the host by the reference expression).
    Returns [N, 2L, 1, 1] fp32 = cat[sin(pos*bases), cos(pos*bases)]; fp64 positions use the fp64 product form -- unless
    round_to_f32: then they are rounded to fp32 first, i.e. positional_encoding(pos.float(), bases) without the conversion launch."""
    L.require_device(pos, "pos")
    N, Lv = pos.shape[0], bases.numel()
    bases = bases.to(device=pos.device, dtype=torch.float32).contiguous()
    out = torch.empty(N, 2 * Lv, 1, 1, dtype=torch.float32, device=pos.device)
    lib = L.load()
    if pos.dtype == torch.float64 and round_to_f32:
        L.check(lib.bnerv_pe_fwd_f32_from_f64(L.stream(), L.ptr(pos.contiguous()), L.ptr(bases), L.ptr(out), N, Lv), "bnerv_pe_fwd_f32_from_f64")
    elif pos.dtype == torch.float64:
        L.check(lib.bnerv_pe_fwd_f64(L.stream(), L.ptr(pos.contiguous()), L.ptr(bases), L.ptr(out), N, Lv), "bnerv_pe_fwd_f64")
    else:
        p = pos.contiguous() if pos.dtype == torch.float32 else pos.float().contiguous()
        L.check(lib.bnerv_pe_fwd_f32(L.stream(), L.ptr(p), L.ptr(bases), L.ptr(out), N, Lv), "bnerv_pe_fwd_f32")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# grouped dense layers
# ----------------------------------------------------------------------------------------------------------------------
_ACT = {"none": L.ACT_NONE, None: L.ACT_NONE, "relu": L.ACT_RELU, "sin": L.ACT_SIN}


class _DenseGrouped(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, n, *tensors):
        xs, ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        lib = L.load()
        B = xs[0].shape[0]
        dev = xs[0].device
        xc, wc, ys, auxs = [], [], [], []
        descs = []
        for i in range(n):
            x = L.f32c(L.require_device(xs[i], "x")).reshape(B, -1)
            w = L.f32c(ws[i]).reshape(ws[i].shape[0], -1)
            b = None if bs[i] is None else L.f32c(bs[i])
            O, I = w.shape
            assert x.shape[1] == I, (x.shape, w.shape)
            y = torch.empty(B, O, dtype=torch.float32, device=dev)
            aux = torch.empty(B, O, dtype=torch.float32, device=dev) if acts[i] == L.ACT_SIN else None
            descs.append(L.DenseFwdDesc(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(aux), I, O, acts[i], 0))
            xc.append(x); wc.append(w); ys.append(y); auxs.append(aux)
        for i0 in range(0, n, L.MAX_DENSE_GROUPS):
            chunk = descs[i0:i0 + L.MAX_DENSE_GROUPS]
            arr = (L.DenseFwdDesc * len(chunk))(*chunk)
            L.check(lib.bnerv_dense_grouped_fwd(L.stream(), arr, len(chunk), B), "bnerv_dense_grouped_fwd")
        ctx.acts, ctx.n, ctx.B = acts, n, B
        ctx.has_b = [b is not None for b in bs]
        ctx.xshapes = [tuple(x.shape) for x in xs]
        ctx.wshapes = [tuple(w.shape) for w in ws]
        ctx.same_x = all(x.data_ptr() == xc[0].data_ptr() and x.shape == xc[0].shape for x in xc)
        ctx.save_for_backward(*xc, *wc, *ys, *auxs)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        if _lazy_depth > 0:
            _flush_deferred()               # the incoming gradients may be deferred slab reductions (the TAT blocks' channel sums)
        n = ctx.n
        sv = ctx.saved_tensors
        xc, wc, ys, auxs = sv[:n], sv[n:2 * n], sv[2 * n:3 * n], sv[3 * n:4 * n]
        need_x = [ctx.needs_input_grad[2 + i] for i in range(n)]
        dxs, dws, dbs = _dense_grouped_bwd(ctx.acts, xc, wc, ys, auxs, dys, need_x, ctx.has_b, ctx.same_x, ctx.B)
        dws = [dw.reshape(sh) for dw, sh in zip(dws, ctx.wshapes)]
        dxs = [None if dx is None else dx.reshape(sh) for dx, sh in zip(dxs, ctx.xshapes)]
        return (None, None) + tuple(dxs) + tuple(dws) + tuple(dbs)


def _dense_grouped_bwd(acts, xc, wc, ys, auxs, dys, need_x, has_b, same_x, B, defer_reduce=False):
    """Backward of n grouped dense layers y_i = act_i(W_i x_i + b_i) (bnerv_dense_grouped_bwd): returns ([dx_i or None], [dW_i [O, I]], [db_i or
    None]).  same_x: every layer read the SAME input tensor -- its gradient is the sum over the layers, returned in the first needed slot.
    defer_reduce: input gradients that need a slab reduction (shared inputs, layers wider than one dx chunk) are QUEUED on the stream context
    (deferred slab reductions, _reduce_slabs(defer=True)) instead of launched: the caller flushes once for several of them."""
    n = len(xc)
    lib = L.load()
    dev = ys[0].device
    descs, keep = [], []
    dws, dbs, dxps, nchunks = [], [], [], []
    I0 = wc[0].shape[1]
    shared = same_x and n > 1 and any(need_x) and all(w.shape[0] <= L.DENSE_DX_CHUNK and w.shape[1] == I0 for w in wc)
    shared_buf = torch.empty(n, B, I0, dtype=torch.float32, device=dev) if shared else None
    for i in range(n):
        x, w = xc[i], wc[i]
        O, I = w.shape
        dy = dys[i]
        if dy is None:
            dy = torch.zeros(B, O, dtype=torch.float32, device=dev)
        dy = L.f32c(dy).reshape(B, O)
        dw = torch.empty(O, I, dtype=torch.float32, device=dev)
        db = torch.empty(O, dtype=torch.float32, device=dev) if has_b[i] else None
        dpre = torch.empty(B, O, dtype=torch.float32, device=dev)
        nch = (O + L.DENSE_DX_CHUNK - 1) // L.DENSE_DX_CHUNK
        if shared:
            dxp = shared_buf[i]
        elif need_x[i]:
            dxp = torch.empty(nch, B, I, dtype=torch.float32, device=dev)
        else:
            dxp = None
        descs.append(L.DenseBwdDesc(L.ptr(x), L.ptr(w), L.ptr(ys[i]), L.ptr(auxs[i]), L.ptr(dy), L.ptr(dpre), L.ptr(dw), L.ptr(db),
                                    L.ptr(dxp), I, O, acts[i], 0))
        keep.append((dy, dpre))
        dws.append(dw); dbs.append(db); dxps.append(dxp); nchunks.append(nch)
    for i0 in range(0, n, L.MAX_DENSE_GROUPS):
        chunk = descs[i0:i0 + L.MAX_DENSE_GROUPS]
        arr = (L.DenseBwdDesc * len(chunk))(*chunk)
        L.check(lib.bnerv_dense_grouped_bwd(L.stream(), arr, len(chunk), B), "bnerv_dense_grouped_bwd")
    dxs = [None] * n
    if shared:
        tot = torch.empty(B, I0, dtype=torch.float32, device=dev)
        _reduce_slabs(shared_buf, n, B * I0, tot, defer=defer_reduce)
        first = next(i for i in range(n) if need_x[i])
        dxs[first] = tot               # the same tensor was passed n times: its whole gradient goes to one slot
    else:
        for i in range(n):
            if dxps[i] is None:
                continue
            if nchunks[i] == 1:
                dxs[i] = dxps[i][0]
            else:
                I = wc[i].shape[1]
                tot = torch.empty(B, I, dtype=torch.float32, device=dev)
                _reduce_slabs(dxps[i], nchunks[i], B * I, tot, defer=defer_reduce)
                dxs[i] = tot
    return dxs, dws, dbs


DENSE_GEMM_MIN_B = 16


class _DenseGemm(torch.autograd.Function):
    """y = act(x W^T + b) for many rows on the MFMA GEMM kernel (bnerv_dense_gemm_fwd / _bwd, csrc/gemm.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x2 = L.f32c(L.require_device(x, "x")).reshape(x.shape[0], -1)
        w2 = L.f32c(w).reshape(w.shape[0], -1)
        b2 = None if b is None else L.f32c(b)
        B, I = x2.shape
        O = w2.shape[0]
        assert w2.shape[1] == I, (x.shape, w.shape)
        y = torch.empty(B, O, dtype=torch.float32, device=x2.device)
        aux = torch.empty(B, O, dtype=torch.float32, device=x2.device) if act == L.ACT_SIN else None
        L.check(L.load().bnerv_dense_gemm_fwd(L.stream(), L.ptr(x2), L.ptr(w2), L.ptr(b2), L.ptr(y), L.ptr(aux), B, I, O, act), "bnerv_dense_gemm_fwd")
        ctx.save_for_backward(x2, w2, y, aux)
        ctx.act, ctx.has_b, ctx.xshape, ctx.wshape = act, b is not None, tuple(x.shape), tuple(w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        if _lazy_depth > 0:
            _flush_deferred()
        x2, w2, y, aux = ctx.saved_tensors
        B, I = x2.shape
        O = w2.shape[0]
        dy = L.f32c(dy).reshape(B, O)
        dw = torch.empty(O, I, dtype=torch.float32, device=x2.device)
        db = torch.empty(O, dtype=torch.float32, device=x2.device) if ctx.has_b else None
        dx = torch.empty(B, I, dtype=torch.float32, device=x2.device) if ctx.needs_input_grad[0] else None
        lib = L.load()
        nbytes = lib.bnerv_dense_gemm_bwd_ws_bytes(B, I, O)
        ws = _ws(nbytes, x2.device) if nbytes else None
        L.check(lib.bnerv_dense_gemm_bwd(L.stream(), L.ptr(x2), L.ptr(w2), L.ptr(y), L.ptr(aux), L.ptr(dy), L.ptr(dx), L.ptr(dw), L.ptr(db),
                                         L.ptr(ws), nbytes, B, I, O, ctx.act), "bnerv_dense_gemm_bwd")
        return (None if dx is None else dx.reshape(ctx.xshape)), dw.reshape(ctx.wshape), db, None


def dense_gemm(x, w, b, act="none"):
    """act(x [rows, I] w[O, I]^T + b) -> [rows, O] on the MFMA GEMM kernel (any row count; meant for >= 16 rows)."""
    return _DenseGemm.apply(x, w, b, _ACT[act])


class _TimeBranch(torch.autograd.Function):
    """PositionEncoding -> stem layer 0 | stem_t (both layers) -> every TAT modulation MLP as ONE launch (include/bnerv.h
    bnerv_time_branch_fwd), then the stem's second layer as the ordinary grouped launch: 2 launches for the reference's chain of
    model_nerv.py:47-51 / model_blocks.py:92-105 instead of 5.  The backward is the unchanged grouped dense backward of the four depths."""

    @staticmethod
    def forward(ctx, pos, bases, n_mlp, *params):
        lib = L.load()
        dev = pos.device
        B, Lv = pos.shape[0], bases.numel()
        p2 = [L.f32c(t).reshape(t.shape[0], -1) if t.dim() > 1 else L.f32c(t) for t in params]
        sw0, sb0, sw1, sb1, tw0, tb0, tw1, tb1 = p2[:8]
        mw = p2[8:]
        SH, SO, TH, TO = sw0.shape[0], sw1.shape[0], tw0.shape[0], tw1.shape[0]
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        pe, sy0, saux0, sy1, saux1 = f(B, 2 * Lv), f(B, SH), f(B, SH), f(B, SO), f(B, SO)
        ty0, taux0, ty1, taux1 = f(B, TH), f(B, TH), f(B, TO), f(B, TO)
        hs = [f(B, TO) for _ in range(n_mlp)]
        outs = [f(B, mw[4 * i + 2].shape[0]) for i in range(n_mlp)]
        d = L.TimeBranchDesc(L.ptr(pos), L.ptr(bases), L.ptr(pe), L.ptr(sw0), L.ptr(sb0), L.ptr(sy0), L.ptr(saux0),
                             L.ptr(tw0), L.ptr(tb0), L.ptr(tw1), L.ptr(tb1), L.ptr(ty0), L.ptr(taux0), L.ptr(ty1), L.ptr(taux1), B, Lv, SH, TH, TO, n_mlp)
        ml = (L.TimeBranchMlp * max(n_mlp, 1))()
        for i in range(n_mlp):
            w1, b1, w2, b2 = mw[4 * i:4 * i + 4]
            ml[i].w1, ml[i].b1, ml[i].w2, ml[i].b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
            ml[i].hs, ml[i].out, ml[i].C = hs[i].data_ptr(), outs[i].data_ptr(), w2.shape[0]
        rc = lib.bnerv_time_branch_fwd(L.stream(), C.byref(d), ml)
        if rc != 0:
            L.check(rc if rc < 0 else -1, "bnerv_time_branch_fwd (shapes were checked by ops.time_branch)")
        g = (L.DenseFwdDesc * 1)(L.DenseFwdDesc(L.ptr(sy0), L.ptr(sw1), L.ptr(sb1), L.ptr(sy1), L.ptr(saux1), SH, SO, L.ACT_SIN, 0))
        L.check(lib.bnerv_dense_grouped_fwd(L.stream(), g, 1, B), "bnerv_dense_grouped_fwd")
        ctx.n_mlp, ctx.B = n_mlp, B
        ctx.pshapes = [tuple(t.shape) for t in params]
        ctx.save_for_backward(pe, sy0, saux0, sy1, saux1, ty0, taux0, ty1, taux1, *hs, *outs, *p2)
        # z_t has no consumer besides the modulation MLPs when they are all evaluated here: autograd would otherwise materialise a zero
        # gradient for it (a fill launch) and the backward would add it (another launch)
        ctx.set_materialize_grads(False)
        return (sy1, ty1, *outs)

    @staticmethod
    def backward(ctx, d_sy1, d_ty1, *d_outs):
        """Five launches instead of the layer-by-layer backward's six (and one reduction launch instead of two): the branches' levels are
        grouped by what is READY, not by which forward launch they came from --
            [every modulation MLP's layer 1 | the stem's layer 1]  ->  [every MLP's layer 0]  ->  ONE flush of both pending input-gradient
            reductions (z_t's over the 32 MLPs, the stem's 68 chunks)  ->  [stem_t layer 1 | stem layer 0]  ->  [stem_t layer 0]."""
        if _lazy_depth > 0:
            _flush_deferred()               # the modulation gradients are deferred slab reductions of the TAT blocks
        n, B = ctx.n_mlp, ctx.B
        sv = ctx.saved_tensors
        pe, sy0, saux0, sy1, saux1, ty0, taux0, ty1, taux1 = sv[:9]
        hs, outs, p2 = sv[9:9 + n], sv[9 + n:9 + 2 * n], sv[9 + 2 * n:]
        sw0, sb0, sw1, sb1, tw0, tb0, tw1, tb1 = p2[:8]
        mw = p2[8:]
        none, relu, sin = L.ACT_NONE, L.ACT_RELU, L.ACT_SIN
        g = [None] * len(p2)
        w1s, w2s = [mw[4 * i] for i in range(n)], [mw[4 * i + 2] for i in range(n)]
        if n + 1 > L.MAX_DENSE_GROUPS or n == 0:
            raise L.BnervError("time branch backward: group count outside the grouped kernel's table")
        # level 1: layer 1 of every modulation MLP + the stem's layer 1 (its input gradient: 68 chunks, reduced below)
        dxa, dwa, dba = _dense_grouped_bwd([none] * n + [sin], list(hs) + [sy0], w2s + [sw1], list(outs) + [sy1], [None] * n + [saux1],
                                           list(d_outs) + [d_sy1], [True] * (n + 1), [True] * (n + 1), False, B, defer_reduce=True)
        dx4, d_sy0 = dxa[:n], dxa[n]
        if any(w.shape[0] > L.DENSE_DX_CHUNK for w in w2s):
            # a modulation MLP wider than one dx chunk (C_i in 65..128): its hidden-layer gradient dx4[i] is itself a queued slab
            # reduction, and level 2 reads it -- reduce now (one more launch, only for such widths; C1 / C4 widths are <= 59)
            _flush_deferred()
        # level 2: layer 0 of every MLP (all read z_t: one shared gradient, summed over the MLPs below)
        dx3, dw3, db3 = _dense_grouped_bwd([relu] * n, [ty1] * n, w1s, hs, [None] * n, dx4, [True] * n, [True] * n, True, B, defer_reduce=True)
        _flush_deferred()                   # both reductions (and whatever else was still queued) in ONE launch
        tot = next(t for t in dx3 if t is not None)
        d_zt = tot if d_ty1 is None else tot + L.f32c(d_ty1).reshape(tot.shape)
        for i in range(n):
            g[8 + 4 * i], g[8 + 4 * i + 1], g[8 + 4 * i + 2], g[8 + 4 * i + 3] = dw3[i], db3[i], dwa[i], dba[i]
        # level 3: stem_t's layer 1 and the stem's layer 0 (the positional encoding has no gradient)
        dxb, dwb, dbb = _dense_grouped_bwd([sin, sin], [ty0, pe], [tw1, sw0], [ty1, sy0], [taux1, saux0], [d_zt, d_sy0], [True, False], [True, True], False, B)
        # level 4: stem_t's layer 0
        _, dwc, dbc = _dense_grouped_bwd([sin], [pe], [tw0], [ty0], [taux0], [dxb[0]], [False], [True], False, B)
        g[0], g[1], g[2], g[3] = dwb[1], dbb[1], dwa[n], dba[n]
        g[4], g[5], g[6], g[7] = dwc[0], dbc[0], dwb[0], dbb[0]
        g = [None if t is None else t.reshape(sh) for t, sh in zip(g, ctx.pshapes)]
        return (None, None, None, *g)


def time_branch(pos, bases, stem, stem_t, mlps):
    """pos [B] fp64, bases fp32 [L]; stem / stem_t = (w0, b0, w1, b1) of the two 2-layer sin MLPs; mlps = [(w1, b1, w2, b2), ...] of the TAT
    modulation branches (relu inside).  Returns (stem_out [B, SO], z_t [B, TO], [out_i [B, C_i]]), or None when the shapes are not the
    kernel's (the caller runs the five grouped launches).  BNERV_TIME_BRANCH=0 switches it off (A/B)."""
    if os.environ.get("BNERV_TIME_BRANCH", "1") == "0" or pos.dtype != torch.float64 or not pos.is_cuda or pos.dim() != 1:
        return None
    B, Lv = pos.shape[0], bases.numel()
    if any(t is None for t in (*stem, *stem_t)) or any(t is None for m in mlps for t in m):
        return None
    fl = lambda t: t.reshape(t.shape[0], -1).shape
    SH, TH, TO = stem[0].shape[0], stem_t[0].shape[0], stem_t[2].shape[0]
    # (the backward groups every MLP's layer 1 WITH the stem's layer 1 in one grouped launch: n + 1 groups)
    if B > 4 or 2 * Lv > 256 or (2 * Lv) % 4 or TH > 64 or TH % 4 or TO > 32 or TO % 4 or len(mlps) + 1 > L.MAX_DENSE_GROUPS:
        return None
    # every layer must have the width the kernel indexes with (an SFTLayer built with factor != 1 has a narrower hidden layer)
    if fl(stem[0])[1] != 2 * Lv or fl(stem_t[0])[1] != 2 * Lv or fl(stem[2])[1] != SH or fl(stem_t[2])[1] != TH:
        return None
    if any(fl(m[0]) != (TO, TO) or fl(m[2])[1] != TO or fl(m[2])[0] > 128 for m in mlps):
        return None
    if any(t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16 for t in (stem_t[0], stem_t[2], *[m[0] for m in mlps], *[m[2] for m in mlps])):
        return None
    flat = [t for m in mlps for t in m]
    out = _TimeBranch.apply(pos.contiguous(), bases.to(device=pos.device, dtype=torch.float32).contiguous(), len(mlps), *stem, *stem_t, *flat)
    return out[0], out[1], list(out[2:])


def dense_grouped(xs, ws, bs, acts):
    """Evaluate n independent dense layers y_i = act_i(W_i x_i + b_i).
    xs[i]: [B, I_i(,1,1)], ws[i]: [O_i, I_i(,1,1)], bs[i]: [O_i] or None, acts[i] in {'none','relu','sin'}.
    Returns a list of [B, O_i] tensors.  B = 1 per GPU (the reference's batch) is a set of GEMVs and goes to the grouped kernel in
    ONE launch; layers applied to 16 or more rows (E-NeRV's 144-token MLPs) are GEMMs and go to the MFMA GEMM kernel."""
    n = len(xs)
    if xs[0].shape[0] >= DENSE_GEMM_MIN_B:
        return [dense_gemm(x, w, b, act if act is not None else "none") for x, w, b, act in zip(xs, ws, bs, acts)]
    a = tuple(_ACT[x] for x in acts)
    return list(_DenseGrouped.apply(a, n, *xs, *ws, *bs))


# ----------------------------------------------------------------------------------------------------------------------
# plain conv (+ bias, + optional pixel shuffle)
# ----------------------------------------------------------------------------------------------------------------------
class _Conv2dPS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, s):
        x = L.f32c(L.require_device(x, "x")); w = L.f32c(w); b = None if b is None else L.f32c(b)
        B, Cin, H, W = x.shape
        Cout, k = w.shape[0], w.shape[-1]
        out = torch.empty(B, Cout // (s * s), H * s, W * s, dtype=torch.float32, device=x.device)
        _conv(x, w, b, out, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS, out_s=s)
        ctx.save_for_backward(x, w)
        ctx.s, ctx.has_b = s, b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s = ctx.s
        g = L.f32c(g)
        B, Cin, H, W = x.shape
        Cout, k = w.shape[0], w.shape[-1]
        dw = torch.empty_like(w)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device) if ctx.has_b else None
        dx = None
        dxq = False
        if ctx.needs_input_grad[0] and k == 3:                 # (dW | dx): one launch where the library pairs them
            dx = torch.empty_like(x)
            _wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=s),
                             dict(x=g, w=w, bias=None, out=dx, B=B, Cin=Cout, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=s, transposed=1))
            dxq = _pair_dx_deferred
        else:
            _wgrad(x, g, dw, db, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=s, defer=True)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _conv(g, w, None, dx, B=B, Cin=Cout, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=s, transposed=1)
        _flush_deferred(block_end=True, dx_deferred=dxq)
        return dx, dw, db, None


def conv2d_ps(x, w, b, shuffle=1):
    """F.conv2d(x, w, b, stride 1, padding (k-1)//2) followed by PixelShuffle(shuffle); k in {1,3}."""
    return _Conv2dPS.apply(x, w, b, int(shuffle))


# ----------------------------------------------------------------------------------------------------------------------
# TAT residual block and the fused SNeRV block
# ----------------------------------------------------------------------------------------------------------------------
def _tat_forward(y0, s0, t0, s1, t1, w0, b0, w1, b1, train=True):
    B, Cc, H, W = y0.shape
    # conv0's epilogue stores h = gelu(v) and gp = gelu'(v) instead of v: conv1, its weight gradient and the dGELU epilogue of
    # the backward then run without erf/exp (v itself has no other consumer)
    out = torch.empty_like(y0)
    hs = torch.empty_like(y0) if train else None
    gp = torch.empty_like(y0) if train else None          # decode / eval (no_grad): neither h nor gelu' is read again, so they are not written
    h = hs if train else torch.empty_like(y0)
    _conv(y0, w0, b0, h, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=s0, shift=t0, out2=gp)
    _conv(h, w1, b1, out, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=s1, shift=t1, aux0=y0)
    return h, gp, out


def _tat_backward(dout, y0, c0, h, gp, s0, t0, s1, t1, w0, w1):
    """Returns (dy0_or_du, ds0, dt0, ds1, dt1, dw0, db0, dw1, db1).  With c0 given the first result is
    d/d(pre-sin) = (dout + dA0*(1+s0)) * c0, otherwise d/dy0."""
    B, Cc, H, W = y0.shape
    dev = y0.device
    dw1 = torch.empty_like(w1); db1 = torch.empty(Cc, dtype=torch.float32, device=dev)
    # every slab reduction below is deferred: it rides on the next launch of this chain; the CALLER flushes the leftovers
    # (dW1 | d conv1) and (dW0 | d conv0): each pair reads one incoming gradient and is ONE launch where the library pairs them
    dv = torch.empty_like(y0)
    st1 = _wgrad_conv_pair(dict(x=h, g=dout, dw=dw1, db=db1, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=s1, shift=t1),
                           dict(x=dout, w=w1, bias=None, out=dv, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED,
                                transposed=1, aux0=gp, aux1=h, scale=s1))
    dw0 = torch.empty_like(w0); db0 = torch.empty(Cc, dtype=torch.float32, device=dev)
    du = torch.empty_like(y0)
    st0 = _wgrad_conv_pair(dict(x=y0, g=dv, dw=dw0, db=db0, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=s0, shift=t0),
                           dict(x=dv, w=w0, bias=None, out=du, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN,
                                transposed=1, aux0=y0, aux1=dout, aux2=c0, scale=s0))
    return du, st0[:, 0], st0[:, 1], st1[:, 0], st1[:, 1], dw0, db0, dw1, db1


class _TATBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, s0, t0, s1, t1, w0, b0, w1, b1):
        x0 = L.f32c(L.require_device(x0, "x0"))
        B, Cc = x0.shape[:2]
        s0, t0, s1, t1 = (_bc(t, B, Cc) for t in (s0, t0, s1, t1))
        w0, b0, w1, b1 = (L.f32c(t) for t in (w0, b0, w1, b1))
        train = any(ctx.needs_input_grad)                   # False under torch.no_grad(): the decode path keeps nothing for backward
        h, gp, out = _tat_forward(x0, s0, t0, s1, t1, w0, b0, w1, b1, train)
        if train:
            ctx.save_for_backward(x0, h, gp, s0, t0, s1, t1, w0, w1)
        ctx.mshape = (B, Cc, 1, 1)
        return out

    @staticmethod
    def backward(ctx, dout):
        x0, h, gp, s0, t0, s1, t1, w0, w1 = ctx.saved_tensors
        dx0, ds0, dt0, ds1, dt1, dw0, db0, dw1, db1 = _tat_backward(L.f32c(dout), x0, None, h, gp, s0, t0, s1, t1, w0, w1)
        _flush_deferred(block_end=True)
        m = ctx.mshape
        return dx0, ds0.reshape(m), dt0.reshape(m), ds1.reshape(m), dt1.reshape(m), dw0, db0, dw1, db1


def tat_block(x0, scale0, shift0, scale1, shift1, w0, b0, w1, b1):
    """x0 + conv1(sft1(gelu(conv0(sft0(x0)))))  with sft_i(a) = a*(scale_i+1)+shift_i  (ResBlock_SFT.forward)."""
    return _TATBlock.apply(x0, scale0, shift0, scale1, shift1, w0, b0, w1, b1)


class _SNeRVBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wu, bu, s0, t0, s1, t1, w0, b0, w1, b1, stride):
        x = L.f32c(L.require_device(x, "x"))
        wu, w0, b0, w1, b1 = (L.f32c(t) for t in (wu, w0, b0, w1, b1))
        bu = None if bu is None else L.f32c(bu)
        B, Cin, H, W = x.shape
        Ct, k = wu.shape[0], wu.shape[-1]
        s = stride
        Cc = Ct // (s * s)
        s0, t0, s1, t1 = (_bc(t, B, Cc) for t in (s0, t0, s1, t1))
        train = any(ctx.needs_input_grad)                   # False under torch.no_grad(): decode path, nothing saved
        y0 = torch.empty(B, Cc, H * s, W * s, dtype=torch.float32, device=x.device)
        c0 = torch.empty_like(y0) if train else None
        _conv(x, wu, bu, y0, B=B, Cin=Cin, Cout=Ct, H=H, W=W, k=k, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out_s=s, out2=c0)
        h, gp, out = _tat_forward(y0, s0, t0, s1, t1, w0, b0, w1, b1, train)
        if train:
            ctx.save_for_backward(x, y0, c0, h, gp, s0, t0, s1, t1, wu, w0, w1)
        ctx.s, ctx.has_bu, ctx.mshape = s, bu is not None, (B, Cc, 1, 1)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y0, c0, h, gp, s0, t0, s1, t1, wu, w0, w1 = ctx.saved_tensors
        s = ctx.s
        du, ds0, dt0, ds1, dt1, dw0, db0, dw1, db1 = _tat_backward(L.f32c(dout), y0, c0, h, gp, s0, t0, s1, t1, w0, w1)
        B, Cin, H, W = x.shape
        Ct, k = wu.shape[0], wu.shape[-1]
        dwu = torch.empty_like(wu)
        dbu = torch.empty(Ct, dtype=torch.float32, device=x.device) if ctx.has_bu else None
        dx = None
        dxq = False
        if ctx.needs_input_grad[0] and k == 3:      # (dW_block | d block conv): one launch where the library pairs them
            dx = torch.empty_like(x)
            _wgrad_conv_pair(dict(x=x, g=du, dw=dwu, db=dbu, B=B, Cin=Cin, Cout=Ct, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=s),
                             dict(x=du, w=wu, bias=None, out=dx, B=B, Cin=Ct, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=s, transposed=1))
            dxq = _pair_dx_deferred
        else:
            _wgrad(x, du, dwu, dbu, B=B, Cin=Cin, Cout=Ct, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=s, defer=True)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _conv(du, wu, None, dx, B=B, Cin=Ct, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=s, transposed=1)
        _flush_deferred(block_end=True, dx_deferred=dxq)
        m = ctx.mshape
        return dx, dwu, dbu, ds0.reshape(m), dt0.reshape(m), ds1.reshape(m), dt1.reshape(m), dw0, db0, dw1, db1, None


def snerv_block(x, w_up, b_up, scale0, shift0, scale1, shift1, w0, b0, w1, b1, stride):
    """tat_block(sin(PixelShuffle_stride(conv(x, w_up, b_up))))  -- NeRVBlock.forward with act='sin', norm='none'."""
    return _SNeRVBlock.apply(x, w_up, b_up, scale0, shift0, scale1, shift1, w0, b0, w1, b1, int(stride))


# ----------------------------------------------------------------------------------------------------------------------
# stand-alone TAT affine
# ----------------------------------------------------------------------------------------------------------------------
class _SFTAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        x = L.f32c(L.require_device(x, "x"))
        B, Cc = x.shape[:2]
        HW = x[0, 0].numel()
        sc, sh = _bc(scale, B, Cc), _bc(shift, B, Cc)
        y = torch.empty_like(x)
        L.check(L.load().bnerv_sft_affine_fwd(L.stream(), L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(y), B, Cc, HW), "bnerv_sft_affine_fwd")
        ctx.save_for_backward(x, sc)
        ctx.mshape = tuple(scale.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        if _lazy_depth > 0:
            _flush_deferred()
        x, sc = ctx.saved_tensors
        g = L.f32c(g)
        B, Cc = x.shape[:2]
        HW = x[0, 0].numel()
        dx = torch.empty_like(x)
        part = torch.empty(L.SFT_CHUNKS, 2, B * Cc, dtype=torch.float32, device=x.device)
        L.check(L.load().bnerv_sft_affine_bwd(L.stream(), L.ptr(x), L.ptr(sc), L.ptr(g), L.ptr(dx), L.ptr(part), B, Cc, HW), "bnerv_sft_affine_bwd")
        tot = torch.empty(2, B * Cc, dtype=torch.float32, device=x.device)
        _reduce_slabs(part, L.SFT_CHUNKS, 2 * B * Cc, tot)
        return dx, tot[0].reshape(ctx.mshape), tot[1].reshape(ctx.mshape)


def sft_affine(x, scale, shift):
    return _SFTAffine.apply(x, scale, shift)


# ----------------------------------------------------------------------------------------------------------------------
# head conv + OutImg('tanh')
# ----------------------------------------------------------------------------------------------------------------------
class _HeadTanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = L.f32c(L.require_device(x, "x")); w = L.f32c(w); b = None if b is None else L.f32c(b)
        B, Cin, H, W = x.shape
        Cout, k = w.shape[0], w.shape[-1]
        img = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
        _conv(x, w, b, img, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_TANH)
        ctx.save_for_backward(x, w, img)
        ctx.has_b = b is not None
        return img

    @staticmethod
    def backward(ctx, g):
        x, w, img = ctx.saved_tensors
        g = L.f32c(g)
        B, Cin, H, W = x.shape
        Cout, k = w.shape[0], w.shape[-1]
        dw = torch.empty_like(w)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device) if ctx.has_b else None
        dx = None
        if k == 3 and Cout <= 4 and Cin >= 16 and os.environ.get("BNERV_HEAD_SWAP", "1") != "0":
            # HNeRV_Boost's 3x3 head (38 -> 3, model_hnerv.py:214).  As dW[co] = sum_p gt[co][p] x[.][p + tap] the MFMA tile has 3 of its 16
            # rows in use (272 us at 1080p).  The same sums with the ROLES SWAPPED -- x as the "gradient" (M = 38 input channels), gt as the
            # "input" (N = 3 couts x 9 taps) -- are S[ci][co][t'] = sum_q x[ci][q] gt[co][q + t' - 1] = dW[co][ci][2 - t'] (zero padding
            # excludes the same terms on both sides): one weight-gradient launch on the well-filled shape, then a transpose + tap flip of
            # 1026 numbers.  gt = the tanh-gradient as a tensor (bnerv_tanh_grad), whose channel sums are the bias gradient.
            lib = L.load()
            HW = H * W
            gt = torch.empty_like(g)
            nblk = lib.bnerv_tanh_grad_blocks(HW)
            part = torch.empty(B * nblk, Cout, dtype=torch.float32, device=x.device)
            L.check(lib.bnerv_tanh_grad(L.stream(), L.ptr(g), L.ptr(img), L.ptr(gt), L.ptr(part), B, Cout, HW), "bnerv_tanh_grad")
            if db is not None:
                _reduce_slabs(part, B * nblk, Cout, db, defer=True)
            sw = torch.empty(Cin, Cout, k, k, dtype=torch.float32, device=x.device)
            sb = torch.empty(Cin, dtype=torch.float32, device=x.device)           # (the swapped call's "bias gradient": channel sums of x, unused)
            _wgrad(gt, x, sw, sb, B=B, Cin=Cout, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
            dw = sw.permute(1, 0, 2, 3).flip(2, 3).contiguous()
            if ctx.needs_input_grad[0]:           # (gt is a tensor now: the data gradient reads 3 planes instead of 6 -- csrc/head3.hip, 78 against 91 us at 1080p)
                dx = torch.empty_like(x)
                _conv(gt, w, None, dx, B=B, Cin=Cout, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_PLAIN, ep_mode=L.EP_PLAIN, transposed=1)
        elif ctx.needs_input_grad[0]:
            # (dW | dx) of the head: one streaming pass where the library pairs them (1x1, tanh-grad prologue), the two launches otherwise
            dx = torch.empty_like(x)
            _wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_TANHGRAD, gaux=img),
                             dict(x=g, w=w, bias=None, out=dx, B=B, Cin=Cout, Cout=Cin, H=H, W=W, k=k, in_mode=L.IN_TANHGRAD, ep_mode=L.EP_PLAIN, transposed=1, aux0=img))
        else:
            _wgrad(x, g, dw, db, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, in_mode=L.IN_PLAIN, g_mode=L.IN_TANHGRAD, gaux=img, defer=True)
        _flush_deferred(block_end=True)
        return dx, dw, db


def head_tanh(x, w, b):
    """tanh(conv(x, w, b)) * 0.5 + 0.5   (head_layer followed by OutImg(..., 'tanh'))."""
    return _HeadTanh.apply(x, w, b)


# ----------------------------------------------------------------------------------------------------------------------
# CEM compression path: fused quantise + rate term over all weight / bias tensors (row N2)
# ----------------------------------------------------------------------------------------------------------------------
class _CemScaleRate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, training, *tensors):
        ws, scales, noises = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        lib = L.load()
        dev = ws[0].device
        wc = [L.f32c(L.require_device(w, "w")) for w in ws]
        sc = [L.f32c(s) for s in scales]
        nz = [None if z is None else L.f32c(z) for z in noises]
        deq = [torch.empty_like(w) for w in wc]
        stats = torch.empty(n, 4, dtype=torch.float32, device=dev)
        for i0 in range(0, n, L.CEM_MAX_TENSORS):
            ck = L.CemChunk()
            m = min(L.CEM_MAX_TENSORS, n - i0)
            for j in range(m):
                i = i0 + j
                assert sc[i].numel() == 1, "the fused CEM kernel handles per-tensor scales"
                ck.it[j].w, ck.it[j].scale, ck.it[j].dequant, ck.it[j].n = wc[i].data_ptr(), sc[i].data_ptr(), deq[i].data_ptr(), wc[i].numel()
                ck.it[j].noise = nz[i].data_ptr() if nz[i] is not None else None
            ck.n_items, ck.training, ck.first = m, int(training), i0
            nb = lib.bnerv_cem_ws_bytes(m, max(wc[i0 + j].numel() for j in range(m)))
            ws_ = _ws(nb, dev)                                 # chunk partial sums (stream-ordered allocation: reused by the next group)
            L.check(lib.bnerv_cem_scale_fwd(L.stream(), C.byref(ck), L.ptr(stats), L.ptr(ws_), nb), "bnerv_cem_scale_fwd")
        ctx.n, ctx.training = n, training
        ctx.save_for_backward(stats, *wc, *sc, *[z for z in nz if z is not None])
        ctx.has_noise = [z is not None for z in nz]
        ctx.wshapes = [tuple(w.shape) for w in ws]
        ctx.mark_non_differentiable(stats)
        bits = stats[:, 0].clone()
        return (bits, stats, *deq)

    @staticmethod
    def backward(ctx, d_bits, _d_stats, *d_deq):
        n = ctx.n
        sv = ctx.saved_tensors
        stats, wc, sc = sv[0], sv[1:1 + n], sv[1 + n:1 + 2 * n]
        rest = list(sv[1 + 2 * n:])
        nz = [rest.pop(0) if h else None for h in ctx.has_noise]
        lib = L.load()
        dev = stats.device
        d_bits = None if d_bits is None else L.f32c(d_bits)
        dds = [None if g is None else L.f32c(g) for g in d_deq]
        dws = [torch.empty_like(w) for w in wc]
        dscale = torch.empty(n, dtype=torch.float32, device=dev)
        for i0 in range(0, n, L.CEM_MAX_TENSORS):
            ck = L.CemChunkBwd()
            m = min(L.CEM_MAX_TENSORS, n - i0)
            for j in range(m):
                i = i0 + j
                ck.it[j].w, ck.it[j].scale, ck.it[j].dw, ck.it[j].n = wc[i].data_ptr(), sc[i].data_ptr(), dws[i].data_ptr(), wc[i].numel()
                ck.it[j].noise = nz[i].data_ptr() if nz[i] is not None else None
                ck.it[j].d_dequant = dds[i].data_ptr() if dds[i] is not None else None
            ck.n_items, ck.training, ck.first = m, int(ctx.training), i0
            nb = lib.bnerv_cem_ws_bytes(m, max(wc[i0 + j].numel() for j in range(m)))
            ws_ = _ws(nb, dev)
            L.check(lib.bnerv_cem_scale_bwd(L.stream(), C.byref(ck), L.ptr(stats), L.ptr(d_bits), L.ptr(dscale), L.ptr(ws_), nb), "bnerv_cem_scale_bwd")
        gs = [dscale[i:i + 1].reshape(sc[i].shape) for i in range(n)]
        return (None, None, *[dw.reshape(sh) for dw, sh in zip(dws, ctx.wshapes)], *gs, *([None] * n))


def cem_scale_rate(ws, scales, noises, training):
    """Scale_T quantise + Gaussian rate for a list of tensors in one fused pass (include/bnerv.h, bnerv_cem_scale_fwd).
    Returns (bits [n], stats [n,4] = {bits, mean, std, numel}, [dequant_i]).  noises: list of U(-.5,.5) tensors (training) or Nones."""
    n = len(ws)
    out = _CemScaleRate.apply(n, bool(training), *ws, *scales, *noises)
    return out[0], out[1], list(out[2:])


# ----------------------------------------------------------------------------------------------------------------------
# depthwise conv of the ConvNeXt encoder block (first kernels of row N3)
# ----------------------------------------------------------------------------------------------------------------------
class _DWConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = L.f32c(L.require_device(x, "x")); w = L.f32c(w); b = None if b is None else L.f32c(b)
        B, Cc, H, W = x.shape
        K = w.shape[-1]
        assert w.shape == (Cc, 1, K, K), w.shape
        y = torch.empty_like(x)
        L.check(L.load().bnerv_dwconv_fwd(L.stream(), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, Cc, H, W, K, 0), "bnerv_dwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = L.f32c(g)
        B, Cc, H, W = x.shape
        K = w.shape[-1]
        lib = L.load()
        nbytes = lib.bnerv_dwconv_wgrad_ws_bytes(B, Cc, H, W, K)
        ws = _ws(nbytes, x.device)
        dwb = torch.empty(Cc, K * K + 1, dtype=torch.float32, device=x.device)
        L.check(lib.bnerv_dwconv_wgrad(L.stream(), L.ptr(x), L.ptr(g), L.ptr(dwb), L.ptr(ws), nbytes, B, Cc, H, W, K, None), "bnerv_dwconv_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(lib.bnerv_dwconv_fwd(L.stream(), L.ptr(g), L.ptr(w), None, L.ptr(dx), B, Cc, H, W, K, 1), "bnerv_dwconv_fwd(flip)")
        dw = dwb[:, :K * K].reshape(Cc, 1, K, K)
        db = dwb[:, K * K] if ctx.has_b else None
        return dx, dw, db


def dwconv(x, w, b):
    """F.conv2d(x, w, b, padding=K//2, groups=C) for w [C,1,K,K], K odd <= 7 (ConvNeXt Block.dwconv, model_blocks.py:231)."""
    return _DWConv.apply(x, w, b)


LN_CF_MAX_C = 64


class _LayerNormCF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = L.f32c(L.require_device(x, "x")); w = L.f32c(w); b = L.f32c(b)
        B, Cc = x.shape[:2]
        HW = x[0, 0].numel()
        y = torch.empty_like(x)
        L.check(L.load().bnerv_lncf_fwd(L.stream(), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, Cc, HW, float(eps)), "bnerv_lncf_fwd")
        ctx.save_for_backward(x, w)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = L.f32c(g)
        B, Cc = x.shape[:2]
        HW = x[0, 0].numel()
        lib = L.load()
        nbytes = lib.bnerv_lncf_bwd_ws_bytes(B, Cc, HW)
        ws = _ws(nbytes, x.device)
        dx = torch.empty_like(x)
        dwb = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        L.check(lib.bnerv_lncf_bwd(L.stream(), L.ptr(x), L.ptr(w), L.ptr(g), L.ptr(dx), L.ptr(dwb), L.ptr(ws), nbytes, B, Cc, HW, ctx.eps), "bnerv_lncf_bwd")
        return dx, dwb[0], dwb[1], None


def layernorm_cf(x, w, b, eps):
    """LayerNorm over dim 1 of an NCHW tensor, C <= 64 (model_blocks.py:250-270 `LayerNorm`, data_format channels_first)."""
    return _LayerNormCF.apply(x, w, b, eps)


CNX_MLP_DIMS = (16, 32, 48, 64)


class _CnxMlp(torch.autograd.Function):
    """inp + gamma * pwconv2(gelu(pwconv1(x))) on NCHW tensors (ConvNeXt Block tail, model_blocks.py:245-258): one fused forward
    kernel; backward = one fused kernel (dx and the two pixel-contraction operands) + two k = 1 weight-gradient launches."""

    @staticmethod
    def forward(ctx, x, inp, w1, b1, w2, b2, gamma):
        x = L.f32c(L.require_device(x, "x")); inp = L.f32c(inp)
        w1, b1, w2, b2 = (L.f32c(t) for t in (w1, b1, w2, b2))
        gm = None if gamma is None else L.f32c(gamma)
        B, Cc, H, W = x.shape
        train = any(ctx.needs_input_grad)
        out = torch.empty_like(x)
        h1 = torch.empty(B, 4 * Cc, H, W, dtype=torch.float32, device=x.device) if train else None
        L.check(L.load().bnerv_cnx_mlp_fwd(L.stream(), L.ptr(x), L.ptr(inp), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(gm), L.ptr(out),
                                           L.ptr(h1), B, Cc, H * W), "bnerv_cnx_mlp_fwd")
        if train:
            ctx.save_for_backward(x, h1, w1, w2, b2, gm)
        ctx.has_gamma = gamma is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h1, w1, w2, b2, gm = ctx.saved_tensors
        dout = L.f32c(dout)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        gbuf, dhbuf = torch.empty_like(h1), torch.empty_like(h1)
        L.check(L.load().bnerv_cnx_mlp_bwd(L.stream(), L.ptr(h1), L.ptr(dout), L.ptr(w1), L.ptr(w2), L.ptr(gm), L.ptr(dx), L.ptr(gbuf), L.ptr(dhbuf),
                                           B, Cc, H * W), "bnerv_cnx_mlp_bwd")
        dw1 = torch.empty(4 * Cc, Cc, 1, 1, dtype=torch.float32, device=dev); db1 = torch.empty(4 * Cc, dtype=torch.float32, device=dev)
        _wgrad(x, dhbuf, dw1, db1, B=B, Cin=Cc, Cout=4 * Cc, H=H, W=W, k=1, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
        S = torch.empty(Cc, 4 * Cc, 1, 1, dtype=torch.float32, device=dev); t = torch.empty(Cc, dtype=torch.float32, device=dev)
        _wgrad(gbuf, dout, S, t, B=B, Cin=4 * Cc, Cout=Cc, H=H, W=W, k=1, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
        S = S.reshape(Cc, 4 * Cc)
        if ctx.has_gamma:      # [C x 4C] bookkeeping in one launch: dw2 = gamma S, db2 = gamma t, dgamma = rowsum(w2 * S) + b2 t
            S = S.contiguous()
            dw2, db2, dgamma = torch.empty_like(S), torch.empty_like(t), torch.empty_like(t)
            L.check(L.load().bnerv_cnx_param_grads(L.stream(), L.ptr(S), L.ptr(t), L.ptr(w2), L.ptr(b2), L.ptr(gm), L.ptr(dw2), L.ptr(db2),
                                                   L.ptr(dgamma), Cc), "bnerv_cnx_param_grads")
        else:
            dgamma, dw2, db2 = None, S, t
        return dx, dout, dw1.reshape(4 * Cc, Cc), db1, dw2, db2, dgamma


def cnx_mlp(x, inp, w1, b1, w2, b2, gamma):
    """inp + gamma * (w2 gelu(w1 x + b1) + b2) per pixel; x, inp [B, C, H, W], w1 [4C, C], w2 [C, 4C]; C in CNX_MLP_DIMS."""
    return _CnxMlp.apply(x, inp, w1, b1, w2, b2, gamma)


# ----------------------------------------------------------------------------------------------------------------------
# loss / metrics
# ----------------------------------------------------------------------------------------------------------------------
LOSS_COEFFS = {            # (c_l1, c_l2, c_ms, c_fft) per hnerv_utils.py:338-385
    "L1": (1.0, 0.0, 0.0, 0.0),
    "L2": (0.0, 1.0, 0.0, 0.0),
    "L1_freq": (60.0, 0.0, 0.0, 1.0),
    "Fusion10": (0.7, 0.0, 0.3, 0.0),
    "Fusion11": (0.9, 0.0, 0.1, 0.0),
    "Fusion12": (0.8, 0.0, 0.2, 0.0),
    "Fusion10_freq": (60 * 0.7, 0.0, 60 * 0.3, 1.0),
}

_fft_ready = set()


def prepare_loss(H, W):
    """Build the FFT twiddle tables for HxW frames (must happen outside hipGraph capture)."""
    if (H, W) not in _fft_ready:
        L.check(L.load().bnerv_fft_prepare(H, W), "bnerv_fft_prepare")
        _fft_ready.add((H, W))


def _loss_launch(pred, target, coeffs, need_grad):
    """One fused call: loss value, per-sample stats and (optionally) d loss / d pred."""
    pred = L.f32c(L.require_device(pred, "pred")); target = L.f32c(target.detach())
    B, Cc, H, W = pred.shape
    c1, c2, cm, cf = coeffs
    lib = L.load()
    if cf:
        prepare_loss(H, W)
    nbytes = lib.bnerv_loss_ws_bytes(B, Cc, H, W, int(cm != 0), int(cf != 0))
    ws = _ws(nbytes, pred.device)
    grad = torch.empty_like(pred) if need_grad else None
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    stats = torch.empty(B, L.LOSS_STATS, dtype=torch.float32, device=pred.device)
    d = L.LossDesc(L.ptr(pred), L.ptr(target), L.ptr(grad), L.ptr(loss), L.ptr(stats), L.ptr(ws), nbytes, B, Cc, H, W, c1, c2, cm, cf)
    L.check(lib.bnerv_loss_fwd_bwd(L.stream(), C.byref(d)), "bnerv_loss_fwd_bwd")
    return loss, stats, grad


def loss_value_grad_stats(pred, target, loss_type="Fusion10_freq"):
    """(loss [scalar], stats [B,5], d loss/d pred) without an autograd node: the train step seeds pred.backward(grad) with
    the gradient directly (no ones-fill, no grad * 1 pass over the frame) and reads the per-sample PSNR from stats[:, 4]."""
    if loss_type not in LOSS_COEFFS:
        raise NotImplementedError(f"loss type {loss_type!r} is not on the HIP path; supported: {sorted(LOSS_COEFFS)}")
    loss, stats, grad = _loss_launch(pred.detach(), target, LOSS_COEFFS[loss_type], True)
    return loss.reshape(()), stats, grad


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, coeffs):
        loss, stats, grad = _loss_launch(pred, target, coeffs, ctx.needs_input_grad[0])
        ctx.grad = grad
        ctx.mark_non_differentiable(stats)
        return loss.reshape(()), stats

    @staticmethod
    def backward(ctx, gl, _gs):
        g = ctx.grad
        ctx.grad = None
        return (g * gl if g is not None else None), None, None


def loss_with_stats(pred, target, loss_type="Fusion10_freq"):
    """Returns (batch-mean loss [scalar tensor], stats [B,5] = {loss_b, sum|d|, sum d^2, ms_ssim_b, psnr_b})."""
    if loss_type not in LOSS_COEFFS:
        raise NotImplementedError(f"loss type {loss_type!r} is not on the HIP path; supported: {sorted(LOSS_COEFFS)}")
    return _Loss.apply(pred, target, LOSS_COEFFS[loss_type])


def psnr(output, gt):
    """-10*log10(mean_{CHW}(output-gt)^2 + 1e-9) per sample, on the device (no host sync)."""
    output = L.f32c(L.require_device(output.detach(), "output")); gt = L.f32c(gt.detach())
    B, Cc, H, W = output.shape
    lib = L.load()
    nbytes = lib.bnerv_psnr_ws_bytes(B, Cc, H, W)
    ws = _ws(nbytes, output.device)
    out = torch.empty(B, dtype=torch.float32, device=output.device)
    L.check(lib.bnerv_psnr(L.stream(), L.ptr(output), L.ptr(gt), L.ptr(out), L.ptr(ws), nbytes, B, Cc, H, W), "bnerv_psnr")
    return out


def msssim(x, y):
    """ms_ssim(x, y, data_range=1, size_average=False) per sample, on the device."""
    x = L.f32c(L.require_device(x.detach(), "x")); y = L.f32c(y.detach())
    B, Cc, H, W = x.shape
    lib = L.load()
    nbytes = lib.bnerv_loss_ws_bytes(B, Cc, H, W, 1, 0)
    ws = _ws(nbytes, x.device)
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    L.check(lib.bnerv_msssim(L.stream(), L.ptr(x), L.ptr(y), L.ptr(out), L.ptr(ws), nbytes, B, Cc, H, W), "bnerv_msssim")
    return out
