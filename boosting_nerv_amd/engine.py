"""The train step of train_nerv_all.py:328-350 as one object: forward -> loss -> backward -> (gradient all-reduce) -> Adan,
with per-step PSNR accumulated on the device.  After a few eager steps the fixed-shape step is captured into hipGraphs and
replayed (≈100 short kernels per step: launch latency, not arithmetic, is what a Python-driven loop would pay for).

  world == 1 :  one graph   [fwd, loss, bwd, Adan]
  world  > 1 :  one graph   [fwd, loss, bwd, bucket-gather, RCCL all-reduce of ONE flat bucket, bucket-scatter, Adan]
                (two graphs around an eager all-reduce when the collective cannot be captured: gloo in the CPU-transport tests)

Everything that changes per step and is not data (lr, Adan bias corrections) lives in device memory written by
Adan.prepare_step(), so the captured kernels never see a stale scalar."""
import torch

from . import hnerv_utils as hu  # noqa: F401
from . import ops
from .dp import GradBucket


class TrainStep:
    def __init__(self, model, optimizer, loss_type, takes_image, batch_shape, device, use_graph=True, warmup_eager=3,
                 process_group=None, world_size=1, clip_max_norm=0.0, force_bucket=False, dp_buckets=None):
        self.model, self.opt, self.loss_type = model, optimizer, loss_type
        self.takes_image = takes_image                       # HNeRV_Boost consumes the frame; NeRV/ENeRV the frame index
        self.dev = device
        B, C, H, W = batch_shape
        self.static_img = torch.zeros(B, C, H, W, dtype=torch.float32, device=device)
        self.static_idx = torch.zeros(B, dtype=torch.float64, device=device)
        self.use_graph = use_graph and clip_max_norm <= 0
        self.clip_max_norm = clip_max_norm
        self.warmup_eager = warmup_eager
        self.n_calls = 0
        self.graph_a = self.graph_b = None
        self.collective_in_graph = False
        self._opt_epoch = getattr(optimizer, "state_epoch", 0)
        self.loss_out = self.psnr_out = None
        self.world = world_size
        # force_bucket: run the multi-GPU code path (bucket gather -> RCCL all-reduce -> scatter, two graphs) on a 1-rank
        # group, so the path the scaling runs take can be tested on a single-GPU box
        # dp_buckets == 2 (or BNERV_DP_BUCKETS=2): the bucket in two segments -- the decoder layers' gradients start their all-reduce on a
        # side stream from an autograd hook at the decoder / stem boundary, next to the rest of the backward (dp.GradBucket); needs a
        # model that names its late parameters and calls the hook (model_nerv.NeRV_Boost).  Default 1: see DESIGN section 5.
        # Two segments are valid only where every early (decoder) gradient is COMPLETE when the stem-output hook fires: the plain
        # step.  A subclass whose backward finishes parameter gradients later (CompressionStep: the convs consume de-quantised weights,
        # the real weight / quantiser gradients appear in the CEM backward, after the hook) would gather zeros for them -- so the
        # environment switch is honoured by TrainStep itself only, and exchange_early() refuses a segment with a missing gradient.
        import os as _os
        if dp_buckets is None:
            dp_buckets = int(_os.environ.get("BNERV_DP_BUCKETS", "1")) if type(self) is TrainStep else 1
        nb = int(dp_buckets)
        if nb == 2 and type(self) is not TrainStep:
            raise ValueError(f"{type(self).__name__}: dp_buckets=2 needs a backward whose decoder gradients are final at the stem boundary "
                             "(the plain TrainStep); use one bucket")
        late = model.dp_late_parameters() if (nb == 2 and hasattr(model, "dp_late_parameters")) else None
        self.bucket = GradBucket(model.parameters(), process_group, force=force_bucket, late_params=late) if (world_size > 1 or force_bucket) else None
        self._early_ok = True                                # False while a graph that cannot hold the collective is being captured
        if self.bucket is not None and self.bucket.two:
            model.dp_hook = self._dp_hook
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._fetching = False                               # inside step_frame(): the step starts with the frame fetch
        self._wplan_entries = 0                              # entries of the capture stream context's weight-fragment plan
        self._clip = self._clip_norms = None                 # bind_clip(): the resident clip step_frame() picks its frame from

    # ---- pieces -------------------------------------------------------------------------------------------------------
    def bind_clip(self, frames, norms):
        """The clip this step trains on, resident on the device: frames [N, C, H, W] fp32, norms [N] fp64 ((idx + 1) / N,
        hnerv_utils.py:47).  step_frame(i) then trains frame i with ONE host -> device copy per step (the 32-byte schedule record that
        also carries i) instead of two device copies and that record: the frame is fetched by the first launch of the (captured) step.
        Needs the Adan optimizer of this package (its schedule record carries the index)."""
        if not hasattr(self.opt, "finish_capture"):
            raise TypeError("bind_clip needs the fused Adan optimizer (its device schedule record carries the frame index)")
        assert frames.shape[1:] == self.static_img.shape[1:] and self.static_img.shape[0] == 1, "bind_clip: one frame per step"
        self._clip = frames.contiguous()
        self._clip_norms = norms.to(torch.float64).contiguous()
        self.graph_a = self.graph_b = None

    def _fetch(self):
        from . import _lib as L
        sel = self.opt._sched[0][1]
        L.check(L.load().bnerv_fetch_frame(L.stream(), L.ptr(self._clip), L.ptr(self._clip_norms), sel.data_ptr() + 5 * 4, self._clip.shape[0],
                                           self._clip[0].numel(), L.ptr(self.static_img), L.ptr(self.static_idx)), "bnerv_fetch_frame")

    def _fwd_bwd(self):
        if self._fetching and not getattr(self, "_fetch_done", False):
            self._fetch()
        self.opt.zero_grad(set_to_none=True)
        inp = self.static_img if self.takes_image else self.static_idx
        img_out, _, _ = self.model(inp, norm_idx=self.static_idx)
        # loss_fn(...).backward() + psnr_fn_single(...) of train_nerv_all.py:337-347 as ONE fused launch sequence: the loss
        # gradient seeds backward directly and the per-sample PSNR comes from the same L2 sums (stats[:, 4])
        loss, stats, grad = ops.loss_value_grad_stats(img_out, self.static_img, self.loss_type)
        if self._lazy_flush_valid():
            # (dx_ok: the model's first block hands its queued input gradient to a flushing operator of this package -- ops._flush_deferred)
            with ops.lazy_flush(dx_ok=getattr(self.model, "lazy_dx_ok", False)):      # slab reductions are flushed by their first reader, not per block
                img_out.backward(grad)
        else:
            img_out.backward(grad)
        self.loss_out = loss
        self.psnr_out = stats[:, 4]

    def _lazy_flush_valid(self):
        """Lazy flushing (ops.lazy_flush) lets backward RETURN weight / bias gradients and TAT channel sums whose slab reductions are
        still queued; only this package's operators (grouped dense, dense GEMM, stand-alone affine) flush before reading.  The invariant
        that makes this safe -- checked here, every step, not only promised by the model's `lazy_flush_ok` flag -- is that NOTHING else
        reads a gradient inside the backward: no parameter holds a .grad (AccumulateGrad would add into it: zero_grad(set_to_none=True)
        above), no parameter carries a tensor hook, and no module carries backward hooks.  Anything else takes the per-block flush."""
        import os
        if os.environ.get("BNERV_LAZY_FLUSH", "1") == "0" or not getattr(self.model, "lazy_flush_ok", False):
            return False
        for p in self.params:
            if p.grad is not None or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                return False
        for m in self.model.modules():
            if m._backward_hooks or m._backward_pre_hooks:
                return False
        return True

    def _planned_fwd_bwd(self):
        """_fwd_bwd inside a capture: the weight fragments of every wide split conv call of the step come from ONE launch at the
        start (the context's frozen plan, include/bnerv.h bnerv_ctx_wplan_*) -- the weights only change in the Adan launch after
        backward.  Without a plan this is _fwd_bwd."""
        from . import _lib as L
        if not self._wplan_entries:
            return self._fwd_bwd()
        lib, c = L.load(), L.ctx()
        fetched = False
        if self._fetching:
            # the plan's launch carries the frame fetch as a second block range: one launch instead of two at the head of every step
            sel = self.opt._sched[0][1]
            rc = lib.bnerv_ctx_wplan_run_fetch(c.handle, L.stream(), L.ptr(self._clip), L.ptr(self._clip_norms), sel.data_ptr() + 5 * 4, self._clip.shape[0],
                                               self._clip[0].numel(), L.ptr(self.static_img), L.ptr(self.static_idx))
            if rc == 0:
                fetched = True
            elif rc != 1:
                L.check(rc, "bnerv_ctx_wplan_run_fetch")
        if not fetched:
            L.check(lib.bnerv_ctx_wplan_run(c.handle, L.stream()), "bnerv_ctx_wplan_run")
        try:
            self._fetch_done = fetched
            self._fwd_bwd()
        finally:
            self._fetch_done = False
            lib.bnerv_ctx_wplan_end(c.handle)

    def _record_wplan(self):
        """One forward + backward on the capture stream, eager, with the stream context recording which weight tensors the wide split
        convs split (gradients are discarded: the next _fwd_bwd starts with zero_grad; no parameter or optimizer state changes).
        Only the plain step: a step whose forward draws random numbers (CompressionStep) would advance its generator here."""
        import os
        from . import _lib as L
        self._wplan_entries = 0
        if type(self) is not TrainStep or os.environ.get("BNERV_WPLAN", "1") == "0":
            return
        # the recording pass is one extra forward + backward: with BatchNorm it would move the running statistics once more than the
        # eager / reference trajectory does, with an active Dropout it would advance the generator -- no plan for such models
        for mod in self.model.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) or (isinstance(mod, torch.nn.modules.dropout._DropoutNd) and mod.p > 0):
                return
        lib = L.load()
        with torch.cuda.stream(self._cap_stream), L.use_ctx(self._cap_ctx) as c:
            L.check(lib.bnerv_ctx_wplan_record(c.handle), "bnerv_ctx_wplan_record")
            early_ok, self._early_ok = self._early_ok, False     # (the recording pass exchanges nothing)
            try:
                self._fwd_bwd()
            finally:
                self._early_ok = early_ok
                n = lib.bnerv_ctx_wplan_freeze(c.handle)
            if n < 0:
                L.check(n, "bnerv_ctx_wplan_freeze")
            self._wplan_entries = n
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()

    def _dp_hook(self, grad):
        """Autograd hook on the stem output (two buckets): the decoder layers' gradients are complete once their deferred slab reductions
        have run -- flush them, then start the early segment's all-reduce."""
        if self._early_ok:
            from . import ops
            ops._flush_deferred(force=True)
            self.bucket.exchange_early()
        return None

    def _eager(self, prepared=False):
        self._fwd_bwd()
        if self.bucket is not None:
            self.bucket.finish()
        if self.clip_max_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip_max_norm)
        if not prepared:
            self.opt.prepare_step()
        self.opt.launch_step()

    def _bucket_calls(self, which):
        import ctypes as C
        from . import _lib as L
        lib = L.load()
        fn, scale = (lib.bnerv_bucket_gather, 1.0 / self.world) if which == "gather" else (lib.bnerv_bucket_scatter, 1.0)
        for ck in self.bucket._build():
            L.check(fn(L.stream(), C.byref(ck), L.ptr(self.bucket.bucket), scale), "bnerv_bucket_" + which)

    def _capture(self):
        """world == 1: one graph.  world > 1 on RCCL: ONE graph as well -- the all-reduce of the flat bucket is captured between the
        gather and the scatter, so a step is a single launch with no host round trip around the collective (BNERV_DP_INGRAPH=0, a
        backend whose collectives cannot be captured (gloo), or a failed capture fall back to graph A -> eager all-reduce -> graph B)."""
        import os
        import torch.distributed as dist
        from . import _lib as L
        torch.cuda.synchronize()
        self._opt_epoch = getattr(self.opt, "state_epoch", 0)
        pool = torch.cuda.graph_pool_handle()
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), None
        # capture on a stream of our own with a library context of our own, whose scratch is reserved BEFORE the capture starts
        # (nothing may allocate inside it; the eager warm-up steps on the current stream sized the scratch)
        if getattr(self, "_cap_stream", None) is None:
            self._cap_stream = torch.cuda.Stream(device=self.dev)
            self._cap_ctx = L.new_ctx()
        else:
            L.reserve_ctx(self._cap_ctx)
        if hasattr(self.opt, "begin_capture"):
            self.opt.begin_capture()            # this capture's own descriptor tables (optimizer.Adan.launch_step, capture contract)
        try:
            with L.use_ctx(self._cap_ctx):
                self._capture_in_ctx(pool)
        finally:
            if hasattr(self.opt, "finish_capture"):
                # uploaded once, outside the graph; the tables live as long as this object's graphs
                self._opt_tables = self.opt.finish_capture()

    def _capture_in_ctx(self, pool):
        import os
        import torch.distributed as dist
        from . import _lib as L
        self._early_ok = False
        self._record_wplan()
        self._early_ok = True
        cap = dict(pool=pool, stream=self._cap_stream)
        if self.bucket is None:
            with L.graph_capture(self.graph_a, **cap):
                self._planned_fwd_bwd()
                self.opt.launch_step()
            return

        def head():
            self._planned_fwd_bwd()
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            self._bucket_calls("gather")

        def tail():
            self._bucket_calls("scatter")
            self.opt.launch_step()
        backend = dist.get_backend(self.bucket.group) if dist.is_initialized() else ""
        if backend == "nccl" and os.environ.get("BNERV_DP_INGRAPH", "1") != "0":
            try:
                with L.graph_capture(self.graph_a, **cap):
                    self._planned_fwd_bwd()                   # (two buckets: the hook starts the early segment's all-reduce in here)
                    self.bucket.finish()                      # gather -> all-reduce -> [join] -> scatter
                    self.opt.launch_step()
                self.collective_in_graph = True
                return
            except Exception as e:                        # capture of the collective is not available on this stack: two graphs
                import warnings
                warnings.warn(f"RCCL all-reduce could not be captured ({type(e).__name__}: {e}); falling back to the two-graph step")
                torch.cuda.synchronize()
                self.graph_a = torch.cuda.CUDAGraph()
        self.collective_in_graph = False
        self._early_ok = False                              # graph A cannot hold a collective: one exchange between the two graphs
        self.bucket._early_inflight = False
        self.bucket._side_used = None
        with L.graph_capture(self.graph_a, **cap):
            head()
        self.graph_b = torch.cuda.CUDAGraph()
        with L.graph_capture(self.graph_b, **cap):
            tail()

    # ---- one step ------------------------------------------------------------------------------------------------------
    def __call__(self, img, norm_idx):
        """img [B,3,H,W] fp32 and norm_idx [B] fp64, already on the device.  Returns (loss, psnr[B]) device tensors that are
        overwritten by the next call."""
        if self._fetching:
            self._fetching = False
            self.graph_a = self.graph_b = None      # the captured step started with a frame fetch: capture the plain form
        self.static_img.copy_(img, non_blocking=True)
        self.static_idx.copy_(norm_idx, non_blocking=True)
        return self._run(None)

    def step_frame(self, i):
        """Train frame i of the clip given to bind_clip().  Same step as __call__(frames[i:i+1], norms[i:i+1])."""
        assert self._clip is not None, "step_frame: call bind_clip(frames, norms) first"
        i = int(i)
        if not 0 <= i < self._clip.shape[0]:    # (the fetch kernel clamps the index it reads from the schedule record: never hand it a bad one)
            raise IndexError(f"step_frame: frame {i} outside the bound clip of {self._clip.shape[0]} frames")
        if not self._fetching:
            self._fetching = True
            self.graph_a = self.graph_b = None
        return self._run(int(i))

    def _run(self, frame):
        if self.graph_a is not None and getattr(self.opt, "state_epoch", 0) != self._opt_epoch:
            self.graph_a = self.graph_b = None      # optimizer state tensors were replaced (restart_opt / load_state_dict): re-capture
        if frame is not None:
            self.opt.prepare_step(aux=frame)    # the schedule record (with the frame index) precedes the step's first launch
        if not self.use_graph or self.n_calls < self.warmup_eager:
            self._eager(prepared=frame is not None)
        else:
            if frame is None:
                self.opt.prepare_step()         # host side of THIS step (a capture below records its kernels, runs nothing)
            if self.graph_a is None:
                self._capture()
            self.graph_a.replay()
            if self.graph_b is not None:
                import torch.distributed as dist
                dist.all_reduce(self.bucket.bucket, op=dist.ReduceOp.SUM, group=self.bucket.group)
                self.graph_b.replay()
        self.n_calls += 1
        return self.loss_out, self.psnr_out


class CompressionStep(TrainStep):
    """The rate-distortion step of train_nerv_compression.py:354-367 (quantise + rate term over all tensors -> forward with the
    embedding's rate -> loss + lambda * bpp while bpp / N exceeds the target -> backward -> Adan) as ONE captured graph.  The
    `bpp > target` decision is a device-side gate and the training noise comes from the capture-aware default generator, so
    nothing in the step needs the host.  With several ranks the quantiser / rate parameters are part of the same flat gradient
    bucket as the weights (they are model parameters), exchanged exactly as in TrainStep.  Returns (final_loss, psnr[B]);
    `bpp_out` holds bits per pixel (x N)."""

    def __init__(self, model, optimizer, entropy_model, args, batch_shape, device, use_graph=True, warmup_eager=3, world_size=1,
                 process_group=None, force_bucket=False):
        super().__init__(model, optimizer, args.loss, "HNeRV_Boost" in args.model or 'pe' not in args.embed, batch_shape, device,
                         use_graph=use_graph, warmup_eager=warmup_eager, clip_max_norm=getattr(args, "clip_max_norm", 0.0),
                         world_size=world_size, process_group=process_group, force_bucket=force_bucket)
        self.entropy_model, self.cargs = entropy_model, args
        self.bpp_out = None

    def _fwd_bwd(self):
        a, m = self.cargs, self.model
        if self._fetching:
            self._fetch()                       # bind_clip() / step_frame(): the step starts with the frame fetch, as TrainStep's does
        self.opt.zero_grad(set_to_none=True)
        m.cal_params(self.entropy_model)
        inp = self.static_img if self.takes_image else self.static_idx
        if a.embed_entropy:
            img_out, _, _ = m(inp, entropy_model=self.entropy_model, norm_idx=self.static_idx)
            bpp = (m.get_bitrate_sum(name="bitrate") + m.bitrate_e_dict["bitrate"] * a.full_data_length) / a.final_size
        else:
            img_out, _, _ = m(inp, norm_idx=self.static_idx)
            bpp = m.get_bitrate_sum(name="bitrate") / a.final_size
        out_loss = hu.loss_fn(img_out, self.static_img, self.loss_type)
        gate = (bpp.detach() / a.full_data_length > a.target_bpp).to(out_loss.dtype)     # `if bpp / N > target_bpp` (:363) on the device
        final_loss = out_loss + gate * a.lambda_rate * bpp
        final_loss.backward()
        self.loss_out = final_loss.detach()
        self.bpp_out = bpp.detach()
        self.psnr_out = ops.psnr(img_out.detach(), self.static_img)

    def _capture(self):
        # The model's rate dictionaries and dequantised tensors still reference the autograd graph of the last eager step; that
        # keeps its AccumulateGrad nodes (created on the default stream) alive, the capture would reuse them and pull the
        # default stream into the capture (hipStreamEndCapture crashes on it).  Detach every such reference first.
        for mod in self.model.modules():
            for k, v in list(vars(mod).items()):
                if torch.is_tensor(v) and v.grad_fn is not None:
                    setattr(mod, k, v.detach())
                elif isinstance(v, dict) and not k.startswith("_"):
                    for kk, vv in list(v.items()):
                        if torch.is_tensor(vv) and vv.grad_fn is not None:
                            v[kk] = vv.detach()
        super()._capture()


class DecodeGraph:
    """Forward-only decode of one batch as a captured hipGraph (row N4: the rate evaluate() logs as "FPS" under --eval_fps,
    train_nerv_all.py:492-496 of the reference, is ~35 Python-driven launches per frame when run eagerly).  The graph replays
    `model(cur_input, embed, norm_idx=norm_idx)` on static copies of its inputs; `__call__` refreshes them, replays, and returns
    (img_out, seconds) with the same synchronised-timer definition the model's own `dec_time` uses."""

    def __init__(self, model, cur_input, embed, norm_idx):
        import time as _time
        self._time = _time
        self.inp = cur_input.clone()
        self.embed = None if embed is None else embed.clone()
        self.norm = norm_idx.clone()
        td = getattr(model, "time_decode", False)
        model.time_decode = False                           # no host synchronisation inside the capture
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        import os
        from . import _lib as L
        lib = L.load()
        self._ctx = L.new_ctx()
        plan = os.environ.get("BNERV_WPLAN", "1") != "0"
        try:
            with torch.no_grad(), torch.cuda.stream(side), L.use_ctx(self._ctx) as c:
                # warm-up: workspaces and lazily built tables exist before capture; the first one records which weight tensors the
                # wide split convs split.  A decoder's weights are fixed, so their 16-bit fragments are prepared ONCE, here
                # (bnerv_ctx_wplan_*), and the captured forward launches no preparation at all; refresh() after a weight change.
                if plan:
                    L.check(lib.bnerv_ctx_wplan_record(c.handle), "bnerv_ctx_wplan_record")
                model(self.inp, self.embed, norm_idx=self.norm)
                self.wplan_entries = lib.bnerv_ctx_wplan_freeze(c.handle) if plan else 0
                if self.wplan_entries < 0:
                    L.check(self.wplan_entries, "bnerv_ctx_wplan_freeze")
                model(self.inp, self.embed, norm_idx=self.norm)
                L.check(lib.bnerv_ctx_wplan_run(c.handle, L.stream()), "bnerv_ctx_wplan_run")
                try:
                    self.graph = torch.cuda.CUDAGraph()
                    with L.graph_capture(self.graph, stream=side):
                        self.out = model(self.inp, self.embed, norm_idx=self.norm)[0]
                finally:
                    lib.bnerv_ctx_wplan_end(c.handle)
        finally:
            model.time_decode = td
            cur.wait_stream(side)

    def refresh(self):
        """Re-prepare the weight fragments the captured forward reads (one launch) after the model's weights were written."""
        from . import _lib as L
        L.check(L.load().bnerv_ctx_wplan_run(self._ctx.handle, L.stream()), "bnerv_ctx_wplan_run")
        L.load().bnerv_ctx_wplan_end(self._ctx.handle)

    def matches(self, cur_input, embed, norm_idx):
        return (cur_input.shape == self.inp.shape and norm_idx.shape == self.norm.shape and
                (embed is None) == (self.embed is None) and (embed is None or embed.shape == self.embed.shape))

    def __call__(self, cur_input, embed, norm_idx):
        self.inp.copy_(cur_input)
        self.norm.copy_(norm_idx)
        if self.embed is not None:
            self.embed.copy_(embed)
        torch.cuda.synchronize()
        t0 = self._time.time()
        self.graph.replay()
        torch.cuda.synchronize()
        return self.out, self._time.time() - t0
