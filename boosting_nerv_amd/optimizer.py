"""Adan optimizer -- host-side mirror of the reference's optimizer.py:39-235 (same constructor signature, defaults,
param_group / state key names: 'step', 'exp_avg', 'exp_avg_sq', 'exp_avg_diff', 'neg_pre_grad'), with the update itself
done by ONE fused multi-tensor HIP kernel per <=48 tensors (bnerv_adan_multi_tensor) instead of ~17 torch._foreach
launches (optimizer.py:296-362).  This is the slot the reference reserves for the external `fused_adan` CUDA extension
(optimizer.py:365-395), which it never ships.

`step()` = `prepare_step()` (host: step count, bias corrections, lr -> an 8-float device buffer, via a pinned async copy)
+ `launch_step()` (kernel launches only -> capturable in a hipGraph; every scalar that changes per step is read from the
device buffer, so a captured step replays with a moving LR schedule)."""
import ctypes as C
import math

import torch
from torch.optim.optimizer import Optimizer

from . import _lib as L


class Adan(Optimizer):
    _RING = 512

    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0,
                 no_prox=False, foreach: bool = True, fused: bool = False):
        if not 0.0 <= max_grad_norm:
            raise ValueError("Invalid Max grad norm: {}".format(max_grad_norm))
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        for i in range(3):
            if not 0.0 <= betas[i] < 1.0:
                raise ValueError("Invalid beta parameter at index {}: {}".format(i, betas[i]))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm, no_prox=no_prox,
                        foreach=foreach, fused=fused)
        super().__init__(params, defaults)
        self._sched = {}          # group index -> (pinned host [5], device [5])
        self._cap_open = None     # begin_capture() .. finish_capture(): group index -> (pinned host, device) descriptor table of THIS capture
        self._chunk_cache = {}    # group index -> (key, device descriptor table, n, blocks, tensors the table points into, pending host table)
        self.state_epoch = 0      # bumped whenever state tensors are replaced: a captured step (engine.TrainStep) re-captures

    def _invalidate(self):
        """The descriptors of launch_step() hold raw device pointers into the state tensors; whoever replaces those tensors
        (restart_opt, load_state_dict, unpickling) must drop the descriptors, and a captured graph of the step with them."""
        self._chunk_cache = {}
        self.state_epoch = getattr(self, "state_epoch", 0) + 1

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("no_prox", False)
        self.__dict__.setdefault("_sched", {})
        self.__dict__["_cap_open"] = None
        self._invalidate()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._invalidate()

    @torch.no_grad()
    def restart_opt(self):
        for group in self.param_groups:
            group["step"] = 0
            for p in group["params"]:
                if p.requires_grad:
                    state = self.state[p]
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                    state["exp_avg_diff"] = torch.zeros_like(p)
        self._invalidate()

    # ------------------------------------------------------------------------------------------------------------------
    def _clip_coef(self):
        if self.defaults["max_grad_norm"] <= 0:
            return 1.0
        # optimizer.py:136-156: global-norm clipping (a device sync, exactly as in the reference; off in every recipe)
        device = self.param_groups[0]["params"][0].device
        total = torch.zeros(1, device=device)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    total.add_(p.grad.pow(2).sum())
        total = torch.sqrt(total)
        return torch.clamp(self.defaults["max_grad_norm"] / (total + self.param_groups[-1]["eps"]), max=1.0).item()

    @torch.no_grad()
    def prepare_step(self, aux=None):
        """Host side of a step: advance the step count and publish {lr, bc1, bc2, sqrt(bc3), first_step, aux} to the device.
        `aux` (a float, optional) rides in slot 5 of the same 32-byte record: engine.TrainStep puts the index of the step's frame
        there when the clip is resident on the device, so a step costs ONE host -> device copy."""
        for gi, group in enumerate(self.param_groups):
            beta1, beta2, beta3 = group["betas"]
            group["step"] = group.get("step", 0) + 1
            bc1 = 1.0 - beta1 ** group["step"]
            bc2 = 1.0 - beta2 ** group["step"]
            bc3 = 1.0 - beta3 ** group["step"]
            dev = group["params"][0].device
            L.require_device(group["params"][0], "parameter")
            if gi not in self._sched:
                # a RING of pinned slots: the async copy of step k may still be queued when the host prepares step k+1.  Every slot
                # carries an event recorded after its copy was enqueued; a slot is rewritten only once that copy has executed, which
                # bounds the host's run-ahead to _RING steps by construction (in practice the HIP launch queue already blocks the host
                # long before 512 graph launches are outstanding; the guard makes that an invariant instead of an observation)
                self._sched[gi] = (torch.zeros(self._RING, 8, dtype=torch.float32).pin_memory(), torch.zeros(8, dtype=torch.float32, device=dev),
                                   [None] * self._RING)
            ring, devbuf, events = self._sched[gi]
            slot = group["step"] % self._RING
            if events[slot] is not None:
                events[slot].synchronize()
            host = ring[slot]
            host[0], host[1], host[2], host[3] = group["lr"], bc1, bc2, math.sqrt(bc3)
            host[4] = 1.0 if group["step"] == 1 else 0.0
            if aux is not None:
                host[5] = float(aux)
            devbuf.copy_(host, non_blocking=True)
            if events[slot] is None:
                events[slot] = torch.cuda.Event()
            events[slot].record()

    def _ensure_state(self, p, step, clip):
        state = self.state[p]
        if len(state) == 0:
            state["exp_avg"] = torch.zeros_like(p)
            state["exp_avg_sq"] = torch.zeros_like(p)
            state["exp_avg_diff"] = torch.zeros_like(p)
        if "neg_pre_grad" not in state:
            # step 1: the kernel's first_step flag substitutes -g; later first appearances follow optimizer.py:190-192
            state["neg_pre_grad"] = torch.zeros_like(p) if step <= 1 else p.grad.clone().mul_(-clip)
        return state

    @torch.no_grad()
    def launch_step(self, clip=1.0):
        """Device side of a step: ONE fused launch over a device-resident descriptor table (any number of tensors).  No sync; the
        only host<->device traffic is the table upload when a tensor address changed (never in a replayed step).

        Capture contract: a captured launch records only the ADDRESS of its descriptor table; the content is uploaded by
        finish_capture().  Inside a stream capture this method therefore raises unless a begin_capture() .. finish_capture() bracket
        is open (engine.TrainStep._capture opens one); every bracket gets a table of its own, which the caller keeps alive with its
        graph (finish_capture() returns it), so a second capture never rewrites the table an earlier graph still replays with."""
        lib = L.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            # the key covers every pointer a descriptor holds (parameter, gradient and the four state tensors), and the cache
            # entry keeps the state tensors alive, so a descriptor can never point at freed memory
            sts = [self._ensure_state(p, group.get("step", 1), clip) for p in ps]
            key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         st["exp_avg_diff"].data_ptr(), st["neg_pre_grad"].data_ptr()) for p, st in zip(ps, sts))
            capturing = torch.cuda.is_current_stream_capturing()
            ck = (gi, capturing)        # a captured launch owns its table: an eager step in between must not rewrite the addresses it replays with
            if capturing and self._cap_open is None:
                raise L.BnervError("Adan.launch_step() inside a stream capture needs an open begin_capture() .. finish_capture() bracket "
                                   "(the captured launch reads a descriptor table that finish_capture() uploads)")
            cached = self._chunk_cache.get(ck)
            if capturing and cached is not None and cached[1] is not self._cap_open[gi][1]:
                cached = None               # a table of an earlier capture: that graph keeps it; this capture writes its own
            if cached is None or cached[0] != key:
                # (not p.grad: it is alive whenever the step launches, and pinning it would move the next eager gradient elsewhere)
                keep = [(p, st["exp_avg"], st["exp_avg_sq"], st["exp_avg_diff"], st["neg_pre_grad"]) for p, st in zip(ps, sts)]
                tab = (L.AdanEntry * len(ps))()
                blocks = 0
                for j, (p, st) in enumerate(zip(ps, sts)):
                    if not (p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                        raise L.BnervError("fused Adan needs contiguous fp32 parameters and gradients")
                    e = tab[j]
                    e.p, e.g = p.data_ptr(), p.grad.data_ptr()
                    e.exp_avg, e.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    e.exp_avg_diff, e.neg_pre_grad = st["exp_avg_diff"].data_ptr(), st["neg_pre_grad"].data_ptr()
                    e.n, e.bstart = p.numel(), blocks
                    blocks += lib.bnerv_adan_table_blocks(p.numel())
                raw = bytes(tab)
                if capturing:
                    # nothing may allocate pinned or device memory inside a capture: begin_capture() set both aside
                    host, dev_tab = self._cap_open[gi]
                    if host.numel() < len(raw):
                        raise L.BnervError("fused Adan: the capture-time descriptor table is larger than the one begin_capture() reserved")
                    C.memmove(host.data_ptr(), raw, len(raw))
                else:
                    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8).pin_memory()
                    # ONE device table per group, reused while the tensor count does not change
                    dev_tab = cached[1] if (cached is not None and cached[1].numel() == host.numel()) else torch.empty(host.numel(), dtype=torch.uint8, device=ps[0].device)
                cached = [key, dev_tab, len(ps), blocks, keep, host]
                self._chunk_cache[ck] = cached
            if cached[5] is not None and not capturing:
                cached[1].copy_(cached[5], non_blocking=True)      # stream-ordered before the launch below
                cached[5] = None                                   # (inside a capture the upload waits for finish_capture(): a copy node would replay every step)
            beta1, beta2, beta3 = group["betas"]
            hyper = L.AdanHyper(beta1, beta2, beta3, group["eps"], group["weight_decay"], clip, int(group["no_prox"]),
                                self._sched[gi][1].data_ptr())
            L.check(lib.bnerv_adan_table(L.stream(), cached[1].data_ptr(), cached[2], cached[3], C.byref(hyper)), "bnerv_adan_table")

    def begin_capture(self):
        """Before a hipGraph capture that will contain launch_step(): reserve THIS capture's descriptor tables (pinned host + device,
        one pair per parameter group, sized for every parameter) -- nothing may allocate inside the capture."""
        tabs = {}
        for gi, group in enumerate(self.param_groups):
            nb = max(len(group["params"]), 1) * C.sizeof(L.AdanEntry)
            dev = group["params"][0].device
            host = torch.empty(nb, dtype=torch.uint8)
            tabs[gi] = (host.pin_memory() if dev.type == "cuda" else host, torch.empty(nb, dtype=torch.uint8, device=dev))
        self._cap_open = tabs

    def finish_capture(self):
        """After a hipGraph capture that contained launch_step(): upload the descriptor tables that capture referenced (the captured
        launch holds the table's ADDRESS; its content is written here, once, outside the graph) and close the bracket.  Returns the
        device tables: the owner of the graph keeps them alive as long as the graph."""
        for cached in self._chunk_cache.values():
            if cached[5] is not None:
                cached[1].copy_(cached[5], non_blocking=True)
                torch.cuda.current_stream().synchronize()          # the pinned buffer dies with the bracket
                cached[5] = None
        tabs, self._cap_open = self._cap_open, None
        return [] if tabs is None else [t[1] for t in tabs.values()]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip = self._clip_coef()
        self.prepare_step()
        self.launch_step(clip)
        return loss
