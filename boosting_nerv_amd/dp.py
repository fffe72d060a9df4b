"""Frame-sharded data parallelism: one process per GPU, the whole decoder replicated, ONE flat fp32 gradient bucket
all-reduced per step over RCCL/xGMI (torch.distributed backend 'nccl' on ROCm), then averaged.

Replaces the reference's DistributedDataParallel wrap (train_nerv_all.py:159-168, :253-254: NCCL all-reduce of every
parameter gradient in 25 MB buckets, mean over ranks).  The models here are 1.5-3 M parameters (6-12 MB), so a single
bucket and a single collective per step is the right granularity for point-to-point xGMI links; gather / scatter of the
bucket are one multi-tensor HIP launch each (bnerv_bucket_gather / _scatter).  On CPU tensors (gloo, used by the
world_size-2 tests) the same logic runs through torch's flatten helpers -- host plumbing, no arithmetic of the path.

Two-segment form (late_params given): the bucket is laid out [early | late].  `late` are the parameters whose gradients the backward
produces LAST (NeRV: the stem MLP, 74 % of the bytes, and the modulation MLPs); `early` the decoder layers'.  exchange_early() is called
from an autograd hook at the decoder / stem boundary: the early segment is gathered and its all-reduce starts on a side stream (an
async work item on gloo) while the rest of the backward runs; finish() all-reduces the late segment, joins and scatters.  Inside a
stream capture the fork and the join become graph edges, so the step stays ONE graph.  The mean is the same sum of the same numbers
in the same order per element as with one bucket: results are bit-equal."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class GradBucket:
    def __init__(self, params, process_group=None, force=False, late_params=None):
        ps = [p for p in params if p.requires_grad]
        late_ids = {id(p) for p in (late_params or [])}
        early = [p for p in ps if id(p) not in late_ids]
        late = [p for p in ps if id(p) in late_ids]
        self.two = bool(early) and bool(late)
        self.params = early + late                           # bucket order: the early segment first
        self.n_early = len(early) if self.two else len(self.params)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force          # run the exchange even on a 1-rank group (tests of the multi-GPU path)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.split = self.offsets[self.n_early] if self.two else off
        p0 = self.params[0]
        self.bucket = torch.zeros(off, dtype=torch.float32, device=p0.device)
        self._chunks, self._key = {}, {}
        # two side streams: torch runs a synchronous collective on the CURRENT stream and records its end event there, and the process
        # group's watchdog thread keeps querying that event for up to 100 ms after the collective has finished.  HIP refuses the query
        # of an event whose last record was on a stream that is capturing now (the watchdog then aborts the process), so the stream
        # of the eager exchange (warm-up steps) is never the one a capture forks onto (_lib.graph_capture)
        self._side = torch.cuda.Stream(device=p0.device) if (self.two and p0.is_cuda) else None
        self._side_cap = torch.cuda.Stream(device=p0.device) if (self.two and p0.is_cuda) else None
        self._side_used = None
        self._early_inflight = False
        self._work = None

    def _range(self, which):
        return {"all": (0, len(self.params)), "early": (0, self.n_early), "late": (self.n_early, len(self.params))}[which]

    # -- HIP gather/scatter descriptors (rebuilt only when a gradient moves)
    def _build(self, which="all"):
        lo, hi = self._range(which)
        key = tuple(p.grad.data_ptr() for p in self.params[lo:hi])
        if self._key.get(which) == key:
            return self._chunks[which]
        chunks = []
        step = L.ADAN_MAX_TENSORS * 2
        for i0 in range(lo, hi, step):
            ck = L.BucketChunk()
            sub = self.params[i0:min(i0 + step, hi)]
            for j, p in enumerate(sub):
                ck.t[j], ck.n[j], ck.off[j] = p.grad.data_ptr(), p.numel(), self.offsets[i0 + j]
            ck.n_tensors = len(sub)
            chunks.append(ck)
        self._chunks[which], self._key[which] = chunks, key
        return chunks

    def _ensure_grads(self, which):
        lo, hi = self._range(which)
        for p in self.params[lo:hi]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)

    def _gather(self, which):
        lo, hi = self._range(which)
        if self.bucket.is_cuda:
            lib = L.load()
            for ck in self._build(which):
                L.check(lib.bnerv_bucket_gather(L.stream(), C.byref(ck), L.ptr(self.bucket), 1.0 / self.world), "bnerv_bucket_gather")
        else:
            for p, off in zip(self.params[lo:hi], self.offsets[lo:hi]):
                self.bucket[off:off + p.numel()].copy_(p.grad.reshape(-1)).mul_(1.0 / self.world)

    def _scatter(self, which):
        lo, hi = self._range(which)
        if self.bucket.is_cuda:
            lib = L.load()
            for ck in self._build(which):
                L.check(lib.bnerv_bucket_scatter(L.stream(), C.byref(ck), L.ptr(self.bucket), 1.0), "bnerv_bucket_scatter")
        else:
            for p, off in zip(self.params[lo:hi], self.offsets[lo:hi]):
                p.grad.copy_(self.bucket[off:off + p.numel()].view_as(p.grad))

    @torch.no_grad()
    def allreduce_mean(self):
        """grad <- mean over ranks of grad (DDP semantics), ONE collective over the whole bucket.  No-op for world size 1."""
        if self.world == 1 and not self.force:
            return
        self._ensure_grads("all")
        self._gather("all")
        dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
        self._scatter("all")

    @torch.no_grad()
    def exchange_early(self):
        """Called INSIDE the backward once the early parameters' gradients are complete on the current stream: start their all-reduce
        next to the rest of the backward.  finish() must follow the backward."""
        if not self.two or (self.world == 1 and not self.force) or self._early_inflight:
            return
        lo, hi = self._range("early")
        missing = sum(1 for p in self.params[lo:hi] if p.grad is None)
        if missing:
            # a zero-filled segment would be all-reduced now and scattered OVER the real gradients by finish(): refuse (the caller's
            # backward produces these gradients after the hook -- one bucket is the valid form for it)
            raise RuntimeError(f"GradBucket.exchange_early: {missing} early parameter(s) have no gradient yet at the hook; "
                               "this backward cannot use two bucket segments")
        self._gather("early")
        seg = self.bucket[:self.split]
        if self._side is not None:
            side = self._side_cap if torch.cuda.is_current_stream_capturing() else self._side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
            self._side_used = side
        else:
            self._work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._early_inflight = True

    @torch.no_grad()
    def finish(self):
        """After the backward: the late segment's all-reduce, the join with the early one, the scatter.  Without an early exchange in
        flight this is allreduce_mean()."""
        if not self._early_inflight:
            return self.allreduce_mean()
        self._ensure_grads("late")
        self._gather("late")
        dist.all_reduce(self.bucket[self.split:], op=dist.ReduceOp.SUM, group=self.group)
        if self._side_used is not None:
            torch.cuda.current_stream().wait_stream(self._side_used)
            self._side_used = None
        elif self._work is not None:
            self._work.wait()
            self._work = None
        self._early_inflight = False
        self._scatter("all")


def shard_indices(n, rank, world, seed=0, epoch=0, shuffle=True, drop_last=False):
    """The index set torch's DistributedSampler(dataset) hands to `rank` (train_nerv_all.py:176,189: shuffle=True, seed=0,
    set_epoch never called => epoch 0 order every epoch): seeded permutation, padded to a multiple of `world`, rank::world."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if not drop_last:
        total = -(-n // world) * world
        pad = total - len(idx)
        if pad > 0:
            idx += (idx * (-(-pad // len(idx))))[:pad]
    else:
        total = (n // world) * world
        idx = idx[:total]
    return idx[rank:total:world]
