"""Frame-sharded data parallelism: one process per GPU, the whole decoder replicated, ONE flat fp32 gradient bucket
all-reduced per step over RCCL/xGMI (torch.distributed backend 'nccl' on ROCm), then averaged.

Replaces the reference's DistributedDataParallel wrap (train_nerv_all.py:159-168, :253-254: NCCL all-reduce of every
parameter gradient in 25 MB buckets, mean over ranks).  The models here are 1.5-3 M parameters (6-12 MB), so a single
bucket and a single collective per step is the right granularity for point-to-point xGMI links; gather / scatter of the
bucket are one multi-tensor HIP launch each (bnerv_bucket_gather / _scatter).  On CPU tensors (gloo, used by the
world_size-2 tests) the same logic runs through torch's flatten helpers -- host plumbing, no arithmetic of the path."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class GradBucket:
    def __init__(self, params, process_group=None, force=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force          # run the exchange even on a 1-rank group (tests of the multi-GPU path)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        p0 = self.params[0]
        self.bucket = torch.zeros(off, dtype=torch.float32, device=p0.device)
        self._chunks = None
        self._key = None

    # -- HIP gather/scatter descriptors (rebuilt only when a gradient moves)
    def _build(self):
        key = tuple(p.grad.data_ptr() for p in self.params)
        if key == self._key:
            return self._chunks
        chunks = []
        step = L.ADAN_MAX_TENSORS * 2
        for i0 in range(0, len(self.params), step):
            ck = L.BucketChunk()
            sub = self.params[i0:i0 + step]
            for j, p in enumerate(sub):
                ck.t[j], ck.n[j], ck.off[j] = p.grad.data_ptr(), p.numel(), self.offsets[i0 + j]
            ck.n_tensors = len(sub)
            chunks.append(ck)
        self._chunks, self._key = chunks, key
        return chunks

    @torch.no_grad()
    def allreduce_mean(self):
        """grad <- mean over ranks of grad (DDP semantics).  No-op for world size 1."""
        if self.world == 1 and not self.force:
            return
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self.bucket.is_cuda:
            lib = L.load()
            chunks = self._build()
            for ck in chunks:
                L.check(lib.bnerv_bucket_gather(L.stream(), C.byref(ck), L.ptr(self.bucket), 1.0 / self.world), "bnerv_bucket_gather")
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            for ck in chunks:
                L.check(lib.bnerv_bucket_scatter(L.stream(), C.byref(ck), L.ptr(self.bucket), 1.0), "bnerv_bucket_scatter")
        else:
            flat = torch.cat([p.grad.reshape(-1) for p in self.params]).mul_(1.0 / self.world)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            for p, off in zip(self.params, self.offsets):
                p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))


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
