"""Host-side utilities -- mirror of the reference's hnerv_utils.py for the train path: dataset, split, LR schedule,
loss_fn / psnr / ms-ssim entry points (HIP kernels underneath), post-hoc 8-bit quantisation used by evaluate(), and the
distributed scalar reduction.  Same function names, argument meaning and return conventions as the reference; lines cited
per function."""
import math
import os
import random

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.data import Dataset

from . import ops, synth


# ----------------------------------------------------------------------------------------------------------------------
# dataset                                                                               reference hnerv_utils.py:16-56
# ----------------------------------------------------------------------------------------------------------------------
def _to_tensor(pil_img):
    arr = np.asarray(pil_img, dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).to(torch.float32).div(255)    # == ToTensor()


def _center_crop(img, ch, cw):
    w, h = img.size
    top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
    return img.crop((left, top, left + cw, top + ch))


class VideoDataSet(Dataset):
    """Sorted directory of frames (PNG/JPG), centre-cropped to --crop_list, ToTensor, norm_idx = (idx+1)/N  (:19-47).
    ``--data_path synthetic:bunny | synthetic:uvg | synthetic:NxHxW`` selects the in-repo deterministic generator instead
    (no dataset ships with the reference)."""

    def __init__(self, args):
        self.crop_h, self.crop_w = [int(x) for x in args.crop_list.split("_")[:2]]
        self.synthetic = None
        if str(args.data_path).startswith("synthetic:"):
            n, h, w = synth.parse_spec(args.data_path)
            if (h, w) != (self.crop_h, self.crop_w):
                h, w = self.crop_h, self.crop_w
            self.synthetic = synth.SyntheticVideo(n, h, w)
            self.samples = list(range(n))
            self._cache = {}
        else:
            self.samples = [os.path.join(args.data_path, x) for x in sorted(os.listdir(args.data_path))]
        if getattr(args, "interpolation", False) and len(self.samples) % 2 == 0:
            self.samples.pop()
        self.crop = True
        self.final_size = self.crop_h * self.crop_w
        self.embed_inter = getattr(args, "embed_inter", False) and getattr(args, "interpolation", False)
        if self.synthetic is None:
            from PIL import Image
            first = Image.open(self.samples[0]).convert("RGB")
            if not (first.height >= self.crop_h and first.width >= self.crop_w):
                raise NotImplementedError("frames smaller than --crop_list (bicubic up-sampling branch, hnerv_utils.py:29-31) are not supported")

    def __len__(self):
        return len(self.samples)

    def _load(self, idx):
        if self.synthetic is not None:
            if idx not in self._cache:
                self._cache[idx] = self.synthetic.frame(idx)
            return self._cache[idx]
        from PIL import Image
        return _to_tensor(_center_crop(Image.open(self.samples[idx]).convert("RGB"), self.crop_h, self.crop_w))

    def __getitem__(self, idx):
        img = self._load(idx)
        norm_idx = float(idx + 1) / len(self.samples)
        if self.embed_inter:
            if idx % 2 == 0:
                pre_img, post_img = img, img
            else:
                pre_img, post_img = self._load(idx - 1), self._load(idx + 1)
            return {"img": img, "idx": idx, "norm_idx": norm_idx, "pre_img": pre_img, "post_img": post_img}
        return {"img": img, "idx": idx, "norm_idx": norm_idx}


class TransformInput(nn.Module):                                                         # reference hnerv_utils.py:59-84
    def __init__(self, args):
        super().__init__()
        self.inpanting = args.inpanting
        if "inpanting_fixed" in self.inpanting:
            self.inpaint_size = int(self.inpanting.split("_")[-1]) // 2

    @property
    def identity(self):
        """True when input == gt and the mask is all ones: the train loop then skips the two full-frame multiplies by one
        (train_nerv_all.py:343) -- identical result, 3 fewer passes over a frame."""
        return "inpanting" not in self.inpanting

    def forward(self, img, idx):
        if self.identity:
            return img, img, None
        gt = img.clone()
        h, w = img.shape[-2:]
        inpaint_mask = torch.ones((h, w), device=img.device)
        if "center" in self.inpanting:
            ih, iw = h // 8, w // 8
            cx, cy = int(0.5 * h), int(0.5 * w)
            inpaint_mask[cx - ih: cx + ih, cy - iw: cy + iw] = 0
        elif "fixed" in self.inpanting:
            for cx, cy in [(1 / 2, 1 / 2), (1 / 4, 1 / 4), (1 / 4, 3 / 4), (3 / 4, 1 / 4), (3 / 4, 3 / 4)]:
                cx, cy = int(cx * h), int(cy * w)
                inpaint_mask[cx - self.inpaint_size: cx + self.inpaint_size, cy - self.inpaint_size: cy + self.inpaint_size] = 0
        inp = (img * inpaint_mask).clamp(min=0, max=1)
        return inp, gt, inpaint_mask.detach()


def data_split(img_list, split_num_list, shuffle_data, rand_num=0):
    """Periodic train / validation partition of the frame list (--data_split a_b_c, reference hnerv_utils.py:87-98): inside every
    period of c frames the first a are training frames and those from position b on are held out.  `shuffle_data` permutes the
    list first with Random(rand_num) -- the same generator call as the reference, so the same split."""
    n_seen, n_train_end, period = split_num_list
    frames = list(img_list)
    if shuffle_data:
        random.Random(rand_num).shuffle(frames)
    train = [f for pos, f in enumerate(frames) if pos % period < n_seen]
    held_out = [f for pos, f in enumerate(frames) if pos % period >= n_train_end]
    return train, held_out


# ----------------------------------------------------------------------------------------------------------------------
# post-hoc 8-bit quantisation used by evaluate() / quant_model (reference hnerv_utils.py:101-134, :183-186): eval-time
# reporting on stock torch ops, not part of the train hot path
# ----------------------------------------------------------------------------------------------------------------------
def quant_tensor(t, bits=8):
    """Uniform `bits`-bit quantisation of a tensor on the best of several affine grids: one (min, step) pair for the whole tensor
    (kept in fp32), and one pair per slice along every axis whose pair count stays below 2 % of the element count (stored as
    fp16, as they are what gets transmitted).  The grid with the smallest mean absolute reconstruction error wins (first one on
    ties).  Returns ({'quant': uint8 codes, 'min', 'scale'}, reconstructed tensor)."""
    top = 2 ** bits - 1
    grids = [(t.min(), (t.max() - t.min()) / top)]
    for axis in range(t.dim()):
        lo, hi = t.amin(axis, keepdim=True), t.amax(axis, keepdim=True)
        if lo.nelement() / t.nelement() < 0.02:
            grids.append((lo.to(torch.float16), ((hi - lo) / top).to(torch.float16)))
    best = None
    for lo, step in grids:
        lo_e, step_e = lo.expand_as(t), step.expand_as(t)
        codes = ((t - lo_e) / step_e).round().clamp(0, top)
        rebuilt = lo_e + step_e * codes
        err = (t - rebuilt).abs().mean()
        if best is None or err < best[0]:
            best = (err, codes, rebuilt, lo, step)
    _, codes, rebuilt, lo, step = best
    return {"quant": codes.to(torch.uint8), "min": lo, "scale": step}, rebuilt


def dequant_tensor(quant_t):
    q, tmin, scale = quant_t["quant"], quant_t["min"], quant_t["scale"]
    return tmin.expand_as(q) + scale.expand_as(q) * q


# ----------------------------------------------------------------------------------------------------------------------
# distributed helpers                                                               reference hnerv_utils.py:213-248
# ----------------------------------------------------------------------------------------------------------------------
def all_reduce(tensors, average=True):
    for tensor in tensors:
        dist.all_reduce(tensor, async_op=False)
    if average:
        world_size = dist.get_world_size()
        for tensor in tensors:
            tensor.mul_(1.0 / world_size)
    return tensors


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def worker_init_fn(worker_id):
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def RoundTensor(x, num=2, group_str=False):                                           # reference hnerv_utils.py:277-289
    if group_str:
        return "/".join(",".join(str(round(ele, num)) for ele in x[i].tolist()) for i in range(x.size(0)))
    return ",".join(str(round(ele, num)) for ele in x.flatten().tolist())


# ----------------------------------------------------------------------------------------------------------------------
# LR schedule                                                                       reference hnerv_utils.py:292-322
# ----------------------------------------------------------------------------------------------------------------------
def adjust_lr(optimizer, cur_epoch, cur_iter, args):
    if "hybrid" in args.lr_type:
        up_ratio, up_pow, down_pow, min_lr, final_lr = [float(x) for x in args.lr_type.split("_")[1:]]
        if cur_epoch < up_ratio:
            lr_mult = min_lr + (1.0 - min_lr) * (cur_epoch / up_ratio) ** up_pow
        else:
            lr_mult = 1 - (1 - final_lr) * ((cur_epoch - up_ratio) / (1.0 - up_ratio)) ** down_pow
    elif "cosine" in args.lr_type:
        up_ratio, up_pow, min_lr = [float(x) for x in args.lr_type.split("_")[1:]]
        if cur_epoch < up_ratio:
            lr_mult = min_lr + (1.0 - min_lr) * (cur_epoch / up_ratio) ** up_pow
        else:
            lr_mult = 0.5 * (math.cos(math.pi * (cur_epoch - up_ratio) / (1 - up_ratio)) + 1.0)
    elif "enerv_sch" in args.lr_type:
        all_iter = args.epochs * args.full_data_length
        now_iter = cur_epoch * args.full_data_length + cur_iter
        if now_iter < all_iter * 0.2:
            lr_mult = 0.1 + 0.9 * now_iter / (all_iter * 0.2)
        else:
            lr_mult = 0.5 * (math.cos(math.pi * (now_iter - all_iter * 0.2) / (all_iter - all_iter * 0.2)) + 1.0)
    else:
        raise NotImplementedError
    for param_group in optimizer.param_groups:
        param_group["lr"] = args.lr * lr_mult
    return args.lr * lr_mult


# ----------------------------------------------------------------------------------------------------------------------
# loss and metrics                                                           reference hnerv_utils.py:335-419
# ----------------------------------------------------------------------------------------------------------------------
def loss_fn(pred, target, loss_type="L2", batch_average=True):
    """Value + gradient come from one fused HIP call (bnerv_loss_fwd_bwd).  Supported: the variants the recipes use
    (Fusion10_freq for the boost models) plus L1, L2, L1_freq, Fusion10/11/12."""
    loss, stats = ops.loss_with_stats(pred, target, loss_type)
    if batch_average:
        return loss
    # per-sample values (no gradient path): stats[:,0]
    return stats[:, 0]


def psnr_fn_device(output, gt):
    """Per-sample PSNR as a DEVICE tensor (no host sync) -- what the train loop accumulates."""
    return ops.psnr(output, gt)


def psnr_fn_single(output, gt):
    return psnr_fn_device(output, gt).cpu()


def psnr_fn_batch(output_list, gt):
    return torch.stack([psnr_fn_single(o.detach(), gt.detach()) for o in output_list], 0).cpu()


def msssim_fn_single(output, gt):
    return ops.msssim(output.float().detach(), gt.detach()).cpu()


def msssim_fn_batch(output_list, gt):
    return torch.stack([msssim_fn_single(o.detach(), gt.detach()) for o in output_list], 0).cpu()
