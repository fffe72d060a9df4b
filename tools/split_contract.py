"""fp32 contract of the wide split kernels (conv_bfw / wgrad_bfw, csrc/convbf.hip, csrc/wgrad.hip): every f32 product is rebuilt from
THREE bf16 pieces per operand and six partial products with f32 accumulation (bf16x6).  This checker measures, per output element,
    |kernel - float64 reference| / sum_k |a_k| |b_k|
for forward convolutions (plain, affine prologue, residual epilogue, PixelShuffle(2) up-conv), data gradients and weight gradients at
38 -> 38, 55 -> 55 and 12 -> 48 channels, and exits 0 iff the maximum over all of them stays below BOUND = 3.5e-7 -- the level of an f32
FMA chain of this length (measured 1e-8 .. 8e-8 for bf16x6; the two-piece mode bf16x3 sits at ~2e-6 and must FAIL this check:
tests/test_gpu_ops.py runs it both ways, so a silent downgrade of the arithmetic cannot pass the suite).
usage: python tools/split_contract.py        (BNERV_SPLIT_WIDE selects the mode under test; checker tool, torch fp64 is the reference)"""
import math, os, sys, torch
os.environ.setdefault("BNERV_SPLIT_WIDE_MIN_TILES", "1")
os.environ.setdefault("BNERV_SMALL", "0")          # keep small test images on the split kernels (not the low-resolution family)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from boosting_nerv_amd import ops, _lib as L

dev = torch.device("cuda:0")
BOUND = 3.5e-7
worst = {}


def ratio(name, got, ref64, bound64):
    r = ((got.double() - ref64).abs() / bound64.clamp_min(1e-30)).max().item()
    worst[name] = max(worst.get(name, 0.0), r)


g = torch.Generator(device="cpu").manual_seed(11)
for (B, Ci, Co, H, W, s) in ((1, 38, 38, 40, 96, 1), (2, 55, 55, 17, 68, 1), (1, 12, 48, 24, 64, 2), (1, 46, 184, 16, 64, 2)):
    x = torch.randn(B, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    x64, w64, b64 = x.double(), w.double(), b.double()
    # forward (+ PixelShuffle)
    out = ops.conv2d_ps(x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True), s)
    ref = F.conv2d(x64, w64, b64, padding=1)
    bnd = F.conv2d(x64.abs(), w64.abs(), b64.abs(), padding=1)
    if s > 1:
        ref, bnd = F.pixel_shuffle(ref, s), F.pixel_shuffle(bnd, s)
    ratio(f"fwd {Ci}->{Co} s{s}", out.detach(), ref, bnd)
    # data and weight gradients of the same layer
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    o = ops.conv2d_ps(xg, wg, bg, s)
    cot = torch.randn(o.shape, generator=g).to(dev)
    dx, dw, db = torch.autograd.grad(o, [xg, wg, bg], cot)
    c64 = cot.double()
    if s > 1:
        c64 = F.pixel_unshuffle(c64, s)
    ratio(f"dgrad {Ci}->{Co} s{s}", dx, F.conv_transpose2d(c64, w64, padding=1), F.conv_transpose2d(c64.abs(), w64.abs(), padding=1))
    dw_ref = torch.nn.grad.conv2d_weight(x64, w64.shape, c64, padding=1)
    dw_bnd = torch.nn.grad.conv2d_weight(x64.abs(), w64.shape, c64.abs(), padding=1)
    ratio(f"wgrad {Ci}->{Co} s{s}", dw, dw_ref, dw_bnd)
    if s == 1 and Ci == Co:
        # TAT modes of the same kernels: affine prologue -> bias, affine prologue -> residual
        sc, sh = (torch.randn(B, Ci, generator=g) * 0.3).to(dev), (torch.randn(B, Ci, generator=g) * 0.3).to(dev)
        y0 = torch.randn(B, Co, H, W, generator=g).to(dev)
        a64 = x64 * (1 + sc.double()[:, :, None, None]) + sh.double()[:, :, None, None]
        for ep, aux in ((L.EP_BIAS, None), (L.EP_BIAS_RES, y0)):
            o2 = torch.empty(B, Co, H, W, device=dev)
            ops._conv(x, w, b, o2, B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=ep, scale=sc, shift=sh, aux0=aux)
            r2 = F.conv2d(a64, w64, b64, padding=1) + (aux.double() if aux is not None else 0)
            b2 = F.conv2d(a64.abs(), w64.abs(), b64.abs(), padding=1) + (aux.double().abs() if aux is not None else 0)
            ratio(f"affine->{'res' if aux is not None else 'bias'} {Ci}->{Co}", o2, r2, b2)
torch.cuda.synchronize()
mode = os.environ.get("BNERV_SPLIT_WIDE", "bf16x6")
for k, v in worst.items():
    print(f"{mode:7s} {k:24s} max |err| / sum|a||b| = {v:.3e}")
m = max(worst.values())
print(f"{mode}: worst {m:.3e} against the bound {BOUND:.1e}: {'within' if m <= BOUND else 'OUTSIDE'} the f32 contract")
from boosting_nerv_amd.runtime import hard_exit  # noqa: E402
hard_exit(0 if m <= BOUND else 1)
