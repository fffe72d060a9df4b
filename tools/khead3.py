"""C3 anomalies, measured one by one (graph-replayed launches): HNeRV's 3x3 head (38 -> 3) weight gradient in today's form against the
swapped-roles form (x <-> g: M = the 38 input channels, N = 3 couts x 9 taps), the ConvNeXt pointwise weight gradients (k = 1), the
depthwise 7x7 weight gradient.  usage: python tools/khead3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
rn = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc
t_us = lambda fn: bench._time_launches(fn, 10) * 1e6

B, Ci, Co, H, W = 1, 38, 3, 1080, 1920
x, g, img = rn(B, Ci, H, W), rn(B, Co, H, W), torch.rand(B, Co, H, W, device=dev)
w = rn(Co, Ci, 3, 3, sc=0.1)
dw, db = torch.empty_like(w), torch.empty(Co, device=dev)
cur = lambda: ops._wgrad(x, g, dw, db, B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_TANHGRAD, gaux=img)
cur(); ops._flush_deferred(); torch.cuda.synchronize()
print(f"head 3x3 wgrad, today (tanh-grad g, 3 of 16 rows): {t_us(cur):8.2f} us")
t = 2 * img - 1
gt = (g * 0.5 * (1 - t * t)).contiguous()
dw2, db2 = torch.empty(Ci, Co, 3, 3, device=dev), torch.empty(Ci, device=dev)
sw = lambda: ops._wgrad(gt, x, dw2, db2, B=B, Cin=Co, Cout=Ci, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
sw(); ops._flush_deferred(); torch.cuda.synchronize()
ref = dw2.permute(1, 0, 2, 3).flip(2, 3)
err = (ref - dw).abs().max().item() / dw.abs().max().item()
print(f"head 3x3 wgrad, swapped roles (M = 38 inputs, N = 27 columns): {t_us(sw):8.2f} us   rel. difference to today's {err:.2e}")
dx = torch.empty_like(x)
d1 = lambda: ops._conv(g, w, None, dx, B=B, Cin=Co, Cout=Ci, H=H, W=W, k=3, in_mode=L.IN_TANHGRAD, ep_mode=L.EP_PLAIN, transposed=1, aux0=img)
d2 = lambda: ops._conv(gt, w, None, dx, B=B, Cin=Co, Cout=Ci, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_PLAIN, transposed=1)
print(f"head 3x3 dgrad, today (tanh-grad prologue): {t_us(d1):8.2f} us ; plain input: {t_us(d2):8.2f} us")

for (C, Hh, Ww) in ((64, 216, 384), (64, 72, 128), (64, 36, 64)):
    xx, dh, dout, gb = rn(B, C, Hh, Ww), rn(B, 4 * C, Hh, Ww), rn(B, C, Hh, Ww), rn(B, 4 * C, Hh, Ww)
    dw1, db1 = torch.empty(4 * C, C, 1, 1, device=dev), torch.empty(4 * C, device=dev)
    S, tt = torch.empty(C, 4 * C, 1, 1, device=dev), torch.empty(C, device=dev)
    a = lambda: ops._wgrad(xx, dh, dw1, db1, B=B, Cin=C, Cout=4 * C, H=Hh, W=Ww, k=1, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
    b = lambda: ops._wgrad(gb, dout, S, tt, B=B, Cin=4 * C, Cout=C, H=Hh, W=Ww, k=1, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
    ta, tb = t_us(a), t_us(b)
    r1 = torch.einsum("bohw,bihw->oi", dh.double(), xx.double())
    r2 = torch.einsum("bohw,bihw->oi", dout.double(), gb.double())
    e1 = (dw1.flatten(1).double() - r1).abs().max().item() / r1.abs().max().item()
    e2 = (S.flatten(1).double() - r2).abs().max().item() / r2.abs().max().item()
    eb = (db1.double() - dh.double().sum((0, 2, 3))).abs().max().item() / dh.double().sum((0, 2, 3)).abs().max().item()
    print(f"pointwise wgrad {C}->{4 * C} @{Hh}x{Ww}: {ta:8.2f} us (err {e1:.1e}, db {eb:.1e}) ; {4 * C}->{C}: {tb:8.2f} us (err {e2:.1e})")
    wd = rn(C, 1, 7, 7)
    y = ops.dwconv(xx.requires_grad_(True), wd.requires_grad_(True), None)
    gy = rn(B, C, Hh, Ww)
    lib = L.load()
    nbytes = lib.bnerv_dwconv_wgrad_ws_bytes(B, C, Hh, Ww, 7)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dwb = torch.empty(C, 50, device=dev)
    f = lambda: L.check(lib.bnerv_dwconv_wgrad(L.stream(), L.ptr(xx), L.ptr(gy), L.ptr(dwb), L.ptr(ws), nbytes, B, C, Hh, Ww, 7, None), "dw")
    tdw = t_us(f)
    refw = torch.nn.grad.conv2d_weight(xx.detach().double(), (C, 1, 7, 7), gy.double(), padding=3, groups=C)
    ew = (dwb[:, :49].reshape(C, 1, 7, 7).double() - refw).abs().max().item() / refw.abs().max().item()
    print(f"depthwise 7x7 wgrad {C} @{Hh}x{Ww}: {tdw:8.2f} us (incl. its slab reduction; err {ew:.1e})")
