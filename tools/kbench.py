"""Micro-benchmark of the hot kernels at the C1 shapes through the C-ABI (HIP events on the launch stream).
usage: python tools/kbench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    """us per launch: the launches are replayed as a captured hipGraph (bench._time_launches), so the figure is kernel time and not
    the host time of the Python-driven ctypes loop (which is of the same size as these kernels)."""
    import bench
    return bench._time_launches(fn, reps) * 1e6


def rnd(*s):
    return torch.randn(*s, device=dev)


rows = []
for (H, W) in ((720, 1280), (360, 640), (180, 320)):
    B, C = 1, 12
    x, y0, v, g = rnd(B, C, H, W), rnd(B, C, H, W), rnd(B, C, H, W), rnd(B, C, H, W)
    w = rnd(C, C, 3, 3) / 10
    b = rnd(C)
    sc, sh = rnd(B, C) * 0.1, rnd(B, C) * 0.1
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    tiles = L.load().bnerv_conv_tiles(H, W)
    part = torch.empty(tiles, B, 2, C, device=dev)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    fl = 2.0 * C * C * 9 * H * W
    cases = {
        "fwd affine->bias (K2)": lambda: ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh),
        "fwd gelu-affine->res (K3)": lambda: ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_GELU_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sc, shift=sh, aux0=y0),
        "fwd affine->gelu,gelu' (K2s)": lambda: ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=out2),
        "fwd affine->res (K3s)": lambda: ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sc, shift=sh, aux0=y0),
        "dgrad ->dgelu saved (K3s bwd)": lambda: ops._conv(g, w, None, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=v, aux1=y0, scale=sc, partial=part),
        "fwd plain->sin s1 (K1)": lambda: ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out2=out2),
        "dgrad ->dgelu (K3 bwd)": lambda: ops._conv(g, w, None, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU, transposed=1, aux0=v, scale=sc, partial=part),
        "dgrad ->dsin (K2 bwd)": lambda: ops._conv(g, w, None, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=y0, aux1=g, aux2=v, scale=sc, partial=part),
        "wgrad plain": lambda: ops._wgrad(x, g, dw, db, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE),
        "wgrad gelu-affine": lambda: ops._wgrad(x, g, dw, db, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_GELU_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh),
    }
    for name, fn in cases.items():
        t = timeit(fn)
        rows.append((f"{name} 12->12 @{H}x{W}", t, fl / t / 1e6))
# up-conv 12->48 (+PS2+sin) at 360x640 and its data/weight gradients
B, Ci, Co, H, W = 1, 12, 48, 360, 640
x = rnd(B, Ci, H, W)
w = rnd(Co, Ci, 3, 3) / 10
b = rnd(Co)
out, out2 = torch.empty(B, 12, 2 * H, 2 * W, device=dev), torch.empty(B, 12, 2 * H, 2 * W, device=dev)
du = rnd(B, 12, 2 * H, 2 * W)
dx = torch.empty_like(x)
dw, db = torch.empty_like(w), torch.empty_like(b)
fl = 2.0 * Ci * Co * 9 * H * W
for name, fn in {
    "fwd upconv 12->48 +PS2+sin": lambda: ops._conv(x, w, b, out, B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out_s=2, out2=out2),
    "dgrad upconv 48->12 (unshuffle2)": lambda: ops._conv(du, w, None, dx, B=B, Cin=Co, Cout=Ci, H=H, W=W, k=3, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=2, transposed=1),
    "wgrad upconv 12->48 (unshuffle2)": lambda: ops._wgrad(x, du, dw, db, B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=2),
}.items():
    t = timeit(fn)
    rows.append((f"{name} @{H}x{W}", t, fl / t / 1e6))
# low-res stage-0 data gradient: Cin_d = 750 at 9x16
B, Ci, Co, H, W = 1, 30, 750, 9, 16
w = rnd(Co, Ci, 3, 3) / 10
du = rnd(B, 30, 45, 80)
dx = torch.empty(B, Ci, H, W, device=dev)
t = timeit(lambda: ops._conv(du, w, None, dx, B=B, Cin=Co, Cout=Ci, H=H, W=W, k=3, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=5, transposed=1))
rows.append(("dgrad stage0 750->30 (unshuffle5) @9x16", t, 2.0 * Ci * Co * 9 * H * W / t / 1e6))
# low-resolution stages of C1: latency-bound (tiny grids)
for (B, Ci, Co, H, W, s_) in ((1, 30, 750, 9, 16, 5), (1, 30, 60, 45, 80, 2), (1, 15, 48, 90, 160, 2)):
    x = rnd(B, Ci, H, W); w = rnd(Co, Ci, 3, 3) / 10; b = rnd(Co)
    Cf = Co // (s_ * s_)
    out, out2 = torch.empty(B, Cf, s_ * H, s_ * W, device=dev), torch.empty(B, Cf, s_ * H, s_ * W, device=dev)
    du = rnd(B, Cf, s_ * H, s_ * W); dx = torch.empty_like(x); dw, db = torch.empty_like(w), torch.empty_like(b)
    fl = 2.0 * Ci * Co * 9 * H * W
    for name, fn in {
        f"fwd upconv {Ci}->{Co} +PS{s_}+sin": lambda: ops._conv(x, w, b, out, B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out_s=s_, out2=out2),
        f"dgrad upconv {Co}->{Ci} (unshuffle{s_})": lambda: ops._conv(du, w, None, dx, B=B, Cin=Co, Cout=Ci, H=H, W=W, k=3, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=s_, transposed=1),
        f"wgrad upconv {Ci}->{Co} (unshuffle{s_})": lambda: ops._wgrad(x, du, dw, db, B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=s_),
    }.items():
        t = timeit(fn)
        rows.append((f"{name} @{H}x{W}", t, fl / t / 1e6))
for (C, H, W) in ((30, 45, 80), (15, 90, 160)):
    B = 1
    x, y0, v, g = rnd(B, C, H, W), rnd(B, C, H, W), rnd(B, C, H, W), rnd(B, C, H, W)
    w = rnd(C, C, 3, 3) / 10; b = rnd(C); sc, sh = rnd(B, C) * 0.1, rnd(B, C) * 0.1
    out = torch.empty_like(x); dw, db = torch.empty_like(w), torch.empty_like(b)
    part = torch.empty(L.load().bnerv_conv_tiles(H, W), B, 2, C, device=dev)
    fl = 2.0 * C * C * 9 * H * W
    for name, fn in {
        "fwd affine->bias": lambda: ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh),
        "dgrad ->dgelu": lambda: ops._conv(g, w, None, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU, transposed=1, aux0=v, scale=sc, partial=part),
        "wgrad gelu-affine": lambda: ops._wgrad(x, g, dw, db, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_GELU_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh),
    }.items():
        t = timeit(fn)
        rows.append((f"{name} {C}->{C} @{H}x{W}", t, fl / t / 1e6))
print(f"{'kernel':58s} {'us':>9s} {'TFLOP/s':>9s}")
for n, t, tf in rows:
    print(f"{n:58s} {t:9.1f} {tf:9.2f}")
