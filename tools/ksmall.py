"""Per-launch times of the LOW-RESOLUTION conv launches of the C1 step (<= 180x320: forward up-convs and TAT convs, and their backward
pairs), each replayed as a captured graph of its own launches (bench._time_launches).  usage: python tools/ksmall.py [reps]
BNERV_LIB=<variant .so> selects an ablation build (csrc/convs_body.h BNERV_ABLS)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rn = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc
rows = []


def t_us(fn):
    return bench._time_launches(fn, reps) * 1e6


def tat(C, H, W):
    B = 1
    y0, h, gp, c0, dout = (rn(B, C, H, W) for _ in range(5))
    w, b = rn(C, C, 3, 3, sc=0.1), rn(C)
    sc, sh = rn(B, C, sc=0.1), rn(B, C, sc=0.1)
    out, out2 = torch.empty_like(y0), torch.empty_like(y0)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    kw = dict(B=B, Cin=C, Cout=C, H=H, W=W, k=3)
    tag = f"{C}->{C} @{H}x{W}"
    rows.append((f"K2s conv0 fwd {tag}", t_us(lambda: ops._conv(y0, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=out2, **kw))))
    rows.append((f"K3s conv1 fwd {tag}", t_us(lambda: ops._conv(h, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sc, shift=sh, aux0=y0, **kw))))
    rows.append((f"pair dK3s {tag}", t_us(lambda: ops._wgrad_conv_pair(
        dict(x=h, g=dout, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
        dict(x=dout, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=gp, aux1=h, scale=sc, **kw)))))
    rows.append((f"pair dK2s {tag}", t_us(lambda: ops._wgrad_conv_pair(
        dict(x=y0, g=dout, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
        dict(x=dout, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=y0, aux1=gp, aux2=c0, scale=sc, **kw)))))
    ops._flush_deferred()


def up(Cin, Ct, H, W, s):
    B = 1
    x = rn(B, Cin, H, W)
    w, b = rn(Ct, Cin, 3, 3, sc=0.1), rn(Ct)
    Cc = Ct // (s * s)
    y0, c0 = torch.empty(B, Cc, H * s, W * s, device=dev), torch.empty(B, Cc, H * s, W * s, device=dev)
    du = rn(B, Cc, H * s, W * s)
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    tag = f"{Cin}->{Ct} s{s} @{H}x{W}"
    rows.append((f"K1 up-conv fwd {tag}", t_us(lambda: ops._conv(x, w, b, y0, B=B, Cin=Cin, Cout=Ct, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out_s=s, out2=c0))))
    rows.append((f"pair up-conv bwd {tag}", t_us(lambda: ops._wgrad_conv_pair(
        dict(x=x, g=du, dw=dw, db=db, B=B, Cin=Cin, Cout=Ct, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=s),
        dict(x=du, w=w, bias=None, out=dx, B=B, Cin=Ct, Cout=Cin, H=H, W=W, k=3, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=s, transposed=1)))))
    ops._flush_deferred()


up(30, 750, 9, 16, 5)
tat(30, 45, 80)
up(30, 60, 45, 80, 2)
tat(15, 90, 160)
up(15, 48, 90, 160, 2)
tat(12, 180, 320)
up(12, 12, 180, 320, 1)
tat(12, 360, 640)
up(12, 48, 180, 320, 2)
tot = 0.0
for name, t in rows:
    print(f"{name:44s} {t:7.2f} us")
    tot += t
print(f"{'sum':44s} {tot:7.2f} us")
