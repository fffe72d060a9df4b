"""Run ONE hot kernel a few times (for rocprofv3 --pmc runs).  usage: kone.py conv|conv_k2s|wgrad|conv38|wgrad38|pair_dk3s|pair_dk2s|pair_dk1|tat_fused [reps]
pair_*: the paired backward launches of a TAT block (weight gradient | data gradient in one grid) exactly as ops._tat_backward issues them.
conv / wgrad: the 12->12 3x3 layer at 720x1280 (C1; suffix _1080: at 1080x1920, C4); conv38 / wgrad38: the 38->38 3x3 layer at 1080x1920 (C3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, C, H, W = (1, 38, 1080, 1920) if "38" in which else ((1, 12, 1080, 1920) if which.endswith("_1080") else (1, 12, 720, 1280))
if which.endswith("_1080"):        # C4's final stage: the 12-channel kernels at 1080x1920
    which = which[:-5]
if which.startswith("pair38"):    # C3's final stage: the backward pair of a 38-channel TAT conv (two launches: wide weight gradient, wide data gradient)
    which = "pair" + which[6:]
x, g = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
w, b = torch.randn(C, C, 3, 3, device=dev) / 10, torch.randn(C, device=dev)
sc, sh = torch.randn(B, C, device=dev) * 0.1, torch.randn(B, C, device=dev) * 0.1
out = torch.empty_like(x); dw, db = torch.empty_like(w), torch.empty_like(b)
h, gp, c0 = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
kw = dict(B=B, Cin=C, Cout=C, H=H, W=W, k=3)
for _ in range(reps):
    if which == "pair_dk3s":      # conv1 backward: weight gradient (affine prologue) | conv^T -> dgelu(saved) + channel sums
        ops._wgrad_conv_pair(dict(x=h, g=g, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=gp, aux1=h, scale=sc, **kw))
    elif which == "pair_dk2s":    # conv0 backward: weight gradient (affine prologue) | conv^T -> dsin + channel sums
        ops._wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=x, aux1=gp, aux2=c0, scale=sc, **kw))
    elif which == "pair_dk1":     # block conv backward: weight gradient (plain) | conv^T
        ops._wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=1, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_PLAIN, transposed=1, **kw))
    elif which == "conv38_k2s":     # wide layer, TAT conv0 forward: the wide split kernel (or lean2 with BNERV_SPLIT_WIDE=off)
        ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=g)
    elif which == "conv_k2s":       # the TAT conv0 forward the train step launches: affine -> conv -> bias -> gelu, gelu'
        ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=g)
    elif which.startswith("conv"):
        ops._conv(x, w, b, out, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh)
    else:
        ops._wgrad(x, g, dw, db, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
ops._flush_deferred()
torch.cuda.synchronize()
