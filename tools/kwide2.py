"""Wide layers of C3 at full size through the C-ABI (HIP events): the f32 lean2 kernel against the wide split kernel.
usage: BNERV_SPLIT_WIDE=off|bf16x6|bf16x3 python tools/kwide2.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ops._flush_deferred()
    return e0.elapsed_time(e1) * 1e3 / reps


rn = lambda *s: torch.randn(*s, device=dev)
print(f"mode BNERV_SPLIT_WIDE={os.environ.get('BNERV_SPLIT_WIDE', '(default)')}")
print(f"{'kernel':58s} {'us':>9s} {'TFLOP/s':>9s}")
for (C, H, W) in ((38, 1080, 1920), (46, 540, 960), (55, 270, 480)):
    B = 1
    x, y0, h, gp, c0, dout = (rn(B, C, H, W) for _ in range(6))
    w = rn(C, C, 3, 3) / 10; b = rn(C); sc, sh = rn(B, C) * 0.1, rn(B, C) * 0.1
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    kw = dict(B=B, Cin=C, Cout=C, H=H, W=W, k=3)
    fl = 2.0 * C * C * 9 * H * W
    for name, fn in {
        "K2s affine->gelu,gelu'": lambda: ops._conv(x, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=out2, **kw),
        "K3s affine->res": lambda: ops._conv(h, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sc, shift=sh, aux0=y0, **kw),
        "K1 plain->sin,cos": lambda: ops._conv(x, w, b, out, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out2=out2, **kw),
        "dK3s ->dgelu saved": lambda: ops._conv(dout, w, None, out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=gp, aux1=h, scale=sc, defer=True, **kw),
        "dK2s ->dsin": lambda: ops._conv(dout, w, None, out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=y0, aux1=dout, aux2=c0, scale=sc, defer=True, **kw),
    }.items():
        t = timeit(fn)
        print(f"{name + f' {C}->{C} @{H}x{W}':58s} {t:9.1f} {fl / t / 1e6:9.2f}")
