"""Fixed per-launch cost of the conv path: the same kernel on a 1-tile / few-tile image (HIP events, back-to-back launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
C = 12
for (H, W) in ((8, 32), (64, 128), (128, 256), (256, 512), (360, 640), (512, 1024), (720, 1280), (1440, 1280), (1440, 2560)):
    x = torch.randn(1, C, H, W, device=dev); w = torch.randn(C, C, 3, 3, device=dev) / 10; b = torch.randn(C, device=dev)
    sc, sh = torch.randn(1, C, device=dev) * .1, torch.randn(1, C, device=dev) * .1
    out = torch.empty_like(x)
    t = timeit(lambda: ops._conv(x, w, b, out, B=1, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh))
    tiles = ((H + 7) // 8) * ((W + 31) // 32)
    print(f"{H:5d}x{W:<5d} tiles {tiles:6d}  {t:8.2f} us   {2.0*C*C*9*H*W/t/1e6:7.2f} TFLOP/s   tiles/1024 = {tiles/1024:.2f}")
a = torch.zeros(64, device=dev)
print("torch tiny add_ :", timeit(lambda: a.add_(1.0)), "us")
