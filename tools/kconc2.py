"""Experiment: do SMALL independent kernels on parallel hipGraph branches overlap?  (45x80 / 180x320 convs: 36..230 blocks.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
def bench(fn, n=20):
    fn(2); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s): fn(n)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
for (C, H, W) in ((30, 45, 80), (15, 90, 160), (12, 180, 320), (12, 360, 640)):
    B = 1
    x, g_ = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / 10; b = torch.randn(C, device=dev)
    sc, sh = torch.randn(B, C, device=dev) * .1, torch.randn(B, C, device=dev) * .1
    o1 = torch.empty_like(x); dw, db = torch.empty_like(w), torch.empty_like(b)
    def conv(): ops._conv(x, w, b, o1, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh)
    def wg(): ops._wgrad(x, g_, dw, db, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh)
    def serial(n):
        for _ in range(n): conv(); wg()
    def forked(n):
        main = torch.cuda.current_stream()
        for _ in range(n):
            side.wait_stream(main)
            with torch.cuda.stream(side): wg()
            conv()
            main.wait_stream(side)
    def only_c(n):
        for _ in range(n): conv()
    def only_w(n):
        for _ in range(n): wg()
    print(f"{C}ch {H}x{W}: conv {bench(only_c):6.2f}  wgrad {bench(only_w):6.2f}  serial pair {bench(serial):6.2f}  forked pair {bench(forked):6.2f} us")
