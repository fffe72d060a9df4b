"""Experiment: do parallel hipGraph branches overlap?  A: big conv, t: tiny reduce (depends on A), B: big conv (independent of t).
serial graph  A -> t -> B   vs   forked graph  A -> {t on a side stream | B} -> join ; also two big independent convs forked."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
B, C, H, W = 1, 12, 720, 1280
x = torch.randn(B, C, H, W, device=dev); w = torch.randn(C, C, 3, 3, device=dev) / 10; b = torch.randn(C, device=dev)
sc, sh = torch.randn(B, C, device=dev) * .1, torch.randn(B, C, device=dev) * .1
o1, o2 = torch.empty_like(x), torch.empty_like(x)
slabs = torch.randn(768, 1308, device=dev); red = torch.empty(1308, device=dev)
lib = L.load()
def conv(o): ops._conv(x, w, b, o, B=B, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh)
def tiny(): L.check(lib.bnerv_reduce_slabs(L.stream(), L.ptr(slabs), 768, 1308, L.ptr(red)), "reduce")
side = torch.cuda.Stream()
def serial(n):
    for _ in range(n): conv(o1); tiny(); conv(o2)
def forked(n):
    main = torch.cuda.current_stream()
    for _ in range(n):
        conv(o1)
        side.wait_stream(main)
        with torch.cuda.stream(side): tiny()
        conv(o2)
        main.wait_stream(side)
def hosted(n):
    for _ in range(n):
        conv(o1)
        L.check(lib.bnerv_reduce_slabs_deferred(L.ptr(slabs), 768, 1308, L.ptr(red)), "defer")
        conv(o2)
def two_serial(n):
    for _ in range(n): conv(o1); conv(o2)
def two_forked(n):
    main = torch.cuda.current_stream()
    for _ in range(n):
        side.wait_stream(main)
        with torch.cuda.stream(side): conv(o1)
        conv(o2)
        main.wait_stream(side)
def only(n):
    for _ in range(n): conv(o1)
def bench(fn, n=20, graph=True):
    fn(2); torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s): fn(n)
        run = g.replay
    else:
        run = lambda: fn(n)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
for name, fn in (("conv only", only), ("A,t,B serial", serial), ("A,{t|B} forked", forked), ("A,B hosting t", hosted), ("2 convs serial", two_serial), ("2 convs forked", two_forked)):
    print(f"{name:18s} graph {bench(fn):8.2f} us/iter   eager {bench(fn, graph=False):8.2f} us/iter")
