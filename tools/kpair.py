"""Event-timed back-to-back launches of one 12-channel backward pair at 720x1280 (or HxW given), deferred reductions flushed OUTSIDE the timed region.
usage: python tools/kpair.py pair_dk2s|pair_dk3s|pair_dk1 [reps=50] [H W]
KPAIR_GRAPH=n: the timed loop is a captured graph of n (pair, flush) launches, replayed reps / n times -- no host launch rate in the figure
(the eager loop issues two ctypes calls per iteration; at ~75 us per iteration it can be the host that is timed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "pair_dk2s"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (720, 1280)
B, C = 1, 12
x, g = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) / 10
sc, sh = torch.randn(B, C, device=dev) * 0.1, torch.randn(B, C, device=dev) * 0.1
out = torch.empty_like(x); dw, db = torch.empty_like(w), torch.empty(C, device=dev)
h, gp, c0 = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
kw = dict(B=B, Cin=C, Cout=C, H=H, W=W, k=3)
def one():
    if which == "pair_dk3s":
        ops._wgrad_conv_pair(dict(x=h, g=g, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=gp, aux1=h, scale=sc, **kw))
    elif which == "pair_dk2s":
        ops._wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=x, aux1=gp, aux2=c0, scale=sc, **kw))
    else:
        ops._wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=1, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_PLAIN, transposed=1, **kw))
for _ in range(5):
    one()
ops._flush_deferred(); torch.cuda.synchronize()
NG = int(os.environ.get("KPAIR_GRAPH", "0"))
graph = None
if NG:
    graph = torch.cuda.CUDAGraph()
    with L.graph_capture(graph):
        for _ in range(NG):
            one()
            ops._flush_deferred()
    reps = max(1, reps // NG)
for trial in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if graph is not None:
            graph.replay()
            continue
        one()
        ops._flush_deferred()      # (a small launch per pair: its cost is in both arms; keeps the queue from growing)
    e1.record(); torch.cuda.synchronize()
    print(f"{which} {H}x{W} FOLD={os.environ.get('BNERV_PAIR_FOLD','default')} FUSED={os.environ.get('BNERV_PAIR_FUSED','default')} graph={NG}: {e0.elapsed_time(e1) / (reps * max(NG, 1)) * 1e3:.1f} us per (pair + flush)")
