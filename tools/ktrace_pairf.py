"""Phase timeline of pair_fused_kernel (csrc/pairf_body.h) from a -DBNERV_TRACE build of the library (debug variant; the shipped one has no tracing).
usage: BNERV_LIB=_variants/lib_trace.so BNERV_PAIR_FUSED=8 python tools/ktrace_pairf.py [pair_dk2s|pair_dk3s|pair_dk1]
Stamps per (block, wave, tile): 0 tile top | 1 data-gradient K loop done | 2 past barrier B | 3 weight-gradient K loop done | 4 epilogue done |
5 next tile's DMA + x loads complete | 6 past barrier A | 7 x committed.  s_memtime ticks = 10 ns (100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "pair_dk2s"
B, Cc, H, W = 1, 12, 720, 1280
x, g = torch.randn(B, Cc, H, W, device=dev), torch.randn(B, Cc, H, W, device=dev)
w, b = torch.randn(Cc, Cc, 3, 3, device=dev) / 10, torch.randn(Cc, device=dev)
sc, sh = torch.randn(B, Cc, device=dev) * 0.1, torch.randn(B, Cc, device=dev) * 0.1
out = torch.empty_like(x); dw, db = torch.empty_like(w), torch.empty_like(b)
h, gp, c0 = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
kw = dict(B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(5):
    if rep == 4:
        torch.cuda.synchronize(); e0.record()
    if which == "pair_dk3s":
        ops._wgrad_conv_pair(dict(x=h, g=g, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=gp, aux1=h, scale=sc, **kw))
    elif which == "pair_dk2s":
        ops._wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=x, aux1=gp, aux2=c0, scale=sc, **kw))
    else:
        ops._wgrad_conv_pair(dict(x=x, g=g, dw=dw, db=db, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE, g_s=1, **kw),
                             dict(x=g, w=w, bias=None, out=out, in_mode=L.IN_PLAIN, ep_mode=L.EP_PLAIN, transposed=1, **kw))
e1.record()
torch.cuda.synchronize()
print(f"{which} @720x1280: {e0.elapsed_time(e1) * 1e3:.1f} us (traced build, one launch, events)")
ops._flush_deferred()
torch.cuda.synchronize()
lib = L.load()
buf = np.zeros(1024 * 4 * 6 * 8, dtype=np.uint64)
fn = lib.bnerv_debug_trace_read_pairf
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf.reshape(1024, 4, 6, 8).astype(np.int64)
names = ["dgrad K loop (+DMA issue, flush)", "barrier B wait", "wgrad K loop (+aux, x loads)", "epilogue", "wait DMA / x", "barrier A wait", "commit x"]
for it in range(6):
    v = (t[:, :, it, 0] > 0) & (t[:, :, it, 7] > 0)
    if not v.any():
        continue
    tt = t[:, :, it][v]
    d = np.diff(tt, axis=1)
    print(f"tile {it}: waves {v.sum():5d}  " + "  ".join(f"{n} {np.median(d[:, i]):.0f}/{np.percentile(d[:, i], 90):.0f}" for i, n in enumerate(names)) +
          f"   tile total med {np.median(tt[:, 7] - tt[:, 0]):.0f} ticks (x10 ns)")
