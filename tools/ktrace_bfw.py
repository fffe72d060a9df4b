"""Stage timeline of the wide split conv kernel (conv_bfw / its pipelined form) from a -DBNERV_TRACE_BFW build of convbf.hip.
usage: BNERV_LIB=_variants/lib_bfwtrace.so [BNERV_BFW_PIPE=0] python tools/ktrace_bfw.py [k2s|k3s|k1|dk3s]
Stamps per stage: 0 top, 1 after barrier A, 2 after DMA / load issue, 3 after the MFMA phases, 4 after the stage's tail (pipelined: DMA wait;
two-block form: barrier B + commit); 5 / 6 epilogue start / end (last chunk of an item)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "k2s"
B, Cc, H, W = 1, 38, 1080, 1920
rn = lambda *s: torch.randn(*s, device=dev)
x, y0, h, gp, c0, dout = (rn(B, Cc, H, W) for _ in range(6))
w, b = rn(Cc, Cc, 3, 3) / 10, rn(Cc)
sc, sh = rn(B, Cc) * 0.1, rn(B, Cc) * 0.1
out, out2 = torch.empty_like(x), torch.empty_like(x)
kw = dict(B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3)
run = {"k2s": lambda: ops._conv(x, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=out2, **kw),
       "k3s": lambda: ops._conv(h, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sc, shift=sh, aux0=y0, **kw),
       "k1": lambda: ops._conv(x, w, b, out, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out2=out2, **kw),
       "dk3s": lambda: ops._conv(dout, w, None, out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=gp, aux1=h, scale=sc, defer=True, **kw)}[mode]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    run()
torch.cuda.synchronize(); e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize(); ops._flush_deferred()
print(f"{mode} 38->38 @1080x1920: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch; BNERV_BFW_PIPE={os.environ.get('BNERV_BFW_PIPE', '(default)')}")
lib = L.load()
buf = np.zeros(256 * 4 * 8 * 8, dtype=np.uint64)
fn = lib.bnerv_debug_trace_read_bfw
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf.reshape(256, 4, 8, 8).astype(np.int64)
names = ["barrier A", "DMA + load issue", "MFMA phases", "stage tail"]
for st in range(8):
    tt = t[:, :, st, :]
    v = (tt[..., 4] > 0) & (tt[..., 0] > 0)
    if not v.any():
        continue
    d = np.diff(tt[v][:, :5], axis=1)
    ep = tt[v][:, 6] - tt[v][:, 5]
    has_ep = (tt[v][:, 6] > 0) & (tt[v][:, 5] > 0)
    print(f"stage {st}: waves {int(v.sum())}  " + "  ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)) + f"   stage {np.median(tt[v][:, 4] - tt[v][:, 0]):.0f}"
          + (f"   epilogue {np.median(ep[has_ep]):.0f} ({int(has_ep.sum())} waves)" if has_ep.any() else ""))
v = (t[:, :, 7, 4] > 0) & (t[:, :, 0, 0] > 0)
if v.any():
    print("stage 0 top -> stage 7 tail (ticks), median:", np.median((t[:, :, 7, 4] - t[:, :, 0, 0])[v]))
