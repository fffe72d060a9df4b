"""Phase timeline of conv_bf_kernel from a -DBNERV_TRACE build of convbf.hip (debug variant; the shipped library has no tracing).
usage: BNERV_LIB=<trace .so> python tools/ktrace_bf.py [mode]     mode: bias | gelu | res | dsin
Stamps per tile iteration: 0 loop top, 1 after barrier A, 2 after issuing the next tile's loads, 3 K loop issued,
4 after barrier B, 5 after commit (convert + LDS writes of the next tile), 6 after the epilogue.  s_memtime ticks."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "bias"
B, Cc, H, W = 1, 12, 720, 1280
x, y0 = torch.randn(B, Cc, H, W, device=dev), torch.randn(B, Cc, H, W, device=dev)
w, b = torch.randn(Cc, Cc, 3, 3, device=dev) / 10, torch.randn(Cc, device=dev)
sc, sh = torch.randn(B, Cc, device=dev) * 0.1, torch.randn(B, Cc, device=dev) * 0.1
out, out2 = torch.empty_like(x), torch.empty_like(x)
part = torch.empty(L.load().bnerv_conv_tiles(H, W), B, 2, Cc, device=dev)
kw = dict(B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3)
run = {"bias": lambda: ops._conv(x, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS, scale=sc, shift=sh, **kw),
       "gelu": lambda: ops._conv(x, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sc, shift=sh, out2=out2, **kw),
       "res": lambda: ops._conv(x, w, b, out, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sc, shift=sh, aux0=y0, **kw),
       "dsin": lambda: ops._conv(x, w, None, out, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=y0, aux1=x, aux2=out2, scale=sc, partial=part, **kw)}[mode]
for _ in range(4):
    run()
torch.cuda.synchronize()
lib = L.load()
buf = np.zeros(1024 * 4 * 6 * 8, dtype=np.uint64)
fn = lib.bnerv_debug_trace_read_bf
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf.reshape(1024, 4, 6, 8).astype(np.int64)
valid = (t[..., 6] > 0) & (t[..., 0] > 0)
names = ["top->barA", "issue", "K loop", "barB wait", "commit", "epilogue"]
print(f"mode {mode}: blocks traced {int(valid[:, 0, 0].sum())}")
for it in range(6):
    v = valid[:, :, it]
    if not v.any():
        continue
    tt = t[:, :, it][v][:, :7]
    d = np.diff(tt, axis=1)
    print(f"iter {it}: waves {int(v.sum()):5d}  " + "  ".join(f"{n} {np.median(d[:, i]):.0f}/{np.percentile(d[:, i], 90):.0f}" for i, n in enumerate(names)) +
          f"   tile total med {np.median(tt[:, 6] - tt[:, 0]):.0f}")
for xc in range(8):
    tx, vx = t[xc::8], valid[xc::8]
    if vx.any():
        print(f"XCD {xc}: first loop top -> last epilogue end {tx[vx][:, 6].max() - tx[vx][:, 0].min()} ticks; per-block loop time median "
              f"{np.median([(tb[vb][:, 6].max() - tb[vb][:, 0].min()) for tb, vb in zip(tx, vx) if vb.any()]):.0f}")
