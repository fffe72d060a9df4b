"""Phase timeline of conv_q4_kernel (conv4.hip) from a -DBNERV_TRACE build (debug variant; the shipped library has no tracing).
usage: python tools/ktrace.py          (BNERV_LIB=<-DBNERV_TRACE build of conv4.hip>; optional argv: ep mode name)
Timestamps are s_memtime ticks (100 MHz constant clock on gfx950: 10 ns each)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
B, Cc, H, W = 1, 12, 720, 1280
x = torch.randn(B, Cc, H, W, device=dev)
w, b = torch.randn(Cc, Cc, 3, 3, device=dev) / 10, torch.randn(Cc, device=dev)
sc, sh = torch.randn(B, Cc, device=dev) * 0.1, torch.randn(B, Cc, device=dev) * 0.1
out = torch.empty_like(x)
mode = sys.argv[1] if len(sys.argv) > 1 else "bias"
y0, g2, v2 = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
part = torch.empty(L.load().bnerv_conv_tiles(H, W), B, 2, Cc, device=dev)
for _ in range(4):
    if mode == "bias":
        ops._conv(x, w, b, out, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS)
    elif mode == "sin":
        ops._conv(x, w, b, out, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out2=y0)
    else:
        ops._conv(x, w, None, out, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=y0, aux1=g2, aux2=v2, scale=sc, partial=part)
torch.cuda.synchronize()
lib = L.load()
buf = np.zeros(1024 * 4 * 6 * 8, dtype=np.uint64)
fn = lib.bnerv_debug_trace4_read
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf.reshape(1024, 4, 6, 8).astype(np.int64)
valid = t[..., 7] > 0
t0 = t[valid][:, 0].min()
tend = t[valid][:, 7].max()
# s_memtime counters are per XCD and not synchronised: spans are only meaningful inside one XCD (block b runs on XCD b % 8)
for x in range(8):
    tx, vx = t[x::8], valid[x::8]
    print(f"XCD {x}: first loop-top -> last epilogue end: {tx[vx][:, 7].max() - tx[vx][:, 0].min()} cycles;  "
          f"per-block loop time median {np.median([(tb[vb][:, 7].max() - tb[vb][:, 0].min()) for tb, vb in zip(tx, vx) if vb.any()]):.0f}")
pv = (t[:, :, 5, 1] > 0) & (t[:, :, 5, 0] > 0)
pro = (t[:, :, 5, 1] - t[:, :, 5, 0])[pv]
print(f"prologue (kernel entry -> first loop top), cycles: median {np.median(pro):.0f}  p90 {np.percentile(pro, 90):.0f}  max {pro.max()}")
pp = t[:, :, 5, :][pv]
if (pp[:, 2] > 0).all():
    for nm, a_, b_ in (("entry -> slot constants done", 0, 2), ("issue first tile", 2, 3), ("weight loads issued", 3, 4), ("weights in LDS (waits for them)", 4, 5), ("84 weight registers", 5, 6), ("commit (waits for the tile)", 6, 1)):
        print(f"   {nm:34s} median {np.median(pp[:, b_] - pp[:, a_]):7.0f} cycles")
valid[:, :, 5] = False
names = ["advance", "issue DMA", "K loop", "-", "-", "epilogue", "wait DMA + barrier"]
for it in range(6):
    v = valid[:, :, it]
    if not v.any():
        continue
    tt = t[:, :, it][v]
    d = np.diff(tt, axis=1)
    print(f"iter {it}: waves {v.sum():5d}  start(min/med/max) {tt[:,0].min()-t0:6d} {int(np.median(tt[:,0]))-t0:6d} {tt[:,0].max()-t0:6d}   "
          + "  ".join(f"{n} {np.median(d[:, i]):.0f}/{d[:, i].max():.0f}" for i, n in enumerate(names)) + f"   tile total med {np.median(tt[:,7]-tt[:,0]):.0f}")
# per-block spread at barrier B (skew between the 4 waves of a block)
sk = []
for it in range(6):
    v = valid[:, :, it].all(axis=1)
    if v.any():
        a = t[v][:, :, it, 3]
        sk.append((a.max(axis=1) - a.min(axis=1)))
print("wave skew arriving at barrier B (ticks): median", np.median(np.concatenate(sk)), "p90", np.percentile(np.concatenate(sk), 90))
first = t[:, 0, 0, 0][valid[:, 0, 0]]
print("block start spread (ticks): p10/p50/p90/max", [int(np.percentile(first - t0, p)) for p in (10, 50, 90, 100)])
