"""Phase timeline of tat_fused_kernel (csrc/tatf.hip) from a -DBNERV_TRACE_TAT build (debug variant; the shipped library has no tracing).
usage: BNERV_LIB=_variants/lib_tattrace.so python tools/ktrace_tat.py [H W]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
B, Cc = 1, 12
x = torch.randn(B, Cc, H, W, device=dev)
w, b = torch.randn(Cc, Cc, 3, 3, device=dev) / 10, torch.randn(Cc, device=dev)
sc, sh = torch.randn(B, Cc, device=dev) * 0.1, torch.randn(B, Cc, device=dev) * 0.1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops._tat_forward(x, sc, sh, sc, sh, w, b, w, b, True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    ops._tat_forward(x, sc, sh, sc, sh, w, b, w, b, True)
e1.record()
torch.cuda.synchronize()
print(f"launch avg {e0.elapsed_time(e1) / 10 * 1e3:.1f} us (eager loop, includes host gaps)")
lib = L.load()
buf = np.zeros(256 * 4 * 4 * 16, dtype=np.uint64)
fn = lib.bnerv_debug_trace_tat_read
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf.reshape(256, 4, 4, 16).astype(np.int64)
names = ["top->dma issued", "kloop c0.0", "epi c0.0", "kloop c0.1", "epi c0.1", "kloop ring", "epi ring", "barrier", "kloop c1.0", "epi c1.0", "kloop c1.1", "epi c1.1", "wait dma+barrier"]
for tile in range(4):
    tt = t[:, :, tile, :14]
    v = (tt[..., 13] > 0) & (tt[..., 0] > 0)
    if not v.any():
        continue
    d = np.diff(tt[v][:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]], axis=1)
    print(f"tile {tile}: waves {v.sum()}  total median {np.median(tt[v][:, 13] - tt[v][:, 0]):.0f} ticks   " + "  ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)))
v = (t[:, :, 0, 0] > 0) & (t[:, :, 1, 0] > 0)
print("tile 0 top -> tile 1 top (ticks), median:", np.median((t[:, :, 1, 0] - t[:, :, 0, 0])[v]))
