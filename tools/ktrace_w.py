"""Phase timeline of wgrad_lean_kernel from a -DBNERV_TRACE build (debug variant).  s_memtime ticks = core cycles."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
B, Cc, H, W = 1, 12, 720, 1280
x, g = torch.randn(B, Cc, H, W, device=dev), torch.randn(B, Cc, H, W, device=dev)
w, b = torch.empty(Cc, Cc, 3, 3, device=dev), torch.empty(Cc, device=dev)
sc, sh = torch.randn(B, Cc, device=dev) * 0.1, torch.randn(B, Cc, device=dev) * 0.1
for _ in range(4):
    ops._wgrad(x, g, w, b, B=B, Cin=Cc, Cout=Cc, H=H, W=W, k=3, in_mode=L.IN_AFFINE, g_mode=L.IN_UNSHUFFLE, scale=sc, shift=sh)
torch.cuda.synchronize()
lib = L.load()
buf = np.zeros(1024 * 4 * 8 * 8, dtype=np.uint64)
fn = lib.bnerv_debug_trace_read_w; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf.reshape(1024, 4, 8, 8).astype(np.int64)
nb = int((t[:, 0, 7, 0] > 0).sum())
print("blocks traced:", nb)
p = t[:nb, :, 7, :]
print(f"prologue {np.median(p[..., 1] - p[..., 0]):.0f}  loop total {np.median(p[..., 2] - p[..., 1]):.0f}  wave-reduce+slab {np.median(p[..., 3] - p[..., 2]):.0f}  hosted side work {np.median(p[..., 4] - p[..., 3]):.0f}  (cycles, median)")
names = ["top->barA", "issue", "K loop", "-", "barB wait", "commit"]
for it in range(7):
    tt = t[:nb, :, it, :]
    v = tt[..., 6] > 0
    if not v.any(): continue
    tt = tt[v]
    d = [tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1], tt[:, 3] - tt[:, 2], None, tt[:, 5] - tt[:, 3], tt[:, 6] - tt[:, 5]]
    print(f"tile {it}: waves {v.sum():5d}  " + "  ".join(f"{n} {np.median(x_):.0f}" for n, x_ in zip(names, d) if x_ is not None) + f"  total {np.median(tt[:, 6] - tt[:, 0]):.0f}")
