"""Phase timeline of wgrad_bfw_kernel from a -DBNERV_TRACE_BW build (debug variant; BNERV_LIB points at it).  The numbers printed are
s_memtime ticks x 10: on this stack the ticks advance at the shader clock (~1.9 GHz under this load), so divide by 10 for cycles.
Build the variant next to the product library (from boosting_nerv_amd/csrc, after build.sh):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBNERV_TRACE_BW -c wgrad.hip -o _obj/wgrad_tbw.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../_dbg/libbnerv_tbw.so _obj/wgrad_tbw.o $(ls _obj/*.o | grep -v "wgrad")
  BNERV_LIB=$PWD/../_dbg/libbnerv_tbw.so python ../../tools/ktrace_bw.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
B, Cc, H, W = 1, int(os.environ.get("KT_C", 38)), 1080, 1920
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
names = ["split g", "issue loads", "barrier A", "MFMA loop", "barrier B", "store x"]
for it in range(7):
    tt = t[:, :, it, :].reshape(-1, 8)
    tt = tt[tt[:, 6] > 0]
    if not len(tt): continue
    d = [tt[:, i + 1] - tt[:, i] for i in range(6)]
    print(f"tile {it + 3}: waves {len(tt):5d}  " + "  ".join(f"{n} {np.median(x_) * 10:.0f}" for n, x_ in zip(names, d)) + f"  total {np.median(tt[:, 6] - tt[:, 0]) * 10:.0f} ns")
