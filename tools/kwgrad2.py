"""Weight gradients of the wide layers at full size through the C-ABI (HIP events): the f32 wide kernel against the split one,
plus the max deviation of both from an f64 torch reference on a small crop.
usage: BNERV_SPLIT_WIDE=off|bf16x6|bf16x3 python tools/kwgrad2.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import _lib as L, ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


rn = lambda *s: torch.randn(*s, device=dev)
print(f"mode BNERV_SPLIT_WIDE={os.environ.get('BNERV_SPLIT_WIDE', '(default)')}")
print(f"{'kernel':58s} {'us':>9s} {'TFLOP/s':>9s}")
for (Ci, Co, H, W) in ((38, 38, 1080, 1920), (46, 46, 540, 960), (55, 55, 270, 480), (95, 95, 135, 240), (22, 22, 1080, 1920), (16, 48, 540, 960), (64, 16, 540, 960)):
    B = 1
    x, g = rn(B, Ci, H, W), rn(B, Co, H, W)
    sc, sh = rn(B, Ci) * 0.1, rn(B, Ci) * 0.1
    dw, db = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, device=dev)
    kw = dict(B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, g_mode=L.IN_UNSHUFFLE)
    fl = 2.0 * Ci * Co * 9 * H * W
    for name, fn in {
        "wgrad plain": lambda: ops._wgrad(x, g, dw, db, in_mode=L.IN_PLAIN, **kw),
        "wgrad affine": lambda: ops._wgrad(x, g, dw, db, in_mode=L.IN_AFFINE, scale=sc, shift=sh, **kw),
    }.items():
        t = timeit(fn)
        print(f"{name + f' {Ci}->{Co} @{H}x{W}':58s} {t:9.1f} {fl / t / 1e6:9.2f}")
    # accuracy on the whole tensor against f64
    xa = (x * (1 + sc)[:, :, None, None] + sh[:, :, None, None]).double()
    xp = torch.nn.functional.pad(xa, (1, 1, 1, 1))
    ref = torch.stack([torch.einsum('bchw,bohw->oc', xp[:, :, ky:ky + H, kx:kx + W], g.double()) for ky in range(3) for kx in range(3)], -1).reshape(Co, Ci, 3, 3)
    mag = torch.stack([torch.einsum('bchw,bohw->oc', xp[:, :, ky:ky + H, kx:kx + W].abs(), g.double().abs()) for ky in range(3) for kx in range(3)], -1).reshape(Co, Ci, 3, 3)
    torch.cuda.synchronize()
    err = ((dw.double() - ref).abs() / mag).max().item()
    berr = ((db.double() - g.double().sum((0, 2, 3))).abs() / g.double().abs().sum((0, 2, 3))).max().item()
    print(f"    max |dw - f64| / sum|x g| = {err:.2e}   bias {berr:.2e}")
