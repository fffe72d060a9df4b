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
CASES = ((38, 38, 1080, 1920, 1), (46, 46, 540, 960, 1), (55, 55, 270, 480, 1), (95, 95, 135, 240, 1), (22, 22, 1080, 1920, 1), (16, 48, 540, 960, 1),
         (64, 16, 540, 960, 1), (46, 152, 540, 960, 2), (55, 184, 270, 480, 2), (12, 48, 360, 640, 2))
for (Ci, Co, H, W, gs) in CASES:
    B = 1
    x = rn(B, Ci, H, W)
    gsh = rn(B, Co // (gs * gs), gs * H, gs * W)          # what the kernel reads (the pixel-shuffled gradient for the up-convs)
    g = torch.nn.functional.pixel_unshuffle(gsh, gs) if gs > 1 else gsh
    sc, sh = rn(B, Ci) * 0.1, rn(B, Ci) * 0.1
    dw, db = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, device=dev)
    kw = dict(B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3, g_mode=L.IN_UNSHUFFLE, g_s=gs)
    fl = 2.0 * Ci * Co * 9 * H * W
    for name, fn in {
        "wgrad plain": lambda: ops._wgrad(x, gsh, dw, db, in_mode=L.IN_PLAIN, **kw),
        "wgrad affine": lambda: ops._wgrad(x, gsh, dw, db, in_mode=L.IN_AFFINE, scale=sc, shift=sh, **kw),
    }.items():
        if gs > 1 and "affine" in name:
            continue
        t = timeit(fn)
        print(f"{name + f' {Ci}->{Co} @{H}x{W}' + (' (shuffled g)' if gs > 1 else ''):58s} {t:9.1f} {fl / t / 1e6:9.2f}")
    if gs > 1:
        sc, sh = sc * 0, sh * 0
    # accuracy on the whole tensor against f64
    xa = (x * (1 + sc)[:, :, None, None] + sh[:, :, None, None]).double()
    xp = torch.nn.functional.pad(xa, (1, 1, 1, 1))
    ref = torch.stack([torch.einsum('bchw,bohw->oc', xp[:, :, ky:ky + H, kx:kx + W], g.double()) for ky in range(3) for kx in range(3)], -1).reshape(Co, Ci, 3, 3)
    mag = torch.stack([torch.einsum('bchw,bohw->oc', xp[:, :, ky:ky + H, kx:kx + W].abs(), g.double().abs()) for ky in range(3) for kx in range(3)], -1).reshape(Co, Ci, 3, 3)
    torch.cuda.synchronize()
    err = ((dw.double() - ref).abs() / mag).max().item()
    berr = ((db.double() - g.double().sum((0, 2, 3))).abs() / g.double().abs().sum((0, 2, 3))).max().item()
    print(f"    max |dw - f64| / sum|x g| = {err:.2e}   bias {berr:.2e}")
