"""A/B of conv_lean2_kernel against the generic conv kernel on the TAT / head shapes of the 3M models.
usage: python tools/klean2.py            (spawns itself twice: BNERV_NO_LEAN2=1 and default; compares outputs, prints times)
       python tools/klean2.py dump FILE  (one leg)"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (B, Cin, Cout, H, W)
    (1, 30, 30, 45, 80), (2, 38, 38, 9, 40), (1, 38, 38, 1080, 1920), (1, 46, 46, 540, 960), (1, 55, 55, 270, 480),
    (1, 22, 22, 540, 960), (1, 44, 44, 270, 480), (1, 38, 3, 1080, 1920), (1, 3, 38, 1080, 1920), (1, 95, 95, 18, 32), (3, 17, 20, 24, 36),
    (1, 12, 12, 360, 640), (1, 12, 15, 90, 160), (2, 5, 7, 19, 36),
]
if os.environ.get("KLEAN2_SMALL"):
    SHAPES = [s for s in SHAPES if s[3] * s[4] < 100000]


def leg(path):
    from boosting_nerv_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    res, times = {}, {}
    for (B, Ci, Co, H, W) in SHAPES:
        g = torch.Generator().manual_seed(B * 1000 + Ci + Co + H)
        rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
        x, gx = rnd(B, Ci, H, W), rnd(B, Ci, H, W)
        a0, a1, a2 = rnd(B, Co, H, W), rnd(B, Co, H, W), rnd(B, Co, H, W)
        w = rnd(Co, Ci, 3, 3, sc=0.1)
        wt = rnd(Ci, Co, 3, 3, sc=0.1)
        b = rnd(Co)
        sci, shi = rnd(B, Ci, sc=0.3), rnd(B, Ci, sc=0.3)
        sco = rnd(B, Co, sc=0.3)
        kw = dict(B=B, Cin=Ci, Cout=Co, H=H, W=W, k=3)
        xu = rnd(B, Ci, 2 * H, 2 * W) if Ci <= 46 else None     # gradient of a PixelShuffle(2) output: conv-space Cin = 4 Ci
        wu = rnd(4 * Ci, Co, 3, 3, sc=0.1) if Ci <= 46 else None
        if Co <= 46 and Co > 3:
            wps, bps = rnd(4 * Co, Ci, 3, 3, sc=0.1), rnd(4 * Co)
            ps_o, ps_o2 = torch.zeros(B, Co, 2 * H, 2 * W, device=dev), torch.zeros(B, Co, 2 * H, 2 * W, device=dev)
        modes = {
            "affine->gelu": lambda o, o2: ops._conv(x, w, b, o, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=sci, shift=shi, out2=o2, **kw),
            "affine->res": lambda o, o2: ops._conv(x, w, b, o, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_RES, scale=sci, shift=shi, aux0=a0, **kw),
            "plain->sin": lambda o, o2: ops._conv(x, w, b, o, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out2=o2, **kw),
            "plain->tanh": lambda o, o2: ops._conv(x, w, b, o, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_TANH, **kw),
            "T plain->dgelu_saved": lambda o, o2: o2.__setitem__(slice(0, B * 2 * Co), ops._conv(x, wt, None, o, in_mode=L.IN_PLAIN, ep_mode=L.EP_DGELU_SAVED, transposed=1, aux0=a0, aux1=a1, scale=sco, **kw).reshape(-1)),
            "T plain->dsin": lambda o, o2: o2.__setitem__(slice(0, B * 2 * Co), ops._conv(x, wt, None, o, in_mode=L.IN_PLAIN, ep_mode=L.EP_DSIN, transposed=1, aux0=a0, aux1=a1, aux2=a2, scale=sco, **kw).reshape(-1)),
            "T unshuffle2->plain": (lambda o, o2: ops._conv(xu, wu, None, o, B=B, Cin=4 * Ci, Cout=Co, H=H, W=W, k=3, in_mode=L.IN_UNSHUFFLE, ep_mode=L.EP_PLAIN, in_s=2, transposed=1)) if Ci <= 46 else None,
            "plain->sin PS2 (x4 cout)": (lambda o, o2: (ops._conv(x, wps, bps, ps_o, B=B, Cin=Ci, Cout=4 * Co, H=H, W=W, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out_s=2, out2=ps_o2),
                                                          o.copy_(ps_o[:, :, 0::2, 1::2]), o2.copy_(ps_o2[:, :, 1::2, 0::2]))) if Co <= 46 and Co > 3 else None,
            "T tanhgrad->plain": lambda o, o2: ops._conv(x, wt, None, o, in_mode=L.IN_TANHGRAD, ep_mode=L.EP_PLAIN, transposed=1, aux0=gx, **kw),
        }
        for name, fn in modes.items():
            if fn is None:
                continue
            o = torch.zeros(B, Co, H, W, device=dev)
            o2 = torch.zeros(B, Co, H, W, device=dev)
            key = f"{name} {B}x{Ci}->{Co}@{H}x{W}"
            try:
                fn(o, o2.view(-1) if name.startswith("T plain") else o2)
                ops._flush_deferred()
            except Exception as e:  # noqa: BLE001
                print("ERR", key, e)
                continue
            torch.cuda.synchronize()
            if H * W < 100000:
                res[key] = (o.cpu(), o2.cpu())
            else:
                res[key] = (o[:, :, ::7, ::5].contiguous().cpu(), o2[:, :, ::7, ::5].contiguous().cpu())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(o, o2.view(-1) if name.startswith("T plain") else o2)
            e1.record()
            torch.cuda.synchronize()
            times[key] = e0.elapsed_time(e1) * 100.0
    torch.save((res, times), path)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "dump":
        leg(sys.argv[2])
        sys.exit(0)
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "dump", "/tmp/kl2_new.pt"], env=env)
    env["BNERV_NO_LEAN2"] = "1"
    subprocess.check_call([sys.executable, __file__, "dump", "/tmp/kl2_old.pt"], env=env)
    rn, tn = torch.load("/tmp/kl2_new.pt")
    ro, to = torch.load("/tmp/kl2_old.pt")
    for k in ro:
        if k not in rn:
            print(f"{k:44s} MISSING")
            continue
        errs = []
        for a, b in zip(rn[k], ro[k]):
            errs.append(float((a - b).abs().max() / (b.abs().max() + 1e-20)))
        flag = "ok " if max(errs) < 2e-5 else "BAD"
        print(f"{k:44s} {flag} rel {errs[0]:.1e} {errs[1]:.1e}   generic {to[k]:8.1f} us   lean2 {tn[k]:8.1f} us   x{to[k] / tn[k]:.2f}")
