"""Randomised shape sweep of the wide split kernels (conv_bfw / wgrad_bfw) against float64 torch references on the GPU: plain and
PixelShuffle(2 / 3 / 5) convolutions with their data / weight / bias gradients, and the TAT block (affine prologues, gelu pair, residual,
dGELU / dSIN epilogues with their per-channel sums).  usage: python tools/fuzz_wide.py [cases=120] [seed=0] [small]
"small": the same sweep over the shapes of the low-resolution family (convs.hip: Cin <= 32, up-conv data gradients up to 64 channels).
(checker tool: torch fp64 is the reference here, not part of the product)"""
import math, os, random, sys, torch
os.environ.setdefault("BNERV_SPLIT_WIDE_MIN_TILES", "1")
SMALL = "small" in sys.argv[3:]
os.environ["BNERV_SMALL"] = "1" if SMALL else "0"   # wide sweep: keep small test images on the split kernels (not the low-resolution family)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from boosting_nerv_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}


def check(name, a, r, case):
    """Forward results (names ending in "fwd"): SURVEY 8(d) per element, |a - r| <= 1e-5 + 1e-3 |r|.  Gradients (sums over up to 1e5
    pixels): within 2e-5 of the tensor's largest magnitude."""
    a, r = a.double(), r.double()
    err = (a - r).abs().max().item() / max(r.abs().max().item(), 1e-6)
    worst[name] = max(worst.get(name, 0.0), err)
    if name.endswith("fwd"):
        bad = ((a - r).abs() > 1e-5 + 1e-3 * r.abs()).sum().item()
        if bad:
            print(f"MISMATCH {name} {case}: {bad} elements outside 1e-5 + 1e-3 |ref| (max abs err {(a - r).abs().max().item():.3e})")
            return 1
        return 0
    if not err < 2e-5:
        print(f"MISMATCH {name} {case}: rel-to-max error {err:.3e}")
        return 1
    return 0


def ref_conv(x, w, b, s):
    y = F.conv2d(x, w, b, padding=1)
    return F.pixel_shuffle(y, s) if s > 1 else y


def ref_tat(x0, mods, w0, b0, w1, b1):
    s0, t0, s1, t1 = mods
    h = F.gelu(F.conv2d(x0 * (1 + s0) + t0, w0, b0, padding=1))
    return x0 + F.conv2d(h * (1 + s1) + t1, w1, b1, padding=1)


bad = 0
for it in range(N):
    os.environ["BNERV_SPLIT_WIDE_MIN_ITEMS"] = rng.choice(["1", "128"])
    kind = rng.choice(["conv", "ps2", "ps3", "ps5", "tat"])
    B = rng.choice([1, 1, 2, 3])
    H, W = rng.randint(3, 45), 4 * rng.randint(1, 40)
    g = torch.Generator(device="cpu").manual_seed(rng.randint(0, 1 << 30))
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    if kind == "tat":
        Cc = rng.randint(13, 32) if SMALL else rng.randint(17, 130)
        case = (kind, B, Cc, H, W, os.environ["BNERV_SPLIT_WIDE_MIN_ITEMS"])
        x0 = rn(B, Cc, H, W).requires_grad_(True)
        mods = [(rn(B, Cc, 1, 1) * 0.3).requires_grad_(True) for _ in range(4)]
        w0, w1 = [(rn(Cc, Cc, 3, 3) / math.sqrt(9 * Cc)).requires_grad_(True) for _ in range(2)]
        b0, b1 = [(rn(Cc) * 0.1).requires_grad_(True) for _ in range(2)]
        leaves = [x0] + mods + [w0, b0, w1, b1]
        out = ops.tat_block(x0, *mods, w0, b0, w1, b1)
        cot = rn(*out.shape)
        grads = torch.autograd.grad(out, leaves, cot)
        ld = [t.detach().double().requires_grad_(True) for t in leaves]
        ref = ref_tat(ld[0], ld[1:5], *ld[5:])
        rgrads = torch.autograd.grad(ref, ld, cot.double())
        bad += check("tat fwd", out, ref, case)
        for n_, a, r in zip(["dx0", "ds0", "dt0", "ds1", "dt1", "dw0", "db0", "dw1", "db1"], grads, rgrads):
            bad += check("tat " + n_, a, r, case)
    else:
        s = {"conv": 1, "ps2": 2, "ps3": 3, "ps5": 5}[kind]
        Cin = rng.randint(9, 32) if SMALL else rng.randint(9, 110)
        Ct = s * s * rng.randint(5 if s == 2 else 2, (16 if SMALL and rng.random() < 0.7 else 40) if s == 2 else (20 if s == 3 else 8)) if s > 1 else rng.randint(17, 110)
        if s > 2:
            H, W = min(H, 20), min(W, 64)
        if Cin <= 16 and Ct <= 16:
            Ct = 24
        case = (kind, B, Cin, Ct, H, W, os.environ["BNERV_SPLIT_WIDE_MIN_ITEMS"])
        x = rn(B, Cin, H, W).requires_grad_(True)
        w = (rn(Ct, Cin, 3, 3) / math.sqrt(9 * Cin)).requires_grad_(True)
        b = (rn(Ct) * 0.1).requires_grad_(True)
        out = ops.conv2d_ps(x, w, b, s)
        cot = rn(*out.shape)
        grads = torch.autograd.grad(out, [x, w, b], cot)
        ld = [t.detach().double().requires_grad_(True) for t in (x, w, b)]
        ref = ref_conv(*ld, s)
        rgrads = torch.autograd.grad(ref, ld, cot.double())
        bad += check(kind + " fwd", out, ref, case)
        for n_, a, r in zip(["dx", "dw", "db"], grads, rgrads):
            bad += check(f"{kind} {n_}", a, r, case)
torch.cuda.synchronize()
print(f"{N} random cases, {bad} mismatches; worst error relative to the tensor's max per quantity:")
for k in sorted(worst):
    print(f"  {k:12s} {worst[k]:.2e}")
from boosting_nerv_amd.runtime import hard_exit  # noqa: E402
hard_exit(1 if bad else 0)
