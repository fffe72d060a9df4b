"""GPU parity tests (``-m gpu``): every HIP operator, called through the C-ABI, against the CPU oracle on the same seeded
inputs, plus the golden fixtures produced by the real reference.  Tolerance per SURVEY 8(d): rtol 1e-3 (atol scaled to
the tensor's magnitude) -- fp32 arithmetic, different summation order than MKLDNN."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import group, load_golden
from oracle import cpu_ref, msssim_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RT = 1e-3


def close(a, b, rtol=RT, atol=None, msg=""):
    """|a - b| <= 1e-5 + 1e-3 |b| per element for forward results (activations, images: every check whose msg says
    "fwd").  Gradients summed over 1e4 .. 1e6 pixels get an absolute part that scales with the tensor (an f32 sum in another order
    than MKLDNN's differs by ~1e-7 of the LARGEST partial sum, not of each element)."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (msg, a.shape, b.shape)
    if atol is None:
        atol = 1e-5 if "fwd" in msg else 1e-5 + 1e-4 * float(b.abs().max())
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), f"{msg}: {int(bad.sum())}/{a.numel()} off, max abs err {float(err.max()):.3e}, ref max {float(b.abs().max()):.3e}"


def gpu(t):
    return t.detach().to(DEV).requires_grad_(t.requires_grad)


@pytest.fixture(scope="module")
def ops():
    from boosting_nerv_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------------------------ PE
def test_pe_against_reference_golden(ops):
    g = load_golden("pe.npz")
    bases = torch.from_numpy(g["bases"])
    t64 = torch.from_numpy(g["t64"])
    o32 = ops.positional_encoding(t64[:, None].float().to(DEV), bases).view(3, -1).cpu()
    o64 = ops.positional_encoding(t64[:, None].to(DEV), bases).view(3, -1).cpu()
    oxy = ops.positional_encoding(torch.from_numpy(g["xy"]).to(DEV), bases).view(16, -1).cpu()
    # sin/cos of arguments up to 1.4e8: libm implementations may differ by an ulp of the RESULT, nothing more
    for name, got, ref in (("f32", o32, g["out_f32"]), ("f64", o64, g["out_f64"]), ("xy", oxy, g["out_xy"])):
        err = np.abs(got.numpy() - ref).max()
        assert err < 2e-6, (name, err)


# --------------------------------------------------------------------------------------------------------------- dense
@pytest.mark.parametrize("B", [1, 3, 144])      # 144: the token MLPs of E-NeRV take the library-GEMM route
def test_dense_grouped_fwd_bwd(ops, B):
    g = torch.Generator().manual_seed(3)
    specs = [(160, 256, "sin"), (160, 64, "sin"), (32, 32, "relu"), (32, 12, "none"), (256, 1152, "sin"), (32, 95, "none")]
    xs = [torch.randn(B, i, generator=g).requires_grad_(True) for i, _, _ in specs]
    ws = [(torch.randn(o, i, 1, 1, generator=g) / math.sqrt(i)).requires_grad_(True) for i, o, _ in specs]
    bs = [torch.randn(o, generator=g).requires_grad_(True) for _, o, _ in specs]
    cots = [torch.randn(B, o, generator=g) for _, o, _ in specs]
    ref = [cpu_ref._act(F.linear(x, w.flatten(1), b), a) for x, w, b, (_, _, a) in zip(xs, ws, bs, specs)]
    rg = torch.autograd.grad(ref, xs + ws + bs, cots)
    xg, wg, bg = [gpu(t) for t in xs], [gpu(t) for t in ws], [gpu(t) for t in bs]
    out = ops.dense_grouped(xg, wg, bg, [a for _, _, a in specs])
    for o, r in zip(out, ref):
        close(o, r, msg="dense fwd")
    gg = torch.autograd.grad(out, xg + wg + bg, [c.to(DEV) for c in cots])
    for i, (a, b) in enumerate(zip(gg, rg)):
        close(a, b, msg=f"dense grad {i}")


def test_dense_shared_input_gradient(ops):
    """32 SFT first layers read the same condition vector: its gradient is the sum over groups."""
    g = torch.Generator().manual_seed(4)
    z = torch.randn(2, 32, 1, 1, generator=g).requires_grad_(True)
    ws = [(torch.randn(32, 32, 1, 1, generator=g) / 6).requires_grad_(True) for _ in range(7)]
    bs = [torch.randn(32, generator=g).requires_grad_(True) for _ in range(7)]
    ref = sum(torch.relu(F.conv2d(z, w, b)).sum() * (i + 1) for i, (w, b) in enumerate(zip(ws, bs)))
    rz, = torch.autograd.grad(ref, [z])
    zg = gpu(z)
    out = ops.dense_grouped([zg] * 7, [gpu(w) for w in ws], [gpu(b) for b in bs], ["relu"] * 7)
    tot = sum(o.sum() * (i + 1) for i, o in enumerate(out))
    gz, = torch.autograd.grad(tot, [zg])
    close(gz, rz, msg="shared dz")


def _poison_pool(shapes, n=24):
    """Fill the caching allocator's free blocks of these sizes with NaN: a following torch.empty of the same shape then hands out NaN,
    so a kernel result that was never written (a slab reduction still queued) cannot pass as last call's value."""
    junk = [torch.full(sh, float("nan"), device=DEV) for sh in shapes for _ in range(n)]
    torch.cuda.synchronize()
    del junk


@pytest.mark.parametrize("widths", [(96, 128, 12), (65, 30, 30, 100)])
def test_time_branch_with_modulations_wider_than_one_dx_chunk(ops, widths):
    """NeRV_Boost's one-launch time branch (ops.time_branch; model_nerv.py:47-51, model_blocks.py:92-105) with TAT modulation MLPs
    of more than DENSE_DX_CHUNK = 64 output channels (a NeRV-boost with fc_dim in 65..128): the layer-1 input gradient of such an MLP
    is a two-chunk slab reduction, which the grouped backward must have SUMMED before layer 0 reads it (round-5 advisor finding: it
    was only queued).  Outputs and every parameter gradient against float64 autograd."""
    from boosting_nerv_amd import _lib as L
    g = torch.Generator().manual_seed(11 + len(widths))
    Lv, SH, SO, TH, TO, B = 80, 256, 30 * 9 * 16, 64, 32, 1
    rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc)
    bases = (1.25 ** torch.arange(8, dtype=torch.float32).repeat_interleave(10)) * math.pi      # (moderate arguments: the PE itself is pinned elsewhere)
    pos = torch.tensor([0.37], dtype=torch.float64)
    stem = [rn(SH, 2 * Lv, 1, 1, sc=0.08), rn(SH, sc=0.1), rn(SO, SH, 1, 1, sc=0.06), rn(SO, sc=0.1)]
    stem_t = [rn(TH, 2 * Lv, 1, 1, sc=0.08), rn(TH, sc=0.1), rn(TO, TH, 1, 1, sc=0.12), rn(TO, sc=0.1)]
    mlps = [[rn(TO, TO, 1, 1, sc=0.2), rn(TO, sc=0.1), rn(C_, TO, 1, 1, sc=0.2), rn(C_, sc=0.1)] for C_ in widths]
    flat = stem + stem_t + [t for m in mlps for t in m]
    cots = [rn(B, SO), rn(B, TO)] + [rn(B, C_) for C_ in widths]

    def ref():
        ps = [t.double().requires_grad_(True) for t in flat]
        arg = pos.float()[:, None] * bases[None, :]                      # the reference's single fp32 product (model_blocks.py:122)
        pe = torch.cat([torch.sin(arg), torch.cos(arg)], 1).double()
        lin = lambda x, w, b: x @ w.flatten(1).T + b
        so = torch.sin(lin(torch.sin(lin(pe, ps[0], ps[1])), ps[2], ps[3]))
        zt = torch.sin(lin(torch.sin(lin(pe, ps[4], ps[5])), ps[6], ps[7]))
        outs = [lin(torch.relu(lin(zt, ps[8 + 4 * i], ps[9 + 4 * i])), ps[10 + 4 * i], ps[11 + 4 * i]) for i in range(len(widths))]
        gr = torch.autograd.grad([so, zt] + outs, ps, [c.double() for c in cots])
        return [so, zt] + outs, gr
    r_out, r_g = ref()
    pg = [t.to(DEV).requires_grad_(True) for t in flat]
    assert max(widths) > L.DENSE_DX_CHUNK
    _poison_pool([(2, B, TO), (B, TO)])
    res = ops.time_branch(pos.to(DEV), bases.to(DEV), tuple(pg[:4]), tuple(pg[4:8]), [tuple(pg[8 + 4 * i:12 + 4 * i]) for i in range(len(widths))])
    assert res is not None, "these shapes are the kernel's"
    got = [res[0], res[1]] + list(res[2])
    for a, b in zip(got, r_out):
        close(a, b.float(), rtol=1e-4, atol=3e-5, msg="time branch fwd")
    gg = torch.autograd.grad(got, pg, [c.to(DEV) for c in cots])
    for i, (a, b) in enumerate(zip(gg, r_g)):
        assert torch.isfinite(a).all(), f"time branch grad {i}: non-finite (an unreduced buffer was read)"
        close(a, b.float(), msg=f"time branch grad {i}")


def test_time_branch_refuses_more_groups_than_its_backward_can_hold(ops):
    """40 modulation MLPs fit the forward's table but not the backward's grouped launch (n + 1 groups): the gate must say no, so the
    caller takes the layer-by-layer path instead of failing in backward (round-5 advisor finding)."""
    from boosting_nerv_amd import _lib as L
    z = lambda *sh: torch.zeros(*sh, device=DEV)
    stem = (z(256, 160, 1, 1), z(256), z(64, 256, 1, 1), z(64))
    stem_t = (z(64, 160, 1, 1), z(64), z(32, 64, 1, 1), z(32))
    mk = lambda n: [(z(32, 32, 1, 1), z(32), z(12, 32, 1, 1), z(12)) for _ in range(n)]
    bases = torch.ones(80, device=DEV)
    pos = torch.tensor([0.5], dtype=torch.float64, device=DEV)
    assert ops.time_branch(pos, bases, stem, stem_t, mk(L.MAX_DENSE_GROUPS)) is None
    assert ops.time_branch(pos, bases, stem, stem_t, mk(L.MAX_DENSE_GROUPS - 1)) is not None


@pytest.mark.parametrize("dx_ok", [False, True])
def test_stem_pair_input_gradient_is_complete_for_a_stock_consumer(ops, dx_ok):
    """E-NeRV's first up-conv (Conv_Up_Block.conv1, model_enerv.py:73-102: 3x3 59 -> 350, PixelShuffle(5) at 9x16) takes the stem pair,
    whose data gradient is a QUEUED slab reduction; its input comes from a stock torch.sin, whose backward does not flush.  Inside
    ops.lazy_flush() the operator must therefore reduce before it returns -- unless the model vouches for a flushing reader (dx_ok,
    NeRV_Boost), in which case the reader's flush (here: an explicit one) completes it."""
    g = torch.Generator().manual_seed(5)
    B, Cin, Ct, H, W, s = 1, 59, 350, 9, 16, 5
    z = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Ct, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Ct, generator=g) * 0.1
    cot = torch.randn(B, Ct // (s * s), H * s, W * s, generator=g)
    zr, wr, br = (t.double().requires_grad_(True) for t in (z, w, b))
    ref = F.pixel_shuffle(F.conv2d(torch.sin(zr), wr, br, padding=1), s)
    rz, rw, rb = torch.autograd.grad(ref, [zr, wr, br], cot.double())
    zg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (z, w, b))
    xin = torch.sin(zg)
    out = ops.conv2d_ps(xin, wg, bg, s)
    close(out, ref.float(), msg="stem-shape up-conv fwd")
    _poison_pool([(B, Cin, H, W)])
    if dx_ok:
        hook = xin.register_hook(lambda gr: ops._flush_deferred())      # the flushing reader the model vouches for
    with ops.lazy_flush(dx_ok=dx_ok):
        gz, gw, gb = torch.autograd.grad(out, [zg, wg, bg], cot.to(DEV))
    assert torch.isfinite(gz).all(), "the data gradient reached torch.sin's backward unreduced"
    close(gz, rz.float(), msg="stem pair dz through a stock sin")
    close(gw, rw.float(), msg="stem pair dw")
    close(gb, rb.float(), msg="stem pair db")


# ---------------------------------------------------------------------------------------------------------------- conv
CONV_CASES = [  # B, Cin, Cout_total, H, W, k, s
    (1, 12, 12, 16, 64, 3, 1), (2, 12, 48, 9, 33, 3, 2), (1, 15, 48, 11, 40, 3, 2), (2, 30, 750, 9, 16, 3, 5),
    (1, 9, 63, 5, 6, 3, 3), (2, 20, 132, 6, 9, 1, 2), (1, 95, 100, 9, 16, 1, 5), (1, 55, 55, 17, 35, 3, 1),
    (1, 3, 3, 1, 1, 3, 1), (1, 70, 18, 8, 32, 3, 1),
    # aligned rows (W % 4 == 0) with more than 16 total output channels: the wide weight-gradient kernels with the PixelShuffle
    # 3 / 5 / 2 gather (the paths C3's 79->594 and C4's 177->792 up-convs take) and the stride-1 wide kernels
    (1, 20, 180, 24, 36, 3, 3), (1, 24, 400, 9, 16, 3, 5), (1, 40, 160, 16, 32, 3, 2), (1, 38, 38, 24, 64, 3, 1),
    (2, 79, 594, 9, 16, 3, 3),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_ps_fwd_bwd(ops, case):
    B, Cin, Ct, H, W, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Ct, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Ct, generator=g).requires_grad_(True)
    ref = cpu_ref.upconv(x, w, b, s)
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    out = ops.conv2d_ps(xg, wg, bg, s)
    close(out, ref, msg="conv fwd")
    gg = torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV))
    for n, a, r in zip("xwb", gg, rg):
        close(a, r, msg=f"conv d{n}")


def test_sft_affine(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 12, 19, 37, generator=g).requires_grad_(True)
    sc = torch.randn(2, 12, 1, 1, generator=g).requires_grad_(True)
    sh = torch.randn(2, 12, 1, 1, generator=g).requires_grad_(True)
    ref = cpu_ref.sft_affine(x, sc, sh)
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, sc, sh], cot)
    xs = [gpu(t) for t in (x, sc, sh)]
    out = ops.sft_affine(*xs)
    close(out, ref, msg="sft fwd")
    for n, a, r in zip(("x", "scale", "shift"), torch.autograd.grad(out, xs, cot.to(DEV)), rg):
        close(a, r, msg=f"sft d{n}")


def _tat_inputs(B, Cc, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).requires_grad_(True)
    x0 = mk(B, Cc, H, W)
    mods = [mk(B, Cc, 1, 1, sc=0.5) for _ in range(4)]
    w0, w1 = mk(Cc, Cc, 3, 3, sc=1 / math.sqrt(9 * Cc)), mk(Cc, Cc, 3, 3, sc=1 / math.sqrt(9 * Cc))
    b0, b1 = mk(Cc, sc=0.1), mk(Cc, sc=0.1)
    return x0, mods, w0, b0, w1, b1, g


def _tat_ref(x0, mods, w0, b0, w1, b1):
    s0, t0, s1, t1 = mods
    f = F.gelu(F.conv2d(x0 * (s0 + 1) + t0, w0, b0, padding=1))
    return x0 + F.conv2d(f * (s1 + 1) + t1, w1, b1, padding=1)


# (95 x 9 x 16, 64 x 20 x 48 and 33 x 7 x 12: the 96-channel staging of convs.hip's low-resolution family -- HNeRV-boost's decoder[0])
@pytest.mark.parametrize("shape", [(1, 12, 16, 64), (2, 15, 11, 13), (1, 30, 45, 80), (2, 38, 9, 40), (2, 55, 17, 36), (1, 95, 10, 44), (3, 17, 8, 32),
                                   (1, 95, 9, 16), (2, 64, 20, 48), (1, 33, 7, 12)])
def test_tat_block(ops, shape):
    x0, mods, w0, b0, w1, b1, g = _tat_inputs(*shape, seed=7)
    ref = _tat_ref(x0, mods, w0, b0, w1, b1)
    cot = torch.randn(ref.shape, generator=g)
    leaves = [x0] + mods + [w0, b0, w1, b1]
    rg = torch.autograd.grad(ref, leaves, cot)
    gl = [gpu(t) for t in leaves]
    out = ops.tat_block(gl[0], gl[1], gl[2], gl[3], gl[4], gl[5], gl[6], gl[7], gl[8])
    close(out, ref, msg="tat fwd")
    names = ["x0", "s0", "t0", "s1", "t1", "w0", "b0", "w1", "b1"]
    for n, a, r in zip(names, torch.autograd.grad(out, gl, cot.to(DEV)), rg):
        close(a, r, msg=f"tat d{n}")


@pytest.mark.parametrize("shape", [(1, 12, 48, 96), (2, 12, 37, 68), (3, 12, 16, 32), (2, 9, 21, 36), (1, 11, 180, 320), (1, 10, 33, 132), (1, 12, 360, 640)])
def test_tat_block_backward_on_the_shared_tile_pair(ops, shape, monkeypatch):
    """The shared-tile backward pair (csrc/pairf_body.h: data gradient + weight gradient of one 12-channel 3x3 layer from ONE staged
    copy of the incoming gradient; default only from 1024 tiles on) forced on for every size (BNERV_PAIR_FUSED=8): the whole TAT block
    against the oracle (stock ops on the CPU, reference semantics model_blocks.py:74-89) -- its conv1 pair is the DGELU_SAVED form, its
    conv0 pair the DSIN form -- and against the interleaved pair it replaces (same quantities, fp32 summation order aside).  Shapes: one
    tile touching every border, ragged bottom / right tiles, B > 1 (the per-sample flush of the per-lane channel sums and the per-sample
    affine), 9 / 10 / 11 channels, several tiles per block (360x640: 900 tiles on 512 blocks)."""
    x0, mods, w0, b0, w1, b1, g = _tat_inputs(*shape, seed=31)
    ref = _tat_ref(x0, mods, w0, b0, w1, b1)
    cot = torch.randn(ref.shape, generator=g)
    leaves = [x0] + mods + [w0, b0, w1, b1]
    rg = torch.autograd.grad(ref, leaves, cot)
    names = ["x0", "s0", "t0", "s1", "t1", "w0", "b0", "w1", "b1"]
    got = {}
    # B == 1: "fused" is the FOLD form (raw input tile by LDS-DMA, the affine applied by the slab reduction, 9 tap-masked gradient sums through
    # the mask plane -- every tile of these shapes touches the image border) and "transform" the form that applies the prologue on the way into
    # LDS; B > 1: both are the transforming form (the fold is per sample)
    for form, env, fold in (("fused", "8", "1"), ("transform", "8", "0"), ("interleaved", "0", "1")):
        monkeypatch.setenv("BNERV_PAIR_FUSED", env)
        monkeypatch.setenv("BNERV_PAIR_FOLD", fold)
        gl = [gpu(t) for t in leaves]
        out = ops.tat_block(*gl)
        close(out, ref, msg=f"tat fwd ({form})")
        got[form] = torch.autograd.grad(out, gl, cot.to(DEV))
        for n, a, r in zip(names, got[form], rg):
            close(a, r, msg=f"tat d{n} ({form})")
    for n, a, b, c in zip(names, got["fused"], got["interleaved"], got["transform"]):
        close(a, b, msg=f"tat d{n}: shared-tile pair vs interleaved pair")
        close(a, c, msg=f"tat d{n}: fold form vs transforming form")


@pytest.mark.parametrize("case", [(1, 12, 12, 16, 64, 3, 1), (2, 12, 12, 9, 33, 3, 2), (1, 30, 15, 9, 16, 3, 5), (1, 20, 33, 6, 9, 1, 2), (1, 9, 7, 5, 6, 3, 3),
                                  (1, 20, 20, 24, 36, 3, 3), (1, 40, 38, 16, 32, 3, 2)])
def test_snerv_block(ops, case):
    B, Cin, Cc, H, W, k, s = case
    x0, mods, w0, b0, w1, b1, g = _tat_inputs(B, Cc, H * s, W * s, seed=11)
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    wu = (torch.randn(Cc * s * s, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).requires_grad_(True)
    bu = (torch.randn(Cc * s * s, generator=g) * 0.1).requires_grad_(True)
    ref = _tat_ref(torch.sin(cpu_ref.upconv(x, wu, bu, s)), mods, w0, b0, w1, b1)
    cot = torch.randn(ref.shape, generator=g)
    leaves = [x, wu, bu] + mods + [w0, b0, w1, b1]
    rg = torch.autograd.grad(ref, leaves, cot)
    gl = [gpu(t) for t in leaves]
    out = ops.snerv_block(*gl, s)
    close(out, ref, msg="snerv fwd")
    names = ["x", "wu", "bu", "s0", "t0", "s1", "t1", "w0", "b0", "w1", "b1"]
    for n, a, r in zip(names, torch.autograd.grad(out, gl, cot.to(DEV)), rg):
        close(a, r, msg=f"snerv d{n}")


@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("hw", [(21, 45), (16, 64)])      # (16, 64): aligned rows -> the streaming 1x1 head data-gradient kernel
def test_head_tanh(ops, k, hw):
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 12, hw[0], hw[1], generator=g).requires_grad_(True)
    w = (torch.randn(3, 12, k, k, generator=g) / math.sqrt(12 * k * k)).requires_grad_(True)
    b = torch.randn(3, generator=g).requires_grad_(True)
    ref = cpu_ref.out_img(F.conv2d(x, w, b, padding=(k - 1) // 2))
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    gl = [gpu(t) for t in (x, w, b)]
    out = ops.head_tanh(*gl)
    close(out, ref, msg="head fwd")
    for n, a, r in zip("xwb", torch.autograd.grad(out, gl, cot.to(DEV)), rg):
        close(a, r, msg=f"head d{n}")


# (from 4096 pixels on, forward and data gradient of the 3-output head are csrc/head3.hip's streaming kernels: 72 x 128, and 64 x 68 with ragged tiles)
@pytest.mark.parametrize("case", [(1, 38, 3, 40, 64), (2, 20, 3, 17, 36), (1, 16, 4, 33, 20), (1, 38, 3, 24, 30), (1, 38, 3, 72, 128), (2, 20, 3, 64, 68), (1, 64, 3, 70, 60)])
def test_wide_head_weight_gradient_with_swapped_roles(ops, case, monkeypatch):
    """HNeRV_Boost's 3x3 output head (38 -> 3 + OutImg tanh, model_hnerv.py:214, model_blocks.py:57-63): the weight gradient computed
    with input and gradient swapped (ops._HeadTanh.backward: bnerv_tanh_grad + one weight-gradient launch with the 38 channels on the M
    side, then a transpose + tap flip) against the CPU oracle and against the direct form (BNERV_HEAD_SWAP=0) -- dx, dw, db; image
    sizes with ragged rows (W % 4 != 0: the streaming kernel's scalar path) and borders in every tile."""
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    ref = cpu_ref.out_img(F.conv2d(x, w, b, padding=1))
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BNERV_HEAD_SWAP", flag)
        gl = [gpu(t) for t in (x, w, b)]
        out = ops.head_tanh(*gl)
        close(out, ref, msg="wide head fwd")
        res[flag] = torch.autograd.grad(out, gl, cot.to(DEV))
        for n, a, r in zip("xwb", res[flag], rg):
            close(a, r, msg=f"wide head d{n} (swap {flag})")
    for a, c in zip(res["1"], res["0"]):
        close(a, c.cpu(), msg="swapped against direct")


def test_blocks_against_reference_goldens():
    """The module-level API (same classes / state_dict keys as the reference) on the reference's own golden vectors."""
    from oracle import configs
    from boosting_nerv_amd import model_blocks as mb
    from boosting_nerv_amd.model_enerv import Conv_Up_Block
    npz = load_golden("blocks.npz")
    args = configs.tiny_nerv()

    def run(name, module):
        b = group(npz, name + "/")
        module.load_state_dict({k[3:]: v for k, v in b.items() if k.startswith("sd/")})
        module.to(DEV)
        x = b["x"].to(DEV).requires_grad_(True)
        z = b["z"].to(DEV).requires_grad_(True)
        y = module((x, z))
        close(y, b["y"], msg=f"{name} fwd")
        params = dict(module.named_parameters())
        gs = torch.autograd.grad(y, [x, z] + list(params.values()), b["cot"].to(DEV))
        close(gs[0], b["dx"], msg=f"{name} dx")
        close(gs[1], b["dz"], msg=f"{name} dz")
        for (pn, _), gval in zip(params.items(), gs[2:]):
            close(gval, b[f"grad/{pn}"], msg=f"{name} grad {pn}")

    run("sft_c12", mb.SFTLayer(32, 12, 1, "relu", 1, args=args))
    run("tat_c15", mb.ResBlock_SFT(15, 15, cond_ch=32, in_act="relu", out_act="gelu", omega=1, args=args))
    for name, ngf, new_ngf, s, k in (("blk_s1_k3_c12", 12, 12, 1, 3), ("blk_s2_k3_c15_12", 15, 12, 2, 3), ("blk_s3_k3_c9_7", 9, 7, 3, 3),
                                     ("blk_s5_k1_c30", 10, 10, 5, 1), ("blk_s2_k1_c20_33", 20, 33, 2, 1)):
        run(name, mb.NeRVBlock(dec_block=True, conv_type="pshuffel_3x3", ngf=ngf, new_ngf=new_ngf, ks=k, strd=s, bias=True, norm="none",
                               act="sin", sft_ngf=32, args=args))
    run("conv_up_block", Conv_Up_Block(ngf=8, new_ngf=24, ks=3, stride=5, bias=True, norm="none", act="sin", conv_type="pshuffel_3x3",
                                       sft_ngf=32, args=args))
    run("hnerv_dec0", mb.NeRVBlock(dec_block=False, conv_type="conv", ngf=4, new_ngf=10, ks=0, strd=1, bias=True, norm="none", act="sin",
                                   sft_ngf=32, args=configs.tiny_hnerv()))


# ---------------------------------------------------------------------------------------------------------------- loss
@pytest.mark.parametrize("lt", ["L1", "L2", "L1_freq", "Fusion10", "Fusion10_freq"])
@pytest.mark.parametrize("tag", ["small", "odd"])
def test_loss_against_goldens_and_oracle(ops, lt, tag):
    npz = load_golden("loss.npz")
    pred = torch.from_numpy(npz[f"{tag}/pred"]).clone().requires_grad_(True)
    tgt = torch.from_numpy(npz[f"{tag}/target"])
    ref = cpu_ref.loss_fn(pred, tgt, lt)
    rgrad, = torch.autograd.grad(ref, [pred])
    pg = gpu(pred)
    loss, stats = ops.loss_with_stats(pg, tgt.to(DEV), lt)
    if f"{tag}/{lt}/loss" in npz.files:      # value produced by the real reference's loss_fn
        gold = float(npz[f"{tag}/{lt}/loss"])
        assert abs(loss.item() - gold) <= 2e-4 * abs(gold), (loss.item(), gold)
    assert abs(loss.item() - ref.item()) <= 2e-4 * abs(ref.item()), (loss.item(), ref.item())
    ggrad, = torch.autograd.grad(loss, [pg])
    close(ggrad, rgrad, rtol=2e-3, atol=2e-3 * float(rgrad.abs().max()), msg=f"{lt} grad")
    close(ops.psnr(pg, tgt.to(DEV)), cpu_ref.psnr_fn_single(pred, tgt), rtol=1e-5, atol=1e-3, msg="psnr")
    # the train step's fused form: same value, same gradient, PSNR from the same L2 sums
    l2, st2, g2 = ops.loss_value_grad_stats(pg, tgt.to(DEV), lt)
    assert l2.item() == loss.item() and torch.equal(g2, ggrad) and torch.equal(st2[:, 4], ops.psnr(pg, tgt.to(DEV)))
    if "Fusion" in lt:
        close(ops.msssim(pg, tgt.to(DEV)), msssim_ref.ms_ssim(pred.detach(), tgt, data_range=1, size_average=False), rtol=1e-4, atol=1e-5, msg="msssim")


def test_fft_loss_full_size_properties(ops):
    """720x1280 and 1080x1920 (BASELINE sizes): value against the golden of the reference (720p) and linearity /
    Parseval-type properties that do not need the oracle."""
    npz = load_golden("loss.npz")
    g = torch.Generator().manual_seed(int(npz["720p/seed"]))
    tgt = torch.rand(1, 3, 720, 1280, generator=g)
    pred = (tgt + 0.1 * torch.randn(tgt.shape, generator=g)).clamp(0, 1)
    for lt in ("L1", "L2", "L1_freq", "Fusion10_freq"):
        pg = pred.to(DEV).requires_grad_(True)
        loss, _ = ops.loss_with_stats(pg, tgt.to(DEV), lt)
        gold = float(npz[f"720p/{lt}/loss"])
        assert abs(loss.item() - gold) <= 3e-4 * abs(gold), (lt, loss.item(), gold)
        gg, = torch.autograd.grad(loss, [pg])
        f = gg.flatten().cpu()
        idx = torch.from_numpy(npz[f"720p/{lt}/grad.idx"])
        ref = torch.from_numpy(npz[f"720p/{lt}/grad.val"])
        close(f[idx], ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()), msg=f"720p {lt} grad samples")
    # a single DC offset: F(d) is one bin of size c*H*W -> loss_fft = |c| * H*W / (H*W*2) = |c|/2, for every size
    for H, W in ((1080, 1920), (720, 1280), (165, 203)):
        t = torch.rand(1, 3, H, W, generator=g).to(DEV)
        loss, _ = ops.loss_with_stats((t + 0.25).requires_grad_(True), t, "L1_freq")
        assert abs(loss.item() - (60 * 0.25 + 0.125)) < 1e-3, (H, W, loss.item())


def test_recipe_loss_at_1080p_against_oracle_and_independent_msssim(ops):
    """the recipe's loss at BASELINE's 1080x1920 -- Fusion10_freq value and gradient, and ops.msssim, against the
    CPU oracle (cpu_ref.loss_fn / msssim_ref: the MS-SSIM pyramid is ODD here, 1080 -> 540 -> 270 -> 135 -> 68, so the padded average
    pool and the un-merged launch path run) and the MS-SSIM value against the independently written float64 form from the definition
    (tests/msssim_independent.py).  The L1 / FFT terms are pinned to the reference by the 1080p model goldens (loss_L1_freq); this is
    the term they do not cover."""
    import numpy as np
    import msssim_independent as ind
    g = torch.Generator().manual_seed(11)
    tgt = torch.rand(1, 3, 1080, 1920, generator=g)
    tgt = torch.nn.functional.avg_pool2d(tgt, 3, stride=1, padding=1)          # some structure at every scale
    pred = (tgt + 0.08 * torch.randn(tgt.shape, generator=g)).clamp(0, 1).requires_grad_(True)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref = cpu_ref.loss_fn(pred, tgt, "Fusion10_freq")
    rgrad, = torch.autograd.grad(ref, [pred])
    pg = gpu(pred)
    loss, stats = ops.loss_with_stats(pg, tgt.to(DEV), "Fusion10_freq")
    assert abs(loss.item() - ref.item()) <= 2e-4 * abs(ref.item()), (loss.item(), ref.item())
    ggrad, = torch.autograd.grad(loss, [pg])
    idx = torch.from_numpy(np.sort(np.random.RandomState(3).choice(pred.numel(), 512, replace=False)))
    gs, rs = ggrad.flatten().cpu()[idx], rgrad.flatten()[idx]
    close(gs, rs, rtol=2e-3, atol=2e-3 * float(rgrad.abs().max()), msg="1080p Fusion10_freq gradient samples")
    close(ggrad, rgrad, rtol=2e-3, atol=2e-3 * float(rgrad.abs().max()), msg="1080p Fusion10_freq gradient (whole)")
    l2, st2, g2 = ops.loss_value_grad_stats(pg, tgt.to(DEV), "Fusion10_freq")       # the train step's fused form: the same bits
    assert l2.item() == loss.item() and torch.equal(g2, ggrad)
    ms_h = ops.msssim(pg, tgt.to(DEV)).cpu().double().numpy()
    ms_r = msssim_ref.ms_ssim(pred.detach(), tgt, data_range=1, size_average=False).double().numpy()
    ms_i = ind.ms_ssim(pred.detach().numpy(), tgt.numpy())
    assert np.abs(ms_h - ms_r).max() < 2e-5, (ms_h, ms_r)
    assert np.abs(ms_h - ms_i).max() < 2e-5 and np.abs(ms_r - ms_i).max() < 5e-6, (ms_h, ms_r, ms_i)
    assert abs(float(stats[0, 3]) - float(ms_i[0])) < 2e-5                       # the loss statistics carry the same MS-SSIM


# ---------------------------------------------------------------------------------------------------------------- Adan
def test_adan_against_reference_trajectory():
    from boosting_nerv_amd.optimizer import Adan
    npz = load_golden("optim.npz")
    params = [torch.from_numpy(npz[f"p0/{i}"]).to(DEV).requires_grad_(True) for i in range(3)]
    opt = Adan(params, lr=0.003)
    for step in range(6):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(npz[f"g{step}/{i}"]).to(DEV)
        for group_ in opt.param_groups:
            group_["lr"] = 0.003 * (0.1 + 0.15 * step)
        opt.step()
        for i, p in enumerate(params):
            close(p, torch.from_numpy(npz[f"p{step + 1}/{i}"]), rtol=2e-5, atol=2e-6, msg=f"adan step {step} tensor {i}")


def test_deferred_reductions_match_immediate(ops):
    """Slab reductions queued with bnerv_reduce_slabs_deferred in a caller-owned context and executed by a hosting conv launch /
    by the flush give the same sums as the immediate kernel (fixed summation order -> compare tightly), for both job layouts;
    contexts are independent (a launch never hosts another context's jobs), and a NULL context runs the reduction at once."""
    import ctypes as C
    from boosting_nerv_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    slabs = torch.randn(700, 1308, generator=g).to(DEV)          # weight-gradient sized job
    parts = torch.randn(3600, 24, generator=g).to(DEV)           # per-tile epilogue sums
    ref_a, ref_b = slabs.double().sum(0), parts.double().sum(0)
    out_a, out_b = torch.empty(1308, device=DEV), torch.empty(24, device=DEV)
    ctx = L.ctx().handle                                         # the context ops._conv passes for the current stream
    other = L.StreamContext()                                    # a second, unrelated context
    assert lib.bnerv_deferred_pending(ctx) == 0
    L.check(lib.bnerv_reduce_slabs_deferred(ctx, L.stream(), L.ptr(slabs), 700, 1308, L.ptr(out_a)), "defer")
    L.check(lib.bnerv_reduce_slabs_deferred(ctx, L.stream(), L.ptr(parts), 3600, 24, L.ptr(out_b)), "defer")
    out_o = torch.zeros(24, device=DEV)
    L.check(lib.bnerv_reduce_slabs_deferred(other.handle, L.stream(), L.ptr(parts), 3600, 24, L.ptr(out_o)), "defer")
    assert lib.bnerv_deferred_pending(ctx) == 2 and lib.bnerv_deferred_pending(other.handle) == 1
    # a launch with a small grid must not host a big reduction (16 tiles here): the jobs stay queued ...
    w = (torch.randn(12, 12, 3, 3, generator=g) / 10).to(DEV); b = torch.randn(12, generator=g).to(DEV)
    xs = torch.randn(1, 12, 64, 128, generator=g).to(DEV)
    ops._conv(xs, w, b, torch.empty_like(xs), B=1, Cin=12, Cout=12, H=64, W=128, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS)
    assert lib.bnerv_deferred_pending(ctx) == 2
    # ... and a lean conv launch with enough blocks hosts both -- of ITS context only
    x = torch.randn(1, 12, 256, 512, generator=g).to(DEV)
    y = torch.empty_like(x)
    ops._conv(x, w, b, y, B=1, Cin=12, Cout=12, H=256, W=512, k=3, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS)
    assert lib.bnerv_deferred_pending(ctx) == 0 and lib.bnerv_deferred_pending(other.handle) == 1
    torch.cuda.synchronize()
    assert float(out_o.abs().max()) == 0.0                      # the other context's job has not run
    torch.testing.assert_close(y.cpu(), F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out_a.double().cpu(), ref_a.cpu(), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(out_b.double().cpu(), ref_b.cpu(), rtol=1e-5, atol=1e-3)
    L.check(lib.bnerv_flush_deferred(other.handle, L.stream()), "flush other")
    assert lib.bnerv_deferred_pending(other.handle) == 0 and torch.equal(out_o, out_b)
    # flush path (no host launch), the NULL-context form and the immediate kernel agree
    out_c, out_d, out_e = torch.empty(1308, device=DEV), torch.empty(1308, device=DEV), torch.empty(1308, device=DEV)
    L.check(lib.bnerv_reduce_slabs_deferred(ctx, L.stream(), L.ptr(slabs), 700, 1308, L.ptr(out_c)), "defer")
    L.check(lib.bnerv_flush_deferred(ctx, L.stream()), "flush")
    assert lib.bnerv_deferred_pending(ctx) == 0
    L.check(lib.bnerv_reduce_slabs(L.stream(), L.ptr(slabs), 700, 1308, L.ptr(out_d)), "reduce")
    L.check(lib.bnerv_reduce_slabs_deferred(None, L.stream(), L.ptr(slabs), 700, 1308, L.ptr(out_e)), "defer without a context")
    assert torch.equal(out_a, out_c) and torch.equal(out_c, out_e)
    torch.testing.assert_close(out_c, out_d, rtol=1e-5, atol=1e-4)


def test_flush_of_many_deferred_reductions_finds_every_job(ops):
    """The stand-alone flush carries up to 32 queued reductions per launch and every block looks its job up from the prefix sums of their
    slice counts (sidejob.h side_find_job: one load + one ballot instead of a walk over the jobs): 70 jobs of very different shapes
    (three launches), each against the immediate kernel's result bit for bit and against float64."""
    from boosting_nerv_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    ctx = L.StreamContext()
    jobs = []
    for i in range(70):
        n_slabs = [1, 3, 17, 64, 200, 700][i % 6]
        count = [5, 24, 333, 1308, 40, 4097, 9][i % 7]
        slabs = torch.randn(n_slabs, count, generator=g).to(DEV)
        out = torch.full((count,), float("nan"), device=DEV)
        L.check(lib.bnerv_reduce_slabs_deferred(ctx.handle, L.stream(), L.ptr(slabs), n_slabs, count, L.ptr(out)), "defer")
        jobs.append((slabs, out, n_slabs, count))
    assert lib.bnerv_deferred_pending(ctx.handle) == 70
    L.check(lib.bnerv_flush_deferred(ctx.handle, L.stream()), "flush")
    assert lib.bnerv_deferred_pending(ctx.handle) == 0
    for slabs, out, n_slabs, count in jobs:
        now = torch.empty(count, device=DEV)
        L.check(lib.bnerv_reduce_slabs_deferred(None, L.stream(), L.ptr(slabs), n_slabs, count, L.ptr(now)), "immediate")
        assert torch.equal(out, now), (n_slabs, count)
        torch.testing.assert_close(out.double(), slabs.double().sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 5, 23, 70, 7), (1, 64, 36, 64, 7), (1, 3, 9, 16, 3), (2, 4, 17, 33, 5), (1, 8, 100, 200, 7), (1, 64, 216, 384, 7)])
def test_dwconv_fwd_bwd(ops, shape):
    """Depthwise KxK conv of the ConvNeXt encoder block against F.conv2d(groups=C): output, dx, dw, db."""
    B, C, H, W, K = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(C, 1, K, K, generator=g) / K).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=K // 2, groups=C)
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    out = ops.dwconv(xg, wg, bg)
    close(out, ref, msg="dwconv fwd")
    for n, a, r in zip("xwb", torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV)), rg):
        close(a, r, msg=f"dwconv d{n}")


@pytest.mark.parametrize("case", [(2, 64, 256, 64, 64), (1, 256, 64, 72, 128), (1, 48, 20, 64, 68), (3, 16, 16, 40, 104), (1, 80, 130, 216, 384)])
def test_pointwise_weight_gradient_as_a_gemm_over_pixels(ops, case):
    """k = 1 weight / bias gradient of layers with 16 or more channels on both sides from 4096 pixels on (csrc/wgrad1.hip: the ConvNeXt
    block's pwconv1 / pwconv2 of the reference's encoder, model_blocks.py:245-258, through lib/quant_ops.py:39-41's backward) against the
    float64 contraction: channel counts that are not multiples of 16 / 64, several samples, a pixel count that does not divide into the
    slabs evenly, and the bias column."""
    from boosting_nerv_amd import _lib as L
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    gy = torch.randn(B, Cout, H, W, generator=g).to(DEV)
    dw, db = torch.full((Cout, Cin, 1, 1), float("nan"), device=DEV), torch.full((Cout,), float("nan"), device=DEV)
    ops._wgrad(x, gy, dw, db, B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=1, in_mode=L.IN_PLAIN, g_mode=L.IN_UNSHUFFLE)
    ref = torch.einsum("bohw,bihw->oi", gy.double(), x.double())
    bound = torch.einsum("bohw,bihw->oi", gy.double().abs(), x.double().abs())
    err = ((dw.flatten(1).double() - ref).abs() / bound).max().item()
    assert err < 3.5e-7, f"pointwise dW: max |err| / sum|g||x| = {err:.2e}"
    rb = gy.double().sum((0, 2, 3))
    eb = ((db.double() - rb).abs() / gy.double().abs().sum((0, 2, 3))).max().item()
    assert eb < 3.5e-7, f"pointwise db: {eb:.2e}"
    # and through the operator the encoder uses (conv2d_ps with k = 1): same gradient as F.conv2d
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(DEV).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(DEV).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    out = ops.conv2d_ps(xr, w, b, 1)
    gw, gb_, gx = torch.autograd.grad(out, [w, b, xr], gy)
    close(gw.flatten(1), ref.float(), msg="conv2d_ps k=1 dw")
    close(gb_, rb.float(), msg="conv2d_ps k=1 db")
    close(gx, torch.einsum("bohw,oi->bihw", gy.double(), w.detach().flatten(1).double()).float(), msg="conv2d_ps k=1 dx")


@pytest.mark.parametrize("shape", [(2, 5, 23, 70), (1, 64, 36, 64), (3, 16, 9, 16), (1, 1, 4, 300)])
def test_layernorm_channels_first(ops, shape):
    """Channel LayerNorm on NCHW (lnorm.hip) against the reference's channels_first arithmetic in float64: y, dx, dw, db."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(B, C, H, W, generator=g) * 2 + 0.5).requires_grad_(True)
    w = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    cot = torch.randn(B, C, H, W, generator=g)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    u = xd.mean(1, keepdim=True)
    s = (xd - u).pow(2).mean(1, keepdim=True)
    ref = wd[:, None, None] * ((xd - u) / torch.sqrt(s + 1e-6)) + bd[:, None, None]
    rg = torch.autograd.grad(ref, [xd, wd, bd], cot.double())
    xs = [gpu(t) for t in (x, w, b)]
    out = ops.layernorm_cf(xs[0], xs[1], xs[2], 1e-6)
    close(out, ref.float(), msg="ln fwd")
    for n, a, r in zip(("x", "w", "b"), torch.autograd.grad(out, xs, cot.to(DEV)), rg):
        close(a, r.float(), msg=f"ln d{n}")


def test_convnext_block_matches_channels_last_form(ops):
    """Block.forward on the GPU (NCHW: HIP dwconv + HIP LayerNorm + 1x1 convs) against the module's own CPU path, which is the
    reference's channels_last sequence (model_blocks.py:223-247): output and every parameter gradient."""
    from boosting_nerv_amd import model_blocks as mb
    torch.manual_seed(5)
    blk = mb.Block(dim=16, layer_scale_init_value=0.5)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.add_(torch.randn_like(p_) * 0.1)
    x = torch.randn(2, 16, 18, 32)
    cot = torch.randn(2, 16, 18, 32)
    xr = x.clone().requires_grad_(True)
    ref = blk(xr)
    rg = torch.autograd.grad(ref, [xr] + list(blk.parameters()), cot)
    import copy
    gblk = copy.deepcopy(blk).to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = gblk(xg)
    close(out, ref, msg="block fwd")
    names = ["x"] + [n for n, _ in blk.named_parameters()]
    for n, a, r in zip(names, torch.autograd.grad(out, [xg] + list(gblk.parameters()), cot.to(DEV)), rg):
        close(a, r, msg=f"block d{n}")


def test_sincos_epilogue_accuracy(ops):
    """The sin/cos pair of the block activation (common.h sincos_f), read back exactly through an identity 1x1 conv (MFMA f32
    is exact, so the epilogue sees x itself): against float64 sin/cos on moderate and on large arguments."""
    from boosting_nerv_amd import _lib as L
    C, H, W = 12, 64, 128
    g = torch.Generator().manual_seed(2)
    for span in (8.0, 60.0, 8000.0, 3.0e4):
        x = ((torch.rand(1, C, H, W, generator=g) * 2 - 1) * span).to(DEV)
        w = torch.eye(C).reshape(C, C, 1, 1).to(DEV)
        s_out, c_out = torch.empty_like(x), torch.empty_like(x)
        ops._conv(x, w, None, s_out, B=1, Cin=C, Cout=C, H=H, W=W, k=1, in_mode=L.IN_PLAIN, ep_mode=L.EP_BIAS_SIN, out2=c_out)
        xd = x.double().cpu()
        es = (s_out.double().cpu() - torch.sin(xd)).abs().max().item()
        ec = (c_out.double().cpu() - torch.cos(xd)).abs().max().item()
        assert es < 3e-7 and ec < 3e-7, (span, es, ec)


def test_direct_epilogue_stores_are_stable_under_load(ops):
    """The kernels that store straight from the accumulators (conv_lean / conv_lean2) follow every 128-bit buffer store with wait
    states (store-data hazard on gfx950).  Timing-dependent corruption would show up as launch-to-launch differences:
    every epilogue family is launched repeatedly, back to back with a bandwidth-heavy copy on a second stream, and each result
    must equal the first bit for bit and agree with the float64 reference."""
    from boosting_nerv_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device=DEV)
    for (C, H, W) in ((12, 180, 320), (38, 72, 128)):
        x = torch.randn(1, C, H, W, generator=g).to(DEV)
        w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(DEV)
        b = torch.randn(C, generator=g).to(DEV)
        sc, sh = (torch.randn(1, C, generator=g) * 0.3).to(DEV), (torch.randn(1, C, generator=g) * 0.3).to(DEV)
        a0 = torch.randn(1, C, H, W, generator=g).to(DEV)
        pre = F.conv2d((x * (1 + sc[:, :, None, None]) + sh[:, :, None, None]).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
        pre_plain = F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
        refs = {"gelu": (F.gelu(pre), None), "res": (pre + a0.double().cpu(), None), "sin": (torch.sin(pre_plain), torch.cos(pre_plain))}
        kw = dict(B=1, Cin=C, Cout=C, H=H, W=W, k=3)
        aff = dict(in_mode=L.IN_AFFINE, scale=sc, shift=sh)
        for name, ep, extra in (("gelu", L.EP_BIAS_GELU, aff), ("res", L.EP_BIAS_RES, dict(aux0=a0, **aff)), ("sin", L.EP_BIAS_SIN, dict(in_mode=L.IN_PLAIN))):
            first = None
            for it in range(40):
                o, o2 = torch.empty_like(x), torch.empty_like(x)
                with torch.cuda.stream(side):
                    big.add_(1.0)
                ops._conv(x, w, b, o, ep_mode=ep, out2=None if name == "res" else o2, **extra, **kw)
                if first is None:
                    first = (o.clone(), o2.clone())
                    close(o, refs[name][0].float(), msg=f"{name} C={C}")
                else:
                    assert torch.equal(o, first[0]) and (name == "res" or torch.equal(o2, first[1])), (name, C, it)
    torch.cuda.synchronize()


def test_gelu_pair_epilogue_accuracy(ops):
    """gelu(v) and gelu'(v) of the TAT conv0 epilogue (common.h gelu_pair_f: one v_exp_f32, one v_rcp_f32, A-S 7.1.26 erf), read
    back through an identity 3x3 conv with a unit affine prologue: against float64 on small, moderate and tail arguments."""
    from boosting_nerv_amd import _lib as L
    C, H, W = 12, 64, 128
    g = torch.Generator().manual_seed(3)
    w = torch.zeros(C, C, 3, 3)
    w[torch.arange(C), torch.arange(C), 1, 1] = 1.0
    w = w.to(DEV)
    zero = torch.zeros(1, C, device=DEV)
    for span in (0.5, 4.0, 12.0, 40.0):
        x = ((torch.rand(1, C, H, W, generator=g) * 2 - 1) * span).to(DEV)
        h, gp = torch.empty_like(x), torch.empty_like(x)
        ops._conv(x, w, None, h, B=1, Cin=C, Cout=C, H=H, W=W, k=3, in_mode=L.IN_AFFINE, ep_mode=L.EP_BIAS_GELU, scale=zero, shift=zero, out2=gp)
        xd = x.double().cpu()
        cdf = 0.5 * (1 + torch.erf(xd / math.sqrt(2.0)))
        pdf = torch.exp(-0.5 * xd * xd) / math.sqrt(2 * math.pi)
        eh = ((h.double().cpu() - xd * cdf).abs() / (1 + xd.abs())).max().item()
        eg = (gp.double().cpu() - (cdf + xd * pdf)).abs().max().item()
        assert eh < 3e-7 and eg < 6e-7, (span, eh, eg)


@pytest.mark.parametrize("training", [True, False])
def test_cem_fused_quantise_rate_vs_oracle(ops, training):
    """bnerv_cem_scale_fwd / _bwd against the CPU restatement (oracle/cem_ref.py, itself pinned to the reference): bits, mean, std,
    dequant and the gradients wrt every tensor and scale of L = sum_i a_i * bits_i + <cot_i, dequant_i>, over tensors of very
    different sizes and spreads (tiny biases, a 200k-element weight, a nearly constant tensor that drives bits to the 0 floor)."""
    from oracle import cem_ref
    g = torch.Generator().manual_seed(21)
    shapes = [(24, 12, 3, 3), (24,), (750, 30, 3, 3), (3,), (64, 160, 1, 1), (16,)]
    ws = [(torch.randn(*s, generator=g) * (0.05 if i != 5 else 1e-2) + (0.0 if i != 5 else 0.3)).requires_grad_(True) for i, s in enumerate(shapes)]
    scales = [cem_ref.scale_init(w.detach(), 8, True).reshape(1).clone().requires_grad_(True) for w in ws]
    scales[5] = torch.tensor([0.2], requires_grad=True)        # codes 1.5 +- 0.05: sigma << 1, the rate hits the LowerBound floor
    noises = [torch.rand(w.shape, generator=g) - 0.5 for w in ws]
    cots = [torch.randn(w.shape, generator=g) for w in ws]
    a = torch.tensor([1.0, 0.5, 2.0, -1.0, 0.25, 1.5])         # a negative weight exercises the `grad < 0` arm of LowerBound
    ref_bits, ref_deq = [], []
    for w, s, z in zip(ws, scales, noises):
        code, quant, deq = cem_ref.scale_t(w, s)
        ref_bits.append(cem_ref.cal_bitrate(code, quant, training, noise=z)["bitrate"])
        ref_deq.append(deq)
    ref_loss = sum(ai * b for ai, b in zip(a, ref_bits)) + sum((d * c).sum() for d, c in zip(ref_deq, cots))
    ref_g = torch.autograd.grad(ref_loss, ws + scales)
    wg, sg = [gpu(w) for w in ws], [gpu(s) for s in scales]
    bits, stats, deq = ops.cem_scale_rate(wg, sg, [z.to(DEV) for z in noises] if training else [None] * len(ws), training)
    for i in range(len(ws)):
        assert abs(bits[i].item() - ref_bits[i].item()) <= 2e-4 * abs(ref_bits[i].item()) + 1e-2, (i, bits[i].item(), ref_bits[i].item())
        torch.testing.assert_close(deq[i].cpu(), ref_deq[i].detach(), rtol=0, atol=0)
    loss = (bits * a.to(DEV)).sum() + sum((d * c.to(DEV)).sum() for d, c in zip(deq, cots))
    got = torch.autograd.grad(loss, wg + sg)
    for i, (x, r) in enumerate(zip(got, ref_g)):
        close(x, r, rtol=2e-3, atol=2e-3 * float(r.abs().max()) + 1e-6, msg=f"cem grad {i}")


@pytest.mark.parametrize("dim", [64, 16, 32, 48])
def test_convnext_block_against_reference_golden(dim):
    """Row N3: the whole ConvNeXt Block on HIP kernels (depthwise 7x7, channel LayerNorm, ONE fused pwconv1 -> GELU -> pwconv2 ->
    gamma -> + input kernel; backward through the fused data-gradient kernel and two k = 1 weight-gradient launches) against the
    REFERENCE's channels_last Block (tests/golden/blocks_cnx.npz, oracle/make_goldens.py gen_convnext_blocks): output, dx and
    every parameter gradient."""
    from boosting_nerv_amd import model_blocks as mb
    npz = load_golden("blocks_cnx.npz")
    b = group(npz, f"cnx{dim}/")
    blk = mb.Block(dim=dim, drop_path=0.0, layer_scale_init_value=0.5)
    blk.load_state_dict({k[3:]: v for k, v in b.items() if k.startswith("sd/")})
    blk.to(DEV)
    x = b["x"].to(DEV).requires_grad_(True)
    y = blk(x)
    close(y, b["y"], msg=f"cnx{dim} fwd")
    params = dict(blk.named_parameters())
    gs = torch.autograd.grad(y, [x] + list(params.values()), b["cot"].to(DEV))
    close(gs[0], b["dx"], msg=f"cnx{dim} dx")
    for (pn, _), gval in zip(params.items(), gs[1:]):
        close(gval, b[f"grad/{pn}"], msg=f"cnx{dim} grad {pn}")
    with torch.no_grad():                                   # decode-style call: nothing saved, same values
        assert torch.equal(blk(x.detach()), y.detach())


def test_dense_gemm_shapes(ops):
    """The MFMA GEMM behind dense layers with >= 16 rows and the patchify convs: ragged sizes (no multiple of the 32 x 32 block
    tile, K not a multiple of 16), every activation, with and without bias / input gradient."""
    g = torch.Generator().manual_seed(21)
    for (B, I, O, act, bias) in ((144, 128, 128, "relu", True), (17, 75, 64, "none", True), (300, 33, 5, "sin", False), (64, 16, 1152, "sin", True), (16, 1, 3, "none", True),
                                  (9000, 75, 64, "none", True)):       # rows beyond the split threshold: row-split weight gradient (patchify conv)
        x = torch.randn(B, I, generator=g).requires_grad_(True)
        w = (torch.randn(O, I, generator=g) / math.sqrt(I)).requires_grad_(True)
        b = torch.randn(O, generator=g).requires_grad_(True) if bias else None
        ref = cpu_ref._act(F.linear(x, w, b), act)
        cot = torch.randn(B, O, generator=g)
        leaves = [x, w] + ([b] if bias else [])
        rg = torch.autograd.grad(ref, leaves, cot)
        gl = [gpu(t) for t in leaves]
        out = ops.dense_gemm(gl[0], gl[1], gl[2] if bias else None, act)
        close(out, ref, msg=f"gemm fwd {B}x{I}x{O}")
        for n, a, r in zip("xwb", torch.autograd.grad(out, gl, cot.to(DEV)), rg):
            close(a, r, msg=f"gemm d{n} {B}x{I}x{O}")


@pytest.mark.parametrize("min_items", ["1", "128"])          # 1: up to 3 cout tiles per block; 128 (default): small shapes split to 1
@pytest.mark.parametrize("shape", [(1, 38, 24, 64), (2, 55, 17, 36), (1, 95, 10, 44), (2, 22, 9, 40), (1, 30, 16, 32), (2, 38, 21, 100), (1, 177, 10, 36)])   # (interior tiles; C4's 177-channel stage)
def test_wide_split_conv_kernel_tat_block(ops, shape, min_items, monkeypatch):
    """The wide split-16-bit conv kernel (csrc/convbf.hip conv_bfw_kernel: several cout tiles / K chunks per staged input tile,
    bf16x6 products with f32 accumulation) on the TAT block -- its four launches cover the affine -> gelu-pair, affine -> residual,
    dGELU-saved and dSIN modes -- against the oracle, with the tile-count threshold lowered so that small shapes reach it."""
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_TILES", "1")
    monkeypatch.setenv("BNERV_SMALL", "0")                 # (small images would go to the low-resolution family, convs.hip: these tests are about the split kernels)
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_ITEMS", min_items)
    x0, mods, w0, b0, w1, b1, g = _tat_inputs(*shape, seed=7)
    ref = _tat_ref(x0, mods, w0, b0, w1, b1)
    cot = torch.randn(ref.shape, generator=g)
    leaves = [x0] + mods + [w0, b0, w1, b1]
    rg = torch.autograd.grad(ref, leaves, cot)
    gl = [gpu(t) for t in leaves]
    out = ops.tat_block(*gl)
    close(out, ref, msg="wide tat fwd")
    for n, a, r in zip(["x0", "s0", "t0", "s1", "t1", "w0", "b0", "w1", "b1"], torch.autograd.grad(out, gl, cot.to(DEV)), rg):
        close(a, r, msg=f"wide tat d{n}")


@pytest.mark.parametrize("min_items", ["1", "128"])
@pytest.mark.parametrize("case", [(1, 38, 38, 24, 64), (1, 70, 18, 8, 32), (2, 55, 55, 17, 36), (1, 20, 95, 9, 32), (1, 95, 12, 16, 32), (2, 46, 64, 22, 132), (1, 30, 160, 8, 32)])   # (last: split-K data gradient)
def test_wide_split_conv_kernel_plain(ops, case, min_items, monkeypatch):
    """Same kernel through conv2d_ps (plain -> bias forward, plain data gradient) and the sin block conv, incl. Cout <= 16 with several
    K chunks and Cin <= 16 with several cout tiles."""
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_TILES", "1")
    monkeypatch.setenv("BNERV_SMALL", "0")                 # (small images would go to the low-resolution family, convs.hip: these tests are about the split kernels)
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_ITEMS", min_items)
    B, Cin, Ct, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Ct, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.randn(Ct, generator=g).requires_grad_(True)
    ref = cpu_ref.upconv(x, w, b, 1)
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    out = ops.conv2d_ps(xg, wg, bg, 1)
    close(out, ref, msg="wide conv fwd")
    for n, a, r in zip("xwb", torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV)), rg):
        close(a, r, msg=f"wide conv d{n}")


@pytest.mark.parametrize("min_items", ["1", "128"])
@pytest.mark.parametrize("case", [(1, 12, 48, 16, 32), (2, 38, 152, 9, 32), (1, 20, 36, 17, 40), (1, 46, 184, 8, 32), (1, 22, 88, 14, 100), (1, 24, 256, 8, 32)])   # (last: split-K data gradient)
def test_wide_split_conv_kernel_upconv_ps2(ops, case, min_items, monkeypatch):
    """Up-conv + PixelShuffle(2) through the wide split kernel: forward with the pair-up epilogue (plain and sin/cos), data gradient
    through the unshuffle(2) prologue; whole SNeRV block as well (its up-conv, TAT convs and every gradient)."""
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_TILES", "1")
    monkeypatch.setenv("BNERV_SMALL", "0")                 # (small images would go to the low-resolution family, convs.hip: these tests are about the split kernels)
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_ITEMS", min_items)
    B, Cin, Ct, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Ct, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.randn(Ct, generator=g).requires_grad_(True)
    ref = cpu_ref.upconv(x, w, b, 2)
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    out = ops.conv2d_ps(xg, wg, bg, 2)
    close(out, ref, msg="ps2 conv fwd")
    for n, a, r in zip("xwb", torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV)), rg):
        close(a, r, msg=f"ps2 conv d{n}")
    Cc = Ct // 4
    x0, mods, w0, b0, w1, b1, g2 = _tat_inputs(B, Cc, 2 * H, 2 * W, seed=11)
    ref2 = _tat_ref(torch.sin(cpu_ref.upconv(x, w, b, 2)), mods, w0, b0, w1, b1)
    cot2 = torch.randn(ref2.shape, generator=g2)
    leaves = [x, w, b] + mods + [w0, b0, w1, b1]
    rg2 = torch.autograd.grad(ref2, leaves, cot2)
    gl = [gpu(t) for t in leaves]
    out2 = ops.snerv_block(*gl, 2)
    close(out2, ref2, msg="ps2 snerv fwd")
    for n, a, r in zip(["x", "wu", "bu", "s0", "t0", "s1", "t1", "w0", "b0", "w1", "b1"], torch.autograd.grad(out2, gl, cot2.to(DEV)), rg2):
        close(a, r, msg=f"ps2 snerv d{n}")


@pytest.mark.parametrize("min_items", ["1", "128"])
@pytest.mark.parametrize("case", [(1, 20, 45, 9, 32, 3), (1, 17, 50, 8, 32, 5), (2, 40, 90, 14, 100, 3), (1, 33, 175, 11, 36, 5), (1, 79, 594, 10, 32, 3),
                                  (1, 40, 288, 9, 32, 3)])   # (the last two: data gradients with >= 8 K chunks on few tiles -> the split-K plan)
def test_wide_split_conv_kernel_upconv_ps35(ops, case, min_items, monkeypatch):
    """Up-conv + PixelShuffle(3 / 5) forward through the wide split kernel's scatter-store epilogue (plain and sin / cos), gradients
    through the kernels that own them."""
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_TILES", "1")
    monkeypatch.setenv("BNERV_SMALL", "0")                 # (small images would go to the low-resolution family, convs.hip: these tests are about the split kernels)
    monkeypatch.setenv("BNERV_SPLIT_WIDE_MIN_ITEMS", min_items)
    B, Cin, Ct, H, W, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Ct, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.randn(Ct, generator=g).requires_grad_(True)
    ref = cpu_ref.upconv(x, w, b, s)
    cot = torch.randn(ref.shape, generator=g)
    rg = torch.autograd.grad(ref, [x, w, b], cot)
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    out = ops.conv2d_ps(xg, wg, bg, s)
    close(out, ref, msg=f"ps{s} conv fwd")
    for n, a, r in zip("xwb", torch.autograd.grad(out, [xg, wg, bg], cot.to(DEV)), rg):
        close(a, r, msg=f"ps{s} conv d{n}")
    Cc = Ct // (s * s)
    x0, mods, w0, b0, w1, b1, g2 = _tat_inputs(B, Cc, s * H, s * W, seed=13)
    ref2 = _tat_ref(torch.sin(cpu_ref.upconv(x, w, b, s)), mods, w0, b0, w1, b1)
    cot2 = torch.randn(ref2.shape, generator=g2)
    leaves = [x, w, b] + mods + [w0, b0, w1, b1]
    rg2 = torch.autograd.grad(ref2, leaves, cot2)
    gl = [gpu(t) for t in leaves]
    out2 = ops.snerv_block(*gl, s)
    close(out2, ref2, msg=f"ps{s} snerv fwd")
    for n, a, r in zip(["x", "wu", "bu", "s0", "t0", "s1", "t1", "w0", "b0", "w1", "b1"], torch.autograd.grad(out2, gl, cot2.to(DEV)), rg2):
        close(a, r, msg=f"ps{s} snerv d{n}")


def test_wide_split_kernels_keep_the_f32_contract():
    """tools/split_contract.py: per output element |kernel - float64| <= 3e-7 * sum|a||b| for the forward, data-gradient and
    weight-gradient launches of the wide split kernels (38 -> 38, 55 -> 55, 12 -> 48 + PixelShuffle(2), 46 -> 184, affine / residual
    modes) in the default bf16x6 arithmetic -- and the SAME check must fail for BNERV_SPLIT_WIDE=bf16x3 (two pieces, ~2e-6), so a
    silent downgrade of the split cannot pass the suite.  (The mode is read once per process: subprocesses.)"""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "split_contract.py")
    env = dict(os.environ)
    env.pop("BNERV_SPLIT_WIDE", None)
    r6 = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600, env=env)
    assert r6.returncode == 0, r6.stdout[-3000:] + r6.stderr[-2000:]
    env["BNERV_SPLIT_WIDE"] = "bf16x3"
    r3 = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600, env=env)
    assert r3.returncode == 1 and "OUTSIDE" in r3.stdout, r3.stdout[-3000:] + r3.stderr[-2000:]


@pytest.mark.parametrize("shape", [(1, 3, 720, 1280), (2, 3, 176, 208), (1, 1, 192, 352), (1, 3, 1080, 1920), (2, 3, 180, 270)])
def test_fused_msssim_launches_equal_the_level_by_level_form(ops, shape, monkeypatch):
    """(The last two shapes have ODD pyramid levels -- 1080 -> ... -> 135 -> 68, and odd from level 0: there the padded 2x2 means stay a
    launch per level; the statistics of the five levels share one launch (round 4) and, since round 5, the gradient takes the same two
    launches as on an even pyramid -- the 0.25-chain walks the padded cells ((y + pad) / 2 per level).)
    Frames whose pyramid has even sides take the fused MS-SSIM launches (one pyramid kernel, one statistics launch for all five
    levels, the coarser levels' gradients in one launch and their 0.25-chain evaluated inside the level-0 launch: 5 launches instead of
    15).  Same formulas on the same data; the compiler contracts a few multiply-adds differently in the two forms, so the comparison
    with the level-by-level form (BNERV_LOSS_FUSED=0) allows rounding: values to 4 ulp, the gradient to 3e-5 of its largest entry."""
    g = torch.Generator().manual_seed(sum(shape))
    tgt = torch.rand(*shape, generator=g).to(DEV)
    pred = (tgt + 0.1 * torch.randn(*shape, generator=g).to(DEV)).clamp(0, 1)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BNERV_LOSS_FUSED", mode)
        loss, stats, grad = ops.loss_value_grad_stats(pred, tgt, "Fusion10_freq")
        out[mode] = (loss.clone(), stats.clone(), ops.msssim(pred, tgt).clone(), grad.clone())
    for a, b in zip(out["1"][:3], out["0"][:3]):
        torch.testing.assert_close(a, b, rtol=5e-7, atol=0)
    ga, gb = out["1"][3], out["0"][3]
    assert (ga - gb).abs().max().item() <= 3e-5 * gb.abs().max().item()      # (a0 + 2 x a1 + y a2 cancels: a different contraction shows at 1e-5 of the largest entry)


@pytest.mark.parametrize("shape", [(1, 3, 720, 1280), (2, 3, 176, 208), (1, 3, 1080, 1920), (2, 3, 180, 270)])
def test_merged_loss_launches_change_no_bit(ops, shape, monkeypatch):
    """Fusion10_freq: independent launches share one grid as block ranges (row FFTs | pyramid | L1 / L2 sums; column FFTs | MS-SSIM
    coefficients; adjoint row FFTs | coarse SSIM gradients; level-0 gradient | loss_final -- 5 launches instead of 10 on an even pyramid,
    9 instead of 16 on an odd one (the last two shapes), whose pooled levels stay a launch each).  The bodies are the same code in the same
    order (spectral gradient written first, level-0 SSIM gradient added to it), so the loss, the per-sample statistics and the gradient are
    BIT-equal to the one-launch-per-kernel form (BNERV_LOSS_MERGED=0)."""
    g = torch.Generator().manual_seed(7 + sum(shape))
    tgt = torch.rand(*shape, generator=g).to(DEV)
    pred = (tgt + 0.1 * torch.randn(*shape, generator=g).to(DEV)).clamp(0, 1)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BNERV_LOSS_MERGED", mode)
        loss, stats, grad = ops.loss_value_grad_stats(pred, tgt, "Fusion10_freq")
        out[mode] = (loss.clone(), stats.clone(), grad.clone())
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(1, 3, 720, 1280), (1, 3, 1080, 1920), (2, 3, 180, 270)])
def test_spectral_gradient_first_or_last_is_the_same_gradient(ops, shape, monkeypatch):
    """The adjoint row FFTs either write the spectral gradient before the level-0 SSIM launch adds to it (default: they share the coarse
    SSIM launch) or accumulate onto the finished gradient as a last launch (BNERV_LOSS_ADJ=late, round 4's order).  Same terms, one rounding
    apart (d + k r against fma(k, r, d)): loss and statistics bit-equal, the gradient within 2e-7 of its largest entry -- in the merged and
    in the one-launch-per-kernel form."""
    g = torch.Generator().manual_seed(11 + sum(shape))
    tgt = torch.rand(*shape, generator=g).to(DEV)
    pred = (tgt + 0.1 * torch.randn(*shape, generator=g).to(DEV)).clamp(0, 1)
    for merged in ("1", "0"):
        monkeypatch.setenv("BNERV_LOSS_MERGED", merged)
        out = {}
        for mode in ("early", "late"):
            monkeypatch.setenv("BNERV_LOSS_ADJ", mode)
            loss, stats, grad = ops.loss_value_grad_stats(pred, tgt, "Fusion10_freq")
            out[mode] = (loss.clone(), stats.clone(), grad.clone())
        assert torch.equal(out["early"][0], out["late"][0]) and torch.equal(out["early"][1], out["late"][1])
        ga, gb = out["early"][2], out["late"][2]
        assert (ga - gb).abs().max().item() <= 2e-7 * gb.abs().max().item(), (merged, (ga - gb).abs().max().item(), gb.abs().max().item())


def test_msssim_kernel_against_independent_form(ops):
    """bnerv_msssim (the kernels behind the 0.3 * (1 - ms_ssim) term of Fusion10_freq, hnerv_utils.py:369-370, and the MS-SSIM eval
    metric, :410-412) against the independent float64 form of tests/msssim_independent.py -- direct 2-D window, written from the
    definition -- at a small, an odd-sided and the 720x1280 size, and the identities ms_ssim(x, x) = 1 and symmetry."""
    import msssim_independent as ind
    for shape, seed in (((2, 3, 176, 208), 1), ((1, 3, 177, 203), 2), ((1, 1, 161, 161), 3), ((1, 3, 720, 1280), 4)):
        g = torch.Generator().manual_seed(seed)
        x = F.avg_pool2d(torch.rand(*shape, generator=g), 3, stride=1, padding=1)
        y = (x + 0.08 * torch.randn(*shape, generator=g)).clamp(0, 1)
        want = ind.ms_ssim(x.numpy(), y.numpy())
        got = ops.msssim(x.to(DEV), y.to(DEV)).double().cpu().numpy()
        assert np.abs(got - want).max() < 2e-5, (shape, got, want)
        assert np.abs(ops.msssim(y.to(DEV), x.to(DEV)).double().cpu().numpy() - got).max() < 2e-6, shape
        assert np.abs(ops.msssim(x.to(DEV), x.to(DEV)).double().cpu().numpy() - 1.0).max() < 2e-6, shape
        assert np.abs(msssim_ref.ms_ssim(x, y, data_range=1, size_average=False).double().numpy() - want).max() < 1e-5, shape


def test_wide_split_kernels_random_shapes():
    """tools/fuzz_wide.py: random (B, Cin, Cout, H, W) for plain / PixelShuffle(2) convs and TAT blocks through the wide split kernels
    (both cout-tile policies) against float64 torch references -- outputs and every gradient within 2e-5 of the tensor's max."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_wide.py")
    r = subprocess.run([sys.executable, tool, "60", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_low_resolution_family_random_shapes():
    """The same sweep over the shapes convs.hip takes (Cin <= 32, PixelShuffle(2 / 3 / 5) epilogues, the unshuffle(2) prologue of the
    up-conv data gradients up to 64 channels, TAT blocks with their slab reductions)."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_wide.py")
    r = subprocess.run([sys.executable, tool, "80", "11", "small"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
