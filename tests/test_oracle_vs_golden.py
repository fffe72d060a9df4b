"""CPU suite: the oracle (oracle/cpu_ref.py) against golden vectors produced by the REAL reference
(oracle/make_goldens.py importing /root/reference).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import check_summary, group, load_golden
from oracle import cpu_ref, configs

RT, AT = 1e-4, 1e-5      # CPU-vs-CPU, same ATen kernels: tight


def test_pe_f32_f64_and_xy():
    g = load_golden("pe.npz")
    bases = cpu_ref.pe_bases("pe_1.25_80", "pi")
    assert np.array_equal(bases.numpy(), g["bases"])                       # table must be bit-identical
    t64 = torch.from_numpy(g["t64"])
    assert np.array_equal(cpu_ref.pe(t64[:, None].float(), bases).view(3, -1).numpy(), g["out_f32"])
    assert np.array_equal(cpu_ref.pe(t64[:, None], bases).float().view(3, -1).numpy(), g["out_f64"])
    assert np.array_equal(cpu_ref.pe(torch.from_numpy(g["xy"]), bases).view(16, -1).numpy(), g["out_xy"])
    # the two call forms really differ in the chaotic high bands (SURVEY hard part 3) -- both must be kept
    assert np.abs(g["out_f32"] - g["out_f64"]).max() > 0.1


def _run_block(fn, b):
    sd = {k[3:]: v.clone().requires_grad_(True) for k, v in b.items() if k.startswith("sd/")}
    x = b["x"].clone().requires_grad_(True)
    z = b["z"].clone().requires_grad_(True) if "z" in b else None
    y = fn(x, z, sd)
    torch.testing.assert_close(y, b["y"], rtol=RT, atol=AT)
    ins = [x] + ([z] if z is not None else [])
    names = list(sd)
    gs = torch.autograd.grad(y, ins + [sd[n] for n in names], b["cot"])
    torch.testing.assert_close(gs[0], b["dx"], rtol=RT, atol=AT)
    if z is not None:
        torch.testing.assert_close(gs[1], b["dz"], rtol=RT, atol=1e-4)
    for n, gval in zip(names, gs[len(ins):]):
        torch.testing.assert_close(gval, b[f"grad/{n}"], rtol=1e-3, atol=1e-4, msg=lambda m, n=n: f"{n}: {m}")


def test_blocks():
    npz = load_golden("blocks.npz")
    _run_block(lambda x, z, sd: cpu_ref.sft_affine(x, *cpu_ref.sft_scale_shift(z, sd, "")), _strip(group(npz, "sft_c12/")))
    _run_block(lambda x, z, sd: cpu_ref.tat_block(x, z, _pref(sd, "b"), "b"), group(npz, "tat_c15/"))
    for name in ("blk_s1_k3_c12", "blk_s2_k3_c15_12", "blk_s3_k3_c9_7", "blk_s5_k1_c30", "blk_s2_k1_c20_33"):
        _run_block(lambda x, z, sd: cpu_ref.nerv_block(x, z, _pref(sd, "b"), "b"), group(npz, name + "/"))
    _run_block(lambda x, z, sd: cpu_ref.mlp(x, _pref(sd, "m"), "m", 2, "sin"), group(npz, "mlp_sin/"))
    torch.testing.assert_close(cpu_ref.out_img(torch.from_numpy(npz["outimg/x"])), torch.from_numpy(npz["outimg/y"]))


def _strip(b):
    # SFTLayer state_dict keys have no prefix; cpu_ref.sft_scale_shift wants "<prefix>.<name>"
    return {(("sd/." + k[3:]) if k.startswith("sd/") else (("grad/." + k[5:]) if k.startswith("grad/") else k)): v
            for k, v in b.items()}


def _pref(sd, p):
    return {f"{p}.{k}": v for k, v in sd.items()}


def test_conv_up_block_and_hnerv_dec0():
    """ENeRV layer 0 and HNeRV decoder[0] go through the whole-model restatements; pin their pieces here."""
    npz = load_golden("blocks.npz")
    b = group(npz, "conv_up_block/")

    def cub(x, z, sd):
        import math
        import torch.nn.functional as F
        s = int(round(math.sqrt(sd["conv1.upconv.0.weight"].shape[0] / sd["conv2.weight"].shape[1])))
        y = cpu_ref.upconv(x, sd["conv1.upconv.0.weight"], sd["conv1.upconv.0.bias"], s)
        x0 = torch.sin(F.conv2d(y, sd["conv2.weight"], sd["conv2.bias"], padding=1))
        return cpu_ref.tat_block(x0, z, sd, "sft_block")
    _run_block(cub, b)
    b = group(npz, "hnerv_dec0/")

    def d0(x, z, sd):
        import torch.nn.functional as F
        x0 = torch.sin(F.conv2d(x, sd["conv.downconv.weight"], sd["conv.downconv.bias"]))
        return cpu_ref.tat_block(x0, z, sd, "sft_block")
    _run_block(d0, b)


@pytest.mark.parametrize("name", ["tiny_nerv", "tiny_enerv", "tiny_hnerv"])
def test_tiny_models(name):
    npz = load_golden(name + ".npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in group(npz, "sd/").items()}
    frame = torch.rand(2, 3, 180, 320, generator=torch.Generator().manual_seed(int(npz["frame_seed"])))
    norm_idx = torch.from_numpy(npz["norm_idx"])
    if name == "tiny_nerv":
        img, lst = cpu_ref.nerv_boost_forward(sd, norm_idx, return_list=True)
    elif name == "tiny_enerv":
        img, lst = cpu_ref.enerv_boost_forward(sd, norm_idx, return_list=True)
    else:
        img, lst = cpu_ref.hnerv_boost_forward(sd, frame, norm_idx, return_list=True)
    check_summary(img, npz, "img", RT, AT)
    for i, t in enumerate(lst):
        check_summary(t, npz, f"list{i}", RT, AT)
    loss = cpu_ref.loss_fn(img, frame, "L1_freq")
    assert abs(loss.item() - float(npz["loss_L1_freq"])) < 1e-4 * abs(float(npz["loss_L1_freq"]))
    torch.testing.assert_close(cpu_ref.psnr_fn_single(img, frame), torch.from_numpy(npz["psnr"]), rtol=1e-5, atol=1e-4)
    loss.backward()
    for k, p in sd.items():
        gn = float(npz[f"gnorm/{k}"]) if f"gnorm/{k}" in npz.files else None
        if gn is None or gn < 0:
            continue
        assert abs(p.grad.double().norm().item() - gn) <= 2e-3 * gn + 1e-7, k
        if f"grad/{k}" in npz.files:
            torch.testing.assert_close(p.grad, torch.from_numpy(npz[f"grad/{k}"]), rtol=2e-3, atol=1e-5 + 1e-4 * gn)


def test_loss_variants():
    npz = load_golden("loss.npz")
    for tag in ("small", "odd"):
        tgt = torch.from_numpy(npz[f"{tag}/target"])
        for lt in ("L1", "L2", "L1_freq", "Fusion10_freq"):
            pred = torch.from_numpy(npz[f"{tag}/pred"]).clone().requires_grad_(True)
            l = cpu_ref.loss_fn(pred, tgt, lt)
            assert abs(l.item() - float(npz[f"{tag}/{lt}/loss"])) <= 1e-5 * abs(float(npz[f"{tag}/{lt}/loss"])), (tag, lt)
            l.backward()
            check_summary(pred.grad, npz, f"{tag}/{lt}/grad", 1e-4, 1e-7)
        torch.testing.assert_close(cpu_ref.psnr_fn_single(torch.from_numpy(npz[f"{tag}/pred"]), tgt), torch.from_numpy(npz[f"{tag}/psnr"]))


def test_adan_trajectory_and_lr():
    npz = load_golden("optim.npz")
    params = [torch.from_numpy(npz[f"p0/{i}"]).clone() for i in range(3)]
    st = cpu_ref.AdanState(params, lr=0.003)
    for step in range(6):
        st.lr = 0.003 * (0.1 + 0.15 * step)
        st.step([torch.from_numpy(npz[f"g{step}/{i}"]) for i in range(3)])
        for i in range(3):
            torch.testing.assert_close(params[i], torch.from_numpy(npz[f"p{step + 1}/{i}"]), rtol=2e-5, atol=2e-6)
    for x, c, h in zip(npz["lr/x"], npz["lr/cosine"], npz["lr/hybrid"]):
        assert abs(0.003 * cpu_ref.lr_mult(float(x), "cosine_0.1_1_0.1") - c) < 1e-15
        assert abs(0.003 * cpu_ref.lr_mult(float(x), "hybrid_0.2_1_2_0.1_0.05") - h) < 1e-15


def test_cem_quantisers_and_rate_against_reference():
    """oracle/cem_ref.py against the reference's lib/transform_ops.py (Scale_T, ScaleBeta_T) and lib/entropy_model.py
    (DiffEntropyModel, training draw and evaluation form): values and gradients."""
    from oracle import cem_ref
    npz = load_golden("cem.npz")
    for name in ("w", "b"):
        x = torch.from_numpy(npz[f"scale/{name}/x"]).clone().requires_grad_(True)
        scale = torch.from_numpy(npz[f"scale/{name}/scale"]).clone().requires_grad_(True)
        torch.testing.assert_close(cem_ref.scale_init(x.detach(), 8, True).reshape(1), scale.detach(), rtol=0, atol=0)
        code, quant, deq = cem_ref.scale_t(x, scale)
        for k, v in (("code", code), ("quant", quant), ("dequant", deq)):
            torch.testing.assert_close(v.detach(), torch.from_numpy(npz[f"scale/{name}/{k}"]), rtol=0, atol=0)
        cot = torch.from_numpy(npz[f"scale/{name}/cot"])
        gx, gs = torch.autograd.grad((deq * cot).sum(), [x, scale], retain_graph=True)
        torch.testing.assert_close(gx, torch.from_numpy(npz[f"scale/{name}/gx"]), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(gs, torch.from_numpy(npz[f"scale/{name}/gscale"]), rtol=1e-5, atol=1e-6)
        r = cem_ref.cal_bitrate(code, quant, True, noise=torch.from_numpy(npz[f"rate/{name}/noise"]))
        assert abs(r["bitrate"].item() - float(npz[f"rate/{name}/bitrate"])) <= 1e-5 * float(npz[f"rate/{name}/bitrate"])
        gx2, gs2 = torch.autograd.grad(r["bitrate"], [x, scale], retain_graph=True)
        torch.testing.assert_close(gx2, torch.from_numpy(npz[f"rate/{name}/gx"]), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gs2, torch.from_numpy(npz[f"rate/{name}/gscale"]), rtol=1e-4, atol=1e-2)
        ev = cem_ref.cal_bitrate(code, quant, False)["bitrate"]
        assert abs(ev.item() - float(npz[f"rate/{name}/eval_bits"])) <= 1e-5 * float(npz[f"rate/{name}/eval_bits"])
        # the ideal code length of the integers is what the estimate approximates (the 1e-5 floor and the [min,max] renormalisation apart)
        assert abs(cem_ref.ideal_bits(quant, r["mean"], r["std"]) - ev.item()) <= 0.05 * ev.item()
    x = torch.from_numpy(npz["scalebeta/x"]).clone().requires_grad_(True)
    scale = torch.from_numpy(npz["scalebeta/scale"]).clone().requires_grad_(True)
    beta = torch.from_numpy(npz["scalebeta/beta"]).clone().requires_grad_(True)
    s0, b0 = cem_ref.scalebeta_init(x.detach(), 8, False)
    torch.testing.assert_close(s0.reshape(1), scale.detach(), rtol=0, atol=0)
    torch.testing.assert_close(b0.reshape(1), beta.detach(), rtol=0, atol=0)
    code, quant, deq = cem_ref.scalebeta_t(x, scale, beta)
    for k, v in (("code", code), ("quant", quant), ("dequant", deq)):
        torch.testing.assert_close(v.detach(), torch.from_numpy(npz[f"scalebeta/{k}"]), rtol=0, atol=0)
    g = torch.autograd.grad((deq * torch.from_numpy(npz["scalebeta/cot"])).sum(), [x, scale, beta])
    for a, k in zip(g, ("gx", "gscale", "gbeta")):
        torch.testing.assert_close(a, torch.from_numpy(npz[f"scalebeta/{k}"]), rtol=1e-5, atol=1e-5)
