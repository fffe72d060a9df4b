"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz by running the REAL reference on CPU in this container.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_goldens

Reads /root/reference (never copied, never shipped); writes only small input/output vectors.  The fixtures are
data: seeded inputs, the reference's outputs for them, and (for tiny modules) the seeded weights the outputs
belong to.  Each group below names the reference entry point it pins.
"""
import hashlib
import os
import sys

import numpy as np
import torch

from . import configs, ref_harness

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def npf(t):
    return t.detach().cpu().numpy().copy()


def sd_np(module, prefix=""):
    return {prefix + k: npf(v) for k, v in module.state_dict().items()}


def sd_hash(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def sample_idx(n, k=256, seed=7):
    g = np.random.RandomState(seed)
    return np.sort(g.choice(n, size=min(k, n), replace=False)).astype(np.int64)


def summary(t, name, out, k=256):
    f = t.detach().flatten()
    idx = sample_idx(f.numel(), k)
    out[f"{name}.idx"] = idx
    out[f"{name}.val"] = npf(f[torch.from_numpy(idx)])
    out[f"{name}.mean"] = np.float64(f.double().mean().item())
    out[f"{name}.std"] = np.float64(f.double().std().item())
    out[f"{name}.shape"] = np.array(t.shape, dtype=np.int64)


# ---------------------------------------------------------------------------------------------------------------------
def gen_pe(R):
    """PositionEncoding('pe_1.25_80','pi')  (model_blocks.py:108-126), fp32 and fp64 call forms."""
    pe = R.model_blocks.PositionEncoding("pe_1.25_80", "pi")
    idx = [0, 36, 131]
    t64 = torch.tensor([(i + 1) / 132 for i in idx], dtype=torch.float64)
    o32 = pe(t64[:, None].float()).view(len(idx), -1)                       # model_nerv.py:47-48
    o64 = pe(t64[:, None]).float().view(len(idx), -1)                       # model_hnerv.py:241
    xy = (torch.arange(16) / 16)[:, None]                                   # model_enerv.py:281-291
    oxy = pe(xy).view(16, -1)
    np.savez(os.path.join(OUT, "pe.npz"), t64=npf(t64), bases=npf(pe.pe_bases), out_f32=npf(o32), out_f64=npf(o64),
             xy=npf(xy), out_xy=npf(oxy))


# ---------------------------------------------------------------------------------------------------------------------
def _grads(out, inputs, params, seed):
    g = torch.Generator().manual_seed(seed)
    cot = torch.randn(out.shape, generator=g)
    gs = torch.autograd.grad(out, list(inputs) + list(params), cot, allow_unused=True)
    return cot, gs


def gen_blocks(R):
    """Tiny-shape block goldens: SFTLayer, ResBlock_SFT, NeRVBlock(pshuffel_3x3, s in 1,2,3,5, k in 1,3), Conv_Up_Block,
    OutImg, NeRV_MLP  (model_blocks.py:14-105, :57-71; model_enerv.py:73-102)."""
    out = {}
    args = configs.tiny_nerv()
    mb = R.model_blocks

    def run(name, module, x, z=None, seed=0):
        x = x.clone().requires_grad_(True)
        ins = [x]
        if z is not None:
            z = z.clone().requires_grad_(True)
            ins.append(z)
            y = module((x, z))
        else:
            y = module(x)
        params = list(module.parameters())
        cot, gs = _grads(y, ins, params, seed)
        out[f"{name}/x"] = npf(x)
        if z is not None:
            out[f"{name}/z"] = npf(z)
        out[f"{name}/y"] = npf(y)
        out[f"{name}/cot"] = npf(cot)
        out[f"{name}/dx"] = npf(gs[0])
        if z is not None:
            out[f"{name}/dz"] = npf(gs[1])
        for (pn, _), g in zip(module.named_parameters(), gs[len(ins):]):
            out[f"{name}/grad/{pn}"] = npf(g)
        for k, v in sd_np(module).items():
            out[f"{name}/sd/{k}"] = v

    torch.manual_seed(11)
    run("sft_c12", mb.SFTLayer(32, 12, 1, "relu", 1, args=args), torch.randn(2, 12, 10, 14), torch.randn(2, 32, 1, 1), 1)
    torch.manual_seed(12)
    run("tat_c15", mb.ResBlock_SFT(15, 15, cond_ch=32, in_act="relu", out_act="gelu", omega=1, args=args),
        torch.randn(2, 15, 11, 13), torch.randn(2, 32, 1, 1), 2)
    cases = [("blk_s1_k3_c12", 12, 12, 1, 3, (9, 20)), ("blk_s2_k3_c15_12", 15, 12, 2, 3, (7, 10)),
             ("blk_s3_k3_c9_7", 9, 7, 3, 3, (5, 6)), ("blk_s5_k1_c30", 10, 10, 5, 1, (3, 4)),
             ("blk_s2_k1_c20_33", 20, 33, 2, 1, (6, 9))]
    for i, (name, ngf, new_ngf, s, k, hw) in enumerate(cases):
        torch.manual_seed(20 + i)
        blk = mb.NeRVBlock(dec_block=True, conv_type="pshuffel_3x3", ngf=ngf, new_ngf=new_ngf, ks=k, strd=s, bias=True,
                           norm="none", act="sin", sft_ngf=32, args=args)
        run(name, blk, torch.randn(2, ngf, *hw), torch.randn(2, 32, 1, 1), 30 + i)
    torch.manual_seed(40)
    cub = R.model_enerv.Conv_Up_Block(ngf=8, new_ngf=24, ks=3, stride=5, bias=True, norm="none", act="sin",
                                      conv_type="pshuffel_3x3", sft_ngf=32, args=args)
    run("conv_up_block", cub, torch.randn(2, 8, 3, 4), torch.randn(2, 32, 1, 1), 41)
    # HNeRV decoder[0]: DownConv('conv', ks=0, strd=1) + sin + TAT at the embedding resolution (model_hnerv.py:200-202)
    hargs = configs.tiny_hnerv()
    torch.manual_seed(42)
    d0 = mb.NeRVBlock(dec_block=False, conv_type="conv", ngf=4, new_ngf=10, ks=0, strd=1, bias=True, norm="none",
                      act="sin", sft_ngf=32, args=hargs)
    run("hnerv_dec0", d0, torch.randn(2, 4, 9, 16), torch.randn(2, 32, 1, 1), 43)
    torch.manual_seed(44)
    run("mlp_sin", mb.NeRV_MLP(dim_list=[160, 64, 32], bias=True, act="sin", omega=1, args=args), torch.randn(3, 160, 1, 1), None, 45)
    # OutImg
    x = torch.randn(2, 3, 5, 7)
    out["outimg/x"] = npf(x)
    out["outimg/y"] = npf(mb.OutImg(x, "tanh"))
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)


def gen_convnext_blocks(R):
    """ConvNeXt `Block` of the HNeRV_Boost encoder (model_blocks.py:223-247: dwconv 7x7 -> LayerNorm -> pwconv1 -> GELU -> pwconv2 ->
    gamma -> + input, channels_last inside) at the widths the recipes use (64, 16) and two more the fused kernel is built for,
    with a layer scale large enough for the MLP branch to matter in the comparison, plus the channels_first LayerNorm + patchify
    down-sampling pair of the stem (model_blocks.py:305-312)."""
    out = {}
    mb = R.model_blocks
    for i, (dim, B, H, W) in enumerate(((64, 1, 11, 19), (16, 2, 9, 16), (32, 1, 8, 8), (48, 1, 5, 7))):
        torch.manual_seed(60 + i)
        blk = mb.Block(dim=dim, drop_path=0.0, layer_scale_init_value=0.5)
        with torch.no_grad():                               # (the reference initialises with trunc_normal(0.02) / zero biases through
            for p_ in blk.parameters():                     #  ConvNeXt._init_weights; here every parameter gets signal)
                if p_.dim() > 0 and p_ is not blk.gamma:
                    p_.normal_(0.0, 0.2)
            blk.gamma.uniform_(0.2, 0.9)
        x = torch.randn(B, dim, H, W).requires_grad_(True)
        y = blk(x)
        cot, gs = _grads(y, [x], list(blk.parameters()), 70 + i)
        name = f"cnx{dim}"
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/cot"], out[f"{name}/dx"] = npf(x), npf(y), npf(cot), npf(gs[0])
        for (pn, _), g in zip(blk.named_parameters(), gs[1:]):
            out[f"{name}/grad/{pn}"] = npf(g)
        for k, v in sd_np(blk).items():
            out[f"{name}/sd/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "blocks_cnx.npz"), **out)


# ---------------------------------------------------------------------------------------------------------------------
def _model(R, args):
    if args.model == "NeRV_Boost":
        return R.model_nerv.NeRV_Boost(1, args=args)
    if args.model == "ENeRV_Boost":
        return R.model_enerv.ENeRV_Boost(3, args=args)
    return R.model_hnerv.HNeRV_Boost(args)


def gen_tiny_models(R):
    """Whole tiny models, seeded init: full state_dict + forward samples + loss + every parameter gradient's norm
    (model_nerv.py:45-61, model_enerv.py:279-317, model_hnerv.py:224-251; loss L1_freq hnerv_utils.py:378-385)."""
    for name, args, hw in (("tiny_nerv", configs.tiny_nerv(), (180, 320)), ("tiny_enerv", configs.tiny_enerv(), (180, 320)),
                           ("tiny_hnerv", configs.tiny_hnerv(), (180, 320))):
        torch.manual_seed(1)
        m = _model(R, args)
        out = {f"sd/{k}": v for k, v in sd_np(m).items()}
        out["sd_sha256"] = np.array(sd_hash(m.state_dict()))
        g = torch.Generator().manual_seed(5)
        frame = torch.rand(2, 3, *hw, generator=g)
        norm_idx = torch.tensor([3 / 7, 6 / 7], dtype=torch.float64)
        inp = frame if args.model == "HNeRV_Boost" else norm_idx
        img, lst, _ = m(inp, norm_idx=norm_idx)
        assert img.shape[-2:] == hw, img.shape
        loss = R.hnerv_utils.loss_fn(img, frame, "L1_freq")
        loss.backward()
        out["frame_seed"] = np.int64(5)
        out["norm_idx"] = npf(norm_idx)
        summary(img, "img", out, 1024)
        for i, t in enumerate(lst):
            summary(t, f"list{i}", out, 128)
        out["loss_L1_freq"] = np.float64(loss.item())
        out["psnr"] = npf(R.hnerv_utils.psnr_fn_single(img.detach(), frame))
        for pn, p in m.named_parameters():
            out[f"gnorm/{pn}"] = np.float64(p.grad.double().norm().item()) if p.grad is not None else np.float64(-1)
            if p.grad is not None and p.grad.numel() <= 4096:
                out[f"grad/{pn}"] = npf(p.grad)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)


def gen_full_models(R):
    """Real BASELINE shapes, torch.manual_seed(1) construction: state_dict checksum (pins init ORDER), parameter count,
    per-layer output summaries at one t, loss L1_freq and gradient norms.  C1 (NeRV-boost 1.5M 720p) only: the
    1080p models take minutes and >10 GB on CPU; their init checksum and parameter counts are still recorded."""
    out = {}
    for name, args in (("c1", configs.c1()), ("c3", configs.c3()), ("c4", configs.c4())):
        torch.manual_seed(1)
        m = _model(R, args)
        sd = m.state_dict()
        out[f"{name}/sd_sha256"] = np.array(sd_hash(sd))
        out[f"{name}/n_params"] = np.int64(sum(p.numel() for p in m.parameters()))
        out[f"{name}/n_tensors"] = np.int64(len(sd))
        out[f"{name}/keys"] = np.array(list(sd.keys()))
        out[f"{name}/shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        first = next(iter(sd.values()))
        out[f"{name}/first_vals"] = npf(first.flatten()[:8])
        if name == "c1":
            g = torch.Generator().manual_seed(5)
            frame = torch.rand(1, 3, 720, 1280, generator=g)
            norm_idx = torch.tensor([37 / 132], dtype=torch.float64)
            img, lst, _ = m(norm_idx, norm_idx=norm_idx)
            loss = R.hnerv_utils.loss_fn(img, frame, "L1_freq")
            loss.backward()
            summary(img, "c1/img", out, 2048)
            for i, t in enumerate(lst):
                summary(t, f"c1/list{i}", out, 128)
            out["c1/loss_L1_freq"] = np.float64(loss.item())
            out["c1/psnr"] = npf(R.hnerv_utils.psnr_fn_single(img.detach(), frame))
            for pn, p in m.named_parameters():
                out[f"c1/gnorm/{pn}"] = np.float64(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "full_models.npz"), **out)


def gen_full_c1_grads(R):
    """C1 at full size, gradient ELEMENTS (norms alone would pass a sign / permutation error that keeps the norm):
    the run of gen_full_models for c1 once more (same seeds, same frame, same t) and, of EVERY parameter gradient, 512 elements at fixed
    sampled positions (the whole tensor when it has 512 elements or fewer) with mean / std -- stem.2.weight (1.1 M elements), the TAT
    convs, the up-convs, the head and every modulation MLP alike."""
    out = {}
    torch.manual_seed(1)
    m = _model(R, configs.c1())
    out["sd_sha256"] = np.array(sd_hash(m.state_dict()))
    g = torch.Generator().manual_seed(5)
    frame = torch.rand(1, 3, 720, 1280, generator=g)
    norm_idx = torch.tensor([37 / 132], dtype=torch.float64)
    img, lst, _ = m(norm_idx, norm_idx=norm_idx)
    loss = R.hnerv_utils.loss_fn(img, frame, "L1_freq")
    loss.backward()
    out["loss_L1_freq"] = np.float64(loss.item())
    for pn, p in m.named_parameters():
        summary(p.grad, f"grad/{pn}", out, 512)
        out[f"gnorm/{pn}"] = np.float64(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "full_c1_grads.npz"), **out)


def gen_full_1080(R):
    """C3 (HNeRV-boost 3M, model_hnerv.py:224-251, encoder included) and C4 (E-NeRV-boost 3M, model_enerv.py:279-317) at
    1080x1920, torch.manual_seed(1) construction: image / per-stage output summaries at one frame, loss L1_freq, PSNR and
    every parameter's gradient norm.  C3's ConvNeXt encoder initialises through trunc_normal_ (erfinv),
    whose last bit depends on the host CPU's vector ISA, so the encoder's seeded values are stored (0.9 MB) and the test loads
    them over its own seeded construction; the decoder's seeded init is bit-identical (SHA-256 recorded for it alone)."""
    for name, args, t in (("c3", configs.c3(), 37 / 600), ("c4", configs.c4(), 37 / 600)):
        out = {}
        torch.manual_seed(1)
        m = _model(R, args)
        sd = m.state_dict()
        if name == "c3":
            for k, v in sd.items():
                if k.startswith("encoder."):
                    out[f"enc_sd/{k}"] = npf(v)
            out["dec_sha256"] = np.array(sd_hash({k: v for k, v in sd.items() if not k.startswith("encoder.")}))
        else:
            out["sd_sha256"] = np.array(sd_hash(sd))
        g = torch.Generator().manual_seed(5)
        frame = torch.rand(1, 3, 1080, 1920, generator=g)
        norm_idx = torch.tensor([t], dtype=torch.float64)
        inp = frame if args.model == "HNeRV_Boost" else norm_idx
        img, lst, _ = m(inp, norm_idx=norm_idx)
        assert img.shape[-2:] == (1080, 1920), img.shape
        loss = R.hnerv_utils.loss_fn(img, frame, "L1_freq")
        loss.backward()
        out["frame_seed"] = np.int64(5)
        out["norm_idx"] = npf(norm_idx)
        summary(img, "img", out, 2048)
        for i, tt in enumerate(lst):
            summary(tt, f"list{i}", out, 256)
        out["loss_L1_freq"] = np.float64(loss.item())
        out["psnr"] = npf(R.hnerv_utils.psnr_fn_single(img.detach(), frame))
        for pn, p in m.named_parameters():
            out[f"gnorm/{pn}"] = np.float64(p.grad.double().norm().item()) if p.grad is not None else np.float64(-1)
            if p.grad is not None and p.grad.numel() <= 1024:
                out[f"grad/{pn}"] = npf(p.grad)
        np.savez_compressed(os.path.join(OUT, f"full_{name}.npz"), **out)
        del m, img, lst, loss


# ---------------------------------------------------------------------------------------------------------------------
def gen_loss(R):
    """loss_fn variants and psnr_fn_single on seeded inputs (hnerv_utils.py:335-403).  Fusion10_freq goes through the
    msssim_ref stand-in -> that entry is self-consistency only (PARITY UNPINNED)."""
    out = {}
    for tag, shape, seed in (("small", (2, 3, 176, 208), 3), ("odd", (1, 3, 180, 270), 4), ("720p", (1, 3, 720, 1280), 6)):
        g = torch.Generator().manual_seed(seed)
        tgt = torch.rand(shape, generator=g)
        pred = (tgt + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1).requires_grad_(True)
        out[f"{tag}/shape"] = np.array(shape)
        out[f"{tag}/seed"] = np.int64(seed)
        if tag != "720p":
            out[f"{tag}/pred"] = npf(pred).astype(np.float32)
            out[f"{tag}/target"] = npf(tgt)
        for lt in ("L1", "L2", "L1_freq", "Fusion10_freq"):
            pred.grad = None
            l = R.hnerv_utils.loss_fn(pred, tgt, lt)
            l.backward()
            out[f"{tag}/{lt}/loss"] = np.float64(l.item())
            summary(pred.grad, f"{tag}/{lt}/grad", out, 512)
        out[f"{tag}/psnr"] = npf(R.hnerv_utils.psnr_fn_single(pred.detach(), tgt))
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


def gen_optim(R):
    """Adan 6-step trajectory on a 3-tensor toy problem with the train script's call form Adan(params, lr=...)
    (train_nerv_all.py:264, optimizer.py:125-362); adjust_lr table (hnerv_utils.py:292-322)."""
    out = {}
    g = torch.Generator().manual_seed(9)
    ps = [torch.randn(7, 5, generator=g), torch.randn(13, generator=g), torch.randn(2, 3, 3, 3, generator=g)]
    params = [p.clone().requires_grad_(True) for p in ps]
    opt = R.optimizer.Adan(params, lr=0.003)
    for i, p in enumerate(ps):
        out[f"p0/{i}"] = npf(p)
    for step in range(6):
        grads = [torch.randn(p.shape, generator=g) * (0.5 + step) for p in params]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        for group in opt.param_groups:
            group["lr"] = 0.003 * (0.1 + 0.15 * step)
        opt.step()
        for i, (p, gr) in enumerate(zip(params, grads)):
            out[f"g{step}/{i}"] = npf(gr)
            out[f"p{step + 1}/{i}"] = npf(p)
    import types
    a = types.SimpleNamespace(lr_type="cosine_0.1_1_0.1", lr=0.003, epochs=300)

    class _O:
        param_groups = [{"lr": 0.0}]
    xs = np.array([0.0, 0.01, 0.05, 0.0999, 0.1, 0.25, 0.5, 0.9, 0.999, 37.25 / 300.0])
    out["lr/x"] = xs
    out["lr/cosine"] = np.array([R.hnerv_utils.adjust_lr(_O, float(x), 0, a) for x in xs])
    a.lr_type = "hybrid_0.2_1_2_0.1_0.05"
    out["lr/hybrid"] = np.array([R.hnerv_utils.adjust_lr(_O, float(x), 0, a) for x in xs])
    np.savez(os.path.join(OUT, "optim.npz"), **out)


def gen_host(R):
    """Host-side facts: fc_dim from the size solver (train_nerv_all.py:193-217 restated inline there -- no callable, so
    the values recorded are those the reference's expression yields), data_split, quant_tensor, frame order of the
    shuffled loader after model construction (train_nerv_all.py:155, :188-191, :222, :328)."""
    out = {}
    hu = R.hnerv_utils
    tr, va = hu.data_split(list(range(20)), [6, 8, 10], False, 0)
    out["split/train"], out["split/val"] = np.array(tr), np.array(va)
    g = torch.Generator().manual_seed(2)
    for i, t in enumerate((torch.randn(12, 12, 3, 3, generator=g), torch.randn(48, generator=g), torch.randn(256, 160, 1, 1, generator=g))):
        q, new_t = hu.quant_tensor(t, 8)
        out[f"quant/{i}/t"] = npf(t)
        out[f"quant/{i}/q"] = npf(q["quant"])
        out[f"quant/{i}/new"] = npf(new_t)
    # frame order: same call sequence as train(): seeds -> loaders -> model -> iterate
    import random
    from torch.utils.data import DataLoader, Dataset, Subset

    class _DS(Dataset):
        def __len__(self): return 132
        def __getitem__(self, i): return {"idx": i}
    orders = []
    torch.manual_seed(1); np.random.seed(1); random.seed(1)
    full = _DS()
    _ = DataLoader(full, batch_size=1, shuffle=False, num_workers=0)
    tr_idx, _v = hu.data_split(list(range(132)), [1, 1, 1], False, 0)
    dl = DataLoader(Subset(full, tr_idx), batch_size=1, shuffle=True, num_workers=0, drop_last=True)
    _m = R.model_nerv.NeRV_Boost(1, args=configs.c1())
    for _e in range(2):
        orders.append([int(s["idx"][0]) for s in dl])
    out["frame_order/c1"] = np.array(orders)
    np.savez_compressed(os.path.join(OUT, "host.npz"), **out)


def gen_cem(R):
    """CEM pieces of the compression train step: lib/transform_ops.py Scale_T / ScaleBeta_T, lib/entropy_model.py
    DiffEntropyModel (training draw with a fixed torch seed, and the deterministic evaluation form via get_bits)."""
    T, E = R.lib_transform_ops, R.lib_entropy_model
    out = {}
    g = torch.Generator().manual_seed(77)
    for name, shape in (("w", (24, 12, 3, 3)), ("b", (24,))):
        x = (torch.randn(*shape, generator=g) * 0.05).requires_grad_(True)
        q = T.Scale_T(8, signed=True, per_channel=False)
        q.init_form(x)
        q.init_data(x.detach())
        code, quant, deq = q(x)
        cot = torch.randn(*shape, generator=g)
        gx, gs = torch.autograd.grad((deq * cot).sum(), [x, q.scale], retain_graph=True)
        out.update({f"scale/{name}/x": npf(x), f"scale/{name}/cot": npf(cot), f"scale/{name}/scale": npf(q.scale), f"scale/{name}/code": npf(code),
                    f"scale/{name}/quant": npf(quant), f"scale/{name}/dequant": npf(deq), f"scale/{name}/gx": npf(gx), f"scale/{name}/gscale": npf(gs)})
        em = E.DiffEntropyModel(distribution="gaussian")
        torch.manual_seed(5)
        noise = torch.empty_like(code).uniform_(-0.5, 0.5)
        torch.manual_seed(5)
        r = em.cal_bitrate(code, quant, True)
        gx2, gs2 = torch.autograd.grad(r["bitrate"], [x, q.scale], retain_graph=True)
        ev = em.get_bits(quant, torch.mean(code), torch.std(code)).sum()
        out.update({f"rate/{name}/noise": npf(noise), f"rate/{name}/bitrate": npf(r["bitrate"]), f"rate/{name}/mean": npf(r["mean"]),
                    f"rate/{name}/std": npf(r["std"]), f"rate/{name}/gx": npf(gx2), f"rate/{name}/gscale": npf(gs2), f"rate/{name}/eval_bits": npf(ev)})
    e = torch.rand(2, 16, 9, 16, generator=g).requires_grad_(True)
    qe = T.ScaleBeta_T(8, signed=False, per_channel=False)
    qe.init_form(e)
    qe.init_data(e.detach())
    code, quant, deq = qe(e)
    cot = torch.randn(e.shape, generator=g)
    ge, gs, gb = torch.autograd.grad((deq * cot).sum(), [e, qe.scale, qe.beta])
    out.update({"scalebeta/x": npf(e), "scalebeta/cot": npf(cot), "scalebeta/scale": npf(qe.scale), "scalebeta/beta": npf(qe.beta), "scalebeta/code": npf(code),
                "scalebeta/quant": npf(quant), "scalebeta/dequant": npf(deq), "scalebeta/gx": npf(ge), "scalebeta/gscale": npf(gs), "scalebeta/gbeta": npf(gb)})
    np.savez_compressed(os.path.join(OUT, "cem.npz"), **out)


def gen_cem_model(R):
    """The per-step hooks of the compression train loop (train_nerv_compression.py:354-367) on the tiny HNeRV_Boost built with
    --quant: init_data, cal_params(entropy_model), forward(entropy_model=...), get_bitrate_sum; loss = L1 + 1e-6 * bits."""
    args = configs.tiny_hnerv_quant()
    torch.manual_seed(1)
    model = R.model_hnerv.HNeRV_Boost(args)
    model.init_data()
    em = R.lib_entropy_model.DiffEntropyModel(distribution="gaussian")
    frame = torch.rand(1, 3, 180, 320, generator=torch.Generator().manual_seed(5))
    norm_idx = torch.tensor([3 / 7], dtype=torch.float64)
    model.train()
    torch.manual_seed(9)                                     # the uniform draws of cal_bitrate come from the default generator
    model.cal_params(em)
    img, _, _ = model(frame, entropy_model=em, norm_idx=norm_idx)
    bits_w = model.get_bitrate_sum(name="bitrate")
    bits_e = model.bitrate_e_dict["bitrate"]
    loss = (img - frame).abs().mean() + 1e-6 * (bits_w + bits_e)
    loss.backward()
    out = {"frame_seed": np.int64(5), "bits_w": npf(bits_w), "bits_e": npf(bits_e), "loss": npf(loss)}
    out.update(sd_np(model, "sd/"))                          # after init_data and the first forward (embed quantiser initialised)
    summary(img, "img", out)
    for k, p in model.named_parameters():
        out[f"gnorm/{k}"] = np.float64(p.grad.double().norm().item() if p.grad is not None else -1.0)
    for k in ("decoder.1.conv.upconv.0.weight_quantizer.scale", "embed_quantizer.scale", "embed_quantizer.beta", "head_layer.weight"):
        p = dict(model.named_parameters())[k]
        out[f"grad/{k}"] = npf(p.grad)
    np.savez_compressed(os.path.join(OUT, "cem_model.npz"), **out)


def gen_full_c5(R):
    """BASELINE configs[4] at full size: the hooks of ONE compression train step (train_nerv_compression.py:354-367) on the C3 model
    built with --quant at 1080x1920 -- init_data, cal_params(entropy_model) in train mode (uniform rate noise from the seeded CPU
    generator, in module order), forward(entropy_model=...), get_bitrate_sum -- with loss = L1 + 1e-6 * bits.  Stored: the rate terms,
    EVERY module's (bits, mean, std) for weight and bias, the quantiser scales after init_data, image summary, every gradient norm.
    The seeded weights equal full_c3.npz's (same constructor order; the quantisers draw no random numbers): recorded as a flag."""
    args = configs.c5()
    torch.manual_seed(1)
    m = R.model_hnerv.HNeRV_Boost(args)
    out = {}
    c3 = np.load(os.path.join(OUT, "full_c3.npz"))
    sd = m.state_dict()
    out["enc_equals_full_c3"] = np.int64(all(np.array_equal(npf(sd[k[len("enc_sd/"):]]), c3[k]) for k in c3.files if k.startswith("enc_sd/")))
    out["dec_sha256"] = np.array(sd_hash({k: v for k, v in sd.items() if not k.startswith("encoder.") and "quantizer" not in k}))
    m.init_data()
    em = R.lib_entropy_model.DiffEntropyModel(distribution="gaussian")
    frame = torch.rand(1, 3, 1080, 1920, generator=torch.Generator().manual_seed(5))
    norm_idx = torch.tensor([37 / 600], dtype=torch.float64)
    m.train()
    torch.manual_seed(9)
    m.cal_params(em)
    img, _, _ = m(frame, entropy_model=em, norm_idx=norm_idx)
    bits_w = m.get_bitrate_sum(name="bitrate")
    bits_e = m.bitrate_e_dict["bitrate"]
    loss = (img - frame).abs().mean() + 1e-6 * (bits_w + bits_e)
    loss.backward()
    out.update({"frame_seed": np.int64(5), "norm_idx": npf(norm_idx), "bits_w": npf(bits_w), "bits_e": npf(bits_e), "loss": npf(loss)})
    for k, v in m.state_dict().items():
        if "quantizer" in k:
            out[f"q/{k}"] = npf(v)
    for name, mod in m.named_modules():
        if hasattr(mod, "bitrate_w_dict") and "bitrate" in getattr(mod, "bitrate_w_dict", {}):
            d = mod.bitrate_w_dict
            out[f"bw/{name}"] = np.array([d["bitrate"].item(), d["mean"].item(), d["std"].item(), mod.weight.numel()], dtype=np.float64)
            if "bitrate" in mod.bitrate_b_dict:
                d = mod.bitrate_b_dict
                out[f"bb/{name}"] = np.array([d["bitrate"].item(), d["mean"].item(), d["std"].item(), mod.bias.numel()], dtype=np.float64)
    summary(img, "img", out, 2048)
    for pn, p in m.named_parameters():
        out[f"gnorm/{pn}"] = np.float64(p.grad.double().norm().item()) if p.grad is not None else np.float64(-1)
    np.savez_compressed(os.path.join(OUT, "full_c5.npz"), **out)


def gen_inpaint(R):
    """Inpainting masks (row a12): TransformInput (hnerv_utils.py:59-84) in its 'center' and 'fixed' modes and the masked train step
    of train_nerv_all.py:334-346 -- input = (img * mask).clamp(0, 1), loss_fn(img_out * mask, gt * mask), PSNR against the unmasked gt --
    on the tiny NeRV_Boost (index input) and HNeRV_Boost (the masked frame IS the network input), state_dicts of tiny_*.npz."""
    import copy
    out = {}
    for mname, args0 in (("nerv", configs.tiny_nerv()), ("hnerv", configs.tiny_hnerv())):
        for mode in ("inpanting_center", "inpanting_fixed_50"):
            args = copy.copy(args0)
            args.inpanting = mode
            torch.manual_seed(1)
            m = _model(R, args)
            tf = R.hnerv_utils.TransformInput(args)
            g = torch.Generator().manual_seed(5)
            frame = torch.rand(2, 3, 180, 320, generator=g) * 1.2 - 0.1          # (values outside [0, 1]: the clamp of the masked input acts)
            norm_idx = torch.tensor([3 / 7, 6 / 7], dtype=torch.float64)
            inp, gt, mask = tf(frame, torch.tensor([2, 5]))
            img, _, _ = m(inp if args.model == "HNeRV_Boost" else norm_idx, norm_idx=norm_idx)
            loss = R.hnerv_utils.loss_fn(img * mask, gt * mask, "L1_freq")
            loss.backward()
            k = f"{mname}/{mode}"
            out[f"{k}/mask"] = np.packbits(npf(mask).astype(np.uint8))
            out[f"{k}/mask_shape"] = np.array(mask.shape, dtype=np.int64)
            out[f"{k}/mask_zeros"] = np.int64((mask == 0).sum().item())
            summary(inp, f"{k}/inp", out, 512)
            summary(img, f"{k}/img", out, 512)
            out[f"{k}/gt_equals_frame"] = np.bool_(torch.equal(gt, frame))
            out[f"{k}/loss_L1_freq"] = np.float64(loss.item())
            out[f"{k}/psnr"] = npf(R.hnerv_utils.psnr_fn_single(img.detach(), gt))
            for pn, p in m.named_parameters():
                out[f"{k}/gnorm/{pn}"] = np.float64(p.grad.double().norm().item()) if p.grad is not None else np.float64(-1)
    np.savez_compressed(os.path.join(OUT, "inpaint.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    R = ref_harness.load_reference()
    which = sys.argv[1:] or ["pe", "blocks", "cnx", "tiny", "full", "full1080", "loss", "optim", "host", "cem", "cem_model", "full_c5", "inpaint", "full_c1_grads"]
    fns = dict(pe=gen_pe, blocks=gen_blocks, cnx=gen_convnext_blocks, tiny=gen_tiny_models, full=gen_full_models, full1080=gen_full_1080, loss=gen_loss, optim=gen_optim, host=gen_host, cem=gen_cem, cem_model=gen_cem_model, full_c5=gen_full_c5, inpaint=gen_inpaint, full_c1_grads=gen_full_c1_grads)
    for w in which:
        print("generating", w, flush=True)
        fns[w](R)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
