"""CPU suite (``-m "not gpu"``): the C-ABI library loads and exports every symbol include/bnerv.h declares, argument
validation works without a GPU, and the host logic (seeded init / state_dict parity with the reference, size solver, LR
schedule, frame order, data split, quantisation helpers, sampler sharding, synthetic clip, CLI, loud failure on CPU tensors)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import configs


def test_library_exports_every_header_symbol():
    from boosting_nerv_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "bnerv.h")).read()
    declared = set(re.findall(r"\b(bnerv_[a-z0-9_]+)\s*\(", header))
    declared -= {n for n in declared if n.endswith("_desc") or n.endswith("_chunk") or n.endswith("_hyper")}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/bnerv.h but not exported by libbnerv_hip.so"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes binding in boosting_nerv_amd/_lib.py"
    assert set(_lib.SYMBOLS) <= declared
    assert lib.bnerv_abi_version() == _lib.ABI_VERSION == 9 and lib.bnerv_build_arch() == b"gfx950"


def test_abi_argument_validation_without_gpu():
    """Entry points validate before touching the device: bad descriptors return BNERV_E_ARG with a message."""
    import ctypes as C
    from boosting_nerv_amd import _lib as L
    lib = L.load()
    d = L.ConvDesc()
    d.k = 5
    assert lib.bnerv_conv_igemm(None, C.byref(d)) == -1
    assert b"k must be 1 or 3" in lib.bnerv_last_error()
    assert lib.bnerv_conv_wgrad_ws_bytes(1, 12, 12, 720, 1280, 3) > 0
    assert lib.bnerv_conv_wgrad_ws_bytes(1, 12, 12, 720, 1280, 7) == 0
    assert lib.bnerv_loss_ws_bytes(1, 3, 720, 1280, 1, 1) > 3 * 720 * 1280 * 8
    assert lib.bnerv_conv_tiles(720, 1280) == 90 * 40
    assert lib.bnerv_reduce_slabs(None, None, 0, 0, None) == -1
    # struct layouts the kernels rely on
    assert C.sizeof(L.DenseFwdDesc) == 56 and C.sizeof(L.DenseBwdDesc) == 88
    assert C.sizeof(L.DenseBwdDesc) * L.MAX_DENSE_GROUPS + 8 <= 4096      # kernel-argument segment limit


def test_product_path_fails_loudly_on_cpu_tensors():
    from boosting_nerv_amd import _lib, ops
    from boosting_nerv_amd.model_nerv import NeRV_Boost
    torch.manual_seed(1)
    m = NeRV_Boost(1, args=configs.tiny_nerv())
    t = torch.tensor([0.5], dtype=torch.float64)
    with pytest.raises(_lib.BnervError, match="no CPU fallback"):
        m(t, norm_idx=t)
    with pytest.raises(_lib.BnervError):
        ops.loss_with_stats(torch.rand(1, 3, 200, 200), torch.rand(1, 3, 200, 200), "L1")
    with pytest.raises(NotImplementedError):
        ops.loss_with_stats(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8), "Fusion6")


@pytest.mark.parametrize("name", ["c1", "c3", "c4"])
def test_seeded_init_and_state_dict_keys_match_reference(name):
    """Same constructor calls in the same order => the reference's parameters, bit for bit (checked against the SHA-256 of
    the reference's own state_dict; the ConvNeXt encoder is exempt from the bit test across CPU ISAs)."""
    import hashlib
    from boosting_nerv_amd.model_enerv import ENeRV_Boost
    from boosting_nerv_amd.model_hnerv import HNeRV_Boost
    from boosting_nerv_amd.model_nerv import NeRV_Boost
    npz = load_golden("full_models.npz")
    args = getattr(configs, name)()
    torch.manual_seed(1)
    m = {"c1": lambda: NeRV_Boost(1, args=args), "c3": lambda: HNeRV_Boost(args), "c4": lambda: ENeRV_Boost(3, args=args)}[name]()
    sd = m.state_dict()
    assert list(sd.keys()) == list(npz[f"{name}/keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(npz[f"{name}/shapes"])
    assert sum(p.numel() for p in m.parameters()) == int(npz[f"{name}/n_params"])
    h = hashlib.sha256()
    for k, v in sd.items():
        if name == "c3" and k.startswith("encoder."):
            continue
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    if name != "c3":
        assert h.hexdigest() == str(npz[f"{name}/sd_sha256"])
    torch.testing.assert_close(next(iter(sd.values())).flatten()[:8], torch.from_numpy(npz[f"{name}/first_vals"]), rtol=0, atol=1e-6)


def test_size_solver_reproduces_reference_fc_dim():
    import bench
    for cfg, fc in (("c1", 30), ("c3", 95), ("c4", 59)):      # values the reference's solver yields
        from boosting_nerv_amd import train_nerv_all as T
        r = bench.RECIPES[cfg]
        args = T.build_parser().parse_args(r["flags"].split())
        got, _ = T.solve_fc_dim(args, r["h"] * r["w"], r["n"])
        assert got == fc, (cfg, got)


def test_cli_accepts_the_reference_recipe_and_defaults():
    from boosting_nerv_amd import train_nerv_all as T
    p = T.build_parser()
    a = p.parse_args([])
    assert (a.crop_list, a.ks, a.reduce, a.lower_width, a.dec_strds, a.loss, a.optim_type, a.lr_type, a.batchSize, a.epochs, a.manualSeed) == \
        ("640_1280", "0_3_3", 1.2, 32, [5, 3, 2, 2, 2], "Fusion6", "adan", "cosine_0.1_1_0.1", 1, 5, 1)
    line = ("--outf regression/NeRV_Boost/epoch_300 --model NeRV_Boost --sft_block res_sft --ch_t 32 --data_path ./dataset/bunny --vid bunny "
            "--optim_type Adan --conv_type convnext pshuffel_3x3 --act sin --norm none --crop_list 720_1280 --resize_list -1 --loss Fusion10_freq "
            "--embed pe_1.25_80 --fc_hw 9_16 --dec_strds 5 2 2 2 2 --ks 0_3_3 --reduce 2 --dec_blks 1 1 2 2 2 --modelsize 0.8 -e 300 --eval_freq 30 "
            "--lower_width 12 -b 1 --lr 0.003")
    a = p.parse_args(line.split())
    assert a.model == "NeRV_Boost" and a.conv_type == ["convnext", "pshuffel_3x3"] and a.dec_blks == [1, 1, 2, 2, 2]


def test_lr_schedule_and_host_helpers_match_reference_goldens():
    from types import SimpleNamespace
    from boosting_nerv_amd import hnerv_utils as hu
    npz = load_golden("optim.npz")

    class _O:
        param_groups = [{"lr": 0.0}]
    a = SimpleNamespace(lr_type="cosine_0.1_1_0.1", lr=0.003, epochs=300)
    for x, c in zip(npz["lr/x"], npz["lr/cosine"]):
        assert hu.adjust_lr(_O, float(x), 0, a) == c
    a.lr_type = "hybrid_0.2_1_2_0.1_0.05"
    for x, c in zip(npz["lr/x"], npz["lr/hybrid"]):
        assert hu.adjust_lr(_O, float(x), 0, a) == c
    host = load_golden("host.npz")
    tr, va = hu.data_split(list(range(20)), [6, 8, 10], False, 0)
    assert tr == host["split/train"].tolist() and va == host["split/val"].tolist()
    for i in range(3):
        t = torch.from_numpy(host[f"quant/{i}/t"])
        q, new_t = hu.quant_tensor(t, 8)
        assert np.array_equal(q["quant"].numpy(), host[f"quant/{i}/q"])
        torch.testing.assert_close(new_t, torch.from_numpy(host[f"quant/{i}/new"]))


def test_frame_order_matches_reference():
    """Seeds -> loaders -> model -> iterate, as train(): the shuffled frame order of the reference for C1 (pins both the
    init RNG consumption and the loader construction order)."""
    import random
    from torch.utils.data import DataLoader, Subset
    from boosting_nerv_amd import hnerv_utils as hu
    from boosting_nerv_amd.model_nerv import NeRV_Boost
    from boosting_nerv_amd.train_nerv_all import _IndexOnly
    host = load_golden("host.npz")
    torch.manual_seed(1); np.random.seed(1); random.seed(1)
    ds = _IndexOnly(132)
    _ = DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    tr_idx, _v = hu.data_split(list(range(132)), [1, 1, 1], False, 0)
    dl = DataLoader(Subset(ds, tr_idx), batch_size=1, shuffle=True, num_workers=0, drop_last=True)
    NeRV_Boost(1, args=configs.c1())
    orders = [[int(s["idx"][0]) for s in dl] for _ in range(2)]
    assert np.array_equal(np.array(orders), host["frame_order/c1"])
    assert orders[0][:6] == [26, 27, 112, 2, 58, 29]


def test_shard_indices_equal_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from boosting_nerv_amd.dp import shard_indices
    for n, world in ((132, 8), (132, 2), (600, 8), (7, 4)):
        for rank in range(world):
            ref = list(DistributedSampler(range(n), num_replicas=world, rank=rank))
            assert shard_indices(n, rank, world, seed=0) == ref


def test_synthetic_clip_is_deterministic_8bit():
    from boosting_nerv_amd.synth import SyntheticVideo, parse_spec
    assert parse_spec("synthetic:bunny") == (132, 720, 1280) and parse_spec("synthetic:uvg") == (600, 1080, 1920)
    a, b = SyntheticVideo(4, 36, 64), SyntheticVideo(4, 36, 64)
    f = a.frame(2)
    assert torch.equal(f, b.frame(2)) and not torch.equal(f, a.frame(3))
    assert f.shape == (3, 36, 64) and 0 <= f.min() and f.max() <= 1
    assert torch.equal(torch.round(f * 255) / 255, f)


def test_huffman_total_bits():
    from boosting_nerv_amd.train_nerv_all import _huffman_total_bits
    # symbols {a:5,b:2,c:1,d:1} + EOF:1: every optimal code costs 19 bits over all five leaves; the data-only total is 16 or 17
    # depending on which tie the EOF leaf wins (a couple of bits out of millions in the bpp report)
    assert _huffman_total_bits([5, 2, 1, 1]) in (16, 17)
    assert _huffman_total_bits([10]) == 10


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boosting_nerv_amd.dp import GradBucket, shard_indices
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    GradBucket(params).allreduce_mean()
    expect = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect * (i + 1))) for i, p in enumerate(params))
    shards = shard_indices(10, rank, world, seed=0)
    q.put((rank, ok, shards))
    dist.destroy_process_group()


def test_dp_gradient_mean_two_ranks_gloo():
    """world_size-2 CPU test of the N>1 path: flat-bucket all-reduce averages gradients like DDP, frame shards partition
    the (padded) index set."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    allidx = sorted(sum((s for _, _, s in res), []))
    assert allidx == list(range(10))


def _gloo_world8_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from boosting_nerv_amd.dp import GradBucket, shard_indices
        gen = torch.Generator().manual_seed(100)
        shapes = [(256, 160), (256,), (48, 12, 3, 3), (3, 12, 1, 1), (3,), (1000, 33)]
        params = [torch.nn.Parameter(torch.randn(*sh, generator=gen)) for sh in shapes]
        own = [torch.randn(*sh, generator=torch.Generator().manual_seed(7 * rank + i)) for i, sh in enumerate(shapes)]
        for p, g in zip(params, own):
            p.grad = g.clone()
        GradBucket(params).allreduce_mean()
        ok = True
        for p, g in zip(params, own):
            parts = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(parts, g)
            ok &= torch.allclose(p.grad, sum(parts) / world, rtol=1e-6, atol=1e-7)
        # the reference's frame shards at -b 8 -d on 8 GPUs: 132 frames -> 17 per rank, 4 padding duplicates (train_nerv_all.py:168, :189-191)
        sh = shard_indices(132, rank, world, seed=0)
        from torch.utils.data.distributed import DistributedSampler
        ds = DistributedSampler(list(range(132)), num_replicas=world, rank=rank, shuffle=True, seed=0)
        q.put((rank, ok, sh, list(iter(ds))))
    finally:
        dist.destroy_process_group()


def test_dp_world8_bucket_mean_and_padded_shards_gloo():
    """world_size 8 on CPU (gloo), the shape of the driver's 8-GPU run: the flat-bucket all-reduce averages 8 different gradients like
    DDP; `shard_indices(132, r, 8)` equals torch's DistributedSampler on every rank -- 17 frames per rank (132 -> 136 with four padding
    duplicates taken from the head of the permutation), the shards cover every frame, and exactly four frames appear twice."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_world8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _, _ in res)
    for _, _, mine, ref in res:
        assert mine == ref and len(mine) == 17
    allidx = sum((s for _, _, s, _ in res), [])
    assert len(allidx) == 136 and sorted(set(allidx)) == list(range(132))
    from collections import Counter
    assert sorted(Counter(allidx).values())[-5:] == [1, 2, 2, 2, 2]


def _gloo_two_bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    import torch.nn as nn
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boosting_nerv_amd.dp import GradBucket
    try:
        torch.manual_seed(0)
        stem = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 12))          # "late": its gradients come last in the backward
        dec = nn.Sequential(nn.Linear(12, 9), nn.Tanh(), nn.Linear(9, 4))             # "early": the decoder layers
        params = list(stem.parameters()) + list(dec.parameters())
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 + rank))     # every rank its own shard

        def backward(bucket):
            for p in params:
                p.grad = None
            fired = []
            h = stem(x)
            if bucket is not None and bucket.two:
                def hook(g):
                    fired.append(all(p.grad is not None for p in dec.parameters()) and all(p.grad is None for p in stem.parameters()))
                    bucket.exchange_early()
                h.register_hook(hook)
            dec(h).square().sum().backward()
            return fired

        backward(None)
        own = [p.grad.clone() for p in params]
        one = GradBucket(params)
        one.allreduce_mean()
        mean1 = [p.grad.clone() for p in params]
        two = GradBucket(params, late_params=list(stem.parameters()))
        fired = backward(two)
        inflight = two._early_inflight
        two.finish()
        mean2 = [p.grad.clone() for p in params]
        # reference mean from an all-gather of the ranks' own gradients
        ok_ref = True
        for g, m in zip(own, mean1):
            parts = [torch.empty_like(g) for _ in range(world)]
            dist.all_gather(parts, g)
            ok_ref &= torch.allclose(m, sum(parts) / world, rtol=1e-6, atol=1e-7)
        q.put((rank, two.two, two.split == sum(p.numel() for p in dec.parameters()), fired == [True], inflight, ok_ref,
               all(torch.equal(a, b) for a, b in zip(mean1, mean2)), not two._early_inflight))
    finally:
        dist.destroy_process_group()


def test_dp_two_bucket_exchange_two_ranks_gloo():
    """The two-segment bucket with world_size 2 on CPU: the hook at the decoder / stem boundary fires when exactly
    the decoder layers' gradients exist, their all-reduce is in flight while the stem's backward runs, finish() exchanges the rest and
    joins -- and every averaged gradient is BIT-equal to the one-bucket exchange (same sum of the same two numbers per element) and
    equal to the mean of the ranks' own gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_two_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        assert all(r[1:]), r


def test_dp_mean_of_per_rank_grads_equals_batch_grad_oracle():
    """DP equivalence: grad of the batch-mean loss at b=2 == mean of the two single-frame grads."""
    from oracle import cpu_ref
    npz = load_golden("tiny_nerv.npz")
    sd0 = {k[3:]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith("sd/")}
    frame = torch.rand(2, 3, 180, 320, generator=torch.Generator().manual_seed(5))
    norm_idx = torch.from_numpy(npz["norm_idx"])

    def grads(fr, ni):
        sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        cpu_ref.loss_fn(cpu_ref.nerv_boost_forward(sd, ni), fr, "L1_freq").backward()
        return {k: v.grad for k, v in sd.items()}
    gb = grads(frame, norm_idx)
    g0, g1 = grads(frame[:1], norm_idx[:1]), grads(frame[1:], norm_idx[1:])
    for k in gb:
        torch.testing.assert_close(gb[k], 0.5 * (g0[k] + g1[k]), rtol=1e-4, atol=1e-7)


def test_patchify_conv_equals_conv2d():
    """kernel == stride convs of the ConvNeXt encoder as reshape + GEMM: same values and gradients as nn.Conv2d (incl. a
    spatial size that is not a multiple of the stride); other conv geometries fall through to the module."""
    import torch.nn as nn
    from boosting_nerv_amd.model_blocks import patchify_conv
    torch.manual_seed(3)
    for (C, Co, s, H, W) in ((3, 8, 5, 20, 37), (8, 8, 3, 9, 12), (8, 4, 2, 7, 8)):
        conv = nn.Conv2d(C, Co, s, stride=s)
        x = torch.randn(2, C, H, W, requires_grad=True)
        a, b = conv(x), patchify_conv(x, conv)
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
        cot = torch.randn_like(a)
        ga = torch.autograd.grad(a, [x, conv.weight, conv.bias], cot)
        gb = torch.autograd.grad(b, [x, conv.weight, conv.bias], cot)
        for u, v in zip(ga, gb):
            torch.testing.assert_close(v, u, rtol=1e-4, atol=1e-5)
    other = nn.Conv2d(4, 4, 3, stride=1, padding=1)
    x = torch.randn(1, 4, 6, 6)
    assert torch.equal(patchify_conv(x, other), other(x))


def test_cem_model_surface_matches_reference():
    """--quant build of HNeRV_Boost: state_dict keys (quantiser parameters included) equal the reference's (golden), the
    compression CLI parses the recipe's flags, and the un-built quantisers fail loudly."""
    from boosting_nerv_amd import train_nerv_compression as C
    from boosting_nerv_amd.lib.transform_ops import quant_map
    from boosting_nerv_amd.model_hnerv import HNeRV_Boost
    npz = load_golden("cem_model.npz")
    keys = [k[len("sd/"):] for k in npz.files if k.startswith("sd/")]
    torch.manual_seed(1)
    model = HNeRV_Boost(configs.tiny_hnerv_quant())
    assert list(model.state_dict().keys()) == keys
    assert sum("quantizer" in k for k in keys) > 50
    a = C.build_parser().parse_args("--quant --quant_model_bit 8 --quant_bias_bit 8 --quant_embed_bit 8 --quantizer_w scale --quantizer_b scale "
                                    "--quantizer_e scalebeta --lambda_rate 0.05 --target_bit 4 --embed_entropy --lr_type cosine_0_1_0.1 --not_resume".split())
    assert a.quant and a.quantizer_e == "scalebeta" and a.lambda_rate == 0.05 and a.target_bit == 4 and a.embed_entropy
    with pytest.raises(NotImplementedError):
        quant_map["lsq"](8, signed=True)


# ------------------------------------------------------------------------------------------------------------ ANS coder (row N2)
def test_ans_coder_roundtrip_and_rate_on_cem_golden_symbols():
    """The range-ANS coder behind `real_bitrate` (csrc/ans.cpp through the C-ABI; host code, runs without a GPU): on the rounded
    symbols the REFERENCE's quantisers produced (tests/golden/cem.npz) and under the Gaussian the reference's rate model fits to
    them, decode(encode(x)) is bit-exact and the message is within 1 % + 64 bits of the ideal code length."""
    from boosting_nerv_amd.lib import entropy_model as em
    npz = load_golden("cem.npz")
    for name in ("w", "b"):
        quant = torch.from_numpy(npz[f"scale/{name}/quant"])
        mean, std = float(npz[f"rate/{name}/mean"]), float(npz[f"rate/{name}/std"])
        sym = quant.int().flatten().numpy()
        lo, hi = int(sym.min()), int(sym.max())
        words = em.ans_encode_gaussian(sym, lo, hi, mean, std)
        back = em.ans_decode_gaussian(words, sym.size, lo, hi, mean, std)
        assert np.array_equal(back, sym), name
        ideal = em.ideal_code_bits(quant, torch.tensor(mean), torch.tensor(std))
        assert ideal <= words.size * 32 <= 1.01 * ideal + 64, (name, words.size * 32, ideal)
        assert em.compress_matrix_flatten_gaussian_global(quant, torch.tensor(mean), torch.tensor(std)) == words.size * 32
    # the embedding quantiser's symbols (unsigned, wide support)
    q = torch.from_numpy(npz["scalebeta/quant"])
    code = torch.from_numpy(npz["scalebeta/code"])
    sym = q.int().flatten().numpy()
    words = em.ans_encode_gaussian(sym, int(sym.min()), int(sym.max()), float(code.mean()), float(code.std()))
    assert np.array_equal(em.ans_decode_gaussian(words, sym.size, int(sym.min()), int(sym.max()), float(code.mean()), float(code.std())), sym)
    ideal = em.ideal_code_bits(q, code.mean(), code.std())
    assert ideal <= words.size * 32 <= 1.01 * ideal + 64


def test_ans_coder_edge_cases():
    """Empty message, one symbol, a constant tensor (min == max -> the reference widens the support by one), a degenerate
    Gaussian (std at the 1e-5 clamp: every symbol off the mean costs the 24-bit leak), symbols outside the support (error),
    a long random message, and the categorical model on a skewed histogram."""
    from boosting_nerv_amd import _lib as L
    from boosting_nerv_amd.lib import entropy_model as em
    assert em.ans_encode_gaussian(np.zeros(0, np.int32), 0, 1, 0.0, 1.0).size == 0
    assert em.ans_decode_gaussian(np.zeros(0, np.uint32), 0, 0, 1, 0.0, 1.0).size == 0
    one = em.ans_encode_gaussian(np.array([3], np.int32), -4, 4, 0.5, 2.0)
    assert 1 <= one.size <= 2 and em.ans_decode_gaussian(one, 1, -4, 4, 0.5, 2.0)[0] == 3
    const = torch.full((1000,), 7.0)
    assert em.compress_matrix_flatten_gaussian_global(const, const.mean(), const.std()) <= 64          # ~0 bits per symbol
    sym = np.array([0, 0, 5, -5, 0], np.int32)
    w = em.ans_encode_gaussian(sym, -5, 5, 0.0, 1e-5)
    assert np.array_equal(em.ans_decode_gaussian(w, 5, -5, 5, 0.0, 1e-5), sym) and 32 < w.size * 32 <= 2 * 24 + 64
    with pytest.raises(L.BnervError):
        em.ans_encode_gaussian(np.array([9], np.int32), -4, 4, 0.0, 1.0)
    with pytest.raises(L.BnervError):
        em.ans_decode_gaussian(np.array([123456789, 42], np.uint32), 50, -4, 4, 0.0, 1.0)     # garbage does not decode silently
    g = np.random.RandomState(3)
    big = np.clip(np.rint(g.normal(2.0, 9.0, size=200000)), -128, 127).astype(np.int32)
    w = em.ans_encode_gaussian(big, -128, 127, float(big.mean()), float(big.std()))
    assert np.array_equal(em.ans_decode_gaussian(w, big.size, -128, 127, float(big.mean()), float(big.std())), big)
    ideal = em.ideal_code_bits(torch.from_numpy(big).float(), torch.tensor(float(big.mean())), torch.tensor(float(big.std())))
    assert ideal <= w.size * 32 <= 1.01 * ideal + 64
    vals = g.choice([0, 1, 2, 3, 200], p=[0.6, 0.2, 0.1, 0.09, 0.01], size=50000)
    words, counts, unique = em.compress_matrix_flatten_categorical(vals)
    inv = np.searchsorted(unique, vals).astype(np.int32)
    assert np.array_equal(em.ans_decode_categorical(words, vals.size, counts / counts.sum()), inv)
    h = -(counts * np.log2(counts / counts.sum())).sum()
    assert h <= words.size * 32 <= 1.01 * h + 64


@pytest.mark.parametrize("mode", ["inpanting_center", "inpanting_fixed_50"])
def test_inpainting_transform_and_masked_step_match_reference_golden(mode):
    """Row a12 on the host: TransformInput (reference hnerv_utils.py:59-84) builds the reference's mask and masked, clamped input bit for
    bit, returns the untouched frame as gt, and the oracle's masked step -- loss_fn(img * mask, gt * mask) of train_nerv_all.py:343 on
    the tiny NeRV_Boost -- reproduces the reference's loss and PSNR (tests/golden/inpaint.npz, oracle/make_goldens.py inpaint)."""
    import copy
    from boosting_nerv_amd.hnerv_utils import TransformInput
    from conftest import check_summary, group
    from oracle import cpu_ref
    npz = load_golden("inpaint.npz")
    args = copy.copy(configs.tiny_nerv())
    args.inpanting = mode
    tf = TransformInput(args)
    assert not tf.identity
    frame = torch.rand(2, 3, 180, 320, generator=torch.Generator().manual_seed(5)) * 1.2 - 0.1
    inp, gt, mask = tf(frame, torch.tensor([2, 5]))
    k = f"nerv/{mode}"
    want = np.unpackbits(npz[f"{k}/mask"])[:180 * 320].reshape(180, 320)
    assert tuple(mask.shape) == tuple(npz[f"{k}/mask_shape"]) and np.array_equal(mask.numpy().astype(np.uint8), want)
    assert int((mask == 0).sum()) == int(npz[f"{k}/mask_zeros"]) and not mask.requires_grad
    assert torch.equal(gt, frame) and bool(npz[f"{k}/gt_equals_frame"])
    check_summary(inp, npz, f"{k}/inp", 0, 0)
    assert inp.min() >= 0 and inp.max() <= 1
    none = copy.copy(args); none.inpanting = "none"
    a, b, m0 = TransformInput(none)(frame, torch.tensor([2, 5]))
    assert a is frame and b is frame and m0 is None                 # identity: the train loop skips the two multiplies by one
    sd = {kk: v for kk, v in group(load_golden("tiny_nerv.npz"), "sd/").items()}
    norm_idx = torch.tensor([3 / 7, 6 / 7], dtype=torch.float64)
    img = cpu_ref.nerv_boost_forward(sd, norm_idx)
    check_summary(img, npz, f"{k}/img", 1e-3, 1e-5)
    loss = cpu_ref.loss_fn(img * mask, gt * mask, "L1_freq")
    gold = float(npz[f"{k}/loss_L1_freq"])
    assert abs(loss.item() - gold) < 1e-4 * abs(gold), (loss.item(), gold)
    torch.testing.assert_close(cpu_ref.psnr_fn_single(img, gt), torch.from_numpy(npz[f"{k}/psnr"]), rtol=1e-5, atol=1e-4)


def test_two_bucket_segments_are_refused_where_the_backward_finishes_gradients_late(monkeypatch):
    """BNERV_DP_BUCKETS=2 must not reach CompressionStep (its decoder gradients appear in the CEM backward, AFTER the
    stem-output hook: the early segment would all-reduce zeros and finish() would scatter them over the real gradients), and
    exchange_early() must refuse a segment with a missing gradient instead of zero-filling it."""
    import torch.nn as nn
    from boosting_nerv_amd import engine
    from boosting_nerv_amd.dp import GradBucket
    from boosting_nerv_amd.model_nerv import NeRV_Boost
    from boosting_nerv_amd.optimizer import Adan
    monkeypatch.setenv("BNERV_DP_BUCKETS", "2")
    torch.manual_seed(1)
    args = configs.tiny_nerv()
    model = NeRV_Boost(1, args=args)
    opt = Adan(model.parameters(), lr=1e-3)
    dev = torch.device("cpu")
    plain = engine.TrainStep(model, opt, "L1", False, (1, 3, 180, 320), dev, use_graph=False, force_bucket=True)
    assert plain.bucket.two and model.dp_hook is not None           # the plain step honours the switch ...
    model.dp_hook = None

    class _Args:
        loss, model, embed, clip_max_norm = "L1", "NeRV_Boost", "pe_1.25_80", 0.0
    comp = engine.CompressionStep(model, opt, None, _Args(), (1, 3, 180, 320), dev, use_graph=False, force_bucket=True)
    assert not comp.bucket.two and model.dp_hook is None            # ... the rate-distortion step does not
    with pytest.raises(ValueError):
        class _Sub(engine.TrainStep):
            pass
        _Sub(model, opt, "L1", False, (1, 3, 180, 320), dev, use_graph=False, force_bucket=True, dp_buckets=2)

    # exchange_early() with an early gradient that does not exist yet: refused, nothing gathered
    stem, dec = nn.Linear(4, 4), nn.Linear(4, 2)
    b = GradBucket(list(stem.parameters()) + list(dec.parameters()), late_params=list(stem.parameters()), force=True)
    assert b.two
    with pytest.raises(RuntimeError, match="no gradient yet"):
        b.exchange_early()
    assert not b._early_inflight and all(p.grad is None for p in dec.parameters())


def test_step_frame_rejects_indices_outside_the_bound_clip():
    """the fetch kernel clamps the index it reads; the host refuses a bad one before anything is launched."""
    from boosting_nerv_amd import engine
    from boosting_nerv_amd.model_nerv import NeRV_Boost
    from boosting_nerv_amd.optimizer import Adan
    torch.manual_seed(1)
    model = NeRV_Boost(1, args=configs.tiny_nerv())
    step = engine.TrainStep(model, Adan(model.parameters(), lr=1e-3), "L1", False, (1, 3, 180, 320), torch.device("cpu"), use_graph=False)
    step.bind_clip(torch.zeros(4, 3, 180, 320), torch.arange(1, 5, dtype=torch.float64) / 4)
    for bad in (-1, 4, 100):
        with pytest.raises(IndexError):
            step.step_frame(bad)


def test_lazy_flush_is_taken_only_when_nothing_else_reads_gradients_in_backward():
    """engine.TrainStep checks the lazy-flush invariant itself (no .grad to accumulate into, no tensor / module hooks)."""
    from boosting_nerv_amd import engine
    from boosting_nerv_amd.model_nerv import NeRV_Boost
    from boosting_nerv_amd.optimizer import Adan
    torch.manual_seed(1)
    model = NeRV_Boost(1, args=configs.tiny_nerv())
    step = engine.TrainStep(model, Adan(model.parameters(), lr=1e-3), "L1", False, (1, 3, 180, 320), torch.device("cpu"), use_graph=False)
    assert step._lazy_flush_valid()
    p = next(model.parameters())
    p.grad = torch.zeros_like(p)                                     # zero_grad(set_to_none=False) style: AccumulateGrad would read + add
    assert not step._lazy_flush_valid()
    p.grad = None
    h = p.register_hook(lambda g: g)
    assert not step._lazy_flush_valid()
    h.remove()
    assert step._lazy_flush_valid()
    h = model.head_layer.register_full_backward_hook(lambda m, gi, go: None)
    assert not step._lazy_flush_valid()
    h.remove()
    assert step._lazy_flush_valid()


def test_adan_captured_launch_needs_the_capture_bracket():
    """a captured Adan launch reads a table finish_capture() uploads; every bracket owns its tables."""
    from boosting_nerv_amd.optimizer import Adan
    w = torch.nn.Parameter(torch.zeros(8))
    opt = Adan([w], lr=1e-3)
    assert opt._cap_open is None
    opt.begin_capture()
    t1 = opt._cap_open[0][1]
    tabs = opt.finish_capture()
    assert opt._cap_open is None and len(tabs) == 1 and tabs[0] is t1
    opt.begin_capture()
    assert opt._cap_open[0][1] is not t1                           # a second capture never shares the first one's table
    opt.finish_capture()


def test_isolated_child_crash_is_an_ordinary_failure():
    """conftest.run_isolated (what @pytest.mark.isolated runs a test through): a child that dies with SIGABRT must come back as a
    normal pytest failure whose message carries the crash tracer's "died in" line; a passing child passes; a skipping child skips.
    This is the guard that keeps one native crash from erasing the whole GPU suite's evidence (round 4)."""
    import conftest
    from _pytest.outcomes import Failed, Skipped
    with pytest.raises(Failed) as ei:
        conftest.run_isolated("tests/isolation_probe.py::test_probe_aborts", marker="not gpu")
    msg = str(ei.value)
    assert "killed by signal 6" in msg, msg[-800:]
    assert "[bnerv-crashtrace] died in: [bnerv-trail] START tests/isolation_probe.py::test_probe_aborts" in msg, msg[-800:]
    assert "passed" in conftest.run_isolated("tests/isolation_probe.py::test_probe_passes", marker="not gpu")
    with pytest.raises(Skipped):
        conftest.run_isolated("tests/isolation_probe.py::test_probe_skips", marker="not gpu")


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` without a launcher re-launches itself one rank per GPU; on a node with fewer devices it must say so and
    exit instead of hanging in a rendezvous (the driver's 8-GPU line on a smaller box)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BNERV_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible on this node" in (r.stderr + r.stdout), (r.stderr + r.stdout)[-500:]


def test_gpu_sessions_end_without_interpreter_teardown_and_keep_their_status():
    """tests/conftest.py::pytest_unconfigure with BNERV_HARD_EXIT=1 / force (round 5's behaviour, opt-in since the exit-time abort was
    root-caused: runtime.quiesce_autograd): the session leaves through runtime.hard_exit -- atexit callbacks run, then os._exit with pytest's
    status.  Forced on here without a GPU: the summary is printed, the exit status is pytest's (0 for a pass, 5 for "no tests ran"), and an
    atexit callback registered before pytest started still runs."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "sitecustomize.py"), "w") as f:
            f.write("import atexit, os\natexit.register(lambda: open(os.path.join(%r, 'ran'), 'w').write('x'))\n" % td)
        env = dict(os.environ, BNERV_HARD_EXIT="force", PYTHONPATH=td + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/isolation_probe.py::test_probe_passes", "-q", "-p", "no:cacheprovider"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "1 passed" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
        assert os.path.exists(os.path.join(td, "ran")), "atexit callbacks must run before the hard exit"
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/isolation_probe.py", "-q", "-p", "no:cacheprovider", "-k", "nothing_matches"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 5, (r.returncode, r.stdout[-300:])


def test_hard_exit_leaves_the_regular_way_under_a_profiler():
    """runtime.hard_exit: by default (round 6) the REGULAR teardown after quiescing torch's autograd worker -- the exit handler of a C
    library (tests/native/exitmark.c) runs.  With BNERV_HARD_EXIT=1 it skips the interpreter teardown (os._exit): the handler does not
    run -- except under a profiler's environment, where the script must still leave the regular way (rocprofv3 writes its traces from a
    C-level exit handler; `rocprofv3 --kernel-trace --stats -- python bench.py` once left an empty directory).  The exit status is kept
    in every case."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libexitmark.so")
        subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(ROOT, "tests", "native", "exitmark.c")], check=True, timeout=120)
        path = os.path.join(td, "prog.py")
        with open(path, "w") as f:
            f.write("import ctypes, sys\n"
                    "sys.path.insert(0, %r)\n"
                    "ctypes.CDLL(%r)\n"
                    "from boosting_nerv_amd.runtime import hard_exit\n"
                    "print('result line', flush=True)\n"
                    "hard_exit(3)\n" % (ROOT, so))
        base = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "HSA_TOOLS"))}
        base.pop("LD_PRELOAD", None)
        # (ROCPROF_OUTPUT_PATH: one of the variables rocprofv3 always sets for its child, and one the ROCm runtime itself does not act on)
        base.pop("BNERV_HARD_EXIT", None)
        dflt = subprocess.run([sys.executable, path], env=base, capture_output=True, text=True, timeout=120)
        assert dflt.returncode == 3 and "result line" in dflt.stdout and "C-LEVEL-EXIT-HANDLER" in dflt.stdout, (dflt.returncode, dflt.stdout, dflt.stderr[-300:])
        hard = dict(base, BNERV_HARD_EXIT="1")
        plain = subprocess.run([sys.executable, path], env=hard, capture_output=True, text=True, timeout=120)
        assert plain.returncode == 3 and "result line" in plain.stdout and "C-LEVEL-EXIT-HANDLER" not in plain.stdout, (plain.returncode, plain.stdout, plain.stderr[-300:])
        tool = subprocess.run([sys.executable, path], env=dict(hard, ROCPROF_OUTPUT_PATH=td), capture_output=True, text=True, timeout=120)
        assert tool.returncode == 3 and "result line" in tool.stdout and "C-LEVEL-EXIT-HANDLER" in tool.stdout, (tool.returncode, tool.stdout, tool.stderr[-300:])
