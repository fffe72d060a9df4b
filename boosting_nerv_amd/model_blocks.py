"""Decoder building blocks -- host-side mirror of the reference's model_blocks.py (same class names, constructor
arguments, sub-module names => same state_dict keys and the same seeded initialisation), with every forward routed to
the HIP kernels through ``ops``.  Reference lines are cited per class.

What is fused where (one HIP launch each):
  NeRVBlock((x, z))  = [conv + bias + PixelShuffle + sin]  ->  [affine + conv3x3 + bias]  ->  [gelu + affine + conv3x3 + bias + residual]
  and the TAT modulation MLPs of ALL blocks of a model are evaluated together by ``tat_modulations`` (2 launches).
"""
import math
from math import ceil, pi, sqrt  # noqa: F401  (star-imported by the model files, like the reference)

import numpy as np  # noqa: F401
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .lib.quant_ops import CustomConv2d, CustomLinear  # noqa: F401


# ----------------------------------------------------------------------------------------------------------------------
# activations / norms                                                              reference model_blocks.py:129-171
# ----------------------------------------------------------------------------------------------------------------------
class Sin(nn.Module):
    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, input):
        return torch.sin(input)


def ActivationLayer(act_type):
    table = {"relu": lambda: nn.ReLU(True), "leaky": lambda: nn.LeakyReLU(inplace=True),
             "leaky01": lambda: nn.LeakyReLU(negative_slope=0.1, inplace=True), "relu6": lambda: nn.ReLU6(inplace=True),
             "gelu": nn.GELU, "sin": Sin, "swish": lambda: nn.SiLU(inplace=True), "softplus": nn.Softplus,
             "hardswish": lambda: nn.Hardswish(inplace=True)}
    if act_type not in table:
        raise KeyError(f"Unknown activation function {act_type}.")
    return table[act_type]()


def NormLayer(norm_type, ch_width):
    if norm_type == "none":
        return nn.Identity()
    if norm_type == "bn":
        return nn.BatchNorm2d(num_features=ch_width)
    if norm_type == "in":
        return nn.InstanceNorm2d(num_features=ch_width)
    raise NotImplementedError


def _act_name(m):
    if isinstance(m, Sin):
        return "sin"
    if isinstance(m, nn.ReLU):
        return "relu"
    if isinstance(m, nn.Identity):
        return "none"
    return None


def OutImg(x, out_bias="tanh"):                                                      # reference model_blocks.py:57-63
    if out_bias == "sigmoid":
        return torch.sigmoid(x)
    if out_bias == "tanh":
        return (torch.tanh(x) * 0.5) + 0.5
    return x + float(out_bias)


def head_out(head_layer, x, out_bias):
    """OutImg(head_layer(x), out_bias): one fused kernel for the 'tanh' form every recipe uses."""
    if out_bias == "tanh" and head_layer.hip_supported():
        return ops.head_tanh(x, head_layer.effective_weight(), head_layer.effective_bias())
    return OutImg(head_layer(x), out_bias)


# ----------------------------------------------------------------------------------------------------------------------
# positional encoding                                                             reference model_blocks.py:108-126
# ----------------------------------------------------------------------------------------------------------------------
class PositionEncoding(nn.Module):
    def __init__(self, pe_embed, lfreq):
        super().__init__()
        self.pe_embed = pe_embed
        if "pe" in pe_embed:
            lbase, levels = [float(x) for x in pe_embed.split("_")[-2:]]
            # the table is built on the host by the reference's own expression (fp32): the kernel only multiplies by it
            if lfreq == "pi":
                self.pe_bases = lbase ** torch.arange(int(levels)) * pi
            else:
                self.pe_bases = lbase ** torch.arange(int(levels)) * float(lfreq)
            self.embed_length = int(2 * levels)
            self._dev_bases = None

    def forward(self, pos, round_to_f32=False):
        """round_to_f32: `pos` is the fp64 index the caller would pass as pos.float() (model_nerv.py:47): rounded inside the kernel."""
        if "pe" not in self.pe_embed:
            return pos.float() if round_to_f32 else pos
        if self._dev_bases is None or self._dev_bases.device != pos.device:
            self._dev_bases = self.pe_bases.to(pos.device)
        return ops.positional_encoding(pos, self._dev_bases, round_to_f32=round_to_f32)


# ----------------------------------------------------------------------------------------------------------------------
# NeRV_MLP                                                                          reference model_blocks.py:66-71
# ----------------------------------------------------------------------------------------------------------------------
class _MLP(nn.Sequential):
    """nn.Sequential of [conv1x1, act, conv1x1, act, ...] (same indices => same keys).  On [B,C,1,1] inputs each
    (conv, act) pair is one fused dense launch."""

    def forward(self, x):
        mods = list(self)
        if x.dim() == 4 and x.shape[-2:] == (1, 1) and all(_act_name(mods[i + 1]) in ("sin", "relu") for i in range(0, len(mods), 2)):
            B = x.shape[0]
            for i in range(0, len(mods), 2):
                conv = mods[i]
                x = ops.dense_grouped([x], [conv.effective_weight()], [conv.effective_bias()], [_act_name(mods[i + 1])])[0]
            return x.view(B, -1, 1, 1)
        for m in mods:
            x = m(x)
        return x


def NeRV_MLP(dim_list, act="relu", bias=True, omega=1.0, args=None):
    act_fn = ActivationLayer(act)           # ONE shared activation instance at every odd index, as in the reference
    fc_list = []
    for i in range(len(dim_list) - 1):
        fc_list += [CustomConv2d(dim_list[i], dim_list[i + 1], kernel_size=1, bias=bias, args=args), act_fn]
    return _MLP(*fc_list)


def mlp_pair_forward(mlps, xs):
    """Evaluate several 2-layer NeRV_MLPs (e.g. stem and stem_t on the same PE embedding) layer-by-layer with one grouped
    launch per depth.  Returns the [B,C,1,1] outputs."""
    depth = len(mlps[0]) // 2
    assert all(len(m) // 2 == depth for m in mlps)
    hs = list(xs)
    B = xs[0].shape[0]
    for d in range(depth):
        convs = [m[2 * d] for m in mlps]
        acts = [_act_name(m[2 * d + 1]) for m in mlps]
        hs = ops.dense_grouped(hs, [c.effective_weight() for c in convs], [c.effective_bias() for c in convs], acts)
    return [h.view(B, -1, 1, 1) for h in hs]


def time_branch_forward(pe_t, pos, stem, stem_t, sft_layers):
    """NeRV_Boost's time-embedding branch -- pe_t(pos), the stem and stem_t MLPs on it, and the (scale, shift) modulations of every SFTLayer
    in ``sft_layers`` from stem_t's output (model_nerv.py:47-51, model_blocks.py:92-105) -- as TWO launches (ops.time_branch) instead of five.
    Returns (stem_out [B, C, 1, 1], t_embed [B, ch_t, 1, 1], [(scale_i, shift_i)]) or None when the branch is not the kernel's (other
    activations, more than 4 frames per step, no positional encoding, quantised containers): the caller takes the layer-by-layer path."""
    if "pe" not in pe_t.pe_embed or len(stem) != 4 or len(stem_t) != 4 or not sft_layers:
        return None
    if any(_act_name(m[i]) != "sin" for m in (stem, stem_t) for i in (1, 3)) or any(_act_name(l.act) != "relu" for l in sft_layers):
        return None
    if pe_t._dev_bases is None or pe_t._dev_bases.device != pos.device:
        pe_t._dev_bases = pe_t.pe_bases.to(pos.device)
    mlps = []
    for l in sft_layers:
        for c0, c1 in ((l.SFT_scale_conv0, l.SFT_scale_conv1), (l.SFT_shift_conv0, l.SFT_shift_conv1)):
            mlps.append((c0.effective_weight(), c0.effective_bias(), c1.effective_weight(), c1.effective_bias()))
    res = ops.time_branch(pos, pe_t._dev_bases,
                          (stem[0].effective_weight(), stem[0].effective_bias(), stem[2].effective_weight(), stem[2].effective_bias()),
                          (stem_t[0].effective_weight(), stem_t[0].effective_bias(), stem_t[2].effective_weight(), stem_t[2].effective_bias()), mlps)
    if res is None:
        return None
    out, zt, mo = res
    B = out.shape[0]
    mods = [(mo[2 * i].view(B, -1, 1, 1), mo[2 * i + 1].view(B, -1, 1, 1)) for i in range(len(sft_layers))]
    return out.view(B, -1, 1, 1), zt.view(B, -1, 1, 1), mods


# ----------------------------------------------------------------------------------------------------------------------
# TAT: SFTLayer / ResBlock_SFT                                                  reference model_blocks.py:74-105
# ----------------------------------------------------------------------------------------------------------------------
class SFTLayer(nn.Module):
    def __init__(self, in_ch, out_ch, factor=1, act="relu", omega=1.0, args=None):
        super().__init__()
        self.SFT_scale_conv0 = CustomConv2d(in_ch, in_ch // factor, 1, args=args)
        self.SFT_scale_conv1 = CustomConv2d(in_ch // factor, out_ch, 1, args=args)
        self.SFT_shift_conv0 = CustomConv2d(in_ch, in_ch // factor, 1, args=args)
        self.SFT_shift_conv1 = CustomConv2d(in_ch // factor, out_ch, 1, args=args)
        self.act = ActivationLayer(act_type=act)

    def forward(self, x):
        scale, shift = tat_modulations([self], x[1])[0]
        return ops.sft_affine(x[0], scale, shift)


def tat_modulations(sft_layers, cond):
    """(scale, shift) [B,C,1,1] of every SFTLayer in ``sft_layers`` from the condition vector ``cond`` [B,ch_t,1,1]:
    scale = conv1(act(conv0(cond))) for the scale and shift branches (SFTLayer.forward, model_blocks.py:103-104).
    All first layers go in ONE grouped launch, all second layers in another, regardless of how many blocks there are."""
    n = len(sft_layers)
    act = _act_name(sft_layers[0].act)
    if act not in ("relu", "sin"):
        raise NotImplementedError(f"SFT inner activation {type(sft_layers[0].act).__name__} is not on the HIP path")
    first = []
    for m in sft_layers:
        first += [m.SFT_scale_conv0, m.SFT_shift_conv0]
    hs = ops.dense_grouped([cond] * (2 * n), [c.effective_weight() for c in first], [c.effective_bias() for c in first], [act] * (2 * n))
    second = []
    for m in sft_layers:
        second += [m.SFT_scale_conv1, m.SFT_shift_conv1]
    outs = ops.dense_grouped(hs, [c.effective_weight() for c in second], [c.effective_bias() for c in second], ["none"] * (2 * n))
    B = cond.shape[0]
    return [(outs[2 * i].view(B, -1, 1, 1), outs[2 * i + 1].view(B, -1, 1, 1)) for i in range(n)]


class ResBlock_SFT(nn.Module):
    def __init__(self, in_ch, out_ch, cond_ch, factor=1, in_act="relu", out_act="gelu", omega=1.0, args=None):
        super().__init__()
        self.sft0 = SFTLayer(cond_ch, in_ch, factor, in_act, omega, args=args)
        self.conv0 = CustomConv2d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, args=args)
        self.sft1 = SFTLayer(cond_ch, out_ch, factor, in_act, omega, args=args)
        self.conv1 = CustomConv2d(out_ch, out_ch, kernel_size=3, stride=1, padding=1, args=args)
        self.act = ActivationLayer(act_type=out_act)
        if out_act != "gelu" or in_ch != out_ch:
            raise NotImplementedError("ResBlock_SFT on the HIP path is the reference's form: in_ch == out_ch, out_act='gelu'")

    def sft_layers(self):
        return [self.sft0, self.sft1]

    def forward(self, x, mods=None):
        if mods is None:
            mods = tat_modulations(self.sft_layers(), x[1])
        (s0, t0), (s1, t1) = mods
        return ops.tat_block(x[0], s0, t0, s1, t1, self.conv0.effective_weight(), self.conv0.effective_bias(),
                             self.conv1.effective_weight(), self.conv1.effective_bias())


# ----------------------------------------------------------------------------------------------------------------------
# UpConv / DownConv / NeRVBlock                                       reference model_blocks.py:14-46, :174-220
# ----------------------------------------------------------------------------------------------------------------------
class DownConv(nn.Module):
    def __init__(self, **kargs):
        super().__init__()
        ks, ngf, new_ngf, strd = kargs["ks"], kargs["ngf"], kargs["new_ngf"], kargs["strd"]
        if kargs["conv_type"] != "conv" or ks + strd != 1:
            raise NotImplementedError("DownConv: only the HNeRV_Boost decoder[0] form (conv_type='conv', ks=0, strd=1 -> 1x1 conv) "
                                      "is on the path (model_hnerv.py:200-201); encoder down-convs belong to the non-boost HNeRV baseline")
        self.downconv = CustomConv2d(ngf, new_ngf, ks + strd, strd, ceil(ks / 2), bias=kargs["bias"], args=kargs["args"])
        self.stride = 1

    def conv_module(self):
        return self.downconv

    def forward(self, x):
        return self.downconv(x)


class UpConv(nn.Module):
    def __init__(self, **kargs):
        super().__init__()
        ks, ngf, new_ngf, strd = kargs["ks"], kargs["ngf"], kargs["new_ngf"], kargs["strd"]
        args = kargs["args"]
        ct = kargs["conv_type"]
        if ct == "pshuffel_3x3":
            ks = 3 if ks > 3 else ks
        elif ct != "pshuffel":
            raise NotImplementedError(f"UpConv conv_type={ct!r}: the boost recipes use 'pshuffel_3x3' (and 'pshuffel'); "
                                      f"'conv'/'interpolate' are not on the HIP path")
        self.upconv = nn.Sequential(
            CustomConv2d(ngf, new_ngf * strd * strd, ks, 1, ceil((ks - 1) // 2), bias=kargs["bias"], args=args),
            nn.PixelShuffle(strd) if strd != 1 else nn.Identity(),
        )
        self.stride = strd

    def conv_module(self):
        return self.upconv[0]

    def forward(self, x):
        c = self.upconv[0]
        return ops.conv2d_ps(x, c.effective_weight(), c.effective_bias(), self.stride)


class NeRVBlock(nn.Module):
    def __init__(self, **kargs):
        super().__init__()
        conv = UpConv if kargs["dec_block"] else DownConv
        self.conv = conv(ngf=kargs["ngf"], new_ngf=kargs["new_ngf"], strd=kargs["strd"], ks=kargs["ks"],
                         conv_type=kargs["conv_type"], bias=kargs["bias"], args=kargs["args"])
        self.norm = NormLayer(kargs["norm"], kargs["new_ngf"])
        self.act = ActivationLayer(kargs["act"])
        args = kargs["args"]
        self.dec_block = kargs["dec_block"] or len(args.enc_strds)
        if args.sft_block == "res_sft" and kargs.get("sft_ngf", 0) != 0:
            if not self.dec_block:
                raise NotImplementedError("NeRVBlock without dec_block and without encoder strides (fc-reshape TAT form, "
                                          "model_blocks.py:40-43) is only used by the non-boost HNeRV baseline")
            sft_ch = kargs["new_ngf"]
            self.sft_block = ResBlock_SFT(sft_ch, sft_ch, cond_ch=kargs["sft_ngf"], in_act="relu", out_act="gelu", omega=1, args=args)

    def sft_layers(self):
        return self.sft_block.sft_layers()

    def _fusable(self):
        return isinstance(self.act, Sin) and isinstance(self.norm, nn.Identity) and hasattr(self, "sft_block") \
            and self.conv.conv_module().hip_supported()

    def forward(self, x, mods=None):
        if isinstance(x, tuple):
            if not self._fusable():
                raise NotImplementedError("NeRVBlock((x, z)) on the HIP path needs act='sin', norm='none', sft_block='res_sft'")
            feat, embed = x
            if mods is None:
                mods = tat_modulations(self.sft_layers(), embed)
            (s0, t0), (s1, t1) = mods
            c, sb = self.conv.conv_module(), self.sft_block
            return ops.snerv_block(feat, c.effective_weight(), c.effective_bias(), s0, t0, s1, t1,
                                   sb.conv0.effective_weight(), sb.conv0.effective_bias(),
                                   sb.conv1.effective_weight(), sb.conv1.effective_bias(), self.conv.stride)
        return self.act(self.norm(self.conv(x)))


# ----------------------------------------------------------------------------------------------------------------------
# ConvNeXt encoder of HNeRV_Boost                                               reference model_blocks.py:223-347
# Stock PyTorch-ROCm ops (the content encoder is not part of the hand-written path).
# ----------------------------------------------------------------------------------------------------------------------
class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if self.data_format not in ["channels_last", "channels_first"]:
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        if x.is_cuda and x.dim() == 4 and x.shape[1] <= ops.LN_CF_MAX_C:       # one HIP kernel instead of ~8 elementwise launches (row N3)
            return ops.layernorm_cf(x, self.weight, self.bias, self.eps)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class Block(nn.Module):
    def __init__(self, dim, drop_path=0.0, layer_scale_init_value=1e-6):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True) if layer_scale_init_value > 0 else None
        if drop_path > 0.0:
            raise NotImplementedError("stochastic depth is never enabled by the reference (drop_path_rate=0, model_hnerv.py:188)")
        self.drop_path = nn.Identity()

    def forward(self, x):
        inp = x
        if x.is_cuda and x.shape[1] in ops.CNX_MLP_DIMS:
            # Row N3, whole block on HIP kernels, NCHW end to end (same arithmetic as the channels_last form, no permute copies):
            # depthwise 7x7 -> channel LayerNorm -> ONE fused kernel for pwconv1 -> GELU -> pwconv2 -> gamma -> + input
            x = ops.dwconv(x, self.dwconv.weight, self.dwconv.bias)
            x = ops.layernorm_cf(x, self.norm.weight, self.norm.bias, self.norm.eps)
            return ops.cnx_mlp(x, inp, self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias, self.gamma)
        if x.is_cuda and x.shape[1] <= ops.LN_CF_MAX_C:
            # other widths: HIP depthwise conv and channel LayerNorm, the pointwise linears as 1x1 convs on the same weights
            dim = x.shape[1]
            x = ops.dwconv(x, self.dwconv.weight, self.dwconv.bias)
            x = ops.layernorm_cf(x, self.norm.weight, self.norm.bias, self.norm.eps)
            x = F.conv2d(x, self.pwconv1.weight.view(4 * dim, dim, 1, 1), self.pwconv1.bias)
            x = F.conv2d(self.act(x), self.pwconv2.weight.view(dim, 4 * dim, 1, 1), self.pwconv2.bias)
            if self.gamma is not None:
                x = self.gamma.view(1, dim, 1, 1) * x
            return inp + x
        if x.is_cuda:        # HIP depthwise kernels (row N3); MIOpen only has its naive fp32 fallback for these shapes
            x = ops.dwconv(x, self.dwconv.weight, self.dwconv.bias).permute(0, 2, 3, 1)
        else:
            x = self.dwconv(x).permute(0, 2, 3, 1)
        x = self.pwconv2(self.act(self.pwconv1(self.norm(x))))
        if self.gamma is not None:
            x = self.gamma * x
        return inp + x.permute(0, 3, 1, 2)


class ConvNeXt(nn.Module):
    def __init__(self, stage_blocks=0, strds=[2, 2, 2, 2], dims=[96, 192, 384, 768], in_chans=3, drop_path_rate=0.0,
                 layer_scale_init_value=1e-6):
        super().__init__()
        self.downsample_layers = nn.ModuleList()
        self.stages = nn.ModuleList()
        self.stage_num = len(dims)
        dp_rates = [x.item() for x in torch.linspace(0, drop_path_rate, stage_blocks * self.stage_num)]
        cur = 0
        for i in range(self.stage_num):
            if i > 0:
                ds = nn.Sequential(LayerNorm(dims[i - 1], eps=1e-6, data_format="channels_first"),
                                   nn.Conv2d(dims[i - 1], dims[i], kernel_size=strds[i], stride=strds[i]))
            else:
                ds = nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=strds[i], stride=strds[i]),
                                   LayerNorm(dims[0], eps=1e-6, data_format="channels_first"))
            self.downsample_layers.append(ds)
            self.stages.append(nn.Sequential(*[Block(dim=dims[i], drop_path=dp_rates[cur + j], layer_scale_init_value=layer_scale_init_value)
                                               for j in range(stage_blocks)]))
            cur += stage_blocks
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)      # == timm.models.layers.trunc_normal_ (model_blocks.py:8)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        for i in range(self.stage_num):
            for m in self.downsample_layers[i]:
                x = patchify_conv(x, m) if isinstance(m, nn.Conv2d) else m(x)
            x = self.stages[i](x)
        return x


def patchify_conv(x, conv):
    """nn.Conv2d with kernel_size == stride and no padding (every ConvNeXt down-sampling layer, model_blocks.py:250-262 of the
    reference) as a GEMM over non-overlapping patches: im2col is a pure reshape here.  Same parameters / state_dict; the GEMM
    runs on the MFMA kernel of csrc/gemm.hip (ops.dense_gemm)."""
    s = conv.stride[0]
    if conv.kernel_size != (s, s) or conv.stride != (s, s) or conv.padding != (0, 0) or conv.groups != 1 or conv.dilation != (1, 1):
        return conv(x)
    B, C, H, W = x.shape
    Ho, Wo = H // s, W // s
    xp = x[:, :, :Ho * s, :Wo * s].reshape(B, C, Ho, s, Wo, s).permute(0, 2, 4, 1, 3, 5).reshape(B * Ho * Wo, C * s * s)
    w2d = conv.weight.reshape(conv.out_channels, C * s * s)
    y = ops.dense_gemm(xp, w2d, conv.bias) if xp.is_cuda else F.linear(xp, w2d, conv.bias)     # (CPU branch: the patchify equivalence test)
    return y.reshape(B, Ho, Wo, conv.out_channels).permute(0, 3, 1, 2)
