"""NeRV_Boost -- host-side mirror of the reference's model_nerv.py:11-94 (constructor, forward signature/return,
state_dict keys, seeded init order); arithmetic on the HIP kernels."""
import time

import torch
import torch.nn as nn

from .model_blocks import *  # noqa: F401,F403
from .model_blocks import (CustomConv2d, NeRV_MLP, NeRVBlock, PositionEncoding, head_out, mlp_pair_forward,
                           tat_modulations, time_branch_forward)
from .lib.quant_ops import CustomLinear


class _CEMHooks:
    """cal_params / get_bitrate_sum / init_data (model_nerv.py:67-94, model_hnerv.py:292-322): the per-step hooks of the CEM
    compression path (train_nerv_compression.py:354-361).  Active on models built with ``args.quant``."""

    def _quant_modules(self):
        return [m for m in self.modules() if type(m) in (CustomConv2d, CustomLinear)]

    def _cal_params_fused(self, entropy_model):
        """The same step through ops.cem_scale_rate: every weight / bias tensor of the model in one fused pass per direction
        instead of ~40 tiny launches per tensor.  Applies to the recipes' setting (per-tensor Scale_T, Gaussian rate, on the GPU)."""
        from . import ops
        from .lib.transform_ops import Scale_T
        mods = self._quant_modules()
        if entropy_model is None or getattr(entropy_model, "distribution", None) != "gaussian" or not mods or not mods[0].weight.is_cuda:
            return False
        items = []                                             # (module, is_bias, tensor, quantizer)
        for m in mods:
            items.append((m, False, m.weight, m.weight_quantizer))
            if m.bias is not None:
                items.append((m, True, m.bias, m.bias_quantizer))
        if any(type(q) is not Scale_T or q.per_channel for _, _, _, q in items):
            return False
        training = self.training
        noises = [None] * len(items)
        if training:
            if entropy_model.noise_source is not None:         # tests: the reference's draw order (one tensor at a time)
                noises = [entropy_model.noise_source(t) for _, _, t, _ in items]
            else:                                              # one draw for the whole model, sliced per tensor
                total = sum(t.numel() for _, _, t, _ in items)
                flat = torch.empty(total, dtype=torch.float32, device=items[0][2].device).uniform_(-0.5, 0.5)
                off = 0
                for i, (_, _, t, _) in enumerate(items):
                    noises[i] = flat[off:off + t.numel()]
                    off += t.numel()
        bits, stats, deqs = ops.cem_scale_rate([t for _, _, t, _ in items], [q.scale for _, _, _, q in items], noises, training)
        self._fused_bits_total = bits.sum()                    # get_bitrate_sum("bitrate") of this step as ONE reduction (not ~400 adds)
        for i, (m, is_bias, t, _) in enumerate(items):
            d = {"bitrate": bits[i], "mean": stats[i, 1], "std": stats[i, 2], "real_bitrate": 0}
            if is_bias:
                m.dequant_b = deqs[i].reshape(t.shape)
                m.bitrate_b_dict.update(d)
            else:
                m.dequant_w = deqs[i].reshape(t.shape)
                m.bitrate_w_dict.update(d)
        return True

    def cal_params_eval_fused(self, entropy_model):
        """Evaluation-time bit accounting of every weight / bias tensor (reference train_nerv_compression.py:466-489: quantise, keep the
        de-quantised tensor, estimate the rounded symbols' bits, code them) without the per-tensor loop: ONE fused quantise + rate pass
        (ops.cem_scale_rate, rounded symbols), ONE device -> host copy of the flat weights, scales and statistics, and the host
        rANS coder (csrc/ans.cpp) over the symbols rebuilt there -- np.rint(w / s) in float32 is the device's round(w / s) bit for
        bit (IEEE division, ties to even).  Fills the modules' dequant_w / dequant_b and rate dictionaries exactly as the loop does;
        returns False when the model is outside the fused kernel's scope (per-channel scales, other quantisers, CPU tensors)."""
        import numpy as np
        from . import ops
        from .lib.entropy_model import ans_encode_gaussian
        from .lib.transform_ops import Scale_T
        mods = self._quant_modules()
        if entropy_model is None or getattr(entropy_model, "distribution", None) != "gaussian" or not mods or not mods[0].weight.is_cuda:
            return False
        items = []
        for m in mods:
            items.append((m, False, m.weight, m.weight_quantizer))
            if m.bias is not None:
                items.append((m, True, m.bias, m.bias_quantizer))
        if any(type(q) is not Scale_T or q.per_channel for _, _, _, q in items):
            return False
        with torch.no_grad():
            tensors = [t.detach() for _, _, t, _ in items]
            bits, stats, deqs = ops.cem_scale_rate(tensors, [q.scale.detach() for _, _, _, q in items], [None] * len(items), False)
            host = torch.cat([t.reshape(-1) for t in tensors] + [q.scale.detach().reshape(-1) for _, _, _, q in items] + [stats.reshape(-1)]).cpu().numpy()
        total = sum(t.numel() for t in tensors)
        scales_h = host[total:total + len(items)]
        stats_h = host[total + len(items):].reshape(len(items), 4)
        off = 0
        for i, (m, is_bias, t, _) in enumerate(items):
            sym = np.rint(host[off:off + t.numel()] / scales_h[i]).astype(np.int32)
            off += t.numel()
            lo, hi = int(sym.min()), int(sym.max())
            std = float(np.clip(stats_h[i, 2], 1e-5, 1e10))                     # (compress_matrix_flatten_gaussian_global's clamp)
            real = int(ans_encode_gaussian(sym, lo, hi if hi > lo else lo + 1, float(stats_h[i, 1]), std).size) * 32
            d = {"bitrate": bits[i], "mean": stats[i, 1], "std": stats[i, 2], "real_bitrate": real}
            if is_bias:
                m.dequant_b = deqs[i].reshape(t.shape)
                m.bitrate_b_dict.update(d)
            else:
                m.dequant_w = deqs[i].reshape(t.shape)
                m.bitrate_w_dict.update(d)
        return True

    def cal_params(self, entropy_model=None):
        self._fused_bits_total = None
        if self.training and getattr(self, "cem_fused", True) and self._cal_params_fused(entropy_model):
            return
        for m in self._quant_modules():
            code_w, quant_w, dequant_w = m.weight_quantizer(m.weight)
            m.dequant_w = dequant_w
            if m.bias is not None:
                code_b, quant_b, dequant_b = m.bias_quantizer(m.bias)
                m.dequant_b = dequant_b
            if entropy_model is not None:
                m.bitrate_w_dict.update(entropy_model.cal_bitrate(code_w, quant_w, self.training))
                if m.bias is not None:
                    m.bitrate_b_dict.update(entropy_model.cal_bitrate(code_b, quant_b, self.training))

    def get_bitrate_sum(self, name="bitrate"):
        if name == "bitrate" and getattr(self, "_fused_bits_total", None) is not None:
            return self._fused_bits_total
        total = 0
        for m in self._quant_modules():
            total = total + m.bitrate_w_dict[name]
            if name in m.bitrate_b_dict.keys():
                total = total + m.bitrate_b_dict[name]
        return total

    def init_data(self):
        for m in self._quant_modules():
            m.weight_quantizer.init_data(m.weight)
            if m.bias is not None:
                m.bias_quantizer.init_data(m.bias)


def decoder_layers_forward(layers, output, t_embed, out_list, mods=None):
    """Run a list of NeRVBlocks with the TAT modulations of ALL blocks evaluated up front in two grouped launches (or handed in by
    the caller: NeRV_Boost's one-launch time-embedding head computes them with the stems)."""
    if mods is None:
        sfts = []
        for layer in layers:
            sfts += layer.sft_layers()
        mods = tat_modulations(sfts, t_embed)
    for i, layer in enumerate(layers):
        output = layer((output, t_embed), mods=(mods[2 * i], mods[2 * i + 1]))
        out_list.append(output)
    return output


class NeRV_Boost(_CEMHooks, nn.Module):
    lazy_flush_ok = True     # every reader of a deferred slab reduction in this model's backward is an operator of this package (engine.TrainStep)
    lazy_dx_ok = True        # the first block's input gradient (the stem pair's queued reduction) is read by the stem MLP's grouped dense backward, which flushes
    dp_hook = None           # engine.TrainStep with two gradient buckets: called with d(loss)/d(stem output), i.e. after the decoder's backward

    def dp_late_parameters(self):
        """Parameters whose gradients the backward produces AFTER the decoder layers': the stem MLP (74 % of the bytes) and every MLP fed
        by the time embedding (stem_t, the TAT modulation convs) -- their grouped dense backward runs last."""
        from .model_blocks import SFTLayer
        late = list(self.stem.parameters()) + list(self.stem_t.parameters())
        for m in self.modules():
            if isinstance(m, SFTLayer):
                late += list(m.parameters())
        return late
    def __init__(self, expansion=1, args=None):
        super().__init__()
        self.encoder = nn.Identity()
        self.pe_t = PositionEncoding(args.embed, args.lfreq)
        self.fc_h, self.fc_w = [int(x) for x in args.fc_hw.split("_")]
        self.fc_dim = args.fc_dim
        mlp_dim_list = [self.pe_t.embed_length] + [256] + [self.fc_h * self.fc_w * self.fc_dim]
        self.stem = NeRV_MLP(dim_list=mlp_dim_list, bias=True, act=args.act, omega=1, args=args)
        self.stem_t = NeRV_MLP(dim_list=[int(self.pe_t.embed_length), int(args.ch_t * 2), args.ch_t], bias=True, act=args.act, omega=1, args=args)

        self.layers = nn.ModuleList()
        ngf = self.fc_dim
        ks_enc, ks_dec1, ks_dec2 = [int(x) for x in args.ks.split("_")]
        for i, stride in enumerate(args.dec_strds):
            if i == 0:
                new_ngf = int(ngf * expansion)
            else:
                new_ngf = int(max(ngf // (1 if stride == 1 else args.reduce), args.lower_width))
            for j in range(args.dec_blks[i]):
                self.layers.append(NeRVBlock(dec_block=True, conv_type=args.conv_type[1], ngf=ngf, new_ngf=new_ngf,
                                             ks=min(ks_dec1 + 2 * i, ks_dec2), strd=1 if j else stride, bias=True, norm=args.norm,
                                             act=args.act, sft_ngf=args.ch_t, args=args, dump_features=False))
                ngf = new_ngf
        self.head_layer = CustomConv2d(ngf, 3, 1, 1, bias=True, args=args)
        self.out_bias = args.out_bias
        self.outf = args.outf
        self.time_decode = False        # True: synchronise and report wall-clock dec_time like the reference (:58-60)

    def forward(self, input, input_embed=None, norm_idx=None):
        dec_start = time.time()
        sfts = []
        for layer in self.layers:
            sfts += layer.sft_layers()
        head = time_branch_forward(self.pe_t, input, self.stem, self.stem_t, sfts) if input.dtype == torch.float64 else None
        if head is not None:                               # PE -> stem | stem_t -> every TAT modulation: TWO launches (ops.time_branch)
            output, t_embed, mods = head
        else:
            t_embed = self.pe_t(input[:, None], round_to_f32=True)     # pe_t(input[:, None].float()) without the conversion launch
            output, t_embed = mlp_pair_forward([self.stem, self.stem_t], [t_embed, t_embed])
            mods = None
        output = output.view(output.size(0), self.fc_dim, self.fc_h, self.fc_w)
        if self.dp_hook is not None and output.requires_grad:
            output.register_hook(self.dp_hook)             # fires when every decoder layer's backward has run (engine.TrainStep, two buckets)
        out_list = []
        output = decoder_layers_forward(self.layers, output, t_embed, out_list, mods=mods)
        img_out = head_out(self.head_layer, output, self.out_bias)
        if self.time_decode and torch.cuda.is_available():
            torch.cuda.synchronize()
        dec_time = time.time() - dec_start
        return img_out, out_list, dec_time

    def decoder_params(self):
        return (sum([p.data.nelement() for p in self.parameters()])) / 1e6
