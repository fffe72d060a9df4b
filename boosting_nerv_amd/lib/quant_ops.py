"""Parameter containers of the decoder path.  Mirrors lib/quant_ops.py:18-65 of the reference: ``CustomConv2d`` /
``CustomLinear`` are ``nn.Conv2d`` / ``nn.Linear`` (same constructor, same init RNG consumption, same state_dict
keys) whose forward uses ``dequant_w`` / ``dequant_b`` when a CEM quantiser has set them (lib/quant_ops.py:39-41).

The arithmetic goes to the HIP kernels (ops.conv2d_ps / ops.dense_grouped).  With ``args.quant`` (CEM compression path,
train_nerv_compression.py) every container also carries its learned quantisers and rate dictionaries,
exactly as lib/quant_ops.py:22-37 builds them; the model's cal_params() fills dequant_w / dequant_b each step."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .transform_ops import quant_map


def _setup_quant(m, args, bias):
    """lib/quant_ops.py:22-37 (same attribute names, hence the same state_dict keys weight_quantizer.scale / bias_quantizer.scale)."""
    m.quant = bool(getattr(args, "quant", False))
    if m.quant:
        m.weight_quantizer = quant_map[args.quantizer_w](args.quant_model_bit, signed=True, per_channel=args.per_channel_w)
        m.weight_quantizer.init_form(m.weight)
        if bias:
            m.bias_quantizer = quant_map[args.quantizer_b](args.quant_bias_bit, signed=True, per_channel=args.per_channel_b)
            m.bias_quantizer.init_form(m.bias)
        else:
            m.bias_quantizer = None
        m.bitrate_w_dict = {}
        m.bitrate_b_dict = {}
    m.dequant_w = None
    m.dequant_b = None


class CustomConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, **kargs):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        _setup_quant(self, kargs["args"], bias)

    def effective_weight(self):
        return self.weight if self.dequant_w is None else self.dequant_w

    def effective_bias(self):
        return self.bias if self.dequant_b is None else self.dequant_b

    def hip_supported(self):
        k = self.kernel_size[0]
        return (self.kernel_size[0] == self.kernel_size[1] and k in (1, 3) and self.stride == (1, 1)
                and self.padding == ((k - 1) // 2, (k - 1) // 2) and self.dilation == (1, 1) and self.groups == 1)

    def forward(self, x):
        w, b = self.effective_weight(), self.effective_bias()
        if not self.hip_supported():
            raise NotImplementedError(f"CustomConv2d{tuple(self.weight.shape)} stride={self.stride} padding={self.padding}: only "
                                      f"stride-1 'same' 1x1/3x3 convs exist on the HIP decoder path")
        if self.kernel_size[0] == 1 and x.shape[-2:] == (1, 1):
            y = ops.dense_grouped([x], [w], [b], ["none"])[0]
            return y.view(x.shape[0], -1, 1, 1)
        return ops.conv2d_ps(x, w, b, 1)


class CustomLinear(nn.Linear):
    """ENeRV's 144-token transformer stem (model_enerv.py:19-57) -- ~0.1 GFLOP, stays on stock PyTorch-ROCm ops
    (out of scope for hand kernels)."""

    def __init__(self, in_features, out_features, bias=True, **kargs):
        super().__init__(in_features, out_features, bias=bias)
        _setup_quant(self, kargs["args"], bias)

    def forward(self, x):
        return F.linear(x, self.weight if self.dequant_w is None else self.dequant_w,
                        self.bias if self.dequant_b is None else self.dequant_b)
