"""Parameter containers of the decoder path.  Mirrors lib/quant_ops.py:18-65 of the reference: ``CustomConv2d`` /
``CustomLinear`` are ``nn.Conv2d`` / ``nn.Linear`` (same constructor, same init RNG consumption, same state_dict
keys) whose forward uses ``dequant_w`` / ``dequant_b`` when a CEM quantiser has set them (lib/quant_ops.py:39-41).

The arithmetic goes to the HIP kernels (ops.conv2d_ps / ops.dense_grouped).  The learned-quantiser branch
(``args.quant``; lib/transform_ops.py) belongs to the CEM compression path, SURVEY 8(f) row N2 -- not built yet."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _no_quant(args):
    if getattr(args, "quant", False):
        raise NotImplementedError("args.quant (CEM learned quantisers, train_nerv_compression.py) is SURVEY 8(f) row N2: "
                                  "not part of this build")


class CustomConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, **kargs):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        _no_quant(kargs["args"])
        self.dequant_w = None
        self.dequant_b = None
        self.quant = False

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
        _no_quant(kargs["args"])
        self.dequant_w = None
        self.dequant_b = None
        self.quant = False

    def forward(self, x):
        return F.linear(x, self.weight if self.dequant_w is None else self.dequant_w,
                        self.bias if self.dequant_b is None else self.dequant_b)
