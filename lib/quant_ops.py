"""Drop-in module name of the reference (`from lib.quant_ops import CustomConv2d`): re-exports boosting_nerv_amd.lib.quant_ops."""
from boosting_nerv_amd.lib.quant_ops import CustomConv2d, CustomLinear  # noqa: F401
