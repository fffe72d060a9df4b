"""Drop-in shim: the reference's lib/transform_ops.py names, served by the MI355X build."""
from boosting_nerv_amd.lib.transform_ops import *  # noqa: F401,F403
from boosting_nerv_amd.lib.transform_ops import quant_map, ste  # noqa: F401
