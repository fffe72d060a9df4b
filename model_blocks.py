"""Drop-in module name of the reference (`import model_blocks`): re-exports boosting_nerv_amd.model_blocks."""
from boosting_nerv_amd.model_blocks import *  # noqa: F401,F403
from boosting_nerv_amd import model_blocks as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
