"""Drop-in module name of the reference (`import model_enerv`): re-exports boosting_nerv_amd.model_enerv."""
from boosting_nerv_amd.model_enerv import *  # noqa: F401,F403
from boosting_nerv_amd import model_enerv as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
