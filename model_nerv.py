"""Drop-in module name of the reference (`import model_nerv`): re-exports boosting_nerv_amd.model_nerv."""
from boosting_nerv_amd.model_nerv import *  # noqa: F401,F403
from boosting_nerv_amd import model_nerv as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
