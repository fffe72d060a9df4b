"""Drop-in module name of the reference (`import model_hnerv`): re-exports boosting_nerv_amd.model_hnerv."""
from boosting_nerv_amd.model_hnerv import *  # noqa: F401,F403
from boosting_nerv_amd import model_hnerv as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
