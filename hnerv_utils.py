"""Drop-in module name of the reference (`import hnerv_utils`): re-exports boosting_nerv_amd.hnerv_utils."""
from boosting_nerv_amd.hnerv_utils import *  # noqa: F401,F403
from boosting_nerv_amd import hnerv_utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
