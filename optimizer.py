"""Drop-in module name of the reference (`import optimizer`): re-exports boosting_nerv_amd.optimizer."""
from boosting_nerv_amd.optimizer import *  # noqa: F401,F403
from boosting_nerv_amd import optimizer as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
