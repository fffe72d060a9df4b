"""Drop-in shim: the reference's lib/entropy_model.py names, served by the MI355X build."""
from boosting_nerv_amd.lib.entropy_model import DiffEntropyModel, LowerBound, ideal_code_bits  # noqa: F401
