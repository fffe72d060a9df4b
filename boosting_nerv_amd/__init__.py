"""boosting_nerv_amd -- MI355X-native conditional-decoder train path of Boosting-NeRV.

HIP/CDNA4 kernels behind a C-ABI (include/bnerv.h, csrc/) plus the host-side mirror of the reference's module API
(model_blocks / model_nerv / model_enerv / model_hnerv / hnerv_utils / optimizer / train_nerv_all)."""
__version__ = "0.1.0"
