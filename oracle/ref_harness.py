"""TEST INFRASTRUCTURE -- import the real reference (/root/reference) on CPU in THIS container.

Used only by ``oracle/make_goldens.py`` (golden-vector generation) and by a handful of
``-m "not gpu"`` tests that skip when /root/reference is absent (it never exists on the GPU box).
Nothing here is shipped or reachable from the product path.

The reference imports several third-party packages at module import time that are not installed
in this image and are never exercised on the hot path.  We register stand-ins for those
*dependencies* (never for reference source files):

  timm.models.layers.trunc_normal_/DropPath  (model_blocks.py:8)   -> torch.nn.init.trunc_normal_ / Identity
  decord.bridge.set_bridge                   (model_blocks.py:9-10) -> no-op
  pytorch_msssim.ms_ssim/ssim                (hnerv_utils.py:8)    -> oracle.msssim_ref  (PARITY UNPINNED)
  torchvision.transforms[.functional]        (hnerv_utils.py:9-10) -> minimal ToTensor/center_crop
  constriction, compressai.ans               (lib/entropy_model.py) -> empty modules
  dahuffman, imageio, tensorboard            (train_nerv_all.py)   -> empty modules
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BNERV_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model_blocks.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch
    import numpy as np
    from . import msssim_ref

    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree

    if "timm" not in sys.modules:
        class DropPath(torch.nn.Identity):
            def __init__(self, drop_prob=0.0):
                super().__init__()
        layers = _mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)
        models = _mod("timm.models", layers=layers)
        _mod("timm", models=models)
    if "decord" not in sys.modules:
        bridge = _mod("decord.bridge", set_bridge=lambda *_a, **_k: None)
        _mod("decord", bridge=bridge)
    if "pytorch_msssim" not in sys.modules:
        _mod("pytorch_msssim", ms_ssim=msssim_ref.ms_ssim, ssim=msssim_ref.ssim)
    if "torchvision" not in sys.modules:
        class ToTensor:
            def __call__(self, pic):
                arr = np.asarray(pic)
                if arr.ndim == 2:
                    arr = arr[:, :, None]
                t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
                return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t

        def center_crop(img, output_size):
            ch, cw = output_size
            w, h = img.size
            top = int(round((h - ch) / 2.0))
            left = int(round((w - cw) / 2.0))
            return img.crop((left, top, left + cw, top + ch))

        def resize(img, size, *a, **k):
            raise NotImplementedError
        functional = _mod("torchvision.transforms.functional", center_crop=center_crop, resize=resize)
        transforms = _mod("torchvision.transforms", ToTensor=ToTensor, functional=functional)
        utils = _mod("torchvision.utils", save_image=lambda *a, **k: None)
        _mod("torchvision", transforms=transforms, utils=utils)
    for name in ("constriction", "compressai", "compressai.ans", "dahuffman", "imageio"):
        if name not in sys.modules:
            _mod(name)
    sys.modules["dahuffman"].__dict__.setdefault("HuffmanCodec", object)
    for nm in ("BufferedRansEncoder", "RansDecoder"):          # names lib/entropy_model.py:9 imports (never called by the goldens)
        sys.modules["compressai.ans"].__dict__.setdefault(nm, object)
    if "torch.utils.tensorboard" not in sys.modules:
        try:
            importlib.import_module("torch.utils.tensorboard")
        except Exception:
            class SummaryWriter:
                def __init__(self, *a, **k): pass
                def add_scalar(self, *a, **k): pass
            _mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)


_REF_NAMES = ("lib", "lib.quant_ops", "lib.transform_ops", "lib.entropy_model", "model_blocks", "model_nerv",
              "model_enerv", "model_hnerv", "hnerv_utils", "optimizer", "train_nerv_all")


class _RefModules:
    pass


def load_reference():
    """Import the reference's modules from REFERENCE_ROOT and return them as attributes of one object.

    The reference uses flat top-level names (``import model_blocks``); the product keeps same-named shims at the
    repo root, so the reference is imported with REFERENCE_ROOT first on sys.path and then *removed* from
    ``sys.modules`` again so that the two never alias.
    """
    if not reference_available():
        raise FileNotFoundError(f"reference not found at {REFERENCE_ROOT}")
    install_stubs()
    saved = {n: sys.modules.pop(n) for n in list(sys.modules) if n in _REF_NAMES or n.startswith("lib.")}
    sys.path.insert(0, REFERENCE_ROOT)
    # the reference's lib/ has no __init__.py (namespace package) and would lose against the product's same-named shim package
    # at the repo root whatever the sys.path order: bind the name to the reference directory for the duration of the import
    ref_lib = types.ModuleType("lib")
    ref_lib.__path__ = [os.path.join(REFERENCE_ROOT, "lib")]
    sys.modules["lib"] = ref_lib
    out = _RefModules()
    try:
        for n in ("lib.quant_ops", "lib.transform_ops", "lib.entropy_model", "model_blocks", "model_nerv", "model_enerv", "model_hnerv",
                  "hnerv_utils", "optimizer"):
            setattr(out, n.replace(".", "_"), importlib.import_module(n))
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for n in list(sys.modules):
            if n in _REF_NAMES or n.startswith("lib."):
                del sys.modules[n]
        sys.modules.update(saved)
    return out
