"""Evaluation-time bit accounting of the C5 model (HNeRV-boost 3M --quant, 1080p recipe): the reference's tensor-by-tensor loop
(train_nerv_compression.py:466-489) against the fused form (model_nerv._CEMHooks.cal_params_eval_fused).  usage: python tools/kcem_eval.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import configs
from boosting_nerv_amd.model_hnerv import HNeRV_Boost
from boosting_nerv_amd.lib.entropy_model import DiffEntropyModel
torch.manual_seed(1)
model = HNeRV_Boost(configs.c5()).to("cuda:0")
model.init_data(); model.eval()
em = DiffEntropyModel("gaussian")
mods = model._quant_modules()
def loop():
    with torch.no_grad():
        for m in mods:
            for kind in ("weight", "bias"):
                t = getattr(m, kind)
                if t is None:
                    continue
                code, sym, deq = getattr(m, f"{kind}_quantizer")(t)
                (m.bitrate_w_dict if kind == "weight" else m.bitrate_b_dict).update(em.cal_bitrate(code, sym, False))
for name, fn in (("tensor loop", loop), ("fused", lambda: model.cal_params_eval_fused(em))):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"{name:12s} {dt * 1e3:8.1f} ms   estimated bits {float(model.get_bitrate_sum('bitrate')):.1f}   coded bits {int(model.get_bitrate_sum('real_bitrate'))}   tensors {sum(1 + (m.bias is not None) for m in mods)}")
