"""Stage timing of the one-launch time-embedding head (csrc/dense.hip time_head_kernel) on the C1 model: BNERV_TH_DEBUG = 1 stops after the
positional encoding, 2 after stem layer 0 (stem blocks), 4 skips the modulation blocks; 0 = the whole kernel.  Graph-replayed launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import configs
from boosting_nerv_amd.model_nerv import NeRV_Boost
from boosting_nerv_amd.model_blocks import time_head_forward, mlp_pair_forward, tat_modulations
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = NeRV_Boost(1, args=configs.c1()).to(dev)
pos = torch.tensor([37 / 132], dtype=torch.float64, device=dev)
sfts = []
for layer in m.layers:
    sfts += layer.sft_layers()


def timeit(fn, reps=50):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def five():
    t = m.pe_t(pos[:, None], round_to_f32=True)
    o, z = mlp_pair_forward([m.stem, m.stem_t], [t, t])
    tat_modulations(sfts, z)


print(f"five launches: {timeit(five):.2f} us")
for dbg in ("1", "2", "4", "0"):
    os.environ["BNERV_TH_DEBUG"] = dbg
    print(f"one launch, BNERV_TH_DEBUG={dbg}: {timeit(lambda: time_head_forward(m.pe_t, pos, m.stem, m.stem_t, sfts)):.2f} us")
