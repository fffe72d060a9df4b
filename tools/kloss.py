"""Run the fused loss (Fusion10_freq) at 720x1280 a few times -- for rocprofv3 --kernel-trace --stats (per-kernel times)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boosting_nerv_amd import ops
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
p, t = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 3, H, W, device=dev)
for _ in range(12):
    ops.loss_value_grad_stats(p, t, "Fusion10_freq")
torch.cuda.synchronize()
