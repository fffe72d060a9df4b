import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
faulthandler.enable()
import torch
import test_gpu_models as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(n):
    T.test_dp_step_path_on_one_rank_group()
    torch.cuda.synchronize()
    print("iteration", i, "ok", flush=True)
