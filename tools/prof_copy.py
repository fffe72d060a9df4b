import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo
dev = torch.device("cuda:0")
args, model = bench.build("c1"); model = model.to(dev)
opt = Adan(model.parameters(), lr=args.lr)
vid = SyntheticVideo(132, 720, 1280)
frames = torch.stack([vid.frame(i, device=dev) for i in range(4)])
norm = torch.tensor([(i + 1) / 132 for i in range(4)], dtype=torch.float64, device=dev)
step = TrainStep(model, opt, args.loss, False, (1, 3, 720, 1280), dev, use_graph=False)
for s in range(3): step(frames[s:s+1], norm[s:s+1])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(frames[3:4], norm[3:4]); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::mul") and e.device_time_total > 0:
        st = [f for f in (e.stack or []) if "boosting_nerv_amd" in f or "engine" in f][:2]
        print(e.name, [tuple(s) for s in (e.input_shapes or [])][:2] if hasattr(e,'input_shapes') else '', round(e.device_time_total,1), st)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=50)[:2500])
