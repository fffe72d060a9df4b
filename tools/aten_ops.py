"""Which stock torch (ATen) kernels does one eager train step still launch?  usage: python tools/aten_ops.py [config=c3]
Prints the CUDA-kernel-launching ATen ops of one eager step with call counts and the Python call sites of the most frequent ones
(profiling aid: every such op is a candidate for fusion into the library's launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
args, model = bench.build(cfg)
dev = torch.device("cuda:0")
model = model.to(dev)
opt = Adan(model.parameters(), lr=args.lr)
r = bench.RECIPES[cfg]
vid = SyntheticVideo(r["n"], r["h"], r["w"])
frames = torch.stack([vid.frame(i, device=dev) for i in range(2)])
norm = torch.tensor([(i + 1) / r["n"] for i in range(2)], dtype=torch.float64, device=dev)
step = TrainStep(model, opt, args.loss, "HNeRV" in args.model, (1, 3, r["h"], r["w"]), dev, use_graph=False)
for _ in range(3):
    step(frames[:1], norm[:1])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(frames[1:2], norm[1:2])
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.key.startswith("aten::") and getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) > 0]
rows.sort(key=lambda e: -e.count)
tot = {}
for e in rows:
    tot[e.key] = tot.get(e.key, 0) + e.count
print("ATen ops that launch kernels in one eager step:", sorted(tot.items(), key=lambda kv: -kv[1]))
for e in rows[:28]:
    st = [s for s in e.stack if "boosting_nerv_amd" in s or "torch/optim" in s][:3]
    print(f"{e.count:4d} x {e.key:28s} dev_us {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)):9.1f}  {' <- '.join(s.split('/')[-1] for s in st)}")
