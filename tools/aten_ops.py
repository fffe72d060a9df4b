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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(frames[1:2], norm[1:2])
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) > 0]
rows.sort(key=lambda e: -getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)))
for e in rows[:12]:
    print(f"{e.count:4d} x {e.key:28s} device us {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)):9.1f}")
# call sites of the copying ops: count calls that really copy, keyed by the nearest frame inside this package
import collections, traceback
sites = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        out = orig(self, *a, **k)
        copied = name in ("clone", "copy_") or (isinstance(out, torch.Tensor) and out.is_cuda and out.data_ptr() != self.data_ptr())
        if copied and self.is_cuda and self.numel() > 4096:
            fr = [x for x in traceback.extract_stack()[:-1] if "boosting_nerv_amd" in x.filename]
            sites[(name, fr[-1].filename.split("/")[-1] + ":" + str(fr[-1].lineno) if fr else "?", tuple(self.shape))] += 1
        return out
    setattr(torch.Tensor, name, f)
for n in ("clone", "contiguous", "copy_", "reshape"):
    wrap(n)
step(frames[:1], norm[:1])
torch.cuda.synchronize()
for (name, site, shape), c in sites.most_common(40):
    print(f"{c:3d} x {name:11s} {site:32s} {shape}")
