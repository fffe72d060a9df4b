"""Per-step GPU time of the first replays of the captured C1 step (HIP events between steps): how long is the transient after capture?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.hnerv_utils import adjust_lr
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo
dev = torch.device("cuda", 0)
args, model = bench.build("c1")
model = model.to(dev)
opt = Adan(model.parameters(), lr=args.lr)
r = bench.RECIPES["c1"]
vid = SyntheticVideo(r["n"], r["h"], r["w"])
frames = torch.stack([vid.frame(i, device=dev) for i in range(132)])
norm = torch.tensor([(i + 1) / r["n"] for i in range(132)], dtype=torch.float64, device=dev)
step = TrainStep(model, opt, args.loss, False, (1, 3, r["h"], r["w"]), dev, use_graph=True, warmup_eager=3)
args.epochs = 300
step.bind_clip(frames, norm)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
import time
host = []
BUSY = os.environ.get("BUSY", "")
for s in range(N):
    if s == 6 and BUSY:
        if BUSY == "mm":
            a_ = torch.randn(4096, 4096, device=dev); b_ = torch.randn(4096, 4096, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.08:
                for _ in range(10): a_ @ b_
                torch.cuda.synchronize()
        elif BUSY == "copy":
            a_ = torch.empty(1 << 28, device=dev); b_ = torch.empty(1 << 28, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.08:
                for _ in range(10): b_.copy_(a_)
                torch.cuda.synchronize()
        elif BUSY == "sleep":
            torch.cuda.synchronize(); time.sleep(0.08)
    ev[s].record()
    t0 = time.perf_counter()
    adjust_lr(opt, (s / 132) / args.epochs, s % 132, args)
    step.step_frame(s % 132)
    host.append((time.perf_counter() - t0) * 1e3)
ev[N].record()
torch.cuda.synchronize()
print("step: gpu ms between events | host ms in the call")
for s in range(N):
    print(f"{s:3d}: {ev[s].elapsed_time(ev[s + 1]):8.4f} | {host[s]:8.4f}")
