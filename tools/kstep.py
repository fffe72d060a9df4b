"""Where does a C1 step go: GPU time of the graph replay (HIP events) vs wall time of the whole step call vs CPU time of the
replay call itself.  usage: python tools/kstep.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo
dev = torch.device("cuda:0")
args, model = bench.build("c1")
model = model.to(dev)
opt = Adan(model.parameters(), lr=args.lr)
vid = SyntheticVideo(132, 720, 1280)
frames = torch.stack([vid.frame(i, device=dev) for i in range(16)])
norm = torch.tensor([(i + 1) / 132 for i in range(16)], dtype=torch.float64, device=dev)
step = TrainStep(model, opt, args.loss, False, (1, 3, 720, 1280), dev, use_graph=True, warmup_eager=3)
for s in range(8):
    step(frames[s % 16:s % 16 + 1], norm[s % 16:s % 16 + 1])
torch.cuda.synchronize()
N = 200
t0 = time.time()
for s in range(N):
    step(frames[s % 16:s % 16 + 1], norm[s % 16:s % 16 + 1])
torch.cuda.synchronize()
print(f"full step call      : {(time.time() - t0) / N * 1e3:.3f} ms/step (wall, synced at the end)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for s in range(N):
    step.graph_a.replay()
e1.record()
torch.cuda.synchronize()
print(f"graph replay only   : {e0.elapsed_time(e1) / N:.3f} ms/replay (GPU events, back to back)")
t0 = time.time()
for s in range(N):
    step.graph_a.replay()
t1 = time.time()
torch.cuda.synchronize()
print(f"replay() CPU cost   : {(t1 - t0) / N * 1e3:.3f} ms/call (enqueue only), drained {(time.time() - t0) / N * 1e3:.3f}")
t0 = time.time()
for s in range(N):
    step.opt.prepare_step()
    step.static_img.copy_(frames[s % 16:s % 16 + 1], non_blocking=True)
    step.static_idx.copy_(norm[s % 16:s % 16 + 1], non_blocking=True)
t1 = time.time()
torch.cuda.synchronize()
print(f"per-step host work  : {(t1 - t0) / N * 1e3:.3f} ms/step CPU (prepare_step + 2 copies), drained {(time.time() - t0) / N * 1e3:.3f}")
