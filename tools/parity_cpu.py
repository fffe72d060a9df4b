"""SURVEY 8(d) parity gate at full size against a DETERMINISTIC comparator: train C1 (NeRV-boost 1.5M, 720x1280, Fusion10_freq, Adan, cosine
schedule) for E epochs on the synthetic Bunny-shaped clip through the HIP path (TrainStep, hipGraph) and through oracle/cpu_ref.py on the HOST
(torch CPU fp32, bitwise reproducible -- the stock-ops restatement on the GPU is not, tools/parity_stat.py), same initial weights, frame order and
learning rates; then evaluate every frame with both models.  ~2.7 s per oracle step on 64 host threads.
usage: python tools/parity_cpu.py [epochs=6] [seed=123]      (checker tool: imports the oracle, not part of the product)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import cpu_ref
from boosting_nerv_amd import hnerv_utils as hu
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo

E = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 123
RC = bench.RECIPES["c1"]
N, FH, FW = RC["n"], RC["h"], RC["w"]
dev = torch.device("cuda:0")
args, model = bench.build("c1")
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
vid = SyntheticVideo(N, FH, FW)
frames = torch.stack([vid.frame(i, device=dev) for i in range(N)])
norm = torch.tensor([(i + 1) / N for i in range(N)], dtype=torch.float64, device=dev)
frames_c, norm_c = frames.cpu(), norm.cpu()
g = torch.Generator().manual_seed(seed)
order = [int(i) for e in range(E) for i in torch.randperm(N, generator=g)]
lrs = [args.lr * cpu_ref.lr_mult(((s // N) + (s % N) / N) / E) for s in range(len(order))]

model = model.to(dev)
opt = Adan(model.parameters(), lr=lrs[0])
step = TrainStep(model, opt, args.loss, False, (1, 3, FH, FW), dev, use_graph=True, warmup_eager=3)
t0 = time.time()
tr_hip = []
for s, fi in enumerate(order):
    for pg in opt.param_groups:
        pg["lr"] = lrs[s]
    _, ps = step(frames[fi:fi + 1], norm[fi:fi + 1])
    tr_hip.append(ps.clone())
torch.cuda.synchronize()
t_hip = time.time() - t0
model.eval()
with torch.no_grad():
    p_hip = torch.stack([hu.psnr_fn_device(model(norm[i:i + 1], norm_idx=norm[i:i + 1])[0], frames[i:i + 1]) for i in range(N)]).mean().item()
tr_hip = torch.stack(tr_hip).flatten().cpu()
print(f"HIP path        : end PSNR {p_hip:.4f} dB   train {t_hip:.1f} s ({len(order) / t_hip:.1f} frames/s)", flush=True)

torch.set_num_threads(min(os.cpu_count() or 8, 64))      # one thread per physical core at most: with SMT siblings the MKLDNN convs crawl (a 128-thread run did not finish an epoch in 50 min)
sd = {k: v.clone().float().requires_grad_(True) for k, v in sd0.items()}
adan = cpu_ref.AdanState(list(sd.values()), lr=lrs[0])
t0 = time.time()
tr_ref = []
for s, fi in enumerate(order):
    adan.lr = lrs[s]
    _, ps, _ = cpu_ref.train_step(args.model, sd, adan, frames_c[fi:fi + 1], norm_c[fi:fi + 1], args.loss)
    tr_ref.append(ps.detach().clone())
    if s % 66 == 65:
        d = (tr_hip[:s + 1] - torch.stack(tr_ref).flatten())
        print(f"  step {s + 1}: {time.time() - t0:.0f} s, train-PSNR difference over the last 66 steps: mean {d[-66:].mean().item():+.5f}, max |.| {d[-66:].abs().max().item():.5f} dB", flush=True)
t_ref = time.time() - t0
with torch.no_grad():
    p_ref = torch.stack([cpu_ref.psnr_fn_single(cpu_ref.nerv_boost_forward(sd, norm_c[i:i + 1]), frames_c[i:i + 1]) for i in range(N)]).mean().item()
print(f"config c1: epochs {E}, frames {N}, steps {len(order)}, order seed {seed}")
print(f"CPU oracle      : end PSNR {p_ref:.4f} dB   train {t_ref:.1f} s ({len(order) / t_ref:.2f} frames/s, {torch.get_num_threads()} threads)")
print(f"difference      : {p_hip - p_ref:+.4f} dB   (gate +-0.02 dB)")
d = tr_hip - torch.stack(tr_ref).flatten()
marks = [0, 1, 2, 5, 10, 20, 50, 100, 200, 263, 395, 600, 791]
print("per-step train-PSNR difference (HIP - oracle), dB:  " + "  ".join(f"[{m}] {d[m].item():+.5f}" for m in marks if m < len(d)))
