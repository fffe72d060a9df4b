"""SURVEY 8(d) parity gate at full size: train C1 (NeRV-boost 1.5M, 720x1280, Fusion10_freq, Adan, cosine schedule) for E epochs
on the synthetic Bunny-shaped clip twice -- through the HIP path (TrainStep, hipGraph) and through the oracle restatement
(oracle/cpu_ref.py, plain torch ops executed by stock PyTorch-ROCm on the same GPU) -- with the same initial weights, frame
order and learning rates, then evaluate every frame with both models.  Prints the end PSNR of both and their difference.
usage: python tools/parity_run.py [epochs] [n_frames] [seed] [c1|c3|c4]     (checker tool: imports the oracle, not part of the product)
For c3 / c4 (1080x1920, 600-frame recipes) pass a small n_frames: the oracle on stock ops runs at 1-3 frames/s there."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import cpu_ref
from boosting_nerv_amd import hnerv_utils as hu
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CFG = sys.argv[4] if len(sys.argv) > 4 else "c1"
RC = bench.RECIPES[CFG]
NT, FH, FW = RC["n"], RC["h"], RC["w"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else NT
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False
args, model = bench.build(CFG)
torch.backends.cudnn.benchmark = False
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
vid = SyntheticVideo(NT, FH, FW)
frames = torch.stack([vid.frame(i, device=dev) for i in range(N)])
norm = torch.tensor([(i + 1) / NT for i in range(N)], dtype=torch.float64, device=dev)
TAKES_IMG = args.model == "HNeRV_Boost"
g = torch.Generator().manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 123)
order = [int(i) for e in range(E) for i in torch.randperm(N, generator=g)]
lrs = [args.lr * cpu_ref.lr_mult(((s // N) + (s % N) / N) / E) for s in range(len(order))]

# ---- HIP path
model = model.to(dev)
opt = Adan(model.parameters(), lr=lrs[0])
step = TrainStep(model, opt, args.loss, TAKES_IMG, (1, 3, FH, FW), dev, use_graph=True, warmup_eager=3)
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
    p_hip = torch.stack([hu.psnr_fn_device(model(frames[i:i + 1] if TAKES_IMG else norm[i:i + 1], norm_idx=norm[i:i + 1])[0], frames[i:i + 1]) for i in range(N)]).mean().item()

# ---- oracle restatement on stock PyTorch-ROCm ops
sd = {k: v.clone().float().to(dev).requires_grad_(True) for k, v in sd0.items()}
adan = cpu_ref.AdanState(list(sd.values()), lr=lrs[0])
t0 = time.time()
tr_ref = []
for s, fi in enumerate(order):
    adan.lr = lrs[s]
    _, ps, _ = cpu_ref.train_step(args.model, sd, adan, frames[fi:fi + 1], norm[fi:fi + 1], args.loss)
    tr_ref.append(ps.clone())
torch.cuda.synchronize()
t_ref = time.time() - t0
with torch.no_grad():
    FWD = {"NeRV_Boost": lambda i: cpu_ref.nerv_boost_forward(sd, norm[i:i + 1]), "ENeRV_Boost": lambda i: cpu_ref.enerv_boost_forward(sd, norm[i:i + 1]),
           "HNeRV_Boost": lambda i: cpu_ref.hnerv_boost_forward(sd, frames[i:i + 1], norm[i:i + 1])}[args.model]
    p_ref = torch.stack([cpu_ref.psnr_fn_single(FWD(i), frames[i:i + 1]) for i in range(N)]).mean().item()
print(f"config {CFG}: epochs {E}, frames {N}, steps {len(order)}")
print(f"HIP path      : end PSNR {p_hip:.4f} dB   train {t_hip:.1f} s ({len(order) / t_hip:.1f} frames/s)")
print(f"oracle on GPU : end PSNR {p_ref:.4f} dB   train {t_ref:.1f} s ({len(order) / t_ref:.1f} frames/s)")
print(f"difference    : {p_hip - p_ref:+.4f} dB")

d = (torch.stack(tr_hip).flatten() - torch.stack(tr_ref).flatten()).cpu()
marks = [0, 1, 2, 5, 10, 20, 50, 100, 200, 400, 800, 1200, 2000, 3000]
print("per-step train-PSNR difference (HIP - oracle), dB:  " + "  ".join(f"[{m}] {d[m].item():+.4f}" for m in marks if m < len(d)))
w = 100
print("mean |diff| over windows of 100 steps: " + "  ".join(f"{d[i:i + w].abs().mean().item():.4f}" for i in range(0, len(d) - w + 1, max(w, (len(d) // 8) // w * w))))
print(f"mean signed diff over the last quarter of the steps: {d[-(len(d) // 4):].mean().item():+.4f} dB")
