"""SURVEY hard part 7 / 8(d): is the end-PSNR gap between the HIP path and the reference restatement on a LONG schedule larger than
what the restatement shows against ITSELF?  C1 (NeRV-boost 1.5M, 720x1280, Fusion10_freq, Adan, cosine schedule), E epochs over the
synthetic Bunny-shaped clip, for several frame orders: the HIP path (twice: it is bitwise reproducible) and the oracle restatement
on stock PyTorch-ROCm ops R times (MIOpen / hipFFT reductions use atomics: its runs differ from each other on identical inputs).
usage: python tools/parity_stat.py [epochs=10] [n_orders=3] [repeats=3]     (checker tool: imports the oracle, not part of the product)
PARITY_CFG=c1|c3|c4 picks the recipe (c3 / c4: the 3M models at 1080x1920, wide layers on the split bf16x6 kernels); PARITY_NT limits the
clip to its first NT frames (the stock-ops restatement needs ~0.2 s per 1080p step)."""
import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import cpu_ref
from boosting_nerv_amd import hnerv_utils as hu
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10
NO = int(sys.argv[2]) if len(sys.argv) > 2 else 3
RP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
CFG = os.environ.get("PARITY_CFG", "c1")
RC = bench.RECIPES[CFG]
NT, FH, FW = min(RC["n"], int(os.environ.get("PARITY_NT", RC["n"]))), RC["h"], RC["w"]
dev = torch.device("cuda:0")
args, model0 = bench.build(CFG)
TAKES_IMAGE = "HNeRV" in args.model
STOCK_FWD = {"NeRV_Boost": lambda sd, i: cpu_ref.nerv_boost_forward(sd, norm[i:i + 1]),
             "HNeRV_Boost": lambda sd, i: cpu_ref.hnerv_boost_forward(sd, frames[i:i + 1], norm[i:i + 1]),
             "ENeRV_Boost": lambda sd, i: cpu_ref.enerv_boost_forward(sd, norm[i:i + 1])}[args.model]
sd0 = {k: v.clone() for k, v in model0.state_dict().items()}
vid = SyntheticVideo(NT, FH, FW)
frames = torch.stack([vid.frame(i, device=dev) for i in range(NT)])
norm = torch.tensor([(i + 1) / NT for i in range(NT)], dtype=torch.float64, device=dev)


def schedule(seed):
    g = torch.Generator().manual_seed(seed)
    order = [int(i) for e in range(E) for i in torch.randperm(NT, generator=g)]
    lrs = [args.lr * cpu_ref.lr_mult(((s // NT) + (s % NT) / NT) / E) for s in range(len(order))]
    return order, lrs


def run_hip(order, lrs):
    _, model = bench.build(CFG)
    model.load_state_dict(sd0)
    model = model.to(dev)
    opt = Adan(model.parameters(), lr=lrs[0])
    step = TrainStep(model, opt, args.loss, TAKES_IMAGE, (1, 3, FH, FW), dev, use_graph=True, warmup_eager=3)
    for s, fi in enumerate(order):
        for pg in opt.param_groups:
            pg["lr"] = lrs[s]
        step(frames[fi:fi + 1], norm[fi:fi + 1])
    model.eval()
    with torch.no_grad():
        return torch.stack([hu.psnr_fn_device(model(frames[i:i + 1] if TAKES_IMAGE else norm[i:i + 1], norm_idx=norm[i:i + 1])[0], frames[i:i + 1])
                            for i in range(NT)]).mean().item()


def run_stock(order, lrs):
    sd = {k: v.clone().float().to(dev).requires_grad_(True) for k, v in sd0.items()}
    adan = cpu_ref.AdanState(list(sd.values()), lr=lrs[0])
    for s, fi in enumerate(order):
        adan.lr = lrs[s]
        cpu_ref.train_step(args.model, sd, adan, frames[fi:fi + 1], norm[fi:fi + 1], args.loss)
    with torch.no_grad():
        return torch.stack([cpu_ref.psnr_fn_single(STOCK_FWD(sd, i), frames[i:i + 1]) for i in range(NT)]).mean().item()


print(f"{CFG.upper()} ({args.model}, {FH}x{FW}), {E} epochs x {NT} frames = {E * NT} steps per run; {NO} frame orders; stock-ops restatement repeated {RP}x per order")
print("| order seed | HIP run 1 | HIP run 2 | stock-ops runs | stock spread (max - min) | HIP - mean(stock) |")
print("|---|---|---|---|---|---|")
gaps, spreads, pair_diffs = [], [], []
t0 = time.time()
for k in range(NO):
    seed = 123 + 1000 * k
    order, lrs = schedule(seed)
    h1, h2 = run_hip(order, lrs), run_hip(order, lrs)
    st = [run_stock(order, lrs) for _ in range(RP)]
    gaps.append(h1 - statistics.mean(st))
    spreads.append(max(st) - min(st))
    pair_diffs += [abs(a - b) for i, a in enumerate(st) for b in st[i + 1:]]
    print(f"| {seed} | {h1:.4f} | {h2:.4f} | " + ", ".join(f"{v:.4f}" for v in st) + f" | {spreads[-1]:.4f} | {gaps[-1]:+.4f} |", flush=True)
print(f"\nHIP runs of one order: both reported above (the path is bitwise reproducible: equal digits).  |HIP - mean(stock)|: " + ", ".join(f"{abs(g):.4f}" for g in gaps) +
      f" dB (mean {statistics.mean(abs(g) for g in gaps):.4f});  stock vs stock on identical inputs, all pairs: mean {statistics.mean(pair_diffs):.4f}, max {max(pair_diffs):.4f} dB.")
print(f"wall {time.time() - t0:.0f} s")
