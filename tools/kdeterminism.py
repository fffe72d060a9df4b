"""Is the captured train step bitwise reproducible?  Two runs of N steps from the same parameters, frame order and learning rates; per step
the loss bits and a float64 checksum of every parameter are compared.  usage: [BNERV_* switches] python tools/kdeterminism.py [config=c1] [steps=300] [graph=1]
Prints the first step whose loss or parameters differ between the runs (and which tensors), or 'identical'."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boosting_nerv_amd.engine import TrainStep
from boosting_nerv_amd.optimizer import Adan
from boosting_nerv_amd.synth import SyntheticVideo
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
graph = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
dev = torch.device("cuda:0")
args, model0 = bench.build(cfg)
sd0 = {k: v.clone() for k, v in model0.state_dict().items()}
r = bench.RECIPES[cfg]
NT = min(r["n"], 16)
vid = SyntheticVideo(NT, r["h"], r["w"])
frames = torch.stack([vid.frame(i, device=dev) for i in range(NT)])
norm = torch.tensor([(i + 1) / NT for i in range(NT)], dtype=torch.float64, device=dev)
takes_image = "HNeRV" in args.model


def run():
    _, model = bench.build(cfg)
    model.load_state_dict(sd0)
    model = model.to(dev)
    opt = Adan(model.parameters(), lr=args.lr)
    step = TrainStep(model, opt, args.loss, takes_image, (1, 3, r["h"], r["w"]), dev, use_graph=graph, warmup_eager=3)
    names = [k for k, _ in model.named_parameters()]
    out = []
    for s in range(N):
        fi = (s * 7) % NT
        step(frames[fi:fi + 1], norm[fi:fi + 1])
        out.append((step.loss_out.clone(), torch.stack([p.detach().double().sum() for p in model.parameters()])))
    torch.cuda.synchronize()
    return names, out


names, a = run()
_, b = run()
for s in range(N):
    if not torch.equal(a[s][0], b[s][0]) or not torch.equal(a[s][1], b[s][1]):
        bad = [names[i] for i in range(len(names)) if a[s][1][i] != b[s][1][i]]
        print(f"{cfg} graph={int(graph)}: FIRST DIFFERENCE at step {s}: loss {a[s][0].item()!r} vs {b[s][0].item()!r}; {len(bad)} / {len(names)} parameter tensors differ: {bad[:6]}")
        break
else:
    print(f"{cfg} graph={int(graph)}: identical over {N} steps (loss and every parameter tensor, bit for bit)")
from boosting_nerv_amd.runtime import hard_exit  # noqa: E402
hard_exit(0)
