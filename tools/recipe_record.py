"""Full-recipe record: scripts/regression/bunny/nerv_boost.sh:4-9 of the reference -- NeRV-boost 1.5M, 300 epochs, evaluation
every 30 epochs, 8-bit quantised twin, Huffman bits per pixel -- through THIS repo's train_nerv_all.py CLI on the synthetic Bunny-shaped clip
(132 x 3x720x1280; no dataset ships), run TWICE with the clip resident in HBM, plus a shorter run on PNG files of the same clip through the
reference's DataLoader path (VideoDataSet, shuffle=True, num_workers=4, pin_memory, one host -> device copy per step).

    python tools/recipe_record.py [epochs=300] [tag=r05]        (through gpurun, repo root)
        -> gpurun_out/<tag>_cli_c1_e<epochs>.txt   both invocations' wall time, the script's own "Training wo evaluation" line, every Eval line
        -> gpurun_out/<tag>_cli_c1_e<epochs>.json  the numbers bench.py puts on its line as `recipe`

The second invocation must reproduce the first one's final checkpoint BIT FOR BIT (SHA-256 over every parameter tensor): the step is
free of atomics and the loader order is seeded."""
import hashlib
import json
import os
import re
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
FLAGS = ("--data_path synthetic:bunny --vid bunny --model NeRV_Boost --sft_block res_sft --ch_t 32 --optim_type Adan --conv_type convnext pshuffel_3x3 "
         "--act sin --norm none --crop_list 720_1280 --resize_list -1 --loss Fusion10_freq --embed pe_1.25_80 --fc_hw 9_16 --dec_strds 5 2 2 2 2 --ks 0_3_3 "
         f"--reduce 2 --dec_blks 1 1 2 2 2 --modelsize 0.8 -e {E} --eval_freq 30 --lower_width 12 -b 1 --lr 0.003 --overwrite").split()


def sha_of_checkpoint(path):
    import torch
    sd = torch.load(path, map_location="cpu")["state_dict"]
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def find_ckpt(outf):
    for dp, _, fs in os.walk(os.path.join(R, "output")):
        if "model_latest.pth" in fs and outf in dp:
            return os.path.join(dp, "model_latest.pth")
    return None


def run(outf, flags=None):
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(R, "train_nerv_all.py"), "--outf", outf] + (flags or FLAGS), cwd=R, capture_output=True, text=True)
    wall = time.time() - t0
    txt = p.stdout + ("\n[stderr]\n" + p.stderr[-3000:] if p.returncode else "")
    keep = [l for l in txt.splitlines() if re.search(r"Eval at epoch|Training|After quantization|bits per|Traceback|Error|FPS", l)]
    m = re.search(r"Training wo evaluation complete in: .*?, ([0-9.]+)s", txt)
    train_s = float(m.group(1)) if m else None
    ev = [l for l in txt.splitlines() if l.startswith("Eval at epoch") or "Eval at epoch" in l]
    last = ev[-1] if ev else ""
    metrics = {k: float(v) for k, v in re.findall(r"(\w+): ([-0-9.]+) \|", last)}
    bpp = re.findall(r"bits per pixel: ([0-9.]+)", txt)
    ck = find_ckpt(outf)
    return {"rc": p.returncode, "wall_s": round(wall, 2), "train_wo_eval_s": train_s, "final_eval": metrics, "bpp": float(bpp[-1]) if bpp else None,
            "sha256_final_parameters": sha_of_checkpoint(ck) if ck else None, "lines": keep}


os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
a = run(f"{tag}_recipe_a")
b = run(f"{tag}_recipe_b")
# the loader path of the reference (VideoDataSet over a PNG directory, DataLoader(shuffle=True, num_workers=4, pin_memory=True), one host -> device
# copy of the frame per step): the same recipe for EL epochs on PNG files of the same clip
EL = max(1, min(30, E // 10))
sys.path.insert(0, R)
from boosting_nerv_amd import synth  # noqa: E402
png = synth.dump_png("/tmp/bunny_png", 132, 720, 1280, device="cuda" if __import__("torch").cuda.is_available() else "cpu")
lf = list(FLAGS)
lf[lf.index("--data_path") + 1] = png
lf[lf.index("-e") + 1] = str(EL)
c = run(f"{tag}_recipe_loader", lf + ["--host_frames"])
n_frames = 132 * E
rec = {"recipe": "scripts/regression/bunny/nerv_boost.sh:4-9 (NeRV-boost 1.5M, -e %d, eval every 30, 8-bit twin, Huffman) on the synthetic Bunny-shaped clip, frames resident in HBM (the loader path: `dataloader_path`)" % E,
       "epochs": E, "frames_trained": n_frames,
       "wall_s": a["wall_s"], "train_wo_eval_s": a["train_wo_eval_s"],
       "end_to_end_frames_per_s": round(n_frames / a["train_wo_eval_s"], 1) if a["train_wo_eval_s"] else None,
       "pred_seen_psnr_db": a["final_eval"].get("pred_seen_psnr"), "quant_seen_psnr_db": a["final_eval"].get("quant_seen_psnr"),
       "pred_seen_ssim": a["final_eval"].get("pred_seen_ssim"), "bpp": a["bpp"],
       "second_invocation": {"wall_s": b["wall_s"], "train_wo_eval_s": b["train_wo_eval_s"], "pred_seen_psnr_db": b["final_eval"].get("pred_seen_psnr")},
       "dataloader_path": {"what": f"{EL} epochs of the same recipe with --host_frames on a PNG directory of the clip: PNG decode in 4 DataLoader workers, pinned batch, one host -> device copy per step",
                           "epochs": EL, "train_wo_eval_s": c["train_wo_eval_s"], "frames_per_s": round(132 * EL / c["train_wo_eval_s"], 1) if c["train_wo_eval_s"] else None,
                           "pred_seen_psnr_db": c["final_eval"].get("pred_seen_psnr"), "rc": c["rc"]},
       "bit_identical_checkpoints": bool(a["sha256_final_parameters"] and a["sha256_final_parameters"] == b["sha256_final_parameters"]),
       "sha256_final_parameters": a["sha256_final_parameters"], "rc": [a["rc"], b["rc"]],
       "source": f"profiles/{tag}_cli_c1_e{E}.txt (tools/recipe_record.py {E})"}
json.dump(rec, open(os.path.join(R, "gpurun_out", f"{tag}_cli_c1_e{E}.json"), "w"), indent=1)
with open(os.path.join(R, "gpurun_out", f"{tag}_cli_c1_e{E}.txt"), "w") as f:
    f.write(f"# Full recipe through the CLI, twice (tools/recipe_record.py {E}; MI355X, 1 GPU)\n# python train_nerv_all.py --outf <run> {' '.join(FLAGS)}\n\n")
    f.write(json.dumps({k: v for k, v in rec.items()}, indent=1) + "\n")
    for name, r_ in (("first invocation", a), ("second invocation", b), (f"loader path, {EL} epochs on PNG files (--host_frames)", c)):
        f.write(f"\n## {name}: rc {r_['rc']}, wall {r_['wall_s']} s, training wo evaluation {r_['train_wo_eval_s']} s, sha256(parameters) {r_['sha256_final_parameters']}\n")
        f.write("\n".join(r_["lines"]) + "\n")
print(json.dumps(rec))
