"""HBM traffic + issue mix of one hot kernel of the step from the PMC counters, stamped with the kernel sources it was measured on.
    python tools/pmc_traffic.py <which> [round-tag]      (through gpurun, repo root)   -> gpurun_out/<tag>_traffic_<which>.json + <tag>_pmc_<which>.md
        which: k2s | pair_dk3s | pair_dk2s | pair_dk1   (12 -> 12 @720x1280, C1)
               c4 (K2s 12 -> 12 @1080x1920) | wide (K2s 38 -> 38 @1080x1920)
Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (with --kernel-trace only),
chip-wide sums per dispatch averaged over the dispatches of tools/kone.py; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at
64 B).  bench.py matches a profile to a roofline row by `row` + `shape` and refuses it when the SHA-256 of the listed sources no longer
matches the tree."""
import csv, glob, hashlib, json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "k2s"
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
P12 = 12 * 720 * 1280 * 4
WB12 = 12 * 12 * 9 * 4
SRC4 = ["conv.hip", "conv4.hip", "conv4_body.h", "conv_common.h", "common.h"]
SRCP = ["wgrad.hip", "pairf_body.h", "conv4_body.h", "conv4.hip", "conv_common.h", "common.h", "sidejob.h"]
CFG = {
    "k2s": dict(mode="conv_k2s", pat="conv_", row="k2s", shape=[12, 720, 1280], alg=3 * P12 + WB12, sources=SRC4,
                kernel="K2s: TAT conv0 forward (affine -> 3x3 -> bias -> gelu, gelu') 12->12 @720x1280"),
    "pair_dk3s": dict(mode="pair_dk3s", pat="pair_", row="pair_dk3s", shape=[12, 720, 1280], alg=4 * P12 + 2 * WB12, sources=SRCP,
                      kernel="wA|dK3s: conv1 backward pair (weight gradient, affine prologue | conv^T -> dgelu(saved) + sums) 12->12 @720x1280"),
    "pair_dk2s": dict(mode="pair_dk2s", pat="pair_", row="pair_dk2s", shape=[12, 720, 1280], alg=5 * P12 + 2 * WB12, sources=SRCP,
                      kernel="wA|dK2s: conv0 backward pair (weight gradient, affine prologue | conv^T -> dsin + sums) 12->12 @720x1280"),
    "pair_dk1": dict(mode="pair_dk1", pat="pair_", row="pair_dk1", shape=[12, 720, 1280], alg=3 * P12 + 2 * WB12, sources=SRCP,
                     kernel="wP|dK1: block conv backward pair (weight gradient | conv^T) 12->12 @720x1280"),
    "pair_c4": dict(mode="pair_dk2s_1080", pat="pair_", row="pair_dk2s", shape=[12, 1080, 1920], alg=5 * 12 * 1080 * 1920 * 4 + 2 * WB12, sources=SRCP,
                    kernel="wA|dK2s: conv0 backward pair (weight gradient, affine prologue | conv^T -> dsin + sums) 12->12 @1080x1920 (C4's last stage)"),
    "c4": dict(mode="conv_k2s_1080", pat="conv_", row="k2s", shape=[12, 1080, 1920], alg=3 * 12 * 1080 * 1920 * 4 + WB12, sources=SRC4,
               kernel="K2s: TAT conv0 forward (affine -> 3x3 -> bias -> gelu, gelu') 12->12 @1080x1920"),
    "wide_pair": dict(mode="pair38_dk2s", pat="bfw_kernel", multi=True, row="pair_dk2s", shape=[38, 1080, 1920], alg=5 * 38 * 1080 * 1920 * 4 + 2 * 38 * 38 * 9 * 4,
                      sources=["convbf.hip", "wgrad.hip", "wgrad_bfw_body.h", "split16.h", "conv_common.h", "common.h"],
                      kernel="wA|dK2s 38->38 @1080x1920: conv0 backward as its two launches (wgrad_bfw_kernel + conv_bfw_kernel, DSIN epilogue), counters summed over the two"),
    "wide": dict(mode="conv38_k2s", pat="conv_bfw_kernel", row="k2s", shape=[38, 1080, 1920], alg=945561600, sources=["convbf.hip", "split16.h", "conv_common.h", "common.h"],
                 kernel="K2s 38->38 @1080x1920 on the wide split kernel"),
}
cfg = CFG[which]
out_json = f"{tag}_traffic_{which}.json"
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
vals, log = {}, []
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES",
            "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum TCC_MISS_sum"):
    d = f"/tmp/pmc_{which}_{ctr.split()[0]}"
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", *ctr.split(), "-d", d, "--", sys.executable, os.path.join(R, "tools", "kone.py"), cfg["mode"], "4"],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if cfg["pat"] in r["Kernel_Name"] and "wprep" not in r["Kernel_Name"]]
        kernels = sorted({r["Kernel_Name"] for r in rows}) if cfg.get("multi") else [None]
        for name in ctr.split():
            tot = 0.0
            for kn in kernels:                                   # multi: the pair is several launches -- per-kernel averages, summed
                v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name and (kn is None or r["Kernel_Name"] == kn)]
                if v:
                    tot += sum(v) / len(v)
                    log.append(f"   {name:28s} {sum(v) / len(v):16.0f}  (n={len(v)})  {(kn or rows[0]['Kernel_Name'])[:70]}")
            if tot:
                vals[name] = tot
h = hashlib.sha256()
for f in sorted(cfg["sources"]):
    h.update(open(os.path.join(R, "boosting_nerv_amd", "csrc", f), "rb").read())
hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
json.dump({"kernel": cfg["kernel"], "row": cfg["row"], "shape": cfg["shape"], "fetch_size_kb": vals["FETCH_SIZE"], "fetch_correction": 2, "write_size_kb": vals["WRITE_SIZE"],
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": cfg["alg"], "ratio": round(hbm / cfg["alg"], 4), "sources": cfg["sources"], "src_sha256": h.hexdigest(),
           "source": f"profiles/{out_json} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 FETCH correction; tools/pmc_traffic.py {which})"},
          open(os.path.join(R, "gpurun_out", out_json), "w"), indent=1)
mf = vals.get("SQ_INSTS_MFMA", 0) or 1
hit, miss = vals.get("TCC_HIT_sum", 0), vals.get("TCC_MISS_sum", 0)
open(os.path.join(R, "gpurun_out", f"{tag}_pmc_{which}.md"), "w").write(
    f"# PMC counters of {cfg['kernel']} (tools/pmc_traffic.py {which}; rocprofv3 --kernel-trace --pmc, one counter group per pass, 4 dispatches averaged, MI355X)\n\n"
    f"HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE = {hbm / 1e6:.1f} MB against {cfg['alg'] / 1e6:.1f} MB algorithmic ({hbm / cfg['alg']:.3f}x).\n"
    f"Issue mix per MFMA: {(vals.get('SQ_INSTS_VALU', 0) - mf) / mf:.2f} other VALU, {vals.get('SQ_INSTS_SALU', 0) / mf:.2f} SALU, {vals.get('SQ_INSTS_LDS', 0) / mf:.2f} LDS; "
    f"matrix pipe busy {vals.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(vals.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024, 1) * 100:.0f} % of the launch; LDS bank-conflict cycles {vals.get('SQ_LDS_BANK_CONFLICT', 0):.0f}; "
    f"L2 hit rate {100 * hit / max(hit + miss, 1):.0f} %.\n\n```\n" + "\n".join(log) + "\n```\n")
print(open(os.path.join(R, "gpurun_out", out_json)).read())
