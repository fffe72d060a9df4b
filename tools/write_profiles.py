"""Turn the files tools/round_evidence.sh left in gpurun_out/<tag>_* into the tracked summaries under profiles/r01_<tag>_*.
usage: python tools/write_profiles.py f"""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "f"
O, P = os.path.join(R, "gpurun_out", tag + "_"), os.path.join(R, "profiles", f"r01_{tag}_")
for c in ("c1", "c3", "c4", "c5"):
    shutil.copy(O + f"bench_{c}.json", P + f"bench_{c}.json")
pmc = open(O + "pmc.txt").read()
blocks = re.split(r"\n(?=pmc_)", pmc)
val = {}
for b in blocks:
    m = re.match(r"pmc_(\w+?)_\d", b)
    if not m:
        continue
    for cm in re.finditer(r"(FETCH_SIZE|WRITE_SIZE)\s+(\d+)", b):
        val[(m.group(1), cm.group(1))] = int(cm.group(2))
f, w = val[("conv", "FETCH_SIZE")], val[("conv", "WRITE_SIZE")]
json.dump({"kernel": "conv_lean_kernel<3,IN_AFFINE,EP_BIAS,3> 12->12 3x3 @720x1280", "fetch_size_kb": f, "fetch_correction": 2, "write_size_kb": w,
           "hbm_bytes_per_launch": (2 * f + w) * 1024,
           "source": f"profiles/r01_{tag}_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 FETCH correction)"},
          open(P + "traffic.json", "w"), indent=1)
mb = lambda kb: kb * 1024 / 1e6
rows = [("conv", "conv_lean_kernel<3,IN_AFFINE,EP_BIAS,3> 12->12 @720p", "88.5 MB"), ("wgrad", "wgrad_lean_kernel<3,PLAIN,0> 12->12 @720p", "88.5 MB + slabs"),
        ("conv38", "conv_lean2_kernel<3,IN_AFFINE,EP_BIAS,NTB=3> 38->38 @1080p", "630.4 MB (halo rows/columns of 3 K chunks re-read)"),
        ("wgrad38", "wgrad_wide_kernel<PLAIN,0,3,6> 38->38 @1080p", "630.4 MB + slabs (4 column groups re-read the gradient tile; part of it hits L2)")]
tbl = "| kernel | FETCH_SIZE x2 | WRITE_SIZE | HBM bytes / launch | algorithmic bytes / launch |\n|---|---|---|---|---|\n"
for key, name, alg in rows:
    fk, wk = val[(key, "FETCH_SIZE")], val[(key, "WRITE_SIZE")]
    tbl += f"| {name} | 2 x {fk} KB = {mb(2 * fk):.1f} MB | {wk} KB = {mb(wk):.1f} MB | {mb(2 * fk + wk):.1f} MB | {alg} |\n"
open(P + "pmc.md", "w").write(f"""# Round 1 ({tag}) — PMC counters of the dominant kernels, rocprofv3 --pmc, one counter group per pass, MI355X

Command per pass: `rocprofv3 --kernel-trace --output-format csv --pmc <counters> -- python tools/kone.py conv|wgrad|conv38|wgrad38 4`
(tools/round_evidence.sh, tools/pmc_parse.py; chip-wide sums per dispatch, 4 dispatches averaged).  FETCH_SIZE / WRITE_SIZE in KB;
FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section).  conv / wgrad: 12->12 3x3 @720x1280 (C1); conv38 / wgrad38: 38->38 3x3 @1080x1920 (C3).

{tbl}
```
{pmc}```
""")
b1 = json.load(open(O + "bench_c1.json"))
hdr = lambda title, cmd, extra="": f"# Round 1 ({tag}) — {title}\n\nCommand: `{cmd}`.\n{extra}\n"
open(P + "c1_step_kerneltrace.md", "w").write(hdr("C1 train step, end of round; rocprofv3 --kernel-trace --stats (eager, 40+6 steps incl. setup), MI355X",
    f"rocprofv3 --kernel-trace --stats -d /tmp/kt_c1 -- python bench.py --steps 40 --warmup 6 --no_cpu_baseline --no_graph` (tools/round_evidence.sh {tag}",
    f"Bench line of the same build (profiles/r01_{tag}_bench_c1.json): {b1['value']} frames/s, {b1['ms_per_step']} ms/step (hipGraph); dominant conv "
    f"{b1['roofline']['avg_launch_us']} us/launch = {b1['roofline']['achieved']} TFLOP/s = {100 * b1['roofline']['frac']:.1f} % of the fp32 MFMA peak; CPU oracle "
    f"{b1['cpu_baseline']['value']} frames/s on {b1['cpu_baseline']['cores']} threads.  PMC counters: profiles/r01_{tag}_pmc.md.  Rows `at::native::*` are the "
    f"synthetic-clip generator (setup).\n") + open(O + "c1_trace.md").read())
for c, name in (("c3", "HNeRV-boost 3M, 1080x1920"), ("c4", "E-NeRV-boost 3M, 1080x1920")):
    b = json.load(open(O + f"bench_{c}.json"))
    open(P + f"{c}_step_kerneltrace.md", "w").write(hdr(f"{c.upper()} train step ({name}); rocprofv3 --kernel-trace --stats (20+5 steps incl. graph capture warm-up), MI355X",
        f"rocprofv3 --kernel-trace --stats -d /tmp/kt_{c} -- python bench.py --config {c} --steps 20 --warmup 5 --no_cpu_baseline",
        f"Bench line of the same build (profiles/r01_{tag}_bench_{c}.json): {b['value']} frames/s, {b['ms_per_step']} ms/step.\n") + open(O + f"{c}_trace.md").read())
open(P + "kernel_microbench.md", "w").write(f"# Round 1 ({tag}) — hot kernels through the C-ABI (HIP events), MI355X\n\n## C1 shapes (tools/kbench.py 30)\n\n```\n" + open(O + "kbench.txt").read() +
    "```\n\n## Wide conv kernel vs the generic kernel (tools/klean2.py; same inputs, outputs compared)\n\n```\n" + open(O + "klean2.txt").read() +
    "```\n\n## Wide weight-gradient kernel vs the general kernel (tools/kwide.py; same inputs, dw/db compared)\n\n```\n" + open(O + "kwide.txt").read() + "```\n")
for c in ("c1", "c3", "c4", "c5"):
    b = json.load(open(O + f"bench_{c}.json"))
    print(c, b["value"], b["ms_per_step"], (b.get("roofline") or {}).get("achieved"), (b.get("cpu_baseline") or {}).get("value"))
