"""Summarise a rocprofv3 results.db (kernel-trace) as a per-kernel table: calls, total ms, avg us, % -- markdown to stdout.
usage: prof_summary.py <dir> [steps] [max rows]
Kernels that are NOT part of the train step are listed apart and excluded from the totals: bench.py synthesises its clip on the GPU with
stock torch ops before the first step (at::native::* element-wise / reduction / index kernels, rocPRIM scans, the fill and copy kernels of
tensor construction) -- round 4's tables counted them as "kernel time per step" and contradicted the per-launch timelines."""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))


def foreign(name):
    return name.startswith("void at::native") or "at::native::" in name or "rocprim::" in name or name.startswith("void at::") or "at::cuda::" in name


own = [r for r in rows if not foreign(r[0])]
other = [r for r in rows if foreign(r[0])]
tot = sum(r[2] for r in own)
print("| kernel | calls | calls/step | total ms | avg us | % of step kernels |\n|---|---|---|---|---|---|")
for name, calls, total, avg, pct in own[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:80]
    print(f"| {n} | {calls} | {calls / steps:.1f} | {total / 1e3:.3f} | {avg:.2f} | {100.0 * total / max(tot, 1e-9):.2f} |")
print(f"\nstep kernels: total {tot / 1e3:.2f} ms over {steps:.0f} steps = {tot / 1e3 / steps:.3f} ms/step (durations in the DB are microseconds)")
if other:
    print(f"excluded (clip synthesis and tensor construction with stock torch ops, before the first step): {len(other)} kernels, "
          f"{sum(r[1] for r in other)} launches, {sum(r[2] for r in other) / 1e3:.2f} ms in all")
