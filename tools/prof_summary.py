"""Summarise a rocprofv3 results.db (kernel-trace) as a per-kernel table: calls, total ms, avg us, % -- markdown to stdout."""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"| kernel | calls | calls/step | total ms | avg us | % |\n|---|---|---|---|---|---|")
for name, calls, total, avg, pct in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:80]
    print(f"| {n} | {calls} | {calls / steps:.1f} | {total / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")
print(f"\ntotal kernel time {tot / 1e3:.2f} ms over {steps:.0f} steps = {tot / 1e3 / steps:.3f} ms/step (durations in the DB are microseconds)")
