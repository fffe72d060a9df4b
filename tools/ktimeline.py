"""Per-launch timeline of ONE eager step from a rocprofv3 kernel trace (CSV): every launch of one complete step in issue
order with its duration, the gap to its predecessor and its grid -- shows which SHAPES the time goes to (a --stats table merges
all resolutions of one template).  usage: python tools/ktimeline.py <dir with *_kernel_trace.csv> [marker kernel substring]
The marker is a kernel launched exactly once per step (default: adan_table_kernel, the optimizer: a step then reads forward, loss, backward, update)."""
import csv
import glob
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adan_table_kernel"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit(f"marker {marker!r} found {len(marks)} times in {f}")
m = len(marks) // 2 if '--mid' in sys.argv else -3   # --mid: a step from the middle of the run (replayed graph steps of a --steps_only run)
a, b = marks[m], marks[m + 1]        # one full step: from just after one marker to the next (marker = end of the loss)
seg = rows[a + 1:b + 1]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
busy = 0
print(f"# one step: {len(seg)} launches, wall {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
print("| # | start us | dur us | gap us | blocks | wg | lds | vgpr | kernel |\n|---|---|---|---|---|---|---|---|---|")
for i, r in enumerate(seg):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
    print(f"| {i} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.2f} | {(s - prev_end) / 1e3:.2f} | {grid // max(wg, 1)} | {wg} | {r.get('LDS_Block_Size', '')} | {r.get('VGPR_Count', '')} | {n} |")
    busy += e - s
    prev_end = e
print(f"\nkernel time {busy / 1e3:.1f} us of {(prev_end - t0) / 1e3:.1f} us")
