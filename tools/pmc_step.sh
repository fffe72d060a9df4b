#!/usr/bin/env bash
# PMC counters over the kernels of the replayed C1 step: tools/pmc_step.sh "<counters>" <out.md> [config]
set -u
R=$(cd "$(dirname "$0")/.." && pwd); GRP=$1; OUT=$2; C=${3:-c1}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_step
rocprofv3 --kernel-trace --output-format csv --pmc $GRP -d /tmp/pmc_step -- python $R/bench.py --config $C --steps 8 --warmup 6 --steps_only > /tmp/pmc_step.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
f = sorted(glob.glob('/tmp/pmc_step/**/*_counter_collection.csv', recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60] + ' g' + r['Grid_Size']
    acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in acc.values() for c in v})
with open(sys.argv[1], 'w') as o:
    o.write('| kernel grid | n | ' + ' | '.join(names) + ' |\n|---|---|' + '---|' * len(names) + '\n')
    for k, v in acc.items():
        n = max(len(x) for x in v.values())
        o.write(f'| {k} | {n} | ' + ' | '.join(f"{sum(v[c][-4:]) / max(len(v[c][-4:]), 1):.0f}" for c in names) + ' |\n')
PY
tail -3 /tmp/pmc_step.log
