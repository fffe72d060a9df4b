#!/usr/bin/env bash
# Round-4 evidence -> gpurun_out/r04_*: the GPU suite, PMC traffic of the step's dominant kernels (stamped with their sources), bench lines of
# C1 / C3 / C4 / C5, rocprofv3 kernel-trace summaries and per-launch timelines of the replayed steps.
# usage (through gpurun, repo root): tools/round4_evidence.sh        then copy gpurun_out/r04_* into profiles/
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) > $O/r04_pytest_gpu.txt
for k in pair_dk2s pair_dk3s pair_dk1 k2s c4 wide; do python tools/pmc_traffic.py $k r04 > /dev/null 2>&1; done
cp $O/r04_traffic_*.json $R/profiles/ 2>/dev/null          # the bench lines below read (and verify) them
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r04_bench_c1.json 2> $O/r04_bench_c1.err
for c in c3 c4 c5; do python $R/bench.py --config $c 2>/dev/null | tail -1 > $O/r04_bench_$c.json; done
for c in c1 c3 c4; do
  rm -rf /tmp/ks_$c
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --steps_only > /tmp/ks_$c.log 2>&1
  { echo "# Round 4 -- $c: rocprofv3 --kernel-trace --stats -- python bench.py --config $c --steps 20 --warmup 5 --steps_only (MI355X)";
    echo "# The table covers the whole process: 3 eager + 1 recording + 21 replayed steps (no micro-benchmark, no eval)."; echo;
    python $R/tools/prof_summary.py /tmp/ks_$c 25 40; } > $O/r04_${c}_step_kerneltrace.md 2>&1
done
cd $R
for c in c1 c3 c4; do tools/step_timeline.sh $c $O/r04_timeline_$c.md > /dev/null 2>&1; done
tail -3 $O/r04_pytest_gpu.txt; for c in c1 c3 c4 c5; do python -c "
import json,sys; d=json.load(open('$O/r04_bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), (d.get('cpu_baseline') or {}).get('value'))"; done
