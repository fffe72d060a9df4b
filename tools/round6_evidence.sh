#!/usr/bin/env bash
# Round-6 evidence -> gpurun_out/r06_*: PMC traffic of the step's dominant kernels (stamped with their sources), bench lines of C1 / C3 / C4 / C5,
# rocprofv3 kernel-trace summaries (step kernels only) and per-launch timelines of the replayed steps.  The GPU suite runs are separate
# (tools/suite_repeat.sh, one lease each).   usage (through gpurun, repo root): tools/round5_evidence.sh [fast] [nopmc]
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
# (nopmc: keep the committed traffic profiles -- they are stamped with the kernel sources they were measured on and bench.py refuses a stale one)
if [ "${1:-}" != nopmc ] && [ "${2:-}" != nopmc ]; then
  for k in pair_dk2s pair_dk3s pair_dk1 k2s c4 pair_c4 wide wide_pair; do python tools/pmc_traffic.py $k r06 > /dev/null 2>&1; done
fi
cp $O/r06_traffic_*.json $R/profiles/ 2>/dev/null          # the bench lines below read (and verify) them
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r06_bench_c1.json 2> $O/r06_bench_c1.err
if [ "${1:-}" != fast ]; then
  for c in c3 c4 c5; do python $R/bench.py --config $c 2>/dev/null | tail -1 > $O/r06_bench_$c.json; done
fi
for c in c1 c3 c4; do
  rm -rf /tmp/ks_$c
  BNERV_BENCH_SETTLE_MS=0 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --steps_only > /tmp/ks_$c.log 2>&1
  { echo "# Round 6 -- $c: rocprofv3 --kernel-trace --stats -- python bench.py --config $c --steps 20 --warmup 5 --steps_only (MI355X)";
    echo "# The table covers the step kernels of the whole process: 3 eager + 1 recording + 21 replayed steps (no micro-benchmark, no eval)."; echo;
    python $R/tools/prof_summary.py /tmp/ks_$c 25 40; } > $O/r06_${c}_step_kerneltrace.md 2>&1
done
cd $R
for c in c1 c3 c4; do tools/step_timeline.sh $c $O/r06_timeline_$c.md > /dev/null 2>&1; done
for c in c1 c3 c4 c5; do [ -s $O/r06_bench_$c.json ] && python -c "
import json,sys; d=json.load(open('$O/r06_bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'))"; done
