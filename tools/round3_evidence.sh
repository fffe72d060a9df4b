#!/usr/bin/env bash
# Round-3 evidence -> gpurun_out/r03_*: bench lines of C1 / C3 / C4 / C5, rocprofv3 kernel-trace summaries of the replayed C1 / C3 / C4 steps,
# the per-launch timeline of one replayed C1 step, PMC traffic of the dominant kernels (stamped with the kernel sources).
# usage (through gpurun, repo root): tools/round3_evidence.sh        then copy gpurun_out/r03_* into profiles/
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python tools/pmc_traffic.py c1 > /dev/null 2>&1
python tools/pmc_traffic.py wide > /dev/null 2>&1
python tools/pmc_traffic.py c4 > /dev/null 2>&1
cp $O/r03_traffic.json $O/r03_traffic_wide.json $O/r03_traffic_c4.json $R/profiles/ 2>/dev/null      # the bench lines below read (and verify) them
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r03_bench_c1.json 2> $O/r03_bench_c1.err
for c in c3 c4 c5; do python $R/bench.py --config $c 2>/dev/null | tail -1 > $O/r03_bench_$c.json; done
for c in c1 c3 c4; do
  rm -rf /tmp/ks_$c
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --steps_only > /tmp/ks_$c.log 2>&1
  { echo "# Round 3 -- $c: rocprofv3 --kernel-trace --stats -- python bench.py --config $c --steps 20 --warmup 5 --steps_only (MI355X)";
    echo "# The table covers the whole process: 3 eager + 1 recording + 21 replayed steps (no micro-benchmark, no eval)."; echo;
    python $R/tools/prof_summary.py /tmp/ks_$c 25 40; } > $O/r03_${c}_step_kerneltrace.md 2>&1
done
cd $R && tools/step_timeline.sh c1 $O/r03_timeline_c1.md > /dev/null 2>&1
echo done
