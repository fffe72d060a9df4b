#!/usr/bin/env bash
# kernel-trace summaries of the replayed steps only (no settle matmul in the trace) -> gpurun_out/r06_<cfg>_step_kerneltrace.md
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c1 c3 c4; do
  rm -rf /tmp/ks_$c
  BNERV_BENCH_SETTLE_MS=0 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --steps_only > /tmp/ks_$c.log 2>&1
  { echo "# Round 6 -- $c: rocprofv3 --kernel-trace --stats -- python bench.py --config $c --steps 20 --warmup 5 --steps_only (MI355X, BNERV_BENCH_SETTLE_MS=0)";
    echo "# The table covers the step kernels of the whole process: 3 eager + 1 recording + 21 replayed steps (no micro-benchmark, no eval)."; echo;
    python $R/tools/prof_summary.py /tmp/ks_$c 25 40; } > $O/r06_${c}_step_kerneltrace.md 2>&1
done
