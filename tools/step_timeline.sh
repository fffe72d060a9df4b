#!/usr/bin/env bash
# Per-launch timeline of one REPLAYED (hipGraph) train step: rocprofv3 kernel trace of `bench.py --steps_only`, middle step.
# usage (through gpurun, repo root): tools/step_timeline.sh [config] [out.md]
set -u
R=$PWD; C=${1:-c1}; OUT=${2:-$R/gpurun_out/timeline_$C.md}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_$C
BNERV_BENCH_SETTLE_MS=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$C -- python $R/bench.py --config $C --steps 12 --warmup 6 --steps_only > /tmp/kt_$C.log 2>&1
python $R/tools/ktimeline.py /tmp/kt_$C adan_table_kernel --mid > $OUT 2>&1
tail -2 $OUT
