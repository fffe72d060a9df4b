#!/usr/bin/env bash
# Weight-fragment plan (ABI 4) on the GPU box: its tests, then the bench lines of C1 / C3 / C4 / C5 with the plan (default) and C1 / C3 / C4 without
# (BNERV_WPLAN=0), and the kernel trace of the replayed C1 step -> gpurun_out/wp_*.   usage (through gpurun, repo root): tools/wplan_evidence.sh
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_models.py -x -q -k "plan or trajectory or dp_step or short_schedule or tiny_models" > $O/wp_tests.log 2>&1
tail -3 $O/wp_tests.log
cd /tmp && export TMPDIR=/tmp
for c in c1 c3 c4; do
  BNERV_WPLAN=0 python $R/bench.py --config $c --no_cpu_baseline 2>/dev/null | tail -1 > $O/wp_off_bench_$c.json
  python $R/bench.py --config $c --no_cpu_baseline 2>/dev/null | tail -1 > $O/wp_on_bench_$c.json
  python - <<EOF
import json
for k in ("off", "on"):
    try:
        d = json.load(open("$O/wp_%s_bench_$c.json" % k)); print("$c", k, d["value"], d["ms_per_step"])
    except Exception as e:
        print("$c", k, "failed", e)
EOF
done
rocprofv3 --kernel-trace --stats -d /tmp/ks_c1 -- python $R/bench.py --steps 40 --warmup 6 --no_cpu_baseline > /tmp/ks_c1.log 2>&1
python $R/tools/prof_summary.py /tmp/ks_c1 46 45 > $O/wp_c1_trace.md 2>&1
head -30 $O/wp_c1_trace.md
echo done
