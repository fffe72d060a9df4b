#!/usr/bin/env bash
# End-of-round-2 evidence (build with the weight-fragment plan, ABI 4) -> gpurun_out/e3_*: bench lines of C1 / C3 / C4 / C5, kernel traces of the
# replayed C1 / C3 / C4 steps, the eager per-launch timeline of C1.   usage (through gpurun, repo root): tools/round2b_evidence.sh
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/e3_bench_c1.json 2> $O/e3_bench_c1.err
for c in c3 c4 c5; do python $R/bench.py --config $c 2>/dev/null | tail -1 > $O/e3_bench_$c.json; done
for c in c1 c3 c4; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > /tmp/ks_$c.log 2>&1
  { echo "# Round 2 (end, ABI 4 build) -- $c: rocprofv3 --kernel-trace --stats -- python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline (MI355X)";
    echo "# The table covers the whole process: 3 eager + 1 recording + 21 replayed steps, the kernel-family micro-benchmark (5 eager + 5 x 20 replayed launches per member) and the 8-frame eval.";
    echo; python $R/tools/prof_summary.py /tmp/ks_$c 25 40; } > $O/e3_${c}_trace.md 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_c1 -- python $R/bench.py --config c1 --steps 4 --warmup 5 --no_cpu_baseline --no_graph > /tmp/kt_c1.log 2>&1
python $R/tools/ktimeline.py /tmp/kt_c1 loss_final_kernel > $O/e3_timeline_c1.md 2>&1
echo done
