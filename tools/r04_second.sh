#!/usr/bin/env bash
# Round-4 second GPU pass: the fused TAT forward -- its own tests first, then the whole GPU suite, bench line with / without it, timeline.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "tat_block" 2>&1 | tail -25 ) > $O/r04b_pytest_tat.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/r04b_pytest.txt
python bench.py --steps_only --steps 200 > $O/r04b_steps_fused.json 2>$O/r04b_steps_fused.err
BNERV_TATF=0 python bench.py --steps_only --steps 200 > $O/r04b_steps_unfused.json 2>/dev/null
python bench.py --no_cpu_baseline > $O/r04b_bench_c1.json 2> $O/r04b_bench_c1.err
tools/step_timeline.sh c1 $O/r04b_timeline_c1.md > /dev/null 2>&1
tail -4 $O/r04b_pytest_tat.txt; tail -4 $O/r04b_pytest.txt; cat $O/r04b_steps_fused.json $O/r04b_steps_unfused.json
