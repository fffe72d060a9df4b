#!/usr/bin/env bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "time_head or tiny_models or c1_full or trajectory or reproducible" 2>&1 | tail -15 ) > $O/r04h_pytest.txt
python bench.py --steps_only --steps 200 > $O/r04h_steps_head.json 2>$O/r04h_steps.err
BNERV_TIME_HEAD=0 python bench.py --steps_only --steps 200 > $O/r04h_steps_nohead.json 2>/dev/null
tools/step_timeline.sh c1 $O/r04h_timeline_c1.md > /dev/null 2>&1
tail -4 $O/r04h_pytest.txt; cat $O/r04h_steps_head.json $O/r04h_steps_nohead.json; head -14 $O/r04h_timeline_c1.md | cut -c1-110
