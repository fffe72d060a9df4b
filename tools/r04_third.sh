#!/usr/bin/env bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python bench.py --steps_only --steps 200 > $O/r04c_steps_fused.json 2>$O/r04c_steps_fused.err
BNERV_TATF=0 python bench.py --steps_only --steps 200 > $O/r04c_steps_unfused.json 2>/dev/null
tools/step_timeline.sh c1 $O/r04c_timeline_c1.md > /dev/null 2>&1
( timeout 1800 python -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -150 ) > $O/r04c_pytest.txt
cat $O/r04c_steps_fused.json $O/r04c_steps_unfused.json; grep tat_fused $O/r04c_timeline_c1.md | cut -c1-120; tail -5 $O/r04c_pytest.txt
