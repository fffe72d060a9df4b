#!/usr/bin/env bash
# 4x4x1 form of the 12-channel weight gradient (BNERV_WGRAD4): parity tests, step A/B, pair kernel rows
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -p no:cacheprovider -k "tat_block or snerv or conv2d or wgrad or tiny_models or c1_full or trajectory or reproducible" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 ) > $O/r04g_pytest.txt
python bench.py --steps_only --steps 200 > $O/r04g_steps_wg4.json 2>$O/r04g.err
BNERV_WGRAD4=0 python bench.py --steps_only --steps 200 > $O/r04g_steps_nowg4.json 2>/dev/null
python bench.py --steps_only --steps 200 > $O/r04g_steps_wg4b.json 2>/dev/null
tools/step_timeline.sh c1 $O/r04g_timeline_c1.md > /dev/null 2>&1
cat $O/r04g_pytest.txt $O/r04g_steps_wg4.json $O/r04g_steps_nowg4.json $O/r04g_steps_wg4b.json; grep -E "pair_kernel|wgrad_lean" $O/r04g_timeline_c1.md | cut -c1-130
