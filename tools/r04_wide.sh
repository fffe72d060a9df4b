#!/usr/bin/env bash
# wide split kernels, pipelined form (BNERV_BFW_PIPE): correctness sweep, kernel timings, C3 / C4 steps with and without it
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( BNERV_BFW_PIPE=1 BNERV_SPLIT_WIDE_MIN_ITEMS=1 timeout 900 python tools/fuzz_wide.py 150 0 2>&1 | tail -12 ) > $O/r04w_fuzz_pipe.txt
( timeout 600 python tools/kwide2.py 10 2>&1 | tail -18 ) > $O/r04w_kwide_pipe.txt
( BNERV_BFW_PIPE=0 timeout 600 python tools/kwide2.py 10 2>&1 | tail -18 ) > $O/r04w_kwide_nopipe.txt
for c in c3 c4; do
  python bench.py --config $c --steps_only --steps 40 > $O/r04w_steps_${c}_pipe.json 2>/dev/null
  BNERV_BFW_PIPE=0 python bench.py --config $c --steps_only --steps 40 > $O/r04w_steps_${c}_nopipe.json 2>/dev/null
done
tail -5 $O/r04w_fuzz_pipe.txt; paste $O/r04w_kwide_pipe.txt $O/r04w_kwide_nopipe.txt | cut -c1-200; cat $O/r04w_steps_*.json
