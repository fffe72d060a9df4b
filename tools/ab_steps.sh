#!/usr/bin/env bash
# A/B of environment switches on the replayed C1 (or other) step: tools/ab_steps.sh TAG CONFIG "ENV1=.. ENV2=.." "ENV.." ...
# every arm: bench.py --steps_only (200 steps at 720p, 40 at 1080p), twice; plus a per-launch timeline of the arm when TIMELINE=1
TAG=$1; CFG=$2; shift 2
mkdir -p gpurun_out
N=200; [ "$CFG" != c1 ] && N=40
i=0
for arm in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    r=$(env $arm python bench.py --config $CFG --steps_only --steps $N 2>/dev/null | tail -1)
    echo "arm$i [$arm] rep$rep: $r" | tee -a gpurun_out/${TAG}_ab.txt
  done
  if [ "${TIMELINE:-0}" = 1 ]; then env $arm tools/step_timeline.sh $CFG $PWD/gpurun_out/${TAG}_timeline_arm$i.md > /dev/null 2>&1; fi
done
