#!/usr/bin/env bash
# Same-lease A/B of this tree against the round-5 tree (_variants/r05tree: `git archive ef9b0f9` + its own build): captured steps of every config,
# twice each, alternating.  usage (through gpurun, repo root): tools/ab_r05.sh [out]
OUT=${1:-gpurun_out/r06_ab_vs_r05.txt}
: > $OUT
for rep in 1 2; do
  for cfg in c1 c4 c3 c5; do
    n=40; [ $cfg = c1 ] && n=200
    a=$(cd _variants/r05tree && python bench.py --config $cfg --steps $n --warmup 20 --steps_only 2>/dev/null | tail -1)
    b=$(python bench.py --config $cfg --steps $n --warmup 20 --steps_only 2>/dev/null | tail -1)
    echo "$cfg rep$rep r05: $a" | tee -a $OUT
    echo "$cfg rep$rep r06: $b" | tee -a $OUT
  done
done
