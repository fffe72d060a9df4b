#!/usr/bin/env bash
# Round-4 first GPU pass: full GPU suite, bench line, PMC of the paired backward kernels, the full recipe twice.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/r04a_pytest.txt
for k in pair_dk2s pair_dk3s pair_dk1 k2s; do python tools/pmc_traffic.py $k r04 > /dev/null 2> $O/r04_pmc_$k.err; done
cp $O/r04_traffic_*.json $R/profiles/ 2>/dev/null
python bench.py > $O/r04a_bench_c1.json 2> $O/r04a_bench_c1.err
python tools/recipe_record.py 300 r04 > $O/r04_recipe.log 2>&1
tail -3 $O/r04a_pytest.txt; tail -c 600 $O/r04_recipe.log
