#!/usr/bin/env bash
# Run a GPU-suite command N times on one lease; every run keeps its full log and breadcrumb trail under gpurun_out/.
# usage: tools/suite_repeat.sh TAG N [pytest target, default tests/] [extra env assignments...]
TAG=$1; N=$2; TARGET=${3:-tests/}; shift 3 || shift $#
mkdir -p gpurun_out
ulimit -c 0
for i in $(seq 1 $N); do
  export BNERV_TEST_TRAIL=$PWD/gpurun_out/${TAG}_trail_$i.txt
  rm -f $BNERV_TEST_TRAIL
  env "$@" timeout 900 python3 -m pytest $TARGET -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc last=$(grep START $BNERV_TEST_TRAIL | tail -1)" | tee -a gpurun_out/${TAG}_summary.txt
  if [ $rc != 0 ]; then grep -v "^  File\|^python(" gpurun_out/${TAG}_run_$i.log | tail -60 | cut -c1-400; fi
done
