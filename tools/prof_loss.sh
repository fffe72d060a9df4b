#!/usr/bin/env bash
# per-kernel times of the loss path (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loss -- python $R/tools/kloss.py "$@" > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/prof_loss 12 40 | grep -v "at::native" | awk -F'|' 'NR>2{printf "%6s/call %8s us  %s\n", $4, $6, $2}'
rm -rf gpurun_out/prof_loss
